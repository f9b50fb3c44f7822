// Table gradients without float atomics (gfx950).  Contract: include/kpgnn.h, kpgnn_table_grad.
//
//   gtable_t[c,:]  = sum over active pairs (i,k) of table t with code c of g[i,k,:]          (edge-code tables)
//   gdict[u,:]     = sum over (i,k) with uid[i,k] == u of theta[k,:] * gh[i,:]               (peripheral dictionary)
//
// Why a separate kernel: LDS float atomics (ds_add_f32) per gathered edge row made the backward gather 5x slower
// than the gather itself (a handful of hot codes serialise), and per-entry LDS atomics on the ~25 hot dictionary rows
// alone cost 56 us of a 122 us launch (round 2: ~45 cycles of the CU's LDS pipe per wave-wide float atomic).
//
// Design: a block owns a tile of nodes_per_tile nodes; g of the tile ([NT*K][D], contiguous) is copied to LDS with 16-B
// loads that were requested one tile ahead.  The tile's entries arrive SORTED by accumulator row - the edge list by
// (table, code) from kpgnn_csr_build, the dictionary list by uid from kpgnn_dict_tile_pack - and each of the eight
// waves walks a contiguous chunk of each list: lane = two feature columns (one for odd D), the entries are wave-uniform
// (one coalesced load, then v_readlane), a run of equal rows is summed in registers and leaves once when the row
// changes.  Rows whose run lies inside a chunk belong to that wave alone in this tile: the run is added to the block's
// LDS accumulator table without any race.  Only the first run of a chunk can continue a row of the previous wave; it
// is parked in a per-wave LDS slot instead, and after the tile's barrier the wave that OWNS the row (the first one that
// holds it) adds the slots of its followers in wave order.  Every sum therefore has one fixed order: results are
// bitwise reproducible (the replay == eager test relies on it), and no float atomic is contended.
// Per-block tables go to a workspace slab with plain stores; a second launch adds the slabs in block order.
#include "kpgnn_common.h"

namespace kpgnn {
namespace {

constexpr int kWavesTG = 8;                 // walker waves per block
constexpr int kThreadsTG = kWave * kWavesTG;
constexpr int kMaxRows = 64;                // NT*K: at most 8 nodes x 8 hops per tile
constexpr int kSlotRows = 2 * kWavesTG;     // first-run slots: [list][wave]

struct TgParams {
    const int32_t* n_dyn;
    int N, K, D, NT, n0, nk, U, dict_src, KD;
    const int32_t* tptr;
    const uint32_t* tpack;
    const float* g;
    // peripheral dictionary (optional): gdict[uid[i*uid_stride + k]] += theta[k,:] * gh[i,:]   (dict_src 1; 2: g rows)
    const int32_t* uid; int64_t uid_stride;
    const uint32_t* dpack;   // [tiles][64] uid<<8 | node<<3 | hop sorted by uid (0xFFFFFFFF = none)
    const float* theta;
    const float* gh;
    float* slab;             // [gridDim.x][n0 + nk + U][D]
    // FUSE (combine backward computed here instead of read from g): S, the dictionary for the theta gradient, outputs
    const float* f_pre;      // [N,K,D] S saved by the forward
    const float* f_ptab; const int32_t* f_uid; int64_t f_uid_stride; int f_U;   // P = ptab[uid] (theta gradient only)
    float* f_g;              // dL/dS out: element (i, k, c) at i * f_g_sn + k * f_g_sk + c
    int64_t f_g_sn, f_g_sk;  //   ([N,K,D] contiguous, or hop-major [K][N][D]: one contiguous [N,D] slab per hop for agg_bwd's gather)
    float* f_gth;            // [gridDim.x][K][D] theta-gradient partials, or NULL
};

__device__ __forceinline__ void gelu_bwd2(float s, float gv, float& a, float& gg) {   // a = gelu(s), gg = gv * gelu'(s)
    float e2;
    const float cdf = 0.5f * (1.0f + fast_erf(s * 0.70710678118654752440f, &e2));
    a = s * cdf;
    gg = gv * (cdf + s * (e2 * 0.39894228040143267794f));
}

template <int CPL> struct Cols;
template <> struct Cols<1> {
    float a;
    __device__ __forceinline__ void zero() { a = 0.f; }
    __device__ __forceinline__ void load(const float* q) { a = *q; }
    __device__ __forceinline__ void store(float* q) const { *q = a; }
    __device__ __forceinline__ void fma(float m, const Cols& v) { a = fmaf(m, v.a, a); }
    __device__ __forceinline__ void fmul(const Cols& x, const Cols& y) { a = fmaf(x.a, y.a, a); }
    __device__ __forceinline__ void add(const Cols& v) { a += v.a; }
};
template <> struct Cols<2> {
    float a, b;
    __device__ __forceinline__ void zero() { a = b = 0.f; }
    __device__ __forceinline__ void load(const float* q) { const float2 v = *reinterpret_cast<const float2*>(q); a = v.x; b = v.y; }
    __device__ __forceinline__ void store(float* q) const { *reinterpret_cast<float2*>(q) = make_float2(a, b); }
    __device__ __forceinline__ void fma(float m, const Cols& v) { a = fmaf(m, v.a, a); b = fmaf(m, v.b, b); }
    __device__ __forceinline__ void fmul(const Cols& x, const Cols& y) { a = fmaf(x.a, y.a, a); b = fmaf(x.b, y.b, b); }
    __device__ __forceinline__ void add(const Cols& v) { a += v.a; b += v.b; }
};

// LDS (floats): tile [NT*K*D] | acc [(R + kSlotRows)][AS] | ghs [NT][AS] | ths [8][AS] | meta [2*kWavesTG] (uint32)
// AS = the block's column count (gridDim.y == 1: D).  acc rows R.. are the first-run slots.
// BF: g holds bf16 rows (KPGNN_STORE_BF16); they are widened on the way into the LDS tile, everything behind is fp32.
// FUSE (CPL 2, fp32, one column block): the tile is not copied from g but COMPUTED - the backward of the fused KP-GIN+
// epilogue  g = theta[k] * gh[i] * gelu'(S[i,k])  (kpgnn_combine_bwd's arithmetic): wave w owns hop w, reads its 8 rows of S
// (one 416-byte row per request, next tile's rows in flight), writes them to g AND to the LDS tile, and keeps the theta
// gradient of its hop in two registers for the whole launch.  g is then never read back for the table gradients.
// WPH1 (FUSE): K >= 5, one wave per hop - the node stride of the compute phase is then a compile-time 1 (with the run-time
// stride the k = 8 launch took 132 instead of 126 us).
template <int CPL, bool VEC4, bool BF = false, bool FUSE = false, bool WPH1 = false>
__global__ void __launch_bounds__(kThreadsTG, 4)   // (HIP: waves per SIMD) two blocks per CU: 128 VGPRs
table_grad_kernel(TgParams p, int AS) {
    p.N = live_rows(p.N, p.n_dyn);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using CV = Cols<CPL>;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);   // wave-uniform: keeps the walk in SALU control flow
    const int lane = threadIdx.x & (kWave - 1);
    const int cb = blockIdx.y * kWave * CPL;          // first column of this block
    const int c = lane * CPL;                         // this lane's column(s) inside the block
    const bool col_ok = cb + c < p.D;                 // (D % CPL == 0: both columns or none)
    const int cc = col_ok ? c : 0;
    const int D = p.D, K = p.K;
    const int R = p.n0 + p.nk + p.U;
    const int tile_floats = p.NT * K * D;
    float* tile = lds;
    float* acc = lds + ((tile_floats + 3) & ~3);
    float* ghs = acc + (R + kSlotRows) * AS;          // FUSE: two buffers of 8 rows (this tile's / the next tile's gh rows)
    float* ths = ghs + (FUSE ? 16 : 8) * AS;
    float* ptl = ths + 8 * AS;                        // FUSE: the dictionary rows (theta gradient)
    uint32_t* meta = reinterpret_cast<uint32_t*>(ptl + (FUSE ? p.f_U * AS : 0));
    for (int i = threadIdx.x; i < (R + kSlotRows) * AS; i += kThreadsTG) acc[i] = 0.f;
    if (FUSE)
        for (int i = threadIdx.x; i < p.f_U * AS; i += kThreadsTG) ptl[i] = p.f_ptab[i];   // (AS == D)
    if (p.dict_src == 1 || FUSE)
        for (int i = threadIdx.x; i < 8 * AS; i += kThreadsTG) {
            const int k = i / AS, q = i - k * AS;
            ths[i] = (k < K && cb + q < D) ? p.theta[k * D + cb + q] : 0.f;
        }
    const int64_t num_tiles = ((int64_t)p.N + p.NT - 1) / p.NT;
    const int64_t total = (int64_t)p.N * K * D;
    // 16 bytes per thread and request: 4 floats, or 8 bf16 (VEC4 then means "K*D % 8 == 0")
    constexpr int kPref = BF ? 2 : 4;
    constexpr int EPL = BF ? 8 : 4;                  // elements per 16-byte load
    const uint16_t* gbf = reinterpret_cast<const uint16_t*>(p.g);
    float4 pref[kPref];
    auto load16 = [&](int64_t i) -> float4 {
        return BF ? *reinterpret_cast<const float4*>(gbf + i) : *reinterpret_cast<const float4*>(p.g + i);
    };
    auto store16 = [&](float* dst, const float4& v) {         // one 16-byte load's worth into the fp32 tile
        if (BF) {
            const uint32_t* w = reinterpret_cast<const uint32_t*>(&v);
            float4 lo, hi;
            lo.x = __uint_as_float(w[0] << 16); lo.y = __uint_as_float(w[0] & 0xFFFF0000u);
            lo.z = __uint_as_float(w[1] << 16); lo.w = __uint_as_float(w[1] & 0xFFFF0000u);
            hi.x = __uint_as_float(w[2] << 16); hi.y = __uint_as_float(w[2] & 0xFFFF0000u);
            hi.z = __uint_as_float(w[3] << 16); hi.w = __uint_as_float(w[3] & 0xFFFF0000u);
            *reinterpret_cast<float4*>(dst) = lo;
            *reinterpret_cast<float4*>(dst + 4) = hi;
        } else {
            *reinterpret_cast<float4*>(dst) = v;
        }
    };
    // FUSE: the 8 rows (node n, hop w) of S of a tile, two columns per lane - they live in the same 16 registers as pref[]
    auto load_s_rows = [&](int64_t t2, int k2, int sub, int step) {
        float2* sp = reinterpret_cast<float2*>(pref);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = sub + j * step;
            const int64_t node = t2 * p.NT + n;
            sp[j] = make_float2(0.f, 0.f);
            if (t2 < num_tiles && n < p.NT && node < p.N && k2 < K && col_ok)
                sp[j] = *reinterpret_cast<const float2*>(p.f_pre + (node * K + k2) * (int64_t)D + c);
        }
    };
    if (FUSE) {
        int kp = 1; while (kp < K) kp <<= 1;
        const int wph0 = WPH1 ? 1 : kWavesTG / kp;
        load_s_rows(blockIdx.x, w / wph0, w % wph0, wph0);
    } else if (VEC4) {
#pragma unroll
        for (int q = 0; q < kPref; ++q) {
            const int64_t i = (int64_t)blockIdx.x * tile_floats + (q * kThreadsTG + threadIdx.x) * EPL;
            pref[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (blockIdx.x < num_tiles && i < total && (q * kThreadsTG + threadIdx.x) * EPL < tile_floats)
                pref[q] = load16(i);
        }
    }
    // Per-tile metadata travels ahead in registers: the entry-list window (tile_ptr) of tile i+2 and, from the window
    // that arrived an iteration ago, this wave's entries / dictionary entries / gh row of tile i+1, so that no dependent
    // global round trip (tile_ptr -> tile_pack) sits in front of a walk.
    struct TileMeta { int beg, end; uint32_t nxt, dnx; CV gh; int fu; };
    auto load_range = [&](int64_t t2) -> int {       // lane 0: begin of the tile's window, lane 1: its end
        int v = 0;
        if (p.tptr && t2 < num_tiles && lane < 2) v = p.tptr[t2 + lane];
        return v;
    };
    auto load_meta = [&](int64_t t2, int rb, int re, TileMeta& m) {
        // this wave's contiguous chunk of the sorted entry list
        const int per = (re - rb + kWavesTG - 1) / kWavesTG;
        m.beg = min(re, rb + w * per);
        m.end = min(re, m.beg + per);
        m.nxt = (m.beg + lane < m.end) ? p.tpack[m.beg + lane] : 0xFFFFFFFFu;   // hop 63 == skip
        m.dnx = 0xFFFFFFFFu;                          // lanes 0..7: this wave's eight dictionary entries of the tile
        if (p.U > 0 && t2 < num_tiles && lane < 8) {
            m.dnx = p.dpack[t2 * 64 + w * 8 + lane];
        }
        m.gh.zero();                                  // wave w stages the gh row of the tile's node w
        // (FUSE: of the tile AFTER t2 - the compute phase of a tile needs all 8 rows in LDS before its first barrier, so they
        //  are stored one tile early into the other of two buffers)
        const int64_t tg = FUSE ? t2 + gridDim.x : t2;
        const int64_t node = tg * p.NT + w;
        if ((p.dict_src == 1 || FUSE) && w < p.NT && tg < num_tiles && node < p.N && col_ok) m.gh.load(p.gh + node * D + cb + c);
        m.fu = 0;                                     // FUSE: lane n < 8: dictionary id of (node n of tile t2, this wave's hop)
        if (FUSE && p.f_uid && lane < p.NT && t2 < num_tiles && t2 * p.NT + lane < p.N) {
            int kp = 1; while (kp < K) kp <<= 1;
            const int k2 = WPH1 ? w : w / (kWavesTG / kp);
            if (k2 < K) m.fu = p.f_uid[(t2 * p.NT + lane) * p.f_uid_stride + k2];
        }
    };
    TileMeta cm;
    int rng1;
    {
        const int rng0 = load_range(blockIdx.x);
        load_meta(blockIdx.x, __builtin_amdgcn_readlane(rng0, 0), __builtin_amdgcn_readlane(rng0, 1), cm);
        rng1 = load_range((int64_t)blockIdx.x + gridDim.x);
    }
    // FUSE: wave w computes hop fk for the nodes n = fsub, fsub + wph, ... of a tile; with K <= 4 hops several waves share a
    // hop (wph = 8 / pow2ceil(K)) so that the compute phase keeps all waves busy for the early layers too
    int wph = 1;
    if (FUSE && !WPH1) { int kp = 1; while (kp < K) kp <<= 1; wph = kWavesTG / kp; }
    const int fk = w / wph, fsub = w - fk * wph;
    const bool fwave = FUSE && fk < K;
    int gbuf = 0;                                     // FUSE: ghs buffer that holds the CURRENT tile's gh rows
    float gth_a = 0.f, gth_b = 0.f;                   // FUSE: theta gradient of hop w, this lane's two columns
    float th_a = 0.f, th_b = 0.f;
    if (FUSE) {
        const int64_t node = (int64_t)blockIdx.x * p.NT + w;
        CV g0;
        g0.zero();
        if (w < p.NT && blockIdx.x < num_tiles && node < p.N && col_ok) g0.load(p.gh + node * D + c);
        if (w < p.NT && col_ok) g0.store(ghs + w * AS + cc);
        if (fwave && col_ok) { th_a = p.theta[fk * D + c]; th_b = p.theta[fk * D + c + 1]; }
        __syncthreads();
    }
    // what a finished walk leaves for the boundary pass: the last run of each list (registers) and the packed chunk
    // description  nonempty | whole << 1 | first_row << 2 | last_row << 14
    CV lastE, lastD;
    uint32_t myE = 0, myD = 0;
    lastE.zero(); lastD.zero();
    // The boundary pass of the PREVIOUS tile: runs between that tile's closing barrier and this tile's walk.
    auto boundary = [&](int list, uint32_t mine, const CV& last, uint32_t metav) {
        if (!(mine & 1)) return;
        const bool whole = mine & 2;
        const int first_row = (mine >> 2) & 0xFFF, last_row = (mine >> 14) & 0xFFF;
        bool owned = false;                           // does an earlier wave hold my first row?
        for (int w2 = w - 1; w2 >= 0; --w2) {
            const uint32_t m = __builtin_amdgcn_readlane(metav, list * kWavesTG + w2);
            if (!(m & 1)) continue;
            owned = (int)((m >> 14) & 0xFFF) == first_row;
            break;
        }
        // the row whose followers I add up: my last run, or my only run when nobody before me holds it
        const bool has_tail = !whole || !owned;
        const int tail_row = whole ? first_row : last_row;
        uint32_t take = 0;                            // later waves whose parked first run continues that row
        if (has_tail)
            for (int w2 = w + 1; w2 < kWavesTG; ++w2) {
                const uint32_t m = __builtin_amdgcn_readlane(metav, list * kWavesTG + w2);
                if (!(m & 1)) continue;
                if ((int)((m >> 2) & 0xFFF) != tail_row) break;
                take |= 1u << w2;
                if (!(m & 2)) break;
            }
        // all LDS reads first (independent), then the sums in wave order, then the writes: the rows and slots touched here
        // are this wave's alone until the next barrier
        float* slots = acc + (R + list * kWavesTG) * AS + cc;
        CV own, afirst, atail, z;
        z.zero(); own.zero();
        if (!owned) own.load(slots + w * AS);
        const bool first_alone = !owned && !whole;    // my first run ended inside my chunk: nobody else holds its row
        if (first_alone) afirst.load(acc + first_row * AS + cc);
        if (has_tail) atail.load(acc + tail_row * AS + cc);
        CV tot = last;
        if (whole) tot = own;
#pragma unroll
        for (int j0 = 1; j0 < kWavesTG; j0 += 4) {    // followers in groups of four (register budget)
            if ((take >> (w + j0)) == 0) break;
            CV fv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (w + j0 + j < kWavesTG && ((take >> (w + j0 + j)) & 1)) fv[j].load(slots + (w + j0 + j) * AS);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (w + j0 + j < kWavesTG && ((take >> (w + j0 + j)) & 1)) {
                    tot.add(fv[j]);
                    if (col_ok) z.store(slots + (w + j0 + j) * AS);
                }
        }
        if (!owned && col_ok) z.store(slots + w * AS);
        if (first_alone) { afirst.add(own); if (col_ok) afirst.store(acc + first_row * AS + cc); }
        if (has_tail) {
            atail.add(tot);
            if (col_ok) atail.store(acc + tail_row * AS + cc);
        }
    };
    bool have_prev = false;
    for (int64_t tl = blockIdx.x; tl < num_tiles; tl += gridDim.x) {
        // Consume this tile's metadata NOW (requested an iteration ago): the wait then sits in front of the loads issued
        // below for the next tile.
        uint32_t mine = cm.nxt, dmine = cm.dnx;
        CV ghv = cm.gh;
        asm volatile("" : "+v"(mine), "+v"(dmine));
        TileMeta nm;
        load_meta(tl + gridDim.x, __builtin_amdgcn_readlane(rng1, 0), __builtin_amdgcn_readlane(rng1, 1), nm);
        rng1 = load_range(tl + 2 * (int64_t)gridDim.x);
        const int beg = cm.beg, end = cm.end;
        const int64_t base = tl * tile_floats;
        const int nfl = (int)min((int64_t)tile_floats, total - base);
        __syncthreads();                             // previous tile fully walked, its chunk descriptions are in LDS
        if (have_prev) {
            const uint32_t metav = lane < 2 * kWavesTG ? meta[lane] : 0u;
            boundary(0, myE, lastE, metav);
            boundary(1, myD, lastD, metav);
        }
        have_prev = true;
        if (FUSE) {
            // next tile's gh row of this wave -> the other buffer (its last readers were the previous tile's walk)
            if (w < p.NT && col_ok) ghv.store(ghs + ((gbuf ^ 1) * 8 + w) * AS + cc);
            const int fuv = cm.fu;
            const float2* sp = reinterpret_cast<const float2*>(pref);
            if (fwave) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int n = fsub + j * wph;
                    const int64_t node = tl * p.NT + n;
                    if (n >= p.NT || node >= p.N) break;                       // (wave-uniform)
                    const float2 gh2 = *reinterpret_cast<const float2*>(ghs + (gbuf * 8 + n) * AS + cc);
                    const float2 s2 = sp[j];
                    float a0, a1, g0, g1;
                    gelu_bwd2(s2.x, th_a * gh2.x, a0, g0);
                    gelu_bwd2(s2.y, th_b * gh2.y, a1, g1);
                    float2 pr = make_float2(0.f, 0.f);
                    if (p.f_uid) pr = *reinterpret_cast<const float2*>(ptl + __builtin_amdgcn_readlane(fuv, n) * AS + cc);
                    gth_a = fmaf(gh2.x, a0 + pr.x, gth_a);
                    gth_b = fmaf(gh2.y, a1 + pr.y, gth_b);
                    if (col_ok) {
                        *reinterpret_cast<float2*>(p.f_g + node * p.f_g_sn + fk * p.f_g_sk + c) = make_float2(g0, g1);
                        *reinterpret_cast<float2*>(tile + (n * K + fk) * D + c) = make_float2(g0, g1);
                    }
                }
            }
            load_s_rows(tl + gridDim.x, fk, fsub, wph);
            gbuf ^= 1;
        } else if (VEC4) {
            // the first kPref*2048 floats of the tile were prefetched into registers during the previous walk
#pragma unroll
            for (int q = 0; q < kPref; ++q) {
                const int i = (q * kThreadsTG + threadIdx.x) * EPL;
                if (i < nfl) store16(tile + i, pref[q]);
            }
            for (int i = (kPref * kThreadsTG + threadIdx.x) * EPL; i < nfl; i += kThreadsTG * EPL)
                store16(tile + i, load16(base + i));
            const int64_t nb = base + (int64_t)gridDim.x * tile_floats;
#pragma unroll
            for (int q = 0; q < kPref; ++q) {
                const int64_t i = nb + (q * kThreadsTG + threadIdx.x) * EPL;
                if (tl + gridDim.x < num_tiles && i < total && (q * kThreadsTG + threadIdx.x) * EPL < tile_floats)
                    pref[q] = load16(i);
            }
        } else {
            for (int i = threadIdx.x; i < nfl; i += kThreadsTG)
                tile[i] = BF ? __uint_as_float((uint32_t)gbf[base + i] << 16) : p.g[base + i];
        }
        if (!FUSE && p.dict_src == 1 && w < p.NT && col_ok) ghv.store(ghs + w * AS + cc);
        __syncthreads();
        // ---- walk this wave's chunk of the (table,code)-sorted pair list.  Every lane decodes ITS entry once (VALU, 64
        //      entries per instruction); per entry the wave then pays three v_readlane, one LDS read, a compare and the fma.
        //      The LDS reads of 8 entries are issued back to back before the (sequential, register-only) run accumulation.
        {
            int vmul, voff, vrow;
            auto decode = [&](uint32_t e) {
                const int vhop = e & 0x3F;
                const bool vok = vhop < K;
                vmul = __float_as_int((float)(((e >> 6) & 0x3F) + 1));       // multiplicity of the merged entry
                voff = vok ? ((int)((e >> 12) & 7) * K + vhop) * D + cb : 0;
                const int vcc = (int)(e >> 15);                               // table<<16 | code
                vrow = vok ? ((vcc >> 16) ? p.n0 + (vcc & 0xFFFF) : vcc) : -1;
            };
            decode(mine);
            int cur = -1, first_row = 0;
            bool first = true;
            CV run;
            run.zero();
            // A finished run is added to its row with a plain read-modify-write (the row is this wave's alone until the
            // next barrier).  The read is issued when the run ends and the add + write when the NEXT run ends, so that the
            // LDS latency hides behind the next run instead of stalling the walk.  The chunk's first run is parked in this
            // wave's slot with a plain store (the slot's reader left it zero).  (ds_add_f32 is no alternative: ~150 cycles
            // of the CU's LDS pipe per wave-wide float atomic, contended or not; batching the read-modify-writes of a
            // group of entries costs more registers than the kernel has at two blocks per CU - DESIGN.md.)
            bool pending = false;
            float* pq = acc;
            CV pold, psum;
            auto settle = [&]() {
                if (pending) { pold.add(psum); if (col_ok) pold.store(pq); pending = false; }
            };
            auto leave = [&]() {                     // the finished run of row `cur`
                if (first) {
                    first_row = cur; first = false;
                    if (col_ok) run.store(acc + (R + w) * AS + cc);
                } else {
                    settle();
                    pq = acc + cur * AS + cc;
                    pold.load(pq);
                    psum = run;
                    pending = true;
                }
            };
            for (int b0 = beg; b0 < end; b0 += 64) {
                const int cnt = min(64, end - b0);
                for (int e0 = 0; e0 < cnt; e0 += 8) {
                    int off[8], row[8]; float mul[8]; CV val[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        off[u] = __builtin_amdgcn_readlane(voff, e0 + u);
                        row[u] = __builtin_amdgcn_readlane(vrow, e0 + u);
                        mul[u] = __int_as_float(__builtin_amdgcn_readlane(vmul, e0 + u));
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) val[u].load(tile + off[u] + cc);
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (row[u] >= 0) {
                            if (row[u] != cur) {                           // wave-uniform
                                if (cur >= 0) leave();
                                cur = row[u];
                                run.zero();
                            }
                            run.fma(mul[u], val[u]);
                        }
                    }
                }
                if (b0 + 64 < end)                   // (rare: > 64 entries in a wave's chunk)
                    decode((b0 + 64 + lane < end) ? p.tpack[b0 + 64 + lane] : 0xFFFFFFFFu);
            }
            myE = 0;
            if (cur >= 0) {
                const bool whole = first;
                if (whole) leave();                  // a single run: it is the chunk's first run -> parked
                settle();
                myE = 1u | (whole ? 2u : 0u) | ((uint32_t)first_row << 2) | ((uint32_t)cur << 14);
                lastE = run;
            }
        }
        // ---- peripheral dictionary: this wave's eight entries of the tile's uid-sorted list
        if (p.U > 0) {
            const bool dok = dmine != 0xFFFFFFFFu && (int)(dmine & 7) < K;
            const int dn = (dmine >> 3) & 7, dk = dmine & 7;
            const int drow = dok ? p.n0 + p.nk + (int)(dmine >> 8) : -1;
            // gh row / g row of the tile (FUSE: the tile's gh rows sit in the buffer the compute phase just read - gbuf was
            // toggled behind it)
            const int doa = p.dict_src == 1 ? ((FUSE ? (gbuf ^ 1) * 8 : 0) + dn) * AS : (dn * K + dk) * D + cb;
            const int dob = dk * AS;                                              // theta row
            int cur = -1, first_row = 0;
            bool first = true;
            CV run;
            run.zero();
            bool pending = false;
            float* pq = acc;
            CV pold, psum;
            auto settle = [&]() {
                if (pending) { pold.add(psum); if (col_ok) pold.store(pq); pending = false; }
            };
            auto leave = [&]() {
                if (first) {
                    first_row = cur; first = false;
                    if (col_ok) run.store(acc + (R + kWavesTG + w) * AS + cc);
                } else {
                    settle();
                    pq = acc + cur * AS + cc;
                    pold.load(pq);
                    psum = run;
                    pending = true;
                }
            };
            int row[8]; CV va[8], vb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                row[u] = __builtin_amdgcn_readlane(drow, u);
                const int oa = __builtin_amdgcn_readlane(doa, u);
                if (p.dict_src == 1) {
                    va[u].load(ghs + oa + cc);
                    vb[u].load(ths + __builtin_amdgcn_readlane(dob, u) + cc);
                } else {
                    va[u].load(tile + oa + cc);
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (row[u] >= 0) {
                    if (row[u] != cur) {
                        if (cur >= 0) leave();
                        cur = row[u];
                        run.zero();
                    }
                    if (p.dict_src == 1) run.fmul(va[u], vb[u]);
                    else run.add(va[u]);
                }
            }
            myD = 0;
            if (cur >= 0) {
                const bool whole = first;
                if (whole) leave();
                settle();
                myD = 1u | (whole ? 2u : 0u) | ((uint32_t)first_row << 2) | ((uint32_t)cur << 14);
                lastD = run;
            }
        }
        if (lane == 0) { meta[w] = myE; meta[kWavesTG + w] = myD; }
        cm = nm;
    }
    __syncthreads();
    if (have_prev) {
        const uint32_t metav = lane < 2 * kWavesTG ? meta[lane] : 0u;
        boundary(0, myE, lastE, metav);
        boundary(1, myD, lastD, metav);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < R * AS; i += kThreadsTG) {
        const int r = i / AS, q = i - r * AS;
        if (cb + q < D) p.slab[((int64_t)blockIdx.x * R + r) * D + cb + q] = acc[i];
    }
    if (FUSE && p.f_gth && fwave && col_ok)      // slab row = (block, sub-wave of the hop): gridDim.x * wph partial [K,D] tables
        *reinterpret_cast<float2*>(p.f_gth + (((int64_t)blockIdx.x * wph + fsub) * K + fk) * D + c) = make_float2(gth_a, gth_b);
}

// ------------------------------------------------------------------------------------------------ fused combine backward + table gradients on the matrix cores
// The walk above costs ~3.4 us per tile of 64 (node, hop) rows - a sequential, scalar-issue-bound pass over the tile's sorted
// entry list - and with it the fused kernel sat at 0.22-0.26 of the HBM roofline, 60 % of its compute phase waiting for rows
// of S it had no registers left to request earlier.  The same sums are a small matrix product per tile,
//     T[c, :] += sum_r C[c, r] * g[r, :],     C[c, r] = number of pairs of row r = (hop, node) that carry code c,
// C a 64 x 64 matrix of small integers (exact in bf16) and g fp32.  g is split three ways into bf16 pieces
// (hi + mid + lo carry 24 mantissa bits), so C x g runs on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: 12 MFMAs per wave
// and tile instead of the walk, products exact, one fixed summation order (bitwise reproducible), error that of the fp32 fmaf
// chain (scripts/ubench/ub_mfma_bf16_counts.hip: 1.6e-7 of sum |terms| against 1.7e-7).  The count matrix is filled with integer
// LDS adds straight from the tile's entry list - in ANY order: the list no longer has to be sorted for this kernel.
// Wave w computes hop w of the tile (g = theta[w] * gh[i] * gelu'(S[i,w]): written to global memory for the transposed gather
// and, split, to LDS as MFMA B-operands); waves (mt, nt) = (w / 4, w % 4) own the 32 x 32 block of the 64 x 128 accumulator
// table for the whole launch (16 registers), written to the slab once at the end.
typedef __attribute__((ext_vector_type(8))) __bf16 tg_bf16x8;
typedef __attribute__((ext_vector_type(16))) float tg_f32x16;
constexpr int kCntPitch = 72;      // bytes per code row of the count matrix: 64 rows + 8 (stride of 18 words: conflict-free 8-byte reads)

__device__ __forceinline__ void tg_split3(const float (&x)[8], tg_bf16x8& hi, tg_bf16x8& mid, tg_bf16x8& lo) {
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        const __bf16 h = (__bf16)x[n];
        const float r1 = x[n] - (float)h;
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;
        hi[n] = h; mid[n] = m; lo[n] = (__bf16)r2;
    }
}

// SUB = 8 / pow2ceil(K) tiles of 8 nodes form one SUPER-TILE of 64 (hop, node) rows whatever K is (K = 1: 64 nodes, K = 2: 32,
// K = 3, 4: 16, K >= 5: 8), so that the early layers of a KP-GIN+ stack (few hops) do not pay two barriers for 8 rows.  Wave w
// computes the 8 nodes of sub-tile w % SUB at hop w / SUB = row block w of the B planes; count-matrix row of an entry =
// hop * 8 SUB + 8 sub + node_in_tile, its sub-tile read off the list windows of the super-tile's tiles.
// BF (KPGNN_STORE_BF16): S arrives and dL/dS leaves as bf16 rows (same element strides).  The table gradients are taken from
// the ROUNDED dL/dS - the values the transposed gather will read - so g is one exact bf16 piece: one MFMA per k-step instead of
// three.
template <int SUB, bool BF>
__global__ void __launch_bounds__(kThreadsTG, 4)   // two blocks per CU: 128 VGPRs
tg_fuse_mfma_kernel(TgParams p) {
    p.N = live_rows(p.N, p.n_dyn);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int lane = threadIdx.x & (kWave - 1);
    const int tid = threadIdx.x;
    const int D = p.D, K = p.K, R = p.n0 + p.nk;
    const int c = lane * 2;
    const bool col_ok = c < D;                       // D even: both columns or none
    const int cc = col_ok ? c : 0;
    const int PC = D + 1;                            // 16-byte items per row block of a plane; item D stays zero
    uint4* planes = reinterpret_cast<uint4*>(lds);                                   // [3][8][PC]
    uint8_t* cnt = reinterpret_cast<uint8_t*>(planes + 3 * 8 * PC);                  // [2][64][kCntPitch]
    float* ptl = reinterpret_cast<float*>(cnt + 2 * 64 * kCntPitch);                 // [f_U][D]
    float* ghs = ptl + p.f_U * D;                                                    // SUB == 1: [2][8][D] staged gh rows
    for (int i = tid; i < 3 * 8 * PC; i += kThreadsTG) planes[i] = make_uint4(0u, 0u, 0u, 0u);
    for (int i = tid; i < 2 * 64 * kCntPitch / 4; i += kThreadsTG) reinterpret_cast<uint32_t*>(cnt)[i] = 0u;
    for (int i = tid; i < p.f_U * D; i += kThreadsTG) ptl[i] = p.f_ptab[i];
    const int64_t num_tiles = ((int64_t)p.N + 7) / 8;
    const int64_t num_super = (num_tiles + SUB - 1) / SUB;
    const int G = gridDim.x;
    const int hop = w / SUB, sub = w % SUB;          // this wave's hop and sub-tile
    const bool fwave = hop < K;
    // ---- travelling state (requested one super-tile ahead): S rows and gh rows of the wave's 8 nodes, the dictionary ids, the
    //      super-tile's entries and the list windows of its tiles
    float2 sp[8], ghp[8];
    auto load_rows = [&](int64_t st) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t node = (st * SUB + sub) * 8 + j;
            sp[j] = make_float2(0.f, 0.f); ghp[j] = sp[j];
            if (st < num_super && node < p.N && fwave && col_ok) {
                if (BF) {
                    const uint32_t t = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint16_t*>(p.f_pre) + (node * K + hop) * (int64_t)D + c);
                    sp[j] = make_float2(__uint_as_float(t << 16), __uint_as_float(t & 0xFFFF0000u));
                } else {
                    sp[j] = *reinterpret_cast<const float2*>(p.f_pre + (node * K + hop) * (int64_t)D + c);
                }
                if (SUB > 1) ghp[j] = *reinterpret_cast<const float2*>(p.gh + node * D + c);
            }
        }
    };
    // SUB == 1: the 8 gh rows of a tile serve all eight hop waves: wave w fetches node w's row (a tile ahead) and the block reads
    // them from LDS - eight times less L1 / L2 traffic than every wave fetching its own copies (95 -> 89 us at k = 8)
    auto load_gh = [&](int64_t st) -> float2 {
        const int64_t node = st * 8 + w;
        float2 v = make_float2(0.f, 0.f);
        if (SUB == 1 && st < num_super && node < p.N && col_ok) v = *reinterpret_cast<const float2*>(p.gh + node * D + c);
        return v;
    };
    auto load_fu = [&](int64_t st) -> int {          // lane n < 8: dictionary id of (node n of the wave's sub-tile, its hop)
        int v = 0;
        const int64_t node = (st * SUB + sub) * 8 + lane;
        if (p.f_uid && lane < 8 && fwave && st < num_super && node < p.N) v = p.f_uid[node * p.f_uid_stride + hop];
        return v;
    };
    auto load_win = [&](int64_t st) -> int {         // lane l <= SUB: first entry of tile st * SUB + l (clamped to the list's end)
        int v = 0;
        if (lane <= SUB && st < num_super) {
            const int64_t t = st * SUB + lane;
            v = p.tptr[t < num_tiles ? t : num_tiles];
        }
        return v;
    };
    auto load_ent = [&](int b, int e) -> uint32_t { return (b + tid < e) ? p.tpack[b + tid] : 0xFFFFFFFFu; };   // hop 63 == none
    // count-matrix cell of entry number i (value e) of a super-tile whose tile windows are `win`: byte offset, or -1
    auto cell_of = [&](uint32_t e, int i, int win) -> int {
        const int eh = (int)(e & 0x3Fu);
        if (eh >= K) return -1;
        int s2 = 0;
#pragma unroll
        for (int b = 1; b < SUB; ++b) s2 += (i >= __builtin_amdgcn_readlane(win, b)) ? 1 : 0;
        const int vcc = (int)(e >> 15);                                   // table << 16 | code
        const int row = (vcc >> 16) ? p.n0 + (vcc & 0xFFFF) : vcc;
        return row * kCntPitch + eh * (8 * SUB) + s2 * 8 + (int)((e >> 12) & 7u);
    };
    int64_t st = blockIdx.x;
    int wcur = load_win(st), wnext = load_win(st + G);
    uint32_t ecur = load_ent(__builtin_amdgcn_readlane(wcur, 0), __builtin_amdgcn_readlane(wcur, SUB));
    load_rows(st);
    int fucur = load_fu(st);
    float2 ghv = load_gh(st);
    if (SUB == 1 && col_ok) *reinterpret_cast<float2*>(ghs + w * D + c) = ghv;         // buffer 0: this tile's gh rows
    ghv = load_gh(st + G);
    int gbuf = 0;
    float th_a = 0.f, th_b = 0.f, gth_a = 0.f, gth_b = 0.f;
    if (fwave && col_ok) { th_a = p.theta[hop * D + c]; th_b = p.theta[hop * D + c + 1]; }
    tg_f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    const int mt = w >> 2, nt = w & 3, li = lane & 31, kg = lane >> 5;
    int cbuf = 0;
    uint32_t eprev = 0xFFFFFFFFu;
    int wprev = 0;                                   // previous super-tile's windows (its cells are cleared while this one's are filled)
    __syncthreads();
    // (Tried and dropped, all for a longer distance between a row request and its use: a second register set requested before the
    //  compute phase - 35 spills at 128 registers per lane; refilling node j's registers as soon as node j is computed - 63; a
    //  branch-free compute loop with clamped loads and dummy-slot stores - 32.)
    for (; st < num_super; st += G) {
        uint8_t* cnow = cnt + cbuf * 64 * kCntPitch;
        uint8_t* cold = cnt + (cbuf ^ 1) * 64 * kCntPitch;
        const int cb = __builtin_amdgcn_readlane(wcur, 0), ce = __builtin_amdgcn_readlane(wcur, SUB);
        const int pb = __builtin_amdgcn_readlane(wprev, 0), pe = __builtin_amdgcn_readlane(wprev, SUB);
        // ---- requests for the next super-tile (its entries; the windows of the one after)
        const uint32_t enext = load_ent(__builtin_amdgcn_readlane(wnext, 0), __builtin_amdgcn_readlane(wnext, SUB));
        const int wnext2 = load_win(st + 2 * (int64_t)G);
        // ---- count matrix: clear the cells of the super-tile before, add this one's entries (any order: integer adds)
        {
            int q = cell_of(eprev, pb + tid, wprev);
            if (q >= 0) cold[q] = 0;
            for (int i = pb + kThreadsTG + tid; i < pe; i += kThreadsTG) { q = cell_of(p.tpack[i], i, wprev); if (q >= 0) cold[q] = 0; }
            q = cell_of(ecur, cb + tid, wcur);
            if (q >= 0) atomicAdd(reinterpret_cast<uint32_t*>(cnow) + (q >> 2), (((ecur >> 6) & 0x3Fu) + 1u) << (8 * (q & 3)));
            for (int i = cb + kThreadsTG + tid; i < ce; i += kThreadsTG) {
                const uint32_t e2 = p.tpack[i];
                q = cell_of(e2, i, wcur);
                if (q >= 0) atomicAdd(reinterpret_cast<uint32_t*>(cnow) + (q >> 2), (((e2 >> 6) & 0x3Fu) + 1u) << (8 * (q & 3)));
            }
        }
        // ---- compute phase: g = theta[hop] * gh[i] * gelu'(S[i, hop]) for the wave's 8 nodes
        if (SUB == 1 && col_ok) *reinterpret_cast<float2*>(ghs + ((gbuf ^ 1) * 8 + w) * D + c) = ghv;     // next tile's gh row
        if (fwave) {
            float ga[8], gb[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int64_t node = (st * SUB + sub) * 8 + j;
                ga[j] = 0.f; gb[j] = 0.f;
                if (node < p.N) {                                          // (wave-uniform)
                    float a0, a1;
                    if (SUB == 1) ghp[j] = *reinterpret_cast<const float2*>(ghs + (gbuf * 8 + j) * D + cc);
                    gelu_bwd2(sp[j].x, th_a * ghp[j].x, a0, ga[j]);
                    gelu_bwd2(sp[j].y, th_b * ghp[j].y, a1, gb[j]);
                    float2 pr = make_float2(0.f, 0.f);
                    if (p.f_uid) pr = *reinterpret_cast<const float2*>(ptl + __builtin_amdgcn_readlane(fucur, j) * D + cc);
                    gth_a = fmaf(ghp[j].x, a0 + pr.x, gth_a);
                    gth_b = fmaf(ghp[j].y, a1 + pr.y, gth_b);
                    if (BF) {
                        const uint32_t b0 = f32_to_bf16_bits(ga[j]), b1 = f32_to_bf16_bits(gb[j]);
                        if (col_ok) *reinterpret_cast<uint32_t*>(reinterpret_cast<uint16_t*>(p.f_g) + node * p.f_g_sn + hop * p.f_g_sk + c) = b0 | (b1 << 16);
                        ga[j] = __uint_as_float(b0 << 16); gb[j] = __uint_as_float(b1 << 16);     // what the gather will read
                    } else if (col_ok) {
                        *reinterpret_cast<float2*>(p.f_g + node * p.f_g_sn + hop * p.f_g_sk + c) = make_float2(ga[j], gb[j]);
                    }
                }
            }
            if (col_ok) {                           // row block w as MFMA B operands (fp32: split three ways; bf16: exact as is)
                tg_bf16x8 hi, mid, lo;
                tg_split3(ga, hi, mid, lo);
                planes[(0 * 8 + w) * PC + c] = __builtin_bit_cast(uint4, hi);
                if (!BF) {
                    planes[(1 * 8 + w) * PC + c] = __builtin_bit_cast(uint4, mid);
                    planes[(2 * 8 + w) * PC + c] = __builtin_bit_cast(uint4, lo);
                }
                tg_split3(gb, hi, mid, lo);
                planes[(0 * 8 + w) * PC + c + 1] = __builtin_bit_cast(uint4, hi);
                if (!BF) {
                    planes[(1 * 8 + w) * PC + c + 1] = __builtin_bit_cast(uint4, mid);
                    planes[(2 * 8 + w) * PC + c + 1] = __builtin_bit_cast(uint4, lo);
                }
            }
        }
        load_rows(st + G);
        const int funext = load_fu(st + G);
        ghv = load_gh(st + 2 * (int64_t)G);
        __syncthreads();
        // ---- C x g on the matrix cores: wave (mt, nt) owns codes [32 mt, 32 mt + 32) x columns [32 nt, 32 nt + 32)
        {
            const int col = nt * 32 + li;
            const int pc = col < D ? col : D;       // (item D of every row block is zero)
            const uint8_t* crow = cnow + (mt * 32 + li) * kCntPitch + 8 * kg;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const uint2 cw = *reinterpret_cast<const uint2*>(crow + 16 * ks);
                tg_bf16x8 a;
                a[0] = (__bf16)(float)(cw.x & 0xFFu); a[1] = (__bf16)(float)((cw.x >> 8) & 0xFFu);
                a[2] = (__bf16)(float)((cw.x >> 16) & 0xFFu); a[3] = (__bf16)(float)(cw.x >> 24);
                a[4] = (__bf16)(float)(cw.y & 0xFFu); a[5] = (__bf16)(float)((cw.y >> 8) & 0xFFu);
                a[6] = (__bf16)(float)((cw.y >> 16) & 0xFFu); a[7] = (__bf16)(float)(cw.y >> 24);
                const int rb = 2 * ks + kg;
                if (!BF) {
                    const tg_bf16x8 b2 = __builtin_bit_cast(tg_bf16x8, planes[(2 * 8 + rb) * PC + pc]);
                    const tg_bf16x8 b1 = __builtin_bit_cast(tg_bf16x8, planes[(1 * 8 + rb) * PC + pc]);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b2, acc, 0, 0, 0);     // smallest terms first
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, acc, 0, 0, 0);
                }
                const tg_bf16x8 b0 = __builtin_bit_cast(tg_bf16x8, planes[(0 * 8 + rb) * PC + pc]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0, acc, 0, 0, 0);
            }
        }
        __syncthreads();                             // planes and this count buffer are free again
        eprev = ecur; wprev = wcur;
        ecur = enext; wcur = wnext; wnext = wnext2;
        fucur = funext;
        cbuf ^= 1; gbuf ^= 1;
    }
    // ---- this block's partial tables (C/D map: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5))
    {
        const int col = nt * 32 + li;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int row = mt * 32 + (v & 3) + 8 * (v >> 2) + 4 * kg;
            if (row < R && col < D) p.slab[((int64_t)blockIdx.x * R + row) * D + col] = acc[v];
        }
    }
    if (p.f_gth && fwave && col_ok)                  // slab row = (block, sub-tile wave of the hop): gridDim.x * SUB partial [K,D] tables
        *reinterpret_cast<float2*>(p.f_gth + (((int64_t)blockIdx.x * SUB + sub) * K + hop) * D + c) = make_float2(gth_a, gth_b);
}

// LDS bytes of the kernel above
inline size_t tg_fuse_mfma_lds(int D, int f_U) {
    return (size_t)3 * 8 * (D + 1) * 16 + (size_t)2 * 64 * kCntPitch + sizeof(float) * ((size_t)f_U * D + 16 * (size_t)D);
}

// Dictionary entries of every tile, sorted by dictionary row: pack[tile*64 + j] = uid << 8 | node_in_tile << 3 | hop.
// One wave per tile; 64 keys are ranked by counting (a one-off per batch: the ids are data, not parameters).
__global__ void __launch_bounds__(kWave)
dict_tile_pack_kernel(const int32_t* __restrict__ uid, int64_t uid_stride, int N, int K, int NT, uint32_t* __restrict__ pack) {
    const int lane = threadIdx.x;
    const int64_t tl = blockIdx.x;
    const int n = lane / K, k = lane - n * K;
    const int64_t node = tl * NT + n;
    uint32_t key = 0xFFFFFFFFu;
    if (n < NT && node < N) key = ((uint32_t)uid[node * uid_stride + k] << 8) | (uint32_t)(n << 3 | k);
    int rank = 0;
    for (int j = 0; j < kWave; ++j) {
        const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)key, j);
        rank += (o < key || (o == key && j < lane)) ? 1 : 0;
    }
    pack[tl * 64 + rank] = key;
}

// out[e] = sum_b slab[b][e]   (fixed order: deterministic).  16 outputs x 64 slices of the slab range per WG: each
// thread adds nslab/64 values (independent loads), the 64 partials of an output meet in LDS.  A launch works through up
// to three independent jobs (block ranges one after the other) and a theta-gradient finish.
struct SlabJob {
    const float* slab; int nslab; int nblocks; int acc;    // acc bit i: out[i] += instead of =
    int lo;                                                 // a block covers 1 << lo outputs with 1024 >> lo slices of the slab rows
    int64_t elems;
    float* out[4]; int64_t n[3];                            // outputs [0, n0) -> out[0], [n0, n0+n1) -> out[1], ... rest -> out[3]
};
struct SlabArgs { SlabJob j[3]; ThetaFinish tf; };

__global__ void __launch_bounds__(1024)
slab_reduce_kernel(const SlabArgs a) {
    __shared__ float sm[1168];
    int blk = blockIdx.x;
    int ji = 0;
    while (ji < 3 && blk >= a.j[ji].nblocks) { blk -= a.j[ji].nblocks; ++ji; }
    if (ji == 3) {                      // the last blocks finish a theta-gradient slab (kpgnn_common.h)
        theta_finish_block(a.tf, blk, sm);
        return;
    }
    // (the job is picked with uniform selects, not by indexing the argument struct with a runtime value)
    const SlabJob& J = ji == 0 ? a.j[0] : (ji == 1 ? a.j[1] : a.j[2]);
    const float* __restrict__ slab = J.slab;
    const int nslab = J.nslab;
    const int64_t elems = J.elems;
    // few slab rows (small batches): fewer slices and more outputs per block, instead of 64 slices of which most idle
    const int lo = J.lo, outs = 1 << lo, nsl = 1024 >> lo, pitch = outs + 1;
    const int o = threadIdx.x & (outs - 1), slice = threadIdx.x >> lo;
    const int64_t e = (int64_t)blk * outs + o;
    float s = 0.f;
    if (e < elems) {
        int b = slice;
        for (; b + 7 * nsl < nslab; b += 8 * nsl) {          // eight independent loads in flight, added in row order
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = slab[(int64_t)(b + q * nsl) * elems + e];
#pragma unroll
            for (int q = 0; q < 8; ++q) s += v[q];
        }
        for (; b + 3 * nsl < nslab; b += 4 * nsl) {
            const float v0 = slab[(int64_t)b * elems + e], v1 = slab[(int64_t)(b + nsl) * elems + e];
            const float v2 = slab[(int64_t)(b + 2 * nsl) * elems + e], v3 = slab[(int64_t)(b + 3 * nsl) * elems + e];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; b < nslab; b += nsl) s += slab[(int64_t)b * elems + e];
    }
    sm[slice * pitch + o] = s;
    __syncthreads();
    if (slice == 0 && e < elems) {
        float tot = 0.f;
        for (int q = 0; q < nsl; ++q) tot += sm[q * pitch + o];
        int which = 3;
        int64_t off = e;
        if (off < J.n[0]) which = 0;
        else if ((off -= J.n[0]) < J.n[1]) which = 1;
        else if ((off -= J.n[1]) < J.n[2]) which = 2;
        else off -= J.n[2];
        float* q = (which == 0 ? J.out[0] : which == 1 ? J.out[1] : which == 2 ? J.out[2] : J.out[3]) + off;
        *q = ((J.acc >> which) & 1) ? *q + tot : tot;
    }
}

SlabJob empty_job() {
    SlabJob j;
    j.slab = nullptr; j.nslab = 0; j.nblocks = 0; j.acc = 0; j.elems = 0; j.lo = 4;
    for (int i = 0; i < 4; ++i) j.out[i] = nullptr;
    j.n[0] = j.n[1] = j.n[2] = 0;
    return j;
}

// slices for a slab of nslab rows: 64 (16 outputs per block) from 96 rows on, 16 (64 outputs) from 16 rows on, else 4 (256)
void shape_job(SlabJob* j) {
    j->lo = j->nslab >= 1024 ? 4 : (j->nslab >= 128 ? 6 : 8);
    const int64_t outs = (int64_t)1 << j->lo;
    j->nblocks = (int)((j->elems + outs - 1) / outs);
}

int public_job(const kpgnn_reduce_job* r, SlabJob* j) {
    *j = empty_job();
    if (!r || !r->slab || r->elems <= 0) return KPGNN_OK;
    int64_t tot = 0;
    for (int i = 0; i < 4; ++i) {
        KPGNN_REQUIRE(r->n_out[i] >= 0 && (r->n_out[i] == 0 || r->out[i]), "reduce job: output %d is NULL", i);
        tot += r->n_out[i];
    }
    KPGNN_REQUIRE(tot == r->elems && r->nslab >= 1, "reduce job: the outputs cover %lld of %lld elements (nslab %d)",
                  (long long)tot, (long long)r->elems, r->nslab);
    j->slab = r->slab; j->nslab = r->nslab; j->elems = r->elems;
    for (int i = 0; i < 4; ++i) j->out[i] = r->out[i];
    for (int i = 0; i < 3; ++i) j->n[i] = r->n_out[i];
    shape_job(j);
    return KPGNN_OK;
}

int launch_slab(const SlabArgs& a, int nbt, hipStream_t s) {
    const int64_t blocks = (int64_t)a.j[0].nblocks + a.j[1].nblocks + a.j[2].nblocks + nbt;
    if (blocks == 0) return KPGNN_OK;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)blocks), dim3(1024), 0, s, a);
    KPGNN_LAUNCH_CHECK("slab_reduce_kernel");
    return KPGNN_OK;
}

}  // namespace

int slab_reduce(const float* slab, int nslab, int64_t elems, float* out0, int64_t n0, float* out1, int64_t n1,
                float* out2, hipStream_t s, int64_t n2, float* out3, const float* slab_b, int nslab_b, int64_t elems_b,
                float* out_b, const ThetaFinish* tf, int acc_mask, const kpgnn_reduce_job* pending) {
    if (!slab_b || elems_b <= 0) { slab_b = nullptr; elems_b = 0; }
    if (elems < 0) elems = 0;
    SlabArgs a;
    a.tf.slab = nullptr; a.tf.nslab = 0; a.tf.alpha = a.tf.theta = nullptr; a.tf.K = 1; a.tf.D = 0; a.tf.gtheta = a.tf.galpha = nullptr;
    int nbt = 0;
    if (tf && tf->slab && tf->D > 0) {
        if (tf->K > 64) return fail(KPGNN_ELIMIT, "slab_reduce: theta finishing needs K <= 64 (K = %d)", tf->K);
        a.tf = *tf;
        const int CB = 64 / a.tf.K;
        nbt = (a.tf.D + CB - 1) / CB;
    }
    if (!out3) n2 = elems;   // three outputs: the rest goes to out2
    a.j[0] = empty_job();
    if (elems > 0) {
        SlabJob& j = a.j[0];
        j.slab = slab; j.nslab = nslab; j.elems = elems;
        shape_job(&j);
        j.out[0] = out0; j.out[1] = out1; j.out[2] = out2; j.out[3] = out3;
        j.n[0] = n0; j.n[1] = n1; j.n[2] = n2;
        j.acc = (acc_mask & 1) ? 4 : 0;           // bit 0 of acc_mask: the third output accumulates
    }
    a.j[1] = empty_job();
    if (elems_b > 0) {
        SlabJob& j = a.j[1];
        j.slab = slab_b; j.nslab = nslab_b; j.elems = elems_b;
        shape_job(&j);
        j.out[0] = out_b; j.n[0] = elems_b;
        j.acc = (acc_mask & 2) ? 1 : 0;
    }
    const int rc = public_job(pending, &a.j[2]);
    if (rc != KPGNN_OK) return rc;
    return launch_slab(a, nbt, s);
}

extern "C" int kpgnn_reduce_jobs(const kpgnn_reduce_job* jobs, int32_t count, kpgnn_stream_t stream) {
    KPGNN_REQUIRE(count >= 0 && (count == 0 || jobs), "reduce_jobs: bad arguments");
    for (int c0 = 0; c0 < count; c0 += 3) {
        SlabArgs a;
        a.tf.slab = nullptr; a.tf.nslab = 0; a.tf.alpha = a.tf.theta = nullptr; a.tf.K = 1; a.tf.D = 0; a.tf.gtheta = a.tf.galpha = nullptr;
        for (int i = 0; i < 3; ++i) {
            const int rc = public_job(c0 + i < count ? jobs + c0 + i : nullptr, &a.j[i]);
            if (rc != KPGNN_OK) return rc;
        }
        const int rc = launch_slab(a, 0, (hipStream_t)stream);
        if (rc != KPGNN_OK) return rc;
    }
    return KPGNN_OK;
}

namespace {

struct Plan { int grid_x, grid_y, cpl, AS; size_t lds, ws_bytes; int R; };

int make_plan(int N, int K, int D, int NT, int n0, int nk, int U, Plan* pl, int extra_rows = 0) {
    if (NT * K > kMaxRows || K > 8 || NT > 8)
        return fail(KPGNN_ELIMIT, "table_grad: nodes_per_tile=%d x K=%d exceeds the %d-row (8x8) register tile", NT, K, kMaxRows);
    pl->R = n0 + nk + U;
    if (pl->R + kSlotRows > 4095) return fail(KPGNN_ELIMIT, "table_grad: %d table rows exceed the 12-bit row id", pl->R);
    pl->cpl = (D % 2 == 0) ? 2 : 1;
    const int cols = kWave * pl->cpl;
    pl->grid_y = (D + cols - 1) / cols;
    pl->AS = pl->grid_y == 1 ? D : cols;
    pl->lds = sizeof(float) * ((((size_t)NT * K * D + 3) & ~(size_t)3) + (size_t)pl->AS * (pl->R + kSlotRows + 16 + extra_rows)) + 2 * kWavesTG * 4;
    if (pl->lds > 160 * 1024)
        return fail(KPGNN_ELIMIT, "table_grad: %zu B of LDS needed (tile %dx%d rows + %d table rows)", pl->lds, NT, K, pl->R);
    const int64_t num_tiles = ((int64_t)N + NT - 1) / NT;
    int per_cu = (int)((160 * 1024) / pl->lds);
    per_cu = per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu);  // 512-thread blocks
    int64_t gx = (int64_t)device_facts().cu_count * per_cu;
    if (gx > num_tiles) gx = num_tiles;
    if (gx < 1) gx = 1;
    pl->grid_x = (int)gx;
    pl->ws_bytes = sizeof(float) * (size_t)gx * pl->R * D;
    return KPGNN_OK;
}

template <int CPL, bool VEC4, bool BF = false, bool FUSE = false, bool WPH1 = false>
int launch_walk(const TgParams& p, const Plan& pl, hipStream_t s) {
    if (pl.lds > 64 * 1024) KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)table_grad_kernel<CPL, VEC4, BF, FUSE, WPH1>, pl.lds));
    hipLaunchKernelGGL((table_grad_kernel<CPL, VEC4, BF, FUSE, WPH1>), dim3(pl.grid_x, pl.grid_y), dim3(kThreadsTG), pl.lds, s, p, pl.AS);
    KPGNN_LAUNCH_CHECK("table_grad_kernel");
    return KPGNN_OK;
}

}  // namespace
}  // namespace kpgnn

using namespace kpgnn;

extern "C" size_t kpgnn_table_grad_workspace_bytes(int32_t N, int32_t K, int32_t D, int32_t nodes_per_tile,
                                                   int32_t n_code0, int32_t n_codek, int32_t n_dict) {
    Plan pl;
    if (N <= 0 || K < 1 || D < 1 || nodes_per_tile < 1) return 0;
    const size_t mfma = table_grad_mfma_ws_bytes(N, K, D, nodes_per_tile, n_code0, K > 1 ? n_codek : 0, n_dict);
    if (make_plan(N, K, D, nodes_per_tile, n_code0, K > 1 ? n_codek : 0, n_dict, &pl) != KPGNN_OK) return mfma;
    return pl.ws_bytes > mfma ? pl.ws_bytes : mfma;  // 0 means "neither kernel fits": the caller takes its atomic fallback
}

extern "C" size_t kpgnn_table_grad_fuse_workspace_bytes(int32_t K, int32_t D) {
    // per block up to 8 / pow2ceil(K) partial [K,D] tables (several waves share a hop when K <= 4): K * that <= 8 rows
    return K >= 1 && D >= 1 ? sizeof(float) * (size_t)device_facts().cu_count * 2 * 8 * D : 0;
}

extern "C" int kpgnn_dict_tile_pack(const int32_t* uid, int64_t uid_stride, int32_t N, int32_t K, int32_t nodes_per_tile,
                                    uint32_t* pack, kpgnn_stream_t stream) {
    KPGNN_REQUIRE(N >= 0 && K >= 1 && K <= 8 && nodes_per_tile >= 1 && nodes_per_tile <= 8 && nodes_per_tile * K <= kMaxRows,
                  "dict_tile_pack: bad N=%d K=%d nodes_per_tile=%d (tiles of at most 8 nodes x 8 hops)", N, K, nodes_per_tile);
    if (N == 0) return KPGNN_OK;
    KPGNN_REQUIRE(uid && pack && uid_stride >= K, "dict_tile_pack: NULL uid/pack or uid_stride < K");
    const int64_t tiles = ((int64_t)N + nodes_per_tile - 1) / nodes_per_tile;
    hipLaunchKernelGGL(dict_tile_pack_kernel, dim3((unsigned)tiles), dim3(kWave), 0, (hipStream_t)stream, uid, uid_stride,
                       N, K, nodes_per_tile, pack);
    KPGNN_LAUNCH_CHECK("dict_tile_pack_kernel");
    return KPGNN_OK;
}

extern "C" int kpgnn_table_grad(const kpgnn_table_grad_desc* d, kpgnn_stream_t stream) {
    KPGNN_REQUIRE(d != nullptr, "table_grad: NULL descriptor");
    KPGNN_REQUIRE(d->N >= 0 && d->K >= 1 && d->K <= 4096 && d->D >= 1 && d->nodes_per_tile >= 1 && d->nodes_per_tile <= 8,
                  "table_grad: bad N=%d K=%d D=%d nodes_per_tile=%d", d->N, d->K, d->D, d->nodes_per_tile);
    if (d->N == 0) return KPGNN_OK;
    KPGNN_REQUIRE(d->g != nullptr || d->fuse_pre != nullptr, "table_grad: NULL g");
    const bool edges = d->tile_ptr != nullptr;
    KPGNN_REQUIRE(!edges || (d->gtable0 && d->n_code0 >= 1 && (d->K == 1 || (d->gtablek && d->n_codek >= 1))),
                  "table_grad: missing gtable0/gtablek");
    KPGNN_REQUIRE(d->n_dict >= 0 && (d->n_dict == 0 || (d->uid && d->gdict && d->uid_stride >= d->K &&
                  (d->dict_src == 2 || (d->dict_src == 1 && d->theta && d->gh)))),
                  "table_grad: dictionary gradient needs uid/gdict and (theta, gh) or dict_src 2");
    KPGNN_REQUIRE(d->n_dict == 0 || !d->dict_pack || (d->dict_pack_K >= d->K && d->dict_pack_K <= 8),
                  "table_grad: dict_pack was built for %d hops, g has %d", d->dict_pack_K, d->K);
    KPGNN_REQUIRE(edges || d->n_dict > 0, "table_grad: nothing to do");
    KPGNN_REQUIRE(!d->extra_slab || (d->extra_out && d->extra_nslab >= 1 && d->extra_elems >= 1), "table_grad: bad extra slab");
    hipStream_t s = (hipStream_t)stream;
    if (d->fuse_pre) {
        // Combine backward fused in: g = theta[k] * gh[i] * gelu'(S[i,k]) is computed per tile, written to fuse_g and walked
        // from LDS; the theta gradient leaves through fuse_workspace.  Edge-code tables only (the dictionary gradient has its
        // own kernel), fp32, one column block.
        const bool bf = d->storage == KPGNN_STORE_BF16;   // (bf16 S / dL/dS: the matrix-core kernel only)
        KPGNN_REQUIRE(edges && (bf || d->storage == KPGNN_STORE_F32) && d->K <= 8 && d->nodes_per_tile == 8 &&
                      d->D % 2 == 0 && d->D <= 2 * kWave && d->theta && d->gh && d->fuse_g,
                      "table_grad(fused combine): needs the edge lists, K <= 8, tiles of 8 nodes, even D <= 128, theta, gh, fuse_g");
        KPGNN_REQUIRE(d->n_dict == 0 || (d->dict_src == 1 && d->dict_pack && d->gdict && d->dict_pack_K >= d->K && d->dict_pack_K <= 8),
                      "table_grad(fused combine): dictionary rows need dict_src 1, gdict and the uid-sorted list of kpgnn_dict_tile_pack");
        KPGNN_REQUIRE(!d->fuse_uid || (d->fuse_ptab && d->fuse_n_dict >= 1 && d->fuse_uid_stride >= d->K), "table_grad(fused combine): bad dictionary");
        KPGNN_REQUIRE(!d->fuse_gtheta || (d->fuse_workspace && d->fuse_workspace_bytes >= kpgnn_table_grad_fuse_workspace_bytes(d->K, d->D)),
                      "table_grad(fused combine): theta-gradient workspace too small");
        KPGNN_REQUIRE(!d->fuse_galphas || (d->fuse_alphas && d->fuse_gtheta), "table_grad(fused combine): galphas needs alphas and gtheta");
        TgParams p;
        p.N = d->N; p.n_dyn = d->n_dyn; p.K = d->K; p.D = d->D; p.NT = d->nodes_per_tile;
        p.n0 = d->n_code0; p.nk = d->K > 1 ? d->n_codek : 0; p.U = d->n_dict; p.dict_src = d->n_dict > 0 ? 1 : 0; p.KD = d->dict_pack_K;
        p.tptr = d->tile_ptr; p.tpack = d->tile_pack; p.g = nullptr;
        p.uid = d->uid; p.uid_stride = d->uid_stride; p.dpack = d->n_dict > 0 ? d->dict_pack : nullptr; p.theta = d->theta; p.gh = d->gh;
        p.f_pre = d->fuse_pre; p.f_ptab = d->fuse_uid ? d->fuse_ptab : nullptr; p.f_uid = d->fuse_uid; p.f_uid_stride = d->fuse_uid_stride;
        p.f_U = d->fuse_uid ? d->fuse_n_dict : 0; p.f_g = d->fuse_g; p.f_gth = d->fuse_gtheta ? (float*)d->fuse_workspace : nullptr;
        p.f_g_sn = d->g_sn; p.f_g_sk = d->g_sk;
        KPGNN_REQUIRE(p.f_g_sn >= d->D && p.f_g_sn % 2 == 0 && p.f_g_sk % 2 == 0 && (d->K == 1 || p.f_g_sk >= d->D),
                      "table_grad(fused combine): fuse_g strides (g_sn=%lld, g_sk=%lld) must be even and >= D", (long long)p.f_g_sn, (long long)p.f_g_sk);
        Plan pl;
        int rc = make_plan(p.N, p.K, p.D, p.NT, p.n0, p.nk, p.U, &pl, 8 + p.f_U);
        if (rc != KPGNN_OK) return rc;
        KPGNN_REQUIRE(pl.grid_y == 1 && pl.cpl == 2, "table_grad(fused combine): one column block of two columns per lane expected");
        KPGNN_REQUIRE(d->workspace && d->workspace_bytes >= pl.ws_bytes, "table_grad: workspace too small (%zu < %zu)",
                      (size_t)d->workspace_bytes, pl.ws_bytes);
        p.slab = (float*)d->workspace;
        // no dictionary rows in the walk, <= 64 accumulator rows and every (row, code) multiplicity known to be below 64 (no run of
        // the entry list was cut: the 8-bit count cells cannot overflow): the matrix-core kernel
        const bool mfma = p.U == 0 && p.n0 + p.nk <= 64 && d->max_multiplicity >= 1 && d->max_multiplicity < 64 &&
                          tg_fuse_mfma_lds(p.D, p.f_U) <= (size_t)device_facts().lds_per_block && d->kernel != 1;
        if (bf && !mfma)
            return fail(KPGNN_ELIMIT, "table_grad(fused combine): bf16 storage needs the matrix-core kernel (no dictionary rows in the "
                        "walk, <= 64 table rows, max_multiplicity known and < 64)");
        if (mfma) {
            const size_t lds = tg_fuse_mfma_lds(p.D, p.f_U);
            int per_cu = (int)((160 * 1024) / lds);
            per_cu = per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu);
            int kp = 1;
            while (kp < p.K) kp <<= 1;
            const int sub = 8 / kp;                             // tiles of 8 nodes per super-tile of 64 rows
            const int64_t num_super = (((int64_t)p.N + 7) / 8 + sub - 1) / sub;
            int64_t gx = (int64_t)device_facts().cu_count * per_cu;
            if (gx > num_super) gx = num_super;
            if (gx > pl.grid_x) gx = pl.grid_x;                  // (the workspace was sized for the walk's grid)
            pl.grid_x = (int)(gx < 1 ? 1 : gx);
#define KP_TGF2(S, B) do { KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)tg_fuse_mfma_kernel<S, B>, lds)); \
                           hipLaunchKernelGGL((tg_fuse_mfma_kernel<S, B>), dim3(pl.grid_x), dim3(kThreadsTG), lds, s, p); } while (0)
#define KP_TGF(S) do { if (bf) KP_TGF2(S, true); else KP_TGF2(S, false); } while (0)
            if (sub == 1) KP_TGF(1); else if (sub == 2) KP_TGF(2); else if (sub == 4) KP_TGF(4); else KP_TGF(8);
#undef KP_TGF2
#undef KP_TGF
            KPGNN_LAUNCH_CHECK("tg_fuse_mfma_kernel");
            rc = KPGNN_OK;
        } else {
            rc = p.K >= 5 ? launch_walk<2, false, false, true, true>(p, pl, s) : launch_walk<2, false, false, true, false>(p, pl, s);
        }
        if (rc != KPGNN_OK) return rc;
        ThetaFinish tf;
        tf.slab = nullptr; tf.nslab = 0; tf.alpha = d->fuse_alphas; tf.theta = d->theta; tf.K = d->K; tf.D = d->D;
        tf.gtheta = d->fuse_gtheta; tf.galpha = d->fuse_galphas;
        if (d->fuse_gtheta) {
            int kp = 1;
            while (kp < d->K) kp <<= 1;
            tf.slab = p.f_gth; tf.nslab = pl.grid_x * (kWavesTG / kp);
        }
        // ONE finishing launch: table slabs, the dictionary-gradient slab a kpgnn_dict_grad left behind, the theta gradient
        return slab_reduce(p.slab, pl.grid_x, (int64_t)pl.R * p.D, d->gtable0, (int64_t)p.n0 * p.D, d->gtablek,
                           (int64_t)p.nk * p.D, d->gdict, s, 0, nullptr, d->extra_slab, d->extra_nslab, d->extra_elems, d->extra_out,
                           d->fuse_gtheta ? &tf : nullptr, d->accumulate_dict ? 3 : 0, d->pending);
    }
    KPGNN_REQUIRE(d->g_sk == d->D && d->g_sn == (int64_t)d->K * d->D, "table_grad: g must be contiguous [N,K,D]");
    {   // Narrow rows (D <= 32: KP-GIN's dk = hidden / K) and shapes the walk kernel cannot tile (K > 8) go to the
        // count-matrix product on the matrix cores: measured 56 vs 68 us (edge codes) and 70 vs 299 us (with unsorted
        // dictionary rows) at D = 13.  Wide rows stay on the register walk (the 16x16x4 product is matrix-core bound
        // there).  d->kernel = 1 / 2 forces one of them (the parity tests compare the two).
        KPGNN_REQUIRE(d->storage == KPGNN_STORE_F32 || d->storage == KPGNN_STORE_BF16, "table_grad: unknown storage %d", d->storage);
        const int force = d->storage == KPGNN_STORE_BF16 ? 1 : d->kernel;    // (bf16 rows: the walk kernel only)
        // (the walk adds finished runs with plain read-modify-writes: it needs BOTH lists sorted by row)
        const bool walk_fits = d->K <= 8 && d->nodes_per_tile * d->K <= kMaxRows && (d->n_dict == 0 || d->dict_pack);
        KPGNN_REQUIRE(force != 1 || walk_fits, "table_grad: the walk kernel needs K <= 8, tiles of <= 64 (node, hop) rows and, "
                      "with a dictionary, the uid-sorted list of kpgnn_dict_tile_pack");
        if (force != 1 && (force == 2 || d->D <= 32 || !walk_fits)) {
            bool handled = false;
            const int rc = table_grad_mfma(d, s, &handled);
            if (rc != KPGNN_OK) return rc;
            if (handled) return KPGNN_OK;   // (its finishing launch took the deferred dictionary slab and the pending job along)
        }
    }
    TgParams p;
    p.N = d->N; p.n_dyn = d->n_dyn; p.K = d->K; p.D = d->D; p.NT = d->nodes_per_tile;
    p.n0 = edges ? d->n_code0 : 0; p.nk = (edges && d->K > 1) ? d->n_codek : 0;
    p.U = d->n_dict; p.dict_src = d->dict_src;
    p.tptr = d->tile_ptr; p.tpack = d->tile_pack; p.g = d->g;
    p.uid = d->uid; p.uid_stride = d->uid_stride; p.theta = d->theta; p.gh = d->gh;
    // the sorted dictionary list addresses hops with the stride it was built for; it only applies when that is g's K
    // (a list built for all K hops serves every layer: entries of hops >= d->K are skipped)
    p.dpack = d->n_dict > 0 ? d->dict_pack : nullptr; p.KD = d->dict_pack_K;
    p.f_pre = nullptr; p.f_ptab = nullptr; p.f_uid = nullptr; p.f_uid_stride = 0; p.f_U = 0; p.f_g = nullptr; p.f_gth = nullptr;
    p.f_g_sn = p.f_g_sk = 0;
    Plan pl;
    int rc = make_plan(p.N, p.K, p.D, p.NT, p.n0, p.nk, p.U, &pl);
    if (rc != KPGNN_OK) return rc;
    KPGNN_REQUIRE(d->workspace && d->workspace_bytes >= pl.ws_bytes, "table_grad: workspace too small (%zu < %zu)",
                  (size_t)d->workspace_bytes, pl.ws_bytes);
    p.slab = (float*)d->workspace;
    // the tile copy is flat: 16-B loads only need every node's K*D floats to be a multiple of 4
    const bool bf = d->storage == KPGNN_STORE_BF16;
    const bool vec4 = (((int64_t)p.K * p.D) % (bf ? 8 : 4) == 0) && (((uintptr_t)p.g & 15) == 0);
    if (bf) {
        KPGNN_REQUIRE(pl.cpl == 2, "table_grad: bf16 storage needs an even D");
        rc = vec4 ? launch_walk<2, true, true>(p, pl, s) : launch_walk<2, false, true>(p, pl, s);
    } else if (pl.cpl == 2) rc = vec4 ? launch_walk<2, true>(p, pl, s) : launch_walk<2, false>(p, pl, s);
    else rc = vec4 ? launch_walk<1, true>(p, pl, s) : launch_walk<1, false>(p, pl, s);
    if (rc != KPGNN_OK) return rc;
    return slab_reduce(p.slab, pl.grid_x, (int64_t)pl.R * p.D, d->gtable0, (int64_t)p.n0 * p.D, d->gtablek,
                       (int64_t)p.nk * p.D, d->gdict, s, 0, nullptr, d->extra_slab, d->extra_nslab, d->extra_elems, d->extra_out,
                       nullptr, d->accumulate_dict ? 3 : 0, d->pending);
}
