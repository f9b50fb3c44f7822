// Table gradients without per-edge atomics (gfx950).  Contract: include/kpgnn.h, kpgnn_table_grad.
//
//   gtable_t[c,:]  = sum over active pairs (i,k) of table t with code c of g[i,k,:]          (edge-code tables)
//   gdict[u,:]     = sum over (i,k) with uid[i,k] == u of theta[k,:] * gh[i,:]               (peripheral dictionary)
//
// Why a separate kernel: LDS float atomics (ds_add_f32) per gathered edge row made the backward gather 5x
// slower than the gather itself (622 us vs 120 us at N=47k, K=8, D=104: a handful of hot codes serialise),
// and flushing per-block tables with global atomics made every block hammer the same few KB.
// Here the accumulation is column-private: thread t owns feature column t of every table row, the tile's
// pair list arrives sorted by (table, code), so a run of equal codes is summed in ONE register and written
// to the thread's own LDS slot when the code changes.  g is streamed once, coalesced (a row = D floats),
// the next tile's loads are in flight while the current tile is walked, and the pair list is wave-uniform
// (scalar loads).  Per-block partial tables go to a workspace slab with plain stores; a second launch adds
// the slabs in block order (deterministic; only the few per-tile LDS flushes of the walker groups race).
#include "kpgnn_common.h"

namespace kpgnn {
namespace {

constexpr int kCols = 128;    // feature columns per block (one walker group = kCols threads = 2 waves)
constexpr int kGroups = 4;    // walker groups per block: each walks a quarter of the tile's pair list
constexpr int kThreadsTG = kCols * kGroups;
constexpr int kMaxRows = 64;  // NT*K: at most 8 nodes x 8 hops per tile

struct TgParams {
    int N, K, D, NT, n0, nk, U, dict_src;
    const int32_t* tptr;
    const uint32_t* tpack;
    const float* g;
    // peripheral dictionary (optional): gdict[uid[i*uid_stride + k]] += theta[k,:] * gh[i,:]
    const int32_t* uid; int64_t uid_stride;
    const float* theta;
    const float* gh;
    float* slab;          // [gridDim.x][n0 + nk + U][D]
};

// g must be contiguous [N,K,D]: a tile of NT nodes is then one contiguous run of NT*K*D floats, copied to
// LDS as is (16-B loads when D % 4 == 0); thread t reads column d of row r at tile[r*D + d] (bank = d).
// The kGroups walker groups share the tile and the accumulator table; a group keeps the running sum of ITS
// current code in a register and flushes it with an LDS atomic (a code can straddle two groups' chunks; the
// flushes are a handful per tile, so the atomics cost nothing - unlike one atomic per edge).
template <bool VEC4>
__global__ void __launch_bounds__(kThreadsTG)
table_grad_kernel(const TgParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x / kCols);   // wave-uniform: keeps the walk in SALU control flow
    const int t = threadIdx.x % kCols;
    const int lane = t & 63;
    const int d = blockIdx.y * kCols + t;
    const bool col_ok = d < p.D;
    const int dc = col_ok ? d : 0;                   // clamped column for address arithmetic
    const int D = p.D;
    const int R = p.n0 + p.nk + p.U;                 // table rows
    const int tile_floats = p.NT * p.K * D;
    float* tile = lds;                               // [NT*K][D]
    float* acc = lds + ((tile_floats + 3) & ~3);     // [R][kCols], column-private
    float* ghs = acc + R * kCols;                    // [8][kCols]
    for (int r = grp; r < R; r += kGroups) acc[r * kCols + t] = 0.f;
    const int64_t num_tiles = ((int64_t)p.N + p.NT - 1) / p.NT;
    const int64_t total = (int64_t)p.N * p.K * D;
    int cur = -1;       // current accumulator row (table offset + code), -1 = none
    float run = 0.f;
    int ucur = -1;      // current dictionary row
    float urun = 0.f;
    float th[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) th[k] = (p.dict_src == 1 && k < p.K && col_ok) ? p.theta[k * D + d] : 0.f;
    constexpr int kPref = 4;
    float4 pref[kPref];
    if (VEC4) {
#pragma unroll
        for (int q = 0; q < kPref; ++q) {
            const int64_t i = (int64_t)blockIdx.x * tile_floats + (q * kThreadsTG + threadIdx.x) * 4;
            pref[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (blockIdx.x < num_tiles && i < total && (q * kThreadsTG + threadIdx.x) * 4 < tile_floats)
                pref[q] = *reinterpret_cast<const float4*>(p.g + i);
        }
    }
    // Per-tile metadata travels two tiles ahead in registers: the entry-list window (tile_ptr) of tile i+2 and, from the
    // window that arrived an iteration ago, the first 64 entries / dictionary ids / gh values of tile i+1, so that no
    // dependent global round trip (tile_ptr -> tile_pack) sits in front of a walk.  (Worth ~3 us per launch only: the
    // walk itself, not its operands' latency, is the critical path.)
    // (Measured alternatives, k = 8, D = 104: hop-major entry order 90 us, chunks cut at table-row boundaries with plain
    //  read-modify-write flushes 75 us, this version 62 us - DESIGN.md section 5.)
    struct TileMeta { int beg, end; uint32_t nxt; int uid; float gh[2]; };
    static_assert(kGroups * 2 >= 8, "two gh values per thread cover a tile of up to 8 nodes");
    // window of tile t2: lane 0 fetches its begin, lane 1 its end (kept as a per-lane value on purpose: a wave-uniform
    // load is waited for where it is issued, to move it to scalar registers)
    auto load_range = [&](int64_t t2) -> int {
        int v = 0;
        if (p.tptr && t2 < num_tiles && lane < 2) v = p.tptr[t2 + lane];
        return v;
    };
    auto load_meta = [&](int64_t t2, int rb, int re, TileMeta& m) {
        // this group's contiguous chunk of the sorted entry list (multiple of 8 entries except the tail)
        const int per = ((re - rb + kGroups - 1) / kGroups + 7) & ~7;
        m.beg = min(re, rb + grp * per);
        m.end = min(re, m.beg + per);
        m.nxt = (m.beg + lane < m.end) ? p.tpack[m.beg + lane] : 0xFFFFFFFFu;   // hop 63 == skip
        m.uid = -1;                                  // lane l: dictionary row of tile row l (NT*K <= 64)
        if (p.U > 0 && lane < p.NT * p.K && t2 < num_tiles) {
            const int n = lane / p.K;
            const int64_t node = t2 * p.NT + n;
            if (node < p.N) m.uid = p.uid[node * p.uid_stride + (lane - n * p.K)];
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {                // gh rows of the tile's nodes, column t
            const int n = grp + q * kGroups;
            const int64_t node = t2 * p.NT + n;
            m.gh[q] = (p.dict_src == 1 && n < p.NT && t2 < num_tiles && node < p.N && col_ok) ? p.gh[node * D + d] : 0.f;
        }
    };
    TileMeta cm;
    int rng1;
    {
        const int rng0 = load_range(blockIdx.x);
        load_meta(blockIdx.x, __builtin_amdgcn_readlane(rng0, 0), __builtin_amdgcn_readlane(rng0, 1), cm);
        rng1 = load_range((int64_t)blockIdx.x + gridDim.x);
    }
    for (int64_t tl = blockIdx.x; tl < num_tiles; tl += gridDim.x) {
        // Consume this tile's metadata NOW (requested an iteration ago): the wait then sits in front of the loads issued
        // below for the next tile; a wait placed inside the walk would be a vmcnt(0) that also waits for the g rows just
        // requested for the next tile.
        uint32_t mine = cm.nxt;
        int myuid = cm.uid;                           // lane l: dictionary row of tile row l
        float gh0 = cm.gh[0], gh1 = cm.gh[1];
        asm volatile("" : "+v"(mine), "+v"(myuid), "+v"(gh0), "+v"(gh1));
        TileMeta nm;                                  // next tile (its window arrived with the loads above) and the one after
        load_meta(tl + gridDim.x, __builtin_amdgcn_readlane(rng1, 0), __builtin_amdgcn_readlane(rng1, 1), nm);
        rng1 = load_range(tl + 2 * (int64_t)gridDim.x);
        int beg = cm.beg, end = cm.end;
        const int64_t base = tl * tile_floats;
        const int nfl = (int)min((int64_t)tile_floats, total - base);
        __syncthreads();                             // previous tile fully walked
        if (VEC4) {
            // the first kPref*2048 floats of the tile were prefetched into registers during the previous walk
#pragma unroll
            for (int q = 0; q < kPref; ++q) {
                const int i = (q * kThreadsTG + threadIdx.x) * 4;
                if (i < nfl) *reinterpret_cast<float4*>(tile + i) = pref[q];
            }
            for (int i = (kPref * kThreadsTG + threadIdx.x) * 4; i < nfl; i += kThreadsTG * 4)
                *reinterpret_cast<float4*>(tile + i) = *reinterpret_cast<const float4*>(p.g + base + i);
            const int64_t nb = base + (int64_t)gridDim.x * tile_floats;
#pragma unroll
            for (int q = 0; q < kPref; ++q) {
                const int64_t i = nb + (q * kThreadsTG + threadIdx.x) * 4;
                if (tl + gridDim.x < num_tiles && i < total && (q * kThreadsTG + threadIdx.x) * 4 < tile_floats)
                    pref[q] = *reinterpret_cast<const float4*>(p.g + i);
            }
        } else {
            for (int i = threadIdx.x; i < nfl; i += kThreadsTG) tile[i] = p.g[base + i];
        }
        if (p.dict_src == 1) {
            if (grp < p.NT) ghs[grp * kCols + t] = gh0;
            if (grp + kGroups < p.NT) ghs[(grp + kGroups) * kCols + t] = gh1;
        }
        __syncthreads();
        // ---- walk the (table,code)-sorted pair list of this tile.  Each wave fetches 64 entries with ONE
        //      coalesced load (lane l holds entry l) and broadcasts them with v_readlane (SGPR, no LDS, no
        //      scalar-cache misses); the LDS tile reads of 8 entries are issued back to back before the
        //      (sequential, register-only) run accumulation.  The next chunk is fetched while this one is walked.
        // every lane decodes ITS entry once (VALU, 64 entries per instruction); the per-entry scalar work is then three
        // v_readlane, one compare and the fma
        int vmul, voff, vrow;
        auto decode = [&](uint32_t w) {
            const int vhop = w & 0x3F;
            const bool vok = vhop < p.K;
            vmul = __float_as_int((float)(((w >> 6) & 0x3F) + 1));        // multiplicity of the merged entry
            voff = vok ? ((int)((w >> 12) & 7) * p.K + vhop) * D : 0;
            const int vcc = (int)(w >> 15);                                // table<<16 | code
            vrow = vok ? ((vcc >> 16) ? p.n0 + (vcc & 0xFFFF) : vcc) : -1;
        };
        decode(mine);
        for (int b0 = beg; b0 < end; b0 += 64) {
            const int cnt = min(64, end - b0);
            for (int e0 = 0; e0 < cnt; e0 += 8) {
                int off[8], row[8]; float val[8], mul[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    off[u] = __builtin_amdgcn_readlane(voff, e0 + u);
                    row[u] = __builtin_amdgcn_readlane(vrow, e0 + u);
                    mul[u] = __int_as_float(__builtin_amdgcn_readlane(vmul, e0 + u));
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) val[u] = tile[off[u] + dc];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (row[u] >= 0) {
                        if (row[u] != cur) {                               // wave-uniform
                            if (cur >= 0) atomicAdd(&acc[cur * kCols + t], run);
                            cur = row[u];
                            run = 0.f;
                        }
                        run = fmaf(mul[u], val[u], run);
                    }
                }
            }
            if (b0 + 64 < end)                       // (rare: > 64 entries in a group's chunk; fetched and decoded here so
                decode((b0 + 64 + lane < end) ? p.tpack[b0 + 64 + lane] : 0xFFFFFFFFu);   //  that the loop carries no pending load)
        }
        // ---- peripheral dictionary: rows in natural order, equal uids (the common case) stay in a register;
        //      theta / gh sit in registers, the (wave-uniform) uids of a node are fetched together
        if (p.U > 0) {
            const int64_t node0 = tl * p.NT;
            for (int n = grp; n < p.NT && node0 + n < p.N; n += kGroups) {
                const float ghv = ghs[n * kCols + t];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (k >= p.K) break;
                    const int u = __builtin_amdgcn_readlane(myuid, n * p.K + k);   // wave-uniform
                    if (u != ucur) {
                        if (ucur >= 0) atomicAdd(&acc[(p.n0 + p.nk + ucur) * kCols + t], urun);
                        ucur = u;
                        urun = 0.f;
                    }
                    if (p.dict_src == 1) urun = fmaf(th[k], ghv, urun);
                    else urun += tile[(n * p.K + k) * D + dc];
                }
            }
        }
        cm = nm;
    }
    if (cur >= 0) atomicAdd(&acc[cur * kCols + t], run);
    if (ucur >= 0) atomicAdd(&acc[(p.n0 + p.nk + ucur) * kCols + t], urun);
    __syncthreads();
    if (col_ok) {
        float* out = p.slab + (int64_t)blockIdx.x * R * D + d;
        for (int r = grp; r < R; r += kGroups) out[(int64_t)r * D] = acc[r * kCols + t];
    }
}

// out[e] = sum_b slab[b][e]   (fixed order: deterministic).  16 outputs x 64 slices of the slab range per WG: each
// thread adds nslab/64 values (independent loads), the 64 partials of an output meet in LDS.
__global__ void __launch_bounds__(1024)
slab_reduce_kernel(const float* __restrict__ slab, int nslab, int64_t elems, float* __restrict__ out0, int64_t n_out0,
                   float* __restrict__ out1, int64_t n_out1, float* __restrict__ out2, int64_t n_out2,
                   float* __restrict__ out3) {
    __shared__ float part[64][17];
    const int o = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int64_t e = (int64_t)blockIdx.x * 16 + o;
    float s = 0.f;
    if (e < elems) {
        int b = slice;
        for (; b + 192 < nslab; b += 256) {
            const float v0 = slab[(int64_t)b * elems + e], v1 = slab[(int64_t)(b + 64) * elems + e];
            const float v2 = slab[(int64_t)(b + 128) * elems + e], v3 = slab[(int64_t)(b + 192) * elems + e];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; b < nslab; b += 64) s += slab[(int64_t)b * elems + e];
    }
    part[slice][o] = s;
    __syncthreads();
    if (slice == 0 && e < elems) {
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < 64; ++q) tot += part[q][o];
        if (e < n_out0) out0[e] = tot;
        else if (e < n_out0 + n_out1) out1[e - n_out0] = tot;
        else if (e < n_out0 + n_out1 + n_out2) out2[e - n_out0 - n_out1] = tot;
        else out3[e - n_out0 - n_out1 - n_out2] = tot;
    }
}

}  // namespace

int slab_reduce(const float* slab, int nslab, int64_t elems, float* out0, int64_t n0, float* out1, int64_t n1,
                float* out2, hipStream_t s, int64_t n2, float* out3) {
    if (elems <= 0) return KPGNN_OK;
    if (!out3) n2 = elems;   // three outputs: the rest goes to out2
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((elems + 15) / 16)), dim3(1024), 0, s, slab, nslab, elems,
                       out0, n0, out1, n1, out2, n2, out3);
    KPGNN_LAUNCH_CHECK("slab_reduce_kernel");
    return KPGNN_OK;
}

namespace {

struct Plan { int grid_x, grid_y; size_t lds, ws_bytes; int R; };

int make_plan(int N, int K, int D, int NT, int n0, int nk, int U, Plan* pl) {
    if (NT * K > kMaxRows || K > 8 || NT > 8)
        return fail(KPGNN_ELIMIT, "table_grad: nodes_per_tile=%d x K=%d exceeds the %d-row (8x8) register tile", NT, K, kMaxRows);
    pl->R = n0 + nk + U;
    pl->lds = sizeof(float) * ((((size_t)NT * K * D + 3) & ~(size_t)3) + (size_t)kCols * (pl->R + 8));
    if (pl->lds > 160 * 1024)
        return fail(KPGNN_ELIMIT, "table_grad: %zu B of LDS needed (tile %dx%d rows + %d table rows)", pl->lds, NT, K, pl->R);
    const int64_t num_tiles = ((int64_t)N + NT - 1) / NT;
    int per_cu = (int)((160 * 1024) / pl->lds);
    per_cu = per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu);  // 512-thread blocks
    int64_t gx = (int64_t)device_facts().cu_count * per_cu;
    if (gx > num_tiles) gx = num_tiles;
    if (gx < 1) gx = 1;
    pl->grid_x = (int)gx;
    pl->grid_y = (D + kCols - 1) / kCols;
    pl->ws_bytes = sizeof(float) * (size_t)gx * pl->R * D;
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

extern "C" int kpgnn_table_grad(const kpgnn_table_grad_desc* d, kpgnn_stream_t stream) {
    KPGNN_REQUIRE(d != nullptr, "table_grad: NULL descriptor");
    KPGNN_REQUIRE(d->N >= 0 && d->K >= 1 && d->K <= 4096 && d->D >= 1 && d->nodes_per_tile >= 1 && d->nodes_per_tile <= 8,
                  "table_grad: bad N=%d K=%d D=%d nodes_per_tile=%d", d->N, d->K, d->D, d->nodes_per_tile);
    if (d->N == 0) return KPGNN_OK;
    KPGNN_REQUIRE(d->g != nullptr, "table_grad: NULL g");
    const bool edges = d->tile_ptr != nullptr;
    KPGNN_REQUIRE(!edges || (d->gtable0 && d->n_code0 >= 1 && (d->K == 1 || (d->gtablek && d->n_codek >= 1))),
                  "table_grad: missing gtable0/gtablek");
    KPGNN_REQUIRE(d->n_dict >= 0 && (d->n_dict == 0 || (d->uid && d->gdict && d->uid_stride >= d->K &&
                  (d->dict_src == 2 || (d->dict_src == 1 && d->theta && d->gh)))),
                  "table_grad: dictionary gradient needs uid/gdict and (theta, gh) or dict_src 2");
    KPGNN_REQUIRE(edges || d->n_dict > 0, "table_grad: nothing to do");
    hipStream_t s = (hipStream_t)stream;
    KPGNN_REQUIRE(d->g_sk == d->D && d->g_sn == (int64_t)d->K * d->D, "table_grad: g must be contiguous [N,K,D]");
    {   // Narrow rows (D <= 32: KP-GIN's dk = hidden / K) and shapes the walk kernel cannot tile (K > 8) go to the
        // count-matrix product on the matrix cores: measured 56 vs 68 us (edge codes) and 70 vs 299 us (with unsorted
        // dictionary rows) at D = 13.  Wide rows stay on the register walk (D = 104: 78 vs 121 us; the 16x16x4 product
        // is matrix-core bound there).  d->kernel = 1 / 2 forces one of them (the parity tests compare the two).
        const int force = d->kernel;
        const bool walk_fits = d->K <= 8 && d->nodes_per_tile * d->K <= kMaxRows;
        if (force != 1 && (force == 2 || d->D <= 32 || !walk_fits)) {
            bool handled = false;
            const int rc = table_grad_mfma(d, s, &handled);
            if (rc != KPGNN_OK || handled) return rc;
        }
    }
    TgParams p;
    p.N = d->N; p.K = d->K; p.D = d->D; p.NT = d->nodes_per_tile;
    p.n0 = edges ? d->n_code0 : 0; p.nk = (edges && d->K > 1) ? d->n_codek : 0;
    p.U = d->n_dict; p.dict_src = d->dict_src;
    p.tptr = d->tile_ptr; p.tpack = d->tile_pack; p.g = d->g;
    p.uid = d->uid; p.uid_stride = d->uid_stride; p.theta = d->theta; p.gh = d->gh;
    Plan pl;
    int rc = make_plan(p.N, p.K, p.D, p.NT, p.n0, p.nk, p.U, &pl);
    if (rc != KPGNN_OK) return rc;
    KPGNN_REQUIRE(d->workspace && d->workspace_bytes >= pl.ws_bytes, "table_grad: workspace too small (%zu < %zu)",
                  (size_t)d->workspace_bytes, pl.ws_bytes);
    p.slab = (float*)d->workspace;
    // the tile copy is flat: 16-B loads only need every node's K*D floats to be a multiple of 4
    const bool vec4 = (((int64_t)p.K * p.D) % 4 == 0) && (((uintptr_t)p.g & 15) == 0);
    if (pl.lds > 64 * 1024) {
        KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)table_grad_kernel<true>, pl.lds));
        KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)table_grad_kernel<false>, pl.lds));
    }
    if (vec4) hipLaunchKernelGGL((table_grad_kernel<true>), dim3(pl.grid_x, pl.grid_y), dim3(kThreadsTG), pl.lds, s, p);
    else hipLaunchKernelGGL((table_grad_kernel<false>), dim3(pl.grid_x, pl.grid_y), dim3(kThreadsTG), pl.lds, s, p);
    KPGNN_LAUNCH_CHECK("table_grad_kernel");
    return slab_reduce(p.slab, pl.grid_x, (int64_t)pl.R * p.D, d->gtable0, (int64_t)p.n0 * p.D, d->gtablek,
                       (int64_t)p.nk * p.D, d->gdict, s);
}
