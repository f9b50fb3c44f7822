// Tall-skinny fp32 Linears on the bf16 matrix cores through exact three-way bf16 splits (gfx950):
//   mode 0  y = act(sum_l x_l W_l^T + b)      the jumping-knowledge projection over S separate states (kpgnn_linear_group_fwd)
//   mode 1  dx_l = (dy * [mask > 0]) W_l      its input gradient, one [N, I] block per state (kpgnn_linear_fwd, blocked output)
// Contract: include/kpgnn.h, kpgnn_linear_group_fwd / kpgnn_linear_fwd with math = KPGNN_MATH_AUTO and a workspace.
//
// The fp32 matrix instruction runs at 1/16 of the bf16 rate; an fp32 value is exactly h + m + l with three bf16 pieces (8 + 8 + 8
// significant bits, split by truncation), and  a b = ah bh + (ah bm + am bh) + (am bm + ah bl + al bh) + [< 2^-24 |a b|]:
// six exact products on v_mfma_f32_32x32x16_bf16, fp32 accumulation, smallest first (bf3.h; the weight gradient: wgrad.hip).
//
// Two launches.  (1) W is split ONCE into the matrix instruction's B-fragment layout ([state][32-column strip][k step][piece]
// [lane] x 16 bytes, <= 1 MB, L2-resident): every block needs all of it, and splitting a strip per block and state cost the
// multiplying waves a quarter of their time.  (2) A block of 8 waves owns 96 rows (three 32-row tiles): waves 0-3 multiply - a
// 32-column output strip each, B fragments straight from the split copy two k steps ahead, A fragments from LDS, 18 matrix
// instructions per k step - and waves 4-7 stage the NEXT state's three tiles meanwhile (requested two states ahead, split,
// written as three row-major bf16 planes; 240-byte row pitch = conflict-free 16-byte fragment reads, 8-byte stores of 16
// contiguous lanes cover 32 banks).  One barrier per state.  Mode 1 stages its tiles once and writes three output tiles per state.
// Measured (s_memtime, [47450, 104] x 9 states, ~1.55 GHz under this load): a state costs the multiplying wave 4700-5000 cycles
// (126 matrix instructions = 4032), the staging wave NEXT TO IT on the SIMD 6300 for ~300 vector instructions (alone: ~1500),
// mode 1's 48 stores 1350: 86 / 88 us against 138 / 136 for the fp32-instruction kernels, W split included.
#include "bf3.h"
#include "kpgnn_common.h"

namespace kpgnn {
namespace {

constexpr int kL3Tiles = 3, kL3Rows = 32 * kL3Tiles;
// bf16 per staged row: 16 KS are read, [I, pitch) stays zero; 240- and 272-byte pitches = conflict-free 16-byte fragment reads
constexpr int l3_pitch(int ks) { return ks <= 7 ? 120 : 136; }
constexpr int l3_buf(int ks) { return 3 * kL3Rows * l3_pitch(ks); }      // bf16 per buffer (three planes of 96 rows)

struct L3Params {
    int64_t N; const int32_t* n_dyn;
    int O, I, S, relu;
    const float* xs[16]; int64_t xstride;          // mode 0: S inputs; mode 1: xs[0] = dy
    const float* xmask;                            // mode 1: optional ReLU mask of dy (rows xstride apart)
    const uint4* wfrag;                            // split W: [S][4 strips][KS][3 pieces][64 lanes]
    const float* bias;
    float* y; int64_t ystride, yblock;             // mode 0: y[row * ystride + n]; mode 1: y[l * yblock + row * ystride + n]
};

// element (k, n) of state l: w[n * wn + k * wk + l * ws]; one thread per (state, strip, k step, lane)
__global__ void __launch_bounds__(256)
lin3_wsplit_kernel(const float* __restrict__ w, int64_t wn, int64_t wk, int64_t ws, int S, int O, int I, int KS, uint4* __restrict__ frag) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = (int)(t & 63);
    const int64_t f = t >> 6;                       // (l * 4 + strip) * KS + ks
    if (f >= (int64_t)S * 4 * KS) return;
    const int ks = (int)(f % KS), strip = (int)((f / KS) & 3), l = (int)(f / (4 * KS));
    const int n = strip * 32 + (lane & 31), k0 = 16 * ks + 8 * (lane >> 5);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (n < O && k0 + j < I) ? w[(int64_t)n * wn + (int64_t)(k0 + j) * wk + (int64_t)l * ws] : 0.f;
    bf3_u2 h[4], m[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bf3_split2(bf3_f2{v[2 * j], v[2 * j + 1]}, h[j], m[j], lo[j]);
    uint4* q = frag + f * 3 * 64 + lane;
    q[0] = make_uint4(bf3_pack(h[0].x, h[0].y), bf3_pack(h[1].x, h[1].y), bf3_pack(h[2].x, h[2].y), bf3_pack(h[3].x, h[3].y));
    q[64] = make_uint4(bf3_pack(m[0].x, m[0].y), bf3_pack(m[1].x, m[1].y), bf3_pack(m[2].x, m[2].y), bf3_pack(m[3].x, m[3].y));
    q[128] = make_uint4(bf3_pack(lo[0].x, lo[0].y), bf3_pack(lo[1].x, lo[1].y), bf3_pack(lo[2].x, lo[2].y), bf3_pack(lo[3].x, lo[3].y));
}

// the same for up to 64 matrices in one launch (blockIdx.y = matrix): every Linear of a body's MLPs at the start of a step
struct SplitMany { kpgnn_split_job j[64]; };
__global__ void __launch_bounds__(256)
lin3_wsplit_many_kernel(const SplitMany a) {
    const kpgnn_split_job& q = a.j[blockIdx.y];
    const int KS = (q.I + 15) / 16;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = (int)(t & 63);
    const int64_t f = t >> 6;                       // strip * KS + ks
    if (f >= 4 * KS) return;
    const int ks = (int)(f % KS), strip = (int)(f / KS);
    const int n = strip * 32 + (lane & 31), k0 = 16 * ks + 8 * (lane >> 5);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (n < q.O && k0 + j < q.I) ? q.w[(int64_t)n * q.wn + (int64_t)(k0 + j) * q.wk] : 0.f;
    bf3_u2 h[4], m[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bf3_split2(bf3_f2{v[2 * j], v[2 * j + 1]}, h[j], m[j], lo[j]);
    uint4* o = reinterpret_cast<uint4*>(q.frag) + f * 3 * 64 + lane;
    o[0] = make_uint4(bf3_pack(h[0].x, h[0].y), bf3_pack(h[1].x, h[1].y), bf3_pack(h[2].x, h[2].y), bf3_pack(h[3].x, h[3].y));
    o[64] = make_uint4(bf3_pack(m[0].x, m[0].y), bf3_pack(m[1].x, m[1].y), bf3_pack(m[2].x, m[2].y), bf3_pack(m[3].x, m[3].y));
    o[128] = make_uint4(bf3_pack(lo[0].x, lo[0].y), bf3_pack(lo[1].x, lo[1].y), bf3_pack(lo[2].x, lo[2].y), bf3_pack(lo[3].x, lo[3].y));
}

template <int KS, int MODE>
__global__ void __launch_bounds__(512, 1)
lin3_kernel(const L3Params p) {
    extern __shared__ __attribute__((aligned(16))) uint4 l3_lds[];
    constexpr int PK = l3_pitch(KS), BUF = l3_buf(KS), ROWS = kL3Rows;
    constexpr int NB = MODE == 0 ? 2 : 1;            // tile buffers
    constexpr int NCG = KS <= 7 ? 28 : 32;           // upper bound of float4 column groups (I <= 112 / 128)
    constexpr int PF = (ROWS * NCG + 255) / 256;     // float4 requests per staging thread and buffer
    __bf16* pl = reinterpret_cast<__bf16*>(l3_lds);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = lane >> 5, c = lane & 31;
    const int O = p.O, I = p.I, S = p.S;
    const int64_t N = p.n_dyn ? (int64_t)min((int64_t)*p.n_dyn, p.N) : p.N;
    const int64_t rows0 = (int64_t)blockIdx.x * ROWS;
    if (rows0 >= N) return;                           // (uniform: the whole block)
    for (int i = tid; i < NB * BUF / 8; i += 512) l3_lds[i] = make_uint4(0u, 0u, 0u, 0u);      // (the k padding stays zero)
    __syncthreads();
    const int T = MODE == 0 ? S : 1;                  // staged buffers: one per state (mode 0), one in all (mode 1)

    if (wave >= 4) {
        // ---- staging waves: task i of a thread = (row, 4 columns) of the 96-row buffer, column group fastest (coalesced requests)
        const int ptid = tid - 256;
        const int ncg = I >> 2;
        int prow[PF], pcg[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            // (tasks beyond the buffer repeat its last one - the same bytes stored twice - rather than guard the stores: eleven
            //  guarded stores became eleven divergent branches with the split inside them)
            const int e = min(ptid + 256 * i, ROWS * ncg - 1);
            prow[i] = e / ncg; pcg[i] = e % ncg;
        }
        const uint32_t sbytes = (uint32_t)p.xstride * 4u;
        const int lim = (int)min((int64_t)ROWS - 1, N - 1 - rows0);       // last valid row of the block (>= 0)
        const bool masked = MODE == 1 && p.xmask != nullptr;
        // (requests are unconditional - clamped rows, clamped state index, no branch: the compiler then counts what is in flight and
        //  waits for exactly the oldest; wgrad.hip tells what one `if` around a request costs)
        auto issue = [&](int t, float4 (&v)[PF], float4 (&mk)[PF]) {
            t = min(t, T - 1);
            const float* base = p.xs[0];
            if (MODE == 0) {
#pragma unroll
                for (int i = 1; i < 16; ++i)
                    if (t == i) base = p.xs[i];           // (uniform selects, not a runtime index into the argument array)
            }
            const float* cp = base + rows0 * p.xstride;    // (scalar)
            const float* mp = masked ? p.xmask + rows0 * p.xstride : cp;
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const uint32_t off = __umul24((uint32_t)min(prow[i], lim), sbytes) + 16u * (uint32_t)pcg[i];
                v[i] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(cp) + off);
                if (MODE == 1) mk[i] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(mp) + off);
            }
        };
        const bool tail = lim < ROWS - 1;                 // (uniform: only the batch's last block has rows to zero)
        auto commit = [&](int t, float4 (&v)[PF], float4 (&mk)[PF]) {
            if (t >= T) return;                           // (uniform; no request inside)
            __bf16* buf = pl + (t & (NB - 1)) * BUF;
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                float4 x = v[i];
                if (masked) { x.x = mk[i].x > 0.f ? x.x : 0.f; x.y = mk[i].y > 0.f ? x.y : 0.f; x.z = mk[i].z > 0.f ? x.z : 0.f; x.w = mk[i].w > 0.f ? x.w : 0.f; }
                if (tail && prow[i] > lim) x = make_float4(0.f, 0.f, 0.f, 0.f);       // (rows beyond N stay zero)
                bf3_u2 h0, m0, l0, h1, m1, l1;
                bf3_split2(bf3_f2{x.x, x.y}, h0, m0, l0);
                bf3_split2(bf3_f2{x.z, x.w}, h1, m1, l1);
                __bf16* q = buf + prow[i] * PK + 4 * pcg[i];
                *reinterpret_cast<uint2*>(q) = make_uint2(bf3_pack(h0.x, h0.y), bf3_pack(h1.x, h1.y));
                *reinterpret_cast<uint2*>(q + ROWS * PK) = make_uint2(bf3_pack(m0.x, m0.y), bf3_pack(m1.x, m1.y));
                *reinterpret_cast<uint2*>(q + 2 * ROWS * PK) = make_uint2(bf3_pack(l0.x, l0.y), bf3_pack(l1.x, l1.y));
            }
        };
        __builtin_amdgcn_s_setprio(2);
        float4 pv0[PF], pm0[PF];
        if (MODE == 1) {
            issue(0, pv0, pm0);
            commit(0, pv0, pm0);
            __syncthreads();
            return;
        }
        // state t + 1 is staged while state t is multiplied; its registers then take the requests for t + 3.  Two named register
        // sets, the loop unrolled by two; 1 + 2 ceil(S / 2) barriers, as the multiplying waves'
        float4 pv1[PF], pm1[PF];
        issue(0, pv0, pm0);
        issue(1, pv1, pm1);
        commit(0, pv0, pm0);
        issue(2, pv0, pm0);
        __syncthreads();
        for (int t = 0; t < T; t += 2) {
            commit(t + 1, pv1, pm1);
            issue(t + 3, pv1, pm1);
            __syncthreads();
            commit(t + 2, pv0, pm0);
            issue(t + 4, pv0, pm0);
            __syncthreads();
        }
        return;
    }

    // ---- multiplying waves: output columns [32 wave, 32 wave + 32)
    typedef __attribute__((ext_vector_type(16))) float f32x16;
    const int n = wave * 32 + c;
    const bool strip = wave * 32 < O;                 // (uniform)
    const uint4* wf = p.wfrag + (int64_t)wave * KS * 192 + lane;           // this wave's strip; a state's strips are 4 KS 192 items apart
    // position q = l * KS + ks of the fragment stream, clamped at its end (unconditional requests)
    auto load_b = [&](int q, bf3_x8 (&b)[3]) {
        q = min(q, S * KS - 1);
        const uint4* f = wf + ((int64_t)(q / KS) * 4 * KS + q % KS) * 192;
        b[0] = __builtin_bit_cast(bf3_x8, f[0]); b[1] = __builtin_bit_cast(bf3_x8, f[64]); b[2] = __builtin_bit_cast(bf3_x8, f[128]);
    };
    auto ld8 = [&](const __bf16* q) { return __builtin_bit_cast(bf3_x8, *reinterpret_cast<const uint4*>(q)); };
    f32x16 acc[kL3Tiles];
#pragma unroll
    for (int m = 0; m < kL3Tiles; ++m)
        for (int v = 0; v < 16; ++v) acc[m][v] = 0.f;
    // acc[m][v] of lane (c, kg): row 32 m + (v & 3) + 8 (v >> 2) + 4 kg of the block, column n.  One scalar base per call and a
    // running 32-bit lane offset (+1 or +5 rows): 48 separate row pointers and row tests - all loop invariant, so all hoisted -
    // overflowed the scalar registers into vector lanes (3000 cycles per 48 stores).  Only the batch's last block tests rows.
    const uint32_t lane_off = ((uint32_t)(4 * kg) * (uint32_t)p.ystride + (uint32_t)n) * 4u;
    const uint32_t row1 = (uint32_t)p.ystride * 4u, row5 = 5u * row1;
    const int lane_rows = (int)min((int64_t)ROWS, N - rows0) - 4 * kg;        // rows of the block below this lane's first one that exist
    auto store = [&](float* out, float b, bool relu) {
        if (n >= O) return;
        char* base = reinterpret_cast<char*>(out + rows0 * p.ystride);         // (scalar)
        uint32_t off = lane_off;
        if (rows0 + ROWS <= N) {                                               // (uniform)
#pragma unroll
            for (int m = 0; m < kL3Tiles; ++m)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    float val = acc[m][v] + b;
                    if (relu) val = fmaxf(val, 0.f);
                    *reinterpret_cast<float*>(base + off) = val;
                    off += (v & 3) == 3 ? row5 : row1;
                }
        } else {
#pragma unroll
            for (int m = 0; m < kL3Tiles; ++m)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    float val = acc[m][v] + b;
                    if (relu) val = fmaxf(val, 0.f);
                    if (32 * m + (v & 3) + 8 * (v >> 2) < lane_rows) *reinterpret_cast<float*>(base + off) = val;
                    off += (v & 3) == 3 ? row5 : row1;
                }
        }
    };
    bf3_x8 b0[3], b1[3];
    load_b(0, b0);
    load_b(1, b1);
    __syncthreads();                                  // buffer 0 is staged
    for (int l = 0; l < S; ++l) {
        if (strip) {
            const __bf16* ap = pl + (MODE == 0 ? (l & 1) : 0) * BUF + c * PK + 8 * kg;
            bf3_x8 ah = ld8(ap), am = ld8(ap + ROWS * PK), al = ld8(ap + 2 * ROWS * PK);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf3_x8 b2[3];
                load_b(l * KS + ks + 2, b2);          // two k steps ahead (across the state boundary too)
#pragma unroll
                for (int m = 0; m < kL3Tiles; ++m) {
                    bf3_x8 nh = ah, nm = am, nl = al;
                    const int j = ks * kL3Tiles + m + 1;          // the next (k step, tile) of this state, if any
                    if (j < KS * kL3Tiles) {
                        const __bf16* np = ap + (j % kL3Tiles) * 32 * PK + 16 * (j / kL3Tiles);
                        nh = ld8(np); nm = ld8(np + ROWS * PK); nl = ld8(np + 2 * ROWS * PK);
                    }
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b0[0], acc[m], 0, 0, 0);      // smallest terms first
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b0[2], acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b0[1], acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b0[0], acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b0[1], acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b0[0], acc[m], 0, 0, 0);
                    // (the next fragments' LDS reads FIRST, then the six matrix instructions: left alone the scheduler sinks the reads
                    //  to the end of the group and the next group starts by waiting out their latency - 88 cycles per instruction)
                    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    ah = nh; am = nm; al = nl;
                }
#pragma unroll
                for (int i = 0; i < 3; ++i) { b0[i] = b1[i]; b1[i] = b2[i]; }
            }
        }
        if (MODE == 0) {
            __syncthreads();                          // the staging waves are done with state l + 1, these waves with state l
        } else if (strip) {
            store(p.y + (int64_t)l * p.yblock, 0.f, false);
#pragma unroll
            for (int m = 0; m < kL3Tiles; ++m)
                for (int v = 0; v < 16; ++v) acc[m][v] = 0.f;
        }
    }
    if (MODE == 0) {
        if (S & 1) __syncthreads();                   // (the staging loop runs in pairs of states)
        if (strip) store(p.y, p.bias && n < O ? p.bias[n] : 0.f, p.relu != 0);
    }
}

template <int MODE>
int lin3_launch(const L3Params& p, hipStream_t s) {
    const int ks = (p.I + 15) / 16;
    const int64_t grid = (p.N + kL3Rows - 1) / kL3Rows;
    const size_t lds = (size_t)(MODE == 0 ? 2 : 1) * l3_buf(ks) * 2;
#define KP_L3(KSV) do { \
        KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)lin3_kernel<KSV, MODE>, lds)); \
        hipLaunchKernelGGL((lin3_kernel<KSV, MODE>), dim3((unsigned)grid), dim3(512), lds, s, p); } while (0)
    switch (ks) {
        case 2: KP_L3(2); break;
        case 4: KP_L3(4); break;
        case 6: KP_L3(6); break;
        case 7: KP_L3(7); break;
        default: KP_L3(8); break;
    }
#undef KP_L3
    KPGNN_LAUNCH_CHECK("lin3_kernel");
    return KPGNN_OK;
}

int lin3_split_w(const float* w, int64_t wn, int64_t wk, int64_t ws, int S, int O, int I, uint4* frag, hipStream_t s) {
    const int ks = (I + 15) / 16;
    const int64_t threads = (int64_t)S * 4 * ks * 64;
    hipLaunchKernelGGL(lin3_wsplit_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, w, wn, wk, ws, S, O, I, ks, frag);
    KPGNN_LAUNCH_CHECK("lin3_wsplit_kernel");
    return KPGNN_OK;
}

bool lin3_shape_ok(int64_t N, int O, int I, int64_t xstride) {
    return N >= 4096 && O <= 128 && O % 4 == 0 && (I == 32 || I == 64 || I == 96 || I == 104 || I == 128) && xstride % 4 == 0 &&
           xstride < (1 << 22);
}

}  // namespace

int linear3_split_w(const float* w, int64_t wn, int64_t wk, int O, int I, void* frag, hipStream_t s) {
    return lin3_split_w(w, wn, wk, 0, 1, O, I, (uint4*)frag, s);
}

size_t linear3_workspace_bytes(int O, int I, int group) {
    if (O < 1 || O > 128 || I < 1 || I > 128 || group < 1 || group > 16) return 0;
    return (size_t)group * 4 * ((I + 15) / 16) * 3 * 64 * sizeof(uint4);
}

int linear3_group_fwd(const kpgnn_linear_group_desc* d, hipStream_t s, bool* handled) {
    *handled = false;
    if (d->math == KPGNN_MATH_F32 || !d->workspace || d->workspace_bytes < linear3_workspace_bytes(d->O, d->I, d->group) ||
        (((uintptr_t)d->workspace) & 15) != 0 || !lin3_shape_ok(d->N, d->O, d->I, d->x_stride))
        return KPGNN_OK;
    L3Params p = {};
    p.N = d->N; p.n_dyn = d->n_dyn; p.O = d->O; p.I = d->I; p.S = d->group; p.relu = d->relu ? 1 : 0;
    for (int l = 0; l < 16; ++l) p.xs[l] = d->x[l < d->group ? l : 0];
    p.xstride = d->x_stride; p.xmask = nullptr;
    p.wfrag = (const uint4*)d->workspace;
    p.bias = d->bias; p.y = d->y; p.ystride = d->O; p.yblock = 0;
    *handled = true;
    // w [O, S * I]: element (k, n) of state l at n * S * I + l * I + k
    const int rc = lin3_split_w(d->w, (int64_t)d->group * d->I, 1, d->I, d->group, d->O, d->I, (uint4*)d->workspace, s);
    return rc != KPGNN_OK ? rc : lin3_launch<0>(p, s);
}

int linear3_blocked(const kpgnn_linear_desc* d, hipStream_t s, bool* handled) {
    *handled = false;
    const int yb = d->y_block_cols;
    if (d->math == KPGNN_MATH_F32 || yb <= 0 || d->O % yb != 0 || d->O / yb > 16 || d->bias || !d->workspace ||
        d->workspace_bytes < linear3_workspace_bytes(yb, d->I, d->O / yb) || (((uintptr_t)d->workspace) & 15) != 0 ||
        !lin3_shape_ok(d->N, yb, d->I, d->x_stride))
        return KPGNN_OK;
    const int S = d->O / yb;
    L3Params p = {};
    p.N = d->N; p.n_dyn = d->n_dyn; p.O = yb; p.I = d->I; p.S = S; p.relu = 0;
    for (int l = 0; l < 16; ++l) p.xs[l] = d->x;
    p.xstride = d->x_stride; p.xmask = d->x_mask;
    p.wfrag = (const uint4*)d->workspace;
    p.bias = nullptr; p.y = d->y; p.ystride = d->y_stride; p.yblock = d->y_block_stride;
    *handled = true;
    int rc;
    if (d->w_transposed) rc = lin3_split_w(d->w, 1, d->O, yb, S, yb, d->I, (uint4*)d->workspace, s);          // w [I, O]: (k, n) of block l at k * O + l * yb + n
    else rc = lin3_split_w(d->w, d->I, 1, (int64_t)yb * d->I, S, yb, d->I, (uint4*)d->workspace, s);            // w [O, I]
    return rc != KPGNN_OK ? rc : lin3_launch<1>(p, s);
}

}  // namespace kpgnn

extern "C" int kpgnn_linear_split_many(const kpgnn_split_job* jobs, int32_t n, kpgnn_stream_t stream) {
    using namespace kpgnn;
    KPGNN_REQUIRE(jobs != nullptr && n >= 1 && n <= 64, "linear_split_many: 1 <= n <= 64 jobs (n = %d)", n);
    SplitMany a;
    int kmax = 0;
    for (int i = 0; i < 64; ++i) {
        a.j[i] = jobs[i < n ? i : 0];
        if (i < n) {
            const kpgnn_split_job& q = jobs[i];
            KPGNN_REQUIRE(q.w && q.frag && q.O >= 1 && q.O <= 128 && q.I >= 1 && q.I <= 128 && (((uintptr_t)q.frag) & 15) == 0,
                          "linear_split_many: job %d: bad pointers or O=%d / I=%d beyond 128", i, q.O, q.I);
            const int ks = (q.I + 15) / 16;
            kmax = ks > kmax ? ks : kmax;
        }
    }
    hipLaunchKernelGGL(lin3_wsplit_many_kernel, dim3((unsigned)((4 * kmax * 64 + 255) / 256), (unsigned)n), dim3(256), 0, (hipStream_t)stream, a);
    KPGNN_LAUNCH_CHECK("lin3_wsplit_many_kernel");
    return KPGNN_OK;
}

