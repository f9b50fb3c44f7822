// dW = dy^T x, db = sum dy for tall-skinny activations on the fp32 matrix cores (gfx950).
// Contract: include/kpgnn.h, kpgnn_linear_wgrad.
//
// v_mfma_f32_32x32x2_f32 (exact f32, fmaf-chain numerics): lane l feeds A[i = l&31][k = l>>5] and
// B[k = l>>5][j = l&31]; with A = dy^T and B = x and k = two consecutive rows r0, r0+1 both operands are plain
// coalesced reads of 32 consecutive floats of a row - straight from global memory, no LDS, no transposes.
// A wave owns a strip of output rows o in [32s, 32s+32) and ALL column tiles (TI accumulators of 16 VGPRs); the N rows
// are split over blocks and - for O <= 128 - over the RG = 2 row groups of a block, whose partial strips meet in LDS
// (fixed order) before ONE partial per block goes to a slab that is added in block order.  The slab is what a launch
// leaves behind: 512 single-group blocks wrote (and the reduce launch re-read) 44.7 MB per pair of 104 x 104 problems,
// against 80 MB of operands; two groups per block halve it.
#include <cstdlib>
#include <type_traits>

#include "bf3.h"
#include "kpgnn_common.h"

namespace kpgnn {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int kWgradBlocks = 512;

struct WgParams {
    int64_t N; int O, I;
    const float* dy; int64_t dys;
    const float* x; int64_t xs;
    const float* xm; const float* xi; const float* xg; const float* xb; int xrelu;   // optional BatchNorm(+ReLU) of x on load
    float* slab;   // [gridDim.x][slab_row]; this problem's strip starts at slab_off
    int64_t slab_row, slab_off;
};
struct WgPair { WgParams q[2]; };

// One problem, worked through by the whole grid.  `p` is passed BY VALUE from a compile-time index of the kernel argument,
// so its fields live in scalar registers (indexing the argument array with a runtime problem index made every use a
// scalar load from the kernel-argument segment inside the row loop: 46 instead of 28 us per problem).
template <int TI, int RG>
__device__ __forceinline__ void wgrad_problem(const WgParams p, float* lds) {
    const int lane = threadIdx.x & 63;
    const int nstrip = (int)(blockDim.x >> 6) / RG;
    const int wave = (int)(threadIdx.x >> 6) % nstrip, rg = (int)(threadIdx.x >> 6) / nstrip;   // strip, row group
    const int kk = lane >> 5, c = lane & 31;
    const int o = wave * 32 + c;                 // this lane's dy column
    const int64_t vgrid = (int64_t)gridDim.x * RG, vb = (int64_t)blockIdx.x * RG + rg;          // row-group granularity
    const bool o_ok = o < p.O;
    f32x16 acc[TI];
#pragma unroll
    for (int t = 0; t < TI; ++t)
        for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
    float bsum = 0.f;
    float tm[TI], ts[TI], tg[TI], tb[TI];        // per-lane columns i = t*32 + c: fixed for the whole problem
    const bool tr = p.xm != nullptr;
    if (tr) {
#pragma unroll
        for (int t = 0; t < TI; ++t) {
            const int i = t * 32 + c;
            const bool ok = i < p.I;
            tm[t] = ok ? p.xm[i] : 0.f; ts[t] = ok ? p.xi[i] : 0.f; tg[t] = ok ? p.xg[i] : 0.f; tb[t] = ok ? p.xb[i] : 0.f;
        }
    }
    // rows of this block: pairs (r, r+1); row group vb takes pairs vb, vb + vgrid, ...; UN pairs make a batch.
    // Full batches (all but the last) load WITHOUT predicates from a wave-uniform row pointer plus a per-lane 32-bit
    // offset fixed for the whole problem: lanes of padded columns (o >= O, i >= I) are pointed at the last valid column -
    // what they accumulate lands in output rows / columns that are never stored.  (Per-load 64-bit address arithmetic and
    // the row / column selects were a third of the loop's issue slots next to the 4 MFMAs per pair.)
    const int64_t pairs = (p.N + 1) / 2;
    constexpr int UN = 4;
    const int dyo = kk * (int)p.dys + (o_ok ? o : p.O - 1);
    int xo[TI];
#pragma unroll
    for (int t = 0; t < TI; ++t) xo[t] = kk * (int)p.xs + min(t * 32 + c, p.I - 1);
    for (int64_t pr = vb; pr < pairs; pr += vgrid * UN) {
        float a[UN], b[UN][TI];
        const bool full = 2 * (pr + (int64_t)(UN - 1) * vgrid) + 1 < p.N;      // wave-uniform
        if (full) {
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int64_t r0 = 2 * (pr + (int64_t)u * vgrid);
                const float* dyr = p.dy + r0 * p.dys;
                const float* xr = p.x + r0 * p.xs;
                a[u] = dyr[dyo];
#pragma unroll
                for (int t = 0; t < TI; ++t) b[u][t] = xr[xo[t]];
            }
            if (tr) {
#pragma unroll
                for (int u = 0; u < UN; ++u)
#pragma unroll
                    for (int t = 0; t < TI; ++t) {
                        float v = fmaf((b[u][t] - tm[t]) * ts[t], tg[t], tb[t]);
                        if (p.xrelu) v = fmaxf(v, 0.f);
                        b[u][t] = v;
                    }
            }
        } else {
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int64_t r = 2 * (pr + (int64_t)u * vgrid) + kk;
                const bool r_ok = r < p.N;
                a[u] = (r_ok && o_ok) ? p.dy[r * p.dys + o] : 0.f;
#pragma unroll
                for (int t = 0; t < TI; ++t) {
                    const int i = t * 32 + c;
                    b[u][t] = (r_ok && i < p.I) ? p.x[r * p.xs + i] : 0.f;
                }
            }
            if (tr) {
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const bool r_ok = 2 * (pr + (int64_t)u * vgrid) + kk < p.N;
#pragma unroll
                    for (int t = 0; t < TI; ++t) {
                        float v = fmaf((b[u][t] - tm[t]) * ts[t], tg[t], tb[t]);
                        if (p.xrelu) v = fmaxf(v, 0.f);
                        b[u][t] = (r_ok && t * 32 + c < p.I) ? v : 0.f;
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            bsum += a[u];
#pragma unroll
            for (int t = 0; t < TI; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u][t], acc[t], 0, 0, 0);
        }
    }
    bsum += __shfl_xor(bsum, 32);                // the two k halves hold different rows of the same column
    if (RG > 1) {                                // group 1's partial strip joins group 0's: registers -> LDS -> registers
        float* sl = lds + (size_t)wave * (TI * 16 + 1) * 64 + lane;
        __syncthreads();                         // (a previous problem's readers are done with the buffer)
        if (rg == 1) {
#pragma unroll
            for (int t = 0; t < TI; ++t)
                for (int v = 0; v < 16; ++v) sl[(t * 16 + v) * 64] = acc[t][v];
            sl[TI * 16 * 64] = bsum;
        }
        __syncthreads();
        if (rg == 0) {
#pragma unroll
            for (int t = 0; t < TI; ++t)
                for (int v = 0; v < 16; ++v) acc[t][v] += sl[(t * 16 + v) * 64];
            bsum += sl[TI * 16 * 64];
        }
    }
    // C/D map: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float* out = p.slab + (int64_t)blockIdx.x * p.slab_row + p.slab_off;
    if (rg == 0) {
#pragma unroll
    for (int t = 0; t < TI; ++t) {
        const int i = t * 32 + c;
        for (int v = 0; v < 16; ++v) {
            const int orow = wave * 32 + (v & 3) + 8 * (v >> 2) + 4 * kk;
            if (orow < p.O && i < p.I) out[(int64_t)orow * p.I + i] = acc[t][v];
        }
    }
    if (kk == 0 && o_ok) out[(int64_t)p.O * p.I + o] = bsum;
    }
}

// Two problems of one launch are worked through one after the other by the SAME grid: each keeps the single launch's
// access pattern and block count (side by side on half the blocks each they measured 94 us against 2 x 28 us).
template <int TI, int RG>
__global__ void __launch_bounds__(512)
wgrad_kernel(const WgPair pp, int nprob) {
    extern __shared__ __attribute__((aligned(16))) float wg_lds[];     // RG 2: [strips][TI * 16 + 1][64]
    if (gridDim.y > 1) {       // small batches: the two problems side by side (a block's serial chain is what a launch costs there)
        if (blockIdx.y == 0) wgrad_problem<TI, RG>(pp.q[0], wg_lds);
        else wgrad_problem<TI, RG>(pp.q[1], wg_lds);
        return;
    }
    wgrad_problem<TI, RG>(pp.q[0], wg_lds);
    if (nprob > 1) wgrad_problem<TI, RG>(pp.q[1], wg_lds);
}

// Blocks of a launch: every block writes a full O x I partial to the slab, so a small batch must not spread its few rows
// over 512 blocks (N = 1.5k: 44 MB of slab written and re-read for 1.2 MB of operands); >= 16 rows per block (32 measured
// 1.27 against 1.26 ms per batch-64 step: the shorter per-block chain is worth the larger slab).
int wgrad_grid(int64_t N) {
    int64_t g = (N + 15) / 16;
    if (g > kWgradBlocks) g = kWgradBlocks;
    return (int)(g < 1 ? 1 : g);
}

int wgrad_check(const kpgnn_wgrad_desc* d, const char* who) {
    KPGNN_REQUIRE(d != nullptr, "%s: NULL descriptor", who);
    KPGNN_REQUIRE(d->N >= 1 && d->O >= 1 && d->I >= 1, "%s: bad N=%lld O=%d I=%d", who, (long long)d->N, d->O, d->I);
    if (d->O > 256 || d->I > 256) return fail(KPGNN_ELIMIT, "%s: O=%d, I=%d exceed 256", who, d->O, d->I);
    KPGNN_REQUIRE(d->dy && d->x && d->dw && d->dy_stride >= d->O && d->x_stride >= d->I, "%s: bad pointers/strides", who);
    KPGNN_REQUIRE(!d->x_mean || (d->x_invstd && d->x_gamma && d->x_beta), "%s: x transform needs mean, invstd, gamma, beta", who);
    return KPGNN_OK;
}

WgParams wgrad_params(const kpgnn_wgrad_desc* d, float* slab, int64_t slab_row, int64_t slab_off) {
    WgParams p;
    p.N = d->N; p.O = d->O; p.I = d->I; p.dy = d->dy; p.dys = d->dy_stride; p.x = d->x; p.xs = d->x_stride;
    p.xm = d->x_mean; p.xi = d->x_invstd; p.xg = d->x_gamma; p.xb = d->x_beta; p.xrelu = d->x_relu;
    p.slab = slab; p.slab_row = slab_row; p.slab_off = slab_off;
    return p;
}

int wgrad_groups(int O) { return O <= 128 ? 2 : 1; }      // row groups per block (a block has at most 8 waves)

// real blocks of a launch = slab rows
int wgrad_blocks(int64_t N, int O) {
    const int rg = wgrad_groups(O);
    return (wgrad_grid(N) + rg - 1) / rg;
}

int wgrad_launch(const WgPair& pp, int nprob, int O, int I, int grid, hipStream_t s) {
    const int waves = (O + 31) / 32, ti = (I + 31) / 32, rg = wgrad_groups(O);
    // (two problems: one after the other on the same grid when the grid fills the chip - side by side they measured 94 us
    //  against 2 x 28 - but side by side when there are only a few dozen blocks)
    dim3 blk(waves * 64 * rg), gr(grid, (nprob > 1 && grid * rg <= 128) ? 2 : 1);
    const size_t lds = rg > 1 ? sizeof(float) * (size_t)waves * (ti * 16 + 1) * 64 : 0;
#define KP_WG(T) do { \
        if (rg > 1) { KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)wgrad_kernel<T, 2>, lds)); \
                      hipLaunchKernelGGL((wgrad_kernel<T, 2>), gr, blk, lds, s, pp, nprob); } \
        else hipLaunchKernelGGL((wgrad_kernel<T, 1>), gr, blk, 0, s, pp, nprob); } while (0)
    switch (ti) {
        case 1: KP_WG(1); break;
        case 2: KP_WG(2); break;
        case 3: KP_WG(3); break;
        case 4: KP_WG(4); break;
        case 5: KP_WG(5); break;
        case 6: KP_WG(6); break;
        case 7: KP_WG(7); break;
        default: KP_WG(8); break;
    }
#undef KP_WG
    KPGNN_LAUNCH_CHECK("wgrad_kernel");
    return KPGNN_OK;
}

// ------------------------------------------------------------------------------------------------ LDS-staged variant
// The kernel above feeds the matrix cores straight from global memory with one dword per lane and operand: x is fetched
// once per output strip (4x through L1) in 256-byte wave requests, and the launch sat at 35 % matrix-core occupancy whatever
// was done to its latency.  Here a block stages 32 rows of dy and x in LDS with 16-byte loads (requested one
// chunk ahead, held in registers across the MFMA phase) and every strip reads its operands from there: x crosses L1 once,
// the optional transforms (BatchNorm + ReLU of x, ReLU mask of dy) are applied once per element on the way in.
// Several problems that share N, O, I run SIDE BY SIDE on disjoint block ranges of one launch (the two Linears of an MLP;
// the S column blocks of a jumping-knowledge projection, which also share dy): the slab keeps the size of a single problem's.
constexpr int kW2Rows = 32;
constexpr int kW2MaxProb = 16;

struct Wg2Prob {
    const float* dy; const float* x; const float* mask;
    const float* xm; const float* xi; const float* xg; const float* xb;
    int xrelu;
    int64_t dys, xs, out_off, bias_off;      // offsets inside a slab row; bias_off < 0: this problem writes no bias gradient
};
struct Wg2Args {
    int64_t N; const int32_t* n_dyn;
    int O, I, nprob; int64_t ldw;
    float* slab; int64_t slab_row;
    Wg2Prob q[kW2MaxProb];
};

template <int TI>
__global__ void __launch_bounds__(256, 2)
wgrad2_kernel(const Wg2Args A) {
    extern __shared__ __attribute__((aligned(16))) float w2[];
    constexpr int R = kW2Rows;
    const int O = A.O, I = A.I;
    float* dyl = w2;                       // [R][O]
    float* xl = w2 + R * O;                // [R][I]
    float* coef = xl + R * I;              // [4][I]  (mean, invstd, gamma, beta of the x transform)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int kk = lane >> 5, c = lane & 31;
    const int nprob = A.nprob;
    const int pi = (int)(blockIdx.x % nprob), slice = (int)(blockIdx.x / nprob), nslices = (int)(gridDim.x / nprob);
    Wg2Prob q = A.q[0];                    // (uniform selects: a runtime index into the argument array would turn every use into
#pragma unroll                             //  a scalar load from the kernel-argument segment)
    for (int i = 1; i < kW2MaxProb; ++i)
        if (pi == i) q = A.q[i];
    const int64_t N = A.n_dyn ? (int64_t)min((int64_t)*A.n_dyn, A.N) : A.N;
    const bool tr = q.xm != nullptr;
    if (tr)
        for (int i = tid; i < I; i += 256) { coef[i] = q.xm[i]; coef[I + i] = q.xi[i]; coef[2 * I + i] = q.xg[i]; coef[3 * I + i] = q.xb[i]; }
    // this thread's float4 slots of the two tiles (fixed for the whole launch)
    const int o4 = O >> 2, i4 = I >> 2;
    int dy_row[4], dy_cg[4], x_row[TI], x_cg[TI];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int e = tid + 256 * j; dy_row[j] = e < R * o4 ? e / o4 : -1; dy_cg[j] = e % o4; }
#pragma unroll
    for (int j = 0; j < TI; ++j) { const int e = tid + 256 * j; x_row[j] = e < R * i4 ? e / i4 : -1; x_cg[j] = e % i4; }
    float4 pdy[4], pm[4], px[TI];
    const bool has_mask = q.mask != nullptr;
    // Loads are UNCONDITIONAL (rows beyond N and slots this thread does not own are clamped to a valid address and zeroed /
    // skipped at commit time): a predicated load compiles to a divergent branch with an `s_waitcnt vmcnt(0)` at its join, which
    // serialised the seven requests of a chunk into seven round trips.
    const int64_t last = N - 1;            // (N >= 1)
    auto issue = [&](int64_t ch) {
        const int64_t r0 = ch * R;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t r = min(r0 + max(dy_row[j], 0), last);
            pdy[j] = *reinterpret_cast<const float4*>(q.dy + r * q.dys + 4 * dy_cg[j]);
            if (has_mask) pm[j] = *reinterpret_cast<const float4*>(q.mask + r * q.dys + 4 * dy_cg[j]);
        }
#pragma unroll
        for (int j = 0; j < TI; ++j) {
            const int64_t r = min(r0 + max(x_row[j], 0), last);
            px[j] = *reinterpret_cast<const float4*>(q.x + r * q.xs + 4 * x_cg[j]);
        }
    };
    auto commit = [&](int64_t ch) {
        const int64_t r0 = ch * R;
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float4 v = pdy[j];
            if (has_mask) { v.x = pm[j].x > 0.f ? v.x : 0.f; v.y = pm[j].y > 0.f ? v.y : 0.f; v.z = pm[j].z > 0.f ? v.z : 0.f; v.w = pm[j].w > 0.f ? v.w : 0.f; }
            if (r0 + dy_row[j] >= N) v = zero;
            if (dy_row[j] >= 0) *reinterpret_cast<float4*>(dyl + dy_row[j] * O + 4 * dy_cg[j]) = v;
        }
#pragma unroll
        for (int j = 0; j < TI; ++j) {
            float4 v = px[j];
            if (tr) {
                const float4 m = *reinterpret_cast<const float4*>(coef + 4 * x_cg[j]), s = *reinterpret_cast<const float4*>(coef + I + 4 * x_cg[j]);
                const float4 g = *reinterpret_cast<const float4*>(coef + 2 * I + 4 * x_cg[j]), b = *reinterpret_cast<const float4*>(coef + 3 * I + 4 * x_cg[j]);
                v.x = fmaf((v.x - m.x) * s.x, g.x, b.x); v.y = fmaf((v.y - m.y) * s.y, g.y, b.y);
                v.z = fmaf((v.z - m.z) * s.z, g.z, b.z); v.w = fmaf((v.w - m.w) * s.w, g.w, b.w);
                if (q.xrelu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            }
            if (r0 + x_row[j] >= N) v = zero;     // (rows beyond N must stay zero)
            if (x_row[j] >= 0) *reinterpret_cast<float4*>(xl + x_row[j] * I + 4 * x_cg[j]) = v;
        }
    };
    f32x16 acc[TI];
#pragma unroll
    for (int t = 0; t < TI; ++t)
        for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
    float bsum = 0.f;
    const int o = wave * 32 + c;
    const bool o_ok = o < O;
    // (padded lanes read the last valid column: what they accumulate lands in rows / columns that are never stored)
    const float* ap = dyl + kk * O + (o_ok ? o : O - 1);
    const float* bp[TI];
#pragma unroll
    for (int t = 0; t < TI; ++t) bp[t] = xl + kk * I + min(t * 32 + c, I - 1);
    const int64_t chunks = (N + R - 1) / R;
    int64_t ch = slice;
    if (tr) __syncthreads();
    if (ch < chunks) { issue(ch); commit(ch); }
    __syncthreads();
    for (; ch < chunks; ch += nslices) {
        const int64_t nx = ch + nslices;
        const bool more = nx < chunks;
        if (more) issue(nx);
#pragma unroll
        for (int p2 = 0; p2 < R / 2; ++p2) {
            const float a = ap[2 * p2 * O];
            bsum += a;
#pragma unroll
            for (int t = 0; t < TI; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[t][2 * p2 * I], acc[t], 0, 0, 0);
        }
        __syncthreads();                              // every wave is done with this chunk's tiles
        if (more) commit(nx);
        __syncthreads();
    }
    bsum += __shfl_xor(bsum, 32);                     // the two k halves hold different rows of the same column
    float* out = A.slab + (int64_t)slice * A.slab_row;
#pragma unroll
    for (int t = 0; t < TI; ++t) {
        const int i = t * 32 + c;
        for (int v = 0; v < 16; ++v) {
            const int orow = wave * 32 + (v & 3) + 8 * (v >> 2) + 4 * kk;
            if (orow < O && i < I) out[q.out_off + (int64_t)orow * A.ldw + i] = acc[t][v];
        }
    }
    if (q.bias_off >= 0 && kk == 0 && o_ok) out[q.bias_off + o] = bsum;
}

// ------------------------------------------------------------------------------------------------ bf16-split variant
// The fp32 matrix instruction above runs at 1/16 of the bf16 rate and the chip holds ~1.6 GHz under it: the kernel sits at
// 65-85 TFLOP/s, twice its operand time from HBM.  An fp32 value is EXACTLY the sum of three bf16 pieces (8 + 8 + 8 significant
// bits, split by truncation: h = top 16 bits of v, m = top 16 bits of v - h, l = v - h - m), so
//     a b = ah bh + (ah bm + am bh) + (am bm + ah bl + al bh) + [three terms below 2^-24 |a b|: dropped]:
// six exact bf16 products per k on v_mfma_f32_32x32x16_bf16 with fp32 accumulation - 6/16 of the matrix time of the fp32
// instruction for the rounding error of fp32 accumulation itself (smallest terms first within a k step).  The bf16 instruction wants 8 consecutive k (= rows) of one column per lane, so the
// chunk is staged COLUMN-major: a thread fetches 8 rows x 4 columns (8 float4 requests, a chunk ahead), splits them and
// writes 4 x 3 packed 16-byte items; 80-byte column pitch = conflict-free 16-byte reads.
constexpr int kW3Pitch = 40;      // bf16 per staged column: 32 rows + 8 of padding

// Roles: waves 0-3 only multiply (one 32-row output strip each), waves 4-7 only stage (4-5 dy, 6-7 x) into the OTHER of two
// LDS buffers, their global requests two chunks ahead in two register sets; one barrier per chunk.
// Measured (s_memtime per phase, [47450,104] operands): a chunk costs ~3750 cycles - the multiplying wave 2200 (48 matrix
// instructions = 1536, the rest exposed LDS read latency), the staging wave on the same SIMD 400-1300 to have its rows, 1050
// (alone) to 2300 (next to a multiplying wave) for ~160 vector instructions and 12 stores, 400-900 to request the next rows:
// vector and matrix work of two waves on one SIMD add up rather than overlap.  29.6 us per pair of problems against 47.4 for
// the fp32 kernel, 120 against 191 for nine problems (the jumping-knowledge projection); the operands' HBM time is 16 / 40 us.
constexpr int kW3Blocks = 256;    // one 8-wave block per CU

typedef __attribute__((ext_vector_type(2))) float bf3_f2;
typedef __attribute__((ext_vector_type(2))) uint32_t bf3_u2;

// three-way split of two values at once (packed subtracts): word pairs whose top halves are the bf16 pieces
__device__ __forceinline__ void bf3_split2(const bf3_f2 v, bf3_u2& h, bf3_u2& m, bf3_u2& l) {
    h = __builtin_bit_cast(bf3_u2, v) & 0xffff0000u;
    const bf3_f2 r1 = v - __builtin_bit_cast(bf3_f2, h);
    m = __builtin_bit_cast(bf3_u2, r1) & 0xffff0000u;
    l = __builtin_bit_cast(bf3_u2, r1 - __builtin_bit_cast(bf3_f2, m));
}

template <int TI, bool MASK>
__global__ void __launch_bounds__(512, 1)
wgrad3_kernel(const Wg2Args A) {
    extern __shared__ __attribute__((aligned(16))) float w2[];
    constexpr int R = kW2Rows, P = kW3Pitch;
    const int O = A.O, I = A.I;
    const int bufsz = 3 * (O + I) * P;                     // bf16 items per buffer: [3][O][P] pieces of dy, then [3][I][P] of x
    __bf16* pl = reinterpret_cast<__bf16*>(w2);
    float* coef = reinterpret_cast<float*>(pl + 2 * bufsz);   // [4][I] x transform; afterwards [4][O] bias partials
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (scalar: the role tests below become scalar branches)
    const int kg = lane >> 5, c = lane & 31;
    const int nprob = A.nprob;
    const int pi = (int)(blockIdx.x % nprob), slice = (int)(blockIdx.x / nprob), nslices = (int)(gridDim.x / nprob);
    Wg2Prob q = A.q[0];
#pragma unroll
    for (int i = 1; i < kW2MaxProb; ++i)
        if (pi == i) q = A.q[i];
    const int64_t N = A.n_dyn ? (int64_t)min((int64_t)*A.n_dyn, A.N) : A.N;
    const bool tr = q.xm != nullptr;
    if (tr)
        for (int i = tid; i < I; i += 512) { coef[i] = q.xm[i]; coef[I + i] = q.xi[i]; coef[2 * I + i] = q.xg[i]; coef[3 * I + i] = q.xb[i]; }
    const int64_t chunks = (N + R - 1) / R;
    const int64_t mine = slice < chunks ? (chunks - slice + nslices - 1) / nslices : 0;     // chunks slice, slice + nslices, ...
    const bool producer = wave >= 4;
    // ---- staging task of a producer thread, fixed for the launch: (row group of 8, 4 columns) of dy (waves 4-5) or x (6-7)
    const bool is_x = wave >= 6;
    const int ncg = (is_x ? I : O) >> 2;
    const int task = tid & 127;
    const bool owner = producer && task < 4 * ncg;
    // (row group fastest: the 8 contiguous lanes a 16-byte LDS store is serviced in then hit 8 distinct bank quads - column
    //  fastest was a 4-way conflict on every store - and a wave's request still covers 256-byte pieces of 4 rows)
    const int g = owner ? task & 3 : 0, cg = owner ? task >> 2 : 0;
    const float* src = is_x ? q.x : q.dy;
    const int64_t sstride = is_x ? q.xs : q.dys;
    const uint32_t sbytes = (uint32_t)sstride * 4u;           // (row pitch in bytes < 2^24: checked by the host)
    const bool xform = tr && is_x;
    const int64_t last = N - 1;
    bf3_f2 bs01 = {0.f, 0.f}, bs23 = {0.f, 0.f};            // dy tasks: column sums of this thread's rows (the bias gradient)
    // The staging loop, instantiated per role (M: this wave applies the ReLU mask to what it stages, i.e. it stages dy and there is one).
    auto produce = [&](auto mask_c) {
        constexpr bool M = decltype(mask_c)::value;
        float4 pv0[8], pm0[8], pv1[8], pm1[8];
        // Requests are unconditional in every respect - clamped rows, clamped chunk index, no branch on the role or the mask: the
        // compiler then knows how many requests are in flight at each use and waits for exactly the oldest (one `if` around an
        // issue and it must assume the younger set was never requested: vmcnt(7) at the first use, i.e. a wait for both sets).
        auto issue = [&](int64_t k, float4 (&pv)[8], float4 (&pm)[8]) {
            const int64_t c0 = (slice + min(k, mine - 1) * nslices) * R;          // (scalar)
            const float* cp = src + c0 * sstride;                                  // (scalar) the chunk's first row
            const float* mp = M ? q.mask + c0 * sstride : nullptr;
            const int lim = (int)min((int64_t)R, N - c0) - 1;                     // (scalar) last valid row of the chunk
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                // (rows beyond N: the chunk's last valid row.  Unsigned 32-bit BYTE offset from a scalar base: one min, one 24-bit mad)
                const uint32_t off = __umul24((uint32_t)min(8 * g + j, lim), sbytes) + 16u * (uint32_t)cg;
                pv[j] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(cp) + off);
                if (M) pm[j] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(mp) + off);
            }
        };
        auto commit = [&](int64_t k, float4 (&pv)[8], float4 (&pm)[8]) {       // (k >= mine: the clamped chunk again, into a buffer nobody reads)
            const int64_t c0 = (slice + min(k, mine - 1) * nslices) * R;
            if (M) {
    #pragma unroll
                for (int j = 0; j < 8; ++j) {
                    pv[j].x = pm[j].x > 0.f ? pv[j].x : 0.f; pv[j].y = pm[j].y > 0.f ? pv[j].y : 0.f;
                    pv[j].z = pm[j].z > 0.f ? pv[j].z : 0.f; pv[j].w = pm[j].w > 0.f ? pv[j].w : 0.f;
                }
            }
            if (xform) {
                const float4 m4 = *reinterpret_cast<const float4*>(coef + 4 * cg), s4 = *reinterpret_cast<const float4*>(coef + I + 4 * cg);
                const float4 g4 = *reinterpret_cast<const float4*>(coef + 2 * I + 4 * cg), b4 = *reinterpret_cast<const float4*>(coef + 3 * I + 4 * cg);
    #pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float4 v = pv[j];
                    v.x = fmaf((v.x - m4.x) * s4.x, g4.x, b4.x); v.y = fmaf((v.y - m4.y) * s4.y, g4.y, b4.y);
                    v.z = fmaf((v.z - m4.z) * s4.z, g4.z, b4.z); v.w = fmaf((v.w - m4.w) * s4.w, g4.w, b4.w);
                    if (q.xrelu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    pv[j] = v;
                }
            }
            if (c0 + R > N) {                                                      // (uniform: only a launch's last chunk) rows beyond N stay zero
    #pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (c0 + 8 * g + j >= N) pv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (!is_x && k < mine) {
    #pragma unroll
                for (int j = 0; j < 8; ++j) { bs01 += bf3_f2{pv[j].x, pv[j].y}; bs23 += bf3_f2{pv[j].z, pv[j].w}; }
            }
            if (owner) {
                __bf16* base = pl + (k & 1) * bufsz + (is_x ? 3 * O * P : 0) + 8 * g;
                const int cols = is_x ? I : O;
    #pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) {                                   // two columns at a time: 48 live pieces, not 96
                    bf3_u2 h[8], m[8], l[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        bf3_split2(e2 == 0 ? bf3_f2{pv[j].x, pv[j].y} : bf3_f2{pv[j].z, pv[j].w}, h[j], m[j], l[j]);
                    __bf16* col = base + (4 * cg + 2 * e2) * P;
                    *reinterpret_cast<uint4*>(col) = make_uint4(bf3_pack(h[0].x, h[1].x), bf3_pack(h[2].x, h[3].x), bf3_pack(h[4].x, h[5].x), bf3_pack(h[6].x, h[7].x));
                    *reinterpret_cast<uint4*>(col + P) = make_uint4(bf3_pack(h[0].y, h[1].y), bf3_pack(h[2].y, h[3].y), bf3_pack(h[4].y, h[5].y), bf3_pack(h[6].y, h[7].y));
                    *reinterpret_cast<uint4*>(col + cols * P) = make_uint4(bf3_pack(m[0].x, m[1].x), bf3_pack(m[2].x, m[3].x), bf3_pack(m[4].x, m[5].x), bf3_pack(m[6].x, m[7].x));
                    *reinterpret_cast<uint4*>(col + cols * P + P) = make_uint4(bf3_pack(m[0].y, m[1].y), bf3_pack(m[2].y, m[3].y), bf3_pack(m[4].y, m[5].y), bf3_pack(m[6].y, m[7].y));
                    *reinterpret_cast<uint4*>(col + 2 * cols * P) = make_uint4(bf3_pack(l[0].x, l[1].x), bf3_pack(l[2].x, l[3].x), bf3_pack(l[4].x, l[5].x), bf3_pack(l[6].x, l[7].x));
                    *reinterpret_cast<uint4*>(col + 2 * cols * P + P) = make_uint4(bf3_pack(l[0].y, l[1].y), bf3_pack(l[2].y, l[3].y), bf3_pack(l[4].y, l[5].y), bf3_pack(l[6].y, l[7].y));
                }
            }
        };
        __builtin_amdgcn_s_setprio(2);     // (the staging waves' vector work first: measured 3750 against 4100 cycles per chunk)
        if (mine > 0) {
            issue(0, pv0, pm0);
            issue(1, pv1, pm1);
            commit(0, pv0, pm0);
            issue(2, pv0, pm0);
        }
        __syncthreads();
        for (int64_t k = 0; k < mine; k += 2) {
            commit(k + 1, pv1, pm1);
            issue(k + 3, pv1, pm1);
            __syncthreads();
            commit(k + 2, pv0, pm0);
            issue(k + 4, pv0, pm0);
            __syncthreads();
        }
    };
    f32x16 acc[TI];
#pragma unroll
    for (int t = 0; t < TI; ++t)
        for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
    const int o = wave * 32 + c;
    const bool strip = !producer && wave * 32 < O;          // (wave-uniform: O <= 96 leaves the last strip without work)
    // (padded lanes read the last valid column: what they accumulate lands in rows / columns that are never stored)
    const int aoff = min(o, O - 1) * P + 8 * kg;
    int boff[TI];
#pragma unroll
    for (int t = 0; t < TI; ++t) boff[t] = 3 * O * P + min(t * 32 + c, I - 1) * P + 8 * kg;
    auto ld8 = [&](const __bf16* ptr) { return __builtin_bit_cast(bf3_x8, *reinterpret_cast<const uint4*>(ptr)); };
    auto multiply = [&](int64_t k) {
        const __bf16* bf = pl + (k & 1) * bufsz;
#pragma unroll
        for (int ks = 0; ks < R / 16; ++ks) {
            const bf3_x8 ah = ld8(bf + aoff + 16 * ks), am = ld8(bf + aoff + O * P + 16 * ks), al = ld8(bf + aoff + 2 * O * P + 16 * ks);
            bf3_x8 bh[TI], bm[TI], bl[TI];
#pragma unroll
            for (int t = 0; t < TI; ++t) {
                bh[t] = ld8(bf + boff[t] + 16 * ks); bm[t] = ld8(bf + boff[t] + I * P + 16 * ks); bl[t] = ld8(bf + boff[t] + 2 * I * P + 16 * ks);
            }
            // smallest terms first; consecutive instructions feed different accumulators
#pragma unroll
            for (int t = 0; t < TI; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[t], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < TI; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[t], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < TI; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm[t], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < TI; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh[t], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < TI; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm[t], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < TI; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[t], acc[t], 0, 0, 0);
        }
    };
    if (tr) __syncthreads();
    // Chunk k of this block is staged into buffer k & 1 during step k - 1 and multiplied during step k; the requests for k + 2
    // leave when k's registers are free (two named register sets, loop unrolled by two).  The two roles run their OWN loops with
    // the same number of barriers (1 + 2 ceil(mine / 2)): in one shared loop the accumulators stay live across the staging code and the
    // staging registers across the matrix code (140 spills).
    float* out = A.slab + (int64_t)slice * A.slab_row;
    if (producer) {
        if (MASK && !is_x) produce(std::true_type{});
        else produce(std::false_type{});
    } else {
        __syncthreads();
        for (int64_t k = 0; k < mine; k += 2) {
            if (strip) multiply(k);

            __syncthreads();

            if (strip && k + 1 < mine) multiply(k + 1);

            __syncthreads();
        }
        if (strip) {
#pragma unroll
            for (int t = 0; t < TI; ++t) {
                const int i = t * 32 + c;
                for (int v = 0; v < 16; ++v) {
                    const int orow = wave * 32 + (v & 3) + 8 * (v >> 2) + 4 * kg;
                    if (orow < O && i < I) out[q.out_off + (int64_t)orow * A.ldw + i] = acc[t][v];
                }
            }
        }
    }
    if (q.bias_off >= 0) {                            // (uniform) the four row groups' column sums, in row-group order
        if (owner && !is_x) *reinterpret_cast<float4*>(coef + g * O + 4 * cg) = make_float4(bs01.x, bs01.y, bs23.x, bs23.y);
        __syncthreads();
        if (tid < O) out[q.bias_off + tid] = (coef[tid] + coef[O + tid]) + (coef[2 * O + tid] + coef[3 * O + tid]);
    }
}

bool wgrad3_ok(const Wg2Args& A) {
    for (int i = 1; i < A.nprob; ++i)
        if ((A.q[i].mask != nullptr) != (A.q[0].mask != nullptr)) return false;      // (one instantiation per launch)
    for (int i = 0; i < A.nprob; ++i)
        if (A.q[i].dys >= (1 << 22) || A.q[i].xs >= (1 << 22)) return false;                   // (24-bit byte pitch)
    return A.O <= 128 && A.I <= 128;
}
size_t wgrad3_lds(const Wg2Args& A) {
    const int m = A.O > A.I ? A.O : A.I;
    return (size_t)2 * 2 * 3 * (A.O + A.I) * kW3Pitch + sizeof(float) * 4 * (size_t)m;
}

bool wgrad2_ok(const kpgnn_wgrad_desc* d, const float* x) {
    auto al = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
    return d->O % 4 == 0 && d->I % 4 == 0 && d->O <= 128 && d->I <= 256 && d->dy_stride % 4 == 0 && d->x_stride % 4 == 0 &&
           al(d->dy) && al(x) && (!d->dy_mask || al(d->dy_mask));
}

Wg2Prob wgrad2_prob(const kpgnn_wgrad_desc* d, const float* x, int64_t out_off, int64_t bias_off) {
    Wg2Prob q;
    q.dy = d->dy; q.x = x; q.mask = d->dy_mask; q.xm = d->x_mean; q.xi = d->x_invstd; q.xg = d->x_gamma; q.xb = d->x_beta;
    q.xrelu = d->x_relu; q.dys = d->dy_stride; q.xs = d->x_stride; q.out_off = out_off; q.bias_off = bias_off;
    return q;
}

// slices (= slab rows) per problem: the problems share the chip's 2 x 256 block slots; >= 1 chunk of rows per slice
int wgrad2_slices(int64_t N, int nprob, int blocks) {
    const int64_t chunks = (N + kW2Rows - 1) / kW2Rows;
    int64_t s = blocks / nprob;
    if (s > chunks) s = chunks;
    return (int)(s < 1 ? 1 : s);
}

// Launches the problems of A; *nslices = slab rows written (the caller's reduction covers exactly those).
int wgrad2_launch(Wg2Args& A, int math, int* nslices, hipStream_t s) {
    const int ti = (A.I + 31) / 32;
    if (math != KPGNN_MATH_F32 && wgrad3_ok(A)) {
        *nslices = wgrad2_slices(A.N, A.nprob, kW3Blocks);
        dim3 gr3((unsigned)(*nslices * A.nprob)), blk3(512);
        const size_t lds3 = wgrad3_lds(A);
        const bool mask = A.q[0].mask != nullptr;      // (the problems of a launch share dy and its mask, or have none)
#define KP_W3(T) do { if (mask) { KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)wgrad3_kernel<T, true>, lds3)); \
                                  hipLaunchKernelGGL((wgrad3_kernel<T, true>), gr3, blk3, lds3, s, A); } \
                      else { KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)wgrad3_kernel<T, false>, lds3)); \
                             hipLaunchKernelGGL((wgrad3_kernel<T, false>), gr3, blk3, lds3, s, A); } } while (0)
        switch (ti) {
            case 1: KP_W3(1); break;
            case 2: KP_W3(2); break;
            case 3: KP_W3(3); break;
            default: KP_W3(4); break;
        }
#undef KP_W3
        KPGNN_LAUNCH_CHECK("wgrad3_kernel");
        return KPGNN_OK;
    }
    *nslices = wgrad2_slices(A.N, A.nprob, kWgradBlocks);
    dim3 gr((unsigned)(*nslices * A.nprob)), blk(256);
    const size_t lds = sizeof(float) * ((size_t)kW2Rows * (A.O + A.I) + 4 * (size_t)A.I);
#define KP_W2(T) hipLaunchKernelGGL((wgrad2_kernel<T>), gr, blk, lds, s, A)
    switch (ti) {
        case 1: KP_W2(1); break;
        case 2: KP_W2(2); break;
        case 3: KP_W2(3); break;
        case 4: KP_W2(4); break;
        case 5: KP_W2(5); break;
        case 6: KP_W2(6); break;
        case 7: KP_W2(7); break;
        default: KP_W2(8); break;
    }
#undef KP_W2
    KPGNN_LAUNCH_CHECK("wgrad2_kernel");
    return KPGNN_OK;
}

}  // namespace
}  // namespace kpgnn

using namespace kpgnn;

extern "C" size_t kpgnn_wgrad_workspace_bytes(int32_t O, int32_t I) {
    if (O < 1 || I < 1) return 0;
    return sizeof(float) * ((size_t)kWgradBlocks * ((size_t)O * I + O) + O);  // slabs + a sink for an unwanted db
}

extern "C" int kpgnn_linear_wgrad(const kpgnn_wgrad_desc* d, kpgnn_stream_t stream) {
    int rc = wgrad_check(d, "linear_wgrad");
    if (rc != KPGNN_OK) return rc;
    KPGNN_REQUIRE(d->workspace && d->workspace_bytes >= kpgnn_wgrad_workspace_bytes(d->O, d->I), "linear_wgrad: workspace too small");
    const int64_t nw = (int64_t)d->O * d->I;
    float* slab = (float*)d->workspace;
    hipStream_t s = (hipStream_t)stream;
    int grid;
    if (wgrad2_ok(d, d->x)) {
        Wg2Args A;
        A.N = d->N; A.n_dyn = d->n_dyn; A.O = d->O; A.I = d->I; A.nprob = 1; A.ldw = d->I; A.slab = slab; A.slab_row = nw + d->O;
        for (int i = 0; i < kW2MaxProb; ++i) A.q[i] = wgrad2_prob(d, d->x, 0, nw);
        rc = wgrad2_launch(A, d->math, &grid, s);
    } else {
        KPGNN_REQUIRE(!d->dy_mask && !d->n_dyn, "linear_wgrad: dy_mask / n_dyn need 16-B aligned operands with O %% 4 == I %% 4 == 0, O <= 128");
        WgPair pp;
        pp.q[0] = pp.q[1] = wgrad_params(d, slab, nw + d->O, 0);
        grid = wgrad_blocks(d->N, d->O);
        rc = wgrad_launch(pp, 1, d->O, d->I, grid, s);
    }
    if (rc != KPGNN_OK) return rc;
    float* db = d->db ? d->db : slab + (size_t)kWgradBlocks * (nw + d->O);  // sink behind the slabs
    if (d->defer) {
        kpgnn_reduce_job* j = d->defer;
        j->slab = slab; j->nslab = grid; j->elems = nw + d->O;
        j->out[0] = d->dw; j->n_out[0] = nw; j->out[1] = db; j->n_out[1] = d->O;
        j->out[2] = j->out[3] = nullptr; j->n_out[2] = j->n_out[3] = 0;
        return KPGNN_OK;
    }
    return slab_reduce(slab, grid, nw + d->O, d->dw, nw, db, d->O, nullptr, s);
}

extern "C" int kpgnn_linear_wgrad_pair(const kpgnn_wgrad_desc* a, const kpgnn_wgrad_desc* b, kpgnn_stream_t stream) {
    int rc = wgrad_check(a, "linear_wgrad_pair");
    if (rc != KPGNN_OK) return rc;
    rc = wgrad_check(b, "linear_wgrad_pair");
    if (rc != KPGNN_OK) return rc;
    KPGNN_REQUIRE(a->O == b->O && a->I == b->I && a->N == b->N, "linear_wgrad_pair: the two problems must share N, O, I");
    KPGNN_REQUIRE(a->db && b->db, "linear_wgrad_pair: both bias gradients are required");
    KPGNN_REQUIRE(a->workspace && a->workspace_bytes >= 2 * kpgnn_wgrad_workspace_bytes(a->O, a->I), "linear_wgrad_pair: workspace too small");
    const int64_t nw = (int64_t)a->O * a->I, one = nw + a->O;
    float* slab = (float*)a->workspace;
    hipStream_t s = (hipStream_t)stream;
    int grid;
    if (wgrad2_ok(a, a->x) && wgrad2_ok(b, b->x)) {
        Wg2Args A;
        A.N = a->N; A.n_dyn = a->n_dyn; A.O = a->O; A.I = a->I; A.nprob = 2; A.ldw = a->I; A.slab = slab; A.slab_row = 2 * one;
        for (int i = 0; i < kW2MaxProb; ++i) A.q[i] = wgrad2_prob(a, a->x, 0, nw);
        A.q[1] = wgrad2_prob(b, b->x, one, one + nw);
        rc = wgrad2_launch(A, a->math, &grid, s);
    } else {
        KPGNN_REQUIRE(!a->dy_mask && !b->dy_mask && !a->n_dyn, "linear_wgrad_pair: dy_mask / n_dyn need 16-B aligned operands");
        WgPair pp;
        pp.q[0] = wgrad_params(a, slab, 2 * one, 0);
        pp.q[1] = wgrad_params(b, slab, 2 * one, one);
        grid = wgrad_blocks(a->N, a->O);
        rc = wgrad_launch(pp, 2, a->O, a->I, grid, s);
    }
    if (rc != KPGNN_OK) return rc;
    if (a->defer) {
        kpgnn_reduce_job* j = a->defer;
        j->slab = slab; j->nslab = grid; j->elems = 2 * one;
        j->out[0] = a->dw; j->n_out[0] = nw; j->out[1] = a->db; j->n_out[1] = a->O;
        j->out[2] = b->dw; j->n_out[2] = nw; j->out[3] = b->db; j->n_out[3] = b->O;
        return KPGNN_OK;
    }
    return slab_reduce(slab, grid, 2 * one, a->dw, nw, a->db, a->O, b->dw, s, nw, b->db);
}

extern "C" size_t kpgnn_wgrad_group_workspace_bytes(int32_t O, int32_t I, int32_t group) {
    if (O < 1 || I < 1 || group < 1 || group > kW2MaxProb) return 0;
    return sizeof(float) * ((size_t)(kWgradBlocks / group) * ((size_t)O * I * group + O) + O);
}

extern "C" int kpgnn_linear_wgrad_group(const kpgnn_wgrad_desc* d, const float* const* x_group, int32_t group, kpgnn_stream_t stream) {
    KPGNN_REQUIRE(d != nullptr && x_group != nullptr && group >= 1 && group <= kW2MaxProb, "linear_wgrad_group: bad arguments (1 <= group <= %d)", kW2MaxProb);
    kpgnn_wgrad_desc probe = *d;
    probe.x = x_group[0];
    int rc = wgrad_check(&probe, "linear_wgrad_group");
    if (rc != KPGNN_OK) return rc;
    for (int l = 0; l < group; ++l)
        if (!x_group[l] || !wgrad2_ok(d, x_group[l]))
            return fail(KPGNN_ELIMIT, "linear_wgrad_group: needs 16-B aligned operands, O %% 4 == I %% 4 == 0, O <= 128, I <= 256");
    KPGNN_REQUIRE(!d->x_mean, "linear_wgrad_group: no x transform");
    KPGNN_REQUIRE(d->workspace && d->workspace_bytes >= kpgnn_wgrad_group_workspace_bytes(d->O, d->I, group), "linear_wgrad_group: workspace too small");
    const int64_t nw = (int64_t)d->O * d->I * group;
    Wg2Args A;
    A.N = d->N; A.n_dyn = d->n_dyn; A.O = d->O; A.I = d->I; A.nprob = group; A.ldw = (int64_t)d->I * group;
    A.slab = (float*)d->workspace; A.slab_row = nw + d->O;
    for (int i = 0; i < kW2MaxProb; ++i) A.q[i] = wgrad2_prob(d, x_group[i < group ? i : 0], (int64_t)(i < group ? i : 0) * d->I, i == 0 ? nw : -1);
    int grid = 0;
    hipStream_t s = (hipStream_t)stream;
    rc = wgrad2_launch(A, d->math, &grid, s);
    if (rc != KPGNN_OK) return rc;
    float* db = d->db ? d->db : A.slab + (size_t)grid * A.slab_row;     // sink behind the slab rows
    if (d->defer) {
        kpgnn_reduce_job* j = d->defer;
        j->slab = A.slab; j->nslab = grid; j->elems = nw + d->O;
        j->out[0] = d->dw; j->n_out[0] = nw; j->out[1] = db; j->n_out[1] = d->O;
        j->out[2] = j->out[3] = nullptr; j->n_out[2] = j->n_out[3] = 0;
        return KPGNN_OK;
    }
    return slab_reduce(A.slab, grid, nw + d->O, d->dw, nw, db, d->O, nullptr, s);
}

// ------------------------------------------------------------------------------------------------ y = x W^T + b
namespace kpgnn {
namespace {

struct LinParams {
    int64_t N; int O, I, pitch, ypitch, wt;
    const float* x; int64_t xs;
    const float* xmask; const int32_t* n_dyn;   // optional ReLU mask of x (same layout), optional live-row count
    const float* w; const float* bias;
    float* y; int64_t ys;
    int yb; int64_t ybs;   // wide kernel: output column o lands in block o / yb at column o % yb; blocks are ybs floats apart (yb == O: plain rows)
};

// Wide outputs (O > 128, e.g. the input gradient of the jumping-knowledge projection: [N,104] x [104,936]): the x tile
// stays resident in LDS while the block walks the outputs 128 at a time - per chunk every wave reloads its strip of the
// weight (L2-resident) and runs its MFMA chains; results go straight from the accumulators to y (64 x 16-B segments per
// store: measured as fast as staging through LDS), so no barrier separates the chunks.
template <int KS, int M>
__global__ void __launch_bounds__(256, 2)
linear_wide_kernel(const LinParams p) {
    extern __shared__ __attribute__((aligned(16))) float xl[];      // [32*M][pitch]
    constexpr int ROWS = 32 * M;
    constexpr int IC = 2 * KS;                        // == I (host)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int kk = lane >> 5, c = lane & 31;
    const int O = p.O, pitch = p.pitch;
    const int64_t N = p.n_dyn ? (int64_t)min((int64_t)*p.n_dyn, p.N) : p.N;
    const int64_t tiles = (N + ROWS - 1) / ROWS;
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        __syncthreads();                               // previous tile fully consumed
        {   // x tile -> LDS (no register double-buffer: the eight output chunks dwarf this load)
            const int64_t r0 = tile * ROWS;
            const int lim = (int)(N - r0 < ROWS ? N - r0 : ROWS) * IC;
            const float* base = p.x + r0 * p.xs;
            for (int e = 4 * tid; e < ROWS * IC; e += 4 * 256) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < lim) {
                    v = *reinterpret_cast<const float4*>(base + e);
                    if (p.xmask) {
                        const float4 mk = *reinterpret_cast<const float4*>(p.xmask + r0 * p.xs + e);
                        if (mk.x <= 0.f) v.x = 0.f; if (mk.y <= 0.f) v.y = 0.f; if (mk.z <= 0.f) v.z = 0.f; if (mk.w <= 0.f) v.w = 0.f;
                    }
                }
                *reinterpret_cast<float4*>(xl + (e / IC) * pitch + (e % IC)) = v;
            }
        }
        __syncthreads();
        const int64_t r0 = tile * ROWS;
        const float* b0 = xl + c * pitch + kk;
#pragma unroll 1
        for (int chunk = (int)blockIdx.y * 128; chunk < O; chunk += 128 * (int)gridDim.y) {   // (few row tiles: chunks over blockIdx.y)
            // the operand reads below do not depend on the chunk: without this the compiler hoists all of them out of
            // the loop and spills ~500 registers
            int z = 0;
            asm volatile("" : "+v"(z));
            const float* bz = b0 + z;
            const int o = chunk + wave * 32 + c;
            float a[KS];
            if (p.wt) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) a[ks] = o < O ? p.w[(int64_t)(2 * ks + kk) * O + o] : 0.f;
            } else {
#pragma unroll
                for (int j = 0; j < KS / 2; ++j) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (o < O) v = *reinterpret_cast<const float4*>(p.w + (int64_t)o * IC + 4 * j);
                    a[2 * j] = kk ? v.y : v.x;
                    a[2 * j + 1] = kk ? v.w : v.z;
                }
            }
            f32x16 acc[M];
#pragma unroll
            for (int m = 0; m < M; ++m)
                for (int v = 0; v < 16; ++v) acc[m][v] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    const float xv = bz[m * 32 * pitch + 2 * ks];
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks], xv, acc[m], 0, 0, 0);
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ob = chunk + wave * 32 + 8 * g + 4 * kk;
                if (ob < O) {
                    float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (p.bias) bb = *reinterpret_cast<const float4*>(p.bias + ob);
#pragma unroll
                    for (int m = 0; m < M; ++m) {
                        const int64_t r = r0 + m * 32 + c;
                        if (r < N)
                            *reinterpret_cast<float4*>(p.y + (int64_t)(ob / p.yb) * p.ybs + r * p.ys + (ob % p.yb)) =
                                make_float4(acc[m][4 * g] + bb.x, acc[m][4 * g + 1] + bb.y, acc[m][4 * g + 2] + bb.z, acc[m][4 * g + 3] + bb.w);
                    }
                }
            }
        }
    }
}

}  // namespace
}  // namespace kpgnn

extern "C" int kpgnn_linear_fwd(const kpgnn_linear_desc* d, kpgnn_stream_t stream) {
    KPGNN_REQUIRE(d != nullptr, "linear_fwd: NULL descriptor");
    KPGNN_REQUIRE(d->N >= 1 && d->O >= 1 && d->I >= 1, "linear_fwd: bad N=%lld O=%d I=%d", (long long)d->N, d->O, d->I);
    if (d->O > 4096 || d->I > 128) return fail(KPGNN_ELIMIT, "linear_fwd: O=%d exceeds 4096 or I=%d exceeds 128", d->O, d->I);
    KPGNN_REQUIRE(d->x && d->w && d->y, "linear_fwd: NULL pointer");
    const bool blocked = d->y_block_cols > 0 && d->y_block_cols < d->O;   // output split into column blocks ([S, N, yb] layout)
    if (blocked && ((d->y_block_cols % 4) != 0 || (d->O % d->y_block_cols) != 0 || d->y_stride != d->y_block_cols ||
                    (d->y_block_stride % 4) != 0 || d->O <= 128))
        return fail(KPGNN_ELIMIT, "linear_fwd: blocked output needs O > 128, O %% y_block_cols == 0, y_block_cols %% 4 == 0, "
                                  "y_stride == y_block_cols and a 16-B aligned block stride");
    if ((d->O % 4) != 0 || (d->I % 4) != 0 || d->x_stride != d->I || (!blocked && d->y_stride != d->O) ||
        (((uintptr_t)d->x | (uintptr_t)d->y) & 15) != 0 || (d->bias && (((uintptr_t)d->bias) & 15) != 0))
        return fail(KPGNN_ELIMIT, "linear_fwd: needs contiguous 16-B aligned x / y with I %% 4 == 0 and O %% 4 == 0");
    hipStream_t s = (hipStream_t)stream;
    if (d->O <= 128 && d->x_mask) return fail(KPGNN_ELIMIT, "linear_fwd: x_mask is implemented for O > 128");
    if (d->x_mask && (((uintptr_t)d->x_mask) & 15) != 0) return fail(KPGNN_ELIMIT, "linear_fwd: x_mask must be 16-B aligned");
    if (d->O <= 128) {                                  // the plain variant of the fused kernel (lin_fused.h)
        kpgnn_linear_bn_desc f = {};
        f.N = d->N; f.n_dyn = d->n_dyn; f.O = d->O; f.I = d->I; f.x = d->x; f.w = d->w; f.bias = d->bias; f.y = d->y; f.w_transposed = d->w_transposed;
        return kpgnn_linear_bn(&f, stream);
    }
    if (blocked && (((uintptr_t)d->w) & 15) == 0) {
        bool handled = false;
        const int rc = linear3_blocked(d, s, &handled);                            // the bf16-split kernel, where it applies
        if (handled || rc != KPGNN_OK) return rc;
    }
    if (d->I != 32 && d->I != 64 && d->I != 104 && d->I != 128)
        return fail(KPGNN_ELIMIT, "linear_fwd: wide outputs need I in {32, 64, 104, 128} (the k-loop is fully unrolled)");
    LinParams p;
    p.N = d->N; p.O = d->O; p.I = d->I; p.wt = d->w_transposed ? 1 : 0;
    const int rowp = d->I + ((4 - d->I % 8) + 8) % 8;   // pitch = 4 (mod 8) floats: 16-B aligned rows, conflict-free 16-B accesses
    p.pitch = rowp; p.ypitch = rowp;
    p.x = d->x; p.xs = d->x_stride; p.w = d->w; p.bias = d->bias; p.y = d->y; p.ys = d->y_stride;
    p.xmask = d->x_mask; p.n_dyn = d->n_dyn;
    p.yb = blocked ? d->y_block_cols : d->O; p.ybs = blocked ? d->y_block_stride : 0;
    // rows per tile = 32 * m, m in 1..3, the smallest that makes the launch one round over two blocks per CU
    const int64_t slots = (int64_t)device_facts().cu_count * 2;
    int m = (int)((d->N + slots * 32 - 1) / (slots * 32));
    m = m < 1 ? 1 : (m > 3 ? 3 : m);
    const int rows = 32 * m;
    const size_t lds = sizeof(float) * (size_t)rows * rowp;
    const int64_t tiles = (d->N + rows - 1) / rows;
    const int64_t grid = (m == 1 ? slots * 2 : slots) < tiles ? (m == 1 ? slots * 2 : slots) : tiles;
    dim3 blk(256);
    const int ks = (d->I + 1) / 2;
    // a small batch has fewer row tiles than the chip has block slots: the output chunks of a tile are then spread over
    // blockIdx.y instead of walked one after the other (batch 64, [1.5k,104] x [104,936]: one block chain of 8 chunks, 31 us)
    const int64_t nchunks = (d->O + 127) / 128;
    int64_t gy = tiles < slots ? (slots + tiles - 1) / tiles : 1;
    if (gy > nchunks) gy = nchunks;
#define KP_LIN2(KSV, MV) do { \
        KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)linear_wide_kernel<KSV, MV>, lds)); \
        hipLaunchKernelGGL((linear_wide_kernel<KSV, MV>), dim3((unsigned)grid, (unsigned)gy), blk, lds, s, p); } while (0)
#define KP_LIN(KSV) do { if (m == 1) KP_LIN2(KSV, 1); else if (m == 2) KP_LIN2(KSV, 2); else KP_LIN2(KSV, 3); } while (0)
    if (ks == 16) KP_LIN(16);
    else if (ks == 32) KP_LIN(32);
    else if (ks == 52) KP_LIN(52);
    else KP_LIN(64);
#undef KP_LIN
#undef KP_LIN2
    KPGNN_LAUNCH_CHECK("linear_wide_kernel");
    return KPGNN_OK;
}
