// KP-GIN per-hop MLP + geometric hop-combine, forward and backward, on the fp32 matrix cores (gfx950).
// Contract: include/kpgnn.h, kpgnn_hop_mlp_fwd / _bwd  (reference layers/KPGIN.py:106-112, combine.py:52-58).
//
// Shapes: [N, K, dk] activations with dk = hidden / K (13, 20, 6 ...) and K tiny [dk, dk] weights - HBM-bound work
// that a BLAS tile cannot feed.  One 256-thread block keeps a 64-node tile resident in LDS (row pitch LD = 4 mod 8:
// the 16-row x 4-column operand reads of v_mfma_f32_16x16x4_f32 then touch 64 distinct banks) next to the
// zero-padded hop weights, and does everything that needs the tile before it leaves:
//   fwd:  s -> h1 -> h2 (-> out = sum_k theta_k * h2_k); each activation is written to HBM once, coalesced, from LDS.
//   bwd:  g2 = gout (x) theta * [h2 > 0]; dW2 += h1^T g2; gh1 = g2 W2^T * [h1 > 0]; dW1 += s^T gh1; gs = gh1 W1^T;
//         db1, db2, dtheta ride along as per-thread column sums.
// MFMA operand map (16x16x4, exact fp32): lane l feeds A[m = l & 15][k = l >> 4] and B[k = l >> 4][n = l & 15];
// accumulator register r of lane l is C[m = 4 * (l >> 4) + r][n = l & 15].
// Weight-gradient tiles stay in accumulators for the whole launch; per-block partials go to a slab that is added
// in block order (deterministic).
#include "kpgnn_common.h"

namespace kpgnn {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kTN = 64;        // nodes per tile (4 row groups of 16)
constexpr int kHmThreads = 256;
constexpr int kHmWaves = 4;
constexpr int kMaxItems = 8;   // weight-gradient MFMA tiles a wave may own per layer
constexpr int kMaxCols = 4;    // activation columns a thread may own in the column passes (K * D <= 1024)
constexpr int kHmMaxGrid = 1024;

struct HmParams {
    int64_t N, tiles;
    int K, DI, DO, LD;
    int vec_i, vec_o;          // 16-B staging allowed for the [.., K*DI] / [.., K*DO] tensors
    const float *s, *w1, *b1, *w2, *b2, *theta;
    float *h1, *h2, *out;
    const float* gout;
    float* gs;
    float* slab;               // [gridDim.x][slab_w]
    int64_t slab_w;
};

// global [rows, ncol] (contiguous) -> LDS [kTN][LD]; rows beyond `rows` are zero filled.
__device__ __forceinline__ void stage_in(float* buf, const float* __restrict__ src, int rows, int ncol, int LD, int vec) {
    if (vec) {
        const int n4 = (kTN * ncol) >> 2;
        for (int e4 = threadIdx.x; e4 < n4; e4 += kHmThreads) {
            const int e = e4 << 2;
            const int node = e / ncol, off = e - node * ncol;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (node < rows) v = *reinterpret_cast<const float4*>(src + e);
            *reinterpret_cast<float4*>(buf + node * LD + off) = v;
        }
    } else {
        const int n = kTN * ncol;
        for (int e = threadIdx.x; e < n; e += kHmThreads) {
            const int node = e / ncol, off = e - node * ncol;
            buf[node * LD + off] = node < rows ? src[e] : 0.f;
        }
    }
}

// LDS [kTN][LD] -> global [rows, ncol]
__device__ __forceinline__ void stage_out(float* __restrict__ dst, const float* buf, int rows, int ncol, int LD, int vec) {
    if (vec) {
        const int n4 = (rows * ncol) >> 2;
        for (int e4 = threadIdx.x; e4 < n4; e4 += kHmThreads) {
            const int e = e4 << 2;
            const int node = e / ncol, off = e - node * ncol;
            *reinterpret_cast<float4*>(dst + e) = *reinterpret_cast<const float4*>(buf + node * LD + off);
        }
    } else {
        const int n = rows * ncol;
        for (int e = threadIdx.x; e < n; e += kHmThreads) {
            const int node = e / ncol, off = e - node * ncol;
            dst[e] = buf[node * LD + off];
        }
    }
}

// acc[t] (+)= A * B for one (hop, row group): A = 16 rows of `bin` (columns a0 .. a0+din), B = wl [din (pad P)][P].
template <int T>
__device__ __forceinline__ void rows_times_weights(f32x4 (&acc)[T], const float* bin, int a0, int din, const float* wl,
                                                   int rg, int LD, int lr, int lq) {
    constexpr int P = 16 * T;
    const float* arow = bin + (rg * 16 + lr) * LD + a0;
    const int qn = (din + 3) >> 2;
#pragma unroll 2
    for (int q = 0; q < qn; ++q) {
        const int kk = 4 * q + lq;
        float a = arow[kk < din ? kk : din - 1];
        a = kk < din ? a : 0.f;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const float b = wl[kk * P + t * 16 + lr];
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
        }
    }
}

__device__ __forceinline__ float relu_keep_nan(float v) { return v < 0.f ? 0.f : v; }

// ---------------------------------------------------------------------------------------------------------- forward
template <int T>
__global__ void __launch_bounds__(kHmThreads)
hop_mlp_fwd_kernel(const HmParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int P = 16 * T, PP = P * P;
    const int K = p.K, DI = p.DI, DO = p.DO, LD = p.LD;
    float* wl1 = lds;                 // [K][P][P]  wl1[k][i][j] = W1[k][i][j]
    float* wl2 = wl1 + K * PP;        // [K][P][P]
    float* bl1 = wl2 + K * PP;        // [K][P]
    float* bl2 = bl1 + K * P;
    float* th = bl2 + K * P;          // [K][P]
    float* bufA = th + K * P;         // [kTN][LD]
    float* bufB = bufA + kTN * LD;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;

    for (int idx = tid; idx < K * PP; idx += kHmThreads) {
        const int k = idx / PP, r = idx - k * PP, i = r / P, j = r - i * P;
        wl1[idx] = (i < DI && j < DO) ? p.w1[((int64_t)k * DI + i) * DO + j] : 0.f;
        wl2[idx] = (i < DO && j < DO) ? p.w2[((int64_t)k * DO + i) * DO + j] : 0.f;
    }
    for (int idx = tid; idx < K * P; idx += kHmThreads) {
        const int k = idx / P, j = idx - k * P;
        bl1[idx] = j < DO ? p.b1[k * DO + j] : 0.f;
        bl2[idx] = j < DO ? p.b2[k * DO + j] : 0.f;
        th[idx] = (p.theta && j < DO) ? p.theta[k * DO + j] : 0.f;
    }
    const int ncol_i = K * DI, ncol_o = K * DO;
    for (int64_t tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
        const int64_t n0 = tile * kTN;
        const int rows = (int)((p.N - n0) < kTN ? (p.N - n0) : kTN);
        __syncthreads();                                        // previous tile's readers; first pass: the weight fill
        stage_in(bufA, p.s + n0 * ncol_i, rows, ncol_i, LD, p.vec_i);
        __syncthreads();
        for (int t = wave; t < K * 4; t += kHmWaves) {          // layer 1: bufA (s) -> bufB (h1)
            const int k = t >> 2, rg = t & 3;
            f32x4 acc[T];
#pragma unroll
            for (int jt = 0; jt < T; ++jt) { const float b = bl1[k * P + jt * 16 + lr]; acc[jt] = {b, b, b, b}; }
            rows_times_weights<T>(acc, bufA, k * DI, DI, wl1 + k * PP, rg, LD, lr, lq);
#pragma unroll
            for (int jt = 0; jt < T; ++jt) {
                const int col = jt * 16 + lr;
                if (col < DO)
#pragma unroll
                    for (int r = 0; r < 4; ++r) bufB[(rg * 16 + 4 * lq + r) * LD + k * DO + col] = relu_keep_nan(acc[jt][r]);
            }
        }
        __syncthreads();
        stage_out(p.h1 + n0 * ncol_o, bufB, rows, ncol_o, LD, p.vec_o);
        for (int t = wave; t < K * 4; t += kHmWaves) {          // layer 2: bufB (h1) -> bufA (h2)
            const int k = t >> 2, rg = t & 3;
            f32x4 acc[T];
#pragma unroll
            for (int jt = 0; jt < T; ++jt) { const float b = bl2[k * P + jt * 16 + lr]; acc[jt] = {b, b, b, b}; }
            rows_times_weights<T>(acc, bufB, k * DO, DO, wl2 + k * PP, rg, LD, lr, lq);
#pragma unroll
            for (int jt = 0; jt < T; ++jt) {
                const int col = jt * 16 + lr;
                if (col < DO)
#pragma unroll
                    for (int r = 0; r < 4; ++r) bufA[(rg * 16 + 4 * lq + r) * LD + k * DO + col] = relu_keep_nan(acc[jt][r]);
            }
        }
        __syncthreads();
        stage_out(p.h2 + n0 * ncol_o, bufA, rows, ncol_o, LD, p.vec_o);
        if (p.theta) {                                          // out[n, j] = sum_k theta[k, j] h2[n, k, j]
            for (int c = tid; c < rows * DO; c += kHmThreads) {
                const int node = c / DO, j = c - node * DO;
                const float* hrow = bufA + node * LD + j;
                float acc = 0.f;
                for (int k = 0; k < K; ++k) acc = fmaf(th[k * P + j], hrow[k * DO], acc);
                p.out[(n0 + node) * DO + j] = acc;
            }
        }
    }
}

// --------------------------------------------------------------------------------------------------------- backward
template <int T>
__global__ void __launch_bounds__(kHmThreads, 2)
hop_mlp_bwd_kernel(const HmParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int P = 16 * T, PP = P * P, TT = T * T;
    const int K = p.K, DI = p.DI, DO = p.DO, LD = p.LD;
    float* wt2 = lds;                 // [K][P][P]  wt2[k][j][i] = W2[k][i][j]
    float* wt1 = wt2 + K * PP;        // [K][P][P]  wt1[k][j][i] = W1[k][i][j]
    float* th = wt1 + K * PP;         // [K][P]
    float* bufG = th + K * P;         // [kTN][LD]  g2, then gh1
    float* bufH = bufG + kTN * LD;    // [kTN][LD]  h1, then s, then gs
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;

    for (int idx = tid; idx < K * PP; idx += kHmThreads) {
        const int k = idx / PP, r = idx - k * PP, j = r / P, i = r - j * P;
        wt2[idx] = (i < DO && j < DO) ? p.w2[((int64_t)k * DO + i) * DO + j] : 0.f;
        wt1[idx] = (i < DI && j < DO) ? p.w1[((int64_t)k * DI + i) * DO + j] : 0.f;
    }
    for (int idx = tid; idx < K * P; idx += kHmThreads) {
        const int k = idx / P, j = idx - k * P;
        th[idx] = (p.theta && j < DO) ? p.theta[k * DO + j] : 0.f;
    }
    const int ncol_i = K * DI, ncol_o = K * DO;
    // column passes: R row lanes per column when the row is narrower than the block, else several columns per thread
    int R = kHmThreads / ncol_o;
    R = R < 1 ? 1 : (R > kTN ? kTN : R);
    const int rl = ncol_o < kHmThreads ? tid / ncol_o : 0;
    const int c0 = ncol_o < kHmThreads ? tid - rl * ncol_o : tid;
    const bool col_on = ncol_o < kHmThreads ? rl < R : true;
    float gb1[kMaxCols], gb2[kMaxCols], gth[kMaxCols];
#pragma unroll
    for (int m = 0; m < kMaxCols; ++m) gb1[m] = gb2[m] = gth[m] = 0.f;
    const int nitems = K * TT;
    f32x4 acc1[kMaxItems], acc2[kMaxItems];
#pragma unroll
    for (int m = 0; m < kMaxItems; ++m) { acc1[m] = {0.f, 0.f, 0.f, 0.f}; acc2[m] = {0.f, 0.f, 0.f, 0.f}; }

    for (int64_t tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
        const int64_t n0 = tile * kTN;
        const int rows = (int)((p.N - n0) < kTN ? (p.N - n0) : kTN);
        // the (hop, tile) coordinates and operand addresses of the up to 2 x kMaxItems weight-gradient tiles are
        // tile-loop invariant; recomputing them (a few SALU ops) beats keeping ~100 hoisted VGPRs alive
        int wv = __builtin_amdgcn_readfirstlane(wave);
        asm volatile("" : "+s"(wv));
        __syncthreads();
        // A. g2 = gout (x theta) * [h2 > 0] -> bufG, h1 -> bufH; db2 and dtheta column sums
        if (col_on) {
#pragma unroll
            for (int m = 0; m < kMaxCols; ++m) {
                const int c = c0 + m * kHmThreads;
                if (c < ncol_o) {
                    const int k = c / DO, j = c - k * DO;
                    const float thv = th[k * P + j];
#pragma unroll 4
                    for (int n = rl; n < kTN; n += R) {
                        float g2 = 0.f, h1v = 0.f;
                        if (n < rows) {
                            const int64_t e = (n0 + n) * ncol_o + c;
                            const float h2v = p.h2[e];
                            h1v = p.h1[e];
                            float gv;
                            if (p.theta) {
                                const float go = p.gout[(n0 + n) * DO + j];
                                gth[m] = fmaf(go, h2v, gth[m]);
                                gv = go * thv;
                            } else {
                                gv = p.gout[e];
                            }
                            g2 = h2v > 0.f ? gv : 0.f;
                            gb2[m] += g2;
                        }
                        bufG[n * LD + c] = g2;
                        bufH[n * LD + c] = h1v;
                    }
                }
            }
        }
        __syncthreads();
        // B. dW2[k] += h1_k^T g2_k
#pragma unroll
        for (int m = 0; m < kMaxItems; ++m) {
            const int it = wv + m * kHmWaves;
            if (it < nitems) {
                const int k = it / TT, tt = it - k * TT, ti = tt / T, tj = tt - ti * T;
                const int ca = ti * 16 + lr, cb = tj * 16 + lr;
                const float* pa = bufH + k * DO + (ca < DO ? ca : DO - 1) + lq * LD;
                const float* pb = bufG + k * DO + (cb < DO ? cb : DO - 1) + lq * LD;
#pragma unroll 4
                for (int q = 0; q < kTN / 4; ++q) {
                    float a = pa[4 * q * LD], b = pb[4 * q * LD];
                    a = ca < DO ? a : 0.f;
                    b = cb < DO ? b : 0.f;
                    acc2[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc2[m], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        // C. gh1 = (g2 W2^T) * [h1 > 0], in place over g2
        for (int t = wave; t < K * 4; t += kHmWaves) {
            const int k = t >> 2, rg = t & 3;
            f32x4 acc[T];
#pragma unroll
            for (int ti = 0; ti < T; ++ti) acc[ti] = {0.f, 0.f, 0.f, 0.f};
            rows_times_weights<T>(acc, bufG, k * DO, DO, wt2 + k * PP, rg, LD, lr, lq);
#pragma unroll
            for (int ti = 0; ti < T; ++ti) {
                const int col = ti * 16 + lr;
                if (col < DO)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int idx = (rg * 16 + 4 * lq + r) * LD + k * DO + col;
                        bufG[idx] = bufH[idx] > 0.f ? acc[ti][r] : 0.f;
                    }
            }
        }
        __syncthreads();
        // D. s -> bufH (h1 is no longer needed)
        stage_in(bufH, p.s + n0 * ncol_i, rows, ncol_i, LD, p.vec_i);
        __syncthreads();
        // E. dW1[k] += s_k^T gh1_k ; db1 column sums
#pragma unroll
        for (int m = 0; m < kMaxItems; ++m) {
            const int it = wv + m * kHmWaves;
            if (it < nitems) {
                const int k = it / TT, tt = it - k * TT, ti = tt / T, tj = tt - ti * T;
                const int ca = ti * 16 + lr, cb = tj * 16 + lr;
                const float* pa = bufH + k * DI + (ca < DI ? ca : DI - 1) + lq * LD;
                const float* pb = bufG + k * DO + (cb < DO ? cb : DO - 1) + lq * LD;
#pragma unroll 4
                for (int q = 0; q < kTN / 4; ++q) {
                    float a = pa[4 * q * LD], b = pb[4 * q * LD];
                    a = ca < DI ? a : 0.f;
                    b = cb < DO ? b : 0.f;
                    acc1[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc1[m], 0, 0, 0);
                }
            }
        }
        if (col_on) {
#pragma unroll
            for (int m = 0; m < kMaxCols; ++m) {
                const int c = c0 + m * kHmThreads;
                if (c < ncol_o)
#pragma unroll 4
                    for (int n = rl; n < kTN; n += R) gb1[m] += bufG[n * LD + c];
            }
        }
        __syncthreads();
        // F. gs = gh1 W1^T -> bufH (over s)
        for (int t = wave; t < K * 4; t += kHmWaves) {
            const int k = t >> 2, rg = t & 3;
            f32x4 acc[T];
#pragma unroll
            for (int ti = 0; ti < T; ++ti) acc[ti] = {0.f, 0.f, 0.f, 0.f};
            rows_times_weights<T>(acc, bufG, k * DO, DO, wt1 + k * PP, rg, LD, lr, lq);
#pragma unroll
            for (int ti = 0; ti < T; ++ti) {
                const int col = ti * 16 + lr;
                if (col < DI)
#pragma unroll
                    for (int r = 0; r < 4; ++r) bufH[(rg * 16 + 4 * lq + r) * LD + k * DI + col] = acc[ti][r];
            }
        }
        __syncthreads();
        stage_out(p.gs + n0 * ncol_i, bufH, rows, ncol_i, LD, p.vec_i);
    }

    // per-block partials -> slab row [dW1 | db1 | dW2 | db2 | dtheta]
    float* row = p.slab + (int64_t)blockIdx.x * p.slab_w;
    const int64_t o_b1 = (int64_t)K * DI * DO, o_w2 = o_b1 + ncol_o, o_b2 = o_w2 + (int64_t)K * DO * DO, o_th = o_b2 + ncol_o;
#pragma unroll
    for (int m = 0; m < kMaxItems; ++m) {
        const int it = wave + m * kHmWaves;
        if (it < nitems) {
            const int k = it / TT, tt = it - k * TT, ti = tt / T, tj = tt - ti * T;
            const int j = tj * 16 + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = ti * 16 + 4 * lq + r;
                if (i < DI && j < DO) row[((int64_t)k * DI + i) * DO + j] = acc1[m][r];
                if (i < DO && j < DO) row[o_w2 + ((int64_t)k * DO + i) * DO + j] = acc2[m][r];
            }
        }
    }
    // column sums: the R row lanes of a column meet in LDS (bufG is free now)
    for (int which = 0; which < 3; ++which) {
        if (which == 2 && !p.theta) break;
        __syncthreads();
        if (col_on) {
#pragma unroll
            for (int m = 0; m < kMaxCols; ++m) {
                const int c = c0 + m * kHmThreads;
                if (c < ncol_o) bufG[rl * ncol_o + c] = which == 0 ? gb1[m] : (which == 1 ? gb2[m] : gth[m]);
            }
        }
        __syncthreads();
        const int64_t off = which == 0 ? o_b1 : (which == 1 ? o_b2 : o_th);
        for (int c = tid; c < ncol_o; c += kHmThreads) {
            float tot = 0.f;
            for (int r = 0; r < R; ++r) tot += bufG[r * ncol_o + c];
            row[off + c] = tot;
        }
    }
}

struct HmPlan { int T, LD, grid; size_t lds_fwd, lds_bwd; int64_t slab_w; };

int hm_plan(int64_t N, int K, int DI, int DO, bool theta, HmPlan* pl) {
    const int D = DI > DO ? DI : DO;
    if (D > 32) return fail(KPGNN_ELIMIT, "hop_mlp: per-hop width %d exceeds 32", D);
    pl->T = (D + 15) / 16;
    if (K * pl->T * pl->T > kMaxItems * kHmWaves)
        return fail(KPGNN_ELIMIT, "hop_mlp: K=%d x %d^2 weight tiles exceed %d", K, pl->T, kMaxItems * kHmWaves);
    if ((int64_t)K * D > (int64_t)kMaxCols * kHmThreads)
        return fail(KPGNN_ELIMIT, "hop_mlp: K*D=%lld exceeds %d", (long long)K * D, kMaxCols * kHmThreads);
    const int v = K * D;
    pl->LD = v + ((4 - v % 8) + 8) % 8;                      // pitch = 4 (mod 8)
    const size_t P = 16 * (size_t)pl->T;
    pl->lds_fwd = sizeof(float) * (2 * K * P * P + 3 * K * P + 2 * (size_t)kTN * pl->LD);
    pl->lds_bwd = sizeof(float) * (2 * K * P * P + K * P + 2 * (size_t)kTN * pl->LD);
    const size_t cap = (size_t)device_facts().lds_per_block;
    if (pl->lds_fwd > cap || pl->lds_bwd > cap)
        return fail(KPGNN_ELIMIT, "hop_mlp: %zu B of LDS needed", pl->lds_fwd > pl->lds_bwd ? pl->lds_fwd : pl->lds_bwd);
    const int64_t tiles = (N + kTN - 1) / kTN;
    pl->grid = (int)(tiles < kHmMaxGrid ? tiles : kHmMaxGrid);
    if (pl->grid < 1) pl->grid = 1;
    pl->slab_w = (int64_t)K * DI * DO + (int64_t)K * DO * DO + 2 * (int64_t)K * DO + (theta ? (int64_t)K * DO : 0);
    return KPGNN_OK;
}

int hm_fill(const kpgnn_hop_mlp_desc* d, const HmPlan& pl, HmParams* p) {
    p->N = d->N; p->tiles = (d->N + kTN - 1) / kTN;
    p->K = d->K; p->DI = d->DI; p->DO = d->DO; p->LD = pl.LD;
    p->s = d->s; p->w1 = d->w1; p->b1 = d->b1; p->w2 = d->w2; p->b2 = d->b2; p->theta = d->theta;
    p->h1 = d->h1; p->h2 = d->h2; p->out = d->out; p->gout = d->gout; p->gs = d->gs;
    p->slab = (float*)d->workspace; p->slab_w = pl.slab_w;
    auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
    p->vec_i = ((d->K * d->DI) % 4 == 0) && al16(d->s) && (d->gs == nullptr || al16(d->gs));
    p->vec_o = ((d->K * d->DO) % 4 == 0) && al16(d->h1) && al16(d->h2);
    return KPGNN_OK;
}

}  // namespace
}  // namespace kpgnn

using namespace kpgnn;

extern "C" size_t kpgnn_hop_mlp_workspace_bytes(int64_t N, int32_t K, int32_t DI, int32_t DO) {
    HmPlan pl;
    if (N < 1 || K < 1 || DI < 1 || DO < 1) return 0;
    if (hm_plan(N, K, DI, DO, true, &pl) != KPGNN_OK) return 0;  // 0 = "does not fit": the caller keeps its BLAS path
    return sizeof(float) * (size_t)pl.grid * (size_t)pl.slab_w;
}

static int hm_check(const kpgnn_hop_mlp_desc* d, const char* who) {
    KPGNN_REQUIRE(d != nullptr, "%s: NULL descriptor", who);
    KPGNN_REQUIRE(d->N >= 0 && d->K >= 1 && d->DI >= 1 && d->DO >= 1, "%s: bad N=%lld K=%d DI=%d DO=%d", who,
                  (long long)d->N, d->K, d->DI, d->DO);
    KPGNN_REQUIRE(d->s && d->w1 && d->b1 && d->w2 && d->b2 && d->h1 && d->h2, "%s: NULL tensor", who);
    return KPGNN_OK;
}

extern "C" int kpgnn_hop_mlp_fwd(const kpgnn_hop_mlp_desc* d, kpgnn_stream_t stream) {
    if (int rc = hm_check(d, "hop_mlp_fwd")) return rc;
    KPGNN_REQUIRE(!d->theta || d->out, "hop_mlp_fwd: theta without out");
    if (d->N == 0) return KPGNN_OK;
    HmPlan pl;
    if (int rc = hm_plan(d->N, d->K, d->DI, d->DO, d->theta != nullptr, &pl)) return rc;
    HmParams p;
    hm_fill(d, pl, &p);
    hipStream_t s = (hipStream_t)stream;
    if (pl.T == 1) {
        KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)hop_mlp_fwd_kernel<1>, pl.lds_fwd));
        hipLaunchKernelGGL(hop_mlp_fwd_kernel<1>, dim3(pl.grid), dim3(kHmThreads), pl.lds_fwd, s, p);
    } else {
        KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)hop_mlp_fwd_kernel<2>, pl.lds_fwd));
        hipLaunchKernelGGL(hop_mlp_fwd_kernel<2>, dim3(pl.grid), dim3(kHmThreads), pl.lds_fwd, s, p);
    }
    KPGNN_LAUNCH_CHECK("hop_mlp_fwd_kernel");
    return KPGNN_OK;
}

extern "C" int kpgnn_hop_mlp_bwd(const kpgnn_hop_mlp_desc* d, kpgnn_stream_t stream) {
    if (int rc = hm_check(d, "hop_mlp_bwd")) return rc;
    KPGNN_REQUIRE(d->gout && d->gs && d->gflat, "hop_mlp_bwd: NULL gradient tensor");
    HmPlan pl;
    if (int rc = hm_plan(d->N > 0 ? d->N : 1, d->K, d->DI, d->DO, d->theta != nullptr, &pl)) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (d->N == 0) {
        KPGNN_HIP_TRY(hipMemsetAsync(d->gflat, 0, sizeof(float) * (size_t)pl.slab_w, s));
        return KPGNN_OK;
    }
    KPGNN_REQUIRE(d->workspace && d->workspace_bytes >= sizeof(float) * (size_t)pl.grid * (size_t)pl.slab_w,
                  "hop_mlp_bwd: workspace too small");
    HmParams p;
    hm_fill(d, pl, &p);
    if (pl.T == 1) {
        KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)hop_mlp_bwd_kernel<1>, pl.lds_bwd));
        hipLaunchKernelGGL(hop_mlp_bwd_kernel<1>, dim3(pl.grid), dim3(kHmThreads), pl.lds_bwd, s, p);
    } else {
        KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)hop_mlp_bwd_kernel<2>, pl.lds_bwd));
        hipLaunchKernelGGL(hop_mlp_bwd_kernel<2>, dim3(pl.grid), dim3(kHmThreads), pl.lds_bwd, s, p);
    }
    KPGNN_LAUNCH_CHECK("hop_mlp_bwd_kernel");
    return slab_reduce(p.slab, pl.grid, pl.slab_w, d->gflat, pl.slab_w, nullptr, 0, nullptr, s);
}
