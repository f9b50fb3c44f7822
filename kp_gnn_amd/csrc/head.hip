// The graph regressor: score[g] = pooled[g] . w + b, and its backward, one launch each (gfx950).
// Contract: include/kpgnn.h, kpgnn_score_head_fwd / kpgnn_score_head_bwd.
//
// GraphRegression ends with nn.Linear(hidden, 1) on the pooled graph rows (reference models/GraphRegression.py:46-51).  As a
// library GEMM that is a [G, H] x [H, 1] product: two Cijk launches of ~10.7 us each at G = 2048 plus a bias reduce, for 0.4
// MFLOP.  Forward: a wave per graph row (fixed butterfly).  Backward: a block owns 16 columns and walks ALL graphs, so
// dw[c] = sum_g dscore[g] pooled[g][c] needs no partial slabs (64 row lanes summed in a fixed order in LDS): bitwise reproducible.
#include "kpgnn_common.h"

namespace kpgnn {
namespace {

__global__ void __launch_bounds__(256)
score_head_fwd_kernel(const float* __restrict__ pooled, const float* __restrict__ w, const float* __restrict__ bias,
                      int64_t G, int D, float* __restrict__ score) {
    const int lane = threadIdx.x & 63;
    const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= G) return;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s = fmaf(pooled[g * D + c], w[c], s);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) score[g] = s + (bias ? bias[0] : 0.f);
}

constexpr int kHeadCols = 16, kHeadRows = 64;      // 1024 threads: 64 row lanes (a 16-lane block walked 2048 graphs in 47 us)

__global__ void __launch_bounds__(1024)
score_head_bwd_kernel(const float* __restrict__ pooled, const float* __restrict__ w, const float* __restrict__ dscore,
                      int64_t G, int D, float* __restrict__ dpooled, float* __restrict__ dw, float* __restrict__ db) {
    __shared__ float red[kHeadRows][kHeadCols + 1];
    const int cl = threadIdx.x % kHeadCols, rl = threadIdx.x / kHeadCols;
    const int c = blockIdx.x * kHeadCols + cl;
    const bool ok = c < D;
    const float wc = ok ? w[c] : 0.f;
    float acc = 0.f, bsum = 0.f;
    const int cc = ok ? c : D - 1;                     // (padded columns: unconditional loads of the last one, nothing stored)
    int64_t g = rl;
    for (; g + 3 * kHeadRows < G; g += 4 * kHeadRows) {          // four independent rows in flight
        float ds[4], pv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { ds[q] = dscore[g + q * kHeadRows]; pv[q] = pooled[(g + q * kHeadRows) * D + cc]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc = fmaf(ds[q], pv[q], acc);
            bsum += ds[q];
            if (ok && dpooled) dpooled[(g + q * kHeadRows) * D + c] = ds[q] * wc;
        }
    }
    for (; g < G; g += kHeadRows) {
        const float ds = dscore[g];
        acc = fmaf(ds, pooled[g * D + cc], acc);
        if (ok && dpooled) dpooled[g * D + c] = ds * wc;
        bsum += ds;
    }
    red[rl][cl] = acc;
    __syncthreads();
    if (rl == 0 && ok) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < kHeadRows; ++r) s += red[r][cl];
        dw[c] = s;
    }
    if (db && blockIdx.x == 0) {          // (every column lane of a row lane holds the same partial of sum_g dscore[g])
        __syncthreads();
        if (cl == 0) red[rl][0] = bsum;
        __syncthreads();
        if (threadIdx.x == 0) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < kHeadRows; ++r) s += red[r][0];
            db[0] = s;
        }
    }
}

}  // namespace
}  // namespace kpgnn

using namespace kpgnn;

extern "C" int kpgnn_score_head_fwd(const float* pooled, const float* w, const float* bias, int64_t G, int32_t D, float* score,
                                    kpgnn_stream_t stream) {
    KPGNN_REQUIRE(pooled && w && score && G >= 0 && D >= 1, "score_head_fwd: bad arguments (G = %lld, D = %d)", (long long)G, D);
    if (G == 0) return KPGNN_OK;
    hipLaunchKernelGGL(score_head_fwd_kernel, dim3((unsigned)((G + 3) / 4)), dim3(256), 0, (hipStream_t)stream, pooled, w, bias, G, D, score);
    KPGNN_LAUNCH_CHECK("score_head_fwd_kernel");
    return KPGNN_OK;
}

extern "C" int kpgnn_score_head_bwd(const float* pooled, const float* w, const float* dscore, int64_t G, int32_t D, float* dpooled,
                                    float* dw, float* db, kpgnn_stream_t stream) {
    KPGNN_REQUIRE(pooled && w && dscore && dw && G >= 0 && D >= 1, "score_head_bwd: bad arguments (G = %lld, D = %d)", (long long)G, D);
    hipLaunchKernelGGL(score_head_bwd_kernel, dim3((unsigned)((D + kHeadCols - 1) / kHeadCols)), dim3(kHeadCols * kHeadRows), 0, (hipStream_t)stream,
                       pooled, w, dscore, G, D, dpooled, dw, db);
    KPGNN_LAUNCH_CHECK("score_head_bwd_kernel");
    return KPGNN_OK;
}
