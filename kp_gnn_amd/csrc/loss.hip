// Mean absolute / mean squared error of a batch of graph scores and its gradient in one launch (gfx950).
// Contract: include/kpgnn.h, kpgnn_regression_loss.
//
// The training scripts end a step with  loss = (score - y).abs().mean()  (train_ZINC.py:42) or  ((score - y) ** 2).mean()
// (train_qm9.py:96): as framework ops that is three launches forward and four backward for a few thousand numbers - 7 of a
// step's ~200 launches, ~35 us of a 1.2-ms step at batch 64.  One block: every thread sums a strided slice in a fixed
// order, the 1024 partials meet in LDS (fixed tree): bitwise reproducible.
#include "kpgnn_common.h"

namespace kpgnn {
namespace {

__global__ void __launch_bounds__(1024)
regression_loss_kernel(const float* __restrict__ score, const float* __restrict__ y, int64_t n, int kind,
                       float* __restrict__ loss, float* __restrict__ dscore) {
    __shared__ float red[1024];
    const float inv_n = 1.0f / (float)n;
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const float d = score[i] - y[i];
        if (kind == 0) {
            s += fabsf(d);
            if (dscore) dscore[i] = d > 0.f ? inv_n : (d < 0.f ? -inv_n : 0.f);      // sign(0) = 0, as the framework's
        } else {
            s = fmaf(d, d, s);
            if (dscore) dscore[i] = 2.0f * d * inv_n;
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = red[0] * inv_n;
}

}  // namespace
}  // namespace kpgnn

using namespace kpgnn;

extern "C" int kpgnn_regression_loss(const float* score, const float* y, int64_t n, int32_t kind, float* loss, float* dscore,
                                     kpgnn_stream_t stream) {
    KPGNN_REQUIRE(score && y && loss && n >= 1, "regression_loss: bad arguments (n = %lld)", (long long)n);
    KPGNN_REQUIRE(kind == 0 || kind == 1, "regression_loss: kind %d is neither 0 (L1) nor 1 (MSE)", kind);
    hipLaunchKernelGGL(regression_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, score, y, n, kind, loss, dscore);
    KPGNN_LAUNCH_CHECK("regression_loss_kernel");
    return KPGNN_OK;
}
