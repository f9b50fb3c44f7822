// Geometric hop-combine weights (reference layers/combine.py:43-50) and their backward, one tiny launch each.
// Contract: include/kpgnn.h, kpgnn_geo_theta_fwd / _bwd.
//   a = sigmoid(alpha[d]);  t[k] = a (1-a)^k;  theta[k,d] = softmax_k(t)
// The framework's op-by-op version is 6 launches forward and 20 backward per layer (sigmoid, arange, pow, mul,
// softmax, their adjoints, the exponent == 0 mask ...) on a [K, D] ~ 100-element tensor: at ~4.6 us of launch
// granularity each that was 0.9 ms of an 8 ms training step.  One thread per column d, K <= 64 in registers-free
// loops (the powers are rebuilt by repeated multiplication).
#include "kpgnn_common.h"

namespace kpgnn {
namespace {

__global__ void geo_theta_fwd_kernel(const float* __restrict__ alpha, int K, int D, float* __restrict__ theta) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    const float a = 1.0f / (1.0f + __expf(-alpha[d]));
    const float q = 1.0f - a;
    float pw = 1.0f, mx = -INFINITY;
    for (int k = 0; k < K; ++k) { mx = fmaxf(mx, a * pw); pw *= q; }
    float sum = 0.f;
    pw = 1.0f;
    for (int k = 0; k < K; ++k) { sum += __expf(a * pw - mx); pw *= q; }
    const float inv = 1.0f / sum;
    pw = 1.0f;
    for (int k = 0; k < K; ++k) { theta[(int64_t)k * D + d] = __expf(a * pw - mx) * inv; pw *= q; }
}

// dt[k] = theta[k] (G[k] - sum_j theta[j] G[j]);  dt[k]/da = q^k - k a q^(k-1);  da/dalpha = a q
__global__ void geo_theta_bwd_kernel(const float* __restrict__ alpha, const float* __restrict__ theta,
                                     const float* __restrict__ gtheta, int K, int D, float* __restrict__ galpha) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    const float a = 1.0f / (1.0f + __expf(-alpha[d]));
    const float q = 1.0f - a;
    float dot = 0.f;
    for (int k = 0; k < K; ++k) dot = fmaf(theta[(int64_t)k * D + d], gtheta[(int64_t)k * D + d], dot);
    float acc = 0.f, pw = 1.0f, pwm1 = 0.f;   // q^k and q^(k-1) (the k = 0 term of the second part is masked: k * .. = 0)
    for (int k = 0; k < K; ++k) {
        const float dt = theta[(int64_t)k * D + d] * (gtheta[(int64_t)k * D + d] - dot);
        acc = fmaf(dt, pw - (float)k * a * pwm1, acc);
        pwm1 = pw;
        pw *= q;
    }
    galpha[d] = a * q * acc;
}

// The slab of per-block gtheta partials [nslab][K*D] added up in block order AND the theta backward, one launch: a block
// owns CB columns, thread (slice, o) adds every 16th slab row of output o = (k, column), the slices meet in LDS, then one
// thread per column runs the K-term backward.  (Replaces slab_reduce + geo_theta_bwd: one launch less per layer.)
__global__ void __launch_bounds__(1024)
gtheta_finish_kernel(const ThetaFinish f) {
    __shared__ float sm[1168];
    theta_finish_block(f, blockIdx.x, sm);     // (kpgnn_common.h: shared with the table-gradient finishing launch)
}

}  // namespace
}  // namespace kpgnn

namespace kpgnn {
int gtheta_finish_launch(const float* slab, int nslab, const float* alpha, const float* theta, int K, int D, float* gtheta,
                         float* galpha, hipStream_t s) {
    if (K > 64) return fail(KPGNN_ELIMIT, "gtheta_finish: K=%d > 64", K);
    const int CB = 64 / K;                     // columns per block: CB * K <= 64 outputs
    ThetaFinish f;
    f.slab = slab; f.nslab = nslab; f.alpha = alpha; f.theta = theta; f.K = K; f.D = D; f.gtheta = gtheta; f.galpha = galpha;
    hipLaunchKernelGGL(gtheta_finish_kernel, dim3((D + CB - 1) / CB), dim3(1024), 0, s, f);
    KPGNN_LAUNCH_CHECK("gtheta_finish_kernel");
    return KPGNN_OK;
}
int geo_theta_fwd_launch(const float* alpha, int K, int D, float* theta, hipStream_t s) {
    hipLaunchKernelGGL(geo_theta_fwd_kernel, dim3((D + 63) / 64), dim3(64), 0, s, alpha, K, D, theta);
    KPGNN_LAUNCH_CHECK("geo_theta_fwd_kernel");
    return KPGNN_OK;
}
}  // namespace kpgnn

using namespace kpgnn;

extern "C" int kpgnn_geo_theta_fwd(const float* alpha, int32_t K, int32_t D, float* theta, kpgnn_stream_t stream) {
    KPGNN_REQUIRE(alpha && theta && K >= 1 && D >= 1, "geo_theta_fwd: bad arguments K=%d D=%d", K, D);
    return geo_theta_fwd_launch(alpha, K, D, theta, (hipStream_t)stream);
}

extern "C" int kpgnn_geo_theta_bwd(const float* alpha, const float* theta, const float* gtheta, int32_t K, int32_t D,
                                   float* galpha, kpgnn_stream_t stream) {
    KPGNN_REQUIRE(alpha && theta && gtheta && galpha && K >= 1 && D >= 1, "geo_theta_bwd: bad arguments K=%d D=%d", K, D);
    hipLaunchKernelGGL(geo_theta_bwd_kernel, dim3((D + 63) / 64), dim3(64), 0, (hipStream_t)stream, alpha, theta, gtheta, K, D, galpha);
    KPGNN_LAUNCH_CHECK("geo_theta_bwd_kernel");
    return KPGNN_OK;
}
