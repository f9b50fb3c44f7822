// Adam step over ONE flat parameter bucket (gfx950).  Contract: include/kpgnn.h, kpgnn_adam_step.
//
// The training scripts step torch.optim.Adam over ~190 parameter tensors (train_ZINC.py:244).  With parameters and
// gradients re-homed in one flat bucket each (dp.py) the update is one elementwise pass over ~0.5 M floats - but the
// framework's fused multi-tensor kernel walks a tensor in 65,536-element chunks, one block per chunk: 8 blocks, 42 us
// (+ a launch for the step counter) for 14 MB of traffic.  Here: 16 B per lane, one block per 4 KB, the step number and
// the bias corrections ride in the arguments (the step is launched eagerly, outside the captured graph).
#include "kpgnn_common.h"

namespace kpgnn {
namespace {

struct AdamArgs {
    float* p; const float* g; float* m; float* v; int64_t n;
    float lr_over_bc1, inv_sqrt_bc2, omb1, beta2, omb2, eps, wd;    // omb = 1 - beta, formed in double on the host
    long long* state;             // device {step, ticket} or NULL: the step number lives on the device (a captured launch)
    double lr, beta1d, beta2d;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamArgs& a) {
    if (a.wd != 0.f) g = fmaf(a.wd, p, g);
    m = fmaf(g - m, a.omb1, m);                         // lerp(m, g, 1 - beta1)
    v = fmaf(a.beta2, v, a.omb2 * g * g);
    const float denom = sqrtf(v) * a.inv_sqrt_bc2 + a.eps;
    p -= a.lr_over_bc1 * (m / denom);
}

__global__ void __launch_bounds__(256) adam_kernel(AdamArgs a) {
    if (a.state) {
        // Replayed from a graph the arguments never change: the step number is read from device memory by every block
        // as it starts, and bumped by the block that FINISHES last (ticket) - after every other block has read it.
        __shared__ float bc[2];
        if (threadIdx.x == 0) {
            const double t = (double)(a.state[0] + 1);
            bc[0] = (float)(a.lr / (1.0 - pow(a.beta1d, t)));
            bc[1] = (float)(1.0 / sqrt(1.0 - pow(a.beta2d, t)));
        }
        __syncthreads();
        a.lr_over_bc1 = bc[0]; a.inv_sqrt_bc2 = bc[1];
    }
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < a.n) {
        float4 p = *reinterpret_cast<float4*>(a.p + i), m = *reinterpret_cast<float4*>(a.m + i), v = *reinterpret_cast<float4*>(a.v + i);
        const float4 g = *reinterpret_cast<const float4*>(a.g + i);
        adam_one(p.x, g.x, m.x, v.x, a); adam_one(p.y, g.y, m.y, v.y, a);
        adam_one(p.z, g.z, m.z, v.z, a); adam_one(p.w, g.w, m.w, v.w, a);
        *reinterpret_cast<float4*>(a.p + i) = p; *reinterpret_cast<float4*>(a.m + i) = m; *reinterpret_cast<float4*>(a.v + i) = v;
    } else {
        for (int64_t q = i; q < a.n; ++q) adam_one(a.p[q], a.g[q], a.m[q], a.v[q], a);
    }
    if (a.state) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            const unsigned long long ticket = atomicAdd(reinterpret_cast<unsigned long long*>(a.state + 1), 1ull);
            if (ticket == (unsigned long long)gridDim.x - 1) { a.state[1] = 0; a.state[0] += 1; }
        }
    }
}

}  // namespace
}  // namespace kpgnn

using namespace kpgnn;

static int adam_launch(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t step, int64_t* state,
                       double lr, double beta1, double beta2, double eps, double weight_decay, kpgnn_stream_t stream) {
    KPGNN_REQUIRE(param && grad && exp_avg && exp_avg_sq && n >= 0 && (state || step >= 1), "adam_step: bad arguments");
    KPGNN_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
                  "adam_step: the buckets must be 16-B aligned");
    if (n == 0) return KPGNN_OK;
    AdamArgs a;
    a.p = param; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq; a.n = n;
    a.state = reinterpret_cast<long long*>(state); a.lr = lr; a.beta1d = beta1; a.beta2d = beta2;
    if (state) step = 1;           // (placeholders: the kernel forms the corrections from the device step)
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    a.lr_over_bc1 = (float)(lr / bc1); a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    a.omb1 = (float)(1.0 - beta1); a.beta2 = (float)beta2; a.omb2 = (float)(1.0 - beta2); a.eps = (float)eps; a.wd = (float)weight_decay;
    const int64_t blocks = (n + 1023) / 1024;
    if (blocks > 0x7fffffff) return fail(KPGNN_ELIMIT, "adam_step: n = %lld is too large", (long long)n);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    KPGNN_LAUNCH_CHECK("adam_kernel");
    return KPGNN_OK;
}

extern "C" int kpgnn_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t step,
                               double lr, double beta1, double beta2, double eps, double weight_decay, kpgnn_stream_t stream) {
    return adam_launch(param, grad, exp_avg, exp_avg_sq, n, step, nullptr, lr, beta1, beta2, eps, weight_decay, stream);
}

extern "C" int kpgnn_adam_step_device(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t* state,
                                      double lr, double beta1, double beta2, double eps, double weight_decay,
                                      kpgnn_stream_t stream) {
    KPGNN_REQUIRE(state != nullptr, "adam_step_device: NULL state");
    return adam_launch(param, grad, exp_avg, exp_avg_sq, n, 0, state, lr, beta1, beta2, eps, weight_decay, stream);
}
