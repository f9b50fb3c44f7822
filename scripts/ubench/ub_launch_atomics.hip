// Micro-benchmarks that informed the round-2 BatchNorm fusion design (not part of the product library):
//  1. cost of one kernel node inside a replayed hipGraph (empty kernels, chain of 400)
//  2. a 20 MB streaming column-sum kernel whose blocks finish with (a) a slab row, (b) 2C fp64 atomics, (c) 2C fp32 atomics
//  3. consumer prologue: read 2C doubles + "last reader zeroes the slot"
// build: hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics ub_launch_atomics.hip -o ub_launch_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }

template <int MODE>   // 0 slab, 1 fp64 atomics, 2 fp32 atomics, 3 nothing
__global__ void __launch_bounds__(256) colsum_kernel(const float* x, int64_t N, int C, float* slab, double* acc64, float* acc32) {
    const int G = 32, rl = threadIdx.x / G, sl = threadIdx.x % G, c0 = sl * 4;
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    if (c0 < C)
        for (int64_t r = (int64_t)blockIdx.x * 8 + rl; r < N; r += (int64_t)gridDim.x * 8) {
            const float4 v = *reinterpret_cast<const float4*>(x + r * C + c0);
            s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
            q[0] += (double)v.x * v.x; q[1] += (double)v.y * v.y; q[2] += (double)v.z * v.z; q[3] += (double)v.w * v.w;
        }
    __shared__ double red[2][8][128];
    for (int i = 0; i < 4; ++i) { red[0][rl][sl * 4 + i] = s[i]; red[1][rl][sl * 4 + i] = q[i]; }
    __syncthreads();
    if (threadIdx.x < 2 * C) {
        const int which = threadIdx.x / C, c = threadIdx.x % C;
        double t = 0;
        for (int r = 0; r < 8; ++r) t += red[which][r][c];
        if (MODE == 0) slab[(int64_t)blockIdx.x * 2 * C + threadIdx.x] = (float)t;
        if (MODE == 1) atomicAdd(acc64 + threadIdx.x, t);
        if (MODE == 2) atomicAdd(acc32 + threadIdx.x, (float)t);
    }
}

// consumer: every block reads the 2C sums, the last reader zeroes them (self-cleaning slot), then streams x -> y
template <bool CLEAN>
__global__ void __launch_bounds__(256) apply_kernel(const float* x, float* y, int64_t N, int C, double* acc64, unsigned* cnt) {
    __shared__ float sc[128], sh[128];
    if (threadIdx.x < C) {
        const double m = acc64[threadIdx.x] / (double)N, v = acc64[C + threadIdx.x] / (double)N - m * m;
        const float is = (float)(1.0 / sqrt((v > 0 ? v : 0) + 1e-5));
        sc[threadIdx.x] = is; sh[threadIdx.x] = (float)(-m) * is;
    }
    __syncthreads();
    if (CLEAN && threadIdx.x == 0) {
        const unsigned old = atomicAdd(cnt, 1u);
        if (old == gridDim.x - 1) { for (int i = 0; i < 2 * C; ++i) acc64[i] = 0.0; *cnt = 0; }
    }
    const int G = 32, rl = threadIdx.x / G, sl = threadIdx.x % G, c0 = sl * 4;
    if (c0 >= C) return;
    for (int64_t r = (int64_t)blockIdx.x * 8 + rl; r < N; r += (int64_t)gridDim.x * 8) {
        float4 v = *reinterpret_cast<const float4*>(x + r * C + c0);
        v.x = v.x * sc[c0] + sh[c0]; v.y = v.y * sc[c0 + 1] + sh[c0 + 1]; v.z = v.z * sc[c0 + 2] + sh[c0 + 2]; v.w = v.w * sc[c0 + 3] + sh[c0 + 3];
        *reinterpret_cast<float4*>(y + r * C + c0) = v;
    }
}

static float time_graph(hipGraphExec_t ex, hipStream_t s, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipGraphLaunch(ex, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s));
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ex, s));
    CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    const int64_t N = 47450; const int C = 104;
    float *x, *y, *slab, *acc32; double* acc64; unsigned* cnt;
    CK(hipMalloc(&x, N * C * 4)); CK(hipMalloc(&y, N * C * 4)); CK(hipMalloc(&slab, 2048 * 2 * C * 4));
    CK(hipMalloc(&acc32, 2 * C * 4)); CK(hipMalloc(&acc64, 2 * C * 8)); CK(hipMalloc(&cnt, 4));
    CK(hipMemset(acc64, 0, 2 * C * 8)); CK(hipMemset(acc32, 0, 2 * C * 4)); CK(hipMemset(cnt, 0, 4));
    std::vector<float> h(N * C); for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
    CK(hipMemcpy(x, h.data(), N * C * 4, hipMemcpyHostToDevice));
    // ---- 1. empty chain
    for (int n : {100, 400}) {
        hipGraph_t g; hipGraphExec_t ex;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, (int*)nullptr);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        const float ms = time_graph(ex, s, 20);
        printf("empty chain of %d kernels: %.1f us per replay = %.2f us per node\n", n, ms * 1e3, ms * 1e3 / n);
        // eager
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        CK(hipEventRecord(a, s));
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, (int*)nullptr);
        CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
        float el; CK(hipEventElapsedTime(&el, a, b));
        printf("eager chain of %d kernels: %.2f us per launch\n", n, el * 1e3 / n);
    }
    // ---- 2./3. streaming kernels, each variant as a chain of 20 in a graph
    auto chain = [&](const char* name, auto launch) {
        hipGraph_t g; hipGraphExec_t ex;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 20; ++i) launch();
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        const float ms = time_graph(ex, s, 10);
        printf("%-44s %.2f us per launch\n", name, ms * 1e3 / 20);
    };
    for (int grid : {512, 1024, 2048}) {
        printf("grid %d\n", grid);
        chain(" colsum, no epilogue", [&] { hipLaunchKernelGGL(colsum_kernel<3>, dim3(grid), dim3(256), 0, s, x, N, C, slab, acc64, acc32); });
        chain(" colsum, slab row per block", [&] { hipLaunchKernelGGL(colsum_kernel<0>, dim3(grid), dim3(256), 0, s, x, N, C, slab, acc64, acc32); });
        chain(" colsum, 2C fp64 atomics per block", [&] { hipLaunchKernelGGL(colsum_kernel<1>, dim3(grid), dim3(256), 0, s, x, N, C, slab, acc64, acc32); });
        chain(" colsum, 2C fp32 atomics per block", [&] { hipLaunchKernelGGL(colsum_kernel<2>, dim3(grid), dim3(256), 0, s, x, N, C, slab, acc64, acc32); });
    }
    CK(hipMemset(acc64, 0, 2 * C * 8));
    for (int grid : {512, 1483}) {
        printf("apply grid %d\n", grid);
        chain(" apply, plain", [&] { hipLaunchKernelGGL(apply_kernel<false>, dim3(grid), dim3(256), 0, s, x, y, N, C, acc64, cnt); });
        chain(" apply, last reader zeroes the slot", [&] { hipLaunchKernelGGL(apply_kernel<true>, dim3(grid), dim3(256), 0, s, x, y, N, C, acc64, cnt); });
    }
    // correctness of the self-cleaning slot: colsum(fp64 atomics) -> apply(clean) twice, slot must be zero after
    hipLaunchKernelGGL(colsum_kernel<1>, dim3(512), dim3(256), 0, s, x, N, C, slab, acc64, acc32);
    hipLaunchKernelGGL(apply_kernel<true>, dim3(1483), dim3(256), 0, s, x, y, N, C, acc64, cnt);
    CK(hipStreamSynchronize(s));
    std::vector<double> a(2 * C); CK(hipMemcpy(a.data(), acc64, 2 * C * 8, hipMemcpyDeviceToHost));
    double mx = 0; for (double v : a) mx = fabs(v) > mx ? fabs(v) : mx;
    unsigned hc; CK(hipMemcpy(&hc, cnt, 4, hipMemcpyDeviceToHost));
    printf("slot after producer+cleaning consumer: max|acc| = %g, counter = %u (want 0, 0)\n", mx, hc);
    // small-N regime (B = 64: N = 1458)
    {
        const int64_t n2 = 1458;
        chain(" [N=1458] colsum fp64 atomics, grid 46", [&] { hipLaunchKernelGGL(colsum_kernel<1>, dim3(46), dim3(256), 0, s, x, n2, C, slab, acc64, acc32); });
        CK(hipMemset(acc64, 0, 2 * C * 8));
    }
    return 0;
}
