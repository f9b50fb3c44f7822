// Micro-benchmark for the round-2 BatchNorm fusion: column statistics accumulated with fp64 atomics into R replicas
// (same-address atomics serialise at ~18 ns each on MI355X, so 512 blocks must not all hit the same 2C addresses),
// the consumer sums the replicas in its prologue and the LAST consumer block (two-level counters, <= 48 increments
// per address) zeroes the slot, so a slot is all-zero at rest and needs no memset / step hook.
// build: hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics ub_replica_stats.hip -o ub_replica_stats
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int kCnt = 32;
struct Slot { double* acc; unsigned* cnt; int R; };   // acc [R][2C], cnt [kCnt + 1]

template <int MODE>   // 0: slab row, 1: replica atomics
__global__ void __launch_bounds__(256) colsum_kernel(const float* x, int64_t N, int C, float* slab, Slot s) {
    const int G = 32, rl = threadIdx.x / G, sl = threadIdx.x % G, c0 = sl * 4;
    double a[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    if (c0 < C)
        for (int64_t r = (int64_t)blockIdx.x * 8 + rl; r < N; r += (int64_t)gridDim.x * 8) {
            const float4 v = *reinterpret_cast<const float4*>(x + r * C + c0);
            a[0] += v.x; a[1] += v.y; a[2] += v.z; a[3] += v.w;
            q[0] += (double)v.x * v.x; q[1] += (double)v.y * v.y; q[2] += (double)v.z * v.z; q[3] += (double)v.w * v.w;
        }
    __shared__ double red[2][8][128];
    for (int i = 0; i < 4; ++i) { red[0][rl][sl * 4 + i] = a[i]; red[1][rl][sl * 4 + i] = q[i]; }
    __syncthreads();
    if (threadIdx.x < 2 * C) {
        const int which = threadIdx.x / C, c = threadIdx.x % C;
        double t = 0;
        for (int r = 0; r < 8; ++r) t += red[which][r][c];
        if (MODE == 0) slab[(int64_t)blockIdx.x * 2 * C + threadIdx.x] = (float)t;
        else atomicAdd(s.acc + (int64_t)(blockIdx.x % s.R) * 2 * C + threadIdx.x, t);
    }
}

template <bool CLEAN>
__global__ void __launch_bounds__(256) apply_kernel(const float* x, float* y, int64_t N, int C, Slot s) {
    __shared__ float sc[128], sh[128];
    if (threadIdx.x < C) {
        double s0 = 0, s1 = 0;
        for (int r = 0; r < s.R; ++r) { s0 += s.acc[(int64_t)r * 2 * C + threadIdx.x]; s1 += s.acc[(int64_t)r * 2 * C + C + threadIdx.x]; }
        const double m = s0 / (double)N, v = s1 / (double)N - m * m;
        const float is = (float)(1.0 / sqrt((v > 0 ? v : 0) + 1e-5));
        sc[threadIdx.x] = is; sh[threadIdx.x] = (float)(-m) * is;
    }
    __syncthreads();
    const int G = 32, rl = threadIdx.x / G, sl = threadIdx.x % G, c0 = sl * 4;
    if (c0 < C)
        for (int64_t r = (int64_t)blockIdx.x * 8 + rl; r < N; r += (int64_t)gridDim.x * 8) {
            float4 v = *reinterpret_cast<const float4*>(x + r * C + c0);
            v.x = v.x * sc[c0] + sh[c0]; v.y = v.y * sc[c0 + 1] + sh[c0 + 1]; v.z = v.z * sc[c0 + 2] + sh[c0 + 2]; v.w = v.w * sc[c0 + 3] + sh[c0 + 3];
            *reinterpret_cast<float4*>(y + r * C + c0) = v;
        }
    if (CLEAN) {
        // every block is done reading the slot (its prologue) when it gets here: count it on one of kCnt first-level
        // counters; whoever completes a first-level counter counts on the second level; whoever completes that one
        // zeroes the slot and the counters.
        __shared__ int last;
        if (threadIdx.x == 0) {
            last = 0;
            const unsigned g = gridDim.x, j = blockIdx.x % kCnt;
            const unsigned quota = g / kCnt + (j < g % kCnt ? 1u : 0u);
            if (atomicAdd(s.cnt + j, 1u) == quota - 1) {
                const unsigned groups = g < kCnt ? g : kCnt;
                if (atomicAdd(s.cnt + kCnt, 1u) == groups - 1) last = 1;
            }
        }
        __syncthreads();
        if (last) {
            for (int i = threadIdx.x; i < s.R * 2 * C; i += 256) s.acc[i] = 0.0;
            if (threadIdx.x <= kCnt) s.cnt[threadIdx.x] = 0u;
        }
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
    float *x, *y, *slab; double* acc; unsigned* cnt;
    CK(hipMalloc(&x, N * C * 4)); CK(hipMalloc(&y, N * C * 4)); CK(hipMalloc(&slab, 2048 * 2 * C * 4));
    CK(hipMalloc(&acc, 64 * 2 * C * 8)); CK(hipMalloc(&cnt, (kCnt + 1) * 4));
    CK(hipMemset(acc, 0, 64 * 2 * C * 8)); CK(hipMemset(cnt, 0, (kCnt + 1) * 4));
    std::vector<float> h(N * C); for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
    CK(hipMemcpy(x, h.data(), N * C * 4, hipMemcpyHostToDevice));
    auto chain = [&](const char* name, int n, auto launch) {
        hipGraph_t g; hipGraphExec_t ex;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < n; ++i) launch();
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        const float ms = time_graph(ex, s, 10);
        printf("%-64s %.2f us per launch (pair)\n", name, ms * 1e3 / n);
    };
    for (int pgrid : {256, 512}) {
        printf("producer grid %d, consumer grid 1024\n", pgrid);
        chain(" slab producer + plain consumer (R=1, stale sums)", 20, [&] {
            Slot sl{acc, cnt, 1};
            hipLaunchKernelGGL(colsum_kernel<0>, dim3(pgrid), dim3(256), 0, s, x, N, C, slab, sl);
            hipLaunchKernelGGL(apply_kernel<false>, dim3(1024), dim3(256), 0, s, x, y, N, C, sl); });
        for (int R : {1, 4, 8, 16, 32, 64}) {
            char nm[128];
            snprintf(nm, sizeof nm, " R=%d replica atomics + consumer sums R, no clean", R);
            chain(nm, 20, [&] {
                Slot sl{acc, cnt, R};
                hipLaunchKernelGGL(colsum_kernel<1>, dim3(pgrid), dim3(256), 0, s, x, N, C, slab, sl);
                hipLaunchKernelGGL(apply_kernel<false>, dim3(1024), dim3(256), 0, s, x, y, N, C, sl); });
            CK(hipMemset(acc, 0, 64 * 2 * C * 8));
            snprintf(nm, sizeof nm, " R=%d replica atomics + consumer sums R, last reader cleans", R);
            chain(nm, 20, [&] {
                Slot sl{acc, cnt, R};
                hipLaunchKernelGGL(colsum_kernel<1>, dim3(pgrid), dim3(256), 0, s, x, N, C, slab, sl);
                hipLaunchKernelGGL(apply_kernel<true>, dim3(1024), dim3(256), 0, s, x, y, N, C, sl); });
            std::vector<double> a(64 * 2 * C); CK(hipStreamSynchronize(s)); CK(hipMemcpy(a.data(), acc, a.size() * 8, hipMemcpyDeviceToHost));
            double mx = 0; for (double v : a) mx = fabs(v) > mx ? fabs(v) : mx;
            if (mx != 0) printf("   !! slot not clean: %g\n", mx);
        }
    }
    // correctness: y column 0 mean ~ 0, var ~ 1 after producer + cleaning consumer
    { Slot sl{acc, cnt, 16};
      hipLaunchKernelGGL(colsum_kernel<1>, dim3(512), dim3(256), 0, s, x, N, C, slab, sl);
      hipLaunchKernelGGL(apply_kernel<true>, dim3(1024), dim3(256), 0, s, x, y, N, C, sl);
      CK(hipStreamSynchronize(s));
      std::vector<float> hy(N * C); CK(hipMemcpy(hy.data(), y, N * C * 4, hipMemcpyDeviceToHost));
      double m = 0, v = 0; for (int64_t r = 0; r < N; ++r) { m += hy[r * C + 5]; v += (double)hy[r * C + 5] * hy[r * C + 5]; }
      printf("normalised column 5: mean %.3e var %.6f\n", m / N, v / N - (m / N) * (m / N)); }
    return 0;
}
