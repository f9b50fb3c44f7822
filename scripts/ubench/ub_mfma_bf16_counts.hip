// Micro-test for the next table-gradient design: T[c, :] = sum_r C[c, r] * g[r, :] with C a small-integer
// COUNT matrix and g fp32, on v_mfma_f32_32x32x16_bf16 with g split three ways into bf16 (hi + mid + lo).
// Checks (a) the operand / accumulator lane maps of the guide, (b) how close the 3-way split comes to the fp32 sum.
//   hipcc --offload-arch=gfx950 -O3 -o ub_mfma_bf16_counts ub_mfma_bf16_counts.hip && ./ub_mfma_bf16_counts
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int M = 32, N = 32, K = 64;     // codes x columns, summed over the 64 (node, hop) rows of a tile

// one wave: out[M][N] = C[M][K] * g[K][N]
__global__ void counts_kernel(const float* __restrict__ C, const float* __restrict__ g, float* __restrict__ out, int splits) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f32x16 acc;
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        bf16x8 a, b0, b1, b2;
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + 8 * h + j;
            a[j] = (__bf16)C[r * K + k];                      // A[row r][k]: counts are exact in bf16 up to 256
            const float x = g[k * N + r];                      // B[k][col r]
            const __bf16 hi = (__bf16)x;
            const float r1 = x - (float)hi;
            const __bf16 mid = (__bf16)r1;
            const float r2 = r1 - (float)mid;
            b0[j] = hi; b1[j] = mid; b2[j] = (__bf16)r2;
        }
        // smallest terms first
        if (splits >= 3) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b2, acc, 0, 0, 0);
        if (splits >= 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0, acc, 0, 0, 0);
    }
    // C/D map: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    for (int v = 0; v < 16; ++v) out[((v & 3) + 8 * (v >> 2) + 4 * h) * N + r] = acc[v];
}

int main() {
    std::vector<float> C(M * K), g(K * N), out(M * N);
    srand(7);
    for (auto& c : C) c = (rand() % 100 < 85) ? 0.f : (float)(1 + rand() % 5);          // sparse small counts
    for (auto& x : g) x = ((float)rand() / RAND_MAX - 0.5f) * powf(10.f, (float)(rand() % 5 - 2));   // mixed magnitudes
    float *dC, *dg, *dout;
    hipMalloc(&dC, C.size() * 4); hipMalloc(&dg, g.size() * 4); hipMalloc(&dout, out.size() * 4);
    hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dg, g.data(), g.size() * 4, hipMemcpyHostToDevice);
    for (int splits = 1; splits <= 3; ++splits) {
        hipLaunchKernelGGL(counts_kernel, dim3(1), dim3(64), 0, 0, dC, dg, dout, splits);
        hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0.0, worst32 = 0.0;
        for (int c = 0; c < M; ++c)
            for (int n = 0; n < N; ++n) {
                double ref = 0.0, mag = 0.0;
                float f32 = 0.f;
                for (int k = 0; k < K; ++k) { ref += (double)C[c * K + k] * g[k * N + n]; mag += fabs((double)C[c * K + k] * g[k * N + n]); f32 = fmaf(C[c * K + k], g[k * N + n], f32); }
                if (mag > 0) {
                    worst = fmax(worst, fabs(out[c * N + n] - ref) / mag);
                    worst32 = fmax(worst32, fabs((double)f32 - ref) / mag);
                }
            }
        printf("splits %d: max |err| / sum|terms| = %.3e   (fp32 fmaf chain: %.3e)\n", splits, worst, worst32);
    }
    return 0;
}
