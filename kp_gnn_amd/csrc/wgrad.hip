// dW = dy^T x, db = sum dy for tall-skinny activations on the fp32 matrix cores (gfx950).
// Contract: include/kpgnn.h, kpgnn_linear_wgrad.
//
// v_mfma_f32_32x32x2_f32 (exact f32, fmaf-chain numerics): lane l feeds A[i = l&31][k = l>>5] and
// B[k = l>>5][j = l&31]; with A = dy^T and B = x and k = two consecutive rows r0, r0+1 both operands are plain
// coalesced reads of 32 consecutive floats of a row - straight from global memory, no LDS, no transposes.
// Wave w of a block owns output rows o in [32w, 32w+32) and ALL column tiles (TI accumulators of 16 VGPRs);
// blocks split the N rows; partial dW strips go to a slab that is added in block order.
#include "kpgnn_common.h"

namespace kpgnn {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int kWgradBlocks = 256;

struct WgParams {
    int64_t N; int O, I;
    const float* dy; int64_t dys;
    const float* x; int64_t xs;
    float* slab;   // [gridDim.x][O*I + O]
};

template <int TI>
__global__ void __launch_bounds__(512)
wgrad_kernel(const WgParams p) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kk = lane >> 5, c = lane & 31;
    const int o = wave * 32 + c;                 // this lane's dy column
    const bool o_ok = o < p.O;
    f32x16 acc[TI];
#pragma unroll
    for (int t = 0; t < TI; ++t)
        for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
    float bsum = 0.f;
    // rows of this block: pairs (r, r+1); block b takes pairs b, b+grid, ...
    const int64_t pairs = (p.N + 1) / 2;
    constexpr int UN = 4;
    for (int64_t pr = blockIdx.x; pr < pairs; pr += (int64_t)gridDim.x * UN) {
        float a[UN], b[UN][TI];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int64_t r = 2 * (pr + (int64_t)u * gridDim.x) + kk;
            const bool r_ok = r < p.N;
            a[u] = (r_ok && o_ok) ? p.dy[r * p.dys + o] : 0.f;
#pragma unroll
            for (int t = 0; t < TI; ++t) {
                const int i = t * 32 + c;
                b[u][t] = (r_ok && i < p.I) ? p.x[r * p.xs + i] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            bsum += a[u];
#pragma unroll
            for (int t = 0; t < TI; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u][t], acc[t], 0, 0, 0);
        }
    }
    // C/D map: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float* out = p.slab + (int64_t)blockIdx.x * ((int64_t)p.O * p.I + p.O);
#pragma unroll
    for (int t = 0; t < TI; ++t) {
        const int i = t * 32 + c;
        for (int v = 0; v < 16; ++v) {
            const int orow = wave * 32 + (v & 3) + 8 * (v >> 2) + 4 * kk;
            if (orow < p.O && i < p.I) out[(int64_t)orow * p.I + i] = acc[t][v];
        }
    }
    bsum += __shfl_xor(bsum, 32);                // the two k halves hold different rows of the same column
    if (kk == 0 && o_ok) out[(int64_t)p.O * p.I + o] = bsum;
}

}  // namespace
}  // namespace kpgnn

using namespace kpgnn;

extern "C" size_t kpgnn_wgrad_workspace_bytes(int32_t O, int32_t I) {
    if (O < 1 || I < 1) return 0;
    return sizeof(float) * ((size_t)kWgradBlocks * ((size_t)O * I + O) + O);  // slabs + a sink for an unwanted db
}

extern "C" int kpgnn_linear_wgrad(const kpgnn_wgrad_desc* d, kpgnn_stream_t stream) {
    KPGNN_REQUIRE(d != nullptr, "linear_wgrad: NULL descriptor");
    KPGNN_REQUIRE(d->N >= 1 && d->O >= 1 && d->I >= 1, "linear_wgrad: bad N=%lld O=%d I=%d", (long long)d->N, d->O, d->I);
    if (d->O > 256 || d->I > 256) return fail(KPGNN_ELIMIT, "linear_wgrad: O=%d, I=%d exceed 256", d->O, d->I);
    KPGNN_REQUIRE(d->dy && d->x && d->dw && d->dy_stride >= d->O && d->x_stride >= d->I, "linear_wgrad: bad pointers/strides");
    KPGNN_REQUIRE(d->workspace && d->workspace_bytes >= kpgnn_wgrad_workspace_bytes(d->O, d->I), "linear_wgrad: workspace too small");
    WgParams p;
    p.N = d->N; p.O = d->O; p.I = d->I; p.dy = d->dy; p.dys = d->dy_stride; p.x = d->x; p.xs = d->x_stride;
    p.slab = (float*)d->workspace;
    const int waves = (d->O + 31) / 32, ti = (d->I + 31) / 32;
    int64_t pairs = (d->N + 1) / 2;
    int grid = (int)(pairs < kWgradBlocks ? pairs : kWgradBlocks);
    hipStream_t s = (hipStream_t)stream;
    dim3 blk(waves * 64);
    switch (ti) {
        case 1: hipLaunchKernelGGL(wgrad_kernel<1>, dim3(grid), blk, 0, s, p); break;
        case 2: hipLaunchKernelGGL(wgrad_kernel<2>, dim3(grid), blk, 0, s, p); break;
        case 3: hipLaunchKernelGGL(wgrad_kernel<3>, dim3(grid), blk, 0, s, p); break;
        case 4: hipLaunchKernelGGL(wgrad_kernel<4>, dim3(grid), blk, 0, s, p); break;
        case 5: hipLaunchKernelGGL(wgrad_kernel<5>, dim3(grid), blk, 0, s, p); break;
        case 6: hipLaunchKernelGGL(wgrad_kernel<6>, dim3(grid), blk, 0, s, p); break;
        case 7: hipLaunchKernelGGL(wgrad_kernel<7>, dim3(grid), blk, 0, s, p); break;
        default: hipLaunchKernelGGL(wgrad_kernel<8>, dim3(grid), blk, 0, s, p); break;
    }
    KPGNN_LAUNCH_CHECK("wgrad_kernel");
    const int64_t nw = (int64_t)d->O * d->I;
    float* db = d->db ? d->db : p.slab + (size_t)kWgradBlocks * (nw + d->O);  // sink behind the slabs
    return slab_reduce(p.slab, grid, nw + d->O, d->dw, nw, db, d->O, nullptr, s);
}
