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

// ------------------------------------------------------------------------------------------------ y = x W^T + b
namespace kpgnn {
namespace {

struct LinParams {
    int64_t N; int O, I, pitch;
    const float* x; int64_t xs;
    const float* w; const float* bias;
    float* y; int64_t ys;
};

// KS = number of 2-wide k-steps held in registers (I <= 2*KS)
template <int KS>
__global__ void __launch_bounds__(512)
linear_fwd_kernel(const LinParams p) {
    extern __shared__ __attribute__((aligned(16))) float xl[];      // [2][32][pitch]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kk = lane >> 5, c = lane & 31;
    const int nthreads = blockDim.x;
    const int o = wave * 32 + c;
    // A fragments: W[o][2*ks + kk]
    float a[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int i = 2 * ks + kk;
        a[ks] = (o < p.O && i < p.I) ? p.w[(int64_t)o * p.I + i] : 0.f;
    }
    const int64_t tiles = (p.N + 31) / 32;
    const int tile_elems = 32 * p.I;
    // staging map, tile independent: element e = threadIdx.x + q*nthreads of the [32, I] tile -> (row, column).
    // The first NPRE elements per thread travel through registers (loaded before the MFMA loop of the current tile,
    // written to the other LDS buffer after it); shapes with more elements per thread stage the rest directly.
    constexpr int NPRE = 16;
    int lo[NPRE], go[NPRE], rw[NPRE];
#pragma unroll
    for (int q = 0; q < NPRE; ++q) {
        const int e = threadIdx.x + q * nthreads;
        const int r = e / p.I, i = e - r * p.I;
        rw[q] = e < tile_elems ? r : 1 << 30;
        lo[q] = r * p.pitch + i;
        go[q] = (int)(r * p.xs + i);
    }
    float pre[NPRE];
    auto load_regs = [&](int64_t tl) {
        const int64_t r0 = tl * 32;
        const float* base = p.x + r0 * p.xs;
#pragma unroll
        for (int q = 0; q < NPRE; ++q) pre[q] = (rw[q] < 32 && r0 + rw[q] < p.N) ? base[go[q]] : 0.f;
    };
    auto store_regs = [&](float* dst) {
#pragma unroll
        for (int q = 0; q < NPRE; ++q) if (rw[q] < 32) dst[lo[q]] = pre[q];
    };
    auto stage_rest = [&](int64_t tl, float* dst) {
        const int64_t r0 = tl * 32;
        for (int e = threadIdx.x + NPRE * nthreads; e < tile_elems; e += nthreads) {
            const int r = e / p.I, i = e - r * p.I;
            dst[r * p.pitch + i] = (r0 + r < p.N) ? p.x[(r0 + r) * p.xs + i] : 0.f;
        }
    };
    int64_t tile = blockIdx.x;
    int buf = 0;
    if (tile < tiles) { load_regs(tile); store_regs(xl); stage_rest(tile, xl); }
    __syncthreads();
    for (; tile < tiles; tile += gridDim.x) {
        float* cur = xl + buf * 32 * p.pitch;
        float* nxt = xl + (buf ^ 1) * 32 * p.pitch;
        const bool more = tile + gridDim.x < tiles;
        if (more) load_regs(tile + gridDim.x);
        f32x16 acc;
        for (int v = 0; v < 16; ++v) acc[v] = 0.f;
        const float* brow = cur + c * p.pitch + kk;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float b = (2 * ks + kk < p.I) ? brow[2 * ks] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks], b, acc, 0, 0, 0);
        }
        // C/D map: col = lane & 31 (row r of the tile), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (output o)
        const int64_t r = tile * 32 + c;
        if (r < p.N) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ob = wave * 32 + 8 * g + 4 * kk;
                if (ob + 3 < p.O) {
                    float4 v = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
                    if (p.bias) { const float4 bb = *reinterpret_cast<const float4*>(p.bias + ob); v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w; }
                    *reinterpret_cast<float4*>(p.y + r * p.ys + ob) = v;
                } else {
                    for (int q = 0; q < 4; ++q)
                        if (ob + q < p.O) p.y[r * p.ys + ob + q] = acc[4 * g + q] + (p.bias ? p.bias[ob + q] : 0.f);
                }
            }
        }
        if (more) { store_regs(nxt); stage_rest(tile + gridDim.x, nxt); }
        __syncthreads();
        buf ^= 1;
    }
}

}  // namespace
}  // namespace kpgnn

extern "C" int kpgnn_linear_fwd(const kpgnn_linear_desc* d, kpgnn_stream_t stream) {
    KPGNN_REQUIRE(d != nullptr, "linear_fwd: NULL descriptor");
    KPGNN_REQUIRE(d->N >= 1 && d->O >= 1 && d->I >= 1, "linear_fwd: bad N=%lld O=%d I=%d", (long long)d->N, d->O, d->I);
    if (d->O > 256 || d->I > 256) return fail(KPGNN_ELIMIT, "linear_fwd: O=%d, I=%d exceed 256", d->O, d->I);
    KPGNN_REQUIRE(d->x && d->w && d->y && d->x_stride >= d->I && d->y_stride >= d->O, "linear_fwd: bad pointers/strides");
    if ((d->y_stride % 4) != 0 || (((uintptr_t)d->y) & 15) != 0 || (d->bias && (((uintptr_t)d->bias) & 15) != 0))
        return fail(KPGNN_ELIMIT, "linear_fwd: y / bias must be 16-B aligned with y_stride %% 4 == 0");
    LinParams p;
    p.N = d->N; p.O = d->O; p.I = d->I; p.pitch = d->I | 1;   // odd pitch: conflict-free column reads
    p.x = d->x; p.xs = d->x_stride; p.w = d->w; p.bias = d->bias; p.y = d->y; p.ys = d->y_stride;
    const int waves = (d->O + 31) / 32;
    const size_t lds = sizeof(float) * 2 * 32 * (size_t)p.pitch;
    const int64_t tiles = (d->N + 31) / 32;
    int64_t grid = (int64_t)device_facts().cu_count * 4;
    if (grid > tiles) grid = tiles;
    hipStream_t s = (hipStream_t)stream;
    dim3 blk(waves * 64);
    const int ks = (d->I + 1) / 2;
#define KP_LIN(KSV) do { KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)linear_fwd_kernel<KSV>, lds)); \
                         hipLaunchKernelGGL(linear_fwd_kernel<KSV>, dim3((unsigned)grid), blk, lds, s, p); } while (0)
    if (ks <= 8) KP_LIN(8);
    else if (ks <= 16) KP_LIN(16);
    else if (ks <= 32) KP_LIN(32);
    else if (ks <= 52) KP_LIN(52);
    else if (ks <= 64) KP_LIN(64);
    else if (ks <= 96) KP_LIN(96);
    else KP_LIN(128);
#undef KP_LIN
    KPGNN_LAUNCH_CHECK("linear_fwd_kernel");
    return KPGNN_OK;
}
