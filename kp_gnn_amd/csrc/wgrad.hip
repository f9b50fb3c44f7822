// dW = dy^T x, db = sum dy for tall-skinny activations on the fp32 matrix cores (gfx950).
// Contract: include/kpgnn.h, kpgnn_linear_wgrad.
//
// v_mfma_f32_32x32x2_f32 (exact f32, fmaf-chain numerics): lane l feeds A[i = l&31][k = l>>5] and
// B[k = l>>5][j = l&31]; with A = dy^T and B = x and k = two consecutive rows r0, r0+1 both operands are plain
// coalesced reads of 32 consecutive floats of a row - straight from global memory, no LDS, no transposes.
// Wave w of a block owns output rows o in [32w, 32w+32) and ALL column tiles (TI accumulators of 16 VGPRs);
// blocks split the N rows; partial dW strips go to a slab that is added in block order.
#include <cstdlib>

#include "kpgnn_common.h"

namespace kpgnn {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int kWgradBlocks = 512;

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
    int64_t N; int O, I, pitch, ypitch, wt;
    const float* x; int64_t xs;
    const float* w; const float* bias;
    float* y; int64_t ys;
};

constexpr int kLinMaxPF = 12;    // float4 registers per thread of the prefetched x tile (up to 96 x 128 floats)

// y^T tile = W_strip (A operand, registers) x x_tile^T (B operand, LDS): wave w owns outputs [32w, 32w+32) for the
// whole launch (KS k-steps of 2 = its strip of W in KS VGPRs); a block streams tiles of 32*M rows of x through LDS
// (odd pitch: conflict-free transposed operand reads; the next tile travels in registers meanwhile); each wave runs M
// independent v_mfma_f32_32x32x2_f32 chains (one per 32-row group) per k-step, and the result leaves through the same
// LDS buffer as whole rows (coalesced 16-B stores).  M is chosen so that the launch is ONE round of tiles over two
// blocks per CU (N = 47k: 96-row tiles, 495 blocks).  wt = 1 reads the weight transposed (w[k][o]): the same kernel
// gives dx = dy W without a transposed copy.
template <int KS, int M>
__global__ void __launch_bounds__(256, 2)
linear_fwd_kernel(const LinParams p) {
    extern __shared__ __attribute__((aligned(16))) float xl[];      // [32*M][pitch]
    constexpr int ROWS = 32 * M;
    constexpr int PF = (M * 32 * 128 / 4 + 255) / 256 > kLinMaxPF ? kLinMaxPF : (M * 32 * 128 / 4 + 255) / 256;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int kk = lane >> 5, c = lane & 31;
    const int I = p.I, O = p.O, pitch = p.pitch;
    const int o = wave * 32 + c;
    // this wave's strip of the weight as MFMA A-fragments: a[ks] = W[o][2 ks + kk]
    float a[KS];
    if (p.wt) {                                       // w is [I][O]: lanes run along o, coalesced as is
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a[ks] = o < O ? p.w[(int64_t)(2 * ks + kk) * O + o] : 0.f;
    } else {                                          // w is [O][I]: every lane streams ITS row 16 B at a time (the two
#pragma unroll                                        // k-halves share the loads) instead of 2*KS strided dword loads
        for (int j = 0; j < KS / 2; ++j) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (o < O) v = *reinterpret_cast<const float4*>(p.w + (int64_t)o * I + 4 * j);
            a[2 * j] = kk ? v.y : v.x;
            a[2 * j + 1] = kk ? v.w : v.z;
        }
    }
    const int64_t tiles = (p.N + ROWS - 1) / ROWS;
    constexpr int IC = 2 * KS;                        // == I (host): compile-time divisor
    int lo[PF];                                       // LDS offset of this thread's q-th float4 (tile independent)
#pragma unroll
    for (int q = 0; q < PF; ++q) {
        const int e = 4 * (tid + q * 256);
        lo[q] = e < ROWS * IC ? (e / IC) * pitch + (e % IC) : -1;
    }
    const int yp = p.ypitch;
    int yo[PF];                                       // LDS offset of the q-th float4 of the y tile (one division, then steps)
    {
        const int step_r = 1024 / O, step_c = 1024 - step_r * O;
        int r = (4 * tid) / O, cc = 4 * tid - r * O;
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            yo[q] = r < ROWS ? r * yp + cc : 0;
            r += step_r; cc += step_c;
            if (cc >= O) { cc -= O; ++r; }
        }
    }
    float4 pf[PF];
    auto issue = [&](int64_t tl) {
        const int64_t r0 = tl * ROWS;
        const int64_t lim = (p.N - r0 < ROWS ? p.N - r0 : ROWS) * (int64_t)I;
        const float* base = p.x + r0 * p.xs;          // xs == I (host): a tile is one contiguous run
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const int e = 4 * (tid + q * 256);
            pf[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < lim) pf[q] = *reinterpret_cast<const float4*>(base + e);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int q = 0; q < PF; ++q)
            if (lo[q] >= 0) *reinterpret_cast<float4*>(xl + lo[q]) = pf[q];
    };
    int64_t tile = blockIdx.x;
    if (tile < tiles) { issue(tile); commit(); }
    __syncthreads();
    for (; tile < tiles; tile += gridDim.x) {
        const bool more = tile + gridDim.x < tiles;
        if (more) issue(tile + gridDim.x);
        f32x16 acc[M];
#pragma unroll
        for (int m = 0; m < M; ++m)
            for (int v = 0; v < 16; ++v) acc[m][v] = 0.f;
        const float* b0 = xl + c * pitch + kk;
        // I == 2 * KS exactly (host): plain LDS reads the scheduler can hoist ahead of the MFMAs (a per-lane predicate
        // on the read made every MFMA wait for its own ds_read: 31 us instead of ~13)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const float xv = b0[m * 32 * pitch + 2 * ks];
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks], xv, acc[m], 0, 0, 0);
            }
        }
        __syncthreads();                               // every wave is done reading the x tile: it becomes the y tile
        {
        // C/D map: col = lane & 31 (tile row), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (output o)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ob = wave * 32 + 8 * g + 4 * kk;
            if (ob < O) {                              // O % 4 == 0 (host): the 4 outputs of a group are in or out together
                float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.bias) bb = *reinterpret_cast<const float4*>(p.bias + ob);
#pragma unroll
                for (int m = 0; m < M; ++m)            // y view: pitch = 4 (mod 8) floats -> 16-B LDS accesses, no conflicts
                    *reinterpret_cast<float4*>(xl + (m * 32 + c) * yp + ob) =
                        make_float4(acc[m][4 * g] + bb.x, acc[m][4 * g + 1] + bb.y, acc[m][4 * g + 2] + bb.z, acc[m][4 * g + 3] + bb.w);
            }
        }
        __syncthreads();
        {
            const int64_t r0 = tile * ROWS;
            const int rows = (int)(p.N - r0 < ROWS ? p.N - r0 : ROWS);
            float* ybase = p.y + r0 * p.ys;            // ys == O (host): whole rows, contiguous
#pragma unroll
            for (int q = 0; q < PF; ++q) {
                const int e = 4 * (tid + q * 256);
                if (e < rows * O) *reinterpret_cast<float4*>(ybase + e) = *reinterpret_cast<const float4*>(xl + yo[q]);
            }
            for (int e = 4 * (tid + PF * 256); e < rows * O; e += 4 * 256) {
                const int r = e / O, oo = e - r * O;
                *reinterpret_cast<float4*>(ybase + e) = *reinterpret_cast<const float4*>(xl + r * yp + oo);
            }
        }
        }
        __syncthreads();                               // the y tile is out: the buffer takes the next x tile
        if (more) commit();
        __syncthreads();
    }
}

// Wide outputs (O > 128, e.g. the input gradient of the jumping-knowledge projection: [N,104] x [104,936]): the x tile
// stays resident in LDS while the block walks the outputs 128 at a time - per chunk every wave reloads its strip of the
// weight (L2-resident) and runs its MFMA chains; results go straight from the accumulators to y (64 x 16-B segments per
// store: measured as fast as staging through LDS), so no barrier separates the chunks.
template <int KS, int M>
__global__ void __launch_bounds__(256, 2)
linear_wide_kernel(const LinParams p) {
    extern __shared__ __attribute__((aligned(16))) float xl[];      // [32*M][pitch]
    constexpr int ROWS = 32 * M;
    constexpr int IC = 2 * KS;                        // == I (host)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int kk = lane >> 5, c = lane & 31;
    const int O = p.O, pitch = p.pitch;
    const int64_t tiles = (p.N + ROWS - 1) / ROWS;
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        __syncthreads();                               // previous tile fully consumed
        {   // x tile -> LDS (no register double-buffer: the eight output chunks dwarf this load)
            const int64_t r0 = tile * ROWS;
            const int lim = (int)(p.N - r0 < ROWS ? p.N - r0 : ROWS) * IC;
            const float* base = p.x + r0 * p.xs;
            for (int e = 4 * tid; e < ROWS * IC; e += 4 * 256) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < lim) v = *reinterpret_cast<const float4*>(base + e);
                *reinterpret_cast<float4*>(xl + (e / IC) * pitch + (e % IC)) = v;
            }
        }
        __syncthreads();
        const int64_t r0 = tile * ROWS;
        const float* b0 = xl + c * pitch + kk;
#pragma unroll 1
        for (int chunk = 0; chunk < O; chunk += 128) {
            // the operand reads below do not depend on the chunk: without this the compiler hoists all of them out of
            // the loop and spills ~500 registers
            int z = 0;
            asm volatile("" : "+v"(z));
            const float* bz = b0 + z;
            const int o = chunk + wave * 32 + c;
            float a[KS];
            if (p.wt) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) a[ks] = o < O ? p.w[(int64_t)(2 * ks + kk) * O + o] : 0.f;
            } else {
#pragma unroll
                for (int j = 0; j < KS / 2; ++j) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (o < O) v = *reinterpret_cast<const float4*>(p.w + (int64_t)o * IC + 4 * j);
                    a[2 * j] = kk ? v.y : v.x;
                    a[2 * j + 1] = kk ? v.w : v.z;
                }
            }
            f32x16 acc[M];
#pragma unroll
            for (int m = 0; m < M; ++m)
                for (int v = 0; v < 16; ++v) acc[m][v] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    const float xv = bz[m * 32 * pitch + 2 * ks];
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks], xv, acc[m], 0, 0, 0);
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ob = chunk + wave * 32 + 8 * g + 4 * kk;
                if (ob < O) {
                    float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (p.bias) bb = *reinterpret_cast<const float4*>(p.bias + ob);
#pragma unroll
                    for (int m = 0; m < M; ++m) {
                        const int64_t r = r0 + m * 32 + c;
                        if (r < p.N)
                            *reinterpret_cast<float4*>(p.y + r * p.ys + ob) =
                                make_float4(acc[m][4 * g] + bb.x, acc[m][4 * g + 1] + bb.y, acc[m][4 * g + 2] + bb.z, acc[m][4 * g + 3] + bb.w);
                    }
                }
            }
        }
    }
}

}  // namespace
}  // namespace kpgnn

extern "C" int kpgnn_linear_fwd(const kpgnn_linear_desc* d, kpgnn_stream_t stream) {
    KPGNN_REQUIRE(d != nullptr, "linear_fwd: NULL descriptor");
    KPGNN_REQUIRE(d->N >= 1 && d->O >= 1 && d->I >= 1, "linear_fwd: bad N=%lld O=%d I=%d", (long long)d->N, d->O, d->I);
    if (d->O > 4096 || d->I > 128) return fail(KPGNN_ELIMIT, "linear_fwd: O=%d exceeds 4096 or I=%d exceeds 128", d->O, d->I);
    KPGNN_REQUIRE(d->x && d->w && d->y, "linear_fwd: NULL pointer");
    if ((d->O % 4) != 0 || (d->I % 4) != 0 || d->x_stride != d->I || d->y_stride != d->O ||
        (((uintptr_t)d->x | (uintptr_t)d->y) & 15) != 0 || (d->bias && (((uintptr_t)d->bias) & 15) != 0))
        return fail(KPGNN_ELIMIT, "linear_fwd: needs contiguous 16-B aligned x / y with I %% 4 == 0 and O %% 4 == 0");
    LinParams p;
    p.N = d->N; p.O = d->O; p.I = d->I; p.wt = d->w_transposed ? 1 : 0;
    // one pitch = 4 (mod 8) floats for the x and the y view of the buffer: 16-B aligned rows (the tile is committed and
    // drained with b128 LDS accesses); the transposed operand reads then see a 2-way bank conflict, which hides behind
    // the 64-cycle MFMAs
    const bool wide = d->O > 128;
    const int wmax = (wide || d->I > d->O) ? d->I : d->O;
    const int rowp = wmax + ((4 - wmax % 8) + 8) % 8;
    p.pitch = rowp; p.ypitch = rowp;
    p.x = d->x; p.xs = d->x_stride; p.w = d->w; p.bias = d->bias; p.y = d->y; p.ys = d->y_stride;
    // rows per tile = 32 * m, m in 1..3, the smallest that makes the launch one round over two blocks per CU
    const int64_t slots = (int64_t)device_facts().cu_count * 2;
    int m = (int)((d->N + slots * 32 - 1) / (slots * 32));
    m = m < 1 ? 1 : (m > 3 ? 3 : m);
    const int rows = 32 * m;
    const size_t lds = sizeof(float) * (size_t)rows * rowp;
    const int64_t tiles = (d->N + rows - 1) / rows;
    int64_t grid = (m == 1 ? slots * 2 : slots) < tiles ? (m == 1 ? slots * 2 : slots) : tiles;
    hipStream_t s = (hipStream_t)stream;
    dim3 blk(256);
    const int ks = (d->I + 1) / 2;
#define KP_LIN2(KSV, MV) do { if (wide) { \
        KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)linear_wide_kernel<KSV, MV>, lds)); \
        hipLaunchKernelGGL((linear_wide_kernel<KSV, MV>), dim3((unsigned)grid), blk, lds, s, p); \
    } else { \
        KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)linear_fwd_kernel<KSV, MV>, lds)); \
        hipLaunchKernelGGL((linear_fwd_kernel<KSV, MV>), dim3((unsigned)grid), blk, lds, s, p); } } while (0)
#define KP_LIN(KSV) do { if (m == 1) KP_LIN2(KSV, 1); else if (m == 2) KP_LIN2(KSV, 2); else KP_LIN2(KSV, 3); } while (0)
    if (ks == 16) KP_LIN(16);
    else if (ks == 32) KP_LIN(32);
    else if (ks == 52) KP_LIN(52);
    else if (ks == 64) KP_LIN(64);
    else return fail(KPGNN_ELIMIT, "linear_fwd: I=%d is not one of 32, 64, 104, 128 (the k-loop is fully unrolled)", d->I);
#undef KP_LIN
#undef KP_LIN2
    KPGNN_LAUNCH_CHECK("linear_fwd_kernel");
    return KPGNN_OK;
}
