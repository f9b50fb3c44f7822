// y = act(sum_l x_l W_l^T + b): a Linear over the CONCATENATION of S states that live in separate tensors (gfx950).
// Contract: include/kpgnn.h, kpgnn_linear_group_fwd.
//
// The bodies' jumping-knowledge projection (models/GNNs.py:216-218, :455-457, :703-705) is
// `output_proj(torch.cat(h_list, dim=-1))`: a 178-MB concat copy (83 us) + a [47k, 936] x [936, 104] library GEMM (117 us)
// per step at K = L = 8, h = 104.  Here the K-loop of the GEMM runs over the S state POINTERS: a 32M-row tile of state l is
// staged in LDS (requested one state ahead, held in registers across the MFMA phase), every wave reloads its 32 x I strip of
// W's column block l (L2-resident) and continues the same accumulators; bias + ReLU leave with the tile.  No concat exists.
#include "kpgnn_common.h"

namespace kpgnn {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ float4 ldg4(const float* q) { return *reinterpret_cast<const float4*>(q); }

struct LgParams {
    int64_t N; const int32_t* n_dyn;
    int O, I, S, pitch, relu;
    const float* xs[16]; int64_t xstride;
    const float* w; const float* bias; float* y;
};

// One 256-thread block per CU (one wave per SIMD: the whole 512-entry register file per lane): a tile of 32*M rows, M up to 6,
// so that the launch is one round of the chip; the NEXT state's tile rows and weight strip are both requested before the
// MFMA chain of the current state and held in registers across it (at two blocks per CU there was no room for either: the
// strip arrived behind the prefetch it was issued after, and the prefetch spilled).
template <int KS, int M>
__global__ void __launch_bounds__(256, 1)
linear_group_kernel(const LgParams p) {
    extern __shared__ __attribute__((aligned(16))) float xl[];      // [32*M][pitch]
    constexpr int ROWS = 32 * M;
    constexpr int I = 2 * KS, CGI = I / 4;
    constexpr int PF = (ROWS * CGI + 255) / 256;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int kk = lane >> 5, c = lane & 31;
    const int O = p.O, S = p.S, pitch = p.pitch;
    const int64_t N = p.n_dyn ? (int64_t)min((int64_t)*p.n_dyn, p.N) : p.N;
    const int64_t tiles = (N + ROWS - 1) / ROWS;
    if (tiles == 0) return;                            // (n_dyn == 0)
    const int o = wave * 32 + c;
    const int64_t wrow = (int64_t)S * I;
    const float* wo = p.w + (int64_t)(o < O ? o : O - 1) * wrow;    // (padded lanes: last valid row, never stored)
    // this thread's float4 slots of a tile (fixed)
    int prow[PF], pcg[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) { const int e = tid + 256 * j; prow[j] = e < ROWS * CGI ? e / CGI : -1; pcg[j] = e % CGI; }
    float4 pf[PF];
    // (unconditional loads from clamped rows: a predicated load becomes a branch, and the compiler cannot count vector-memory
    //  operations behind branches - it then waits for ALL of them; rows beyond N only feed output rows that are never stored)
    const int64_t last = N - 1;
    auto issue = [&](int64_t tl, int l) {
        const int64_t r0 = tl * ROWS;
        // (uniform selects: a runtime index into the kernel-argument array would put a copy of it in scratch memory, and a
        //  pointer table in LDS loses the address space - flat loads, and the prefetch array lands in scratch too)
        const float* base = p.xs[0];
#pragma unroll
        for (int i = 1; i < 16; ++i)
            if (l == i) base = p.xs[i];
#pragma unroll
        for (int j = 0; j < PF; ++j)
            pf[j] = ldg4(base + min(r0 + max(prow[j], 0), last) * p.xstride + 4 * pcg[j]);
    };
    auto commit = [&]() {
#pragma unroll
        for (int j = 0; j < PF; ++j)
            if (prow[j] >= 0) *reinterpret_cast<float4*>(xl + prow[j] * pitch + 4 * pcg[j]) = pf[j];
    };
    // a wave's strip of W's column block l as MFMA A-fragments: a[ks] = W[o][l*I + 2 ks + kk]
    float an[KS];
    auto load_strip = [&](int l) {
#pragma unroll
        for (int j = 0; j < KS / 2; ++j) {
            const float4 v = ldg4(wo + (int64_t)l * I + 4 * j);
            an[2 * j] = kk ? v.y : v.x;
            an[2 * j + 1] = kk ? v.w : v.z;
        }
    };
    int64_t tile = blockIdx.x;
    load_strip(0);
    issue(tile, 0);
    commit();
    __syncthreads();
    const float* b0 = xl + c * pitch + kk;
    for (; tile < tiles; tile += gridDim.x) {
        f32x16 acc[M];
#pragma unroll
        for (int m = 0; m < M; ++m)
            for (int v = 0; v < 16; ++v) acc[m][v] = 0.f;
        const bool more_tiles = tile + gridDim.x < tiles;
#pragma unroll 1
        for (int l = 0; l < S; ++l) {
            float a[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) a[ks] = an[ks];
            const bool has_next = l + 1 < S || more_tiles;
            const int ln = l + 1 < S ? l + 1 : 0;
            // the next state's strip, then its tile rows (the block's very last request re-reads state 0 and is never used: an
            // unconditional request keeps the instruction count behind the waits fixed)
            load_strip(ln);
            issue(l + 1 < S ? tile : (more_tiles ? tile + gridDim.x : tile), ln);
            int z = 0;
            asm volatile("" : "+v"(z));                          // (keeps the LDS operand reads inside this state's iteration)
            const float* bz = b0 + z;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int m = 0; m < M; ++m)
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks], bz[m * 32 * pitch + 2 * ks], acc[m], 0, 0, 0);
            }
            __syncthreads();                           // every wave is done reading this state's tile
            if (has_next) commit();
            __syncthreads();
        }
        // C/D map: col = lane & 31 (tile row), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (output o)
        const int64_t r0 = tile * ROWS;
        // this lane's bias values, all four requested before the first store (one counter for vector loads and stores: a load
        // between the stores would wait for the stores in front of it)
        float4 bias4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ob = wave * 32 + 8 * g + 4 * kk;
            bias4[g] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.bias) bias4[g] = ldg4(p.bias + (ob < O ? ob : 0));
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ob = wave * 32 + 8 * g + 4 * kk;
            if (ob < O) {                              // O % 4 == 0 (host)
                const float4 bb = bias4[g];
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    const int64_t r = r0 + m * 32 + c;
                    if (r < N) {
                        float4 v = make_float4(acc[m][4 * g] + bb.x, acc[m][4 * g + 1] + bb.y, acc[m][4 * g + 2] + bb.z, acc[m][4 * g + 3] + bb.w);
                        if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                        *reinterpret_cast<float4*>(p.y + r * O + ob) = v;
                    }
                }
            }
        }
    }
}

}  // namespace
}  // namespace kpgnn

using namespace kpgnn;

extern "C" size_t kpgnn_linear_split_workspace_bytes(int32_t O, int32_t I, int32_t group) { return linear3_workspace_bytes(O, I, group); }

extern "C" int kpgnn_linear_group_fwd(const kpgnn_linear_group_desc* d, kpgnn_stream_t stream) {
    KPGNN_REQUIRE(d != nullptr, "linear_group_fwd: NULL descriptor");
    KPGNN_REQUIRE(d->N >= 1 && d->O >= 1 && d->I >= 1 && d->group >= 1 && d->group <= 16, "linear_group_fwd: bad N=%lld O=%d I=%d group=%d",
                  (long long)d->N, d->O, d->I, d->group);
    if (d->O > 128 || d->O % 4 != 0 || (d->I != 32 && d->I != 64 && d->I != 96 && d->I != 104 && d->I != 128))
        return fail(KPGNN_ELIMIT, "linear_group_fwd: needs O <= 128, O %% 4 == 0 and I in {32, 64, 96, 104, 128} (the k-loop is fully unrolled)");
    KPGNN_REQUIRE(d->w && d->y && d->x_stride >= d->I, "linear_group_fwd: NULL pointer or x_stride < I");
    auto al = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
    if (d->x_stride % 4 != 0 || !al(d->w) || !al(d->y) || (d->bias && !al(d->bias)))
        return fail(KPGNN_ELIMIT, "linear_group_fwd: needs 16-B aligned operands");
    LgParams p;
    p.N = d->N; p.n_dyn = d->n_dyn; p.O = d->O; p.I = d->I; p.S = d->group; p.relu = d->relu ? 1 : 0;
    for (int l = 0; l < 16; ++l) {
        p.xs[l] = d->x[l < d->group ? l : 0];
        if (l < d->group && (!p.xs[l] || !al(p.xs[l]))) return fail(KPGNN_ELIMIT, "linear_group_fwd: state %d is NULL or not 16-B aligned", l);
    }
    {
        bool handled = false;
        const int rc = linear3_group_fwd(d, (hipStream_t)stream, &handled);       // the bf16-split kernel, where it applies
        if (handled || rc != KPGNN_OK) return rc;
    }
    p.xstride = d->x_stride; p.w = d->w; p.bias = d->bias; p.y = d->y;
    p.pitch = d->I + ((4 - d->I % 8) + 8) % 8;          // pitch = 4 (mod 8) floats: 16-B aligned rows, conflict-free operand reads
    // rows per tile = 32 * m, m in {1, 2, 3, 4, 6}: the smallest that makes the launch one round over one block per CU
    const int64_t slots = (int64_t)device_facts().cu_count;
    int m = (int)((d->N + slots * 32 - 1) / (slots * 32));
    m = m < 1 ? 1 : (m > 4 ? 6 : m);
    const int rows = 32 * m;
    const size_t lds = sizeof(float) * (size_t)rows * p.pitch;
    const int64_t tiles = (d->N + rows - 1) / rows;
    const int64_t grid = slots < tiles ? slots : tiles;
    hipStream_t s = (hipStream_t)stream;
#define KP_LG2(KSV, MV) do { \
        KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)linear_group_kernel<KSV, MV>, lds)); \
        hipLaunchKernelGGL((linear_group_kernel<KSV, MV>), dim3((unsigned)grid), dim3(256), lds, s, p); } while (0)
#define KP_LG(KSV) do { if (m == 1) KP_LG2(KSV, 1); else if (m == 2) KP_LG2(KSV, 2); else if (m == 3) KP_LG2(KSV, 3); \
                        else if (m == 4) KP_LG2(KSV, 4); else KP_LG2(KSV, 6); } while (0)
    switch (d->I) {
        case 32: KP_LG(16); break;
        case 64: KP_LG(32); break;
        case 96: KP_LG(48); break;
        case 104: KP_LG(52); break;
        default: KP_LG(64); break;
    }
#undef KP_LG
#undef KP_LG2
    KPGNN_LAUNCH_CHECK("linear_group_kernel");
    return KPGNN_OK;
}
