// Backward pre-pass of the fused aggregation epilogue (gfx950).  Contract: include/kpgnn.h, kpgnn_combine_bwd.
//
// The forward epilogue is v = act(S) + P, hout = sum_k theta[k] * v[k] (KPGINplus.py:76-78 + combine.py:43-46,
// KPGCN.py:113-116).  Its backward in the framework was a dozen elementwise launches over [N,K,D] (broadcast
// multiply, erf, exp, products, sums, an einsum).  Here one streaming pass reads S once and writes g = dL/dS once:
// a sub-group of G lanes owns a node, lanes span the D columns 16 B wide, the per-thread theta-gradient partial
// sums live in registers (KMAX x VEC) and leave through a per-block slab reduced in block order (deterministic).
#include <cstdlib>
#include <initializer_list>

#include "kpgnn_common.h"

namespace kpgnn {
namespace {

constexpr int kBlock = 256;

template <int VEC> struct VT;
template <> struct VT<1> { using T = float; };
template <> struct VT<2> { using T = float2; };
template <> struct VT<4> { using T = float4; };

template <int VEC> __device__ __forceinline__ void ldv(const float* p, float (&v)[VEC]) {
    typename VT<VEC>::T t = *reinterpret_cast<const typename VT<VEC>::T*>(p);
    for (int q = 0; q < VEC; ++q) v[q] = reinterpret_cast<const float*>(&t)[q];
}
template <int VEC> __device__ __forceinline__ void ldv_stream(const float* p, float (&v)[VEC]) {   // read-once stream (nt)
    for (int q = 0; q < VEC; ++q) v[q] = __builtin_nontemporal_load(p + q);
}
template <int VEC> __device__ __forceinline__ void stv(float* p, const float (&v)[VEC]) {
    typename VT<VEC>::T t;
    for (int q = 0; q < VEC; ++q) reinterpret_cast<float*>(&t)[q] = v[q];
    *reinterpret_cast<typename VT<VEC>::T*>(p) = t;
}

struct CbParams {
    int N, K, D, mode;
    const float* pre;
    const float* gh;
    const float* theta;
    const float* gout; int64_t go_sn, go_sk;
    const float* periph; int64_t p_sn, p_sk;
    const float* ptab; const int32_t* uid; int64_t uid_stride;
    float* g;
    float* gv;
    float* slab;     // [gridDim.x][K][D] theta-gradient partials, or NULL
    int lds_ptab;    // floats of ptab staged in LDS (0: read from global)
    int bf;          // pre and g are bf16 rows
};

// Row streaming: a sub-group of G lanes owns one (node, hop) ROW of S per step and strides over the rows; the grid
// holds a multiple of K sub-groups, so a sub-group always meets the SAME hop k: its theta row and its theta-gradient
// partial are VEC registers, and two rows handled by one wave are adjacent in memory (the earlier node-per-sub-group
// walk touched two 416-byte segments 3.3 KB apart per instruction and ran at half the speed of a copy).
// UN rows per trip keep UN independent loads in flight per lane; `pre` and `g` never alias.
// ACT: 1 = GELU (KP-GIN+), 2 = ReLU (KP-GCN), 0 = none; WGT: theta gradient wanted.  Compile-time so that the
// per-element arithmetic is straight-line code (as runtime switches it was a chain of uniform branches per element
// that kept the VALU from overlapping the four rows in flight).
// BF: `pre` and `g` are bf16 rows (2 bytes per element); the arithmetic and every other operand stay fp32.
template <int VEC, int G, int ACT, bool WGT, bool BF = false>
__global__ void __launch_bounds__(kBlock)
combine_bwd_kernel(const CbParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [lds_ptab] dictionary rows, then [NODES][D] reduction
    constexpr int NODES = kBlock / G;
    constexpr int UN = WGT ? 2 : 4;   // (the theta-gradient path keeps ghv / pv / a alive per row: 158 VGPRs at 4 rows)
    const int sg = threadIdx.x / G, sl = threadIdx.x % G;
    const int c0 = sl * VEC;
    const int D = p.D, K = p.K;
    const bool col_ok = c0 < D;
    const bool fused = p.theta != nullptr;
    constexpr bool want_gt = WGT;
    float* pt_l = lds;
    float* red = lds + ((p.lds_ptab + 3) & ~3);
    for (int t = threadIdx.x; t < p.lds_ptab; t += kBlock) pt_l[t] = p.ptab[t];
    if (p.lds_ptab) __syncthreads();
    const float* __restrict__ ptp = p.lds_ptab ? pt_l : p.ptab;
    const float* __restrict__ pre = p.pre;
    float* __restrict__ gout_p = p.g;

    const int64_t R = (int64_t)p.N * K;
    const int64_t total_sg = (int64_t)gridDim.x * NODES;            // multiple of K (host)
    const int64_t r0 = (int64_t)blockIdx.x * NODES + sg;
    const int k = (int)(r0 % K);
    const int64_t istep = total_sg / K;
    float thv[VEC], gt[VEC];
    for (int q = 0; q < VEC; ++q) { thv[q] = 0.f; gt[q] = 0.f; }
    if (fused && col_ok) ldv<VEC>(p.theta + k * D + c0, thv);
    if (col_ok) {
        // (explicit double buffering of the row loads was tried: 160 vs 114 us - the second register set costs a wave
        //  per SIMD; UN independent rows per trip with one set is the better trade)
        int64_t i = r0 / K;
        for (int64_t r = r0; r < R; r += total_sg * UN) {
            float s[UN][VEC], gvv[UN][VEC], ghv[UN][VEC], pv[UN][VEC];
            int u_id[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int64_t ru = r + u * total_sg, iu = i + u * istep;
                for (int q = 0; q < VEC; ++q) { s[u][q] = 0.f; gvv[u][q] = 0.f; ghv[u][q] = 0.f; pv[u][q] = 0.f; }
                u_id[u] = -1;
                if (ru < R) {
                    if (BF) ld_bf16_stream<VEC>(reinterpret_cast<const uint16_t*>(pre) + ru * D + c0, s[u]);
                    else ldv_stream<VEC>(pre + ru * D + c0, s[u]);
                    if (fused) ldv<VEC>(p.gh + iu * D + c0, ghv[u]);
                    else ldv<VEC>(p.gout + iu * p.go_sn + (int64_t)k * p.go_sk + c0, gvv[u]);
                    if (want_gt) {
                        if (p.periph) ldv<VEC>(p.periph + iu * p.p_sn + (int64_t)k * p.p_sk + c0, pv[u]);
                        else if (p.uid) u_id[u] = p.uid[iu * p.uid_stride + k];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int64_t ru = r + u * total_sg;
                if (ru >= R) break;
                float gg[VEC], a[VEC];
                if (fused) for (int q = 0; q < VEC; ++q) gvv[u][q] = thv[q] * ghv[u][q];
                for (int q = 0; q < VEC; ++q) {
                    const float sv = s[u][q];
                    if (ACT == 1) {
                        float e2;  // exp(-s^2/2) comes with the erf approximation
                        const float cdf = 0.5f * (1.0f + fast_erf(sv * 0.70710678118654752440f, &e2));
                        const float pdf = e2 * 0.39894228040143267794f;
                        a[q] = sv * cdf;
                        gg[q] = gvv[u][q] * (cdf + sv * pdf);
                    } else if (ACT == 2) {
                        a[q] = fmaxf(sv, 0.f);
                        gg[q] = sv > 0.f ? gvv[u][q] : 0.f;
                    } else {
                        a[q] = sv;
                        gg[q] = gvv[u][q];
                    }
                }
                if (BF) st_bf16<VEC>(reinterpret_cast<uint16_t*>(gout_p) + ru * D + c0, gg);
                else stv<VEC>(gout_p + ru * D + c0, gg);
                if (p.gv) stv<VEC>(p.gv + ru * D + c0, gvv[u]);
                if (want_gt) {
                    if (u_id[u] >= 0) ldv<VEC>(ptp + (int64_t)u_id[u] * D + c0, pv[u]);
                    for (int q = 0; q < VEC; ++q) gt[q] = fmaf(ghv[u][q], a[q] + pv[u][q], gt[q]);
                }
            }
            i += istep * UN;
        }
    }
    if (!want_gt) return;
    // per-block theta-gradient partial: sub-groups of the block that met hop k2 are added in sub-group order
    if (col_ok) for (int q = 0; q < VEC; ++q) red[sg * D + c0 + q] = gt[q];
    __syncthreads();
    // (hop of sub-group s2 = (kb0 + s2) mod K: walked in sub-group order, no per-element division / 64-bit modulo -
    //  the earlier form of this tail cost ~15 us of a 150 us launch)
    float* out = p.slab + (size_t)blockIdx.x * K * D;
    const int kb0 = (int)(((int64_t)blockIdx.x * NODES) % K);
    for (int dcol = threadIdx.x; dcol < D; dcol += kBlock) {
        for (int k2 = 0; k2 < K; ++k2) {
            float tot = 0.f;
            int kk = kb0;
            for (int s2 = 0; s2 < NODES; ++s2) {
                if (kk == k2) tot += red[s2 * D + dcol];
                if (++kk == K) kk = 0;
            }
            out[k2 * D + dcol] = tot;
        }
    }
}

// grid: <= 8 blocks per CU, (grid * sub-groups per block) a multiple of K
int cb_grid(int N, int K, int G, int per_cu) {
    const int nodes = kBlock / G;
    const int64_t rows = (int64_t)N * K;
    int64_t g = (int64_t)device_facts().cu_count * (per_cu >= 1 && per_cu <= 8 ? per_cu : 8);
    const int64_t need = (rows + nodes - 1) / nodes;
    if (g > need) g = need;
    // smallest m with (m * nodes) % K == 0
    int m = 1;
    while ((m * nodes) % K) ++m;
    g = ((g + m - 1) / m) * m;
    return (int)(g < m ? m : g);
}

int cb_shape(const kpgnn_combine_bwd_desc* d, int* vec, int* g) {
    int v = (d->D % 4 == 0) ? 4 : (d->D % 2 == 0 ? 2 : 1);
    auto al = [&](const void* q) { while (v > 1 && q && ((uintptr_t)q % (v * 4))) v >>= 1; };
    al(d->pre); al(d->gh); al(d->theta); al(d->gout); al(d->periph); al(d->ptab); al(d->g); al(d->gv);
    for (int64_t s : {d->gout ? d->go_sn : 0, d->gout ? d->go_sk : 0, d->periph ? d->p_sn : 0, d->periph ? d->p_sk : 0})
        while (v > 1 && (s % v)) v >>= 1;
    const int lanes = (d->D + v - 1) / v;
    if (lanes > 64) return fail(KPGNN_ELIMIT, "combine_bwd: D=%d needs %d lanes > 64", d->D, lanes);
    int gg = 4;
    while (gg < lanes) gg <<= 1;
    *vec = v; *g = gg;
    return KPGNN_OK;
}

template <int VEC, int G, int ACT, bool WGT, bool BF = false>
int cb_launch2(const CbParams& p, int* grid_out, size_t lds, hipStream_t s) {
    if (lds > 64 * 1024) KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)combine_bwd_kernel<VEC, G, ACT, WGT, BF>, lds));
    // grid-stride kernel: one resident round (kpgnn_common.h, resident_blocks)
    const int grid = cb_grid(p.N, p.K, G, resident_blocks(combine_bwd_kernel<VEC, G, ACT, WGT, BF>, kBlock, lds));
    *grid_out = grid;
    hipLaunchKernelGGL((combine_bwd_kernel<VEC, G, ACT, WGT, BF>), dim3(grid), dim3(kBlock), lds, s, p);
    KPGNN_LAUNCH_CHECK("combine_bwd_kernel");
    return KPGNN_OK;
}

template <int VEC, int G>
int cb_launch(const CbParams& p, int* grid, hipStream_t s) {
    const size_t lds = sizeof(float) * (size_t)(((p.lds_ptab + 3) & ~3) + (p.slab ? (kBlock / G) * p.D : 0));
    const int act = p.mode == KPGNN_MODE_GINPLUS ? 1 : (p.mode == KPGNN_MODE_GCN ? 2 : 0);
    if (p.bf) {     // bf16 rows: the KP-GIN+ epilogue, 4 elements per lane
        if constexpr (VEC == 4) {
            if (act == 1 && !p.gv) return p.slab ? cb_launch2<VEC, G, 1, true, true>(p, grid, lds, s) : cb_launch2<VEC, G, 1, false, true>(p, grid, lds, s);
        }
        return fail(KPGNN_EINVAL, "combine_bwd: bf16 storage needs mode GINPLUS, no gv and D %% 4 == 0");
    }
    if (p.slab) {
        if (act == 1) return cb_launch2<VEC, G, 1, true>(p, grid, lds, s);
        if (act == 2) return cb_launch2<VEC, G, 2, true>(p, grid, lds, s);
        return cb_launch2<VEC, G, 0, true>(p, grid, lds, s);
    }
    if (act == 1) return cb_launch2<VEC, G, 1, false>(p, grid, lds, s);
    if (act == 2) return cb_launch2<VEC, G, 2, false>(p, grid, lds, s);
    return cb_launch2<VEC, G, 0, false>(p, grid, lds, s);
}

}  // namespace
}  // namespace kpgnn

using namespace kpgnn;

extern "C" size_t kpgnn_combine_bwd_workspace_bytes(int32_t N, int32_t K, int32_t D) {
    if (N <= 0 || K < 1 || D < 1) return 0;
    return sizeof(float) * ((size_t)device_facts().cu_count * 8 + 16) * K * D;  // upper bound on grid * K * D (cb_grid)
}

extern "C" int kpgnn_combine_bwd(const kpgnn_combine_bwd_desc* d, kpgnn_stream_t stream) {
    KPGNN_REQUIRE(d != nullptr, "combine_bwd: NULL descriptor");
    KPGNN_REQUIRE(d->N >= 0 && d->K >= 1 && d->K <= 16 && d->D >= 1, "combine_bwd: bad N=%d K=%d D=%d (K <= 16)", d->N, d->K, d->D);
    if (d->N == 0) return KPGNN_OK;
    KPGNN_REQUIRE(d->pre && d->g, "combine_bwd: NULL pre/g");
    KPGNN_REQUIRE(d->theta ? d->gh != nullptr : d->gout != nullptr, "combine_bwd: need (theta, gh) or gout");
    KPGNN_REQUIRE(d->gtheta == nullptr || d->theta != nullptr, "combine_bwd: gtheta without theta");
    KPGNN_REQUIRE(!d->galphas || (d->alphas && d->gtheta), "combine_bwd: galphas needs alphas and gtheta");
    KPGNN_REQUIRE(d->mode >= KPGNN_MODE_GIN && d->mode <= KPGNN_MODE_SUM, "combine_bwd: unknown mode %d", d->mode);
    int vec = 1, g = 4;
    int rc = cb_shape(d, &vec, &g);
    if (rc != KPGNN_OK) return rc;
    CbParams p;
    p.N = d->N; p.K = d->K; p.D = d->D; p.mode = d->mode; p.pre = d->pre; p.gh = d->gh; p.theta = d->theta;
    p.gout = d->gout; p.go_sn = d->go_sn; p.go_sk = d->go_sk; p.periph = d->periph; p.p_sn = d->p_sn; p.p_sk = d->p_sk;
    p.ptab = d->periph ? nullptr : d->ptab; p.uid = d->periph ? nullptr : d->uid; p.uid_stride = d->uid_stride;
    p.g = d->g; p.gv = d->gv; p.slab = nullptr; p.bf = d->storage == KPGNN_STORE_BF16 ? 1 : 0;
    KPGNN_REQUIRE(d->storage == KPGNN_STORE_F32 || d->storage == KPGNN_STORE_BF16, "combine_bwd: unknown storage %d", d->storage);
    int grid = cb_grid(d->N, d->K, g, 8);           // upper bound (workspace check); the launcher picks the real one
    p.lds_ptab = 0;
    if (d->gtheta) {
        const size_t need = sizeof(float) * (size_t)grid * d->K * d->D;
        KPGNN_REQUIRE(d->workspace && d->workspace_bytes >= need, "combine_bwd: workspace too small (%zu < %zu)",
                      (size_t)d->workspace_bytes, need);
        p.slab = (float*)d->workspace;
        if (p.ptab && p.uid && d->n_dict > 0 && (size_t)d->n_dict * d->D * sizeof(float) <= 16 * 1024) p.lds_ptab = d->n_dict * d->D;
    }
    hipStream_t s = (hipStream_t)stream;
#define KP_CB(V, GG) rc = cb_launch<V, GG>(p, &grid, s); break
    switch (vec * 100 + g) {
        case 404: KP_CB(4, 4); case 408: KP_CB(4, 8); case 416: KP_CB(4, 16); case 432: KP_CB(4, 32); case 464: KP_CB(4, 64);
        case 204: KP_CB(2, 4); case 208: KP_CB(2, 8); case 216: KP_CB(2, 16); case 232: KP_CB(2, 32); case 264: KP_CB(2, 64);
        case 104: KP_CB(1, 4); case 108: KP_CB(1, 8); case 116: KP_CB(1, 16); case 132: KP_CB(1, 32); case 164: KP_CB(1, 64);
        default: return fail(KPGNN_EINVAL, "combine_bwd: no kernel for vec=%d g=%d", vec, g);
    }
#undef KP_CB
    if (rc != KPGNN_OK) return rc;
    if (p.slab && d->alphas && d->galphas)    // geometric combine: the finishing launch also differentiates theta(alphas)
        return gtheta_finish_launch(p.slab, grid, d->alphas, d->theta, d->K, d->D, d->gtheta, d->galphas, s);
    if (p.slab) return slab_reduce(p.slab, grid, (int64_t)d->K * d->D, d->gtheta, (int64_t)d->K * d->D, nullptr, 0, nullptr, s);
    return KPGNN_OK;
}
