// Backward pre-pass of the fused aggregation epilogue (gfx950).  Contract: include/kpgnn.h, kpgnn_combine_bwd.
//
// The forward epilogue is v = act(S) + P, hout = sum_k theta[k] * v[k] (KPGINplus.py:76-78 + combine.py:43-46,
// KPGCN.py:113-116).  Its backward in the framework was a dozen elementwise launches over [N,K,D] (broadcast
// multiply, erf, exp, products, sums, an einsum).  Here one streaming pass reads S once and writes g = dL/dS once:
// a sub-group of G lanes owns a node, lanes span the D columns 16 B wide, the per-thread theta-gradient partial
// sums live in registers (KMAX x VEC) and leave through a per-block slab reduced in block order (deterministic).
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
    float* slab;   // [gridDim.x][K][D] theta-gradient partials, or NULL
};

template <int VEC, int G, int KMAX>
__global__ void __launch_bounds__(kBlock)
combine_bwd_kernel(const CbParams p) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [kBlock/G][K][D] for the block reduction
    constexpr int NODES = kBlock / G;
    const int sg = threadIdx.x / G, sl = threadIdx.x % G;
    const int c0 = sl * VEC;
    const int D = p.D;
    const bool col_ok = c0 < D;
    const bool fused = p.theta != nullptr;
    const bool want_gt = p.slab != nullptr;
    float gt[KMAX][VEC];
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        for (int q = 0; q < VEC; ++q) gt[k][q] = 0.f;
    const int64_t tiles = ((int64_t)p.N + NODES - 1) / NODES;
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t i = tile * NODES + sg;
        if (i >= p.N || !col_ok) continue;
        float ghv[VEC];
        for (int q = 0; q < VEC; ++q) ghv[q] = 0.f;
        if (fused) ldv<VEC>(p.gh + i * D + c0, ghv);
        // all K rows of S of this node are requested before the first one is used (the stores to g below would
        // otherwise fence every load behind them: one exposed round trip per hop)
        float sall[KMAX][VEC];
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < p.K) ldv<VEC>(p.pre + (i * p.K + k) * (int64_t)D + c0, sall[k]);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if (k >= p.K) break;
            float s[VEC], gvv[VEC], gg[VEC], a[VEC];
            for (int q = 0; q < VEC; ++q) s[q] = sall[k][q];
            if (fused) {
                float th[VEC];
                ldv<VEC>(p.theta + k * D + c0, th);
                for (int q = 0; q < VEC; ++q) gvv[q] = th[q] * ghv[q];
            } else {
                ldv<VEC>(p.gout + i * p.go_sn + (int64_t)k * p.go_sk + c0, gvv);
            }
            for (int q = 0; q < VEC; ++q) {
                if (p.mode == KPGNN_MODE_GINPLUS) {
                    float e2;  // exp(-s^2/2) comes with the erf approximation
                    const float cdf = 0.5f * (1.0f + fast_erf(s[q] * 0.70710678118654752440f, &e2));
                    const float pdf = e2 * 0.39894228040143267794f;
                    a[q] = s[q] * cdf;
                    gg[q] = gvv[q] * (cdf + s[q] * pdf);
                } else if (p.mode == KPGNN_MODE_GCN) {
                    a[q] = fmaxf(s[q], 0.f);
                    gg[q] = s[q] > 0.f ? gvv[q] : 0.f;
                } else {
                    a[q] = s[q];
                    gg[q] = gvv[q];
                }
            }
            stv<VEC>(p.g + (i * p.K + k) * (int64_t)D + c0, gg);
            if (p.gv) stv<VEC>(p.gv + (i * p.K + k) * (int64_t)D + c0, gvv);
            if (want_gt) {
                float pv[VEC];
                for (int q = 0; q < VEC; ++q) pv[q] = 0.f;
                if (p.periph) ldv<VEC>(p.periph + i * p.p_sn + (int64_t)k * p.p_sk + c0, pv);
                else if (p.uid) ldv<VEC>(p.ptab + (int64_t)p.uid[i * p.uid_stride + k] * D + c0, pv);
                for (int q = 0; q < VEC; ++q) gt[k][q] = fmaf(ghv[q], a[q] + pv[q], gt[k][q]);
            }
        }
    }
    if (!want_gt) return;
    // block reduction over the sub-groups (same columns), then one slab row per block
    float* mine = red + (size_t)sg * p.K * D;
    if (col_ok) {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if (k >= p.K) break;
            for (int q = 0; q < VEC; ++q) mine[k * D + c0 + q] = gt[k][q];
        }
    }
    __syncthreads();
    float* out = p.slab + (size_t)blockIdx.x * p.K * D;
    for (int e = threadIdx.x; e < p.K * D; e += kBlock) {
        float tot = 0.f;
        for (int s2 = 0; s2 < NODES; ++s2) tot += red[(size_t)s2 * p.K * D + e];
        out[e] = tot;
    }
}

int cb_grid(int N, int G) {
    const int64_t tiles = ((int64_t)N + (kBlock / G) - 1) / (kBlock / G);
    int64_t g = (int64_t)device_facts().cu_count * 6;   // (the 26 KB block-reduction buffer allows 6 blocks per CU)
    if (g > tiles) g = tiles;
    return (int)(g < 1 ? 1 : g);
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

template <int VEC, int G>
int cb_launch(const CbParams& p, int grid, hipStream_t s) {
    const size_t lds = p.slab ? sizeof(float) * (size_t)(kBlock / G) * p.K * p.D : 0;
    if (p.K <= 8) {
        if (lds > 64 * 1024) KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)combine_bwd_kernel<VEC, G, 8>, lds));
        hipLaunchKernelGGL((combine_bwd_kernel<VEC, G, 8>), dim3(grid), dim3(kBlock), lds, s, p);
    } else {
        if (lds > 64 * 1024) KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)combine_bwd_kernel<VEC, G, 16>, lds));
        hipLaunchKernelGGL((combine_bwd_kernel<VEC, G, 16>), dim3(grid), dim3(kBlock), lds, s, p);
    }
    KPGNN_LAUNCH_CHECK("combine_bwd_kernel");
    return KPGNN_OK;
}

}  // namespace
}  // namespace kpgnn

using namespace kpgnn;

extern "C" size_t kpgnn_combine_bwd_workspace_bytes(int32_t N, int32_t K, int32_t D) {
    if (N <= 0 || K < 1 || D < 1) return 0;
    return sizeof(float) * (size_t)device_facts().cu_count * 6 * K * D;  // upper bound on grid * K * D (cb_grid)
}

extern "C" int kpgnn_combine_bwd(const kpgnn_combine_bwd_desc* d, kpgnn_stream_t stream) {
    KPGNN_REQUIRE(d != nullptr, "combine_bwd: NULL descriptor");
    KPGNN_REQUIRE(d->N >= 0 && d->K >= 1 && d->K <= 16 && d->D >= 1, "combine_bwd: bad N=%d K=%d D=%d (K <= 16)", d->N, d->K, d->D);
    if (d->N == 0) return KPGNN_OK;
    KPGNN_REQUIRE(d->pre && d->g, "combine_bwd: NULL pre/g");
    KPGNN_REQUIRE(d->theta ? d->gh != nullptr : d->gout != nullptr, "combine_bwd: need (theta, gh) or gout");
    KPGNN_REQUIRE(d->gtheta == nullptr || d->theta != nullptr, "combine_bwd: gtheta without theta");
    KPGNN_REQUIRE(d->mode >= KPGNN_MODE_GIN && d->mode <= KPGNN_MODE_SUM, "combine_bwd: unknown mode %d", d->mode);
    int vec = 1, g = 4;
    int rc = cb_shape(d, &vec, &g);
    if (rc != KPGNN_OK) return rc;
    CbParams p;
    p.N = d->N; p.K = d->K; p.D = d->D; p.mode = d->mode; p.pre = d->pre; p.gh = d->gh; p.theta = d->theta;
    p.gout = d->gout; p.go_sn = d->go_sn; p.go_sk = d->go_sk; p.periph = d->periph; p.p_sn = d->p_sn; p.p_sk = d->p_sk;
    p.ptab = d->periph ? nullptr : d->ptab; p.uid = d->periph ? nullptr : d->uid; p.uid_stride = d->uid_stride;
    p.g = d->g; p.gv = d->gv; p.slab = nullptr;
    const int grid = cb_grid(d->N, g);
    if (d->gtheta) {
        const size_t need = sizeof(float) * (size_t)grid * d->K * d->D;
        KPGNN_REQUIRE(d->workspace && d->workspace_bytes >= need, "combine_bwd: workspace too small (%zu < %zu)",
                      (size_t)d->workspace_bytes, need);
        const size_t lds = sizeof(float) * (size_t)(kBlock / g) * d->K * d->D;
        if (lds > 160 * 1024) return fail(KPGNN_ELIMIT, "combine_bwd: theta-gradient reduction needs %zu B of LDS", lds);
        p.slab = (float*)d->workspace;
    }
    hipStream_t s = (hipStream_t)stream;
#define KP_CB(V, GG) rc = cb_launch<V, GG>(p, grid, s); break
    switch (vec * 100 + g) {
        case 404: KP_CB(4, 4); case 408: KP_CB(4, 8); case 416: KP_CB(4, 16); case 432: KP_CB(4, 32); case 464: KP_CB(4, 64);
        case 204: KP_CB(2, 4); case 208: KP_CB(2, 8); case 216: KP_CB(2, 16); case 232: KP_CB(2, 32); case 264: KP_CB(2, 64);
        case 104: KP_CB(1, 4); case 108: KP_CB(1, 8); case 116: KP_CB(1, 16); case 132: KP_CB(1, 32); case 164: KP_CB(1, 64);
        default: return fail(KPGNN_EINVAL, "combine_bwd: no kernel for vec=%d g=%d", vec, g);
    }
#undef KP_CB
    if (rc != KPGNN_OK) return rc;
    if (p.slab) return slab_reduce(p.slab, grid, (int64_t)d->K * d->D, d->gtheta, (int64_t)d->K * d->D, nullptr, 0, nullptr, s);
    return KPGNN_OK;
}
