// Training-mode BatchNorm1d (+ReLU, +residual) over [N, C] rows for gfx950.  Contract: include/kpgnn.h.
// HBM-bound streaming: stats pass (read x) -> ordered slab reduce -> apply pass (read x, write z); backward the
// same shape (reduce pass reads x, dz; apply pass reads x, dz, writes dx).  A sub-group of G lanes spans the C
// columns 16 B wide; sub-groups stride over rows; per-thread partials are register-resident.
#include <initializer_list>

#include "kpgnn_common.h"

namespace kpgnn {
namespace {

constexpr int kBlock = 256;
constexpr int kStatBlocks = 512;   // partial-sum slabs

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

struct BnParams {
    int64_t N; int C, relu; float eps, momentum;
    const float* x; int64_t xs;
    const float* dz; int64_t dzs;
    const float* gamma; const float* beta;
    float* rmean; float* rvar;
    float* mean; float* invstd;        // fwd: outputs; bwd: inputs
    float* z; int64_t zs;              // fwd output / bwd dx
    const float* res; int64_t rs;
    float* slab;                       // [gridDim.x][2][C]
    float* sums;                       // [2][C] reduced
    int64_t* nbt;                      // num_batches_tracked or NULL
    float* dgamma; float* dbeta;
};

// Block reduction of per-thread (a,b)[VEC] over the row-lanes, then one slab row per block.
// (Letting the last block to finish add the slab up in-kernel was tried: one block pulling 512 x 2C partials through
//  L2 takes far longer than the ~4.6 us launch of the 2C/16-block ordered slab_reduce it would save.)
template <int VEC, int G>
__device__ __forceinline__ void block_to_slab(const BnParams& p, float (&a)[VEC], float (&b)[VEC], int c0, bool col_ok) {
    __shared__ float red[2][kBlock / G][G * VEC];
    const int rl = threadIdx.x / G, sl = threadIdx.x % G;
    for (int q = 0; q < VEC; ++q) { red[0][rl][sl * VEC + q] = a[q]; red[1][rl][sl * VEC + q] = b[q]; }
    __syncthreads();
    if (rl == 0 && col_ok) {
        for (int q = 0; q < VEC; ++q) {
            float s0 = 0.f, s1 = 0.f;
            for (int r = 0; r < kBlock / G; ++r) { s0 += red[0][r][sl * VEC + q]; s1 += red[1][r][sl * VEC + q]; }
            p.slab[((int64_t)blockIdx.x * 2 + 0) * p.C + c0 + q] = s0;
            p.slab[((int64_t)blockIdx.x * 2 + 1) * p.C + c0 + q] = s1;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && p.nbt) *p.nbt += 1;   // nn.BatchNorm1d.num_batches_tracked
}

// fwd stats: s0 = sum (x - pivot), s1 = sum (x - pivot)^2, pivot = x[0, c]
template <int VEC, int G>
__global__ void __launch_bounds__(kBlock) bn_stats_kernel(const BnParams p) {
    const int rl = threadIdx.x / G, sl = threadIdx.x % G, c0 = sl * VEC;
    const bool col_ok = c0 < p.C;
    float a[VEC], b[VEC], piv[VEC];
    for (int q = 0; q < VEC; ++q) { a[q] = 0.f; b[q] = 0.f; piv[q] = 0.f; }
    if (col_ok) {
        ldv<VEC>(p.x + c0, piv);
        // four rows per trip: the loads are independent, a one-row loop keeps a single request in flight per thread
        // (9 us for 20 MB)
        const int64_t step = (int64_t)gridDim.x * (kBlock / G);
        int64_t r = (int64_t)blockIdx.x * (kBlock / G) + rl;
        for (; r + 3 * step < p.N; r += 4 * step) {
            float v0[VEC], v1[VEC], v2[VEC], v3[VEC];
            ldv<VEC>(p.x + r * p.xs + c0, v0);
            ldv<VEC>(p.x + (r + step) * p.xs + c0, v1);
            ldv<VEC>(p.x + (r + 2 * step) * p.xs + c0, v2);
            ldv<VEC>(p.x + (r + 3 * step) * p.xs + c0, v3);
            for (int q = 0; q < VEC; ++q) {
                const float d0 = v0[q] - piv[q], d1 = v1[q] - piv[q], d2 = v2[q] - piv[q], d3 = v3[q] - piv[q];
                a[q] += d0; b[q] = fmaf(d0, d0, b[q]);
                a[q] += d1; b[q] = fmaf(d1, d1, b[q]);
                a[q] += d2; b[q] = fmaf(d2, d2, b[q]);
                a[q] += d3; b[q] = fmaf(d3, d3, b[q]);
            }
        }
        for (; r < p.N; r += step) {
            float v[VEC];
            ldv<VEC>(p.x + r * p.xs + c0, v);
            for (int q = 0; q < VEC; ++q) { const float d = v[q] - piv[q]; a[q] += d; b[q] = fmaf(d, d, b[q]); }
        }
    }
    block_to_slab<VEC, G>(p, a, b, c0, col_ok);
}

// fwd apply (every block recomputes mean / invstd of its columns from the reduced sums; block 0 publishes
// them and updates the running statistics)
template <int VEC, int G>
__global__ void __launch_bounds__(kBlock) bn_apply_kernel(const BnParams p) {
    const int rl = threadIdx.x / G, sl = threadIdx.x % G, c0 = sl * VEC;
    if (c0 >= p.C) return;
    float piv[VEC], mean[VEC], istd[VEC], g[VEC], bt[VEC];
    ldv<VEC>(p.x + c0, piv);
    ldv<VEC>(p.gamma + c0, g);
    ldv<VEC>(p.beta + c0, bt);
    const double inv_n = 1.0 / (double)p.N;
    for (int q = 0; q < VEC; ++q) {
        const double m1 = (double)p.sums[c0 + q] * inv_n;
        double var = (double)p.sums[p.C + c0 + q] * inv_n - m1 * m1;
        if (var < 0.0) var = 0.0;
        mean[q] = (float)((double)piv[q] + m1);
        istd[q] = (float)(1.0 / sqrt(var + (double)p.eps));
        if (blockIdx.x == 0 && rl == 0) {
            p.mean[c0 + q] = mean[q];
            p.invstd[c0 + q] = istd[q];
            if (p.rmean) {
                const double unb = p.N > 1 ? var * (double)p.N / (double)(p.N - 1) : var;
                p.rmean[c0 + q] = (1.f - p.momentum) * p.rmean[c0 + q] + p.momentum * mean[q];
                p.rvar[c0 + q] = (1.f - p.momentum) * p.rvar[c0 + q] + p.momentum * (float)unb;
            }
        }
    }
    for (int64_t r = (int64_t)blockIdx.x * (kBlock / G) + rl; r < p.N; r += (int64_t)gridDim.x * (kBlock / G)) {
        float v[VEC], o[VEC];
        ldv<VEC>(p.x + r * p.xs + c0, v);
        for (int q = 0; q < VEC; ++q) {
            o[q] = fmaf((v[q] - mean[q]) * istd[q], g[q], bt[q]);
            if (p.relu) o[q] = fmaxf(o[q], 0.f);
        }
        if (p.res) {
            float rr[VEC];
            ldv<VEC>(p.res + r * p.rs + c0, rr);
            for (int q = 0; q < VEC; ++q) o[q] += rr[q];
        }
        stv<VEC>(p.z + r * p.zs + c0, o);
    }
}

// bwd reduce: s0 = sum dy, s1 = sum dy * xhat  (dy = dz masked by the recomputed pre-activation when relu)
template <int VEC, int G>
__global__ void __launch_bounds__(kBlock) bn_bwd_reduce_kernel(const BnParams p) {
    const int rl = threadIdx.x / G, sl = threadIdx.x % G, c0 = sl * VEC;
    const bool col_ok = c0 < p.C;
    float a[VEC], b[VEC];
    for (int q = 0; q < VEC; ++q) { a[q] = 0.f; b[q] = 0.f; }
    if (col_ok) {
        float mean[VEC], istd[VEC], g[VEC], bt[VEC];
        ldv<VEC>(p.mean + c0, mean); ldv<VEC>(p.invstd + c0, istd); ldv<VEC>(p.gamma + c0, g); ldv<VEC>(p.beta + c0, bt);
        const int64_t step = (int64_t)gridDim.x * (kBlock / G);
        int64_t r = (int64_t)blockIdx.x * (kBlock / G) + rl;
        auto one = [&](const float (&v)[VEC], float (&dy)[VEC]) {
            for (int q = 0; q < VEC; ++q) {
                const float xh = (v[q] - mean[q]) * istd[q];
                if (p.relu && fmaf(xh, g[q], bt[q]) <= 0.f) dy[q] = 0.f;
                a[q] += dy[q];
                b[q] = fmaf(dy[q], xh, b[q]);
            }
        };
        for (; r + step < p.N; r += 2 * step) {          // two rows (four independent loads) per trip
            float v0[VEC], y0[VEC], v1[VEC], y1[VEC];
            ldv<VEC>(p.x + r * p.xs + c0, v0);
            ldv<VEC>(p.dz + r * p.dzs + c0, y0);
            ldv<VEC>(p.x + (r + step) * p.xs + c0, v1);
            ldv<VEC>(p.dz + (r + step) * p.dzs + c0, y1);
            one(v0, y0);
            one(v1, y1);
        }
        for (; r < p.N; r += step) {
            float v[VEC], dy[VEC];
            ldv<VEC>(p.x + r * p.xs + c0, v);
            ldv<VEC>(p.dz + r * p.dzs + c0, dy);
            one(v, dy);
        }
    }
    block_to_slab<VEC, G>(p, a, b, c0, col_ok);
}

template <int VEC, int G>
__global__ void __launch_bounds__(kBlock) bn_bwd_apply_kernel(const BnParams p) {
    const int rl = threadIdx.x / G, sl = threadIdx.x % G, c0 = sl * VEC;
    if (c0 >= p.C) return;
    float mean[VEC], istd[VEC], g[VEC], bt[VEC], s0[VEC], s1[VEC];
    ldv<VEC>(p.mean + c0, mean); ldv<VEC>(p.invstd + c0, istd); ldv<VEC>(p.gamma + c0, g); ldv<VEC>(p.beta + c0, bt);
    ldv<VEC>(p.sums + c0, s0); ldv<VEC>(p.sums + p.C + c0, s1);
    if (blockIdx.x == 0 && rl == 0) { stv<VEC>(p.dbeta + c0, s0); stv<VEC>(p.dgamma + c0, s1); }
    const float inv_n = 1.0f / (float)p.N;
    for (int64_t r = (int64_t)blockIdx.x * (kBlock / G) + rl; r < p.N; r += (int64_t)gridDim.x * (kBlock / G)) {
        float v[VEC], dy[VEC], o[VEC];
        ldv<VEC>(p.x + r * p.xs + c0, v);
        ldv<VEC>(p.dz + r * p.dzs + c0, dy);
        for (int q = 0; q < VEC; ++q) {
            const float xh = (v[q] - mean[q]) * istd[q];
            if (p.relu && fmaf(xh, g[q], bt[q]) <= 0.f) dy[q] = 0.f;
            o[q] = g[q] * istd[q] * (dy[q] - s0[q] * inv_n - xh * s1[q] * inv_n);
        }
        stv<VEC>(p.z + r * p.zs + c0, o);
    }
}

int bn_shape(int C, std::initializer_list<const void*> ptrs, std::initializer_list<int64_t> strides, int* vec, int* g) {
    int v = (C % 4 == 0) ? 4 : (C % 2 == 0 ? 2 : 1);
    for (const void* q : ptrs) while (v > 1 && q && ((uintptr_t)q % (v * 4))) v >>= 1;
    for (int64_t s : strides) while (v > 1 && (s % v)) v >>= 1;
    const int lanes = (C + v - 1) / v;
    if (lanes > 64) return fail(KPGNN_ELIMIT, "batch norm: C=%d needs %d lanes > 64 (C <= 256)", C, lanes);
    int gg = 4;
    while (gg < lanes) gg <<= 1;
    *vec = v; *g = gg;
    return KPGNN_OK;
}

int stream_grid(int64_t N, int G) {
    const int64_t rows_per_block = kBlock / G;
    int64_t g = (N + rows_per_block * 4 - 1) / (rows_per_block * 4);   // >= 4 rows per thread
    const int64_t cap = (int64_t)device_facts().cu_count * 8;
    if (g > cap) g = cap;
    return (int)(g < 1 ? 1 : g);
}

#define KP_BN_SWITCH(KERNEL, GRID)                                                                                   \
    switch (vec * 100 + g) {                                                                                         \
        case 404: hipLaunchKernelGGL((KERNEL<4, 4>), dim3(GRID), dim3(kBlock), 0, s, p); break;                      \
        case 408: hipLaunchKernelGGL((KERNEL<4, 8>), dim3(GRID), dim3(kBlock), 0, s, p); break;                      \
        case 416: hipLaunchKernelGGL((KERNEL<4, 16>), dim3(GRID), dim3(kBlock), 0, s, p); break;                     \
        case 432: hipLaunchKernelGGL((KERNEL<4, 32>), dim3(GRID), dim3(kBlock), 0, s, p); break;                     \
        case 464: hipLaunchKernelGGL((KERNEL<4, 64>), dim3(GRID), dim3(kBlock), 0, s, p); break;                     \
        case 204: hipLaunchKernelGGL((KERNEL<2, 4>), dim3(GRID), dim3(kBlock), 0, s, p); break;                      \
        case 208: hipLaunchKernelGGL((KERNEL<2, 8>), dim3(GRID), dim3(kBlock), 0, s, p); break;                      \
        case 216: hipLaunchKernelGGL((KERNEL<2, 16>), dim3(GRID), dim3(kBlock), 0, s, p); break;                     \
        case 232: hipLaunchKernelGGL((KERNEL<2, 32>), dim3(GRID), dim3(kBlock), 0, s, p); break;                     \
        case 264: hipLaunchKernelGGL((KERNEL<2, 64>), dim3(GRID), dim3(kBlock), 0, s, p); break;                     \
        case 104: hipLaunchKernelGGL((KERNEL<1, 4>), dim3(GRID), dim3(kBlock), 0, s, p); break;                      \
        case 108: hipLaunchKernelGGL((KERNEL<1, 8>), dim3(GRID), dim3(kBlock), 0, s, p); break;                      \
        case 116: hipLaunchKernelGGL((KERNEL<1, 16>), dim3(GRID), dim3(kBlock), 0, s, p); break;                     \
        case 132: hipLaunchKernelGGL((KERNEL<1, 32>), dim3(GRID), dim3(kBlock), 0, s, p); break;                     \
        case 164: hipLaunchKernelGGL((KERNEL<1, 64>), dim3(GRID), dim3(kBlock), 0, s, p); break;                     \
        default: return fail(KPGNN_EINVAL, "batch norm: no kernel for vec=%d g=%d", vec, g);                         \
    }                                                                                                                \
    KPGNN_LAUNCH_CHECK(#KERNEL)

}  // namespace
}  // namespace kpgnn

using namespace kpgnn;

extern "C" size_t kpgnn_bn_workspace_bytes(int32_t C) {
    if (C < 1) return 0;
    return sizeof(float) * (size_t)(kStatBlocks + 1) * 2 * C;  // slabs + reduced sums
}

extern "C" int kpgnn_bn_fwd(const kpgnn_bn_desc* d, kpgnn_stream_t stream) {
    KPGNN_REQUIRE(d != nullptr, "bn_fwd: NULL descriptor");
    KPGNN_REQUIRE(d->N >= 1 && d->C >= 1, "bn_fwd: bad N=%lld C=%d", (long long)d->N, d->C);
    KPGNN_REQUIRE(d->x && d->gamma && d->beta && d->mean && d->invstd && d->z, "bn_fwd: NULL pointer");
    KPGNN_REQUIRE(d->x_stride >= d->C && d->z_stride >= d->C, "bn_fwd: bad strides");
    KPGNN_REQUIRE(d->workspace && d->workspace_bytes >= kpgnn_bn_workspace_bytes(d->C), "bn_fwd: workspace too small");
    int vec, g;
    int rc = bn_shape(d->C, {d->x, d->z, d->gamma, d->beta, d->residual}, {d->x_stride, d->z_stride, d->residual ? d->r_stride : 0}, &vec, &g);
    if (rc != KPGNN_OK) return rc;
    BnParams p = {};
    p.N = d->N; p.C = d->C; p.relu = d->relu; p.eps = d->eps; p.momentum = d->momentum;
    p.x = d->x; p.xs = d->x_stride; p.gamma = d->gamma; p.beta = d->beta; p.rmean = d->running_mean; p.rvar = d->running_var;
    p.mean = d->mean; p.invstd = d->invstd; p.z = d->z; p.zs = d->z_stride; p.res = d->residual; p.rs = d->r_stride;
    p.slab = (float*)d->workspace;
    float* sums = p.slab + (size_t)kStatBlocks * 2 * d->C;
    p.sums = sums;
    p.nbt = d->num_batches_tracked;
    hipStream_t s = (hipStream_t)stream;
    int nstat = stream_grid(d->N, g);
    if (nstat > kStatBlocks) nstat = kStatBlocks;
    KP_BN_SWITCH(bn_stats_kernel, nstat);
    rc = slab_reduce(p.slab, nstat, (int64_t)2 * d->C, sums, (int64_t)2 * d->C, nullptr, 0, nullptr, s);
    if (rc != KPGNN_OK) return rc;
    const int napply = stream_grid(d->N, g);
    KP_BN_SWITCH(bn_apply_kernel, napply);
    return KPGNN_OK;
}

extern "C" int kpgnn_bn_bwd(const kpgnn_bn_bwd_desc* d, kpgnn_stream_t stream) {
    KPGNN_REQUIRE(d != nullptr, "bn_bwd: NULL descriptor");
    KPGNN_REQUIRE(d->N >= 1 && d->C >= 1, "bn_bwd: bad N=%lld C=%d", (long long)d->N, d->C);
    KPGNN_REQUIRE(d->x && d->dz && d->gamma && d->beta && d->mean && d->invstd && d->dx && d->dgamma && d->dbeta, "bn_bwd: NULL pointer");
    KPGNN_REQUIRE(d->workspace && d->workspace_bytes >= kpgnn_bn_workspace_bytes(d->C), "bn_bwd: workspace too small");
    int vec, g;
    int rc = bn_shape(d->C, {d->x, d->dz, d->dx, d->gamma, d->beta, d->mean, d->invstd, d->dgamma, d->dbeta},
                      {d->x_stride, d->dz_stride, d->dx_stride}, &vec, &g);
    if (rc != KPGNN_OK) return rc;
    BnParams p = {};
    p.N = d->N; p.C = d->C; p.relu = d->relu;
    p.x = d->x; p.xs = d->x_stride; p.dz = d->dz; p.dzs = d->dz_stride; p.gamma = d->gamma; p.beta = d->beta;
    p.mean = const_cast<float*>(d->mean); p.invstd = const_cast<float*>(d->invstd);
    p.z = d->dx; p.zs = d->dx_stride; p.dgamma = d->dgamma; p.dbeta = d->dbeta;
    p.slab = (float*)d->workspace;
    float* sums = p.slab + (size_t)kStatBlocks * 2 * d->C;
    p.sums = sums;
    p.nbt = nullptr;
    hipStream_t s = (hipStream_t)stream;
    int nstat = stream_grid(d->N, g);
    if (nstat > kStatBlocks) nstat = kStatBlocks;
    KP_BN_SWITCH(bn_bwd_reduce_kernel, nstat);
    rc = slab_reduce(p.slab, nstat, (int64_t)2 * d->C, sums, (int64_t)2 * d->C, nullptr, 0, nullptr, s);
    if (rc != KPGNN_OK) return rc;
    const int napply = stream_grid(d->N, g);
    KP_BN_SWITCH(bn_bwd_apply_kernel, napply);
    return KPGNN_OK;
}
