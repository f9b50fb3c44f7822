// Graph readout: out[g,:] = sum (or mean) of the rows of graph g, and its backward (gfx950).
// Contract: include/kpgnn.h, kpgnn_segment_pool_fwd / _bwd.  Replaces PyG's global_add_pool / global_mean_pool
// (models/GraphRegression.py:46-51), which the framework runs as a zero fill + index_add_ with fp32 atomics (the order of
// the adds, hence the last bits of the loss, changed from run to run) and a gather in backward.  Collated batches keep
// the nodes of a graph contiguous, so a readout is a segmented sum: a sub-group of lanes owns one graph, lanes span the
// feature columns 16 B wide, rows are added in node order - bitwise reproducible, one launch per direction.
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

struct PoolParams {
    const int32_t* n_dyn;
    int64_t N; int G, D, mean;
    const int32_t* ptr; const int64_t* batch;
    const float* x; int64_t xs;
    float* out;
    const float* gout; float* gx; int64_t gxs;
};

template <int VEC, int L>
__global__ void __launch_bounds__(kBlock) pool_fwd_kernel(PoolParams p) {
    p.N = live_rows(p.N, p.n_dyn);
    const int sg = threadIdx.x / L, sl = threadIdx.x % L, c0 = sl * VEC;
    const int64_t g = (int64_t)blockIdx.x * (kBlock / L) + sg;
    if (g >= p.G || c0 >= p.D) return;
    const int beg = p.ptr[g], end = p.ptr[g + 1];
    float acc[VEC];
    for (int q = 0; q < VEC; ++q) acc[q] = 0.f;
    int r = beg;
    for (; r + 3 < end; r += 4) {             // four independent row loads per trip, added in row order
        float v[4][VEC];
#pragma unroll
        for (int u = 0; u < 4; ++u) ldv<VEC>(p.x + (int64_t)(r + u) * p.xs + c0, v[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            for (int q = 0; q < VEC; ++q) acc[q] += v[u][q];
    }
    for (; r < end; ++r) {
        float v[VEC];
        ldv<VEC>(p.x + (int64_t)r * p.xs + c0, v);
        for (int q = 0; q < VEC; ++q) acc[q] += v[q];
    }
    const float sc = (p.mean && end > beg) ? 1.0f / (float)(end - beg) : 1.0f;
    for (int q = 0; q < VEC; ++q) acc[q] *= sc;
    stv<VEC>(p.out + g * p.D + c0, acc);
}

template <int VEC, int L>
__global__ void __launch_bounds__(kBlock) pool_bwd_kernel(PoolParams p) {
    p.N = live_rows(p.N, p.n_dyn);
    const int sg = threadIdx.x / L, sl = threadIdx.x % L, c0 = sl * VEC;
    if (c0 >= p.D) return;
    for (int64_t n = (int64_t)blockIdx.x * (kBlock / L) + sg; n < p.N; n += (int64_t)gridDim.x * (kBlock / L)) {
        const int64_t g = p.batch[n];
        float sc = 1.0f;
        if (p.mean) { const int c = p.ptr[g + 1] - p.ptr[g]; sc = c > 0 ? 1.0f / (float)c : 1.0f; }
        float v[VEC];
        ldv<VEC>(p.gout + g * p.D + c0, v);
        for (int q = 0; q < VEC; ++q) v[q] *= sc;
        stv<VEC>(p.gx + n * p.gxs + c0, v);
    }
}

int shape(int D, std::initializer_list<const void*> ptrs, std::initializer_list<int64_t> strides, int* vec, int* lanes) {
    int v = (D % 4 == 0) ? 4 : (D % 2 == 0 ? 2 : 1);
    for (const void* q : ptrs) while (v > 1 && q && ((uintptr_t)q % (v * 4))) v >>= 1;
    for (int64_t s : strides) while (v > 1 && (s % v)) v >>= 1;
    const int need = (D + v - 1) / v;
    if (need > 256) return fail(KPGNN_ELIMIT, "segment_pool: D=%d too wide", D);
    int l = 4;
    while (l < need) l <<= 1;
    *vec = v; *lanes = l;
    return KPGNN_OK;
}

#define KP_POOL_SWITCH(KERNEL, GRID)                                                                         \
    switch (vec * 1000 + lanes) {                                                                            \
        case 4004: hipLaunchKernelGGL((KERNEL<4, 4>), dim3(GRID), dim3(kBlock), 0, s, p); break;             \
        case 4008: hipLaunchKernelGGL((KERNEL<4, 8>), dim3(GRID), dim3(kBlock), 0, s, p); break;             \
        case 4016: hipLaunchKernelGGL((KERNEL<4, 16>), dim3(GRID), dim3(kBlock), 0, s, p); break;            \
        case 4032: hipLaunchKernelGGL((KERNEL<4, 32>), dim3(GRID), dim3(kBlock), 0, s, p); break;            \
        case 4064: hipLaunchKernelGGL((KERNEL<4, 64>), dim3(GRID), dim3(kBlock), 0, s, p); break;            \
        case 2004: hipLaunchKernelGGL((KERNEL<2, 4>), dim3(GRID), dim3(kBlock), 0, s, p); break;             \
        case 2008: hipLaunchKernelGGL((KERNEL<2, 8>), dim3(GRID), dim3(kBlock), 0, s, p); break;             \
        case 2016: hipLaunchKernelGGL((KERNEL<2, 16>), dim3(GRID), dim3(kBlock), 0, s, p); break;            \
        case 2032: hipLaunchKernelGGL((KERNEL<2, 32>), dim3(GRID), dim3(kBlock), 0, s, p); break;            \
        case 2064: hipLaunchKernelGGL((KERNEL<2, 64>), dim3(GRID), dim3(kBlock), 0, s, p); break;            \
        case 2128: hipLaunchKernelGGL((KERNEL<2, 128>), dim3(GRID), dim3(kBlock), 0, s, p); break;           \
        case 1004: hipLaunchKernelGGL((KERNEL<1, 4>), dim3(GRID), dim3(kBlock), 0, s, p); break;             \
        case 1008: hipLaunchKernelGGL((KERNEL<1, 8>), dim3(GRID), dim3(kBlock), 0, s, p); break;             \
        case 1016: hipLaunchKernelGGL((KERNEL<1, 16>), dim3(GRID), dim3(kBlock), 0, s, p); break;            \
        case 1032: hipLaunchKernelGGL((KERNEL<1, 32>), dim3(GRID), dim3(kBlock), 0, s, p); break;            \
        case 1064: hipLaunchKernelGGL((KERNEL<1, 64>), dim3(GRID), dim3(kBlock), 0, s, p); break;            \
        case 1128: hipLaunchKernelGGL((KERNEL<1, 128>), dim3(GRID), dim3(kBlock), 0, s, p); break;           \
        case 1256: hipLaunchKernelGGL((KERNEL<1, 256>), dim3(GRID), dim3(kBlock), 0, s, p); break;           \
        default: return fail(KPGNN_EINVAL, "segment_pool: no kernel for vec=%d lanes=%d", vec, lanes);       \
    }                                                                                                        \
    KPGNN_LAUNCH_CHECK(#KERNEL)

int check(const kpgnn_pool_desc* d, const char* who) {
    KPGNN_REQUIRE(d != nullptr, "%s: NULL descriptor", who);
    KPGNN_REQUIRE(d->N >= 0 && d->G >= 0 && d->D >= 1 && d->D <= 1024, "%s: bad N=%lld G=%d D=%d", who, (long long)d->N, d->G, d->D);
    KPGNN_REQUIRE(d->mode == 0 || d->mode == 1, "%s: mode must be 0 (sum) or 1 (mean)", who);
    KPGNN_REQUIRE(d->G == 0 || d->graph_ptr, "%s: NULL graph_ptr", who);
    return KPGNN_OK;
}

}  // namespace
}  // namespace kpgnn

using namespace kpgnn;

extern "C" int kpgnn_segment_pool_fwd(const kpgnn_pool_desc* d, kpgnn_stream_t stream) {
    int rc = check(d, "segment_pool_fwd");
    if (rc != KPGNN_OK) return rc;
    if (d->G == 0) return KPGNN_OK;
    KPGNN_REQUIRE(d->out && (d->N == 0 || (d->x && d->x_stride >= d->D)), "segment_pool_fwd: NULL x/out or bad stride");
    int vec, lanes;
    rc = shape(d->D, {d->x, d->out}, {d->x_stride}, &vec, &lanes);
    if (rc != KPGNN_OK) return rc;
    PoolParams p = {};
    p.N = d->N; p.n_dyn = d->n_dyn; p.G = d->G; p.D = d->D; p.mean = d->mode; p.ptr = d->graph_ptr; p.x = d->x; p.xs = d->x_stride; p.out = d->out;
    hipStream_t s = (hipStream_t)stream;
    const unsigned grid = (unsigned)((d->G + (kBlock / lanes) - 1) / (kBlock / lanes));
    KP_POOL_SWITCH(pool_fwd_kernel, grid);
    return KPGNN_OK;
}

extern "C" int kpgnn_segment_pool_bwd(const kpgnn_pool_desc* d, kpgnn_stream_t stream) {
    int rc = check(d, "segment_pool_bwd");
    if (rc != KPGNN_OK) return rc;
    if (d->N == 0) return KPGNN_OK;
    KPGNN_REQUIRE(d->batch && d->gout && d->gx && d->gx_stride >= d->D, "segment_pool_bwd: NULL batch/gout/gx or bad stride");
    int vec, lanes;
    rc = shape(d->D, {d->gout, d->gx}, {d->gx_stride}, &vec, &lanes);
    if (rc != KPGNN_OK) return rc;
    PoolParams p = {};
    p.N = d->N; p.n_dyn = d->n_dyn; p.G = d->G; p.D = d->D; p.mean = d->mode; p.ptr = d->graph_ptr; p.batch = d->batch;
    p.gout = d->gout; p.gx = d->gx; p.gxs = d->gx_stride;
    hipStream_t s = (hipStream_t)stream;
    const int rows = kBlock / lanes;
    int64_t g = (d->N + rows * 4 - 1) / (rows * 4);
    const int64_t cap = (int64_t)device_facts().cu_count * 8;
    const unsigned grid = (unsigned)(g > cap ? cap : (g < 1 ? 1 : g));
    KP_POOL_SWITCH(pool_bwd_kernel, grid);
    return KPGNN_OK;
}
