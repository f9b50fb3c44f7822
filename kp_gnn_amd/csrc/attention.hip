// Attention hop-combine (bi-LSTM scorer over the K hop slots) for gfx950.  Contract: include/kpgnn.h.
//
// Reference: layers/combine.py:17,22-27 (nn.LSTM(hidden_size, K, bidirectional) -> sum -> softmax -> weighted
// sum).  The recurrence is tiny (hidden size K <= 16, K steps) but strictly sequential per node: one thread
// owns one (node, direction); the 4K x K recurrent matrix is wave-uniform (scalar loads), h / c / the 4K gate
// pre-activations live in registers (K is a template parameter).  The [N*K, D] x [D, 8K] input projection runs as
// a library GEMM on the matrix cores before this kernel.
#include "kpgnn_common.h"

namespace kpgnn {
namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ float sigm(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_(float x) { return 2.0f * sigm(2.0f * x) - 1.0f; }

struct AtParams {
    int N, D;
    const float* x; int64_t x_sn, x_sk;
    const float* gin; const float* whh;
    float* acts; float* hsum; float* w; float* out;
    const float* gout; float* dx; float* ds; float* dgin; float* hprev;
};

// ---- forward recurrence: thread = (node n, direction blockIdx.y)
template <int K>
__global__ void __launch_bounds__(kBlock) attn_lstm_fwd_kernel(const AtParams p) {
    const int dir = blockIdx.y;
    const int64_t n = blockIdx.x * (int64_t)kBlock + threadIdx.x;
    if (n >= p.N) return;
    const float* whh = p.whh + dir * 4 * K * K;     // [4K][K], uniform
    float h[K], c[K];
#pragma unroll
    for (int q = 0; q < K; ++q) { h[q] = 0.f; c[q] = 0.f; }
#pragma unroll
    for (int s = 0; s < K; ++s) {
        const int t = dir == 0 ? s : K - 1 - s;
        const float* gi = p.gin + ((n * K + t) * 2 + dir) * (int64_t)(4 * K);
        float g[4 * K];
#pragma unroll
        for (int q = 0; q < 4 * K; ++q) g[q] = gi[q];
#pragma unroll
        for (int q = 0; q < 4 * K; ++q)
#pragma unroll
            for (int r = 0; r < K; ++r) g[q] = fmaf(whh[q * K + r], h[r], g[q]);
        float* a = p.acts + (((int64_t)dir * p.N + n) * K + t) * (5 * K);
        float hs = 0.f;
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const float ig = sigm(g[q]), fg = sigm(g[K + q]), gg = tanh_(g[2 * K + q]), og = sigm(g[3 * K + q]);
            c[q] = fmaf(fg, c[q], ig * gg);
            h[q] = og * tanh_(c[q]);
            hs += h[q];
            a[q] = ig; a[K + q] = fg; a[2 * K + q] = gg; a[3 * K + q] = og; a[4 * K + q] = c[q];
        }
        p.hsum[((int64_t)dir * p.N + n) * K + t] = hs;
    }
}

// ---- softmax over the slots + weighted sum: sub-group of G lanes per node, 16-B columns
template <int VEC> struct VT;
template <> struct VT<1> { using T = float; };
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

template <int G, int VEC>
__global__ void __launch_bounds__(kBlock) attn_apply_fwd_kernel(const AtParams p, int K) {
    const int sg = threadIdx.x / G, sl = threadIdx.x % G;
    const int c0 = sl * VEC;
    const bool col_ok = c0 < p.D;
    constexpr int NODES = kBlock / G;
    for (int64_t n = (int64_t)blockIdx.x * NODES + sg; n < p.N; n += (int64_t)gridDim.x * NODES) {
        float sc[16], m = -INFINITY, den = 0.f;
        for (int t = 0; t < K; ++t) { sc[t & 15] = p.hsum[n * K + t] + p.hsum[((int64_t)p.N + n) * K + t]; m = fmaxf(m, sc[t & 15]); }
        for (int t = 0; t < K; ++t) { sc[t & 15] = __expf(sc[t & 15] - m); den += sc[t & 15]; }
        const float inv = 1.0f / den;
        float acc[VEC];
        for (int q = 0; q < VEC; ++q) acc[q] = 0.f;
        for (int t = 0; t < K; ++t) {
            const float wt = sc[t & 15] * inv;
            if (sl == 0) p.w[n * K + t] = wt;
            if (col_ok) {
                float v[VEC];
                ldv<VEC>(p.x + n * p.x_sn + (int64_t)t * p.x_sk + c0, v);
                for (int q = 0; q < VEC; ++q) acc[q] = fmaf(wt, v[q], acc[q]);
            }
        }
        if (col_ok) stv<VEC>(p.out + n * p.D + c0, acc);
    }
}

// ---- backward of softmax + weighted sum: dw_t = <gout, x_t>, dx_direct = w_t gout, ds = w (dw - sum w dw)
template <int G, int VEC>
__global__ void __launch_bounds__(kBlock) attn_apply_bwd_kernel(const AtParams p, int K) {
    const int sg = threadIdx.x / G, sl = threadIdx.x % G;
    const int c0 = sl * VEC;
    const bool col_ok = c0 < p.D;
    constexpr int NODES = kBlock / G;
    for (int64_t n = (int64_t)blockIdx.x * NODES + sg; n < p.N; n += (int64_t)gridDim.x * NODES) {
        float go[VEC];
        for (int q = 0; q < VEC; ++q) go[q] = 0.f;
        if (col_ok) ldv<VEC>(p.gout + n * p.D + c0, go);
        float dw[16], wt[16], dot = 0.f;
        for (int t = 0; t < K; ++t) {
            float part = 0.f;
            wt[t & 15] = p.w[n * K + t];
            if (col_ok) {
                float v[VEC], d1[VEC];
                ldv<VEC>(p.x + n * p.x_sn + (int64_t)t * p.x_sk + c0, v);
                for (int q = 0; q < VEC; ++q) { part = fmaf(go[q], v[q], part); d1[q] = wt[t & 15] * go[q]; }
                stv<VEC>(p.dx + (n * K + t) * (int64_t)p.D + c0, d1);
            }
            for (int off = G / 2; off > 0; off >>= 1) part += __shfl_xor(part, off);   // stays inside the sub-group
            dw[t & 15] = part;
            dot = fmaf(wt[t & 15], part, dot);
        }
        if (sl == 0)
            for (int t = 0; t < K; ++t) p.ds[n * K + t] = wt[t & 15] * (dw[t & 15] - dot);
    }
}

// ---- BPTT: thread = (node, direction)
template <int K>
__global__ void __launch_bounds__(kBlock) attn_lstm_bwd_kernel(const AtParams p) {
    const int dir = blockIdx.y;
    const int64_t n = blockIdx.x * (int64_t)kBlock + threadIdx.x;
    if (n >= p.N) return;
    const float* whh = p.whh + dir * 4 * K * K;
    float dh[K], dc[K];
#pragma unroll
    for (int q = 0; q < K; ++q) { dh[q] = 0.f; dc[q] = 0.f; }
#pragma unroll
    for (int s = K - 1; s >= 0; --s) {              // reverse of the forward visiting order
        const int t = dir == 0 ? s : K - 1 - s;
        const int tp = dir == 0 ? t - 1 : t + 1;    // slot visited just before t (state h_{prev}, c_{prev})
        const float* a = p.acts + (((int64_t)dir * p.N + n) * K + t) * (5 * K);
        const float* ap = p.acts + (((int64_t)dir * p.N + n) * K + (s > 0 ? tp : t)) * (5 * K);
        const float dst = p.ds[n * K + t];
        float dg[4 * K];
        float* hp = p.hprev + ((n * K + t) * 2 + dir) * (int64_t)K;
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const float ig = a[q], fg = a[K + q], gg = a[2 * K + q], og = a[3 * K + q], ct = a[4 * K + q];
            const float cp = s > 0 ? ap[4 * K + q] : 0.f;
            const float hprev = s > 0 ? ap[3 * K + q] * tanh_(cp) : 0.f;
            hp[q] = hprev;
            const float tc = tanh_(ct);
            const float dhq = dh[q] + dst;
            const float dcq = fmaf(dhq * og, 1.0f - tc * tc, dc[q]);
            dg[q] = dcq * gg * ig * (1.0f - ig);
            dg[K + q] = dcq * cp * fg * (1.0f - fg);
            dg[2 * K + q] = dcq * ig * (1.0f - gg * gg);
            dg[3 * K + q] = dhq * tc * og * (1.0f - og);
            dc[q] = dcq * fg;
        }
        float* dgo = p.dgin + ((n * K + t) * 2 + dir) * (int64_t)(4 * K);
#pragma unroll
        for (int q = 0; q < 4 * K; ++q) dgo[q] = dg[q];
#pragma unroll
        for (int r = 0; r < K; ++r) {
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < 4 * K; ++q) acc = fmaf(whh[q * K + r], dg[q], acc);
            dh[r] = acc;
        }
    }
}



#define KP_K_SWITCH(KERNEL, GRID, BLOCK, S, P)                                                                   \
    switch (K) {                                                                                                 \
        case 1: hipLaunchKernelGGL(KERNEL<1>, GRID, BLOCK, 0, S, P); break;                                      \
        case 2: hipLaunchKernelGGL(KERNEL<2>, GRID, BLOCK, 0, S, P); break;                                      \
        case 3: hipLaunchKernelGGL(KERNEL<3>, GRID, BLOCK, 0, S, P); break;                                      \
        case 4: hipLaunchKernelGGL(KERNEL<4>, GRID, BLOCK, 0, S, P); break;                                      \
        case 5: hipLaunchKernelGGL(KERNEL<5>, GRID, BLOCK, 0, S, P); break;                                      \
        case 6: hipLaunchKernelGGL(KERNEL<6>, GRID, BLOCK, 0, S, P); break;                                      \
        case 7: hipLaunchKernelGGL(KERNEL<7>, GRID, BLOCK, 0, S, P); break;                                      \
        case 8: hipLaunchKernelGGL(KERNEL<8>, GRID, BLOCK, 0, S, P); break;                                      \
        case 9: hipLaunchKernelGGL(KERNEL<9>, GRID, BLOCK, 0, S, P); break;                                      \
        case 10: hipLaunchKernelGGL(KERNEL<10>, GRID, BLOCK, 0, S, P); break;                                    \
        case 11: hipLaunchKernelGGL(KERNEL<11>, GRID, BLOCK, 0, S, P); break;                                    \
        case 12: hipLaunchKernelGGL(KERNEL<12>, GRID, BLOCK, 0, S, P); break;                                    \
        case 13: hipLaunchKernelGGL(KERNEL<13>, GRID, BLOCK, 0, S, P); break;                                    \
        case 14: hipLaunchKernelGGL(KERNEL<14>, GRID, BLOCK, 0, S, P); break;                                    \
        case 15: hipLaunchKernelGGL(KERNEL<15>, GRID, BLOCK, 0, S, P); break;                                    \
        case 16: hipLaunchKernelGGL(KERNEL<16>, GRID, BLOCK, 0, S, P); break;                                    \
        default: return fail(KPGNN_ELIMIT, "attention combine: K=%d > 16", K);                                   \
    }

int apply_vec(const kpgnn_attn_desc* d) {
    const bool v4 = d->D % 4 == 0 && (d->x_sn % 4) == 0 && (d->x_sk % 4) == 0 && (((uintptr_t)d->x) & 15) == 0 &&
                    (((uintptr_t)d->out) & 15) == 0 && (((uintptr_t)d->gout) & 15) == 0 && (((uintptr_t)d->dx) & 15) == 0;
    return v4 ? 4 : 1;
}
int apply_group(int D, int vec) {
    int g = 4;
    while (g * vec < D) g <<= 1;
    return g;
}

int check(const kpgnn_attn_desc* d, bool bwd) {
    KPGNN_REQUIRE(d != nullptr, "attn: NULL descriptor");
    KPGNN_REQUIRE(d->N >= 0 && d->K >= 1 && d->D >= 1, "attn: bad N=%d K=%d D=%d", d->N, d->K, d->D);
    if (d->K > 16) return fail(KPGNN_ELIMIT, "attention combine: K=%d > 16", d->K);
    if (d->D > 256 || (d->D % 4 != 0 && d->D > 64)) return fail(KPGNN_ELIMIT, "attention combine: D=%d unsupported (<= 256, or <= 64 when not a multiple of 4)", d->D);
    KPGNN_REQUIRE(d->x && d->gin && d->whh && d->acts && d->hsum && d->w, "attn: NULL pointer");
    if (bwd) KPGNN_REQUIRE(d->gout && d->dx && d->ds && d->dgin && d->hprev, "attn_bwd: NULL pointer");
    else KPGNN_REQUIRE(d->out != nullptr, "attn_fwd: NULL out");
    return KPGNN_OK;
}

void fill(const kpgnn_attn_desc* d, AtParams* p) {
    p->N = d->N; p->D = d->D; p->x = d->x; p->x_sn = d->x_sn; p->x_sk = d->x_sk; p->gin = d->gin; p->whh = d->whh;
    p->acts = d->acts; p->hsum = d->hsum; p->w = d->w; p->out = d->out; p->gout = d->gout; p->dx = d->dx; p->ds = d->ds;
    p->dgin = d->dgin; p->hprev = d->hprev;
}

}  // namespace
}  // namespace kpgnn

using namespace kpgnn;

extern "C" int kpgnn_attn_fwd(const kpgnn_attn_desc* d, kpgnn_stream_t stream) {
    int rc = check(d, false);
    if (rc != KPGNN_OK) return rc;
    if (d->N == 0) return KPGNN_OK;
    AtParams p; fill(d, &p);
    hipStream_t s = (hipStream_t)stream;
    const int K = d->K;
    dim3 grid((unsigned)((d->N + kBlock - 1) / kBlock), 2), blk(kBlock);
    KP_K_SWITCH(attn_lstm_fwd_kernel, grid, blk, s, p)
    KPGNN_LAUNCH_CHECK("attn_lstm_fwd_kernel");
    const int vec = apply_vec(d);
    const int g = apply_group(d->D, vec);
    if (g > 64) return fail(KPGNN_ELIMIT, "attention combine: D=%d needs more than 64 lanes", d->D);
    int64_t nb = ((int64_t)d->N + (kBlock / g) - 1) / (kBlock / g);
    const int64_t cap = (int64_t)device_facts().cu_count * 8;
    if (nb > cap) nb = cap;
#define KP_AP(GG) do { if (vec == 4) hipLaunchKernelGGL((attn_apply_fwd_kernel<GG, 4>), dim3((unsigned)nb), blk, 0, s, p, K); \
                       else hipLaunchKernelGGL((attn_apply_fwd_kernel<GG, 1>), dim3((unsigned)nb), blk, 0, s, p, K); } while (0)
    switch (g) { case 4: KP_AP(4); break; case 8: KP_AP(8); break; case 16: KP_AP(16); break; case 32: KP_AP(32); break; default: KP_AP(64); break; }
#undef KP_AP
    KPGNN_LAUNCH_CHECK("attn_apply_fwd_kernel");
    return KPGNN_OK;
}

extern "C" int kpgnn_attn_bwd(const kpgnn_attn_desc* d, kpgnn_stream_t stream) {
    int rc = check(d, true);
    if (rc != KPGNN_OK) return rc;
    if (d->N == 0) return KPGNN_OK;
    AtParams p; fill(d, &p);
    hipStream_t s = (hipStream_t)stream;
    const int K = d->K;
    dim3 blk(kBlock);
    const int vec = apply_vec(d);
    const int g = apply_group(d->D, vec);
    if (g > 64) return fail(KPGNN_ELIMIT, "attention combine: D=%d needs more than 64 lanes", d->D);
    int64_t nb = ((int64_t)d->N + (kBlock / g) - 1) / (kBlock / g);
    const int64_t cap = (int64_t)device_facts().cu_count * 8;
    if (nb > cap) nb = cap;
#define KP_AP(GG) do { if (vec == 4) hipLaunchKernelGGL((attn_apply_bwd_kernel<GG, 4>), dim3((unsigned)nb), blk, 0, s, p, K); \
                       else hipLaunchKernelGGL((attn_apply_bwd_kernel<GG, 1>), dim3((unsigned)nb), blk, 0, s, p, K); } while (0)
    switch (g) { case 4: KP_AP(4); break; case 8: KP_AP(8); break; case 16: KP_AP(16); break; case 32: KP_AP(32); break; default: KP_AP(64); break; }
#undef KP_AP
    KPGNN_LAUNCH_CHECK("attn_apply_bwd_kernel");
    dim3 grid((unsigned)((d->N + kBlock - 1) / kBlock), 2);
    KP_K_SWITCH(attn_lstm_bwd_kernel, grid, blk, s, p)
    KPGNN_LAUNCH_CHECK("attn_lstm_bwd_kernel");
    return KPGNN_OK;
}
