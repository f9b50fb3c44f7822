// Attention hop-combine (bi-LSTM scorer over the K hop slots) for gfx950.  Contract: include/kpgnn.h.
//
// Reference: layers/combine.py:17,22-27 (nn.LSTM(hidden_size, K, bidirectional) -> sum -> softmax -> weighted
// sum).  The recurrence is tiny (hidden size K <= 16, K steps) but strictly sequential per node: one thread
// owns one (node, direction); the 4K x K recurrent matrix is wave-uniform (scalar loads), h / c / the 4K gate
// pre-activations live in registers (K is a template parameter).  The [N*K, D] x [D, 8K] input projection runs as
// a library GEMM on the matrix cores before this kernel.
#include "bf3.h"
#include "kpgnn_common.h"

namespace kpgnn {
namespace {

constexpr int kBlock = 256;
using f32x16 = __attribute__((ext_vector_type(16))) float;

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

// (KMAX: the slots' rows of a node are requested together, before the softmax arithmetic - a load per loop iteration made the
//  weighted sum a chain of K round trips per node)
template <int G, int VEC, int KMAX>
__global__ void __launch_bounds__(kBlock) attn_apply_fwd_kernel(const AtParams p, int K) {
    const int sg = threadIdx.x / G, sl = threadIdx.x % G;
    const int c0 = sl * VEC;
    const bool col_ok = c0 < p.D;
    const int cc = col_ok ? c0 : 0;
    constexpr int NODES = kBlock / G;
    for (int64_t n = (int64_t)blockIdx.x * NODES + sg; n < p.N; n += (int64_t)gridDim.x * NODES) {
        float xv[KMAX][VEC];
#pragma unroll
        for (int t = 0; t < KMAX; ++t) ldv<VEC>(p.x + n * p.x_sn + (int64_t)(t < K ? t : 0) * p.x_sk + cc, xv[t]);
        float sc[KMAX], m = -INFINITY, den = 0.f;
#pragma unroll
        for (int t = 0; t < KMAX; ++t) {
            const int tt = t < K ? t : 0;
            sc[t] = p.hsum[n * K + tt] + p.hsum[((int64_t)p.N + n) * K + tt];
            if (t < K) m = fmaxf(m, sc[t]);
        }
#pragma unroll
        for (int t = 0; t < KMAX; ++t) { sc[t] = t < K ? __expf(sc[t] - m) : 0.f; den += sc[t]; }
        const float inv = 1.0f / den;
        float acc[VEC];
        for (int q = 0; q < VEC; ++q) acc[q] = 0.f;
#pragma unroll
        for (int t = 0; t < KMAX; ++t) {
            if (t < K) {
                const float wt = sc[t] * inv;
                if (sl == 0) p.w[n * K + t] = wt;
                for (int q = 0; q < VEC; ++q) acc[q] = fmaf(wt, xv[t][q], acc[q]);
            }
        }
        if (col_ok) stv<VEC>(p.out + n * p.D + c0, acc);
    }
}

// ---- backward of softmax + weighted sum: dw_t = <gout, x_t>, dx_direct = w_t gout, ds = w (dw - sum w dw)
template <int G, int VEC, int KMAX>
__global__ void __launch_bounds__(kBlock) attn_apply_bwd_kernel(const AtParams p, int K) {
    const int sg = threadIdx.x / G, sl = threadIdx.x % G;
    const int c0 = sl * VEC;
    const bool col_ok = c0 < p.D;
    const int cc = col_ok ? c0 : 0;
    constexpr int NODES = kBlock / G;
    for (int64_t n = (int64_t)blockIdx.x * NODES + sg; n < p.N; n += (int64_t)gridDim.x * NODES) {
        float xv[KMAX][VEC];
#pragma unroll
        for (int t = 0; t < KMAX; ++t) ldv<VEC>(p.x + n * p.x_sn + (int64_t)(t < K ? t : 0) * p.x_sk + cc, xv[t]);
        float go[VEC];
        for (int q = 0; q < VEC; ++q) go[q] = 0.f;
        if (col_ok) ldv<VEC>(p.gout + n * p.D + c0, go);
        float dw[KMAX], wt[KMAX], dot = 0.f;
#pragma unroll
        for (int t = 0; t < KMAX; ++t) wt[t] = p.w[n * K + (t < K ? t : 0)];
#pragma unroll
        for (int t = 0; t < KMAX; ++t) {
            float part = 0.f;
            if (t < K && col_ok) {
                float d1[VEC];
                for (int q = 0; q < VEC; ++q) { part = fmaf(go[q], xv[t][q], part); d1[q] = wt[t] * go[q]; }
                if (p.dx) stv<VEC>(p.dx + (n * K + t) * (int64_t)p.D + c0, d1);    // (the scan form adds it in its dX product)
            }
            for (int off = G / 2; off > 0; off >>= 1) part += __shfl_xor(part, off);   // stays inside the sub-group
            dw[t] = part;
            dot = t < K ? fmaf(wt[t], part, dot) : dot;
        }
        if (sl == 0) {
#pragma unroll
            for (int t = 0; t < KMAX; ++t) if (t < K) p.ds[n * K + t] = wt[t] * (dw[t] - dot);
        }
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



// ------------------------------------------------------------------------------------------------
// The scan form (hidden size K <= 8, D % 4 == 0): input projection, recurrence and BPTT on the fp32 matrix
// instruction, no gin tensor, no per-thread rows.
//
// A wave owns (32 nodes, one direction).  With the hidden size padded to 8 units the 4 x 8 = 32 gate rows of a
// direction are ONE 32-row MFMA tile in the parameter's own order (i, f, g, o blocks of 8), and the products are
// taken transposed - gates x nodes = W (32 x D) . x_t^T (D x 32) - so that lane l ends up with node l & 31 and,
// in accumulator register r, gate r / 4 of unit 4 (l >> 5) + r % 4: the four gates of four units of its node, which
// is what the recurrence needs.  The recurrence's three products exchange nothing between lanes:
//   * B operand of the input product = 8 consecutive floats of the lane's node row, straight from global memory
//     (lane (node, half) feeds columns 16 ks + 8 half + e of k-step ks), split in registers into three bf16 pieces;
//     A = the weight row of gate l & 31, same columns, split once per block into LDS fragments; six products per k-step on
//     v_mfma_f32_32x32x16_bf16 give the fp32 product (bf3.h).
//   * the recurrent product W_hh h_{t-1} is 4 more instructions on the same accumulators: the B operand of the
//     i-th one is the lane's OWN h of unit 4 half + i.
//   * BPTT's dh_{t-1} = W_hh^T dg is 16 instructions whose B operands are the lane's own 16 dg values; rows 0..7
//     of the result land in accumulator registers 0..3 as the lane's own four units.
// Padded units (>= K) have zero weights and biases: c = h = 0 for them, and their dg is masked.  The activations
// saved for BPTT are written lane-contiguous ([tile][dir][t][20][64], 256 B per store instruction); dgin leaves in the
// padded layout [N*K, 64] (16-B stores) that the weight-gradient and dX products read.  The one product that contracts over
// NODES - dW_hh = sum dg h_prev^T - is taken inside the BPTT walk from operands a wave transposes through LDS.
struct ScanParams {
    int N, K, D;
    const float* x; int64_t x_sn, x_sk;
    const float* w_ih[2]; const float* w_hh[2]; const float* b_ih[2]; const float* b_hh[2];
    float* acts; float* hsum; float* w_pad;
    const float* ds; float* dgin; float* whh_slab;
};

constexpr int kScanThreads = 256;                    // 2 node tiles x 2 directions

template <int KS>                                     // KS >= ceil(D / 16) k-steps of the bf16 instruction
__global__ void __launch_bounds__(kScanThreads) attn_scan_fwd_kernel(const ScanParams p) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int dir = wv & 1;
    const int tile = blockIdx.x * 2 + (wv >> 1);
    const int ntiles = (p.N + 31) >> 5;
    const int K = p.K, D = p.D;
    const int half = lane >> 5;
    const int g = lane & 31, gtype = g >> 3, gunit = g & 7;
    const bool grow_ok = gunit < K;
    const int grow = gtype * K + (grow_ok ? gunit : 0);
    const float* wih_p = p.w_ih[dir] + (int64_t)grow * D;
    const float* whh_p = p.w_hh[dir] + grow * K;
    // The input projection runs on the bf16 instruction from exact three-way splits (bf3.h): the fp32 instruction made
    // the launch matrix-bound at ~70 TFLOP/s.  A = the gate row's weights, columns 16 ks + 8 half + e: split once per
    // block by the first wave of each direction and kept in LDS in fragment order (84 registers otherwise).
    __shared__ uint4 wl[2][KS][3][64];
    const bool pad_out = blockIdx.x == 0;              // block 0 also leaves the padded fp32 matrix for the backward
    if (wv < 2) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        uint32_t hw[4], mw[4], lw[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int col = 16 * ks + 8 * half + 2 * e;
            const bool ok = col < D;                   // (D is a multiple of 4: the pair is in or out together)
            bf3_f2 v;
            v.x = (grow_ok && ok) ? wih_p[col] : 0.f;
            v.y = (grow_ok && ok) ? wih_p[col + 1] : 0.f;
            if (pad_out && ok) { float* o = p.w_pad + (int64_t)(dir * 32 + g) * D + col; o[0] = v.x; o[1] = v.y; }
            bf3_u2 h, m, l;
            bf3_split2(v, h, m, l);
            hw[e] = bf3_pack(h.x, h.y); mw[e] = bf3_pack(m.x, m.y); lw[e] = bf3_pack(l.x, l.y);
        }
        wl[dir][ks][0][lane] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        wl[dir][ks][1][lane] = make_uint4(mw[0], mw[1], mw[2], mw[3]);
        wl[dir][ks][2][lane] = make_uint4(lw[0], lw[1], lw[2], lw[3]);
    }
    }
    __syncthreads();
    if (tile >= ntiles) return;                       // whole wave; no barriers below
    float whh[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 4 * half + i;
        whh[i] = (grow_ok && r < K) ? whh_p[r] : 0.f;
    }
    f32x16 bias;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int unit = 4 * half + (j & 3), row = (j >> 2) * K + unit;
        bias[j] = unit < K ? p.b_ih[dir][row] + p.b_hh[dir][row] : 0.f;
    }
    const int node = lane & 31;
    const int64_t n = min((int64_t)tile * 32 + node, (int64_t)p.N - 1);
    const float* xrow = p.x + n * p.x_sn;
    // 16-byte pieces past the row's end are read from column 0 instead; their weights are zero
    int coff[KS][2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int q = 0; q < 2; ++q) { const int col = 16 * ks + 8 * half + 4 * q; coff[ks][q] = col < D ? col : 0; }
    float4 xv[KS][2];
    {
        const float* xr = xrow + (int64_t)(dir ? K - 1 : 0) * p.x_sk;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            xv[ks][0] = *reinterpret_cast<const float4*>(xr + coff[ks][0]);
            xv[ks][1] = *reinterpret_cast<const float4*>(xr + coff[ks][1]);
        }
    }
    float cst[4] = {0.f, 0.f, 0.f, 0.f}, hst[4] = {0.f, 0.f, 0.f, 0.f};
    float* ap = p.acts + ((int64_t)(tile * 2 + dir) * K) * (20 * 64) + lane;
    float* hs_p = p.hsum + ((int64_t)dir * p.N + n) * K;
    // The first step is peeled: the waits of a loop header are the merge of its two entries, and entered straight
    // from the prologue (no stores in flight) the merge drains the 21 stores of every step before the next one starts.
    auto step = [&](const int s) {
        const int t = dir ? K - 1 - s : s;
        f32x16 acc = bias;
        // each k-step's registers are reloaded with the next slot's columns right behind the instructions that read
        // them, in consumption order (the memory counter is in order: the header then waits for the first piece only)
        const int sn = s + 1 < K ? s + 1 : s;
        const float* xr = xrow + (int64_t)(dir ? K - 1 - sn : sn) * p.x_sk;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf3_u2 h0, m0, l0, h1, m1, l1, h2, m2, l2, h3, m3, l3;
            bf3_split2(bf3_f2{xv[ks][0].x, xv[ks][0].y}, h0, m0, l0);
            bf3_split2(bf3_f2{xv[ks][0].z, xv[ks][0].w}, h1, m1, l1);
            bf3_split2(bf3_f2{xv[ks][1].x, xv[ks][1].y}, h2, m2, l2);
            bf3_split2(bf3_f2{xv[ks][1].z, xv[ks][1].w}, h3, m3, l3);
            xv[ks][0] = *reinterpret_cast<const float4*>(xr + coff[ks][0]);
            xv[ks][1] = *reinterpret_cast<const float4*>(xr + coff[ks][1]);
            const bf3_x8 bh = __builtin_bit_cast(bf3_x8, make_uint4(bf3_pack(h0.x, h0.y), bf3_pack(h1.x, h1.y), bf3_pack(h2.x, h2.y), bf3_pack(h3.x, h3.y)));
            const bf3_x8 bm = __builtin_bit_cast(bf3_x8, make_uint4(bf3_pack(m0.x, m0.y), bf3_pack(m1.x, m1.y), bf3_pack(m2.x, m2.y), bf3_pack(m3.x, m3.y)));
            const bf3_x8 bl = __builtin_bit_cast(bf3_x8, make_uint4(bf3_pack(l0.x, l0.y), bf3_pack(l1.x, l1.y), bf3_pack(l2.x, l2.y), bf3_pack(l3.x, l3.y)));
            const bf3_x8 ah = __builtin_bit_cast(bf3_x8, wl[dir][ks][0][lane]), am = __builtin_bit_cast(bf3_x8, wl[dir][ks][1][lane]),
                         al = __builtin_bit_cast(bf3_x8, wl[dir][ks][2][lane]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);                  // smallest terms first
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(whh[i], hst[i], acc, 0, 0, 0);
        float* a = ap + (int64_t)t * (20 * 64);
        float hs = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float ig = sigm(acc[u]), fg = sigm(acc[4 + u]), gg = tanh_(acc[8 + u]), og = sigm(acc[12 + u]);
            cst[u] = fmaf(fg, cst[u], ig * gg);
            hst[u] = og * tanh_(cst[u]);
            hs += hst[u];
            a[u * 64] = ig; a[(4 + u) * 64] = fg; a[(8 + u) * 64] = gg; a[(12 + u) * 64] = og; a[(16 + u) * 64] = cst[u];
        }
        hs += __shfl_xor(hs, 32);
        hs_p[t] = hs;       // unconditional (a store under a branch makes every wait in the loop a full drain): both halves hold the
                            // same sum, and lanes past N recompute node N-1 bit for bit, so the duplicates write what is there
    };
    step(0);
    for (int s = 1; s < K; ++s) step(s);
}

template <int K>                                      // slots = hidden size (2..8): the walk is unrolled, the three
                                                      // activation buffers rotate by name and every wait is exact
__global__ void __launch_bounds__(kScanThreads) attn_scan_bwd_kernel(const ScanParams p) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int dir = wv & 1;
    const int tile = blockIdx.x * 2 + (wv >> 1);
    const int ntiles = (p.N + 31) >> 5;
    if (tile >= ntiles) return;
    const int half = lane >> 5, m = lane & 31, node = lane & 31;
    float wt[16];                                     // A operand of dh_{t-1} = W_hh^T dg: W_hh[gate(j, half)][m]
    float live[4];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int unit = 4 * half + (j & 3), row = (j >> 2) * K + unit;
        wt[j] = (unit < K && m < K) ? p.w_hh[dir][row * K + m] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) live[u] = 4 * half + u < K ? 1.f : 0.f;
    const int64_t n = min((int64_t)tile * 32 + node, (int64_t)p.N - 1);
    const float* ap = p.acts + ((int64_t)(tile * 2 + dir) * K) * (20 * 64) + lane;
    const float* ds_p = p.ds + n * K;
    // forward visiting order: dir 0 goes t = 0..K-1, dir 1 goes t = K-1..0; BPTT walks it backwards:
    // slot of BPTT step s (s = K-1..0) is t(s) = dir ? K-1-s : s, its predecessor in the forward order is t(s-1).
    auto slot = [&](int s) { return dir ? K - 1 - s : s; };
    float buf[3][20];
    float dsv[K];
#pragma unroll
    for (int s = 0; s < K; ++s) dsv[s] = ds_p[slot(s)];
    {
        const float* a = ap + (int64_t)slot(K - 1) * (20 * 64);
        const float* b = ap + (int64_t)slot(K > 1 ? K - 2 : 0) * (20 * 64);
#pragma unroll
        for (int q = 0; q < 20; ++q) buf[0][q] = a[q * 64];
#pragma unroll
        for (int q = 0; q < 20; ++q) buf[1][q] = b[q * 64];
    }
    float dh[4] = {0.f, 0.f, 0.f, 0.f}, dc[4] = {0.f, 0.f, 0.f, 0.f};
    // dW_hh[q][r] = sum over (node, slot) of dg[q] h_prev[r] is a contraction over NODES - over lanes - so this one product
    // needs its operands transposed: a wave parks dg (32 gate rows) and h_prev (8 rows) of its 32 nodes in LDS, node-minor,
    // and reads them back with lane = row: 16 matrix instructions (two nodes each) per slot into ONE accumulator it keeps for the
    // whole walk.  (The separate weight-gradient launch this replaces re-read dgin and an hprev tensor: 33 us + a reduce.)
    __shared__ float tr_all[kScanThreads / 64][40 * 33];
    float* tr = tr_all[wv];
    const float okf = (int64_t)tile * 32 + node < p.N ? 1.f : 0.f;      // (lanes past N recompute node N-1: counted once)
    f32x16 accw;
#pragma unroll
    for (int j = 0; j < 16; ++j) accw[j] = 0.f;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const int s = K - 1 - i;
        const int t = slot(s);
        float (&cur)[20] = buf[i % 3];
        float (&prv)[20] = buf[(i + 1) % 3];
        float (&pp)[20] = buf[(i + 2) % 3];
        if (s > 1) {                                   // two slots ahead: lands while this step computes
            const float* a = ap + (int64_t)slot(s - 2) * (20 * 64);
#pragma unroll
            for (int q = 0; q < 20; ++q) pp[q] = a[q * 64];
        }
        const float dst = dsv[s];
        float dg[16], hp[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float ig = cur[u], fg = cur[4 + u], gg = cur[8 + u], og = cur[12 + u], ct = cur[16 + u];
            const float cp = s > 0 ? prv[16 + u] : 0.f;                 // the first visited slot starts from h = c = 0
            hp[u] = s > 0 ? prv[12 + u] * tanh_(cp) : 0.f;
            const float tc = tanh_(ct);
            const float dhq = dh[u] + dst;
            const float dcq = fmaf(dhq * og, 1.0f - tc * tc, dc[u]) * live[u];
            dg[u] = dcq * gg * ig * (1.0f - ig);
            dg[4 + u] = dcq * cp * fg * (1.0f - fg);
            dg[8 + u] = dcq * ig * (1.0f - gg * gg);
            dg[12 + u] = dhq * tc * og * (1.0f - og) * live[u];
            dc[u] = dcq * fg;
        }
        {   // unconditional: lanes past N recompute node N-1 bit for bit (the forward wrote them copies of its activations)
            float* dgo = p.dgin + (n * K + t) * 64 + dir * 32 + 4 * half;
#pragma unroll
            for (int ty = 0; ty < 4; ++ty)
                *reinterpret_cast<float4*>(dgo + ty * 8) = make_float4(dg[4 * ty], dg[4 * ty + 1], dg[4 * ty + 2], dg[4 * ty + 3]);
        }
        if (s > 0) {                                   // (the first visited slot has h_prev = 0: nothing to add)
#pragma unroll
            for (int j = 0; j < 16; ++j) tr[((j >> 2) * 8 + half * 4 + (j & 3)) * 33 + node] = dg[j];
#pragma unroll
            for (int u = 0; u < 4; ++u) tr[(32 + half * 4 + u) * 33 + node] = hp[u] * okf;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float a = tr[m * 33 + 2 * q + half];
                const float b = m < 8 ? tr[(32 + m) * 33 + 2 * q + half] : 0.f;
                accw = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, accw, 0, 0, 0);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (s > 0) {
            f32x16 acc;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wt[j], dg[j], acc, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 4; ++u) dh[u] = acc[u];
        }
    }
    if (m < 8) {                                       // this wave's partial [32 gate rows][8 units]: slab row = tile, added in tile order
        float* o = p.whh_slab + ((int64_t)tile * 2 + dir) * 256 + m;
#pragma unroll
        for (int v = 0; v < 16; ++v) o[((v >> 2) * 8 + half * 4 + (v & 3)) * 8] = accw[v];
    }
}

// ---- dx[r, :] = w[r] gout[r / K, :] + dgin[r, 0:64] . w_pad[64, D]   (r = n K + t): the input-gradient product of
// the scan form with the direct part of the weighted sum as its epilogue, so dx is written once.  Taken transposed
// like the scan (d x rows = w_pad^T . dgin^T): wave dt owns output columns [32 dt, 32 dt + 32) with its 64 x 32 slice
// of w_pad resident as the A operands; the 32 x 64 tile of dgin is one contiguous 8-KB block, staged through LDS
// (double-buffered, one barrier per tile) and read as 16-byte B fragments by all four waves; a lane ends up with four
// runs of 4 consecutive output columns of one row, stored as 16 bytes each.  The products run on the bf16 matrix
// instruction from exact three-way splits (bf3.h; the fp32 instruction made this launch matrix-bound at ~70 TFLOP/s):
// w_pad is split once per wave, the tile once by the threads that stage it.
struct DxParams {
    int64_t R; int K, D;
    const float* dgin; const float* w_pad; const float* w; const float* gout; float* dx;
};
constexpr int kDxPk = 144;                            // bytes per staged row of one piece plane: 64 bf16 + 16 (16-B fragment
                                                      // reads of 16 consecutive rows land on distinct banks)

__global__ void __launch_bounds__(256) attn_dx_kernel(const DxParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char planes[2][3][32 * kDxPk];
    const int lane = threadIdx.x & 63, dt = threadIdx.x >> 6;
    const int half = lane >> 5, row = lane & 31;
    const int D = p.D;
    bf3_x8 wa[4][3];                                  // w_pad^T, split once: rows d = 32 dt + row, k = 16 ks + 8 half + e
    {
        const int d = 32 * dt + row;
        const bool ok = d < D;
        const float* wc = p.w_pad + (ok ? d : 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            uint32_t hw[4], mw[4], lw[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k0 = 16 * ks + 8 * half + 2 * e;
                bf3_f2 v;
                v.x = ok ? wc[(int64_t)k0 * D] : 0.f;
                v.y = ok ? wc[(int64_t)(k0 + 1) * D] : 0.f;
                bf3_u2 h, m, l;
                bf3_split2(v, h, m, l);
                hw[e] = bf3_pack(h.x, h.y); mw[e] = bf3_pack(m.x, m.y); lw[e] = bf3_pack(l.x, l.y);
            }
            wa[ks][0] = __builtin_bit_cast(bf3_x8, make_uint4(hw[0], hw[1], hw[2], hw[3]));
            wa[ks][1] = __builtin_bit_cast(bf3_x8, make_uint4(mw[0], mw[1], mw[2], mw[3]));
            wa[ks][2] = __builtin_bit_cast(bf3_x8, make_uint4(lw[0], lw[1], lw[2], lw[3]));
        }
    }
    const int64_t ntiles = (p.R + 31) >> 5;
    const int64_t last4 = p.R * 16 - 1;               // last 16-byte piece of dgin
    const float4* src = reinterpret_cast<const float4*>(p.dgin);
    float4 st0, st1;
    int64_t tile = blockIdx.x;
    auto fetch = [&](int64_t tl) {
        const int64_t base = tl * 512 + threadIdx.x;
        st0 = src[min(base, last4)];
        st1 = src[min(base + 256, last4)];
    };
    // the staging threads split: every element once, three bf16 planes in LDS (8-byte stores, 16 lanes = one row)
    auto park1 = [&](int b, int idx, const float4 v) {
        bf3_u2 h0, m0, l0, h1, m1, l1;
        bf3_split2(bf3_f2{v.x, v.y}, h0, m0, l0);
        bf3_split2(bf3_f2{v.z, v.w}, h1, m1, l1);
        unsigned char* q = &planes[b][0][(idx >> 4) * kDxPk + (idx & 15) * 8];
        *reinterpret_cast<uint2*>(q) = make_uint2(bf3_pack(h0.x, h0.y), bf3_pack(h1.x, h1.y));
        *reinterpret_cast<uint2*>(q + 32 * kDxPk) = make_uint2(bf3_pack(m0.x, m0.y), bf3_pack(m1.x, m1.y));
        *reinterpret_cast<uint2*>(q + 64 * kDxPk) = make_uint2(bf3_pack(l0.x, l0.y), bf3_pack(l1.x, l1.y));
    };
    auto park = [&](int b) { park1(b, threadIdx.x, st0); park1(b, threadIdx.x + 256, st1); };
    auto ld8 = [&](const unsigned char* q) { return __builtin_bit_cast(bf3_x8, *reinterpret_cast<const uint4*>(q)); };
    if (tile < ntiles) { fetch(tile); park(0); }
    __syncthreads();
    int b = 0;
    for (; tile < ntiles; tile += gridDim.x) {
        const int64_t nxt = tile + gridDim.x;
        if (nxt < ntiles) fetch(nxt);
        f32x16 acc;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.f;
        const unsigned char* bp = &planes[b][0][row * kDxPk + 16 * half];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf3_x8 bh = ld8(bp + 32 * ks), bm = ld8(bp + 32 * kDxPk + 32 * ks), bl = ld8(bp + 64 * kDxPk + 32 * ks);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[ks][2], bh, acc, 0, 0, 0);          // smallest terms first
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[ks][0], bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[ks][1], bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[ks][1], bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[ks][0], bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[ks][0], bh, acc, 0, 0, 0);
        }
        const int64_t r = tile * 32 + row;
        if (r < p.R) {
            const int64_t n = r / p.K;
            const float wr = p.w[r];
            const float* g = p.gout + n * D;
            float* o = p.dx + r * D;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int d = 32 * dt + 8 * q + 4 * half;
                if (d < D) {
                    const float4 gv = *reinterpret_cast<const float4*>(g + d);
                    *reinterpret_cast<float4*>(o + d) = make_float4(fmaf(wr, gv.x, acc[4 * q]), fmaf(wr, gv.y, acc[4 * q + 1]),
                                                                    fmaf(wr, gv.z, acc[4 * q + 2]), fmaf(wr, gv.w, acc[4 * q + 3]));
                }
            }
        }
        if (nxt < ntiles) park(b ^ 1);
        __syncthreads();
        b ^= 1;
    }
}

// ---- the padded gradients back in the parameters' shapes, one launch: dw [2,4K,D], db [2,4K], dwhh [2,4K,K]
__global__ void __launch_bounds__(256) attn_unpad_kernel(const float* dw_pad, const float* db_pad, const float* dwhh_pad,
                                                         float* dw, float* db, float* dwhh, int K, int D) {
    const int n_w = 2 * 4 * K * D, n_b = 2 * 4 * K, n_h = 2 * 4 * K * K;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_w + n_b + n_h; i += gridDim.x * 256) {
        if (i < n_w) {
            const int d = i % D, row = i / D, dir = row / (4 * K), g = (row % (4 * K)) / K, u = row % K;
            dw[i] = dw_pad[(int64_t)(dir * 32 + g * 8 + u) * D + d];
        } else if (i < n_w + n_b) {
            const int row = i - n_w, dir = row / (4 * K), g = (row % (4 * K)) / K, u = row % K;
            db[row] = db_pad[dir * 32 + g * 8 + u];
        } else {
            const int j = i - n_w - n_b, c = j % K, row = j / K, dir = row / (4 * K), g = (row % (4 * K)) / K, u = row % K;
            dwhh[j] = dwhh_pad[(dir * 32 + g * 8 + u) * 8 + c];
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

int launch_apply_fwd(const kpgnn_attn_desc* d, const AtParams& p, hipStream_t s) {
    const int K = d->K;
    dim3 blk(kBlock);
    const int vec = apply_vec(d);
    const int g = apply_group(d->D, vec);
    if (g > 64) return fail(KPGNN_ELIMIT, "attention combine: D=%d needs more than 64 lanes", d->D);
    int64_t nb = ((int64_t)d->N + (kBlock / g) - 1) / (kBlock / g);
    const int64_t cap = (int64_t)device_facts().cu_count * 8;
    if (nb > cap) nb = cap;
#define KP_AP(GG) do { if (vec == 4 && K <= 8) hipLaunchKernelGGL((attn_apply_fwd_kernel<GG, 4, 8>), dim3((unsigned)nb), blk, 0, s, p, K); \
                       else if (vec == 4) hipLaunchKernelGGL((attn_apply_fwd_kernel<GG, 4, 16>), dim3((unsigned)nb), blk, 0, s, p, K); \
                       else if (K <= 8) hipLaunchKernelGGL((attn_apply_fwd_kernel<GG, 1, 8>), dim3((unsigned)nb), blk, 0, s, p, K); \
                       else hipLaunchKernelGGL((attn_apply_fwd_kernel<GG, 1, 16>), dim3((unsigned)nb), blk, 0, s, p, K); } while (0)
    switch (g) { case 4: KP_AP(4); break; case 8: KP_AP(8); break; case 16: KP_AP(16); break; case 32: KP_AP(32); break; default: KP_AP(64); break; }
#undef KP_AP
    KPGNN_LAUNCH_CHECK("attn_apply_fwd_kernel");
    return KPGNN_OK;
}

int launch_apply_bwd(const kpgnn_attn_desc* d, const AtParams& p, hipStream_t s) {
    const int K = d->K;
    dim3 blk(kBlock);
    const int vec = apply_vec(d);
    const int g = apply_group(d->D, vec);
    if (g > 64) return fail(KPGNN_ELIMIT, "attention combine: D=%d needs more than 64 lanes", d->D);
    int64_t nb = ((int64_t)d->N + (kBlock / g) - 1) / (kBlock / g);
    const int64_t cap = (int64_t)device_facts().cu_count * 8;
    if (nb > cap) nb = cap;
#define KP_AP(GG) do { if (vec == 4 && K <= 8) hipLaunchKernelGGL((attn_apply_bwd_kernel<GG, 4, 8>), dim3((unsigned)nb), blk, 0, s, p, K); \
                       else if (vec == 4) hipLaunchKernelGGL((attn_apply_bwd_kernel<GG, 4, 16>), dim3((unsigned)nb), blk, 0, s, p, K); \
                       else if (K <= 8) hipLaunchKernelGGL((attn_apply_bwd_kernel<GG, 1, 8>), dim3((unsigned)nb), blk, 0, s, p, K); \
                       else hipLaunchKernelGGL((attn_apply_bwd_kernel<GG, 1, 16>), dim3((unsigned)nb), blk, 0, s, p, K); } while (0)
    switch (g) { case 4: KP_AP(4); break; case 8: KP_AP(8); break; case 16: KP_AP(16); break; case 32: KP_AP(32); break; default: KP_AP(64); break; }
#undef KP_AP
    KPGNN_LAUNCH_CHECK("attn_apply_bwd_kernel");
    return KPGNN_OK;
}

int check_scan(const kpgnn_attn_scan_desc* d, bool bwd) {
    KPGNN_REQUIRE(d != nullptr, "attn_scan: NULL descriptor");
    KPGNN_REQUIRE(d->N >= 0 && d->K >= 1 && d->D >= 1, "attn_scan: bad N=%d K=%d D=%d", d->N, d->K, d->D);
    if (d->K > 8) return fail(KPGNN_ELIMIT, "attention scan: K=%d > 8 (use kpgnn_attn_fwd / _bwd)", d->K);
    if (d->D > 128 || d->D % 4 != 0) return fail(KPGNN_ELIMIT, "attention scan: D=%d unsupported (a multiple of 4, <= 128)", d->D);
    KPGNN_REQUIRE(d->x && d->acts && d->hsum && d->w, "attn_scan: NULL pointer");
    KPGNN_REQUIRE((((uintptr_t)d->x) & 15) == 0 && d->x_sn % 4 == 0 && d->x_sk % 4 == 0, "attn_scan: x must be 16-byte aligned with strides that are multiples of 4");
    for (int q = 0; q < 2; ++q) KPGNN_REQUIRE(d->w_ih[q] && d->w_hh[q] && d->b_ih[q] && d->b_hh[q], "attn_scan: NULL parameter pointer");
    if (bwd) {
        KPGNN_REQUIRE(d->gout && d->dx && d->ds && d->dgin && d->whh_slab && d->dwhh_pad, "attn_scan_bwd: NULL pointer");
        KPGNN_REQUIRE(d->w_pad != nullptr, "attn_scan_bwd: NULL w_pad");
        KPGNN_REQUIRE(((((uintptr_t)d->dgin) | ((uintptr_t)d->gout) | ((uintptr_t)d->dx)) & 15) == 0,
                      "attn_scan_bwd: dgin / gout / dx must be 16-byte aligned");
    } else {
        KPGNN_REQUIRE(d->out && d->w_pad, "attn_scan_fwd: NULL out / w_pad");
    }
    return KPGNN_OK;
}

void fill_scan(const kpgnn_attn_scan_desc* d, ScanParams* q, AtParams* p, kpgnn_attn_desc* a) {
    q->N = d->N; q->K = d->K; q->D = d->D; q->x = d->x; q->x_sn = d->x_sn; q->x_sk = d->x_sk;
    for (int i = 0; i < 2; ++i) { q->w_ih[i] = d->w_ih[i]; q->w_hh[i] = d->w_hh[i]; q->b_ih[i] = d->b_ih[i]; q->b_hh[i] = d->b_hh[i]; }
    q->acts = d->acts; q->hsum = d->hsum; q->w_pad = d->w_pad; q->ds = d->ds; q->dgin = d->dgin; q->whh_slab = d->whh_slab;
    *a = kpgnn_attn_desc{};
    a->N = d->N; a->K = d->K; a->D = d->D; a->x = d->x; a->x_sn = d->x_sn; a->x_sk = d->x_sk;
    a->hsum = d->hsum; a->w = d->w; a->out = d->out; a->gout = d->gout; a->dx = d->dx; a->ds = d->ds;
    fill(a, p);
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
    return launch_apply_fwd(d, p, s);
}

extern "C" int kpgnn_attn_bwd(const kpgnn_attn_desc* d, kpgnn_stream_t stream) {
    int rc = check(d, true);
    if (rc != KPGNN_OK) return rc;
    if (d->N == 0) return KPGNN_OK;
    AtParams p; fill(d, &p);
    hipStream_t s = (hipStream_t)stream;
    const int K = d->K;
    dim3 blk(kBlock);
    rc = launch_apply_bwd(d, p, s);
    if (rc != KPGNN_OK) return rc;
    dim3 grid((unsigned)((d->N + kBlock - 1) / kBlock), 2);
    KP_K_SWITCH(attn_lstm_bwd_kernel, grid, blk, s, p)
    KPGNN_LAUNCH_CHECK("attn_lstm_bwd_kernel");
    return KPGNN_OK;
}

extern "C" int kpgnn_attn_scan_fwd(const kpgnn_attn_scan_desc* d, kpgnn_stream_t stream) {
    int rc = check_scan(d, false);
    if (rc != KPGNN_OK) return rc;
    if (d->N == 0) return KPGNN_OK;
    ScanParams q; AtParams p; kpgnn_attn_desc a;
    fill_scan(d, &q, &p, &a);
    hipStream_t s = (hipStream_t)stream;
    const int ntiles = (d->N + 31) / 32;
    dim3 grid((unsigned)((ntiles + 1) / 2)), blk(kScanThreads);
    const int ks = (d->D + 15) / 16;
    if (ks <= 2) hipLaunchKernelGGL(attn_scan_fwd_kernel<2>, grid, blk, 0, s, q);
    else if (ks <= 4) hipLaunchKernelGGL(attn_scan_fwd_kernel<4>, grid, blk, 0, s, q);
    else if (ks <= 7) hipLaunchKernelGGL(attn_scan_fwd_kernel<7>, grid, blk, 0, s, q);
    else hipLaunchKernelGGL(attn_scan_fwd_kernel<8>, grid, blk, 0, s, q);
    KPGNN_LAUNCH_CHECK("attn_scan_fwd_kernel");
    return launch_apply_fwd(&a, p, s);
}

extern "C" int kpgnn_attn_scan_bwd(const kpgnn_attn_scan_desc* d, kpgnn_stream_t stream) {
    int rc = check_scan(d, true);
    if (rc != KPGNN_OK) return rc;
    if (d->N == 0) return KPGNN_OK;
    ScanParams q; AtParams p; kpgnn_attn_desc a;
    fill_scan(d, &q, &p, &a);
    hipStream_t s = (hipStream_t)stream;
    p.dx = nullptr;                                    // ds only: the direct part of dx is the epilogue of attn_dx_kernel
    rc = launch_apply_bwd(&a, p, s);
    if (rc != KPGNN_OK) return rc;
    const int ntiles = (d->N + 31) / 32;
    dim3 grid((unsigned)((ntiles + 1) / 2)), blk(kScanThreads);
    switch (d->K) {
        case 1: hipLaunchKernelGGL(attn_scan_bwd_kernel<1>, grid, blk, 0, s, q); break;
        case 2: hipLaunchKernelGGL(attn_scan_bwd_kernel<2>, grid, blk, 0, s, q); break;
        case 3: hipLaunchKernelGGL(attn_scan_bwd_kernel<3>, grid, blk, 0, s, q); break;
        case 4: hipLaunchKernelGGL(attn_scan_bwd_kernel<4>, grid, blk, 0, s, q); break;
        case 5: hipLaunchKernelGGL(attn_scan_bwd_kernel<5>, grid, blk, 0, s, q); break;
        case 6: hipLaunchKernelGGL(attn_scan_bwd_kernel<6>, grid, blk, 0, s, q); break;
        case 7: hipLaunchKernelGGL(attn_scan_bwd_kernel<7>, grid, blk, 0, s, q); break;
        default: hipLaunchKernelGGL(attn_scan_bwd_kernel<8>, grid, blk, 0, s, q); break;
    }
    KPGNN_LAUNCH_CHECK("attn_scan_bwd_kernel");
    rc = slab_reduce(d->whh_slab, ntiles, 512, d->dwhh_pad, 512, nullptr, 0, nullptr, s);     // dW_hh (padded): tiles in order
    if (rc != KPGNN_OK) return rc;
    DxParams x;
    x.R = (int64_t)d->N * d->K; x.K = d->K; x.D = d->D;
    x.dgin = d->dgin; x.w_pad = d->w_pad; x.w = d->w; x.gout = d->gout; x.dx = d->dx;
    const int64_t rtiles = (x.R + 31) / 32;
    int64_t nb = (int64_t)device_facts().cu_count * 4;
    if (nb > rtiles) nb = rtiles;
    hipLaunchKernelGGL(attn_dx_kernel, dim3((unsigned)nb), dim3(256), 0, s, x);
    KPGNN_LAUNCH_CHECK("attn_dx_kernel");
    return KPGNN_OK;
}

extern "C" int kpgnn_attn_scan_unpad(const float* dw_pad, const float* db_pad, const float* dwhh_pad, float* dw, float* db,
                                     float* dwhh, int32_t K, int32_t D, kpgnn_stream_t stream) {
    KPGNN_REQUIRE(dw_pad && db_pad && dwhh_pad && dw && db && dwhh, "attn_scan_unpad: NULL pointer");
    KPGNN_REQUIRE(K >= 1 && K <= 8 && D >= 1, "attn_scan_unpad: bad K=%d D=%d", K, D);
    const int total = 8 * K * D + 8 * K + 8 * K * K;
    hipLaunchKernelGGL(attn_unpad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       dw_pad, db_pad, dwhh_pad, dw, db, dwhh, (int)K, (int)D);
    KPGNN_LAUNCH_CHECK("attn_unpad_kernel");
    return KPGNN_OK;
}
