// KP-GIN per-hop MLP + geometric hop-combine + combine_proj, forward and backward, on the fp32 matrix cores (gfx950).
// Contract: include/kpgnn.h, kpgnn_hop_mlp_fwd / _bwd  (reference layers/KPGIN.py:106-112, combine.py:52-58).
//
// Shapes: [N, K, dk] activations with dk = hidden / K (13, 20, 6 ...) and K tiny [dk, dk] weights - HBM-bound work
// that a BLAS tile cannot feed (the library needs 80 us for the [N,13] x [13,104] projection alone).  One 256-thread
// block keeps a 32-node tile resident in LDS (row pitch LD = 4 mod 8: the 16-row x 4-column operand reads of
// v_mfma_f32_16x16x4_f32 then touch 64 distinct banks) next to the zero-padded weights, and does everything that
// needs the tile before it leaves:
//   fwd:  s -> h1 -> h2 (in place, hop by hop) -> comb = sum_k theta_k * h2_k -> out = comb Wc^T + bc;
//         every activation is written to HBM once, coalesced, from LDS; the next tile's s is in flight meanwhile.
//   bwd:  gcomb = gout Wc; dWc += gout^T comb; g2 = gcomb (x) theta * [h2 > 0]; dW2 += h1^T g2;
//         gh1 = g2 W2^T * [h1 > 0]; dW1 += s^T gh1; gs = gh1 W1^T; db1, db2, dbc, dtheta ride along as column sums.
// MFMA operand map (16x16x4, exact fp32): lane l feeds A[m = l & 15][k = l >> 4] and B[k = l >> 4][n = l & 15];
// accumulator register r of lane l is C[m = 4 * (l >> 4) + r][n = l & 15].
// Weight-gradient tiles stay in accumulators for the whole launch; per-block partials go to a slab that is added
// in block order (deterministic).
#include "kpgnn_common.h"

namespace kpgnn {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kTN = 32;          // nodes per tile (2 row groups of 16)
constexpr int kHmThreads = 256;
constexpr int kHmWaves = 4;
constexpr int kNPF = 4;          // float4 registers per thread of a prefetched tile (covers rows up to 128 floats)
constexpr int kMaxCols = 4;      // activation columns a thread may own in the column passes (row width <= 1024)

struct HmParams {
    int64_t N, tiles;
    int K, DI, DO, H, LD, Hp;    // H = 0: no projection; Hp = LDS pitch of the transposed projection weight
    int vec_i, vec_o, vec_h, vec_g;   // 16-B staging allowed for s/gs, h1/h2, out/gout[N,H], gout[N,K,DO]
    const float *s, *w1, *b1, *w2, *b2, *theta, *wc, *bc;
    float *h1, *h2, *out;
    const float* gout;
    float* gs;
    float* slab;                 // [gridDim.x][slab_w]
    int64_t slab_w;
};

// One [rows, K*d] (or [rows, H]) tensor <-> the LDS tile [kTN][LD] whose hop slots are Dm wide.
struct TileIO {
    int ncol, d, Dm, LD, vec;
    int lo[kNPF];                // LDS offset of this thread's q-th float4 (vec path), -1 beyond the tile
    __device__ void init(int ncol_, int d_, int Dm_, int LD_, int vec_) {
        ncol = ncol_; d = d_; Dm = Dm_; LD = LD_; vec = vec_ && (d_ == Dm_);
#pragma unroll
        for (int q = 0; q < kNPF; ++q) {
            const int e = 4 * (threadIdx.x + q * kHmThreads);
            lo[q] = e < kTN * ncol ? (e / ncol) * LD + (e % ncol) : -1;
        }
    }
    __device__ __forceinline__ int lds_col(int c) const { return d == Dm ? c : (c / d) * Dm + (c % d); }
    // issue the loads of a tile into registers (vec path only; the scalar path loads at commit time)
    __device__ __forceinline__ void issue(float4 (&r)[kNPF], const float* __restrict__ src, int rows) const {
        if (!vec) return;
#pragma unroll
        for (int q = 0; q < kNPF; ++q) {
            const int e = 4 * (threadIdx.x + q * kHmThreads);
            r[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < rows * ncol) r[q] = *reinterpret_cast<const float4*>(src + e);
        }
    }
    // registers (+ whatever did not fit, + everything on the scalar path) -> LDS; rows beyond `rows` become zero
    __device__ __forceinline__ void commit(const float4 (&r)[kNPF], float* buf, const float* __restrict__ src, int rows) const {
        if (vec) {
#pragma unroll
            for (int q = 0; q < kNPF; ++q)
                if (lo[q] >= 0) *reinterpret_cast<float4*>(buf + lo[q]) = r[q];
            for (int e = 4 * (threadIdx.x + kNPF * kHmThreads); e < kTN * ncol; e += 4 * kHmThreads) {
                const int n = e / ncol;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (n < rows) v = *reinterpret_cast<const float4*>(src + e);
                *reinterpret_cast<float4*>(buf + n * LD + (e - n * ncol)) = v;
            }
        } else {
            for (int e = threadIdx.x; e < kTN * ncol; e += kHmThreads) {
                const int n = e / ncol;
                buf[n * LD + lds_col(e - n * ncol)] = n < rows ? src[e] : 0.f;
            }
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ dst, const float* buf, int rows) const {
        if (vec) {
#pragma unroll
            for (int q = 0; q < kNPF; ++q) {
                const int e = 4 * (threadIdx.x + q * kHmThreads);
                if (e < rows * ncol) *reinterpret_cast<float4*>(dst + e) = *reinterpret_cast<const float4*>(buf + lo[q]);
            }
            for (int e = 4 * (threadIdx.x + kNPF * kHmThreads); e < rows * ncol; e += 4 * kHmThreads) {
                const int n = e / ncol;
                *reinterpret_cast<float4*>(dst + e) = *reinterpret_cast<const float4*>(buf + n * LD + (e - n * ncol));
            }
        } else {
            for (int e = threadIdx.x; e < rows * ncol; e += kHmThreads) {
                const int n = e / ncol;
                dst[e] = buf[n * LD + lds_col(e - n * ncol)];
            }
        }
    }
};

// acc[rg][t] += A_rg * B for both 16-row groups of the tile: A = rows of `bin`, columns a0 .. a0+din; B = wl, `din`
// rows (zero padded beyond) of pitch `pitch`, column tiles t.  Two independent MFMA chains per column tile.
template <int T>
__device__ __forceinline__ void tile_times_weights(f32x4 (&acc)[2][T], const float* bin, int a0, int din, const float* wl,
                                                   int pitch, int LD, int lr, int lq) {
    const float* ar0 = bin + lr * LD + a0;
    const float* ar1 = ar0 + 16 * LD;
    const int qn = (din + 3) >> 2;
#pragma unroll 2
    for (int q = 0; q < qn; ++q) {
        const int kk = 4 * q + lq;
        const int kc = kk < din ? kk : din - 1;
        float a0v = ar0[kc], a1v = ar1[kc];
        a0v = kk < din ? a0v : 0.f;
        a1v = kk < din ? a1v : 0.f;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const float b = wl[kk * pitch + t * 16 + lr];
            acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0v, b, acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1v, b, acc[1][t], 0, 0, 0);
        }
    }
}

__device__ __forceinline__ float relu_keep_nan(float v) { return v < 0.f ? 0.f : v; }

// ---------------------------------------------------------------------------------------------------------- forward
template <int T>
__global__ void __launch_bounds__(kHmThreads)
hop_mlp_fwd_kernel(const HmParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int P = 16 * T, PP = P * P, CP = P + 4;
    const int K = p.K, DI = p.DI, DO = p.DO, H = p.H, LD = p.LD, Hp = p.Hp;
    const int Dm = DI > DO ? DI : DO;
    float* wl1 = lds;                 // [K][P][P]  wl1[k][i][j] = W1[k][i][j]
    float* wl2 = wl1 + K * PP;        // [K][P][P]
    float* bl1 = wl2 + K * PP;        // [K][P]
    float* bl2 = bl1 + K * P;
    float* th = bl2 + K * P;          // [K][P]
    float* wct = th + K * P;          // [P][Hp]    wct[j][o] = Wc[o][j]   (H > 0)
    float* bcl = wct + (H ? P * Hp : 0);   // [Hp]
    float* cmb = bcl + (H ? Hp : 0);  // [kTN][CP]
    float* buf = cmb + kTN * CP;      // [kTN][LD]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;

    for (int idx = tid; idx < K * PP; idx += kHmThreads) {
        const int k = idx / PP, r = idx - k * PP, i = r / P, j = r - i * P;
        wl1[idx] = (i < DI && j < DO) ? p.w1[((int64_t)k * DI + i) * DO + j] : 0.f;
        wl2[idx] = (i < DO && j < DO) ? p.w2[((int64_t)k * DO + i) * DO + j] : 0.f;
    }
    for (int idx = tid; idx < K * P; idx += kHmThreads) {
        const int k = idx / P, j = idx - k * P;
        bl1[idx] = j < DO ? p.b1[k * DO + j] : 0.f;
        bl2[idx] = j < DO ? p.b2[k * DO + j] : 0.f;
        th[idx] = (p.theta && j < DO) ? p.theta[k * DO + j] : 0.f;
    }
    if (H) {
        for (int idx = tid; idx < P * Hp; idx += kHmThreads) {
            const int j = idx / Hp, o = idx - j * Hp;
            wct[idx] = (j < DO && o < H) ? p.wc[(int64_t)o * DO + j] : 0.f;
        }
        for (int o = tid; o < Hp; o += kHmThreads) bcl[o] = (o < H && p.bc) ? p.bc[o] : 0.f;
    }
    const int ncol_i = K * DI, ncol_o = K * DO;
    TileIO io_i, io_o, io_h;
    io_i.init(ncol_i, DI, Dm, LD, p.vec_i);
    io_o.init(ncol_o, DO, Dm, LD, p.vec_o);
    io_h.init(H ? H : 4, H ? H : 4, H ? H : 4, LD, p.vec_h);
    float4 pf[kNPF];
    if ((int64_t)blockIdx.x < p.tiles) {
        const int64_t n0 = (int64_t)blockIdx.x * kTN;
        io_i.issue(pf, p.s + n0 * ncol_i, (int)((p.N - n0) < kTN ? (p.N - n0) : kTN));
    }
    for (int64_t tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
        const int64_t n0 = tile * kTN;
        const int rows = (int)((p.N - n0) < kTN ? (p.N - n0) : kTN);
        __syncthreads();                                        // previous tile's readers; first pass: the weight fill
        io_i.commit(pf, buf, p.s + n0 * ncol_i, rows);
        __syncthreads();
        {
            const int64_t nt = tile + gridDim.x;
            if (nt < p.tiles) {
                const int64_t m0 = nt * kTN;
                io_i.issue(pf, p.s + m0 * ncol_i, (int)((p.N - m0) < kTN ? (p.N - m0) : kTN));
            }
        }
        for (int k = wave; k < K; k += kHmWaves) {              // layer 1, in place: slot k (s) -> slot k (h1)
            f32x4 acc[2][T];
#pragma unroll
            for (int jt = 0; jt < T; ++jt) { const float b = bl1[k * P + jt * 16 + lr]; acc[0][jt] = {b, b, b, b}; acc[1][jt] = acc[0][jt]; }
            tile_times_weights<T>(acc, buf, k * Dm, DI, wl1 + k * PP, P, LD, lr, lq);
#pragma unroll
            for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                for (int jt = 0; jt < T; ++jt) {
                    const int col = jt * 16 + lr;
                    if (col < DO)
#pragma unroll
                        for (int r = 0; r < 4; ++r) buf[(rg * 16 + 4 * lq + r) * LD + k * Dm + col] = relu_keep_nan(acc[rg][jt][r]);
                }
        }
        __syncthreads();
        io_o.store(p.h1 + n0 * ncol_o, buf, rows);
        __syncthreads();
        for (int k = wave; k < K; k += kHmWaves) {              // layer 2, in place: h1 -> h2
            f32x4 acc[2][T];
#pragma unroll
            for (int jt = 0; jt < T; ++jt) { const float b = bl2[k * P + jt * 16 + lr]; acc[0][jt] = {b, b, b, b}; acc[1][jt] = acc[0][jt]; }
            tile_times_weights<T>(acc, buf, k * Dm, DO, wl2 + k * PP, P, LD, lr, lq);
#pragma unroll
            for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                for (int jt = 0; jt < T; ++jt) {
                    const int col = jt * 16 + lr;
                    if (col < DO)
#pragma unroll
                        for (int r = 0; r < 4; ++r) buf[(rg * 16 + 4 * lq + r) * LD + k * Dm + col] = relu_keep_nan(acc[rg][jt][r]);
                }
        }
        __syncthreads();
        io_o.store(p.h2 + n0 * ncol_o, buf, rows);
        if (p.theta) {                                          // comb[n, j] = sum_k theta[k, j] h2[n, k, j]
            for (int c = tid; c < kTN * DO; c += kHmThreads) {
                const int node = c / DO, j = c - node * DO;
                const float* hrow = buf + node * LD + j;
                float acc = 0.f;
                for (int k = 0; k < K; ++k) acc = fmaf(th[k * P + j], hrow[k * Dm], acc);
                if (H) cmb[node * CP + j] = acc;
                else if (node < rows) p.out[(n0 + node) * DO + j] = acc;
            }
        }
        if (H) {                                                // out = comb Wc^T + bc, staged through the tile buffer
            __syncthreads();
            const int not16 = (H + 15) >> 4;
            for (int ot = wave; ot < not16; ot += kHmWaves) {
                f32x4 acc[2][1];
                const float b = bcl[ot * 16 + lr];
                acc[0][0] = {b, b, b, b}; acc[1][0] = acc[0][0];
                tile_times_weights<1>(acc, cmb, 0, DO, wct + ot * 16, Hp, CP, lr, lq);
                const int col = ot * 16 + lr;
                if (col < H)
#pragma unroll
                    for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                        for (int r = 0; r < 4; ++r) buf[(rg * 16 + 4 * lq + r) * LD + col] = acc[rg][0][r];
            }
            __syncthreads();
            io_h.store(p.out + n0 * H, buf, rows);
        }
    }
}

// --------------------------------------------------------------------------------------------------------- backward
// Column passes: a thread owns activation column c (row lanes rl share a column when the row is narrower than the block).
struct ColMap {
    int R, rl, c0; bool on;
    __device__ void init(int ncol) {
        R = kHmThreads / ncol;
        R = R < 1 ? 1 : (R > kTN ? kTN : R);
        rl = ncol < kHmThreads ? (int)threadIdx.x / ncol : 0;
        c0 = ncol < kHmThreads ? (int)threadIdx.x - rl * ncol : (int)threadIdx.x;
        on = ncol < kHmThreads ? rl < R : true;
    }
};

template <int T, int MAXI>
__global__ void __launch_bounds__(kHmThreads, 2)
hop_mlp_bwd_kernel(const HmParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int P = 16 * T, PP = P * P, TT = T * T, CP = P + 4, PC = (T == 1) ? 16 : 48;
    const int K = p.K, DI = p.DI, DO = p.DO, H = p.H, LD = p.LD;
    const int Dm = DI > DO ? DI : DO;
    const int H4 = (H + 3) & ~3;
    float* wt2 = lds;                 // [K][P][P]  wt2[k][j][i] = W2[k][i][j]
    float* wt1 = wt2 + K * PP;        // [K][P][P]  wt1[k][j][i] = W1[k][i][j]
    float* th = wt1 + K * PP;         // [K][P]
    float* wcn = th + K * P;          // [H4][PC]   wcn[o][j] = Wc[o][j]   (H > 0)
    float* cmb = wcn + (H ? H4 * PC : 0);   // [kTN][CP]  comb
    float* gcm = cmb + kTN * CP;      // [kTN][CP]  d(comb)
    float* bufG = gcm + kTN * CP;     // [kTN][LD]  h2, then g2, then gh1
    float* bufX = bufG + kTN * LD;    // [kTN][LD]  gout, then h1, then s, then gs
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;

    for (int idx = tid; idx < K * PP; idx += kHmThreads) {
        const int k = idx / PP, r = idx - k * PP, j = r / P, i = r - j * P;
        wt2[idx] = (i < DO && j < DO) ? p.w2[((int64_t)k * DO + i) * DO + j] : 0.f;
        wt1[idx] = (i < DI && j < DO) ? p.w1[((int64_t)k * DI + i) * DO + j] : 0.f;
    }
    for (int idx = tid; idx < K * P; idx += kHmThreads) {
        const int k = idx / P, j = idx - k * P;
        th[idx] = (p.theta && j < DO) ? p.theta[k * DO + j] : 0.f;
    }
    if (H)
        for (int idx = tid; idx < H4 * PC; idx += kHmThreads) {
            const int o = idx / PC, j = idx - o * PC;
            wcn[idx] = (o < H && j < DO) ? p.wc[(int64_t)o * DO + j] : 0.f;
        }
    const int ncol_i = K * DI, ncol_o = K * DO;
    const bool plain = p.theta == nullptr;           // gout is [N,K,DO]
    const int gcols = H ? H : ncol_o;                // row width of a gout tile staged in bufX (H, or plain mode)
    TileIO io_i, io_o, io_g;
    io_i.init(ncol_i, DI, Dm, LD, p.vec_i);
    io_o.init(ncol_o, DO, Dm, LD, p.vec_o);
    if (H) io_g.init(H, H, H, LD, p.vec_h);
    else io_g.init(ncol_o, DO, Dm, LD, p.vec_g);
    const bool gtile = H || plain;
    ColMap cm, ch;
    cm.init(ncol_o);
    ch.init(H ? H : 1);
    float gb1[kMaxCols], gb2[kMaxCols], gth[kMaxCols], gbc[kMaxCols];
#pragma unroll
    for (int m = 0; m < kMaxCols; ++m) gb1[m] = gb2[m] = gth[m] = gbc[m] = 0.f;
    const int nitems = K * TT;
    const int not16 = (H + 15) >> 4, nitems_c = not16 * T;
    f32x4 acc1[MAXI], acc2[MAXI], accC[MAXI];
#pragma unroll
    for (int m = 0; m < MAXI; ++m) { acc1[m] = {0.f, 0.f, 0.f, 0.f}; acc2[m] = acc1[m]; accC[m] = acc1[m]; }

    float4 r1[kNPF], r2[kNPF];        // tiles in flight: r1 = gout -> h1 -> s, r2 = h2
    if ((int64_t)blockIdx.x < p.tiles) {
        const int64_t n0 = (int64_t)blockIdx.x * kTN;
        const int rows = (int)((p.N - n0) < kTN ? (p.N - n0) : kTN);
        if (gtile) io_g.issue(r1, p.gout + n0 * gcols, rows);
        io_o.issue(r2, p.h2 + n0 * ncol_o, rows);
    }
    for (int64_t tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
        const int64_t n0 = tile * kTN;
        const int rows = (int)((p.N - n0) < kTN ? (p.N - n0) : kTN);
        // the (hop, tile) coordinates and operand addresses of the weight-gradient tiles are tile-loop invariant;
        // recomputing them (a few SALU ops) beats keeping ~100 hoisted VGPRs alive
        int wv = __builtin_amdgcn_readfirstlane(wave);
        asm volatile("" : "+s"(wv));
        __syncthreads();
        if (gtile) io_g.commit(r1, bufX, p.gout + n0 * gcols, rows);
        io_o.commit(r2, bufG, p.h2 + n0 * ncol_o, rows);
        if (!gtile)                                             // theta without projection: gout [N, DO] is d(comb)
            for (int c = tid; c < kTN * DO; c += kHmThreads) {
                const int n = c / DO, j = c - n * DO;
                gcm[n * CP + j] = n < rows ? p.gout[(n0 + n) * DO + j] : 0.f;
            }
        __syncthreads();
        io_o.issue(r1, p.h1 + n0 * ncol_o, rows);
        if (H) {
            // comb (needed by dWc), d(comb) = gout Wc, dbc
            for (int c = tid; c < kTN * DO; c += kHmThreads) {
                const int n = c / DO, j = c - n * DO;
                const float* hrow = bufG + n * LD + j;
                float a = 0.f;
                for (int k = 0; k < K; ++k) a = fmaf(th[k * P + j], hrow[k * Dm], a);
                cmb[n * CP + j] = a;
            }
            if (wv < T) {                                       // wave jt: both row groups, K-dim = the H outputs
                f32x4 acc[2][1];
                acc[0][0] = {0.f, 0.f, 0.f, 0.f}; acc[1][0] = acc[0][0];
                tile_times_weights<1>(acc, bufX, 0, H, wcn + wv * 16, PC, LD, lr, lq);
                const int col = wv * 16 + lr;
                if (col < DO)
#pragma unroll
                    for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                        for (int r = 0; r < 4; ++r) gcm[(rg * 16 + 4 * lq + r) * CP + col] = acc[rg][0][r];
            }
            if (ch.on) {
#pragma unroll
                for (int m = 0; m < kMaxCols; ++m) {
                    const int c = ch.c0 + m * kHmThreads;
                    if (c < H)
#pragma unroll 4
                        for (int n = ch.rl; n < kTN; n += ch.R) gbc[m] += bufX[n * LD + c];
                }
            }
            __syncthreads();
            // dWc[o][j] += sum_n gout[n][o] comb[n][j]
#pragma unroll
            for (int m = 0; m < MAXI; ++m) {
                const int it = wv + m * kHmWaves;
                if (it < nitems_c) {
                    const int ot = it / T, jt = it - ot * T;
                    const int ca = ot * 16 + lr, cb = jt * 16 + lr;
                    const float* pa = bufX + (ca < H ? ca : H - 1) + lq * LD;
                    const float* pb = cmb + (cb < DO ? cb : DO - 1) + lq * CP;
#pragma unroll 4
                    for (int q = 0; q < kTN / 4; ++q) {
                        float a = pa[4 * q * LD], b = pb[4 * q * CP];
                        a = ca < H ? a : 0.f;
                        b = cb < DO ? b : 0.f;
                        accC[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, accC[m], 0, 0, 0);
                    }
                }
            }
        }
        // A. g2 = d(h2) * [h2 > 0] in place over h2; db2 and dtheta column sums
        if (cm.on) {
#pragma unroll
            for (int m = 0; m < kMaxCols; ++m) {
                const int c = cm.c0 + m * kHmThreads;
                if (c < ncol_o) {
                    const int k = c / DO, j = c - k * DO, lc = k * Dm + j;
                    const float thv = th[k * P + j];
#pragma unroll 4
                    for (int n = cm.rl; n < kTN; n += cm.R) {
                        const float h2v = bufG[n * LD + lc];
                        float gv;
                        if (plain) {
                            gv = bufX[n * LD + lc];
                        } else {
                            const float go = gcm[n * CP + j];
                            gth[m] = fmaf(go, h2v, gth[m]);
                            gv = go * thv;
                        }
                        const float g2 = h2v > 0.f ? gv : 0.f;
                        gb2[m] += g2;
                        bufG[n * LD + lc] = g2;
                    }
                }
            }
        }
        __syncthreads();
        io_o.commit(r1, bufX, p.h1 + n0 * ncol_o, rows);       // gout is dead: h1 takes its place
        io_i.issue(r1, p.s + n0 * ncol_i, rows);
        __syncthreads();
        // B. dW2[k] += h1_k^T g2_k
#pragma unroll
        for (int m = 0; m < MAXI; ++m) {
            const int it = wv + m * kHmWaves;
            if (it < nitems) {
                const int k = it / TT, tt = it - k * TT, ti = tt / T, tj = tt - ti * T;
                const int ca = ti * 16 + lr, cb = tj * 16 + lr;
                const float* pa = bufX + k * Dm + (ca < DO ? ca : DO - 1) + lq * LD;
                const float* pb = bufG + k * Dm + (cb < DO ? cb : DO - 1) + lq * LD;
#pragma unroll 4
                for (int q = 0; q < kTN / 4; ++q) {
                    float a = pa[4 * q * LD], b = pb[4 * q * LD];
                    a = ca < DO ? a : 0.f;
                    b = cb < DO ? b : 0.f;
                    acc2[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc2[m], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        // C. gh1 = (g2 W2^T) * [h1 > 0], in place over g2 (a hop slot is read and written by one wave only)
        for (int k = wv; k < K; k += kHmWaves) {
            f32x4 acc[2][T];
#pragma unroll
            for (int ti = 0; ti < T; ++ti) { acc[0][ti] = {0.f, 0.f, 0.f, 0.f}; acc[1][ti] = acc[0][ti]; }
            tile_times_weights<T>(acc, bufG, k * Dm, DO, wt2 + k * PP, P, LD, lr, lq);
#pragma unroll
            for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                for (int ti = 0; ti < T; ++ti) {
                    const int col = ti * 16 + lr;
                    if (col < DO)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int idx = (rg * 16 + 4 * lq + r) * LD + k * Dm + col;
                            bufG[idx] = bufX[idx] > 0.f ? acc[rg][ti][r] : 0.f;
                        }
                }
        }
        __syncthreads();
        // D. s -> bufX (h1 is no longer needed); the next tile's gout / h2 start their flight
        io_i.commit(r1, bufX, p.s + n0 * ncol_i, rows);
        {
            const int64_t nt = tile + gridDim.x;
            if (nt < p.tiles) {
                const int64_t m0 = nt * kTN;
                const int nrows = (int)((p.N - m0) < kTN ? (p.N - m0) : kTN);
                if (gtile) io_g.issue(r1, p.gout + m0 * gcols, nrows);
                io_o.issue(r2, p.h2 + m0 * ncol_o, nrows);
            }
        }
        __syncthreads();
        // E. dW1[k] += s_k^T gh1_k ; db1 column sums
#pragma unroll
        for (int m = 0; m < MAXI; ++m) {
            const int it = wv + m * kHmWaves;
            if (it < nitems) {
                const int k = it / TT, tt = it - k * TT, ti = tt / T, tj = tt - ti * T;
                const int ca = ti * 16 + lr, cb = tj * 16 + lr;
                const float* pa = bufX + k * Dm + (ca < DI ? ca : DI - 1) + lq * LD;
                const float* pb = bufG + k * Dm + (cb < DO ? cb : DO - 1) + lq * LD;
#pragma unroll 4
                for (int q = 0; q < kTN / 4; ++q) {
                    float a = pa[4 * q * LD], b = pb[4 * q * LD];
                    a = ca < DI ? a : 0.f;
                    b = cb < DO ? b : 0.f;
                    acc1[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc1[m], 0, 0, 0);
                }
            }
        }
        if (cm.on) {
#pragma unroll
            for (int m = 0; m < kMaxCols; ++m) {
                const int c = cm.c0 + m * kHmThreads;
                if (c < ncol_o) {
                    const int k = c / DO, lc = k * Dm + (c - k * DO);
#pragma unroll 4
                    for (int n = cm.rl; n < kTN; n += cm.R) gb1[m] += bufG[n * LD + lc];
                }
            }
        }
        __syncthreads();
        // F. gs = gh1 W1^T -> bufX (over s)
        for (int k = wv; k < K; k += kHmWaves) {
            f32x4 acc[2][T];
#pragma unroll
            for (int ti = 0; ti < T; ++ti) { acc[0][ti] = {0.f, 0.f, 0.f, 0.f}; acc[1][ti] = acc[0][ti]; }
            tile_times_weights<T>(acc, bufG, k * Dm, DO, wt1 + k * PP, P, LD, lr, lq);
#pragma unroll
            for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                for (int ti = 0; ti < T; ++ti) {
                    const int col = ti * 16 + lr;
                    if (col < DI)
#pragma unroll
                        for (int r = 0; r < 4; ++r) bufX[(rg * 16 + 4 * lq + r) * LD + k * Dm + col] = acc[rg][ti][r];
                }
        }
        __syncthreads();
        io_i.store(p.gs + n0 * ncol_i, bufX, rows);
    }

    // per-block partials -> slab row [dW1 | db1 | dW2 | db2 | dtheta | dWc | dbc]
    float* row = p.slab + (int64_t)blockIdx.x * p.slab_w;
    const int64_t o_b1 = (int64_t)K * DI * DO, o_w2 = o_b1 + ncol_o, o_b2 = o_w2 + (int64_t)K * DO * DO, o_th = o_b2 + ncol_o;
    const int64_t o_wc = o_th + (p.theta ? ncol_o : 0), o_bc = o_wc + (int64_t)H * DO;
#pragma unroll
    for (int m = 0; m < MAXI; ++m) {
        const int it = wave + m * kHmWaves;
        if (it < nitems) {
            const int k = it / TT, tt = it - k * TT, ti = tt / T, tj = tt - ti * T;
            const int j = tj * 16 + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = ti * 16 + 4 * lq + r;
                if (i < DI && j < DO) row[((int64_t)k * DI + i) * DO + j] = acc1[m][r];
                if (i < DO && j < DO) row[o_w2 + ((int64_t)k * DO + i) * DO + j] = acc2[m][r];
            }
        }
        if (it < nitems_c) {
            const int ot = it / T, jt = it - ot * T;
            const int j = jt * 16 + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = ot * 16 + 4 * lq + r;
                if (o < H && j < DO) row[o_wc + (int64_t)o * DO + j] = accC[m][r];
            }
        }
    }
    // column sums: the row lanes of a column meet in LDS (bufG is free now)
    for (int which = 0; which < 4; ++which) {
        if (which == 2 && !p.theta) continue;
        if (which == 3 && !H) continue;
        const ColMap& mp = which == 3 ? ch : cm;
        const int ncol = which == 3 ? H : ncol_o;
        __syncthreads();
        if (mp.on) {
#pragma unroll
            for (int m = 0; m < kMaxCols; ++m) {
                const int c = mp.c0 + m * kHmThreads;
                if (c < ncol) bufG[mp.rl * ncol + c] = which == 0 ? gb1[m] : (which == 1 ? gb2[m] : (which == 2 ? gth[m] : gbc[m]));
            }
        }
        __syncthreads();
        const int64_t off = which == 0 ? o_b1 : (which == 1 ? o_b2 : (which == 2 ? o_th : o_bc));
        for (int c = tid; c < ncol; c += kHmThreads) {
            float tot = 0.f;
            for (int r = 0; r < mp.R; ++r) tot += bufG[r * ncol + c];
            row[off + c] = tot;
        }
    }
}

struct HmPlan { int T, maxi, LD, Hp, grid_fwd, grid_bwd; size_t lds_fwd, lds_bwd; int64_t slab_w; };

int hm_plan(int64_t N, int K, int DI, int DO, int H, bool theta, HmPlan* pl) {
    const int D = DI > DO ? DI : DO;
    if (D > 32) return fail(KPGNN_ELIMIT, "hop_mlp: per-hop width %d exceeds 32", D);
    if (H < 0 || H > 1024) return fail(KPGNN_ELIMIT, "hop_mlp: projection width %d exceeds 1024", H);
    const int T = (D + 15) / 16;
    pl->T = T;
    int items = K * T * T;
    const int items_c = ((H + 15) / 16) * T;
    if (items_c > items) items = items_c;
    if (items > 8 * kHmWaves) return fail(KPGNN_ELIMIT, "hop_mlp: %d weight-gradient tiles exceed %d", items, 8 * kHmWaves);
    pl->maxi = items <= 2 * kHmWaves ? 2 : (items <= 4 * kHmWaves ? 4 : 8);
    if ((int64_t)K * D > (int64_t)kMaxCols * kHmThreads)
        return fail(KPGNN_ELIMIT, "hop_mlp: K*D=%lld exceeds %d", (long long)K * D, kMaxCols * kHmThreads);
    int v = K * D;
    if (H > v) v = H;
    pl->LD = v + ((4 - v % 8) + 8) % 8;                      // pitch = 4 (mod 8)
    int Hp = (H + 15) & ~15;
    if (Hp % 32 == 0) Hp += 16;                              // pitch = 16 (mod 32): the 4 k-rows of a B read are 16 banks apart
    pl->Hp = H ? Hp : 0;
    const size_t P = 16 * (size_t)T, CP = P + 4, PC = T == 1 ? 16 : 48, H4 = (size_t)((H + 3) & ~3);
    pl->lds_fwd = sizeof(float) * (2 * K * P * P + 3 * K * P + (H ? P * Hp + Hp : 0) + kTN * CP + (size_t)kTN * pl->LD);
    pl->lds_bwd = sizeof(float) * (2 * K * P * P + K * P + (H ? H4 * PC : 0) + 2 * kTN * CP + 2 * (size_t)kTN * pl->LD);
    const size_t cap = (size_t)device_facts().lds_per_block;
    if (pl->lds_fwd > cap || pl->lds_bwd > cap)
        return fail(KPGNN_ELIMIT, "hop_mlp: %zu B of LDS needed", pl->lds_fwd > pl->lds_bwd ? pl->lds_fwd : pl->lds_bwd);
    const int64_t tiles = (N + kTN - 1) / kTN;
    const int cu = device_facts().cu_count;
    auto grid_for = [&](size_t lds, int reg_blocks) {
        int per_cu = (int)(cap / lds);
        per_cu = per_cu < 1 ? 1 : (per_cu > reg_blocks ? reg_blocks : per_cu);
        int64_t g = (int64_t)cu * per_cu;
        if (g > tiles) g = tiles;
        return (int)(g < 1 ? 1 : g);
    };
    pl->grid_fwd = grid_for(pl->lds_fwd, 8);
    pl->grid_bwd = grid_for(pl->lds_bwd, 2);
    pl->slab_w = (int64_t)K * DI * DO + (int64_t)K * DO * DO + 2 * (int64_t)K * DO + (theta ? (int64_t)K * DO : 0) +
                 (int64_t)H * DO + H;
    return KPGNN_OK;
}

void hm_fill(const kpgnn_hop_mlp_desc* d, const HmPlan& pl, HmParams* p) {
    p->N = d->N; p->tiles = (d->N + kTN - 1) / kTN;
    p->K = d->K; p->DI = d->DI; p->DO = d->DO; p->H = d->H; p->LD = pl.LD; p->Hp = pl.Hp;
    p->s = d->s; p->w1 = d->w1; p->b1 = d->b1; p->w2 = d->w2; p->b2 = d->b2; p->theta = d->theta;
    p->wc = d->wc; p->bc = d->bc;
    p->h1 = d->h1; p->h2 = d->h2; p->out = d->out; p->gout = d->gout; p->gs = d->gs;
    p->slab = (float*)d->workspace; p->slab_w = pl.slab_w;
    auto al16 = [](const void* q) { return q != nullptr && (((uintptr_t)q) & 15) == 0; };
    // (tiles start at multiples of 32 rows, so a row width that is a multiple of 4 floats keeps every tile 16-B aligned)
    p->vec_i = ((d->K * d->DI) % 4 == 0) && al16(d->s) && (d->gs == nullptr || al16(d->gs));
    p->vec_o = ((d->K * d->DO) % 4 == 0) && al16(d->h1) && al16(d->h2);
    p->vec_h = d->H > 0 && (d->H % 4 == 0) && (d->out == nullptr || al16(d->out)) && (d->gout == nullptr || al16(d->gout));
    p->vec_g = ((d->K * d->DO) % 4 == 0) && al16(d->gout);
}

int hm_check(const kpgnn_hop_mlp_desc* d, const char* who) {
    KPGNN_REQUIRE(d != nullptr, "%s: NULL descriptor", who);
    KPGNN_REQUIRE(d->N >= 0 && d->K >= 1 && d->DI >= 1 && d->DO >= 1 && d->H >= 0, "%s: bad N=%lld K=%d DI=%d DO=%d H=%d", who,
                  (long long)d->N, d->K, d->DI, d->DO, d->H);
    KPGNN_REQUIRE(d->s && d->w1 && d->b1 && d->w2 && d->b2 && d->h1 && d->h2, "%s: NULL tensor", who);
    KPGNN_REQUIRE(d->H == 0 || (d->theta && d->wc), "%s: the projection needs theta and wc", who);
    return KPGNN_OK;
}

}  // namespace
}  // namespace kpgnn

using namespace kpgnn;

extern "C" size_t kpgnn_hop_mlp_workspace_bytes(int64_t N, int32_t K, int32_t DI, int32_t DO, int32_t H) {
    HmPlan pl;
    if (N < 1 || K < 1 || DI < 1 || DO < 1 || H < 0) return 0;
    if (hm_plan(N, K, DI, DO, H, true, &pl) != KPGNN_OK) return 0;  // 0 = "does not fit": the caller keeps its BLAS path
    return sizeof(float) * (size_t)pl.grid_bwd * (size_t)pl.slab_w;
}

extern "C" int kpgnn_hop_mlp_fwd(const kpgnn_hop_mlp_desc* d, kpgnn_stream_t stream) {
    if (int rc = hm_check(d, "hop_mlp_fwd")) return rc;
    KPGNN_REQUIRE(!d->theta || d->out, "hop_mlp_fwd: theta without out");
    if (d->N == 0) return KPGNN_OK;
    HmPlan pl;
    if (int rc = hm_plan(d->N, d->K, d->DI, d->DO, d->H, d->theta != nullptr, &pl)) return rc;
    HmParams p;
    hm_fill(d, pl, &p);
    hipStream_t s = (hipStream_t)stream;
    // (persistent tile loop: the grid is one resident round - the plan sizes it by LDS, the registers may allow fewer)
    auto one_round = [&](int nb, int grid) {
        const int64_t cap = (int64_t)nb * device_facts().cu_count;
        return nb > 0 && cap < grid ? (int)cap : grid;
    };
    if (pl.T == 1) {
        KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)hop_mlp_fwd_kernel<1>, pl.lds_fwd));
        const int grid = one_round(resident_blocks(hop_mlp_fwd_kernel<1>, kHmThreads, pl.lds_fwd), pl.grid_fwd);
        hipLaunchKernelGGL(hop_mlp_fwd_kernel<1>, dim3(grid), dim3(kHmThreads), pl.lds_fwd, s, p);
    } else {
        KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)hop_mlp_fwd_kernel<2>, pl.lds_fwd));
        const int grid = one_round(resident_blocks(hop_mlp_fwd_kernel<2>, kHmThreads, pl.lds_fwd), pl.grid_fwd);
        hipLaunchKernelGGL(hop_mlp_fwd_kernel<2>, dim3(grid), dim3(kHmThreads), pl.lds_fwd, s, p);
    }
    KPGNN_LAUNCH_CHECK("hop_mlp_fwd_kernel");
    return KPGNN_OK;
}

extern "C" int kpgnn_hop_mlp_bwd(const kpgnn_hop_mlp_desc* d, kpgnn_stream_t stream) {
    if (int rc = hm_check(d, "hop_mlp_bwd")) return rc;
    KPGNN_REQUIRE(d->gout && d->gs && d->gflat, "hop_mlp_bwd: NULL gradient tensor");
    HmPlan pl;
    if (int rc = hm_plan(d->N > 0 ? d->N : 1, d->K, d->DI, d->DO, d->H, d->theta != nullptr, &pl)) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (d->N == 0) {
        KPGNN_HIP_TRY(hipMemsetAsync(d->gflat, 0, sizeof(float) * (size_t)pl.slab_w, s));
        return KPGNN_OK;
    }
    KPGNN_REQUIRE(d->workspace && d->workspace_bytes >= sizeof(float) * (size_t)pl.grid_bwd * (size_t)pl.slab_w,
                  "hop_mlp_bwd: workspace too small");
    HmParams p;
    hm_fill(d, pl, &p);
#define KP_HM_BWD(TV, MV) do { \
        KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)hop_mlp_bwd_kernel<TV, MV>, pl.lds_bwd)); \
        hipLaunchKernelGGL((hop_mlp_bwd_kernel<TV, MV>), dim3(pl.grid_bwd), dim3(kHmThreads), pl.lds_bwd, s, p); } while (0)
    if (pl.T == 1) {
        if (pl.maxi == 2) KP_HM_BWD(1, 2); else if (pl.maxi == 4) KP_HM_BWD(1, 4); else KP_HM_BWD(1, 8);
    } else {
        if (pl.maxi == 2) KP_HM_BWD(2, 2); else if (pl.maxi == 4) KP_HM_BWD(2, 4); else KP_HM_BWD(2, 8);
    }
#undef KP_HM_BWD
    KPGNN_LAUNCH_CHECK("hop_mlp_bwd_kernel");
    return slab_reduce(p.slab, pl.grid_bwd, pl.slab_w, d->gflat, pl.slab_w, nullptr, 0, nullptr, s);
}
