// Training-mode BatchNorm1d (+ReLU, +residual) over [N, C] rows for gfx950.  Contract: include/kpgnn.h.
// HBM-bound streaming kernels around a column-statistics slot (kpgnn.h, "Column-statistics slots"):
//   forward   [stats: sum x, sum x^2 -> slot]  ->  apply: finish mean / invstd from the slot, z = bn(x) (+relu, +residual),
//             optionally the statistics of z into a second slot (the next BatchNorm's stats pass disappears)
//   backward  reduce: sum dy, sum dy*xhat -> slot  ->  apply: dx, dgamma, dbeta
// The stats / reduce half is skipped when the producer of the tensor has already filled the slot (the fused linear
// kernels do, lin_fused.h).  Round 1 had a slab + an ordered slab-reduce launch between the halves (3 launches per
// direction, ~4.7 us each at any size); the slot makes it 2, or 1.
// A sub-group of G lanes spans the C columns 16 B wide; sub-groups stride over rows; partial sums are fp64 registers.
#include <initializer_list>

#include "kpgnn_common.h"

namespace kpgnn {
namespace {

constexpr int kBlock = 256;
constexpr int kProducerBlocks = 512;   // blocks of a launch that adds to a slot: 512 / 8 replicas = 64 adds per address

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
    const int32_t* n_dyn;
    int64_t N; int C, relu; float eps, momentum;
    const float* x; int64_t xs;
    const float* dz; int64_t dzs;
    const float* gamma; const float* beta;
    float* rmean; float* rvar; int64_t* nbt;
    float* mean; float* invstd;        // fwd: outputs; bwd: inputs
    float* z; int64_t zs;              // fwd output / bwd dx
    const float* res; int64_t rs;
    const double* in_slot;             // consumer side
    double* out_slot;                  // producer side
    float* dgamma; float* dbeta;
    float* racc; int64_t racs;         // bwd apply: racc[r,:] += dz[r,:] (the residual branch's gradient, accumulated in place)
    const float* o_mean; const float* o_invstd;   // stacked reduce: the OUTER BatchNorm's statistics
    // stacked forward: the outer norm
    const float* og; const float* ob; float oeps, omom; float* ormean; float* orvar; int64_t* onbt; float* omean; float* oinvstd;
};

__device__ __forceinline__ double slot_sum(const double* slot, int C, int which, int c) {
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < KPGNN_STAT_REPLICAS; ++r) s += slot[((int64_t)r * 2 + which) * C + c];
    return s;
}

// Block reduction of per-thread (a,b)[VEC] over the row lanes (fixed order), then 2C fp64 atomics into this block's replica.
template <int VEC, int G>
__device__ __forceinline__ void block_to_slot(double* slot, int C, const double (&a)[VEC], const double (&b)[VEC], int c0, bool col_ok) {
    __shared__ double red[2][kBlock / G][G * VEC];
    const int rl = threadIdx.x / G, sl = threadIdx.x % G;
    for (int q = 0; q < VEC; ++q) { red[0][rl][sl * VEC + q] = a[q]; red[1][rl][sl * VEC + q] = b[q]; }
    __syncthreads();
    // thread t < 2 * G * VEC: statistic t / (G*VEC), column t % (G*VEC)   (G * VEC <= 256 / 2 except VEC 4, G 64: loop)
    for (int t = threadIdx.x; t < 2 * G * VEC; t += kBlock) {
        const int which = t / (G * VEC), col = t - which * (G * VEC);
        if (col < C) {
            double s = 0.0;
            for (int r = 0; r < kBlock / G; ++r) s += red[which][r][col];
            atomicAdd(slot + ((int64_t)(blockIdx.x % KPGNN_STAT_REPLICAS) * 2 + which) * C + col, s);
        }
    }
    (void)c0; (void)col_ok;
}

// The same for NS statistics (NS even), two at a time through the same LDS: slot = double[replica][NS][C].
template <int VEC, int G, int NS>
__device__ __forceinline__ void block_to_slot_n(double* slot, int C, const double (&acc)[NS][VEC]) {
    __shared__ double red[2][kBlock / G][G * VEC];
    const int rl = threadIdx.x / G, sl = threadIdx.x % G;
#pragma unroll
    for (int pr = 0; pr < NS; pr += 2) {
        for (int q = 0; q < VEC; ++q) { red[0][rl][sl * VEC + q] = acc[pr][q]; red[1][rl][sl * VEC + q] = acc[pr + 1][q]; }
        __syncthreads();
        for (int t = threadIdx.x; t < 2 * G * VEC; t += kBlock) {
            const int which = t / (G * VEC), col = t - which * (G * VEC);
            if (col < C) {
                double s = 0.0;
                for (int r = 0; r < kBlock / G; ++r) s += red[which][r][col];
                atomicAdd(slot + ((int64_t)(blockIdx.x % KPGNN_STAT_REPLICAS) * NS + pr + which) * C + col, s);
            }
        }
        __syncthreads();
    }
}

// fwd stats: sum x, sum x^2 (fp64: no pivot needed)
template <int VEC, int G>
__global__ void __launch_bounds__(kBlock) bn_stats_kernel(BnParams p) {
    p.N = live_rows(p.N, p.n_dyn);
    const int rl = threadIdx.x / G, sl = threadIdx.x % G, c0 = sl * VEC;
    const bool col_ok = c0 < p.C;
    double a[VEC], b[VEC];
    for (int q = 0; q < VEC; ++q) { a[q] = 0.0; b[q] = 0.0; }
    if (col_ok) {
        // four rows per trip: the loads are independent, a one-row loop keeps a single request in flight per thread
        const int64_t step = (int64_t)gridDim.x * (kBlock / G);
        int64_t r = (int64_t)blockIdx.x * (kBlock / G) + rl;
        for (; r + 3 * step < p.N; r += 4 * step) {
            float v0[VEC], v1[VEC], v2[VEC], v3[VEC];
            ldv<VEC>(p.x + r * p.xs + c0, v0);
            ldv<VEC>(p.x + (r + step) * p.xs + c0, v1);
            ldv<VEC>(p.x + (r + 2 * step) * p.xs + c0, v2);
            ldv<VEC>(p.x + (r + 3 * step) * p.xs + c0, v3);
            for (int q = 0; q < VEC; ++q) {
                a[q] += v0[q]; b[q] = fma((double)v0[q], (double)v0[q], b[q]);
                a[q] += v1[q]; b[q] = fma((double)v1[q], (double)v1[q], b[q]);
                a[q] += v2[q]; b[q] = fma((double)v2[q], (double)v2[q], b[q]);
                a[q] += v3[q]; b[q] = fma((double)v3[q], (double)v3[q], b[q]);
            }
        }
        for (; r < p.N; r += step) {
            float v[VEC];
            ldv<VEC>(p.x + r * p.xs + c0, v);
            for (int q = 0; q < VEC; ++q) { a[q] += v[q]; b[q] = fma((double)v[q], (double)v[q], b[q]); }
        }
    }
    block_to_slot<VEC, G>(p.out_slot, p.C, a, b, c0, col_ok);
}

// fwd apply: every block finishes mean / invstd of the columns from the slot; block 0 publishes them and updates the
// running statistics.  OUT: the statistics of z go to out_slot (z is then the input of another BatchNorm).
template <int VEC, int G, bool OUT>
__global__ void __launch_bounds__(kBlock) bn_apply_kernel(BnParams p) {
    p.N = live_rows(p.N, p.n_dyn);
    __shared__ float cm[2][G * VEC];
    const int rl = threadIdx.x / G, sl = threadIdx.x % G, c0 = sl * VEC;
    const bool col_ok = c0 < p.C;
    for (int c = threadIdx.x; c < p.C; c += kBlock) {
        const double inv_n = 1.0 / (double)p.N;
        const double m1 = slot_sum(p.in_slot, p.C, 0, c) * inv_n;
        double var = slot_sum(p.in_slot, p.C, 1, c) * inv_n - m1 * m1;
        if (var < 0.0) var = 0.0;
        const float mean = (float)m1, istd = (float)(1.0 / sqrt(var + (double)p.eps));
        cm[0][c] = mean; cm[1][c] = istd;
        if (blockIdx.x == 0) {
            p.mean[c] = mean; p.invstd[c] = istd;
            if (p.rmean) {
                const double unb = p.N > 1 ? var * (double)p.N / (double)(p.N - 1) : var;
                p.rmean[c] = (1.f - p.momentum) * p.rmean[c] + p.momentum * mean;
                p.rvar[c] = (1.f - p.momentum) * p.rvar[c] + p.momentum * (float)unb;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && p.nbt) *p.nbt += 1;   // nn.BatchNorm1d.num_batches_tracked
    __syncthreads();
    double a[VEC], b[VEC];
    for (int q = 0; q < VEC; ++q) { a[q] = 0.0; b[q] = 0.0; }
    if (col_ok) {
        float mean[VEC], istd[VEC], g[VEC], bt[VEC];
        for (int q = 0; q < VEC; ++q) { mean[q] = cm[0][c0 + q]; istd[q] = cm[1][c0 + q]; }
        ldv<VEC>(p.gamma + c0, g);
        ldv<VEC>(p.beta + c0, bt);
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
            if (OUT) for (int q = 0; q < VEC; ++q) { a[q] += o[q]; b[q] = fma((double)o[q], (double)o[q], b[q]); }
            stv<VEC>(p.z + r * p.zs + c0, o);
        }
    }
    if (OUT) block_to_slot<VEC, G>(p.out_slot, p.C, a, b, c0, col_ok);
}

// bwd reduce: s0 = sum dy, s1 = sum dy * xhat  (dy = dz masked by the recomputed pre-activation when relu)
template <int VEC, int G>
__global__ void __launch_bounds__(kBlock) bn_bwd_reduce_kernel(BnParams p) {
    p.N = live_rows(p.N, p.n_dyn);
    const int rl = threadIdx.x / G, sl = threadIdx.x % G, c0 = sl * VEC;
    const bool col_ok = c0 < p.C;
    double a[VEC], b[VEC];
    for (int q = 0; q < VEC; ++q) { a[q] = 0.0; b[q] = 0.0; }
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
                b[q] = fma((double)dy[q], (double)xh, b[q]);
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
    block_to_slot<VEC, G>(p.out_slot, p.C, a, b, c0, col_ok);
}

// Stacked forward, pass 1: the statistics (sum z, sum z^2) of z = [relu](bn(x)) into out_slot WITHOUT writing z; every block
// finishes bn's mean / invstd from in_slot, block 0 publishes them and updates the running statistics.
template <int VEC, int G>
__global__ void __launch_bounds__(kBlock) bn_act_stats_kernel(BnParams p) {
    p.N = live_rows(p.N, p.n_dyn);
    __shared__ float cm[2][G * VEC];
    const int rl = threadIdx.x / G, sl = threadIdx.x % G, c0 = sl * VEC;
    const bool col_ok = c0 < p.C;
    for (int c = threadIdx.x; c < p.C; c += kBlock) {
        const double inv_n = 1.0 / (double)p.N;
        const double m1 = slot_sum(p.in_slot, p.C, 0, c) * inv_n;
        double var = slot_sum(p.in_slot, p.C, 1, c) * inv_n - m1 * m1;
        if (var < 0.0) var = 0.0;
        const float mean = (float)m1, istd = (float)(1.0 / sqrt(var + (double)p.eps));
        cm[0][c] = mean; cm[1][c] = istd;
        if (blockIdx.x == 0) {
            p.mean[c] = mean; p.invstd[c] = istd;
            if (p.rmean) {
                const double unb = p.N > 1 ? var * (double)p.N / (double)(p.N - 1) : var;
                p.rmean[c] = (1.f - p.momentum) * p.rmean[c] + p.momentum * mean;
                p.rvar[c] = (1.f - p.momentum) * p.rvar[c] + p.momentum * (float)unb;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && p.nbt) *p.nbt += 1;
    __syncthreads();
    double a[VEC], b[VEC];
    for (int q = 0; q < VEC; ++q) { a[q] = 0.0; b[q] = 0.0; }
    if (col_ok) {
        float mean[VEC], istd[VEC], g[VEC], bt[VEC];
        for (int q = 0; q < VEC; ++q) { mean[q] = cm[0][c0 + q]; istd[q] = cm[1][c0 + q]; }
        ldv<VEC>(p.gamma + c0, g);
        ldv<VEC>(p.beta + c0, bt);
        const int64_t step = (int64_t)gridDim.x * (kBlock / G);
        int64_t r = (int64_t)blockIdx.x * (kBlock / G) + rl;
        auto one = [&](const float (&v)[VEC]) {
            for (int q = 0; q < VEC; ++q) {
                float z = fmaf((v[q] - mean[q]) * istd[q], g[q], bt[q]);
                if (p.relu) z = fmaxf(z, 0.f);
                a[q] += z; b[q] = fma((double)z, (double)z, b[q]);
            }
        };
        for (; r + 3 * step < p.N; r += 4 * step) {
            float v0[VEC], v1[VEC], v2[VEC], v3[VEC];
            ldv<VEC>(p.x + r * p.xs + c0, v0);
            ldv<VEC>(p.x + (r + step) * p.xs + c0, v1);
            ldv<VEC>(p.x + (r + 2 * step) * p.xs + c0, v2);
            ldv<VEC>(p.x + (r + 3 * step) * p.xs + c0, v3);
            one(v0); one(v1); one(v2); one(v3);
        }
        for (; r < p.N; r += step) {
            float v[VEC];
            ldv<VEC>(p.x + r * p.xs + c0, v);
            one(v);
        }
    }
    block_to_slot<VEC, G>(p.out_slot, p.C, a, b, c0, col_ok);
}

// Stacked forward, pass 2: out = bn_o([relu](bn(x))) + residual; bn's mean / invstd are read from where pass 1 published
// them, bn_o's are finished from in_slot (the statistics of the intermediate) by every block.
template <int VEC, int G>
__global__ void __launch_bounds__(kBlock) bn_apply2_kernel(BnParams p) {
    p.N = live_rows(p.N, p.n_dyn);
    __shared__ float cm[2][G * VEC];
    const int rl = threadIdx.x / G, sl = threadIdx.x % G, c0 = sl * VEC;
    for (int c = threadIdx.x; c < p.C; c += kBlock) {
        const double inv_n = 1.0 / (double)p.N;
        const double m1 = slot_sum(p.in_slot, p.C, 0, c) * inv_n;
        double var = slot_sum(p.in_slot, p.C, 1, c) * inv_n - m1 * m1;
        if (var < 0.0) var = 0.0;
        const float mean = (float)m1, istd = (float)(1.0 / sqrt(var + (double)p.oeps));
        cm[0][c] = mean; cm[1][c] = istd;
        if (blockIdx.x == 0) {
            p.omean[c] = mean; p.oinvstd[c] = istd;
            if (p.ormean) {
                const double unb = p.N > 1 ? var * (double)p.N / (double)(p.N - 1) : var;
                p.ormean[c] = (1.f - p.omom) * p.ormean[c] + p.omom * mean;
                p.orvar[c] = (1.f - p.omom) * p.orvar[c] + p.omom * (float)unb;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && p.onbt) *p.onbt += 1;
    __syncthreads();
    if (c0 >= p.C) return;
    float mean[VEC], istd[VEC], g[VEC], bt[VEC], om[VEC], oi[VEC], og[VEC], ob[VEC];
    ldv<VEC>(p.mean + c0, mean); ldv<VEC>(p.invstd + c0, istd); ldv<VEC>(p.gamma + c0, g); ldv<VEC>(p.beta + c0, bt);
    ldv<VEC>(p.og + c0, og); ldv<VEC>(p.ob + c0, ob);
    for (int q = 0; q < VEC; ++q) { om[q] = cm[0][c0 + q]; oi[q] = cm[1][c0 + q]; }
    for (int64_t r = (int64_t)blockIdx.x * (kBlock / G) + rl; r < p.N; r += (int64_t)gridDim.x * (kBlock / G)) {
        float v[VEC], o[VEC];
        ldv<VEC>(p.x + r * p.xs + c0, v);
        for (int q = 0; q < VEC; ++q) {
            float z = fmaf((v[q] - mean[q]) * istd[q], g[q], bt[q]);
            if (p.relu) z = fmaxf(z, 0.f);
            o[q] = fmaf((z - om[q]) * oi[q], og[q], ob[q]);
        }
        if (p.res) {
            float rr[VEC];
            ldv<VEC>(p.res + r * p.rs + c0, rr);
            for (int q = 0; q < VEC; ++q) o[q] += rr[q];
        }
        stv<VEC>(p.z + r * p.zs + c0, o);
    }
}

// Stacked backward reduce: y -> z = [relu](bn_in(y)) -> h = bn_out(z) (+ residual), incoming gradient dh.  ONE pass over
// (dh, y) leaves the eight column sums from which the consumer (lin_fused.h, PRO 3) finishes BOTH BatchNorms' backward
// coefficients - the outer one's (sum dh, sum dh*xo) directly, the inner one's (sum dzm, sum dzm*xi with
// dz = a_o*(dh - s0/N - xo*s1/N), dzm = m*dz) by linearity:
//   T0 sum dh   T1 sum dh*xo   T2 sum m*dh   T3 sum m   T4 sum m*xo   T5 sum m*dh*xi   T6 sum m*xi   T7 sum m*xo*xi
// (xi = xhat of the inner norm, m = [its pre-activation > 0], xo = xhat of the outer norm of z, z recomputed from y).
// dz is never written: 3 launches (reduce, apply, reduce) and 100 MB per layer become 1 launch and 40 MB.
template <int VEC, int G>
__global__ void __launch_bounds__(kBlock) bn_bwd_reduce2_kernel(BnParams p) {
    p.N = live_rows(p.N, p.n_dyn);
    const int rl = threadIdx.x / G, sl = threadIdx.x % G, c0 = sl * VEC;
    const bool col_ok = c0 < p.C;
    double t[8][VEC];
    for (int n = 0; n < 8; ++n) for (int q = 0; q < VEC; ++q) t[n][q] = 0.0;
    if (col_ok) {
        float mean[VEC], istd[VEC], g[VEC], bt[VEC], om[VEC], oi[VEC];
        ldv<VEC>(p.mean + c0, mean); ldv<VEC>(p.invstd + c0, istd); ldv<VEC>(p.gamma + c0, g); ldv<VEC>(p.beta + c0, bt);
        ldv<VEC>(p.o_mean + c0, om); ldv<VEC>(p.o_invstd + c0, oi);
        const int64_t step = (int64_t)gridDim.x * (kBlock / G);
        int64_t r = (int64_t)blockIdx.x * (kBlock / G) + rl;
        auto one = [&](const float (&v)[VEC], const float (&dh)[VEC]) {
            for (int q = 0; q < VEC; ++q) {
                const float xi = (v[q] - mean[q]) * istd[q];
                const float pre = fmaf(xi, g[q], bt[q]);
                const bool m = !p.relu || pre > 0.f;
                const float z = (p.relu && pre <= 0.f) ? 0.f : pre;
                const float xo = (z - om[q]) * oi[q];
                const double d = (double)dh[q], dxo = (double)xo;
                t[0][q] += d;
                t[1][q] = fma(d, dxo, t[1][q]);
                if (m) {
                    const double dxi = (double)xi;
                    t[2][q] += d;
                    t[3][q] += 1.0;
                    t[4][q] += dxo;
                    t[5][q] = fma(d, dxi, t[5][q]);
                    t[6][q] += dxi;
                    t[7][q] = fma(dxo, dxi, t[7][q]);
                }
            }
        };
        auto racc = [&](int64_t row, const float (&dh)[VEC]) {
            float a[VEC];
            ldv<VEC>(p.racc + row * p.racs + c0, a);
            for (int q = 0; q < VEC; ++q) a[q] += dh[q];
            stv<VEC>(p.racc + row * p.racs + c0, a);
        };
        for (; r + step < p.N; r += 2 * step) {
            float v0[VEC], y0[VEC], v1[VEC], y1[VEC];
            ldv<VEC>(p.x + r * p.xs + c0, v0);
            ldv<VEC>(p.dz + r * p.dzs + c0, y0);
            ldv<VEC>(p.x + (r + step) * p.xs + c0, v1);
            ldv<VEC>(p.dz + (r + step) * p.dzs + c0, y1);
            one(v0, y0);
            one(v1, y1);
            if (p.racc) { racc(r, y0); racc(r + step, y1); }
        }
        for (; r < p.N; r += step) {
            float v[VEC], dh[VEC];
            ldv<VEC>(p.x + r * p.xs + c0, v);
            ldv<VEC>(p.dz + r * p.dzs + c0, dh);
            one(v, dh);
            if (p.racc) racc(r, dh);
        }
    }
    block_to_slot_n<VEC, G, 8>(p.out_slot, p.C, t);
}

template <int VEC, int G>
__global__ void __launch_bounds__(kBlock) bn_bwd_apply_kernel(BnParams p) {
    p.N = live_rows(p.N, p.n_dyn);
    __shared__ float cs[2][G * VEC];
    const int rl = threadIdx.x / G, sl = threadIdx.x % G, c0 = sl * VEC;
    for (int c = threadIdx.x; c < p.C; c += kBlock) {
        const float s0 = (float)slot_sum(p.in_slot, p.C, 0, c), s1 = (float)slot_sum(p.in_slot, p.C, 1, c);
        cs[0][c] = s0; cs[1][c] = s1;
        if (blockIdx.x == 0) { p.dbeta[c] = s0; p.dgamma[c] = s1; }
    }
    __syncthreads();
    if (c0 >= p.C) return;
    float mean[VEC], istd[VEC], g[VEC], bt[VEC], s0[VEC], s1[VEC];
    ldv<VEC>(p.mean + c0, mean); ldv<VEC>(p.invstd + c0, istd); ldv<VEC>(p.gamma + c0, g); ldv<VEC>(p.beta + c0, bt);
    for (int q = 0; q < VEC; ++q) { s0[q] = cs[0][c0 + q]; s1[q] = cs[1][c0 + q]; }
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
        if (p.racc) {                       // z = bn(x) + residual: d/dresidual is the incoming gradient itself
            float a[VEC], raw[VEC];
            ldv<VEC>(p.racc + r * p.racs + c0, a);
            ldv<VEC>(p.dz + r * p.dzs + c0, raw);
            for (int q = 0; q < VEC; ++q) a[q] += raw[q];
            stv<VEC>(p.racc + r * p.racs + c0, a);
        }
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

int stream_grid(int64_t N, int G, int cap_blocks) {
    const int64_t rows_per_block = kBlock / G;
    int64_t g = (N + rows_per_block * 4 - 1) / (rows_per_block * 4);   // >= 4 rows per thread
    if (g > cap_blocks) g = cap_blocks;
    return (int)(g < 1 ? 1 : g);
}

#define KP_BN_CASE(KERNEL, V, GG, GRID) case V * 100 + GG: hipLaunchKernelGGL((KERNEL<V, GG>), dim3(GRID), dim3(kBlock), 0, s, p); break;
#define KP_BN_SWITCH(KERNEL, GRID)                                                                                   \
    switch (vec * 100 + g) {                                                                                         \
        KP_BN_CASE(KERNEL, 4, 4, GRID) KP_BN_CASE(KERNEL, 4, 8, GRID) KP_BN_CASE(KERNEL, 4, 16, GRID)                \
        KP_BN_CASE(KERNEL, 4, 32, GRID) KP_BN_CASE(KERNEL, 4, 64, GRID)                                              \
        KP_BN_CASE(KERNEL, 2, 4, GRID) KP_BN_CASE(KERNEL, 2, 8, GRID) KP_BN_CASE(KERNEL, 2, 16, GRID)                \
        KP_BN_CASE(KERNEL, 2, 32, GRID) KP_BN_CASE(KERNEL, 2, 64, GRID)                                              \
        KP_BN_CASE(KERNEL, 1, 4, GRID) KP_BN_CASE(KERNEL, 1, 8, GRID) KP_BN_CASE(KERNEL, 1, 16, GRID)                \
        KP_BN_CASE(KERNEL, 1, 32, GRID) KP_BN_CASE(KERNEL, 1, 64, GRID)                                              \
        default: return fail(KPGNN_EINVAL, "batch norm: no kernel for vec=%d g=%d", vec, g);                         \
    }                                                                                                                \
    KPGNN_LAUNCH_CHECK(#KERNEL)

}  // namespace
}  // namespace kpgnn

using namespace kpgnn;

extern "C" int kpgnn_bn_fwd(const kpgnn_bn_desc* d, kpgnn_stream_t stream) {
    KPGNN_REQUIRE(d != nullptr, "bn_fwd: NULL descriptor");
    KPGNN_REQUIRE(d->N >= 1 && d->C >= 1, "bn_fwd: bad N=%lld C=%d", (long long)d->N, d->C);
    KPGNN_REQUIRE(d->x && d->gamma && d->beta && d->mean && d->invstd && d->z, "bn_fwd: NULL pointer");
    KPGNN_REQUIRE(d->x_stride >= d->C && d->z_stride >= d->C, "bn_fwd: bad strides");
    KPGNN_REQUIRE(d->stat_slot != nullptr, "bn_fwd: NULL stat_slot");
    int vec, g;
    int rc = bn_shape(d->C, {d->x, d->z, d->gamma, d->beta, d->residual}, {d->x_stride, d->z_stride, d->residual ? d->r_stride : 0}, &vec, &g);
    if (rc != KPGNN_OK) return rc;
    BnParams p = {};
    p.N = d->N; p.n_dyn = d->n_dyn; p.C = d->C; p.relu = d->relu; p.eps = d->eps; p.momentum = d->momentum;
    p.x = d->x; p.xs = d->x_stride; p.gamma = d->gamma; p.beta = d->beta; p.rmean = d->running_mean; p.rvar = d->running_var;
    p.mean = d->mean; p.invstd = d->invstd; p.z = d->z; p.zs = d->z_stride; p.res = d->residual; p.rs = d->r_stride;
    p.nbt = d->num_batches_tracked;
    hipStream_t s = (hipStream_t)stream;
    if (d->outer_gamma) {
        KPGNN_REQUIRE(d->stats_ready && d->out_slot && d->outer_beta && d->outer_mean && d->outer_invstd,
                      "bn_fwd(stacked): needs stats_ready, out_slot, outer_beta, outer_mean, outer_invstd");
        KPGNN_REQUIRE(!d->outer_running_mean == !d->outer_running_var, "bn_fwd(stacked): outer running statistics come in pairs");
        {   // the outer parameters join the alignment that decides the vector width
            int v2, g2;
            rc = bn_shape(d->C, {d->x, d->z, d->gamma, d->beta, d->residual, d->outer_gamma, d->outer_beta, d->mean, d->invstd},
                          {d->x_stride, d->z_stride, d->residual ? d->r_stride : 0}, &v2, &g2);
            if (rc != KPGNN_OK) return rc;
            vec = v2; g = g2;
        }
        p.in_slot = d->stat_slot; p.out_slot = d->out_slot;
        const int nstat = stream_grid(d->N, g, kProducerBlocks);
        KP_BN_SWITCH(bn_act_stats_kernel, nstat);
        p.in_slot = d->out_slot; p.out_slot = nullptr;
        p.og = d->outer_gamma; p.ob = d->outer_beta; p.oeps = d->outer_eps; p.omom = d->outer_momentum;
        p.ormean = d->outer_running_mean; p.orvar = d->outer_running_var; p.onbt = d->outer_num_batches_tracked;
        p.omean = d->outer_mean; p.oinvstd = d->outer_invstd;
        const int napply = stream_grid(d->N, g, device_facts().cu_count * 8);
        KP_BN_SWITCH(bn_apply2_kernel, napply);
        return KPGNN_OK;
    }
    if (!d->stats_ready) {
        p.out_slot = d->stat_slot;
        const int nstat = stream_grid(d->N, g, kProducerBlocks);
        KP_BN_SWITCH(bn_stats_kernel, nstat);
    }
    p.in_slot = d->stat_slot;
    p.out_slot = d->out_slot;
    if (d->out_slot) {
        const int napply = stream_grid(d->N, g, kProducerBlocks);
        switch (vec * 100 + g) {
#define KP_C(V, GG) case V * 100 + GG: hipLaunchKernelGGL((bn_apply_kernel<V, GG, true>), dim3(napply), dim3(kBlock), 0, s, p); break;
            KP_C(4, 4) KP_C(4, 8) KP_C(4, 16) KP_C(4, 32) KP_C(4, 64) KP_C(2, 4) KP_C(2, 8) KP_C(2, 16) KP_C(2, 32) KP_C(2, 64)
            KP_C(1, 4) KP_C(1, 8) KP_C(1, 16) KP_C(1, 32) KP_C(1, 64)
#undef KP_C
            default: return fail(KPGNN_EINVAL, "batch norm: no kernel for vec=%d g=%d", vec, g);
        }
    } else {
        const int napply = stream_grid(d->N, g, device_facts().cu_count * 8);
        switch (vec * 100 + g) {
#define KP_C(V, GG) case V * 100 + GG: hipLaunchKernelGGL((bn_apply_kernel<V, GG, false>), dim3(napply), dim3(kBlock), 0, s, p); break;
            KP_C(4, 4) KP_C(4, 8) KP_C(4, 16) KP_C(4, 32) KP_C(4, 64) KP_C(2, 4) KP_C(2, 8) KP_C(2, 16) KP_C(2, 32) KP_C(2, 64)
            KP_C(1, 4) KP_C(1, 8) KP_C(1, 16) KP_C(1, 32) KP_C(1, 64)
#undef KP_C
            default: return fail(KPGNN_EINVAL, "batch norm: no kernel for vec=%d g=%d", vec, g);
        }
    }
    KPGNN_LAUNCH_CHECK("bn_apply_kernel");
    return KPGNN_OK;
}

extern "C" int kpgnn_bn_bwd(const kpgnn_bn_bwd_desc* d, kpgnn_stream_t stream) {
    KPGNN_REQUIRE(d != nullptr, "bn_bwd: NULL descriptor");
    KPGNN_REQUIRE(d->N >= 1 && d->C >= 1, "bn_bwd: bad N=%lld C=%d", (long long)d->N, d->C);
    KPGNN_REQUIRE(d->x && d->dz && d->gamma && d->beta && d->mean && d->invstd, "bn_bwd: NULL pointer");
    KPGNN_REQUIRE(d->reduce_only || (d->dx && d->dgamma && d->dbeta), "bn_bwd: NULL output");
    KPGNN_REQUIRE(d->stat_slot != nullptr, "bn_bwd: NULL stat_slot");
    int vec, g;
    int rc = bn_shape(d->C, {d->x, d->dz, d->dx, d->gamma, d->beta, d->mean, d->invstd, d->dgamma, d->dbeta, d->residual_grad, d->outer_mean, d->outer_invstd},
                      {d->x_stride, d->dz_stride, d->dx ? d->dx_stride : 0, d->residual_grad ? d->rg_stride : 0}, &vec, &g);
    if (rc != KPGNN_OK) return rc;
    BnParams p = {};
    p.N = d->N; p.n_dyn = d->n_dyn; p.C = d->C; p.relu = d->relu;
    p.x = d->x; p.xs = d->x_stride; p.dz = d->dz; p.dzs = d->dz_stride; p.gamma = d->gamma; p.beta = d->beta;
    p.mean = const_cast<float*>(d->mean); p.invstd = const_cast<float*>(d->invstd);
    p.z = d->dx; p.zs = d->dx_stride; p.dgamma = d->dgamma; p.dbeta = d->dbeta;
    p.racc = d->residual_grad; p.racs = d->rg_stride;
    p.out_slot = d->stat_slot;
    p.in_slot = d->stat_slot;
    hipStream_t s = (hipStream_t)stream;
    const int nstat = stream_grid(d->N, g, kProducerBlocks);
    if (d->outer_mean) {
        KPGNN_REQUIRE(d->reduce_only && d->outer_invstd, "bn_bwd: the stacked reduce (outer_mean) needs reduce_only and outer_invstd");
        p.o_mean = d->outer_mean; p.o_invstd = d->outer_invstd;
        KP_BN_SWITCH(bn_bwd_reduce2_kernel, nstat);
        return KPGNN_OK;
    }
    KP_BN_SWITCH(bn_bwd_reduce_kernel, nstat);
    if (d->reduce_only) return KPGNN_OK;
    const int napply = stream_grid(d->N, g, device_facts().cu_count * 8);
    KP_BN_SWITCH(bn_bwd_apply_kernel, napply);
    return KPGNN_OK;
}
