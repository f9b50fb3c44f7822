// y = f(x) W^T + b on the fp32 matrix cores with BatchNorm work folded into the tile's load and store phases
// (gfx950).  Contract: include/kpgnn.h, kpgnn_linear_bn.
//
// The Linear-BatchNorm-ReLU x2 MLP of KPGINPlusConv / GINEConv (KPGINplus.py:25-30, gine.py:31-38) used to be
// 2 GEMMs + 2 x (stats, slab reduce, apply) launches forward and 2 x (reduce, slab reduce, apply, dx GEMM, weight
// GEMM, slab reduce) backward: 22 launches, each at least ~4.7 us, 6 of them extra passes over [N,H].  BatchNorm
// needs batch-wide column sums, i.e. a grid-wide dependency, but nothing says the sums have to be finished by a
// kernel of their own: here the PRODUCER of a tensor adds its blocks' partial column sums (fp64) to a small
// statistics slot with atomics and the CONSUMER of the tensor finishes mean / invstd from the slot in its prologue
// and applies the normalisation while it loads its tile.
//
//   PRO 0  x is used as is
//   PRO 1  x' = [relu]((x - mean) * invstd * gamma + beta), mean / invstd from `in_slot` (sum x, sum x^2);
//          block 0 publishes mean / invstd and updates the running statistics (nn.BatchNorm1d training forward)
//   PRO 2  BatchNorm backward on load: dy = gamma*invstd*(dzm - s0/N - xhat*s1/N), dzm = dz * [pre-activation > 0],
//          xhat from the saved forward input x2, (s0, s1) = (sum dzm, sum dzm*xhat) from `in_slot`; dy is also
//          written out (the weight-gradient kernel reads it); block 0 publishes dbeta = s0, dgamma = s1
//   EPI 0  y is stored as is
//   PRO 3  two stacked BatchNorms' backward on load (y -> z = relu(bn_in(y)) -> h = bn_out(z)): x is dh, x2 is y, in_slot
//          holds the eight sums of bn.hip's stacked reduce; dz = a_o*(dh - s0o/N - xo*s1o/N) is formed in registers (z and
//          xo recomputed from y), then PRO 2's arithmetic on it.  Writes dgamma / dbeta of both norms.
//   EPI 1  + column sums (sum y, sum y^2) of the result into `out_slot`            (statistics of the next BatchNorm)
//   EPI 2  y is masked by the ReLU of the PREVIOUS BatchNorm (recomputed from its saved input e_x) and the column
//          sums (sum ym, sum ym * xhat_e) go to `out_slot`                        (backward reduce of that BatchNorm)
//
// Statistics slot: double[KPGNN_STAT_REPLICAS][2][C], all zero before the producing launch.  Same-address atomics
// serialise at ~18 ns on MI355X (measured, scripts/ubench): a block adds to replica blockIdx % 8, so an address sees
// grid/8 adds; the consumer sums the 8 replicas.  Per block the sums are formed in a fixed order in fp64; only the
// order of the <= 64 fp64 adds per address varies between runs (differences below 2^-52 relative).
//
// Tile mechanics are those of the plain kernel this replaces (round 1: 21.8 us for [47k,104] x [104,104]): a wave
// keeps its 32-output strip of W as v_mfma_f32_32x32x2_f32 A-fragments, 32M-row x tiles go through LDS once, the
// result leaves through the same buffer as whole rows.  New: the elementwise phases map a thread to a fixed group of
// 4 columns (thread t -> column group t % (C/4), row lane t / (C/4)), so per-column coefficients and partial sums
// live in registers and a tile is still one contiguous, coalesced run of float4s.
#pragma once
#include "kpgnn_common.h"

namespace kpgnn {

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct LinFParams {
    const int32_t* n_dyn;
    int64_t N; int O, I, pitch, wt;
    const float* x; const float* w; const float* bias; float* y;
    // PRO 1 / 2
    const double* in_slot; const float* in_gamma; const float* in_beta; int pro_relu; float in_eps, momentum;
    float* in_mean; float* in_invstd;            // PRO 1: outputs (block 0); PRO 2: inputs
    float* rmean; float* rvar; int64_t* nbt;     // PRO 1: running statistics (or NULL)
    const float* x2;                             // PRO 2: forward input of the BatchNorm ([N,I], contiguous)
    float* xt;                                   // PRO 2: transformed tile written out ([N,I]) or NULL
    float* dgamma; float* dbeta;                 // PRO 2: outputs (block 0)
    const float* o_mean; const float* o_invstd; const float* o_gamma;   // PRO 3: the outer BatchNorm
    float* o_dgamma; float* o_dbeta;             // PRO 3: outputs (block 0)
    // EPI 1 / 2
    double* out_slot;
    const float* e_x; const float* e_mean; const float* e_invstd; const float* e_gamma; const float* e_beta;   // EPI 2
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }

// sum of the replicas of one slot entry
__device__ __forceinline__ double slot_sum(const double* slot, int C, int which, int c) {
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < KPGNN_STAT_REPLICAS; ++r) s += slot[((int64_t)r * 2 + which) * C + c];
    return s;
}

template <int KS, int M, int PRO, int EPI>
__global__ void __launch_bounds__(256, 2)
lin_fused_kernel(LinFParams p) {
    p.N = live_rows(p.N, p.n_dyn);
    if (p.N <= 0) return;                             // (only under a dynamic count of zero)
    extern __shared__ __attribute__((aligned(16))) float xl[];      // [32*M][pitch] tile, then the coefficient rows
    constexpr int ROWS = 32 * M;
    constexpr int I = 2 * KS, CGI = I / 4, RLI = 256 / CGI, NAI = CGI * RLI, PFI = (ROWS + RLI - 1) / RLI;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int kk = lane >> 5, c = lane & 31;
    const int O = p.O, pitch = p.pitch;
    float* cin = xl + ROWS * pitch;                   // PRO coefficients: [12][I] (rows 7..11: PRO 3's outer norm)
    float* cout = cin + 12 * I;                       // EPI 2 coefficients: [4][O]
    const int o = wave * 32 + c;
    // this wave's strip of the weight as MFMA A-fragments: a[ks] = W[o][2 ks + kk]
    float a[KS];
    if (p.wt) {                                       // w is [I][O]: lanes run along o, coalesced as is
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a[ks] = o < O ? p.w[(int64_t)(2 * ks + kk) * O + o] : 0.f;
    } else {                                          // w is [O][I]: every lane streams ITS row 16 B at a time
#pragma unroll
        for (int j = 0; j < KS / 2; ++j) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (o < O) v = ld4(p.w + (int64_t)o * I + 4 * j);
            a[2 * j] = kk ? v.y : v.x;
            a[2 * j + 1] = kk ? v.w : v.z;
        }
    }
    // ---- per-column coefficients (every block finishes the statistics it consumes from the slot)
    if (PRO == 1 && tid < I) {
        const double inv_n = 1.0 / (double)p.N;
        const double m1 = slot_sum(p.in_slot, I, 0, tid) * inv_n;
        double var = slot_sum(p.in_slot, I, 1, tid) * inv_n - m1 * m1;
        if (var < 0.0) var = 0.0;
        const float mean = (float)m1, istd = (float)(1.0 / sqrt(var + (double)p.in_eps));
        cin[tid] = mean; cin[I + tid] = istd; cin[2 * I + tid] = p.in_gamma[tid]; cin[3 * I + tid] = p.in_beta[tid];
        if (blockIdx.x == 0) {
            p.in_mean[tid] = mean; p.in_invstd[tid] = istd;
            if (p.rmean) {
                const double unb = p.N > 1 ? var * (double)p.N / (double)(p.N - 1) : var;
                p.rmean[tid] = (1.f - p.momentum) * p.rmean[tid] + p.momentum * mean;
                p.rvar[tid] = (1.f - p.momentum) * p.rvar[tid] + p.momentum * (float)unb;
            }
            if (tid == 0 && p.nbt) *p.nbt += 1;
        }
    }
    if (PRO == 2 && tid < I) {
        const double inv_n = 1.0 / (double)p.N;
        const double s0 = slot_sum(p.in_slot, I, 0, tid), s1 = slot_sum(p.in_slot, I, 1, tid);
        const float mean = p.in_mean[tid], istd = p.in_invstd[tid], g = p.in_gamma[tid];
        const float ai = g * istd;
        cin[tid] = mean; cin[I + tid] = istd; cin[2 * I + tid] = g; cin[3 * I + tid] = p.in_beta[tid];
        cin[4 * I + tid] = ai; cin[5 * I + tid] = ai * (float)(s0 * inv_n); cin[6 * I + tid] = ai * (float)(s1 * inv_n);
        if (blockIdx.x == 0) { p.dbeta[tid] = (float)s0; p.dgamma[tid] = (float)s1; }
    }
    if (PRO == 3 && tid < I) {
        const double inv_n = 1.0 / (double)p.N;
        double t[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            double a = 0.0;
#pragma unroll
            for (int r = 0; r < KPGNN_STAT_REPLICAS; ++r) a += p.in_slot[((int64_t)r * 8 + n) * I + tid];
            t[n] = a;
        }
        const float mean = p.in_mean[tid], istd = p.in_invstd[tid], g = p.in_gamma[tid];
        const float om = p.o_mean[tid], oi = p.o_invstd[tid];
        const float ai = g * istd, ao = p.o_gamma[tid] * oi;
        const double m0 = t[0] * inv_n, m1 = t[1] * inv_n;                  // outer: s0/N, s1/N
        const double s0 = (double)ao * (t[2] - m0 * t[3] - m1 * t[4]);        // inner: sum dzm
        const double s1 = (double)ao * (t[5] - m0 * t[6] - m1 * t[7]);        //        sum dzm * xhat_in
        cin[tid] = mean; cin[I + tid] = istd; cin[2 * I + tid] = g; cin[3 * I + tid] = p.in_beta[tid];
        cin[4 * I + tid] = ai; cin[5 * I + tid] = ai * (float)(s0 * inv_n); cin[6 * I + tid] = ai * (float)(s1 * inv_n);
        cin[7 * I + tid] = om; cin[8 * I + tid] = oi; cin[9 * I + tid] = ao;
        cin[10 * I + tid] = ao * (float)m0; cin[11 * I + tid] = ao * (float)m1;
        if (blockIdx.x == 0) {
            p.dbeta[tid] = (float)s0; p.dgamma[tid] = (float)s1;
            p.o_dbeta[tid] = (float)t[0]; p.o_dgamma[tid] = (float)t[1];
        }
    }
    if (EPI == 2 && tid < O) {
        cout[tid] = p.e_mean[tid]; cout[O + tid] = p.e_invstd[tid]; cout[2 * O + tid] = p.e_gamma[tid]; cout[3 * O + tid] = p.e_beta[tid];
    }
    if (PRO != 0 || EPI == 2) __syncthreads();
    // ---- input side: thread -> (column group, row lane); a tile is one contiguous run of NAI float4s per row-lane step
    const int cgi = tid % CGI, rli = tid / CGI;
    const bool act_i = tid < NAI;
    const int64_t tiles = (p.N + ROWS - 1) / ROWS;
    float4 pf[PFI];
    // The loads are UNCONDITIONAL, from rows clamped to N - 1 (threads without a slot and rows beyond N read a valid row whose
    // values only ever reach tile rows that are never stored or summed).  A predicated load is a branch, and with branches
    // around its vector-memory instructions the compiler cannot count them: it put `s_waitcnt vmcnt(0)` in front of the
    // first MFMA of every tile - i.e. it waited for the NEXT tile's rows before computing this one (round 2: "the phases of a
    // tile do not overlap").
    const int64_t lastrow = p.N - 1;
    auto issue = [&](int64_t tl) {                    // PRO 0 / 1: the next tile travels in registers
        const int64_t r0 = tl * ROWS;
#pragma unroll
        for (int j = 0; j < PFI; ++j) {
            int64_t r = r0 + rli + j * RLI;
            r = r < lastrow ? r : lastrow;
            pf[j] = ld4(p.x + r * I + 4 * cgi);
        }
    };
    auto commit = [&]() {
        float4 mean, istd, g, bt;
        if (PRO == 1) { mean = ld4(cin + 4 * cgi); istd = ld4(cin + I + 4 * cgi); g = ld4(cin + 2 * I + 4 * cgi); bt = ld4(cin + 3 * I + 4 * cgi); }
#pragma unroll
        for (int j = 0; j < PFI; ++j) {
            if (act_i && rli + j * RLI < ROWS) {
                float4 v = pf[j];
                if (PRO == 1) {
                    v.x = fmaf((v.x - mean.x) * istd.x, g.x, bt.x); v.y = fmaf((v.y - mean.y) * istd.y, g.y, bt.y);
                    v.z = fmaf((v.z - mean.z) * istd.z, g.z, bt.z); v.w = fmaf((v.w - mean.w) * istd.w, g.w, bt.w);
                    if (p.pro_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                }
                st4(xl + (rli + j * RLI) * pitch + 4 * cgi, v);
            }
        }
    };
    // PRO 2: two source tensors per tile; loaded, combined and committed at the start of the tile (two blocks per CU
    // overlap each other's phases; a register double-buffer of both would not fit next to the accumulators)
    auto load_bwd = [&](int64_t tl) {
        const int64_t r0 = tl * ROWS;
        const float4 mean = ld4(cin + 4 * cgi), istd = ld4(cin + I + 4 * cgi), g = ld4(cin + 2 * I + 4 * cgi),
                     bt = ld4(cin + 3 * I + 4 * cgi), ai = ld4(cin + 4 * I + 4 * cgi), k0 = ld4(cin + 5 * I + 4 * cgi),
                     k1 = ld4(cin + 6 * I + 4 * cgi);
        float4 om, oi, ao, q0, q1;
        if (PRO == 3) { om = ld4(cin + 7 * I + 4 * cgi); oi = ld4(cin + 8 * I + 4 * cgi); ao = ld4(cin + 9 * I + 4 * cgi);
                        q0 = ld4(cin + 10 * I + 4 * cgi); q1 = ld4(cin + 11 * I + 4 * cgi); }
        float4 dz[PFI], xs[PFI];
#pragma unroll
        for (int j = 0; j < PFI; ++j) {              // (unconditional, clamped rows: see `issue`; rows beyond N are zeroed below)
            int64_t r = r0 + rli + j * RLI;
            r = r < lastrow ? r : lastrow;
            dz[j] = ld4(p.x + r * I + 4 * cgi);
            xs[j] = ld4(p.x2 + r * I + 4 * cgi);
        }
#pragma unroll
        for (int j = 0; j < PFI; ++j) {
            if (act_i && rli + j * RLI < ROWS) {
                float4 v;
                const bool in = r0 + rli + j * RLI < p.N;
#define KP_BWD1(f) { const float xh = (xs[j].f - mean.f) * istd.f; float d = dz[j].f; \
                     const float pre = fmaf(xh, g.f, bt.f); \
                     if (PRO == 3) { const float zz = (p.pro_relu && pre <= 0.f) ? 0.f : pre; \
                                     const float xo = (zz - om.f) * oi.f; \
                                     d = fmaf(-xo, q1.f, fmaf(ao.f, d, -q0.f)); } \
                     if (p.pro_relu && pre <= 0.f) d = 0.f; \
                     v.f = in ? fmaf(-xh, k1.f, fmaf(ai.f, d, -k0.f)) : 0.f; }
                KP_BWD1(x) KP_BWD1(y) KP_BWD1(z) KP_BWD1(w)
#undef KP_BWD1
                st4(xl + (rli + j * RLI) * pitch + 4 * cgi, v);
                if (p.xt && in) st4(p.xt + r0 * I + 4 * tid + (int64_t)j * NAI * 4, v);
            }
        }
    };
    // ---- output side (O is a runtime value): same mapping over the O/4 column groups
    const int CGO = O >> 2, RLO = 256 / CGO, NAO = CGO * RLO;
    const int cgo = tid % CGO, rlo = tid / CGO;
    const bool act_o = tid < NAO;
    double s0[4] = {0.0, 0.0, 0.0, 0.0}, s1[4] = {0.0, 0.0, 0.0, 0.0};   // EPI 1 / 2 column partials of this thread

    // (a use of the weight strip in front of the loop: the wait for its loads happens HERE, once - otherwise the compiler,
    //  unable to tell at the loop header whether they have landed, waits for ALL vector-memory operations before the first MFMA
    //  of every tile, the next tile's prefetch included)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(a[ks]));
    // this lane's bias values (4 groups of 4 outputs), fetched once: inside the tile loop each of the four loads was a
    // dependent round trip in front of the LDS staging of its group
    // (the backward variants, PRO >= 2, carry no bias - the host rejects one - and have no registers to spare for it)
    constexpr bool HAS_BIAS = PRO < 2;
    float4 bias4[HAS_BIAS ? 4 : 1];
    if (HAS_BIAS) {
#pragma unroll
        for (int g = 0; g < (HAS_BIAS ? 4 : 1); ++g) {
            const int ob = wave * 32 + 8 * g + 4 * kk;
            bias4[g] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.bias && ob < O) bias4[g] = ld4(p.bias + ob);
        }
#pragma unroll
        for (int g = 0; g < (HAS_BIAS ? 4 : 1); ++g) asm volatile("" : "+v"(bias4[g].x), "+v"(bias4[g].y), "+v"(bias4[g].z), "+v"(bias4[g].w));
    }
    int64_t tile = blockIdx.x;
    if (PRO < 2) { if (tile < tiles) { issue(tile); commit(); } }
    else if (tile < tiles) load_bwd(tile);
    __syncthreads();
    for (; tile < tiles; tile += gridDim.x) {
        const bool more = tile + gridDim.x < tiles;
        if (PRO < 2 && more) issue(tile + gridDim.x);
        f32x16 acc[M];
#pragma unroll
        for (int m = 0; m < M; ++m)
            for (int v = 0; v < 16; ++v) acc[m][v] = 0.f;
        const float* b0 = xl + c * pitch + kk;
        // I == 2 * KS exactly (host): plain LDS reads the scheduler can hoist ahead of the MFMAs
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const float xv = b0[m * 32 * pitch + 2 * ks];
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks], xv, acc[m], 0, 0, 0);
            }
        }
        __syncthreads();                               // every wave is done reading the x tile: it becomes the y tile
        // C/D map: col = lane & 31 (tile row), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (output o)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ob = wave * 32 + 8 * g + 4 * kk;
            if (ob < O) {                              // O % 4 == 0 (host): the 4 outputs of a group are in or out together
                const float4 bb = HAS_BIAS ? bias4[HAS_BIAS ? g : 0] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int m = 0; m < M; ++m)            // pitch = 4 (mod 8) floats -> 16-B LDS accesses, no conflicts
                    st4(xl + (m * 32 + c) * pitch + ob,
                        make_float4(acc[m][4 * g] + bb.x, acc[m][4 * g + 1] + bb.y, acc[m][4 * g + 2] + bb.z, acc[m][4 * g + 3] + bb.w));
            }
        }
        __syncthreads();
        {
            const int64_t r0 = tile * ROWS;
            const int rows = (int)(p.N - r0 < ROWS ? p.N - r0 : ROWS);
            float* ybase = p.y + r0 * O + 4 * tid;
            float4 em, ei, eg, eb;
            if (EPI == 2) { em = ld4(cout + 4 * cgo); ei = ld4(cout + O + 4 * cgo); eg = ld4(cout + 2 * O + 4 * cgo); eb = ld4(cout + 3 * O + 4 * cgo); }
            if (act_o)
                for (int r = rlo, q = 0; r < rows; r += RLO, ++q) {
                    float4 v = ld4(xl + r * pitch + 4 * cgo);
                    if (EPI == 1) {
                        s0[0] += v.x; s0[1] += v.y; s0[2] += v.z; s0[3] += v.w;
                        s1[0] = fma((double)v.x, (double)v.x, s1[0]); s1[1] = fma((double)v.y, (double)v.y, s1[1]);
                        s1[2] = fma((double)v.z, (double)v.z, s1[2]); s1[3] = fma((double)v.w, (double)v.w, s1[3]);
                    }
                    if (EPI == 2) {
                        const float4 ex = ld4(p.e_x + r0 * O + 4 * tid + (int64_t)q * NAO * 4);
#define KP_EPI2(f, n) { const float xh = (ex.f - em.f) * ei.f; if (fmaf(xh, eg.f, eb.f) <= 0.f) v.f = 0.f; \
                        s0[n] += v.f; s1[n] = fma((double)v.f, (double)xh, s1[n]); }
                        KP_EPI2(x, 0) KP_EPI2(y, 1) KP_EPI2(z, 2) KP_EPI2(w, 3)
#undef KP_EPI2
                    }
                    st4(ybase + (int64_t)q * NAO * 4, v);
                }
        }
        __syncthreads();                               // the y tile is out: the buffer takes the next x tile
        if (more) { if (PRO < 2) commit(); else load_bwd(tile + gridDim.x); }
        __syncthreads();
    }
    if (EPI != 0) {
        // block partials: the row lanes of a column meet in LDS (fixed order), then 2*O fp64 atomics into this block's replica
        double* red = reinterpret_cast<double*>(xl);   // [RLO][2][O] doubles <= ROWS * pitch floats (host checks)
        if (act_o) {
#pragma unroll
            for (int n = 0; n < 4; ++n) { red[(rlo * 2 + 0) * O + 4 * cgo + n] = s0[n]; red[(rlo * 2 + 1) * O + 4 * cgo + n] = s1[n]; }
        }
        __syncthreads();
        if (tid < 2 * O) {
            const int which = tid / O, col = tid - which * O;
            double t = 0.0;
            for (int r = 0; r < RLO; ++r) t += red[(r * 2 + which) * O + col];
            atomicAdd(p.out_slot + ((int64_t)(blockIdx.x % KPGNN_STAT_REPLICAS) * 2 + which) * O + col, t);
        }
    }
}

struct LinLaunch { int m; unsigned grid; size_t lds; int pitch; };

// Tile rows (32 * m, the smallest that makes the launch one round over two blocks per CU), grid, LDS bytes and the row
// pitch (= 4 mod 8 floats: 16-B aligned rows, conflict-free 16-B accesses) shared by the x and the y view of the buffer.
inline LinLaunch lin_plan(int64_t N, int O, int I) {
    LinLaunch L;
    const int wmax = I > O ? I : O;
    L.pitch = wmax + ((4 - wmax % 8) + 8) % 8;
    const int64_t slots = (int64_t)device_facts().cu_count * 2;
    int m = (int)((N + slots * 32 - 1) / (slots * 32));
    L.m = m < 1 ? 1 : (m > 3 ? 3 : m);
    const int rows = 32 * L.m;
    // tile + 12 input-side + 4 output-side coefficient rows; the fp64 block reduction of the statistics reuses the tile
    size_t fl = (size_t)rows * L.pitch + 12 * (size_t)I + 4 * (size_t)O;
    const size_t red = 2 * (size_t)(256 / (O / 4)) * 2 * O;      // doubles, counted in floats
    if (fl < red) fl = red;
    L.lds = sizeof(float) * fl;
    const int64_t tiles = (N + rows - 1) / rows;
    const int64_t cap = L.m == 1 ? slots * 2 : slots;
    L.grid = (unsigned)(cap < tiles ? cap : tiles);
    return L;
}

template <int PRO, int EPI>
int lin_fused_launch(const LinFParams& p, hipStream_t s) {
    const LinLaunch L = lin_plan(p.N, p.O, p.I);
    LinFParams q = p;
    q.pitch = L.pitch;
    dim3 blk(256);
#define KP_LF2(KSV, MV) do { \
        KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)lin_fused_kernel<KSV, MV, PRO, EPI>, L.lds)); \
        hipLaunchKernelGGL((lin_fused_kernel<KSV, MV, PRO, EPI>), dim3(L.grid), blk, L.lds, s, q); } while (0)
#define KP_LF(KSV) do { if (L.m == 1) KP_LF2(KSV, 1); else if (L.m == 2) KP_LF2(KSV, 2); else KP_LF2(KSV, 3); } while (0)
    switch (p.I) {
        case 32: KP_LF(16); break;
        case 64: KP_LF(32); break;
        case 96: KP_LF(48); break;
        case 104: KP_LF(52); break;
        case 128: KP_LF(64); break;
        default: return fail(KPGNN_ELIMIT, "linear: I=%d is not one of 32, 64, 96, 104, 128 (the k-loop is fully unrolled)", p.I);
    }
#undef KP_LF
#undef KP_LF2
    KPGNN_LAUNCH_CHECK("lin_fused_kernel");
    return KPGNN_OK;
}

inline bool lin_supported_width(int I) { return I == 32 || I == 64 || I == 96 || I == 104 || I == 128; }

// one explicit instantiation set per translation unit (lin_fused_*.hip), so that the five variants compile in parallel
int lin_launch_plain(const LinFParams& p, hipStream_t s);       // PRO 0, EPI 0
int lin_launch_stats(const LinFParams& p, hipStream_t s);       // PRO 0, EPI 1
int lin_launch_bn_stats(const LinFParams& p, hipStream_t s);    // PRO 1, EPI 1
int lin_launch_bn(const LinFParams& p, hipStream_t s);          // PRO 1, EPI 0
int lin_launch_bwd_reduce(const LinFParams& p, hipStream_t s);  // PRO 2, EPI 2
int lin_launch_bwd(const LinFParams& p, hipStream_t s);         // PRO 2, EPI 0
int lin_launch_bwd2_reduce(const LinFParams& p, hipStream_t s); // PRO 3, EPI 2
// linear_bf3_fused.hip: the same launches on the bf16 matrix cores (wfrag: linear3_split_w's copy of W)
int linear3_fused(const LinFParams& p, int pro, int epi, const uint4* wfrag, hipStream_t s);

}  // namespace kpgnn
