// The MLPs' Linear + BatchNorm launches (lin_fused.h: y = f(x) W^T + b with BatchNorm work folded into the load and store phases)
// on the bf16 matrix cores through exact three-way bf16 splits (bf3.h; skeleton: linear_bf3.hip).  gfx950.
// Contract: include/kpgnn.h, kpgnn_linear_bn with math = KPGNN_MATH_AUTO and a workspace; same PRO / EPI semantics, same
// statistics slots as lin_fused_kernel, which stays the path for small batches and for KPGNN_MATH_F32.
//
// A block of 8 waves works through groups of 32 rows (blocks = CUs, ~6 groups each at the bench shape).  Waves 4-7 stage: the
// prologue arithmetic of PRO 1 / 2 / 3 happens on the float4 they fetched (PRO >= 2: two source tensors, the transformed rows
// also leave for the weight-gradient kernel), then the split into three bf16 planes.  Waves 0-3 multiply: the 32-column strip
// of the pre-split W stays in 21 register fragments for the whole launch; the accumulators hold an output tile with a lane per
// COLUMN, so the column sums of EPI 1 / 2 are lane-local fp64 adds in row order - no LDS round trip - and leave as 2 O atomics
// per block into the block's replica of the slot, like lin_fused's.
#include "bf3.h"
#include "lin_fused.h"

namespace kpgnn {
namespace {

constexpr int kF3Rows = 32;       // rows per pipeline step (one 32-row tile: 96-row steps ran 17-35 us per launch, stage and multiply of a block's ~2 steps barely overlapping)
constexpr int f3_pitch(int ks) { return ks <= 7 ? 120 : 136; }
constexpr int f3_buf(int ks) { return 3 * kF3Rows * f3_pitch(ks); }      // bf16 per buffer (three planes of one group)

template <int KS, int PRO, int EPI>
__global__ void __launch_bounds__(512, 1)
lin3f_kernel(LinFParams p, const uint4* __restrict__ wfrag) {
    p.N = live_rows(p.N, p.n_dyn);
    if (p.N <= 0) return;
    extern __shared__ __attribute__((aligned(16))) uint4 f3_lds[];
    constexpr int PK = f3_pitch(KS), BUF = f3_buf(KS), ROWS = kF3Rows;
    constexpr int NCG = 4 * KS, RLMIN = 256 / NCG;   // float4 column groups of a row (upper bound), row lanes of the staging waves
    constexpr int PF = (ROWS + RLMIN - 1) / RLMIN;   // rows per staging thread and group
    __bf16* pl = reinterpret_cast<__bf16*>(f3_lds);
    float* cin = reinterpret_cast<float*>(pl + 2 * BUF);       // PRO coefficients: [12][I]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = lane >> 5, c = lane & 31;
    const int O = p.O, I = p.I;
    float* cout = cin + 12 * I;                                // EPI 2 coefficients: [4][O]
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
    for (int i = tid; i < 2 * BUF / 8; i += 512) f3_lds[i] = make_uint4(0u, 0u, 0u, 0u);       // (the k padding stays zero)
    __syncthreads();
    const int64_t groups = (p.N + ROWS - 1) / ROWS;
    if ((int64_t)blockIdx.x >= groups) return;       // (uniform; under a dynamic row count the grid was sized for the capacity)
    const int G = (int)((groups - (int64_t)blockIdx.x + gridDim.x - 1) / gridDim.x);           // groups b, b + grid, ... (>= 1)

    if (wave >= 4) {
        // ---- staging waves: a thread owns ONE group of 4 columns (its per-column coefficients stay in registers, as in
        // lin_fused_kernel) and rows rl, rl + RL, ...; a wave's request is still a contiguous run of float4s.  Rows beyond the
        // group and the threads beyond RL * ncg repeat another thread's task - the same bytes stored twice - rather than branch.
        const int ptid = tid - 256;
        const int ncg = I >> 2, RL = 256 / ncg;
        const int cg = ptid % ncg, rl = min(ptid / ncg, RL - 1);
        int prow[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) prow[i] = min(rl + i * RL, ROWS - 1);
        const uint32_t sbytes = (uint32_t)I * 4u;                 // (x, x2, xt are contiguous [N, I])
        float4 pvs[4][PF], pus[4][PF];                    // four groups' requests in flight (a group is ~1 us of matrix work, a round trip 2-3)
        // (unconditional requests: clamped rows, clamped group index - see wgrad.hip)
        auto issue = [&](int g, float4 (&pv)[PF], float4 (&pu)[PF]) {
            g = min(g, G - 1);
            const int64_t r0 = ((int64_t)blockIdx.x + (int64_t)g * gridDim.x) * ROWS;
            const int lim = (int)min((int64_t)ROWS - 1, p.N - 1 - r0);
            const char* cp = reinterpret_cast<const char*>(p.x + r0 * I);
            const char* up = reinterpret_cast<const char*>((PRO >= 2 ? p.x2 : p.x) + r0 * I);
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const uint32_t off = __umul24((uint32_t)min(prow[i], lim), sbytes) + 16u * (uint32_t)cg;
                pv[i] = *reinterpret_cast<const float4*>(cp + off);
                if (PRO >= 2) pu[i] = *reinterpret_cast<const float4*>(up + off);
            }
        };
        auto commit = [&](int g, float4 (&pv)[PF], float4 (&pu)[PF]) {
            if (g >= G) return;                                   // (uniform; no request inside)
            const int64_t r0 = ((int64_t)blockIdx.x + (int64_t)g * gridDim.x) * ROWS;
            const int lim = (int)min((int64_t)ROWS - 1, p.N - 1 - r0);
            __bf16* buf = pl + (g & 1) * BUF;
            char* tp = PRO >= 2 ? reinterpret_cast<char*>(p.xt + r0 * I) : nullptr;
            float4 mean, istd, gm, bt, ai, k0, k1, om, oi, ao, q0, q1;
            if (PRO >= 1) { mean = ld4(cin + 4 * cg); istd = ld4(cin + I + 4 * cg); gm = ld4(cin + 2 * I + 4 * cg); bt = ld4(cin + 3 * I + 4 * cg); }
            if (PRO >= 2) { ai = ld4(cin + 4 * I + 4 * cg); k0 = ld4(cin + 5 * I + 4 * cg); k1 = ld4(cin + 6 * I + 4 * cg); }
            if (PRO == 3) { om = ld4(cin + 7 * I + 4 * cg); oi = ld4(cin + 8 * I + 4 * cg); ao = ld4(cin + 9 * I + 4 * cg);
                            q0 = ld4(cin + 10 * I + 4 * cg); q1 = ld4(cin + 11 * I + 4 * cg); }
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                float4 v = pv[i];
                if (PRO == 1) {
                    v.x = fmaf((v.x - mean.x) * istd.x, gm.x, bt.x); v.y = fmaf((v.y - mean.y) * istd.y, gm.y, bt.y);
                    v.z = fmaf((v.z - mean.z) * istd.z, gm.z, bt.z); v.w = fmaf((v.w - mean.w) * istd.w, gm.w, bt.w);
                    if (p.pro_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                }
                if (PRO >= 2) {
                    const float4 dz = pv[i], xs = pu[i];
#define KP_BWD1(f) { const float xh = (xs.f - mean.f) * istd.f; float d = dz.f; \
                     const float pre = fmaf(xh, gm.f, bt.f); \
                     if (PRO == 3) { const float zz = (p.pro_relu && pre <= 0.f) ? 0.f : pre; \
                                     const float xo = (zz - om.f) * oi.f; \
                                     d = fmaf(-xo, q1.f, fmaf(ao.f, d, -q0.f)); } \
                     if (p.pro_relu && pre <= 0.f) d = 0.f; \
                     v.f = fmaf(-xh, k1.f, fmaf(ai.f, d, -k0.f)); }
                    KP_BWD1(x) KP_BWD1(y) KP_BWD1(z) KP_BWD1(w)
#undef KP_BWD1
                    // (the transformed rows leave for the weight-gradient kernel; repeated tasks store the same bytes again)
                    *reinterpret_cast<float4*>(tp + __umul24((uint32_t)min(prow[i], lim), sbytes) + 16u * (uint32_t)cg) = v;
                }
                if (prow[i] > lim) v = make_float4(0.f, 0.f, 0.f, 0.f);               // (rows beyond N stay zero)
                bf3_u2 h0, m0, l0, h1, m1, l1;
                bf3_split2(bf3_f2{v.x, v.y}, h0, m0, l0);
                bf3_split2(bf3_f2{v.z, v.w}, h1, m1, l1);
                __bf16* q = buf + prow[i] * PK + 4 * cg;
                *reinterpret_cast<uint2*>(q) = make_uint2(bf3_pack(h0.x, h0.y), bf3_pack(h1.x, h1.y));
                *reinterpret_cast<uint2*>(q + ROWS * PK) = make_uint2(bf3_pack(m0.x, m0.y), bf3_pack(m1.x, m1.y));
                *reinterpret_cast<uint2*>(q + 2 * ROWS * PK) = make_uint2(bf3_pack(l0.x, l0.y), bf3_pack(l1.x, l1.y));
            }
        };
        issue(0, pvs[0], pus[0]);
        issue(1, pvs[1], pus[1]);
        issue(2, pvs[2], pus[2]);
        issue(3, pvs[3], pus[3]);
        commit(0, pvs[0], pus[0]);
        issue(4, pvs[0], pus[0]);
        __syncthreads();
        // group g + 1 is staged while group g is multiplied; its register set then takes the requests of group g + 5 (set = group
        // mod 4: the loop is unrolled by four so that the set is a compile-time choice)
        for (int g0 = 0; g0 < G; g0 += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int g = g0 + u;
                if (g < G) {                                      // (uniform)
                    commit(g + 1, pvs[(u + 1) & 3], pus[(u + 1) & 3]);
                    issue(g + 5, pvs[(u + 1) & 3], pus[(u + 1) & 3]);
                    __syncthreads();
                }
            }
        }
        return;
    }

    // ---- multiplying waves: output columns [32 wave, 32 wave + 32)
    // (here the multiplying waves are the critical path - 1620 cycles of matrix work + the epilogue per group against ~1300 of
    //  staging - so THEY get the SIMD's vector issue first; in wgrad.hip it is the other way round)
    __builtin_amdgcn_s_setprio(2);
    const int n = wave * 32 + c;
    const bool strip = wave * 32 < O;                 // (uniform)
    const bool col = n < O;
    bf3_x8 wb[KS][3];                                 // this lane's pieces of W[n][16 ks + 8 kg .. + 8], for the whole launch
    {
        const uint4* f = wfrag + (int64_t)wave * KS * 192 + lane;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            wb[ks][0] = __builtin_bit_cast(bf3_x8, f[ks * 192]); wb[ks][1] = __builtin_bit_cast(bf3_x8, f[ks * 192 + 64]);
            wb[ks][2] = __builtin_bit_cast(bf3_x8, f[ks * 192 + 128]);
        }
    }
    const float bias = (PRO < 2 && p.bias && col) ? p.bias[n] : 0.f;
    float em = 0.f, ei = 0.f, eg = 0.f, eb = 0.f;
    if (EPI == 2 && col) { em = cout[n]; ei = cout[O + n]; eg = cout[2 * O + n]; eb = cout[3 * O + n]; }
    double s0 = 0.0, s1 = 0.0;                        // EPI 1 / 2: this lane's column, its rows, in row order
    auto ld8 = [&](const __bf16* q) { return __builtin_bit_cast(bf3_x8, *reinterpret_cast<const uint4*>(q)); };
    const uint32_t lane_off = ((uint32_t)(4 * kg) * (uint32_t)O + (uint32_t)n) * 4u;
    const uint32_t row1 = (uint32_t)O * 4u, row5 = 5u * row1;
    __syncthreads();                                  // buffer 0 is staged
    // One 32-row tile at a time: its 16 e_x values (EPI 2) are requested before its 42 matrix instructions and used after them;
    // 16 accumulator registers instead of 48 next to the 84 of the strip.  acc[v] of lane (c, kg): row 32 m + (v & 3) + 8 (v >> 2)
    // + 4 kg of the group, column n: stores and loads go through a scalar base + a running 32-bit lane offset (linear_bf3.hip);
    // only the batch's last group tests rows.
    for (int g = 0; g < G; ++g) {
        const int64_t r0 = ((int64_t)blockIdx.x + (int64_t)g * gridDim.x) * ROWS;
        char* yb = reinterpret_cast<char*>(p.y + r0 * O);
        const char* xb = reinterpret_cast<const char*>((EPI == 2 ? p.e_x : p.y) + r0 * O);
        const int lane_rows = (int)min((int64_t)ROWS, p.N - r0) - 4 * kg;
        const bool full = r0 + ROWS <= p.N;           // (uniform)
        const __bf16* ap0 = pl + (g & 1) * BUF + c * PK + 8 * kg;
#pragma unroll
        for (int m = 0; m < ROWS / 32; ++m) {
            f32x16 acc;
            for (int v = 0; v < 16; ++v) acc[v] = 0.f;
            float ex[16];
            if (EPI == 2 && strip && col) {
                uint32_t off = lane_off + (uint32_t)(32 * m) * row1;
                asm volatile("" : "+v"(off));          // (opaque: otherwise all 48 offsets of a group are hoisted out of the group loop)
                if (full) {
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        ex[v] = *reinterpret_cast<const float*>(xb + off);
                        off += (v & 3) == 3 ? row5 : row1;
                    }
                } else {
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        ex[v] = 32 * m + (v & 3) + 8 * (v >> 2) < lane_rows ? *reinterpret_cast<const float*>(xb + off) : 0.f;
                        off += (v & 3) == 3 ? row5 : row1;
                    }
                }
            }
            if (strip) {
                const __bf16* ap = ap0 + m * 32 * PK;
                bf3_x8 ah = ld8(ap), am = ld8(ap + ROWS * PK), al = ld8(ap + 2 * ROWS * PK);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    bf3_x8 nh = ah, nm = am, nl = al;
                    if (ks + 1 < KS) { nh = ld8(ap + 16 * (ks + 1)); nm = ld8(ap + ROWS * PK + 16 * (ks + 1)); nl = ld8(ap + 2 * ROWS * PK + 16 * (ks + 1)); }
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, wb[ks][0], acc, 0, 0, 0);      // smallest terms first
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wb[ks][2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, wb[ks][1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, wb[ks][0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wb[ks][1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wb[ks][0], acc, 0, 0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    ah = nh; am = nm; al = nl;
                }
            }
            if (m == ROWS / 32 - 1) __syncthreads();  // these waves are done with buffer g & 1; group g + 1 is staged
            if (strip && col) {
                uint32_t off = lane_off + (uint32_t)(32 * m) * row1;
                asm volatile("" : "+v"(off));
                // (the tile's 16 values of this lane's column are summed in fp32, in row order, and join the fp64 column sums once
                //  per tile: 48 fp64 operations per tile made the epilogue longer than the tile's 42 matrix instructions - 1900
                //  against 1620 cycles; a 16-term fp32 partial carries ~1e-7 of unbiased rounding into sums of thousands of them)
                float t0 = 0.f, t1 = 0.f;
                if (full) {                            // (uniform: every group but the batch's last - no row tests, no exec masks)
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        float val = acc[v] + bias;
                        if (EPI == 2) {
                            const float xh = (ex[v] - em) * ei;
                            if (fmaf(xh, eg, eb) <= 0.f) val = 0.f;
                            t0 += val; t1 = fmaf(val, xh, t1);
                        }
                        if (EPI == 1) { t0 += val; t1 = fmaf(val, val, t1); }
                        *reinterpret_cast<float*>(yb + off) = val;
                        off += (v & 3) == 3 ? row5 : row1;
                    }
                } else {
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const bool in = 32 * m + (v & 3) + 8 * (v >> 2) < lane_rows;
                        float val = acc[v] + bias;
                        if (EPI == 2) {
                            const float xh = (ex[v] - em) * ei;
                            if (fmaf(xh, eg, eb) <= 0.f) val = 0.f;
                            if (in) { t0 += val; t1 = fmaf(val, xh, t1); }
                        }
                        if (EPI == 1 && in) { t0 += val; t1 = fmaf(val, val, t1); }
                        if (in) *reinterpret_cast<float*>(yb + off) = val;
                        off += (v & 3) == 3 ? row5 : row1;
                    }
                }
                if (EPI != 0) { s0 += (double)t0; s1 += (double)t1; }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (EPI != 0 && strip) {
        // the two row halves of a column meet by a lane exchange (fixed order), then 2 O fp64 atomics into this block's replica
        s0 += __shfl_xor(s0, 32, 64);
        s1 += __shfl_xor(s1, 32, 64);
        if (kg == 0 && col) {
            double* slot = p.out_slot + (int64_t)(blockIdx.x % KPGNN_STAT_REPLICAS) * 2 * O;
            atomicAdd(slot + n, s0);
            atomicAdd(slot + O + n, s1);
        }
    }
}

template <int PRO, int EPI>
int lin3f_launch(const LinFParams& p, const uint4* wfrag, hipStream_t s) {
    const int ks = (p.I + 15) / 16;
    const int64_t groups = (p.N + kF3Rows - 1) / kF3Rows;
    const int64_t cus = (int64_t)device_facts().cu_count;
    const int64_t grid = groups < cus ? groups : cus;
    const size_t lds = (size_t)2 * f3_buf(ks) * 2 + sizeof(float) * (12 * (size_t)p.I + 4 * (size_t)p.O);
#define KP_F3(KSV) do { \
        KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)lin3f_kernel<KSV, PRO, EPI>, lds)); \
        hipLaunchKernelGGL((lin3f_kernel<KSV, PRO, EPI>), dim3((unsigned)grid), dim3(512), lds, s, p, wfrag); } while (0)
    switch (ks) {
        case 2: KP_F3(2); break;
        case 4: KP_F3(4); break;
        case 6: KP_F3(6); break;
        case 7: KP_F3(7); break;
        default: return fail(KPGNN_ELIMIT, "linear_bn (bf16-split): I=%d does not fit the LDS plan", p.I);      // (I = 128: two buffers + coefficients > 160 KB)
    }
#undef KP_F3
    KPGNN_LAUNCH_CHECK("lin3f_kernel");
    return KPGNN_OK;
}

}  // namespace

// The bf16-split variant of kpgnn_linear_bn: p is what the fp32 launch would get; wfrag = the split copy of W (lin3_split_w).
int linear3_fused(const LinFParams& p, int pro, int epi, const uint4* wfrag, hipStream_t s) {
    switch (pro * 10 + epi) {
        case 0: return lin3f_launch<0, 0>(p, wfrag, s);
        case 1: return lin3f_launch<0, 1>(p, wfrag, s);
        case 10: return lin3f_launch<1, 0>(p, wfrag, s);
        case 11: return lin3f_launch<1, 1>(p, wfrag, s);
        case 20: return lin3f_launch<2, 0>(p, wfrag, s);
        case 22: return lin3f_launch<2, 2>(p, wfrag, s);
        case 32: return lin3f_launch<3, 2>(p, wfrag, s);
        default: return fail(KPGNN_ELIMIT, "linear_bn: combination pro=%d epi=%d is not instantiated", pro, epi);
    }
}

}  // namespace kpgnn
