// Fused K-hop aggregation for gfx950 (MI355X): forward and backward of
//     S[i,k,:] = sum_{a in segment(i,k)} ( x[col[a],k,:] + table_k[code[a],:] )
// plus the per-layer epilogue (peripheral add, (1+eps)x / GELU / ReLU + symmetric degree norm) and the
// optional geometric hop-combine.  Replaces the reference's materialised [E,K,D] chain
//   embedding -> index_select -> add -> masked_fill_ -> scatter-sum     (layers/KPGIN.py:90-118,
//   KPGINplus.py:64-88, KPGCN.py:96-126, gine.py:49-59; PyG MessagePassing.propagate).
//
// Mapping (wave = 64 lanes): a SUB-GROUP of G lanes (G = pow2 >= D/VEC) owns one node and walks its
// hop segments in order; lanes span the D feature columns with VEC-wide (up to 16 B) accesses, so one
// neighbour row is one coalesced burst.  Neighbour ids/codes of a segment are loaded G at a time by
// the sub-group (coalesced) and broadcast with ds_bpermute; rows are gathered 4 deep.  The two
// edge-code embedding tables live in LDS.  Summation order inside a segment is the edge-list order
// (the CPU reference's index_add_ order), so results are run-to-run deterministic.
//
// HBM-bound: per launch the algorithmic traffic is x + P + out (+pre) once, ids/codes once, tables once
// (DESIGN.md "Algorithmic bytes").  The gather re-reads (A*D*4 bytes) are meant to be served by L2:
// tiles of consecutive nodes (= same graph) are walked per XCD (kpgnn_common.h XcdTileWalk).
#include <initializer_list>

#include "kpgnn_common.h"

namespace kpgnn {
namespace {

constexpr int kBlock = 256;
constexpr int kMaxLdsTableBytes = 96 * 1024;

template <int VEC> struct Vec;
template <> struct Vec<1> { using T = float; };
template <> struct Vec<2> { using T = float2; };
template <> struct Vec<4> { using T = float4; };

template <int VEC> struct V {
    float v[VEC];
    __device__ __forceinline__ static V zero() { V r; for (int i = 0; i < VEC; ++i) r.v[i] = 0.f; return r; }
    __device__ __forceinline__ static V load(const float* p) {
        V r;
        typename Vec<VEC>::T t = *reinterpret_cast<const typename Vec<VEC>::T*>(p);
        const float* f = reinterpret_cast<const float*>(&t);
        for (int i = 0; i < VEC; ++i) r.v[i] = f[i];
        return r;
    }
    __device__ __forceinline__ void store(float* p) const {
        typename Vec<VEC>::T t;
        float* f = reinterpret_cast<float*>(&t);
        for (int i = 0; i < VEC; ++i) f[i] = v[i];
        *reinterpret_cast<typename Vec<VEC>::T*>(p) = t;
    }
    // streaming store (nt): for rows nobody re-reads in this launch - they should not displace the gathered rows in L2
    __device__ __forceinline__ void store_stream(float* p) const {
        for (int i = 0; i < VEC; ++i) __builtin_nontemporal_store(v[i], p + i);
    }
    // a row stored as bf16 (BF) or fp32: `p` is a byte address
    template <bool BF> __device__ __forceinline__ static V load_row(const char* p) {
        if (BF) { V r; ld_bf16<VEC>(p, r.v); return r; }
        return load(reinterpret_cast<const float*>(p));
    }
    __device__ __forceinline__ void add(const V& o) { for (int i = 0; i < VEC; ++i) v[i] += o.v[i]; }
    __device__ __forceinline__ void fma(float s, const V& o) { for (int i = 0; i < VEC; ++i) v[i] = fmaf(s, o.v[i], v[i]); }
};

__device__ __forceinline__ float gelu_exact(float x) {  // F.gelu(approximate='none'), erf to 5e-7 abs
    float e2;
    return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752440f, &e2));
}

struct FwdParams {
    const int32_t* n_dyn;
    int N, K, D, K_csr, n_code0, n_codek, mode, combine, bf;
    const int32_t* rowptr;
    const int32_t* col;
    const uint16_t* code;
    const float* dis;
    const float* x; int64_t x_sn, x_sk;
    const float* table0;
    const float* tablek;
    const float* periph; int64_t p_sn, p_sk;
    const float* eps;
    float* out; int64_t o_sn, o_sk;
    float* pre;
    const float* theta;
    const float* alphas;        // geometric combine computed in-kernel (theta is then this launch's OUTPUT via theta_out)
    float* theta_out;
    float* hout;
    const float* hinit;         // fused combine: initial value of the hop sum ([N,D], may alias hout) or NULL
    const float* hinit2;        // ... and a second addend ([N,D]) or NULL
    const float* xbias;
    const float* ptab; const int32_t* uid; int64_t uid_stride;
    const float* xs[16];        // per-hop inputs (x == NULL)
    int lds_theta, lds_ptab;    // floats of theta / ptab staged in LDS behind the code tables (TAB == 1), 0 = read from global
};

// TAB: 0 = no tables, 1 = tables in LDS, 2 = tables read from global (too large for LDS).
// GCN selects the weighted inner loop; the other epilogues are wave-uniform runtime switches.
// FAST: the KP-GIN+ training configuration (GELU epilogue, fused geometric combine, dictionary P, S saved) with its
// epilogue switches resolved at compile time.
// BF (FAST only): the gathered hop slots (xs) and the saved S (`pre`) are bf16 rows; sums, tables, theta, P, hout fp32.
template <int VEC, int G, bool GCN, int TAB, bool FAST = false, bool BF = false>
__global__ void __launch_bounds__(kBlock, FAST ? 5 : 4)
agg_fwd_kernel(const FwdParams p) {
    // (a local: writing to the by-value argument would move the whole struct - pointer arrays indexed at run time - to scratch)
    const int N_live = live_rows(p.N, p.n_dyn);
    const int MODE = FAST ? (int)KPGNN_MODE_GINPLUS : p.mode;
    const bool COMBINE = FAST ? true : p.combine != 0;
    extern __shared__ __attribute__((aligned(16))) float lds_tab[];
    const int D = p.D;
    const float* thp = p.theta;
    const float* ptp = p.ptab;
    if (TAB == 1) {
        const int n0 = p.n_code0 * D, nk = p.n_codek * D;
        for (int t = threadIdx.x; t < n0; t += kBlock) lds_tab[t] = p.table0[t];
        for (int t = threadIdx.x; t < nk; t += kBlock) lds_tab[n0 + t] = p.tablek[t];
        // theta and the peripheral dictionary ride along: the per-hop epilogue then has no dependent global load
        float* th_l = lds_tab + ((n0 + nk + 3) & ~3);
        float* pt_l = th_l + ((p.lds_theta + 3) & ~3);
        if (p.alphas) {
            // theta[k,d] = softmax_k(a (1-a)^k), a = sigmoid(alphas[d]) (combine.py:43-50): K*D values, rebuilt by every
            // block instead of a launch of their own; block 0 publishes them for the backward
            for (int d = threadIdx.x; d < D; d += kBlock) {
                const float a = 1.0f / (1.0f + __expf(-p.alphas[d]));
                const float q = 1.0f - a;
                float pw = 1.0f, mx = -INFINITY;
                for (int k = 0; k < p.K; ++k) { mx = fmaxf(mx, a * pw); pw *= q; }
                float sum = 0.f;
                pw = 1.0f;
                for (int k = 0; k < p.K; ++k) { sum += __expf(a * pw - mx); pw *= q; }
                const float inv = 1.0f / sum;
                pw = 1.0f;
                for (int k = 0; k < p.K; ++k) {
                    const float th = __expf(a * pw - mx) * inv;
                    th_l[k * D + d] = th;
                    if (blockIdx.x == 0) p.theta_out[(int64_t)k * D + d] = th;
                    pw *= q;
                }
            }
        } else
        for (int t = threadIdx.x; t < p.lds_theta; t += kBlock) th_l[t] = p.theta[t];
        for (int t = threadIdx.x; t < p.lds_ptab; t += kBlock) pt_l[t] = p.ptab[t];
        if (p.lds_theta) thp = th_l;
        if (p.lds_ptab) ptp = pt_l;
        __syncthreads();
    }
    const float* tab0 = TAB == 1 ? lds_tab : p.table0;
    const float* tabk = TAB == 1 ? lds_tab + p.n_code0 * D : p.tablek;

    constexpr int NODES = kBlock / G;       // nodes per tile
    const int sg = threadIdx.x / G;
    const int sl = threadIdx.x % G;
    const int lane = threadIdx.x & (kWave - 1);
    const int sg_lane0 = lane - sl;        // first lane of this sub-group inside its wave
    const int c0 = sl * VEC;
    const bool col_ok = c0 < D;
    const float eps1 = 1.0f + (p.eps ? p.eps[0] : 0.0f);
    const int64_t num_tiles = ((int64_t)N_live + NODES - 1) / NODES;
    constexpr uint32_t XB = BF ? 2u : 4u;                       // bytes per stored element of a gathered row
    const uint32_t xrow_b = (uint32_t)p.x_sn * XB, trow_b = (uint32_t)D * 4u;   // (host: N * x_sn * 4 < 2^32)
    const uint32_t lane_b = col_ok ? (uint32_t)c0 * 4u : 0u;   // idle lanes re-read column 0 and are dropped below
    const uint32_t lane_bx = col_ok ? (uint32_t)c0 * XB : 0u;

    int last_u = -1;                        // dictionary row held in registers
    V<VEC> prow = V<VEC>::zero();
    // (A variant that walks a node's whole neighbour list in one pass - hop from row-pointer compares - was
    //  measured slower: 134 us vs 109 us per launch at N = 47k, its 120 VGPRs cost a wave per SIMD.)
    for (XcdTileWalk w(num_tiles); w.valid(); w.next()) {
        const int64_t i = w.cur * NODES + sg;
        if (i >= N_live) continue;          // whole sub-group leaves together
        const int32_t* rp = p.rowptr + i * p.K_csr;
        V<VEC> hsum = V<VEC>::zero();
        if (!FAST && p.hinit && col_ok) hsum = V<VEC>::load(p.hinit + i * (int64_t)D + c0);       // (pull form of the backward gather)
        if (!FAST && p.hinit2 && col_ok) hsum.add(V<VEC>::load(p.hinit2 + i * (int64_t)D + c0));
        // The waves of this kernel sit in s_waitcnt 85 % of their cycles (PMC, profiles/r01): per hop there were three
        // DEPENDENT round trips (row pointer -> pair list -> neighbour rows, then uid -> dictionary row).  So the node's
        // K+1 row pointers, its K dictionary ids and its whole pair list (all hops, one chunk of G pairs at a time -
        // a ZINC node has ~21) are fetched up front by the lanes of the sub-group and handed out with ds_bpermute.
        const bool lane_meta = G > p.K;                    // K+1 row pointers fit the sub-group's lanes
        int myrp = 0, myuid = 0;
        if (lane_meta) {
            myrp = rp[sl <= p.K ? sl : p.K];
            if (FAST || (p.uid && !p.periph)) myuid = p.uid[i * p.uid_stride + (sl < p.K ? sl : 0)];
        }
        int beg = lane_meta ? __shfl(myrp, sg_lane0) : rp[0];
        const int end_all = lane_meta ? __shfl(myrp, sg_lane0 + p.K) : rp[p.K];
        // pair-list chunk [cbase, cbase+G): lane sl holds the byte offsets of pair cbase+sl (non-GCN path)
        int cbase = beg;
        uint32_t coff = 0, ctab = 0;
        if (!GCN) {
            const int idx = cbase + sl;
            if (idx < end_all) {
                coff = (uint32_t)p.col[idx] * xrow_b;
                if (TAB != 0) ctab = (uint32_t)p.code[idx] * trow_b;
            }
        }
        // Rows of the NEXT hop's first pairs are requested before the current hop is summed and finished, so the one
        // dependent round trip left per hop (the neighbour rows) overlaps the previous hop's epilogue.
        constexpr int PF = G >= 32 ? 4 : 2;   // (narrow rows, 4+ nodes per wave: deeper prefetch measured slower, 60 vs 49 us at D = 13)
        V<VEC> pr[PF];
        uint32_t prb[PF];
        int prn = 0;
        auto prefetch = [&](int kk, int bpos, int bend) {
            prn = min(PF, min(bend, cbase + G) - bpos);
            if (prn < 0) prn = 0;
            const char* xb = reinterpret_cast<const char*>(p.x ? p.x + (int64_t)kk * p.x_sk : p.xs[kk]);
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                if (u < prn) {
                    const int l = sg_lane0 + (bpos - cbase) + u;
                    const uint32_t o = __shfl(coff, l);
                    if (TAB != 0) prb[u] = __shfl(ctab, l);
                    pr[u] = V<VEC>::template load_row<BF>(xb + (size_t)(o + lane_bx));
                }
            }
        };
        int end_next = lane_meta ? __shfl(myrp, sg_lane0 + 1) : rp[1];
        if (!GCN) prefetch(0, beg, end_next);
        for (int k = 0; k < p.K; ++k) {
            const int end = end_next;
            if (GCN && k + 1 < p.K) end_next = lane_meta ? __shfl(myrp, sg_lane0 + k + 2) : rp[k + 2];
            const int uk = lane_meta ? __shfl(myuid, sg_lane0 + k) : 0;     // (shuffles need the whole sub-group active)
            const float* xk = (p.x ? p.x + (int64_t)k * p.x_sk : p.xs[k]) + c0;
            const float* tab = k == 0 ? tab0 : tabk;
            V<VEC> acc = V<VEC>::zero();
            float wacc = 0.f;  // GCN: sum of edge weights of the segment (for the constant x-bias term)
            // Each lane of the sub-group fetches ONE pair of the segment and turns it into byte offsets (row of x,
            // row of the code table) once; the gather loop then only broadcasts two 32-bit offsets per pair
            // (ds_bpermute) and adds the lane's column offset: no 64-bit / quarter-rate integer math per pair
            // (it was ~20 of the ~35 VALU slots a pair cost).  Two pairs per trip + a one-pair tail: segments hold
            // 2.7 pairs on average, a 4-deep body mostly ran masked.
            const char* xkb = reinterpret_cast<const char*>(p.x ? p.x + (int64_t)k * p.x_sk : p.xs[k]);   // wave-uniform
            const char* tabb = reinterpret_cast<const char*>(tab);
            if (!GCN) {
                // (sum the prefetched rows first, then reuse their registers for the next hop's prefetch: a copy would
                //  have to wait for the rows just the same and costs 20 VGPRs, i.e. a wave per SIMD)
                const int cn = prn;
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    if (u < cn) {
                        V<VEC> r = pr[u];
                        if (TAB != 0) r.add(V<VEC>::load(reinterpret_cast<const float*>(tabb + (size_t)(prb[u] + lane_b))));
                        acc.add(r);
                    }
                }
                if (k + 1 < p.K) {
                    end_next = lane_meta ? __shfl(myrp, sg_lane0 + k + 2) : rp[k + 2];
                    prefetch(k + 1, end, end_next);
                }
                int pos = beg + cn;
                while (pos < end) {
                    if (pos >= cbase + G) {                 // next chunk of the node's pair list (nodes with > G pairs)
                        cbase += G;
                        const int idx = cbase + sl;
                        coff = 0; ctab = 0;
                        if (idx < end_all) {
                            coff = (uint32_t)p.col[idx] * xrow_b;
                            if (TAB != 0) ctab = (uint32_t)p.code[idx] * trow_b;
                        }
                    }
                    const int lim = min(end, cbase + G) - cbase;
                    int t = pos - cbase;
                    for (; t + 1 < lim; t += 2) {
                        const int l0 = sg_lane0 + t, l1 = l0 + 1;
                        const uint32_t o0 = __shfl(coff, l0), o1 = __shfl(coff, l1);
                        uint32_t b0 = 0, b1 = 0;
                        if (TAB != 0) { b0 = __shfl(ctab, l0); b1 = __shfl(ctab, l1); }
                        V<VEC> r0 = V<VEC>::template load_row<BF>(xkb + (size_t)(o0 + lane_bx));
                        V<VEC> r1 = V<VEC>::template load_row<BF>(xkb + (size_t)(o1 + lane_bx));
                        if (TAB != 0) {
                            r0.add(V<VEC>::load(reinterpret_cast<const float*>(tabb + (size_t)(b0 + lane_b))));
                            r1.add(V<VEC>::load(reinterpret_cast<const float*>(tabb + (size_t)(b1 + lane_b))));
                        }
                        acc.add(r0);
                        acc.add(r1);
                    }
                    if (t < lim) {
                        const int l0 = sg_lane0 + t;
                        const uint32_t o0 = __shfl(coff, l0);
                        V<VEC> r0 = V<VEC>::template load_row<BF>(xkb + (size_t)(o0 + lane_bx));
                        if (TAB != 0) {
                            const uint32_t b0 = __shfl(ctab, l0);
                            r0.add(V<VEC>::load(reinterpret_cast<const float*>(tabb + (size_t)(b0 + lane_b))));
                        }
                        acc.add(r0);
                    }
                    pos = cbase + lim;
                }
            } else
            for (int base = beg; base < end; base += G) {
                const int idx = base + sl;
                uint32_t myoff = 0, mytab = 0;
                float myw = 1.0f;
                if (idx < end) {
                    const int myj = p.col[idx];
                    myoff = (uint32_t)myj * xrow_b;
                    if (TAB != 0) mytab = (uint32_t)p.code[idx] * trow_b;
                    if (GCN) myw = p.dis[(int64_t)myj * p.K_csr + k];
                }
                const int cnt = min(G, end - base);
                int t = 0;
                for (; t + 1 < cnt; t += 2) {
                    const int l0 = sg_lane0 + t, l1 = l0 + 1;
                    const uint32_t o0 = __shfl(myoff, l0), o1 = __shfl(myoff, l1);
                    uint32_t b0 = 0, b1 = 0;
                    if (TAB != 0) { b0 = __shfl(mytab, l0); b1 = __shfl(mytab, l1); }
                    float w0 = 1.f, w1 = 1.f;
                    if (GCN) { w0 = __shfl(myw, l0); w1 = __shfl(myw, l1); }
                    V<VEC> r0 = V<VEC>::load(reinterpret_cast<const float*>(xkb + (size_t)(o0 + lane_b)));
                    V<VEC> r1 = V<VEC>::load(reinterpret_cast<const float*>(xkb + (size_t)(o1 + lane_b)));
                    if (TAB != 0) {
                        r0.add(V<VEC>::load(reinterpret_cast<const float*>(tabb + (size_t)(b0 + lane_b))));
                        r1.add(V<VEC>::load(reinterpret_cast<const float*>(tabb + (size_t)(b1 + lane_b))));
                    }
                    if (GCN) { acc.fma(w0, r0); acc.fma(w1, r1); wacc += w0; wacc += w1; }
                    else { acc.add(r0); acc.add(r1); }
                }
                if (t < cnt) {
                    const int l0 = sg_lane0 + t;
                    const uint32_t o0 = __shfl(myoff, l0);
                    V<VEC> r0 = V<VEC>::load(reinterpret_cast<const float*>(xkb + (size_t)(o0 + lane_b)));
                    if (TAB != 0) {
                        const uint32_t b0 = __shfl(mytab, l0);
                        r0.add(V<VEC>::load(reinterpret_cast<const float*>(tabb + (size_t)(b0 + lane_b))));
                    }
                    if (GCN) { const float w0 = __shfl(myw, l0); acc.fma(w0, r0); wacc += w0; }
                    else acc.add(r0);
                }
            }
            const int seglen = end - beg;
            beg = end;
            if (!col_ok) continue;
            // ---- epilogue for (i,k)
            V<VEC> v = acc;
            // constant row added to every x row of hops >= 1: hopk_node_path_emb(pe_attr == 0), KPGIN.py:92-94
            const bool biased = p.xbias != nullptr && k >= 1;
            V<VEC> xb = V<VEC>::zero();
            if (biased) { xb = V<VEC>::load(p.xbias + c0); v.fma(GCN ? wacc : (float)seglen, xb); }
            if (GCN) {
                const float di = p.dis[i * p.K_csr + k];
                V<VEC> self = V<VEC>::load(xk + i * p.x_sn);
                self.add(xb);
                if (TAB != 0) self.add(V<VEC>::load(tab + 1 * D + c0));  // self-loop code 1 (KPGCN.py:87-89)
                v.fma(di, self);                                          // last term of the edge list
                for (int q = 0; q < VEC; ++q) v.v[q] *= di;
            }
            if (BF) st_bf16_stream<VEC>(reinterpret_cast<uint16_t*>(p.pre) + (i * p.K + k) * (int64_t)D + c0, v.v);
            else if (FAST || p.pre) v.store_stream(p.pre + (i * p.K + k) * (int64_t)D + c0);
            if (MODE == KPGNN_MODE_GINPLUS) { for (int q = 0; q < VEC; ++q) v.v[q] = gelu_exact(v.v[q]); }
            if (GCN) { for (int q = 0; q < VEC; ++q) v.v[q] = fmaxf(v.v[q], 0.f); }
            if (!FAST && p.periph) v.add(V<VEC>::load(p.periph + i * p.p_sn + (int64_t)k * p.p_sk + c0));
            else if (FAST || p.uid) {
                const int u = lane_meta ? uk : p.uid[i * p.uid_stride + k];
                if (FAST) v.add(V<VEC>::load(ptp + (int64_t)u * D + c0));        // (dictionary staged in LDS: no register copy)
                else { if (u != last_u) { prow = V<VEC>::load(ptp + (int64_t)u * D + c0); last_u = u; } v.add(prow); }  // mostly one row
            }
            if (MODE == KPGNN_MODE_GIN) { V<VEC> xs = V<VEC>::load(xk + i * p.x_sn); xs.add(xb); v.fma(eps1, xs); }
            if (COMBINE) {
                const V<VEC> th = V<VEC>::load(thp + k * D + c0);
                for (int q = 0; q < VEC; ++q) hsum.v[q] = fmaf(th.v[q], v.v[q], hsum.v[q]);
            } else {
                v.store_stream(p.out + i * p.o_sn + (int64_t)k * p.o_sk + c0);
            }
        }
        if (COMBINE && col_ok) hsum.store_stream(p.hout + i * (int64_t)D + c0);
    }
}

struct BwdParams {
    const int32_t* n_dyn;
    int N, K, D, K_csr, n_code0, n_codek, mode, bf;
    const int32_t* rowptr;
    const int32_t* col;
    const uint16_t* code;
    const float* dis;
    const float* g; int64_t g_sn, g_sk;
    const float* eps;
    float* gx; int64_t gx_sn, gx_sk;
    float* gtable0;
    float* gtablek;
    float* gxs[16];             // per-hop outputs (gx == NULL)
    uint32_t acc_mask;          // bit k: gxs[k] += instead of =
};

// TAB: 0 = no table grads, 1 = accumulate table grads in LDS then flush with global fp32 atomics,
//      2 = straight global atomics (tables too large for LDS)
// BF (the chunked gather, !GCN && TAB == 0, not GIN): g rows are bf16; the sums and gx stay fp32.
// GXACC (chunked gather with ONE [N,K,D] output): bit k of acc_mask adds to what gx[:, k, :] already holds.  A separate
// instantiation: the select between the two output forms inside the hop loop cost the slot path 10 us per launch.
template <int VEC, int G, bool GCN, int TAB, bool BF = false, bool GXACC = false>
__global__ void __launch_bounds__(kBlock, 6)
agg_bwd_kernel(const BwdParams p) {
    const int N_live = live_rows(p.N, p.n_dyn);
    const int MODE = p.mode;
    extern __shared__ __attribute__((aligned(16))) float lds_gt[];
    const int D = p.D;
    const int n0 = p.n_code0 * D, nk = p.n_codek * D;
    if (TAB == 1) {
        for (int t = threadIdx.x; t < n0 + nk; t += kBlock) lds_gt[t] = 0.f;
        __syncthreads();
    }
    float* gt0 = TAB == 1 ? lds_gt : p.gtable0;
    float* gtk = TAB == 1 ? lds_gt + n0 : p.gtablek;

    constexpr int NODES = kBlock / G;
    const int sg = threadIdx.x / G;
    const int sl = threadIdx.x % G;
    const int lane = threadIdx.x & (kWave - 1);
    const int sg_lane0 = lane - sl;
    const int c0 = sl * VEC;
    const bool col_ok = c0 < D;
    const float eps1 = 1.0f + (p.eps ? p.eps[0] : 0.0f);
    const int64_t num_tiles = ((int64_t)N_live + NODES - 1) / NODES;
    constexpr uint32_t GB = BF ? 2u : 4u;                       // bytes per stored element of g
    const uint32_t grow_b = (uint32_t)p.g_sn * GB;              // (host: N * g_sn * 4 < 2^32)
    const uint32_t lane_b = col_ok ? (uint32_t)c0 * GB : 0u;   // idle lanes re-read column 0 and are dropped below

    for (XcdTileWalk w(num_tiles); w.valid(); w.next()) {
        const int64_t j = w.cur * NODES + sg;
        if (j >= N_live) continue;
        const int32_t* rp = p.rowptr + j * p.K_csr;
        // (as in the forward: row pointers and the node's whole pair list are fetched up front by the sub-group's lanes)
        const bool lane_meta = G > p.K;
        int myrp = 0;
        if (lane_meta) myrp = rp[sl <= p.K ? sl : p.K];
        int beg = lane_meta ? __shfl(myrp, sg_lane0) : rp[0];
        const int end_all = lane_meta ? __shfl(myrp, sg_lane0 + p.K) : rp[p.K];
        constexpr bool CHUNKED = !GCN && TAB == 0;
        int cbase = beg;
        uint32_t coff = 0;
        if (CHUNKED) {
            const int idx = cbase + sl;
            // (a streaming hint on these pair-list loads: within noise at D = 104, 32 -> 40 us at D = 13; on the forward's
            //  col / code loads 69 -> 72 us - not used)
            if (idx < end_all) coff = (uint32_t)p.col[idx] * grow_b;
        }
        constexpr int PF = G >= 32 ? 4 : 2;
        V<VEC> pr[PF];
        int prn = 0;
        auto prefetch = [&](int kk, int bpos, int bend) {      // rows of hop kk's first pairs, requested one hop ahead
            prn = min(PF, min(bend, cbase + G) - bpos);
            if (prn < 0) prn = 0;
            const char* gb = reinterpret_cast<const char*>(p.g) + (int64_t)kk * p.g_sk * GB;
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                if (u < prn) {
                    const uint32_t o = __shfl(coff, sg_lane0 + (bpos - cbase) + u);
                    pr[u] = V<VEC>::template load_row<BF>(gb + (size_t)(o + lane_b));
                }
            }
        };
        // gfx9 has ONE counter for vector loads and stores and they may complete out of order with respect to each other,
        // so any wait for a load while a store is pending is a vmcnt(0): storing hop k's row right before waiting for hop
        // k+1's prefetched rows put a full store round trip on every hop's critical path.  The store is therefore held
        // back one hop and issued together with the next prefetch (both then share one wait); the epilogue's own operands
        // (GIN self row, old value of an accumulating slot) are requested one hop ahead for the same reason.
        V<VEC> pend = V<VEC>::zero(), nself = V<VEC>::zero(), nold = V<VEC>::zero();
        float* pend_dst = nullptr;
        auto epi_prefetch = [&](int kk) {
            if (!CHUNKED || !col_ok) return;
            if (MODE == KPGNN_MODE_GIN) nself = V<VEC>::load(p.g + (int64_t)kk * p.g_sk + c0 + j * p.g_sn);
            if (GXACC) { if ((p.acc_mask >> kk) & 1u) nold = V<VEC>::load(p.gx + (int64_t)kk * p.gx_sk + j * p.gx_sn + c0); }
            else if (!p.gx && ((p.acc_mask >> kk) & 1u)) nold = V<VEC>::load(p.gxs[kk] + j * p.gx_sn + c0);
        };
        int end_next = lane_meta ? __shfl(myrp, sg_lane0 + 1) : rp[1];
        if (CHUNKED) { prefetch(0, beg, end_next); epi_prefetch(0); }
        for (int k = 0; k < p.K; ++k) {
            const int end = end_next;
            if (k + 1 < p.K) end_next = lane_meta ? __shfl(myrp, sg_lane0 + k + 2) : rp[k + 2];
            const float* gk = p.g + (int64_t)k * p.g_sk + c0;
            float* gt = k == 0 ? gt0 : gtk;
            const float dj = GCN ? p.dis[j * p.K_csr + k] : 1.0f;
            V<VEC> acc = V<VEC>::zero();
            // (same pair walk as the forward: per-pair byte offsets are made once per fetched entry, the gather loop
            //  broadcasts them and adds the lane's column offset)
            const char* gkb = reinterpret_cast<const char*>(p.g) + (int64_t)k * p.g_sk * GB;   // wave-uniform
            if (CHUNKED) {
                // (the prefetched rows are summed in place and their registers reused for the next prefetch: a copy would
                //  wait for them just the same)
                const int cn = prn;
#pragma unroll
                for (int u = 0; u < PF; ++u)
                    if (u < cn) acc.add(pr[u]);
                if (MODE == KPGNN_MODE_GIN) acc.fma(eps1, nself);      // (this hop's epilogue operands arrived with its rows)
                if (GXACC ? ((p.acc_mask >> k) & 1u) != 0 : (!p.gx && ((p.acc_mask >> k) & 1u))) acc.add(nold);
                if (pend_dst) { pend.store_stream(pend_dst); pend_dst = nullptr; }   // previous hop's result
                if (k + 1 < p.K) { prefetch(k + 1, end, end_next); epi_prefetch(k + 1); }
                int pos = beg + cn;
                while (pos < end) {
                    if (pos >= cbase + G) {
                        cbase += G;
                        const int idx = cbase + sl;
                        coff = 0;
                        if (idx < end_all) coff = (uint32_t)p.col[idx] * grow_b;
                    }
                    const int lim = min(end, cbase + G) - cbase;
                    int t = pos - cbase;
                    for (; t + 1 < lim; t += 2) {
                        const int l0 = sg_lane0 + t, l1 = l0 + 1;
                        const uint32_t o0 = __shfl(coff, l0), o1 = __shfl(coff, l1);
                        V<VEC> r0 = V<VEC>::template load_row<BF>(gkb + (size_t)(o0 + lane_b));
                        V<VEC> r1 = V<VEC>::template load_row<BF>(gkb + (size_t)(o1 + lane_b));
                        acc.add(r0);
                        acc.add(r1);
                    }
                    if (t < lim) {
                        const uint32_t o0 = __shfl(coff, sg_lane0 + t);
                        acc.add(V<VEC>::template load_row<BF>(gkb + (size_t)(o0 + lane_b)));
                    }
                    pos = cbase + lim;
                }
            } else
            for (int base = beg; base < end; base += G) {
                const int idx = base + sl;
                uint32_t myoff = 0;
                int myc = 0;
                float myw = 1.0f;
                if (idx < end) {
                    const int myi = p.col[idx];
                    myoff = (uint32_t)myi * grow_b;
                    if (TAB != 0) myc = p.code[idx];
                    if (GCN) myw = dj * p.dis[(int64_t)myi * p.K_csr + k];
                }
                const int cnt = min(G, end - base);
                int t = 0;
                for (; t + 1 < cnt; t += 2) {
                    const int l0 = sg_lane0 + t, l1 = l0 + 1;
                    const uint32_t o0 = __shfl(myoff, l0), o1 = __shfl(myoff, l1);
                    V<VEC> r0 = V<VEC>::load(reinterpret_cast<const float*>(gkb + (size_t)(o0 + lane_b)));
                    V<VEC> r1 = V<VEC>::load(reinterpret_cast<const float*>(gkb + (size_t)(o1 + lane_b)));
                    if (GCN) {
                        const float w0 = __shfl(myw, l0), w1 = __shfl(myw, l1);
                        for (int q = 0; q < VEC; ++q) { r0.v[q] *= w0; r1.v[q] *= w1; }
                    }
                    acc.add(r0);
                    acc.add(r1);
                    int c0c = 0, c1c = 0;
                    if (TAB != 0) { c0c = __shfl(myc, l0); c1c = __shfl(myc, l1); }   // (all lanes take part in the shuffle)
                    if (TAB != 0 && col_ok) {
                        float* d0 = gt + c0c * D + c0;
                        float* d1 = gt + c1c * D + c0;
                        for (int q = 0; q < VEC; ++q) { atomicAdd(d0 + q, r0.v[q]); atomicAdd(d1 + q, r1.v[q]); }
                    }
                }
                if (t < cnt) {
                    const int l0 = sg_lane0 + t;
                    const uint32_t o0 = __shfl(myoff, l0);
                    V<VEC> r0 = V<VEC>::load(reinterpret_cast<const float*>(gkb + (size_t)(o0 + lane_b)));
                    if (GCN) { const float w0 = __shfl(myw, l0); for (int q = 0; q < VEC; ++q) r0.v[q] *= w0; }
                    acc.add(r0);
                    int c0c = 0;
                    if (TAB != 0) c0c = __shfl(myc, l0);
                    if (TAB != 0 && col_ok) {
                        float* d0 = gt + c0c * D + c0;
                        for (int q = 0; q < VEC; ++q) atomicAdd(d0 + q, r0.v[q]);
                    }
                }
            }
            beg = end;
            if (!col_ok) continue;
            if (!CHUNKED && MODE == KPGNN_MODE_GIN) acc.fma(eps1, V<VEC>::load(gk + j * p.g_sn));
            if (GCN) {
                V<VEC> self = V<VEC>::load(gk + j * p.g_sn);
                for (int q = 0; q < VEC; ++q) self.v[q] *= dj * dj;
                acc.add(self);
                if (TAB != 0) {
                    float* dst = gt + 1 * D + c0;
                    for (int q = 0; q < VEC; ++q) atomicAdd(dst + q, self.v[q]);
                }
            }
            float* dst = (p.gx ? p.gx + (int64_t)k * p.gx_sk : p.gxs[k]) + j * p.gx_sn + c0;
            if (CHUNKED) {                                   // (one sub-group owns the row: no atomics)
                pend = acc;
                pend_dst = dst;
            } else {
                if ((GXACC || !p.gx) && ((p.acc_mask >> k) & 1u)) acc.add(V<VEC>::load(dst));
                acc.store(dst);
            }
        }
        if (pend_dst) pend.store_stream(pend_dst);                  // last hop: overlaps the next node's metadata loads
    }
    if (TAB == 1) {
        __syncthreads();
        for (int t = threadIdx.x; t < n0; t += kBlock) { const float v = lds_gt[t]; if (v != 0.f) atomicAdd(p.gtable0 + t, v); }
        for (int t = threadIdx.x; t < nk; t += kBlock) { const float v = lds_gt[n0 + t]; if (v != 0.f) atomicAdd(p.gtablek + t, v); }
    }
}

// ---------------------------------------------------------------------------------------------- dispatch
struct Shape { int vec, g; };

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }
bool aligned8(const void* p) { return ((uintptr_t)p & 7) == 0; }

int pick_vec(int D, std::initializer_list<const void*> ptrs, std::initializer_list<int64_t> strides) {
    int vec = 4;
    if (D % 4 != 0) vec = (D % 2 == 0) ? 2 : 1;
    for (const void* q : ptrs) {
        if (!q) continue;
        if (vec == 4 && !aligned16(q)) vec = aligned8(q) ? 2 : 1;
        if (vec == 2 && !aligned8(q)) vec = 1;
    }
    for (int64_t s : strides) {
        if (vec == 4 && s % 4 != 0) vec = (s % 2 == 0) ? 2 : 1;
        if (vec == 2 && s % 2 != 0) vec = 1;
    }
    return vec;
}

int pick_group(int lanes_needed) {
    int g = 4;
    while (g < lanes_needed) g <<= 1;
    return g;
}

unsigned pick_grid(int64_t num_tiles, int blocks_per_cu) {
    const int64_t cap = (int64_t)device_facts().cu_count * blocks_per_cu;
    int64_t g = num_tiles < cap ? num_tiles : cap;
    if (g >= kNumXcd) g = g / kNumXcd * kNumXcd;  // XcdTileWalk wants a multiple of 8
    return (unsigned)(g > 0 ? g : 1);
}

template <int VEC, int G, bool GCN, int TAB, bool FAST = false, bool BF = false>
int launch_fwd(const FwdParams& p, size_t lds, hipStream_t s) {
    const int64_t tiles = ((int64_t)p.N + (kBlock / G) - 1) / (kBlock / G);
    if (lds > 64 * 1024)
        KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)agg_fwd_kernel<VEC, G, GCN, TAB, FAST, BF>, lds));
    const int nb = resident_blocks(agg_fwd_kernel<VEC, G, GCN, TAB, FAST, BF>, kBlock, lds);
    const unsigned grid = pick_grid(tiles, nb > 0 ? nb : 4);
    hipLaunchKernelGGL((agg_fwd_kernel<VEC, G, GCN, TAB, FAST, BF>), dim3(grid), dim3(kBlock), lds, s, p);
    KPGNN_LAUNCH_CHECK("agg_fwd_kernel");
    return KPGNN_OK;
}

template <int VEC, int G>
int launch_fwd_mode(const FwdParams& p, int tab, size_t lds, hipStream_t s) {
    const bool gcn = p.mode == KPGNN_MODE_GCN;
    const bool fast = tab == 1 && p.mode == KPGNN_MODE_GINPLUS && p.combine && !p.periph && p.uid && p.ptab && p.pre;
    if (p.bf) {     // bf16 rows: one instantiation family only (the KP-GIN+ training epilogue, 4 elements per lane)
        if constexpr (VEC == 4) {
            if (fast && !p.x) return launch_fwd<VEC, G, false, 1, true, true>(p, lds, s);
        }
        return fail(KPGNN_EINVAL, "aggregate_fwd: bf16 storage needs the fused KP-GIN+ epilogue (GELU, theta, dictionary P, "
                    "code tables in LDS, S saved), per-hop inputs and D %% 4 == 0");
    }
    switch (tab) {
        case 0: return gcn ? launch_fwd<VEC, G, true, 0>(p, 0, s) : launch_fwd<VEC, G, false, 0>(p, 0, s);
        case 1: {
            if (gcn) return launch_fwd<VEC, G, true, 1>(p, lds, s);
            return fast ? launch_fwd<VEC, G, false, 1, true>(p, lds, s) : launch_fwd<VEC, G, false, 1>(p, lds, s);
        }
        default: return gcn ? launch_fwd<VEC, G, true, 2>(p, 0, s) : launch_fwd<VEC, G, false, 2>(p, 0, s);
    }
}

template <int VEC, int G, bool GCN, int TAB, bool BF = false, bool GXACC = false>
int launch_bwd(const BwdParams& p, size_t lds, hipStream_t s) {
    const int64_t tiles = ((int64_t)p.N + (kBlock / G) - 1) / (kBlock / G);
    if (lds > 64 * 1024)
        KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)agg_bwd_kernel<VEC, G, GCN, TAB, BF, GXACC>, lds));
    const int nb = resident_blocks(agg_bwd_kernel<VEC, G, GCN, TAB, BF, GXACC>, kBlock, lds);
    const unsigned grid = pick_grid(tiles, nb > 0 ? nb : 4);
    hipLaunchKernelGGL((agg_bwd_kernel<VEC, G, GCN, TAB, BF, GXACC>), dim3(grid), dim3(kBlock), lds, s, p);
    KPGNN_LAUNCH_CHECK("agg_bwd_kernel");
    return KPGNN_OK;
}

template <int VEC, int G>
int launch_bwd_mode(const BwdParams& p, int tab, size_t lds, hipStream_t s) {
    const bool gcn = p.mode == KPGNN_MODE_GCN;
    if (p.bf) {     // bf16 rows of g: the atomics-free gather only (no GCN weights, no table grads, no GIN self term)
        if constexpr (VEC == 4) {
            if (!gcn && tab == 0 && p.mode != KPGNN_MODE_GIN) return launch_bwd<VEC, G, false, 0, true>(p, 0, s);
        }
        return fail(KPGNN_EINVAL, "aggregate_bwd: bf16 storage needs mode GINPLUS/SUM without table gradients and D %% 4 == 0");
    }
    if (p.gx && p.acc_mask) {      // one [N,K,D] output that already holds part of the gradient: the chunked gather only
        if (!gcn && tab == 0) return launch_bwd<VEC, G, false, 0, false, true>(p, 0, s);
        return fail(KPGNN_EINVAL, "aggregate_bwd: accumulate_mask with gx needs a non-GCN mode without table gradients");
    }
    switch (tab) {
        case 0: return gcn ? launch_bwd<VEC, G, true, 0>(p, 0, s) : launch_bwd<VEC, G, false, 0>(p, 0, s);
        case 1: return gcn ? launch_bwd<VEC, G, true, 1>(p, lds, s) : launch_bwd<VEC, G, false, 1>(p, lds, s);
        default: return gcn ? launch_bwd<VEC, G, true, 2>(p, 0, s) : launch_bwd<VEC, G, false, 2>(p, 0, s);
    }
}

// (VEC, G) pairs instantiated: G in {4,8,16,32,64}, G*VEC >= D  (D <= 256 for VEC 4).
#define KP_DISPATCH_SHAPE(CALL)                                                                    \
    switch (vec * 100 + g) {                                                                       \
        case 404: return CALL(4, 4); case 408: return CALL(4, 8); case 416: return CALL(4, 16);    \
        case 432: return CALL(4, 32); case 464: return CALL(4, 64);                                \
        case 204: return CALL(2, 4); case 208: return CALL(2, 8); case 216: return CALL(2, 16);    \
        case 232: return CALL(2, 32); case 264: return CALL(2, 64);                                \
        case 104: return CALL(1, 4); case 108: return CALL(1, 8); case 116: return CALL(1, 16);    \
        case 132: return CALL(1, 32); case 164: return CALL(1, 64);                                \
    }

}  // namespace
}  // namespace kpgnn

using namespace kpgnn;

extern "C" int kpgnn_aggregate_fwd(const kpgnn_agg_fwd_desc* d, kpgnn_stream_t stream) {
    KPGNN_REQUIRE(d != nullptr, "aggregate_fwd: NULL descriptor");
    KPGNN_REQUIRE(d->N >= 0 && d->K >= 1 && d->D >= 1 && d->K_csr >= d->K, "aggregate_fwd: bad N=%d K=%d D=%d K_csr=%d",
                  d->N, d->K, d->D, d->K_csr);
    if ((int64_t)d->N * d->K_csr >= ((int64_t)1 << 31)) return fail(KPGNN_ELIMIT, "aggregate_fwd: N*K exceeds int32");
    if (d->N == 0) return KPGNN_OK;
    KPGNN_REQUIRE(d->rowptr != nullptr, "aggregate_fwd: NULL rowptr");
    if (!d->x) {
        KPGNN_REQUIRE(d->K <= 16, "aggregate_fwd: per-hop inputs need K <= 16");
        for (int k = 0; k < d->K; ++k) KPGNN_REQUIRE(d->x_slot[k] != nullptr, "aggregate_fwd: NULL x and NULL x_slot[%d]", k);
    }
    KPGNN_REQUIRE(d->mode >= KPGNN_MODE_GIN && d->mode <= KPGNN_MODE_SUM, "aggregate_fwd: unknown mode %d", d->mode);
    KPGNN_REQUIRE(d->mode != KPGNN_MODE_GCN || d->dis, "aggregate_fwd: GCN mode needs dis");
    const bool combine = d->theta != nullptr;
    KPGNN_REQUIRE(combine ? d->hout != nullptr : d->out != nullptr, "aggregate_fwd: NULL output");
    KPGNN_REQUIRE(!d->alphas || combine, "aggregate_fwd: alphas needs the theta buffer it fills");
    int tab = 0;
    size_t lds = 0;
    if (d->use_tables)
        KPGNN_REQUIRE(d->table0 && d->n_code0 >= 1 && (d->K == 1 || (d->tablek && d->n_codek >= 1)),
                      "aggregate_fwd: missing embedding tables");
    KPGNN_REQUIRE(d->periph || !d->uid || (d->ptab && d->uid_stride >= d->K), "aggregate_fwd: dictionary P needs ptab and uid_stride >= K");
    if (d->use_tables) {
        KPGNN_REQUIRE(d->mode != KPGNN_MODE_GCN || (d->n_code0 >= 2 && (d->K == 1 || d->n_codek >= 2)),
                      "aggregate_fwd: GCN needs code row 1 (self loop) in both tables");
        lds = sizeof(float) * (size_t)d->D * ((size_t)d->n_code0 + (size_t)(d->K > 1 ? d->n_codek : 0));
        tab = lds <= (size_t)kMaxLdsTableBytes ? 1 : 2;
    }
    {   // dense neighbourhoods with the graph boundaries known: the graph's hop slab staged in LDS
        bool handled = false;
        const int rc = agg_lds_fwd(d, (hipStream_t)stream, &handled);
        if (rc != KPGNN_OK || handled) return rc;
    }
    KPGNN_REQUIRE((!d->hinit && !d->hinit2) || (combine && !d->alphas), "aggregate_fwd: hinit / hinit2 need a fused combine with a given theta");
    if (!d->hinit && !d->hinit2) {   // narrow rows (KP-GIN's hidden / K): one thread per output element, all hops of a node in parallel
        bool handled = false;
        int rc = d->n_dyn ? KPGNN_OK : agg_narrow_fwd(d, (hipStream_t)stream, &handled);
        if (rc != KPGNN_OK || handled) return rc;
        if (!d->n_dyn) rc = agg_small_fwd(d, (hipStream_t)stream, &handled);      // small batches: all hops of a node at once
        if (rc != KPGNN_OK || handled) return rc;
    }
    FwdParams p;
    p.lds_theta = p.lds_ptab = 0;
    if (tab == 1) {      // small side tables behind the code tables (keeps >= 4 blocks of 256 threads per CU)
        lds = (lds + 15) & ~(size_t)15;
        const size_t cap = 36 * 1024;
        const size_t th_b = combine ? sizeof(float) * (size_t)d->K * d->D : 0;
        if (th_b && lds + th_b <= cap) { p.lds_theta = d->K * d->D; lds += (th_b + 15) & ~(size_t)15; }
        const size_t pt_b = (!d->periph && d->ptab && d->n_dict > 0) ? sizeof(float) * (size_t)d->n_dict * d->D : 0;
        if (pt_b && lds + pt_b <= cap) { p.lds_ptab = d->n_dict * d->D; lds += (pt_b + 15) & ~(size_t)15; }
    }
    p.N = d->N; p.n_dyn = d->n_dyn; p.K = d->K; p.D = d->D; p.K_csr = d->K_csr; p.n_code0 = d->n_code0; p.n_codek = d->K > 1 ? d->n_codek : 0;
    p.mode = d->mode; p.combine = combine ? 1 : 0; p.bf = d->storage == KPGNN_STORE_BF16 ? 1 : 0;
    KPGNN_REQUIRE(d->storage == KPGNN_STORE_F32 || d->storage == KPGNN_STORE_BF16, "aggregate_fwd: unknown storage %d", d->storage);
    p.rowptr = d->rowptr; p.col = d->col; p.code = d->code; p.dis = d->dis;
    p.x = d->x; p.x_sn = d->x_sn; p.x_sk = d->x_sk;
    p.table0 = d->table0; p.tablek = d->tablek;
    p.periph = d->periph; p.p_sn = d->p_sn; p.p_sk = d->p_sk;
    p.eps = d->eps; p.out = d->out; p.o_sn = d->o_sn; p.o_sk = d->o_sk; p.pre = d->pre; p.theta = d->theta; p.hout = d->hout; p.hinit = d->hinit; p.hinit2 = d->hinit2; p.xbias = d->xbias;
    p.alphas = nullptr; p.theta_out = nullptr;
    if (d->alphas) {
        if (p.lds_theta) { p.alphas = d->alphas; p.theta_out = const_cast<float*>(d->theta); }   // theta staged in LDS: computed there
        else {                                                                                    // otherwise by a launch of its own
            const int rc = geo_theta_fwd_launch(d->alphas, d->K, d->D, const_cast<float*>(d->theta), (hipStream_t)stream);
            if (rc != KPGNN_OK) return rc;
        }
    }
    p.ptab = d->periph ? nullptr : d->ptab; p.uid = d->periph ? nullptr : d->uid; p.uid_stride = d->uid_stride;
    KPGNN_REQUIRE(p.uid == nullptr || (p.ptab != nullptr && p.uid_stride >= d->K), "aggregate_fwd: dictionary P needs ptab and uid_stride >= K");
    uintptr_t slot_bits = 0;            // low address bits of ALL per-hop inputs OR-ed: the least aligned one decides the vector width
    for (int k = 0; k < 16; ++k) {
        p.xs[k] = (!d->x && k < d->K) ? d->x_slot[k] : nullptr;
        slot_bits |= (uintptr_t)p.xs[k] & 15;
    }
    const void* slot_align = (const void*)(slot_bits | 16);   // synthetic address carrying that alignment (never dereferenced)
    if ((uint64_t)d->N * (uint64_t)d->x_sn * 4u >= (1ull << 32))
        return fail(KPGNN_ELIMIT, "aggregate_fwd: N * x row stride = %lld floats exceeds the 32-bit byte offsets of the gather", (long long)d->N * d->x_sn);
    const int vec = pick_vec(d->D, {d->x ? (const void*)d->x : slot_align, d->periph, d->out, d->pre, d->table0, d->tablek, d->theta, d->hout, d->hinit, d->hinit2, d->xbias, d->periph ? nullptr : d->ptab},
                             {d->x_sn, d->x ? d->x_sk : 0, d->periph ? d->p_sn : 0, d->periph ? d->p_sk : 0,
                              d->out ? d->o_sn : 0, d->out ? d->o_sk : 0});
    const int lanes = (d->D + vec - 1) / vec;
    if (lanes > 64) return fail(KPGNN_ELIMIT, "aggregate_fwd: D=%d with %d-wide access needs %d lanes > 64", d->D, vec, lanes);
    const int g = pick_group(lanes);
    hipStream_t s = (hipStream_t)stream;
#define KP_CALL(VEC_, G_) launch_fwd_mode<VEC_, G_>(p, tab, lds, s)
    KP_DISPATCH_SHAPE(KP_CALL)
#undef KP_CALL
    return fail(KPGNN_EINVAL, "aggregate_fwd: no kernel for vec=%d g=%d", vec, g);
}

extern "C" int kpgnn_aggregate_bwd(const kpgnn_agg_bwd_desc* d, kpgnn_stream_t stream) {
    KPGNN_REQUIRE(d != nullptr, "aggregate_bwd: NULL descriptor");
    KPGNN_REQUIRE(d->N >= 0 && d->K >= 1 && d->D >= 1 && d->K_csr >= d->K, "aggregate_bwd: bad N=%d K=%d D=%d K_csr=%d",
                  d->N, d->K, d->D, d->K_csr);
    if ((int64_t)d->N * d->K_csr >= ((int64_t)1 << 31)) return fail(KPGNN_ELIMIT, "aggregate_bwd: N*K exceeds int32");
    if (d->N == 0) return KPGNN_OK;
    KPGNN_REQUIRE(d->rowptr_src && d->g, "aggregate_bwd: NULL rowptr/g");
    if (!d->gx) {
        KPGNN_REQUIRE(d->K <= 16, "aggregate_bwd: per-hop outputs need K <= 16");
        for (int k = 0; k < d->K; ++k) KPGNN_REQUIRE(d->gx_slot[k] != nullptr, "aggregate_bwd: NULL gx and NULL gx_slot[%d]", k);
    }
    KPGNN_REQUIRE(d->mode >= KPGNN_MODE_GIN && d->mode <= KPGNN_MODE_SUM, "aggregate_bwd: unknown mode %d", d->mode);
    KPGNN_REQUIRE(d->mode != KPGNN_MODE_GCN || d->dis, "aggregate_bwd: GCN mode needs dis");
    int tab = 0;
    size_t lds = 0;
    if (d->use_tables && d->gtable0) {
        KPGNN_REQUIRE(d->n_code0 >= 1 && (d->K == 1 || (d->gtablek && d->n_codek >= 1)), "aggregate_bwd: missing table grads");
        lds = sizeof(float) * (size_t)d->D * ((size_t)d->n_code0 + (size_t)(d->K > 1 ? d->n_codek : 0));
        tab = lds <= (size_t)kMaxLdsTableBytes ? 1 : 2;
    }
    {   // narrow rows: the element-per-thread gather (aggregate_narrow.hip)
        bool handled = false;
        const int rc = d->n_dyn ? KPGNN_OK : agg_narrow_bwd(d, (hipStream_t)stream, &handled);
        if (rc != KPGNN_OK || handled) return rc;
    }
    BwdParams p;
    p.N = d->N; p.n_dyn = d->n_dyn; p.K = d->K; p.D = d->D; p.K_csr = d->K_csr; p.n_code0 = d->n_code0; p.n_codek = d->K > 1 ? d->n_codek : 0;
    p.mode = d->mode; p.bf = d->storage == KPGNN_STORE_BF16 ? 1 : 0;
    KPGNN_REQUIRE(d->storage == KPGNN_STORE_F32 || d->storage == KPGNN_STORE_BF16, "aggregate_bwd: unknown storage %d", d->storage);
    p.rowptr = d->rowptr_src; p.col = d->col_src; p.code = d->code_src; p.dis = d->dis;
    p.g = d->g; p.g_sn = d->g_sn; p.g_sk = d->g_sk; p.eps = d->eps;
    p.gx = d->gx; p.gx_sn = d->gx_sn; p.gx_sk = d->gx_sk; p.gtable0 = d->gtable0; p.gtablek = d->gtablek;
    uintptr_t slot_bits = 0;            // (as in the forward: OR of the low address bits of all slot outputs)
    for (int k = 0; k < 16; ++k) {
        p.gxs[k] = (!d->gx && k < d->K) ? d->gx_slot[k] : nullptr;
        slot_bits |= (uintptr_t)p.gxs[k] & 15;
    }
    const void* slot_align = (const void*)(slot_bits | 16);
    p.acc_mask = d->accumulate_mask;      // (with gx: bit k adds hop k's gradient to what gx[:, k, :] already holds)
    KPGNN_REQUIRE(d->accumulate_mask == 0 || d->K <= 32, "aggregate_bwd: accumulate_mask covers 32 hops");
    // (the kernel requests the old value of an accumulating slot one hop ahead and stores a hop's result one hop late:
    //  two hops of one launch must not share a slot buffer)
    for (int a = 0; a < d->K && a < 16; ++a)
        for (int b = a + 1; b < d->K && b < 16; ++b)
            if (p.gxs[a] && p.gxs[a] == p.gxs[b] && (((p.acc_mask >> a) | (p.acc_mask >> b)) & 1u))
                return fail(KPGNN_EINVAL, "aggregate_bwd: hop slots %d and %d share one gradient buffer with accumulate_mask set", a, b);
    {   // small batches: one block per node, all hops at once (aggregate_small.hip)
        bool handled = false;
        const int rc = d->n_dyn ? KPGNN_OK : agg_small_bwd(d, (hipStream_t)stream, &handled);
        if (rc != KPGNN_OK || handled) return rc;
    }
    if ((uint64_t)d->N * (uint64_t)d->g_sn * 4u >= (1ull << 32))
        return fail(KPGNN_ELIMIT, "aggregate_bwd: N * g row stride = %lld floats exceeds the 32-bit byte offsets of the gather", (long long)d->N * d->g_sn);
    const int vec = pick_vec(d->D, {d->g, d->gx ? (const void*)d->gx : slot_align}, {d->g_sn, d->g_sk, d->gx_sn, d->gx ? d->gx_sk : 0});
    const int lanes = (d->D + vec - 1) / vec;
    if (lanes > 64) return fail(KPGNN_ELIMIT, "aggregate_bwd: D=%d with %d-wide access needs %d lanes > 64", d->D, vec, lanes);
    const int g = pick_group(lanes);
    hipStream_t s = (hipStream_t)stream;
#define KP_CALL(VEC_, G_) launch_bwd_mode<VEC_, G_>(p, tab, lds, s)
    KP_DISPATCH_SHAPE(KP_CALL)
#undef KP_CALL
    return fail(KPGNN_EINVAL, "aggregate_bwd: no kernel for vec=%d g=%d", vec, g);
}
