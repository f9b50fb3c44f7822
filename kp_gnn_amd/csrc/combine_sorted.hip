// Backward pre-pass of the fused aggregation epilogue WITH the table gradients (gfx950).
// Contract: include/kpgnn.h, kpgnn_combine_sorted.
//
// Same arithmetic as combine.hip (g = dL/dS from the saved S, theta gradient), but the (node, hop) rows are visited in
// the (hop, code) order of csr_segments.hip instead of memory order.  A sub-group of lanes then meets runs of <= 32 rows
// that all add into the SAME table row: sum_m mult * g[row] stays in VEC registers and leaves as one slab row per
// segment; the theta-gradient partial of the segment likewise.  One pass over S (and gh, L2-resident) writes g AND
// yields the edge-code table gradients - the separate table_grad pass over g (round 1: 0.70 ms of a 5.8 ms step, the
// largest kernel) disappears for the layers that need this pre-pass (KP-GIN+, KP-GCN... every epilogue with an
// activation or a fused combine).  Rows holding several distinct codes are visited once per code (16 % more rows on
// ZINC-shaped batches; only the `first` visit stores g and counts for theta).
// The peripheral-dictionary gradient (sum of theta[k]*gh[i] grouped by uid[i,k]) has a different grouping key and is a
// small kernel of its own over gh (dict_grad_kernel).  A finishing kernel adds the slab rows of every output row in a
// fixed order (bitwise reproducible) and, for a geometric combine, turns the theta gradient into the alpha gradient.
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

struct CsParams {
    int N, K, D, mode;
    const float* pre; const float* gh; const float* theta;
    const float* gout; int64_t go_sn, go_sk;
    const float* periph; int64_t p_sn, p_sk;
    const float* ptab; const int32_t* uid; int64_t uid_stride;
    float* g; float* gv;
    const uint32_t* ent; const int32_t* seg_ptr; const uint32_t* seg_key; int nseg;
    float* slab_tab;     // [nseg][D] or NULL (no table gradients wanted)
    float* slab_th;      // [nseg][D] or NULL
    int lds_theta, lds_ptab;
};

// ACT: 1 = GELU (KP-GIN+), 2 = ReLU (KP-GCN), 0 = none; WGT: theta gradient wanted.
template <int VEC, int G, int ACT, bool WGT>
__global__ void __launch_bounds__(kBlock)
combine_sorted_kernel(const CsParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [lds_theta] theta rows, [lds_ptab] dictionary rows
    constexpr int NODES = kBlock / G;
    constexpr int UN = WGT ? 2 : 4;
    const int sg = threadIdx.x / G, sl = threadIdx.x % G;
    const int lane = threadIdx.x & (kWave - 1);
    const int sg_lane0 = lane - sl;
    const int c0 = sl * VEC;
    const int D = p.D, K = p.K;
    const bool col_ok = c0 < D;
    const bool fused = p.theta != nullptr;
    float* th_l = lds;
    float* pt_l = lds + ((p.lds_theta + 3) & ~3);
    for (int t = threadIdx.x; t < p.lds_theta; t += kBlock) th_l[t] = p.theta[t];
    for (int t = threadIdx.x; t < p.lds_ptab; t += kBlock) pt_l[t] = p.ptab[t];
    if (p.lds_theta || p.lds_ptab) __syncthreads();
    const float* __restrict__ thp = p.lds_theta ? th_l : p.theta;
    const float* __restrict__ ptp = p.lds_ptab ? pt_l : p.ptab;
    const int cc = col_ok ? c0 : 0;                   // idle lanes re-read column 0 and are dropped at the stores

    const int64_t total_sg = (int64_t)gridDim.x * NODES;
    for (int64_t seg = (int64_t)blockIdx.x * NODES + sg; seg < p.nseg; seg += total_sg) {
        const int beg = p.seg_ptr[seg], end = p.seg_ptr[seg + 1];
        const uint32_t key = p.seg_key[seg];
        const int k = (int)(key >> 16);
        float thv[VEC], acc[VEC], gt[VEC];
        for (int q = 0; q < VEC; ++q) { thv[q] = 0.f; acc[q] = 0.f; gt[q] = 0.f; }
        if (fused) ldv<VEC>(thp + k * D + cc, thv);
        for (int base = beg; base < end; base += G) {
            // lane sl holds entry base + sl of the segment (one coalesced load), broadcast below
            uint32_t my_node = 0, my_mw = 0;
            if (base + sl < end) { my_node = p.ent[2 * (int64_t)(base + sl)]; my_mw = p.ent[2 * (int64_t)(base + sl) + 1]; }
            const int cnt = min(G, end - base);
            for (int t0 = 0; t0 < cnt; t0 += UN) {
                float s[UN][VEC], gvv[UN][VEC], ghv[UN][VEC], pv[UN][VEC];
                int64_t node[UN]; uint32_t mw[UN]; int u_id[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int l = sg_lane0 + min(t0 + u, cnt - 1);
                    node[u] = (int64_t)__shfl(my_node, l);
                    mw[u] = t0 + u < cnt ? __shfl(my_mw, l) : 0u;       // (padding slots: multiplicity 0, not first)
                    if (t0 + u >= cnt) mw[u] = 0u;
                    u_id[u] = -1;
                    for (int q = 0; q < VEC; ++q) pv[u][q] = 0.f;
                    ldv<VEC>(p.pre + (node[u] * K + k) * D + cc, s[u]);
                    if (fused) ldv<VEC>(p.gh + node[u] * D + cc, ghv[u]);
                    else ldv<VEC>(p.gout + node[u] * p.go_sn + (int64_t)k * p.go_sk + cc, gvv[u]);
                    if (WGT && (mw[u] >> 31)) {
                        if (p.periph) ldv<VEC>(p.periph + node[u] * p.p_sn + (int64_t)k * p.p_sk + cc, pv[u]);
                        else if (p.uid) u_id[u] = p.uid[node[u] * p.uid_stride + k];
                    }
                }
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    if (t0 + u >= cnt) break;
                    const bool first = (mw[u] >> 31) != 0;
                    const float mult = (float)(mw[u] & 0x7FFFFFFFu);
                    float gg[VEC], a[VEC];
                    if (fused) for (int q = 0; q < VEC; ++q) gvv[u][q] = thv[q] * ghv[u][q];
                    for (int q = 0; q < VEC; ++q) {
                        const float sv = s[u][q];
                        if (ACT == 1) {
                            float e2;  // exp(-s^2/2) comes with the erf approximation
                            const float cdf = 0.5f * (1.0f + fast_erf(sv * 0.70710678118654752440f, &e2));
                            const float pdf = e2 * 0.39894228040143267794f;
                            a[q] = sv * cdf;
                            gg[q] = gvv[u][q] * (cdf + sv * pdf);
                        } else if (ACT == 2) {
                            a[q] = fmaxf(sv, 0.f);
                            gg[q] = sv > 0.f ? gvv[u][q] : 0.f;
                        } else {
                            a[q] = sv;
                            gg[q] = gvv[u][q];
                        }
                        acc[q] = fmaf(mult, gg[q], acc[q]);
                    }
                    if (first && col_ok) {
                        stv<VEC>(p.g + (node[u] * K + k) * D + c0, gg);
                        if (p.gv) stv<VEC>(p.gv + (node[u] * K + k) * D + c0, gvv[u]);
                    }
                    if (WGT && first) {
                        if (u_id[u] >= 0) ldv<VEC>(ptp + (int64_t)u_id[u] * D + cc, pv[u]);
                        for (int q = 0; q < VEC; ++q) gt[q] = fmaf(ghv[u][q], a[q] + pv[u][q], gt[q]);
                    }
                }
            }
        }
        if (col_ok) {
            if (p.slab_tab) stv<VEC>(p.slab_tab + seg * D + c0, acc);
            if (WGT) stv<VEC>(p.slab_th + seg * D + c0, gt);
        }
    }
}

// ---------------------------------------------------------------------------------------------- dictionary gradient
// slab[b][u][:] = sum over the nodes of block b and hops k with uid[i,k] == u of theta[k,:] * gh[i,:].
// Block = NG groups of CW threads (thread = column); a group walks ITS contiguous run of nodes in order and adds into a
// private [U][CW] table in LDS (plain read-modify-write: no races, fixed order); groups are then added in order.
struct DgParams { int N, K, D, U; const float* gh; const float* theta; const int32_t* uid; int64_t uid_stride; float* slab; };

__global__ void __launch_bounds__(kBlock)
dict_grad_kernel(const DgParams p, int CW, int NG) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [NG][U][CW]
    for (int t = threadIdx.x; t < NG * p.U * CW; t += kBlock) lds[t] = 0.f;
    __syncthreads();
    const int grp = threadIdx.x / CW, col = threadIdx.x % CW;
    const int64_t per_block = ((int64_t)p.N + gridDim.x - 1) / gridDim.x;
    const int64_t b0 = (int64_t)blockIdx.x * per_block;
    const int64_t b1 = b0 + per_block < p.N ? b0 + per_block : p.N;
    if (grp < NG && col < p.D && b0 < b1) {
        const int64_t per_grp = (b1 - b0 + NG - 1) / NG;
        const int64_t m0 = b0 + grp * per_grp;
        const int64_t m1 = m0 + per_grp < b1 ? m0 + per_grp : b1;
        float th[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) th[k] = k < p.K ? p.theta[k * p.D + col] : 0.f;
        float* acc = lds + (int64_t)grp * p.U * CW + col;
        for (int64_t m = m0; m < m1; m += 2) {       // two nodes per trip: their loads do not wait for the LDS updates
            const float g0 = p.gh[m * p.D + col];
            const float g1 = m + 1 < m1 ? p.gh[(m + 1) * p.D + col] : 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (k >= p.K) break;
                const int u0 = p.uid[m * p.uid_stride + k];                       // wave-uniform when CW >= 64
                const int u1 = m + 1 < m1 ? p.uid[(m + 1) * p.uid_stride + k] : -1;
                acc[u0 * CW] = fmaf(th[k], g0, acc[u0 * CW]);
                if (u1 >= 0) acc[u1 * CW] = fmaf(th[k], g1, acc[u1 * CW]);
            }
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < p.U * p.D; t += kBlock) {
        const int r = t / p.D, c = t - r * p.D;
        float v = 0.f;
        for (int g = 0; g < NG; ++g) v += lds[((int64_t)g * p.U + r) * CW + c];
        p.slab[((int64_t)blockIdx.x * p.U + r) * p.D + c] = v;
    }
}

// ---------------------------------------------------------------------------------------------- finish
// One block per (output row, 16 columns): 64 slices add every 64th slab row of the row's segment ranges, the 64 partials
// meet in LDS in slice order.  Output rows: n0 rows of gtable0 (hop 0, code r), nk rows of gtablek (code r, hops 1..K-1),
// one row block for the theta gradient of ALL K hops (then galpha, if alpha is given), U rows of gdict (dict slab).
struct FinParams {
    int K, D, n0, nk, U, nseg, ndict_blocks, want_tab, want_th;
    const float* slab_tab; const float* slab_th; const float* slab_dict;
    const uint32_t* seg_key; const int32_t* hop_seg;
    float* gtable0; float* gtablek; float* gtheta; float* gdict;
    const float* alpha; const float* theta; float* galpha;
};

__device__ __forceinline__ int seg_lower_bound(const uint32_t* keys, int n, uint32_t key) {
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}

__global__ void __launch_bounds__(1024)
combine_finish_kernel(const FinParams p) {
    __shared__ float part[64][17];
    __shared__ int range[2];
    const int o = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int col = blockIdx.y * 16 + o;
    const bool col_ok = col < p.D;
    const int r = blockIdx.x;
    auto block_sum = [&](const float* slab, int a, int b) -> float {      // sum of slab rows [a, b), valid in slice 0
        float s = 0.f;
        if (col_ok) for (int i = a + slice; i < b; i += 64) s += slab[(int64_t)i * p.D + col];
        __syncthreads();
        part[slice][o] = s;
        __syncthreads();
        float tot = 0.f;
        if (slice == 0) for (int q = 0; q < 64; ++q) tot += part[q][o];
        return tot;
    };
    auto key_range = [&](uint32_t key) {                                   // segments with exactly this (hop, code)
        __syncthreads();
        if (threadIdx.x == 0) { range[0] = seg_lower_bound(p.seg_key, p.nseg, key); range[1] = seg_lower_bound(p.seg_key, p.nseg, key + 1); }
        __syncthreads();
    };
    if (r < p.n0) {
        if (!p.want_tab) return;
        key_range((uint32_t)r);
        const float tot = block_sum(p.slab_tab, range[0], range[1]);
        if (slice == 0 && col_ok) p.gtable0[(int64_t)r * p.D + col] = tot;
    } else if (r < p.n0 + p.nk) {
        if (!p.want_tab) return;
        const int code = r - p.n0;
        float tot = 0.f;
        for (int h = 1; h < p.K; ++h) {
            key_range(((uint32_t)h << 16) | (uint32_t)code);
            tot += block_sum(p.slab_tab, range[0], range[1]);
        }
        if (slice == 0 && col_ok) p.gtablek[(int64_t)code * p.D + col] = tot;
    } else if (r == p.n0 + p.nk) {
        if (!p.want_th) return;
        float gth[16];
        for (int k = 0; k < p.K; ++k) {
            const int a = p.hop_seg[k], b = min(p.hop_seg[k + 1], p.nseg);
            gth[k] = block_sum(p.slab_th, a, b);
            if (slice == 0 && col_ok && p.gtheta) p.gtheta[(int64_t)k * p.D + col] = gth[k];
        }
        if (p.galpha && slice == 0 && col_ok) {
            // dt[k] = theta[k] (G[k] - sum_j theta[j] G[j]);  dt[k]/da = q^k - k a q^(k-1);  da/dalpha = a q  (geo_theta.hip)
            const float a = 1.0f / (1.0f + __expf(-p.alpha[col]));
            const float q = 1.0f - a;
            float dot = 0.f;
            for (int k = 0; k < p.K; ++k) dot = fmaf(p.theta[(int64_t)k * p.D + col], gth[k], dot);
            float acc = 0.f, pw = 1.0f, pwm1 = 0.f;
            for (int k = 0; k < p.K; ++k) {
                const float dt = p.theta[(int64_t)k * p.D + col] * (gth[k] - dot);
                acc = fmaf(dt, pw - (float)k * a * pwm1, acc);
                pwm1 = pw;
                pw *= q;
            }
            p.galpha[col] = a * q * acc;
        }
    } else {
        const int u = r - (p.n0 + p.nk + 1);
        if (u >= p.U) return;
        float s = 0.f;
        if (col_ok) for (int b = slice; b < p.ndict_blocks; b += 64) s += p.slab_dict[((int64_t)b * p.U + u) * p.D + col];
        part[slice][o] = s;
        __syncthreads();
        if (slice == 0 && col_ok) {
            float tot = 0.f;
            for (int q = 0; q < 64; ++q) tot += part[q][o];
            p.gdict[(int64_t)u * p.D + col] = tot;
        }
    }
}

int cs_shape(const kpgnn_combine_sorted_desc* d, int* vec, int* g) {
    int v = (d->D % 4 == 0) ? 4 : (d->D % 2 == 0 ? 2 : 1);
    auto al = [&](const void* q) { while (v > 1 && q && ((uintptr_t)q % (v * 4))) v >>= 1; };
    al(d->pre); al(d->gh); al(d->theta); al(d->gout); al(d->periph); al(d->ptab); al(d->g); al(d->gv); al(d->workspace);
    for (int64_t s : {d->gout ? d->go_sn : 0, d->gout ? d->go_sk : 0, d->periph ? d->p_sn : 0, d->periph ? d->p_sk : 0})
        while (v > 1 && (s % v)) v >>= 1;
    const int lanes = (d->D + v - 1) / v;
    if (lanes > 64) return fail(KPGNN_ELIMIT, "combine_sorted: D=%d needs %d lanes > 64", d->D, lanes);
    int gg = 4;
    while (gg < lanes) gg <<= 1;
    *vec = v; *g = gg;
    return KPGNN_OK;
}

template <int VEC, int G, int ACT, bool WGT>
int cs_launch2(const CsParams& p, size_t lds, hipStream_t s) {
    if (lds > 64 * 1024) KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)combine_sorted_kernel<VEC, G, ACT, WGT>, lds));
    int nb = resident_blocks(combine_sorted_kernel<VEC, G, ACT, WGT>, kBlock, lds);   // one resident round
    if (nb < 1) nb = 4;
    int64_t grid = (int64_t)device_facts().cu_count * nb;
    const int64_t need = ((int64_t)p.nseg + (kBlock / G) - 1) / (kBlock / G);
    if (grid > need) grid = need;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL((combine_sorted_kernel<VEC, G, ACT, WGT>), dim3((unsigned)grid), dim3(kBlock), lds, s, p);
    KPGNN_LAUNCH_CHECK("combine_sorted_kernel");
    return KPGNN_OK;
}

template <int VEC, int G>
int cs_launch(const CsParams& p, hipStream_t s) {
    const size_t lds = sizeof(float) * (size_t)(((p.lds_theta + 3) & ~3) + ((p.lds_ptab + 3) & ~3));
    const int act = p.mode == KPGNN_MODE_GINPLUS ? 1 : (p.mode == KPGNN_MODE_GCN ? 2 : 0);
    if (p.slab_th) {
        if (act == 1) return cs_launch2<VEC, G, 1, true>(p, lds, s);
        if (act == 2) return cs_launch2<VEC, G, 2, true>(p, lds, s);
        return cs_launch2<VEC, G, 0, true>(p, lds, s);
    }
    if (act == 1) return cs_launch2<VEC, G, 1, false>(p, lds, s);
    if (act == 2) return cs_launch2<VEC, G, 2, false>(p, lds, s);
    return cs_launch2<VEC, G, 0, false>(p, lds, s);
}

struct DictPlan { int CW, NG, blocks; size_t lds; };

bool dict_plan(int N, int D, int U, DictPlan* pl) {
    if (D > kBlock || U < 1) return false;
    int cw = 1;
    while (cw < D) cw <<= 1;
    const size_t one = sizeof(float) * (size_t)U * cw;
    if (one > 96 * 1024) return false;
    int ng = kBlock / cw;
    while (ng > 1 && one * ng > 96 * 1024) ng >>= 1;
    int64_t blocks = ((int64_t)N + 32 * ng - 1) / (32 * ng);      // >= 32 nodes per group
    if (blocks > device_facts().cu_count) blocks = device_facts().cu_count;
    if (blocks < 1) blocks = 1;
    pl->CW = cw; pl->NG = ng; pl->blocks = (int)blocks; pl->lds = one * ng;
    return true;
}

}  // namespace
}  // namespace kpgnn

using namespace kpgnn;

extern "C" size_t kpgnn_combine_sorted_workspace_bytes(int32_t num_segments, int32_t N, int32_t D, int32_t n_dict) {
    if (num_segments < 0 || D < 1) return 0;
    size_t b = sizeof(float) * 2 * (size_t)(num_segments > 0 ? num_segments : 1) * D;
    DictPlan pl;
    if (n_dict > 0 && dict_plan(N, D, n_dict, &pl)) b += sizeof(float) * (size_t)pl.blocks * n_dict * D;
    return b + 256;
}

extern "C" int kpgnn_combine_sorted(const kpgnn_combine_sorted_desc* d, kpgnn_stream_t stream) {
    KPGNN_REQUIRE(d != nullptr, "combine_sorted: NULL descriptor");
    KPGNN_REQUIRE(d->N >= 0 && d->K >= 1 && d->K <= 16 && d->D >= 1, "combine_sorted: bad N=%d K=%d D=%d (K <= 16)", d->N, d->K, d->D);
    if (d->N == 0) return KPGNN_OK;
    KPGNN_REQUIRE(d->pre && d->g, "combine_sorted: NULL pre/g");
    KPGNN_REQUIRE(d->theta ? d->gh != nullptr : d->gout != nullptr, "combine_sorted: need (theta, gh) or gout");
    KPGNN_REQUIRE(d->mode >= KPGNN_MODE_GIN && d->mode <= KPGNN_MODE_SUM, "combine_sorted: unknown mode %d", d->mode);
    KPGNN_REQUIRE(d->entries && d->seg_ptr && d->seg_key && d->hop_seg && d->num_segments >= 1, "combine_sorted: missing segment list");
    const bool want_tab = d->gtable0 != nullptr;
    const bool want_th = (d->gtheta != nullptr || d->galpha != nullptr) && d->theta != nullptr;
    KPGNN_REQUIRE(!want_tab || (d->n_code0 >= 1 && (d->K == 1 || (d->gtablek && d->n_codek >= 1))), "combine_sorted: missing gtable0/gtablek");
    KPGNN_REQUIRE(!d->galpha || d->alpha, "combine_sorted: galpha needs alpha");
    const bool want_dict = d->gdict != nullptr && d->n_dict > 0;
    KPGNN_REQUIRE(!want_dict || (d->uid && d->theta && d->gh && d->uid_stride >= d->K), "combine_sorted: dictionary gradient needs uid, theta, gh");
    KPGNN_REQUIRE(d->workspace && d->workspace_bytes >= kpgnn_combine_sorted_workspace_bytes(d->num_segments, d->N, d->D, want_dict ? d->n_dict : 0),
                  "combine_sorted: workspace too small");
    int vec = 1, g = 4;
    int rc = cs_shape(d, &vec, &g);
    if (rc != KPGNN_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    float* ws = (float*)d->workspace;
    const size_t seg_rows = (size_t)d->num_segments * d->D;
    CsParams p;
    p.N = d->N; p.K = d->K; p.D = d->D; p.mode = d->mode; p.pre = d->pre; p.gh = d->gh; p.theta = d->theta;
    p.gout = d->gout; p.go_sn = d->go_sn; p.go_sk = d->go_sk; p.periph = d->periph; p.p_sn = d->p_sn; p.p_sk = d->p_sk;
    p.ptab = d->periph ? nullptr : d->ptab; p.uid = d->periph ? nullptr : d->uid; p.uid_stride = d->uid_stride;
    p.g = d->g; p.gv = d->gv;
    p.ent = d->entries; p.seg_ptr = d->seg_ptr; p.seg_key = d->seg_key; p.nseg = d->num_segments;
    p.slab_tab = want_tab ? ws : nullptr;
    p.slab_th = want_th ? ws + seg_rows : nullptr;
    p.lds_theta = d->theta ? d->K * d->D : 0;
    p.lds_ptab = (want_th && p.ptab && p.uid && d->n_dict > 0 && (size_t)d->n_dict * d->D * sizeof(float) <= 16 * 1024) ? d->n_dict * d->D : 0;
#define KP_CS(V, GG) rc = cs_launch<V, GG>(p, s); break
    switch (vec * 100 + g) {
        case 404: KP_CS(4, 4); case 408: KP_CS(4, 8); case 416: KP_CS(4, 16); case 432: KP_CS(4, 32); case 464: KP_CS(4, 64);
        case 204: KP_CS(2, 4); case 208: KP_CS(2, 8); case 216: KP_CS(2, 16); case 232: KP_CS(2, 32); case 264: KP_CS(2, 64);
        case 104: KP_CS(1, 4); case 108: KP_CS(1, 8); case 116: KP_CS(1, 16); case 132: KP_CS(1, 32); case 164: KP_CS(1, 64);
        default: return fail(KPGNN_EINVAL, "combine_sorted: no kernel for vec=%d g=%d", vec, g);
    }
#undef KP_CS
    if (rc != KPGNN_OK) return rc;
    DictPlan dpl = {};
    float* slab_dict = ws + 2 * seg_rows;
    if (want_dict) {
        if (!dict_plan(d->N, d->D, d->n_dict, &dpl))
            return fail(KPGNN_ELIMIT, "combine_sorted: dictionary of %d rows x %d columns does not fit the LDS kernel", d->n_dict, d->D);
        DgParams q;
        q.N = d->N; q.K = d->K; q.D = d->D; q.U = d->n_dict; q.gh = d->gh; q.theta = d->theta; q.uid = d->uid;
        q.uid_stride = d->uid_stride; q.slab = slab_dict;
        KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)dict_grad_kernel, dpl.lds));
        hipLaunchKernelGGL(dict_grad_kernel, dim3(dpl.blocks), dim3(kBlock), dpl.lds, s, q, dpl.CW, dpl.NG);
        KPGNN_LAUNCH_CHECK("dict_grad_kernel");
    }
    if (!want_tab && !want_th && !want_dict) return KPGNN_OK;
    FinParams f;
    f.K = d->K; f.D = d->D; f.n0 = want_tab ? d->n_code0 : 0; f.nk = (want_tab && d->K > 1) ? d->n_codek : 0;
    f.U = want_dict ? d->n_dict : 0; f.nseg = d->num_segments; f.ndict_blocks = dpl.blocks;
    f.want_tab = want_tab ? 1 : 0; f.want_th = want_th ? 1 : 0;
    f.slab_tab = p.slab_tab; f.slab_th = p.slab_th; f.slab_dict = slab_dict;
    f.seg_key = d->seg_key; f.hop_seg = d->hop_seg;
    f.gtable0 = d->gtable0; f.gtablek = d->gtablek; f.gtheta = d->gtheta; f.gdict = d->gdict;
    f.alpha = d->alpha; f.theta = d->theta; f.galpha = d->galpha;
    const unsigned rows = (unsigned)(f.n0 + f.nk + 1 + f.U);
    hipLaunchKernelGGL(combine_finish_kernel, dim3(rows, (unsigned)((d->D + 15) / 16)), dim3(1024), 0, s, f);
    KPGNN_LAUNCH_CHECK("combine_finish_kernel");
    return KPGNN_OK;
}
