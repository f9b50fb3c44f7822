// K-hop aggregation for NARROW rows (D <= 32: KP-GIN's per-hop width hidden / K = 13), forward and backward gather.
// Contract: include/kpgnn.h, kpgnn_aggregate_fwd / kpgnn_aggregate_bwd (this file is one of their kernels).
//
// The wide-row kernels (aggregate.hip) give a node to a sub-group of lanes that walks its hops one after the other: per
// hop one dependent round trip, ~10 per node, and at D = 13 a sub-group moves only 52 bytes per trip - 0.13 of the HBM
// roofline (profiles/r02/bench_kpgin.json).  Here a THREAD owns one output element (i, k, c) and walks the pairs of ITS
// segment (i, k): all K hops of a node advance in parallel across lanes, the dependency chain is three loads deep
// (row pointer -> pair -> row) whatever K is, consecutive lanes read consecutive floats of the same 52-byte row (one
// request) and the same pair-list entry (a broadcast).  Pairs are added in list order, as the reference's
// index_add_ over edges does (KPGIN.py:96-105) - results do not depend on the kernel choice.
// Measured (round 2): it wins where a launch is latency-bound - 3-regular n = 1280, 582 pairs per node, batch 1: 31 us
// against 239; QM9-shaped batch 128: 11 us against 33 - and loses where the sub-group kernels' prefetch across hops keeps
// more rows in flight per wave: ZINC batch 2048 (N = 47k, D = 13) 64 us against 47, regular batch 100 1.34 ms against 0.95.
// Four elements per thread did not help (84 us: registers, divergence).  Hence kNarrowMaxElems below.
#include "kpgnn_common.h"

namespace kpgnn {
namespace {

constexpr int kBlockN = 256;
constexpr int64_t kNarrowMaxElems = 1 << 21;   // N*K*D up to which this kernel is used (see the header)

struct NarrowParams {
    int N, K, D, K_csr, n_code0, n_codek, mode;
    int64_t total;                 // N * K * D
    const int32_t* rowptr;
    const int32_t* col;
    const uint16_t* code;
    const float* src; int64_t s_sn, s_sk;        // x (forward) or g (backward)
    const float* table0;
    const float* tablek;           // NULL: no tables
    const float* periph; int64_t p_sn, p_sk;
    const float* ptab; const int32_t* uid; int64_t uid_stride;
    const float* xbias;
    const float* eps;
    int self_term;                 // 1: add (1 + eps) * (src[i,k,c] + xbias)   (GIN forward / backward)
    float* out; int64_t o_sn, o_sk;
    float* pre;
    uint32_t acc_mask;             // bit k: out[:, k, :] += instead of =  (backward into a shared gradient buffer)
};

__device__ __forceinline__ float gelu_exact_n(float x) {
    float e2;
    return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752440f, &e2));
}

template <bool TAB>
__global__ void __launch_bounds__(kBlockN)
agg_narrow_kernel(const NarrowParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds_tab[];
    const int D = p.D, K = p.K;
    if (TAB) {
        const int n0 = p.n_code0 * D, nk = p.n_codek * D;
        for (int t = threadIdx.x; t < n0; t += kBlockN) lds_tab[t] = p.table0[t];
        for (int t = threadIdx.x; t < nk; t += kBlockN) lds_tab[n0 + t] = p.tablek[t];
        __syncthreads();
    }
    const float eps1 = 1.0f + (p.eps ? p.eps[0] : 0.0f);
    const int KD = K * D;
    for (int64_t e = (int64_t)blockIdx.x * kBlockN + threadIdx.x; e < p.total; e += (int64_t)gridDim.x * kBlockN) {
        const int i = (int)(e / KD);
        const int r = (int)(e - (int64_t)i * KD);
        const int k = r / D, c = r - k * D;
        const int32_t* rp = p.rowptr + (int64_t)i * p.K_csr + k;
        const int beg = rp[0], end = rp[1];
        const float* sk = p.src + (int64_t)k * p.s_sk + c;
        const float* tab = TAB ? (k == 0 ? lds_tab : lds_tab + p.n_code0 * D) + c : nullptr;
        float acc = 0.f;
        int q = beg;
        for (; q + 3 < end; q += 4) {               // four pairs in flight; added in list order
            const int j0 = p.col[q], j1 = p.col[q + 1], j2 = p.col[q + 2], j3 = p.col[q + 3];
            float v0 = sk[(int64_t)j0 * p.s_sn], v1 = sk[(int64_t)j1 * p.s_sn];
            float v2 = sk[(int64_t)j2 * p.s_sn], v3 = sk[(int64_t)j3 * p.s_sn];
            if (TAB) {
                v0 += tab[(int)p.code[q] * D]; v1 += tab[(int)p.code[q + 1] * D];
                v2 += tab[(int)p.code[q + 2] * D]; v3 += tab[(int)p.code[q + 3] * D];
            }
            acc += v0; acc += v1; acc += v2; acc += v3;
        }
        for (; q < end; ++q) {
            float v = sk[(int64_t)p.col[q] * p.s_sn];
            if (TAB) v += tab[(int)p.code[q] * D];
            acc += v;
        }
        // ---- epilogue for (i, k, c): the order of aggregate.hip's
        float v = acc;
        const float xb = (p.xbias && k >= 1) ? p.xbias[c] : 0.f;      // hopk_node_path_emb(pe_attr == 0), KPGIN.py:92-94
        if (p.xbias && k >= 1) v = fmaf((float)(end - beg), xb, v);
        if (p.pre) p.pre[e] = v;
        if (p.mode == KPGNN_MODE_GINPLUS) v = gelu_exact_n(v);
        if (p.periph) v += p.periph[(int64_t)i * p.p_sn + (int64_t)k * p.p_sk + c];
        else if (p.uid) v += p.ptab[(int64_t)p.uid[(int64_t)i * p.uid_stride + k] * D + c];
        if (p.self_term) v = fmaf(eps1, sk[(int64_t)i * p.s_sn] + xb, v);
        float* dst = p.out + (int64_t)i * p.o_sn + (int64_t)k * p.o_sk + c;
        if ((p.acc_mask >> k) & 1u) v += *dst;
        *dst = v;
    }
}

int agg_narrow_launch(const NarrowParams& p, hipStream_t s) {
    const bool tab = p.tablek != nullptr || p.table0 != nullptr;
    const size_t lds = tab ? sizeof(float) * (size_t)p.D * ((size_t)p.n_code0 + (size_t)p.n_codek) : 0;
    int64_t blocks = (p.total + kBlockN - 1) / kBlockN;
    const int64_t cap = (int64_t)device_facts().cu_count * 32;     // grid-stride beyond 8192 blocks (tables are staged per block)
    if (blocks > cap) blocks = cap;
    if (tab) hipLaunchKernelGGL(agg_narrow_kernel<true>, dim3((unsigned)blocks), dim3(kBlockN), lds, s, p);
    else hipLaunchKernelGGL(agg_narrow_kernel<false>, dim3((unsigned)blocks), dim3(kBlockN), 0, s, p);
    KPGNN_LAUNCH_CHECK("agg_narrow_kernel");
    return KPGNN_OK;
}

}  // namespace

// *handled = false: the shape is not this kernel's (the caller goes on to the sub-group kernels)
int agg_narrow_fwd(const kpgnn_agg_fwd_desc* d, hipStream_t s, bool* handled) {
    *handled = false;
    const bool tables = d->use_tables != 0;
    const size_t lds = tables ? sizeof(float) * (size_t)d->D * ((size_t)d->n_code0 + (size_t)(d->K > 1 ? d->n_codek : 0)) : 0;
    if (d->D > 32 || !d->x || d->theta || d->mode == KPGNN_MODE_GCN || lds > 48 * 1024 || d->storage != KPGNN_STORE_F32 ||
        (int64_t)d->N * d->K * d->D > kNarrowMaxElems) return KPGNN_OK;
    if (tables && d->K > 1 && !d->tablek) return KPGNN_OK;
    NarrowParams p;
    p.N = d->N; p.K = d->K; p.D = d->D; p.K_csr = d->K_csr; p.mode = d->mode;
    p.n_code0 = tables ? d->n_code0 : 0; p.n_codek = (tables && d->K > 1) ? d->n_codek : 0;
    p.total = (int64_t)d->N * d->K * d->D;
    p.rowptr = d->rowptr; p.col = d->col; p.code = d->code;
    p.src = d->x; p.s_sn = d->x_sn; p.s_sk = d->x_sk;
    p.table0 = tables ? d->table0 : nullptr; p.tablek = tables ? (d->K > 1 ? d->tablek : d->table0) : nullptr;
    p.periph = d->periph; p.p_sn = d->p_sn; p.p_sk = d->p_sk;
    p.ptab = d->periph ? nullptr : d->ptab; p.uid = d->periph ? nullptr : d->uid; p.uid_stride = d->uid_stride;
    p.xbias = d->xbias; p.eps = d->eps; p.self_term = d->mode == KPGNN_MODE_GIN ? 1 : 0;
    p.out = d->out; p.o_sn = d->o_sn; p.o_sk = d->o_sk; p.pre = d->pre; p.acc_mask = 0;
    *handled = true;
    return agg_narrow_launch(p, s);
}

int agg_narrow_bwd(const kpgnn_agg_bwd_desc* d, hipStream_t s, bool* handled) {
    *handled = false;
    const bool want_tables = d->use_tables && d->gtable0;
    if (d->D > 32 || !d->gx || d->K > 32 || d->mode == KPGNN_MODE_GCN || want_tables || d->storage != KPGNN_STORE_F32 ||
        (int64_t)d->N * d->K * d->D > kNarrowMaxElems) return KPGNN_OK;
    NarrowParams p;
    p.N = d->N; p.K = d->K; p.D = d->D; p.K_csr = d->K_csr; p.mode = KPGNN_MODE_SUM;   // (no activation on the way back)
    p.n_code0 = p.n_codek = 0;
    p.total = (int64_t)d->N * d->K * d->D;
    p.rowptr = d->rowptr_src; p.col = d->col_src; p.code = d->code_src;
    p.src = d->g; p.s_sn = d->g_sn; p.s_sk = d->g_sk;
    p.table0 = p.tablek = nullptr;
    p.periph = nullptr; p.p_sn = p.p_sk = 0; p.ptab = nullptr; p.uid = nullptr; p.uid_stride = 0;
    p.xbias = nullptr; p.eps = d->eps; p.self_term = d->mode == KPGNN_MODE_GIN ? 1 : 0;
    p.out = d->gx; p.o_sn = d->gx_sn; p.o_sk = d->gx_sk; p.pre = nullptr; p.acc_mask = d->accumulate_mask;
    *handled = true;
    return agg_narrow_launch(p, s);
}

}  // namespace kpgnn
