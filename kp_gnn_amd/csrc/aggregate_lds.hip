// K-hop aggregation with a graph's hop slab STAGED IN LDS (gfx950).  Contract: include/kpgnn.h, kpgnn_aggregate_fwd (this is
// one of its kernels, chosen when the caller hands over the graph boundaries: kpgnn_agg_fwd_desc.graph_ptr).
//
// For DENSE K-hop neighbourhoods the gather of kpgnn_aggregate_fwd is bound by L2, not by HBM: run_simulation.py's 3-regular
// graphs on 1280 nodes (:100-129) have 582 K-hop pairs per node at K = 8, so a batch of 100 graphs reads 74.5 M neighbour rows of
// 64 bytes - 4.8 GB per launch out of a 65-MB tensor (round 2: 0.64 ms, 0.11 of the HBM roofline, whose algorithmic bytes are
// the 0.45 GB of pair lists).  But the rows one (graph, hop) pair can touch are one SLAB x[graph's nodes, hop, :] of
// 1280 x 16 x 4 B = 82 KB, and a CU has 160 KB of LDS at ~0.6 TB/s per CU: a block stages the slab once (coalesced 16-byte
// loads), then its sub-groups walk their destination nodes' pair lists and take every neighbour row from LDS (ds_read_b128).
// Only the pair lists and the output stream through HBM.  Pairs are added in list order, like every other kernel of
// kpgnn_aggregate_fwd (the reference's index_add_ order): results do not depend on which kernel ran.
#include "kpgnn_common.h"

namespace kpgnn {
namespace {

constexpr int kLdsThreads = 1024;

struct LdsAggParams {
    int N, K, D, K_csr, G, CH, self_term;
    const int32_t* gptr;
    const int32_t* rowptr;
    const int32_t* col;
    const float* x; int64_t x_sn, x_sk;
    const float* eps;
    float* out; int64_t o_sn, o_sk;
};

// LG = lanes per destination node = D / 4 (1, 2, 4, 8 or 16)
template <int LG>
__global__ void __launch_bounds__(kLdsThreads)
agg_lds_fwd_kernel(const LdsAggParams p) {
    extern __shared__ __attribute__((aligned(16))) float slab[];       // [nodes of the graph][D]
    constexpr int D = 4 * LG;
    const int per_hop = p.G * p.CH;
    // heavy hops first (shortest-path hops grow with k): the tail of the launch is then made of light blocks
    const int k = p.K - 1 - (int)(blockIdx.x / per_hop);
    const int rem = (int)(blockIdx.x % per_hop);
    const int g = rem / p.CH, ch = rem % p.CH;
    const int n0 = p.gptr[g], n1 = p.gptr[g + 1], ng = n1 - n0;
    const int tid = threadIdx.x;
    // ---- stage the slab x[n0 .. n1, k, :]
    for (int e = tid; e < ng * LG; e += kLdsThreads) {
        const int r = e / LG, q = e % LG;
        *reinterpret_cast<float4*>(slab + r * D + 4 * q) =
            *reinterpret_cast<const float4*>(p.x + (int64_t)(n0 + r) * p.x_sn + (int64_t)k * p.x_sk + 4 * q);
    }
    __syncthreads();
    const float eps1 = 1.0f + (p.eps ? p.eps[0] : 0.0f);
    const int per = (ng + p.CH - 1) / p.CH;
    const int i0 = n0 + ch * per, i1 = min(n1, i0 + per);
    const int sg = tid / LG, sl = tid % LG;
    const int lane = tid & (kWave - 1);
    const int sg_lane0 = lane - sl;                                    // first lane of this sub-group inside its wave
    for (int i = i0 + sg; i < i1; i += kLdsThreads / LG) {
        const int32_t* rp = p.rowptr + (int64_t)i * p.K_csr + k;
        const int b = rp[0], e = rp[1];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // the sub-group's LG lanes fetch LG pairs per request (two requests in flight), every lane then adds all of them
        for (int a = b; a < e; a += 2 * LG) {
            const int c0 = (a + sl < e) ? p.col[a + sl] - n0 : 0;
            const int c1 = (a + LG + sl < e) ? p.col[a + LG + sl] - n0 : 0;
#pragma unroll
            for (int q = 0; q < LG; ++q) {
                const int r = __shfl(c0, sg_lane0 + q);
                if (a + q < e) {
                    const float4 v = *reinterpret_cast<const float4*>(slab + r * D + 4 * sl);
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                }
            }
#pragma unroll
            for (int q = 0; q < LG; ++q) {
                const int r = __shfl(c1, sg_lane0 + q);
                if (a + LG + q < e) {
                    const float4 v = *reinterpret_cast<const float4*>(slab + r * D + 4 * sl);
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                }
            }
        }
        if (p.self_term) {                                             // GIN: + (1 + eps) * x[i, k, :]
            const float4 v = *reinterpret_cast<const float4*>(slab + (i - n0) * D + 4 * sl);
            acc.x = fmaf(eps1, v.x, acc.x); acc.y = fmaf(eps1, v.y, acc.y); acc.z = fmaf(eps1, v.z, acc.z); acc.w = fmaf(eps1, v.w, acc.w);
        }
        *reinterpret_cast<float4*>(p.out + (int64_t)i * p.o_sn + (int64_t)k * p.o_sk + 4 * sl) = acc;
    }
}

}  // namespace

// Returns KPGNN_OK with *handled = true when the launch was done here; *handled = false leaves it to the other kernels.
int agg_lds_fwd(const kpgnn_agg_fwd_desc* d, hipStream_t s, bool* handled) {
    *handled = false;
    if (!d->graph_ptr || d->num_graphs < 1 || d->max_graph_nodes < 1) return KPGNN_OK;
    // mask-only aggregation into a plain [N,K,D] output: no tables, no peripheral features, no fused combine, no saved S
    if (d->use_tables || d->periph || d->uid || d->theta || d->pre || d->xbias || !d->x || !d->out || d->storage != KPGNN_STORE_F32 ||
        d->n_dyn || (d->mode != KPGNN_MODE_GIN && d->mode != KPGNN_MODE_SUM))
        return KPGNN_OK;
    const int D = d->D, LG = D / 4;
    if (D % 4 != 0 || (LG != 1 && LG != 2 && LG != 4 && LG != 8 && LG != 16)) return KPGNN_OK;
    auto al = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
    if (!al(d->x) || !al(d->out) || d->x_sn % 4 || d->x_sk % 4 || d->o_sn % 4 || d->o_sk % 4) return KPGNN_OK;
    const size_t lds = sizeof(float) * (size_t)d->max_graph_nodes * D;
    if (lds > (size_t)device_facts().lds_per_block) return KPGNN_OK;
    LdsAggParams p;
    p.N = d->N; p.K = d->K; p.D = D; p.K_csr = d->K_csr; p.G = d->num_graphs;
    // destination chunks per (graph, hop): enough blocks for four rounds of the chip, at most 8 (every block stages the slab)
    int64_t ch = (4LL * device_facts().cu_count + (int64_t)p.G * p.K - 1) / ((int64_t)p.G * p.K);
    p.CH = (int)(ch < 1 ? 1 : (ch > 8 ? 8 : ch));
    p.self_term = d->mode == KPGNN_MODE_GIN ? 1 : 0;
    p.gptr = d->graph_ptr; p.rowptr = d->rowptr; p.col = d->col;
    p.x = d->x; p.x_sn = d->x_sn; p.x_sk = d->x_sk; p.eps = d->eps;
    p.out = d->out; p.o_sn = d->o_sn; p.o_sk = d->o_sk;
    const int64_t blocks = (int64_t)p.K * p.G * p.CH;
    if (blocks >= ((int64_t)1 << 31)) return KPGNN_OK;
#define KP_LDSAGG(L) do { KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)agg_lds_fwd_kernel<L>, lds)); \
        hipLaunchKernelGGL(agg_lds_fwd_kernel<L>, dim3((unsigned)blocks), dim3(kLdsThreads), lds, s, p); } while (0)
    switch (LG) {
        case 1: KP_LDSAGG(1); break;
        case 2: KP_LDSAGG(2); break;
        case 4: KP_LDSAGG(4); break;
        case 8: KP_LDSAGG(8); break;
        default: KP_LDSAGG(16); break;
    }
#undef KP_LDSAGG
    KPGNN_LAUNCH_CHECK("agg_lds_fwd_kernel");
    *handled = true;
    return KPGNN_OK;
}

}  // namespace kpgnn
