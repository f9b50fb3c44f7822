// LDS-staged K-hop aggregation, forward (gfx950).  Contract: include/kpgnn.h, kpgnn_aggregate_fwd with `tile_start`.
//
// The global-gather kernel (aggregate.hip) is bound by chains of dependent global loads (row pointer -> neighbour
// id -> row): at N = 47k it runs at ~2x its HBM floor.  Collated batches are block diagonal, so the node range can
// be cut into component-aligned tiles (kpgnn_csr_component_tiles) whose gathers never leave the tile.  Here a
// 512-thread workgroup owns one tile: it stages the tile's row pointers, neighbour ids/codes (as 16-bit local
// ids) and, hop by hop and double buffered, the tile's x rows in LDS with plain coalesced loads; the segmented
// sums then read LDS only.  HBM sees exactly the algorithmic bytes: x once, ids once, outputs once.
// Tiles flagged oversize (a component larger than the LDS window) take the global-gather path per node.
#include <cstdlib>
#include <initializer_list>

#include "kpgnn_common.h"

namespace kpgnn {
namespace {

constexpr int kBlockL = 512;
constexpr int kMaxSlot = 3;       // nodes per sub-group per tile

struct LdsFwdParams {
    int N, K, D, K_csr, n_code0, n_codek, mode, combine, use_tables;
    int node_cap, pair_cap, num_tiles;
    const int32_t* tile_start; const uint8_t* tile_flag;
    const int32_t* rowptr; const int32_t* col; const uint16_t* code;
    const float* x; int64_t x_sn, x_sk;
    const float* table0; const float* tablek;
    const float* periph; int64_t p_sn, p_sk;
    const float* eps;
    float* out; int64_t o_sn, o_sk;
    float* pre;
    const float* theta; float* hout;
    const float* xbias;
    const float* ptab; const int32_t* uid; int64_t uid_stride;
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 fma4(float s, float4 a, float4 b) {
    return make_float4(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y), fmaf(s, a.z, b.z), fmaf(s, a.w, b.w));
}
__device__ __forceinline__ float gelu1(float x) {
    float e2;
    return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752440f, &e2));
}

template <int G>
__global__ void __launch_bounds__(kBlockL)
agg_fwd_lds_kernel(const LdsFwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NSG = kBlockL / G;
    const int D = p.D, K = p.K, Kc = p.K_csr;
    const int ntab = p.use_tables ? (p.n_code0 + p.n_codek) * D : 0;
    float* tab0 = lds;
    float* tabk = lds + p.n_code0 * D;
    float* xbuf0 = lds + ((ntab + 3) & ~3);
    float* xbuf1 = xbuf0 + p.node_cap * D;
    uint32_t* ent = reinterpret_cast<uint32_t*>(xbuf1 + p.node_cap * D);
    int32_t* rps = reinterpret_cast<int32_t*>(ent + p.pair_cap);
    int32_t* uids = rps + (p.node_cap * Kc + 4);

    if (p.use_tables) {
        for (int t = threadIdx.x; t < p.n_code0 * D; t += kBlockL) tab0[t] = p.table0[t];
        for (int t = threadIdx.x; t < p.n_codek * D; t += kBlockL) tabk[t] = p.tablek[t];
    }
    const int sg = threadIdx.x / G, sl = threadIdx.x % G;
    const int c0 = sl * 4;
    const bool col_ok = c0 < D;
    const int MODE = p.mode;
    const bool COMBINE = p.combine != 0;
    const float eps1 = 1.0f + (p.eps ? p.eps[0] : 0.0f);
    int last_u = -1;
    float4 prow = make_float4(0.f, 0.f, 0.f, 0.f);

    for (XcdTileWalk w(p.num_tiles); w.valid(); w.next()) {
        const int n0 = p.tile_start[w.cur], n1 = p.tile_start[w.cur + 1];
        const int nn = n1 - n0;
        const bool spill = p.tile_flag[w.cur] != 0;     // gathers leave the tile: rows come from global memory
        const int e0 = p.rowptr[(int64_t)n0 * Kc];
        const int ne = spill ? 0 : p.rowptr[(int64_t)n1 * Kc] - e0;
        __syncthreads();                                // previous tile done with every LDS region
        // ---- stage row pointers, neighbour ids + codes (16-bit local id | code<<16), uids, hop-0 rows
        for (int t = threadIdx.x; t <= nn * Kc; t += kBlockL) rps[t] = p.rowptr[(int64_t)n0 * Kc + t] - e0;
        for (int t = threadIdx.x; t < ne; t += kBlockL) {
            uint32_t v = (uint32_t)(p.col[e0 + t] - n0);
            if (p.use_tables) v |= (uint32_t)p.code[e0 + t] << 16;
            ent[t] = v;
        }
        if (p.uid)  // sub-group sg stages the uids of its rows (lane k < K)
            for (int r = sg; r < nn; r += NSG)
                for (int k2 = sl; k2 < K; k2 += G) uids[r * K + k2] = p.uid[(int64_t)(n0 + r) * p.uid_stride + k2];
        // rows are staged with the compute mapping (sub-group sg -> rows sg, sg+NSG, ..; lane -> 16-B column
        // chunk): no index arithmetic beyond one multiply-add, every access a coalesced row burst
#pragma unroll
        for (int s = 0; s < kMaxSlot; ++s) {
            const int r = sg + s * NSG;
            if (r < nn && col_ok) st4(xbuf0 + r * D + c0, ld4(p.x + (int64_t)(n0 + r) * p.x_sn + c0));
        }
        __syncthreads();

        float4 hsum[kMaxSlot];
#pragma unroll
        for (int s = 0; s < kMaxSlot; ++s) hsum[s] = make_float4(0.f, 0.f, 0.f, 0.f);

        // Two hops of rows are kept in flight in registers (pfA = hop k+1, pfB = hop k+2): ~40 KB of loads per
        // workgroup outstanding while hop k is summed out of LDS.
        float4 pfA[kMaxSlot], pfB[kMaxSlot];
#pragma unroll
        for (int u = 0; u < kMaxSlot; ++u) {
            const int r = sg + u * NSG;
            if (r < nn && col_ok) {
                const float* src = p.x + (int64_t)(n0 + r) * p.x_sn + c0;
                if (1 < K) pfA[u] = ld4(src + (int64_t)1 * p.x_sk);
                if (2 < K) pfB[u] = ld4(src + (int64_t)2 * p.x_sk);
            }
        }
        for (int k = 0; k < K; ++k) {
            float* xb = (k & 1) ? xbuf1 : xbuf0;
            float* xn = (k & 1) ? xbuf0 : xbuf1;
            const float* tab = k == 0 ? tab0 : tabk;
            float4 th = make_float4(0.f, 0.f, 0.f, 0.f), xbv = th;
            if (COMBINE && col_ok) th = ld4(p.theta + k * D + c0);
            const bool biased = p.xbias != nullptr && k >= 1;
            if (biased && col_ok) xbv = ld4(p.xbias + c0);
#pragma unroll
            for (int s = 0; s < kMaxSlot; ++s) {
                const int r = sg + s * NSG;
                if (r >= nn) break;
                const int64_t i = n0 + r;
                const int beg = rps[r * Kc + k], end = rps[r * Kc + k + 1];
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!spill) {
                    for (int a = beg; a < end; a += 4) {
                        uint32_t en[4]; float4 row[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) en[u] = ent[min(a + u, end - 1)];
                        if (col_ok) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                row[u] = ld4(xb + (en[u] & 0xFFFF) * D + c0);
                                if (p.use_tables) row[u] = add4(row[u], ld4(tab + (en[u] >> 16) * D + c0));
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u) if (a + u < end) acc = add4(acc, row[u]);
                        }
                    }
                } else if (col_ok) {  // oversize component: gather from global memory
                    for (int a = beg; a < end; ++a) {
                        const int j = p.col[e0 + a];
                        float4 row = ld4(p.x + (int64_t)j * p.x_sn + (int64_t)k * p.x_sk + c0);
                        if (p.use_tables) row = add4(row, ld4(tab + (int)p.code[e0 + a] * D + c0));
                        acc = add4(acc, row);
                    }
                }
                if (!col_ok) continue;
                // ---- epilogue of segment (i, k)
                float4 v = acc;
                if (biased) v = fma4((float)(end - beg), xbv, v);
                if (p.pre) st4(p.pre + (i * K + k) * (int64_t)D + c0, v);
                if (MODE == KPGNN_MODE_GINPLUS) v = make_float4(gelu1(v.x), gelu1(v.y), gelu1(v.z), gelu1(v.w));
                if (p.periph) v = add4(v, ld4(p.periph + i * p.p_sn + (int64_t)k * p.p_sk + c0));
                else if (p.uid) {
                    const int u = uids[r * K + k];
                    if (u != last_u) { prow = ld4(p.ptab + (int64_t)u * D + c0); last_u = u; }
                    v = add4(v, prow);
                }
                if (MODE == KPGNN_MODE_GIN) v = fma4(eps1, add4(ld4(xb + r * D + c0), xbv), v);
                if (COMBINE) {
                    hsum[s].x = fmaf(th.x, v.x, hsum[s].x); hsum[s].y = fmaf(th.y, v.y, hsum[s].y);
                    hsum[s].z = fmaf(th.z, v.z, hsum[s].z); hsum[s].w = fmaf(th.w, v.w, hsum[s].w);
                } else {
                    st4(p.out + i * p.o_sn + (int64_t)k * p.o_sk + c0, v);
                }
            }
#pragma unroll
            for (int u = 0; u < kMaxSlot; ++u) {
                const int r = sg + u * NSG;
                if (r < nn && col_ok) {
                    if (k + 1 < K) st4(xn + r * D + c0, pfA[u]);
                    pfA[u] = pfB[u];
                    if (k + 3 < K) pfB[u] = ld4(p.x + (int64_t)(n0 + r) * p.x_sn + (int64_t)(k + 3) * p.x_sk + c0);
                }
            }
            __syncthreads();
        }
        if (COMBINE && col_ok) {
#pragma unroll
            for (int s = 0; s < kMaxSlot; ++s) {
                const int r = sg + s * NSG;
                if (r < nn) st4(p.hout + (int64_t)(n0 + r) * D + c0, hsum[s]);
            }
        }
    }
}

}  // namespace

// Host launcher (called from kpgnn_aggregate_fwd when the descriptor carries tiles and the shape qualifies).
// Returns KPGNN_ELIMIT when the shape does not fit this kernel: the caller then runs the global-gather kernel.
int launch_agg_fwd_lds(const kpgnn_agg_fwd_desc* d, hipStream_t s) {
    if (d->mode == KPGNN_MODE_GCN || (d->D % 4) != 0 || d->D > 256) return KPGNN_ELIMIT;
    for (const void* q : {(const void*)d->x, (const void*)d->periph, (const void*)d->out, (const void*)d->pre,
                          (const void*)d->table0, (const void*)d->tablek, (const void*)d->theta, (const void*)d->hout,
                          (const void*)d->xbias, (const void*)(d->periph ? nullptr : d->ptab)})
        if (q && ((uintptr_t)q & 15)) return KPGNN_ELIMIT;
    for (int64_t st : {d->x_sn, d->x_sk, d->periph ? d->p_sn : 0, d->periph ? d->p_sk : 0, d->out ? d->o_sn : 0, d->out ? d->o_sk : 0})
        if (st % 4) return KPGNN_ELIMIT;
    int g = 4;
    while (g * 4 < d->D) g <<= 1;
    if (g < 8) g = 8;
    const int nsg = kBlockL / g;
    if (d->tile_node_cap > kMaxSlot * nsg) return KPGNN_ELIMIT;
    const bool combine = d->theta != nullptr;
    LdsFwdParams p;
    p.N = d->N; p.K = d->K; p.D = d->D; p.K_csr = d->K_csr; p.n_code0 = d->use_tables ? d->n_code0 : 0;
    p.n_codek = (d->use_tables && d->K > 1) ? d->n_codek : 0; p.mode = d->mode; p.combine = combine ? 1 : 0;
    p.use_tables = d->use_tables;
    p.node_cap = d->tile_node_cap; p.pair_cap = d->tile_pair_cap; p.num_tiles = d->num_tiles;
    p.tile_start = d->tile_start; p.tile_flag = d->tile_flag;
    p.rowptr = d->rowptr; p.col = d->col; p.code = d->code;
    p.x = d->x; p.x_sn = d->x_sn; p.x_sk = d->x_sk; p.table0 = d->table0; p.tablek = d->tablek;
    p.periph = d->periph; p.p_sn = d->p_sn; p.p_sk = d->p_sk; p.eps = d->eps;
    p.out = d->out; p.o_sn = d->o_sn; p.o_sk = d->o_sk; p.pre = d->pre; p.theta = d->theta; p.hout = d->hout;
    p.xbias = d->xbias; p.ptab = d->periph ? nullptr : d->ptab; p.uid = d->periph ? nullptr : d->uid; p.uid_stride = d->uid_stride;
    const size_t ntab = (((size_t)(p.n_code0 + p.n_codek) * p.D) + 3) & ~(size_t)3;
    const size_t lds = sizeof(float) * (ntab + 2 * (size_t)p.node_cap * p.D) + sizeof(uint32_t) * (size_t)p.pair_cap +
                       sizeof(int32_t) * ((size_t)p.node_cap * p.K_csr + 4) + sizeof(int32_t) * (size_t)p.node_cap * p.K + 64;
    if (lds > 160 * 1024) return KPGNN_ELIMIT;
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > 4) per_cu = 4;
    if (const char* e = getenv("KPGNN_LDS_BLOCKS_PER_CU")) { const int v = atoi(e); if (v > 0) per_cu = v; }
    int64_t grid = (int64_t)device_facts().cu_count * per_cu;
    if (grid > p.num_tiles) grid = p.num_tiles;
    if (grid >= kNumXcd) grid = grid / kNumXcd * kNumXcd;
    if (grid < 1) grid = 1;
#define KP_LDS(GG)                                                                                                        \
    do {                                                                                                                  \
        if (lds > 64 * 1024)                                                                                              \
            KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)agg_fwd_lds_kernel<GG>, lds)); \
        hipLaunchKernelGGL((agg_fwd_lds_kernel<GG>), dim3((unsigned)grid), dim3(kBlockL), lds, s, p);                    \
    } while (0)
    switch (g) {
        case 8: KP_LDS(8); break;
        case 16: KP_LDS(16); break;
        case 32: KP_LDS(32); break;
        case 64: KP_LDS(64); break;
        default: return KPGNN_ELIMIT;
    }
#undef KP_LDS
    KPGNN_LAUNCH_CHECK("agg_fwd_lds_kernel");
    return KPGNN_OK;
}

}  // namespace kpgnn
