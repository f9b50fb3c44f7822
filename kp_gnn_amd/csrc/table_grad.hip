// Edge-code table gradients without per-edge atomics (gfx950).  Contract: include/kpgnn.h, kpgnn_table_grad.
//
// Why a separate kernel: LDS float atomics (ds_add_f32) per gathered edge row made the backward gather 5x
// slower than the gather itself (622 us vs 120 us at N=47k, K=8, D=104: a handful of hot codes serialise).
// Here the accumulation is column-private: thread t owns feature column t of every table row, the tile's
// pair list arrives sorted by (table, code), so a run of equal codes is summed in ONE register and written
// to the thread's own LDS slot when the code changes.  g is streamed once, coalesced (a row = D floats).
#include "kpgnn_common.h"

namespace kpgnn {
namespace {

constexpr int kCols = 128;   // threads per block = feature columns per block
constexpr int kEnt = 128;    // pair-list entries staged per round

struct TgParams {
    int N, K, D, NT, n0, nk;
    const int32_t* tptr;
    const uint32_t* tpack;
    const float* g; int64_t g_sn, g_sk;
    float* gt0;
    float* gtk;
};

__global__ void __launch_bounds__(kCols)
table_grad_kernel(const TgParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x;
    const int d = blockIdx.y * kCols + t;
    const bool col_ok = d < p.D;
    const int rows_per_tile = p.NT * p.K;
    float* tile = lds;                               // [rows_per_tile][kCols]
    float* acc = lds + rows_per_tile * kCols;        // [(n0 + nk)][kCols], column-private
    uint32_t* ent = reinterpret_cast<uint32_t*>(acc + (p.n0 + p.nk) * kCols);  // [kEnt]
    for (int r = 0; r < p.n0 + p.nk; ++r) acc[r * kCols + t] = 0.f;
    const int64_t num_tiles = ((int64_t)p.N + p.NT - 1) / p.NT;
    int cur = -1;       // current accumulator row (table offset + code), -1 = none
    float run = 0.f;
    for (int64_t tl = blockIdx.x; tl < num_tiles; tl += gridDim.x) {
        __syncthreads();  // previous tile fully consumed
        // ---- stream the tile of g into LDS (row = D consecutive floats, one float per thread)
        const int64_t node0 = tl * p.NT;
        for (int r0 = 0; r0 < rows_per_tile; r0 += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = r0 + u;
                const int64_t node = node0 + r / p.K;
                const int hop = r % p.K;
                v[u] = 0.f;
                if (r < rows_per_tile && node < p.N && col_ok) v[u] = p.g[node * p.g_sn + (int64_t)hop * p.g_sk + d];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (r0 + u < rows_per_tile) tile[(r0 + u) * kCols + t] = v[u];
        }
        // ---- walk the (table,code)-sorted pair list of this tile
        const int beg = p.tptr[tl], end = p.tptr[tl + 1];
        for (int base = beg; base < end; base += kEnt) {
            __syncthreads();  // ent[] free again (and, first round, the tile is complete)
            if (base + t < end) ent[t] = p.tpack[base + t];
            __syncthreads();
            const int cnt = min(kEnt, end - base);
            for (int e0 = 0; e0 < cnt; e0 += 4) {
                uint32_t en[4]; float val[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) en[u] = ent[min(e0 + u, cnt - 1)];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int hop = en[u] & 0xFFF;
                    const int nit = (en[u] >> 12) & 7;
                    const bool ok = (e0 + u < cnt) && hop < p.K;
                    val[u] = ok ? tile[(nit * p.K + hop) * kCols + t] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (e0 + u >= cnt) break;
                    if ((int)(en[u] & 0xFFF) >= p.K) continue;
                    const int cc = (int)(en[u] >> 15);                 // table<<16 | code
                    const int row = (cc >> 16) ? p.n0 + (cc & 0xFFFF) : cc;
                    if (row != cur) {                                  // wave-uniform
                        if (cur >= 0) acc[cur * kCols + t] += run;
                        cur = row;
                        run = 0.f;
                    }
                    run += val[u];
                }
            }
        }
    }
    if (cur >= 0) acc[cur * kCols + t] += run;
    if (col_ok) {
        for (int r = 0; r < p.n0; ++r) { const float v = acc[r * kCols + t]; if (v != 0.f) atomicAdd(p.gt0 + (int64_t)r * p.D + d, v); }
        for (int r = 0; r < p.nk; ++r) { const float v = acc[(p.n0 + r) * kCols + t]; if (v != 0.f) atomicAdd(p.gtk + (int64_t)r * p.D + d, v); }
    }
}

}  // namespace
}  // namespace kpgnn

using namespace kpgnn;

extern "C" int kpgnn_table_grad(const kpgnn_table_grad_desc* d, kpgnn_stream_t stream) {
    KPGNN_REQUIRE(d != nullptr, "table_grad: NULL descriptor");
    KPGNN_REQUIRE(d->N >= 0 && d->K >= 1 && d->K <= 4096 && d->D >= 1 && d->nodes_per_tile >= 1 && d->nodes_per_tile <= 8,
                  "table_grad: bad N=%d K=%d D=%d nodes_per_tile=%d", d->N, d->K, d->D, d->nodes_per_tile);
    if (d->N == 0) return KPGNN_OK;
    KPGNN_REQUIRE(d->tile_ptr && d->g && d->gtable0 && d->n_code0 >= 1, "table_grad: NULL tile_ptr/g/gtable0");
    KPGNN_REQUIRE(d->K == 1 || (d->gtablek && d->n_codek >= 1), "table_grad: missing gtablek");
    TgParams p;
    p.N = d->N; p.K = d->K; p.D = d->D; p.NT = d->nodes_per_tile; p.n0 = d->n_code0; p.nk = d->K > 1 ? d->n_codek : 0;
    p.tptr = d->tile_ptr; p.tpack = d->tile_pack; p.g = d->g; p.g_sn = d->g_sn; p.g_sk = d->g_sk;
    p.gt0 = d->gtable0; p.gtk = d->gtablek;
    const size_t lds = sizeof(float) * (size_t)kCols * ((size_t)p.NT * p.K + p.n0 + p.nk) + sizeof(uint32_t) * kEnt;
    if (lds > 160 * 1024) return fail(KPGNN_ELIMIT, "table_grad: %zu B of LDS needed (tile %dx%d rows + %d table rows)", lds, p.NT, p.K, p.n0 + p.nk);
    if (lds > 64 * 1024)
        KPGNN_HIP_TRY(hipFuncSetAttribute((const void*)table_grad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t num_tiles = ((int64_t)p.N + p.NT - 1) / p.NT;
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 8) per_cu = 8;
    int64_t gx = (int64_t)device_facts().cu_count * per_cu;
    if (gx > num_tiles) gx = num_tiles;
    dim3 grid((unsigned)gx, (unsigned)((p.D + kCols - 1) / kCols));
    hipLaunchKernelGGL(table_grad_kernel, grid, dim3(kCols), lds, (hipStream_t)stream, p);
    KPGNN_LAUNCH_CHECK("table_grad_kernel");
    return KPGNN_OK;
}
