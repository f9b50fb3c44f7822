// Table gradients without per-edge atomics (gfx950).  Contract: include/kpgnn.h, kpgnn_table_grad.
//
//   gtable_t[c,:]  = sum over active pairs (i,k) of table t with code c of g[i,k,:]          (edge-code tables)
//   gdict[u,:]     = sum over (i,k) with uid[i,k] == u of theta[k,:] * gh[i,:]               (peripheral dictionary)
//
// Why a separate kernel: LDS float atomics (ds_add_f32) per gathered edge row made the backward gather 5x
// slower than the gather itself (622 us vs 120 us at N=47k, K=8, D=104: a handful of hot codes serialise),
// and flushing per-block tables with global atomics made every block hammer the same few KB.
// Here the accumulation is column-private: thread t owns feature column t of every table row, the tile's
// pair list arrives sorted by (table, code), so a run of equal codes is summed in ONE register and written
// to the thread's own LDS slot when the code changes.  g is streamed once, coalesced (a row = D floats),
// the next tile's loads are in flight while the current tile is walked, and the pair list is wave-uniform
// (scalar loads).  Per-block partial tables go to a workspace slab with plain stores; a second launch adds
// the slabs in block order, so the result is deterministic (bitwise reproducible).
#include "kpgnn_common.h"

namespace kpgnn {
namespace {

constexpr int kCols = 128;    // threads per block = feature columns per block
constexpr int kMaxRows = 64;  // tile rows held in registers while the previous tile is being walked

struct TgParams {
    int N, K, D, NT, n0, nk, U, dict_src;
    const int32_t* tptr;
    const uint32_t* tpack;
    const float* g; int64_t g_sn, g_sk;
    // peripheral dictionary (optional): gdict[uid[i*uid_stride + k]] += theta[k,:] * gh[i,:]
    const int32_t* uid; int64_t uid_stride;
    const float* theta;
    const float* gh;
    float* slab;          // [gridDim.x][n0 + nk + U][D]
};

// Load one tile's column-t values into registers (rows beyond the tile / past N read as 0).
__device__ __forceinline__ void load_tile_regs(const TgParams& p, int64_t tl, int d, bool col_ok, int rows,
                                               float (&v)[kMaxRows]) {
    const int64_t node0 = tl * p.NT;
#pragma unroll
    for (int r = 0; r < kMaxRows; ++r) {
        v[r] = 0.f;
        if (r < rows) {
            const int64_t node = node0 + r / p.K;
            const int hop = r % p.K;
            if (node < p.N && col_ok) v[r] = p.g[node * p.g_sn + (int64_t)hop * p.g_sk + d];
        }
    }
}

__global__ void __launch_bounds__(kCols)
table_grad_kernel(const TgParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x;
    const int d = blockIdx.y * kCols + t;
    const bool col_ok = d < p.D;
    const int rows = p.NT * p.K;                     // <= kMaxRows (checked by the launcher)
    const int R = p.n0 + p.nk + p.U;
    float* tile = lds;                               // [rows][kCols]
    float* acc = lds + rows * kCols;                 // [R][kCols], column-private
    for (int r = 0; r < R; ++r) acc[r * kCols + t] = 0.f;
    const int64_t num_tiles = ((int64_t)p.N + p.NT - 1) / p.NT;
    int cur = -1;       // current accumulator row (table offset + code), -1 = none
    float run = 0.f;
    int ucur = -1;      // current dictionary row
    float urun = 0.f;
    float v[kMaxRows];
    int64_t tl = blockIdx.x;
    if (tl < num_tiles) load_tile_regs(p, tl, d, col_ok, rows, v);
    for (; tl < num_tiles; tl += gridDim.x) {
        int beg = 0, end = 0;                        // uniform (scalar) loads, ahead of the prefetch
        if (p.tptr) { beg = p.tptr[tl]; end = p.tptr[tl + 1]; }
        __syncthreads();                             // previous tile fully walked
#pragma unroll
        for (int r = 0; r < kMaxRows; ++r)
            if (r < rows) tile[r * kCols + t] = v[r];
        __syncthreads();
        if (tl + gridDim.x < num_tiles)              // next tile's loads fly while this tile is walked
            load_tile_regs(p, tl + gridDim.x, d, col_ok, rows, v);
        // ---- walk the (table,code)-sorted pair list of this tile; the list is wave-uniform -> scalar loads
        for (int e0 = beg; e0 < end; e0 += 8) {
            uint32_t en[8]; float val[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) en[u] = p.tpack[min(e0 + u, end - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int hop = en[u] & 0xFFF;
                const int nit = (en[u] >> 12) & 7;
                const bool ok = (e0 + u < end) && hop < p.K;
                val[u] = tile[(ok ? (nit * p.K + hop) : 0) * kCols + t];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int hop = en[u] & 0xFFF;
                if (e0 + u < end && hop < p.K) {
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
        // ---- peripheral dictionary: rows in natural order, equal uids (the common case) stay in a register
        if (p.U > 0) {
            const int64_t node0 = tl * p.NT;
            for (int n = 0; n < p.NT; ++n) {
                const int64_t node = node0 + n;
                if (node >= p.N) break;
                const float ghv = (p.dict_src == 1 && col_ok) ? p.gh[node * p.D + d] : 0.f;
                for (int k = 0; k < p.K; ++k) {
                    const int u = p.uid[node * p.uid_stride + k];      // uniform
                    if (u != ucur) {
                        if (ucur >= 0) acc[(p.n0 + p.nk + ucur) * kCols + t] += urun;
                        ucur = u;
                        urun = 0.f;
                    }
                    if (p.dict_src == 1) urun = fmaf(col_ok ? p.theta[k * p.D + d] : 0.f, ghv, urun);
                    else urun += tile[(n * p.K + k) * kCols + t];
                }
            }
        }
    }
    if (cur >= 0) acc[cur * kCols + t] += run;
    if (ucur >= 0) acc[(p.n0 + p.nk + ucur) * kCols + t] += urun;
    if (col_ok) {
        float* out = p.slab + (int64_t)blockIdx.x * R * p.D + d;
        for (int r = 0; r < R; ++r) out[(int64_t)r * p.D] = acc[r * kCols + t];
    }
}

// out[e] = sum_b slab[b][e]   (fixed order: deterministic).  64 outputs x 16 slices of the block range per WG.
__global__ void __launch_bounds__(1024)
slab_reduce_kernel(const float* __restrict__ slab, int nslab, int64_t elems, float* __restrict__ out0, int64_t n_out0,
                   float* __restrict__ out1, int64_t n_out1, float* __restrict__ out2) {
    __shared__ float part[16][64];
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 64 + lane;
    float s = 0.f;
    if (e < elems)
        for (int b = slice; b < nslab; b += 16) s += slab[(int64_t)b * elems + e];
    part[slice][lane] = s;
    __syncthreads();
    if (slice == 0 && e < elems) {
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) tot += part[q][lane];
        if (e < n_out0) out0[e] = tot;
        else if (e < n_out0 + n_out1) out1[e - n_out0] = tot;
        else out2[e - n_out0 - n_out1] = tot;
    }
}

}  // namespace

int slab_reduce(const float* slab, int nslab, int64_t elems, float* out0, int64_t n0, float* out1, int64_t n1,
                float* out2, hipStream_t s) {
    if (elems <= 0) return KPGNN_OK;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((elems + 63) / 64)), dim3(1024), 0, s, slab, nslab, elems,
                       out0, n0, out1, n1, out2);
    KPGNN_LAUNCH_CHECK("slab_reduce_kernel");
    return KPGNN_OK;
}

namespace {

struct Plan { int grid_x, grid_y; size_t lds, ws_bytes; int R; };

int make_plan(int N, int K, int D, int NT, int n0, int nk, int U, Plan* pl) {
    if (NT * K > kMaxRows)
        return fail(KPGNN_ELIMIT, "table_grad: nodes_per_tile*K = %d rows exceed the %d-row register tile", NT * K, kMaxRows);
    pl->R = n0 + nk + U;
    pl->lds = sizeof(float) * (size_t)kCols * ((size_t)NT * K + pl->R);
    if (pl->lds > 160 * 1024)
        return fail(KPGNN_ELIMIT, "table_grad: %zu B of LDS needed (tile %dx%d rows + %d table rows)", pl->lds, NT, K, pl->R);
    const int64_t num_tiles = ((int64_t)N + NT - 1) / NT;
    int per_cu = (int)((160 * 1024) / pl->lds);
    per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
    int64_t gx = (int64_t)device_facts().cu_count * per_cu;
    if (gx > num_tiles) gx = num_tiles;
    if (gx < 1) gx = 1;
    pl->grid_x = (int)gx;
    pl->grid_y = (D + kCols - 1) / kCols;
    pl->ws_bytes = sizeof(float) * (size_t)gx * pl->R * D;
    return KPGNN_OK;
}

}  // namespace
}  // namespace kpgnn

using namespace kpgnn;

extern "C" size_t kpgnn_table_grad_workspace_bytes(int32_t N, int32_t K, int32_t D, int32_t nodes_per_tile,
                                                   int32_t n_code0, int32_t n_codek, int32_t n_dict) {
    Plan pl;
    if (N <= 0 || K < 1 || D < 1 || nodes_per_tile < 1) return 0;
    if (make_plan(N, K, D, nodes_per_tile, n_code0, K > 1 ? n_codek : 0, n_dict, &pl) != KPGNN_OK) return 0;
    return pl.ws_bytes;  // 0 also means "does not fit": the caller then takes its atomic fallback
}

extern "C" int kpgnn_table_grad(const kpgnn_table_grad_desc* d, kpgnn_stream_t stream) {
    KPGNN_REQUIRE(d != nullptr, "table_grad: NULL descriptor");
    KPGNN_REQUIRE(d->N >= 0 && d->K >= 1 && d->K <= 4096 && d->D >= 1 && d->nodes_per_tile >= 1 && d->nodes_per_tile <= 8,
                  "table_grad: bad N=%d K=%d D=%d nodes_per_tile=%d", d->N, d->K, d->D, d->nodes_per_tile);
    if (d->N == 0) return KPGNN_OK;
    KPGNN_REQUIRE(d->g != nullptr, "table_grad: NULL g");
    const bool edges = d->tile_ptr != nullptr;
    KPGNN_REQUIRE(!edges || (d->gtable0 && d->n_code0 >= 1 && (d->K == 1 || (d->gtablek && d->n_codek >= 1))),
                  "table_grad: missing gtable0/gtablek");
    KPGNN_REQUIRE(d->n_dict >= 0 && (d->n_dict == 0 || (d->uid && d->gdict && d->uid_stride >= d->K &&
                  (d->dict_src == 2 || (d->dict_src == 1 && d->theta && d->gh)))),
                  "table_grad: dictionary gradient needs uid/gdict and (theta, gh) or dict_src 2");
    KPGNN_REQUIRE(edges || d->n_dict > 0, "table_grad: nothing to do");
    TgParams p;
    p.N = d->N; p.K = d->K; p.D = d->D; p.NT = d->nodes_per_tile;
    p.n0 = edges ? d->n_code0 : 0; p.nk = (edges && d->K > 1) ? d->n_codek : 0;
    p.U = d->n_dict; p.dict_src = d->dict_src;
    p.tptr = d->tile_ptr; p.tpack = d->tile_pack; p.g = d->g; p.g_sn = d->g_sn; p.g_sk = d->g_sk;
    p.uid = d->uid; p.uid_stride = d->uid_stride; p.theta = d->theta; p.gh = d->gh;
    Plan pl;
    int rc = make_plan(p.N, p.K, p.D, p.NT, p.n0, p.nk, p.U, &pl);
    if (rc != KPGNN_OK) return rc;
    KPGNN_REQUIRE(d->workspace && d->workspace_bytes >= pl.ws_bytes, "table_grad: workspace too small (%zu < %zu)",
                  (size_t)d->workspace_bytes, pl.ws_bytes);
    p.slab = (float*)d->workspace;
    hipStream_t s = (hipStream_t)stream;
    if (pl.lds > 64 * 1024)
        KPGNN_HIP_TRY(hipFuncSetAttribute((const void*)table_grad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds));
    hipLaunchKernelGGL(table_grad_kernel, dim3(pl.grid_x, pl.grid_y), dim3(kCols), pl.lds, s, p);
    KPGNN_LAUNCH_CHECK("table_grad_kernel");
    return slab_reduce(p.slab, pl.grid_x, (int64_t)pl.R * p.D, d->gtable0, (int64_t)p.n0 * p.D, d->gtablek,
                       (int64_t)p.nk * p.D, d->gdict, s);
}
