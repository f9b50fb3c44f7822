// Table / dictionary gradients as a count-matrix product on the fp32 matrix cores (gfx950).
// Contract: include/kpgnn.h, kpgnn_table_grad (this is its default kernel; table_grad.hip keeps the register-walk
// kernel for the fused pre-pass and for shapes beyond the limits below).
//
//   gtable[r, :] = sum over the tile rows (node, hop) of  C[r, row] * g[row, :]
// where C[r, row] = number of active pairs of segment `row` whose (table, code) maps to accumulator row r, plus one
// dictionary row per (node, hop).  C is a small integer matrix (R <= 256 accumulator rows x <= 128 tile rows), built
// per tile in LDS with integer atomics (one per pair, spread over the matrix: no hot address), and C x g_tile runs on
// v_mfma_f32_16x16x4_f32 (exact fp32 products of small integers).  Against the register walk this removes the
// sequential, scalar-issue-bound per-pair loop (1.0 M pairs: ~45 us + ~60 us for unsorted dictionary rows at
// D = 13, where 115 of 128 column lanes idled) and makes the cost independent of how the pair list is ordered:
// the matrix-core work is R_pad x D_pad x rows per tile, i.e. ~7 us (D = 13) .. ~50 us (D = 104) per launch.
// Accumulators live in registers for the whole launch; per-block partials are added in block order (deterministic).
#include <cstdlib>

#include "kpgnn_common.h"

namespace kpgnn {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kTgmThreads = 256;
constexpr int kTgmWaves = 4;
constexpr int kTgmPF = 8;     // float4 registers per thread of the prefetched g tile (covers 8192 floats)

__device__ __forceinline__ int mult_of(uint32_t w) { return (int)((w >> 6) & 0x3Fu) + 1; }   // merged-entry multiplicity

struct TgmParams {
    const int32_t* n_dyn;
    int N, K, D, NT, n0, U, dict_src;
    int RE, REp, Rp, R, MT, MTE, rows, QS, CP, NTILES, KQ, vec4;
    const int32_t* tptr;
    const uint32_t* tpack;
    const float* g;
    const int32_t* uid; int64_t uid_stride;
    const float* theta;
    const float* gh;
    float* slab;              // [gridDim.x * KQ][R][D]
};

template <int MAXIT, int MTMAX>
__global__ void __launch_bounds__(kTgmThreads)
table_grad_mfma_kernel(TgmParams p) {
    p.N = live_rows(p.N, p.n_dyn);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int D = p.D, K = p.K, rows = p.rows, CP = p.CP, QS = p.QS;
    const int tile_floats = rows * D;
    const int tile_alloc = (4 * QS * D + 3) & ~3;         // + up to 3 all-zero slack rows (the k-dim is cut in 4s)
    float* tile = lds;
    int* cnt = reinterpret_cast<int*>(tile + tile_alloc);   // [Rp][CP]
    float* thl = reinterpret_cast<float*>(cnt + p.Rp * CP); // [K][D]   (dict_src 1)
    float* ghs = thl + K * D;                               // [NT][D]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
    const bool dsrc1 = p.U > 0 && p.dict_src == 1;

    for (int i = tid; i < tile_alloc; i += kTgmThreads) tile[i] = 0.f;
    for (int i = tid; i < p.Rp * CP; i += kTgmThreads) cnt[i] = 0;
    if (dsrc1)
        for (int i = tid; i < K * D; i += kTgmThreads) thl[i] = p.theta[i];

    // this wave's work items: (column tile nt, k-range kq); all items of a wave share kq
    const int nitems = p.NTILES * p.KQ;
    int nt_of[MAXIT];
    bool it_on[MAXIT];
#pragma unroll
    for (int j = 0; j < MAXIT; ++j) {
        const int it = wave + j * kTgmWaves;
        it_on[j] = it < nitems;
        nt_of[j] = it_on[j] ? it / p.KQ : 0;
    }
    const int kq = (wave < nitems) ? wave % p.KQ : 0;
    const int q0 = kq * QS / p.KQ, q1 = (kq + 1) * QS / p.KQ;
    // (node, hop) of tile row 4*q0 + lq, advanced by 4 rows per k-step (dict_src 1: the B operand is theta[k]*gh[n])
    const int row0 = 4 * q0 + lq;
    const int n_start = row0 / K, k_start = row0 - n_start * K;

    f32x4 acc[MAXIT][MTMAX];
#pragma unroll
    for (int j = 0; j < MAXIT; ++j)
#pragma unroll
        for (int m = 0; m < MTMAX; ++m) acc[j][m] = {0.f, 0.f, 0.f, 0.f};

    const int64_t num_tiles = ((int64_t)p.N + p.NT - 1) / p.NT;
    const int64_t total = (int64_t)p.N * K * D;
    float4 pf[kTgmPF];
    auto issue_tile = [&](int64_t tl) {
        if (!p.vec4) return;
        const int64_t base = tl * tile_floats;
#pragma unroll
        for (int q = 0; q < kTgmPF; ++q) {
            const int i = 4 * (tid + q * kTgmThreads);
            pf[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < tile_floats && base + i < total) pf[q] = *reinterpret_cast<const float4*>(p.g + base + i);
        }
    };
    // the pair-list window and this thread's first entry / dictionary row of the NEXT tile travel in registers too
    int nbeg = 0, nend = 0, nu = -1;
    uint32_t nw = 0xFFFFFFFFu;
    auto issue_meta = [&](int64_t tl) {
        nbeg = nend = 0; nw = 0xFFFFFFFFu; nu = -1;
        if (p.tptr) {
            nbeg = p.tptr[tl]; nend = p.tptr[tl + 1];
            if (nbeg + tid < nend) nw = p.tpack[nbeg + tid];
        }
        if (p.U > 0 && tid < rows) {
            const int n = tid / K;
            const int64_t node = tl * p.NT + n;
            if (node < p.N) nu = p.uid[node * p.uid_stride + (tid - n * K)];
        }
    };
    if ((int64_t)blockIdx.x < num_tiles) { issue_tile(blockIdx.x); issue_meta(blockIdx.x); }

    auto cell_of = [&](uint32_t w) -> int {          // LDS index of the count cell of a packed pair, -1 = skip
        const int hop = (int)(w & 0x3F);
        if (hop >= K) return -1;
        const int nit = (int)((w >> 12) & 7);
        const int cc = (int)(w >> 15);               // table<<16 | code
        const int r = (cc >> 16) ? p.n0 + (cc & 0xFFFF) : cc;
        if (r >= p.RE) return -1;
        return r * CP + nit * K + hop;
    };

    for (int64_t tl = blockIdx.x; tl < num_tiles; tl += gridDim.x) {
        const int64_t base = tl * tile_floats;
        const int beg = nbeg, end = nend, myu = nu;
        const uint32_t myw = nw;
        __syncthreads();                             // previous tile: MFMA reads and cell resets are done
        if (p.vec4) {
#pragma unroll
            for (int q = 0; q < kTgmPF; ++q) {
                const int i = 4 * (tid + q * kTgmThreads);
                if (i < tile_floats) *reinterpret_cast<float4*>(tile + i) = pf[q];
            }
            for (int i = 4 * (tid + kTgmPF * kTgmThreads); i < tile_floats; i += 4 * kTgmThreads) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (base + i < total) v = *reinterpret_cast<const float4*>(p.g + base + i);
                *reinterpret_cast<float4*>(tile + i) = v;
            }
        } else {
            for (int i = tid; i < tile_floats; i += kTgmThreads) tile[i] = (base + i < total) ? p.g[base + i] : 0.f;
        }
        // count matrix of this tile
        {
            const int c = cell_of(myw);
            if (c >= 0) atomicAdd(&cnt[c], mult_of(myw));
            for (int e = beg + tid + kTgmThreads; e < end; e += kTgmThreads) {
                const uint32_t w2 = p.tpack[e];
                const int c2 = cell_of(w2);
                if (c2 >= 0) atomicAdd(&cnt[c2], mult_of(w2));
            }
            if ((unsigned)myu < (unsigned)p.U) cnt[(p.REp + myu) * CP + tid] = 1;   // one dictionary row per tile row
        }
        if (dsrc1) {
            const int64_t node0 = tl * p.NT;
            for (int i = tid; i < p.NT * D; i += kTgmThreads) {
                const int n = i / D;
                ghs[i] = (node0 + n < p.N) ? p.gh[node0 * D + i] : 0.f;
            }
        }
        __syncthreads();
        {
            const int64_t nx = tl + gridDim.x;
            if (nx < num_tiles) { issue_tile(nx); issue_meta(nx); }
        }
        // C x g_tile
        if (wave < nitems) {
            int n_idx = n_start, k_idx = k_start;
            for (int q = q0; q < q1; ++q) {
                const int row = 4 * q + lq;
                float b[MAXIT], bd[MAXIT];
#pragma unroll
                for (int j = 0; j < MAXIT; ++j) {
                    b[j] = 0.f; bd[j] = 0.f;
                    if (it_on[j]) {
                        const int col = nt_of[j] * 16 + lr;
                        const int cc = col < D ? col : D - 1;
                        const float v = tile[row * D + cc];
                        b[j] = col < D ? v : 0.f;
                        if (dsrc1) {
                            const int nn = n_idx < p.NT ? n_idx : p.NT - 1;
                            const float w = thl[k_idx * D + cc] * ghs[nn * D + cc];
                            bd[j] = (col < D && n_idx < p.NT) ? w : 0.f;
                        }
                    }
                }
#pragma unroll
                for (int m = 0; m < MTMAX; ++m) {
                    if (m < p.MT) {
                        const float a = (float)cnt[(m * 16 + lr) * CP + row];
                        const bool dict_tile = dsrc1 && m >= p.MTE;
#pragma unroll
                        for (int j = 0; j < MAXIT; ++j)
                            if (it_on[j]) acc[j][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, dict_tile ? bd[j] : b[j], acc[j][m], 0, 0, 0);
                    }
                }
                k_idx += 4;
                while (k_idx >= K) { k_idx -= K; ++n_idx; }
            }
        }
        __syncthreads();
        // reset the touched cells (cheaper than clearing Rp x CP words per tile)
        {
            const int c = cell_of(myw);
            if (c >= 0) cnt[c] = 0;
            for (int e = beg + tid + kTgmThreads; e < end; e += kTgmThreads) {
                const int c2 = cell_of(p.tpack[e]);
                if (c2 >= 0) cnt[c2] = 0;
            }
            if ((unsigned)myu < (unsigned)p.U) cnt[(p.REp + myu) * CP + tid] = 0;
        }
    }
    // partial tables -> slab slice (block, kq): rows [0, RE) edge codes, [RE, RE+U) dictionary
    if (wave < nitems) {
        float* out = p.slab + ((int64_t)blockIdx.x * p.KQ + kq) * ((int64_t)p.R * D);
#pragma unroll
        for (int j = 0; j < MAXIT; ++j) {
            if (!it_on[j]) continue;
            const int col = nt_of[j] * 16 + lr;
            if (col >= D) continue;
#pragma unroll
            for (int m = 0; m < MTMAX; ++m) {
                if (m < p.MT) {
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const int rr = m * 16 + 4 * lq + r4;
                        int orow = -1;
                        if (rr < p.RE) orow = rr;
                        else if (rr >= p.REp && rr - p.REp < p.U) orow = p.RE + (rr - p.REp);
                        if (orow >= 0) out[(int64_t)orow * D + col] = acc[j][m][r4];
                    }
                }
            }
        }
    }
}

struct TgmPlan { int rows, QS, CP, RE, REp, Rp, R, MT, MTE, NTILES, KQ, maxit, mtmax, grid; size_t lds, ws_bytes; };

bool tgm_plan(int N, int K, int D, int NT, int n0, int nk, int U, bool dsrc1, TgmPlan* pl) {
    if (N < 1 || K < 1 || D < 1 || NT < 1 || NT > 8) return false;
    pl->rows = NT * K;
    if (pl->rows > 128 || D > 256) return false;
    pl->QS = (pl->rows + 3) / 4;
    const int cp = 4 * pl->QS;
    pl->CP = cp + ((4 - cp % 8) + 8) % 8;             // pitch = 4 (mod 8): conflict-free 16 x 4 operand reads
    pl->RE = n0 + nk;
    pl->REp = (pl->RE + 15) & ~15;
    pl->Rp = pl->REp + ((U + 15) & ~15);
    pl->R = pl->RE + U;
    if (pl->Rp < 16 || pl->Rp > 256) return false;
    pl->MT = pl->Rp / 16;
    pl->MTE = pl->REp / 16;
    pl->NTILES = (D + 15) / 16;
    int kq = pl->NTILES >= 4 ? 1 : (pl->NTILES == 1 ? 4 : (pl->NTILES == 2 ? 2 : 1));
    if (kq > pl->QS) kq = pl->QS;
    pl->KQ = kq;
    const int items = pl->NTILES * kq;
    const int per_wave = (items + kTgmWaves - 1) / kTgmWaves;
    pl->maxit = per_wave <= 1 ? 1 : (per_wave <= 2 ? 2 : 4);
    if (per_wave > 4) return false;
    pl->mtmax = pl->MT <= 4 ? 4 : (pl->MT <= 8 ? 8 : 16);
    if (pl->maxit * pl->mtmax > 32) return false;
    const size_t tile_alloc = ((size_t)4 * pl->QS * D + 3) & ~(size_t)3;
    pl->lds = sizeof(float) * (tile_alloc + (size_t)pl->Rp * pl->CP + (dsrc1 ? (size_t)(K + NT) * D : 0));
    const size_t cap = (size_t)device_facts().lds_per_block;
    if (pl->lds > cap) return false;
    const int64_t tiles = ((int64_t)N + NT - 1) / NT;
    int per_cu = (int)(cap / pl->lds);
    per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
    int64_t g = (int64_t)device_facts().cu_count * per_cu;
    if (g > tiles) g = tiles;
    pl->grid = (int)(g < 1 ? 1 : g);
    pl->ws_bytes = sizeof(float) * (size_t)pl->grid * kq * (size_t)pl->R * D;
    return true;
}

}  // namespace

size_t table_grad_mfma_ws_bytes(int N, int K, int D, int NT, int n0, int nk, int U) {
    TgmPlan pl;
    size_t a = 0;
    for (int ds = 0; ds < 2; ++ds)         // (the LDS footprint, hence the grid, depends on the dictionary source)
        if (tgm_plan(N, K, D, NT, n0, nk, U, ds != 0, &pl) && pl.ws_bytes > a) a = pl.ws_bytes;
    return a;
}

// Returns KPGNN_OK with *handled = true when the launch was done here; *handled = false leaves it to the walk kernel.
int table_grad_mfma(const kpgnn_table_grad_desc* d, hipStream_t s, bool* handled) {
    *handled = false;
    const bool edges = d->tile_ptr != nullptr;
    const int n0 = edges ? d->n_code0 : 0, nk = (edges && d->K > 1) ? d->n_codek : 0;
    const bool dsrc1 = d->n_dict > 0 && d->dict_src == 1;
    TgmPlan pl;
    if (!tgm_plan(d->N, d->K, d->D, d->nodes_per_tile, n0, nk, d->n_dict, dsrc1, &pl)) return KPGNN_OK;
    if (!d->workspace || d->workspace_bytes < pl.ws_bytes) return KPGNN_OK;
    TgmParams p;
    p.N = d->N; p.n_dyn = d->n_dyn; p.K = d->K; p.D = d->D; p.NT = d->nodes_per_tile; p.n0 = n0; p.U = d->n_dict; p.dict_src = d->dict_src;
    p.RE = pl.RE; p.REp = pl.REp; p.Rp = pl.Rp; p.R = pl.R; p.MT = pl.MT; p.MTE = pl.MTE; p.rows = pl.rows; p.QS = pl.QS;
    p.CP = pl.CP; p.NTILES = pl.NTILES; p.KQ = pl.KQ;
    p.vec4 = (((int64_t)d->K * d->D) % 4 == 0) && ((((uintptr_t)d->g) & 15) == 0);
    p.tptr = d->tile_ptr; p.tpack = d->tile_pack; p.g = d->g;
    p.uid = d->uid; p.uid_stride = d->uid_stride; p.theta = d->theta; p.gh = d->gh;
    p.slab = (float*)d->workspace;
    // one resident round: the plan sizes the grid by LDS alone, the registers (accumulators of all row tiles) usually
    // allow fewer blocks per CU - a smaller grid also means a smaller slab to write and reduce
    int grid = pl.grid;
#define KP_TGM(IT, MTV) do { \
        KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)table_grad_mfma_kernel<IT, MTV>, pl.lds)); \
        const int nb = resident_blocks(table_grad_mfma_kernel<IT, MTV>, kTgmThreads, pl.lds); \
        if (nb > 0 && (int64_t)nb * device_facts().cu_count < grid) grid = nb * device_facts().cu_count; \
        hipLaunchKernelGGL((table_grad_mfma_kernel<IT, MTV>), dim3(grid), dim3(kTgmThreads), pl.lds, s, p); } while (0)
    if (pl.maxit == 1) { if (pl.mtmax == 4) KP_TGM(1, 4); else if (pl.mtmax == 8) KP_TGM(1, 8); else KP_TGM(1, 16); }
    else if (pl.maxit == 2) { if (pl.mtmax == 4) KP_TGM(2, 4); else if (pl.mtmax == 8) KP_TGM(2, 8); else KP_TGM(2, 16); }
    else { if (pl.mtmax == 4) KP_TGM(4, 4); else KP_TGM(4, 8); }
#undef KP_TGM
    KPGNN_LAUNCH_CHECK("table_grad_mfma_kernel");
    *handled = true;
    // ONE finishing launch: this kernel's slabs, a deferred dictionary slab (extra_*) and a pending job of an earlier call
    return slab_reduce(p.slab, grid * pl.KQ, (int64_t)pl.R * p.D, d->gtable0, (int64_t)n0 * p.D, d->gtablek,
                       (int64_t)nk * p.D, d->gdict, s, 0, nullptr, d->extra_slab, d->extra_nslab, d->extra_slab ? d->extra_elems : 0,
                       d->extra_out, nullptr, d->accumulate_dict ? 3 : 0, d->pending);
}

}  // namespace kpgnn
