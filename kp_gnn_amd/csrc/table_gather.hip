// Multi-table gather-sum (fwd) and its table-gradient histogram (bwd) for gfx950.
//   out[m,:] = bias + sum_c table[col_offset[c] + idx[m,c], :]
// Replaces the reference's peripheral feature build: FeatureConcatEncoder = per-column nn.Embedding ->
// concat -> Linear (layers/feature_encoder.py:62-67, called from models/GNNs.py:172-179/:393-400/:637-644)
// and, in backward, 9 sort-based embedding_dense_backward calls over ~N*K*T indices that hit a handful of
// distinct rows.  Here the (tiny) projected tables sit in LDS; a sub-group of G lanes owns one row m, reads
// its C uint16 indices with one coalesced load and sums C LDS rows.  Backward (tgs_bwd_kernel) is a histogram without
// atomics: thread = feature column, a group of threads walks ITS contiguous run of rows in order and adds into a
// group-private LDS copy of the table grads (column-private: no races, fixed order), the groups of a block are added in
// order into a per-block slab row and the slab is reduced in block order: bitwise reproducible (round 1 used ds_add_f32
// + global fp32 atomics, whose order varies between runs).
// Wide tables are split by feature columns across blockIdx.y so that each block's slice fits LDS.
#include "kpgnn_common.h"

namespace kpgnn {
namespace {

constexpr int kBlock = 256;
constexpr size_t kMaxLds = 96 * 1024;

template <int VEC> struct VT;
template <> struct VT<1> { using T = float; };
template <> struct VT<2> { using T = float2; };
template <> struct VT<4> { using T = float4; };

struct TgsParams {
    const int32_t* n_dyn;
    int64_t M;
    int C, D, R, Ds;            // Ds = columns per split
    const uint16_t* idx;
    const int32_t* col_offset;
    const float* table;
    const float* bias;
    float* out; int64_t out_stride;
    const float* gout; int64_t gout_stride;
    float* gtable;
};

template <int VEC, int G>
__global__ void __launch_bounds__(kBlock)
tgs_kernel(TgsParams p) {
    p.M = live_rows(p.M, p.n_dyn);
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [R, Ds]
    const int Ds = p.Ds, R = p.R;
    const int col_base = blockIdx.y * Ds;
    for (int t = threadIdx.x; t < R * Ds; t += kBlock) {
        const int r = t / Ds, c = t - r * Ds;
        lds[t] = p.table[(int64_t)r * p.D + col_base + c];
    }
    __syncthreads();
    constexpr int ROWS = kBlock / G;
    const int sg = threadIdx.x / G, sl = threadIdx.x % G;
    const int lane = threadIdx.x & (kWave - 1);
    const int sg_lane0 = lane - sl;
    const int c0 = sl * VEC;
    const bool col_ok = c0 < Ds;
    using T = typename VT<VEC>::T;
    const int64_t tiles = (p.M + ROWS - 1) / ROWS;
    // the first G indices of a row travel one trip ahead (the idx load -> shuffle -> LDS reads -> store chain of a trip
    // otherwise starts with a global round trip: 344 us for the dense [N*K, 13] peripheral indices, latency-bound)
    auto load_rows = [&](int64_t tile, int cb) -> int {
        const int64_t m = tile * ROWS + sg;
        return (tile < tiles && m < p.M && cb + sl < p.C) ? p.col_offset[cb + sl] + (int)p.idx[m * p.C + cb + sl] : 0;
    };
    int next_rows = load_rows(blockIdx.x, 0);
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t m = tile * ROWS + sg;
        int myrow = next_rows;
        next_rows = load_rows(tile + gridDim.x, 0);
        if (m >= p.M) continue;
        float acc[VEC];
        for (int q = 0; q < VEC; ++q) acc[q] = 0.f;
        if (col_ok && p.bias) {
            T b = *reinterpret_cast<const T*>(p.bias + col_base + c0);
            for (int q = 0; q < VEC; ++q) acc[q] = reinterpret_cast<const float*>(&b)[q];
        }
        for (int cb = 0; cb < p.C; cb += G) {
            if (cb > 0) myrow = load_rows(tile, cb);
            const int cnt = min(G, p.C - cb);
            for (int t = 0; t < cnt; ++t) {
                const int row = __shfl(myrow, sg_lane0 + t);
                if (!col_ok) continue;
                T v = *reinterpret_cast<const T*>(lds + row * Ds + c0);
                for (int q = 0; q < VEC; ++q) acc[q] += reinterpret_cast<const float*>(&v)[q];
            }
        }
        if (col_ok) {
            T o;
            for (int q = 0; q < VEC; ++q) reinterpret_cast<float*>(&o)[q] = acc[q];
            *reinterpret_cast<T*>(p.out + m * p.out_stride + col_base + c0) = o;
        }
    }
}

// ---- few rows (M <= kSmallRows, M * C <= kSmallPairs): the peripheral DICTIONARY of a batch has ~25 distinct (node, hop) feature tuples,
// whatever the batch size.  The kernels above then run as ONE block that first copies a ~250 KB table into LDS (forward,
// 15 us) or walks 25 x 13 LDS float adds as one serial chain (backward, 32 us + the slab reduce).  Here:
//   forward  one block per row m; a thread owns a column and sums its C table rows straight from L2 (independent loads);
//   backward "pull": one block per TABLE row r scans the M * C (row, component) pairs 64 at a time (ballot), and adds
//            gout[m, :] for the pairs that address r, in (m, c) order - every table row is written once, by one block:
//            no accumulator table, no slab, no second launch, bitwise reproducible.
constexpr int kSmallPairs = 4096;
constexpr int kSmallBlock = 128;

__global__ void __launch_bounds__(kSmallBlock)
tgs_small_fwd_kernel(TgsParams p) {
    p.M = live_rows(p.M, p.n_dyn);
    const int64_t m = blockIdx.x;
    const uint16_t* ix = p.idx + m * p.C;
    for (int col = threadIdx.x; col < p.D; col += kSmallBlock) {
        float acc = p.bias ? p.bias[col] : 0.f;
        int c = 0;
        for (; c + 3 < p.C; c += 4) {        // four rows in flight; added in component order
            const float v0 = p.table[(int64_t)(p.col_offset[c] + (int)ix[c]) * p.D + col];
            const float v1 = p.table[(int64_t)(p.col_offset[c + 1] + (int)ix[c + 1]) * p.D + col];
            const float v2 = p.table[(int64_t)(p.col_offset[c + 2] + (int)ix[c + 2]) * p.D + col];
            const float v3 = p.table[(int64_t)(p.col_offset[c + 3] + (int)ix[c + 3]) * p.D + col];
            acc += v0; acc += v1; acc += v2; acc += v3;
        }
        for (; c < p.C; ++c) acc += p.table[(int64_t)(p.col_offset[c] + (int)ix[c]) * p.D + col];
        p.out[m * p.out_stride + col] = acc;
    }
}

__global__ void __launch_bounds__(kSmallBlock)
tgs_small_bwd_kernel(TgsParams p) {
    p.M = live_rows(p.M, p.n_dyn);
    const int r = blockIdx.x;
    const int lane = threadIdx.x & (kWave - 1);
    const int pairs = (int)p.M * p.C;
    float acc[2] = {0.f, 0.f};               // columns tid and tid + 128 (D <= 256)
    for (int p0 = 0; p0 < pairs; p0 += kWave) {
        const int q = p0 + lane;
        bool hit = false;
        if (q < pairs) hit = p.col_offset[q % p.C] + (int)p.idx[q] == r;
        unsigned long long mask = __ballot(hit);                // (every wave of the block computes the same mask)
        while (mask) {
            const int j = (int)__builtin_ctzll(mask);
            mask &= mask - 1;
            const int64_t m = (p0 + j) / p.C;
            const float* g = p.gout + m * p.gout_stride;
            if ((int)threadIdx.x < p.D) acc[0] += g[threadIdx.x];
            if ((int)threadIdx.x + kSmallBlock < p.D) acc[1] += g[threadIdx.x + kSmallBlock];
        }
    }
    if ((int)threadIdx.x < p.D) p.gtable[(int64_t)r * p.D + threadIdx.x] = acc[0];
    if ((int)threadIdx.x + kSmallBlock < p.D) p.gtable[(int64_t)r * p.D + threadIdx.x + kSmallBlock] = acc[1];
}

// (few ROWS, not just few pairs: a table row that most of 1,480 single-component rows address - the carbon row of a node-feature
//  table - makes the pull kernel's block a chain of a thousand dependent loads: 51 us against the histogram kernel's 16)
constexpr int kSmallRows = 256;
bool tgs_small(const kpgnn_tgs_desc* d) {
    return d->M <= kSmallRows && d->M * (int64_t)d->C <= kSmallPairs && d->D <= 2 * kSmallBlock;
}

TgsParams tgs_params(const kpgnn_tgs_desc* d) {
    TgsParams p;
    p.M = d->M; p.n_dyn = d->n_dyn; p.C = d->C; p.D = d->D; p.R = d->R; p.Ds = d->D; p.idx = d->idx; p.col_offset = d->col_offset;
    p.table = d->table; p.bias = d->bias; p.out = d->out; p.out_stride = d->out_stride;
    p.gout = d->gout; p.gout_stride = d->gout_stride; p.gtable = d->gtable;
    return p;
}

// Backward: gtable[col_offset[c] + idx[m,c], :] += gout[m, :].  Block = NG groups of CW threads (CW = pow2 >= Ds); the
// block's contiguous run of rows is cut into NG contiguous sub-runs, group g adds ITS rows in order into its private
// accumulator table [R][CW] in LDS; then the groups are added in order and leave as slab row blockIdx.x.
// slab: [gridDim.x][R][D].  Each accumulator cell belongs to ONE thread (thread = column), so there are no races and
// the order of the additions is the program order of that thread.  They are issued as LDS atomics WITHOUT return value
// (ds_add_f32) all the same: a plain `+=` is a load-add-store chain that waits for the LDS latency at every index, while
// the fire-and-forget form keeps the LDS pipeline full and stays ordered per address within a wave.
__global__ void __launch_bounds__(kBlock)
tgs_bwd_kernel(TgsParams p, int CW, int NG, float* __restrict__ slab, int csplit) {
    p.M = live_rows(p.M, p.n_dyn);
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [NG][R][CW]
    const int Ds = p.Ds, R = p.R;
    const int col_base = blockIdx.y * Ds;
    for (int t = threadIdx.x; t < NG * R * CW; t += kBlock) lds[t] = 0.f;
    __syncthreads();
    // csplit (one table per block, CW == 64): ALL waves of the block walk the block's rows, each for ITS components - the
    // components of different waves address disjoint table rows (components that share a table, i.e. an offset, stay
    // together), so the waves share the one accumulator table without ever touching the same cell: four waves instead of
    // one on the scalar chain that bounds this kernel, same fixed order of every sum.
    const int grp = csplit ? 0 : threadIdx.x / CW, col = threadIdx.x % CW;
    const int wv = threadIdx.x / kWave, nwv = kBlock / kWave;
    const int64_t per_block = (p.M + gridDim.x - 1) / gridDim.x;
    const int64_t b0 = (int64_t)blockIdx.x * per_block;
    const int64_t b1 = b0 + per_block < p.M ? b0 + per_block : p.M;
    if (grp < NG && b0 < b1) {
        const int64_t per_grp = (b1 - b0 + NG - 1) / NG;
        const int64_t m0 = b0 + grp * per_grp;
        const int64_t m1 = m0 + per_grp < b1 ? m0 + per_grp : b1;
        float* acc = lds + (int64_t)grp * R * CW + col;
        const bool col_ok = col < Ds;
        if (CW % kWave == 0 && p.C <= kWave) {
            // A group is made of whole waves: the indices of up to four rows arrive with ONE coalesced load per wave
            // (lane j holds idx[m*C + j]) a trip ahead and are broadcast with v_readlane.  (One dependent global load per
            // (row, component) in the loop cost 3.0 ms per launch on the dense [N*K, 13] peripheral indices.)
            const int lane = threadIdx.x & (kWave - 1);
            const int C = p.C, RT = min(4, kWave / C);
            const int offv = lane < C ? p.col_offset[lane] : 0;
            unsigned long long mycomps = ~0ull;                  // components this wave handles
            if (csplit) {
                // distinct tables (offsets) are dealt to the waves longest-first onto the least loaded wave: a table used by m
                // components costs m items per row
                bool first = true;                               // no earlier component has my offset
                int mult = 0;                                    // components with my offset
                for (int j = 0; j < C; ++j) {
                    const int oj = __builtin_amdgcn_readlane(offv, j);
                    if (oj == offv) { ++mult; if (j < lane) first = false; }
                }
                unsigned long long todo = __ballot(lane < C && first);
                int wload[4] = {0, 0, 0, 0};
                int mywave = -1;
                while (todo) {
                    int best = -1, bm = -1;
                    for (unsigned long long t = todo; t; t &= t - 1) {
                        const int j = (int)__builtin_ctzll(t);
                        const int m = __builtin_amdgcn_readlane(mult, j);
                        if (m > bm) { bm = m; best = j; }
                    }
                    int wmin = 0;
                    for (int q = 1; q < nwv && q < 4; ++q) if (wload[q] < wload[wmin]) wmin = q;
                    wload[wmin] += bm;
                    if (offv == __builtin_amdgcn_readlane(offv, best)) mywave = wmin;
                    todo &= ~(1ull << best);
                }
                mycomps = __ballot(lane < C && mywave == wv);
            }
            auto load_idx = [&](int64_t mm) -> int {
                const int64_t n = (m1 - mm < RT ? m1 - mm : RT) * C;
                return (mm < m1 && lane < n) ? (int)p.idx[mm * C + lane] : 0;
            };
            auto load_g = [&](int64_t mm, float* o) {
#pragma unroll
                for (int u = 0; u < 4; ++u) o[u] = (u < RT && mm + u < m1 && col_ok) ? p.gout[(mm + u) * p.gout_stride + col_base + col] : 0.f;
            };
            constexpr int kRunC = 16;
            int cur[kRunC]; float run[kRunC];
#pragma unroll
            for (int c = 0; c < kRunC; ++c) { cur[c] = -1; run[c] = 0.f; }
            // Four batches of RT rows are in flight: with ONE wave per group and one group per block (the accumulator table
            // fills the LDS) nothing else hides the ~2 us of a global round trip - a batch requested one batch ahead made
            // every batch wait for it (1.7 ms per launch on the dense peripheral tensors).
            constexpr int PD = 4;
            int iq[PD]; float gq[PD][4];
#pragma unroll
            for (int b = 0; b < PD; ++b) { iq[b] = load_idx(m0 + (int64_t)b * RT); load_g(m0 + (int64_t)b * RT, gq[b]); }
            for (int64_t mb = m0; mb < m1; mb += (int64_t)PD * RT) {
#pragma unroll
                for (int b = 0; b < PD; ++b) {
                    const int64_t m = mb + (int64_t)b * RT;
                    const int idxv = iq[b];
                    float g[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) g[u] = gq[b][u];
                    iq[b] = load_idx(m + (int64_t)PD * RT);
                    load_g(m + (int64_t)PD * RT, gq[b]);
                    if (m >= m1) continue;
                    const int nrows = (int)(m1 - m < RT ? m1 - m : RT);
                    if (C <= kRunC) {
                        // Consecutive rows mostly carry the SAME index in a component (94 % on the peripheral tensors of a
                        // molecule batch: most slots are the "none" code), and an LDS float add costs ~150 cycles of the CU's
                        // LDS pipe: per component the run of equal rows is summed in a register and added once when the
                        // index changes.  cur[] are wave-uniform (scalar registers), the component loop is unrolled.
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (u < nrows) {
#pragma unroll
                                for (int c = 0; c < kRunC; ++c) {
                                    if (c < C && ((mycomps >> c) & 1ull)) {
                                        const int row = __builtin_amdgcn_readlane(offv, c) + __builtin_amdgcn_readlane(idxv, u * C + c);
                                        if (row != cur[c]) {
                                            if (cur[c] >= 0 && col_ok) atomicAdd(acc + cur[c] * CW, run[c]);
                                            cur[c] = row;
                                            run[c] = 0.f;
                                        }
                                        run[c] += g[u];
                                    }
                                }
                            }
                        }
                    } else
                    for (int c = 0; c < C; ++c) {
                        const int off = __builtin_amdgcn_readlane(offv, c);
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (u < nrows) {
                                const int row = off + __builtin_amdgcn_readlane(idxv, u * C + c);
                                if (col_ok) atomicAdd(acc + row * CW, g[u]);   // ds_add_f32, no return: see the header
                            }
                    }
                }
            }
            if (C <= kRunC) {
#pragma unroll
                for (int c = 0; c < kRunC; ++c)
                    if (c < C && ((mycomps >> c) & 1ull) && cur[c] >= 0 && col_ok) atomicAdd(acc + cur[c] * CW, run[c]);
            }
        } else if (col_ok) {
            // four rows per trip: their loads are independent of the LDS adds of the previous rows
            for (int64_t m = m0; m < m1; m += 4) {
                float g[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) g[u] = m + u < m1 ? p.gout[(m + u) * p.gout_stride + col_base + col] : 0.f;
                for (int c = 0; c < p.C; ++c) {
                    const int off = p.col_offset[c];
                    int row[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) row[u] = m + u < m1 ? off + (int)p.idx[(m + u) * p.C + c] : -1;
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (row[u] >= 0) atomicAdd(acc + row[u] * CW, g[u]);   // ds_add_f32, no return: see the header
                }
            }
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < R * Ds; t += kBlock) {
        const int r = t / Ds, c = t - r * Ds;
        float v = 0.f;
        for (int g = 0; g < NG; ++g) v += lds[((int64_t)g * R + r) * CW + c];
        slab[((int64_t)blockIdx.x * R + r) * p.D + col_base + c] = v;
    }
}

template <int VEC, int G>
int launch(const TgsParams& p, int splits, size_t lds, hipStream_t s) {
    const int64_t tiles = (p.M + (kBlock / G) - 1) / (kBlock / G);
    int64_t gx = (int64_t)device_facts().cu_count * (lds > 40 * 1024 ? 1 : 3);
    if (gx > tiles) gx = tiles;
    if (gx < 1) gx = 1;
    dim3 grid((unsigned)gx, (unsigned)splits);
    if (lds > 64 * 1024)
        KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)tgs_kernel<VEC, G>, lds));
    hipLaunchKernelGGL((tgs_kernel<VEC, G>), grid, dim3(kBlock), lds, s, p);
    KPGNN_LAUNCH_CHECK("tgs_kernel");
    return KPGNN_OK;
}

// Backward plan: column splits so that at least one [R][CW] accumulator table fits LDS, groups per block, grid.
struct BwdPlan { int splits, Ds, CW, NG, gx; size_t lds, ws_bytes; };
constexpr size_t kMaxLdsBwd = 144 * 1024;

bool bwd_plan(int64_t M, int D, int R, BwdPlan* pl) {
    for (int sct = 1; sct <= D; ++sct) {
        if (D % sct) continue;
        const int w = D / sct;
        if (w > kBlock) continue;
        int cw = 1;
        while (cw < w) cw <<= 1;
        const size_t one = sizeof(float) * (size_t)R * cw;
        if (one > kMaxLdsBwd) continue;
        int ng = kBlock / cw;
        while (ng > 1 && one * ng > kMaxLdsBwd) ng >>= 1;
        pl->splits = sct; pl->Ds = w; pl->CW = cw; pl->NG = ng; pl->lds = one * ng;
        // rows per group >= 16 where possible; at most one block per CU and a slab of <= 32 MB
        int64_t gx = (M + 16 * ng - 1) / (16 * ng);
        const int64_t cap_cu = device_facts().cu_count;
        const int64_t cap_ws = (int64_t)(32u << 20) / ((int64_t)sizeof(float) * R * D);
        if (gx > cap_cu) gx = cap_cu;
        if (gx > cap_ws) gx = cap_ws;
        if (gx < 1) gx = 1;
        pl->gx = (int)gx;
        pl->ws_bytes = sizeof(float) * (size_t)gx * R * D;
        return true;
    }
    return false;
}

int run_bwd(const kpgnn_tgs_desc* d, hipStream_t s) {
    KPGNN_REQUIRE(d != nullptr, "table_gather_sum_bwd: NULL descriptor");
    KPGNN_REQUIRE(d->M >= 0 && d->C >= 1 && d->D >= 1 && d->R >= 1, "table_gather_sum_bwd: bad M=%lld C=%d D=%d R=%d",
                  (long long)d->M, d->C, d->D, d->R);
    KPGNN_REQUIRE(d->gtable != nullptr, "table_gather_sum_bwd: NULL gtable");
    if (d->M == 0) { KPGNN_HIP_TRY(hipMemsetAsync(d->gtable, 0, sizeof(float) * (size_t)d->R * d->D, s)); return KPGNN_OK; }
    KPGNN_REQUIRE(d->idx && d->col_offset && d->gout && d->gout_stride >= d->D, "table_gather_sum_bwd: NULL idx/col_offset/gout");
    if (tgs_small(d)) {
        hipLaunchKernelGGL(tgs_small_bwd_kernel, dim3((unsigned)d->R), dim3(kSmallBlock), 0, s, tgs_params(d));
        KPGNN_LAUNCH_CHECK("tgs_small_bwd_kernel");
        return KPGNN_OK;
    }
    BwdPlan pl;
    if (!bwd_plan(d->M, d->D, d->R, &pl))
        return fail(KPGNN_ELIMIT, "table_gather_sum_bwd: R=%d rows do not fit LDS at any column split of D=%d", d->R, d->D);
    KPGNN_REQUIRE(d->workspace && d->workspace_bytes >= pl.ws_bytes, "table_gather_sum_bwd: workspace too small (%zu < %zu)",
                  (size_t)d->workspace_bytes, pl.ws_bytes);
    TgsParams p;
    p.M = d->M; p.n_dyn = d->n_dyn; p.C = d->C; p.D = d->D; p.R = d->R; p.Ds = pl.Ds; p.idx = d->idx; p.col_offset = d->col_offset;
    p.table = nullptr; p.bias = nullptr; p.out = nullptr; p.out_stride = 0;
    p.gout = d->gout; p.gout_stride = d->gout_stride; p.gtable = d->gtable;
    KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)tgs_bwd_kernel, pl.lds));
    const int csplit = (pl.NG == 1 && pl.CW == kWave && d->C <= 16 && d->C >= 2) ? 1 : 0;
    hipLaunchKernelGGL(tgs_bwd_kernel, dim3(pl.gx, pl.splits), dim3(kBlock), pl.lds, s, p, pl.CW, pl.NG, (float*)d->workspace, csplit);
    KPGNN_LAUNCH_CHECK("tgs_bwd_kernel");
    return slab_reduce((const float*)d->workspace, pl.gx, (int64_t)d->R * d->D, d->gtable, (int64_t)d->R * d->D, nullptr, 0, nullptr, s);
}

int run(const kpgnn_tgs_desc* d, hipStream_t s) {
    const bool bwd = false;
    KPGNN_REQUIRE(d != nullptr, "table_gather_sum: NULL descriptor");
    KPGNN_REQUIRE(d->M >= 0 && d->C >= 1 && d->D >= 1 && d->R >= 1, "table_gather_sum: bad M=%lld C=%d D=%d R=%d",
                  (long long)d->M, d->C, d->D, d->R);
    if (d->M == 0) return KPGNN_OK;
    KPGNN_REQUIRE(d->idx && d->col_offset, "table_gather_sum: NULL idx/col_offset");
    if (bwd) KPGNN_REQUIRE(d->gout && d->gtable && d->gout_stride >= d->D, "table_gather_sum_bwd: NULL gout/gtable");
    else KPGNN_REQUIRE(d->table && d->out && d->out_stride >= d->D, "table_gather_sum_fwd: NULL table/out");
    if (!bwd && tgs_small(d)) {
        hipLaunchKernelGGL(tgs_small_fwd_kernel, dim3((unsigned)d->M), dim3(kSmallBlock), 0, s, tgs_params(d));
        KPGNN_LAUNCH_CHECK("tgs_small_fwd_kernel");
        return KPGNN_OK;
    }
    // column splits: the fewest such that the LDS slice fits and the slice width stays VEC-aligned
    const float* data = bwd ? d->gout : d->out;
    const int64_t stride = bwd ? d->gout_stride : d->out_stride;
    int splits = 0, Ds = 0, vec = 1;
    for (int sct = 1; sct <= d->D; ++sct) {
        if (d->D % sct) continue;
        const int w = d->D / sct;
        if ((size_t)d->R * w * sizeof(float) > kMaxLds) continue;
        int v = (w % 4 == 0) ? 4 : (w % 2 == 0 ? 2 : 1);
        while (v > 1 && (((uintptr_t)data % (v * 4)) || (stride % v) ||
                         (!bwd && d->bias && ((uintptr_t)d->bias % (v * 4)))))
            v >>= 1;
        if ((w + v - 1) / v > 64) continue;
        splits = sct; Ds = w; vec = v;
        break;
    }
    if (!splits) return fail(KPGNN_ELIMIT, "table_gather_sum: R=%d rows do not fit LDS at any column split of D=%d", d->R, d->D);
    TgsParams p;
    p.M = d->M; p.n_dyn = d->n_dyn; p.C = d->C; p.D = d->D; p.R = d->R; p.Ds = Ds; p.idx = d->idx; p.col_offset = d->col_offset;
    p.table = d->table; p.bias = d->bias; p.out = d->out; p.out_stride = d->out_stride;
    p.gout = d->gout; p.gout_stride = d->gout_stride; p.gtable = d->gtable;
    const size_t lds = (size_t)d->R * Ds * sizeof(float);
    int g = 4;
    while (g * vec < Ds) g <<= 1;
#define KP_TGS(V, GG) launch<V, GG>(p, splits, lds, s)
    switch (vec * 100 + g) {
        case 404: return KP_TGS(4, 4); case 408: return KP_TGS(4, 8); case 416: return KP_TGS(4, 16);
        case 432: return KP_TGS(4, 32); case 464: return KP_TGS(4, 64);
        case 204: return KP_TGS(2, 4); case 208: return KP_TGS(2, 8); case 216: return KP_TGS(2, 16);
        case 232: return KP_TGS(2, 32); case 264: return KP_TGS(2, 64);
        case 104: return KP_TGS(1, 4); case 108: return KP_TGS(1, 8); case 116: return KP_TGS(1, 16);
        case 132: return KP_TGS(1, 32); case 164: return KP_TGS(1, 64);
    }
#undef KP_TGS
    return fail(KPGNN_EINVAL, "table_gather_sum: no kernel for vec=%d g=%d", vec, g);
}

}  // namespace
}  // namespace kpgnn

extern "C" int kpgnn_table_gather_sum_fwd(const kpgnn_tgs_desc* d, kpgnn_stream_t stream) {
    return kpgnn::run(d, (hipStream_t)stream);
}

extern "C" int kpgnn_table_gather_sum_bwd(const kpgnn_tgs_desc* d, kpgnn_stream_t stream) {
    return kpgnn::run_bwd(d, (hipStream_t)stream);
}

extern "C" size_t kpgnn_table_gather_sum_bwd_workspace_bytes(int64_t M, int32_t D, int32_t R) {
    kpgnn::BwdPlan pl;
    if (M < 1 || D < 1 || R < 1 || !kpgnn::bwd_plan(M, D, R, &pl)) return 0;
    return pl.ws_bytes;
}
