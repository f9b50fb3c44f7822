// Multi-table gather-sum (fwd) and its table-gradient histogram (bwd) for gfx950.
//   out[m,:] = bias + sum_c table[col_offset[c] + idx[m,c], :]
// Replaces the reference's peripheral feature build: FeatureConcatEncoder = per-column nn.Embedding ->
// concat -> Linear (layers/feature_encoder.py:62-67, called from models/GNNs.py:172-179/:393-400/:637-644)
// and, in backward, 9 sort-based embedding_dense_backward calls over ~N*K*T indices that hit a handful of
// distinct rows.  Here the (tiny) projected tables sit in LDS; a sub-group of G lanes owns one row m, reads
// its C uint16 indices with one coalesced load and sums C LDS rows; backward accumulates gout rows into an
// LDS copy of the table grads with ds_add_f32 and flushes once per block with global fp32 atomics.
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

template <int VEC, int G, bool BWD>
__global__ void __launch_bounds__(kBlock)
tgs_kernel(const TgsParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [R, Ds]
    const int Ds = p.Ds, R = p.R;
    const int col_base = blockIdx.y * Ds;
    for (int t = threadIdx.x; t < R * Ds; t += kBlock) {
        const int r = t / Ds, c = t - r * Ds;
        lds[t] = BWD ? 0.f : p.table[(int64_t)r * p.D + col_base + c];
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
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t m = tile * ROWS + sg;
        if (m >= p.M) continue;
        float acc[VEC];
        if (BWD) {
            if (col_ok) {
                T g = *reinterpret_cast<const T*>(p.gout + m * p.gout_stride + col_base + c0);
                for (int q = 0; q < VEC; ++q) acc[q] = reinterpret_cast<const float*>(&g)[q];
            }
        } else {
            for (int q = 0; q < VEC; ++q) acc[q] = 0.f;
            if (col_ok && p.bias) {
                T b = *reinterpret_cast<const T*>(p.bias + col_base + c0);
                for (int q = 0; q < VEC; ++q) acc[q] = reinterpret_cast<const float*>(&b)[q];
            }
        }
        for (int cb = 0; cb < p.C; cb += G) {
            int myrow = 0;
            if (cb + sl < p.C) myrow = p.col_offset[cb + sl] + (int)p.idx[m * p.C + cb + sl];
            const int cnt = min(G, p.C - cb);
            for (int t = 0; t < cnt; ++t) {
                const int row = __shfl(myrow, sg_lane0 + t);
                if (!col_ok) continue;
                float* lrow = lds + row * Ds + c0;
                if (BWD) {
                    for (int q = 0; q < VEC; ++q) atomicAdd(lrow + q, acc[q]);
                } else {
                    T v = *reinterpret_cast<const T*>(lrow);
                    for (int q = 0; q < VEC; ++q) acc[q] += reinterpret_cast<const float*>(&v)[q];
                }
            }
        }
        if (!BWD && col_ok) {
            T o;
            for (int q = 0; q < VEC; ++q) reinterpret_cast<float*>(&o)[q] = acc[q];
            *reinterpret_cast<T*>(p.out + m * p.out_stride + col_base + c0) = o;
        }
    }
    if (BWD) {
        __syncthreads();
        for (int t = threadIdx.x; t < R * Ds; t += kBlock) {
            const float v = lds[t];
            if (v != 0.f) {
                const int r = t / Ds, c = t - r * Ds;
                atomicAdd(p.gtable + (int64_t)r * p.D + col_base + c, v);
            }
        }
    }
}

template <int VEC, int G>
int launch(const TgsParams& p, bool bwd, int splits, size_t lds, hipStream_t s) {
    const int64_t tiles = (p.M + (kBlock / G) - 1) / (kBlock / G);
    int64_t gx = (int64_t)device_facts().cu_count * (lds > 40 * 1024 ? 1 : 3);
    if (gx > tiles) gx = tiles;
    if (gx < 1) gx = 1;
    dim3 grid((unsigned)gx, (unsigned)splits);
    if (bwd) {
        if (lds > 64 * 1024)
            KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)tgs_kernel<VEC, G, true>, lds));
        hipLaunchKernelGGL((tgs_kernel<VEC, G, true>), grid, dim3(kBlock), lds, s, p);
    } else {
        if (lds > 64 * 1024)
            KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)tgs_kernel<VEC, G, false>, lds));
        hipLaunchKernelGGL((tgs_kernel<VEC, G, false>), grid, dim3(kBlock), lds, s, p);
    }
    KPGNN_LAUNCH_CHECK("tgs_kernel");
    return KPGNN_OK;
}

int run(const kpgnn_tgs_desc* d, bool bwd, hipStream_t s) {
    KPGNN_REQUIRE(d != nullptr, "table_gather_sum: NULL descriptor");
    KPGNN_REQUIRE(d->M >= 0 && d->C >= 1 && d->D >= 1 && d->R >= 1, "table_gather_sum: bad M=%lld C=%d D=%d R=%d",
                  (long long)d->M, d->C, d->D, d->R);
    if (d->M == 0) return KPGNN_OK;
    KPGNN_REQUIRE(d->idx && d->col_offset, "table_gather_sum: NULL idx/col_offset");
    if (bwd) KPGNN_REQUIRE(d->gout && d->gtable && d->gout_stride >= d->D, "table_gather_sum_bwd: NULL gout/gtable");
    else KPGNN_REQUIRE(d->table && d->out && d->out_stride >= d->D, "table_gather_sum_fwd: NULL table/out");
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
    p.M = d->M; p.C = d->C; p.D = d->D; p.R = d->R; p.Ds = Ds; p.idx = d->idx; p.col_offset = d->col_offset;
    p.table = d->table; p.bias = d->bias; p.out = d->out; p.out_stride = d->out_stride;
    p.gout = d->gout; p.gout_stride = d->gout_stride; p.gtable = d->gtable;
    const size_t lds = (size_t)d->R * Ds * sizeof(float);
    int g = 4;
    while (g * vec < Ds) g <<= 1;
#define KP_TGS(V, GG) launch<V, GG>(p, bwd, splits, lds, s)
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
    return kpgnn::run(d, false, (hipStream_t)stream);
}

extern "C" int kpgnn_table_gather_sum_bwd(const kpgnn_tgs_desc* d, kpgnn_stream_t stream) {
    return kpgnn::run(d, true, (hipStream_t)stream);
}
