// PROTOTYPE, not part of the library (round 3): kpgnn_linear_fwd with three-way bf16 splits on the bf16 matrix cores.
// Measured on MI355X at [47450,104] x [104,104]: 29.4 us against 20.6 us for the fp32-MFMA kernel of lin_fused.h; max error against
// float64 1.84e-6 (fp32 MFMA: 2.59e-6, torch fp32: 2.33e-6).  The arithmetic is sound - and 2.7x lighter on the matrix cores - but
// at 185 rows per CU a launch is one tile per block: weight-strip split, tile split + LDS writes and the row loads are all exposed
// latency, the matrix time was never the bound (47 spills at 256 registers did not help).  Kept for the record.
// y = x W^T + b with fp32 operands split three ways into bf16 and multiplied on v_mfma_f32_32x32x16_bf16 (gfx950).
// Contract: include/kpgnn.h, kpgnn_linear_fwd with math = KPGNN_MATH_BF16X3.
//
// The fp32 matrix instructions (v_mfma_f32_32x32x2_f32) run at the VECTOR rate - 1/16 of the bf16 ones - and the chip holds
// only ~1.6 GHz under them: every dense kernel of the KP-GIN+ step sits at 45-85 TFLOP/s whatever its structure (the BLAS
// library's own fp32 GEMM: 76).  An fp32 value is the exact sum of three bf16 pieces (hi + mid + lo = 24 mantissa bits), so
//     a * b = ah*bh + (ah*bm + am*bh) + (am*bm + ah*bl + al*bh) + [terms below 2^-24 of the product: dropped],
// six bf16 products with fp32 accumulation, smallest first: 6/16 of the matrix-core time of the fp32 instruction for an error
// of the order of fp32 rounding itself (the products are exact, the sum is the usual fp32 accumulation).
#include "kpgnn_common.h"

namespace kpgnn {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct Bf3Params {
    int64_t N; const int32_t* n_dyn;
    int O, I, wt;
    const float* x; const float* w; const float* bias; float* y;
};

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}

// IP = I rounded up to 16; M row tiles of 32 per block tile
template <int IP, int M>
__global__ void __launch_bounds__(256, 2)
lin_bf3_kernel(Bf3Params p) {
    p.N = live_rows(p.N, p.n_dyn);
    if (p.N <= 0) return;
    extern __shared__ __attribute__((aligned(16))) uint4 pl[];        // [3][ROWS][PI] 16-byte items (8 bf16 along k)
    constexpr int ROWS = 32 * M, NQ = IP / 8, PI = NQ + 1, KSN = IP / 16;
    const int I = p.I, O = p.O;
    const int CGI = I / 4;                                             // float4 groups per row
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int kg = lane >> 5, c = lane & 31;
    const int o = wave * 32 + c;
    // ---- this wave's strip of W as bf16 A fragments: a*[ks][j] = piece of W[o][16 ks + 8 kg + j]
    bf16x8 ah[KSN], am[KSN], al[KSN];
#pragma unroll
    for (int ks = 0; ks < KSN; ++ks) {
        float wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * ks + 8 * kg + j;
            wv[j] = 0.f;
            if (o < O && k < I) wv[j] = p.wt ? p.w[(int64_t)k * O + o] : p.w[(int64_t)o * I + k];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { __bf16 h, m, l; split3(wv[j], h, m, l); ah[ks][j] = h; am[ks][j] = m; al[ks][j] = l; }
    }
    float4 bias4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int ob = wave * 32 + 8 * g + 4 * kg;
        bias4[g] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias && ob < O) bias4[g] = *reinterpret_cast<const float4*>(p.bias + ob);
    }
    // zero the planes once (the pad columns I..IP and the pad item stay zero)
    for (int i = tid; i < 3 * ROWS * PI; i += 256) pl[i] = make_uint4(0u, 0u, 0u, 0u);
    // ---- tile staging: thread -> float4 slots (row, column group), fixed
    const int slots = ROWS * CGI;
    constexpr int PF = (ROWS * (IP / 4) + 255) / 256;
    int prow[PF], pcg[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) { const int e = tid + 256 * j; prow[j] = e < slots ? e / CGI : -1; pcg[j] = e % CGI; }
    float4 pf[PF];
    const int64_t last = p.N - 1;
    const int64_t tiles = (p.N + ROWS - 1) / ROWS;
    auto issue = [&](int64_t tl) {
        const int64_t r0 = tl * ROWS;
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            int64_t r = r0 + (prow[j] < 0 ? 0 : prow[j]);
            r = r < last ? r : last;
            pf[j] = *reinterpret_cast<const float4*>(p.x + r * I + 4 * pcg[j]);
        }
    };
    uint2* pl8 = reinterpret_cast<uint2*>(pl);                        // 8-byte halves of the items
    auto commit = [&]() {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            if (prow[j] < 0) continue;
            bf16x4 h, m, l;
            { __bf16 a, b, cc; split3(pf[j].x, a, b, cc); h[0] = a; m[0] = b; l[0] = cc; }
            { __bf16 a, b, cc; split3(pf[j].y, a, b, cc); h[1] = a; m[1] = b; l[1] = cc; }
            { __bf16 a, b, cc; split3(pf[j].z, a, b, cc); h[2] = a; m[2] = b; l[2] = cc; }
            { __bf16 a, b, cc; split3(pf[j].w, a, b, cc); h[3] = a; m[3] = b; l[3] = cc; }
            const int half = (prow[j] * PI) * 2 + pcg[j];             // (item = column group / 2, half = column group & 1)
            pl8[half] = __builtin_bit_cast(uint2, h);
            pl8[(ROWS * PI) * 2 + half] = __builtin_bit_cast(uint2, m);
            pl8[2 * (ROWS * PI) * 2 + half] = __builtin_bit_cast(uint2, l);
        }
    };
    int64_t tile = blockIdx.x;
    __syncthreads();
    if (tile < tiles) { issue(tile); commit(); }
    __syncthreads();
    for (; tile < tiles; tile += gridDim.x) {
        const bool more = tile + gridDim.x < tiles;
        issue(more ? tile + gridDim.x : tile);
        f32x16 acc[M];
#pragma unroll
        for (int m = 0; m < M; ++m)
            for (int v = 0; v < 16; ++v) acc[m][v] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks) {
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const int it = (m * 32 + c) * PI + 2 * ks + kg;
                const bf16x8 bh = __builtin_bit_cast(bf16x8, pl[it]);
                const bf16x8 bm = __builtin_bit_cast(bf16x8, pl[ROWS * PI + it]);
                const bf16x8 bl = __builtin_bit_cast(bf16x8, pl[2 * ROWS * PI + it]);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks], bl, acc[m], 0, 0, 0);     // smallest terms first
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks], bh, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[ks], bm, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks], bm, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[ks], bh, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks], bh, acc[m], 0, 0, 0);
            }
        }
        // C/D map: col = lane & 31 (tile row), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (output o): straight to y
        const int64_t r0 = tile * ROWS;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ob = wave * 32 + 8 * g + 4 * kg;
            if (ob < O) {
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    const int64_t r = r0 + m * 32 + c;
                    if (r < p.N)
                        *reinterpret_cast<float4*>(p.y + r * O + ob) = make_float4(acc[m][4 * g] + bias4[g].x, acc[m][4 * g + 1] + bias4[g].y,
                                                                                   acc[m][4 * g + 2] + bias4[g].z, acc[m][4 * g + 3] + bias4[g].w);
                }
            }
        }
        __syncthreads();                               // every wave is done reading the planes
        if (more) commit();
        __syncthreads();
    }
}

}  // namespace

int linear_bf3_launch(const kpgnn_linear_desc* d, hipStream_t s, bool* handled) {
    *handled = false;
    const int I = d->I, O = d->O;
    if (O > 128 || O % 4 || I % 4 || I > 128 || d->x_stride != I || d->y_stride != O || d->x_mask) return KPGNN_OK;
    Bf3Params p;
    p.N = d->N; p.n_dyn = d->n_dyn; p.O = O; p.I = I; p.wt = d->w_transposed ? 1 : 0;
    p.x = d->x; p.w = d->w; p.bias = d->bias; p.y = d->y;
    const int ip = (I + 15) / 16 * 16;
    const int64_t slots = (int64_t)device_facts().cu_count * 2;
    int m = (int)((d->N + slots * 32 - 1) / (slots * 32));
    m = m < 1 ? 1 : (m > 3 ? 3 : m);
    const int rows = 32 * m;
    const size_t lds = (size_t)3 * rows * (ip / 8 + 1) * 16;
    const int64_t tiles = (d->N + rows - 1) / rows;
    const int64_t grid = slots < tiles ? slots : tiles;
#define KP_B3(IPV, MV) do { KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)lin_bf3_kernel<IPV, MV>, lds)); \
        hipLaunchKernelGGL((lin_bf3_kernel<IPV, MV>), dim3((unsigned)grid), dim3(256), lds, s, p); } while (0)
#define KP_B3M(IPV) do { if (m == 1) KP_B3(IPV, 1); else if (m == 2) KP_B3(IPV, 2); else KP_B3(IPV, 3); } while (0)
    switch (ip) {
        case 16: KP_B3M(16); break;
        case 32: KP_B3M(32); break;
        case 48: KP_B3M(48); break;
        case 64: KP_B3M(64); break;
        case 80: KP_B3M(80); break;
        case 96: KP_B3M(96); break;
        case 112: KP_B3M(112); break;
        default: KP_B3M(128); break;
    }
#undef KP_B3M
#undef KP_B3
    KPGNN_LAUNCH_CHECK("lin_bf3_kernel");
    *handled = true;
    return KPGNN_OK;
}

}  // namespace kpgnn
