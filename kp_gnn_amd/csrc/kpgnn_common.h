// Shared helpers for libkpgnn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "kpgnn.h"

namespace kpgnn {

char* error_buffer();  // thread-local, 512 bytes (defined in csr_build.hip)

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define KPGNN_HIP_TRY(expr)                                                                      \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return ::kpgnn::fail(KPGNN_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                 __FILE__, __LINE__);                                            \
    } while (0)

#define KPGNN_LAUNCH_CHECK(name)                                                                          \
    do {                                                                                                  \
        hipError_t e_ = hipGetLastError();                                                                \
        if (e_ != hipSuccess)                                                                             \
            return ::kpgnn::fail(KPGNN_EHIP, "launch of %s failed: %s", name, hipGetErrorString(e_));     \
    } while (0)

#define KPGNN_REQUIRE(cond, ...)                                  \
    do {                                                          \
        if (!(cond)) return ::kpgnn::fail(KPGNN_EINVAL, __VA_ARGS__); \
    } while (0)

constexpr int kWave = 64;   // CDNA wavefront

// Live row count of a launch whose descriptor carries a CAPACITY N and an optional device-side count (kpgnn.h, `n_dyn`): a
// hipGraph captured once for the capacity then serves batches of any size up to it.  One scalar load at kernel entry.
template <typename T>
__device__ __forceinline__ T live_rows(T N, const int32_t* n_dyn) {
    if (!n_dyn) return N;
    const T n = (T)*n_dyn;
    return n < N ? (n < 0 ? (T)0 : n) : N;
}
constexpr int kNumXcd = 8;  // MI355X: 8 XCDs, blocks are dealt round-robin over them

struct DeviceFacts {
    int cu_count = 256;
    int lds_per_block = 160 * 1024;
    bool valid = false;
};
const DeviceFacts& device_facts();

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a slow host call: do it once per kernel, raising only.
hipError_t ensure_dynamic_lds(const void* func, size_t bytes);
// linear_bf3.hip: the bf16-split kernels behind kpgnn_linear_group_fwd / kpgnn_linear_fwd (blocked output); *handled = false
// when the shape (or d->math) leaves the launch to the fp32 kernels
size_t linear3_workspace_bytes(int O, int I, int group);
int linear3_group_fwd(const kpgnn_linear_group_desc* d, hipStream_t s, bool* handled);
int linear3_blocked(const kpgnn_linear_desc* d, hipStream_t s, bool* handled);
// the split copy of ONE weight matrix (element (k, n) at w[n * wn + k * wk]) in B-fragment order, for linear3_fused
int linear3_split_w(const float* w, int64_t wn, int64_t wk, int O, int I, void* frag, hipStream_t s);

// out_j[e] = sum_b slab[b][e] in block order (deterministic); the `elems` outputs are split over up to three
// destination arrays of n0 / n1 / rest elements (table_grad.hip).
// Optionally the same launch reduces a second, independent slab [nslab_b][elems_b] into out_b, and (tf) finishes a
// theta-gradient slab: gtheta[k,d] = sum_b slab[b][k,d] in block order and galpha = d(theta)/d(alpha)^T gtheta (galpha may
// be NULL: gtheta only).  One launch for everything a table-gradient call leaves behind.
struct ThetaFinish {
    const float* slab; int nslab;
    const float* alpha; const float* theta; int K, D;
    float* gtheta; float* galpha;
};
int slab_reduce(const float* slab, int nslab, int64_t elems, float* out0, int64_t n0, float* out1, int64_t n1,
                float* out2, hipStream_t s, int64_t n2 = 0, float* out3 = nullptr, const float* slab_b = nullptr,
                int nslab_b = 0, int64_t elems_b = 0, float* out_b = nullptr, const ThetaFinish* tf = nullptr,
                int acc_mask = 0,    // bit 0: out2 += , bit 1: out_b +=  (a gradient collected over several calls)
                const kpgnn_reduce_job* pending = nullptr);   // a third, caller-described job (kpgnn.h)

// A 1024-thread block's share of the theta finishing (block `blk` owns CB = 64 / K columns); sm: >= 1168 floats of LDS.
__device__ __forceinline__ void theta_finish_block(const ThetaFinish& f, int blk, float* sm) {
    float (*part)[65] = reinterpret_cast<float (*)[65]>(sm);      // [16][65]
    float* tot = sm + 16 * 65;
    float* ths = tot + 64;
    const int K = f.K, D = f.D, CB = 64 / K;
    const int o = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int k = o / CB, cq = o - k * CB;
    const int d = blk * CB + cq;
    const bool ok = k < K && cq < CB && d < D;
    const int64_t KD = (int64_t)K * D, e = (int64_t)k * D + d;
    // theta / alpha of this block's columns are requested before the slab loop (their latency hides behind it)
    float th_mine = 0.f, al_mine = 0.f;
    if (slice == 0 && ok) th_mine = f.theta[e];
    const bool colthread = threadIdx.x < CB && blk * CB + threadIdx.x < D;
    if (colthread && f.galpha) al_mine = f.alpha[blk * CB + threadIdx.x];
    float s = 0.f;
    if (ok) {
        int b = slice;
        for (; b + 48 < f.nslab; b += 64) {
            const float v0 = f.slab[(int64_t)b * KD + e], v1 = f.slab[(int64_t)(b + 16) * KD + e];
            const float v2 = f.slab[(int64_t)(b + 32) * KD + e], v3 = f.slab[(int64_t)(b + 48) * KD + e];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; b < f.nslab; b += 16) s += f.slab[(int64_t)b * KD + e];
    }
    part[slice][o] = s;
    __syncthreads();
    if (slice == 0) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += part[q][o];
        tot[o] = t;
        ths[o] = th_mine;
        if (ok) f.gtheta[e] = t;
    }
    __syncthreads();
    if (colthread && f.galpha) {
        const int c = threadIdx.x, dd = blk * CB + c;
        const float a = 1.0f / (1.0f + __expf(-al_mine));
        const float q = 1.0f - a;
        float dot = 0.f;
        for (int kk = 0; kk < K; ++kk) dot = fmaf(ths[kk * CB + c], tot[kk * CB + c], dot);
        float acc = 0.f, pw = 1.0f, pwm1 = 0.f;
        for (int kk = 0; kk < K; ++kk) {
            const float dt = ths[kk * CB + c] * (tot[kk * CB + c] - dot);
            acc = fmaf(dt, pw - (float)kk * a * pwm1, acc);
            pwm1 = pw;
            pw *= q;
        }
        f.galpha[dd] = a * q * acc;
    }
}

// theta[k,d] = softmax_k(a (1-a)^k), a = sigmoid(alpha[d])  (geo_theta.hip)
int geo_theta_fwd_launch(const float* alpha, int K, int D, float* theta, hipStream_t s);
// gtheta[k,d] = sum_b slab[b][k,d] (block order) and galpha = d(theta)/d(alpha)^T gtheta, one launch (geo_theta.hip)
int gtheta_finish_launch(const float* slab, int nslab, const float* alpha, const float* theta, int K, int D, float* gtheta,
                         float* galpha, hipStream_t s);

// Count-matrix x g-tile table gradients on the matrix cores (table_grad_mfma.hip): *handled tells whether the launch
// was done; otherwise kpgnn_table_grad falls back to its register-walk kernel.
int table_grad_mfma(const kpgnn_table_grad_desc* d, hipStream_t s, bool* handled);
size_t table_grad_mfma_ws_bytes(int N, int K, int D, int NT, int n0, int nk, int U);

// Element-per-thread aggregation for narrow rows (aggregate_narrow.hip): *handled tells whether the launch was done.
int agg_narrow_fwd(const kpgnn_agg_fwd_desc* d, hipStream_t s, bool* handled);
int agg_narrow_bwd(const kpgnn_agg_bwd_desc* d, hipStream_t s, bool* handled);
// A graph's hop slab staged in LDS, for dense K-hop neighbourhoods (aggregate_lds.hip; needs desc.graph_ptr): *handled as above.
int agg_lds_fwd(const kpgnn_agg_fwd_desc* d, hipStream_t s, bool* handled);
// One block per node, one unit per hop, for small batches (aggregate_small.hip): *handled as above.
int agg_small_fwd(const kpgnn_agg_fwd_desc* d, hipStream_t s, bool* handled);
int agg_small_bwd(const kpgnn_agg_bwd_desc* d, hipStream_t s, bool* handled);

// erf(z) by Abramowitz-Stegun 7.1.26 (max abs error 5.4e-7 in fp32 over [-6,6]; exact +-1 beyond): ~14 VALU ops
// against ~30 for libm's erff, which made the GELU epilogue VALU-bound (28 us of a 160 us launch).  Also
// returns e2 = exp(-z*z), which the backward needs for the Gaussian density.
__device__ __forceinline__ float fast_erf(float z, float* e2_out) {
    const float az = fabsf(z);
    // v_rcp_f32 (1 ulp) - __frcp_rn expands to the 12-instruction correctly-rounded division sequence, which was
    // ~45 % of the GELU epilogue's VALU work; the A&S formula itself is only good to 5e-7
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
    float poly = 1.061405429f;
    poly = fmaf(poly, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e2 = __expf(-az * az);
    *e2_out = e2;
    return copysignf(fmaf(-poly * t, e2, 1.0f), z);
}

// bf16 STORAGE of rows that are otherwise fp32 (kpgnn.h: `storage` = KPGNN_STORE_BF16): the value is the upper half of the
// fp32 pattern, rounded to nearest even on the way out; every sum stays fp32.
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;      // (inf stays inf; a NaN stays a NaN unless its payload is only low bits)
}
template <int VEC> __device__ __forceinline__ void ld_bf16(const void* q, float (&v)[VEC]) {
    if (VEC == 4) {
        const uint2 t = *reinterpret_cast<const uint2*>(q);
        v[0] = __uint_as_float(t.x << 16); v[1 % VEC] = __uint_as_float(t.x & 0xFFFF0000u);
        v[2 % VEC] = __uint_as_float(t.y << 16); v[3 % VEC] = __uint_as_float(t.y & 0xFFFF0000u);
    } else if (VEC == 2) {
        const uint32_t t = *reinterpret_cast<const uint32_t*>(q);
        v[0] = __uint_as_float(t << 16); v[1 % VEC] = __uint_as_float(t & 0xFFFF0000u);
    } else {
        v[0] = __uint_as_float((uint32_t)*reinterpret_cast<const uint16_t*>(q) << 16);
    }
}
// streaming forms (nt): rows nobody re-reads in this launch should not displace the gathered rows in L2
template <int VEC> __device__ __forceinline__ void ld_bf16_stream(const void* q, float (&v)[VEC]) {
    if (VEC == 4) {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(q);
        const uint32_t x = __builtin_nontemporal_load(w), y = __builtin_nontemporal_load(w + 1);
        v[0] = __uint_as_float(x << 16); v[1 % VEC] = __uint_as_float(x & 0xFFFF0000u);
        v[2 % VEC] = __uint_as_float(y << 16); v[3 % VEC] = __uint_as_float(y & 0xFFFF0000u);
    } else {
        ld_bf16<VEC>(q, v);
    }
}
template <int VEC> __device__ __forceinline__ void st_bf16_stream(void* q, const float (&v)[VEC]) {
    if (VEC == 4) {
        uint32_t* w = reinterpret_cast<uint32_t*>(q);
        __builtin_nontemporal_store(f32_to_bf16_bits(v[0]) | (f32_to_bf16_bits(v[1 % VEC]) << 16), w);
        __builtin_nontemporal_store(f32_to_bf16_bits(v[2 % VEC]) | (f32_to_bf16_bits(v[3 % VEC]) << 16), w + 1);
    } else if (VEC == 2) {
        __builtin_nontemporal_store(f32_to_bf16_bits(v[0]) | (f32_to_bf16_bits(v[1 % VEC]) << 16), reinterpret_cast<uint32_t*>(q));
    } else {
        *reinterpret_cast<uint16_t*>(q) = (uint16_t)f32_to_bf16_bits(v[0]);
    }
}
template <int VEC> __device__ __forceinline__ void st_bf16(void* q, const float (&v)[VEC]) {
    if (VEC == 4) {
        uint2 t;
        t.x = f32_to_bf16_bits(v[0]) | (f32_to_bf16_bits(v[1 % VEC]) << 16);
        t.y = f32_to_bf16_bits(v[2 % VEC]) | (f32_to_bf16_bits(v[3 % VEC]) << 16);
        *reinterpret_cast<uint2*>(q) = t;
    } else if (VEC == 2) {
        *reinterpret_cast<uint32_t*>(q) = f32_to_bf16_bits(v[0]) | (f32_to_bf16_bits(v[1 % VEC]) << 16);
    } else {
        *reinterpret_cast<uint16_t*>(q) = (uint16_t)f32_to_bf16_bits(v[0]);
    }
}

// XCD-aware tile order.  Blocks b and b+8 share an XCD (own L2).  The tile range is cut into 8
// contiguous slabs, one per XCD, and the blocks of one XCD walk their slab tile by tile, so that
// the rows a slab's graphs gather stay inside one 4 MiB L2.  Placement only affects speed.
struct XcdTileWalk {
    int64_t cur, end, step;
    __device__ XcdTileWalk(int64_t num_tiles) {
        const int64_t b = blockIdx.x, g = gridDim.x;
        if (g < kNumXcd || (g % kNumXcd) != 0) {  // plain grid-stride
            cur = b; end = num_tiles; step = g;
            return;
        }
        const int64_t xcd = b % kNumXcd, slot = b / kNumXcd, nslot = g / kNumXcd;
        const int64_t per = (num_tiles + kNumXcd - 1) / kNumXcd;
        const int64_t lo = xcd * per;
        int64_t hi = lo + per;
        if (hi > num_tiles) hi = num_tiles;
        cur = lo + slot; end = hi; step = nslot;
    }
    __device__ bool valid() const { return cur < end; }
    __device__ void next() { cur += step; }
};

// Blocks of a kernel instantiation that fit one CU with `lds` bytes of dynamic LDS.  Persistent grids are sized
// to ONE resident round: with more blocks than fit, the late starters leave the tail unbalanced
// (agg_fwd at 6 blocks/CU where 5 fit: 96 vs 85 us), with fewer the CU runs below its occupancy (agg_bwd at 4 of 6:
// 67 vs 60 us).  Cached per instantiation and LDS size; 0 on failure (the caller then uses its default).
template <typename Kernel>
int resident_blocks(Kernel kernel, int block_threads, size_t lds) {
    // (all instantiations share this function's statics - they have the same pointer type - so the small cache is keyed
    //  by the kernel's address too)
    struct Entry { const void* k; size_t lds; int threads, nb; };
    static thread_local Entry cache[64] = {};
    static thread_local int next = 0;
    for (const Entry& e : cache)
        if (e.k == (const void*)kernel && e.lds == lds && e.threads == block_threads) return e.nb;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, block_threads, lds) != hipSuccess) { (void)hipGetLastError(); nb = 0; }
    cache[next] = Entry{(const void*)kernel, lds, block_threads, nb};
    next = (next + 1) % 64;
    return nb;
}


}  // namespace kpgnn
