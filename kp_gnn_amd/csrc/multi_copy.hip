// Many small device-to-device copies in one or two launches (gfx950).  Contract: include/kpgnn.h, kpgnn_multi_copy.
//
// A training step ends by moving ~190 freshly computed parameter gradients into their views of ONE flat bucket (the unit of
// the RCCL all-reduce and of the fused optimiser step, dp.py).  The framework's multi-tensor copy takes three ~15-us launches
// for that; here the (source, destination, count) triples travel BY VALUE in the kernel arguments - inside a captured
// hipGraph the addresses are the same at every replay, so the table is baked into the graph node - and a launch copies up
// to kMcMax tensors: blockIdx.y = tensor, blockIdx.x = 4 KB chunk.
#include "kpgnn_common.h"

namespace kpgnn {
namespace {

constexpr int kMcMax = 192;           // 192 * (8 + 8 + 4) B = 3.8 KB of kernel arguments (the limit is 4 KB)
constexpr int kMcChunk = 1024;        // floats per block

struct McArgs {
    const float* src[kMcMax];
    float* dst[kMcMax];
    int32_t n[kMcMax];
};

__global__ void __launch_bounds__(256) multi_copy_kernel(const McArgs a) {
    const int t = blockIdx.y;
    const int n = a.n[t];
    const int base = blockIdx.x * kMcChunk;
    if (base >= n) return;
    const float* __restrict__ s = a.src[t];
    float* __restrict__ d = a.dst[t];
    const int end = min(n, base + kMcChunk);
    if ((((uintptr_t)s | (uintptr_t)d) & 15) == 0) {
        for (int i = base + threadIdx.x * 4; i < end; i += 256 * 4) {
            if (i + 3 < end) *reinterpret_cast<float4*>(d + i) = *reinterpret_cast<const float4*>(s + i);
            else for (int q = i; q < end; ++q) d[q] = s[q];
        }
    } else {
        for (int i = base + threadIdx.x; i < end; i += 256) d[i] = s[i];
    }
}

}  // namespace
}  // namespace kpgnn

using namespace kpgnn;

extern "C" int kpgnn_multi_copy(int32_t count, const float* const* src, float* const* dst, const int64_t* numel,
                                kpgnn_stream_t stream) {
    KPGNN_REQUIRE(count >= 0 && (count == 0 || (src && dst && numel)), "multi_copy: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    for (int c0 = 0; c0 < count; c0 += kMcMax) {
        McArgs a;
        const int m = count - c0 < kMcMax ? count - c0 : kMcMax;
        int64_t biggest = 0;
        for (int i = 0; i < kMcMax; ++i) {
            if (i < m) {
                KPGNN_REQUIRE(numel[c0 + i] >= 0 && numel[c0 + i] < (1ll << 31) && (numel[c0 + i] == 0 || (src[c0 + i] && dst[c0 + i])),
                              "multi_copy: bad entry %d", c0 + i);
                a.src[i] = src[c0 + i]; a.dst[i] = dst[c0 + i]; a.n[i] = (int32_t)numel[c0 + i];
                if (numel[c0 + i] > biggest) biggest = numel[c0 + i];
            } else { a.src[i] = nullptr; a.dst[i] = nullptr; a.n[i] = 0; }
        }
        if (biggest == 0) continue;
        const unsigned gx = (unsigned)((biggest + kMcChunk - 1) / kMcChunk);
        if (gx > 65535u) return fail(KPGNN_ELIMIT, "multi_copy: a tensor of %lld floats exceeds 64 M", (long long)biggest);
        hipLaunchKernelGGL(multi_copy_kernel, dim3(gx, (unsigned)m), dim3(256), 0, s, a);
        KPGNN_LAUNCH_CHECK("multi_copy_kernel");
    }
    return KPGNN_OK;
}
