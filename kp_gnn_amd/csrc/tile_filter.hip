// Hop-prefix copies of the table-gradient entry list (gfx950).  Contract: include/kpgnn.h, kpgnn_tile_pack_filter.
//
// kpgnn_csr_build writes ONE (tile_ptr, tile_pack) list over all K hops, sorted by (tile, table, code, hop, node).  A layer
// that aggregates only the first k hops (models/GNNs.py:421-423: layer l sees hops < min(l+1, K)) needs the entries with
// hop < k.  kpgnn_table_grad skips the others, but it splits a tile's list among its waves by position: with the inactive
// entries still in the list the active ones bunch up in a few waves (3.4 K cycles in wave 0 against 0.6 K in wave 7 at
// k = 2).  The filtered copy keeps the order (so it stays sorted by row) and is static per batch: built once per k.
#include "kpgnn_common.h"

namespace kpgnn {
namespace {

// one wave per tile: cnt[tile] = entries with hop < k
__global__ void __launch_bounds__(kWave)
tile_count_kernel(const int32_t* __restrict__ tptr, const uint32_t* __restrict__ tpack, int k, int32_t* __restrict__ cnt) {
    const int tl = blockIdx.x, lane = threadIdx.x;
    const int b = tptr[tl], e = tptr[tl + 1];
    int n = 0;
    for (int i = b + lane; i < e; i += kWave) n += ((int)(tpack[i] & 0x3F) < k) ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) n += __shfl_down(n, o);
    if (lane == 0) cnt[tl] = n;
}

// exclusive scan of cnt[0..n) into out[0..n], one block (n is the number of tiles: ~N/8)
__global__ void __launch_bounds__(1024)
tile_scan_kernel(const int32_t* __restrict__ cnt, int n, int32_t* __restrict__ out) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int b = t * per, e = min(n, b + per);
    int s = 0;
    for (int i = b; i < e; ++i) s += cnt[i];
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = t ? part[t - 1] : 0;
    for (int i = b; i < e; ++i) { out[i] = run; run += cnt[i]; }
    if (t == 1023) out[n] = part[1023];
}

// one wave per tile: stable compaction of the entries with hop < k
__global__ void __launch_bounds__(kWave)
tile_compact_kernel(const int32_t* __restrict__ tptr, const uint32_t* __restrict__ tpack, int k,
                    const int32_t* __restrict__ optr, uint32_t* __restrict__ opack) {
    const int tl = blockIdx.x, lane = threadIdx.x;
    const int b = tptr[tl], e = tptr[tl + 1];
    int o = optr[tl];
    for (int i0 = b; i0 < e; i0 += kWave) {
        const int i = i0 + lane;
        const uint32_t w = i < e ? tpack[i] : 0xFFFFFFFFu;
        const bool keep = i < e && (int)(w & 0x3F) < k;
        const unsigned long long m = __ballot(keep);
        if (keep) opack[o + __popcll(m & ((1ull << lane) - 1))] = w;
        o += __popcll(m);
    }
}

}  // namespace
}  // namespace kpgnn

using namespace kpgnn;

extern "C" int kpgnn_tile_pack_filter(const int32_t* tile_ptr, const uint32_t* tile_pack, int64_t num_tiles, int32_t k,
                                      int32_t* out_ptr, uint32_t* out_pack, int32_t* scratch, kpgnn_stream_t stream) {
    KPGNN_REQUIRE(num_tiles >= 0 && num_tiles < (1ll << 30) && k >= 1, "tile_pack_filter: bad num_tiles=%lld k=%d",
                  (long long)num_tiles, k);
    if (num_tiles == 0) return KPGNN_OK;
    KPGNN_REQUIRE(tile_ptr && tile_pack && out_ptr && out_pack && scratch, "tile_pack_filter: NULL argument");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(tile_count_kernel, dim3((unsigned)num_tiles), dim3(kWave), 0, s, tile_ptr, tile_pack, k, scratch);
    KPGNN_LAUNCH_CHECK("tile_count_kernel");
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, s, scratch, (int)num_tiles, out_ptr);
    KPGNN_LAUNCH_CHECK("tile_scan_kernel");
    hipLaunchKernelGGL(tile_compact_kernel, dim3((unsigned)num_tiles), dim3(kWave), 0, s, tile_ptr, tile_pack, k, out_ptr, out_pack);
    KPGNN_LAUNCH_CHECK("tile_compact_kernel");
    return KPGNN_OK;
}
