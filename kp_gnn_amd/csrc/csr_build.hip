// K-hop CSR construction on the device (gfx950).  See include/kpgnn.h for the layout contract.
//
// The reference hands PyG the raw [2,E] edge list plus an [E,K] int64 mask/code matrix and lets
// propagate() walk all E*K slots (layers/KPGIN.py:100,115-118).  Here the active (edge,hop) pairs
// are compacted and stably radix-sorted by (node,hop) once per batch, in both orientations, so the
// aggregation kernels stream int32 neighbour ids + uint16 codes and never touch an inactive slot.
#include <cstring>
#include <mutex>

#include <rocprim/rocprim.hpp>

#include "kpgnn_common.h"

namespace kpgnn {

char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

const DeviceFacts& device_facts() {
    static DeviceFacts f = [] {
        DeviceFacts d;
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) {
            d.cu_count = p.multiProcessorCount;
            d.lds_per_block = (int)p.sharedMemPerBlock;
            d.valid = true;
        }
        return d;
    }();
    return f;
}

hipError_t ensure_dynamic_lds(const void* func, size_t bytes) {
    // small fixed table keyed by function pointer; guarded by a mutex (launches may come from several threads)
    static std::mutex mu;
    static const void* funcs[256];
    static size_t sizes[256];
    static int count = 0;
    if (bytes <= 64 * 1024) return hipSuccess;
    std::lock_guard<std::mutex> lock(mu);
    int idx = -1;
    for (int i = 0; i < count; ++i) if (funcs[i] == func) { idx = i; break; }
    if (idx >= 0 && sizes[idx] >= bytes) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    if (idx < 0 && count < 256) { idx = count++; funcs[idx] = func; sizes[idx] = 0; }
    if (idx >= 0) sizes[idx] = bytes;
    return hipSuccess;
}

namespace {

constexpr int kThreads = 256;

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// ---------------------------------------------------------------------------------------------- stats
__global__ void __launch_bounds__(kThreads)
csr_stats_kernel(const int64_t* __restrict__ ei, int64_t ei_stride, const int64_t* __restrict__ attr,
                 int64_t attr_stride, int64_t E, int K, unsigned long long* __restrict__ stats) {
    long long active = 0, max0 = 0, maxk = 0, minv = 0, nmin = INT64_MAX, nmax = -1;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t* row = attr + e * attr_stride;
        for (int k = 0; k < K; ++k) {
            long long v = row[k];
            active += (v != 0);
            if (k == 0) max0 = max(max0, v); else maxk = max(maxk, v);
            minv = min(minv, v);
        }
        long long s = ei[e], d = ei[ei_stride + e];
        nmin = min(nmin, min(s, d));
        nmax = max(nmax, max(s, d));
    }
    // wave reduction (64 lanes), then one atomic per wave
    for (int off = 32; off > 0; off >>= 1) {
        active += __shfl_down(active, off);
        max0 = max(max0, __shfl_down(max0, off));
        maxk = max(maxk, __shfl_down(maxk, off));
        minv = min(minv, __shfl_down(minv, off));
        nmin = min(nmin, __shfl_down(nmin, off));
        nmax = max(nmax, __shfl_down(nmax, off));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&stats[0], (unsigned long long)active);
        atomicMax((long long*)&stats[1], max0);
        atomicMax((long long*)&stats[2], maxk);
        atomicMin((long long*)&stats[3], minv);
        atomicMin((long long*)&stats[4], nmin);
        atomicMax((long long*)&stats[5], nmax);
    }
}

__global__ void csr_stats_init_kernel(long long* stats) {
    stats[0] = 0; stats[1] = 0; stats[2] = 0; stats[3] = 0; stats[4] = INT64_MAX; stats[5] = -1; stats[6] = 0; stats[7] = 0;
}

// ---------------------------------------------------------------------------------------------- build
__global__ void __launch_bounds__(kThreads)
edge_active_count_kernel(const int64_t* __restrict__ attr, int64_t attr_stride, int64_t E, int K,
                         int32_t* __restrict__ cnt) {
    int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e > E) return;
    int c = 0;
    if (e < E) {
        const int64_t* row = attr + e * attr_stride;
        for (int k = 0; k < K; ++k) c += (row[k] != 0);
    }
    cnt[e] = c;  // cnt[E] = 0 so the exclusive scan yields the total at offs[E]
}

// orientation 0: key = dst*K+k, payload col = src ; orientation 1: key = src*K+k, payload col = dst
__global__ void __launch_bounds__(kThreads)
expand_pairs_kernel(const int64_t* __restrict__ ei, int64_t ei_stride, const int64_t* __restrict__ attr,
                    int64_t attr_stride, int64_t E, int K, int orientation, const int32_t* __restrict__ offs,
                    uint32_t* __restrict__ keys, uint64_t* __restrict__ vals) {
    int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int64_t src = ei[e], dst = ei[ei_stride + e];
    const int64_t owner = orientation == 0 ? dst : src;
    const int64_t other = orientation == 0 ? src : dst;
    const int64_t* row = attr + e * attr_stride;
    int32_t pos = offs[e];
    for (int k = 0; k < K; ++k) {
        const int64_t v = row[k];
        if (v != 0) {
            keys[pos] = (uint32_t)(owner * K + k);
            vals[pos] = ((uint64_t)other << 16) | (uint64_t)(v & 0xFFFF);
            ++pos;
        }
    }
}

// rowptr[s] = first sorted position whose key >= s (binary search per segment: no atomics, no
// dependence on how empty segments cluster); the same launch unpacks the sorted payloads.
__global__ void __launch_bounds__(kThreads)
rowptr_unpack_kernel(const uint32_t* __restrict__ keys, const uint64_t* __restrict__ vals, int64_t A, int64_t S,
                     int32_t* __restrict__ rowptr, int32_t* __restrict__ col, uint16_t* __restrict__ code) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t < A) {
        const uint64_t v = vals[t];
        col[t] = (int32_t)(v >> 16);
        code[t] = (uint16_t)(v & 0xFFFF);
    }
    if (t <= S) {
        int64_t lo = 0, hi = A;  // first a in [0,A] with keys[a] >= t
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)keys[mid] < t) lo = mid + 1; else hi = mid;
        }
        rowptr[t] = (int32_t)lo;
    }
}

// Third ordering, for the table-gradient kernels: the active pairs of a tile of `nodes_per_tile` destination nodes
// sorted by (tile, table, code, hop, node_in_tile), with DUPLICATES MERGED: every pair of one (node, hop) segment that
// carries the same code adds the same row g[node, hop, :] to the same table row, so the list holds one entry per
// distinct (node, hop, code) with a multiplicity (2.3x fewer entries on ZINC-shaped batches).  Code-major on purpose:
// the walk kernel keeps the running sum of a table row in a register and flushes it with an LDS atomic when the row
// changes - a hop-major list (which would let a layer with k < K hops walk a prefix) triples the flushes and was
// measured slower (90 vs 78 us at k = 8) although it walks fewer entries.
//   key   = tile<<26 | table<<25 | code<<9 | hop<<3 | node_in_tile          (sorted, K <= 62; table = hop > 0)
//   entry = table<<31 | code<<15 | node_in_tile<<12 | (multiplicity-1)<<6 | hop
// Runs are cut every 64 entries (counted from the run's first entry), so a multiplicity fits its 6 bits.
__global__ void __launch_bounds__(kThreads)
expand_tile_keys_kernel(const int64_t* __restrict__ ei, int64_t ei_stride, const int64_t* __restrict__ attr,
                        int64_t attr_stride, int64_t E, int K, int nodes_per_tile,
                        const int32_t* __restrict__ offs, uint64_t* __restrict__ keys) {
    int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int64_t dst = ei[ei_stride + e];
    const uint64_t tile = (uint64_t)(dst / nodes_per_tile);
    const uint64_t nit = (uint64_t)(dst % nodes_per_tile);
    const int64_t* row = attr + e * attr_stride;
    int32_t pos = offs[e];
    for (int k = 0; k < K; ++k) {
        if (row[k] != 0) {
            keys[pos] = (tile << 26) | ((uint64_t)(k > 0) << 25) | ((uint64_t)(row[k] & 0xFFFF) << 9) | ((uint64_t)k << 3) | nit;
            ++pos;
        }
    }
}

// A run of equal keys is cut every 64 entries COUNTED FROM ITS OWN FIRST ENTRY (not from the list position): the merged list
// of a tile is then a function of the tile's pairs alone, so per-node lists built once per dataset can be merged into the
// same list at collate time (collate.hip).  The first entry of a run is found by galloping back, then bisecting.
__device__ __forceinline__ bool tile_run_start(const uint64_t* __restrict__ keys, int64_t i) {
    if (i == 0) return true;
    const uint64_t key = keys[i];
    if (keys[i - 1] != key) return true;
    int64_t step = 2, lo;                 // keys[i - 1] == key
    for (;;) {
        lo = i - step;
        if (lo <= 0) { lo = 0; break; }
        if (keys[lo] != key) break;
        step <<= 1;
    }
    // invariant: keys[hi] == key; keys[lo] != key unless lo == 0 reached by clamping
    int64_t hi = i - (step >> 1);
    if (keys[lo] == key) hi = lo;
    else
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (keys[mid] == key) hi = mid; else lo = mid;
        }
    return ((i - hi) & 63) == 0;          // hi = first entry of the run
}

__global__ void __launch_bounds__(kThreads)
tile_run_flag_kernel(const uint64_t* __restrict__ keys, int64_t A, int32_t* __restrict__ flag) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < A) flag[i] = tile_run_start(keys, i) ? 1 : 0;
}

// idx = exclusive scan of the run-start flags = position of a run in the merged list
__global__ void __launch_bounds__(kThreads)
tile_emit_kernel(const uint64_t* __restrict__ keys, const int32_t* __restrict__ idx, int64_t A,
                 uint32_t* __restrict__ pack) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= A || !tile_run_start(keys, i)) return;
    const uint64_t key = keys[i];
    uint32_t len = 1;
    while (i + len < A && len < 64 && keys[i + len] == key) ++len;
    const uint32_t hop = (uint32_t)(key >> 3) & 63u, code = (uint32_t)(key >> 9) & 0xFFFFu, nit = (uint32_t)key & 7u;
    pack[idx[i]] = ((hop > 0 ? 1u : 0u) << 31) | (code << 15) | (nit << 12) | ((len - 1) << 6) | hop;
}

// tptr[tile] = first merged entry of the tile; tptr[num_tiles] = number of entries
__global__ void __launch_bounds__(kThreads)
tile_ptr_kernel(const uint64_t* __restrict__ keys, const int32_t* __restrict__ idx, int64_t A, int64_t num_tiles,
                int32_t* __restrict__ tptr) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t > num_tiles) return;
    int64_t lo = 0, hi = A;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)(keys[mid] >> 26) < t) lo = mid + 1; else hi = mid;
    }
    // (a tile boundary is a key change, i.e. a run start: idx[lo] is its merged position)
    tptr[t] = lo < A ? idx[lo] : idx[A - 1] + (tile_run_start(keys, A - 1) ? 1 : 0);   // (else: number of runs)
}

struct Workspace {
    int32_t* offs;
    uint32_t *keys_a, *keys_b;
    uint64_t *vals_a, *vals_b;
    void* prim_temp;
    size_t prim_bytes;
    size_t total;
};

int sort_bits(int64_t S) {
    int b = 1;
    while (b < 32 && ((int64_t)1 << b) < S) ++b;
    return b;
}

hipError_t plan_workspace(int64_t E, int64_t A, int64_t S, char* base, Workspace* w) {
    size_t scan_bytes = 0, sort_bytes = 0;
    hipError_t e = rocprim::exclusive_scan(nullptr, scan_bytes, (int32_t*)nullptr, (int32_t*)nullptr, 0,
                                           (size_t)(E + 1 > A ? E + 1 : A), rocprim::plus<int32_t>());
    if (e != hipSuccess) return e;
    e = rocprim::radix_sort_pairs(nullptr, sort_bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint64_t*)nullptr,
                                  (uint64_t*)nullptr, (size_t)(A > 0 ? A : 1), 0, sort_bits(S));
    if (e != hipSuccess) return e;
    size_t sort64_bytes = 0;
    e = rocprim::radix_sort_keys(nullptr, sort64_bytes, (uint64_t*)nullptr, (uint64_t*)nullptr,
                                 (size_t)(A > 0 ? A : 1), 0, 64);
    if (e != hipSuccess) return e;
    if (sort64_bytes > sort_bytes) sort_bytes = sort64_bytes;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes); return p; };
    const size_t a = (size_t)(A > 0 ? A : 1);
    w->offs = (int32_t*)take(sizeof(int32_t) * (size_t)(E + 1));
    w->keys_a = (uint32_t*)take(sizeof(uint32_t) * a);
    w->keys_b = (uint32_t*)take(sizeof(uint32_t) * a);
    w->vals_a = (uint64_t*)take(sizeof(uint64_t) * a);
    w->vals_b = (uint64_t*)take(sizeof(uint64_t) * a);
    w->prim_bytes = scan_bytes > sort_bytes ? scan_bytes : sort_bytes;
    w->prim_temp = take(w->prim_bytes ? w->prim_bytes : 1);
    w->total = off;
    return hipSuccess;
}

}  // namespace
}  // namespace kpgnn

using namespace kpgnn;

extern "C" int kpgnn_abi_version(void) { return KPGNN_ABI_VERSION; }

extern "C" const char* kpgnn_last_error(void) { return error_buffer(); }

extern "C" int kpgnn_device_info(int* cu_count, int* lds_bytes_per_block, int* wavefront, char* arch, int arch_len) {
    int dev = 0;
    KPGNN_HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t p;
    KPGNN_HIP_TRY(hipGetDeviceProperties(&p, dev));
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (lds_bytes_per_block) *lds_bytes_per_block = (int)p.sharedMemPerBlock;
    if (wavefront) *wavefront = p.warpSize;
    if (arch && arch_len > 0) snprintf(arch, (size_t)arch_len, "%s", p.gcnArchName);
    return KPGNN_OK;
}

extern "C" int kpgnn_csr_stats(const int64_t* edge_index, int64_t ei_stride, const int64_t* edge_attr,
                               int64_t attr_stride, int64_t E, int32_t K, int64_t* stats, kpgnn_stream_t stream) {
    KPGNN_REQUIRE(stats != nullptr, "csr_stats: stats is NULL");
    KPGNN_REQUIRE(E >= 0 && K >= 1 && K <= 4096, "csr_stats: bad E=%lld K=%d", (long long)E, K);
    KPGNN_REQUIRE(E == 0 || (edge_index && edge_attr), "csr_stats: NULL edge_index/edge_attr with E>0");
    KPGNN_REQUIRE(E == 0 || (ei_stride >= E && attr_stride >= K), "csr_stats: strides too small");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(csr_stats_init_kernel, dim3(1), dim3(1), 0, s, (long long*)stats);
    KPGNN_LAUNCH_CHECK("csr_stats_init_kernel");
    if (E > 0) {
        int64_t blocks = (E + kThreads - 1) / kThreads;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(csr_stats_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, s, edge_index, ei_stride,
                           edge_attr, attr_stride, E, (int)K, (unsigned long long*)stats);
        KPGNN_LAUNCH_CHECK("csr_stats_kernel");
    }
    return KPGNN_OK;
}

extern "C" size_t kpgnn_csr_workspace_bytes(int64_t E, int64_t A, int64_t N, int32_t K) {
    if (E < 0 || A < 0 || N < 0 || K < 1) return 0;
    Workspace w;
    if (plan_workspace(E, A, N * (int64_t)K, nullptr, &w) != hipSuccess) return 0;
    return w.total;
}

extern "C" int kpgnn_csr_build(const int64_t* edge_index, int64_t ei_stride, const int64_t* edge_attr,
                               int64_t attr_stride, int64_t E, int32_t K, int64_t N, int64_t A,
                               int32_t* rowptr_dst, int32_t* col_dst, uint16_t* code_dst,
                               int32_t* rowptr_src, int32_t* col_src, uint16_t* code_src,
                               int32_t nodes_per_tile, int32_t* tile_ptr, uint32_t* tile_pack,
                               void* workspace, size_t workspace_bytes, kpgnn_stream_t stream) {
    KPGNN_REQUIRE(E >= 0 && N >= 0 && A >= 0 && K >= 1, "csr_build: bad sizes E=%lld N=%lld A=%lld K=%d",
                  (long long)E, (long long)N, (long long)A, K);
    const int64_t S = N * (int64_t)K;
    if (S >= ((int64_t)1 << 31) || A >= ((int64_t)1 << 31) || E >= ((int64_t)1 << 31) - 1)
        return fail(KPGNN_ELIMIT, "csr_build: N*K=%lld, A=%lld or E=%lld exceeds the int32 index range",
                    (long long)S, (long long)A, (long long)E);
    KPGNN_REQUIRE(A <= E * (int64_t)K, "csr_build: A=%lld > E*K", (long long)A);
    KPGNN_REQUIRE(rowptr_dst && rowptr_src, "csr_build: NULL rowptr");
    KPGNN_REQUIRE(A == 0 || (col_dst && code_dst && col_src && code_src), "csr_build: NULL col/code with A>0");
    KPGNN_REQUIRE(E == 0 || (edge_index && edge_attr && ei_stride >= E && attr_stride >= K),
                  "csr_build: bad edge_index/edge_attr pointers or strides");
    hipStream_t s = (hipStream_t)stream;
    KPGNN_REQUIRE(tile_ptr == nullptr || (nodes_per_tile >= 1 && nodes_per_tile <= 8 && (A == 0 || tile_pack)),
                  "csr_build: tile list needs 1 <= nodes_per_tile <= 8 and tile_pack");
    if (tile_ptr && K > 62) return fail(KPGNN_ELIMIT, "csr_build: the tile list packs the hop in 6 bits (K=%d > 62)", K);
    const int64_t num_tiles = tile_ptr ? (N + nodes_per_tile - 1) / nodes_per_tile : 0;
    if (A == 0) {
        KPGNN_HIP_TRY(hipMemsetAsync(rowptr_dst, 0, sizeof(int32_t) * (size_t)(S + 1), s));
        KPGNN_HIP_TRY(hipMemsetAsync(rowptr_src, 0, sizeof(int32_t) * (size_t)(S + 1), s));
        if (tile_ptr) KPGNN_HIP_TRY(hipMemsetAsync(tile_ptr, 0, sizeof(int32_t) * (size_t)(num_tiles + 1), s));
        return KPGNN_OK;
    }
    Workspace w;
    KPGNN_HIP_TRY(plan_workspace(E, A, S, (char*)workspace, &w));
    KPGNN_REQUIRE(workspace != nullptr && workspace_bytes >= w.total, "csr_build: workspace too small (%zu < %zu)",
                  workspace_bytes, w.total);
    KPGNN_REQUIRE(((uintptr_t)workspace & 255) == 0, "csr_build: workspace must be 256-byte aligned");

    const unsigned eblocks = (unsigned)((E + 1 + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(edge_active_count_kernel, dim3(eblocks), dim3(kThreads), 0, s, edge_attr, attr_stride, E,
                       (int)K, w.offs);
    KPGNN_LAUNCH_CHECK("edge_active_count_kernel");
    size_t tb = w.prim_bytes;
    KPGNN_HIP_TRY(rocprim::exclusive_scan(w.prim_temp, tb, w.offs, w.offs, 0, (size_t)(E + 1),
                                          rocprim::plus<int32_t>(), s));
    const int bits = sort_bits(S);
    const unsigned ablocks = (unsigned)(((A > S + 1 ? A : S + 1) + kThreads - 1) / kThreads);
    for (int orientation = 0; orientation < 2; ++orientation) {
        hipLaunchKernelGGL(expand_pairs_kernel, dim3(eblocks), dim3(kThreads), 0, s, edge_index, ei_stride,
                           edge_attr, attr_stride, E, (int)K, orientation, w.offs, w.keys_a, w.vals_a);
        KPGNN_LAUNCH_CHECK("expand_pairs_kernel");
        tb = w.prim_bytes;
        KPGNN_HIP_TRY(rocprim::radix_sort_pairs(w.prim_temp, tb, w.keys_a, w.keys_b, w.vals_a, w.vals_b, (size_t)A,
                                                0, bits, s));
        hipLaunchKernelGGL(rowptr_unpack_kernel, dim3(ablocks), dim3(kThreads), 0, s, w.keys_b, w.vals_b, A, S,
                           orientation == 0 ? rowptr_dst : rowptr_src, orientation == 0 ? col_dst : col_src,
                           orientation == 0 ? code_dst : code_src);
        KPGNN_LAUNCH_CHECK("rowptr_unpack_kernel");
    }
    if (tile_ptr) {  // merged (tile, table, code, hop, node)-sorted entry list; the u64/u32 buffers swap roles
        uint64_t* k64a = w.vals_a; uint64_t* k64b = w.vals_b;
        int32_t* idx = (int32_t*)w.keys_a;
        hipLaunchKernelGGL(expand_tile_keys_kernel, dim3(eblocks), dim3(kThreads), 0, s, edge_index, ei_stride,
                           edge_attr, attr_stride, E, (int)K, (int)nodes_per_tile, w.offs, k64a);
        KPGNN_LAUNCH_CHECK("expand_tile_keys_kernel");
        int tbits = 26 + sort_bits(num_tiles + 1);
        if (tbits > 64) tbits = 64;
        tb = w.prim_bytes;
        KPGNN_HIP_TRY(rocprim::radix_sort_keys(w.prim_temp, tb, k64a, k64b, (size_t)A, 0, tbits, s));
        const unsigned pblocks = (unsigned)((A + kThreads - 1) / kThreads);
        hipLaunchKernelGGL(tile_run_flag_kernel, dim3(pblocks), dim3(kThreads), 0, s, k64b, A, idx);
        KPGNN_LAUNCH_CHECK("tile_run_flag_kernel");
        tb = w.prim_bytes;
        KPGNN_HIP_TRY(rocprim::exclusive_scan(w.prim_temp, tb, idx, idx, 0, (size_t)A, rocprim::plus<int32_t>(), s));
        hipLaunchKernelGGL(tile_emit_kernel, dim3(pblocks), dim3(kThreads), 0, s, k64b, idx, A, tile_pack);
        KPGNN_LAUNCH_CHECK("tile_emit_kernel");
        const unsigned tblocks = (unsigned)((num_tiles + 1 + kThreads - 1) / kThreads);
        hipLaunchKernelGGL(tile_ptr_kernel, dim3(tblocks), dim3(kThreads), 0, s, k64b, idx, A, num_tiles, tile_ptr);
        KPGNN_LAUNCH_CHECK("tile_ptr_kernel");
    }
    return KPGNN_OK;
}
