// Code-sorted segment list of a K-hop CSR (gfx950).  Contract: include/kpgnn.h, kpgnn_csr_code_segments.
//
// The table gradients of the aggregation  gtable[c,:] = sum over active pairs (i,k) with code c of g[i,k,:]  are a
// scatter of the N*K rows of g into <= 57 accumulator rows.  Round 1 streamed g through LDS tiles a second time for
// it (table_grad.hip: the largest kernel of the step).  With the rows visited in (hop, code) ORDER the scatter
// becomes a segmented sum: consecutive rows add into the same accumulator, which then lives in registers, and the
// backward pre-pass that computes g (combine_sorted.hip) can form the table gradients on the way - g is never re-read.
//
// This file builds that order once per batch from the (dst, hop)-keyed CSR:
//   entries   one per DISTINCT (node, hop, code) with its multiplicity, plus one code-0xFFFF entry for every (node, hop)
//             row without pairs (every row must be visited once: its g row is written and its theta-gradient term added);
//             sorted by (hop, code, node) - hop-major, so the rows of the first k hops are a PREFIX (GNNPlus layer l
//             walks k = min(l, K) hops of the same CSR); `first` marks the entry with the smallest code of its row
//             (the one that writes the row).
//   segments  runs of <= 32 consecutive entries with one (hop, code): the unit a sub-group of lanes sums in registers.
//   hop_seg   hop_seg[k] = number of segments of hops < k.
// All integer work, no atomics: the list is bitwise reproducible.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "kpgnn_common.h"

namespace kpgnn {
namespace {

constexpr int kThreads = 256;
constexpr int kSegEntries = 32;

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// keys per row: max(1, pairs); mincode of the row (0xFFFF when empty)
__global__ void __launch_bounds__(kThreads)
row_count_kernel(const int32_t* __restrict__ rp, const uint16_t* __restrict__ code, int64_t S, int32_t* __restrict__ cnt,
                 uint32_t* __restrict__ mincode) {
    const int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (s > S) return;
    if (s == S) { cnt[S] = 0; return; }
    const int b = rp[s], e = rp[s + 1];
    uint32_t m = 0xFFFFu;
    for (int a = b; a < e; ++a) { const uint32_t c = code[a]; m = c < m ? c : m; }
    cnt[s] = e > b ? e - b : 1;
    mincode[s] = m;
}

// key = hop << 48 | code << 32 | node
__global__ void __launch_bounds__(kThreads)
row_keys_kernel(const int32_t* __restrict__ rp, const uint16_t* __restrict__ code, int64_t S, int K,
                const int32_t* __restrict__ offs, uint64_t* __restrict__ keys) {
    const int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (s >= S) return;
    const uint64_t node = (uint64_t)(s / K), hop = (uint64_t)(s % K);
    const int b = rp[s], e = rp[s + 1];
    int64_t pos = offs[s];
    if (e == b) { keys[pos] = (hop << 48) | (0xFFFFull << 32) | node; return; }
    for (int a = b; a < e; ++a) keys[pos++] = (hop << 48) | ((uint64_t)code[a] << 32) | node;
}

__global__ void __launch_bounds__(kThreads)
key_flag_kernel(const uint64_t* __restrict__ keys, int64_t T, int32_t* __restrict__ flag) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < T) flag[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

// one thread per sorted key that starts a run: emits the merged entry and (hop, code) of the entry
__global__ void __launch_bounds__(kThreads)
entry_emit_kernel(const uint64_t* __restrict__ keys, const int32_t* __restrict__ idx, int64_t T, int K,
                  const uint32_t* __restrict__ mincode, uint32_t* __restrict__ ent, uint32_t* __restrict__ ekey,
                  int32_t* __restrict__ counts) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= T) return;
    const uint64_t key = keys[i];
    if (key == ~0ull) return;                          // sentinel tail (see the host code)
    if (i > 0 && keys[i - 1] == key) return;
    int64_t len = 1;
    while (i + len < T && keys[i + len] == key) ++len;
    const uint32_t node = (uint32_t)key, code = (uint32_t)(key >> 32) & 0xFFFFu, hop = (uint32_t)(key >> 48);
    const uint32_t mult = code == 0xFFFFu ? 0u : (uint32_t)(len > 0x7FFFFFFF ? 0x7FFFFFFF : len);
    const uint32_t first = mincode[(int64_t)node * K + hop] == code ? 0x80000000u : 0u;
    const int32_t e = idx[i];
    ent[2 * (int64_t)e] = node;
    ent[2 * (int64_t)e + 1] = mult | first;
    ekey[e] = (hop << 16) | code;
    if (i + len == T || keys[i + len] == ~0ull) counts[0] = e + 1;   // the last real run: number of entries
}

// v[e] = e where the (hop, code) key changes, else 0  -> inclusive max scan gives the run start of every entry
__global__ void __launch_bounds__(kThreads)
run_start_kernel(const uint32_t* __restrict__ ekey, const int32_t* __restrict__ counts, int64_t cap, int32_t* __restrict__ v) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= cap) return;
    v[e] = (e < counts[0] && e > 0 && ekey[e] != ekey[e - 1]) ? (int32_t)e : 0;
}

__global__ void __launch_bounds__(kThreads)
seg_flag_kernel(const int32_t* __restrict__ rstart, const int32_t* __restrict__ counts, int64_t cap, int32_t* __restrict__ flag) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= cap) return;
    flag[e] = (e < counts[0] && ((e - rstart[e]) % kSegEntries) == 0) ? 1 : 0;
}

__global__ void __launch_bounds__(kThreads)
seg_emit_kernel(const int32_t* __restrict__ rstart, const int32_t* __restrict__ sidx, const uint32_t* __restrict__ ekey,
                int32_t* __restrict__ counts, int32_t* __restrict__ seg_ptr, uint32_t* __restrict__ seg_key) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int32_t ne = counts[0];
    if (e >= ne) return;
    if (((e - rstart[e]) % kSegEntries) == 0) {
        seg_ptr[sidx[e]] = (int32_t)e;
        seg_key[sidx[e]] = ekey[e];
    }
    if (e == ne - 1) {
        const int32_t ns = sidx[e] + ((((e - rstart[e]) % kSegEntries) == 0) ? 1 : 0);   // = number of segment starts
        counts[1] = ns;
        seg_ptr[ns] = ne;
    }
}

// hop_seg[k] = first segment whose hop >= k  (k = 0..K)
__global__ void hop_seg_kernel(const uint32_t* __restrict__ seg_key, const int32_t* __restrict__ counts, int K,
                               int32_t* __restrict__ hop_seg) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > K) return;
    int lo = 0, hi = counts[1];
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((int)(seg_key[mid] >> 16) < k) lo = mid + 1; else hi = mid;
    }
    hop_seg[k] = lo;
}

struct SegWs { int32_t* cnt; uint32_t* mincode; uint64_t *keys_a, *keys_b; int32_t *flag, *idx; uint32_t* ekey; void* prim; size_t prim_bytes, total; };

hipError_t plan(int64_t S, int64_t T, char* base, SegWs* w) {
    size_t scan_b = 0, sort_b = 0, max_b = 0;
    hipError_t e = rocprim::exclusive_scan(nullptr, scan_b, (int32_t*)nullptr, (int32_t*)nullptr, 0,
                                           (size_t)(T > S + 1 ? T : S + 1), rocprim::plus<int32_t>());
    if (e != hipSuccess) return e;
    e = rocprim::radix_sort_keys(nullptr, sort_b, (uint64_t*)nullptr, (uint64_t*)nullptr, (size_t)(T > 0 ? T : 1), 0, 64);
    if (e != hipSuccess) return e;
    e = rocprim::inclusive_scan(nullptr, max_b, (int32_t*)nullptr, (int32_t*)nullptr, (size_t)(T > 0 ? T : 1),
                                rocprim::maximum<int32_t>());
    if (e != hipSuccess) return e;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes); return p; };
    const size_t t = (size_t)(T > 0 ? T : 1);
    w->cnt = (int32_t*)take(sizeof(int32_t) * (size_t)(S + 1));
    w->mincode = (uint32_t*)take(sizeof(uint32_t) * (size_t)(S > 0 ? S : 1));
    w->keys_a = (uint64_t*)take(sizeof(uint64_t) * t);
    w->keys_b = (uint64_t*)take(sizeof(uint64_t) * t);
    w->flag = (int32_t*)take(sizeof(int32_t) * t);
    w->idx = (int32_t*)take(sizeof(int32_t) * t);
    w->ekey = (uint32_t*)take(sizeof(uint32_t) * t);
    w->prim_bytes = scan_b > sort_b ? scan_b : sort_b;
    if (max_b > w->prim_bytes) w->prim_bytes = max_b;
    w->prim = take(w->prim_bytes ? w->prim_bytes : 1);
    w->total = off;
    return hipSuccess;
}

}  // namespace
}  // namespace kpgnn

using namespace kpgnn;

extern "C" size_t kpgnn_csr_code_segments_workspace_bytes(int64_t N, int32_t K, int64_t A) {
    if (N < 0 || K < 1 || A < 0) return 0;
    SegWs w;
    if (plan(N * (int64_t)K, A + N * (int64_t)K, nullptr, &w) != hipSuccess) return 0;
    return w.total;
}

extern "C" int kpgnn_csr_code_segments(const int32_t* rowptr_dst, const uint16_t* code_dst, int64_t N, int32_t K, int64_t A,
                                       uint32_t* entries, int32_t* seg_ptr, uint32_t* seg_key, int32_t* hop_seg,
                                       int32_t* counts, void* workspace, size_t workspace_bytes, kpgnn_stream_t stream) {
    KPGNN_REQUIRE(N >= 0 && K >= 1 && K <= 62 && A >= 0, "csr_code_segments: bad N=%lld K=%d A=%lld", (long long)N, K, (long long)A);
    const int64_t S = N * (int64_t)K, T = A + S;        // upper bound of the number of keys
    if (T >= ((int64_t)1 << 31)) return fail(KPGNN_ELIMIT, "csr_code_segments: A + N*K = %lld exceeds the int32 index range", (long long)T);
    KPGNN_REQUIRE(entries && seg_ptr && seg_key && hop_seg && counts, "csr_code_segments: NULL output");
    hipStream_t s = (hipStream_t)stream;
    if (S == 0) {
        KPGNN_HIP_TRY(hipMemsetAsync(counts, 0, 2 * sizeof(int32_t), s));
        KPGNN_HIP_TRY(hipMemsetAsync(seg_ptr, 0, sizeof(int32_t), s));
        KPGNN_HIP_TRY(hipMemsetAsync(hop_seg, 0, sizeof(int32_t) * (size_t)(K + 1), s));
        return KPGNN_OK;
    }
    KPGNN_REQUIRE(rowptr_dst && (A == 0 || code_dst), "csr_code_segments: NULL CSR");
    SegWs w;
    KPGNN_HIP_TRY(plan(S, T, (char*)workspace, &w));
    KPGNN_REQUIRE(workspace && workspace_bytes >= w.total, "csr_code_segments: workspace too small (%zu < %zu)", workspace_bytes, w.total);
    KPGNN_REQUIRE(((uintptr_t)workspace & 255) == 0, "csr_code_segments: workspace must be 256-byte aligned");
    const unsigned sblocks = (unsigned)((S + 1 + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(row_count_kernel, dim3(sblocks), dim3(kThreads), 0, s, rowptr_dst, code_dst, S, w.cnt, w.mincode);
    KPGNN_LAUNCH_CHECK("row_count_kernel");
    size_t tb = w.prim_bytes;
    KPGNN_HIP_TRY(rocprim::exclusive_scan(w.prim, tb, w.cnt, w.cnt, 0, (size_t)(S + 1), rocprim::plus<int32_t>(), s));
    // number of keys = cnt[S] after the scan = A + (#empty rows) <= T.  It is only known on the device; rather than
    // synchronising, the kernels below run over the upper bound T with the tail filled by an all-ones sentinel that
    // sorts last (a real key's hop field is <= 62) and is never emitted
    KPGNN_HIP_TRY(hipMemsetAsync(w.keys_a, 0xFF, sizeof(uint64_t) * (size_t)T, s));
    hipLaunchKernelGGL(row_keys_kernel, dim3(sblocks), dim3(kThreads), 0, s, rowptr_dst, code_dst, S, (int)K, w.cnt, w.keys_a);
    KPGNN_LAUNCH_CHECK("row_keys_kernel");
    tb = w.prim_bytes;
    KPGNN_HIP_TRY(rocprim::radix_sort_keys(w.prim, tb, w.keys_a, w.keys_b, (size_t)T, 0, 64, s));
    const unsigned tblocks = (unsigned)((T + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(key_flag_kernel, dim3(tblocks), dim3(kThreads), 0, s, w.keys_b, T, w.flag);
    KPGNN_LAUNCH_CHECK("key_flag_kernel");
    tb = w.prim_bytes;
    KPGNN_HIP_TRY(rocprim::exclusive_scan(w.prim, tb, w.flag, w.idx, 0, (size_t)T, rocprim::plus<int32_t>(), s));
    hipLaunchKernelGGL(entry_emit_kernel, dim3(tblocks), dim3(kThreads), 0, s, w.keys_b, w.idx, T, (int)K, w.mincode,
                       entries, w.ekey, counts);
    KPGNN_LAUNCH_CHECK("entry_emit_kernel");
    hipLaunchKernelGGL(run_start_kernel, dim3(tblocks), dim3(kThreads), 0, s, w.ekey, counts, T, w.flag);
    KPGNN_LAUNCH_CHECK("run_start_kernel");
    tb = w.prim_bytes;
    KPGNN_HIP_TRY(rocprim::inclusive_scan(w.prim, tb, w.flag, w.flag, (size_t)T, rocprim::maximum<int32_t>(), s));
    int32_t* sflag = (int32_t*)w.keys_a;              // reuse: T int32 (keys_a holds T uint64)
    int32_t* sidx = sflag + T;
    hipLaunchKernelGGL(seg_flag_kernel, dim3(tblocks), dim3(kThreads), 0, s, w.flag, counts, T, sflag);
    KPGNN_LAUNCH_CHECK("seg_flag_kernel");
    tb = w.prim_bytes;
    KPGNN_HIP_TRY(rocprim::exclusive_scan(w.prim, tb, sflag, sidx, 0, (size_t)T, rocprim::plus<int32_t>(), s));
    hipLaunchKernelGGL(seg_emit_kernel, dim3(tblocks), dim3(kThreads), 0, s, w.flag, sidx, w.ekey, counts, seg_ptr, seg_key);
    KPGNN_LAUNCH_CHECK("seg_emit_kernel");
    hipLaunchKernelGGL(hop_seg_kernel, dim3(1), dim3(64), 0, s, seg_key, counts, (int)K, hop_seg);
    KPGNN_LAUNCH_CHECK("hop_seg_kernel");
    return KPGNN_OK;
}
