// Per-step collate from a dataset-resident K-hop CSR (gfx950).  Contract: include/kpgnn.h, kpgnn_collate.
//
// The reference stores a pre-transformed dataset as PyG's (data, slices) pair (datasets/ZINC_dataset.py:139-140) and
// every training step builds a new batch from a shuffled subset of it: Batch.from_data_list on the host + `.to(device)`
// of ~120 MB of int64 indices (train_ZINC.py:224,36-40).  Feeding such a batch to kpgnn_csr_build costs two radix sorts,
// a 64-bit sort, five scans and a host round trip per step - as long as the training step itself.
//
// Here the CSR is built ONCE per dataset (graph by graph the same arrays kpgnn_csr_build emits, with node ids local to
// their graph and offsets relative to the graph's first pair) and stays in HBM.  Graphs of a batch are contiguous node
// ranges and a segment key is node * K + hop, so the batch CSR in either orientation is the CONCATENATION of the graphs'
// CSRs plus an offset - no sort.  The only list that mixes graphs is the table-gradient entry list (tiles of 8
// consecutive batch nodes straddle graph boundaries): every node's entries are kept sorted by (table, code, hop), and an
// entry's place in its tile's list is its own position plus the number of smaller entries in the tile's other nodes
// (seven short binary searches) - one thread per entry, no sort, no barrier.
//
// Launches: nodes, pairs, entries (+ 3 for the hop-prefix copies of the entry list).  No host synchronisation: the batch
// sizes (N, A, entries) are sums of per-graph counts the host already has.
#include "kpgnn_common.h"

namespace kpgnn {
namespace {

constexpr int kCT = 256;

struct Hdr {
    const int32_t *ids, *node_base, *pair_base, *ent_base;
    __host__ __device__ Hdr(const int32_t* h, int B) : ids(h), node_base(h + B), pair_base(h + 2 * B + 1), ent_base(h + 3 * B + 2) {}
};

// last g in [0, n) with base[g] <= v   (base is non-decreasing, base[0] <= v)
__device__ __forceinline__ int seg_of(const int32_t* __restrict__ base, int n, int64_t v) {
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((int64_t)base[mid] <= v) lo = mid; else hi = mid;
    }
    return lo;
}

struct RowGathers { const char* src[8]; char* dst[8]; int bytes[8]; int n; };

__device__ __forceinline__ void copy_row(const char* __restrict__ s, char* __restrict__ d, int bytes) {
    if (((bytes | (int)(uintptr_t)s | (int)(uintptr_t)d) & 3) == 0) {
        for (int b = 0; b < bytes; b += 4) *reinterpret_cast<uint32_t*>(d + b) = *reinterpret_cast<const uint32_t*>(s + b);
    } else if (((bytes | (int)(uintptr_t)s | (int)(uintptr_t)d) & 1) == 0) {
        for (int b = 0; b < bytes; b += 2) *reinterpret_cast<uint16_t*>(d + b) = *reinterpret_cast<const uint16_t*>(s + b);
    } else {
        for (int b = 0; b < bytes; ++b) d[b] = s[b];
    }
}

// one thread per batch node (and, for i < B, per batch graph)
__global__ void __launch_bounds__(kCT)
collate_nodes_kernel(const kpgnn_dataset_view ds, int B, const int32_t* __restrict__ hdr,
                     int32_t* __restrict__ rowptr_dst, int32_t* __restrict__ rowptr_src, int64_t* __restrict__ batch,
                     int32_t* __restrict__ node_src, int32_t* __restrict__ ent_node_ptr,
                     const RowGathers nodes, const RowGathers graphs) {
    const Hdr h(hdr, B);
    const int N = h.node_base[B];
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < B) {
        const int64_t gs = h.ids[i];
        for (int r = 0; r < graphs.n; ++r)
            copy_row(graphs.src[r] + gs * graphs.bytes[r], graphs.dst[r] + i * graphs.bytes[r], graphs.bytes[r]);
    }
    if (i == N) {
        const int K = ds.K;
        rowptr_dst[(int64_t)N * K] = h.pair_base[B];
        rowptr_src[(int64_t)N * K] = h.pair_base[B];
        if (ent_node_ptr) ent_node_ptr[N] = h.ent_base[B];
    }
    if (i >= N) return;
    const int g = seg_of(h.node_base, B + 1, i);
    const int64_t s = ds.node_ptr[h.ids[g]] + (i - h.node_base[g]);
    const int K = ds.K;
    const int pb = h.pair_base[g];
    for (int k = 0; k < K; ++k) {
        rowptr_dst[i * K + k] = ds.rowptr_dst[s * K + k] + pb;
        rowptr_src[i * K + k] = ds.rowptr_src[s * K + k] + pb;
    }
    batch[i] = g;
    node_src[i] = (int32_t)s;
    if (ent_node_ptr) ent_node_ptr[i] = ds.ent_rel[s] + h.ent_base[g];
    for (int r = 0; r < nodes.n; ++r)
        copy_row(nodes.src[r] + s * nodes.bytes[r], nodes.dst[r] + i * nodes.bytes[r], nodes.bytes[r]);
}

// one thread per active pair: both orientations (a graph has the same number of pairs in each)
__global__ void __launch_bounds__(kCT)
collate_pairs_kernel(const kpgnn_dataset_view ds, int B, const int32_t* __restrict__ hdr,
                     int32_t* __restrict__ col_dst, uint16_t* __restrict__ code_dst,
                     int32_t* __restrict__ col_src, uint16_t* __restrict__ code_src) {
    const Hdr h(hdr, B);
    const int64_t A = h.pair_base[B];
    const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (p >= A) return;
    const int g = seg_of(h.pair_base, B + 1, p);
    const int64_t sp = ds.pair_ptr[h.ids[g]] + (p - h.pair_base[g]);
    const int nb = h.node_base[g];
    col_dst[p] = ds.col_dst[sp] + nb;
    code_dst[p] = ds.code_dst[sp];
    col_src[p] = ds.col_src[sp] + nb;
    code_src[p] = ds.code_src[sp];
}

__device__ __forceinline__ uint32_t ent_key(uint32_t w) { return ((w >> 15) << 6) | (w & 63u); }   // (table, code, hop)

// one thread per entry: its rank inside the tile's merged list
__global__ void __launch_bounds__(kCT)
collate_tiles_kernel(const kpgnn_dataset_view ds, int B, const int32_t* __restrict__ hdr, const int64_t* __restrict__ batch,
                     const int32_t* __restrict__ ent_node_ptr, int NT, int32_t* __restrict__ tile_ptr,
                     uint32_t* __restrict__ tile_pack) {
    const Hdr h(hdr, B);
    const int N = h.node_base[B];
    const int64_t n_ent = h.ent_base[B];
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t ntiles = ((int64_t)N + NT - 1) / NT;
    if (e <= ntiles) tile_ptr[e] = ent_node_ptr[min((int64_t)N, e * NT)];
    if (e >= n_ent) return;
    const int i = seg_of(ent_node_ptr, N + 1, e);
    const int g = (int)batch[i];
    const uint32_t w = ds.ent[ds.ent_ptr[h.ids[g]] + (e - h.ent_base[g])];
    const uint32_t key = ent_key(w);
    const int t0 = (i / NT) * NT;
    int rank = (int)(e - ent_node_ptr[i]);
    for (int j = t0; j < t0 + NT && j < N; ++j) {
        if (j == i) continue;
        const int b0 = ent_node_ptr[j], len = ent_node_ptr[j + 1] - b0;
        if (len == 0) continue;
        const int gj = (int)batch[j];
        const uint32_t* __restrict__ lst = ds.ent + (ds.ent_ptr[h.ids[gj]] + (b0 - h.ent_base[gj]));
        // entries of node j that sort before this one: keys < key, and for an earlier node also keys == key
        const uint32_t bound = key + (j < i ? 1u : 0u);
        int lo = 0, hi = len;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (ent_key(lst[mid]) < bound) lo = mid + 1; else hi = mid;
        }
        rank += lo;
    }
    tile_pack[ent_node_ptr[t0] + rank] = w | ((uint32_t)(i - t0) << 12);
}

// ---------------------------------------------------------------------------------------- hop-prefix copies, all k at once
// cnt[(k-1) * ntiles + tile] = entries of the tile with hop < k, k = 1..P   (one wave per tile)
__global__ void __launch_bounds__(kWave)
prefix_count_kernel(const int32_t* __restrict__ tptr, const uint32_t* __restrict__ tpack, int P, int64_t ntiles,
                    int32_t* __restrict__ cnt, const int32_t* __restrict__ n_dyn, int NT) {
    __shared__ int hist[64];
    const int lane = threadIdx.x;
    const int64_t tl = blockIdx.x;
    if (n_dyn && tl >= ((int64_t)*n_dyn + NT - 1) / NT) {       // a tile beyond the live nodes (capacity launch): no entries
        if (lane < P) cnt[(int64_t)lane * ntiles + tl] = 0;
        return;
    }
    hist[lane] = 0;
    __syncthreads();
    const int b = tptr[tl], e = tptr[tl + 1];
    for (int i = b + lane; i < e; i += kWave) atomicAdd(&hist[tpack[i] & 63u], 1);
    __syncthreads();
    if (lane < P) {
        int s = 0;
        for (int hop = 0; hop <= lane; ++hop) s += hist[hop];
        cnt[(int64_t)lane * ntiles + tl] = s;
    }
}

// block k-1: exclusive scan of cnt[k-1][0..n) into out[k-1][0..n]
__global__ void __launch_bounds__(1024)
prefix_scan_kernel(const int32_t* __restrict__ cnt_all, int64_t n, int32_t* __restrict__ out_all) {
    __shared__ int part[1024];
    const int32_t* cnt = cnt_all + (int64_t)blockIdx.x * n;
    int32_t* out = out_all + (int64_t)blockIdx.x * (n + 1);
    const int t = threadIdx.x;
    const int64_t per = (n + 1023) / 1024;
    const int64_t b = t * per, e = min(n, b + per);
    int s = 0;
    for (int64_t i = b; i < e; ++i) s += cnt[i];
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = t ? part[t - 1] : 0;
    for (int64_t i = b; i < e; ++i) { out[i] = run; run += cnt[i]; }
    if (t == 1023) out[n] = part[1023];
}

// wave (tile, k-1): stable compaction of the entries with hop < k
__global__ void __launch_bounds__(kWave)
prefix_compact_kernel(const int32_t* __restrict__ tptr, const uint32_t* __restrict__ tpack, int64_t ntiles, int64_t pack_stride,
                      const int32_t* __restrict__ optr_all, uint32_t* __restrict__ opack_all, const int32_t* __restrict__ n_dyn, int NT) {
    const int lane = threadIdx.x;
    const int64_t tl = blockIdx.x;
    if (n_dyn && tl >= ((int64_t)*n_dyn + NT - 1) / NT) return;
    const int k = blockIdx.y + 1;
    const int32_t* optr = optr_all + (int64_t)blockIdx.y * (ntiles + 1);
    uint32_t* opack = opack_all + (int64_t)blockIdx.y * pack_stride;
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

int to_gathers(const kpgnn_row_gather* r, int n, RowGathers* out, const char* what) {
    KPGNN_REQUIRE(n >= 0 && n <= 8, "collate: at most 8 %s row gathers (got %d)", what, n);
    out->n = n;
    for (int i = 0; i < 8; ++i) { out->src[i] = nullptr; out->dst[i] = nullptr; out->bytes[i] = 0; }
    for (int i = 0; i < n; ++i) {
        KPGNN_REQUIRE(r[i].src && r[i].dst && r[i].row_bytes >= 1 && r[i].row_bytes <= 4096,
                      "collate: %s row gather %d needs src, dst and 1 <= row_bytes <= 4096", what, i);
        out->src[i] = (const char*)r[i].src; out->dst[i] = (char*)r[i].dst; out->bytes[i] = r[i].row_bytes;
    }
    return KPGNN_OK;
}

}  // namespace
}  // namespace kpgnn

using namespace kpgnn;

extern "C" int kpgnn_tile_pack_prefixes(const int32_t* tile_ptr, const uint32_t* tile_pack, int64_t num_tiles, int32_t num_prefix,
                                        int64_t pack_stride, int32_t* out_ptr, uint32_t* out_pack, int32_t* scratch,
                                        const int32_t* n_dyn, int32_t nodes_per_tile, kpgnn_stream_t stream) {
    KPGNN_REQUIRE(!n_dyn || nodes_per_tile >= 1, "tile_pack_prefixes: n_dyn needs nodes_per_tile");
    KPGNN_REQUIRE(num_tiles >= 0 && num_tiles < (1ll << 30) && num_prefix >= 0 && num_prefix <= 62 && pack_stride >= 0,
                  "tile_pack_prefixes: bad num_tiles=%lld num_prefix=%d", (long long)num_tiles, num_prefix);
    if (num_tiles == 0 || num_prefix == 0) return KPGNN_OK;
    KPGNN_REQUIRE(tile_ptr && tile_pack && out_ptr && out_pack && scratch, "tile_pack_prefixes: NULL argument");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(prefix_count_kernel, dim3((unsigned)num_tiles), dim3(kWave), 0, s, tile_ptr, tile_pack, (int)num_prefix,
                       num_tiles, scratch, n_dyn, (int)nodes_per_tile);
    KPGNN_LAUNCH_CHECK("prefix_count_kernel");
    hipLaunchKernelGGL(prefix_scan_kernel, dim3((unsigned)num_prefix), dim3(1024), 0, s, scratch, num_tiles, out_ptr);
    KPGNN_LAUNCH_CHECK("prefix_scan_kernel");
    hipLaunchKernelGGL(prefix_compact_kernel, dim3((unsigned)num_tiles, (unsigned)num_prefix), dim3(kWave), 0, s, tile_ptr,
                       tile_pack, num_tiles, pack_stride, out_ptr, out_pack, n_dyn, (int)nodes_per_tile);
    KPGNN_LAUNCH_CHECK("prefix_compact_kernel");
    return KPGNN_OK;
}

extern "C" int kpgnn_collate(const kpgnn_collate_desc* d, kpgnn_stream_t stream) {
    KPGNN_REQUIRE(d != nullptr, "collate: NULL descriptor");
    const kpgnn_dataset_view& ds = d->ds;
    KPGNN_REQUIRE(d->B >= 1 && d->N >= 0 && d->A >= 0 && d->n_ent >= 0 && ds.K >= 1 && ds.K <= 62,
                  "collate: bad B=%d N=%d A=%lld entries=%lld K=%d", d->B, d->N, (long long)d->A, (long long)d->n_ent, ds.K);
    KPGNN_REQUIRE((int64_t)d->N * ds.K < ((int64_t)1 << 31) && d->A < ((int64_t)1 << 31) && d->n_ent < ((int64_t)1 << 31),
                  "collate: the batch exceeds the int32 index range");
    KPGNN_REQUIRE(d->hdr && ds.node_ptr && ds.pair_ptr && ds.rowptr_dst && ds.rowptr_src, "collate: NULL header or dataset arrays");
    KPGNN_REQUIRE(d->A == 0 || (ds.col_dst && ds.col_src && ds.code_dst && ds.code_src), "collate: NULL dataset pair arrays");
    KPGNN_REQUIRE(d->rowptr_dst && d->rowptr_src && d->batch && d->node_src, "collate: NULL rowptr / batch / node_src output");
    KPGNN_REQUIRE(d->A == 0 || (d->col_dst && d->col_src && d->code_dst && d->code_src), "collate: NULL pair outputs");
    const bool tiles = d->tile_ptr != nullptr;
    KPGNN_REQUIRE(!tiles || (d->nodes_per_tile >= 1 && d->nodes_per_tile <= 8 && ds.ent_ptr && ds.ent_rel && d->ent_node_ptr &&
                             (d->n_ent == 0 || (ds.ent && d->tile_pack))),
                  "collate: the entry list needs 1 <= nodes_per_tile <= 8, the dataset's per-node lists, ent_node_ptr and tile_pack");
    KPGNN_REQUIRE(d->num_prefix >= 0 && d->num_prefix < ds.K && (d->num_prefix == 0 || (tiles && d->prefix_ptr && d->prefix_scratch &&
                                                                                   (d->n_ent == 0 || d->prefix_pack))),
                  "collate: hop-prefix copies need the entry list and their three buffers (num_prefix=%d)", d->num_prefix);
    RowGathers rn, rg;
    int rc = to_gathers(d->node_rows, d->n_node_rows, &rn, "node");
    if (rc != KPGNN_OK) return rc;
    rc = to_gathers(d->graph_rows, d->n_graph_rows, &rg, "graph");
    if (rc != KPGNN_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int64_t nthreads = (d->N + 1 > d->B ? d->N + 1 : d->B);
    hipLaunchKernelGGL(collate_nodes_kernel, dim3((unsigned)((nthreads + kCT - 1) / kCT)), dim3(kCT), 0, s, ds, (int)d->B, d->hdr,
                       d->rowptr_dst, d->rowptr_src, d->batch, d->node_src, tiles ? d->ent_node_ptr : nullptr, rn, rg);
    KPGNN_LAUNCH_CHECK("collate_nodes_kernel");
    if (d->A > 0) {
        hipLaunchKernelGGL(collate_pairs_kernel, dim3((unsigned)((d->A + kCT - 1) / kCT)), dim3(kCT), 0, s, ds, (int)d->B, d->hdr,
                           d->col_dst, d->code_dst, d->col_src, d->code_src);
        KPGNN_LAUNCH_CHECK("collate_pairs_kernel");
    }
    if (tiles) {
        const int64_t ntiles = ((int64_t)d->N + d->nodes_per_tile - 1) / d->nodes_per_tile;
        const int64_t nt = d->n_ent > ntiles + 1 ? d->n_ent : ntiles + 1;
        hipLaunchKernelGGL(collate_tiles_kernel, dim3((unsigned)((nt + kCT - 1) / kCT)), dim3(kCT), 0, s, ds, (int)d->B, d->hdr,
                           d->batch, d->ent_node_ptr, (int)d->nodes_per_tile, d->tile_ptr, d->tile_pack);
        KPGNN_LAUNCH_CHECK("collate_tiles_kernel");
        if (d->num_prefix > 0 && ntiles > 0)
            return kpgnn_tile_pack_prefixes(d->tile_ptr, d->tile_pack, ntiles, d->num_prefix, d->n_ent, d->prefix_ptr, d->prefix_pack,
                                            d->prefix_scratch, d->hdr + 2 * d->B, d->nodes_per_tile, stream);
    }
    return KPGNN_OK;
}
