/*
 * kpgnn.h - C ABI of libkpgnn_hip.so: the MI355X (gfx950) K-hop message-passing hot path of KP-GNN.
 *
 * This is the drop-in boundary.  The reference (JiaruiFeng/KP-GNN) is pure Python; the arithmetic of
 * this path lives in its layer files and in PyG's MessagePassing.propagate.  Each entry point below
 * names the reference code it replaces.  Signatures carry plain pointers, sizes and strides only (no
 * torch types); every pointer marked "device" is HBM memory owned by the caller; nothing is allocated,
 * freed or synchronised inside a call, so calls are HIP-graph capturable.  `stream` is a hipStream_t
 * passed as void*.  All entry points return 0 on success and a negative KPGNN_E* code on failure;
 * kpgnn_last_error() returns a thread-local message for the last failure.
 *
 * Tensor layout conventions: N nodes of the collated batch, K hop slots, D per-hop feature width.
 * Feature tensors are fp32, logical shape [N, K, D], innermost dimension contiguous, node and hop
 * strides given in ELEMENTS (so a [N,k,H] view of a history buffer or of a [N, K*dk] matrix needs no copy).
 *
 * K-hop CSR ("khop_csr"): the K-hop edge list (edge_index [2,E] int64, edge_attr [E,K] int64 where
 * attr==0 means "edge not active in this hop", reference data_utils.py:80-93) is re-laid out once per
 * batch as a CSR keyed by (node, hop): segment s = node*K + hop holds the active (edge,hop) pairs of that
 * node in ORIGINAL EDGE ORDER (stable), as parallel arrays col[] (the other endpoint, int32) and code[]
 * (the attr value = embedding row, uint16).  Two orientations are built: by destination (forward
 * aggregation) and by source (backward).
 */
#ifndef KPGNN_H_
#define KPGNN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KPGNN_ABI_VERSION 1

#define KPGNN_OK 0
#define KPGNN_EINVAL (-1)   /* bad argument (shape, stride, null pointer, limits) */
#define KPGNN_EHIP (-2)     /* a HIP runtime call or kernel launch failed */
#define KPGNN_ELIMIT (-3)   /* size exceeds an implementation limit (int32 indices, LDS) */

typedef void* kpgnn_stream_t; /* hipStream_t */

int kpgnn_abi_version(void);
const char* kpgnn_last_error(void);

/* Device facts used by the host side for launch sizing / roofline reporting (needs a GPU). */
int kpgnn_device_info(int* cu_count, int* lds_bytes_per_block, int* wavefront, char* arch, int arch_len);

/* ------------------------------------------------------------------------------------------------
 * K-hop CSR construction (device).  Replaces nothing in the reference one-to-one: the reference feeds
 * the raw edge list to PyG (`self.propagate(edge_index, ...)`, layers/KPGIN.py:100, KPGINplus.py:74,
 * KPGCN.py:110, gine.py:52) which gathers/scatters over all E*K (edge,hop) slots and multiplies the
 * inactive ones by a zero mask (message(): KPGIN.py:115-118).
 * ---------------------------------------------------------------------------------------------- */

/* Pass 1: statistics of a K-hop edge list.  stats (device, int64[8]) receives
 *   [0] A = number of active (edge,hop) pairs   [1] max attr in column 0   [2] max attr in columns 1..K-1
 *   [3] min attr over all columns               [4] min node index         [5] max node index
 * edge_index rows are `ei_stride` elements apart; edge_attr rows `attr_stride` elements apart. */
int kpgnn_csr_stats(const int64_t* edge_index, int64_t ei_stride, const int64_t* edge_attr, int64_t attr_stride,
                    int64_t E, int32_t K, int64_t* stats, kpgnn_stream_t stream);

/* Workspace (bytes, device) needed by kpgnn_csr_build for E edges, A active pairs, N nodes, K hops. */
size_t kpgnn_csr_workspace_bytes(int64_t E, int64_t A, int64_t N, int32_t K);

/* Pass 2: build both orientations.  A must be stats[0].  Outputs (device):
 *   rowptr_dst int32[N*K+1], col_dst int32[A] (= source node), code_dst uint16[A]   -- keyed by (dst,hop)
 *   rowptr_src int32[N*K+1], col_src int32[A] (= dest node),   code_src uint16[A]   -- keyed by (src,hop)
 * Entries of one segment keep the order of the input edge list (stable), which is the reference's CPU
 * summation order (index_add_ over edges in order).
 * Optional third ordering for kpgnn_table_grad (tile_ptr == NULL skips it; needs K <= 62): per tile of
 * nodes_per_tile (1..8) destination nodes, one entry per DISTINCT (node, hop, code) with its multiplicity (pairs of
 * one segment that carry the same code add the same gradient row to the same table row), sorted by
 * (tile, table, code, hop, node_in_tile):
 *   tile_ptr  int32[ceil(N/nodes_per_tile)+1]   (tile_ptr[last] = number of entries <= A)
 *   tile_pack uint32[A] (at most A entries are written) =
 *             table<<31 | code<<15 | node_in_tile<<12 | (multiplicity-1)<<6 | hop
 *   (table 0 = hop 0 -> hop1_edge_emb, table 1 = hops >= 1 -> hopk_edge_emb; codes < 2^16; multiplicity 1..64, a longer
 *   run is split every 64 pairs counted from its own first pair - the list of a tile depends on that tile's pairs only,
 *   which is what lets kpgnn_collate merge per-node lists built once per dataset into the very same list). */
int kpgnn_csr_build(const int64_t* edge_index, int64_t ei_stride, const int64_t* edge_attr, int64_t attr_stride,
                    int64_t E, int32_t K, int64_t N, int64_t A,
                    int32_t* rowptr_dst, int32_t* col_dst, uint16_t* code_dst,
                    int32_t* rowptr_src, int32_t* col_src, uint16_t* code_src,
                    int32_t nodes_per_tile, int32_t* tile_ptr, uint32_t* tile_pack,
                    void* workspace, size_t workspace_bytes, kpgnn_stream_t stream);

/* Hop-prefix copy of the table-gradient entry list: the entries of (tile_ptr, tile_pack) with hop < k, in the same
 * order.  out_ptr int32[num_tiles+1], out_pack uint32[>= tile_ptr[num_tiles]], scratch int32[num_tiles].  A layer that
 * aggregates k < K hops (models/GNNs.py:421-423) hands kpgnn_table_grad this copy: the kernel splits a tile's list
 * among its waves by position, and inactive entries would leave most waves idle.  Static per batch and k. */
int kpgnn_tile_pack_filter(const int32_t* tile_ptr, const uint32_t* tile_pack, int64_t num_tiles, int32_t k,
                           int32_t* out_ptr, uint32_t* out_pack, int32_t* scratch, kpgnn_stream_t stream);

/* All hop-prefix copies at once: for k = 1..num_prefix the entries with hop < k, in the same order (three launches in all,
 * where kpgnn_tile_pack_filter takes three per k).  out_ptr int32[num_prefix][num_tiles+1], out_pack uint32[num_prefix][
 * pack_stride] (pack_stride >= tile_ptr[num_tiles]), scratch int32[num_prefix][num_tiles]. */
int kpgnn_tile_pack_prefixes(const int32_t* tile_ptr, const uint32_t* tile_pack, int64_t num_tiles, int32_t num_prefix,
                             int64_t pack_stride, int32_t* out_ptr, uint32_t* out_pack, int32_t* scratch,
                             const int32_t* n_dyn, int32_t nodes_per_tile, kpgnn_stream_t stream);
/* (n_dyn: optional device int32[1] live NODE count - num_tiles is then a capacity and the tiles beyond
 *  ceil(*n_dyn / nodes_per_tile) count as empty; NULL: all num_tiles tiles are live) */

/* ------------------------------------------------------------------------------------------------
 * Dataset-resident K-hop CSR + per-step collate (device).  Replaces the reference's storage / batching pair:
 *   datasets/ZINC_dataset.py:139-140  `torch.save(self.collate(data_list), ...)` - PyG's (data, slices) store of the
 *                                      pre-transformed dataset (tensors concatenated over graphs + per-graph offsets), and
 *   train_ZINC.py:224,36-40            DataLoader(shuffle=True) -> Batch.from_data_list + `.to(device)` on EVERY step.
 * The dataset's CSR (both orientations) and per-node table-gradient entry lists are built once - graph by graph the
 * arrays kpgnn_csr_build emits, with node ids LOCAL to their graph and offsets RELATIVE to the graph's first pair / entry -
 * and stay in HBM.  A batch of B graphs is then their concatenation plus offsets: no sort, no host synchronisation, no
 * int64 traffic.  Result: bit-identical to kpgnn_csr_build on the PyG-collated batch of the same graphs in the same order.
 * ---------------------------------------------------------------------------------------------- */
typedef struct kpgnn_dataset_view {
    int32_t K;                  /* hops of the CSR */
    int32_t G;                  /* graphs in the dataset */
    const int64_t* node_ptr;    /* [G+1] first node of graph g in the node-level arrays (PyG slices['x']) */
    const int64_t* pair_ptr;    /* [G+1] first active (edge,hop) pair of graph g; the same offsets serve both orientations */
    const int64_t* ent_ptr;     /* [G+1] first table-gradient entry of graph g (NULL: no entry lists) */
    const int32_t* rowptr_dst;  /* [Nd*K] first pair of segment (node, hop) keyed by destination, relative to pair_ptr[g] */
    const int32_t* rowptr_src;  /* [Nd*K] the same keyed by source */
    const int32_t* col_dst;     /* [Ad] other endpoint of a pair, node id local to its graph */
    const int32_t* col_src;
    const uint16_t* code_dst;   /* [Ad] edge code of a pair */
    const uint16_t* code_src;
    const int32_t* ent_rel;     /* [Nd] first entry of a node, relative to ent_ptr[g] */
    const uint32_t* ent;        /* per-node entry lists = kpgnn_csr_build's tile_pack with nodes_per_tile = 1 */
} kpgnn_dataset_view;

typedef struct kpgnn_row_gather { const void* src; void* dst; int32_t row_bytes; } kpgnn_row_gather;

typedef struct kpgnn_collate_desc {
    kpgnn_dataset_view ds;
    int32_t B;                  /* graphs in the batch */
    int32_t N;                  /* its nodes, pairs and entries: sums of per-graph counts the host keeps.  They size the   */
    int64_t A;                  /*   LAUNCHES only - the kernels read the real counts from the header - so they may be     */
    int64_t n_ent;              /*   CAPACITIES of static buffers: one captured call then collates any batch that fits.   */
    /* device int32[4B+3]: ids[B] (dataset graph of batch slot b) | node_base[B+1] | pair_base[B+1] | ent_base[B+1], the three
     * exclusive prefix sums of the chosen graphs' node / pair / entry counts (last element = N / A / n_ent) */
    const int32_t* hdr;
    int32_t* rowptr_dst; int32_t* col_dst; uint16_t* code_dst;    /* int32[N*K+1], int32[A], uint16[A] */
    int32_t* rowptr_src; int32_t* col_src; uint16_t* code_src;
    int64_t* batch;             /* [N] batch slot of every node (PyG's Batch.batch) */
    int32_t* node_src;          /* [N] dataset node of every batch node */
    /* table-gradient entry list of the batch (tile_ptr NULL: skipped) and its hop-prefix copies for k = 1..num_prefix
     * (as kpgnn_tile_pack_prefixes lays them out, pack_stride = n_ent) */
    int32_t nodes_per_tile; int32_t* tile_ptr; uint32_t* tile_pack;
    int32_t* ent_node_ptr;      /* [N+1] scratch (first entry of every batch node) */
    int32_t num_prefix; int32_t* prefix_ptr; uint32_t* prefix_pack; int32_t* prefix_scratch;
    /* dense per-node / per-graph attributes: dst[i] = src[node_src[i]] (node rows), dst[b] = src[ids[b]] (graph rows) */
    int32_t n_node_rows; kpgnn_row_gather node_rows[8];
    int32_t n_graph_rows; kpgnn_row_gather graph_rows[8];
} kpgnn_collate_desc;

int kpgnn_collate(const kpgnn_collate_desc* d, kpgnn_stream_t stream);

/* loss[0] = mean_i |score[i] - y[i]| (kind 0; train_ZINC.py:42) or mean_i (score[i] - y[i])^2 (kind 1; train_qm9.py:96) and,
 * when dscore != NULL, dscore[i] = d loss / d score[i].  score, y: device [n] contiguous.  One launch, one block, fixed
 * summation order (bitwise reproducible); meant for batches of graph scores (n up to ~1e5). */
int kpgnn_regression_loss(const float* score, const float* y, int64_t n, int32_t kind, float* loss, float* dscore,
                          kpgnn_stream_t stream);

/* The graph regressor nn.Linear(hidden, 1) on the pooled graph rows (reference models/GraphRegression.py:17,46-51):
 * score[g] = sum_c pooled[g][c] w[c] + bias[0] (bias may be NULL).  pooled: device [G, D] contiguous fp32; w: [D]; score: [G]. */
int kpgnn_score_head_fwd(const float* pooled, const float* w, const float* bias, int64_t G, int32_t D, float* score,
                         kpgnn_stream_t stream);
/* Its backward in one launch: dpooled[g][c] = dscore[g] w[c] (NULL: skipped), dw[c] = sum_g dscore[g] pooled[g][c],
 * db[0] = sum_g dscore[g] (NULL: skipped); fixed summation order (bitwise reproducible). */
int kpgnn_score_head_bwd(const float* pooled, const float* w, const float* dscore, int64_t G, int32_t D, float* dpooled,
                         float* dw, float* db, kpgnn_stream_t stream);

/* One torch.optim.Adam step (amsgrad off, L2 weight_decay as in train_ZINC.py:244) over a flat bucket of n fp32
 * parameters with its gradient and the two moment buffers (all device, 16-B aligned, updated in place); step = 1 for the
 * first call.  Elementwise, so stepping the flat bucket of dp.py equals stepping the ~190 tensors one by one. */
int kpgnn_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t step,
                    double lr, double beta1, double beta2, double eps, double weight_decay, kpgnn_stream_t stream);

/* The same step with the step number kept on the device: state = int64[2] {steps taken so far, 0}, both zero before the
 * first call; the launch reads state[0], steps with t = state[0] + 1 and leaves state[0] = t.  Its arguments never change,
 * so it can be captured in a hipGraph together with the forward and backward passes (one process, no all-reduce between). */
int kpgnn_adam_step_device(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t* state,
                           double lr, double beta1, double beta2, double eps, double weight_decay, kpgnn_stream_t stream);

/* count contiguous fp32 tensors copied device-to-device in ceil(count / 192) launches: dst[i][0..numel[i]) = src[i][..].  The
 * pointer tables are HOST arrays read at call time and passed to the kernel by value (capturable: a hipGraph node keeps
 * them).  Used to move a step's parameter gradients into their views of the flat all-reduce bucket (train_ZINC.py:34-36
 * leaves that to DataParallel's per-tensor reduce). */
int kpgnn_multi_copy(int32_t count, const float* const* src, float* const* dst, const int64_t* numel, kpgnn_stream_t stream);

/* A reduction that a launch left for later: slab holds nslab rows of elems partial sums (block order); the finished sums
 * out[e] = sum_b slab[b][e] go, in order, to n_out[0] elements of out[0], n_out[1] of out[1], ... (sum n_out == elems).
 * kpgnn_linear_wgrad(_pair) fills one instead of launching its own reduce when desc.defer is set; the caller hands it to the
 * next call that has a finishing launch anyway (kpgnn_table_grad_desc.pending) or runs it with kpgnn_reduce_jobs.  The slab
 * (the producing call's workspace) must stay untouched until then.  Fixed summation order: the result does not depend on
 * which launch does the adding. */
typedef struct kpgnn_reduce_job {
    const float* slab; int32_t nslab; int64_t elems;
    float* out[4]; int64_t n_out[4];
} kpgnn_reduce_job;
int kpgnn_reduce_jobs(const kpgnn_reduce_job* jobs, int32_t count, kpgnn_stream_t stream);   /* ceil(count / 3) launches */

/* ------------------------------------------------------------------------------------------------
 * Fused K-hop aggregation.
 * ---------------------------------------------------------------------------------------------- */
/* Storage of the big per-(node,hop) streams.  KPGNN_STORE_BF16: the rows a launch reads or writes as marked "(storage)"
 * below hold bf16 (2 bytes per element, same shapes and ELEMENT strides); every sum, the tables, theta, P and the layer
 * outputs stay fp32.  Implemented for the KP-GIN+ training path (fused GELU + geometric combine, dictionary P, D % 4 == 0):
 * kpgnn_aggregate_fwd (x_slot rows, pre), kpgnn_combine_bwd (pre, g), kpgnn_table_grad (g), kpgnn_aggregate_bwd (g);
 * other configurations answer KPGNN_EINVAL.  (BASELINE configs[1] names bf16; tolerance 2e-2 relative) */
enum { KPGNN_STORE_F32 = 0, KPGNN_STORE_BF16 = 1 };

/* Arithmetic of the dense fp32 products that offer a choice.  AUTO: the bf16-split product where the kernel has one for the
 * shape - every fp32 operand is the exact sum of three bf16 pieces, the six leading piece products run on the bf16 matrix
 * cores with fp32 accumulation (dropped terms < 2^-24 of a product: the error of fp32 accumulation itself) - else F32.
 * F32: v_mfma_f32_32x32x2_f32 always. */
enum { KPGNN_MATH_AUTO = 0, KPGNN_MATH_F32 = 1 };

enum {
    KPGNN_MODE_GIN = 0,     /* out = S + P + (1+eps)*x                  KPGIN.py:100-105, gine.py:52-53 */
    KPGNN_MODE_GINPLUS = 1, /* out = gelu(S) + P                        KPGINplus.py:74-77,87-88        */
    KPGNN_MODE_GCN = 2,     /* out = relu(S_norm) + P, self loops + sym. degree norm  KPGCN.py:85-114,120-126 */
    KPGNN_MODE_SUM = 3      /* out = S (+P), no activation: building block / KGINConv run_simulation.py:73 */
};
/* where, for node i and hop k,  S[i,k,:] = sum over active pairs a of segment (i,k) of
 *      x[col[a], k, :] + table_k[code[a], :]          (table_0 = hop1_edge_emb, table_k>0 = hopk_edge_emb)
 * and for GCN each term is scaled by dis[col[a],k]*dis[i,k] and the self loop (code 1) is added,
 * dis = (segment length + 1)^-1/2  (KPGCN.py:11-25,106-109).  P = peripheral_attr (nullable). */

typedef struct kpgnn_agg_fwd_desc {
    int32_t N, K, D;            /* nodes, ACTIVE hop slots (k <= K_csr), per-hop width */
    int32_t K_csr;              /* hop slots per node in rowptr (GNNPlus layer l<K uses a prefix, GNNs.py:429) */
    int32_t mode;               /* KPGNN_MODE_* */
    int32_t n_code0, n_codek;   /* rows of table0 / tablek (codes are validated against these by the caller) */
    int32_t use_tables;         /* 0: mask-only aggregation (no edge-code embedding; run_simulation.py:87-90) */
    const int32_t* rowptr;      /* device, [N*K_csr+1]  (by destination) */
    const int32_t* col;         /* device, [A] */
    const uint16_t* code;       /* device, [A] */
    const float* dis;           /* device, [N*K_csr] deg^-1/2, GCN only (else NULL) */
    const float* x;             /* device, [N,K,D] */
    int64_t x_sn, x_sk;
    const float* table0;        /* device, [n_code0, D] contiguous */
    const float* tablek;        /* device, [n_codek, D] contiguous (NULL iff K_csr == 1) */
    const float* periph;        /* device, [N,K,D] or NULL */
    int64_t p_sn, p_sk;
    const float* eps;           /* device scalar (GIN) or NULL (== 0) */
    float* out;                 /* device, [N,K,D] */
    int64_t o_sn, o_sk;
    float* pre;                 /* device, [N,K,D] contiguous or NULL: S before the activation (saved for bwd) */
    /* Optional fused geometric hop-combine (combine.py:43-58): if theta != NULL, `out` is not written;
     * hout[i,:] = sum_k theta[k,:] * (act(S[i,k,:]) + P[i,k,:]) is.  theta: device [K, D]. */
    const float* theta;
    float* hout;                /* device, [N, D] contiguous */
    /* Optional constant row added to every x row of hops >= 1 (gathered and self terms): the reference's
     * `x[:, 1:] += hopk_node_path_emb(pe_attr)` (KPGIN.py:92-94) when pe_attr is all padding (always so for
     * the reference's own pre-transform, data_utils.py:91,123): xbias = that table's row 0.  device [D]. */
    const float* xbias;
    /* Optional DICTIONARY form of the peripheral features (used when periph == NULL): P[i,k,:] =
     * ptab[uid[i*uid_stride + k], :].  Peripheral-subgraph feature tuples repeat massively (25 distinct
     * rows among 379,600 (node,hop) slots of a 2048-molecule batch), so the [N,K,D] stream becomes an
     * L2-resident table.  ptab: device [U, D] contiguous; uid: device int32. */
    const float* ptab;
    const int32_t* uid;
    int64_t uid_stride;
    /* Per-hop inputs (used when x == NULL): hop slot k reads x_slot[k], a [N,D] matrix with row stride x_sn.
     * GNNPlus stacks the previous layers' states into [N,k,H] with torch.cat every layer (models/GNNs.py:413-418);
     * with slots the kernel reads the k states where they are and the copy disappears.  K <= 16. */
    const float* x_slot[16];
    /* Rows of ptab (0 = unknown).  When given and small, the dictionary and theta are staged in LDS next to the
     * code tables, so the per-hop epilogue has no dependent global loads left. */
    int32_t n_dict;
    /* Geometric combine computed by the launch itself: alphas [D] (device) - theta[k,d] = softmax_k(a (1-a)^k) with
     * a = sigmoid(alphas[d]) (combine.py:43-50) is then an OUTPUT, written to `theta` ([K,D]) for the backward. */
    const float* alphas;
    /* KPGNN_STORE_*: with BF16 the x_slot rows (storage) and pre (storage) are bf16 - pass them through these float
     * pointers; x_sn counts ELEMENTS. */
    int32_t storage;
    const int32_t* n_dyn;       /* optional live-row count (device int32[1], <= N; kpgnn_wgrad_desc explains); NULL: all N rows */
    /* Optional graph boundaries of the collated batch (device int32[num_graphs+1] node offsets, nodes of a graph contiguous as
     * PyG's collate lays them out; max_graph_nodes >= the largest graph).  With them a mask-only aggregation (no tables, no
     * peripheral features, no fused combine) whose largest graph's hop slab x[graph, hop, :] fits LDS is gathered FROM LDS: the
     * slab is staged once per (graph, hop) block and every neighbour row comes from there - the kernel for dense K-hop
     * neighbourhoods (run_simulation.py's 3-regular n = 1280 graphs: 582 pairs per node), where the plain gather is L2-bound. */
    const int32_t* graph_ptr; int32_t num_graphs; int32_t max_graph_nodes;
    /* Optional, with a fused combine and a given theta: hout[i,:] = hinit[i,:] + sum_k ... (device [N,D] contiguous; may be hout
     * itself).  This is what makes the launch the PULL form of the backward gather: with rowptr / col keyed by (source, hop),
     * x_slot[k] = hop k's slab of dL/dS of the layer k + 1 steps later, mode SUM, theta = 1, it writes a state's whole gradient
     * in one pass - every later reader's share plus what hinit already holds - instead of one read-modify-write per reader. */
    const float* hinit;
    const float* hinit2;        /* a second addend of the same kind (the pull form: the residual branch's share) or NULL */
} kpgnn_agg_fwd_desc;

int kpgnn_aggregate_fwd(const kpgnn_agg_fwd_desc* d, kpgnn_stream_t stream);

/* Backward of the aggregation w.r.t. x and the two edge-code tables, given g = dL/dS [N,K,D]
 * (the caller applies the activation derivative; for GCN g already includes relu').
 *   gx[j,k,:]      = sum over pairs a of segment (j,k) of the BY-SOURCE csr of w_a * g[col[a],k,:]  (+ self terms)
 *   gtable_k[c,:] += sum over pairs with code c of w_a * g[col[a],k,:]      (fp32 atomics; caller zeroes)
 * self terms: GIN adds (1+eps)*g[j,k,:]; GCN adds dis[j,k]^2*g[j,k,:] and the same into gtable_k[1,:]. */
typedef struct kpgnn_agg_bwd_desc {
    int32_t N, K, D, K_csr, mode, n_code0, n_codek, use_tables;
    const int32_t* rowptr_src;
    const int32_t* col_src;
    const uint16_t* code_src;
    const float* dis;           /* GCN only */
    const float* g;             /* device [N,K,D] */
    int64_t g_sn, g_sk;
    const float* eps;
    float* gx;                  /* device [N,K,D] */
    int64_t gx_sn, gx_sk;
    float* gtable0;             /* device [n_code0, D], accumulated into (NULL: skip table grads) */
    float* gtablek;             /* device [n_codek, D] */
    /* Per-hop outputs (used when gx == NULL): the gradient of hop slot k goes to gx_slot[k], [N,D] with row
     * stride gx_sn (the backward of the per-hop inputs above: no [N,k,D] tensor to slice up afterwards). */
    float* gx_slot[16];
    /* Bit k set: ADD the gradient of hop slot k to what gx_slot[k] already holds (a state that several layers read as
     * a slot collects its gradient in one buffer instead of one tensor per reader plus an add each).  Two hop slots of
     * one call must not share a buffer when either accumulates (KPGNN_EINVAL).  With gx (one [N,K,D] tensor) bit k adds hop
     * k's gradient to what gx[:, k, :] already holds: a state that the bodies' jumping-knowledge projection and the next
     * norm's residual branch also read collects its whole gradient in one buffer (K <= 32). */
    uint32_t accumulate_mask;
    int32_t storage;            /* KPGNN_STORE_*: with BF16, g (storage) holds bf16 rows; g_sn / g_sk count elements */
    const int32_t* n_dyn;       /* optional live-row count (device int32[1], <= N; kpgnn_wgrad_desc explains); NULL: all N rows */
} kpgnn_agg_bwd_desc;

int kpgnn_aggregate_bwd(const kpgnn_agg_bwd_desc* d, kpgnn_stream_t stream);

/* Table gradients WITHOUT per-edge atomics:
 *   gtable_t[c,:] = sum over active pairs (i,k) of table t with code c of g[i,k,:]      (g = dL/dS)
 * (the reference gets them from nn.Embedding's backward of the materialised [E,K,D] embedding tensor,
 * KPGIN.py:90-96), and optionally the gradient of a peripheral-feature DICTIONARY (see kpgnn_agg_fwd_desc.ptab):
 *   gdict[u,:] = sum over (i,k) with uid[i,k] == u of  theta[k,:]*gh[i,:]   (dict_src 1, fused combine)
 *                                                  or  g[i,k,:]             (dict_src 2, g is dL/dP itself)
 * A block streams tiles of `nodes_per_tile` destination nodes of g through LDS; thread t owns feature column
 * t, walks the tile's (table,code)-sorted pair list and keeps the running sum of the current code in a
 * register, so the LDS accumulators are column-private (no atomics).  Per-block partial tables go to
 * `workspace`; a second launch adds them in block order: outputs are OVERWRITTEN and bitwise reproducible.
 * K is the number of ACTIVE hops of g (pairs with hop >= K are skipped).  tile_ptr == NULL skips the
 * edge-code part (dictionary gradient only; gtable0/gtablek may then be NULL). */
typedef struct kpgnn_table_grad_desc {
    int32_t N, K, D, nodes_per_tile, n_code0, n_codek;
    int32_t n_dict, dict_src;   /* dictionary rows (0: none) and source (1: theta*gh, 2: g rows) */
    const int32_t* tile_ptr;
    const uint32_t* tile_pack;
    const float* g;             /* device [N,K,D] */
    int64_t g_sn, g_sk;
    const int32_t* uid;         /* device [N, uid_stride]: dictionary row of (node, hop) */
    int64_t uid_stride;
    const float* theta;         /* device [K, D]  (dict_src 1) */
    const float* gh;            /* device [N, D]  (dict_src 1) */
    float* gtable0;             /* device [n_code0, D] */
    float* gtablek;             /* device [n_codek, D] or NULL when K == 1 */
    float* gdict;               /* device [n_dict, D] */
    void* workspace;            /* device, >= kpgnn_table_grad_workspace_bytes(...) */
    size_t workspace_bytes;
    /* Kernel choice: 0 = automatic (narrow rows D <= 32 and K > 8 go to the count-matrix product on the matrix cores,
     * wide rows to the register walk), 1 = force the walk, 2 = force the count-matrix kernel.  The parity tests
     * compare the two. */
    int32_t kernel;
    /* Optional (walk kernel): the dictionary entries of every tile sorted by dictionary row, from kpgnn_dict_tile_pack
     * on the same uid / nodes_per_tile with dict_pack_K >= K hops (one list serves every layer: hops >= K are skipped).
     * Equal rows then leave the registers once per run, and a row belongs to one wave at a time (no float atomics:
     * bitwise reproducible).  Without it a dictionary gradient goes to the count-matrix kernel. */
    const uint32_t* dict_pack;
    int32_t dict_pack_K;
    /* Optional: a second slab [extra_nslab][extra_elems] (e.g. the one kpgnn_dict_grad left with defer_reduce) that the
     * finishing launch adds up into extra_out[extra_elems] as well - one launch instead of two. */
    const float* extra_slab;
    int32_t extra_nslab;
    int64_t extra_elems;
    float* extra_out;
    int32_t storage;            /* KPGNN_STORE_*: with BF16, g (storage) holds bf16 rows (walk kernel, D % 8 == 0) */
    /* Fused combine backward (fuse_pre != NULL; KP-GIN+ epilogue, fp32, edge tables only, K <= 8, even D <= 128): the
     * launch COMPUTES g = theta[k] * gh[i] * gelu'(S[i,k]) per tile (kpgnn_combine_bwd's arithmetic), writes it to fuse_g
     * and walks it from LDS - `g` above is not read.  fuse_gtheta [K,D] (optional) = sum_i gh[i] * (gelu(S[i,k]) + P[i,k]) with
     * P from the dictionary (fuse_ptab, fuse_uid; NULL = no P); with fuse_alphas / fuse_galphas the finishing launch also
     * differentiates theta(alphas) as kpgnn_combine_bwd does.  fuse_workspace >= kpgnn_table_grad_fuse_workspace_bytes(K, D). */
    const float* fuse_pre;      /* device [N,K,D] contiguous: S saved by the forward */
    const float* fuse_ptab;     /* device [fuse_n_dict, D] */
    const int32_t* fuse_uid;    /* device [N, fuse_uid_stride] */
    int64_t fuse_uid_stride;
    int32_t fuse_n_dict;
    float* fuse_g;              /* device, written: element (i,k,c) at i*g_sn + k*g_sk + c (even strides >= D; [N,K,D] contiguous or hop-major [K][N][D]) */
    float* fuse_gtheta;         /* device [K,D] or NULL */
    const float* fuse_alphas;   /* device [D] or NULL */
    float* fuse_galphas;        /* device [D] or NULL */
    void* fuse_workspace;
    size_t fuse_workspace_bytes;
    /* 1: the dictionary gradient is ADDED to what gdict / extra_out already hold (gdict += ...): a
     * dictionary read by every layer collects its gradient in one buffer instead of one tensor per layer for the
     * framework to sum (7 add launches per step at L = 8). */
    int32_t accumulate_dict;
    /* Optional (host pointer): one deferred reduction of an earlier call, added up by this call's finishing launch. */
    const kpgnn_reduce_job* pending;
    const int32_t* n_dyn;       /* optional live-row count (device int32[1], <= N; kpgnn_wgrad_desc explains); NULL: all N rows */
    /* Optional: an upper bound (>= 1) on the multiplicity of any single entry of tile_pack, when the caller knows it (0 = unknown).
     * Below 64 no run of equal entries was cut, i.e. every (node, hop, code) has ONE entry: the fused kernel may then take the
     * table gradients as a count-matrix product on the matrix cores (8-bit count cells, entry order irrelevant). */
    int32_t max_multiplicity;
} kpgnn_table_grad_desc;

size_t kpgnn_table_grad_workspace_bytes(int32_t N, int32_t K, int32_t D, int32_t nodes_per_tile,
                                        int32_t n_code0, int32_t n_codek, int32_t n_dict);
int kpgnn_table_grad(const kpgnn_table_grad_desc* d, kpgnn_stream_t stream);
size_t kpgnn_table_grad_fuse_workspace_bytes(int32_t K, int32_t D);
/* pack[tile*64 + j] = uid<<8 | node_in_tile<<3 | hop for the (node, hop) entries of a tile of nodes_per_tile nodes
 * (nodes_per_tile * K <= 64, K <= 8), sorted by uid; unused places hold 0xFFFFFFFF.  pack: uint32[ceil(N/npt) * 64].
 * A one-off per batch (the ids are data): the reference recomputes nothing like it - its embedding backward sorts the
 * same ids on every step (aten embedding_dense_backward under models/GNNs.py:172-179). */
int kpgnn_dict_tile_pack(const int32_t* uid, int64_t uid_stride, int32_t N, int32_t K, int32_t nodes_per_tile,
                         uint32_t* pack, kpgnn_stream_t stream);

/* Peripheral-dictionary gradient under the fused geometric combine, from gh alone (no [N,K,D] operand):
 *   gdict[u,:] = sum_k theta[k,:] * sum over nodes i with uid[i,k] == u of gh[i,:]
 * (the same sums as kpgnn_table_grad's dict_src 1, which stays as the path for shapes this kernel does not take:
 * kpgnn_dict_grad_workspace_bytes returns 0 unless K <= 8, D even and <= 128 and n_dict*K*D*4 B + 32 KB fit LDS).
 * One wave per hop, accumulator rows (u, hop) private to a wave: no atomics, bitwise reproducible.
 * Replaces the embedding_dense_backward calls under models/GNNs.py:393-400 for the dictionary form of the features. */
typedef struct kpgnn_dict_grad_desc {
    int32_t N, K, D, n_dict;
    const int32_t* uid;         /* device [N, uid_stride]: dictionary row of (node, hop), values in [0, n_dict) */
    int64_t uid_stride;
    const float* theta;         /* device [K, D] */
    const float* gh;            /* device [N, D] contiguous */
    float* gdict;               /* device [n_dict, D] (overwritten) */
    void* workspace;            /* device, >= kpgnn_dict_grad_workspace_bytes(N, K, D, n_dict) */
    size_t workspace_bytes;
    /* defer_reduce != 0: leave the per-block partial sums in workspace as float[kpgnn_dict_grad_slabs(N)][n_dict*D] and
     * do not write gdict: the caller adds them up in block order (kpgnn_table_grad's extra_slab does it in its own
     * finishing launch). */
    int32_t defer_reduce;
    /* Optional, device [K]: one designated id per hop.  Its sum is taken as (sum of all gh rows) - (the hop's other ids)
     * instead of being accumulated node by node - exact for ANY choice (out-of-range values are read as 0), and 3-4x less
     * work when it is the hop's most frequent id (one id covers 82-99.9 % of the nodes of a hop in a molecule batch). */
    const int32_t* dominant;
    const int32_t* n_dyn;       /* optional live-row count (device int32[1], <= N; kpgnn_wgrad_desc explains); NULL: all N rows */
} kpgnn_dict_grad_desc;

size_t kpgnn_dict_grad_workspace_bytes(int32_t N, int32_t K, int32_t D, int32_t n_dict);
int32_t kpgnn_dict_grad_slabs(int32_t N);
int kpgnn_dict_grad(const kpgnn_dict_grad_desc* d, kpgnn_stream_t stream);

/* The same gradient for ALL the layers of a sequential stack that read one dictionary (models/GNNs.py:393-400: every layer
 * adds the same peripheral rows), in one launch:
 *   gdict[u,:] = sum_l sum_{k < K[l]} theta[l][k,:] * sum over nodes i with uid[i,k] == u of gh[l][i,:]
 * A launch of kpgnn_dict_grad is mostly fixed-size passes over its [n_dict*K, D] accumulator table (one block per CU);
 * here the table is filled once, each layer's rows enter it already multiplied by that layer's theta, and the
 * designated-row fix and the final hop sum run once.  `dominant` (a designated id per hop, kpgnn_dict_grad_desc) is
 * required; K[l] <= 8, L <= 16, LDS: (n_dict*Kmax + 9 L) * D * 4 bytes <= 160 KB.  workspace as kpgnn_dict_grad's for Kmax. */
typedef struct kpgnn_dict_grad_multi_desc {
    int32_t N, D, n_dict, L;
    const int32_t* uid;         /* device [N, uid_stride] */
    int64_t uid_stride;
    const float* theta[16];     /* device [K[l], D] each */
    const float* gh[16];        /* device [N, D] contiguous each */
    int32_t K[16];
    float* gdict;               /* device [n_dict, D] (overwritten) */
    void* workspace;
    size_t workspace_bytes;
    const int32_t* dominant;    /* device [Kmax] */
    const int32_t* n_dyn;       /* optional live-row count */
} kpgnn_dict_grad_multi_desc;
int kpgnn_dict_grad_multi(const kpgnn_dict_grad_multi_desc* d, kpgnn_stream_t stream);

/* Backward pre-pass of the fused epilogue (elementwise, streaming):  with v = act(S) + P,
 *   gv[i,k,:] = theta[k,:] * gh[i,:]     (fused geometric combine)   or   gout[i,k,:]   (theta == NULL)
 *   g[i,k,:]  = gv * act'(S[i,k,:])                      act = gelu (GINPLUS) / relu (GCN) / identity
 *   gtheta[k,:] = sum_i gh[i,:] * v[i,k,:]               (only with theta)
 * Replaces a dozen framework elementwise kernels (broadcast mul, erf, exp, mul, add, einsum ...) by one pass:
 * reads S (`pre`) once, writes g once.  P comes dense (periph) or from a dictionary (ptab + uid). */
typedef struct kpgnn_combine_bwd_desc {
    int32_t N, K, D, mode;
    const float* pre;           /* device [N,K,D] contiguous (S saved by the forward) */
    const float* gh;            /* device [N,D] contiguous          (with theta) */
    const float* theta;         /* device [K,D] or NULL */
    const float* gout;          /* device [N,K,D] strided            (without theta) */
    int64_t go_sn, go_sk;
    const float* periph;        /* dense P [N,K,D] or NULL  (only read for gtheta) */
    int64_t p_sn, p_sk;
    const float* ptab;          /* dictionary P: [U,D] rows + uid, or NULL */
    const int32_t* uid;
    int64_t uid_stride;
    float* g;                   /* device [N,K,D] contiguous */
    float* gv;                  /* device [N,K,D] contiguous or NULL: dL/dP when a dense P needs its gradient */
    float* gtheta;              /* device [K,D] (overwritten) or NULL */
    void* workspace;            /* device, >= kpgnn_combine_bwd_workspace_bytes(N,K,D) when gtheta != NULL */
    size_t workspace_bytes;
    int32_t n_dict;             /* rows of ptab (0 = unknown): small dictionaries are staged in LDS */
    /* Geometric combine (theta = softmax_k(a (1-a)^k), a = sigmoid(alphas), layers/combine.py:43-50): with both given the
     * launch that adds up gtheta also writes galphas[D] = (d theta / d alphas)^T gtheta (kpgnn_geo_theta_bwd's result). */
    const float* alphas;        /* device [D] or NULL */
    float* galphas;             /* device [D] or NULL */
    int32_t storage;            /* KPGNN_STORE_*: with BF16, pre (storage) and g (storage) hold bf16 rows */
} kpgnn_combine_bwd_desc;

size_t kpgnn_combine_bwd_workspace_bytes(int32_t N, int32_t K, int32_t D);
int kpgnn_combine_bwd(const kpgnn_combine_bwd_desc* d, kpgnn_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Multi-table gather-sum: out[m,:] = bias + sum_c table[col_offset[c] + idx[m,c], :].
 * This is the reference's peripheral feature build (models/GNNs.py:172-179 / :393-400 / :637-644 via
 * FeatureConcatEncoder, layers/feature_encoder.py:62-67): Linear(cat_c Emb_c[idx_c]) equals a sum of rows
 * of the PROJECTED tables Emb_c @ W_c^T, so the [N,K,T,2H] concat, the Linear over it and the (sort-based)
 * embedding backward disappear.  m runs over (node,hop), C = 2*max_edge_type + max_hop_num + 1 columns.
 * The caller validates idx against the table sizes.  Tables are staged in LDS (column-split when large).
 * ---------------------------------------------------------------------------------------------- */
typedef struct kpgnn_tgs_desc {
    int64_t M;                  /* rows */
    int32_t C, D, R;            /* index columns, feature width, total table rows */
    const uint16_t* idx;        /* device [M, C] */
    const int32_t* col_offset;  /* device [C]: first row of column c's table inside `table` */
    const float* table;         /* device [R, D] contiguous (forward) */
    const float* bias;          /* device [D] or NULL (forward) */
    float* out;                 /* device [M, D], row stride out_stride (forward) */
    int64_t out_stride;
    const float* gout;          /* device [M, D], row stride gout_stride (backward) */
    int64_t gout_stride;
    float* gtable;              /* device [R, D] contiguous, OVERWRITTEN (backward; no atomics: bitwise reproducible) */
    void* workspace; size_t workspace_bytes;   /* backward: >= kpgnn_table_gather_sum_bwd_workspace_bytes(M, D, R) */
    const int32_t* n_dyn;       /* optional live-row count (device int32[1], <= N; kpgnn_wgrad_desc explains); NULL: all N rows */
} kpgnn_tgs_desc;

int kpgnn_table_gather_sum_fwd(const kpgnn_tgs_desc* d, kpgnn_stream_t stream);
int kpgnn_table_gather_sum_bwd(const kpgnn_tgs_desc* d, kpgnn_stream_t stream);
size_t kpgnn_table_gather_sum_bwd_workspace_bytes(int64_t M, int32_t D, int32_t R);

/* ------------------------------------------------------------------------------------------------
 * Training-mode BatchNorm1d (+ optional ReLU, + optional residual) over [N, C] rows, forward and backward.  These
 * are the nn.BatchNorm1d calls of KPGINPlusConv.mlp (KPGINplus.py:25-30), GINEConv.mlp (gine.py:31-38) and of
 * the bodies' per-layer norm (models/GNNs.py:104,431): at C ~ 100 and N ~ 50k the framework's kernels take
 * 100-140 us per pass; these stream the tensor at HBM speed around a column-statistics slot (above):
 *   fwd:  [stats pass: sum x, sum x^2 -> stat_slot, skipped when stats_ready - the producer of x filled the slot]
 *         apply pass: mean, invstd (biased var, eps) from the slot, z = [relu]( (x-mean)*invstd*gamma + beta )
 *         (+ residual); running stats updated in place with `momentum` and the unbiased variance, exactly as
 *         nn.BatchNorm1d; with out_slot the statistics of z are accumulated for a following BatchNorm.
 *   bwd:  dy = dz * [pre-activation > 0 if relu];  reduce pass: dbeta = sum dy, dgamma = sum dy*xhat -> stat_slot
 *         (reduce_only stops here: a fused consumer applies them, kpgnn_linear_bn pro 2);
 *         apply pass: dx = gamma*invstd*(dy - dbeta/N - xhat*dgamma/N).
 * Every slot must be all zero when the call is issued and is left dirty.  C <= 256.
 * ---------------------------------------------------------------------------------------------- */
typedef struct kpgnn_bn_desc {
    int64_t N;
    int32_t C, relu;
    float eps, momentum;
    const float* x;  int64_t x_stride;      /* device [N,C], row stride in elements */
    const float* gamma; const float* beta;  /* device [C] */
    float* running_mean; float* running_var;/* device [C], updated in place, or NULL */
    float* mean; float* invstd;             /* device [C] outputs (saved for backward) */
    float* z;  int64_t z_stride;            /* device [N,C] output */
    const float* residual; int64_t r_stride;/* optional: z += residual (after the activation) */
    double* stat_slot;                      /* device, kpgnn_stat_slot_bytes(C): statistics of x */
    int32_t stats_ready;                    /* 1: stat_slot already holds sum x / sum x^2 (no stats pass) */
    double* out_slot;                       /* optional (zeroed) slot receiving the statistics of z, or NULL */
    int64_t* num_batches_tracked;           /* device scalar, += 1 per call, or NULL (nn.BatchNorm1d's counter) */
    /* Optional SECOND BatchNorm applied to the result by the same call (the layers' MLP-tail norm followed by the bodies'
     * per-layer norm, KPGINplus.py:25-30 + models/GNNs.py:440-441):
     *   z = bn_outer([relu](bn(x))) + residual
     * with stats_ready = 1 and out_slot a zeroed slot (it receives the statistics of the intermediate).  Two launches: a
     * statistics-only pass over x and one apply pass - the intermediate is never written (one [N,C] store and load less
     * than two kpgnn_bn_fwd calls).  outer_mean / outer_invstd [C] are outputs like mean / invstd; the outer running
     * statistics are updated like the inner ones. */
    const float* outer_gamma; const float* outer_beta;   /* NULL: no second norm */
    float outer_eps, outer_momentum;
    float* outer_running_mean; float* outer_running_var; int64_t* outer_num_batches_tracked;
    float* outer_mean; float* outer_invstd;
    const int32_t* n_dyn;       /* optional live-row count (device int32[1], <= N; kpgnn_wgrad_desc explains); NULL: all N rows */
} kpgnn_bn_desc;

typedef struct kpgnn_bn_bwd_desc {
    int64_t N;
    int32_t C, relu;
    const float* x;  int64_t x_stride;
    const float* dz; int64_t dz_stride;
    const float* gamma; const float* beta; const float* mean; const float* invstd;
    float* dx; int64_t dx_stride;           /* NULL with reduce_only */
    float* dgamma; float* dbeta;            /* device [C] (overwritten); NULL with reduce_only */
    double* stat_slot;                      /* device, kpgnn_stat_slot_bytes(C), zero on entry */
    int32_t reduce_only;
    /* Optional, for z = bn(x) + residual: the residual branch's gradient (= dz) is ADDED in place to this [N,C]
     * buffer by the apply pass (a state read by several layers collects its gradient in one buffer). */
    float* residual_grad; int64_t rg_stride;
    /* Optional, with reduce_only: the STACKED reduce for  x -> z = [relu](bn(x)) -> h = bn_outer(z) (+ residual)  with dz
     * the gradient of h.  outer_mean / outer_invstd [C] are the outer norm's batch statistics (of z); stat_slot is then
     * 4 * kpgnn_stat_slot_bytes(C) (eight column sums, see bn.hip) and is consumed by kpgnn_linear_bn with pro = 3, which
     * applies both norms' backward while it loads its tile.  residual_grad (if set) still receives += dz. */
    const float* outer_mean; const float* outer_invstd;
    const int32_t* n_dyn;       /* optional live-row count (device int32[1], <= N; kpgnn_wgrad_desc explains); NULL: all N rows */
} kpgnn_bn_bwd_desc;

int kpgnn_bn_fwd(const kpgnn_bn_desc* d, kpgnn_stream_t stream);
int kpgnn_bn_bwd(const kpgnn_bn_bwd_desc* d, kpgnn_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Column-statistics slots.  BatchNorm needs batch-wide column sums; instead of a reduction kernel between the
 * producer and the consumer of a tensor, the producing kernel's blocks add their partial sums (fp64 atomics) to a
 * slot and the consuming kernel finishes mean / invstd (or the backward's dbeta / dgamma) from it in its prologue.
 *   slot = double[KPGNN_STAT_REPLICAS][2][C]  (block b adds to replica b % KPGNN_STAT_REPLICAS: same-address
 *   atomics serialise on this hardware), ALL ZERO before the producing launch; the calls leave it dirty - the
 *   caller zeroes its slots in bulk (one hipMemsetAsync per training step over an arena of slots).
 * Per block the sums are formed in a fixed order; only the order of the <= grid/8 fp64 adds per address varies
 * between runs (below 2^-52 relative: invisible after rounding to fp32 except on exact ties).
 * ---------------------------------------------------------------------------------------------- */
#define KPGNN_STAT_REPLICAS 8
size_t kpgnn_stat_slot_bytes(int32_t C);   /* sizeof(double) * KPGNN_STAT_REPLICAS * 2 * C */

/* ------------------------------------------------------------------------------------------------
 * Weight / bias gradient of y = x W^T + b for tall-skinny activations (N ~ 50k rows, <= 256 features):
 *   dW[o,i] = sum_r dy[r,o] * x[r,i]      db[o] = sum_r dy[r,o]
 * i.e. the backward of the nn.Linear layers behind the aggregation (KPGINplus.py:25-30, gine.py:31-38,
 * KPGIN.py:54,112).  The BLAS library's kernel for this K = N reduction runs at ~7 TFLOP/s (139 us);
 * here each wave streams row PAIRS straight from HBM into v_mfma_f32_32x32x2_f32 (exact fp32: A = dy^T
 * fragment, B = x fragment are plain coalesced row reads, no LDS), keeps a 32 x I strip of dW in
 * accumulators, and per-block partials are added in block order (deterministic).  O, I <= 256.
 * ---------------------------------------------------------------------------------------------- */
typedef struct kpgnn_wgrad_desc {
    int64_t N;
    int32_t O, I;               /* out / in features */
    const float* dy; int64_t dy_stride;   /* device [N,O] */
    const float* x;  int64_t x_stride;    /* device [N,I] */
    float* dw;                  /* device [O,I] contiguous (overwritten) */
    float* db;                  /* device [O] (overwritten) or NULL */
    void* workspace; size_t workspace_bytes;  /* >= kpgnn_wgrad_workspace_bytes(O, I) */
    /* Optional transform of x on load: x' = [relu]((x - x_mean) * x_invstd * x_gamma + x_beta), i.e. the Linear's real
     * input when that was a BatchNorm(+ReLU) output the forward never materialised (kpgnn_linear_bn pro 1). */
    const float* x_mean; const float* x_invstd; const float* x_gamma; const float* x_beta;   /* device [I] or all NULL */
    int32_t x_relu;
    /* Optional (host pointer): leave the per-block partials in `workspace` and describe their reduction here instead of
     * launching it (kpgnn_reduce_job above; for kpgnn_linear_wgrad_pair: a->defer, one job for all four outputs).  dw / db
     * are NOT written until the job has run. */
    kpgnn_reduce_job* defer;
    /* Optional ReLU mask of dy on load (device [N,O], rows dy_stride apart): dy' = dy where dy_mask > 0, else 0 - the gradient
     * behind a ReLU whose OUTPUT was saved (the jumping-knowledge projection), without materialising the masked gradient. */
    const float* dy_mask;
    /* Optional (device int32[1]): the number of LIVE rows, <= N.  A launch captured in a hipGraph for a batch CAPACITY of N
     * rows then serves batches of any size up to it (kpgnn_collate leaves the count in its header): rows >= *n_dyn are
     * neither read nor summed.  NULL: all N rows. */
    const int32_t* n_dyn;
    int32_t math;               /* KPGNN_MATH_* (bf16-split product: O <= 128, I <= 128, 16-B aligned operands) */
} kpgnn_wgrad_desc;

size_t kpgnn_wgrad_workspace_bytes(int32_t O, int32_t I);
int kpgnn_linear_wgrad(const kpgnn_wgrad_desc* d, kpgnn_stream_t stream);
/* Two weight gradients with the same (O, I) in ONE launch + one ordered reduction (the two Linears of a
 * Linear-BatchNorm-ReLU x2 MLP): workspace >= 2 * kpgnn_wgrad_workspace_bytes(O, I), taken from a. */
int kpgnn_linear_wgrad_pair(const kpgnn_wgrad_desc* a, const kpgnn_wgrad_desc* b, kpgnn_stream_t stream);
/* Grouped-K weight gradient: the Linear's input was the CONCATENATION of `group` (<= 16) states x_l [N,I] that live in
 * separate tensors (the bodies' jumping-knowledge projection, models/GNNs.py:216-218 / :455-457: cat(h_list) -> Linear):
 * dw [O, group*I] with dw[:, l*I:(l+1)*I] = dy^T x_l, db = sum dy.  d->x is ignored (x_group is a HOST array of device
 * pointers, rows d->x_stride apart); one launch (the column blocks run side by side and share dy) + one ordered reduction.
 * workspace >= kpgnn_wgrad_group_workspace_bytes(O, I, group).  Needs 16-B aligned operands, O % 4 == I % 4 == 0, O <= 128. */
size_t kpgnn_wgrad_group_workspace_bytes(int32_t O, int32_t I, int32_t group);
int kpgnn_linear_wgrad_group(const kpgnn_wgrad_desc* d, const float* const* x_group, int32_t group, kpgnn_stream_t stream);

/* y = x W^T (+ b) for tall-skinny x ([N, I], N ~ 50k, I, O <= 256) on the fp32 matrix cores: the nn.Linear
 * forward of the layers' MLPs, and - called with W^T - their input gradient dx = dy W.  Each wave keeps its
 * 32 x I strip of W in registers as MFMA A-fragments for the whole launch; 32-row tiles of x are staged in LDS
 * (padded pitch: conflict-free transposed reads) and stream through v_mfma_f32_32x32x2_f32; x is read once
 * and y written once (the BLAS library's kernel for this shape takes 24 us, ~3x the HBM time). */
typedef struct kpgnn_linear_desc {
    int64_t N;
    int32_t O, I;
    const float* x; int64_t x_stride;     /* device [N,I] */
    const float* w;                       /* device [O,I] contiguous */
    const float* bias;                    /* device [O] or NULL */
    float* y; int64_t y_stride;           /* device [N,O] */
    int32_t w_transposed;                 /* 1: w is [I,O] (y = x w): dx = dy W needs no transposed copy of W */
    /* Optional blocked output (O > 128 only): output column o is written to block o / y_block_cols, column
     * o % y_block_cols, i.e. y is [O / y_block_cols][N][y_block_cols] with y_stride = y_block_cols and the blocks
     * y_block_stride floats apart.  The input gradient of a jumping-knowledge projection (models/GNNs.py:216-218) then
     * comes out as one contiguous [N,H] matrix per layer state instead of [N, S*H] column slices.  0: plain [N,O]. */
    int32_t y_block_cols; int64_t y_block_stride;
    /* Optional (O > 128 only) ReLU mask of x on load (device [N,I], rows x_stride apart): x' = x where x_mask > 0, else 0. */
    const float* x_mask;
    const int32_t* n_dyn;                 /* optional live-row count (device int32[1], <= N), as in kpgnn_wgrad_desc */
    int32_t math;                         /* KPGNN_MATH_* (bf16-split product: blocked output with y_block_cols <= 128, no bias, N >= 4096) */
    /* Optional device scratch for the bf16-split product (the split copy of w in matrix-fragment order), 16-B aligned,
     * >= kpgnn_linear_split_workspace_bytes(y_block_cols, I, O / y_block_cols).  NULL: the fp32 kernels. */
    void* workspace; size_t workspace_bytes;
} kpgnn_linear_desc;

int kpgnn_linear_fwd(const kpgnn_linear_desc* d, kpgnn_stream_t stream);
/* Bytes of the split copy of `group` weight blocks of [O, I] (O, I <= 128) for the bf16-split kernels; 0 for shapes they do not take. */
size_t kpgnn_linear_split_workspace_bytes(int32_t O, int32_t I, int32_t group);
/* The split copies of up to 64 weight matrices in ONE launch (a body prepares all its MLPs' Linears, both orientations, at the
 * start of a step; kpgnn_linear_bn then takes them with w_split_ready = 1 instead of splitting per call - 34 five-microsecond
 * launches per step otherwise).  Element (k, n) of job i at w[n * wn + k * wk], n < O, k < I; frag: device,
 * >= kpgnn_linear_split_workspace_bytes(O, I, 1), 16-B aligned. */
typedef struct kpgnn_split_job { const float* w; int64_t wn, wk; int32_t O, I; void* frag; } kpgnn_split_job;
int kpgnn_linear_split_many(const kpgnn_split_job* jobs, int32_t n, kpgnn_stream_t stream);

/* y = act(sum_l x_l W[:, l*I:(l+1)*I]^T + b): nn.Linear over the concatenation of `group` (<= 16) states x_l [N,I] that live
 * in SEPARATE tensors - the bodies' jumping-knowledge projection `output_proj(torch.cat(h_list, dim=-1))`
 * (models/GNNs.py:216-218, :455-457, :703-705; ReLU from the Sequential at :58-59) without the concatenated copy: the
 * GEMM's K-loop runs over the state pointers.  w: device [O, group*I] contiguous; y: device [N,O] contiguous.
 * O <= 128, O % 4 == 0, I in {32, 64, 96, 104, 128}, 16-B aligned operands. */
typedef struct kpgnn_linear_group_desc {
    int64_t N;
    int32_t O, I, group;
    const float* x[16]; int64_t x_stride;
    const float* w; const float* bias; float* y;
    int32_t relu;
    const int32_t* n_dyn;                 /* optional live-row count (device int32[1], <= N) */
    int32_t math;                         /* KPGNN_MATH_* (bf16-split product: N >= 4096) */
    void* workspace; size_t workspace_bytes;  /* optional, >= kpgnn_linear_split_workspace_bytes(O, I, group); NULL: the fp32 kernel */
} kpgnn_linear_group_desc;
int kpgnn_linear_group_fwd(const kpgnn_linear_group_desc* d, kpgnn_stream_t stream);

/* The same GEMM with BatchNorm work folded into the tile's load and store phases (csrc/lin_fused.h), so that the
 * Linear-BatchNorm-ReLU x2 MLP (KPGINplus.py:25-30, gine.py:31-38) is 3 launches forward and 5 backward instead of
 * 8 + 14, with 6 fewer passes over [N,H]:
 *   pro 0: x as is.
 *   pro 1: x' = [pro_relu]((x - mean)*invstd*in_gamma + in_beta) applied while the tile is loaded; mean / invstd are
 *          finished from in_slot (sum x, sum x^2 left there by the launch that produced x, epi 1); the launch writes
 *          in_mean / in_invstd and updates running_mean / running_var / num_batches_tracked like kpgnn_bn_fwd.
 *   pro 2: BatchNorm BACKWARD on load: x is dz, x2 the BatchNorm's forward input, in_mean / in_invstd its saved
 *          statistics, in_slot holds (sum dzm, sum dzm*xhat) (kpgnn_bn_bwd reduce_only, or epi 2 of the previous
 *          launch); the tile becomes dy = gamma*invstd*(dzm - s0/N - xhat*s1/N) with dzm = dz * [bn(x2) > 0 if
 *          pro_relu]; dy is also written to xt (the weight-gradient kernel reads it); dgamma / dbeta are written.
 *   pro 3: TWO stacked BatchNorms' backward on load, for  x2 -> z = [pro_relu](bn_in(x2)) -> h = bn_o(z) (+ residual):
 *          x is dh, in_slot the eight sums of kpgnn_bn_bwd's stacked reduce (outer_mean set), in_* the inner norm, o_mean /
 *          o_invstd / o_gamma the outer one; dz is formed in registers (never stored), the tile becomes the inner norm's
 *          dy as in pro 2 (also written to xt); dgamma / dbeta and o_dgamma / o_dbeta are written.  Only with epi 2.
 *   epi 0: y stored as is.
 *   epi 1: + (sum y, sum y^2) per column into out_slot.
 *   epi 2: y is masked by the ReLU of the BatchNorm whose input was e_x (y = 0 where bn_e(e_x) <= 0) and
 *          (sum y, sum y*xhat_e) go to out_slot: the backward reduce of that BatchNorm.
 * x, x2, xt, e_x, y contiguous and 16-B aligned; I in {32, 64, 96, 104, 128}; O % 4 == 0, O <= 128.
 * KPGNN_ELIMIT when the shape is not covered.  Slots: zero on entry (out_slot), dirty on return. */
typedef struct kpgnn_linear_bn_desc {
    int64_t N;
    int32_t O, I;
    const float* x; const float* w; const float* bias; float* y;
    int32_t w_transposed;
    int32_t pro, epi, pro_relu;
    const double* in_slot;
    const float* in_gamma; const float* in_beta;
    float in_eps, momentum;
    float* in_mean; float* in_invstd;
    float* running_mean; float* running_var; int64_t* num_batches_tracked;
    const float* x2; float* xt; float* dgamma; float* dbeta;
    double* out_slot;
    const float* e_x; const float* e_mean; const float* e_invstd; const float* e_gamma; const float* e_beta;
    const float* o_mean; const float* o_invstd; const float* o_gamma; float* o_dgamma; float* o_dbeta;   /* pro 3 */
    const int32_t* n_dyn;       /* optional live-row count (device int32[1], <= N; kpgnn_wgrad_desc explains); NULL: all N rows */
    int32_t math;               /* KPGNN_MATH_* (bf16-split product: N >= 4096, I <= 104, a workspace; pro >= 2 then needs xt) */
    void* workspace; size_t workspace_bytes;   /* optional, 16-B aligned, >= kpgnn_linear_split_workspace_bytes(O, I, 1); NULL: fp32 kernel */
    int32_t w_split_ready;      /* 1: workspace already holds the split copy of w in this orientation (kpgnn_linear_split_many) */
} kpgnn_linear_bn_desc;

int kpgnn_linear_bn(const kpgnn_linear_bn_desc* d, kpgnn_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Graph readout over a collated batch (models/GraphRegression.py:46-51: PyG global_add_pool / global_mean_pool):
 *   out[g,:] = sum (mode 0) or mean (mode 1) of x[n,:] over the nodes n of graph g;  gx[n,:] = gout[batch[n],:] (/ count)
 * The nodes of a graph are contiguous (graph_ptr[g] .. graph_ptr[g+1]), rows are added in node order: bitwise
 * reproducible, unlike a scatter with fp32 atomics.  One launch per direction.
 * ---------------------------------------------------------------------------------------------- */
typedef struct kpgnn_pool_desc {
    int64_t N;                  /* nodes */
    int32_t G, D, mode;         /* graphs, feature width, 0 = sum / 1 = mean */
    const int32_t* graph_ptr;   /* device [G+1], non-decreasing, graph_ptr[G] == N */
    const int64_t* batch;       /* device [N]: graph of node n (backward only) */
    const float* x; int64_t x_stride;      /* device [N,D] (forward) */
    float* out;                 /* device [G,D] contiguous (forward) */
    const float* gout;          /* device [G,D] contiguous (backward) */
    float* gx; int64_t gx_stride;          /* device [N,D] (backward, overwritten) */
    const int32_t* n_dyn;       /* optional live-row count (device int32[1], <= N; kpgnn_wgrad_desc explains); NULL: all N rows */
} kpgnn_pool_desc;

int kpgnn_segment_pool_fwd(const kpgnn_pool_desc* d, kpgnn_stream_t stream);
int kpgnn_segment_pool_bwd(const kpgnn_pool_desc* d, kpgnn_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Projected peripheral-feature tables (csrc/enc_tables.hip): for every encoder e (a FeatureConcatEncoder: per-column
 * nn.Embedding -> concat -> Linear, layers/feature_encoder.py:37-67, gated by squash(pew / pcw), models/GNNs.py:172-179 /
 * :393-400 / :637-644) and every component c of it
 *     table[rows of c, :] = squash(gate_e) * Emb_c.weight @ W_e[:, cH:(c+1)H]^T        bias = sum_e squash(gate_e)*mult_e*b_e
 * so that the features are one gather-sum over `table` (kpgnn_table_gather_sum).  One launch per direction instead of
 * ~14 + ~28 framework launches on tiny tensors.  Components are listed encoder by encoder; H <= 256; <= 16 components,
 * <= 4 encoders.  enc_squash: 0 sigmoid (GNN, GNNPrime), 1 tanh (GNNPlus).  enc_mult: how many index columns share the
 * encoder's bias (T edge types for the edge encoder, 1 for the configuration encoder).
 * ---------------------------------------------------------------------------------------------- */
typedef struct kpgnn_enc_tables_desc {
    int32_t H, num_components, num_encoders;
    const float* comp_emb[16]; int32_t comp_rows[16]; int32_t comp_encoder[16];
    const float* enc_w[4]; const float* enc_b[4]; const float* enc_gate[4]; float enc_mult[4]; int32_t enc_squash[4];
    float* table; float* pre; float* bias;       /* forward outputs: [R,H], [R,H] (un-gated, kept for backward), [H] */
    const float* gtable; const float* gbias;     /* backward inputs (plus pre) */
    float* comp_gemb[16];                        /* backward outputs: like comp_emb */
    float* enc_gw[4]; float* enc_gb[4]; float* enc_ggate[4];
} kpgnn_enc_tables_desc;

int kpgnn_enc_tables_fwd(const kpgnn_enc_tables_desc* d, kpgnn_stream_t stream);
int kpgnn_enc_tables_bwd(const kpgnn_enc_tables_desc* d, kpgnn_stream_t stream);

/* Identifier of the capture the stream is part of (hipStreamGetCaptureInfo), 0 when it is not capturing: lets a
 * caller that zeroes its statistics slots once per step put that memset into every graph it captures. */
int kpgnn_stream_capture_id(kpgnn_stream_t stream, uint64_t* id);

/* ------------------------------------------------------------------------------------------------
 * Attention hop-combine (reference layers/combine.py:8-27): a 1-layer bidirectional LSTM with hidden size K
 * over the K hop slots scores each slot, softmax over slots, weighted sum of the slots:
 *     score[n,t] = sum_c ( h_fwd[n,t,c] + h_bwd[n,t,c] ),  w = softmax_t(score),  out[n,:] = sum_t w[n,t] x[n,t,:]
 * The input projection  gin = x W_ih^T + b_ih + b_hh  ([N*K, D] x [D, 8K], both directions side by side) is a plain
 * GEMM and stays on the matrix-core library; these entry points do the rest:
 *   fwd: the K-step recurrence, one thread per (node, direction), W_hh wave-uniform, state in registers; then
 *        softmax + weighted sum.  Saves the gate activations and cell states (5K floats per step) for backward.
 *   bwd: d(out) -> dw, dx (direct part), softmax backward, BPTT in registers -> dgin [N*K, 8K] and h_prev [N*K,2,K]
 *        (dW_ih, db, dW_hh are then weight-gradient GEMMs of dgin: kpgnn_linear_wgrad; dx += dgin W_ih).
 * K <= 16.
 * ---------------------------------------------------------------------------------------------- */
typedef struct kpgnn_attn_desc {
    int32_t N, K, D;
    const float* x; int64_t x_sn, x_sk;     /* device [N,K,D] */
    const float* gin;                       /* device [N,K,2,4K] contiguous: input projections incl. both biases */
    const float* whh;                       /* device [2,4K,K] contiguous: weight_hh_l0, weight_hh_l0_reverse */
    float* acts;                            /* device [2,N,K,5K]: i,f,g,o (activated), c per step (fwd: out, bwd: in) */
    float* hsum;                            /* device [2,N,K] workspace: sum_c h */
    float* w;                               /* device [N,K] softmax weights (fwd: out, bwd: in) */
    float* out;                             /* device [N,D] (fwd) */
    /* backward only */
    const float* gout;                      /* device [N,D] */
    float* dx;                              /* device [N,K,D] contiguous: receives the DIRECT part w[n,t]*gout[n,:] */
    float* ds;                              /* device [N,K] workspace: d(score) */
    float* dgin;                            /* device [N,K,2,4K] */
    float* hprev;                           /* device [N,K,2,K]: h_{t-1} of each step, for dW_hh */
} kpgnn_attn_desc;

int kpgnn_attn_fwd(const kpgnn_attn_desc* d, kpgnn_stream_t stream);
int kpgnn_attn_bwd(const kpgnn_attn_desc* d, kpgnn_stream_t stream);

/* The same operator for K <= 8 and D % 4 == 0, D <= 128 with the input projection inside (no gin tensor, no library
 * GEMM in the forward): a wave owns 32 nodes of one direction, takes the projection W_ih x_t and the recurrent
 * product W_hh h_{t-1} transposed (gates x nodes) on the fp32 matrix instruction - the accumulator layout then IS
 * "the four gates of four hidden units of my node" - and runs the recurrence / BPTT on its accumulators
 * (csrc/attention.hip, "scan form").  Takes the nn.LSTM parameters as they are ([4K,D], [4K,K], [4K] per direction;
 * index 0 = forward, 1 = reverse; reference layers/combine.py:17).
 *   fwd: hsum, w, out as kpgnn_attn_fwd; acts in the kernel's own lane-contiguous layout; w_pad [64, D] = the two
 *        W_ih with the hidden size padded to 8 (row dir*32 + gate*8 + unit, zero rows for unit >= K).
 *   bwd: ds as kpgnn_attn_bwd; dgin [N*K, 64] in the padded layout (column dir*32 + gate*8 + unit, zeros in the padding);
 *        dwhh_pad [2,32,8] = the recurrent weight gradient sum dg h_prev^T, taken inside the walk (the one product that contracts
 *        over nodes: operands transposed through LDS per wave); then the COMPLETE dx[n,t,:] = w[n,t] gout[n,:] +
 *        dgin[(n,t),:] w_pad in one launch (w_pad from the forward).  dW_pad = dgin^T x and db_pad = sum dgin are a plain
 *        weight-gradient product (kpgnn_linear_wgrad); kpgnn_attn_scan_unpad drops the padding:
 *        dw [2,4K,D], db [2,4K], dwhh [2,4K,K] from dw_pad [64,D], db_pad [64], dwhh_pad [2,32,8]. */
typedef struct kpgnn_attn_scan_desc {
    int32_t N, K, D;
    const float* x; int64_t x_sn, x_sk;     /* device [N,K,D], 16-byte aligned, strides multiples of 4 */
    const float* w_ih[2];                   /* device [4K,D] weight_ih_l0, weight_ih_l0_reverse */
    const float* w_hh[2];                   /* device [4K,K] */
    const float* b_ih[2];                   /* device [4K] */
    const float* b_hh[2];                   /* device [4K] */
    float* acts;                            /* device [ceil(N/32)*2*K*20*64] (fwd: out, bwd: in) */
    float* hsum;                            /* device [2,N,K] workspace */
    float* w;                               /* device [N,K] softmax weights (fwd: out, bwd: in) */
    float* out;                             /* device [N,D] (fwd) */
    float* w_pad;                           /* device [64,D] (fwd: out, bwd: in) */
    /* backward only */
    const float* gout;                      /* device [N,D] */
    float* dx;                              /* device [N,K,D] contiguous, complete */
    float* ds;                              /* device [N,K] workspace */
    float* dgin;                            /* device [N*K,64] */
    float* whh_slab;                        /* device workspace, ceil(N/32) * 512 floats: per-tile partials of dW_hh */
    float* dwhh_pad;                        /* device [2,32,8] out: dW_hh of both directions, hidden size padded to 8 */
} kpgnn_attn_scan_desc;

int kpgnn_attn_scan_fwd(const kpgnn_attn_scan_desc* d, kpgnn_stream_t stream);
int kpgnn_attn_scan_bwd(const kpgnn_attn_scan_desc* d, kpgnn_stream_t stream);
int kpgnn_attn_scan_unpad(const float* dw_pad, const float* db_pad, const float* dwhh_pad, float* dw, float* db, float* dwhh,
                          int32_t K, int32_t D, kpgnn_stream_t stream);

/* Geometric hop-combine weights (reference layers/combine.py:43-50, GeometricCombine.geometric_distribution):
 *     a = sigmoid(alpha[d]);  theta[k,d] = softmax_k( a (1-a)^k )            theta: device [K,D], alpha: device [D]
 * and the backward  galpha[d] = d/dalpha sum_k gtheta[k,d] theta[k,d].  One launch each (the op-by-op version is
 * 6 + 20 launches per layer on a ~100-element tensor). */
int kpgnn_geo_theta_fwd(const float* alpha, int32_t K, int32_t D, float* theta, kpgnn_stream_t stream);
int kpgnn_geo_theta_bwd(const float* alpha, const float* theta, const float* gtheta, int32_t K, int32_t D,
                        float* galpha, kpgnn_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * KP-GIN per-hop MLP + geometric hop-combine + combine_proj (reference layers/KPGIN.py:106-112, combine.py:52-58):
 *     h1[n,k,:] = relu( s[n,k,:] W1[k] + b1[k] )      h2[n,k,:] = relu( h1[n,k,:] W2[k] + b2[k] )
 *     comb[n,:] = sum_k theta[k,:] * h2[n,k,:]        (only with theta; otherwise h2 [N,K,DO] is the result)
 *     out[n,:]  = comb[n,:] Wc^T + bc                 (only with wc: H outputs; otherwise comb [N,DO] is the result)
 * The reference runs this as two batched matmuls over [K, N, dk] with dk = hidden/K (13 for ZINC, 20 for QM9,
 * 6 for K=16), a broadcast multiply + sum and an nn.Linear(dk, hidden) - all far too narrow for a BLAS tile.  Here a
 * block stages a 32-node tile of s in LDS, every wave runs v_mfma_f32_16x16x4_f32 (exact fp32) over hop slots with
 * the zero-padded weights in LDS, and the activations, the combine, the projection and (backward) every weight, bias
 * and theta gradient plus d(s) come out of the same tile residency: s, h1, h2, out cross HBM once per direction.
 *   fwd: writes h1, h2 (saved for backward) and out ([N,H] with wc, [N,DO] with theta only).
 *   bwd: gout is [N,H] with wc, [N,DO] with theta only, [N,K,DO] otherwise; writes gs [N,K,DI] and
 *        gflat = [ dW1 (K*DI*DO) | db1 (K*DO) | dW2 (K*DO*DO) | db2 (K*DO) | dtheta (K*DO, with theta) |
 *                  dWc (H*DO) | dbc (H) (with wc) ],  per-block partials added in block order (deterministic).
 * Limits (KPGNN_ELIMIT beyond): DI, DO <= 32; K * ceil(max(DI,DO)/16)^2 <= 32; H <= 1024 and
 * ceil(H/16) * ceil(max(DI,DO)/16) <= 32; K*max(DI,DO) <= 1024; the LDS footprint (<= 160 KB).
 * ---------------------------------------------------------------------------------------------- */
typedef struct kpgnn_hop_mlp_desc {
    int64_t N;
    int32_t K, DI, DO;
    int32_t H;                              /* combine_proj outputs; 0 = no projection */
    const float* s;                         /* device [N,K,DI] contiguous */
    const float* w1; const float* b1;       /* device [K,DI,DO], [K,DO] */
    const float* w2; const float* b2;       /* device [K,DO,DO], [K,DO] */
    const float* theta;                     /* device [K,DO] or NULL */
    const float* wc; const float* bc;       /* device [H,DO], [H] (bc may be NULL); wc needs theta */
    float* h1; float* h2;                   /* device [N,K,DO] contiguous (fwd: out, bwd: in) */
    float* out;                             /* device [N,H] / [N,DO] (fwd, with theta) */
    /* backward only */
    const float* gout;                      /* device [N,H] / [N,DO] / [N,K,DO] */
    float* gs;                              /* device [N,K,DI] */
    float* gflat;                           /* device, layout above */
    void* workspace; size_t workspace_bytes;  /* >= kpgnn_hop_mlp_workspace_bytes(N, K, DI, DO, H) */
} kpgnn_hop_mlp_desc;

/* 0 = the shape is not covered (the caller keeps its BLAS path). */
size_t kpgnn_hop_mlp_workspace_bytes(int64_t N, int32_t K, int32_t DI, int32_t DO, int32_t H);
int kpgnn_hop_mlp_fwd(const kpgnn_hop_mlp_desc* d, kpgnn_stream_t stream);
int kpgnn_hop_mlp_bwd(const kpgnn_hop_mlp_desc* d, kpgnn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* KPGNN_H_ */
