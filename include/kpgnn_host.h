/*
 * kpgnn_host.h - C ABI of libkpgnn_host.so: HOST-side (CPU, OpenMP) exact K-hop pre-transform and batch
 * builder for the KP-GNN hot path.  Integer work only; results are bit-identical to the reference's
 * /root/reference/data_utils.py:20-241 (extract_multi_hop_neighbors, adj_K_order, get_peripheral_attr,
 * extract_peripheral_attr_v2, nx_compute_shortest_path_length) including its quirks,
 * while every walk count stays below 2^31 (beyond that the reference's float32->int32 cast wraps and its
 * embedding lookup raises; this library reports KPGNN_HOST_ERANGE instead).
 *
 * The reference builds dense N x N matrix powers with scipy, a networkx DiGraph per graph and one
 * networkx subgraph + all-pairs BFS per (node, hop) (16.5 s for one 1280-node graph).  Here every source
 * node is processed independently: K sparse walk-count rows by frontier expansion, then per hop the
 * peripheral subgraph statistics by BFS over adjacency lists.  Graphs of a batch run in parallel and the
 * output is written already collated (PyG Batch.from_data_list layout: node-level tensors concatenated,
 * edge_index offset by the graph's first node).
 *
 * Plain pointers and sizes only; all buffers are caller-owned host memory except the opaque plan.
 */
#ifndef KPGNN_HOST_H_
#define KPGNN_HOST_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KPGNN_HOST_ABI_VERSION 1
#define KPGNN_HOST_OK 0
#define KPGNN_HOST_EINVAL (-1)
#define KPGNN_HOST_ERANGE (-2) /* a walk count reached 2^31: outside the reference's defined range */
#define KPGNN_HOST_ENOMEM (-3)

#define KPGNN_KERNEL_SPD 0 /* shortest-path-distance kernel, data_utils.py:63-74 */
#define KPGNN_KERNEL_GD 1  /* graph-diffusion kernel,        data_utils.py:57-62 */

typedef struct kpgnn_khop_args {
    int32_t K;                  /* number of hops */
    int32_t max_edge_attr_num;  /* clamp of the per-hop path count code (code = min(count, max)+1) */
    int32_t max_hop_num;        /* BFS cutoff inside a peripheral subgraph */
    int32_t max_edge_type;      /* edge types kept per peripheral subgraph */
    int32_t max_edge_count;     /* clamp of per-type edge counts */
    int32_t max_distance_count; /* clamp of per-distance pair counts */
    int32_t kernel;             /* KPGNN_KERNEL_SPD / KPGNN_KERNEL_GD */
} kpgnn_khop_args;

typedef struct kpgnn_khop_plan kpgnn_khop_plan; /* opaque */

int kpgnn_host_abi_version(void);
const char* kpgnn_host_last_error(void);

/* Phase 1: transform G graphs.  node_ptr[G+1] / edge_ptr[G+1] delimit each graph's nodes / input edges;
 * edge_index is [2, edge_ptr[G]] (row 0 sources, row 1 targets) with node ids LOCAL to their graph;
 * edge_attr is [edge_ptr[G]] 1-hop edge types (>= 2 by the reference's convention) or NULL (all 2,
 * data_utils.py:48-50).  num_threads <= 0 uses the OpenMP default.  Returns a plan in *out. */
int kpgnn_khop_plan_create(int64_t G, const int64_t* node_ptr, const int64_t* edge_ptr, const int64_t* edge_index,
                           const int64_t* edge_attr, const kpgnn_khop_args* args, int32_t num_threads,
                           kpgnn_khop_plan** out);

/* Phase 2: sizes.  out_edge_ptr[G+1] receives the K-hop edge offsets (out_edge_ptr[G] = total K-hop edges). */
int kpgnn_khop_plan_sizes(const kpgnn_khop_plan* plan, int64_t* out_edge_ptr);

/* Phase 3: collated export (any pointer may be NULL to skip that tensor).  With N = node_ptr[G],
 * E = out_edge_ptr[G]:  edge_index int64 [2,E] (GLOBAL node ids), edge_attr int64 [E,K],
 * pe_attr int64 [N,K-1] (all zero, Q1), peripheral_edge_attr int64 [N,K,max_edge_type,2],
 * peripheral_configuration_attr int64 [N,K,max_hop_num+1], batch int64 [N]. */
int kpgnn_khop_plan_export(const kpgnn_khop_plan* plan, int64_t* edge_index, int64_t* edge_attr, int64_t* pe_attr,
                           int64_t* peripheral_edge_attr, int64_t* peripheral_configuration_attr, int64_t* batch);

void kpgnn_khop_plan_destroy(kpgnn_khop_plan* plan);

/* Synthetic ZINC-12k-shaped molecule graphs (SURVEY.md 8d "S1"): graph g uses seed0+g.  n ~ clip(round(
 * N(23.2,4.5)),9,37); random tree with max degree 3 grown towards recent nodes; 1-3 ring closures between
 * nodes at tree distance 4 or 5; bond types {1,2,3} w.p. {.75,.2,.05} (+1 applied: edge types 2..4, as
 * train_ZINC.py:96-99); atom types uniform in [0,21).  Two calls: with all outputs NULL to get the
 * sizes (node_ptr/edge_ptr must be given), then with buffers. */
int kpgnn_synth_molecules(int64_t G, uint64_t seed0, int64_t* node_ptr, int64_t* edge_ptr, int64_t* edge_index,
                          int64_t* edge_attr, int64_t* atom_type);

/* The same generator with the shape spelled out (NULL = the ZINC-12k shape above): node count ~ clip(round(N(mean_nodes,
 * std_nodes)), min_nodes, max_nodes); bond type t+1 with probability bond_prob[t] (t < num_bond_types <= 8); atom types
 * uniform in [0, num_atom_types).  QM9-shaped batches: N(18,3) clipped to [4,29], 4 bond types. */
typedef struct kpgnn_synth_shape {
    double mean_nodes, std_nodes;
    int32_t min_nodes, max_nodes;
    int32_t num_bond_types;
    double bond_prob[8];
    int32_t num_atom_types;
} kpgnn_synth_shape;
int kpgnn_synth_molecules_ex(const kpgnn_synth_shape* shape, int64_t G, uint64_t seed0, int64_t* node_ptr,
                             int64_t* edge_ptr, int64_t* edge_index, int64_t* edge_attr, int64_t* atom_type);

#ifdef __cplusplus
}
#endif
#endif /* KPGNN_HOST_H_ */
