"""ORACLE (test infrastructure, not product code): CPU restatement of KP-GNN's K-hop pre-transform.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.  The
product path (kp_gnn_amd.khop / csrc/khop_host.cpp) never does.

Restates, in dense numpy and in the reference's own order of operations,
    /root/reference/data_utils.py:20-107   extract_multi_hop_neighbors
    /root/reference/data_utils.py:110-125  adj_K_order
    /root/reference/data_utils.py:128-162  get_peripheral_attr
    /root/reference/data_utils.py:165-221  extract_peripheral_attr_v2
    /root/reference/data_utils.py:224-241  nx_compute_shortest_path_length
Third-party pieces restated from their documented behaviour: PyG to_scipy_sparse_matrix (COO,
duplicates summed, float32 ones), networkx from_numpy_array(DiGraph).edges (row-major nonzeros),
networkx all_pairs_shortest_path_length(cutoff) (directed BFS).

Parity pin: tests/golden/khop_preprocess.npz, produced by running the reference's data_utils.py
itself in the build container (tests/golden/make_golden.py); tests/test_oracle_golden.py asserts
bit-equality for every case.

Integer range note: the reference computes walk counts in float32 and casts with
.int(); this restatement uses exact int64.  Both agree while every walk count is < 2**31 (beyond that
the reference's cast wraps to INT_MIN and its embedding lookup raises) because every use of a count
is clamped to max_edge_attr_num << 2**24 first.
"""
import numpy as np


def _dense_counts(num_nodes, edge_index, values):
    m = np.zeros((num_nodes, num_nodes), dtype=np.int64)
    np.add.at(m, (edge_index[0], edge_index[1]), values)  # COO -> dense sums duplicates
    return m


def adj_K_order(adj, K):
    """data_utils.py:110-125: powers first (with their diagonals), diagonals zeroed afterwards."""
    powers = [adj.copy()]
    for _ in range(K - 1):
        powers.append(powers[-1] @ adj)
    out = []
    for p in powers:
        p = p.copy()
        np.fill_diagonal(p, 0)
        out.append(p)
    return out


def _sub_apsp(sub_nz, max_length):
    """Directed BFS distances (1..max_length) inside a subgraph; 0 = self / unreachable / beyond cutoff.
    data_utils.py:224-241."""
    m = sub_nz.shape[0]
    dist = np.zeros((m, m), dtype=np.int64)
    reach = np.eye(m, dtype=bool)
    frontier = np.eye(m, dtype=bool)
    a = sub_nz.astype(np.int64)
    for h in range(1, max_length + 1):
        frontier = ((frontier.astype(np.int64) @ a) > 0) & ~reach
        if not frontier.any():
            break
        dist[frontier] = h
        reach |= frontier
    return dist


def extract_peripheral_attr_v2(adj, k_adj, max_hop_num, max_edge_type, max_edge_count, max_distance_count):
    """data_utils.py:165-221 for one hop: returns ([N,max_edge_type,2], [N,max_hop_num+1])."""
    n = adj.shape[0]
    pe = np.zeros((n, max_edge_type, 2), dtype=np.int64)
    pc = np.zeros((n, max_hop_num + 1), dtype=np.int64)
    for i in range(n):
        row = np.nonzero(k_adj[i] > 0)[0]
        if row.size < 2:
            continue
        sub = adj[np.ix_(row, row)]
        weights = sub[sub != 0]  # edge "weight" list of the induced DiGraph, row-major (:189-190)
        if weights.size == 0:
            continue
        edge_count = np.bincount(weights, minlength=max_edge_type + 2)[2:]  # :195-197 (types 0,1 dropped)
        order = np.argsort(-edge_count, kind="stable")  # torch.sort(descending) tie order: ascending index (Q6)
        sort_count = edge_count[order][:max_edge_type].copy()
        sort_type = order[:max_edge_type]  # NB index into edge_count[2:], i.e. type-2 (Q4)
        sort_count[sort_count > max_edge_count] = max_edge_count
        pe[i, :, 0] = sort_type
        pe[i, :, 1] = sort_count
        spm = _sub_apsp(sub != 0, max_hop_num)
        num_sub_p_edges = 0
        for j in range(row.size):
            for h in range(1, max_hop_num + 1):
                h_nodes = np.nonzero(spm[j] == h)[0]
                if h_nodes.size < 2:
                    continue
                num_sub_p_edges += int(sub[np.ix_(h_nodes, h_nodes)].sum())  # sums type VALUES (Q5)
        conf = np.bincount(spm.reshape(-1), minlength=max_hop_num + 1)
        conf[0] = num_sub_p_edges
        conf[conf > max_distance_count] = max_distance_count
        pc[i, :] = conf
    return pe, pc


def extract_multi_hop_neighbors(num_nodes, edge_index, edge_attr, K, max_edge_attr_num, max_hop_num,
                                max_edge_type, max_edge_count, max_distance_count, kernel):
    """data_utils.py:20-107.  edge_index int [2,E]; edge_attr int [E] or None.  Returns a dict of
    int64 arrays with the reference's attribute names (absent attributes are absent keys)."""
    edge_index = np.asarray(edge_index, dtype=np.int64).reshape(2, -1)
    if edge_index.shape[1] == 0:  # :36-44 (Q8: different name and width, no pe_attr)
        return {"edge_index": edge_index,
                "peripheral_edge_attr": np.zeros((num_nodes, K, max_edge_type, 2), dtype=np.int64),
                "peripheral_configuration": np.zeros((num_nodes, K, max_hop_num), dtype=np.int64)}
    if edge_attr is None:
        edge_attr = np.full(edge_index.shape[1], 2, dtype=np.int64)  # :48-50
    edge_attr = np.asarray(edge_attr, dtype=np.int64).reshape(-1)
    adj = _dense_counts(num_nodes, edge_index, np.ones(edge_index.shape[1], dtype=np.int64))
    edge_attr_adj = _dense_counts(num_nodes, edge_index, edge_attr)
    adj_list = adj_K_order(adj, K)

    if kernel == "gd":  # :57-62
        final_adj = np.zeros_like(adj)
        for a in adj_list:
            final_adj = final_adj + a
        final_adj[final_adj > 1] = 1
    else:  # :63-74
        exist = adj_list[0].copy()
        for i in range(1, K):
            a = adj_list[i].copy()
            a[exist > 0] = 0
            exist = exist + a
            exist[exist > 1] = 1
            adj_list[i] = a
        final_adj = exist

    src, dst = np.nonzero(final_adj)  # nx DiGraph.edges order == row-major nonzeros (:76-78)
    new_edge_index = np.stack([src, dst]).astype(np.int64)
    cols = [edge_attr_adj[src, dst]]
    pe_cols = []
    for i in range(1, K):
        a = adj_list[i].copy()
        a[a > max_edge_attr_num] = max_edge_attr_num
        a[a > 0] += 1
        cols.append(a[src, dst])
        pe_cols.append(np.diag(a))
    out = {"edge_index": new_edge_index, "edge_attr": np.stack(cols, axis=-1)}
    if K > 1:
        out["pe_attr"] = np.stack(pe_cols, axis=-1)
    if max_hop_num > 0 and max_edge_type > 0:  # :141-159
        pes, pcs = [], []
        for i in range(K):
            pe, pc = extract_peripheral_attr_v2(edge_attr_adj, adj_list[i], max_hop_num, max_edge_type,
                                                max_edge_count, max_distance_count)
            pes.append(pe)
            pcs.append(pc)
        out["peripheral_edge_attr"] = np.stack(pes, axis=1)
        out["peripheral_configuration_attr"] = np.stack(pcs, axis=1)
    return out
