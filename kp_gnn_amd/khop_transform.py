"""Host mirror of the reference's data_utils.py pre-transform API over libkpgnn_host.so
(include/kpgnn_host.h): exact (bit-identical) K-hop edge lists, edge codes and peripheral-subgraph
features, for one graph (`extract_multi_hop_neighbors`, reference data_utils.py:20-107) or for a whole
batch written already collated (`khop_batch`, the batch builder either side of the hot path)."""
import ctypes
import os

import numpy as np
import torch

from ._env import usable_cpus
from ._lib import KpgnnError

_PKG = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_PKG, "libkpgnn_host.so")

c_i64p = ctypes.POINTER(ctypes.c_int64)


class KhopArgs(ctypes.Structure):
    _fields_ = [("K", ctypes.c_int32), ("max_edge_attr_num", ctypes.c_int32), ("max_hop_num", ctypes.c_int32),
                ("max_edge_type", ctypes.c_int32), ("max_edge_count", ctypes.c_int32),
                ("max_distance_count", ctypes.c_int32), ("kernel", ctypes.c_int32)]


HOST_SIGNATURES = {
    "kpgnn_host_abi_version": (ctypes.c_int, []),
    "kpgnn_host_last_error": (ctypes.c_char_p, []),
    "kpgnn_khop_plan_create": (ctypes.c_int, [ctypes.c_int64, c_i64p, c_i64p, c_i64p, c_i64p,
                                              ctypes.POINTER(KhopArgs), ctypes.c_int32,
                                              ctypes.POINTER(ctypes.c_void_p)]),
    "kpgnn_khop_plan_sizes": (ctypes.c_int, [ctypes.c_void_p, c_i64p]),
    "kpgnn_khop_plan_export": (ctypes.c_int, [ctypes.c_void_p] + [c_i64p] * 6),
    "kpgnn_khop_plan_destroy": (None, [ctypes.c_void_p]),
    "kpgnn_synth_molecules": (ctypes.c_int, [ctypes.c_int64, ctypes.c_uint64] + [c_i64p] * 5),
    "kpgnn_synth_molecules_ex": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64] + [c_i64p] * 5),
}

_host = None


def load_host(path=None):
    global _host
    if _host is not None and path is None:
        return _host
    path = path or HOST_LIB_PATH
    if not os.path.exists(path):
        raise KpgnnError(f"{path} not found: build it with `python -m kp_gnn_amd.build`")
    lib = ctypes.CDLL(path)
    for name, (res, args) in HOST_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.kpgnn_host_abi_version() != 1:
        raise KpgnnError("libkpgnn_host.so ABI version mismatch")
    _host = lib
    return lib


def _check(rc, what):
    if rc != 0:
        msg = load_host().kpgnn_host_last_error()
        exc = ValueError if rc == -1 else KpgnnError
        raise exc(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


def _p(a):
    return None if a is None else a.ctypes.data_as(c_i64p)


def _i64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int64))


def khop_batch(node_ptr, edge_ptr, edge_index, edge_attr, K, max_edge_attr_num, max_hop_num, max_edge_type,
               max_edge_count, max_distance_count, kernel, num_threads=0):
    """Transform G graphs at once; returns a dict of collated int64 torch tensors (host memory):
    edge_index [2,E], edge_attr [E,K], pe_attr [N,K-1] (None if K == 1), peripheral_edge_attr
    [N,K,max_edge_type,2], peripheral_configuration_attr [N,K,max_hop_num+1] (both None when
    max_hop_num == 0 or max_edge_type == 0, data_utils.py:141,157-159), batch [N], edge_ptr [G+1],
    node_ptr [G+1]."""
    lib = load_host()
    node_ptr, edge_ptr = _i64(node_ptr), _i64(edge_ptr)
    G = node_ptr.shape[0] - 1
    edge_index = _i64(edge_index).reshape(2, -1)
    if edge_index.shape[1] != edge_ptr[-1]:
        raise ValueError("edge_index does not match edge_ptr")
    ea = None if edge_attr is None else _i64(edge_attr).reshape(-1)
    if kernel not in ("spd", "gd"):
        raise ValueError(f"unknown kernel {kernel!r}")
    args = KhopArgs(K, max_edge_attr_num, max_hop_num, max_edge_type, max_edge_count, max_distance_count,
                    0 if kernel == "spd" else 1)
    if num_threads <= 0:
        num_threads = usable_cpus()
    plan = ctypes.c_void_p()
    _check(lib.kpgnn_khop_plan_create(G, _p(node_ptr), _p(edge_ptr), _p(edge_index), _p(ea), ctypes.byref(args),
                                      num_threads, ctypes.byref(plan)), "kpgnn_khop_plan_create")
    try:
        out_eptr = np.zeros(G + 1, dtype=np.int64)
        _check(lib.kpgnn_khop_plan_sizes(plan, _p(out_eptr)), "kpgnn_khop_plan_sizes")
        N, E = int(node_ptr[-1]), int(out_eptr[-1])
        want_p = max_hop_num > 0 and max_edge_type > 0
        o_ei = np.empty((2, E), dtype=np.int64)
        o_ea = np.empty((E, K), dtype=np.int64)
        o_pe = np.empty((N, K - 1), dtype=np.int64) if K > 1 else None
        o_pea = np.empty((N, K, max_edge_type, 2), dtype=np.int64) if want_p else None
        o_pca = np.empty((N, K, max_hop_num + 1), dtype=np.int64) if want_p else None
        o_b = np.empty(N, dtype=np.int64)
        _check(lib.kpgnn_khop_plan_export(plan, _p(o_ei), _p(o_ea), _p(o_pe), _p(o_pea), _p(o_pca), _p(o_b)),
               "kpgnn_khop_plan_export")
    finally:
        lib.kpgnn_khop_plan_destroy(plan)
    t = lambda a: None if a is None else torch.from_numpy(a)  # noqa: E731
    return {"edge_index": t(o_ei), "edge_attr": t(o_ea), "pe_attr": t(o_pe), "peripheral_edge_attr": t(o_pea),
            "peripheral_configuration_attr": t(o_pca), "batch": t(o_b), "edge_ptr": t(out_eptr),
            "node_ptr": t(node_ptr.copy())}


def extract_multi_hop_neighbors(data, K, max_edge_attr_num, max_hop_num, max_edge_type, max_edge_count,
                                max_distance_count, kernel):
    """Same contract as the reference's data_utils.extract_multi_hop_neighbors (:20-107): rewrites and
    returns `data` (any object with x / edge_index / optional edge_attr / num_nodes attributes)."""
    edge_index = data.edge_index
    num_nodes = data.num_nodes if getattr(data, "num_nodes", None) is not None else data.x.size(0)
    if edge_index.size(1) == 0:  # reference :36-44 (note the different attribute name and width, Q8)
        data.peripheral_edge_attr = torch.zeros([num_nodes, K, max_edge_type, 2], dtype=torch.long)
        data.peripheral_configuration = torch.zeros([num_nodes, K, max_hop_num], dtype=torch.long)
        return data
    ea = getattr(data, "edge_attr", None)
    out = khop_batch([0, num_nodes], [0, edge_index.size(1)], edge_index.cpu().numpy(),
                     None if ea is None else ea.cpu().numpy(), K, max_edge_attr_num, max_hop_num, max_edge_type,
                     max_edge_count, max_distance_count, kernel, num_threads=1)
    data.edge_index = out["edge_index"]
    data.edge_attr = out["edge_attr"]
    data.peripheral_edge_attr = out["peripheral_edge_attr"]
    data.peripheral_configuration_attr = out["peripheral_configuration_attr"]
    data.pe_attr = out["pe_attr"]
    return data


class SynthShape(ctypes.Structure):
    """kpgnn_synth_shape (include/kpgnn_host.h)."""
    _fields_ = [("mean_nodes", ctypes.c_double), ("std_nodes", ctypes.c_double), ("min_nodes", ctypes.c_int32),
                ("max_nodes", ctypes.c_int32), ("num_bond_types", ctypes.c_int32), ("bond_prob", ctypes.c_double * 8),
                ("num_atom_types", ctypes.c_int32)]


def qm9_shape():
    """QM9-shaped molecules: n ~ N(18, 3) clipped to [4, 29], 4 bond types."""
    sh = SynthShape()
    sh.mean_nodes, sh.std_nodes, sh.min_nodes, sh.max_nodes, sh.num_bond_types, sh.num_atom_types = 18.0, 3.0, 4, 29, 4, 5
    for i, v in enumerate((0.70, 0.20, 0.07, 0.03)):
        sh.bond_prob[i] = v
    return sh


def synth_molecules(num_graphs, seed0=0, shape=None):
    """Synthetic molecule graphs (see kpgnn_host.h; shape None = ZINC-12k-shaped).  Returns numpy int64 arrays
    node_ptr [G+1], edge_ptr [G+1], edge_index [2,E] (local ids), edge_attr [E] (types 2..), x [N]."""
    lib = load_host()
    G = int(num_graphs)
    sp = ctypes.byref(shape) if shape is not None else None
    node_ptr = np.zeros(G + 1, dtype=np.int64)
    edge_ptr = np.zeros(G + 1, dtype=np.int64)
    _check(lib.kpgnn_synth_molecules_ex(sp, G, seed0, _p(node_ptr), _p(edge_ptr), None, None, None), "kpgnn_synth_molecules_ex")
    ei = np.empty((2, int(edge_ptr[-1])), dtype=np.int64)
    ea = np.empty(int(edge_ptr[-1]), dtype=np.int64)
    x = np.empty(int(node_ptr[-1]), dtype=np.int64)
    _check(lib.kpgnn_synth_molecules_ex(sp, G, seed0, _p(node_ptr), _p(edge_ptr), _p(ei), _p(ea), _p(x)),
           "kpgnn_synth_molecules_ex")
    return node_ptr, edge_ptr, ei, ea, x
