"""K-hop CSR: the device-resident re-layout of a collated K-hop edge list that the HIP aggregation
kernels stream (include/kpgnn.h, "K-hop CSR").

The reference passes `edge_index [2,E]` + `edge_attr [E,K]` (int64) straight to PyG's propagate on every
layer call (layers/KPGIN.py:100, KPGINplus.py:74, KPGCN.py:110, gine.py:52).  Here the pair is converted
once per batch - on the GPU, no host round trip of the indices - into int32/uint16 CSR arrays keyed by
(node, hop) in both orientations, and the result is remembered ON the edge_index tensor object, so the L
layer calls of one forward (and GNNPlus's per-layer `edge_attr[:, :k]` prefixes, models/GNNs.py:429)
share one build.  Nothing is cached on the nn.Modules (deepcopy / state_dict stay clean).
"""
import ctypes
import weakref

import torch

from . import _lib

_ATTR = "_kpgnn_khop_csr"


class KHopCSR:
    """int32 CSR by (dst,hop) and by (src,hop) of the active (edge,hop) pairs."""

    __slots__ = ("N", "K", "E", "A", "rowptr_dst", "col_dst", "code_dst", "rowptr_src", "col_src", "code_src",
                 "tile_ptr", "tile_pack", "nodes_per_tile", "max_code0", "max_codek", "_dis", "_apairs",
                 "_dict_packs", "_tile_lists", "_keep", "_max_mult", "graph_ptr", "max_graph_nodes", "device")

    NODES_PER_TILE = 8  # destination nodes per LDS tile of the table-gradient kernel

    def __init__(self):
        self._dis = None
        self._apairs = {}
        self._dict_packs = {}   # ops.dict_tile_pack: uid-sorted dictionary entries per tile, keyed by the uid tensor
        self._tile_lists = {}   # tile_list(k): hop-prefix copies of (tile_ptr, tile_pack)
        self._max_mult = None   # largest multiplicity of an entry of tile_pack (host int), once known
        # graph boundaries of the collated batch (int32 [G+1] node offsets) and the largest graph, when the batch builder knows
        # them: dense neighbourhoods are then gathered from an LDS-staged hop slab (kpgnn_agg_fwd_desc.graph_ptr)
        self.graph_ptr, self.max_graph_nodes = None, 0

    def tile_list(self, k_active):
        """(tile_ptr, tile_pack) restricted to hops < k_active (kpgnn_tile_pack_filter): what kpgnn_table_grad walks for a
        layer that aggregates a hop prefix.  Static per batch; built on first use and kept."""
        if self.tile_ptr is None:
            raise _lib.KpgnnError("this CSR was built without the table-gradient entry list")
        if k_active >= self.K:
            return self.tile_ptr, self.tile_pack
        hit = self._tile_lists.get(k_active)
        if hit is None:
            from . import _lib
            ntiles = self.tile_ptr.numel() - 1
            optr = torch.empty_like(self.tile_ptr)
            opack = torch.empty_like(self.tile_pack)
            scratch = torch.empty(max(ntiles, 1), dtype=torch.int32, device=self.tile_ptr.device)
            with torch.cuda.device(self.tile_ptr.device):
                _lib.check(_lib.load().kpgnn_tile_pack_filter(
                    self.tile_ptr.data_ptr(), self.tile_pack.data_ptr(), ntiles, k_active, optr.data_ptr(), opack.data_ptr(),
                    scratch.data_ptr(), torch.cuda.current_stream().cuda_stream), "kpgnn_tile_pack_filter")
            hit = self._tile_lists[k_active] = (optr, opack)
        return hit

    def max_multiplicity(self):
        """Largest multiplicity of a single table-gradient entry (1..64), or 0 when it is not known and cannot be read back
        now (inside a stream capture).  Below 64 no run of equal (node, hop, code) pairs was cut, so every cell of a tile's
        count matrix holds at most 63: the condition of the matrix-core table-gradient kernel (kpgnn_table_grad_desc.
        max_multiplicity).  One host sync per CSR object; batches collated from a KHopDataset inherit the dataset's value."""
        if self._max_mult is None:
            if self.tile_ptr is None or torch.cuda.is_current_stream_capturing():
                return 0
            n = self.tile_ptr[-1]
            idx = torch.arange(self.tile_pack.numel(), device=self.tile_pack.device)
            m = torch.where(idx < n, (self.tile_pack >> 6) & 63, torch.zeros_like(self.tile_pack))
            self._max_mult = int(m.max().item()) + 1 if self.tile_pack.numel() else 1
        return self._max_mult

    def active_pairs(self, k_active):
        """Number of active (edge,hop) pairs within the first k_active hops (== A for k_active == K).
        Used for byte accounting only; one host sync per distinct k, cached."""
        if k_active >= self.K:
            return self.A
        if k_active not in self._apairs:
            rp = self.rowptr_dst[:-1].view(self.N, self.K)
            self._apairs[k_active] = int((self.rowptr_dst[k_active::self.K][:self.N] - rp[:, 0]).sum().item()) \
                if self.N > 0 else 0
        return self._apairs[k_active]

    def gcn_dis(self):
        """deg^-1/2 per (node,hop) with the KP-GCN self loop counted (layers/KPGCN.py:11-25,106-108)."""
        from .ops import dyn_ptr
        if self._dis is None or dyn_ptr(self.N) is not None:     # (a StaticBatch refills rowptr in place: no caching there)
            deg = (self.rowptr_dst[1:] - self.rowptr_dst[:-1] + 1).to(torch.float32)
            self._dis = deg.pow(-0.5).contiguous()
        return self._dis

    @staticmethod
    def build(edge_index, edge_attr, num_nodes, nodes_per_tile=NODES_PER_TILE):
        """nodes_per_tile: tile size of the table-gradient entry list (1..8; 1 = per-node lists, what KHopDataset keeps);
        None skips the list."""
        if not edge_index.is_cuda:
            raise _lib.KpgnnError("KHopCSR.build needs device tensors: the HIP path has no CPU fallback")
        lib = _lib.load()
        if edge_attr.dim() == 1:
            edge_attr = edge_attr.unsqueeze(-1)
        if edge_index.dtype != torch.int64 or edge_index.stride(1) != 1:
            edge_index = edge_index.to(torch.int64).contiguous()
        if edge_attr.dtype != torch.int64 or edge_attr.stride(1) != 1:
            edge_attr = edge_attr.to(torch.int64).contiguous()
        E, K = edge_attr.shape
        if edge_index.shape != (2, E):
            raise ValueError(f"edge_index {tuple(edge_index.shape)} does not match edge_attr {tuple(edge_attr.shape)}")
        N = int(num_nodes)
        dev = edge_index.device
        stream = torch.cuda.current_stream(dev).cuda_stream
        ei_stride = edge_index.stride(0) if E > 0 else max(E, 1)
        at_stride = edge_attr.stride(0) if E > 0 else K
        stats = torch.empty(8, dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.kpgnn_csr_stats(edge_index.data_ptr(), ei_stride, edge_attr.data_ptr(), at_stride, E, K,
                                           stats.data_ptr(), stream), "kpgnn_csr_stats")
            A, max0, maxk, minv, nmin, nmax = stats.tolist()[:6]  # the one host sync of a batch build
            if minv < 0:
                raise ValueError("edge_attr holds negative codes")
            if E > 0 and (nmin < 0 or nmax >= N):
                raise IndexError(f"edge_index out of range [0, {N}): min {nmin}, max {nmax}")
            if max(max0, maxk) > 65535:
                raise _lib.KpgnnError(f"edge code {max(max0, maxk)} exceeds the uint16 code range")
            c = KHopCSR()
            c.N, c.K, c.E, c.A, c.device = N, K, E, int(A), dev
            c.max_code0, c.max_codek = int(max0), int(maxk)
            S = N * K
            i32 = dict(dtype=torch.int32, device=dev)
            c.rowptr_dst = torch.empty(S + 1, **i32)
            c.rowptr_src = torch.empty(S + 1, **i32)
            c.col_dst = torch.empty(max(c.A, 1), **i32)
            c.col_src = torch.empty(max(c.A, 1), **i32)
            c.code_dst = torch.empty(max(c.A, 1), dtype=torch.int16, device=dev)  # uint16 payload
            c.code_src = torch.empty(max(c.A, 1), dtype=torch.int16, device=dev)
            c.nodes_per_tile = nodes_per_tile if nodes_per_tile is not None else KHopCSR.NODES_PER_TILE
            ntiles = (N + c.nodes_per_tile - 1) // c.nodes_per_tile
            c.tile_ptr = torch.empty(ntiles + 1, **i32) if nodes_per_tile is not None else None
            c.tile_pack = torch.empty(max(c.A, 1), **i32) if nodes_per_tile is not None else None  # uint32 payload
            ws_bytes = lib.kpgnn_csr_workspace_bytes(E, c.A, N, K)
            ws = torch.empty(max(int(ws_bytes), 256), dtype=torch.uint8, device=dev)
            _lib.check(lib.kpgnn_csr_build(edge_index.data_ptr(), ei_stride, edge_attr.data_ptr(), at_stride, E, K, N,
                                           c.A, c.rowptr_dst.data_ptr(), c.col_dst.data_ptr(), c.code_dst.data_ptr(),
                                           c.rowptr_src.data_ptr(), c.col_src.data_ptr(), c.code_src.data_ptr(),
                                           c.nodes_per_tile, None if c.tile_ptr is None else c.tile_ptr.data_ptr(),
                                           None if c.tile_pack is None else c.tile_pack.data_ptr(),
                                           ws.data_ptr(), ctypes.c_size_t(ws.numel()), stream), "kpgnn_csr_build")
        return c


def _attr_base(edge_attr):
    """The [E,K_full] tensor a column-prefix view `edge_attr[:, :k]` was sliced from (else edge_attr itself)."""
    b = edge_attr._base
    if (b is not None and b.dim() == 2 and edge_attr.dim() == 2 and b.shape[0] == edge_attr.shape[0]
            and edge_attr.shape[1] <= b.shape[1] and b.stride(1) == 1 and edge_attr.stride(1) == 1
            and edge_attr.stride(0) == b.stride(0) and edge_attr.storage_offset() == b.storage_offset()
            and b.dtype == edge_attr.dtype):
        return b
    return edge_attr


def get_khop_csr(edge_index, edge_attr, num_nodes):
    """CSR for (edge_index, edge_attr), built once per batch object.  Returns (csr, k_active)."""
    if edge_attr.dim() == 1:
        edge_attr = edge_attr.unsqueeze(-1)
    base = _attr_base(edge_attr)
    k_active = edge_attr.shape[1]
    rec = getattr(edge_index, _ATTR, None)
    if rec is not None:
        ref, bver, ever, n, csr = rec
        if ref() is base and bver == base._version and ever == edge_index._version and n == num_nodes:
            return csr, k_active
    csr = KHopCSR.build(edge_index, base, num_nodes)
    try:
        setattr(edge_index, _ATTR, (weakref.ref(base), base._version, edge_index._version, num_nodes, csr))
    except Exception:  # pragma: no cover - tensors normally accept attributes
        pass
    return csr, k_active


def attach_khop_csr(edge_index, edge_attr, num_nodes, csr):
    """Pre-attach a CSR (e.g. built by the batch builder at collate time) so layers reuse it."""
    setattr(edge_index, _ATTR, (weakref.ref(edge_attr), edge_attr._version, edge_index._version, num_nodes, csr))


_ZATTR = "_kpgnn_all_zero"


def path_encoding_is_zero(pe_attr):
    """True iff pe_attr holds only the padding index 0 (always the case for the reference's own
    pre-transform: data_utils.py:91 reads a diagonal that :123 zeroed).  One sync per batch object."""
    base = pe_attr._base if pe_attr._base is not None else pe_attr
    rec = getattr(base, _ZATTR, None)
    if rec is not None and rec[0] == base._version:
        return rec[1]
    z = not bool(base.any().item())
    try:
        setattr(base, _ZATTR, (base._version, z))
    except Exception:  # pragma: no cover
        pass
    return z
