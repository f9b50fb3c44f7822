"""A pre-transformed dataset resident in HBM, and the per-step collate that turns a list of graph ids into a batch.

Reference: a dataset is pre-transformed once and stored as PyG's `(data, slices)` pair - every tensor concatenated over
the graphs plus per-graph offsets (datasets/ZINC_dataset.py:139-140); every training step then draws a NEW shuffled
subset, PyG's DataLoader collates it on the host (Batch.from_data_list: concatenate, offset edge_index by the graph's
first node, build `batch`) and the training loop ships ~120 MB of int64 indices to the GPU (train_ZINC.py:224,36-40).

Here the same (data, slices) layout lives on the device as int32 / uint16 arrays together with the K-hop CSR the HIP
kernels stream - built ONCE per dataset with the device builder (kpgnn_csr_build), graph-local - and `collate(ids)` is a
handful of launches (kpgnn_collate): concatenation + offsets, no sort, no host synchronisation, no int64 traffic.  The
sizes of a batch (nodes, active pairs, table-gradient entries) are sums of per-graph counts kept on the host.  The
result is bit-identical to building the CSR from the PyG-collated batch of the same graphs (tests/test_dataset.py).
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .batch import KHopBatch
from .khop_csr import _ZATTR, KHopCSR, attach_khop_csr

_ZERO_PE = {}      # (device, K-1) -> persistent all-zero [cap, K-1] int64 buffer the batches' pe_attr are views of (Q1)


class PeripheralDict:
    """Dataset-level dictionary of the peripheral index tuples: `rows` [U, C] int16 (C = 2T type/count columns, then the
    configuration columns) and the per-column maxima (host ints, for the embedding-range check without a device
    round trip); a batch carries `uid` [N, K] int32 into it."""

    def __init__(self, rows, col_max, T, Hc, dominant):
        self.rows, self.col_max, self.T, self.Hc, self.dominant = rows, tuple(int(v) for v in col_max), T, Hc, dominant
        self._sel = {}

    def select(self, use_e, use_c):
        """rows restricted to the columns a model reads (peripheral edge features and / or configuration features)."""
        key = (bool(use_e), bool(use_c))
        hit = self._sel.get(key)
        if hit is None:
            cols = (list(range(2 * self.T)) if use_e else []) + (list(range(2 * self.T, 2 * self.T + self.Hc)) if use_c else [])
            rows = self.rows if len(cols) == self.rows.shape[1] else self.rows[:, cols].contiguous()
            hit = self._sel[key] = (rows, tuple(self.col_max[c] for c in cols))
        return hit


class BatchPeripheral:
    """What a collated batch hands the bodies instead of the two dense int64 peripheral tensors."""

    def __init__(self, pdict, uid):
        self.pdict, self.uid = pdict, uid


class KHopDataset:
    """G pre-transformed graphs in HBM: node / graph attributes in (data, slices) layout + the graph-local K-hop CSR."""

    NODES_PER_TILE = KHopCSR.NODES_PER_TILE

    def __init__(self):
        self.node_rows = {}      # name -> [Nd, ...] device tensor gathered per batch node (x, z, ...)
        self.graph_rows = {}     # name -> [G, ...] device tensor gathered per batch graph (y, ...)
        self.index_bounds = {}   # name -> max value of an integer node attribute (host int; embedding range check)
        self.pdict = None
        self.uid = None          # [Nd, K] int32 dictionary id per (node, hop), or None

    # ------------------------------------------------------------------------------------------------ construction
    @staticmethod
    def from_collated(batch, node_ptr, device, chunk_graphs=4096, with_entry_lists=True):
        """`batch`: a HOST KHopBatch holding ALL graphs collated (batch.collate_khop: the reference's pre_transform output in
        PyG's concatenated layout); node_ptr [G+1] its slices.  The CSR is built chunk by chunk on the device."""
        if batch.edge_index.is_cuda:
            raise ValueError("from_collated expects the host-side collated dataset")
        ds = KHopDataset()
        node_ptr = np.asarray(node_ptr, dtype=np.int64)
        G = node_ptr.shape[0] - 1
        Nd = int(node_ptr[-1])
        ei, ea = batch.edge_index, batch.edge_attr
        if ea.dim() == 1:
            ea = ea.unsqueeze(-1)
        E, K = ea.shape
        ds.G, ds.K, ds.device = G, K, torch.device(device)
        # K-hop edges per graph: edges are grouped by graph in collate order (PyG's slices['edge_index'])
        edge_ptr = getattr(batch, "edge_ptr", None)
        if edge_ptr is None:
            src_graph = torch.bucketize(ei[0], torch.from_numpy(node_ptr[1:]), right=True) if E else ei.new_zeros(0)
            if E and bool((src_graph[1:] < src_graph[:-1]).any()):
                raise ValueError("the collated edge list must be grouped by graph")
            edge_cnt = np.bincount(src_graph.numpy(), minlength=G).astype(np.int64) if E else np.zeros(G, np.int64)
            edge_ptr = np.concatenate([[0], np.cumsum(edge_cnt)])
        edge_ptr = np.asarray(edge_ptr, dtype=np.int64)
        edge_cnt = np.diff(edge_ptr)
        parts = {k: [] for k in ("rowptr_dst", "rowptr_src", "col_dst", "col_src", "code_dst", "code_src", "ent_rel", "ent")}
        pair_cnt, ent_cnt, hop_cnt = [], [], []
        max0 = maxk = 0
        dev = ds.device
        for g0 in range(0, G, chunk_graphs):
            g1 = min(G, g0 + chunk_graphs)
            n0, n1, e0, e1 = int(node_ptr[g0]), int(node_ptr[g1]), int(edge_ptr[g0]), int(edge_ptr[g1])
            Nc = n1 - n0
            cei = (ei[:, e0:e1] - n0).to(dev)
            cea = ea[e0:e1].to(dev)
            csr = KHopCSR.build(cei, cea, Nc, nodes_per_tile=1 if with_entry_lists else None)
            max0, maxk = max(max0, csr.max_code0), max(maxk, csr.max_codek)
            nptr = torch.from_numpy(node_ptr[g0:g1 + 1] - n0).to(dev)                 # [gc+1] chunk-local node offsets
            gnodes = nptr[1:] - nptr[:-1]
            graph_of_node = torch.repeat_interleave(torch.arange(g1 - g0, device=dev), gnodes)
            pptr = csr.rowptr_dst[(nptr * K).long()].long()                           # [gc+1] first pair of each graph
            assert torch.equal(pptr, csr.rowptr_src[(nptr * K).long()].long()), "orientations disagree on a graph's pair count"
            seg_graph = graph_of_node.repeat_interleave(K)
            parts["rowptr_dst"].append((csr.rowptr_dst[:-1].long() - pptr[seg_graph]).to(torch.int32))
            parts["rowptr_src"].append((csr.rowptr_src[:-1].long() - pptr[seg_graph]).to(torch.int32))
            if csr.A:
                graph_of_pair = torch.bucketize(torch.arange(csr.A, device=dev), pptr[1:], right=True)
                nb = nptr[graph_of_pair].to(torch.int32)
                parts["col_dst"].append(csr.col_dst[:csr.A] - nb)
                parts["col_src"].append(csr.col_src[:csr.A] - nb)
                parts["code_dst"].append(csr.code_dst[:csr.A].clone())
                parts["code_src"].append(csr.code_src[:csr.A].clone())
            pair_cnt.append((pptr[1:] - pptr[:-1]).cpu().numpy())
            # active pairs per graph within the first k hops (byte accounting of layers that see a hop prefix)
            seg_len = (csr.rowptr_dst[1:] - csr.rowptr_dst[:-1]).view(Nc, K).long()
            per_graph = torch.zeros((g1 - g0, K), dtype=torch.long, device=dev).index_add_(0, graph_of_node, seg_len)
            hop_cnt.append(per_graph.cumsum(1).cpu().numpy())
            if with_entry_lists:
                eptr = csr.tile_ptr[nptr.long()].long()                               # nodes_per_tile = 1: tile == node
                parts["ent_rel"].append((csr.tile_ptr[:-1].long() - eptr[graph_of_node]).to(torch.int32))
                n_e = int(eptr[-1].item())
                parts["ent"].append(csr.tile_pack[:n_e].clone())
                ent_cnt.append((eptr[1:] - eptr[:-1]).cpu().numpy())
            del csr, cei, cea
        cat = lambda k, dt: (torch.cat(parts[k]) if parts[k] else torch.zeros(0, dtype=dt, device=dev))  # noqa: E731
        ds.rowptr_dst, ds.rowptr_src = cat("rowptr_dst", torch.int32), cat("rowptr_src", torch.int32)
        ds.col_dst, ds.col_src = cat("col_dst", torch.int32), cat("col_src", torch.int32)
        ds.code_dst, ds.code_src = cat("code_dst", torch.int16), cat("code_src", torch.int16)
        ds.has_entries = with_entry_lists
        ds.max_mult = 0
        if with_entry_lists:
            ds.ent_rel, ds.ent = cat("ent_rel", torch.int32), cat("ent", torch.int32)
            ds.max_mult = int(((ds.ent >> 6) & 63).max().item()) + 1 if ds.ent.numel() else 1     # largest entry multiplicity
        ds.max_code0, ds.max_codek = max0, maxk
        # per-graph counts (host) and their running sums (device, int64)
        ds.h_nodes = np.diff(node_ptr)
        ds.h_pairs = np.concatenate(pair_cnt) if pair_cnt else np.zeros(0, np.int64)
        ds.h_ents = np.concatenate(ent_cnt) if ent_cnt else np.zeros(G, np.int64)
        ds.h_edges = edge_cnt
        ds.h_hop_pairs = np.concatenate(hop_cnt) if hop_cnt else np.zeros((0, K), np.int64)
        run = lambda c: torch.from_numpy(np.concatenate([[0], np.cumsum(c)]).astype(np.int64)).to(dev)  # noqa: E731
        ds.node_ptr, ds.pair_ptr, ds.ent_ptr = run(ds.h_nodes), run(ds.h_pairs), run(ds.h_ents)
        # node / graph attributes
        x = batch.x
        if x is not None:
            if not x.is_floating_point():
                ds._add_index_rows("x", x)
            else:
                ds.node_rows["x"] = x.to(dev).contiguous()
        if getattr(batch, "z", None) is not None:
            ds._add_index_rows("z", batch.z)
        if batch.y is not None:
            ds.graph_rows["y"] = batch.y.to(dev).contiguous()
        pea, pca = batch.peripheral_edge_attr, batch.peripheral_configuration_attr
        if pea is not None and pca is not None:
            ds._build_dictionary(pea, pca)
        ds.Nd = Nd
        return ds

    def _add_index_rows(self, name, t):
        lo, hi = int(t.min()), int(t.max())
        if lo < 0 or hi >= 32768:
            raise ValueError(f"integer node attribute {name!r} outside [0, 32768)")
        self.node_rows[name] = t.to(torch.int16).to(self.device).contiguous()
        self.index_bounds[name] = hi

    def _build_dictionary(self, pea, pca):
        """Distinct (node, hop) peripheral index tuples of the whole dataset (35 for 10,000 ZINC-shaped molecules)."""
        N, K, T, _ = pea.shape
        Hc = pca.shape[-1]
        idx = torch.cat([pea.reshape(N * K, 2 * T), pca.reshape(N * K, Hc)], dim=1).to(self.device)
        if idx.numel() and int(idx.min()) < 0:
            raise IndexError("negative peripheral attribute index")
        col_max = idx.max(0).values.tolist() if idx.numel() else [0] * (2 * T + Hc)
        if max(col_max) >= 32768:
            return                                                # (keeps the dense path: uint16 tuples only)
        rows, inv = torch.unique(idx, dim=0, return_inverse=True)
        uid = inv.to(torch.int32).view(N, K).contiguous()
        dom = torch.mode(uid, dim=0).values.to(torch.int32).contiguous()      # most frequent id per hop (kpgnn_dict_grad's hint)
        self.pdict = PeripheralDict(rows.to(torch.int16).contiguous(), col_max, T, Hc, dom)
        self.uid = uid

    # ------------------------------------------------------------------------------------------------ (data, slices)
    def state(self):
        """(data, slices) in the reference's on-disk convention (ZINC_dataset.py:139-140): `data` the concatenated tensors,
        `slices` the per-graph offsets of each - here with the CSR arrays in place of edge_index / edge_attr."""
        data = {"rowptr_dst": self.rowptr_dst, "rowptr_src": self.rowptr_src, "col_dst": self.col_dst, "col_src": self.col_src,
                "code_dst": self.code_dst, "code_src": self.code_src}
        slices = {"node": self.node_ptr, "pair": self.pair_ptr}
        if self.has_entries:
            data.update(ent_rel=self.ent_rel, ent=self.ent)
            slices["ent"] = self.ent_ptr
        for k, v in self.node_rows.items():
            data["node." + k] = v
        for k, v in self.graph_rows.items():
            data["graph." + k] = v
        if self.pdict is not None:
            data.update(uid=self.uid, dict_rows=self.pdict.rows, dict_dominant=self.pdict.dominant)
        meta = {"K": self.K, "G": self.G, "max_code0": self.max_code0, "max_codek": self.max_codek, "max_mult": self.max_mult,
                "index_bounds": dict(self.index_bounds), "h_edges": torch.from_numpy(self.h_edges),
                "h_hop_pairs": torch.from_numpy(self.h_hop_pairs),
                "dict": None if self.pdict is None else {"col_max": list(self.pdict.col_max), "T": self.pdict.T, "Hc": self.pdict.Hc}}
        return {k: v.cpu() for k, v in data.items()}, {k: v.cpu() for k, v in slices.items()}, meta

    def save(self, path):
        torch.save(self.state(), path)

    @staticmethod
    def load(path, device):
        data, slices, meta = torch.load(path, weights_only=True)
        ds = KHopDataset()
        dev = ds.device = torch.device(device)
        ds.K, ds.G, ds.max_code0, ds.max_codek = meta["K"], meta["G"], meta["max_code0"], meta["max_codek"]
        ds.max_mult = meta.get("max_mult", 0)
        for k in ("rowptr_dst", "rowptr_src", "col_dst", "col_src", "code_dst", "code_src"):
            setattr(ds, k, data[k].to(dev))
        ds.node_ptr, ds.pair_ptr = slices["node"].to(dev), slices["pair"].to(dev)
        ds.has_entries = "ent" in data
        ds.h_nodes, ds.h_pairs = np.diff(slices["node"].numpy()), np.diff(slices["pair"].numpy())
        ds.h_ents = np.diff(slices["ent"].numpy()) if ds.has_entries else np.zeros(ds.G, np.int64)
        if ds.has_entries:
            ds.ent_rel, ds.ent, ds.ent_ptr = data["ent_rel"].to(dev), data["ent"].to(dev), slices["ent"].to(dev)
        else:
            ds.ent_ptr = torch.zeros(ds.G + 1, dtype=torch.int64, device=dev)
        ds.h_edges, ds.h_hop_pairs = meta["h_edges"].numpy(), meta["h_hop_pairs"].numpy()
        ds.index_bounds = dict(meta["index_bounds"])
        for k, v in data.items():
            if k.startswith("node."):
                ds.node_rows[k[5:]] = v.to(dev)
            elif k.startswith("graph."):
                ds.graph_rows[k[6:]] = v.to(dev)
        if meta["dict"] is not None:
            m = meta["dict"]
            ds.pdict = PeripheralDict(data["dict_rows"].to(dev), m["col_max"], m["T"], m["Hc"], data["dict_dominant"].to(dev))
            ds.uid = data["uid"].to(dev)
        ds.Nd = int(slices["node"][-1])
        return ds

    # ------------------------------------------------------------------------------------------------ collate
    def plan(self, ids):
        """Host side of a collate: the header (ids | node_base | pair_base | ent_base, int32) and the batch sizes."""
        ids = np.asarray(ids, dtype=np.int64).reshape(-1)
        if ids.size == 0 or ids.min() < 0 or ids.max() >= self.G:
            raise IndexError(f"graph ids outside [0, {self.G})")
        B = ids.size
        hdr = np.empty(4 * B + 3, dtype=np.int32)
        hdr[:B] = ids
        tot = []
        for j, cnt in enumerate((self.h_nodes, self.h_pairs, self.h_ents)):
            c = np.cumsum(cnt[ids])
            if c[-1] >= 2 ** 31:
                raise _lib.KpgnnError("batch exceeds the int32 index range")
            o = B + j * (B + 1)
            hdr[o] = 0
            hdr[o + 1:o + 1 + B] = c
            tot.append(int(c[-1]))
        return hdr, B, tot[0], tot[1], tot[2]

    def _view(self):
        v = _lib.DatasetView()
        v.K, v.G = self.K, self.G
        v.node_ptr, v.pair_ptr, v.ent_ptr = self.node_ptr.data_ptr(), self.pair_ptr.data_ptr(), self.ent_ptr.data_ptr()
        v.rowptr_dst, v.rowptr_src = self.rowptr_dst.data_ptr(), self.rowptr_src.data_ptr()
        v.col_dst, v.col_src = self.col_dst.data_ptr(), self.col_src.data_ptr()
        v.code_dst, v.code_src = self.code_dst.data_ptr(), self.code_src.data_ptr()
        if self.has_entries:
            v.ent_rel, v.ent = self.ent_rel.data_ptr(), self.ent.data_ptr()
        return v

    def _buffers(self, B, N, A, n_ent, hdr, prefixes=True):
        """Output buffers of one collate call for B graphs / N nodes / A pairs / n_ent entries (exact sizes, or the capacities
        of a StaticBatch), wired into a KHopBatch with its CSR, entry lists, dictionary ids and readout offsets in place, and
        the kpgnn_collate descriptor that fills them.  `hdr`: the device int32[4B+3] header the call will read."""
        dev = self.device
        K = self.K
        i32 = dict(dtype=torch.int32, device=dev)
        c = KHopCSR()
        c.N, c.K, c.A, c.device = N, K, A, dev
        c.max_code0, c.max_codek = self.max_code0, self.max_codek
        c._max_mult = self.max_mult if self.has_entries else None
        c.rowptr_dst, c.rowptr_src = torch.empty(N * K + 1, **i32), torch.empty(N * K + 1, **i32)
        c.col_dst, c.col_src = torch.empty(max(A, 1), **i32), torch.empty(max(A, 1), **i32)
        c.code_dst = torch.empty(max(A, 1), dtype=torch.int16, device=dev)
        c.code_src = torch.empty(max(A, 1), dtype=torch.int16, device=dev)
        c.nodes_per_tile = NT = self.NODES_PER_TILE
        batch_vec = torch.empty(N, dtype=torch.int64, device=dev)
        node_src = torch.empty(N, **i32)
        d = _lib.CollateDesc()
        d.ds = self._view()
        d.B, d.N, d.A, d.n_ent, d.hdr = B, N, A, n_ent, hdr.data_ptr()
        d.rowptr_dst, d.col_dst, d.code_dst = c.rowptr_dst.data_ptr(), c.col_dst.data_ptr(), c.code_dst.data_ptr()
        d.rowptr_src, d.col_src, d.code_src = c.rowptr_src.data_ptr(), c.col_src.data_ptr(), c.code_src.data_ptr()
        d.batch, d.node_src = batch_vec.data_ptr(), node_src.data_ptr()
        keep = [hdr, node_src]
        if self.has_entries:
            ntiles = (N + NT - 1) // NT
            c.tile_ptr, c.tile_pack = torch.empty(ntiles + 1, **i32), torch.empty(max(n_ent, 1), **i32)
            enp = torch.empty(N + 1, **i32)
            d.nodes_per_tile, d.tile_ptr, d.tile_pack, d.ent_node_ptr = NT, c.tile_ptr.data_ptr(), c.tile_pack.data_ptr(), enp.data_ptr()
            keep.append(enp)
            P = K - 1 if (prefixes and ntiles > 0) else 0
            if P > 0:
                pptr = torch.empty((P, ntiles + 1), **i32)
                ppack = torch.empty((P, max(n_ent, 1)), **i32)
                pscr = torch.empty((P, ntiles), **i32)
                d.num_prefix, d.prefix_ptr, d.prefix_pack, d.prefix_scratch = P, pptr.data_ptr(), ppack.data_ptr(), pscr.data_ptr()
                keep.append(pscr)
                c._tile_lists = {k: (pptr[k - 1], ppack[k - 1]) for k in range(1, K)}
        else:
            c.tile_ptr = c.tile_pack = None
        out = KHopBatch(num_graphs=B)
        rows = []
        for name, src in self.node_rows.items():
            dst = torch.empty((N,) + tuple(src.shape[1:]), dtype=src.dtype, device=dev)
            rows.append((src, dst))
            setattr(out, name, dst)
        uid = None
        if self.uid is not None:
            uid = torch.empty((N, K), **i32)
            rows.append((self.uid, uid))
        if len(rows) > 8 or len(self.graph_rows) > 8:
            raise _lib.KpgnnError("at most 8 node-level and 8 graph-level attributes per collate")
        d.n_node_rows = len(rows)
        for j, (src, dst) in enumerate(rows):
            d.node_rows[j].src, d.node_rows[j].dst = src.data_ptr(), dst.data_ptr()
            d.node_rows[j].row_bytes = src.element_size() * (src.numel() // max(src.shape[0], 1))
        d.n_graph_rows = len(self.graph_rows)
        for j, (name, src) in enumerate(self.graph_rows.items()):
            dst = torch.empty((B,) + tuple(src.shape[1:]), dtype=src.dtype, device=dev)
            d.graph_rows[j].src, d.graph_rows[j].dst = src.data_ptr(), dst.data_ptr()
            d.graph_rows[j].row_bytes = src.element_size() * (src.numel() // max(src.shape[0], 1))
            setattr(out, name, dst)
        c._keep = keep
        # integer node attributes arrive validated (dataset-wide maximum): the encoders skip their range check
        for name, hi in self.index_bounds.items():
            t = getattr(out, name, None)
            if t is not None:
                t._kp_index_bound = (t._version, hi)
        out.batch = batch_vec
        from .ops import _GPTR
        setattr(batch_vec, _GPTR, ((batch_vec._version, B), hdr[B:2 * B + 1]))
        c.graph_ptr, c.max_graph_nodes = hdr[B:2 * B + 1], int(self.h_nodes.max()) if self.G else 0
        out.edge_index = torch.empty((2, 0), dtype=torch.int64, device=dev)
        out.edge_attr = torch.empty((0, K), dtype=torch.int64, device=dev)
        out.csr = c
        attach_khop_csr(out.edge_index, out.edge_attr, N, c)
        if K > 1:
            out.pe_attr = _zero_pe(dev, K - 1, N)
        if uid is not None:
            uid._kp_dom = self.pdict.dominant
            out.peripheral_dict = BatchPeripheral(self.pdict, uid)
        return out, c, d

    def collate(self, ids, prefixes=True):
        """Batch of the graphs `ids` (dataset indices, in batch order) as a KHopBatch whose CSR, entry lists, dictionary ids
        and readout offsets are already in place: no layer call, encoder or readout pays a build or a host sync for it.
        `edge_index` / `edge_attr` of the result are zero-length HANDLES ([2,0] / [0,K]): the layers only use them to find
        the attached CSR (and to slice the hop prefix `edge_attr[:, :k]`), the [E]-sized int64 tensors are never made."""
        lib = _lib.load()
        dev = self.device
        hdr_h, B, N, A, n_ent = self.plan(ids)
        with torch.cuda.device(dev):
            hdr = torch.from_numpy(hdr_h).pin_memory().to(dev, non_blocking=True)
            out, c, d = self._buffers(B, N, A, n_ent, hdr, prefixes)
            c.E = int(self.h_edges[hdr_h[:B]].sum())
            hp = self.h_hop_pairs[hdr_h[:B]].sum(0)
            c._apairs = {k + 1: int(hp[k]) for k in range(self.K - 1)}
            _lib.check(lib.kpgnn_collate(ctypes.byref(d), torch.cuda.current_stream(dev).cuda_stream), "kpgnn_collate")
        out.num_khop_edges = c.E
        return out

    def capacities(self, B, sigmas=6.0):
        """(N, A, entries) that a batch of B graphs drawn without replacement stays below with overwhelming probability:
        B * mean + sigmas * std * sqrt(B) of the per-graph counts (+ one graph's maximum), never more than the B largest."""
        caps = []
        for cnt in (self.h_nodes, self.h_pairs, self.h_ents):
            c = np.asarray(cnt, dtype=np.float64)
            top = float(np.sort(c)[-min(B, c.size):].sum())
            est = B * c.mean() + sigmas * c.std() * np.sqrt(B) + c.max()
            caps.append(int(min(top, np.ceil(est))) if B < c.size else int(top))
        return tuple(caps)

    def static_batch(self, B, capacities=None):
        return StaticBatch(self, B, capacities or self.capacities(B))


class CapacityError(_lib.KpgnnError):
    pass


class StaticBatch:
    """A batch whose tensors have FIXED addresses and CAPACITY shapes, refilled in place by every collate: what a captured
    hipGraph of the training step needs.  Every tensor of `self.batch` is allocated for `N_cap` nodes; the number of LIVE
    nodes of the current batch sits in the device-side header (`self.live_nodes`, int32[1]) and reaches the kernels as
    their descriptors' n_dyn (ops.dynamic_rows), so that rows beyond it are never read, written or summed.

        sb = dataset.static_batch(B)
        with sb.dynamic():                        # every launch on N_cap rows gets the live count
            sb.stage(ids); sb.launch_collate(); step(sb.batch)        # eager, or inside a hipGraph capture
        ...
        sb.stage(ids_next); graph.replay()        # per step: 16 KB of header to the device, one graph launch
    """

    def __init__(self, ds, B, capacities):
        self.ds, self.B = ds, int(B)
        self.N_cap, self.A_cap, self.E_cap = (int(v) for v in capacities)
        dev = ds.device
        with torch.cuda.device(dev):
            self.hdr = torch.zeros(4 * self.B + 3, dtype=torch.int32, device=dev)
            self.batch, self.csr, self._desc = ds._buffers(self.B, self.N_cap, self.A_cap, self.E_cap, self.hdr)
        self.csr.E = self.A_cap
        self.csr._apairs = {k: self.A_cap for k in range(1, ds.K)}          # (byte accounting only: upper bounds)
        self.live_nodes = self.hdr[2 * self.B:2 * self.B + 1]
        self.live = None            # (N, A, entries) of the staged batch

    def dynamic(self):
        from .ops import dynamic_rows
        return dynamic_rows(self.live_nodes, self.N_cap)

    def stage(self, ids):
        """Host side of a collate: plan the batch and send its header (asynchronously, stream-ordered).  Raises CapacityError
        when the batch does not fit the static buffers (the caller then takes an exact-shape `dataset.collate(ids)` step)."""
        hdr_h, B, N, A, n_ent = self.ds.plan(ids)
        if B != self.B:
            raise ValueError(f"StaticBatch built for {self.B} graphs, got {B}")
        if N > self.N_cap or A > self.A_cap or n_ent > self.E_cap:
            raise CapacityError(f"batch of {N} nodes / {A} pairs / {n_ent} entries exceeds the static capacity "
                                f"{self.N_cap} / {self.A_cap} / {self.E_cap}")
        with torch.cuda.device(self.ds.device):
            self.hdr.copy_(torch.from_numpy(hdr_h).pin_memory(), non_blocking=True)
        self.live = (N, A, n_ent)

    def launch_collate(self):
        """Device side: kpgnn_collate into the static buffers (capturable: its arguments never change)."""
        dev = self.ds.device
        with torch.cuda.device(dev):
            _lib.check(_lib.load().kpgnn_collate(ctypes.byref(self._desc), torch.cuda.current_stream(dev).cuda_stream), "kpgnn_collate")


def _zero_pe(dev, width, n):
    """[n, width] view of a persistent all-zero int64 buffer: the reference's pe_attr is always zero (Q1), and a view of a
    buffer that is known to be zero needs neither a fill per batch nor the layers' one-off all-zero check."""
    key = (dev, width)
    buf = _ZERO_PE.get(key)
    if buf is None or buf.shape[0] < n:
        cap = max(n, 1 << 16) if buf is None else max(n, 2 * buf.shape[0])
        buf = torch.zeros((cap, width), dtype=torch.int64, device=dev)
        setattr(buf, _ZATTR, (buf._version, True))
        _ZERO_PE[key] = buf
    return buf[:n]
