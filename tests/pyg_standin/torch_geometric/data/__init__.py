"""Stand-in for torch_geometric.data (test-only; see package docstring)."""
import torch


class Data:
    """Attribute bag: `key in data` <=> attribute present and not None; num_nodes = x.size(0)."""

    def __init__(self, **kwargs):
        object.__setattr__(self, "_store", {})
        for k, v in kwargs.items():
            setattr(self, k, v)

    def __setattr__(self, key, value):
        if value is None:
            self._store.pop(key, None)
        else:
            self._store[key] = value

    def __getattr__(self, key):
        if key.startswith("__"):
            raise AttributeError(key)
        store = object.__getattribute__(self, "_store")
        if key == "num_nodes" and "num_nodes" not in store:
            if "x" in store:
                return store["x"].size(0)
            return int(store["edge_index"].max().item()) + 1
        return store.get(key, None)

    def __contains__(self, key):
        return key in self._store

    @property
    def keys(self):
        return list(self._store.keys())

    def to(self, device):
        for k, v in list(self._store.items()):
            if torch.is_tensor(v):
                self._store[k] = v.to(device)
        return self


class Batch(Data):
    """Block-diagonal collate: node-level tensors concatenated, edge_index offset by node counts."""

    @staticmethod
    def from_data_list(data_list):
        out = Batch()
        keys = data_list[0].keys
        offset = 0
        cat = {k: [] for k in keys}
        batch = []
        for gi, d in enumerate(data_list):
            n = d.num_nodes
            for k in keys:
                v = getattr(d, k)
                if k == "edge_index":
                    v = v + offset
                cat[k].append(v)
            batch.append(torch.full((n,), gi, dtype=torch.long))
            offset += n
        for k in keys:
            dim = 1 if k == "edge_index" else 0
            setattr(out, k, torch.cat(cat[k], dim=dim))
        out.batch = torch.cat(batch)
        out.num_graphs = len(data_list)
        return out
