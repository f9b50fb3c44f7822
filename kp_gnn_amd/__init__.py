"""kp_gnn_amd - MI355X-native K-hop message-passing hot path of KP-GNN.

Product code.  The arithmetic runs in libkpgnn_hip.so (hand-written HIP for gfx950, C ABI in
include/kpgnn.h); this package is the Python host side that mirrors the reference's layer surface
(layers/KPGIN.py, KPGINplus.py, KPGCN.py, gine.py, combine.py, layer_utils.py).  There is no CPU
fallback: without the library, or on CPU tensors, the ops raise.
"""
from . import _lib  # noqa: F401
from ._lib import KpgnnError  # noqa: F401

__all__ = ["KpgnnError"]
