"""CPU: the C-ABI library builds, loads and exports every symbol include/*.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(kpgnn_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def built():
    from kp_gnn_amd import build
    return build.build_all()


def test_hip_library_exports_every_declared_symbol(built):
    hip_lib, _ = built
    names = _declared("kpgnn.h")
    assert {"kpgnn_csr_build", "kpgnn_aggregate_fwd", "kpgnn_aggregate_bwd"} <= set(names)
    lib = ctypes.CDLL(hip_lib)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in kpgnn.h but not exported"
    lib.kpgnn_abi_version.restype = ctypes.c_int
    assert lib.kpgnn_abi_version() == 1


def test_ctypes_binding_covers_header(built):
    from kp_gnn_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared("kpgnn.h")
    _lib.load()  # resolves every symbol, checks the ABI version


def test_argument_validation_without_gpu(built):
    """Entry points reject malformed descriptors before touching the device."""
    from kp_gnn_amd import _lib
    lib = _lib.load()
    d = _lib.AggFwdDesc()
    d.N, d.K, d.D, d.K_csr = 4, 3, 8, 2  # K > K_csr
    assert lib.kpgnn_aggregate_fwd(ctypes.byref(d), None) == -1
    assert b"bad N=" in lib.kpgnn_last_error()
    assert lib.kpgnn_aggregate_fwd(None, None) == -1
    b = _lib.AggBwdDesc()
    b.N, b.K, b.D, b.K_csr, b.mode = 4, 2, 8, 2, 7
    assert lib.kpgnn_aggregate_bwd(ctypes.byref(b), None) == -1
    assert lib.kpgnn_csr_stats(None, 0, None, 0, 5, 2, None, None) == -1


def test_product_refuses_cpu_tensors(built):
    import torch
    from kp_gnn_amd import KpgnnError
    from kp_gnn_amd.layers import KPGINConv
    layer = KPGINConv(8, 8, 2, num_hop1_edge=1, num_pe=3)
    with pytest.raises(KpgnnError):
        layer(torch.randn(3, 8), torch.tensor([[0, 1], [1, 2]]), torch.tensor([[2, 0], [0, 2]]))


def test_product_does_not_import_oracle():
    import subprocess
    import sys
    code = ("import sys; import kp_gnn_amd, kp_gnn_amd.layers, kp_gnn_amd.ops; "
            "bad=[m for m in sys.modules if m=='oracle' or m.startswith('oracle.')]; assert not bad, bad")
    subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "kp_gnn_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
