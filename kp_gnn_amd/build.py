"""Build the native libraries in-tree (hipcc cross-compiles gfx950 without a GPU).

    python -m kp_gnn_amd.build            # build what is stale
    python -m kp_gnn_amd.build --force

Outputs (git-ignored, shipped to the GPU box by gpurun):
    kp_gnn_amd/libkpgnn_hip.so    HIP kernels + the C ABI of include/kpgnn.h
    kp_gnn_amd/libkpgnn_host.so   host-side (CPU) C ABI of include/kpgnn_host.h (K-hop pre-transform)
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(ROOT, "include")
OBJ = os.path.join(PKG, "csrc", "_obj")

HIP_LIB = os.path.join(PKG, "libkpgnn_hip.so")
HOST_LIB = os.path.join(PKG, "libkpgnn_host.so")

HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
CXX = os.environ.get("CXX") or shutil.which("g++") or "g++"
HIP_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-I" + INCLUDE, "-I" + CSRC]
HOST_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-fopenmp", "-I" + INCLUDE, "-I" + CSRC, "-Wall"]


def _sources(ext):
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(ext))


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    return hs


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def _compile(compiler, flags, src):
    obj = os.path.join(OBJ, os.path.basename(src) + ".o")
    if _stale(obj, [src] + _headers()):
        _run([compiler] + flags + ["-c", src, "-o", obj])
    return obj


def build_hip(force=False):
    srcs = _sources(".hip")
    if not force and not _stale(HIP_LIB, srcs + _headers()):
        return HIP_LIB
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            if f.endswith(".hip.o"):
                os.remove(os.path.join(OBJ, f))
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(max(2, (os.cpu_count() or 4) - 1), len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(HIPCC, HIP_FLAGS, s), srcs))
    _run([HIPCC, "-shared", "--offload-arch=gfx950", "-o", HIP_LIB] + objs)
    return HIP_LIB


def build_host(force=False):
    srcs = _sources(".cpp")
    if not srcs:
        return None
    if not force and not _stale(HOST_LIB, srcs + _headers()):
        return HOST_LIB
    os.makedirs(OBJ, exist_ok=True)
    objs = [_compile(CXX, HOST_FLAGS, s) for s in srcs]
    _run([CXX, "-shared", "-fopenmp", "-o", HOST_LIB] + objs)
    return HOST_LIB


def build_all(force=False):
    return build_hip(force), build_host(force)


if __name__ == "__main__":
    out = build_all(force="--force" in sys.argv)
    print("built:", out)
