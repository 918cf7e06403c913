"""Build libdifusco_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m difusco_amd.build [--force]

The library is built IN-TREE (difusco_amd/lib/) so that it travels with a snapshot of the repo.
"""
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libdifusco_hip.so")
SOURCES = ["linear.hip", "linear_split.hip", "edge_layer.hip", "edge_layer_bf16.hip", "edge_layer_abl.hip", "graph_kernels.hip", "decode.hip", "two_opt.hip", "knn.hip", "mis_decode.hip", "api.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "kernels.h"), os.path.join(CSRC, "edge_layer_common.h"), os.path.join(CSRC, "edge_layer_kernel.h"),
           os.path.join(os.path.dirname(PKG), "include", "difusco_hip.h")]


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into one shared library; returns its path."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    os.makedirs(LIB_DIR, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
    # one hipcc per source, in parallel (the fused edge-layer files hold many template instantiations), then one link;
    # objects live in a temporary directory: only the .so stays in the tree
    with tempfile.TemporaryDirectory(prefix="difusco_build_") as tmp:
        def compile_one(src):
            obj = os.path.join(tmp, os.path.splitext(src)[0] + ".o")
            cmd = [hipcc] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            res = subprocess.run(cmd, capture_output=True, text=True)
            if res.returncode != 0:
                raise RuntimeError(f"hipcc failed on {src}:\n" + res.stdout + res.stderr)
            return obj
        with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as pool:
            objs = list(pool.map(compile_one, SOURCES))
        cmd = [hipcc] + flags + ["-shared", "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
