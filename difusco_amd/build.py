"""Build libdifusco_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m difusco_amd.build [--force] [--prof]

The library is built IN-TREE (difusco_amd/lib/) so that it travels with a snapshot of the repo.  ``--prof`` also builds
libdifusco_hip_prof.so: the same sources with -DDIFUSCO_PROFILING plus edge_layer_abl.hip - the timing-only kernel
variants (ablations, phase stamps, A/B option sets) and the process-wide knobs that select them.  None of that is in
the production library.
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
PROF_LIB_PATH = os.path.join(LIB_DIR, "libdifusco_hip_prof.so")
SOURCES = ["linear.hip", "linear_split.hip", "node_linear.hip", "edge_embed.hip", "edge_layer.hip", "edge_layer_bf16.hip", "graph_kernels.hip", "decode.hip", "two_opt.hip", "knn.hip", "mis_decode.hip", "formats.hip", "api.hip"]
PROF_SOURCES = SOURCES + ["edge_layer_abl.hip", "stage_lab.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "kernels.h"), os.path.join(CSRC, "edge_layer_common.h"), os.path.join(CSRC, "edge_layer_kernel.h"),
           os.path.join(os.path.dirname(PKG), "include", "difusco_hip.h")]


# per-source compiler options.  The fused edge-layer translation units are compiled with LLVM's iterative-maxocc
# instruction scheduler: with the default scheduler the kernels need 256 VGPRs and still spill 44-120 B per lane; with this
# one they need 234-242 and no scratch, and the step is 3.7 % faster (DIFUSCO_FUSED_SCHED=default restores the default).
_FUSED_SCHED = os.environ.get("DIFUSCO_FUSED_SCHED", "iterative-maxocc")
EXTRA_FLAGS = {}
# No packed fp32 arithmetic (v_pk_add / v_pk_mul / v_pk_fma_f32) in the fused edge-layer kernels.  On gfx950 a packed fp32
# operation has the throughput of two plain ones (the vector peak, 256 flop/clk/CU, is already reached by v_fma_f32) and is an
# anti-lever beside MFMAs (MI355X_MICROARCH.md, "price of one filler": one v_pk_fma_f32 costs +22 cycles against two v_fma_f32);
# the kernel issues ~1,600 of them per tile next to the partner wave's matrix phases.  Same-box A/B (profiles/r04/
# exp_packed_fp32.txt): TSP-1000 +2.3 %, MIS +2.2 %, TSP-10000 +2.0 %, TSP-500 +1.6 %.  The source keeps its v2f pair
# arithmetic: with the target feature off the backend splits every pair operation into two plain ones on the adjacent
# registers - no moves (DIFUSCO_FUSED_PACKED_FP32=1 restores the packed instructions for an A/B).
NO_PK = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
_FUSED_NO_PK = os.environ.get("DIFUSCO_FUSED_PACKED_FP32", "0") in ("", "0")
for _src in ("edge_layer.hip", "edge_layer_bf16.hip", "edge_layer_abl.hip"):
    EXTRA_FLAGS[_src] = (["-mllvm", "-amdgpu-sched-strategy=" + _FUSED_SCHED] if _FUSED_SCHED != "default" else []) + \
                        (NO_PK if _FUSED_NO_PK else [])

# formats.hip reproduces a numpy float32 program bit for bit: IEEE divide / square root, no multiply-add contraction
EXTRA_FLAGS["formats.hip"] = ["-fhip-fp32-correctly-rounded-divide-sqrt", "-ffp-contract=off"]

TORCH_LIB_PATH = os.path.join(LIB_DIR, "libdifusco_torch.so")
TORCH_SRC = os.path.join(CSRC, "torch_ops.cpp")


def build_torch_ops(force: bool = False, verbose: bool = False) -> str:
    """The PyTorch custom-op shim (csrc/torch_ops.cpp: ``torch.ops.difusco.*`` over the C ABI), compiled with the host
    compiler against this interpreter's torch and linked to libdifusco_hip.so next to it.  In-tree, like the HIP library."""
    deps = [TORCH_SRC, HEADERS[-1], LIB_PATH]
    if not force and os.path.exists(TORCH_LIB_PATH) and all(os.path.getmtime(d) <= os.path.getmtime(TORCH_LIB_PATH) for d in deps):
        return TORCH_LIB_PATH
    import torch
    from torch.utils import cpp_extension
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ([os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-w", TORCH_SRC, "-o", TORCH_LIB_PATH]
           + [f"-I{p}" for p in cpp_extension.include_paths()] + [f"-I{os.path.join(rocm_root(), 'include')}", "-D__HIP_PLATFORM_AMD__=1",
              "-DUSE_ROCM=1", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch.compiled_with_cxx11_abi())}", f"-L{tlib}", "-ltorch",
              "-ltorch_cpu", "-lc10", "-ltorch_hip", "-lc10_hip", f"-L{LIB_DIR}", "-ldifusco_hip", "-Wl,-rpath,$ORIGIN",
              f"-Wl,-rpath,{tlib}"])
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("building libdifusco_torch.so failed:\n" + res.stdout + res.stderr)
    return TORCH_LIB_PATH


def _stale(lib_path, sources) -> bool:
    if not os.path.exists(lib_path):
        return True
    t = os.path.getmtime(lib_path)
    deps = [os.path.join(CSRC, s) for s in sources] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _hipcc() -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    return hipcc if os.path.exists(hipcc) else "hipcc"


def rocm_root() -> str:
    """The ROCm installation the compiler in use belongs to (ROCM_PATH, else two levels above hipcc, else /opt/rocm)."""
    if os.environ.get("ROCM_PATH"):
        return os.environ["ROCM_PATH"]
    import shutil
    exe = shutil.which(_hipcc()) or _hipcc()
    root = os.path.dirname(os.path.dirname(os.path.realpath(exe)))
    return root if os.path.isdir(os.path.join(root, "include")) else "/opt/rocm"


# A/B builds of the production sources with different compiler options (benchmarking only; loaded through
# DIFUSCO_HIP_LIBRARY=<path>): name -> (extra flags, the sources they apply to; None = every source)
VARIANTS = {
    "pk_fused": (["-Xclang", "-target-feature", "-Xclang", "+packed-fp32-ops"], ("edge_layer.hip", "edge_layer_bf16.hip")),   # round 3
    "nopk_all": (NO_PK, None),
    # the fused translation units under LLVM's other instruction-scheduling strategies (round 5 A/B; all compile without scratch)
    "sched_maxilp": (["-mllvm", "-amdgpu-sched-strategy=max-ilp"], ("edge_layer.hip", "edge_layer_bf16.hip")),
    "sched_memclause": (["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"], ("edge_layer.hip", "edge_layer_bf16.hip")),
    "sched_iterilp": (["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"], ("edge_layer.hip", "edge_layer_bf16.hip")),
}


def variant_path(name: str) -> str:
    return os.path.join(LIB_DIR, f"libdifusco_hip_{name}.so")


def build(force: bool = False, verbose: bool = False, prof: bool = False, variant: str = None) -> str:
    """Compile every HIP source for gfx950 into one shared library; returns its path.  prof: the profiling library;
    variant: one of VARIANTS (an A/B build of the production sources, libdifusco_hip_<variant>.so)."""
    lib_path, sources = (PROF_LIB_PATH, PROF_SOURCES) if prof else (LIB_PATH, SOURCES)
    var_flags, var_sources = [], ()
    if variant is not None:
        lib_path = variant_path(variant)
        var_flags, var_sources = VARIANTS[variant]
    if not force and not _stale(lib_path, sources):
        return lib_path
    hipcc = _hipcc()
    os.makedirs(LIB_DIR, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + (["-DDIFUSCO_PROFILING=1"] if prof else [])
    # one hipcc per source, in parallel (the fused edge-layer files hold many template instantiations), then one link;
    # objects live in a temporary directory: only the .so stays in the tree
    with tempfile.TemporaryDirectory(prefix="difusco_build_") as tmp:
        def compile_one(src):
            obj = os.path.join(tmp, os.path.splitext(src)[0] + ".o")
            extra = EXTRA_FLAGS.get(src, []) + (var_flags if (var_sources is None or src in var_sources) else [])
            if sum(f.startswith("-amdgpu-sched-strategy=") for f in extra) > 1:      # a variant's strategy replaces the default one
                first = next(i for i, f in enumerate(extra) if f.startswith("-amdgpu-sched-strategy="))
                extra = extra[:first - 1] + extra[first + 1:]
            if src == "stage_lab.hip@nopk":      # the stage-loop laboratory a second time, without packed fp32 arithmetic
                src, extra = "stage_lab.hip", extra + NO_PK + ["-DDIFUSCO_LAB_NOPK=1"]
                obj = os.path.join(tmp, "stage_lab_nopk.o")
            cmd = [hipcc] + flags + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            res = subprocess.run(cmd, capture_output=True, text=True)
            if res.returncode != 0:
                raise RuntimeError(f"hipcc failed on {src}:\n" + res.stdout + res.stderr)
            return obj
        todo = list(sources) + (["stage_lab.hip@nopk"] if prof else [])
        with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 4)) as pool:
            objs = list(pool.map(compile_one, todo))
        cmd = [hipcc] + flags + ["-shared", "-o", lib_path] + objs
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
    return lib_path


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_torch_ops(force="--force" in sys.argv, verbose=True))
    if "--prof" in sys.argv:
        print(build(force="--force" in sys.argv, verbose=True, prof=True))
    for name in VARIANTS:
        if "--variant=" + name in sys.argv:
            print(build(force="--force" in sys.argv, verbose=True, variant=name))
