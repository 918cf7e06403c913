"""``torch.ops.difusco.*``: the denoise step registered as PyTorch custom ops (BASELINE.json north_star).

The ops are a thin C++ shim (``csrc/torch_ops.cpp``, ``TORCH_LIBRARY(difusco, ...)``) over the C ABI of
``libdifusco_hip.so`` - the C ABI stays the primary boundary, this is the PyTorch-facing registration of it:

    torch.ops.difusco.prepare_graph(edge_index_cpu, n_nodes)      -> rowptr, col, row, perm, identity   (host)
    torch.ops.difusco.weights_layout(hidden, n_layers, C)          -> int64 offsets (+ total)
    torch.ops.difusco.workspace_bytes(hidden, n_layers, N, E, S)   -> int
    torch.ops.difusco.prepare_state(blob, points, N, E, S, workspace, cfg) -> uint8 buffer;  time_bias_rows(blob, times, cfg) -> [n_t, L, H]
    torch.ops.difusco.denoise_step_categorical(...) / denoise_step_gaussian(...) -> (xt_next, pred, prob)

These ops are the DEFAULT binding of ``DenoiseEngine`` / ``TSPModel`` / ``MISModel`` (``backend=None``); ``backend="ctypes"`` routes
through ctypes instead (and is chosen automatically with the profiling library / a ``DIFUSCO_HIP_LIBRARY`` build, because the shim is
linked against the production library).  Both launch the same kernels on the current stream, are bitwise identical (GPU tests) and
equally fast (profiles/r04/bench_binding_ab.txt).  ``prepare_state`` / ``time_bias_rows`` are the ops of the optional prepared state
(include/difusco_hip.h: ``difusco_prepare``, ``difusco_time_bias_rows``)."""
import os

import torch

from .build import TORCH_LIB_PATH

_loaded = False


def load():
    """Load libdifusco_torch.so (once) and return the ``torch.ops.difusco`` namespace.  No fallback: raises if the
    library has not been built (``python -m difusco_amd.build``)."""
    global _loaded
    if not _loaded:
        if not os.path.exists(TORCH_LIB_PATH):
            raise RuntimeError(f"{TORCH_LIB_PATH} is missing: build it with `python -m difusco_amd.build`")
        torch.ops.load_library(TORCH_LIB_PATH)
        _loaded = True
    return torch.ops.difusco
