"""difusco_amd - MI355X-native (gfx950) denoising-diffusion sampler for DIFUSCO's TSP/MIS inference
path.  The compute lives in ``lib/libdifusco_hip.so`` (hand-written HIP, C ABI in
``include/difusco_hip.h``); this package is the Python host side mirroring the reference's
``categorical_denoise_step`` / ``gaussian_denoise_step`` interface."""
from .schedules import CategoricalDiffusion, GaussianDiffusion, InferenceSchedule  # noqa: F401

__all__ = ["CategoricalDiffusion", "GaussianDiffusion", "InferenceSchedule", "TSPModel", "MISModel",
           "DenoiseEngine", "build_csr", "knn_edge_index_gpu", "merge_tours", "batched_two_opt_torch", "mis_decode_np",
           "solve_tsp", "solve_mis"]


def __getattr__(name):  # heavy imports (ctypes library, torch) on first use
    if name in ("TSPModel", "MISModel", "COMetaModel"):
        from . import models
        return getattr(models, name)
    if name == "DenoiseEngine":
        from .engine import DenoiseEngine
        return DenoiseEngine
    if name in ("build_csr", "CsrGraph", "complete_graph_batch", "knn_edge_index_gpu"):
        from . import graph
        return getattr(graph, name)
    if name in ("merge_tours", "batched_two_opt_torch", "mis_decode_np"):
        from . import decode
        return getattr(decode, name)
    if name in ("solve_tsp", "solve_mis"):
        from . import pipeline
        return getattr(pipeline, name)
    raise AttributeError(name)
