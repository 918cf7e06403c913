"""ctypes binding of the C ABI in include/difusco_hip.h.  No fallback: if the HIP library is missing
the import of anything that computes fails loudly (the product path never routes through a CPU path)."""
import ctypes
import os

from .build import LIB_PATH, PROF_LIB_PATH

FLAG_NO_L0_FOLD, FLAG_NO_TAIL_FOLD, FLAG_CHECK_FINITE = 1, 2, 4      # difusco_step_args.flags

ABI_VERSION = 12
TASK_TSP, TASK_MIS = 0, 1
CATEGORICAL, GAUSSIAN = 0, 1
RAND_NONE, RAND_INJECTED, RAND_PHILOX = 0, 1, 2
PREC_FP32, PREC_BF16X3, PREC_BF16X6, PREC_FP16X3 = 0, 1, 2, 3
PRECISIONS = {"fp32": PREC_FP32, "bf16x3": PREC_BF16X3, "bf16x6": PREC_BF16X6, "fp16x3": PREC_FP16X3}
AGGREGATIONS = {"sum": 0, "mean": 1, "max": 2}      # DIFUSCO_AGG_* (--aggregation, train.py:52; gnn_encoder.py:170-191)

# indices into difusco_weights_layout() (mirrors the enums of include/difusco_hip.h)
W_GLOBAL = ["node_embed.weight", "node_embed.bias", "edge_embed.weight", "edge_embed.bias",
            "time_embed.0.weight", "time_embed.0.bias", "time_embed.2.weight", "time_embed.2.bias",
            "out.0.weight", "out.0.bias", "out.2.weight", "out.2.bias",
            "@time_freqs", "@dimt_pos", "@dimt_scalar", "@planes:edge_embed.weight"]
W_LAYER = ["@node4.weight", "@node4.bias", "layers.{l}.C.weight", "layers.{l}.C.bias",
           "layers.{l}.norm_h.weight", "layers.{l}.norm_h.bias", "layers.{l}.norm_e.weight", "layers.{l}.norm_e.bias",
           "time_embed_layers.{l}.1.weight", "time_embed_layers.{l}.1.bias",
           "per_layer_out.{l}.0.weight", "per_layer_out.{l}.0.bias",
           "per_layer_out.{l}.2.weight", "per_layer_out.{l}.2.bias",
           "@planes:layers.{l}.C.weight", "@planes:per_layer_out.{l}.2.weight", "@planes:@node4.weight",
           "@fused_scales", "@node4.fused_bias", "@node4.fused_scale"]


class StepArgs(ctypes.Structure):
    """difusco_step_args (include/difusco_hip.h)."""
    _fields_ = [
        ("struct_size", ctypes.c_uint32), ("abi_version", ctypes.c_uint32),
        ("hidden", ctypes.c_int32), ("n_layers", ctypes.c_int32),
        ("out_channels", ctypes.c_int32), ("task", ctypes.c_int32),
        ("weights", ctypes.c_void_p),
        ("n_nodes", ctypes.c_int32), ("n_edges", ctypes.c_int32),
        ("rowptr", ctypes.c_void_p), ("col", ctypes.c_void_p), ("perm", ctypes.c_void_p),
        ("n_segments", ctypes.c_int32), ("seg_ptr", ctypes.c_void_p),
        ("points", ctypes.c_void_p), ("xt", ctypes.c_void_p),
        ("t", ctypes.c_float), ("xt_is_binary", ctypes.c_int32),
        ("diffusion", ctypes.c_int32), ("post", ctypes.c_float * 8),
        ("rand_mode", ctypes.c_int32), ("rand", ctypes.c_void_p),
        ("seed", ctypes.c_uint64), ("offset", ctypes.c_uint64),
        ("xt_out", ctypes.c_void_p), ("pred_out", ctypes.c_void_p), ("prob_out", ctypes.c_void_p),
        ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_size_t), ("stream", ctypes.c_void_p),
        ("precision", ctypes.c_int32), ("no_fusion", ctypes.c_int32),
        ("row", ctypes.c_void_p),
        ("gn_phase", ctypes.c_int32), ("flags", ctypes.c_int32), ("gn_sums", ctypes.c_void_p),
        ("prepared", ctypes.c_void_p), ("tbias", ctypes.c_void_p),      # optional prepared state (ABI 9)
        ("aggregation", ctypes.c_int32), ("reserved0", ctypes.c_int32),   # DIFUSCO_AGG_* (ABI 10)
        ("gen_table", ctypes.c_void_p),                                     # optional generated-input table (ABI 12)
    ]


class DifuscoHipError(RuntimeError):
    pass


_lib = None


def lib():
    """The loaded shared library (loads on first use; raises if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    # DIFUSCO_PROFILING_LIB=1 loads the profiling build (libdifusco_hip_prof.so: the same code plus the timing-only
    # kernel variants and their process-wide knobs, `python -m difusco_amd.build --prof`); never set in production
    path = PROF_LIB_PATH if os.environ.get("DIFUSCO_PROFILING_LIB", "0") not in ("", "0") else LIB_PATH
    path = os.environ.get("DIFUSCO_HIP_LIBRARY", path)      # A/B of two builds of the same ABI (benchmarking only)
    if not os.path.exists(path):
        raise DifuscoHipError(
            f"{path} is missing: build it with `python -m difusco_amd.build` "
            "(there is deliberately no CPU / PyTorch fallback)")
    L = ctypes.CDLL(path)
    i32, i64, vp, f32p = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p
    L.difusco_abi_version.restype = ctypes.c_int
    L.difusco_last_error.restype = ctypes.c_char_p
    L.difusco_weights_layout.argtypes = [i32, i32, i32, ctypes.POINTER(i64), i32, ctypes.POINTER(i64)]
    L.difusco_csr_from_coo_host.argtypes = [vp, i64, i64, vp, vp, vp, vp, ctypes.POINTER(ctypes.c_int)]
    L.difusco_workspace_bytes.restype = ctypes.c_size_t
    L.difusco_workspace_bytes.argtypes = [i32, i32, i32, i32, i32]
    L.difusco_denoise_step.argtypes = [ctypes.POINTER(StepArgs)]
    L.difusco_prepared_bytes.restype = ctypes.c_size_t
    L.difusco_prepared_bytes.argtypes = [i32, i32]
    L.difusco_prepare.argtypes = [ctypes.POINTER(StepArgs), vp, ctypes.c_size_t]
    L.difusco_time_bias_rows.argtypes = [i32, i32, i32, f32p, ctypes.POINTER(ctypes.c_float), i32, f32p, vp]
    L.difusco_edge_embed.argtypes = [i32, i32, i32, f32p, i32, f32p, vp, i64, f32p, f32p, f32p, vp]
    L.difusco_gen_table_bytes.restype = ctypes.c_size_t
    L.difusco_gen_table_bytes.argtypes = [i32]
    L.difusco_gen_table_build.argtypes = [i32, i32, i32, f32p, f32p, ctypes.c_size_t, vp]
    L.difusco_linear_rows.argtypes = [f32p, f32p, f32p, f32p, f32p, i64, i32, i32, i64, vp]
    L.difusco_linear_rows_split.argtypes = [f32p, vp, i32, f32p, f32p, f32p, i64, i32, i32, i64, f32p, vp]
    L.difusco_fused_scratch_bytes.restype = ctypes.c_size_t
    L.difusco_fused_scratch_bytes.argtypes = [i32, i32]
    L.difusco_edge_layer_fused.argtypes = [i32, i32, i32, vp, vp, vp, f32p, f32p, f32p, vp, vp] + [f32p] * 9 + [i32, f32p, vp, vp]
    L.difusco_edge_gate_aggregate.argtypes = [i32, i32, vp, vp, f32p, f32p, f32p] + [f32p] * 7 + [i32, vp]
    L.difusco_categorical_posterior.argtypes = [f32p, f32p, ctypes.POINTER(ctypes.c_float), i32, f32p,
                                                ctypes.c_uint64, ctypes.c_uint64, f32p, f32p, i64, vp]
    L.difusco_gaussian_posterior.argtypes = [f32p, f32p, ctypes.POINTER(ctypes.c_float), i32, f32p,
                                             ctypes.c_uint64, ctypes.c_uint64, f32p, i64, vp]
    L.difusco_tsp_merge_workspace_bytes.argtypes = [i64, ctypes.POINTER(ctypes.c_size_t)]
    L.difusco_tsp_merge_tour.argtypes = [i32, i64, vp, vp, f32p, f32p, vp, ctypes.c_size_t, vp,
                                         ctypes.POINTER(i64), ctypes.POINTER(i32), vp]
    L.difusco_tsp_merge_tours.argtypes = [i32, i64, vp, vp, f32p, f32p, i32, vp, ctypes.c_size_t, vp, vp, vp, vp]
    L.difusco_tsp_two_opt_workspace_bytes.argtypes = [i32, i32, ctypes.POINTER(ctypes.c_size_t)]
    L.difusco_tsp_two_opt.argtypes = [i32, i32, vp, vp, i64, vp, ctypes.c_size_t, ctypes.POINTER(i64), vp]
    L.difusco_knn_graph_workspace_bytes.argtypes = [i32, i32, ctypes.POINTER(ctypes.c_size_t)]
    L.difusco_knn_graph.argtypes = [i32, i32, vp, i64, vp, vp, vp, ctypes.c_size_t, vp]
    L.difusco_mis_decode_workspace_bytes.argtypes = [i32, ctypes.POINTER(ctypes.c_size_t)]
    L.difusco_mis_decode.argtypes = [i32, vp, vp, f32p, vp, vp, ctypes.c_size_t, ctypes.POINTER(i32), vp]
    L.difusco_host_rowsum_f32.argtypes = [f32p, i32, ctypes.POINTER(ctypes.c_float)]
    L.difusco_mcts_heatmap_workspace_bytes.argtypes = [i32, i64, ctypes.POINTER(ctypes.c_size_t)]
    L.difusco_mcts_heatmap_prepare.argtypes = [i32, i64, vp, vp, f32p, f32p, ctypes.c_double, vp, ctypes.c_size_t,
                                               ctypes.POINTER(ctypes.c_float), vp]
    L.difusco_mcts_heatmap_rows.argtypes = [i32, i64, f32p, vp, ctypes.c_size_t, i32, i32, f32p, vp]
    if path == PROF_LIB_PATH:
        # profiling library only: DIFUSCO_DEBUG_SET="key=value,key=value" applies difusco_debug_set at load time, so that
        # whole test runs / benches can be pointed at an A/B kernel variant (e.g. "7=8051")
        L.difusco_debug_set.argtypes = [i32, i32]
        for kv in filter(None, os.environ.get("DIFUSCO_DEBUG_SET", "").split(",")):
            k_, v_ = kv.split("=")
            if L.difusco_debug_set(int(k_), int(v_)) < 0:
                raise DifuscoHipError(f"difusco_debug_set({kv}) failed: {L.difusco_last_error().decode()}")
    if L.difusco_abi_version() != ABI_VERSION:
        raise DifuscoHipError(f"ABI version mismatch: library {L.difusco_abi_version()} != binding {ABI_VERSION}")
    L._difusco_path = path
    _lib = L
    return L


def loaded_path() -> str:
    """Path of the shared library behind ``lib()`` (production, profiling build or DIFUSCO_HIP_LIBRARY)."""
    return lib()._difusco_path


def check(code: int):
    if code < 0:
        raise DifuscoHipError(f"libdifusco_hip error {code}: {lib().difusco_last_error().decode()}")
    return code


def weights_layout(hidden: int, n_layers: int, out_channels: int):
    """(offsets, total_floats) of the packed weight blob - the C library is the source of truth."""
    n = len(W_GLOBAL) + n_layers * len(W_LAYER)
    off = (ctypes.c_int64 * n)()
    tot = ctypes.c_int64()
    got = check(lib().difusco_weights_layout(hidden, n_layers, out_channels, off, n, ctypes.byref(tot)))
    if got != n:
        raise DifuscoHipError(f"weight layout has {got} entries, binding expects {n}")
    return list(off), tot.value
