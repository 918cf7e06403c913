"""Heatmap -> tour on the GPU box: drop-in for ``merge_tours`` of the reference
(``difusco/utils/tsp_utils.py:89-145``), sparse (k-NN) and dense heatmaps.

Same signature and return value as the reference function: ``(tours, merge_iterations)`` with one closed tour
(list starting and ending at node 0) per parallel sample and the mean of the per-sample iteration counters.  The
work goes through ``difusco_tsp_merge_tours`` of libdifusco_hip.so (pair keys, scores and the two sorts on the GPU,
the reference's route bookkeeping on the host); there is no CPU fallback.  Keyword-only extensions: ``device``,
``return_completed`` (adds the per-sample flag that says whether the tour was assembled from positive-score
candidate pairs, the regime pinned against the reference)."""
import ctypes

import numpy as np
import torch

from . import _lib


def _dev(x, dtype, device):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    return x.to(device=device, dtype=dtype).contiguous()


def merge_tours(adj_mat, np_points, edge_index_np, sparse_graph=False, parallel_sampling=1, *, device="cuda:0",
                return_completed=False):
    """``adj_mat``: [parallel_sampling * E] (or [parallel_sampling, E]) heat values in the edge order of
    ``edge_index_np`` ([2, E], the ONE graph's edges); ``np_points`` [N, 2].  numpy arrays or torch tensors."""
    device = torch.device(device)
    L = _lib.lib()
    pts = _dev(np_points, torch.float32, device)
    n = pts.shape[0]
    if not sparse_graph:
        # dense heatmaps [parallel_sampling, N, N] (tsp_utils.py:105-108: adj_mat[0] + adj_mat[0].T): the same greedy
        # insertion over the complete directed edge list - entry (i, j) at i * N + j, so that the pair sums
        # fl32(A_ij + A_ji), the doubled diagonal and the order of equal keys are those of the dense matrix
        idx = torch.arange(n, dtype=torch.int32, device=device)
        ei = torch.stack([idx.repeat_interleave(n), idx.repeat(n)])
    else:
        ei = _dev(edge_index_np, torch.int32, device)
    heat = _dev(adj_mat, torch.float32, device).reshape(parallel_sampling, -1)
    E = ei.shape[1]
    if heat.shape[1] != E:
        raise ValueError(f"adj_mat holds {heat.shape[1]} values per sample for {E} edges")
    nbytes = ctypes.c_size_t()
    _lib.check(L.difusco_tsp_merge_workspace_bytes(E, ctypes.byref(nbytes)))
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=device)
    stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    row, col = ei[0].contiguous(), ei[1].contiguous()
    # one C call for all samples of the graph (the pair-key sort is shared; the per-sample loop of tsp_utils.py:100-145 runs
    # inside the library)
    tours_np = np.empty((parallel_sampling, n + 1), dtype=np.int32)
    iters = np.zeros(parallel_sampling, dtype=np.int64)
    done_np = np.zeros(parallel_sampling, dtype=np.int32)
    _lib.check(L.difusco_tsp_merge_tours(n, E, ctypes.c_void_p(row.data_ptr()), ctypes.c_void_p(col.data_ptr()),
                                         ctypes.c_void_p(heat.data_ptr()), ctypes.c_void_p(pts.data_ptr()), parallel_sampling,
                                         ctypes.c_void_p(ws.data_ptr()), nbytes.value,
                                         tours_np.ctypes.data_as(ctypes.c_void_p), iters.ctypes.data_as(ctypes.c_void_p),
                                         done_np.ctypes.data_as(ctypes.c_void_p), stream))
    tours = [t.tolist() for t in tours_np]
    done = [bool(v) for v in done_np]
    merge_iterations = float(np.mean(iters))
    return (tours, merge_iterations, done) if return_completed else (tours, merge_iterations)


def batched_two_opt_torch(points, tour, max_iterations=1000, device="cuda:0"):
    """Drop-in for ``batched_two_opt_torch`` of the reference (``difusco/utils/tsp_utils.py:12-49``): ``points``
    float64 [N,2] numpy, ``tour`` int [B, N+1] numpy (closed tours over the same points); returns
    ``(tour int64 numpy [B, N+1], iterator)``.  Runs ``difusco_tsp_two_opt``; GPU only."""
    device = torch.device(device)
    if device.type != "cuda":
        raise _lib.DifuscoHipError("batched_two_opt_torch of difusco_amd runs on the GPU only (no CPU fallback)")
    L = _lib.lib()
    pts = _dev(np.asarray(points, dtype=np.float64), torch.float64, device)
    tours = _dev(np.asarray(tour), torch.int32, device)
    if tours.dim() != 2 or tours.shape[1] != pts.shape[0] + 1:
        raise ValueError("tour must be [batch, N + 1] over the N points")
    n, batch = pts.shape[0], tours.shape[0]
    nbytes = ctypes.c_size_t()
    _lib.check(L.difusco_tsp_two_opt_workspace_bytes(n, batch, ctypes.byref(nbytes)))
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=device)
    it = ctypes.c_int64()
    _lib.check(L.difusco_tsp_two_opt(n, batch, ctypes.c_void_p(pts.data_ptr()), ctypes.c_void_p(tours.data_ptr()),
                                     int(max_iterations), ctypes.c_void_p(ws.data_ptr()), nbytes.value, ctypes.byref(it),
                                     ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)))
    return tours.cpu().numpy().astype(np.int64), int(it.value)


def mis_decode_np(predictions, adj_matrix=None, *, graph=None, edge_index=None, device="cuda:0"):
    """Drop-in for ``mis_decode_np`` of the reference (``difusco/utils/mis_utils.py:3-18``): ``predictions`` [N] node
    scores (numpy or tensor), ``adj_matrix`` a scipy sparse adjacency (as built at ``pl_mis_model.py:152-154``).
    Returns the 0/1 int numpy array.  Instead of a scipy matrix the caller may pass the ``CsrGraph`` of the denoise
    steps (``graph=``, no host round trip) or the ``edge_index`` the adjacency was built from.  GPU only."""
    from .graph import build_csr
    device = torch.device(device)
    L = _lib.lib()
    scores = _dev(predictions, torch.float32, device).reshape(-1)
    n = scores.shape[0]
    if graph is None:
        if edge_index is None:
            coo = adj_matrix.tocoo()
            edge_index = np.stack([coo.row, coo.col]).astype(np.int64)
        graph = build_csr(edge_index if isinstance(edge_index, torch.Tensor) else torch.from_numpy(np.asarray(edge_index)),
                          n, device)
    nbytes = ctypes.c_size_t()
    _lib.check(L.difusco_mis_decode_workspace_bytes(n, ctypes.byref(nbytes)))
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=device)
    sol = torch.empty(n, dtype=torch.int32, device=device)
    rounds = ctypes.c_int32()
    _lib.check(L.difusco_mis_decode(n, ctypes.c_void_p(graph.rowptr.data_ptr()), ctypes.c_void_p(graph.col.data_ptr()),
                                    ctypes.c_void_p(scores.data_ptr()), ctypes.c_void_p(sol.data_ptr()),
                                    ctypes.c_void_p(ws.data_ptr()), nbytes.value, ctypes.byref(rounds),
                                    ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)))
    return sol.cpu().numpy().astype(int)
