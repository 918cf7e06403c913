"""The inference flow of ``TSPModel.test_step`` (``difusco/pl_tsp_model.py:153-256``) and ``MISModel.test_step``
(``difusco/pl_mis_model.py:142-209``) without Lightning, every stage on the GPU path of this package: k-NN graph ->
``sequential_sampling`` rounds of ``parallel_sampling`` noise samples through the denoising loop -> heatmap -> greedy
tour merge -> batched 2-opt -> best tour.  One instance per call, like the reference (its test batch size is 1); the
parallel samples form the batch of the denoise steps (disjoint union, ``duplicate_edge_index``), the sequential rounds
repeat the whole loop with fresh noise and stack the results (``pl_tsp_model.py:185,238,240``).

This is host-side orchestration only - each stage is one of the drop-in entry points (``graph.knn_edge_index_gpu``,
``TSPModel.sample``, ``decode.merge_tours``, ``decode.batched_two_opt_torch``) and can be used on its own."""
import time
from typing import Dict, Optional

import numpy as np
import torch

from .decode import batched_two_opt_torch, merge_tours
from .graph import knn_edge_index_gpu


def tour_length(points: np.ndarray, tour) -> float:
    """``TSPEvaluator.evaluate`` (``utils/tsp_utils.py:148-156``) without the N x N distance matrix."""
    t = np.asarray(tour)
    return float(np.linalg.norm(points[t[:-1]] - points[t[1:]], axis=1).sum())


def _ticker(timings, dev):
    def tick(name, t0):
        if timings is not None:
            torch.cuda.synchronize(dev)
            timings[name] = timings.get(name, 0.0) + time.perf_counter() - t0
    return tick


def solve_tsp(model, points: np.ndarray, sparse_factor: int, parallel_sampling: int = 1, two_opt_iterations: int = 1000,
              generator: Optional[torch.Generator] = None, timings: Optional[Dict[str, float]] = None,
              sequential_sampling: int = 1):
    """points: float64/float32 [N,2] of ONE instance.  ``sparse_factor`` > 0: k-NN graph (the sparse models);
    <= 0: dense mode (``pl_tsp_model.py:158-160``, TSP-50/100).  Returns (best_tour list, best_cost, all_costs, info):
    ``all_costs`` has ``parallel_sampling * sequential_sampling`` entries in the reference's stacking order, info holds
    merge_iterations / 2-opt moves of the LAST round - the quantities the reference logs (``pl_tsp_model.py:244-251``)."""
    dev = model.device
    pts64 = np.ascontiguousarray(points, dtype=np.float64)
    n = pts64.shape[0]
    sparse = sparse_factor is not None and sparse_factor > 0
    tick = _ticker(timings, dev)

    t0 = time.perf_counter()
    edge_index = knn_edge_index_gpu(pts64, sparse_factor, device=dev) if sparse else None    # tsp_graph_dataset.py:53-62
    tick("knn", t0)
    # the reference's np_points are the float32 coordinates of the batch (graph_data.x / points tensor); merge, 2-opt
    # (after .astype("float64")) and the cost evaluation all start from these rounded values (pl_tsp_model.py:159,172,233,240)
    pts32 = torch.from_numpy(pts64.astype(np.float32)).to(dev)
    np_points = pts32.cpu().numpy()
    np_points64 = np_points.astype(np.float64)
    if sparse:
        pts_rep = pts32.repeat(parallel_sampling, 1)                                        # :178-183
        ei_rep = model.duplicate_edge_index(edge_index, n, dev, copies=parallel_sampling) if parallel_sampling > 1 else edge_index
    else:
        pts_rep, ei_rep = pts32.reshape(1, n, 2).repeat(parallel_sampling, 1, 1), None

    stacked, merged_costs = [], []
    merge_iterations, ns = 0.0, 0
    for _ in range(sequential_sampling):                                                    # :185
        t0 = time.perf_counter()
        heat = model.sample(pts_rep, ei_rep, generator=generator)                           # :186-222, on the device
        tick("sampling", t0)
        t0 = time.perf_counter()
        tours, merge_iterations = merge_tours(heat, pts32, edge_index, sparse_graph=sparse,  # :226-230
                                              parallel_sampling=parallel_sampling, device=dev)
        tick("merge", t0)
        t0 = time.perf_counter()
        solved, ns = batched_two_opt_torch(np_points64, np.asarray(tours, dtype=np.int64),   # :233-236
                                           max_iterations=two_opt_iterations, device=dev)
        tick("two_opt", t0)
        stacked.append(solved)
        merged_costs += [tour_length(np_points64, t) for t in tours]
    solved = np.concatenate(stacked, axis=0)                                                # :238
    costs = [tour_length(np_points64, t) for t in solved]                                   # :240-246
    best = int(np.argmin(costs))
    return solved[best].tolist(), costs[best], costs, {"merge_iterations": merge_iterations, "two_opt_iterations": ns,
                                                       "merged_costs": merged_costs}


def solve_mis(model, n_nodes: int, edge_index, parallel_sampling: int = 1, generator: Optional[torch.Generator] = None,
              timings: Optional[Dict[str, float]] = None, sequential_sampling: int = 1):
    """``MISModel.test_step`` (``difusco/pl_mis_model.py:142-206``): ``sequential_sampling`` rounds of
    ``parallel_sampling`` noise samples of ONE graph through the denoising loop (disjoint union), greedy decode of every
    sample, best = largest set.  ``edge_index``: int64 [2,E] in the dataset's layout (both directions + self loops).
    Returns (best 0/1 array, best size, sizes).  Upstream re-duplicates ``edge_index`` inside the sequential loop
    (``pl_mis_model.py:168-169``), which breaks ``parallel > 1 and sequential > 1`` there; here the duplication happens
    once, which is what that combination means."""
    from .decode import mis_decode_np
    dev = model.device
    ei = edge_index if isinstance(edge_index, torch.Tensor) else torch.from_numpy(np.asarray(edge_index))
    ei = ei.to(dev)
    tick = _ticker(timings, dev)
    ei_rep = model.duplicate_edge_index(ei, n_nodes, dev, copies=parallel_sampling) if parallel_sampling > 1 else ei   # pl_mis_model.py:168-169
    graph = model.prepare_graph(ei_rep, n_nodes * parallel_sampling)
    sols = []
    for _ in range(sequential_sampling):                                                          # :156
        t0 = time.perf_counter()
        scores = model.sample(n_nodes * parallel_sampling, ei_rep, generator=generator)           # :157-192
        tick("sampling", t0)
        t0 = time.perf_counter()
        sols.append(mis_decode_np(scores, graph=graph, device=dev).reshape(parallel_sampling, n_nodes))   # :195-198
        tick("decode", t0)
    sol = np.concatenate(sols, axis=0)
    sizes = sol.sum(axis=1)
    best = int(np.argmax(sizes))
    return sol[best], int(sizes[best]), sizes.tolist()
