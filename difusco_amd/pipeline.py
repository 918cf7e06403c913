"""The inference flow of ``TSPModel.test_step`` (``difusco/pl_tsp_model.py:152-241``) without Lightning, every stage on
the GPU path of this package: k-NN graph -> ``parallel_sampling`` noise samples through the 50-step denoising loop
-> heatmap -> greedy tour merge -> batched 2-opt -> best tour.  One instance per call, like the reference (its test
batch size is 1); the parallel samples form the batch of the denoise steps (disjoint union, ``duplicate_edge_index``).

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


def solve_tsp(model, points: np.ndarray, sparse_factor: int, parallel_sampling: int = 1, two_opt_iterations: int = 1000,
              generator: Optional[torch.Generator] = None, timings: Optional[Dict[str, float]] = None):
    """points: float64/float32 [N,2] of ONE instance.  Returns (best_tour list, best_cost, all_costs, info) where info
    holds merge_iterations / 2-opt moves, the quantities the reference logs (``pl_tsp_model.py:244-251``)."""
    dev = model.device
    pts64 = np.ascontiguousarray(points, dtype=np.float64)
    n = pts64.shape[0]

    def tick(name, t0):
        if timings is not None:
            torch.cuda.synchronize(dev)
            timings[name] = timings.get(name, 0.0) + time.perf_counter() - t0

    t0 = time.perf_counter()
    edge_index = knn_edge_index_gpu(pts64, sparse_factor, device=dev)                       # tsp_graph_dataset.py:53-62
    tick("knn", t0)
    t0 = time.perf_counter()
    pts32 = torch.from_numpy(pts64.astype(np.float32)).to(dev)
    pts_rep = pts32.repeat(parallel_sampling, 1)                                            # pl_tsp_model.py:178-183
    model.args.parallel_sampling = parallel_sampling    # duplicate_edge_index reads it from the args, like the reference
    ei_rep = model.duplicate_edge_index(edge_index, n, dev) if parallel_sampling > 1 else edge_index
    heat = model.sample(pts_rep, ei_rep, generator=generator)                               # :185-222, on the device
    tick("sampling", t0)
    t0 = time.perf_counter()
    tours, merge_iterations = merge_tours(heat, pts32, edge_index, sparse_graph=True,       # :226-230
                                          parallel_sampling=parallel_sampling, device=dev)
    tick("merge", t0)
    t0 = time.perf_counter()
    solved, ns = batched_two_opt_torch(pts64, np.asarray(tours, dtype=np.int64),            # :233-236
                                       max_iterations=two_opt_iterations, device=dev)
    tick("two_opt", t0)
    costs = [tour_length(pts64, t) for t in solved]
    best = int(np.argmin(costs))
    return solved[best].tolist(), costs[best], costs, {"merge_iterations": merge_iterations, "two_opt_iterations": ns,
                                                       "merged_costs": [tour_length(pts64, t) for t in tours]}


def solve_mis(model, n_nodes: int, edge_index, parallel_sampling: int = 1, generator: Optional[torch.Generator] = None,
              timings: Optional[Dict[str, float]] = None):
    """``MISModel.test_step`` (``difusco/pl_mis_model.py:142-206``): ``parallel_sampling`` noise samples of ONE graph
    through the denoising loop (disjoint union), greedy decode of every sample, best = largest set.  ``edge_index``:
    int64 [2,E] in the dataset's layout (both directions + self loops).  Returns (best 0/1 array, best size, sizes)."""
    from .decode import mis_decode_np
    dev = model.device
    ei = edge_index if isinstance(edge_index, torch.Tensor) else torch.from_numpy(np.asarray(edge_index))
    ei = ei.to(dev)

    def tick(name, t0):
        if timings is not None:
            torch.cuda.synchronize(dev)
            timings[name] = timings.get(name, 0.0) + time.perf_counter() - t0

    t0 = time.perf_counter()
    model.args.parallel_sampling = parallel_sampling
    ei_rep = model.duplicate_edge_index(ei, n_nodes, dev) if parallel_sampling > 1 else ei        # pl_mis_model.py:168-169
    scores = model.sample(n_nodes * parallel_sampling, ei_rep, generator=generator)               # :171-192
    tick("sampling", t0)
    t0 = time.perf_counter()
    graph = model.prepare_graph(ei_rep, n_nodes * parallel_sampling)
    sol = mis_decode_np(scores, graph=graph, device=dev).reshape(parallel_sampling, n_nodes)      # :195-196, one call
    tick("decode", t0)
    sizes = sol.sum(axis=1)
    best = int(np.argmax(sizes))
    return sol[best], int(sizes[best]), sizes.tolist()
