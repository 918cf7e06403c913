#!/usr/bin/env python
"""Decode (heatmap -> tour) timing on the GPU box: difusco_amd.decode.merge_tours (GPU sorts + host bookkeeping) with
the heat already on the device, beside the CPU oracle (dense N x N restatement of the reference's merge_tours +
merge_cython) on a bounded sample.  Prints one JSON line.  The oracle is only the cpu_baseline here."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from difusco_amd.decode import batched_two_opt_torch, merge_tours  # noqa: E402
from difusco_amd.synthetic import tsp_instance  # noqa: E402

dev = torch.device("cuda:0")
out = {"metric": "tours decoded per second (one sample per call)", "unit": "tours/s", "data": "synthetic", "cases": []}
for n, k, reps, cpu in ((1000, 100, 20, True), (10000, 100, 5, False)):
    pts, ei = tsp_instance(n, k, seed=11)
    rng = np.random.default_rng(n)
    d = np.linalg.norm(pts[ei[0]] - pts[ei[1]], axis=1)
    heat = (np.exp(-d / (0.5 * d.mean())) * rng.random(ei.shape[1])).astype(np.float32) + np.float32(1e-6)
    heat_d, pts_d, ei_d = torch.from_numpy(heat).to(dev), torch.from_numpy(pts).to(dev), torch.from_numpy(ei).to(dev)
    merge_tours(heat_d, pts_d, ei_d, sparse_graph=True, device=dev)           # warm-up (rocPRIM kernels, allocator)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        tours, it, done = merge_tours(heat_d, pts_d, ei_d, sparse_graph=True, device=dev, return_completed=True)
    dt = (time.perf_counter() - t0) / reps
    case = {"workload": f"TSP-{n} K={k} ({ei.shape[1]} heat entries)", "ms_per_tour": 1e3 * dt, "value": 1.0 / dt,
            "completed_within_candidates": bool(done[0]), "merge_iterations": it}
    if cpu:
        from oracle import tsp_decode_oracle as D
        t0 = time.perf_counter()
        ref_tours, _, _ = D.merge_tours(heat, pts, ei, sparse_graph=True)
        dtc = time.perf_counter() - t0
        case["cpu_baseline"] = {"value": 1.0 / dtc, "unit": "tours/s", "cores": 1, "kind": "port",
                                "sample": f"1 tour, dense {n}x{n} numpy argsort + Python bookkeeping ({dtc:.2f} s)"}
        case["equals_cpu_oracle"] = bool(ref_tours == tours)
    # 2-opt on the decoded tour (tsp_utils.py:12-49): time per applied move, float64
    tour0 = np.asarray(tours, dtype=np.int64)
    cap = 200 if n <= 1000 else 50
    batched_two_opt_torch(pts.astype(np.float64), tour0, max_iterations=2, device=dev)       # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    refined, moves = batched_two_opt_torch(pts.astype(np.float64), tour0, max_iterations=cap, device=dev)
    dt2 = time.perf_counter() - t0
    length = lambda t: float(np.linalg.norm(pts[t[:-1]] - pts[t[1:]], axis=1).sum())
    case["two_opt"] = {"moves": moves, "cap": cap, "ms_total": 1e3 * dt2, "ms_per_move": 1e3 * dt2 / max(moves, 1),
                       "pairs_per_move": n * (n - 3) // 2, "tour_length_before": length(tour0[0]),
                       "tour_length_after": length(refined[0])}
    if cpu:
        t0 = time.perf_counter()
        ref_refined, ref_moves = D.batched_two_opt(pts.astype(np.float64), tour0, max_iterations=10)
        dtc = time.perf_counter() - t0
        case["two_opt"]["cpu_baseline"] = {"ms_per_move": 1e3 * dtc / max(ref_moves, 1), "cores": 1, "kind": "port",
                                           "sample": f"{ref_moves} moves of the numpy restatement ({dtc:.2f} s)"}
    # MCTS heatmap text (tsp_mcts/convert_numpy_to_txt.py): numeric part on the GPU, %.6f formatting + file on the host;
    # beside it the host numpy sweeps of the same module (the reference converter itself needs five dense N x N arrays)
    from difusco_amd import formats
    import tempfile
    pts32 = torch.from_numpy(pts.astype(np.float32)).to(dev)
    list(formats.mcts_heatmap_rows_gpu(heat_d, ei_d, pts32, n, 0.02, device=dev))               # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_rows = sum(1 for _ in formats.mcts_heatmap_rows_gpu(heat_d, ei_d, pts32, n, 0.02, device=dev))
    t_rows = time.perf_counter() - t0
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        t0 = time.perf_counter()
        path = formats.write_mcts_heatmap(heat_d, pts32, n, tmp, 0, edge_index=ei_d, use_gpu=True)
        t_file = time.perf_counter() - t0
        size = os.path.getsize(path)
    case["mcts_text"] = {"rows_to_host_s": t_rows, "rows": n_rows, "file_s": t_file, "file_bytes": size,
                         "note": "rows_to_host = GPU numeric part + device-to-host copy of the N^2 floats; file = + %.6f formatting"}
    if cpu:
        t0 = time.perf_counter()
        sum(1 for _ in formats.mcts_heatmap_rows(heat, ei, pts.astype(np.float32), n, 0.02))
        case["mcts_text"]["host_numpy_rows_s"] = time.perf_counter() - t0
    out["cases"].append(case)
print(json.dumps(out))
