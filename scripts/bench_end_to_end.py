#!/usr/bin/env python
"""Whole inference flow of the reference's test_step on the GPU path (difusco_amd.pipeline.solve_tsp): k-NN graph ->
50-step sampling of `parallel_sampling` noise samples -> merge -> 2-opt, per-stage wall time.  Random-init weights
(no checkpoints offline): the tours are only as good as 2-opt makes them; the point is the time split.  One JSON line."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from difusco_amd.models import TSPModel  # noqa: E402
from difusco_amd.pipeline import solve_tsp  # noqa: E402
from difusco_amd.synthetic import random_state_dict  # noqa: E402

dev = torch.device("cuda:0")
out = {"data": "synthetic, random-init weights", "cases": []}
for n, k, par, cap in ((1000, 100, 8, 1000), (10000, 100, 1, 1000)):
    params = random_state_dict(256, 12, 2, seed=1)
    args = dict(diffusion_type="categorical", diffusion_schedule="linear", diffusion_steps=1000, sparse_factor=k, n_layers=12,
                hidden_dim=256, inference_trick="ddim", inference_diffusion_steps=50, inference_schedule="cosine")
    m = TSPModel(args, params, device=dev, seed=7)
    pts = np.random.default_rng(n).random((n, 2))
    solve_tsp(m, pts, k, parallel_sampling=par, two_opt_iterations=2)                     # warm-up
    t = {}
    tour, cost, costs, info = solve_tsp(m, pts, k, parallel_sampling=par, two_opt_iterations=cap, timings=t)
    out["cases"].append({"workload": f"TSP-{n} K={k}, parallel_sampling={par}, 50 steps, 2-opt cap {cap}",
                         "seconds": {a: round(b, 4) for a, b in t.items()}, "total_s": round(sum(t.values()), 4),
                         "two_opt_moves": info["two_opt_iterations"], "merge_iterations": info["merge_iterations"],
                         "best_cost": cost, "merged_cost_mean": float(np.mean(info["merged_costs"]))})
print(json.dumps(out))
