#!/usr/bin/env python
"""How much would hipGraph capture of a denoise step buy?  One TSP-1000 x 8 categorical step (the bench workload) is captured
with torch.cuda.CUDAGraph (hipGraph underneath) and replayed; the replay is timed against the same number of direct calls.
The replayed step repeats ONE (t, target_t, Philox offset): timing only.  GPU only."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from difusco_amd.engine import DenoiseEngine  # noqa: E402
from difusco_amd.models import TSPModel  # noqa: E402
from difusco_amd.schedules import InferenceSchedule  # noqa: E402
from difusco_amd.synthetic import random_state_dict, tsp_batch_gpu  # noqa: E402

dev = torch.device("cuda:0")
H, LAYERS, N, K, G, STEPS = 256, 12, 1000, 100, 8, 30
params = random_state_dict(H, LAYERS, 2, seed=20240926)
engine = DenoiseEngine(params, device=dev, precision="fp16x3")
margs = dict(diffusion_type="categorical", diffusion_schedule="linear", diffusion_steps=1000, inference_diffusion_steps=50,
             inference_schedule="cosine", sparse_factor=K, n_layers=LAYERS, hidden_dim=H, inference_trick="ddim")
model = TSPModel(margs, engine=engine, seed=1234)
points, edge_index = tsp_batch_gpu(N, K, range(G), dev)
xt = (torch.randn(edge_index.shape[1]) > 0).float().to(dev)
sched = InferenceSchedule("cosine", T=1000, inference_T=50)


def step(i, x):
    t1, t2 = sched(i % 49)
    return model.categorical_denoise_step(points, x, np.array([t1]), dev, edge_index, target_t=np.array([t2]))


for i in range(5):
    xt = step(i, xt)
torch.cuda.synchronize()
res = {}
for rep in range(2):
    t0 = time.perf_counter()
    x = xt
    for i in range(STEPS):
        x = step(5 + i, x)
    torch.cuda.synchronize()
    res[f"direct_ms_per_step_{rep}"] = 1e3 * (time.perf_counter() - t0) / STEPS
xt = x
graph = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.graph(graph, stream=side):
    out = step(7, xt)          # xt is the model's own last output: known binary without a device check
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    for i in range(STEPS):
        graph.replay()
    torch.cuda.synchronize()
    res[f"graph_replay_ms_per_step_{rep}"] = 1e3 * (time.perf_counter() - t0) / STEPS
# same fixed step called directly (same t as the captured one), for a like-for-like comparison
t0 = time.perf_counter()
for i in range(STEPS):
    o2 = step(7, xt)
torch.cuda.synchronize()
res["direct_fixed_t_ms_per_step"] = 1e3 * (time.perf_counter() - t0) / STEPS
res["workload"] = f"TSP-{N} K={K} x {G} graphs categorical, H={H} L={LAYERS}, {STEPS} steps per timing"
print(json.dumps(res))
