#!/usr/bin/env python
"""How many torch threads should the CPU-oracle leg of bench.py use on this host?  Times one denoise step of the oracle
on one TSP-1000 / K=100 graph (H=256, 12 layers) per thread count.  Test infrastructure (imports oracle/)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import difusco_oracle as O  # noqa: E402

p = O.init_params(256, 12, 2, seed=1)
pts, ei = O.tsp_instance(1000, 100, seed=1000)
pts, ei = torch.from_numpy(pts), torch.from_numpy(ei)
xt = (torch.randn(ei.shape[1], generator=torch.Generator().manual_seed(0)) > 0).float()
tab = O.CategoricalTables()
for th in [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64, 128]:
    torch.set_num_threads(th)
    with torch.no_grad():
        t0 = time.perf_counter()
        O.tsp_categorical_denoise_step(p, tab, pts, xt, 500, ei, 469, generator=torch.Generator().manual_seed(1))
        dt = time.perf_counter() - t0
    print(f"threads {th:4d}: {dt:7.2f} s/step = {1 / dt:.3f} graph-steps/s", flush=True)
