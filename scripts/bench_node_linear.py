#!/usr/bin/env python
"""Node-row linear ([m,256] x [256,1024], fp16x3) alone, on the GPU box: the production kernel (node_linear.hip), the general
split kernel of rounds 1-2 and the timing ablations of the profiling library (difusco_debug_set keys 8 / 10).  Launches back to
back on one stream, HIP events around 200 of them.  Prints one JSON line.
  DIFUSCO_PROFILING_LIB=1 python scripts/bench_node_linear.py [m]"""
import ctypes
import json
import os
import sys

import torch

os.environ.setdefault("DIFUSCO_PROFILING_LIB", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from difusco_amd import _lib as L, weights  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
k, n_out = 256, 1024
x = torch.randn(m, k, generator=g).to(dev)
w = torch.randn(n_out, k, generator=g) / 16
planes = weights.split_planes(w, per_row=True).to(dev)
b = torch.randn(n_out, generator=g).to(dev)
y, rs = torch.empty(m, n_out, device=dev), torch.ones(m, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
lib = L.lib()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(reps):
    for _ in range(reps):
        L.check(lib.difusco_linear_rows_split(p(x), p(planes), L.PRECISIONS["fp16x3"], p(b), None, p(y), m, k, n_out, n_out, p(rs), st))


out = {"m": m, "k": k, "n_out": n_out, "note": "us per call = row_pow2_scale pass (~5 us at m = 8000) + the linear kernel; four passes over the cases in rotated order", "cases": {}}
CASES = (("node_linear.hip", ((8, 0), (10, 0))), ("general split kernel, lookahead 4", ((8, 4), (10, 0))),
         ("variant: direct lane = row loads of x", ((8, 0), (10, 16))),
         ("ablation: no stores", ((8, 0), (10, 1))), ("ablation: no x loads", ((8, 0), (10, 2))),
         ("ablation: no weight stream / mfma", ((8, 0), (10, 4))), ("ablation: no stores, no x loads", ((8, 0), (10, 3))),
         ("ablation: no stores, no weights", ((8, 0), (10, 5))), ("ablation: launch only", ((8, 0), (10, 7))))
run(300)      # clocks up
for rot in range(4):          # four passes, the order rotated: the position in the sequence moves a case by up to 1.5 us
    order = CASES[rot * 3 % len(CASES):] + CASES[:rot * 3 % len(CASES)]
    for name, sets in order:
        for key, val in sets:
            assert lib.difusco_debug_set(key, val) >= 0
        run(20)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(500)
        e1.record()
        torch.cuda.synchronize()
        out["cases"].setdefault(name, []).append(round(e0.elapsed_time(e1) * 1e3 / 500, 2))
print(json.dumps(out))
