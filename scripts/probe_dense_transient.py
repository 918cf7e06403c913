#!/usr/bin/env python
"""the first 10 timed steps of the dense TSP-50 x 16 sub-record ran 8x slower than the next 20 after 4 warm-up
steps.  What is it?  Per-step wall times (device sync after every step) of that workload after idle gaps of 0 / 2 / 8 s, with the
engine clock read from the hwmon file beside every step; then the same for TSP-500 x 16.

    python scripts/probe_dense_transient.py
"""
import glob
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
print("hwmon dirs:", glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
    print(d, sorted(os.listdir(d))[:60])
pr = torch.cuda.get_device_properties(dev)
print("props:", {k: getattr(pr, k) for k in dir(pr) if k.startswith("pci") or k in ("name", "multi_processor_count")})
freq = (glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input") or [None])[0]
powf = ((glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") + glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input")) or [None])[0]


def rd(f, scale):
    try:
        return round(float(open(f).read()) * scale)
    except Exception:
        return None


params = random_state_dict(256, 12, 2, seed=20240926)
eng = DenoiseEngine(params, device=dev)
margs = dict(diffusion_type="categorical", diffusion_schedule="linear", diffusion_steps=1000, inference_diffusion_steps=50,
             inference_schedule="cosine", sparse_factor=-1, n_layers=12, hidden_dim=256, inference_trick="ddim")
sched = InferenceSchedule("cosine", T=1000, inference_T=50)
gen = torch.Generator().manual_seed(1)


def run(name, model, points, ei, xt, gaps=(0.0, 2.0, 8.0), n=24):
    model.prepare_schedule([sched(i)[0] for i in range(50)])
    for i in range(4):
        t1, t2 = sched(i)
        xt = model.categorical_denoise_step(points, xt, np.array([t1]), dev, ei, target_t=np.array([t2]))
    torch.cuda.synchronize()
    for gap in gaps:
        time.sleep(gap)
        ts, cl = [], []
        for i in range(n):
            t1, t2 = sched((4 + i) % 49)
            t0 = time.perf_counter()
            xt = model.categorical_denoise_step(points, xt, np.array([t1]), dev, ei, target_t=np.array([t2]))
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
            cl.append(rd(freq, 1e-6))
        # and the same steps enqueued back to back (what bench.py times)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(10):
            t1, t2 = sched((4 + i) % 49)
            xt = model.categorical_denoise_step(points, xt, np.array([t1]), dev, ei, target_t=np.array([t2]))
        torch.cuda.synchronize()
        b2b = 1e2 * (time.perf_counter() - t0)
        print(f"{name} after {gap:.0f} s idle: per-step ms (sync each) {[round(t, 2) for t in ts]}\n    sclk MHz {cl}\n    then 10 steps back to back: {b2b:.3f} ms/step, power {rd(powf, 1e-6)} W", flush=True)
    return xt


m = TSPModel(margs, engine=eng, seed=1)
pts = torch.rand(16, 50, 2, generator=gen).to(dev)
x = (torch.randn(16, 50, 50, generator=gen) > 0).float().to(dev)
run("tsp50dense x16", m, pts, None, x)
# long idle like the CPU-oracle leg of the previous workload, then ONLY 4 warm-up steps + 3 x 10 timed steps (bench.py's pattern)
time.sleep(20.0)
for i in range(4):
    t1, t2 = sched(i)
    x = m.categorical_denoise_step(pts, x, np.array([t1]), dev, None, target_t=np.array([t2]))
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(10):
        t1, t2 = sched((4 + i) % 49)
        x = m.categorical_denoise_step(pts, x, np.array([t1]), dev, None, target_t=np.array([t2]))
    torch.cuda.synchronize()
    print(f"bench pattern after 20 s idle, repetition {rep}: {1e2 * (time.perf_counter() - t0):.3f} ms/step  sclk {rd(freq, 1e-6)}", flush=True)

margs2 = dict(margs, sparse_factor=50)
m2 = TSPModel(margs2, engine=eng, seed=2)
p2, e2 = tsp_batch_gpu(500, 50, range(16), dev)
x2 = (torch.randn(e2.shape[1], generator=gen) > 0).float().to(dev)
run("tsp500 x16", m2, p2, e2, x2, gaps=(0.0, 8.0), n=12)
