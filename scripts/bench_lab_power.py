#!/usr/bin/env python
"""Is the matrix-core rate of the edge GEMM set by socket power?  GEMM 1 of the fused edge layer alone (stage_lab.hip,
PROFILING library, variant 142020 = the production geometry) runs for ~5 s per case while rocm-smi is sampled every 0.3 s:

    randn   the e stream holds N(0, 1) values (what the bench uses): every mantissa bit of both planes toggles
    zeros   the e stream is all zeros: same instructions, same memory traffic, the multipliers see one zero operand

Same instruction stream, same bytes - if the time per launch differs, the difference is the power / clock governor.

    python scripts/bench_lab_power.py [variant] [E]
"""
import ctypes
import json
import os
import re
import subprocess
import sys
import threading
import time

import torch

os.environ["DIFUSCO_PROFILING_LIB"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from difusco_amd import _lib, graph, weights  # noqa: E402

variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [142020]      # several: randn only, alternating
variant = variants[0]
E = int(sys.argv[2]) if len(sys.argv) > 2 else 800_000
H = 256
dev = torch.device("cuda:0")
L = _lib.lib()
L.difusco_lab_gemm1_nopk.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_float,
                                     ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
gen = torch.Generator().manual_seed(0)
Wc = (torch.rand(H, H, generator=gen) * 2 - 1) / 16
planes = weights.split_planes(Wc).to(dev)
inv_c = float(weights.plane_scale_inv(planes, H, H)[0])
fp16_planes = planes[3 * H * H // 2:]
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())


def smi():
    out = subprocess.run(["rocm-smi", "-P", "-g"], capture_output=True, text=True).stdout
    p = re.search(r"Power \(W\): ([0-9.]+)", out)
    c = re.search(r"\((\d+)Mhz\)", out)
    return (float(p.group(1)) if p else None, int(c.group(1)) if c else None)


def case(name, x, seconds=5.0):
    global variant
    e_t = graph.to_tiled(x.to(dev))
    out = torch.zeros_like(e_t)
    run = lambda: _lib.check(L.difusco_lab_gemm1_nopk(variant, P(e_t), P(fp16_planes), P(out), E, inv_c, 0, 0, st))
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append(smi())
            time.sleep(0.3)
    th = threading.Thread(target=sampler)
    th.start()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n, t_start = 0, time.perf_counter()
    t0.record()
    while time.perf_counter() - t_start < seconds:
        for _ in range(500):
            run()
        n += 500
        torch.cuda.synchronize()
    t1.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    ms = t0.elapsed_time(t1) / n
    body = [s for s in samples[2:-1] if s[0] is not None]      # (the first samples see the ramp)
    rec = {"case": name, "launches": n, "ms_per_launch": ms, "mfma_TF_issued": 2.0 * E * H * H * 3 / (ms * 1e-3) / 1e12,
           "power_W": [s[0] for s in body], "sclk_MHz": [s[1] for s in body]}
    pw = sorted(rec["power_W"])
    ck = sorted(c for c in rec["sclk_MHz"] if c)
    print(f"{name:14s}: {ms:.4f} ms / launch ({rec['mfma_TF_issued']:.0f} TF issued = {rec['mfma_TF_issued'] / 2500:.3f} of 2.5 PF), "
          f"socket power median {pw[len(pw) // 2] if pw else None} W, sclk median {ck[len(ck) // 2] if ck else None} MHz  ({n} launches)", flush=True)
    return rec


recs = [case("idle-check", torch.zeros(E, H), seconds=0.5)]      # (warms the driver path; not reported)
if len(variants) == 1:
    recs = [case("randn", torch.randn(E, H, generator=gen)), case("zeros", torch.zeros(E, H)),
            case("randn", torch.randn(E, H, generator=gen)), case("zeros", torch.zeros(E, H))]
else:      # same data, the variants in turn, three rounds: a same-box comparison in the power-limited steady state
    x = torch.randn(E, H, generator=gen)
    recs = []
    secs = float(sys.argv[3]) if len(sys.argv) > 3 else 5.0
    for rnd in range(3 if len(variants) <= 3 else 2):
        for v in variants:
            variant = v
            recs.append(case(f"randn v{v}", x, seconds=secs))
cap = subprocess.run(["rocm-smi", "-M"], capture_output=True, text=True).stdout
m = re.search(r"Power \(W\): ([0-9.]+)", cap)
print(json.dumps({"variants": variants, "E": E, "power_cap_W": float(m.group(1)) if m else None, "cases": recs}))
