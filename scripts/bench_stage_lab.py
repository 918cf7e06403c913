#!/usr/bin/env python
"""Stage-loop laboratory (csrc/stage_lab.hip, PROFILING library): GEMM 1 of the fused edge layer alone - Ce = C e on the
tiled e stream, fp16x3, weight planes through LDS by LDS-DMA - in several workgroup geometries / synchronisation schemes.
Every variant is checked against a float64 product, then timed in interleaved rounds (median / min of 7 x 5 launches).

    python scripts/bench_stage_lab.py [variant,variant,...] [E]

Variant code = EPW/32 * 100000 + WAVES * 10000 + NBUF * 1000 + SYNC * 100 + RING * 10 + PRIO (see stage_lab.hip);
16142020 / 16142021 = the production scheme on the 16x16x32 MFMA shape (two / one workgroup per CU).
Output: one line per variant + a JSON record (stdout) for profiles/."""
import ctypes
import json
import os
import sys

import torch

os.environ["DIFUSCO_PROFILING_LIB"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from difusco_amd import _lib, graph, weights  # noqa: E402

ALL = [142020, 142120, 143120, 143130, 143121, 143220, 144220,
       242020, 243120, 244120, 244130, 244121, 244220,
       182020, 183120, 184120, 184121, 184220]
# a trailing "n" selects the translation unit compiled WITHOUT packed fp32 arithmetic (difusco_lab_gemm1_nopk)
variants = [v for v in sys.argv[1].split(",")] if len(sys.argv) > 1 and sys.argv[1] else [str(v) for v in ALL]
E = int(sys.argv[2]) if len(sys.argv) > 2 else 800_000
H = 256
dev = torch.device("cuda:0")
L = _lib.lib()
for fn in (L.difusco_lab_gemm1, L.difusco_lab_gemm1_nopk):
    fn.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int,
                   ctypes.c_int, ctypes.c_void_p]
gen = torch.Generator().manual_seed(0)
x = torch.randn(E, H, generator=gen)
Wc = (torch.rand(H, H, generator=gen) * 2 - 1) / 16
planes = weights.split_planes(Wc).to(dev)
inv_c = float(weights.plane_scale_inv(planes, H, H)[0])
fp16_planes = planes[3 * H * H // 2:]                    # the fp16 hi | lo planes follow the three bf16 planes
e_t = graph.to_tiled(x.to(dev))
out = torch.zeros_like(e_t)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())


def run(v, store):
    fn = L.difusco_lab_gemm1_nopk if v.endswith("n") else L.difusco_lab_gemm1
    _lib.check(fn(int(v.rstrip("n")), P(e_t), P(fp16_planes), P(out), E, inv_c, store, 0, st))


# correctness: the first / last 4096 edges against float64
ref_rows = torch.cat([torch.arange(0, 4096), torch.arange(E - 4096, E)])
ref = (x[ref_rows].double() @ Wc.double().T)
ok = {}
for v in variants:
    out.zero_()
    run(v, 1)
    torch.cuda.synchronize()
    code = int(v.rstrip("n"))
    vmix = code // 1_000_000 if code < 10_000_000 else 0      # (16xxxxxx: the 16x16x32 MFMA shape, real results)
    if vmix not in (0, 2):      # VMIX 1 / 3: the synthetic epilogue rewrites the accumulators (timing only)
        ok[v] = None
        continue
    got = graph.from_tiled(out, E)[ref_rows.to(dev)].cpu().double()
    err = float((got - ref).abs().max())
    ok[v] = err
    print(f"variant {v}: max |err| vs float64 {err:.2e}", flush=True)
    assert err < 5e-6, (v, err)

ROUNDS, ITERS = 7, 5
times = {v: [] for v in variants}
for r in range(ROUNDS + 1):
    for v in variants:
        run(v, 0)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(ITERS):
            run(v, 0)
        t1.record()
        torch.cuda.synchronize()
        if r > 0:
            times[v].append(t0.elapsed_time(t1) / ITERS)
rec = {"E": E, "what": "GEMM 1 only (Ce = C e, fp16x3, no store), ms per launch", "variants": {}}
n_mfma_cycles = (E / 32) * 384 * 32 / 1024        # matrix-pipe cycles per SIMD
for v in variants:
    ts = sorted(times[v])
    med = ts[len(ts) // 2]
    pipe = n_mfma_cycles / (med * 1e-3 * 2.4e9)
    tf = 2.0 * E * H * H * 3 / (med * 1e-3) / 1e12
    rec["variants"][str(v)] = {"median_ms": med, "min_ms": ts[0], "max_ms": ts[-1], "mfma_TF_issued": tf,
                               "pipe_frac_at_2.4GHz": pipe, "max_err": ok[v]}
    print(f"variant {v}: median {med:.4f} ms  min {ts[0]:.4f}  max {ts[-1]:.4f}   {tf:7.1f} TF issued = {tf / 2500:.3f} of 2.5 PF "
          f"(pipe {100 * pipe:.1f} % at 2.4 GHz)", flush=True)
print(json.dumps(rec))
