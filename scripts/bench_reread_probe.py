#!/usr/bin/env python
"""Where is a second read of freshly streamed data served?  The fused edge kernel reads a 32 KB tile of e for
GEMM 1 and re-reads it ~35-50 us later as the residual of GEMM 2; ~70-100 MB of other tiles stream through in between.  This probe
streams a buffer of S bytes twice (two launches, float4 loads with the kernel's cache policies) and compares pass 2 with the cold
pass 1, for S below and above the 256 MiB memory-side cache, with an eviction sweep of 2 GiB before every pair.
    python scripts/bench_reread_probe.py"""
import ctypes
import json
import os
import sys

import torch

os.environ["DIFUSCO_PROFILING_LIB"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from difusco_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
L.difusco_lab_reread_pass.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
sink = torch.zeros(4, device=dev)
big = torch.ones(512 << 20, device=dev)          # 2 GiB eviction sweep
rec = {}
for mb in (16, 64, 100, 160, 224, 320, 512, 1024):
    n = (mb << 20) // 4
    buf = torch.ones(n, device=dev)
    for nt1, nt2, label in ((0, 2, "pass 1 default (GEMM 1 slabs), pass 2 non-temporal (residual)"), (2, 2, "both non-temporal"),
                            (0, 0, "both default")):
        t1s, t2s = [], []
        for _ in range(5):
            L.difusco_lab_reread_pass(P(big), big.numel(), 0, P(sink), st)       # evict
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            ev[0].record()
            L.difusco_lab_reread_pass(P(buf), n, nt1, P(sink), st)
            ev[1].record()
            L.difusco_lab_reread_pass(P(buf), n, nt2, P(sink), st)
            ev[2].record()
            torch.cuda.synchronize()
            t1s.append(ev[0].elapsed_time(ev[1]))
            t2s.append(ev[1].elapsed_time(ev[2]))
        t1, t2 = sorted(t1s)[2], sorted(t2s)[2]
        gb = mb * (1 << 20) / 1e9
        rec[f"{mb}MiB:{label}"] = {"pass1_ms": t1, "pass2_ms": t2, "pass1_GBs": gb / t1 * 1e3, "pass2_GBs": gb / t2 * 1e3}
        print(f"{mb:5d} MiB  {label:66s} pass 1 {t1:7.4f} ms = {gb / t1 * 1e3:7.0f} GB/s   pass 2 {t2:7.4f} ms = {gb / t2 * 1e3:7.0f} GB/s", flush=True)
print(json.dumps(rec))
