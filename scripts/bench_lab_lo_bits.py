#!/usr/bin/env python
"""Is the matrix cores' power DATA dependent enough to pay for a truncated low plane?  GEMM 1 of the fused edge layer alone
(stage_lab.hip, PROFILING library, production geometry 142020), ~4 s of back-to-back launches per case, the same instruction stream
over the same number of bytes - only the number of significand bits the LOW fp16 planes carry differs (operand x = hi + lo with lo
rounded to `b` bits, weight lo plane masked to `b` bits):

    b = 11 (N(0,1) as is: the production data)   8   6   4   0 (lo planes all zero)   and x = 0 (round 4's reference point)

Accuracy side: scripts/lab_correction_precision.py.      python scripts/bench_lab_lo_bits.py [E]
"""
import ctypes
import glob
import json
import os
import sys
import threading
import time

import torch

os.environ["DIFUSCO_PROFILING_LIB"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from difusco_amd import _lib, graph, weights  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 800_000
H = 256
dev = torch.device("cuda:0")
L = _lib.lib()
L.difusco_lab_gemm1_nopk.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_float,
                                     ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
gen = torch.Generator().manual_seed(0)
Wc = (torch.rand(H, H, generator=gen) * 2 - 1) / 16
planes = weights.split_planes(Wc)
inv_c = float(weights.plane_scale_inv(planes, H, H)[0])
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
pr = torch.cuda.get_device_properties(dev)
hw = sorted(glob.glob(f"/sys/bus/pci/devices/{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0/hwmon/hwmon*"))
hw = hw[0] if hw else None


def smi():
    try:
        return float(open(os.path.join(hw, "power1_input")).read()) * 1e-6, float(open(os.path.join(hw, "freq1_input")).read()) * 1e-6
    except Exception:
        return None, None


def lo_bits_tensor(x, bits):
    """x (fp32) -> hi + lo with lo = the fp16 remainder rounded to `bits` significand bits (0: no low plane)"""
    hi = x.to(torch.float16).float()
    if bits == 0:
        return hi
    lo = (x - hi).to(torch.float16).float()
    m, e = torch.frexp(lo)
    lo = torch.ldexp(torch.round(m * 2 ** bits) / 2 ** bits, e)
    return hi + lo      # exact in fp32: hi has 11 bits, lo sits at most 22 bits below its leading bit


def masked_planes(bits):
    p16 = planes.clone().view(torch.int16)
    lo = p16[4 * H * H: 5 * H * H]      # planes: bf16 hi | mid | lo, fp16 hi | lo
    if bits == 0:
        lo.zero_()
    elif bits < 11:
        lo &= ~((1 << (11 - bits)) - 1)      # (truncation: the power question does not need round-to-nearest)
    return p16.view(torch.float32)[3 * H * H // 2:].to(dev)


def case(name, x, bits, seconds=4.0):
    e_t = graph.to_tiled(x.to(dev))
    out = torch.zeros_like(e_t)
    pl = masked_planes(bits)
    run = lambda: _lib.check(L.difusco_lab_gemm1_nopk(142020, P(e_t), P(pl), P(out), E, inv_c, 0, 0, st))
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append(smi())
            time.sleep(0.05)
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
    body = [s for s in samples[10:-2] if s[0] is not None]
    pw = sorted(s[0] for s in body)
    ck = sorted(s[1] for s in body if s[1])
    rec = {"case": name, "lo_bits": bits, "ms_per_launch": ms, "mfma_TF_issued": 2.0 * E * H * H * 3 / (ms * 1e-3) / 1e12,
           "power_W_median": pw[len(pw) // 2] if pw else None, "sclk_MHz_median": ck[len(ck) // 2] if ck else None, "launches": n}
    print(f"{name:22s}: {ms:.4f} ms / launch ({rec['mfma_TF_issued']:.0f} TF issued = {rec['mfma_TF_issued'] / 2500:.3f} of 2.5 PF), "
          f"power {rec['power_W_median']} W, sclk {rec['sclk_MHz_median']} MHz ({n} launches)", flush=True)
    return rec


x = torch.randn(E, H, generator=gen)
case("warm-up", x, 11, seconds=1.0)
recs = []
for rnd in range(2):
    for bits in (11, 8, 6, 4, 0):
        recs.append(case(f"lo planes {bits:2d} bits", lo_bits_tensor(x, bits) if bits < 11 else x, bits))
    recs.append(case("x = 0", torch.zeros(E, H), 11))
print(json.dumps({"E": E, "cases": recs}))
