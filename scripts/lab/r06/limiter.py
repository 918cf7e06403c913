#!/usr/bin/env python
"""Round 6, VERDICT r5 #2: NAME the limiter of the engine clock with the firmware's own flags instead of inferring it.

Each case runs for `seconds` in a steady loop while (1) the SMU metrics table is snapshotted (scripts/smu_metrics.py: throttler
residency accumulators PPT / socket-thermal / VR-thermal / HBM-thermal / PROCHOT and, per XCD, "engine clock below the host limit
because of power | temperature | anything") and (2) hwmon power / clock are sampled every 5 ms.

    python scripts/lab/r06/limiter.py step        production library: the TSP-1000 x 8 denoise step (default engine),
                                                  the same step with ALL-ZERO weights (same instruction stream, zero operands),
                                                  the TSP-10000 Gaussian step, the MIS step
    python scripts/lab/r06/limiter.py lab         profiling library: GEMM 1 of the fused layer alone, N(0,1) vs zero operands
Output: one table row per case + one JSON line (-> profiles/r06/limiter*.txt).
"""
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

MODE = sys.argv[1] if len(sys.argv) > 1 else "step"
SECONDS = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
if MODE == "lab":
    os.environ["DIFUSCO_PROFILING_LIB"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from smu_metrics import SmuMetrics, SmuSampler      # noqa: E402
import bench      # noqa: E402  (PowerSampler: hwmon files of the device)
from difusco_amd import _lib, graph, weights      # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
pr = torch.cuda.get_device_properties(dev)
bdf = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
smu = SmuMetrics(pci_bdf=bdf)
print(f"# device {pr.name} {bdf}; SMU metrics available={smu.available} {smu.why} version={getattr(smu, 'version', None)}", flush=True)
records = []


def run_case(name, body, unit_per_call, unit, seconds=SECONDS):
    """body(): enqueue one batch of work (returns nothing); loops for `seconds`, fenced every batch."""
    for _ in range(3):
        body()
    torch.cuda.synchronize()
    time.sleep(0.5)
    hw = bench.PowerSampler(dev)
    sm = SmuSampler(smu, period=0.02) if smu.available else None
    hw.start()
    if sm:
        sm.start()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        body()
        torch.cuda.synchronize()
        n += 1
    dt = time.perf_counter() - t0
    if sm:
        sm.stop()
    hw.stop()
    h = hw.summary(dt / max(n, 1), 1)
    s = sm.summary() if sm else {"available": False}
    rec = {"case": name, "calls": n, "seconds": dt, "rate": n * unit_per_call / dt, "unit": unit, "hwmon": h, "smu": s}
    records.append(rec)
    r = s.get("residency", {})
    print(f"{name:34s} {rec['rate']:10.1f} {unit:14s} hwmon {h.get('power_W_median', 0):6.0f} W med {h.get('power_W_max', 0):6.0f} max "
          f"{h.get('sclk_MHz_median') or 0:5.0f} MHz | smu {s.get('socket_power_W_median')} W gfxclk med {s.get('gfxclk_MHz_median')} "
          f"[{s.get('gfxclk_MHz_min')}..{s.get('gfxclk_MHz_max')}] hotspot {s.get('temp_hotspot_C_max')} C hbm {s.get('temp_mem_C_max')} C | "
          f"residency ppt {r.get('ppt', float('nan')):.3f} thm {r.get('socket_thm', float('nan')):.3f} vr {r.get('vr_thm', float('nan')):.3f} "
          f"hbm {r.get('hbm_thm', float('nan')):.3f} prochot {r.get('prochot', float('nan')):.3f} | clk<host-limit: ppt "
          f"{r.get('gfx_below_host_limit_ppt', float('nan')):.3f} thm {r.get('gfx_below_host_limit_thm', float('nan')):.3f} total "
          f"{r.get('gfx_below_host_limit_total', float('nan')):.3f} low-util {r.get('gfx_low_utilization', float('nan')):.3f} | "
          f"E-acc {r.get('avg_power_W_from_energy', float('nan')):.0f} W", flush=True)
    return rec


if MODE == "step":
    from difusco_amd.engine import DenoiseEngine
    from difusco_amd.models import MISModel, TSPModel
    from difusco_amd.schedules import InferenceSchedule
    from difusco_amd.synthetic import er_mis_edge_index, random_state_dict, tsp_batch_gpu
    H, LAYERS = 256, 12
    sched = InferenceSchedule("cosine", T=1000, inference_T=50)

    def margs(diff, knn):
        return dict(diffusion_type=diff, diffusion_schedule="linear", diffusion_steps=1000, inference_diffusion_steps=50,
                    inference_schedule="cosine", sparse_factor=knn, n_layers=LAYERS, hidden_dim=H, inference_trick="ddim")

    def tsp_case(name, nodes, knn, graphs, diff, zero_weights=False, steps_per_call=10):
        params = random_state_dict(H, LAYERS, 1 if diff == "gaussian" else 2, seed=20240926)
        if zero_weights:
            params = {k: torch.zeros_like(v) for k, v in params.items()}
        eng = DenoiseEngine(params, device=dev)
        mdl = TSPModel(margs(diff, knn), engine=eng, seed=1234)
        pts, ei = tsp_batch_gpu(nodes, knn, range(graphs), dev)
        gen = torch.Generator().manual_seed(77)
        xt = torch.randn(ei.shape[1], generator=gen)
        state = {"xt": (xt if diff == "gaussian" else (xt > 0).float()).to(dev), "i": 0}
        mdl.prepare_schedule([sched(i)[0] for i in range(50)])

        def body():
            for _ in range(steps_per_call):
                t1, t2 = sched(state["i"] % 49)
                state["i"] += 1
                if diff == "gaussian":
                    state["xt"] = mdl.gaussian_denoise_step(pts, state["xt"], np.array([t1]), dev, ei, target_t=np.array([t2]))
                else:
                    state["xt"] = mdl.categorical_denoise_step(pts, state["xt"], np.array([t1]), dev, ei, target_t=np.array([t2]))
        run_case(name, body, graphs * steps_per_call, "graph-steps/s")

    tsp_case("step tsp1000x8 (production)", 1000, 100, 8, "categorical")
    tsp_case("step tsp1000x8 ALL-ZERO weights", 1000, 100, 8, "categorical", zero_weights=True)
    tsp_case("step tsp1000x8 (production) again", 1000, 100, 8, "categorical")
    tsp_case("step tsp10000x1 gaussian", 10000, 100, 1, "gaussian", steps_per_call=5)
    # MIS
    params = random_state_dict(H, LAYERS, 2, seed=20240926)
    eng = DenoiseEngine(params, device=dev)
    mdl = MISModel(margs("categorical", -1), engine=eng, seed=1234)
    eis, n_off = [], 0
    for gid in range(16):
        n = int(np.random.default_rng(5000 + gid).integers(700, 801))
        eis.append(er_mis_edge_index(n, 0.15, seed=1000 + gid) + n_off)
        n_off += n
    ei = torch.from_numpy(np.concatenate(eis, 1)).to(dev)
    st = {"xt": (torch.randn(n_off) > 0).float().to(dev), "i": 0}

    def mis_body():
        for _ in range(5):
            t1, t2 = sched(st["i"] % 49)
            st["i"] += 1
            st["xt"] = mdl.categorical_denoise_step(st["xt"], np.array([t1]), dev, ei, target_t=np.array([t2]))
    run_case("step mis x16", mis_body, 16 * 5, "graph-steps/s")
else:
    E, H = 800_000, 256
    L = _lib.lib()
    L.difusco_lab_gemm1_nopk.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_float,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    gen = torch.Generator().manual_seed(0)
    Wc = (torch.rand(H, H, generator=gen) * 2 - 1) / 16
    planes = weights.split_planes(Wc).to(dev)
    inv_c = float(weights.plane_scale_inv(planes, H, H)[0])
    fp16_planes = planes[3 * H * H // 2:]
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())      # noqa: E731

    def lab_case(name, x, variant=142020):
        e_t = graph.to_tiled(x.to(dev))
        out = torch.zeros_like(e_t)

        def body():
            for _ in range(200):
                _lib.check(L.difusco_lab_gemm1_nopk(variant, P(e_t), P(fp16_planes), P(out), E, inv_c, 0, 0, stream))
        rec = run_case(name, body, 200, "launches/s")
        rec["mfma_TF_issued"] = 2.0 * E * H * H * 3 * rec["rate"] / 1e12
        rec["J_per_launch"] = (rec["smu"].get("residency", {}).get("avg_power_W_from_energy") or rec["hwmon"].get("power_W_median", 0.0)) / rec["rate"]
        print(f"    -> {1e3 / rec['rate']:.4f} ms per launch, {rec['J_per_launch']:.4f} J per launch; as fp16x3 work: {rec['mfma_TF_issued']:.0f} TF = "
              f"{rec['mfma_TF_issued'] / 2500:.3f} of 2.5 PF", flush=True)

    lab_case("lab GEMM1 fp16x3 N(0,1) operands", torch.randn(E, H, generator=gen))
    lab_case("lab GEMM1 fp16x3 zero operands", torch.zeros(E, H))
    lab_case("lab GEMM1 fp16x3 N(0,1) again", torch.randn(E, H, generator=gen))
    # hi.hi in fp16 + both correction products as one FP8 MFMA (K = 64) per two slabs - TIMING / POWER ONLY (stage_lab.hip,
    # VMIX 4: the operand bytes are those of the fp16 fragments, the result is not the product)
    xr = torch.randn(E, H, generator=gen)
    lab_case("lab GEMM1 fp16 + fp8 corrections", xr, 4142020)
    lab_case("lab GEMM1 fp16x3 (same data)", xr, 142020)
    lab_case("lab GEMM1 fp16 + fp8 corrections", xr, 4142020)
print(json.dumps({"mode": MODE, "device": pr.name, "bdf": bdf, "cases": records}))
