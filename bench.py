#!/usr/bin/env python
"""bench.py - denoising-step throughput of the MI355X-native DIFUSCO sampler.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python bench.py --gpus N --steps K --warmup W          # N > 1 without a launcher: bench.py starts its own N ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port P bench.py --gpus N --steps K --warmup W

Both N > 1 forms run the same code: one process per GPU (the reference's deployment, difusco/train.py:106-115; independent
graph instances per rank, pl_tsp_model.py:254-255), RCCL for the one weight broadcast.  A command whose --gpus does not match
the ranks that actually run (WORLD_SIZE from a launcher, the visible devices, the ranks RCCL gathered) exits non-zero: a
`--gpus 8` command never prints an `n_gpus: 1` line.

Metric (BASELINE.json): graph-steps/s = graphs in flight x denoise steps / wall time, on the headline
configuration TSP-1000 k-NN-sparse (K=100), categorical diffusion, H=256, 12 layers, fp32,
8 graphs per GPU (configs[2]: batch 64 sharded over 8 GPUs -> weak scaling).  One "step" = one
reverse-diffusion step (12-layer GNN forward + categorical posterior + Bernoulli draw) over the
rank's whole batch; inputs are resident in HBM when the timed region starts.  Synthetic data:
uniform random points, k-NN graph incl. self, random-init weights of the reference architecture with
per_layer_out re-randomised.  One JSON line on stdout (rank 0).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, LAYERS = 256, 12
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0     # dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16)
PEAK_HBM_GBS = 8000.0              # HBM3E spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def pmc_traffic_bytes(kernel_name, workload, n_edges, variant):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes (FETCH_SIZE doubled per the
    gfx950 half-count correction + WRITE_SIZE; scripts/gpu_profile.sh + scripts/summarize_prof.py).  The tables
    profiles/r03/pmc_traffic.json (this round's kernel), then profiles/r02/pmc_traffic.json, are keyed by
    "<workload>:<edges on rank 0>:<variant>", i.e. by the exact run the counters were collected on; a run with no profile
    of its own reports None (never another workload's bytes)."""
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        path = os.path.join(ROOT, "profiles", rnd, "pmc_traffic.json")
        try:
            table = json.load(open(path))
        except (OSError, ValueError):
            continue
        entry = table.get(f"{workload}:{n_edges}:{variant}")
        if not entry:
            continue
        for key, val in entry.items():
            if key and key in kernel_name:
                out = {"bytes_per_launch": val["fetch_bytes"] + val["write_bytes"], "fetch_bytes": val["fetch_bytes"],
                       "write_bytes": val["write_bytes"],
                       "source": f"profiles/{rnd}/pmc_traffic.json[{workload}:{n_edges}:{variant}] (rocprofv3 --pmc FETCH_SIZE x 1024 x 2 "
                                 f"(gfx950 half-count correction) + WRITE_SIZE x 1024; memory-side (L2 -> fabric) requests: MALL hits "
                                 f"are counted, no counter separates them from HBM)"}
                # per level, when the L2 / EA passes were collected for this run (round 4): what the L2 saw and what left it
                if "l2_read_requests_128B" in val:
                    out["levels"] = {"l2_read_bytes": val["l2_read_requests_128B"] * 128.0, "l2_hits": val.get("l2_hits"),
                                     "l2_misses": val.get("l2_misses"),
                                     "ea_read_bytes": val.get("ea_read_bytes_128B_requests", 0.0) + val.get("ea_read_bytes_64B_requests", 0.0)
                                     + val.get("ea_read_bytes_32B_requests", 0.0),
                                     "ea_write_bytes": val.get("ea_write_bytes_64B_requests"),
                                     "ea_read_latency_cycles": val.get("ea_read_latency_cycles")}
                return out
    return None


def pmc_pipe_busy(workload, n_edges, variant, avg_launch_s, n_simd=1024, clock_hz=2.4e9):
    """Matrix-pipe occupancy of the dominant kernel: SQ_VALU_MFMA_BUSY_CYCLES per launch (rocprofv3 --pmc pass of the same
    run key, profiles/r04/pmc_sq.json; averaged over the launches of a step) / (SIMDs x live average launch time x the
    2.4 GHz the peak is quoted at).  None when that run was never profiled."""
    entry, rnd = None, None
    for rnd in ("r06", "r05", "r04"):      # the newest round that profiled this run key
        try:
            entry = json.load(open(os.path.join(ROOT, "profiles", rnd, "pmc_sq.json"))).get(f"{workload}:{n_edges}:{variant}")
        except (OSError, ValueError):
            entry = None
        if entry:
            break
    if not entry or "SQ_VALU_MFMA_BUSY_CYCLES" not in entry:
        return None
    busy = float(entry["SQ_VALU_MFMA_BUSY_CYCLES"])
    return {"value": busy / (n_simd * avg_launch_s * clock_hz), "mfma_busy_cycles_per_launch": busy,
            "valu_insts_per_launch": entry.get("SQ_INSTS_VALU"),
            "source": f"profiles/{rnd}/pmc_sq.json[{workload}:{n_edges}:{variant}] (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES) / "
                      f"({n_simd} SIMDs x live avg launch x 2.4 GHz)"}


def power_limited_mfma(mfma_tf_issued, precision):
    """The matrix rate the part sustains on THIS kind of data: GEMM 1 of the fused layer alone, on N(0,1) operands, sits at
    the 1,400 W socket cap with the engine clock held at 1.43 GHz (profiles/r04/lab_power.json, lab_power_randn_vs_zeros.txt:
    the same instruction stream on zeroed operands runs 43 % faster at 2.07 GHz).  Reported beside `frac_issued`; `peak`
    stays the nominal dense peak."""
    if precision != "fp16x3":
        return None
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "r04", "lab_power.json")))
        tf = float(rec["randn"]["mfma_TFLOPs_issued"])
    except (OSError, ValueError, KeyError):
        return None
    return {"gemm_only_TFLOPs_issued": tf, "frac_of_nominal_peak": tf / PEAK_BF16_MFMA_TFLOPS,
            "frac_issued_of_power_limited": mfma_tf_issued / tf, "power_cap_W": rec.get("power_cap_W"),
            "power_W": rec["randn"].get("power_W_median"), "sclk_MHz": rec["randn"].get("sclk_MHz_median"),
            "source": "STATIC (another box, round 4; NOT measured in this run - the live figures are under `power`): "
                      "profiles/r04/lab_power.json (scripts/bench_lab_power.py: GEMM 1 alone, fp16x3, N(0,1) operands, rocm-smi beside it)"}


class PowerSampler:
    """Socket power and engine clock of one GPU, sampled by a background thread WHILE the timed loops run (start / stop around each
    repetition).  Source: the amdgpu hwmon files of the device (power1_average | power1_input in microwatts, freq1_input in Hz: a
    file read costs microseconds, so the period can be 5 ms); when they are not readable, `rocm-smi -P -g` (one sample per ~0.2 s).
    Under the socket power cap time is energy: joules per graph-step is what ranks kernel variants across boxes."""

    def __init__(self, device, period=0.005):
        import glob
        import threading
        self._threading = threading
        self.period, self.samples, self._stop, self._thread = period, [], None, None
        self.power_file = self.freq_file = None
        self.source = "rocm-smi -P -g"
        dirs = []
        try:
            pr = torch.cuda.get_device_properties(device)
            addr = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            dirs = sorted(glob.glob(f"/sys/bus/pci/devices/{addr}/hwmon/hwmon*"))
        except Exception:
            dirs = []
        if not dirs:
            cand = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
            cand = [d for d in cand if os.path.exists(os.path.join(d, "power1_average")) or os.path.exists(os.path.join(d, "power1_input"))]
            idx = device.index or 0
            dirs = cand[idx:idx + 1] if len(cand) > idx else []
        for d in dirs:
            for name in ("power1_average", "power1_input"):
                f = os.path.join(d, name)
                try:
                    float(open(f).read())
                    self.power_file = f
                    break
                except (OSError, ValueError):
                    continue
            f = os.path.join(d, "freq1_input")
            try:
                float(open(f).read())
                self.freq_file = f
            except (OSError, ValueError):
                pass
            if self.power_file:
                self.source = f"hwmon {os.path.basename(self.power_file)}" + (" + freq1_input" if self.freq_file else "")
                break

    def _read(self):
        if self.power_file:
            try:
                w = float(open(self.power_file).read()) * 1e-6
                mhz = float(open(self.freq_file).read()) * 1e-6 if self.freq_file else None
                return (w, mhz)
            except (OSError, ValueError):
                return (None, None)
        import re
        import subprocess
        try:
            txt = subprocess.run(["rocm-smi", "-P", "-g"], capture_output=True, text=True, timeout=5).stdout
        except Exception:
            return (None, None)
        pw = re.search(r"Power \(W\): ([0-9.]+)", txt)
        ck = re.search(r"\((\d+)Mhz\)", txt)
        return (float(pw.group(1)) if pw else None, float(ck.group(1)) if ck else None)

    def _run(self, stop):
        while not stop.is_set():
            self.samples.append(self._read())
            stop.wait(self.period)

    def start(self):
        self._stop = self._threading.Event()
        self._thread = self._threading.Thread(target=self._run, args=(self._stop,), daemon=True)
        self._thread.start()

    def stop(self):
        if self._thread is not None:
            self._stop.set()
            self._thread.join()
            self._thread = None

    def summary(self, seconds_per_step, graphs):
        pw = sorted(s[0] for s in self.samples if s[0] is not None)
        ck = sorted(s[1] for s in self.samples if s[1] is not None)
        if not pw:
            return {"samples": 0, "source": self.source}
        med = pw[len(pw) // 2]
        return {"power_W_median": med, "power_W_max": pw[-1], "sclk_MHz_median": ck[len(ck) // 2] if ck else None, "samples": len(pw),
                "J_per_graph_step": med * seconds_per_step / max(graphs, 1), "source": self.source + ", sampled inside the timed loops"}


def smu_sampler(device):
    """The firmware's own account of what holds the engine clock (scripts/smu_metrics.py: gpu_metrics v1.8 through libamd_smi):
    throttler residency shares (PPT / thermal / VR / HBM / PROCHOT) and, per XCD, the share of the timed loops the engine clock
    sat below the host limit because of power or temperature.  None when the metrics table cannot be read on this box."""
    try:
        if os.path.join(ROOT, "scripts") not in sys.path:
            sys.path.insert(0, os.path.join(ROOT, "scripts"))
        from smu_metrics import SmuMetrics, SmuSampler
        pr = torch.cuda.get_device_properties(device)
        m = SmuMetrics(pci_bdf=f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0",
                       index=device.index or 0)
        if not m.available:
            log(f"[bench] SMU metrics not available: {m.why}")
            return None
        return SmuSampler(m, period=0.02)
    except Exception as exc:      # noqa: BLE001
        log(f"[bench] SMU metrics not available: {exc}")
        return None


def oracle_threads():
    """torch threads for the CPU-oracle leg.  Measured on the GPU box's host (2 x EPYC 9575F, 128 physical cores,
    profiles/r02/cpu_threads_scan.txt, one TSP-1000 step): 16 threads 4.8 s, 32 threads 4.2 s, 64 threads 5.7 s,
    128 threads 10.5 s - the E-row GEMMs of one graph do not scale past one CCD group, so 32 (capped by the physical
    core count) is what the leg uses and reports as `cores`."""
    try:
        import psutil
        phys = int(psutil.cpu_count(logical=False) or os.cpu_count() or 1)
    except Exception:
        phys = int(os.cpu_count() or 1)
    return max(1, min(32, phys))


def cpu_baseline(wl, steps, params, gpu_model, device, warm=True):
    """The CPU oracle (port of the reference op sequence, incl. V applied on E gathered rows) on a bounded
    sample: ONE graph of the same workload, 1 warm-up (`warm`; skipped for the minute-long TSP-10000 step) + `steps` timed
    denoise steps, torch threads = oracle_threads()
    (the default, one thread per SMT sibling, oversubscribes the GEMMs 2.5x).  The first timed step is also run on the GPU
    (same graph, same x_t, same injected uniforms, default engine) and the network outputs are compared:
    "parity_linf".  This leg is the only place where bench.py touches oracle/."""
    from oracle import difusco_oracle as O
    from difusco_amd.synthetic import er_mis_edge_index, tsp_instance
    cores = oracle_threads()
    prev_threads = torch.get_num_threads()
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    t_chk, tt_chk = O.inference_schedule("cosine", 1000, 50, 1)
    if wl["task"] == "mis":
        n = 750
        ei = torch.from_numpy(er_mis_edge_index(n, 0.15, seed=1000))
        tab = O.CategoricalTables()
        xt = (torch.randn(n, generator=g) > 0).float()
        u = torch.rand(n, generator=g)
        step = lambda xt, t1, t2, **kw: O.mis_categorical_denoise_step(params, tab, xt, t1, ei, t2, **kw)
        gpu = lambda x: gpu_model.categorical_denoise_step(x.to(device), np.array([t_chk]), device, ei.to(device),
                                                           target_t=np.array([tt_chk]), uniform=u, return_aux=True)
        what = f"1 graph ER-{n} p=0.15 ({ei.shape[1]} directed edges incl. self loops)"
    elif wl.get("dense"):
        V = wl["nodes"]
        pts = torch.rand(1, V, 2, generator=g)
        tab = O.CategoricalTables()
        xt = (torch.randn(1, V, V, generator=g) > 0).float()
        u = torch.rand(1, V, V, generator=g)
        what = f"1 sample dense TSP-{V} ({V * V} edges)"

        def step(xt, t1, t2, **kw):
            r = O.tsp_categorical_denoise_step(params, tab, pts, xt, t1, None, t2, **kw)
            if isinstance(r, tuple):      # logits [B,2,V,V] -> [B,V,V,2] (the layout the HIP path returns)
                return (r[0], r[1].permute(0, 2, 3, 1).contiguous(), r[2])
            return r
        gpu = lambda x: gpu_model.categorical_denoise_step(pts.to(device), x.to(device), np.array([t_chk]), device, None,
                                                           target_t=np.array([tt_chk]), uniform=u.reshape(-1), return_aux=True)
        ei = None
    else:
        pts, ei = tsp_instance(wl["nodes"], wl["knn"], seed=1000)
        pts, ei = torch.from_numpy(pts), torch.from_numpy(ei)
        what = f"1 graph TSP-{wl['nodes']} K={wl['knn']}"
        if wl["diffusion"] == "gaussian":
            tab = O.GaussianTables()
            xt = torch.randn(ei.shape[1], generator=g)
            u = None
            step = lambda xt, t1, t2, **kw: O.tsp_gaussian_denoise_step(params, tab, pts, xt, t1, ei, t2,
                                                                        **{k: v for k, v in kw.items() if k == "return_aux"})
            gpu = lambda x: gpu_model.gaussian_denoise_step(pts.to(device), x.to(device), np.array([t_chk]), device,
                                                            ei.to(device), target_t=np.array([tt_chk]), return_aux=True)
        else:
            tab = O.CategoricalTables()
            xt = (torch.randn(ei.shape[1], generator=g) > 0).float()
            u = torch.rand(ei.shape[1], generator=g)
            step = lambda xt, t1, t2, **kw: O.tsp_categorical_denoise_step(params, tab, pts, xt, t1, ei, t2, **kw)
            gpu = lambda x: gpu_model.categorical_denoise_step(pts.to(device), x.to(device), np.array([t_chk]), device,
                                                               ei.to(device), target_t=np.array([tt_chk]), uniform=u,
                                                               return_aux=True)
    parity = {}
    with torch.no_grad():
        if warm:
            xt = step(xt, 1000, 969, **({} if u is None else {"generator": g}))       # warm-up
        t0 = time.perf_counter()
        for i in range(steps):
            t1, t2 = O.inference_schedule("cosine", 1000, 50, i + 1)
            if i == 0:          # the checked step: teacher-forced on both sides
                res = step(xt, t1, t2, return_aux=True, **({} if u is None else {"uniform": u}))
                got = gpu(xt)
                torch.cuda.synchronize(device)
                parity["parity_linf"] = float((got[1].cpu().reshape(res[1].shape) - res[1]).abs().max())
                if u is not None:
                    pr = res[2].reshape(-1)
                    parity["parity_prob_linf"] = float((got[2].cpu().reshape(-1) - pr).abs().max())
                    safe = (u.reshape(-1) - pr).abs() > 1e-5      # ties excluded within 1e-5
                    parity["parity_bits_equal"] = bool(torch.equal(got[0].cpu().reshape(-1)[safe], res[0].reshape(-1)[safe]))
                else:
                    parity["parity_xt_linf"] = float((got[0].cpu() - res[0]).abs().max())
                parity["parity_what"] = (f"network output of step (t={t1} -> {t2}) of that graph, HIP path (default engine) vs "
                                         f"CPU oracle, same x_t and uniforms; tolerance 1e-4")
                xt = res[0]
            else:
                xt = step(xt, t1, t2, **({} if u is None else {"generator": g}))
        dt = time.perf_counter() - t0
    torch.set_num_threads(prev_threads)
    out = {"value": steps / dt, "unit": "graph-steps/s", "cores": cores, "kind": "port",
           "sample_short": f"{what}, {steps} oracle step(s), {dt:.1f} s",
           "sample": f"{what} H={H} L={LAYERS} fp32 {wl['diffusion']}, {'1 warm-up + ' if warm else 'no warm-up, '}{steps} timed step(s) "
                     f"of the CPU oracle ({dt:.1f} s, incl. one GPU step for the parity check), "
                     f"torch.set_num_threads({cores}) (fastest of 16/32/64/128 on this host class, profiles/r02/cpu_threads_scan.txt)"}
    out.update(parity)
    return out


def _r(v, nd=4):
    """Round floats for the compact line (significant digits, not decimals)."""
    if isinstance(v, float):
        return float(f"{v:.{nd + 2}g}")
    if isinstance(v, (list, tuple)):
        return [_r(x, nd) for x in v]
    return v


def compact_record(out, full_path=None):
    """The stdout line: the bench contract's fields + roofline + cpu_baseline + per-workload one-liners, no prose.  Everything
    else (kernels table, per-level traffic, sample descriptions, exact-fp32 sub-record ...) is in the full record."""
    keep = {k: (out[k] if k in ("value", "ms_per_step") else _r(out[k])) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                    "vs_baseline", "dtype", "data", "dry_run", "ranks_seen", "rank_ms_per_step", "broadcast_ms",
                                    "parity_linf", "host_enqueue_us_per_step", "prepare_ms_per_sampling_run") if k in out}
    cfg = out.get("config", {})
    keep["config"] = {k: cfg[k] for k in ("workload", "name", "graphs_per_gpu", "global_batch", "nodes", "knn", "nodes_rank0", "edges_rank0",
                                          "gn_stats", "binding", "edge_linear_arithmetic", "fused_edge_layer", "aggregation", "gaussian_xt") if k in cfg}
    rep = out.get("repeats")
    if rep:
        keep["repeats"] = {"n": rep["n"], "ms_per_step": _r(rep["ms_per_step"]), "min_ms_per_step": _r(rep["min_ms_per_step"]),
                           "median_ms_per_step": rep["median_ms_per_step"], "max_ms_per_step": _r(rep["max_ms_per_step"])}

    def roof(r):
        if not r:
            return None
        t = r.get("traffic")
        o = {k: _r(r[k]) for k in ("bound", "achieved", "peak", "unit", "frac", "frac_issued", "hbm_GBs_algorithmic", "hbm_frac_of_8TBs",
                                   "fabric_frac_of_8TBs", "avg_launch_ms", "launches", "other_ms_per_step") if k in r}
        o["traffic"] = None if not t else {"bytes_per_launch": _r(t["bytes_per_launch"]), "fetch_bytes": _r(t["fetch_bytes"]),
                                           "write_bytes": _r(t["write_bytes"])}
        o["kernel"] = r.get("kernel", "").split(" (")[0]
        return o
    if "roofline" in out:
        keep["roofline"] = roof(out["roofline"])
    cb = out.get("cpu_baseline")
    if cb:
        keep["cpu_baseline"] = {k: _r(cb[k]) for k in ("value", "unit", "cores", "kind", "parity_linf", "parity_prob_linf",
                                                       "parity_bits_equal", "parity_xt_linf") if k in cb}
        keep["cpu_baseline"]["sample"] = cb.get("sample_short", "")
    if "power" in out:
        pw = dict(out["power"])
        thr = pw.pop("throttle", None)
        keep["power"] = {k: _r(v) for k, v in pw.items() if k != "source"}
        if thr and thr.get("residency"):      # shares of the timed loops (firmware accumulators): which limiter held the clock
            rs = thr["residency"]
            keep["power"]["throttle"] = {k: _r(rs.get(k), 3) for k in ("ppt", "socket_thm", "vr_thm", "hbm_thm", "prochot",
                                                                        "gfx_below_host_limit_ppt", "gfx_below_host_limit_thm",
                                                                        "gfx_below_host_limit_total") if rs.get(k) is not None}
            keep["power"]["throttle"].update({"gfxclk_MHz": thr.get("gfxclk_MHz_median"), "hotspot_C": thr.get("temp_hotspot_C_max")})
    if "exact_fp32" in out:
        keep["exact_fp32"] = {"value": _r(out["exact_fp32"]["value"]), "ms_per_step": _r(out["exact_fp32"]["ms_per_step"])}
    if "workloads" in out:
        keep["workloads"] = {}
        for name, w in out["workloads"].items():
            r, c = w.get("roofline") or {}, w.get("cpu_baseline") or {}
            keep["workloads"][name] = {"value": _r(w["value"]), "ms_per_step": _r(w["ms_per_step"]), "graphs": w["config"]["graphs_per_gpu"],
                                       "rep_ms": _r(w["repeats"]["ms_per_step"]), "frac": _r(r.get("frac")), "frac_issued": _r(r.get("frac_issued")),
                                       "hbm_frac": _r(r.get("hbm_frac_of_8TBs")), "avg_launch_ms": _r(r.get("avg_launch_ms")),
                                       "other_ms": _r(r.get("other_ms_per_step")), "parity_linf": _r(w.get("parity_linf")),
                                       "cpu": _r(c.get("value")), "cpu_cores": c.get("cores")}
            if "gaussian_xt" in w["config"]:      # (Gaussian workloads: which x_t the steps received, --gaussian-xt)
                keep["workloads"][name]["xt"] = "N(0,1) per step" if w["config"]["gaussian_xt"].startswith("N(0,1)") else "free running"
    keep["full_record"] = os.path.basename(full_path) if full_path else "stderr"
    return keep


# BASELINE.json configs[1..4].  tsp1000 is the configuration the metric is quoted on (the default; it fits one GPU
# at 8 graphs per GPU); the others are the remaining single-GPU shards of the reference's configurations.
WORKLOADS = {
    # `global_batch` = the literal batch of the BASELINE config (--scaling strong splits IT over the N ranks); `graphs_per_gpu` =
    # its per-GPU share on the 8-GPU node the config names (--scaling weak, the default: fixed per-GPU work)
    "tsp1000": dict(task="tsp", diffusion="categorical", nodes=1000, knn=100, graphs_per_gpu=8, global_batch=64),
    "tsp500": dict(task="tsp", diffusion="categorical", nodes=500, knn=50, graphs_per_gpu=16, global_batch=16),
    "tsp10000": dict(task="tsp", diffusion="gaussian", nodes=10000, knn=100, graphs_per_gpu=1, global_batch=8),
    "mis": dict(task="mis", diffusion="categorical", nodes=None, knn=None, graphs_per_gpu=16, global_batch=128),
    # BASELINE configs[0]'s model (dense TSP-50, no k-NN sparsification) with 16 parallel samples in one call: per-sample GroupNorm
    # statistics (gnn_encoder.py:380), complete-graph CSR, fused layers since round 4
    "tsp50dense": dict(task="tsp", diffusion="categorical", nodes=50, knn=None, graphs_per_gpu=16, global_batch=16, dense=True),
}


def self_launch(n):
    """`python bench.py --gpus N` (N > 1) without a launcher: re-run this command under `python -m torch.distributed.run
    --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>` - one process per GPU, LOCAL_RANK = device - and
    return its exit code.  stdout is inherited: rank 0's compact line is the only thing on it."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"[bench] --gpus {n} without a launcher: starting {n} ranks: {' '.join(cmd)}")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="tsp1000", choices=sorted(WORKLOADS),
                    help="tsp1000 = the configuration BASELINE.json's metric is quoted on (default); tsp500, "
                         "tsp10000 (Gaussian diffusion, 1 graph per GPU) and mis (ER-[700,800], p=0.15) are the "
                         "per-GPU shards of the other configurations")
    ap.add_argument("--nodes", type=int, default=None, help="override the workload's node count (TSP)")
    ap.add_argument("--knn", type=int, default=None, help="override the workload's neighbour count (TSP)")
    ap.add_argument("--graphs-per-gpu", type=int, default=None, help="override the workload's graphs per GPU")
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed CPU-oracle steps (0 = skip cpu_baseline)")
    ap.add_argument("--no-profile", action="store_true", help="skip the in-library HIP-event brackets")
    ap.add_argument("--profile-all", action="store_true", help="bracket every kernel launch (full per-kernel table in the JSON; "
                    "costs ~2.7 %% of a step) instead of the dominant kernel only")
    ap.add_argument("--no-fusion", action="store_true", help="unfused kernel sequence (A/B against the fused layer kernel)")
    ap.add_argument("--no-l0-fold", action="store_true", help="A/B: write the first layer's edge input to HBM as a separate "
                    "pass instead of reading it from the 2-row table inside the fused kernel")
    ap.add_argument("--no-gn-fold", action="store_true", help="A/B: head GroupNorm statistics by a separate pass over e "
                    "instead of per-tile partial sums emitted by the last fused layer")
    ap.add_argument("--gn-stats", default="per_shard_call", choices=["per_shard_call", "global"],
                    help="head GroupNorm statistics: over each rank's own call (default, no collective in the loop) or "
                         "over the whole sharded batch (one all-reduce of 65 doubles per step)")
    ap.add_argument("--fused-opt", type=int, default=None, help="A/B: scheduling options of the fused kernel "
                    "(difusco_debug_set key 7: 0 = all off, default = production set)")
    ap.add_argument("--node-linear-depth", type=int, default=None, choices=[1, 4],
                    help="A/B: k steps of global-load lookahead in the node-row linear (difusco_debug_set key 8)")
    ap.add_argument("--debug-set", action="append", default=[], metavar="KEY=VALUE",
                    help="profiling library only: difusco_debug_set(KEY, VALUE) before the run (repeatable)")
    ap.add_argument("--streams", type=int, default=1, help="EXPERIMENT: split the rank's graphs into this many groups, each "
                    "stepped by its own engine on its own HIP stream (TSP workloads), so that one group's small node kernels "
                    "and launch tails overlap another group's edge kernels")
    ap.add_argument("--prof-lib", action="store_true", help="load libdifusco_hip_prof.so (profiling build) instead of the "
                    "production library")
    ap.add_argument("--no-node-reorder", action="store_true", help="A/B: keep the caller's node numbering (no Morton order)")
    ap.add_argument("--no-exact-fp32", action="store_true", help="skip the exact-fp32 (v_mfma_f32_32x32x2_f32) sub-record")
    ap.add_argument("--precision", default="fp16x3", choices=["fp32", "bf16x3", "bf16x6", "fp16x3"],
                    help="arithmetic of the E-row linears: exact fp32 MFMA, or fp32 split into 2/3 bf16 planes "
                         "(bf16x6 keeps all 24 significand bits: fp32-class accuracy)")
    ap.add_argument("--no-workloads", action="store_true", help="skip the `workloads` sub-record (the other north_star sizes: "
                    "tsp500, tsp10000, mis - a few timed steps each with their own roofline, cpu_baseline and parity_linf)")
    ap.add_argument("--repeats", type=int, default=3, help="repetitions of the timed K-step loop (each between its own fences); the "
                    "headline is the median repetition, all of them are listed under `repeats`")
    ap.add_argument("--backend", default=None, choices=["ctypes", "torch"], help="host binding of the C ABI: the PyTorch custom ops "
                    "torch.ops.difusco.* (csrc/torch_ops.cpp; the default) or ctypes (the default with the profiling library)")
    ap.add_argument("--aggregation", default="sum", choices=["sum", "mean", "max"], help="A/B: the reference's --aggregation "
                    "(train.py:52); every published run - and the metric - uses sum")
    ap.add_argument("--no-prepare", action="store_true", help="A/B: recompute the step-invariant part of a TSP step (node "
                    "embedding, layer-0 node linear, time-bias rows) in every step instead of once per (graph, schedule)")
    ap.add_argument("--no-power", action="store_true", help="do not sample socket power / engine clock during the timed loops")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): every rank steps `graphs_per_gpu` graphs, the global batch grows with N; strong: the "
                         "BASELINE config's literal global batch (64 x TSP-1000, --global-batch to override) is split over the N ranks")
    ap.add_argument("--global-batch", type=int, default=None, help="--scaling strong: override the workload's global batch")
    ap.add_argument("--gaussian-xt", default="renoised", choices=["renoised", "free"],
                    help="Gaussian workloads: the x_t a step receives.  renoised (default): a fresh N(0,1) draw per step from a pool generated "
                         "before the timed region - the marginal of x_t under the forward process, which a TRAINED denoiser keeps the reverse "
                         "chain on (|x_t| < ~5).  free: feed every step its predecessor's output; with the random-init weights of this bench "
                         "the predicted noise is uncorrelated with x_t, so the DDIM update multiplies x_t by sqrt(abar'/abar) per step and "
                         "|x_t| runs to ~100 within a schedule - off the generated-input table (|x| < 8), onto the contraction path")
    ap.add_argument("--sub-steps", type=int, default=10, help="timed steps of each `workloads` entry (4 warm-up steps; three repetitions)")
    args = ap.parse_args()
    if args.aggregation != "sum":      # an A/B of the kernels only: the oracle legs and the sub-records are written for the metric (sum)
        args.cpu_steps, args.no_exact_fp32, args.no_workloads = 0, True, True

    # Test hook (tests/test_dist_cpu.py): BENCH_PLUMBING_DRY_RUN=1 walks the multi-rank plumbing of this file on CPU tensors
    # over gloo - the self-launch, rendezvous, rank-0 packing + blob broadcast, graph sharding, the optional statistics
    # all-reduce, the barriers and the max-over-ranks timing - with the denoise step replaced by a stand-in that runs NO
    # kernel.  Its JSON line is marked "dry_run" and is not a measurement.
    dry = os.environ.get("BENCH_PLUMBING_DRY_RUN") == "1"
    single_device = os.environ.get("BENCH_SINGLE_DEVICE") == "1"      # test hook: several ranks on one GPU (with BENCH_BACKEND=gloo)
    if not dry and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback path)")
    if args.gpus < 1:
        raise SystemExit(f"bench.py: --gpus {args.gpus} is not a rank count")
    if not dry and not single_device and torch.cuda.device_count() < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but only {torch.cuda.device_count()} device(s) are visible: refusing to "
                         f"run fewer ranks than asked for")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: start the N ranks ourselves (the same `torch.distributed.run` command the driver uses) and hand back
        # its exit code; rank 0 of that job prints the one line
        raise SystemExit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: the launcher started WORLD_SIZE={world} ranks but the command says --gpus {args.gpus}: refusing "
                         f"to print a line whose n_gpus differs from the command")
    if single_device:
        local_rank = 0
    if dry:
        device = torch.device("cpu")
        os.environ["BENCH_BACKEND"] = "gloo"
        args.no_profile, args.no_exact_fp32, args.cpu_steps = True, True, 0
    else:
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)

    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("BENCH_BACKEND", "nccl")            # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)

    if args.fused_opt is not None or args.node_linear_depth is not None or args.prof_lib or args.debug_set:
        os.environ["DIFUSCO_PROFILING_LIB"] = "1"      # A/B kernel variants exist in libdifusco_hip_prof.so only
    from difusco_amd import _lib
    step_flags = (_lib.FLAG_NO_L0_FOLD if args.no_l0_fold else 0) | (_lib.FLAG_NO_TAIL_FOLD if args.no_gn_fold else 0)
    if args.fused_opt is not None:
        _lib.check(_lib.lib().difusco_debug_set(7, args.fused_opt))
    if args.node_linear_depth is not None:
        _lib.check(_lib.lib().difusco_debug_set(8, args.node_linear_depth))
    for kv in args.debug_set:
        k_, v_ = kv.split("=")
        _lib.check(_lib.lib().difusco_debug_set(int(k_), int(v_)))
    out = measure(args, args.workload, args.steps, args.warmup, args.cpu_steps, not args.no_exact_fp32, rank, world, device, dry,
                  step_flags, dist, overrides=True)
    if rank == 0:
        # the other sizes north_star names, on the same GPU in the same run: a few timed steps each, their own roofline
        # (live HIP events), one oracle step on one graph with the parity of that step.  Headline fields stay TSP-1000.
        if world == 1 and not dry and not args.no_workloads and args.workload == "tsp1000" and args.streams == 1:
            out["workloads"] = {}
            for name in ("tsp500", "tsp10000", "mis", "tsp50dense"):
                t0 = time.perf_counter()
                sub = measure(args, name, args.sub_steps, 4, 1, False, rank, world, device, dry, step_flags, dist, overrides=False)
                keep = {k: sub[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "config", "repeats") if k in sub}
                for k in ("roofline", "cpu_baseline", "parity_linf"):
                    if k in sub:
                        keep[k] = sub[k]
                keep["wall_s"] = time.perf_counter() - t0
                out["workloads"][name] = keep
        # The FULL record (every sub-record, the prose fields, per-level traffic) goes to a file and to stderr; the LAST - and
        # only - stdout line is a compact record (< 4 KB) with the contract's fields, so that a log tail always holds all of it.
        full_path = os.environ.get("BENCH_FULL_JSON", os.path.join(ROOT, "bench_full.json"))      # (git-ignored, never tracked)
        try:
            with open(full_path, "w") as fh:
                json.dump(out, fh)
        except OSError as exc:
            log(f"[bench] could not write {full_path}: {exc}")
            full_path = None
        log("[bench] full record: " + json.dumps(out))
        rec = compact_record(out, full_path)
        line = json.dumps(rec, separators=(",", ":"))
        for drop in ("host_enqueue_us_per_step", "prepare_ms_per_sampling_run", "exact_fp32", "workloads", "power", "repeats",
                     "rank_ms_per_step"):
            if len(line) < 4096:      # (never lose the line to its own size: shed the optional fields first - they stay in the full record)
                break
            log(f"[bench] compact line is {len(line)} bytes: dropping `{drop}` from it")
            rec.pop(drop, None)
            line = json.dumps(rec, separators=(",", ":"))
        if len(line) >= 4096:      # still too long (a very long workload description): the contract's fields only
            log(f"[bench] ERROR: compact line is still {len(line)} bytes after shedding every optional field; printing the contract's fields only")
            rec = {k: rec[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                       "scaling", "vs_baseline", "dtype", "data", "roofline", "cpu_baseline", "full_record") if k in rec}
            rec["config"] = {"workload": str(out.get("config", {}).get("workload", ""))[:200]}
            line = json.dumps(rec, separators=(",", ":"))
        print(line, flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def measure(args, workload, steps, warmup, cpu_steps, exact_fp32, rank, world, device, dry, step_flags, dist, overrides):
    """One workload on this rank's GPU: `warmup` untimed + `steps` timed denoise steps between fences; returns the JSON
    record (rank 0; other ranks return None after taking part in the fences / collectives)."""
    from difusco_amd import _lib
    wl = dict(WORKLOADS[workload])
    if overrides:
        for key in ("nodes", "knn", "graphs_per_gpu"):
            if getattr(args, key) is not None:
                wl[key] = getattr(args, key)
    nodes, knn, graphs_per_gpu = wl["nodes"], wl["knn"], wl["graphs_per_gpu"]
    gaussian, mis, dense = wl["diffusion"] == "gaussian", wl["task"] == "mis", bool(wl.get("dense"))
    from difusco_amd.dist import engine_from_broadcast, gn_allreduce, shard_range
    from difusco_amd.engine import DenoiseEngine
    from difusco_amd.models import MISModel, TSPModel
    from difusco_amd.schedules import InferenceSchedule
    from difusco_amd.synthetic import er_mis_edge_index, random_state_dict, tsp_batch_gpu

    broadcast_ms = None
    # frozen weights: rank 0 creates, RCCL broadcast of the packed blob over xGMI
    params = random_state_dict(H, LAYERS, 1 if gaussian else 2, seed=20240926) if rank == 0 or world == 1 else None
    if dry:
        from difusco_amd.dist import broadcast_weights
        if world > 1:
            (h_, l_, c_), blob = broadcast_weights(params, device, src=0)
            assert (h_, l_, c_) == (H, LAYERS, 1 if gaussian else 2) and blob.numel() > 0 and bool(torch.isfinite(blob).all())
        engine = None
    elif world > 1:
        torch.cuda.synchronize(device)
        dist.barrier()
        tb0 = time.perf_counter()
        engine = engine_from_broadcast(params, device, src=0, precision=args.precision, fused=not args.no_fusion,
                                       flags=step_flags, backend=args.backend, aggregation=args.aggregation)
        torch.cuda.synchronize(device)
        broadcast_ms = 1e3 * (time.perf_counter() - tb0)      # rank-0 packing + the RCCL broadcast of the blob
    else:
        engine = DenoiseEngine(params, device=device, precision=args.precision, fused=not args.no_fusion, flags=step_flags,
                               backend=args.backend, aggregation=args.aggregation)
    margs = dict(diffusion_type=wl["diffusion"], diffusion_schedule="linear", diffusion_steps=1000,
                 inference_diffusion_steps=50, inference_schedule="cosine", sparse_factor=knn if not (mis or dense) else -1,
                 n_layers=LAYERS, hidden_dim=H, inference_trick="ddim", aggregation=args.aggregation)
    gn_reduce = gn_allreduce() if (args.gn_stats == "global" and world > 1) else None
    if dry:
        class _PlumbingModel:      # no kernel: hands x_t back, and exercises the statistics all-reduce when asked for
            def _step(self, xt):
                if gn_reduce is not None:
                    sums = torch.full((65,), float(rank + 1), dtype=torch.float64)
                    gn_reduce(sums)
                    assert float(sums[0]) == world * (world + 1) / 2, "all-reduce of the GroupNorm sums did not add up"
                return xt

            def categorical_denoise_step(self, *a, **k):
                return self._step(a[1] if not mis else a[0])
            gaussian_denoise_step = categorical_denoise_step
        model = _PlumbingModel()
    else:
        model = (MISModel if mis else TSPModel)(margs, engine=engine, seed=1234 + rank, gn_reduce=gn_reduce,
                                                reorder_nodes=not args.no_node_reorder, prepare=not args.no_prepare)

    # this rank's shard of the global batch.  weak: graphs_per_gpu fixed, the global batch grows with N; strong (headline run
    # only): the config's literal global batch split over the ranks (shard_range: sizes differ by at most one graph)
    strong = args.scaling == "strong" and overrides
    if strong:
        G_total = int(args.global_batch or wl["global_batch"])
        if G_total < world:
            raise SystemExit(f"bench.py: --scaling strong: global batch {G_total} < {world} ranks")
    else:
        G_total = graphs_per_gpu * world
    lo, hi = shard_range(G_total, rank, world)
    if strong:
        graphs_per_gpu = hi - lo      # (rank 0's share: what `config.graphs_per_gpu` reports)
    gen = torch.Generator().manual_seed(77 + rank)
    if mis:
        # Erdos-Renyi G(n, 0.15), n ~ U{700..800} per graph (data/README.md:60-68), + reversed copies + self loops
        eis, n_off = [], 0
        for gid in range(lo, hi):
            n = int(np.random.default_rng(5000 + gid).integers(700, 801))
            eis.append(er_mis_edge_index(n, 0.15, seed=1000 + gid) + n_off)
            n_off += n
        points, edge_index = None, torch.from_numpy(np.concatenate(eis, 1)).to(device)
        N_local = n_off
        xt = (torch.randn(N_local, generator=gen) > 0).float().to(device)
    elif dense:
        points = torch.rand(hi - lo, nodes, 2, generator=gen).to(device)
        edge_index = None
        N_local = (hi - lo) * nodes
        xt = (torch.randn(hi - lo, nodes, nodes, generator=gen) > 0).float().to(device)
    else:
        if dry:
            from difusco_amd.synthetic import tsp_batch
            points, edge_index = tsp_batch(nodes, knn, range(lo, hi), device)
        else:
            points, edge_index = tsp_batch_gpu(nodes, knn, range(lo, hi), device)   # k-NN graphs built on the GPU
        N_local = points.shape[0]
        xt = torch.randn(edge_index.shape[1], generator=gen)
        xt = (xt if gaussian else (xt > 0).float()).to(device)
    G_local = hi - lo
    E_local = edge_index.shape[1] if edge_index is not None else (hi - lo) * nodes * nodes
    sched = InferenceSchedule("cosine", T=1000, inference_T=50)

    xt_pool = None
    if gaussian and args.gaussian_xt == "renoised" and not dry:      # (resident in HBM before the timed region starts)
        xt_pool = [torch.randn(E_local, generator=gen).to(device) for _ in range(4)]

    def one_step(i, xt, mdl=None):
        mdl = model if mdl is None else mdl
        t1, t2 = sched(i % 49)                                  # never the final (t2 = 0) step: keeps xt binary
        t1, t2 = np.array([t1]), np.array([t2])
        if mis:
            return mdl.categorical_denoise_step(xt, t1, device, edge_index, target_t=t2)
        if gaussian:
            if xt_pool is not None:      # x_t ~ N(0,1) per step (--gaussian-xt); the step's output is computed and dropped
                xt = xt_pool[i % len(xt_pool)]
            return mdl.gaussian_denoise_step(points, xt, t1, device, edge_index, target_t=t2)
        return mdl.categorical_denoise_step(points, xt, t1, device, edge_index, target_t=t2)

    groups = None
    if args.streams > 1 and not dry and not mis:
        # experiment: S independent sub-batches on S streams.  Each group is its own call (its own head statistics).
        groups = []
        per = (hi - lo) // args.streams
        assert per * args.streams == hi - lo, "--streams must divide the graphs per GPU"
        for k in range(args.streams):
            eng_k = DenoiseEngine(params, device=device, blob=engine.blob, precision=args.precision, fused=not args.no_fusion,
                                  flags=step_flags, aggregation=args.aggregation, backend=args.backend)
            m_k = TSPModel(margs, engine=eng_k, seed=1234 + rank + 100 * k, reorder_nodes=not args.no_node_reorder,
                           prepare=not args.no_prepare)
            p_k, e_k = tsp_batch_gpu(nodes, knn, range(lo + k * per, lo + (k + 1) * per), device)
            x_k = torch.randn(e_k.shape[1], generator=gen)
            x_k = (x_k if gaussian else (x_k > 0).float()).to(device)
            groups.append(dict(model=m_k, points=p_k, ei=e_k, xt=x_k, stream=torch.cuda.Stream(device=device)))
        base_one_step = one_step

        def one_step(i, xt, mdl=None):      # noqa: F811
            if mdl is not None:
                return base_one_step(i, xt, mdl)
            t1, t2 = sched(i % 49)
            t1, t2 = np.array([t1]), np.array([t2])
            for gk in groups:      # no cross-stream waits: the groups are independent chains, fenced by device syncs
                with torch.cuda.stream(gk["stream"]):
                    if gaussian:
                        gk["xt"] = gk["model"].gaussian_denoise_step(gk["points"], gk["xt"], t1, device, gk["ei"], target_t=t2)
                    else:
                        gk["xt"] = gk["model"].categorical_denoise_step(gk["points"], gk["xt"], t1, device, gk["ei"], target_t=t2)
            return xt

    def fence():
        if not dry:
            torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize(device)

    # per-sampling-run preparation, as TSPModel.sample / MISModel.sample do at entry: the time-bias rows of the 50 schedule steps
    # in one launch; the graph's step-invariant state (TSP) is built by the first warm-up step and reused.  Timed separately:
    # it happens once per 50-step run, the timed region below is `steps` steps of that run (pl_tsp_model.py:207-217).
    prepare_ms = None
    if not dry:
        torch.cuda.synchronize(device)
        tp0 = time.perf_counter()
        if groups is None:
            model.prepare_schedule([sched(i)[0] for i in range(50)])
        else:
            for gk in groups:      # (the rows are cached per HIP stream: prepared on the stream that will step)
                with torch.cuda.stream(gk["stream"]):
                    gk["model"].prepare_schedule([sched(i)[0] for i in range(50)])
        torch.cuda.synchronize(device)
        prepare_ms = 1e3 * (time.perf_counter() - tp0)
    for i in range(warmup):
        xt = one_step(i, xt)
    NCAT = 5
    repeats = max(1, args.repeats)
    rep_dt, rep_enq, rep_local, rep_prof = [], [], [], []
    sampler = PowerSampler(device) if (rank == 0 and not dry and not args.no_power) else None
    smu = smu_sampler(device) if sampler is not None else None      # the firmware's throttler residency counters (power.throttle)
    for rep in range(repeats):      # every repetition: exactly `steps` steps between two fences, max over ranks
        if not args.no_profile:     # (re-arms the HIP-event brackets: the events exist after the first call, nothing is created here)
            _lib.check(_lib.lib().difusco_profile_enable(1 if args.profile_all else 2, steps * (4 * LAYERS + 16)))
        fence()
        if sampler is not None:
            sampler.start()
        if smu is not None:
            smu.start()
        t0 = time.perf_counter()
        for i in range(steps):
            xt = one_step(warmup + rep * steps + i, xt)
        t_enq = time.perf_counter() - t0      # host time to enqueue the steps (binding + launch overhead; the GPU runs behind)
        fence()
        dt_r = time.perf_counter() - t0
        if sampler is not None:
            sampler.stop()
        if smu is not None:
            smu.stop()
        if not args.no_profile:     # this repetition's brackets (outside the timed region)
            ms = (ctypes.c_double * NCAT)()
            cnt = (ctypes.c_int64 * NCAT)()
            _lib.check(_lib.lib().difusco_profile_collect(ms, cnt, NCAT))
            rep_prof.append({"ms": list(ms), "launches": list(cnt)})
        rep_local.append(dt_r)
        if world > 1:
            tmax = torch.tensor([dt_r], dtype=torch.float64, device=device)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt_r = float(tmax.item())
        rep_dt.append(dt_r)
        rep_enq.append(t_enq)
    # the reported region is the MEDIAN repetition (one K-step loop between fences); all repetitions are listed
    order = sorted(range(repeats), key=lambda k: rep_dt[k])
    dt = rep_dt[order[(repeats - 1) // 2]]
    # evidence that RCCL saw every rank: world size and the per-rank loop times, all-gathered
    ranks_seen, rank_ms = 1, [1e3 * dt / steps]
    if world > 1:
        mine = torch.tensor([float(rank), 1e3 * rep_local[order[(repeats - 1) // 2]] / steps], dtype=torch.float64, device=device)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        ranks_seen = len({int(v[0].item()) for v in allr})
        rank_ms = [float(v[1].item()) for v in allr]
    if ranks_seen != world:      # (never print a line for fewer ranks than the command names)
        raise SystemExit(f"bench.py: the collective gathered {ranks_seen} distinct ranks, the job has {world}")

    prof = None
    if not args.no_profile:      # the brackets of the REPORTED (median) repetition: every per-launch figure below belongs to it
        _lib.lib().difusco_profile_enable(0, 0)
        prof = rep_prof[order[(repeats - 1) // 2]]

    if rank == 0:
        value = G_total * steps / dt
        out = {
            "metric": ("PLUMBING DRY RUN, NO KERNEL RAN - " if dry else "") + "denoising steps/sec (graphs x steps / s), " + (
                "MIS ER-[700,800] sparse categorical" if mis else f"TSP-{nodes} dense {wl['diffusion']}" if dense else
                f"TSP-{nodes} k-NN sparse {wl['diffusion']}"),
            "value": value, "unit": "graph-steps/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": 1e3 * dt / steps, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": {"fp16x3": "f32 (fp16x3 split MFMA, fp32 accumulate)", "bf16x3": "f32 (bf16x3 split MFMA, fp32 accumulate)",
                      "bf16x6": "f32 (bf16x6 split MFMA, fp32 accumulate)", "fp32": "f32 (fp32 MFMA)"}[args.precision],
            "data": "synthetic", **({"dry_run": True} if dry else {}),
            "repeats": {"n": repeats, "reported": "median repetition (each = `steps` steps between two fences, max over ranks)",
                        "ms_per_step": [1e3 * d / steps for d in rep_dt], "min_ms_per_step": 1e3 * min(rep_dt) / steps,
                        "median_ms_per_step": 1e3 * dt / steps, "max_ms_per_step": 1e3 * max(rep_dt) / steps,
                        "value_min": G_total * steps / max(rep_dt), "value_max": G_total * steps / min(rep_dt)},
            "host_enqueue_us_per_step": 1e6 * sorted(rep_enq)[(repeats - 1) // 2] / steps,
            "prepare_ms_per_sampling_run": prepare_ms,
            "ranks_seen": ranks_seen, "rank_ms_per_step": rank_ms, "broadcast_ms": broadcast_ms,
            "config": {"workload": (f"MIS Erdos-Renyi n~U[700,800] p=0.15 (+reverse edges, +self loops) sparse categorical"
                                    if mis else f"TSP-{nodes} dense (complete graph, one GroupNorm statistic segment per sample) {wl['diffusion']}"
                                    if dense else f"TSP-{nodes} k-NN K={knn} sparse {wl['diffusion']}") +
                                   f", cosine 50-step schedule, {graphs_per_gpu} graphs per GPU "
                                   f"(global batch {G_total}), H={H}, {LAYERS} layers",
                       "name": workload, "graphs_per_gpu": graphs_per_gpu, "global_batch": G_total,
                       "nodes": nodes, "knn": knn, "nodes_rank0": N_local, "edges_rank0": E_local,
                       "gn_stats": args.gn_stats,
                       "rng": "on-device philox", "weights_seed": 20240926, "edge_linear_arithmetic": args.precision,
                       "fused_edge_layer": (not args.no_fusion) and args.precision in ("bf16x3", "fp16x3"),
                       "node_order": "caller" if (args.no_node_reorder or mis) else "morton per graph (graph.py)",
                       "fused_opt": args.fused_opt, "debug_set": args.debug_set or None, "streams": args.streams,
                       "binding": ("dry run" if engine is None else "ctypes -> C ABI" if engine.backend == "ctypes" else
                                   "torch.ops.difusco.* custom ops -> C ABI"),
                       "prepared_state": (not args.no_prepare), "aggregation": args.aggregation,
                       **({"gaussian_xt": ("N(0,1) per step from a pre-generated pool (the forward marginal; what a trained denoiser keeps "
                                           "x_t on)" if xt_pool is not None else "free running (random-init weights: |x_t| drifts to ~100)"),
                           "xt_abs_max_after_run": float(xt.abs().max())} if gaussian and not dry else {})},
        }
        if sampler is not None:      # (N = 1: one socket; N > 1: rank 0's socket, graphs of rank 0)
            out["power"] = sampler.summary(dt / steps, G_local)
            if smu is not None:
                out["power"]["throttle"] = smu.summary()
        if prof is not None and prof["launches"][0] > 0:
            n_lin = prof["launches"][0]
            avg_s = prof["ms"][0] / n_lin * 1e-3
            # one E-row linear [E,H] x [H,H]^T.  Algorithmic bytes: read X + write Y (+ read residual for the
            # per_layer_out linear = every second launch) -> 2.5 passes of E*H*4 on average.
            fused = (not args.no_fusion) and args.precision in ("bf16x3", "fp16x3")
            # fused: both E-row GEMMs of a layer in one launch.  The first layer of a table-input step (categorical TSP,
            # MIS) has no GEMM 1 and reads no e; the last layer of a MIS step has no GEMM 2 and writes no e: the
            # per-launch averages below account for that (12 launches per step).
            first_light = fused and not args.no_l0_fold and (mis or not gaussian)      # (dense TSP categorical included)
            last_light = fused and not args.no_gn_fold and mis and LAYERS >= 2
            gemms = (2.0 * LAYERS - first_light - last_light) / LAYERS if fused else 1.0
            passes = (2.0 * LAYERS - first_light - last_light) / LAYERS if fused else 2.5
            flops = 2.0 * gemms * E_local * H * H
            n_prod = {"fp32": 1, "bf16x3": 3, "bf16x6": 6, "fp16x3": 3}[args.precision]
            mfma_peak = PEAK_FP32_MFMA_TFLOPS if args.precision == "fp32" else PEAK_BF16_MFMA_TFLOPS
            mfma_tf = flops * n_prod / avg_s / 1e12          # matrix-core work actually issued
            bytes_alg = passes * E_local * H * 4   # fused: e read once + written once per layer (see above)
            hbm_gbs = bytes_alg / avg_s / 1e9
            kname = (f"edge_layer_fused_kernel<{'FFp16' if args.precision == 'fp16x3' else 'FBf16'}> (whole edge pass of a "
                     f"layer: 2 chained E-row GEMMs, 3 MFMA products each, gate, 2 LayerNorms, neighbour-sum pieces)"
                     if fused else
                     "linear_rows_kernel<256,256,16> (E-row linear, exact fp32 MFMA)" if args.precision == "fp32" else
                     f"linear_rows_split_kernel<256,256,{3 if args.precision == 'bf16x6' else 2},"
                     f"{'Fp16' if args.precision == 'fp16x3' else 'Bf16'}> "
                     f"(E-row linear, fp32 split into 16-bit planes, {n_prod} MFMA products)")
            # Headline fraction = ALGORITHMIC work / time / peak: every fp32 multiply-add of the two E-row GEMMs
            # counted ONCE against the dense 16-bit MFMA peak.  The matrix cores actually execute n_prod 16-bit products per
            # fp32 product (split precision): that is `frac_issued` (= the pipe occupancy the PMC pass reports as `pipe_busy`).
            alg_tf = flops / avg_s / 1e12
            variant = ("fused" if fused else "unfused") + "-" + args.precision
            bound = "mfma" if mfma_tf / mfma_peak >= hbm_gbs / PEAK_HBM_GBS else "hbm"
            traffic = pmc_traffic_bytes(kname, workload, E_local, variant)
            if bound == "mfma":
                out["roofline"] = {"bound": "mfma", "achieved": alg_tf, "peak": mfma_peak, "unit": "TFLOP/s",
                                   "frac": alg_tf / mfma_peak, "traffic": traffic}
            else:
                out["roofline"] = {"bound": "hbm", "achieved": hbm_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                   "frac": hbm_gbs / PEAK_HBM_GBS, "traffic": traffic}
            out["roofline"].update({"kernel": kname, "avg_launch_ms": avg_s * 1e3, "launches": n_lin,
                                    "bound_note": "the kernel issues n_prod x the algorithmic flops on the matrix cores: by ISSUED work "
                                                  "it sits nearer the MFMA roof than the HBM roof, hence bound = mfma; frac counts "
                                                  "each algorithmic flop once",
                                    "algorithmic_flops_per_launch": flops, "mfma_products": n_prod,
                                    "algorithmic_bytes_per_launch": bytes_alg,
                                    "algorithmic_TFLOPs": alg_tf, "mfma_peak_TFLOPs": mfma_peak,
                                    "frac_algorithmic": alg_tf / mfma_peak,
                                    "mfma_TFLOPs_issued": mfma_tf, "frac_issued": mfma_tf / mfma_peak,
                                    "pipe_busy": pmc_pipe_busy(workload, E_local, variant, avg_s),
                                    "power_limited_mfma": power_limited_mfma(mfma_tf, args.precision) if fused else None,
                                    "frac_fp32_mfma_peak": alg_tf / PEAK_FP32_MFMA_TFLOPS,
                                    "hbm_GBs_algorithmic": hbm_gbs, "hbm_frac_of_8TBs": hbm_gbs / PEAK_HBM_GBS,
                                    "other_ms_per_step": 1e3 * dt / steps - avg_s * 1e3 * n_lin / steps})
            if traffic:      # measured memory-side bytes (PMC pass of this run key) over the live launch time: what the fabric moves,
                # memory-side cache hits included (no counter separates them from HBM reads)
                out["roofline"]["fabric_GBs_measured"] = traffic["bytes_per_launch"] / avg_s / 1e9
                out["roofline"]["fabric_frac_of_8TBs"] = traffic["bytes_per_launch"] / avg_s / 1e9 / PEAK_HBM_GBS
            n_g = max(prof["launches"][2], 1)
            g_s = prof["ms"][2] / n_g * 1e-3
            if fused:
                out["kernels"] = {
                    "edge_layer_fused": {"ms_total": prof["ms"][0], "launches": prof["launches"][0],
                                         "hbm_GBs_algorithmic": hbm_gbs, "hbm_frac_of_8TBs": hbm_gbs / PEAK_HBM_GBS,
                                         "mfma_TFLOPs_issued": mfma_tf, "mfma_frac": mfma_tf / mfma_peak},
                    "node_linear": {"ms_total": prof["ms"][1], "launches": prof["launches"][1]},
                    "node_finalize": {"ms_total": prof["ms"][2], "launches": prof["launches"][2]},
                }
            else:
                gate_bytes = 2.0 * E_local * H * 4              # read C e, write act (node tables are L2/MALL traffic)
                out["kernels"] = {
                    "edge_linear": {"ms_total": prof["ms"][0], "launches": prof["launches"][0]},
                    "node_linear": {"ms_total": prof["ms"][1], "launches": prof["launches"][1]},
                    "edge_gate_aggregate": {"ms_total": prof["ms"][2], "launches": prof["launches"][2],
                                            **({"achieved_GBs": gate_bytes / g_s / 1e9, "peak_GBs": PEAK_HBM_GBS,
                                                "frac": gate_bytes / g_s / 1e9 / PEAK_HBM_GBS} if g_s > 0 else {}),      # (--profile-all)
                                            "bound": "hbm", "algorithmic_bytes_per_launch": gate_bytes},
                }
            out["kernels"]["profiled"] = ("every launch (--profile-all)" if args.profile_all else
                                          "dominant kernel only; the other entries are empty (use --profile-all)")
            out["kernels"].update({
                "head": {"ms_total": prof["ms"][3], "launches": prof["launches"][3]},
                "embed_misc": {"ms_total": prof["ms"][4], "launches": prof["launches"][4]},
                "sum_ms_per_step": sum(prof["ms"]) / steps,
            })
        if world == 1 and exact_fp32 and args.precision != "fp32":
            # the same workload with every E-row contraction on v_mfma_f32_32x32x2_f32 (exact fp32, no split planes):
            # the number to read when the split-precision arithmetic of the headline is not accepted
            eng32 = DenoiseEngine(params, device=device, blob=engine.blob, precision="fp32", fused=False, aggregation=args.aggregation)
            m32 = (MISModel if mis else TSPModel)(margs, engine=eng32, seed=1234, reorder_nodes=not args.no_node_reorder)
            x32 = one_step(0, xt, m32)
            fence()
            n32 = max(2, min(4, steps))
            t0 = time.perf_counter()
            for i in range(n32):
                x32 = one_step(1 + i, x32, m32)
            fence()
            d32 = time.perf_counter() - t0
            del eng32, m32, x32
            out["exact_fp32"] = {"value": G_total * n32 / d32, "unit": "graph-steps/s", "ms_per_step": 1e3 * d32 / n32,
                                 "steps": n32, "edge_linear_arithmetic": "fp32 (v_mfma_f32_32x32x2_f32), unfused kernel sequence",
                                 "same_workload": True}
        if world == 1 and cpu_steps > 0:
            out["cpu_baseline"] = cpu_baseline(wl, cpu_steps, params, model, device, warm=overrides or workload != "tsp10000")
            if "parity_linf" in out["cpu_baseline"]:
                out["parity_linf"] = out["cpu_baseline"]["parity_linf"]
        return out
    return None


if __name__ == "__main__":
    main()
