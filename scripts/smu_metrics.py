"""SMU (power-management firmware) metrics of one MI355X beside a run: WHICH limiter holds the engine clock.

The amdgpu driver exposes the firmware's metrics table (`gpu_metrics`, v1.x on the MI300 family) through libamd_smi
(`amdsmi_get_gpu_metrics_info`).  Besides instantaneous readings (socket power, the eight XCD engine clocks, hotspot / HBM
temperature, the throttle-status word) it holds ACCUMULATED RESIDENCY COUNTERS: `accumulation_counter` ticks once per firmware
sampling period and `ppt_residency_acc`, `socket_thm_residency_acc`, `vr_thm_residency_acc`, `hbm_thm_residency_acc`,
`prochot_residency_acc` tick in the periods the named limiter was ACTIVE; metrics v1.8 adds, per XCD, the periods the engine
clock sat below the host limit BECAUSE OF power (`gfx_below_host_limit_ppt_acc`), BECAUSE OF temperature (`..._thm_acc`), for
any reason (`..._total_acc`) and the periods of low utilisation (`gfx_low_utilization_acc`).  The difference of two snapshots
around a timed region divided by the difference of `accumulation_counter` is the share of that region each limiter was active:
the hardware's own answer to "what bounds the clock", independent of any sampling period here.

Measurement infrastructure (bench.py's `power.throttle`, scripts/lab/r06/limiter.py); nothing in the product path imports it.
"""
import ctypes
import threading
import time

_ACC_FIELDS = ("accumulation_counter", "prochot_residency_acc", "ppt_residency_acc", "socket_thm_residency_acc",
               "vr_thm_residency_acc", "hbm_thm_residency_acc")
_XCP_ACC_FIELDS = ("gfx_busy_acc", "gfx_below_host_limit_acc", "gfx_below_host_limit_ppt_acc", "gfx_below_host_limit_thm_acc",
                   "gfx_low_utilization_acc", "gfx_below_host_limit_total_acc")
_U16, _U32, _U64 = 0xFFFF, 0xFFFFFFFF, 0xFFFFFFFFFFFFFFFF


def _ok(v, nil):
    return None if v == nil else int(v)


_AMDSMI_READY = False      # amdsmi_init() once per process (bench.py builds one SmuMetrics per measured workload)


class SmuMetrics:
    """One device's metrics table through libamd_smi (the `amdsmi` Python package of the ROCm image: ctypes structs only, the
    C call releases the GIL).  `available` is False - and `why` says so - when the library, the device or the call is missing;
    nothing raises."""

    def __init__(self, pci_bdf=None, index=0):
        self.available, self.why, self._h, self._w = False, "", None, None
        try:
            import amdsmi
            from amdsmi import amdsmi_wrapper as w
            self._w = w
            global _AMDSMI_READY
            if not _AMDSMI_READY:
                try:
                    amdsmi.amdsmi_init()
                    _AMDSMI_READY = True
                except Exception as exc:      # noqa: BLE001
                    self.why = f"amdsmi_init: {exc}"
                    return
            handles = amdsmi.amdsmi_get_processor_handles()
            pick = None
            if pci_bdf:
                for h in handles:
                    try:
                        if amdsmi.amdsmi_get_gpu_device_bdf(h).lower() == pci_bdf.lower():
                            pick = h
                            break
                    except Exception:      # noqa: BLE001
                        continue
            if pick is None and len(handles) > index:
                pick = handles[index]
            if pick is None:
                self.why = "no amdsmi processor handle"
                return
            self._h = pick
            self._buf = w.amdsmi_gpu_metrics_t()
            rc = w.amdsmi_get_gpu_metrics_info(self._h, ctypes.byref(self._buf))
            if rc != 0:
                self.why = f"amdsmi_get_gpu_metrics_info rc={rc}"
                return
            self.available = True
            self.version = f"{self._buf.common_header.format_revision}.{self._buf.common_header.content_revision}"
        except Exception as exc:      # noqa: BLE001
            self.why = f"{type(exc).__name__}: {exc}"

    def read(self):
        """One snapshot: instantaneous readings + the accumulators (None where the firmware reports "unsupported")."""
        m = self._w.amdsmi_gpu_metrics_t()
        if self._w.amdsmi_get_gpu_metrics_info(self._h, ctypes.byref(m)) != 0:
            return None
        clks = [c for c in (_ok(v, _U16) for v in m.current_gfxclks) if c]
        out = {"t": time.perf_counter(),
               "socket_power_W": _ok(m.current_socket_power, _U16) or _ok(m.average_socket_power, _U16),
               "gfxclk_MHz": clks, "uclk_MHz": _ok(m.current_uclk, _U16),
               "temp_hotspot_C": _ok(m.temperature_hotspot, _U16), "temp_mem_C": _ok(m.temperature_mem, _U16),
               "temp_vrsoc_C": _ok(m.temperature_vrsoc, _U16),
               "throttle_status": _ok(m.throttle_status, _U32), "indep_throttle_status": _ok(m.indep_throttle_status, _U64),
               "gfx_activity": _ok(m.average_gfx_activity, _U16), "umc_activity": _ok(m.average_umc_activity, _U16),
               "energy_acc": _ok(m.energy_accumulator, _U64), "firmware_timestamp": _ok(m.firmware_timestamp, _U64)}
        for f in _ACC_FIELDS:
            out[f] = _ok(getattr(m, f), _U64)
        nx = 8
        for f in _XCP_ACC_FIELDS:      # partition 0 (SPX mode): one counter per XCD
            out[f] = [_ok(v, _U64) for v in getattr(m.xcp_stats[0], f)][:nx]
        return out


def residency(a, b):
    """Share of the interval between snapshots a and b each limiter was active (accumulator differences over the difference of
    `accumulation_counter`; per-XCD counters: mean over the XCDs that report)."""
    if not a or not b or a.get("accumulation_counter") is None or b.get("accumulation_counter") is None:
        return None
    ticks = b["accumulation_counter"] - a["accumulation_counter"]
    if ticks <= 0:
        return {"ticks": ticks}
    out = {"ticks": ticks, "seconds": b["t"] - a["t"]}
    for f in _ACC_FIELDS[1:]:
        if a.get(f) is not None and b.get(f) is not None:
            out[f.replace("_residency_acc", "")] = (b[f] - a[f]) / ticks
    for f in _XCP_ACC_FIELDS:
        d = [(y - x) for x, y in zip(a.get(f) or [], b.get(f) or []) if x is not None and y is not None]
        if d:
            out[f.replace("_acc", "") + "_per_xcd"] = [v / ticks for v in d]
            out[f.replace("_acc", "")] = sum(d) / len(d) / ticks
    if a.get("energy_acc") is not None and b.get("energy_acc") is not None and b["t"] > a["t"]:
        out["energy_J"] = (b["energy_acc"] - a["energy_acc"]) * 15.259e-6      # 2^-16 J units (gpu_metrics v1.x)
        out["avg_power_W_from_energy"] = out["energy_J"] / (b["t"] - a["t"])
    return out


class SmuSampler:
    """Background sampling of SmuMetrics (period ~20 ms: the call takes ~0.1-1 ms) between start() and stop(); the residency
    shares come from the first and last snapshot, the medians from all of them."""

    def __init__(self, metrics, period=0.02):
        self.m, self.period, self.samples, self._stop, self._thread = metrics, period, [], None, None
        self.windows = []      # [(first, last)] per start/stop pair

    def _run(self, stop):
        while True:
            s = self.m.read()
            if s:
                self.samples.append(s)
            if stop.wait(self.period):
                break
        s = self.m.read()
        if s:
            self.samples.append(s)

    def start(self):
        self._first = len(self.samples)
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, args=(self._stop,), daemon=True)
        self._thread.start()

    def stop(self):
        if self._thread is not None:
            self._stop.set()
            self._thread.join()
            self._thread = None
            if len(self.samples) > self._first + 1:
                self.windows.append((self.samples[self._first], self.samples[-1]))

    def summary(self):
        if not self.m.available:
            return {"available": False, "why": self.m.why}
        if not self.samples:
            return {"available": True, "samples": 0}

        def med(vals):
            vals = sorted(v for v in vals if v is not None)
            return vals[len(vals) // 2] if vals else None
        flat_clk = [c for s in self.samples for c in s["gfxclk_MHz"]]
        out = {"available": True, "metrics_version": self.m.version, "samples": len(self.samples),
               "socket_power_W_median": med(s["socket_power_W"] for s in self.samples),
               "socket_power_W_max": max((s["socket_power_W"] or 0) for s in self.samples),
               "gfxclk_MHz_median": med(flat_clk), "gfxclk_MHz_min": min(flat_clk) if flat_clk else None,
               "gfxclk_MHz_max": max(flat_clk) if flat_clk else None, "uclk_MHz_median": med(s["uclk_MHz"] for s in self.samples),
               "temp_hotspot_C_max": max((s["temp_hotspot_C"] or 0) for s in self.samples),
               "temp_mem_C_max": max((s["temp_mem_C"] or 0) for s in self.samples),
               "throttle_status_nonzero_share": sum(1 for s in self.samples if s["throttle_status"]) / len(self.samples),
               "indep_throttle_status_or": "0x%x" % _or(s["indep_throttle_status"] or 0 for s in self.samples)}
        # residency shares: the accumulators summed over the timed windows
        tot = {}
        for a, b in self.windows:
            r = residency(a, b)
            if not r or r.get("ticks", 0) <= 0:
                continue
            for k, v in r.items():
                if isinstance(v, (int, float)):
                    tot[k] = tot.get(k, 0.0) + (v if k in ("ticks", "seconds", "energy_J") else v * r["ticks"])
        if tot.get("ticks"):
            res = {k: (v if k in ("ticks", "seconds", "energy_J") else v / tot["ticks"]) for k, v in tot.items() if k != "avg_power_W_from_energy"}
            if "energy_J" in res and res.get("seconds"):
                res["avg_power_W_from_energy"] = res["energy_J"] / res["seconds"]
            out["residency"] = res
        return out


def _or(vals):
    acc = 0
    for v in vals:
        acc |= int(v)
    return acc
