"""Round 6 probe: what the GPU box lets an ordinary user read about the power-management state (run once, output committed
as profiles/r06/probe_smu.txt).  No kernels; safe."""
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def sh(cmd, t=30):
    try:
        r = subprocess.run(cmd, shell=True, capture_output=True, text=True, timeout=t)
        return (r.stdout + r.stderr).strip()
    except Exception as exc:      # noqa: BLE001
        return f"<{exc}>"


print("== sysfs")
for f in sorted(glob.glob("/sys/class/drm/card*/device/gpu_metrics")):
    try:
        raw = open(f, "rb").read()
        print(f, "bytes", len(raw), "header structure_size", int.from_bytes(raw[0:2], "little"), "format", raw[2], "content", raw[3])
    except OSError as exc:
        print(f, exc)
for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
    print(d)
    for g in sorted(glob.glob(d + "/*")):
        if os.path.isfile(g):
            try:
                print("   ", os.path.basename(g), open(g).read().strip()[:80])
            except OSError as exc:
                print("   ", os.path.basename(g), exc)
for f in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")) + sorted(glob.glob("/sys/class/drm/card*/device/pp_power_profile_mode")) + sorted(glob.glob("/sys/class/drm/card*/device/power_dpm_force_performance_level")):
    try:
        print(f, "|", open(f).read().strip().replace("\n", " ; ")[:300])
    except OSError as exc:
        print(f, exc)

print("== amd-smi")
print(sh("amd-smi version"))
print(sh("amd-smi metric -g 0 --throttle", 60)[:3000])
print(sh("amd-smi metric -g 0 -p -c -t", 60)[:3000])
print(sh("amd-smi static -g 0 -l", 60)[:3000])
print("== rocm-smi")
print(sh("rocm-smi --showmaxpower --showpower --showclocks --showperflevel --showtemp", 60)[:3000])

print("== amdsmi python (scripts/smu_metrics.py)")
from smu_metrics import SmuMetrics, SmuSampler, residency      # noqa: E402

m = SmuMetrics()
print("available", m.available, m.why, getattr(m, "version", None))
if m.available:
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        s = m.read()
    print("read() ms", 1e3 * (time.perf_counter() - t0) / n)
    print(json.dumps(s, indent=1))
    a = m.read()
    time.sleep(1.0)
    b = m.read()
    print("idle 1 s residency", json.dumps(residency(a, b)))
