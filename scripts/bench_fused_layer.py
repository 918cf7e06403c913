#!/usr/bin/env python
"""Micro-benchmark of the fused edge-layer kernel (+ node finalize) at the bench workload size, with the
profiling-only ablation masks of difusco_debug_set(0, mask).  GPU only; loads the PROFILING library
(libdifusco_hip_prof.so, `python -m difusco_amd.build --prof`)."""
import ctypes
import sys
import os
import numpy as np
import torch

os.environ["DIFUSCO_PROFILING_LIB"] = "1"

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from difusco_amd import _lib, graph, synthetic, weights  # noqa: E402

dev = torch.device("cuda:0")
H, N1, K, G = 256, 1000, 100, 8
prec = sys.argv[1] if len(sys.argv) > 1 else "fp16x3"
# variants: "<ablation mask>" or "<mask>/<opt>" (opt = difusco_debug_set key 7: bit 0 XCD ranges, bit 1 MFMA chains)
def _var(v):
    a, _, o = v.partition("/")
    return (int(a), int(o or 0))
masks = [_var(m) for m in sys.argv[2].split(",")] if len(sys.argv) > 2 else [(0, 0), (0, 1), (0, 2), (0, 3), (15, 0), (17, 0)]
pts, ei = synthetic.tsp_batch(N1, K, range(G))
g = graph.build_csr(ei, N1 * G, dev, points=pts if os.environ.get("NODE_ORDER", "morton") == "morton" else None)
print("node order:", "morton" if g.node_order is not None else "caller")
E, N = g.n_edges, g.n_nodes
gen = torch.Generator().manual_seed(0)
node4 = torch.randn(N, 4 * H, generator=gen).to(dev)
e0 = graph.to_tiled(torch.randn(E, H, generator=gen).to(dev))
h0 = torch.randn(N, H, generator=gen).to(dev)
Wc = ((torch.rand(H, H, generator=gen) * 2 - 1) / 16)
Wo = ((torch.rand(H, H, generator=gen) * 2 - 1) / 16)
pc, po = weights.split_planes(Wc).to(dev), weights.split_planes(Wo).to(dev)
vec = [torch.randn(H, generator=gen).to(dev) * 0.1 for _ in range(4)] + [(1 + 0.1 * torch.randn(H, generator=gen)).to(dev) for _ in range(3)]
bc, bo, tb, bh, gh, ge, go = vec[0], vec[1], vec[2], vec[3], vec[4], vec[5], vec[6]
scales = weights.fused_scales(Wc, Wo, go.cpu(), bh.cpu()).to(dev)
scratch = torch.zeros(_lib.lib().difusco_fused_scratch_bytes(N, E), dtype=torch.uint8, device=dev)
L = _lib.lib()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())


def run(e, h):
    _lib.check(L.difusco_edge_layer_fused(_lib.PRECISIONS[prec], N, E, P(g.rowptr), P(g.row), P(g.col), P(node4), P(e), P(h),
                                          P(pc), P(po), P(bc), P(gh), P(bh), P(ge), P(bh), P(go), P(bh), P(bo), P(tb), 1,
                                          P(scales), P(scratch), st))


# interleaved rounds (variants alternate inside one process; report median and min - cdna guide rule 24)
ROUNDS, ITERS = 7, 5
L.difusco_debug_set(6, int(os.environ.get('LDS_PAD', '0')))      # profiling: extra LDS bytes -> fewer workgroups per CU
times = {m: [] for m in masks}
e, h = e0.clone(), h0.clone()
for r in range(ROUNDS + 1):
    for mask in masks:
        L.difusco_debug_set(0, mask[0])
        L.difusco_debug_set(7, mask[1])
        run(e, h)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(ITERS):
            run(e, h)
        t1.record()
        torch.cuda.synchronize()
        if r > 0:                       # round 0 = warm-up
            times[mask].append(t0.elapsed_time(t1) / ITERS)
for mask in masks:
    ts = sorted(times[mask])
    ms = ts[len(ts) // 2]
    print(f"{prec} ablate/opt={mask}: median {ms:.3f} ms  min {ts[0]:.3f}  max {ts[-1]:.3f} per layer  (E={E}, "
          f"{2*E*H*4/ms/1e6:.0f} GB/s algorithmic, {4*E*H*H*3/ms/1e9:.0f} TF issued)")
L.difusco_debug_set(0, 0)
L.difusco_debug_set(7, 0)
# the scheduling options must not change a single bit
outs = []
OPTS = (0, 115)
for opt in OPTS:
    L.difusco_debug_set(7, opt)
    e, h = e0.clone(), h0.clone()
    run(e, h)
    torch.cuda.synchronize()
    outs.append((e, h))
L.difusco_debug_set(7, 0)
for idx, opt in enumerate(OPTS):
    if idx:
        print(f"opt {opt} vs opt 0: bit-identical e {torch.equal(outs[0][0], outs[idx][0])}, h {torch.equal(outs[0][1], outs[idx][1])}; "
              f"max |diff| e {(outs[0][0] - outs[idx][0]).abs().max().item():.2e} h {(outs[0][1] - outs[idx][1]).abs().max().item():.2e}")
if len(sys.argv) > 3 and sys.argv[3] == 'nostamp':
    sys.exit(0)
# phase timestamps (s_memtime, 100 MHz-class constant clock or shader clock - reported as raw ticks and as shares)
ntile = ((E + 255) // 256) * 8
dbg = torch.zeros(ntile * 16, dtype=torch.int64, device=dev)
L.difusco_debug_set_ptr.argtypes = [ctypes.c_int, ctypes.c_void_p]
L.difusco_debug_set_ptr(1, ctypes.c_void_p(dbg.data_ptr()))
L.difusco_debug_set(0, int(os.environ.get('STAMP_VARIANT', '16')))
e, h = e0.clone(), h0.clone()
run(e, h)
torch.cuda.synchronize()
L.difusco_debug_set_ptr(1, None)
L.difusco_debug_set(0, 0)
d = dbg.reshape(ntile, 16)[:, :14].cpu().numpy().astype(np.float64)
d = d[d[:, 0] > 0]
names = ["prologue", "G1 stage 0", "G1 stages 1-3", "G1 stages 4-7", "gather+agg", "LN+act", "G2 stage 8", "G2 rest of quarter 0 + out", "G2 quarters 1-3"]
tot = (d[:, 9] - d[:, 0])
print(f"phase stamps over {d.shape[0]} waves: total {tot.mean():.0f} ticks/wave (min {tot.min():.0f} max {tot.max():.0f})")
for i, nme in enumerate(names):
    seg = d[:, i + 1] - d[:, i]
    print(f"  {nme:28s} {seg.mean():9.0f} ticks  {100 * seg.mean() / tot.mean():5.1f} %   (min {seg.min():.0f} max {seg.max():.0f})")
# wall-clock concurrency: how many waves are alive on average
for a_, b_, nme in [(7, 10, "G2 stage 9 (2nd of q0)"), (10, 8, "output q0 (ein wait + 8 stores)"), (8, 11, "G2 stage 10 (1st of q1)"),
                    (11, 12, "G2 stage 11 (2nd of q1)"), (12, 13, "output q1"), (13, 9, "quarters 2-3")]:
    seg = d[:, b_] - d[:, a_]
    print(f"  {nme:34s} {seg.mean():9.0f} ticks   (min {seg.min():.0f} max {seg.max():.0f})")
t0, t1 = d[:, 0].min(), d[:, 9].max()
print(f"  kernel span {t1 - t0:.0f} ticks; sum of wave lifetimes / span = {tot.sum() / (t1 - t0):.0f} waves in flight (2048 = full)")
