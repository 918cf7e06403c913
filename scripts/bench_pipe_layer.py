#!/usr/bin/env python
"""A/B of the two fused edge-layer kernels (difusco_debug_set(2, v): 0 = phase-serial, 1 = software pipelined) at the
bench workload size: result difference, then interleaved timing rounds.  GPU only."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from difusco_amd import _lib, graph, synthetic, weights  # noqa: E402

dev = torch.device("cuda:0")
H, K = 256, 100
prec = sys.argv[1] if len(sys.argv) > 1 else "fp16x3"
N1 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
G = int(sys.argv[3]) if len(sys.argv) > 3 else 8
pts, ei = synthetic.tsp_batch(N1, min(K, N1 - 1), range(G))
g = graph.build_csr(ei, N1 * G, dev)
E, N = g.n_edges, g.n_nodes
gen = torch.Generator().manual_seed(0)
node4 = torch.randn(N, 4 * H, generator=gen).to(dev)
e0 = graph.to_tiled(torch.randn(E, H, generator=gen).to(dev))
h0 = torch.randn(N, H, generator=gen).to(dev)
Wc = ((torch.rand(H, H, generator=gen) * 2 - 1) / 16)
Wo = ((torch.rand(H, H, generator=gen) * 2 - 1) / 16)
pc, po = weights.split_planes(Wc).to(dev), weights.split_planes(Wo).to(dev)
vec = [torch.randn(H, generator=gen).to(dev) * 0.1 for _ in range(4)] + \
      [(1 + 0.1 * torch.randn(H, generator=gen)).to(dev) for _ in range(3)]
bc, bo, tb, bh, gh, ge, go = vec
L = _lib.lib()
scratch = torch.zeros(L.difusco_fused_scratch_bytes(N, E), dtype=torch.uint8, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())


def run(e, h):
    _lib.check(L.difusco_edge_layer_fused(_lib.PRECISIONS[prec], N, E, P(g.rowptr), P(g.row), P(g.col), P(node4), P(e),
                                          P(h), P(pc), P(po), P(bc), P(gh), P(bh), P(ge), P(bh), P(go), P(bh), P(bo),
                                          P(tb), 1, P(scratch), st))


outs = []
for v in (0, 1):
    L.difusco_debug_set(2, v)
    e, h = e0.clone(), h0.clone()
    scratch.zero_()
    run(e, h)
    torch.cuda.synchronize()
    outs.append((e, h))
de = (outs[0][0] - outs[1][0]).abs().max().item()
dh = (outs[0][1] - outs[1][1]).abs().max().item()
print(f"E={E} N={N}: serial vs pipelined max |diff| e {de:.3e}  h {dh:.3e}   (|e| max {outs[0][0].abs().max().item():.2f})")

ROUNDS, ITERS = 7, 5
times = {0: [], 1: []}
e, h = e0.clone(), h0.clone()
for r in range(ROUNDS + 1):
    for v in (0, 1):
        L.difusco_debug_set(2, v)
        run(e, h)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(ITERS):
            run(e, h)
        t1.record()
        torch.cuda.synchronize()
        if r > 0:
            times[v].append(t0.elapsed_time(t1) / ITERS)
for v in (0, 1):
    ts = sorted(times[v])
    ms = ts[len(ts) // 2]
    print(f"{prec} variant {v}: median {ms:.3f} ms  min {ts[0]:.3f}  max {ts[-1]:.3f} per layer "
          f"({2*E*H*4/ms/1e6:.0f} GB/s algorithmic, {4*E*H*H*3/ms/1e9:.0f} TF issued)")
L.difusco_debug_set(2, 0)

# phase timestamps of body 2 of every wave of the pipelined kernel (s_memtime ticks, 100 MHz)
import numpy as np  # noqa: E402
nw = 256 * 4
dbg = torch.zeros(nw * 32, dtype=torch.int64, device=dev)
L.difusco_debug_set_ptr.argtypes = [ctypes.c_int, ctypes.c_void_p]
L.difusco_debug_set_ptr(1, ctypes.c_void_p(dbg.data_ptr()))
L.difusco_debug_set(2, 1)
e, h = e0.clone(), h0.clone()
run(e, h)
torch.cuda.synchronize()
L.difusco_debug_set_ptr(1, None)
L.difusco_debug_set(2, 0)
d = dbg.reshape(nw, 32)[:, :28].cpu().numpy().astype(np.float64)
d = d[d[:, 0] > 0]
if d.shape[0]:
    tot = d[:, 17] - d[:, 0]
    print(f"body 2 of {d.shape[0]} waves: {tot.mean():.0f} ticks (min {tot.min():.0f} max {tot.max():.0f})")
    names = [f"A{s} (GEMM 2 | epilogue 1)" for s in range(8)] + ["epilogue 2 (LN, SiLU, split)"] + \
            [f"B{t} (GEMM 1)" for t in range(8)]
    for i, nme in enumerate(names):
        seg = d[:, i + 1] - d[:, i]
        print(f"  {nme:30s} {seg.mean():8.0f} ticks {100 * seg.mean() / tot.mean():5.1f} %  (min {seg.min():.0f} max {seg.max():.0f})")
    for st_, b in ((2, 18), (3, 23)):
        pts = [d[:, st_], d[:, b], d[:, b + 1], d[:, b + 2], d[:, b + 3], d[:, st_ + 1]]
        lab = ["stage copy (ds_write + loads)", "GEMM 2 + quads", "neighbour sum", "output", "barrier wait"]
        print(f"  inside A{st_}: " + ", ".join(f"{l} {np.mean(pts[i + 1] - pts[i]):.0f}" for i, l in enumerate(lab)))

# profiling-only ablations of the pipelined kernel (difusco_debug_set(0, mask) while variant 1 is selected)
if len(sys.argv) > 4 and sys.argv[4] == "ablate":
    names = {0: "production", 1: "no neighbour-table gathers", 2: "no neighbour sum", 4: "no gate math", 7: "no gathers/sum/gate",
             8: "no residual read / e store", 16: "no LayerNorm/SiLU", 32: "no weight copy", 64: "no GEMM 2 MFMAs",
             128: "no e stream loads"}
    L.difusco_debug_set(2, 1)
    times = {m: [] for m in names}
    e, h = e0.clone(), h0.clone()
    for r in range(6):
        for m in names:
            L.difusco_debug_set(0, m)
            run(e, h)
            torch.cuda.synchronize()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(ITERS):
                run(e, h)
            t1.record()
            torch.cuda.synchronize()
            if r > 0:
                times[m].append(t0.elapsed_time(t1) / ITERS)
    L.difusco_debug_set(0, 0)
    L.difusco_debug_set(2, 0)
    for m, nme in names.items():
        ts = sorted(times[m])
        print(f"  pipelined, {nme:32s} median {ts[len(ts) // 2]:.3f} ms")
