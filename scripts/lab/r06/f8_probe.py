#!/usr/bin/env python
"""Round 6 probe: operand / scale / result layout of v_mfma_scale_f32_32x32x64_f8f6f4 (E4M3 x E4M3) and of v_permlane32_swap_b32 on
gfx950, checked against a float64 product under the layout hypothesis
    A[m][k]: lane = m + 32 * (k // 32), byte k % 32 of the lane's 8 dwords;  B[k][n] likewise with n;  one E8M0 scale per lane = per (row, 32-k block);
    D[m][n]: lane = n + 32 * ((m // 4) % 2), register = (m % 4) + 4 * (m // 8)      (the 32x32 accumulator layout of the fp16 MFMAs)
Profiling library only (stage_lab.hip: difusco_lab_f8_probe)."""
import ctypes
import os
import sys

import numpy as np
import torch

os.environ["DIFUSCO_PROFILING_LIB"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from difusco_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
L.difusco_lab_f8_probe.argtypes = [ctypes.c_void_p] * 8
g = torch.Generator().manual_seed(0)
M = N = 32
K = 64
A = torch.randn(M, K, generator=g).to(torch.float8_e4m3fn)
B = torch.randn(K, N, generator=g).to(torch.float8_e4m3fn)
MODE = sys.argv[1] if len(sys.argv) > 1 else "random"
ea = torch.randint(-3, 4, (M, 2), generator=g)      # scale exponent per (row, k block)
eb = torch.randint(-3, 4, (N, 2), generator=g)
if MODE == "noscale":
    ea, eb = torch.zeros_like(ea), torch.zeros_like(eb)
elif MODE == "uniform":      # one exponent per operand
    ea, eb = torch.full_like(ea, 2), torch.full_like(eb, -1)
elif MODE == "perblock":     # one exponent per k block, the same for every row
    ea = torch.tensor([[1, -2]]).expand(M, 2).contiguous()
    eb = torch.tensor([[3, 0]]).expand(N, 2).contiguous()
a_img = torch.zeros(64, 32, dtype=torch.uint8)
b_img = torch.zeros(64, 32, dtype=torch.uint8)
sa = torch.zeros(64, dtype=torch.int32)
sb = torch.zeros(64, dtype=torch.int32)
Ab, Bb = A.view(torch.uint8), B.view(torch.uint8)
for lane in range(64):
    m, kb = lane % 32, lane // 32
    a_img[lane] = Ab[m, 32 * kb:32 * kb + 32]
    b_img[lane] = Bb[32 * kb:32 * kb + 32, m]
    sa[lane] = 127 + int(ea[m, kb])
    sb[lane] = 127 + int(eb[m, kb])
sw = torch.arange(128, dtype=torch.int32)      # lane l: dwords (2 l, 2 l + 1)
out = torch.zeros(64 * 16 + 128 + 4, dtype=torch.float32, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())      # noqa: E731
ad, bd, sad, sbd, swd = (t.contiguous().to(dev) for t in (a_img.view(torch.int32), b_img.view(torch.int32), sa, sb, sw))
_lib.check(L.difusco_lab_f8_probe(P(ad), P(bd), P(sad), P(sbd), P(swd), P(out), None, None))
torch.cuda.synchronize()
o = out.cpu()
acc = o[:1024].reshape(64, 16)
As = A.double() * (2.0 ** ea.double()).repeat_interleave(32, dim=1)
Bs = B.double() * (2.0 ** eb.double()).repeat_interleave(32, dim=1).T
D = As @ Bs
got = torch.zeros(M, N, dtype=torch.float64)
for lane in range(64):
    n, hh = lane % 32, lane // 32
    for r in range(16):
        m = (r % 4) + 8 * (r // 4) + 4 * hh
        got[m, n] = acc[lane, r]
err = (got - D).abs().max().item()
print(f"f8 MFMA 32x32x64 (E4M3, per-lane E8M0 scales): max |err| vs float64 under the layout hypothesis {err:.3e} (|D| max {D.abs().max().item():.2f})")
if True:      # which hypothesis holds?
    D0 = A.double() @ B.double()
    print("  no scales at all:", (got - D0).abs().max().item(), " transposed result:", (got.T - D).abs().max().item())
    for name, ea_, eb_ in (("scale of lane % 32 for both k blocks", ea[:, :1].expand(-1, 2), eb[:, :1].expand(-1, 2)),
                           ("scale of lane % 32 + 32 for both", ea[:, 1:].expand(-1, 2), eb[:, 1:].expand(-1, 2)),
                           ("only A scales", ea, torch.zeros_like(eb)), ("only B scales", torch.zeros_like(ea), eb)):
        Dx = (A.double() * (2.0 ** ea_.double()).repeat_interleave(32, dim=1)) @ (B.double() * (2.0 ** eb_.double()).repeat_interleave(32, dim=1).T)
        print(f"  {name}: {(got - Dx).abs().max().item():.3e}")
    # H2: the two 16-byte register groups of a lane are the two scale blocks; block g takes its scale from lane (row + 32 g)
    a_r = a_img.view(torch.float8_e4m3fn).double().reshape(2, 32, 2, 16)      # [half][row][group][byte]
    b_r = b_img.view(torch.float8_e4m3fn).double().reshape(2, 32, 2, 16)
    D2 = torch.zeros(32, 32, dtype=torch.float64)
    for grp in range(2):
        sa_g = 2.0 ** (sa[32 * grp:32 * grp + 32].double() - 127)      # scale of block grp: lanes 32 grp ..
        sb_g = 2.0 ** (sb[32 * grp:32 * grp + 32].double() - 127)
        for h in range(2):
            D2 += (a_r[h, :, grp, :] * sa_g[:, None]) @ (b_r[h, :, grp, :] * sb_g[:, None]).T
    err2 = (got - D2).abs().max().item()
    print(f"  H2 (register groups = scale blocks, block g scaled by lanes 32 g ..): {err2:.3e} = {err2 / D2.abs().max().item():.1e} of max |D|")
    print("  got[0,:4]", got[0, :4].tolist(), "want", D[0, :4].tolist(), "ratio", (got[0, :4] / D[0, :4]).tolist())
swo = o[1024:1024 + 128].view(torch.int32).reshape(64, 2)
cv = o[1152:1153].view(torch.uint8)[:4].view(torch.float8_e4m3fn).float().tolist()
print('cvt_scalef32_pk_fp8_f32(100, -3; scale 256 | scale 1/16) ->', cv, ' (100/256 = 0.3906, -3/256 = -0.0117; 100*256 saturates; 100*(1/16) = 6.25, 100/(1/16) = 1600 saturates)')
# hypothesis: the upper 32 lanes of the first operand are exchanged with the lower 32 lanes of the second
exp = torch.zeros(64, 2, dtype=torch.int32)
for lane in range(64):
    p_, q_ = 2 * lane, 2 * lane + 1
    if lane < 32:
        exp[lane, 0], exp[lane, 1] = p_, 2 * (lane + 32)          # P own, Q <- P of lane + 32
    else:
        exp[lane, 0], exp[lane, 1] = 2 * (lane - 32) + 1, q_      # P <- Q of lane - 32, Q own
print("permlane32_swap hypothesis (P.upper <-> Q.lower):", bool(torch.equal(swo, exp)))
if not torch.equal(swo, exp):
    print(swo[:4].tolist(), swo[32:36].tolist())
print('MODE', MODE, ': lane-half hypothesis', 'holds' if err < 1e-4 * D.abs().max().item() else 'FAILS', '| H2', 'holds' if err2 < 1e-4 * D2.abs().max().item() else 'FAILS')

# accumulation into a LARGE accumulator: C = 2^11 x the size of the products' sum (the situation of a correction product)
if MODE == "noscale":
    cin = (torch.randn(64, 16, generator=g) * 2.0 ** 15).float()
    cd = cin.to(dev)
    out2 = torch.zeros_like(out)
    _lib.check(L.difusco_lab_f8_probe(P(ad), P(bd), P(sad), P(sbd), P(swd), P(out2), None, P(cd)))
    torch.cuda.synchronize()
    acc2 = out2.cpu()[:1024].reshape(64, 16)
    want = cin.double() + acc.double()          # (acc = the same products summed into 0)
    e2 = (acc2.double() - want).abs().max().item()
    ulp = float(torch.finfo(torch.float32).eps) * cin.abs().max().item()
    print(f"accumulating into |C| ~ 2^15..2^17: max |err| vs C + (the same sum into 0) {e2:.3e}; one fp32 ulp of max |C| = {ulp:.3e}")
