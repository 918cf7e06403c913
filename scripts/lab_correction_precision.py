#!/usr/bin/env python
"""could the two CORRECTION products of the split-precision E-row GEMMs (hi*lo, lo*hi) run on cheaper arithmetic?
A CPU emulation of one [E,256] x [256,256] product with the operand statistics of the fused layer (N(0,1) rows, weights
U(-1/16, 1/16)), every variant accumulated in float64 so that only the OPERAND roundings differ:

    fp16x3        hi*hi + hi*lo + lo*hi on fp16 planes                         (the default)
    bf16x3        the same on bf16 planes                                        (--precision bf16x3: logits L_inf ~8e-6 measured)
    fp16+bf16corr hi*hi on fp16 planes, corrections on bf16 planes (bf16(hi) * bf16(lo))
    fp16+fp8corr  hi*hi on fp16 planes, corrections on FP8 E4M3 planes, lo scaled by 2^11 (v_mfma_f32_32x32x64_f8f6f4 class)

Printed: max / rms relative-to-row-scale error of the product.  The measured logits errors of the two existing precisions
(fp16x3 1.4e-6, bf16x3 8e-6 on TSP-1000) calibrate what the others would give.     python scripts/lab_correction_precision.py
"""
import torch

torch.manual_seed(0)
E, K, F = 4096, 256, 256
x = torch.randn(E, K, dtype=torch.float32)
w = (torch.rand(F, K, dtype=torch.float32) * 2 - 1) / 16
exact = x.double() @ w.double().T
scale = exact.abs().mean()


def planes(t, dt):
    hi = t.to(dt)
    lo = (t - hi.float()).to(dt)
    return hi.float().double(), lo.float().double()


def fp8(t, pre=1.0):      # round to E4M3 after an exact power-of-two pre-scale, undo the scale
    return (t * pre).to(torch.float8_e4m3fn).float().double() / pre


def report(name, y):
    err = (y - exact).abs()
    print(f"{name:16s} max {err.max().item() / scale:.3e}   rms {err.pow(2).mean().sqrt().item() / scale:.3e}   (relative to mean |y| = {scale:.3f})")


xh, xl = planes(x, torch.float16)
wh, wl = planes(w * 2 ** 17, torch.float16)      # (weights scaled into fp16's upper binades, as weights.py does)
wh, wl = wh / 2 ** 17, wl / 2 ** 17
report("fp16x3", xh @ wh.T + xh @ wl.T + xl @ wh.T)
bxh, bxl = planes(x, torch.bfloat16)
bwh, bwl = planes(w, torch.bfloat16)
report("bf16x3", bxh @ bwh.T + bxh @ bwl.T + bxl @ bwh.T)
# corrections on bf16: the factors of the correction products rounded to bf16 (8 significand bits)
b = lambda t: t.float().to(torch.bfloat16).float().double()
report("fp16+bf16corr", xh @ wh.T + b(xh) @ b(wl).T + b(xl) @ b(wh).T)
# corrections on FP8 E4M3 (4 significand bits, max 448): every plane pre-scaled by an exact power of two into [64, 256)
report("fp16+fp8corr", xh @ wh.T + fp8(xh.float(), 2.0 ** 5) @ fp8(wl.float(), 2.0 ** 22).T + fp8(xl.float(), 2.0 ** 15) @ fp8(wh.float(), 2.0 ** 11).T)


# lo planes with FEWER significand bits (same fp16 format, low mantissa bits zero): the multiplier arrays see fewer toggling partial
# products - an energy lever under the socket power cap if the matrix cores' power is data dependent (profiles/r04/lab_power_randn_vs_zeros.txt)
def trunc_lo(lo, bits):      # keep `bits` significand bits of an fp16-valued double tensor (round to nearest)
    m, e = torch.frexp(lo)
    return torch.ldexp(torch.round(m * 2 ** bits) / 2 ** bits, e)


for bits in (10, 8, 6, 4, 2):
    report(f"fp16x3 lo:{bits}b", xh @ wh.T + xh @ trunc_lo(wl, bits).T + trunc_lo(xl, bits) @ wh.T)
