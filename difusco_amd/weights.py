"""Packing of the reference ``GNNEncoder`` ``state_dict`` into the flat fp32 blob the kernels read.

Key set (``difusco/models/gnn_encoder.py:303-347``), with or without the
Lightning ``model.`` prefix.  Three constant tables are appended, computed here with the same torch
CPU expressions the reference evaluates every forward pass (``models/nn.py:114-116``,
``gnn_encoder.py:214-215,242-243``) so the device kernels use bit-identical frequencies.
"""
import math
import re

import numpy as np
import torch

from . import _lib


def strip_prefix(state, prefix="model."):
    return {(k[len(prefix):] if k.startswith(prefix) else k): v for k, v in state.items()}


def infer_config(state):
    state = strip_prefix(state)
    hidden = int(state["node_embed.weight"].shape[0])
    n_layers = 1 + max(int(m.group(1)) for m in (re.match(r"layers\.(\d+)\.", k) for k in state) if m)
    out_channels = int(state["out.2.weight"].shape[0])
    return hidden, n_layers, out_channels


def constant_tables(hidden: int):
    half = hidden // 2
    freqs = torch.exp(-math.log(10000) * torch.arange(start=0, end=half, dtype=torch.float32) / half)

    def dim_t(n):
        d = torch.arange(n, dtype=torch.float32)
        return 10000 ** (2.0 * torch.div(d, 2, rounding_mode="trunc") / n)

    tabs = {"@time_freqs": freqs, "@dimt_pos": dim_t(half), "@dimt_scalar": dim_t(hidden)}
    # the generated-input kernels take one sincos per (sin, cos) feature pair: entries 2j, 2j+1 of a dim_t table must be equal
    # (include/difusco_hip.h, DIFUSCO_W_DIMT_SCALAR)
    for name in ("@dimt_pos", "@dimt_scalar"):
        assert torch.equal(tabs[name][0::2], tabs[name][1::2]), f"{name}: entries 2j and 2j+1 differ"
    return tabs


_SLAB_ORDER = [0, 2, 1, 3]   # slab position group g holds k group _SLAB_ORDER[g] (middle groups of 4 swapped)


def _slab_layout(p16: torch.Tensor, n_out: int, k: int) -> torch.Tensor:
    """[n_out, k] 16-bit -> [k/16 slabs][n_out rows][16] with the k permutation of linear_split.hip."""
    t = p16.reshape(n_out, k // 16, 4, 4)[:, :, _SLAB_ORDER, :]
    return t.permute(1, 0, 2, 3).contiguous().reshape(-1)


def pow2_scale(maxabs: torch.Tensor) -> torch.Tensor:
    """Elementwise 2^k (float32) with maxabs * 2^k in [2^14, 2^15); k clamped to [-100, 100] (zero / denormal / huge maxima
    get the clamp).  The same rule as ``pow2_scale_for`` in csrc/common.h."""
    m = maxabs.detach().float().abs().reshape(-1)
    ex = ((m.view(torch.int32) >> 23) & 255).clamp(41, 241)          # biased exponent
    k = (141 - ex).to(torch.float32)
    return torch.pow(torch.tensor(2.0), k).reshape(maxabs.shape)


def split_planes(w: torch.Tensor, per_row: bool = False) -> torch.Tensor:
    """[n_out, k] fp32 -> five 16-bit planes: bf16 hi | mid | lo, then fp16 hi | lo (all RNE, each the
    rounding of what the previous planes left), every plane in the slab layout above, followed by n_out
    floats: the inverse power-of-two scales of the fp16 planes.

    The fp16 planes are those of ``w * 2^k`` with ``max|w| * 2^k`` in [2^14, 2^15) - one k for the matrix, or one per
    output row (``per_row``).  fp16 is normal only down to 2^-14: unscaled, the low plane of every ``|w| < 2^-3`` would be
    a subnormal with an absolute 2^-25 floor instead of 11 further significand bits.  The scale is exact; the kernels
    multiply the accumulator by the stored ``2^-k`` where they add the bias.  Returned as a flat fp32-typed view
    (5*n_out*k/2 + n_out floats) ready to be copied into the blob."""
    n_out, k = w.shape
    w = w.detach().float()
    planes, rest = [], w.clone()
    for _ in range(3):
        p = rest.to(torch.bfloat16)
        rest = rest - p.float()
        planes.append(_slab_layout(p.view(torch.int16), n_out, k))
    if per_row:
        scale = pow2_scale(w.abs().amax(dim=1, keepdim=True))
    else:
        scale = pow2_scale(w.abs().amax().reshape(1, 1)).expand(n_out, 1)
    rest = w * scale
    for _ in range(2):
        p = rest.to(torch.float16)
        rest = rest - p.float()
        planes.append(_slab_layout(p.view(torch.int16), n_out, k))
    inv = (1.0 / scale).reshape(-1).contiguous()
    return torch.cat([torch.cat(planes).view(torch.float32), inv])


def plane_scale_inv(planes: torch.Tensor, n_out: int, k: int) -> torch.Tensor:
    """The n_out inverse fp16 scales stored behind the five planes of ``split_planes``."""
    return planes[5 * n_out * k // 2: 5 * n_out * k // 2 + n_out]


LOG2E = 1.4426950408889634


def fused_scales(w_c: torch.Tensor, w_o: torch.Tensor, g_o: torch.Tensor, b_o: torch.Tensor) -> torch.Tensor:
    """The 8-float operand-scale record of the fused edge kernel (include/difusco_hip.h: DIFUSCO_WL_FUSED_SCALES):
    {log2(e) 2^-kc, 2^-(ko+ka) / log2(e), log2(e), 2^-ka, 0...}.  ka comes from a bound that needs no data: the kernel's GEMM 2
    operand is a log2(e), a = SiLU(z), z = LN(y) g_o + b_o with |LN(y)| <= sqrt(H - 1) < 16, and |SiLU(z)| <= max(|z|, 0.2785)."""
    inv_c = 1.0 / pow2_scale(w_c.detach().float().abs().amax().reshape(1))[0]      # the per-matrix scales of split_planes
    inv_o = 1.0 / pow2_scale(w_o.detach().float().abs().amax().reshape(1))[0]
    c = torch.tensor(LOG2E, dtype=torch.float32)
    bound = torch.clamp(16.0 * g_o.detach().float().abs().max() + b_o.detach().float().abs().max(), min=0.2785) * c
    sa = pow2_scale(bound.reshape(1))[0]
    return torch.stack([inv_c * c, (inv_o / sa) / c, c, 1.0 / sa] + [torch.tensor(0.0)] * 4).float()


def node4_fused_vectors(state, l: int, hidden: int, planes: torch.Tensor):
    """Bias [4H] and column scales [2][4H] with which the node linear U | V | A | B produces the rows the FUSED edge kernel reads
    (include/difusco_hip.h, ABI 11 note): A | B columns in the log2(e) domain, b_C folded into the A columns.  ``planes`` = the
    packed split planes of the node linear (their tail holds the fp16 row scales 2^-k_f)."""
    c = torch.tensor(LOG2E, dtype=torch.float32)
    b = [state[f"layers.{l}.{m}.bias"].float() for m in "UVAB"]
    bias = torch.cat([b[0], b[1], (b[2] + state[f"layers.{l}.C.bias"].float()) * c, b[3] * c])
    col = torch.cat([torch.ones(2 * hidden), torch.full((2 * hidden,), LOG2E)]).float()
    w_inv = plane_scale_inv(planes, 4 * hidden, hidden)
    return bias, torch.cat([w_inv * col, col])


def pack_state_dict(state) -> torch.Tensor:
    """-> 1-D fp32 CPU tensor laid out per ``difusco_weights_layout``."""
    state = {k: (v.detach().float().cpu() if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v)).float())
             for k, v in strip_prefix(state).items()}
    hidden, n_layers, out_channels = infer_config(state)
    offsets, total = _lib.weights_layout(hidden, n_layers, out_channels)
    blob = torch.zeros(total, dtype=torch.float32)
    consts = constant_tables(hidden)

    def put(idx, tensor):
        flat = tensor.reshape(-1)
        blob[offsets[idx]: offsets[idx] + flat.numel()] = flat

    def fetch(name):
        if name.startswith("@planes:"):
            return split_planes(state[name[len("@planes:"):]].reshape(hidden, hidden))
        return consts[name] if name.startswith("@") else state[name]

    for i, name in enumerate(_lib.W_GLOBAL):
        put(i, fetch(name))
    for l in range(n_layers):
        base = len(_lib.W_GLOBAL) + l * len(_lib.W_LAYER)
        node4_planes = None      # split planes of this layer's [4H, H] node linear: needed by three entries, computed once
        for i, name in enumerate(_lib.W_LAYER):
            if name in ("@planes:@node4.weight", "@node4.fused_bias", "@node4.fused_scale") and node4_planes is None:
                node4_planes = split_planes(torch.cat([state[f"layers.{l}.{m}.weight"] for m in "UVAB"], dim=0), per_row=True)
            if name == "@node4.weight":      # rows U | V | A | B  -> one [4H, H] linear on node rows
                t = torch.cat([state[f"layers.{l}.{m}.weight"] for m in "UVAB"], dim=0)
            elif name == "@node4.bias":
                t = torch.cat([state[f"layers.{l}.{m}.bias"] for m in "UVAB"], dim=0)
            elif name == "@planes:@node4.weight":
                t = node4_planes
            elif name == "@node4.fused_bias":
                t = node4_fused_vectors(state, l, hidden, node4_planes)[0]
            elif name == "@node4.fused_scale":
                t = node4_fused_vectors(state, l, hidden, node4_planes)[1]
            elif name == "@fused_scales":
                t = fused_scales(state[f"layers.{l}.C.weight"], state[f"per_layer_out.{l}.2.weight"],
                                 state[f"per_layer_out.{l}.0.weight"], state[f"per_layer_out.{l}.0.bias"])
            else:
                t = fetch(name.format(l=l))
            put(base + i, t)
    return blob
