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

    return {"@time_freqs": freqs, "@dimt_pos": dim_t(half), "@dimt_scalar": dim_t(hidden)}


_SLAB_ORDER = [0, 2, 1, 3]   # slab position group g holds k group _SLAB_ORDER[g] (middle groups of 4 swapped)


def _slab_layout(p16: torch.Tensor, n_out: int, k: int) -> torch.Tensor:
    """[n_out, k] 16-bit -> [k/16 slabs][n_out rows][16] with the k permutation of linear_split.hip."""
    t = p16.reshape(n_out, k // 16, 4, 4)[:, :, _SLAB_ORDER, :]
    return t.permute(1, 0, 2, 3).contiguous().reshape(-1)


def split_planes(w: torch.Tensor) -> torch.Tensor:
    """[n_out, k] fp32 -> five 16-bit planes: bf16 hi | mid | lo, then fp16 hi | lo (all RNE, each the
    rounding of what the previous planes left), every plane in the slab layout above.  Returned as a
    flat fp32-typed view (5*n_out*k/2 floats) ready to be copied into the blob."""
    n_out, k = w.shape
    planes, rest = [], w.clone()
    for _ in range(3):
        p = rest.to(torch.bfloat16)
        rest = rest - p.float()
        planes.append(_slab_layout(p.view(torch.int16), n_out, k))
    rest = w.clone()
    for _ in range(2):
        p = rest.to(torch.float16)
        rest = rest - p.float()
        planes.append(_slab_layout(p.view(torch.int16), n_out, k))
    return torch.cat(planes).view(torch.float32)


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
        for i, name in enumerate(_lib.W_LAYER):
            if name == "@node4.weight":      # rows U | V | A | B  -> one [4H, H] linear on node rows
                t = torch.cat([state[f"layers.{l}.{m}.weight"] for m in "UVAB"], dim=0)
            elif name == "@node4.bias":
                t = torch.cat([state[f"layers.{l}.{m}.bias"] for m in "UVAB"], dim=0)
            elif name == "@planes:@node4.weight":
                t = split_planes(torch.cat([state[f"layers.{l}.{m}.weight"] for m in "UVAB"], dim=0))
            else:
                t = fetch(name.format(l=l))
            put(base + i, t)
    return blob
