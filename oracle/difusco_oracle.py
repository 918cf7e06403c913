"""CPU oracle for the DIFUSCO inference denoise step.  TEST INFRASTRUCTURE ONLY.

This file is the *checker*, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it.
``difusco_amd`` (the product) must never import anything from ``oracle/``.

It is an op-for-op CPU restatement (PyTorch CPU, fp32) of the reference's hot path,
written functionally over a flat ``{name: tensor}`` parameter dict that uses the
reference's ``state_dict`` key set.  Every function cites the reference
``file:line`` (relative to ``/root/reference/``) whose arithmetic it follows.

Parity pin status (see tests/golden/PROVENANCE.md and DESIGN.md):
  * schedules / Q_bar / alphabar, timestep embedding, posteriors, dense encoder:
    pinned against the imported reference (pure reference arithmetic).
  * sparse encoders: pinned against reference Python + a *substitute* neighbour sum
    (``index_add_``), because torch-sparse 0.6.15 / torch-scatter 2.0.9
    (``environment.yml:131,133``) are not installable here and their source is not under
    ``/root/reference``.  The aggregation semantics are taken from the call site
    ``difusco/models/gnn_encoder.py:177-191``: out[r] = sum of value[k] over edge_index[0][k]==r.
    For the sparse aggregation alone the parity is therefore "unpinned" against torch-sparse.
  * aggregation = "mean" / "max": the dense branch is pinned against the imported reference
    (tests/golden/make_golden_agg.py); the sparse branch (torch_sparse.mean / max) is restated in
    ``segment_aggregate`` and must reproduce the reference's DENSE outputs on the complete graph.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


# ----------------------------------------------------------------------------------------------
# Diffusion tables and inference schedule (host side, float64)
# ----------------------------------------------------------------------------------------------

def _beta_schedule(T: int, schedule: str):
    """difusco/utils/diffusion_schedulers.py:16-23 and :53-60."""
    if schedule == "linear":
        return np.linspace(1e-4, 2e-2, T)
    if schedule == "cosine":
        def cosn(t):
            return np.cos(math.pi * 0.5 * (t / T + 0.008) / (1 + 0.008)) ** 2
        ab = cosn(np.arange(0, T + 1, 1)) / cosn(0)
        return np.clip(1 - (ab[1:] / ab[:-1]), None, 0.999)
    raise ValueError(f"unknown diffusion schedule {schedule}")


class CategoricalTables:
    """Q_bar[t] = prod_{s<=t} Q_s, Q_s = (1-beta_s) I + beta_s/2 * 11^T.
    difusco/utils/diffusion_schedulers.py:49-72."""

    def __init__(self, T: int = 1000, schedule: str = "linear"):
        self.T = T
        self.beta = _beta_schedule(T, schedule)
        qbar = [np.eye(2)]
        for b in self.beta:
            q = (1 - b) * np.eye(2) + (b / 2) * np.ones((2, 2))
            qbar.append(qbar[-1] @ q)
        self.Q_bar = np.stack(qbar, axis=0)  # [T+1, 2, 2] float64


class GaussianTables:
    """alphabar = cumprod([1, 1-beta]).  difusco/utils/diffusion_schedulers.py:12-28."""

    def __init__(self, T: int = 1000, schedule: str = "linear"):
        self.T = T
        self.beta = _beta_schedule(T, schedule)
        self.alpha = np.concatenate((np.array([1.0]), 1 - self.beta))
        self.alphabar = np.cumprod(self.alpha)


def inference_schedule(kind: str, T: int, steps: int, i: int):
    """(t1, t2) of inference step i.  difusco/utils/diffusion_schedulers.py:91-111."""
    assert 0 <= i < steps
    if kind == "linear":
        t1 = T - int((float(i) / steps) * T)
        t2 = T - int((float(i + 1) / steps) * T)
    elif kind == "cosine":
        t1 = T - int(np.sin((float(i) / steps) * np.pi / 2) * T)
        t2 = T - int(np.sin((float(i + 1) / steps) * np.pi / 2) * T)
    else:
        raise ValueError(f"Unknown inference schedule: {kind}")
    return int(np.clip(t1, 1, T)), int(np.clip(t2, 0, T - 1))


# ----------------------------------------------------------------------------------------------
# Embeddings
# ----------------------------------------------------------------------------------------------

def timestep_embedding(t: torch.Tensor, dim: int, max_period: int = 10000) -> torch.Tensor:
    """[cos(t f), sin(t f)], f_k = exp(-ln(max_period) k / half).  difusco/models/nn.py:103-121."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def _dim_t(n: int, temperature: float = 10000.0) -> torch.Tensor:
    k = torch.arange(n, dtype=torch.float32)
    return temperature ** (2.0 * torch.div(k, 2, rounding_mode="trunc") / n)


def _interleave_sin_cos(pos: torch.Tensor) -> torch.Tensor:
    """even channels -> sin, odd channels -> cos, interleaved back in place."""
    out = torch.empty_like(pos)
    out[..., 0::2] = pos[..., 0::2].sin()
    out[..., 1::2] = pos[..., 1::2].cos()
    return out


def position_embedding_sine(points: torch.Tensor, hidden: int) -> torch.Tensor:
    """PositionEmbeddingSine(hidden//2, normalize=True): coords * 2pi / dim_t, cat(y, x).
    difusco/models/gnn_encoder.py:194-227.  points [..., 2] -> [..., hidden]."""
    n = hidden // 2
    dt = _dim_t(n)
    y = points[..., 0] * (2 * math.pi)
    x = points[..., 1] * (2 * math.pi)
    pos_x = _interleave_sin_cos(x[..., None] / dt)
    pos_y = _interleave_sin_cos(y[..., None] / dt)
    return torch.cat((pos_y, pos_x), dim=-1).contiguous()


def scalar_embedding_sine(x: torch.Tensor, hidden: int) -> torch.Tensor:
    """ScalarEmbeddingSine / ScalarEmbeddingSine1D (normalize=False): x / dim_t, sin/cos interleave.
    difusco/models/gnn_encoder.py:230-271.  x [...] -> [..., hidden]."""
    return _interleave_sin_cos(x[..., None] / _dim_t(hidden))


def time_features(p: Params, t: torch.Tensor, hidden: int) -> torch.Tensor:
    """time_embed MLP on the sinusoidal embedding.  gnn_encoder.py:311-315,396."""
    te = timestep_embedding(t, hidden)
    te = F.linear(te, p["time_embed.0.weight"], p["time_embed.0.bias"])
    te = F.relu(te)
    return F.linear(te, p["time_embed.2.weight"], p["time_embed.2.bias"])


def n_layers_of(p: Params) -> int:
    return 1 + max(int(k.split(".")[1]) for k in p if k.startswith("layers."))


def strip_prefix(state: Params, prefix: str = "model.") -> Params:
    return {(k[len(prefix):] if k.startswith(prefix) else k): v for k, v in state.items()}


# ----------------------------------------------------------------------------------------------
# One anisotropic gated-GCN layer + the per-layer epilogue
# ----------------------------------------------------------------------------------------------

def _lin(p: Params, name: str, x: torch.Tensor) -> torch.Tensor:
    return F.linear(x, p[name + ".weight"], p[name + ".bias"])


def _ln(p: Params, name: str, x: torch.Tensor) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), p[name + ".weight"], p[name + ".bias"], 1e-5)


def segment_sum(values: torch.Tensor, rows: torch.Tensor, n_rows: int) -> torch.Tensor:
    """out[r] = sum_{k: rows[k]==r} values[k]; empty rows -> 0.  Call-site semantics of
    torch_sparse.sum(SparseTensor(row, col, value), dim=1) at gnn_encoder.py:177-191."""
    out = torch.zeros((n_rows, values.shape[-1]), dtype=values.dtype)
    out.index_add_(0, rows, values)
    return out


def segment_aggregate(values: torch.Tensor, rows: torch.Tensor, n_rows: int, aggregation: str = "sum") -> torch.Tensor:
    """Aggr over the entries of each row, gnn_encoder.py:177-191: torch_sparse.sum / mean / max of
    SparseTensor(row, col, value) over dim=1.  torch-sparse 0.6.15 (environment.yml:131) reduces over dim 1 with
    torch_scatter.segment_csr(value, rowptr, reduce=...) (torch_sparse/reduce.py; torch-scatter 2.0.9), whose published
    semantics are restated here: "mean" = sum / number of entries of the row, "max" = element-wise maximum of the row's
    entries (values only, like ``torch.max(dim)[0]``); a row without entries yields 0 for every reduction.  The DENSE
    branch of the same function (:169-175: ``sum / sum(graph)`` with graph = ones, ``torch.max(Vh, dim=2)[0]``) is pure torch
    and pins these semantics through the reference-generated dense fixtures (tests/golden/tsp_dense_agg_*.npz)."""
    if aggregation == "sum":
        return segment_sum(values, rows, n_rows)
    count = torch.zeros(n_rows, dtype=torch.long).index_add_(0, rows, torch.ones_like(rows))
    if aggregation == "mean":
        return segment_sum(values, rows, n_rows) / count.clamp(min=1).to(values.dtype)[:, None]
    if aggregation == "max":
        out = torch.full((n_rows, values.shape[-1]), float("-inf"), dtype=values.dtype)
        out = out.scatter_reduce(0, rows[:, None].expand_as(values), values, reduce="amax", include_self=True)
        return torch.where((count > 0)[:, None], out, torch.zeros_like(out))
    raise ValueError(f"unknown aggregation {aggregation}")


def sparse_layer(p: Params, l: int, h: torch.Tensor, e: torch.Tensor, ei: torch.Tensor,
                 tbias: torch.Tensor, time_on_edge: bool, v_on_edges: bool = True, aggregation: str = "sum"):
    """GNNLayer.forward(sparse=True, mode='direct') + the epilogue of sparse_encoding.
    gnn_encoder.py:67-142 (layer), :442-449 (epilogue), :339-347 (per_layer_out).
    ei[0] = centre node i (row), ei[1] = neighbour j.  tbias: [1, H] = time_layer(temb)."""
    L = f"layers.{l}."
    row, col = ei[0], ei[1]
    Uh = _lin(p, L + "U", h)                                  # :94
    if v_on_edges:
        Vh = _lin(p, L + "V", h[col])                         # :99  (reference order: V on E rows)
    else:
        Vh = _lin(p, L + "V", h)[col]                         # hoisted, mathematically identical
    Ah = _lin(p, L + "A", h)                                  # :102
    Bh = _lin(p, L + "B", h)                                  # :103
    Ce = _lin(p, L + "C", e)                                  # :104
    e_new = Ah[col] + Bh[row] + Ce                            # :110
    gates = torch.sigmoid(e_new)                              # :112
    h_new = Uh + segment_aggregate(gates * Vh, row, h.shape[0], aggregation)   # :115,163,177-191
    h_new = F.relu(_ln(p, L + "norm_h", h_new))               # :123,134
    e_new = F.relu(_ln(p, L + "norm_e", e_new))               # :131,135
    if time_on_edge:
        e_new = e_new + tbias                                 # :445
    else:
        h_new = h_new + tbias                                 # :447
    h_out = h + h_new                                         # :448
    o = _ln(p, f"per_layer_out.{l}.0", e_new)
    o = F.silu(o)
    o = _lin(p, f"per_layer_out.{l}.2", o)
    e_out = e + o                                             # :449
    return h_out, e_out


def _layer_time_bias(p: Params, l: int, temb: torch.Tensor) -> torch.Tensor:
    """time_embed_layers[l] = ReLU -> Linear(H/2, H).  gnn_encoder.py:329-337."""
    return _lin(p, f"time_embed_layers.{l}.1", F.relu(temb))


def group_norm_head(p: Params, feat: torch.Tensor, groups: int = 32) -> torch.Tensor:
    """GroupNorm32(32, H) over (H/32 channels x ALL rows of the call), ReLU, 1x1 conv.
    gnn_encoder.py:316-322,400-401,412-413; nn.py:17-19,93-100.
    feat: [R, H] (all rows of one call share the statistics - SURVEY F3).  -> [R, C]."""
    R, H = feat.shape
    x = feat.t().reshape(1, H, R, 1)
    x = F.group_norm(x.float(), groups, p["out.0.weight"], p["out.0.bias"], 1e-5)
    x = F.relu(x)
    x = F.conv2d(x, p["out.2.weight"], p["out.2.bias"])
    return x.reshape(-1, R).t().contiguous()


def encoder_sparse_edge(p: Params, points: torch.Tensor, xt: torch.Tensor, t: torch.Tensor,
                        ei: torch.Tensor, v_on_edges: bool = True,
                        return_features: bool = False, aggregation: str = "sum") -> torch.Tensor:
    """GNNEncoder.sparse_forward (TSP).  gnn_encoder.py:383-402,416-450.
    points [N,2] f32, xt [E] f32, t [1] f32, ei [2,E] i64 -> logits [E, C]."""
    H = p["node_embed.weight"].shape[0]
    h = _lin(p, "node_embed", position_embedding_sine(points.float(), H))      # :394
    e = _lin(p, "edge_embed", scalar_embedding_sine(xt.float(), H))            # :395
    temb = time_features(p, t, H)                                              # :396
    ei = ei.long()
    for l in range(n_layers_of(p)):
        h, e = sparse_layer(p, l, h, e, ei, _layer_time_bias(p, l, temb), True, v_on_edges, aggregation)
    out = group_norm_head(p, e)
    return (out, h, e) if return_features else out


def encoder_sparse_node(p: Params, xt: torch.Tensor, t: torch.Tensor, ei: torch.Tensor,
                        v_on_edges: bool = True, return_features: bool = False, aggregation: str = "sum") -> torch.Tensor:
    """GNNEncoder.sparse_forward_node_feature_only (MIS).  gnn_encoder.py:404-414.
    xt [N] f32, ei [2,E] -> logits [N, C]."""
    H = p["node_embed.weight"].shape[0]
    h = _lin(p, "node_embed", scalar_embedding_sine(xt.float(), H))            # :405
    e = torch.zeros(ei.shape[1], H)                                           # :407
    temb = time_features(p, t, H)
    ei = ei.long()
    for l in range(n_layers_of(p)):
        h, e = sparse_layer(p, l, h, e, ei, _layer_time_bias(p, l, temb), False, v_on_edges, aggregation)
    out = group_norm_head(p, h)
    return (out, h, e) if return_features else out


def encoder_sparse_f64(p: Params, points: Optional[torch.Tensor], xt: torch.Tensor, t: torch.Tensor, ei: torch.Tensor,
                       node_feature_only: bool = False) -> torch.Tensor:
    """The same network as encoder_sparse_edge / encoder_sparse_node evaluated in FLOAT64 from the same fp32 inputs and fp32
    parameters: the exact-arithmetic value that every fp32 implementation (the reference, this oracle, the HIP path) approximates.
    Used by the adversarial-weight tests (tests/test_gpu_round6.py) to CALIBRATE a bound: with outlier channels or a residual
    stream that grows by orders of magnitude, two faithful fp32 evaluations differ from each other by more than 1e-5, so the HIP
    path is held to "no further from the float64 value than a small multiple of what the fp32 oracle itself is".
    Same op sequence and citations as the fp32 functions above (gnn_encoder.py:383-450, :67-142, nn.py:93-121)."""
    q = {k: v.double() for k, v in p.items()}
    H = q["node_embed.weight"].shape[0]
    te = timestep_embedding(t, H).double()
    te = F.relu(F.linear(te, q["time_embed.0.weight"], q["time_embed.0.bias"]))
    temb = F.linear(te, q["time_embed.2.weight"], q["time_embed.2.bias"])
    ei = ei.long()
    if node_feature_only:
        h = _lin(q, "node_embed", scalar_embedding_sine(xt.double(), H))
        e = torch.zeros(ei.shape[1], H, dtype=torch.float64)
    else:
        h = _lin(q, "node_embed", position_embedding_sine(points.double(), H))
        e = _lin(q, "edge_embed", scalar_embedding_sine(xt.double(), H))
    for l in range(n_layers_of(q)):
        h, e = sparse_layer(q, l, h, e, ei, _layer_time_bias(q, l, temb), not node_feature_only)
    feat = h if node_feature_only else e
    R = feat.shape[0]
    x = F.group_norm(feat.t().reshape(1, H, R, 1), 32, q["out.0.weight"], q["out.0.bias"], 1e-5)
    x = F.conv2d(F.relu(x), q["out.2.weight"], q["out.2.bias"])
    return x.reshape(-1, R).t().contiguous()


def encoder_dense(p: Params, points: torch.Tensor, xt: torch.Tensor, t: torch.Tensor, aggregation: str = "sum") -> torch.Tensor:
    """GNNEncoder.dense_forward.  gnn_encoder.py:350-381 and the dense branches of GNNLayer
    (:97,108,119-129,169-175).  points [B,V,2], xt [B,V,V] f32, t [B] -> logits [B,C,V,V].
    Statistics of the head GroupNorm are per sample here (tensor is (B,H,V,V))."""
    H = p["node_embed.weight"].shape[0]
    B, V, _ = points.shape
    h = _lin(p, "node_embed", position_embedding_sine(points.float(), H))      # [B,V,H]
    e = _lin(p, "edge_embed", scalar_embedding_sine(xt.float(), H))            # [B,V,V,H]
    temb = time_features(p, t, H)                                              # [B,H/2]
    for l in range(n_layers_of(p)):
        L = f"layers.{l}."
        Uh = _lin(p, L + "U", h)
        Vh = _lin(p, L + "V", h)[:, None, :, :]                                # j on dim 2
        Ah = _lin(p, L + "A", h)
        Bh = _lin(p, L + "B", h)
        Ce = _lin(p, L + "C", e)
        e_new = Ah[:, None, :, :] + Bh[:, :, None, :] + Ce                     # :108
        gates = torch.sigmoid(e_new)
        if aggregation == "mean":                                              # :170-171 (graph = ones, :364: the divisor is V)
            agg = torch.sum(gates * Vh, dim=2) / torch.sum(torch.ones(B, V, V, dtype=torch.long), dim=2).unsqueeze(-1).type_as(Vh)
        elif aggregation == "max":                                             # :172-173
            agg = torch.max(gates * Vh, dim=2)[0]
        else:
            agg = torch.sum(gates * Vh, dim=2)                                 # :175
        h_new = Uh + agg
        h_new = F.relu(_ln(p, L + "norm_h", h_new))
        e_new = F.relu(_ln(p, L + "norm_e", e_new))
        e_new = e_new + _layer_time_bias(p, l, temb)[:, None, None, :]         # :375
        h = h + h_new
        o = _lin(p, f"per_layer_out.{l}.2", F.silu(_ln(p, f"per_layer_out.{l}.0", e_new)))
        e = e + o
    x = e.permute(0, 3, 1, 2)
    x = F.group_norm(x.float(), 32, p["out.0.weight"], p["out.0.bias"], 1e-5)
    x = F.relu(x)
    return F.conv2d(x, p["out.2.weight"], p["out.2.bias"])


# ----------------------------------------------------------------------------------------------
# Posteriors
# ----------------------------------------------------------------------------------------------

def categorical_posterior_prob(tables: CategoricalTables, t: int, target_t: Optional[int],
                               x0_prob: torch.Tensor, xt: torch.Tensor) -> torch.Tensor:
    """p(x_{target_t}=1 | x_t, x0_pred) before sampling.  difusco/pl_meta_model.py:102-137.
    x0_prob [..., 2], xt any shape with x0_prob.shape[:-1] elements."""
    if target_t is None:
        target_t = t - 1
    Q_t = np.linalg.inv(tables.Q_bar[target_t]) @ tables.Q_bar[t]              # :115
    Q_t = torch.from_numpy(Q_t).float()
    Qb_t = torch.from_numpy(tables.Q_bar[t]).float()
    Qb_s = torch.from_numpy(tables.Q_bar[target_t]).float()
    x = F.one_hot(xt.long(), num_classes=2).float().reshape(x0_prob.shape)     # :122-123
    part1 = torch.matmul(x, Q_t.permute((1, 0)).contiguous())                  # :125
    den0 = (Qb_t[0] * x).sum(dim=-1, keepdim=True)                             # :127
    prob0 = (part1 * Qb_s[0]) / den0                                           # :129
    s = prob0[..., 1] * x0_prob[..., 0]                                        # :131
    den1 = (Qb_t[1] * x).sum(dim=-1, keepdim=True)                             # :133
    prob1 = (part1 * Qb_s[1]) / den1                                           # :135
    return s + prob1[..., 1] * x0_prob[..., 1]                                 # :137


def categorical_posterior(tables: CategoricalTables, t: int, target_t: Optional[int],
                          x0_prob: torch.Tensor, xt: torch.Tensor, sparse: bool,
                          uniform: Optional[torch.Tensor] = None,
                          generator: Optional[torch.Generator] = None):
    """pl_meta_model.py:102-146.  Returns (x_next, prob).  With ``uniform`` given the Bernoulli
    draw is replaced by ``uniform < prob`` (injected-uniform test mode)."""
    tt = (t - 1) if target_t is None else target_t
    prob = categorical_posterior_prob(tables, t, tt, x0_prob, xt)
    if tt > 0:
        pc = prob.clamp(0, 1)                                                  # :140
        if uniform is not None:
            nxt = (uniform.reshape(pc.shape) < pc).float()
        else:
            nxt = torch.bernoulli(pc, generator=generator)
    else:
        nxt = prob.clamp(min=0)                                                # :142
    if sparse:
        nxt = nxt.reshape(-1)                                                  # :144-145
    return nxt, prob


def gaussian_posterior(tables: GaussianTables, t: int, target_t: Optional[int], pred: torch.Tensor,
                       xt: torch.Tensor, inference_trick: Optional[str] = "ddim",
                       noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """pl_meta_model.py:148-175.  ``noise`` (standard normal, same shape as xt) replaces
    randn_like in the DDPM branch when given."""
    tt = (t - 1) if target_t is None else target_t
    atbar = tables.alphabar[t]
    atbar_target = tables.alphabar[tt]
    if inference_trick is None or t <= 1:
        at = tables.alpha[t]
        z = torch.randn_like(xt) if noise is None else noise
        atbar_prev = tables.alphabar[t - 1]
        beta_tilde = tables.beta[t - 1] * (1 - atbar_prev) / (1 - atbar)       # :165
        out = float(1 / np.sqrt(at)) * (xt - float((1 - at) / np.sqrt(1 - atbar)) * pred)
        return out + float(np.sqrt(beta_tilde)) * z                            # :167-169
    if inference_trick == "ddim":
        out = float(np.sqrt(atbar_target / atbar)) * (xt - float(np.sqrt(1 - atbar)) * pred)
        return out + float(np.sqrt(1 - atbar_target)) * pred                   # :171-172
    raise ValueError(f"Unknown inference trick {inference_trick}")


# ----------------------------------------------------------------------------------------------
# Denoise steps (the drop-in methods) and the sampling loop
# ----------------------------------------------------------------------------------------------

def tsp_categorical_denoise_step(p, tables, points, xt, t, edge_index=None, target_t=None,
                                 uniform=None, generator=None, v_on_edges=True, return_aux=False, aggregation="sum"):
    """TSPModel.categorical_denoise_step.  difusco/pl_tsp_model.py:122-138.
    Sparse when edge_index is given (points [N,2], xt [E]); dense otherwise (points [B,V,2],
    xt [B,V,V]).  t / target_t are python ints (the reference passes np arrays of shape (1,))."""
    tf = torch.tensor([float(t)])
    if edge_index is not None:
        logits = encoder_sparse_edge(p, points, xt.float(), tf, edge_index, v_on_edges, aggregation=aggregation)
        prob0 = logits.reshape((1, points.shape[0], -1, 2)).softmax(dim=-1)    # :135
        nxt, prob = categorical_posterior(tables, t, target_t, prob0, xt, True, uniform, generator)
    else:
        logits = encoder_dense(p, points, xt.float(), tf, aggregation)   # one shared timestep, shape (1,)
        prob0 = logits.permute((0, 2, 3, 1)).contiguous().softmax(dim=-1)      # :133
        nxt, prob = categorical_posterior(tables, t, target_t, prob0, xt, False, uniform, generator)
    return (nxt, logits, prob) if return_aux else nxt


def tsp_gaussian_denoise_step(p, tables, points, xt, t, edge_index=None, target_t=None,
                              inference_trick="ddim", noise=None, v_on_edges=True, return_aux=False, aggregation="sum"):
    """TSPModel.gaussian_denoise_step.  pl_tsp_model.py:140-151."""
    tf = torch.tensor([float(t)])
    if edge_index is not None:
        pred = encoder_sparse_edge(p, points, xt.float(), tf, edge_index, v_on_edges, aggregation=aggregation)
    else:
        pred = encoder_dense(p, points, xt.float(), tf, aggregation)
    pred = pred.squeeze(1)                                                     # :149
    nxt = gaussian_posterior(tables, t, target_t, pred, xt, inference_trick, noise)
    return (nxt, pred) if return_aux else nxt


def mis_categorical_denoise_step(p, tables, xt, t, edge_index, target_t=None, uniform=None,
                                 generator=None, v_on_edges=True, return_aux=False, aggregation="sum"):
    """MISModel.categorical_denoise_step.  difusco/pl_mis_model.py:118-128."""
    tf = torch.tensor([float(t)])
    logits = encoder_sparse_node(p, xt.float(), tf, edge_index, v_on_edges, aggregation=aggregation)
    prob0 = logits.reshape((1, xt.shape[0], -1, 2)).softmax(dim=-1)            # :126
    nxt, prob = categorical_posterior(tables, t, target_t, prob0, xt, True, uniform, generator)
    return (nxt, logits, prob) if return_aux else nxt


def mis_gaussian_denoise_step(p, tables, xt, t, edge_index, target_t=None, inference_trick="ddim",
                              noise=None, v_on_edges=True, return_aux=False, aggregation="sum"):
    """MISModel.gaussian_denoise_step.  pl_mis_model.py:130-140."""
    tf = torch.tensor([float(t)])
    pred = encoder_sparse_node(p, xt.float(), tf, edge_index, v_on_edges, aggregation=aggregation).squeeze(1)
    nxt = gaussian_posterior(tables, t, target_t, pred, xt, inference_trick, noise)
    return (nxt, pred) if return_aux else nxt


def duplicate_edge_index(edge_index: torch.Tensor, num_nodes: int, copies: int) -> torch.Tensor:
    """Disjoint union of ``copies`` replicas: ids of replica g offset by g*num_nodes.
    pl_meta_model.py:177-184."""
    ei = edge_index.reshape((2, 1, -1))
    off = torch.arange(0, copies).view(1, -1, 1) * num_nodes
    return (ei + off).reshape((2, -1))


# ----------------------------------------------------------------------------------------------
# Synthetic inputs and weights; used by tests / bench cpu_baseline only.
# ----------------------------------------------------------------------------------------------

def init_params(hidden: int, n_layers: int, out_channels: int, seed: int,
                randomize_zero_init: bool = True) -> Params:
    """Random weights with the reference module's shapes and PyTorch-default init scale:
    nn.Linear -> U(-1/sqrt(fan_in), 1/sqrt(fan_in)); LayerNorm/GroupNorm affine perturbed around
    (1, 0) so they are visible to a parity check.  per_layer_out.*.2 is zero-initialised upstream
    (gnn_encoder.py:343-345, nn.py:68-74) which hides all 12 layers from the edge output
     - it is re-randomised here unless randomize_zero_init=False."""
    g = torch.Generator().manual_seed(seed)
    H, T2 = hidden, hidden // 2
    p: Params = {}

    def lin(name, fo, fi):
        b = 1.0 / math.sqrt(fi)
        p[name + ".weight"] = (torch.rand(fo, fi, generator=g) * 2 - 1) * b
        p[name + ".bias"] = (torch.rand(fo, generator=g) * 2 - 1) * b

    def norm(name, n):
        p[name + ".weight"] = 1.0 + 0.1 * torch.randn(n, generator=g)
        p[name + ".bias"] = 0.1 * torch.randn(n, generator=g)

    lin("node_embed", H, H)
    lin("edge_embed", H, H)
    lin("time_embed.0", T2, H)
    lin("time_embed.2", T2, T2)
    norm("out.0", H)
    b = 1.0 / math.sqrt(H)
    p["out.2.weight"] = ((torch.rand(out_channels, H, generator=g) * 2 - 1) * b).reshape(out_channels, H, 1, 1)
    p["out.2.bias"] = (torch.rand(out_channels, generator=g) * 2 - 1) * b
    for l in range(n_layers):
        for m in "UVABC":
            lin(f"layers.{l}.{m}", H, H)
        norm(f"layers.{l}.norm_h", H)
        norm(f"layers.{l}.norm_e", H)
        lin(f"time_embed_layers.{l}.1", H, T2)
        norm(f"per_layer_out.{l}.0", H)
        lin(f"per_layer_out.{l}.2", H, H)
        if not randomize_zero_init:
            p[f"per_layer_out.{l}.2.weight"].zero_()
            p[f"per_layer_out.{l}.2.bias"].zero_()
    return p


def params_sha256(p: Params) -> str:
    """SHA-256 over the (sorted) names and fp32 bytes of a parameter dict: fixtures whose weights are regenerated
    from a seed (tests/golden/make_golden_h256.py) carry this hash so that an RNG change cannot go unnoticed."""
    import hashlib
    h = hashlib.sha256()
    for k in sorted(p):
        h.update(k.encode())
        h.update(np.ascontiguousarray(p[k].detach().cpu().numpy()).tobytes())
    return h.hexdigest()


def knn_graph(points: np.ndarray, k: int) -> np.ndarray:
    """Row-sorted constant-degree k-NN edge list including self as nearest neighbour, neighbours in
    distance order - the layout TSPGraphDataset produces (co_datasets/tsp_graph_dataset.py:53-62).
    Brute force in blocks (the reference uses sklearn KDTree; same result up to distance ties)."""
    n = points.shape[0]
    cols = np.empty((n, k), dtype=np.int64)
    blk = 2048
    for s in range(0, n, blk):
        d = ((points[s:s + blk, None, :] - points[None, :, :]) ** 2).sum(-1)
        idx = np.argpartition(d, k - 1, axis=1)[:, :k]
        dd = np.take_along_axis(d, idx, axis=1)
        order = np.argsort(dd, axis=1, kind="stable")
        cols[s:s + blk] = np.take_along_axis(idx, order, axis=1)
    rows = np.repeat(np.arange(n, dtype=np.int64), k)
    return np.stack([rows, cols.reshape(-1)], axis=0)


def tsp_instance(n: int, k: int, seed: int):
    """Uniform random points in the unit square (data/generate_tsp_data.py:44) + k-NN graph."""
    pts = np.random.default_rng(seed).random((n, 2))
    return pts.astype(np.float32), knn_graph(pts, k)


def er_mis_instance(n: int, prob: float, seed: int) -> np.ndarray:
    """Erdos-Renyi G(n,p) edges both directions + self loops, NOT row sorted
    (co_datasets/mis_dataset.py:43-48; data/mis-benchmark-framework/data_generation/random_graph.py:19-30)."""
    rng = np.random.default_rng(seed)
    iu = np.triu_indices(n, 1)
    keep = rng.random(iu[0].shape[0]) < prob
    und = np.stack([iu[0][keep], iu[1][keep]], axis=1).astype(np.int64)
    edges = np.concatenate([und, und[:, ::-1]], axis=0)
    loops = np.arange(n, dtype=np.int64).reshape(-1, 1).repeat(2, axis=1)
    return np.concatenate([edges, loops], axis=0).T.copy()
