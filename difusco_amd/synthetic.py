"""Synthetic inputs for benchmarks and smoke runs (there is no network for datasets or checkpoints):
random-Euclidean TSP instances with the reference's k-NN graph layout, Erdos-Renyi MIS graphs, and
random weights with the reference architecture's shapes and PyTorch-default init scale."""
import math

import numpy as np
import torch


def knn_edge_index(points: np.ndarray, k: int) -> np.ndarray:
    """[2, N*k] int64: edge_index[0] = i repeated k times, edge_index[1] = the k nearest neighbours of i
    in distance order, self first (``co_datasets/tsp_graph_dataset.py:53-62``).  Blocked brute force."""
    n = points.shape[0]
    cols = np.empty((n, k), dtype=np.int64)
    sq = (points ** 2).sum(1)
    for s in range(0, n, 2048):
        blk = points[s:s + 2048]
        d = ((blk[:, None, :] - points[None, :, :]) ** 2).sum(-1)
        idx = np.argpartition(d, k - 1, axis=1)[:, :k]
        order = np.argsort(np.take_along_axis(d, idx, axis=1), axis=1, kind="stable")
        cols[s:s + 2048] = np.take_along_axis(idx, order, axis=1)
    del sq
    return np.stack([np.repeat(np.arange(n, dtype=np.int64), k), cols.reshape(-1)], axis=0)


def tsp_instance(n: int, k: int, seed: int):
    """Uniform points in the unit square (``data/generate_tsp_data.py:44``) and their k-NN graph."""
    pts = np.random.default_rng(seed).random((n, 2))
    return pts.astype(np.float32), knn_edge_index(pts, k)


def tsp_batch(n: int, k: int, graph_ids, device=None):
    """Disjoint union (node ids offset per graph, ``pl_meta_model.py:177-184``) of independent instances."""
    pts, eis = [], []
    for slot, gid in enumerate(graph_ids):
        p, ei = tsp_instance(n, k, seed=1000 + int(gid))
        pts.append(p)
        eis.append(ei + slot * n)
    points, edge_index = torch.from_numpy(np.concatenate(pts, 0)), torch.from_numpy(np.concatenate(eis, 1))
    return (points.to(device), edge_index.to(device)) if device is not None else (points, edge_index)


def tsp_batch_gpu(n: int, k: int, graph_ids, device):
    """Same instances as :func:`tsp_batch` (same seeds, same float32 coordinates), with the k-NN graphs built on the
    GPU by ``difusco_knn_graph`` instead of the numpy brute force (identical edge_index, checked by the GPU tests)."""
    from .graph import knn_edge_index_gpu
    graph_ids = list(graph_ids)
    pts64 = np.concatenate([np.random.default_rng(1000 + int(g)).random((n, 2)) for g in graph_ids], 0)
    edge_index = knn_edge_index_gpu(pts64, k, device=device, graphs=len(graph_ids))
    return torch.from_numpy(pts64.astype(np.float32)).to(device), edge_index


def er_mis_edge_index(n: int, prob: float, seed: int) -> np.ndarray:
    """G(n,p) undirected edges + reversed copies + self loops, not row-sorted
    (``co_datasets/mis_dataset.py:43-48``)."""
    rng = np.random.default_rng(seed)
    iu = np.triu_indices(n, 1)
    keep = rng.random(iu[0].shape[0]) < prob
    und = np.stack([iu[0][keep], iu[1][keep]], axis=1).astype(np.int64)
    loops = np.arange(n, dtype=np.int64).reshape(-1, 1).repeat(2, axis=1)
    return np.concatenate([und, und[:, ::-1], loops], axis=0).T.copy()


def random_state_dict(hidden: int, n_layers: int, out_channels: int, seed: int):
    """Reference ``GNNEncoder`` key set (``gnn_encoder.py:303-347``) with nn.Linear-default scale
    U(-1/sqrt(fan_in), 1/sqrt(fan_in)); norm affines perturbed around (1, 0).  ``per_layer_out.*.2`` is
    zero-initialised upstream, which makes all layers invisible in the edge output, so it
    is drawn like every other linear here."""
    g = torch.Generator().manual_seed(seed)
    H, T2 = hidden, hidden // 2
    sd = {}

    def lin(name, fo, fi):
        b = 1.0 / math.sqrt(fi)
        sd[name + ".weight"] = (torch.rand(fo, fi, generator=g) * 2 - 1) * b
        sd[name + ".bias"] = (torch.rand(fo, generator=g) * 2 - 1) * b

    def norm(name, n):
        sd[name + ".weight"] = 1.0 + 0.1 * torch.randn(n, generator=g)
        sd[name + ".bias"] = 0.1 * torch.randn(n, generator=g)

    lin("node_embed", H, H)
    lin("edge_embed", H, H)
    lin("time_embed.0", T2, H)
    lin("time_embed.2", T2, T2)
    norm("out.0", H)
    b = 1.0 / math.sqrt(H)
    sd["out.2.weight"] = ((torch.rand(out_channels, H, generator=g) * 2 - 1) * b).reshape(out_channels, H, 1, 1)
    sd["out.2.bias"] = (torch.rand(out_channels, generator=g) * 2 - 1) * b
    for l in range(n_layers):
        for m in "UVABC":
            lin(f"layers.{l}.{m}", H, H)
        norm(f"layers.{l}.norm_h", H)
        norm(f"layers.{l}.norm_e", H)
        lin(f"time_embed_layers.{l}.1", H, T2)
        norm(f"per_layer_out.{l}.0", H)
        lin(f"per_layer_out.{l}.2", H, H)
    return sd
