"""Graph layout: COO ``edge_index`` (reference contract) -> CSR over the centre node, int32, on device.

Input contract: ``edge_index[0]`` = centre node i, ``edge_index[1]`` = neighbour
j; TSP k-NN graphs are row-sorted with constant degree and include the self edge
(``co_datasets/tsp_graph_dataset.py:53-62``); MIS graphs are undirected edges + reversed copy + self
loops, not row-sorted (``co_datasets/mis_dataset.py:43-48``); a batch is the disjoint union with node
ids offset per graph (``pl_meta_model.py:177-184``).  The conversion runs once per instance, outside
the denoising loop, on the host (C helper ``difusco_csr_from_coo_host``).
"""
import ctypes
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import _lib


@dataclass
class CsrGraph:
    n_nodes: int
    n_edges: int
    rowptr: torch.Tensor            # int32 [n_nodes+1], device
    col: torch.Tensor               # int32 [n_edges], device
    perm: Optional[torch.Tensor]    # int32 [n_edges] CSR slot -> caller edge id; None = identity
    row: Optional[torch.Tensor] = None       # int32 [n_edges] centre node of each CSR slot, device
    seg_ptr: Optional[torch.Tensor] = None   # int32 [S+1] device; GroupNorm statistic segments
    n_segments: int = 1
    # internal node id -> caller node id (int64, device) when the nodes were renumbered for locality (TSP: Morton order
    # of the coordinates inside every graph of the batch); None = caller numbering.  Only node-indexed INPUTS (points)
    # have to be gathered with it; outputs of a TSP step are per edge and go through ``perm``.
    node_order: Optional[torch.Tensor] = None


def csr_from_coo_host(edge_index: np.ndarray, n_nodes: int):
    """numpy int64 [2,E] -> (rowptr, col, row, perm, identity) int32 numpy arrays."""
    ei = np.ascontiguousarray(edge_index, dtype=np.int64)
    assert ei.ndim == 2 and ei.shape[0] == 2
    E = ei.shape[1]
    rowptr = np.empty(n_nodes + 1, dtype=np.int32)
    col = np.empty(E, dtype=np.int32)
    row = np.empty(E, dtype=np.int32)
    perm = np.empty(E, dtype=np.int32)
    ident = ctypes.c_int(0)
    _lib.check(_lib.lib().difusco_csr_from_coo_host(
        ei.ctypes.data, E, n_nodes, rowptr.ctypes.data, col.ctypes.data, row.ctypes.data, perm.ctypes.data,
        ctypes.byref(ident)))
    return rowptr, col, row, perm, bool(ident.value)


def _id_blocks(rowptr: np.ndarray, col: np.ndarray, n_nodes: int) -> np.ndarray:
    """Block id per node of the finest partition of 0..n-1 into CONTIGUOUS id ranges that no edge crosses - for a
    disjoint-union batch (``pl_meta_model.py:177-184``) these are the graphs of the batch (or unions of them)."""
    deg = np.diff(rowptr)
    ids = np.arange(n_nodes, dtype=np.int64)
    lo, hi = ids.copy(), ids.copy()
    nz = np.flatnonzero(deg > 0)
    if nz.size:
        starts = rowptr[nz].astype(np.int64)
        lo[nz] = np.minimum(lo[nz], np.minimum.reduceat(col, starts))
        hi[nz] = np.maximum(hi[nz], np.maximum.reduceat(col, starts))
    reach = np.maximum.accumulate(hi)                  # furthest id touched by the nodes 0..i
    back = np.minimum.accumulate(lo[::-1])[::-1]       # lowest id touched by the nodes i..n-1
    cut = np.zeros(n_nodes, dtype=np.int64)            # cut[b] = 1: a block starts at node b
    if n_nodes > 1:
        cut[1:] = (reach[:-1] < ids[1:]) & (back[1:] >= ids[1:])
    return np.cumsum(cut)


def _morton_keys(points: np.ndarray) -> np.ndarray:
    """Z-order key of 2-D points (16 bits per axis over the bounding box)."""
    p = np.asarray(points, dtype=np.float64).reshape(-1, 2)
    mn, mx = p.min(axis=0), p.max(axis=0)
    q = np.clip((p - mn) / np.maximum(mx - mn, 1e-30) * 65535.0, 0, 65535).astype(np.uint64)

    def spread(v):
        v = (v | (v << 8)) & np.uint64(0x00FF00FF)
        v = (v | (v << 4)) & np.uint64(0x0F0F0F0F)
        v = (v | (v << 2)) & np.uint64(0x33333333)
        return (v | (v << 1)) & np.uint64(0x55555555)

    return spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1))


def locality_node_order(rowptr: np.ndarray, col: np.ndarray, points: np.ndarray) -> np.ndarray:
    """new -> old node numbering: inside every graph of the batch (see :func:`_id_blocks`) the nodes are sorted along
    the Morton curve of their coordinates.  Spatially close centre nodes then sit in neighbouring CSR rows, and since
    k-NN neighbours are spatially close too, the rows A h[j], V h[j] gathered by consecutive 32-edge tiles are a small
    compact set (L2-resident per XCD with the XCD-contiguous tile ranges of the fused kernel)."""
    n = rowptr.shape[0] - 1
    return np.lexsort((_morton_keys(points), _id_blocks(rowptr, col, n))).astype(np.int64)


def build_csr(edge_index: torch.Tensor, n_nodes: int, device, seg_rows: Optional[np.ndarray] = None,
              points=None) -> CsrGraph:
    """edge_index int64 [2,E] (any device).  ``seg_rows``: boundaries [S+1] of the head-GroupNorm
    statistic segments over output rows (None = one segment = the reference's sparse behaviour).
    ``points`` ([n_nodes,2], optional): renumber the nodes for locality (``locality_node_order``); invisible to the
    caller - edge outputs keep the caller's edge order through ``perm``, ``node_order`` gathers the point input."""
    ei = edge_index.detach().cpu().numpy()
    rowptr, col, _row, perm, ident = csr_from_coo_host(ei, n_nodes)
    order = None
    if points is not None and n_nodes > 1 and col.shape[0] > 0:
        pts = points.detach().cpu().numpy() if isinstance(points, torch.Tensor) else np.asarray(points)
        order = locality_node_order(rowptr, col, pts.reshape(-1, 2)[:n_nodes])
        if np.array_equal(order, np.arange(n_nodes)):
            order = None
        else:
            inv = np.empty(n_nodes, dtype=np.int64)
            inv[order] = np.arange(n_nodes, dtype=np.int64)
            rowptr, col, _row, perm, ident = csr_from_coo_host(inv[np.ascontiguousarray(ei, dtype=np.int64)], n_nodes)
    g = CsrGraph(
        n_nodes=n_nodes, n_edges=int(col.shape[0]),
        rowptr=torch.from_numpy(rowptr).to(device), col=torch.from_numpy(col).to(device),
        perm=None if ident else torch.from_numpy(perm).to(device), row=torch.from_numpy(_row).to(device),
        node_order=None if order is None else torch.from_numpy(order).to(device))
    if seg_rows is not None and len(seg_rows) > 2:
        g.seg_ptr = torch.from_numpy(np.asarray(seg_rows, dtype=np.int32)).to(device)
        g.n_segments = len(seg_rows) - 1
    return g


def complete_graph_batch(batch: int, n: int, device) -> CsrGraph:
    """Dense mode (``gnn_encoder.py:350-381``): B graphs with all n*n ordered pairs, edge (b,i,j) at slot
    b*n*n + i*n + j - exactly the flattening of the reference's [B,V,V] tensors.  One GroupNorm
    statistic segment per sample (the dense head normalises a (B,H,V,V) tensor)."""
    rowptr = (torch.arange(batch * n + 1, dtype=torch.int64) * n).to(torch.int32)
    col = (torch.arange(n, dtype=torch.int32).repeat(batch * n)
           + torch.arange(batch, dtype=torch.int32).repeat_interleave(n * n) * n)
    row = torch.arange(batch * n, dtype=torch.int32).repeat_interleave(n)
    g = CsrGraph(n_nodes=batch * n, n_edges=batch * n * n, rowptr=rowptr.to(device), col=col.to(device), perm=None,
                 row=row.to(device))
    if batch > 1:
        g.seg_ptr = (torch.arange(batch + 1, dtype=torch.int64) * n * n).to(torch.int32).to(device)
        g.n_segments = batch
    return g


def edge_tiled_offsets(n_edges: int) -> torch.Tensor:
    """Flat offsets [E_pad, 256] (int64) of the tiled edge-feature layout the fused path keeps ``e`` in
    (``csrc/kernels.h: edge_tiled_offset``): rows padded to a multiple of 256 edges, tile = 32 edges,
    [slab f/16][(f/8)%2][((f/4)%2)*32 + s%32][f%4].  Test / debugging helper."""
    e_pad = (n_edges + 255) // 256 * 256
    s = torch.arange(e_pad, dtype=torch.int64)[:, None]
    f = torch.arange(256, dtype=torch.int64)[None, :]
    return (s >> 5) * 8192 + (f >> 4) * 512 + ((f >> 3) & 1) * 256 + ((((f >> 2) & 1) * 32) + (s & 31)) * 4 + (f & 3)


def to_tiled(e: torch.Tensor) -> torch.Tensor:
    """[E, 256] row-major -> flat tiled buffer of E_pad*256 floats (pad rows zero)."""
    off = edge_tiled_offsets(e.shape[0]).to(e.device)
    out = torch.zeros(off.shape[0] * 256, dtype=e.dtype, device=e.device)
    out[off[: e.shape[0]].reshape(-1)] = e.reshape(-1)
    return out


def from_tiled(buf: torch.Tensor, n_edges: int) -> torch.Tensor:
    off = edge_tiled_offsets(n_edges).to(buf.device)
    return buf[off[:n_edges].reshape(-1)].reshape(n_edges, 256)


def knn_edge_index_gpu(points, k: int, device="cuda:0", graphs: int = 1) -> torch.Tensor:
    """``edge_index`` int64 [2, G*n*k] on ``device`` in the reference's layout (``co_datasets/tsp_graph_dataset.py:
    53-62``; batch = disjoint union with node ids offset by g*n, ``pl_meta_model.py:177-184``), built by
    ``difusco_knn_graph``.  ``points``: float64 [G*n, 2] (numpy or tensor), G graphs of n points each."""
    import ctypes
    L = _lib.lib()
    device = torch.device(device)
    if isinstance(points, np.ndarray):
        points = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float64))
    pts = points.to(device=device, dtype=torch.float64).contiguous()
    n = pts.shape[0] // graphs
    if n * graphs != pts.shape[0]:
        raise ValueError("points must hold `graphs` instances of equal size")
    ei = torch.empty((2, graphs * n * k), dtype=torch.int64, device=device)
    nbytes = ctypes.c_size_t()
    _lib.check(L.difusco_knn_graph_workspace_bytes(n, k, ctypes.byref(nbytes)))
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=device)
    stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    for g in range(graphs):
        _lib.check(L.difusco_knn_graph(n, k, ctypes.c_void_p(pts[g * n:].data_ptr()), g * n,
                                       ctypes.c_void_p(ei[0, g * n * k:].data_ptr()),
                                       ctypes.c_void_p(ei[1, g * n * k:].data_ptr()),
                                       ctypes.c_void_p(ws.data_ptr()), nbytes.value, stream))
    return ei
