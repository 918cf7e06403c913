"""On-disk heatmap formats of the reference.

* ``save_numpy_heatmap``  - the ``.npy`` pair written by ``TSPModel.run_save_numpy_heatmap``
  (``difusco/pl_tsp_model.py:258-267``): ``{split}-heatmap-{idx}.npy`` and ``{split}-points-{idx}.npy`` under
  ``<dir>/numpy_heatmap``.
* ``write_mcts_heatmap`` / ``mcts_heatmap_rows`` - the text file the C++ MCTS tool reads
  (``tsp_mcts/convert_numpy_to_txt.py:18-72``): first line N, then N rows of N ``%.6f`` numbers.

What the converter computes, per instance (all in the dtype of its inputs, float32 for the model's heatmaps):

    v[i][j] = heat[i][j] + 0.01 * (1 - |p_i - p_j|)                  every ordered pair (the distance prior is dense)
    keep    = { v > T } U { the 3 largest entries of every row },     T = the K-th largest positive v, K = int(N*N*prob)
    m       = (v on keep, 0 elsewhere), + 0.01 on its non-zeros;  out = (m + m^T) / rowsum(m + m^T)

The reference materialises five N x N arrays for this (800 MB each in float64 at N = 10^4).  Here the
heatmap stays SPARSE - the E entries of the k-NN models' output - and nothing N x N is ever allocated:

The distance prior makes every one of the N^2 entries a candidate (at N = 10^4 with the default 2 % the threshold lies
among entries that carry the prior alone), and the output is N^2 numbers by definition - so the work is O(N^2) either
way.  What need not be O(N^2) is the memory: ``v`` is recomputed for ``block_rows`` rows at a time with the converter's
own numpy expression, and three sweeps over the block rows do everything:

1. histogram of the positive values by the 16 high bits of their float32 pattern + the row top-3;
2. the one histogram bucket that contains T is collected and sorted: T exactly, no N^2 sort;
3. per block: the masked block, the masked TRANSPOSE block (the distance is symmetric, the transposed heat entries and
   the transposed top-3 memberships are scattered into it), their sum, numpy's own row sums, the division - the same
   operations on the same values as the reference's whole-matrix statements - then vectorised ``%.6f`` formatting
   (including the reference's ``-0.000000`` where both orientations carry a negative prior) streamed to the file.

Memory: O(E + block_rows * N) (about 2 M entries per buffer by default).

Two implementations of the numeric part, the same numbers bit for bit:

* ``mcts_heatmap_rows_gpu`` (default of ``write_mcts_heatmap`` whenever a GPU is present and the inputs are float32 - what the
  models produce): ``csrc/formats.hip`` through the C ABI (``difusco_mcts_heatmap_prepare / _rows``); the E entries never leave
  the device, the rows come back one block at a time for formatting.  N = 10^4: ~0.1 s instead of ~30 s.
* ``mcts_heatmap_rows`` (host numpy; float64 inputs, hosts without a GPU): the block-row sweeps described above.

The output is identical, character for character, to the reference converter's on the committed fixtures
(``tests/golden/mcts_text_n30.npz``, ``mcts_text_n64.npz``, ``mcts_sparse_text_n1000_k50.npz``, produced by importing
``tsp_mcts/convert_numpy_to_txt.py``)."""
import os
from typing import Iterator, Optional, Tuple

import numpy as np


def densify(heat, edge_index, n_nodes: int) -> np.ndarray:
    """[E] heat values on the directed edges of ``edge_index`` -> dense [N,N] float32 (entries off the graph are 0).
    Small instances / tests only - the MCTS text path below never needs it."""
    a = np.zeros((n_nodes, n_nodes), dtype=np.float32)
    ei = np.asarray(edge_index)
    a[ei[0], ei[1]] = np.asarray(heat, dtype=np.float32)
    return a


def sparsify(dense: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Dense [N,N] heatmap -> (heat [E], edge_index [2,E]) of its non-zero entries, row-major order."""
    r, c = np.nonzero(dense)
    return dense[r, c], np.stack([r, c]).astype(np.int64)


def save_numpy_heatmap(adj_mat, np_points, save_dir: str, real_batch_idx: int, split: str = "test"):
    """pl_tsp_model.py:258-267.  Returns the two paths."""
    heatmap_path = os.path.join(save_dir, "numpy_heatmap")
    os.makedirs(heatmap_path, exist_ok=True)
    hp = os.path.join(heatmap_path, f"{split}-heatmap-{real_batch_idx}.npy")
    pp = os.path.join(heatmap_path, f"{split}-points-{real_batch_idx}.npy")
    np.save(hp, np.asarray(adj_mat))
    np.save(pp, np.asarray(np_points))
    return hp, pp


# ------------------------------------------------------------------------------------------------
# the converter's arithmetic on gathered operands (the only place the reference's formula appears)
# ------------------------------------------------------------------------------------------------
def _prior_values(heat, pts, rows, cols):
    """v = heat + 0.01 * (1 - dist) for the entries (rows[k], cols[k]); same dtype promotion as the reference's
    ``adj_matrix + 0.01 * (1.0 - dists)`` (convert_numpy_to_txt.py:21-22), inf -> 0 (:24)."""
    d = np.linalg.norm(pts[rows] - pts[cols], axis=-1)
    v = heat + 0.01 * (1.0 - d)
    v[v == np.inf] = 0.0
    return v


def _block_values(heat_rows, pts, lo, hi):
    """The dense values of rows lo..hi-1: [hi-lo, N]; ``heat_rows`` holds the scattered heat of those rows."""
    d = np.linalg.norm(pts[lo:hi, None, :] - pts[None, :, :], axis=-1)
    v = heat_rows + 0.01 * (1.0 - d)
    v[v == np.inf] = 0.0
    return v


class _SparseHeat:
    """The E heat entries in CSR order (sorted by row, then column) with duplicate-free (row, col) pairs."""

    def __init__(self, heat, edge_index, n):
        ei = np.asarray(edge_index, dtype=np.int64)
        heat = np.asarray(heat)
        if heat.dtype not in (np.float32, np.float64):
            heat = heat.astype(np.float32)
        order = np.lexsort((ei[1], ei[0]))
        self.row, self.col, self.heat = ei[0][order], ei[1][order], heat.reshape(-1)[order]
        key = self.row * n + self.col
        if key.size > 1 and np.any(key[1:] == key[:-1]):
            raise ValueError("duplicate (row, col) entries in the sparse heatmap")
        self.n = n
        self.rowptr = np.searchsorted(self.row, np.arange(n + 1))

    def block(self, lo, hi, dtype):
        a = np.zeros((hi - lo, self.n), dtype=dtype)
        s, e = self.rowptr[lo], self.rowptr[hi]
        a[self.row[s:e] - lo, self.col[s:e]] = self.heat[s:e]
        return a


def _threshold_and_top3(sp: _SparseHeat, pts, k: int, block_rows: int, dtype):
    """(T, top3): T = the k-th largest positive entry of the dense value matrix (``np.sort(valid_values)[-k]``,
    convert_numpy_to_txt.py:27-29), top3 [N,3] = the last three columns of every row's argsort (:33).  Two block-row
    sweeps: the first histograms the positive values by the 16 high bits of their float32 pattern (positive floats
    order like their bit patterns) and takes the row top-3, the second collects the one bucket that holds T."""
    n = sp.n

    def blocks():
        for lo in range(0, n, block_rows):
            hi = min(n, lo + block_rows)
            yield lo, hi, _block_values(sp.block(lo, hi, dtype), pts, lo, hi)
    top3 = np.empty((n, 3), dtype=np.int64)
    if dtype != np.float32:                       # float64 heatmaps: keep every positive value (small inputs only)
        parts = []
        for lo, hi, v in blocks():
            top3[lo:hi] = np.argsort(v, axis=1)[:, -3:]
            parts.append(v[v > 0.0])
        return np.sort(np.concatenate(parts))[-k], top3       # (k = 0: index -0 = the smallest value, like the reference)
    hist = np.zeros(1 << 16, dtype=np.int64)
    for lo, hi, v in blocks():
        top3[lo:hi] = np.argsort(v, axis=1)[:, -3:]
        hist += np.bincount(v[v > 0.0].view(np.uint32) >> 16, minlength=1 << 16)
    if hist.sum() == 0:
        raise IndexError("no positive entry")            # the reference indexes an empty valid_values array here (k = 0 too)
    if k > hist.sum():
        raise IndexError("fewer positive entries than expected_valid_value_num")     # the reference fails here as well
    if k == 0:                                    # valid_values[-0] is valid_values[0]: the SMALLEST positive value
        k = int(hist.sum())
    above = np.cumsum(hist[::-1])[::-1]                                   # positive entries in buckets >= b
    bucket = int(np.nonzero(above >= k)[0].max())
    rank = k - (int(above[bucket + 1]) if bucket + 1 < (1 << 16) else 0)  # T is the rank-th largest inside the bucket
    cand = []
    for lo, hi, v in blocks():
        pos = v[v > 0.0]
        cand.append(pos[(pos.view(np.uint32) >> 16) == bucket])
    return np.sort(np.concatenate(cand))[-rank], top3


def mcts_heatmap_rows(heat, edge_index, points, num_nodes: int, expected_valid_prob: float = 0.02,
                      block_rows: Optional[int] = None) -> Iterator[np.ndarray]:
    """Yields the normalised rows ([N] arrays, dtype of the inputs) of the MCTS heatmap for the SPARSE heatmap
    ``heat`` [E] on the directed edges ``edge_index`` [2,E]; ``points`` [N,2].  See the module docstring."""
    n = int(num_nodes)
    pts = np.asarray(points)
    heat = np.asarray(heat)
    dtype = np.result_type(heat.dtype if heat.dtype.kind == "f" else np.float32, pts.dtype)
    pts = pts.astype(dtype, copy=False)
    if block_rows is None:
        block_rows = max(1, min(n, (1 << 21) // max(n, 1)))                 # ~2 M entries per block-row buffer
    sp = _SparseHeat(heat.astype(dtype, copy=False), edge_index, n)
    spt = _SparseHeat(sp.heat, np.stack([sp.col, sp.row]), n)               # the transposed entries: heat[j][i] at (i, j)
    k = int(n * n * expected_valid_prob)                                    # convert_numpy_to_txt.py:26
    thr, top3 = _threshold_and_top3(sp, pts, k, block_rows, dtype)
    t3_rows = np.repeat(np.arange(n, dtype=np.int64), 3)
    t3_cols = top3.reshape(-1)
    by_col = np.argsort(t3_cols, kind="stable")                             # top-3 membership, indexed by column
    t3c_sorted, t3r_sorted = t3_cols[by_col], t3_rows[by_col]
    bump = np.asarray(1e-2, dtype=dtype)

    def masked(v, keep):
        m = v * keep                                                        # :44  (a negative value times False is -0.0)
        m[m != 0.0] += bump                                                 # :45
        return m

    for lo in range(0, n, block_rows):
        hi = min(n, lo + block_rows)
        v = _block_values(sp.block(lo, hi, dtype), pts, lo, hi)             # v[i][j], i in the block
        keep = v > thr                                                      # :35
        keep[np.arange(lo, hi)[:, None] - lo, top3[lo:hi]] = True           # :41-43
        vt = _block_values(spt.block(lo, hi, dtype), pts, lo, hi)           # v[j][i] at [i][j]: the distance is symmetric
        keept = vt > thr
        s, e = np.searchsorted(t3c_sorted, lo), np.searchsorted(t3c_sorted, hi)
        keept[t3c_sorted[s:e] - lo, t3r_sorted[s:e]] = True                 # (j, i) with i among the top 3 of row j
        out = masked(v, keep) + masked(vt, keept)                           # :46
        out = out / out.sum(axis=1, keepdims=True)                          # :47, numpy's own row sum
        for r in range(hi - lo):
            yield out[r]


def mcts_heatmap_rows_gpu(heat, edge_index, points, num_nodes: int, expected_valid_prob: float = 0.02,
                          block_rows: Optional[int] = None, device=None) -> Iterator[np.ndarray]:
    """The same rows as ``mcts_heatmap_rows``, computed on the GPU (csrc/formats.hip; float32 only).  ``heat`` [E],
    ``edge_index`` [2,E], ``points`` [N,2]: torch tensors (any device) or numpy arrays.  Yields float32 numpy rows."""
    import ctypes
    import torch
    from . import _lib
    n = int(num_nodes)
    if device is not None:
        dev = torch.device(device)
    elif isinstance(heat, torch.Tensor) and heat.is_cuda:
        dev = heat.device
    else:
        dev = torch.device("cuda", torch.cuda.current_device())
    if dev.type != "cuda":
        raise _lib.DifuscoHipError("mcts_heatmap_rows_gpu needs a GPU device")
    as_t = lambda x, dt: torch.as_tensor(np.asarray(x) if not hasattr(x, "detach") else x).to(dev, dtype=dt).contiguous()
    heat_d = as_t(heat, torch.float32).reshape(-1)
    ei = as_t(edge_index, torch.int32)
    row_d, col_d = ei[0].contiguous(), ei[1].contiguous()
    pts_d = as_t(points, torch.float32).reshape(n, 2).contiguous()
    E = int(heat_d.numel())
    L = _lib.lib()
    nbytes = ctypes.c_size_t()
    _lib.check(L.difusco_mcts_heatmap_workspace_bytes(n, E, ctypes.byref(nbytes)))
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    with torch.cuda.device(dev):
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        thr = ctypes.c_float()
        rc = L.difusco_mcts_heatmap_prepare(n, E, P(row_d), P(col_d), P(heat_d), P(pts_d), float(expected_valid_prob), P(ws),
                                            ws.numel(), ctypes.byref(thr), st)
        if rc < 0:
            msg = L.difusco_last_error().decode()
            if "IndexError" in msg:
                raise IndexError(msg)
            if "duplicate" in msg:
                raise ValueError(msg)
            _lib.check(rc)
        if block_rows is None:
            block_rows = max(1, min(n, (1 << 23) // max(n, 1)))             # ~8 M entries (32 MB) per device block
        out = torch.empty((min(block_rows, n), n), dtype=torch.float32, device=dev)
        for lo in range(0, n, block_rows):
            cnt = min(block_rows, n - lo)
            _lib.check(L.difusco_mcts_heatmap_rows(n, E, P(pts_d), P(ws), ws.numel(), lo, cnt, P(out), st))
            host = out[:cnt].cpu().numpy()
            for r in range(cnt):
                yield host[r]


def _format_rows(rows: Iterator[np.ndarray], n: int) -> Iterator[bytes]:
    """Rows -> text lines of ``%.6f`` numbers.  Vectorised: every slot is [sign or nothing][8 characters][space]; zeros
    (and the reference's ``-0.000000`` where both orientations of a pair carry a negative prior) come from a template,
    only the non-zeros are formatted."""
    slot = np.frombuffer(b"\x000.000000 ", dtype=np.uint8)
    minus = ord("-")
    for row in rows:
        tok = np.tile(slot, (n, 1))
        tok[np.signbit(row), 0] = minus
        wide = False
        for j in np.nonzero(row)[0]:
            txt = f"{abs(float(row[j])):.6f}".encode()
            if len(txt) != 8:                     # >= 10: not produced by a normalised row; fall back to plain formatting
                wide = True
                break
            tok[j, 1:9] = np.frombuffer(txt, dtype=np.uint8)
        if wide:
            yield (" ".join(f"{x:.6f}" for x in row) + "\n").encode()
            continue
        flat = tok.reshape(-1)
        yield flat[flat != 0].tobytes()[:-1] + b"\n"


def mcts_heatmap_text(adj_matrix: np.ndarray, points: np.ndarray, num_nodes: int, expected_valid_prob: float = 0.02) -> str:
    """The converter's text for a DENSE [N,N] heatmap (small instances, the reference's input form): its non-zero
    entries go through the sparse path above."""
    dense = np.asarray(adj_matrix)
    heat, ei = sparsify(dense)
    rows = mcts_heatmap_rows(heat.astype(dense.dtype if dense.dtype.kind == "f" else np.float32), ei, points, num_nodes,
                             expected_valid_prob)
    return f"{num_nodes}\n" + b"".join(_format_rows(rows, num_nodes)).decode()


def _gpu_available() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def write_mcts_heatmap(heat=None, points=None, num_nodes: int = None, output_dir: str = None, index: int = 0,
                       heatmap_prefix: str = "heatmap", expected_valid_prob: float = 0.02, edge_index=None,
                       block_rows: Optional[int] = None, use_gpu: Optional[bool] = None, adj_matrix=None) -> str:
    """File name and directory layout of convert_numpy_to_txt.py:60-66; rows are streamed to the file.  ``heat`` (alias
    ``adj_matrix``, the reference's name for the dense input) is either a dense [N,N] array (``edge_index`` None) or the
    sparse [E] heatmap on ``edge_index`` [2,E] - numpy arrays or torch tensors.

    ``use_gpu``: None / True = the GPU kernels (csrc/formats.hip), which compute in float32 like the reference's own data
    flow (``pl_tsp_model.py:258-267`` dumps float32 heatmaps and float32 coordinates): inputs of any other dtype are cast to
    float32 ON THE DEVICE, and a host without a GPU raises - nothing drops to the host silently (at N = 10^4 the host
    sweeps take 31 s against 2.8 ms).  ``use_gpu=False``, explicitly: the host numpy sweeps in the dtype of the inputs
    (float64 arithmetic for float64 inputs; GPU-less hosts and the CPU test-suite use this); same text for float32 inputs."""
    if heat is None:
        heat = adj_matrix
    if heat is None or points is None or num_nodes is None or output_dir is None:
        raise TypeError("write_mcts_heatmap needs heat (or adj_matrix), points, num_nodes and output_dir")

    def host(x):
        return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)

    if use_gpu is None:
        use_gpu = True
    if use_gpu and not _gpu_available():
        from . import _lib
        raise _lib.DifuscoHipError("write_mcts_heatmap: no GPU is visible; pass use_gpu=False for the host numpy program")
    folder = f"{output_dir}/{heatmap_prefix}/tsp{num_nodes}"
    os.makedirs(folder, exist_ok=True)
    path = f"{folder}/heatmaptsp{num_nodes}_{index}.txt"
    if use_gpu:
        if edge_index is None:
            dense = host(heat)
            heat, edge_index = sparsify(dense.astype(np.float32, copy=False))
        rows = mcts_heatmap_rows_gpu(heat, edge_index, points, num_nodes, expected_valid_prob, block_rows)
    else:
        heat, points = host(heat), host(points)
        if edge_index is None:
            heat, edge_index = sparsify(heat)
        else:
            edge_index = host(edge_index)
        rows = mcts_heatmap_rows(heat, edge_index, points, num_nodes, expected_valid_prob, block_rows or 256)
    with open(path, "wb") as f:
        f.write(f"{num_nodes}\n".encode())
        for line in _format_rows(rows, num_nodes):
            f.write(line)
    return path
