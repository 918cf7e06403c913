"""CPU restatement of the reference's heatmap -> tour decode.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py):
imported by tests/ and by the cpu_baseline leg of scripts/bench_decode.py, never by the product path.

Follows ``difusco/utils/tsp_utils.py:89-145`` (``merge_tours``: A + A^T densification in float32, walk from node 0)
and ``difusco/utils/cython_merge/cython_merge.pyx:19-120`` (``merge_cython``: float64 distance matrix, argsort of
-A/d over all N^2 entries, route_begin/route_end bookkeeping with path compression).  Parity pin: the fixtures
``tests/golden/tsp_decode_*.npz`` were produced by the reference's own functions (the .pyx compiled with the
Cython of this image), see tests/golden/make_golden_decode.py and PROVENANCE.md.

One deliberate difference: numpy's default argsort is unstable, so the reference's order among EXACTLY equal
scores (the zero entries of non-edges, the two orientations of a pair) is an implementation accident.  This
restatement sorts stably (ties in flat-index order); the tour is invariant under the orientation order, and the
fixtures only pin cases that finish before the zero entries (``completed``)."""
import numpy as np


def _find(link, i):
    r = i
    while link[r] != r:
        r = link[r]
    while link[i] != r:
        link[i], i = r, link[i]
    return r


def merge_dense(points, adj_mat):
    """cython_merge.pyx:19-104 on a dense symmetric matrix.  Returns (A_tour [N,N] 0/1, merge_iterations,
    completed) where completed says that the N-1 insertions happened on entries with a negative sort key
    (positive heat / distance), i.e. before the zero block."""
    pts = np.asarray(points, dtype=np.float64)
    adj = np.asarray(adj_mat, dtype=np.float64)
    n = pts.shape[0]
    dist = np.linalg.norm(pts[:, None] - pts, axis=-1)
    with np.errstate(divide="ignore", invalid="ignore"):
        key = (-adj / dist).flatten()
    order = np.argsort(key, kind="stable")
    begin, end = np.arange(n), np.arange(n)
    A = np.zeros((n, n))
    it = cnt = 0
    completed = True
    for edge in order:
        it += 1
        i, j = int(edge // n), int(edge % n)
        bi, ei_, bj, ej = _find(begin, i), _find(end, i), _find(begin, j), _find(end, j)
        if bi == bj or (i != bi and i != ei_) or (j != bj and j != ej):
            continue
        if not key[edge] < 0:
            completed = False
        A[i, j] = A[j, i] = 1
        cnt += 1
        if i == bi and j == ej:
            begin[bi] = bj
            end[ej] = ei_
        elif i == ei_ and j == bj:
            begin[bj] = bi
            end[ei_] = ej
        elif i == bi and j == bj:
            begin[bi] = ej
            begin[bj] = ej
            begin[ej] = ej
            end[ej] = ei_
            end[bj] = ei_
        else:
            end[ei_] = bj
            begin[bj] = bi
            begin[ej] = bi
            end[ej] = bj
            end[bj] = bj
        if cnt == n - 1:
            break
    fb, fe = _find(begin, 0), _find(end, 0)
    A[fe, fb] = A[fb, fe] = 1
    return A, it, completed


def merge_tours(adj_mat, np_points, edge_index_np, sparse_graph=True, parallel_sampling=1):
    """tsp_utils.py:89-145 (dense branch :105-108, sparse branch :109-116, tour walk :131-141).  Returns (tours, mean merge_iterations,
    completed flags)."""
    n = np_points.shape[0]
    if sparse_graph:
        parts = np.split(np.asarray(adj_mat, dtype=np.float32).reshape(-1), parallel_sampling, axis=0)
    else:                                                # :105-108: [parallel_sampling, N, N], adj_mat[0] + adj_mat[0].T
        parts = [p[0] for p in np.split(np.asarray(adj_mat, dtype=np.float32).reshape(-1, n, n), parallel_sampling, axis=0)]
    tours, iters, done = [], [], []
    for part in parts:
        if sparse_graph:
            a = np.zeros((n, n), dtype=np.float32)
            np.add.at(a, (edge_index_np[0], edge_index_np[1]), part)
        else:
            a = part
        dense = a + a.T                                  # float32 add, as scipy's toarray() + toarray()
        real, it, ok = merge_dense(np_points, dense)
        tour = [0]
        while len(tour) < n + 1:
            nb = np.nonzero(real[tour[-1]])[0]
            if len(tour) > 1:
                nb = nb[nb != tour[-2]]
            tour.append(int(nb.max()))
        tours.append(tour)
        iters.append(it)
        done.append(ok)
    return tours, float(np.mean(iters)), done


def batched_two_opt(points, tour, max_iterations=1000):
    """tsp_utils.py:12-49 (``batched_two_opt_torch``) in numpy float64: the four N x N distance matrices, their
    combination in the reference's order, triu(diagonal=2), one global min over the batch for the stopping test,
    per-tour argmin (first occurrence on the flattened matrix) and segment reversal."""
    pts = np.asarray(points, dtype=np.float64)
    tour = np.array(tour, dtype=np.int64, copy=True)
    n = len(pts)
    iterator = 0
    min_change = -1.0
    while min_change < 0.0:
        pi = pts[tour[:, :-1]]                                   # [B, N, 2]
        pi1 = pts[tour[:, 1:]]

        def dmat(a, b):
            d = a[:, :, None, :] - b[:, None, :, :]
            return np.sqrt((d ** 2).sum(-1))
        a_ij, a_i1j1 = dmat(pi, pi), dmat(pi1, pi1)
        d_i = np.sqrt(((pi - pi1) ** 2).sum(-1))                 # [B, N]
        change = a_ij + a_i1j1 - d_i[:, :, None] - d_i[:, None, :]
        valid = np.triu(change, k=2)
        min_change = valid.min()
        flat = valid.reshape(len(tour), -1).argmin(-1)
        mi, mj = flat // n, flat % n
        if min_change < -1e-6:
            for b in range(len(tour)):
                tour[b, mi[b] + 1:mj[b] + 1] = tour[b, mi[b] + 1:mj[b] + 1][::-1].copy()
            iterator += 1
        else:
            break
        if iterator >= max_iterations:
            break
    return tour, iterator


def mis_decode(predictions, edge_index, n_nodes):
    """mis_utils.py:3-18 (``mis_decode_np``) with the adjacency given as the edge_index it is built from
    (pl_mis_model.py:150-154); nodes are visited in decreasing score, equal scores by increasing id (the reference's
    argsort is unstable there)."""
    pred = np.asarray(predictions).reshape(-1)
    nbrs = [[] for _ in range(n_nodes)]
    for a, b in zip(edge_index[0].tolist(), edge_index[1].tolist()):
        nbrs[a].append(b)
    solution = np.zeros(n_nodes, dtype=int)
    for i in np.argsort(-pred, kind="stable"):
        if solution[i] == -1:
            continue
        solution[nbrs[i]] = -1
        solution[i] = 1
    return (solution == 1).astype(int)
