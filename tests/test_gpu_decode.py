"""Heatmap -> tour through the C ABI (difusco_tsp_merge_tour) on a real GPU: against the reference's fixtures,
against the CPU oracle on seeded synthetic heatmaps, and size-independent properties at TSP-10000."""
import os

import numpy as np
import pytest
import torch

from oracle import tsp_decode_oracle as D
from test_decode_oracle import GOLDEN, check_against_fixture

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[11:-4] for p in GOLDEN])
def test_merge_tours_matches_reference_fixture(dev, path):
    from difusco_amd.decode import merge_tours
    z = np.load(path)
    par = int(z["parallel_sampling"])
    tours, it_mean, done = merge_tours(z["heat"], z["points"], z["edge_index"], sparse_graph=True, parallel_sampling=par,
                                       device=dev, return_completed=True)
    iters = [merge_tours(part, z["points"], z["edge_index"], sparse_graph=True, device=dev)[1]
             for part in np.split(z["heat"], par)]
    check_against_fixture(z, tours, iters, done)
    if z["completed"].all():
        assert it_mean == float(z["merge_iterations"])


def test_dense_merge_matches_reference_fixture(dev):
    """The dense branch of merge_tours (tsp_utils.py:105-108): [parallel_sampling, N, N] heatmaps of the dense TSP-50
    models (BASELINE configs[0]) against the reference's own output."""
    from difusco_amd.decode import merge_tours
    from test_decode_oracle import DENSE
    assert len(DENSE) >= 3
    for path in DENSE:
        z = np.load(path)
        par = int(z["parallel_sampling"])
        tours, it, done = merge_tours(z["heat"], z["points"], None, sparse_graph=False, parallel_sampling=par, device=dev,
                                      return_completed=True)
        assert all(done) and np.array_equal(np.asarray(tours), z["tours"]) and it == float(z["merge_iterations"])


def _heat(kind, pts, ei, rng):
    d = np.linalg.norm(pts[ei[0]] - pts[ei[1]], axis=1)
    if kind == "bits":
        return ((rng.random(ei.shape[1]) < np.exp(-d / (0.6 * d.mean()))).astype(np.float32) + np.float32(1e-6))
    if kind == "prob":
        return (np.exp(-d / (0.5 * d.mean())) * rng.random(ei.shape[1])).astype(np.float32) + np.float32(1e-6)
    return (rng.standard_normal(ei.shape[1]).astype(np.float32) * np.float32(0.25) + np.float32(0.75))


@pytest.mark.parametrize("n,k,kind,shuffle", [(30, 29, "bits", False), (64, 63, "prob", True), (97, 96, "gauss", True),
                                              (150, 15, "bits", True), (400, 40, "prob", False), (257, 31, "gauss", True),
                                              (500, 50, "bits", False)])
def test_merge_tours_matches_oracle(dev, n, k, kind, shuffle):
    """Exact tour equality with the CPU oracle in BOTH regimes (the GPU path walks the zero block in the same stable
    order as the oracle), with the edge list in arbitrary order."""
    from difusco_amd.decode import merge_tours
    from difusco_amd.synthetic import tsp_instance
    rng = np.random.default_rng(n * 1000 + k)
    pts, ei = tsp_instance(n, k, seed=n)
    heat = _heat(kind, pts, ei, rng)
    if shuffle:
        perm = rng.permutation(ei.shape[1])
        ei, heat = ei[:, perm], heat[perm]
    ref_tours, ref_it, ref_done = D.merge_tours(heat, pts, ei, sparse_graph=True)
    tours, it, done = merge_tours(heat, pts, ei, sparse_graph=True, device=dev, return_completed=True)
    assert done == ref_done
    assert tours == ref_tours
    if done[0]:
        assert it == ref_it


def test_merge_tours_batch_entry_equals_per_sample_entry(dev):
    """difusco_tsp_merge_tours (one key sort for all samples of a graph, what decode.merge_tours calls) against
    difusco_tsp_merge_tour called sample by sample: tours, iteration counters and completion flags are equal."""
    import ctypes
    import torch
    from difusco_amd import _lib
    from difusco_amd.decode import merge_tours
    from difusco_amd.synthetic import tsp_instance
    rng = np.random.default_rng(7)
    n, k, par = 300, 30, 5
    pts, ei = tsp_instance(n, k, seed=3)
    heat = np.stack([_heat(kind, pts, ei, rng) for kind in ("bits", "prob", "gauss", "bits", "prob")])
    tours, it_mean, done = merge_tours(heat, pts, ei, sparse_graph=True, parallel_sampling=par, device=dev,
                                       return_completed=True)
    L = _lib.lib()
    d = lambda a, t: torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=t)
    row, col, p32 = d(ei[0], torch.int32), d(ei[1], torch.int32), d(pts, torch.float32)
    nbytes = ctypes.c_size_t()
    _lib.check(L.difusco_tsp_merge_workspace_bytes(ei.shape[1], ctypes.byref(nbytes)))
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    its = []
    for smp in range(par):
        h = d(heat[smp], torch.float32)
        tour = np.empty(n + 1, dtype=np.int32)
        it, ok = ctypes.c_int64(), ctypes.c_int32()
        _lib.check(L.difusco_tsp_merge_tour(n, ei.shape[1], row.data_ptr(), col.data_ptr(), ctypes.c_void_p(h.data_ptr()),
                                            ctypes.c_void_p(p32.data_ptr()), ws.data_ptr(), nbytes.value,
                                            tour.ctypes.data_as(ctypes.c_void_p), ctypes.byref(it), ctypes.byref(ok), None))
        assert tour.tolist() == tours[smp] and bool(ok.value) == done[smp]
        its.append(it.value)
    assert it_mean == float(np.mean(its))
    assert sorted(tours[0][:-1]) == list(range(n)) and tours[0][0] == tours[0][-1] == 0


def test_merge_tours_full_size_properties(dev):
    """TSP-10000 / K=100 (10^6 heat entries; the reference would sort 10^8): a valid closed tour, bitwise
    deterministic, invariant under a permutation of the edge list, and made of candidate edges wherever the
    positive-score phase supplied them."""
    from difusco_amd.decode import merge_tours
    from difusco_amd.synthetic import tsp_instance
    n, k = 10000, 100
    pts, ei = tsp_instance(n, k, seed=3)
    rng = np.random.default_rng(0)
    heat = _heat("prob", pts, ei, rng)
    t1, it1, d1 = merge_tours(heat, pts, ei, sparse_graph=True, device=dev, return_completed=True)
    t2, it2, d2 = merge_tours(heat, pts, ei, sparse_graph=True, device=dev, return_completed=True)
    assert t1 == t2 and it1 == it2
    tour = t1[0]
    assert len(tour) == n + 1 and tour[0] == 0 and tour[-1] == 0 and sorted(tour[:-1]) == list(range(n))
    perm = rng.permutation(ei.shape[1])
    t3, _, _ = merge_tours(heat[perm], pts, ei[:, perm], sparse_graph=True, device=dev, return_completed=True)
    assert t3 == t1
    cand = set(zip(np.minimum(ei[0], ei[1]).tolist(), np.maximum(ei[0], ei[1]).tolist()))
    on_graph = sum((min(a, b), max(a, b)) in cand for a, b in zip(tour[:-1], tour[1:]))
    print(f"TSP-10000 decode: {on_graph}/{n} tour edges are k-NN candidates, completed={d1[0]}, merge_iterations={it1:.0f}")
    assert on_graph >= 0.98 * n


def test_sample_then_decode(dev):
    """Sampling loop + decode end to end on a small instance (pl_tsp_model.py:185-228 without 2-opt)."""
    from difusco_amd import TSPModel
    from difusco_amd.decode import merge_tours
    from oracle import difusco_oracle as O
    p = O.init_params(64, 2, 2, seed=0)
    pts, ei = O.tsp_instance(60, 12, seed=1)
    args = dict(diffusion_type="categorical", diffusion_schedule="linear", diffusion_steps=1000, sparse_factor=12,
                n_layers=2, hidden_dim=64, inference_trick="ddim", inference_diffusion_steps=10, inference_schedule="cosine")
    m = TSPModel(args, p, device=dev, seed=3)
    heat = m.sample(torch.from_numpy(pts).to(dev), torch.from_numpy(ei).to(dev))
    tours, it = merge_tours(heat, pts, ei, sparse_graph=True, device=dev)
    assert sorted(tours[0][:-1]) == list(range(60)) and it > 0


# ---- 2-opt ---------------------------------------------------------------------------------------------------
from test_decode_oracle import TWO_OPT  # noqa: E402


def _tour_len(pts, tour):
    return float(np.linalg.norm(pts[tour[:-1]] - pts[tour[1:]], axis=1).sum())


@pytest.mark.parametrize("path", TWO_OPT, ids=[os.path.basename(p)[11:-4] for p in TWO_OPT])
def test_two_opt_matches_reference_fixture(dev, path):
    """Exact moves: same tours and same iteration counter as the reference's batched_two_opt_torch."""
    from difusco_amd.decode import batched_two_opt_torch
    z = np.load(path)
    out, it = batched_two_opt_torch(z["points"], z["tours_in"], max_iterations=int(z["max_iterations"]), device=dev)
    assert it == int(z["iterations"])
    assert np.array_equal(out, z["tours_out"])


@pytest.mark.parametrize("n,batch,max_it", [(33, 2, 1000), (257, 1, 1000), (500, 4, 40), (1000, 1, 25)])
def test_two_opt_matches_oracle(dev, n, batch, max_it):
    from difusco_amd.decode import batched_two_opt_torch
    rng = np.random.default_rng(n + batch)
    pts = rng.random((n, 2))
    tours = np.stack([np.concatenate([[0], 1 + rng.permutation(n - 1), [0]]) for _ in range(batch)])
    ref, ref_it = D.batched_two_opt(pts, tours, max_iterations=max_it)
    out, it = batched_two_opt_torch(pts, tours, max_iterations=max_it, device=dev)
    assert it == ref_it and np.array_equal(out, ref)


def test_two_opt_full_size_properties(dev):
    """TSP-10000 (the reference would hold four 10^8-entry float64 matrices per iteration): tours stay permutations,
    every applied move shortens the tour, the run is bitwise deterministic, and a capped run is a prefix of a longer
    one (same moves in the same order)."""
    from difusco_amd.decode import batched_two_opt_torch
    n = 10000
    rng = np.random.default_rng(5)
    pts = rng.random((n, 2))
    order = np.argsort(pts[:, 0] // 0.05 * 10 + pts[:, 1] * (1 - 2 * ((pts[:, 0] // 0.05) % 2)))   # strip tour
    order = np.roll(order, -int(np.where(order == 0)[0][0]))
    tour0 = np.concatenate([order, [0]])[None, :]
    a, it_a = batched_two_opt_torch(pts, tour0, max_iterations=30, device=dev)
    b, it_b = batched_two_opt_torch(pts, tour0, max_iterations=30, device=dev)
    assert it_a == it_b == 30 and np.array_equal(a, b)
    assert sorted(a[0][:-1].tolist()) == list(range(n)) and a[0][0] == 0 and a[0][-1] == 0
    c, it_c = batched_two_opt_torch(pts, tour0, max_iterations=10, device=dev)
    d, it_d = batched_two_opt_torch(pts, c, max_iterations=20, device=dev)
    assert it_c == 10 and it_d == 20 and np.array_equal(d, a)
    l0, l10, l30 = _tour_len(pts, tour0[0]), _tour_len(pts, c[0]), _tour_len(pts, a[0])
    print(f"TSP-10000 2-opt: length {l0:.3f} -> {l10:.3f} (10 moves) -> {l30:.3f} (30 moves)")
    assert l30 < l10 < l0


# ---- k-NN graph construction -----------------------------------------------------------------------------------
def test_knn_graph_matches_sklearn_fixture(dev, golden_dir):  # noqa: F811
    import glob
    from difusco_amd.graph import knn_edge_index_gpu
    for p in sorted(glob.glob(os.path.join(golden_dir, "knn_*.npz"))):
        z = np.load(p)
        k, n = int(z["k"]), z["points"].shape[0]
        ei = knn_edge_index_gpu(z["points"], k, device=dev).cpu().numpy()
        assert np.array_equal(ei[0], np.repeat(np.arange(n), k))
        assert np.array_equal(ei[1].reshape(n, k), z["idx_knn"]), p


@pytest.mark.parametrize("n,k,graphs", [(1000, 100, 3), (10000, 100, 1), (17000, 8, 1), (5, 5, 2), (300, 1, 1)])
def test_knn_graph_matches_host_generator(dev, n, k, graphs):
    """Full sizes (TSP-1000 batch, TSP-10000, the global-scratch path above 16000 points) against the numpy brute
    force, including the node-id offsets of a disjoint-union batch."""
    from difusco_amd.graph import knn_edge_index_gpu
    from difusco_amd.synthetic import knn_edge_index
    pts = np.random.default_rng(n + k).random((graphs * n, 2))
    ref = np.concatenate([knn_edge_index(pts[g * n:(g + 1) * n], k) + g * n for g in range(graphs)], axis=1)
    ei = knn_edge_index_gpu(pts, k, device=dev, graphs=graphs).cpu().numpy()
    assert np.array_equal(ei, ref)


def test_knn_graph_ties_and_errors(dev):
    from difusco_amd import _lib
    from difusco_amd.graph import knn_edge_index_gpu
    pts = np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0], [1.0, 1.0], [0.5, 0.5], [0.5, 0.5]])   # duplicates + exact ties
    ei = knn_edge_index_gpu(pts, 4, device=dev).cpu().numpy()[1].reshape(6, 4)
    d = ((pts[:, None] - pts[None]) ** 2).sum(-1)
    for i in range(6):
        order = sorted(range(6), key=lambda j: (d[i, j], j))[:4]      # ties -> lower index
        assert ei[i].tolist() == order
    with pytest.raises(_lib.DifuscoHipError):
        knn_edge_index_gpu(pts, 7, device=dev)
    # many exact ties: a 20 x 20 lattice (ties at every rank, collected in parallel) and 700 coincident points (more
    # than 256 keys equal to the k-th: the serial fallback); lowest indices win in both
    gx, gy = np.meshgrid(np.arange(20) / 32.0, np.arange(20) / 32.0)
    for pts, k in [(np.stack([gx.ravel(), gy.ravel()], 1), 9), (np.full((700, 2), 0.25), 5)]:
        n = pts.shape[0]
        ei = knn_edge_index_gpu(pts, k, device=dev).cpu().numpy()[1].reshape(n, k)
        d = ((pts[:, None] - pts[None]) ** 2).sum(-1)
        ref = np.lexsort((np.broadcast_to(np.arange(n), (n, n)), d), axis=1)[:, :k]
        assert np.array_equal(ei, ref)


def test_tsp_batch_gpu_equals_host_batch(dev):
    from difusco_amd.synthetic import tsp_batch, tsp_batch_gpu
    p0, e0 = tsp_batch(200, 20, range(3, 6))
    p1, e1 = tsp_batch_gpu(200, 20, range(3, 6), dev)
    assert torch.equal(p0, p1.cpu()) and torch.equal(e0, e1.cpu())


# ---- MIS decode ------------------------------------------------------------------------------------------------
from test_decode_oracle import MIS  # noqa: E402


@pytest.mark.parametrize("path", MIS, ids=[os.path.basename(p)[11:-4] for p in MIS])
def test_mis_decode_matches_reference_fixture(dev, path):
    from difusco_amd.decode import mis_decode_np
    z = np.load(path)
    sol = mis_decode_np(z["predictions"], edge_index=z["edge_index"].astype(np.int64), device=dev)
    assert np.array_equal(sol, z["solution"].astype(int))


def test_mis_decode_batch_and_properties(dev):
    """The per-GPU MIS shard of BASELINE (16 ER graphs, n in [700,800], one call): equals the oracle, is an
    independent set, is maximal, handles score ties like a stable sort, accepts the scipy matrix of the reference."""
    import scipy.sparse
    from difusco_amd.decode import mis_decode_np
    from difusco_amd.synthetic import er_mis_edge_index
    eis, off = [], 0
    for g in range(16):
        n = int(np.random.default_rng(100 + g).integers(700, 801))
        eis.append(er_mis_edge_index(n, 0.15, seed=g) + off)
        off += n
    ei = np.concatenate(eis, 1)
    rng = np.random.default_rng(0)
    pred = rng.random(off).astype(np.float32)
    pred[rng.integers(0, off, 500)] = 0.5                      # exact ties
    ref = D.mis_decode(pred, ei, off)
    sol = mis_decode_np(pred, edge_index=ei, device=dev)
    assert np.array_equal(sol, ref)
    a, b = ei[0], ei[1]
    nonself = a != b
    assert not np.any((sol[a] == 1) & (sol[b] == 1) & nonself)          # independent
    covered = np.zeros(off, dtype=bool)
    covered[a[sol[b] == 1]] = True
    assert np.all(covered | (sol == 1))                                   # maximal
    adj = scipy.sparse.coo_matrix((np.ones_like(ei[0]), (ei[0], ei[1])))
    assert np.array_equal(mis_decode_np(pred, adj, device=dev), ref)


def test_solve_tsp_pipeline(dev):
    """k-NN -> sampling loop -> merge -> 2-opt end to end (pl_tsp_model.py:152-241) on a small instance with 3 parallel
    samples: valid tours, 2-opt never lengthens a merged tour, the best cost is the minimum."""
    from difusco_amd import TSPModel
    from difusco_amd.pipeline import solve_tsp, tour_length
    from oracle import difusco_oracle as O
    p = O.init_params(64, 2, 2, seed=0)
    pts = np.random.default_rng(4).random((80, 2))
    args = dict(diffusion_type="categorical", diffusion_schedule="linear", diffusion_steps=1000, sparse_factor=10,
                n_layers=2, hidden_dim=64, inference_trick="ddim", inference_diffusion_steps=10, inference_schedule="cosine")
    m = TSPModel(args, p, device=dev, seed=3)
    timings = {}
    tour, cost, costs, info = solve_tsp(m, pts, sparse_factor=10, parallel_sampling=3, two_opt_iterations=200, timings=timings)
    assert sorted(tour[:-1]) == list(range(80)) and tour[0] == tour[-1] == 0
    pts32 = pts.astype(np.float32).astype(np.float64)      # the reference evaluates on the batch's float32 coordinates
    assert abs(tour_length(pts32, tour) - cost) < 1e-12 and cost == min(costs) and len(costs) == 3
    assert all(c <= mc + 1e-9 for c, mc in zip(costs, info["merged_costs"]))
    assert set(timings) == {"knn", "sampling", "merge", "two_opt"}


def test_solve_tsp_sequential_and_dense(dev):
    """``sequential_sampling`` rounds (pl_tsp_model.py:185,238) and the dense TSP-50 flow of BASELINE configs[0]
    (dense denoise steps -> dense merge -> 2-opt): valid tours, parallel x sequential costs, best = minimum."""
    from difusco_amd import TSPModel
    from difusco_amd.pipeline import solve_tsp, tour_length
    from oracle import difusco_oracle as O
    p = O.init_params(64, 2, 2, seed=0)
    pts = np.random.default_rng(8).random((50, 2))
    base = dict(diffusion_type="categorical", diffusion_schedule="linear", diffusion_steps=1000, n_layers=2, hidden_dim=64,
                inference_trick="ddim", inference_diffusion_steps=8, inference_schedule="cosine")
    for sparse_factor in (10, -1):
        m = TSPModel(dict(base, sparse_factor=sparse_factor), p, device=dev, seed=3)
        tour, cost, costs, info = solve_tsp(m, pts, sparse_factor=sparse_factor, parallel_sampling=2, two_opt_iterations=100,
                                            sequential_sampling=3)
        assert sorted(tour[:-1]) == list(range(50)) and tour[0] == tour[-1] == 0
        assert len(costs) == 6 and cost == min(costs) and len(info["merged_costs"]) == 6
        pts32 = pts.astype(np.float32).astype(np.float64)          # costs are evaluated on the float32 coordinates
        assert abs(tour_length(pts32, tour) - cost) < 1e-12
        assert all(c <= mc + 1e-9 for c, mc in zip(costs, info["merged_costs"]))
        assert len(set(np.round(costs, 9))) > 1                     # the rounds drew different noise


def test_solve_mis_sequential(dev):
    from difusco_amd import MISModel
    from difusco_amd.pipeline import solve_mis
    from difusco_amd.synthetic import er_mis_edge_index
    from oracle import difusco_oracle as O
    n = 90
    ei = er_mis_edge_index(n, 0.1, seed=4)
    p = O.init_params(64, 2, 2, seed=1)
    args = dict(diffusion_type="categorical", diffusion_schedule="linear", diffusion_steps=1000, sparse_factor=-1, n_layers=2,
                hidden_dim=64, inference_trick="ddim", inference_diffusion_steps=6, inference_schedule="cosine")
    m = MISModel(args, p, device=dev, seed=5)
    sol, size, sizes = solve_mis(m, n, ei, parallel_sampling=2, sequential_sampling=3)
    assert len(sizes) == 6 and size == max(sizes) == int(sol.sum())
    a, b = ei[0], ei[1]
    assert not np.any((sol[a] == 1) & (sol[b] == 1) & (a != b))


def test_solve_mis_pipeline(dev):
    """Sampling loop -> greedy decode for 4 parallel samples of one ER graph (pl_mis_model.py:142-206): every sample's
    set is independent and maximal, equals the CPU oracle's decode of the same scores, best = largest."""
    from difusco_amd import MISModel
    from difusco_amd.decode import mis_decode_np
    from difusco_amd.pipeline import solve_mis
    from difusco_amd.synthetic import er_mis_edge_index
    from oracle import difusco_oracle as O
    n = 120
    ei = er_mis_edge_index(n, 0.1, seed=2)
    p = O.init_params(64, 2, 2, seed=1)
    args = dict(diffusion_type="categorical", diffusion_schedule="linear", diffusion_steps=1000, sparse_factor=-1, n_layers=2,
                hidden_dim=64, inference_trick="ddim", inference_diffusion_steps=10, inference_schedule="cosine")
    m = MISModel(args, p, device=dev, seed=5)
    sol, size, sizes = solve_mis(m, n, ei, parallel_sampling=4)
    assert size == max(sizes) == int(sol.sum()) and len(sizes) == 4
    a, b = ei[0], ei[1]
    assert not np.any((sol[a] == 1) & (sol[b] == 1) & (a != b))
    covered = np.zeros(n, dtype=bool)
    covered[a[sol[b] == 1]] = True
    assert np.all(covered | (sol == 1))


# ------------------------------------------------------------------------------------------------
# (f)-4 on the GPU: the MCTS heatmap text (tsp_mcts/convert_numpy_to_txt.py:18-72) from csrc/formats.hip
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["mcts_text_n30", "mcts_text_n64", "mcts_sparse_text_n1000_k50"])
def test_mcts_text_gpu_is_character_identical_to_the_reference(dev, golden_dir, tmp_path, name):  # noqa: F811
    """The GPU rows, formatted, equal the text the reference converter wrote (fixtures produced by importing
    tsp_mcts/convert_numpy_to_txt.py): every one of the N^2 "%.6f" numbers, the signed zeros included."""
    from difusco_amd import formats
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    n, prob = int(z["num_nodes"]), float(z["expected_valid_prob"])
    ref_text = bytes(z["text"]).decode()
    ei = z["edge_index"] if "edge_index" in z.files else None
    path = formats.write_mcts_heatmap(torch.from_numpy(z["heat"]).to(dev), torch.from_numpy(z["points"]).to(dev), n, str(tmp_path), 0,
                                      expected_valid_prob=prob, edge_index=None if ei is None else torch.from_numpy(ei).to(dev),
                                      use_gpu=True)
    assert open(path).read() == ref_text
    if ei is not None:          # block size and entry order do not matter; rows equal the host numpy sweeps bit for bit
        host = np.stack(list(formats.mcts_heatmap_rows(z["heat"], ei, z["points"], n, prob)))
        perm = np.random.default_rng(0).permutation(z["heat"].shape[0])
        for br, hh, ee in [(37, z["heat"], ei), (1000, z["heat"][perm], ei[:, perm])]:
            got = np.stack(list(formats.mcts_heatmap_rows_gpu(hh, ee, z["points"], n, prob, block_rows=br, device=dev)))
            assert got.dtype == np.float32 and np.array_equal(got.view(np.uint32), host.view(np.uint32))


def test_mcts_rows_gpu_equal_host_numpy_and_edge_cases(dev):
    """A k-NN heatmap at N = 3000 (9 M values, threshold among prior-only entries): GPU rows == the host numpy sweeps bit
    for bit; k = 0 (the reference's valid_values[-0]), too few positive values (IndexError upstream), duplicate entries."""
    from difusco_amd import formats
    from difusco_amd.synthetic import tsp_instance
    n, K = 3000, 40
    pts, ei = tsp_instance(n, K, seed=5)
    rng = np.random.default_rng(6)
    heat = (rng.random(ei.shape[1]) ** 3).astype(np.float32)
    heat[rng.random(ei.shape[1]) < 0.3] = 0.0
    host = np.stack(list(formats.mcts_heatmap_rows(heat, ei, pts.astype(np.float32), n, 0.02)))
    got = np.stack(list(formats.mcts_heatmap_rows_gpu(heat, ei, pts.astype(np.float32), n, 0.02, device=dev)))
    assert np.array_equal(got.view(np.uint32), host.view(np.uint32))
    # small instance: k = 0 and an unreachable k
    n2 = 12
    p2 = rng.random((n2, 2)).astype(np.float32)
    h2, e2 = formats.sparsify((rng.random((n2, n2)) ** 2).astype(np.float32))
    a = np.stack(list(formats.mcts_heatmap_rows(h2, e2, p2, n2, 0.001)))          # int(144 * 0.001) = 0
    b = np.stack(list(formats.mcts_heatmap_rows_gpu(h2, e2, p2, n2, 0.001, device=dev)))
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    far = (p2 * 50).astype(np.float32)                                            # distances > 1: negative priors
    with pytest.raises(IndexError):
        list(formats.mcts_heatmap_rows_gpu(np.zeros(1, np.float32), np.array([[0], [1]]), far, n2, 0.9, device=dev))
    with pytest.raises(ValueError):
        list(formats.mcts_heatmap_rows_gpu(np.ones(2, np.float32), np.array([[0, 0], [1, 1]]), p2, n2, 0.5, device=dev))


def test_mcts_rows_gpu_full_size(dev):
    """N = 10^4, K = 100 (the TSP-10000 heatmap): the numeric part in well under a second (31 s of host numpy in round 2);
    the threshold is the k-th largest of the 10^8 values (counted independently with torch), rows are normalised, the
    support is symmetric, and a sample of rows equals the converter's statements evaluated for those rows."""
    import ctypes
    import time
    from difusco_amd import _lib, formats
    from difusco_amd.synthetic import tsp_batch_gpu
    n, K, prob = 10000, 100, 0.02
    pts, ei = tsp_batch_gpu(n, K, range(1), dev)
    g = torch.Generator().manual_seed(8)
    heat = (torch.rand(ei.shape[1], generator=g) ** 4).to(dev)
    row, col = ei[0].int().contiguous(), ei[1].int().contiguous()
    pts = pts.float().contiguous()
    L = _lib.lib()
    nb = ctypes.c_size_t()
    _lib.check(L.difusco_mcts_heatmap_workspace_bytes(n, ei.shape[1], ctypes.byref(nb)))
    ws = torch.empty(nb.value, dtype=torch.uint8, device=dev)
    out = torch.empty((n, n), dtype=torch.float32, device=dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    thr = ctypes.c_float()
    best = 1e9
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.check(L.difusco_mcts_heatmap_prepare(n, ei.shape[1], P(row), P(col), P(heat), P(pts), prob, P(ws), ws.numel(),
                                                  ctypes.byref(thr), st))
        _lib.check(L.difusco_mcts_heatmap_rows(n, ei.shape[1], P(pts), P(ws), ws.numel(), 0, n, P(out), st))
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"MCTS heatmap rows, N={n} K={K}: {best * 1e3:.1f} ms for the numeric part (threshold {thr.value:.6g})")
    assert best < 1.0
    # the threshold is the k-th largest positive value: count with torch, block by block (same float32 operations)
    k = int(n * n * prob)
    dense_heat = torch.zeros((n, n), dtype=torch.float32, device=dev)
    dense_heat[ei[0], ei[1]] = heat
    gt = ge = 0
    top3 = torch.empty((n, 3), dtype=torch.int64, device=dev)
    for lo in range(0, n, 1000):
        dx = pts[lo:lo + 1000, None, 0] - pts[None, :, 0]
        dy = pts[lo:lo + 1000, None, 1] - pts[None, :, 1]
        v = dense_heat[lo:lo + 1000] + 0.01 * (1.0 - torch.sqrt(dx * dx + dy * dy))
        gt += int((v > thr.value).sum())
        ge += int((v >= thr.value).sum())
        top3[lo:lo + 1000] = v.topk(3, dim=1).indices
    assert gt < k <= ge, (gt, k, ge)
    sums = out.sum(dim=1)
    assert (sums - 1).abs().max().item() < 1e-4 and torch.isfinite(out).all()
    nz = out != 0
    assert torch.equal(nz, nz.t())
    # rows 3000..3007 against the converter's statements evaluated for those rows on the host (numpy, float32)
    ph, hh, t3 = pts.cpu().numpy(), dense_heat.cpu().numpy(), top3.cpu().numpy()
    rows = np.arange(3000, 3008)
    d = np.linalg.norm(ph[rows][:, None, :] - ph[None, :, :], axis=-1)                  # [8, n]
    v_r = hh[rows] + 0.01 * (1.0 - d)                                                   # v[i][j]
    v_c = hh[:, rows].T + 0.01 * (1.0 - np.linalg.norm(ph[None, :, :] - ph[rows][:, None, :], axis=-1))   # v[j][i] at [i][j]
    keep_r = v_r > thr.value
    keep_r[np.arange(8)[:, None], t3[rows]] = True
    keep_c = v_c > thr.value
    keep_c |= (t3[None, :, :] == rows[:, None, None]).any(axis=2)                      # i among the top 3 of row j
    m_r, m_c = v_r * keep_r, v_c * keep_c
    m_r[m_r != 0.0] += np.float32(1e-2)
    m_c[m_c != 0.0] += np.float32(1e-2)
    want = m_r + m_c
    want = want / want.sum(axis=1, keepdims=True)
    assert np.array_equal(out[3000:3008].cpu().numpy().view(np.uint32), want.astype(np.float32).view(np.uint32))
