"""The CPU restatement of the heatmap -> tour decode (oracle/tsp_decode_oracle.py) against fixtures produced by the
reference's own merge_tours + merge_cython (tests/golden/make_golden_decode.py).  No GPU."""
import glob
import os

import numpy as np
import pytest

from oracle import tsp_decode_oracle as D

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "tsp_decode_*.npz")))


def tour_edges(tour):
    return {(min(a, b), max(a, b)) for a, b in zip(tour[:-1], tour[1:])}


def positive_pairs(z, sample):
    """Undirected pairs whose entry of A + A^T is positive for this sample (the entries the reference walks in a
    reproducible order)."""
    ei, heat = z["edge_index"], np.split(z["heat"], int(z["parallel_sampling"]))[sample]
    n = z["points"].shape[0]
    a = np.zeros((n, n), dtype=np.float32)
    a[ei[0], ei[1]] = heat
    s = a + a.T
    i, j = np.nonzero(np.triu(s > 0, 1))
    return set(zip(i.tolist(), j.tolist()))


def check_against_fixture(z, tours, iters, done):
    par = int(z["parallel_sampling"])
    n = z["points"].shape[0]
    for s in range(par):
        ref_tour = z["tours"][s].tolist()
        assert sorted(tours[s][:-1]) == list(range(n)) and tours[s][0] == 0 and tours[s][-1] == 0
        assert bool(done[s]) == bool(z["completed"][s])
        if z["completed"][s]:
            assert tours[s] == ref_tour
            assert iters[s] == z["merge_iterations_per_sample"][s]
        else:
            # the reference ran into its zero block (order = numpy's unstable argsort): only the insertions made
            # from positive entries are reproducible; the two closing edges may or may not be positive pairs
            pos = positive_pairs(z, s)
            mine, ref = tour_edges(tours[s]) & pos, tour_edges(ref_tour) & pos
            assert len(mine ^ ref) <= 2, (len(mine), len(ref), len(mine ^ ref))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[11:-4] for p in GOLDEN])
def test_oracle_matches_reference_fixture(path):
    z = np.load(path)
    par = int(z["parallel_sampling"])
    tours, iters, done = [], [], []
    for part in np.split(z["heat"], par):
        t, it, ok = D.merge_tours(part, z["points"], z["edge_index"], sparse_graph=True, parallel_sampling=1)
        tours += t
        iters.append(it)
        done += ok
    check_against_fixture(z, tours, iters, done)
    # the batched call returns the mean counter, like the reference
    _, it_mean, _ = D.merge_tours(z["heat"], z["points"], z["edge_index"], sparse_graph=True, parallel_sampling=par)
    assert it_mean == np.mean(iters)
    if z["completed"].all():
        assert it_mean == float(z["merge_iterations"])


DENSE = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "tsp_densemerge_*.npz")))


@pytest.mark.parametrize("path", DENSE, ids=[os.path.basename(p)[15:-4] for p in DENSE])
def test_oracle_dense_merge_matches_reference_fixture(path):
    """The dense branch of merge_tours (tsp_utils.py:105-108; BASELINE configs[0], TSP-50 dense heatmaps)."""
    z = np.load(path)
    par = int(z["parallel_sampling"])
    assert len(DENSE) >= 3 and z["completed"].all()
    tours, it, done = D.merge_tours(z["heat"], z["points"], None, sparse_graph=False, parallel_sampling=par)
    assert all(done) and np.array_equal(np.asarray(tours), z["tours"]) and it == float(z["merge_iterations"])


def test_fixtures_cover_both_regimes():
    flags = np.concatenate([np.load(p)["completed"] for p in GOLDEN])
    assert flags.any() and (~flags).any() and len(GOLDEN) >= 6


TWO_OPT = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "tsp_twoopt_*.npz")))


@pytest.mark.parametrize("path", TWO_OPT, ids=[os.path.basename(p)[11:-4] for p in TWO_OPT])
def test_two_opt_oracle_matches_reference_fixture(path):
    z = np.load(path)
    out, it = D.batched_two_opt(z["points"], z["tours_in"], max_iterations=int(z["max_iterations"]))
    assert it == int(z["iterations"])
    assert np.array_equal(out, z["tours_out"])


MIS = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "mis_decode_*.npz")))


@pytest.mark.parametrize("path", MIS, ids=[os.path.basename(p)[11:-4] for p in MIS])
def test_mis_decode_oracle_matches_reference_fixture(path):
    z = np.load(path)
    n = z["predictions"].shape[0]
    sol = D.mis_decode(z["predictions"], z["edge_index"], n)
    assert np.array_equal(sol, z["solution"].astype(int))
