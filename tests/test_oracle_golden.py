"""The CPU oracle (oracle/difusco_oracle.py) against the golden fixtures generated from the imported
reference by tests/golden/make_golden.py.  This is what pins the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import difusco_oracle as O

TOL = 2e-5  # fp32 op-for-op restatement on the same CPU backend; observed differences are ~1e-6


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


@pytest.fixture(scope="module")
def weights(golden_dir):
    z = load(golden_dir, "weights_h64_l2.npz")
    cat = {k: torch.from_numpy(z[k]) for k in z.files if k != "provenance" and not k.startswith("gaussian_")}
    gau = dict(cat)
    for k in z.files:
        if k.startswith("gaussian_"):
            gau[k[len("gaussian_"):]] = torch.from_numpy(z[k])
    return cat, gau


def test_tables_and_schedule(golden_dir):
    z = load(golden_dir, "schedules.npz")
    assert str(z["provenance"]).startswith("reference")
    for kind in ("linear", "cosine"):
        c, g = O.CategoricalTables(1000, kind), O.GaussianTables(1000, kind)
        np.testing.assert_array_equal(c.Q_bar, z[f"Q_bar_{kind}"])
        np.testing.assert_array_equal(g.alphabar, z[f"alphabar_{kind}"])
        for S in (50, 7, 1000):
            got = np.array([O.inference_schedule(kind, 1000, S, i) for i in range(S)])
            np.testing.assert_array_equal(got, z[f"sched_{kind}_{S}"])
    t = torch.from_numpy(z["temb_t"])
    np.testing.assert_array_equal(O.timestep_embedding(t, 64).numpy(), z["temb_64"])
    np.testing.assert_array_equal(O.timestep_embedding(t, 256).numpy(), z["temb_256"])
    # the published configuration: 50 cosine steps start at (1000, 969) and end at (1, 0)
    s = z["sched_cosine_50"]
    assert tuple(s[0]) == (1000, 969) and tuple(s[-1]) == (1, 0)


def test_posteriors(golden_dir):
    z = load(golden_dir, "posteriors.npz")
    tab = O.CategoricalTables(1000, "linear")
    for i in range(6):
        t, tt = (int(v) for v in z[f"cat{i}_t"])
        tt = None if tt < 0 else tt
        x0, xt = torch.from_numpy(z[f"cat{i}_x0"]), torch.from_numpy(z[f"cat{i}_xt"])
        u = torch.from_numpy(z[f"cat{i}_uniform"]) if f"cat{i}_uniform" in z.files else None
        out, prob = O.categorical_posterior(tab, t, tt, x0, xt, True, uniform=u)
        if u is not None:
            np.testing.assert_allclose(prob.numpy(), z[f"cat{i}_prob"], rtol=0, atol=1e-7)
        np.testing.assert_allclose(out.numpy(), z[f"cat{i}_out"], rtol=0, atol=1e-7)
    gt = O.GaussianTables(1000, "linear")
    for i in range(5):
        t, tt = (int(v) for v in z[f"gau{i}_t"])
        trick = "ddim" if int(z[f"gau{i}_trick"]) else None
        noise = torch.from_numpy(z[f"gau{i}_noise"]) if f"gau{i}_noise" in z.files else None
        out = O.gaussian_posterior(gt, t, tt, torch.from_numpy(z[f"gau{i}_pred"]),
                                   torch.from_numpy(z[f"gau{i}_xt"]), trick, noise)
        np.testing.assert_allclose(out.numpy(), z[f"gau{i}_out"], rtol=0, atol=1e-6)


def test_dense_tsp(golden_dir, weights):
    z = load(golden_dir, "tsp_dense_h64_l2.npz")
    assert "none executed" in str(z["provenance"])
    cat, gau = weights
    pts = torch.from_numpy(z["points"])
    tab, gt = O.CategoricalTables(), O.GaussianTables()
    for i in range(3):
        t, tt = (int(v) for v in z[f"cat{i}_t"])
        xt = torch.from_numpy(z[f"cat{i}_xt"])
        u = torch.from_numpy(z[f"cat{i}_uniform"]) if f"cat{i}_uniform" in z.files else None
        out, logits, prob = O.tsp_categorical_denoise_step(cat, tab, pts, xt, t, None, tt, uniform=u, return_aux=True)
        np.testing.assert_allclose(logits.numpy(), z[f"cat{i}_logits"], rtol=0, atol=TOL)
        if u is not None:
            ref_p = z[f"cat{i}_prob"]
            np.testing.assert_allclose(prob.numpy(), ref_p, rtol=0, atol=TOL)
            safe = np.abs(z[f"cat{i}_uniform"] - ref_p) > 1e-5
            np.testing.assert_array_equal(out.numpy()[safe], z[f"cat{i}_out"][safe])
        else:
            np.testing.assert_allclose(out.numpy(), z[f"cat{i}_out"], rtol=0, atol=TOL)
    for i in range(2):
        t, tt = (int(v) for v in z[f"gau{i}_t"])
        noise = torch.from_numpy(z[f"gau{i}_noise"]) if f"gau{i}_noise" in z.files else None
        out, pred = O.tsp_gaussian_denoise_step(gau, gt, pts, torch.from_numpy(z[f"gau{i}_xt"]), t, None, tt,
                                                noise=noise, return_aux=True)
        np.testing.assert_allclose(pred.numpy(), z[f"gau{i}_pred"].squeeze(1), rtol=0, atol=TOL)
        np.testing.assert_allclose(out.numpy(), z[f"gau{i}_out"], rtol=0, atol=TOL)


@pytest.mark.parametrize("G", [1, 3])
@pytest.mark.parametrize("v_on_edges", [True, False])
def test_sparse_tsp(golden_dir, weights, G, v_on_edges):
    z = load(golden_dir, f"tsp_sparse_h64_l2_g{G}.npz")
    assert "substitute aggregation" in str(z["provenance"])
    cat, gau = weights
    pts, ei = torch.from_numpy(z["points"]), torch.from_numpy(z["edge_index"])
    tab, gt = O.CategoricalTables(), O.GaussianTables()
    for i in range(4):
        t, tt = (int(v) for v in z[f"cat{i}_t"])
        xt = torch.from_numpy(z[f"cat{i}_xt"])
        u = torch.from_numpy(z[f"cat{i}_uniform"]) if f"cat{i}_uniform" in z.files else None
        out, logits, prob = O.tsp_categorical_denoise_step(cat, tab, pts, xt, t, ei, tt, uniform=u,
                                                           v_on_edges=v_on_edges, return_aux=True)
        np.testing.assert_allclose(logits.numpy(), z[f"cat{i}_logits"], rtol=0, atol=TOL)
        if u is not None:
            ref_p = z[f"cat{i}_prob"]
            np.testing.assert_allclose(prob.numpy(), ref_p, rtol=0, atol=TOL)
            safe = (np.abs(z[f"cat{i}_uniform"] - ref_p) > 1e-5).reshape(-1)
            np.testing.assert_array_equal(out.numpy()[safe], z[f"cat{i}_out"][safe])
        else:
            np.testing.assert_allclose(out.numpy(), z[f"cat{i}_out"], rtol=0, atol=TOL)
    for i in range(2):
        t, tt = (int(v) for v in z[f"gau{i}_t"])
        noise = torch.from_numpy(z[f"gau{i}_noise"]) if f"gau{i}_noise" in z.files else None
        out, pred = O.tsp_gaussian_denoise_step(gau, gt, pts, torch.from_numpy(z[f"gau{i}_xt"]), t, ei, tt,
                                                noise=noise, v_on_edges=v_on_edges, return_aux=True)
        np.testing.assert_allclose(pred.numpy(), z[f"gau{i}_pred"].squeeze(1), rtol=0, atol=TOL)
        np.testing.assert_allclose(out.numpy(), z[f"gau{i}_out"], rtol=0, atol=TOL)


def test_groupnorm_couples_the_batch(golden_dir, weights):
    """the sparse head normalises over ALL edges of the call - the G=3 fixture must not
    equal three independent G=1 calls, and the oracle must reproduce exactly that coupling."""
    z1, z3 = load(golden_dir, "tsp_sparse_h64_l2_g1.npz"), load(golden_dir, "tsp_sparse_h64_l2_g3.npz")
    cat, _ = weights
    E1 = z1["edge_index"].shape[1]
    xt3 = torch.from_numpy(z3["cat0_xt"])
    one = O.encoder_sparse_edge(cat, torch.from_numpy(z1["points"]), xt3[:E1].float(), torch.tensor([1000.0]),
                                torch.from_numpy(z1["edge_index"]))
    assert np.abs(one.numpy() - z3["cat0_logits"][:E1]).max() > 1e-3


def test_sparse_mis(golden_dir, weights):
    z = load(golden_dir, "mis_sparse_h64_l2.npz")
    cat, gau = weights
    ei = torch.from_numpy(z["edge_index"])
    tab, gt = O.CategoricalTables(), O.GaussianTables()
    for i in range(3):
        t, tt = (int(v) for v in z[f"cat{i}_t"])
        xt = torch.from_numpy(z[f"cat{i}_xt"])
        u = torch.from_numpy(z[f"cat{i}_uniform"]) if f"cat{i}_uniform" in z.files else None
        out, logits, prob = O.mis_categorical_denoise_step(cat, tab, xt, t, ei, tt, uniform=u, return_aux=True)
        np.testing.assert_allclose(logits.numpy(), z[f"cat{i}_logits"], rtol=0, atol=TOL)
        if u is not None:
            ref_p = z[f"cat{i}_prob"]
            np.testing.assert_allclose(prob.numpy(), ref_p, rtol=0, atol=TOL)
            safe = (np.abs(z[f"cat{i}_uniform"] - ref_p) > 1e-5).reshape(-1)
            np.testing.assert_array_equal(out.numpy()[safe], z[f"cat{i}_out"][safe])
        else:
            np.testing.assert_allclose(out.numpy(), z[f"cat{i}_out"], rtol=0, atol=TOL)
    for i in range(2):
        t, tt = (int(v) for v in z[f"gau{i}_t"])
        noise = torch.from_numpy(z[f"gau{i}_noise"]) if f"gau{i}_noise" in z.files else None
        out, pred = O.mis_gaussian_denoise_step(gau, gt, torch.from_numpy(z[f"gau{i}_xt"]), t, ei, tt,
                                                noise=noise, return_aux=True)
        np.testing.assert_allclose(pred.numpy(), z[f"gau{i}_pred"].squeeze(1), rtol=0, atol=TOL)
        np.testing.assert_allclose(out.numpy(), z[f"gau{i}_out"], rtol=0, atol=TOL)


def test_duplicate_edge_index_matches_fixture(golden_dir):
    z1, z3 = load(golden_dir, "tsp_sparse_h64_l2_g1.npz"), load(golden_dir, "tsp_sparse_h64_l2_g3.npz")
    got = O.duplicate_edge_index(torch.from_numpy(z1["edge_index"]), int(z1["nodes_per_graph"]), 3)
    np.testing.assert_array_equal(got.numpy(), z3["edge_index"])


def test_knn_graph_layout():
    pts, ei = O.tsp_instance(64, 8, seed=3)
    assert ei.shape == (2, 64 * 8)
    assert (ei[0] == np.repeat(np.arange(64), 8)).all()
    assert (ei[1].reshape(64, 8)[:, 0] == np.arange(64)).all()  # self is the nearest neighbour
    d = np.linalg.norm(pts[ei[0]] - pts[ei[1]], axis=1).reshape(64, 8)
    assert (np.diff(d, axis=1) >= 0).all()


# ------------------------------------------------------------------------------------------------
# production width (H=256, 3 layers): fixtures of tests/golden/make_golden_h256.py.  The weights come from
# O.init_params(seed) and were loaded into the imported reference with strict=True (hash checked on load).
# ------------------------------------------------------------------------------------------------
def _check_cat_steps(z, n_steps, step):
    for i in range(n_steps):
        t, tt = (int(v) for v in z[f"cat{i}_t"])
        u = torch.from_numpy(z[f"cat{i}_uniform"]) if f"cat{i}_uniform" in z.files else None
        out, logits, prob = step(torch.from_numpy(z[f"cat{i}_xt"]), t, tt, u)
        np.testing.assert_allclose(logits.numpy(), z[f"cat{i}_logits"], rtol=0, atol=TOL)
        if u is not None:
            ref_p = z[f"cat{i}_prob"]
            np.testing.assert_allclose(prob.numpy().reshape(ref_p.shape), ref_p, rtol=0, atol=TOL)
            safe = (np.abs(z[f"cat{i}_uniform"] - ref_p) > 1e-5).reshape(-1)
            np.testing.assert_array_equal(out.numpy().reshape(-1)[safe], z[f"cat{i}_out"].reshape(-1)[safe])
        else:
            np.testing.assert_allclose(out.numpy(), z[f"cat{i}_out"], rtol=0, atol=TOL)


def _check_gau_steps(z, n_steps, step):
    for i in range(n_steps):
        t, tt = (int(v) for v in z[f"gau{i}_t"])
        out, pred = step(torch.from_numpy(z[f"gau{i}_xt"]), t, tt)
        np.testing.assert_allclose(pred.numpy(), z[f"gau{i}_pred"].squeeze(1), rtol=0, atol=TOL)
        np.testing.assert_allclose(out.numpy(), z[f"gau{i}_out"], rtol=0, atol=TOL)


def test_h256_dense_tsp():
    from conftest import load_h256_fixture
    z, cat, gau = load_h256_fixture("tsp_dense_h256_l3_b1.npz")
    assert "none executed" in str(z["provenance"])
    pts = torch.from_numpy(z["points"])
    tab, gt = O.CategoricalTables(), O.GaussianTables()
    _check_cat_steps(z, 3, lambda xt, t, tt, u: O.tsp_categorical_denoise_step(cat, tab, pts, xt, t, None, tt, uniform=u,
                                                                               return_aux=True))
    _check_gau_steps(z, 2, lambda xt, t, tt: O.tsp_gaussian_denoise_step(gau, gt, pts, xt, t, None, tt, return_aux=True))


@pytest.mark.parametrize("G", [1, 3])
def test_h256_sparse_tsp(G):
    from conftest import load_h256_fixture
    z, cat, gau = load_h256_fixture(f"tsp_sparse_h256_l3_g{G}.npz")
    assert "substitute aggregation" in str(z["provenance"])
    pts, ei = torch.from_numpy(z["points"]), torch.from_numpy(z["edge_index"])
    tab, gt = O.CategoricalTables(), O.GaussianTables()
    _check_cat_steps(z, 4, lambda xt, t, tt, u: O.tsp_categorical_denoise_step(cat, tab, pts, xt, t, ei, tt, uniform=u,
                                                                               return_aux=True))
    _check_gau_steps(z, 2, lambda xt, t, tt: O.tsp_gaussian_denoise_step(gau, gt, pts, xt, t, ei, tt, return_aux=True))


def test_h256_sparse_mis():
    from conftest import load_h256_fixture
    z, cat, gau = load_h256_fixture("mis_sparse_h256_l3.npz")
    ei = torch.from_numpy(z["edge_index"])
    tab, gt = O.CategoricalTables(), O.GaussianTables()
    _check_cat_steps(z, 3, lambda xt, t, tt, u: O.mis_categorical_denoise_step(cat, tab, xt, t, ei, tt, uniform=u,
                                                                               return_aux=True))
    _check_gau_steps(z, 2, lambda xt, t, tt: O.mis_gaussian_denoise_step(gau, gt, xt, t, ei, tt, return_aux=True))


def test_tsp50_dense_full_width_full_length(golden_dir):
    """configs[0] of BASELINE.json pinned end to end on PURE reference arithmetic (tier B, no substitute code anywhere):
    TSP-50 dense categorical, H=256, 12 layers, all 50 cosine steps of test_step (pl_tsp_model.py:185-222).  The oracle is
    teacher-forced with the reference's own x_t of every step."""
    import os
    z = np.load(os.path.join(golden_dir, "tsp50_dense_h256_l12_50steps.npz"))
    assert "none executed" in str(z["provenance"])
    p = O.init_params(int(z["hidden"]), int(z["n_layers"]), 2, seed=int(z["seed"]))
    assert O.params_sha256(p) == str(z["sha"])
    pts, tab = torch.from_numpy(z["points"]), O.CategoricalTables()
    worst_l = worst_p = 0.0
    for i in range(int(z["steps"])):
        t, tt = (int(v) for v in z["t"][i])
        assert (t, tt) == tuple(O.inference_schedule("cosine", 1000, 50, i))
        xt = torch.from_numpy(z["xt_in"][i]).float()
        u = torch.from_numpy(z["uniform"][i]) if tt > 0 else None
        out, logits, prob = O.tsp_categorical_denoise_step(p, tab, pts, xt, t, None, tt, uniform=u, return_aux=True)
        worst_l = max(worst_l, float(np.abs(logits.numpy() - z["logits"][i]).max()))      # both [B,C,V,V]
        if tt > 0:
            worst_p = max(worst_p, float(np.abs(prob.numpy().reshape(-1) - z["prob"][i].reshape(-1)).max()))
            safe = np.abs(z["uniform"][i].reshape(-1) - z["prob"][i].reshape(-1)) > 1e-5
            np.testing.assert_array_equal(out.numpy().reshape(-1)[safe], z["out"][i].reshape(-1)[safe])
            if i + 1 < int(z["steps"]):       # the fixture is the reference's own free-running chain
                np.testing.assert_array_equal(z["out"][i].astype(np.int8), z["xt_in"][i + 1])
        else:
            np.testing.assert_allclose(out.numpy().reshape(-1), z["out"][i].reshape(-1), rtol=0, atol=2e-5)
    print(f"TSP-50 dense 50 steps, oracle vs imported reference: logits L_inf {worst_l:.2e}, prob L_inf {worst_p:.2e}")
    assert worst_l < 2e-5 and worst_p < 2e-5


# ------------------------------------------------------------------------------------------------
# --aggregation mean / max (gnn_encoder.py:144-191).  The dense branch is pure torch, so the dense fixtures
# (tests/golden/make_golden_agg.py) are reference outputs; the sparse branch's torch_sparse.mean / max are restated in
# O.segment_aggregate and tied to the dense semantics through the complete graph.
# ------------------------------------------------------------------------------------------------
AGG_FIXTURES = [("mean", 64, 2, 2), ("mean", 256, 3, 1), ("max", 64, 2, 2), ("max", 256, 3, 1)]


@pytest.mark.parametrize("agg,H,L,B", AGG_FIXTURES)
def test_dense_aggregation_mean_max_vs_reference(agg, H, L, B):
    from conftest import load_h256_fixture
    z, cat, gau = load_h256_fixture(f"tsp_dense_agg_{agg}_h{H}_l{L}_b{B}.npz")
    assert "none executed" in str(z["provenance"]) and str(z["aggregation"]) == agg
    pts = torch.from_numpy(z["points"])
    tab, gt = O.CategoricalTables(), O.GaussianTables()
    _check_cat_steps(z, 3, lambda xt, t, tt, u: O.tsp_categorical_denoise_step(cat, tab, pts, xt, t, None, tt, uniform=u,
                                                                               return_aux=True, aggregation=agg))
    _check_gau_steps(z, 2, lambda xt, t, tt: O.tsp_gaussian_denoise_step(gau, gt, pts, xt, t, None, tt, return_aux=True,
                                                                         aggregation=agg))
    # the fixture discriminates: the other two aggregations are far outside the tolerance
    xt0 = torch.from_numpy(z["cat0_xt"]).float()
    for other in {"sum", "mean", "max"} - {agg}:
        wrong = O.encoder_dense(cat, pts, xt0, torch.tensor([1000.0]), other)
        assert np.abs(wrong.numpy() - z["cat0_logits"]).max() > 100 * TOL, other


def complete_graph(V):
    """edge_index of the complete graph with self loops, row-major: the edge list the dense mode stands for"""
    i = torch.arange(V).repeat_interleave(V)
    j = torch.arange(V).repeat(V)
    return torch.stack([i, j])


@pytest.mark.parametrize("agg", ["mean", "max"])
def test_sparse_aggregation_on_the_complete_graph_is_the_reference_dense_output(agg):
    """One sample: a single GroupNorm statistic segment, so the sparse encoder on the complete graph IS the dense forward
    (gnn_encoder.py:350-381 vs :383-402) - the restated torch_sparse.mean / max against reference outputs."""
    from conftest import load_h256_fixture
    z, cat, gau = load_h256_fixture(f"tsp_dense_agg_{agg}_h256_l3_b1.npz")
    V = z["points"].shape[1]
    ei = complete_graph(V)
    pts = torch.from_numpy(z["points"][0])
    for i in range(3):
        t = int(z[f"cat{i}_t"][0])
        xt = torch.from_numpy(z[f"cat{i}_xt"]).float().reshape(-1)
        logits = O.encoder_sparse_edge(cat, pts, xt, torch.tensor([float(t)]), ei, aggregation=agg)      # [E, 2]
        ref = np.transpose(z[f"cat{i}_logits"], (0, 2, 3, 1)).reshape(-1, 2)
        np.testing.assert_allclose(logits.numpy(), ref, rtol=0, atol=TOL)
    xg = torch.from_numpy(z["gau0_xt"]).float().reshape(-1)
    pred = O.encoder_sparse_edge(gau, pts, xg, torch.tensor([1000.0]), ei, aggregation=agg)
    np.testing.assert_allclose(pred.numpy().reshape(-1), z["gau0_pred"].reshape(-1), rtol=0, atol=TOL)


def test_segment_aggregate_semantics():
    """sum / mean / max over the entries of a row, 0 for a row without entries, any entry order (segment_csr semantics)"""
    g = torch.Generator().manual_seed(3)
    rows = torch.tensor([4, 0, 2, 0, 4, 4, 6, 2, 0])
    v = torch.randn(9, 5, generator=g) - 2.0          # all-negative rows too: the maximum of a non-empty row may be < 0
    for agg, fn in (("sum", lambda x: x.sum(0)), ("mean", lambda x: x.mean(0)), ("max", lambda x: x.max(0)[0])):
        out = O.segment_aggregate(v, rows, 8, agg)
        for r in range(8):
            sel = v[rows == r]
            want = fn(sel) if sel.shape[0] else torch.zeros(5)
            np.testing.assert_allclose(out[r].numpy(), want.numpy(), rtol=0, atol=1e-6)
    with pytest.raises(ValueError):
        O.segment_aggregate(v, rows, 8, "median")


def test_segment_aggregate_against_torch_segment_reduce():
    """An independent implementation of the same reductions (PyTorch's own ``torch.segment_reduce`` over sorted segments): equal on
    every non-empty row; the empty-row convention differs by design (torch: 0 / nan / -inf; torch_scatter's segment_csr, which is
    what torch_sparse calls: 0 for every reduction) and is the one part that rests on the published semantics alone."""
    g = torch.Generator().manual_seed(0)
    n_rows = 50
    lengths = torch.randint(0, 9, (n_rows,), generator=g)
    rows = torch.repeat_interleave(torch.arange(n_rows), lengths)
    v = torch.randn(rows.numel(), 7, generator=g)
    ne = lengths > 0
    assert bool((~ne).any()) and bool(ne.any())
    for agg in ("sum", "mean", "max"):
        ref = torch.segment_reduce(v, agg, lengths=lengths, axis=0, unsafe=False)
        out = O.segment_aggregate(v, rows, n_rows, agg)
        np.testing.assert_allclose(out[ne].numpy(), ref[ne].numpy(), rtol=0, atol=1e-6)
        assert float(out[~ne].abs().max()) == 0.0
