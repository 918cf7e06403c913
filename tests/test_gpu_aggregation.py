"""``--aggregation mean`` / ``--aggregation max`` (``train.py:52``; ``GNNLayer.aggregate``, ``gnn_encoder.py:144-191``) on the
HIP path (``-m gpu``; ABI 10: ``difusco_step_args.aggregation``).

* the reference-generated dense fixtures (``tests/golden/make_golden_agg.py``: pure reference arithmetic) through
  ``TSPModel`` at H = 64 (general kernels, two samples = two statistic segments) and at H = 256 with one sample (the FUSED
  layers + ``node_finalize``: mean divides the assembled row sum, max has instantiations whose per-tile pieces are maxima);
* sparse k-NN TSP batches (edge counts that are and are not multiples of the 32-edge tile; rows longer than a tile) and an
  Erdos-Renyi MIS graph against the oracle's restated ``torch_sparse.mean / max``, fused and unfused, both bindings, a node
  without edges included;
* ``sum`` is untouched: the default engine and an explicit ``aggregation="sum"`` are bit-identical.

Tolerance: network outputs 1e-4 absolute (north_star), observed values printed."""
import numpy as np
import pytest
import torch

from oracle import difusco_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: torch.cuda.is_available() is False")
    return torch.device("cuda:0")


def _args(kind, sparse_factor, agg, H, L, trick="ddim"):
    return dict(diffusion_type=kind, diffusion_schedule="linear", diffusion_steps=1000, sparse_factor=sparse_factor,
                n_layers=L, hidden_dim=H, inference_trick=trick, aggregation=agg)


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("agg,H,L,B", [("mean", 64, 2, 2), ("mean", 256, 3, 1), ("max", 64, 2, 2), ("max", 256, 3, 1)])
def test_golden_dense_mean_max_vs_the_imported_reference(dev, agg, H, L, B, fused):
    from conftest import load_h256_fixture
    from difusco_amd import TSPModel
    z, cat, gau = load_h256_fixture(f"tsp_dense_agg_{agg}_h{H}_l{L}_b{B}.npz")
    assert str(z["aggregation"]) == agg
    pts = torch.from_numpy(z["points"]).to(dev)
    m = TSPModel(_args("categorical", -1, agg, H, L), cat, device=dev, fused=fused)
    worst = 0.0
    for i in range(3):
        t, tt = (int(v) for v in z[f"cat{i}_t"])
        u = torch.from_numpy(z[f"cat{i}_uniform"]) if f"cat{i}_uniform" in z.files else None
        out, logits, prob = m.categorical_denoise_step(pts, torch.from_numpy(z[f"cat{i}_xt"]).to(dev), np.array([t]), dev, None,
                                                       target_t=np.array([tt]), uniform=u, return_aux=True)
        ref = np.transpose(z[f"cat{i}_logits"], (0, 2, 3, 1))
        err = float(np.abs(logits.cpu().numpy().reshape(ref.shape) - ref).max())
        worst = max(worst, err)
        assert err < TOL, f"step {i}: logits L_inf {err}"
        if tt > 0:
            ref_p = z[f"cat{i}_prob"].reshape(-1)
            e_prob = float(np.abs(prob.cpu().numpy().reshape(-1) - ref_p).max())
            assert e_prob < TOL
            safe = np.abs(z[f"cat{i}_uniform"].reshape(-1) - ref_p) > max(1e-5, e_prob)      # tie band: a bit can flip only inside the observed |prob| error
            np.testing.assert_array_equal(out.cpu().numpy().reshape(-1)[safe], z[f"cat{i}_out"].reshape(-1)[safe])
    mg = TSPModel(_args("gaussian", -1, agg, H, L), gau, device=dev, fused=fused)
    for i in range(2):
        t, tt = (int(v) for v in z[f"gau{i}_t"])
        out, pred = mg.gaussian_denoise_step(pts, torch.from_numpy(z[f"gau{i}_xt"]).to(dev), np.array([t]), dev, None,
                                             target_t=np.array([tt]), return_aux=True)
        ref = z[f"gau{i}_pred"].squeeze(1)
        err = float(np.abs(pred.cpu().numpy().reshape(ref.shape) - ref).max())
        worst = max(worst, err)
        assert err < TOL and np.abs(out.cpu().numpy().reshape(z[f"gau{i}_out"].shape) - z[f"gau{i}_out"]).max() < TOL
    print(f"dense {agg} H={H} B={B} fused={fused}: L_inf {worst:.2e} vs the imported reference")


@pytest.mark.parametrize("backend", ["ctypes", "torch"])
@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("agg", ["mean", "max"])
@pytest.mark.parametrize("H,Lyr,N,K,G", [(256, 4, 120, 10, 2), (256, 3, 101, 7, 3), (256, 2, 300, 40, 1), (128, 2, 33, 5, 3)])
def test_sparse_tsp_mean_max_vs_oracle(dev, H, Lyr, N, K, G, agg, fused, backend):
    from difusco_amd import TSPModel
    p = O.init_params(H, Lyr, 2, seed=H + N)
    pts1, ei1 = O.tsp_instance(N, K, seed=N)
    pts = torch.from_numpy(pts1).repeat(G, 1)
    ei = O.duplicate_edge_index(torch.from_numpy(ei1), N, G)
    g = torch.Generator().manual_seed(5)
    xt = (torch.randn(ei.shape[1], generator=g) > 0).float()
    u = torch.rand(ei.shape[1], generator=g)
    t, tt = 700, 650
    ref_out, ref_logits, ref_prob = O.tsp_categorical_denoise_step(p, O.CategoricalTables(), pts, xt, t, ei, tt, uniform=u,
                                                                   return_aux=True, aggregation=agg)
    # the aggregation matters for this input: sum is far away
    sum_logits = O.encoder_sparse_edge(p, pts, xt, torch.tensor([float(t)]), ei)
    assert (sum_logits - ref_logits).abs().max() > 100 * TOL
    m = TSPModel(_args("categorical", K, agg, H, Lyr), p, device=dev, fused=fused, backend=backend)
    assert m.model.aggregation == agg
    out, lg, pr = m.categorical_denoise_step(pts.to(dev), xt.to(dev), np.array([t]), dev, ei.to(dev), target_t=np.array([tt]),
                                             uniform=u, return_aux=True)
    e_log, e_prob = (lg.cpu() - ref_logits).abs().max().item(), (pr.cpu() - ref_prob.reshape(-1)).abs().max().item()
    print(f"TSP N={N} K={K} G={G} H={H} {agg} fused={fused} {backend}: logits L_inf {e_log:.2e}, prob L_inf {e_prob:.2e}")
    assert e_log < TOL and e_prob < TOL
    safe = (u - ref_prob.reshape(-1)).abs() > max(1e-5, e_prob)      # tie band: 1e-5, or the observed prob error of this (TOL-bounded) call
    assert torch.equal(out.cpu()[safe], ref_out[safe])
    # Gaussian model of the same shape, one DDIM step
    pg = O.init_params(H, Lyr, 1, seed=H + N + 1)
    xg = torch.randn(ei.shape[1], generator=g)
    ref_o, ref_p = O.tsp_gaussian_denoise_step(pg, O.GaussianTables(), pts, xg, t, ei, tt, return_aux=True, aggregation=agg)
    mg = TSPModel(_args("gaussian", K, agg, H, Lyr), pg, device=dev, fused=fused, backend=backend)
    o, pr = mg.gaussian_denoise_step(pts.to(dev), xg.to(dev), np.array([t]), dev, ei.to(dev), target_t=np.array([tt]),
                                     return_aux=True)
    assert (pr.cpu() - ref_p).abs().max() < TOL and (o.cpu() - ref_o).abs().max() < TOL


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("agg", ["mean", "max"])
def test_mis_mean_max_vs_oracle_with_an_isolated_node(dev, agg, fused):
    """ER graph in the dataset's layout (both directions + self loops) with the self loop and every edge of ONE node removed:
    a row without entries aggregates to 0 (segment_csr semantics) - h of that node is LayerNorm(U h) alone."""
    from difusco_amd import MISModel
    from difusco_amd.synthetic import er_mis_edge_index
    H, Lyr, n = 256, 3, 200
    p = O.init_params(H, Lyr, 2, seed=77)
    ei = er_mis_edge_index(n, 0.1, seed=9)
    lone = 57
    ei = torch.from_numpy(ei[:, (ei[0] != lone) & (ei[1] != lone)])
    g = torch.Generator().manual_seed(6)
    xt = (torch.randn(n, generator=g) > 0).float()
    u = torch.rand(n, generator=g)
    t, tt = 500, 469
    ref_out, ref_logits, ref_prob = O.mis_categorical_denoise_step(p, O.CategoricalTables(), xt, t, ei, tt, uniform=u,
                                                                   return_aux=True, aggregation=agg)
    m = MISModel(_args("categorical", -1, agg, H, Lyr), p, device=dev, fused=fused)
    out, lg, pr = m.categorical_denoise_step(xt.to(dev), np.array([t]), dev, ei.to(dev), target_t=np.array([tt]), uniform=u,
                                             return_aux=True)
    e_log, e_prob = (lg.cpu() - ref_logits).abs().max().item(), (pr.cpu() - ref_prob.reshape(-1)).abs().max().item()
    print(f"MIS n={n} ({ei.shape[1]} edges, node {lone} isolated) {agg} fused={fused}: logits L_inf {e_log:.2e}, prob L_inf {e_prob:.2e}")
    assert e_log < TOL and e_prob < TOL
    safe = (u - ref_prob.reshape(-1)).abs() > max(1e-5, e_prob)      # tie band: 1e-5, or the observed prob error of this (TOL-bounded) call
    assert torch.equal(out.cpu()[safe], ref_out[safe])


def test_sum_is_the_default_and_unchanged(dev):
    from difusco_amd import TSPModel
    H, Lyr, N, K = 256, 3, 80, 8
    p = O.init_params(H, Lyr, 2, seed=3)
    pts, ei = O.tsp_instance(N, K, seed=2)
    pts, ei = torch.from_numpy(pts).to(dev), torch.from_numpy(ei).to(dev)
    g = torch.Generator().manual_seed(1)
    xt = (torch.randn(ei.shape[1], generator=g) > 0).float().to(dev)
    u = torch.rand(ei.shape[1], generator=g)
    a0 = dict(diffusion_type="categorical", diffusion_schedule="linear", diffusion_steps=1000, sparse_factor=K, n_layers=Lyr,
              hidden_dim=H, inference_trick="ddim")
    outs = []
    for args in (a0, dict(a0, aggregation="sum")):
        m = TSPModel(args, p, device=dev)
        outs.append(m.categorical_denoise_step(pts, xt, np.array([600]), dev, ei, target_t=np.array([550]), uniform=u,
                                               return_aux=True))
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    with pytest.raises(ValueError):
        TSPModel(dict(a0, aggregation="median"), p, device=dev)
