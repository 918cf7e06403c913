"""GPU parity tests added in round 5 (``-m gpu``).

* the observed bound CLASS of the default engine on the five BASELINE config shapes: logits / epsilon L_inf < 1e-5
  against the CPU oracle (the north_star bound is 1e-4), tie band |u - p| > 1e-5;
* ABI 11, the fused kernel's log2(e) domain: the stand-alone layer entry converts the reference's node rows itself; the step on
  bf16 planes (unscaled: its own constants) and with max aggregation; NOTB instantiations (MIS layers carry no time bias on e);
* a prepared buffer handed to a call whose C-side predicate turns the fused path off still steps.
"""
import numpy as np
import pytest
import torch

from oracle import difusco_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-4          # north_star's bound
CLASS_TOL = 1e-5    # what the default engine (fp16x3: 22 significand bits) must stay under; observed 1e-6 .. 2e-6


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: torch.cuda.is_available() is False")
    return torch.device("cuda:0")


def _args(kind, sparse_factor=8, trick="ddim", H=256, L=12, aggregation="sum"):
    return dict(diffusion_type=kind, diffusion_schedule="linear", diffusion_steps=1000, sparse_factor=sparse_factor,
                n_layers=L, hidden_dim=H, inference_trick=trick, aggregation=aggregation)


def test_default_engine_class_tsp1000(dev):
    """ONE TSP-1000 / K = 100 graph (the metric's instance shape, BASELINE configs[2]), H = 256, 12 layers: one teacher-forced
    categorical step on the default engine against the CPU oracle, at the bound class of the engine."""
    from difusco_amd import TSPModel
    p = O.init_params(256, 12, 2, seed=20240926)
    pts, ei = O.tsp_instance(1000, 100, seed=77)
    pts, ei = torch.from_numpy(pts), torch.from_numpy(ei)
    g = torch.Generator().manual_seed(31)
    xt = (torch.randn(ei.shape[1], generator=g) > 0).float()
    u = torch.rand(ei.shape[1], generator=g)
    t, tt = 907, 876
    ref_out, ref_logits, ref_prob = O.tsp_categorical_denoise_step(p, O.CategoricalTables(), pts, xt, t, ei, tt, uniform=u, return_aux=True)
    m = TSPModel(_args("categorical", 100), p, device=dev)
    out, lg, pr = m.categorical_denoise_step(pts.to(dev), xt.to(dev), np.array([t]), dev, ei.to(dev), target_t=np.array([tt]),
                                             uniform=u, return_aux=True)
    e_log, e_prob = (lg.cpu() - ref_logits).abs().max().item(), (pr.cpu() - ref_prob.reshape(-1)).abs().max().item()
    print(f"TSP-1000 K=100 H=256 L=12 vs oracle: logits L_inf {e_log:.3e}, prob L_inf {e_prob:.3e}")
    assert e_log < CLASS_TOL and e_prob < CLASS_TOL
    safe = (u - ref_prob.reshape(-1)).abs() > 1e-5
    assert torch.equal(out.cpu()[safe], ref_out[safe])


def test_default_engine_class_gaussian_k100(dev):
    """The Gaussian step of BASELINE configs[4] (TSP-10000 / K = 100) at a size the oracle finishes in seconds (N = 2,000, same K,
    same kernels: general-x first layer, scalar-embedding GEMM, DDIM posterior), default engine, bound class 1e-5 on epsilon."""
    from difusco_amd import TSPModel
    p = O.init_params(256, 12, 1, seed=20240926)
    pts, ei = O.tsp_instance(2000, 100, seed=78)
    pts, ei = torch.from_numpy(pts), torch.from_numpy(ei)
    g = torch.Generator().manual_seed(32)
    xt = torch.randn(ei.shape[1], generator=g)
    t, tt = 624, 598
    ref_out, ref_eps = O.tsp_gaussian_denoise_step(p, O.GaussianTables(), pts, xt, t, ei, tt, return_aux=True)[:2]
    m = TSPModel(_args("gaussian", 100), p, device=dev)
    out, eps = m.gaussian_denoise_step(pts.to(dev), xt.to(dev), np.array([t]), dev, ei.to(dev), target_t=np.array([tt]), return_aux=True)
    e_eps, e_out = (eps.cpu() - ref_eps.reshape(-1)).abs().max().item(), (out.cpu() - ref_out.reshape(-1)).abs().max().item()
    print(f"Gaussian TSP-2000 K=100 H=256 L=12 vs oracle: epsilon L_inf {e_eps:.3e}, x_t-1 L_inf {e_out:.3e}")
    assert e_eps < CLASS_TOL and e_out < CLASS_TOL


def test_default_engine_class_dense_tsp50(dev):
    """BASELINE configs[0]: dense TSP-50, batch 1, default engine, bound class 1e-5."""
    from difusco_amd import TSPModel
    p = O.init_params(256, 12, 2, seed=20240926)
    g = torch.Generator().manual_seed(33)
    pts = torch.rand(1, 50, 2, generator=g)
    xt = (torch.randn(1, 50, 50, generator=g) > 0).float()
    u = torch.rand(1, 50, 50, generator=g)
    t, tt = 969, 938
    ref_out, ref_logits, ref_prob = O.tsp_categorical_denoise_step(p, O.CategoricalTables(), pts, xt, t, None, tt, uniform=u, return_aux=True)
    ref_logits = ref_logits.permute(0, 2, 3, 1)
    m = TSPModel(_args("categorical", -1), p, device=dev)
    out, lg, pr = m.categorical_denoise_step(pts.to(dev), xt.to(dev), np.array([t]), dev, None, target_t=np.array([tt]),
                                             uniform=u.reshape(-1), return_aux=True)
    e_log = (lg.cpu().reshape(ref_logits.shape) - ref_logits).abs().max().item()
    e_prob = (pr.cpu().reshape(-1) - ref_prob.reshape(-1)).abs().max().item()
    print(f"dense TSP-50 B=1 vs oracle: logits L_inf {e_log:.3e}, prob L_inf {e_prob:.3e}")
    assert e_log < CLASS_TOL and e_prob < CLASS_TOL
    safe = (u.reshape(-1) - ref_prob.reshape(-1)).abs() > 1e-5
    assert torch.equal(out.cpu().reshape(-1)[safe], ref_out.reshape(-1)[safe])


@pytest.mark.parametrize("precision,bound", [("fp16x3", 1e-5), ("bf16x3", 1e-4)])
@pytest.mark.parametrize("task", ["tsp", "mis"])
def test_log2e_domain_fused_equals_unfused(dev, precision, bound, task):
    """ABI 11: the fused layers read node rows in the log2(e) domain (bias / column-scale vectors of the node linear, constants of
    their own for the unscaled bf16 planes); the unfused kernel sequence reads the reference's rows.  Same weights, same inputs:
    the two agree at the precision class, and both agree with the oracle."""
    from difusco_amd import MISModel, TSPModel
    from difusco_amd.synthetic import er_mis_edge_index
    p = O.init_params(256, 4, 2, seed=5)
    g = torch.Generator().manual_seed(41)
    t, tt = 500, 469
    if task == "tsp":
        pts, ei = O.tsp_instance(300, 20, seed=9)
        pts, ei = torch.from_numpy(pts), torch.from_numpy(ei)
        xt = (torch.randn(ei.shape[1], generator=g) > 0).float()
        u = torch.rand(ei.shape[1], generator=g)
        ref = O.tsp_categorical_denoise_step(p, O.CategoricalTables(), pts, xt, t, ei, tt, uniform=u, return_aux=True)
        run = lambda fused: TSPModel(_args("categorical", 20, L=4), p, device=dev, precision=precision, fused=fused).categorical_denoise_step(
            pts.to(dev), xt.to(dev), np.array([t]), dev, ei.to(dev), target_t=np.array([tt]), uniform=u, return_aux=True)
    else:
        ei = torch.from_numpy(er_mis_edge_index(400, 0.05, seed=3))
        xt = (torch.randn(400, generator=g) > 0).float()
        u = torch.rand(400, generator=g)
        ref = O.mis_categorical_denoise_step(p, O.CategoricalTables(), xt, t, ei, tt, uniform=u, return_aux=True)
        run = lambda fused: MISModel(_args("categorical", -1, L=4), p, device=dev, precision=precision, fused=fused).categorical_denoise_step(
            xt.to(dev), np.array([t]), dev, ei.to(dev), target_t=np.array([tt]), uniform=u, return_aux=True)
    _, lf, pf = run(True)
    _, lu, pu = run(False)
    e_fu = (lf - lu).abs().max().item()
    e_fo, e_uo = (lf.cpu() - ref[1]).abs().max().item(), (lu.cpu() - ref[1]).abs().max().item()
    print(f"{task} {precision}: fused vs unfused {e_fu:.3e}, fused vs oracle {e_fo:.3e}, unfused vs oracle {e_uo:.3e}")
    assert e_fu < bound and e_fo < bound and e_uo < bound


def test_prepared_buffer_with_a_step_that_leaves_the_fused_path(dev):
    """``prepare()`` may hand out a buffer for a configuration whose STEP then decides against the fused path on the C
    side; the step must still run (it recomputes from the points) and equal the stateless step."""
    from difusco_amd import TSPModel
    p = O.init_params(256, 2, 2, seed=6)
    pts, ei = O.tsp_instance(200, 10, seed=2)
    pts, ei = torch.from_numpy(pts).to(dev), torch.from_numpy(ei).to(dev)
    g = torch.Generator().manual_seed(4)
    xt = (torch.randn(ei.shape[1], generator=g) > 0).float().to(dev)
    u = torch.rand(ei.shape[1], generator=g)
    m = TSPModel(_args("categorical", 10, L=2), p, device=dev)
    g_csr = m.prepare_graph(ei, pts.shape[0], points=pts)
    buf = m.model.prepare(g_csr, pts)
    assert buf is not None
    t, tt = 400, 380
    a = m.categorical_denoise_step(pts, xt, np.array([t]), dev, ei, target_t=np.array([tt]), uniform=u, return_aux=True)
    m.model.fused = False      # the same engine, now on the unfused sequence: the prepared buffer is ignored by the C side
    m._prep_cache.clear()
    b = m.categorical_denoise_step(pts, xt, np.array([t]), dev, ei, target_t=np.array([tt]), uniform=u, return_aux=True)
    assert (a[1] - b[1]).abs().max().item() < CLASS_TOL
