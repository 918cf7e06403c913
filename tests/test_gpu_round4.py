"""GPU parity tests added in round 4 (``-m gpu``): the thin spots VERDICT r3 named.

* on-device normal RNG (``philox_normal``, csrc/common.h): moments, Kolmogorov-Smirnov distance, stream independence by
  (seed, offset), and the DDPM branch of a Gaussian step driven by it (``pl_meta_model.py:161-169``) - every earlier
  Gaussian parity test injects ``noise=``;
* full-depth (H = 256, L = 12) oracle steps at TSP-500 / K = 50 and on an ER-750 MIS graph through the DEFAULT engine;
* the whole batch of BASELINE configs[2] in ONE call (64 x TSP-1000, E = 6.4 M): determinism, fused == unfused;
* a call with n_nodes >= 2^20 (register-gather instantiation of the fused kernel, chosen automatically);
* prepared state (ABI 9: ``difusco_prepare`` / ``difusco_time_bias_rows``): bit-identical to the stateless step.

Tolerances as in test_gpu_parity.py: network outputs 1e-4 absolute (north_star), observed values printed."""
import ctypes
import math

import numpy as np
import pytest
import torch

from oracle import difusco_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-4          # north_star's bound
CLASS_TOL = 1e-5    # the bound CLASS of the default engine (fp16x3, 22 significand bits; observed 1e-6 .. 2e-6): VERDICT r4 #5 - a tenfold
                    # precision regression must not pass silently under the 1e-4 bound


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: torch.cuda.is_available() is False")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def L():
    from difusco_amd import _lib
    return _lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _args(kind, sparse_factor=8, trick="ddim", H=256, L=12):
    return dict(diffusion_type=kind, diffusion_schedule="linear", diffusion_steps=1000, sparse_factor=sparse_factor,
                n_layers=L, hidden_dim=H, inference_trick=trick)


# ------------------------------------------------------------------------------------------------
# Philox normal stream
# ------------------------------------------------------------------------------------------------
def _normal_draws(L, dev, n, seed, offset):
    """z[i] = philox_normal(seed, offset, i) through the C ABI: x_s = a (x_t - b eps) + d z with a = d = 1, b = 0, x_t = 0."""
    post = np.array([1, 0, 0, 1, 1, 0, 0, 0], dtype=np.float32)
    zero = torch.zeros(n, device=dev)
    out = torch.empty(n, device=dev)
    L.check(L.lib().difusco_gaussian_posterior(_p(zero), _p(zero), post.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                               L.RAND_PHILOX, None, seed, offset, _p(out), n, _stream()))
    torch.cuda.synchronize()
    return out.cpu().double().numpy()


def _assert_standard_normal(z, what):
    """Moments within ~5 standard errors, KS distance below the 0.1 % critical value, no lag-1 correlation."""
    from scipy import stats
    n = z.size
    assert np.isfinite(z).all(), what
    m, v = z.mean(), z.var()
    sk = ((z - m) ** 3).mean() / v ** 1.5
    m4 = ((z - m) ** 4).mean() / v ** 2
    ks = stats.kstest(z, "norm").statistic
    lag1 = np.corrcoef(z[:-1], z[1:])[0, 1]
    print(f"{what}: n {n}  mean {m:+.2e}  var {v:.5f}  skew {sk:+.2e}  kurtosis {m4:.4f}  KS D {ks:.2e}  lag-1 corr {lag1:+.2e}  "
          f"max |z| {np.abs(z).max():.2f}")
    assert abs(m) < 5 / math.sqrt(n)
    assert abs(v - 1) < 5 * math.sqrt(2 / n)
    assert abs(sk) < 5 * math.sqrt(6 / n)
    assert abs(m4 - 3) < 5 * math.sqrt(96 / n)              # var of the 4th-moment estimator of N(0,1) is 96/n
    assert ks < 1.95 / math.sqrt(n)                         # Kolmogorov: alpha = 0.001
    assert abs(lag1) < 5 / math.sqrt(n)
    # tails: Box-Muller on 24-bit uniforms reaches sqrt(2 ln 2^24) = 5.77 sigma; at n = 2^20 draws beyond 4 sigma exist
    assert 4.0 < np.abs(z).max() < 5.8


def test_philox_normal_statistics_and_stream_independence(dev, L):
    n = 1 << 20
    z = _normal_draws(L, dev, n, seed=11, offset=0)
    _assert_standard_normal(z, "philox_normal(11, 0, .)")
    assert np.array_equal(z, _normal_draws(L, dev, n, seed=11, offset=0))          # reproducible
    for seed, off in [(11, 1), (12, 0), (11, 1 << 40), (11 + (1 << 40), 0)]:          # every key word / offset word matters
        w = _normal_draws(L, dev, n, seed=seed, offset=off)
        _assert_standard_normal(w, f"philox_normal({seed}, {off}, .)")
        c = np.corrcoef(z, w)[0, 1]
        print(f"  corr with (11, 0): {c:+.2e}")
        assert not np.array_equal(z, w) and abs(c) < 5 / math.sqrt(n)
    # the stream is a function of the element index only: a shorter call sees the same leading draws
    assert np.array_equal(z[: 1000], _normal_draws(L, dev, 1000, seed=11, offset=0))


def test_gaussian_ddpm_branch_with_on_device_noise(dev):
    """``gaussian_posterior``'s DDPM branch (``pl_meta_model.py:161-169``: ``inference_trick is None or t <= 1``) with the
    noise drawn ON THE DEVICE (no ``noise=``): x_s - a (x_t - b eps) must be d z with z a standard normal stream.  (1) trick
    None at t = 500: d > 0, z recovered from the returned eps passes the statistics; a second call draws a different
    stream (the call counter is the Philox offset); (2) the final step of every Gaussian chain, t = 1 -> 0 under the default
    DDIM trick: it takes this branch with d = sqrt(beta_tilde) = 0 (alphabar[0] = 1), so the result is a (x_t - b eps)
    exactly - and finite, which needs a finite z."""
    from difusco_amd import TSPModel
    H, Lyr, N, K, G = 256, 3, 512, 16, 8
    p = O.init_params(H, Lyr, 1, seed=5)
    from difusco_amd.synthetic import tsp_batch
    pts, ei = tsp_batch(N, K, range(G), device=dev)
    E = ei.shape[1]
    xt = torch.randn(E, generator=torch.Generator().manual_seed(3)).to(dev)
    m = TSPModel(_args("gaussian", K, trick=None, L=Lyr), p, device=dev, seed=77)
    a, b, c, d, branch = (float(v) for v in m.diffusion.posterior_constants(500, 499, None))
    assert branch == 1.0 and d > 0
    zs = []
    for _ in range(2):
        out, eps = m.gaussian_denoise_step(pts, xt, np.array([500]), dev, ei, target_t=np.array([499]), return_aux=True)
        base = np.float32(a) * (xt.cpu().numpy() - np.float32(b) * eps.cpu().numpy())
        zs.append((out.cpu().numpy().astype(np.float64) - base.astype(np.float64)) / d)
    n = zs[0].size
    for k, z in enumerate(zs):
        # recovered through an fp32 subtraction: |x_s| ~ 1, d ~ 0.1 -> noise of ~1e-6 on z, invisible to the statistics
        from scipy import stats
        ks = stats.kstest(z, "norm").statistic
        print(f"DDPM step call {k}: n {n}  mean {z.mean():+.2e}  var {z.var():.5f}  KS D {ks:.2e}")
        assert abs(z.mean()) < 5 / math.sqrt(n) and abs(z.var() - 1) < 5 * math.sqrt(2 / n) and ks < 1.95 / math.sqrt(n)
    assert abs(np.corrcoef(zs[0], zs[1])[0, 1]) < 5 / math.sqrt(n)          # offset = call counter: a fresh stream per step
    # (2) final step under DDIM
    m2 = TSPModel(_args("gaussian", K, trick="ddim", L=Lyr), p, device=dev, seed=78)
    a, b, c, d, branch = (float(v) for v in m2.diffusion.posterior_constants(1, 0, "ddim"))
    assert branch == 1.0 and d == 0.0
    out, eps = m2.gaussian_denoise_step(pts, xt, np.array([1]), dev, ei, target_t=np.array([0]), return_aux=True)
    want = np.float32(a) * (xt.cpu().numpy() - np.float32(b) * eps.cpu().numpy())
    assert torch.isfinite(out).all()
    np.testing.assert_array_equal(out.cpu().numpy(), want.astype(np.float32))


# ------------------------------------------------------------------------------------------------
# full-depth oracle steps on the default engine
# ------------------------------------------------------------------------------------------------
def test_full_depth_oracle_step_tsp500(dev):
    """ONE TSP-500 / K = 50 graph (BASELINE configs[1]'s instance shape), H = 256, 12 layers, one teacher-forced categorical
    step on the default engine (fused, fp16x3, prepared state) against the CPU oracle."""
    from difusco_amd import TSPModel
    H, Lyr, N, K = 256, 12, 500, 50
    p = O.init_params(H, Lyr, 2, seed=20240926)
    pts, ei = O.tsp_instance(N, K, seed=1234)
    pts, ei = torch.from_numpy(pts), torch.from_numpy(ei)
    g = torch.Generator().manual_seed(21)
    xt = (torch.randn(ei.shape[1], generator=g) > 0).float()
    u = torch.rand(ei.shape[1], generator=g)
    t, tt = 969, 938
    ref_out, ref_logits, ref_prob = O.tsp_categorical_denoise_step(p, O.CategoricalTables(), pts, xt, t, ei, tt, uniform=u,
                                                                   return_aux=True)
    m = TSPModel(_args("categorical", K), p, device=dev)
    out, lg, pr = m.categorical_denoise_step(pts.to(dev), xt.to(dev), np.array([t]), dev, ei.to(dev), target_t=np.array([tt]),
                                             uniform=u, return_aux=True)
    e_log, e_prob = (lg.cpu() - ref_logits).abs().max().item(), (pr.cpu() - ref_prob.reshape(-1)).abs().max().item()
    print(f"TSP-500 K=50 H=256 L=12 vs oracle: logits L_inf {e_log:.3e}, prob L_inf {e_prob:.3e}")
    assert e_log < CLASS_TOL and e_prob < CLASS_TOL
    safe = (u - ref_prob.reshape(-1)).abs() > 1e-5
    assert torch.equal(out.cpu()[safe], ref_out[safe])


def test_full_depth_oracle_step_mis_er750(dev):
    """ONE Erdos-Renyi graph n = 750, p = 0.15 (+ reverse edges + self loops, ``mis_dataset.py:43-48``; BASELINE configs[3]'s
    instance shape), H = 256, 12 layers, one teacher-forced categorical step on the default engine against the oracle."""
    from difusco_amd import MISModel
    from difusco_amd.synthetic import er_mis_edge_index
    H, Lyr, n = 256, 12, 750
    p = O.init_params(H, Lyr, 2, seed=20240926)
    ei = torch.from_numpy(er_mis_edge_index(n, 0.15, seed=4321))
    g = torch.Generator().manual_seed(22)
    xt = (torch.randn(n, generator=g) > 0).float()
    u = torch.rand(n, generator=g)
    t, tt = 500, 469
    ref_out, ref_logits, ref_prob = O.mis_categorical_denoise_step(p, O.CategoricalTables(), xt, t, ei, tt, uniform=u,
                                                                   return_aux=True)
    m = MISModel(_args("categorical", -1), p, device=dev)
    out, lg, pr = m.categorical_denoise_step(xt.to(dev), np.array([t]), dev, ei.to(dev), target_t=np.array([tt]), uniform=u,
                                             return_aux=True)
    e_log, e_prob = (lg.cpu() - ref_logits).abs().max().item(), (pr.cpu() - ref_prob.reshape(-1)).abs().max().item()
    print(f"MIS ER-750 ({ei.shape[1]} edges) H=256 L=12 vs oracle: logits L_inf {e_log:.3e}, prob L_inf {e_prob:.3e}")
    assert e_log < CLASS_TOL and e_prob < CLASS_TOL
    safe = (u - ref_prob.reshape(-1)).abs() > 1e-5
    assert torch.equal(out.cpu()[safe], ref_out[safe])


# ------------------------------------------------------------------------------------------------
# call shapes at the edges of the size range
# ------------------------------------------------------------------------------------------------
def test_tsp1000_x64_single_call(dev):
    """BASELINE configs[2]'s WHOLE batch in one call on one GPU: 64 x TSP-1000 / K = 100 = 6.4 M edges (6.6 GB of edge state):
    bitwise determinism, {0,1} outputs, fused == unfused kernel sequence."""
    from difusco_amd import TSPModel
    from difusco_amd.synthetic import tsp_batch_gpu
    H, Lyr, N, K, G = 256, 12, 1000, 100, 64
    p = O.init_params(H, Lyr, 2, seed=20240926)
    pts, ei = tsp_batch_gpu(N, K, range(G), dev)
    E = ei.shape[1]
    assert E == 6_400_000
    g = torch.Generator().manual_seed(31)
    xt = (torch.randn(E, generator=g) > 0).float().to(dev)
    u = torch.rand(E, generator=g)
    t, tt = np.array([500]), np.array([469])
    mf = TSPModel(_args("categorical", K), p, device=dev)
    a, la, pa = mf.categorical_denoise_step(pts, xt, t, dev, ei, target_t=tt, uniform=u, return_aux=True)
    b, lb, pb = mf.categorical_denoise_step(pts, xt, t, dev, ei, target_t=tt, uniform=u, return_aux=True)
    assert torch.equal(a, b) and torch.equal(la, lb) and torch.isfinite(la).all() and set(a.unique().tolist()) <= {0.0, 1.0}
    del b, lb, pb
    mu = TSPModel(_args("categorical", K), p, device=dev, fused=False)
    c, lc, pc = mu.categorical_denoise_step(pts, xt, t, dev, ei, target_t=tt, uniform=u, return_aux=True)
    e_fu = (la - lc).abs().max().item()
    print(f"64 x TSP-1000 in one call (E = {E}): fused vs unfused logits L_inf {e_fu:.3e}, prob L_inf {(pa - pc).abs().max().item():.3e}")
    assert e_fu < 5e-5 and (pa - pc).abs().max().item() < 5e-5
    safe = ((u.to(dev) - pa).abs() > max(1e-5, (pa - pc).abs().max().item()))      # (fused vs unfused, bounded by 5e-5: the band follows the observed difference)
    assert torch.equal(a[safe], c[safe])


def test_call_with_more_than_2_pow_20_nodes(dev):
    """n_nodes >= 2^20 in ONE call: the full-line neighbour-table gathers of the fused kernel address node rows by 32-bit byte
    offsets (4 KB per row), so the step driver switches to the register-gather instantiation (64-bit addresses) - round 3
    returned DIFUSCO_EUNSUPPORTED here.  256 graphs of 4,100 nodes (1,049,600 nodes, K = 4, E = 4.2 M), 2 layers: bitwise
    determinism, fused == unfused, and in particular on the rows of the LAST graph (node ids above 2^20 - 4,100 ... 2^20 + 1,024)."""
    from difusco_amd import TSPModel
    H, Lyr, NB, G, K = 256, 2, 4100, 256, 4
    N = NB * G
    assert N >= (1 << 20)
    p = O.init_params(H, Lyr, 2, seed=9)
    rng = np.random.default_rng(17)
    pts = torch.from_numpy(rng.random((N, 2)).astype(np.float32)).to(dev)
    loc = np.arange(NB, dtype=np.int64)
    cols = np.stack([loc, (loc + 1) % NB, (loc + 7) % NB, (loc + 64) % NB], axis=1)                # self first, row-sorted
    ei_one = np.stack([np.repeat(loc, K), cols.reshape(-1)], axis=0)
    ei = torch.from_numpy(np.concatenate([ei_one + k * NB for k in range(G)], axis=1)).to(dev)
    E = ei.shape[1]
    g = torch.Generator().manual_seed(32)
    xt = (torch.randn(E, generator=g) > 0).float().to(dev)
    u = torch.rand(E, generator=g)
    t, tt = np.array([500]), np.array([469])
    mf = TSPModel(_args("categorical", K, L=Lyr), p, device=dev)
    a, la, pa = mf.categorical_denoise_step(pts, xt, t, dev, ei, target_t=tt, uniform=u, return_aux=True)
    b, lb, _ = mf.categorical_denoise_step(pts, xt, t, dev, ei, target_t=tt, uniform=u, return_aux=True)
    assert torch.equal(a, b) and torch.equal(la, lb) and torch.isfinite(la).all()
    mu = TSPModel(_args("categorical", K, L=Lyr), p, device=dev, fused=False)
    c, lc, pc = mu.categorical_denoise_step(pts, xt, t, dev, ei, target_t=tt, uniform=u, return_aux=True)
    e_all = (la - lc).abs().max().item()
    e_last = (la[-NB * K:] - lc[-NB * K:]).abs().max().item()
    print(f"N = {N} (>= 2^20), E = {E}: fused (register gathers) vs unfused logits L_inf {e_all:.3e}; last graph {e_last:.3e}")
    assert e_all < 5e-5 and (pa - pc).abs().max().item() < 5e-5
    # the stateless step (no prepared state) takes the same kernels
    ms = TSPModel(_args("categorical", K, L=Lyr), p, device=dev, prepare=False)
    d, ld, _ = ms.categorical_denoise_step(pts, xt, t, dev, ei, target_t=tt, uniform=u, return_aux=True)
    assert torch.equal(a, d) and torch.equal(la, ld)


# ------------------------------------------------------------------------------------------------
# prepared state
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("backend", ["ctypes", "torch"])
@pytest.mark.parametrize("kind", ["categorical", "gaussian"])
def test_prepared_state_is_bit_identical_to_the_stateless_step(dev, kind, backend):
    """ABI 9: ``difusco_prepare`` (node embedding, layer 0's node linear, the two-row edge-input table) and
    ``difusco_time_bias_rows`` (the time MLP of every schedule step in one launch) move step-invariant work out of the
    50-step loop - same kernels, same operands, so every output bit must equal the stateless step's, on both bindings,
    for several steps of a chain, and for a graph whose nodes were renumbered (Morton order) as well as one that was not."""
    from difusco_amd import TSPModel
    from difusco_amd.schedules import InferenceSchedule
    from difusco_amd.synthetic import tsp_batch
    H, Lyr, N, K, G = 256, 4, 300, 20, 3
    C = 2 if kind == "categorical" else 1
    p = O.init_params(H, Lyr, C, seed=41)
    pts, ei = tsp_batch(N, K, range(G), device=dev)
    E = ei.shape[1]
    sched = InferenceSchedule("cosine", T=1000, inference_T=50)
    for reorder in (True, False):
        ma = TSPModel(_args(kind, K, L=Lyr), p, device=dev, seed=5, backend=backend, reorder_nodes=reorder)                  # prepared (default)
        mb = TSPModel(_args(kind, K, L=Lyr), p, device=dev, seed=5, backend=backend, reorder_nodes=reorder, prepare=False)   # stateless
        ma.prepare_schedule([sched(i)[0] for i in range(50)])
        assert len(ma.model._tbias) == len({int(sched(i)[0]) for i in range(50)}) and not mb.model._tbias
        g = torch.Generator().manual_seed(6)
        x0 = torch.randn(E, generator=g)
        xa = xb = (x0 if kind == "gaussian" else (x0 > 0).float()).to(dev)
        for i in (0, 1, 2, 25, 48, 49):
            t1, t2 = (np.array([v]) for v in sched(i))
            if kind == "categorical":
                ra = ma.categorical_denoise_step(pts, xa, t1, dev, ei, target_t=t2, return_aux=True)
                rb = mb.categorical_denoise_step(pts, xb, t1, dev, ei, target_t=t2, return_aux=True)
            else:
                ra = ma.gaussian_denoise_step(pts, xa, t1, dev, ei, target_t=t2, return_aux=True)
                rb = mb.gaussian_denoise_step(pts, xb, t1, dev, ei, target_t=t2, return_aux=True)
            for va, vb in zip(ra, rb):
                assert torch.equal(va, vb), (kind, backend, reorder, i)
            xa, xb = ra[0], rb[0]
        assert len(ma._prep_cache) == 1 and not mb._prep_cache
    # a time outside the prepared schedule falls back to the in-step time MLP (same bits as a prepared row)
    ta = np.array([777])
    r1 = ma.categorical_denoise_step(pts, xa, ta, dev, ei, target_t=ta - 30, return_aux=True) if kind == "categorical" else \
        ma.gaussian_denoise_step(pts, xa, ta, dev, ei, target_t=ta - 30, return_aux=True)
    ma.prepare_schedule([777])
    ma.model.calls -= 1          # same Philox offset for the repeated call
    r2 = ma.categorical_denoise_step(pts, xa, ta, dev, ei, target_t=ta - 30, return_aux=True) if kind == "categorical" else \
        ma.gaussian_denoise_step(pts, xa, ta, dev, ei, target_t=ta - 30, return_aux=True)
    for va, vb in zip(r1, r2):
        assert torch.equal(va, vb)


def test_time_bias_rows_match_the_oracle(dev, L):
    """``difusco_time_bias_rows`` (one launch for a whole schedule) against the oracle's time MLP
    (``nn.py:103-121``, ``gnn_encoder.py:311-315,329-337``) for 70 times (> one 64-time launch)."""
    from difusco_amd import weights
    H, Lyr = 256, 5
    p = O.init_params(H, Lyr, 2, seed=3)
    blob = weights.pack_state_dict(p).to(dev)
    ts = [float(t) for t in range(1, 1001, 15)] + [1000.0, 969.0, 2.0]
    arr = (ctypes.c_float * len(ts))(*ts)
    out = torch.empty(len(ts), Lyr, H, device=dev)
    L.check(L.lib().difusco_time_bias_rows(H, Lyr, 2, _p(blob), arr, len(ts), _p(out), _stream()))
    torch.cuda.synchronize()
    for i, t in enumerate(ts):
        te = O.time_features(p, torch.tensor([t]), H)
        for l in range(Lyr):
            ref = O._layer_time_bias(p, l, te)
            err = (out[i, l].cpu() - ref.reshape(-1)).abs().max().item()
            assert err < 2e-5, (t, l, err)


# ------------------------------------------------------------------------------------------------
# (f)-4: no silent host path
# ------------------------------------------------------------------------------------------------
def test_mcts_heatmap_default_is_the_gpu_path_for_any_input_dtype(dev, golden_dir, tmp_path):
    """``write_mcts_heatmap`` with the default ``use_gpu=None``: float64 heat / points are cast to float32 ON THE DEVICE (the
    reference's data flow is float32, ``pl_tsp_model.py:258-267``) and give the text of the float32 inputs - round 3 dropped
    to the host numpy sweeps for them (31 s at N = 10^4).  No positive value at all: IndexError like the reference."""
    import os
    from difusco_amd import formats
    z = np.load(os.path.join(golden_dir, "mcts_sparse_text_n1000_k50.npz"))
    n, prob = int(z["num_nodes"]), float(z["expected_valid_prob"])
    ref_text = bytes(z["text"]).decode()
    calls = []
    orig = formats.mcts_heatmap_rows
    formats.mcts_heatmap_rows = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        path = formats.write_mcts_heatmap(z["heat"].astype(np.float64), z["points"].astype(np.float64), n, str(tmp_path), 0,
                                          expected_valid_prob=prob, edge_index=z["edge_index"])
    finally:
        formats.mcts_heatmap_rows = orig
    assert not calls, "the default path must not run the host numpy sweeps"
    assert open(path).read() == ref_text
    far = np.array([[0.0, 0.0], [3.0, 0.0], [0.0, 3.0], [3.0, 3.0]], np.float32)
    diag = np.array([[0, 1, 2, 3], [0, 1, 2, 3]])
    for pr in (0.5, 0.0):
        with pytest.raises(IndexError):
            list(formats.mcts_heatmap_rows_gpu(np.full(4, -5.0, np.float32), diag, far, 4, pr, device=dev))


# ------------------------------------------------------------------------------------------------
# per-sample GroupNorm segments on the fused path (dense mode with several samples)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("V,B", [(50, 4), (37, 3), (100, 2)])
def test_dense_batch_through_the_fused_kernel_vs_oracle(dev, V, B):
    """Dense TSP (``dense_forward``, ``gnn_encoder.py:350-381``: GroupNorm statistics PER SAMPLE) with B samples in one call - the
    shape of TSP-50 / 100 with ``parallel_sampling > 1`` (``pl_tsp_model.py:178-192``).  Round 3 sent every call with more than one
    statistic segment down the unfused kernel sequence; now the fused layers run and only the head is segment aware (V^2 is not
    a multiple of 32: segments straddle tiles).  Default engine vs the oracle's dense encoder, vs the unfused sequence, and
    sample b of the batch vs the same sample alone (per-sample statistics make them equal)."""
    from difusco_amd import TSPModel
    H, Lyr = 256, 12
    p = O.init_params(H, Lyr, 2, seed=20240926)
    g = torch.Generator().manual_seed(V * 10 + B)
    pts = torch.rand(B, V, 2, generator=g)
    xt = (torch.randn(B, V, V, generator=g) > 0).float()
    u = torch.rand(B, V, V, generator=g)
    t, tt = 500, 469
    ref_out, ref_logits, ref_prob = O.tsp_categorical_denoise_step(p, O.CategoricalTables(), pts, xt, t, None, tt, uniform=u,
                                                                   return_aux=True)
    ref_logits = ref_logits.permute(0, 2, 3, 1) if ref_logits.shape[1] == 2 and ref_logits.dim() == 4 else ref_logits
    m = TSPModel(_args("categorical", -1), p, device=dev)
    out, lg, pr = m.categorical_denoise_step(pts.to(dev), xt.to(dev), np.array([t]), dev, None, target_t=np.array([tt]),
                                             uniform=u.reshape(-1), return_aux=True)
    e_log = (lg.cpu().reshape(ref_logits.shape) - ref_logits).abs().max().item()
    e_prob = (pr.cpu().reshape(-1) - ref_prob.reshape(-1)).abs().max().item()
    print(f"dense V={V} B={B} H=256 L=12 (fused layers, per-sample statistics) vs oracle: logits L_inf {e_log:.3e}, prob {e_prob:.3e}")
    assert e_log < CLASS_TOL and e_prob < CLASS_TOL
    safe = (u.reshape(-1) - ref_prob.reshape(-1)).abs() > 1e-5
    assert torch.equal(out.cpu().reshape(-1)[safe], ref_out.reshape(-1)[safe])
    mu = TSPModel(_args("categorical", -1), p, device=dev, fused=False)
    _, lu, pu = mu.categorical_denoise_step(pts.to(dev), xt.to(dev), np.array([t]), dev, None, target_t=np.array([tt]),
                                            uniform=u.reshape(-1), return_aux=True)
    assert (lg - lu).abs().max().item() < 5e-5 and (pr - pu).abs().max().item() < 5e-5
    for b in (0, B - 1):          # a sample alone (one segment: the round-3 fused path) == the same sample inside the batch
        _, l1, _ = m.categorical_denoise_step(pts[b:b + 1].to(dev), xt[b:b + 1].to(dev), np.array([t]), dev, None,
                                              target_t=np.array([tt]), uniform=u[b].reshape(-1), return_aux=True)
        e_b = (lg.reshape(B, V, V, 2)[b] - l1.reshape(V, V, 2)).abs().max().item()
        print(f"  sample {b} in the batch vs alone: logits L_inf {e_b:.3e}")
        assert e_b < 2e-5
