"""GPU parity tests added in round 6 (``-m gpu``), VERDICT r5 "next" #3.

* TRAINED-NETWORK STATISTICS.  Every earlier adversarial case multiplies whole parameter groups by one power of two; a trained
  checkpoint looks different: a few outlier channels in the LayerNorm affines, a residual stream whose norm grows layer by layer,
  heavy-tailed weight matrices.  Three such classes x the three step kinds, default engine (fused kernel, fp16x3 planes with per-tile /
  per-matrix power-of-two operand scales).  The bound is CALIBRATED against the same network evaluated in float64
  (``oracle.encoder_sparse_f64``): where the fp32 oracle itself sits further than 1e-5 from the float64 value (two faithful fp32
  evaluations of such a network do not agree to 1e-5), the HIP path must be no further from float64 than 4 x the oracle is;
  everywhere else it must meet the 1e-5 class against the fp32 oracle directly.
* ``edge_embed_tiled_kernel`` through its own C entry (``difusco_edge_embed``): partial tiles and partial workgroups
  (E in {31, 33, 127, 129, 4099}), permuted inputs, |x_t| up to 6, at the 1e-5 class (reference: gnn_encoder.py:230-249, :304, :395).
* the tie band of the sampled-bit comparison follows the bound that is asserted.
"""
import ctypes
import math

import numpy as np
import pytest
import torch

from oracle import difusco_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-4          # north_star's bound
CLASS_TOL = 1e-5    # the default engine's class


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: torch.cuda.is_available() is False")
    return torch.device("cuda:0")


def _args(kind, sparse_factor=8, trick="ddim", H=256, L=12):
    return dict(diffusion_type=kind, diffusion_schedule="linear", diffusion_steps=1000, sparse_factor=sparse_factor,
                n_layers=L, hidden_dim=H, inference_trick=trick)


def _student_t(shape, nu, g):
    z = torch.randn(shape, generator=g)
    chi = torch.randn((nu,) + tuple(shape), generator=g).pow(2).sum(0) / nu
    return z / chi.sqrt()


def trained_like_params(kind, H, L, C, seed):
    """Random-init parameters bent towards what training produces (difusco/models/gnn_encoder.py:58-65 LayerNorm affines,
    :339-347 per_layer_out; pl_tsp_model.py:140-151 consumes whatever the checkpoint holds)."""
    g = torch.Generator().manual_seed(1000 + seed)
    p = {k: v.clone() for k, v in O.init_params(H, L, C, seed=seed).items()}
    if kind == "outlier_channels":       # 4 of 256 features: gains x 64, biases x 16, in both LayerNorms of every layer
        for l in range(L):
            idx = torch.randperm(H, generator=g)[:4]
            for nm in (f"layers.{l}.norm_e", f"per_layer_out.{l}.0"):
                p[nm + ".weight"][idx] *= 64.0
                p[nm + ".bias"][idx] *= 16.0
    elif kind == "growing_residual":     # per_layer_out[l][2] x 4^(l+1): the rms of e grows ~4x per layer (2 -> 7.6e6 over 12 layers)
        for l in range(L):
            p[f"per_layer_out.{l}.2.weight"] *= 4.0 ** (l + 1)
            p[f"per_layer_out.{l}.2.bias"] *= 4.0 ** (l + 1)
    elif kind == "heavy_tailed":         # Student-t (nu = 3) entries for C and per_layer_out[l][2], same variance scale as the default init
        for l in range(L):
            for nm in (f"layers.{l}.C", f"per_layer_out.{l}.2"):
                p[nm + ".weight"] = _student_t((H, H), 3, g) / math.sqrt(3.0 * H)
    else:
        raise ValueError(kind)
    return p


@pytest.mark.parametrize("weights_kind", ["outlier_channels", "growing_residual", "heavy_tailed"])
@pytest.mark.parametrize("step", ["tsp_categorical", "tsp_gaussian", "mis_categorical"])
def test_default_engine_on_trained_like_weights(dev, step, weights_kind):
    from difusco_amd import MISModel, TSPModel
    H, L = 256, 12
    C = 1 if step == "tsp_gaussian" else 2
    p = trained_like_params(weights_kind, H, L, C, seed=91)
    g = torch.Generator().manual_seed(17)
    t, tt = 500, 469
    tvec = torch.tensor([float(t)])
    if step == "mis_categorical":
        ei = torch.from_numpy(O.er_mis_instance(300, 0.06, seed=5))
        xt = (torch.randn(300, generator=g) > 0).float()
        u = torch.rand(300, generator=g)
        ref_x, ref, ref_prob = O.mis_categorical_denoise_step(p, O.CategoricalTables(), xt, t, ei, tt, uniform=u, return_aux=True)
        truth = O.encoder_sparse_f64(p, None, xt, tvec, ei, node_feature_only=True)
        m = MISModel(_args("categorical", -1), p, device=dev)
        out_x, out, prob = m.categorical_denoise_step(xt.to(dev), np.array([t]), dev, ei.to(dev), target_t=np.array([tt]), uniform=u,
                                                      return_aux=True)
    else:
        pts, ei = O.tsp_instance(150, 20, seed=6)      # 3,000 edges: 94 tiles, the last one partial (24 of 32 edges)
        pts, ei = torch.from_numpy(pts), torch.from_numpy(ei)
        if step == "tsp_categorical":
            xt = (torch.randn(ei.shape[1], generator=g) > 0).float()
            u = torch.rand(ei.shape[1], generator=g)
            ref_x, ref, ref_prob = O.tsp_categorical_denoise_step(p, O.CategoricalTables(), pts, xt, t, ei, tt, uniform=u, return_aux=True)
            m = TSPModel(_args("categorical", 20), p, device=dev)
            out_x, out, prob = m.categorical_denoise_step(pts.to(dev), xt.to(dev), np.array([t]), dev, ei.to(dev), target_t=np.array([tt]),
                                                          uniform=u, return_aux=True)
        else:
            xt = torch.randn(ei.shape[1], generator=g)
            u = None
            ref_x, ref = O.tsp_gaussian_denoise_step(p, O.GaussianTables(), pts, xt, t, ei, tt, return_aux=True)[:2]
            m = TSPModel(_args("gaussian", 20), p, device=dev)
            out_x, out = m.gaussian_denoise_step(pts.to(dev), xt.to(dev), np.array([t]), dev, ei.to(dev), target_t=np.array([tt]),
                                                 return_aux=True)
            prob = ref_prob = None
        truth = O.encoder_sparse_f64(p, pts, xt, tvec, ei)
    assert torch.isfinite(out).all()
    got = out.cpu().reshape(ref.shape)
    e_hip_ref = (got - ref).abs().max().item()
    e_ref_true = (ref.double() - truth.reshape(ref.shape)).abs().max().item()
    e_hip_true = (got.double() - truth.reshape(ref.shape)).abs().max().item()
    print(f"{step} {weights_kind}: |out| max {ref.abs().max().item():.2e}; HIP vs fp32 oracle {e_hip_ref:.2e}; fp32 oracle vs float64 "
          f"{e_ref_true:.2e}; HIP vs float64 {e_hip_true:.2e}")
    # the class bound, calibrated: against float64 the HIP path may sit at the 1e-5 class or at 4 x the fp32 oracle's own distance
    assert e_hip_true < max(CLASS_TOL, 4.0 * e_ref_true), (e_hip_true, e_ref_true)
    if e_ref_true < CLASS_TOL / 3:      # the fp32 oracle is a usable arbiter at the class: direct comparison
        assert e_hip_ref < CLASS_TOL, e_hip_ref
        if prob is not None:
            e_prob = (prob.cpu().reshape(-1) - ref_prob.reshape(-1)).abs().max().item()
            assert e_prob < CLASS_TOL, e_prob
            safe = (u - ref_prob.reshape(-1)).abs() > CLASS_TOL      # tie band = the bound asserted on prob
            assert torch.equal(out_x.cpu().reshape(-1)[safe], ref_x.reshape(-1)[safe])
        else:
            assert (out_x.cpu().reshape(-1) - ref_x.reshape(-1)).abs().max().item() < CLASS_TOL


def _gen_table(blob, H, Lyr, C, dev):
    from difusco_amd import _lib
    L = _lib.lib()
    need = L.difusco_gen_table_bytes(H)
    assert need >= 71 * 256 * 4
    tab = torch.empty(need // 4, dtype=torch.float32, device=dev)
    _lib.check(L.difusco_gen_table_build(H, Lyr, C, ctypes.c_void_p(blob.data_ptr()), ctypes.c_void_p(tab.data_ptr()), need,
                                         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return tab


@pytest.mark.parametrize("path,precision,bound", [("gemm", "fp16x3", CLASS_TOL), ("gemm", "bf16x3", TOL), ("table", "fp16x3", 3e-6)])
@pytest.mark.parametrize("E", [31, 33, 127, 129, 4099])
def test_edge_embed_kernel_partial_tiles_permuted(dev, E, path, precision, bound):
    """``difusco_edge_embed``: e0 = edge_embed(ScalarEmbeddingSine(x_t)) on E edges that do not fill a tile / a workgroup, inputs
    reached through a non-identity permutation, |x_t| up to 6 (Gaussian x_t lives in about +-5); the rows past E stay untouched and
    the per-tile maxima equal the maxima of what was written.  ``gemm``: the K = 256 contraction on generated sinusoid planes;
    ``table`` (round 6): degree-7 interpolation in the table of difusco_gen_table_build (LDS-resident, edge_embed_table_kernel) - held
    to 3e-6 against the fp32 oracle and, against the float64 curve, to the fp32 oracle's OWN distance from it (~1.7e-6: the rounding of
    x / dim_t and of the fp32 contraction; the interpolation itself contributes 2e-9)."""
    from difusco_amd import _lib, graph, weights
    H, Lyr, C = 256, 1, 1
    p = O.init_params(H, Lyr, C, seed=123)
    blob = weights.pack_state_dict(p).to(dev)
    tab = _gen_table(blob, H, Lyr, C, dev) if path == "table" else None
    g = torch.Generator().manual_seed(E)
    x_slot = (torch.rand(E, generator=g) * 12.0 - 6.0)
    x_slot[0], x_slot[-1] = 6.0, -6.0
    perm = torch.randperm(E, generator=g).to(torch.int32)        # CSR slot s reads caller index perm[s]
    x_caller = torch.empty(E)
    x_caller[perm.long()] = x_slot
    ref = O._lin(p, "edge_embed", O.scalar_embedding_sine(x_slot, H))
    truth = O._lin({k: v.double() for k, v in p.items()}, "edge_embed", O.scalar_embedding_sine(x_slot.double(), H))
    E_pad = (E + 255) // 256 * 256
    e_t = torch.full((E_pad * H,), 7.0, device=dev)
    tmax = torch.full((E_pad // 32,), -1.0, device=dev)
    L = _lib.lib()
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None      # noqa: E731
    xc, pm = x_caller.to(dev), perm.to(dev)
    _lib.check(L.difusco_edge_embed(H, Lyr, C, P(blob), _lib.PRECISIONS[precision], P(xc), P(pm), E, P(e_t), P(tmax), P(tab),
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    got = graph.from_tiled(e_t, E).cpu()
    err = (got - ref).abs().max().item()
    err64 = (got.double() - truth).abs().max().item()
    print(f"edge_embed {path} {precision} E={E}: L_inf vs fp32 oracle {err:.2e}, vs float64 {err64:.2e} (oracle vs float64 "
          f"{(ref.double() - truth).abs().max().item():.2e}; |e0| max {ref.abs().max().item():.2f})")
    assert err < bound, err
    if path == "table":
        assert err64 < 2.5e-6, err64      # (the fp32 class of this product: rows 1.3e-6, oracle 0.5 .. 1.7e-6 by seed)
    # rows past E: never written
    off = graph.edge_tiled_offsets(E_pad).to(dev)
    if E_pad > E:
        assert bool((e_t[off[E:].reshape(-1)] == 7.0).all())
    # tile maxima: tiles that hold edges report max |e0| of their rows; the pad tiles of the last workgroup report 0, later ones are untouched
    n_tiles, n_wg_tiles = (E + 31) // 32, (E + 127) // 128 * 4
    tm = tmax.cpu()
    for tl in range(n_tiles):
        rows = got[tl * 32:min(E, tl * 32 + 32)]
        assert abs(tm[tl].item() - rows.abs().max().item()) == 0.0
    assert bool((tm[n_tiles:n_wg_tiles] == 0.0).all()) and bool((tm[n_wg_tiles:] == -1.0).all())
    # identity permutation given as NULL: same bits
    e_t2 = torch.zeros_like(e_t)
    xs = x_slot.to(dev)
    _lib.check(L.difusco_edge_embed(H, Lyr, C, P(blob), _lib.PRECISIONS[precision], P(xs), None, E, P(e_t2), None, P(tab),
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    assert torch.equal(graph.from_tiled(e_t2, E).cpu(), got)


def test_edge_embed_table_range_and_fallback(dev):
    """The table covers -8 <= x < 8; a 32-edge tile with ANY edge outside it - or a non-finite x - is flagged by the table kernel and
    computed by the contraction kernel (which recomputes the 128-edge workgroup around it), the other tiles interpolate: every finite
    row stays at the 1e-5 class, rows of cells at the table's two ends (x = -8, x just below 8) included; a NaN input gives a NaN row
    (as in the reference: sin(nan))."""
    from difusco_amd import _lib, graph, weights
    H, Lyr, C = 256, 1, 1
    p = O.init_params(H, Lyr, C, seed=124)
    blob = weights.pack_state_dict(p).to(dev)
    tab = _gen_table(blob, H, Lyr, C, dev)
    E = 128 * 5
    g = torch.Generator().manual_seed(3)
    x = torch.rand(E, generator=g) * 16.0 - 8.0                     # workgroups 0, 1: inside
    x[0], x[1], x[2], x[3] = -8.0, 7.99, 0.0, float(np.float32(-1e-30))
    x[2 * 128 + 5] = 8.0                                            # workgroup 2: one edge AT the upper end (outside)
    x[2 * 128 + 6] = float(np.nextafter(np.float32(8.0), np.float32(0.0)))      # (x + 8 rounds to 16: also outside - either path is right)
    x[3 * 128 + 77] = -30.0                                         # workgroup 3: far outside
    x[4 * 128 + 1] = float("nan")                                   # workgroup 4: a NaN
    ref = O._lin(p, "edge_embed", O.scalar_embedding_sine(x, H))
    E_pad = (E + 255) // 256 * 256
    e_t = torch.zeros(E_pad * H, device=dev)
    L = _lib.lib()
    P = lambda t: ctypes.c_void_p(t.data_ptr())      # noqa: E731
    xd = x.to(dev)
    _lib.check(L.difusco_edge_embed(H, Lyr, C, P(blob), _lib.PRECISIONS["fp16x3"], P(xd), None, E, P(e_t), None, P(tab),
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    got = graph.from_tiled(e_t, E).cpu()
    fin = torch.isfinite(x)
    err = (got[fin] - ref[fin]).abs().max().item()
    print(f"edge_embed table + fallback: L_inf {err:.2e} over {int(fin.sum())} finite rows; inside workgroups {(got[:256] - ref[:256]).abs().max().item():.2e}")
    assert err < CLASS_TOL and (got[:256] - ref[:256]).abs().max().item() < 3e-6
    assert bool(torch.isnan(got[4 * 128 + 1]).all())
    # table values themselves: rows r hold e0 at x = -8 + (r - 3) / 4, exact fp32 arithmetic
    R = 71
    rows = tab[:R * H].reshape(R, H).cpu()
    xs = -8.0 + (torch.arange(R, dtype=torch.float32) - 3.0) / 4.0
    want = O._lin(p, "edge_embed", O.scalar_embedding_sine(xs, H))
    assert (rows - want).abs().max().item() < 2e-6


def test_edge_embed_rejects_bad_arguments(dev):
    from difusco_amd import _lib
    L = _lib.lib()
    x = torch.zeros(64, device=dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr())      # noqa: E731
    assert L.difusco_edge_embed(128, 1, 1, P(x), 3, P(x), None, 64, P(x), None, None, None) < 0          # hidden != 256
    assert L.difusco_edge_embed(256, 1, 1, P(x), 0, P(x), None, 64, P(x), None, None, None) < 0          # fp32: no tiled kernel
    assert L.difusco_edge_embed(256, 1, 1, None, 3, P(x), None, 64, P(x), None, None, None) < 0
    assert L.difusco_edge_embed(256, 1, 1, P(x), 3, P(x), None, 0, P(x), None, None, None) == 0          # empty: nothing to do
    assert L.difusco_gen_table_bytes(128) == 0 and L.difusco_gen_table_build(128, 1, 1, P(x), P(x), 1 << 22, None) < 0
    assert L.difusco_gen_table_build(256, 1, 1, P(x), P(x), 16, None) < 0                                # buffer too small


@pytest.mark.parametrize("prep_precision", ["fp32", "bf16x6", "bf16x3"])
def test_prepared_buffer_is_in_the_fused_domain_whatever_precision_prepared_it(dev, prep_precision):
    """``difusco_prepare`` under FP32 / BF16X6 used to leave the reference's node rows in the buffer while the fused
    step reads them in its log2(e) domain with b_C folded in - a C caller mixing the two got wrong gates silently.  The buffer is now
    in the fused domain whatever precision prepared it: a default-engine step on such a buffer equals the stateless step to the
    rounding of the node linear."""
    from difusco_amd import TSPModel
    from difusco_amd.engine import DenoiseEngine
    p = O.init_params(256, 3, 2, seed=8)
    pts, ei = O.tsp_instance(200, 10, seed=3)
    pts, ei = torch.from_numpy(pts).to(dev), torch.from_numpy(ei).to(dev)
    g = torch.Generator().manual_seed(4)
    xt = (torch.randn(ei.shape[1], generator=g) > 0).float().to(dev)
    u = torch.rand(ei.shape[1], generator=g)
    m = TSPModel(_args("categorical", 10, L=3), p, device=dev, prepare=False, backend="ctypes")
    csr = m.prepare_graph(ei, pts.shape[0], points=pts)
    t, tt = 400, 380
    base = m.categorical_denoise_step(pts, xt, np.array([t]), dev, ei, target_t=np.array([tt]), uniform=u, return_aux=True)
    other = DenoiseEngine(p, device=dev, precision=prep_precision, backend="ctypes")
    buf = other.prepare(csr, pts, force=True)
    assert buf is not None
    from difusco_amd import _lib
    post = np.zeros(8, dtype=np.float32)
    post[:4] = m.diffusion.posterior_constants(t, tt)
    post[4] = 1.0
    out = m.model.step(csr, _lib.TASK_TSP, _lib.CATEGORICAL, xt, float(t), post, points=pts, xt_is_binary=True, rand=u.to(dev),
                       want_pred=True, want_prob=True, prepared=buf)
    err = (out[1] - base[1]).abs().max().item()
    print(f"prepared under {prep_precision}, stepped under fp16x3: logits L_inf vs the stateless step {err:.2e}")
    assert err < CLASS_TOL


def test_cached_state_is_ordered_across_streams(dev):
    """time-bias rows and the prepared buffer are produced asynchronously on the stream that is current at the miss; a
    step issued on ANOTHER stream waits for the entry's event (no stream handle in the cache keys).  Same bits as the stateless step."""
    from difusco_amd import TSPModel
    p = O.init_params(256, 2, 2, seed=9)
    pts, ei = O.tsp_instance(300, 12, seed=5)
    pts, ei = torch.from_numpy(pts).to(dev), torch.from_numpy(ei).to(dev)
    g = torch.Generator().manual_seed(6)
    xt = (torch.randn(ei.shape[1], generator=g) > 0).float().to(dev)
    u = torch.rand(ei.shape[1], generator=g)
    t, tt = 700, 650
    ref = TSPModel(_args("categorical", 12, L=2), p, device=dev, prepare=False).categorical_denoise_step(
        pts, xt, np.array([t]), dev, ei, target_t=np.array([tt]), uniform=u, return_aux=True)
    m = TSPModel(_args("categorical", 12, L=2), p, device=dev)
    sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(sa):
        m.prepare_schedule([t])
        a = m.categorical_denoise_step(pts, xt, np.array([t]), dev, ei, target_t=np.array([tt]), uniform=u, return_aux=True)      # builds the prepared buffer on sa
    with torch.cuda.stream(sb):      # no host synchronisation in between: the events order sb behind sa's launches
        b = m.categorical_denoise_step(pts, xt, np.array([t]), dev, ei, target_t=np.array([tt]), uniform=u, return_aux=True)
    torch.cuda.synchronize()
    assert len(m.model._tbias) == 1 and len(m._prep_cache) == 1
    ev = next(iter(m.model._tbias.values()))[1]
    assert ev.ordered == {sa.cuda_stream, sb.cuda_stream}
    assert torch.equal(a[1], ref[1]) and torch.equal(b[1], ref[1]) and torch.equal(a[0], ref[0]) and torch.equal(b[0], ref[0])
