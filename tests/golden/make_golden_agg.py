#!/usr/bin/env python
"""Golden fixtures for ``--aggregation mean`` and ``--aggregation max`` (``train.py:52``; ``GNNLayer.aggregate``,
``gnn_encoder.py:144-191``) from the IMPORTED reference.

    python tests/golden/make_golden_agg.py      # rewrites tests/golden/tsp_dense_agg_*.npz

The DENSE branch of ``aggregate`` (:169-175) is pure torch - ``torch.sum(Vh, 2) / torch.sum(graph, 2)`` with
``graph = ones`` (:364), ``torch.max(Vh, 2)[0]`` - so these fixtures are tier B ("reference, import placeholders, none
executed"; asserted: the torch_sparse stand-in is never called).  The sparse branch needs torch_sparse.mean / max, which
do not exist here; their semantics (segment mean / segment max over the entries of a row, 0 for an empty row) are
restated in ``oracle.difusco_oracle.segment_aggregate`` and are the dense semantics on the complete graph, which is how
the product runs the dense mode (complete-graph CSR).

Same machinery and file format as make_golden_h256.py: weights from ``oracle.init_params(H, L, C, seed)`` loaded into the
reference with ``strict=True``, (seed, SHA-256) recorded; Bernoulli recorder for injected uniforms.
Two sizes per aggregation: H=64, L=2, B=2 samples of V=12 (general kernels) and H=256, L=3, B=1, V=16 (the fused layers for
"mean"; "max" runs the unfused sequence at every width).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import make_golden as MG                      # noqa: E402
from oracle import difusco_oracle as O        # noqa: E402


def main():
    calls = MG.install_placeholders()
    sys.path.insert(0, MG.REF)
    from pl_meta_model import COMetaModel
    from pl_tsp_model import TSPModel

    dev = torch.device("cpu")
    before = calls["torch_sparse"]
    for agg in ("mean", "max"):
        for (H, L, B, V, seed) in ((64, 2, 2, 12, 5100), (256, 3, 1, 16, 5200)):
            seed_cat, seed_gau = seed + (0 if agg == "mean" else 10), seed + (1 if agg == "mean" else 11)
            p_cat, p_gau = O.init_params(H, L, 2, seed=seed_cat), O.init_params(H, L, 1, seed=seed_gau)
            meta = dict(hidden=np.array(H), n_layers=np.array(L), seed_cat=np.array(seed_cat), seed_gau=np.array(seed_gau),
                        sha_cat=O.params_sha256(p_cat), sha_gau=O.params_sha256(p_gau), aggregation=np.array(agg))

            def ref_model(kind):
                args = MG.make_args(kind, -1, H=H, L=L)
                args.aggregation = agg
                torch.manual_seed(0)
                obj = TSPModel.__new__(TSPModel)
                COMetaModel.__init__(obj, param_args=args, node_feature_only=False)
                assert obj.model.layers[0].aggregation == agg
                obj.model.load_state_dict(p_cat if kind == "categorical" else p_gau, strict=True)
                obj.eval()
                return obj

            g = torch.Generator().manual_seed(seed + 7)
            pts = torch.from_numpy(np.random.default_rng(seed).random((B, V, 2))).float()
            fx = {"points": pts.numpy()}
            m = ref_model("categorical")
            xt = (torch.randn(B, V, V, generator=g) > 0).long()
            for si, (t, tt) in enumerate([(1000, 969), (400, 350), (1, 0)]):
                with MG.Recorder(seed + 20 + si) as rec:
                    with torch.no_grad():
                        logits = m.forward(pts, xt.float(), torch.tensor([float(t)]), None)
                    out = m.categorical_denoise_step(pts, xt, np.array([t]), dev, None, target_t=np.array([tt]))
                fx[f"cat{si}_t"], fx[f"cat{si}_xt"] = np.array([t, tt]), xt.numpy()
                fx[f"cat{si}_logits"], fx[f"cat{si}_out"] = logits.numpy(), out.numpy()
                if rec.prob is not None:
                    fx[f"cat{si}_prob"], fx[f"cat{si}_uniform"] = rec.prob.numpy(), rec.uniform.numpy()
                if tt > 0:
                    xt = out.long()
            m = ref_model("gaussian")
            xg = torch.randn(B, V, V, generator=g)
            for si, (t, tt) in enumerate([(1000, 969), (1, 0)]):
                with MG.Recorder(seed + 30 + si):
                    with torch.no_grad():
                        pred = m.forward(pts, xg, torch.tensor([float(t)]), None)
                    out = m.gaussian_denoise_step(pts, xg, np.array([t]), dev, None, target_t=np.array([tt]))
                fx[f"gau{si}_t"], fx[f"gau{si}_xt"] = np.array([t, tt]), xg.numpy()
                fx[f"gau{si}_pred"], fx[f"gau{si}_out"] = pred.numpy(), out.numpy()
                xg = out
            name = f"tsp_dense_agg_{agg}_h{H}_l{L}_b{B}.npz"
            np.savez_compressed(os.path.join(HERE, name), provenance=MG.TIER_B, **meta, **fx)
            print("wrote", name, {k: v.shape for k, v in fx.items() if k.endswith("logits")})
    assert calls["torch_sparse"] == before, "the dense path must not touch the torch_sparse stand-in"


if __name__ == "__main__":
    main()
