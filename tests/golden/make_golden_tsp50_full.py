#!/usr/bin/env python
"""configs[0] of BASELINE.json at FULL width and FULL length, pure reference arithmetic (tier B): TSP-50 dense
categorical diffusion, batch 1, hidden_dim 256, 12 layers, all 50 steps of the cosine inference schedule - the loop of
``TSPModel.test_step`` (difusco/pl_tsp_model.py:185-222) driven through the IMPORTED reference.

    python tests/golden/make_golden_tsp50_full.py      # rewrites tests/golden/tsp50_dense_h256_l12_50steps.npz

Dense mode never touches torch_sparse (asserted: the stand-in's call counter does not move), so nothing in this
fixture is substitute code.  Weights: ``oracle.difusco_oracle.init_params(256, 12, 2, seed)`` loaded with
``load_state_dict(strict=True)`` (seed + SHA-256 recorded, not the 21 MB of weights).  Per step the fixture holds the
x_t that went in, the network output (logits, [1,2,50,50] as the reference returns them), the argument of
``torch.bernoulli`` (captured by wrapping it - the reference source is unmodified), the injected uniforms and the
sampled x_{t-1}; the chain is the reference's own free-running chain (each step consumes the previous step's sample).
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

H, L, V, STEPS, SEED = 256, 12, 50, 50, 5050


def main():
    calls = MG.install_placeholders()
    sys.path.insert(0, MG.REF)
    from pl_meta_model import COMetaModel
    from pl_tsp_model import TSPModel
    from utils.diffusion_schedulers import InferenceSchedule

    dev = torch.device("cpu")
    p = O.init_params(H, L, 2, seed=SEED)
    args = MG.make_args("categorical", -1, H=H, L=L)
    torch.manual_seed(0)
    m = TSPModel.__new__(TSPModel)
    COMetaModel.__init__(m, param_args=args, node_feature_only=False)
    m.model.load_state_dict(p, strict=True)
    m.eval()

    before = calls["torch_sparse"]
    pts = torch.from_numpy(np.random.default_rng(50).random((1, V, 2))).float()   # data/generate_tsp_data.py:44 distribution
    g = torch.Generator().manual_seed(51)
    xt = (torch.randn(1, V, V, generator=g) > 0).long()                          # pl_tsp_model.py:189-195
    sched = InferenceSchedule(inference_schedule="cosine", T=1000, inference_T=STEPS)
    fx = {"points": pts.numpy(), "hidden": np.array(H), "n_layers": np.array(L), "seed": np.array(SEED),
          "sha": O.params_sha256(p), "steps": np.array(STEPS)}
    ts, xin, logit, prob, uni, outs = [], [], [], [], [], []
    for i in range(STEPS):
        t1, t2 = sched(i)
        t1, t2 = np.array([t1]).astype(int), np.array([t2]).astype(int)
        with MG.Recorder(9000 + i) as rec:
            with torch.no_grad():
                lg = m.forward(pts, xt.float(), torch.from_numpy(t1).float(), None)
            out = m.categorical_denoise_step(pts, xt, t1, dev, None, target_t=t2)
        ts.append([int(t1[0]), int(t2[0])])
        xin.append(xt.numpy().astype(np.int8))
        logit.append(lg.numpy())
        outs.append(out.numpy().astype(np.float32))
        if int(t2[0]) > 0:
            prob.append(rec.prob.numpy())
            uni.append(rec.uniform.numpy())
            xt = out.long()
        else:
            assert i == STEPS - 1
    assert calls["torch_sparse"] == before, "dense path must not touch the torch_sparse stand-in"
    fx.update(t=np.array(ts), xt_in=np.stack(xin), logits=np.stack(logit), prob=np.stack(prob), uniform=np.stack(uni),
              out=np.stack(outs))
    path = os.path.join(HERE, "tsp50_dense_h256_l12_50steps.npz")
    np.savez_compressed(path, provenance=MG.TIER_B, **fx)
    print(path, os.path.getsize(path), "bytes;", "final heatmap range", float(outs[-1].min()), float(outs[-1].max()))


if __name__ == "__main__":
    main()
