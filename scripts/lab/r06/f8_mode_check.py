#!/usr/bin/env python
"""(branch fp8c only: needs the fp16f8 precision, which main does not have)  fp16f8 (DIFUSCO_PREC_FP16F8) against the CPU oracle and against fp16x3: one teacher-forced step per task on small graphs (H = 256, L = 12)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from difusco_amd import MISModel, TSPModel  # noqa: E402
from oracle import difusco_oracle as O  # noqa: E402

dev = torch.device("cuda:0")


def args(kind, k, L=12):
    return dict(diffusion_type=kind, diffusion_schedule="linear", diffusion_steps=1000, sparse_factor=k, n_layers=L, hidden_dim=256,
                inference_trick="ddim")


for seed in (77, 78):
    p = O.init_params(256, 12, 2, seed=seed)
    pts, ei = O.tsp_instance(300, 20, seed=seed)
    pts, ei = torch.from_numpy(pts), torch.from_numpy(ei)
    g = torch.Generator().manual_seed(seed)
    xt = (torch.randn(ei.shape[1], generator=g) > 0).float()
    u = torch.rand(ei.shape[1], generator=g)
    ref = O.tsp_categorical_denoise_step(p, O.CategoricalTables(), pts, xt, 500, ei, 469, uniform=u, return_aux=True)
    for prec in ("fp16x3", "bf16x3", "fp16f8"):
        m = TSPModel(args("categorical", 20), p, device=dev, precision=prec)
        out = m.categorical_denoise_step(pts.to(dev), xt.to(dev), np.array([500]), dev, ei.to(dev), target_t=np.array([469]), uniform=u, return_aux=True)
        print(f"TSP cat seed {seed} {prec}: logits L_inf {(out[1].cpu() - ref[1]).abs().max().item():.2e} prob {(out[2].cpu() - ref[2].reshape(-1)).abs().max().item():.2e}", flush=True)
p = O.init_params(256, 12, 1, seed=5)
pts, ei = O.tsp_instance(300, 20, seed=5)
pts, ei = torch.from_numpy(pts), torch.from_numpy(ei)
g = torch.Generator().manual_seed(5)
xt = torch.randn(ei.shape[1], generator=g)
ref = O.tsp_gaussian_denoise_step(p, O.GaussianTables(), pts, xt, 500, ei, 469, return_aux=True)
for prec in ("fp16x3", "fp16f8"):
    m = TSPModel(args("gaussian", 20), p, device=dev, precision=prec)
    out = m.gaussian_denoise_step(pts.to(dev), xt.to(dev), np.array([500]), dev, ei.to(dev), target_t=np.array([469]), return_aux=True)
    print(f"TSP gaussian {prec}: eps L_inf {(out[1].cpu() - ref[1].reshape(-1)).abs().max().item():.2e}", flush=True)
p = O.init_params(256, 12, 2, seed=6)
ei = torch.from_numpy(O.er_mis_instance(400, 0.05, seed=6))
g = torch.Generator().manual_seed(6)
xt = (torch.randn(400, generator=g) > 0).float()
u = torch.rand(400, generator=g)
ref = O.mis_categorical_denoise_step(p, O.CategoricalTables(), xt, 500, ei, 469, uniform=u, return_aux=True)
for prec in ("fp16x3", "fp16f8"):
    m = MISModel(args("categorical", -1), p, device=dev, precision=prec)
    out = m.categorical_denoise_step(xt.to(dev), np.array([500]), dev, ei.to(dev), target_t=np.array([469]), uniform=u, return_aux=True)
    print(f"MIS {prec}: logits L_inf {(out[1].cpu() - ref[1]).abs().max().item():.2e}", flush=True)
