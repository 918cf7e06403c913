#!/usr/bin/env python
"""tests/golden/tsp_densemerge_*.npz: the DENSE branch of the reference's ``merge_tours``
(/root/reference/difusco/utils/tsp_utils.py:105-108, ``sparse_graph=False``: ``adj_mat[0] + adj_mat[0].T`` on the
[parallel_sampling, N, N] heatmap of the dense TSP-50/100 models, BASELINE configs[0]) followed by ``merge_cython``.
Same machinery as make_golden_decode.py (the .pyx is compiled into a temporary directory; build container only)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import make_golden_decode as MD      # noqa: E402


def main():
    tu = MD.load_reference()
    for name, n, par, kind in [("n50_prob", 50, 1, "prob"), ("n50_bits_p2", 50, 2, "bits"), ("n30_gauss", 30, 1, "gauss")]:
        rng = np.random.default_rng(sum(map(ord, name)))
        pts = rng.random((n, 2)).astype(np.float32)                   # data/generate_tsp_data.py:44
        d = np.linalg.norm(pts[:, None] - pts[None], axis=-1)
        heats = []
        for _ in range(par):
            if kind == "prob":        # final categorical step: probabilities, + 1e-6 (pl_tsp_model.py:222)
                h = np.clip(np.exp(-d / (0.5 * d.mean())) * rng.random((n, n)), 0, 1).astype(np.float32) + np.float32(1e-6)
            elif kind == "bits":
                h = (rng.random((n, n)) < np.exp(-d / (0.6 * d.mean()))).astype(np.float32) + np.float32(1e-6)
            else:                     # gaussian: x * 0.5 + 0.5 (:220)
                h = rng.standard_normal((n, n)).astype(np.float32) * np.float32(0.25) + np.float32(0.75)
            heats.append(h.astype(np.float32))
        heat = np.stack(heats)                                        # [parallel_sampling, N, N]
        with np.errstate(all="ignore"):
            tours, it = tu.merge_tours(heat, pts, None, sparse_graph=False, parallel_sampling=par)
            per = [tu.merge_tours(heat[s:s + 1], pts, None, sparse_graph=False, parallel_sampling=1)[1] for s in range(par)]
        dist = np.linalg.norm(pts.astype(np.float64)[:, None] - pts.astype(np.float64), axis=-1)
        with np.errstate(all="ignore"):
            neg = [int(((-(h + h.T).astype(np.float64) / dist) < 0).sum()) for h in heat]
        completed = [bool(p <= m) for p, m in zip(per, neg)]
        np.savez_compressed(os.path.join(HERE, f"tsp_densemerge_{name}.npz"), points=pts, heat=heat, parallel_sampling=par,
                            tours=np.asarray(tours, dtype=np.int32), merge_iterations=np.float64(it),
                            merge_iterations_per_sample=np.asarray(per, dtype=np.float64), completed=np.asarray(completed),
                            provenance="reference: tsp_utils.merge_tours(sparse_graph=False) + cython_merge.merge_cython")
        print(name, "merge_iterations", per, "completed", completed)


if __name__ == "__main__":
    main()
