#!/usr/bin/env python
"""Generates tests/golden/tsp_decode_*.npz with the REFERENCE's own decode: ``merge_tours``
(/root/reference/difusco/utils/tsp_utils.py:89-145) calling ``merge_cython`` compiled from
/root/reference/difusco/utils/cython_merge/cython_merge.pyx with this image's Cython (build products go to a
temporary directory; nothing of the reference is copied into the repo).  Runs in the build container only - the
GPU box has no /root/reference.  Fixtures are data: points, edge_index, heat values, and the reference's tours and
iteration counters."""
import importlib.util
import os
import subprocess
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference/difusco"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def load_reference():
    tmp = tempfile.mkdtemp(prefix="difusco_cymerge_")
    setup = os.path.join(tmp, "setup.py")
    with open(setup, "w") as f:
        f.write("from setuptools import setup, Extension\nfrom Cython.Build import cythonize\nimport numpy\n"
                f"setup(ext_modules=cythonize(Extension('cython_merge', ['{REF}/utils/cython_merge/cython_merge.pyx'],"
                "include_dirs=[numpy.get_include()]), language_level=3, build_dir='build'))\n")
    subprocess.check_call([sys.executable, setup, "build_ext", "--build-lib", tmp, "--build-temp", os.path.join(tmp, "bt")],
                          cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    so = [f for f in os.listdir(tmp) if f.startswith("cython_merge") and f.endswith(".so")][0]
    spec = importlib.util.spec_from_file_location("utils.cython_merge.cython_merge", os.path.join(tmp, so))
    cm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cm)
    for name in ("utils", "utils.cython_merge"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["utils.cython_merge.cython_merge"] = cm
    spec = importlib.util.spec_from_file_location("ref_tsp_utils", os.path.join(REF, "utils", "tsp_utils.py"))
    tu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tu)
    return tu


def heat_case(kind, E, rng, ei, pts):
    if kind == "bits":            # a non-final categorical step: {0,1} + 1e-6 (pl_tsp_model.py:222)
        d = np.linalg.norm(pts[ei[0]] - pts[ei[1]], axis=1)
        p = np.exp(-d / (0.6 * d.mean()))
        return ((rng.random(E) < p).astype(np.float32) + np.float32(1e-6)).astype(np.float32)
    if kind == "prob":            # the final step returns probabilities (clamped), + 1e-6
        d = np.linalg.norm(pts[ei[0]] - pts[ei[1]], axis=1)
        return (np.clip(np.exp(-d / (0.5 * d.mean())) * rng.random(E), 0, 1).astype(np.float32) + np.float32(1e-6))
    if kind == "gauss":           # Gaussian diffusion: x * 0.5 + 0.5, may leave [0,1] (pl_tsp_model.py:220)
        return (rng.standard_normal(E).astype(np.float32) * np.float32(0.5) + np.float32(0.5)) * np.float32(0.5) + np.float32(0.5)
    raise ValueError(kind)


def main():
    from difusco_amd.synthetic import tsp_instance
    tu = load_reference()
    # complete k-NN graphs (K = N-1): every pair is a candidate, the reference finishes before its zero block and the
    # whole tour is pinned; sparse graphs: the reference usually runs into its zero block (order = numpy's unstable
    # argsort, not reproducible) - there only the insertions made from positive entries are pinned.
    cases = [("n40_k39_bits", 40, 39, "bits", 1), ("n60_k59_prob_p2", 60, 59, "prob", 2), ("n50_k49_gauss", 50, 49, "gauss", 1),
             ("n50_k10_bits", 50, 10, "bits", 1), ("n200_k20_prob_p2", 200, 20, "prob", 2), ("n120_k12_gauss", 120, 12, "gauss", 1),
             ("n200_k60_prob", 200, 60, "prob", 1)]
    for name, n, k, kind, par in cases:
        rng = np.random.default_rng(sum(map(ord, name)))
        pts, ei = tsp_instance(n, k, seed=n + k)
        E = ei.shape[1]
        heat = np.concatenate([heat_case(kind, E, rng, ei, pts) for _ in range(par)]).astype(np.float32)
        with np.errstate(all="ignore"):
            tours, it = tu.merge_tours(heat, pts, ei, sparse_graph=True, parallel_sampling=par)
        # per sample: how many entries of the reference's dense sorted list have a negative key (= positive heat /
        # distance, incl. the -inf self entries)?  A sample whose walk is longer ran into the zero block.
        dist = np.linalg.norm(pts.astype(np.float64)[:, None] - pts.astype(np.float64), axis=-1)
        neg = []
        for part in np.split(heat, par):
            a = np.zeros((n, n), dtype=np.float32)
            a[ei[0], ei[1]] = part
            with np.errstate(all="ignore"):
                neg.append(int(((-(a + a.T).astype(np.float64) / dist) < 0).sum()))
        # merge_iterations is the MEAN over samples; per-sample counters are recovered by one call per sample
        per = []
        for part in np.split(heat, par):
            with np.errstate(all="ignore"):
                per.append(tu.merge_tours(part, pts, ei, sparse_graph=True, parallel_sampling=1)[1])
        completed = [bool(p <= m) for p, m in zip(per, neg)]
        np.savez_compressed(os.path.join(HERE, f"tsp_decode_{name}.npz"), points=pts, edge_index=ei, heat=heat,
                            parallel_sampling=par, tours=np.asarray(tours, dtype=np.int32), merge_iterations=np.float64(it),
                            merge_iterations_per_sample=np.asarray(per, dtype=np.float64),
                            negative_key_entries=np.asarray(neg, dtype=np.int64), completed=np.asarray(completed),
                            provenance="reference: tsp_utils.merge_tours + cython_merge.merge_cython (pyx compiled here)")
        print(name, "E", E, "merge_iterations", per, "negative-key entries", neg, "completed", completed)

    # ---- 2-opt: the reference's batched_two_opt_torch (tsp_utils.py:12-49) on the CPU, float64 points ----------
    two_opt_cases = [("n50_random", 50, 1, 1000, "random"), ("n120_merge_b3", 120, 3, 1000, "merge"),
                     ("n200_cap5_b2", 200, 2, 5, "random"), ("n40_converged", 40, 1, 1000, "converged"),
                     ("n300_merge", 300, 1, 1000, "merge")]
    for name, n, batch, max_it, kind in two_opt_cases:
        rng = np.random.default_rng(sum(map(ord, name)))
        pts32, ei = tsp_instance(n, min(20, n - 1), seed=7 * n)
        pts = pts32.astype(np.float64)
        if kind == "merge":
            heat = np.concatenate([heat_case("prob", ei.shape[1], rng, ei, pts32) for _ in range(batch)])
            with np.errstate(all="ignore"):
                tours0, _ = tu.merge_tours(heat, pts32, ei, sparse_graph=True, parallel_sampling=batch)
            tours0 = np.asarray(tours0, dtype=np.int64)
        else:
            tours0 = np.stack([np.concatenate([[0], 1 + rng.permutation(n - 1), [0]]) for _ in range(batch)]).astype(np.int64)
        if kind == "converged":
            tours0, _ = tu.batched_two_opt_torch(pts, tours0, max_iterations=10000, device="cpu")
        out, it = tu.batched_two_opt_torch(pts, tours0, max_iterations=max_it, device="cpu")
        np.savez_compressed(os.path.join(HERE, f"tsp_twoopt_{name}.npz"), points=pts, tours_in=tours0.astype(np.int32),
                            tours_out=np.asarray(out, dtype=np.int32), iterations=np.int64(it), max_iterations=np.int64(max_it),
                            provenance="reference: tsp_utils.batched_two_opt_torch (torch CPU, float64)")
        print(name, "2-opt iterations", it)


def mis_fixtures():
    """mis_decode_np of the reference (utils/mis_utils.py, imports unmodified) on Erdos-Renyi graphs in the dataset's
    layout (both directions + self loops) with continuous, tie-free scores."""
    import scipy.sparse
    from difusco_amd.synthetic import er_mis_edge_index
    spec = importlib.util.spec_from_file_location("ref_mis_utils", os.path.join(REF, "utils", "mis_utils.py"))
    mu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mu)
    for name, n, p, seed in [("n60_p15", 60, 0.15, 1), ("n300_p05", 300, 0.05, 2), ("n750_p15", 750, 0.15, 3)]:
        ei = er_mis_edge_index(n, p, seed)
        rng = np.random.default_rng(seed)
        pred = (rng.random(n).astype(np.float32) + np.float32(1e-6))
        adj = scipy.sparse.coo_matrix((np.ones_like(ei[0]), (ei[0], ei[1])))          # pl_mis_model.py:152-154
        sol = mu.mis_decode_np(pred, adj)
        assert len(np.unique(pred)) == n
        np.savez_compressed(os.path.join(HERE, f"mis_decode_{name}.npz"), edge_index=ei.astype(np.int32), predictions=pred,
                            solution=sol.astype(np.int8), provenance="reference: utils/mis_utils.mis_decode_np")
        print(name, "set size", int(sol.sum()))


if __name__ == "__main__":
    mis_fixtures()
    main()
