#!/usr/bin/env python
"""Generates tests/golden/knn_*.npz with the library call the reference's dataset makes
(co_datasets/tsp_graph_dataset.py:56-57): sklearn.neighbors.KDTree(points, leaf_size=30, metric='euclidean')
.query(points, k).  Fixtures are data (points, neighbour indices)."""
import os

import numpy as np
from sklearn.neighbors import KDTree

HERE = os.path.dirname(os.path.abspath(__file__))
for name, n, k, seed in [("n50_k10", 50, 10, 1), ("n500_k50", 500, 50, 2), ("n700_k40", 700, 40, 3), ("n64_k63", 64, 63, 4)]:
    pts = np.random.default_rng(seed).random((n, 2))
    kdt = KDTree(pts, leaf_size=30, metric="euclidean")
    dis, idx = kdt.query(pts, k=k, return_distance=True)
    np.savez_compressed(os.path.join(HERE, f"knn_{name}.npz"), points=pts, k=np.int64(k), idx_knn=idx.astype(np.int32),
                        provenance="sklearn.neighbors.KDTree(points, leaf_size=30, metric='euclidean').query(points, k) "
                                   "as in co_datasets/tsp_graph_dataset.py:56-57")
    print(name, idx.shape)
