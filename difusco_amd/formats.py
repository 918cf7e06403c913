"""On-disk heatmap formats of the reference, host side only:

* ``save_numpy_heatmap``  - the ``.npy`` pair written by ``TSPModel.run_save_numpy_heatmap``
  (``difusco/pl_tsp_model.py:255-267``): ``{split}-heatmap-{idx}.npy`` and ``{split}-points-{idx}.npy`` under
  ``<dir>/numpy_heatmap``.
* ``mcts_heatmap_text``   - the text file the C++ MCTS tool reads (``tsp_mcts/convert_numpy_to_txt.py:18-72``,
  reader ``tsp_mcts/.../TSP_IO.h:461-492``): first line N, then N rows of ``%.6f`` after adding the distance prior,
  keeping the top ``expected_valid_prob`` share of the entries plus the 3 largest of every row, symmetrising and
  row-normalising.  The input is a DENSE N x N heatmap, as in the reference; ``densify`` turns the sparse (E-entry)
  heatmap of the k-NN models into one.

Plain numpy with the reference's dtypes and operation order, so the text is identical character for character."""
import os

import numpy as np


def densify(heat, edge_index, n_nodes: int) -> np.ndarray:
    """[E] heat values on the directed edges of ``edge_index`` -> dense [N,N] float32 (entries off the graph are 0)."""
    a = np.zeros((n_nodes, n_nodes), dtype=np.float32)
    ei = np.asarray(edge_index)
    a[ei[0], ei[1]] = np.asarray(heat, dtype=np.float32)
    return a


def save_numpy_heatmap(adj_mat, np_points, save_dir: str, real_batch_idx: int, split: str = "test"):
    """pl_tsp_model.py:255-267.  Returns the two paths."""
    heatmap_path = os.path.join(save_dir, "numpy_heatmap")
    os.makedirs(heatmap_path, exist_ok=True)
    hp = os.path.join(heatmap_path, f"{split}-heatmap-{real_batch_idx}.npy")
    pp = os.path.join(heatmap_path, f"{split}-points-{real_batch_idx}.npy")
    np.save(hp, np.asarray(adj_mat))
    np.save(pp, np.asarray(np_points))
    return hp, pp


def mcts_normalise(adj_matrix: np.ndarray, points: np.ndarray, num_nodes: int, expected_valid_prob: float = 0.02) -> np.ndarray:
    """convert_numpy_to_txt.py:21-47 on one instance."""
    dists = np.linalg.norm(points[:, None, :] - points[None, :, :], axis=-1)
    adj_matrix = adj_matrix + 0.01 * (1.0 - dists)
    adj_matrix[adj_matrix == np.inf] = 0.0
    expected_valid_value_num = int(num_nodes * num_nodes * expected_valid_prob)
    valid_values = adj_matrix[(adj_matrix > 0.0)]
    valid_values = np.sort(valid_values)
    valid_value_threshold = valid_values[-expected_valid_value_num]
    top3_nodes_per_node = np.argsort(adj_matrix, axis=1)[:, -3:]
    valid_mask = adj_matrix > valid_value_threshold
    top3_mask = np.zeros_like(adj_matrix, dtype=bool)
    top3_mask[np.arange(num_nodes)[:, None], top3_nodes_per_node] = True
    valid_mask = valid_mask | top3_mask
    adj_matrix = adj_matrix * valid_mask
    adj_matrix[adj_matrix != 0.0] += 1e-2
    adj_matrix = adj_matrix + adj_matrix.T
    adj_matrix = adj_matrix / adj_matrix.sum(axis=1, keepdims=True)
    return adj_matrix


def mcts_heatmap_text(adj_matrix: np.ndarray, points: np.ndarray, num_nodes: int, expected_valid_prob: float = 0.02) -> str:
    """convert_numpy_to_txt.py:57-71: the normalised matrix as text."""
    m = mcts_normalise(np.array(adj_matrix), np.asarray(points), num_nodes, expected_valid_prob)
    out = [f"{num_nodes}\n"]
    for row in range(num_nodes):
        out.append(" ".join([f"{x:.6f}" for x in m[row]]) + "\n")
    return "".join(out)


def write_mcts_heatmap(adj_matrix, points, num_nodes: int, output_dir: str, index: int, heatmap_prefix: str = "heatmap",
                       expected_valid_prob: float = 0.02) -> str:
    """File name and directory layout of convert_numpy_to_txt.py:60-66."""
    folder = f"{output_dir}/{heatmap_prefix}/tsp{num_nodes}"
    os.makedirs(folder, exist_ok=True)
    path = f"{folder}/heatmaptsp{num_nodes}_{index}.txt"
    with open(path, "w") as f:
        f.write(mcts_heatmap_text(adj_matrix, points, num_nodes, expected_valid_prob))
    return path
