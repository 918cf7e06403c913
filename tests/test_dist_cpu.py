"""CPU tests of the multi-GPU plumbing: graph sharding and the one-off weight broadcast, exercised with
world_size 2 over gloo (the GPU path uses the same code over RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from difusco_amd import synthetic
from difusco_amd.dist import broadcast_weights, gn_allreduce, shard_range
from oracle import difusco_oracle as O


def test_shard_range_partitions_everything():
    for n in (0, 1, 7, 8, 64, 65, 128):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_synthetic_generators_match_oracle():
    for (n, k, seed) in [(50, 6, 1), (200, 20, 7)]:
        p1, e1 = synthetic.tsp_instance(n, k, seed)
        p2, e2 = O.tsp_instance(n, k, seed)
        np.testing.assert_array_equal(p1, p2)
        np.testing.assert_array_equal(e1, e2)
    np.testing.assert_array_equal(synthetic.er_mis_edge_index(60, 0.2, 3), O.er_mis_instance(60, 0.2, 3))
    a, b = synthetic.random_state_dict(64, 2, 2, 5), O.init_params(64, 2, 2, 5)
    assert a.keys() == b.keys()
    for k in a:
        assert torch.equal(a[k], b[k])
    pts, ei = synthetic.tsp_batch(30, 5, [0, 1, 2])
    assert pts.shape == (90, 2) and ei.shape == (2, 450)
    assert ei[:, 150:300].min() >= 30 and ei[:, 150:300].max() < 60      # disjoint union, ids offset


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sd = synthetic.random_state_dict(64, 2, 2, seed=11) if rank == 0 else None
        cfg, blob = broadcast_weights(sd, torch.device("cpu"), src=0)
        lo, hi = shard_range(5, rank, world)
        # every rank builds ITS graphs only; no collective is needed afterwards
        pts, ei = synthetic.tsp_batch(20, 4, range(lo, hi))
        # the optional global-statistics exchange: 32 x (sum, sumsq) + row count, summed over the ranks in place
        sums = torch.arange(65, dtype=torch.float64) * (rank + 1)
        gn_allreduce()(sums)
        ret[rank] = (cfg, float(blob.double().sum()), blob.numel(), (lo, hi), tuple(pts.shape), tuple(ei.shape),
                     bool(torch.equal(sums, torch.arange(65, dtype=torch.float64) * 3)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_broadcast_and_shard_two_processes():
    from difusco_amd.weights import pack_state_dict
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
        ret = dict(ret)
    ref = pack_state_dict(synthetic.random_state_dict(64, 2, 2, seed=11))
    assert ret[0][0] == ret[1][0] == (64, 2, 2)
    assert ret[0][2] == ret[1][2] == ref.numel()
    assert ret[0][1] == ret[1][1] == float(ref.double().sum())
    assert ret[0][3] == (0, 3) and ret[1][3] == (3, 5)
    assert ret[0][4] == (60, 2) and ret[1][4] == (40, 2)
    assert ret[0][6] and ret[1][6]            # gn_allreduce summed the 65 doubles over both ranks


@pytest.mark.parametrize("gn_stats", ["per_shard_call", "global"])
def test_bench_py_multi_rank_plumbing_dry_run(gn_stats):
    """bench.py's world > 1 branch end to end, the way the driver launches it (``python -m torch.distributed.run
    --nproc-per-node 2 ... bench.py --gpus 2``), on CPU tensors over gloo with BENCH_PLUMBING_DRY_RUN=1: rendezvous,
    rank-0 packs and the blob is broadcast, graphs are sharded, the optional 65-double all-reduce runs every step, the
    barriers and the max-over-ranks timing complete, rank 0 prints ONE JSON line marked ``dry_run`` (no kernel ran)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_PLUMBING_DRY_RUN="1", MASTER_ADDR="127.0.0.1", BENCH_FULL_JSON=os.devnull)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--nodes", "40", "--knn", "6", "--graphs-per-gpu", "3", "--gn-stats", gn_stats]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    # the stdout line is the COMPACT record (a 20 KB line was not parsed by the driver): the last line, < 4 KB
    assert res.stdout.strip().splitlines()[-1] == lines[0] and len(lines[0]) < 4096
    out = json.loads(lines[0])
    assert "full record" in res.stderr      # the full record goes to stderr (and to bench_full.json)
    assert out["dry_run"] is True and out["metric"].startswith("PLUMBING DRY RUN")
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 6 and out["config"]["graphs_per_gpu"] == 3 and out["config"]["gn_stats"] == gn_stats
    assert out["config"]["nodes_rank0"] == 120 and out["config"]["edges_rank0"] == 720       # rank 0 holds 3 of the 6 graphs
    assert out["value"] > 0 and abs(out["value"] - 6 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]
    # evidence that the collective backend saw every rank: all-gathered world size, one loop time per rank
    assert out["ranks_seen"] == 2 and len(out["rank_ms_per_step"]) == 2 and all(v > 0 for v in out["rank_ms_per_step"])
    rep = out["repeats"]
    assert rep["n"] == 3 and len(rep["ms_per_step"]) == 3
    assert rep["min_ms_per_step"] <= rep["median_ms_per_step"] == out["ms_per_step"] <= rep["max_ms_per_step"]


def _run_bench(argv, env_extra=None, drop_env=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in drop_env}
    env.update(BENCH_PLUMBING_DRY_RUN="1", BENCH_FULL_JSON=os.devnull)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, capture_output=True, text=True, timeout=600,
                          env=env, cwd=root)


def test_bench_py_gpus_2_without_a_launcher_starts_two_ranks():
    """``python bench.py --gpus 2`` with no launcher (WORLD_SIZE unset) must run TWO ranks - bench.py re-runs
    itself under ``torch.distributed.run`` - and print the same keys as the launcher form, never an ``n_gpus: 1`` line."""
    import json
    argv = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--nodes", "40", "--knn", "6", "--graphs-per-gpu", "3"]
    res = _run_bench(argv)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and res.stdout.strip().splitlines()[-1] == lines[0] and len(lines[0]) < 4096, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and len(out["rank_ms_per_step"]) == 2 and out["dry_run"] is True
    assert out["scaling"] == "weak" and out["config"]["global_batch"] == 6 and out["config"]["graphs_per_gpu"] == 3
    assert "starting 2 ranks" in res.stderr
    # the launcher form of the same command: same keys, same configuration
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_PLUMBING_DRY_RUN="1", MASTER_ADDR="127.0.0.1", BENCH_FULL_JSON=os.devnull)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py")] + argv
    res2 = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert res2.returncode == 0, res2.stderr[-3000:]
    out2 = json.loads([l for l in res2.stdout.splitlines() if l.startswith("{")][0])
    assert sorted(out) == sorted(out2) and sorted(out["config"]) == sorted(out2["config"])
    assert out2["n_gpus"] == 2 and out2["config"] == out["config"]


def test_bench_py_refuses_a_rank_count_that_differs_from_the_command():
    """A launcher that started ONE rank for a ``--gpus 2`` command (or two for ``--gpus 1``) is an error, not a one-rank line."""
    for world, gpus in (("1", "2"), ("2", "1")):
        res = _run_bench(["--gpus", gpus, "--steps", "2", "--warmup", "1", "--nodes", "40", "--knn", "6"],
                         env_extra={"WORLD_SIZE": world, "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                                    "MASTER_PORT": str(_free_port())}, drop_env=())
        assert res.returncode != 0 and "refusing" in res.stderr, (res.returncode, res.stderr[-1000:])
        assert not [l for l in res.stdout.splitlines() if l.startswith("{")]
    # outside the dry run a --gpus larger than the visible device count is refused before anything is launched
    res = _run_bench(["--gpus", "2"], env_extra={"BENCH_PLUMBING_DRY_RUN": "0"})
    assert res.returncode != 0 and not res.stdout.strip()


def test_bench_py_strong_scaling_splits_the_literal_global_batch():
    """``--scaling strong``: BASELINE configs[2]'s batch (here --global-batch 5) is split over the ranks by shard_range (3 + 2),
    the line says ``strong`` and value counts the GLOBAL batch."""
    import json
    res = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--nodes", "40", "--knn", "6", "--scaling", "strong",
                      "--global-batch", "5"])
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][0])
    assert out["scaling"] == "strong" and out["n_gpus"] == 2 and out["ranks_seen"] == 2
    assert out["config"]["global_batch"] == 5 and out["config"]["graphs_per_gpu"] == 3 and out["config"]["nodes_rank0"] == 120
    assert abs(out["value"] - 5 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]
    # fewer graphs than ranks: refused
    res = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--nodes", "40", "--knn", "6", "--scaling", "strong",
                      "--global-batch", "1"])
    assert res.returncode != 0
