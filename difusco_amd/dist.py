"""Multi-GPU layout of the sampler: one process per GPU, graphs sharded by rank, frozen weights
broadcast once over RCCL (``torch.distributed`` backend "nccl" on ROCm), no collective in the
denoising loop.

Graphs of a batch are independent through all GNN layers (disjoint union, ``pl_meta_model.py:177-184``);
the only coupling in the reference is the head GroupNorm, whose statistics span all graphs of ONE
call.  Each rank's call holds only its shard, so statistics are per shard: this equals the
reference invoked once per shard with that shard's graphs (option "per-shard
statistics").  The flag is recorded with every bench result as ``gn_stats="per_shard_call"``.
"""
from typing import Optional, Tuple

import torch
import torch.distributed as dist

from .weights import infer_config, pack_state_dict

GN_STATS_MODE = "per_shard_call"      # the default; "global" = pass gn_allreduce(group) as the models' gn_reduce


def gn_allreduce(group=None):
    """``gn_reduce`` callable for :class:`~difusco_amd.engine.DenoiseEngine.step`: one all-reduce (SUM) of 65 doubles
    per denoise step makes every rank normalise the head with the statistics of the WHOLE sharded batch (SURVEY 8(e)
    option "global statistics" = the reference called once over all graphs).  This is the only collective the loop
    can contain, and only when asked for; the default per-shard statistics need none."""
    def reduce_(sums: torch.Tensor):
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    return reduce_


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of ``n_items`` owned by ``rank``; sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_weights(state_dict: Optional[dict], device, src: int = 0, group=None):
    """Rank ``src`` packs the reference state_dict into the kernel blob; every rank receives
    (config, blob-on-device).  One collective of ~21 MB for the 12x256 model, at start-up only."""
    rank = dist.get_rank(group)
    if rank == src:
        hidden, n_layers, out_channels = infer_config(state_dict)
        blob = pack_state_dict(state_dict)
        meta = torch.tensor([hidden, n_layers, out_channels, blob.numel()], dtype=torch.int64)
    else:
        blob, meta = None, torch.zeros(4, dtype=torch.int64)
    meta = meta.to(device)
    dist.broadcast(meta, src=src, group=group)
    hidden, n_layers, out_channels, numel = (int(v) for v in meta.tolist())
    buf = blob.to(device) if rank == src else torch.empty(numel, dtype=torch.float32, device=device)
    dist.broadcast(buf, src=src, group=group)
    return (hidden, n_layers, out_channels), buf


def engine_from_broadcast(state_dict: Optional[dict], device, src: int = 0, group=None, precision: str = "fp16x3",
                          fused: bool = True, flags: int = 0, backend: Optional[str] = None, aggregation: str = "sum"):
    from .engine import DenoiseEngine
    (hidden, n_layers, out_channels), blob = broadcast_weights(state_dict, device, src, group)
    return DenoiseEngine(device=device, blob=blob, config=(hidden, n_layers, out_channels), precision=precision, fused=fused,
                         flags=flags, backend=backend, aggregation=aggregation)
