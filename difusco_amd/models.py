"""Drop-in host side of the denoise step: ``TSPModel`` / ``MISModel`` with the reference's method
names, argument order and argument meaning (``difusco/pl_tsp_model.py:122-151``,
``difusco/pl_mis_model.py:118-140``), backed by the gfx950 library instead of the PyTorch-Lightning
modules.  Only the inference path exists here (no training, data loading, tour decoding).

    model = TSPModel(param_args, state_dict)            # param_args: the reference's argparse namespace
    xt = model.categorical_denoise_step(points, xt, t, device, edge_index, target_t=target_t)

``t`` / ``target_t`` are numpy int arrays of shape (1,) exactly as the reference's ``test_step`` passes
them (``pl_tsp_model.py:209-210``); python ints are accepted too.  Returned tensors are new fp32
tensors on ``device``, flattened in sparse mode (``pl_meta_model.py:144-145``).

Extensions beyond the reference signature are keyword-only: ``uniform=`` / ``noise=`` inject the
random numbers (teacher-forced parity tests), ``return_aux=True`` also returns the network
prediction and the pre-sampling probability.
"""
from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch

from . import _lib
from .engine import DenoiseEngine
from .graph import CsrGraph, build_csr, complete_graph_batch
from .schedules import CategoricalDiffusion, GaussianDiffusion, InferenceSchedule

_DEFAULTS = dict(  # difusco/train.py:19-68 (only what the inference path reads)
    diffusion_type="gaussian", diffusion_schedule="linear", diffusion_steps=1000,
    inference_diffusion_steps=1000, inference_schedule="linear", inference_trick="ddim",
    sequential_sampling=1, parallel_sampling=1, n_layers=12, hidden_dim=256, sparse_factor=-1,
    aggregation="sum")


def _as_int(t) -> int:
    if t is None:
        return None
    return int(np.asarray(t.detach().cpu() if isinstance(t, torch.Tensor) else t).reshape(-1)[0])


class COMetaModel:
    """Inference-side state of the reference's ``COMetaModel`` (``pl_meta_model.py:16-47``): the
    diffusion tables, the denoiser weights (as a ``DenoiseEngine``) and the two posteriors."""

    def __init__(self, param_args=None, state_dict=None, node_feature_only=False, device="cuda:0",
                 seed: Optional[int] = None, engine: Optional[DenoiseEngine] = None, precision: str = "fp16x3",
                 fused: bool = True, gn_reduce=None, reorder_nodes: bool = True, backend: Optional[str] = None, flags: int = 0,
                 prepare: bool = True, strict_binary_check: bool = False):
        args = dict(_DEFAULTS)
        if param_args is not None:
            args.update(vars(param_args) if not isinstance(param_args, dict) else param_args)
        self.args = SimpleNamespace(**args)
        self.diffusion_type = self.args.diffusion_type
        self.diffusion_schedule = self.args.diffusion_schedule
        self.diffusion_steps = self.args.diffusion_steps
        self.node_feature_only = node_feature_only
        self.sparse = self.args.sparse_factor > 0 or node_feature_only
        if self.args.aggregation not in ("sum", "mean", "max"):      # gnn_encoder.py:170-191 (every published run uses sum)
            raise ValueError(f"Unknown aggregation {self.args.aggregation}")
        if self.diffusion_type == "gaussian":
            self.diffusion = GaussianDiffusion(T=self.diffusion_steps, schedule=self.diffusion_schedule)
            out_channels = 1
        elif self.diffusion_type == "categorical":
            self.diffusion = CategoricalDiffusion(T=self.diffusion_steps, schedule=self.diffusion_schedule)
            out_channels = 2
        else:
            raise ValueError(f"Unknown diffusion type {self.diffusion_type}")
        if engine is None:
            if state_dict is None:
                raise ValueError("state_dict (reference GNNEncoder weights) or engine required")
            engine = DenoiseEngine(state_dict, device=device, precision=precision, fused=fused, backend=backend, flags=flags,
                                   aggregation=self.args.aggregation)
        elif engine.aggregation != self.args.aggregation:
            raise ValueError(f"the engine aggregates by {engine.aggregation}, the model arguments say {self.args.aggregation}")
        if engine.out_channels != out_channels:
            raise ValueError(f"weights have {engine.out_channels} output channels, "
                             f"{self.diffusion_type} diffusion needs {out_channels}")
        self.model = engine
        self.device = engine.device
        if seed is None:
            # default Philox key: torch's initial seed, made different per rank when torch.distributed runs (ranks of a
            # sharded batch must not draw the same Bernoulli / normal streams).  An explicit ``seed=`` is used as given:
            # callers that shard samples over ranks pass different seeds (bench.py: 1234 + rank).
            seed = torch.initial_seed()
            if torch.distributed.is_available() and torch.distributed.is_initialized():
                seed ^= (torch.distributed.get_rank() + 1) * 0x9E3779B97F4A7C15
        self.seed = int(seed) & (2 ** 63 - 1)
        self._binary_out = None     # (tensor, version) of the last Bernoulli-sampled output: known to be exactly {0,1}
        self._graph_cache = {}
        # prepared state (difusco_step_args.prepared / .tbias): what a TSP step computes from (weights, graph, points) alone
        # is computed once per (graph, points) and reused by every step of a sampling loop; `prepare=False` = stateless steps
        self.prepare = bool(prepare)
        self._prep_cache = {}
        # always verify on the device that a categorical x_t is exactly {0,1} (one reduction + sync per step) instead of
        # trusting the identity of our own last Bernoulli output - see _xt_is_binary
        self.strict_binary_check = bool(strict_binary_check)
        self.reorder_nodes = reorder_nodes      # TSP: Morton-order the nodes of every graph for L2 locality (graph.py)
        # optional: shard-summing callable for the head GroupNorm statistics (difusco_amd.dist.gn_allreduce); None =
        # statistics of each call's own rows, the reference's behaviour for that call
        self.gn_reduce = gn_reduce

    # ---- graph handling --------------------------------------------------------------------------
    def prepare_graph(self, edge_index: torch.Tensor, num_nodes: int, points=None) -> CsrGraph:
        """COO -> CSR once per instance; cached on the identity of ``edge_index`` so the 50 calls of
        a sampling loop convert once.  ``points`` (TSP): lets the graph renumber its nodes along a space-filling
        curve (``graph.locality_node_order``) - a pure locality choice, valid for any coordinates, invisible outside.
        Tensors created under ``torch.inference_mode()`` (Lightning's default for ``trainer.test``) have no version
        counter: for those the key is the storage address alone, which the cache keeps alive."""
        version = 0 if edge_index.is_inference() else edge_index._version
        key = (edge_index.data_ptr(), tuple(edge_index.shape), version, int(num_nodes))
        g = self._graph_cache.get(key)
        if g is None:
            if len(self._graph_cache) > 8:
                self._graph_cache.clear()
                self._prep_cache.clear()      # (prepared buffers hold their graphs alive)
            g = build_csr(edge_index, int(num_nodes), self.device, points=points if self.reorder_nodes else None)
            self._graph_cache[key] = (g, edge_index)   # keep edge_index alive: data_ptr stays unique
            return g
        return g[0]

    def _dense_graph(self, batch: int, n: int) -> CsrGraph:
        key = ("dense", batch, n)
        g = self._graph_cache.get(key)
        if g is None:
            g = (complete_graph_batch(batch, n, self.device), None)
            self._graph_cache[key] = g
        return g[0]

    def _prepared(self, g: CsrGraph, points):
        """Prepared buffer of (this engine, graph ``g``, ``points``) or None; cached on the identity of ``points`` (the
        sampling loop passes the same tensor 50 times - same rule as ``prepare_graph``)."""
        if not self.prepare or points is None:
            return None
        version = 0 if points.is_inference() else points._version
        # (the buffer is produced asynchronously on the stream that is current at the miss: the entry carries an event, and a step on
        #  another stream waits for it once - engine._ReadyEvent; the key holds no stream handle)
        key = (id(g), points.data_ptr(), tuple(points.shape), version)
        hit = self._prep_cache.get(key)
        if hit is None:
            if len(self._prep_cache) >= 2:      # a sampling loop reuses ONE entry; each pins 5 N H floats + the graph + the points
                self._prep_cache.clear()
            buf = self.model.prepare(g, points)
            from .engine import _ReadyEvent
            hit = ((buf, _ReadyEvent(self.device)) if buf is not None else None, g, points)      # keep g / points alive: id() and data_ptr() stay unique
            self._prep_cache[key] = hit
        return hit[0]

    def prepare_schedule(self, times) -> None:
        """Precompute the time-bias rows of every diffusion time a sampling loop will visit (one launch; ``sample`` calls it
        with the schedule's ``t1`` values).  Steps at other times run the time MLP themselves."""
        if self.prepare:
            self.model.prepare_times([_as_int(t) for t in times])

    def duplicate_edge_index(self, edge_index, num_nodes, device, copies: Optional[int] = None):
        """Disjoint union of ``parallel_sampling`` replicas (pl_meta_model.py:177-184).  ``copies`` (extension): the number of
        replicas, instead of ``self.args.parallel_sampling`` - callers need not write to the args to ask for a batch."""
        P = self.args.parallel_sampling if copies is None else int(copies)
        shift = torch.arange(0, P, device=device).view(1, -1, 1) * num_nodes
        return (edge_index.reshape((2, 1, -1)).to(device) + shift).reshape((2, -1))

    # ---- the two step flavours -------------------------------------------------------------------
    def _next_offset(self) -> int:
        return self.model.calls

    def _xt_is_binary(self, xt: torch.Tensor) -> bool:
        """Is ``xt`` exactly {0,1}-valued?  The reference embeds whatever value arrives (``pl_tsp_model.py:127``:
        ``xt.float()``) and truncates with ``.long()`` in the posterior (``pl_meta_model.py:122``); the two-row embedding
        table and the table-input first layer are exact only for 0/1 inputs.  The sampling loop feeds our own Bernoulli
        outputs back, which are known without looking (same storage, unmodified); anything else is checked on the
        device (one small reduction + sync per call).  Values whose truncation is not 0/1 raise, like ``one_hot``."""
        # NOTE (tensors without a version counter, i.e. created under torch.inference_mode()): an in-place edit of the tensor
        # this model returned cannot be detected without looking at the data; such an x_t must not be mutated in place
        # between steps (or construct the model with strict_binary_check=True, which always looks).
        known = None if getattr(self, "strict_binary_check", False) else self._binary_out      # (tensor kept alive, version or None)
        if known is not None and known[0].data_ptr() == xt.data_ptr() and known[0].numel() == xt.numel() \
                and (known[1] is None) == xt.is_inference() and (known[1] is None or known[1] == xt._version):
            return True
        if bool(((xt == 0) | (xt == 1)).all()):
            return True
        tr = xt.long()
        if not bool(((tr == 0) | (tr == 1)).all()):
            raise ValueError("categorical x_t must truncate to 0/1 (F.one_hot(xt.long(), num_classes=2), "
                             "pl_meta_model.py:122-123)")
        return False

    def _categorical(self, g, task, points, xt, t, target_t, uniform, return_aux):
        t, target_t = _as_int(t), _as_int(target_t)
        if target_t is None:
            target_t = t - 1                                              # pl_meta_model.py:108-109
        post = np.zeros(8, dtype=np.float32)
        post[:4] = self.diffusion.posterior_constants(t, target_t)
        post[4] = 1.0 if target_t > 0 else 0.0                            # :139-142
        out, pred, prob = self.model.step(
            g, task, _lib.CATEGORICAL, xt, float(t), post, points=points, xt_is_binary=self._xt_is_binary(xt),
            rand=uniform if target_t > 0 else None, seed=self.seed, offset=self._next_offset(),
            want_pred=return_aux, want_prob=return_aux, gn_reduce=self.gn_reduce,
            prepared=self._prepared(g, points) if task == _lib.TASK_TSP else None)
        # tensors created under torch.inference_mode() (Lightning's default for trainer.test) have no version counter:
        # for those the held reference + storage identity is the whole key (no host sync in the 50-step loop either way)
        self._binary_out = (out, None if out.is_inference() else out._version) if target_t > 0 else None
        return (out, pred, prob) if return_aux else out

    def _gaussian(self, g, task, points, xt, t, target_t, noise, return_aux):
        t, target_t = _as_int(t), _as_int(target_t)
        if target_t is None:
            target_t = t - 1
        post = np.zeros(8, dtype=np.float32)
        post[:5] = self.diffusion.posterior_constants(t, target_t, self.args.inference_trick)
        out, pred, _ = self.model.step(
            g, task, _lib.GAUSSIAN, xt, float(t), post, points=points, xt_is_binary=False,
            rand=noise if post[4] != 0 else None, seed=self.seed, offset=self._next_offset(), want_pred=return_aux,
            gn_reduce=self.gn_reduce, prepared=self._prepared(g, points) if task == _lib.TASK_TSP else None)
        return (out, pred) if return_aux else out


class TSPModel(COMetaModel):
    """Inference half of ``difusco/pl_tsp_model.py``."""

    def __init__(self, param_args=None, state_dict=None, **kw):
        super().__init__(param_args=param_args, state_dict=state_dict, node_feature_only=False, **kw)

    def _graph_and_inputs(self, points, xt, edge_index):
        if edge_index is not None:
            g = self.prepare_graph(edge_index, points.shape[0], points=points)
            return g, points.reshape(-1, 2), xt.reshape(-1), None
        if points.dim() != 3:
            raise ValueError("dense mode expects points [B,V,2] and xt [B,V,V]")
        B, V = points.shape[0], points.shape[1]
        return self._dense_graph(B, V), points.reshape(-1, 2), xt.reshape(-1), (B, V, V)

    def categorical_denoise_step(self, points, xt, t, device, edge_index=None, target_t=None, *,
                                 uniform=None, return_aux=False):
        g, pts, x, dense_shape = self._graph_and_inputs(points, xt, edge_index)
        res = self._categorical(g, _lib.TASK_TSP, pts, x.float(), t, target_t, uniform, return_aux)
        if dense_shape is None:
            return res
        if return_aux:
            out, pred, prob = res
            return out.reshape(dense_shape), pred.reshape(dense_shape + (2,)), prob.reshape(dense_shape)
        return res.reshape(dense_shape)

    def gaussian_denoise_step(self, points, xt, t, device, edge_index=None, target_t=None, *,
                              noise=None, return_aux=False):
        g, pts, x, dense_shape = self._graph_and_inputs(points, xt, edge_index)
        res = self._gaussian(g, _lib.TASK_TSP, pts, x.float(), t, target_t, noise, return_aux)
        if dense_shape is None:
            return res
        if return_aux:
            out, pred = res
            return out.reshape(dense_shape), pred.reshape(dense_shape)
        return res.reshape(dense_shape)

    def sample(self, points, edge_index=None, xt0=None, generator=None):
        """The sampling loop of ``test_step`` (``pl_tsp_model.py:185-222``) for ONE noise sample per
        graph of the call: returns the heatmap tensor (``+1e-6`` categorical, ``*0.5+0.5`` gaussian),
        still on the device.  ``points``/``edge_index`` already hold the (possibly duplicated) batch."""
        steps = self.args.inference_diffusion_steps
        sched = InferenceSchedule(inference_schedule=self.args.inference_schedule, T=self.diffusion.T,
                                  inference_T=steps)
        if xt0 is None:
            shape = (edge_index.shape[1],) if edge_index is not None else (points.shape[0], points.shape[1], points.shape[1])
            xt0 = torch.randn(shape, generator=generator, device=self.device if generator is None else generator.device)
        xt = xt0.to(self.device)
        if self.diffusion_type == "categorical":
            xt = (xt > 0).float()
        self.prepare_schedule([sched(i)[0] for i in range(steps)])
        for i in range(steps):
            t1, t2 = sched(i)
            t1, t2 = np.array([t1]).astype(int), np.array([t2]).astype(int)
            if self.diffusion_type == "gaussian":
                xt = self.gaussian_denoise_step(points, xt, t1, self.device, edge_index, target_t=t2)
            else:
                xt = self.categorical_denoise_step(points, xt, t1, self.device, edge_index, target_t=t2)
        return xt * 0.5 + 0.5 if self.diffusion_type == "gaussian" else xt + 1e-6


class MISModel(COMetaModel):
    """Inference half of ``difusco/pl_mis_model.py`` (node features only)."""

    def __init__(self, param_args=None, state_dict=None, **kw):
        super().__init__(param_args=param_args, state_dict=state_dict, node_feature_only=True, **kw)

    def categorical_denoise_step(self, xt, t, device, edge_index=None, target_t=None, *, uniform=None,
                                 return_aux=False):
        g = self.prepare_graph(edge_index, xt.reshape(-1).shape[0])
        return self._categorical(g, _lib.TASK_MIS, None, xt.reshape(-1).float(), t, target_t, uniform, return_aux)

    def gaussian_denoise_step(self, xt, t, device, edge_index=None, target_t=None, *, noise=None, return_aux=False):
        g = self.prepare_graph(edge_index, xt.reshape(-1).shape[0])
        return self._gaussian(g, _lib.TASK_MIS, None, xt.reshape(-1).float(), t, target_t, noise, return_aux)

    def sample(self, n_nodes, edge_index, xt0=None, generator=None):
        """``pl_mis_model.py:156-192`` for one noise sample per graph of the call."""
        steps = self.args.inference_diffusion_steps
        sched = InferenceSchedule(inference_schedule=self.args.inference_schedule, T=self.diffusion.T,
                                  inference_T=steps)
        if xt0 is None:
            xt0 = torch.randn((n_nodes,), generator=generator, device=self.device if generator is None else generator.device)
        xt = xt0.to(self.device)
        if self.diffusion_type == "categorical":
            xt = (xt > 0).float()
        self.prepare_schedule([sched(i)[0] for i in range(steps)])
        for i in range(steps):
            t1, t2 = sched(i)
            t1, t2 = np.array([t1]).astype(int), np.array([t2]).astype(int)
            if self.diffusion_type == "gaussian":
                xt = self.gaussian_denoise_step(xt, t1, self.device, edge_index, target_t=t2)
            else:
                xt = self.categorical_denoise_step(xt, t1, self.device, edge_index, target_t=t2)
        return xt * 0.5 + 0.5 if self.diffusion_type == "gaussian" else xt + 1e-6
