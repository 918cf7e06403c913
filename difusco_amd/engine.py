"""The denoise-step engine: owns the packed weights and the workspace on one GPU and drives
``difusco_denoise_step`` (C ABI).  PyTorch is used for device memory and streams only."""
import ctypes
from typing import Optional

import numpy as np
import torch

from . import _lib
from .graph import CsrGraph
from .weights import infer_config, pack_state_dict


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class DenoiseEngine:
    def __init__(self, state_dict, device="cuda:0", blob: Optional[torch.Tensor] = None, precision: str = "fp16x3",
                 fused: bool = True, backend: str = "ctypes", flags: int = 0):
        """state_dict: reference GNNEncoder weights (optionally with the Lightning ``model.`` prefix).
        ``blob``: an already packed blob (e.g. received by RCCL broadcast) instead of packing here."""
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.DifuscoHipError("DenoiseEngine needs a GPU device (no CPU fallback exists)")
        _lib.lib()  # fail loudly, now, if the HIP library is missing
        self.hidden, self.n_layers, self.out_channels = infer_config(state_dict)
        if blob is None:
            blob = pack_state_dict(state_dict)
        self.blob = blob.to(self.device, dtype=torch.float32).contiguous()
        if precision not in _lib.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_lib.PRECISIONS)}")
        self.precision = precision
        self.fused = fused          # fused edge-layer kernel (H == 256, precision bf16x3 / fp16x3)
        self.flags = int(flags)     # difusco_step_args.flags (_lib.FLAG_*): per-call A/B switches of the fused path
        if backend not in ("ctypes", "torch"):
            raise ValueError("backend must be 'ctypes' (C ABI through ctypes) or 'torch' (torch.ops.difusco custom ops)")
        self.backend = backend
        if backend == "torch":
            from . import torch_ops
            self._ops = torch_ops.load()
        self._ws = None
        self.calls = 0

    # ---- workspace -----------------------------------------------------------------------------
    def _workspace(self, g: CsrGraph) -> torch.Tensor:
        need = _lib.lib().difusco_workspace_bytes(self.hidden, self.n_layers, g.n_nodes, g.n_edges, g.n_segments)
        if need == 0:
            raise _lib.DifuscoHipError("difusco_workspace_bytes rejected the problem shape")
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    # ---- one step ------------------------------------------------------------------------------
    def step(self, g: CsrGraph, task: int, diffusion: int, xt: torch.Tensor, t: float, post: np.ndarray,
             points: Optional[torch.Tensor] = None, xt_is_binary: bool = False,
             rand: Optional[torch.Tensor] = None, seed: int = 0, offset: int = 0,
             want_pred: bool = False, want_prob: bool = False, gn_reduce=None):
        """xt: fp32, TSP [E] in caller edge order / MIS [N].  Returns (xt_next, pred|None, prob|None);
        asynchronous on the current stream.

        ``gn_reduce``: optional callable taking the device tensor of 65 doubles (32 x (sum, sum of squares) of the head
        GroupNorm input over THIS call's rows + the row count) and adding the other shards' values in place - e.g.
        ``lambda t: torch.distributed.all_reduce(t)``.  The step then runs in two phases around it, which makes a
        batch sharded over several GPUs use the statistics of the whole batch, like the reference's single call over
        all graphs (SURVEY 8(e) "global statistics").  Without it the statistics are those of this call."""
        dev = self.device
        xt = xt.to(dev, dtype=torch.float32).contiguous().reshape(-1)
        rows = g.n_edges if task == _lib.TASK_TSP else g.n_nodes
        if xt.numel() != rows:
            raise ValueError(f"xt has {xt.numel()} elements, the graph has {rows} output rows")
        if points is not None:
            points = points.to(dev, dtype=torch.float32).contiguous()
            if points.numel() != 2 * g.n_nodes:
                raise ValueError("points must be [n_nodes, 2]")
            if g.node_order is not None:          # the graph numbers its nodes for locality (graph.build_csr)
                points = points.reshape(-1, 2).index_select(0, g.node_order)
        draws = float(post[4]) != 0.0
        if rand is not None:
            rand = rand.to(dev, dtype=torch.float32).contiguous().reshape(-1)
            if rand.numel() != rows:
                raise ValueError("injected randomness must have one value per output row")
        xt_out = torch.empty(rows, dtype=torch.float32, device=dev)
        C = self.out_channels
        pred = torch.empty((rows, 2) if C == 2 else (rows,), dtype=torch.float32, device=dev) if want_pred else None
        prob = torch.empty(rows, dtype=torch.float32, device=dev) if (want_prob and C == 2) else None
        ws = self._workspace(g)
        # the Philox key and offset are 63-bit on both backends (the torch op schema carries signed 64-bit ints)
        seed, offset = int(seed) & (2 ** 63 - 1), int(offset) & (2 ** 63 - 1)
        if self.backend == "torch":
            return self._step_torch_op(g, task, diffusion, xt, t, post, points, xt_is_binary, rand, seed, offset,
                                       want_pred, want_prob, gn_reduce, ws)

        a = _lib.StepArgs()
        a.struct_size = ctypes.sizeof(_lib.StepArgs)
        a.abi_version = _lib.ABI_VERSION
        a.hidden, a.n_layers, a.out_channels, a.task = self.hidden, self.n_layers, C, task
        a.weights = _ptr(self.blob)
        a.n_nodes, a.n_edges = g.n_nodes, g.n_edges
        a.rowptr, a.col, a.perm = _ptr(g.rowptr), _ptr(g.col), _ptr(g.perm)
        a.n_segments, a.seg_ptr = g.n_segments, _ptr(g.seg_ptr)
        a.points, a.xt = _ptr(points), _ptr(xt)
        a.t = float(t)
        a.xt_is_binary = 1 if xt_is_binary else 0
        a.diffusion = diffusion
        for i in range(8):
            a.post[i] = float(post[i]) if i < len(post) else 0.0
        a.rand_mode = _lib.RAND_INJECTED if rand is not None else (_lib.RAND_PHILOX if draws else _lib.RAND_NONE)
        a.rand = _ptr(rand)
        a.seed, a.offset = seed, offset
        a.xt_out, a.pred_out, a.prob_out = _ptr(xt_out), _ptr(pred), _ptr(prob)
        a.workspace, a.workspace_bytes = _ptr(ws), ws.numel()
        a.stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        a.precision = _lib.PRECISIONS[self.precision]
        a.no_fusion = 0 if self.fused else 1
        a.row = _ptr(g.row)
        a.gn_phase, a.gn_sums = 0, None
        a.flags = self.flags
        with torch.cuda.device(dev):
            if gn_reduce is None:
                _lib.check(_lib.lib().difusco_denoise_step(ctypes.byref(a)))
            else:
                if g.n_segments != 1:
                    raise ValueError("global GroupNorm statistics need one statistic segment per call")
                sums = torch.zeros(65, dtype=torch.float64, device=dev)
                a.gn_sums = _ptr(sums)
                a.gn_phase = 1
                _lib.check(_lib.lib().difusco_denoise_step(ctypes.byref(a)))
                gn_reduce(sums)
                a.gn_phase = 2
                _lib.check(_lib.lib().difusco_denoise_step(ctypes.byref(a)))
        self.calls += 1
        return xt_out, pred, prob

    def _step_torch_op(self, g, task, diffusion, xt, t, post, points, xt_is_binary, rand, seed, offset, want_pred,
                       want_prob, gn_reduce, ws):
        """The same step through ``torch.ops.difusco.denoise_step_{categorical,gaussian}`` (csrc/torch_ops.cpp)."""
        op = self._ops.denoise_step_categorical if diffusion == _lib.CATEGORICAL else self._ops.denoise_step_gaussian
        cfg = [self.hidden, self.n_layers, self.out_channels, task, _lib.PRECISIONS[self.precision], 0 if self.fused else 1,
               1 if xt_is_binary else 0, 0, self.flags]
        seg = g.seg_ptr if g.n_segments > 1 else None
        post = [float(v) for v in post]

        def call(phase, sums):
            cfg[7] = phase
            return op(self.blob, g.rowptr, g.col, g.perm, g.row, seg, points, xt, float(t), post, rand, seed, offset, ws, cfg,
                      want_pred, want_prob, sums)
        if gn_reduce is None:
            out = call(0, None)
        else:
            if g.n_segments != 1:
                raise ValueError("global GroupNorm statistics need one statistic segment per call")
            sums = torch.zeros(65, dtype=torch.float64, device=self.device)
            call(1, sums)
            gn_reduce(sums)
            out = call(2, sums)
        self.calls += 1
        xt_out, pred, prob = out
        return xt_out, (pred if want_pred else None), (prob if (want_prob and self.out_channels == 2) else None)
