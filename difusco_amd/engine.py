"""The denoise-step engine: owns the packed weights and the workspace on one GPU and drives
``difusco_denoise_step`` (C ABI).  PyTorch is used for device memory and streams only."""
import ctypes
import os
from typing import Optional

import numpy as np
import torch

from . import _lib
from .graph import CsrGraph
from .weights import infer_config, pack_state_dict


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _op_call(fn, *args):
    """A ``torch.ops.difusco.*`` call with the error type of the ctypes binding: the shim reports a failing C entry point as
    ``RuntimeError("libdifusco_hip <entry> failed (<code>): <message>")``; both bindings raise ``DifuscoHipError`` for it."""
    try:
        return fn(*args)
    except RuntimeError as exc:
        if "libdifusco_hip " in str(exc):
            raise _lib.DifuscoHipError(str(exc).split("\n")[0]) from None
        raise


class _ReadyEvent:
    """Cross-stream ordering for cached device buffers that were produced asynchronously: an event recorded on the producing stream
    right behind the producing launch; a consumer on another stream waits for it ONCE (later work of that stream is ordered behind
    the wait anyway).  The producing stream itself never waits (stream order), which also keeps a HIP-graph capture on that stream free
    of foreign events."""

    def __init__(self, device):
        self.device = device
        st = torch.cuda.current_stream(device)
        self.event = torch.cuda.Event()
        self.event.record(st)
        self.ordered = {st.cuda_stream}

    def wait_on_current_stream(self):
        st = torch.cuda.current_stream(self.device)
        if st.cuda_stream not in self.ordered:
            st.wait_event(self.event)
            self.ordered.add(st.cuda_stream)


class DenoiseEngine:
    def __init__(self, state_dict=None, device="cuda:0", blob: Optional[torch.Tensor] = None, precision: str = "fp16x3",
                 fused: bool = True, backend: Optional[str] = None, flags: int = 0, config=None, aggregation: str = "sum",
                 use_gen_table: bool = True):
        """state_dict: reference GNNEncoder weights (optionally with the Lightning ``model.`` prefix).
        ``blob`` + ``config=(hidden, n_layers, out_channels)``: an already packed blob (e.g. received by RCCL broadcast)
        instead of a state_dict to pack here.  ``aggregation``: the reference's ``--aggregation`` (``train.py:52``,
        ``gnn_encoder.py:170-191``): "sum" (every published run), "mean" or "max"."""
        self.device = torch.device(device)
        self.use_gen_table = bool(use_gen_table)      # False: general edge inputs always take the K = 256 contraction (A/B, parity tests)
        if self.device.type != "cuda":
            raise _lib.DifuscoHipError("DenoiseEngine needs a GPU device (no CPU fallback exists)")
        _lib.lib()  # fail loudly, now, if the HIP library is missing
        if config is not None:
            if blob is None:
                raise ValueError("config=(hidden, n_layers, out_channels) describes a packed blob: pass blob= as well")
            self.hidden, self.n_layers, self.out_channels = (int(v) for v in config)
        else:
            if state_dict is None:
                raise ValueError("state_dict, or blob= with config=(hidden, n_layers, out_channels), required")
            self.hidden, self.n_layers, self.out_channels = infer_config(state_dict)
        if blob is None:
            blob = pack_state_dict(state_dict)
        _, total = _lib.weights_layout(self.hidden, self.n_layers, self.out_channels)
        if blob.numel() != total:
            raise ValueError(f"packed blob has {blob.numel()} floats, the layout of this model has {total}")
        self.blob = blob.to(self.device, dtype=torch.float32).contiguous()
        if precision not in _lib.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_lib.PRECISIONS)}")
        self.precision = precision
        if aggregation not in _lib.AGGREGATIONS:
            raise ValueError(f"aggregation must be one of {sorted(_lib.AGGREGATIONS)}")
        self.aggregation = aggregation
        self.fused = fused          # fused edge-layer kernel (H == 256, precision bf16x3 / fp16x3)
        self.flags = int(flags)     # difusco_step_args.flags (_lib.FLAG_*): per-call A/B switches of the fused path
        if backend is None:
            # default binding = the PyTorch custom ops (what BASELINE.json's north_star names; same speed as ctypes, bench.py
            # A/B); the ctypes binding when the ops cannot apply: profiling library / DIFUSCO_HIP_LIBRARY behind ctypes (the
            # ops are linked against the production library), or libdifusco_torch.so not built
            from .build import LIB_PATH, TORCH_LIB_PATH
            same = os.path.realpath(_lib.loaded_path()) == os.path.realpath(LIB_PATH)
            backend = "torch" if (same and os.path.exists(TORCH_LIB_PATH)) else "ctypes"
        if backend not in ("ctypes", "torch"):
            raise ValueError("backend must be 'ctypes' (C ABI through ctypes) or 'torch' (torch.ops.difusco custom ops)")
        self.backend = backend
        if backend == "torch":
            from . import torch_ops
            from .build import LIB_PATH
            # libdifusco_torch.so is linked against the PRODUCTION library: with the profiling build (or another
            # DIFUSCO_HIP_LIBRARY) behind ctypes, the debug knobs / profiler state set through ctypes would not apply to
            # steps issued through torch.ops - an A/B taken that way would silently measure the production kernels
            if os.path.realpath(_lib.loaded_path()) != os.path.realpath(LIB_PATH):
                raise _lib.DifuscoHipError(
                    f"backend='torch' launches through {LIB_PATH}, but ctypes loaded {_lib.loaded_path()}: "
                    "use backend='ctypes' with the profiling library / DIFUSCO_HIP_LIBRARY")
            self._ops = torch_ops.load()
        self._ws = {}               # HIP stream -> workspace tensor
        self._gen_table = None      # (table, _ReadyEvent): the generated-input table of this blob (difusco_gen_table_build), built on first use
        self._tbias = {}            # t -> ([n_layers, hidden] time-bias rows on the device, _ReadyEvent) (prepare_times)
        self.calls = 0

    # ---- workspace -----------------------------------------------------------------------------
    def _workspace(self, g: CsrGraph) -> torch.Tensor:
        """The scratch buffer of a step, one per HIP stream the engine is used on (steps on different streams may overlap; a
        step owns its workspace from launch to completion, so two streams must not share one)."""
        need = _lib.lib().difusco_workspace_bytes(self.hidden, self.n_layers, g.n_nodes, g.n_edges, g.n_segments)
        if need == 0:
            raise _lib.DifuscoHipError("difusco_workspace_bytes rejected the problem shape")
        key = torch.cuda.current_stream(self.device).cuda_stream
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            self._ws.pop(key, None)
            ws = self._ws[key] = torch.empty(need, dtype=torch.uint8, device=self.device)
        return ws

    def _cfg(self, task: int = _lib.TASK_TSP, xt_is_binary: bool = False, phase: int = 0):
        """cfg list of the torch ops: {hidden, n_layers, out_channels, task, precision, no_fusion, xt_is_binary, gn_phase, flags,
        aggregation}"""
        return [self.hidden, self.n_layers, self.out_channels, task, _lib.PRECISIONS[self.precision], 0 if self.fused else 1,
                1 if xt_is_binary else 0, phase, self.flags, _lib.AGGREGATIONS[self.aggregation]]

    # ---- generated-input table (difusco_step_args.gen_table, ABI 12) --------------------------------
    def gen_table(self) -> Optional[torch.Tensor]:
        """The table that lets a TSP step with a general edge input (Gaussian diffusion, non-binary categorical x_t) evaluate
        ``edge_embed(ScalarEmbeddingSine(x_t))`` (``gnn_encoder.py:230-249,304,395``) by interpolation for |x_t| < 8 - a function of the
        weights only, built once per engine on the fused path (H = 256); None otherwise or when ``use_gen_table`` is False."""
        if not self.use_gen_table or not (self.fused and self.hidden == 256 and self.precision in ("bf16x3", "fp16x3")):
            return None
        if self._gen_table is None:
            if self.backend == "torch":
                tab = _op_call(self._ops.gen_table_build, self.blob, self._cfg())
            else:
                need = _lib.lib().difusco_gen_table_bytes(self.hidden)
                tab = torch.empty(need // 4, dtype=torch.float32, device=self.device)
                with torch.cuda.device(self.device):
                    _lib.check(_lib.lib().difusco_gen_table_build(
                        self.hidden, self.n_layers, self.out_channels, _ptr(self.blob), _ptr(tab), need,
                        ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
            self._gen_table = (tab, _ReadyEvent(self.device))
        self._gen_table[1].wait_on_current_stream()
        return self._gen_table[0]

    # ---- prepared state (difusco_step_args.prepared / .tbias) --------------------------------------
    def prepare_times(self, ts) -> None:
        """Time-bias rows of every diffusion time in ``ts`` (e.g. the 50 steps of a schedule) in ONE launch; ``step`` then
        passes the row block of its ``t`` instead of running the time MLP (``gnn_encoder.py:396,329-337``)."""
        # The rows are produced asynchronously on the stream that is current HERE; each cache entry carries an event recorded
        # behind that launch, and a step on ANOTHER stream waits for the event once (the key is the time alone - a
        # raw stream handle can be reused by a new stream after the old one is destroyed, and an eviction no longer drops the
        # rows of every other stream)
        todo = sorted({float(t) for t in ts} - set(self._tbias))
        if not todo:
            return
        if self.backend == "torch":
            out = _op_call(self._ops.time_bias_rows, self.blob, todo, self._cfg())
        else:
            out = torch.empty((len(todo), self.n_layers, self.hidden), dtype=torch.float32, device=self.device)
            arr = (ctypes.c_float * len(todo))(*todo)
            with torch.cuda.device(self.device):
                _lib.check(_lib.lib().difusco_time_bias_rows(
                    self.hidden, self.n_layers, self.out_channels, _ptr(self.blob), arr, len(todo), _ptr(out),
                    ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        ready = _ReadyEvent(self.device)
        while len(self._tbias) + len(todo) > 4096 and self._tbias:      # oldest first (dicts keep insertion order)
            self._tbias.pop(next(iter(self._tbias)))
        for i, t in enumerate(todo):
            self._tbias[t] = (out[i], ready)

    def _uses_prepared(self, g: CsrGraph) -> bool:
        """Will ``difusco_denoise_step`` read a prepared buffer for a call on graph ``g``?  The C side's ``fused`` predicate
        (api.hip): H = 256, a split precision with a fused kernel, edges, and not (max aggregation with >= 2^20 nodes)."""
        return (self.fused and self.hidden == 256 and self.precision in ("bf16x3", "fp16x3") and g.n_edges > 0
                and not (self.aggregation == "max" and g.n_nodes >= (1 << 20)))

    def prepare(self, g: CsrGraph, points: torch.Tensor, force: bool = False) -> Optional[torch.Tensor]:
        """The step-invariant part of a TSP step for (these weights, this graph, these coordinates) - node embedding,
        layer 0's node linear, the two-row edge-input table (``difusco_prepare``) - as an opaque device buffer to hand to
        ``step(prepared=...)``.  None when the fused path does not apply (the step then computes everything itself)."""
        if not force and not self._uses_prepared(g):      # (a buffer for a call that would ignore it is never handed out;
            return None                                   #  force=True: tests of the C entry under a precision without a fused path)
        pts = points.to(self.device, dtype=torch.float32).contiguous()
        if pts.numel() != 2 * g.n_nodes:
            raise ValueError("points must be [n_nodes, 2]")
        if g.node_order is not None:
            pts = pts.reshape(-1, 2).index_select(0, g.node_order)
        ws = self._workspace(g)
        if self.backend == "torch":
            return _op_call(self._ops.prepare_state, self.blob, pts, g.n_nodes, g.n_edges, g.n_segments, ws, self._cfg())
        need = _lib.lib().difusco_prepared_bytes(self.hidden, g.n_nodes)
        buf = torch.empty(need, dtype=torch.uint8, device=self.device)
        a = _lib.StepArgs()
        a.struct_size = ctypes.sizeof(_lib.StepArgs)
        a.abi_version = _lib.ABI_VERSION
        a.hidden, a.n_layers, a.out_channels, a.task = self.hidden, self.n_layers, self.out_channels, _lib.TASK_TSP
        a.weights = _ptr(self.blob)
        a.n_nodes, a.n_edges, a.n_segments = g.n_nodes, g.n_edges, g.n_segments
        a.points = _ptr(pts)
        a.workspace, a.workspace_bytes = _ptr(ws), ws.numel()
        a.stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        a.precision = _lib.PRECISIONS[self.precision]
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().difusco_prepare(ctypes.byref(a), _ptr(buf), buf.numel()))
        return buf

    # ---- one step ------------------------------------------------------------------------------
    def step(self, g: CsrGraph, task: int, diffusion: int, xt: torch.Tensor, t: float, post: np.ndarray,
             points: Optional[torch.Tensor] = None, xt_is_binary: bool = False,
             rand: Optional[torch.Tensor] = None, seed: int = 0, offset: int = 0,
             want_pred: bool = False, want_prob: bool = False, gn_reduce=None, prepared: Optional[torch.Tensor] = None):
        """xt: fp32, TSP [E] in caller edge order / MIS [N].  Returns (xt_next, pred|None, prob|None);
        asynchronous on the current stream.

        ``gn_reduce``: optional callable taking the device tensor of 65 doubles (32 x (sum, sum of squares) of the head
        GroupNorm input over THIS call's rows + the row count) and adding the other shards' values in place - e.g.
        ``lambda t: torch.distributed.all_reduce(t)``.  The step then runs in two phases around it, which makes a
        batch sharded over several GPUs use the statistics of the whole batch, like the reference's single call over
        all graphs (SURVEY 8(e) "global statistics").  Without it the statistics are those of this call.

        ``prepared``: the buffer ``prepare(g, points)`` returned for this graph and these coordinates (TSP): the step skips
        what it holds; ``points`` may then be omitted.  The time-bias rows of ``t`` are taken from the ``prepare_times``
        cache when present.  Both are bit-identical to the stateless step."""
        dev = self.device
        xt = xt.to(dev, dtype=torch.float32).contiguous().reshape(-1)
        rows = g.n_edges if task == _lib.TASK_TSP else g.n_nodes
        if xt.numel() != rows:
            raise ValueError(f"xt has {xt.numel()} elements, the graph has {rows} output rows")
        if prepared is not None and task == _lib.TASK_TSP and self._uses_prepared(g):
            points = None      # (h0 and layer 0's node rows come from the prepared buffer)
        elif prepared is not None and not self._uses_prepared(g):
            prepared = None    # (the step would ignore it and needs the points: e.g. the engine was switched to the unfused sequence)
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
        tbias = self._tbias.get(float(t))
        if tbias is not None:
            tbias[1].wait_on_current_stream()      # (no-op on the producing stream and after the first wait of this stream)
            tbias = tbias[0]
        if prepared is not None and isinstance(prepared, tuple):      # (buffer, _ReadyEvent) from models._prepared
            prepared[1].wait_on_current_stream()
            prepared = prepared[0]
        gen_table = self.gen_table() if (task == _lib.TASK_TSP and not xt_is_binary) else None
        # the Philox key and offset are 63-bit on both backends (the torch op schema carries signed 64-bit ints)
        seed, offset = int(seed) & (2 ** 63 - 1), int(offset) & (2 ** 63 - 1)
        if self.backend == "torch":
            return self._step_torch_op(g, task, diffusion, xt, t, post, points, xt_is_binary, rand, seed, offset,
                                       want_pred, want_prob, gn_reduce, ws, prepared, tbias, gen_table)

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
        a.prepared, a.tbias = _ptr(prepared), _ptr(tbias)
        a.gen_table = _ptr(gen_table)
        a.aggregation = _lib.AGGREGATIONS[self.aggregation]
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
                       want_prob, gn_reduce, ws, prepared=None, tbias=None, gen_table=None):
        """The same step through ``torch.ops.difusco.denoise_step_{categorical,gaussian}`` (csrc/torch_ops.cpp)."""
        op = self._ops.denoise_step_categorical if diffusion == _lib.CATEGORICAL else self._ops.denoise_step_gaussian
        cfg = self._cfg(task, xt_is_binary)
        seg = g.seg_ptr if g.n_segments > 1 else None
        post = [float(v) for v in post]

        def call(phase, sums):
            cfg[7] = phase
            return _op_call(op, self.blob, g.rowptr, g.col, g.perm, g.row, seg, points, xt, float(t), post, rand, seed, offset, ws,
                            cfg, want_pred, want_prob, sums, prepared, tbias, gen_table)
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
