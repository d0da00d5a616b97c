"""Flat parameter arena + Adam (A12) + data-parallel gradient reduction (A13), MI355X-first.

All 6.9 M parameters live in ONE fp32 buffer (27.6 MB) with matching gradient / exp_avg / exp_avg_sq arenas:
  * the optimizer step is a single HBM-bound HIP kernel (csrc/misc.hip adam_kernel) instead of ~130 per-tensor ops;
  * the data-parallel gradient all-reduce is ONE RCCL collective over the whole arena -- xGMI is point-to-point and
    per-link bound, so one large message beats many 25 MB DDP buckets' latency, and 27.6 MB is < 0.5 ms of a
    > 100 ms step either way;
  * conv weights are stored O,kh,kw,I (what the MFMA kernels read) and exposed as [O,I,kh,kw] channels_last views, so
    state_dict()/load_state_dict() keep the reference's shapes with no per-step layout transform;
  * gru.convz / gru.convr are placed back to back so the decoder kernel's packed [z|r] matrix is a free view.
Semantics = torch.optim.Adam(params, lr) defaults (no weight decay, no amsgrad) as the OpenSceneFlow trainer uses
(lr from the command line [REF README.md:66]).
"""
from __future__ import annotations

import os

from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import ops
from ._lib import call, img, ptr, stream


def _arena_order(named: List[tuple]) -> List[tuple]:
    names = [n for n, _ in named]
    d = dict(named)
    for group in (("gru.convz.weight", "gru.convr.weight"), ("gru.convz.bias", "gru.convr.bias")):
        hits = [[n for n in names if n.endswith(sfx)] for sfx in group]
        if all(len(h) == 1 for h in hits):
            a, b = hits[0][0], hits[1][0]
            names.remove(b)
            names.insert(names.index(a) + 1, b)
    return [(n, d[n]) for n in names]


class FlatParams:
    def __init__(self, module: nn.Module):
        named = _arena_order([(n, p) for n, p in module.named_parameters()])
        dev = named[0][1].device
        self.slots: Dict[str, tuple] = {}
        off = 0
        for n, p in named:
            self.slots[n] = (off, p.numel())
            off += p.numel()
            if not (n.endswith("gru.convz.weight") or n.endswith("gru.convz.bias")):
                off = (off + 3) // 4 * 4  # 16-byte aligned slots (z|r pairs stay contiguous)
        self.numel = (off + 3) // 4 * 4
        self.param = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.params: List[nn.Parameter] = []
        self.named = named
        for n, p in named:
            o, k = self.slots[n]
            view, gview = self._view(self.param, o, p), self._view(self.grad, o, p)
            with torch.no_grad():
                view.copy_(p.data)
            p.data = view
            p.grad = gview
            self.params.append(p)

    @staticmethod
    def _view(buf: torch.Tensor, off: int, p: torch.Tensor) -> torch.Tensor:
        flat = buf[off:off + p.numel()]
        if p.dim() == 4:  # conv weight: O,kh,kw,I memory, [O,I,kh,kw] logical
            o, i, kh, kw = p.shape
            return flat.view(o, kh, kw, i).permute(0, 3, 1, 2)
        return flat.view(p.shape)

    def zero_grad(self):
        self.grad.zero_()


class GradSink:
    """Takes parameter gradients from INSIDE the hand-sequenced backward, phase by phase (decoder head -> UNet decoder ->
    encoder stages 3, 2, 1 -> pillar feature net = reverse order of the arena), instead of after it:

      * the gradients of a phase are copied into their arena views with one multi-tensor copy (the ~100 per-parameter
        AccumulateGrad adds of autograd disappear; the backward returns None for delivered parameters);
      * with more than one rank, every contiguous arena run a phase completes is all-reduced asynchronously right away
        (RCCL on its own stream, ordered after the copies), so the collective overlaps the rest of the backward --
        SURVEY section 8(e) / BASELINE configs[3].  Runs never span undelivered parameters, so nothing is reduced twice.

    Overwrite semantics: one backward per Trainer.step (the Trainer zero-fills the arena first); plain autograd users
    (no sink installed) keep the usual accumulate-into-.grad behaviour."""

    def __init__(self, flat: "FlatParams", dist=None, pg=None, world: int = 1, collective: bool = None):
        self.flat, self.dist, self.pg, self.world = flat, dist, pg, world
        self.collective = (world > 1) if collective is None else collective
        self.slot_of = {p.data_ptr(): flat.slots[n] for n, p in flat.named}
        self.delivered = set()
        self.works: list = []
        self.cap: Optional["SegmentedCapture"] = None   # set while Trainer.capture records the step
        # DF_ONE_BUCKET=1 (or .one_bucket = True): no bucketing -- the phases only copy, ONE all-reduce over the whole arena goes out
        # after the backward (nothing overlapped): the fallback / A-B leg of the first multi-GPU runs
        self.one_bucket = os.environ.get("DF_ONE_BUCKET") == "1"
        # trace_on(): per-bucket record of an eager step -- arena range, bytes, issue -> complete times on the device -- without
        # changing the step's own ordering (a probe stream waits for each collective and drops an event)
        self.trace: Optional[list] = None
        self._probe = None
        self._t0 = None

    def begin(self):
        self.delivered.clear()
        self.works.clear()
        if self.trace is not None:
            self.trace.clear()
            self._t0 = torch.cuda.Event(enable_timing=True)
            self._t0.record()

    def trace_on(self, on: bool = True):
        self.trace = [] if on else None
        if on and self._probe is None and self.flat.grad.is_cuda:
            self._probe = torch.cuda.Stream(device=self.flat.grad.device)

    def trace_report(self) -> list:
        """after a traced step (and a device synchronize): one dict per collective in issue order -- arena range, bytes, when it was
        issued / completed relative to the start of the step (ms, device clock) and how long it was in flight"""
        out = []
        for lo, hi, e_issue, e_done in (self.trace or []):
            out.append({"arena_range": [lo, hi], "bytes": (hi - lo) * 4, "issued_ms": self._t0.elapsed_time(e_issue),
                        "completed_ms": self._t0.elapsed_time(e_done), "in_flight_ms": e_issue.elapsed_time(e_done)})
        return out

    def deliver(self, params, grads):
        dsts, srcs, slots = [], [], []
        for p in params:
            key = p.data_ptr()
            if key in self.delivered or key not in self.slot_of:
                continue
            g = grads.lookup(p)
            if g is None:
                continue
            if g.shape != p.shape:
                g = g.reshape(p.shape)
            dsts.append(p.grad)
            srcs.append(g)
            slots.append(self.slot_of[key])
            self.delivered.add(key)
        if not dsts:
            return
        torch._foreach_copy_(dsts, srcs)
        if self.cap is not None:
            self.cap.touch()
        if self.collective and not self.one_bucket:
            slots.sort()
            lo, hi = slots[0][0], slots[0][0] + slots[0][1]
            runs = []
            for off, n in slots[1:]:
                if off <= hi + 3:   # adjacent up to the 16-byte slot padding (zeros)
                    hi = max(hi, off + n)
                else:
                    runs.append((lo, hi))
                    lo, hi = off, off + n
            runs.append((lo, hi))
            self.all_reduce_runs(runs)

    def all_reduce_runs(self, runs):
        """asynchronous sum all-reduce of arena ranges [lo, hi) -- or, while a step is being captured, a split of the HIP graph
        at this point with the collectives recorded between the two segments (SegmentedCapture)"""
        if self.cap is not None:
            self.cap.emit(("allreduce", list(runs)))
            return
        for lo, hi in runs:
            if self.trace is not None and self._probe is not None:
                e_issue, e_done = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e_issue.record()
                w = self.dist.all_reduce(self.flat.grad[lo:hi], group=self.pg, async_op=True)
                with torch.cuda.stream(self._probe):     # the probe stream (only) waits for this collective
                    w.wait()
                    e_done.record()
                self.trace.append((lo, hi, e_issue, e_done))
                self.works.append(w)
                continue
            self.works.append(self.dist.all_reduce(self.flat.grad[lo:hi], group=self.pg, async_op=True))

    def was_delivered(self, p) -> bool:
        return p.data_ptr() in self.delivered

    def finish(self):
        if self.cap is not None:
            self.cap.emit(("wait",))
            return
        for w in self.works:
            w.wait()
        self.works.clear()


class SegmentedCapture:
    """The training step recorded as a PROGRAM: HIP-graph segments with the data-parallel collectives between them,

        [graph 0: forward, loss, decoder backward] [all-reduce head bucket] [graph 1: UNet decoder backward] [all-reduce ...]
        ... [graph 5: pillar feature net backward] [all-reduce] [wait] [graph 6: Adam]

    so that replaying a data-parallel step costs the host a handful of graph launches and `all_reduce` calls instead of ~410
    Python-issued kernel launches, while RCCL keeps overlapping every bucket with the rest of the backward exactly as in the
    eager step (the collective is enqueued right after the segment that produced its bucket; the next segment does not wait
    for it).  The collectives themselves stay OUTSIDE the graphs: they run on the process group's own stream / threads
    (RCCL, or gloo in the tests) and need no capture support from the backend.  All segments share one allocator pool and are
    replayed in capture order, so tensors that live across a split (the tape, the gradient buffers) stay valid.
    `fresh` = nothing has been launched into the open segment yet: an op emitted then needs no split (no empty graphs)."""

    def __init__(self):
        self.ops: list = []
        self.pool = torch.cuda.graph_pool_handle()
        self.g: Optional[torch.cuda.CUDAGraph] = None
        self.fresh = True

    def begin(self):
        self.g = torch.cuda.CUDAGraph()
        # thread_local: the process group's watchdog / worker threads keep calling into HIP while this thread captures
        self.g.capture_begin(pool=self.pool, capture_error_mode="thread_local")
        self.fresh = True

    def touch(self):
        self.fresh = False

    def end(self):
        self.g.capture_end()
        if not self.fresh:
            self.ops.append(("graph", self.g))
        self.g = None

    def emit(self, op):
        if self.fresh:                 # nothing launched since the last split: the op simply follows the previous one
            self.ops.append(op)
            return
        self.end()
        self.ops.append(op)
        self.begin()

    @property
    def n_graphs(self) -> int:
        return sum(1 for o in self.ops if o[0] == "graph")


class FlatAdam:
    def __init__(self, flat: FlatParams, lr: float = 2e-4, betas=(0.9, 0.999), eps: float = 1e-8):
        self.flat, self.lr, self.betas, self.eps = flat, lr, betas, eps
        self.exp_avg = torch.zeros_like(flat.param)
        self.exp_avg_sq = torch.zeros_like(flat.param)
        self.step_count = 0
        self.step_dev: Optional[torch.Tensor] = None   # int32 device copy of step_count once a graph has been captured

    def step(self, grad_scale: float = 1.0):
        self.step_count += 1
        ops.PARAM_GEN[0] += 1  # the kernel writes the parameter arena behind autograd's version counters
        ops.WEIGHT_GEN[0] += 1
        f = self.flat
        if self.step_dev is not None:   # graph-capturable form: the step number lives on the device (Trainer.capture)
            self.step_dev.add_(1)
            call("df_adam_step_dev", ptr(f.param), ptr(f.grad), ptr(self.exp_avg), ptr(self.exp_avg_sq), f.numel, self.lr,
                 self.betas[0], self.betas[1], self.eps, ptr(self.step_dev), grad_scale, stream())
            return
        call("df_adam_step", ptr(f.param), ptr(f.grad), ptr(self.exp_avg), ptr(self.exp_avg_sq), f.numel, self.lr,
             self.betas[0], self.betas[1], self.eps, self.step_count, grad_scale, stream())

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "lr": self.lr}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        if self.step_dev is not None:
            self.step_dev.fill_(self.step_count)
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.lr = float(sd.get("lr", self.lr))


class Trainer:
    """One data-parallel DeFlow training step: forward (HIP) -> gt gather + deflowLoss (HIP) -> backward (HIP) ->
    all-reduce of the gradient arena (RCCL over xGMI via torch.distributed, or gloo in CPU tests) -> Adam (HIP)."""

    def __init__(self, model: nn.Module, lr: float = 2e-4, process_group=None, loss_fn: str = "deflowLoss",
                 gradient_clip_val: float = 0.0, sync_bn: bool = False, dtype: str = "fp32"):
        if loss_fn not in ("deflowLoss", "ff3dLoss", "zeroflowLoss"):
            raise ValueError(f"unknown loss_fn {loss_fn!r}")
        if dtype not in ("fp32", "bf16"):
            raise ValueError(f"unknown dtype {dtype!r} (fp32, bf16)")
        # dtype="bf16" (BASELINE configs[4] "bf16 MFMA"; Lightning's precision="bf16-mixed"): every GEMM-shaped kernel -- the
        # UNet's convolutions (forward, data gradient, weight gradient) and the point decoder's gate / head GEMMs (forward,
        # backward, gate weight gradients) -- multiplies bf16-rounded operands on v_mfma_f32_{32x32x16,16x16x32}_bf16 with fp32
        # accumulation; master weights, Adam state, activations, BatchNorm statistics, GRU state and gates, and the loss stay
        # fp32 (no loss scaling needed: bf16 has fp32's exponent range)
        self.mfma_bf16 = dtype == "bf16"
        # bf16 STORAGE inside the UNet encoder (ops.BF16_STORE) goes with the mode; DF_BF16_STORE=0 keeps fp32 tensors (A/B, tests)
        self.bf16_store = os.environ.get("DF_BF16_STORE", "1") != "0"
        self.loss_fn = loss_fn
        # Lightning's gradient_clip_val (norm clipping of the synchronised gradient, torch.nn.utils.clip_grad_norm_); 0 = off
        self.gradient_clip_val = float(gradient_clip_val)
        self.model = model
        self.flat = FlatParams(model)
        self.opt = FlatAdam(self.flat, lr)
        import torch.distributed as dist
        self.dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.pg = process_group
        self.world = self.dist.get_world_size(process_group) if self.dist else 1
        # DF_FORCE_COLLECTIVES=1 issues the broadcast / all-reduces also on a 1-rank group: the RCCL path of a 1-GPU box
        self.collective = self.dist is not None and (self.world > 1 or os.environ.get("DF_FORCE_COLLECTIVES") == "1")
        if self.collective:  # identical replicas: broadcast rank 0's arena once
            self.dist.broadcast(self.flat.param, src=0, group=process_group)
        # models whose backward is hand-sequenced (DeFlowFn) deliver gradients phase by phase through the sink
        # sync_bn: BatchNorm batch statistics over all ranks (ops.SyncBN); a process-wide switch of the engine, like
        # torch's convert_sync_batchnorm is of the module tree
        ops.SYNC = ops.SyncBN(self.dist, process_group, self.world) if (sync_bn and self.collective) else None
        self.sink = GradSink(self.flat, self.dist, process_group, self.world, self.collective)
        model._grad_sink = self.sink

    def loss_on_last_forward(self, batch) -> torch.Tensor:
        from .autograd import DeflowLossFn
        st = self.model.last_state
        flow = st["flow"]
        B, N, _ = flow.shape
        gt = torch.empty(B, N, 3, dtype=torch.float32, device=flow.device)
        gtf = batch["flow"].contiguous().float()
        call("df_gather_gt", ptr(gtf), ptr(st["pose_flow"]), ptr(st["idx_c0"]), ptr(st["counts0"]), B, N, ptr(gt), 64,
             stream())
        if self.loss_fn == "deflowLoss":
            return DeflowLossFn.apply(flow, gt, st["counts0"])
        from . import losses
        if self.loss_fn == "zeroflowLoss":
            return losses.zeroflow_loss(flow, gt, st["counts0"])
        cls = batch.get("flow_category_indices")
        if cls is not None:   # classes of the compacted valid points, like gt
            cls = torch.gather(cls.long(), 1, st["idx_c0"].clamp(0, cls.shape[1] - 1))
        return losses.ff3d_loss(flow, gt, st["counts0"], cls)

    def reduce_gradients(self) -> float:
        """Sum the gradient arena over the data-parallel ranks (ONE collective over 27.6 MB); returns the scale that
        turns the sum into DDP's mean (folded into the Adam kernel instead of a separate divide pass)."""
        if self.collective:
            if self.sink.delivered and not self.sink.one_bucket:   # bucketed, already in flight; then whatever did not go through the sink
                rest = sorted(self.sink.slot_of[p.data_ptr()] for p in self.flat.params if not self.sink.was_delivered(p))
                if rest:
                    if self.sink.cap is not None:
                        # capture: these gradients were written into the arena by kernels of the OPEN segment, which `fresh` does not
                        # know about (only deliver() clears it) -- close the segment before their collective (the
                        # all-reduce must not be placed in front of the graph that computes its operand)
                        self.sink.cap.touch()
                    runs = [[rest[0][0], rest[0][0] + rest[0][1]]]
                    for off, n in rest[1:]:       # merge adjacent slots (16-byte padding between them), as deliver() does
                        if off <= runs[-1][1] + 3:
                            runs[-1][1] = max(runs[-1][1], off + n)
                        else:
                            runs.append([off, off + n])
                    self.sink.all_reduce_runs([(lo, hi) for lo, hi in runs])
            else:
                self.sink.all_reduce_runs([(0, self.flat.numel)])
            self.sink.finish()
        return 1.0 / self.world

    def sync_buffers(self):
        """BatchNorm buffers (running_mean / running_var / num_batches_tracked) are rank-local during training; torch DDP
        (Lightning's default, broadcast_buffers=True) makes every rank use rank 0's.  Called before validation and before a
        checkpoint so that all ranks evaluate the SAME model and the file holds the statistics that were validated: the
        float buffers go out as one flat tensor, the integer counters as another (two broadcasts, not ~60)."""
        if not self.collective:
            return
        bufs = [b for b in self.model.buffers()]
        for sel in (lambda b: b.dtype.is_floating_point, lambda b: not b.dtype.is_floating_point):
            group = [b for b in bufs if sel(b)]
            if not group:
                continue
            flat = torch.cat([b.detach().reshape(-1) for b in group])
            self.dist.broadcast(flat, src=0, group=self.pg)
            off = 0
            with torch.no_grad():
                for b in group:
                    b.copy_(flat[off:off + b.numel()].view(b.shape))
                    off += b.numel()
        ops.PARAM_GEN[0] += 1   # cached eval-mode BatchNorm folds are stale

    @staticmethod
    def shard_seed(base_seed: int, rank: int, per_rank_batch: int) -> int:
        """frame pairs are sharded by global sample index: rank r owns samples [r*b, (r+1)*b)"""
        return base_seed + rank * per_rank_batch

    # ---- the whole step as captured HIP graph(s) -------------------------------------------------------------------
    def capture(self, batch) -> None:
        """Capture `step(batch)` -- forward, loss, hand-sequenced backward, Adam: ~410 launches -- as HIP graph(s) on the shapes
        of `batch` (whose tensors become the static inputs; `step_captured(new_batch)` copies into them).  The step is sync-free
        and every host-side quantity the kernels take is constant across steps except Adam's step number, which moves to device
        memory (df_adam_step_dev).  Replaying costs the host ~1 ms instead of the ~44 ms Python needs to enqueue the step; the
        GPU time is about the same -- the point is a host-free step, not a faster one.
        One rank: ONE graph.  Data-parallel ranks: the graph is SPLIT at every gradient bucket and the collectives are issued
        between the segments (SegmentedCapture), still overlapped with the rest of the backward.
        The two warm-up steps capture needs are NOT training steps: parameters, Adam state, BatchNorm buffers and the step
        counters are restored afterwards, so the first `step_captured()` is exactly the step an eager trainer would take."""
        dev = self.flat.param.device
        if ops.SYNC is not None:
            raise RuntimeError("Trainer.capture: sync_bn issues blocking collectives inside the forward and cannot be captured")
        self.opt.step_dev = torch.full((1,), self.opt.step_count, dtype=torch.int32, device=dev)
        self._static = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
        bufs = list(self.model.buffers())
        snap = (self.flat.param.clone(), self.opt.exp_avg.clone(), self.opt.exp_avg_sq.clone(), self.opt.step_count,
                [b.clone() for b in bufs])
        cap_stream = torch.cuda.Stream(device=dev)
        cap_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap_stream):   # warm-up on a side stream, as torch's capture rules ask
            for _ in range(2):
                self.step(self._static)
            with torch.no_grad():
                self.flat.param.copy_(snap[0]); self.opt.exp_avg.copy_(snap[1]); self.opt.exp_avg_sq.copy_(snap[2])
                for b, v in zip(bufs, snap[4]):
                    b.copy_(v)
                self.opt.step_count = snap[3]
                self.opt.step_dev.fill_(snap[3])
            torch.cuda.synchronize()
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            cap = SegmentedCapture()
            self.sink.cap = cap
            try:
                cap.begin()
                cap.touch()
                self._graph_loss = self.step(self._static)
                cap.end()
            finally:
                self.sink.cap = None
        torch.cuda.current_stream().wait_stream(cap_stream)
        self._program = cap.ops
        self._cap = cap      # keeps the pool handle alive
        from .deflow import _CANVASES
        self._canvas_pin = dict(_CANVASES.get(self.model, {}))      # the graphs hold the persistent canvas's ADDRESS: it must outlive the store's entry
        self._graph = cap    # (truthy marker older callers test for)
        # the captured launch sequence was RECORDED, not executed: undo the host-side counter
        self.opt.step_count = snap[3]
        ops.PARAM_GEN[0] += 1
        ops.WEIGHT_GEN[0] += 1

    def step_captured(self, batch=None) -> torch.Tensor:
        """Replay the captured step (optionally on a new batch of the captured shapes).  -> loss tensor of the replay"""
        if batch is not None:
            for k, v in batch.items():
                if isinstance(v, torch.Tensor):
                    if v.shape != self._static[k].shape:
                        raise ValueError(f"step_captured: {k} has shape {tuple(v.shape)}, the graph was captured on {tuple(self._static[k].shape)}")
                    self._static[k].copy_(v, non_blocking=True)
        works = []
        for op in self._program:
            kind = op[0]
            if kind == "graph":
                op[1].replay()
            elif kind == "allreduce":
                for lo, hi in op[1]:
                    works.append(self.dist.all_reduce(self.flat.grad[lo:hi], group=self.pg, async_op=True))
            else:   # "wait": the optimizer segment follows
                for w in works:
                    w.wait()
                works.clear()
        self.opt.step_count += 1
        ops.PARAM_GEN[0] += 1
        ops.WEIGHT_GEN[0] += 1
        return self._graph_loss

    def _side_on(self) -> Optional[bool]:
        """weight-gradient GEMMs on a second stream?  DF_SIDE_STREAM=0/1 decides if set; otherwise on in bf16 mode (with
        data-parallel ranks each gradient phase joins the side stream before its bucket goes out: autograd.deflow_backward)."""
        env = os.environ.get("DF_SIDE_STREAM")
        if env is not None:
            return None if env == "1" else False   # "1": autograd creates the stream itself (every caller, not only Trainer)
        return bool(self.mfma_bf16)

    def _forward_backward(self, batch) -> torch.Tensor:
        """forward -> loss -> backward WITHOUT the autograd engine when the model is the hand-sequenced DeFlow engine: the
        forward keeps its own tape, the loss kernels produce d(flow) directly and autograd.deflow_backward runs on this thread
        (the autograd path -- DeFlowFn / loss.backward(), which every non-Trainer caller uses -- computes the same launches
        from a worker thread; tests/test_gpu_model.py::test_direct_step_equals_autograd_step)."""
        model = self.model
        # fp16x2 convolutions (fp32 mode): a fresh zero-filled pool of max |x| slots for this step, and ONE bound for every weight
        # tensor -- max |p| over the whole parameter arena (any upper bound serves; the convolutions then need no per-call pass)
        ops.amax_pool_reset()
        ops.W_AMAX = None
        ops.WPREP = None
        if ops.h2_active() and self.flat.param.is_cuda:
            wa = ops.amax_slot(self.flat.param.device)
            call("df_absmax", img(self.flat.param.view(1, 1, -1, 4)), ptr(wa), stream())       # (numel is a multiple of 4)
            ops.W_AMAX = wa
            # every conv layer's transposed weights / fp16 planes / row norms of THIS step's parameters in one launch (round 4:
            # ~80 small launches per step before); DF_WPREP=0: the per-call launches
            backbone = getattr(model, "backbone", None)
            if backbone is not None and os.environ.get("DF_WPREP", "1") != "0":
                wp = getattr(self, "_wprep", None)
                if wp and wp.stale():      # a parameter was re-assigned since the table of raw addresses was built
                    wp = None
                if wp is None:
                    convs = [m for m in backbone.modules() if isinstance(m, torch.nn.Conv2d)]
                    wp = self._wprep = ops.WeightPrep(convs, wa) if convs else False
                if wp:
                    wp.run(wa)
                    ops.WPREP = wp
        try:
            return self._forward_backward_impl(batch)
        finally:
            ops.W_AMAX = None
            ops.WPREP = None

    def _forward_backward_impl(self, batch) -> torch.Tensor:
        model = self.model
        if not hasattr(model, "forward_padded") or os.environ.get("DF_TRAINER_AUTOGRAD") == "1":
            model.forward_padded(batch)
            loss = self.loss_on_last_forward(batch)
            loss.backward()
            return loss.detach()
        from .autograd import deflow_backward
        with torch.no_grad():
            model._persist_canvas = True     # one forward, then its backward, per step: the engine may keep its BEV canvas across steps (deflow.py)
            try:
                st = model.forward_padded(batch, engine_tape=True)
            finally:
                model._persist_canvas = False
            flow = st["flow"]
            B, N, _ = flow.shape
            dev = flow.device
            gt = torch.empty(B, N, 3, dtype=torch.float32, device=dev)
            gtf = batch["flow"].contiguous().float()
            call("df_gather_gt", ptr(gtf), ptr(st["pose_flow"]), ptr(st["idx_c0"]), ptr(st["counts0"]), B, N, ptr(gt), 64, stream())
            if self.loss_fn == "deflowLoss":
                nblk = max(1, min(32, (N + 255) // 256))
                partial = torch.empty(B, nblk, 6, dtype=torch.float32, device=dev)
                call("df_deflow_loss_fwd", ptr(flow), ptr(gt), ptr(st["counts0"]), B, N, ptr(partial), nblk, stream())
                bins = torch.empty(B, 6, dtype=torch.float32, device=dev)
                loss = torch.empty(1, dtype=torch.float32, device=dev)
                call("df_deflow_loss_finalize", ptr(partial), B, nblk, ptr(bins), ptr(loss), stream())
                dflow = torch.empty_like(flow)
                if not hasattr(self, "_one") or self._one.device != dev:
                    self._one = torch.ones(1, dtype=torch.float32, device=dev)
                call("df_deflow_loss_bwd", ptr(flow), ptr(gt), ptr(st["counts0"]), B, N, ptr(bins), ptr(self._one), 1.0, ptr(dflow),
                     nblk, stream())
                loss = loss[0]
            else:   # the ablation losses: df_wloss_* (round 5; the torch form of losses.py stays for autograd callers and as the test's twin)
                kind = 1 if self.loss_fn == "zeroflowLoss" else 0
                cls = None
                if kind == 0:
                    cls = batch.get("flow_category_indices")
                    if cls is None:
                        raise ValueError("loss_fn=ff3dLoss needs batch['flow_category_indices'] (labelled scene files)")
                    cls = cls.to(device=dev, dtype=torch.int64).contiguous()
                nblk = max(1, min(32, (N + 255) // 256))
                partial = torch.empty(B, nblk, 2, dtype=torch.float32, device=dev)
                idx_c = st["idx_c0"]
                call("df_wloss_fwd", ptr(flow), ptr(gt), ptr(st["counts0"]), B, N, kind, ptr(cls), ptr(idx_c), 0 if cls is None else cls.shape[1],
                     ptr(partial), nblk, stream())
                bins = torch.empty(B, 2, dtype=torch.float32, device=dev)
                loss = torch.empty(1, dtype=torch.float32, device=dev)
                call("df_wloss_finalize", ptr(partial), B, nblk, ptr(bins), ptr(loss), stream())
                dflow = torch.empty_like(flow)
                call("df_wloss_bwd", ptr(flow), ptr(gt), ptr(st["counts0"]), B, N, kind, ptr(cls), ptr(idx_c), 0 if cls is None else cls.shape[1],
                     ptr(bins), None, 1.0, ptr(dflow), nblk, stream())
                loss = loss[0]
            deflow_backward(model, st.pop("engine"), dflow.contiguous(), self.flat.params, self.sink)
        return loss

    def step(self, batch) -> torch.Tensor:
        self.flat.zero_grad()
        self.sink.begin()
        with ops.mfma_bf16(self.mfma_bf16, self.bf16_store), ops.side_stream(self._side_on(), self.flat.grad.device):
            loss = self._forward_backward(batch)
        scale = self.reduce_gradients()
        if self.gradient_clip_val > 0:       # two launches on the 27.6 MB arena, no host sync
            total_norm = torch.linalg.vector_norm(self.flat.grad) * scale
            self.flat.grad.mul_(torch.clamp(self.gradient_clip_val / (total_norm + 1e-6), max=1.0))
        if self.sink.cap is not None:
            self.sink.cap.touch()
        self.opt.step(grad_scale=scale)
        return loss.detach()
