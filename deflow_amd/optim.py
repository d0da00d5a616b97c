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
from ._lib import call, ptr, stream


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

    def begin(self):
        self.delivered.clear()
        self.works.clear()

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
        if self.collective:
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
            for lo, hi in runs:
                self.works.append(self.dist.all_reduce(self.flat.grad[lo:hi], group=self.pg, async_op=True))

    def was_delivered(self, p) -> bool:
        return p.data_ptr() in self.delivered

    def finish(self):
        for w in self.works:
            w.wait()
        self.works.clear()


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
            if self.sink.delivered:   # bucketed, already in flight: wait; then whatever did not go through the sink
                self.sink.finish()
                rest = [p for p in self.flat.params if not self.sink.was_delivered(p)]
                if rest:
                    for p in rest:
                        self.dist.all_reduce(p.grad if p.grad.is_contiguous() else p.grad.permute(0, 2, 3, 1), group=self.pg)
            else:
                self.dist.all_reduce(self.flat.grad, group=self.pg)
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

    # ---- the whole step as ONE captured HIP graph ----------------------------------------------------------------
    def capture(self, batch) -> None:
        """Capture `step(batch)` -- forward, loss, hand-sequenced backward, Adam: ~410 launches -- as one HIP graph on the
        shapes of `batch` (whose tensors become the graph's static inputs; `step_captured(new_batch)` copies into them).  The
        step is sync-free and every host-side quantity the kernels take is constant across steps except Adam's step number,
        which moves to device memory (df_adam_step_dev).  Replaying costs the host 0.1-1 ms instead of the ~44 ms it takes
        Python to enqueue the step; the GPU time is about the same (fp32: equal; bf16, where the step runs on two streams:
        50.6 ms replayed vs 48.9 eager) -- the point is a host-free step, not a faster one.  One rank only: the RCCL buckets issued from inside the backward are not
        captured."""
        if self.collective:
            raise RuntimeError("Trainer.capture: graph capture of the data-parallel step (RCCL inside the backward) is not supported")
        dev = self.flat.param.device
        self.opt.step_dev = torch.full((1,), self.opt.step_count, dtype=torch.int32, device=dev)
        self._static = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):        # warm-up on a side stream, as torch's capture rules ask; these ARE training steps
            for _ in range(2):
                self.step(self._static)
        torch.cuda.current_stream().wait_stream(side)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self._graph_loss = self.step(self._static)
        # the captured launch sequence was RECORDED, not executed; the two warm-up steps and the capture call advanced the host
        # counter by 3, the device counter by 2
        self.opt.step_count -= 1

    def step_captured(self, batch=None) -> torch.Tensor:
        """Replay the captured step (optionally on a new batch of the captured shapes).  -> loss tensor of the replay"""
        if batch is not None:
            for k, v in batch.items():
                if isinstance(v, torch.Tensor):
                    if v.shape != self._static[k].shape:
                        raise ValueError(f"step_captured: {k} has shape {tuple(v.shape)}, the graph was captured on {tuple(self._static[k].shape)}")
                    self._static[k].copy_(v, non_blocking=True)
        self._graph.replay()
        self.opt.step_count += 1
        ops.PARAM_GEN[0] += 1
        return self._graph_loss

    def _side_on(self) -> Optional[bool]:
        """weight-gradient GEMMs on a second stream?  DF_SIDE_STREAM=0/1 decides if set; otherwise on in bf16 mode on one GPU
        (with data-parallel ranks the phase-by-phase gradient delivery -- the overlapped all-reduce -- needs them in order)."""
        env = os.environ.get("DF_SIDE_STREAM")
        if env is not None:
            return None if env == "1" else False   # "1": autograd creates the stream itself (every caller, not only Trainer)
        return bool(self.mfma_bf16 and not self.collective)

    def step(self, batch) -> torch.Tensor:
        self.flat.zero_grad()
        self.sink.begin()
        with ops.mfma_bf16(self.mfma_bf16), ops.side_stream(self._side_on(), self.flat.grad.device):
            self.model.forward_padded(batch)
            loss = self.loss_on_last_forward(batch)
            loss.backward()
        scale = self.reduce_gradients()
        if self.gradient_clip_val > 0:       # two launches on the 27.6 MB arena, no host sync
            total_norm = torch.linalg.vector_norm(self.flat.grad) * scale
            self.flat.grad.mul_(torch.clamp(self.gradient_clip_val / (total_norm + 1e-6), max=1.0))
        self.opt.step(grad_scale=scale)
        return loss.detach()
