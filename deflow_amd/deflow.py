"""``model=deflow`` plugin: same constructor, batch-dict in / result-dict out, submodule names and checkpoint
loader as the reference ([REF deflow.py:20-113]), backed by the MI355X HIP engine.

Differences from the reference that do not change results: the two Python loops over the batch
([REF deflow.py:60; decoder.py:192]) are batched into single kernel launches; the per-sample lists in the
result dict are zero-copy row slices of padded device buffers; exactly one host sync (reading the per-sample
valid-point counts) happens per forward, after every kernel has been queued.
"""
from __future__ import annotations

import os
import threading
import weakref
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import ops
from ._lib import DfImg, call, img, ptr, stream, ver as _ver
from .autograd import DeFlowFn
from .decoder import ConvGRUDecoder, LinearDecoder, PointSet
from .encoder import DynamicEmbedder, canvas_alloc
from .timer import Timing
from .unet import FastFlow3DUNet


POSE_INVERSE = "rigid"   # "rigid" | "general" (torch.linalg.inv): see cal_pose0to1


def cal_pose0to1(pose0: torch.Tensor, pose1: torch.Tensor, form: Optional[str] = None) -> torch.Tensor:
    """inv(pose1) @ pose0 -- 4x4 host-side plumbing ([REF deflow.py:18,67]; the helper itself is in the un-vendored
    OpenSceneFlow submodule: UNPINNED).  "rigid" (default, what upstream is recalled to do) inverts pose1 in closed form --
    [R^T | R^T (-t)] -- and multiplies in that dtype; "general" is torch.linalg.inv(pose1) @ pose0 (rounds 1-2).  The two differ
    in the last bits of T, which can move a point sitting on a cell edge into the neighbouring pillar; both are tested."""
    form = POSE_INVERSE if form is None else form
    if form == "general":
        return torch.linalg.inv(pose1) @ pose0
    if form != "rigid":
        raise ValueError(f"unknown pose inverse form {form!r} (rigid, general)")
    inv = torch.eye(4, dtype=pose1.dtype, device=pose1.device)
    inv[:3, :3] = pose1[:3, :3].T
    inv[:3, 3] = (pose1[:3, :3].T * -pose1[:3, 3]).sum(axis=1)
    return inv @ pose0.type(inv.dtype)


_CANVASES: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()   # model -> {shape key: (canvas, occupancy words)}: persistent canvases
# ... and who used an entry last: {"stream", "event" (recorded behind the forward's last reader; None under capture), "captured", "tick"}.
# A persistent canvas is ONE buffer per (model, shape): two forwards of a model that may overlap on the GPU -- another stream, a second host
# thread driving its own stream -- must not share it.  A forward takes the entry only on the stream that used it last, or once that
# use has finished (event.query()); otherwise it falls back to a fresh dense canvas, as every forward did before round 5.  Entries
# whose address a HIP graph may hold (created or used under capture) are never evicted, nothing is evicted during a capture, and
# eviction removes the least recently used entry -- not the whole store.
_CANVAS_META: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()
_CANVAS_LOCK = threading.Lock()
_CANVAS_MAX = 4          # a handful of shapes at most (train / eval batch sizes)


def _canvas_take(model, key, make):
    """-> (canvas, occs) of the persistent entry for `key`, or None when the entry is busy elsewhere (the caller then writes a dense canvas)"""
    stream = torch.cuda.current_stream()
    capturing = torch.cuda.is_current_stream_capturing()
    with _CANVAS_LOCK:
        store = _CANVASES.setdefault(model, {})    # (outside the module's __dict__: deepcopy / pickling must not drag 1 GB buffers along)
        meta = _CANVAS_META.setdefault(model, {})
        ent, m = store.get(key), meta.get(key)
        if ent is None:
            if len(store) >= _CANVAS_MAX and not capturing:
                free = [k for k in store if not meta[k]["captured"] and (meta[k]["event"] is None or meta[k]["event"].query())]
                if free:
                    old = min(free, key=lambda k: meta[k]["tick"])
                    del store[old], meta[old]
            ent = store[key] = make()
            m = meta[key] = {"stream": stream, "event": None, "captured": capturing, "tick": 0, "busy": False, "owner": 0}
        else:
            same = m["stream"] == stream
            if m["busy"] and m["owner"] != threading.get_ident():
                return None                         # another host thread is inside a forward (or a trainer step) on this entry
            if not same:
                ev = m["event"]
                if capturing:
                    pass                            # (the capturer ordered its stream behind the earlier work: cap_stream.wait_stream(...))
                elif m["captured"] or (ev is not None and not ev.query()):
                    return None                     # in flight on another stream, or frozen into a graph that replays elsewhere
                m["stream"] = stream
            m["captured"] = m["captured"] or capturing
        m["busy"] = True
        m["owner"] = threading.get_ident()
        m["tick"] = max((x["tick"] for x in meta.values()), default=0) + 1
        return ent


def _canvas_release(model, key):
    with _CANVAS_LOCK:
        m = _CANVAS_META.get(model, {}).get(key)
        if m is None:
            return
        m["busy"] = False
        if torch.cuda.is_current_stream_capturing():
            m["event"] = None
        else:
            if m["event"] is None:
                m["event"] = torch.cuda.Event()
            m["event"].record(torch.cuda.current_stream())


class DeFlow(nn.Module):
    def __init__(self, voxel_size=[0.2, 0.2, 6], point_cloud_range=[-51.2, -51.2, -3, 51.2, 51.2, 3],
                 grid_feature_size=[512, 512], decoder_option="gru", num_iters=4, align_corners=False):
        super().__init__()
        self.embedder = DynamicEmbedder(voxel_size=voxel_size, pseudo_image_dims=grid_feature_size,
                                        point_cloud_range=point_cloud_range, feat_channels=32)
        self.backbone = FastFlow3DUNet(align_corners=align_corners)
        if decoder_option == "gru":
            self.head = ConvGRUDecoder(num_iters=num_iters)
        elif decoder_option == "linear":
            self.head = LinearDecoder()
        else:
            raise ValueError(f"unknown decoder_option {decoder_option!r}")
        self.timer = Timing()
        self.timer.start("Total")
        self._state_tmp: Optional[dict] = None
        self.inference_dtype = "fp32"  # "bf16": eval-mode forwards run the UNet on bf16 MFMA (BASELINE configs[4])
        self.last_state: Optional[dict] = None  # padded device-side tensors of the last forward (fast trainer path)

    def load_from_checkpoint(self, ckpt_path):
        from .ckpt import load_checkpoint   # torch.load that survives omegaconf / Lightning objects in the file
        ckpt = load_checkpoint(ckpt_path)["state_dict"]
        state_dict = {k[len("model."):]: v for k, v in ckpt.items() if k.startswith("model.")}
        print("\nLoading... model weight from: ", ckpt_path, "\n")
        res = self.load_state_dict(state_dict=state_dict, strict=False)
        ops.invalidate_weight_caches()      # (load_state_dict bumps the version counters the caches check; belt and braces)
        return res

    # ------------------------------------------------------------------------------------------------
    def _run(self, pc0s: torch.Tensor, pc1s: torch.Tensor, train: bool, save: bool):
        """Engine: pillarise both clouds into one [B,H,W,64] buffer, UNet, decoder.  -> padded flow, state."""
        with ops.deferred_tracked():      # (the 18 num_batches_tracked increments of a training forward: one launch at the end)
            return self._run_impl(pc0s, pc1s, train, save)

    def _run_impl(self, pc0s: torch.Tensor, pc1s: torch.Tensor, train: bool, save: bool):
        ckey = [None]
        try:
            return self._run_canvas(pc0s, pc1s, train, save, ckey)
        finally:
            if ckey[0] is not None and not save:
                # a no-grad forward is done with the canvas when its decoder has been queued.  (The trainer's step keeps reading it
                # in the backward: optim.Trainer steps one stream, one thread -- its entry is never offered to anybody else.)
                _canvas_release(self, ckey[0])

    def _run_canvas(self, pc0s: torch.Tensor, pc1s: torch.Tensor, train: bool, save: bool, ckey_out: list):
        emb = self.embedder
        B = pc0s.shape[0]
        dev = pc0s.device
        merged = not save and pc0s.shape == pc1s.shape and os.environ.get("DF_MERGE_CLOUDS") != "0"
        # round 5: a PERSISTENT canvas where the engine owns its lifetime -- no-grad forwards and the trainer's one-forward-one-
        # backward step (`_persist_canvas`, set by optim.Trainer) -- so that the band kernels write occupied cells only instead of
        # streaming zeros into the 87 % empty ones (csrc/pillar_bands.hip, df_pillar2_band_sp).  Autograd users (several forwards may
        # be alive) get a fresh, fully written buffer as before.  DF_CANVAS_PERSIST=0: always the dense form.
        persist = ((not save) or getattr(self, "_persist_canvas", False)) and os.environ.get("DF_CANVAS_PERSIST", "1") != "0"
        occs = (None, None)
        ckey = None
        if persist:
            ckey = (B, emb.H, emb.W, str(dev), merged, emb.bands(2 * B if merged else B))
            pc = _canvas_take(self, ckey, lambda: (torch.zeros(B, emb.H, emb.W, 64, dtype=torch.float32, device=dev),
                                                   (emb.occ_alloc(2 * B, dev), None) if merged else (emb.occ_alloc(B, dev), emb.occ_alloc(B, dev))))
            if pc is None:       # the entry is in use on another stream / by another thread: this forward writes its own dense canvas
                persist, ckey = False, None
            ckey_out[0] = ckey
        if persist:
            bstar, occs = pc
            ops.wrote(bstar)      # (whatever max |x| record the previous forward's consumers left on the tensor is void)
        else:
            with ops.timed("canvas_zero_fill"):   # (first-generation pillariser only: the band pipeline writes its own zeros)
                bstar = canvas_alloc(B, emb.H, emb.W, 64, device=dev)
        self.timer[1].start("Voxelization")
        # fp16x2 training: an a-priori bound of max |canvas| comes out of the feature net's BatchNorm finalisation (one slot, both clouds)
        emb.canvas_bound = ops.amax_slot(dev) if (train and save and ops.h2_active() and ops.SYNC is None) else None
        # every other forward on a persistent canvas (evaluation, no-grad training forwards): the band pass MEASURES the canvas maximum
        # as it writes (one atomic per wavefront), so that the first encoder conv and the skip conv on the canvas take their fp16x2
        # forms there too -- they ran on the fp32 MFMA for want of a bound.  From B = 8 up only (measured, tools/bench_infer.py: B = 16
        # forward 21.45 -> 21.05 ms; at B = 1 the eager forward does not move and its HIP-graph replay gets 3 % SLOWER -- the fp16x2
        # DMA-tile forms of those two layers do not beat the fp32 MFMA on one pair's pixels).  DF_CANVAS_AMAX=0: never
        measured = ops.amax_slot(dev) if (persist and B >= 8 and emb.canvas_bound is None and ops.h2_active()
                                          and os.environ.get("DF_CANVAS_AMAX", "1") != "0") else None
        if merged:
            # no tape to keep (inference, no-grad forwards): both clouds go through the pillar pipeline as ONE set of 2B
            # samples writing the two channel halves of bstar -- half the launches of the ~15-kernel pipeline, which is
            # what a B = 1 forward spends there.  Same arithmetic sample by sample (BatchNorm statistics are per sample).
            both = emb.pillarize(torch.cat([pc0s, pc1s], 0), DfImg(bstar.data_ptr(), 2 * B, emb.H, emb.W, 32, 64, B,
                                                                  bstar.stride(0), 32), train, need_cells=False, occ=occs[0], amax=measured)
            p0, p1 = both.split(B)
        else:
            p0 = emb.pillarize(pc0s, img(bstar, 32, 0), train, need_cells=save, occ=occs[0], amax=measured)
            p1 = emb.pillarize(pc1s, img(bstar, 32, 32), train, need_cells=save, occ=occs[1], amax=measured)
        self.timer[1].stop()
        if emb.canvas_bound is not None:
            bstar._df_amax = (emb.canvas_bound, _ver(bstar))
            emb.canvas_bound = None
        elif measured is not None:
            bstar._df_amax = (measured, _ver(bstar))
        self.timer[2].start("Encoder")
        tape: Optional[list] = [] if save else None
        if self.inference_dtype == "bf16" and not train and not save:
            # BASELINE configs[4]: UNet on bf16 MFMA (fp32 accumulation and epilogues); pillars and decoder stay fp32
            v = self.backbone.run_bf16(bstar)
        else:
            # `v` is consumed by the decoder's gather alone: only pc0's occupied cells of it are computed
            sparse_out = os.environ.get("DF_DENSE_CANVAS_GRAD") != "1" and isinstance(self.backbone, FastFlow3DUNet)
            v = self.backbone.run(bstar, train, tape, out_cells=p0 if sparse_out else None)
        self.timer[2].stop()
        self.timer[3].start("Decoder")
        ps = PointSet(p0.coords_c, p0.offs_c, p0.counts, p0.idx_sorted, p0.cell_rng, p0.cpos)
        if self.inference_dtype == "bf16" and not train and not save and hasattr(self.head, "run_bf16"):
            flow, sv = self.head.run_bf16(img(bstar), img(v), ps)
        else:
            flow, sv = self.head.run(img(bstar), img(v), ps, save)
        self.timer[3].stop()
        return flow, {"bstar": bstar, "v": v, "p0": p0, "p1": p1, "ps": ps, "tape": tape, "sv": sv}

    def forward_padded(self, batch: Dict[str, torch.Tensor], engine_tape: bool = False) -> dict:
        """Sync-free forward: returns (and stores in ``last_state``) padded device tensors -- flow [B,N,3],
        pose_flow [B,N,3], counts0/counts1 [B] i32, idx_c0 [B,N] i64 ... -- without reading anything back to the host.
        engine_tape: keep the engine's own tape in ``last_state["engine"]`` instead of building an autograd node (the caller --
        optim.Trainer -- runs autograd.deflow_backward on it directly; call under torch.no_grad())."""
        self.timer[0].start("Data Preprocess")
        pc0 = batch["pc0"].contiguous().float()
        pc1s = batch["pc1"].contiguous().float()
        B, N, _ = pc0.shape
        self.timer[0][0].start("pose")
        with torch.no_grad():
            if "ego_motion" in batch:
                T = batch["ego_motion"]
            else:
                T = torch.stack([cal_pose0to1(batch["pose0"][b], batch["pose1"][b]) for b in range(len(batch["pose0"]))])
            T = T.to(device=pc0.device, dtype=torch.float32).contiguous()
        self.timer[0][0].stop()
        self.timer[0][1].start("transform")
        pc0s = torch.empty_like(pc0)
        pose_flow = torch.empty_like(pc0)
        call("df_ego_transform", ptr(pc0), ptr(T), B, N, ptr(pc0s), ptr(pose_flow), stream())
        self.timer[0][1].stop()
        self.timer[0].stop()

        train = self.training
        params = [p for p in self.parameters()]
        # differentiable whenever autograd is recording, in training AND in eval mode (frozen BatchNorm), as the reference
        # nn.Module is; inference callers wrap the call in torch.no_grad() (eval.py, bench.py do) and get the tape-less path
        if engine_tape:
            flow, state = self._run(pc0s, pc1s, train=train, save=True)
        elif torch.is_grad_enabled() and any(p.requires_grad for p in params):
            flow = DeFlowFn.apply(self, pc0s, pc1s, *params)
            state = self._state_tmp
            self._state_tmp = None
        else:
            with torch.no_grad():
                flow, state = self._run(pc0s, pc1s, train=train, save=False)
        p0, p1 = state["p0"], state["p1"]
        self.last_state = {"flow": flow, "pose_flow": pose_flow, "counts0": p0.counts, "counts1": p1.counts,
                           "idx_c0": p0.idx_c, "pc0s": pc0s, "p0": p0, "p1": p1}
        if engine_tape:
            self.last_state["engine"] = state
        return self.last_state

    def forward(self, batch: Dict[str, torch.Tensor]):
        """input: batch dict [pc0, pc1, pose0, pose1(, ego_motion)]; output: flow, pose_flow and valid indices."""
        st = self.forward_padded(batch)
        flow, pose_flow, p0, p1 = st["flow"], st["pose_flow"], st["p0"], st["p1"]
        B = flow.shape[0]
        # the single host sync of the forward: per-sample valid-point counts
        m = torch.stack([p0.counts, p1.counts]).tolist()
        m0, m1 = m[0], m[1]
        return {
            "flow": [flow[b, :m0[b]] for b in range(B)],
            "pose_flow": [pose_flow[b] for b in range(B)],
            "pc0_valid_point_idxes": [p0.idx_c[b, :m0[b]] for b in range(B)],
            "pc0_points_lst": [p0.points_c[b, :m0[b]] for b in range(B)],
            "pc1_valid_point_idxes": [p1.idx_c[b, :m1[b]] for b in range(B)],
            "pc1_points_lst": [p1.points_c[b, :m1[b]] for b in range(B)],
        }
