"""The step before the hot path: preprocessed scene files -> NaN-padded batch dicts on the
GPU, sharded over the data-parallel ranks.

Mirrors the pieces of OpenSceneFlow's ``src/dataset.py`` the reference's ``train.py`` wires into its DataLoader
([REF assets/slurm/1_train.sh:16-30]: node-local staging, ``train_data=<dir> num_workers=16``; the module itself is in the
absent submodule, so names and semantics are recalled):

* ``HDF5Dataset(directory)``: ``index_total.pkl`` = [[scene_id, timestamp], ...]; item = sweep + the next sweep of the same
  scene: ``pc0 gm0 pose0 pc1 gm1 pose1`` and, when labelled, ``flow flow_is_valid flow_category_indices ego_motion``.
  Files are read with the in-tree HDF5 reader (``h5scene.py``; h5py is not in this image).
* ``collate_fn_pad``: drops ground points, pads every cloud of the batch to the longest with NaN rows (the padding the
  voxeliser drops again [REF deflow.py:51-52,82-83]); ``flow`` is padded alike, masks / classes with 0.
* ``ShardedSampler``: torch ``DistributedSampler`` semantics (seeded per-epoch shuffle, padded to a multiple of the world
  size, rank r takes indices r, r + world, ...), no torch.distributed dependency.
* ``SceneLoader``: worker processes (torch DataLoader) read and collate ahead of the training loop; batches go
  host-pinned -> HBM on a copy stream and are handed over with an event, so the H2D traffic (2 clouds x ~1 MB per pair)
  overlaps the previous step.
* ``stage_to_local``: the parallel copy of the scene files to node-local scratch that ``1_train.sh`` does with xargs.
"""
from __future__ import annotations

import os
import pickle
import shutil
import threading
from collections import OrderedDict
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, Iterator, List, Optional, Sequence

import torch

from .h5scene import H5File


class HDF5Dataset:
    def __init__(self, directory: str, max_open_files: int = 8, eval: bool = False):
        """``eval=True`` (the evaluation entry): read ``index_eval.pkl`` -- the frames of the official validation benchmark, the ones
        that carry an ``eval_mask`` -- when the directory has one, as upstream's dataset does for ``av2_mode=val`` (recalled: the
        module is in the absent submodule); ``index_total.pkl`` lists every sweep of every scene."""
        self.directory = directory
        name = "index_total.pkl"
        if eval and os.path.exists(os.path.join(directory, "index_eval.pkl")):
            name = "index_eval.pkl"
        self.index_file = name
        with open(os.path.join(directory, name), "rb") as f:
            self.data_index: List[Sequence] = [list(e) for e in pickle.load(f)]
        self._files: "OrderedDict[str, H5File]" = OrderedDict()
        self._lock = threading.Lock()
        self._max_open = max_open_files

    def __len__(self) -> int:
        return len(self.data_index)

    def _file(self, scene_id: str) -> H5File:
        with self._lock:
            f = self._files.get(scene_id)
            if f is None:
                f = H5File(os.path.join(self.directory, f"{scene_id}.h5"))
                f.sweeps = sorted(f.keys(), key=int)          # timestamps in time order
                self._files[scene_id] = f
                while len(self._files) > self._max_open:
                    self._files.popitem(last=False)   # not closed here: a worker may still be reading it; the map is
                                                      # released when the last reference goes
            else:
                self._files.move_to_end(scene_id)
            return f

    def __getitem__(self, index: int) -> Dict[str, object]:
        scene_id, timestamp = self.data_index[index][0], str(self.data_index[index][1])
        f = self._file(scene_id)
        k = f.sweeps.index(timestamp)
        if k + 1 >= len(f.sweeps):
            # upstream's index (create_reading_index) lists EVERY sweep of a scene; the last one has no successor, and
            # upstream's HDF5Dataset steps such an entry back by one sweep instead of failing
            if len(f.sweeps) < 2:
                raise IndexError(f"{scene_id} holds a single sweep: nothing to pair {timestamp} with")
            k -= 1
            timestamp = f.sweeps[k]
        g0, g1 = f[timestamp], f[f.sweeps[k + 1]]
        t = lambda d: torch.from_numpy(d.read())
        item = {"scene_id": scene_id, "timestamp": int(timestamp),
                "pc0": t(g0["lidar"])[:, :3], "gm0": t(g0["ground_mask"]), "pose0": t(g0["pose"]),
                "pc1": t(g1["lidar"])[:, :3], "gm1": t(g1["ground_mask"]), "pose1": t(g1["pose"])}
        if "flow" in g0:
            item.update(flow=t(g0["flow"]), flow_is_valid=t(g0["flow_is_valid"]),
                        flow_category_indices=t(g0["flow_category_indices"]))
        if "ego_motion" in g0:
            item["ego_motion"] = t(g0["ego_motion"])
        if "eval_mask" in g0:          # the benchmark's point mask of the official validation split
            item["eval_mask"] = t(g0["eval_mask"]).reshape(-1).bool()
        return item


def _pad(seqs: List[torch.Tensor], value) -> torch.Tensor:
    n = max(int(s.shape[0]) for s in seqs)
    out = seqs[0].new_full((len(seqs), n) + tuple(seqs[0].shape[1:]), value)
    for i, s in enumerate(seqs):
        out[i, : s.shape[0]] = s
    return out


def collate_fn_pad(batch: List[Dict[str, object]]) -> Dict[str, object]:
    keep0 = [~b["gm0"] for b in batch]
    keep1 = [~b["gm1"] for b in batch]
    res: Dict[str, object] = {
        "pc0": _pad([b["pc0"][k].float() for b, k in zip(batch, keep0)], float("nan")),
        "pc1": _pad([b["pc1"][k].float() for b, k in zip(batch, keep1)], float("nan")),
        "pose0": torch.stack([b["pose0"].float() for b in batch]),
        "pose1": torch.stack([b["pose1"].float() for b in batch]),
        "scene_id": [b["scene_id"] for b in batch], "timestamp": [b["timestamp"] for b in batch],
    }
    if "flow" in batch[0]:
        res["flow"] = _pad([b["flow"][k].float() for b, k in zip(batch, keep0)], float("nan"))
        res["flow_is_valid"] = _pad([b["flow_is_valid"][k] for b, k in zip(batch, keep0)], False)
        res["flow_category_indices"] = _pad([b["flow_category_indices"][k] for b, k in zip(batch, keep0)], 0)
    if "ego_motion" in batch[0]:
        res["ego_motion"] = torch.stack([b["ego_motion"].float() for b in batch])
    if any("eval_mask" in b for b in batch):
        # per SAMPLE: upstream's preprocessing writes the mask on the official evaluation frames only, so a batch of the
        # validation split mixes frames with and without one.  A frame without a mask gets an all-True one (every valid point
        # counts) and has_eval_mask = False, which lets the evaluation skip it where the benchmark would.
        res["eval_mask"] = _pad([(b["eval_mask"][k] if "eval_mask" in b else torch.ones(int(k.sum()), dtype=torch.bool))
                                 for b, k in zip(batch, keep0)], False)
        res["has_eval_mask"] = torch.tensor(["eval_mask" in b for b in batch], dtype=torch.bool)
    return res


class ShardedSampler:
    def __init__(self, n: int, rank: int = 0, world: int = 1, shuffle: bool = True, seed: int = 0, drop_last: bool = False):
        self.n, self.rank, self.world, self.shuffle, self.seed, self.epoch = n, rank, world, shuffle, seed, 0
        self.num_samples = n // world if drop_last else (n + world - 1) // world
        self.total = self.num_samples * world

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def __len__(self) -> int:
        return self.num_samples

    def __iter__(self) -> Iterator[int]:
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            idx = torch.randperm(self.n, generator=g).tolist()
        else:
            idx = list(range(self.n))
        if len(idx) < self.total:                     # pad by wrapping around, as DistributedSampler does
            idx += (idx * ((self.total - len(idx)) // max(len(idx), 1) + 1))[: self.total - len(idx)]
        return iter(idx[: self.total][self.rank: self.total: self.world])


class SceneLoader:
    """for batch in SceneLoader(...): batch tensors are on `device`, ready on the current stream.

    Reading and collating run in `num_workers` worker PROCESSES (torch DataLoader, as in the reference's train.py): the
    training loop launches ~400 kernels per step from Python, and reader threads in the same interpreter starve it through
    the GIL (measured: 44 pairs/s with threads vs the resident-batch rate with processes).  The parent keeps `prefetch`
    batches in flight host-pinned -> HBM on a copy stream and hands each over with an event."""

    def __init__(self, dataset, batch_size: int, sampler: Optional[ShardedSampler] = None, device=None, num_workers: int = 4,
                 prefetch: int = 3, drop_last: bool = True):
        self.ds, self.bs, self.device = dataset, batch_size, device
        self.sampler = sampler or ShardedSampler(len(dataset), shuffle=False)
        self.workers, self.prefetch, self.drop_last = max(0, num_workers), max(1, prefetch), drop_last

    def __len__(self) -> int:
        n = len(self.sampler)
        return n // self.bs if self.drop_last else (n + self.bs - 1) // self.bs

    def __iter__(self):
        from torch.utils.data import DataLoader
        cuda = self.device is not None and torch.device(self.device).type == "cuda"
        if getattr(self, "_dl", None) is None:   # one DataLoader for all epochs: its worker processes persist (a restart
            kw = dict(prefetch_factor=2, persistent_workers=True) if self.workers else {}   # costs seconds per epoch)
            self._dl = DataLoader(self.ds, batch_size=self.bs, sampler=self.sampler, collate_fn=collate_fn_pad,
                                  num_workers=self.workers, pin_memory=cuda, drop_last=self.drop_last, **kw)
        host = iter(self._dl)
        if not cuda:
            yield from host
            return
        copy_stream = torch.cuda.Stream(device=self.device)
        inflight: List = []

        def fetch():
            b = next(host, None)
            if b is None:
                return False
            with torch.cuda.stream(copy_stream):
                dev = {k: (v.to(self.device, non_blocking=True) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            inflight.append((dev, ev, b))          # b: the pinned source stays alive until the copy is consumed
            return True

        while len(inflight) < self.prefetch and fetch():
            pass
        while inflight:
            dev, ev, _ = inflight.pop(0)
            torch.cuda.current_stream().wait_event(ev)
            for v in dev.values():                 # allocated on the copy stream: tie their lifetime to this stream
                if isinstance(v, torch.Tensor):
                    v.record_stream(torch.cuda.current_stream())
            fetch()
            yield dev


def stage_to_local(src: str, dst: str, workers: int = 16) -> int:
    """copy every file under `src` to `dst` (flat, like `find ... | xargs -P16 cp -t` [REF 1_train.sh:16-24]); returns
    the number of bytes copied; files already present with the same size are skipped"""
    os.makedirs(dst, exist_ok=True)
    jobs = []
    for root, _, names in os.walk(src):
        for n in names:
            s, d = os.path.join(root, n), os.path.join(dst, n)
            if not (os.path.exists(d) and os.path.getsize(d) == os.path.getsize(s)):
                jobs.append((s, d))
    with ThreadPoolExecutor(max(1, workers)) as pool:
        list(pool.map(lambda sd: shutil.copyfile(*sd), jobs))
    return sum(os.path.getsize(s) for s, _ in jobs)
