"""CPU: the in-tree HDF5 reader against fixtures written by real h5py (tests/golden/gen_h5_fixtures.py), and the
dataset / collate / sampler / loader built on it."""
import os
import pickle

import numpy as np
import pytest
import torch

from deflow_amd.data import HDF5Dataset, SceneLoader, ShardedSampler, collate_fn_pad, stage_to_local
from deflow_amd.h5scene import H5File, H5FormatError

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "golden", "av2_mini", "train")


@pytest.fixture(scope="module")
def expected():
    return np.load(os.path.join(HERE, "golden", "av2_mini_expected.npz"))


@pytest.mark.parametrize("scene,nsweeps", [("scene_a", 25), ("scene_b", 70), ("scene_chunked", 3)])
def test_reader_equals_h5py(expected, scene, nsweeps):
    """every dataset of every sweep: dtype, shape and bytes equal what h5py read back when the fixture was written --
    contiguous f32 / f64 / u8 / bool-enum data, multi-node and two-level group B-trees, chunked + gzip + shuffle, empty"""
    with H5File(os.path.join(ROOT, scene + ".h5")) as f:
        sweeps = list(f.keys())
        assert len(sweeps) == nsweeps and sweeps == sorted(sweeps)
        n = 0
        for ts in sweeps:
            for name, d in f[ts].items():
                want = expected[f"{scene}/{ts}/{name}"]
                got = d.read()
                assert got.dtype == want.dtype and got.shape == want.shape == d.shape, (ts, name)
                assert np.array_equal(got, want), (ts, name)
                n += 1
        assert n == sum(1 for k in expected.files if k.startswith(scene + "/"))


def test_reader_rejects_what_it_does_not_implement(tmp_path):
    p = tmp_path / "junk.h5"
    p.write_bytes(b"not hdf5" * 200)
    with pytest.raises(H5FormatError):
        H5File(str(p))
    raw = bytearray(open(os.path.join(ROOT, "scene_a.h5"), "rb").read())
    raw[8] = 3                                   # superblock version 3 (libver='latest')
    q = tmp_path / "v3.h5"
    q.write_bytes(bytes(raw))
    with pytest.raises(H5FormatError, match="superblock version 3"):
        H5File(str(q))


def test_dataset_pairs_a_sweep_with_its_successor(expected):
    ds = HDF5Dataset(ROOT)
    index = pickle.load(open(os.path.join(ROOT, "index_total.pkl"), "rb"))
    assert len(ds) == len(index) == 95
    for i in (0, 23, 24, 60, 94):
        scene, ts = index[i]
        it = ds[i]
        assert it["scene_id"] == scene and it["timestamp"] == int(ts)
        with H5File(os.path.join(ROOT, scene + ".h5")) as f:
            nxt = sorted(f.keys(), key=int)[sorted(f.keys(), key=int).index(ts) + 1]
        assert torch.equal(it["pc0"], torch.from_numpy(expected[f"{scene}/{ts}/lidar"]))
        assert torch.equal(it["pc1"], torch.from_numpy(expected[f"{scene}/{nxt}/lidar"]))
        assert torch.equal(it["pose1"].double(), torch.from_numpy(expected[f"{scene}/{nxt}/pose"]).double())
        assert torch.equal(it["flow"], torch.from_numpy(expected[f"{scene}/{ts}/flow"]))
        assert it["gm0"].dtype == torch.bool and it["flow_category_indices"].dtype == torch.uint8
    # upstream's index lists EVERY sweep of a scene (create_reading_index); the last one has no successor and upstream's
    # dataset steps it back by one sweep (raising here aborts the first epoch on real preprocessed data)
    sweeps_a = sorted((k.split("/")[1] for k in expected.files if k.startswith("scene_a/") and k.endswith("/lidar")), key=int)
    ds.data_index.append(["scene_a", sweeps_a[-1]])
    last = ds[len(ds) - 1]
    assert last["timestamp"] == int(sweeps_a[-2])
    assert torch.equal(last["pc0"], torch.from_numpy(expected[f"scene_a/{sweeps_a[-2]}/lidar"]))
    assert torch.equal(last["pc1"], torch.from_numpy(expected[f"scene_a/{sweeps_a[-1]}/lidar"]))


def test_index_listing_every_sweep_loads_end_to_end(tmp_path):
    """index_total.pkl as upstream's preprocessing writes it -- all timestamps of all scenes -- through the loader"""
    import shutil
    from deflow_amd.h5scene import H5File as HF
    d = tmp_path / "full"
    d.mkdir()
    index = []
    for scene in ("scene_a", "scene_b"):
        shutil.copy(os.path.join(ROOT, scene + ".h5"), d / (scene + ".h5"))
        with HF(os.path.join(ROOT, scene + ".h5")) as f:
            index += [[scene, ts] for ts in sorted(f.keys(), key=int)]
    pickle.dump(index, open(d / "index_total.pkl", "wb"))
    ds = HDF5Dataset(str(d))
    assert len(ds) == 95          # 25 + 70 sweeps, the two scene-final ones included
    n = 0
    for batch in SceneLoader(ds, batch_size=8, num_workers=0, device="cpu", drop_last=False):
        n += batch["pc0"].shape[0]
    assert n == 95


def test_collate_drops_ground_and_pads_with_nan():
    ds = HDF5Dataset(ROOT)
    items = [ds[0], ds[30], ds[93], ds[94]]      # scene_chunked's sweep 1 is empty: item 93 has no pc1, item 94 no pc0
    b = collate_fn_pad(items)
    n0 = [int((~it["gm0"]).sum()) for it in items]
    n1 = [int((~it["gm1"]).sum()) for it in items]
    assert b["pc0"].shape == (4, max(n0), 3) and b["pc1"].shape == (4, max(n1), 3) and b["flow"].shape == b["pc0"].shape
    assert len(b["pose0"]) == 4 and b["pose0"].dtype == torch.float32 and b["ego_motion"].shape == (4, 4, 4)
    for i, it in enumerate(items):
        assert torch.equal(b["pc0"][i, : n0[i]], it["pc0"][~it["gm0"]])
        assert torch.isnan(b["pc0"][i, n0[i]:]).all() and torch.isnan(b["flow"][i, n0[i]:]).all()
        assert torch.equal(b["flow"][i, : n0[i]], it["flow"][~it["gm0"]])
        assert not b["flow_is_valid"][i, n0[i]:].any()
        assert torch.isnan(b["pc1"][i, n1[i]:]).all()
    assert n1[2] == 0 and n0[3] == 0


@pytest.mark.parametrize("n,world", [(95, 8), (95, 2), (7, 4), (3, 8)])
def test_sharded_sampler_is_torch_distributed_sampler(n, world):
    from torch.utils.data import DistributedSampler
    data = list(range(n))
    for epoch in (0, 3):
        seen = []
        for rank in range(world):
            ref = DistributedSampler(data, num_replicas=world, rank=rank, shuffle=True, seed=11)
            ref.set_epoch(epoch)
            mine = ShardedSampler(n, rank, world, shuffle=True, seed=11)
            mine.set_epoch(epoch)
            assert list(mine) == list(ref) and len(mine) == len(ref)
            seen += list(mine)
        assert set(seen) == set(data)
    assert list(ShardedSampler(n, 1 % world, world, shuffle=False)) == list(DistributedSampler(data, world, 1 % world, shuffle=False))


def _worker_processes_usable() -> bool:
    """torch DataLoader workers hand batches over through shared memory: needs a writable /dev/shm"""
    return os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK)


@pytest.mark.skipif(not _worker_processes_usable(), reason="no writable /dev/shm for DataLoader worker processes")
def test_loader_prefetches_in_order_and_surfaces_errors():
    ds = HDF5Dataset(ROOT)
    sampler = ShardedSampler(len(ds), rank=1, world=2, shuffle=True, seed=5)
    loader = SceneLoader(ds, 4, sampler, device=None, num_workers=3, prefetch=2)
    got = list(loader)
    idx = list(sampler)
    assert len(got) == len(loader) == len(idx) // 4
    for k, b in enumerate(got):
        want = collate_fn_pad([ds[i] for i in idx[4 * k: 4 * k + 4]])
        assert b["timestamp"] == want["timestamp"]
        assert torch.equal(torch.nan_to_num(b["pc0"]), torch.nan_to_num(want["pc0"]))
    it = iter(SceneLoader(ds, 4, sampler, device=None, num_workers=2))   # abandoning an iterator must not hang the workers
    next(it)
    it.close()
    ds.data_index[idx[0]] = ["no_such_scene", 1]
    with pytest.raises(FileNotFoundError):
        list(SceneLoader(ds, 4, sampler, device=None))


def test_loader_in_process():
    """num_workers=0: same batches, read in the calling process"""
    ds = HDF5Dataset(ROOT)
    sampler = ShardedSampler(len(ds), rank=0, world=4, shuffle=False)
    got = list(SceneLoader(ds, 5, sampler, device=None, num_workers=0, drop_last=False))
    idx = list(sampler)
    assert len(got) == (len(idx) + 4) // 5 and sum(len(b["timestamp"]) for b in got) == len(idx)
    want = collate_fn_pad([ds[i] for i in idx[:5]])
    assert got[0]["timestamp"] == want["timestamp"] and torch.equal(torch.nan_to_num(got[0]["pc1"]), torch.nan_to_num(want["pc1"]))


def test_stage_to_local(tmp_path):
    n = stage_to_local(ROOT, str(tmp_path / "scratch"), workers=4)
    names = sorted(os.listdir(tmp_path / "scratch"))
    assert names == sorted(os.listdir(ROOT)) and n == sum(os.path.getsize(os.path.join(ROOT, x)) for x in names)
    assert stage_to_local(ROOT, str(tmp_path / "scratch")) == 0          # second call: everything already there
    assert len(HDF5Dataset(str(tmp_path / "scratch"))) == 95
