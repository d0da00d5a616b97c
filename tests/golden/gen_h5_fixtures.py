"""Writes the HDF5 scene fixtures the reader tests use.  Run HERE with the one interpreter of this image that has h5py
(real HDF5 1.10.6, h5py 3.3.0):   /opt/conda/bin/python3.9 tests/golden/gen_h5_fixtures.py

Layout = what OpenSceneFlow's dataprocess/extract_av2.py writes (recalled; the script is not in the reference tree): one
file per scene, one group per LiDAR sweep named by its timestamp, datasets lidar [N,3] f32, ground_mask [N] bool,
pose [4,4] f32 and -- for all but the last sweep -- flow [N,3] f32, flow_is_valid [N] bool, flow_category_indices [N] u8,
ego_motion [4,4] f32; next to the files an index_total.pkl = [[scene_id, timestamp], ...] of the sweeps that have flow.
The fixtures are DATA: seeded random scenes, small, plus av2_mini_expected.npz holding every array as h5py read it back.
One extra file exercises what the reader must handle beyond the writer's defaults: chunked + gzip + shuffle datasets, an
empty dataset, a float64 pose, a scalar attribute."""
import os
import pickle

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "av2_mini", "train")
os.makedirs(ROOT, exist_ok=True)
rng = np.random.default_rng(20240116)
expected, index = {}, []


def pose(i):
    yaw = 0.01 * i
    T = np.eye(4, dtype=np.float32)
    T[:2, :2] = [[np.cos(yaw), -np.sin(yaw)], [np.sin(yaw), np.cos(yaw)]]
    T[:3, 3] = [0.7 * i, 0.05 * i, 0.0]
    return T


def write_scene(name, n_sweeps, chunked=False):
    path = os.path.join(ROOT, name + ".h5")
    if os.path.exists(path):
        os.remove(path)
    t0 = 315968000000000000 + int(rng.integers(0, 10 ** 9))
    with h5py.File(path, "w") as f:
        if chunked:
            f.attrs["note"] = "chunked"
        for i in range(n_sweeps):
            ts = str(t0 + i * 100000000)
            n = int(rng.integers(60, 160)) if not (chunked and i == 1) else 0
            g = f.create_group(ts)
            pc = (rng.normal(0, 12, (n, 3)) * [1, 1, 0.1]).astype(np.float32)
            gm = rng.random(n) < 0.3
            kw = dict(chunks=(max(1, n // 3), 3), compression="gzip", shuffle=True) if (chunked and n) else {}
            g.create_dataset("lidar", data=pc, **kw)
            g.create_dataset("ground_mask", data=gm.astype(bool))
            g.create_dataset("pose", data=pose(i).astype(np.float64 if chunked else np.float32))
            if i + 1 < n_sweeps:
                fl = np.where(rng.random((n, 1)) < 0.1, rng.normal(0, 0.5, (n, 3)), 0).astype(np.float32)
                g.create_dataset("flow", data=fl, **kw)
                g.create_dataset("flow_is_valid", data=(rng.random(n) < 0.95).astype(bool))
                g.create_dataset("flow_category_indices", data=rng.integers(0, 30, n).astype(np.uint8))
                g.create_dataset("ego_motion", data=(np.linalg.inv(pose(i + 1)) @ pose(i)).astype(np.float32))
                index.append([name, ts])
            if chunked and i == 0:   # constructs beyond the writer's defaults that the reader claims to handle
                g.create_dataset("x_int64", data=rng.integers(-2 ** 40, 2 ** 40, 17))
                g.create_dataset("x_int16_be", data=rng.integers(-3000, 3000, (5, 4)).astype(">i2"))
                g.create_dataset("x_f16", data=rng.normal(0, 1, 33).astype(np.float16))
                g.create_dataset("x_f64_be", data=rng.normal(0, 1, (3, 3)).astype(">f8"))
                g.create_dataset("x_chunked_plain", data=rng.normal(0, 1, (50, 7)).astype(np.float32), chunks=(16, 4))
                g.create_dataset("x_fletcher", data=rng.integers(0, 255, (40, 3)).astype(np.uint8), chunks=(16, 3),
                                 fletcher32=True, compression="gzip", compression_opts=9)
                g.create_dataset("x_scalar", data=np.float32(2.5))
                dcpl = h5py.h5p.create(h5py.h5p.DATASET_CREATE)
                dcpl.set_layout(h5py.h5d.COMPACT)
                arr = rng.integers(0, 100, 12).astype(np.int32)
                space = h5py.h5s.create_simple(arr.shape)
                did = h5py.h5d.create(g.id, b"x_compact", h5py.h5t.NATIVE_INT32, space, dcpl=dcpl)
                did.write(h5py.h5s.ALL, h5py.h5s.ALL, arr)
    with h5py.File(path, "r") as f:   # what h5py reads back is the golden
        for ts in f:
            for k in f[ts]:
                expected[f"{name}/{ts}/{k}"] = f[ts][k][()]


write_scene("scene_a", 25)      # > 8 groups: several symbol-table nodes under the root group's B-tree
write_scene("scene_b", 70)      # > 64: a two-level B-tree
write_scene("scene_chunked", 3, chunked=True)
with open(os.path.join(ROOT, "index_total.pkl"), "wb") as f:
    pickle.dump(index, f, protocol=4)
np.savez_compressed(os.path.join(HERE, "av2_mini_expected.npz"), **expected)
print(len(index), "indexed sweeps;", len(expected), "arrays;", {n: os.path.getsize(os.path.join(ROOT, n)) for n in sorted(os.listdir(ROOT))})
