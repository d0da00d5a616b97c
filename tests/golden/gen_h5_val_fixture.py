"""Writes tests/golden/av2_mini/val: one labelled VALIDATION scene whose flow labels are consistent with its ego motion (static
points carry exactly the ego-motion flow, moving objects add their own displacement over every speed bucket of the Argoverse-2
metrics), with points on both sides of the 35 m evaluation range, every label meta-class, unlabelled points and an ``eval_mask``
dataset (the benchmark's point mask).  Run HERE with the interpreter that has h5py:
    /opt/conda/bin/python3.9 tests/golden/gen_h5_val_fixture.py
Layout as tests/golden/gen_h5_fixtures.py (what OpenSceneFlow's dataprocess/extract_av2.py writes; recalled).  DATA only: seeded
random arrays."""
import os
import pickle

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "av2_mini", "val")
os.makedirs(ROOT, exist_ok=True)
rng = np.random.default_rng(20250929)
# label indices (0 = none; av2 AnnotationCategories alphabetical + 1): car, pedestrian, bicycle, truck, bus, bollard, sign, stroller, dog
CATS = np.array([19, 17, 3, 25, 7, 5, 21, 23, 10])


def pose(i):
    yaw = 0.012 * i
    T = np.eye(4)
    T[:2, :2] = [[np.cos(yaw), -np.sin(yaw)], [np.sin(yaw), np.cos(yaw)]]
    T[:3, 3] = [0.9 * i, 0.04 * i * i, 0.0]
    return T


path = os.path.join(ROOT, "scene_val.h5")
if os.path.exists(path):
    os.remove(path)
index = []
t0 = 315970000000000000
N_SWEEPS = 11
with h5py.File(path, "w") as f:
    for i in range(N_SWEEPS):
        ts = str(t0 + i * 100000000)
        n = int(rng.integers(350, 450))
        g = f.create_group(ts)
        pc = np.concatenate([rng.uniform(-48, 48, (n, 2)), rng.normal(0, 0.6, (n, 1))], 1).astype(np.float32)
        gm = rng.random(n) < 0.25
        g.create_dataset("lidar", data=pc)
        g.create_dataset("ground_mask", data=gm.astype(bool))
        g.create_dataset("pose", data=pose(i).astype(np.float32))
        if i + 1 < N_SWEEPS:
            ego = (np.linalg.inv(pose(i + 1)) @ pose(i)).astype(np.float32)
            rigid = pc @ ego[:3, :3].T + ego[:3, 3] - pc
            cats = np.where(rng.random(n) < 0.5, 0, CATS[rng.integers(0, len(CATS), n)]).astype(np.uint8)
            moving = (cats != 0) & (rng.random(n) < 0.45)
            speed = rng.choice([0.03, 0.06, 0.3, 0.9, 1.7, 2.6], n)                  # m per frame: static bucket .. the open bucket
            ang = rng.uniform(0, 2 * np.pi, n)
            obj = np.stack([speed * np.cos(ang), speed * np.sin(ang), np.zeros(n)], 1) * moving[:, None]
            g.create_dataset("flow", data=(rigid + obj).astype(np.float32))
            g.create_dataset("flow_is_valid", data=(rng.random(n) < 0.95).astype(bool))
            g.create_dataset("flow_category_indices", data=cats)
            g.create_dataset("ego_motion", data=ego)
            g.create_dataset("eval_mask", data=((np.abs(pc[:, :2]) <= 45.0).all(1) & ~gm).astype(bool))
            index.append(["scene_val", ts])
with open(os.path.join(ROOT, "index_total.pkl"), "wb") as f:
    pickle.dump(index, f, protocol=4)
print(len(index), "indexed sweeps,", os.path.getsize(path), "bytes")
