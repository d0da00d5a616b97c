"""Seeded synthetic Argoverse-2-shaped frame pairs: no dataset is reachable offline.

Per cloud N rows; xy ~ N(0, 20 m) (about 2 % outside the +-51.2 m range), z ~ U(-3.3, 3.3) (about 9 % outside),
the last 2 % of rows NaN padding; pc1 = rigid(pc0) + flow + N(0, 0.02); ego yaw ~ U(-2, 2) deg, t_x ~ U(0, 1.5) m;
10 % of the points dynamic with speed ~ U(0, 2) m/frame so all three deflowLoss bins are populated."""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch


def _rigid(p: torch.Tensor, T) -> torch.Tensor:
    """p R^T + t with a FIXED evaluation order of elementwise fp32 operations (one rounding each): bit-identical on every host.
    (`p @ R.T` goes through BLAS, whose association / FMA use depends on the CPU: measured, 160 000 points: pc1 differed in the last
    bit between the build container and the GPU box, which moves points across 0.1 m voxel edges and the deep-layer gradients by
    percents -- enough to break a digest comparison that is about 1e-4.)"""
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    t = [[float(T[i, j]) for j in range(4)] for i in range(3)]
    return torch.stack([(x * t[i][0] + y * t[i][1]) + (z * t[i][2] + t[i][3]) for i in range(3)], 1)


def _sqrt_f32(sq: torch.Tensor) -> torch.Tensor:
    """correctly rounded fp32 square root on every host: torch.sqrt on CPU is NOT (measured: the build container and the GPU box
    differ in the last bit for some of 160 000 values -- different vector code paths).  Candidate from the double root, then the
    choice among it and its two fp32 neighbours by comparing sq with the squared midpoints -- all exact in double (a midpoint has
    25 significant bits, its square 50)."""
    sq64 = sq.double()
    y = torch.sqrt(sq64).float()
    inf = torch.full_like(y, float("inf"))
    lo, hi = torch.nextafter(y, -inf), torch.nextafter(y, inf)
    m1, m2 = (lo.double() + y.double()) * 0.5, (y.double() + hi.double()) * 0.5
    return torch.where(sq64 < m1 * m1, lo, torch.where(sq64 > m2 * m2, hi, y))


EDGE_VOXEL, EDGE_TOL, EDGE_NUDGE = 0.1, 2e-4, 7e-4     # metres


def _near_edge(p64: torch.Tensor) -> torch.Tensor:
    """rows of a float64 cloud with x or y within EDGE_TOL of a pillar edge of the 0.1 m grid (hence of the 0.2 / 0.4 m grids, whose
    edges are a subset, and of the +-51.2 m range limits), or z within EDGE_TOL of the +-3 m range limits"""
    u = (p64[:, :2] + 51.2) / EDGE_VOXEL
    f = u - torch.floor(u)
    t = EDGE_TOL / EDGE_VOXEL
    return ((f < t) | (f > 1.0 - t)).any(1) | ((p64[:, 2].abs() - 3.0).abs() < EDGE_TOL)


def synth_pair(seed: int, n: int, grid_hw=(512, 512), nan_frac: float = 0.02, exact: bool = False
               ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> pc0 [n,3], pc1 [n,3], ego motion T (pc0 -> pc1 frame) [4,4], gt flow [n,3] (total, incl. ego motion).
    exact: host-independent arithmetic (no BLAS / LAPACK: see _rigid) -- what the committed oracle digests of round 4 are generated
    and checked with; the default keeps round 1-3's generator (their digests and goldens were made with it).
    exact also keeps every point that gets pillarised -- pc0 moved into pc1's frame, and pc1 -- at least EDGE_TOL = 0.2 mm away from
    every pillar edge and range limit.  The pillar index is a step function of the coordinate, and the model computes the moved pc0
    in fp32 (the reference through a BLAS matmul [REF deflow.py:103-108], this engine in ego_transform_kernel, the float64 oracle
    exactly): for a point within ~1e-5 m of an edge the three disagree about its pillar, which moved the deep encoder gradients of
    the configs[4] shape by 1-10 % BETWEEN TWO HOSTS RUNNING THE SAME fp32 ORACLE (tools/archive/cfg4_layer_probe.py: the first difference
    is in the pillar feature net's input of two of the eight clouds) -- a 1e-4 comparison is only well-posed on clouds without such
    points."""
    g = torch.Generator().manual_seed(seed)
    sigma = 20.0 * grid_hw[0] / 512.0
    xy = torch.randn(n, 2, generator=g) * sigma
    z = (torch.rand(n, 1, generator=g) * 6.6) - 3.3
    pc0 = torch.cat([xy, z], 1)
    yaw = (torch.rand(1, generator=g).item() * 4 - 2) * math.pi / 180
    T = torch.eye(4)
    T[0, 0] = math.cos(yaw); T[0, 1] = -math.sin(yaw); T[1, 0] = math.sin(yaw); T[1, 1] = math.cos(yaw)
    T[0, 3] = torch.rand(1, generator=g).item() * 1.5
    dyn = torch.rand(n, generator=g) < 0.1
    flow = torch.zeros(n, 3)
    d = torch.randn(n, 3, generator=g)
    if exact:
        nrm = _sqrt_f32((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]).unsqueeze(1)
    else:
        nrm = d.norm(dim=1, keepdim=True)
    d = d / nrm * (torch.rand(n, 1, generator=g) * 2.0)
    flow[dyn] = d[dyn]
    if exact:
        T64 = T.double()
        for _ in range(16):      # nudge the pc0 rows whose image under T sits on an edge (a nudge can land on the next edge: repeat)
            near = _near_edge(pc0.double() @ T64[:3, :3].T + T64[:3, 3])
            if not bool(near.any()):
                break
            pc0[near] = pc0[near] + EDGE_NUDGE
        assert not bool(_near_edge(pc0.double() @ T64[:3, :3].T + T64[:3, 3]).any())
    moved = _rigid(pc0, T) if exact else pc0 @ T[:3, :3].T + T[:3, 3]
    pc1 = moved + flow + torch.randn(n, 3, generator=g) * 0.02
    if exact:
        for _ in range(16):
            near = _near_edge(pc1.double())
            if not bool(near.any()):
                break
            pc1[near] = pc1[near] + EDGE_NUDGE
        assert not bool(_near_edge(pc1.double()).any())
    k = int(n * nan_frac)
    if k:
        pc0[-k:] = float("nan")
        pc1[-k:] = float("nan")
    gt_flow = (moved - pc0) + flow
    return pc0, pc1, T, gt_flow


def _rigid_inverse(T: torch.Tensor) -> torch.Tensor:
    """[R | t]^-1 = [R^T | -R^T t] evaluated in Python floats (host-independent; torch.linalg.inv goes through LAPACK)"""
    R = [[float(T[i, j]) for j in range(3)] for i in range(3)]
    t = [float(T[i, 3]) for i in range(3)]
    out = torch.eye(4)
    for i in range(3):
        for j in range(3):
            out[i, j] = R[j][i]
        out[i, 3] = -(R[0][i] * t[0] + R[1][i] * t[1] + R[2][i] * t[2])
    return out


def synth_batch(batch_size: int, n: int, seed: int = 20240116, grid_hw=(512, 512), device="cpu", exact: bool = False) -> Dict[str, torch.Tensor]:
    pairs = [synth_pair(seed + b, n, grid_hw, exact=exact) for b in range(batch_size)]
    return {
        "pc0": torch.stack([p[0] for p in pairs]).to(device),
        "pc1": torch.stack([p[1] for p in pairs]).to(device),
        "pose0": torch.eye(4).repeat(batch_size, 1, 1).to(device),
        "pose1": torch.stack([_rigid_inverse(p[2]) if exact else torch.linalg.inv(p[2]) for p in pairs]).to(device),
        "ego_motion": torch.stack([p[2] for p in pairs]).to(device),
        "flow": torch.stack([p[3] for p in pairs]).to(device),
    }
