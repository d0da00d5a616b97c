#!/usr/bin/env python
"""bench.py -- frame-pairs/s of the DeFlow training step (deflowLoss, Adam lr=2e-4, bs=16 per GPU, 512x512 BEV,
80 000-point clouds, 4 GRU iterations, fp32) on N MI355X of one node.  One process per GPU, RCCL over xGMI.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK / WORLD_SIZE in
the environment) or started bare -- then this file re-launches itself under torch.distributed.run (127.0.0.1 rendezvous,
free port) and rank 0's JSON line is the output.

A "step" = one full training pass over one synthetic, HBM-resident batch: ego-motion, pillarise both clouds, UNet
forward, GRU decoder, gt gather + deflowLoss, complete backward, gradient all-reduce (N > 1), Adam.  Nothing is
skipped or cached between steps.  Prints ONE JSON line on rank 0 (contract in the round prompt), including
  roofline        live HIP-event timing of the dominant kernel (fp32 MFMA implicit-GEMM conv) vs the 157.3 TFLOP/s peak
  roofline_hbm    the HBM-bound stages (pillarise, BatchNorm+GELU passes) and the GRU kernels: algorithmic bytes / time / 8 TB/s
  forward_only    BASELINE configs[1] (B=1 inference), timed AFTER the training region
  bf16_inference  BASELINE configs[4] shape (1024x1024, 160k points, 8 iterations, bf16 MFMA), after the training region
  cpu_baseline    the CPU oracle (oracle/ref_torch.py, a PyTorch port of the reference algorithm) timed on this host:
                  median of 5 after 2 warm-ups, training step and forward only
  (N > 1) rccl_ranks, per-rank step times, bf16_training over all ranks, allreduce_exposed_ms (steps with vs without the
          gradient collectives).
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "frame-pairs/sec training (deflow, bs=16, 512x512 BEV)"
WORKLOAD = "deflow train step (deflowLoss, Adam lr=2e-4): BASELINE configs[2] per GPU"      # the same strings at every N (SCALE vs BENCH)
PER_GPU_BATCH = 16
N_POINTS = 80000
GRID = 512
NUM_ITERS = 4
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16
PEAK_HBM_GBPS = 8000.0         # HBM3E spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=PER_GPU_BATCH, help="per-GPU batch (the metric is quoted at 16)")
    ap.add_argument("--rotate", type=int, default=4, help="distinct HBM-resident batches the timed steps rotate over")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-samples", type=int, default=5, help="timed CPU-oracle repetitions per leg (after 2 warm-ups)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip forward_only / bf16_inference / allreduce_exposed (e.g. for a rocprofv3 trace that should hold "
                         "the training launches only)")
    ap.add_argument("--forward-only", action="store_true", help="(kept for compatibility: the forward-only leg now always runs)")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-launch HIP events")
    ap.add_argument("--no-loader", action="store_true", help="skip the loader_fed extra (scene files written at run time; ~1 min)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / collective plumbing only: gloo, no GPU, no kernels (CPU test of the N > 1 start-up)")
    return ap.parse_args()


def _git(args):
    """best-effort git query (the GPU box's snapshot has no .git: None there)"""
    try:
        r = subprocess.run(["git", "-C", ROOT, *args], capture_output=True, text=True, timeout=10)
        return r.stdout.strip() or None if r.returncode == 0 else None
    except Exception:   # noqa: BLE001
        return None


def _build_info() -> dict:
    """deflow_amd/_build_info.json (written by deflow_amd.build at build time): the git state the library was built from -- the GPU
    box's snapshot has no .git"""
    try:
        with open(os.path.join(ROOT, "deflow_amd", "_build_info.json")) as f:
            return json.load(f)
    except Exception:   # noqa: BLE001
        return {}


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: one process per GPU under torch.distributed.run on this node."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (see the environment notes): RCCL needs it
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    return subprocess.call(cmd, env=env)


def sustained_mfma():
    """What the 16-bit matrix pipe of THIS chip sustains, measured now (tools/mfma_peak_probe.hip, built by __graft_entry__.build()):
    register-resident v_mfma_f32_32x32x16_f16 streams, nothing else running -- zero operands (no toggling: the nominal 2.5 PFLOP/s)
    and random fp16 operands (the chip clocks down to its power budget).  The second figure / 3 is the roof an fp16x2 kernel can
    reach on real data before it spends a joule on LDS, L2 or VALU."""
    exe = os.path.join(ROOT, "deflow_amd", "_build", "mfma_peak_probe")
    if not os.path.exists(exe):
        return None
    try:
        r = subprocess.run([exe, "30000", "2"], capture_output=True, text=True, timeout=120)
        z = [float(l.split()[-3]) for l in r.stdout.splitlines() if l.startswith("operands zero")]
        q = [float(l.split()[-3]) for l in r.stdout.splitlines() if l.startswith("operands random")]
        if not z or not q:
            return None
        return {"zero_operands_tflops": statistics.median(z), "random_operands_tflops": statistics.median(q)}
    except Exception:   # noqa: BLE001 -- a missing figure, never a failed bench
        return None


def strict_leg(env_extra, steps, warmup, batch):
    """the SAME training step in a fresh process with kernel forms switched by environment variables (the C side reads them once
    per process): `python bench.py --no-extras --no-cpu-baseline --no-profile` -> (ms_per_step, pairs/s, loss)"""
    env = dict(os.environ)
    env.update(env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--no-extras", "--no-cpu-baseline", "--no-profile", "--steps", str(steps),
                            "--warmup", str(warmup), "--batch", str(batch)], capture_output=True, text=True, env=env, timeout=900)
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        return {"ms_per_step": d["ms_per_step"], "pairs_per_s": d["value"], "loss": d["config"]["loss"], "env": env_extra}
    except Exception as e:   # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"[:300], "env": env_extra}


def cpu_baseline(samples: int):
    """The oracle on ONE frame pair of the same workload (bounded sample), all host threads: a full training step and a
    forward-only pass, each the median of `samples` repetitions after 2 warm-ups."""
    import torch
    from deflow_amd.synth import synth_batch
    from oracle import ref_torch as O
    torch.manual_seed(0)
    ref = O.DeFlow(grid_feature_size=[GRID, GRID], num_iters=NUM_ITERS).train()
    opt = torch.optim.Adam(ref.parameters(), lr=2e-4)
    batch = synth_batch(1, N_POINTS, seed=20240116)

    def train_step():
        opt.zero_grad()
        loss = O.training_loss(ref(batch), batch)
        loss.backward()
        opt.step()

    def forward():
        with torch.no_grad():
            ref(batch)

    def median_time(fn):
        for _ in range(2):
            fn()
        ts = []
        for _ in range(max(1, samples)):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts), min(ts), max(ts)

    tr = median_time(train_step)
    ref.eval()
    fw = median_time(forward)
    return {"value": 1.0 / tr[0], "unit": "frame-pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"training step on 1 synthetic frame pair ({N_POINTS} pts/cloud, {GRID}x{GRID}, {NUM_ITERS} GRU iters), "
                      f"oracle/ref_torch.py fp32: median of {samples} after 2 warm-ups = {tr[0]:.2f} s (min {tr[1]:.2f}, max {tr[2]:.2f}); "
                      f"host cpu_count={os.cpu_count()}",
            "forward_only": {"value": 1.0 / fw[0], "unit": "frame-pairs/s",
                             "sample": f"eval-mode forward of the same pair: median of {samples} after 2 warm-ups = {fw[0]:.2f} s "
                                       f"(min {fw[1]:.2f}, max {fw[2]:.2f})"}}


def bind_to_gpu_numa_node(local_rank: int):
    """First contact with a multi-GPU node: pin this rank's host threads to the NUMA node its GPU hangs off
    (eight ranks share the node's cores; the loader / Python of one rank should not run on the far socket).  Best effort: -> the
    node id, or None when sysfs does not say."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(local_rank)
        bdf = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
            node = int(f.read().strip())
        if node < 0:
            return None
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            cpus = set()
            for part in f.read().strip().split(","):
                a, _, b = part.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
        if cpus:
            os.sched_setaffinity(0, cpus)
        return node
    except Exception:   # noqa: BLE001
        return None


def dry_run(args, rank, world):
    """N > 1 start-up without a GPU: process group (gloo), barrier, MAX-reduced timing, all-gathered rank ids, one JSON line."""
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank)], dtype=torch.float64)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        g = torch.ones(1024)
        dist.all_reduce(g)
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    ids = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(ids, t)
    if rank == 0:
        print(json.dumps({"metric": METRIC, "dry_run": True, "value": None, "unit": "frame-pairs/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ranks": [int(x.item()) for x in ids],
                          "allreduce_sum_ok": bool(float(g[0]) == world), "ms_per_step": float(dt.item()) / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak",
                          "config": {"workload": WORKLOAD, "per_gpu_batch": args.batch, "global_batch": world * args.batch,
                                     "points_per_cloud": N_POINTS, "bev": [GRID, GRID], "gru_iters": NUM_ITERS, "parallelism": f"dp{world}"}}),
              flush=True)
    dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it bare (`python bench.py --gpus N`, it launches "
                 f"its own ranks) or with torch.distributed.run --nproc-per-node equal to --gpus")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if args.dry_run:
        return dry_run(args, rank, world)

    import torch
    assert torch.cuda.is_available(), "bench.py measures the HIP path; there is no CPU fallback"
    # test hook (tests/test_gpu_model.py): all ranks on cuda:0 with gloo carrying the collectives, so the N > 1 flow of this file
    # runs end to end with the real kernels on a one-GPU box (RCCL refuses two ranks on one device).  Never a measurement.
    share_gpu = os.environ.get("DF_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    numa = bind_to_gpu_numa_node(local_rank) if (world > 1 and not share_gpu) else None
    import torch.distributed as dist
    use_dist = world > 1 or "RANK" in os.environ  # under torch.distributed.run always go through RCCL (also at N=1)
    # an explicit collective timeout (the default is 10 min per call -- three stuck calls outlive the driver's patience; DF_BENCH_PG_TIMEOUT
    # seconds): a rank that never arrives then shows up as an exception in a guarded leg, not as a silent hang
    import datetime
    pg_timeout = datetime.timedelta(seconds=float(os.environ.get("DF_BENCH_PG_TIMEOUT", "300")))
    if use_dist and share_gpu:
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=pg_timeout)
    elif use_dist:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=pg_timeout)  # "nccl" == RCCL on ROCm

    import deflow_amd
    from deflow_amd import ops
    from deflow_amd.optim import Trainer
    from deflow_amd.synth import synth_batch

    torch.manual_seed(0)
    model = deflow_amd.DeFlow(grid_feature_size=[GRID, GRID], num_iters=NUM_ITERS).to(dev).train()
    trainer = Trainer(model, lr=2e-4)
    # weak scaling: every rank owns its own shard of frame pairs (seeded by global sample index), resident in HBM
    batch = synth_batch(args.batch, N_POINTS, seed=Trainer.shard_seed(20240116, rank, args.batch), device=dev)
    # round 5: the timed steps ROTATE over --rotate distinct resident batches (other scenes: other occupied pillars) -- the engine
    # keeps its BEV canvas across steps and rewrites only the cells whose occupancy changed (csrc/pillar_bands.hip), so stepping one
    # batch over and over would skip the re-zeroing a real epoch pays for; `pillar_canvas` in the line reports the measured fractions
    batches = [batch] + [synth_batch(args.batch, N_POINTS, seed=Trainer.shard_seed(20240116 + 100003 * j, rank, args.batch), device=dev)
                         for j in range(1, max(1, args.rotate))]
    step_no = [0]

    def next_batch():
        step_no[0] += 1
        return batches[step_no[0] % len(batches)]

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_steps(k):
        barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            loss_ = trainer.step(next_batch())
        torch.cuda.synchronize()
        local = time.perf_counter() - t0
        barrier()
        return time.perf_counter() - t0, local, loss_

    for _ in range(args.warmup):
        trainer.step(next_batch())
    selfcheck = None
    if use_dist:
        # RCCL self-check before the timed region: after the arena broadcast and the warm-up steps (each rank on its OWN shard, the
        # gradients all-reduced) the parameter arena must be BIT-IDENTICAL on every rank -- a wrong scaling curve is then diagnosable
        # from the one line.  Two 64-bit digests of the arena's bit pattern per rank, all-gathered.
        bits = trainer.flat.param.view(torch.int32).to(torch.int64)
        dgst = torch.stack([bits.sum(), (bits * torch.arange(1, bits.numel() + 1, device=dev, dtype=torch.int64) % 1000003).sum()])
        alld = [torch.zeros_like(dgst) for _ in range(world)]
        dist.all_gather(alld, dgst)
        selfcheck = {"params_bit_identical_across_ranks_after_warmup": bool(all(torch.equal(a, alld[0]) for a in alld)),
                     "steps_checked": args.warmup, "numa_node": numa}
    prof = None if args.no_profile else ops.KernelProfiler()
    ops.PROFILER = prof
    dt, dt_local, loss = timed_steps(args.steps)       # THE timed region: exactly --steps steps between barriers
    ops.PROFILER = None
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    per_rank = None
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        mine = torch.tensor([dt_local / args.steps * 1e3, float(torch.cuda.current_device())], dtype=torch.float64, device=dev)
        per_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(per_rank, mine)
    dt = float(t.item())

    # ---- everything below the timed region is an EXTRA.  At N > 1 each extra leg runs under a wall-clock budget (DF_BENCH_LEG_BUDGET
    # seconds per leg, default 240): a leg that hangs -- a collective one rank never joins, a capture that deadlocks on first contact with
    # eight real devices -- must not cost the HEADLINE already measured above.  When a leg overruns, every rank's watchdog thread ends
    # its process; rank 0 first prints the headline line with `extras_aborted` naming the leg.
    headline = {"metric": METRIC, "value": world * args.batch * args.steps / dt, "unit": "frame-pairs/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": WORKLOAD, "per_gpu_batch": args.batch, "global_batch": world * args.batch,
                           "points_per_cloud": N_POINTS, "bev": [GRID, GRID], "gru_iters": NUM_ITERS, "parallelism": f"dp{world}"}}
    leg_state = {"name": None, "deadline": None, "done": []}

    def _watchdog():
        while True:
            time.sleep(0.5)
            dl = leg_state["deadline"]
            if dl is not None and time.monotonic() > dl:
                if rank == 0:
                    line = dict(headline)
                    line["extras_aborted"] = {"leg": leg_state["name"], "budget_s": leg_budget, "legs_completed": list(leg_state["done"]),
                                              "note": "an extra leg behind the timed region overran its wall-clock budget; the headline above it was "
                                                      "measured before any extra ran"}
                    print(json.dumps(line), flush=True)
                os._exit(0)

    leg_budget = float(os.environ.get("DF_BENCH_LEG_BUDGET", "240"))
    if use_dist and world > 1 and not args.no_extras:
        import threading
        threading.Thread(target=_watchdog, daemon=True).start()

    class leg:      # `with leg("name"):` arms the watchdog for one extra leg
        def __init__(self, name):
            self.name = name

        def __enter__(self):
            leg_state["name"], leg_state["deadline"] = self.name, time.monotonic() + leg_budget
            if os.environ.get("DF_BENCH_TEST_HANG") == self.name:      # test hook: this leg never returns (tests/test_gpu_model.py)
                time.sleep(10 ** 6)

        def __exit__(self, *exc):
            leg_state["deadline"] = None
            leg_state["done"].append(self.name)
            return False

    exposed = None
    bf16_multi = None
    graph_multi = None
    buckets = None
    one_bucket = None
    one_bucket_default = bool(getattr(getattr(trainer, "sink", None), "one_bucket", False))   # (read before any A/B leg flips it)
    if use_dist and trainer.collective and not args.no_extras:
        # the gradient collectives of one eager step, bucket by bucket (after the timed region): arena range, bytes, issue ->
        # complete on the device clock.  A first 8-GPU run that scales badly can then be read from this one line: which bucket is
        # late, how long each is in flight, how much of the last one sticks out behind the backward
        try:
            with leg("allreduce_buckets"):
                trainer.sink.trace_on(True)
                for _ in range(2):
                    trainer.step(batch)
                torch.cuda.synchronize()
                buckets = trainer.sink.trace_report()
        except Exception as e:      # noqa: BLE001
            buckets = {"error": f"{type(e).__name__}: {e}"[:300]}
        trainer.sink.trace_on(False)
        # fallback leg: no bucketing, ONE all-reduce of the whole arena after the backward (DF_ONE_BUCKET=1 makes it the default)
        prev = trainer.sink.one_bucket          # (= the default every other leg runs with: read BEFORE the A/B leg flips it)
        one_bucket_default = bool(prev)
        try:
            with leg("one_bucket"):
                trainer.sink.one_bucket = True
                trainer.step(batch)
                dt1, _, _ = timed_steps(args.steps)
                t1_ = torch.tensor([dt1], dtype=torch.float64, device=dev)
                dist.all_reduce(t1_, op=dist.ReduceOp.MAX)
            one_bucket = {"ms_per_step": float(t1_.item()) / args.steps * 1e3, "bytes": trainer.flat.numel * 4,
                          "note": "the same step with DF_ONE_BUCKET=1: one all-reduce of the whole gradient arena after the backward, nothing overlapped"}
        except Exception as e:      # noqa: BLE001
            one_bucket = {"error": f"{type(e).__name__}: {e}"[:300]}
        finally:
            trainer.sink.one_bucket = prev      # whatever happened: the later legs run the bucketed configuration they report
    if world > 1 and not args.no_extras:
        # BASELINE configs[4] ("bf16 MFMA, 8xMI355X"): the same data-parallel step with dtype=bf16, collectives on, timed the same way
        try:
            with leg("bf16_training"):
                trainer.mfma_bf16 = True
                for _ in range(2):
                    trainer.step(batch)
                dtb_, _, lossb_ = timed_steps(args.steps)
                trainer.mfma_bf16 = False
                tb_ = torch.tensor([dtb_], dtype=torch.float64, device=dev)
                dist.all_reduce(tb_, op=dist.ReduceOp.MAX)
                bf16_multi = (float(tb_.item()), float(lossb_))
        except Exception as e:      # noqa: BLE001 -- reported, never fatal for the headline
            bf16_multi = None
            headline.setdefault("extras_failed", {})["bf16_training"] = f"{type(e).__name__}: {e}"[:300]
        finally:
            trainer.mfma_bf16 = False
        # how much of the gradient all-reduce is NOT hidden behind the backward: the same steps without the collectives
        # (replicas drift apart afterwards -- only timings are taken after this)
        try:
            with leg("allreduce_exposed"):
                trainer.collective = trainer.sink.collective = False
                dt_nc, _, _ = timed_steps(args.steps)
                tn = torch.tensor([dt_nc], dtype=torch.float64, device=dev)
                dist.all_reduce(tn, op=dist.ReduceOp.MAX)
                exposed = (dt - float(tn.item())) / args.steps * 1e3
        except Exception as e:      # noqa: BLE001
            headline.setdefault("extras_failed", {})["allreduce_exposed"] = f"{type(e).__name__}: {e}"[:300]
        finally:
            trainer.collective = trainer.sink.collective = True
        # the data-parallel bf16 step as a captured PROGRAM (HIP-graph segments split at the gradient buckets, the collectives
        # between them: optim.SegmentedCapture) -- the host-free form of the step the 8-GPU deployment runs.  Last, and guarded:
        # a failure here must not take the headline with it
        try:
          with leg("hip_graph"):
              trainer.mfma_bf16 = True
              trainer.capture(batch)
              for _ in range(2):
                  trainer.step_captured()
              barrier()
              tg0 = time.perf_counter()
              for _ in range(args.steps):
                  trainer.step_captured()
              torch.cuda.synchronize()
              tloc = time.perf_counter() - tg0
              barrier()
              tg = torch.tensor([time.perf_counter() - tg0], dtype=torch.float64, device=dev)
              dist.all_reduce(tg, op=dist.ReduceOp.MAX)
              hosts = []
              for _ in range(5):          # host cost of ONE replayed step with an empty queue (no back-pressure from the GPU)
                  barrier()
                  th = time.perf_counter()
                  if share_gpu:           # gloo's all_reduce blocks the host on the device: time the graph launches alone
                      for o in trainer._program:
                          if o[0] == "graph":
                              o[1].replay()
                  else:
                      trainer.step_captured()
                  hosts.append((time.perf_counter() - th) * 1e3)
              torch.cuda.synchronize()
              kinds = [o[0] for o in trainer._program]
              graph_multi = {"workload": "the data-parallel bf16 step replayed as HIP-graph segments split at the gradient buckets, collectives between them",
                             "ms_per_step": float(tg.item()) / args.steps * 1e3, "host_ms_per_step": statistics.median(hosts),
                             "graph_segments": kinds.count("graph"), "allreduce_calls": sum(len(o[1]) for o in trainer._program if o[0] == "allreduce")}
        except Exception as e:      # noqa: BLE001 -- reported in the line, never fatal
            graph_multi = {"error": f"{type(e).__name__}: {e}"[:300]}
        trainer.mfma_bf16 = False
    if rank != 0:
        dist.destroy_process_group()
        return

    ms = dt / args.steps * 1e3
    out = {
        "metric": METRIC,
        "value": world * args.batch * args.steps / dt, "unit": "frame-pairs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        # every tensor, accumulator, statistic and the optimizer are fp32; what the GEMMs multiply with (round 3): the 3x3 stride-1
        # convs and weight gradients as two scaled fp16 planes per operand (22 significant bits, three MFMAs: <= 2e-6 vs float64),
        # the GRU decoder's GEMMs as two bf16 planes (16 bits, three MFMAs: <= 2.9e-5 vs float64), everything else fp32 MFMA.
        # DF_CONV_X3=0 DF_WGRAD_X3=0 DF_GRU_X2=0 put every GEMM back on the fp32 MFMA (107 pairs/s, round 2's step)
        "dtype_note": ("f32 tensors and accumulation; GEMM operands: 3x3 convs fp16x2 (2 scaled fp16 planes, 3 MFMAs; between 3x3 layers the "
                       "activations / gradients are STORED as those planes, 4 bytes per element), GRU decoder bf16x2 "
                       "(2 bf16 planes, 3 MFMAs), other layers fp32 MFMA or fp16x2 on fragments; strict_fp32 / gru_fp32 = the same step with those forms off" if (os.environ.get("DF_CONV_X3", "1") != "0" or os.environ.get("DF_GRU_X2", "1") != "0")
                       else "f32 tensors, accumulation and MFMA operands"),
        "config": {"workload": WORKLOAD, "per_gpu_batch": args.batch,
                   "global_batch": world * args.batch, "points_per_cloud": N_POINTS, "bev": [GRID, GRID], "gru_iters": NUM_ITERS,
                   "parallelism": f"dp{world}", "loss": float(loss)},
    }
    out["config"]["distinct_batches_rotated"] = len(batches)
    try:
        # what the persistent BEV canvas saved / paid between two consecutive (different) batches: cells occupied now, cells that had
        # to be re-zeroed because they were occupied by the previous batch only -- read off the band kernels' occupancy words
        from deflow_amd import deflow as _D
        ents = [v for k, v in (_D._CANVASES.get(model) or {}).items() if k[0] == args.batch and not k[4]]
        if ents and len(batches) > 1:
            occs = [o for o in ents[0][1] if o is not None]
            sh = torch.arange(32, device=dev, dtype=torch.int32)
            nbits = lambda w: int(((w.reshape(-1, 1) >> sh) & 1).sum().item())
            trainer.step(batches[0])
            prev = [o.clone() for o in occs]
            trainer.step(batches[1])
            cells = float(2 * args.batch * GRID * GRID)
            out["pillar_canvas"] = {"persistent": True, "occupied_frac": sum(nbits(o) for o in occs) / cells,
                                    "rezeroed_frac": sum(nbits(p_ & ~o) for p_, o in zip(prev, occs)) / cells,
                                    "note": "fractions of the 2 x B x 512 x 512 canvas cells written per step (pillar rows / zero rows); a dense "
                                            "canvas (DF_CANVAS_PERSIST=0) writes all of them"}
        else:
            out["pillar_canvas"] = {"persistent": False}
    except Exception as e:      # noqa: BLE001
        out["pillar_canvas"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    if "extras_failed" in headline:
        out["extras_failed"] = headline["extras_failed"]
    if leg_state["done"]:
        out["extras_completed"] = list(leg_state["done"])
    if selfcheck is not None:
        out["rccl_selfcheck"] = selfcheck
    if per_rank is not None:
        out["rccl_ranks"] = [int(x[1].item()) for x in per_rank]
        out["per_rank_ms_per_step"] = {"min": min(float(x[0]) for x in per_rank), "max": max(float(x[0]) for x in per_rank)}
    if exposed is not None:
        out["allreduce_exposed_ms"] = exposed
    if buckets is not None:
        out["allreduce_buckets"] = buckets
    if one_bucket is not None:
        out["one_bucket"] = one_bucket
    if use_dist:
        out["collectives"] = {"backend": "gloo (test hook)" if share_gpu else "nccl (RCCL)", "one_bucket_default": one_bucket_default,
                              "arena_bytes": trainer.flat.numel * 4, "ranks": world}
    if share_gpu:
        out["collective_backend"] = "gloo, all ranks on cuda:0 (DF_BENCH_SHARE_GPU test hook: not a measurement)"
    if bf16_multi is not None:
        out["bf16_training"] = {"workload": "the headline data-parallel step with dtype=bf16 (bf16 MFMA operands, fp32 accumulate / tensors / "
                                            "master weights), gradient collectives on, MAX over ranks",
                                "ms_per_step": bf16_multi[0] / args.steps * 1e3, "pairs_per_s": world * args.batch * args.steps / bf16_multi[0],
                                "speedup_vs_fp32": dt / bf16_multi[0], "loss": bf16_multi[1]}
    if graph_multi is not None:
        out["hip_graph"] = graph_multi
    if prof is not None and os.environ.get("DF_BENCH_DUMP"):
        with open(os.environ["DF_BENCH_DUMP"], "w") as f:
            per = len(prof.records) // args.steps
            for name, flops, e0, e1, tag, *rest in prof.records[:per]:
                ms_ = e0.elapsed_time(e1)
                f.write(f"{name:28s} {flops / 1e9:10.2f} GF {(rest[0] if rest else 0) / 1e6:10.2f} MB {ms_:8.3f} ms "
                        f"{flops / ms_ / 1e9:8.1f} TF/s  {tag}\n")
    if prof is not None:
        summ = prof.summary()
        # dominant kernel = the MFMA kernel (convolutions AND weight gradients; the GRU kernels are priced in roofline_hbm) with the
        # largest share of the timed region
        conv = {k: v for k, v in summ.items() if v["flops"] > 0 and not k.startswith("gru_") and (k.startswith("conv_") or k.startswith("wgrad"))}
        dom = max(conv, key=lambda k: conv[k]["ms"])
        d = conv[dom]
        achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
        traffic = None  # HBM bytes per launch from the committed PMC passes (profiles/pmc_traffic.json), same workload
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                kk = json.load(f)["kernels"]
                # rocprof prints the full template list (the trailing `false` = fp32 operands)
                # (later rounds added template arguments: "wgrad3_h2p_kernel<4>" is "wgrad3_h2p_kernel<4, true>" in the trace)
                norm = {k_.replace(" ", ""): v for k_, v in kk.items()}
                stem = dom.replace(" ", "")[:-1]
                ent = norm.get(stem + ">") or norm.get(stem + ",false>") or next((v for k_, v in norm.items() if k_.startswith(stem + ",")), None)
                traffic = ent["hbm_bytes_per_launch"] if (ent and args.batch == PER_GPU_BATCH) else None
        except Exception:
            traffic = None
        mfma = {k: v for k, v in summ.items() if v["flops"] > 0 and not k.startswith("gru_")}
        # conv_halo_x3_kernel / wgrad3_x3_kernel compute the fp32 product on the 16-bit matrix pipe: the fp16x2
        # forms (template argument NP = 2, the default since round 3) as THREE fp16 MFMAs per algorithmic multiply -- roof = dense
        # fp16 MFMA peak (= the bf16 one, 2500) / 3 = 833.3 algorithmic TFLOP/s; the bf16x3 forms (DF_CONV_H2=0) as SIX bf16 MFMAs
        # -- roof 416.7.  Neither is the fp32-MFMA peak (157.3) the round-1/2 kernels were priced against, which they exceed.
        x3 = "_x3_" in dom or "wgrad3_h2p" in dom
        nmfma = (3 if (dom.rstrip(">").endswith(",2") or dom.endswith(",xp>") or "h2p" in dom or dom == "wgrad3_x3_kernel<2>") else 6) if x3 else 1
        peak = PEAK_BF16_MFMA_TFLOPS / nmfma if x3 else PEAK_F32_MFMA_TFLOPS
        out["roofline"] = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                           "frac": achieved / peak, "traffic": traffic, "kernel": dom,
                           "peak_note": ("fp32-accurate product = %d exact 16-bit MFMAs per multiply (%s): peak = 2500 dense fp16/bf16 TFLOP/s / %d; "
                                         "achieved = %.2f x the fp32-MFMA peak (157.3) the fp32 kernels of rounds 1-2 were bound by"
                                         % (nmfma, "fp16x2: two scaled fp16 planes per operand" if nmfma == 3 else "bf16x3: three bf16 planes per operand",
                                            nmfma, achieved / PEAK_F32_MFMA_TFLOPS)) if x3
                           else "dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32)",
                           "executed_16bit_tflops": achieved * nmfma if x3 else None,
                           "launches_per_step": d["launches"] / args.steps, "avg_launch_ms": d["ms"] / d["launches"],
                           "algorithmic_gflop_per_launch": d["flops"] / d["launches"] / 1e9,
                           "all_mfma_kernels": {k: {"launches_per_step": v["launches"] / args.steps,
                                                    "ms_per_step": v["ms"] / args.steps,
                                                    "tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12,
                                                    # input + output (+ weight) bytes once / time: a kernel under 150 TFLOP/s at >= 3.8 TB/s
                                                    # is a small-K layer waiting for the memory system, not for the matrix pipe
                                                    "operand_tbps": v["bytes"] / (v["ms"] * 1e-3) / 1e12} for k, v in mfma.items()}}
        # the whole fp16x2 family (every 3x3 stride-1 convolution, data gradient and weight gradient of the step): flop-weighted
        fam = [v for k, v in mfma.items() if ("_x3_" in k and (k.rstrip(">").endswith(",2") or k.endswith(",xp>"))) or "wgrad3_h2p" in k or k == "wgrad3_x3_kernel<2>"]
        if fam:
            ft = sum(v["flops"] for v in fam) / (sum(v["ms"] for v in fam) * 1e-3) / 1e12
            out["roofline"]["fp16x2_family"] = {"ms_per_step": sum(v["ms"] for v in fam) / args.steps, "tflops": ft,
                                                "frac": ft / (PEAK_BF16_MFMA_TFLOPS / 3.0),
                                                "share_of_step": sum(v["ms"] for v in fam) / args.steps / ms}
        sus = None if os.environ.get("DF_BENCH_NO_SUBPROC") == "1" else sustained_mfma()
        if sus is not None:
            # the chip sustains the nominal 16-bit peak only on operands that toggle nothing; under random operands a PURE MFMA stream
            # (no LDS, no memory) runs at the power-limited clock: that figure / 3 is the reachable roof of the fp16x2 kernels
            clk = 2.4 * sus["random_operands_tflops"] / PEAK_BF16_MFMA_TFLOPS
            out["roofline"]["sustained"] = {
                "mfma_16bit_tflops_zero_operands": sus["zero_operands_tflops"], "mfma_16bit_tflops_random_operands": sus["random_operands_tflops"],
                "sustained_clock_ghz": clk, "peak_at_sustained_clock": sus["random_operands_tflops"] / nmfma if x3 else None,
                "frac_at_sustained_clock": (achieved / (sus["random_operands_tflops"] / nmfma)) if x3 else None,
                "note": "register-resident v_mfma_f32_32x32x16_f16 stream measured in this run (tools/mfma_peak_probe.hip): the matrix pipe alone, "
                        "clock = 2.4 GHz x random / 2500"}
        try:   # worst three-way errors at configs[2] from the newest COMMITTED parity report (a GPU test run's record, NOT measured
            # by this bench run: `source` / `source_commit` say which file and which code state it describes)
            worst = {}
            reports = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_parity_report.jsonl"))
            for rname in reversed(reports):       # the newest report that holds the configs[2] digest rows (a partial run's file does not)
                report = os.path.join(ROOT, "profiles", rname)
                with open(report) as f:
                    for line in f:
                        e = json.loads(line)
                        if e.get("test") in ("bs16_512", "bs16_512_gru_fp32") and "max_proj_err_over_l2" in e:
                            w = worst.setdefault(e["test"], {"worst_rms_rel": 0.0, "tensor": "", "bound": e.get("rms_bound")})
                            if e["max_proj_err_over_l2"] > w["worst_rms_rel"]:
                                w["worst_rms_rel"], w["tensor"] = e["max_proj_err_over_l2"], e.get("tensor", "")
                if worst:
                    break
            if worst:
                out["model_error_budget"] = {"workload": "configs[2] (B=16, 512x512, 80k pts) training step vs the float64 oracle digests: worst "
                                                         "|projection error| / ||g|| over every parameter gradient (a few sigma of the rms-relative error); "
                                                         "bound 1e-4 x 4.5 sigma", **worst,
                                             "source": os.path.relpath(report, ROOT), "source_mtime": int(os.path.getmtime(report)),
                                             "source_commit": _git(["log", "-1", "--format=%h", "--", os.path.relpath(report, ROOT)])
                                             or _build_info().get("profiles", {}).get(os.path.relpath(report, ROOT)),
                                             "head_commit": _git(["rev-parse", "--short", "HEAD"]) or _build_info().get("head_commit"),
                                             "built_from_dirty_tree": _build_info().get("dirty"),
                                             "measured_in_this_run": False}
        except Exception:   # noqa: BLE001
            pass
        # HBM-bound stages and the GRU kernels: algorithmic bytes / measured time / 8 TB/s
        hbm = {}

        def entry(name, keys, note):
            sel = [summ[k] for k in keys if k in summ]
            if not sel:
                return
            b = sum(v["bytes"] for v in sel)
            msx = sum(v["ms"] for v in sel)
            fl = sum(v["flops"] for v in sel)
            e = {"ms_per_step": msx / args.steps, "algorithmic_mb_per_step": b / args.steps / 1e6,
                 "achieved_gbps": b / (msx * 1e-3) / 1e9, "peak_gbps": PEAK_HBM_GBPS, "frac": b / (msx * 1e-3) / 1e9 / PEAK_HBM_GBPS,
                 "note": note}
            mv = sum(v.get("moved", 0.0) for v in sel)
            if mv:   # what the implementation streams through HBM by construction (saved / gradient planes), beside the algorithmic minimum
                e["moved_mb_per_step"] = mv / args.steps / 1e6
                e["moved_gbps"] = mv / (msx * 1e-3) / 1e9
                e["moved_over_algorithmic"] = mv / b if b else None
            if fl:
                e["tflops"] = fl / (msx * 1e-3) / 1e12
                # the decoder GEMMs run as bf16x2 (three 16-bit MFMAs per multiply: roof = 2500 / 3) unless DF_GRU_X2=0 (fp32 MFMA: 157.3)
                x2 = os.environ.get("DF_GRU_X2", "1") != "0"
                e["mfma_peak_tflops"] = PEAK_BF16_MFMA_TFLOPS / 3.0 if x2 else PEAK_F32_MFMA_TFLOPS
                e["frac_mfma"] = e["tflops"] / e["mfma_peak_tflops"]
                e["x_fp32_mfma_peak"] = e["tflops"] / PEAK_F32_MFMA_TFLOPS
            hbm[name] = e

        entry("pillarise_fwd", ["pillarise_fwd", "canvas_zero_fill"],
              "both clouds: voxelise, sort, feature net (train: + batch statistics), canvas incl. its zero fill; "
              "bytes = points in + dense 32-channel canvas out (SURVEY 8d: 69.0 MB/pair)")
        entry("bn_gelu_apply", ["bn_gelu_apply"], "BatchNorm normalise + GELU: read y, write z (8 B/element)")
        entry("bn_gelu_bwd", ["bn_gelu_bwd_reduce", "bn_gelu_bwd_apply"], "two passes: read dz,y | read dz,y write dy (20 B/element)")
        # the decoder kernels were priced per padded row; only the valid rows (not NaN padding / out of range) do work
        vf = float(model.last_state["counts0"].sum().item()) / float(args.batch * N_POINTS)
        for k in ("gru_fwd", "gru_bwd", "gru_wgrad"):
            if k in summ:
                summ[k]["flops"] *= vf
                summ[k]["moved"] = summ[k].get("moved", 0.0) * vf
                if k != "gru_wgrad":        # (its algorithmic bytes are the weights: fused into the data pass it would re-read nothing)
                    summ[k]["bytes"] *= vf
        entry("gru_fwd", ["gru_fwd"], "gather + T GRU steps + head, training form; bytes = fused minimum (548 B/point)")
        entry("gru_bwd", ["gru_bwd"], "data gradients (flops = the forward's un-hoisted count: the same GEMMs against transposed weights); "
                                      "bytes = the FUSED minimum (re-gather 512 B + dh0 512 B + dflow/offsets per point); moved = the hidden-state "
                                      "planes read + the gate-gradient / r*h planes written for the weight-gradient pass")
        entry("gru_wgrad", ["gru_wgrad"], "gate weight gradients; bytes = the FUSED minimum (the weights: fused into the data pass it would re-read "
                                          "no activations); moved = the planes it streams (3 gate-gradient planes + h_in + r*h per step)")
        entry("gru_trio", ["gru_fwd", "gru_bwd", "gru_wgrad"], "the three decoder kernels together against SURVEY 8(d)'s fused minimum")
        out["roofline_hbm"] = hbm

    if not args.no_extras:
        # BASELINE configs[4] names "bf16 MFMA": the same training step with the UNet convolutions (forward, data gradient,
        # weight gradient) on bf16 MFMA operands, fp32 accumulation, fp32 tensors / master weights (Trainer(dtype="bf16")).
        # Timed AFTER and reported BESIDE the fp32 headline, never instead of it.  (N > 1: measured above, before the collectives
        # are switched off for allreduce_exposed_ms.)
        if world == 1:
            trainer.mfma_bf16 = True
            for _ in range(2):
                trainer.step(batch)
            dtb, _, lossb = timed_steps(args.steps)
            # the same bf16 step replayed as ONE captured HIP graph (Trainer.capture): host-free steps
            trainer.capture(batch)
            for _ in range(2):
                trainer.step_captured()
            barrier()
            tg0 = time.perf_counter()
            for _ in range(args.steps):
                trainer.step_captured()
            torch.cuda.synchronize()
            graph_ms = (time.perf_counter() - tg0) / args.steps * 1e3
            hosts = []
            for _ in range(5):      # host cost of ONE replay with an empty queue: back-to-back replays block on the GPU's queue
                torch.cuda.synchronize()   # depth (round 2's figure grew with --steps), which is GPU time, not host work
                th = time.perf_counter()
                trainer.step_captured()
                hosts.append((time.perf_counter() - th) * 1e3)
            torch.cuda.synchronize()
            host_ms = statistics.median(hosts)
            trainer.mfma_bf16 = False
            out["hip_graph"] = {"workload": "the bf16 step above as one captured HIP graph (Trainer.capture / step_captured)",
                                "ms_per_step": graph_ms, "host_ms_per_step": host_ms, "eager_ms_per_step": dtb / args.steps * 1e3}
            out["bf16_training"] = {"workload": "the headline step with dtype=bf16 (bf16 MFMA operands in every UNet conv fwd/dgrad/wgrad and every "
                                                "decoder GEMM, fp32 accumulate; fp32 activations, gradients, master weights and GRU state; 3x3 conv tiles and the GRU's "
                                                "saved planes in bf16)",
                                    "ms_per_step": dtb / args.steps * 1e3, "pairs_per_s": args.batch * args.steps / dtb,
                                    "speedup_vs_fp32": dt / dtb, "loss": float(lossb)}
    if world == 1 and not args.no_extras:
        # forward-only pairs/s of BASELINE configs[1] (one 80k-point pair, eval mode) beside the headline
        def time_forward(m, b, reps):
            with torch.no_grad():
                for _ in range(3):
                    m.forward_padded(b)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(reps):
                    m.forward_padded(b)
                torch.cuda.synchronize()
            return (time.perf_counter() - t1) / reps * 1e3

        model.eval()
        b1 = synth_batch(1, N_POINTS, seed=20240116, device=dev)
        fwd_ms = time_forward(model, b1, 20)
        out["forward_only"] = {"workload": "BASELINE configs[1]: deflow inference, 1 pair (B=1), 80000 pts/cloud, 512x512, 4 GRU iters, fp32",
                               "ms_per_pair": fwd_ms, "pairs_per_s": 1e3 / fwd_ms,
                               "algorithmic_tflops": 391.6e9 / (fwd_ms * 1e-3) / 1e12,
                               # the 3x3 stride-1 convs (343 of the 391.6 GFLOP) run on the 16-bit matrix pipe (fp16x2: peak / 3 =
                               # 833 algorithmic TFLOP/s), the GRU decoder and the 1x1 / stride-2 convs on the fp32 MFMA (157.3): the
                               # figure below is against the FORMER only as an upper bound of the roof; rounds 1-2 quoted x / 157.3
                               "frac_mfma_16bit_over_3": 391.6e9 / (fwd_ms * 1e-3) / 1e12 / (PEAK_BF16_MFMA_TFLOPS / 3.0),
                               "x_fp32_mfma_peak": 391.6e9 / (fwd_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS}
        fwd16 = time_forward(model, batch, 10)
        out["forward_only"]["b16_ms_per_pair"] = fwd16 / args.batch
        # the pillarise stage of that B = 16 inference forward (both clouds as one 32-sample set) against the HBM roof
        ops.PROFILER = pp = ops.KernelProfiler()
        time_forward(model, batch, 5)
        ops.PROFILER = None
        ps_ = pp.summary().get("pillarise_fwd")
        if ps_ and "roofline_hbm" in out:
            ms_pair = ps_["ms"] / ps_["launches"] / args.batch
            out["roofline_hbm"]["pillarise_fwd_inference_b16"] = {
                "us_per_pair": ms_pair * 1e3, "algorithmic_mb_per_pair": ps_["bytes"] / ps_["launches"] / args.batch / 1e6,
                "achieved_gbps": ps_["bytes"] / (ps_["ms"] * 1e-3) / 1e9, "peak_gbps": PEAK_HBM_GBPS,
                "frac": ps_["bytes"] / (ps_["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBPS,
                "note": "eval forward, B=16: hist + scan + scatter + band (4 launches for both clouds), canvas zeros included"}
        model.inference_dtype = "bf16"
        f16 = time_forward(model, b1, 20)
        out["forward_only"]["bf16_ms_per_pair"] = f16
        model.train()
        model.inference_dtype = "fp32"
        del b1
        # BASELINE configs[4] shape: 1024x1024 grid (voxel 0.1 m), 160k points per cloud, 8 GRU iterations, bf16 MFMA; one pair
        torch.manual_seed(0)
        big = deflow_amd.DeFlow(voxel_size=[0.1, 0.1, 6], grid_feature_size=[1024, 1024], num_iters=8).to(dev).eval()
        bb = synth_batch(1, 160000, seed=20240116, device=dev)
        f32_ms = time_forward(big, bb, 5)
        big.inference_dtype = "bf16"
        bf_ms = time_forward(big, bb, 10)
        # roofs from SURVEY 8(d): 1564 GFLOP and 3.85 GB per pair at this shape
        out["bf16_inference"] = {"workload": "BASELINE configs[4] shape: deflow inference, 1 pair, 160000 pts/cloud, 1024x1024, 8 GRU iters",
                                 "dtype": "bf16 MFMA (fp32 accumulate, fp32 pillars / gates)", "ms_per_pair": bf_ms, "pairs_per_s": 1e3 / bf_ms,
                                 "fp32_ms_per_pair": f32_ms, "algorithmic_tflops": 1564e9 / (bf_ms * 1e-3) / 1e12,
                                 "frac_mfma_bf16": 1564e9 / (bf_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS,
                                 "frac_hbm": 3.85e9 / (bf_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS}
        del big, bb
        # ... and the TRAINING step at that shape (configs[4]: "1024x1024 BEV grid, 160k-pt pair, 8 GRU iters, bf16 MFMA"): 4 pairs
        # per GPU (the activation footprint of 16 pairs at 512x512), fp32 and bf16 MFMA, same Trainer
        del trainer, model
        torch.cuda.empty_cache()
        torch.manual_seed(0)
        big = deflow_amd.DeFlow(voxel_size=[0.1, 0.1, 6], grid_feature_size=[1024, 1024], num_iters=8).to(dev).train()
        tb = Trainer(big, lr=2e-4)
        bb = synth_batch(4, 160000, seed=20240116, device=dev)
        res4 = {}
        for name, flag in (("fp32", False), ("bf16", True)):
            tb.mfma_bf16 = flag
            for _ in range(2):
                tb.step(bb)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                l4 = tb.step(bb)
            torch.cuda.synchronize()
            res4[name] = (time.perf_counter() - t1) / 3 * 1e3
        out["bf16_training"]["configs4_shape"] = {
            "workload": "train step, 4 pairs per GPU, 160000 pts/cloud, 1024x1024, 8 GRU iters (BASELINE configs[4] per GPU)",
            "fp32_ms_per_step": res4["fp32"], "bf16_ms_per_step": res4["bf16"], "bf16_pairs_per_s": 4e3 / res4["bf16"],
            "speedup_vs_fp32": res4["fp32"] / res4["bf16"], "loss": float(l4)}
        del big, bb, tb
        torch.cuda.empty_cache()
        # the precision ladder beside the headline: the same step with EVERY GEMM on the fp32 MFMA, and with only the
        # GRU decoder back on it -- fresh processes, because the library reads its switches once
        # N2 (the step before the path): the same training step fed by the scene-file loader (h5 scenes of AV2-sized sweeps written at run
        # time, 4 reader processes) against one resident batch -- is the loader able to feed this step?  (tools/bench_loader.py)
        nosub = os.environ.get("DF_BENCH_NO_SUBPROC") == "1"     # (under rocprofv3: child processes would be traced into the same output)
        for dt_ in (() if (args.no_loader or nosub) else ("fp32", "bf16")):
            try:
                env = dict(os.environ, DF_LOADER_QUICK="1", DF_LOADER_DTYPE=dt_)
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_loader.py")], capture_output=True, text=True, env=env, timeout=600)
                out.setdefault("loader_fed", {})[dt_] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
            except Exception as e:   # noqa: BLE001
                out.setdefault("loader_fed", {})[dt_] = {"error": f"{type(e).__name__}: {e}"[:200]}
        if not nosub:
          out["strict_fp32"] = strict_leg({"DF_CONV_X3": "0", "DF_WGRAD_X3": "0", "DF_CONV_H2F": "0", "DF_GRU_X2": "0"}, args.steps, args.warmup, args.batch)
          out["gru_fp32"] = strict_leg({"DF_GRU_X2": "0"}, args.steps, args.warmup, args.batch)
    if use_dist:
        dist.destroy_process_group()
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_samples)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
