#!/usr/bin/env python
"""bench.py -- frame-pairs/s of the DeFlow training step (deflowLoss, Adam lr=2e-4, bs=16 per GPU, 512x512 BEV,
80 000-point clouds, 4 GRU iterations, fp32) on N MI355X of one node.  One process per GPU (torchrun), RCCL.

A "step" = one full training pass over one synthetic, HBM-resident batch: ego-motion, pillarise both clouds, UNet
forward, GRU decoder, gt gather + deflowLoss, complete backward, gradient all-reduce (N > 1), Adam.  Nothing is
skipped or cached between steps.  Prints ONE JSON line on rank 0 (contract in the round prompt), including
  roofline      live HIP-event timing of the dominant kernel (fp32 MFMA implicit-GEMM conv) vs the 157.3 TFLOP/s peak
  cpu_baseline  the CPU oracle (oracle/ref_torch.py, a PyTorch port of the reference algorithm) timed on this host.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PER_GPU_BATCH = 16
N_POINTS = 80000
GRID = 512
NUM_ITERS = 4
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense


def cpu_baseline():
    """One training step of the oracle on ONE frame pair of the same workload (bounded sample), all host threads."""
    from deflow_amd.synth import synth_batch
    from oracle import ref_torch as O
    torch.manual_seed(0)
    ref = O.DeFlow(grid_feature_size=[GRID, GRID], num_iters=NUM_ITERS).train()
    opt = torch.optim.Adam(ref.parameters(), lr=2e-4)
    batch = synth_batch(1, N_POINTS, seed=20240116)
    t0 = time.perf_counter()
    opt.zero_grad()
    loss = O.training_loss(ref(batch), batch)
    loss.backward()
    opt.step()
    dt = time.perf_counter() - t0
    return {"value": 1.0 / dt, "unit": "frame-pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 training step on 1 synthetic frame pair ({N_POINTS} pts/cloud, {GRID}x{GRID}, {NUM_ITERS} GRU iters), "
                      f"oracle/ref_torch.py fp32, {dt:.1f} s; host cpu_count={os.cpu_count()}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=PER_GPU_BATCH, help="per-GPU batch (the metric is quoted at 16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--forward-only", action="store_true",
                    help="also time BASELINE configs[1] (B=1 inference) after the training steps; off by default so that a\n"
                         "rocprofv3 trace of the default command holds training launches only")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-launch HIP events of the conv kernels")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    assert torch.cuda.is_available(), "bench.py measures the HIP path; there is no CPU fallback"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    use_dist = world > 1 or "RANK" in os.environ  # under torch.distributed.run always go through RCCL (also at N=1)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # "nccl" == RCCL on ROCm

    import deflow_amd
    from deflow_amd import ops
    from deflow_amd.optim import Trainer
    from deflow_amd.synth import synth_batch

    torch.manual_seed(0)
    model = deflow_amd.DeFlow(grid_feature_size=[GRID, GRID], num_iters=NUM_ITERS).to(dev).train()
    trainer = Trainer(model, lr=2e-4)
    # weak scaling: every rank owns its own shard of frame pairs (seeded by global sample index), resident in HBM
    batch = synth_batch(args.batch, N_POINTS, seed=Trainer.shard_seed(20240116, rank, args.batch), device=dev)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.step(batch)
    prof = None if args.no_profile else ops.KernelProfiler()
    barrier()
    ops.PROFILER = prof
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.step(batch)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    ops.PROFILER = None
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    if rank != 0:
        dist.destroy_process_group()
        return

    ms = dt / args.steps * 1e3
    out = {
        "metric": "frame-pairs/sec training (deflow, bs=16, 512x512 BEV)",
        "value": world * args.batch * args.steps / dt, "unit": "frame-pairs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "deflow train step (deflowLoss, Adam lr=2e-4): BASELINE configs[2] per GPU", "per_gpu_batch": args.batch,
                   "global_batch": world * args.batch, "points_per_cloud": N_POINTS, "bev": [GRID, GRID], "gru_iters": NUM_ITERS,
                   "parallelism": f"dp{world}", "loss": float(loss)},
    }
    if prof is not None and os.environ.get("DF_BENCH_DUMP"):
        with open(os.environ["DF_BENCH_DUMP"], "w") as f:
            per = len(prof.records) // args.steps
            for name, flops, e0, e1, *_ in prof.records[:per]:
                ms_ = e0.elapsed_time(e1)
                f.write(f"{name:28s} {flops / 1e9:10.2f} GF {ms_:8.3f} ms {flops / ms_ / 1e9:8.1f} TF/s\n")
    if prof is not None:
        summ = prof.summary()
        conv = {k: v for k, v in summ.items() if k.startswith("conv_")}
        dom = max(conv, key=lambda k: conv[k]["ms"])
        d = conv[dom]
        achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
        traffic = None  # HBM bytes per launch from the committed PMC passes (profiles/pmc_traffic.json), same workload
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                traffic = json.load(f)["kernels"][dom]["hbm_bytes_per_launch"] if args.batch == PER_GPU_BATCH else None
        except Exception:
            traffic = None
        out["roofline"] = {"bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic, "kernel": dom,
                           "launches_per_step": d["launches"] / args.steps, "avg_launch_ms": d["ms"] / d["launches"],
                           "algorithmic_gflop_per_launch": d["flops"] / d["launches"] / 1e9,
                           "all_mfma_kernels": {k: {"launches_per_step": v["launches"] / args.steps,
                                                    "ms_per_step": v["ms"] / args.steps,
                                                    "tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12} for k, v in summ.items()}}
    if world == 1 and args.forward_only:
        # forward-only pairs/s of BASELINE configs[1] (one 80k-point pair, eval mode) beside the headline
        model.eval()
        b1 = synth_batch(1, N_POINTS, seed=20240116, device=dev)
        with torch.no_grad():
            for _ in range(3):
                model.forward_padded(b1)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(20):
                model.forward_padded(b1)
            torch.cuda.synchronize()
            fwd_ms = (time.perf_counter() - t1) / 20 * 1e3
        model.train()
        out["forward_only"] = {"workload": "BASELINE configs[1]: deflow inference, 1 pair (B=1), 80000 pts/cloud, 512x512, 4 GRU iters",
                               "ms_per_pair": fwd_ms, "pairs_per_s": 1e3 / fwd_ms,
                               "algorithmic_tflops": 391.6e9 / (fwd_ms * 1e-3) / 1e12}
    if use_dist:
        dist.destroy_process_group()
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
