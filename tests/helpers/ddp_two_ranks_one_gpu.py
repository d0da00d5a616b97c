"""Run under `python -m torch.distributed.run --nproc-per-node 2` on a 1-GPU box: TWO data-parallel ranks of the real engine
(HIP kernels, GradSink bucketed asynchronous all-reduce from inside the backward, flat-arena Adam) sharing cuda:0, with the
`gloo` backend carrying the collectives (RCCL cannot put two ranks on one device; the RCCL calls themselves are covered by
rccl_world1.py).  Each rank trains on its own shard; after two steps rank 0 writes its parameters and the losses, and the
parent test compares them with the oracle stepping on the average of the two per-rank gradients."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out, sync_bn = sys.argv[1], len(sys.argv) > 2 and sys.argv[2] == "sync_bn"
    graph = len(sys.argv) > 2 and sys.argv[2].startswith("graph")          # "graph" | "graph_bf16"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from test_gpu_model import build_pair, make_batch, to_dev
    from deflow_amd.optim import Trainer
    dev = torch.device("cuda", 0)
    _, model = build_pair(dev, 41 + 100 * rank, decoder_option="gru", num_iters=2)   # different init per rank on purpose:
    model.train()                                                                     # the Trainer broadcasts rank 0's
    if graph:
        _, m2 = build_pair(dev, 41 + 100 * rank, decoder_option="gru", num_iters=2)   # a second, identical replica
        batch = to_dev(make_batch(2, 1500, 7000 + 50 * rank), dev)
        return graph_mode(out, rank, dev, model, m2.train(), batch, "bf16" if sys.argv[2].endswith("bf16") else "fp32")
    tr = Trainer(model, lr=2e-4, sync_bn=sync_bn)
    assert tr.collective and tr.world == 2
    batch = to_dev(make_batch(2, 1500, 7000 + 50 * rank), dev)                         # rank-specific shard
    losses, works = [], 0
    for _ in range(2):
        tr.flat.zero_grad(); tr.sink.begin()
        model.forward_padded(batch)
        loss = tr.loss_on_last_forward(batch)
        loss.backward()
        works += len(tr.sink.works)
        tr.opt.step(grad_scale=tr.reduce_gradients())
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    gathered = [None, None]
    dist.all_gather_object(gathered, {"losses": losses, "works": works, "param_sum": float(tr.flat.param.double().sum())})
    if rank == 0:
        torch.save({"state": {k: v.detach().cpu() for k, v in model.state_dict().items()}, "ranks": gathered}, out)
    dist.barrier()
    dist.destroy_process_group()


def graph_mode(out, rank, dev, model, m2, batch, dtype):
    """the data-parallel step captured as HIP-graph SEGMENTS split at the gradient buckets (optim.SegmentedCapture), replayed
    on changing batches, against the eager data-parallel step from the same start: same losses, bit-identical parameters,
    Adam moments and BatchNorm buffers on both ranks (gloo's two-rank sum is order-independent); plus the host time of a replay"""
    import time
    from test_gpu_model import make_batch, to_dev
    from deflow_amd.optim import Trainer
    t1, t2 = Trainer(model, lr=1e-3, dtype=dtype), Trainer(m2, lr=1e-3, dtype=dtype)
    seq = [batch, to_dev(make_batch(2, 1500, 9000 + 50 * rank), dev), batch]
    want = [float(t1.step(b)) for b in seq]
    t2.capture(seq[0])
    got = [float(t2.step_captured(b)) for b in seq]
    torch.cuda.synchronize()
    kinds = [o[0] for o in t2._program]
    n_graph, n_ar = kinds.count("graph"), sum(len(o[1]) for o in t2._program if o[0] == "allreduce")
    same = bool(torch.equal(t1.flat.param, t2.flat.param) and torch.equal(t1.opt.exp_avg_sq, t2.opt.exp_avg_sq)
                and all(torch.equal(a, b) for a, b in zip(model.buffers(), m2.buffers())))
    diag = {"param": float((t1.flat.param - t2.flat.param).abs().max()), "exp_avg_sq": float((t1.opt.exp_avg_sq - t2.opt.exp_avg_sq).abs().max()),
            "buffers": max(float((a.double() - b.double()).abs().max()) for a, b in zip(model.buffers(), m2.buffers())),
            "n_param_diff": int((t1.flat.param != t2.flat.param).sum())}
    if os.environ.get("DF_DIAG"):
        bad = [n for (n, p), (_, q) in zip(t1.flat.named, t2.flat.named) if not torch.equal(p, q)]
        print(f"[rank {rank}] diag {diag} differing params: {bad[:12]} ({len(bad)})", flush=True)
        w1_, w2_ = t1.flat.param[:352].double(), t2.flat.param[:352].double()
        print(f"[rank {rank}] hash eager {float((w1_ * torch.arange(1, 353, device=dev)).sum()):.12e} captured {float((w2_ * torch.arange(1, 353, device=dev)).sum()):.12e}", flush=True)
    steps = (t1.opt.step_count, t2.opt.step_count, int(t2.opt.step_dev))
    param_sum = float(t2.flat.param.double().sum())
    # host cost of one replayed data-parallel step (gloo's all_reduce blocks the host on the device, so time the graph
    # launches alone: the collectives are a handful of calls either way)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for o in t2._program:
        if o[0] == "graph":
            o[1].replay()
    host_ms = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    res = {"want": want, "got": got, "same": same, "n_graph": n_graph, "n_allreduce": n_ar, "kinds": kinds, "host_ms": host_ms, "diag": diag,
           "steps": steps, "param_sum": param_sum}
    gathered = [None, None]
    dist.all_gather_object(gathered, res)
    if rank == 0:
        torch.save({"ranks": gathered}, out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
