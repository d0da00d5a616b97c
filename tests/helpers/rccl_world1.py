"""Run under `python -m torch.distributed.run --nproc-per-node 1` on a 1-GPU box: the data-parallel code path over RCCL
(process-group init on the device, rank-0 broadcast of the parameter arena, the phase-wise asynchronous all-reduces issued
from inside the backward, the wait before Adam) on a 1-rank group, where the sum is the identity -- so two training steps
must leave exactly the parameters a trainer without collectives leaves."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def run(force: bool):
    import deflow_amd
    from deflow_amd.optim import Trainer
    from deflow_amd.synth import synth_batch
    if force:
        os.environ["DF_FORCE_COLLECTIVES"] = "1"
    else:
        os.environ.pop("DF_FORCE_COLLECTIVES", None)
    torch.manual_seed(0)
    dev = torch.device("cuda", 0)
    model = deflow_amd.DeFlow(grid_feature_size=[128, 128], point_cloud_range=[-12.8, -12.8, -3, 12.8, 12.8, 3],
                              num_iters=2).to(dev).train()
    tr = Trainer(model, lr=2e-4)
    assert tr.collective == force
    batch = synth_batch(2, 6000, seed=5, device=dev, grid_hw=(128, 128))
    issued = 0
    for _ in range(2):
        tr.flat.zero_grad()
        tr.sink.begin()
        model.forward_padded(batch)
        loss = tr.loss_on_last_forward(batch)
        loss.backward()
        issued += len(tr.sink.works)
        tr.opt.step(grad_scale=tr.reduce_gradients())
    torch.cuda.synchronize()
    if force:
        # the same two steps as a captured PROGRAM over RCCL: graph segments split at the buckets, real (1-rank) all-reduces
        # on the process group's stream between them, the stream-level wait before the Adam segment (optim.SegmentedCapture)
        torch.manual_seed(0)
        m2 = deflow_amd.DeFlow(grid_feature_size=[128, 128], point_cloud_range=[-12.8, -12.8, -3, 12.8, 12.8, 3],
                               num_iters=2).to(dev).train()
        t2 = Trainer(m2, lr=2e-4)
        t2.capture(batch)
        for _ in range(2):
            lg = t2.step_captured()
        torch.cuda.synchronize()
        kinds = [o[0] for o in t2._program]
        assert kinds.count("graph") >= 6 and kinds[-2:] == ["wait", "graph"], kinds
        # against the HAND-DRIVEN steps above: the same kernels, but outside a Trainer step the weights' fp16x2 planes take each
        # tensor's own max |w| as their bound and inside it the arena-wide one (ops.W_AMAX) -- two valid scales whose products agree
        # bit for bit only while no lo-plane element underflows; round 5 saw 2.3e-7 of a gradient's largest entry after the first Adam
        # step moved the arena's maximum (a BatchNorm weight of 1.0) across a power of two.  Stated: 2e-6 per tensor (the kernels'
        # own bound against float64), BatchNorm-shadowed conv biases (pure rounding noise) aside; the loss to 1e-6.
        assert abs(float(lg) - float(loss.detach())) <= 1e-6 * abs(float(loss.detach())), (float(lg), float(loss.detach()))
        for n_, _ in tr.flat.named:
            if n_.endswith(".conv.bias") and "backbone" in n_:
                continue
            o_, k_ = tr.flat.slots[n_]
            a_, b_ = tr.flat.grad[o_:o_ + k_], t2.flat.grad[o_:o_ + k_]
            assert float((a_ - b_).abs().max()) <= 2e-6 * float(a_.abs().max()), ("captured RCCL program vs hand-driven step", n_)
        print(f"RCCL_WORLD1_GRAPH_OK segments={kinds.count('graph')}")
        # the per-bucket trace bench.py reports at N > 1 (optim.GradSink.trace_on) and the DF_ONE_BUCKET fallback, over RCCL:
        # tracing must not change the step; one bucket must give the bits of the bucketed step (a 1-rank sum is the identity)
        torch.manual_seed(0)
        m3 = deflow_amd.DeFlow(grid_feature_size=[128, 128], point_cloud_range=[-12.8, -12.8, -3, 12.8, 12.8, 3],
                               num_iters=2).to(dev).train()
        t3 = Trainer(m3, lr=2e-4)
        t3.sink.trace_on(True)
        t3.step(batch)
        torch.cuda.synchronize()
        rep = t3.sink.trace_report()
        assert len(rep) >= 4 and all(r["completed_ms"] >= r["issued_ms"] >= 0 and r["in_flight_ms"] >= 0 for r in rep), rep
        assert 0.9 * t3.flat.numel * 4 <= sum(r["bytes"] for r in rep) <= t3.flat.numel * 4
        assert all(a["issued_ms"] <= b["issued_ms"] for a, b in zip(rep, rep[1:]))
        t3.sink.trace_on(False)
        t3.sink.one_bucket = True
        t3.step(batch)
        torch.cuda.synchronize()
        # ... and the Trainer's EAGER steps (traced, then one bucket) are the captured program's steps bit for bit
        assert torch.equal(t3.flat.param, t2.flat.param) and torch.equal(t3.flat.grad, t2.flat.grad), "traced + one-bucket eager steps != the captured program"
        print(f"RCCL_WORLD1_TRACE_OK buckets={len(rep)} in_flight_ms={[round(r['in_flight_ms'], 3) for r in rep]}")
    return tr.flat.param.clone(), tr.flat.grad.clone(), float(loss.detach()), issued


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]),
                            device_id=torch.device("cuda", 0))
    p1, g1, l1, n1 = run(True)
    p0, g0, l0, n0 = run(False)
    dist.barrier()
    dist.destroy_process_group()
    assert n1 >= 4 and n0 == 0, (n1, n0)          # several bucketed all-reduces per backward were really issued
    assert torch.equal(g1, g0) and torch.equal(p1, p0) and l1 == l0, (float((p1 - p0).abs().max()), l1, l0)
    print(f"RCCL_WORLD1_OK works={n1} loss={l1:.6f}")


if __name__ == "__main__":
    main()
