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
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from test_gpu_model import build_pair, make_batch, to_dev
    from deflow_amd.optim import Trainer
    dev = torch.device("cuda", 0)
    _, model = build_pair(dev, 41 + 100 * rank, decoder_option="gru", num_iters=2)   # different init per rank on purpose:
    model.train()                                                                     # the Trainer broadcasts rank 0's
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


if __name__ == "__main__":
    main()
