"""CPU: the data-parallel plumbing (flat parameter arena, rank-0 broadcast, gradient all-reduce + mean scale,
per-rank data shards) with world_size 2 over gloo.  The HIP kernels are not involved."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import deflow_amd
    from deflow_amd.optim import Trainer
    from deflow_amd.synth import synth_batch
    torch.manual_seed(100 + rank)  # different init per rank on purpose: the Trainer must broadcast rank 0's arena
    model = deflow_amd.DeFlow(voxel_size=[0.2, 0.2, 6], point_cloud_range=[-6.4, -6.4, -3, 6.4, 6.4, 3],
                              grid_feature_size=[64, 64], num_iters=2)
    tr = Trainer(model, lr=2e-4)
    p_sum = float(tr.flat.param.double().sum())
    # rank-specific gradients -> all-reduce sum, scale = 1/world
    g = torch.Generator().manual_seed(7 + rank)
    tr.flat.grad.copy_(torch.randn(tr.flat.numel, generator=g))
    local = tr.flat.grad.clone()
    scale = tr.reduce_gradients()
    others = [torch.randn(tr.flat.numel, generator=torch.Generator().manual_seed(7 + r)) for r in range(world)]
    want_mean = torch.stack(others).sum(0) * scale
    err = float((tr.flat.grad * scale - want_mean).abs().max())
    # bucketed delivery (optim.GradSink): phase-wise copies into the arena + async all-reduce of each contiguous run,
    # including a phase that leaves a hole (delivered later) -- must equal one all-reduce of everything
    from deflow_amd.autograd import GradDict
    tr.flat.zero_grad()
    tr.sink.begin()
    gd, want_sink = GradDict(), torch.zeros(tr.flat.numel)
    for r in range(world):
        gr = torch.Generator().manual_seed(1000 + r)
        for n, p in tr.flat.named:
            t = torch.randn(p.shape, generator=gr)
            if r == rank:
                gd[p] = t
            off, k = tr.flat.slots[n]
            want_sink[off:off + k] += (t.permute(0, 2, 3, 1) if t.dim() == 4 else t).reshape(-1)
    bb = model.backbone
    hole = bb.encoder_step_2[2].conv.weight
    tr.sink.deliver(model.head.parameters(), gd)
    tr.sink.deliver([p for m in (bb.decoder_step1, bb.decoder_step2, bb.decoder_step3, bb.decoder_step4) for p in m.parameters()], gd)
    for stage in (bb.encoder_step_3, bb.encoder_step_2, bb.encoder_step_1):
        tr.sink.deliver([p for p in stage.parameters() if p is not hole], gd)
    n_works = len(tr.sink.works)
    tr.sink.deliver(list(model.parameters()), gd)   # the embedder and the hole
    tr.reduce_gradients()
    sink_err = float((tr.flat.grad - want_sink).abs().max())
    all_delivered = all(tr.sink.was_delivered(p) for p in model.parameters())
    # DF_ONE_BUCKET fallback (GradSink.one_bucket): the phases only copy, ONE all-reduce over the whole arena after the last one
    tr.flat.zero_grad()
    tr.sink.begin()
    tr.sink.one_bucket = True
    tr.sink.deliver(model.head.parameters(), gd)
    tr.sink.deliver([p for p in bb.parameters()], gd)
    one_works_mid = len(tr.sink.works)
    tr.sink.deliver(list(model.parameters()), gd)
    tr.reduce_gradients()
    tr.sink.one_bucket = False
    one_err = float((tr.flat.grad - want_sink).abs().max())
    sink_err = max(sink_err, one_err + (1.0 if one_works_mid != 0 else 0.0))
    # parameter views see the arena; gradient views alias the gradient arena
    w = model.backbone.decoder_step4.weight
    alias_ok = w.grad.data_ptr() >= tr.flat.grad.data_ptr() and w.data_ptr() >= tr.flat.param.data_ptr()
    # shards: the two ranks draw different frame pairs
    seed = Trainer.shard_seed(20240116, rank, 2)
    b = synth_batch(2, 64, seed=seed, grid_hw=(64, 64))
    # BatchNorm buffers drift apart during training (rank-local statistics); sync_buffers() = DDP's broadcast_buffers
    bn = model.backbone.encoder_step_1[0].batchnorm
    with torch.no_grad():
        bn.running_mean.fill_(1.0 + rank); bn.running_var.fill_(2.0 + rank); bn.num_batches_tracked.fill_(5 + rank)
    tr.sync_buffers()
    buf_ok = float(bn.running_mean[0]) == 1.0 and float(bn.running_var[3]) == 2.0 and int(bn.num_batches_tracked) == 5
    q.put((rank, p_sum, err, scale, alias_ok, float(b["pc0"][0, 0, 0]), float(local.abs().sum()), sink_err, all_delivered, n_works, buf_ok))
    dist.destroy_process_group()


def test_two_rank_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, s0, e0, sc0, a0, x0, l0, se0, ad0, nw0, bo0), (r1, s1, e1, sc1, a1, x1, l1, se1, ad1, nw1, bo1) = res
    assert bo0 and bo1, "sync_buffers must leave rank 0's BatchNorm statistics on every rank"
    assert se0 < 1e-5 and se1 < 1e-5 and ad0 and ad1, "bucketed gradient delivery must equal one all-reduce"
    assert nw0 == nw1 and nw0 >= 6, "head, UNet decoder and the encoder stages go out as separate overlapped buckets"
    assert s0 == s1, "rank 1 must hold rank 0's parameters after the broadcast"
    assert e0 < 1e-5 and e1 < 1e-5 and sc0 == sc1 == 0.5
    assert a0 and a1
    assert x0 != x1, "ranks must own different frame pairs (weak scaling shards)"
    assert l0 != l1


def test_flat_arena_layout():
    import deflow_amd
    from deflow_amd.optim import FlatParams
    torch.manual_seed(0)
    m = deflow_amd.DeFlow(voxel_size=[0.2, 0.2, 6], point_cloud_range=[-6.4, -6.4, -3, 6.4, 6.4, 3],
                          grid_feature_size=[64, 64])
    before = {k: v.clone() for k, v in m.state_dict().items()}
    fp = FlatParams(m)
    after = m.state_dict()
    for k, v in before.items():
        assert torch.equal(after[k], v), k                      # values and logical shapes unchanged
    g = m.head.gru
    assert g.convz.weight.data_ptr() + g.convz.weight.numel() * 4 == g.convr.weight.data_ptr()   # packed [z | r]
    assert g.convz.bias.data_ptr() + 128 * 4 == g.convr.bias.data_ptr()
    w = m.backbone.encoder_step_1[0].conv.weight
    assert w.permute(0, 2, 3, 1).is_contiguous()                   # O,kh,kw,I memory: what the MFMA kernels read
    assert all(p.data_ptr() % 16 == 0 for n, p in m.named_parameters() if not n.endswith(("convr.weight", "convr.bias")))
    assert fp.numel % 4 == 0 and fp.numel >= sum(p.numel() for p in m.parameters())
    # load_state_dict keeps the views inside the arena
    m.load_state_dict(before)
    assert w.data_ptr() >= fp.param.data_ptr() and w.data_ptr() < fp.param.data_ptr() + fp.numel * 4
    # gradient accumulation lands in the gradient arena in place
    fp.zero_grad()
    (m.head.decoder[2].weight.sum() * 2.0).backward()
    assert float(fp.grad.sum()) == 2.0 * 96


def test_timer_api():
    from deflow_amd import Timing
    t = Timing()
    t.start("Total")
    t[0].start("a"); t[0][1].start("b"); t[0][1].stop(); t[0].stop()
    assert t[0].count == 1 and t[0][1].count == 1 and "a" in t.report()
