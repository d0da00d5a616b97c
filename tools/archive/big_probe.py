"""Maximum-size probes of the training step: (a) 1024 x 1024 grid, 160k points, B = 2; (b) B = 32 at 512 x 512, where the
concat activations pass 4 GB (32-bit DMA / buffer offsets no longer fit: the 64-bit paths must take over).  Prints loss,
gradient norm and time; run twice with and without DF_CONV_NO_DMA=1 DF_CONV_WIDE_EPI=1 to compare the two offset paths."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, deflow_amd
from deflow_amd.optim import Trainer
from deflow_amd.synth import synth_batch
dev = torch.device("cuda")
which = sys.argv[1]
torch.manual_seed(0)
if which == "grid1024":
    m = deflow_amd.DeFlow(voxel_size=[0.1, 0.1, 6], grid_feature_size=[1024, 1024], num_iters=8).to(dev).train()
    batch = synth_batch(2, 160000, device=dev)
else:
    m = deflow_amd.DeFlow().to(dev).train()
    batch = synth_batch(32, 80000, device=dev)
tr = Trainer(m, lr=2e-4)
for i in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.flat.zero_grad(); tr.sink.begin()
    m.forward_padded(batch)
    loss = tr.loss_on_last_forward(batch)
    loss.backward()
    torch.cuda.synchronize()
    g = tr.flat.grad
    print(which, "step", i, "loss %.6f" % float(loss.detach()), "grad norm %.6e" % float(g.double().norm()), "finite", bool(torch.isfinite(g).all()),
          "ms %.1f" % ((time.perf_counter() - t0) * 1e3), "peak GB %.1f" % (torch.cuda.max_memory_allocated() / 2**30))
