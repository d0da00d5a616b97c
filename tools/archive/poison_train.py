"""uninitialised-memory / determinism probe of the training step: the gradient arena must be bit-identical when the same
step is repeated, also after the allocator's free blocks were filled with NaN"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, deflow_amd
from deflow_amd.optim import Trainer
from deflow_amd.synth import synth_batch
dev = torch.device("cuda"); torch.manual_seed(0)
m = deflow_amd.DeFlow().to(dev).train()
tr = Trainer(m, lr=2e-4)
batch = synth_batch(4, 80000, device=dev)
def grads():
    tr.flat.zero_grad(); tr.sink.begin()
    m.forward_padded(batch)
    loss = tr.loss_on_last_forward(batch)
    loss.backward()
    return tr.flat.grad.clone(), float(loss)
g0, l0 = grads()
g1, l1 = grads()
m.last_state = None
torch.cuda.empty_cache()
junk = torch.full((8 * 1024 ** 3,), float("nan"), device=dev); del junk
g2, l2 = grads()
print("loss", l0, l1, l2)
print("repeat  max|dgrad|", float((g1 - g0).abs().max()), " poisoned max|dgrad|", float((g2 - g0).abs().max()),
      " nan in grads:", int(torch.isnan(g2).sum()), " max|grad|", float(g0.abs().max()))
