"""Parity of the training step against the oracle over many seeds and ragged shapes (confidence beyond the fixed-seed tests):
worst flow / loss / gradient error over the sweep."""
import sys, os
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
import torch
from test_gpu_model import build_pair, make_batch, to_dev, rel_err
from oracle import ref_torch as O
dev = torch.device("cuda")
worst = {"flow": 0.0, "loss": 0.0, "grad": 0.0}
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    B = 1 + seed % 3
    N = [777, 1500, 2049, 3000][seed % 4]
    opt = [dict(decoder_option="gru", num_iters=4), dict(decoder_option="gru", num_iters=1), dict(decoder_option="linear")][seed % 3]
    ref, mine = build_pair(dev, 100 + seed, **opt)
    ref.train(); mine.train()
    batch = make_batch(B, N, 3000 + 17 * seed)
    if seed % 5 == 0:
        batch["pc0"][0, N // 3:] = float("nan")
    res_r = ref(batch); loss_r = O.training_loss(res_r, batch); loss_r.backward()
    bd = to_dev(batch, dev)
    res_m = mine(bd); loss_m = O.training_loss(res_m, bd); loss_m.backward()
    ef = max(rel_err(res_m["flow"][b], res_r["flow"][b]) for b in range(B) if res_r["flow"][b].numel())
    el = rel_err(loss_m.reshape(1), loss_r.reshape(1))
    pr = dict(ref.named_parameters())
    eg = max(rel_err(p.grad, pr[k].grad) for k, p in mine.named_parameters() if not (k.endswith("conv.bias") and "encoder_step" in k))
    worst = {"flow": max(worst["flow"], ef), "loss": max(worst["loss"], el), "grad": max(worst["grad"], eg)}
    print(f"seed {seed}: B={B} N={N} {opt} flow {ef:.2e} loss {el:.2e} grad {eg:.2e}")
print("worst", worst)
assert worst["flow"] < 1e-4 and worst["loss"] < 1e-4 and worst["grad"] < 2e-3
