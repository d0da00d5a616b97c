"""train-mode forward must be bit-reproducible: component-wise probe"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, deflow_amd
from deflow_amd.synth import synth_batch
from deflow_amd._lib import img
dev = torch.device("cuda"); torch.manual_seed(0)
m = deflow_amd.DeFlow().to(dev).train(True)
batch = synth_batch(2, 80000, device=dev)
pc = batch["pc0"].contiguous().float()
emb = m.embedder
with torch.no_grad():
    cans = []
    for _ in range(3):
        c = torch.zeros(2, 512, 512, 64, device=dev)
        emb.pillarize(pc, img(c, 32, 0), True)
        emb.pillarize(batch["pc1"].contiguous().float(), img(c, 32, 32), True)
        cans.append(c)
    print("canvas repeat max|diff|", [float((c - cans[0]).abs().max()) for c in cans[1:]])
    vs = [m.backbone.run(cans[0], True, None).clone() for _ in range(3)]
    print("unet(train) repeat max|diff|", [float((v - vs[0]).abs().max()) for v in vs[1:]], "max|v|", float(vs[0].abs().max()))
    m.backbone.eval()
    ve = [m.backbone.run(cans[0], False, None).clone() for _ in range(3)]
    print("unet(eval) repeat max|diff|", [float((v - ve[0]).abs().max()) for v in ve[1:]])
    m.backbone.train()
    x0 = cans[1].clone()
    v1 = m.backbone.run(cans[1], True, None).clone()
    print("input modified by a train-mode run:", float((cans[1] - x0).abs().max()))
    # fresh model, same weights: is the FIRST train-mode run reproducible across model copies?
    import copy
    m2 = copy.deepcopy(m); m3 = copy.deepcopy(m)
    a = m2.backbone.run(x0, True, None).clone(); b = m3.backbone.run(x0, True, None).clone()
    print("two fresh copies, first run each: max|diff|", float((a - b).abs().max()))
    a2 = m2.backbone.run(x0, True, None).clone()
    print("same copy, second run vs first: max|diff|", float((a2 - a).abs().max()))
    # which state changed?
    for (n1, t1), (n2, t2) in zip(m2.backbone.state_dict().items(), m3.backbone.state_dict().items()):
        d = float((t1.float() - t2.float()).abs().max())
        if d > 0 and "running" not in n1 and "num_batches" not in n1:
            print("  param differs after an extra run:", n1, d)
