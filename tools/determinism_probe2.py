import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, deflow_amd
from deflow_amd import ops
import deflow_amd.unet as U
from deflow_amd._lib import img, img_pair, call, ptr, stream
from deflow_amd.synth import synth_batch
dev = torch.device("cuda"); torch.manual_seed(0)
m = deflow_amd.DeFlow().to(dev).train(True)
B = 2
batch = synth_batch(B, 80000, device=dev)
with torch.no_grad():
    c = torch.zeros(B, 512, 512, 64, device=dev)
    m.embedder.pillarize(batch["pc0"].contiguous().float(), img(c, 32, 0), True)
    m.embedder.pillarize(batch["pc1"].contiguous().float(), img(c, 32, 32), True)
    x = img_pair(c, 32)
    cur = x
    for li, layer in enumerate(list(m.backbone.encoder_step_1) + list(m.backbone.encoder_step_2)[:2]):
        outs = []
        for rep in range(3):
            h = (cur.h + 2 - 3) // layer.stride + 1
            z = torch.empty(2 * B, h, h, layer.conv.out_channels, device=dev)
            tape = []
            U._cwn_forward(layer, cur, img(z), 2 * B, 2, True, tape)
            _, _, _, y, bn_ss, ipg, groups = tape[0]
            outs.append((z.clone(), y.clone(), bn_ss.clone()))
        dz = max(float((o[0] - outs[0][0]).abs().max()) for o in outs[1:])
        dy = max(float((o[1] - outs[0][1]).abs().max()) for o in outs[1:])
        ds = max(float((o[2] - outs[0][2]).abs().max()) for o in outs[1:])
        print(f"layer {li} {layer.conv.in_channels}->{layer.conv.out_channels} s{layer.stride} @{h}: repeat diffs  y(conv) {dy:.3e}  bn_ss {ds:.3e}  z {dz:.3e}")
        zkeep = outs[0][0]
        cur = img(zkeep)
