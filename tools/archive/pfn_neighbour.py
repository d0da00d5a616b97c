"""One kind of GPU work in a loop: the neighbour of tools/pfn_bwd_stress.py (round 4: which neighbour kernel makes the SLP-built
pillar feature net backward return wrong sums -- and does it have to sit in another process?)
    python tools/pfn_neighbour.py <kind> [seconds]          kinds: see KINDS
in-process use: import pfn_neighbour; step = pfn_neighbour.make(kind, dev); step() enqueues one batch of work on the current stream"""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

KINDS = ("infer", "fwd_train", "train_linear", "embed", "torch_ops", "matmul_bf16", "conv_bf16", "conv_h2", "conv_fp32", "wgrad_bf16", "wgrad_h2", "bn_gelu", "train_bf16", "train_fp32")


def _install_filter():
    """DF_NB_ONLY=<regex>: of this library's entry points only the pillar pipeline (always: the rest of the forward needs its counts) and
    the ones matching the regex run; every other call returns at once (the neighbour's results are garbage, which nobody reads) --
    bisects WHICH kernel of a forward pass the victim reacts to.  Patches the `call` every deflow_amd module imported."""
    import re
    pat = os.environ.get("DF_NB_ONLY")
    if not pat:
        return
    import deflow_amd  # noqa: F401
    from deflow_amd import _lib
    rx = re.compile(pat)
    always = re.compile(r"df_pillar2|df_pfn_bn|df_ego|df_cell|df_version|_ok$|_splits$|df_conv2d_tile_m|df_conv2d_variant|df_conv2d_last_dma")
    real = _lib.call

    def filtered(name, *a):
        if always.search(name) or rx.search(name):
            return real(name, *a)
        return 0
    for m in list(sys.modules.values()):
        if getattr(m, "__name__", "").startswith("deflow_amd") and getattr(m, "call", None) is real:
            m.call = filtered


def make(kind, dev):
    from deflow_amd import ops
    _install_filter()
    from deflow_amd._lib import img, call
    g = torch.Generator().manual_seed(1)
    if kind == "matmul_bf16":
        a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16); b = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
        return lambda: [a @ b for _ in range(4)]
    if kind in ("conv_bf16", "conv_h2", "conv_fp32"):
        x = torch.randn(8, 128, 128, 128, generator=g).to(dev); y = torch.empty(8, 128, 128, 128, device=dev)
        w = (torch.randn(128, 3, 3, 128, generator=g) * 0.03).to(dev); b = torch.zeros(128, device=dev)
        def step():
            ops.MFMA_BF16 = kind == "conv_bf16"
            if kind == "conv_fp32":
                call("df_conv2d", img(x), ops.ptr(w), ops.ptr(b), img(y), 3, 1, 1, ops.CONV_FWD, ops.EPI_BIAS, None, None, None, 0, ops.stream())
            else:
                ops.conv2d(img(x), w, b, img(y), 3, 1)
            ops.MFMA_BF16 = False
        return step
    if kind in ("wgrad_bf16", "wgrad_h2"):
        dt = torch.bfloat16 if kind == "wgrad_bf16" else torch.float32
        x = torch.randn(8, 128, 128, 128, generator=g).to(dev).to(dt); dy = torch.randn(8, 128, 128, 128, generator=g).to(dev).to(dt)
        dw = torch.empty(128, 3, 3, 128, device=dev)
        return lambda: ops.conv2d_wgrad(img(x), img(dy), 3, 1, dw)
    if kind == "bn_gelu":
        y = torch.randn(8, 128, 128, 128, generator=g).to(dev); z = torch.empty_like(y)
        ss = torch.ones(4, 128, device=dev)
        return lambda: ops.bn_gelu_apply(y, ss, 8, img(z))
    if kind in ("train_bf16", "train_fp32"):
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
        from deflow_amd.optim import Trainer
        from test_gpu_model import build_pair, make_batch, to_dev
        _, model = build_pair(dev, 41, decoder_option="gru", num_iters=2)
        model.train()
        batch = to_dev(make_batch(2, 1500, 7000), dev)
        tr = Trainer(model, lr=0.0, dtype="bf16" if kind == "train_bf16" else "fp32")
        return lambda: tr.step(batch)
    if kind in ("infer", "fwd_train", "train_linear", "embed", "torch_ops"):
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
        from deflow_amd.optim import Trainer
        from test_gpu_model import build_pair, make_batch, to_dev
        _, model = build_pair(dev, 41, decoder_option="linear" if kind == "train_linear" else "gru", num_iters=2)
        batch = to_dev(make_batch(2, 1500, 7000), dev)
        if kind == "train_linear":
            model.train()
            tr = Trainer(model, lr=0.0)
            return lambda: tr.step(batch)
        if kind == "torch_ops":            # a stream of small torch kernels (fills, copies, adds): no kernel of this library
            bufs = [torch.empty(1 << 16, device=dev) for _ in range(8)]
            def step():
                for b_ in bufs:
                    b_.zero_(); b_.add_(1.0)
                return bufs[0].sum()
            return step
        if kind == "embed":
            model.eval()
            def step():
                with torch.no_grad():
                    model.embedder(batch["pc0"])
            return step
        model.train() if kind == "fwd_train" else model.eval()
        def step():
            with torch.no_grad():
                model.forward_padded(batch)
        return step
    raise SystemExit(f"kind must be one of {KINDS}")


if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    step = make(sys.argv[1], dev)
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
    t0, n = time.time(), 0
    while time.time() - t0 < secs:
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        n += 20
    print(f"neighbour {sys.argv[1]}: {n} steps in {time.time() - t0:.0f} s")
