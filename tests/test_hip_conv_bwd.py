"""Convolution backward building blocks (SURVEY.md 8f-3 groundwork) against torch autograd on the CPU:
pf_conv2d_wgrad (weight + bias gradient, transposed-LDS-read MFMA kernel) and the data gradient through
pf_conv2d on flipped / transposed weights (engine.Conv.dgrad_of)."""
import pytest
import torch

import golden_cases as gc
import kernel_cases as kc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from prior_flow_amd import _lib
    return _lib.load()


CASES = [  # kh, kw, c0, c1, cout, B, H, W
    (3, 3, 64, 0, 64, 2, 8, 32),
    (3, 3, 128, 0, 256, 1, 16, 64),
    (1, 5, 128, 256, 256, 1, 16, 32),          # GRU gates: [h | x] virtual concat
    (5, 1, 128, 256, 128, 2, 12, 32),
    (1, 1, 324, 0, 256, 1, 16, 32),            # convc1: ragged channel tail
    (3, 3, 72, 0, 126, 2, 7, 45),              # partial tiles, Cout not a tile multiple (126 % 4 != 0 -> 128 padded)
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c)))
def test_wgrad_and_dgrad_vs_autograd(lib, case):
    from prior_flow_amd._lib import EPI_LINEAR, PREC_BF16X3
    from prior_flow_amd.engine import Conv
    kh, kw, c0, c1, cout, B, H, W = case
    dev = torch.device("cuda")
    cin = c0 + c1
    tag = "bw/" + "x".join(map(str, case))
    x = gc.uni(tag + "/x", (B, cin, H, W), -1, 1).requires_grad_(True)
    w = gc.uni(tag + "/w", (cout, cin, kh, kw), -0.1, 0.1).requires_grad_(True)
    b = gc.uni(tag + "/b", (cout,), -0.1, 0.1).requires_grad_(True)
    gy = gc.uni(tag + "/gy", (B, cout, H, W), -1, 1)
    y = torch.nn.functional.conv2d(x, w, b, padding=(kh // 2, kw // 2))
    (y * gy).sum().backward()
    # device buffers: activations and dY live in wider row buffers (column offsets like the engine's)
    cpad = (cout + 3) // 4 * 4
    dy = torch.zeros(B * H * W, cpad + 8, device=dev)
    dy[:, 4:4 + cout] = kc.cl(gy).to(dev)
    x0 = torch.full((B * H * W, c0 + 4), 3.0, device=dev)
    x0[:, :c0] = kc.cl(x.detach()[:, :c0]).to(dev)
    x1 = None
    if c1:
        x1 = torch.full((B * H * W, c1 + 8), -2.0, device=dev)
        x1[:, 8:8 + c1] = kc.cl(x.detach()[:, c0:]).to(dev)
    cin_pad, cout_pad = (cin + 31) // 32 * 32, (cout + 127) // 128 * 128
    dw = torch.zeros(cout_pad, kh * kw, cin_pad, device=dev)
    db = torch.zeros(cout_pad, device=dev)
    lib.conv2d_wgrad(x0, 0, c0, dy, 4, cpad, dw, db, kh, kw, B, H, W, x1=x1, off1=8, c1=c1)
    got_w = Conv.unpack_wgrad(dw, cout, cin, kh, kw).cpu()
    scale = float(w.grad.abs().max())
    kc.check(got_w, w.grad, 3e-5 * scale + 1e-6, "weight gradient")
    kc.check(db[:cout], b.grad, 3e-5 * float(b.grad.abs().max()) + 1e-6, "bias gradient")
    assert float(dw[cout:].abs().max() if cout < cout_pad else 0.0) == 0.0 and float(dw[:, :, cin:].abs().max() if cin < cin_pad else 0.0) == 0.0
    # accumulation semantics: a second call doubles
    lib.conv2d_wgrad(x0, 0, c0, dy, 4, cpad, dw, None, kh, kw, B, H, W, x1=x1, off1=8, c1=c1)
    kc.check(Conv.unpack_wgrad(dw, cout, cin, kh, kw).cpu(), 2 * w.grad, 6e-5 * scale + 2e-6, "weight gradient accumulates")
    # data gradient through the forward kernel
    if cout % 4 == 0:
        dg = Conv.dgrad_of(w.detach().to(dev), PREC_BF16X3)
        dx = torch.empty(B * H * W, cin, device=dev)
        lib.conv2d([dg.desc(dy, 4, cout, dx, 0, EPI_LINEAR)], B, H, W, dy)
        kc.check(kc.uncl(dx.cpu(), B, H, W), x.grad, 2e-4 * float(x.grad.abs().max()), "data gradient")


@pytest.mark.parametrize("case", [(3, 64, 7, 2, 2, 36, 52), (2, 128, 7, 1, 2, 17, 27), (4, 64, 3, 1, 1, 9, 10)],
                         ids=lambda c: "x".join(map(str, c)))
def test_small_cin_wgrad_vs_autograd(lib, case):
    """pf_conv2d_wgrad_small (the 7x7 stems' weight / bias gradient: 3->64 stride 2, 2->128 stride 1) against torch
    autograd, on maps with partial 8x8 tiles; accumulation semantics; NCHW and channel-last inputs."""
    cin, cout, k, stride, B, Ho, Wo = case
    dev = torch.device("cuda")
    tag = "bws/" + "x".join(map(str, case))
    x = gc.uni(tag + "/x", (B, cin, Ho * stride, Wo * stride), -1, 1)
    w = gc.uni(tag + "/w", (cout, cin, k, k), -0.1, 0.1).requires_grad_(True)
    b = gc.uni(tag + "/b", (cout,), -0.1, 0.1).requires_grad_(True)
    gy = gc.uni(tag + "/gy", (B, cout, Ho, Wo), -1, 1)
    y = torch.nn.functional.conv2d(x, w, b, stride=stride, padding=k // 2)
    assert y.shape[-2:] == (Ho, Wo)
    (y * gy).sum().backward()
    dy = torch.zeros(B * Ho * Wo, cout + 8, device=dev)
    dy[:, 4:4 + cout] = kc.cl(gy).to(dev)
    dw, db = torch.zeros(cout, cin, k, k, device=dev), torch.zeros(cout, device=dev)
    lib.conv2d_wgrad_small(x.to(dev), True, 0, cin, dy, 4, cout, dw, db, k, k, stride, B, Ho, Wo)
    tol = 3e-6 * float(w.grad.abs().max()) + 1e-6
    kc.check(dw, w.grad, tol, "weight gradient (NCHW input)")
    kc.check(db, b.grad, 3e-6 * float(b.grad.abs().max()) + 1e-6, "bias gradient")
    xcl = torch.full((B * Ho * stride * Wo * stride, cin + 3), 5.0, device=dev)
    xcl[:, 2:2 + cin] = kc.cl(x).to(dev)
    lib.conv2d_wgrad_small(xcl, False, 2, cin, dy, 4, cout, dw, None, k, k, stride, B, Ho, Wo)      # accumulates
    kc.check(dw, 2 * w.grad, 2 * tol, "weight gradient accumulates (channel-last input)")


def test_training_ops_without_pytorch_kernels(lib):
    """The autograd Functions that replaced the last PyTorch-ROCm operators of a training step against torch autograd on the
    CPU: stride-2 convolutions (zero-stuffed gradient), the 7x7 stems, frozen BatchNorm (dx, d gamma, d beta)."""
    import torch.nn as nn
    from prior_flow_amd import autograd as ag
    dev = torch.device("cuda")

    def compare(mod, x, hip_fn, tol, needs_dx=True):
        xc = x.clone().requires_grad_(needs_dx)
        ref = mod(xc)
        gy = gc.uni("tops/gy/" + type(mod).__name__ + str(tuple(ref.shape)), tuple(ref.shape), -1, 1)
        (ref * gy).sum().backward()
        want = {k: p.grad.clone() for k, p in mod.named_parameters()}
        mod.zero_grad()
        md = mod.to(dev)
        xd = x.to(dev).clone().requires_grad_(needs_dx)
        ag.STATS["hip"] = ag.STATS["torch"] = 0
        out = hip_fn(md, xd)
        kc.check(out.detach(), ref.detach(), tol * float(ref.abs().max()), "forward")
        (out * gy.to(dev)).sum().backward()
        assert ag.STATS["torch"] == 0 and ag.STATS["hip"] >= 2, ag.STATS
        for k, p in md.named_parameters():
            kc.check(p.grad, want[k], tol * float(want[k].abs().max()) + 1e-6, "d " + k)
        if needs_dx:
            kc.check(xd.grad, xc.grad, tol * float(xc.grad.abs().max()), "dx")
        mod.cpu()

    torch.manual_seed(3)
    compare(nn.Conv2d(64, 96, 3, stride=2, padding=1), gc.uni("tops/x1", (2, 64, 20, 36), -1, 1), lambda m, x: ag.conv2d(x, m), 3e-4)
    compare(nn.Conv2d(96, 128, 1, stride=2), gc.uni("tops/x2", (1, 96, 18, 10), -1, 1), lambda m, x: ag.conv2d(x, m), 3e-4)
    compare(nn.Conv2d(3, 64, 7, stride=2, padding=3), gc.uni("tops/x3", (2, 3, 40, 56), -1, 1), lambda m, x: ag.conv2d(x, m), 1e-5, needs_dx=False)
    compare(nn.Conv2d(2, 128, 7, padding=3), gc.uni("tops/x4", (1, 2, 17, 27), -3, 3), lambda m, x: ag.conv2d(x, m), 1e-5, needs_dx=False)
    bn = nn.BatchNorm2d(96)
    with torch.no_grad():
        bn.weight.copy_(gc.uni("tops/bn/w", (96,), 0.5, 1.5)); bn.bias.copy_(gc.uni("tops/bn/b", (96,), -0.3, 0.3))
        bn.running_mean.copy_(gc.uni("tops/bn/m", (96,), -0.5, 0.5)); bn.running_var.copy_(gc.uni("tops/bn/v", (96,), 0.5, 2.0))
    bn.eval()
    compare(bn, gc.uni("tops/x5", (2, 96, 12, 20), -2, 2), lambda m, x: ag._norm(m, x), 2e-5)
