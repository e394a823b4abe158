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
