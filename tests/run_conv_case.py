"""Child process of tests/test_hip_kernels.py::test_conv_roles_kernel_matches_symmetric_kernel_bitwise:
   python tests/run_conv_case.py OUT.pt      (PRIORFLOW_CONV_WS picks the kernel form; it is read once per process)
Runs seeded bf16x3 convolutions of the update blocks' shapes through pf_conv2d and saves the outputs.
PF_CASE_SPLIT=1: the operands are handed over as split twins only (pf_split_bf16 of the same fp32 inputs; the fp32 pointers
are NULL) and every output is requested in both forms -- the all-DMA kernel; `twin_ok` records that each output twin equals
pf_split_bf16 of the fp32 output bit for bit."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from prior_flow_amd import _lib
from prior_flow_amd._lib import EPI_GRU_Q, EPI_GRU_ZR, EPI_LINEAR, EPI_RELU, PREC_BF16X3
from prior_flow_amd.engine import Conv, pack_mfma, split_twin

lib = _lib.load()
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(7)


def rnd(*shape, s=1.0):
    return (torch.rand(*shape, generator=g) * 2 - 1).mul_(s).to(dev)


def conv(cin, cout, kh, kw):
    w = rnd(cout, cin, kh, kw, s=(1.0 / (cin * kh * kw)) ** 0.5)
    wp, bp = pack_mfma(w, rnd(cout, s=0.1))
    return Conv(wp, bp, kh, kw, cin, cout, PREC_BF16X3)


out, roles, twin_ok = {}, {}, {}
SPLIT = os.environ.get("PF_CASE_SPLIT", "0") == "1"


def twin_of(t):
    return lib.split_bf16(t, split_twin(t.shape[0], t.shape[1], dev))


for B, H8, W8 in ((1, 64, 128), (2, 22, 40), (4, 64, 128), (8, 64, 128)):      # full tiles; ragged both ways; enough pixels for the 256-px tile; for tile 5
    N = B * H8 * W8
    x = rnd(N, 320)
    h = rnd(N, 128)
    z = torch.rand(N, 128, generator=g).to(dev)
    cases = {"relu3x3_256": (256, 256, 3, 3, EPI_RELU), "lin3x3_192": (128, 192, 3, 3, EPI_LINEAR),
             "zr1x5": (384, 256, 1, 5, EPI_GRU_ZR), "q5x1": (384, 128, 5, 1, EPI_GRU_Q), "lin1x5_96": (96, 128, 1, 5, EPI_LINEAR)}
    if B == 8:      # 64 output channels on a big map: pf_conv2d_tile 5 (8-row tile) -- the flow stems' 3x3 at batch
        cases = {"relu3x3_64": (128, 64, 3, 3, EPI_RELU)}
    for name, (cin, cout, kh, kw, epi) in cases.items():
        cv = [conv(cin, cout, kh, kw) for _ in range(2)]           # two groups, like branch A / branch B
        y = [torch.zeros(N, 256, device=dev) for _ in range(2)]
        aux = [torch.zeros(N, 128, device=dev) for _ in range(2)]
        kx = [dict() for _ in range(2)]
        xi, hi = x, h
        if SPLIT:
            xs, hs = twin_of(x), twin_of(h)
            ys = [split_twin(N, 256, dev) for _ in range(2)]
            auxs = [split_twin(N, 128, dev) for _ in range(2)]
            xi = hi = None
        if epi == EPI_GRU_ZR:
            if SPLIT:
                kx = [dict(in0s=hs, in1s=xs, outs=ys[i], auxs=auxs[i]) for i in range(2)]
            descs = [cv[i].desc(hi, 0, 128, y[i], 0, epi, in1=xi, off1=0, c1=256, h=h, aux=aux[i], **kx[i]) for i in range(2)]
        elif epi == EPI_GRU_Q:
            if SPLIT:
                kx = [dict(in0s=hs, in1s=xs, outs=ys[i]) for i in range(2)]
            descs = [cv[i].desc(hi, 0, 128, y[i], 64 * i, epi, in1=xi, off1=0, c1=256, h=h, z=z, **kx[i]) for i in range(2)]
        else:
            if SPLIT:
                kx = [dict(in0s=xs, outs=ys[i]) for i in range(2)]
            descs = [cv[i].desc(xi, 32 * i, cin, y[i], 0, epi, **kx[i]) for i in range(2)]
        lib.conv2d(descs, B, H8, W8, x)
        torch.cuda.synchronize()
        key = f"{name}@{B}x{H8}x{W8}"
        out[key] = torch.cat(y + aux, 1).cpu()
        roles[key] = lib.conv2d_roles(descs, B, H8, W8)
        if SPLIT:
            # columns a launch does not write are zero in both forms, so whole buffers compare
            twin_ok[key] = all(torch.equal(twin_of(y[i]), ys[i]) for i in range(2)) and \
                (epi != EPI_GRU_ZR or all(torch.equal(twin_of(aux[i]), auxs[i]) for i in range(2)))
torch.save({"out": out, "roles": roles, "twin_ok": twin_ok}, sys.argv[1])
