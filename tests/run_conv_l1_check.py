"""Checks pf_enc_conv64_kernel (weights-stationary 3x3 64 -> 64) against the halo kernel it replaces, bit for bit, outputs and fused
InstanceNorm partials, with and without the input affine; run as a child process per PRIORFLOW_ENC_CONV64 value (read once)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from prior_flow_amd import _lib
from prior_flow_amd._lib import EPI_LINEAR, EPI_RELU, EPI_RELU_RES, PREC_BF16X3
from prior_flow_amd.engine import Conv, pack_mfma, split_twin


def run(shape, out_path):
    lib = _lib.load()
    dev = torch.device("cuda:0")
    Bn, H, W = shape
    g = torch.Generator().manual_seed(H * W + Bn)
    x = ((torch.rand(Bn * H * W, 64, generator=g) * 2 - 1) * 1.5).to(dev)
    w = ((torch.rand(64, 64, 3, 3, generator=g) * 2 - 1) * 0.06).to(dev)
    b = ((torch.rand(64, generator=g) * 2 - 1) * 0.2).to(dev)
    sc = (torch.rand(Bn, 64, generator=g) + 0.5).to(dev)
    sh = ((torch.rand(Bn, 64, generator=g) - 0.5)).to(dev)
    cv = Conv(*pack_mfma(w, b), 3, 3, 64, 64, PREC_BF16X3)
    res = []
    hres = (torch.rand(Bn * H * W, 64, generator=g) * 2 - 0.7).to(dev)          # the residual operand of the RELU_RES tail
    for affine, relu, stats, epi in ((False, False, True, EPI_LINEAR), (True, True, True, EPI_LINEAR), (True, False, False, EPI_RELU),
                                     (False, False, False, EPI_RELU_RES)):
        out = torch.full((Bn * H * W, 64), float("nan"), device=dev)
        kw = {}
        if epi == EPI_RELU_RES:
            kw.update(h=hres)
        if affine:
            kw.update(in_scale=sc, in_shift=sh, in_relu=relu)
        d = cv.desc(x, 0, 64, out, 0, epi, **kw)
        tile = lib.conv2d_tile([d], Bn, H, W)
        nblk = lib.conv2d_stats_blocks([d], Bn, H, W)
        scale = shift = torch.zeros(1)
        if stats:
            part = torch.full((Bn * nblk * 64 * 2,), float("nan"), dtype=torch.float64, device=dev)
            d = cv.desc(x, 0, 64, out, 0, epi, stats=part, **kw)
            lib.conv2d([d], Bn, H, W, x)
            scale, shift = torch.empty(Bn, 64, device=dev), torch.empty(Bn, 64, device=dev)
            lib.channel_stats_final(part, Bn, H * W, 64, nblk, scale, shift)
        else:
            lib.conv2d([d], Bn, H, W, x)
        torch.cuda.synchronize()
        res += [out.cpu(), scale.cpu(), shift.cpu(), torch.tensor([tile])]
    torch.save(res, out_path)


if __name__ == "__main__":
    shape = tuple(int(v) for v in sys.argv[1].split("x"))
    run(shape, sys.argv[2])
