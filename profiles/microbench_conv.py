#!/usr/bin/env python3
"""Micro-benchmark of single pf_conv2d launches at the 512x1024 problem size (B=1, 64x128 map):
   python profiles/microbench_conv.py [reps] [which]
which: l1 / l2 / l3 (encoder 3x3 convs, MB_BATCH images), c1 (1x1 324->256 x2), zr (grouped 1x5 384->256 GRU gates), q (1x5 384->128), c2 (3x3 256->128|192), fh1 (3x3 128->256 x3)
Prints HIP-event time per launch and algorithmic TFLOP/s; used under rocprofv3 --pmc."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from prior_flow_amd import _lib
from prior_flow_amd._lib import EPI_GRU_Q, EPI_GRU_ZR, EPI_RELU, PREC_BF16X3, PREC_F32
from prior_flow_amd.engine import Conv, pack_mfma

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
which = sys.argv[2] if len(sys.argv) > 2 else "zr"
prec = PREC_F32 if os.environ.get("PRIORFLOW_PRECISION", "bf16x3") == "fp32" else PREC_BF16X3
lib = _lib.load()
dev = torch.device("cuda:0")
H8, W8 = 64, 128
BATCH = int(os.environ.get("MB_BATCH", "1"))      # stacked along H for the micro-benchmark (same kernels, longer dispatch)
H8 *= BATCH
N = H8 * W8
g = torch.Generator(device="cpu").manual_seed(0)


def rnd(*shape, s=1.0):
    return (torch.rand(*shape, generator=g) * 2 - 1).mul_(s).to(dev)


def conv(cin, cout, kh, kw):
    w = rnd(cout, cin, kh, kw, s=(1.0 / (cin * kh * kw)) ** 0.5)
    b = rnd(cout, s=0.1)
    wp, bp = pack_mfma(w, b)
    return Conv(wp, bp, kh, kw, cin, cout, prec)


def time_launches(fn, reps):
    """us per launch: back to back (default), or MB_COLD=1: each launch behind a 512 MB fill (operands from HBM, as in a forward
    where the previous kernel wrote them), median of the per-launch event times."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if os.environ.get("MB_COLD", "0") != "1":
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) * 1e3 / reps
    spoil = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    ts = []
    for _ in range(reps):
        spoil.fill_(1)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    return sorted(ts)[len(ts) // 2]


if which in ("l2", "l3"):  # encoder layer-2 / layer-3 convs: 3x3 96 -> 96 at 1/4 (128 x 256), 128 -> 128 at 1/8 (64 x 128), MB_BATCH images;
    # MB_AFFINE=1: with the folded input norm + ReLU and the fused statistics (fnet's form: the symmetric halo kernel)
    from prior_flow_amd._lib import EPI_LINEAR
    H8, W8, C = (128, 256, 96) if which == "l2" else (64, 128, 128)
    N = H8 * W8
    xin = rnd(BATCH * N, C)
    yout = torch.empty(BATCH * N, C, device=dev)
    cvx = conv(C, C, 3, 3)
    kw = {}
    if os.environ.get("MB_AFFINE", "0") == "1":
        sc, sh = (torch.rand(BATCH, C, generator=g) + 0.5).to(dev), (torch.rand(BATCH, C, generator=g) - 0.5).to(dev)
        kw = dict(in_scale=sc, in_shift=sh, in_relu=True)
    descs = [cvx.desc(xin, 0, C, yout, 0, EPI_LINEAR, **kw)]
    if kw:
        nblk = lib.conv2d_stats_blocks(descs, BATCH, H8, W8)
        part = torch.empty(BATCH * nblk * C * 2, dtype=torch.float64, device=dev)
        descs = [cvx.desc(xin, 0, C, yout, 0, EPI_LINEAR, stats=part, **kw)]
    flops = 2.0 * BATCH * N * C * 9 * C
    us = time_launches(lambda: lib.conv2d(descs, BATCH, H8, W8, xin), reps)
    print(f"{which} x{BATCH}{' affine+stats' if kw else ''}: tile {lib.conv2d_tile(descs, BATCH, H8, W8)} roles {lib.conv2d_roles(descs, BATCH, H8, W8)}  "
          f"{us:.1f} us/launch  {flops / us / 1e6:.1f} TFLOP/s algorithmic")
    sys.exit(0)

if which == "l1":        # encoder layer-1 conv: 3x3 64 -> 64 at 1/2 resolution (256 x 512), MB_BATCH images
    H8, W8 = 256, 512
    N = H8 * W8
    xin = rnd(BATCH * N, 64)
    yout = torch.empty(BATCH * N, 64, device=dev)
    cv1 = conv(64, 64, 3, 3)
    descs = [cv1.desc(xin, 0, 64, yout, 0, EPI_RELU)]
    if os.environ.get("MB_AFFINE", "0") == "1":        # fnet's form: folded input norm + ReLU, fused statistics
        from prior_flow_amd._lib import EPI_LINEAR
        sc, sh = (torch.rand(BATCH, 64, generator=g) + 0.5).to(dev), (torch.rand(BATCH, 64, generator=g) - 0.5).to(dev)
        kw = dict(in_scale=sc, in_shift=sh, in_relu=True)
        nblk = lib.conv2d_stats_blocks([cv1.desc(xin, 0, 64, yout, 0, EPI_LINEAR, **kw)], BATCH, H8, W8)
        part = torch.empty(BATCH * nblk * 64 * 2, dtype=torch.float64, device=dev)
        descs = [cv1.desc(xin, 0, 64, yout, 0, EPI_LINEAR, stats=part, **kw)]
    flops = 2.0 * BATCH * N * 64 * 9 * 64
    us = time_launches(lambda: lib.conv2d(descs, BATCH, H8, W8, xin), reps)
    print(f"l1 x{BATCH}{' affine+stats' if os.environ.get('MB_AFFINE', '0') == '1' else ''}{' cold' if os.environ.get('MB_COLD', '0') == '1' else ''}: tile {lib.conv2d_tile(descs, BATCH, H8, W8)}  {us:.1f} us/launch  {flops / us / 1e6:.1f} TFLOP/s algorithmic  "
          f"{BATCH * N * 64 * 8 / us / 1e6:.2f} TB/s in+out")
    sys.exit(0)
net = [rnd(N, 128) for _ in range(2)]
x = [rnd(N, 256) for _ in range(2)]
z = [torch.empty(N, 128, device=dev) for _ in range(2)]
rh = [torch.empty(N, 128, device=dev) for _ in range(2)]
out = [torch.empty(N, 256, device=dev) for _ in range(3)]
if which == "zr":
    cv = [conv(384, 256, 1, 5) for _ in range(2)]
    descs = [cv[i].desc(net[i], 0, 128, z[i], 0, EPI_GRU_ZR, in1=x[i], off1=0, c1=256, h=net[i], aux=rh[i]) for i in range(2)]
    flops = 2 * 2.0 * N * 256 * 5 * 384
elif which == "q":
    cv = [conv(384, 128, 1, 5) for _ in range(2)]
    descs = [cv[i].desc(rh[i], 0, 128, out[i], 0, EPI_GRU_Q, in1=x[i], off1=0, c1=256, h=net[i], z=z[i]) for i in range(2)]
    flops = 2 * 2.0 * N * 128 * 5 * 384
elif which == "c1":      # convc1: 1x1 324 -> 256 on both branches' correlation features
    corr = [rnd(N, 324) for _ in range(2)]
    cv = [conv(324, 256, 1, 1) for _ in range(2)]
    descs = [cv[i].desc(corr[i], 0, 324, out[i], 0, EPI_RELU) for i in range(2)]
    flops = 2 * 2.0 * N * 256 * 324
elif which == "c2":
    cv = [conv(256, 128, 3, 3), conv(256, 192, 3, 3)]
    descs = [cv[i].desc(x[i], 0, 256, out[i], 0, EPI_RELU) for i in range(2)]
    flops = 2.0 * N * (128 + 192) * 9 * 256
else:
    cv = [conv(128, 256, 3, 3) for _ in range(3)]
    descs = [cv[i].desc(net[i % 2], 0, 128, out[i], 0, EPI_RELU) for i in range(3)]
    flops = 3 * 2.0 * N * 256 * 9 * 128
# MB_WSETS=n: rotate through n copies of the weights (as in the forward, where ~40 MB of other layers' weights pass
# through the 4 MB L2 between two uses of a layer: the weight stream then comes from the Infinity Cache, not from L2)
WSETS = int(os.environ.get("MB_WSETS", "1"))
variants = [descs]
if WSETS > 1:
    import ctypes
    for k in range(1, WSETS):
        ds = []
        for d_ in descs:
            d2 = type(d_)()
            ctypes.memmove(ctypes.byref(d2), ctypes.byref(d_), ctypes.sizeof(d_))
            w2 = torch.empty_like(d_._keep[9]).copy_(d_._keep[9])
            d2.weight = w2.data_ptr()
            d2._keep = d_._keep + (w2,)
            ds.append(d2)
        variants.append(ds)
for k in range(3):
    lib.conv2d(variants[k % WSETS], 1, H8, W8, x[0])
torch.cuda.synchronize()
if os.environ.get("MB_EACH", "0") == "1":      # per-launch times (looking for outliers)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for k in range(reps):
        evs[k][0].record(); lib.conv2d(variants[k % WSETS], 1, H8, W8, x[0]); evs[k][1].record()
    torch.cuda.synchronize()
    print(which, "per-launch us:", " ".join(f"{a.elapsed_time(b) * 1e3:.0f}" for a, b in evs))
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for k in range(reps):
    lib.conv2d(variants[k % WSETS], 1, H8, W8, x[0])
e.record()
torch.cuda.synchronize()
us = s.elapsed_time(e) * 1e3 / reps
print(f"{which}: tile {lib.conv2d_tile(descs, 1, H8, W8)} roles {lib.conv2d_roles(descs, 1, H8, W8)}  {us:.1f} us/launch  {flops / us / 1e6:.1f} TFLOP/s algorithmic "
      f"({'fp32' if prec == PREC_F32 else 'bf16x3'})")
