#!/usr/bin/env python3
"""Single launches of the update blocks' convolutions at the 512x1024 problem size (B=1, 64x128 map, both branches as two
groups): the all-DMA kernel on split twins against the role-specialised kernel on the same fp32 inputs, interleaved in one
process.
   python profiles/microbench_conv_dma.py [reps] [zr|q|c2|out|fh1 ...]
With a diagnostic build (hipcc -DPF_DMA_STAMPS ... pf_conv_dma.hip, PRIORFLOW_LIB=<that .so>) it also prints the s_memtime
stamps of workgroup 0: per K-step the cycles an MFMA wave spends issuing (barrier exit -> next barrier arrival) and waiting in
the barrier, the loader waves' DMA issue time and vmcnt wait, and the shader clock (s_memtime / s_memrealtime)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from prior_flow_amd import _lib
from prior_flow_amd._lib import EPI_GRU_Q, EPI_GRU_ZR, EPI_RELU, PREC_BF16X3
from prior_flow_amd.engine import Conv, pack_mfma, split_twin

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
which = sys.argv[2:] or ["zr", "q", "c2", "out", "fh1"]
lib = _lib.load()
dev = torch.device("cuda:0")
B, H8, W8 = int(os.environ.get("MB_BATCH", "1")), 64, 128
N = B * H8 * W8
g = torch.Generator(device="cpu").manual_seed(0)


def rnd(*shape, s=1.0):
    return (torch.rand(*shape, generator=g) * 2 - 1).mul_(s).to(dev)


def conv(cin, cout, kh, kw):
    w = rnd(cout, cin, kh, kw, s=(1.0 / (cin * kh * kw)) ** 0.5)
    wp, bp = pack_mfma(w, rnd(cout, s=0.1))
    return Conv(wp, bp, kh, kw, cin, cout, PREC_BF16X3)


def twin_of(t):
    return lib.split_bf16(t, split_twin(t.shape[0], t.shape[1], dev))


x = [rnd(N, 288) for _ in range(2)]
h = [rnd(N, 128) for _ in range(2)]
z = [torch.rand(N, 128, generator=g).to(dev) for _ in range(2)]
xs, hs = [twin_of(t) for t in x], [twin_of(t) for t in h]
y = [torch.zeros(N, 256, device=dev) for _ in range(2)]
aux = [torch.zeros(N, 128, device=dev) for _ in range(2)]
ys = [split_twin(N, 256, dev) for _ in range(2)]
auxs = [split_twin(N, 128, dev) for _ in range(2)]
SHAPES = {"zr": (384, 256, 1, 5, EPI_GRU_ZR), "q": (384, 128, 5, 1, EPI_GRU_Q), "c2": (256, 192, 3, 3, EPI_RELU),
          "out": (272, 126, 3, 3, EPI_RELU), "fh1": (128, 256, 3, 3, EPI_RELU)}


def descs(name, split):
    cin, cout, kh, kw, epi = SHAPES[name]
    cv = CONVS[name]
    out = []
    for i in range(2):
        if epi == EPI_GRU_ZR:
            kw_ = dict(in0s=hs[i], in1s=xs[i], auxs=auxs[i]) if split else {}
            out.append(cv[i].desc(None if split else h[i], 0, 128, y[i], 0, epi, in1=None if split else x[i], off1=0, c1=256,
                                  h=h[i], aux=None if split else aux[i], **kw_))
        elif epi == EPI_GRU_Q:
            kw_ = dict(in0s=hs[i], in1s=xs[i], outs=ys[i]) if split else {}
            out.append(cv[i].desc(None if split else h[i], 0, 128, y[i], 0, epi, in1=None if split else x[i], off1=0, c1=256,
                                  h=h[i], z=z[i], **kw_))
        elif name == "fh1":        # fp32 output only (the FlowHead tail reads fp32)
            out.append(cv[i].desc(None if split else h[i], 0, cin, y[i], 0, epi, in0s=hs[i] if split else None))
        else:
            out.append(cv[i].desc(None if split else x[i], 0, cin, None if split else y[i], 0, epi,
                                  in0s=xs[i] if split else None, outs=ys[i] if split else None))
    return out


if "l1" in which:      # encoder layer-1 conv: 3x3 64 -> 64 at 1/2 resolution, MB_BATCH images (fnet: 4, cnet: 2), one group
    which = [w for w in which if w != "l1"]
    Bl, Hl, Wl = int(os.environ.get("MB_BATCH", "4")), 256, 512
    Nl = Bl * Hl * Wl
    xl = rnd(Nl, 64); yl = torch.zeros(Nl, 64, device=dev)
    xls, yls = twin_of(xl), split_twin(Nl, 64, dev)
    cv = conv(64, 64, 3, 3)
    forms = {"dma": [cv.desc(None, 0, 64, yl, 0, EPI_RELU, in0s=xls, outs=yls)], "halo": [cv.desc(xl, 0, 64, yl, 0, EPI_RELU)]}
    flops = 2.0 * Nl * 64 * 9 * 64
    for k, d in forms.items():
        for _ in range(3):
            lib.conv2d(d, Bl, Hl, Wl, xl)
    torch.cuda.synchronize()
    tm = {k: [] for k in forms}
    for _ in range(5):
        for k, d in forms.items():
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record()
            for _ in range(reps):
                lib.conv2d(d, Bl, Hl, Wl, xl)
            e_.record()
            torch.cuda.synchronize()
            tm[k].append(s_.elapsed_time(e_) * 1e3 / reps)
    for k in forms:
        t = sorted(tm[k])
        print(f"l1 x{Bl} {k:4s} tile {lib.conv2d_tile(forms[k], Bl, Hl, Wl)} roles {lib.conv2d_roles(forms[k], Bl, Hl, Wl):2d}  median {t[2]:6.1f} us  "
              f"{flops / t[2] / 1e6:6.1f} TFLOP/s algorithmic  (x3 = {3 * flops / t[2] / 1e6 / 2500:.3f} of the bf16 pipe)")
CONVS = {n: [conv(*SHAPES[n][:4]) for _ in range(2)] for n in which}
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
for name in which:
    cin, cout, kh, kw, epi = SHAPES[name]
    flops = 2.0 * 2 * N * cout * kh * kw * cin
    forms = {"dma": descs(name, True), "ws": descs(name, False)}
    times = {k: [] for k in forms}
    for k, d in forms.items():
        for _ in range(3):
            lib.conv2d(d, B, H8, W8, x[0])
    torch.cuda.synchronize()
    for rnd_i in range(5):                      # interleaved rounds in one process
        for k, d in forms.items():
            s, e = ev(), ev()
            s.record()
            for _ in range(reps):
                lib.conv2d(d, B, H8, W8, x[0])
            e.record()
            torch.cuda.synchronize()
            times[k].append(s.elapsed_time(e) * 1e3 / reps)
    for k in forms:
        t = sorted(times[k])
        print(f"{name:4s} {k:3s} roles {lib.conv2d_roles(forms[k], B, H8, W8):2d}  median {t[2]:6.1f} us  min {t[0]:6.1f} us  "
              f"{flops / t[2] / 1e6:6.1f} TFLOP/s algorithmic  (x3 = {3 * flops / t[2] / 1e6 / 2500:.3f} of the bf16 pipe)")
    if hasattr(lib._dll, "pf_conv_dma_read_stamps"):
        lib.conv2d(forms["dma"], B, H8, W8, x[0])
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * (8 * 64 * 4 + 4))()
        assert lib._dll.pf_conv_dma_read_stamps(buf) == 0
        a = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
        t, glob = a[:8 * 64 * 4].reshape(8, 64, 4), a[8 * 64 * 4:]
        nsteps = min(60, kh * kw * ((cin + 31) // 32))
        clk = (glob[2] - glob[0]) / max(1, (glob[3] - glob[1])) * 100.0
        print(f"  shader clock over the K loop: {clk:.0f} MHz; loop {glob[2] - glob[0]} cycles = {(glob[3] - glob[1]) / 100.0:.1f} us for "
              f"{kh * kw * ((cin + 31) // 32)} steps")
        print("  step | M waves: issue (barrier exit -> next arrival), wait in barrier | L waves: DMA issue, vmcnt wait, barrier wait | step length")
        for s_ in range(0, nsteps - 1):
            mi = t[:4, s_ + 1, 0] - t[:4, s_, 1]
            mw = t[:4, s_ + 1, 1] - t[:4, s_ + 1, 0]
            li = t[4:, s_, 2] - t[4:, s_, 1]
            lw = t[4:, s_, 3] - t[4:, s_, 2]
            lb = t[4:, s_ + 1, 1] - t[4:, s_, 3]
            print(f"  {s_:3d} | " + " ".join(f"{int(v):5d}" for v in mi) + " ; " + " ".join(f"{int(v):4d}" for v in mw) + " | "
                  + " ".join(f"{int(v):4d}" for v in li) + " ; " + " ".join(f"{int(v):4d}" for v in lw) + " ; "
                  + " ".join(f"{int(v):4d}" for v in lb) + f" | {int(t[0, s_ + 1, 1] - t[0, s_, 1]):5d}")
        total = kh * kw * ((cin + 31) // 32)
        if total < 64:
            print(f"  epilogue (M waves, K-loop end -> stores retired): " + " ".join(f"{int(t[w, total, 1] - t[w, total, 0]):6d}" for w in range(4)))
