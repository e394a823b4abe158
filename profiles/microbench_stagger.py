#!/usr/bin/env python3
"""Round 6: do two half-chip chains of the update blocks' convolutions run faster when they are OUT of step?
One branch's launch = one group with pf_conv_desc.co_groups = 1 (128 work items at 64x128: half the CUs).  Cases, interleaved:
  grouped   both branches as two groups of one launch (256 items), N launches back to back on one stream   [the shipped form]
  lockstep  the two branches as one-group launches on two streams, started together
  stagger   the same, the second stream started `delay` us later (a spin kernel in front of its chain)
Reported: us per (A + B) launch pair = (end of the later stream - start) / N, the stagger's delay subtracted.
   python profiles/microbench_stagger.py [N] [zr|q|c2|out|fh1 ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from prior_flow_amd import _lib
from prior_flow_amd._lib import EPI_GRU_Q, EPI_GRU_ZR, EPI_RELU, PREC_BF16X3
from prior_flow_amd.engine import Conv, pack_mfma, split_twin

N_L = int(sys.argv[1]) if len(sys.argv) > 1 else 40
which = sys.argv[2:] or ["zr", "q", "out", "fh1"]
lib = _lib.load()
dev = torch.device("cuda:0")
B, H8, W8 = 1, 64, 128
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
ys = [split_twin(N, 256, dev) for _ in range(2)]
auxs = [split_twin(N, 128, dev) for _ in range(2)]
# the per-iteration shapes with the hoisted context: K = [h | motion] = 256 for the GRU convolutions
SHAPES = {"zr": (256, 256, 1, 5, EPI_GRU_ZR), "q": (256, 128, 1, 5, EPI_GRU_Q), "c2": (256, 192, 3, 3, EPI_RELU),
          "out": (272, 126, 3, 3, EPI_RELU), "fh1": (128, 256, 3, 3, EPI_RELU)}


def desc(name, i, cv):
    cin, cout, kh, kw, epi = SHAPES[name]
    if epi == EPI_GRU_ZR:
        return cv.desc(None, 0, 128, y[i], 0, epi, off1=0, c1=128, h=h[i], in0s=hs[i], in1s=xs[i], auxs=auxs[i])
    if epi == EPI_GRU_Q:
        return cv.desc(None, 0, 128, y[i], 0, epi, off1=0, c1=128, h=h[i], z=z[i], in0s=hs[i], in1s=xs[i], outs=ys[i])
    if name == "fh1":
        return cv.desc(None, 0, cin, y[i], 0, epi, in0s=hs[i])
    return cv.desc(None, 0, cin, None, 0, epi, in0s=xs[i], outs=ys[i])


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
for name in which:
    cvs = [conv(*SHAPES[name][:4]) for _ in range(2)]
    both = [desc(name, i, cvs[i]) for i in range(2)]
    single = [desc(name, i, cvs[i]) for i in range(2)]
    for d in single:
        d.co_groups = 1

    def grouped():
        a, b = ev(), ev()
        with torch.cuda.stream(s1):
            a.record()
            for _ in range(N_L):
                lib.conv2d(both, B, H8, W8, x[0])
            b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e3 / N_L

    def two(delay_cycles):
        a, m, b1, b2 = ev(), ev(), ev(), ev()
        go = torch.cuda.Event()
        with torch.cuda.stream(s1):
            torch.cuda._sleep(2000000)          # park both queues until the host has enqueued everything
            go.record()
            a.record()
        s2.wait_event(go)
        with torch.cuda.stream(s2):
            if delay_cycles:
                torch.cuda._sleep(delay_cycles)
            m.record()
            for _ in range(N_L):
                lib.conv2d([single[1]], B, H8, W8, x[0])
            b2.record()
        with torch.cuda.stream(s1):
            for _ in range(N_L):
                lib.conv2d([single[0]], B, H8, W8, x[0])
            b1.record()
        torch.cuda.synchronize()
        d_us = a.elapsed_time(m) * 1e3
        return max(a.elapsed_time(b1), a.elapsed_time(b2)) * 1e3, d_us, a.elapsed_time(b1) * 1e3, m.elapsed_time(b2) * 1e3

    for _ in range(2):
        grouped(); two(0)
    res = {"grouped": [], "lockstep": [], "stagger10": [], "stagger20": [], "stagger30": []}
    for _ in range(5):
        res["grouped"].append(grouped())
        res["lockstep"].append(two(0))
        res["stagger10"].append(two(21000))      # ~2.1 GHz shader clock: 10 / 20 / 30 us
        res["stagger20"].append(two(42000))
        res["stagger30"].append(two(63000))
    tg = sorted(res["grouped"])[2]
    print(f"{name:4s} roles {lib.conv2d_roles(both, B, H8, W8)} / {lib.conv2d_roles([single[0]], B, H8, W8)}  grouped {tg:6.1f} us per launch pair")
    for k in ("lockstep", "stagger10", "stagger20", "stagger30"):
        r = sorted(res[k], key=lambda t: t[0])[2]
        print(f"     {k:10s} whole {r[0]:8.1f} us, delay {r[1]:5.1f} us: stream 1 {r[2] / N_L:6.1f} us / launch, stream 2 {r[3] / N_L:6.1f} us / launch")
