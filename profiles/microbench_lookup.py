#!/usr/bin/env python3
"""Micro-benchmark of pf_dccl_lookup + pf_dccl_combine at the 512x1024 problem size (one branch, N=8192):
   python profiles/microbench_lookup.py [reps]"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from prior_flow_amd import _lib
from prior_flow_amd.engine import rotation_x

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
# PF_LIB=<path> times another build of the per-element kernels (e.g. the ablation variants of
# round-1 ablation builds, which held only pf_elem_kernels.hip)
lib = _lib.PfLib(os.environ["PF_LIB"], optional=tuple(_lib._SIGNATURES)) if os.environ.get("PF_LIB") else _lib.load()
dev = torch.device("cuda:0")
B, H8, W8 = 1, 64, 128
N = H8 * W8
g = torch.Generator().manual_seed(0)
pyr_a = [torch.rand(B * N, (H8 >> i) * (W8 >> i), generator=g).to(dev) for i in range(4)]
pyr_b = [torch.rand(B * N, (H8 >> i) * (W8 >> i), generator=g).to(dev) for i in range(4)]
xs = torch.arange(W8).view(1, 1, 1, W8).expand(B, 1, H8, W8).float()
ys = torch.arange(H8).view(1, 1, H8, 1).expand(B, 1, H8, W8).float()
coords = (torch.cat([xs, ys], 1) + (torch.rand(B, 2, H8, W8, generator=g) * 12 - 6)).contiguous().to(dev)
g8 = torch.empty(2, H8, W8, device=dev)
lib.sample_grid(g8, rotation_x(math.pi / 2))
own, raw, out = (torch.empty(B * N, 324, device=dev) for _ in range(3))


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


g8_il = g8.reshape(2, -1).t().contiguous()
print(f"lookup : {timed(lambda: lib.dccl_lookup(coords, pyr_a, pyr_b, g8, own, raw)):.1f} us/launch")
print(f"lookup (interleaved grid): {timed(lambda: lib.dccl_lookup(coords, pyr_a, pyr_b, g8, own, raw, g8_il)):.1f} us/launch")
print(f"combine: {timed(lambda: lib.dccl_combine(own, raw, g8, out, B, H8, W8)):.1f} us/launch")
