#!/usr/bin/env python3
"""Micro-benchmark of the training step's per-iteration scatter kernels at the training crop 384x512 (N = 48 x 64 queries) or any
H8 W8: pf_dccl_lookup_bwd (the lookup's gradient into the two pyramids' gradients) and pf_upsample_flow_bwd (the convex
upsampling's backward).      python profiles/microbench_train_elem.py [reps] [H8 W8]
PF_LIB=<path> times another build of the library."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from prior_flow_amd import _lib
from prior_flow_amd.engine import rotation_x

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
H8, W8 = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (48, 64)
lib = _lib.PfLib(os.environ["PF_LIB"], optional=tuple(_lib._SIGNATURES)) if os.environ.get("PF_LIB") else _lib.load()
dev = torch.device("cuda:0")
B, N = 1, H8 * W8
g = torch.Generator().manual_seed(0)
xs = torch.arange(W8).view(1, 1, 1, W8).expand(B, 1, H8, W8).float()
ys = torch.arange(H8).view(1, 1, H8, 1).expand(B, 1, H8, W8).float()
coords = (torch.cat([xs, ys], 1) + (torch.rand(B, 2, H8, W8, generator=g) * 12 - 6)).contiguous().to(dev)
g8 = torch.empty(2, H8, W8, device=dev)
lib.sample_grid(g8, rotation_x(math.pi / 2))
d_own = (torch.rand(B * N, 324, generator=g) - 0.5).to(dev)
d_raw = (torch.rand(B * N, 324, generator=g) - 0.5).to(dev)
g_own = [torch.zeros(B * N, (H8 >> i) * (W8 >> i), device=dev) for i in range(4)]
g_oth = [torch.zeros(B * N, (H8 >> i) * (W8 >> i), device=dev) for i in range(4)]


mask = (torch.rand(B * N, 576, generator=g) * 4 - 2).to(dev)
g_up = (torch.rand(B, 2, 8 * H8, 8 * W8, generator=g) - 0.5).to(dev)
d_mask = torch.empty(B * N, 576, device=dev)
d_flow = torch.zeros(B, 2, H8, W8, device=dev)


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


t = timed(lambda: lib.dccl_lookup_bwd(coords, g8, d_own, d_raw, g_own, g_oth))
print(f"lookup_bwd   {H8}x{W8}: {t:6.1f} us/launch   checksum {sum(float(x.double().sum()) for x in g_own + g_oth) / (reps + 3):.6f}")
t = timed(lambda: lib.upsample_flow_bwd(coords, mask, g_up, d_mask, d_flow))
print(f"upsample_bwd {H8}x{W8}: {t:6.1f} us/launch   checksum {float(d_flow.double().sum()) / (reps + 3):.6f} {float(d_mask.double().abs().sum()):.6f}")
