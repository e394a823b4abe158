#!/usr/bin/env python3
"""Micro-benchmark of pf_corr_pyramid_bf16x3 at the 512x1024 problem size (B=1, N=8192, C=256):
   python profiles/microbench_corr.py [reps]
Prints HIP-event time per launch and the algorithmic HBM rate (373.3 MB written+read per launch)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from prior_flow_amd import _lib

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lib = _lib.load()
dev = torch.device("cuda:0")
B, H8, W8, C = int(os.environ.get("MB_BATCH", "1")), 64, 128, 256
N = H8 * W8
g = torch.Generator().manual_seed(0)
f = [((torch.rand(B * N, C, generator=g) * 2 - 1)).to(dev) for _ in range(2)]
fs = [lib.split_bf16(x, torch.empty(B * N, C // 32, 2, 32, dtype=torch.bfloat16, device=dev)) for x in f]
lv = [torch.empty(B * N, (H8 >> i) * (W8 >> i), device=dev) for i in range(4)]
for _ in range(3):
    lib.corr_pyramid_bf16x3(fs[0], fs[1], lv, B, H8, W8, C)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(reps):
    lib.corr_pyramid_bf16x3(fs[0], fs[1], lv, B, H8, W8, C)
e.record()
torch.cuda.synchronize()
us = s.elapsed_time(e) * 1e3 / reps
mb = B * (4.0 * N * N * 85 / 64 + 2 * 4 * N * C) / 1e6
print(f"corr+pyramid bf16x3: {us:.1f} us/launch  {mb / us:.3f} TB/s algorithmic ({mb:.1f} MB)  "
      f"{2.0 * B * N * N * C / us / 1e6:.1f} TFLOP/s algorithmic")
