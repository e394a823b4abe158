#!/usr/bin/env python3
"""Micro-benchmark of pf_conv2d_wgrad at the update-block shapes (B=1 512x1024: 64x128 map, both branches
stacked as B=2).   python profiles/microbench_wgrad.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from prior_flow_amd import _lib

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lib = _lib.load()
dev = torch.device("cuda:0")
B, H, W = 2, 64, 128
N = B * H * W
for name, kh, kw, cin, cout in (("gru z|r 1x5", 1, 5, 384, 256), ("gru q 5x1", 5, 1, 384, 128), ("heads 3x3", 3, 3, 128, 256),
                                ("convc2 3x3", 3, 3, 256, 192), ("convc1 1x1", 1, 1, 324, 256)):
    x = torch.rand(N, cin, device=dev) * 2 - 1
    dy = torch.rand(N, cout, device=dev) * 2 - 1
    dw = torch.zeros((cout + 127) // 128 * 128, kh * kw, (cin + 31) // 32 * 32, device=dev)
    db = torch.zeros((cout + 127) // 128 * 128, device=dev)
    for _ in range(3):
        lib.conv2d_wgrad(x, 0, cin, dy, 0, cout, dw, db, kh, kw, B, H, W)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        lib.conv2d_wgrad(x, 0, cin, dy, 0, cout, dw, db, kh, kw, B, H, W)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / reps
    fl = 2.0 * N * cout * cin * kh * kw
    print(f"{name:14s} {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s algorithmic")
