#!/usr/bin/env python3
"""Micro-benchmark of pf_norm_act (the ResidualBlock tail relu(res + relu(y * s + t)), core/extractor.py:44-47) at fnet's layer-1
size (2 images of 256 x 512 pixels, 64 channels: 201 MB through the kernel) and layer-2 size:
   python profiles/microbench_norm_act.py [reps]        PF_LIB=<path> times another build of the library."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from prior_flow_amd import _lib

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
lib = _lib.PfLib(os.environ["PF_LIB"], optional=tuple(_lib._SIGNATURES)) if os.environ.get("PF_LIB") else _lib.load()
dev = torch.device("cuda:0")
for B, Np, C in ((2, 256 * 512, 64), (2, 128 * 256, 96), (2, 64 * 128, 128)):
    g = torch.Generator().manual_seed(0)
    y, res = (torch.randn(B * Np, C, generator=g).to(dev) for _ in range(2))
    s, t, rs, rt = ((torch.rand(B, C, generator=g) + 0.5).to(dev) for _ in range(4))
    out = torch.empty_like(y)
    spoil = torch.empty(512 << 20, dtype=torch.uint8, device=dev)          # flushes L2 / MALL between launches

    def run():
        lib.norm_act(y, s, t, out, B, Np, C, res=res, rs=rs, rt=rt, res_relu=True)

    run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        spoil.fill_(1)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    mb = 3 * B * Np * C * 4 / 1e6
    print(f"norm_act {B} x {Np} x {C}: median {ts[len(ts) // 2]:6.1f} us  ({mb / ts[len(ts) // 2]:.2f} TB/s over {mb:.0f} MB)   checksum {float(out.double().sum()):.3f}")
