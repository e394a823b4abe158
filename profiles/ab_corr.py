#!/usr/bin/env python3
"""Same-process A/B of the corr + pyramid kernels (tile kernel vs role-split kernel), interleaved rounds:
   python profiles/ab_corr.py [rounds] [reps] [variant ...]      (MB_BATCH = pairs, MB_H8 / MB_W8 = map size)
A variant is `tile` or `rs` (-> PRIORFLOW_CORR_RS=0 / 1; timing-only ablations of the role-split kernel are compile-time builds:
profiles/ab_corr_libs.py).  Prints per-variant median / min HIP-event time per launch and the algorithmic HBM rate."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from prior_flow_amd import _lib

args = sys.argv[1:]
rounds = int(args[0]) if len(args) > 0 else 7
reps = int(args[1]) if len(args) > 1 else 10
PREDEF = {"tile": "RS=0", "rs": "RS=1"}
variants = []
for v in (args[2:] or ["tile", "rs"]):
    name, _, spec = v.partition(":")
    spec = spec or PREDEF[name]
    variants.append((name, dict(kv.split("=") for kv in spec.split(","))))
KEYS = ("RS",)
lib = _lib.load()
dev = torch.device("cuda:0")
B, H8, W8, C = int(os.environ.get("MB_BATCH", "1")), int(os.environ.get("MB_H8", "64")), int(os.environ.get("MB_W8", "128")), 256
N = H8 * W8
g = torch.Generator().manual_seed(0)
f = [((torch.rand(B * N, C, generator=g) * 2 - 1)).to(dev) for _ in range(2)]
fs = [lib.split_bf16(x, torch.empty(B * N, C // 32, 2, 32, dtype=torch.bfloat16, device=dev)) for x in f]
lv = [torch.empty(B * N, (H8 >> i) * (W8 >> i), device=dev) for i in range(4)]
mb = B * (4.0 * N * N * 85 / 64 + 2 * 4 * N * C) / 1e6
times = {name: [] for name, _ in variants}
for rnd in range(rounds + 1):
    for name, kn in variants:
        for k in KEYS:
            os.environ.pop("PRIORFLOW_CORR_" + k, None)
        for k, val in kn.items():
            os.environ["PRIORFLOW_CORR_" + k] = val
        lib.corr_pyramid_bf16x3(fs[0], fs[1], lv, B, H8, W8, C)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            lib.corr_pyramid_bf16x3(fs[0], fs[1], lv, B, H8, W8, C)
        e.record()
        torch.cuda.synchronize()
        if rnd:                     # round 0 is warm-up
            times[name].append(s.elapsed_time(e) * 1e3 / reps)
for name, t in times.items():
    med, mn = statistics.median(t), min(t)
    print(f"{name:>22}: median {med:6.1f} us  min {mn:6.1f} us per launch   {mb / med:.3f} TB/s algorithmic = "
          f"{mb / med / 8.0:.3f} of 8 TB/s   ({mb:.1f} MB, B={B}, {H8}x{W8})")
