#!/usr/bin/env python3
"""Single launches of pf_enc_stem at the forward's shapes (fnet: 4 images of 512x1024 -> fp32 rows + statistics; cnet: 2 images ->
split twin, ReLU), HIP-event time per launch.   python profiles/microbench_stem.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from prior_flow_amd import _lib
from prior_flow_amd.engine import pack_stem7x7, split_twin

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
lib = _lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
H, W = 512, 1024
w = (torch.rand(64, 3, 7, 7, device=dev) * 2 - 1) * 0.2
b = (torch.rand(64, device=dev) * 2 - 1) * 0.3
wp = pack_stem7x7(w)
for name, Bn in (("fnet", 4), ("cnet", 2)):
    img = torch.rand(Bn, 3, H, W, device=dev) * 2 - 1
    rows = Bn * (H // 2) * (W // 2)
    out = torch.empty(rows, 64, device=dev) if name == "fnet" else None
    tw = split_twin(rows, 64, dev) if name == "cnet" else None
    nblk = ((H // 2 + 7) // 8) * ((W // 2 + 31) // 32)
    part = torch.zeros(Bn * nblk * 64 * 2, dtype=torch.float64, device=dev) if name == "fnet" else None

    def go():
        lib.enc_stem(img, wp, b, out=out, out_split=tw, relu=name == "cnet", stats=part)
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            go()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3 / reps)
    mb = rows * 64 * 4 / 1e6
    print(f"{name}: median {sorted(ts)[2]:6.1f} us  min {min(ts):6.1f} us   ({mb:.0f} MB out)")
