#!/usr/bin/env python3
"""Forward time of the drop-in module at other sizes / iteration counts (not bench lines):
   python profiles/time_sizes.py 480x960:12 640x1280:32 512x1024:12"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from prior_flow_amd import det_state_dict, synthetic_pair
from prior_flow_amd.modules import state_dict_shapes
from prior_flow_amd.prior_raft import PriOr_RAFT

m = PriOr_RAFT(argparse.Namespace(mixed_precision=False, dropout=0.0))
m.load_state_dict(det_state_dict(state_dict_shapes()), strict=True)
m = m.cuda().eval()
for spec in sys.argv[1:] or ["512x1024:12"]:
    size, iters = spec.split(":")
    h, w = (int(v) for v in size.split("x"))
    i1, i2 = synthetic_pair(1, h, w, seed=1234)
    i1, i2 = i1.cuda(), i2.cuda()
    with torch.no_grad():
        for _ in range(3):
            m(i1, i2, iters=int(iters), test_mode=True)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            m(i1, i2, iters=int(iters), test_mode=True)
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t) * 100
    print(f"{h}x{w} iters={iters}: {ms:.2f} ms/pair  ({1e3 / ms:.1f} pairs/s)")
