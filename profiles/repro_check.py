#!/usr/bin/env python3
"""Race screen: the multi-stream forward (graph replay and eager) must be bitwise reproducible from run
to run and equal to the single-stream result.   python profiles/repro_check.py [runs]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from prior_flow_amd import det_state_dict, synthetic_pair
from prior_flow_amd.modules import state_dict_shapes
from prior_flow_amd.prior_raft import PriOr_RAFT

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
params = det_state_dict(state_dict_shapes())


def model(streams: bool, graph: bool):
    m = PriOr_RAFT(argparse.Namespace(mixed_precision=False, dropout=0.0))
    m.load_state_dict(params, strict=True)
    m = m.cuda().eval()
    m.use_streams, m.use_graph = streams, graph
    return m


bad = 0
for h, w, iters in ((512, 1024, 12), (256, 512, 6), (480, 960, 4)):
    i1, i2 = synthetic_pair(1, h, w, seed=11)
    i1, i2 = i1.cuda(), i2.cuda()
    with torch.no_grad():
        ref = model(False, False)(i1, i2, iters=iters, test_mode=True).clone()          # single stream, eager
        for streams, graph in ((True, True), (True, False)):
            m = model(streams, graph)
            diffs = 0
            for _ in range(runs):
                out = m(i1, i2, iters=iters, test_mode=True)
                diffs += int(not torch.equal(out, ref))
            print(f"{h}x{w} iters={iters} streams={streams} graph={graph}: {diffs} of {runs} runs differ from the single-stream result")
            bad += diffs
        m = model(True, False)
        pa, pb = m(i1, i2, iters=iters)
        pa2, pb2 = model(False, False)(i1, i2, iters=iters)
        d = sum(int(not torch.equal(x, y)) for x, y in zip(pa + pb, pa2 + pb2))
        print(f"{h}x{w} all {2 * iters} predictions, streams vs single stream: {d} differ")
        bad += d
print("RACE SCREEN:", "clean" if bad == 0 else f"{bad} mismatches")
sys.exit(1 if bad else 0)
