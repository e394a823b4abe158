#!/usr/bin/env python3
"""Round 6: what is the phase between branch A's and branch B's chains worth?  Branch B's recurrence does not depend on branch A
(core/prior_raft.py:176-190: update_block sees net_B, inp_B, corr_B, flow_B only), so B may run any distance ahead of A and the
two chains need not be in the same phase of an iteration.  The head of a chain (lookup, combine + 1x1, convc2) is gather /
memory bound, its tail (motion-encoder out conv, SepConvGRU, FlowHead stem) MFMA bound.  On the real workspace of a forward:
  alone      one chain by itself on the chip (one-group launches, co_groups = 1), head and tail separately
  in phase   A: head tail head tail ...   beside   B: head tail head tail ...
  anti phase A: head tail head tail ...   beside   B: tail head tail head ...
us per iteration of each stream, N iterations back to back, both streams released together.
   python profiles/microbench_phase.py [N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from prior_flow_amd import synthetic_pair
from prior_flow_amd._lib import EPI_RELU
from prior_flow_amd.engine import Engine

N_IT = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda:0")
bench.torch = torch
model, params = bench.build_model(dev)
model.use_graph = False
i1, i2 = synthetic_pair(1, bench.H, bench.W, seed=1234)
i1, i2 = i1.to(dev), i2.to(dev)
with torch.no_grad():
    model(i1, i2, iters=12, test_mode=True)
torch.cuda.synchronize()
ws = next(iter(model._ws.values()))
P = model._weights()
lib = model._lib()
eng = Engine(lib, None)
B, H8, W8 = ws.B, ws.H8, ws.W8
like = ws.x_a


def conv(d):
    d.co_groups = 1
    lib.conv2d([d], B, H8, W8, like)


def head(t):
    if t == "a":
        lib.dccl_lookup(ws.c1a, ws.pyr_a, ws.pyr_b, ws.g_b2a_8, ws.own, ws.raw, ws.g_b2a_8_il)
        lib.dccl_combine_conv1x1([(ws.own, ws.raw, ws.g_b2a_8, P["a.c1"], None, 0, ws.c1_a_s)], B, H8, W8)
        conv(P["a.c2"].desc(None, 0, 256, None, 0, EPI_RELU, in0s=ws.c1_a_s, outs=ws.cat_a_s))
    else:
        lib.dccl_lookup(ws.c1b, ws.pyr_b, ws.pyr_a, ws.g_a2b_8, ws.own_b, ws.raw_b, ws.g_a2b_8_il)
        lib.dccl_combine_conv1x1([(ws.own_b, ws.raw_b, ws.g_a2b_8, P["b.c1"], None, 0, ws.c1_b_s)], B, H8, W8)
        conv(P["b.c2"].desc(None, 0, 256, None, 0, EPI_RELU, in0s=ws.c1_b_s, outs=ws.cat_b_s))


def tail(t):        # without the FlowHead's second conv: coords1 stays where the forward left it
    if t == "a":
        conv(P["a.out"].desc(None, 0, 272, None, 128, EPI_RELU, in0s=ws.cat_a_s, outs=ws.x_a_s))
        eng._gru_branch(ws, P, "a", 0, conv)
        conv(P["a.fh1"].desc(None, 0, 128, ws.fh_a, 0, EPI_RELU, in0s=ws.net_a_s[0]))
    else:
        conv(P["b.out"].desc(None, 0, 272, None, 128, EPI_RELU, in0s=ws.cat_b_s, outs=ws.x_b_s))
        eng._gru_branch(ws, P, "b", 0, conv)
        conv(P["b.fh1"].desc(None, 0, 128, ws.fh_b, 0, EPI_RELU, in0s=ws.net_b_s[0]))


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731


def run(seq1, seq2):
    """seq = list of callables for ONE iteration of a stream (None: the stream stays idle)."""
    a, b1, b2 = ev(), ev(), ev()
    go = torch.cuda.Event()
    with torch.cuda.stream(s1):
        torch.cuda._sleep(4000000)
        go.record()
        a.record()
    s2.wait_event(go)
    with torch.cuda.stream(s2):
        if seq2:
            for _ in range(N_IT):
                for f in seq2:
                    f()
        b2.record()
    with torch.cuda.stream(s1):
        if seq1:
            for _ in range(N_IT):
                for f in seq1:
                    f()
        b1.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b1) * 1e3 / N_IT, a.elapsed_time(b2) * 1e3 / N_IT


hA, tA, hB, tB = (lambda: head("a")), (lambda: tail("a")), (lambda: head("b")), (lambda: tail("b"))
cases = [
    ("A head alone", [hA], None), ("A tail alone", [tA], None), ("B head alone", None, [hB]), ("B tail alone", None, [tB]),
    ("A chain alone", [hA, tA], None), ("B chain alone", None, [hB, tB]),
    ("heads beside each other", [hA], [hB]), ("tails beside each other", [tA], [tB]),
    ("A head beside B tail", [hA], [tB]), ("A tail beside B head", [tA], [hB]),
    ("chains in phase", [hA, tA], [hB, tB]), ("chains in anti phase", [hA, tA], [tB, hB]),
]
with torch.no_grad():
    for _, x, y in cases[:6]:
        run(x, y)
    res = {n: [] for n, _, _ in cases}
    for _ in range(5):
        for n, x, y in cases:
            res[n].append(run(x, y))
print(f"# {N_IT} iterations per stream, median of 5; us per iteration: stream 1, stream 2")
for n, x, y in cases:
    r = sorted(res[n], key=lambda t: max(t))[2]
    print(f"{n:28s} {r[0] if x else float('nan'):8.1f} {r[1] if y else float('nan'):8.1f}")
