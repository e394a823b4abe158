#!/usr/bin/env python3
"""fnet and cnet each ALONE on the chip, one stream, eager: run under `rocprofv3 --kernel-trace` and read the per-kernel
durations with profiles/enc_alone_report.py.  usage: enc_alone.py [batch]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from prior_flow_amd import det_state_dict, synthetic_pair  # noqa: E402
from prior_flow_amd._lib import EPI_LINEAR, EPI_TANH_RELU  # noqa: E402
from prior_flow_amd.engine import Engine  # noqa: E402
from prior_flow_amd.prior_raft import PriOr_RAFT, state_dict_shapes  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
model = PriOr_RAFT(argparse.Namespace(mixed_precision=False, dropout=0.0))
model.load_state_dict(det_state_dict(state_dict_shapes()), strict=True)
model = model.to(dev).eval()
i1, i2 = (t.to(dev) for t in synthetic_pair(B, 512, 1024))
with torch.no_grad():
    ws = model._workspace(B, 512, 1024, dev)
    eng = Engine(model._lib(), None)
    cplan, fplan = model._encoder_plans()
    ctx = dict(outs=ws.net0_ab_s, auxs=ws.x_ab_s) if eng.presplit(model._weights()) else dict(aux=ws.x_ab)
    marker = torch.zeros(1 << 20, device=dev)
    for rep in range(3):
        eng.prepare_images(ws, i1, i2)
        torch.cuda.synchronize()
        marker.fill_(1.0)                 # marker kernel: fnet follows
        fplan.run(ws.img_f, ws.f_all, EPI_LINEAR)
        torch.cuda.synchronize()
        marker.fill_(2.0)                 # marker: cnet follows
        cplan.run(ws.img_c, ws.net0_ab, EPI_TANH_RELU, **ctx)
        torch.cuda.synchronize()
        marker.fill_(3.0)
print("done")
