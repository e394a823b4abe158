"""Where the training step's PyTorch kernels come from: one eager train_step under torch.profiler; per (aten op, innermost frame
of this package -- or, where the build records no Python stacks, the operand shapes) the GPU kernels launched and their time.
    python profiles/native_sources.py"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from prior_flow_amd import det_state_dict, synthetic_pair
from prior_flow_amd import train as tr
from prior_flow_amd.modules import state_dict_shapes
from prior_flow_amd.prior_raft import PriOr_RAFT

dev = torch.device("cuda:0")
H, W = 384, 512
model = PriOr_RAFT(argparse.Namespace(mixed_precision=False, dropout=0.0))
model.load_state_dict(det_state_dict(state_dict_shapes()), strict=True)
model = model.to(dev).train()
model.freeze_bn()
opt, sched = tr.fetch_optimizer(argparse.Namespace(lr=2e-5, wdecay=5e-5, epsilon=1e-8, num_steps=100000), model)
i1, i2 = (t.to(dev) for t in synthetic_pair(1, H, W, seed=1234))
gt = (torch.rand(1, 2, H, W) * 8 - 4).to(dev)
valid = torch.ones(1, H, W, device=dev)
crit = tr.uniform_loss(H, W, device=dev)
for _ in range(2):
    tr.train_step(model, opt, sched, crit, i1, i2, gt, valid, iters=12, clip=1.0)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tr.train_step(model, opt, sched, crit, i1, i2, gt, valid, iters=12, clip=1.0)
    torch.cuda.synchronize()
by = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0 or not ev.kernels:
        continue
    frame = "?"
    for fr in ev.stack:
        if "prior_flow_amd" in fr or "prior-flow_amd" in fr:
            frame = fr.split("prior")[-1][-70:]
            break
    if frame == "?":                      # (no Python stacks on this build: the operand shapes identify the call site)
        frame = str([tuple(x) for x in (ev.input_shapes or []) if x])[:90]
    d = by[(ev.name, frame)]
    d[0] += len(ev.kernels)
    d[1] += sum(k.duration for k in ev.kernels)
tot = sum(v[0] for v in by.values())
print(f"# {tot} kernels launched by aten ops in one eager step")
for (name, frame), (n, t) in sorted(by.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f"{n:4d} kernels {t:8.1f} us  {name:24s} {frame}")
