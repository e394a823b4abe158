"""Generate tests/golden/train.npz by running the REFERENCE's training-step helpers
(train_flow.py: uniform_loss :55-79, fetch_optimizer :86-91, the clip + AdamW + scheduler sequence
:135-140, flo_A2B / valid_B :123-126) imported from /root/reference.  Build container only.

Extra inert shims on top of _refharness: stub cv2 / wandb / torchvision.transforms.ColorJitter,
and PriOr-RAFT/core first on sys.path so `import datasets` resolves to the reference's module.
"""
from __future__ import annotations

import argparse
import importlib
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, _HERE)

import golden_cases as gc  # noqa: E402
from _refharness import REF_ROOT, load_reference  # noqa: E402
from gen_golden import save  # noqa: E402


def loss_case(h=64, w=128, B=2, n_pred=3):
    gt = gc.flows("train/gt", B, h, w)
    gt[0, :, 7, :] = 350.0                       # |gt| = 495 > MAX_FLOW: masked out
    valid = (gc.uni("train/valid", (B, h, w), 0.0, 1.0) > 0.2).float()
    preds = [gt + torch.stack([gc.uni(f"train/du{i}", (B, h, w), -4, 4), gc.uni(f"train/dv{i}", (B, h, w), -3, 3)], 1)
             for i in range(n_pred)]
    preds[-1][1, :, 9, :] = gt[1, :, 9, :]       # exact hits: sign(0) = 0 in the gradient
    return preds, gt, valid


def adam_case(n=4099, steps=6):
    p0 = gc.uni("train/p0", (n,), -0.5, 0.5)
    grads = [gc.uni(f"train/g{i}", (n,), -1.0, 1.0) * (0.02 if i % 2 else 3.0) for i in range(steps)]   # clipped and unclipped steps
    return p0, grads


def main():
    load_reference()
    for name in ("cv2", "wandb"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["cv2"].setNumThreads = lambda n: None
    sys.modules["cv2"].ocl = types.SimpleNamespace(setUseOpenCL=lambda b: None)
    tv = types.ModuleType("torchvision"); tvt = types.ModuleType("torchvision.transforms")
    tvt.ColorJitter = type("ColorJitter", (), {"__init__": lambda self, *a, **k: None})
    tv.transforms = tvt
    sys.modules.setdefault("torchvision", tv); sys.modules.setdefault("torchvision.transforms", tvt)
    sys.path[:0] = [os.path.join(REF_ROOT, "core")]
    tf = importlib.import_module("train_flow")
    proj = importlib.import_module("core.utils.projection_prim_ortho")

    # ---- uniform_loss: value, metrics, autograd gradients --------------------------------------
    preds, gt, valid = loss_case()
    preds = [p.clone().requires_grad_(True) for p in preds]
    loss, metrics = tf.uniform_loss(64, 128)(preds, gt, valid, gamma=0.8, extro_info="A-")
    loss.backward()
    # ---- flo_A2B / valid_B ---------------------------------------------------------------------------
    with torch.no_grad():
        gt_b = proj.flo_A2B(gt)
        valid_b = ((gt_b[:, 0].abs() < 1000) & (gt_b[:, 1].abs() < 1000)).float()
    # ---- optimizer + scheduler ---------------------------------------------------------------------
    args = argparse.Namespace(lr=1e-4, wdecay=5e-5, epsilon=1e-8, num_steps=60000, clip=1.0)
    p0, grads = adam_case()
    model = torch.nn.ParameterList([torch.nn.Parameter(p0[:1000].clone()), torch.nn.Parameter(p0[1000:].clone())])
    opt, sched = tf.fetch_optimizer(args, model)
    lrs, norms = [], []
    for g in grads:
        opt.zero_grad()
        model[0].grad = g[:1000].clone(); model[1].grad = g[1000:].clone()
        norms.append(float(torch.nn.utils.clip_grad_norm_(model.parameters(), args.clip)))
        opt.step(); sched.step()
        lrs.append(opt.param_groups[0]["lr"])
    p_final = torch.cat([model[0].detach(), model[1].detach()])
    # the whole schedule, sampled
    m2 = torch.nn.Linear(2, 2)
    o2, s2 = tf.fetch_optimizer(args, m2)
    sched_lr = [o2.param_groups[0]["lr"]]
    for _ in range(args.num_steps + 99):
        o2.step(); s2.step()
        sched_lr.append(o2.param_groups[0]["lr"])
    sched_lr = np.asarray(sched_lr)
    idx = np.unique(np.concatenate([np.arange(0, 8), np.arange(2998, 3012), np.arange(0, 60100, 997), np.arange(60090, 60100)]))
    save("train", loss=loss.detach(), metrics=np.asarray([metrics["A-epe"], metrics["A-1px"], metrics["A-3px"], metrics["A-5px"]]),
         grad0=preds[0].grad[:, :, ::4, ::4], grad2=preds[2].grad[:, :, ::2, ::2],
         gt_b=gt_b[:, :, ::2, ::2], valid_b_sum=valid_b.sum(), lrs=np.asarray(lrs), norms=np.asarray(norms),
         p_final=p_final, sched_idx=idx, sched_lr=sched_lr[idx])
    print("first lrs", lrs[:5], "norms", norms)


if __name__ == "__main__":
    main()
