"""Generate tests/golden/train_step.npz: loss, total gradient norm and gradient slices of ONE training step
of the REFERENCE (train_flow.py:119-137: forward in train mode with frozen BN, uniform_loss on both branches,
backward) at B=2, 128x256, iters=3 -- the pin for the backward kernels.  Build container only."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, _HERE)

import golden_cases as gc  # noqa: E402
from _refharness import load_reference  # noqa: E402
from gen_golden import save  # noqa: E402

SLICES = {"fnet.conv1.weight": (slice(0, 4), slice(None), slice(0, 3), slice(0, 3)),
          "fnet.layer3.1.conv2.weight": (slice(0, 2), slice(0, 8)),
          "cnet.conv2.weight": (slice(0, 4), slice(0, 8)),
          "ODDC.gru.convz1.weight": (slice(0, 2), slice(0, 8)),
          "ODDC.encoder.convc1_A.weight": (slice(0, 2), slice(0, 16)),
          "ODDC.encoder.conv_conf1.weight": (slice(0, 2),),
          "ODDC.flow_head.conv2.weight": (slice(None), slice(0, 8)),
          "ODDC.mask.2.bias": (slice(0, 16),),
          "update_block.gru.convq2.weight": (slice(0, 2), slice(0, 8)),
          "update_block.encoder.convf1.weight": (slice(0, 2),),
          "update_block.mask.0.weight": (slice(0, 2), slice(0, 8))}


def step_inputs():
    i1, i2 = gc.synthetic_pair(2, 128, 256, seed=5)
    gt = gc.flows("step/gt", 2, 128, 256) * 0.5
    valid = (gc.uni("step/valid", (2, 128, 256), 0.0, 1.0) > 0.1).float()
    return i1, i2, gt, valid


def main():
    torch.set_num_threads(8)
    ref = load_reference()
    import importlib
    import types
    for name in ("cv2", "wandb"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sph = importlib.import_module("core.utils.spherical")
    model = ref.prior_raft.PriOr_RAFT(ref.args())
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(gc.det_state_dict(shapes), strict=True)
    model.train()
    model.freeze_bn()
    i1, i2, gt, valid = step_inputs()
    with torch.no_grad():
        gt_b = ref.proj.flo_A2B(gt)
        valid_b = ((gt_b[:, 0].abs() < 1000) & (gt_b[:, 1].abs() < 1000)).float()
    uni = torch.from_numpy(sph.spherical_mask(128, 256))[None]

    def loss_fn(preds, g, v, gamma=0.8):          # the arithmetic of train_flow.py:62-71
        n = len(preds)
        mag = torch.sum(g ** 2, dim=1).sqrt()
        ok = (v >= 0.5) & (mag < 400)
        tot = 0.0
        for i in range(n):
            tot = tot + gamma ** (n - i - 1) * torch.sum(ok * uni * torch.sum((preds[i] - g).abs(), dim=1))
        return tot

    pa, pb = model(i1, i2, iters=3)
    loss = loss_fn(pa, gt, valid) + loss_fn(pb, gt_b, valid_b)
    loss.backward()
    grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values()))
    norms = {k: float(g.double().norm()) for k, g in grads.items()}
    print("loss", float(loss), "total grad norm", float(total), "params with grad", len(grads))
    out = {"loss": loss.detach(), "grad_norm": total, "names": np.array(sorted(norms)),
           "norms": np.array([norms[k] for k in sorted(norms)])}
    for k, sl in SLICES.items():
        out["g:" + k] = grads[k][sl]
    save("train_step", **out)


if __name__ == "__main__":
    main()
