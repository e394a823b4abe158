"""Generate tests/golden/eval.npz and tests/golden/flo_io.npz by running the REFERENCE's
evaluation helpers imported from /root/reference.

Run only in the build container:  ``python oracle/gen_golden_eval.py``.
Fixtures hold inputs/outputs only.  Inputs come from ``oracle/golden_cases.py``.
Extra inert shims on top of _refharness: a stub ``cv2`` (frame_utils.py:6-8 only calls
``setNumThreads`` / ``ocl.setUseOpenCL`` at import).
"""
from __future__ import annotations

import importlib
import os
import sys
import tempfile
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, _HERE)

import golden_cases as gc  # noqa: E402
from _refharness import load_reference  # noqa: E402
from gen_golden import save  # noqa: E402


def eval_samples(n=3, h=64, w=128):
    """(pred, gt) flow pairs [2,h,w] with wraps, clamps and a pole-crossing row."""
    out = []
    for i in range(n):
        gt = gc.flows(f"eval/gt{i}", 1, h, w)[0]
        pr = gt + torch.stack([gc.uni(f"eval/du{i}", (h, w), -1.5, 1.5), gc.uni(f"eval/dv{i}", (h, w), -1.0, 1.0)])
        pr[1, 0, :] = -3.0                    # leaves the map at the top: clamped end point
        pr[0, 5, :] = w * 1.5 + 0.25          # more than one wrap
        out.append((pr.contiguous(), gt.contiguous()))
    return out


@torch.no_grad()
def main():
    load_reference()
    cv2 = types.ModuleType("cv2")
    cv2.setNumThreads = lambda n: None
    cv2.ocl = types.SimpleNamespace(setUseOpenCL=lambda b: None)
    sys.modules.setdefault("cv2", cv2)
    sph = importlib.import_module("core.utils.spherical")
    pm = importlib.import_module("core.utils.polemask")
    fu = importlib.import_module("core.utils.frame_utils")

    # ---- SEPE -----------------------------------------------------------------------------
    pre, gt = gc.flows("eval/pre", 2), gc.flows("eval/gt", 2)
    sd_rand = sph.calculate_great_circle_distance(pre, gt)
    kat = torch.zeros(1, 2, 64, 128)
    kat[:, 0] = 4.0
    sd_kat = sph.calculate_great_circle_distance(kat, torch.zeros_like(kat))[0, :, 0]
    # ---- masks ----------------------------------------------------------------------------
    masks = {}
    for h, w in ((16, 32), (64, 128)):
        a, b = pm.generate_polemask(h, w)
        masks[f"pole_a_{h}x{w}"] = a.numpy().astype(np.uint8)
        masks[f"pole_b_{h}x{w}"] = b.numpy().astype(np.uint8)
    uni = sph.spherical_mask(64, 128)
    # ---- region metrics: the arithmetic of evaluate.py:196-227 / :234-282 on fixed flows ------------
    samples = eval_samples()
    h, w = 64, 128
    pole, center = pm.generate_polemask(h, w)
    regions = {"All": torch.ones((h, w), dtype=torch.long).view(-1) >= 0.5,
               "Equator": (1 - pole).squeeze(0).view(-1) >= 0.5,
               "Poles": pole.squeeze(0).view(-1) >= 0.5,
               "Center": center.squeeze(0).view(-1) >= 0.5}
    uniform_mask = torch.from_numpy(uni)
    res = []
    for name in ("All", "Equator", "Poles", "Center"):
        mk = regions[name]
        epe_list, sd_list, sd_uni_list = [], [], []
        for flow, flow_gt in samples:
            epe = torch.sum((flow - flow_gt) ** 2, dim=0).sqrt()
            sd = sph.calculate_great_circle_distance(flow[None], flow_gt[None])[0]
            epe_list.append(epe.view(-1)[mk].numpy())
            sd_list.append(sd.view(-1)[mk].numpy())
            u = (sd * uniform_mask).view(-1)
            u = u[mk] / torch.sum(uniform_mask.view(-1)[mk])
            sd_uni_list.append(torch.sum(u).item())
        res.append([np.mean(np.concatenate(epe_list)), np.mean(np.array(sd_list)), np.mean(np.array(sd_uni_list))])
    save("eval", sd_rand=sd_rand, sd_kat=sd_kat, uni_col=uni[:, 0], uni_sum=np.float64(uni.sum()),
         regions=np.asarray(res, dtype=np.float64), **masks)

    # ---- .flo -----------------------------------------------------------------------------
    flo = gc.uni("flo/uv", (5, 7, 2), -30, 30).numpy()
    with tempfile.TemporaryDirectory() as d:
        fn = os.path.join(d, "x.flo")
        fu.writeFlow(fn, flo)
        raw = np.fromfile(fn, dtype=np.uint8)
        back = fu.readFlow(fn)
        fu.writeFlow(fn, flo[:, :, 0], flo[:, :, 1])
        raw2 = np.fromfile(fn, dtype=np.uint8)
    assert np.array_equal(raw, raw2)
    save("flo_io", bytes=raw, read=back)


if __name__ == "__main__":
    main()
