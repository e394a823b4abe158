"""Generate the fixtures of the BASELINE.json configurations that are not bench lines, by running the
REFERENCE (imported from /root/reference) on CPU.  Build container only:  ``python oracle/gen_golden_configs.py``.

  * forward_256x512_demo.npz -- configs[0]: demo.py:11-19, two ``torch.randn(1,3,256,512)`` "images", iters=4.
    The inputs are stored in the fixture (rounded to fp16 to halve it; the reference ran on exactly the
    stored values), so the test does not depend on torch's RNG stream.
  * forward_640x1280_it32.npz -- configs[4]: FlowScape-sized 640x1280 panorama pair, iters=32, test_mode
    (evaluate.py:366-397 validate_FlowScape at iters=32) plus the EPE-by-region numbers of
    validate_FlowScape_regions' arithmetic (evaluate.py:285-330, regions All / Equator / Poles / Center from
    core/utils/polemask.py) for the reference's flow against the closed-form ground truth of
    ``config4_gt`` below.  The flow is stored at every 4th pixel.
"""
from __future__ import annotations

import importlib
import os
import sys
import time

import numpy as np
import torch

sys.dont_write_bytecode = True
_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, _HERE)

import golden_cases as gc  # noqa: E402
from _refharness import reference_model  # noqa: E402
from gen_golden import save  # noqa: E402

CFG4 = dict(h=640, w=1280, iters=32, seed=640)


def demo_inputs():
    """demo.py's inputs: N(0,1) noise straight into the 0..255 image domain (near-constant images)."""
    g = torch.Generator().manual_seed(1234)
    d1 = torch.randn(1, 3, 256, 512, generator=g).half()
    d2 = torch.randn(1, 3, 256, 512, generator=g).half()
    return d1, d2


def config4_gt(h: int = CFG4["h"], w: int = CFG4["w"]) -> torch.Tensor:
    """Closed-form stand-in for FlowScape's ground truth [2,h,w] (no dataset exists offline): a smooth field of
    the size of the synthetic pair's true motion (+5 px, +2 px) with a latitude-dependent part, so that the
    four regions see different errors."""
    ys = torch.linspace(-1.0, 1.0, h).view(h, 1).expand(h, w)
    xs = torch.linspace(-1.0, 1.0, w).view(1, w).expand(h, w)
    u = 5.0 + 2.0 * torch.cos(3.0 * ys) * torch.sin(2.0 * xs)
    v = 2.0 + 1.5 * ys * torch.cos(4.0 * xs)
    return torch.stack([u, v]).float().contiguous()


def reference_region_metrics(flow: torch.Tensor, flow_gt: torch.Tensor):
    """evaluate.py:285-330 for ONE sample [2,H,W]: rows All / Equator / Poles / Center, columns epe, sd, sd_uni."""
    sph = importlib.import_module("core.utils.spherical")
    pm = importlib.import_module("core.utils.polemask")
    h, w = flow.shape[-2:]
    pole, center = pm.generate_polemask(h, w)
    regions = {"All": torch.ones((h, w), dtype=torch.long).view(-1) >= 0.5,
               "Equator": (1 - pole).squeeze(0).view(-1) >= 0.5,
               "Poles": pole.squeeze(0).view(-1) >= 0.5,
               "Center": center.squeeze(0).view(-1) >= 0.5}
    uniform_mask = torch.from_numpy(sph.spherical_mask(h, w))
    epe = torch.sum((flow - flow_gt) ** 2, dim=0).sqrt()
    sd = sph.calculate_great_circle_distance(flow[None], flow_gt[None])[0]
    rows = []
    for name in ("All", "Equator", "Poles", "Center"):
        mk = regions[name]
        u = (sd * uniform_mask).view(-1)
        rows.append([epe.view(-1)[mk].mean().item(), sd.view(-1)[mk].mean().item(),
                     (u[mk] / torch.sum(uniform_mask.view(-1)[mk])).sum().item()])
    return np.asarray(rows, dtype=np.float64)


@torch.no_grad()
def main():
    torch.set_num_threads(8)
    with reference_model(gc.det_state_dict) as m:
        d1, d2 = demo_inputs()
        out = m(d1.float(), d2.float(), iters=4, test_mode=True)
        save("forward_256x512_demo", out=out[:, :, ::2, ::2], image1=d1.numpy(), image2=d2.numpy())
        c = CFG4
        i1, i2 = gc.synthetic_pair(1, c["h"], c["w"], seed=c["seed"])
        t0 = time.time()
        flow = m(i1, i2, iters=c["iters"], test_mode=True)
        print(f"reference forward {c['h']}x{c['w']} iters={c['iters']}: {time.time() - t0:.1f} s on 8 cores")
        regions = reference_region_metrics(flow[0], config4_gt())
        save("forward_640x1280_it32", out=flow[:, :, ::4, ::4], regions=regions,
             flow_absmean=np.float64(flow.abs().mean()))
        print("regions (epe, sd, sd_uni):\n", regions)


if __name__ == "__main__":
    main()
