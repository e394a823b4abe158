"""Evaluation counterpart of the reference, on the HIP kernels.

Mirrors, with the reference's names and argument meaning:
  * ``InputPadder`` (core/utils/utils.py:7-27),
  * ``spherical_mask`` / ``calculate_great_circle_distance`` (core/utils/spherical.py:11-17, :20-53),
  * ``generate_polemask`` (core/utils/polemask.py:7-26),
  * the region loop of ``validate_MPF_regions`` / ``validate_FlowScape_regions`` /
    ``validate_360cityflow`` (evaluate.py:234-282, :285-330, :160-227): EPE, SEPE (``sd``) and the
    cos-latitude weighted SEPE (``sd_uni``) over All / Equator / Poles / Center.

Differences that are deliberate: the datasets do not exist offline, so ``validate_regions`` takes any
iterable of ``(image1, image2, flow_gt[, valid])`` samples; the model runs ONCE per sample (the
reference re-runs it for every region, evaluate.py:245-263) and all regions are reduced by one kernel
pass; ``validate_FlowScape_regions``' unpacking bug (evaluate.py:300) is not reproduced.
Per-pixel metrics and region sums run on the GPU (``pf_flow_metrics`` / ``pf_region_sums``); there is
no CPU fallback.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib
from .engine import rotation_x

REGIONS = ("All", "Equator", "Poles", "Center")


class InputPadder:
    """Replicate-pads a batch up to the next multiple of 8 in H and W and crops results back
    (API of core/utils/utils.py:7-27: ``pad(*tensors)``, ``unpad(tensor)``).  ``mode='sintel'`` centres the
    image in the padded frame (the odd pixel goes to the bottom / right); any other mode keeps the top edge
    and pads the bottom only, the width still being centred."""

    MULTIPLE = 8

    def __init__(self, dims, mode="sintel"):
        height, width = int(dims[-2]), int(dims[-1])
        extra_h, extra_w = -height % self.MULTIPLE, -width % self.MULTIPLE
        top = extra_h // 2 if mode == "sintel" else 0
        left = extra_w // 2
        self.ht, self.wd = height, width
        # (left, right, top, bottom): the order torch.nn.functional.pad takes for the last two dims
        self._pad = [left, extra_w - left, top, extra_h - top]

    def pad(self, *inputs):
        if not any(self._pad):
            return list(inputs)
        return [F.pad(t, self._pad, mode="replicate") for t in inputs]

    def unpad(self, x):
        left, right, top, bottom = self._pad
        rows, cols = x.shape[-2] - bottom, x.shape[-1] - right     # relative to x's own size, like the reference
        return x[..., top:rows, left:cols]


def spherical_mask(H: int, W: int) -> np.ndarray:
    """cos(latitude) weights normalised to sum 1, as a numpy array (core/utils/spherical.py:11-17)."""
    n = torch.arange(0, H).view(-1, 1).repeat(1, W)
    phi = (0.5 - (n + 0.5) / H) * math.pi
    m = torch.cos(phi).numpy()
    return m / np.sum(m)


def calculate_great_circle_distance(pre_flow: torch.Tensor, gt_flow: torch.Tensor, method: str = "Haversine",
                                    R: float = 1) -> torch.Tensor:
    """[B,2,H,W] x2 -> [B,H,W] great-circle distance between the flows' end points (core/utils/spherical.py:20-53; both
    methods: 'Haversine', which evaluate.py uses, and 'Cosine', the arccos form -- NaN where rounding takes its argument past 1,
    as in the reference)."""
    assert method in ["Haversine", "Cosine"]
    assert (pre_flow.shape == gt_flow.shape) and (pre_flow.shape[1] == 2)
    lib = _lib.load()
    pre = pre_flow.float().contiguous()
    gt = gt_flow.float().contiguous()
    sd = torch.empty(pre.shape[0], pre.shape[2], pre.shape[3], device=pre.device, dtype=torch.float32)
    lib.flow_metrics(pre, gt, None, sd, cosine=(method == "Cosine"))
    return sd if R == 1 else sd * R


@torch.no_grad()
def generate_polemask(H: int, W: int, delta_phi: float = np.pi / 2, device=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """(pole_mask_A, pole_mask_B) long [1,H,W] on the GPU (core/utils/polemask.py:7-26).  B is A
    resampled into view B by the img_rotate kernel with the grid of Rx(-pi/2), binarised at 0.5."""
    lib = _lib.load()
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    phi2n = lambda phi: (0.5 - phi / np.pi) * H - 0.5                       # noqa: E731
    min_n = int(np.round(phi2n(delta_phi / 2)))
    max_n = int(np.round(phi2n(-delta_phi / 2)))
    center = torch.zeros((1, H, W), device=device)
    center[:, min_n:max_n, :] = 1
    pole_a = (1 - center).contiguous()
    grid = torch.empty(2, H, W, device=device)
    lib.sample_grid(grid, rotation_x(-math.pi / 2))
    pole_b = torch.empty(1, 1, H, W, device=device)
    lib.img_rotate(pole_a.view(1, 1, H, W), grid, pole_b)
    pole_b = pole_b.view(1, H, W)
    pole_b[pole_b < 0.5] = 0
    pole_b[pole_b > 0] = 1
    return pole_a.long(), pole_b.long()


class RegionEvaluator:
    """Accumulates the region metrics of evaluate.py:234-282 over samples of one resolution."""

    NBLK = 64

    def __init__(self, H: int, W: int, device=None, regions: Sequence[str] = REGIONS):
        self.lib = _lib.load()
        self.H, self.W = H, W
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        pole, center = generate_polemask(H, W, device=self.device)
        named = {"All": torch.ones(H * W, dtype=torch.bool, device=self.device),
                 "Equator": (1 - pole).view(-1) >= 0.5, "Poles": pole.view(-1) >= 0.5, "Center": center.view(-1) >= 0.5}
        self.regions = tuple(regions)
        self.masks = {k: named[k] for k in self.regions}
        bits = torch.zeros(H * W, dtype=torch.int32, device=self.device)
        for r, k in enumerate(self.regions):
            bits |= self.masks[k].int() << r
        self.bits = bits.to(torch.uint8).contiguous()
        self.uniform = torch.from_numpy(spherical_mask(H, W).astype(np.float32)).to(self.device).view(-1).contiguous()
        self.count = torch.stack([self.masks[k].sum() for k in self.regions]).double()           # pixels per region
        self.wsum = torch.stack([self.uniform[self.masks[k]].double().sum() for k in self.regions])
        self.sum_epe = torch.zeros(len(self.regions), dtype=torch.float64, device=self.device)
        self.sum_sd = torch.zeros_like(self.sum_epe)
        self.sum_uni = torch.zeros_like(self.sum_epe)
        self.images = 0

    @torch.no_grad()
    def update(self, flow: torch.Tensor, flow_gt: torch.Tensor):
        """flow, flow_gt: [B,2,H,W] (or [2,H,W]) on the GPU.  Returns the per-pixel (epe, sd) maps."""
        if flow.dim() == 3:
            flow, flow_gt = flow[None], flow_gt[None]
        flow = flow.float().contiguous()
        flow_gt = flow_gt.to(flow.device).float().contiguous()
        B = flow.shape[0]
        epe = torch.empty(B, self.H, self.W, device=flow.device)
        sd = torch.empty_like(epe)
        self.lib.flow_metrics(flow, flow_gt, epe, sd)
        part = torch.empty(B, self.NBLK, len(self.regions), 3, dtype=torch.float64, device=flow.device)
        self.lib.region_sums(epe, sd, self.uniform, self.bits, len(self.regions), part)
        tot = part.sum(dim=1)                                    # [B, R, 3]
        self.sum_epe += tot[:, :, 0].sum(0)
        self.sum_sd += tot[:, :, 1].sum(0)
        self.sum_uni += (tot[:, :, 2] / self.wsum.to(tot.device)).sum(0)     # per-image weighted mean (:208-213)
        self.images += B
        return epe, sd

    def results(self) -> Dict[str, Dict[str, float]]:
        n = self.count.to(self.sum_epe.device) * max(self.images, 1)
        epe = (self.sum_epe / n).tolist()
        sd = (self.sum_sd / n).tolist()
        uni = (self.sum_uni / max(self.images, 1)).tolist()
        return {k: {"epe": epe[r], "sd": sd[r], "sd_uni": uni[r]} for r, k in enumerate(self.regions)}


@torch.no_grad()
def validate_regions(model, dataset: Iterable, iters: int = 12, scene: str = "synthetic",
                     regions: Sequence[str] = REGIONS, verbose: bool = True) -> Dict[str, Dict[str, float]]:
    """The body of validate_MPF_regions (evaluate.py:234-282) for any iterable of
    ``(image1 [3,H,W], image2, flow_gt [2,H,W], ...)`` samples: pad, ``model(..., test_mode=True)``,
    unpad, EPE / SEPE per region.  Returns ``{region: {"epe", "sd", "sd_uni"}}`` and prints the
    reference's per-region line."""
    ev: Optional[RegionEvaluator] = None
    for sample in dataset:
        image1, image2, flow_gt = sample[0], sample[1], sample[2]
        image1 = image1[None].cuda()
        image2 = image2[None].cuda()
        if ev is None:
            ev = RegionEvaluator(image1.shape[-2], image1.shape[-1], device=image1.device, regions=regions)
        padder = InputPadder(image1.shape)
        image1, image2 = padder.pad(image1, image2)
        flow_pr = model(image1.contiguous(), image2.contiguous(), iters=iters, test_mode=True)
        flow = padder.unpad(flow_pr[0])
        ev.update(flow[None], flow_gt[None].cuda())
    if ev is None:
        raise ValueError("validate_regions: empty dataset")
    res = ev.results()
    if verbose:
        for region in res:
            print(f"{region:>7}-{scene}: epe {res[region]['epe']: .3f}, sd {res[region]['sd']: .8f}")
    return res


def validate_MPF_regions(model, iters=12, scene="EFT", dataset=None):
    """evaluate.py:234-282.  The MPFDataset files are not available offline: pass ``dataset``."""
    if dataset is None:
        raise FileNotFoundError("MPFDataset is not available in this build: pass dataset=<iterable of samples>")
    return validate_regions(model, dataset, iters=iters, scene=scene)


def validate_FlowScape_regions(model, iters=12, scene="sunny", dataset=None):
    """evaluate.py:285-330 (without its list-unpacking bug at :300)."""
    if dataset is None:
        raise FileNotFoundError("FlowScape is not available in this build: pass dataset=<iterable of samples>")
    return validate_regions(model, dataset, iters=iters, scene=scene)


@torch.no_grad()
def validate(model, dataset: Iterable, iters: int = 12, name: str = "synthetic", verbose: bool = True) -> Dict[str, float]:
    """The body of the reference's plain validation loops (evaluate.py:337-366 ``validate_MPF``, :369-397 ``validate_FlowScape`` --
    what ``train_flow.py:187-194`` calls every VAL_FREQ steps): ``model.eval()``, per sample pad, ``model(..., test_mode=True)``,
    unpad; EPE = mean of sqrt(du^2 + dv^2) over the pixels of ALL samples, SEPE = mean over samples of the per-sample mean
    great-circle distance.  The per-pixel maps come from ``pf_flow_metrics`` (one launch per sample), the sums stay on the device
    in fp64 until the end.  Returns ``{name + "-epe", name + "-SEPE"}`` and prints the reference's line."""
    lib = _lib.load()
    # every module's own flag, not just the root's: a caller that froze its BatchNorm layers (freeze_bn, train_flow.py:107-108)
    # must get them back frozen -- a blanket model.train() would put them into training mode (batch statistics, running stats
    # mutated) while an already captured GraphedTrainStep keeps replaying the frozen kernels
    modes = {m: m.training for m in model.modules()} if isinstance(model, torch.nn.Module) else {}
    model.eval()
    epe_sum = None
    pixels, sd_means = 0, []
    for sample in dataset:
        image1, image2, flow_gt = sample[0], sample[1], sample[2]
        image1 = image1[None].cuda()
        image2 = image2[None].cuda()
        padder = InputPadder(image1.shape)
        image1, image2 = padder.pad(image1, image2)
        flow_pr = model(image1.contiguous(), image2.contiguous(), iters=iters, test_mode=True)
        flow = padder.unpad(flow_pr[0])[None].float().contiguous()
        gt = flow_gt[None].to(flow.device).float().contiguous()
        epe = torch.empty(1, flow.shape[-2], flow.shape[-1], device=flow.device)
        sd = torch.empty_like(epe)
        lib.flow_metrics(flow, gt, epe, sd)
        e = epe.double().sum()
        epe_sum = e if epe_sum is None else epe_sum + e
        pixels += epe.numel()
        sd_means.append(sd.double().mean())
    if epe_sum is None:
        raise ValueError("validate: empty dataset")
    for m, was in modes.items():            # (the reference's callers do model.train() + freeze_bn() themselves, train_flow.py:196-198)
        m.training = was
    epe_v = float(epe_sum / pixels)
    sd_v = float(torch.stack(sd_means).mean())
    if verbose:
        print("Validation (%s) EPE: %f, SEPE: %f" % (name, epe_v, sd_v))
    return {f"{name}-epe": epe_v, f"{name}-SEPE": sd_v}


def validate_MPF(model, iters=12, scene="EFT", dataset=None):
    """evaluate.py:337-366.  The MPFDataset files are not available offline: pass ``dataset``."""
    if dataset is None:
        raise FileNotFoundError("MPFDataset is not available in this build: pass dataset=<iterable of samples>")
    return validate(model, dataset, iters=iters, name=scene)


def validate_FlowScape(model, iters=12, scene="sunny", dataset=None):
    """evaluate.py:369-397."""
    if dataset is None:
        raise FileNotFoundError("FlowScape is not available in this build: pass dataset=<iterable of samples>")
    return validate(model, dataset, iters=iters, name=f"FlowScape-{scene}")
