"""Deterministic synthetic weights and image pairs (no datasets / checkpoints offline).

Both the golden-vector generator (``oracle/gen_golden.py``, which imports the
reference in the build container) and the product / tests / bench regenerate
*identical* weights from this closed-form filler, so no 33 MB fixture has to be
committed.  Everything is integer arithmetic (splitmix64) followed
by one exact int->float conversion, so values are bit-identical on every machine.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Mapping, Tuple

import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        z = z ^ (z >> np.uint64(31))
    return z


def _uniform01(name: str, n: int, salt: int = 0) -> np.ndarray:
    """n values in [0,1) with 24 random bits each, keyed by (name, salt)."""
    seed = np.uint64(zlib.crc32(name.encode("utf-8")) + (salt << 32))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + seed
    h = _splitmix64(idx)
    return (h >> np.uint64(40)).astype(np.float64) / float(1 << 24)


def det_tensor(name: str, shape: Tuple[int, ...]) -> torch.Tensor:
    """Closed-form value for one state_dict entry, chosen by its name.

    Scales follow the statistics of a freshly constructed reference model (what demo.py
    runs): encoder conv weights have kaiming_normal(fan_out, relu) variance 2/fan_out
    (core/extractor.py:123-125), update-block conv weights and all conv biases PyTorch's
    default U(-1/sqrt(fan_in), 1/sqrt(fan_in)).  With that scale the refinement loop is
    contractive like a trained network (a 1e-4 input perturbation grows to ~6e-6 EPE after
    12 iterations at 512x1024), whereas unit-gain weights make it chaotic (2e-3) and no two
    fp32 implementations -- not even the oracle and the reference -- agree to 1e-3.
    Norm statistics are deliberately non-trivial so BatchNorm arithmetic is exercised.
    """
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if len(shape) else 1
    if name.endswith("num_batches_tracked"):
        return torch.zeros(shape, dtype=torch.long)
    u = _uniform01(name, n)
    encoder = name.startswith(("fnet.", "cnet."))
    if name.endswith("running_var"):
        v = 0.5 + u  # [0.5, 1.5)
    elif name.endswith("running_mean"):
        v = (u - 0.5) * 0.2
    elif len(shape) == 4:  # conv weight [Cout, Cin, KH, KW]
        fan_in = shape[1] * shape[2] * shape[3]
        fan_out = shape[0] * shape[2] * shape[3]
        bound = np.sqrt(6.0 / fan_out) if encoder else 1.0 / np.sqrt(fan_in)
        v = (2.0 * u - 1.0) * bound
    elif ".norm" in name or ".downsample.1." in name:  # BatchNorm affine
        v = (0.8 + 0.4 * u) if name.endswith("weight") else (2.0 * u - 1.0) * 0.05
    else:  # conv bias: U(-1/sqrt(fan_in), 1/sqrt(fan_in)); fan_in from the sibling weight
        v = (2.0 * u - 1.0) * _BIAS_BOUND.get(name, 0.05)
    return torch.from_numpy(v.astype(np.float32).reshape(shape))


_BIAS_BOUND: Dict[str, float] = {}


def det_state_dict(shapes: Mapping[str, Iterable[int]]) -> Dict[str, torch.Tensor]:
    """Deterministic state_dict for a mapping name -> shape (e.g. from ``model.state_dict()``)."""
    for k, shp in shapes.items():      # conv bias bound = 1/sqrt(fan_in of the sibling weight)
        shp = tuple(shp)
        if k.endswith(".weight") and len(shp) == 4:
            _BIAS_BOUND[k[:-len("weight")] + "bias"] = 1.0 / float(np.sqrt(shp[1] * shp[2] * shp[3]))
    return {k: det_tensor(k, tuple(s)) for k, s in shapes.items()}


def synthetic_pair(batch: int, height: int, width: int, seed: int = 1234,
                   shift: Tuple[int, int] = (2, 5)) -> Tuple[torch.Tensor, torch.Tensor]:
    """Textured in-range ERP pair with a real displacement and a horizontal wrap.

    image1 = 255 * bilinear_x4(U[0,1) noise [B,3,H/4,W/4]); image2 = roll(image1, shift)
    (SURVEY.md §8d).  The noise is the splitmix stream above, not torch's RNG.
    """
    h4, w4 = height // 4, width // 4
    u = _uniform01("synthetic_pair", batch * 3 * h4 * w4, salt=seed)
    low = torch.from_numpy(u.astype(np.float32).reshape(batch, 3, h4, w4))
    img1 = 255.0 * torch.nn.functional.interpolate(
        low, size=(height, width), mode="bilinear", align_corners=True)
    img2 = torch.roll(img1, shifts=shift, dims=(2, 3))
    return img1.contiguous(), img2.contiguous()
