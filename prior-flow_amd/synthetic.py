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
    """Closed-form value for one state_dict entry, chosen by its name suffix."""
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if len(shape) else 1
    if name.endswith("num_batches_tracked"):
        return torch.zeros(shape, dtype=torch.long)
    u = _uniform01(name, n)
    if name.endswith("running_var"):
        v = 0.5 + u  # [0.5, 1.5)
    elif name.endswith("running_mean"):
        v = (u - 0.5) * 0.2
    elif len(shape) == 4:  # conv weight [Cout, Cin, KH, KW]
        fan_in = shape[1] * shape[2] * shape[3]
        bound = np.sqrt(3.0 / fan_in)  # unit-variance-preserving uniform
        v = (2.0 * u - 1.0) * bound
    elif name.endswith("weight"):  # norm scale
        v = 0.8 + 0.4 * u
    else:  # bias
        v = (2.0 * u - 1.0) * 0.05
    return torch.from_numpy(v.astype(np.float32).reshape(shape))


def det_state_dict(shapes: Mapping[str, Iterable[int]]) -> Dict[str, torch.Tensor]:
    """Deterministic state_dict for a mapping name -> shape (e.g. from ``model.state_dict()``)."""
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
