"""Seeded input builders shared by the golden generator and the parity tests.

TEST INFRASTRUCTURE.  Inputs are regenerated from the closed-form filler in
``prior-flow_amd/synthetic.py`` (bit-identical everywhere), so fixtures under
``tests/golden/`` only hold the reference's OUTPUTS (plus tiny inputs where handy).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from prior_flow_amd.synthetic import _uniform01, det_state_dict, synthetic_pair  # noqa: E402

GOLDEN_DIR = os.path.join(_ROOT, "tests", "golden")

H8, W8 = 16, 32            # 1/8-resolution test map (128x256 image; smallest legal size)
C = 256


def uni(name: str, shape, lo: float = -1.0, hi: float = 1.0) -> torch.Tensor:
    n = int(np.prod(shape))
    u = _uniform01("golden/" + name, n)
    return torch.from_numpy((lo + (hi - lo) * u).astype(np.float32).reshape(shape))


def fmaps(tag: str, B: int = 1, h: int = H8, w: int = W8):
    """Two feature maps with the statistics of instance-normed encoder outputs."""
    return uni(tag + "/f1", (B, C, h, w), -1.7, 1.7), uni(tag + "/f2", (B, C, h, w), -1.7, 1.7)


def nasty_coords(tag: str, B: int = 1, h: int = H8, w: int = W8) -> torch.Tensor:
    """coords1 [B,2,h,w]: identity grid + flows with negatives, seam crossers
    (x in (w-1, w)), out-of-range y, large displacements and exact integers."""
    xs = torch.arange(w).view(1, 1, w).expand(B, h, w).float()
    ys = torch.arange(h).view(1, h, 1).expand(B, h, w).float()
    fx = uni(tag + "/fx", (B, h, w), -6.0, 6.0)
    fy = uni(tag + "/fy", (B, h, w), -6.0, 6.0)
    # structured special cases in the first rows
    fx[:, 0, :] = 0.0                      # zero flow -> integer coordinates
    fy[:, 0, :] = 0.0
    fx[:, 1, :] = (w - 0.5) - xs[:, 1, :]  # every pixel samples x = w-0.5 (seam fade)
    fx[:, 2, :] = -xs[:, 2, :] - 0.25      # x = -0.25 (wraps to w-0.25)
    fy[:, 3, :] = -ys[:, 3, :] - 0.5       # y = -0.5 (half out of range)
    fy[:, 4, :] = (h + 3.0) - ys[:, 4, :]  # far below the map
    fx[:, 5, :] = 2.5 * w                  # multiple wraps
    return torch.stack([xs + fx, ys + fy], dim=1).contiguous()


def flows(tag: str, B: int = 1, h: int = H8, w: int = W8) -> torch.Tensor:
    f = torch.stack([uni(tag + "/u", (B, h, w), -7.0, 7.0), uni(tag + "/v", (B, h, w), -5.0, 5.0)], 1)
    f[:, :, 0, :] = 0.0                    # zero flow row -> exactly zero rotated flow
    f[:, 0, 1, :] = w / 2.0                # +-W/2 wrap
    f[:, 0, 2, :] = -w / 2.0
    f[:, 1, 3, :] = -h                     # clamp at the top
    f[:, 1, 4, :] = 2.0 * h                # clamp at the bottom
    return f.contiguous()


def volumes(tag: str, B: int = 1, h: int = H8, w: int = W8):
    """Two independent level-0 volumes [B,h,w,h,w] (white noise: worst case for sampling)."""
    return uni(tag + "/va", (B, h, w, h, w), -4.0, 4.0), uni(tag + "/vb", (B, h, w, h, w), -4.0, 4.0)


def update_inputs(tag: str, B: int = 1, h: int = H8, w: int = W8):
    d = dict(
        net=torch.tanh(uni(tag + "/net", (B, 128, h, w), -2, 2)),
        inp=torch.relu(uni(tag + "/inp", (B, 128, h, w), -1, 1)),
        flow_a=uni(tag + "/flow_a", (B, 2, h, w), -6, 6),
        flow_ba=uni(tag + "/flow_ba", (B, 2, h, w), -6, 6),
        corr=uni(tag + "/corr", (B, 324, h, w), -3, 3),
        flaw_a=uni(tag + "/flaw_a", (B, 4, h, w), -0.5, 0.5),
        flaw_ba=uni(tag + "/flaw_ba", (B, 4, h, w), -0.5, 0.5),
    )
    return d


def load(name: str):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"))


__all__ = ["uni", "fmaps", "nasty_coords", "flows", "volumes", "update_inputs", "load",
           "det_state_dict", "synthetic_pair", "GOLDEN_DIR", "H8", "W8", "C"]
