"""Generate tests/golden/forward_odd.npz: the REFERENCE's forward at sizes whose 1/8 map is not a
multiple of 8 (odd pyramid levels pool with avg_pool2d's floor, core/corr.py:108; partial conv tiles).
Build container only:  ``python oracle/gen_golden_odd.py``."""
from __future__ import annotations

import os
import sys

import torch

sys.dont_write_bytecode = True
_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, _HERE)

import golden_cases as gc  # noqa: E402
from _refharness import reference_model  # noqa: E402
from gen_golden import save  # noqa: E402

ODD_SIZES = ((136, 216), (160, 360))      # 1/8 maps 17x27 (odd at every level) and 20x45


@torch.no_grad()
def main():
    torch.set_num_threads(8)
    out = {}
    with reference_model(gc.det_state_dict) as m:
        for h, w in ODD_SIZES:
            i1, i2 = gc.synthetic_pair(1, h, w, seed=31)
            pa, pb = m(i1, i2, iters=3)
            out[f"a_{h}x{w}"] = pa[-1]
            out[f"b_{h}x{w}"] = pb[-1][:, :, ::2, ::2]
    save("forward_odd", **out)


if __name__ == "__main__":
    main()
