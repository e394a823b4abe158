"""Generate tests/golden/things_ckpt.npz: what the REFERENCE's ``load_things_ckpt``
(core/prior_raft.py:85-104) does with a RAFT-things-shaped checkpoint.

Run only in the build container:  ``python oracle/gen_golden_ckpt.py``.

The checkpoint is synthetic and is NOT stored: ``things_checkpoint()`` below rebuilds it from key names
(``module.``-prefixed fnet / cnet / update_block entries with the shapes of RAFT's BasicUpdateBlock, which
are the shapes of PriOr-RAFT's own ``update_block``; one entry with a wrong shape, one unknown entry and
one entry without the prefix exercise the skip branches).  The fixture holds, for every state_dict key of
the reference model after the call, the checkpoint key its tensor came from ("" = left at its initial
value) and the "Skip loading parameter" lines the reference printed.
"""
from __future__ import annotations

import contextlib
import io
import os
import sys
import tempfile

import numpy as np
import torch

sys.dont_write_bytecode = True
_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, _HERE)

import golden_cases as gc  # noqa: E402


def things_checkpoint(shapes):
    """shapes: {state_dict key: shape} of a PriOr_RAFT.  Returns the synthetic RAFT-things checkpoint."""
    ckpt = {}
    for key, shape in shapes.items():
        top = key.split(".")[0]
        if top == "ODDC":                       # RAFT has no ODDC: its counterparts live under update_block.*
            continue
        if key.endswith("num_batches_tracked"):
            ckpt["module." + key] = torch.tensor(1000 + len(ckpt), dtype=torch.long)     # distinct per entry
        else:
            ckpt["module." + key] = gc.uni("things/" + key, shape, -0.5, 0.5)
    # branches of the loader: wrong shape (both the plain key and its ODDC alias must be skipped) ...
    ckpt["module.update_block.mask.2.bias"] = gc.uni("things/badshape", (575,), -0.5, 0.5)
    # ... an entry the model does not have, and an entry without the DataParallel prefix (ignored)
    ckpt["module.update_block.encoder.extra.weight"] = gc.uni("things/extra", (3, 3), -0.5, 0.5)
    ckpt["fnet.conv1.bias"] = gc.uni("things/noprefix", tuple(shapes["fnet.conv1.bias"]), -0.5, 0.5)
    return ckpt


def main():
    from _refharness import load_reference
    ns = load_reference()
    torch.manual_seed(1234)
    model = ns.prior_raft.PriOr_RAFT(ns.args())
    before = {k: v.clone() for k, v in model.state_dict().items()}
    shapes = {k: tuple(v.shape) for k, v in before.items()}
    ckpt = things_checkpoint(shapes)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "raft-things.pth")
        torch.save(ckpt, path)
        out = io.StringIO()
        with contextlib.redirect_stdout(out):
            model.load_things_ckpt(path)
    after = model.state_dict()
    keys, source = [], []
    for k, v in after.items():
        hits = [ck for ck, cv in ckpt.items() if cv.shape == v.shape and torch.equal(cv, v)]
        if hits:
            assert len(hits) == 1, (k, hits)
            src = hits[0]
        else:
            assert torch.equal(v, before[k]), k
            src = ""
        keys.append(k)
        source.append(src)
    skipped = [line for line in out.getvalue().splitlines() if line.strip()]
    np.savez_compressed(os.path.join(gc.GOLDEN_DIR, "things_ckpt.npz"),
                        keys=np.array(keys), source=np.array(source), skipped=np.array(skipped))
    n_alias = sum(1 for k, s in zip(keys, source) if s and s != "module." + k)
    print(f"things_ckpt.npz: {len(keys)} keys, {sum(bool(s) for s in source)} loaded "
          f"({n_alias} through the ODDC -> update_block alias), {len(skipped)} skip lines")


if __name__ == "__main__":
    main()
