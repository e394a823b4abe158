"""Import the *reference* PriOr-RAFT (read-only at /root/reference) on CPU.

TEST INFRASTRUCTURE ONLY, and only usable in the build container: the GPU box has
no /root/reference.  Used by ``oracle/gen_golden.py`` to emit the golden vectors
under ``tests/golden/`` (and by the other ``oracle/gen_golden_*.py`` generators).  Nothing from the
reference is copied into this repo; nothing under ``tests/`` imports this file.

Shims (all inert on the hot path):
  * empty stub modules for ``timm`` / ``omegaconf`` (imported, never used),
  * ``Tensor.cuda`` / ``Module.cuda`` -> identity (the reference hard-codes .cuda()),
  * cwd-independent sys.path entries for the reference's ``core`` package,
  * no bytecode is written into the reference tree.
"""
from __future__ import annotations

import argparse
import contextlib
import os
import sys
import types

REF_ROOT = "/root/reference/PriOr-RAFT"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "core"))


_loaded = {}


def load_reference():
    """Returns a namespace with the reference modules (cached)."""
    if _loaded:
        return _loaded["ns"]
    if not reference_available():
        raise RuntimeError("reference tree not present (expected only in the build container)")
    sys.dont_write_bytecode = True
    import torch

    for name in ("timm", "omegaconf"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            if name == "omegaconf":
                m.OmegaConf = type("OmegaConf", (), {})
                m.ListConfig = type("ListConfig", (), {})
            sys.modules[name] = m
    # .cuda() -> identity (CPU-only container)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self

    sys.path[:0] = [REF_ROOT]
    # the reference package is called ``core``; make sure ours never shadows it
    for k in [k for k in sys.modules if k == "core" or k.startswith("core.")]:
        del sys.modules[k]
    import importlib

    ns = types.SimpleNamespace()
    ns.prior_raft = importlib.import_module("core.prior_raft")
    ns.corr = importlib.import_module("core.corr")
    ns.update = importlib.import_module("core.update")
    ns.extractor = importlib.import_module("core.extractor")
    ns.utils = importlib.import_module("core.utils.utils")
    ns.proj = importlib.import_module("core.utils.projection_prim_ortho")
    ns.cyc = importlib.import_module("core.utils.my_cycle_sample")
    ns.args = lambda: argparse.Namespace(mixed_precision=False, dropout=0.0)
    _loaded["ns"] = ns
    return ns


@contextlib.contextmanager
def reference_model(state_fill):
    """Yields an eval-mode reference PriOr_RAFT whose weights come from ``state_fill(shapes)``."""
    import torch

    ns = load_reference()
    model = ns.prior_raft.PriOr_RAFT(ns.args())
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(state_fill(shapes), strict=True)
    model.eval()
    with torch.no_grad():
        yield model
