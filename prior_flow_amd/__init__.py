"""Importable alias for the hyphenated product directory ``prior-flow_amd/``.

The repo layout contract names the package directory ``prior-flow_amd/`` (not a
legal Python identifier), so this thin package points its ``__path__`` at that
directory; ``import prior_flow_amd.prior_raft`` resolves to
``prior-flow_amd/prior_raft.py``.
"""
import os as _os

_HERE = _os.path.dirname(_os.path.abspath(__file__))
PACKAGE_DIR = _os.path.join(_os.path.dirname(_HERE), "prior-flow_amd")
if not _os.path.isdir(PACKAGE_DIR):  # pragma: no cover
    raise ImportError(f"product directory missing: {PACKAGE_DIR}")
__path__ = [PACKAGE_DIR]

from .synthetic import det_state_dict, synthetic_pair  # noqa: E402,F401
