import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
sys.dont_write_bytecode = False


def _run_two_rank_graphed():
    """Two ranks of the captured training step on this box's one card over gloo (tests/test_hip_train_step.py).  Started from
    pytest_configure, i.e. before anything in this process has initialised the GPU."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "run_train_2rank.py"), "--graphed"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=420, env=env, cwd=ROOT, start_new_session=True)
        return {"rc": p.returncode, "out": p.stdout.decode(errors="replace")}
    except subprocess.TimeoutExpired as e:
        return {"rc": -1, "out": "timeout after 420 s\n" + (e.stdout or b"").decode(errors="replace")}


def pytest_configure(config):
    config._pf_two_rank = None     # {"rc": int, "out": str} of tests/run_train_2rank.py --graphed, or None when it was not started
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    expr = (config.getoption("markexpr", "") or "").strip()
    if expr == "gpu" and not os.environ.get("PRIORFLOW_SKIP_2RANK") and not getattr(config.option, "collectonly", False):
        kexpr = (config.getoption("keyword", "") or "").strip()
        files = [str(a) for a in config.args]
        wanted = (not kexpr or "two_ranks" in kexpr or "graphed" in kexpr) and \
                 (all(not f.endswith(".py") for f in files) or any("test_hip_train_step" in f for f in files))
        # count the cards WITHOUT the HIP runtime (visibility masks, else the KFD topology -- bench.visible_gpus, whose module imports
        # nothing but the standard library): torch.cuda.device_count() falls back to hipGetDeviceCount where amdsmi is missing,
        # which initialises HSA in this process just before it spawns GPU children.  With two or more cards the
        # children take one each and the all-reduce runs over RCCL ("nccl"), else both share card 0 over gloo (run_train_2rank.py).
        try:
            from bench import visible_gpus
            ngpu = visible_gpus()
        except Exception:
            ngpu = 0
        if wanted and ngpu > 0:
            config._pf_two_rank = _run_two_rank_graphed()


def _has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
