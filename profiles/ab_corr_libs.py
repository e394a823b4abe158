#!/usr/bin/env python3
"""Same-process A/B of corr + pyramid builds compiled with different -D flags (timing-only diagnosis builds of the role-split
kernel, PF_RS_ABL / PF_RS_AHEAD ...): every variant is its own small shared object made from pf_corr_mfma.hip alone.
   python profiles/ab_corr_libs.py build name:FLAGS [name:FLAGS ...]     (here, no GPU: hipcc -> profiles/scratch/corr_<name>.so)
   python profiles/ab_corr_libs.py run [rounds] [reps] name[:ENV=VAL,...] [...]   (on the GPU box; `main` = the product library)
FLAGS: comma-separated macros, e.g. `onlystore:PF_RS_ABL=70`.  ENV: PRIORFLOW_CORR_* switches for that variant."""
import ctypes as C
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCR = os.path.join(ROOT, "profiles", "scratch")
SRC = os.path.join(ROOT, "prior-flow_amd", "csrc", "pf_corr_mfma.hip")


def so(name):
    return os.path.join(SCR, f"corr_{name}.so")


if sys.argv[1] == "build":
    os.makedirs(SCR, exist_ok=True)
    procs = []
    for v in sys.argv[2:]:
        name, _, flags = v.partition(":")
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared"] + \
              ["-D" + f for f in flags.split(",") if f] + [SRC, "-o", so(name)]
        procs.append((name, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for name, p in procs:
        out = p.communicate()[0].decode()
        print(name, "ok" if p.returncode == 0 else "FAILED\n" + out)
    sys.exit(0)

sys.path.insert(0, ROOT)
import torch
from prior_flow_amd import _lib

args = sys.argv[2:]
rounds, reps = int(args[0]), int(args[1])
variants = []
for v in args[2:]:
    name, _, env = v.partition(":")
    variants.append((v, name, dict(kv.split("=") for kv in env.split(",") if kv)))
lib = _lib.load()
dev = torch.device("cuda:0")
B, H8, W8, Cc = int(os.environ.get("MB_BATCH", "1")), int(os.environ.get("MB_H8", "64")), int(os.environ.get("MB_W8", "128")), 256
N = H8 * W8
g = torch.Generator().manual_seed(0)
f = [((torch.rand(B * N, Cc, generator=g) * 2 - 1)).to(dev) for _ in range(2)]
fs = [lib.split_bf16(x, torch.empty(B * N, Cc // 32, 2, 32, dtype=torch.bfloat16, device=dev)) for x in f]
# two volumes, alternated like the forward's A and B builds (a single 356 MB target partly lives in the 256 MB memory-side cache)
lvs = [[torch.empty(B * N, (H8 >> i) * (W8 >> i), device=dev) for i in range(4)] for _ in range(2)]
dlls = {}
for _, name, _e in variants:
    if name not in dlls:
        d = C.CDLL(_lib.LIB_PATH if name == "main" else so(name))
        d.pf_corr_pyramid_bf16x3.argtypes = [C.c_void_p] * 6 + [C.c_int] * 4 + [C.c_void_p]
        d.pf_corr_pyramid_bf16x3.restype = C.c_int
        dlls[name] = d


def launch(d, k):
    lv = lvs[k & 1]
    rc = d.pf_corr_pyramid_bf16x3(fs[0].data_ptr(), fs[1].data_ptr(), lv[0].data_ptr(), lv[1].data_ptr(), lv[2].data_ptr(),
                                  lv[3].data_ptr(), B, H8, W8, Cc, None)
    assert rc == 0, rc


mb = B * (4.0 * N * N * 85 / 64 + 2 * 4 * N * Cc) / 1e6
times = {v: [] for v, _, _ in variants}
for rnd in range(rounds + 1):
    for v, name, env in variants:
        for k in list(os.environ):
            if k.startswith("PRIORFLOW_CORR_"):
                del os.environ[k]
        for k, val in env.items():
            os.environ["PRIORFLOW_CORR_" + k] = val
        launch(dlls[name], 0); launch(dlls[name], 1)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for k in range(reps):
            launch(dlls[name], k)
        e.record()
        torch.cuda.synchronize()
        if rnd:
            times[v].append(s.elapsed_time(e) * 1e3 / reps)
for v, t in times.items():
    med, mn = statistics.median(t), min(t)
    print(f"{v:>28}: median {med:6.1f} us  min {mn:6.1f} us per launch   {mb / med:.3f} TB/s algorithmic = {mb / med / 8.0:.3f} of 8 TB/s")
