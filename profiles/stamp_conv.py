#!/usr/bin/env python3
"""Per-step timing INSIDE one workgroup of the halo conv kernel (diagnostic build, -DPF_STAMPS):
    hipcc ... -DPF_STAMPS -o prior-flow_amd/lib/diag/STAMPS.so
    PRIORFLOW_LIB=$PWD/prior-flow_amd/lib/diag/STAMPS.so python profiles/stamp_conv.py [zr|q|fh1|c2]
Every wave of workgroup 0 stamps s_memtime twice per K-step: A = just before the step's barrier (all of
the previous step's instructions issued), B = just after it.  B(s) - A(s) = wait for outstanding LDS ops +
barrier; A(s+1) - B(s) = the issue phase of step s (MFMAs + staging).  Prints both per step (max over waves)."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
which = sys.argv[1] if len(sys.argv) > 1 else "zr"
env = dict(os.environ, STAMP_CHILD="1")
if os.environ.get("STAMP_CHILD") != "1":
    # run the launches in this process through microbench_conv's setup (imported as a module would re-parse argv)
    sys.argv = [sys.argv[0], "3", which]
    os.environ["STAMP_CHILD"] = "1"
    src = open(os.path.join(ROOT, "profiles", "microbench_conv.py")).read()
    try:
        exec(compile(src, "microbench_conv.py", "exec"), {"__name__": "__main__", "__file__": os.path.join(ROOT, "profiles", "microbench_conv.py")})
    except SystemExit:
        pass
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from prior_flow_amd import _lib
    torch.cuda.synchronize()
    h = _lib.load()._dll if hasattr(_lib.load(), "_dll") else ctypes.CDLL(os.environ["PRIORFLOW_LIB"])
    buf = (ctypes.c_ulonglong * (8 * 40 * 8))()
    rc = h.pf_conv_read_stamps(buf)
    assert rc == 0, rc
    t = np.frombuffer(buf, dtype=np.uint64).reshape(8, 40, 8).astype(np.int64)
    nsteps = 38
    if os.environ.get("STAMP_SIMPLE", "0") == "1":      # barrier stamps only (the role-specialised kernel): issue phase and wait per wave
        print(f"{which}: per step and wave: issue = cycles from the barrier exit to the next barrier arrival, wait = cycles in the barrier")
        for s in range(1, nsteps):
            issue = t[:, s + 1, 0] - t[:, s, 1]
            wait = t[:, s + 1, 1] - t[:, s + 1, 0]
            print(f"step {s:2d}  issue: " + " ".join(f"{int(x):5d}" for x in issue) + "   wait: " + " ".join(f"{int(x):5d}" for x in wait)
                  + f"   step {int(t[:, s + 1, 1].max() - t[:, s, 1].max()):5d}"
                  + "   loaders: ring slot written at " + " ".join(f"{int(t[w, s, 2] - t[w, s, 1]):4d}" for w in range(4, 8))
                  + ", loads issued at " + " ".join(f"{int(t[w, s, 3] - t[w, s, 1]):4d}" for w in range(4, 8)))
        sys.exit(0)
    print(f"{which}: s_memtime ticks, workgroup 0; per step and wave: cycles from the barrier exit (B) to")
    print("   m0 = after gap 2 (3 MFMAs issued) | m1 = after gap 3 (ring piece: start = M3 issued, stored = slot s+2 written from registers, then addresses + loads of s+4) | m2 = after gap 6 | m3 = after gap 9 | A = all issued | next B")
    for s in range(1, nsteps):
        print(f"step {s}")
        for w in range(8):
            b = t[w, s, 1]
            m = [int(t[w, s, 2 + k] - b) for k in range(4)]
            r = [int(t[w, s, 6 + k] - b) for k in range(2)]
            print(f"   wave {w}: m0 {m[0]:5d}  [ring piece: start {r[0]:5d} stored {r[1]:5d}]  m1 {m[1]:5d}  m2 {m[2]:5d}  m3 {m[3]:5d}  A {int(t[w, s + 1, 0] - b):5d}  nextB {int(t[w, s + 1, 1] - b):5d}")
