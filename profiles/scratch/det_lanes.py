"""[needs the -DPF_FO_DEBUG hooks of pf_flow_out_strip as of commit 6e5c0fe; the shipped kernel has none] Per-lane partial sums of pf_flow_out_strip inside the captured graph: which lanes deviate in a bad replay?"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from prior_flow_amd import det_state_dict, synthetic_pair
from prior_flow_amd.modules import state_dict_shapes
from prior_flow_amd.prior_raft import PriOr_RAFT
params = det_state_dict(state_dict_shapes())
m = PriOr_RAFT(argparse.Namespace(mixed_precision=False, dropout=0.0)); m.load_state_dict(params, strict=True)
m = m.cuda().eval(); m.use_streams, m.use_graph = True, True
dll = m._lib()._dll
buf = torch.zeros(128 * 64 * 8 + 256, device="cuda")
dbg = buf[:128 * 64 * 8].view(128, 64, 8)
ids = buf[128 * 64 * 8:].view(torch.int32).view(128, 2)
print("device", torch.cuda.get_device_properties(0).name, getattr(torch.cuda.get_device_properties(0), "uuid", None), os.uname().nodename)
dll.pf_debug_set_flow_out.argtypes = [ctypes.c_void_p]
assert dll.pf_debug_set_flow_out(buf.data_ptr()) == 0
i1, i2 = synthetic_pair(1, 128, 256); i1, i2 = i1.cuda(), i2.cuda()
with torch.no_grad():
    m(i1, i2, iters=1, test_mode=True); torch.cuda.synchronize()
    ws = next(iter(m._ws.values()))
    ref_d, ref_p = ws.delta_a.clone(), dbg.clone()
    for r in range(int(os.environ.get('RUNS', 150))):
        m(i1, i2, iters=1, test_mode=True); torch.cuda.synchronize()
        bad = torch.nonzero((ws.delta_a != ref_d).any(1)).flatten().tolist()
        dp = (dbg != ref_p)
        if bad or dp.any():
            print("run", r, "bad pixels", [(i // 32, i % 32) for i in bad])
            for st in sorted(set(torch.nonzero(dp)[:, 0].tolist())):
                h, x = int(ids[st, 0]) & 0xffffffff, int(ids[st, 1]) & 0xf
                lanes = sorted(set(torch.nonzero(dp[st])[:, 0].tolist())); js = sorted(set(torch.nonzero(dp[st])[:, 1].tolist()))
                print(f"   strip {st}: xcc {x} se {(h >> 13) & 7} sh {(h >> 12) & 1} cu {(h >> 8) & 15} simd {(h >> 4) & 3} wave {h & 15} | lanes {lanes[0]}..{lanes[-1]} sums {js}")
            for st, lane, j in []:
                print(f"   strip {st} (y {st // 8}, x0 {(st % 8) * 4}) lane {lane} sum j={j} (pixel +{j >> 1}, o={j & 1}): ref {float(ref_p[st, lane, j]):+.6f} now {float(dbg[st, lane, j]):+.6f}")
print("wave slots of the strip waves in the last replay:", torch.bincount(ids[:, 0] & 15, minlength=8).tolist())
print("done")
