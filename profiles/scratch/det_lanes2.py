"""[needs the -DPF_FO_DEBUG hooks of pf_flow_out_strip as of commit 6e5c0fe; the shipped kernel has none] Per-lane, per-kernel-row state of pf_flow_out_strip inside the captured graph (diag build -DPF_FO_DEBUG=2)."""
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
dbg = torch.zeros(128, 64, 3, 56, device="cuda")
dll.pf_debug_set_flow_out.argtypes = [ctypes.c_void_p]
assert dll.pf_debug_set_flow_out(dbg.data_ptr()) == 0
i1, i2 = synthetic_pair(1, 128, 256); i1, i2 = i1.cuda(), i2.cuda()
names = [f"s{j}" for j in range(8)] + [f"v{i}.{c}" for i in range(6) for c in "xyzw"] + \
        [f"w0[{k}].{c}" for k in range(3) for c in "xyzw"] + [f"w1[{k}].{c}" for k in range(3) for c in "xyzw"]
shown = 0
with torch.no_grad():
    m(i1, i2, iters=1, test_mode=True); torch.cuda.synchronize()
    ws = next(iter(m._ws.values()))
    ref_d, ref_p = ws.delta_a.clone(), dbg.clone()
    for r in range(60):
        m(i1, i2, iters=1, test_mode=True); torch.cuda.synchronize()
        bad = torch.nonzero((ws.delta_a != ref_d).any(1)).flatten().tolist()
        dp = (dbg != ref_p)
        if bad or dp.any():
            print("run", r, "bad pixels", [(i // 32, i % 32) for i in bad])
            idx = torch.nonzero(dp)
            for st in sorted(set(idx[:, 0].tolist())):
                sub = idx[idx[:, 0] == st]
                lanes = sorted(set(sub[:, 1].tolist()))
                print(f"  strip {st} (y {st // 8}, x0 {(st % 8) * 4}) lanes {lanes}")
                for ky in range(3):
                    fields = sorted(set(sub[sub[:, 2] == ky][:, 3].tolist()))
                    print(f"     ky={ky}: fields that differ:", [names[f] for f in fields])
            shown += 1
            if shown >= 4: break
print("done")
