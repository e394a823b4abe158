"""Which workspace buffers differ between graph replays (race localisation)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from prior_flow_amd import det_state_dict, synthetic_pair
from prior_flow_amd.modules import state_dict_shapes
from prior_flow_amd.prior_raft import PriOr_RAFT
params = det_state_dict(state_dict_shapes())
m = PriOr_RAFT(argparse.Namespace(mixed_precision=False, dropout=0.0)); m.load_state_dict(params, strict=True)
m = m.cuda().eval(); m.use_streams, m.use_graph = True, True
iters = int(os.environ.get("ITERS", "2"))
i1, i2 = synthetic_pair(1, 128, 256); i1, i2 = i1.cuda(), i2.cuda()

def snap():
    ws = next(iter(m._ws.values()))
    out = {}
    for k, v in vars(ws).items():
        if isinstance(v, torch.Tensor): out[k] = v.clone()
        elif isinstance(v, (list, tuple)):
            for i, t in enumerate(v):
                if isinstance(t, torch.Tensor): out[f"{k}[{i}]"] = t.clone()
        elif isinstance(v, dict):
            for kk, t in v.items():
                if isinstance(t, torch.Tensor): out[f"{k}.{kk}"] = t.clone()
    return out

with torch.no_grad():
    m(i1, i2, iters=iters, test_mode=True); torch.cuda.synchronize()
    ref = snap()
    counts = {}
    for r in range(int(os.environ.get("RUNS", "20"))):
        m(i1, i2, iters=iters, test_mode=True); torch.cuda.synchronize()
        s = snap()
        bad = [k for k in ref if not torch.equal(torch.nan_to_num(ref[k]), torch.nan_to_num(s[k]))]
        for k in bad: counts[k] = counts.get(k, 0) + 1
print("iters", iters, "forks", os.environ.get("PRIORFLOW_FORKS"), "buffers that differed:", dict(sorted(counts.items())))
with torch.no_grad():
    # pattern of the differences in delta_a (rows = B*N, 4 floats per row)
    H8, W8 = 16, 32
    for r in range(6):
        m(i1, i2, iters=iters, test_mode=True); torch.cuda.synchronize()
        s = snap()
        d = (s["delta_a"][:, :2] - ref["delta_a"][:, :2]).abs().amax(1).view(H8, W8)
        e = (s["fh_a"] - ref["fh_a"]).abs().amax(1).view(H8, W8) if "fh_a" in s else None
        print("run", r, "max", float(d.max()), "fh diff", None if e is None else float(e.max()))
        P = m._weights()
        want = torch.nn.functional.conv2d(s["fh_a"].view(1, H8, W8, 256).permute(0, 3, 1, 2).double(),
                                          P["a.fh2w"].view(2, 3, 3, 256).permute(0, 3, 1, 2).double(), P["a.fh2b"].double(), padding=1)
        want = want[0].permute(1, 2, 0).reshape(-1, 2)
        for idx in torch.nonzero(d.flatten() > 0).flatten().tolist():
            print("  pixel y,x", idx // W8, idx % W8, "ref", ref["delta_a"][idx, :2].tolist(), "now", s["delta_a"][idx, :2].tolist(),
                  "fp64", want[idx].tolist(), "c1a-c0 now", (s["c1a"].view(2, -1)[:, idx] - torch.tensor([idx % W8, idx // W8], device="cuda")).tolist())
