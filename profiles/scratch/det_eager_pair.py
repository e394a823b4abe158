"""Eager (no graph): pf_flow_head_out at 16x32 on one stream while the mask head's 1x1 MFMA conv runs at SIZE on another."""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from prior_flow_amd import det_state_dict
from prior_flow_amd.modules import state_dict_shapes
from prior_flow_amd.prior_raft import PriOr_RAFT
from prior_flow_amd._lib import EPI_LINEAR
m = PriOr_RAFT(argparse.Namespace(mixed_precision=False, dropout=0.0)); m.load_state_dict(det_state_dict(state_dict_shapes()), strict=True)
m = m.cuda().eval()
lib, P = m._lib(), m._weights()
H8, W8 = 16, 32
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(H8 * W8, 256, device="cuda", generator=g).relu_()
coords = torch.zeros(1, 2, H8, W8, device="cuda")
delta = torch.zeros(H8 * W8, 4, device="cuda")
CH, CW = [int(v) for v in os.environ.get("SIZE", "64x128").split("x")]
mh = torch.randn(CH * CW, 256, device="cuda", generator=g).relu_()
mask = torch.zeros(CH * CW, 576, device="cuda")
n = H8 * ((W8 + 3) // 4)
buf = torch.zeros(n * 64 * 8 + 2 * n, device="cuda")
if hasattr(lib._dll, "pf_debug_set_flow_out"):
    lib._dll.pf_debug_set_flow_out.argtypes = [ctypes.c_void_p]
    lib._dll.pf_debug_set_flow_out(buf.data_ptr())
side = torch.cuda.Stream()
lib.flow_head_out(x, 256, P["a.fh2w"], P["a.fh2b"], coords, delta); torch.cuda.synchronize()
ref = delta.clone()
bad, slots = 0, torch.zeros(8, dtype=torch.long, device="cuda")
runs = int(os.environ.get("RUNS", 2000))
desc = [P["a.m2"].desc(mh, 0, 256, mask, 0, EPI_LINEAR, scale=0.25)]
for it in range(runs):
    with torch.cuda.stream(side):
        lib.conv2d(desc, 1, CH, CW, mh)
    lib.flow_head_out(x, 256, P["a.fh2w"], P["a.fh2b"], coords, delta)
    bad += int(not torch.equal(delta, ref))            # syncs
    slots += torch.bincount(buf[n * 64 * 8:].view(torch.int32).view(n, 2)[:, 0] & 15, minlength=8)
print(f"co-runner conv {CH}x{CW}: {bad} of {runs} launches differ; strip-wave slots {slots.tolist()}")
