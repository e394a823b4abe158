"""Micro-benchmark of the three 7x7 flow stems of an iteration (pf_conv2d_direct_group, 2 -> 128 at 64x128)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from prior_flow_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
B, H8, W8 = 1, 64, 128
rows = B * H8 * W8
g = torch.Generator().manual_seed(0)
x4 = torch.randn(rows, 4, generator=g).to(dev); x2 = torch.randn(rows, 2, generator=g).to(dev)
ws = [torch.randn(49, 2, 128, generator=g).to(dev) * 0.1 for _ in range(3)]
bs = [torch.randn(128, generator=g).to(dev) for _ in range(3)]
outs = [torch.empty(rows, 128, device=dev) for _ in range(3)]
probs = [(x4, 0, ws[0], bs[0], outs[0], 0), (x4, 2, ws[1], bs[1], outs[1], 0), (x2, 0, ws[2], bs[2], outs[2], 0)]
fn = lambda: lib.conv2d_direct_group(probs, 2, 128, 7, 7, True, B, H8, W8)
for _ in range(5): fn()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(200): fn()
e.record(); torch.cuda.synchronize()
# reference value check against torch conv (fp32)
xin = x4[:, :2].reshape(B, H8, W8, 2).permute(0, 3, 1, 2)
ref = torch.relu(torch.nn.functional.conv2d(xin, ws[0].reshape(7, 7, 2, 128).permute(3, 2, 0, 1), bs[0], padding=3))
err = float((outs[0].reshape(B, H8, W8, 128).permute(0, 3, 1, 2) - ref).abs().max())
print(f"stems x3: {s.elapsed_time(e) * 1e3 / 200:.1f} us/launch   max err vs torch conv {err:.2e}")
