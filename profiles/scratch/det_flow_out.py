"""pf_flow_head_out alone, repeated while another stream keeps the GPU busy: is its result reproducible?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from prior_flow_amd._lib import load
lib = load()
H8, W8 = int(os.environ.get("H8", 16)), int(os.environ.get("W8", 32))
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(H8 * W8, 256, device="cuda", generator=g)
w = torch.randn(2 * 9 * 256, device="cuda", generator=g) * 0.05
b = torch.randn(2, device="cuda", generator=g)
coords = torch.zeros(1, 2, H8, W8, device="cuda")
delta = torch.zeros(H8 * W8, 4, device="cuda")
busy_a = torch.randn(2048, 2048, device="cuda")
side = torch.cuda.Stream()
mode = os.environ.get("BUSY", "1")
outs = []
torch.cuda.synchronize()
for it in range(int(os.environ.get("RUNS", 300))):
    if mode == "1":
        with torch.cuda.stream(side):
            for _ in range(3):
                busy_b = busy_a @ busy_a
    lib.flow_head_out(x, 256, w, b, coords, delta)
    outs.append(delta.clone())
torch.cuda.synchronize()
bad = sum(int(not torch.equal(o, outs[0])) for o in outs)
print("busy", mode, f"{H8}x{W8}", bad, "of", len(outs), "launches differ from the first")
