"""Small captured graph: FlowHead.conv2 (pf_flow_head_out) on the capture stream beside the mask head's 1x1 conv on
a side stream.  Is pf_flow_head_out's result reproducible across replays?  CO=mask|flowb|none picks the co-runner."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from prior_flow_amd import det_state_dict, synthetic_pair
from prior_flow_amd.modules import state_dict_shapes
from prior_flow_amd.prior_raft import PriOr_RAFT
from prior_flow_amd._lib import EPI_LINEAR, EPI_RELU
params = det_state_dict(state_dict_shapes())
m = PriOr_RAFT(argparse.Namespace(mixed_precision=False, dropout=0.0)); m.load_state_dict(params, strict=True)
m = m.cuda().eval()
H, W = int(os.environ.get("H", 128)), int(os.environ.get("W", 256))
ws = m._workspace(1, H, W, torch.device("cuda"))
P = m._weights()
lib = m._lib()
g = torch.Generator(device="cuda").manual_seed(3)
if os.environ.get("DATA", "random") == "model":
    i1, i2 = synthetic_pair(1, H, W)
    with torch.no_grad():
        m.use_graph = False
        m(i1.cuda(), i2.cuda(), iters=1, test_mode=True)       # leaves real activations in the workspace
    ws.fh_b.copy_(ws.fh_a)
else:
    for t in (ws.fh_a, ws.fh_b, ws.mh_a, ws.net_a[0]):
        t.copy_(torch.randn(t.shape, device="cuda", generator=g).relu_())
co = os.environ.get("CO", "mask")
pre = os.environ.get("PRE", "1") == "1"
s2 = torch.cuda.Stream()
graph = torch.cuda.CUDAGraph()
def body():
    main = torch.cuda.current_stream()
    if pre:                                   # earlier fork/join on the same side stream
        s2.wait_stream(main)
        with torch.cuda.stream(s2):
            lib.flow_prep(ws.c1b, ws.flow_b, ws.flow2_b, 0, ws.x_b, 254)
        lib.flow_prep(ws.c1a, None, ws.flow4_a, 0, ws.x_a, 252)
        main.wait_stream(s2)
    ws.c1a.zero_()
    for _ in range(int(os.environ.get('PRELOAD', '0'))):      # keep the chip busy (clocks up) right before the pair
        lib.conv2d([P['a.fh1'].desc(ws.net_a[0], 0, 128, ws.fh_b, 0, EPI_RELU), P['a.m0'].desc(ws.net_a[0], 0, 128, ws.mh_b, 0, EPI_RELU)], 1, ws.H8, ws.W8, ws.x_a)
    if os.environ.get('HEADS', '0') == '1':      # the 3x3 head stems that produce fh_a / mh_a, as in the model
        lib.conv2d([P['a.fh1'].desc(ws.net_a[0], 0, 128, ws.fh_a, 0, EPI_RELU), P['a.m0'].desc(ws.net_a[0], 0, 128, ws.mh_a, 0, EPI_RELU)], 1, ws.H8, ws.W8, ws.x_a)
    if co != "none":
        s2.wait_stream(main)
        with torch.cuda.stream(s2):
            if co == "mask":
                lib.conv2d([P["a.m2"].desc(ws.mh_a, 0, 256, ws.mask_a, 0, EPI_LINEAR, scale=0.25)], 1, ws.H8, ws.W8, ws.x_a)
            else:
                lib.flow_head_out(ws.fh_b, 256, P["b.fh2w"], P["b.fh2b"], ws.c1b, ws.delta_b)
    lib.flow_head_out(ws.fh_a, 256, P["a.fh2w"], P["a.fh2b"], ws.c1a, ws.delta_a)
    if co != "none":
        main.wait_stream(s2)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    body()
torch.cuda.current_stream().wait_stream(side)
with torch.cuda.graph(graph):
    body()
buf = None
if hasattr(lib._dll, "pf_debug_set_flow_out"):
    import ctypes
    n = ws.H8 * ((ws.W8 + 3) // 4)
    buf = torch.zeros(n * 64 * 8 + 2 * n, device="cuda")
    lib._dll.pf_debug_set_flow_out.argtypes = [ctypes.c_void_p]
    lib._dll.pf_debug_set_flow_out(buf.data_ptr())
graph.replay(); torch.cuda.synchronize()
ref = ws.delta_a.clone()
bad = 0
runs = int(os.environ.get("RUNS", 400))
for r in range(runs):
    graph.replay(); torch.cuda.synchronize()
    bad += int(not torch.equal(ws.delta_a, ref))
if buf is not None:
    ids = buf[n * 64 * 8:].view(torch.int32).view(n, 2)[:, 0]
    print("wave slots of the strip waves in the last replay:", torch.bincount(ids & 15, minlength=8).tolist())
print(f"co-runner {co} pre-fork {pre} {H}x{W}: {bad} of {runs} replays differ")
