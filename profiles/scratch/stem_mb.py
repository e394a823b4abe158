import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from prior_flow_amd import _lib
from prior_flow_amd.engine import pack_stem7x7, split_twin
lib = _lib.load(); dev = torch.device("cuda:0")
Bn = int(sys.argv[1]) if len(sys.argv) > 1 else 4
img = torch.rand(Bn, 3, 512, 1024, device=dev) * 2 - 1
w = pack_stem7x7((torch.rand(64, 3, 7, 7, device=dev) - 0.5) * 0.2); b = torch.rand(64, device=dev)
rows = Bn * 256 * 512
out = torch.empty(rows, 64, device=dev); tw = split_twin(rows, 64, dev)
part = torch.zeros(Bn * 512 * 64 * 2, dtype=torch.float64, device=dev)
for name, kw in (("fnet form (fp32 + stats)", dict(out=out, stats=part)), ("cnet form (fp32 + twin, relu)", dict(out=out, out_split=tw, relu=True)), ("twin only", dict(out_split=tw, relu=True))):
    for _ in range(3): lib.enc_stem(img, w, b, **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(30): lib.enc_stem(img, w, b, **kw)
    e.record(); torch.cuda.synchronize()
    print(f"{os.environ.get('PRIORFLOW_LIB','shipped')[-16:]:16s} Bn={Bn} {name:32s} {s.elapsed_time(e)/30*1e3:7.1f} us")
