import argparse, sys, time, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'oracle'))
import torch
from prior_flow_amd import det_state_dict, synthetic_pair
from prior_flow_amd.prior_raft import PriOr_RAFT, state_dict_shapes
import priorflow_oracle as po
params = det_state_dict(state_dict_shapes())
m = PriOr_RAFT(argparse.Namespace(mixed_precision=False, dropout=0.0)); m.load_state_dict(params); m = m.cuda().eval()
# config 5 geometry: 640x1280, iters 6 (oracle on CPU is slow), vs oracle
i1, i2 = synthetic_pair(1, 640, 1280, seed=3)
with torch.no_grad():
    out = m(i1.cuda(), i2.cuda(), iters=6, test_mode=True)
torch.cuda.synchronize()
t=time.time(); ref = po.forward(params, i1, i2, iters=6, test_mode=True); print('oracle 640x1280 it6 %.1fs' % (time.time()-t))
e = po.epe(out.cpu(), ref); print('640x1280 iters=6: EPE mean %.3e max %.3e |flow| %.2f' % (e.mean(), e.max(), ref.abs().mean()))
# timing 640x1280 iters=32
with torch.no_grad():
    for _ in range(3): m(i1.cuda(), i2.cuda(), iters=32, test_mode=True)
    torch.cuda.synchronize(); t=time.time()
    for _ in range(5): m(i1.cuda(), i2.cuda(), iters=32, test_mode=True)
    torch.cuda.synchronize(); print('640x1280 iters=32: %.2f ms/pair' % ((time.time()-t)/5*1e3))
# batch sizes at 512x1024 iters 12
for B in (1, 4, 16, 32):
    j1, j2 = synthetic_pair(B, 512, 1024, seed=11)
    j1, j2 = j1.cuda(), j2.cuda()
    with torch.no_grad():
        for _ in range(2): o = m(j1, j2, iters=12, test_mode=True)
        torch.cuda.synchronize(); t=time.time(); n = 3
        for _ in range(n): o = m(j1, j2, iters=12, test_mode=True)
        torch.cuda.synchronize(); dt=(time.time()-t)/n
    print('B=%d: %.2f ms/step  %.1f pairs/s  mem %.1f GB finite=%s' % (B, dt*1e3, B/dt, torch.cuda.max_memory_allocated()/2**30, bool(torch.isfinite(o).all())))
    if B == 4:
        solo = m(j1[2:3], j2[2:3], iters=12, test_mode=True)
        print('  batch independence: EPE(solo, batched[2]) = %.2e' % po.epe(solo.cpu(), o[2:3].cpu()).mean())
