import argparse, os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from prior_flow_amd import synthetic_pair, det_state_dict
from prior_flow_amd import train as tr
from prior_flow_amd.modules import state_dict_shapes
from prior_flow_amd.prior_raft import PriOr_RAFT
H, W, iters = 128, 256, 3
i1, i2 = (t.cuda() for t in synthetic_pair(1, H, W, seed=21))
gen = torch.Generator().manual_seed(6)
gt = (torch.rand(1, 2, H, W, generator=gen) * 6 - 3).cuda()
valid = torch.ones(1, H, W).cuda()
args = argparse.Namespace(lr=1e-4, wdecay=5e-5, epsilon=1e-8, num_steps=1000, clip=1.0)
def run(sink):
    os.environ["PRIORFLOW_GRAD_SINK"] = "1" if sink else "0"
    model = PriOr_RAFT(argparse.Namespace(mixed_precision=False, dropout=0.0))
    model.load_state_dict(det_state_dict(state_dict_shapes()), strict=True)
    model = model.cuda().train(); model.freeze_bn()
    opt, sched = tr.fetch_optimizer(args, model)
    crit = tr.uniform_loss(H, W)
    _, m = tr.train_step(model, opt, sched, crit, i1, i2, gt, valid, iters=iters, clip=args.clip)
    return opt.grad.detach().clone(), opt
a, _ = run(False); b, _ = run(False); c, opt = run(True); d, _ = run(True)
print("off vs off", float((a-b).norm()/a.norm()), "on vs on", float((c-d).norm()/c.norm()), "on vs off", float((a-c).norm()/a.norm()))
# per-parameter deviation on vs off
off = 0
worst = []
for p in opt.params:
    n = p.numel()
    da = a[off:off+n]; dc = c[off:off+n]
    worst.append((float((da-dc).norm()/(da.norm()+1e-12)), float(da.norm()), n)); off += n
names = [n for n, _ in opt_named] if False else None
import itertools
for i, wv in sorted(enumerate(worst), key=lambda t: -t[1][0])[:8]:
    print(i, wv)
