"""Two-rank check of the data-parallel training step ("N-rank grads == single-process grads on the
concatenated batch").  Not collected by pytest (a GPU-initialised pytest process must not spawn GPU children on
the pool's boxes); launched directly:

    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tests/run_train_2rank.py

Rank r takes sample r of the reference's golden training step (tests/golden/train_step.npz: B=2, 128x256, iters=3),
runs the product forward + backward on the HIP kernels, and the ranks SUM-all-reduce the flat gradient buffer
exactly as prior_flow_amd.train.train_step does.  The reduced gradient must reproduce the reference's B=2 loss,
total gradient norm and per-parameter norms.  Backend: "gloo" on CUDA tensors when both ranks share one GPU (RCCL
refuses two ranks on one device), "nccl" (= RCCL over xGMI) when every rank has its own."""
import argparse
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)


def main():
    import golden_cases as gc
    from gen_golden_train_step import step_inputs
    from prior_flow_amd import train as tr
    from prior_flow_amd.modules import state_dict_shapes
    from prior_flow_amd.prior_raft import PriOr_RAFT
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    ngpu = torch.cuda.device_count()
    own_gpu = ngpu >= world
    dev = torch.device("cuda", rank if own_gpu else 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl" if own_gpu else "gloo", rank=rank, world_size=world)
    g = gc.load("train_step")
    model = PriOr_RAFT(argparse.Namespace(mixed_precision=False, dropout=0.0))
    model.load_state_dict(gc.det_state_dict(state_dict_shapes()), strict=True)
    model = model.to(dev).train()
    model.freeze_bn()
    if rank > 0:
        # replicas need not be constructed identically: fetch_optimizer broadcasts rank 0's weights and BatchNorm buffers
        # (the reference's DataParallel re-broadcasts its module every step, train_flow.py:96)
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.01 * rank)
            model.cnet.norm1.running_mean.add_(1.0)
    opt, sched = tr.fetch_optimizer(argparse.Namespace(lr=1e-4, wdecay=5e-5, epsilon=1e-8, num_steps=1000), model)
    i1, i2, gt, valid = (x[rank:rank + 1].to(dev) for x in step_inputs())       # this rank's pair only
    crit = tr.uniform_loss(128, 256, device=dev)
    opt.zero_grad()
    gt_b, valid_b = tr.rotate_gt(gt)
    pa, pb = model(i1, i2, iters=3)
    la, _ = crit(pa, gt, valid, 0.8)
    seeds = list(crit.grads)
    lb, _ = crit(pb, gt_b, valid_b, 0.8)
    seeds += list(crit.grads)
    torch.autograd.backward(list(pa) + list(pb), seeds)
    local_norm = opt.total_grad_norm()
    loss = (la + lb).reshape(1).clone()
    tr.parallel.all_reduce_sum_(opt.grad)                    # the one data-path collective of training
    dist.all_reduce(loss, op=dist.ReduceOp.SUM)
    total = opt.total_grad_norm()
    params = dict(model.named_parameters())
    want = dict(zip([str(n) for n in g["names"]], [float(v) for v in g["norms"]]))
    worst = max(abs(float(params[k].grad.double().norm()) - wn) / (wn + 1e-3 * float(g["grad_norm"])) for k, wn in want.items())
    ok = (abs(float(loss) - float(g["loss"])) < 2e-4 * float(g["loss"])
          and abs(total - float(g["grad_norm"])) < 2e-3 * float(g["grad_norm"]) and worst < 2e-2)
    print(f"rank {rank}/{world} backend={dist.get_backend()} device={dev}: local grad norm {local_norm:.4f}; all-reduced: "
          f"loss {float(loss):.4f} (reference B=2 step {float(g['loss']):.4f}), grad norm {total:.4f} "
          f"(reference {float(g['grad_norm']):.4f}), worst per-parameter norm deviation {worst:.2e} -> {'OK' if ok else 'MISMATCH'}",
          flush=True)
    # identical clip + AdamW on every rank keeps the replicas bit-identical
    tr.clip_grad_norm_(opt, 1.0)
    opt.step()
    try:
        opt.assert_in_sync()
        same = True
    except RuntimeError as exc:
        print(f"rank {rank}: {exc}", flush=True)
        same = False
    if rank == 0:
        print(f"parameter checksums after the step identical on all ranks: {same}", flush=True)
    dist.destroy_process_group()
    sys.exit(0 if ok and same else 1)


if __name__ == "__main__":
    main()
