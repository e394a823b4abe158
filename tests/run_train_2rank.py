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


def graphed_main():
    """--graphed: train.GraphedTrainStep with two ranks (graph A, eager all-reduce, graph B) against the eager two-rank train_step
    from the same weights on the same data: four steps each (the stepper's first one is its eager warm-up step); the first
    step's all-reduced gradient norm is the reference's."""
    import golden_cases as gc
    from gen_golden_train_step import step_inputs
    from prior_flow_amd import train as tr
    from prior_flow_amd.modules import state_dict_shapes
    from prior_flow_amd.prior_raft import PriOr_RAFT
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    own_gpu = torch.cuda.device_count() >= world
    dev = torch.device("cuda", rank if own_gpu else 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl" if own_gpu else "gloo", rank=rank, world_size=world)
    g = gc.load("train_step")
    i1, i2, gt, valid = (x[rank:rank + 1].to(dev) for x in step_inputs())
    args = argparse.Namespace(lr=1e-4, wdecay=5e-5, epsilon=1e-8, num_steps=1000)

    def run(graphed):
        model = PriOr_RAFT(argparse.Namespace(mixed_precision=False, dropout=0.0))
        model.load_state_dict(gc.det_state_dict(state_dict_shapes()), strict=True)
        model = model.to(dev).train()
        model.freeze_bn()
        opt, sched = tr.fetch_optimizer(args, model)
        crit = tr.uniform_loss(128, 256, device=dev)
        stepper = tr.GraphedTrainStep(model, opt, sched, crit, iters=3, clip=1.0, warmup=1) if graphed else None
        norms, losses = [], []
        for k in range(4):                   # graphed: one eager step (lazy initialisations), the capture + replay, two replays
            a = (i1 + float(k)).clamp(0, 255)
            if graphed:
                loss, m = stepper(a, i2, gt, valid)
            else:
                loss, m = tr.train_step(model, opt, sched, crit, a, i2, gt, valid, iters=3, clip=1.0)
            norms.append(float(m["grad_norm"]))
            losses.append(float(loss))
        opt.assert_in_sync()
        return norms, losses, opt.flat.detach().clone(), (len(stepper.graphs) if graphed else 0)

    n0, l0, p0, _ = run(False)
    n1, l1, p1, ngraphs = run(True)
    rel = float((p0 - p1).norm() / p0.norm())
    ref = float(g["grad_norm"])
    ok = (ngraphs == 2 and abs(n1[0] - ref) < 2e-3 * ref and all(abs(a - b) < 1e-3 * abs(a) for a, b in zip(n0, n1))
          and all(abs(a - b) < 1e-4 * abs(a) for a, b in zip(l0, l1)) and rel < 1e-5)
    print(f"rank {rank}/{world} backend={dist.get_backend()} device={dev}: graphed ({ngraphs} graphs) grad norms {[round(x, 3) for x in n1]} "
          f"vs eager {[round(x, 3) for x in n0]} (reference first step {ref:.3f}); local losses {[round(x, 4) for x in l1]} vs "
          f"{[round(x, 4) for x in l0]}; parameter deviation {rel:.2e} -> {'OK' if ok else 'MISMATCH'}", flush=True)
    flag = torch.tensor([1.0 if ok else 0.0], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0 and float(flag) == 1.0:
        print("graphed two-rank step: OK", flush=True)
    dist.destroy_process_group()
    sys.exit(0 if float(flag) == 1.0 else 1)


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
    graphed_main() if "--graphed" in sys.argv[1:] else main()
