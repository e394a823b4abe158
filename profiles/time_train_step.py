"""Times the product training step (prior_flow_amd.train.train_step: forward + backward on the HIP kernels,
[RCCL all-reduce of the flat gradient buffer], clip, fused AdamW) on synthetic data at the reference's training
crop (train_flow.py:217 default 384x512, iters=12), one pair per GPU as in BASELINE.json configs[3] (batch 8 on
8 GPUs).  Not the headline metric (bench.py is); numbers go to DESIGN.md section 6.

    python profiles/time_train_step.py [--size 384 512] [--batch 1] [--iters 12] [--steps 5]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 profiles/time_train_step.py
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, nargs=2, default=[384, 512])
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--sync-every-step", action="store_true", help="with --graph: read the loss on the host after every step (exposes the host's per-step work)")
    ap.add_argument("--graph", action="store_true", help="the step as captured HIP graph(s) (train.GraphedTrainStep; any number of ranks)")
    a = ap.parse_args()
    import torch.distributed as dist
    from prior_flow_amd import autograd as ag
    from prior_flow_amd import det_state_dict, synthetic_pair
    from prior_flow_amd import train as tr
    from prior_flow_amd.modules import state_dict_shapes
    from prior_flow_amd.parallel import shard_seed
    from prior_flow_amd.prior_raft import PriOr_RAFT
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    own_gpu = torch.cuda.device_count() >= world          # rehearsal on one card: every rank on cuda:0 over gloo (RCCL refuses that)
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) if own_gpu else 0)
    torch.cuda.set_device(dev)
    if world > 1:
        if own_gpu:
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    H, W = a.size
    model = PriOr_RAFT(argparse.Namespace(mixed_precision=False, dropout=0.0))
    model.load_state_dict(det_state_dict(state_dict_shapes()), strict=True)
    model = model.to(dev).train()
    model.freeze_bn()
    opt, sched = tr.fetch_optimizer(argparse.Namespace(lr=2e-5, wdecay=5e-5, epsilon=1e-8, num_steps=100000), model)
    i1, i2 = synthetic_pair(a.batch, H, W, seed=shard_seed(1234, rank))
    i1, i2 = i1.to(dev), i2.to(dev)
    gen = torch.Generator().manual_seed(shard_seed(99, rank))
    gt = (torch.rand(a.batch, 2, H, W, generator=gen) * 8 - 4).to(dev)
    valid = torch.ones(a.batch, H, W, device=dev)
    crit = tr.uniform_loss(H, W, device=dev)
    losses = []

    # one rank: one graph; more ranks: graph A, the eager all-reduce of the flat gradient buffer, graph B (train.GraphedTrainStep)
    graphed = tr.GraphedTrainStep(model, opt, sched, crit, iters=a.iters, clip=1.0, warmup=1) if a.graph else None

    ap_sync = a.sync_every_step

    def step():
        if graphed is not None:
            loss, m = graphed(i1, i2, gt, valid)
            # the captured step returns device tensors (overwritten by the next replay): keep a copy, read it after the timed
            # region -- the host then prepares step k+1 (input copies, three AdamW scalars) while the GPU runs step k
            losses.append(float(loss) if ap_sync else loss.detach().clone())
        else:
            loss, m = tr.train_step(model, opt, sched, crit, i1, i2, gt, valid, iters=a.iters, clip=1.0)
            losses.append(float(loss))

    for _ in range(a.warmup + (2 if graphed is not None else 0)):      # graphed: one eager step, the capture, one replay
        step()
    ag.STATS["hip"] = ag.STATS["torch"] = 0
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], device=dev)
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    if rank == 0:
        ms = float(dt) / a.steps * 1e3
        print(json.dumps({"metric": "training pairs/sec (forward+backward+AdamW)", "value": round(world * a.batch / (ms / 1e3), 3),
                          "unit": "pairs/s", "n_gpus": world, "ms_per_step": round(ms, 2), "steps": a.steps,
                          "config": {"workload": f"train_step {H}x{W} iters={a.iters} batch/GPU={a.batch}",
                                     "hip_graph": graphed is not None, "graphs": len(graphed.graphs) if graphed is not None else 0,
                                     "backend": dist.get_backend() if world > 1 else None, "ranks_share_one_gpu": not own_gpu},
                          "hip_launches_per_step": ag.STATS["hip"] // a.steps,
                          "torch_conv_launches_per_step": ag.STATS["torch"] // a.steps,
                          "peak_mem_GB": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2),
                          "loss_read": "every step" if (graphed is None or a.sync_every_step) else "after the timed steps",
                          "losses": [round(float(x), 3) for x in losses]}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
