"""World-size-2 `gloo` tests (CPU) of the multi-GPU plumbing: pair sharding covers every pair
exactly once, results reassemble in order, and the flat gradient all-reduce reproduces
single-process summed gradients (nn.DataParallel's reduce_add semantics, train_flow.py:96)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        from prior_flow_amd.parallel import FlatGradAllReduce, gather_results, shard_indices, shard_seed

        # --- sharding of 7 pairs over 2 ranks, per-pair "result" = f(pair index)
        n_pairs = 7
        mine = shard_indices(n_pairs, rank, world)
        local = [torch.tensor([float(i * i)]) for i in mine]
        full = gather_results(local, n_pairs, rank, world)
        assert [float(t) for t in full] == [float(i * i) for i in range(n_pairs)]
        assert shard_seed(1234, rank) != shard_seed(1234, (rank + 1) % world)

        # --- gradient all-reduce == gradients of the summed loss over the global batch
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(4, 2, 1))
        data = torch.arange(2 * 3 * 8 * 8, dtype=torch.float32).reshape(2, 3, 8, 8) / 100.0
        loss = net(data[rank:rank + 1]).abs().sum()          # this rank's sample only
        loss.backward()
        FlatGradAllReduce(net.parameters())()
        got = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
        # single-process reference: loss summed over both samples
        ref_net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(4, 2, 1))
        ref_net.load_state_dict(net.state_dict())
        ref_net(data).abs().sum().backward()
        want = torch.cat([p.grad.reshape(-1) for p in ref_net.parameters()])
        assert torch.allclose(got, want, atol=1e-5), float((got - want).abs().max())
        ret[rank] = True
    finally:
        dist.destroy_process_group()


def test_world2_gloo():
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret.get(0) and ret.get(1)


def test_shard_indices_partition():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from prior_flow_amd.parallel import shard_indices
    for world in (1, 2, 4, 8):
        for n in (0, 1, 7, 8, 33):
            seen = sorted(i for r in range(world) for i in shard_indices(n, r, world))
            assert seen == list(range(n))
