"""World-size-2 `gloo` tests (CPU) of the multi-GPU plumbing: pair sharding covers every pair
exactly once, results reassemble in order, replicas that were constructed with different seeds are
synchronised from rank 0 (what nn.DataParallel's per-step broadcast gives the reference), the flat
gradient all-reduce reproduces single-process summed gradients (reduce_add semantics,
train_flow.py:96), and diverged replicas are detected."""
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
        from prior_flow_amd import parallel
        from prior_flow_amd.parallel import gather_results, shard_indices, shard_seed

        # --- sharding of 7 pairs over 2 ranks, per-pair "result" = f(pair index)
        n_pairs = 7
        mine = shard_indices(n_pairs, rank, world)
        local = [torch.tensor([float(i * i)]) for i in mine]
        full = gather_results(local, n_pairs, rank, world)
        assert [float(t) for t in full] == [float(i * i) for i in range(n_pairs)]
        assert shard_seed(1234, rank) != shard_seed(1234, (rank + 1) % world)

        # --- replicas constructed with DIFFERENT seeds: flat buffers, broadcast from rank 0
        def make():
            return torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.BatchNorm2d(4), torch.nn.ReLU(),
                                       torch.nn.Conv2d(4, 2, 1))
        torch.manual_seed(100 + rank)
        net = make()
        net[1].running_mean.fill_(float(rank + 1))
        plist, flat, grad = parallel.flatten_parameters(net.parameters())
        assert all(p.data_ptr() >= flat.data_ptr() and p.grad.data_ptr() >= grad.data_ptr() for p in plist)
        try:
            parallel.assert_replicas_in_sync(flat)
            raise SystemExit("diverged replicas were not detected")
        except RuntimeError as e:
            assert "diverged" in str(e)
        parallel.sync_replicas(flat, list(net.buffers()))
        parallel.assert_replicas_in_sync(flat)
        torch.manual_seed(100)
        ref_net = make()                                      # what rank 0 constructed
        ref_net[1].running_mean.fill_(1.0)
        for a, b in zip(net.state_dict().values(), ref_net.state_dict().values()):
            assert torch.equal(a, b)

        # --- gradient all-reduce == gradients of the summed loss over the global batch
        net.eval(); ref_net.eval()                            # frozen BatchNorm, as in training (freeze_bn)
        data = torch.arange(2 * 3 * 8 * 8, dtype=torch.float32).reshape(2, 3, 8, 8) / 100.0
        loss = net(data[rank:rank + 1]).abs().sum()          # this rank's sample only
        loss.backward()
        assert all(p.grad.data_ptr() >= grad.data_ptr() for p in plist)      # autograd accumulated in place
        parallel.all_reduce_sum_(grad)
        got = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
        ref_net(data).abs().sum().backward()
        want = torch.cat([p.grad.reshape(-1) for p in ref_net.parameters()])
        assert torch.allclose(got, want, atol=1e-5), float((got - want).abs().max())
        assert torch.equal(got, grad)

        # --- re-binding behind the optimizer's back (zero_grad(set_to_none=True), .to()) is repaired
        net.zero_grad(set_to_none=True)
        net[0].weight.data = net[0].weight.data.clone() + 1.0
        assert parallel.realias(plist, flat, grad, keep_values=True) >= 2
        assert plist[0].data_ptr() == flat.data_ptr() and float(grad.abs().sum()) == 0.0
        assert torch.equal(flat[:plist[0].numel()].view_as(plist[0]), net[0].weight.data)
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


def test_bench_self_launch_refuses_without_enough_gpus():
    """`python bench.py --gpus N` launches its own ranks; the parent checks the visible device count without
    initialising the GPU and fails with a clear message (exit status 2) when there are too few."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64", "--steps", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "--gpus 64 but only" in r.stderr, (r.returncode, r.stderr[-300:])


def test_bench_launching_parent_never_loads_torch_or_hip(tmp_path):
    """The parent of `python bench.py --gpus N` counts GPUs from the visibility masks / the KFD topology and must not map torch
    or libamdhip64 (a process that initialised the GPU must not fork + exec on the pool's boxes).  Here the two gloo ranks die
    at once (no GPU in this container); what is checked is the parent's own list of mapped shared objects."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    maps = tmp_path / "maps.txt"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(HIP_VISIBLE_DEVICES="0", PRIORFLOW_BENCH_BACKEND="gloo", PRIORFLOW_BENCH_PARENT_MAPS=str(maps))
    subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--no-cpu-baseline"],
                   env=env, capture_output=True, text=True, timeout=600)
    mapped = maps.read_text()
    assert "libamdhip64" not in mapped and "libtorch" not in mapped and "libc" in mapped, mapped[-400:]


def test_bench_counts_gpus_without_hip(monkeypatch):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2")
    assert bench.visible_gpus() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpus() == 0
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    assert bench.visible_gpus() >= 0          # KFD topology (none in the build container)
