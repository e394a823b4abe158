#!/usr/bin/env python3
"""bench.py -- PriOr-RAFT inference throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one PriOr_RAFT forward (iters=12, test_mode=True) over one resident batch of
synthetic 512x1024 ERP pairs per GPU (BASELINE.json configs[1]: one pair).  Ranks are
independent (the path shards over image pairs; no data-path collective) => weak scaling;
`value` = pairs processed by all ranks / max-over-ranks wall time of the K timed steps.

Multi-GPU: one process per GPU over torch.distributed ("nccl" = RCCL).  Either the caller provides the
ranks (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`: RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* come from the environment), or `python bench.py --gpus N` alone launches them itself:
the parent makes NO GPU call, starts N fresh child processes of this script with that environment on
127.0.0.1, relays rank 0's JSON line and exits with the worst child status.

Extra objects on the JSON line:
  roofline      the dominant kernel (MFMA implicit-GEMM conv of the update blocks): algorithmic FLOPs of
                its launches in one forward / their HIP-event time, vs the dense bf16 MFMA peak (bf16x3
                mode: 3 MFMA FLOPs per algorithmic FLOP, reported as mfma_pipe_util) or the fp32 MFMA peak;
  roofline_corr the fused correlation-volume + pyramid build (north_star's HBM target):
                algorithmic bytes / HIP-event time vs 8 TB/s;
  cpu_baseline  the CPU oracle (oracle/priorflow_oracle.py, the checker -- never the product)
                timed on the host cores on the same pair, rank 0 at N=1 only, plus the EPE of
                the GPU flow against it (`parity`);
  fp32_exact    the same forward in exact-fp32 MFMA arithmetic (the strict reference point beside the
                bf16x3 headline): pairs/s and EPE vs the oracle, rank 0 at N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

H, W, ITERS = 512, 1024, 12
PMC_FILE = "r2_pmc_traffic.json"   # rocprofv3 --pmc passes of this command (profiles/pmc_traffic.py); not re-measured per run
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: bf16 MFMA, dense (not the 2:1-sparse figure)
PEAK_HBM_GBPS = 8000.0            # MI355X_MICROARCH.md: HBM3E spec
TILE_NAMES = {0: "pf_conv_mfma_kernel<4,1,1> (128x32)", 1: "pf_conv_mfma_kernel<2,2,1> (64x64)",
              2: "pf_conv_mfma_kernel<2,2,2> (64x128)", 3: "pf_conv_halo_kernel<1> (128x64)",
              4: "pf_conv_halo_kernel<2> (128x128)"}


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def usable_cpus() -> int:
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota
    (os.cpu_count() reports the host's cores even inside a quota-limited container)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, int(quota / period + 0.5)))
        except Exception:
            pass
    return max(1, n)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1, help="pairs per GPU per step (configs[1] = 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--precision", choices=["bf16x3", "fp32"], default=None,
                    help="update-block conv arithmetic (default: PRIORFLOW_PRECISION or bf16x3)")
    return ap.parse_args()


def build_model(device):
    from prior_flow_amd import det_state_dict
    from prior_flow_amd.prior_raft import PriOr_RAFT, state_dict_shapes
    params = det_state_dict(state_dict_shapes())
    model = PriOr_RAFT(argparse.Namespace(mixed_precision=False, dropout=0.0))
    model.load_state_dict(params, strict=True)
    return model.to(device).eval(), params


def pmc_traffic(kernel_substr):
    """HBM-side bytes per launch of a kernel from the committed PMC passes (profiles/pmc_traffic.py:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, FETCH_SIZE doubled on gfx950, KiB -> bytes).
    Counters cannot be collected inside this process; None when the profile file is absent."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", PMC_FILE)
    try:
        with open(path) as f:
            kernels = json.load(f)["kernels"]
    except (OSError, ValueError, KeyError):
        return None
    for name, v in kernels.items():
        if kernel_substr in name:
            return round(v["hbm_bytes_per_launch"])
    return None


def profile_kernels(model, i1, i2):
    """One eager forward with HIP events around every MFMA-conv and corr-build launch."""
    lib = model._lib()
    recs = []          # (kind, tile, work, start_event, end_event)
    orig_conv, orig_corr, orig_corr3 = lib.conv2d, lib.corr_pyramid, lib.corr_pyramid_bf16x3

    def ev():
        return torch.cuda.Event(enable_timing=True)

    def conv2d(descs, B, H8, W8, like):
        flops = sum(2.0 * B * H8 * W8 * d.cout * d.kh * d.kw * (d.c0 + d.c1) for d in descs)
        tile = lib.conv2d_tile(descs, B, H8, W8)
        d0 = descs[0]
        roles = lib.conv2d_roles(descs, B, H8, W8) if tile in (3, 4) else 0
        if roles >= 16:    # pre-split operands: the all-DMA kernel, <NT, KH, KW, WN> exactly as rocprof names it
            r = roles - 16
            tile = "pf_conv_dma_kernel<%d, %d, %d, %d>" % (2 if (tile == 4 or r == 2) else 1, d0.kh, d0.kw, 3 - r)
        elif roles:        # role-specialised waves: <NT, KH, KW, WN> exactly as rocprof names it
            tile = "pf_conv_ws_kernel<%d, %d, %d, %d>" % (2 if (tile == 4 or roles == 2) else 1, d0.kh, d0.kw, 3 - roles)
        elif tile >= 3:    # halo kernel: the instantiation is <NT, KH, KW, AFFINE, TH> exactly as rocprof names it
            tile = "pf_conv_halo_kernel<%d, %d, %d, %s, %d>" % (
                1 if tile == 3 else 2, d0.kh, d0.kw, "true" if d0.in_scale else "false", 8 if tile == 5 else 4)
        else:
            tile = TILE_NAMES[tile].split(" ")[0][:-1] + (", true>" if d0.precision == 1 else ", false>")
        s, e = ev(), ev()
        s.record()
        orig_conv(descs, B, H8, W8, like)
        e.record()
        recs.append(("conv", tile, flops, s, e))

    def corr_pyramid(f1, f2, levels, B, H8, W8):
        n, c = H8 * W8, f1.shape[-1]
        nbytes = B * (4.0 * n * n * 85.0 / 64.0 + 2.0 * 4.0 * n * c)
        s, e = ev(), ev()
        s.record()
        orig_corr(f1, f2, levels, B, H8, W8)
        e.record()
        recs.append(("corr", -1, nbytes, s, e))

    def corr_pyramid_bf16x3(f1s, f2s, levels, B, H8, W8, c):
        n = H8 * W8
        nbytes = B * (4.0 * n * n * 85.0 / 64.0 + 2.0 * 4.0 * n * c)
        s, e = ev(), ev()
        s.record()
        orig_corr3(f1s, f2s, levels, B, H8, W8, c)
        e.record()
        recs.append(("corr", -1, nbytes, s, e))

    lib.conv2d, lib.corr_pyramid, lib.corr_pyramid_bf16x3 = conv2d, corr_pyramid, corr_pyramid_bf16x3
    was, was_streams = model.use_graph, model.use_streams
    model.use_graph = False
    model.use_streams = False          # one stream: an event interval must contain exactly one kernel
    # an event pair also times the dispatch gap in front of the kernel: calibrate it on empty pairs
    # (same parked-queue conditions) and subtract it, so the averages agree with rocprofv3's kernel durations
    torch.cuda._sleep(int(10e6))
    cal = [(ev(), ev()) for _ in range(64)]
    tiny = torch.zeros(64, device=i1.device)
    for s, e in cal:
        s.record(); tiny.add_(1.0); e.record()       # a ~2 us kernel: interval = gap + tiny kernel
    torch.cuda.synchronize()
    gaps = sorted(s.elapsed_time(e) for s, e in cal)
    gap_ms = max(0.0, gaps[len(gaps) // 2] - 0.002)
    try:
        with torch.no_grad():
            for _ in range(2):          # second pass is the measured one (caches warm)
                recs.clear()
                # park the GPU for ~15 ms so the host enqueues the whole eager forward ahead of it:
                # the HIP-event intervals then contain kernel time only, not host launch gaps
                torch.cuda._sleep(int(30e6))
                model(i1, i2, iters=ITERS, test_mode=True)
                torch.cuda.synchronize()
    finally:
        lib.conv2d, lib.corr_pyramid, lib.corr_pyramid_bf16x3 = orig_conv, orig_corr, orig_corr3
        model.use_graph, model.use_streams = was, was_streams
    by_tile = {}
    corr_t, corr_b, corr_n = 0.0, 0.0, 0
    for kind, tile, work, s, e in recs:
        ms = max(s.elapsed_time(e) - gap_ms, 1e-4)
        if kind == "conv":
            t = by_tile.setdefault(tile, [0.0, 0.0, 0])
            t[0] += work; t[1] += ms; t[2] += 1
        else:
            corr_b += work; corr_t += ms; corr_n += 1
    dom = max(by_tile, key=lambda k: by_tile[k][1])
    fl, ms, n = by_tile[dom]
    all_fl = sum(v[0] for v in by_tile.values())
    all_ms = sum(v[1] for v in by_tile.values())
    achieved = fl / (ms * 1e-3) / 1e12
    from prior_flow_amd._lib import PREC_BF16X3
    split = model._weights()["precision"] == PREC_BF16X3
    peak = PEAK_BF16_MFMA_TFLOPS if split else PEAK_F32_MFMA_TFLOPS
    roofline = {"kernel": str(dom) + (" bf16x3" if split else " fp32"), "bound": "mfma",
                "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4),
                # the 3-pass split issues 3 bf16 MFMA FLOPs per algorithmic FLOP: pipe utilisation
                "mfma_issue_tflops": round(achieved * (3 if split else 1), 2),
                "mfma_pipe_util": round(achieved * (3 if split else 1) / peak, 4),
                "traffic": pmc_traffic(str(dom)), "traffic_note": "HBM-side bytes/launch from the committed PMC passes (profiles/%s): counters cannot be read inside this process" % PMC_FILE,
                "launches_per_forward": n, "avg_launch_us": round(ms / n * 1e3, 1),
                "event_gap_us_subtracted": round(gap_ms * 1e3, 2),
                "gflop_per_forward": round(fl / 1e9, 1),
                "all_conv_kernels": {"gflop": round(all_fl / 1e9, 1), "ms": round(all_ms, 3),
                                     "tflops": round(all_fl / (all_ms * 1e-3) / 1e12, 2)}}
    gbps = corr_b / (corr_t * 1e-3) / 1e9
    roofline_corr = {"kernel": "pf_corr_kernel<fused pool, %s> (corr volume + 4-level pyramid)" % ("bf16x3" if split else "fp32"),
                     "bound": "hbm",
                     "achieved": round(gbps, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                     "frac": round(gbps / PEAK_HBM_GBPS, 4), "traffic": pmc_traffic("pf_corr_kernel"),
                     "traffic_note": "HBM-side bytes/launch from the committed PMC passes (profiles/%s)" % PMC_FILE,
                     "launches_per_forward": corr_n, "avg_launch_us": round(corr_t / corr_n * 1e3, 1),
                     "mb_per_launch": round(corr_b / corr_n / 1e6, 1),
                     "note": ("34.4 GFLOP/launch: 3-pass bf16 MFMA + 373 MB of once-written output" if split else
                              "exact-fp32 MFMA makes this kernel compute-bound (34.4 GFLOP per launch)")}
    return roofline, roofline_corr


def cpu_baseline(params, i1, i2, flow_gpu):
    import priorflow_oracle as po
    cores = usable_cpus()
    torch.set_num_threads(cores)
    log(f"cpu baseline: os.cpu_count()={os.cpu_count()} usable={cores} torch threads={torch.get_num_threads()}")
    t0 = time.time()
    ref = po.forward(params, i1, i2, iters=ITERS, test_mode=True)        # warm-up + parity
    first = time.time() - t0
    log(f"cpu baseline: first oracle forward {first:.1f}s")
    times = []
    budget = 25.0 - first
    while len(times) < 2 and (not times or budget > times[-1]):
        t0 = time.time()
        po.forward(params, i1, i2, iters=ITERS, test_mode=True)
        times.append(time.time() - t0)
        budget -= times[-1]
    best = min(times) if times else first
    epe = po.epe(flow_gpu.cpu(), ref)
    cb = {"value": round(i1.shape[0] / best, 4), "unit": "frame-pairs/s", "cores": cores,
          "kind": "port", "sample": f"{1 + len(times)} forwards of the same {i1.shape[0]}x{H}x{W} pair, "
                                      f"iters={ITERS}, torch CPU threads={cores}, best of the timed ones",
          "seconds_per_pair": round(best / i1.shape[0], 3)}
    cb["reference_note"] = ("the reference itself (imported on CPU in the build container, 8 cores): 7.56 s/pair = 0.132 pairs/s "
                            "(BASELINE.md section 2); this line times the oracle restatement on this host")
    parity = {"epe_mean": float(epe.mean()), "epe_max": float(epe.max()), "bar": 1e-3,
              "flow_mean_abs": float(ref.abs().mean()),
              "weights": "deterministic synthetic weights with the statistics of a freshly initialised model (no checkpoints "
                         "offline); the bar is on the mean, isolated pixels reach 1e-2 in sweeps (profiles/r1_parity_sweep.txt)"}
    return cb, parity, ref


def fp32_reference_point(params, device, i1, i2, ref_cpu, steps=5):
    """The strict reference point beside the bf16x3 headline: the same forward in exact-fp32 MFMA arithmetic
    (model.precision = PREC_F32; encoders and update blocks on the HIP library), pairs/s and EPE vs the CPU oracle."""
    from prior_flow_amd._lib import PREC_F32
    from prior_flow_amd.prior_raft import PriOr_RAFT
    import priorflow_oracle as po
    m = PriOr_RAFT(argparse.Namespace(mixed_precision=False, dropout=0.0))
    m.load_state_dict(params, strict=True)
    m = m.to(device).eval()
    m.precision = PREC_F32
    with torch.no_grad():
        for _ in range(2):
            flow = m(i1, i2, iters=ITERS, test_mode=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            flow = m(i1, i2, iters=ITERS, test_mode=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    out = {"dtype": "f32 (v_mfma_f32_32x32x2_f32, exact)", "value": round(i1.shape[0] * steps / dt, 3), "unit": "frame-pairs/s",
           "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps}
    if ref_cpu is not None:
        epe = po.epe(flow.cpu(), ref_cpu)
        out["epe_mean"], out["epe_max"] = float(epe.mean()), float(epe.max())
    return out


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (fresh processes; this parent
    never touches the GPU -- torch.cuda.device_count() does not initialise it) and relay their output."""
    import socket
    import subprocess
    n = args.gpus
    have = torch.cuda.device_count()
    rehearsal = os.environ.get("PRIORFLOW_BENCH_BACKEND", "nccl") != "nccl"     # gloo: ranks may share a card
    if have < n and not (rehearsal and have >= 1):
        print(f"bench.py: --gpus {n} but only {have} GPU(s) are visible", file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    codes = [p.wait() for p in procs]
    return max(abs(c) for c in codes)


def init_ranks(world: int, rank: int, device):
    """torch.distributed for the barrier / max-over-ranks of the timing (the forward itself has no collective).
    RCCL first; PRIORFLOW_BENCH_BACKEND=gloo selects the CPU backend for rehearsals on a box without N GPUs."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = os.environ.get("PRIORFLOW_BENCH_BACKEND", "nccl")       # nccl == RCCL on ROCm
    dist.init_process_group(backend, rank=rank, world_size=world)
    probe = torch.ones(1, device=device if backend == "nccl" else "cpu")
    dist.all_reduce(probe)                                             # first collective: builds the communicator
    assert int(probe.item()) == world, f"all-reduce over {world} ranks returned {probe.item()}"
    return dist, backend


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but the launcher provides WORLD_SIZE={world}; running {world} rank(s)")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    local = local % max(torch.cuda.device_count(), 1)        # gloo rehearsal of N ranks on one card
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist, backend = None, None
    if world > 1:
        dist, backend = init_ranks(world, rank, device)

    log(f"rank {rank}/{world} on {torch.cuda.get_device_name(local)}; building model")
    model, params = build_model(device)
    if args.precision is not None:
        from prior_flow_amd._lib import PREC_BF16X3, PREC_F32
        model.precision = PREC_F32 if args.precision == "fp32" else PREC_BF16X3
    if args.no_graph:
        model.use_graph = False
    from prior_flow_amd import synthetic_pair
    from prior_flow_amd.parallel import shard_seed
    # every rank owns different pairs (weak scaling; no data-path collective)
    i1c, i2c = synthetic_pair(args.batch, H, W, seed=shard_seed(1234, rank))
    i1, i2 = i1c.to(device), i2c.to(device)

    def rank_barrier():
        # RCCL's barrier is an all-reduce on a device tensor: name the device, or it guesses from the global rank
        if backend == "nccl":
            dist.barrier(device_ids=[local])
        else:
            dist.barrier()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            rank_barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for w in range(max(args.warmup, 1)):
            flow = model(i1, i2, iters=ITERS, test_mode=True)
            torch.cuda.synchronize()
            log(f"warm-up {w} done")
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            flow = model(i1, i2, iters=ITERS, test_mode=True)
        barrier()
        elapsed = time.perf_counter() - t0
    log(f"timed {args.steps} steps in {elapsed:.3f}s")
    if dist is not None:
        t = torch.tensor([elapsed], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        world = dist.get_world_size()                        # the size the communicator really has

    result = None
    if rank == 0:
        pairs = args.batch * world * args.steps
        result = {
            "metric": "frame-pairs/sec @512x1024 iters=12; EPE vs reference",
            "value": round(pairs / elapsed, 3), "unit": "frame-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("bf16x3 split MFMA, f32 accumulate/storage" if model._weights()["precision"] == 1 else "f32"),
            "data": "synthetic",
            "config": {"workload": f"PriOr-RAFT forward, {args.batch} synthetic 512x1024 ERP pair(s) per GPU per step, "
                                   "iters=12, test_mode (BASELINE.json configs[1])",
                       "pairs_per_gpu_per_step": args.batch, "height": H, "width": W, "iters": ITERS,
                       "parallelism": f"pairs sharded over {world} rank(s), no data-path collective"
                                      + (f"; timing barrier / max over {backend}" if dist is not None else ""),
                       "weights": "deterministic closed-form fill (no checkpoints offline)",
                       "hip_graph": bool(model.use_graph),
                       "encoders": "libpriorflow_hip.so (HIP kernels, both precisions)"},
        }
        try:
            log("per-kernel HIP-event pass")
            result["roofline"], result["roofline_corr"] = profile_kernels(model, i1, i2)
        except Exception as exc:  # measured extras must not hide the headline number
            result["roofline"] = {"error": repr(exc)}
        if world == 1 and not args.no_cpu_baseline:
            ref_cpu = None
            try:
                result["cpu_baseline"], result["parity"], ref_cpu = cpu_baseline(params, i1c, i2c, flow)
            except Exception as exc:
                result["cpu_baseline"] = {"error": repr(exc)}
            if model._weights()["precision"] == 1:
                try:
                    result["fp32_exact"] = fp32_reference_point(params, device, i1, i2, ref_cpu)
                except Exception as exc:
                    result["fp32_exact"] = {"error": repr(exc)}
        print(json.dumps(result), flush=True)
    if dist is not None:
        rank_barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
