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
  roofline      the kernel with the largest summed time of one forward, WHATEVER its kind (every library launch is timed
                with HIP events in a single-stream eager pass): an MFMA conv -> algorithmic FLOPs / time vs the dense bf16
                MFMA peak (bf16x3 mode issues 3 MFMA FLOPs per algorithmic FLOP: mfma_pipe_util) or the fp32 MFMA peak;
                a lookup / corr build -> algorithmic bytes / time vs 8 TB/s (roofline_corr: 20 launches back to back
                on the forward's own operands -- an event pair around one launch of the eager pass is a poor clock for that kernel);
  roofline_conv / roofline_gru / roofline_corr / roofline_lookup / roofline_combine
                the same object for the dominant MFMA conv, the SepConvGRU's convolutions (north_star's "MFMA utilisation on
                the GRU convs": flops / time vs the bf16 peak + the committed PMC MFMA-busy fraction), the fused
                correlation-volume + pyramid build (north_star's HBM target), the DCCL lookup and the fused rotate-back +
                1x1; kernels_by_time: the top launches by time;
  cpu_baseline  the CPU oracle (oracle/priorflow_oracle.py, the checker -- never the product)
                timed on the host cores on the same pair, rank 0 at N=1 only, plus the EPE of
                the GPU flow against it (`parity`);
  batch32       BASELINE.json configs[2]: 32 resident pairs, a few timed steps, pair 0 against the oracle;
  train_step    BASELINE.json configs[3] per GPU: the product training step at 384x512, iters=12, one pair, captured as a HIP
                graph (train.GraphedTrainStep): ms/step, training pairs/s, launches of an eager step, loss trajectory;
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

torch = None     # imported by main() AFTER the self-launch decision: the launching parent never loads torch / the HIP runtime

H, W, ITERS = 512, 1024, 12
# Committed rocprofv3 summaries of THIS command, one entry per workload (profiles/profile_index.json: batch, gpus, H, W, iters ->
# `stats` = --kernel-trace --stats of the graph replay, `pmc` = the FETCH_SIZE / WRITE_SIZE passes of profiles/pmc_traffic.py).
# `in_replay_us` and `traffic` are attached only when an entry matches the run's workload; otherwise they are null with a note.
PROFILE_INDEX = "profile_index.json"
PROFILE = {"stats": None, "pmc": None, "mfma": None, "round": None, "note": "no workload selected yet"}      # set by select_profile()
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: bf16 MFMA, dense (not the 2:1-sparse figure)
# what a loop of v_mfma_f32_32x32x16_bf16 and nothing else sustains on all 256 CUs with operands that toggle like real data
# (profiles/mfma_rate/mfma_rate.hip -> profiles/r3_mfma_rate.txt: 1.83 PFLOP/s; 2.2-2.5 with constant operands or on 64 CUs).
# Reported beside `peak` as context for `mfma_pipe_util`; `frac` stays against the nameplate.
SUSTAINED_BF16_MFMA_TFLOPS = 1830.0
PEAK_HBM_GBPS = 8000.0            # MI355X_MICROARCH.md: HBM3E spec
# generic kernel by pf_conv2d_tile code, spelled as rocprofv3 prints the instantiation (the committed profiles are matched by name)
TILE_NAMES = {0: "pf_conv_mfma_kernel<4, 1, 1", 1: "pf_conv_mfma_kernel<2, 2, 1", 2: "pf_conv_mfma_kernel<2, 2, 2"}


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
    ap.add_argument("--no-batch32", action="store_true", help="skip the bounded configs[2] leg of the N=1 line")
    ap.add_argument("--no-train-step", action="store_true", help="skip the bounded training-step leg (configs[3] per GPU) of the N=1 line")
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


def select_profile(batch, gpus):
    """Pick the committed profile files whose recorded workload is this run's (a batch-32 or 2-rank line must not
    carry the B=1 single-GPU numbers)."""
    path = os.path.join(ROOT, "profiles", PROFILE_INDEX)
    want = {"batch": batch, "gpus": gpus, "H": H, "W": W, "iters": ITERS}
    try:
        with open(path) as f:
            entries = json.load(f)["profiles"]
    except (OSError, ValueError, KeyError):
        entries = []
    for e in entries:
        if all(e.get(k) == v for k, v in want.items()):
            PROFILE.update(stats=e.get("stats"), pmc=e.get("pmc"), mfma=e.get("mfma"), round=e.get("round"),
                           note="profiles/%s entry for %s" % (PROFILE_INDEX, want))
            return
    PROFILE.update(stats=None, pmc=None, mfma=None, note="profiles/%s has no entry for %s: in_replay_us / traffic not reported "
                                              "(the committed profiles are of other workloads)" % (PROFILE_INDEX, want))


def pmc_traffic(kernel_substr):
    """HBM-side bytes per launch of a kernel from the committed PMC passes (profiles/pmc_traffic.py:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, FETCH_SIZE doubled on gfx950, KiB -> bytes).
    Counters cannot be collected inside this process; None when no committed profile matches this run's workload."""
    if not PROFILE["pmc"]:
        return None
    path = os.path.join(ROOT, "profiles", PROFILE["pmc"])
    try:
        with open(path) as f:
            kernels = json.load(f)["kernels"]
    except (OSError, ValueError, KeyError):
        return None
    for name, v in kernels.items():
        if kernel_substr in name:
            return round(v["hbm_bytes_per_launch"])
    return None


def pmc_mfma_busy(kernel_substr):
    """Matrix-pipe busy fraction of a kernel from the committed PMC pass (profiles/mfma_busy.py: SQ_VALU_MFMA_BUSY_CYCLES over
    4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8, median over the launches of a single-stream eager run); None when no committed
    profile matches this run's workload or the file has no such kernel."""
    if not PROFILE.get("mfma"):
        return None
    try:
        with open(os.path.join(ROOT, "profiles", PROFILE["mfma"])) as f:
            kernels = json.load(f)["kernels"]
    except (OSError, ValueError, KeyError):
        return None
    for name, v in kernels.items():
        if kernel_substr in name:
            return round(v["mfma_busy_fraction"], 4)
    return None


def replay_stats(kernel_substr):
    """Average duration of a kernel INSIDE the graph replay (where the side streams' kernels share the chip), from the
    committed rocprofv3 --kernel-trace --stats summary of this workload (select_profile); None when there is none."""
    import csv
    if not PROFILE["stats"]:
        return None
    path = os.path.join(ROOT, "profiles", PROFILE["stats"])
    try:
        with open(path) as f:
            for row in csv.DictReader(f):
                if kernel_substr in row.get("Name", ""):
                    return round(float(row["AverageNs"]) / 1e3, 1)
    except (OSError, ValueError, KeyError):
        pass
    return None


def traffic_fields(kernel_substr):
    """`traffic` (+ a loud note when the committed PMC passes have no entry for this kernel, e.g. after a rename)."""
    t = pmc_traffic(kernel_substr)
    if not PROFILE["pmc"]:
        return {"traffic": None, "traffic_note": PROFILE["note"]}
    out = {"traffic": t, "traffic_note": "HBM-side bytes/launch from the committed PMC passes (profiles/%s): counters cannot be "
                                         "read inside this process" % PROFILE["pmc"]}
    if t is None:
        out["traffic_error"] = "profiles/%s has no entry matching %r -- re-run profiles/pmc_traffic.py" % (PROFILE["pmc"], kernel_substr)
        log("WARNING: " + out["traffic_error"])
    return out


def replay_ranking(top=6):
    """The committed replay profile's own ranking (summed kernel time inside the captured multi-stream forward): the bench
    line's `roofline` is chosen by alone-on-chip event time, and the two rankings differ (by replay time the lookup
    leads)."""
    import csv
    if not PROFILE["stats"]:
        return None
    try:
        with open(os.path.join(ROOT, "profiles", PROFILE["stats"])) as f:
            rows = [r for r in csv.DictReader(f) if "pf_" in r.get("Name", "")]
    except (OSError, ValueError, KeyError):
        return None
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows) or 1.0
    import re

    def short(n):
        m = re.search(r"_Z\d+(pf_\w+?_elem)l", n)          # pf_elem_kernel<Args, &pf_xxx_elem>: name the element function
        if m:
            return m.group(1)
        return n.replace("(anonymous namespace)::", "").replace("pfconv::", "").replace("void ", "")[:70]
    return {"source": "profiles/%s" % PROFILE["stats"],
            "kernels": [{"kernel": short(r["Name"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 1),
                         "share_of_kernel_time": round(float(r["TotalDurationNs"]) / tot, 4)} for r in rows[:top]]}


def profile_kernels(model, i1, i2):
    """One eager, single-stream forward with HIP events around EVERY library launch.  Returns the roofline objects of the bench
    line: `roofline` is the kernel with the largest summed time over the whole forward (whatever its kind), plus the named
    ones (`roofline_conv`: the MFMA conv with the largest summed time, `roofline_corr`, `roofline_lookup`) and the table of
    all kinds by time.  `avg_launch_us` is the kernel ALONE on the chip (one stream, event gap subtracted); `in_replay_us`,
    where the committed profile has the kernel, is its average inside the captured multi-stream forward."""
    lib = model._lib()
    recs = []          # (kind, name, work, start_event, end_event)
    B = i1.shape[0]

    def ev():
        return torch.cuda.Event(enable_timing=True)

    def conv_name(descs, Bc, H8, W8):
        tile = lib.conv2d_tile(descs, Bc, H8, W8)
        d0 = descs[0]
        if tile == 6:      # the weights-stationary kernel of the encoders' 3x3 64 -> 64 convolutions (round 5)
            return "pf_enc_conv64_kernel"
        roles = lib.conv2d_roles(descs, Bc, H8, W8) if tile in (3, 4, 5) else 0
        if roles >= 16:    # pre-split operands: the all-DMA kernel, <NT, KH, KW, WN> exactly as rocprof names it
            r = roles - 16
            return "pf_conv_dma_kernel<%d, %d, %d, %d>" % (2 if (tile >= 4 or r == 2) else 1, d0.kh, d0.kw, 3 - r)
        if roles:          # role-specialised waves
            return "pf_conv_ws_kernel<%d, %d, %d, %d>" % (2 if (tile == 4 or roles == 2) else 1, d0.kh, d0.kw, 3 - roles)
        if tile >= 3:      # halo kernel <NT, KH, KW, AFFINE, TH>
            return "pf_conv_halo_kernel<%d, %d, %d, %s, %d>" % (
                1 if tile == 3 else 2, d0.kh, d0.kw, "true" if d0.in_scale else "false", 8 if tile == 5 else 4)
        return TILE_NAMES[tile] + (", true>" if d0.precision == 1 else ", false>")

    def wrap(attr, kind, name_fn, work_fn):
        orig = getattr(lib, attr)

        def f(*a, **k):
            s, e = ev(), ev()
            s.record()
            r = orig(*a, **k)
            e.record()
            recs.append((kind, name_fn(*a, **k), work_fn(*a, **k), s, e))
            return r
        setattr(lib, attr, f)
        return attr, orig

    def corr_bytes(Bc, H8, W8, c):
        n = H8 * W8
        return Bc * (4.0 * n * n * 85.0 / 64.0 + 2.0 * 4.0 * n * c)

    px = lambda c: c.shape[0] * c.shape[2] * c.shape[3]  # noqa: E731  (coords [B,2,H8,W8] -> pixels)
    saved = []          # filled inside the try below: a failing wrap() must not leave earlier wrappers installed
    specs = [
        ("conv2d", "conv", lambda d, Bc, H8, W8, like: conv_name(d, Bc, H8, W8),
             lambda d, Bc, H8, W8, like: sum(2.0 * Bc * H8 * W8 * x.cout * x.kh * x.kw * (x.c0 + x.c1) for x in d)),
        ("corr_pyramid", "corr", lambda *a: "pf_corr_kernel", lambda f1, f2, lv, Bc, H8, W8: corr_bytes(Bc, H8, W8, f1.shape[-1])),
        # bf16x3: the role-split kernel (round 5) on maps with W/8 % 64 == 0 and H/8 % 8 == 0 unless PRIORFLOW_CORR_RS=0
        ("corr_pyramid_bf16x3", "corr",
             lambda f1, f2, lv, Bc, H8, W8, c: ("pf_corr_rs_kernel" if W8 % 64 == 0 and H8 % 8 == 0 and c == 256
                                                and os.environ.get("PRIORFLOW_CORR_RS", "1") != "0" else "pf_corr_kernel"),
             lambda f1, f2, lv, Bc, H8, W8, c: corr_bytes(Bc, H8, W8, c)),
        # SURVEY.md 8(d), one branch: own 10x10 patch x 4 B x 4 levels read + 324 x 4 B written; cross <= 81 x 4 taps x 4 B x 4
        # levels read + 324 x 4 B written = 9 376 B per pixel (76.8 MB at 64x128)
        ("dccl_lookup", "lookup", lambda *a, **k: "pf_lookup", lambda coords, *a, **k: px(coords) * (1600.0 + 1296.0 + 5184.0 + 1296.0)),
        # compulsory bytes per pixel and branch: its own row + one raw row read (324 fp32 each: the four bilinear corners of the
        # rotate-back are rows that neighbouring pixels share, every raw row is fetched once) + 256 channels written (4 B in
        # either form).  (Round 3 counted the four corners per pixel: 117.8 MB "algorithmic" against 82.6 MB of PMC traffic.)
        ("dccl_combine_conv1x1", "combine", lambda *a: "pf_combine_conv_kernel",
             lambda items, Bc, H8, W8: len(items) * Bc * H8 * W8 * (2 * 1296.0 + 1024.0)),
        ("motion_prep", "other", lambda *a, **k: "pf_motion_prep_kernel", lambda *a, **k: 0.0),
        ("conv2d_direct_group", "other", lambda *a, **k: "pf_stem7x7c2_valu", lambda *a, **k: 0.0),
        ("conf_stem", "other", lambda *a, **k: "pf_conf_stem_kernel", lambda *a, **k: 0.0),
        ("flow_head_out", "other", lambda *a, **k: "pf_flow_out_strip", lambda *a, **k: 0.0),
        ("norm_act", "other", lambda *a, **k: "pf_norm_act_vec", lambda *a, **k: 0.0),
        ("channel_stats_final", "other", lambda *a, **k: "pf_stats_final", lambda *a, **k: 0.0),
        ("channel_stats", "other", lambda *a, **k: "pf_stats_partial+final", lambda *a, **k: 0.0),
    ]
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
        for spec in specs:
            saved.append(wrap(*spec))
        with torch.no_grad():
            for _ in range(2):          # second pass is the measured one (caches warm)
                recs.clear()
                # park the GPU so the host enqueues the whole eager forward ahead of it:
                # the HIP-event intervals then contain kernel time only, not host launch gaps
                torch.cuda._sleep(int(30e6))
                model(i1, i2, iters=ITERS, test_mode=True)
                torch.cuda.synchronize()
    finally:
        for attr, orig in saved:
            setattr(lib, attr, orig)
        model.use_graph, model.use_streams = was, was_streams
    by = {}            # (kind, name) -> [work, ms, launches]
    for kind, name, work, s, e in recs:
        ms = max(s.elapsed_time(e) - gap_ms, 1e-4)
        t = by.setdefault((kind, name), [0.0, 0.0, 0])
        t[0] += work; t[1] += ms; t[2] += 1
    from prior_flow_amd._lib import PREC_BF16X3
    split = model._weights()["precision"] == PREC_BF16X3
    peak_mfma = PEAK_BF16_MFMA_TFLOPS if split else PEAK_F32_MFMA_TFLOPS
    how = "avg_launch_us: the kernel alone on the chip (single-stream eager pass, HIP events, dispatch gap subtracted); " \
          "in_replay_us: its rocprofv3 average inside the captured multi-stream forward (%s)" % (
              ("profiles/%s: the committed round-%s profile, taken on ANOTHER box than the one being timed -- boxes differ by up "
               "to +-4 %%" % (PROFILE["stats"], PROFILE.get("round"))) if PROFILE["stats"] else PROFILE["note"])

    def obj(kind, name):
        work, ms, n = by[(kind, name)]
        o = {"kernel": name, "launches_per_forward": n, "avg_launch_us": round(ms / n * 1e3, 1), "in_replay_us": replay_stats(name),
             "ms_per_forward": round(ms, 3), "timing": how, "event_gap_us_subtracted": round(gap_ms * 1e3, 2)}
        if kind == "conv":
            ach = work / (ms * 1e-3) / 1e12
            o.update({"kernel": name + (" bf16x3" if split else " fp32"), "bound": "mfma", "achieved": round(ach, 2), "peak": peak_mfma,
                      "unit": "TFLOP/s", "frac": round(ach / peak_mfma, 4),
                      # the 3-pass split issues 3 bf16 MFMA FLOPs per algorithmic FLOP: pipe utilisation
                      "mfma_issue_tflops": round(ach * (3 if split else 1), 2),
                      "mfma_pipe_util": round(ach * (3 if split else 1) / peak_mfma, 4),
                      **({"mfma_real_data_rate_tflops": SUSTAINED_BF16_MFMA_TFLOPS,
                          "mfma_issue_vs_real_data_rate": round(ach * 3 / SUSTAINED_BF16_MFMA_TFLOPS, 4),
                          "mfma_real_data_rate_source": "profiles/r3_mfma_rate.txt (MFMA-only loop, toggling operands, 256 CUs)"}
                         if split else {}),
                      "gflop_per_forward": round(work / 1e9, 1),
                      "mfma_busy_pmc": pmc_mfma_busy(name)})
        elif work > 0:
            gbps = work / (ms * 1e-3) / 1e9
            o.update({"bound": "hbm", "achieved": round(gbps, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                      "frac": round(gbps / PEAK_HBM_GBPS, 4), "mb_per_launch": round(work / n / 1e6, 1),
                      "bytes": "algorithmic bytes per launch (SURVEY.md 8(d))"})
        else:
            o.update({"bound": "latency", "achieved": None, "peak": None, "unit": None, "frac": None})
        o.update(traffic_fields(name))
        return o

    order = sorted(by, key=lambda k: -by[k][1])
    convs = [k for k in order if k[0] == "conv"]
    out = {"roofline": obj(*order[0])}
    out["roofline"]["chosen_as"] = ("the kernel with the largest summed ALONE-ON-CHIP time of the forward (HIP events, single "
                                    "stream), all kinds; `replay_ranking` is the committed rocprofv3 ranking inside the graph replay")
    out["replay_ranking"] = replay_ranking()
    if convs:
        out["roofline_conv"] = obj(*convs[0])
        all_fl = sum(by[k][0] for k in convs)
        all_ms = sum(by[k][1] for k in convs)
        out["roofline_conv"]["all_conv_kernels"] = {"gflop": round(all_fl / 1e9, 1), "ms": round(all_ms, 3),
                                                    "tflops": round(all_fl / (all_ms * 1e-3) / 1e12, 2)}
    # the SepConvGRU's convolutions (core/update.py:46-60; 1x5 then 5x1: z|r fused, then q) -- the north star asks for the MFMA
    # utilisation "on the GRU convs", so they get an object of their own whatever kernel leads the forward
    gru = [k for k in convs if ", 1, 5," in k[1] or ", 5, 1," in k[1]]
    if gru:
        out["roofline_gru"] = obj(*gru[0])
        g_fl = sum(by[k][0] for k in gru)
        g_ms = sum(by[k][1] for k in gru)
        out["roofline_gru"]["all_gru_kernels"] = {
            "kernels": [{"kernel": k[1], "launches": by[k][2], "avg_launch_us": round(by[k][1] / by[k][2] * 1e3, 1),
                         "in_replay_us": replay_stats(k[1]), "mfma_busy_pmc": pmc_mfma_busy(k[1])} for k in gru],
            "gflop": round(g_fl / 1e9, 1), "ms": round(g_ms, 3), "tflops": round(g_fl / (g_ms * 1e-3) / 1e12, 2),
            "frac": round(g_fl / (g_ms * 1e-3) / 1e12 / peak_mfma, 4)}
    for kind, key in (("corr", "roofline_corr"), ("lookup", "roofline_lookup"), ("combine", "roofline_combine")):
        ks = [k for k in order if k[0] == kind]
        if ks:
            out[key] = obj(*ks[0])
    if "roofline_corr" in out:
        out["roofline_corr"]["note"] = ("34.4 GFLOP/launch: 3-pass bf16 MFMA + 373 MB of once-written output" if split else
                                        "exact-fp32 MFMA makes this kernel compute-bound (34.4 GFLOP per launch)")
        if split:
            # The event pair around ONE launch of the eager pass is a poor clock for this kernel: it reads 100-150 us for a launch
            # that takes 87-92 us inside the replay and 93-105 us back to back (bimodal from run to run, whichever MFMA form the
            # kernel uses: profiles/r6_ab_m16_forward.txt).  So the object's figure is a dedicated measurement on the forward's own
            # operands: 20 launches back to back, alternating the two volumes like the forward does, one event pair around all of
            # them; the eager pass's reading stays beside it.
            try:
                r = out["roofline_corr"]
                ws = next(w for k, w in model._ws.items() if k[:3] == (B, H, W))
                rows = ws.B * ws.N
                fs = [ws.f_split[i * rows:(i + 1) * rows] for i in range(4)]
                launch = lambda k: lib.corr_pyramid_bf16x3(fs[2 * (k & 1)], fs[2 * (k & 1) + 1], ws.pyr_b if k & 1 else ws.pyr_a,   # noqa: E731
                                                           ws.B, ws.H8, ws.W8, 256)
                for k in range(4):
                    launch(k)
                torch.cuda.synchronize()
                s_, e_ = ev(), ev()
                s_.record()
                for k in range(20):
                    launch(k)
                e_.record()
                torch.cuda.synchronize()
                us = s_.elapsed_time(e_) * 1e3 / 20
                bytes_per = r["mb_per_launch"] * 1e6
                r["event_pair_in_eager_pass_us"] = r["avg_launch_us"]
                r["avg_launch_us"] = round(us, 1)
                r["achieved"] = round(bytes_per / (us * 1e-6) / 1e9, 1)
                r["frac"] = round(r["achieved"] / PEAK_HBM_GBPS, 4)
                r["ms_per_forward"] = round(2 * us / 1e3, 3)
                r["timing"] = ("avg_launch_us: 20 launches back to back on the forward's own operands, the two volumes alternating, one HIP "
                               "event pair around all of them; event_pair_in_eager_pass_us: one event pair around one launch of the "
                               "single-stream eager pass (what the other objects quote); in_replay_us: rocprofv3 average inside the captured forward")
            except Exception as exc:      # noqa: BLE001  (a measured extra must not hide the line)
                out["roofline_corr"]["back_to_back_error"] = repr(exc)
    out["kernels_by_time"] = [{"kernel": k[1], "kind": k[0], "launches": by[k][2], "ms": round(by[k][1], 3)} for k in order[:12]]
    return out


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


def batch32_point(params, device, ref_cpu, steps=4, warmup=2, batch=32):
    """BASELINE.json configs[2] on the driver's own line: the same forward over a resident batch of 32 synthetic pairs
    (HIP-graph replay, default precision), timed over `steps` steps; pair 0 of that batch is the B=1 pair (the synthetic
    generator is a counter-based stream), so its EPE is taken against the same CPU-oracle flow."""
    from prior_flow_amd import synthetic_pair
    import priorflow_oracle as po
    m, _ = build_model(device)
    i1, i2 = (t.to(device) for t in synthetic_pair(batch, H, W, seed=1234))
    with torch.no_grad():
        for _ in range(warmup):
            flow = m(i1, i2, iters=ITERS, test_mode=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            flow = m(i1, i2, iters=ITERS, test_mode=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    out = {"workload": f"{batch} synthetic 512x1024 ERP pairs resident on one GPU, iters={ITERS}, test_mode (BASELINE.json configs[2])",
           "value": round(batch * steps / dt, 3), "unit": "frame-pairs/s", "ms_per_step": round(dt / steps * 1e3, 3),
           "steps": steps, "warmup": warmup, "hip_graph": bool(m.use_graph)}
    if ref_cpu is not None:
        epe = po.epe(flow[:1].cpu(), ref_cpu)
        out["epe_mean_pair0"], out["epe_max_pair0"] = float(epe.mean()), float(epe.max())
    del m, i1, i2, flow
    torch.cuda.empty_cache()
    return out


def train_step_point(device, steps=10, size=(384, 512), iters=ITERS):
    """BASELINE.json configs[3] per GPU on the driver's own line: the product training step (zero_grad, GT rotation, forward of
    both branches, sequence loss, backward, clip, fused AdamW -- train_flow.py:120-141) at the reference's training crop
    (train_flow.py:217: 384x512, iters=12), one pair per GPU, as train.GraphedTrainStep: one eager step, the capture, one replay,
    then `steps` timed replays on fresh data each (the loss is read after the timed region).  Also one eager step's launch count."""
    from prior_flow_amd import autograd as ag
    from prior_flow_amd import det_state_dict, synthetic_pair
    from prior_flow_amd import train as tr
    from prior_flow_amd.modules import state_dict_shapes
    from prior_flow_amd.prior_raft import PriOr_RAFT
    Ht, Wt = size
    model = PriOr_RAFT(argparse.Namespace(mixed_precision=False, dropout=0.0))
    model.load_state_dict(det_state_dict(state_dict_shapes()), strict=True)
    model = model.to(device).train()
    model.freeze_bn()                                                            # train_flow.py:107-108
    opt, sched = tr.fetch_optimizer(argparse.Namespace(lr=2e-5, wdecay=5e-5, epsilon=1e-8, num_steps=100000), model)
    i1, i2 = (t.to(device) for t in synthetic_pair(1, Ht, Wt, seed=1234))
    gen = torch.Generator().manual_seed(99)
    gt = (torch.rand(1, 2, Ht, Wt, generator=gen) * 8 - 4).to(device)
    valid = torch.ones(1, Ht, Wt, device=device)
    crit = tr.uniform_loss(Ht, Wt, device=device)
    stepper = tr.GraphedTrainStep(model, opt, sched, crit, iters=iters, clip=1.0, warmup=1)
    torch.cuda.reset_peak_memory_stats(device)
    ag.STATS["hip"] = ag.STATS["torch"] = 0
    losses = [stepper(i1, i2, gt, valid)[0]]                                     # eager step
    torch.cuda.synchronize()
    launches = {"hip_library": int(ag.STATS["hip"]), "torch_conv_or_gemm": int(ag.STATS["torch"])}
    for _ in range(2):                                                           # capture + first replay, one more replay
        losses.append(stepper(i1, i2, gt, valid)[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        losses.append(stepper((i1 + float(k % 3)).clamp(0, 255), i2, gt, valid)[0])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"workload": f"training step {Ht}x{Wt}, iters={iters}, 1 pair per GPU (BASELINE.json configs[3] per GPU): forward + backward + "
                       "clip + AdamW on the HIP library, captured as one HIP graph (train.GraphedTrainStep)",
           "value": round(steps / dt, 3), "unit": "training frame-pairs/s", "ms_per_step": round(dt / steps * 1e3, 3),
           "steps": steps, "graphs": len(stepper.graphs), "eager_step_launches": launches,
           "loss_trajectory": [round(float(x), 4) for x in losses],
           "grad_parity": "tests/test_hip_train_step.py (reference step golden, oracle autograd, loop node vs tape at this crop)",
           "peak_mem_GB": round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 2)}
    del stepper, model, opt
    torch.cuda.empty_cache()
    return out


def visible_gpus() -> int:
    """GPUs this process would see, WITHOUT touching the HIP runtime (on this pool a process that has initialised the GPU must
    not fork + exec children): the visibility masks first, else the KFD topology (a node with simd_count > 0 is a GPU)."""
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([t for t in v.split(",") if t.strip() != ""])
    import glob
    n = 0
    for path in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            for line in open(path):
                if line.startswith("simd_count") and int(line.split()[1]) > 0:
                    n += 1
        except (OSError, ValueError, IndexError):
            pass
    return n


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (fresh processes; this parent
    never touches the GPU -- not even to count devices, see visible_gpus) and relay their output."""
    import socket
    import subprocess
    n = args.gpus
    have = visible_gpus()
    rehearsal = os.environ.get("PRIORFLOW_BENCH_BACKEND", "nccl") != "nccl"     # gloo: ranks may share a card
    if have < n and not (rehearsal and have >= 1):
        print(f"bench.py: --gpus {n} but only {have} GPU(s) are visible", file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    dump = os.environ.get("PRIORFLOW_BENCH_PARENT_MAPS")        # tests: which shared objects has the launching parent mapped?
    if dump:
        with open("/proc/self/maps") as f, open(dump, "w") as o:
            o.write("\n".join(sorted({ln.split()[-1] for ln in f if ".so" in ln})))
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    codes = [p.wait() for p in procs]
    return max(abs(c) for c in codes)


def rccl_version():
    """Version of the RCCL library torch is linked against (reported, never required: the line must print without it)."""
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception as exc:        # noqa: BLE001
        return "unavailable: " + repr(exc)


def init_ranks(world: int, rank: int, device):
    """torch.distributed for the barrier / max-over-ranks of the timing (the forward itself has no collective).
    RCCL first; PRIORFLOW_BENCH_BACKEND=gloo selects the CPU backend for rehearsals on a box without N GPUs."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = os.environ.get("PRIORFLOW_BENCH_BACKEND", "nccl")       # nccl == RCCL on ROCm
    if backend == "nccl":      # bind the communicator to this rank's device eagerly: a bad rendezvous fails here, not in a barrier
        dist.init_process_group(backend, rank=rank, world_size=world, device_id=device)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    probe = torch.ones(1, device=device if backend == "nccl" else "cpu")
    dist.all_reduce(probe)                                             # first collective: builds the communicator
    assert int(probe.item()) == world, f"all-reduce over {world} ranks returned {probe.item()}"
    return dist, backend


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    global torch
    import torch
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

    select_profile(args.batch, world)
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
                                   "iters=12, test_mode (BASELINE.json " + ("configs[1]" if args.batch == 1 else
                                   "configs[2]" if args.batch == 32 else f"configs[1] at batch {args.batch}") + ")",
                       "pairs_per_gpu_per_step": args.batch, "height": H, "width": W, "iters": ITERS,
                       "parallelism": f"pairs sharded over {world} rank(s), no data-path collective"
                                      + (f"; timing barrier / max over {backend}" if dist is not None else ""),
                       # what the communicator really saw (a SCALE record shows RCCL spanning N ranks, or that it was a gloo rehearsal)
                       "collective_backend": backend, "rccl_world": world if backend == "nccl" else None,
                       "rccl_version": rccl_version() if backend == "nccl" else None,
                       "weights": "deterministic closed-form fill (no checkpoints offline)",
                       "hip_graph": bool(model.use_graph),
                       "encoders": "libpriorflow_hip.so (HIP kernels, both precisions)"},
        }
        try:
            log("per-kernel HIP-event pass")
            result.update(profile_kernels(model, i1, i2))
        except Exception as exc:  # measured extras must not hide the headline number
            result["roofline"] = {"error": repr(exc)}
        if world == 1 and not args.no_cpu_baseline:
            ref_cpu = None
            try:
                # bounded sample: the first pair of the batch (a pair's result does not depend on its batch mates)
                result["cpu_baseline"], result["parity"], ref_cpu = cpu_baseline(params, i1c[:1], i2c[:1], flow[:1])
            except Exception as exc:
                result["cpu_baseline"] = {"error": repr(exc)}
            if model._weights()["precision"] == 1:
                try:
                    result["fp32_exact"] = fp32_reference_point(params, device, i1[:1], i2[:1], ref_cpu)
                except Exception as exc:
                    result["fp32_exact"] = {"error": repr(exc)}
            if args.batch == 1 and not args.no_batch32:
                try:
                    log("batch-32 leg (configs[2])")
                    result["batch32"] = batch32_point(params, device, ref_cpu)
                except Exception as exc:
                    result["batch32"] = {"error": repr(exc)}
            if args.batch == 1 and not args.no_train_step:
                try:
                    log("training-step leg (configs[3] per GPU)")
                    del model, flow
                    torch.cuda.empty_cache()
                    result["train_step"] = train_step_point(device)
                except Exception as exc:
                    result["train_step"] = {"error": repr(exc)}
        print(json.dumps(result), flush=True)
    if dist is not None:
        rank_barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
