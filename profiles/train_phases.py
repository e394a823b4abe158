"""Phases of ONE replayed training step (profiles/time_train_step.py --graph under rocprofv3 --kernel-trace): wall span, union of
kernel-busy time and launch count per phase, split at marker kernels.
    python3 profiles/train_phases.py <t_kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
step = rows[ends[-2] + 1:ends[-1] + 1]
t0 = int(step[0]["Start_Timestamp"])


def first(key, after=0):
    for i in range(after, len(step)):
        if key in step[i]["Kernel_Name"]:
            return i
    return None


def last(key):
    for i in range(len(step) - 1, -1, -1):
        if key in step[i]["Kernel_Name"]:
            return i
    return None


marks = [("prep + encoders fwd", 0)]
i_corr = first("pf_corr_rs_kernel") or first("pf_corr_kernel")
marks.append(("corr pyramids", i_corr - 2 if i_corr else 0))
i_loop = first("pf_motion_prep")
marks.append(("loop forward", i_loop))
i_loss = first("pf_seq_loss")
marks.append(("loss", i_loss))
i_bwd = first("pf_upsample_bwd", i_loss)
marks.append(("loop backward", i_bwd))
i_wg = first("pf_wgrad_kernel", i_bwd)
marks.append(("loop weight gradients", i_wg))
i_pyr = first("pf_pyramid_bwd", i_wg)
marks.append(("corr + encoders backward", i_pyr))
i_sq = last("sumsq") or last("sum_squares") or len(step) - 3
marks.append(("clip + AdamW", i_sq))
marks = [(n, i) for n, i in marks if i is not None]
print(f"# one replayed step: {len(step)} launches, {(int(step[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
for k, (name, i) in enumerate(marks):
    j = marks[k + 1][1] if k + 1 < len(marks) else len(step)
    seg = step[i:j]
    if not seg:
        continue
    a = int(seg[0]["Start_Timestamp"])
    b = max(int(r["End_Timestamp"]) for r in seg)
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
    busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    ksum = sum(e - s for s, e in iv)
    by = {}
    for r in seg:
        n = r["Kernel_Name"]
        import re
        m = re.search(r"pf_\w+?(_elem|_kernel|_wave|_vec\d?|_valu|_strip)?(<[^>]*>)?(?=[\(lR]|$)", n)
        key = (m.group(0) if m and "at::native" not in n else ("torch:" + (re.search(r"(\w+Functor\w*|direct_copy|copyBuffer|threshold|clamp|reduce_kernel|CatArray|index)", n) or [n[:40]])[0]))
        d = by.setdefault(key, [0, 0])
        d[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); d[1] += 1
    top = sorted(by.items(), key=lambda kv: -kv[1][0])[:7]
    print(f"{name:28s} start {(a - t0) / 1e3:9.1f} us  span {(b - a) / 1e3:9.1f} us  busy {busy / 1e3:9.1f} us  kernel sum {ksum / 1e3:9.1f} us  launches {len(seg):5d}")
    for k2, (t, c) in top:
        print(f"        {t / 1e3:9.1f} us  x{c:4d}  {k2[:90]}")
