"""Ordered kernel sequence of ONE training step from a rocprofv3 kernel trace of profiles/time_train_step.py (eager step):
the launches between the last two AdamW kernels, run-length compressed, with at::native / copy kernels marked '*'.
    python3 profiles/train_sequence.py <t_kernel_trace.csv>"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
lo, hi = ends[-2] + 1, ends[-1] + 1
step = rows[lo:hi]


def short(n):
    nat = "at::native" in n or "rocclr" in n or "Cijk" in n
    m = re.search(r"pf_\w+", n)
    if m and not nat:
        t = re.search(r"<[^>]*>", n)
        return m.group(0) + (t.group(0) if t and len(t.group(0)) < 24 else "")
    for key in ("FillFunctor", "CUDAFunctor_add", "direct_copy", "copyBuffer", "MulFunctor", "clamp", "sum_functor", "threshold",
                "AUnaryFunctor", "BinaryFunctor", "reduce_kernel", "addcmul", "CatArray", "index", "neg", "sqrt", "div"):
        if key in n:
            return "*" + key
    return "*" + n[:60]


seq = [short(r["Kernel_Name"]) for r in step]
print(f"# {len(seq)} launches in the step; at::native/copy: {sum(s[0] == '*' for s in seq)}")
out, i = [], 0
while i < len(seq):
    j = i
    while j < len(seq) and seq[j] == seq[i]:
        j += 1
    out.append(seq[i] + (f" x{j - i}" if j - i > 1 else ""))
    i = j
line = ""
for tok in out:
    if len(line) + len(tok) > 150:
        print(line)
        line = ""
    line += tok + " | "
print(line)
