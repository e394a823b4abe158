#!/usr/bin/env python3
"""What can this MI355X write?  Pure streaming stores (torch fill_ / zero_), a copy, and a pure read (sum) on
buffers of the size of one correlation volume + pyramid (373 MB) and larger, HIP-event timed.
   python profiles/microbench_hbm_write.py
The corr + pyramid build writes 356.5 MB and reads 16.8 MB per launch: its HBM roofline is the WRITE rate."""
import torch

dev = torch.device("cuda:0")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


for mb in (373, 1024, 4096):
    n = mb * 1000 * 1000 // 4
    x = torch.empty(n, device=dev)
    y = torch.empty(n, device=dev)
    t_fill = timed(lambda: x.fill_(1.5))
    t_zero = timed(lambda: x.zero_())
    t_copy = timed(lambda: y.copy_(x))
    t_sum = timed(lambda: x.sum())
    b = n * 4 / 1e6
    print(f"{mb:5d} MB: fill_ {t_fill:7.1f} us = {b / t_fill:.2f} TB/s written | zero_ {t_zero:7.1f} us = {b / t_zero:.2f} TB/s | "
          f"copy_ {t_copy:7.1f} us = {b / t_copy:.2f} TB/s written (+ the same read) | sum {t_sum:7.1f} us = {b / t_sum:.2f} TB/s read")
