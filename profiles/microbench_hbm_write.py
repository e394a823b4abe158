#!/usr/bin/env python3
"""Round 6: what does this chip accept as a pure WRITE stream?  The corr + pyramid build writes 356.5 MB per launch and reads 17 MB
(373.3 MB algorithmic), so its HBM roofline is a write roofline.  Plain linear fills (torch's vectorised fill
kernel and hipMemsetAsync) of the launch's byte count, cycling over buffers that together exceed the 256 MiB memory-side cache, beside
a device-to-device copy of the same bytes (the read + write mix the 6.3 TB/s 'achievable' figure of the guide comes from).
   python profiles/microbench_hbm_write.py"""
import torch

dev = torch.device("cuda:0")
MB = 356.5
n = int(MB * 1e6 / 4)
bufs = [torch.empty(n, device=dev) for _ in range(6)]          # 2.1 GB: a fill never finds its lines in the memory-side cache
src = torch.randn(n, device=dev)


def timed(f, reps=12):
    for i in range(3):
        f(i)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for i in range(reps):
        f(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


cases = {
    "fill_ (torch vectorised fill kernel)": lambda i: bufs[i % 6].fill_(1.0),
    "zero_ (hipMemsetAsync)": lambda i: bufs[i % 6].zero_(),
    "copy_ device to device (read + write)": lambda i: bufs[i % 6].copy_(src),
}
for name, f in cases.items():
    ts = sorted(timed(f) for _ in range(5))
    us = ts[2]
    moved = MB * (2 if "copy" in name else 1)
    print(f"{name:42s} {us:7.1f} us per {MB} MB   {moved * 1e6 / us / 1e6:6.2f} TB/s of HBM traffic"
          f"   ({MB * 1e6 / us / 1e6 / 8:5.3f} of 8 TB/s counted as the corr build counts its bytes)")
