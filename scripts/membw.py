#!/usr/bin/env python3
"""Calibration: what this MI355X sustains for plain streaming fill / copy / read (torch kernels),
at a size inside the 256 MiB Infinity Cache and at sizes far beyond it.  Gives the practical
ceiling the Gobblet kernels' achieved GB/s should be read against."""
import torch

dev = "cuda:0"


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n / 1e3


for mb in (128, 1024, 4096):
    nbytes = mb << 20
    a = torch.empty(nbytes // 4, dtype=torch.int32, device=dev); b = torch.empty_like(a)
    tf = t(lambda: a.fill_(1)); tc = t(lambda: b.copy_(a)); tr = t(lambda: a.sum())
    print(f"{mb} MiB: fill {nbytes / tf / 1e12:.2f} TB/s  copy(r+w) {2 * nbytes / tc / 1e12:.2f} TB/s  "
          f"read(sum) {nbytes / tr / 1e12:.2f} TB/s")
