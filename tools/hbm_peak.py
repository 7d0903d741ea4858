#!/usr/bin/env python3
"""Measured HBM ceilings on this box (torch fill / copy), for context next to
the 8 TB/s spec used as the roofline peak."""
import torch
dev = 'cuda:0'
n = 1 << 29   # 2 GiB of fp32
a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
def t(fn, reps=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / reps * 1e-3
dt = t(lambda: a.fill_(1.0)); print('fill  (write only) : %.2f TB/s' % (4 * n / dt / 1e12))
dt = t(lambda: b.copy_(a));   print('copy  (read+write) : %.2f TB/s' % (8 * n / dt / 1e12))
dt = t(lambda: a.sum());      print('sum   (read only)  : %.2f TB/s' % (4 * n / dt / 1e12))
