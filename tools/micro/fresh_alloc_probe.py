#!/usr/bin/env python3
"""Does a kernel pay for writing freshly allocated device memory?  Times the summarizer
and the staging copy over 100k ShadowHand trajectories into fresh / reused buffers."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bayes_sim_ig_amd as B   # noqa: E402

dev = 'cuda:0'
n = 100_000
s = torch.randn(n, 11, 211, device=dev)
a = torch.rand(n, 11, 20, device=dev)


def t(fn, tag):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    torch.cuda.synchronize()
    print('%-40s %8.3f ms' % (tag, e0.elapsed_time(e1)), flush=True)
    return r


keep = []
for i in range(3):
    keep.append(t(lambda: B.summary_start(s, a), 'summary_start fresh output #%d' % i))
out = torch.empty(n, 2312, device=dev)
for i in range(3):
    t(lambda: B.summary_start(s, a, out=out), 'summary_start reused output #%d' % i)
for i in range(3):
    t(lambda: torch.empty(n, 2312, device=dev).fill_(1.0), 'torch fill fresh #%d' % i)
    keep.append(torch.empty(n, 2312, device=dev))
x = keep[0]
dst = torch.empty(n, 2312, device=dev)
lib = B._lib.load()
for i in range(3):
    t(lambda: B._lib.check(lib.bsig_copy_rows(B._lib.ptr(x), x.stride(0), None, B._lib.ptr(dst), 2312, n, 2310,
                                              B._lib.stream())), 'copy_rows #%d' % i)
for i in range(3):
    t(lambda: dst.copy_(out), 'torch copy_ #%d' % i)
print(torch.cuda.memory_allocated() / 1e9, 'GB allocated')
