#!/usr/bin/env python3
"""HBM roofline of the summarizer kernels at BASELINE sizes (HIP events)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayes_sim_ig_amd as B   # noqa: E402

B._lib.require_gpu()
dev = 'cuda:0'


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


cases = [
    ('summary_start  ShadowHand cfg5', 'summary_start', 100_000, 11, 211, 20, {}),
    ('summary_start  ShadowHand chunk', 'summary_start', 1000, 11, 211, 20, {}),
    ('summary_corrdiff Ant cfg3', 'summary_corrdiff', 50_000, 51, 60, 8, {'check_finite': False}),
    ('summary_corrdiff Ant chunk', 'summary_corrdiff', 1000, 51, 60, 8, {'check_finite': False}),
    ('summary_corrdiff Cartpole cfg2', 'summary_corrdiff', 100_000, 21, 4, 1, {'check_finite': False}),
    ('signature d=22 depth3 cfg4B', 'summary_signatory', 100_000, 11, 17, 4, {}),
    ('signature d=232 depth1 cfg4A', 'summary_signatory', 100_000, 11, 211, 20, {}),
    ('signature d=6 depth3 Cartpole', 'summary_signatory', 100_000, 21, 4, 1, {}),
]
for name, fn, n, t, sd, ad, kw in cases:
    s = torch.randn(n, t, sd, device=dev)
    a = torch.rand(n, t, ad, device=dev)
    f = B.summarizers.summary_dim(fn, t, sd, ad)
    out = torch.empty(n, B._lib.round_up(f, 4), device=dev)
    us = timeit(lambda: getattr(B.summarizers, fn)(s, a, out=out, **kw))
    if fn == 'summary_start':
        nbytes = 2 * 4 * 10 * (sd + ad) * n
    elif fn == 'summary_corrdiff':
        w = min(5 if sd > 50 else 10, t)
        nbytes = 4 * (w * (sd + ad) + f) * n
    else:
        rows = 2 if f == 1 + sd + ad else t            # depth 1 reads first and last step only
        nbytes = 4 * (rows * (sd + ad) + f) * n        # time channel is generated, not read
    print('%-34s N=%-7d F=%-6d %9.1f us  %7.1f GB/s algorithmic (%.3f of 8 TB/s)'
          % (name, n, f, us, nbytes / us / 1e3, nbytes / us / 1e3 / 8000), flush=True)
