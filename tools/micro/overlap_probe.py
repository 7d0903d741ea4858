#!/usr/bin/env python3
"""Can held-out evaluations run beside the persistent update kernel?  Times a run of
20 updates (bsig_fit_updates) alone and with evaluation-sized GEMMs issued on a second
(lower-priority) stream at the same moment."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench                     # noqa: E402
import bayes_sim_ig_amd as B     # noqa: E402

B.MDNN.VERBOSE = False
dev = 'cuda:0'
lib = B._lib.require_gpu()
L = B._lib
name = sys.argv[1] if len(sys.argv) > 1 else 'cfg5'
cfg = dict(bench.CONFIGS[name])
theta, states, actions = bench.synth_pairs(cfg, 1000, 3, dev)
bs = bench.build_gpu_model(B, cfg, dev, 77)
summ = bs._summarize(states, actions)
bs.model.run_training(summ, theta, 100, 100)
plan = bs.model._plan
hi = torch.cuda.Stream(priority=-1)
lo = torch.cuda.Stream(priority=0)
a = torch.randn(200, 4096, device=dev)
w = torch.randn(4096, 260, device=dev)


def run(concurrent):
    torch.cuda.synchronize()
    with torch.cuda.stream(hi):
        L.check(lib.bsig_fit_begin(plan, 1, 100, L.stream()))
        L.check(lib.bsig_fit_updates(plan, 1, L.stream()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(hi):
        e0.record()
        L.check(lib.bsig_fit_updates(plan, 20, L.stream()))
        e1.record()
    if concurrent:
        with torch.cuda.stream(lo):
            g0.record()
            for _ in range(4):
                (a @ w).sum()
            g1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3, (g0.elapsed_time(g1) * 1e3 if concurrent else 0.0)


for c in (False, True, False, True, True):
    t, g = run(c)
    print('%s: 20 updates %.1f us%s' % (name, t, '   | 4 eval-sized GEMM+sum on the other stream: %.1f us' % g if c else ''))
