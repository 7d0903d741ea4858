#!/usr/bin/env python3
"""Soak of the persistent update kernels: N fits of a BASELINE-shaped config back to back, every
run_training call of which is one launch of ~100 updates with ~10^3 bounded cross-workgroup waits per
workgroup.  Reports fits, launches, time-out fallbacks (RuntimeWarning of MDNN._give_up_a_level) and whether
the model still runs the persistent kernel at the end; the held-out NLL of every fit must be finite and
the first and last fits of the same seed bitwise equal (fixed summation orders)."""
import os
import sys
import time
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                     # noqa: E402
import bayes_sim_ig_amd as B     # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'cfg5'
fits = int(sys.argv[2]) if len(sys.argv) > 2 else 100
pairs = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
B.MDNN.VERBOSE = False
dev = 'cuda:0'
cfg = dict(bench.CONFIGS[name])
theta, states, actions = bench.synth_pairs(cfg, pairs, 3, dev)
fallbacks, first, t0 = 0, None, time.perf_counter()
for i in range(fits):
    bs = bench.build_gpu_model(B, cfg, dev, 77)
    np.random.seed(5)
    torch.manual_seed(5)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        logs = bs.fit(theta, states, actions)
        fallbacks += sum(1 for x in w if issubclass(x.category, RuntimeWarning))
    torch.cuda.synchronize()
    last = np.array([lg['test_loss'][-1] for lg in logs])
    assert np.isfinite(last).all(), (i, last)
    if first is None:
        first = (last.copy(), bs.model._flat.clone())
    persistent = int(B._lib.load().bsig_fit_is_persistent(bs.model._plan))
dt = time.perf_counter() - t0
same = bool(np.array_equal(first[0], last) and torch.equal(first[1], bs.model._flat))
print('%s: %d fits x %d pairs = %d launches in %.1f s (%.0f pairs/s incl. model builds); time-out fallbacks: %d; '
      'persistent kernel at the end: %d; first == last fit bitwise: %s'
      % (name, fits, pairs, fits * ((pairs + 999) // 1000), dt, fits * pairs / dt, fallbacks, persistent, same))
sys.exit(0 if (fallbacks == 0 and same) else 1)
