#!/usr/bin/env python3
"""Held-out NLL of the HIP path against the oracle over several data seeds (bench.nll_check: first
chunk of 1000 synthetic pairs, 100 teacher-forced updates, EPS_NOISE = 0) -- how typical is the one
figure the bench line reports per configuration?
usage: python tools/parity_seeds.py [n_seeds] [config ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                     # noqa: E402
import bayes_sim_ig_amd as B     # noqa: E402

B.MDNN.VERBOSE = False
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
names = sys.argv[2:] or ['cfg5', 'cfg2', 'cfg3', 'cfg4', 'cfg4b']
dev = 'cuda:0'
print('held-out NLL after 100 teacher-forced updates, |hip - oracle| / |oracle|, data seeds 0..%d' % (n_seeds - 1))
for name in names:
    cfg = dict(bench.CONFIGS[name])
    diffs = []
    for seed in range(n_seeds):
        theta, states, actions = bench.synth_pairs(cfg, 1000, seed, dev)
        r = bench.nll_check(B, cfg, theta, states, actions, dev)
        diffs.append(r['rel_diff'])
    print('%-6s max %.2e  median %.2e   [%s]' % (name, max(diffs), float(np.median(diffs)),
                                                 ' '.join('%.1e' % d for d in diffs)), flush=True)
