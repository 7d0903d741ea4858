#!/usr/bin/env python3
"""Host-side cost of one chunk of BayesSim.fit (cProfile over a fit of a bench config): what Python
does between two launches of the persistent kernel.  Usage: host_chunk_profile.py [cfg] [pairs]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                     # noqa: E402
import bayes_sim_ig_amd as B     # noqa: E402

B.MDNN.VERBOSE = False
dev = 'cuda:0'
name = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
cfg = dict(bench.CONFIGS[name])
theta, states, actions = bench.synth_pairs(cfg, n, 3, dev)
bs = bench.build_gpu_model(B, cfg, dev, 77)
bs.fit(theta, states, actions)      # warm-up: plans, graphs
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
logs = bs.fit(theta, states, actions)
t_host = time.perf_counter() - t0
pr.disable()
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print('%s, %d pairs: fit() returned after %.1f ms of host work, GPU drained at %.1f ms' % (name, n, 1e3 * t_host, 1e3 * t_all))
pstats.Stats(pr).sort_stats('cumulative').print_stats(35)
