#!/usr/bin/env python3
"""Wall-clock pieces of one scaled-batch fit (all pairs as one chunk, minibatch 8192)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                     # noqa: E402
import bayes_sim_ig_amd as B     # noqa: E402

B.MDNN.VERBOSE = False
dev = 'cuda:0'
cfg = dict(bench.CONFIGS['cfg5'])
theta, states, actions = bench.synth_pairs(cfg, 100000, 3, dev)
for in_place in (True, False, True):
    B.MDNN.BIND_IN_PLACE_BYTES = (64 << 20) if in_place else (1 << 60)
    bs = bench.build_gpu_model(B, cfg, dev, 4321)
    np.random.seed(4321)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        summ = bs._summarize(states, actions)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        pend = bs.model.run_training(summ, theta, 122, 8192, test_frac=0.2, _defer=True)
        t2 = time.perf_counter()
        torch.cuda.synchronize(); t3 = time.perf_counter()
        pend.result()
        t4 = time.perf_counter()
        print('in_place=%d rep %d: summarize %.2f ms, run_training returns after %.2f ms, GPU drained +%.2f ms, logs +%.2f ms; total %.2f ms'
              % (in_place, rep, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3), 1e3 * (t4 - t0)))
