#!/usr/bin/env python3
"""A/B of two builds of the library (BSIG_LIB_PATH): one teacher-forced chunk of a bench config,
prints a hash of the trained parameters and of the logged losses + the time per update of the
chunk's launch.  Same hash = the change is bit-neutral.
usage: [BSIG_LIB_PATH=...] python tools/ab_bitwise.py cfg3 [eps_noise]"""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                     # noqa: E402
import bayes_sim_ig_amd as B     # noqa: E402

B.MDNN.VERBOSE = False
name = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
eps = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
B.MDNN.EPS_NOISE = eps
dev = 'cuda:0'
cfg = dict(bench.CONFIGS[name])
theta, states, actions = bench.synth_pairs(cfg, 1000, 3, dev)
bs = bench.build_gpu_model(B, cfg, dev, 77)
summ = bs._summarize(states, actions, lazy=True) if cfg.get('summarizer') == 'summary_corrdiff' else bs._summarize(states, actions)
ids = np.random.RandomState(5).randint(0, 800, (100, 100))
torch.manual_seed(3)
logs = bs.model.run_training(summ, theta, 100, 100, ids_table=ids)
flat = bs.model._flat.detach().cpu().numpy()
h = hashlib.sha256(flat.tobytes()).hexdigest()[:16]
hl = hashlib.sha256(np.asarray(logs['train_loss'] + logs['test_loss'], dtype=np.float64).tobytes()).hexdigest()[:16]
# time: three more calls
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    bs.model.run_training(summ, theta, 100, 100, ids_table=ids)
e1.record(); torch.cuda.synchronize()
print('%s eps=%g lib=%s: weights %s logs %s test_loss[-1] %.9g   %.2f us per update (whole call / 100)'
      % (name, eps, os.environ.get('BSIG_LIB_PATH', 'in-tree'), h, hl, logs['test_loss'][-1], e0.elapsed_time(e1) * 10 / 5))
