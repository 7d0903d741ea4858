#!/usr/bin/env python3
"""Every estimator configuration the reference ships (bayes_sim_ig/cfg/*.yaml, `bayessim:` block +
the number of `realParams` means) through a short fit on one MI355X: pairs/s in fit() under the
reference chunk protocol, which update path the plan takes, and the teacher-forced held-out NLL
against the oracle (test infrastructure).  All twelve are MDNN [128, 128], diagonal covariance
(bayes_sim.py:61-63), lr 1e-4 (the bench uses 1e-3 like its other configs).
Observation / action widths come from the Isaac Gym task classes, which are not in the reference
tree ([ext]): they only set the summary width I.
Usage: yaml_configs_bench.py [name ...]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                     # noqa: E402
import bayes_sim_ig_amd as B     # noqa: E402

# name: (yaml lines of the bayessim block, obs, act [ext], D = len(realParams.means[0]), K, summarizer, trainTrajLen)
YAMLS = {
    'ant':              ('cfg/ant.yaml:65-71', 60, 8, 17, 10, 'summary_corrdiff', 50),
    'anymal':           ('cfg/anymal.yaml:103-109', 48, 12, 13, 10, 'summary_corrdiff', 20),
    'ball_balance':     ('cfg/ball_balance.yaml:48-54', 24, 3, 7, 10, 'summary_start', 10),
    'cartpole':         ('cfg/cartpole.yaml:41-47', 4, 1, 13, 10, 'summary_corrdiff', 20),
    'cartpole_more':    ('cfg/cartpole_more.yaml:41-47', 4, 1, 13, 10, 'summary_signatory', 20),
    'franka_cabinet':   ('cfg/franka_cabinet.yaml:63-69', 23, 9, 19, 10, 'summary_start', 10),
    'humanoid':         ('cfg/humanoid.yaml:76-82', 108, 21, 37, 10, 'summary_start', 10),
    'ingenuity':        ('cfg/ingenuity.yaml:42-48', 13, 6, 9, 10, 'summary_start', 10),
    'pendulum':         ('cfg/pendulum.yaml:18-24', 3, 1, 2, 10, 'summary_start', 20),
    'quadcopter':       ('cfg/quadcopter.yaml:42-48', 21, 12, 9, 10, 'summary_start', 10),
    'shadow_hand':      ('cfg/shadow_hand.yaml:76-82', 211, 20, 32, 4, 'summary_start', 10),
    'shadow_hand_more': ('cfg/shadow_hand_more.yaml:76-83', 211, 20, 32, 10, 'summary_corrdiff', 50),
}
KIND = {0: 'per-phase kernels', 1: 'persistent (linear heads)', 2: 'persistent MDNN'}

B.MDNN.VERBOSE = False
dev = 'cuda:0'
lib = B._lib.require_gpu()
names = sys.argv[1:] or list(YAMLS)
rows = []
for name in names:
    ref, sd, ad, d, k, summ, tlen = YAMLS[name]
    cfg = dict(task=name, model='MDNN', summarizer=summ, t=tlen + 1, sd=sd, ad=ad, d=d, k=k,
               hidden=[128, 128], n_feat=0)
    n = 5000 if name in ('shadow_hand_more', 'anymal') else 20000
    theta, states, actions = bench.synth_pairs(cfg, n, 7, dev)
    bs = bench.build_gpu_model(B, cfg, dev, 1234)
    np.random.seed(1234)
    bs.fit(theta, states, actions)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bs.fit(theta, states, actions)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    m = bs.model
    kind = int(lib.bsig_fit_is_persistent(m._plan))
    nh = k * (1 + 2 * d)
    path = KIND[kind] + (', wide heads' if kind == 2 and nh > 272 else '')
    nll = bench.nll_check(B, cfg, theta, states, actions, dev)
    row = {'yaml': ref, 'obs_act_ext': [sd, ad], 'D': d, 'K': k, 'Nh': nh, 'summarizer': summ,
           'I': m.input_dim, 'pairs': n, 'pairs_per_s': n / dt, 'update_path': path,
           'nll_rel_diff_100_updates': nll['rel_diff']}
    rows.append(row)
    print('%-17s I=%6d D=%2d Nh=%3d  %9.0f pairs/s  %-36s NLL rel diff %.1e' %
          (name, m.input_dim, d, nh, n / dt, path, nll['rel_diff']), flush=True)
    del bs, theta, states, actions
    torch.cuda.empty_cache()
print(json.dumps(rows))
