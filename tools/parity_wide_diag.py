#!/usr/bin/env python3
"""Where do the HIP path and the reference's fp32 arithmetic leave the fp64 run of a teacher-forced
chunk of the wide cross-correlation configs (cfg/anymal.yaml, cfg/shadow_hand_more.yaml)?  The fp32
oracle is run in several EVALUATION ORDERS: as is, and with the input columns (and the first layer's
weight columns with them) permuted -- the same network, the same fp32 operations, another grouping
of the 56-105 k-term sums (8 vs 1 threads changes next to nothing: same blocking).
usage: parity_wide_diag.py <config> <seed> [n_perms] [ENV=VALUE ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bayes_sim_ig_amd as B      # noqa: E402
import bench                      # noqa: E402
from oracle import summarize as osum   # noqa: E402

B.MDNN.VERBOSE = False
B.MDNN.EPS_NOISE = 0.0
name, seed = sys.argv[1], int(sys.argv[2])
n_perms = int(sys.argv[3]) if len(sys.argv) > 3 and '=' not in sys.argv[3] else 2
env = dict(kv.split('=') for kv in sys.argv[3:] if '=' in kv)
lazy = 'BSIG_NO_PERSISTENT' not in env
cfg = dict(bench.CONFIGS[name])
torch.set_num_threads(8)
theta, states, actions = bench.synth_pairs(cfg, 1000, seed, 'cuda:0')
ids = np.random.RandomState(5).randint(0, 800, (100, 100))
bs = bench.build_gpu_model(B, cfg, 'cuda:0', 77)
w0 = {k: v.cpu().clone() for k, v in bs.model.state_dict().items()}
summ = bs._summarize(states, actions, lazy=lazy)
os.environ.update(env)
hip = bs.model.run_training(summ, theta, 100, 100, ids_table=ids)
torch.cuda.synchronize()
s_cpu = osum.SUMMARIZERS[cfg['summarizer']](states.cpu(), actions.cpu())
k1 = [k for k, v in w0.items() if v.dim() == 2 and v.shape[1] == s_cpu.shape[1]][0]


def run32(perm):
    o = bench.build_oracle(cfg, s_cpu.shape[1], 77, 0.0)
    w = dict(w0)
    x = s_cpu
    if perm is not None:
        w[k1] = w0[k1][:, perm].contiguous()
        x = s_cpu[:, perm].contiguous()
    o.load_state_dict(w)
    return o.run_training(x, theta.cpu(), 100, 100, ids_table=ids)


f32 = [('cpu32      ', run32(None))]
for i in range(n_perms):
    perm = torch.from_numpy(np.random.RandomState(100 + i).permutation(s_cpu.shape[1]))
    f32.append(('cpu32 perm%d' % i, run32(perm)))
o64 = bench.build_oracle(cfg, s_cpu.shape[1], 77, 0.0).double()
o64.load_state_dict({k: v.double() for k, v in w0.items()})
o64.output_lows, o64.output_highs = o64.output_lows.double(), o64.output_highs.double()
f64 = o64.run_training(s_cpu.double(), theta.cpu().double(), 100, 100, ids_table=ids)
np.set_printoptions(linewidth=200, precision=3)
for key in ('test_loss', 'train_loss'):
    r = np.asarray(f64[key])
    print(key, 'f64 chunk      ', r)
    print(key, 'hip         -f64', (np.asarray(hip[key]) - r) / np.abs(r))
    for tag, f in f32:
        print(key, tag, '-f64', (np.asarray(f[key]) - r) / np.abs(r))
