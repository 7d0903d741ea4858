#!/usr/bin/env python3
"""Teacher-forced first chunk of a bench config, HIP path vs the oracle (test infrastructure): the
relative difference of every logged loss -- shows whether a held-out NLL gap is there from update 0
(a bug) or grows along the trajectory (sensitivity of the fit to rounding).
Usage: parity_trajectory.py [cfg] [n_updates]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                     # noqa: E402
import bayes_sim_ig_amd as B     # noqa: E402
from oracle import summarize as osum   # noqa: E402

B.MDNN.VERBOSE = False
B.MDNN.EPS_NOISE = 0.0
dev = 'cuda:0'
name = sys.argv[1] if len(sys.argv) > 1 else 'shadow_more'
nup = int(sys.argv[2]) if len(sys.argv) > 2 else 100
cfg = dict(bench.CONFIGS[name])
theta, states, actions = bench.synth_pairs(cfg, 1000, 3, dev)
torch.set_num_threads(8)
bs = bench.build_gpu_model(B, cfg, dev, 77)
ids = np.random.RandomState(5).randint(0, 800, (nup, 100))
summ = bs._summarize(states, actions)
got = bs.model.run_training(summ, theta, nup, 100, ids_table=ids)
ora = bench.build_oracle(cfg, summ.shape[1], 77, 0.0,
                         freqs=bs.model.rff.freqs.cpu().numpy() if cfg['model'] == 'MDRFF' else None)
bs2 = bench.build_gpu_model(B, cfg, dev, 77)
ora.load_state_dict({k: v.cpu() for k, v in bs2.model.state_dict().items()})
if cfg['model'] == 'MDRFF':
    ora.rff.freqs = bs2.model.rff.freqs.cpu()
ref = ora.run_training(osum.SUMMARIZERS[cfg['summarizer']](states.cpu(), actions.cpu()), theta.cpu(), nup, 100,
                       ids_table=ids)
g, r = np.asarray(got['train_loss']), np.asarray(ref['train_loss'])
rel = np.abs(g - r) / np.maximum(np.abs(r), 1e-12)
print('%s: train loss, HIP vs oracle' % name)
every = max(nup // 5, 1)
its = [it for it in range(nup) if it % every == 0 or it + 1 == nup]
for j, it in enumerate(its[:len(g)]):
    print('  after update %3d: hip %.7g  oracle %.7g  rel %.2e' % (it, g[j], r[j], rel[j]))
print('held-out loss (hip / oracle):', ['%.6g / %.6g' % (a, b) for a, b in zip(got['test_loss'], ref['test_loss'])])
w_rel = []
sd = {k: v.cpu() for k, v in bs.model.state_dict().items()}
for k, v in ora.state_dict().items():
    w_rel.append((k, float((sd[k] - v).abs().max() / v.abs().max().clamp_min(1e-12))))
print('max |dW| / max |W| per tensor after the chunk:', ', '.join('%s %.1e' % kv for kv in w_rel))
