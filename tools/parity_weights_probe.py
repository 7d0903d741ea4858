#!/usr/bin/env python3
"""Where do the HIP path and the fp32 oracle leave the fp64 run of a teacher-forced chunk?  Per
parameter tensor, after 1 / 5 / 20 updates: max and mean |hip - f64| against |cpu_f32 - f64|."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
import bench
import bayes_sim_ig_amd as B
from oracle import summarize as osum
B.MDNN.VERBOSE = False
B.MDNN.EPS_NOISE = 0.0
name, seed = sys.argv[1], int(sys.argv[2])
cfg = dict(bench.CONFIGS[name])
torch.set_num_threads(8)
theta, states, actions = bench.synth_pairs(cfg, 1000, seed, 'cuda:0')
s_cpu = osum.SUMMARIZERS[cfg['summarizer']](states.cpu(), actions.cpu())
for n_up in (1, 5, 20):
    ids = np.random.RandomState(5).randint(0, 800, (100, 100))[:n_up]
    bs = bench.build_gpu_model(B, cfg, 'cuda:0', 77)
    w0 = {k: v.cpu().clone() for k, v in bs.model.state_dict().items()}
    summ = bs._summarize(states, actions, lazy=True)
    bs.model.run_training(summ, theta, n_up, 100, ids_table=ids)
    hip = {k: v.cpu().double() for k, v in bs.model.state_dict().items()}
    outs = {}
    for dbl in (False, True):
        o = bench.build_oracle(cfg, s_cpu.shape[1], 77, 0.0)
        x, y, sd = s_cpu, theta.cpu(), w0
        if dbl:
            o = o.double(); sd = {k: v.double() for k, v in w0.items()}
            o.output_lows, o.output_highs = o.output_lows.double(), o.output_highs.double()
            x, y = x.double(), y.double()
        o.load_state_dict(sd)
        o.run_training(x, y, n_up, 100, ids_table=ids)
        outs[dbl] = {k: v.double() for k, v in o.state_dict().items()}
    print('after %d updates' % n_up)
    for k in hip:
        dh, dc = (hip[k] - outs[True][k]).abs(), (outs[False][k] - outs[True][k]).abs()
        step = (outs[True][k] - w0[k].double()).abs()
        print('  %-16s |hip-f64| max %.2e mean %.2e   |cpu32-f64| max %.2e mean %.2e   |f64 - start| max %.2e mean %.2e'
              % (k, dh.max(), dh.mean(), dc.max(), dc.mean(), step.max(), step.mean()))
    if n_up == 1:
        k = [kk for kk in hip if hip[kk].dim() == 2 and hip[kk].shape[1] == s_cpu.shape[1]][0]
        dh = (hip[k] - outs[True][k]).abs().flatten(); dc = (outs[False][k] - outs[True][k]).abs().flatten()
        for thr in (1e-9, 1e-7, 1e-5, 1e-4, 5e-4):
            print('    W1 elements off by more than %.0e: hip %d, cpu32 %d of %d' % (thr, int((dh > thr).sum()), int((dc > thr).sum()), dh.numel()))
