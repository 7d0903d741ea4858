import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import bayes_sim_ig_amd as B
B.MDNN.VERBOSE = False
import test_gpu_fit as T
for name in (sys.argv[1:] or ['cfg4b', 'cfg4']):
    for seed in range(6):
        out = {}
        for fast in ('1', '0'):
            hip, f32, f64, _ = T._bracket_chunk(B, name, seed, False, must_factor=False, hip_env={'BSIG_MDNN_FAST_ROWS': '2' if fast == '1' else '0'})
            out[fast] = hip
        g1, g0, r, r64 = out['1']['test_loss'][-1], out['0']['test_loss'][-1], f32['test_loss'][-1], f64['test_loss'][-1]
        print('%s seed %d: f64 %.6f | cpu32-f64 %.2e | hip(fast)-f64 %.2e | hip(generic)-f64 %.2e | hip(fast)-cpu32 rel %.2e | hip(generic)-cpu32 rel %.2e'
              % (name, seed, r64, (r - r64), (g1 - r64), (g0 - r64), abs(g1 - r) / abs(r), abs(g0 - r) / abs(r)), flush=True)
