#!/usr/bin/env python3
"""Where is the fp32 noise floor of a teacher-forced chunk?  HIP (fp32) and the
fp32 CPU oracle are both compared with the SAME chunk run by the oracle in
fp64 (identical start weights / ids, EPS_NOISE=0)."""
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
torch.set_num_threads(8)
for name in sys.argv[1:] or ['cfg3', 'cfg5']:
    cfg = dict(bench.CONFIGS[name])
    for seed in (3, 4, 5):
        theta, states, actions = bench.synth_pairs(cfg, 1000, seed, dev)
        ids = np.random.RandomState(5).randint(0, 800, (100, 100))
        bs = bench.build_gpu_model(B, cfg, dev, 77)
        w0 = {k: v.cpu().clone() for k, v in bs.model.state_dict().items()}
        summ = bs._summarize(states, actions)
        hip = bs.model.run_training(summ, theta, 100, 100, ids_table=ids)['test_loss']
        freqs = bs.model.rff.freqs.cpu().numpy() if cfg['model'] == 'MDRFF' else None
        s_cpu = osum.SUMMARIZERS[cfg['summarizer']](states.cpu(), actions.cpu())
        out = {}
        for tag, dt in (('f32', torch.float32), ('f64', torch.float64)):
            o = bench.build_oracle(cfg, summ.shape[1], 77, 0.0, freqs=freqs)
            o.load_state_dict(w0)
            if dt == torch.float64:
                o = o.double()
                o.output_lows, o.output_highs = o.output_lows.double(), o.output_highs.double()
                if freqs is not None:
                    o.rff.freqs, o.rff.sigma = o.rff.freqs.double(), o.rff.sigma.double()
            out[tag] = o.run_training(s_cpu.to(dt), theta.cpu().to(dt), 100, 100, ids_table=ids)['test_loss']
        r = lambda a, b: abs(a[-1] - b[-1]) / abs(b[-1])   # noqa: E731
        print('%s seed %d  NLL f64 %.6f | hip-f64 %.2e  cpu32-f64 %.2e  hip-cpu32 %.2e'
              % (name, seed, out['f64'][-1], r(hip, out['f64']), r(out['f32'], out['f64']),
                 r(hip, out['f32'])), flush=True)
