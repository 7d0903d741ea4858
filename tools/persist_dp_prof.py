#!/usr/bin/env python3
"""Phase breakdown of ONE data-parallel launch of the persistent update kernel
(fit_persistent.hip, one update per launch: pending Adam step in, gradients out)."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                     # noqa: E402
import bayes_sim_ig_amd as B     # noqa: E402

B.MDNN.VERBOSE = False
dev = 'cuda:0'
lib = B._lib.require_gpu()
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device(dev))
cfg = dict(bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'cfg5'])
theta, states, actions = bench.synth_pairs(cfg, 1000, 3, dev)
bs = bench.build_gpu_model(B, cfg, dev, 77)
bs.model.enable_data_parallel()
summ = bs._summarize(states, actions)
bs.model.run_training(summ, theta, 100, 100)
buf = torch.zeros(256 * 8 * 16, dtype=torch.int64, device=dev)
lib.bsig_debug_persist_profile(buf.data_ptr())
bs.model.run_training(summ, theta, 100, 100)
torch.cuda.synchronize()
lib.bsig_debug_persist_profile(None)
st = buf.cpu().numpy().reshape(256, 8, 16).astype(np.float64) / 100.0
tiles = [g for g in range(256) if st[g, 0, 14] > 0 and st[g, 0, 12] > 0]
t0 = min(st[g, 0, 14] for g in tiles)
print('%d tile workgroups; the last launch that ran an update; us after the first workgroup entered the kernel' % len(tiles))
for label, k, fn in (('last workgroup entered', 14, np.max), ('first has W, m, v, g loaded (+ pending Adam stored)', 0, np.min),
                     ('last has W, m, v, g loaded (+ pending Adam stored)', 0, np.max),
                     ('last feature tile in LDS', 1, np.max), ('last forward flag', 3, np.max),
                     ('last released by the owners', 10, np.max), ('last dW done (gradients stored)', 12, np.max),
                     ('last out of the kernel', 13, np.max)):
    print('    %-56s %6.2f' % (label, fn([st[g, 0, k] for g in tiles]) - t0))
dist.destroy_process_group()
