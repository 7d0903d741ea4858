#!/usr/bin/env python3
"""Per-workgroup spread of the persistent kernel's standard stamps (cfg5): who finishes an update last?"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench                     # noqa: E402
import bayes_sim_ig_amd as B     # noqa: E402

B.MDNN.VERBOSE = False
dev = 'cuda:0'
lib = B._lib.require_gpu()
lib.bsig_debug_persist_profile.restype = None
lib.bsig_debug_persist_profile.argtypes = [C.c_void_p]
name = sys.argv[1] if len(sys.argv) > 1 else 'cfg5'
cfg = dict(bench.CONFIGS[name])
theta, states, actions = bench.synth_pairs(cfg, 1000, 3, dev)
bs = bench.build_gpu_model(B, cfg, dev, 77)
summ = bs._summarize(states, actions)
bs.model.run_training(summ, theta, 100, 100)
buf = torch.zeros(2 * 256 * 8 * 16, dtype=torch.int64, device=dev)
lib.bsig_debug_persist_profile(buf.data_ptr())
bs.model.run_training(summ, theta, 100, 100)
torch.cuda.synchronize()
lib.bsig_debug_persist_profile(None)
st = buf.cpu().numpy().reshape(2, 256, 8, 16).astype(np.float64)[0] / 100.0
tiles = [g for g in range(256) if st[g, 1, 3] > 0]
nb_count = 9
print('per tile workgroup, mean over updates 2..6, us relative to the first workgroup starting the update')
print(' wg  ks nb | start  fwd-done flag  | released dO^T-in  dW-done(w0) end')
rows = []
for g in tiles:
    v = []
    for u in range(2, 7):
        t0 = min(st[h, u, 0] for h in tiles)
        v.append([st[g, u, k] - t0 for k in (0, 1, 3, 10, 11, 8, 12)])
    rows.append((g, np.mean(v, axis=0)))
for g, m in rows:
    if g < 12 or g % 9 == 0 or g > 188:
        print('%3d %3d %2d | %5.2f %5.2f %5.2f | %5.2f %5.2f %5.2f %5.2f   barrier %.2f' % (g, g // nb_count, g % nb_count, *m, m[6] - m[5]))
a = np.array([m for _, m in rows])
for k, n in enumerate(('start', 'fwd-done', 'flag', 'released', 'dO^T-in', 'dW-done(w0)', 'end')):
    print('%-12s min %.2f  median %.2f  max %.2f  (argmax wg %d)' % (n, a[:, k].min(), np.median(a[:, k]), a[:, k].max(), tiles[int(a[:, k].argmax())]))
b = a[:, 6] - a[:, 5]
print('end barrier: median %.2f, > 0.3 us in workgroups %s' % (np.median(b), [tiles[i] for i in np.nonzero(b > 0.3)[0]]))
