#!/usr/bin/env python3
"""Phase breakdown of the persistent MDNN update kernel (fit_persistent_mdnn.hip)
from its wall-clock stamps (bsig_debug_persist_profile): one chunk of a bench config."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                     # noqa: E402
import bayes_sim_ig_amd as B     # noqa: E402

B.MDNN.VERBOSE = False
dev = 'cuda:0'
lib = B._lib.require_gpu()
lib.bsig_debug_persist_profile.restype = None
lib.bsig_debug_persist_profile.argtypes = [C.c_void_p]
name = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
cfg = dict(bench.CONFIGS[name])
theta, states, actions = bench.synth_pairs(cfg, 1000, 3, dev)
bs = bench.build_gpu_model(B, cfg, dev, 77)
summ = bs._summarize(states, actions, lazy=os.environ.get('LAZY') == '1')
bs.model.run_training(summ, theta, 100, 100)          # warm-up (plan, graphs)
assert lib.bsig_fit_is_persistent(bs.model._plan) == 2
buf = torch.zeros(256 * 8 * 16, dtype=torch.int64, device=dev)
lib.bsig_debug_persist_profile(buf.data_ptr())
bs.model.run_training(summ, theta, 100, 100)
torch.cuda.synchronize()
lib.bsig_debug_persist_profile(None)
st = buf.cpu().numpy().reshape(256, 8, 16).astype(np.float64) / 100.0   # 100 MHz -> us
live = [g for g in range(256) if st[g, 1, 0] > 0]
tiles = [g for g in live if st[g, 1, 12] > 0]
owners = [g for g in live if st[g, 1, 15] > 0]
small = [g for g in live if g not in tiles and g not in owners]
print('%s: %d tile + %d owner + %d small-weight workgroups; mean over updates 2..6 of the last launch'
      % (name, len(tiles), len(owners), len(small)))
rows = [('last tile wg has its summary tile in LDS', tiles, 1, np.max),
        ('last tile wg done with the forward MFMAs', tiles, 2, np.max),
        ('last forward flag raised', tiles, 3, np.max),
        ('tiles: last has the next tile prefetched (factor rows only)', tiles, 8, np.max),
        ('owners: first / last start of the update', owners, 0, np.min),
        ('owners: last start of the update', owners, 0, np.max),
        ('owners: last has the small weights flags', owners, 4, np.max),
        ('owners: wave 0 has seen the forward flags (the others fetch Wh, W2)', owners, 5, np.max),
        ('owners: first has Wh + W2 registers and the forward flags', owners, 6, np.min),
        ('owners: last has Wh + W2 registers and the forward flags', owners, 6, np.max),
        ('owners: last h1 (k-slice sum, tanh)', owners, 7, np.max),
        ('owners: last h2', owners, 8, np.max),
        ('owners: last head outputs', owners, 9, np.max),
        ('owners: last rows finished (NLL fwd/bwd)', owners, 13, np.max),
        ('owners: last d_out corrected + published', owners, 14, np.max),
        ('owners: last flag (dz2, dz1 out)', owners, 15, np.max),
        ('tiles: first released', tiles, 10, np.min),
        ('tiles: last released', tiles, 10, np.max),
        ('tiles: last dz1^T in LDS', tiles, 11, np.max),
        ('tiles: first finished the update', tiles, 12, np.min),
        ('tiles: last finished the update', tiles, 12, np.max),
        ('small: first released', small, 4, np.min),
        ('small: last inputs in LDS', small, 5, np.max),
        ('small: last weights published', small, 6, np.max)]
acc = {r[0]: [] for r in rows}
per = []
for u in range(2, 7):
    t0 = min(st[g, u, 0] for g in tiles)
    per.append(min(st[g, u + 1, 0] for g in tiles) - t0)
    for label, grp, k, fn in rows:
        vals = [st[g, u, k] for g in grp]
        # (a stamp this instantiation never writes -- the factor-row prefetch of a plan on summary rows --
        # stays 0: not a time)
        acc[label].append(fn(vals) - t0 if vals and min(vals) > 0 else float('nan'))
print('update period %.2f us' % np.mean(per) + ('   per update: ' + ' '.join('%.2f' % x for x in per) if os.environ.get('PER_UPDATE') == '1' else ''))
for label, _, _, _ in rows:
    if np.isnan(acc[label]).any():
        print('    %-68s    n/a' % label)
        continue
    print('    %-68s %6.2f' % (label, np.mean(acc[label])) + ('   [' + ' '.join('%.2f' % x for x in acc[label]) + ']' if os.environ.get('PER_UPDATE') == '1' else ''))
if os.environ.get('DETAIL') == '1':
    u = 4
    t0 = min(st[g, u, 0] for g in tiles)
    print('update %d: small-weight workgroups: released / inputs / published (us after t0), then into update %d' % (u, u + 1))
    for g in small:
        print('   wg %3d: %6.2f %6.2f %6.2f' % (g, st[g, u, 4] - t0, st[g, u, 5] - t0, st[g, u, 6] - t0))
    t1 = min(st[g, u + 1, 0] for g in tiles)
    print('update %d starts at %.2f; owners: start / small flags / Wh in LDS / fwd flags' % (u + 1, t1 - t0))
    for g in owners:
        print('   wg %3d: %6.2f %6.2f %6.2f %6.2f' % (g, st[g, u + 1, 0] - t1, st[g, u + 1, 4] - t1, st[g, u + 1, 5] - t1, st[g, u + 1, 6] - t1))
if os.environ.get('WIDE_DETAIL') == '1':
    u = 4
    t0 = min(st[g, u, 0] for g in tiles)
    print('update %d, head-block workgroups (us after t0): h2 flags seen / h2 in LDS / head outputs flagged / d_out flags seen / d_out block in LDS / dz2 share flagged / weights published' % u)
    for g in small[4:]:
        print('   wg %3d: %6.2f %6.2f %6.2f %6.2f %6.2f %6.2f %6.2f' % tuple([g] + [st[g, u, k] - t0 for k in (7, 8, 9, 4, 5, 10, 6)]) + '   [mfma issued %.2f, exchanged %.2f, stores issued %.2f, acked %.2f]' % tuple(st[g, u, k] - t0 for k in (1, 2, 11, 3)))
    print('owners: fwd flags seen / h1 / h2 (flagged) / head outputs in LDS + exp sum published / rows finished / d_out published / flag_own')
    for g in owners[:6]:
        print('   wg %3d: %6.2f %6.2f %6.2f %6.2f %6.2f %6.2f %6.2f' % tuple([g] + [st[g, u, k] - t0 for k in (6, 7, 8, 9, 13, 14, 15)]))
