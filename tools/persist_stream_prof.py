#!/usr/bin/env python3
"""Phase breakdown of the persistent MDNN update kernel with a STREAMED first layer
(fit_persistent_mdnn_stream.hip) from its wall-clock stamps (bsig_debug_persist_profile):
one chunk of a bench config (anymal_yaml / shadow_more), the launch of updates 21..40."""
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
name = sys.argv[1] if len(sys.argv) > 1 else 'anymal_yaml'
cfg = dict(bench.CONFIGS[name])
theta, states, actions = bench.synth_pairs(cfg, 1000, 3, dev)
bs = bench.build_gpu_model(B, cfg, dev, 77)
summ = bs._summarize(states, actions, lazy=True)
bs.model.run_training(summ, theta, 100, 100)          # warm-up (plan, graphs)
assert lib.bsig_fit_is_persistent(bs.model._plan) == 2 and lib.bsig_fit_accepts_factors(bs.model._plan) == 0
buf = torch.zeros(256 * 8 * 16, dtype=torch.int64, device=dev)
lib.bsig_debug_persist_profile(buf.data_ptr())
bs.model.run_training(summ, theta, 100, 100)
torch.cuda.synchronize()
lib.bsig_debug_persist_profile(None)
st = buf.cpu().numpy().reshape(256, 8, 16).astype(np.float64) / 100.0   # 100 MHz -> us
live = [g for g in range(256) if st[g, 1, 0] > 0]
owners = [g for g in live if st[g, 1, 13] > 0 and st[g, 1, 12] == 0]
tiles = [g for g in live if g < owners[0]]
small = [g for g in live if g > owners[-1]]
print('%s: %d tile + %d owner + %d small-weight workgroups; mean over updates 2..6 of the last launch'
      % (name, len(tiles), len(owners), len(small)))
rows = [('tiles: first / last start of the update', tiles, 0, np.min),
        ('tiles: last start of the update', tiles, 0, np.max),
        ('tiles: last has the factor columns transposed', tiles, 1, np.max),
        ('tiles: last has the factor rows in LDS', tiles, 6, np.max),
        ('tiles: first released by the owners', tiles, 2, np.min),
        ('tiles: last released by the owners', tiles, 2, np.max),
        ('tiles: last has dz1 in registers, b1 step', tiles, 3, np.max),
        ('tiles: first done with the first chunk', tiles, 7, np.min),
        ('tiles: last done with the first chunk', tiles, 7, np.max),
        ('tiles: first done with the pass', tiles, 4, np.min),
        ('tiles: last done with the pass', tiles, 4, np.max),
        ('tiles: first slab out, flag', tiles, 14, np.min),
        ('tiles: last slab out, flag', tiles, 14, np.max),
        ('tiles: last has seen every slab flag', tiles, 15, np.max),
        ('tiles: first slab out + summed', tiles, 5, np.min),
        ('tiles: last slab out + summed (= next update)', tiles, 5, np.max),
        ('owners: first start of the update', owners, 0, np.min),
        ('owners: last has the small weights flags', owners, 4, np.max),
        ('owners: wave 0 has seen the sum flags', owners, 5, np.max),
        ('owners: last has Wh + W2 registers and the flags', owners, 6, np.max),
        ('owners: last h1', owners, 7, np.max),
        ('owners: last h2', owners, 8, np.max),
        ('owners: last head outputs', owners, 9, np.max),
        ('owners: last rows finished (NLL fwd/bwd)', owners, 13, np.max),
        ('owners: last d_out corrected + published', owners, 14, np.max),
        ('owners: last flag (dz2, dz1 out)', owners, 15, np.max),
        ('small: last weights published', small, 6, np.max)]
def rel(v, t0, w=7):
    """a stamp relative to t0; a stamp that was never taken (0) prints as n/a"""
    return ('%%%d.2f' % w) % (v - t0) if v > 0 else ' ' * (w - 3) + 'n/a'


acc = {r[0]: [] for r in rows}
per = []
for u in range(2, 7):
    t0 = min(st[g, u, 0] for g in tiles)
    per.append(min(st[g, u + 1, 0] for g in tiles) - t0)
    for label, grp, k, fn in rows:
        acc[label].append(fn([st[g, u, k] for g in grp]) - t0)
print('update period %.2f us   per update: ' % np.mean(per) + ' '.join('%.2f' % x for x in per))
for label, _, _, _ in rows:
    print('    %-60s %7.2f' % (label, np.mean(acc[label])))
u = 4
t0 = min(st[g, u, 0] for g in tiles)
print('second chunk of update %d, wavefront 0 of four tile workgroups: start / dW MFMAs done / Adam, stores, W in LDS / '
      'behind the barrier / forward MFMAs done / behind the barrier' % u)
for g in tiles[3::55]:
    print('   wg %3d: ' % g + ' '.join(rel(st[g, u, k], t0) for k in range(8, 14)))
t0w = st[254, 0, 0]
if t0w > 0:
    print('tile workgroup 3, update 4, second chunk, per wavefront (us after wavefront 0 entered): entered / dW || Adam done / '
          'behind the barrier / forward done (stores, loads issued inside) / behind the barrier / gradient in LDS')
    for w in range(8):
        print('   wave %d: ' % w + ' '.join(rel(st[254, w, k], t0w, 6) for k in (0, 1, 2, 7, 8, 9)) +
              '   forward: %d shader clocks in %.2f us = %.2f GHz' % ((st[254, w, 11] - st[254, w, 10]) * 100, st[254, w, 7] - st[254, w, 2], (st[254, w, 11] - st[254, w, 10]) * 100 / max(st[254, w, 7] - st[254, w, 2], 1e-9) / 1e3))
if os.environ.get('DETAIL') == '1':
    u = 4
    t0 = min(st[g, u, 0] for g in tiles)
    print('update %d, tile workgroups: start / prefetched / released / dz1 / pass done / summed' % u)
    for g in tiles[::8]:
        print('   wg %3d: ' % g + ' '.join(rel(st[g, u, k], t0) for k in range(6)))
if os.environ.get('WIDE_DETAIL') == '1':
    u = 4
    t0 = min(st[g, u, 0] for g in tiles)
    print('update %d, head-block workgroups (us after t0): h2 flags seen / h2 in LDS / head outputs flagged / d_out flags seen / '
          'd_out block in LDS / dz2 share flagged / weights published' % u)
    for g in small[4:]:
        print('   wg %3d: ' % g + ' '.join(rel(st[g, u, k], t0, 6) for k in (7, 8, 9, 4, 5, 10, 6)) +
              '   [mfma issued %s, exchanged %s, stores issued %s, acked %s]' % tuple(rel(st[g, u, k], t0, 1).strip() for k in (1, 2, 11, 3)))
    print('owners: sum flags seen / h1 / h2 (flagged) / head outputs in LDS + exp sum published / rows finished / d_out published / flag_own')
    for g in owners[:6]:
        print('   wg %3d: ' % g + ' '.join(rel(st[g, u, k], t0, 6) for k in (6, 7, 8, 9, 13, 14, 15)))
