#!/usr/bin/env python3
"""Phase breakdown of the persistent update kernel of the linear heads from its wall-clock
stamps (bsig_debug_persist_profile): one cfg5-shaped chunk.  fit_persistent.hip (unified
workgroups: tile workgroups, row owners on CUs of their own where the chip has them)."""
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
name = sys.argv[1] if len(sys.argv) > 1 else 'cfg5'
cfg = dict(bench.CONFIGS[name])
theta, states, actions = bench.synth_pairs(cfg, 1000, 3, dev)
bs = bench.build_gpu_model(B, cfg, dev, 77)
summ = bs._summarize(states, actions)
bs.model.run_training(summ, theta, 100, 100)          # warm-up (plan, graphs)
buf = torch.zeros(2 * 256 * 8 * 16, dtype=torch.int64, device=dev)
lib.bsig_debug_persist_profile(buf.data_ptr())
bs.model.run_training(summ, theta, 100, 100)
torch.cuda.synchronize()
lib.bsig_debug_persist_profile(None)
raw = buf.cpu().numpy().reshape(2, 256, 8, 16).astype(np.float64) / 100.0   # 100 MHz -> us
st, rowst = raw[0], raw[1]
live = [g for g in range(256) if st[g, 1, 0] > 0]
v1 = False      # (fit_persistent_v1.hip was retired in round 6)


def phases(g, table):
    for n, a, b in table:
        print('    %-22s %6.2f us' % (n, (st[g, 1:8, b] - st[g, 1:8, a]).mean()))


if v1:
    owners = [g for g in live if st[g, 1, 4] > 0]
    tiles = [g for g in live if st[g, 1, 4] == 0]
    print('%s (v1): %d tile workgroups + %d row owners; updates 1..7 of the last launch' % (name, len(tiles), len(owners)))
    for g in (tiles[0], tiles[-1]):
        print('tile wg %3d: update = %.1f us' % (g, np.mean(st[g, 2:8, 0] - st[g, 1:7, 0])))
        phases(g, (('feat tile -> LDS', 0, 1), ('fwd mfma', 1, 2), ('slab store+flag', 2, 3),
                   ('wait for owners', 3, 10), ('dO^T load', 10, 11), ('dW mfma+adam', 11, 12)))
    for g in (owners[0], owners[-1]):
        print('owner wg %3d: update = %.1f us' % (g, np.mean(st[g, 2:8, 0] - st[g, 1:7, 0])))
        phases(g, (('y + wait fwd flags', 0, 4), ('slab sum', 4, 5), ('row: pre-eps part', 5, 6),
                   ('eps gather + row', 6, 7), ('dO/E store+publish', 7, 9)))
        print('    %-22s %6.2f us' % ('idle until next', (st[g, 2:8, 0] - st[g, 1:7, 9]).mean()))
    rows = [('last tile wg has its feature tile in LDS', tiles, 1, np.max),
            ('last tile wg done with the forward MFMAs', tiles, 2, np.max),
            ('last forward flag raised', tiles, 3, np.max)]
else:
    owners = [g for g in live if st[g, 1, 4] > 0]
    tiles = [g for g in live if st[g, 1, 3] > 0]
    print('%s: %d workgroups with a weight tile, %d of the launch own a minibatch row; updates %d + 1..7 of the last launch'
          % (name, len(tiles), len(owners), int(os.environ.get('BSIG_PROF_T0', '0'))))
    both = [g for g in owners if g in tiles]
    only_t = [g for g in tiles if g not in owners]
    for g in both[:1] + both[-1:]:
        print('tile + row wg %3d: update = %.1f us' % (g, np.mean(st[g, 2:8, 0] - st[g, 1:7, 0])))
        phases(g, (('fwd mfma (wave 0)', 0, 1), ('slab store+flag', 1, 3), ('wait fwd flags', 3, 4),
                   ('slab sum', 4, 5), ('row (diag_row)', 5, 7),
                   ('dO/E store+publish', 7, 9), ('next tile requested', 9, 2), ('wait for owners', 2, 10), ('dO^T load', 10, 11),
                   ('dW mfma+adam (wave 0)', 11, 8), ('end barrier', 8, 12)))
    only_o = [g for g in owners if g not in tiles]
    for g in only_o[:1] + only_o[-1:]:
        print('row wg %3d (no tile): update = %.1f us' % (g, np.mean(st[g, 2:8, 0] - st[g, 1:7, 0])))
        phases(g, (('wait fwd flags', 0, 4), ('slab sum', 4, 5), ('row (diag_row)', 5, 7),
                   ('dO/E store+publish', 7, 9)))
        print('    %-22s %6.2f us' % ('idle until next', (st[g, 2:8, 0] - st[g, 1:7, 9]).mean()))
    for g in only_t[:1] + only_t[-1:]:
        print('tile wg %3d: update = %.1f us' % (g, np.mean(st[g, 2:8, 0] - st[g, 1:7, 0])))
        phases(g, (('fwd mfma (wave 0)', 0, 1), ('slab store+flag', 1, 3), ('next tile requested', 3, 2), ('wait for owners', 2, 10),
                   ('dO^T load', 10, 11), ('dW mfma+adam (wave 0)', 11, 8),
                   ('end barrier', 8, 12)))
    rows = [('last wg done with the forward MFMAs (wave 0)', tiles, 1, np.max),
            ('last forward flag raised', tiles, 3, np.max)]

# chip-level critical path of one update (wall_clock64 is one 100 MHz counter for the chip)
print('chip-level timeline, mean over updates 2..6 of the launch (us after the first tile workgroup starts the update):')
rows += [('first owner released', owners, 4, np.min),
         ('last owner released', owners, 4, np.max),
         ('last owner has its row (slab sum, exp published)', owners, 5, np.max),
         ('last row finished', owners, 7, np.max),
         ('last owner published d_out', owners, 9, np.max),
         ('first tile wg released', tiles, 10, np.min),
         ('last tile wg released', tiles, 10, np.max),
         ('last d_out^T tile in LDS', tiles, 11, np.max),
         ('first tile wg finished the update', tiles, 12, np.min),
         ('last tile wg finished the update', tiles, 12, np.max)]
acc = {r[0]: [] for r in rows}
for u in range(2, 7):
    t0 = min(st[g, u, 0] for g in tiles)
    for label, grp, k, fn in rows:
        acc[label].append(fn([st[g, u, k] for g in grp]) - t0)
for label, _, _, _ in rows:
    print('    %-52s %6.2f' % (label, np.mean(acc[label])))

if os.environ.get('FAST_ROW_STAMPS') == '1' and rowst.any():
    g = owners[0]
    names = ['(stamp 4 ->) k-slice loads back, partials in LDS', 'barrier', 'row image read, exp published, mixture weights',
             'eps get', 'sigma, z, log, DPP sums, LDS write', 'barrier (partial sums of the two wavefronts)',
             'logp, logsumexp, backward', 'quad stores, uds', 'drain (vmcnt 0)']
    print('inside the fast row (wg %d, wavefront 0):' % g)
    prev = st[g, 1:8, 4]
    for i, n in enumerate(names):
        cur = rowst[g, 1:8, i]
        print('    %-52s %6.2f us' % (n, (cur - prev).mean()))
        prev = cur
    print('    %-52s %6.2f us' % ('barrier + publish (-> stamp 9)', (st[g, 1:8, 9] - prev).mean()))
elif rowst.any():      # a BSIG_ROW_PROF build: stamps inside diag_row (wave 0 of the first row owner)
    g = owners[0]
    names = ['LDS reads issued', 'exp(pre), y - mu, noise', 'mixture weights', 'eps get (gather)', 'sigma, z, log',
             'sums over d (LDS)', 'logsumexp', 'backward']
    print('inside diag_row (wg %d):' % g)
    for i, n in enumerate(names):
        print('    %-26s %6.2f us' % (n, (rowst[g, 1:8, i + 1] - rowst[g, 1:8, i]).mean()))
    print('    %-26s %6.2f us' % ('stamp 5 -> diag_row start', (rowst[g, 1:8, 0] - st[g, 1:8, 5]).mean()))
    print('    %-26s %6.2f us' % ('diag_row end -> stamp 7', (st[g, 1:8, 7] - rowst[g, 1:8, 8]).mean()))
