#!/usr/bin/env python3
"""Phase breakdown of the persistent update kernel of a data-parallel rank that stays RESIDENT across
the gradient exchange (fit_persistent.hip, XR instantiation; BSIG_DP_RESIDENT=1): a 1-rank RCCL group,
one cfg5-shaped chunk, from the kernel's wall-clock stamps."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

os.environ['BSIG_DP_RESIDENT'] = '1'      # (the default on a 1-rank group)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                     # noqa: E402
import bayes_sim_ig_amd as B     # noqa: E402

B.MDNN.VERBOSE = False
dev = 'cuda:0'
lib = B._lib.require_gpu()
lib.bsig_debug_persist_profile.restype = None
lib.bsig_debug_persist_profile.argtypes = [C.c_void_p]
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29534')
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device(dev))
if len(sys.argv) < 2:
    sys.argv.append('cfg5')
cfg = dict(bench.CONFIGS[sys.argv[1]])
theta, states, actions = bench.synth_pairs(cfg, 2000, 3, dev)
bs = bench.build_gpu_model(B, cfg, dev, 77)
bs.model.enable_data_parallel()
bs.fit(theta, states, actions)       # (the fit's own chunk protocol: pre-projected features)
buf = torch.zeros(2 * 256 * 8 * 16, dtype=torch.int64, device=dev)
lib.bsig_debug_persist_profile(buf.data_ptr())
logs = bs.fit(theta, states, actions)
torch.cuda.synchronize()
lib.bsig_debug_persist_profile(None)
st = buf.cpu().numpy().reshape(2, 256, 8, 16).astype(np.float64)[0] / 100.0
tiles = [g for g in range(256) if st[g, 1, 13] > 0 and st[g, 1, 12] > 0 and
         (cfg['model'] != 'MDNN' or (st[g, 1, 11] > 0 and st[g, 1, 15] == 0))]      # (MDNN: the owners stamp 13 / 14 too)
if not tiles:
    sys.exit('no stamps of a resident launch: the call ran one launch per update (BSIG_DP_XR_TRACE=1 shows the probes)')
mdnn = cfg['model'] == 'MDNN'
T0 = os.environ.get('BSIG_PROF_T0', '0')
if mdnn:
    # (fit_persistent_mdnn.hip: stamps of the tile workgroups; the small-weight workgroups hand off too)
    print('%s: %d tile workgroups (resident across the exchange); updates %s + 1..7' % (sys.argv[1], len(tiles), T0))
    table = (('forward MFMAs', 0, 2), ('slab store + flag', 2, 3), ('wait for the owners', 3, 10),
             ('dz1^T load', 10, 11), ('dW1 MFMAs + gradients out, acked', 11, 13), ('exchange wait', 13, 14),
             ('reduced gradients in + Adam', 14, 12))
    rel = 10
else:
    owners = [g for g in range(256) if st[g, 1, 4] > 0 and g not in tiles]
    print('%d tile workgroups (resident across the exchange), %d row owners; updates %s + 1..7' % (len(tiles), len(owners), T0))
    table = (('fwd mfma (wave 0)', 0, 1), ('slab store+flag', 1, 3), ('next tile requested', 3, 2),
             ('wait for owners', 2, 10), ('dO^T load', 10, 11), ('dW mfma + gradients out, acked', 11, 13),
             ('exchange wait', 13, 14), ('reduced gradients in + Adam', 14, 8), ('end barrier', 8, 12))
    rel = 10
for g in tiles[:1] + tiles[-1:]:
    print('tile wg %3d: update = %.1f us' % (g, np.mean(st[g, 2:8, 0] - st[g, 1:7, 0])))
    for n, a, b in table:
        print('    %-34s %6.2f us' % (n, (st[g, 1:8, b] - st[g, 1:8, a]).mean()))
u = slice(2, 7)
t0 = np.min(st[tiles][:, u, 0], axis=0)
print('chip-level, mean over updates 2..6 (us after the first tile workgroup starts the update):')
for label, k, fn in (('last tile wg released by the owners', rel, np.max),
                     ('first tile wg has its gradients out', 13, np.min),
                     ('last tile wg has its gradients out', 13, np.max),
                     ('first tile wg sees the exchange done', 14, np.min),
                     ('last tile wg sees the exchange done', 14, np.max),
                     ('last tile wg finished the update', 12, np.max)):
    print('    %-56s %6.2f' % (label, np.mean(fn(st[tiles][:, u, k], axis=0) - t0)))
dist.destroy_process_group()
