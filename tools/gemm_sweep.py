#!/usr/bin/env python3
"""Time bsig_gemm_f32 on the shapes of one update for tile / split-K choices
(BSIG_GEMM_TILE / BSIG_GEMM_SPLITS env overrides)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayes_sim_ig_amd as B   # noqa: E402

L = B._lib
lib = L.require_gpu()
dev = 'cuda:0'


def run(m, n, k, akm, bkm, epi=0, reps=100):
    a = torch.randn((k, m) if akm else (m, k), device=dev)
    b = torch.randn((k, n) if bkm else (n, k), device=dev)
    c = torch.empty(m, 2 * n if epi == L.EPI_COS_SIN else n, device=dev)
    ws = torch.empty(64 * m * n + 16, device=dev)

    def go():
        L.check(lib.bsig_gemm_f32(L.ptr(a), a.stride(0), akm, None, L.ptr(b), b.stride(0), bkm,
                                  None, L.ptr(c), c.stride(0), m, n, k, epi, 0, None, None, 0,
                                  1.0, L.ptr(ws), ws.numel() * 4, L.stream()))
    for _ in range(10):
        go()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        go()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


shapes = [('dW2  128x128x100 TN', 128, 128, 100, 1, 1, 0),
          ('dWh  175x128x100 TN', 175, 128, 100, 1, 1, 0),
          ('dX   100x128x175 NN', 100, 128, 175, 0, 1, 0),
          ('fwd2 100x128x128 NT', 100, 128, 128, 0, 0, 0),
          ('rff  100x2048x2310 NT cos/sin', 100, 2048, 2310, 0, 0, L.EPI_COS_SIN),
          ('head 100x260x4096 NT', 100, 260, 4096, 0, 0, 0),
          ('dW   260x4096x100 TN', 260, 4096, 100, 1, 1, 0),
          ('tr1  100x128x11802 NT', 100, 128, 11802, 0, 0, 0),
          ('dW1  128x11802x100 TN', 128, 11802, 100, 1, 1, 0),
          ('rffL 8192x2048x2310 NT cos/sin', 8192, 2048, 2310, 0, 0, L.EPI_COS_SIN)]
for name, m, n, k, akm, bkm, epi in shapes:
    row = []
    for tile in (0, 2, 1):
        if tile == 1 and m < 128:
            continue
        for sp in ((1,) if m > 1000 else (1, 2, 4, 8, 16, 32, 64)):
            if sp > 1 and k // sp < 64:
                continue
            os.environ['BSIG_GEMM_TILE'] = str(tile)
            os.environ['BSIG_GEMM_SPLITS'] = str(sp)
            row.append('t%d/s%d=%.1f' % (tile, sp, run(m, n, k, akm, bkm, epi, 20 if m > 1000 else 100)))
    os.environ.pop('BSIG_GEMM_TILE'); os.environ.pop('BSIG_GEMM_SPLITS')
    row.append('auto=%.1f' % run(m, n, k, akm, bkm, epi, 20 if m > 1000 else 100))
    fl = 2.0 * m * n * k
    print('%-32s %s' % (name, ' '.join(row)), '| %.2f GFLOP' % (fl / 1e9), flush=True)
