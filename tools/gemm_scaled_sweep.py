#!/usr/bin/env python3
"""Time bsig_gemm_f32 on the shapes of a scaled-batch update (minibatch 8192) for every
tile / split-K choice (BSIG_GEMM_TILE / BSIG_GEMM_SPLITS overrides) and the planner's pick."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayes_sim_ig_amd as B   # noqa: E402

L = B._lib
lib = L.require_gpu()
dev = 'cuda:0'
TILES = {0: '64x64', 1: '128x128', 3: '128x64', 4: '128x96', 5: '96x128'}


def run(m, n, k, akm, bkm, gather, reps=10):
    pool = 80000 if k < 50000 and n < 50000 else 1000
    if gather == 'a':        # A rows gathered from a pool (minibatch rows)
        a = torch.randn(pool, k, device=dev)
        rows_a = torch.randint(0, pool, (m,), device=dev, dtype=torch.int32)
    else:
        a = torch.randn((k, m) if akm else (m, k), device=dev)
        rows_a = None
    if gather == 'b':        # k-major B: contraction rows gathered from the pool
        b = torch.randn(pool, n, device=dev)
        rows_b = torch.randint(0, pool, (k,), device=dev, dtype=torch.int32)
    else:
        b = torch.randn((k, n) if bkm else (n, k), device=dev)
        rows_b = None
    c = torch.empty(m, n, device=dev)
    ws = torch.empty(int(lib.bsig_gemm_workspace_bytes(m, n, k)) // 4 + 64 * 1024 * 1024, device=dev)

    def go():
        L.check(lib.bsig_gemm_f32(L.ptr(a), a.stride(0), akm, L.ptr(rows_a), L.ptr(b), b.stride(0), bkm,
                                  L.ptr(rows_b), L.ptr(c), c.stride(0), m, n, k, 0, 0, None, None, 0,
                                  1.0, L.ptr(ws), ws.numel() * 4, L.stream()))
    for _ in range(3):
        go()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        go()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


shapes = [('fwd head 8192x260x4096 NT gathered A', 8192, 260, 4096, 0, 0, 'a'),
          ('dW head 260x4096x8192 TN gathered B', 260, 4096, 8192, 1, 1, 'b'),
          ('fwd L1 8192x128x11802 NT gathered A', 8192, 128, 11802, 0, 0, 'a'),
          ('dW L1 128x11802x8192 TN gathered B', 128, 11802, 8192, 1, 1, 'b'),
          ('eval head 20000x260x4096 NT', 20000, 260, 4096, 0, 0, None),
          # minibatch-100 products of a first layer too wide for the persistent kernel
          # (cfg/shadow_hand_more.yaml, I = 105002; cfg/anymal.yaml, I = 56402)
          ('wide L1 fwd 100x128x105002 NT gathered A', 100, 128, 105002, 0, 0, 'a'),
          ('wide L1 dW 128x105002x100 TN gathered B', 128, 105002, 100, 1, 1, 'b'),
          ('wide L1 fwd 100x128x56402 NT gathered A', 100, 128, 56402, 0, 0, 'a'),
          # held-out evaluation of a streamed plan: the held-out fifth of a 1000-pair chunk
          ('heldout L1 200x128x56402 NT', 200, 128, 56402, 0, 0, None),
          ('heldout L1 200x128x105002 NT', 200, 128, 105002, 0, 0, None)]
only = sys.argv[1:]
for name, m, n, k, akm, bkm, g in shapes:
    if only and not any(o in name for o in only):
        continue
    fl = 2.0 * m * n * k
    os.environ.pop('BSIG_GEMM_TILE', None); os.environ.pop('BSIG_GEMM_SPLITS', None)
    auto = run(m, n, k, akm, bkm, g)
    print('%-40s %.2f GFLOP  planner: %.1f us = %.1f TFLOP/s' % (name, fl / 1e9, auto, fl / auto / 1e6), flush=True)
    for tile, tname in TILES.items():
        row = []
        for sp in ((1, 2, 3, 4, 6, 8, 12, 16) if k < 50000 else (16, 32, 48, 64, 96, 128, 192, 256)):
            os.environ['BSIG_GEMM_TILE'] = str(tile)
            os.environ['BSIG_GEMM_SPLITS'] = str(sp)
            us = run(m, n, k, akm, bkm, g, 5)
            row.append('s%d=%.0f' % (sp, us))
        print('    %-8s %s' % (tname, ' '.join(row)), flush=True)
os.environ.pop('BSIG_GEMM_TILE', None); os.environ.pop('BSIG_GEMM_SPLITS', None)
