#!/usr/bin/env python3
"""A/B of two builds of the row arithmetic (head_device.h: diag_row) through bsig_mdn_head_nll:
writes loss and d_out of a few shapes to an .npz; run once per library (BSIG_LIB_PATH), then
`python tools/micro/diag_row_ab.py cmp a.npz b.npz` prints where the bits differ."""
import ctypes as C
import os
import sys

import numpy as np

if sys.argv[1] == 'cmp':
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        x, y = a[k], b[k]
        same = np.array_equal(x.view(np.uint32), y.view(np.uint32))
        nd = int((x.view(np.uint32) != y.view(np.uint32)).sum())
        print('%-28s %s  differing %d / %d  max|d| %.3e' % (k, 'BITWISE' if same else 'differs', nd, x.size,
                                                           float(np.abs(x - y).max()) if x.size else 0.0))
        if not same and x.ndim == 2:
            cols = np.where((x.view(np.uint32) != y.view(np.uint32)).any(axis=0))[0]
            print('    columns:', cols[:40])
    sys.exit(0)

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bayes_sim_ig_amd as B

lib = B._lib.require_gpu()
DEV = 'cuda:0'
out = {}
for (b, d, k, eps) in [(100, 32, 4, 1e-5), (100, 32, 4, 0.0), (100, 17, 5, 0.0), (64, 13, 10, 1e-5), (16, 2, 10, 1e-5)]:
    hd = B._lib.HeadDims()
    hd.out_dim, hd.n_comp, hd.full_cov = d, k, 0
    hd.eps_noise, hd.min_weight, hd.ll_limit = eps, 1e-5, 1e5
    nh = int(lib.bsig_head_width(C.byref(hd)))
    gen = torch.Generator().manual_seed(b * 7 + d)
    o = torch.randn(b, nh, generator=gen) * 0.5
    o[:, :k] *= 6.0
    y = torch.rand(b, d, generator=gen)
    noise = torch.rand(b, d, k, generator=gen)
    od, yd, nd = o.to(DEV), y.to(DEV), noise.to(DEV)
    loss = torch.zeros(1, device=DEV)
    d_o = torch.full((b, nh), float('nan'), device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    ws = torch.empty(int(lib.bsig_head_workspace_bytes(C.byref(hd), b)) // 4 + 64, device=DEV)
    B._lib.check(lib.bsig_mdn_head_nll(
        C.byref(hd), B._lib.ptr(od), nh, B._lib.ptr(yd), d, None, b, b, B._lib.ptr(nd), 0, 0,
        B._lib.ptr(loss), B._lib.ptr(d_o), B._lib.ptr(flag), B._lib.ptr(ws), ws.numel() * 4,
        B._lib.stream()))
    torch.cuda.synchronize()
    tag = 'b%d_d%d_k%d_eps%g' % (b, d, k, eps)
    out['loss_' + tag] = loss.cpu().numpy()
    out['dout_' + tag] = d_o.cpu().numpy()
np.savez(sys.argv[1], **out)
print('wrote', sys.argv[1])
