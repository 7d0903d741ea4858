#!/usr/bin/env python3
"""Time bsig_mdn_head_nll variants (which part of the head kernel costs)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayes_sim_ig_amd as B   # noqa: E402

L = B._lib
lib = L.require_gpu()
dev = 'cuda:0'


def run(b, d, k, full, eps, use_noise, bwd, reps=200):
    hd = L.HeadDims()
    hd.out_dim, hd.n_comp, hd.full_cov = d, k, 1 if full else 0
    hd.eps_noise, hd.min_weight, hd.ll_limit = eps, 1e-5, 1e5
    nh = int(lib.bsig_head_width(C.byref(hd)))
    o = torch.randn(b, nh, device=dev) * 0.5
    y = torch.rand(b, d, device=dev)
    noise = torch.rand(b, d, k, device=dev) if use_noise else None
    loss = torch.zeros(1, device=dev)
    d_o = torch.empty(b, nh, device=dev) if bwd else None
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = torch.empty(int(lib.bsig_head_workspace_bytes(C.byref(hd), b)) // 4 + 64, device=dev)

    def go():
        L.check(lib.bsig_mdn_head_nll(C.byref(hd), L.ptr(o), nh, L.ptr(y), d, None, b, b,
                                      L.ptr(noise), 1, 2, L.ptr(loss), L.ptr(d_o), L.ptr(flag),
                                      L.ptr(ws), ws.numel() * 4, L.stream()))
    for _ in range(10):
        go()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        go()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


for (b, d, k, full) in ((100, 32, 4, False), (200, 32, 4, False), (100, 17, 5, False),
                        (8192, 32, 4, False), (100, 5, 3, True)):
    for eps, nz, bwd in ((0.0, False, False), (0.0, False, True), (1e-5, True, True),
                         (1e-5, False, True), (1e-5, False, False)):
        print('B=%d D=%d K=%d full=%d eps=%g noise_tensor=%d bwd=%d : %.1f us (nll+finish)'
              % (b, d, k, full, eps, nz, bwd, run(b, d, k, full, eps, nz, bwd)), flush=True)
