#!/usr/bin/env python3
"""A/B of the two fp32 MFMA GEMM kernels through bsig_gemm_f32: gemm_lean_kernel (BSIG_GEMM_LEAN=1,
the default) against gemm_mfma_kernel (BSIG_GEMM_LEAN=0).
  1. bit-identity on ragged problems in every operand layout, tile shape and K split
     (the lean loop keeps every output's fma chain);
  2. times on the shapes of the scaled-batch update and of the RFF projection."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayes_sim_ig_amd as B   # noqa: E402

L = B._lib
lib = L.require_gpu()
dev = 'cuda:0'
TILES = {0: '64x64', 1: '128x128', 2: '128x32', 3: '128x64', 4: '128x96', 5: '96x128'}


_bufs = {}


def gemm(a, b, m, n, k, akm, bkm, rows_a=None, rows_b=None, epi=0, alpha=1.0, ldc=None):
    ldc = ldc or (2 * n if epi == L.EPI_COS_SIN else n)
    key = (m, ldc)
    if key not in _bufs:      # (allocated once per shape: not part of any arm's time)
        _bufs.clear()
        _bufs[key] = (torch.zeros(m, ldc, device=dev),
                      torch.empty(int(lib.bsig_gemm_workspace_bytes(m, n, k)) // 4 + 16 * 1024 * 1024, device=dev))
    c, ws = _bufs[key]
    L.check(lib.bsig_gemm_f32(L.ptr(a), a.stride(0), akm, L.ptr(rows_a), L.ptr(b), b.stride(0), bkm,
                              L.ptr(rows_b), L.ptr(c), c.stride(0), m, n, k, epi, 0, None, None, 0,
                              alpha, L.ptr(ws), ws.numel() * 4, L.stream()))
    return c


def timed(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


def keep(r):      # (gemm() returns its one output buffer per shape)
    return r.clone() if torch.is_tensor(r) else r


def both(fn):
    """0: gemm_mfma_kernel, 1: gemm_lean_kernel (both without the whole-width kernels), 2: as shipped"""
    out = {}
    for lean in (0, 1):
        os.environ['BSIG_GEMM_LEAN'] = str(lean)
        os.environ['BSIG_GEMM_WIDE'] = '0'
        out[lean] = keep(fn())
    os.environ.pop('BSIG_GEMM_LEAN')
    os.environ.pop('BSIG_GEMM_WIDE')
    out[2] = keep(fn())
    return out


def check_wide():
    """The whole-width kernels (gemm_wide.h) at the scaled-batch update's shapes against fp64 on
    sampled outputs, and against the generic kernels everywhere (same products, another summation
    order: 1e-5 of the row scale)."""
    gen = torch.Generator(device=dev).manual_seed(11)
    pool, bsz, nh, f = 20000, 8192, 260, 4096
    feats = torch.randn(pool, f, device=dev, generator=gen) * 0.02
    ids = torch.randint(0, pool, (bsz,), device=dev, generator=gen).to(torch.int32)
    w = torch.randn(nh, f, device=dev, generator=gen)
    bad = 0
    out = gemm(feats, w, bsz, nh, f, 0, 0, rows_a=ids).clone()
    os.environ['BSIG_GEMM_WIDE'] = '0'
    old = gemm(feats, w, bsz, nh, f, 0, 0, rows_a=ids).clone()
    os.environ.pop('BSIG_GEMM_WIDE')
    rows = torch.arange(0, bsz, 97, device=dev)
    ref = feats[ids.long()[rows]].double() @ w.double().T
    e64 = float((out[rows].double() - ref).abs().max())
    eold = float((out - old).abs().max())
    print('wide forward: max |err| vs fp64 %.2e (scale %.2f), vs generic kernel %.2e' % (e64, float(ref.abs().max()), eold))
    bad += e64 > 2e-5 or eold > 2e-5
    d_o = torch.zeros(bsz, 272, device=dev)
    d_o[:, :nh] = torch.randn(bsz, nh, device=dev, generator=gen) * 0.01
    out = gemm(d_o, feats, nh, f, bsz, 1, 1, rows_b=ids).clone()
    os.environ['BSIG_GEMM_WIDE'] = '0'
    old = gemm(d_o, feats, nh, f, bsz, 1, 1, rows_b=ids).clone()
    os.environ.pop('BSIG_GEMM_WIDE')
    cols = torch.arange(0, f, 61, device=dev)
    ref = d_o[:, :nh].double().T @ feats[ids.long()][:, cols].double()
    e64 = float((out[:, cols].double() - ref).abs().max())
    eold = float((out - old).abs().max())
    print('wide gradient: max |err| vs fp64 %.2e (scale %.3f), vs generic kernel %.2e' % (e64, float(ref.abs().max()), eold))
    bad += e64 > 2e-6 or eold > 2e-6
    return bad


def check_bits():
    gen = torch.Generator(device=dev).manual_seed(3)
    bad = 0
    n_cases = 0
    for (m, n, k) in ((300, 260, 530), (129, 97, 64), (100, 512, 2310), (260, 400, 1000), (33, 700, 96)):
        for akm in (0, 1):
            for bkm in (0, 1):
                pool = 900
                a_t = torch.randn(pool if not akm else k, (k if not akm else m) + 0, device=dev, generator=gen)
                b_t = torch.randn(pool, n, device=dev, generator=gen) if bkm else \
                    torch.randn(n, k, device=dev, generator=gen)
                # pad the pitches to multiples of 4 where the shape is not
                def pad4(t):
                    w = (t.shape[1] + 3) // 4 * 4
                    buf = torch.zeros(t.shape[0], w, device=dev)
                    buf[:, :t.shape[1]] = t
                    return buf[:, :t.shape[1]]
                a_t, b_t = pad4(a_t), pad4(b_t)
                rows_a = torch.randint(0, pool, (m,), device=dev, generator=gen).to(torch.int32) if not akm else None
                rows_b = torch.randint(0, pool, (k,), device=dev, generator=gen).to(torch.int32) if bkm else None
                for tile in TILES:
                    for sp in (1, 3):
                        os.environ['BSIG_GEMM_TILE'], os.environ['BSIG_GEMM_SPLITS'] = str(tile), str(sp)
                        r = both(lambda: gemm(a_t, b_t, m, n, k, akm, bkm, rows_a, rows_b))
                        n_cases += 1
                        if not torch.equal(r[0], r[1]):
                            bad += 1
                            print('MISMATCH m=%d n=%d k=%d akm=%d bkm=%d tile=%s splits=%d maxdiff=%g' % (
                                m, n, k, akm, bkm, TILES[tile], sp, float((r[0] - r[1]).abs().max())))
    os.environ.pop('BSIG_GEMM_TILE'), os.environ.pop('BSIG_GEMM_SPLITS')
    # against fp64 on one case per layout (the old kernel is tested against it elsewhere)
    m, n, k = 300, 260, 530
    a64 = torch.randn(m, k, device=dev, generator=gen)
    b64 = torch.randn(n, k, device=dev, generator=gen)
    ref = a64.double() @ b64.double().T
    for akm in (0, 1):
        for bkm in (0, 1):
            def pad(t):
                w = (t.shape[1] + 3) // 4 * 4
                buf = torch.zeros(t.shape[0], w, device=dev)
                buf[:, :t.shape[1]] = t
                return buf[:, :t.shape[1]]
            a = pad((a64.T if akm else a64).contiguous())
            b = pad((b64.T if bkm else b64).contiguous())
            out = gemm(a, b, m, n, k, akm, bkm).clone()
            err = float((out.double() - ref).abs().max())
            print('vs fp64 akm=%d bkm=%d max abs err %.2e' % (akm, bkm, err))
            if err > 3e-4:
                bad += 1
    print('bit-identity: %d cases, %d mismatches' % (n_cases, bad), flush=True)
    return bad


def times():
    pool = 40000
    for a in sys.argv[1:]:
        if a.startswith('--pool='):
            pool = int(a.split('=')[1])
    feats = torch.randn(pool, 4096, device=dev)
    ids = torch.randint(0, pool, (8192,), device=dev, dtype=torch.int32)
    if '--sorted-ids' in sys.argv:
        ids = torch.sort(ids)[0].contiguous()
    if '--iota-ids' in sys.argv:
        ids = torch.arange(8192, device=dev, dtype=torch.int32)
    w = torch.randn(260, 4096, device=dev)
    d_o = torch.randn(8192, 272, device=dev)[:, :260]
    x = torch.randn(32000, 2312, device=dev)[:, :2310]
    co = torch.randn(2048, 2312, device=dev)[:, :2310]
    peak = 157.3
    cases = [
        ('fwd head 8192x260x4096 gathered A', 2.0 * 8192 * 260 * 4096,
         lambda: gemm(feats, w, 8192, 260, 4096, 0, 0, rows_a=ids)),
        ('dW head 260x4096x8192 gathered k', 2.0 * 8192 * 260 * 4096,
         lambda: gemm(d_o, feats, 260, 4096, 8192, 1, 1, rows_b=ids)),
        ('fwd head 8192x256x4096 gathered A', 2.0 * 8192 * 256 * 4096,
         lambda: gemm(feats, w, 8192, 256, 4096, 0, 0, rows_a=ids)),
        ('RFF chunk 800x2048x2310 cos|sin', 2.0 * 800 * 2048 * 2310,
         lambda: gemm(x, co, 800, 2048, 2310, 0, 0, epi=L.EPI_COS_SIN, alpha=0.02)),
        ('RFF chunk+heldout 1000x2048x2310 cos|sin', 2.0 * 1000 * 2048 * 2310,
         lambda: gemm(x, co, 1000, 2048, 2310, 0, 0, epi=L.EPI_COS_SIN, alpha=0.02)),
        ('RFF heldout 200x2048x2310 cos|sin', 2.0 * 200 * 2048 * 2310,
         lambda: gemm(x, co, 200, 2048, 2310, 0, 0, epi=L.EPI_COS_SIN, alpha=0.02)),
        ('RFF 10000x2048x2310 cos|sin', 2.0 * 10000 * 2048 * 2310,
         lambda: gemm(x, co, 10000, 2048, 2310, 0, 0, epi=L.EPI_COS_SIN, alpha=0.02)),
        ('RFF 32000x2048x2310 cos|sin', 2.0 * 32000 * 2048 * 2310,
         lambda: gemm(x, co, 32000, 2048, 2310, 0, 0, epi=L.EPI_COS_SIN, alpha=0.02)),
        ('square 4096^3', 2.0 * 4096 ** 3,
         lambda: gemm(feats, feats[4096:], 4096, 4096, 4096, 0, 0)),
    ]
    only = [a for a in sys.argv[1:] if not a.startswith('-')]
    print('pool', pool, [a for a in sys.argv[1:] if a.startswith('--')])
    for name, fl, fn in cases:
        if only and not any(o in name for o in only):
            continue
        # (allocation of c / ws inside fn is part of neither arm's kernel time but of both walls:
        # pre-allocate by running once, torch's caching allocator then reuses the blocks)
        r = both(lambda: timed(fn))
        print('%-40s old %8.1f us = %6.1f TF (%.3f)   lean %8.1f us = %6.1f TF (%.3f)   shipped %8.1f us = %6.1f TF (%.3f)' % (
            name, r[0], fl / r[0] / 1e6, fl / r[0] / 1e6 / peak, r[1], fl / r[1] / 1e6, fl / r[1] / 1e6 / peak,
            r[2], fl / r[2] / 1e6, fl / r[2] / 1e6 / peak), flush=True)
        if '--sweep' in sys.argv:
            for tile, tname in TILES.items():
                row = []
                for sp in (1, 2, 3, 4, 6, 8):
                    os.environ['BSIG_GEMM_TILE'], os.environ['BSIG_GEMM_SPLITS'] = str(tile), str(sp)
                    try:
                        us = timed(fn, 5, 2)
                        row.append('s%d=%.0f' % (sp, us))
                    except Exception as ex:   # noqa: BLE001
                        row.append('s%d=ERR' % sp)
                print('    lean %-8s %s' % (tname, ' '.join(row)), flush=True)
            os.environ.pop('BSIG_GEMM_TILE'), os.environ.pop('BSIG_GEMM_SPLITS')


if __name__ == '__main__':
    bad = 0
    if '--no-bits' not in sys.argv:
        bad = check_bits()
    if '--no-check' not in sys.argv:
        bad += check_wide()
    if '--no-times' not in sys.argv:
        times()
    sys.exit(1 if bad else 0)
