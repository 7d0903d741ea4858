"""The RFF projection at the per-chunk shape (run_training called per chunk, the reference's own
call pattern: bayes_sim_main.py:157-167): 800 x 2048 x 2310 with fused cos/sin, every tile shape of
the MFMA GEMM x split-K, against the planner's pick (rff.py:128-132)."""
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bayes_sim_ig_amd as B
L = B._lib; lib = L.require_gpu()
NAMES = {0: '64x64', 1: '128x128', 2: '128x32', 3: '128x64', 4: '128x96', 5: '96x128'}
def run(m, n, k):
    a = torch.randn(m, k + (4 - k % 4) % 4, device='cuda:0'); b = torch.randn(n, a.shape[1], device='cuda:0')
    c = torch.empty(m, 2 * n, device='cuda:0'); ws = torch.empty(8 * m * n + 16, device='cuda:0')
    def go():
        assert lib.bsig_gemm_f32(L.ptr(a), a.stride(0), 0, None, L.ptr(b), b.stride(0), 0, None, L.ptr(c), c.stride(0), m, n, k, L.EPI_COS_SIN, 0, None, None, 0, 1.0, L.ptr(ws), ws.numel() * 4, L.stream()) == 0
    for _ in range(5): go()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(30): go()
    e.record(); e.synchronize()
    return s.elapsed_time(e) * 1e3 / 30
for (m, n, k) in ((800, 2048, 2310), (1000, 2048, 2310), (200, 2048, 2310)):
    best = None
    for tile in (0, 1, 3, 4, 5):
        row = []
        for sp in (1, 2, 3, 4):
            os.environ['BSIG_GEMM_TILE'] = str(tile); os.environ['BSIG_GEMM_SPLITS'] = str(sp)
            us = run(m, n, k)
            tf = 2.0 * m * n * k / us / 1e6
            row.append('s%d %.0f us %.0f TF (%.2f)' % (sp, us, tf, tf / 157.3))
            if best is None or us < best[0]: best = (us, tile, sp)
        print('%d x %d x %d tile %-7s %s' % (m, n, k, NAMES[tile], ' | '.join(row)), flush=True)
    os.environ.pop('BSIG_GEMM_TILE'); os.environ.pop('BSIG_GEMM_SPLITS')
    us = run(m, n, k)
    print('%d x %d x %d planner: %.0f us %.0f TF (%.2f of the fp32 MFMA peak); best of the sweep: tile %s split %d %.0f us'
          % (m, n, k, us, 2.0 * m * n * k / us / 1e6, 2.0 * m * n * k / us / 1e6 / 157.3, NAMES[best[1]], best[2], best[0]), flush=True)
