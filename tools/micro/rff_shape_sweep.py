import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bayes_sim_ig_amd as B
L = B._lib; lib = L.require_gpu()
def run(m, n, k):
    a = torch.randn(m, k + (4 - k % 4) % 4, device='cuda:0'); b = torch.randn(n, a.shape[1], device='cuda:0')
    c = torch.empty(m, 2 * n, device='cuda:0'); ws = torch.empty(64 * m * n // 8 + 16, device='cuda:0')
    def go():
        assert lib.bsig_gemm_f32(L.ptr(a), a.stride(0), 0, None, L.ptr(b), b.stride(0), 0, None, L.ptr(c), c.stride(0), m, n, k, L.EPI_COS_SIN, 0, None, None, 0, 1.0, L.ptr(ws), ws.numel() * 4, L.stream()) == 0
    for _ in range(5): go()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(30): go()
    e.record(); e.synchronize()
    return s.elapsed_time(e) * 1e3 / 30
for (m, n, k) in ((800, 2048, 2310), (200, 2048, 2310), (800, 512, 302), (1000, 2048, 2310)):
    row = []
    for tile in (0, 3, 1):
        os.environ['BSIG_GEMM_TILE'] = str(tile)
        for sp in (1, 2):
            os.environ['BSIG_GEMM_SPLITS'] = str(sp)
            us = run(m, n, k)
            row.append('t%d/s%d=%.0fus(%.0fTF)' % (tile, sp, us, 2.0 * m * n * k / us / 1e6))
    os.environ.pop('BSIG_GEMM_TILE'); os.environ.pop('BSIG_GEMM_SPLITS')
    us = run(m, n, k)
    print(m, n, k, ' '.join(row), 'auto=%.0fus(%.0fTF)' % (us, 2.0 * m * n * k / us / 1e6), flush=True)
