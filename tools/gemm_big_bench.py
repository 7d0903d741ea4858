import os, sys, ctypes as C, torch
sys.path.insert(0, os.getcwd())
import bayes_sim_ig_amd as B
L = B._lib
def bench(libpath):
    lib = C.CDLL(libpath)
    for name in ('bsig_gemm_f32', 'bsig_gemm_workspace_bytes'):
        fn = getattr(lib, name); fn.restype, fn.argtypes = L._PROTOS[name]
    out = []
    for (m, n, k, rows) in ((8192, 2048, 2310, None), (10000, 2048, 2310, 800)):
        a = torch.randn(rows or m, 2312, device='cuda:0')[:, :2310]
        b = torch.randn(n, 2312, device='cuda:0')[:, :2310]
        c = torch.empty(m, 2 * n, device='cuda:0')
        ids = torch.randint(0, rows, (m,), device='cuda:0', dtype=torch.int32) if rows else None
        ws = torch.empty(16, device='cuda:0')
        def go():
            rc = lib.bsig_gemm_f32(L.ptr(a), a.stride(0), 0, L.ptr(ids), L.ptr(b), b.stride(0), 0, None, L.ptr(c), c.stride(0), m, n, k, L.EPI_COS_SIN, 0, None, None, 0, 1.0, L.ptr(ws), 64, L.stream())
            assert rc == 0
        for _ in range(3): go()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): go()
        e.record(); e.synchronize()
        us = s.elapsed_time(e) * 100
        out.append('%dx%dx%d%s %.0f us %.1f TF' % (m, n, k, ' gather' if rows else '', us, 2.0*m*n*k/us/1e6))
    return ' | '.join(out)
for v in ('hip',):
    print('%-12s %s' % (v, bench('bayes_sim_ig_amd/lib/libbsig_%s.so' % v)), flush=True)
