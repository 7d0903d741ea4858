import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, numpy as np
import bench
import bayes_sim_ig_amd as pkg
pkg._lib.require_gpu()
pkg.MDNN.VERBOSE = False
dev = 'cuda:0'
cfg = dict(bench.CONFIGS['cfg5'])
theta, states, actions = bench.synth_pairs(cfg, 100000, 1234, dev)
bs = bench.build_gpu_model(pkg, cfg, dev, 4321)
summ = bs._summarize(states, actions)
rff = bs.model.rff
print('summ', summ.shape, summ.stride(), float(summ.abs().mean()))
def run(x):
    return rff.to_features(x)
for rows in (20000, 32000, 80000, 100000):
    x = summ[:rows]
    xr = torch.randn(rows, 2312, device=dev)[:, :2310]
    for name, xx in (('real', x), ('randn', xr)):
        for lean in ('0', '1'):
            os.environ['BSIG_GEMM_LEAN'] = lean
            run(xx); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3): run(xx)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            print(rows, name, 'lean' if lean == '1' else 'old ', '%.2f ms  %.1f TF' % (dt * 1e3, 2.0 * rows * 2048 * 2310 / dt / 1e12), flush=True)
