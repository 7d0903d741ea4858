import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import bayes_sim_ig_amd as B
B.MDNN.VERBOSE = False
import test_gpu_persistent_mdnn as T
cfg = T._cfg(4, 6, 'summary_corrdiff', 12, 7, 3)
def run(env, eps, nu=100):
    for k in ('BSIG_MDNN_WIDE_HEADS', 'BSIG_NO_INKERNEL_EVAL'): os.environ.pop(k, None)
    os.environ.update(env)
    r = T._chunk(B, cfg, n=1000, batch=100, n_updates=nu, eps=eps)
    for k in env: os.environ.pop(k, None)
    return r
for eps in (0.0, 1e-5):
  for nu in (100, 81, 82, 90):
    a = run({}, eps, nu); b = run({'BSIG_NO_INKERNEL_EVAL': '1'}, eps, nu); c = run({'BSIG_MDNN_WIDE_HEADS': '1'}, eps, nu)
    print('eps', eps, 'nu', nu)
    print('  in-kernel  ', a[0]['test_loss'])
    print('  graph eval ', b[0]['test_loss'])
    print('  wide       ', c[0]['test_loss'])
    print('  train a/c  ', a[0]['train_loss'][-2:], c[0]['train_loss'][-2:])
    print('  max |w_a - w_b|', float((a[1]-b[1]).abs().max()), ' max |w_a - w_c|', float((a[1]-c[1]).abs().max()))
