import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import bayes_sim_ig_amd as B
B.MDNN.VERBOSE = False
import test_gpu_persistent_mdnn as T
torch.set_num_threads(8)
for (d, k, summ, sd, ad, t) in [(2, 10, 'summary_start', 3, 1, 21), (6, 5, 'summary_corrdiff', 7, 3, 12), (8, 10, 'summary_start', 5, 2, 11), (2, 3, 'summary_start', 3, 1, 21)]:
    cfg = dict(T._cfg(d, k, summ, t, sd, ad), full=True)
    a = T._chunk(B, cfg, eps=0.0)
    b = T._chunk(B, cfg, eps=0.0, env={'BSIG_NO_PERSISTENT': '1'})
    os.environ.pop('BSIG_NO_PERSISTENT', None)
    ref, ora = T._oracle_chunk(B, cfg, a[2], *a[3])
    print('D', d, 'K', k, 'persistent', B._lib.load().bsig_fit_is_persistent(a[2].model._plan))
    print('   persistent train', a[0]['train_loss'])
    print('   per-phase  train', b[0]['train_loss'])
    print('   oracle     train', ref['train_loss'])
    print('   persistent test ', a[0]['test_loss'])
    print('   per-phase  test ', b[0]['test_loss'])
    print('   oracle     test ', ref['test_loss'])
