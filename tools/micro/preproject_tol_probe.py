import os, sys
import numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
import bayes_sim_ig_amd as B
B.MDNN.VERBOSE = False
DEV = 'cuda:0'
for variant in ['block', 'full_cov', 'no_persistent', 'no_inkernel_eval', 'no_graph']:
    if variant == 'block':
        cfg = dict(task='synthetic', model='MDRFF', summarizer='summary_start', t=11, sd=5, ad=2, d=4, k=6, hidden=[], n_feat=512, pairs=3500)
    else:
        cfg = dict(task='synthetic', model='MDRFF', summarizer='summary_start', t=11, sd=5, ad=2, d=3, k=4, hidden=[], n_feat=256, pairs=3500, full=variant == 'full_cov')
    env = {'no_persistent': {'BSIG_NO_PERSISTENT': '1'}, 'no_inkernel_eval': {'BSIG_NO_INKERNEL_EVAL': '1'}}.get(variant, {})
    B.MDNN.USE_GRAPH = variant != 'no_graph'
    B.MDNN.EPS_NOISE = 0.0 if variant != 'block' else B.MDNN.EPS_NOISE
    theta, states, actions = bench.synth_pairs(cfg, 3500, 3, DEV)
    out = []
    os.environ.update(env)
    for pre in ('0', '1'):
        os.environ['BSIG_NO_FIT_PREPROJECT'] = pre
        bs = bench.build_gpu_model(B, cfg, DEV, 77)
        np.random.seed(11)
        out.append((bs.fit(theta, states, actions), bs.model._flat.clone()))
    for k in list(env) + ['BSIG_NO_FIT_PREPROJECT']:
        os.environ.pop(k, None)
    (la, fa), (lb, fb) = out
    dl = max(np.max(np.abs(np.array(x[k]) - np.array(y[k])) / (np.abs(np.array(y[k])) + 1.0)) for x, y in zip(la, lb) for k in ('train_loss', 'test_loss'))
    dw = (fa - fb).abs().max().item()
    rw = ((fa - fb).abs() / (fb.abs() + 1e-3)).max().item()
    print('%-18s max loss diff / (|loss| + 1) = %.2e   max |dW| = %.2e   max |dW| / (|W| + 1e-3) = %.2e' % (variant, dl, dw, rw), flush=True)
