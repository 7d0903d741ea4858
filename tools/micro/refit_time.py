"""Wall time of BayesSim.predict's multi-trajectory refit (500 updates of a full-covariance MDNN)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import bayes_sim_ig_amd as B
B.MDNN.VERBOSE = False
dev = 'cuda:0'
theta, states, actions = B.pairs.pendulum_pairs(2000, 20, policy='random', seed=0, device=dev)
cfg = {'modelClass': 'MDNN', 'summarizerFxn': 'summary_start', 'trainTrajLen': 20, 'components': 10,
       'hiddenLayers': (128, 128), 'lr': 1e-4, 'fullCovariance': True}
for env in ({}, {'BSIG_NO_PERSISTENT': '1'}):
    os.environ.pop('BSIG_NO_PERSISTENT', None); os.environ.update(env)
    torch.manual_seed(0); np.random.seed(0)
    bs = B.BayesSim(cfg, obs_dim=3, act_dim=1, params_dim=2, params_lows=np.array([0.01, 0.01]),
                    params_highs=np.array([2.0, 2.0]), prior=None, device=dev)
    bs.fit(theta, states, actions)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mog = bs.predict(states[:5], actions[:5])
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(env, 'predict (refit of 5 trajectories): %.1f ms' % (dt * 1e3))
