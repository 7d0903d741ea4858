import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd())
import bench, bayes_sim_ig_amd as pkg
pkg.MDNN.VERBOSE=False
cfg=dict(bench.CONFIGS['cfg5'])
dev='cuda:0'
theta,states,actions=bench.synth_pairs(cfg,100000,1234,dev)
for env in ({}, {'BSIG_GEMM_NO_LARGE_PLAN':'1'}):
    os.environ.update(env)
    bs=bench.build_gpu_model(pkg,cfg,dev,4321)
    np.random.seed(4321)
    summ=bs._summarize(states,actions)
    logs=bs.model.run_training(summ,theta,122,8192,test_frac=0.2)
    print(env, logs['train_loss'], logs['test_loss'])
