import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo')); sys.path.insert(0, ''+os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/tests')
import bayes_sim_ig_amd as B, bench
from oracle import summarize as osum
import test_gpu_persistent_mdnn as T
B.MDNN.VERBOSE = False
cfg = T._cfg(32, 4, 'summary_start', 11, 97, 20)
torch.set_num_threads(8)
logs, flat, bs, (theta, states, actions, ids) = T._chunk(B, cfg, eps=0.0)
logs_k, flat_k, bs_k, _ = T._chunk(B, cfg, eps=0.0, env={'BSIG_NO_PERSISTENT': '1'})
os.environ.pop('BSIG_NO_PERSISTENT')
ora = bench.build_oracle(cfg, bs.model.input_dim, 77, 0.0)
bs0 = bench.build_gpu_model(B, cfg, 'cuda:0', 77)
ora.load_state_dict({kk: v.cpu() for kk, v in bs0.model.state_dict().items()})
ref = ora.run_training(osum.SUMMARIZERS['summary_start'](states.cpu(), actions.cpu()), theta.cpu(), 100, 100, ids_table=ids)
sd_p, sd_k = bs.model.state_dict(), bs_k.model.state_dict()
for name, v in ora.state_dict().items():
    dp = (sd_p[name].cpu() - v).abs(); dk = (sd_k[name].cpu() - v).abs()
    print(name, 'persist max %.2e n>2e-4 %d | phase max %.2e n>2e-4 %d | of %d' % (dp.max(), (dp > 2e-4).sum(), dk.max(), (dk > 2e-4).sum(), v.numel()))
print(logs['test_loss'], ref['test_loss'], logs_k['test_loss'])
