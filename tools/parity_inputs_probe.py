import sys, numpy as np, torch
sys.path.insert(0, '.')
import bench, bayes_sim_ig_amd as B
from oracle import summarize as osum
B.MDNN.VERBOSE = False; B.MDNN.EPS_NOISE = 0.0
name, seed = sys.argv[1], int(sys.argv[2])
cfg = dict(bench.CONFIGS[name]); torch.set_num_threads(8)
theta, states, actions = bench.synth_pairs(cfg, 1000, seed, 'cuda:0')
ids = np.random.RandomState(5).randint(0, 800, (100, 100))
bs = bench.build_gpu_model(B, cfg, 'cuda:0', 77)
w0 = {k: v.cpu().clone() for k, v in bs.model.state_dict().items()}
summ = bs._summarize(states, actions, lazy=True)
hip = bs.model.run_training(summ, theta, 100, 100, ids_table=ids)
s_gpu = summ.materialize()[:, :bs.model.input_dim].cpu().contiguous()
s_cpu = osum.SUMMARIZERS[cfg['summarizer']](states.cpu(), actions.cpu())
print('summary max abs diff gpu vs cpu', float((s_gpu - s_cpu).abs().max()), 'in columns', torch.nonzero((s_gpu != s_cpu).any(0)).flatten()[:10].tolist())
def run(x, dbl):
    o = bench.build_oracle(cfg, x.shape[1], 77, 0.0)
    sd = w0
    y = theta.cpu()
    if dbl:
        o = o.double(); sd = {k: v.double() for k, v in w0.items()}
        o.output_lows, o.output_highs = o.output_lows.double(), o.output_highs.double(); x = x.double(); y = y.double()
    o.load_state_dict(sd)
    return np.array(o.run_training(x, y, 100, 100, ids_table=ids)['test_loss'], dtype=np.float64)
h = np.array(hip['test_loss'], dtype=np.float64)
f64c = run(s_cpu, True); f64g = run(s_gpu, True)
print('f64(cpu summ)      ', f64c)
print('f64(gpu) - f64(cpu)', f64g - f64c)
print('hip - f64(cpu summ)', h - f64c)
print('hip - f64(gpu summ)', h - f64g)
print('f32(cpu) - f64(cpu)', run(s_cpu, False) - f64c)
print('f32(gpu) - f64(gpu)', run(s_gpu, False) - f64g)
