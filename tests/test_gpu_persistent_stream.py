"""The persistent update kernel with a STREAMED first layer
(csrc/fit_persistent_mdnn_stream.hip; SURVEY.md 8(f2) at the widths of cfg/anymal.yaml and
cfg/shadow_hand_more.yaml, summarizers.py:112-119 into mdnn.py:71,108): parity with the oracle
(teacher-forced chunks from cross-correlation FACTOR rows, every logged loss), with the
per-phase kernels on materialised summaries, bitwise reruns, ragged minibatches, wide heads,
a data-parallel rank."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def B():
    import bayes_sim_ig_amd as pkg
    pkg._lib.require_gpu()
    pkg.MDNN.VERBOSE = False
    return pkg


@pytest.fixture(autouse=True)
def _guards():
    import bayes_sim_ig_amd as pkg
    old = pkg.MDNN.EPS_NOISE
    yield
    pkg.MDNN.EPS_NOISE = old
    pkg.MDNN.USE_GRAPH = True
    for k in ('BSIG_NO_PERSISTENT', 'BSIG_NO_STREAMED_W1', 'BSIG_NO_STREAM_EVAL', 'BSIG_NO_WIDE_EVAL'):
        os.environ.pop(k, None)


def _cfg(d, k, t, sd, ad, summarizer='summary_corrdiff'):
    return dict(task='synthetic', model='MDNN', summarizer=summarizer, t=t, sd=sd, ad=ad,
                d=d, k=k, hidden=[128, 128], n_feat=0, pairs=1000)


# T = 8 < 10 waypoints -> W = 8: S = 8 (sd - 1) = 160, A = 8 ad = 96, I = 15362: 61 k-slices x 4
# tiles + 25 owners + 13 small-weight workgroups > 256 CUs -> the first layer is streamed
SMALL = dict(d=3, k=5, t=8, sd=21, ad=12)


def _chunk(B, cfg, n=1000, batch=100, n_updates=100, seed=3, eps=0.0, env=None, lazy=True):
    import bench
    for k in ('BSIG_NO_PERSISTENT', 'BSIG_NO_STREAMED_W1', 'BSIG_NO_STREAM_EVAL', 'BSIG_NO_WIDE_EVAL'):
        os.environ.pop(k, None)
    os.environ.update(env or {})
    B.MDNN.EPS_NOISE = eps
    theta, states, actions = bench.synth_pairs(cfg, n, seed, DEV)
    bs = bench.build_gpu_model(B, cfg, DEV, 77)
    n_train = max(n - int(n * 0.2), 1)
    ids = np.random.RandomState(5).randint(0, n_train, (n_updates, batch))
    summ = bs._summarize(states, actions, lazy=lazy)
    logs = bs.model.run_training(summ, theta, n_updates, batch, ids_table=ids)
    return logs, bs.model._flat.clone(), bs, (theta, states, actions, ids)


def _streams(B, bs):
    lib = B._lib.load()
    got = (lib.bsig_fit_is_persistent(bs.model._plan), lib.bsig_fit_accepts_factors(bs.model._plan))
    assert got == (2, 0), got
    return True


def _oracles(bench, cfg, in_dim, w0, summ_cpu, theta, n_updates, ids, double):
    o = bench.build_oracle(cfg, in_dim, 77, 0.0)
    x, y = summ_cpu, theta.cpu()
    if double:
        o = o.double()
        o.output_lows, o.output_highs = o.output_lows.double(), o.output_highs.double()
        x, y = x.double(), y.double()
    o.load_state_dict({k: (v.double() if double else v) for k, v in w0.items()})
    return o.run_training(x, y, n_updates, 100, ids_table=ids), o


def test_streamed_chunk_matches_oracle(B):
    """Teacher-forced chunks (EPS_NOISE = 0, same start weights, same ids) from factor rows
    against the fp32 oracle on the materialised summaries.  40 updates: every logged loss within
    the north-star 1e-4, the weights within Adam-step noise.  100 updates: the first layer sums
    15362 fp32 products per unit and two fp32 evaluation orders of this chunk part by ~1e-4 at
    the end (like cfg3, tests/test_gpu_fit.py) -- there the yardstick is the chunk in fp64:
    |hip - f64| <= |cpu_f32 - f64| + 1e-4 |f64| at every logging point."""
    import bench
    from oracle import summarize as osum
    cfg = _cfg(**SMALL)
    torch.set_num_threads(8)
    bs0 = bench.build_gpu_model(B, cfg, DEV, 77)
    w0 = {kk: v.cpu().clone() for kk, v in bs0.model.state_dict().items()}
    for n_updates in (40, 100):
        logs, flat, bs, (theta, states, actions, ids) = _chunk(B, cfg, n_updates=n_updates)
        assert _streams(B, bs)
        s_cpu = osum.SUMMARIZERS[cfg['summarizer']](states.cpu(), actions.cpu())
        ref, ora = _oracles(bench, cfg, bs.model.input_dim, w0, s_cpu, theta, n_updates, ids, False)
        if n_updates == 40:
            for key in ('train_loss', 'test_loss'):
                got, exp = np.array(logs[key]), np.array(ref[key])
                assert got.shape == exp.shape == (6,)
                assert np.all(np.abs(got - exp) <= 1e-4 * np.maximum(np.abs(exp), 1.0)), (key, got, exp)
            sd_ = bs.model.state_dict()
            for name, v in ora.state_dict().items():
                diff = (sd_[name].cpu() - v).abs()
                assert float(diff.mean()) < 2e-6 and float(diff.max()) < 2.5e-3, (name, diff.mean(), diff.max())
        else:
            r64, _ = _oracles(bench, cfg, bs.model.input_dim, w0, s_cpu, theta, n_updates, ids, True)
            for key in ('train_loss', 'test_loss'):
                h, a, r = (np.asarray(v[key], dtype=np.float64) for v in (logs, ref, r64))
                assert (np.abs(h - r) <= np.abs(a - r) + 1e-4 * np.abs(r) + 1e-6).all(), (key, h - r, a - r)


@pytest.mark.parametrize('eps', [0.0, 1e-5])
def test_streamed_equals_phase_kernels(B, eps):
    """The same chunk through the streamed kernel (factor rows) and through the per-phase
    kernels (materialised rows): same jitter streams, summation order differs (40 updates:
    before the two fp32 orders part)."""
    cfg = _cfg(**SMALL)
    logs_s, flat_s, bs_s, _ = _chunk(B, cfg, eps=eps, n_updates=40)
    logs_k, flat_k, bs_k, _ = _chunk(B, cfg, eps=eps, n_updates=40, env={'BSIG_NO_STREAMED_W1': '1'})
    assert _streams(B, bs_s)
    assert B._lib.load().bsig_fit_is_persistent(bs_k.model._plan) == 0
    for key in ('train_loss', 'test_loss'):
        assert np.allclose(logs_s[key], logs_k[key], rtol=2e-5, atol=2e-5), (key, logs_s, logs_k)
    assert torch.allclose(flat_s, flat_k, atol=2.5e-3, rtol=1e-2)
    assert float((flat_s - flat_k).abs().mean()) < 2e-6


def test_streamed_reruns_are_bitwise(B):
    cfg = _cfg(**SMALL)
    a = _chunk(B, cfg, eps=1e-5, n_updates=30)
    b = _chunk(B, cfg, eps=1e-5, n_updates=30)
    assert a[0] == b[0]
    assert torch.equal(a[1], b[1])


@pytest.mark.parametrize('n,batch,n_updates', [(1000, 64, 13), (1000, 33, 6), (60, 100, 5), (1000, 104, 7)])
def test_streamed_ragged_shapes_match_phase_kernels(B, n, batch, n_updates):
    cfg = _cfg(**SMALL)
    logs_s, flat_s, bs_s, _ = _chunk(B, cfg, n=n, batch=batch, n_updates=n_updates)
    logs_k, flat_k, _, _ = _chunk(B, cfg, n=n, batch=batch, n_updates=n_updates,
                                  env={'BSIG_NO_STREAMED_W1': '1'})
    assert _streams(B, bs_s)
    for key in ('train_loss', 'test_loss'):
        assert np.allclose(logs_s[key], logs_k[key], rtol=1e-4, atol=1e-4), (key, logs_s, logs_k)
    assert float((flat_s - flat_k).abs().mean()) < 2e-6


def test_streamed_wide_heads_match_phase_kernels(B):
    """10 components x D = 17 (Nh = 350): the head-block workgroups form the head outputs."""
    cfg = _cfg(d=17, k=10, t=8, sd=21, ad=12)
    logs_s, flat_s, bs_s, _ = _chunk(B, cfg, n_updates=20)
    logs_k, flat_k, _, _ = _chunk(B, cfg, n_updates=20, env={'BSIG_NO_STREAMED_W1': '1'})
    assert _streams(B, bs_s)
    for key in ('train_loss', 'test_loss'):
        assert np.allclose(logs_s[key], logs_k[key], rtol=1e-4, atol=1e-4), (key, logs_s, logs_k)
    assert float((flat_s - flat_k).abs().mean()) < 2e-6


def test_streamed_plan_runs_summary_rows_through_phase_kernels(B):
    """Materialised rows of the same width: the plan binds them to the per-phase kernels."""
    cfg = _cfg(**SMALL)
    logs_s, _, _, _ = _chunk(B, cfg, n_updates=10)
    logs_r, _, bs_r, _ = _chunk(B, cfg, n_updates=10, lazy=False)
    assert B._lib.load().bsig_fit_is_persistent(bs_r.model._plan) == 0
    for key in ('train_loss', 'test_loss'):
        assert np.allclose(logs_s[key], logs_r[key], rtol=1e-4, atol=1e-4)


def test_streamed_fit_never_materialises_the_summaries(B):
    """BayesSim.fit on a streamed plan: logs of every chunk, no [N, I] tensor beyond the
    held-out fifth of a chunk (peak memory stays far below the materialised block)."""
    import bench
    cfg = _cfg(**SMALL)
    theta, states, actions = bench.synth_pairs(cfg, 3000, 3, DEV)
    bs = bench.build_gpu_model(B, cfg, DEV, 77)
    np.random.seed(3)
    bs.fit(theta, states, actions)                      # plan, workspaces
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    logs = bs.fit(theta, states, actions)
    torch.cuda.synchronize()
    assert len(logs) == 3 and all(np.isfinite(lg['test_loss']).all() for lg in logs)
    assert _streams(B, bs)
    block = 3000 * bs.model.input_dim * 4
    assert torch.cuda.max_memory_allocated() - base < block / 2


def test_streamed_data_parallel_rank_matches_resident(B):
    """A 1-rank data-parallel group (gradients out, flat Adam after the exchange) against the
    single-rank run: same minibatches, arithmetic differs in the Adam kernel's rounding only."""
    import torch.distributed as dist
    cfg = _cfg(**SMALL)
    logs_s, flat_s, _, _ = _chunk(B, cfg, n_updates=20)
    created = False
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29577')
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device(DEV))
        created = True
    try:
        import bench
        B.MDNN.EPS_NOISE = 0.0
        theta, states, actions = bench.synth_pairs(cfg, 1000, 3, DEV)
        bs = bench.build_gpu_model(B, cfg, DEV, 77)
        bs.model.enable_data_parallel()
        ids = np.random.RandomState(5).randint(0, 800, (20, 100))
        summ = bs._summarize(states, actions, lazy=True)
        logs_d = bs.model.run_training(summ, theta, 20, 100, ids_table=ids)
        assert _streams(B, bs)
        for key in ('train_loss', 'test_loss'):
            assert np.allclose(logs_s[key], logs_d[key], rtol=1e-4, atol=1e-4), (key, logs_s, logs_d)
        assert float((flat_s - bs.model._flat).abs().mean()) < 2e-6
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize('n', [1000, 650])
def test_streamed_evaluations_inside_the_launch_match_the_evaluation_graphs(B, n):
    """mdnn.py:235-242: the six held-out evaluations of a chunk.  A streamed plan (narrow heads) takes
    them inside its ONE launch -- the tile workgroups form the held-out pairs' first-layer products from
    their factor rows in the window in which they wait for the row owners, the owners finish them -- ;
    BSIG_NO_STREAM_EVAL=1 runs them as graphs on materialised held-out rows between six launches.  Same
    weights at every evaluation point (the updates are bitwise the same), the evaluation arithmetic
    differs (summation order of the first layer).  n = 650: a ragged second evaluation pass (130
    held-out pairs)."""
    cfg = _cfg(**SMALL)
    logs_i, flat_i, bs_i, _ = _chunk(B, cfg, n=n, eps=1e-5, n_updates=40)
    logs_g, flat_g, bs_g, _ = _chunk(B, cfg, n=n, eps=1e-5, n_updates=40, env={'BSIG_NO_STREAM_EVAL': '1'})
    assert _streams(B, bs_i) and _streams(B, bs_g)
    assert logs_i['train_loss'] == logs_g['train_loss']
    assert torch.equal(flat_i, flat_g)
    assert len(logs_i['test_loss']) == 6
    assert np.allclose(logs_i['test_loss'], logs_g['test_loss'], rtol=2e-5, atol=2e-5), (logs_i, logs_g)


@pytest.mark.parametrize('streamed', [True, False])
def test_wide_heads_evaluate_inside_the_launch(B, streamed):
    """Wide heads (10 components x D = 17: the head-block workgroups form the head outputs): the
    evaluations inside the launch -- owners publish the pass's h2 rows, the head blocks return the head
    outputs with the evaluated update's weights (their operand copies, before the refresh) -- against
    BSIG_NO_WIDE_EVAL=1 (evaluation graphs between the launches).  streamed: first layer streamed
    (I = 15362); else resident (I = 1026)."""
    cfg = _cfg(d=17, k=10, t=8, sd=21, ad=12) if streamed else _cfg(d=17, k=10, t=8, sd=5, ad=4)
    logs_i, flat_i, bs_i, _ = _chunk(B, cfg, eps=1e-5, n_updates=40)
    logs_g, flat_g, bs_g, _ = _chunk(B, cfg, eps=1e-5, n_updates=40, env={'BSIG_NO_WIDE_EVAL': '1'})
    lib = B._lib.load()
    assert lib.bsig_fit_is_persistent(bs_i.model._plan) == 2 and lib.bsig_fit_is_persistent(bs_g.model._plan) == 2
    if streamed:
        assert _streams(B, bs_i)
    assert logs_i['train_loss'] == logs_g['train_loss']
    assert torch.equal(flat_i, flat_g)
    assert len(logs_i['test_loss']) == 6
    assert np.allclose(logs_i['test_loss'], logs_g['test_loss'], rtol=2e-5, atol=2e-5), (logs_i, logs_g)
