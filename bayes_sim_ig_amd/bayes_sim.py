"""The core BayesSim class on MI355X — counterpart of the reference's
bayes_sim_ig/bayes_sim.py (class BayesSim: same constructor, constants,
``run_training`` / ``predict`` / ``get_n_trajs_per_batch``).

The reference's file cannot travel with this repo; this orchestrator keeps
its behaviour (summarizer and model picked by name from ``model_cfg``, the
hard-coded chunk schedule of bayes_sim.py:20-25, the multi-trajectory refit
of :148-179) and drives the HIP summarizers and estimators.  ``fit`` adds the
caller-side chunk loop of bayes_sim_main.py:157-167 for pre-recorded pairs.
"""
import os

import numpy as np
import torch

from . import pdf
from . import summarizers as _summ
from .mdnn import MDNN
from .mdrff import MDRFF
from .summarizers import (pad_states_actions, summary_start, summary_waypts,  # noqa: F401
                          summary_corr, summary_corrdiff, summary_signatory,
                          signature_depth, cross_correlation)

_SUMMARIZERS = {
    'summary_start': summary_start, 'summary_waypts': summary_waypts,
    'summary_corr': summary_corr, 'summary_corrdiff': summary_corrdiff,
    'summary_signatory': summary_signatory,
}
_MODELS = {'MDNN': MDNN, 'MDRFF': MDRFF}


class BayesSim(object):
    # The reference's protocol constants (bayes_sim.py:20-25), same names and values -- the chunk size a
    # run_training call consumes, how often a chunk is swept, the SGD minibatch, the held-out share:
    NUM_TRAIN_TRAJ_PER_BATCH = 1000  # (theta, trajectory) pairs per run_training call ("chunk")
    NUM_TRAIN_EPOCHS = 10            # sweeps over a chunk
    MINIBATCH_SIZE = 100             # rows per Adam update
    NUM_GRAD_UPDATES = NUM_TRAIN_EPOCHS * NUM_TRAIN_TRAJ_PER_BATCH // MINIBATCH_SIZE   # = 100 updates per chunk
    TEST_FRACTION = 0.2              # last 20 % of a chunk: held out (mdnn.py:206-211)
    FIT_BLOCK_CHUNKS = 32            # (not in the reference) chunks summarised / projected at once by fit()
    # the multi-trajectory refit of predict(): literals in the reference (bayes_sim.py:162,173-174)
    REFIT_SAMPLES = int(1e4)         # samples drawn from the per-trajectory MoGs
    REFIT_MINIBATCH = 100
    REFIT_EPOCHS = 5

    def __init__(self, model_cfg, obs_dim, act_dim, params_dim, params_lows,
                 params_highs, prior, proposal=None, device='cpu'):
        """Arguments as in the reference (bayes_sim.py:27-52).  Optional
        ``model_cfg`` keys beyond the reference's: ``nFeat`` (RFF features,
        default 200 as hard-coded at bayes_sim.py:81), ``sigDepth``."""
        self.prior = prior
        self.proposal = proposal
        model_class = model_cfg['modelClass']
        name = model_cfg['summarizerFxn']
        if name not in _SUMMARIZERS:
            raise NameError("name '%s' is not defined" % name)   # eval() in the reference
        self.summarizer_name = name
        self.summarizer_fxn = _SUMMARIZERS[name]
        self._sig_depth = model_cfg.get('sigDepth', None)
        # the reference pushes a zero trajectory through the summarizer to get
        # the width (bayes_sim.py:57-60); the width is a closed form
        traj_summaries_dim = _summ.summary_dim(
            name, model_cfg['trainTrajLen'], obs_dim, act_dim,
            self._sig_depth or 0)
        full_covariance = bool(model_cfg.get('fullCovariance', False))
        kwargs = {'input_dim': traj_summaries_dim, 'output_dim': params_dim,
                  'output_lows': params_lows, 'output_highs': params_highs,
                  'n_gaussians': model_cfg['components'],
                  'hidden_layers': model_cfg['hiddenLayers'],
                  'lr': model_cfg['lr'],
                  'activation': torch.nn.Tanh,
                  'full_covariance': full_covariance,
                  'device': device}
        if model_class.startswith('MDRFF'):      # "MDRFF_<kernel>_<sigma>"
            kernel, sigma = 'RBF', 4.0
            if '_' in model_class:
                parts = model_class.split('_')
                model_class, kernel = parts[0], parts[1]
                if len(parts) > 2:
                    sigma = float(parts[2])
            kwargs.update({'n_feat': int(model_cfg.get('nFeat', 200)),
                           'sigma': sigma, 'kernel': kernel})
        if model_class not in _MODELS:
            raise NameError("name '%s' is not defined" % model_class)
        self.model = _MODELS[model_class](**kwargs)

    @staticmethod
    def get_n_trajs_per_batch(n_train_trajs, n_train_trajs_done):
        n = BayesSim.NUM_TRAIN_TRAJ_PER_BATCH
        if n_train_trajs_done + n > n_train_trajs:
            n = n_train_trajs - n_train_trajs_done
        return n

    def _summarize(self, states, actions, finite_flag=None, lazy=False):
        if self.summarizer_name == 'summary_signatory' and self._sig_depth:
            return self.summarizer_fxn(states, actions, depth=self._sig_depth)
        if self.summarizer_name in ('summary_corr', 'summary_corrdiff'):
            kw = {}
            if finite_flag is not None:
                # the isfinite assert of summarizers.py:120, deferred with the chunk's logs
                kw['check_finite'] = finite_flag
            # training never needs the outer product itself: the estimator's first layer forms
            # it from the factor rows (summarizers.CrossCorrFactors)
            if lazy and self._lazy_summaries():
                kw['lazy'] = True
            return self.summarizer_fxn(states, actions, **kw)
        return self.summarizer_fxn(states, actions)

    def _lazy_summaries(self):
        return (getattr(self.model, 'rff', None) is None and self.model._flat.is_cuda and
                os.environ.get('BSIG_NO_FUSED_SUMMARY') != '1')

    def run_training(self, params, traj_states, traj_actions, _defer=False, _finite_flag=None,
                     _summaries=None, _feats=None):
        """One chunk: summarize, then NUM_GRAD_UPDATES Adam updates of
        MINIBATCH_SIZE (reference bayes_sim.py:91-114)."""
        traj_summaries = _summaries if _summaries is not None else \
            self._summarize(traj_states, traj_actions, _finite_flag, lazy=True)
        kw = {} if _feats is None else {'_feats': _feats}
        return self.model.run_training(
            x_data=traj_summaries, y_data=params,
            n_updates=BayesSim.NUM_GRAD_UPDATES,
            batch_size=BayesSim.MINIBATCH_SIZE,
            test_frac=BayesSim.TEST_FRACTION, _defer=_defer, **kw)

    FIT_BLOCK_BYTES = 4 << 30        # (not in the reference) bound on a block's summaries in fit()

    def fit(self, params, traj_states, traj_actions):
        """The caller-side loop of bayes_sim_main.py:157-167 over
        pre-recorded pairs: consecutive chunks of at most
        NUM_TRAIN_TRAJ_PER_BATCH pairs, ``run_training`` on each.
        Returns the list of per-chunk log dicts."""
        if self.model._flat.is_cuda and self.model._may_time_out():
            # the chunks' logs are read after the last chunk is enqueued: if a persistent launch
            # turns out not to have had the GPU to itself (its bounded polls gave up), the whole
            # loop is repeated from here on the per-phase kernels.  A data-parallel group does the
            # same TOGETHER: the time-out bit travels in the logs every call sums over the ranks
            # (bsig_fit_run_dp), the rank that timed out keeps enqueueing its all-reduces, so every
            # rank reads the same flag at the same chunk, restores and repeats
            # A rank that stayed resident across the exchange first goes back to one launch per update
            # (MDNN._give_up_a_level), the per-phase kernels come second.
            from .mdnn import PersistentTimeout
            snap = self.model._snapshot()
            for _ in range(2):
                calls0 = self.model._resident_calls()
                try:
                    return self._fit_once(params, traj_states, traj_actions)
                except PersistentTimeout:
                    self.model._restore(snap)
                    self.model._give_up_a_level(calls0)
        return self._fit_once(params, traj_states, traj_actions)

    def _fit_once(self, params, traj_states, traj_actions):
        n, done, pending = params.shape[0], 0, []
        dp = getattr(self.model, '_dp', None)
        if dp is not None:
            # one all-reduce per update: every rank must run the same chunk schedule, or the
            # rank with an extra chunk waits for its peers forever
            counts = dp.gather_counts(n, device=self.model._flat.device)
            if len(set(counts)) != 1:
                raise ValueError('data-parallel fit needs the same number of pairs on every rank '
                                 '(got %s); see dp.equal_shards' % (counts,))
        flag = None
        if torch.is_tensor(traj_states) and traj_states.is_cuda:
            flag = torch.zeros(1, dtype=torch.int32, device=traj_states.device)
        # The summaries (and an MDRFF's RFF features, rff.py:128-132) are pure functions of the
        # row: both are computed for a block of chunks at once -- one summarizer launch over
        # up to 32000 trajectories (and one large MFMA GEMM) instead of one small one per chunk
        block = 0
        is_rff = getattr(self.model, 'rff', None) is not None
        if flag is not None and os.environ.get('BSIG_NO_FIT_PREPROJECT') != '1':
            row_bytes = 4 * (self.model.input_dim + (self.model.rff.n_feat if is_rff else 0))
            chunks = max(min(BayesSim.FIT_BLOCK_CHUNKS,
                             BayesSim.FIT_BLOCK_BYTES // (row_bytes * BayesSim.NUM_TRAIN_TRAJ_PER_BATCH)), 1)
            block = chunks * BayesSim.NUM_TRAIN_TRAJ_PER_BATCH
        lo = hi = 0
        summ = feats = None
        while done < n:
            m = BayesSim.get_n_trajs_per_batch(n, done)
            if block and done + m > hi:
                lo, hi = done, min(n, done + block)
                summ = self._summarize(traj_states[lo:hi], traj_actions[lo:hi], flag, lazy=True)
                feats = self.model.rff.to_features(summ) if is_rff else None
            if block:
                pending.append(self.run_training(
                    params[done:done + m], None, None, _defer=True,
                    _summaries=summ[done - lo:done - lo + m],
                    _feats=None if feats is None else feats[done - lo:done - lo + m]))
            else:
                pending.append(self.run_training(params[done:done + m],
                                                 traj_states[done:done + m],
                                                 traj_actions[done:done + m], _defer=True,
                                                 _finite_flag=flag))
            done += m
        # one host synchronisation for the whole fit: the chunks' logs (and the
        # isfinite asserts) are read back after the last chunk is enqueued
        logs = [p.result() for p in pending]
        assert flag is None or int(flag.item()) == 0   # summarizers.py:120
        return logs

    def predict(self, states, actions, threshold=0.005):
        """Posterior for the given real trajectories (the contract of reference bayes_sim.py:116-179):
        ONE trajectory -> the model's mixture for it (divided by the proposal when there is one);
        SEVERAL -> REFIT_SAMPLES draws from their mixtures, refitted by a fresh unconditional MDNN
        (input: a constant) whose single mixture is returned."""
        summaries = self._summarize(states, actions)
        mixtures = [self._correct_for_proposal(m, threshold) for m in self.model.predict_MoGs(summaries)]
        return mixtures[0] if len(mixtures) == 1 else self._refit(mixtures)

    def _correct_for_proposal(self, mog, threshold):
        """bayes_sim.py:135-145: prune, then posterior = mixture x prior / proposal (a uniform prior drops out)."""
        if self.proposal is None:
            return mog
        mog.prune_negligible_components(threshold=threshold)
        if isinstance(self.prior, pdf.Uniform):
            return mog / self.proposal
        if isinstance(self.prior, pdf.Gaussian):
            return (mog * self.prior) / self.proposal
        raise NotImplementedError

    def _refit(self, mixtures):
        """bayes_sim.py:149-179: equal shares of REFIT_SAMPLES from every trajectory's mixture, REFIT_EPOCHS
        sweeps of minibatch REFIT_MINIBATCH through the fit engine (the estimator's own hyper-parameters,
        a [128, 128] trunk on a constant input), the fitted model's mixture at that input."""
        src = self.model
        refit = MDNN(input_dim=1, output_dim=src.output_dim,
                     output_lows=src.output_lows.detach().cpu().numpy(),
                     output_highs=src.output_highs.detach().cpu().numpy(),
                     n_gaussians=src.n_gaussians, hidden_layers=(128, 128), lr=src.lr,
                     activation=src.activation, full_covariance=src.L_size > 0, device=src.device)
        share = int(BayesSim.REFIT_SAMPLES / len(mixtures))
        draws = torch.from_numpy(np.concatenate([m.gen(n_samples=share) for m in mixtures], axis=0))
        draws = draws.float().to(src.device)
        if MDNN.VERBOSE:
            print(f'Fitting posterior from {len(mixtures):d} mogs')
        const_in = torch.zeros(draws.shape[0], 1, device=draws.device)
        refit.run_training(const_in, draws,
                           BayesSim.REFIT_EPOCHS * BayesSim.REFIT_SAMPLES // BayesSim.REFIT_MINIBATCH,
                           BayesSim.REFIT_MINIBATCH)
        fitted = refit.predict_MoGs(const_in[0:1, :])
        assert len(fitted) == 1
        return fitted[0]
