"""Mixture Density NN on Random Fourier Features — mirror of the reference's
bayes_sim_ig/models/mdrff.py (class MDRFF).  ``forward`` = RFF projection
(fp32-MFMA GEMM + sincos epilogue) followed by the MDNN heads.  The reference
projects the gathered minibatch inside every update (mdrff.py:28-30 inside
mdnn.py:231); the features are a pure function of the row, so ``run_training``
projects every distinct row of the chunk ONCE per call (feature cache,
csrc/estimator.hip) and ``BayesSim.fit`` projects blocks of chunks in one GEMM --
bitwise the same features, 12.5x fewer row projections."""
import torch

from .mdnn import MDNN
from .rff import RFF


class MDRFF(MDNN):
    def __init__(self, input_dim, output_dim, output_lows, output_highs,
                 n_gaussians, lr, activation, full_covariance, device='cpu',
                 n_feat=500, kernel='RBF', sigma=1.0, freqs=None, **kwargs):
        """Same arguments as the reference (mdrff.py:15-17); ``freqs``
        optionally injects the [n_feat/2, input_dim] frequency matrix."""
        self._rff_input_dim = input_dim
        a = (2.0 / float(n_feat)) ** 0.5          # rff.py:107
        super().__init__(n_feat, output_dim, output_lows, output_highs,
                         n_gaussians, hidden_layers=[], lr=lr,
                         activation=activation,
                         full_covariance=full_covariance, device=device,
                         _rff_feats=n_feat, _rff_scale=a)
        # heads see n_feat inputs; the estimator's rows are input_dim wide
        self.input_dim = input_dim
        self.rff = RFF(n_feat, input_dim, sigma, cos_only=False,
                       quasi_random=False if input_dim > 100 else True,
                       kernel=kernel, device=device, freqs=freqs)
        self._rff_scale = float(self.rff.a)
        if MDNN.VERBOSE:
            print('MDRFF n_feat', n_feat, 'sigma', sigma)
            print(self)

    def _cfg(self):
        cfg = super()._cfg()
        cfg.input_dim = int(self._rff_input_dim)
        return cfg

    def _rff_args(self):
        co = self.rff.coeff()
        return co, co.stride(0), None

    def _dp_sync_extra(self):
        """Data-parallel replicas must share the feature map: the frequencies come from
        the global numpy RNG (rff.py:111-120), which ranks may have seeded differently, and
        are not part of the flat parameter buffer -- rank 0's are broadcast."""
        rff = self.rff
        for name in ('freqs', 'sigma', 'offset'):
            t = getattr(rff, name)
            if t is None:
                continue
            flat = t.to(self._flat.device, torch.float32).contiguous().view(-1).clone()
            self._dp.broadcast(flat)
            setattr(rff, name, flat.view(t.shape).to(t.device))
        rff._coeff = None
