"""Mixture Density NN on Random Fourier Features — mirror of the reference's
bayes_sim_ig/models/mdrff.py (class MDRFF).  ``forward`` = RFF projection
(fp32-MFMA GEMM + sincos epilogue) followed by the MDNN heads; in
``run_training`` the projection runs inside every captured update, on the
gathered minibatch rows, exactly where the reference computes it
(mdrff.py:28-30 inside mdnn.py:231)."""
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
