"""bayes_sim_ig_amd — the BayesSim posterior-estimator training path of
NVlabs/bayes-sim-ig on AMD Instinct MI355X (gfx950).

Host-side mirror of the reference's Python interface for this path
(summarizers, RFF, MDNN, MDRFF, BayesSim, pdf) over libbsig_hip.so
(include/bsig.h, csrc/*.hip).  ``compat.install()`` registers the
reference's module paths (``bayes_sim_ig.models.mdnn`` ...) as aliases.
"""
from . import _lib, pairs, pdf, summarizers   # noqa: F401
from .bayes_sim import BayesSim               # noqa: F401
from .mdnn import MDNN                        # noqa: F401
from .mdrff import MDRFF                      # noqa: F401
from .rff import RFF                          # noqa: F401
from .summarizers import (pad_states_actions, summary_start, summary_waypts,  # noqa: F401
                          summary_corr, summary_corrdiff, summary_signatory,
                          signature_depth, cross_correlation)
