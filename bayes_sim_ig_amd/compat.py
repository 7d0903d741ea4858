"""Register this package under the reference's module paths so that code
written against NVlabs/bayes-sim-ig imports the MI355X path unchanged:

    import bayes_sim_ig_amd.compat as compat; compat.install()
    from bayes_sim_ig.bayes_sim import BayesSim
    from bayes_sim_ig.models.mdnn import MDNN
    from bayes_sim_ig.utils.summarizers import summary_corrdiff
"""
import sys
import types


def install():
    from . import bayes_sim, mdnn, mdrff, pdf, rff, summarizers
    root = types.ModuleType('bayes_sim_ig')
    models = types.ModuleType('bayes_sim_ig.models')
    utils = types.ModuleType('bayes_sim_ig.utils')
    root.__path__, models.__path__, utils.__path__ = [], [], []
    table = {
        'bayes_sim_ig': root, 'bayes_sim_ig.models': models,
        'bayes_sim_ig.utils': utils, 'bayes_sim_ig.bayes_sim': bayes_sim,
        'bayes_sim_ig.models.mdnn': mdnn, 'bayes_sim_ig.models.mdrff': mdrff,
        'bayes_sim_ig.models.rff': rff, 'bayes_sim_ig.utils.pdf': pdf,
        'bayes_sim_ig.utils.summarizers': summarizers,
    }
    root.bayes_sim, root.models, root.utils = bayes_sim, models, utils
    models.mdnn, models.mdrff, models.rff = mdnn, mdrff, rff
    utils.pdf, utils.summarizers = pdf, summarizers
    sys.modules.update(table)
    return root
