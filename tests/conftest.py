import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')
    # the C-ABI library must exist for the load/symbol tests and the host
    # logic; build it once if this checkout has not been built yet
    lib = os.path.join(ROOT, 'bayes_sim_ig_amd', 'lib', 'libbsig_hip.so')
    if not os.path.exists(lib):
        import subprocess
        subprocess.run(['bash', os.path.join(ROOT, 'build.sh')], check=True, cwd=ROOT)


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
