"""bench.py --gpus N on a node (no GPU needed): the launcher's command line, and the refusal of a rank
count the node cannot seat -- before any GPU call, from a process that starts children and never
re-executes itself."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def test_rank_launch_command_is_the_drivers_form():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.rank_launch_command(4, ['--gpus', '4', '--steps', '2', '--warmup', '1'], 29123)
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nnodes=1' in cmd
    assert cmd[cmd.index('--nproc-per-node') + 1] == '4'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[cmd.index('--master-port') + 1] == '29123'
    script = cmd.index(os.path.join(ROOT, 'bench.py'))
    assert cmd[script + 1:] == ['--gpus', '4', '--steps', '2', '--warmup', '1']


def test_more_ranks_than_gpus_is_refused_before_any_gpu_call():
    """`python bench.py --gpus 64`: no node of this pool has 64 GPUs (this container has none).  The
    process must say so and exit non-zero without importing the HIP library or starting a rank."""
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    code = ("import sys, runpy; sys.argv = ['bench.py', '--gpus', '64']\n"
            "try:\n"
            "    runpy.run_path(%r, run_name='__main__')\n"
            "except SystemExit as e:\n"
            "    print('EXIT', e.code)\n"
            "print('HIPLIB', 'bayes_sim_ig_amd._lib' in sys.modules and "
            "sys.modules['bayes_sim_ig_amd._lib']._lib is not None)\n" % os.path.join(ROOT, 'bench.py'))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, timeout=300)
    assert 'EXIT bench.py --gpus 64: this node shows' in r.stdout, (r.stdout, r.stderr)
    assert 'one rank per GPU' in r.stdout
    assert 'HIPLIB False' in r.stdout, r.stdout
