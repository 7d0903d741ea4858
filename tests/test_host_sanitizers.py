"""The host side of libbsig_hip (argument checks, parameter layouts, GEMM planners, persistent-kernel
geometry, plan binding, the external-exchange communicator) under AddressSanitizer +
UndefinedBehaviorSanitizer: tools/build_host_san.sh builds every source with host-side sanitizers
(device code as usual; GPU sanitizers are not available on the MI355X pool) and
tests/host/host_san_main.cpp drives the C ABI without launching anything.  Runs without a GPU."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.skipif(shutil.which('hipcc') is None, reason='hipcc not on PATH')
def test_host_logic_is_clean_under_asan_and_ubsan():
    if os.environ.get('BSIG_SKIP_SANITIZER_BUILD') == '1':
        pytest.skip('BSIG_SKIP_SANITIZER_BUILD=1')
    build = subprocess.run(['bash', os.path.join(ROOT, 'tools', 'build_host_san.sh')], cwd=ROOT,
                           capture_output=True, text=True, timeout=1500)
    assert build.returncode == 0, build.stdout[-2000:] + build.stderr[-4000:]
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=0:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1')
    run = subprocess.run([os.path.join(ROOT, 'build', 'host_san', 'host_san_test')], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=300)
    out = run.stdout + run.stderr
    assert run.returncode == 0 and 'host sanitizer test: ok' in out, out[-6000:]
    assert 'runtime error' not in out and 'AddressSanitizer' not in out, out[-6000:]
