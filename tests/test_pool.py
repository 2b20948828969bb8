"""pav_amd/csrc/pool.h - the optional helper threads of the scan driver's per-region loops (PAV_HOST_THREADS) - checked on the
host: built with g++ and run with 0, 1 and 5 helpers, plain and under ThreadSanitizer (CPU build only)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module', params=['plain', 'tsan'])
def pool_check(request, tmp_path_factory):
    exe = str(tmp_path_factory.mktemp('pool') / ('pool_check_' + request.param))
    flags = ['-O2'] if request.param == 'plain' else ['-O1', '-g', '-fsanitize=thread', '-fno-omit-frame-pointer']
    subprocess.run(['g++', *flags, '-std=c++17', '-pthread', '-o', exe, os.path.join(ROOT, 'tests', 'native', 'pool_check.cpp')],
                   check=True)
    return exe, request.param


@pytest.mark.parametrize('helpers', [0, 1, 5])
def test_every_index_once_and_no_deadlock(pool_check, helpers):
    exe, kind = pool_check
    env = dict(os.environ, TSAN_OPTIONS='halt_on_error=1 exitcode=66')
    out = subprocess.run([exe, str(helpers), '1500' if kind == 'plain' else '300'], capture_output=True, text=True, timeout=600,
                         env=env)
    assert out.returncode == 0, out.stderr[-4000:]
    assert out.stdout.startswith('ok ')
