"""pav_amd/csrc/pool.h - the optional helper threads of the scan driver's per-region loops (PAV_HOST_THREADS) - checked on the
host: built with g++ and run with 0, 1 and 5 helpers."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def pool_check(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp('pool') / 'pool_check')
    subprocess.run(['g++', '-O2', '-std=c++17', '-pthread', '-o', exe, os.path.join(ROOT, 'tests', 'native', 'pool_check.cpp')], check=True)
    return exe


@pytest.mark.parametrize('helpers', [0, 1, 5])
def test_every_index_once_and_no_deadlock(pool_check, helpers):
    out = subprocess.run([pool_check, str(helpers), '1500'], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.startswith('ok ')
