"""The text primitives of the device table writer (pav_amd/csrc/fmt_dev.h: decimal integers, repr(float) by the Ryu algorithm)
compiled for the host (tests/native/fmt_check.cpp) and compared with (a) the host writer's formatter - std::to_chars, which the
reference-text goldens pin (tests/test_gpu_inv.py::test_native_density_tables_equal_pandas_text) - on a seeded sweep of every
kind of double, and (b) Python's own repr(), which is what pandas.DataFrame.to_csv writes (rules/call_inv.snakefile:287-291).
The generated table header must equal what its generator computes."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def fmt_check(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp('fmt') / 'fmt_check')
    subprocess.run(['g++', '-std=c++17', '-O2', '-Wall', '-Wextra', '-Werror', os.path.join(ROOT, 'tests', 'native', 'fmt_check.cpp'), '-o', exe],
                   check=True)
    return exe


def test_ryu_tables_are_what_the_generator_computes():
    sys.path.insert(0, os.path.join(ROOT, 'tools', 'gen'))
    import gen_ryu_tables
    with open(os.path.join(ROOT, 'pav_amd', 'csrc', 'ryu_tables.h')) as fh:
        assert fh.read() == gen_ryu_tables.render()
    inv, pw = gen_ryu_tables.tables()
    assert all(v < 2 ** 128 for v in inv + pw) and all(v.bit_length() == 125 for v in pw)


def test_integers_equal_printf(fmt_check):
    out = subprocess.run([fmt_check, 'ints'], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == 'ok ints', out.stdout + out.stderr


def test_float_repr_equals_to_chars_on_a_sweep(fmt_check):
    out = subprocess.run([fmt_check, 'sweep', '3000000', '20261003'], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.startswith('ok '), out.stdout + out.stderr


def test_float_repr_equals_python_repr(fmt_check):
    rng = np.random.default_rng(5)
    vals = np.concatenate([
        rng.integers(0, 2 ** 63, 40000, dtype=np.uint64).view(np.float64),                 # any pattern (NaN among them)
        rng.random(20000), np.exp(-rng.random(20000) * 745.0),                             # densities and their tails
        np.ldexp(rng.random(20000), rng.integers(-1074, 1024, 20000)),
        rng.integers(0, 2 ** 53, 5000).astype(np.float64), np.array([0.0, -0.0, np.inf, -np.inf, 5e-324, 1e16, 1e-4, 9.999e-5, 1e22, 1e23]),
        -rng.random(2000)])
    text = ''.join('%016x\n' % struct.unpack('<Q', struct.pack('<d', float(v)))[0] for v in vals)
    out = subprocess.run([fmt_check, 'repr'], input=text, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0
    got = out.stdout.split('\n')[:-1]
    assert len(got) == len(vals)
    for v, g in zip(vals.tolist(), got):
        want = '' if v != v else repr(v)
        assert g == want, (v.hex() if v == v else 'nan', g, want)
