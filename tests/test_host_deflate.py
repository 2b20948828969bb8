"""The serial parts of the device deflate encoder (pav_amd/csrc/deflate_dev.h: Huffman lengths with the length limit, canonical
codes, the dynamic block header, match symbols, the CRC-32 algebra) compiled for the host into a scalar encoder
(tests/native/deflate_check.cpp, ASan + UBSan) whose gzip members zlib must inflate back to the input.  The kernel that uses
these functions is on trial in tests/test_gpu_gzip.py."""
import gzip
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp('dfl') / 'deflate_check')
    subprocess.run(['g++', '-std=c++17', '-O1', '-g', '-Wall', '-Wextra', '-Werror', '-fsanitize=address,undefined', '-fno-sanitize-recover=undefined',
                    os.path.join(ROOT, 'tests', 'native', 'deflate_check.cpp'), '-o', out, '-lz'], check=True)
    return out


def test_tables_checksums_limits_and_round_trips(exe):
    out = subprocess.run([exe, 'self'], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and out.stdout.strip() == 'ok self', out.stdout + out.stderr


def test_a_table_file_round_trips_through_python_gzip(exe, tmp_path):
    rng = np.random.default_rng(9)
    rows = ''.join(f'chr{1 + i % 3}\t{p}\t{p + 1}\tchr{1 + i % 3}-{p + 1}-SNV-AG\tSNV\t1\tA\tG\th1\ttig{i % 7}:{p - 11}-{p - 11}\t+\t0\t{i % 40}\tCIGAR\tPASS\n'
                   for i, p in enumerate(np.sort(rng.integers(1, 10 ** 8, 30000)).tolist()))
    src, dst = tmp_path / 't.tsv', tmp_path / 't.tsv.gz'
    src.write_text(rows)
    out = subprocess.run([exe, 'file', str(src), str(dst)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert gzip.open(dst, 'rt').read() == rows
