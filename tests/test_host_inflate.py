"""The serial half of the device inflate (pav_amd/csrc/inflate_dev.h: a lane's walk over the deflate blocks of one BGZF member -
stored, fixed, dynamic; first-level tables, the canonical walk of the longer codes, tokens of up to three literals or a copy)
compiled for the host (tests/native/inflate_check.cpp, ASan + UBSan) against zlib: raw deflate streams written by zlib at every
level and strategy must decode to tokens that resolve to the text; corrupt streams must end in an error, never out of bounds; the
CRC-32 as the resolve kernel joins it (256 zero-padded pieces) must be zlib's.  The kernels that use these functions are on trial
in tests/test_gpu_bgzf.py."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp('ifl') / 'inflate_check')
    subprocess.run(['g++', '-std=c++17', '-O1', '-g', '-Wall', '-Wextra', '-Werror', '-fsanitize=address,undefined', '-fno-sanitize-recover=undefined',
                    os.path.join(ROOT, 'tests', 'native', 'inflate_check.cpp'), '-o', out, '-lz'], check=True)
    return out


def test_round_trips_corrupt_streams_and_the_checksum(exe):
    out = subprocess.run([exe, 'self'], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and out.stdout.strip() == 'ok self', out.stdout + out.stderr


@pytest.mark.parametrize('level', [1, 6, 9])
def test_a_fasta_file_in_bgzf_sized_members(exe, tmp_path, level):
    rng = np.random.default_rng(level)
    seq = rng.choice(np.frombuffer(b'ACGT', dtype=np.uint8), 1_500_000)
    seq[200_000:260_000] |= 0x20                          # soft-masked
    seq[700_000:790_000] = ord('N')
    seq[1_000_000:1_004_000] = np.resize(seq[1_000_000:1_000_007], 4000)
    raw = seq.tobytes()
    text = b'>chr1 test\n' + b''.join(raw[i:i + 60] + b'\n' for i in range(0, len(raw), 60))
    src, dst = tmp_path / 'x.fa', tmp_path / 'x.out'
    src.write_bytes(text)
    out = subprocess.run([exe, 'file', str(src), str(dst), str(level)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.startswith('ok'), out.stdout + out.stderr
    assert dst.read_bytes() == text
