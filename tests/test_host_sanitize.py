"""CPU sanitizers on the native host code (no GPU, no HIP call):

* the CPU oracle (oracle/*.c) built with AddressSanitizer + UndefinedBehaviorSanitizer and run against the golden vectors of the
  reference - the checker itself must not read outside its buffers;
* the file readers of libpav_amd (pav_amd/csrc/fastaio.cpp, bedio.cpp, samio.cpp: the code that parses files the library does
  not control) built with ASan + UBSan into tests/native/hostio_check.cpp and run on well-formed files (digests compared with a
  Python reading), on the malformed inputs of tests/test_host_*.py, and on a seeded sweep of damaged copies (flipped bytes,
  truncations, spliced lines): a reader parses or refuses with a message, a sanitizer report fails the test.

pool.h under ThreadSanitizer: tests/test_pool.py.  GPU sanitizers are not available on the pool (and not asked for)."""
import gzip
import json
import os
import subprocess
import sys
import zlib

import numpy as np
import pandas as pd
import pytest

import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = util.GOLD
CSRC = os.path.join(ROOT, 'pav_amd', 'csrc')
SAN = ['-fsanitize=address,undefined', '-fno-sanitize-recover=undefined', '-fno-omit-frame-pointer', '-g', '-O1']
ENV = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0:exitcode=77', UBSAN_OPTIONS='print_stacktrace=1:halt_on_error=1')


@pytest.fixture(scope='module')
def hostio(tmp_path_factory):
    d = tmp_path_factory.mktemp('hostio')
    exe = str(d / 'hostio_check')
    srcs = [os.path.join(ROOT, 'tests', 'native', 'hostio_check.cpp')] + [os.path.join(CSRC, f) for f in ('fastaio.cpp', 'bedio.cpp', 'samio.cpp')]
    subprocess.run(['g++', '-std=c++17', *SAN, '-D__HIP_PLATFORM_AMD__', '-I/opt/rocm/include', *srcs, '-o', exe, '-lz', '-pthread'],
                   check=True)
    return exe


def run(exe, kind, *paths):
    out = subprocess.run([exe, kind, *map(str, paths)], capture_output=True, timeout=600, env=ENV)
    assert out.returncode == 0, f'{kind} {paths[:3]}...: exit {out.returncode}\n{out.stderr.decode(errors="replace")[-6000:]}'
    lines = out.stdout.decode(errors='replace').split('\n')[:-1]         # (a message may quote bytes of a damaged file)
    assert len(lines) == len(paths)
    return lines


def fields(line):
    assert line.startswith('ok '), line
    return dict(kv.split('=') for kv in line.split()[1:])


def crc_names(names):
    c = 0
    for n in names:
        c = zlib.crc32(n.encode() + b'\0', c)
    return '%08x' % c


def read_fasta_py(path):
    raw = open(path, 'rb').read()
    if raw[:2] == b'\x1f\x8b':
        raw = gzip.decompress(raw)
    names, seqs = [], []
    for ln in raw.replace(b'\r', b'').split(b'\n'):
        if ln.startswith(b'>'):
            names.append(ln[1:].split()[0].decode() if ln[1:].split() else '')
            seqs.append([])
        elif seqs:
            seqs[-1].append(ln)
    return names, [b''.join(s) for s in seqs]


def test_readers_on_wellformed_files(built, hostio, tmp_path):
    # FASTA: golden files plain, and re-written as gzip / BGZF-like multi-member gzip with CRLF and no final newline
    fas = [os.path.join(GOLD, 'inv_fwd', 'ref.fa'), os.path.join(GOLD, 'inv_hap', 'tig.fa')]
    text = open(fas[1], 'rb').read()
    gz = tmp_path / 'tig.fa.gz'
    gz.write_bytes(gzip.compress(text.replace(b'\n', b'\r\n').rstrip(b'\r\n')))
    mm = tmp_path / 'tig_members.fa.gz'
    mm.write_bytes(b''.join(gzip.compress(text[a:a + 50_000]) for a in range(0, len(text), 50_000)))
    fas += [str(gz), str(mm)]
    for path, line in zip(fas, run(hostio, 'fasta', *fas)):
        names, seqs = read_fasta_py(path)
        f = fields(line)
        c = 0
        for s in seqs:
            c = zlib.crc32(s, c)
        assert (int(f['records']), int(f['bases']), f['names_crc'], f['seq_crc']) == (len(names), sum(map(len, seqs)), crc_names(names), '%08x' % c), path
    # alignment tables
    beds = [os.path.join(GOLD, p) for p in ('cigar_synth/align.tsv', 'trim_overlap/align_none.tsv.gz', 'flag_hap/align.tsv', 'lgsv_hap/align.tsv.gz')]
    for path, line in zip(beds, run(hostio, 'bed', *beds)):
        df = pd.read_csv(path, sep='\t', dtype={'#CHROM': str, 'QRY_ID': str}, keep_default_na=False, low_memory=False)
        f = fields(line)
        cig = ''.join(df['CIGAR']).encode()
        assert (int(f['rows']), int(f['pos']), int(f['end']), int(f['cigar_bytes']), f['cigar_crc']) == \
            (df.shape[0], int(df['POS'].sum()), int(df['END'].sum()), len(cig), '%08x' % zlib.crc32(cig)), path
        assert f['names_crc'] == crc_names(list(dict.fromkeys(df['#CHROM'])) + list(dict.fromkeys(df['QRY_ID']))), path
    # SAM
    sams = [os.path.join(GOLD, 'align_ingest', n) for n in ('hap_noseq.sam.gz', 'hap_seq.sam.gz')]
    for path, line in zip(sams, run(hostio, 'sam', *sams)):
        recs = [ln for ln in gzip.decompress(open(path, 'rb').read()).decode().splitlines() if ln and not ln.startswith('@')]
        f = fields(line)
        assert int(f['records']) == len(recs) and 0 < int(f['rows']) <= len(recs), path


def test_readers_refuse_the_malformed_inputs_of_the_host_tests(built, hostio, tmp_path):
    """The error cases of tests/test_host_fasta.py::test_reader_errors, test_host_next.py::test_native_reader_errors and
    test_host_ingest.py (malformed CIGAR, the reference's six offending records), under the sanitizers."""
    from test_host_fasta import bgzf_bytes, make_text
    blob = bytearray(bgzf_bytes(make_text(5)))
    blob[40] ^= 0xff
    (tmp_path / 'bad.fa.gz').write_bytes(bytes(blob))
    (tmp_path / 'trunc.fa.gz').write_bytes(gzip.compress(make_text(6))[:-20])
    (tmp_path / 'empty.fa').write_bytes(b'')
    (tmp_path / 'only_header.fa').write_bytes(b'>')
    (tmp_path / 'no_header.fa').write_bytes(b'ACGT\nACGT')
    lines = run(hostio, 'fasta', tmp_path / 'missing.fa', tmp_path / 'bad.fa.gz', tmp_path / 'trunc.fa.gz', tmp_path / 'empty.fa',
                tmp_path / 'only_header.fa', tmp_path / 'no_header.fa')
    assert 'cannot open' in lines[0] and 'corrupt' in lines[1] and lines[2].startswith('error') and lines[3].startswith('ok ')
    (tmp_path / 'a.tsv').write_text('#CHROM\tPOS\tEND\tREV\tCIGAR\nchr1\t10\tx20\tTrue\t5=\n')
    (tmp_path / 'b.tsv').write_text('#CHROM\tPOS\nchr1\t10\t99\n')
    (tmp_path / 'c.tsv').write_text('#CHROM\tPOS\tEND\tREV\tCIGAR\nchr1\t10\n\n\t\t\t\t\t\t\t\t\nchr1')
    (tmp_path / 'd.tsv').write_text('')
    (tmp_path / 'e.tsv').write_text('#CHROM\tPOS\tEND\tREV\tCIGAR')
    (tmp_path / 'f.tsv.gz').write_bytes(gzip.compress(b'#CHROM\tPOS\tEND\tREV\tCIGAR\nchr1\t1\t2\tTrue\t1=\n' * 400)[:-9])
    lines = run(hostio, 'bed', *(tmp_path / n for n in ('a.tsv', 'b.tsv', 'c.tsv', 'd.tsv', 'e.tsv', 'f.tsv.gz', 'missing.tsv.gz')))
    assert "cannot parse 'x20' in column END" in lines[0] and 'has no CIGAR column' in lines[1] and 'cannot open' in lines[6]
    with open(os.path.join(GOLD, 'align_ingest', 'errors.json')) as fh:
        cases = json.load(fh)
    paths = []
    for key, c in cases.items():
        p = tmp_path / (key + '.sam')
        p.write_text('@HD\tVN:1.6\n' + c['sam'] + '\n')
        paths.append(p)
    (tmp_path / 'bad.sam').write_text('t\t0\tchr1\t1\t60\t10=5\t*\t0\t0\t*\t*\n')
    (tmp_path / 'short.sam').write_text('t\t0\tchr1\n\n@CO\tx\nu\t0\tchr1\t1\t60\n')
    (tmp_path / 'empty.sam.gz').write_bytes(b'')
    lines = run(hostio, 'sam', *paths, tmp_path / 'bad.sam', tmp_path / 'short.sam', tmp_path / 'empty.sam.gz')
    assert 'malformed CIGAR' in lines[len(paths)]


def damaged(raw, rng, is_text):
    """One damaged copy of a file's bytes: flipped bytes, a truncation, a duplicated or dropped stretch, or (text) fields
    replaced by junk."""
    b = bytearray(raw)
    kind = int(rng.integers(0, 5))
    if kind == 0:
        for _ in range(int(rng.integers(1, 12))):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
    elif kind == 1:
        b = b[:int(rng.integers(0, len(b)))]
    elif kind == 2:
        a, n = int(rng.integers(0, len(b))), int(rng.integers(1, 2000))
        b[a:a] = b[a:a + n]
    elif kind == 3:
        a, n = int(rng.integers(0, len(b))), int(rng.integers(1, 2000))
        del b[a:a + n]
    else:
        junk = [b'\t', b'\n', b'-1', b'99999999999999999999', b'\x00', b'\r', b'>', b'@', b'*', b'4294967296M', b'0=', b'=', b'\xff\xfe']
        for _ in range(int(rng.integers(1, 8))):
            a = int(rng.integers(0, len(b)))
            j = junk[int(rng.integers(0, len(junk)))]
            b[a:a + (len(j) if is_text and rng.random() < 0.5 else 0)] = j
    return bytes(b)


@pytest.mark.parametrize('kind', ['fasta', 'bed', 'sam'])
def test_readers_on_damaged_files(built, hostio, tmp_path, kind):
    """A seeded sweep: 150 damaged copies per format (plain text and gzip - damage applied to the text before compressing and
    to the compressed bytes).  Every file is parsed or refused; no sanitizer report."""
    src = {'fasta': os.path.join(GOLD, 'inv_small', 'tig.fa'), 'bed': os.path.join(GOLD, 'cigar_synth', 'align.tsv'),
           'sam': os.path.join(GOLD, 'align_ingest', 'hap_seq.sam.gz')}[kind]
    raw = open(src, 'rb').read()
    if raw[:2] == b'\x1f\x8b':
        raw = gzip.decompress(raw)
    if kind == 'fasta':
        raw = raw[:30_000]
    rng = np.random.default_rng({'fasta': 11, 'bed': 12, 'sam': 13}[kind])
    paths = []
    for i in range(150):
        mode = i % 3
        if mode == 0:
            p, data = tmp_path / f'd{i}.txt', damaged(raw, rng, True)
        elif mode == 1:
            p, data = tmp_path / f'd{i}.gz', gzip.compress(damaged(raw, rng, True), 1)
        else:
            p, data = tmp_path / f'd{i}.gz', damaged(gzip.compress(raw, 1), rng, False)
        p.write_bytes(data)
        paths.append(p)
    lines = run(hostio, kind, *paths)
    n_ok = sum(ln.startswith('ok ') for ln in lines)
    assert all(ln.startswith(('ok ', 'error ')) for ln in lines)
    assert 0 < n_ok < len(lines) or kind == 'fasta', (kind, n_ok)        # (almost any byte soup is a FASTA file)


@pytest.mark.parametrize('san', ['asan+ubsan', 'tsan'])
def test_table_writer_helpers_under_sanitizers(built, tmp_path, san):
    """pav_amd/csrc/textio.h - integer and repr(float) formatting, csv quoting, the chunked writer with its worker threads and
    gzip members - built into tests/native/textio_check.cpp with ASan + UBSan / with TSan: 150 k rows (two chunks and a half)
    of extreme and random values, one thread and seven, plain text equal to what pandas writes, the gzip file equal to the plain."""
    import io
    import struct
    exe = str(tmp_path / 'textio_check')
    flags = SAN if san == 'asan+ubsan' else ['-fsanitize=thread', '-fno-omit-frame-pointer', '-g', '-O1']
    subprocess.run(['g++', '-std=c++17', *flags, '-D__HIP_PLATFORM_AMD__', '-I/opt/rocm/include',
                    os.path.join(ROOT, 'tests', 'native', 'textio_check.cpp'), '-o', exe, '-lz', '-pthread'], check=True)
    rng = np.random.default_rng(21)
    n = 150_000
    ints = np.concatenate([[0, -1, 1, np.iinfo(np.int64).min, np.iinfo(np.int64).max, 10 ** 18, -10 ** 18], rng.integers(-2 ** 62, 2 ** 62, n - 7)]).astype(np.int64)
    flts = np.concatenate([[0.0, -0.0, 1.0, 1e16, 1e15, 1e-4, 1e-5, 5e-324, 1.7976931348623157e308, np.nan, np.inf, -np.inf, 0.1, 1 / 3],
                           np.exp(rng.uniform(-700, 700, n - 14)) * rng.choice([-1, 1], n - 14)])
    words = ['', 'chr1', 'a\tb', 'say "x"', 'line\nbreak', 'plain text', 'ACGT' * 50]
    texts = [words[int(i)] for i in rng.integers(0, len(words), n)]
    blob = io.BytesIO()
    blob.write(struct.pack('<Q', n))
    for a, b, t in zip(ints.tolist(), flts.tolist(), texts):
        tb = t.encode()
        blob.write(struct.pack('<qdI', a, b, len(tb)))
        blob.write(tb)
    (tmp_path / 'values.bin').write_bytes(blob.getvalue())
    want = pd.DataFrame({'ROW': np.arange(n), 'INT': ints, 'FLOAT': flts, 'TEXT': texts}).to_csv(sep='\t', index=False, lineterminator='\n')
    env = dict(ENV, TSAN_OPTIONS='halt_on_error=1 exitcode=66')
    for threads in (1, 7):
        out = subprocess.run([exe, str(tmp_path / 'values.bin'), str(tmp_path / 'out.tsv'), str(tmp_path / 'out.tsv.gz'), str(threads)],
                             capture_output=True, text=True, timeout=900, env=env)
        assert out.returncode == 0, out.stderr[-5000:]
        got = (tmp_path / 'out.tsv').read_bytes()
        assert gzip.decompress((tmp_path / 'out.tsv.gz').read_bytes()) == got
        # ('\\r' inside a field is left out: csv.QUOTE_MINIMAL quotes it from Python 3.13 on and not before; no PAV column can hold one)
        # pandas quotes the empty string field as "" (QUOTE_MINIMAL on an all-empty trailing field); the writers never emit an
        # empty TEXT (FLANK / MATCH are written by their own code): compare with those fields normalised
        assert got.decode().replace('\t\n', '\t""\n') == want.replace('\t\n', '\t""\n')


def test_oracle_under_asan_and_ubsan(built, tmp_path):
    """oracle/*.c rebuilt with -fsanitize=address,undefined and run - in a child interpreter that preloads the sanitizer
    runtime - over the golden vectors: the CIGAR cases and error cases, homology known answers, every scan iteration of the small
    inversion cases and the near-tie tables."""
    subprocess.run(['make', '-C', os.path.join(ROOT, 'oracle'), '-s', 'asan'], check=True)
    lib = os.path.join(ROOT, 'oracle', '_build', 'libpavoracle_asan.so')
    asan_rt = subprocess.run(['gcc', '-print-file-name=libasan.so'], capture_output=True, text=True, check=True).stdout.strip()
    env = dict(ENV, LD_PRELOAD=asan_rt, PAV_ORACLE_LIB=lib, ASAN_OPTIONS='detect_leaks=0:exitcode=77', PYTHONPATH=ROOT)
    tests = ['tests/test_oracle_cigar.py', 'tests/test_host_inv.py::test_oracle_density_matches_reference',
             'tests/test_host_inv.py::test_oracle_density_on_constructed_near_ties']
    out = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-p', 'no:cacheprovider', *tests], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-6000:]
    assert ' passed' in out.stdout
    which = subprocess.run([sys.executable, '-c', 'from oracle import oracle; oracle.load(); print(oracle.LIB_PATH); '
                            'print(open("/proc/self/maps").read().count("libpavoracle_asan.so") > 0)'], cwd=ROOT, env=env,
                           capture_output=True, text=True, timeout=120)
    assert which.stdout.split() == [lib, 'True'], which.stdout + which.stderr
