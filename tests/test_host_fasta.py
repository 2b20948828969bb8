"""Native FASTA reader (pav_fasta_open, csrc/fastaio.cpp) against a plain-Python reading of the same files: plain text, gzip
(one and several members), BGZF with parallel inflate, CRLF, missing final newline, empty records, '>' inside lines.
No GPU needed: the reader is host code of the library.  Semantics = what pysam.FastaFile(...).fetch(name) returns for whole
records (pavlib/cigarcall.py:59-66): name = first word of the header, sequence = the lines joined, case kept."""
import gzip
import os
import struct
import zlib

import numpy as np
import pytest

from pav_amd import _lib
from pav_amd.fasta import Fasta, open_fasta

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def python_records(text):
    """[(name, sequence bytes)] of FASTA text, the slow obvious way."""
    recs = []
    for line in text.split(b'\n'):
        if line.startswith(b'>'):
            words = line[1:].split()
            recs.append([words[0].decode() if words else '', []])
        elif recs:
            recs[-1][1].append(line.replace(b'\r', b''))
    return [(n, b''.join(parts)) for n, parts in recs]


def bgzf_bytes(data, block=0xff00):
    """BGZF container (SAM specification 4.1) of ``data``, incl. the empty end-of-file block."""
    out = []
    for i in list(range(0, len(data), block)) + [None]:
        chunk = b'' if i is None else data[i:i + block]
        comp = zlib.compressobj(6, zlib.DEFLATED, -15)
        payload = comp.compress(chunk) + comp.flush()
        bsize = 12 + 6 + len(payload) + 8
        out.append(struct.pack('<4BI2BH', 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6) + b'BC' + struct.pack('<HH', 2, bsize - 1) + payload +
                   struct.pack('<II', zlib.crc32(chunk), len(chunk)))
    return b''.join(out)


def make_text(seed, n_rec=7, crlf=False, final_newline=True):
    rng = np.random.default_rng(seed)
    parts = []
    alphabet = np.frombuffer(b'ACGTacgtNnRYKM', dtype=np.uint8)
    for r in range(n_rec):
        n = int(rng.integers(0, 200_000)) if r != 3 else 0                     # record 3 is empty
        width = int(rng.choice([50, 60, 61, 80, 1000]))
        seq = alphabet[rng.integers(0, alphabet.size, n)].tobytes()
        parts.append(b'>rec%d some description > with a bracket\n' % r if r % 2 else b'>rec%d\n' % r)
        for i in range(0, n, width):
            parts.append(seq[i:i + width] + b'\n')
        if r == 5:
            parts.append(b'\n')                                                    # blank line inside the file
    text = b''.join(parts)
    if crlf:
        text = text.replace(b'\n', b'\r\n')
    if not final_newline:
        text = text.rstrip(b'\r\n')
    return text


@pytest.mark.parametrize('container', ['plain', 'gzip', 'gzip-members', 'bgzf'])
@pytest.mark.parametrize('crlf,final_newline', [(False, True), (True, True), (False, False)])
def test_native_reader_equals_python(built, tmp_path, container, crlf, final_newline):
    text = make_text(11 + crlf + 2 * final_newline, crlf=crlf, final_newline=final_newline)
    path = str(tmp_path / ('x.fa' if container == 'plain' else 'x.fa.gz'))
    if container == 'plain':
        blob = text
    elif container == 'gzip':
        blob = gzip.compress(text, 1)
    elif container == 'gzip-members':
        cut = len(text) // 3
        blob = gzip.compress(text[:cut], 6) + gzip.compress(text[cut:], 1)
    else:
        blob = bgzf_bytes(text)
        assert gzip.decompress(blob) == text                                      # the helper writes valid gzip members
    with open(path, 'wb') as fh:
        fh.write(blob)
    want = python_records(text)
    for threads in (1, 4):
        fa = _lib.FastaFile(path, threads=threads)
        assert fa.kind == {'plain': 'plain', 'gzip': 'gzip', 'gzip-members': 'gzip', 'bgzf': 'bgzf'}[container]
        assert fa.names == [n for n, _ in want]
        assert fa.lengths == [len(s) for _, s in want]
        for i, (_, s) in enumerate(want):
            assert fa.seq(i).tobytes() == s, i
        fa.close()


def test_fasta_class_matches_the_golden_files(built):
    """The host class the rules use (pav_amd.fasta.Fasta), on committed fixtures: names, lengths (= .fai), fetch()."""
    d = os.path.join(GOLD, 'inv_fwd')
    fa = open_fasta(os.path.join(d, 'tig.fa'))
    with open(os.path.join(d, 'tig.fa'), 'rb') as fh:
        want = dict(python_records(fh.read()))
    assert fa.names == list(want)
    fai = {ln.split('\t')[0]: int(ln.split('\t')[1]) for ln in open(os.path.join(d, 'tig.fa.fai'))}
    assert fa.lengths() == fai
    name = fa.names[0]
    assert fa.fetch(name) == want[name].decode() and fa.fetch(name, 5, 25) == want[name][5:25].decode()
    assert fa.record_numbers([name]) == [0]


def test_reader_errors(built, tmp_path):
    with pytest.raises(_lib.PavDeviceError, match='cannot open'):
        _lib.FastaFile(str(tmp_path / 'missing.fa'))
    bad = tmp_path / 'bad.fa.gz'
    blob = bytearray(bgzf_bytes(make_text(5)))
    blob[40] ^= 0xff                                                              # damage the first block's deflate stream
    bad.write_bytes(bytes(blob))
    with pytest.raises(_lib.PavDeviceError, match='corrupt'):
        _lib.FastaFile(str(bad))
    trunc = tmp_path / 'trunc.fa.gz'
    trunc.write_bytes(gzip.compress(make_text(6))[:-20])
    with pytest.raises(_lib.PavDeviceError):
        _lib.FastaFile(str(trunc))
    empty = tmp_path / 'empty.fa'
    empty.write_bytes(b'')
    fa = Fasta(str(empty))
    assert fa.names == [] and fa.seqs == {}
