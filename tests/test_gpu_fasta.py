"""The device-side FASTA loader (pav_amd/csrc/fastadev.hip: pav_seq_load_fasta_path - the file's text is uploaded as it is and
loses its header lines and line breaks on the device) against the host parser (pav_fasta_open, itself equal to a line-by-line
Python reading: tests/test_host_fasta.py): same record names, same lengths, same bytes in the store (read back with
pav_seq_fetch), for plain / gzip / BGZF files with CRLF line ends, blank lines, empty records, a '>' inside a sequence line, no
final newline - and both roles of one context loaded from two threads at the same time."""
import gzip
import os
import struct
import threading
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bgzf(data, block=40000):
    out = bytearray()
    for a in list(range(0, len(data), block)) + [None]:
        chunk = b'' if a is None else data[a:a + block]
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = c.compress(chunk) + c.flush()
        out += b'\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00' + struct.pack('<H', len(body) + 25)
        out += body + struct.pack('<II', zlib.crc32(chunk), len(chunk))
    return bytes(out)


def make_text(rng, n_rec, crlf=False, line=60, final_newline=True):
    nl = b'\r\n' if crlf else b'\n'
    parts, want = [], []
    for r in range(n_rec):
        length = int(rng.choice([0, 1, 59, 60, 61, 1000, 70001, 300007]))
        seq = rng.choice(np.frombuffer(b'ACGTacgtNnRY', dtype=np.uint8), length).tobytes()
        name = f'rec{r}_{length}'
        parts.append(b'>' + name.encode() + (b' some description\twith tabs' if r % 3 == 0 else b'') + nl)
        body = bytearray()
        for i in range(0, length, line):
            body += seq[i:i + line] + nl
            if r % 4 == 1 and i == line:
                body += nl                                   # a blank line inside a record
        if r % 5 == 2 and length > 200:
            seq = seq[:100] + b'>' + seq[101:]                # '>' inside a sequence line is sequence
            body = bytearray()
            for i in range(0, length, line):
                body += seq[i:i + line] + nl
        parts.append(bytes(body))
        want.append((name, seq))
    text = b''.join(parts)
    if not final_newline and text.endswith(nl):
        text = text[:-len(nl)]
    return text, want


@pytest.fixture(scope='module')
def ctx():
    from pav_amd import _lib
    with _lib.Context(0) as c:
        yield c


@pytest.mark.parametrize('kind', ['plain', 'plain_crlf', 'plain_nofinal', 'gzip', 'bgzf'])
def test_device_loader_equals_the_host_parser(ctx, tmp_path, kind):
    from pav_amd import _lib
    rng = np.random.default_rng({'plain': 1, 'plain_crlf': 2, 'plain_nofinal': 3, 'gzip': 4, 'bgzf': 5}[kind])
    text, want = make_text(rng, 23, crlf=kind == 'plain_crlf', final_newline=kind != 'plain_nofinal')
    path = str(tmp_path / ('x.fa' + ('.gz' if kind in ('gzip', 'bgzf') else '')))
    with open(path, 'wb') as fh:
        fh.write(gzip.compress(text) if kind == 'gzip' else bgzf(text) if kind == 'bgzf' else text)
    host = _lib.FastaFile(path)
    assert host.names == [n for n, _ in want]
    for role in (_lib.PAV_ROLE_REF, _lib.PAV_ROLE_TIG):
        names = ctx.seq_load_fasta_path(role, path)
        assert names == host.names
        assert ctx.seq_lengths(role) == host.lengths == [len(s) for _, s in want]
        for i, (_, seq) in enumerate(want):
            got = ctx.seq_fetch(role, i, 0, len(seq)).tobytes()
            assert got == seq == host.seq(i).tobytes(), (kind, i)
            if len(seq) > 300:
                assert ctx.seq_fetch(role, i, 17, 290).tobytes() == seq[17:290]
    with pytest.raises(_lib.PavDeviceError):
        ctx.seq_fetch(_lib.PAV_ROLE_TIG, 0, 0, 10 ** 9)


def test_both_roles_from_two_threads_and_the_calls_that_follow(ctx, tmp_path):
    """A synthetic haplotype written as FASTA: reference and contigs go up side by side through the device loader, the CIGAR
    calls made from the stores equal the ones made from stores loaded from the generator's arrays."""
    import __graft_entry__ as g
    g.build_cpu_side()
    from pav_amd import _lib, cigarcall, synth
    hap = synth.config2(seed=61, scale=0.01, threads=4)
    ref_fa, tig_fa = str(tmp_path / 'ref.fa'), str(tmp_path / 'tig.fa')
    synth.write_fasta(ref_fa, hap.ref.names, hap.ref.seqs, line=80)
    synth.write_fasta(tig_fa, hap.tig_names, hap.tig_seqs, line=70)
    names = hap.ref.names
    ctx.seq_load(_lib.PAV_ROLE_REF, names, [hap.ref.seqs[n] for n in names])
    ctx.seq_load(_lib.PAV_ROLE_TIG, hap.tig_names, [hap.tig_seqs[n] for n in hap.tig_names])
    want = cigarcall.call_records(ctx, hap.df_align)
    errs = []

    def load(role, path):
        try:
            ctx.seq_load_fasta_path(role, path)
        except BaseException as ex:                            # noqa: BLE001
            errs.append(ex)
    ths = [threading.Thread(target=load, args=a) for a in ((_lib.PAV_ROLE_REF, ref_fa), (_lib.PAV_ROLE_TIG, tig_fa))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    assert ctx.seq_names(_lib.PAV_ROLE_REF) == list(names) and ctx.seq_names(_lib.PAV_ROLE_TIG) == list(hap.tig_names)
    got = cigarcall.call_records(ctx, hap.df_align)
    for a, b in zip(want[:3], got[:3]):
        assert a.tobytes() == b.tobytes()
    n = hap.tig_names[3]
    assert ctx.seq_fetch(_lib.PAV_ROLE_TIG, 3, 0, hap.tig_seqs[n].shape[0]).tobytes() == hap.tig_seqs[n].tobytes()
