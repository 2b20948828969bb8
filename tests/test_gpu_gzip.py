"""gzip on the device (pav_amd/csrc/deflate.hip) through the C ABI (pav_gzip_buffer): whatever the text, zlib must inflate the
member back to it - every byte, CRC-32 and ISIZE checked by zlib itself - and on table-shaped text the output must stay close to
what zlib's level 6 makes of it (the writers' files are compared with the reference's after gunzip; here the container format and
the encoder are on trial).  The serial parts of the encoder (trees, headers, checksum algebra) run on the host against zlib in
tests/test_host_deflate.py."""
import gzip
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    from pav_amd import _lib
    with _lib.Context(0) as c:
        yield c


def table_text(rng, rows, kind):
    if kind == 'snv':
        pos = np.sort(rng.integers(10_000, 240_000_000, rows))
        qp = pos + rng.integers(-5000, 5000)
        out = []
        for p, q in zip(pos.tolist(), qp.tolist()):
            r, a = rng.choice(list('ACGTacgt'), 2)
            out.append(f'chr7\t{p}\t{p + 1}\tchr7-{p + 1}-SNV-{r.upper()}{a.upper()}\tSNV\t1\t{r}\t{a}\th1\ttig0000{p % 9}:{q}-{q}\t+\t0\t{p % 31}\tCIGAR\tPASS\n')
        return ''.join(out).encode()
    if kind == 'density':
        out = ['INDEX\tSTATE_MER\tSTATE\tKERN_FWD\tKERN_FWDREV\tKERN_REV\tKMER\tFLANK\tMATCH\n']
        k = np.exp(-rng.random((rows, 2)) * 30)
        kmer = rng.integers(0, 2 ** 62, rows)
        for i in range(rows):
            out.append(f'{i * 3}\t{i % 3 - 1 if i % 7 else 0}\t{i % 3}\t{k[i, 0]!r}\t0.0\t{k[i, 1]!r}\t{int(kmer[i])}\t{"UP" if i % 50 == 0 else ""}\t{"SAME" if i % 50 == 0 else ""}\n')
        return ''.join(out).encode()
    raise KeyError(kind)


def check(ctx, data, level=0):
    gz = ctx.gzip_buffer(data, level)
    assert gz[:4] == b'\x1f\x8b\x08\x00'
    back = zlib.decompress(gz, 15 + 16)                       # the gzip container: zlib verifies CRC-32 and ISIZE
    assert back == bytes(data), f'{len(data)} bytes in, {len(back)} bytes back'
    assert gzip.decompress(gz) == bytes(data)
    return gz


def test_edge_sizes_round_trip(ctx):
    rng = np.random.default_rng(1)
    base = table_text(rng, 3000, 'snv')
    for n in (0, 1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 257, 4095, 4096, 4097, 65535, 65536, 65537, 131072, 131073, 200001):
        check(ctx, base[:n])
    for n in (1, 5, 64, 65536, 70000):
        check(ctx, b'\0' * n)
        check(ctx, b'A' * n)                                   # one distance code, matches of 258
        check(ctx, bytes(range(256)) * (n // 256 + 1))


def test_every_kind_of_text_round_trips(ctx):
    rng = np.random.default_rng(2)
    texts = {
        'random bytes': rng.integers(0, 256, 300_000, dtype=np.uint8).tobytes(),          # incompressible: long codes everywhere
        'dna': rng.choice(np.frombuffer(b'ACGT', dtype=np.uint8), 400_000).tobytes(),
        'digits': rng.choice(np.frombuffer(b'0123456789', dtype=np.uint8), 250_000).tobytes(),
        'skewed': np.where(rng.random(300_000) < 0.999, 97, rng.integers(0, 256, 300_000)).astype(np.uint8).tobytes(),
        'long runs': b''.join(bytes([c]) * int(n) for c, n in zip(rng.integers(65, 70, 400).tolist(), rng.integers(1, 3000, 400).tolist())),
        'snv rows': table_text(rng, 20_000, 'snv'),
        'density rows': table_text(rng, 20_000, 'density'),
        'tandem': (b'ACGTTGCA' * 5 + b'\n') * 9000,
    }
    for name, t in texts.items():
        for level in (1, 6, 9):
            gz = check(ctx, t, level)
            assert len(gz) <= len(t) + len(t) // 8 + 1024, name


def test_size_against_zlib_level_6_on_table_text(ctx):
    rng = np.random.default_rng(3)
    report = {}
    for kind, rows in (('snv', 60_000), ('density', 50_000)):
        t = table_text(rng, rows, kind)
        ours = len(check(ctx, t, 6))
        ref6 = len(zlib.compress(t, 6))
        ref1 = len(zlib.compress(t, 1))
        report[kind] = (len(t), ours, ref6, ref1)
        print(f'{kind}: text {len(t)}  device gzip {ours}  zlib-6 {ref6}  zlib-1 {ref1}  ratio to zlib-6 {ours / ref6:.3f}')
        assert ours <= 1.10 * ref6, report


def test_many_segments_in_one_member(ctx):
    rng = np.random.default_rng(4)
    t = table_text(rng, 30_000, 'snv') * 12                    # ~35 MB: several hundred segments, every wave busy
    gz = check(ctx, t, 6)
    assert len(gz) < len(t) // 3


def test_table_files_come_from_the_device_writer_and_equal_the_host_writer(tmp_path, monkeypatch):
    """The two writers of the merged SNV / INS-DEL tables on one seeded haplotype: device (text + gzip in HBM; the default) and host
    (PAV_WRITER=host: threads + zlib).  Same text after gunzip; the device's files are ONE gzip member each (OS byte 255 in the
    header, where zlib writes 3) and stay within 1.10 x of the host writer's level-6 size."""
    import __graft_entry__ as g
    g.build_cpu_side()
    from pav_amd import _lib, cigarcall, synth
    hap = synth.config2(seed=77, scale=0.02, threads=4)
    names = hap.ref.names
    with _lib.Context(0) as c:
        c.seq_load(_lib.PAV_ROLE_REF, names, [hap.ref.seqs[n] for n in names])
        c.seq_load(_lib.PAV_ROLE_TIG, hap.tig_names, [hap.tig_seqs[n] for n in hap.tig_names])
        cigarcall.call_records(c, hap.df_align)
        index = hap.df_align['INDEX'].to_numpy(dtype='int64')
        trim = hap.df_trim[['POS', 'END', 'INDEX']].set_index('INDEX').astype(int).reindex(list(index), fill_value=-1)
        tp, te = trim['POS'].to_numpy(dtype='int64'), trim['END'].to_numpy(dtype='int64')
        batch = (index % 10).astype('int64')
        out = {}
        for mode in ('device', 'host'):
            monkeypatch.setenv('PAV_WRITER', mode)
            p_snv, p_ins = str(tmp_path / f'snv_{mode}.tsv.gz'), str(tmp_path / f'insdel_{mode}.tsv.gz')
            n = c.cigar_write_tables('h1', index, tp, te, snv_path=p_snv, insdel_path=p_ins, call_batch=batch, gzip_level=6)
            out[mode] = (open(p_snv, 'rb').read(), open(p_ins, 'rb').read(), n)
    assert out['device'][2] == out['host'][2] and out['device'][2][0] > 10000
    for k in (0, 1):
        dev, host = out['device'][k], out['host'][k]
        assert dev[9] == 255 and host[9] == 3
        assert gzip.decompress(dev) == gzip.decompress(host)
        assert zlib.decompress(dev, 15 + 16) == gzip.decompress(host)         # one member holds the whole table
        assert len(dev) <= 1.10 * len(host), (k, len(dev), len(host))
