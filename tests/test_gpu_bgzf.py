"""BGZF inflated on the device (pav_amd/csrc/inflate.hip + inflate_dev.h: a lane per member decodes the Huffman codes into tokens,
a wave per member resolves the copies in an LDS window, every member's CRC-32 and ISIZE are checked) against zlib: the text of
members written at every level and strategy zlib has - FASTA-like text, runs, incompressible bytes (stored blocks), members of
every size from 0 to 64 KiB at odd places in the text - and the errors of members that are not what their footer says.  PAV reads
bgzipped FASTA files through pysam.FastaFile (pavlib/cigarcall.py:59-64; rules/call.snakefile:796); the loader that uses this
step is on trial in tests/test_gpu_fasta.py (kind 'bgzf') and below against the host inflate."""
import struct
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def member(chunk, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, mem_level=8):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, mem_level, strategy)
    body = c.compress(chunk) + c.flush()
    assert len(body) + 26 <= 65536
    return (b'\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00' + struct.pack('<H', len(body) + 25) + body
            + struct.pack('<II', zlib.crc32(chunk), len(chunk)))


def bgzf(data, sizes, **kw):
    """data cut into members of the given sizes in turn (the last size repeats), then the end-of-file member"""
    out, at, k = bytearray(), 0, 0
    while at < len(data):
        n = sizes[min(k, len(sizes) - 1)]
        out += member(data[at:at + n], **kw)
        at += n
        k += 1
    out += member(b'')
    return bytes(out)


def fasta_like(rng, n):
    seq = rng.choice(np.frombuffer(b'ACGT', dtype=np.uint8), n)
    # soft-masked stretches, runs of N, tandem repeats: what the copies of a real assembly's members look like
    for _ in range(max(1, n // 20000)):
        a = int(rng.integers(0, max(1, n - 5000)))
        kind = int(rng.integers(0, 3))
        m = int(rng.integers(50, 5000))
        if kind == 0:
            seq[a:a + m] = seq[a:a + m] | 0x20
        elif kind == 1:
            seq[a:a + m] = ord('N')
        else:
            unit = seq[a:a + int(rng.integers(1, 40))].copy()
            seq[a:a + m] = np.resize(unit, len(seq[a:a + m]))
    lines = [b'>tig1 a description\n']
    raw = seq.tobytes()
    lines += [raw[i:i + 60] + b'\n' for i in range(0, n, 60)]
    return b''.join(lines)


@pytest.fixture(scope='module')
def ctx():
    from pav_amd import _lib
    with _lib.Context(0) as c:
        yield c


@pytest.mark.parametrize('level', [1, 6, 9])
def test_fasta_text_at_bgzip_sizes(ctx, level):
    rng = np.random.default_rng(level)
    text = fasta_like(rng, 3_000_000)
    assert ctx.bgzf_inflate(bgzf(text, [65280], level=level)) == text


def test_members_of_every_shape(ctx):
    rng = np.random.default_rng(7)
    text = fasta_like(rng, 700_000)
    # sizes that put members at every alignment of the text, the largest member BGZF allows, one-byte members, an empty one in the middle
    sizes = [1, 2, 3, 15, 16, 17, 0, 255, 4097, 65536, 65535, 31, 33333, 1, 64, 65281, 12345]
    assert ctx.bgzf_inflate(bgzf(text, sizes)) == text
    for strategy in (zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
        assert ctx.bgzf_inflate(bgzf(text, [50001, 7, 65280], strategy=strategy)) == text
    assert ctx.bgzf_inflate(bgzf(text, [60000], level=4, mem_level=1)) == text          # many deflate blocks per member
    assert ctx.bgzf_inflate(bgzf(text, [65280], level=0)) == text                       # stored blocks only
    assert ctx.bgzf_inflate(member(b'')) == b''
    assert ctx.bgzf_inflate(b'') == b''


def test_runs_random_bytes_and_rare_symbols(ctx):
    rng = np.random.default_rng(11)
    noise = rng.integers(0, 256, 400_000, dtype=np.uint8).tobytes()                     # incompressible: zlib stores it
    assert ctx.bgzf_inflate(bgzf(noise, [65000])) == noise
    runs = b'N' * 300_000 + b'\n' + b'AC' * 100_000 + b'ACGTTGCA' * 30_000              # copies at distance 1, 2, 8, chained through the batches
    assert ctx.bgzf_inflate(bgzf(runs, [65280])) == runs
    skew = bytes((rng.choice(np.frombuffer(b'abc', dtype=np.uint8), 500_000) ^ (rng.random(500_000) < 0.004) * rng.integers(1, 255, 500_000)).astype(np.uint8))
    assert ctx.bgzf_inflate(bgzf(skew, [65280], level=9)) == skew                       # codes longer than the first-level tables


def test_both_resolve_kernels(ctx, monkeypatch):
    """The default resolve kernel keeps a ring of 36 KiB (history + the step's text, flushed as it is made; CRC-32 by k_bgzf_crc);
    PAV_INFLATE_WINDOW=full runs the first version (the member's whole text in LDS, CRC-32 from the window).  Same text, and both
    name a corrupt member.  Long runs cut the ring kernel's steps short (a step's text stays under 4 080 bytes) and wrap the ring."""
    from pav_amd import _lib
    rng = np.random.default_rng(29)
    text = fasta_like(rng, 900_000) + b'N' * 200_000 + b'ACGTTGCAAC' * 30_000 + rng.integers(0, 256, 100_000, dtype=np.uint8).tobytes()
    for sizes in ([65280], [65536, 1, 40000, 17]):
        data = bgzf(text, sizes)
        assert ctx.bgzf_inflate(data) == text
        bad = bytearray(data)
        bsize = struct.unpack('<H', data[16:18])[0] + 1
        bad[bsize + 18 + (struct.unpack('<H', data[bsize + 16:bsize + 18])[0] + 1 - 26) // 2] ^= 0x04       # in the middle of the second member's payload
        with pytest.raises(_lib.PavDeviceError, match='corrupt BGZF member'):
            ctx.bgzf_inflate(bytes(bad))
        monkeypatch.setenv('PAV_INFLATE_WINDOW', 'full')
        assert ctx.bgzf_inflate(data) == text
        with pytest.raises(_lib.PavDeviceError, match='corrupt BGZF member'):
            ctx.bgzf_inflate(bytes(bad))
        monkeypatch.delenv('PAV_INFLATE_WINDOW')


def test_codes_of_the_maximum_length(ctx):
    """Symbol counts that grow like Fibonacci numbers give Huffman codes of every length up to deflate's limit of 15 bits - the
    canonical walk of the long codes from the first-level table's length to the last - for literals (Huffman-only members) and,
    with copies at two dozen distances whose counts grow the same way, for the distance code."""
    fib = [1, 1]
    while len(fib) < 24:
        fib.append(fib[-1] + fib[-2])
    rng = np.random.default_rng(23)
    sym = np.concatenate([np.full(f, 33 + i, dtype=np.uint8) for i, f in enumerate(fib)])
    rng.shuffle(sym)
    text = sym.tobytes()
    assert ctx.bgzf_inflate(bgzf(text, [65000], level=9, strategy=zlib.Z_HUFFMAN_ONLY)) == text
    assert ctx.bgzf_inflate(bgzf(text, [65000], level=9)) == text
    # distances: a block of noise, then copies of 40 bytes from 24 places in it, place i used fib[i] times (scaled down)
    noise = rng.integers(0, 256, 30000, dtype=np.uint8).tobytes()
    places = rng.integers(0, 29000, 24)
    uses = np.concatenate([np.full(max(1, f // 40), i) for i, f in enumerate(fib)])
    rng.shuffle(uses)
    body = bytearray(noise)
    for i in uses[:800].tolist():
        body += noise[int(places[i]):int(places[i]) + 40] + bytes([int(rng.integers(0, 256))])
    text = bytes(body)
    assert ctx.bgzf_inflate(bgzf(text, [65280], level=9)) == text


def test_corrupt_members_are_named(ctx):
    from pav_amd import _lib
    rng = np.random.default_rng(13)
    text = fasta_like(rng, 400_000)
    good = bgzf(text, [65280])
    assert ctx.bgzf_inflate(good) == text
    bsize = struct.unpack('<H', good[16:18])[0] + 1
    second = bsize                                          # the second member starts here
    bsize2 = struct.unpack('<H', good[second + 16:second + 18])[0] + 1

    def broken(pos, xor=0x40):
        b = bytearray(good)
        b[pos] ^= xor
        return bytes(b)

    with pytest.raises(_lib.PavDeviceError, match=r'corrupt BGZF member 1 of .*CRC-32'):
        ctx.bgzf_inflate(broken(second + bsize2 - 8))       # the member's CRC-32
    with pytest.raises(_lib.PavDeviceError, match=r'corrupt BGZF member 1 of'):
        ctx.bgzf_inflate(broken(second + bsize2 - 4, 0x01))  # its ISIZE: one byte more than the text
    with pytest.raises(_lib.PavDeviceError, match=r'corrupt BGZF member 1 of'):
        ctx.bgzf_inflate(broken(second + bsize2 - 2, 0x01))  # ISIZE above 64 KiB
    hit = 0
    for k in range(40):                                     # a flipped bit in the payload: whatever the decoder makes of it, the checks catch it
        pos = second + 18 + int(rng.integers(0, bsize2 - 27))   # (not the last byte: the bits behind the end-of-block code are padding)
        with pytest.raises(_lib.PavDeviceError, match=r'corrupt BGZF member 1 of'):
            ctx.bgzf_inflate(broken(pos, 1 << int(rng.integers(0, 8))))
        hit += 1
    assert hit == 40
    with pytest.raises(_lib.PavDeviceError, match='not a series of BGZF members'):
        ctx.bgzf_inflate(good[:-5])                         # cut inside the end-of-file member
    with pytest.raises(_lib.PavDeviceError, match='not a series of BGZF members'):
        ctx.bgzf_inflate(good + b'trailing bytes')
    assert ctx.bgzf_inflate(good) == text                   # and the context goes on


def test_loader_device_inflate_equals_host_inflate(ctx, tmp_path, monkeypatch):
    """pav_seq_load_fasta_path on a bgzipped file: members inflated on the device (the default) and by host threads
    (PAV_FASTA_INFLATE=host) leave the same records in the store."""
    from pav_amd import _lib
    rng = np.random.default_rng(17)
    recs = [(f'tig{i}', rng.choice(np.frombuffer(b'ACGTacgtN', dtype=np.uint8), int(n)).tobytes()) for i, n in enumerate([5, 70001, 0, 1_300_007, 61, 250_000])]
    text = b''.join(b'>' + n.encode() + b' len=%d\n' % len(s) + b''.join(s[i:i + 70] + b'\n' for i in range(0, len(s), 70)) for n, s in recs)
    path = str(tmp_path / 'contigs_h1.fa.gz')
    with open(path, 'wb') as fh:
        fh.write(bgzf(text, [65280]))
    got = {}
    for mode in ('device', 'host'):
        if mode == 'host':
            monkeypatch.setenv('PAV_FASTA_INFLATE', 'host')
        names = ctx.seq_load_fasta_path(_lib.PAV_ROLE_TIG, path)
        assert names == [n for n, _ in recs]
        assert ctx.seq_lengths(_lib.PAV_ROLE_TIG) == [len(s) for _, s in recs]
        got[mode] = [ctx.seq_fetch(_lib.PAV_ROLE_TIG, i, 0, len(s)).tobytes() for i, (_, s) in enumerate(recs)]
        assert got[mode] == [s for _, s in recs]
    monkeypatch.delenv('PAV_FASTA_INFLATE')
    # a member that does not inflate to its footer's checksum stops the load, the file and the member named
    raw = bytearray(open(path, 'rb').read())
    raw[30000] ^= 0x10
    bad = str(tmp_path / 'bad.fa.gz')
    with open(bad, 'wb') as fh:
        fh.write(bytes(raw))
    with pytest.raises(_lib.PavDeviceError, match=r'bad\.fa\.gz: corrupt BGZF member 0 of'):
        ctx.seq_load_fasta_path(_lib.PAV_ROLE_TIG, bad)
