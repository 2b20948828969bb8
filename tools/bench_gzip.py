#!/usr/bin/env python3
"""The device gzip alone (pav_amd/csrc/deflate.hip through pav_gzip_buffer) on table-shaped text: GB/s of text and size against
zlib at levels 1 / 6, for the window sizes and search depths given.  The text is made by the product's own writers from a
synthetic haplotype (SNV table, density-table-like rows), so the byte statistics are the real ones.
    python tools/bench_gzip.py [--mb 200]"""
import argparse
import json
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class StderrCapture:
    """The library's PAV_TIMING lines (C stdio on fd 2) of the calls made inside the block."""

    def __enter__(self):
        import tempfile
        sys.stderr.flush()
        self.tmp = tempfile.TemporaryFile()
        self.saved = os.dup(2)
        os.dup2(self.tmp.fileno(), 2)
        return self

    def __exit__(self, *exc):
        os.dup2(self.saved, 2)
        os.close(self.saved)
        self.tmp.seek(0)
        self.text = self.tmp.read().decode(errors='replace')
        self.tmp.close()
        return False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mb', type=int, default=200)
    ap.add_argument('--variants', default='12:16,13:16,12:8,12:4,12:32')
    args = ap.parse_args()
    import numpy as np
    import torch  # noqa: F401
    import __graft_entry__ as g
    g.build_cpu_side()
    from pav_amd import _lib
    rng = np.random.default_rng(1)
    n = args.mb * 1_000_000 // 100
    pos = np.sort(rng.integers(10_000, 240_000_000, n))
    chrom = np.sort(rng.integers(1, 23, n))
    q = pos + rng.integers(-50000, 50000, n)
    ra = rng.integers(0, 8, (n, 2))
    L = 'ACGTacgt'
    t0 = time.time()
    snv = ''.join(f'chr{c}\t{p}\t{p + 1}\tchr{c}-{p + 1}-SNV-{L[r].upper()}{L[a].upper()}\tSNV\t1\t{L[r]}\t{L[a]}\th1\ttig{c:07d}:{x}-{x}\t{"+" if p & 1 else "-"}\t0\t{c * 41 + (p >> 22)}\tCIGAR\tPASS\n'
                  for c, p, x, r, a in zip(chrom.tolist(), pos.tolist(), q.tolist(), ra[:, 0].tolist(), ra[:, 1].tolist())).encode()
    k = np.exp(-rng.random((n, 2)) * 40)
    k[rng.random(n) < 0.3, 1] = 0.0
    kmer = rng.integers(0, 2 ** 62, n)
    den = ''.join(f'{i * 2}\t{1 if i % 9 == 0 else 0}\t{0 if (i // 5000) % 2 else 2}\t{a!r}\t0.0\t{b!r}\t{m}\t\t\n'
                  for i, (a, b, m) in enumerate(zip(k[:, 0].tolist(), k[:, 1].tolist(), kmer.tolist()))).encode()
    print(f'[bench_gzip] texts made in {time.time() - t0:.1f} s: snv {len(snv) / 1e6:.1f} MB, density {len(den) / 1e6:.1f} MB', file=sys.stderr)
    out = {}
    os.environ['PAV_TIMING'] = '1'
    with _lib.Context(0) as ctx:
        for name, text in (('snv', snv), ('density', den)):
            sample = text[:20_000_000]
            z6, z1 = len(zlib.compress(sample, 6)) / len(sample), len(zlib.compress(sample, 1)) / len(sample)
            res = {'text_mb': round(len(text) / 1e6, 1), 'zlib6_ratio': round(z6, 4), 'zlib1_ratio': round(z1, 4), 'variants': {}}
            buf = np.frombuffer(text, dtype=np.uint8)
            for v in args.variants.split(','):
                wb, ch = v.split(':')
                os.environ['PAV_GZ_WBITS'], os.environ['PAV_GZ_CHAIN'] = wb, ch
                ctx.gzip_buffer(buf[:1_000_000], 6)
                best, size, kern = 1e9, 0, 1e9
                for _ in range(3):
                    with StderrCapture() as cap:
                        t0 = time.perf_counter()
                        gz = ctx.gzip_buffer(buf, 6)
                        best = min(best, time.perf_counter() - t0)
                    size = len(gz)
                    for ln in cap.text.splitlines():                     # "[pav timing] gz_files: ...; deflate + crc 21.7 ms (10.0 GB/s), ..."
                        if 'deflate + crc' in ln:
                            kern = min(kern, float(ln.split('deflate + crc')[1].split('ms')[0]))
                assert zlib.decompress(gz, 31) == text
                res['variants'][v] = {'deflate_plus_crc_ms': kern, 'gb_per_s': round(len(text) / (kern * 1e-3) / 1e9, 2),
                                      'ms_incl_pcie_from_pageable_memory': round(best * 1e3, 1),
                                      'ratio': round(size / len(text), 4), 'vs_zlib6': round(size / len(text) / z6, 3)}
            out[name] = res
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
