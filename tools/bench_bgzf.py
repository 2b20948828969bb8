#!/usr/bin/env python3
"""A bgzipped FASTA file into the sequence store: members inflated on the device (pav_amd/csrc/inflate.hip, the default of
pav_seq_load_fasta_path) against the same file inflated by host threads (PAV_FASTA_INFLATE=host) and against the plain-text file.
The text is a synthetic assembly: random bases, half of them soft-masked, ~3 % in tandem repeats, ~3 % in runs of N (hg38's
proportions), 80 bases a line, contigs of 2 - 40 Mbp - bgzipped at level 6 in members of 65 280 bytes as `bgzip` writes them.
    python tools/bench_bgzf.py [--mb 1000] [--repeat 3]"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.bench_gzip import StderrCapture  # noqa: E402


def assembly_text(path, mb, seed=3, line=80):
    import numpy as np
    rng = np.random.default_rng(seed)
    left = mb * 1_000_000
    k = 0
    with open(path, 'wb') as fh:
        while left > 0:
            n = int(min(left, rng.integers(2_000_000, 40_000_000)))
            seq = rng.choice(np.frombuffer(b'ACGT', dtype=np.uint8), n)
            # in hg38's proportions: half the bases soft-masked (interspersed repeats are not copies of anything near), ~3 % in tandem
            # repeats, ~3 % in runs of N
            for _ in range(n // 4000):
                a = int(rng.integers(0, n)); seq[a:a + int(rng.integers(100, 4000))] |= 0x20
            for _ in range(n // 20000):
                a = int(rng.integers(0, n)); m = int(rng.integers(30, 1200))
                unit = seq[a:a + int(rng.integers(1, 60))].copy()
                seq[a:a + m] = np.resize(unit, seq[a:a + m].shape[0])
            for _ in range(int(rng.integers(1, 3))):
                a = int(rng.integers(0, n)); seq[a:a + int(n * rng.uniform(0.005, 0.04))] = ord('N')
            fh.write(b'>contig_%06d\n' % k)
            body = np.full((n + line - 1) // line * (line + 1), ord('\n'), dtype=np.uint8)
            idx = np.arange(n, dtype=np.int64)
            body[idx + idx // line] = seq
            fh.write(body[:n + (n + line - 1) // line].tobytes())
            left -= n
            k += 1
    return k


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mb', type=int, default=1000)
    ap.add_argument('--repeat', type=int, default=3)
    ap.add_argument('--level', type=int, default=6)
    args = ap.parse_args()
    import torch  # noqa: F401
    import __graft_entry__ as g
    g.build_cpu_side()
    from pav_amd import _lib, synth
    from pav_amd.shard import effective_cpus
    work = tempfile.mkdtemp(prefix='pav_bgzf_')
    try:
        plain, gz = os.path.join(work, 'asm.fa'), os.path.join(work, 'asm.fa.gz')
        t0 = time.time()
        n_rec = assembly_text(plain, args.mb)
        synth.bgzip(plain, gz, level=args.level, threads=min(16, effective_cpus()))
        t_in = time.time() - t0
        out = {'text_bytes': os.path.getsize(plain), 'bgzf_bytes': os.path.getsize(gz), 'records': n_rec, 'level': args.level, 'inputs_written_s': round(t_in, 1),
               'usable_cores': effective_cpus()}
        os.environ['PAV_TIMING'] = '1'
        with _lib.Context(0) as ctx:
            for name, path, env in (('bgzf_device', gz, None), ('bgzf_host', gz, 'host'), ('plain', plain, None)):
                if env:
                    os.environ['PAV_FASTA_INFLATE'] = env
                else:
                    os.environ.pop('PAV_FASTA_INFLATE', None)
                times, lines = [], ''
                for _ in range(args.repeat):
                    with StderrCapture() as cap:
                        t0 = time.perf_counter()
                        names = ctx.seq_load_fasta_path(_lib.PAV_ROLE_TIG, path)
                        ctx.sync()
                        times.append(round(time.perf_counter() - t0, 4))
                    lines = cap.text
                assert len(names) == n_rec
                out[name] = {'load_s': times, 'text_GB_per_s': round(out['text_bytes'] / 1e9 / min(times), 2),
                             'timing': [ln.replace('[pav timing] ', '') for ln in lines.splitlines() if 'bgzf_inflate_device' in ln or 'seq_load_fasta_path' in ln or 'pav profile' in ln]}
            # the two stores of a context loaded side by side from two threads, as a haplotype's reference and contigs are
            import threading
            for name, path in (('bgzf_device_both_roles', gz), ('plain_both_roles', plain)):
                times = []
                for _ in range(args.repeat):
                    th = [threading.Thread(target=ctx.seq_load_fasta_path, args=(role, path)) for role in (_lib.PAV_ROLE_REF, _lib.PAV_ROLE_TIG)]
                    with StderrCapture():
                        t0 = time.perf_counter()
                        for t in th:
                            t.start()
                        for t in th:
                            t.join()
                        ctx.sync()
                        times.append(round(time.perf_counter() - t0, 4))
                out[name] = {'load_s': times, 'text_GB_per_s': round(2 * out['text_bytes'] / 1e9 / min(times), 2)}
        print(json.dumps(out), flush=True)
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == '__main__':
    main()
