#!/usr/bin/env python3
"""Inversion-scan throughput probe (BASELINE config 3 shape, one haplotype on one GPU): every flagged region of a
synthetic haplotype through pav_amd.inv.scan_for_inv_batch; prints per-kernel HIP-event times.  Development tool;
bench.py --workload cigar+inv reports the same path on the driver's JSON line."""
import argparse
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scale', type=float, default=0.1)
    ap.add_argument('--seed', type=int, default=1003)
    ap.add_argument('--max-regions', type=int, default=0)
    ap.add_argument('--repeat', type=int, default=1)
    args = ap.parse_args()
    import numpy as np
    import pandas as pd
    from pav_amd import _lib, inv as pavinv, seq as pavseq, synth, density as pavden
    from pav_amd.align import AlignLift
    from pav_amd.kmer import KmerUtil

    hap = synth.config2(seed=args.seed, scale=args.scale, threads=8)
    names = hap.ref.names
    print('stats', hap.stats, flush=True)
    ctx = _lib.Context(0)
    ctx.seq_load(_lib.PAV_ROLE_REF, names, [hap.ref.seqs[n] for n in names])
    ctx.seq_load(_lib.PAV_ROLE_TIG, hap.tig_names, [hap.tig_seqs[n] for n in hap.tig_names])
    ctx._inv_loaded = ('mem', 'mem')
    lift = AlignLift(hap.df_trim, hap.tig_lengths)
    # scan_for_inv reads "<ref>.fai": provide one
    import tempfile
    d = tempfile.mkdtemp()
    with open(os.path.join(d, 'ref.fa.fai'), 'w') as fh:
        for n in names:
            fh.write(f'{n}\t{hap.ref.seqs[n].shape[0]}\t0\t0\t0\n')
    regions = [pavseq.Region(r['#CHROM'], r['POS'], r['END']) for _, r in hap.df_flag.iterrows()]
    if args.max_regions:
        regions = regions[:args.max_regions]
    k_util = KmerUtil(31)

    class FakeCtx:
        pass
    orig = pavinv.ensure_sequences
    pavinv.ensure_sequences = lambda *a, **k: None
    for rep in range(args.repeat):
        ctx.prof_reset()
        ctx.prof_enable(True)
        logs = [io.StringIO() for _ in regions]
        t0 = time.perf_counter()
        out = pavinv.scan_for_inv_batch(regions, os.path.join(d, 'ref.fa'), 'tig.fa', lift, k_util, logs=logs, ctx=ctx)
        dt = time.perf_counter() - t0
        prof = ctx.prof_read()
        ctx.prof_enable(False)
        calls = [o for o in out if o is not None and not isinstance(o, RuntimeError)]
        errs = [o for o in out if isinstance(o, RuntimeError)]
        scanned = 0
        iters = 0
        for lg in logs:
            for line in lg.getvalue().splitlines():
                if line.startswith('Scanning region: '):
                    r = pavseq.region_from_string(line.split(': ')[1])
                    scanned += len(r)
                    iters += 1
        kms = sum(v[1] for v in prof.values())
        print(json.dumps({'regions': len(regions), 'calls': len(calls), 'errors': len(errs), 'planted': hap.stats['n_inv'],
                          'iterations': iters, 'scanned_bp': scanned, 'wall_s': round(dt, 3), 'kernel_ms': round(kms, 2),
                          'scanned_Mbp_per_s_wall': round(scanned / dt / 1e6, 2),
                          'kernels': {k: [v[0], round(v[1], 3)] for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])}}), flush=True)
    ids = sorted(c.id for c in calls)
    print('calls:', ids[:8], '...')


if __name__ == '__main__':
    main()
