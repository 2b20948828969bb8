#!/usr/bin/env python3
"""
Golden vectors for LARGE regions of the inversion scan: the reference itself (pavlib.inv.scan_for_inv ->
scripts/density.py -> scipy gaussian_kde, unmodified) on `synth.large_inversions()` - two inversions of 150 / 200 kb
whose flagged regions sit inside them, so the scan expands two / three times and ends on regions of 337 / 687 kbp
(`pavlib/inv.py:223-351`), forward and reverse-complemented contig, an N run inside the largest region.
Build container only; takes ~30 min (the reference's KDE is O(sample points x k-mers)).

Committed (tests/golden/inv_large/): digests, not tables -
    scans.json        inputs' md5 (the test regenerates the sequences from the seed), per flagged region every scan
                      iteration (regions, row count, INDEX / STATE_MER / STATE sha1, rl_encoder runs, KERN_* sums),
                      log lines, InvCall fields, INV BED row (SEQ as sha1 + length), sha1 of KMER / FLANK / MATCH
    kern_<ID>.npz     KERN_* of a sample of table rows (every 499th + every row within 40 rows of a STATE or
                      STATE_MER change) as exact float64, with their row numbers
"""
import hashlib
import io
import json
import os
import sys
import time

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import gen_golden_inv as g  # noqa: E402  (imports the reference through refenv)

from pav_amd import synth  # noqa: E402

pavlib, kanapy, svpoplib = g.pavlib, g.kanapy, g.svpoplib


def md5(b):
    return hashlib.md5(b).hexdigest()


def sample_rows(df):
    n = df.shape[0]
    keep = np.zeros(n, dtype=bool)
    keep[::499] = True
    keep[-1] = True
    for col in ('STATE', 'STATE_MER'):
        v = df[col].to_numpy()
        for c in np.flatnonzero(v[1:] != v[:-1]) + 1:
            keep[max(0, c - 40):c + 40] = True
    return np.flatnonzero(keep)


def main():
    d = os.path.join(g.GOLD, 'inv_large')
    os.makedirs(d, exist_ok=True)
    work = os.environ.get('PAV_GOLDEN_WORK', '/tmp/pav_inv_large')
    os.makedirs(work, exist_ok=True)
    ref, hap, flags = synth.large_inversions()
    ref_fa, tig_fa = os.path.join(work, 'ref.fa'), os.path.join(work, 'tig.fa')
    synth.write_fasta(ref_fa, ref.names, ref.seqs, line=100)
    synth.write_fasta(tig_fa, hap.tig_names, hap.tig_seqs, line=100)
    align_text = hap.df_trim.to_csv(sep='\t', index=False)
    with open(os.path.join(work, 'align.tsv'), 'w') as fh:
        fh.write(align_text)
    inputs = {'ref_md5': md5(b''.join(ref.seqs[n].tobytes() for n in ref.names)),
              'tig_md5': md5(b''.join(hap.tig_seqs[n].tobytes() for n in hap.tig_names)),
              'align_tsv_md5': md5(align_text.encode())}

    k_util = kanapy.util.kmer.KmerUtil(31)
    df_aln = pd.read_csv(os.path.join(work, 'align.tsv'), sep='\t')
    align_lift = pavlib.align.AlignLift(df_aln, svpoplib.ref.get_df_fai(tig_fa + '.fai'))
    only = os.environ.get('PAV_GOLDEN_FLAGS')
    scans = []
    for fi, (c, p, e, ftype, kw) in enumerate(flags):
        if only and str(fi) not in only.split(','):
            continue
        t0 = time.time()
        log = io.StringIO()
        with g.Capture(align_lift) as cap:
            call = pavlib.inv.scan_for_inv(pavlib.seq.Region(c, p, e), ref_fa, tig_fa, align_lift, k_util, threads=8,
                                           log=log, **(kw or {}))
        rec = {'flag': {'chrom': c, 'pos': p, 'end': e, 'type': ftype}, 'kwargs': kw or {},
               'iterations': cap.iterations, 'log': log.getvalue().splitlines(), 'call': None}
        if call is not None:
            row = g.inv_bed_row(call, hap.hap, ftype, tig_fa)
            bed = {k2: (int(v) if isinstance(v, (int, np.integer)) else v) for k2, v in row.items()}
            seq = bed.pop('SEQ')
            bed['SEQ_sha1'], bed['SEQ_len'] = hashlib.sha1(seq.encode()).hexdigest(), len(seq)
            df = call.df
            rows = sample_rows(df)
            rec['call'] = {
                'id': call.id, 'svlen': int(call.svlen),
                **{nm: g.region_dict(getattr(call, nm)) for nm in (
                    'region_ref_outer', 'region_ref_inner', 'region_tig_outer', 'region_tig_inner',
                    'region_ref_discovery', 'region_tig_discovery')},
                'bed_row': bed, 'n_rows': int(df.shape[0]),
                'index_sha1': g.digest(df['INDEX'].to_numpy(dtype=np.int64)),
                'state_mer_sha1': g.digest(df['STATE_MER'].to_numpy(dtype=np.int8)),
                'state_sha1': g.digest(df['STATE'].to_numpy(dtype=np.int8)),
                'kmer_sha1': g.digest(df['KMER'].to_numpy(dtype=np.uint64)),
                'flank_counts': {str(k2): int(v) for k2, v in df['FLANK'].value_counts().items()},
                'match_counts': {str(k2): int(v) for k2, v in df['MATCH'].fillna('NA').value_counts().items()},
                'flank_sha1': hashlib.sha1('\n'.join(df['FLANK'].tolist()).encode()).hexdigest(),
                'match_sha1': hashlib.sha1('\n'.join(df['MATCH'].fillna('NA').tolist()).encode()).hexdigest(),
                'unpinned_columns': ['KMER', 'MATCH'],
            }
            np.savez_compressed(os.path.join(d, f'kern_{call.id}.npz'), rows=rows.astype(np.int64),
                                **{col: df[col].to_numpy(dtype=np.float64)[rows] for col in ('KERN_FWD', 'KERN_FWDREV', 'KERN_REV')})
        rec['reference_seconds'] = round(time.time() - t0, 1)
        scans.append(rec)
        print(f'  inv_large {c}:{p}-{e} -> {call} iterations={len(cap.iterations)} {rec["reference_seconds"]} s', flush=True)
        with open(os.path.join(d, 'scans.json' if not only else f'scans_part_{only}.json'), 'w') as fh:
            json.dump({'generator': 'pav_amd.synth.large_inversions()', 'inputs': inputs, 'scans': scans}, fh, indent=1)


if __name__ == '__main__':
    main()
