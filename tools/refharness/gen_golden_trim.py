#!/usr/bin/env python3
"""
Golden vectors for alignment trimming: the reference's trim_alignments
(pavlib/align/trim.py:11-354) run through its own rule bodies (rules/align.snakefile:54-97) on seeded overlapping
alignment tables.

  tests/golden/trim_<case>/  align_none.tsv.gz  tig.fa.fai         inputs
                             trim_tig.tsv.gz                        rule align_trim_tig       (mode='tig')
                             trim_tigref.tsv.gz                     rule align_trim_tigref    (mode='ref' on trim_tig)
                             trim_tigref_redundant.tsv.gz           same with redundant_callset (match_tig=True)
                             trim_both.tsv.gz                       trim_alignments(mode='both') called directly
  tests/golden/trim_kat.json   trim_alignment_record / trace_cigar_to_zero / find_cut_sites known answers
"""
import gzip
import json
import os
import shutil
import sys
import tempfile

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import refenv  # noqa: E402

pavlib = refenv.import_pavlib()
import svpoplib  # noqa: E402
from run_rule import Bag, exec_rule  # noqa: E402
from pav_amd import synth  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
RULES = os.path.join(refenv.REFERENCE, 'rules')
GZ = {'method': 'gzip', 'mtime': 0}


def regz(src, dst):
    """Re-compress with a fixed mtime so that regenerating the fixture gives identical bytes."""
    with gzip.open(src, 'rb') as fh:
        data = fh.read()
    with open(dst, 'wb') as raw, gzip.GzipFile(fileobj=raw, mode='wb', mtime=0, filename='') as out:
        out.write(data)


def case(name, seed, min_trim_tig_len=1000, split_scale=None, **kw):
    out = os.path.join(GOLD, name)
    os.makedirs(out, exist_ok=True)
    if split_scale is None:
        df, fai = synth.make_overlap_table(seed, **kw)
    else:                                      # the bench haplotype in small: clean records cut into overlapping pieces
        hap = synth.config2(seed=seed, scale=split_scale, threads=2)
        df, fai = synth.split_overlaps(hap.df_align, seed, **kw), hap.tig_lengths
    df.to_csv(os.path.join(out, 'align_none.tsv.gz'), sep='\t', index=False, compression=GZ)
    fai_path = os.path.join(out, 'tig.fa.fai')
    with open(fai_path, 'w') as fh:
        for tig, n in fai.items():
            fh.write(f'{tig}\t{n}\t0\t0\t0\n')
    conf = {'min_trim_tig_len': min_trim_tig_len}

    def get_config(wildcards, key, default=None, default_none=False):
        return conf.get(key, default)

    tmp = tempfile.mkdtemp()
    try:
        ns = dict(pd=pd, np=np, pavlib=pavlib, svpoplib=svpoplib, get_config=get_config, wildcards=Bag(asm_name='t', hap='h1'))
        tig = os.path.join(tmp, 'trim_tig.bed.gz')
        exec_rule(os.path.join(RULES, 'align.snakefile'), 'align_trim_tig', dict(
            ns, input=Bag(bed=os.path.join(out, 'align_none.tsv.gz'), tig_fai=fai_path), output=Bag(bed=tig),
            params=Bag(min_trim_tig_len=min_trim_tig_len)))
        regz(tig, os.path.join(out, 'trim_tig.tsv.gz'))
        for redundant, fname in ((False, 'trim_tigref.tsv.gz'), (True, 'trim_tigref_redundant.tsv.gz')):
            ref = os.path.join(tmp, 'trim_tigref.bed.gz')
            exec_rule(os.path.join(RULES, 'align.snakefile'), 'align_trim_tigref', dict(
                ns, input=Bag(bed=tig, tig_fai=fai_path), output=Bag(bed=ref),
                params=Bag(min_trim_tig_len=min_trim_tig_len, redundant_callset=redundant)))
            regz(ref, os.path.join(out, fname))
        both = pavlib.align.trim_alignments(pd.read_csv(os.path.join(out, 'align_none.tsv.gz'), sep='\t', dtype={'#CHROM': str}),
                                            min_trim_tig_len, fai_path, mode='both')
        both.to_csv(os.path.join(out, 'trim_both.tsv.gz'), sep='\t', index=False, compression=GZ)
    finally:
        shutil.rmtree(tmp)
    n0 = df.shape[0]
    n1 = pd.read_csv(os.path.join(out, 'trim_tig.tsv.gz'), sep='\t').shape[0]
    n2 = pd.read_csv(os.path.join(out, 'trim_tigref.tsv.gz'), sep='\t')
    n3 = pd.read_csv(os.path.join(out, 'trim_tigref_redundant.tsv.gz'), sep='\t').shape[0]
    print(name, 'rows', n0, '-> tig', n1, '-> tigref', n2.shape[0], '(redundant', n3, ') both', both.shape[0],
          'trimmed records', int(((n2[['TRIM_REF_L', 'TRIM_REF_R', 'TRIM_QRY_L', 'TRIM_QRY_R']] > 0).any(axis=1)).sum()))


def kat():
    """Pair-level known answers: trim_alignment_record on hand-made and random pairs, incl. the errors it raises."""
    rng = np.random.default_rng(11)
    items = []

    def rec(chrom, pos, qid, qpos, qlen, rev, ops, index):
        ref_bp = sum(n for n, o in ops if o in '=XD')
        qry_bp = sum(n for n, o in ops if o in '=XI')
        lead, trail = (qlen - qpos - qry_bp, qpos) if rev else (qpos, qlen - qpos - qry_bp)
        cigar = ('%dH' % lead if lead else '') + ''.join('%d%s' % o for o in ops) + ('%dH' % trail if trail else '')
        return pd.Series({'#CHROM': chrom, 'POS': pos, 'END': pos + ref_bp, 'INDEX': index, 'QRY_ID': qid, 'QRY_POS': qpos,
                          'QRY_END': qpos + qry_bp, 'QRY_LEN': qlen, 'REV': rev, 'CIGAR': cigar, 'TRIM_REF_L': 0, 'TRIM_REF_R': 0,
                          'TRIM_QRY_L': 0, 'TRIM_QRY_R': 0})

    def run(record_l, record_r, match_coord, rev_l, rev_r):
        item = {'l': json.loads(record_l.to_json()), 'r': json.loads(record_r.to_json()), 'match_coord': match_coord,
                'rev_l': bool(rev_l), 'rev_r': bool(rev_r)}
        try:
            a, b = pavlib.align.trim_alignment_record(record_l, record_r, match_coord, rev_l=rev_l, rev_r=rev_r)
            item['out_l'], item['out_r'] = json.loads(a.to_json()), json.loads(b.to_json())
        except Exception as ex:  # noqa: BLE001
            item['error'] = [type(ex).__name__, str(ex)]
        items.append(item)

    for _ in range(60):
        qlen = 30_000
        cut = int(rng.integers(8_000, 20_000))
        ov = int(rng.integers(1, 3_000))
        ops_l, _ = synth._random_cigar_ops(rng, cut + ov - 100, 4e-3, 2e-3, 40, rng.random() < 0.3)
        ops_r, _ = synth._random_cigar_ops(rng, qlen - 100 - cut, 4e-3, 2e-3, 40, rng.random() < 0.3)
        rev = bool(rng.integers(0, 2))
        if rev:
            ops_l, ops_r = ops_l[::-1], ops_r[::-1]
        l = rec('chr1', 100_000, 'tigA', 100, qlen, rev, ops_l, 0)
        r = rec('chr1' if rng.random() < 0.7 else 'chr2', 100_000 + cut + int(rng.integers(-2000, 2000)), 'tigA', cut, qlen, rev, ops_r, 1)
        run(l, r, 'query', rev_l=not rev, rev_r=rev)
        run(r, l, 'query', rev_l=rev, rev_r=not rev)
    for _ in range(40):
        ops_l, ref_l = synth._random_cigar_ops(rng, int(rng.integers(5_000, 12_000)), 4e-3, 2e-3, 40, rng.random() < 0.3)
        ops_r, _ = synth._random_cigar_ops(rng, int(rng.integers(5_000, 12_000)), 4e-3, 2e-3, 40, rng.random() < 0.3)
        ov = int(rng.integers(1, min(3_000, ref_l - 1)))
        l = rec('chr3', 50_000, 'tigB', 500, 20_000, bool(rng.integers(0, 2)), ops_l, 2)
        r = rec('chr3', 50_000 + ref_l - ov, 'tigC', 700, 20_000, bool(rng.integers(0, 2)), ops_r, 3)
        run(l, r, 'subject', rev_l=True, rev_r=False)
    # errors
    l = rec('chr1', 1000, 'tigD', 0, 5000, False, [(2000, '=')], 4)
    r = rec('chr1', 9000, 'tigD', 2500, 5000, False, [(2000, '=')], 5)
    run(l, r, 'query', True, False)                       # negative distance
    run(r, l, 'subject', True, False)                     # incorrectly ordered
    run(l, r, 'nonsense', True, False)
    l = rec('chr1', 1000, 'tigE', 0, 5000, False, [(1500, '='), (10, 'M'), (490, '=')], 6)
    r = rec('chr1', 4000, 'tigE', 1400, 5000, False, [(3000, '=')], 7)
    run(l, r, 'query', True, False)                       # illegal op inside the trace
    l = rec('chr1', 1000, 'tigF', 0, 5000, False, [(30, 'I'), (40, 'D')], 8)
    r = rec('chr1', 1020, 'tigF', 10, 5000, False, [(2000, '=')], 9)
    run(l, r, 'query', True, False)                       # no cut site
    with open(os.path.join(GOLD, 'trim_kat.json'), 'w') as fh:
        json.dump(items, fh, separators=(',', ':'))
    print('trim_kat', len(items), 'pairs,', sum(1 for i in items if 'error' in i), 'errors')


def main():
    case('trim_overlap', 21)
    case('trim_dense', 22, n_tigs=25, max_overlap=9_000, snv_rate=6e-3, indel_rate=4e-3, short_frac=0.15, min_trim_tig_len=1500)
    case('trim_split', 1002, split_scale=0.004)
    kat()


if __name__ == '__main__':
    main()
