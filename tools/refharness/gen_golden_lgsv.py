#!/usr/bin/env python3
"""
Golden vectors for the large-SV caller: rules call_lg_split and call_lg_discover
(rules/call_lg.snakefile:40-139) executed through tools/refharness/run_rule.py, i.e. pavlib.lgsv.scan_for_events
(pavlib/lgsv.py:31-642) on a seeded haplotype whose alignments are truncated by large insertions, deletions and inversions.

  tests/golden/lgsv_hap/  ref.fa tig.fa(.fai) align.tsv.gz n_gap.tsv        inputs
                          batch.tsv                                          rule call_lg_split
                          sv_{ins,del,inv}_{batch}.tsv  lg_sv_{batch}.log     rule call_lg_discover, per batch
                          density_tables.json                                names + sha256 of the density tables written
"""
import collections
import gzip
import hashlib
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
import intervaltree  # noqa: E402
import svpoplib  # noqa: E402
from run_rule import Bag, exec_rule  # noqa: E402
from pav_amd import synth  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
RULES = os.path.join(refenv.REFERENCE, 'rules')
BATCHES = 2


def gunzip_to(path, out):
    with gzip.open(path, 'rt') as fh, open(out, 'w') as oh:
        oh.write(fh.read())


def main():
    out = os.path.join(GOLD, 'lgsv_hap')
    os.makedirs(out, exist_ok=True)
    seed = 131
    ref = synth.make_reference(seed, {'chr1': 160_000, 'chr2': 120_000, 'chr10': 90_000}, n_every=0, inv_every=0, threads=1)
    for c, p, e, rep in [('chr1', 30_000, 38_000, 600), ('chr1', 110_000, 114_000, 0), ('chr2', 50_000, 62_000, 0),
                         ('chr10', 20_000, 26_500, 400), ('chr2', 90_000, 93_000, 0)]:
        s = ref.seqs[c]
        if rep:
            s[e - rep:e] = synth.revcomp(s[p:p + rep])
        ref.inversions.append(synth.Inversion(c, p, e, rep))
    for c, p, n in [('chr1', 70_000, 3_000), ('chr10', 60_000, 1_500)]:       # N gaps (data/ref/n_gap.bed.gz)
        ref.seqs[c][p:p + n] = ord('N')
    hap = synth.make_haplotype(ref, seed * 64, 'h1', seg_median=70_000, seg_sigma=0.4, rev_frac=0.5, threads=1, snv_rate=1.5e-3,
                               indel_rate=6e-4, pareto_alpha=0.55, max_indel=3000, decoys_per_inv=0, zone_factor=1, zone_pad=5_000)
    df = synth.make_truncating_table(hap, seed)
    pairs = collections.Counter(df[['#CHROM', 'QRY_ID']].apply(tuple, axis=1))
    print('rows', hap.df_align.shape[0], '->', df.shape[0], '; (chrom, tig) pairs with several records:', sum(1 for v in pairs.values() if v > 1))
    synth.write_fasta(os.path.join(out, 'ref.fa'), ref.names, ref.seqs, line=100)
    synth.write_fasta(os.path.join(out, 'tig.fa'), hap.tig_names, hap.tig_seqs, line=100)
    df.to_csv(os.path.join(out, 'align.tsv.gz'), sep='\t', index=False, compression={'method': 'gzip', 'mtime': 0})
    pd.DataFrame([('chr1', 70_000, 73_000), ('chr10', 60_000, 61_500)], columns=['#CHROM', 'POS', 'END']).to_csv(
        os.path.join(out, 'n_gap.tsv'), sep='\t', index=False)

    conf = {'lg_batch_count': BATCHES}

    def get_config(wildcards, key, default=None, default_none=False):
        return conf.get(key, default)

    tmp = tempfile.mkdtemp()
    cwd = os.getcwd()
    os.chdir(tmp)                                              # the rule writes density tables under results/...
    try:
        ns = dict(pd=pd, np=np, os=os, collections=collections, intervaltree=intervaltree, pavlib=pavlib, svpoplib=svpoplib,
                  get_config=get_config, REF_FA=os.path.join(out, 'ref.fa'))
        batch_tsv = os.path.join(tmp, 'batch.tsv.gz')
        exec_rule(os.path.join(RULES, 'call_lg.snakefile'), 'call_lg_split', dict(
            ns, wildcards=Bag(asm_name='t', hap='h1'), input=Bag(bed=os.path.join(out, 'align.tsv.gz')), output=Bag(tsv=batch_tsv),
            params=Bag(batch_count=BATCHES)))
        gunzip_to(batch_tsv, os.path.join(out, 'batch.tsv'))
        for batch in range(BATCHES):
            o = Bag(bed_ins=os.path.join(tmp, f'ins_{batch}.bed.gz'), bed_del=os.path.join(tmp, f'del_{batch}.bed.gz'),
                    bed_inv=os.path.join(tmp, f'inv_{batch}.bed.gz'))
            log = os.path.join(out, f'lg_sv_{batch}.log')
            exec_rule(os.path.join(RULES, 'call_lg.snakefile'), 'call_lg_discover', dict(
                ns, wildcards=Bag(asm_name='t', hap='h1', batch=str(batch)), threads=1, log=Bag(log=log),
                input=Bag(bed=os.path.join(out, 'align.tsv.gz'), tsv_group=batch_tsv, fa=os.path.join(out, 'tig.fa'),
                          fai=os.path.join(out, 'tig.fa.fai'), bed_n=os.path.join(out, 'n_gap.tsv')),
                output=o, params=Bag(k_size=31, inv_region_limit=None)))
            for name in ('ins', 'del', 'inv'):
                gunzip_to(o['bed_' + name], os.path.join(out, f'sv_{name}_{batch}.tsv'))
                n = pd.read_csv(os.path.join(out, f'sv_{name}_{batch}.tsv'), sep='\t').shape[0]
                print(f'batch {batch} {name}: {n} rows')
        dens = {}
        ddir = os.path.join(tmp, 'results', 't', 'inv_caller', 'density_table_lg')
        for f in sorted(os.listdir(ddir)):
            with gzip.open(os.path.join(ddir, f), 'rb') as fh:
                dens[f] = hashlib.sha256(fh.read()).hexdigest()
        with open(os.path.join(out, 'density_tables.json'), 'w') as fh:
            json.dump(dens, fh, indent=1)
        print('density tables', len(dens))
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp)
    for batch in range(BATCHES):
        print(open(os.path.join(out, f'lg_sv_{batch}.log')).read()[:600])


if __name__ == '__main__':
    main()
