#!/usr/bin/env python3
"""
Golden vectors for the inversion-flagging rules, produced by executing the reference's
own rule bodies (rules/call_inv.snakefile:321-692) through tools/refharness/run_rule.py:

  tests/golden/flag_hap/   ref.fa tig.fa align.tsv trim.tsv           inputs of the CIGAR caller
                           svindel_insdel.tsv  snv_snv.tsv            merged call_cigar tables (reference rules)
                           cluster_snv.tsv cluster_indel.tsv insdel_sv.tsv insdel_indel.tsv   the four flag tables
                           flagged_regions.tsv                        rule call_inv_merge_flagged_loci
"""
import gzip
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
import collections, gc, intervaltree, kanapy, svpoplib  # noqa: E402,E401
from run_rule import Bag, exec_rule, top_level_def  # noqa: E402
from pav_amd import synth  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
RULES = os.path.join(refenv.REFERENCE, 'rules')
MODS = dict(pd=pd, np=np, os=os, gc=gc, collections=collections, intervaltree=intervaltree, pavlib=pavlib, kanapy=kanapy,
            svpoplib=svpoplib)


def gunzip_to(path, out):
    with gzip.open(path, 'rt') as fh, open(out, 'w') as oh:
        oh.write(fh.read())


def get_config(wildcards, key=None, default=None, default_none=False):
    return default


def case(name, seed, empty=False, **hap_kw):
    out = os.path.join(GOLD, name)
    os.makedirs(out, exist_ok=True)
    ref = synth.make_reference(seed, {'chr1': 180_000, 'chr2': 120_000, 'chr10': 90_000}, n_every=0, inv_every=0, threads=1)
    for c, p, e, rep in [('chr1', 30_000, 41_000, 800), ('chr1', 120_000, 124_000, 0), ('chr2', 50_000, 68_000, 0), ('chr10', 20_000, 26_500, 500)]:
        s = ref.seqs[c]
        if rep:
            s[e - rep:e] = synth.revcomp(s[p:p + rep])
        ref.inversions.append(synth.Inversion(c, p, e, rep))
    kw = dict(seg_median=60_000, seg_sigma=0.5, rev_frac=0.4, threads=1, snv_rate=2e-3, indel_rate=6e-3, max_indel=400,
              pair_frac=0.25, decoys_per_inv=0, zone_factor=1, zone_pad=6_000)
    kw.update(hap_kw)
    hap = synth.make_haplotype(ref, seed * 64, 'h1', **kw)
    if empty:
        hap.df_align = hap.df_align.iloc[:0]
        hap.df_trim = hap.df_trim.iloc[:0]
    print('rows', hap.df_align.shape[0], hap.stats)
    synth.write_fasta(os.path.join(out, 'ref.fa'), ref.names, ref.seqs, line=100)
    synth.write_fasta(os.path.join(out, 'tig.fa'), hap.tig_names, hap.tig_seqs, line=100)
    hap.df_align.to_csv(os.path.join(out, 'align.tsv'), sep='\t', index=False)
    hap.df_trim.to_csv(os.path.join(out, 'trim.tsv'), sep='\t', index=False)

    tmp = tempfile.mkdtemp()
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        ins, snv = [], []
        for batch in range(10):
            o = Bag(bed_insdel=os.path.join(tmp, f'insdel_{batch}.bed.gz'), bed_snv=os.path.join(tmp, f'snv_{batch}.bed.gz'))
            exec_rule(os.path.join(RULES, 'call.snakefile'), 'call_cigar', dict(
                MODS, REF_FA=os.path.join(out, 'ref.fa'), wildcards=Bag(batch=str(batch), hap='h1', asm_name='t'),
                input=Bag(bed=os.path.join(out, 'align.tsv'), bed_trim=os.path.join(out, 'trim.tsv'), tig_fa_name=os.path.join(out, 'tig.fa')),
                output=o))
            ins.append(o.bed_insdel)
            snv.append(o.bed_snv)
        merged = Bag(bed_insdel=os.path.join(tmp, 'svindel_insdel_h1.bed.gz'), bed_snv=os.path.join(tmp, 'snv_snv_h1.bed.gz'))
        exec_rule(os.path.join(RULES, 'call.snakefile'), 'call_cigar_merge', dict(MODS, input=Bag(bed_insdel=ins, bed_snv=snv), output=merged))
        # fixtures: the indel table in full, the SNV table cut to the columns the flag rules read (usecols at :633)
        pd.read_csv(merged.bed_insdel, sep='\t', dtype=str, keep_default_na=False).to_csv(
            os.path.join(out, 'svindel_insdel.tsv.gz'), sep='\t', index=False, compression={'method': 'gzip', 'mtime': 0})
        pd.read_csv(merged.bed_snv, sep='\t', dtype=str, keep_default_na=False,
                    usecols=['#CHROM', 'POS', 'END', 'SVTYPE', 'SVLEN', 'FILTER', 'ALIGN_INDEX']).to_csv(
            os.path.join(out, 'snv_snv.tsv.gz'), sep='\t', index=False, compression={'method': 'gzip', 'mtime': 0})

        flag = {}
        for vartype, src in (('snv', merged.bed_snv), ('indel', merged.bed_insdel)):
            o = Bag(bed=os.path.join(tmp, f'cluster_{vartype}.bed.gz'))
            exec_rule(os.path.join(RULES, 'call_inv.snakefile'), 'call_inv_cluster', dict(
                MODS, wildcards=Bag(asm_name='t', hap='h1', vartype=vartype), input=Bag(bed=[src]), output=o,
                params=Bag(cluster_win=200, cluster_win_min=500, cluster_min_snv=20, cluster_min_indel=10)))
            flag[f'cluster_{vartype}'] = o.bed
        for vartype in ('sv', 'indel'):
            o = Bag(bed=os.path.join(tmp, f'insdel_{vartype}.bed.gz'))
            exec_rule(os.path.join(RULES, 'call_inv.snakefile'), 'call_inv_flag_insdel_cluster', dict(
                MODS, wildcards=Bag(asm_name='t', hap='h1', vartype=vartype), input=Bag(bed=merged.bed_insdel), output=o,
                params=Bag(flank_cluster=2, flank_merge=2000, cluster_min_svlen=4)))
            flag[f'insdel_{vartype}'] = o.bed
        for name, path in flag.items():
            gunzip_to(path, os.path.join(out, name + '.tsv'))
        o = Bag(bed=os.path.join(tmp, 'flagged_regions_h1.bed.gz'))
        ns = exec_rule(os.path.join(RULES, 'call_inv.snakefile'), 'call_inv_merge_flagged_loci', dict(
            MODS, get_config=get_config, BATCH_COUNT_DEFAULT=60, wildcards=Bag(asm_name='t', hap='h1'), output=o,
            _call_inv_accept_flagged_region=top_level_def(os.path.join(RULES, 'call_inv.snakefile'), '_call_inv_accept_flagged_region'),
            input=Bag(bed_insdel_sv=flag['insdel_sv'], bed_insdel_indel=flag['insdel_indel'], bed_cluster_indel=flag['cluster_indel'],
                      bed_cluster_snv=flag['cluster_snv'])))
        gunzip_to(o.bed, os.path.join(out, 'flagged_regions.tsv'))
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp)
    if empty:
        # the FASTA files of the empty case were inputs of the reference's run only - no test reads them (tests/test_gpu_flag.py
        # starts from the tables): they are not kept, so a regeneration leaves no untracked files behind
        for f in ('ref.fa', 'ref.fa.fai', 'tig.fa', 'tig.fa.fai'):
            if os.path.exists(os.path.join(out, f)):
                os.remove(os.path.join(out, f))
    for f in ('cluster_snv', 'cluster_indel', 'insdel_sv', 'insdel_indel', 'flagged_regions'):
        with open(os.path.join(out, f + '.tsv')) as fh:
            txt = fh.read()
        print(f, len(txt.splitlines()) - 1, 'rows')
    print(open(os.path.join(out, 'flagged_regions.tsv')).read()[:600])


def main():
    case('flag_hap', 91)
    case('flag_sparse', 92, snv_rate=1e-3, indel_rate=3e-4, pair_frac=0.0)          # most flag tables empty
    case('flag_empty', 93, empty=True)                                              # no alignments at all


if __name__ == '__main__':
    main()
