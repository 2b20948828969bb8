#!/usr/bin/env python3
"""
Rule-level golden vectors: the ``run:`` bodies of the reference's own rules (executed unmodified through
tools/refharness/run_rule.py) on the committed inputs of tests/golden/cigar_synth and tests/golden/inv_hap.

  tests/golden/rule_call_cigar/{snv,insdel}_merged.tsv      call_cigar x 10 batches -> call_cigar_merge
  tests/golden/rule_call_inv_batch/{inv_merged.tsv,log_*.txt,density_index.json}   call_inv_batch x 2 -> call_inv_batch_merge
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
import collections, gc, intervaltree, kanapy, svpoplib  # noqa: E402,E401
from run_rule import Bag, exec_rule  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
RULES = os.path.join(refenv.REFERENCE, 'rules')
MODS = dict(pd=pd, np=np, os=os, gc=gc, collections=collections, intervaltree=intervaltree, pavlib=pavlib, kanapy=kanapy,
            svpoplib=svpoplib)


def gunzip_text(path):
    with gzip.open(path, 'rt') as fh:
        return fh.read()


def rule_call_cigar():
    src = os.path.join(GOLD, 'cigar_synth')
    out = os.path.join(GOLD, 'rule_call_cigar')
    os.makedirs(out, exist_ok=True)
    tmp = tempfile.mkdtemp()
    ins, snv = [], []
    for batch in range(10):
        o = Bag(bed_insdel=os.path.join(tmp, f'insdel_h1_{batch}.bed.gz'), bed_snv=os.path.join(tmp, f'snv_h1_{batch}.bed.gz'))
        exec_rule(os.path.join(RULES, 'call.snakefile'), 'call_cigar', dict(
            MODS, REF_FA=os.path.join(src, 'ref.fa'), wildcards=Bag(batch=str(batch), hap='h1', asm_name='t'),
            input=Bag(bed=os.path.join(src, 'align.tsv'), bed_trim=os.path.join(src, 'trim.tsv'), tig_fa_name=os.path.join(src, 'tig.fa')),
            output=o))
        ins.append(o.bed_insdel)
        snv.append(o.bed_snv)
    o = Bag(bed_insdel=os.path.join(tmp, 'svindel_insdel_h1.bed.gz'), bed_snv=os.path.join(tmp, 'snv_snv_h1.bed.gz'))
    exec_rule(os.path.join(RULES, 'call.snakefile'), 'call_cigar_merge', dict(MODS, input=Bag(bed_insdel=ins, bed_snv=snv), output=o))
    for name, path in (('insdel_merged.tsv', o.bed_insdel), ('snv_merged.tsv', o.bed_snv)):
        with open(os.path.join(out, name), 'w') as fh:
            fh.write(gunzip_text(path))
    print('rule_call_cigar', [len(gunzip_text(p).splitlines()) for p in (o.bed_insdel, o.bed_snv)])
    shutil.rmtree(tmp)


def rule_call_inv_batch():
    src = os.path.join(GOLD, 'inv_hap')
    out = os.path.join(GOLD, 'rule_call_inv_batch')
    os.makedirs(out, exist_ok=True)
    tmp = tempfile.mkdtemp()
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        beds = []
        for batch in (0, 1):
            bed = os.path.join(tmp, f'inv_call_{batch}.bed.gz')
            logp = os.path.join(tmp, f'inv_call_{batch}.log')
            exec_rule(os.path.join(RULES, 'call_inv.snakefile'), 'call_inv_batch', dict(
                MODS, REF_FA=os.path.join(src, 'ref.fa'), get_config=lambda *a, **k: (dict() if len(a) == 1 else (a[2] if len(a) > 2 else None)),
                wildcards=Bag(asm_name='t', hap='h1', batch=str(batch)), threads=1, log=Bag(log=logp),
                input=Bag(bed_flag=os.path.join(src, 'flag.tsv'), bed_aln=os.path.join(src, 'align.tsv'), tig_fa=os.path.join(src, 'tig.fa'),
                          fai=os.path.join(src, 'tig.fa.fai')),
                output=Bag(bed=bed)))
            beds.append(bed)
            shutil.copy(logp, os.path.join(out, f'log_{batch}.txt'))
        merged = os.path.join(tmp, 'sv_inv_h1.bed.gz')
        exec_rule(os.path.join(RULES, 'call_inv.snakefile'), 'call_inv_batch_merge', dict(
            MODS, get_config=lambda *a, **k: 2, BATCH_COUNT_DEFAULT=2, wildcards=Bag(asm_name='t', hap='h1'), input=Bag(bed=beds),
            output=Bag(bed=merged)))
        with open(os.path.join(out, 'inv_merged.tsv'), 'w') as fh:
            fh.write(gunzip_text(merged))
        # density tables written by the rule: keep the integer / text columns (exact) and the float columns as float64
        index = {}
        ddir = os.path.join(tmp, 'results', 't', 'inv_caller', 'density_table')
        for f in sorted(os.listdir(ddir)):
            df = pd.read_csv(os.path.join(ddir, f), sep='\t', keep_default_na=False)
            index[f] = {'rows': int(df.shape[0]), 'columns': list(df.columns),
                        'head': gunzip_text(os.path.join(ddir, f)).splitlines()[:3]}
        with open(os.path.join(out, 'density_index.json'), 'w') as fh:
            json.dump(index, fh, indent=1)
        print('rule_call_inv_batch', len(gunzip_text(merged).splitlines()) - 1, 'calls;', sorted(index))
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp)


if __name__ == '__main__':
    rule_call_cigar()
    rule_call_inv_batch()
