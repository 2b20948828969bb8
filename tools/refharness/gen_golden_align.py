#!/usr/bin/env python3
"""
Golden vectors for the alignment ingest: the reference's get_align_bed
(pavlib/align/align.py:666-794) run through its own rule body (rule align_get_read_bed, rules/align.snakefile:101-173) on
seeded SAM files.  pysam is not in this image: the reference code runs on the SAM-text stand-in of tools/refharness/shims/pysam.py
(record semantics restated from the SAM specification / htslib), so these vectors pin the reference's own logic - clipping,
coordinates, INDEX, sort order, check_record, CALL_BATCH and the file text - on top of those semantics.

  tests/golden/align_ingest/  <case>.sam.gz  tig.fa.fai     inputs (SAM of the alignments of a seeded haplotype; soft and
                                                           hard clips, supplementary and unmapped records, tags, @ lines)
                              <case>.bed.gz  <case>.headers.gz   what the rule writes
                              errors.json                   error cases: SAM text -> exception type and message
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
from pav_amd.align.cigar import tokenize  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden', 'align_ingest')
RULES = os.path.join(refenv.REFERENCE, 'rules')
COMP = bytes.maketrans(b'ACGTacgtNn', b'TGCAtgcaNn')


def regz(src, dst):
    with gzip.open(src, 'rb') as fh:
        data = fh.read()
    with open(dst, 'wb') as raw, gzip.GzipFile(fileobj=raw, mode='wb', mtime=0, filename='') as out:
        out.write(data)


def sam_lines(hap, rng, with_seq=True):
    """SAM records of the haplotype's alignment rows, as an aligner would write them: the table's hard clips become soft clips
    (SEQ = whole contig), stay hard (SEQ = aligned part, supplementary flag) or are split into H + S; reverse rows carry the
    reverse complement; a few unmapped / secondary records without CIGAR are mixed in."""
    lines = ['@HD\tVN:1.6\tSO:unsorted', '@PG\tID:synth\tPN:pav_amd.synth']
    for name in hap.ref.names:
        lines.insert(len(lines) - 1, f'@SQ\tSN:{name}\tLN:{hap.ref.seqs[name].shape[0]}')
    seen = set()
    rows = list(hap.df_align.sample(frac=1.0, random_state=int(rng.integers(1 << 30))).iterrows())   # file order != table order
    for k, (_, r) in enumerate(rows):
        lens, ops = tokenize(r['CIGAR'])
        ops = ops.tobytes().decode()
        toks = list(zip(lens.tolist(), ops))
        tig = hap.tig_seqs[r['QRY_ID']].tobytes()
        seq_all = tig[::-1].translate(COMP) if r['REV'] else tig
        lead = toks[0][0] if toks[0][1] == 'H' else 0
        trail = toks[-1][0] if toks[-1][1] == 'H' else 0
        core = [t for t in toks if t[1] != 'H']
        flag = 16 if r['REV'] else 0
        mode = ('soft', 'hard', 'mixed')[k % 3] if r['QRY_ID'] not in seen else ('hard', 'mixed')[k % 2]
        seen.add(r['QRY_ID'])
        if mode == 'soft':
            cig = ([(lead, 'S')] if lead else []) + core + ([(trail, 'S')] if trail else [])
            seq = seq_all
        elif mode == 'hard':
            flag |= 2048
            cig = ([(lead, 'H')] if lead else []) + core + ([(trail, 'H')] if trail else [])
            seq = seq_all[lead:len(seq_all) - trail]
        else:                                                   # H then S on the left, S then H on the right
            flag |= 2048
            hl, hr = lead // 2, trail // 2
            cig = ([(hl, 'H')] if hl else []) + ([(lead - hl, 'S')] if lead - hl else []) + core + \
                  ([(trail - hr, 'S')] if trail - hr else []) + ([(hr, 'H')] if hr else [])
            seq = seq_all[hl:len(seq_all) - hr]
        tags = []
        if k % 3 != 1:
            tags.append('RG:Z:grp%d' % (k % 4))
        if k % 4 == 0:
            tags.append('AO:i:%d' % (k // 4))
        tags.append('NM:i:%d' % int(sum(n for n, o in core if o in 'XID')))
        lines.append('\t'.join([r['QRY_ID'], str(flag), r['#CHROM'], str(int(r['POS']) + 1), str(int(rng.integers(1, 61))),
                                ''.join('%d%s' % t for t in cig), '*', '0', '0', (seq.decode() if with_seq else '*'), '*'] + tags))
        if k % 5 == 2:                                          # records the reference drops, but counts in INDEX
            lines.append('\t'.join([r['QRY_ID'], '4', '*', '0', '0', '*', '*', '0', '0', 'ACGT', '*']))
        if k % 7 == 3:
            lines.append('\t'.join([r['QRY_ID'], '256', r['#CHROM'], '100', '0', '*', '*', '0', '0', '*', '*']))
    return lines


def run_reference_rule(sam_path, fai_path, out_dir, name, hap='h1'):
    ns = dict(pd=pd, np=np, os=os, gzip=gzip, pavlib=pavlib, svpoplib=svpoplib, wildcards=Bag(asm_name='t', hap=hap))
    bed, head = os.path.join(out_dir, name + '.bed.gz'), os.path.join(out_dir, name + '.headers.gz')
    exec_rule(os.path.join(RULES, 'align.snakefile'), 'align_get_read_bed', dict(
        ns, input=Bag(sam=sam_path, tig_fai=fai_path), output=Bag(bed=bed, align_head=head)))
    return bed, head


def main():
    os.makedirs(GOLD, exist_ok=True)
    rng = np.random.default_rng(2024)
    hap = synth.config2(seed=57, scale=0.0012, threads=2)
    fai_path = os.path.join(GOLD, 'tig.fa.fai')
    with open(fai_path, 'w') as fh:
        for tig in hap.tig_names:
            fh.write(f'{tig}\t{hap.tig_seqs[tig].shape[0]}\t0\t0\t0\n')
    tmp = tempfile.mkdtemp()
    try:
        for name, with_seq in (('hap_seq', True), ('hap_noseq', False)):
            text = '\n'.join(sam_lines(hap, rng, with_seq=with_seq)) + '\n'
            sam = os.path.join(GOLD, name + '.sam.gz')
            with open(sam, 'wb') as raw, gzip.GzipFile(fileobj=raw, mode='wb', mtime=0, filename='', compresslevel=9) as out:
                out.write(text.encode())
            bed, head = run_reference_rule(sam, fai_path, tmp, name)
            regz(bed, os.path.join(GOLD, name + '.bed.gz'))
            regz(head, os.path.join(GOLD, name + '.headers.gz'))
            df = pd.read_csv(os.path.join(GOLD, name + '.bed.gz'), sep='\t')
            print(name, 'records', text.count('\n') - text.count('\n@') - (1 if text.startswith('@') else 0), 'rows', df.shape[0],
                  'rev', int(df['REV'].sum()), 'sam bytes', os.path.getsize(sam))
        # empty SAM file
        empty = os.path.join(tmp, 'empty.sam.gz')
        open(empty, 'wb').close()
        bed, head = run_reference_rule(empty, fai_path, tmp, 'empty')
        regz(bed, os.path.join(GOLD, 'empty.bed.gz'))
        # error cases: the exception the reference raises for one offending record
        fai = svpoplib.ref.get_df_fai(fai_path)
        tig = hap.tig_names[0]
        n = int(fai[tig])
        base = [tig, '0', hap.ref.names[0], '1001', '60']
        cases = {
            'm_operation': base + ['%dM' % n, '*', '0', '0', '*', '*'],
            's_before_h': base + ['5S3H%d=' % (n - 8), '*', '0', '0', '*', '*'],
            'clip_in_the_middle': base + ['10=5S%d=' % (n - 15), '*', '0', '0', '*', '*'],
            'unknown_contig': ['nosuchtig', '0', hap.ref.names[0], '1001', '60', '100=', '*', '0', '0', '*', '*'],
            'longer_than_contig': base + ['%d=' % (n + 10), '*', '0', '0', '*', '*'],
            'n_operation': base + ['10=5N%d=' % (n - 10), '*', '0', '0', '*', '*'],
        }
        errors = {}
        for key, fields in cases.items():
            path = os.path.join(tmp, key + '.sam')
            with open(path, 'w') as fh:
                fh.write('@HD\tVN:1.6\n' + '\t'.join(fields) + '\n')
            try:
                pavlib.align.get_align_bed(path, fai, 'h1')
                errors[key] = {'sam': '\t'.join(fields), 'type': None, 'message': None}
            except Exception as ex:  # noqa: BLE001
                errors[key] = {'sam': '\t'.join(fields), 'type': type(ex).__name__, 'message': str(ex)}
            print(key, errors[key]['type'], (errors[key]['message'] or '')[:100])
        with open(os.path.join(GOLD, 'errors.json'), 'w') as fh:
            json.dump(errors, fh, indent=1)
    finally:
        shutil.rmtree(tmp)


if __name__ == '__main__':
    main()
