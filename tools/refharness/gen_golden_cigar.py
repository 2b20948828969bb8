#!/usr/bin/env python3
"""
Generate the CIGAR-call golden vectors by running the *reference itself* (pavlib, imported read-only from
/root/reference through tools/refharness/refenv.py) on small seeded inputs.

Outputs (committed): tests/golden/cigar_<case>/
    ref.fa(.fai)  tig.fa(.fai)  align.tsv  trim.tsv        inputs
    snv.tsv  insdel.tsv                                    what rule call_cigar writes (incl. FILTER)
and tests/golden/cigar_errors.json, tests/golden/kat.json (homology / tokenizer / Region known answers),
tests/golden/config1.json (BASELINE configs[0]: digests of the reference's output on the regenerable 1 Mb case).

The FILTER step is the body of rule call_cigar (rules/call.snakefile:813-842), a Snakemake ``run:`` block that
cannot be imported; it is restated below on the frames the reference function returned.

Run only in the build container:  python tools/refharness/gen_golden_cigar.py
"""

import json
import os
import sys

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import refenv  # noqa: E402

pavlib = refenv.import_pavlib()

from pav_amd import synth  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')

_COMP = str.maketrans('ACGTRYSWKMBDHVNUacgtryswkmbdhvnu', 'TGCAYRSWMKVHDBNAtgcayrswmkvhdbna')


def revcomp_str(s):
    return s.translate(_COMP)[::-1]


def write_fa(path, names, seqs):
    synth.write_fasta(path, names, {n: np.frombuffer(seqs[n].encode(), dtype=np.uint8) for n in names}, line=60)


def rule_filter(df, df_trim):
    """rules/call.snakefile:813-842 restated (adds FILTER)."""
    df = df.copy()
    df_pass = df_trim.reindex(list(df['ALIGN_INDEX']), fill_value=-1).set_index(df.index, drop=True)
    df['FILTER'] = ((df['POS'] > df_pass['POS']) & (df['END'] < df_pass['END'])).apply(
        lambda val: 'PASS' if val else 'TRIM')
    return df


def run_reference(case_dir, df_align, df_trim, hap='h1'):
    df_snv, df_insdel = pavlib.cigarcall.make_insdel_snv_calls(
        df_align, os.path.join(case_dir, 'ref.fa'), os.path.join(case_dir, 'tig.fa'), hap, version_id=False)
    trim = df_trim[['POS', 'END', 'INDEX']].set_index('INDEX').astype(int)
    df_snv = rule_filter(df_snv, trim)
    df_insdel = rule_filter(df_insdel, trim)
    df_snv.to_csv(os.path.join(case_dir, 'snv.tsv'), sep='\t', index=False)
    df_insdel.to_csv(os.path.join(case_dir, 'insdel.tsv'), sep='\t', index=False)
    return df_snv, df_insdel


ALIGN_COLS = ['#CHROM', 'POS', 'END', 'INDEX', 'QRY_ID', 'QRY_POS', 'QRY_END', 'QRY_LEN', 'RG', 'AO', 'MAPQ', 'REV',
              'FLAGS', 'HAP', 'CIGAR', 'CALL_BATCH']


# ---------------------------------------------------------------------------------------------------------
# Case 1: seeded synthetic table (two chromosomes whose names sort as strings: chr10 < chr2)
# ---------------------------------------------------------------------------------------------------------

def case_synth():
    d = os.path.join(GOLD, 'cigar_synth')
    os.makedirs(d, exist_ok=True)
    ref = synth.make_reference(4242, {'chr2': 70_000, 'chr10': 50_000}, n_every=0, inv_every=0, threads=1)
    ref.seqs['chr2'][30_000:30_400] = ord('N')
    ref.seqs['chr10'][100:130] = ord('n')
    hap = synth.make_haplotype(
        ref, 4242 * 64, 'h1', snv_rate=4e-3, indel_rate=2.5e-3, max_indel=300, threads=1,
        segments={'chr2': [(0, 33_000), (33_500, 70_000), (10_000, 24_000)], 'chr10': [(200, 25_000), (25_400, 50_000)]},
        rev_frac=0.5)
    synth.write_fasta(os.path.join(d, 'ref.fa'), ref.names, ref.seqs, line=80)
    synth.write_fasta(os.path.join(d, 'tig.fa'), hap.tig_names, hap.tig_seqs, line=80)
    hap.df_align.to_csv(os.path.join(d, 'align.tsv'), sep='\t', index=False)
    hap.df_trim.to_csv(os.path.join(d, 'trim.tsv'), sep='\t', index=False)
    snv, insdel = run_reference(d, hap.df_align, hap.df_trim)
    print('cigar_synth', snv.shape, insdel.shape, 'rev rows', int(hap.df_align['REV'].sum()),
          'max left shift', insdel['LEFT_SHIFT'].max())


# ---------------------------------------------------------------------------------------------------------
# Case 2: hand-built edge cases
# ---------------------------------------------------------------------------------------------------------

def build_tig(ref, pos, ops):
    """ops: list of (op, len_or_payload).  '=' n | 'X' alt-string | 'I' ins-string | 'D' n | 'H'/'S' string.
    Returns (oriented contig, cigar, ref_end)."""
    t, cig, p = [], [], pos
    for op, arg in ops:
        if op == '=':
            t.append(ref[p:p + arg]); cig.append(f'{arg}='); p += arg
        elif op == 'X':
            assert all(a.upper() != r.upper() for a, r in zip(arg, ref[p:p + len(arg)])), (arg, ref[p:p + len(arg)])
            t.append(arg); cig.append(f'{len(arg)}X'); p += len(arg)
        elif op == 'I':
            t.append(arg); cig.append(f'{len(arg)}I')
        elif op == 'D':
            cig.append(f'{arg}D'); p += arg
        elif op in 'HS':
            t.append(arg); cig.append(f'{len(arg)}{op}')
        else:
            raise ValueError(op)
    return ''.join(t), ''.join(cig), p


def case_edge():
    d = os.path.join(GOLD, 'cigar_edge')
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(7)

    def rnd(n):
        return ''.join('ACGT'[i] for i in rng.integers(0, 4, n))

    # chrE: hand-placed motifs inside random sequence
    parts = [
        rnd(40),                      # 0
        'ACACACACACACACACACAC',       # 40  AC tandem (20)
        rnd(30),                      # 60
        'TTTTTTTTTTTT',               # 90  homopolymer (12)
        rnd(28),                      # 102
        'acgtNNNNacgtacgt',           # 130 lower case + N
        rnd(34),                      # 146
        'GATTACAGATTACAGATTACA',      # 180 7-mer tandem x3
        rnd(59),                      # 201
        'CAGCAGCAGCAGCAGCAG',         # 260
        rnd(122),                     # 278
    ]
    chrE = ''.join(parts)
    assert len(chrE) == 400
    chrE = chrE[:300] + chrE[300:350].lower() + chrE[350:]
    ref = {'chrE': chrE, 'chr1': rnd(120)}
    names = ['chrE', 'chr1']

    def alt(c):
        return {'A': 'C', 'C': 'G', 'G': 'T', 'T': 'A', 'N': 'A'}[c.upper()]

    rows, tigs = [], {}

    def add(name, chrom, pos, ops, rev=False, index=None):
        o, cig, end = build_tig(ref[chrom], pos, ops)
        stored = revcomp_str(o) if rev else o
        tigs[name] = stored
        clip_l = len(ops[0][1]) if ops and ops[0][0] in 'HS' else 0
        clip_r = len(ops[-1][1]) if ops and ops[-1][0] in 'HS' and len(ops) > 1 else 0
        L = len(stored)
        qpos, qend = (clip_l, L - clip_r) if not rev else (clip_r, L - clip_l)
        idx = len(rows) if index is None else index
        rows.append((chrom, pos, end, idx, name, qpos, qend, L, 'NA', 'NA', 60, rev, '0x0010' if rev else '0x0000',
                     'h1', cig, idx % 10))

    E = ref['chrE']
    # a: matches only
    add('t_a', 'chrE', 5, [('=', 30)])
    # b: hard clip then insertion first (last_op == 'H': no shift), then SNV run, then deletion
    add('t_b', 'chrE', 10, [('H', 'GGGTT'), ('I', 'CA'), ('=', 12), ('X', alt(E[22]) + alt(E[23]) + alt(E[24])),
                           ('=', 5), ('D', 3), ('=', 20), ('H', 'AC')])
    # c: X directly before I (last_op == 'X': no shift although upstream matches)
    add('t_c', 'chrE', 38, [('=', 10), ('X', alt(E[48])), ('I', 'AC'), ('=', 30)])
    # d: tandem insertion after a short '=' run: homology 20 but shift capped by last_oplen = 4
    add('t_d', 'chrE', 20, [('=', 36), ('X', alt(E[56])), ('=', 3), ('I', 'ACAC'), ('=', 40)])
    # d2: same insertion with a long '=' run in front (shift = full tandem, wraps through seq_sv)
    add('t_d2', 'chrE', 0, [('=', 60), ('I', 'ACAC'), ('=', 60)])
    # e: deletion inside homopolymer and inside the 7-mer tandem; 1-base '=' between two indels
    add('t_e', 'chrE', 70, [('=', 28), ('D', 2), ('=', 1), ('I', 'T'), ('=', 80), ('D', 7), ('=', 40)])
    # f: insertion with N, next to N in the reference; deletion over lower case + N
    add('t_f', 'chrE', 110, [('=', 20), ('I', 'ANNT'), ('=', 4), ('D', 6), ('=', 30)])
    # g: homology running into the contig start / end (alignment starts at 0 of both, ends at contig end)
    add('t_g', 'chr1', 0, [('=', 3), ('I', ref['chr1'][0:3]), ('=', 114), ('D', 2), ('=', 1)])
    # h: soft clips
    add('t_h', 'chrE', 200, [('S', 'ACG'), ('=', 10), ('I', 'g'), ('=', 10), ('S', 'T')])
    # i: lower-case insertion / mixed case (SEQ keeps case, homology folds it)
    add('t_i', 'chrE', 255, [('=', 23), ('I', 'cagCAG'), ('=', 60)])
    # j: reverse-strand row with SNV, INS, DEL and IUPAC codes in the contig (ALT complemented, case kept)
    add('t_j', 'chrE', 150, [('H', 'TTGCA'), ('=', 20), ('X', 'R'), ('=', 9), ('I', 'GATTACA'), ('=', 30), ('D', 5),
                            ('=', 10), ('X', 'y' + 'K'), ('=', 12), ('H', 'GG')], rev=True)
    # k: reverse-strand row whose insertion shifts left through the CAG tandem
    add('t_k', 'chrE', 240, [('=', 38), ('I', 'CAGCAG'), ('=', 70)], rev=True)
    # l: insertion longer than the preceding '=' run, 1-base '=' separators
    add('t_l', 'chrE', 88, [('=', 2), ('I', 'TTTTTTTT'), ('=', 1), ('D', 1), ('=', 1), ('I', 'TT'), ('=', 40)])
    # m: two rows over the same locus with the same and with different alleles (duplicate IDs / ID tie-break)
    add('t_m1', 'chrE', 300, [('=', 10), ('X', 'A' if E[310].upper() != 'A' else 'C'), ('=', 10), ('I', 'GGA'), ('=', 30)])
    add('t_m2', 'chrE', 300, [('=', 10), ('X', 'A' if E[310].upper() != 'A' else 'C'), ('=', 10), ('I', 'GGA'), ('=', 30)])
    add('t_m3', 'chrE', 300, [('=', 10), ('X', 'T' if E[310].upper() != 'T' else 'G'), ('=', 10), ('I', 'GGAGGAGGAG'),
                             ('=', 9), ('D', 1), ('=', 20)], rev=True)
    # p: row with an empty CIGAR string (no operations)
    rows.append(('chr1', 7, 7, len(rows), 't_a', 0, 0, len(tigs['t_a']), 'NA', 'NA', 60, False, '0x0000', 'h1', '',
                 len(rows) % 10))

    df_align = pd.DataFrame(rows, columns=ALIGN_COLS)
    # trim table: row 1 shrunk (TRIM for flank variants), row 3 absent (=> -1 => TRIM)
    df_trim = df_align.copy()
    df_trim.loc[df_trim['INDEX'] == 1, 'POS'] += 14
    df_trim.loc[df_trim['INDEX'] == 1, 'END'] -= 18
    df_trim = df_trim.loc[df_trim['INDEX'] != 3]

    tnames = list(tigs)
    write_fa(os.path.join(d, 'ref.fa'), names, ref)
    write_fa(os.path.join(d, 'tig.fa'), tnames, tigs)
    df_align.to_csv(os.path.join(d, 'align.tsv'), sep='\t', index=False)
    df_trim.to_csv(os.path.join(d, 'trim.tsv'), sep='\t', index=False)
    snv, insdel = run_reference(d, df_align, df_trim)
    print('cigar_edge', snv.shape, insdel.shape)
    print(insdel[['POS', 'END', 'ID', 'QRY_REGION', 'LEFT_SHIFT', 'HOM_REF', 'HOM_TIG', 'SEQ', 'FILTER']].to_string())

    # empty table
    d0 = os.path.join(GOLD, 'cigar_empty')
    os.makedirs(d0, exist_ok=True)
    write_fa(os.path.join(d0, 'ref.fa'), names, ref)
    write_fa(os.path.join(d0, 'tig.fa'), tnames, tigs)
    df_align.iloc[0:0].to_csv(os.path.join(d0, 'align.tsv'), sep='\t', index=False)
    df_trim.iloc[0:0].to_csv(os.path.join(d0, 'trim.tsv'), sep='\t', index=False)
    snv0, insdel0 = pavlib.cigarcall.make_insdel_snv_calls(
        df_align.iloc[0:0], os.path.join(d0, 'ref.fa'), os.path.join(d0, 'tig.fa'), 'h1', version_id=False)
    snv0.to_csv(os.path.join(d0, 'snv.tsv'), sep='\t', index=False)
    insdel0.to_csv(os.path.join(d0, 'insdel.tsv'), sep='\t', index=False)
    print('cigar_empty', snv0.shape, insdel0.shape)

    # ---- error cases: same inputs, one row's CIGAR replaced -----------------------------------------------
    errors = []

    def err_case(label, edits):
        df = df_align.copy()
        for idx, cig in edits.items():
            df.loc[df['INDEX'] == idx, 'CIGAR'] = cig
        try:
            pavlib.cigarcall.make_insdel_snv_calls(df, os.path.join(d, 'ref.fa'), os.path.join(d, 'tig.fa'), 'h1',
                                                   version_id=False)
            raise AssertionError(f'{label}: reference did not raise')
        except (RuntimeError, IndexError, TypeError) as ex:
            errors.append({'label': label, 'edits': {str(k): v for k, v in edits.items()},
                           'type': type(ex).__name__, 'message': str(ex)})
            print(f'  {label}: {type(ex).__name__}: {ex}')

    err_case('M_op', {2: '10=5M26='})
    err_case('N_op', {4: '30=4N30=2I60='})
    err_case('P_op', {0: '5H3P30='})
    err_case('missing_len', {5: '28=2D=1I80=7D40='})
    err_case('missing_len_first', {0: '=30'})
    err_case('unknown_op', {6: '20=4Z4=6D30='})
    err_case('truncated', {1: '5H2I12=3X5=3D20=2'})
    err_case('only_digits', {1: '1234'})
    err_case('first_error_wins_row_order', {3: '36=1X3=4I40M', 1: '5H2I12Q'})
    err_case('first_error_wins_op_order', {3: '36=1M3=4I4Q'})
    err_case('tok_before_M_same_row', {3: '36=1X3Q4I40M'})
    with open(os.path.join(GOLD, 'cigar_errors.json'), 'w') as fh:
        json.dump(errors, fh, indent=1)


# ---------------------------------------------------------------------------------------------------------
# Known answers for the small pure functions on the path
# ---------------------------------------------------------------------------------------------------------

def case_kat():
    rng = np.random.default_rng(99)
    T = 'ACGTACGTACGTTTGCA'
    hom = []
    fixed = [('L', 11, T, 'ACGT'), ('L', 11, T, 'CGTA'), ('L', 3, T, 'ACGT'), ('L', 0, T, 'A'), ('L', -1, T, 'A'),
             ('L', 11, 'ACGTACGNACGT', 'ACGT'), ('L', 5, 'AAAAAA', 'A'), ('R', 0, T, 'ACGT'), ('R', 12, T, 'T'),
             ('R', 17, T, 'A'), ('R', 4, 'ACGTACGNACGT', 'ACGT'), ('R', 0, 'AAAAAA', 'A'), ('R', 0, 'acgt', 'ACGT'),
             ('L', 3, 'ACGT', 'ACGN'), ('R', 0, 'ACGT', 'NCGT'), ('L', 7, 'CACACACA', 'CA'), ('R', 1, 'CACACACAT', 'AC')]
    for _ in range(300):
        unit = ''.join('ACGT'[i] for i in rng.integers(0, 4, int(rng.integers(1, 6))))
        seq = ''.join('ACGT'[i] for i in rng.integers(0, 4, int(rng.integers(0, 12)))) + unit * int(rng.integers(1, 9)) + \
            ''.join('ACGTN'[i] for i in rng.integers(0, 5, int(rng.integers(0, 12))))
        sv = unit * int(rng.integers(1, 3)) if rng.random() < 0.7 else \
            ''.join('ACGT'[i] for i in rng.integers(0, 4, int(rng.integers(1, 8))))
        pos = int(rng.integers(-1, len(seq) + 1))
        fixed.append(('L' if rng.random() < 0.5 else 'R', pos, seq, sv))
    for d, pos, seq, sv in fixed:
        if d == 'L':
            if pos >= len(seq):
                pos = len(seq) - 1
            val = pavlib.call.left_homology(pos, seq, sv)
        else:
            val = pavlib.call.right_homology(pos, seq, sv)
        hom.append({'dir': d, 'pos': pos, 'seq': seq, 'sv': sv, 'value': int(val)})
    tok = []
    for c in ['100H5=1X3I10=2D7=50H', '1=', '12345678=', '3S4=2X1I1D1N1P1M', '']:
        tok.append({'cigar': c, 'tuples': [[int(l), o] for l, o in pavlib.align.cigar_str_to_tuples(c)]})
    with open(os.path.join(GOLD, 'kat.json'), 'w') as fh:
        json.dump({'homology': hom, 'tokenize': tok}, fh, indent=0)
    print('kat', len(hom), 'homology answers;', len(tok), 'tokenizer answers')


# ---------------------------------------------------------------------------------------------------------
# BASELINE.json configs[0]: one 1 Mb contig vs a 1 Mb "chr20 slice" through pavlib.cigarcall itself.
# The 2 MB of sequence are regenerated from the seed (pav_amd.synth.config1) wherever the case is needed; what is committed
# is the digest of what the reference wrote for it: tests/golden/config1.json (md5 of the two TSV texts, row counts, and the
# md5 of the generated inputs, so that a drifting generator is told apart from a drifting caller).
# ---------------------------------------------------------------------------------------------------------

def case_config1():
    import hashlib
    import tempfile
    import time
    hap = synth.config1()
    with tempfile.TemporaryDirectory(prefix='pav_config1_') as d:
        synth.write_fasta(os.path.join(d, 'ref.fa'), hap.ref.names, hap.ref.seqs, line=80)
        synth.write_fasta(os.path.join(d, 'tig.fa'), hap.tig_names, hap.tig_seqs, line=80)
        t0 = time.time()
        snv, insdel = run_reference(d, hap.df_align, hap.df_trim)
        wall = time.time() - t0
        text = {}
        for name in ('snv', 'insdel'):
            with open(os.path.join(d, name + '.tsv'), 'rb') as fh:
                text[name] = fh.read()
    md5 = lambda b: hashlib.md5(b).hexdigest()   # noqa: E731
    align_text = hap.df_align.to_csv(sep='\t', index=False).encode()
    out = {
        'config': 'BASELINE.json configs[0]: pav_amd.synth.config1(seed=1001) - one 1 Mb contig vs a 1 Mb chr20 slice',
        'reference': 'pavlib.cigarcall.make_insdel_snv_calls(version_id=False) + the FILTER step of rule call_cigar '
                     '(rules/call.snakefile:813-842), run by tools/refharness/gen_golden_cigar.py',
        'inputs': {'ref_md5': md5(hap.ref.seqs['chr20'].tobytes()),
                   'tig_md5': md5(b''.join(hap.tig_seqs[n].tobytes() for n in hap.tig_names)),
                   'align_tsv_md5': md5(align_text), 'n_aln': int(hap.df_align.shape[0])},
        'snv': {'rows': int(snv.shape[0]), 'tsv_md5': md5(text['snv']), 'tsv_bytes': len(text['snv'])},
        'insdel': {'rows': int(insdel.shape[0]), 'tsv_md5': md5(text['insdel']), 'tsv_bytes': len(text['insdel']),
                   'max_left_shift': int(insdel['LEFT_SHIFT'].max())},
    }
    with open(os.path.join(GOLD, 'config1.json'), 'w') as fh:
        json.dump(out, fh, indent=1)
        fh.write('\n')
    print('config1', snv.shape, insdel.shape, 'reference wall %.2f s' % wall)


if __name__ == '__main__':
    os.makedirs(GOLD, exist_ok=True)
    case_synth()
    case_edge()
    case_kat()
    case_config1()
