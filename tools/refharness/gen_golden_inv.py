#!/usr/bin/env python3
"""
Generate the inversion-path golden vectors by running the *reference itself* (pavlib.inv.scan_for_inv, which
spawns scripts/density.py -> scipy gaussian_kde) on small seeded loci.  Build container only.

Outputs (committed): tests/golden/inv_<case>/
    ref.fa(.fai) tig.fa(.fai) align.tsv flag.tsv        inputs (align.tsv is the trim-tigref table)
    scans.json                                          per flagged region: every scan iteration (regions, table
                                                        size, rl_encoder runs, digests), log lines, the InvCall
                                                        fields and the INV BED row of rule call_inv_batch
    density_<ID>.npz                                    the density table of each call (exact float64)
and tests/golden/lift_kat.json, tests/golden/region_kat.json.

kanapy / pysam / Bio / svpoplib are absent from the reference snapshot and stubbed by tools/refharness/shims
(semantics: SURVEY.md section 8(c)); columns that depend on the k-mer integer encoding (KMER, MATCH) are therefore
"parity unpinned" and flagged as such in scans.json.
"""

import hashlib
import io
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
import kanapy  # noqa: E402
import svpoplib  # noqa: E402

from pav_amd import synth  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')


def digest(arr):
    return hashlib.sha1(np.ascontiguousarray(arr).tobytes()).hexdigest()


def region_dict(r):
    if r is None:
        return None
    return {'chrom': r.chrom, 'pos': int(r.pos), 'end': int(r.end), 'is_rev': bool(r.is_rev),
            'pos_aln_index': _aln(r.pos_aln_index), 'end_aln_index': _aln(r.end_aln_index)}


def _aln(x):
    if x is None:
        return None
    out = []
    for v in x:
        out.append([int(w) for w in v] if isinstance(v, (tuple, list)) else int(v))
    return out


class Capture:
    """Record every scan iteration by wrapping the two library calls scan_for_inv makes per iteration."""

    def __init__(self, align_lift):
        self.align_lift = align_lift
        self.iterations = []
        self._orig_rl = pavlib.density.rl_encoder
        self._orig_lift = align_lift.lift_region_to_qry

    def __enter__(self):
        cap = self

        def lift(region):
            out = cap._orig_lift(region)
            cap.iterations.append({'region_ref': region_dict(region), 'region_tig': region_dict(out)})
            return out

        def rl(df, state_col='STATE'):
            runs = [tuple(int(v) for v in rec) for rec in cap._orig_rl(df, state_col)]
            it = cap.iterations[-1]
            it['n_rows'] = int(df.shape[0])
            it['finalised'] = 'KERN_FWD' in df.columns
            it['state_rl'] = runs
            it['index_sha1'] = digest(df['INDEX'].to_numpy(dtype=np.int64))
            it['state_mer_sha1'] = digest(df['STATE_MER'].to_numpy(dtype=np.int8))
            it['state_sha1'] = digest(df['STATE'].to_numpy(dtype=np.int8))
            if it['finalised']:
                it['kern_sum'] = [float(df[c].sum()) for c in ('KERN_FWD', 'KERN_FWDREV', 'KERN_REV')]
            return iter(runs)

        self.align_lift.lift_region_to_qry = lift
        pavlib.density.rl_encoder = rl
        return self

    def __exit__(self, *exc):
        pavlib.density.rl_encoder = self._orig_rl
        self.align_lift.lift_region_to_qry = self._orig_lift
        return False


def inv_bed_row(inv_call, hap, flag_type, tig_fa):
    """rules/call_inv.snakefile:203-282 restated (the rule body cannot be imported)."""
    seq = pavlib.seq.region_seq_fasta(inv_call.region_tig_outer, tig_fa, rev_compl=inv_call.region_tig_outer.is_rev)
    align_index = ','.join(sorted(pavlib.util.collapse_to_set(
        (inv_call.region_ref_outer.pos_aln_index, inv_call.region_ref_outer.end_aln_index,
         inv_call.region_ref_inner.pos_aln_index, inv_call.region_ref_inner.end_aln_index), to_type=str)))
    return pd.Series(
        [inv_call.region_ref_outer.chrom, inv_call.region_ref_outer.pos, inv_call.region_ref_outer.end,
         inv_call.id, 'INV', inv_call.svlen, hap,
         inv_call.region_tig_outer.to_base1_string(), '-' if inv_call.region_tig_outer.is_rev else '+', 0,
         inv_call.region_ref_inner.to_base1_string(), inv_call.region_tig_inner.to_base1_string(),
         inv_call.region_ref_discovery.to_base1_string(), inv_call.region_tig_discovery.to_base1_string(),
         inv_call.region_flag.region_id(), flag_type, align_index, pavlib.inv.CALL_SOURCE, 'PASS', seq],
        index=['#CHROM', 'POS', 'END', 'ID', 'SVTYPE', 'SVLEN', 'HAP', 'QRY_REGION', 'QRY_STRAND', 'CI',
               'RGN_REF_INNER', 'RGN_QRY_INNER', 'RGN_REF_DISC', 'RGN_QRY_DISC', 'FLAG_ID', 'FLAG_TYPE', 'ALIGN_INDEX',
               'CALL_SOURCE', 'FILTER', 'SEQ'])


def run_case(name, ref, hap, flags, scan_kwargs=None, k=31):
    d = os.path.join(GOLD, name)
    os.makedirs(d, exist_ok=True)
    synth.write_fasta(os.path.join(d, 'ref.fa'), ref.names, ref.seqs, line=100)
    synth.write_fasta(os.path.join(d, 'tig.fa'), hap.tig_names, hap.tig_seqs, line=100)
    hap.df_trim.to_csv(os.path.join(d, 'align.tsv'), sep='\t', index=False)
    df_flag = pd.DataFrame(
        [(c, p, e, f'{c}-{p}-RGN-{e - p}', 'RGN', e - p, t, 0, 0, True, i % 2) for i, (c, p, e, t, _) in enumerate(flags)],
        columns=['#CHROM', 'POS', 'END', 'ID', 'SVTYPE', 'SVLEN', 'TYPE', 'COUNT_INDEL', 'COUNT_SNV', 'TRY_INV', 'BATCH'])
    df_flag.to_csv(os.path.join(d, 'flag.tsv'), sep='\t', index=False)

    ref_fa, tig_fa = os.path.join(d, 'ref.fa'), os.path.join(d, 'tig.fa')
    k_util = kanapy.util.kmer.KmerUtil(k)
    df_aln = pd.read_csv(os.path.join(d, 'align.tsv'), sep='\t')
    align_lift = pavlib.align.AlignLift(df_aln, svpoplib.ref.get_df_fai(tig_fa + '.fai'))

    scans = []
    for (c, p, e, ftype, kw) in flags:
        kwargs = dict(scan_kwargs or {})
        kwargs.update(kw or {})
        log = io.StringIO()
        region_flag = pavlib.seq.Region(c, p, e)
        with Capture(align_lift) as cap:
            err = None
            try:
                call = pavlib.inv.scan_for_inv(region_flag, ref_fa, tig_fa, align_lift, k_util, threads=1, log=log, **kwargs)
            except RuntimeError as ex:
                call, err = None, str(ex)
        rec = {'flag': {'chrom': c, 'pos': p, 'end': e, 'type': ftype}, 'kwargs': kwargs,
               'iterations': cap.iterations, 'log': log.getvalue().splitlines(), 'error': err, 'call': None}
        if call is not None:
            row = inv_bed_row(call, hap.hap, ftype, tig_fa)
            rec['call'] = {
                'id': call.id, 'svlen': int(call.svlen),
                'region_ref_outer': region_dict(call.region_ref_outer), 'region_ref_inner': region_dict(call.region_ref_inner),
                'region_tig_outer': region_dict(call.region_tig_outer), 'region_tig_inner': region_dict(call.region_tig_inner),
                'region_ref_discovery': region_dict(call.region_ref_discovery),
                'region_tig_discovery': region_dict(call.region_tig_discovery),
                'bed_row': {k2: (int(v) if isinstance(v, (int, np.integer)) else v) for k2, v in row.items()},
                'unpinned_columns': ['KMER', 'MATCH'],
            }
            df = call.df
            buf = io.StringIO()
            df.to_csv(buf, sep='\t', index=False)
            rec['call']['density_tsv_sha1'] = hashlib.sha1(buf.getvalue().encode()).hexdigest()
            np.savez_compressed(
                os.path.join(d, f'density_{call.id}.npz'),
                INDEX=df['INDEX'].to_numpy(dtype=np.int64), STATE_MER=df['STATE_MER'].to_numpy(dtype=np.int8),
                STATE=df['STATE'].to_numpy(dtype=np.int8), KERN_FWD=df['KERN_FWD'].to_numpy(dtype=np.float64),
                KERN_FWDREV=df['KERN_FWDREV'].to_numpy(dtype=np.float64), KERN_REV=df['KERN_REV'].to_numpy(dtype=np.float64),
                KMER=df['KMER'].to_numpy(dtype=np.uint64), FLANK=df['FLANK'].to_numpy(dtype=str),
                MATCH=df['MATCH'].fillna('NA').to_numpy(dtype=str))
            # first rows of the TSV text, to pin pandas' float formatting of the table the rule writes
            rec['call']['density_tsv_head'] = buf.getvalue().splitlines()[:6]
        scans.append(rec)
        print(f'  {name} {c}:{p}-{e} {kwargs} -> {call}  iterations={len(cap.iterations)} '
              f'last_log={rec["log"][-1] if rec["log"] else ""}')
    with open(os.path.join(d, 'scans.json'), 'w') as fh:
        json.dump(scans, fh, indent=1)


def locus(seed, length, inv, rev, name='chrT', n_run=None, tandem=None, small_ir=None):
    """One chromosome with one planted inversion (pos, end, repeat) aligned end-to-end by one contig."""
    ref = synth.make_reference(seed, {name: length}, n_every=0, inv_every=0, threads=1)
    s = ref.seqs[name]
    if n_run:
        s[n_run[0]:n_run[1]] = ord('N')
    if tandem:                               # low-complexity block: `copies` copies of a random unit
        p, unit, copies = tandem
        rng = np.random.default_rng(seed)
        u = np.frombuffer(b'ACGT', dtype=np.uint8)[rng.integers(0, 4, unit)]
        s[p:p + unit * copies] = np.tile(u, copies)
    if small_ir:                             # short inverted pair: a state with < 20 k-mers (removed)
        p, q, ln = small_ir
        s[q:q + ln] = synth.revcomp(s[p:p + ln])
    invs = []
    if inv:
        v = synth.Inversion(name, inv[0], inv[1], inv[2])
        if v.repeat:
            s[v.end - v.repeat:v.end] = synth.revcomp(s[v.pos:v.pos + v.repeat])
        invs.append(v)
    ref.inversions = invs
    hap = synth.make_haplotype(ref, seed * 64, 'h1', segments={name: [(0, length)]}, rev_frac=1.0 if rev else 0.0,
                               threads=1, decoys_per_inv=0, snv_rate=1e-3, indel_rate=2e-4)
    return ref, hap


def main():
    os.makedirs(GOLD, exist_ok=True)

    # 1. 12 kb inversion with 1.5 kb inverted-repeat flanks, forward contig; decoys; partial flag => expansion;
    #    region limit; short inverted pair (state removed by the min-state-count rule)
    ref, hap = locus(5, 80_000, (30_000, 42_000, 1_500), rev=False, small_ir=(15_000, 17_200, 46))
    run_case('inv_fwd', ref, hap, [
        ('chrT', 14_200, 17_800, 'CLUSTER_INDEL', None),
        ('chrT', 30_000, 42_000, 'CLUSTER_SNV', None),
        ('chrT', 34_000, 38_000, 'CLUSTER_SNV', None),
        ('chrT', 34_000, 38_000, 'CLUSTER_SNV', {'max_region_size': 15_000}),
        ('chrT', 31_000, 33_000, 'CLUSTER_SNV,MATCH_INDEL', {'min_exp_count': 2}),
        ('chrT', 54_759, 56_465, 'MATCH_SV', None),
        ('chrT', 100, 900, 'CLUSTER_INDEL', None),
        ('chrT', 79_000, 79_900, 'CLUSTER_INDEL', None),
    ])

    # 2. same locus, contig stored reverse-complemented (REV row => density.py -r true)
    ref, hap = locus(6, 70_000, (28_000, 37_000, 1_200), rev=True)
    run_case('inv_rev', ref, hap, [
        ('chrT', 28_000, 37_000, 'CLUSTER_SNV', None),
        ('chrT', 10_000, 12_500, 'CLUSTER_INDEL', None),
    ])

    # 3. 3 kb inversion without repeats; N run; low-complexity tandem; small chromosome limits
    ref, hap = locus(7, 60_000, (20_000, 23_000, 0), rev=False, n_run=(40_000, 52_000), tandem=(8_000, 7, 180))
    run_case('inv_small', ref, hap, [
        ('chrT', 20_000, 23_000, 'CLUSTER_SNV', None),
        ('chrT', 43_000, 49_000, 'CLUSTER_INDEL', None),      # inside the N run: no reference k-mers
        ('chrT', 39_000, 40_500, 'CLUSTER_INDEL', None),      # mostly N after expansion (still >= 2000 informative)
        ('chrT', 40_200, 41_200, 'CLUSTER_INDEL', None),      # < 2000 informative k-mers: un-finalised table, STATE = -1
        ('chrT', 8_200, 9_000, 'CLUSTER_INDEL', None),        # k-mer count > 100
    ])

    # 4. inversion filling most of a small chromosome: expansion reaches the reference limits
    ref, hap = locus(8, 24_000, (5_000, 19_500, 0), rev=False)
    run_case('inv_limits', ref, hap, [
        ('chrT', 9_000, 15_000, 'CLUSTER_SNV', None),
    ])

    # 5. two alignment rows: a region that cannot be lifted (spans the gap between rows)
    ref = synth.make_reference(9, {'chrT': 50_000}, n_every=0, inv_every=0, threads=1)
    hap = synth.make_haplotype(ref, 9 * 64, 'h1', segments={'chrT': [(0, 24_000), (26_000, 50_000)]}, rev_frac=0.0,
                               threads=1, decoys_per_inv=0)
    run_case('inv_nolift', ref, hap, [
        ('chrT', 23_000, 27_000, 'CLUSTER_SNV', None),
        ('chrT', 24_500, 25_500, 'CLUSTER_SNV', None),
    ])

    # ---- AlignLift known answers (pavlib/align/lift.py) on the inv_fwd and inv_rev alignments ---------------
    kat = []
    for case in ('inv_fwd', 'inv_rev', 'inv_nolift'):
        d = os.path.join(GOLD, case)
        df_aln = pd.read_csv(os.path.join(d, 'align.tsv'), sep='\t')
        fai = svpoplib.ref.get_df_fai(os.path.join(d, 'tig.fa.fai'))
        lift = pavlib.align.AlignLift(df_aln, fai)
        rng = np.random.default_rng(3)
        row = df_aln.iloc[0]
        # positions around every indel boundary + random ones
        pts_ref, pts_qry = set(), set()
        sub, qry = int(row['POS']), 0
        for ln, op in pavlib.align.cigar_str_to_tuples(row['CIGAR']):
            if op in 'ID':
                for dlt in (-1, 0, 1, ln - 1, ln, ln + 1):
                    pts_ref.add(sub + dlt)
                    pts_qry.add(qry + dlt)
            if op in '=XD':
                sub += ln
            if op in '=XISH':
                qry += ln
        pts_ref = sorted(pts_ref)[:120] + [int(x) for x in rng.integers(0, int(df_aln['END'].max()) + 50, 60)] + \
            [int(row['POS']), int(row['END']) - 1, int(row['END'])]
        tl = int(fai[row['QRY_ID']])
        pts_qry = sorted(x for x in pts_qry if 0 <= x <= tl)[:120] + [int(x) for x in rng.integers(0, tl + 1, 60)] + \
            [int(row['QRY_POS']), int(row['QRY_END']) - 1, int(row['QRY_END'])]
        for p in pts_ref:
            out = lift.lift_to_qry(row['#CHROM'], p)
            kat.append({'case': case, 'dir': 'to_qry', 'id': row['#CHROM'], 'pos': int(p),
                        'out': None if out is None else [out[0], int(out[1]), bool(out[2]), int(out[3]), int(out[4]), [int(v) for v in out[5]]]})
        for p in pts_qry:
            for gap in (False, True):
                try:
                    out = lift.lift_to_sub(row['QRY_ID'], p, gap)
                    err = None
                except RuntimeError as ex:
                    out, err = None, str(ex)
                kat.append({'case': case, 'dir': 'to_sub', 'id': row['QRY_ID'], 'pos': int(p), 'gap': gap, 'error': err,
                            'out': None if out is None else [out[0], int(out[1]), None if out[2] is None else bool(out[2]),
                                                             int(out[3]), int(out[4]), [int(v) for v in out[5]]]})
    with open(os.path.join(GOLD, 'lift_kat.json'), 'w') as fh:
        json.dump(kat, fh)
    print('lift_kat', len(kat))

    # ---- Region known answers (pavlib/seq.py) ----------------------------------------------------------------
    rk = []
    fai = pd.Series({'c': 1000, 'chr1': 250_000})
    rng = np.random.default_rng(4)
    cases = [('c', 100, 200, 4000, 0.5), ('c', 100, 200, 150, 0.25), ('c', 0, 10, 50, 0.75), ('c', 990, 1000, 300, 0.5),
             ('zz', 5, 9, 100, 0.5), ('chr1', 1000, 1500, 4000, 0.5)]
    for _ in range(60):
        p = int(rng.integers(0, 250_000))
        e = p + int(rng.integers(1, 40_000))
        cases.append(('chr1', p, min(e, 250_000), int(rng.integers(1, 400_000)), float(rng.choice([0.25, 0.5, 0.75]))))
    for chrom, p, e, bp, bal in cases:
        r = pavlib.seq.Region(chrom, p, e)
        r.expand(np.int32(bp), min_pos=0, max_end=fai, shift=True, balance=bal)
        rk.append({'chrom': chrom, 'pos': p, 'end': e, 'expand_bp': bp, 'balance': bal, 'out': [int(r.pos), int(r.end)],
                   'base1': r.to_base1_string(), 'region_id': r.region_id(), 'len': len(r)})
    strings = ['chr1:1001-1500', 'tig0001:5-5', 'chr1:1,001-2,500']
    sk = []
    for s in strings:
        r = pavlib.seq.region_from_string(s)
        sk.append({'s': s, 'out': [r.chrom, int(r.pos), int(r.end), bool(r.is_rev)]})
    r = pavlib.seq.region_from_id('chr1-1001-RGN-500')
    r2 = pavlib.seq.Region('chr1', 200, 100)
    with open(os.path.join(GOLD, 'region_kat.json'), 'w') as fh:
        json.dump({'expand': rk, 'from_string': sk, 'from_id': ['chr1-1001-RGN-500', r.chrom, int(r.pos), int(r.end)],
                   'swapped': [int(r2.pos), int(r2.end), bool(r2.is_rev)]}, fh, indent=0)
    print('region_kat', len(rk))


def hap_case():
    """6. A small multi-chromosome haplotype (several alignment rows, reverse rows, inverted repeats, decoys):
    every flagged region scanned by the reference - the batched-scan parity case."""
    ref = synth.make_reference(77, {'chr1': 260_000, 'chr2': 200_000, 'chr10': 160_000}, n_every=0, inv_every=0, threads=1)
    plan = [('chr1', 40_000, 52_000, 1_000), ('chr1', 150_000, 153_500, 0), ('chr2', 60_000, 95_000, 2_500),
            ('chr10', 30_000, 36_000, 600), ('chr10', 100_000, 124_000, 0)]
    for c, p, e, rep in plan:
        s = ref.seqs[c]
        if rep:
            s[e - rep:e] = synth.revcomp(s[p:p + rep])
        ref.inversions.append(synth.Inversion(c, p, e, rep))
    ref.seqs['chr2'][150_000:153_000] = ord('N')
    hap = synth.make_haplotype(ref, 77 * 64, 'h1', seg_median=70_000, seg_sigma=0.5, rev_frac=0.5, threads=1,
                               decoys_per_inv=2, zone_factor=1, zone_pad=9_000)
    flags = [(r['#CHROM'], int(r['POS']), int(r['END']), r['TYPE'], None) for _, r in hap.df_flag.iterrows()]
    print('  inv_hap: rows', hap.df_align.shape[0], 'planted', hap.stats['n_inv'], 'flagged', len(flags))
    run_case('inv_hap', ref, hap, flags)


if __name__ == '__main__':
    if os.environ.get('PAV_GOLDEN_ONLY', '') != 'hap':
        main()
    hap_case()
