"""Host-side mirrors on the inversion path (no GPU): Region, AlignLift, srs tree, rl_encoder, k-mer utility,
and the CPU oracle's density restatement against the golden vectors made by the reference."""
import hashlib
import json
import os

import numpy as np
import pandas as pd
import pytest

import util
from pav_amd import density as pavden
from pav_amd import inv as pavinv
from pav_amd import seq as pavseq
from pav_amd.align import AlignLift, cigar_str_to_tuples
from pav_amd.fasta import open_fasta, read_fai
from pav_amd.kmer import KmerUtil

GOLD = util.GOLD
INV_CASES = ['inv_fwd', 'inv_rev', 'inv_small', 'inv_limits', 'inv_nolift', 'inv_hap', 'inv_k32']   # inv_k32: inv_k_size = 32, poly-T / poly-A tracts


def sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_region_known_answers():
    with open(os.path.join(GOLD, 'region_kat.json')) as fh:
        kat = json.load(fh)
    fai = pd.Series({'c': 1000, 'chr1': 250_000})
    for k in kat['expand']:
        r = pavseq.Region(k['chrom'], k['pos'], k['end'])
        r.expand(np.int32(k['expand_bp']), min_pos=0, max_end=fai, shift=True, balance=k['balance'])
        assert [r.pos, r.end] == k['out'], k
        assert (r.to_base1_string(), r.region_id(), len(r)) == (k['base1'], k['region_id'], k['len'])
    for k in kat['from_string']:
        r = pavseq.region_from_string(k['s'])
        assert [r.chrom, r.pos, r.end, bool(r.is_rev)] == k['out']
    rid, chrom, pos, end = kat['from_id']
    r = pavseq.region_from_id(rid)
    assert (r.chrom, r.pos, r.end) == (chrom, pos, end)
    r = pavseq.Region('chr1', 200, 100)
    assert [r.pos, r.end, bool(r.is_rev)] == kat['swapped']


def test_alignlift_known_answers():
    with open(os.path.join(GOLD, 'lift_kat.json')) as fh:
        kat = json.load(fh)
    lifts = {}
    for k in kat:
        c = k['case']
        if c not in lifts:
            d = os.path.join(GOLD, c)
            lifts[c] = AlignLift(pd.read_csv(os.path.join(d, 'align.tsv'), sep='\t'), read_fai(os.path.join(d, 'tig.fa.fai')))
        err = None
        try:
            out = lifts[c].lift_to_qry(k['id'], k['pos']) if k['dir'] == 'to_qry' else lifts[c].lift_to_sub(k['id'], k['pos'], k['gap'])
        except RuntimeError as ex:
            out, err = None, str(ex)
        norm = None if out is None else [out[0], int(out[1]), None if out[2] is None else bool(out[2]), int(out[3]),
                                         int(out[4]), [int(v) for v in out[5]]]
        assert norm == k['out'], k
        if k['dir'] == 'to_sub':
            assert err == k.get('error'), k


def test_cigar_iterator_known_answers():
    for k in util.kat()['tokenize']:
        assert [[l, o] for l, o in cigar_str_to_tuples(k['cigar'])] == k['tuples']


def test_srs_tree_and_kmer_util():
    t = pavinv.get_srs_tree(None)
    assert list(t[12345])[0].data == 20
    t = pavinv.get_srs_tree([(0, 20), (100000, 40), (500000, 80)])
    assert [int(list(t[x])[0].data) for x in (5, 99999, 100000, 499999, 500000, 10 ** 7)] == [20, 20, 40, 40, 80, 80]
    with pytest.raises(RuntimeError):
        pavinv.get_srs_tree([(0, 2)])
    ku = KmerUtil(5)
    km = ku.to_kmer('ACGTT')
    assert ku.to_string(km) == 'ACGTT' and ku.to_string(ku.rev_complement(km)) == 'AACGT'
    assert ku.canonical_complement(km) == min(km, ku.rev_complement(km))


def test_rl_encoder_matches_golden_runs():
    for case in ('inv_fwd', 'inv_rev', 'inv_small'):
        with open(os.path.join(GOLD, case, 'scans.json')) as fh:
            scans = json.load(fh)
        for rec in scans:
            if rec['call'] is None:
                continue
            g = np.load(os.path.join(GOLD, case, 'density_%s.npz' % rec['call']['id']))
            df = pd.DataFrame({'STATE': g['STATE'].astype(np.int64), 'INDEX': g['INDEX']})
            assert [list(r) for r in pavden.rl_encoder(df)] == rec['iterations'][-1]['state_rl']


@pytest.mark.parametrize('case', INV_CASES)
def test_oracle_density_matches_reference(built, case):
    """oracle/pav_oracle_density.c vs every scan iteration the reference executed: row count, INDEX, STATE_MER, STATE,
    rl_encoder runs exact; KERN_* within 1e-12 relative (np.cov's summation order is not reproducible, exp is libm's)."""
    from oracle import oracle
    d = os.path.join(GOLD, case)
    k = util.case_k(d)
    ref, tig = open_fasta(os.path.join(d, 'ref.fa')), open_fasta(os.path.join(d, 'tig.fa'))
    with open(os.path.join(d, 'scans.json')) as fh:
        scans = json.load(fh)
    for rec in scans:
        o = None
        for it in rec['iterations']:
            rr, rt = it['region_ref'], it['region_tig']
            if rt is None:
                continue
            o = oracle.density(ref[rr['chrom']][rr['pos']:rr['end']], tig[rt['chrom']][rt['pos']:rt['end']], rt['is_rev'], oracle.den_params(k=k))
            if 'n_rows' not in it:
                assert o['status'] == 125
                continue
            assert o['status'] == (0 if it['finalised'] else 1)
            assert o['n'] == it['n_rows']
            assert sha(o['INDEX']) == it['index_sha1'] and sha(o['STATE_MER']) == it['state_mer_sha1']
            assert sha(o['STATE']) == it['state_sha1']
            assert [list(r) for r in oracle.rl_encode(o['STATE'], o['INDEX'])] == it['state_rl']
            if it['finalised']:
                ks = [float(o[c].sum()) for c in ('KERN_FWD', 'KERN_FWDREV', 'KERN_REV')]
                assert np.allclose(ks, it['kern_sum'], rtol=1e-12, atol=0)
        if rec['call'] is not None:
            g = np.load(os.path.join(d, 'density_%s.npz' % rec['call']['id']))
            for c in ('KERN_FWD', 'KERN_FWDREV', 'KERN_REV'):
                assert np.allclose(o[c], g[c], rtol=1e-12, atol=1e-300), c
            assert np.array_equal(o['KMER'], g['KMER'])
            # FLANK / MATCH (pavlib/inv.py:457-561)
            call = rec['call']
            ro, ri, to, ti = (call[k] for k in ('region_ref_outer', 'region_ref_inner', 'region_tig_outer', 'region_tig_inner'))
            chrom = ref[ro['chrom']]
            flank, match = oracle.annotate(o['KMER'], o['INDEX'], k, call['region_ref_discovery']['pos'],
                                           (to['pos'], ti['pos']), (ti['end'], to['end']),
                                           chrom[ro['pos']:ri['pos']], chrom[ri['end']:ro['end']])
            assert np.array_equal(np.array(['', 'UP', 'DN'])[flank], g['FLANK'])
            assert np.array_equal(np.array(['', 'SAME', 'OTHER', 'NA'])[match], g['MATCH'])


def test_oracle_density_matches_reference_on_large_regions(built):
    """tests/golden/inv_large: the reference's own scans of a 150 kb and a 200 kb inversion, flagged inside, so that the regions
    grow over two expansion rounds to 337 / 462 kbp (forward and reverse-complemented contig, an N run in the largest).  The
    oracle on every iteration the reference executed: row count, INDEX / STATE_MER / STATE / KMER digests, rl_encoder runs
    exact; KERN_* sums and the committed sample of table rows (every 499th + every row near a state change) to 1e-12."""
    from oracle import oracle
    d, ref, hap, gold = util.inv_large_case()
    threads = min(8, util.usable_cpus())
    n_big = 0
    for rec in gold['scans']:
        o = None
        for it in rec['iterations']:
            rr, rt = it['region_ref'], it['region_tig']
            o = oracle.density(ref.seqs[rr['chrom']][rr['pos']:rr['end']], hap.tig_seqs[rt['chrom']][rt['pos']:rt['end']],
                               rt['is_rev'], threads=threads)
            if 'n_rows' not in it:
                assert o['status'] == 125 or o['n'] == 0
                continue
            assert o['status'] == (0 if it['finalised'] else 1) and o['n'] == it['n_rows']
            assert sha(o['INDEX']) == it['index_sha1'] and sha(o['STATE_MER']) == it['state_mer_sha1']
            assert sha(o['STATE']) == it['state_sha1']
            assert [list(r) for r in oracle.rl_encode(o['STATE'], o['INDEX'])] == it['state_rl']
            if it['finalised']:
                assert np.allclose([float(o[c].sum()) for c in ('KERN_FWD', 'KERN_FWDREV', 'KERN_REV')], it['kern_sum'],
                                   rtol=1e-12, atol=0)
            n_big += rr['end'] - rr['pos'] > 300_000
        if rec['call'] is not None:
            call = rec['call']
            g = np.load(os.path.join(GOLD, 'inv_large', 'kern_%s.npz' % call['id']))
            assert o['n'] == call['n_rows'] and sha(o['KMER']) == call['kmer_sha1']
            for c in ('KERN_FWD', 'KERN_FWDREV', 'KERN_REV'):
                assert np.allclose(o[c][g['rows']], g[c], rtol=1e-12, atol=1e-300), c
    assert n_big == 2


def _region_dict(r):
    def aln(x):
        return None if x is None else [[int(w) for w in v] if isinstance(v, (tuple, list)) else int(v) for v in x]
    return {'chrom': r.chrom, 'pos': int(r.pos), 'end': int(r.end), 'is_rev': bool(r.is_rev),
            'pos_aln_index': aln(r.pos_aln_index), 'end_aln_index': aln(r.end_aln_index)}


@pytest.mark.parametrize('case', INV_CASES + ['inv_large'])
def test_oracle_driven_scan_matches_reference(built, case, capsys):
    """The WHOLE scan without a GPU: pav_amd.inv's Python state machine (expansion, stop rules, lifts, breakpoints, log text)
    answered by the oracle's tables (tests/oracle_scan.py) against what pavlib.inv.scan_for_inv did on every flagged region of
    the golden cases - same log lines, same None / InvCall, same six regions, same iterations.  This pair is what pins the
    full-size INV calls (tests/golden/fullsize_inv_calls.json)."""
    import oracle_scan
    if case == 'inv_large':
        d, ref, hap, gold = util.inv_large_case()
        scans, lift = gold['scans'], AlignLift(hap.df_trim, hap.tig_lengths)
        ref_names, ref_seqs, tig_names, tig_seqs = ref.names, ref.seqs, hap.tig_names, hap.tig_seqs
    else:
        d = os.path.join(GOLD, case)
        rf, tf = open_fasta(os.path.join(d, 'ref.fa')), open_fasta(os.path.join(d, 'tig.fa'))
        ref_names, ref_seqs, tig_names, tig_seqs = rf.names, {n: rf[n] for n in rf.names}, tf.names, {n: tf[n] for n in tf.names}
        lift = AlignLift(pd.read_csv(os.path.join(d, 'align.tsv'), sep='\t'), read_fai(os.path.join(d, 'tig.fa.fai')))
        with open(os.path.join(d, 'scans.json')) as fh:
            scans = json.load(fh)
    k_util = KmerUtil(31 if case == 'inv_large' else util.case_k(d))
    threads = min(8, util.usable_cpus())
    for rec in scans:
        f = rec['flag']
        try:
            out, log, ctx = oracle_scan.oracle_scan([pavseq.Region(f['chrom'], f['pos'], f['end'])], ref_names, ref_seqs, tig_names,
                                                    tig_seqs, lift, k_util, threads=threads, **rec['kwargs'])
            call = out[0]
        except RuntimeError as ex:
            call = ex
        if rec.get('error'):
            assert isinstance(call, RuntimeError) and str(call) == rec['error']
            continue
        assert log.splitlines() == rec['log'], f
        its = [it for it in rec['iterations'] if it['region_tig'] is not None]
        assert len(ctx.iterations) == len(its)
        for got, it in zip(ctx.iterations, its):
            assert (got[1], got[2], got[4], got[5], bool(got[6])) == (it['region_ref']['pos'], it['region_ref']['end'],
                                                                     it['region_tig']['pos'], it['region_tig']['end'],
                                                                     it['region_tig']['is_rev'])
            assert got[8] == it.get('state_rl', [])
        if rec['call'] is None:
            assert call is None
            continue
        g = rec['call']
        assert call.id == g['id'] and call.svlen == g['svlen']
        for name in ('region_ref_outer', 'region_ref_inner', 'region_tig_outer', 'region_tig_inner', 'region_ref_discovery',
                     'region_tig_discovery'):
            assert _region_dict(getattr(call, name)) == g[name], name
        df = call.df
        if case == 'inv_large':
            assert df.shape[0] == g['n_rows'] and sha(df['STATE'].to_numpy(dtype=np.int8)) == g['state_sha1']
            assert {str(k): int(v) for k, v in df['FLANK'].value_counts().items()} == g['flank_counts']
        else:
            t = np.load(os.path.join(d, 'density_%s.npz' % g['id']))
            assert np.array_equal(df['STATE'].to_numpy(), t['STATE']) and np.array_equal(df['FLANK'].to_numpy(dtype=str), t['FLANK'])
            assert np.array_equal(df['MATCH'].fillna('NA').to_numpy(dtype=str), t['MATCH'])


NEARTIE = ['argmax_search', 'argmax_mirror', 'delta_above', 'delta_below']


def load_neartie(case):
    g = np.load(os.path.join(GOLD, 'den_neartie', case + '.npz'))
    return g, json.loads(str(g['params']))


@pytest.mark.parametrize('case', NEARTIE)
def test_oracle_density_on_constructed_near_ties(built, case):
    """tests/golden/den_neartie (tools/refharness/gen_golden_neartie.py): inputs on which scripts/density.py takes a float
    decision by a hair - the arg-max of one row with a margin of 2e-11 (argmax_search) / of 1e-15, an exact tie up to rounding
    (argmax_mirror), and density_change of one window 1e-10 above / below --staterundelta.  The scalar oracle follows scipy's
    order, so it reproduces the reference's table; only the exact tie may go either way."""
    from oracle import oracle
    g, p = load_neartie(case)
    o = oracle.density(g['ref'], g['tig'], False, oracle.den_params(k=p['k'], state_run_smooth=p['staterunsmooth'],
                                                                    state_run_delta=p['staterundelta']))
    assert o['status'] == 0 and np.array_equal(o['INDEX'], g['INDEX']) and np.array_equal(o['STATE_MER'], g['STATE_MER'])
    for c in ('KERN_FWD', 'KERN_FWDREV', 'KERN_REV'):
        assert np.allclose(o[c], g[c], rtol=1e-12, atol=1e-300), c
    diff = np.flatnonzero(o['STATE'] != g['STATE'])
    if case == 'argmax_mirror':
        assert set(diff) <= {p['row']} and g['STATE'][p['row']] in (0, 2)
        kk = np.array([g[c][p['row']] for c in ('KERN_FWD', 'KERN_REV')])
        assert abs(kk[0] - kk[1]) < 1e-13 * kk.max()
    else:
        assert diff.size == 0
    if case == 'argmax_search':
        r = p['row']
        m = abs(g['KERN_FWD'][r] - g['KERN_REV'][r]) / max(g['KERN_FWD'][r], g['KERN_REV'][r])
        assert 1e-12 < m < 1e-9 and g['KERN_FWDREV'][r] == 0.0
    if case.startswith('delta'):
        a, b = p['window']
        d = max(abs(g[c][a] - g[c][b]) for c in ('KERN_FWD', 'KERN_FWDREV', 'KERN_REV'))
        assert abs(d - p['staterundelta']) < 2e-10 * d and (d > p['staterundelta']) == (case == 'delta_below')
        # the decision shows in the table: interpolated rows lie on the chord, evaluated rows do not
        x = np.arange(a + 1, b)
        chord = g['KERN_FWD'][a] + (g['KERN_FWD'][b] - g['KERN_FWD'][a]) / (b - a) * (x - a)
        on_chord = np.allclose(g['KERN_FWD'][a + 1:b], chord, rtol=1e-13, atol=0)
        assert on_chord == (case == 'delta_above')
