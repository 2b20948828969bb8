"""GPU parity tests of the inversion path: pav_amd.inv.scan_for_inv / the density kernels (through the C ABI) against
the golden vectors produced by the reference (tests/golden/inv_*) and against the CPU oracle on seeded inputs."""
import hashlib
import io
import json
import os

import numpy as np
import pandas as pd
import pytest

import util
from pav_amd import _lib, density as pavden, inv as pavinv, rules, seq as pavseq, synth
from pav_amd.align import AlignLift
from pav_amd.fasta import open_fasta, read_fai
from pav_amd.kmer import KmerUtil

pytestmark = pytest.mark.gpu
GOLD = util.GOLD
INV_CASES = ['inv_fwd', 'inv_rev', 'inv_small', 'inv_limits', 'inv_nolift', 'inv_hap', 'inv_k32']   # inv_k32: inv_k_size = 32, poly-T / poly-A tracts
KERN = ('KERN_FWD', 'KERN_FWDREV', 'KERN_REV')
RTOL = 1e-12      # KERN_* tolerance vs the reference: device exp() and np.cov's summation order


def sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def region_dict(r):
    def aln(x):
        return None if x is None else [[int(w) for w in v] if isinstance(v, (tuple, list)) else int(v) for v in x]
    return {'chrom': r.chrom, 'pos': int(r.pos), 'end': int(r.end), 'is_rev': bool(r.is_rev),
            'pos_aln_index': aln(r.pos_aln_index), 'end_aln_index': aln(r.end_aln_index)}


def load_case(ctx, case):
    d = os.path.join(GOLD, case)
    ctx._inv_loaded = None
    pavinv.ensure_sequences(ctx, os.path.join(d, 'ref.fa'), os.path.join(d, 'tig.fa'))
    lift = AlignLift(pd.read_csv(os.path.join(d, 'align.tsv'), sep='\t'), read_fai(os.path.join(d, 'tig.fa.fai')))
    with open(os.path.join(d, 'scans.json')) as fh:
        scans = json.load(fh)
    return d, lift, scans


def check_call(d, rec, call):
    g = rec['call']
    assert call is not None and call.id == g['id'] and call.svlen == g['svlen']
    for name in ('region_ref_outer', 'region_ref_inner', 'region_tig_outer', 'region_tig_inner', 'region_ref_discovery',
                 'region_tig_discovery'):
        assert region_dict(getattr(call, name)) == g[name], name
    row = rules.inv_bed_row(call, 'h1', rec['flag']['type'], os.path.join(d, 'tig.fa'))
    got = {k: (int(v) if isinstance(v, (int, np.integer)) else v) for k, v in row.items()}
    assert got == g['bed_row']
    t = np.load(os.path.join(d, 'density_%s.npz' % g['id']))
    df = call.df
    assert list(df.columns) == ['INDEX', 'STATE_MER', 'STATE', 'KERN_FWD', 'KERN_FWDREV', 'KERN_REV', 'KMER', 'FLANK', 'MATCH']
    assert np.array_equal(df['INDEX'].to_numpy(), t['INDEX'])
    assert np.array_equal(df['STATE_MER'].to_numpy(), t['STATE_MER'])
    assert np.array_equal(df['STATE'].to_numpy(), t['STATE'])
    assert np.array_equal(df['KMER'].to_numpy(dtype=np.uint64), t['KMER'])
    assert np.array_equal(df['FLANK'].to_numpy(dtype=str), t['FLANK'])
    assert np.array_equal(df['MATCH'].fillna('NA').to_numpy(dtype=str), t['MATCH'])
    for c in KERN:
        assert np.allclose(df[c].to_numpy(), t[c], rtol=RTOL, atol=1e-300), c
    # text the rule writes: integer columns byte-identical; float columns printed by the same pandas formatter
    buf = io.StringIO()
    df.to_csv(buf, sep='\t', index=False)
    head = buf.getvalue().splitlines()[:6]
    assert head[0] == g['density_tsv_head'][0]
    for a, b in zip(head[1:], g['density_tsv_head'][1:]):
        fa, fb = a.split('\t'), b.split('\t')
        assert [fa[i] for i in (0, 1, 2, 6, 7, 8)] == [fb[i] for i in (0, 1, 2, 6, 7, 8)]


@pytest.mark.parametrize('case', INV_CASES)
def test_scan_for_inv_vs_reference(built, gpu_ctx, case, capsys):
    """Every flagged region of the golden cases: same log lines, same None / InvCall, same regions, BED row and table."""
    d, lift, scans = load_case(gpu_ctx, case)
    k_util = KmerUtil(util.case_k(d))
    for rec in scans:
        f = rec['flag']
        log = io.StringIO()
        call = pavinv.scan_for_inv(pavseq.Region(f['chrom'], f['pos'], f['end']), os.path.join(d, 'ref.fa'),
                                   os.path.join(d, 'tig.fa'), lift, k_util, log=log, ctx=gpu_ctx, **rec['kwargs'])
        assert log.getvalue().splitlines() == rec['log'], f
        if rec['call'] is None:
            assert call is None
        else:
            check_call(d, rec, call)


MODES = [(_lib.KDE_RUNS, _lib.KMER_LDS), (_lib.KDE_DIRECT, _lib.KMER_LDS), (_lib.KDE_RUNS, _lib.KMER_HBM)]
MODE_IDS = ['runs-lds', 'direct-lds', 'runs-hbm']


# ---- large regions (tests/golden/inv_large; 0.1 - 1.2 Mbp vs the oracle) --------------------------------------------------------

def check_large_call(d, rec, call):
    """A call of tests/golden/inv_large against the reference's digests: regions, BED row (SEQ by sha1), integer columns by
    sha1, KERN_* at the committed sample of rows."""
    g = rec['call']
    assert call is not None and call.id == g['id'] and call.svlen == g['svlen']
    for name in ('region_ref_outer', 'region_ref_inner', 'region_tig_outer', 'region_tig_inner', 'region_ref_discovery',
                 'region_tig_discovery'):
        assert region_dict(getattr(call, name)) == g[name], name
    row = rules.inv_bed_row(call, 'h1', rec['flag']['type'], os.path.join(d, 'tig.fa'))
    got = {k: (int(v) if isinstance(v, (int, np.integer)) else v) for k, v in row.items()}
    seq = got.pop('SEQ')
    got['SEQ_sha1'], got['SEQ_len'] = hashlib.sha1(seq.encode()).hexdigest(), len(seq)
    assert got == g['bed_row']
    df = call.df
    assert df.shape[0] == g['n_rows']
    assert sha(df['INDEX'].to_numpy(dtype=np.int64)) == g['index_sha1']
    assert sha(df['STATE_MER'].to_numpy(dtype=np.int8)) == g['state_mer_sha1']
    assert sha(df['STATE'].to_numpy(dtype=np.int8)) == g['state_sha1']
    assert sha(df['KMER'].to_numpy(dtype=np.uint64)) == g['kmer_sha1']
    assert hashlib.sha1('\n'.join(df['FLANK'].tolist()).encode()).hexdigest() == g['flank_sha1']
    assert hashlib.sha1('\n'.join(df['MATCH'].fillna('NA').tolist()).encode()).hexdigest() == g['match_sha1']
    t = np.load(os.path.join(GOLD, 'inv_large', 'kern_%s.npz' % g['id']))
    for c in KERN:
        assert np.allclose(df[c].to_numpy()[t['rows']], t[c], rtol=RTOL, atol=1e-300), c


@pytest.mark.parametrize('native', [False, True], ids=['python-driver', 'native-driver'])
def test_large_regions_scan_vs_reference(built, gpu_ctx, native, capsys):
    """pavlib.inv.scan_for_inv itself on a 150 kb and a 200 kb inversion flagged INSIDE (tools/refharness/
    gen_golden_inv_large.py): two expansion rounds each, regions of 54 k -> 135 k -> 337 kbp (forward contig, inverted-repeat
    flanks) and 74 k -> 185 k ([REV, FWD]: the balanced expansion of inv.py:329-332) -> 462 kbp (reverse-complemented contig,
    an N run inside), 9 - 260 LDS partitions per k-mer set, multi-tile run sums.  Both scan drivers: the same log lines, calls,
    regions, BED rows, table digests; KERN_* to 1e-12 at the sampled rows."""
    d, ref, hap, gold = util.inv_large_case()
    gpu_ctx._inv_loaded = None
    lift = AlignLift(hap.df_trim, hap.tig_lengths)
    k_util = KmerUtil(31)
    flags = [pavseq.Region(r['flag']['chrom'], r['flag']['pos'], r['flag']['end']) for r in gold['scans']]
    logs = [io.StringIO() for _ in flags]
    out = pavinv.scan_for_inv_batch(flags, os.path.join(d, 'ref.fa'), os.path.join(d, 'tig.fa'), lift, k_util, logs=logs,
                                    ctx=gpu_ctx, native=native)
    n_calls = 0
    for rec, call, log in zip(gold['scans'], out, logs):
        assert not isinstance(call, RuntimeError), call
        assert log.getvalue().splitlines() == rec['log'], rec['flag']
        if rec['call'] is None:
            assert call is None
        else:
            check_large_call(d, rec, call)
            n_calls += 1
            if native:
                assert call.n_unresolved == 0
    assert n_calls == 2


@pytest.mark.parametrize('mode,kmer', MODES, ids=MODE_IDS)
def test_large_regions_density_iterations_vs_reference(built, gpu_ctx, mode, kmer):
    """pav_density_batch on every (region_ref, region_tig) pair the reference scanned in tests/golden/inv_large (up to 462 kbp),
    one batch, every kernel mode: row counts, INDEX / STATE_MER / STATE digests and rl_encoder runs exact, KERN_* sums to 1e-12."""
    d, ref, hap, gold = util.inv_large_case()
    gpu_ctx._inv_loaded = None
    pavinv.ensure_sequences(gpu_ctx, os.path.join(d, 'ref.fa'), os.path.join(d, 'tig.fa'))
    ref_i = {n: i for i, n in enumerate(gpu_ctx.seq_names(_lib.PAV_ROLE_REF))}
    tig_i = {n: i for i, n in enumerate(gpu_ctx.seq_names(_lib.PAV_ROLE_TIG))}
    its = [it for rec in gold['scans'] for it in rec['iterations'] if it['region_tig'] is not None]
    jobs = [_lib.DenJob(ref_i[it['region_ref']['chrom']], tig_i[it['region_tig']['chrom']], it['region_ref']['pos'],
                        it['region_ref']['end'], it['region_tig']['pos'], it['region_tig']['end'],
                        1 if it['region_tig']['is_rev'] else 0, 20) for it in its]
    assert max(j.ref_end - j.ref_pos for j in jobs) > 450_000
    res = gpu_ctx.density_batch(jobs, pavden.den_params(kde_mode=mode, kmer_mode=kmer))
    for j, (it, r) in enumerate(zip(its, res)):
        if 'n_rows' not in it:
            assert r.status == _lib.DEN_FAIL or r.n_rows == 0
            continue
        assert r.status == (_lib.DEN_OK if it['finalised'] else _lib.DEN_UNFINALISED) and r.n_rows == it['n_rows']
        cols = gpu_ctx.density_table(j, r.n_rows)
        assert sha(cols['INDEX']) == it['index_sha1'] and sha(cols['STATE_MER']) == it['state_mer_sha1']
        assert sha(cols['STATE']) == it['state_sha1']
        assert [list(x) for x in gpu_ctx.density_runs(j, r.n_runs)] == it['state_rl']
        assert r.n_unresolved == 0
        if it['finalised']:
            assert np.allclose([cols[c].sum() for c in KERN], it['kern_sum'], rtol=1e-12, atol=0)


_LARGE_ORACLE = {}
LARGE_REGIONS = [('chrA', 425_000, 525_000), ('chrA', 275_000, 675_000), ('chrA', 1_000, 999_000),
                 ('chrB', 550_000, 650_000), ('chrB', 400_000, 800_000), ('chrB', 50_000, 1_250_000)]


@pytest.mark.parametrize('mode,kmer', MODES, ids=MODE_IDS)
def test_density_vs_oracle_on_large_regions(built, gpu_ctx, mode, kmer):
    """Regions of 100 kbp, 400 kbp and 1.0 / 1.2 Mbp (MAX_REGION_SIZE, pavlib/inv.py:23) - forward contig (chrA, here with an N
    run in the reference AND one in the contig) and reverse-complemented contig (chrB, N run in the reference) - through
    pav_density_batch in every kernel mode against the scalar oracle (evaluation points spread over the host's cores, each
    summed in scipy's order): status, row count, INDEX / STATE_MER / STATE / KMER exact, run lists exact, bandwidths
    bit-identical, KERN_* to 1e-12 in EVERY row.  The oracle's tables are computed once for the three modes (~1 minute)."""
    from oracle import oracle
    d, ref0, hap0, gold = util.inv_large_case()
    names, tnames = ref0.names, hap0.tig_names
    seqs = {n: ref0.seqs[n].copy() for n in names}
    tseqs = {n: hap0.tig_seqs[n].copy() for n in tnames}
    lift = AlignLift(hap0.df_trim, hap0.tig_lengths)
    seqs['chrA'][300_000:304_000] = ord('N')                              # k-mers skipped on both sides (density.py:48-50)
    ta = lift.lift_region_to_qry(pavseq.Region('chrA', 600_000, 602_500))
    tseqs[ta.chrom][ta.pos:ta.end] = ord('n')
    gpu_ctx._inv_loaded = None
    gpu_ctx.seq_load(_lib.PAV_ROLE_REF, names, [seqs[n] for n in names])
    gpu_ctx.seq_load(_lib.PAV_ROLE_TIG, tnames, [tseqs[n] for n in tnames])
    ref_i, tig_i = {n: i for i, n in enumerate(names)}, {n: i for i, n in enumerate(tnames)}
    pairs = []
    for c, p, e in LARGE_REGIONS:
        r = pavseq.Region(c, p, e)
        pairs.append((r, lift.lift_region_to_qry(r)))
    jobs = [_lib.DenJob(ref_i[r.chrom], tig_i[t.chrom], r.pos, r.end, t.pos, t.end, 1 if t.is_rev else 0, 20) for r, t in pairs]
    res = gpu_ctx.density_batch(jobs, pavden.den_params(kde_mode=mode, kmer_mode=kmer))
    threads = util.usable_cpus()
    for j, ((r, t), g) in enumerate(zip(pairs, res)):
        if j not in _LARGE_ORACLE:
            _LARGE_ORACLE[j] = oracle.density(seqs[r.chrom][r.pos:r.end], tseqs[t.chrom][t.pos:t.end], t.is_rev, threads=threads)
        o = _LARGE_ORACLE[j]
        assert g.status == o['status'] == 0 and g.n_rows == o['n'] > 0.9 * (len(r) - 8_000), r
        cols = gpu_ctx.density_table(j, g.n_rows)
        for c in ('INDEX', 'STATE_MER', 'STATE', 'KMER'):
            assert np.array_equal(cols[c], o[c]), (c, r)
        assert gpu_ctx.density_runs(j, g.n_runs) == oracle.rl_encode(o['STATE'], o['INDEX'])
        assert g.n_eval == o['n_eval'] and g.n_unresolved == 0
        assert np.allclose(list(g.h), o['h'], rtol=0, atol=0)
        for c in KERN:
            assert np.allclose(cols[c], o[c], rtol=RTOL, atol=1e-300), (c, r)
    assert len(pairs[5][0]) == 1_200_000 and res[5].n_rows > 1_100_000



@pytest.mark.parametrize('mode,kmer', MODES, ids=MODE_IDS)
@pytest.mark.parametrize('case', INV_CASES)
def test_density_iterations_vs_reference(built, gpu_ctx, case, mode, kmer):
    """pav_density_batch on every (region_ref, region_tig) pair the reference scanned, all in one batch: row counts,
    INDEX / STATE_MER / STATE digests and rl_encoder runs exact, density column sums to 1e-12."""
    d, lift, scans = load_case(gpu_ctx, case)
    ref_i = {n: i for i, n in enumerate(gpu_ctx.seq_names(_lib.PAV_ROLE_REF))}
    tig_i = {n: i for i, n in enumerate(gpu_ctx.seq_names(_lib.PAV_ROLE_TIG))}
    its = [it for rec in scans for it in rec['iterations'] if it['region_tig'] is not None]
    jobs = [_lib.DenJob(ref_i[it['region_ref']['chrom']], tig_i[it['region_tig']['chrom']], it['region_ref']['pos'],
                        it['region_ref']['end'], it['region_tig']['pos'], it['region_tig']['end'],
                        1 if it['region_tig']['is_rev'] else 0, 20) for it in its]
    if not jobs:
        pytest.skip('no liftable iteration in this case')
    res = gpu_ctx.density_batch(jobs, pavden.den_params(k=util.case_k(d), kde_mode=mode, kmer_mode=kmer))
    for j, (it, r) in enumerate(zip(its, res)):
        if 'n_rows' not in it:
            assert r.status == _lib.DEN_FAIL
            continue
        assert r.status == (_lib.DEN_OK if it['finalised'] else _lib.DEN_UNFINALISED)
        assert r.n_rows == it['n_rows']
        cols = gpu_ctx.density_table(j, r.n_rows)
        assert sha(cols['INDEX']) == it['index_sha1'] and sha(cols['STATE_MER']) == it['state_mer_sha1']
        assert sha(cols['STATE']) == it['state_sha1']
        assert [list(x) for x in gpu_ctx.density_runs(j, r.n_runs)] == it['state_rl']
        if it['finalised']:
            assert np.allclose([cols[c].sum() for c in KERN], it['kern_sum'], rtol=1e-12, atol=0)


def test_alignlift_device_tables(built, gpu_ctx):
    """AlignLift with lift tables built by pav_align_index (device tokenizer + scan) against the reference's answers."""
    with open(os.path.join(GOLD, 'lift_kat.json')) as fh:
        kat = json.load(fh)
    lifts = {}
    for k in kat:
        c = k['case']
        if c not in lifts:
            d = os.path.join(GOLD, c)
            lifts[c] = AlignLift(pd.read_csv(os.path.join(d, 'align.tsv'), sep='\t'), read_fai(os.path.join(d, 'tig.fa.fai')),
                                 ctx=gpu_ctx)
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
    # device tables == host tokenizer tables on a table with many rows, reverse rows, clips
    hap = synth.config2(seed=41, scale=0.003, threads=2)
    dev = AlignLift(hap.df_trim, hap.tig_lengths, ctx=gpu_ctx)
    host = AlignLift(hap.df_trim, hap.tig_lengths)
    for index in list(hap.df_trim.index)[:40]:
        dev._add_align(index)
        host._add_align(index)
        for ax in (0, 1):
            a, b = (dev.ref_cache[index], host.ref_cache[index]) if ax == 0 else (dev.tig_cache[index], host.tig_cache[index])
            assert np.array_equal(a.t.code, b.t.code) and np.array_equal(a.t.len, b.t.len)
            assert np.array_equal(a.t.begin[0], b.t.begin[0]) and np.array_equal(a.t.begin[1], b.t.begin[1])


def test_scan_with_hbm_tables_vs_reference(built, gpu_ctx, monkeypatch):
    """The whole scan with the k-mer sets forced into HBM tables (PAV_KMER_HBM): same logs (incl. the 'K-mer count exceeds
    max' failure, whose count and k-mer come from the HBM table in both modes) and calls as the reference."""
    monkeypatch.setenv('PAV_KMER_HBM', '1')
    d, lift, scans = load_case(gpu_ctx, 'inv_small')
    k_util = KmerUtil(31)
    assert any('K-mer count exceeds max' in ln for rec in scans for ln in rec['log'])
    for rec in scans:
        f = rec['flag']
        log = io.StringIO()
        call = pavinv.scan_for_inv(pavseq.Region(f['chrom'], f['pos'], f['end']), os.path.join(d, 'ref.fa'),
                                   os.path.join(d, 'tig.fa'), lift, k_util, log=log, ctx=gpu_ctx, **rec['kwargs'])
        assert log.getvalue().splitlines() == rec['log'], f
        if rec['call'] is None:
            assert call is None
        else:
            check_call(d, rec, call)


def test_kmer_sets_lds_equal_hbm_on_large_regions(built, gpu_ctx):
    """Regions of 60 kbp - 1.2 Mbp (9 - 170 LDS partitions per region), forward and reverse, with an inverted segment, a
    tandem array above the count limit and N runs: the LDS-partitioned sets and the HBM tables give identical tables."""
    rng = np.random.default_rng(77)
    n = 2_600_000
    ref = rng.integers(0, 4, n, dtype=np.uint8)
    acgt = np.frombuffer(b'ACGT', dtype=np.uint8)
    tig = ref.copy()
    snv = rng.random(n) < 0.002
    tig[snv] = (tig[snv] + rng.integers(1, 4, int(snv.sum()), dtype=np.uint8)) & 3
    tig[400_000:520_000] = 3 - tig[400_000:520_000][::-1]                      # inversion
    ref_a, tig_a = acgt[ref].copy(), acgt[tig].copy()
    ref_a[2_000_000:2_000_500] = ord('N')
    tig_a[700_000:700_040] = ord('n')
    unit = acgt[rng.integers(0, 4, 37, dtype=np.uint8)]
    ref_a[2_300_000:2_300_000 + 37 * 160] = np.tile(unit, 160)                  # 160 copies > MAX_REF_KMER_COUNT
    gpu_ctx._inv_loaded = None
    gpu_ctx.seq_load(_lib.PAV_ROLE_REF, ['chrL'], [ref_a])
    gpu_ctx.seq_load(_lib.PAV_ROLE_TIG, ['tigL'], [tig_a])
    spans = [(300_000, 360_000, 0), (350_000, 650_000, 0), (0, 1_200_000, 0), (380_000, 540_000, 1),
             (1_900_000, 2_100_000, 0), (2_250_000, 2_350_000, 0), (690_000, 710_000, 0)]
    jobs = [_lib.DenJob(0, 0, a, b, a, b, rc, 20) for a, b, rc in spans]
    out = {}
    for kmer in (_lib.KMER_LDS, _lib.KMER_HBM):
        res = gpu_ctx.density_batch(jobs, pavden.den_params(kmer_mode=kmer))
        out[kmer] = [(r.status, r.fail_kind, r.n_rows, r.max_count, r.max_kmer, list(r.state_count),
                      gpu_ctx.density_table(j, r.n_rows) if r.status != _lib.DEN_FAIL else None,
                      gpu_ctx.density_runs(j, r.n_runs) if r.status != _lib.DEN_FAIL else None) for j, r in enumerate(res)]
    statuses = [o[0] for o in out[_lib.KMER_LDS]]
    assert statuses.count(_lib.DEN_OK) >= 5 and statuses[5] == _lib.DEN_FAIL and out[_lib.KMER_LDS][5][3] == 160
    for a, b in zip(out[_lib.KMER_LDS], out[_lib.KMER_HBM]):
        assert a[:6] == b[:6]
        if a[6] is not None:
            assert a[7] == b[7]
            for c in a[6]:
                assert np.array_equal(a[6][c], b[6][c]), c
    inv_rows = out[_lib.KMER_LDS][1][6]
    assert (inv_rows['STATE_MER'] == 2).sum() > 100_000                           # the inverted segment shows up as REV

    # Low-complexity sequence sends far more k-mers to one partition than its list holds: the batch is redone with HBM
    # tables (same answers).  Job 0: 20 kbp of poly-A on the contig only; job 1: on the reference too (count gate fails).
    ref_b, tig_b = ref_a.copy(), tig_a.copy()
    tig_b[1_300_000:1_320_000] = ord('A')
    ref_b[1_500_000:1_520_000] = ord('a')
    tig_b[1_500_000:1_520_000] = ord('A')
    gpu_ctx.seq_load(_lib.PAV_ROLE_REF, ['chrL'], [ref_b])
    gpu_ctx.seq_load(_lib.PAV_ROLE_TIG, ['tigL'], [tig_b])
    jobs = [_lib.DenJob(0, 0, 1_280_000, 1_340_000, 1_280_000, 1_340_000, 0, 20),
            _lib.DenJob(0, 0, 1_480_000, 1_540_000, 1_480_000, 1_540_000, 0, 20),
            _lib.DenJob(0, 0, 300_000, 360_000, 300_000, 360_000, 0, 20)]
    got = {}
    for kmer in (_lib.KMER_LDS, _lib.KMER_HBM):
        res = gpu_ctx.density_batch(jobs, pavden.den_params(kmer_mode=kmer))
        got[kmer] = [(r.status, r.fail_kind, r.n_rows, r.max_count, r.max_kmer,
                      gpu_ctx.density_table(j, r.n_rows)['STATE_MER'].tobytes() if r.status != _lib.DEN_FAIL else None)
                     for j, r in enumerate(res)]
    assert got[_lib.KMER_LDS] == got[_lib.KMER_HBM]
    assert got[_lib.KMER_LDS][0][0] == _lib.DEN_OK and got[_lib.KMER_LDS][2][0] == _lib.DEN_OK
    assert got[_lib.KMER_LDS][1][:2] == (_lib.DEN_FAIL, 2) and 19_970 <= got[_lib.KMER_LDS][1][3] < 19_990 and got[_lib.KMER_LDS][1][4] == 0


@pytest.mark.parametrize('native', [True, False])
@pytest.mark.parametrize('case', INV_CASES)
def test_batched_scan_vs_reference(built, gpu_ctx, case, native):
    """scan_for_inv_batch (native C++ driver in the library, and the Python state machine) on every flagged region of
    every golden case, grouped by keyword arguments: logs, None / InvCall, regions, BED rows and tables equal the
    reference's sequential results."""
    d, lift, scans = load_case(gpu_ctx, case)
    k_util = KmerUtil(util.case_k(d))
    groups = {}
    for rec in scans:
        groups.setdefault(json.dumps(rec['kwargs'], sort_keys=True), []).append(rec)
    for key, recs in groups.items():
        kwargs = json.loads(key)
        regions = [pavseq.Region(r['flag']['chrom'], r['flag']['pos'], r['flag']['end']) for r in recs]
        logs = [io.StringIO() for _ in regions]
        out = pavinv.scan_for_inv_batch(regions, os.path.join(d, 'ref.fa'), os.path.join(d, 'tig.fa'), lift, k_util, logs=logs,
                                        ctx=gpu_ctx, native=native, eager_tables=(len(recs) % 2 == 0), **kwargs)
        for rec, call, lg in zip(recs, out, logs):
            assert lg.getvalue().splitlines() == rec['log'], (rec['flag'], native)
            if rec['call'] is None:
                assert call is None
            else:
                check_call(d, rec, call)


def test_scan_with_helper_threads_gives_the_reference_results(built, monkeypatch):
    """PAV_HOST_THREADS = 4: the per-region loops of the native driver (lift results -> regions, job descriptors, decisions, the
    texts written while a round's kernels run) are shared with three helper threads (csrc/pool.h).  Logs and calls equal the
    reference's, as with the caller's thread alone."""
    monkeypatch.setenv('PAV_HOST_THREADS', '4')
    with _lib.Context(0) as ctx:                                      # a context of its own: the pool is made with the first index
        for case in ('inv_hap', 'inv_rev', 'inv_nolift'):
            d, lift, scans = load_case(ctx, case)
            recs = [r for r in scans if not r['kwargs']]
            regions = [pavseq.Region(r['flag']['chrom'], r['flag']['pos'], r['flag']['end']) for r in recs]
            for rep in range(3):                                      # (threads: more than one go)
                logs = [io.StringIO() for _ in regions]
                out = pavinv.scan_for_inv_batch(regions, os.path.join(d, 'ref.fa'), os.path.join(d, 'tig.fa'), lift, KmerUtil(31), logs=logs,
                                                ctx=ctx, native=True)
                for rec, call, lg in zip(recs, out, logs):
                    assert lg.getvalue().splitlines() == rec['log'], (case, rec['flag'], rep)
                    if rec['call'] is None:
                        assert call is None
                    else:
                        check_call(d, rec, call)


@pytest.mark.parametrize('budget', ['1', '30000', '120000'])
def test_small_batch_budget_gives_the_reference_results(built, gpu_ctx, monkeypatch, budget):
    """The native driver takes the regions of a round in batches of at most 64 Mbp (reference + contig bases); the regions behind
    the budget wait, are lifted again with the next batch's regions and scanned then.  PAV_SCAN_BATCH_BP shrinks the budget so
    that the golden cases run that path - one region per batch, a few, most: logs and calls equal the reference's."""
    monkeypatch.setenv('PAV_SCAN_BATCH_BP', budget)
    for case in ('inv_hap', 'inv_rev'):
        d, lift, scans = load_case(gpu_ctx, case)
        recs = [r for r in scans if not r['kwargs']]
        regions = [pavseq.Region(r['flag']['chrom'], r['flag']['pos'], r['flag']['end']) for r in recs]
        logs = [io.StringIO() for _ in regions]
        out = pavinv.scan_for_inv_batch(regions, os.path.join(d, 'ref.fa'), os.path.join(d, 'tig.fa'), lift, KmerUtil(31), logs=logs,
                                        ctx=gpu_ctx, native=True)
        assert len(recs) >= 2
        for rec, call, lg in zip(recs, out, logs):
            assert lg.getvalue().splitlines() == rec['log'], (case, rec['flag'], budget)
            if rec['call'] is None:
                assert call is None
            else:
                check_call(d, rec, call)


@pytest.mark.parametrize('case', ['inv_hap', 'inv_nolift', 'inv_rev'])
def test_device_lifts_equal_the_host_tables(built, gpu_ctx, case, monkeypatch, capfd):
    """The native driver lifts the ends of every region and the breakpoints of every flanked region in batches on the device
    (csrc/lift_dev.hip: the operation tables never leave HBM); PAV_LIFT_HOST=1 keeps the host lookup tables of round 3.  Same
    logs, same calls, same regions - and both equal the reference's (test_batched_scan_vs_reference runs the default)."""
    d, lift, scans = load_case(gpu_ctx, case)
    flags = [pavseq.Region(r['flag']['chrom'], r['flag']['pos'], r['flag']['end']) for r in scans if not r['kwargs']]
    got = {}
    for name, env in (('device', None), ('host', '1')):
        if env is None:
            monkeypatch.delenv('PAV_LIFT_HOST', raising=False)
        else:
            monkeypatch.setenv('PAV_LIFT_HOST', env)
        lift._native_loaded = None                                   # the index is built again, on the other side
        logs = [io.StringIO() for _ in flags]
        out = pavinv.scan_for_inv_batch(flags, os.path.join(d, 'ref.fa'), os.path.join(d, 'tig.fa'), lift, KmerUtil(31), logs=logs,
                                        ctx=gpu_ctx, native=True)
        got[name] = ([lg.getvalue() for lg in logs],
                     [None if c is None else (str(c) if isinstance(c, RuntimeError) else
                                              (c.id, [region_dict(getattr(c, n)) for n in ('region_ref_outer', 'region_ref_inner',
                                                                                            'region_tig_outer', 'region_tig_inner')]))
                      for c in out])
    assert got['device'] == got['host']
    want = [r['log'] for r in scans if not r['kwargs']]
    assert [t.splitlines() for t in got['device'][0]] == want


def test_lift_errors_equal_on_every_path(built, gpu_ctx, monkeypatch, tmp_path):
    """A record with an N operation: AlignLift refuses it when it is first used (lift.py:463-471: 'Unhandled CIGAR operation: N:
    Alignment chrom:pos (contig)').  scan_for_inv raises that RuntimeError; the batch drivers hand it back per region.  The device
    lifts (an error code + the record, formatted on the host), the host tables and the Python driver's AlignLift give the same text."""
    d, lift0, scans = load_case(gpu_ctx, 'inv_fwd')
    df = pd.read_csv(os.path.join(d, 'align.tsv'), sep='\t')
    cig = df.loc[0, 'CIGAR']
    m = __import__('re').search(r'(\d+)=', cig)
    n = int(m.group(1))
    assert n > 20
    df.loc[0, 'CIGAR'] = cig[:m.start()] + f'{n - 10}=10N' + cig[m.end():]
    flag = pavseq.Region(scans[1]['flag']['chrom'], scans[1]['flag']['pos'], scans[1]['flag']['end'])
    texts = {}
    for name, env, native in (('device', None, True), ('host', '1', True), ('python', None, False)):
        if env is None:
            monkeypatch.delenv('PAV_LIFT_HOST', raising=False)
        else:
            monkeypatch.setenv('PAV_LIFT_HOST', env)
        lift = AlignLift(df, read_fai(os.path.join(d, 'tig.fa.fai')))
        out = pavinv.scan_for_inv_batch([flag], os.path.join(d, 'ref.fa'), os.path.join(d, 'tig.fa'), lift, KmerUtil(31), ctx=gpu_ctx,
                                        native=native)
        assert isinstance(out[0], RuntimeError), (name, out[0])
        texts[name] = str(out[0])
    assert texts['device'] == texts['host'] == texts['python']
    row = df.iloc[0]
    assert texts['device'] == 'Unhandled CIGAR operation: N: Alignment {}:{} ({})'.format(row['#CHROM'], row['POS'], row['QRY_ID'])


@pytest.mark.parametrize('native', [True, False])
def test_batch_log_sink_equals_the_sequential_log(built, gpu_ctx, native):
    """``log=``: one file-like object for the whole batch, as rule call_inv_batch hands its log file to every scan_for_inv
    call (rules/call_inv.snakefile:172-196): the text is the reference's per-region logs in region order."""
    d, lift, scans = load_case(gpu_ctx, 'inv_hap')
    recs = [r for r in scans if not r['kwargs']]
    regions = [pavseq.Region(r['flag']['chrom'], r['flag']['pos'], r['flag']['end']) for r in recs]
    log = io.StringIO()
    out = pavinv.scan_for_inv_batch(regions, os.path.join(d, 'ref.fa'), os.path.join(d, 'tig.fa'), lift, KmerUtil(31), log=log,
                                    ctx=gpu_ctx, native=native)
    assert log.getvalue().splitlines() == [ln for r in recs for ln in r['log']]
    assert [o is not None for o in out] == [r['call'] is not None for r in recs]


def test_lazy_table_expires_with_the_next_scan(built, gpu_ctx):
    d, lift, scans = load_case(gpu_ctx, 'inv_small')
    k_util = KmerUtil(31)
    rec = [r for r in scans if r['call'] is not None][0]
    region = pavseq.Region(rec['flag']['chrom'], rec['flag']['pos'], rec['flag']['end'])
    args = (os.path.join(d, 'ref.fa'), os.path.join(d, 'tig.fa'), lift, k_util)
    first = pavinv.scan_for_inv_batch([region], *args, ctx=gpu_ctx, eager_tables=False)[0]
    second = pavinv.scan_for_inv_batch([region], *args, ctx=gpu_ctx, eager_tables=False)[0]
    check_call(d, rec, second)
    with pytest.raises(_lib.PavDeviceError):
        first.df


def test_rule_call_inv_batch_haplotype(built, gpu_ctx, tmp_path):
    """Multi-row haplotype (reverse rows, inverted repeats, N run, decoys) through the rule mirror in lock-step batches:
    the merged INV BED equals the reference's rows and the log equals the sequential reference log."""
    d, lift, scans = load_case(gpu_ctx, 'inv_hap')
    beds, logs = [], []
    for batch in (0, 1):
        out, lg = str(tmp_path / f'inv_call_{batch}.bed.gz'), str(tmp_path / f'inv_call_{batch}.log')
        rules.call_inv_batch(os.path.join(d, 'flag.tsv'), os.path.join(d, 'align.tsv'), os.path.join(d, 'tig.fa'),
                             os.path.join(d, 'tig.fa.fai'), os.path.join(d, 'ref.fa'), 'h1', batch, bed_out=out, log_path=lg,
                             density_out_dir=str(tmp_path / 'density'), ctx=gpu_ctx)
        beds.append(out)
        logs.append(lg)
    df = rules.call_inv_batch_merge(beds)
    expect = [rec['call']['bed_row'] for rec in scans if rec['call'] is not None]
    assert sorted(df['ID']) == sorted(r['ID'] for r in expect)
    got = {r['ID']: r for _, r in df.iterrows()}
    for row in expect:
        for k, v in row.items():
            assert str(got[row['ID']][k]) == str(v), (row['ID'], k)
    flag = pd.read_csv(os.path.join(d, 'flag.tsv'), sep='\t')
    for batch in (0, 1):
        want = []
        for (_, f), rec in zip(flag.iterrows(), scans):
            if f['BATCH'] == batch:
                want.extend(rec['log'])
        with open(logs[batch]) as fh:
            assert fh.read().splitlines() == want


def test_rule_outputs_equal_the_reference_rule(built, gpu_ctx, tmp_path):
    """rules.call_inv_batch x 2 + call_inv_batch_merge vs the files the reference's own rule bodies wrote
    (tests/golden/rule_call_inv_batch: rules/call_inv.snakefile:94-311 executed unmodified): merged INV BED and both
    logs byte-identical; every density table file has the same rows / columns and identical non-float fields."""
    import gzip
    d, lift, scans = load_case(gpu_ctx, 'inv_hap')
    g = os.path.join(GOLD, 'rule_call_inv_batch')
    beds = []
    for batch in (0, 1):
        out, lg = str(tmp_path / f'inv_call_{batch}.bed.gz'), str(tmp_path / f'inv_call_{batch}.log')
        rules.call_inv_batch(os.path.join(d, 'flag.tsv'), os.path.join(d, 'align.tsv'), os.path.join(d, 'tig.fa'),
                             os.path.join(d, 'tig.fa.fai'), os.path.join(d, 'ref.fa'), 'h1', batch, bed_out=out, log_path=lg,
                             density_out_dir=str(tmp_path / 'density'), ctx=gpu_ctx)
        beds.append(out)
        with open(lg) as fh, open(os.path.join(g, f'log_{batch}.txt')) as gh:
            assert fh.read() == gh.read()
    merged = str(tmp_path / 'sv_inv.bed.gz')
    rules.call_inv_batch_merge(beds, merged)
    with gzip.open(merged, 'rt') as fh, open(os.path.join(g, 'inv_merged.tsv')) as gh:
        assert fh.read() == gh.read()
    with open(os.path.join(g, 'density_index.json')) as fh:
        index = json.load(fh)
    assert sorted(os.listdir(tmp_path / 'density')) == sorted(index)
    for name, meta in index.items():
        with gzip.open(tmp_path / 'density' / name, 'rt') as fh:
            lines = fh.read().splitlines()
        assert len(lines) - 1 == meta['rows'] and lines[0].split('\t') == meta['columns']
        for a, b in zip(lines[1:3], meta['head'][1:3]):
            fa, fb = a.split('\t'), b.split('\t')
            assert [fa[i] for i in (0, 1, 2, 6, 7, 8)] == [fb[i] for i in (0, 1, 2, 6, 7, 8)]
            assert np.allclose([float(fa[i]) for i in (3, 4, 5)], [float(fb[i]) for i in (3, 4, 5)], rtol=RTOL, atol=1e-300)


def test_native_density_tables_equal_pandas_text(built, gpu_ctx, tmp_path):
    """pav_inv_write_tables (what rule call_inv_batch's density_*.tsv.gz files are written with) against
    call.df.to_csv(sep='\\t', index=False) of the same calls: identical text, plain and gzip, one and many threads."""
    import gzip
    d, lift, scans = load_case(gpu_ctx, 'inv_hap')
    regions = [pavseq.Region(r['flag']['chrom'], r['flag']['pos'], r['flag']['end']) for r in scans]
    out = pavinv.scan_for_inv_batch(regions, os.path.join(d, 'ref.fa'), os.path.join(d, 'tig.fa'), lift, KmerUtil(31), ctx=gpu_ctx,
                                    eager_tables=False)
    calls = [(i, c) for i, c in enumerate(out) if c is not None and not isinstance(c, RuntimeError)]
    assert len(calls) >= 3 and all(c.native_table[1] == i for i, c in calls)
    plain = [str(tmp_path / f'{c.id}.tsv') for _, c in calls]
    gz = [str(tmp_path / f'{c.id}.tsv.gz') for _, c in calls]
    gpu_ctx.inv_write_tables([i for i, _ in calls], plain, threads=1)
    gpu_ctx.inv_write_tables([i for i, _ in calls], gz, threads=5, gzip_level=1)
    for (i, c), p, g in zip(calls, plain, gz):
        want = c.df.to_csv(sep='\t', index=False)
        with open(p) as fh:
            assert fh.read() == want, c.id
        with gzip.open(g, 'rt') as fh:
            assert fh.read() == want, c.id
    flank = pd.concat([c.df['FLANK'] for _, c in calls])
    match = pd.concat([c.df['MATCH'] for _, c in calls])
    assert {'', 'UP', 'DN'} <= set(flank) and (match == 'OTHER').any() and match.isna().any()     # every text form occurs
    with pytest.raises(_lib.PavDeviceError, match='no call'):
        gpu_ctx.inv_write_tables([next(i for i, c in enumerate(out) if c is None)], [str(tmp_path / 'none.tsv')])


def test_rule_call_inv_batch_files(built, gpu_ctx, tmp_path):
    """File contract of rule call_inv_batch: INV BED rows equal the reference rows; density tables are written."""
    d, lift, scans = load_case(gpu_ctx, 'inv_fwd')
    beds = []
    for batch in (0, 1):
        out = str(tmp_path / f'inv_call_{batch}.bed.gz')
        rules.call_inv_batch(os.path.join(d, 'flag.tsv'), os.path.join(d, 'align.tsv'), os.path.join(d, 'tig.fa'),
                             os.path.join(d, 'tig.fa.fai'), os.path.join(d, 'ref.fa'), 'h1', batch, bed_out=out,
                             log_path=str(tmp_path / f'inv_call_{batch}.log'), density_out_dir=str(tmp_path / 'density'),
                             ctx=gpu_ctx)
        beds.append(out)
    df = rules.call_inv_batch_merge(beds)
    # flag.tsv holds every flagged region of the case; kwargs variants (region limit, min_exp_count) use defaults here
    expect = {}
    flag = pd.read_csv(os.path.join(d, 'flag.tsv'), sep='\t')
    for rec in scans:
        if rec['call'] is not None and not rec['kwargs']:
            expect[rec['call']['id']] = rec['call']['bed_row']
    got = {r['ID']: r for _, r in df.iterrows()}
    for vid, row in expect.items():
        assert vid in got
        for k, v in row.items():
            assert str(got[vid][k]) == str(v), (vid, k)
        assert os.path.exists(tmp_path / 'density' / f'density_{vid}_h1.tsv.gz')
    assert flag.shape[0] == len(scans)


@pytest.mark.parametrize('mode,kmer', MODES, ids=MODE_IDS)
@pytest.mark.parametrize('seed', [31, 32])
def test_density_vs_oracle_seeded(built, gpu_ctx, seed, mode, kmer):
    """Seeded haplotype with planted inversions, inverted repeats, N runs and reverse rows: the first scan iteration of
    every flagged region as one device batch vs the scalar oracle - integer columns exact, KERN_* to 1e-12."""
    from oracle import oracle
    hap = synth.config2(seed=seed, scale=0.004, threads=4)
    names = hap.ref.names
    gpu_ctx._inv_loaded = None
    gpu_ctx.seq_load(_lib.PAV_ROLE_REF, names, [hap.ref.seqs[n] for n in names])
    gpu_ctx.seq_load(_lib.PAV_ROLE_TIG, hap.tig_names, [hap.tig_seqs[n] for n in hap.tig_names])
    lift = AlignLift(hap.df_trim, hap.tig_lengths)
    fai = pd.Series(hap.ref.lengths)
    ref_i = {n: i for i, n in enumerate(names)}
    tig_i = {n: i for i, n in enumerate(hap.tig_names)}
    jobs, pairs = [], []
    for _, row in hap.df_flag.iterrows():
        r = pavseq.Region(row['#CHROM'], row['POS'], row['END'])
        r.expand(4000, min_pos=0, max_end=fai, shift=True)
        try:
            t = lift.lift_region_to_qry(r)
        except RuntimeError:
            t = None
        if t is None:
            continue
        jobs.append(_lib.DenJob(ref_i[r.chrom], tig_i[t.chrom], r.pos, r.end, t.pos, t.end, 1 if t.is_rev else 0, 20))
        pairs.append((r, t))
        if len(jobs) >= 24:
            break
    assert len(jobs) >= 8
    res = gpu_ctx.density_batch(jobs, pavden.den_params(kde_mode=mode, kmer_mode=kmer))
    n_final = 0
    for j, ((r, t), g) in enumerate(zip(pairs, res)):
        o = oracle.density(hap.ref.seqs[r.chrom][r.pos:r.end], hap.tig_seqs[t.chrom][t.pos:t.end], t.is_rev,
                           threads=util.usable_cpus() if len(r) > 30_000 else 1)
        assert g.status == o['status'], (r, t)
        if o['status'] == 125:
            assert g.fail_kind == o['fail_kind']
            continue
        assert g.n_rows == o['n']
        cols = gpu_ctx.density_table(j, g.n_rows)
        for c in ('INDEX', 'STATE_MER', 'STATE', 'KMER'):
            assert np.array_equal(cols[c], o[c]), c
        assert gpu_ctx.density_runs(j, g.n_runs) == oracle.rl_encode(o['STATE'], o['INDEX'])
        if o['status'] == 0:
            n_final += 1
            assert g.n_eval == o['n_eval']
            assert np.allclose(list(g.h), o['h'], rtol=0, atol=0)          # bandwidths are computed identically
            for c in KERN:
                assert np.allclose(cols[c], o[c], rtol=RTOL, atol=1e-300), c
    assert n_final >= 4


@pytest.mark.parametrize('k', [6, 12, 31])
@pytest.mark.parametrize('ref_rc', [False, True])
def test_kmer_states_for_even_k_and_palindromes(built, gpu_ctx, k, ref_rc):
    """The LDS k-mer sets are keyed by the canonical k-mer with one count per orientation.  For even k a k-mer can be its own
    reverse complement (both membership answers are then the same count); with k = 6 nearly every k-mer of a few kbp occurs in
    both orientations.  STATE_MER / INDEX / KMER and the per-state counts against the scalar oracle and against the HBM-table
    kernels (oriented keys, two probes: an independent implementation), with and without -r."""
    from oracle import oracle
    rng = np.random.default_rng(100 + k)
    lut = np.frombuffer(b'ACGT', dtype=np.uint8)
    n = 6000 if k > 6 else 700
    ref = lut[rng.integers(0, 4, n)]
    pal = np.frombuffer(b'ACGTACGTTGCATGCAGAATTCGGATCC' * 4, dtype=np.uint8)          # runs of self-reverse-complement words
    ref[100:100 + pal.shape[0]] = pal
    tig = ref.copy()
    cut = n // 3
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    tig[cut:2 * cut] = np.array([comp[int(b)] for b in tig[cut:2 * cut][::-1]], dtype=np.uint8)   # an inverted third
    tig[10:40] = lut[rng.integers(0, 4, 30)]
    gpu_ctx._inv_loaded = None
    gpu_ctx.seq_load(_lib.PAV_ROLE_REF, ['chrP'], [ref])
    gpu_ctx.seq_load(_lib.PAV_ROLE_TIG, ['tigP'], [tig])
    job = _lib.DenJob(0, 0, 0, n, 0, n, 1 if ref_rc else 0, 20)
    limit = 250
    o = oracle.density(ref, tig, ref_rc, oracle.den_params(k=k, min_informative=10, min_state_count=1, max_ref_kmer_count=limit))
    got = {}
    for kmer in (_lib.KMER_LDS, _lib.KMER_HBM):
        res = gpu_ctx.density_batch([job], pavden.den_params(k=k, kmer_mode=kmer, min_informative=10, min_state_count=1,
                                                             max_ref_kmer_count=limit))[0]
        assert res.status == o['status'] and res.n_rows == o['n'], (kmer, res.status, o['status'])
        cols = gpu_ctx.density_table(0, res.n_rows)
        got[kmer] = {c: cols[c].copy() for c in ('INDEX', 'STATE_MER', 'STATE', 'KMER')}
        for c in ('INDEX', 'STATE_MER', 'KMER'):
            assert np.array_equal(cols[c], o[c]), (kmer, c)
    assert all(np.array_equal(got[_lib.KMER_LDS][c], got[_lib.KMER_HBM][c]) for c in got[_lib.KMER_LDS])
    present = set(np.unique(o['STATE_MER']).tolist())
    assert {0, 2} <= present or {1} <= present                    # forward and inverted (or both-orientation) k-mers are there


@pytest.mark.parametrize('ref_rc', [False, True])
def test_k32_states_with_the_all_t_kmer(built, gpu_ctx, ref_rc):
    """inv_k_size = 32: a 32-mer fills the 64-bit word, its sets live in the HBM tables whatever kmer_mode says - and the 32-mer made
    of T is the word that marks a free table slot, so the library keeps its count aside.  A region with poly-T and poly-A tracts in
    the reference and the contig (forward third, inverted third, a tract that is only in the contig) against the scalar oracle:
    STATE_MER / INDEX / KMER / STATE exact, the tracts' rows present; and a low count limit: the all-T k-mer is the one the failure
    message names (scripts/density.py:519-526) when it is the most frequent.  k = 33 is refused."""
    from oracle import oracle
    rng = np.random.default_rng(3200 + int(ref_rc))
    lut = np.frombuffer(b'ACGT', dtype=np.uint8)
    n = 9000
    ref = lut[rng.integers(0, 4, n)]
    ref[500:560] = ord('T'); ref[1500:1545] = ord('A'); ref[4000:4040] = ord('T'); ref[7000:7050] = ord('A')
    tig = ref.copy()
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    tig[3000:6000] = np.array([comp[int(b)] for b in tig[3000:6000][::-1]], dtype=np.uint8)      # an inverted third (its poly-T reads poly-A)
    tig[8000:8040] = ord('T')                                                                     # a tract the reference region lacks here
    gpu_ctx._inv_loaded = None
    gpu_ctx.seq_load(_lib.PAV_ROLE_REF, ['chrK'], [ref])
    gpu_ctx.seq_load(_lib.PAV_ROLE_TIG, ['tigK'], [tig])
    job = _lib.DenJob(0, 0, 0, n, 0, n, 1 if ref_rc else 0, 20)
    ones = np.uint64(0xFFFFFFFFFFFFFFFF)
    for limit in (100, 20):
        o = oracle.density(ref, tig, ref_rc, oracle.den_params(k=32, min_informative=10, min_state_count=1, max_ref_kmer_count=limit))
        for kmer in (_lib.KMER_LDS, _lib.KMER_HBM):
            res = gpu_ctx.density_batch([job], pavden.den_params(k=32, kmer_mode=kmer, min_informative=10, min_state_count=1,
                                                                 max_ref_kmer_count=limit))[0]
            assert res.status == (_lib.DEN_FAIL if o['status'] == 125 else (_lib.DEN_OK if o['status'] == 0 else _lib.DEN_UNFINALISED)), (limit, kmer)
            if o['status'] == 125:
                assert (res.fail_kind, res.max_count, res.max_kmer) == (o['fail_kind'], o['max_count'], o['max_kmer']), (limit, kmer)
                continue
            assert res.n_rows == o['n']
            cols = gpu_ctx.density_table(0, res.n_rows)
            for c in ('INDEX', 'STATE_MER', 'STATE', 'KMER'):
                assert np.array_equal(cols[c], o[c]), (limit, kmer, c)
            assert int((cols['KMER'] == ones).sum()) > 20 and int((cols['KMER'] == np.uint64(0)).sum()) > 20
            df = pavden.table_frame(cols)
            assert df['KMER'].dtype == np.uint64                                     # what pandas makes of Python integers above 2^63
    # the 20-limit run fails on the all-T (or all-A) tract: 29 occurrences of one 32-mer in a 60-base tract
    o = oracle.density(ref, tig, ref_rc, oracle.den_params(k=32, min_informative=10, min_state_count=1, max_ref_kmer_count=20))
    assert o['status'] == 125 and o['max_kmer'] in (0, 0xFFFFFFFFFFFFFFFF)
    with pytest.raises(RuntimeError, match='k = 33'):
        gpu_ctx.density_batch([job], pavden.den_params(k=33))
    with pytest.raises(RuntimeError, match='1..32'):
        pavinv._check_k_size(KmerUtil(33))
    pavinv._check_k_size(KmerUtil(32))


NEARTIE = ['argmax_search', 'argmax_mirror', 'delta_above', 'delta_below']


def run_neartie(ctx, case, **kw):
    g = np.load(os.path.join(GOLD, 'den_neartie', case + '.npz'))
    p = json.loads(str(g['params']))
    ctx._inv_loaded = None
    ctx.seq_load(_lib.PAV_ROLE_REF, ['chrN'], [g['ref']])
    ctx.seq_load(_lib.PAV_ROLE_TIG, ['tigN'], [g['tig']])
    job = _lib.DenJob(0, 0, 0, g['ref'].shape[0], 0, g['tig'].shape[0], 0, p['staterunsmooth'])
    res = ctx.density_batch([job], pavden.den_params(k=p['k'], state_run_delta=p['staterundelta'], **kw))[0]
    assert res.status == _lib.DEN_OK and res.n_rows == g['INDEX'].shape[0]
    return g, p, res, ctx.density_table(0, res.n_rows)


@pytest.mark.parametrize('mode', [_lib.KDE_RUNS, _lib.KDE_DIRECT], ids=['runs', 'direct'])
@pytest.mark.parametrize('case', NEARTIE)
def test_constructed_near_ties_vs_reference(built, gpu_ctx, case, mode):
    """Inputs built so that scripts/density.py takes a float decision by a hair (tools/refharness/gen_golden_neartie.py; the
    tables are the reference's own): the arg-max of a row with a margin of 2e-11, an exact tie (1e-15: rounding decides in
    the reference too), and density_change of one window 1e-10 above / below --staterundelta.  The guard must see each of
    them, evaluate the sites they rest on in scipy's order and reproduce the reference's decision; only the exact tie is
    allowed to go either way, and is reported as unresolved."""
    g, p, res, cols = run_neartie(gpu_ctx, case, kde_mode=mode)
    assert np.array_equal(cols['INDEX'], g['INDEX']) and np.array_equal(cols['STATE_MER'], g['STATE_MER'])
    for c in KERN:
        assert np.allclose(cols[c], g[c], rtol=RTOL, atol=1e-300), c
    diff = np.flatnonzero(cols['STATE'] != g['STATE'])
    assert res.n_near_tie >= 1 and res.guard_fallback == 0
    assert res.n_spike_near > 100                    # KERN = 1.0 to the last bits inside a long run: counted, continuous
    if mode == _lib.KDE_RUNS:
        assert res.n_reeval >= (2 if case.startswith('delta') else 1)
    else:
        assert res.n_reeval == 0                     # every sum is in scipy's order already
    if case == 'argmax_mirror':
        assert set(diff) <= {p['row']} and cols['STATE'][p['row']] in (0, 2)
        assert res.n_unresolved >= 1
    else:
        assert diff.size == 0 and res.n_unresolved == 0
    if case.startswith('delta'):
        a, b = p['window']
        x = np.arange(a + 1, b)
        chord = cols['KERN_FWD'][a] + (cols['KERN_FWD'][b] - cols['KERN_FWD'][a]) / (b - a) * (x - a)
        assert np.allclose(cols['KERN_FWD'][a + 1:b], chord, rtol=1e-13, atol=0) == (case == 'delta_above')


def test_near_tie_sites_are_evaluated_in_scipy_order(built, gpu_ctx):
    """The re-evaluated densities are the PAV_KDE_DIRECT ones bit for bit: at the doubtful row of argmax_search the guarded
    run-sum table equals the direct table exactly, while the unguarded run sums differ from it in the last digits (that
    difference, 1e-13 relative, is of the order of the margin the decision hangs on)."""
    g, p, r_dir, direct = run_neartie(gpu_ctx, 'argmax_search', kde_mode=_lib.KDE_DIRECT)
    _, _, r_on, guarded = run_neartie(gpu_ctx, 'argmax_search', kde_mode=_lib.KDE_RUNS)
    _, _, r_off, raw = run_neartie(gpu_ctx, 'argmax_search', kde_mode=_lib.KDE_RUNS, guard_rel=-1.0)
    row = p['row']
    assert r_off.n_near_tie == 0 and r_off.n_reeval == 0 and r_on.n_reeval >= 1
    for c in ('KERN_FWD', 'KERN_REV'):
        assert guarded[c][row] == direct[c][row]
        assert np.isclose(raw[c][row], direct[c][row], rtol=1e-11, atol=0)
    assert guarded['STATE'][row] == direct['STATE'][row] == g['STATE'][row]


@pytest.mark.parametrize('cap', [0, 300], ids=['list', 'overflow'])
def test_guard_on_everything_equals_direct_mode(built, gpu_ctx, cap):
    """guard_rel = 2 makes every decision doubtful: all sampled sites and all evaluated rows are summed again in scipy's
    order and everything downstream is redone from them, so the run-sum table must come out bit-identical to the
    PAV_KDE_DIRECT table (KERN_* included).  With a list of 300 entries the list overflows and the library falls back to
    PAV_KDE_DIRECT for the batch: same tables, guard_fallback set."""
    hap = synth.config2(seed=31, scale=0.004, threads=4)
    names = hap.ref.names
    gpu_ctx._inv_loaded = None
    gpu_ctx.seq_load(_lib.PAV_ROLE_REF, names, [hap.ref.seqs[n] for n in names])
    gpu_ctx.seq_load(_lib.PAV_ROLE_TIG, hap.tig_names, [hap.tig_seqs[n] for n in hap.tig_names])
    lift = AlignLift(hap.df_trim, hap.tig_lengths)
    fai = pd.Series(hap.ref.lengths)
    ref_i = {n: i for i, n in enumerate(names)}
    tig_i = {n: i for i, n in enumerate(hap.tig_names)}
    jobs = []
    for _, row in hap.df_flag.iterrows():
        r = pavseq.Region(row['#CHROM'], row['POS'], row['END'])
        r.expand(4000, min_pos=0, max_end=fai, shift=True)
        try:
            t = lift.lift_region_to_qry(r)
        except RuntimeError:
            t = None
        if t is None or len(r) > 40_000:
            continue
        jobs.append(_lib.DenJob(ref_i[r.chrom], tig_i[t.chrom], r.pos, r.end, t.pos, t.end, 1 if t.is_rev else 0, 20))
        if len(jobs) >= 10:
            break
    assert len(jobs) >= 6

    def tables(**kw):
        res = gpu_ctx.density_batch(jobs, pavden.den_params(**kw))
        return res, [gpu_ctx.density_table(j, r.n_rows) if r.status != _lib.DEN_FAIL else None for j, r in enumerate(res)], \
            [gpu_ctx.density_runs(j, r.n_runs) if r.status != _lib.DEN_FAIL else None for j, r in enumerate(res)]
    r_d, t_d, runs_d = tables(kde_mode=_lib.KDE_DIRECT)
    r_g, t_g, runs_g = tables(kde_mode=_lib.KDE_RUNS, guard_rel=2.0, guard_cap=cap)
    n_ok = 0
    for a, b, ta, tb, ra, rb in zip(r_d, r_g, t_d, t_g, runs_d, runs_g):
        assert (a.status, a.n_rows, a.n_eval) == (b.status, b.n_rows, b.n_eval)
        if a.status == _lib.DEN_FAIL:
            continue
        assert ra == rb
        for c in ('INDEX', 'STATE_MER', 'STATE'):
            assert np.array_equal(ta[c], tb[c]), c
        if a.status == _lib.DEN_OK:
            n_ok += 1
            for c in KERN:
                assert np.array_equal(ta[c], tb[c]), c
            assert b.guard_fallback == (1 if cap else 0)
            assert b.n_near_tie > 0 and (cap or b.n_reeval >= b.n_sample)
    assert n_ok >= 4


def test_contig_planes_on_demand_equal_the_full_pack(built, gpu_ctx, monkeypatch):
    """Lazy contig pack (ctx.hip): with the contig planes packed on demand - homology windows decoded from the ASCII arena, the
    blocks under the scanned regions packed per batch, everything packed for verify mode - the SNV / indel records (homology
    columns included), the flagged loci, the scan log, the calls and their density tables, and the verify counters are the ones
    the full pack at load time gives (PAV_EAGER_PACK=1).  The contigs carry N runs and lower case, half of the rows are reverse."""
    hap = synth.config2(seed=919, scale=0.01, threads=4, pair_frac=0.01)
    names = hap.ref.names
    k_util = KmerUtil(31)
    rng = np.random.default_rng(4)
    tig_seqs = {n: a.copy() for n, a in hap.tig_seqs.items()}
    for n in hap.tig_names[:6]:                                # N runs inside the contigs: next to and inside homology windows
        a = tig_seqs[n]
        for s0 in rng.integers(0, max(1, a.shape[0] - 50), 20):
            a[int(s0):int(s0) + int(rng.integers(1, 40))] = ord('N')
    from pav_amd import cigarcall

    def run(eager):
        monkeypatch.setenv('PAV_EAGER_PACK', '1' if eager else '0')
        gpu_ctx._inv_loaded = None
        gpu_ctx.seq_load(_lib.PAV_ROLE_REF, names, [hap.ref.seqs[n] for n in names])
        gpu_ctx.seq_load(_lib.PAV_ROLE_TIG, hap.tig_names, [tig_seqs[n] for n in hap.tig_names])
        aln, text, off = cigarcall.pack_alignments(hap.df_align, names, hap.tig_names)
        gpu_ctx.cigar_load(aln, text, off)
        gpu_ctx._inv_loaded = ('ref.fa', 'tig.fa')
        out = []
        for rep in range(2):                                  # second pass: after pav_seq_pack (planes stale again / re-packed)
            if rep:
                gpu_ctx.seq_pack(_lib.PAV_ROLE_TIG)
            counts = gpu_ctx.cigar_call()
            snv, indel, blob = gpu_ctx.cigar_fetch(counts)
            index = hap.df_align['INDEX'].to_numpy(dtype='int64')
            trim = hap.df_trim[['POS', 'END', 'INDEX']].set_index('INDEX').astype(int).reindex(list(index), fill_value=-1)
            _, loci, _ = gpu_ctx.cigar_flag(trim['POS'].to_numpy(dtype='int64'), trim['END'].to_numpy(dtype='int64'),
                                            gpu_ctx.flag_params(sig_filter=_lib.SIG_SINGLE_CLUSTER))
            regions = pavinv.loci_regions(gpu_ctx, loci)
            log, found = io.StringIO(), io.StringIO()
            calls = pavinv.scan_for_inv_batch(regions, 'ref.fa', 'tig.fa', AlignLift(hap.df_trim, hap.tig_lengths), k_util, log=log,
                                              ctx=gpu_ctx, eager_tables=False, found_out=found)
            tables = {c.id: sha(np.concatenate([c.df[k].to_numpy().astype(np.float64) for k in ('INDEX', 'STATE', 'KERN_FWD', 'KERN_REV')]))
                      for c in calls if c is not None and not isinstance(c, RuntimeError)}
            out.append((sha(snv.tobytes()), sha(indel.tobytes()), sha(bytes(blob)), loci.tobytes(), log.getvalue(), found.getvalue(), tables))
        out.append(gpu_ctx.cigar_verify())
        return out, len(regions)

    work0 = gpu_ctx.kde_work()
    lazy, n_regions = run(False)
    work1 = gpu_ctx.kde_work()
    eager, _ = run(True)
    # pav_kde_work: cumulative counters - evaluation points, (point, run) pairs, (point, data point) pairs of scipy's double loop
    d_pts, d_runs, d_data = (b - a for a, b in zip(work0, work1))
    assert d_pts > 1000 and d_runs >= d_pts and d_data > 100 * d_runs
    assert n_regions >= 10 and len(lazy[0][6]) >= 1
    assert lazy[0] == lazy[1] and eager[0] == eager[1]
    assert lazy == eager


@pytest.mark.parametrize('eager', [False, True])
def test_dense_round_hands_its_block_over(built, gpu_ctx, monkeypatch, capfd, eager):
    """A scan round whose calls make up most of its batch keeps its tables in the batch's own column block (density_fetch_calls:
    the block changes hands with the stage, FLANK / MATCH are computed in place); PAV_CALL_GATHER=1 packs every round as round 2
    did.  Every column of every call, the log and the results must be the same - for tables copied inside the scan (eager) and
    for tables read afterwards (lazy), and over two scans in a row (the blocks rotate)."""
    from pav_amd import cigarcall
    hap = synth.config2(seed=919, scale=0.01, threads=4, pair_frac=0.01)
    names = hap.ref.names
    k_util = KmerUtil(31)
    gpu_ctx._inv_loaded = None
    gpu_ctx.seq_load(_lib.PAV_ROLE_REF, names, [hap.ref.seqs[n] for n in names])
    gpu_ctx.seq_load(_lib.PAV_ROLE_TIG, hap.tig_names, [hap.tig_seqs[n] for n in hap.tig_names])
    gpu_ctx.cigar_load(*cigarcall.pack_alignments(hap.df_align, names, hap.tig_names))
    gpu_ctx._inv_loaded = ('ref.fa', 'tig.fa')
    gpu_ctx.cigar_call()
    index = hap.df_align['INDEX'].to_numpy(dtype='int64')
    trim = hap.df_trim[['POS', 'END', 'INDEX']].set_index('INDEX').astype(int).reindex(list(index), fill_value=-1)
    _, loci, _ = gpu_ctx.cigar_flag(trim['POS'].to_numpy(dtype='int64'), trim['END'].to_numpy(dtype='int64'),
                                    gpu_ctx.flag_params(sig_filter=_lib.SIG_SINGLE_CLUSTER))
    regions = pavinv.loci_regions(gpu_ctx, loci)
    lift = AlignLift(hap.df_trim, hap.tig_lengths)
    monkeypatch.setenv('PAV_TIMING', '1')                       # the library says which way a round's tables went

    def scan():
        log, found = io.StringIO(), io.StringIO()
        calls = pavinv.scan_for_inv_batch(regions, 'ref.fa', 'tig.fa', lift, k_util, log=log, ctx=gpu_ctx, eager_tables=eager,
                                          found_out=found)
        tables = {c.id: c.df.to_csv(sep='\t', index=False) for c in calls if c is not None and not isinstance(c, RuntimeError)}
        return log.getvalue(), found.getvalue(), tables

    capfd.readouterr()
    handed = [scan(), scan()]
    said = capfd.readouterr().err
    monkeypatch.setenv('PAV_CALL_GATHER', '1')
    packed = scan()
    said_packed = capfd.readouterr().err
    assert "the batch's own block" in said and "the batch's own block" not in said_packed and 'packed, resident' in said_packed
    assert len(packed[2]) >= 3
    assert handed[0] == packed and handed[1] == packed
    assert any(('UP' in t or 'DN' in t) for t in packed[2].values())     # FLANK annotated in place


def test_regions_with_forward_kmers_only_are_settled_from_their_counts(built, gpu_ctx, monkeypatch, capfd):
    """The native scan marks its density batches scan-only: a region whose k-mers are all FWD (after the low-count states are
    dropped) has STATE 0 in every row whatever the densities are, so it is neither compacted nor evaluated and its run list is
    made from its counts.  PAV_SCAN_FULL=1 evaluates every region as the density API does.  Log, 'INV Found' lines, calls and
    their tables must be the same; most flagged regions of a haplotype are of that kind."""
    import re
    from pav_amd import cigarcall
    hap = synth.config2(seed=919, scale=0.01, threads=4, pair_frac=0.01)
    names = hap.ref.names
    k_util = KmerUtil(31)
    gpu_ctx._inv_loaded = None
    gpu_ctx.seq_load(_lib.PAV_ROLE_REF, names, [hap.ref.seqs[n] for n in names])
    gpu_ctx.seq_load(_lib.PAV_ROLE_TIG, hap.tig_names, [hap.tig_seqs[n] for n in hap.tig_names])
    gpu_ctx.cigar_load(*cigarcall.pack_alignments(hap.df_align, names, hap.tig_names))
    gpu_ctx._inv_loaded = ('ref.fa', 'tig.fa')
    gpu_ctx.cigar_call()
    index = hap.df_align['INDEX'].to_numpy(dtype='int64')
    trim = hap.df_trim[['POS', 'END', 'INDEX']].set_index('INDEX').astype(int).reindex(list(index), fill_value=-1)
    _, loci, _ = gpu_ctx.cigar_flag(trim['POS'].to_numpy(dtype='int64'), trim['END'].to_numpy(dtype='int64'),
                                    gpu_ctx.flag_params(sig_filter=_lib.SIG_SINGLE_CLUSTER))
    regions = pavinv.loci_regions(gpu_ctx, loci)
    lift = AlignLift(hap.df_trim, hap.tig_lengths)
    monkeypatch.setenv('PAV_TIMING', '1')

    def scan():
        log, found = io.StringIO(), io.StringIO()
        calls = pavinv.scan_for_inv_batch(regions, 'ref.fa', 'tig.fa', lift, k_util, log=log, ctx=gpu_ctx, eager_tables=False, found_out=found)
        tables = {c.id: sha(c.df.to_csv(sep='\t', index=False).encode()) for c in calls if c is not None and not isinstance(c, RuntimeError)}
        outcome = [None if c is None else (str(c) if isinstance(c, RuntimeError) else c.id) for c in calls]
        return log.getvalue(), found.getvalue(), tables, outcome

    capfd.readouterr()
    short = scan()
    said = capfd.readouterr().err
    monkeypatch.setenv('PAV_SCAN_FULL', '1')
    full = scan()
    assert short == full and len(full[2]) >= 3
    m = re.search(r'in (\d+) of (\d+) jobs with FWD k-mers only', said)
    assert m and int(m.group(1)) >= 5 and int(m.group(1)) < int(m.group(2))
    assert 'Found no inverted k-mer states after 1 expansion(s)' in full[0]


def test_concurrent_haplotype_lanes_equal_sequential_runs(built, gpu_ctx):
    """bench.py's lanes in small: four haplotypes resident at once - one context each, sharing the reference planes - run
    the whole chain (CIGAR-call -> flagging -> scan of the flagged loci with lazy tables) from four host threads at the same
    time, twice.  Every lane's loci, calls and density tables equal what the same haplotype gives alone on one context."""
    import threading
    haps = [synth.config2(seed=717, scale=0.01, hap_index=0, threads=4, pair_frac=0.01)]
    ref = haps[0].ref
    names = ref.names
    haps += [synth.config2(seed=717, scale=0.01, hap_index=h, ref=ref, threads=4, pair_frac=0.01) for h in range(1, 4)]
    k_util = KmerUtil(31)

    def chain(ctx, hap, lift):
        from pav_amd import cigarcall
        index = hap.df_align['INDEX'].to_numpy(dtype='int64')
        trim = hap.df_trim[['POS', 'END', 'INDEX']].set_index('INDEX').astype(int).reindex(list(index), fill_value=-1)
        ctx.seq_pack(_lib.PAV_ROLE_TIG)
        counts = ctx.cigar_call()
        _, loci, _ = ctx.cigar_flag(trim['POS'].to_numpy(dtype='int64'), trim['END'].to_numpy(dtype='int64'),
                                    ctx.flag_params(sig_filter=_lib.SIG_SINGLE_CLUSTER))
        regions = pavinv.loci_regions(ctx, loci)
        log, found = io.StringIO(), io.StringIO()
        out = pavinv.scan_for_inv_batch(regions, 'ref.fa', 'tig.fa', lift, k_util, log=log, ctx=ctx, eager_tables=False, found_out=found)
        tables = {c.id: sha(np.concatenate([c.df[k].to_numpy().astype(np.float64) for k in ('INDEX', 'STATE', 'KERN_FWD', 'KERN_REV')]))
                  for c in out if c is not None and not isinstance(c, RuntimeError)}
        return counts.n_snv, loci.tobytes(), log.getvalue(), found.getvalue(), tables

    def setup(ctx, hap):
        from pav_amd import cigarcall
        ctx.seq_load(_lib.PAV_ROLE_TIG, hap.tig_names, [hap.tig_seqs[n] for n in hap.tig_names])
        ctx.cigar_load(*cigarcall.pack_alignments(hap.df_align, names, hap.tig_names))
        ctx._inv_loaded = ('ref.fa', 'tig.fa')
        return AlignLift(hap.df_trim, hap.tig_lengths)

    gpu_ctx._inv_loaded = None
    gpu_ctx.seq_load(_lib.PAV_ROLE_REF, names, [ref.seqs[n] for n in names])
    alone = []
    for hap in haps:
        alone.append(chain(gpu_ctx, hap, setup(gpu_ctx, hap)))
    assert sum(len(a[4]) for a in alone) >= 4 and len({a[0] for a in alone}) == 4
    lanes = [_lib.Context(0) for _ in haps]
    try:
        lifts = []
        for c, hap in zip(lanes, haps):
            c.seq_share(gpu_ctx, _lib.PAV_ROLE_REF)
            lifts.append(setup(c, hap))
        for _ in range(2):
            got, errs = [None] * len(haps), []

            def work(i):
                try:
                    got[i] = chain(lanes[i], haps[i], lifts[i])
                except BaseException as ex:      # noqa: BLE001
                    errs.append(ex)
            ths = [threading.Thread(target=work, args=(i,)) for i in range(len(haps))]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            assert not errs, errs
            assert got == alone
    finally:
        for c in lanes:
            c.close()


def test_files_to_files_tool(built, tmp_path):
    """tools/bench_e2e.py at a small scale: FASTA + alignment tables in, every output file of the rule chain out (merged
    SNV / INS-DEL tables, five flag tables, INV BED, density tables, log), through the native readers and writers
    (pav_amd.rules.call_haplotype: the reference's file names under <out>/out)."""
    import gzip
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / 'e2e'
    p = subprocess.run([sys.executable, os.path.join(root, 'tools', 'bench_e2e.py'), '--scale', '0.02', '--inv-sig-filter', 'single_cluster',
                        '--out', str(out)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line['snv_rows'] > 1000 and line['insdel_rows'] > 100 and line['aligned_bp'] > 10_000_000
    assert 0 < line['scanned_regions'] <= line['flagged_regions']
    res = out / 'out'
    names = {'snv': res / 'temp/sample/cigar/merged/snv_snv_h1.bed.gz', 'insdel': res / 'temp/sample/cigar/merged/svindel_insdel_h1.bed.gz',
             'flagged': res / 'results/sample/inv_caller/flagged_regions_h1.bed.gz', 'inv': res / 'temp/sample/inv_caller/sv_inv_h1.bed.gz',
             'log0': res / 'log/sample/inv_caller/log/h1/inv_call_0.log'}
    for name, path in names.items():
        assert path.exists(), name
    with gzip.open(names['snv'], 'rt') as fh:
        snv = pd.read_csv(fh, sep='\t')
    assert snv.shape[0] == line['snv_rows'] and list(snv.columns[:4]) == ['#CHROM', 'POS', 'END', 'ID']
    key = list(zip(snv['#CHROM'], snv['POS']))
    assert key == sorted(key)                                                    # merged order of rule call_cigar_merge
    if line['inv_calls']:
        inv = pd.read_csv(names['inv'], sep='\t')
        assert inv.shape[0] == line['inv_calls']
        assert len(os.listdir(res / 'results/sample/inv_caller/density_table')) == line['inv_calls']


def test_measurement_tools_run_on_a_small_case(built):
    """tools/lane_scaling.py and tools/batch_probe.py (the round-4 measurements of DESIGN.md section 5) on a shrunk haplotype:
    both finish, print their summary line, and every lane count / batch size reports a positive time and the same call counts."""
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'lane_scaling.py'), '--no-build', '--scale', '0.004', '--lanes', '1,2',
                          '--steps', '2', '--phases', 'call,all'], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('LANE_SCALING ')][0][len('LANE_SCALING '):])
    assert set(rec) == {'call', 'all'} and all(rec[p][n]['ms_per_pass'] > 0 for p in rec for n in ('1', '2'))
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'batch_probe.py'), '--no-build', '--scale', '0.004', '--batch', '1,2',
                          '--steps', '2'], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('BATCH_PROBE ')][0][len('BATCH_PROBE '):])
    assert rec['1']['ms_per_haplotype'] > 0 and rec['2']['n_snv'] > rec['1']['n_snv'] > 0
