"""Full-size parity (BASELINE configs[1] / configs[2] at scale 1.0, seed 1002 - the haplotype bench.py times): the device
records of the whole hg38-shaped haplotype against the scalar oracle, byte for byte, and the density tables of the first
scan iteration of >= 300 flagged loci of ANY size (integer columns exact, KERN_* to 1e-12), the complete scan against the
calls of an oracle-driven scan, and the last iterations of the largest calls row by row.  Needs ~12 GB of host memory."""
import numpy as np
import pandas as pd
import pytest

import util
from pav_amd import _lib, cigarcall, density as pavden, inv as pavinv, synth
from pav_amd.align import AlignLift

pytestmark = pytest.mark.gpu
KERN = ('KERN_FWD', 'KERN_FWDREV', 'KERN_REV')


_FULL = {}


def fullsize(ctx):
    """The bench haplotype resident on the session context, its records called and its loci flagged - once for the tests of this
    module that look at it (generating it takes a minute and 12 GB; test_chm13_* drops it before it needs the memory)."""
    if 'hap' not in _FULL:
        hap = synth.config2(seed=1002, scale=1.0, threads=8, pair_frac=0.009)
        names = hap.ref.names
        ctx._inv_loaded = None
        ctx.seq_load(_lib.PAV_ROLE_REF, names, [hap.ref.seqs[n] for n in names])
        ctx.seq_load(_lib.PAV_ROLE_TIG, hap.tig_names, [hap.tig_seqs[n] for n in hap.tig_names])
        snv, indel, blob, counts = cigarcall.call_records(ctx, hap.df_align)
        index = hap.df_align['INDEX'].to_numpy(dtype='int64')
        trim = hap.df_trim[['POS', 'END', 'INDEX']].set_index('INDEX').astype(int).reindex(list(index), fill_value=-1)
        tables, loci, _ = ctx.cigar_flag(trim['POS'].to_numpy(dtype='int64'), trim['END'].to_numpy(dtype='int64'),
                                         ctx.flag_params(sig_filter=_lib.SIG_SINGLE_CLUSTER))
        # the scans below name their sequences by these paths; only the .fai of the reference is ever read (inv.py:201)
        import os
        import tempfile
        import oracle_scan
        d = tempfile.mkdtemp(prefix='pav_fullsize_')
        fa = (os.path.join(d, 'ref.fa'), os.path.join(d, 'tig.fa'))
        oracle_scan.write_fai(fa[0] + '.fai', names, hap.ref.lengths)
        _FULL.update(hap=hap, records=(snv, indel, blob, counts), tables=tables, loci=loci,
                     lift=AlignLift(hap.df_trim, hap.tig_lengths), fa=fa)
        ctx._inv_loaded = fa
    return _FULL


def test_full_size_haplotype_is_bit_exact_vs_the_oracle(built, gpu_ctx):
    from oracle import oracle
    ctx = gpu_ctx
    full = fullsize(ctx)
    hap = full['hap']
    names = hap.ref.names
    snv, indel, blob, counts = full['records']
    assert counts.aligned_bases == hap.stats['aligned_bp'] > 3.0e9 and counts.n_snv > 6_000_000 and counts.n_indel > 500_000

    # ---- CIGAR-call: every record of the haplotype ------------------------------------------------------------------
    o_snv, o_indel, o_blob, err = util.oracle_records(names, [hap.ref.seqs[n] for n in names], hap.tig_names,
                                                      [hap.tig_seqs[n] for n in hap.tig_names], hap.df_align)
    assert err.kind == 0
    util.assert_records_equal(snv, o_snv, 'snv')
    util.assert_records_equal(indel, o_indel, 'indel')
    assert blob.tobytes() == o_blob.tobytes()
    del o_snv, o_indel, o_blob

    # ---- flagging of those calls -> loci -> first scan iteration of the first 320 liftable loci, whatever their size -----
    tables, loci = full['tables'], full['loci']
    regions = pavinv.loci_regions(ctx, loci)
    assert len(regions) >= 900 and len(tables['cluster_snv']) >= 90          # ~1 k flagged loci per haplotype
    lift = full['lift']
    fai = pd.Series(hap.ref.lengths)
    ref_i, tig_i = {n: i for i, n in enumerate(names)}, {n: i for i, n in enumerate(hap.tig_names)}
    jobs, pairs = [], []
    for r in regions:
        r.expand(4000, min_pos=0, max_end=fai, shift=True)
        try:
            t = lift.lift_region_to_qry(r)
        except RuntimeError:
            t = None
        if t is None:
            continue
        jobs.append(_lib.DenJob(ref_i[r.chrom], tig_i[t.chrom], r.pos, r.end, t.pos, t.end, 1 if t.is_rev else 0, 20))
        pairs.append((r, t))
        if len(jobs) >= 320:
            break
    assert len(jobs) >= 300 and max(len(r) for r, _ in pairs) > 100_000
    res = ctx.density_batch(jobs, pavden.den_params())
    n_final = near = 0
    threads = util.usable_cpus()
    for j, ((r, t), g) in enumerate(zip(pairs, res)):
        o = oracle.density(hap.ref.seqs[r.chrom][r.pos:r.end], hap.tig_seqs[t.chrom][t.pos:t.end], t.is_rev,
                           threads=threads if len(r) > 30_000 else 1)
        assert g.status == o['status'], (r, t)
        if o['status'] == 125:
            continue
        assert g.n_rows == o['n']
        cols = ctx.density_table(j, g.n_rows)
        for c in ('INDEX', 'STATE_MER', 'STATE', 'KMER'):
            assert np.array_equal(cols[c], o[c]), (c, r)
        assert ctx.density_runs(j, g.n_runs) == oracle.rl_encode(o['STATE'], o['INDEX'])
        if o['status'] == 0:
            n_final += 1
            near += g.n_near_tie
            assert g.n_unresolved == 0
            for c in KERN:
                assert np.allclose(cols[c], o[c], rtol=1e-12, atol=1e-300), (c, r)
    assert n_final >= 250
    print(f'{len(jobs)} loci (largest region {max(len(r) for r, _ in pairs)} bp), {n_final} finalised tables equal the oracle; '
          f'{near} near-tie decisions re-evaluated')


def test_full_size_scan_with_helper_threads(built, gpu_ctx, monkeypatch):
    """The native driver's per-region loops on four host threads (PAV_HOST_THREADS; csrc/pool.h) over the 1 128 loci of the bench
    haplotype - the loops the golden cases are too small to share out: the same regions, calls and log text as the digest of the
    oracle-driven scan (tests/golden/fullsize_inv_calls.json), three scans in a row."""
    import hashlib
    import io
    import json
    import os
    import oracle_scan
    from pav_amd.kmer import KmerUtil
    full = fullsize(gpu_ctx)
    with open(os.path.join(util.GOLD, 'fullsize_inv_calls.json')) as fh:
        gold = json.load(fh)
    want = {c['region']: c for c in gold['calls']}
    monkeypatch.setenv('PAV_HOST_THREADS', '4')
    lift = AlignLift(full['hap'].df_trim, full['hap'].tig_lengths)       # an index of its own: the helpers are made with it
    with _lib.Context(0) as ctx:
        ctx.seq_share(gpu_ctx, _lib.PAV_ROLE_REF)
        ctx.seq_share(gpu_ctx, _lib.PAV_ROLE_TIG)
        ctx._inv_loaded = full['fa']
        regions = pavinv.loci_regions(ctx, full['loci'])
        for rep in range(3):
            log = io.StringIO()
            out = pavinv.scan_for_inv_batch(regions, full['fa'][0], full['fa'][1], lift, KmerUtil(31), log=log, ctx=ctx, native=True,
                                            eager_tables=False, found_out=io.StringIO())
            got = {i: c for i, c in enumerate(out) if c is not None}
            assert sorted(got) == sorted(want), rep
            for i, c in got.items():
                for k2, v in oracle_scan.call_record(c).items():
                    assert want[i][k2] == v, (rep, i, k2)
            text = log.getvalue()
            assert text.count('\n') == gold['log_lines'] and hashlib.sha1(text.encode()).hexdigest() == gold['log_sha1'], rep


def test_full_size_inversion_calls_vs_the_oracle_driven_scan(built, gpu_ctx):
    """The COMPLETE scan of the bench haplotype - every flagged locus, every expansion round, up to regions of ~0.5 Mbp -
    against tests/golden/fullsize_inv_calls.json: the calls the Python state machine makes when the CPU oracle answers every
    density job (tools/gen_fullsize_inv_digest.py; neither the density kernels nor the native driver take part in it).  Both
    drivers on the device must give the same regions in, the same calls (ids, outer / inner / discovery regions on both
    sequences, table rows, STATE / STATE_MER digests) and the same log text out.  Then the LAST iteration of the 20 calls with
    the largest discovery regions - the second round of their scans, 100 - 500 kbp - is run again as one density batch and
    compared with the oracle row by row (integer columns exact, KERN_* to 1e-12)."""
    import contextlib
    import hashlib
    import io
    import json
    import os
    import oracle_scan
    from oracle import oracle
    from pav_amd.kmer import KmerUtil
    ctx = gpu_ctx
    full = fullsize(ctx)
    hap, lift = full['hap'], full['lift']
    names = hap.ref.names
    with open(os.path.join(util.GOLD, 'fullsize_inv_calls.json')) as fh:
        gold = json.load(fh)
    regions = pavinv.loci_regions(ctx, full['loci'])
    assert len(regions) == gold['n_regions']
    assert hashlib.sha1('\n'.join(f'{r.chrom}:{r.pos}-{r.end}' for r in regions).encode()).hexdigest() == gold['regions_sha1']
    k_util = KmerUtil(31)
    want = {c['region']: c for c in gold['calls']}
    sha = lambda a: hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()   # noqa: E731
    calls_native = None
    for native in (True, False):
        log = io.StringIO()
        ctx._inv_loaded = full['fa']
        with contextlib.redirect_stdout(io.StringIO()):                 # the Python driver prints 'INV Found' (inv.py:408)
            out = pavinv.scan_for_inv_batch(regions, full['fa'][0], full['fa'][1], lift, k_util, log=log, ctx=ctx, native=native,
                                            eager_tables=False, found_out=io.StringIO())
        assert not any(isinstance(c, RuntimeError) for c in out)
        got = {i: c for i, c in enumerate(out) if c is not None}
        assert sorted(got) == sorted(want), (native, sorted(set(got) ^ set(want)))
        for i, c in got.items():
            rec = oracle_scan.call_record(c)
            for k2, v in rec.items():
                assert want[i][k2] == v, (native, i, k2)
            df = c.df
            assert df.shape[0] == want[i]['n_rows'], (native, i)
            assert sha(df['STATE'].to_numpy(dtype=np.int8)) == want[i]['state_sha1'], (native, c.id)
            assert sha(df['STATE_MER'].to_numpy(dtype=np.int8)) == want[i]['state_mer_sha1'], (native, c.id)
        text = log.getvalue()
        assert text.count('\n') == gold['log_lines'] and hashlib.sha1(text.encode()).hexdigest() == gold['log_sha1'], native
        if native:
            calls_native = got
            assert sum(c.n_unresolved for c in got.values()) == 0
    assert len(want) >= 90

    # ---- the last iteration of the 20 largest calls, row by row vs the oracle -------------------------------------------------
    big = sorted(calls_native.values(), key=lambda c: -len(c.region_ref_discovery))[:20]
    ref_i, tig_i = {n: i for i, n in enumerate(names)}, {n: i for i, n in enumerate(hap.tig_names)}
    pairs = [(c.region_ref_discovery, c.region_tig_discovery) for c in big]
    jobs = [_lib.DenJob(ref_i[r.chrom], tig_i[t.chrom], r.pos, r.end, t.pos, t.end, 1 if t.is_rev else 0, 20) for r, t in pairs]
    res = ctx.density_batch(jobs, pavden.den_params())
    threads = util.usable_cpus()
    for j, ((r, t), g) in enumerate(zip(pairs, res)):
        o = oracle.density(hap.ref.seqs[r.chrom][r.pos:r.end], hap.tig_seqs[t.chrom][t.pos:t.end], t.is_rev, threads=threads)
        assert g.status == o['status'] == 0 and g.n_rows == o['n'] == want[big[j]._i]['n_rows']
        cols = ctx.density_table(j, g.n_rows)
        for c in ('INDEX', 'STATE_MER', 'STATE', 'KMER'):
            assert np.array_equal(cols[c], o[c]), (c, r)
        assert ctx.density_runs(j, g.n_runs) == oracle.rl_encode(o['STATE'], o['INDEX'])
        assert g.n_unresolved == 0
        for c in KERN:
            assert np.allclose(cols[c], o[c], rtol=1e-12, atol=1e-300), (c, r)
    print(f'{len(want)} calls of {len(regions)} regions equal the oracle-driven scan (both drivers); last iterations of the 20 largest '
          f'({min(len(r) for r, _ in pairs)} - {max(len(r) for r, _ in pairs)} bp) equal the oracle row by row')


def test_full_size_loci_vs_the_reference_itself(built, gpu_ctx, monkeypatch):
    """Loci of the bench haplotype on which the REFERENCE was run (tests/golden/fullsize_loci: pavlib.inv.scan_for_inv ->
    scripts/density.py -> scipy, unmodified, on the haplotype's own 3 GB FASTA files and its complete alignment table -
    tools/refharness/gen_golden_fullsize_loci.py): the ten calls with the largest discovery regions (two scan rounds, up to 479
    kbp), loci of every kind of ending the scan has on this haplotype, the loci with the most rounds, the smallest calls.  The
    native driver's log text, calls (id, the six regions with their alignment indices, the INV BED row incl. SEQ) and final
    density tables (INDEX / STATE_MER / STATE digests, FLANK, KERN_* sums to 1e-9) must be the reference's."""
    import hashlib
    import io
    import json
    import os
    from pav_amd import rules
    from pav_amd.kmer import KmerUtil
    gold_path = os.path.join(util.GOLD, 'fullsize_loci', 'scans.json')
    with open(gold_path) as fh:
        gold = json.load(fh)
    ctx = gpu_ctx
    full = fullsize(ctx)
    hap = full['hap']
    regions = pavinv.loci_regions(ctx, full['loci'])
    assert len(regions) == gold['n_loci_of_the_haplotype']
    ctx._inv_loaded = full['fa']
    logs = [io.StringIO() for _ in regions]
    out = pavinv.scan_for_inv_batch(regions, full['fa'][0], full['fa'][1], full['lift'], KmerUtil(31), logs=logs, ctx=ctx, native=True,
                                    eager_tables=False, found_out=io.StringIO())
    sha = lambda a: hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()   # noqa: E731
    from pav_amd import seq as pavseq
    monkeypatch.setattr(pavseq, 'open_fasta', lambda path: hap.tig_seqs)          # the SEQ column: the haplotype is in memory, not in a file

    def rg(d):
        return None if d is None else (d['chrom'], d['pos'], d['end'], d['is_rev'], d['pos_aln_index'], d['end_aln_index'])

    def ours(r):
        aln = lambda x: None if x is None else [[int(w) for w in v] if isinstance(v, (tuple, list)) else int(v) for v in x]   # noqa: E731
        return (r.chrom, int(r.pos), int(r.end), bool(r.is_rev), aln(r.pos_aln_index), aln(r.end_aln_index))
    n_calls = 0
    for rec in gold['scans']:
        i = rec['region']
        f = rec['flag']
        assert (regions[i].chrom, int(regions[i].pos), int(regions[i].end)) == (f['chrom'], f['pos'], f['end']), i
        assert logs[i].getvalue().splitlines() == rec['log'], (i, rec['why'])
        c = out[i]
        if rec['error'] is not None:
            assert isinstance(c, RuntimeError) and str(c) == rec['error'], i
            continue
        if rec['call'] is None:
            assert c is None, (i, rec['why'])
            continue
        want = rec['call']
        assert c is not None and not isinstance(c, RuntimeError), (i, rec['why'])
        n_calls += 1
        assert (c.id, int(c.svlen)) == (want['id'], want['svlen'])
        for nm in ('region_ref_outer', 'region_ref_inner', 'region_tig_outer', 'region_tig_inner', 'region_ref_discovery', 'region_tig_discovery'):
            assert ours(getattr(c, nm)) == rg(want[nm]), (c.id, nm)
        df = c.df
        assert df.shape[0] == want['n_rows'], c.id
        assert sha(df['INDEX'].to_numpy(dtype=np.int64)) == want['index_sha1'], c.id
        assert sha(df['STATE_MER'].to_numpy(dtype=np.int8)) == want['state_mer_sha1'], c.id
        assert sha(df['STATE'].to_numpy(dtype=np.int8)) == want['state_sha1'], c.id
        assert hashlib.sha1('\n'.join(df['FLANK'].tolist()).encode()).hexdigest() == want['flank_sha1'], c.id
        for col, v in zip(('KERN_FWD', 'KERN_FWDREV', 'KERN_REV'), want['kern_sum']):
            assert np.isclose(float(df[col].sum()), v, rtol=1e-9, atol=1e-300), (c.id, col)
        row = rules.inv_bed_row(c, hap.hap, 'RGN', full['fa'][1])
        bed = {k2: (int(v) if isinstance(v, (int, np.integer)) else v) for k2, v in row.items()}
        seq = bed.pop('SEQ')
        assert (hashlib.sha1(seq.encode()).hexdigest(), len(seq)) == (want['bed_row']['SEQ_sha1'], want['bed_row']['SEQ_len']), c.id
        for k2, v in bed.items():
            assert want['bed_row'][k2] == v, (c.id, k2)
    assert n_calls >= 15 and len(gold['scans']) >= 20


def test_chm13_cohort_batch_of_eight_full_size(built, gpu_ctx):
    """BASELINE configs[4] at its real size: a T2T-CHM13-shaped reference (24 sequences, CHM13v2.0
    lengths, 3.1 Gbp, no N runs) packed ONCE, eight haplotypes of the cohort resident against it at the same time (one
    context each), and the WHOLE path - CIGAR-call -> flagging -> k-mer inversion scan of every locus with TRY_INV - driven on
    all eight lanes concurrently, twice.  Size-independent properties per haplotype: counts == the generator's ground truth;
    SNV rows in (row, position) order with REF != ALT; INS / DEL SEQ bytes add up; the planted (aligned-through) inversions
    are flagged and most of them called; no float decision is left unresolved; the second concurrent pass reproduces the
    first byte for byte (records, loci, scan log, call tables); lane 0 equals the same haplotype run alone.  Prints the HBM
    the batch needs.  ~25 GB of host memory is never needed: a haplotype's host copy is dropped once it is resident."""
    import hashlib
    import io
    import threading
    from pav_amd.kmer import KmerUtil
    n_lanes = 8
    k_util = KmerUtil(31)
    _FULL.clear()                                               # the bench haplotype of the tests above: 12 GB of host memory
    free0, total = gpu_ctx.mem_info()
    hap0 = synth.config5(seed=1005, scale=1.0, hap_index=0, threads=8, pair_frac=0.009)
    ref = hap0.ref
    names = ref.names
    assert sum(ref.lengths.values()) > 3.0e9 and not any((ref.seqs[n] == ord('N')).any() for n in names[-2:])
    gpu_ctx._inv_loaded = None
    gpu_ctx.seq_load(_lib.PAV_ROLE_REF, names, [ref.seqs[n] for n in names])
    rank_of = {n: i for i, n in enumerate(sorted(names))}

    def digest(*arrays):
        h = hashlib.sha1()
        for a in arrays:
            h.update(np.ascontiguousarray(a).tobytes())
        return h.hexdigest()

    class Lane:
        pass

    def setup(ctx, hap):
        ln = Lane()
        ln.ctx, ln.stats, ln.inversions = ctx, hap.stats, ref.inversions
        ctx.seq_load(_lib.PAV_ROLE_TIG, hap.tig_names, [hap.tig_seqs[n] for n in hap.tig_names])
        ctx.cigar_load(*cigarcall.pack_alignments(hap.df_align, names, hap.tig_names))
        ctx._inv_loaded = ('ref.fa', 'tig.fa')
        index = hap.df_align['INDEX'].to_numpy(dtype='int64')
        trim = hap.df_trim[['POS', 'END', 'INDEX']].set_index('INDEX').astype(int).reindex(list(index), fill_value=-1)
        ln.tp, ln.te = trim['POS'].to_numpy(dtype='int64'), trim['END'].to_numpy(dtype='int64')
        ln.lift = AlignLift(hap.df_trim, hap.tig_lengths)
        hap.tig_seqs.clear()
        return ln

    def chain(ln, check):
        ctx = ln.ctx
        ctx.seq_pack(_lib.PAV_ROLE_TIG)
        counts = ctx.cigar_call()
        _, loci, _ = ctx.cigar_flag(ln.tp, ln.te, ctx.flag_params(sig_filter=_lib.SIG_SINGLE_CLUSTER))
        regions = pavinv.loci_regions(ctx, loci)
        log, found = io.StringIO(), io.StringIO()
        out = pavinv.scan_for_inv_batch(regions, 'ref.fa', 'tig.fa', ln.lift, k_util, log=log, ctx=ctx, eager_tables=False, found_out=found)
        calls = [(i, c) for i, c in enumerate(out) if c is not None and not isinstance(c, RuntimeError)]
        assert not any(isinstance(c, RuntimeError) for c in out)
        tabs = []
        for i, c in calls:
            cols, flank, match = ctx.inv_table_view(i, c.native_table[2])
            tabs.append((c.id, digest(cols['INDEX'], cols['STATE_MER'], cols['STATE'], cols['KERN_FWD'], cols['KERN_FWDREV'],
                                      cols['KERN_REV'], cols['KMER'], flank, match)))
        snv, indel, blob = ctx.cigar_fetch(counts)
        if check:
            st = ln.stats
            assert (counts.n_ops, counts.n_snv, counts.n_indel, counts.aligned_bases) == \
                (st['n_ops'], st['n_snv'], st['n_ins'] + st['n_del'], st['aligned_bp'])
            assert counts.aligned_bases > 3.0e9
            key = snv['aln'].astype(np.int64) << 32 | snv['pos'].astype(np.int64)
            assert np.all(np.diff(key) > 0)
            assert np.all((snv['ref'] & 0xDF) != (snv['alt'] & 0xDF))
            assert int(indel['svlen'].sum()) == counts.seq_bytes == blob.shape[0]
            assert np.all(np.diff(indel['seq_off'].astype(np.int64)) == indel['svlen'][:-1])
            assert int((indel['svtype'] == 0).sum()) == st['n_ins']
            hit = sum(bool(len(loci[(loci['chrom'] == rank_of[iv.chrom]) & (loci['pos'] < iv.end) & (loci['end'] > iv.pos)]))
                      for iv in ln.inversions)
            n_inv = st['n_inv']                                      # inversions planted inside this haplotype's alignments
            assert hit >= 0.9 * n_inv and len(calls) >= 0.6 * n_inv, (hit, len(calls), n_inv, len(ln.inversions))
            assert sum(c.n_unresolved for _, c in calls) == 0
            assert len(regions) >= 900
        return digest(snv, indel, blob), loci.tobytes(), log.getvalue(), found.getvalue(), tabs

    lanes = []
    ctxs = [_lib.Context(0) for _ in range(n_lanes)]
    try:
        alone = None
        for li, c in enumerate(ctxs):
            hap = hap0 if li == 0 else synth.config5(seed=1005, scale=1.0, hap_index=li, ref=ref, threads=8, pair_frac=0.009)
            c.seq_share(gpu_ctx, _lib.PAV_ROLE_REF)
            lanes.append(setup(c, hap))
            if li == 0:
                alone = chain(lanes[0], True)                        # lane 0 with the GPU to itself
            del hap
        assert len({ln.stats['n_snv'] for ln in lanes}) == n_lanes      # eight different haplotypes
        free_min = [gpu_ctx.mem_info()[0]]
        passes = []
        for p in range(2):
            got, errs = [None] * n_lanes, []

            def work(i, p=p):
                try:
                    got[i] = chain(lanes[i], p == 0)
                except BaseException as ex:      # noqa: BLE001
                    errs.append(ex)
            ths = [threading.Thread(target=work, args=(i,)) for i in range(n_lanes)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            assert not errs, errs
            passes.append(got)
            free_min.append(gpu_ctx.mem_info()[0])
        assert passes[0] == passes[1]                                   # idempotent under concurrency
        assert passes[0][0] == alone                                    # and independent of what runs beside it
        used = (free0 - min(free_min)) / 1e9
        n_calls = [len(g[4]) for g in passes[0]]
        print(f'CHM13 batch: {n_lanes} haplotypes resident against one reference: {used:.1f} GB of {total / 1e9:.0f} GB HBM in use at the '
              f'peak; inversion calls per haplotype {n_calls}')
        assert used < 0.5 * total / 1e9
    finally:
        for c in ctxs:
            c.close()
