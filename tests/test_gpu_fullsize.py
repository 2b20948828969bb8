"""Full-size parity (BASELINE configs[1] / configs[2] at scale 1.0, seed 1002 - the haplotype bench.py times): the device
records of the whole hg38-shaped haplotype against the scalar oracle, byte for byte, and the density tables of the first
scan iteration of >= 300 flagged loci (integer columns exact, KERN_* to 1e-12).  Needs ~12 GB of host memory and ~1 minute."""
import numpy as np
import pandas as pd
import pytest

import util
from pav_amd import _lib, cigarcall, density as pavden, inv as pavinv, synth
from pav_amd.align import AlignLift

pytestmark = pytest.mark.gpu
KERN = ('KERN_FWD', 'KERN_FWDREV', 'KERN_REV')


def test_full_size_haplotype_is_bit_exact_vs_the_oracle(built, gpu_ctx):
    from oracle import oracle
    hap = synth.config2(seed=1002, scale=1.0, threads=8, pair_frac=0.009)
    names = hap.ref.names
    ctx = gpu_ctx
    ctx._inv_loaded = None
    ctx.seq_load(_lib.PAV_ROLE_REF, names, [hap.ref.seqs[n] for n in names])
    ctx.seq_load(_lib.PAV_ROLE_TIG, hap.tig_names, [hap.tig_seqs[n] for n in hap.tig_names])
    snv, indel, blob, counts = cigarcall.call_records(ctx, hap.df_align)
    assert counts.aligned_bases == hap.stats['aligned_bp'] > 3.0e9 and counts.n_snv > 6_000_000 and counts.n_indel > 500_000

    # ---- CIGAR-call: every record of the haplotype ------------------------------------------------------------------
    o_snv, o_indel, o_blob, err = util.oracle_records(names, [hap.ref.seqs[n] for n in names], hap.tig_names,
                                                      [hap.tig_seqs[n] for n in hap.tig_names], hap.df_align)
    assert err.kind == 0
    util.assert_records_equal(snv, o_snv, 'snv')
    util.assert_records_equal(indel, o_indel, 'indel')
    assert blob.tobytes() == o_blob.tobytes()
    del o_snv, o_indel, o_blob

    # ---- flagging of those calls -> loci -> first scan iteration of the first 300 liftable loci ----------------------------
    index = hap.df_align['INDEX'].to_numpy(dtype='int64')
    trim = hap.df_trim[['POS', 'END', 'INDEX']].set_index('INDEX').astype(int).reindex(list(index), fill_value=-1)
    tables, loci, _ = ctx.cigar_flag(trim['POS'].to_numpy(dtype='int64'), trim['END'].to_numpy(dtype='int64'),
                                     ctx.flag_params(sig_filter=_lib.SIG_SINGLE_CLUSTER))
    regions = pavinv.loci_regions(ctx, loci)
    assert len(regions) >= 900 and len(tables['cluster_snv']) >= 90          # ~1 k flagged loci per haplotype
    lift = AlignLift(hap.df_trim, hap.tig_lengths)
    fai = pd.Series(hap.ref.lengths)
    ref_i, tig_i = {n: i for i, n in enumerate(names)}, {n: i for i, n in enumerate(hap.tig_names)}
    jobs, pairs = [], []
    for r in regions:
        r.expand(4000, min_pos=0, max_end=fai, shift=True)
        try:
            t = lift.lift_region_to_qry(r)
        except RuntimeError:
            t = None
        if t is None or len(r) > 80_000:
            continue
        jobs.append(_lib.DenJob(ref_i[r.chrom], tig_i[t.chrom], r.pos, r.end, t.pos, t.end, 1 if t.is_rev else 0, 20))
        pairs.append((r, t))
        if len(jobs) >= 320:
            break
    assert len(jobs) >= 300
    res = ctx.density_batch(jobs, pavden.den_params())
    n_final = near = 0
    for j, ((r, t), g) in enumerate(zip(pairs, res)):
        o = oracle.density(hap.ref.seqs[r.chrom][r.pos:r.end], hap.tig_seqs[t.chrom][t.pos:t.end], t.is_rev)
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
    print(f'{len(jobs)} loci, {n_final} finalised tables equal the oracle; {near} near-tie decisions re-evaluated')


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
