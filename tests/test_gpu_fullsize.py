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
