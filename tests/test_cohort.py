"""pav_amd.cohort.run_cohort - many haplotypes on the GPUs of one node, one process per GPU, no data-path collective
(BASELINE configs[3] / [4]): the plan, and - with two ranks - that every output file of the sharded run
equals the unsharded run's.  CPU: two gloo ranks with the oracle-backed stand-in engine (tests/cohort_engines.py).  GPU: two
ranks sharing GPU 0 with the real engine, also against the tables the reference's rule bodies wrote."""
import os

import pandas as pd
import pytest

import cohort_engines as ce
import util
from pav_amd import cohort, rules

GOLD = util.GOLD
CFG = {'inv_sig_batch_count': 6}
CFG_SC = {'inv_sig_batch_count': 6, 'inv_sig_filter': 'single_cluster'}     # also try the loci that only show a cluster (aligned-through inversions)


def _golden_merged_tables(tree, case, asm, hap):
    """The merged tables of rule call_cigar_merge against the reference's (tests/golden/flag_*: the six columns the flag rules
    read, as the rule bodies of call_cigar x 10 + call_cigar_merge wrote them)."""
    import io
    for name, gold in ((f'temp/{asm}/cigar/merged/svindel_insdel_{hap}.bed.gz', 'svindel_insdel.tsv.gz'),
                       (f'temp/{asm}/cigar/merged/snv_snv_{hap}.bed.gz', 'snv_snv.tsv.gz')):
        want = pd.read_csv(os.path.join(GOLD, case, gold), sep='\t', dtype=str, keep_default_na=False)
        got = pd.read_csv(io.BytesIO(tree[name]), sep='\t', dtype=str, keep_default_na=False)
        assert got[list(want.columns)].equals(want), name


class _W:
    def __init__(self, w):
        self.w = w

    def weight(self):
        return self.w


def test_plan_deals_whole_haplotypes_longest_first_and_splits_when_there_are_fewer_than_ranks():
    # configs[3]: 16 haplotypes on 8 GPUs: two each, every haplotype exactly once, whole
    p = cohort.plan([_W(100 + i) for i in range(16)], 8)
    assert all(len(r) == 2 for r in p) and sorted(j for r in p for (j, _, n, _) in r) == list(range(16))
    assert all(n == 1 and lead == r for r, items in enumerate(p) for (_, _, n, lead) in items)
    # configs[4]: 64 on 8: eight each, loads within one haplotype of each other
    w = [50 + (i * 37) % 23 for i in range(64)]
    p = cohort.plan([_W(x) for x in w], 8)
    loads = [sum(w[j] for (j, _, _, _) in r) for r in p]
    assert all(len(r) == 8 for r in p) and max(loads) - min(loads) <= max(w)
    # configs[2]: a diploid sample on 8 GPUs: the ranks are shared out by weight, parts of a haplotype on consecutive ranks
    p = cohort.plan([_W(3), _W(1)], 8)
    assert [r[0][:3] for r in p] == [(0, q, 6) for q in range(6)] + [(1, 0, 2), (1, 1, 2)]
    assert {lead for r in p for (j, _, _, lead) in r if j == 0} == {0} and {lead for r in p for (j, _, _, lead) in r if j == 1} == {6}
    assert cohort.plan([], 4) == [[], [], [], []]
    assert cohort.plan([_W(1)], 1) == [[(0, 0, 1, 0)]]


@pytest.mark.parametrize('split', [False, True], ids=['whole-haplotypes', 'one-haplotype-shared'])
def test_two_gloo_ranks_equal_the_unsharded_run(built, tmp_path, split):
    """World size 2 over gloo, no GPU: (a) two haplotypes, one per rank; (b) one haplotype shared by both ranks - its CALL_BATCH
    jobs and its flagged-region BATCH jobs are split, the lead rank merges as call_cigar_merge / call_inv_batch_merge do.
    Every file under the output directory - merged SNV / INS-DEL tables, flag tables, per-batch INV tables and logs, density
    tables, the merged INV table - must equal the one-rank run's."""
    jobs = [ce.golden_job('flag_hap', 'sampleA', 'h1')] + ([] if split else [ce.golden_job('flag_sparse', 'sampleA', 'h2')])
    ref_fa = os.path.join(GOLD, 'flag_hap', 'ref.fa')
    one, two = tmp_path / 'one', tmp_path / 'two'
    m1 = cohort.run_cohort(jobs, 1, str(one), ref_fa, config=CFG_SC, engine_factory=ce.oracle_engine)
    m2 = cohort.run_cohort(jobs, 2, str(two), ref_fa, config=CFG_SC, engine_factory=ce.oracle_engine, backend='gloo', split=split, timeout=900)
    assert sorted((m['asm_name'], m['hap'], m['inv_calls']) for m in m1) == sorted((m['asm_name'], m['hap'], m['inv_calls']) for m in m2)
    assert {m['rank'] for m in m2} == ({0} if split else {0, 1})
    a, b = ce.tree_text(one), ce.tree_text(two)
    if split:                                                    # the one-rank run of the stand-in writes the batch files too
        assert sorted(a) == sorted(b)
    assert sorted(k for k in a if '/batched/' not in k) == sorted(k for k in b if '/batched/' not in k)
    for k in a:
        if k in b:
            assert a[k] == b[k], k
    assert any('/density_table/' in k for k in a) and sum(m['inv_calls'] for m in m1) >= 3
    _golden_merged_tables(b, 'flag_hap', 'sampleA', 'h1')


def test_four_gloo_ranks_uneven_shares_and_a_rank_without_work(built, tmp_path):
    """World size 4 over gloo, no GPU - what the first 8-GPU run meets in small.  (a) five haplotypes on four ranks: whole
    haplotypes, one rank takes two (uneven counts).  (b) two haplotypes on four ranks with TWO flagged-region batches: the ranks are
    shared out by weight, and in a group of more than two ranks the third has no inversion batch at all - it must pass the stages'
    barriers with nothing to do.  Every file equals the one-rank run's."""
    ref_fa = os.path.join(GOLD, 'flag_hap', 'ref.fa')
    five = [ce.golden_job('flag_hap', 'sampleA', 'h1'), ce.golden_job('flag_sparse', 'sampleA', 'h2'), ce.golden_job('flag_sparse', 'sampleB', 'h1'),
            ce.golden_job('flag_hap', 'sampleB', 'h2'), ce.golden_job('flag_sparse', 'sampleC', 'h1')]
    p = cohort.plan(five, 4)
    assert sorted(len(r) for r in p) == [1, 1, 1, 2] and sorted(j for r in p for (j, _, n, _) in r) == [0, 1, 2, 3, 4]
    one, four = tmp_path / 'one', tmp_path / 'four'
    m1 = cohort.run_cohort(five, 1, str(one), ref_fa, config=CFG_SC, engine_factory=ce.oracle_engine)
    m4 = cohort.run_cohort(five, 4, str(four), ref_fa, config=CFG_SC, engine_factory=ce.oracle_engine, backend='gloo', timeout=900)
    key = lambda m: (m['asm_name'], m['hap'], m['inv_calls'])   # noqa: E731
    assert sorted(map(key, m1)) == sorted(map(key, m4)) and {m['rank'] for m in m4} == {0, 1, 2, 3}
    a, b = ce.tree_text(one), ce.tree_text(four)
    assert sorted(a) == sorted(b)
    for k in a:
        assert a[k] == b[k], k
    # (b) groups of two ranks for one inversion batch each
    cfg2 = {'inv_sig_batch_count': 1, 'inv_sig_filter': 'single_cluster'}
    two = [ce.golden_job('flag_hap', 'sampleA', 'h1'), ce.golden_job('flag_sparse', 'sampleA', 'h2')]
    p = cohort.plan(two, 4)
    groups = sorted({(j, n) for r in p for (j, _, n, _) in r})
    assert sum(n for _, n in groups) == 4 and max(n for _, n in groups) >= 2
    one2, four2 = tmp_path / 'one2', tmp_path / 'four2'
    s1 = cohort.run_cohort(two, 1, str(one2), ref_fa, config=cfg2, engine_factory=ce.oracle_engine)
    s4 = cohort.run_cohort(two, 4, str(four2), ref_fa, config=cfg2, engine_factory=ce.oracle_engine, backend='gloo', timeout=900)
    assert sorted(map(key, s1)) == sorted(map(key, s4))
    a, b = ce.tree_text(one2), ce.tree_text(four2)
    assert sorted(k for k in a if '/batched/' not in k) == sorted(k for k in b if '/batched/' not in k)
    for k in a:
        if k in b:
            assert a[k] == b[k], k


# ---- GPU ----------------------------------------------------------------------------------------------------------------

def _golden_checks(tree, case, asm, hap):
    _golden_merged_tables(tree, case, asm, hap)
    for name in rules.FLAG_OUTPUTS:
        key = f'results/{asm}/inv_caller/flagged_regions_{hap}.bed.gz' if name == 'flagged_regions' else f'temp/{asm}/inv_caller/flag/{name}_{hap}.bed.gz'
        with open(os.path.join(GOLD, case, name + '.tsv'), 'rb') as fh:
            assert tree[key] == fh.read(), key


@pytest.mark.gpu
@pytest.mark.parametrize('split', [False, True], ids=['whole-haplotypes', 'one-haplotype-shared'])
def test_two_ranks_on_one_gpu_equal_the_unsharded_run_and_the_reference_tables(built, tmp_path, split):
    """The real engine, two rank processes sharing GPU 0 (gloo control plane): three haplotypes dealt to two ranks (one resident
    reference per rank, two haplotypes at a time on contexts that share it) / one haplotype shared by both ranks.  All files equal the one-rank
    run's; the merged SNV / INS-DEL tables and the five flag tables of sampleA h1 equal what the reference's rule bodies wrote
    (tests/golden/flag_hap)."""
    if split:
        jobs = [ce.golden_job('flag_hap', 'sampleA', 'h1')]
    else:
        jobs = [ce.golden_job('flag_hap', 'sampleA', 'h1'), ce.golden_job('flag_sparse', 'sampleA', 'h2'), ce.golden_job('flag_hap', 'sampleB', 'h1')]
    ref_fa = os.path.join(GOLD, 'flag_hap', 'ref.fa')
    cfg = dict(CFG, inv_sig_filter='single_cluster')
    one, two = tmp_path / 'one', tmp_path / 'two'
    m1 = cohort.run_cohort(jobs, 1, str(one), ref_fa, config=cfg)
    m2 = cohort.run_cohort(jobs, 2, str(two), ref_fa, config=dict(cfg, pav_amd_lanes=2), share_gpu=True, split=split, timeout=900)
    key = lambda m: (m['asm_name'], m['hap'], m['inv_calls'])   # noqa: E731
    assert sorted(map(key, m1)) == sorted(map(key, m2)) and len(m1) == len(jobs)
    assert {m['rank'] for m in m2} == ({0} if split else {0, 1})
    a, b = ce.tree_text(one), ce.tree_text(two)
    keep = lambda t: {k: v for k, v in t.items() if '/batched/' not in k}   # noqa: E731  (the unshared route writes no CALL_BATCH files)
    a, b = keep(a), keep(b)
    assert sorted(a) == sorted(b)
    for k in a:
        assert a[k] == b[k], k
    assert any('/density_table/' in k for k in a)
    # (sampleA h2 = the contigs of flag_sparse against THIS reference: a second, different haplotype for the equality above; its
    #  golden tables were made with another reference and do not apply)
    # (single_cluster changes TRY_INV / BATCH of the flagged regions only; the other tables are the default configuration's)
    tree = {k: v for k, v in b.items()}
    cfg_default = tmp_path / 'default'
    cohort.run_cohort(jobs[:1], 1, str(cfg_default), ref_fa, config={})          # the reference's defaults: 60 batches, svindel
    _golden_checks(ce.tree_text(cfg_default), 'flag_hap', 'sampleA', 'h1')
    assert tree['temp/sampleA/cigar/merged/snv_snv_h1.bed.gz'] == ce.tree_text(cfg_default)['temp/sampleA/cigar/merged/snv_snv_h1.bed.gz']
