"""GPU parity of the inversion-signature flagging path against the tables the reference's own rule bodies produced
(tests/golden/flag_*; generator tools/refharness/gen_golden_flag.py)."""
import io
import os

import numpy as np
import pandas as pd
import pytest

from pav_amd import flag, rules

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
CASES = ['flag_hap', 'flag_sparse', 'flag_empty']


def golden_text(case, name):
    with open(os.path.join(GOLD, case, name + '.tsv')) as fh:
        return fh.read()


def as_text(df):
    buf = io.StringIO()
    df.to_csv(buf, sep='\t', index=False)
    return buf.getvalue()


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('vartype', ['snv', 'indel'])
def test_rule_call_inv_cluster(gpu_ctx, case, vartype, tmp_path):
    src = os.path.join(GOLD, case, 'snv_snv.tsv.gz' if vartype == 'snv' else 'svindel_insdel.tsv.gz')
    out = tmp_path / 'cluster.bed.gz'
    rules.call_inv_cluster([src], vartype, bed_out=str(out), ctx=gpu_ctx)
    assert pd.read_csv(out, sep='\t', dtype=str).to_csv(sep='\t', index=False) == golden_text(case, f'cluster_{vartype}')


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('vartype', ['sv', 'indel'])
def test_rule_call_inv_flag_insdel_cluster(gpu_ctx, case, vartype, tmp_path):
    out = tmp_path / 'insdel.bed.gz'
    rules.call_inv_flag_insdel_cluster(os.path.join(GOLD, case, 'svindel_insdel.tsv.gz'), vartype, bed_out=str(out), ctx=gpu_ctx)
    assert pd.read_csv(out, sep='\t', dtype=str).to_csv(sep='\t', index=False) == golden_text(case, f'insdel_{vartype}')


@pytest.mark.parametrize('case', CASES)
def test_rule_call_inv_merge_flagged_loci(gpu_ctx, case, tmp_path):
    out = tmp_path / 'flagged.bed.gz'
    src = [os.path.join(GOLD, case, n + '.tsv') for n in ('insdel_sv', 'insdel_indel', 'cluster_indel', 'cluster_snv')]
    df = rules.call_inv_merge_flagged_loci(*src, bed_out=str(out), ctx=gpu_ctx)
    assert as_text(df) == golden_text(case, 'flagged_regions')
    assert pd.read_csv(out, sep='\t', dtype=str, keep_default_na=False).to_csv(sep='\t', index=False) == golden_text(case, 'flagged_regions')


@pytest.mark.parametrize('sig,expect', [('sv', 2), ('single_cluster', None), (None, None)])
def test_sig_filters(gpu_ctx, sig, expect):
    """TRY_INV of _call_inv_accept_flagged_region (call_inv.snakefile:56-79) for the other inv_sig_filter values."""
    src = [pd.read_csv(os.path.join(GOLD, 'flag_hap', n + '.tsv'), sep='\t') for n in ('insdel_sv', 'insdel_indel', 'cluster_indel', 'cluster_snv')]
    df = flag.merge_flagged(gpu_ctx, *src, inv_sig_filter=sig, batch_count=7)
    types = df['TYPE'].apply(lambda s: set(s.split(',')))
    if sig == 'sv':
        want = types.apply(lambda t: 'MATCH_SV' in t)
    elif sig == 'single_cluster':
        want = types.apply(lambda t: True)
    else:
        want = types.apply(lambda t: t not in ({'CLUSTER_SNV'}, {'CLUSTER_INDEL'}))
    assert list(df['TRY_INV']) == list(want)
    accepted = df.loc[df['TRY_INV'], 'BATCH'].to_numpy()
    assert list(accepted) == [i % 7 for i in range(len(accepted))]
    assert (df.loc[~df['TRY_INV'], 'BATCH'] == -1).all()
    with pytest.raises(RuntimeError, match='Unrecognized region filter'):
        flag.merge_flagged(gpu_ctx, *src, inv_sig_filter='bogus')


@pytest.mark.parametrize('case', ['flag_hap', 'flag_sparse'])
def test_fused_from_device_calls(gpu_ctx, case, tmp_path):
    """CIGAR call of every alignment row + pav_cigar_flag == the reference's rule chain (10 batches, merge, flag rules)."""
    d = os.path.join(GOLD, case)
    out = {n: str(tmp_path / (n + '.bed.gz')) for n in rules.FLAG_OUTPUTS}
    res = rules.call_inv_flag(os.path.join(d, 'align.tsv'), os.path.join(d, 'trim.tsv'), os.path.join(d, 'tig.fa'), os.path.join(d, 'ref.fa'),
                              out=out, ctx=gpu_ctx)
    for n in rules.FLAG_OUTPUTS:
        assert as_text(res[n]) == golden_text(case, n), n
        assert pd.read_csv(out[n], sep='\t', dtype=str, keep_default_na=False).to_csv(sep='\t', index=False) == golden_text(case, n), n
    snv = pd.read_csv(os.path.join(d, 'snv_snv.tsv.gz'), sep='\t')
    assert res['n_snv_pass'] == int((snv['FILTER'] == 'PASS').sum())


def test_fused_with_alignment_rows_in_another_order(gpu_ctx, tmp_path):
    """The cluster keys of a get_align_bed table arrive in the rules' order and are only compacted; with the rows of the table
    reversed they do not, and the general path (one radix sort) must give the same tables."""
    d = os.path.join(GOLD, 'flag_hap')
    df = pd.read_csv(os.path.join(d, 'align.tsv'), sep='\t', dtype=str, keep_default_na=False)
    assert df.shape[0] > 3
    shuffled = str(tmp_path / 'align_reversed.tsv')
    df.iloc[::-1].to_csv(shuffled, sep='\t', index=False)
    res = rules.call_inv_flag(shuffled, os.path.join(d, 'trim.tsv'), os.path.join(d, 'tig.fa'), os.path.join(d, 'ref.fa'), ctx=gpu_ctx)
    for n in rules.FLAG_OUTPUTS:
        assert as_text(res[n]) == golden_text('flag_hap', n), n


def test_long_clusters_and_ties(gpu_ctx):
    """Clusters far longer than the serial search (wave path), chromosome switches inside a wave, decreasing midpoints."""
    rng = np.random.default_rng(5)
    chrom, pos, end = [], [], []
    for c in range(3):
        p = 1000
        for _ in range(40):
            run = int(rng.choice([1, 3, 31, 32, 33, 64, 65, 700, 5000]))
            for _ in range(run):
                p += int(rng.integers(0, 150))
                chrom.append(c); pos.append(p); end.append(p + int(rng.integers(1, 50)))
            p += 200 + 50 + int(rng.integers(0, 300))
    chrom, pos, end = np.array(chrom, dtype=np.uint32), np.array(pos), np.array(end)
    order = np.lexsort((pos, chrom))
    chrom, pos, end = chrom[order], pos[order], end[order]
    rec = gpu_ctx.flag_cluster(chrom, pos, end, 200, 200, 10)
    # the rule's loop (call_inv.snakefile:646-684), restated
    mid = (end + pos) // 2
    want, cur = [], None
    for c, m in zip(chrom, mid):
        if cur is not None and m < cur[2] + 200 and c == cur[0]:
            cur[3] += 1; cur[2] = m
        else:
            if cur is not None and cur[3] >= 10 and cur[2] - cur[1] >= 200:
                want.append(tuple(cur))
            cur = [c, m, m, 1]
    if cur is not None and cur[3] >= 10 and cur[2] - cur[1] >= 200:
        want.append(tuple(cur))
    got = [(int(r['chrom']), int(r['pos']), int(r['end']), int(r['count'])) for r in rec]
    assert got == [tuple(int(x) for x in w) for w in want]
    assert max(w[3] for w in want) > 1000
