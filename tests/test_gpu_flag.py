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


def test_planned_stage_equals_general_path(built, monkeypatch):
    """pav_cigar_flag planned on the device (ordered INS / DEL split, scan-free sweeps, one synchronisation) against the general
    path of round 2 (PAV_FLAG_HOST=1: counts read back, radix-sorted DEL rows, max-scan sweeps) on a synthetic haplotype whose
    planted inversions make clusters of thousands of rows - longer than a 256-row block and than what one lane walks."""
    from pav_amd import _lib, cigarcall, synth
    hap = synth.config2(seed=77, scale=0.03, threads=4, pair_frac=0.01)
    names = hap.ref.names
    ctx = _lib.Context(0)
    try:
        ctx.seq_load(_lib.PAV_ROLE_REF, names, [hap.ref.seqs[n] for n in names])
        ctx.seq_load(_lib.PAV_ROLE_TIG, hap.tig_names, [hap.tig_seqs[n] for n in hap.tig_names])
        ctx.cigar_load(*cigarcall.pack_alignments(hap.df_align, names, hap.tig_names))
        ctx.cigar_call()
        index = hap.df_align['INDEX'].to_numpy(dtype='int64')
        trim = hap.df_trim[['POS', 'END', 'INDEX']].set_index('INDEX').astype(int).reindex(list(index), fill_value=-1)
        tp, te = trim['POS'].to_numpy(dtype='int64'), trim['END'].to_numpy(dtype='int64')
        got = {}
        for mode in ('planned', 'general'):
            if mode == 'general':
                monkeypatch.setenv('PAV_FLAG_HOST', '1')
            tables, loci, cnt = ctx.cigar_flag(tp, te, ctx.flag_params(sig_filter=_lib.SIG_SINGLE_CLUSTER))
            got[mode] = ({k: v.copy() for k, v in tables.items()}, loci.copy(), dict(cnt))
        monkeypatch.delenv('PAV_FLAG_HOST')
        for name in got['planned'][0]:
            assert got['planned'][0][name].tobytes() == got['general'][0][name].tobytes(), name
        assert got['planned'][1].tobytes() == got['general'][1].tobytes() and got['planned'][2] == got['general'][2]
        snv = got['planned'][0]['cluster_snv']
        assert len(snv) > 5 and int(snv['count'].max()) > 1000 and len(got['planned'][0]['insdel_indel']) > 0
        # a second call on the same context (trim table and ranks already on the device) gives the same
        tables, loci, cnt = ctx.cigar_flag(tp, te, ctx.flag_params(sig_filter=_lib.SIG_SINGLE_CLUSTER))
        assert loci.tobytes() == got['planned'][1].tobytes()
        # ... and a changed trim table is noticed
        tp2 = tp.copy(); tp2[:] = np.iinfo(np.int64).max // 2
        tables, loci, cnt = ctx.cigar_flag(tp2, te, ctx.flag_params(sig_filter=_lib.SIG_SINGLE_CLUSTER))
        assert cnt['n_snv_pass'] == 0 and cnt['n_indel_pass'] == 0 and len(loci) == 0
    finally:
        ctx.close()


@pytest.mark.parametrize('kind', ['snv_only', 'indel_only', 'ins_only', 'one_row_each', 'long_cluster', 'many_hits'])
def test_planned_stage_on_one_sided_tables(built, monkeypatch, kind):
    """Tables with nothing in one of the branches - no INS / DEL rows at all, no SNV rows, insertions without a single deletion
    (the matches have nothing to search), one row of each - through the planned stage and through the general path."""
    from pav_amd import _lib
    rng = np.random.default_rng(11)
    n = {'long_cluster': 260_000, 'many_hits': 5_000_000}.get(kind, 60_000)
    ref = rng.integers(0, 4, n).astype(np.uint8)
    tig = ref.copy()
    ops = []                                                   # (code, length) over the reference; the contig is edited to match
    pos = 0
    tig_parts = []
    def eq(m):
        nonlocal pos
        ops.append('%d=' % m); tig_parts.append(ref[pos:pos + m]); pos += m
    def snv():
        nonlocal pos
        ops.append('1X'); tig_parts.append(np.array([(ref[pos] + 1) & 3], dtype=np.uint8)); pos += 1
    def ins(m):
        ops.append('%dI' % m); tig_parts.append(rng.integers(0, 4, m).astype(np.uint8))
    def dele(m):
        nonlocal pos
        ops.append('%dD' % m); pos += m
    eq(500)
    if kind == 'snv_only':
        for _ in range(400):
            snv(); eq(int(rng.integers(2, 9)))
    elif kind == 'indel_only':
        for i in range(300):
            (ins if i % 2 else dele)(int(rng.integers(4, 60))); eq(int(rng.integers(20, 60)))
    elif kind == 'ins_only':
        for _ in range(200):
            ins(int(rng.integers(4, 90))); eq(int(rng.integers(5, 40)))
    elif kind == 'long_cluster':                                # one cluster of 45 000 SNV rows (176 blocks of 256: the wave walks the
        for _ in range(300):                                    # block notes 64 at a time), short ones around it
            snv(); eq(int(rng.integers(2, 9)))
        eq(300)
        for _ in range(45_000):
            snv(); eq(int(rng.integers(1, 4)))
        eq(300)
        for _ in range(25):
            snv(); eq(3)
    elif kind == 'many_hits':                                   # more hits in every list than come back with the counters (4096 each)
        for g in range(4500):
            for _ in range(22):                                 # a cluster of 22 SNVs over 220 bp ...
                snv(); eq(9)
            eq(150)
            ins(int(rng.integers(5, 12))); eq(3); dele(int(rng.integers(5, 12)))              # ... and an INS next to a DEL
            eq(2600 if g % 10 == 9 else 300)                    # (matches closer than 2 kbp merge into one interval)
    else:
        snv(); eq(100); ins(7); eq(100); dele(9); eq(100)
    eq(500)
    tig = np.concatenate(tig_parts)
    lut = np.frombuffer(b'ACGT', dtype=np.uint8)
    ctx = _lib.Context(0)
    try:
        ctx.seq_load(_lib.PAV_ROLE_REF, ['chr1'], [lut[ref]])
        ctx.seq_load(_lib.PAV_ROLE_TIG, ['tig1'], [lut[tig]])
        aln = np.zeros(1, dtype=_lib.ALN_DTYPE)
        text = np.frombuffer(''.join(ops).encode(), dtype=np.uint8)
        ctx.cigar_load(aln, text, np.array([0, text.shape[0]], dtype=np.uint64))
        counts = ctx.cigar_call()
        assert (counts.n_snv == 0) == (kind in ('indel_only', 'ins_only')) and (counts.n_indel == 0) == (kind in ('snv_only', 'long_cluster'))
        tp, te = np.array([-1], dtype=np.int64), np.array([1 << 40], dtype=np.int64)
        got = {}
        for mode in ('planned', 'general'):
            if mode == 'general':
                monkeypatch.setenv('PAV_FLAG_HOST', '1')
            tables, loci, cnt = ctx.cigar_flag(tp, te, ctx.flag_params(sig_filter=_lib.SIG_SINGLE_CLUSTER))
            got[mode] = ({k: v.tobytes() for k, v in tables.items()}, loci.tobytes(), dict(cnt), {k: len(v) for k, v in tables.items()})
        assert got['planned'] == got['general']
        sizes = got['planned'][3]
        if kind == 'snv_only':
            assert sizes['cluster_snv'] >= 1 and sizes['insdel_indel'] == 0
        if kind == 'many_hits':
            assert sizes['cluster_snv'] == 4500 and sizes['insdel_indel'] == 449       # 450 merged intervals, the last never written (:575-583)
        if kind == 'long_cluster':
            rec = np.frombuffer(got['planned'][0]['cluster_snv'], dtype=_lib.FLAG_RGN_DTYPE)
            assert sorted(rec['count'].tolist()) == [300, 45_000]          # (the 25-row cluster behind them spans 100 bp: too short)
        if kind == 'indel_only':
            assert sizes['cluster_indel'] >= 1 and sizes['insdel_indel'] + sizes['insdel_sv'] >= 1
        if kind == 'ins_only':
            assert sizes['insdel_indel'] == 0 and sizes['insdel_sv'] == 0
    finally:
        ctx.close()
