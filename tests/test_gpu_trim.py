"""GPU-side parity of alignment trimming against tables produced by the reference's own rule bodies and function
(tests/golden/trim_*; generator tools/refharness/gen_golden_trim.py)."""
import gzip
import io
import json
import os

import pandas as pd
import pytest

from pav_amd import rules
from pav_amd.align import trim_alignment_record, trim_alignments

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
CASES = [('trim_overlap', 1000), ('trim_dense', 1500), ('trim_split', 1000)]


def gz_text(path):
    with gzip.open(path, 'rt') as fh:
        return fh.read()


def as_text(df):
    buf = io.StringIO()
    df.to_csv(buf, sep='\t', index=False)
    return buf.getvalue()


@pytest.mark.parametrize('case,min_len', CASES)
def test_rule_align_trim_tig(gpu_ctx, case, min_len, tmp_path):
    d = os.path.join(GOLD, case)
    out = tmp_path / 'trim_tig.bed.gz'
    rules.align_trim(os.path.join(d, 'align_none.tsv.gz'), os.path.join(d, 'tig.fa.fai'), 'tig', bed_out=str(out), min_trim_tig_len=min_len,
                     ctx=gpu_ctx)
    assert gz_text(out) == gz_text(os.path.join(d, 'trim_tig.tsv.gz'))


@pytest.mark.parametrize('case,min_len', CASES)
@pytest.mark.parametrize('redundant', [False, True])
def test_rule_align_trim_tigref(gpu_ctx, case, min_len, redundant, tmp_path):
    d = os.path.join(GOLD, case)
    out = tmp_path / 'trim_tigref.bed.gz'
    rules.align_trim(os.path.join(d, 'trim_tig.tsv.gz'), os.path.join(d, 'tig.fa.fai'), 'ref', bed_out=str(out), min_trim_tig_len=min_len,
                     redundant_callset=redundant, ctx=gpu_ctx)
    assert gz_text(out) == gz_text(os.path.join(d, 'trim_tigref_redundant.tsv.gz' if redundant else 'trim_tigref.tsv.gz'))


@pytest.mark.parametrize('case,min_len', CASES)
def test_trim_alignments_both(gpu_ctx, case, min_len):
    d = os.path.join(GOLD, case)
    df = trim_alignments(pd.read_csv(os.path.join(d, 'align_none.tsv.gz'), sep='\t', dtype={'#CHROM': str}), min_len,
                         os.path.join(d, 'tig.fa.fai'), mode='both', ctx=gpu_ctx)
    assert as_text(df) == gz_text(os.path.join(d, 'trim_both.tsv.gz'))
    with pytest.raises(RuntimeError, match='Unrecognized trimming mode'):
        trim_alignments(df, min_len, os.path.join(d, 'tig.fa.fai'), mode='sideways', ctx=gpu_ctx)


def test_trim_alignment_record_known_answers(gpu_ctx):
    with open(os.path.join(GOLD, 'trim_kat.json')) as fh:
        items = json.load(fh)
    n_err = 0
    for it in items:
        l, r = pd.Series(it['l']), pd.Series(it['r'])
        if 'error' in it:
            n_err += 1
            with pytest.raises(RuntimeError) as ei:
                trim_alignment_record(l, r, it['match_coord'], rev_l=it['rev_l'], rev_r=it['rev_r'], ctx=gpu_ctx)
            assert str(ei.value) == it['error'][1]
            continue
        a, b = trim_alignment_record(l, r, it['match_coord'], rev_l=it['rev_l'], rev_r=it['rev_r'], ctx=gpu_ctx)
        assert json.loads(a.to_json()) == it['out_l']
        assert json.loads(b.to_json()) == it['out_r']
    assert n_err == 5


def test_trimmed_table_has_no_overlaps(gpu_ctx):
    """Size-independent property on a larger table than the fixtures: after mode='both' no two records of a contig share
    contig bases and no two records share reference bases; spans still match the CIGARs (check_record ran)."""
    from pav_amd import synth
    hap = synth.config2(seed=77, scale=0.02, threads=4)
    df0 = synth.split_overlaps(hap.df_align, 77)
    df = trim_alignments(df0, 1000, hap.tig_lengths, mode='both', ctx=gpu_ctx)
    assert 0 < df.shape[0] <= df0.shape[0]
    for key, a, b in (('QRY_ID', 'QRY_POS', 'QRY_END'), ('#CHROM', 'POS', 'END')):
        for _, g in df.groupby(key):
            g = g.sort_values(a)
            assert (g[a].to_numpy()[1:] >= g[b].to_numpy()[:-1]).all(), key
    trimmed = df[['TRIM_QRY_L', 'TRIM_QRY_R']].to_numpy().sum()
    assert trimmed > 0
    # idempotence: a trimmed table is a fixed point
    again = trim_alignments(df, 1000, hap.tig_lengths, mode='both', ctx=gpu_ctx)
    assert as_text(again) == as_text(df)


def test_device_passes_equal_the_host_loops(gpu_ctx, monkeypatch):
    """The pair loops run on the device (csrc/trim_dev.hip: a wave per contig / chromosome group, wave-parallel
    trace_cigar_to_zero and find_cut_sites); the same loops on one host thread (PAV_TRIM_HOST=1) are kept as the cross-check.
    A seeded table with thousands of overlapping pieces - containment, both strands, records that also overlap on the
    reference (both trim orders tried), chromosome groups of hundreds of rows - must come out identical, table and CIGARs."""
    from pav_amd import synth
    hap = synth.config2(seed=77, scale=0.05, threads=4)
    df0 = synth.split_overlaps(hap.df_align, 77)
    assert df0.shape[0] > 1.5 * hap.df_align.shape[0]
    out = {}
    for name, env in (('device', None), ('host', '1')):
        if env is None:
            monkeypatch.delenv('PAV_TRIM_HOST', raising=False)
        else:
            monkeypatch.setenv('PAV_TRIM_HOST', env)
        tig = trim_alignments(df0, 1000, hap.tig_lengths, mode='tig', ctx=gpu_ctx)
        ref = trim_alignments(tig, 1000, hap.tig_lengths, mode='ref', ctx=gpu_ctx)
        both = trim_alignments(df0, 1000, hap.tig_lengths, mode='both', match_tig=True, ctx=gpu_ctx)
        out[name] = tuple(d.to_csv(sep='\t', index=False) for d in (tig, ref, both))
    assert out['device'] == out['host']
    assert out['device'][0] != df0.to_csv(sep='\t', index=False)              # something was trimmed


@pytest.mark.parametrize('which', ['Illegal operation', 'Found no cut-sites'])
def test_a_failing_pass_reports_the_same_pair_on_both_paths_and_poisons_the_table(gpu_ctx, monkeypatch, which):
    """A pass over a table in which ONE pair cannot be trimmed (an M operation inside the overlap / no cut site: the error
    records of tests/golden/trim_kat.json put behind a pair that trims fine): the device pass and the host loops raise the
    reference's message with the same error record (kind, rows, operation), and the loaded table is undefined afterwards on
    both paths - pav_trim_fetch / pav_trim_pass answer PAV_E_STATE until the table is loaded again."""
    from pav_amd import _lib
    from pav_amd.align import trim as trimmod
    with open(os.path.join(GOLD, 'trim_kat.json')) as fh:
        items = json.load(fh)
    bad = next(it for it in items if 'error' in it and which in it['error'][1] and it['match_coord'] == 'query')
    good = next(it for it in items if 'error' not in it and it['match_coord'] == 'query' and it['l']['QRY_ID'] == it['r']['QRY_ID']
                and it['l']['QRY_ID'] != bad['l']['QRY_ID'])
    rows = [dict(good['l']), dict(good['r']), dict(bad['l']), dict(bad['r'])]
    for i, r in enumerate(rows):
        r['INDEX'] = i
        for c in ('TRIM_REF_L', 'TRIM_REF_R', 'TRIM_QRY_L', 'TRIM_QRY_R'):
            r[c] = 0
    df = pd.DataFrame(rows)
    fai = pd.Series({r['QRY_ID']: int(r['QRY_LEN']) for r in rows})
    seen = {}
    for name, env in (('device', None), ('host', '1')):
        if env is None:
            monkeypatch.delenv('PAV_TRIM_HOST', raising=False)
        else:
            monkeypatch.setenv('PAV_TRIM_HOST', env)
        with pytest.raises(RuntimeError) as ei:
            trim_alignments(df, 1, fai, mode='tig', ctx=gpu_ctx)
        cause = ei.value.__cause__ if isinstance(ei.value.__cause__, _lib.TrimDeviceError) else ei.value.__context__
        err = cause.detail
        rows_after, _, _ = gpu_ctx.trim_fetch(with_cigar=False, with_counts=False)
        seen[name] = (str(ei.value), err.kind, err.row_l, err.row_r, err.op_index, err.op_char,
                      rows_after[[err.row_l, err.row_r]].tobytes())
        with pytest.raises(_lib.PavDeviceError, match='undefined after a failed pass'):
            gpu_ctx.trim_fetch()
        with pytest.raises(_lib.PavDeviceError, match='undefined after a failed pass'):
            gpu_ctx.trim_pass([0, 1], _lib.TRIM_QUERY, 1)
    assert seen['device'] == seen['host'], seen
    assert which in seen['device'][0]
    trim_alignments(df.iloc[:2], 1, fai, mode='tig', ctx=gpu_ctx)             # a fresh load makes the context usable again
