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
