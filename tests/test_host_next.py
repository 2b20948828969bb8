"""CPU-side checks of the host layers of the trimming / flagging / large-SV rows: generators, table builders, error texts.
(The device paths are covered by tests/test_gpu_{trim,flag,lgsv}.py.)"""
import collections
import os

import numpy as np
import pandas as pd
import pytest

from pav_amd import _lib, flag, lgsv, synth
from pav_amd.align import trim as ptrim
from pav_amd.align.cigar import tokenize
from pav_amd.inv import IntervalSet

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def spans(cigar):
    lens, ops = tokenize(cigar)
    q = int(lens[np.isin(ops, np.frombuffer(b'=XI', dtype=np.uint8))].sum())
    r = int(lens[np.isin(ops, np.frombuffer(b'=XD', dtype=np.uint8))].sum())
    total = int(lens[np.isin(ops, np.frombuffer(b'=XISH', dtype=np.uint8))].sum())
    return q, r, total


def consistent(df):
    for _, row in df.iterrows():
        q, r, total = spans(row['CIGAR'])
        assert row['QRY_POS'] + q == row['QRY_END'] and row['POS'] + r == row['END'] and total == row['QRY_LEN'], row['INDEX']


def test_overlap_table_generator_is_consistent_and_deterministic():
    df, fai = synth.make_overlap_table(21)
    consistent(df)
    assert (df['QRY_LEN'] == df['QRY_ID'].map(fai)).all()
    golden = pd.read_csv(os.path.join(GOLD, 'trim_overlap', 'align_none.tsv.gz'), sep='\t', dtype={'#CHROM': str}, keep_default_na=False)
    assert golden['CIGAR'].tolist() == df['CIGAR'].tolist()            # the committed fixture is this seed's table
    overlaps = 0
    for _, g in df.groupby('QRY_ID'):
        g = g.sort_values('QRY_POS')
        overlaps += int((g['QRY_POS'].to_numpy()[1:] < g['QRY_END'].to_numpy()[:-1]).sum())
    assert overlaps > 20


def test_split_and_truncating_generators():
    hap = synth.config2(seed=41, scale=0.004, threads=2)
    split = synth.split_overlaps(hap.df_align, 3)
    consistent(split)
    assert split.shape[0] > hap.df_align.shape[0]
    trunc = synth.make_truncating_table(hap, 5)
    consistent(trunc)
    pairs = collections.Counter(trunc[['#CHROM', 'QRY_ID']].apply(tuple, axis=1))
    assert sum(1 for v in pairs.values() if v > 1) >= 5
    for _, g in trunc.groupby('QRY_ID'):                                # truncated records never share contig bases
        g = g.sort_values('QRY_POS')
        assert (g['QRY_POS'].to_numpy()[1:] >= g['QRY_END'].to_numpy()[:-1]).all()


def test_flag_host_helpers():
    names, (a, b) = flag.chrom_ranks(np.array(['chr2', 'chr10', 'chr1'], dtype=object), np.array(['chr10'], dtype=object))
    assert list(names) == ['chr1', 'chr10', 'chr2'] and list(a) == [2, 1, 0] and list(b) == [1]      # Python str order
    names, (a,) = flag.chrom_ranks(np.array([3, 1, 2]))
    assert list(names) == [1, 2, 3] and list(a) == [2, 0, 1]
    with pytest.raises(TypeError):
        flag.chrom_ranks(np.array(['x'], dtype=object), np.array([1]))
    assert flag.sig_filter_code('svindel') == _lib.SIG_SVINDEL and flag.sig_filter_code(None) == _lib.SIG_NONE
    with pytest.raises(RuntimeError, match='Unrecognized region filter: both'):
        flag.sig_filter_code('both')
    loci = np.zeros(2, dtype=_lib.FLAG_LOCUS_DTYPE)
    loci['chrom'], loci['pos'], loci['end'] = [0, 1], [10, 500], [60, 450]
    loci['type_mask'] = [_lib.FLAG_MATCH_SV | _lib.FLAG_CLUSTER_SNV, _lib.FLAG_CLUSTER_INDEL]
    loci['count_snv'], loci['count_indel'], loci['try_inv'], loci['batch'] = [25, 0], [0, 12], [1, 0], [0, -1]
    df = flag._locus_frame(np.array(['chrA', 'chrB'], dtype=object), loci)
    assert df.to_csv(sep='\t', index=False) == (
        '#CHROM\tPOS\tEND\tID\tSVTYPE\tSVLEN\tTYPE\tCOUNT_INDEL\tCOUNT_SNV\tTRY_INV\tBATCH\n'
        'chrA\t10\t60\tchrA-10-RGN-50\tRGN\t50\tCLUSTER_SNV,MATCH_SV\t0\t25\tTrue\t0\n'
        'chrB\t500\t450\tchrB-500-RGN--50\tRGN\t-50\tCLUSTER_INDEL\t12\t0\tFalse\t-1\n')
    empty = flag._locus_frame(np.empty(0, dtype=object), np.zeros(0, dtype=_lib.FLAG_LOCUS_DTYPE))
    assert empty.to_csv(sep='\t', index=False) == '\t'.join(flag.LOCUS_COLUMNS) + '\n'


def test_check_record_messages():
    row = pd.Series({'INDEX': 7, 'QRY_ID': 'tigA', 'QRY_POS': 10, 'QRY_END': 110, 'QRY_LEN': 500, '#CHROM': 'chr1', 'POS': 1000, 'END': 1100})
    fai = pd.Series({'tigA': 500})
    cnt = np.zeros(1, dtype=_lib.TRIM_COUNT_DTYPE)[0]
    cnt['ref_bp'], cnt['tig_bp'] = 100, 100
    ptrim.check_record(row, cnt, fai)
    where = '(INDEX=7, QRY=tigA:10-110, REF=chr1:1000-1100)'
    cnt['ref_bp'] = 99
    with pytest.raises(RuntimeError) as ei:
        ptrim.check_record(row, cnt, fai)
    assert str(ei.value) == 'END mismatch: POS + ref_bp != END (1099 != 1100) ' + where
    cnt['ref_bp'], cnt['err_kind'], cnt['err_op'], cnt['err_len'], cnt['err_char'] = 100, 4, 3, 25, ord('I')
    with pytest.raises(RuntimeError) as ei:
        ptrim.check_record(row, cnt, fai)
    assert str(ei.value) == ('CIGAR parsing error: Found clipped bases before last non-clipped CIGAR operation at operation 3 (25I) ' + where)
    with pytest.raises(RuntimeError, match='QRY_LEN != length from FAI \\(500 != 400\\)'):
        cnt['err_kind'] = 0
        ptrim.check_record(row, cnt, pd.Series({'tigA': 400}))


def test_check_records_raises_what_the_row_loop_would():
    """check_records (array expressions over the table) against the loop of check_record over its rows (trim.py:352-353): the
    same first failing row and message for every kind of failure, and silence on a clean table."""
    rng = np.random.default_rng(5)
    n = 40
    base = pd.DataFrame({'INDEX': np.arange(n), 'QRY_ID': ['tig%d' % (i % 3) for i in range(n)], 'QRY_POS': 10, 'QRY_END': 110, 'QRY_LEN': 500,
                         '#CHROM': 'chr1', 'POS': 1000 + 10 * np.arange(n), 'END': 1100 + 10 * np.arange(n)})
    fai = pd.Series({'tig0': 500, 'tig1': 500, 'tig2': 500})
    cnt0 = np.zeros(n, dtype=_lib.TRIM_COUNT_DTYPE)
    cnt0['ref_bp'], cnt0['tig_bp'] = 100, 100
    ptrim.check_records(base, cnt0, fai)                                  # clean

    def loop(df, cnt):
        for i in range(df.shape[0]):
            ptrim.check_record(df.iloc[i], cnt[i], fai)
    breaks = [('err_kind', lambda d, c, i: c.__setitem__('err_kind', np.where(np.arange(n) == i, 9, c['err_kind']))),
              ('QRY_LEN', lambda d, c, i: d.__setitem__('QRY_LEN', np.where(np.arange(n) == i, 400, d['QRY_LEN']))),
              ('QRY_POS>=END', lambda d, c, i: d.__setitem__('QRY_POS', np.where(np.arange(n) == i, 200, d['QRY_POS']))),
              ('POS>=END', lambda d, c, i: d.__setitem__('POS', np.where(np.arange(n) == i, 10 ** 6, d['POS']))),
              ('POS<0', lambda d, c, i: d.__setitem__('POS', np.where(np.arange(n) == i, -5, d['POS']))),
              ('ref_bp', lambda d, c, i: c.__setitem__('ref_bp', np.where(np.arange(n) == i, 99, c['ref_bp']))),
              ('tig_bp', lambda d, c, i: c.__setitem__('tig_bp', np.where(np.arange(n) == i, 101, c['tig_bp']))),
              ('QRY_END>len', lambda d, c, i: (d.__setitem__('QRY_END', np.where(np.arange(n) == i, 610, d['QRY_END'])),
                                               c.__setitem__('tig_bp', np.where(np.arange(n) == i, 600, c['tig_bp']))))]
    for name, brk in breaks:
        for _ in range(3):
            d, c = base.copy(), cnt0.copy()
            rows = sorted(set(int(x) for x in rng.integers(0, n, 3)))
            for i in rows:
                brk(d, c, i)
            with pytest.raises(RuntimeError) as want:
                loop(d, c)
            with pytest.raises(RuntimeError) as got:
                ptrim.check_records(d, c, fai)
            assert str(got.value) == str(want.value), name
            assert 'INDEX=%d,' % rows[0] in str(got.value)


def test_interval_set_and_lgsv_constants():
    t = IntervalSet()
    t[100:200] = True
    assert len(t[150:160]) == 1 and len(t[200:300]) == 0 and len(t[50:101]) == 1 and len(t[199]) == 1 and len(t[200]) == 0
    assert lgsv.match_bp({'CIGAR': '10='}, False) == 0
    assert lgsv.INSDEL_COLUMNS[11:14] == ['LEFT_SHIFT', 'HOM_REF', 'HOM_TIG'] and len(lgsv.INV_COLUMNS) == 20


@pytest.mark.parametrize('rel', ['cigar_synth/align.tsv', 'cigar_synth/trim.tsv', 'inv_hap/align.tsv', 'trim_overlap/align_none.tsv.gz', 'trim_split/trim_tigref.tsv.gz', 'flag_hap/align.tsv',
                                 'lgsv_hap/align.tsv.gz'])
def test_native_alignment_table_reader_equals_pandas(built, rel, tmp_path):
    """pav_bed_open (no GPU needed) against pandas.read_csv on committed alignment tables, gzip and plain."""
    path = os.path.join(GOLD, rel)
    if os.path.isdir(path):
        cand = [f for f in sorted(os.listdir(path)) if 'align' in f and (f.endswith('.tsv') or f.endswith('.gz') or f.endswith('.bed'))]
        assert cand, os.listdir(path)
        path = os.path.join(path, cand[0])
    df = pd.read_csv(path, sep='\t', dtype={'#CHROM': str, 'QRY_ID': str}, keep_default_na=False, low_memory=False)
    t = _lib.BedTable(path)
    cols = t.fetch()
    assert t.n_rows == df.shape[0]
    for c in ('POS', 'END', 'INDEX', 'QRY_POS', 'QRY_END', 'QRY_LEN', 'MAPQ', 'CALL_BATCH'):
        if c in df.columns:
            assert np.array_equal(cols[c], df[c].to_numpy(dtype=np.int64)), c
    assert [t.chrom_names[i] for i in cols['#CHROM']] == df['#CHROM'].tolist()
    assert [t.qry_names[i] for i in cols['QRY_ID']] == df['QRY_ID'].tolist()
    assert cols['REV'].tolist() == [bool(v) for v in df['REV']]
    text = cols['CIGAR_TEXT'].tobytes().decode()
    off = cols['CIGAR_OFF']
    assert [text[int(off[i]):int(off[i + 1])] for i in range(t.n_rows)] == df['CIGAR'].tolist()
    t.close()
    plain = tmp_path / 'plain.tsv'                                     # the same table uncompressed, CRLF line ends
    with open(plain, 'w', newline='') as fh:
        fh.write(df.to_csv(sep='\t', index=False).replace('\n', '\r\n'))
    t2 = _lib.BedTable(str(plain), with_cigar=False)
    assert t2.n_rows == df.shape[0] and 'CIGAR_TEXT' not in t2.fetch() and np.array_equal(t2.fetch()['POS'], cols['POS'])
    t2.close()


def test_native_reader_errors(built, tmp_path):
    p = tmp_path / 'bad.tsv'
    p.write_text('#CHROM\tPOS\tEND\tREV\tCIGAR\nchr1\t10\tx20\tTrue\t5=\n')
    with pytest.raises(_lib.PavDeviceError, match="cannot parse 'x20' in column END"):
        _lib.BedTable(str(p))
    p.write_text('#CHROM\tPOS\nchr1\t10\t99\n')
    with pytest.raises(_lib.PavDeviceError, match='has no CIGAR column'):
        _lib.BedTable(str(p))
    with pytest.raises(_lib.PavDeviceError, match='cannot open'):
        _lib.BedTable(str(tmp_path / 'missing.tsv.gz'))


def test_float_text_of_the_table_writers_equals_pandas(built):
    """pav_repr_f64 (float columns of pav_inv_write_tables) against DataFrame.to_csv and repr(): magnitudes from denormal to
    1e308, exact powers of ten around the positional / exponent switch, negative zero, NaN (empty na_rep), infinities."""
    import ctypes
    import io
    lib = _lib.load()
    buf = ctypes.create_string_buffer(64)
    rng = np.random.default_rng(9)
    vals = np.concatenate([
        [0.0, -0.0, 1.0, -1.5, 1e15, 1e16, 1e17, 1e-4, 1e-5, 9.999999999999999e-05, 5e-324, 1.7976931348623157e308,
         2.2250738585072014e-308, 0.1, 1 / 3, 1e22, 1e23, 9007199254740993.0, 99999999999999.98, np.nan, np.inf, -np.inf],
        rng.random(20000), rng.random(20000) * 1e-7, np.exp(rng.uniform(-700, 700, 20000)),
        rng.integers(0, 2 ** 63, 20000, dtype=np.uint64).view(np.float64)])
    want = pd.DataFrame({'x': vals}).to_csv(io.StringIO(), sep='\t', index=False, header=False)
    text = io.StringIO()
    pd.DataFrame({'x': vals}).to_csv(text, sep='\t', index=False, header=False)
    want = text.getvalue().split('\n')[:-1]
    got = []
    for v in vals:
        n = lib.pav_repr_f64(float(v), buf, 64)
        assert n >= 0
        got.append(buf.value.decode())
    bad = [(float(v), g, w) for v, g, w in zip(vals, got, want) if g != w and not (w == '""' and g == '')]
    assert not bad, bad[:5]
    assert got[0] == '0.0' and got[1] == '-0.0' and got[5] == '1e+16' and got[4] == '1000000000000000.0' and got[8] == '1e-05'


def test_effective_cpus_is_bounded_by_the_host_and_the_cgroup_quota():
    """pav_amd.shard.effective_cpus: what bench.py / tools size their host thread pools with."""
    import os
    from pav_amd.shard import effective_cpus
    n = effective_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            assert n <= max(1, int(quota) // int(period))
    except (OSError, ValueError):
        pass
