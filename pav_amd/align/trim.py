"""
Alignment trimming: mirror of ``pavlib.align.trim_alignments`` and ``trim_alignment_record``
(pavlib/align/trim.py:11-599; rules align_trim_tig / align_trim_tigref, rules/align.snakefile:54-97).

The DataFrame, the reference's sorts and the error texts live here; the pair loops (``trim_alignment_record``,
``trace_cigar_to_zero``, ``find_cut_sites``) run inside the library on CIGARs tokenised once on the device
(``pav_trim_load`` / ``pav_trim_pass`` / ``pav_trim_fetch``, csrc/trim.cpp).
"""

import numpy as np
import pandas as pd

from .. import _lib

_TRIM_COLS = (('pos', 'POS'), ('end', 'END'), ('qry_pos', 'QRY_POS'), ('qry_end', 'QRY_END'), ('index', 'INDEX'),
              ('trim_ref_l', 'TRIM_REF_L'), ('trim_ref_r', 'TRIM_REF_R'), ('trim_qry_l', 'TRIM_QRY_L'), ('trim_qry_r', 'TRIM_QRY_R'))


def _codes(values):
    """Equality codes of a column (names are only compared for equality by the pair loops)."""
    return pd.factorize(np.asarray(values, dtype=object), sort=False)[0].astype(np.uint32)


def _load(ctx, df):
    from .. import cigarcall
    n = df.shape[0]
    rows = np.zeros(n, dtype=_lib.TRIM_ROW_DTYPE)
    if n:
        rows['chrom'] = _codes(df['#CHROM'])
        rows['qry_id'] = _codes(df['QRY_ID'])
        for f, c in _TRIM_COLS:
            rows[f] = df[c].to_numpy(dtype=np.int64)
        rows['rev'] = df['REV'].to_numpy().astype(bool).astype(np.int32)
    cig = [str(c).encode() for c in df['CIGAR']]
    off = np.zeros(n + 1, dtype=np.uint64)
    if n:
        off[1:] = np.cumsum([len(c) for c in cig], dtype=np.uint64)
    text = np.frombuffer(b''.join(cig), dtype=np.uint8) if n else np.zeros(0, dtype=np.uint8)
    try:
        ctx.trim_load(rows, text, off)
    except _lib.CigarDeviceError as ex:
        if ex.detail is None:
            raise
        cigarcall._raise_reference_error(ex.detail, df)


def _raise_trim_error(detail, df, match_coord):
    """pav_trim_err -> the RuntimeError of trim_alignment_record / trace_cigar_to_zero for the same pair."""
    rec_l, rec_r = df.iloc[int(detail.row_l)], df.iloc[int(detail.row_r)]
    # POS in the messages is the record's current (possibly already trimmed) position: the caller refreshed `df`
    kind = detail.kind
    if kind == 1:      # trim.py:428-434, 445-451
        raise RuntimeError('Cannot trim to negative distance {}: {} ({}:{}) vs {} ({}:{}), match_coord={}'.format(
            detail.diff_bp, rec_l['QRY_ID'], rec_l['#CHROM'], rec_l['POS'], rec_r['QRY_ID'], rec_r['#CHROM'], rec_r['POS'], match_coord))
    if kind == 2:      # trim.py:436-441
        raise RuntimeError('Contigs are incorrectly ordered in subject space: {} ({}:{}) vs {} ({}:{}), match_coord={}'.format(
            rec_l['QRY_ID'], rec_l['#CHROM'], rec_l['POS'], rec_r['QRY_ID'], rec_r['#CHROM'], rec_r['POS'], match_coord))
    if kind == 3:      # trim.py:880-883
        rec = rec_r if detail.side else rec_l
        raise RuntimeError((
            'Illegal operation in contig alignment while trimming alignment: {}{} '
            '(start={}:{}): CIGAR operation #{}: Expected CIGAR op in "IDSH=X"'
        ).format(detail.op_len, chr(detail.op_char), rec['#CHROM'], rec['POS'], detail.op_index))
    if kind == 4:      # trim.py:465-470
        raise RuntimeError('Program bug: Found no cut-sites: {} (INDEX={}) vs {} (INDEX={}), match_coord={}'.format(
            rec_l['QRY_ID'], rec_l['INDEX'], rec_r['QRY_ID'], rec_r['INDEX'], match_coord))
    raise RuntimeError(f'unknown trimming error kind {kind}')


def _store(df, rows, cigars):
    """Write the library's row state back into the frame (positional)."""
    for f, c in _TRIM_COLS:
        df[c] = rows[f]
    mod = np.flatnonzero(rows['modified'])
    if len(mod):
        col = df['CIGAR'].to_numpy(dtype=object).copy()
        for i in mod:
            col[i] = cigars[i]
        df['CIGAR'] = col
    return df


def _run_pass(ctx, df, order, mode, min_trim_tig_len, match_tig):
    """``df`` holds the loaded rows in load order (positional == loaded row number)."""
    try:
        ctx.trim_pass(order, mode, min_trim_tig_len, match_tig)
    except _lib.TrimDeviceError as ex:
        if ex.detail is None:
            raise
        rows, _, _ = ctx.trim_fetch(with_cigar=False, with_counts=False)   # positions as they were when the pair failed
        cur = df.copy()
        for f, c in _TRIM_COLS:
            cur[c] = rows[f]
        _raise_trim_error(ex.detail, cur, 'query' if mode == _lib.TRIM_QUERY else 'subject')


def trim_alignment_record(record_l, record_r, match_coord, rev_l=True, rev_r=False, ctx=None, device_id=0):
    """``pavlib.align.trim_alignment_record`` (trim.py:357-599): -> (record_l_mod, record_r_mod) as Series."""
    if match_coord not in {'query', 'subject'}:
        raise RuntimeError('Unknown match_coord parameter: {}: Expected "query" or "subject"'.format(match_coord))
    df = pd.DataFrame([record_l, record_r]).reset_index(drop=True)
    own = ctx is None
    if own:
        ctx = _lib.Context(device_id)
    try:
        _load(ctx, df)
        try:
            ctx.trim_pair(0, 1, _lib.TRIM_QUERY if match_coord == 'query' else _lib.TRIM_SUBJECT, rev_l, rev_r)
        except _lib.TrimDeviceError as ex:
            if ex.detail is None:
                raise
            _raise_trim_error(ex.detail, df, match_coord)
        rows, _, cigars = ctx.trim_fetch(with_counts=False)
    finally:
        if own:
            ctx.close()
    out = []
    for i, rec in enumerate((record_l, record_r)):
        rec = rec.copy()
        for f, c in _TRIM_COLS:
            rec[c] = int(rows[f][i])
        if cigars[i] is not None:
            rec['CIGAR'] = cigars[i]
        out.append(rec)
    return out[0], out[1]


_CHECK_TEXT = {
    1: 'Duplicate S records (left) at index {op}', 2: 'Duplicate H records (left) at index {op}', 3: 'S record before H (left) at index {op}',
    4: 'Found clipped bases before last non-clipped CIGAR operation at operation {op} ({len}{char})',
    5: 'Duplicate S records (right) at operation {op}', 6: 'H record before S record (right) at operation {op}',
    7: 'Duplicate H records (right) at operation {op}', 8: 'CIGAR op "M" is not allowed', 9: 'Bad CIGAR op: {char}',
}


def check_record(row, cnt, df_tig_fai):
    """``pavlib.align.check_record`` (align.py:364-509) with ``count_cigar`` taken from the library (pav_trim_count)."""
    where = '(INDEX={}, QRY={}:{}-{}, REF={}:{}-{})'.format(row['INDEX'], row['QRY_ID'], row['QRY_POS'], row['QRY_END'], row['#CHROM'],
                                                            row['POS'], row['END'])
    if cnt['err_kind']:
        msg = _CHECK_TEXT[int(cnt['err_kind'])].format(op=int(cnt['err_op']), len=int(cnt['err_len']), char=chr(int(cnt['err_char']) or 63))
        raise RuntimeError('CIGAR parsing error: {} {}'.format(msg, where))
    ref_bp, tig_bp = int(cnt['ref_bp']), int(cnt['tig_bp'])
    tig_len = df_tig_fai[row['QRY_ID']]
    if row['QRY_LEN'] != tig_len:
        raise RuntimeError('QRY_LEN != length from FAI ({} != {}) {}'.format(row['QRY_LEN'], tig_len, where))
    if row['QRY_POS'] >= row['QRY_END']:
        raise RuntimeError('QRY_POS >= QRY_END ({} >= {}) {}'.format(row['QRY_POS'], row['QRY_END'], where))
    if row['POS'] >= row['END']:
        raise RuntimeError('POS >= END ({} >= {}) {}'.format(row['POS'], row['END'], where))
    if row['POS'] < 0:
        raise RuntimeError('POS ({}) < 0 {}'.format(row['POS'], where))
    if row['QRY_POS'] < 0:
        raise RuntimeError('QRY_POS ({}) < 0 {}'.format(row['QRY_POS'], where))
    if row['POS'] + ref_bp != row['END']:
        raise RuntimeError('END mismatch: POS + ref_bp != END ({} != {}) {}'.format(row['POS'] + ref_bp, row['END'], where))
    if row['QRY_POS'] + tig_bp != row['QRY_END']:
        raise RuntimeError('QRY_POS + tig_bp != QRY_END: {} != {} {}'.format(row['QRY_POS'] + tig_bp, row['QRY_END'], where))
    if row['QRY_END'] > tig_len:
        raise RuntimeError('QRY_END > tig_len ({} > {}) {}'.format(row['QRY_END'], tig_len, where))


def check_records(df, counts, df_tig_fai):
    """``check_record`` for every row of ``df`` (``counts``: the rows' pav_trim_count records, same order), as the loop of
    trim.py:352-353 would raise: the first row that fails, with the first of its checks that fails.  The tests run as array
    expressions; the one offending row goes through :func:`check_record` for its message."""
    n = df.shape[0]
    if n == 0:
        return
    col = {c: df[c].to_numpy(dtype=np.int64) for c in ('QRY_LEN', 'QRY_POS', 'QRY_END', 'POS', 'END')}
    tig_len = df['QRY_ID'].map(df_tig_fai)
    if tig_len.isna().any():
        i = int(np.flatnonzero(tig_len.isna().to_numpy())[0])
        bad_first = i
    else:
        tl = tig_len.to_numpy(dtype=np.int64)
        bad = (counts['err_kind'] != 0) | (col['QRY_LEN'] != tl) | (col['QRY_POS'] >= col['QRY_END']) | (col['POS'] >= col['END']) | \
              (col['POS'] < 0) | (col['QRY_POS'] < 0) | (col['POS'] + counts['ref_bp'] != col['END']) | \
              (col['QRY_POS'] + counts['tig_bp'] != col['QRY_END']) | (col['QRY_END'] > tl)
        if not bad.any():
            return
        bad_first = int(np.flatnonzero(bad)[0])
    check_record(df.iloc[bad_first], counts[bad_first], df_tig_fai)       # raises with the reference's message


def trim_alignments(df, min_trim_tig_len, tig_fai, match_tig=False, mode='both', ctx=None, device_id=0):
    """Same arguments and result as ``pavlib.align.trim_alignments`` (trim.py:11-354); ``tig_fai`` may be a path or a
    Series of contig lengths."""
    from .. import fasta
    if mode is None:
        mode = 'both'
    mode = mode.lower()
    if mode == 'tig':
        do_trim_tig, do_trim_ref = True, False
    elif mode == 'ref':
        do_trim_tig, do_trim_ref = False, True
    elif mode == 'both':
        do_trim_tig, do_trim_ref = True, True
    else:
        raise RuntimeError(f'Unrecognized trimming mode "{mode}": Expected "tig", "ref", or "both"')

    # Remove short alignments (:51-57)
    df = df.copy()
    df.loc[(df['QRY_END'] - df['QRY_POS']) < min_trim_tig_len, 'INDEX'] = -1
    df = df.loc[df['INDEX'] >= 0].copy()

    own = ctx is None
    if own:
        ctx = _lib.Context(device_id)
    counts = None                     # count_cigar of every record's current CIGAR, from the pass that ran last (pav_trim_count)
    try:
        if do_trim_tig:
            df.sort_values(['QRY_ID', 'QRY_LEN'], ascending=(True, False), inplace=True)     # :64
            df.reset_index(inplace=True, drop=True)
            _load(ctx, df)
            _run_pass(ctx, df, np.arange(df.shape[0], dtype=np.uint32), _lib.TRIM_QUERY, min_trim_tig_len, False)
            rows, counts, cigars = ctx.trim_fetch(with_counts=not do_trim_ref)
            df = _store(df, rows, cigars)
            df['_LOADED_ROW'] = np.arange(df.shape[0])
            df = df.loc[df['INDEX'] >= 0].copy()                                              # :253
        if do_trim_ref:
            df = df.loc[
                pd.concat([df['#CHROM'], df['END'] - df['POS']], axis=1).sort_values(['#CHROM', 0], ascending=(True, False)).index
            ].reset_index(drop=True)                                                          # :267-274
            _load(ctx, df)
            _run_pass(ctx, df, np.arange(df.shape[0], dtype=np.uint32), _lib.TRIM_SUBJECT, min_trim_tig_len, match_tig)
            rows, counts, cigars = ctx.trim_fetch(with_counts=True)
            df = _store(df, rows, cigars)
            df['_LOADED_ROW'] = np.arange(df.shape[0])
            df = df.loc[df['INDEX'] >= 0].copy()                                              # :333

        # Post trim formatting (:342-354)
        df = df.loc[df['INDEX'] >= 0].copy()
        df = df.loc[(df['END'] - df['POS']) > 0]
        df = df.loc[(df['QRY_END'] - df['QRY_POS']) > 0]
        df = df.sort_values(['#CHROM', 'POS', 'END', 'QRY_ID'], ascending=[True, True, False, True])

        df_tig_fai = tig_fai if isinstance(tig_fai, pd.Series) else fasta.read_fai(tig_fai)
        # check_record of every row (:352-353): count_cigar of a record's final CIGAR is what the library holds for the row since
        # the last pass - the table is not loaded and tokenised a third time for it
        loaded = df.pop('_LOADED_ROW').to_numpy() if '_LOADED_ROW' in df else None
        if df.shape[0]:
            check_records(df, counts[loaded], df_tig_fai)
    finally:
        if own:
            ctx.close()
    return df
