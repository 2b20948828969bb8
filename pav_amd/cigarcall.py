"""
Call variants by CIGAR string - host side of the MI355X path.

Mirror of ``pavlib/cigarcall.py`` (PAV 2.4.6): same function name, arguments, return frames and errors, so the
Snakemake rule ``call_cigar`` (rules/call.snakefile:792-846) can swap ``pavlib.cigarcall`` for this module.
The per-row CIGAR walk, the left-shift and the breakpoint-homology scans run on the GPU through
``libpav_amd.so`` (include/pav_amd.h); this module only (a) hands byte buffers to the library and (b) turns the
integer record stream back into the reference's tables (string columns, sort order).  No CPU fallback exists.
"""

import numpy as np
import pandas as pd

from . import _lib
from .fasta import open_fasta

# Tag variants called with this source (pavlib/cigarcall.py:19)
CALL_SOURCE = 'CIGAR'

CALL_CIGAR_BATCH_COUNT = 10  # pavlib/cigarcall.py:21

SNV_COLUMNS = ['#CHROM', 'POS', 'END', 'ID', 'SVTYPE', 'SVLEN', 'REF', 'ALT', 'HAP', 'QRY_REGION', 'QRY_STRAND',
               'CI', 'ALIGN_INDEX', 'CALL_SOURCE']                              # pavlib/cigarcall.py:125-134
INSDEL_COLUMNS = ['#CHROM', 'POS', 'END', 'ID', 'SVTYPE', 'SVLEN', 'HAP', 'QRY_REGION', 'QRY_STRAND', 'CI',
                  'ALIGN_INDEX', 'LEFT_SHIFT', 'HOM_REF', 'HOM_TIG', 'CALL_SOURCE', 'SEQ']   # :199-209


# ---------------------------------------------------------------------------------------------------------
# Input marshalling
# ---------------------------------------------------------------------------------------------------------

def pack_alignments(df_align, ref_names, tig_names):
    """Alignment BED rows -> (pav_aln array, concatenated CIGAR bytes, offsets) in table order.

    Reads exactly the columns the reference walk reads (pavlib/cigarcall.py:53-86): ``#CHROM``, ``POS``,
    ``QRY_ID``, ``REV``, ``CIGAR``.
    """
    n = df_align.shape[0]
    ref_index = {str(name): i for i, name in enumerate(ref_names)}
    tig_index = {str(name): i for i, name in enumerate(tig_names)}
    aln = np.zeros(n, dtype=_lib.ALN_DTYPE)
    if n:
        try:
            aln['ref_id'] = [ref_index[str(c)] for c in df_align['#CHROM']]
            aln['tig_id'] = [tig_index[str(q)] for q in df_align['QRY_ID']]
        except KeyError as ex:
            raise KeyError(f'sequence {ex} of the alignment table is not in the FASTA') from ex
        pos = df_align['POS'].to_numpy(dtype=np.int64)
        if pos.min() < 0 or pos.max() >= 0xFFFFFF00:
            raise ValueError('POS of the alignment table must be in [0, 2^32 - 256): got {} .. {}'.format(pos.min(), pos.max()))
        aln['pos'] = pos
        aln['rev'] = [1 if bool(v) else 0 for v in df_align['REV']]
    cigars = [str(c).encode() for c in df_align['CIGAR']] if n else []
    off = np.zeros(n + 1, dtype=np.uint64)
    if n:
        off[1:] = np.cumsum([len(c) for c in cigars], dtype=np.uint64)
    text = np.frombuffer(b''.join(cigars), dtype=np.uint8) if n else np.zeros(0, dtype=np.uint8)
    return aln, text, off


def _file_key(path):
    import os
    st = os.stat(path)
    return (os.path.realpath(str(path)), st.st_size, st.st_mtime_ns)


def load_reference(ctx, ref_fa_name):
    """Make every record of ``ref_fa_name`` resident on the context and remember it: later ``load_sequences`` calls with the
    same file keep the reference where it is (a cohort's haplotypes are called one after the other against ONE resident
    reference, ``pav_amd.cohort``; the packed planes of an hg38-sized reference are 4.3 GB and 0.3 s of work per upload)."""
    key = _file_key(ref_fa_name)
    if getattr(ctx, '_ref_resident', None) != key:
        if device_fasta():
            ctx.seq_load_fasta_path(_lib.PAV_ROLE_REF, ref_fa_name)      # text up as it is, header lines and line breaks removed on the device
        else:
            ref_fa = open_fasta(ref_fa_name)
            ctx.seq_load_fasta(_lib.PAV_ROLE_REF, ref_fa.native, ref_fa.record_numbers(ref_fa.names))
        ctx._ref_resident = key
    return ctx.seq_names(_lib.PAV_ROLE_REF)


def device_fasta():
    """Whole FASTA files go to the device without a host-side parse (``pav_seq_load_fasta_path``) unless ``PAV_FASTA_DEVICE=0``."""
    import os
    return os.environ.get('PAV_FASTA_DEVICE', '1') != '0'


def load_sequences(ctx, ref_fa_name, tig_fa_name, df_align=None, names=None):
    """Upload the records the alignment table touches (all records when ``df_align`` is None); ``names`` gives the two
    sets of record names directly.  A reference made resident by :func:`load_reference` stays (records are found by name)."""
    keep_ref = getattr(ctx, '_ref_resident', None) is not None and ctx._ref_resident == _file_key(ref_fa_name)
    all_records = names is None and not (df_align is not None and df_align.shape[0])
    if keep_ref and all_records and device_fasta():
        return list(ctx.seq_names(_lib.PAV_ROLE_REF)), list(ctx.seq_load_fasta_path(_lib.PAV_ROLE_TIG, tig_fa_name))
    if keep_ref:
        tig_fa = open_fasta(tig_fa_name)
        resident = ctx.seq_names(_lib.PAV_ROLE_REF)
        if names is not None or (df_align is not None and df_align.shape[0]):
            want_ref = {str(c) for c in (names[0] if names is not None else df_align['#CHROM'])}
            want_tig = {str(c) for c in (names[1] if names is not None else df_align['QRY_ID'])}
            tig_names = [n for n in tig_fa.names if n in want_tig]
            missing = (want_ref - set(resident)) | (want_tig - set(tig_names))
            if missing:
                raise KeyError(f'sequence(s) {sorted(missing)} of the alignment table are not in the FASTA files')
        else:
            tig_names = list(tig_fa.names)
        ctx.seq_load_fasta(_lib.PAV_ROLE_TIG, tig_fa.native, tig_fa.record_numbers(tig_names))
        return list(resident), tig_names
    ctx._ref_resident = None
    ref_fa = open_fasta(ref_fa_name)
    tig_fa = open_fasta(tig_fa_name)
    if names is not None or (df_align is not None and df_align.shape[0]):
        if names is not None:
            want_ref, want_tig = {str(c) for c in names[0]}, {str(c) for c in names[1]}
        else:
            want_ref = {str(c) for c in df_align['#CHROM']}
            want_tig = {str(c) for c in df_align['QRY_ID']}
        ref_names = [n for n in ref_fa.names if n in want_ref]
        tig_names = [n for n in tig_fa.names if n in want_tig]
        missing = (want_ref - set(ref_names)) | (want_tig - set(tig_names))
        if missing:
            raise KeyError(f'sequence(s) {sorted(missing)} of the alignment table are not in the FASTA files')
    else:
        ref_names, tig_names = list(ref_fa.names), list(tig_fa.names)
    ctx.seq_load_fasta(_lib.PAV_ROLE_REF, ref_fa.native, ref_fa.record_numbers(ref_names))   # straight from the reader's buffers
    ctx.seq_load_fasta(_lib.PAV_ROLE_TIG, tig_fa.native, tig_fa.record_numbers(tig_names))
    return ref_names, tig_names


# ---------------------------------------------------------------------------------------------------------
# Errors (same exception type and text as the reference)
# ---------------------------------------------------------------------------------------------------------

def _raise_reference_error(detail, df_align):
    """Map ``pav_cigar_err`` to what pavlib raises for the same input."""
    row = df_align.iloc[int(detail.aln)]
    kind = detail.kind
    if kind == 1:      # pavlib/cigarcall.py:292-299
        raise RuntimeError((
            'Illegal operation code in CIGAR string at operation {}: '
            'Alignments must be generated with =/X (not M): '
            'opcode={}, subject={}:{}, query={}:{}, align-index={}'
        ).format(detail.op_index, chr(detail.op_char), row['#CHROM'], detail.pos_ref, row['QRY_ID'], detail.pos_tig,
                 row['INDEX']))
    if kind == 2:      # pavlib/cigarcall.py:301-307
        raise RuntimeError((
            'Illegal operation code in CIGAR string at operation {}: '
            'opcode={}, subject={}:{} , query={}:{}, align-index={}'
        ).format(detail.op_index, chr(detail.op_char), row['#CHROM'], detail.pos_ref, row['QRY_ID'], detail.pos_tig,
                 row['INDEX']))
    if kind == 3:      # pavlib/align/align.py:310-313
        raise RuntimeError('Missing length in CIGAR string for contig {} alignment starting at {}:{}: CIGAR index {}'.format(
            row['QRY_ID'], row['#CHROM'], row['POS'], detail.op_index))
    if kind == 4:      # pavlib/align/align.py:315-318
        raise RuntimeError('Unknown CIGAR operation for contig {} alignment starting at {}:{}: CIGAR operation {}'.format(
            row['QRY_ID'], row['#CHROM'], row['POS'], chr(detail.op_char)))
    if kind == 5:      # ``cigar[len_pos]`` past the end of the string, pavlib/align/align.py:307
        raise IndexError('string index out of range')
    if kind == 6:
        raise RuntimeError('CIGAR operation length >= 2^28 for contig {} alignment starting at {}:{}: CIGAR index {}'.format(
            row['QRY_ID'], row['#CHROM'], row['POS'], detail.op_index))
    if kind == 7:      # the alignment row does not fit the sequences: ``seq_ref[pos_ref + i]`` past the end, cigarcall.py:104-105
        raise IndexError('string index out of range')
    raise RuntimeError(f'unknown CIGAR error kind {kind}')


# ---------------------------------------------------------------------------------------------------------
# Record stream -> reference tables
# ---------------------------------------------------------------------------------------------------------

def _obj(values):
    """Object-dtype array of Python scalars (the reference frames are all-object: built from pd.Series rows)."""
    out = np.empty(len(values), dtype=object)
    out[:] = values
    return out


def _str_ints(arr):
    return np.asarray(arr, dtype=np.int64).astype(str).astype(object)


def _sort_perm(chrom_obj, pos, end, id_obj):
    """Row order of ``df.sort_values(['#CHROM', 'POS', 'END', 'ID'])`` (pavlib/cigarcall.py:320,343).

    pandas lexsorts the four keys stably; #CHROM and ID compare as Python strings.  The numeric keys go through
    numpy's stable lexsort and only runs of equal (#CHROM, POS, END) are ordered by their ID strings."""
    n = len(pos)
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    chroms = sorted(set(chrom_obj))
    rank = {c: i for i, c in enumerate(chroms)}
    crank = np.fromiter((rank[c] for c in chrom_obj), dtype=np.int64, count=n)
    perm = np.lexsort((end, pos, crank))
    c, p, e = crank[perm], pos[perm], end[perm]
    same = (c[1:] == c[:-1]) & (p[1:] == p[:-1]) & (e[1:] == e[:-1])
    if not same.any():
        return perm
    # runs of ties
    starts = np.flatnonzero(np.concatenate(([True], ~same)))
    ends = np.concatenate((starts[1:], [n]))
    for s, t in zip(starts[(ends - starts) > 1], ends[(ends - starts) > 1]):
        idx = perm[s:t]
        ids = [id_obj[i] for i in idx]
        order = sorted(range(len(idx)), key=lambda k: ids[k])     # stable
        perm[s:t] = idx[order]
    return perm


def records_to_frames(snv, indel, seq_blob, df_align, hap, sort=True):
    """Build ``(df_snv, df_insdel)`` exactly as pavlib/cigarcall.py:313-362 returns them."""
    chrom_col = df_align['#CHROM'].to_numpy(dtype=object) if df_align.shape[0] else np.zeros(0, dtype=object)
    qry_col = df_align['QRY_ID'].to_numpy(dtype=object) if df_align.shape[0] else np.zeros(0, dtype=object)
    index_col = df_align['INDEX'].to_numpy(dtype=object) if df_align.shape[0] else np.zeros(0, dtype=object)
    strand_col = _obj(['-' if bool(v) else '+' for v in df_align['REV']]) if df_align.shape[0] else np.zeros(0, dtype=object)

    # ---- SNV (pavlib/cigarcall.py:112-135) --------------------------------------------------------------
    n = snv.shape[0]
    if n:
        a = snv['aln'].astype(np.int64)
        pos = snv['pos'].astype(np.int64)
        chrom = chrom_col[a]
        qry = qry_col[a]
        ref_c = snv['ref'].view('S1').astype(str).astype(object)
        alt_c = snv['alt'].view('S1').astype(str).astype(object)
        chrom_s = chrom.astype(str).astype(object)
        var_id = chrom_s + '-' + _str_ints(pos + 1) + '-SNV-' + np.char.upper(ref_c.astype(str)).astype(object) + \
            np.char.upper(alt_c.astype(str)).astype(object)
        q1 = _str_ints(snv['qry_pos'].astype(np.int64) + 1)
        qry_region = qry.astype(str).astype(object) + ':' + q1 + '-' + q1
        df_snv = pd.DataFrame({
            '#CHROM': chrom, 'POS': _obj(pos.tolist()), 'END': _obj((pos + 1).tolist()),
            'ID': var_id, 'SVTYPE': _obj(['SNV'] * n), 'SVLEN': _obj([1] * n),
            'REF': ref_c, 'ALT': alt_c, 'HAP': _obj([hap] * n),
            'QRY_REGION': qry_region, 'QRY_STRAND': strand_col[a],
            'CI': _obj([0] * n), 'ALIGN_INDEX': index_col[a], 'CALL_SOURCE': _obj([CALL_SOURCE] * n),
        }, columns=SNV_COLUMNS)
        if sort:
            df_snv = df_snv.iloc[_sort_perm(chrom, pos, pos + 1, var_id)]
    else:
        df_snv = pd.DataFrame([], columns=SNV_COLUMNS)                           # :323-335

    # ---- INS / DEL (pavlib/cigarcall.py:185-210, 254-279) -----------------------------------------------
    n = indel.shape[0]
    if n:
        a = indel['aln'].astype(np.int64)
        pos = indel['pos'].astype(np.int64)
        end = indel['end'].astype(np.int64)
        svlen = indel['svlen'].astype(np.int64)
        chrom = chrom_col[a]
        qry = qry_col[a]
        svtype = np.where(indel['svtype'] == 0, 'INS', 'DEL').astype(object)
        chrom_s = chrom.astype(str).astype(object)
        var_id = chrom_s + '-' + _str_ints(pos + 1) + '-' + svtype + '-' + _str_ints(svlen)
        qry_region = qry.astype(str).astype(object) + ':' + _str_ints(indel['qry_pos'].astype(np.int64) + 1) + '-' + \
            _str_ints(indel['qry_end'])
        hom_ref = _str_ints(indel['hom_ref_l']) + ',' + _str_ints(indel['hom_ref_r'])
        hom_tig = _str_ints(indel['hom_tig_l']) + ',' + _str_ints(indel['hom_tig_r'])
        blob = seq_blob.tobytes()
        off = indel['seq_off'].astype(np.int64)
        seq = _obj([blob[o:o + l].decode() for o, l in zip(off.tolist(), svlen.tolist())])
        df_insdel = pd.DataFrame({
            '#CHROM': chrom, 'POS': _obj(pos.tolist()), 'END': _obj(end.tolist()),
            'ID': var_id, 'SVTYPE': svtype, 'SVLEN': _obj(svlen.tolist()), 'HAP': _obj([hap] * n),
            'QRY_REGION': qry_region, 'QRY_STRAND': strand_col[a], 'CI': _obj([0] * n),
            'ALIGN_INDEX': index_col[a],
            'LEFT_SHIFT': _obj(indel['left_shift'].astype(np.int64).tolist()),
            'HOM_REF': hom_ref, 'HOM_TIG': hom_tig, 'CALL_SOURCE': _obj([CALL_SOURCE] * n), 'SEQ': seq,
        }, columns=INSDEL_COLUMNS)
        if sort:
            df_insdel = df_insdel.iloc[_sort_perm(chrom, pos, end, var_id)]
    else:
        df_insdel = pd.DataFrame([], columns=INSDEL_COLUMNS)                     # :346-359

    return df_snv, df_insdel


# ---------------------------------------------------------------------------------------------------------
# Public entry point (drop-in for pavlib.cigarcall.make_insdel_snv_calls)
# ---------------------------------------------------------------------------------------------------------

def call_records(ctx, df_align, ref_names=None, tig_names=None):
    """Run the device walk for ``df_align`` on sequences already resident in ``ctx``; return raw records."""
    ref_names = ctx.seq_names(_lib.PAV_ROLE_REF) if ref_names is None else ref_names
    tig_names = ctx.seq_names(_lib.PAV_ROLE_TIG) if tig_names is None else tig_names
    aln, text, off = pack_alignments(df_align, ref_names, tig_names)
    ctx.cigar_load(aln, text, off)
    try:
        counts = ctx.cigar_call()
    except _lib.CigarDeviceError as ex:
        if ex.detail is None:
            raise
        _raise_reference_error(ex.detail, df_align)
    snv, indel, blob = ctx.cigar_fetch(counts)
    return snv, indel, blob, counts


def make_insdel_snv_calls(df_align, ref_fa_name, tig_fa_name, hap, version_id=False, ctx=None, device_id=0):
    """
    Parse variants from CIGAR strings (GPU).  Same contract as pavlib/cigarcall.py:24-36.

    :param df_align: Post-cut BED of read alignments.
    :param ref_fa_name: Reference FASTA file name.
    :param tig_fa_name: Contig FASTA file name.
    :param hap: String identifying the haplotype ("h1", "h2").
    :param version_id: Version duplicate variant IDs if `True`.  DEFAULT DIFFERS FROM THE REFERENCE (`True` there,
        pavlib/cigarcall.py:24): the only caller, rule call_cigar, passes `False` (rules/call.snakefile:810), and `True`
        needs svpoplib.variant.version_id, which is not vendored in the reference snapshot (empty dep/svpop), so it
        raises NotImplementedError here - a default call must not fail.
    :param ctx: Optional live :class:`pav_amd._lib.Context` (sequences are (re)loaded into it).
    :param device_id: GPU to use when no context is given.

    :return: ``(df_snv, df_insdel)`` - note the order (pavlib/cigarcall.py:362), SNVs first.
    """
    if version_id:
        raise NotImplementedError(
            'version_id=True needs svpoplib.variant.version_id (un-vendored in the reference snapshot); '
            'call with version_id=False as rules/call.snakefile:810 does')
    own = ctx is None
    if own:
        ctx = _lib.Context(device_id)
    try:
        ref_names, tig_names = load_sequences(ctx, ref_fa_name, tig_fa_name, df_align)
        snv, indel, blob, _ = call_records(ctx, df_align, ref_names, tig_names)
    finally:
        if own:
            ctx.close()
    return records_to_frames(snv, indel, blob, df_align, hap)
