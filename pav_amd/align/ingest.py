"""
Alignment ingest: SAM -> the alignment table the rest of PAV works on.

Mirror of ``pavlib.align.get_align_bed`` (pavlib/align/align.py:666-794) without pysam: the SAM text is parsed by the library
(``pav_sam_open``, csrc/samio.cpp - record fields, soft clipping folded into hard clipping, ``count_cigar``), this module
applies the reference's per-record rules, raises its errors and builds its all-object DataFrame, sorted with the reference's
own ``sort_values`` call.
"""

import pandas as pd

from .. import _lib
from .trim import check_record

ALIGN_BED_COLUMNS = ['#CHROM', 'POS', 'END', 'INDEX', 'QRY_ID', 'QRY_POS', 'QRY_END', 'QRY_LEN', 'RG', 'AO', 'MAPQ', 'REV', 'FLAGS',
                     'HAP', 'CIGAR']


def _tag_value(kind, text):
    """Value of an optional SAM field as ``dict(record.get_tags())`` holds it: type ``i`` -> int, ``f`` -> float, else str."""
    if kind == 0:
        return 'NA'                                                     # align.py:758-759
    if kind == 1:
        return int(text)
    if kind == 3:
        return float(text)
    return text


def get_align_bed(align_file, df_tig_fai, hap, min_mapq=0, threads=0):
    """
    Read a SAM file (plain, gzip or BGZF text) as the alignment table PAV processes.  Drops records that are unmapped, below
    ``min_mapq`` or without CIGAR; ``INDEX`` counts every alignment line of the file (align.py:688-696).

    :param align_file: SAM file name.
    :param df_tig_fai: Series, contig name -> length.
    :param hap: Haplotype of this alignment file.
    :param min_mapq: Minimum MAPQ.

    :return: Alignment table (same columns, object dtype and row order as ``pavlib.align.get_align_bed``).
    """
    sam = _lib.SamFile(align_file, min_mapq=min_mapq, threads=threads)
    c = sam.cols
    index, pos, end = c['index'].tolist(), c['pos'].tolist(), c['end'].tolist()
    qas, qae, clip_h, map_pos = (c[k].tolist() for k in ('query_alignment_start', 'query_alignment_end', 'clip_h', 'tig_map_pos'))
    mapq, flag, has_m, status = c['mapq'].tolist(), c['flag'].tolist(), c['has_m'].tolist(), c['status'].tolist()
    rows = []
    for i in range(sam.n_rows):
        qry, chrom = sam.qry_names[c['qry_id'][i]], sam.ref_names[c['chrom_id'][i]]
        tig_len = df_tig_fai[qry]                                       # KeyError like the reference (:699)
        if status[i] == 1:
            raise ValueError('Invalid clipping in CIGAR string')        # pysam's query_alignment_start / _end
        if status[i] == 2:
            raise RuntimeError('Alignment record {} of {} consists of clipping operations only'.format(index[i], align_file))
        tig_map_pos = map_pos[i]
        tig_map_end = tig_map_pos + (qae[i] - qas[i])                  # :716-717
        if qas[i] + clip_h[i] != tig_map_pos:                           # :719-720
            raise RuntimeError(f'First aligned based from pysam ({qas[i]}) does not match clipping ({tig_map_pos}) at alignment record {index[i]}')
        if has_m[i]:                                                    # :725-729
            raise RuntimeError((
                'Found alignment match CIGAR operation (M) for record {} (Start = {}:{}): '
                'Alignment requires CIGAR base-level match/mismatch (=X)'
            ).format(qry, chrom, pos[i]))
        rev = bool(flag[i] & 16)
        rows.append([chrom, pos[i], end[i], index[i], qry,
                     tig_len - tig_map_end if rev else tig_map_pos, tig_len - tig_map_pos if rev else tig_map_end, tig_len,
                     _tag_value(c['rg_kind'][i], sam.rg[i]), _tag_value(c['ao_kind'][i], sam.ao[i]), mapq[i], rev,
                     f'0x{flag[i]:04x}', hap, sam.cigars[i]])
    # all-object frame, like pd.concat(list of Series, axis=1).T (:781-782); the empty frame keeps default dtypes (:783-797)
    df = pd.DataFrame(rows, columns=ALIGN_BED_COLUMNS, dtype=object) if rows else pd.DataFrame([], columns=ALIGN_BED_COLUMNS)
    order = {int(v): i for i, v in enumerate(index)}
    df.sort_values(['#CHROM', 'POS', 'END', 'QRY_ID'], ascending=[True, True, False, True], inplace=True)   # :799
    counts = {name: c[name] for name in ('ref_bp', 'tig_bp', 'err_kind', 'err_op', 'err_len', 'err_char')}
    for _, row in df.iterrows():                                        # :802 check_record on every row, table order
        i = order[int(row['INDEX'])]
        check_record(row, {k: v[i] for k, v in counts.items()}, df_tig_fai)
    df.attrs['sam_header'] = sam.header
    return df
