"""
Inversion-signature flagging: the three rules between the CIGAR caller and the inversion scan.

* :func:`cluster_table`   <-> rule ``call_inv_cluster``              (rules/call_inv.snakefile:603-692)
* :func:`insdel_table`    <-> rule ``call_inv_flag_insdel_cluster``  (rules/call_inv.snakefile:480-599)
* :func:`merge_flagged`   <-> rule ``call_inv_merge_flagged_loci``   (rules/call_inv.snakefile:321-474)
* :func:`flag_from_calls` - all of them at once from the call records still resident on the device (``pav_cigar_flag``).

The per-variant sweeps run on the GPU through the C ABI (``pav_flag_cluster`` / ``pav_flag_insdel``); the functions here
only turn table columns into arrays and result records back into the reference's tables (same columns, same text when
written with ``to_csv(sep='\\t', index=False)``).  The reference's behaviour is kept where it looks accidental
(DESIGN.md lists the cases): the minimum cluster span is ``cluster_win`` (``inv_sig_cluster_win_min`` is never read), the
last open INS/DEL interval is not written, a merged locus ends at the END of the last row merged into it.
"""

import numpy as np
import pandas as pd

from . import _lib

CLUSTER_COLUMNS = ['#CHROM', 'POS', 'END', 'COUNT']
INSDEL_COLUMNS = ['#CHROM', 'POS', 'END']
LOCUS_COLUMNS = ['#CHROM', 'POS', 'END', 'ID', 'SVTYPE', 'SVLEN', 'TYPE', 'COUNT_INDEL', 'COUNT_SNV', 'TRY_INV', 'BATCH']

TYPE_NAMES = {_lib.FLAG_MATCH_SV: 'MATCH_SV', _lib.FLAG_MATCH_INDEL: 'MATCH_INDEL', _lib.FLAG_CLUSTER_INDEL: 'CLUSTER_INDEL',
              _lib.FLAG_CLUSTER_SNV: 'CLUSTER_SNV'}
SIG_FILTERS = {'svindel': _lib.SIG_SVINDEL, 'sv': _lib.SIG_SV, 'single_cluster': _lib.SIG_SINGLE_CLUSTER, None: _lib.SIG_NONE}


def sig_filter_code(inv_sig_filter):
    """inv_sig_filter -> PAV_SIG_*; unknown values raise as the rule does (call_inv.snakefile:354-355)."""
    try:
        return SIG_FILTERS[inv_sig_filter]
    except (KeyError, TypeError):
        raise RuntimeError(f'Unrecognized region filter: {inv_sig_filter} (must be "single_cluster", "svindel", or "sv")')


def chrom_ranks(*columns):
    """Rank of every chromosome value in sort order (the order ``sort_values('#CHROM')`` uses): -> (names, codes...)."""
    arrays = [np.asarray(c, dtype=object) if getattr(c, 'dtype', None) == object else np.asarray(c) for c in columns]
    if not arrays:
        return np.empty(0, dtype=object), []
    if len({a.dtype == object for a in arrays if a.size}) > 1:
        raise TypeError("'<' not supported between instances of 'str' and 'int'")       # what sort_values would raise
    nonempty = [a for a in arrays if a.size]
    if not nonempty:
        return np.empty(0, dtype=object), [np.empty(0, dtype=np.uint32) for _ in arrays]
    names, codes = np.unique(np.concatenate(nonempty), return_inverse=True)
    out, at = [], 0
    for a in arrays:
        out.append(codes[at:at + a.size].astype(np.uint32))
        at += a.size
    return names, out


def _rgn_frame(names, rec, columns):
    data = {'#CHROM': names[rec['chrom']] if len(rec) else np.empty(0, dtype=object), 'POS': rec['pos'], 'END': rec['end']}
    if 'COUNT' in columns:
        data['COUNT'] = rec['count']
    return pd.DataFrame(data, columns=columns)


def cluster_table(ctx, df, vartype, cluster_win=200, cluster_min_snv=20, cluster_min_indel=10):
    """Rule call_inv_cluster on a merged variant table (columns #CHROM POS END SVLEN FILTER; call_inv.snakefile:616-692)."""
    if vartype == 'indel':
        cluster_min = cluster_min_indel
    elif vartype == 'snv':
        cluster_min = cluster_min_snv
    else:
        raise RuntimeError('Bad variant type {}: Expected "indel" or "snv"')               # :626, text as in the reference
    keep = (df['FILTER'] == 'PASS').to_numpy()
    if vartype == 'indel':
        keep &= (df['SVLEN'] < 50).to_numpy()
    names, (code,) = chrom_ranks(df['#CHROM'].to_numpy()[keep])
    pos = df['POS'].to_numpy(dtype=np.int64)[keep]
    end = df['END'].to_numpy(dtype=np.int64)[keep]
    order = np.lexsort((pos, code))                                   # stable: sort_values(['#CHROM', 'POS']) (:637)
    rec = ctx.flag_cluster(code[order], pos[order], end[order], cluster_win, cluster_win, cluster_min)   # win_min: see module doc
    return _rgn_frame(names, rec, CLUSTER_COLUMNS)


def insdel_table(ctx, df, vartype, flank_cluster=2, flank_merge=2000, cluster_min_svlen=4):
    """Rule call_inv_flag_insdel_cluster on the merged INS/DEL table (call_inv.snakefile:492-599)."""
    svlen_min = cluster_min_svlen if vartype == 'indel' else 50
    svlen = df['SVLEN'].to_numpy(dtype=np.int64)
    keep = (df['FILTER'] == 'PASS').to_numpy() & (svlen >= svlen_min)
    if vartype == 'indel':
        keep &= svlen < 50
    svtype = df['SVTYPE'].to_numpy()
    names, (code,) = chrom_ranks(df['#CHROM'].to_numpy()[keep])
    pos, end, svlen, svtype = (a[keep] for a in (df['POS'].to_numpy(dtype=np.int64), df['END'].to_numpy(dtype=np.int64), svlen, svtype))
    is_ins, is_del = svtype == 'INS', svtype == 'DEL'
    rec = ctx.flag_insdel(code[is_ins], pos[is_ins], svlen[is_ins], code[is_del], pos[is_del], end[is_del], flank_cluster, flank_merge)
    return _rgn_frame(names, rec, INSDEL_COLUMNS)


def _rgn_records(df, code, with_count):
    rec = np.zeros(df.shape[0], dtype=_lib.FLAG_RGN_DTYPE)
    rec['chrom'] = code
    rec['pos'] = df['POS'].to_numpy(dtype=np.int64)
    rec['end'] = df['END'].to_numpy(dtype=np.int64)
    if with_count:
        rec['count'] = df['COUNT'].to_numpy(dtype=np.int64)
    return rec


def _locus_frame(names, loci):
    n = len(loci)
    chrom = names[loci['chrom']] if n else np.empty(0, dtype=object)
    pos, end = loci['pos'], loci['end']
    svlen = end - pos
    type_str = {m: ','.join(sorted(v for k, v in TYPE_NAMES.items() if m & k)) for m in range(16)}
    return pd.DataFrame({
        '#CHROM': chrom, 'POS': pos, 'END': end,
        'ID': np.array(['{}-{}-RGN-{}'.format(c, p, l) for c, p, l in zip(chrom, pos, svlen)], dtype=object),
        'SVTYPE': np.full(n, 'RGN', dtype=object), 'SVLEN': svlen,
        'TYPE': np.array([type_str[int(m)] for m in loci['type_mask']], dtype=object),
        'COUNT_INDEL': loci['count_indel'], 'COUNT_SNV': loci['count_snv'],
        'TRY_INV': loci['try_inv'].astype(bool), 'BATCH': loci['batch'].astype(np.int64),
    }, columns=LOCUS_COLUMNS)


def merge_flagged(ctx, df_insdel_sv, df_insdel_indel, df_cluster_indel, df_cluster_snv, flank=500, batch_count=60,
                  inv_sig_filter='svindel'):
    """Rule call_inv_merge_flagged_loci on the four flag tables (call_inv.snakefile:329-474)."""
    sig = sig_filter_code(inv_sig_filter)
    frames = [df_insdel_sv, df_insdel_indel, df_cluster_indel, df_cluster_snv]
    names, codes = chrom_ranks(*[f['#CHROM'].to_numpy() for f in frames])
    tables = [_rgn_records(f, c, i >= 2) for i, (f, c) in enumerate(zip(frames, codes))]
    loci = ctx.flag_merge_loci(tables, flank, batch_count, sig)
    return _locus_frame(names, loci)


def flag_from_calls(ctx, trim_pos, trim_end, inv_sig_filter='svindel', **config):
    """Everything above in one device pass over the records of the last ``ctx.cigar_call()`` (``pav_cigar_flag``).

    ``trim_pos`` / ``trim_end``: per alignment row, POS / END of its INDEX in the trimmed table (-1 when absent), the
    FILTER rule of ``call_cigar``.  ``config`` takes the pav_flag_params field names.  Returns a dict with the four flag
    tables (``insdel_sv`` ...), ``flagged_regions`` and the PASS counts."""
    params = ctx.flag_params(sig_filter=sig_filter_code(inv_sig_filter), **config)
    tables, loci, counts = ctx.cigar_flag(trim_pos, trim_end, params)
    names = np.array(sorted(ctx.seq_names(_lib.PAV_ROLE_REF)), dtype=object)              # rank -> name (pav_cigar_flag ranks)
    out = {name: _rgn_frame(names, rec, CLUSTER_COLUMNS if name.startswith('cluster') else INSDEL_COLUMNS)
           for name, rec in tables.items()}
    out['flagged_regions'] = _locus_frame(names, loci)
    out.update(counts)
    return out
