"""
File-contract mirrors of the two Snakemake rules that sit on the hot path, callable without Snakemake.

* :func:`call_cigar`        <-> rule ``call_cigar``        (rules/call.snakefile:792-846)
* :func:`call_cigar_merge`  <-> rule ``call_cigar_merge``  (rules/call.snakefile:755-786)
* :func:`call_inv_batch`    <-> rule ``call_inv_batch``    (rules/call_inv.snakefile:115-311)   [pav_amd.inv]

Same inputs, same outputs, same column order; the ``run:`` bodies can call these functions instead of pavlib
(INTEGRATION.md shows the two-line change).
"""

import pandas as pd

from . import cigarcall


def apply_trim_filter(df, df_trim):
    """FILTER = PASS iff POS > trim.POS and END < trim.END of the row's ALIGN_INDEX, else TRIM; alignment
    indexes absent from the trimmed table reindex to -1 and so give TRIM (rules/call.snakefile:813-842)."""
    df_pass = df_trim.reindex(list(df['ALIGN_INDEX']), fill_value=-1).set_index(df.index, drop=True)
    df['FILTER'] = ((df['POS'] > df_pass['POS']) & (df['END'] < df_pass['END'])).apply(
        lambda val: 'PASS' if val else 'TRIM')
    return df


def read_align_bed(path):
    """rules/call.snakefile:805"""
    return pd.read_csv(path, sep='\t', dtype={'#CHROM': str}, keep_default_na=False, low_memory=False)


def read_trim_bed(path):
    """rules/call.snakefile:813-816"""
    return pd.read_csv(path, sep='\t', usecols=['POS', 'END', 'INDEX'], index_col='INDEX').astype(int)


def call_cigar_files(bed, bed_trim, tig_fa_name, ref_fa_name, hap, batch, bed_insdel, bed_snv, ctx=None, device_id=0,
                     threads=0):
    """Body of rule call_cigar without pandas: the alignment tables are parsed by the library (``pav_bed_open``), the rows of
    ``batch`` go straight to the caller (``pav_cigar_load_bed``), FILTER, the sort order and the TSV text are produced by
    ``pav_cigar_write_tables``.  Same files as :func:`call_cigar` (gunzipped text byte-identical).  Returns
    ``(n_snv_rows, n_insdel_rows)``."""
    import numpy as np
    from . import _lib
    batch = int(batch)
    table = _lib.BedTable(bed, with_cigar=True)                                     # :805
    trim_table = _lib.BedTable(bed_trim, with_cigar=False)                          # :813-816 (POS, END, INDEX)
    own = ctx is None
    if own:
        ctx = _lib.Context(device_id)
    try:
        cols = table.fetch()
        sel = cols['CALL_BATCH'] == batch if table.n_rows else np.zeros(0, dtype=bool)    # :807
        want_ref = [table.chrom_names[i] for i in np.unique(cols['#CHROM'][sel])] if table.n_rows else []
        want_tig = [table.qry_names[i] for i in np.unique(cols['QRY_ID'][sel])] if table.n_rows else []
        if sel.any():
            cigarcall.load_sequences(ctx, ref_fa_name, tig_fa_name, names=(want_ref, want_tig))
        else:
            ctx.seq_load(_lib.PAV_ROLE_REF, [], [])
            ctx.seq_load(_lib.PAV_ROLE_TIG, [], [])
        index = ctx.cigar_load_bed(table, batch)
        try:
            ctx.cigar_call()
        except _lib.CigarDeviceError as ex:
            if ex.detail is None:
                raise
            rows = np.flatnonzero(sel)                                               # the reference's message names the row
            df_err = pd.DataFrame({'#CHROM': [table.chrom_names[i] for i in cols['#CHROM'][rows]], 'POS': cols['POS'][rows],
                                   'QRY_ID': [table.qry_names[i] for i in cols['QRY_ID'][rows]], 'INDEX': cols['INDEX'][rows]})
            cigarcall._raise_reference_error(ex.detail, df_err)
        tc = trim_table.fetch()
        trim = pd.DataFrame({'POS': tc['POS'], 'END': tc['END']}, index=tc['INDEX']).astype(int)
        trim = trim.reindex(list(index), fill_value=-1)                             # absent INDEX => -1 => TRIM (:818-822)
        return ctx.cigar_write_tables(hap, index, trim['POS'].to_numpy(dtype='int64'), trim['END'].to_numpy(dtype='int64'),
                                      snv_path=bed_snv, insdel_path=bed_insdel, threads=threads)
    finally:
        table.close()
        trim_table.close()
        if own:
            ctx.close()


def call_cigar(bed, bed_trim, tig_fa_name, ref_fa_name, hap, batch, bed_insdel=None, bed_snv=None, ctx=None,
               device_id=0):
    """Body of rule call_cigar.  Writes the two batch tables when output names are given; returns the frames."""
    batch = int(batch)
    df_align = read_align_bed(bed)
    df_align = df_align.loc[df_align['CALL_BATCH'] == batch]                      # :807
    df_snv, df_insdel = cigarcall.make_insdel_snv_calls(
        df_align, ref_fa_name, tig_fa_name, hap, version_id=False, ctx=ctx, device_id=device_id)   # :810
    df_trim = read_trim_bed(bed_trim)
    df_snv = apply_trim_filter(df_snv, df_trim)
    df_insdel = apply_trim_filter(df_insdel, df_trim)
    if bed_insdel is not None:
        df_insdel.to_csv(bed_insdel, sep='\t', index=False, compression='gzip')    # :845
    if bed_snv is not None:
        df_snv.to_csv(bed_snv, sep='\t', index=False, compression='gzip')          # :846
    return df_snv, df_insdel


def call_cigar_merged_files(bed, bed_trim, tig_fa_name, ref_fa_name, hap, bed_insdel, bed_snv, ctx=None, device_id=0, threads=0):
    """Rules call_cigar (all CALL_BATCH values) + call_cigar_merge in one pass: every alignment row of the haplotype is called at
    once and ``pav_cigar_write_tables`` writes the *merged* tables (rules/call.snakefile:755-786) - the row order of the batch
    files concatenated in batch order and stable-sorted, which the writer reproduces from CALL_BATCH.  Same text as
    :func:`call_cigar_files` x 10 -> :func:`call_cigar_merge`.  Returns ``(n_snv_rows, n_insdel_rows)``."""
    import numpy as np
    from . import _lib
    table = _lib.BedTable(bed, with_cigar=True)
    trim_table = _lib.BedTable(bed_trim, with_cigar=False)
    own = ctx is None
    if own:
        ctx = _lib.Context(device_id)
    try:
        cols = table.fetch()
        if table.n_rows:
            cigarcall.load_sequences(ctx, ref_fa_name, tig_fa_name, names=(table.chrom_names, table.qry_names))
        else:
            ctx.seq_load(_lib.PAV_ROLE_REF, [], [])
            ctx.seq_load(_lib.PAV_ROLE_TIG, [], [])
        index = ctx.cigar_load_bed(table, -1)                                        # every row, table order
        try:
            ctx.cigar_call()
        except _lib.CigarDeviceError as ex:
            if ex.detail is None:
                raise
            df_err = pd.DataFrame({'#CHROM': [table.chrom_names[i] for i in cols['#CHROM']], 'POS': cols['POS'],
                                   'QRY_ID': [table.qry_names[i] for i in cols['QRY_ID']], 'INDEX': cols['INDEX']})
            cigarcall._raise_reference_error(ex.detail, df_err)
        tc = trim_table.fetch()
        trim = pd.DataFrame({'POS': tc['POS'], 'END': tc['END']}, index=tc['INDEX']).astype(int).reindex(list(index), fill_value=-1)
        batch = cols['CALL_BATCH'] if table.n_rows else np.zeros(0, dtype=np.int64)
        return ctx.cigar_write_tables(hap, index, trim['POS'].to_numpy(dtype='int64'), trim['END'].to_numpy(dtype='int64'),
                                      snv_path=bed_snv, insdel_path=bed_insdel, threads=threads, call_batch=batch)
    finally:
        table.close()
        trim_table.close()
        if own:
            ctx.close()


def call_cigar_merge(bed_insdel_list, bed_snv_list, out_insdel=None, out_snv=None):
    """Body of rule call_cigar_merge (rules/call.snakefile:763-786)."""
    df_insdel = pd.concat(
        [pd.read_csv(f, sep='\t', keep_default_na=False) for f in bed_insdel_list], axis=0
    ).reset_index(drop=True).sort_values(['#CHROM', 'POS', 'END', 'ID'])
    df_snv = pd.concat(
        [pd.read_csv(f, sep='\t', keep_default_na=False) for f in bed_snv_list], axis=0
    ).reset_index(drop=True).sort_values(['#CHROM', 'POS'])
    if out_insdel is not None:
        df_insdel.to_csv(out_insdel, sep='\t', index=False, compression='gzip')
    if out_snv is not None:
        df_snv.to_csv(out_snv, sep='\t', index=False, compression='gzip')
    return df_snv, df_insdel


# ---------------------------------------------------------------------------------------------------------
# rule call_inv_batch (rules/call_inv.snakefile:115-311) and call_inv_batch_merge (:94-112)
# ---------------------------------------------------------------------------------------------------------

INV_BED_COLUMNS = ['#CHROM', 'POS', 'END', 'ID', 'SVTYPE', 'SVLEN', 'HAP', 'QRY_REGION', 'QRY_STRAND', 'CI',
                   'RGN_REF_INNER', 'RGN_QRY_INNER', 'RGN_REF_DISC', 'RGN_QRY_DISC', 'FLAG_ID', 'FLAG_TYPE', 'ALIGN_INDEX',
                   'CALL_SOURCE', 'FILTER', 'SEQ']                                   # call_inv.snakefile:270-282


def collapse_to_set(items, to_type=None):
    """Flatten nested tuples / lists into a set (pavlib/util.py:105-122)."""
    stack, out = list(items), set()
    while stack:
        v = stack.pop()
        if isinstance(v, (tuple, list)):
            stack.extend(v)
        else:
            out.add(to_type(v) if to_type is not None else v)
    return out


def inv_bed_row(inv_call, hap, flag_type, tig_fa, ctx=None, seq=None, as_list=False):
    """One INV BED record as rule call_inv_batch builds it (call_inv.snakefile:203-282).  ``ctx``: a context on which the contig
    file is resident - the SEQ column then comes from HBM instead of a host-side parse of the file; ``seq``: the column itself,
    fetched by the caller (:func:`inv_seq_columns`: all calls of a haplotype in one round trip)."""
    from . import _lib, inv as pavinv, seq as pavseq
    if seq is not None:
        pass
    elif ctx is not None:
        seq = pavseq.region_seq_resident(ctx, _lib.PAV_ROLE_TIG, inv_call.region_tig_outer, rev_compl=inv_call.region_tig_outer.is_rev)
    else:
        seq = pavseq.region_seq_fasta(inv_call.region_tig_outer, tig_fa, rev_compl=inv_call.region_tig_outer.is_rev)
    align_index = ','.join(sorted(collapse_to_set(
        (inv_call.region_ref_outer.pos_aln_index, inv_call.region_ref_outer.end_aln_index,
         inv_call.region_ref_inner.pos_aln_index, inv_call.region_ref_inner.end_aln_index), to_type=str)))
    values = [inv_call.region_ref_outer.chrom, inv_call.region_ref_outer.pos, inv_call.region_ref_outer.end,
              inv_call.id, 'INV', inv_call.svlen, hap,
              inv_call.region_tig_outer.to_base1_string(), '-' if inv_call.region_tig_outer.is_rev else '+', 0,
              inv_call.region_ref_inner.to_base1_string(), inv_call.region_tig_inner.to_base1_string(),
              inv_call.region_ref_discovery.to_base1_string(), inv_call.region_tig_discovery.to_base1_string(),
              inv_call.region_flag.region_id(), flag_type, align_index, pavinv.CALL_SOURCE, 'PASS', seq]
    return values if as_list else pd.Series(values, index=INV_BED_COLUMNS)


def inv_seq_columns(ctx, calls):
    """SEQ column of every InvCall of ``calls`` from the contig records resident on ``ctx`` (one device round trip):
    ``region_seq_fasta(call.region_tig_outer, tig_fa, rev_compl=is_rev)`` of rules/call_inv.snakefile:212 for each."""
    from . import _lib, seq as pavseq
    names = {n: i for i, n in enumerate(ctx.seq_names(_lib.PAV_ROLE_TIG))}
    regions = [c.region_tig_outer for c in calls]
    parts = ctx.seq_fetch_many(_lib.PAV_ROLE_TIG, [(names[r.chrom], r.pos, r.end) for r in regions])
    return [(pavseq._COMP[b[::-1]] if r.is_rev else b).tobytes().decode() for r, b in zip(regions, parts)]


# ---- the per-batch INV tables and their merge as text ------------------------------------------------------------------
# rule call_inv_batch ends with pd.concat(call_list, axis=1).T.sort_values([...]).to_csv(...) (call_inv.snakefile:297-311), rule
# call_inv_batch_merge reads the sixty files back, concatenates, drops duplicate IDs, sorts and writes (:101-112).  On sixty frames of
# one to three rows that is 0.25 s of pandas per haplotype - a third of the whole files-to-files time once the device writes the big
# tables.  The functions below produce the SAME bytes from the row values: the csv module pandas itself writes with (QUOTE_MINIMAL,
# tab-delimited), Python's stable sort on the same key columns, the column union in order of first appearance.  Values that
# pandas.read_csv would re-type on the way through the merge (a text field that reads as a number or as NA) make _merge_safe refuse,
# and the caller then runs the rules' own pandas code; tests/test_gpu_inv.py compares the two paths byte for byte.
_NA_TEXT = {'', '#N/A', '#N/A N/A', '#NA', '-1.#IND', '-1.#QNAN', '-NaN', '-nan', '1.#IND', '1.#QNAN', '<NA>', 'N/A', 'NA', 'NULL', 'NaN', 'None',
            'n/a', 'nan', 'null'}


def _tsv_text(columns, rows):
    import csv
    import io
    buf = io.StringIO()
    w = csv.writer(buf, delimiter='\t', lineterminator='\n', quoting=csv.QUOTE_MINIMAL)
    w.writerow(columns)
    w.writerows(rows)
    return buf.getvalue()


def _merge_safe(rows):
    """True when every text field of ``rows`` comes back from pandas.read_csv as the same text (so that the merge rule's round trip
    through DataFrames changes nothing)."""
    import re
    plain_int = re.compile(r'^-?(0|[1-9]\d{0,17})$')                      # reads as int64 and is written back with the same digits

    def reads_as_number(v):
        try:
            float(v)
            return True
        except ValueError:
            return False
    for r in rows:
        if not isinstance(r[0], str) or plain_int.match(r[0]):              # a numeric #CHROM column would be SORTED as numbers
            return False
        for v in r:
            if not isinstance(v, str):
                continue
            if v in _NA_TEXT or v != v.strip() or '\r' in v or '\n' in v or '"' in v or '\t' in v \
                    or v in ('True', 'False', 'TRUE', 'FALSE', 'true', 'false') or (reads_as_number(v) and not plain_int.match(v)):
                return False
    return True


def inv_batch_texts(batch_rows):
    """``batch_rows[b]``: None (the batch has no flagged region: full header, call_inv.snakefile:148-167), or the list of its calls'
    INV BED rows as lists in INV_BED_COLUMNS order (empty: regions but no call -> the header without FILTER, :300-308 sic).
    -> (text of every batch table, text of the merged table, number of merged rows), or None when :func:`_merge_safe` refuses."""
    cols = list(INV_BED_COLUMNS)
    key = [cols.index(c) for c in ('#CHROM', 'POS', 'END', 'ID')]
    no_filter = [c for c in cols if c != 'FILTER']
    texts, merged_cols, merged_rows, seen = [], [], [], set()
    for rows in batch_rows:
        if rows is None:
            texts.append(_tsv_text(cols, []))
            header = cols
        elif not rows:
            texts.append(_tsv_text(no_filter, []))
            header = no_filter
        else:
            if not _merge_safe(rows):
                return None
            rows = sorted(rows, key=lambda r: tuple(r[k] for k in key))                 # sort_values: stable, by the four columns
            texts.append(_tsv_text(cols, rows))
            header = cols
            for r in rows:                                                             # drop_duplicates('ID') keeps the first in concat order
                if r[3] not in seen:
                    seen.add(r[3])
                    merged_rows.append(r)
        for c in header:                                                               # pd.concat: columns in order of first appearance
            if c not in merged_cols:
                merged_cols.append(c)
    merged_rows.sort(key=lambda r: tuple(r[k] for k in key))
    if merged_cols != cols:
        pick = [cols.index(c) for c in merged_cols]
        merged_rows = [[r[k] for k in pick] for r in merged_rows]
    return texts, _tsv_text(merged_cols, merged_rows), len(merged_rows)


def call_inv_batch(bed_flag, bed_aln, tig_fa, fai, ref_fa, hap, batch, bed_out=None, log_path=None,
                   density_out_dir=None, k_size=31, inv_region_limit=None, inv_min_expand=None, srs_list=None, ctx=None,
                   device_id=0):
    """Body of rule call_inv_batch: scan every flagged region of ``batch``, write the INV BED, the per-call density
    tables and the log.  All regions of the batch are scanned in lock-step on the GPU; logs are emitted in region
    order, so the files equal those of the sequential reference loop."""
    import gc
    import io
    import os
    from . import _lib, inv as pavinv, seq as pavseq
    from .align import AlignLift
    from .fasta import read_fai
    from .kmer import KmerUtil

    batch = int(batch)
    if density_out_dir is not None:
        os.makedirs(density_out_dir, exist_ok=True)
    srs_tree = pavinv.get_srs_tree(srs_list)
    df_flag = pd.read_csv(bed_flag, sep='\t', header=0)
    df_flag = df_flag.loc[df_flag['BATCH'] == batch]
    empty_cols = [c for c in INV_BED_COLUMNS]
    if df_flag.shape[0] == 0:
        df_bed = pd.DataFrame([], columns=empty_cols)                                # :148-167
    else:
        k_util = KmerUtil(k_size)
        own = ctx is None
        if own:
            ctx = _lib.Context(device_id)
        try:
            align_lift = AlignLift(pd.read_csv(bed_aln, sep='\t'), read_fai(fai), ctx=ctx)   # lift tables built on the GPU
            regions = [pavseq.Region(row['#CHROM'], row['POS'], row['END']) for _, row in df_flag.iterrows()]
            logs = [io.StringIO() for _ in regions]
            results = pavinv.scan_for_inv_batch(regions, ref_fa, tig_fa, align_lift, k_util, max_region_size=inv_region_limit,
                                                logs=logs, srs_tree=srs_tree, min_exp_count=inv_min_expand, ctx=ctx,
                                                eager_tables=False)   # tables are consumed below, before any other scan
            id_set = set()
            call_list = []
            native_tables = []                                                        # (region number, path): one library call
            log_file = open(log_path, 'w') if log_path is not None else None
            try:
                for (_, row), res, lg in zip(df_flag.iterrows(), results, logs):
                    if log_file is not None:
                        log_file.write(lg.getvalue())
                    if isinstance(res, RuntimeError):                                 # :198-200
                        if log_file is not None:
                            log_file.write('RuntimeError in scan_for_inv(): {}\n'.format(res))
                        res = None
                    if res is not None and res.id not in id_set:                      # :203
                        call_list.append(inv_bed_row(res, hap, row['TYPE'], tig_fa))
                        id_set.add(res.id)
                        if density_out_dir is not None:                               # :287-291
                            path = os.path.join(density_out_dir, 'density_{}_{}.tsv.gz'.format(res.id, hap))
                            nt = res.native_table
                            if nt is not None and nt[0] is ctx and nt[2] == ctx._inv_generation and callable(res._df):
                                native_tables.append((nt[1], path))                   # text from the library's host copy
                            else:
                                res.df.to_csv(path, sep='\t', index=False, compression='gzip')
                if native_tables:
                    ctx.inv_write_tables([r for r, _ in native_tables], [p for _, p in native_tables])
            finally:
                if log_file is not None:
                    log_file.close()
        finally:
            if own:
                ctx.close()
        if len(call_list) > 0:
            df_bed = pd.concat(call_list, axis=1).T.sort_values(['#CHROM', 'POS', 'END', 'ID'])   # :297
        else:
            df_bed = pd.DataFrame([], columns=[c for c in INV_BED_COLUMNS if c != 'FILTER'])      # :300-308 (sic)
    if bed_out is not None:
        df_bed.to_csv(bed_out, sep='\t', index=False, compression='gzip')              # :311
    return df_bed


def call_inv_batch_merge(bed_list, bed_out=None, gzip_level=None):
    """Body of rule call_inv_batch_merge (call_inv.snakefile:101-112).  ``gzip_level``: deflate level of the output (default:
    pandas' 9, as the reference writes it; the text inside is the same at every level)."""
    df = pd.concat([pd.read_csv(f, sep='\t') for f in bed_list], axis=0)
    df.drop_duplicates('ID', inplace=True)
    df = df.sort_values(['#CHROM', 'POS', 'END', 'ID'])
    if bed_out is not None:
        df.to_csv(bed_out, sep='\t', index=False,
                  compression='gzip' if gzip_level is None else {'method': 'gzip', 'compresslevel': int(gzip_level)})
    return df


# ---------------------------------------------------------------------------------------------------------
# rules call_inv_cluster (:603-692), call_inv_flag_insdel_cluster (:480-599), call_inv_merge_flagged_loci (:321-474)
# ---------------------------------------------------------------------------------------------------------

# ---------------------------------------------------------------------------------------------------------
# rule align_get_read_bed (rules/align.snakefile:101-171)
# ---------------------------------------------------------------------------------------------------------

CALL_CIGAR_BATCH_COUNT = 10                                             # pavlib/cigarcall.py:21


def align_get_read_bed(sam, tig_fai, hap, bed_out=None, align_head_out=None):
    """Body of rule align_get_read_bed: SAM -> trim-none alignment table (+ the SAM header lines).  No GPU involved."""
    import gzip
    import os
    from .align import get_align_bed
    from .fasta import read_fai
    if os.stat(sam).st_size == 0:                                       # :111-132 (sic: QRY_MAPPED, no QRY_LEN)
        df = pd.DataFrame([], columns=['#CHROM', 'POS', 'END', 'INDEX', 'QRY_ID', 'QRY_POS', 'QRY_END', 'QRY_MAPPED', 'RG', 'AO',
                                       'MAPQ', 'REV', 'FLAGS', 'HAP', 'CIGAR'])
        if bed_out is not None:
            df.to_csv(bed_out, sep='\t', index=False, compression='gzip')
        if align_head_out is not None:
            with open(align_head_out, 'w'):
                pass
        return df
    df_tig_fai = read_fai(tig_fai).copy()
    df_tig_fai.index = df_tig_fai.index.astype(str)                     # :135-136
    df = get_align_bed(sam, df_tig_fai, hap)                            # :139
    if align_head_out is not None:                                      # :142-161
        with gzip.open(align_head_out, 'wb') as out_file:
            out_file.write(df.attrs['sam_header'])
    df['CALL_BATCH'] = df['INDEX'].apply(lambda val: val % CALL_CIGAR_BATCH_COUNT)   # :164
    df['TRIM_REF_L'] = 0                                                # :167-170
    df['TRIM_REF_R'] = 0
    df['TRIM_QRY_L'] = 0
    df['TRIM_QRY_R'] = 0
    if bed_out is not None:
        df.to_csv(bed_out, sep='\t', index=False, compression='gzip')   # :173
    return df


def _with_ctx(ctx, device_id):
    from . import _lib
    return (ctx, False) if ctx is not None else (_lib.Context(device_id), True)


def _write_bed(df, path):
    if path is not None:
        df.to_csv(path, sep='\t', index=False, compression='gzip')
    return df


def call_inv_cluster(bed_list, vartype, bed_out=None, ctx=None, device_id=0, cluster_win=200, cluster_min_snv=20,
                     cluster_min_indel=10):
    """Body of rule call_inv_cluster: ``bed_list`` is what _input_call_inv_cluster returns (call_inv.snakefile:32-54)."""
    from . import flag
    if vartype not in ('indel', 'snv'):
        raise RuntimeError('Bad variant type {}: Expected "indel" or "snv"')           # :626
    df = pd.concat([pd.read_csv(f, sep='\t', usecols=('#CHROM', 'POS', 'END', 'SVTYPE', 'SVLEN', 'FILTER'), low_memory=False,
                                dtype={'#CHROM': str}) for f in bed_list], axis=0)     # :629-637
    ctx, own = _with_ctx(ctx, device_id)
    try:
        return _write_bed(flag.cluster_table(ctx, df, vartype, cluster_win, cluster_min_snv, cluster_min_indel), bed_out)
    finally:
        if own:
            ctx.close()


def call_inv_flag_insdel_cluster(bed, vartype, bed_out=None, ctx=None, device_id=0, flank_cluster=2, flank_merge=2000,
                                 cluster_min_svlen=4):
    """Body of rule call_inv_flag_insdel_cluster."""
    from . import flag
    df = pd.read_csv(bed, sep='\t', header=0, low_memory=False)                       # :500
    ctx, own = _with_ctx(ctx, device_id)
    try:
        return _write_bed(flag.insdel_table(ctx, df, vartype, flank_cluster, flank_merge, cluster_min_svlen), bed_out)
    finally:
        if own:
            ctx.close()


def call_inv_merge_flagged_loci(bed_insdel_sv, bed_insdel_indel, bed_cluster_indel, bed_cluster_snv, bed_out=None, ctx=None,
                                device_id=0, flank=500, batch_count=60, inv_sig_filter='svindel'):
    """Body of rule call_inv_merge_flagged_loci (config keys inv_sig_merge_flank / inv_sig_batch_count / inv_sig_filter)."""
    from . import flag
    frames = [pd.read_csv(f, sep='\t') for f in (bed_insdel_sv, bed_insdel_indel, bed_cluster_indel, bed_cluster_snv)]   # :358-361
    ctx, own = _with_ctx(ctx, device_id)
    try:
        return _write_bed(flag.merge_flagged(ctx, *frames, flank=flank, batch_count=batch_count, inv_sig_filter=inv_sig_filter),
                          bed_out)
    finally:
        if own:
            ctx.close()


FLAG_OUTPUTS = ('insdel_sv', 'insdel_indel', 'cluster_indel', 'cluster_snv', 'flagged_regions')


def call_inv_flag(bed, bed_trim, tig_fa_name, ref_fa_name, out=None, ctx=None, device_id=0, inv_sig_filter='svindel', **config):
    """The three flag rules for one haplotype without the intermediate variant tables: CIGAR calls of *all* alignment rows
    (every CALL_BATCH) are made on the device and flagged in place (``pav_cigar_flag``).  ``out`` maps FLAG_OUTPUTS names
    to file names (any subset).  The tables equal those of the rule chain call_cigar x10 -> call_cigar_merge ->
    call_inv_cluster x2 / call_inv_flag_insdel_cluster x2 -> call_inv_merge_flagged_loci."""
    from . import _lib, flag
    df_align = read_align_bed(bed)
    df_trim = read_trim_bed(bed_trim)
    ctx, own = _with_ctx(ctx, device_id)
    try:
        ref_names, tig_names = cigarcall.load_sequences(ctx, ref_fa_name, tig_fa_name, df_align)
        aln, text, off = cigarcall.pack_alignments(df_align, ref_names, tig_names)
        ctx.cigar_load(aln, text, off)
        try:
            ctx.cigar_call()
        except _lib.CigarDeviceError as ex:
            if ex.detail is None:
                raise
            cigarcall._raise_reference_error(ex.detail, df_align)
        trim = df_trim.reindex(list(df_align['INDEX'].to_numpy(dtype='int64')), fill_value=-1)
        res = flag.flag_from_calls(ctx, trim['POS'].to_numpy(dtype='int64'), trim['END'].to_numpy(dtype='int64'),
                                   inv_sig_filter=inv_sig_filter, **config)
    finally:
        if own:
            ctx.close()
    for name, path in (out or {}).items():
        if name not in FLAG_OUTPUTS:
            raise KeyError(name)
        _write_bed(res[name], path)
    return res


# ---------------------------------------------------------------------------------------------------------
# One haplotype, files to files: call_cigar x 10 -> call_cigar_merge -> the five flag rules -> call_inv_batch x N -> merge
# ---------------------------------------------------------------------------------------------------------

def haplotype_paths(out_dir, asm_name, hap, batch_count=60):
    """The files the rule chain of one haplotype reads and writes, under ``out_dir`` with the reference's relative names
    (rules/call.snakefile:755-803, rules/call_inv.snakefile:94-131, 326-331, 485-487, 605-607)."""
    import os
    j = lambda *p: os.path.join(out_dir, *p)                                          # noqa: E731
    return {
        'cigar_batch_insdel': [j('temp', asm_name, 'cigar', 'batched', f'insdel_{hap}_{b}.bed.gz') for b in range(CALL_CIGAR_BATCH_COUNT)],
        'cigar_batch_snv': [j('temp', asm_name, 'cigar', 'batched', f'snv.bed_{hap}_{b}.gz') for b in range(CALL_CIGAR_BATCH_COUNT)],
        'insdel': j('temp', asm_name, 'cigar', 'merged', f'svindel_insdel_{hap}.bed.gz'),
        'snv': j('temp', asm_name, 'cigar', 'merged', f'snv_snv_{hap}.bed.gz'),
        'insdel_sv': j('temp', asm_name, 'inv_caller', 'flag', f'insdel_sv_{hap}.bed.gz'),
        'insdel_indel': j('temp', asm_name, 'inv_caller', 'flag', f'insdel_indel_{hap}.bed.gz'),
        'cluster_indel': j('temp', asm_name, 'inv_caller', 'flag', f'cluster_indel_{hap}.bed.gz'),
        'cluster_snv': j('temp', asm_name, 'inv_caller', 'flag', f'cluster_snv_{hap}.bed.gz'),
        'flagged_regions': j('results', asm_name, 'inv_caller', f'flagged_regions_{hap}.bed.gz'),
        'inv_batch': [j('temp', asm_name, 'inv_caller', 'batch', hap, f'inv_call_{b}.bed.gz') for b in range(batch_count)],
        'inv_log': [j('log', asm_name, 'inv_caller', 'log', hap, f'inv_call_{b}.log') for b in range(batch_count)],
        'density_dir': j('results', asm_name, 'inv_caller', 'density_table'),
        'inv': j('temp', asm_name, 'inv_caller', f'sv_inv_{hap}.bed.gz'),
    }


def _makedirs_for(paths):
    import os
    for v in paths.values():
        for f in (v if isinstance(v, list) else [v]):
            os.makedirs(f if f.endswith('density_table') else os.path.dirname(f), exist_ok=True)


def call_haplotype(bed, bed_trim, tig_fa_name, ref_fa_name, asm_name, hap, out_dir, ctx=None, device_id=0, config=None, threads=0,
                   gzip_level=0, timings=None):
    """The whole call path of ONE haplotype on one GPU, from its files to its files - what the rules call_cigar (all ten
    CALL_BATCH jobs) -> call_cigar_merge -> call_inv_cluster x 2 / call_inv_flag_insdel_cluster x 2 -> call_inv_merge_flagged_loci
    -> call_inv_batch (all batches) -> call_inv_batch_merge produce, with the calls made once and kept resident between the
    stages: every alignment row is called at once and the merged tables come from the native writer, the signature flagging
    runs on the resident records (``pav_cigar_flag``), ONE scan covers the flagged regions of all batches; the per-batch INV
    tables, logs and density tables are then written batch by batch in the reference's order (duplicate calls are dropped per
    batch, then across batches by the merge rule's own code), so every file equals the rule chain's.

    ``config``: the per-assembly configuration keys of the rules (inv_sig_filter, inv_sig_batch_count, inv_sig_merge_flank,
    inv_sig_cluster_*, inv_sig_insdel_*, inv_k_size, inv_region_limit, inv_min_expand, srs_list).  A reference made resident
    by ``cigarcall.load_reference`` stays resident.  Returns a manifest dict (counts + file names)."""
    import io
    import os
    import time
    import numpy as np
    from . import _lib, flag, inv as pavinv, seq as pavseq
    from .align import AlignLift
    from .fasta import read_fai
    from .kmer import KmerUtil
    cfg = dict(config or {})
    batch_count = int(cfg.get('inv_sig_batch_count', 60))
    P = haplotype_paths(out_dir, asm_name, hap, batch_count)
    _makedirs_for(P)
    t_last = [time.perf_counter()]

    def lap(name):
        if timings is not None:
            now = time.perf_counter()
            timings[name] = timings.get(name, 0.0) + now - t_last[0]
            t_last[0] = now
    ctx, own = _with_ctx(ctx, device_id)
    table = trim_table = th = None
    writing = False
    try:
        # the contig file is parsed on a second thread while the tables are parsed and - the first time - the reference is parsed,
        # uploaded and packed (the native readers and the upload release the GIL)
        import threading
        from . import fasta as pavfasta
        dev_fa = cigarcall.device_fasta()
        if not dev_fa:                                                                # (host parser: the contig file beside the reference)
            th = threading.Thread(target=pavfasta.open_fasta, args=(tig_fa_name,))
            th.start()
        # ... and so are the two alignment tables (one gzip stream each: 0.12 s of inflate apiece, serial by nature), each on a
        # thread of its own beside the FASTA parsers
        opened, open_err = {}, []

        def open_table(key, path):
            try:
                opened[key] = _lib.BedTable(path, with_cigar=True)
            except BaseException as ex:                                                   # noqa: BLE001 - raised again on this thread
                open_err.append(ex)
        tab_threads = [threading.Thread(target=open_table, args=(k, f)) for k, f in (('bed', bed), ('trim', bed_trim))]
        for t in tab_threads:
            t.start()
        tig_err, th_tig = [], None
        if dev_fa:
            # device loader: the contig file goes up on a thread of its own beside the reference (the two stores of a context may
            # be loaded side by side, include/pav_amd.h); nothing of either file is parsed on the host
            def load_tig():
                try:
                    ctx.seq_load_fasta_path(_lib.PAV_ROLE_TIG, tig_fa_name)
                except BaseException as ex:                                               # noqa: BLE001 - raised again on this thread
                    tig_err.append(ex)
            th_tig = threading.Thread(target=load_tig)
            th_tig.start()
        try:
            cigarcall.load_reference(ctx, ref_fa_name)
        finally:
            for t in tab_threads:
                t.join()
            if th_tig is not None:
                th_tig.join()
            table, trim_table = opened.get('bed'), opened.get('trim')
        if open_err or tig_err:
            raise (open_err or tig_err)[0]
        cols = table.fetch()
        if th is not None:
            th.join()
        if not dev_fa:
            cigarcall.load_sequences(ctx, ref_fa_name, tig_fa_name)                    # every contig record: the scan may lift anywhere
        ctx._inv_loaded = (str(ref_fa_name), str(tig_fa_name))
        lap('sequences')
        index = ctx.cigar_load_bed(table, -1)
        try:
            counts = ctx.cigar_call()
        except _lib.CigarDeviceError as ex:
            if ex.detail is None:
                raise
            cigarcall._raise_reference_error(ex.detail, pd.DataFrame({
                '#CHROM': [table.chrom_names[i] for i in cols['#CHROM']], 'POS': cols['POS'],
                'QRY_ID': [table.qry_names[i] for i in cols['QRY_ID']], 'INDEX': cols['INDEX']}))
        tc = trim_table.fetch()
        trim = pd.DataFrame({'POS': tc['POS'], 'END': tc['END']}, index=tc['INDEX']).astype(int).reindex(list(index), fill_value=-1)
        tp, te = trim['POS'].to_numpy(dtype='int64'), trim['END'].to_numpy(dtype='int64')
        lap('tables in + CIGAR-call')
        call_batch = cols['CALL_BATCH'] if table.n_rows else np.zeros(0, dtype=np.int64)
        # The two merged tables are written on a thread of the library's own beside the stages below - nothing on this path reads them
        # (rule call_cigar_merge is a leaf for the inversion rules).  The HOST writer starts now (it leaves the GPU alone after its
        # sort); the DEVICE writer - 0.13 s of text and gzip kernels that fill the GPU - starts behind the flagging and the scan,
        # which are short and which this thread waits for, and runs beside the host-side stages that follow them.
        def start_table_writer():
            ctx.cigar_write_tables(hap, index, tp, te, snv_path=P['snv'], insdel_path=P['insdel'], threads=threads,
                                   gzip_level=gzip_level, call_batch=call_batch, background=True)
        writer_late = os.environ.get('PAV_WRITER', 'device') != 'host'
        if not writer_late:
            start_table_writer()
            writing = True
        lap('merged SNV / INS-DEL tables')
        flag_cfg = {}
        for key, name in (('inv_sig_cluster_win', 'cluster_win'), ('inv_sig_cluster_snv_min', 'cluster_min_snv'),
                          ('inv_sig_cluster_indel_min', 'cluster_min_indel'), ('inv_sig_insdel_cluster_flank', 'insdel_flank_cluster'),
                          ('inv_sig_insdel_merge_flank', 'insdel_flank_merge'), ('inv_sig_cluster_svlen_min', 'insdel_min_svlen'),
                          ('inv_sig_merge_flank', 'merge_flank'), ('inv_sig_batch_count', 'batch_count')):
            if key in cfg:
                flag_cfg[name] = int(cfg[key])
        res = flag.flag_from_calls(ctx, tp, te, inv_sig_filter=cfg.get('inv_sig_filter', 'svindel'), **flag_cfg)
        for name in FLAG_OUTPUTS:
            _write_bed(res[name], P[name])
        lap('flag tables')
        df_flag = pd.read_csv(P['flagged_regions'], sep='\t', header=0)                # as rule call_inv_batch reads it (:144)
        k_util = KmerUtil(int(cfg.get('inv_k_size', 31)))
        srs_tree = pavinv.get_srs_tree(cfg.get('srs_list'))
        sel = df_flag.loc[df_flag['BATCH'] >= 0] if df_flag.shape[0] else df_flag
        results, logs = [], []
        if sel.shape[0]:
            # the trimmed table as rule call_inv_batch reads it (:174-177), from the columns the native reader has parsed already
            off = tc['CIGAR_OFF'].astype(np.int64) if 'CIGAR_OFF' in tc else np.zeros(trim_table.n_rows + 1, dtype=np.int64)
            text = tc['CIGAR_TEXT'].tobytes() if 'CIGAR_TEXT' in tc else b''
            df_trim = pd.DataFrame({
                '#CHROM': [trim_table.chrom_names[i] for i in tc['#CHROM'].tolist()], 'POS': tc['POS'], 'END': tc['END'], 'INDEX': tc['INDEX'],
                'QRY_ID': [trim_table.qry_names[i] for i in tc['QRY_ID'].tolist()], 'QRY_POS': tc['QRY_POS'], 'QRY_END': tc['QRY_END'],
                'REV': tc['REV'], 'CIGAR': [text[a:b].decode() for a, b in zip(off[:-1].tolist(), off[1:].tolist())]})
            align_lift = AlignLift(df_trim, read_fai(str(tig_fa_name) + '.fai'), ctx=ctx)
            regions = [pavseq.Region(c, p, e) for c, p, e in zip(sel['#CHROM'], sel['POS'], sel['END'])]
            logs = [io.StringIO() for _ in regions]
            lap('lift-over index')
            results = pavinv.scan_for_inv_batch(regions, ref_fa_name, tig_fa_name, align_lift, k_util,
                                                max_region_size=cfg.get('inv_region_limit'), logs=logs, srs_tree=srs_tree,
                                                min_exp_count=cfg.get('inv_min_expand'), ctx=ctx, eager_tables=False,
                                                found_out=io.StringIO())
            lap('scan')
        if writer_late:
            start_table_writer()
            writing = True
            lap('merged SNV / INS-DEL tables')
        # ---- per batch, in the order the reference's jobs see the rows: INV table, log, density tables ----------------------
        where = {ix: q for q, ix in enumerate(sel.index)}
        # first which calls each batch keeps (:203) and where their density tables go: the native writer starts on those - a
        # thread of its own, the library's formatting and gzip threads behind it - while the batch tables, the logs and the merge
        # are made here
        native_tables, kept, n_calls = [], [], 0
        for b in range(batch_count):
            rows_b = df_flag.loc[df_flag['BATCH'] == b] if df_flag.shape[0] else df_flag
            id_set, keep_b = set(), []
            for ix, row in rows_b.iterrows():
                r = results[where[ix]]
                if r is None or isinstance(r, RuntimeError) or r.id in id_set:
                    continue
                id_set.add(r.id)
                path = '{}/density_{}_{}.tsv.gz'.format(P['density_dir'], r.id, hap)
                nt = r.native_table
                native = nt is not None and nt[0] is ctx and nt[2] == ctx._inv_generation and callable(r._df)
                if native:
                    native_tables.append((nt[1], path))
                keep_b.append((ix, None if native else path))
            kept.append((rows_b, dict(keep_b)))
        table_writer = table_error = None
        if native_tables:
            # A call found through two flagged regions of different batches names the same file twice; the reference's batch jobs
            # would each write it (whichever job runs last wins there).  Here every file is written once, by the call of the
            # last batch that holds it - one writer per file.
            last = {}
            for rgn, path in native_tables:
                last[path] = rgn
            table_error = []

            def write_density():
                try:
                    ctx.inv_write_tables(list(last.values()), list(last.keys()), threads=threads, gzip_level=gzip_level)
                except BaseException as ex:                                               # noqa: BLE001 - re-raised on the caller's thread
                    table_error.append(ex)
            table_writer = threading.Thread(target=write_density)
            table_writer.start()
        try:
            batch_texts = []
            batch_rows = []                                                       # per batch: None / the rows as lists (inv_batch_texts)
            seq_of = {}
            if dev_fa:                                                            # the SEQ columns of every kept call, one round trip
                kept_ix = [ix for _, keep_b in kept for ix in keep_b]
                for ix, sq in zip(kept_ix, inv_seq_columns(ctx, [results[where[ix]] for ix in kept_ix])):
                    seq_of[ix] = sq
            for b in range(batch_count):
                rows_b, keep_b = kept[b]
                call_list = []
                with open(P['inv_log'][b], 'w') as log_file:
                    for ix, row in rows_b.iterrows():
                        r = results[where[ix]]
                        log_file.write(logs[where[ix]].getvalue())
                        if isinstance(r, RuntimeError):                                       # :198-200
                            log_file.write('RuntimeError in scan_for_inv(): {}\n'.format(r))
                        if ix in keep_b:
                            call_list.append(inv_bed_row(r, hap, row['TYPE'], tig_fa_name, seq=seq_of.get(ix), as_list=True))
                            if keep_b[ix] is not None:
                                r.df.to_csv(keep_b[ix], sep='\t', index=False, compression='gzip')
                batch_rows.append(None if rows_b.shape[0] == 0 else call_list)
                n_calls += len(call_list)
            # the per-batch tables (temporary files, read back by the merge rule) and the merged table: their text from the row values
            # (inv_batch_texts: the bytes the rules' pandas code writes), gzip on the device - all sixty-one in two launch sets
            fast = None if os.environ.get('PAV_INV_TABLES') == 'pandas' else inv_batch_texts(batch_rows)
            if fast is not None:
                texts, merged_text, n_inv = fast
                batch_texts = [t.encode() for t in texts]
            else:                                                                 # the rules' own code (:297-311, :101-112)
                for rows in batch_rows:
                    if rows is None:
                        df_bed = pd.DataFrame([], columns=list(INV_BED_COLUMNS))                  # :148-167
                    elif rows:
                        df_bed = pd.concat([pd.Series(r, index=INV_BED_COLUMNS) for r in rows], axis=1).T.sort_values(['#CHROM', 'POS', 'END', 'ID'])   # :297
                    else:
                        df_bed = pd.DataFrame([], columns=[c for c in INV_BED_COLUMNS if c != 'FILTER'])       # :300-308 (sic)
                    batch_texts.append(df_bed.to_csv(None, sep='\t', index=False).encode())
            for path, gz in zip(P['inv_batch'], ctx.gzip_buffers(batch_texts, 1)):
                with open(path, 'wb') as fh:
                    fh.write(gz)
            lap('INV batch tables + logs')
            if fast is None:
                df_merged = call_inv_batch_merge([io.BytesIO(t) for t in batch_texts], None)
                merged_text, n_inv = df_merged.to_csv(None, sep='\t', index=False), int(df_merged.shape[0])
            with open(P['inv'], 'wb') as fh:                                              # rule call_inv_batch_merge's to_csv, gzip on the device
                fh.write(ctx.gzip_buffer(merged_text.encode(), gzip_level or 6))
            lap('INV merge')
        finally:
            if table_writer is not None:
                table_writer.join()
        if table_error:
            raise table_error[0]
        lap('density tables')
        writing = False
        n_snv, n_insdel = ctx.cigar_write_wait()
        lap('merged SNV / INS-DEL tables')
        return {'asm_name': asm_name, 'hap': hap, 'aligned_bp': int(counts.aligned_bases), 'snv_rows': int(n_snv),
                'insdel_rows': int(n_insdel), 'flagged_regions': int(df_flag.shape[0]), 'scanned_regions': int(sel.shape[0]),
                'inv_calls_in_batches': int(n_calls), 'inv_calls': int(n_inv),
                'files': {k: P[k] for k in ('snv', 'insdel', 'flagged_regions', 'inv')}}
    finally:
        if writing:                                                # an error on the way: the writer thread is waited for all the same
            try:
                ctx.cigar_write_wait()
            except Exception:                                      # noqa: BLE001 - the error that brought us here is the one to report
                pass
        for t in (table, trim_table):
            if t is not None:
                t.close()
        from . import fasta as pavfasta2
        if th is not None and th.is_alive():                       # an error before the join: the helper would memoise the file after forget()
            th.join()
        pavfasta2.forget(tig_fa_name)                              # (the reference stays memoised; a haplotype's contigs are read once)
        if own:
            ctx.close()


def run_cohort(jobs, n_gpus, out_dir, ref_fa, **kw):
    """Many haplotypes on the GPUs of one node, one process per GPU, haplotypes dealt longest-first: see :mod:`pav_amd.cohort`."""
    from . import cohort
    return cohort.run_cohort(jobs, n_gpus, out_dir, ref_fa, **kw)


# ---------------------------------------------------------------------------------------------------------
# rules align_trim_tig / align_trim_tigref (rules/align.snakefile:54-97)
# ---------------------------------------------------------------------------------------------------------

def align_trim(bed, tig_fai, mode, bed_out=None, min_trim_tig_len=1000, redundant_callset=False, ctx=None, device_id=0):
    """Body of rule align_trim_tig (``mode='tig'``, input = trim-none table) or align_trim_tigref (``mode='ref'``, input =
    trim-tig table; ``redundant_callset`` -> match_tig)."""
    from .align import trim_alignments
    df = trim_alignments(pd.read_csv(bed, sep='\t', dtype={'#CHROM': str}), int(min_trim_tig_len), tig_fai,
                         match_tig=bool(redundant_callset) if mode == 'ref' else False, mode=mode, ctx=ctx, device_id=device_id)
    if bed_out is not None:
        df.to_csv(bed_out, sep='\t', index=False, compression='gzip')
    return df


# ---------------------------------------------------------------------------------------------------------
# rules call_lg_split / call_lg_discover (rules/call_lg.snakefile:40-139)
# ---------------------------------------------------------------------------------------------------------

def call_lg_split(bed, tsv_out=None, batch_count=10):
    """Body of rule call_lg_split: (CHROM, TIG, BATCH) of every chromosome / contig pair with several alignment records."""
    import collections
    df = pd.read_csv(bed, sep='\t')
    tig_map_count = collections.Counter(df[['#CHROM', 'QRY_ID']].apply(tuple, axis=1)) if df.shape[0] else {}
    rows = [pd.Series([chrom, tig, index % batch_count], index=['CHROM', 'TIG', 'BATCH'])
            for index, (chrom, tig) in enumerate([k for k, count in tig_map_count.items() if count > 1])]
    df_group = pd.concat(rows, axis=1).T if rows else pd.DataFrame([], columns=['CHROM', 'TIG', 'BATCH'])
    if tsv_out is not None:
        df_group.to_csv(tsv_out, sep='\t', index=False, compression='gzip')
    return df_group


def call_lg_discover(bed, tsv_group, fa, fai, bed_n, ref_fa, hap, batch, bed_ins=None, bed_del=None, bed_inv=None, log_path=None,
                     density_out_dir=None, k_size=31, inv_region_limit=None, srs_list=None, threads=1, ctx=None, device_id=0):
    """Body of rule call_lg_discover: alignment-truncating INS / DEL / INV of one batch of (chromosome, contig) pairs."""
    import collections
    import os
    import sys
    from . import fasta, inv as pavinv, lgsv
    srs_tree = pavinv.get_srs_tree(srs_list)
    df = pd.read_csv(bed, sep='\t', dtype={'#CHROM': str, 'QRY_ID': str})
    df_tig_fai = fasta.read_fai(fai)
    df_group = pd.read_csv(tsv_group, sep='\t', dtype={'CHROM': str, 'TIG': str})
    df_group = df_group.loc[df_group['BATCH'] == int(batch)]
    group_set = set(df_group[['CHROM', 'TIG']].apply(tuple, axis=1)) if df_group.shape[0] else set()
    if df.shape[0] > 0:
        df = df.loc[df.apply(lambda row: (row['#CHROM'], row['QRY_ID']) in group_set, axis=1)]
    n_tree = collections.defaultdict(pavinv.IntervalSet)
    for _, row in pd.read_csv(bed_n, sep='\t', dtype={'#CHROM': str}).iterrows():
        n_tree[row['#CHROM']][row['POS']:row['END']] = True
    if density_out_dir is not None:
        os.makedirs(density_out_dir, exist_ok=True)
    log_file = open(log_path, 'wt') if log_path is not None else sys.stdout
    try:
        df_ins, df_del, df_inv = lgsv.scan_for_events(df, df_tig_fai, hap, ref_fa, fa, k_size=k_size, n_tree=n_tree, srs_tree=srs_tree,
                                                      threads=threads, log=log_file, density_out_dir=density_out_dir,
                                                      max_region_size=inv_region_limit, version_id=False, ctx=ctx, device_id=device_id)
    finally:
        if log_path is not None:
            log_file.close()
    for frame, path in ((df_ins, bed_ins), (df_del, bed_del), (df_inv, bed_inv)):
        if path is not None:
            frame.to_csv(path, sep='\t', index=False, compression='gzip')
    return df_ins, df_del, df_inv
