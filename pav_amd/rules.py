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
