import pandas as pd


def get_df_fai(fai_file_name, usecols=('CHROM', 'LEN'), index_col='CHROM', squeeze=True):
    """FAI table -> Series name -> length (used at pavlib/inv.py:201, rules/call_inv.snakefile:176)."""
    df = pd.read_csv(fai_file_name, sep='\t', header=None,
                     names=['CHROM', 'LEN', 'POS', 'LINE_BP', 'LINE_BYTES'],
                     dtype={'CHROM': str, 'LEN': int})
    return df.set_index('CHROM')['LEN']
