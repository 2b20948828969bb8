#!/usr/bin/env python3
"""Trimming throughput probe (GPU box): the bench haplotype's alignment records cut into overlapping pieces, then
trim_alignments(mode='tig') + (mode='ref') like rules align_trim_tig / align_trim_tigref.  Prints one JSON object."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scale', type=float, default=1.0)
    ap.add_argument('--seed', type=int, default=1002)
    args = ap.parse_args()
    import __graft_entry__ as g
    g.build_cpu_side()
    from pav_amd import _lib, synth
    from pav_amd.align import trim as ptrim

    hap = synth.config2(seed=args.seed, scale=args.scale, threads=8)
    t0 = time.perf_counter()
    df0 = synth.split_overlaps(hap.df_align, args.seed)
    t_split = time.perf_counter() - t0
    ctx = _lib.Context(0)
    laps = {}
    real_load, real_pass, real_fetch = ctx.trim_load, ctx.trim_pass, ctx.trim_fetch

    def timed(name, fn):
        def wrap(*a, **k):
            t = time.perf_counter()
            r = fn(*a, **k)
            laps[name] = laps.get(name, 0.0) + time.perf_counter() - t
            return r
        return wrap
    ctx.trim_load, ctx.trim_pass, ctx.trim_fetch = timed('load+tokenise', real_load), timed('pair loops', real_pass), timed('fetch rows+cigar', real_fetch)
    out = {}
    for rep in range(2):
        laps.clear()
        t0 = time.perf_counter()
        df_tig = ptrim.trim_alignments(df0, 1000, hap.tig_lengths, mode='tig', ctx=ctx)
        t_tig = time.perf_counter() - t0
        t0 = time.perf_counter()
        df_ref = ptrim.trim_alignments(df_tig, 1000, hap.tig_lengths, mode='ref', ctx=ctx)
        t_ref = time.perf_counter() - t0
        out = {'rows_in': int(df0.shape[0]), 'rows_tig': int(df_tig.shape[0]), 'rows_tigref': int(df_ref.shape[0]),
               'cigar_bytes_in': int(df0['CIGAR'].str.len().sum()),
               'records_trimmed': int((df_ref[['TRIM_QRY_L', 'TRIM_QRY_R']].to_numpy().sum(axis=1) > 0).sum()),
               'bases_trimmed_qry': int(df_ref[['TRIM_QRY_L', 'TRIM_QRY_R']].to_numpy().sum()),
               'wall_s': {'align_trim_tig': round(t_tig, 3), 'align_trim_tigref': round(t_ref, 3)},
               'library_s': {k: round(v, 4) for k, v in laps.items()}, 'split_table_s': round(t_split, 2)}
    ok = True
    for key, a, b in (('QRY_ID', 'QRY_POS', 'QRY_END'), ('#CHROM', 'POS', 'END')):
        for _, grp in df_ref.groupby(key):
            grp = grp.sort_values(a)
            ok = ok and bool((grp[a].to_numpy()[1:] >= grp[b].to_numpy()[:-1]).all())
    out['no_overlaps_left'] = ok
    print(json.dumps(out, indent=1))
    ctx.close()


if __name__ == '__main__':
    main()
