#!/usr/bin/env python3
"""End to end, files to files, for one synthetic hg38-shaped haplotype on one MI355X (reported next to bench.py's device
figure; SURVEY.md section 8(d): "end-to-end incl. H2D, formatting, gzip reported separately").

Inputs written first (not timed): ref.fa / contigs.fa (+ .fai), the trim-none and trim-tigref alignment tables (gzip TSV).
Timed stages, each with the rule(s) of the reference it stands for:
  sequences   FASTA -> host arrays -> HBM (2-bit + mask planes)                    pysam.FastaFile fetches (cigarcall.py:59-75)
  call        alignment table -> CIGAR calls of all rows -> merged SNV / INS-DEL tables    call_cigar x 10 + call_cigar_merge
  flag        signature flagging of the resident calls -> flagged regions table            call_inv_cluster x 2,
                                                                                           call_inv_flag_insdel_cluster x 2,
                                                                                           call_inv_merge_flagged_loci
  scan        k-mer density scan of every flagged region -> INV BED, density tables, log   call_inv_batch (all batches) + merge
    python tools/bench_e2e.py [--scale 1.0] [--out DIR]
"""
import argparse
import io
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pav_amd.shard import effective_cpus  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scale', type=float, default=1.0)
    ap.add_argument('--seed', type=int, default=1002)
    ap.add_argument('--out', default=None)
    ap.add_argument('--fasta-line', type=int, default=80)
    ap.add_argument('--inv-sig-filter', default='svindel', help="config inv_sig_filter (reference default 'svindel')")
    ap.add_argument('--threads', type=int, default=min(64, effective_cpus()),
                    help='host threads of the native table writers (text + gzip members; the library default is 16)')
    ap.add_argument('--gzip-level', type=int, default=1,
                    help='deflate level of the native writers (the library default is 6; pandas / gzip.open use 9).  The text inside is '
                         'the same at every level; level 1 compresses 3 - 4 x faster and 25 - 35 % larger')
    args = ap.parse_args()
    import torch  # noqa: F401  first: one HIP runtime per process (pav_amd/_lib.py)
    import __graft_entry__ as g
    g.build_cpu_side()
    import numpy as np
    import pandas as pd
    from pav_amd import _lib, cigarcall, flag, rules, synth, inv as pavinv, seq as pavseq
    from pav_amd.align import AlignLift
    from pav_amd.fasta import open_fasta, read_fai
    from pav_amd.kmer import KmerUtil

    work = args.out or tempfile.mkdtemp(prefix='pav_e2e_')
    os.makedirs(work, exist_ok=True)
    t0 = time.time()
    hap = synth.config2(seed=args.seed, scale=args.scale, threads=min(16, effective_cpus()))
    ref_fa, tig_fa = os.path.join(work, 'ref.fa'), os.path.join(work, 'contigs_h1.fa')
    synth.write_fasta(ref_fa, hap.ref.names, hap.ref.seqs, line=args.fasta_line)
    synth.write_fasta(tig_fa, hap.tig_names, hap.tig_seqs, line=args.fasta_line)
    df_align = hap.df_align.copy()
    if 'CALL_BATCH' not in df_align:
        df_align['CALL_BATCH'] = df_align['INDEX'] % 10
    bed, bed_trim = os.path.join(work, 'aligned_tig_h1.bed.gz'), os.path.join(work, 'aligned_tig_h1.trim.bed.gz')
    df_align.to_csv(bed, sep='\t', index=False, compression={'method': 'gzip', 'compresslevel': 1})
    hap.df_trim.to_csv(bed_trim, sep='\t', index=False, compression={'method': 'gzip', 'compresslevel': 1})
    t_inputs = time.time() - t0
    aligned_bp = None
    stages = {}

    def stage(name, t_start):
        stages[name] = round(time.time() - t_start, 3)

    with _lib.Context(0) as ctx:
        # ---- sequences ---------------------------------------------------------------------------------------------
        # the contig file is parsed on a second thread while the reference is parsed and uploaded (the readers and the upload
        # release the GIL): parse, H2D and pack of the two files overlap
        import threading
        t = time.time()
        box = {}
        th = threading.Thread(target=lambda: box.__setitem__('tig', open_fasta(tig_fa)))
        th.start()
        fa_ref = open_fasta(ref_fa)
        ctx.seq_load_fasta(_lib.PAV_ROLE_REF, fa_ref.native, fa_ref.record_numbers(fa_ref.names))
        th.join()
        fa_tig = box['tig']
        ctx.seq_load_fasta(_lib.PAV_ROLE_TIG, fa_tig.native, fa_tig.record_numbers(fa_tig.names))
        ctx.sync()
        stage('sequences: parse FASTA, H2D + pack (the two files overlapped)', t)
        # ---- call: all rows, merged tables -----------------------------------------------------------------------------
        t = time.time()
        table, trim_table = _lib.BedTable(bed, with_cigar=True), _lib.BedTable(bed_trim, with_cigar=False)
        cols = table.fetch()
        index = ctx.cigar_load_bed(table, -1)
        counts = ctx.cigar_call()
        aligned_bp = int(counts.aligned_bases)
        tc = trim_table.fetch()
        trim = pd.DataFrame({'POS': tc['POS'], 'END': tc['END']}, index=tc['INDEX']).astype(int).reindex(list(index), fill_value=-1)
        tp, te = trim['POS'].to_numpy(dtype='int64'), trim['END'].to_numpy(dtype='int64')
        stage('call: read tables + CIGAR-call', t)
        t = time.time()
        n_snv, n_ins = ctx.cigar_write_tables('h1', index, tp, te, snv_path=os.path.join(work, 'snv_snv_h1.bed.gz'),
                                              insdel_path=os.path.join(work, 'svindel_insdel_h1.bed.gz'), call_batch=cols['CALL_BATCH'],
                                              threads=args.threads, gzip_level=args.gzip_level)
        stage('call: merged tables (sort, text, gzip)', t)
        # ---- flag ----------------------------------------------------------------------------------------------------------
        t = time.time()
        res = flag.flag_from_calls(ctx, tp, te, inv_sig_filter=args.inv_sig_filter)
        for name in rules.FLAG_OUTPUTS:
            res[name].to_csv(os.path.join(work, f'flag_{name}_h1.bed.gz'), sep='\t', index=False, compression='gzip')
        df_flag = res['flagged_regions']
        stage('flag: five tables', t)
        # ---- scan ----------------------------------------------------------------------------------------------------------
        t = time.time()
        ctx._inv_loaded = (ref_fa, tig_fa)                                  # the sequences are resident already
        lift = AlignLift(pd.read_csv(bed_trim, sep='\t'), read_fai(tig_fa + '.fai'), ctx=ctx)
        stage('scan: lift-over index of the trimmed table', t)
        t = time.time()
        df_try = df_flag.loc[df_flag['TRY_INV']] if 'TRY_INV' in df_flag else df_flag
        regions = [pavseq.Region(r['#CHROM'], r['POS'], r['END']) for _, r in df_try.iterrows()]
        log = io.StringIO()
        import contextlib
        with contextlib.redirect_stdout(io.StringIO()):
            out = pavinv.scan_for_inv_batch(regions, ref_fa, tig_fa, lift, KmerUtil(31), log=log, ctx=ctx, eager_tables=False)
        stage('scan: density scan of the flagged regions', t)
        t = time.time()
        calls = [(i, c) for i, c in enumerate(out) if c is not None and not isinstance(c, RuntimeError)]
        den_dir = os.path.join(work, 'density_table')
        os.makedirs(den_dir, exist_ok=True)
        ctx.inv_write_tables([i for i, _ in calls], [os.path.join(den_dir, f'density_{c.id}_h1.tsv.gz') for _, c in calls],
                             threads=args.threads, gzip_level=args.gzip_level)
        stage('scan: density tables (text, gzip)', t)
        t = time.time()
        rows = [rules.inv_bed_row(c, 'h1', df_try.iloc[i]['TYPE'] if 'TYPE' in df_try else 'NA', tig_fa) for i, c in calls]
        stage('scan: INV BED rows (regions, SEQ of every call)', t)
        t = time.time()
        if rows:
            pd.concat(rows, axis=1).T.sort_values(['#CHROM', 'POS', 'END', 'ID']).to_csv(
                os.path.join(work, 'sv_inv_h1.bed.gz'), sep='\t', index=False, compression={'method': 'gzip', 'compresslevel': args.gzip_level})
        with open(os.path.join(work, 'inv_call_h1.log'), 'w') as fh:
            fh.write(log.getvalue())
        stage('scan: INV BED table (sort, text, gzip) + log', t)
    total = round(sum(stages.values()), 3)
    sizes = {f: os.path.getsize(os.path.join(work, f)) for f in sorted(os.listdir(work)) if os.path.isfile(os.path.join(work, f))}
    print(json.dumps({'workload': f'one synthetic hg38-shaped haplotype, seed {args.seed}, scale {args.scale}', 'aligned_bp': aligned_bp,
                      'snv_rows': n_snv, 'insdel_rows': n_ins, 'flagged_regions': int(df_flag.shape[0]), 'scanned_regions': len(regions),
                      'inv_calls': len(calls), 'inv_sig_filter': args.inv_sig_filter, 'stages_s': stages, 'total_s': total,
                      'end_to_end_Gbp_per_s': round(aligned_bp / total / 1e9, 3), 'writer_threads': args.threads, 'gzip_level': args.gzip_level, 'inputs_written_s': round(t_inputs, 1),
                      'host_cores': os.cpu_count(), 'usable_cores': effective_cpus(), 'file_bytes': sizes}), flush=True)
    if args.out is None:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == '__main__':
    main()
