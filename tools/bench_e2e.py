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
    ap.add_argument('--pair-frac', type=float, default=0.009, help="generator: matched DEL + INS events (0.009 = bench.py's workload, ~1 k flagged loci; 0 = round 3's e2e workload)")
    ap.add_argument('--inv-sig-filter', default='svindel', help="config inv_sig_filter (reference default 'svindel')")
    ap.add_argument('--threads', type=int, default=min(64, effective_cpus()),
                    help='host threads of the native table writers (text + gzip members; the library default is 16)')
    ap.add_argument('--gzip-level', type=int, default=1,
                    help='deflate level of the native writers (the library default is 6; pandas / gzip.open use 9).  The text inside is '
                         'the same at every level; level 1 compresses 3 - 4 x faster and 25 - 35 % larger')
    ap.add_argument('--repeat', type=int, default=1, help='call the haplotype this many times in the process (fresh output directory, fresh context '
                                                          'and sequences each time); the line reports the LAST run and lists every total - the first '
                                                          'run of a process pays for pinned buffers, device allocations and cold pools')
    ap.add_argument('--bgzf', action='store_true', help='the two FASTA files bgzipped (ref.fa.gz, contigs_h1.fa.gz) - the form PAV keeps them in '
                                                        '(rules/call.snakefile:796); PAV_FASTA_INFLATE=host inflates them on host threads instead of the device')
    args = ap.parse_args()
    import torch  # noqa: F401  first: one HIP runtime per process (pav_amd/_lib.py)
    import __graft_entry__ as g
    g.build_cpu_side()
    import pandas as pd
    from pav_amd import _lib, rules, synth

    work = args.out or tempfile.mkdtemp(prefix='pav_e2e_')
    os.makedirs(work, exist_ok=True)
    t0 = time.time()
    hap = synth.config2(seed=args.seed, scale=args.scale, threads=min(16, effective_cpus()), **({'pair_frac': args.pair_frac} if args.pair_frac > 0 else {}))
    ref_fa, tig_fa = os.path.join(work, 'ref.fa'), os.path.join(work, 'contigs_h1.fa')
    synth.write_fasta(ref_fa, hap.ref.names, hap.ref.seqs, line=args.fasta_line)
    synth.write_fasta(tig_fa, hap.tig_names, hap.tig_seqs, line=args.fasta_line)
    if args.bgzf:
        for plain in (ref_fa, tig_fa):
            synth.bgzip(plain, plain + '.gz', threads=min(16, effective_cpus()))
            shutil.copyfile(plain + '.fai', plain + '.gz.fai')
            os.remove(plain)
        ref_fa, tig_fa = ref_fa + '.gz', tig_fa + '.gz'
    df_align = hap.df_align.copy()
    if 'CALL_BATCH' not in df_align:
        df_align['CALL_BATCH'] = df_align['INDEX'] % 10
    bed, bed_trim = os.path.join(work, 'aligned_tig_h1.bed.gz'), os.path.join(work, 'aligned_tig_h1.trim.bed.gz')
    df_align.to_csv(bed, sep='\t', index=False, compression={'method': 'gzip', 'compresslevel': 1})
    hap.df_trim.to_csv(bed_trim, sep='\t', index=False, compression={'method': 'gzip', 'compresslevel': 1})
    t_inputs = time.time() - t0
    totals = []
    for rep in range(max(1, args.repeat)):
        stages = {}
        shutil.rmtree(os.path.join(work, 'out'), ignore_errors=True)
        with _lib.Context(0) as ctx:
            man = rules.call_haplotype(bed, bed_trim, tig_fa, ref_fa, 'sample', 'h1', os.path.join(work, 'out'), ctx=ctx,
                                       config={'inv_sig_filter': args.inv_sig_filter}, threads=args.threads, gzip_level=args.gzip_level,
                                       timings=stages)
        totals.append(round(sum(stages.values()), 3))
    stages = {k: round(v, 3) for k, v in stages.items()}
    aligned_bp, n_snv, n_ins = man['aligned_bp'], man['snv_rows'], man['insdel_rows']
    df_flag = pd.read_csv(man['files']['flagged_regions'], sep='\t')
    regions = [None] * man['scanned_regions']
    calls = [None] * man['inv_calls']
    work_files = os.path.join(work, 'out')
    total = round(sum(stages.values()), 3)
    sizes = {}
    for base, _, files in os.walk(work_files):
        for f in files:
            if '/batch/' in base or '/log/' in base or '/density_table' in base:
                key = os.path.relpath(base, work_files) + '/*'
                sizes[key] = sizes.get(key, 0) + os.path.getsize(os.path.join(base, f))
            else:
                sizes[os.path.relpath(os.path.join(base, f), work_files)] = os.path.getsize(os.path.join(base, f))
    print(json.dumps({'workload': f'one synthetic hg38-shaped haplotype, seed {args.seed}, scale {args.scale}, pair_frac {args.pair_frac} (pav_amd.rules.call_haplotype)', 'aligned_bp': aligned_bp,
                      'snv_rows': n_snv, 'insdel_rows': n_ins, 'flagged_regions': int(df_flag.shape[0]), 'scanned_regions': len(regions),
                      'inv_calls': len(calls), 'inv_sig_filter': args.inv_sig_filter, 'stages_s': stages, 'total_s': total, 'runs_total_s': totals,
                      'writer': os.environ.get('PAV_WRITER', 'device'),
                      'fasta': ('BGZF, inflated on the ' + os.environ.get('PAV_FASTA_INFLATE', 'device')) if args.bgzf else 'plain text',
                      'fasta_bytes': os.path.getsize(ref_fa) + os.path.getsize(tig_fa),
                      'end_to_end_Gbp_per_s': round(aligned_bp / total / 1e9, 3), 'writer_threads': args.threads, 'gzip_level': args.gzip_level, 'inputs_written_s': round(t_inputs, 1),
                      'host_cores': os.cpu_count(), 'usable_cores': effective_cpus(), 'file_bytes': sizes}), flush=True)
    if args.out is None:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == '__main__':
    main()
