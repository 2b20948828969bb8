#!/usr/bin/env python3
"""pav_amd.rules.run_cohort on synthetic haplotypes, files to files: H haplotypes of one reference (hg38-shaped, shrunk by --scale)
are written as FASTA + alignment tables, then called by 1 rank and - when the box has them - by one rank per GPU.
    python tools/bench_cohort.py [--scale 0.25] [--haplotypes 4] [--ranks N] [--share-gpu]
--share-gpu puts all ranks on GPU 0 (what the tests do on tiny inputs).  It is NOT a way to use one GPU harder: two PROCESSES
time-slice a GPU, and a path of thousands of short kernels with a synchronisation every few of them then runs an order of
magnitude slower (measured: 4 haplotypes in 12 s with one rank, 504 s with two ranks on one GPU,
profiles/r04_cohort_two_ranks_one_gpu.json).  Several haplotypes per GPU are threads of ONE process (bench.py's lanes).
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scale', type=float, default=0.25)
    ap.add_argument('--haplotypes', type=int, default=4)
    ap.add_argument('--ranks', type=int, default=0, help='ranks of the second run (0 = one per visible GPU; skipped on a one-GPU box)')
    ap.add_argument('--share-gpu', action='store_true', help='all ranks on GPU 0 (see above: a correctness device, not a fast one)')
    ap.add_argument('--seed', type=int, default=1004)
    ap.add_argument('--bgzf', action='store_true', help='the FASTA files bgzipped (ref.fa.gz, contigs_{hap}.fa.gz: the form PAV keeps them in); their members '
                                                        'are inflated on the device (PAV_FASTA_INFLATE=host: by host threads)')
    ap.add_argument('--lanes', type=int, nargs='+', default=[1, 4], help="haplotypes a rank calls at the same time (config 'pav_amd_lanes'); one run per value")
    ap.add_argument('--gpus', type=int, default=0, help='ONE run on this many GPUs (one rank each, --lanes[0] lanes per rank) and ONE JSON line in the '
                                                       "schema of bench.py: metric haplotypes/s files to files, per-rank wall time, device and cores")
    args = ap.parse_args()
    import __graft_entry__ as g
    g.build_cpu_side()
    from pav_amd import cohort, synth
    from pav_amd.shard import effective_cpus
    work = tempfile.mkdtemp(prefix='pav_cohort_')
    try:
        t0 = time.time()
        ref = None
        jobs = []
        aligned = 0
        for h in range(args.haplotypes):
            hap = synth.config2(seed=args.seed, scale=args.scale, hap_index=h, ref=ref, threads=min(16, effective_cpus()), pair_frac=0.009)
            ref = hap.ref
            asm, hname = f'sample{h // 2}', f'h{h % 2 + 1}'
            d = os.path.join(work, 'in', asm)
            os.makedirs(d, exist_ok=True)
            if h == 0:
                synth.write_fasta(os.path.join(work, 'in', 'ref.fa'), ref.names, ref.seqs, line=80)
            tig = os.path.join(d, f'contigs_{hname}.fa')
            synth.write_fasta(tig, hap.tig_names, hap.tig_seqs, line=80)
            if args.bgzf:
                for plain in ([os.path.join(work, 'in', 'ref.fa')] if h == 0 else []) + [tig]:
                    synth.bgzip(plain, plain + '.gz', threads=min(16, effective_cpus()))
                    shutil.copyfile(plain + '.fai', plain + '.gz.fai')
                    os.remove(plain)
                tig += '.gz'
            bed, bed_trim = os.path.join(d, f'aligned_tig_{hname}.bed.gz'), os.path.join(d, f'aligned_tig_{hname}.trim.bed.gz')
            df = hap.df_align.copy()
            if 'CALL_BATCH' not in df:
                df['CALL_BATCH'] = df['INDEX'] % 10
            df.to_csv(bed, sep='\t', index=False, compression={'method': 'gzip', 'compresslevel': 1})
            hap.df_trim.to_csv(bed_trim, sep='\t', index=False, compression={'method': 'gzip', 'compresslevel': 1})
            jobs.append(cohort.HaplotypeJob(asm, hname, tig, bed, bed_trim))
            aligned += hap.stats['aligned_bp']
            del hap
        t_inputs = time.time() - t0
        ref_fa = os.path.join(work, 'in', 'ref.fa' + ('.gz' if args.bgzf else ''))
        out = {'fasta': ('BGZF, inflated on the ' + os.environ.get('PAV_FASTA_INFLATE', 'device')) if args.bgzf else 'plain text'}
        cfg = {'inv_sig_filter': 'single_cluster'}
        if args.gpus > 0:
            lanes = args.lanes[0]
            t0 = time.time()
            ms = cohort.run_cohort(jobs, args.gpus, os.path.join(work, 'out'), ref_fa,
                                   config=dict(cfg, pav_amd_lanes=lanes), share_gpu=args.share_gpu, timeout=3600)
            dt = time.time() - t0
            per_rank = {}
            for m in ms:
                r = per_rank.setdefault(m['rank'], {'rank': m['rank'], 'haplotypes': 0, 'wall_s': m.get('rank_wall_s'), 'device_name': m.get('device_name'),
                                                    'pci_bus_id': m.get('pci_bus_id'), 'usable_cores': m.get('rank_cores')})
                r['haplotypes'] += 1
            print(json.dumps({'metric': 'haplotypes/s, files to files (pav_amd.rules.run_cohort: FASTA + alignment tables in, every table of the rule chain out)',
                              'value': round(len(jobs) / dt, 3), 'unit': 'haplotypes/s', 'n_gpus': args.gpus, 'steps': len(jobs), 'warmup': 0,
                              'ms_per_step': round(dt / len(jobs) * 1e3, 1), 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
                              'dtype': 'u8/u32 + f64', 'data': 'synthetic',
                              'aligned_Gbp_per_s': round(aligned / dt / 1e9, 3), 'wall_s': round(dt, 2), 'per_rank': [per_rank[k] for k in sorted(per_rank)],
                              'config': {'workload': f'{args.haplotypes} synthetic haplotypes of one hg38-shaped reference, scale {args.scale}, seed {args.seed}',
                                         'lanes_per_rank': lanes, 'gzip_level': 6, 'writer': os.environ.get('PAV_WRITER', 'device'), 'fasta': out['fasta'],
                                         'inputs_written_s': round(t_inputs, 1), 'includes_process_start': args.gpus > 1}}))
            return
        import torch
        n_dev = torch.cuda.device_count()                      # (counting devices does not initialise the GPU in this process)
        ranks = args.ranks or n_dev
        worlds = [1] + ([ranks] if ranks > 1 and (args.share_gpu or ranks <= n_dev) else [])
        for world in worlds:
            for lanes in args.lanes:
                t0 = time.time()
                ms = cohort.run_cohort(jobs, world, os.path.join(work, f'out{world}_{lanes}'), ref_fa,
                                       config=dict(cfg, pav_amd_lanes=lanes), share_gpu=args.share_gpu, timeout=3600)
                dt = time.time() - t0
                out[f'{world}_rank' + ('s' if world > 1 else '') + f'_{lanes}_lanes'] = {
                    'wall_s': round(dt, 2), 'haplotypes_per_s': round(len(jobs) / dt, 3), 'aligned_Gbp_per_s': round(aligned / dt / 1e9, 3),
                    'inv_calls': sum(m['inv_calls'] for m in ms), 'ranks_used': sorted({m['rank'] for m in ms})}
                shutil.rmtree(os.path.join(work, f'out{world}_{lanes}'), ignore_errors=True)
        print(json.dumps({'workload': f'{args.haplotypes} synthetic haplotypes, scale {args.scale}, seed {args.seed}, one reference; files to files through '
                                      'pav_amd.rules.run_cohort (whole haplotypes per rank)' + (', all ranks on GPU 0' if args.share_gpu else ''), 'gpus_visible': n_dev,
                          'aligned_bp': aligned, 'inputs_written_s': round(t_inputs, 1), 'usable_cores': effective_cpus(), **out}))
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == '__main__':
    main()
