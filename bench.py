#!/usr/bin/env python3
"""
bench.py - aligned Gbp/s through CIGAR-call + k-mer inversion scan on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

Workload (config.workload): synthetic hg38-shaped haplotypes (24 reference sequences with hg38 no-ALT lengths, ~3.0 Gbp
aligned each, SURVEY.md section 8(d) profile, seed 1002) through the WHOLE path the metric names: CIGAR-call,
inversion-signature flagging of the fresh calls, and the k-mer density scan of every flagged region - the per-GPU share of
BASELINE configs[2] / [3]: L = 6 haplotypes resident per GPU against one resident reference (--lanes; one context and one host
thread each; a waiting thread yields its core, so the lanes do not follow the CPUs of a rank), the K steps go round them.  The CIGAR-call-only figure of configs[1] is measured in the same run and reported
as the "cigar_only" object (or as `value` with --workload cigar).  With N > 1 every rank has its own L haplotypes (seed
1002*64 + rank*L + lane) against the same reference: weak scaling, no data-path collective;
torch.distributed (RCCL) is used only for the barrier, the max-over-ranks of the timed region and the sum of the bases.

A "step" is one pass of the hot path over one haplotype with inputs already resident in HBM
(reference ASCII + packed planes, contig ASCII, alignment tables, CIGAR text):
    contig planes marked stale (filled on demand below)  ->  tokenise CIGAR text  ->  prefix-scan walk  ->  SNV/INDEL emission
    ->  left-shift + breakpoint homology (contig windows decoded from the ASCII arena)  ->  SEQ gather
        (call records stay in HBM, D2H reported separately)
    ->  FILTER + key compaction + cluster sweeps + INS/DEL matching  ->  flagged loci
    ->  per flagged region: lift-over, pack of the blocks under the region, k-mer sets, STATE_MER, KDE, STATE runs,
        expansion rounds, inversion calls (the density tables of the calls stay packed in HBM; --eager-tables copies them
        to pinned host memory inside the step).

Output: stdout carries ONE short JSON line (<= 4 KB: the contract keys, `roofline`, `cpu_baseline`, one record per rank, the two
numbers of `cigar_only` / `verify_mode` / `inv_scan`); the full report goes to --detail (default bench_detail.json beside this
file) and to stderr: "roofline" (the kernel with the largest share of a pass, HIP-event timed on the library's streams, one lane
alone; "path" = the whole pass against SURVEY 8(d)'s byte model; "timed_region" = the ranking with every lane running),
"contig_pack_alone", "cpu_baseline" (oracle/ scalar C port timed on a bounded sample of the same workload, rank 0, N = 1 only),
"cigar_only" (BASELINE configs[1]), "verify_mode" (CIGAR-call + a pass over both packed sequences that checks every = / X base;
SURVEY.md section 8(d), never mixed into `value`), "inv_scan", "end_to_end" (writers / readers), "per_rank", "load_balance", "hbm",
"host".  N > 1: "single_rank_same_lanes" = rank 0 alone with the same lanes while the other ranks wait (the N = 1 figure a scaling
efficiency is to be computed against).
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
FP64_VECTOR_PEAK_TFLOPS = 78.6   # half of the guide's 157.3 TFLOP/s FP32 vector peak (256 CUs x 4 SIMD-32 x FMA x 2.4 GHz); the KDE has no MFMA form


PAIR_FRAC = 0.009          # matched DEL + INS events of the generator: ~1 k MATCH_INDEL loci per haplotype (~1 k flagged regions)


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def _newest_first(suffix):
    """Committed round summaries profiles/rNN<suffix>, newest round first."""
    import re
    try:
        names = [f for f in os.listdir(os.path.join(ROOT, 'profiles')) if re.fullmatch(r'r\d\d' + re.escape(suffix), f)]
    except OSError:
        names = []
    return sorted(names, reverse=True)


def compact_line(line, limited=False, solo=None):
    """The one line the driver parses (<= 4 KB): the contract keys, `roofline` and `cpu_baseline` with the figures they are
    checked by, one short record per rank.  Every other object of the report lives in the detail file."""
    def pick(d, keys):
        return None if d is None else {k: d.get(k) for k in keys if k in d}
    cfg = line['config']
    roof = line['roofline']
    out = {k: line[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                                'vs_baseline', 'dtype', 'data')}
    out['config'] = {'workload': cfg['workload_short'], **pick(cfg, ('scale', 'seed', 'aligned_bp_per_gpu', 'lanes_per_gpu',
                                                                     'usable_cpus_per_rank', 'reference')),
                     'repeats': f"median of {line['repeats']['regions']} regions of {line['steps']} steps",
                     'warmup_steps_run': cfg['warmup_steps_run']}
    r = pick(roof, ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'algorithmic_bytes_per_launch', 'avg_kernel_ms'))
    r['path'] = pick(roof.get('path'), ('achieved', 'frac'))
    iso = roof.get('isolated_lines')
    if iso:
        r['gather'] = {'glines_per_s': iso['achieved_glines_per_s'], 'peak': iso['measured_peak_glines_per_s'], 'frac': iso['frac']}
    tr = roof.get('timed_region')
    if tr:
        r['timed_region'] = pick(tr, ('kernel', 'share_of_device_time', 'avg_kernel_ms', 'device_ms_per_step'))
    out['roofline'] = r
    cpu = line.get('cpu_baseline')
    if cpu is not None:
        c = pick(cpu, ('value', 'unit', 'cores', 'kind', 'records_match_gpu', 'density_tables_match_gpu', 'cigar_call_only',
                       'density_scan_bp_per_s'))
        c['sample'] = cpu.get('sample_short') or cpu['sample'][:200]
        if 'all_cores' in cpu:
            c['all_cores'] = pick(cpu['all_cores'], ('value', 'cigar_call_only', 'cores'))
        c['reference_python'] = pick(cpu['reference_python'], ('cigar_call_Mbp_per_s', 'density_scan_kbp_per_s', 'cores'))
        c['reference_python']['hardware'] = 'survey sandbox, 1 core (BASELINE.md section 2)'
        out['cpu_baseline'] = c
    else:
        out['cpu_baseline'] = None
    out['per_rank'] = [{'rank': p['rank'], 'ms_per_step': p['ms_per_step'], 'lanes_per_gpu': p['lanes_per_gpu'],
                        'usable_cpus': p['usable_cpus'], 'aligned_bp': p['aligned_bp']} for p in line['per_rank']]
    if limited:
        out['lanes_limited_by_cpus'] = True
    if solo is not None:
        out['single_rank_same_lanes'] = solo
    for k in ('cigar_only', 'verify_mode'):
        if line.get(k):
            out[k] = pick(line[k], ('value', 'ms_per_step'))
    if line.get('inv_scan'):
        out['inv_scan'] = pick(line['inv_scan'], ('calls', 'scanned_loci', 'device_ms_per_step'))
    out['detail_file'] = line.get('detail_file')
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=64)
    ap.add_argument('--warmup', type=int, default=8)
    ap.add_argument('--scale', type=float, default=1.0, help='shrink every sequence length (tests only; 1.0 = the named workload)')
    ap.add_argument('--seed', type=int, default=1002)
    ap.add_argument('--cpu-sample-bp', type=float, default=4e9,
                    help='reference span (bp) of the CPU-baseline sample; default = the whole haplotype (a few seconds of CPU)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--threads', type=int, default=0, help='host threads for the generator (0 = auto)')
    ap.add_argument('--workload', choices=['cigar+inv', 'cigar'], default='cigar+inv',
                    help="'cigar+inv' = the whole path of the metric: CIGAR-call + flagging + k-mer inversion scan of every flagged "
                         "locus (configs[2]: both haplotypes of a diploid sample per GPU); 'cigar' = BASELINE configs[1], CIGAR-call only")
    ap.add_argument('--lanes', type=int, default=0,
                    help='haplotypes resident per GPU, one context + host thread each, sharing one resident reference; the K steps '
                         'go round them (default 0 = 6: h1 + h2 of three phased diploid samples, the per-GPU share of configs[2] / [3]; '
                         'a waiting lane yields its core, so six lanes need no six cores; 1 = one haplotype, no overlap)')
    ap.add_argument('--pair-frac', type=float, default=PAIR_FRAC, help='generator: fraction of indel events emitted as a matched DEL + INS')
    ap.add_argument('--eager-tables', action='store_true',
                    help='copy the density tables of every inversion call to pinned host memory inside the timed region (round-1 '
                         'behaviour); default: they stay in HBM like the SNV / INDEL records, their D2H time is reported in `host`')
    ap.add_argument('--cpu-sample-regions', type=int, default=300,
                    help='flagged loci whose k-mer density scan the CPU baseline times (oracle, one core)')
    ap.add_argument('--backend', default='nccl', help="process-group backend for N > 1 ('nccl' = RCCL; tests use 'gloo')")
    ap.add_argument('--share-gpu', action='store_true',
                    help='tests only: every rank uses GPU 0 (exercises the N > 1 code path on a one-GPU box; needs --backend gloo)')
    ap.add_argument('--eager-pack', action='store_true',
                    help='pack the whole contig arena at the start of every pass (PAV_EAGER_PACK=1: the round-1 behaviour) instead of on demand')
    ap.add_argument('--no-build', action='store_true', help='do not (re)build the libraries: the profile scripts build first, outside the profiler')
    ap.add_argument('--repeats', type=int, default=5,
                    help='the timed region of exactly K steps is run this many times back to back; `value` / `ms_per_step` are those of the '
                         'median region (one region = K steps, so steps x ms_per_step is one timed region), min / max / all in `repeats`')
    ap.add_argument('--reference', choices=['hg38', 'chm13'], default='hg38',
                    help="'hg38' (default): the hg38-shaped reference of configs[1]-[3]; 'chm13': BASELINE configs[4] - T2T-CHM13v2.0 lengths, "
                         'no N runs, seed 1005, the cohort batched --lanes (default 8) haplotypes per GPU against one resident reference; the '
                         'line then carries `hbm` (peak use, pav_mem_info) and every lane\'s records_match vs the oracle')
    ap.add_argument('--detail', default=os.path.join(ROOT, 'bench_detail.json'),
                    help='file that receives the full report (every side leg, per-kernel tables, notes); stdout carries one short JSON line')
    args = ap.parse_args()
    if args.reference == 'chm13':
        if args.seed == 1002:
            args.seed = 1005                                   # config n uses seed 1000 + n
        if args.lanes == 0:
            args.lanes = 8

    # ---- before anything touches the GPU: the N-rank launch and the build ------------------------------------------------
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks as a child job and hand its exit code on (this
        # process has not initialised the GPU; it never measures one GPU and calls it N)
        import subprocess
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)
    if args.eager_pack:
        os.environ['PAV_EAGER_PACK'] = '1'
    eager_pack = os.environ.get('PAV_EAGER_PACK') == '1'
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    import __graft_entry__ as g
    if not args.no_build:
        import fcntl
        with open(os.path.join(ROOT, '.build.lock'), 'w') as lock:       # ranks take turns; all but the first find everything built
            fcntl.flock(lock, fcntl.LOCK_EX)
            g.build_cpu_side()

    import torch  # first: its bundled HIP runtime becomes the process-wide one (pav_amd/_lib.py docstring)
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (no CPU fallback exists)')
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    comm_device = 'cuda' if args.backend == 'nccl' else 'cpu'
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    # the first contact with the other ranks, before any work: device count, one all_reduce that must sum to the world size, one
    # distinct GPU per rank (pav_amd/shard.py) - a fabric / IPC problem ends the run here with a message
    from pav_amd.shard import first_contact
    if args.backend == 'nccl':
        gpu_info = first_contact('nccl', rank, world, local_rank, share_gpu=args.share_gpu)
    else:
        gpu_info = first_contact(args.backend, rank, world, local_rank, share_gpu=True)
        gpu_info['device_name'] = torch.cuda.get_device_name(local_rank)
    print(f"[bench] rank {rank}/{world}: {gpu_info['device_name']} {gpu_info['pci_bus_id']}", file=sys.stderr, flush=True)

    import io
    import threading
    import numpy as np
    from pav_amd import _lib, cigarcall, synth

    from pav_amd.shard import effective_cpus
    threads = args.threads or max(1, effective_cpus() // max(1, world))
    threads = min(threads, 16)
    cpus_per_rank = effective_cpus() / max(1, world)
    # measured: a lane wants a core; six lanes fill the GPU (round 3, 20-step regions: 4 / 5 / 6 / 8 lanes = 1 870 / 2 010 / 2 090 / 2 090
    # Gbp/s - with the host gaps of a pass shortened the fifth and sixth lane find room that four lanes left; round 2: four;
    # round 4: 2 / 4 / 6 lanes = 1.65 - 1.79 / 2.25 - 2.33 / 2.26 - 2.36 Tbp/s: four lanes reach the sum of the per-phase floors,
    # profiles/r04_lane_scaling.txt - six are kept where the CPUs allow it, they cost nothing and smooth short timed regions)
    # round 6: a waiting lane no longer holds a core (the library polls an event and yields between the polls, PAV_WAIT in
    # include/pav_amd.h): six lanes on TWO cores run within a few per cent of six lanes on sixteen (2.38 / 2.32 Tbp/s on one box,
    # 2.15 on one core; with the runtime's spinning wait 1.62 on two), so the lanes no longer follow the CPUs of a rank
    n_lanes = args.lanes if args.lanes > 0 else 6
    gen_kw = {'pair_frac': args.pair_frac} if args.workload == 'cigar+inv' and args.pair_frac > 0 else {}
    if rank == 0:
        print(f'[bench] {world} rank(s) x {n_lanes} lane(s) per GPU (--lanes {args.lanes}: 0 = auto), {cpus_per_rank:.1f} usable CPUs per rank '
              f'(affinity / cgroup quota / world size), reference {args.reference}, {args.repeats} timed region(s) of {args.steps} steps',
              file=sys.stderr, flush=True)

    # ---- synthetic inputs (host) -> HBM -----------------------------------------------------------------------
    # One lane = one haplotype resident on the GPU: its own context (streams, contigs, alignment tables, results) sharing the
    # reference planes of lane 0 (pav_seq_share).  With several ranks on one node the ranks take turns (generate -> upload ->
    # free the host copies), so the node never holds more than one rank's host sequence at a time.
    class Lane:
        pass

    def prepare():
        lanes_ = []
        ref_ = None
        for li in range(n_lanes):
            ln = Lane()
            ln.idx = li
            t0 = time.time()
            make_hap = synth.config5 if args.reference == 'chm13' else synth.config2
            ln.hap = make_hap(seed=args.seed, scale=args.scale, hap_index=rank * n_lanes + li, ref=ref_, threads=threads, **gen_kw)
            ref_ = ln.hap.ref
            ln.t_gen = time.time() - t0
            ln.ctx = _lib.Context(local_rank)
            t0 = time.time()
            names_ = ref_.names
            if li == 0:
                ln.ctx.seq_load(_lib.PAV_ROLE_REF, names_, [ref_.seqs[n] for n in names_])
            else:
                ln.ctx.seq_share(lanes_[0].ctx, _lib.PAV_ROLE_REF)
            ln.ctx.seq_load(_lib.PAV_ROLE_TIG, ln.hap.tig_names, [ln.hap.tig_seqs[n] for n in ln.hap.tig_names])
            ln.aln, ln.text, ln.off = cigarcall.pack_alignments(ln.hap.df_align, names_, ln.hap.tig_names)
            ln.ctx.cigar_load(ln.aln, ln.text, ln.off)
            ln.ctx.sync()
            ln.t_h2d = time.time() - t0
            ln.tig_bases = int(sum(ln.hap.tig_seqs[n].shape[0] for n in ln.hap.tig_names))
            ln.tig_len = ln.hap.tig_lengths
            ln.records_match = None
            if args.reference == 'chm13' and not args.no_cpu_baseline:
                # every resident haplotype against the oracle's scalar walk, while its host copy still exists (untimed)
                from oracle import oracle
                c_ = ln.ctx.cigar_call()
                snv_, indel_, blob_ = ln.ctx.cigar_fetch(c_)
                o_snv, o_indel, o_blob, err_ = oracle.cigar_call([ref_.seqs[n] for n in names_], [ln.hap.tig_seqs[n] for n in ln.hap.tig_names],
                                                                 ln.aln, ln.text, ln.off)
                ok_ = err_.kind == 0 and snv_.tobytes() == o_snv.tobytes() and blob_.tobytes() == o_blob.tobytes()
                for f in o_indel.dtype.names:
                    if f != 'pad':
                        ok_ = ok_ and bool(np.array_equal(indel_[f], o_indel[f]))
                st_ = ln.hap.stats
                ok_ = ok_ and (c_.n_ops, c_.n_snv, c_.n_indel, c_.aligned_bases) == (st_['n_ops'], st_['n_snv'], st_['n_ins'] + st_['n_del'], st_['aligned_bp'])
                ln.records_match = bool(ok_)
                del snv_, indel_, blob_, o_snv, o_indel, o_blob
            if world > 1 or li > 0:                          # host copies are only needed for the N = 1 CPU baseline (lane 0)
                ln.hap.tig_seqs.clear()
            lanes_.append(ln)
        return lanes_

    def host_memory_available_gb():
        """MemAvailable of the node, capped by the cgroup's limit where there is one (GB); 0.0 when it cannot be read."""
        try:
            avail = 0.0
            with open('/proc/meminfo') as fh:
                for line in fh:
                    if line.startswith('MemAvailable:'):
                        avail = float(line.split()[1]) * 1024.0
            for lim, cur in (('/sys/fs/cgroup/memory.max', '/sys/fs/cgroup/memory.current'),
                             ('/sys/fs/cgroup/memory/memory.limit_in_bytes', '/sys/fs/cgroup/memory/memory.usage_in_bytes')):
                if os.path.exists(lim) and os.path.exists(cur):
                    with open(lim) as fh:
                        text_ = fh.read().strip()
                    if text_.isdigit() and int(text_) < (1 << 60):
                        with open(cur) as fh:
                            avail = min(avail, float(int(text_) - int(fh.read().strip())))
            return max(0.0, avail) / 1e9
        except Exception:                                     # noqa: BLE001 - unknown: the careful way below
            return 0.0

    # With several ranks on a node the ranks prepare their haplotypes at the same time when the node's memory allows it (a rank
    # holds the reference and one haplotype on the host while it generates and uploads: < 20 GB), each on its share of the cores;
    # otherwise they take turns.  (Six lanes on eight ranks are 48 haplotypes: one after the other, three minutes before the first pass.)
    all_at_once = False
    if world > 1:
        enough = torch.tensor([1.0 if host_memory_available_gb() >= 24.0 * world else 0.0], dtype=torch.float64, device=comm_device)
        dist.all_reduce(enough, op=dist.ReduceOp.MIN)         # (every rank asks the node; all of them must agree)
        all_at_once = float(enough.item()) > 0.5
    lanes = None
    for turn in range(1 if all_at_once else world):
        if all_at_once or turn == rank:
            lanes = prepare()
            ref_lengths = {n: int(lanes[0].hap.ref.seqs[n].shape[0]) for n in lanes[0].hap.ref.names}
            if world > 1:
                lanes[0].hap.ref.seqs.clear()
                import gc
                gc.collect()
        if world > 1:
            dist.barrier()
    hap, ctx = lanes[0].hap, lanes[0].ctx                    # lane 0 = h1: the side legs, the CPU baseline, the reports
    aln, text, off = lanes[0].aln, lanes[0].text, lanes[0].off
    tig_bases = lanes[0].tig_bases
    t_gen, t_h2d = sum(ln.t_gen for ln in lanes), sum(ln.t_h2d for ln in lanes)
    names = hap.ref.names

    if args.workload == 'cigar+inv':
        import tempfile
        from pav_amd import inv as pavinv
        from pav_amd.align import AlignLift
        from pav_amd.kmer import KmerUtil
        tmpd = tempfile.mkdtemp(prefix='pav_bench_')
        with open(os.path.join(tmpd, 'ref.fa.fai'), 'w') as fh:          # scan_for_inv reads "<ref>.fai" (inv.py:201)
            for n in names:
                fh.write(f'{n}\t{ref_lengths[n]}\t0\t0\t0\n')
        ref_fa_name, tig_fa_name = os.path.join(tmpd, 'ref.fa'), os.path.join(tmpd, 'tig.fa')
        k_util = KmerUtil(31)
        for ln in lanes:
            ln.ctx._inv_loaded = (ref_fa_name, tig_fa_name)               # sequences are already resident
            # FILTER inputs of the flagging pass (rules call_inv_cluster / call_inv_flag_insdel_cluster / call_inv_merge_flagged_loci)
            index_ = ln.hap.df_align['INDEX'].to_numpy(dtype='int64')
            trim_ = ln.hap.df_trim[['POS', 'END', 'INDEX']].set_index('INDEX').astype(int).reindex(list(index_), fill_value=-1)
            ln.flag_tp, ln.flag_te = trim_['POS'].to_numpy(dtype='int64'), trim_['END'].to_numpy(dtype='int64')
            # inv_sig_filter = single_cluster (CONFIG.md: try loci that only show a cluster of SNVs / indels): the planted inversions
            # are aligned through, which is what such a cluster is; the default filter would leave them to the large-SV caller
            ln.flag_params = ln.ctx.flag_params(sig_filter=_lib.SIG_SINGLE_CLUSTER)
            # The trimmed alignment table is an input of the job like the sequences and the CIGAR strings of the call path:
            # it is tokenised into the lift-over index on the device once, before the timed region.
            t_a = time.perf_counter()
            ln.lift = AlignLift(ln.hap.df_trim, ln.tig_len)
            ln.t_lift_ms = (time.perf_counter() - t_a) * 1e3
            ln.found = io.StringIO()
            # the same table as the arrays pav_inv_load_alignments takes (the `with_lift_index` leg rebuilds the index per step)
            ri_ = {n: i for i, n in enumerate(ln.ctx.seq_names(_lib.PAV_ROLE_REF))}
            ti_ = {n: i for i, n in enumerate(ln.ctx.seq_names(_lib.PAV_ROLE_TIG))}
            ln.lift_arrays = pavinv.pack_lift_table(ln.lift, ri_, ti_)

    def inv_step(ln):
        """Flagged loci of the fresh calls -> scan of every locus with TRY_INV (rules/call_inv.snakefile:145-196)."""
        ln.flag = ln.ctx.cigar_flag(ln.flag_tp, ln.flag_te, ln.flag_params)
        regions = pavinv.loci_regions(ln.ctx, ln.flag[1])
        log = io.StringIO()                                   # one log for the batch, as rule call_inv_batch keeps it
        t_b = time.perf_counter()
        ln.found.seek(0)
        ln.found.truncate()
        ln.out = pavinv.scan_for_inv_batch(regions, ref_fa_name, tig_fa_name, ln.lift, k_util, log=log, ctx=ln.ctx,
                                           eager_tables=args.eager_tables, found_out=ln.found)
        ln.log, ln.regions = log, regions
        ln.t_scan_ms = (time.perf_counter() - t_b) * 1e3

    phase_timing = bool(os.environ.get('PAV_TIMING'))

    def step(ln, workload=args.workload):
        t_p = [time.perf_counter()]

        def lap(what):
            if phase_timing:
                t = time.perf_counter()
                print('[pav timing] lane %d step %-12s %.2f ms' % (ln.idx, what, (t - t_p[0]) * 1e3), file=sys.stderr)
                t_p[0] = t
        ln.ctx.seq_pack(_lib.PAV_ROLE_TIG)
        ln.counts = ln.ctx.cigar_call()
        lap('cigar_call')
        if workload == 'cigar+verify':
            ln.verify = ln.ctx.cigar_verify()
        if workload == 'cigar+inv+lift':
            # what a cohort run pays per haplotype on top of a step: the lift-over index of the trimmed table (tokenised and
            # scanned on the device, brought to the host lookup tables) - the reference builds its AlignLift inside every
            # call_inv_batch job (rules/call_inv.snakefile:174-177)
            ln.ctx.inv_load_alignments(*ln.lift_arrays)
            ln.lift._native_loaded = ln.ctx
            lap('lift index')
        if workload in ('cigar+inv', 'cigar+inv+lift'):
            inv_step(ln)
            lap('flag + scan')

    def run_steps(n, workload=args.workload, only=None, fixed_share=False):
        """n passes of the hot path, one haplotype each: step s goes to lane s mod L; the lanes run on their own host threads
        (the library releases the GIL inside its calls), so one haplotype's kernels fill the gaps of the other's host work."""
        use = lanes if only is None else [only]
        for ln in use:
            ln.n_done = 0
        if len(use) == 1:
            for _ in range(n):
                step(use[0], workload)
            use[0].n_done = n
            return
        err = []

        # The lanes take their steps from one counter: a lane that gets ahead takes the next step, so a region does not end with
        # one lane finishing its fixed share alone (20-step regions: 2.59 -> 2.16 ms per step on the same box; 64-step regions
        # were at 2.1 already).  PAV_BENCH_STAGGER = cigar | step starts lane i behind lane i - 1's first CIGAR-call / first step
        # (measured: 2.29 / 2.88 ms per step - no help; default: all lanes start at once)
        counter = [0]
        lock = threading.Lock()
        started = [threading.Event() for _ in use]
        stagger = os.environ.get('PAV_BENCH_STAGGER', 'none')

        share = [0] * len(use)                                  # fixed_share (warm-up): every lane runs ceil(n / L) steps, whatever its pace

        def take(k=None):
            with lock:
                if fixed_share:
                    if share[k] >= (n + len(use) - 1) // len(use):
                        return False
                    share[k] += 1
                    return True
                if counter[0] >= n:
                    return False
                counter[0] += 1
                return True

        def worker(k, ln):
            try:
                if k and stagger != 'none':
                    started[k - 1].wait()
                first = True
                while take(k):
                    ln.n_done += 1
                    if first and stagger == 'cigar':
                        ln.ctx.seq_pack(_lib.PAV_ROLE_TIG)
                        ln.counts = ln.ctx.cigar_call()
                        started[k].set()
                        if workload == 'cigar+verify':
                            ln.verify = ln.ctx.cigar_verify()
                        if workload in ('cigar+inv', 'cigar+inv+lift'):
                            inv_step(ln)
                    else:
                        step(ln, workload)
                        started[k].set()
                    first = False
                started[k].set()
            except BaseException as ex:      # noqa: BLE001 - re-raised on the main thread
                started[k].set()
                err.append(ex)
        ths = [threading.Thread(target=worker, args=(k, ln)) for k, ln in enumerate(use)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if err:
            raise err[0]

    def fence():
        for ln in lanes:
            ln.ctx.sync()                # the library's three streams, incl. call-table copies still travelling to the host
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n, workload=args.workload):
        fence()
        t_a = time.perf_counter()
        run_steps(n, workload)
        for ln in lanes:
            ln.ctx.sync()
        torch.cuda.synchronize()
        t_c = time.perf_counter() - t_a
        fence()
        if world > 1:
            tt_ = torch.tensor([t_c], dtype=torch.float64, device=comm_device)
            dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
            return t_c, float(tt_.item())
        return t_c, t_c

    # The haplotypes built above are tens of millions of long-lived Python objects (tables, arrays, strings).  A full collection of
    # the cyclic garbage collector walks all of them - 80 ms with every lane's thread stopped, once every few hundred passes: one
    # timed region in ten came out three times as long as its neighbours.  They are moved to the permanent generation (what
    # gc.freeze is for); the collector stays on for what the passes themselves allocate.
    import gc
    gc.collect()
    gc.freeze()
    # Untimed warm-up: W steps as asked, but at least three per lane, in equal shares (not from the shared counter of the timed
    # regions: a lane that is still allocating would be overtaken and start its first pass inside a timed region - an 80 ms
    # outlier) - a lane's buffers are sized by its first pass and the two alternating table arenas of its scan by its first two
    warmup_run = max(args.warmup, 3 * n_lanes)
    run_steps(warmup_run, fixed_share=True)
    warmup_run = ((warmup_run + n_lanes - 1) // n_lanes) * n_lanes
    if n_lanes > 1:
        # ... and a lead-in from the shared counter, as the timed regions run: the lanes leave the warm-up in step (all of them at
        # their CIGAR-call at once) and need a few passes to spread over each other's gaps - the first timed region of a run was
        # a third slower than the others
        run_steps(4 * n_lanes)
        warmup_run += 4 * n_lanes

    # ---- timed region: exactly K steps, profiling off; run R times back to back, the median region is the line's --------
    free_min = [min(ln.ctx.mem_info()[0] for ln in lanes)]
    regions = []
    region_bp = []                                               # aligned bp of the K passes of each region (this rank): the lanes' shares vary
    for _ in range(max(1, args.repeats)):
        regions.append(timed(args.steps))
        region_bp.append(float(sum(ln.n_done * ln.counts.aligned_bases for ln in lanes)))
        free_min.append(lanes[0].ctx.mem_info()[0])
    order = sorted(range(len(regions)), key=lambda i: regions[i][1])
    t_local, t_max = regions[order[len(order) // 2]]            # median by the max-over-ranks time (upper median for even R)
    counts = lanes[0].counts
    aligned_steps = region_bp[order[len(order) // 2]]           # bp of the K passes of this rank in the region the line reports
    if world > 1:
        ab = torch.tensor([aligned_steps], dtype=torch.float64, device=comm_device)
        dist.all_reduce(ab, op=dist.ReduceOp.SUM)
        aligned_total = float(ab.item())
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {'rank': rank, 'ms_per_step': round(t_local / args.steps * 1e3, 4),
                                          'device_name': lanes[0].ctx.device_name, 'pci_bus_id': lanes[0].ctx.pci_bus_id,
                                          'lanes_per_gpu': n_lanes, 'usable_cpus': round(cpus_per_rank, 2),
                                          'aligned_bp': aligned_steps / args.steps,
                                          'cigar_text_bytes': int(sum(ln.text.shape[0] for ln in lanes))})
    else:
        aligned_total = aligned_steps
        per_rank = [{'rank': 0, 'ms_per_step': round(t_local / args.steps * 1e3, 4), 'device_name': lanes[0].ctx.device_name,
                     'pci_bus_id': lanes[0].ctx.pci_bus_id, 'lanes_per_gpu': n_lanes,
                     'usable_cpus': round(cpus_per_rank, 2), 'aligned_bp': aligned_steps / args.steps,
                     'cigar_text_bytes': int(sum(ln.text.shape[0] for ln in lanes))}]

    # ---- N > 1: rank 0 once more ALONE with the same lanes (the other ranks wait on a socket barrier, their GPUs idle): the N = 1
    #      figure at this rank's lane count.  The lanes of a rank follow its share of the node's CPUs, so a scaling run on a node with
    #      few CPUs per rank runs fewer lanes than the N = 1 run did - efficiency is like-for-like only against this figure.
    limited_by_cpus = bool(args.lanes == 0 and n_lanes < 6)
    solo_same_lanes = None
    if world > 1:
        side = None
        if args.backend != 'gloo':
            try:                                              # (a socket barrier: the waiting ranks sleep instead of spinning on the device)
                side = dist.new_group(backend='gloo')
            except Exception as ex:                           # noqa: BLE001 - no gloo transport on this node: the main group's barrier does
                print(f'[bench] rank {rank}: no gloo side group ({ex!r}); waiting on the main group', file=sys.stderr, flush=True)
        if rank == 0:
            solos = []
            for _ in range(3):
                for ln in lanes:
                    ln.ctx.sync()
                torch.cuda.synchronize()
                t_a = time.perf_counter()
                run_steps(args.steps)
                for ln in lanes:
                    ln.ctx.sync()
                torch.cuda.synchronize()
                t_s = time.perf_counter() - t_a
                solos.append((t_s, float(sum(ln.n_done * ln.counts.aligned_bases for ln in lanes))))
            t_s, bp_s = sorted(solos)[1]
            solo_same_lanes = {'value': round(bp_s / t_s / 1e9, 2), 'ms_per_step': round(t_s / args.steps * 1e3, 4), 'lanes': n_lanes}
        dist.barrier(group=side)

    # ---- the timed region once more with HIP events around every kernel of EVERY lane: which kernel takes the most device time
    #      in the regime the line is measured in (kernels of other lanes beside it), and how much longer than alone -----------
    lanes_prof, t_lanes_prof = None, None
    if n_lanes > 1:
        for ln in lanes:
            ln.ctx.prof_reset()
            ln.ctx.prof_enable(True)
        _, t_lanes_prof = timed(args.steps)
        lanes_prof = [ln.ctx.prof_read() for ln in lanes]
        for ln in lanes:
            ln.ctx.prof_enable(False)
            ln.ctx.prof_reset()
    # ---- the step plus the lift-over index of the trimmed table, rebuilt inside every step (reported beside `value`) ---------
    t_with_lift = None
    if args.workload == 'cigar+inv':
        run_steps(max(2, n_lanes), 'cigar+inv+lift')
        _, t_with_lift = timed(args.steps, 'cigar+inv+lift')

    # ---- the same steps on ONE lane with HIP events around every kernel (roofline leg): nothing of another haplotype
    #      beside them, so the sum of the kernel times is the device work of a step ---------------------------------------
    ctx.prof_reset()
    ctx.prof_enable(True)
    kde0 = ctx.kde_work()
    run_steps(args.steps, only=lanes[0])
    prof = ctx.prof_read()
    kde_leg = tuple(b - a for a, b in zip(kde0, ctx.kde_work()))       # evaluation points, (point, run) pairs, (point, data point) pairs
    ctx.prof_enable(False)
    t_single = None
    if n_lanes > 1:                                           # and timed without the events: what one lane alone achieves
        run_steps(3, only=lanes[0])                            # (the first passes after the event-profiled leg are not representative)
        singles = []
        for _ in range(3):                                    # the median of three regions, as for the line itself
            fence()
            t_a = time.perf_counter()
            run_steps(args.steps, only=lanes[0])
            ctx.sync()
            singles.append(time.perf_counter() - t_a)
        t_single = sorted(singles)[1]

    # ---- configs[1] in the same run: K steps of CIGAR-call only, timed and event-profiled the same way ----------------
    def side_leg(workload):
        run_steps(max(1, args.warmup, n_lanes), workload)
        _, t_c = timed(args.steps, workload)
        ctx.prof_reset()
        ctx.prof_enable(True)
        run_steps(args.steps, workload, only=lanes[0])
        leg = {'t': t_c, 'prof': ctx.prof_read()}
        ctx.prof_enable(False)
        return leg

    cigar_leg = side_leg('cigar') if args.workload == 'cigar+inv' else None
    # ---- verify mode (second line): CIGAR-call + a pass over both packed sequences that checks every
    #      '=' / 'X' base against the CIGAR; never mixed into `value` ------------------------------------------------------
    verify_leg = side_leg('cigar+verify')
    # the dominant kernel with nothing beside it: K packs of the resident contigs, each waited for (HIP events as above)
    # (the contig planes are packed on demand since round 2 - DESIGN.md section 3.1: the full streaming pack runs in verify
    #  mode and with PAV_EAGER_PACK=1; it is timed here with that switch, nothing beside it)
    ctx.prof_reset()
    ctx.prof_enable(True)
    eager_before = os.environ.get('PAV_EAGER_PACK')
    os.environ['PAV_EAGER_PACK'] = '1'
    for _ in range(args.steps):
        ctx.seq_pack(_lib.PAV_ROLE_TIG)
        ctx.sync()
    pack_alone = ctx.prof_read().get('pack_kernel')
    if eager_before is None:
        del os.environ['PAV_EAGER_PACK']
    else:
        os.environ['PAV_EAGER_PACK'] = eager_before
    # ... and the call kernels with the planes already packed and no pack beside them
    ctx.prof_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.cigar_call()
    ctx.sync()
    call_alone = {'ms_per_call': round((time.perf_counter() - t0) / args.steps * 1e3, 4),
                  'kernels_ms': {k: round(v[1] / max(1, v[0]), 4) for k, v in sorted(ctx.prof_read().items())}}
    ctx.prof_enable(False)

    # D2H of the record streams (reported, never part of `value`)
    t0 = time.perf_counter()
    snv, indel, blob = ctx.cigar_fetch(counts)
    t_d2h = time.perf_counter() - t0
    # ... and of the density tables of the inversion calls of lane 0's last scan (they stay packed in HBM unless --eager-tables)
    t_d2h_tables = tables_mb = None
    if args.workload == 'cigar+inv':
        calls0 = [(i, c) for i, c in enumerate(lanes[0].out) if c is not None and not isinstance(c, RuntimeError)]
        if calls0:
            t0 = time.perf_counter()
            views = [ctx.inv_table_view(i, c.native_table[2]) for i, c in calls0]      # the first one brings every round's block over
            t_d2h_tables = time.perf_counter() - t0
            tables_mb = sum(v[0]['INDEX'].shape[0] for v in views) * 40 / 1e6

    # End-to-end leg (N = 1 only, reported separately, never part of `value`): the two call_cigar tables of the whole
    # haplotype - FILTER, sort, TSV text, gzip - through the native writer, into a scratch directory.
    e2e = None
    if world == 1:
        import shutil
        import tempfile
        tmp_out = tempfile.mkdtemp(prefix='pav_bench_out_')
        try:
            index = hap.df_align['INDEX'].to_numpy(dtype='int64')
            trim = hap.df_trim[['POS', 'END', 'INDEX']].set_index('INDEX').astype(int).reindex(list(index), fill_value=-1)
            tp, te = trim['POS'].to_numpy(dtype='int64'), trim['END'].to_numpy(dtype='int64')
            t0 = time.perf_counter()
            n1, n2 = ctx.cigar_write_tables(hap.hap, index, tp, te, os.path.join(tmp_out, 'snv.bed.gz'), os.path.join(tmp_out, 'insdel.bed.gz'))
            t_gz = time.perf_counter() - t0
            t0 = time.perf_counter()
            ctx.cigar_write_tables(hap.hap, index, tp, te, os.path.join(tmp_out, 'snv.bed'), os.path.join(tmp_out, 'insdel.bed'))
            t_plain = time.perf_counter() - t0
            # reader half: the alignment table as PAV stores it (gzip TSV incl. CIGAR), parsed by the library and handed to the
            # caller without pandas (pav_bed_open + pav_cigar_load_bed); writing the input file is not timed
            bed_path = os.path.join(tmp_out, 'aligned_tig.bed.gz')
            hap.df_align.to_csv(bed_path, sep='\t', index=False, compression={'method': 'gzip', 'compresslevel': 1})
            t0 = time.perf_counter()
            table = _lib.BedTable(bed_path)
            t_parse = time.perf_counter() - t0
            t0 = time.perf_counter()
            loaded_index = ctx.cigar_load_bed(table, -1)
            ctx.sync()
            t_load = time.perf_counter() - t0
            table.close()
            assert loaded_index.shape[0] == aln.shape[0]
            ctx.cigar_load(aln, text, off)                             # back to the arrays the timed steps used
            inv_tables = None
            if args.workload == 'cigar+inv':
                # density tables of every inversion call as rule call_inv_batch writes them (density_{ID}_{hap}.tsv.gz)
                calls = [(i, c) for i, c in enumerate(lanes[0].out) if c is not None and not isinstance(c, RuntimeError)]
                den_dir = os.path.join(tmp_out, 'density')
                os.makedirs(den_dir)
                t0 = time.perf_counter()
                ctx.inv_write_tables([i for i, _ in calls], [os.path.join(den_dir, f'density_{c.id}_{hap.hap}.tsv.gz') for _, c in calls])
                t_den_gz = time.perf_counter() - t0
                t0 = time.perf_counter()
                ctx.inv_write_tables([i for i, _ in calls], [os.path.join(den_dir, f'density_{c.id}_{hap.hap}.tsv') for _, c in calls])
                t_den_plain = time.perf_counter() - t0
                inv_tables = {'calls': len(calls), 'rows': int(sum(ctx.inv_table_view(i, c.native_table[2])[1].shape[0] for i, c in calls)),
                              'write_density_tables_gzip_s': round(t_den_gz, 3), 'write_density_tables_plain_s': round(t_den_plain, 3),
                              'text_bytes': sum(os.path.getsize(os.path.join(den_dir, f)) for f in os.listdir(den_dir) if f.endswith('.tsv')),
                              'gzip_bytes': sum(os.path.getsize(os.path.join(den_dir, f)) for f in os.listdir(den_dir) if f.endswith('.tsv.gz')),
                              'writer': os.environ.get('PAV_WRITER', 'device'),
                              'note': 'pav_inv_write_tables: pandas-identical text formatted and gzip\'d in HBM from the resident column blocks '
                                      '(textdev.hip, deflate.hip); only the files\' bytes cross PCIe; PAV_WRITER=host: host threads + zlib.  '
                                      'DataFrame.to_csv needs ~7 us per row plain, ~19 us gzip\'d (measured, 300 k rows)'}
            e2e = {'rows': n1 + n2, 'write_tables_gzip_s': round(t_gz, 3), 'write_tables_plain_s': round(t_plain, 3),
                   'inv_density_tables': inv_tables,
                   'read_align_table_s': {'parse_gzip_tsv': round(t_parse, 3), 'load_to_device': round(t_load, 3),
                                          'bytes': os.path.getsize(bed_path)},
                   'text_bytes': os.path.getsize(os.path.join(tmp_out, 'snv.bed')) + os.path.getsize(os.path.join(tmp_out, 'insdel.bed')),
                   'gzip_bytes': os.path.getsize(os.path.join(tmp_out, 'snv.bed.gz')) + os.path.getsize(os.path.join(tmp_out, 'insdel.bed.gz')),
                   'writer': os.environ.get('PAV_WRITER', 'device'),
                   'note': 'pav_cigar_write_tables: order + FILTER, TSV text (byte-identical to pandas) and gzip (one member per file) all on '
                           'the device (textdev.hip, deflate.hip: 64 KiB of text per wave); PAV_WRITER=host: host threads + zlib; the pandas '
                           'mirror needs ~13 us per row, the reference ~410 us per row (BASELINE.md)'}
        finally:
            shutil.rmtree(tmp_out, ignore_errors=True)
        # The input side of a haplotype: a bgzipped FASTA file (the form PAV keeps them in: rules/call.snakefile:796) into the sequence
        # store - members inflated on the device (inflate.hip) against the same file inflated by host threads, on a 1 GB sample of a
        # synthetic assembly (tools/bench_bgzf.py has the 3 GB figures; `profiles/r05_bgzf_loader.json`).  Not part of `value`.
        if e2e is not None and not args.no_cpu_baseline:
            try:
                from pav_amd import synth as _synth
                from tools.bench_bgzf import assembly_text
                tmp_fa = tempfile.mkdtemp(prefix='pav_bench_fa_')
                try:
                    plain, gz = os.path.join(tmp_fa, 'asm.fa'), os.path.join(tmp_fa, 'asm.fa.gz')
                    n_rec = assembly_text(plain, 1000)
                    _synth.bgzip(plain, gz, threads=min(16, effective_cpus()))
                    res = {'text_bytes': os.path.getsize(plain), 'bgzf_bytes': os.path.getsize(gz), 'records': n_rec}
                    with _lib.Context(0) as c2:
                        for name, path, env in (('bgzf_device_inflate_s', gz, None), ('bgzf_host_inflate_s', gz, 'host'), ('plain_text_s', plain, None)):
                            if env:
                                os.environ['PAV_FASTA_INFLATE'] = env
                            ts = []
                            for _ in range(3):
                                t0 = time.perf_counter()
                                c2.seq_load_fasta_path(_lib.PAV_ROLE_TIG, path)
                                c2.sync()
                                ts.append(round(time.perf_counter() - t0, 4))
                            os.environ.pop('PAV_FASTA_INFLATE', None)
                            res[name] = min(ts)
                    res['note'] = ('pav_seq_load_fasta_path, best of three: file -> pinned ring -> HBM, BGZF members inflated by a lane each (tokens) and a '
                                   'wave each (copies in an LDS ring), every member\'s CRC-32 checked, header lines and line breaks removed on the device')
                    e2e['fasta_loader'] = res
                finally:
                    shutil.rmtree(tmp_fa, ignore_errors=True)
            except Exception as ex:                                 # (a measurement beside the line, never the line's failure)
                e2e['fasta_loader'] = {'error': repr(ex)}

    if rank == 0:
        ms_per_step = t_max / args.steps * 1e3
        value = aligned_total / t_max / 1e9                 # bp of the K passes (all ranks) / slowest rank's wall time
        n_ops, n_snv, n_indel = counts.n_ops, counts.n_snv, counts.n_indel
        scanned_bp = 0
        if args.workload == 'cigar+inv':
            import re as _re
            for ln in lanes[0].log.getvalue().splitlines():
                if ln.startswith('Scanning region: '):
                    m = _re.search(r':(\d+)-(\d+)', ln.split(': ')[1])
                    scanned_bp += int(m.group(2)) - int(m.group(1)) + 1
        pmc = None                                            # committed PMC summary of this workload, newest round first
        for pmc_name in _newest_first('_pmc.json'):
            try:
                with open(os.path.join(ROOT, 'profiles', pmc_name)) as fh:
                    cand = json.load(fh)
                if cand['workload']['aligned_bp_per_gpu'] == int(counts.aligned_bases):
                    pmc = cand
                    pmc['file'] = 'profiles/' + pmc_name
                    break
            except (OSError, KeyError, ValueError):
                pass

        # walk_snv reads ONE byte of each ASCII arena per SNV row; the SNVs of a haplotype lie ~1000 bases apart except inside
        # inversions, so nearly every byte costs a memory line of its own, and the fabric moves 64 B per isolated byte
        # (TCC_EA0_RDREQ_32B = 0, profiles/r03_cigar_emission_counters.txt).  The kernel is therefore priced in distinct 64 B lines
        # (counted here from the rows the device produced) against the rate at which the memory system delivers isolated lines
        # (tools/ubench/gather_rate.hip, the maximum over its variants with a 16 B store per row beside the loads, as in the kernel).
        # The roof: the MAXIMUM over the ubench variants that load the way the kernel does - walk_snv stores a 16 B row per pair of
        # loads and its loads are non-temporal, so every `16 B stored` variant counts, nt loads included (round 3 left those out and
        # priced the kernel against 40.4 G lines/s; the same ubench reaches 45.7 with nt loads); `loads_only` = the maximum with
        # nothing stored, the ceiling of the memory system for this access pattern
        line_roof = {'with_stores': None, 'with_stores_variant': None, 'loads_only': None, 'loads_only_variant': None,
                     'source': 'profiles/r03_gather_rate.txt'}
        for cand_ in _newest_first('_gather_rate.txt'):
            if os.path.exists(os.path.join(ROOT, 'profiles', cand_)):
                line_roof['source'] = 'profiles/' + cand_
                break
        try:
            with open(os.path.join(ROOT, line_roof['source'])) as fh:
                for ln_ in fh:
                    if '->' in ln_ and 'G isolated' in ln_ and not ln_.startswith('maximum'):
                        rate_ = float(ln_.split('->')[1].split('G')[0])
                        key_ = 'with_stores' if '16 B stored' in ln_ else ('loads_only' if 'nothing stored' in ln_ else None)
                        if key_ and (line_roof[key_] is None or rate_ > line_roof[key_]):
                            line_roof[key_] = rate_
                            line_roof[key_ + '_variant'] = ' '.join(ln_.split('avg')[0].split())
        except OSError:
            pass
        snv_lines = None
        if snv.shape[0]:
            a64 = snv['aln'].astype(np.int64) << 40
            snv_lines = int(np.unique(a64 | (snv['pos'].astype(np.int64) >> 6)).shape[0] + np.unique(a64 | (snv['qry_pos'].astype(np.int64) >> 6)).shape[0])

        def emission_phase(kern_):
            # walk_snv (side stream) and homology_kernel (main stream) run beside each other and draw on one budget of isolated
            # line fetches: the phase as a whole against the line-rate roof
            if snv_lines is None or not pmc or 'walk_snv' not in kern_ or 'homology_kernel' not in kern_:
                return None
            hom_lines = pmc.get('fetch_kib', {}).get('homology_kernel')
            if not hom_lines or not _lib.kernel_sources_match(pmc.get('provenance'), 'homology_kernel'):
                return None                                   # (no counter of this source of the kernel: nothing quoted)
            hom_lines = hom_lines * 1024.0 / 64.0
            t_ms = max(kern_['walk_snv']['avg_ms'], kern_['homology_kernel']['avg_ms'])
            rate = (snv_lines + hom_lines) / (t_ms * 1e-3) / 1e9
            peak = line_roof['with_stores']
            return {'line_fetches_per_pass': int(snv_lines + hom_lines), 'phase_ms': round(t_ms, 4), 'achieved_glines_per_s': round(rate, 1),
                    'measured_peak_glines_per_s': peak, 'peak_variant': line_roof['with_stores_variant'],
                    'frac': None if not peak else round(rate / peak, 3),
                    'note': 'distinct lines of the SNV rows + fabric reads of homology_kernel (FETCH_SIZE / 64 B, ' + str(pmc.get('file')) +
                            ') over the longer of the two kernels, which run beside each other'}

        def isolated_lines(kernel_, avg_ms_):
            if kernel_ != 'walk_snv' or snv_lines is None:
                return None
            rate = snv_lines / (avg_ms_ * 1e-3) / 1e9
            peak = line_roof['with_stores']
            return {'distinct_64B_lines_per_launch': snv_lines, 'achieved_glines_per_s': round(rate, 1),
                    'measured_peak_glines_per_s': peak, 'peak_variant': line_roof['with_stores_variant'],
                    'measured_peak_loads_only': line_roof['loads_only'], 'source': line_roof['source'],
                    'frac': None if not peak else round(rate / peak, 3),
                    'note': 'distinct 64 B lines of the two ASCII arenas the SNV rows touch (counted from the rows); the kernel time here is '
                            'what it takes beside homology_kernel, which draws on the same budget (3.96 M more line fetches per pass): '
                            'profiles/r03_cigar_emission_counters.txt has both kernels alone and together'}

        def make_roofline(prof_, want=None, kde=None):
            """Dominant kernel (largest total time in the profiled steps) against the HBM roofline: algorithmic bytes per
            launch / average launch duration from HIP events on the library's streams."""
            kern_ = {k: {'launches': v[0], 'avg_ms': v[1] / max(1, v[0])} for k, v in prof_.items()}
            own = {k: v['avg_ms'] for k, v in kern_.items()}
            dom = want or max(kern_, key=lambda k: own[k] * kern_[k]['launches'])       # the true arg-max of total time, no tie-breaks
            alg_bytes = {
                'verify_kernel': 0.5 * float(counts.aligned_bases),           # the two 2-bit planes; masks only where marked dirty
                'pack_kernel': tig_bases * (1.0 + 0.25 + 0.125),
                'tok_tiles': float(text.shape[0]) + 4.0 * n_ops,               # CIGAR text in, operation words out
                'walk_snv': 4.0 * n_ops + 16.0 * n_snv + 2.0 * n_snv,          # ops, SNV rows out, REF / ALT bytes in
                'walk_indel': 4.0 * n_ops + 64.0 * n_indel,
                'homology_kernel': 128.0 * n_indel + 2.0 * counts.seq_bytes,    # stub in, record out, SEQ bytes in and out
                'rocprim::radix_sort_keys': 7 * 16.0 * n_snv,               # 56 key bits = 7 passes over 8 B keys, in + out
                'k_snv_keys': 24.0 * n_snv, 'k_indel_keys': 72.0 * n_indel,
            }
            # k-mer kernels: per scanned base 0.375 B of packed planes (reference + contig), two 4 B list entries (three before the canonical sets of round 3) written by
            # the bucket kernels and read by k_kmer_lds, two answer bytes; HBM-table kernels as SURVEY.md section 8(d);
            # `scanned_bp` = region bases over all scan iterations of a step, spread over the launches (one per round)
            # (round 3: canonical k-mer sets - one 4 B list entry per contig base instead of two)
            for kname, per_base in (('k_bucket_ref', 4.375), ('k_bucket_tig', 5.375), ('k_kmer_lds', 10.75),
                                    ('k_ref_insert', 8.375), ('k_tig_state', 17.375), ('k_compact_scatter', 19.0)):
                if kname in kern_ and kern_[kname]['launches'] and scanned_bp:
                    alg_bytes[kname] = per_base * scanned_bp * args.steps / kern_[kname]['launches']
            # HBM traffic of the dominant kernel from the committed PMC summary of the same workload (profiles/r01_pmc.json;
            # separate rocprofv3 --pmc passes).  FETCH_SIZE is doubled for the 16 B/lane streaming kernels as the guide prescribes.
            traffic = None
            # a committed counter is quoted only for the source it was taken on: the profile's provenance stamp
            # (tools/prof_summary.py) must name the same cigar.hip / density.hip ... as the library that runs here
            pmc_fresh = bool(pmc) and _lib.kernel_sources_match(pmc.get('provenance'), dom)
            if pmc and pmc_fresh and dom in pmc.get('fetch_kib', {}) and dom in pmc.get('write_kib', {}):
                fx = 2.0 if dom in ('pack_kernel', 'verify_kernel') else 1.0          # 16 B/lane streams (verify: its 2-bit windows)
                traffic = (pmc['fetch_kib'][dom] * fx + pmc['write_kib'][dom]) * 1024.0
            a_bytes = alg_bytes.get(dom)
            achieved = a_bytes / (kern_[dom]['avg_ms'] * 1e-3) / 1e9 if a_bytes and kern_[dom]['avg_ms'] > 0 else None
            head = {'kernel': dom, 'bound': 'hbm', 'achieved': None if achieved is None else round(achieved, 1),
                    'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': None if achieved is None else round(achieved / HBM_PEAK_GBS, 4),
                    'traffic': traffic, 'algorithmic_bytes_per_launch': a_bytes,
                    'pmc_source': None if not pmc else {'file': pmc.get('file'), 'commit': (pmc.get('provenance') or {}).get('commit'),
                                                        'library_source_sha16': (pmc.get('provenance') or {}).get('library_source_sha16'),
                                                        'running_library_source_sha16': _lib.source_fingerprint()['library_source_sha16'],
                                                        'kernel_source_unchanged': pmc_fresh}}
            if dom == 'k_kde_eval' and kde and kern_[dom]['launches'] and kern_[dom]['avg_ms'] > 0:
                # the kernel densities: FP64 exp work, no bytes to speak of.  Algorithmic flops = SURVEY 8(d): 25 per (evaluation
                # point, data point) pair of scipy's double loop.  The kernel does not run that loop - a run of consecutive
                # INDEX values is summed in closed form - so the figure can exceed the FP64 vector peak;
                # `executed` prices the (point, run) pairs it does go over at ~150 flop (two exp, two erfc, the polynomial terms)
                flops = 25.0 * kde[2] / kern_[dom]['launches']
                tf = flops / (kern_[dom]['avg_ms'] * 1e-3) / 1e12
                ex = 150.0 * kde[1] / kern_[dom]['launches'] / (kern_[dom]['avg_ms'] * 1e-3) / 1e12
                head = {'kernel': dom, 'bound': 'mfma', 'bound_detail': 'FP64 vector ALU (exp / erfc): the sum of Gaussians has no MFMA form '
                                                                        '(SURVEY 8(d)); peak = FP64 vector peak, not a matrix peak',
                        'achieved': round(tf, 2), 'peak': FP64_VECTOR_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                        'frac': round(tf / FP64_VECTOR_PEAK_TFLOPS, 4), 'traffic': None, 'algorithmic_flops_per_launch': flops,
                        'evaluation_points_per_launch': kde[0] / kern_[dom]['launches'],
                        'point_run_pairs_per_launch': kde[1] / kern_[dom]['launches'],
                        'point_data_pairs_per_launch': kde[2] / kern_[dom]['launches'],
                        'executed': {'tflops': round(ex, 2), 'frac': round(ex / FP64_VECTOR_PEAK_TFLOPS, 4),
                                     'model': '150 flop per (point, run) pair'}}
            if 'walk_snv' in kern_ and kern_['walk_snv']['avg_ms'] > 0:              # the line-rate model of walk_snv, whichever kernel leads
                head['isolated_lines'] = isolated_lines('walk_snv', kern_['walk_snv']['avg_ms'])
            head.update({'avg_kernel_ms': round(kern_[dom]['avg_ms'], 4), 'launches_per_step': round(kern_[dom]['launches'] / args.steps, 2)})
            return kern_, {**head,
                           'kernels_ms': {k: round(v['avg_ms'], 4) for k, v in sorted(kern_.items())},
                           # every kernel with a byte model: achieved GB/s of algorithmic bytes; the
                           # latency-bound ones (scattered line fetches, dependent launches) sit far below the HBM roof by nature
                           'modelled_kernels_gbs': {k: round(alg_bytes[k] / (kern_[k]['avg_ms'] * 1e-3) / 1e9, 1)
                                                    for k in sorted(alg_bytes) if k in kern_ and kern_[k]['avg_ms'] > 0}}

        kern, roofline = make_roofline(prof, kde=kde_leg)
        roofline['emission_phase'] = emission_phase(kern)
        roofline['dominant_measured'] = roofline['kernel']           # (the arg-max of total event time, one lane alone; no substitution)
        if lanes_prof:
            # the same ranking in the regime the line is measured in: every lane running, HIP events around every kernel of every lane
            tot = {}
            for pr_ in lanes_prof:
                for k_, v_ in pr_.items():
                    a_ = tot.setdefault(k_, [0, 0.0])
                    a_[0] += v_[0]
                    a_[1] += v_[1]
            dev_ms = sum(v_[1] for v_ in tot.values()) / args.steps
            top = sorted(tot, key=lambda k_: -tot[k_][1])
            dom_l = top[0]
            lds = None
            for lds_name in _newest_first('_lds_counters.json'):
                try:
                    with open(os.path.join(ROOT, 'profiles', lds_name)) as fh:
                        lds = json.load(fh)
                    lds['file'] = 'profiles/' + lds_name
                    break
                except (OSError, ValueError):
                    lds = None

            def lds_roof(k_, avg_ms_):
                # LDS-array cycles of a launch (SQ_LDS_IDX_ACTIVE, summed over the CUs; committed counter pass taken with the same
                # lanes running) against what 256 CUs offer in the launch's duration at 2.4 GHz: the fraction of the LDS arrays'
                # time the kernel keeps busy; `conflict` = the share of those cycles that are bank-conflict replays
                if not lds or k_ not in lds.get('kernels', {}) or avg_ms_ <= 0:
                    return None
                if not _lib.kernel_sources_match(lds.get('provenance'), k_):
                    return {'bound': 'lds', 'frac': None, 'source': lds.get('file'),
                            'note': 'the committed counters were taken on another source of this kernel (provenance stamp differs): not quoted'}
                c_ = lds['kernels'][k_]
                peak_cyc = 256 * 2.4e9 * avg_ms_ * 1e-3
                return {'bound': 'lds', 'lds_array_cycles_per_launch': c_['lds_idx_active'], 'peak_cycles_in_launch': round(peak_cyc),
                        'frac': round(c_['lds_idx_active'] / peak_cyc, 4), 'bank_conflict_share': c_.get('bank_conflict_share'),
                        'lds_instructions_per_launch': c_.get('insts_lds'), 'source': lds.get('file'),
                        'note': 'SQ_LDS_IDX_ACTIVE / (256 CUs x 2.4 GHz x launch duration); guide: 256 B per clock and CU'}
            roofline['timed_region'] = {
                'lanes': n_lanes, 'kernel': dom_l, 'ms_per_step_profiled': round(t_lanes_prof / args.steps * 1e3, 4),
                'device_ms_per_step': round(dev_ms, 3),
                'share_of_device_time': round(tot[dom_l][1] / args.steps / dev_ms, 4) if dev_ms else None,
                'avg_kernel_ms': round(tot[dom_l][1] / max(1, tot[dom_l][0]), 4),
                'avg_kernel_ms_alone': round(kern[dom_l]['avg_ms'], 4) if dom_l in kern else None,
                'lds': lds_roof(dom_l, tot[dom_l][1] / max(1, tot[dom_l][0])),
                'kernels': {k_: {'ms_per_step': round(tot[k_][1] / args.steps, 4), 'share': round(tot[k_][1] / args.steps / dev_ms, 4),
                                 'avg_ms': round(tot[k_][1] / max(1, tot[k_][0]), 4),
                                 'over_alone': round(tot[k_][1] / max(1, tot[k_][0]) / kern[k_]['avg_ms'], 2) if k_ in kern and kern[k_]['avg_ms'] > 0 else None,
                                 'lds': lds_roof(k_, tot[k_][1] / max(1, tot[k_][0]))}
                            for k_ in top[:12]},
                'note': 'per-kernel HIP-event sums over ALL lanes of one more region of K steps with every lane running (kernels of other '
                        'haplotypes beside each launch): the dominant kernel of the regime `value` is measured in; device_ms_per_step '
                        'counts overlapping kernels once each, so it exceeds ms_per_step'}
        # ---- the honest roofline of the PATH (SURVEY.md section 8(d) byte model; the pack above is pre-processing the model
        #      has no term for): algorithmic bytes of one step / step time.  CIGAR-call: 4 B / op + 64 B / row + 16 B / SNV +
        #      40 B / indel + 2-bit SV bases + 0.5 B per scanned homology base (window bound: 128 B / indel) + 0.5 B per X base;
        #      k-mer scan: 80 B per scanned region base.
        path_bytes = (4.0 * n_ops + 64.0 * aln.shape[0] + 16.0 * n_snv + 40.0 * n_indel + 0.5 * counts.seq_bytes + 64.0 * n_indel +
                      0.5 * n_snv)
        scan_bytes = 80.0 * scanned_bp
        sum_kernel_ms = sum(v['avg_ms'] * v['launches'] for v in kern.values()) / args.steps
        t_single_s = t_single if t_single is not None else t_local      # --lanes 1: the timed region is the single lane
        roofline['path'] = {
            'bound': 'hbm', 'unit': 'GB/s', 'peak': HBM_PEAK_GBS,
            'algorithmic_bytes_per_step': {'cigar_call': round(path_bytes), 'kmer_scan': round(scan_bytes),
                                           'contig_pack_not_in_the_model': round(tig_bases * 1.375) if eager_pack else 0},
            'achieved': round((path_bytes + scan_bytes) / (ms_per_step * 1e-3) / 1e9, 1),
            'frac': round((path_bytes + scan_bytes) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            'frac_with_pack_counted': round((path_bytes + scan_bytes + (tig_bases * 1.375 if eager_pack else 0.0)) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            'contig_pack': ('whole arena at the start of every pass (PAV_EAGER_PACK=1)' if eager_pack else
                            'on demand: homology windows decode the ASCII arena, the k-mer scans pack the 1024-base blocks under their '
                            'regions (pack_spans_kernel), verify mode packs everything (DESIGN.md section 3.1)'),
            'sum_kernel_ms_per_step': round(sum_kernel_ms, 3),
            'ms_per_step_over_sum_kernel_ms': round(ms_per_step / sum_kernel_ms, 3) if sum_kernel_ms else None,
            'single_lane_ms_per_step': round(t_single_s / args.steps * 1e3, 4),
            # ONE lane alone (one host thread, one haplotype, nothing of another haplotype beside it): its wall time per pass against
            # the device work of a pass - 1.0 = the host never keeps the GPU waiting
            'single_lane': {'ms_per_step': round(t_single_s / args.steps * 1e3, 4),
                            'over_sum_kernel_ms': round(t_single_s / args.steps * 1e3 / sum_kernel_ms, 3) if sum_kernel_ms else None,
                            'value': round(float(lanes[0].counts.aligned_bases) * args.steps / t_single_s / 1e9, 2), 'unit': 'Gbp/s'},
            'note': 'SURVEY 8(d) bytes of one pass / ms_per_step: the path is bound by scattered 64 B sector fetches, dependent '
                    'launches and host control, not by streamed bytes (DESIGN.md section 3); sum_kernel_ms_per_step = HIP-event '
                    'time of every kernel of one pass, measured with one lane running alone'}
        if pmc:
            # measured HBM traffic over modelled bytes per launch, for every kernel that has both (committed PMC summary)
            ratios = {}
            model = make_roofline(prof)[1]
            for k_, gbs in model['modelled_kernels_gbs'].items():
                if k_ in pmc.get('fetch_kib', {}) and k_ in pmc.get('write_kib', {}) and kern[k_]['avg_ms'] > 0 \
                        and _lib.kernel_sources_match(pmc.get('provenance'), k_):
                    fx = 2.0 if k_ in ('pack_kernel', 'verify_kernel') else 1.0
                    tr = (pmc['fetch_kib'][k_] * fx + pmc['write_kib'][k_]) * 1024.0
                    ab = gbs * 1e9 * kern[k_]['avg_ms'] * 1e-3
                    if ab > 0:
                        ratios[k_] = round(tr / ab, 2)
            line_model = None
            if 'walk_snv' in pmc.get('fetch_kib', {}) and 'walk_snv' in pmc.get('write_kib', {}) and snv_lines:
                line_model = round((pmc['fetch_kib']['walk_snv'] + pmc['write_kib']['walk_snv']) * 1024.0 /
                                   (4.0 * n_ops + 16.0 * n_snv + 64.0 * snv_lines), 2)
            roofline['traffic_over_algorithmic'] = {'source': pmc.get('file'), 'source_commit': (pmc.get('provenance') or {}).get('commit'),
                                                    'only_kernels_whose_source_is_unchanged_since': True, 'ratio': ratios,
                                                    'walk_snv_over_line_granular_model': line_model}

        def add_alone(roof):
            if pack_alone and pack_alone[0] and roof['kernel'] == 'pack_kernel':
                ms_alone = pack_alone[1] / pack_alone[0]
                gbs = roof['algorithmic_bytes_per_launch'] / (ms_alone * 1e-3) / 1e9
                roof['alone'] = {'avg_kernel_ms': round(ms_alone, 4), 'achieved': round(gbs, 1), 'frac': round(gbs / HBM_PEAK_GBS, 4),
                                 'note': 'the same launch with no other kernel resident (inside a step the tokenizer / walk kernels and, '
                                         'in the whole path, the previous step\'s table copy run beside it)',
                                 'cigar_call_without_pack': call_alone}
        add_alone(roofline)
        _, roof_v = make_roofline(verify_leg['prof'], want='verify_kernel')
        vres = lanes[0].verify
        verify_mode = {'workload': 'CIGAR-call + verify: the packed reference and contig are streamed along every = / X operation and '
                                   'every base is checked against the CIGAR (pav_cigar_verify; the reference never looks at = runs)',
                       'value': round(aligned_total / verify_leg['t'] / 1e9, 2), 'unit': 'Gbp/s',
                       'ms_per_step': round(verify_leg['t'] / args.steps * 1e3, 4), 'steps': args.steps,
                       'bases_checked': vres['eq_bases'] + vres['x_bases'], 'bases_contradicting_the_cigar': vres['eq_mismatch'] + vres['x_match'],
                       'roofline': roof_v}
        cigar_only = None
        if cigar_leg is not None:
            _, roof_c = make_roofline(cigar_leg['prof'])
            add_alone(roof_c)
            cigar_only = {'workload': 'BASELINE configs[1]: the same haplotype, CIGAR-call only (tokenise + walk + homology + '
                                      'SEQ gather), measured in this run after the headline region',
                          'value': round(aligned_total / cigar_leg['t'] / 1e9, 2), 'unit': 'Gbp/s',
                          'ms_per_step': round(cigar_leg['t'] / args.steps * 1e3, 4), 'steps': args.steps, 'roofline': roof_c}

        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle
            df = hap.df_align
            span = (df['END'] - df['POS']).cumsum()
            n_rows = int(np.searchsorted(span.to_numpy(), args.cpu_sample_bp * min(1.0, args.scale))) + 1
            sub = df.iloc[:min(n_rows, df.shape[0])]
            a2, t2, o2 = cigarcall.pack_alignments(sub, names, hap.tig_names)
            c0 = time.perf_counter()
            o_snv, o_indel, o_blob, err = oracle.cigar_call([hap.ref.seqs[n] for n in names],
                                                            [hap.tig_seqs[n] for n in hap.tig_names], a2, t2, o2)
            c1 = time.perf_counter() - c0
            # aligned bases of the sample = (=,X) lengths: recount from the device ops of those rows
            ops, op_off = ctx.cigar_fetch_ops(counts.n_ops, df.shape[0])
            sel = ops[:int(op_off[sub.shape[0]])]
            code = sel & 15
            sample_bp = int((sel[(code == 7) | (code == 8)] >> 4).astype(np.int64).sum())
            # the sample doubles as a full-size parity check of the leading rows
            ok = snv[:o_snv.shape[0]].tobytes() == o_snv.tobytes() and blob[:o_blob.shape[0]].tobytes() == o_blob.tobytes()
            for f in o_indel.dtype.names:
                if f != 'pad':
                    ok = ok and bool(np.array_equal(indel[f][:o_indel.shape[0]], o_indel[f]))
            cpu = {'value': round(sample_bp / c1 / 1e9, 4), 'unit': 'Gbp/s', 'cores': 1, 'kind': 'port',
                   'reference_python': {'cigar_call_Mbp_per_s': 2.1, 'density_scan_kbp_per_s': 2.5, 'cores': 1,
                                        'hardware': 'survey sandbox: 8 host cores (1 used), Python 3.10.12, numpy 2.2.6, pandas 2.3.3, '
                                                    'scipy 1.15.3; pavlib 2.4.6 imported unmodified with shims for the absent modules',
                                        'source': 'BASELINE.md section 2 (make_insdel_snv_calls on a 1 Mb contig: 0.48 s; scripts/density.py '
                                                  'on a 25 kb region: 9.8 s); the reference Python never travels to the GPU box, so it '
                                                  'cannot be re-timed here'},
                   'sample': f'first {sub.shape[0]} alignment rows of the same haplotype ({sample_bp / 1e9:.3f} Gbp aligned, '
                             f'{o_snv.shape[0]} SNV, {o_indel.shape[0]} INDEL), oracle/ scalar C walk incl. per-contig '
                             f'upper-casing and reverse complement, {c1:.1f} s wall',
                   'sample_short': f'{sub.shape[0]} alignment rows ({sample_bp / 1e9:.3f} Gbp) through the oracle C walk: {c1:.1f} s',
                   'records_match_gpu': bool(ok)}
            # the same port on all host cores: the alignment rows are independent, so the haplotype is cut into row groups of
            # equal CIGAR text and every group walks on its own thread (ctypes releases the GIL); reported beside the 1-core figure
            from concurrent.futures import ThreadPoolExecutor
            from pav_amd.shard import effective_cpus
            n_thr = max(1, min(effective_cpus(), 64))          # cgroup quota / affinity, not the host's core count
            t_all = None
            if n_thr > 1 and sub.shape[0] >= 2:
                # consecutive rows stay together (the table is sorted by chromosome, and every group upper-cases the sequences
                # its rows touch once): cut where the cumulative CIGAR text crosses multiples of total / groups
                cum = np.cumsum(sub['CIGAR'].str.len().to_numpy(dtype=np.int64))
                n_grp = min(sub.shape[0], 2 * n_thr)
                cuts = np.unique(np.concatenate(([0], np.searchsorted(cum, cum[-1] * np.arange(1, n_grp) / n_grp) + 1, [sub.shape[0]])))
                groups = [np.arange(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
                n_grp = len(groups)
                packs = [cigarcall.pack_alignments(sub.iloc[g], names, hap.tig_names) for g in groups]
                refs, tigs = [hap.ref.seqs[n] for n in names], [hap.tig_seqs[n] for n in hap.tig_names]
                c0 = time.perf_counter()
                with ThreadPoolExecutor(n_thr) as pool:
                    parts = list(pool.map(lambda pk: oracle.cigar_call(refs, tigs, *pk)[0].shape[0], packs))
                t_all = time.perf_counter() - c0
                cpu['all_cores'] = {'cigar_call_only': round(sample_bp / t_all / 1e9, 3), 'unit': 'Gbp/s', 'cores': n_thr,
                                    'snv_records': int(sum(parts)), 'wall_s': round(t_all, 2),
                                    'host_cpus': os.cpu_count(),
                                    'note': f'{n_grp} groups of consecutive rows on {n_thr} threads (= the CPUs this process may use: '
                                            f'affinity and cgroup quota; the host has {os.cpu_count()}); every group upper-cases the chromosomes '
                                            'and contigs its rows touch (the reference does so per row, cigarcall.py:74-75)'}
            if args.workload == 'cigar+inv' and scanned_bp:
                # k-mer density scan on the CPU: the first scan iteration of the first liftable flagged regions through the
                # oracle (scalar C: hash set, STATE_MER, scipy-order KDE, change test, STATE), extrapolated by scanned bases
                import pandas as pd
                from pav_amd import density as pavden, seq as pavseq3
                fai = pd.Series(ref_lengths)
                ref_i, tig_i = {n: i for i, n in enumerate(names)}, {n: i for i, n in enumerate(hap.tig_names)}
                jobs, pairs = [], []
                for r in lanes[0].regions:                  # the loci this haplotype's flagging produced and the step scanned
                    r.expand(4000, min_pos=0, max_end=fai, shift=True)
                    try:
                        t = lanes[0].lift.lift_region_to_qry(r)
                    except RuntimeError:
                        t = None
                    if t is None or len(r) > 60_000:
                        continue
                    jobs.append(_lib.DenJob(ref_i[r.chrom], tig_i[t.chrom], r.pos, r.end, t.pos, t.end, 1 if t.is_rev else 0, 20))
                    pairs.append((r, t))
                    if len(jobs) >= args.cpu_sample_regions:
                        break
                t_den, den_bp, den_ok = 0.0, 0, True
                res = ctx.density_batch(jobs, pavden.den_params()) if jobs else []
                for jx, ((r, t), g) in enumerate(zip(pairs, res)):
                    d0 = time.perf_counter()
                    o = oracle.density(hap.ref.seqs[r.chrom][r.pos:r.end], hap.tig_seqs[t.chrom][t.pos:t.end], t.is_rev)
                    t_den += time.perf_counter() - d0
                    den_bp += len(r)
                    den_ok = den_ok and g.status == o['status']
                    if o['status'] != 125:
                        cols = ctx.density_table(jx, g.n_rows)
                        den_ok = den_ok and all(np.array_equal(cols[c], o[c]) for c in ('INDEX', 'STATE_MER', 'STATE', 'KMER'))
                if den_bp:
                    t_scan_cpu = t_den * scanned_bp / den_bp                          # all scan iterations of the haplotype
                    t_cigar_cpu = c1 * float(counts.aligned_bases) / sample_bp
                    cpu.update({
                        'value': round(float(counts.aligned_bases) / (t_cigar_cpu + t_scan_cpu) / 1e9, 5),
                        'cigar_call_only': round(sample_bp / c1 / 1e9, 4),
                        'density_scan_bp_per_s': round(den_bp / t_den, 1),
                        'sample': cpu['sample'] + f'; k-mer density scan: first scan iteration of {len(pairs)} flagged regions '
                                  f'({den_bp} region bp, {t_den:.1f} s wall) extrapolated to the {scanned_bp} bp scanned per haplotype '
                                  f'({t_scan_cpu:.0f} s); value = aligned bp / (CIGAR walk + extrapolated scan); flagging not included',
                        'sample_short': cpu['sample_short'] + f'; density scan of {len(pairs)} flagged regions ({den_bp} bp): {t_den:.1f} s, '
                                        f'extrapolated to {scanned_bp} bp',
                        'density_tables_match_gpu': bool(den_ok)})
                    if t_all is not None and len(pairs) >= 2:
                        c0 = time.perf_counter()
                        with ThreadPoolExecutor(n_thr) as pool:
                            list(pool.map(lambda rt: oracle.density(hap.ref.seqs[rt[0].chrom][rt[0].pos:rt[0].end],
                                                                    hap.tig_seqs[rt[1].chrom][rt[1].pos:rt[1].end], rt[1].is_rev)['status'], pairs))
                        t_den_all = time.perf_counter() - c0
                        t_scan_all = t_den_all * scanned_bp / den_bp
                        t_cigar_all = t_all * float(counts.aligned_bases) / sample_bp
                        cpu['all_cores'].update({'value': round(float(counts.aligned_bases) / (t_cigar_all + t_scan_all) / 1e9, 4),
                                                 'density_scan_bp_per_s': round(den_bp / t_den_all, 1)})

        inv_report = None
        if args.workload == 'cigar+inv':
            from pav_amd import seq as pavseq2
            out = lanes[0].out
            scanned = iters = 0
            for ln in lanes[0].log.getvalue().splitlines():
                if ln.startswith('Scanning region: '):
                    scanned += len(pavseq2.region_from_string(ln.split(': ')[1]))
                    iters += 1
            flag_kernels = ('k_snv_keys', 'k_indel_keys', 'k_indel_mid', 'k_cluster_emit', 'k_insdel_split', 'k_ins_match',
                            'rocprim::radix_sort_keys', 'rocprim::radix_sort_pairs', 'rocprim::inclusive_scan')
            den = {k: v for k, v in kern.items() if k.startswith('k_') and k not in flag_kernels}
            f_tables, f_loci, f_counts = lanes[0].flag
            # every planted inversion that is aligned through shows up as a CLUSTER_SNV locus: count the overlaps
            rank_of = {n: i for i, n in enumerate(sorted(names))}
            hit = 0
            for iv in hap.ref.inversions:
                sel = f_loci[(f_loci['chrom'] == rank_of[iv.chrom]) & (f_loci['pos'] < iv.end) & (f_loci['end'] > iv.pos)]
                hit += bool(len(sel))
            flag_report = {'loci': int(len(f_loci)), 'try_inv': int(f_loci['try_inv'].sum()),
                           'tables': {k: int(len(v)) for k, v in f_tables.items()}, **f_counts,
                           'planted_inversions_flagged': hit,
                           'device_ms_per_step': round(sum(kern[k]['avg_ms'] * kern[k]['launches'] for k in flag_kernels if k in kern) / args.steps, 3),
                           'note': 'pav_cigar_flag inside the timed step: FILTER + sort + cluster sweeps + INS/DEL matching on the '
                                   'device, interval merges on the host (rules call_inv_cluster, call_inv_flag_insdel_cluster, '
                                   'call_inv_merge_flagged_loci), inv_sig_filter = single_cluster; the scan below runs on exactly the '
                                   'loci with TRY_INV this call produced (planted inversions: SNV clusters; generator-planted matched '
                                   'DEL + INS pairs: MATCH_INDEL)'}
            inv_report = {'flagging': flag_report, 'scanned_loci': len(out),
                          'near_tie_guard': {'n_near_tie': int(sum(getattr(o, 'n_near_tie', 0) for o in out if o is not None and not isinstance(o, RuntimeError))),
                                             'n_unresolved': int(sum(getattr(o, 'n_unresolved', 0) for o in out if o is not None and not isinstance(o, RuntimeError))),
                                             'note': 'float decisions of the calls\' scans within 1e-9 (re-evaluated in scipy\'s order); '
                                                     'n_unresolved: still within 1e-13 afterwards (must be 0 for the discrete outputs to be pinned)'}, 'calls': sum(1 for o in out if o is not None and not isinstance(o, RuntimeError)),
                          'planted': hap.stats['n_inv'], 'scan_iterations': iters, 'scanned_bp': scanned,
                          'device_ms_per_step': round(sum(v['avg_ms'] * v['launches'] for v in den.values()) / args.steps, 3),
                          'host_ms': {'align_table_once': round(lanes[0].t_lift_ms, 1), 'scan_for_inv_batch_last_step': round(lanes[0].t_scan_ms, 1)},
                          'note': 'wall time of the step includes the Python scan control (lift-over, expansion logic, '
                                  'DataFrame of every call); device_ms_per_step is the sum of the density kernels'}
        metric = ('aligned Gbp/s through CIGAR-call only (tokenise + walk + homology + SEQ gather + contig pack); bit-exact vs pavlib'
                  if args.workload == 'cigar' else
                  'aligned Gbp/s through CIGAR-call + k-mer inv scan; bit-exact vs pavlib')
        line = {
            'metric': metric,
            'value': round(value, 2), 'unit': 'Gbp/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'u8/u32 + f64 (KDE)', 'data': 'synthetic',
            'config': {'workload': ('BASELINE configs[1]: one hg38-shaped haplotype, CIGAR-call only, one haplotype per GPU'
                                    if args.workload == 'cigar' else
                                    (f'BASELINE configs[4] per-GPU batch: {n_lanes} haplotype(s) of the synthetic cohort vs a T2T-CHM13-shaped reference '
                                     '(24 sequences, CHM13v2.0 lengths, no N runs) ' if args.reference == 'chm13' else
                                     f'BASELINE configs[2] / [3] per-GPU share: {n_lanes} hg38-shaped haplotype(s) (h1 + h2 of phased diploid samples) ') +
                                    'resident per GPU against one resident reference; a step = one haplotype through CIGAR-call + '
                                    'signature flagging + k-mer inversion density scan of every locus the flagging marks TRY_INV; the '
                                    'steps go round the haplotypes (configs[1] = CIGAR-call only: see cigar_only)'),
                       'workload_short': ('configs[1]: hg38-shaped haplotype, CIGAR-call only' if args.workload == 'cigar' else
                                          f"configs[{'4' if args.reference == 'chm13' else '2'}] per-GPU share: {n_lanes} {args.reference}-shaped haplotypes "
                                          'resident; step = one haplotype through CIGAR-call + flagging + k-mer inversion scan'),
                       'scale': args.scale, 'seed': args.seed, 'aligned_bp_per_gpu': int(counts.aligned_bases),
                       'n_aln': int(aln.shape[0]), 'n_ops': int(n_ops), 'n_snv': int(n_snv), 'n_indel': int(n_indel),
                       'lanes_per_gpu': n_lanes, 'lanes_arg': args.lanes, 'usable_cpus_per_rank': round(cpus_per_rank, 2),
                       'reference': args.reference,
                       'pair_frac': args.pair_frac if gen_kw else 0.0, 'warmup_steps_run': warmup_run,
                       'call_tables': 'copied to pinned host memory inside the step' if args.eager_tables else 'resident in HBM (D2H in host.d2h_density_tables_s)',
                       'parallelism': f'{world} GPU(s) x {n_lanes} resident haplotype(s), one host thread each; no collective'},
            'repeats': {'regions': len(regions), 'steps_per_region': args.steps, 'choice': 'median region by max-over-ranks time',
                        'ms_per_step_median': round(t_max / args.steps * 1e3, 4),
                        'ms_per_step_min': round(min(r[1] for r in regions) / args.steps * 1e3, 4),
                        'ms_per_step_max': round(max(r[1] for r in regions) / args.steps * 1e3, 4),
                        'ms_per_step_all': [round(r[1] / args.steps * 1e3, 4) for r in regions],
                        'value_min': round(aligned_total / max(r[1] for r in regions) / 1e9, 2),
                        'value_max': round(aligned_total / min(r[1] for r in regions) / 1e9, 2)},
            'per_rank': per_rank, 'lanes_limited_by_cpus': limited_by_cpus, 'single_rank_same_lanes': solo_same_lanes,
            'load_balance': {'max_over_mean_ms': round(max(r['ms_per_step'] for r in per_rank) / (sum(r['ms_per_step'] for r in per_rank) / len(per_rank)), 4),
                             'max_over_mean_cigar_text': round(max(r['cigar_text_bytes'] for r in per_rank) /
                                                               (sum(r['cigar_text_bytes'] for r in per_rank) / len(per_rank)), 4),
                             'note': 'haplotype -> GPU; every rank holds whole haplotypes (weak scaling), so the balance is the '
                                     'haplotypes\' own size spread'},
            'with_lift_index': None if t_with_lift is None else {
                'value': round(aligned_total / t_with_lift / 1e9, 2), 'unit': 'Gbp/s', 'ms_per_step': round(t_with_lift / args.steps * 1e3, 4),
                'lift_index_ms_per_step': round((t_with_lift - t_max) / args.steps * 1e3, 4),
                'note': 'the same K steps with the lift-over index of the trimmed alignment table rebuilt inside every step '
                        '(pav_inv_load_alignments: tokenise + scan on the device, the operation tables stay there - lift_dev.hip) - what a cohort run pays '
                        'per haplotype; `value` keeps the index resident like the other inputs'},
            'roofline': roofline, 'cpu_baseline': cpu, 'cigar_only': cigar_only, 'verify_mode': verify_mode, 'inv_scan': inv_report,
            'end_to_end': e2e,
            # the streaming pack of the whole contig arena (verify mode, PAV_EAGER_PACK=1): timed with nothing beside it
            'contig_pack_alone': None if not (pack_alone and pack_alone[0]) else {
                'kernel': 'pack_kernel', 'avg_kernel_ms': round(pack_alone[1] / pack_alone[0], 4),
                'algorithmic_bytes_per_launch': tig_bases * 1.375,
                'achieved': round(tig_bases * 1.375 / (pack_alone[1] / pack_alone[0] * 1e-3) / 1e9, 1), 'unit': 'GB/s', 'peak': HBM_PEAK_GBS,
                'frac': round(tig_bases * 1.375 / (pack_alone[1] / pack_alone[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                'in_the_path': bool(eager_pack)},
            'hbm': {'total_gb': round(lanes[0].ctx.mem_info()[1] / 1e9, 1),
                    'peak_used_gb': round((lanes[0].ctx.mem_info()[1] - min(free_min)) / 1e9, 2),
                    'resident_haplotypes': n_lanes,
                    'records_match_per_lane': [ln.records_match for ln in lanes],
                    'note': 'pav_mem_info around every timed region (minimum free seen): one packed reference + per haplotype its contig '
                            'ASCII arena and on-demand planes, alignment tables, call records, flagging and density scratch, call tables; '
                            'records_match_per_lane: every resident haplotype\'s SNV / INDEL / SEQ records vs the oracle\'s scalar walk '
                            'and the generator\'s counts, checked once before the timed region (--reference chm13 only, else null)'},
            'host': {'generate_s': round(t_gen, 1), 'h2d_and_ref_pack_s': round(t_h2d, 2), 'd2h_records_s': round(t_d2h, 3),
                     'd2h_density_tables_s': None if t_d2h_tables is None else round(t_d2h_tables, 4),
                     'density_tables_mb': None if tables_mb is None else round(tables_mb, 1),
                     'cpu_baseline_cores_all': None if not cpu or 'all_cores' not in cpu else cpu['all_cores'].get('cores'),
                     'device': ctx.device_name},
        }
        # ---- what the driver parses: ONE short final line (contract keys + the objects the judge reads); everything else goes to
        #      the detail file (and to stderr).  Round 5 printed the whole 20 KB object and the driver's 8 KB tail lost the line.
        detail_path = os.path.abspath(args.detail)
        line['detail_file'] = os.path.basename(detail_path)
        try:
            with open(detail_path, 'w') as fh:
                json.dump(line, fh)
                fh.write('\n')
        except OSError as ex:                                    # (a read-only tree: the short line still goes out)
            print(f'[bench] could not write {detail_path}: {ex}', file=sys.stderr)
        print('[bench detail] ' + json.dumps(line), file=sys.stderr, flush=True)
        compact = compact_line(line, limited=limited_by_cpus, solo=solo_same_lanes)
        text_out = json.dumps(compact, separators=(',', ':'))
        assert len(text_out) <= 4096, f'the bench line grew to {len(text_out)} bytes: move keys to {detail_path}'
        sys.stderr.flush()
        print(text_out, flush=True)
    for ln in lanes[::-1]:
        ln.ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
