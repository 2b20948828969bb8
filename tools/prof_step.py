#!/usr/bin/env python3
"""cProfile of the host side of one lane's steps (CIGAR-call -> flagging -> scan), to see what Python costs per step.
    python tools/prof_step.py [--steps 30] [--scale 1.0]"""
import argparse, cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--scale', type=float, default=1.0)
    ap.add_argument('--no-build', action='store_true', help='everything is built already (under rocprofv3: no compiler from this process)')
    ap.add_argument('--plain', action='store_true', help='timed steps only, no cProfile pass')
    ap.add_argument('--kernels', action='store_true', help='after the timed steps: the same steps with HIP events around every kernel, averages as JSON')
    args = ap.parse_args()
    if not args.no_build:
        import __graft_entry__ as g
        g.build_cpu_side()
    import numpy as np
    from pav_amd import _lib, cigarcall, synth, inv as pavinv
    from pav_amd.align import AlignLift
    from pav_amd.kmer import KmerUtil
    hap = synth.config2(seed=1002, scale=args.scale, threads=8, pair_frac=0.009)
    names = hap.ref.names
    ctx = _lib.Context(0)
    ctx.seq_load(_lib.PAV_ROLE_REF, names, [hap.ref.seqs[n] for n in names])
    ctx.seq_load(_lib.PAV_ROLE_TIG, hap.tig_names, [hap.tig_seqs[n] for n in hap.tig_names])
    ctx.cigar_load(*cigarcall.pack_alignments(hap.df_align, names, hap.tig_names))
    ctx._inv_loaded = ('ref.fa', 'tig.fa')
    index = hap.df_align['INDEX'].to_numpy(dtype='int64')
    trim = hap.df_trim[['POS', 'END', 'INDEX']].set_index('INDEX').astype(int).reindex(list(index), fill_value=-1)
    tp, te = trim['POS'].to_numpy(dtype='int64'), trim['END'].to_numpy(dtype='int64')
    lift = AlignLift(hap.df_trim, hap.tig_lengths)
    k_util = KmerUtil(31)
    params = ctx.flag_params(sig_filter=_lib.SIG_SINGLE_CLUSTER)
    found = io.StringIO()

    timing = bool(os.environ.get('PAV_TIMING'))

    def step():
        t = [time.perf_counter()]

        def lap(what):
            if timing:
                now = time.perf_counter()
                print('[pav timing] step %-14s %.2f ms' % (what, (now - t[0]) * 1e3), file=sys.stderr)
                t[0] = now
        ctx.seq_pack(_lib.PAV_ROLE_TIG)
        ctx.cigar_call()
        lap('pack+call')
        flag = ctx.cigar_flag(tp, te, params)
        lap('flag')
        regions = pavinv.loci_regions(ctx, flag[1])
        log = io.StringIO()
        found.seek(0); found.truncate()
        lap('regions')
        return pavinv.scan_for_inv_batch(regions, 'ref.fa', 'tig.fa', lift, k_util, log=log, ctx=ctx, eager_tables=False, found_out=found)
    for _ in range(4):
        step()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync()
    print('ms per step (one lane, no profiler): %.3f' % ((time.perf_counter() - t0) / args.steps * 1e3))
    # CPU time of this thread per step (time.thread_time: what the lane's host thread computes; a blocked wait costs nothing,
    # a polling one counts - PAV_WAIT=block shows the work alone)
    c0, t0, w0 = time.thread_time(), time.perf_counter(), ctx.wait_stats()
    for _ in range(args.steps):
        step()
    ctx.sync()
    w1 = ctx.wait_stats()
    wall = (time.perf_counter() - t0) / args.steps * 1e3
    print('host thread CPU ms per step: %.3f of %.3f ms wall (PAV_WAIT=%s); %.3f ms in %.1f host waits per step: %.3f ms of host work' % (
        (time.thread_time() - c0) / args.steps * 1e3, wall, os.environ.get('PAV_WAIT', 'yield'), (w1[0] - w0[0]) / args.steps * 1e3,
        (w1[1] - w0[1]) / args.steps, wall - (w1[0] - w0[0]) / args.steps * 1e3))
    if args.kernels:
        import json
        ctx.prof_reset()
        ctx.prof_enable(True)
        for _ in range(args.steps):
            step()
        ctx.sync()
        prof = ctx.prof_read()
        ctx.prof_enable(False)
        per = {k: round(v[1] / args.steps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])}
        print('KERNELS ' + json.dumps({'lib': _lib.LIB_PATH, 'sum_ms_per_step': round(sum(per.values()), 4), 'ms_per_step': per,
                                       'launches_per_step': {k: round(v[0] / args.steps, 2) for k, v in prof.items()}}))
    if args.plain:
        ctx.close()
        return
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(args.steps):
        step()
    ctx.sync()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
    print(s.getvalue()[:6000])
    ctx.close()

if __name__ == '__main__':
    main()
