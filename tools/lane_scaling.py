#!/usr/bin/env python3
"""Where the multi-lane pass time comes from: L lanes (host threads, one context each, the reference shared) repeat ONE phase
of the pass - call / flag / scan - or all of them, for L = 1, 2, ...; prints ms per haplotype pass (wall / (steps x L)).
A phase whose figure stops falling with L has reached what the device gives it; the sum of the phases' floors is the floor
of bench.py's step.
    python tools/lane_scaling.py [--lanes 1,2,4,6,8] [--steps 24] [--phases call,flag,scan,all]"""
import argparse, io, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lanes', default='1,2,4,6,8')
    ap.add_argument('--steps', type=int, default=24)
    ap.add_argument('--scale', type=float, default=1.0)
    ap.add_argument('--phases', default='call,flag,scan,all')
    ap.add_argument('--no-build', action='store_true')
    ap.add_argument('--kernels', action='store_true', help='also the HIP-event time of every kernel per pass, per phase and lane count')
    args = ap.parse_args()
    if not args.no_build:
        import __graft_entry__ as g
        g.build_cpu_side()
    import numpy as np
    from pav_amd import _lib, cigarcall, synth, inv as pavinv
    from pav_amd.align import AlignLift
    from pav_amd.kmer import KmerUtil
    lane_counts = [int(x) for x in args.lanes.split(',')]
    hap = synth.config2(seed=1002, scale=args.scale, threads=8, pair_frac=0.009)
    names = hap.ref.names
    packed = cigarcall.pack_alignments(hap.df_align, names, hap.tig_names)
    index = hap.df_align['INDEX'].to_numpy(dtype='int64')
    trim = hap.df_trim[['POS', 'END', 'INDEX']].set_index('INDEX').astype(int).reindex(list(index), fill_value=-1)
    tp, te = trim['POS'].to_numpy(dtype='int64'), trim['END'].to_numpy(dtype='int64')
    k_util = KmerUtil(31)

    class Lane:
        pass
    lanes = []
    for li in range(max(lane_counts)):
        ln = Lane()
        ln.ctx = _lib.Context(0)
        if li == 0:
            ln.ctx.seq_load(_lib.PAV_ROLE_REF, names, [hap.ref.seqs[n] for n in names])
        else:
            ln.ctx.seq_share(lanes[0].ctx, _lib.PAV_ROLE_REF)
        ln.ctx.seq_load(_lib.PAV_ROLE_TIG, hap.tig_names, [hap.tig_seqs[n] for n in hap.tig_names])
        ln.ctx.cigar_load(*packed)
        ln.ctx._inv_loaded = ('ref.fa', 'tig.fa')
        ln.lift = AlignLift(hap.df_trim, hap.tig_lengths)
        ln.params = ln.ctx.flag_params(sig_filter=_lib.SIG_SINGLE_CLUSTER)
        ln.found = io.StringIO()
        ln.ctx.seq_pack(_lib.PAV_ROLE_TIG)
        ln.ctx.cigar_call()
        ln.regions = pavinv.loci_regions(ln.ctx, ln.ctx.cigar_flag(tp, te, ln.params)[1])
        lanes.append(ln)

    def step(ln, phase):
        if phase in ('call', 'all'):
            ln.ctx.seq_pack(_lib.PAV_ROLE_TIG)
            ln.ctx.cigar_call()
            if phase == 'call':
                ln.ctx.sync()
        if phase in ('flag', 'all'):
            flag = ln.ctx.cigar_flag(tp, te, ln.params)
            if phase == 'all':
                ln.regions = pavinv.loci_regions(ln.ctx, flag[1])
        if phase in ('scan', 'all'):
            ln.found.seek(0); ln.found.truncate()
            pavinv.scan_for_inv_batch(ln.regions, 'ref.fa', 'tig.fa', ln.lift, k_util, log=io.StringIO(), ctx=ln.ctx,
                                      eager_tables=False, found_out=ln.found)

    def run(phase, n, steps):
        bar = threading.Barrier(n + 1)

        def work(ln):
            bar.wait()
            for _ in range(steps):
                step(ln, phase)
            ln.ctx.sync()
            bar.wait()
        th = [threading.Thread(target=work, args=(lanes[i],)) for i in range(n)]
        for t in th:
            t.start()
        bar.wait()
        t0 = time.perf_counter()
        bar.wait()
        dt = time.perf_counter() - t0
        for t in th:
            t.join()
        return dt / (steps * n) * 1e3

    out = {}
    for phase in args.phases.split(','):
        out[phase] = {}
        for n in lane_counts:
            run(phase, n, 4)
            ms = sorted(run(phase, n, args.steps) for _ in range(3))[1]
            rec = {'ms_per_pass': round(ms, 4)}
            if args.kernels:
                for ln in lanes[:n]:
                    ln.ctx.prof_reset(); ln.ctx.prof_enable(True)
                run(phase, n, args.steps)
                tot = {}
                for ln in lanes[:n]:
                    for k, v in ln.ctx.prof_read().items():
                        tot[k] = tot.get(k, 0.0) + v[1]
                    ln.ctx.prof_enable(False)
                per = {k: round(v / (args.steps * n), 4) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])}
                rec['kernel_ms_per_pass'] = round(sum(per.values()), 4)
                rec['kernels'] = per
            out[phase][str(n)] = rec
            print(phase, n, json.dumps(rec), flush=True)
    print('LANE_SCALING ' + json.dumps(out))


if __name__ == '__main__':
    main()
