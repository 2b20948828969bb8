#!/usr/bin/env python3
"""Probe: B haplotypes as ONE problem instance of one context and one host thread (reference loaded B times under names
chr@i, contigs tig@i, the alignment tables concatenated) - what a pass costs per haplotype when the launches are B times
larger instead of B lanes running beside each other.
    python tools/batch_probe.py [--batch 1,2,3,6] [--steps 12]"""
import argparse, io, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', default='1,2,3,6')
    ap.add_argument('--steps', type=int, default=12)
    ap.add_argument('--scale', type=float, default=1.0)
    ap.add_argument('--no-build', action='store_true')
    ap.add_argument('--kernels', action='store_true')
    args = ap.parse_args()
    if not args.no_build:
        import __graft_entry__ as g
        g.build_cpu_side()
    import numpy as np
    import pandas as pd
    from pav_amd import _lib, cigarcall, synth, inv as pavinv
    from pav_amd.align import AlignLift
    from pav_amd.kmer import KmerUtil
    sizes = [int(x) for x in args.batch.split(',')]
    haps, ref = [], None
    for i in range(max(sizes)):
        h = synth.config2(seed=1002, scale=args.scale, hap_index=i, ref=ref, threads=8, pair_frac=0.009)
        ref = h.ref
        haps.append(h)
    k_util = KmerUtil(31)
    out = {}
    for B in sizes:
        ctx = _lib.Context(0)
        ref_names = [f'{n}@{i}' for i in range(B) for n in ref.names]
        ctx.seq_load(_lib.PAV_ROLE_REF, ref_names, [ref.seqs[n] for i in range(B) for n in ref.names])
        tig_names = [f'{n}@{i}' for i in range(B) for n in haps[i].tig_names]
        ctx.seq_load(_lib.PAV_ROLE_TIG, tig_names, [haps[i].tig_seqs[n] for i in range(B) for n in haps[i].tig_names])
        al, tr, base = [], [], 0
        for i in range(B):
            a, t = haps[i].df_align.copy(), haps[i].df_trim.copy()
            for d in (a, t):
                d['#CHROM'] = d['#CHROM'].astype(str) + f'@{i}'
                d['QRY_ID'] = d['QRY_ID'].astype(str) + f'@{i}'
                d['INDEX'] = d['INDEX'] + base
            base += int(haps[i].df_align['INDEX'].max()) + 1
            al.append(a); tr.append(t)
        df_align, df_trim = pd.concat(al, ignore_index=True), pd.concat(tr, ignore_index=True)
        ctx.cigar_load(*cigarcall.pack_alignments(df_align, ref_names, tig_names))
        ctx._inv_loaded = ('ref.fa', 'tig.fa')
        index = df_align['INDEX'].to_numpy(dtype='int64')
        trim = df_trim[['POS', 'END', 'INDEX']].set_index('INDEX').astype(int).reindex(list(index), fill_value=-1)
        tp, te = trim['POS'].to_numpy(dtype='int64'), trim['END'].to_numpy(dtype='int64')
        tig_lengths = pd.Series({n: int(haps[i].tig_seqs[n0].shape[0]) for i in range(B) for n0, n in ((m, f'{m}@{i}') for m in haps[i].tig_names)}, dtype=np.int64)
        lift = AlignLift(df_trim, tig_lengths)
        params = ctx.flag_params(sig_filter=_lib.SIG_SINGLE_CLUSTER)
        found = io.StringIO()
        seen = {}

        def step():
            ctx.seq_pack(_lib.PAV_ROLE_TIG)
            c = ctx.cigar_call()
            flag = ctx.cigar_flag(tp, te, params)
            regions = pavinv.loci_regions(ctx, flag[1])
            found.seek(0); found.truncate()
            res = pavinv.scan_for_inv_batch(regions, 'ref.fa', 'tig.fa', lift, k_util, log=io.StringIO(), ctx=ctx, eager_tables=False, found_out=found)
            seen.update(n_snv=int(c.n_snv), n_indel=int(c.n_indel), regions=len(regions), calls=sum(1 for r in res if r is not None and not isinstance(r, Exception)))
        for _ in range(3):
            step()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        ctx.sync()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        rec = {'ms_per_pass_of_the_batch': round(ms, 4), 'ms_per_haplotype': round(ms / B, 4), **seen}
        if args.kernels:
            ctx.prof_reset(); ctx.prof_enable(True)
            for _ in range(args.steps):
                step()
            ctx.sync()
            per = {k: round(v[1] / args.steps / B, 4) for k, v in sorted(ctx.prof_read().items(), key=lambda kv: -kv[1][1])}
            ctx.prof_enable(False)
            rec['kernel_ms_per_haplotype'] = round(sum(per.values()), 4)
            rec['kernels'] = per
        out[str(B)] = rec
        print(B, json.dumps(rec), flush=True)
        ctx.close()
    print('BATCH_PROBE ' + json.dumps(out))


if __name__ == '__main__':
    main()
