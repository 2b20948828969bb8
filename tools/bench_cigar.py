#!/usr/bin/env python3
"""CIGAR-call only, ONE lane (one haplotype, one host thread): wall time per call and the HIP-event time of every kernel.
    python tools/bench_cigar.py [--steps 40] [--scale 1.0]
Used to iterate on the call chain (tok_tiles -> tile_scan -> walk_indel -> homology_kernel || walk_snv)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--scale', type=float, default=1.0)
    ap.add_argument('--seed', type=int, default=1002)
    ap.add_argument('--no-build', action='store_true', help='under rocprofv3: everything is built beforehand')
    args = ap.parse_args()
    if not args.no_build:
        import __graft_entry__ as g
        g.build_cpu_side()
    from pav_amd import _lib, cigarcall, synth
    hap = synth.config2(seed=args.seed, scale=args.scale, threads=8, pair_frac=0.009)
    names = hap.ref.names
    ctx = _lib.Context(0)
    ctx.seq_load(_lib.PAV_ROLE_REF, names, [hap.ref.seqs[n] for n in names])
    ctx.seq_load(_lib.PAV_ROLE_TIG, hap.tig_names, [hap.tig_seqs[n] for n in hap.tig_names])
    ctx.cigar_load(*cigarcall.pack_alignments(hap.df_align, names, hap.tig_names))
    for _ in range(3):
        ctx.seq_pack(_lib.PAV_ROLE_TIG)
        c = ctx.cigar_call()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.seq_pack(_lib.PAV_ROLE_TIG)
        c = ctx.cigar_call()
    ctx.sync()
    wall = (time.perf_counter() - t0) / args.steps * 1e3
    ctx.prof_reset()
    ctx.prof_enable(True)
    for _ in range(args.steps):
        ctx.seq_pack(_lib.PAV_ROLE_TIG)
        c = ctx.cigar_call()
    ctx.sync()
    prof = ctx.prof_read()
    ctx.prof_enable(False)
    kern = {k: round(v[1] / max(1, v[0]), 4) for k, v in sorted(prof.items())}
    print(f'cigar-only single lane: {wall:.4f} ms per call = {c.aligned_bases / wall / 1e6:.0f} Gbp/s; '
          f'{c.n_ops} ops, {c.n_snv} SNV, {c.n_indel} INDEL; PAV_PRIO={os.environ.get("PAV_PRIO", "")}')
    print('  kernels (ms, HIP events, same lane):', kern, 'sum', round(sum(kern.values()), 4))
    ctx.close()


if __name__ == '__main__':
    main()
