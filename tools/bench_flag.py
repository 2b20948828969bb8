#!/usr/bin/env python3
"""Time pav_cigar_flag on the bench haplotype and cross-check it against the array-level entry points (GPU box)."""
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
    ap.add_argument('--reps', type=int, default=5)
    args = ap.parse_args()
    import __graft_entry__ as g
    g.build_cpu_side()
    from pav_amd import _lib, cigarcall, synth, flag

    hap = synth.config2(seed=args.seed, scale=args.scale, hap_index=0, threads=8)
    ctx = _lib.Context(0)
    names = hap.ref.names
    ctx.seq_load(_lib.PAV_ROLE_REF, names, [hap.ref.seqs[n] for n in names])
    ctx.seq_load(_lib.PAV_ROLE_TIG, hap.tig_names, [hap.tig_seqs[n] for n in hap.tig_names])
    aln, text, off = cigarcall.pack_alignments(hap.df_align, names, hap.tig_names)
    ctx.cigar_load(aln, text, off)
    counts = ctx.cigar_call()
    index = hap.df_align['INDEX'].to_numpy(dtype='int64')
    trim = hap.df_trim[['POS', 'END', 'INDEX']].set_index('INDEX').astype(int).reindex(list(index), fill_value=-1)
    tp, te = trim['POS'].to_numpy(dtype='int64'), trim['END'].to_numpy(dtype='int64')
    report = {'n_snv': int(counts.n_snv), 'n_indel': int(counts.n_indel)}
    for sig in ('svindel', 'single_cluster'):
        res = flag.flag_from_calls(ctx, tp, te, inv_sig_filter=sig)
        ctx.prof_reset(); ctx.prof_enable(True)
        t0 = time.perf_counter()
        for _ in range(args.reps):
            tables, loci, cnt = ctx.cigar_flag(tp, te, ctx.flag_params(sig_filter=flag.sig_filter_code(sig)))
        wall = (time.perf_counter() - t0) / args.reps * 1e3
        prof = ctx.prof_read(); ctx.prof_enable(False)
        df = res['flagged_regions']
        report[sig] = {'wall_ms': round(wall, 3), 'kernels_ms': {k: round(v[1] / max(1, v[0]), 4) for k, v in sorted(prof.items())},
                       'tables': {k: int(len(v)) for k, v in tables.items()}, 'loci': int(len(loci)), 'try_inv': int(df['TRY_INV'].sum()),
                       'by_type': df['TYPE'].value_counts().to_dict(), **cnt}

    # cross-check: array-level entry points on the fetched records (numpy FILTER + lexsort on the host)
    snv, indel, _ = ctx.cigar_fetch(counts)
    ref_rank = np.argsort(np.argsort(np.array(names, dtype=object))).astype(np.uint32)
    def passing(rec):
        return (rec['pos'].astype(np.int64) > tp[rec['aln']]) & (rec['end'].astype(np.int64) < te[rec['aln']]) if 'end' in rec.dtype.names else \
               (rec['pos'].astype(np.int64) > tp[rec['aln']]) & (rec['pos'].astype(np.int64) + 1 < te[rec['aln']])
    ok = {}
    s = snv[passing(snv)]
    chrom = ref_rank[aln['ref_id'][s['aln']]]
    pos = s['pos'].astype(np.int64)
    o = np.lexsort((pos, chrom))
    a = ctx.flag_cluster(chrom[o], pos[o], pos[o] + 1, 200, 200, 20)
    tables, loci, cnt = ctx.cigar_flag(tp, te)
    ok['cluster_snv'] = a.tobytes() == tables['cluster_snv'].tobytes()
    v = indel[passing(indel)]
    chrom = ref_rank[aln['ref_id'][v['aln']]]
    pos, end, svlen = v['pos'].astype(np.int64), v['end'].astype(np.int64), v['svlen'].astype(np.int64)
    small = svlen < 50
    o = np.lexsort((end[small], pos[small], chrom[small]))
    a = ctx.flag_cluster(chrom[small][o], pos[small][o], end[small][o], 200, 200, 10)
    tables, loci, cnt = ctx.cigar_flag(tp, te)
    ok['cluster_indel'] = a.tobytes() == tables['cluster_indel'].tobytes()
    for name, sel in (('insdel_sv', svlen >= 50), ('insdel_indel', (svlen >= 4) & (svlen < 50))):
        ins, dl = sel & (v['svtype'] == 0), sel & (v['svtype'] == 1)
        a = ctx.flag_insdel(chrom[ins], pos[ins], svlen[ins], chrom[dl], pos[dl], end[dl], 2, 2000)
        tables, loci, cnt = ctx.cigar_flag(tp, te)
        ok[name] = a.tobytes() == tables[name].tobytes()
    report['fused_equals_array_path'] = ok
    print(json.dumps(report, indent=1))
    ctx.close()


if __name__ == '__main__':
    main()
