#!/usr/bin/env python3
"""The flagged loci of the bench haplotype (synth.config2(seed=1002, scale=1.0, pair_frac=0.009)) and what the native scan makes of
each - region, log text, call - as JSON: the list tools/refharness/gen_golden_fullsize_loci.py picks its loci from (the reference
itself is then run on those in the build container).  Needs the GPU box:
    python tools/dump_fullsize_loci.py gpurun_out/r05/fullsize_loci.json"""
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import __graft_entry__ as g  # noqa: E402

g.build_cpu_side()

from pav_amd import _lib, cigarcall, inv as pavinv, synth  # noqa: E402
from pav_amd.align import AlignLift  # noqa: E402
from pav_amd.kmer import KmerUtil  # noqa: E402


def main(out_path):
    hap = synth.config2(seed=1002, scale=1.0, threads=16, pair_frac=0.009)
    names = hap.ref.names
    with _lib.Context(0) as ctx:
        ctx.seq_load(_lib.PAV_ROLE_REF, names, [hap.ref.seqs[n] for n in names])
        ctx.seq_load(_lib.PAV_ROLE_TIG, hap.tig_names, [hap.tig_seqs[n] for n in hap.tig_names])
        cigarcall.call_records(ctx, hap.df_align)
        index = hap.df_align['INDEX'].to_numpy(dtype='int64')
        trim = hap.df_trim[['POS', 'END', 'INDEX']].set_index('INDEX').astype(int).reindex(list(index), fill_value=-1)
        _, loci, _ = ctx.cigar_flag(trim['POS'].to_numpy(dtype='int64'), trim['END'].to_numpy(dtype='int64'),
                                    ctx.flag_params(sig_filter=_lib.SIG_SINGLE_CLUSTER))
        regions = pavinv.loci_regions(ctx, loci)
        lift = AlignLift(hap.df_trim, hap.tig_lengths)
        import tempfile
        import oracle_scan
        d = tempfile.mkdtemp(prefix='pav_fullsize_')
        fa = (os.path.join(d, 'ref.fa'), os.path.join(d, 'tig.fa'))          # only the .fai of the reference is ever read (inv.py:201)
        oracle_scan.write_fai(fa[0] + '.fai', names, hap.ref.lengths)
        ctx._inv_loaded = fa
        logs = [io.StringIO() for _ in regions]
        out = pavinv.scan_for_inv_batch(regions, fa[0], fa[1], lift, KmerUtil(31), logs=logs, ctx=ctx, native=True, eager_tables=False,
                                        found_out=io.StringIO())
    doc = []
    for i, (r, c, lg) in enumerate(zip(regions, out, logs)):
        text = lg.getvalue()
        doc.append({'region': i, 'chrom': r.chrom, 'pos': int(r.pos), 'end': int(r.end), 'log': text.splitlines(),
                    'call': None if c is None or isinstance(c, RuntimeError) else
                    {'id': c.id, 'svlen': int(c.svlen), 'disc_len': int(len(c.region_ref_discovery)), 'n_rows': int(c.df.shape[0]) if False else None},
                    'error': str(c) if isinstance(c, RuntimeError) else None})
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    with open(out_path, 'w') as fh:
        json.dump(doc, fh)
    print(len(doc), 'loci ->', out_path)


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'fullsize_loci.json'))
