#!/usr/bin/env python3
"""
tests/golden/fullsize_inv_calls.json: the inversion calls of the bench haplotype (BASELINE configs[2] per-GPU share: seed 1002,
scale 1.0, pair_frac 0.009 - the haplotype bench.py times) as made by a scan that never uses the density kernels or the native
scan driver:

    flagged loci     pav_cigar_flag on the device (the records are bit-exact vs the oracle at this size,
                     tests/test_gpu_fullsize.py; the flagging itself is pinned on the reference's rule bodies)
    scan control     pav_amd.inv's Python state machine (pinned on the reference's logs / calls, tests/golden/inv_*)
    density tables   the CPU oracle (pinned on the reference's tables up to 462 kbp, tests/golden/inv_large), every region of
                     every round (tests/oracle_scan.py)

Needs the GPU box for the flagging (run there:  python tools/gen_fullsize_inv_digest.py gpurun_out/fullsize_inv_calls.json),
~12 GB of host memory and a few minutes of oracle time on all cores.  The test asserts that the native driver + the device
kernels produce exactly these calls, regions and log text.
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import __graft_entry__ as g  # noqa: E402

g.build_cpu_side()

import oracle_scan  # noqa: E402
import util  # noqa: E402
from pav_amd import _lib, cigarcall, inv as pavinv, synth  # noqa: E402
from pav_amd.align import AlignLift  # noqa: E402
from pav_amd.kmer import KmerUtil  # noqa: E402


def main(out_path):
    t0 = time.time()
    hap = synth.config2(seed=1002, scale=1.0, threads=8, pair_frac=0.009)
    names = hap.ref.names
    ctx = _lib.Context(0)
    ctx.seq_load(_lib.PAV_ROLE_REF, names, [hap.ref.seqs[n] for n in names])
    ctx.seq_load(_lib.PAV_ROLE_TIG, hap.tig_names, [hap.tig_seqs[n] for n in hap.tig_names])
    cigarcall.call_records(ctx, hap.df_align)
    index = hap.df_align['INDEX'].to_numpy(dtype='int64')
    trim = hap.df_trim[['POS', 'END', 'INDEX']].set_index('INDEX').astype(int).reindex(list(index), fill_value=-1)
    _, loci, _ = ctx.cigar_flag(trim['POS'].to_numpy(dtype='int64'), trim['END'].to_numpy(dtype='int64'),
                                ctx.flag_params(sig_filter=_lib.SIG_SINGLE_CLUSTER))
    regions = pavinv.loci_regions(ctx, loci)
    ctx.close()
    print(f'{len(regions)} flagged regions after {time.time() - t0:.0f} s', flush=True)
    lift = AlignLift(hap.df_trim, hap.tig_lengths)
    cpus = util.usable_cpus()
    flags = list(regions)
    out, log, octx = oracle_scan.oracle_scan(flags, names, hap.ref.seqs, hap.tig_names, hap.tig_seqs, lift, KmerUtil(31),
                                             threads=1, pool=cpus)
    assert not any(isinstance(c, RuntimeError) for c in out)
    calls = []
    for i, c in enumerate(out):
        if c is None:
            continue
        rec = oracle_scan.call_record(c)
        df = c.df
        rec.update(region=i, n_rows=int(df.shape[0]), state_sha1=hashlib.sha1(df['STATE'].to_numpy(dtype=np.int8).tobytes()).hexdigest(),
                   state_mer_sha1=hashlib.sha1(df['STATE_MER'].to_numpy(dtype=np.int8).tobytes()).hexdigest())
        calls.append(rec)
    sizes = sorted((it[2] - it[1] for it in octx.iterations), reverse=True)
    doc = {'generator': 'tools/gen_fullsize_inv_digest.py', 'haplotype': 'synth.config2(seed=1002, scale=1.0, pair_frac=0.009)',
           'n_regions': len(flags),
           'regions_sha1': hashlib.sha1('\n'.join(f'{r.chrom}:{r.pos}-{r.end}' for r in flags).encode()).hexdigest(),
           'n_iterations': len(octx.iterations), 'largest_regions_bp': sizes[:10],
           'log_sha1': hashlib.sha1(log.encode()).hexdigest(), 'log_lines': log.count('\n'), 'calls': calls,
           'oracle_seconds': round(time.time() - t0, 1), 'cpus': cpus}
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    with open(out_path, 'w') as fh:
        json.dump(doc, fh, indent=1)
    print(f'{len(calls)} calls of {len(flags)} regions, {len(octx.iterations)} iterations, largest {sizes[:3]}, '
          f'{time.time() - t0:.0f} s -> {out_path}')


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'fullsize_inv_calls.json'))
