#!/usr/bin/env python3
"""
The reference ITSELF on loci of the full-size bench haplotype: pavlib.inv.scan_for_inv (-> scripts/density.py -> scipy, unmodified)
is run on twenty-odd flagged loci of synth.config2(seed=1002, scale=1.0, pair_frac=0.009) - the haplotype bench.py times - with the
haplotype's own 3 GB FASTA files and its complete trimmed alignment table, so coordinates, lifts and expansion limits are the
real ones: the ten calls with the largest discovery regions (two rounds, 350 - 480 kbp), no-call loci of every kind of ending the
scan has on this haplotype, and the loci with the most expansion rounds.  Until now the scan control at this size was pinned on
the product's own Python state machine answered by the oracle (tests/golden/fullsize_inv_calls.json); with this the native driver
is compared with the reference's logs, calls and final tables directly (tests/test_gpu_fullsize.py).

Build container only (needs /root/reference).  Input: the loci list written on the GPU box by tools/dump_fullsize_loci.py.
    python tools/refharness/gen_golden_fullsize_loci.py gpurun_out/r05/fullsize_loci.json
Takes ~6 min on 8 cores for the default selection, ~40 min with PAV_GOLDEN_ALL=1 (every call of the haplotype + every 25th of the
other loci: what is committed); PAV_GOLDEN_WORK names the scratch directory (7 GB).
Committed: tests/golden/fullsize_loci/scans.json (digests, not tables).
"""
import hashlib
import io
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

WORK = os.environ.get('PAV_GOLDEN_WORK', '/tmp/pav_fullsize_loci')


def choose(loci):
    """Which loci the reference is run on: (region number, why)."""
    picked, seen = [], set()

    def take(i, why):
        if i not in seen:
            seen.add(i)
            picked.append((i, why))
    calls = sorted((x for x in loci if x['call']), key=lambda x: -x['call']['disc_len'])
    for x in calls[:10]:
        take(x['region'], 'one of the ten calls with the largest discovery region')
    # every kind of ending without a call, the case with the most scan rounds of each kind first
    kinds = {}
    for x in loci:
        if x['call'] or x['error'] or not x['log']:
            continue
        last = x['log'][-1]
        kind = ''.join(ch for ch in last.split(':')[0] if not ch.isdigit())[:60]
        rounds = sum(1 for ln in x['log'] if ln.startswith('Scanning region'))
        kinds.setdefault(kind, []).append((rounds, x['region']))
    for kind, lst in sorted(kinds.items()):
        lst.sort(key=lambda t: (-t[0], t[1]))
        for rounds, i in lst[:2]:
            take(i, f'no call: "{kind.strip()}" after {rounds} round(s)')
    # the loci with the most expansion rounds, whatever their ending
    by_rounds = sorted(loci, key=lambda x: (-sum(1 for ln in x['log'] if ln.startswith('Scanning region')), x['region']))
    for x in by_rounds[:5]:
        take(x['region'], 'one of the five loci with the most scan rounds')
    for x in calls[-3:]:
        take(x['region'], 'one of the three calls with the smallest discovery region')
    if os.environ.get('PAV_GOLDEN_ALL') == '1':                    # every call of the haplotype + every 25th of the other loci
        for x in calls:
            take(x['region'], 'a call of the haplotype (all of them are run)')
        for x in [y for y in loci if not y['call']][::25]:
            take(x['region'], 'every 25th locus without a call')
    return picked


def prepare():
    from pav_amd import synth
    os.makedirs(WORK, exist_ok=True)
    ref_fa, tig_fa, aln = os.path.join(WORK, 'ref.fa'), os.path.join(WORK, 'tig.fa'), os.path.join(WORK, 'align.tsv')
    if not (os.path.exists(ref_fa + '.fai') and os.path.exists(tig_fa + '.fai') and os.path.exists(aln)):
        hap = synth.config2(seed=1002, scale=1.0, threads=8, pair_frac=0.009)
        synth.write_fasta(ref_fa, hap.ref.names, hap.ref.seqs, line=0)
        synth.write_fasta(tig_fa, hap.tig_names, hap.tig_seqs, line=0)
        hap.df_trim.to_csv(aln, sep='\t', index=False)
        with open(os.path.join(WORK, 'hap.txt'), 'w') as fh:
            fh.write(hap.hap)
    return ref_fa, tig_fa, aln


def run_one(job):
    i, why, chrom, pos, end, ref_fa, tig_fa, aln, threads = job
    import gen_golden_inv as g                                    # imports the reference through refenv
    pavlib, kanapy, svpoplib = g.pavlib, g.kanapy, g.svpoplib
    t0 = time.time()
    df_aln = pd.read_csv(aln, sep='\t')
    align_lift = pavlib.align.AlignLift(df_aln, svpoplib.ref.get_df_fai(tig_fa + '.fai'))
    k_util = kanapy.util.kmer.KmerUtil(31)
    log = io.StringIO()
    with g.Capture(align_lift) as cap:
        try:
            call = pavlib.inv.scan_for_inv(pavlib.seq.Region(chrom, pos, end), ref_fa, tig_fa, align_lift, k_util, threads=threads, log=log)
            err = None
        except RuntimeError as ex:
            call, err = None, str(ex)
    rec = {'region': i, 'why': why, 'flag': {'chrom': chrom, 'pos': pos, 'end': end}, 'iterations': cap.iterations,
           'log': log.getvalue().splitlines(), 'error': err, 'call': None}
    if call is not None:
        with open(os.path.join(WORK, 'hap.txt')) as fh:
            hap_name = fh.read().strip()
        row = g.inv_bed_row(call, hap_name, 'RGN', tig_fa)
        bed = {k2: (int(v) if isinstance(v, (int, np.integer)) else v) for k2, v in row.items()}
        seq = bed.pop('SEQ')
        bed['SEQ_sha1'], bed['SEQ_len'] = hashlib.sha1(seq.encode()).hexdigest(), len(seq)
        df = call.df
        rec['call'] = {
            'id': call.id, 'svlen': int(call.svlen),
            **{nm: g.region_dict(getattr(call, nm)) for nm in ('region_ref_outer', 'region_ref_inner', 'region_tig_outer', 'region_tig_inner',
                                                                'region_ref_discovery', 'region_tig_discovery')},
            'bed_row': bed, 'n_rows': int(df.shape[0]),
            'index_sha1': g.digest(df['INDEX'].to_numpy(dtype=np.int64)),
            'state_mer_sha1': g.digest(df['STATE_MER'].to_numpy(dtype=np.int8)),
            'state_sha1': g.digest(df['STATE'].to_numpy(dtype=np.int8)),
            'kern_sum': [float(df[c].sum()) for c in ('KERN_FWD', 'KERN_FWDREV', 'KERN_REV')],
            'flank_counts': {str(k2): int(v) for k2, v in df['FLANK'].value_counts().items()},
            'flank_sha1': hashlib.sha1('\n'.join(df['FLANK'].tolist()).encode()).hexdigest(),
            'unpinned_columns': ['KMER', 'MATCH'],
        }
    rec['reference_seconds'] = round(time.time() - t0, 1)
    print(f'  region {i} {chrom}:{pos}-{end} ({why}) -> {call} rounds={len(cap.iterations)} {rec["reference_seconds"]} s', flush=True)
    return rec


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'r05', 'fullsize_loci.json')
    with open(src) as fh:
        loci = json.load(fh)
    picked = choose(loci)
    print(f'{len(picked)} of {len(loci)} loci chosen', flush=True)
    ref_fa, tig_fa, aln = prepare()
    procs = int(os.environ.get('PAV_GOLDEN_PROCS', '4'))
    jobs = [(i, why, loci[i]['chrom'], loci[i]['pos'], loci[i]['end'], ref_fa, tig_fa, aln, max(1, 8 // procs)) for i, why in picked]
    # the longest first: the ten large calls take minutes each
    jobs.sort(key=lambda j: -(loci[j[0]]['call']['disc_len'] if loci[j[0]]['call'] else 0))
    with mp.get_context('spawn').Pool(procs) as pool:
        recs = pool.map(run_one, jobs, chunksize=1)
    recs.sort(key=lambda r: r['region'])
    d = os.path.join(ROOT, 'tests', 'golden', 'fullsize_loci')
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, 'scans.json'), 'w') as fh:
        json.dump({'generator': 'tools/refharness/gen_golden_fullsize_loci.py', 'haplotype': 'synth.config2(seed=1002, scale=1.0, pair_frac=0.009)',
                   'n_loci_of_the_haplotype': len(loci), 'scans': recs}, fh, indent=1)
    print('wrote', os.path.join(d, 'scans.json'))


if __name__ == '__main__':
    main()
