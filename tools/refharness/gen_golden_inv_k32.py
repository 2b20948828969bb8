#!/usr/bin/env python3
"""
Golden vectors of the inversion scan with inv_k_size = 32 (rules/call_inv.snakefile:131 takes any k; PAV's default is 31), made by
running the *reference itself* (pavlib.inv.scan_for_inv -> scripts/density.py -> scipy) like tools/refharness/gen_golden_inv.py, whose
capture code this script reuses.  Build container only.

A 32-mer fills the 64-bit word the device packs k-mers into, and one of them - thirty-two T, all bits set as kanapy spells it - is the
word that marks a free slot of the device's hash tables.  The locus therefore carries poly-T and poly-A tracts of 40 - 48 bases inside
the scanned region (outside and inside the planted inversion), so that this k-mer and its reverse complement are in the reference
set, in the contig stream and in the density tables.

Outputs (committed): tests/golden/inv_k32/  - the files of an inv_<case> directory plus params.json {"k": 32}.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import gen_golden_inv as g  # noqa: E402  (imports the reference through refenv)
from pav_amd import synth  # noqa: E402


def main():
    name, length = 'chrT', 64_000
    ref = synth.make_reference(32, {name: length}, n_every=0, inv_every=0, threads=1)
    s = ref.seqs[name]
    inv = synth.Inversion(name, 22_000, 31_000, 1_200)
    s[inv.end - inv.repeat:inv.end] = synth.revcomp(s[inv.pos:inv.pos + inv.repeat])
    for at, n, base in ((18_500, 45, 'T'), (26_000, 40, 'T'), (27_500, 48, 'A'), (34_200, 41, 'A'), (36_000, 33, 'T')):
        s[at:at + n] = ord(base)
    ref.inversions = [inv]
    hap = synth.make_haplotype(ref, 32 * 64, 'h1', segments={name: [(0, length)]}, rev_frac=0.0, threads=1, decoys_per_inv=0,
                               snv_rate=1e-3, indel_rate=2e-4)
    g.run_case('inv_k32', ref, hap, [
        (name, 22_000, 31_000, 'CLUSTER_SNV', None),
        (name, 25_000, 28_000, 'CLUSTER_SNV', None),             # partial flag: the scan expands over the tracts
        (name, 8_000, 9_500, 'CLUSTER_INDEL', None),             # no inversion here
    ], k=32)
    with open(os.path.join(g.GOLD, 'inv_k32', 'params.json'), 'w') as fh:
        json.dump({'k': 32}, fh)
    # the k-mer made of thirty-two T must be in the tables, or the case does not test what it is for
    with open(os.path.join(g.GOLD, 'inv_k32', 'scans.json')) as fh:
        scans = json.load(fh)
    ones = 0
    for rec in scans:
        if rec['call']:
            t = np.load(os.path.join(g.GOLD, 'inv_k32', 'density_%s.npz' % rec['call']['id']))
            ones += int((t['KMER'] == np.uint64(0xFFFFFFFFFFFFFFFF)).sum())
            print(rec['call']['id'], 'rows', t['KMER'].shape[0], 'all-T rows', int((t['KMER'] == np.uint64(0xFFFFFFFFFFFFFFFF)).sum()),
                  'all-A rows', int((t['KMER'] == np.uint64(0)).sum()), 'max', int(t['KMER'].max()))
    assert ones > 0, 'no all-T 32-mer in any density table'


if __name__ == '__main__':
    main()
