"""Shared helpers for the parity tests (golden-case loading, oracle pipeline, frame comparison)."""
import io
import json
import os

import numpy as np
import pandas as pd

from pav_amd import cigarcall, rules
from pav_amd.fasta import open_fasta

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def case_k(case_dir):
    """k-mer size a golden inversion case was made with (params.json beside its scans.json; PAV's default 31 without one)."""
    import json
    path = os.path.join(case_dir if os.path.isabs(case_dir) else os.path.join(GOLD, case_dir), 'params.json')
    if not os.path.exists(path):
        return 31
    with open(path) as fh:
        return int(json.load(fh)['k'])


def golden_case(name):
    d = os.path.join(GOLD, name)
    df_align = rules.read_align_bed(os.path.join(d, 'align.tsv'))
    df_trim = rules.read_trim_bed(os.path.join(d, 'trim.tsv'))
    return d, df_align, df_trim


def golden_text(name, table):
    with open(os.path.join(GOLD, name, table + '.tsv')) as fh:
        return fh.read()


def frame_text(df):
    buf = io.StringIO()
    df.to_csv(buf, sep='\t', index=False)
    return buf.getvalue()


def kat():
    with open(os.path.join(GOLD, 'kat.json')) as fh:
        return json.load(fh)


def cigar_errors():
    with open(os.path.join(GOLD, 'cigar_errors.json')) as fh:
        return json.load(fh)


def seq_arrays(case_dir, df_align):
    ref_fa = open_fasta(os.path.join(case_dir, 'ref.fa'))
    tig_fa = open_fasta(os.path.join(case_dir, 'tig.fa'))
    return ref_fa, tig_fa


def oracle_records(ref_names, ref_arrays, tig_names, tig_arrays, df_align):
    """Run the CPU oracle on the same marshalled inputs the device gets."""
    from oracle import oracle
    aln, text, off = cigarcall.pack_alignments(df_align, ref_names, tig_names)
    return oracle.cigar_call(ref_arrays, tig_arrays, aln, text, off)


def oracle_frames(case_dir, df_align, df_trim, hap='h1', with_filter=True):
    ref_fa, tig_fa = seq_arrays(case_dir, df_align)
    snv, indel, blob, err = oracle_records(ref_fa.names, [ref_fa[n] for n in ref_fa.names],
                                           tig_fa.names, [tig_fa[n] for n in tig_fa.names], df_align)
    assert err.kind == 0, f'oracle error kind {err.kind}'
    df_snv, df_insdel = cigarcall.records_to_frames(snv, indel, blob, df_align, hap)
    if with_filter:
        df_snv = rules.apply_trim_filter(df_snv, df_trim)
        df_insdel = rules.apply_trim_filter(df_insdel, df_trim)
    return df_snv, df_insdel


def assert_records_equal(a, b, what):
    assert a.dtype == b.dtype, what
    assert a.shape == b.shape, f'{what}: {a.shape} vs {b.shape}'
    if a.shape[0] == 0:
        return
    for name in a.dtype.names:
        if name == 'pad':
            continue
        bad = np.flatnonzero(a[name] != b[name])
        assert bad.size == 0, f'{what}.{name}: {bad.size} mismatches, first at {bad[0]}: {a[bad[0]]} vs {b[bad[0]]}'


def config1_case():
    """BASELINE.json configs[0] regenerated from its seed + the digests of what pavlib wrote for it (tests/golden/config1.json,
    written by tools/refharness/gen_golden_cigar.py).  Asserts that the generator still produces the inputs the reference saw."""
    import hashlib
    from pav_amd import synth
    with open(os.path.join(GOLD, 'config1.json')) as fh:
        gold = json.load(fh)
    hap = synth.config1()
    md5 = lambda b: hashlib.md5(b).hexdigest()   # noqa: E731
    assert md5(hap.ref.seqs['chr20'].tobytes()) == gold['inputs']['ref_md5'], 'synth.config1() no longer generates the committed case'
    assert md5(b''.join(hap.tig_seqs[n].tobytes() for n in hap.tig_names)) == gold['inputs']['tig_md5']
    assert md5(hap.df_align.to_csv(sep='\t', index=False).encode()) == gold['inputs']['align_tsv_md5']
    return hap, gold


def assert_config1_text(name, text, gold):
    import hashlib
    assert len(text) == gold[name]['tsv_bytes'], name
    assert hashlib.md5(text).hexdigest() == gold[name]['tsv_md5'], f'{name} table differs from what pavlib wrote for configs[0]'


def assert_config1_tables(df_snv, df_insdel, gold):
    """The two tables of rule call_cigar as text against the digests of the reference's own output."""
    for name, df in (('snv', df_snv), ('insdel', df_insdel)):
        assert df.shape[0] == gold[name]['rows'], name
        assert_config1_text(name, frame_text(df).encode(), gold)


_INV_LARGE = {}


def inv_large_case():
    """tests/golden/inv_large (tools/refharness/gen_golden_inv_large.py): the reference's scans of two 150 / 200 kb
    inversions whose regions grow to 337 / 462 kbp.  The 4.6 Mbp of sequence are regenerated from the seed (asserting the md5
    of what the reference saw) and written as FASTA files into a scratch directory once per session.
    -> (dir with ref.fa / tig.fa, ref, hap, golden dict)"""
    import hashlib
    import tempfile
    from pav_amd import synth
    if not _INV_LARGE:
        with open(os.path.join(GOLD, 'inv_large', 'scans.json')) as fh:
            gold = json.load(fh)
        ref, hap, flags = synth.large_inversions()
        md5 = lambda b: hashlib.md5(b).hexdigest()   # noqa: E731
        assert md5(b''.join(ref.seqs[n].tobytes() for n in ref.names)) == gold['inputs']['ref_md5'], \
            'synth.large_inversions() no longer generates the committed case'
        assert md5(b''.join(hap.tig_seqs[n].tobytes() for n in hap.tig_names)) == gold['inputs']['tig_md5']
        assert md5(hap.df_trim.to_csv(sep='\t', index=False).encode()) == gold['inputs']['align_tsv_md5']
        assert [(f[0], f[1], f[2]) for f in flags] == [(s['flag']['chrom'], s['flag']['pos'], s['flag']['end']) for s in gold['scans']]
        d = tempfile.mkdtemp(prefix='pav_inv_large_')
        synth.write_fasta(os.path.join(d, 'ref.fa'), ref.names, ref.seqs)
        synth.write_fasta(os.path.join(d, 'tig.fa'), hap.tig_names, hap.tig_seqs)
        _INV_LARGE['case'] = (d, ref, hap, gold)
    return _INV_LARGE['case']


def usable_cpus():
    from pav_amd import shard
    return max(1, int(shard.effective_cpus()))
