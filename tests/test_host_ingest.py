"""Alignment ingest: pav_amd.align.get_align_bed / pav_amd.rules.align_get_read_bed against
the files the reference's own rule body wrote (tests/golden/align_ingest, tools/refharness/gen_golden_align.py).  No GPU."""
import gzip
import json
import os

import pandas as pd
import pytest

from pav_amd import _lib, rules
from pav_amd.align import get_align_bed
from pav_amd.fasta import read_fai

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'align_ingest')


def gunzip(path):
    with gzip.open(path, 'rb') as fh:
        return fh.read()


@pytest.mark.parametrize('case', ['hap_seq', 'hap_noseq'])
def test_rule_align_get_read_bed_files_equal_the_reference(built, tmp_path, case):
    bed, head = str(tmp_path / 'aligned.bed.gz'), str(tmp_path / 'aligned.headers.gz')
    df = rules.align_get_read_bed(os.path.join(GOLD, case + '.sam.gz'), os.path.join(GOLD, 'tig.fa.fai'), 'h1', bed_out=bed,
                                  align_head_out=head)
    assert gunzip(bed) == gunzip(os.path.join(GOLD, case + '.bed.gz'))             # byte-identical table text
    assert gunzip(head) == gunzip(os.path.join(GOLD, case + '.headers.gz'))
    assert df.shape[0] == 106 and set(df['FLAGS']) >= {'0x0000', '0x0010', '0x0800', '0x0810'}
    assert (df['RG'] == 'NA').any() and (df['AO'] != 'NA').any()
    assert df['INDEX'].max() > df.shape[0]                                         # dropped records count in INDEX


def test_empty_sam_file(built, tmp_path):
    sam = tmp_path / 'empty.sam.gz'
    sam.write_bytes(b'')
    bed, head = str(tmp_path / 'e.bed.gz'), str(tmp_path / 'e.headers.gz')
    rules.align_get_read_bed(str(sam), os.path.join(GOLD, 'tig.fa.fai'), 'h1', bed_out=bed, align_head_out=head)
    assert gunzip(bed) == gunzip(os.path.join(GOLD, 'empty.bed.gz'))
    assert os.path.getsize(head) == 0


def test_errors_equal_the_reference(built, tmp_path):
    with open(os.path.join(GOLD, 'errors.json')) as fh:
        cases = json.load(fh)
    fai = read_fai(os.path.join(GOLD, 'tig.fa.fai'))
    assert len(cases) >= 6
    for key, c in cases.items():
        sam = tmp_path / (key + '.sam')
        sam.write_text('@HD\tVN:1.6\n' + c['sam'] + '\n')
        with pytest.raises(Exception) as info:
            get_align_bed(str(sam), fai, 'h1')
        assert type(info.value).__name__ == c['type'], key
        assert str(info.value) == c['message'], key


def test_plain_bgzf_and_min_mapq(built, tmp_path):
    """Container formats and the MAPQ filter: same table from plain text; min_mapq drops rows but keeps INDEX."""
    text = gunzip(os.path.join(GOLD, 'hap_noseq.sam.gz'))
    plain = tmp_path / 'x.sam'
    plain.write_bytes(text.replace(b'\n', b'\r\n'))                               # CRLF line ends are tolerated
    fai = read_fai(os.path.join(GOLD, 'tig.fa.fai'))
    a = get_align_bed(os.path.join(GOLD, 'hap_noseq.sam.gz'), fai, 'h1')
    b = get_align_bed(str(plain), fai, 'h1')
    assert a.equals(b)
    c = get_align_bed(str(plain), fai, 'h1', min_mapq=30)
    assert 0 < c.shape[0] < a.shape[0] and (c['MAPQ'] >= 30).all()
    assert c['INDEX'].tolist() == a.loc[a['MAPQ'] >= 30, 'INDEX'].tolist()
    with pytest.raises(_lib.PavDeviceError, match='malformed CIGAR'):
        bad = tmp_path / 'bad.sam'
        bad.write_text('t\t0\tchr1\t1\t60\t10=5\t*\t0\t0\t*\t*\n')
        get_align_bed(str(bad), fai, 'h1')
