"""The bench line contract (driver + judge read it): checked on the committed round profiles, which are verbatim outputs of
bench.py on an MI355X (tools/scripts/profile_round.sh).  No GPU needed."""
import json
import os

import pytest

PROFILES = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles')
BASELINE = os.path.join(os.path.dirname(PROFILES), 'BASELINE.json')


def load(name):
    with open(os.path.join(PROFILES, name)) as fh:
        return json.loads(fh.read().strip().splitlines()[-1])


@pytest.mark.parametrize('name', ['r01_full_path_bench.json', 'r01_cigar_only_bench.json'])
def test_bench_line_has_the_contract_fields(name):
    line = load(name)
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in line, key
    assert line['unit'] == 'Gbp/s' and line['higher_is_better'] is True and line['scaling'] == 'weak' and line['vs_baseline'] is None
    assert line['data'] == 'synthetic' and 'workload' in line['config'] and 'model' not in line['config']
    assert abs(line['value'] - line['config']['aligned_bp_per_gpu'] * line['n_gpus'] / (line['ms_per_step'] * 1e-3) / 1e9) < 0.01 * line['value']
    r = line['roofline']
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] == 'GB/s' and r['peak'] == 8000.0
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
    assert abs(r['achieved'] - r['algorithmic_bytes_per_launch'] / (r['avg_kernel_ms'] * 1e-3) / 1e9) < 0.01 * r['achieved']
    assert r['traffic'] is None or 0.9 < r['traffic'] / r['algorithmic_bytes_per_launch'] < 1.5     # no wasted re-reads
    c = line['cpu_baseline']
    assert c['kind'] in ('port', 'reference') and c['cores'] >= 1 and c['unit'] == 'Gbp/s' and c['sample']
    assert c['records_match_gpu'] is True


def test_headline_is_the_whole_metric_path():
    with open(BASELINE) as fh:
        base = json.load(fh)
    line = load('r01_full_path_bench.json')
    assert line['metric'] == base['metric']
    assert line['inv_scan']['calls'] > 0 and line['inv_scan']['flagging']['planted_inversions_flagged'] > 0
    assert line['cpu_baseline']['density_tables_match_gpu'] is True
    co = line['cigar_only']                                                     # BASELINE configs[1] measured in the same run
    assert co['unit'] == 'Gbp/s' and co['value'] > line['value'] and co['roofline']['kernel'] == 'pack_kernel'
    assert line['value'] >= 50.0                                                # north-star target on one MI355X
