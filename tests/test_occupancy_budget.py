"""The LDS and register budgets that decide how many workgroups of the hot kernels a CU holds, read from the code objects' notes
(tools/occupancy_scan.py; hipcc cross-compiles for gfx950 without a GPU).  Round 6 found `k_kmer_lds` at 40 968 bytes of LDS - three
workgroups a CU where four fit in 40 960 - and three more kernels a few bytes or registers over a boundary; a later edit that adds
a word of shared memory or a few registers would lose a workgroup per CU again without any test noticing.  This one does."""
import os
import shutil
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

# kernel name fragment -> (most LDS bytes, most registers, workgroups a CU that follow)
BUDGET = {
    'cigar': {'walk_emitILi2E': (32768, 96, 5)},                                   # walk_snv
    'density': {'k_kmer_lds': (40960, 56, 4), 'k_compact_scatter': (32768, 96, 5), 'k_bucket_ref': (27306, 80, 6), 'k_bucket_tig': (27306, 80, 6)},
}


@pytest.mark.skipif(shutil.which('hipcc') is None and not os.path.exists('/opt/rocm/bin/hipcc'), reason='needs hipcc')
@pytest.mark.parametrize('source', sorted(BUDGET))
def test_hot_kernels_keep_their_workgroups_per_cu(source):
    import occupancy_scan as occ
    with tempfile.TemporaryDirectory() as tmp:
        kernels = {k['name']: k for k in occ.kernels_of(source, tmp)}
    for frag, (lds_max, regs_max, want) in BUDGET[source].items():
        hit = [k for n, k in kernels.items() if frag in n]
        assert len(hit) == 1, (frag, sorted(kernels))
        k = hit[0]
        regs = (k['vgpr_count'] + k.get('agpr_count', 0) + 7) // 8 * 8
        assert k.get('group_segment_fixed_size', 0) <= lds_max, (frag, k)
        assert regs <= regs_max, (frag, k)
        assert k.get('private_segment_fixed_size', 0) == 0, (frag, 'scratch memory in a hot kernel', k)
        wg = k.get('max_flat_workgroup_size', 256)
        by_regs = min(8, occ.REGS_PER_LANE // regs) * 4 // max(1, wg // 64)
        by_lds = occ.LDS_PER_CU // k['group_segment_fixed_size'] if k.get('group_segment_fixed_size') else 99
        assert min(by_regs, by_lds, occ.LANES_PER_CU // wg) >= want, (frag, by_regs, by_lds)
