#!/usr/bin/env python3
"""Workgroups per CU of every kernel of the library, from the code objects' notes (no GPU needed):  python tools/occupancy_scan.py [--all]

Each .hip source is compiled for the device only (hipcc --cuda-device-only), unbundled, and `llvm-readelf --notes` gives every
kernel's LDS bytes, registers (VGPR + AGPR), scratch bytes and workgroup size.  Against gfx950's 160 KiB of LDS a CU, 512 registers
a SIMD lane (allocated in eights) and 2 048 lanes a CU that is a number of resident workgroups per limit - and a kernel that is a
few bytes or registers short of one more is printed as a near-miss.  Round 6 found four of those in a pass this way:
k_kmer_lds 8 bytes over four workgroups, walk_snv 208 bytes over five, k_compact_scatter, k_bucket_* two registers over six.  Without --all only near-misses and kernels that use scratch memory are listed."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ['ctx', 'cigar', 'density', 'flag', 'lift_dev', 'tables', 'textdev', 'deflate', 'inflate', 'fastadev', 'trim_dev']
LDS_PER_CU, REGS_PER_LANE, LANES_PER_CU = 163840, 512, 2048
LLVM = '/opt/rocm/lib/llvm/bin'


def kernels_of(name, tmp):
    obj, co = os.path.join(tmp, name + '.o'), os.path.join(tmp, name + '.co')
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '--cuda-device-only', '-c', '-x', 'hip',
                    os.path.join(ROOT, 'pav_amd', 'csrc', name + '.hip'), '-o', obj], check=True, stderr=subprocess.DEVNULL)
    subprocess.run([os.path.join(LLVM, 'clang-offload-bundler'), '--unbundle', '--type=o', '--input=' + obj,
                    '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + co], check=True)
    notes = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', co], capture_output=True, text=True, check=True).stdout
    cur = {}
    for line in notes.splitlines():
        m = re.match(r'\s+\.(\w+):\s+(.*)', line)
        if not m:
            continue
        key, val = m.group(1), m.group(2).strip()
        if key == 'name':
            cur['name'] = val
        elif key in ('group_segment_fixed_size', 'max_flat_workgroup_size', 'vgpr_count', 'agpr_count', 'private_segment_fixed_size'):
            cur[key] = int(val)
        if key == 'vgpr_count' and 'name' in cur:                      # (the last field of a kernel's record)
            if 'rocprim' not in cur['name'] and 'amd_rocclr' not in cur['name']:
                yield cur
            cur = {}


def main():
    show_all = '--all' in sys.argv
    with tempfile.TemporaryDirectory() as tmp:
        for name in FILES:
            for k in kernels_of(name, tmp):
                wg = k.get('max_flat_workgroup_size', 256)
                waves = max(1, wg // 64)
                regs = (k['vgpr_count'] + k.get('agpr_count', 0) + 7) // 8 * 8
                per_simd = min(8, REGS_PER_LANE // max(regs, 8))
                by_regs = per_simd * 4 // waves
                lds = k.get('group_segment_fixed_size', 0)
                by_lds = LDS_PER_CU // lds if lds else 99
                by_lanes = LANES_PER_CU // wg
                limit = min(by_regs, by_lds, by_lanes)
                note = ''
                if lds and by_lds == limit and by_lds < min(by_regs, by_lanes):
                    need = lds - LDS_PER_CU // (by_lds + 1)
                    if need <= 0.08 * lds:
                        note = f'LDS near-miss: {need} bytes fewer -> {by_lds + 1} workgroups'
                if by_regs == limit and by_regs < min(by_lds, by_lanes) and per_simd < 8:
                    target = REGS_PER_LANE // (per_simd + 1) // 8 * 8
                    if regs - target <= 8:
                        note = f'register near-miss: {regs} -> {target}'
                scratch = k.get('private_segment_fixed_size', 0)
                if show_all or note or scratch:
                    short = re.sub(r'^_ZN3pav\d*|^_ZN\d+_GLOBAL__N_1\d+', '', k['name'])[:44]
                    print(f'{name:9s} {short:46s} lanes {wg:4d} regs {regs:3d} lds {lds:6d} scratch {scratch:4d}  workgroups/CU: regs {by_regs} '
                          f'lds {by_lds} lanes {by_lanes} -> {limit}  {note}')


if __name__ == '__main__':
    main()
