"""
ctypes binding of ``libpav_amd.so`` (C ABI: ``include/pav_amd.h``).

This is the binding a PAV maintainer adds.  There is no CPU fallback: if the HIP library is
missing or no gfx950 device is usable, every entry point raises ``PavDeviceError`` - loudly, by design.
"""

import ctypes
import importlib.util
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PAV_AMD_LIB: another build of the same library (tuning builds of tools/bench_variants.py); there is still no CPU path behind it
LIB_PATH = os.environ.get('PAV_AMD_LIB') or os.path.join(_HERE, 'lib', 'libpav_amd.so')

PAV_ROLE_REF, PAV_ROLE_TIG = 0, 1
PAV_OK, PAV_E_ARG, PAV_E_HIP, PAV_E_NODEV, PAV_E_CIGAR, PAV_E_STATE, PAV_E_LIMIT, PAV_E_TRIM = 0, -1, -2, -3, -4, -5, -6, -7


class PavDeviceError(RuntimeError):
    """The HIP library or the GPU is unavailable / failed.  Never swallowed into a CPU path."""


# numpy dtypes mirroring the C structs (include/pav_amd.h)
ALN_DTYPE = np.dtype([('ref_id', '<u4'), ('tig_id', '<u4'), ('pos', '<u4'), ('rev', '<u4')])
SNV_DTYPE = np.dtype([('aln', '<u4'), ('pos', '<u4'), ('qry_pos', '<u4'), ('ref', 'u1'), ('alt', 'u1'),
                      ('pad', '<u2')])
INDEL_DTYPE = np.dtype([('aln', '<u4'), ('op_index', '<u4'), ('pos', '<u4'), ('end', '<u4'), ('svlen', '<u4'),
                        ('qry_pos', '<u4'), ('qry_end', '<u4'), ('left_shift', '<u4'),
                        ('hom_ref_l', '<u4'), ('hom_ref_r', '<u4'), ('hom_tig_l', '<u4'), ('hom_tig_r', '<u4'),
                        ('seq_off', '<u8'), ('svtype', 'u1'), ('pad', 'u1', (7,))])
HOM_QUERY_DTYPE = np.dtype([('role', '<i4'), ('seq_id', '<i4'), ('rev', '<i4'), ('_p0', '<i4'), ('pos', '<i8'),
                            ('sv_role', '<i4'), ('sv_seq_id', '<i4'), ('sv_rev', '<i4'), ('_p1', '<i4'),
                            ('sv_pos', '<i8'), ('svlen', '<u4'), ('dir', '<i4')])
assert SNV_DTYPE.itemsize == 16 and INDEL_DTYPE.itemsize == 64 and ALN_DTYPE.itemsize == 16
assert HOM_QUERY_DTYPE.itemsize == 56


class CigarCounts(ctypes.Structure):
    _fields_ = [('n_ops', ctypes.c_uint64), ('n_snv', ctypes.c_uint64), ('n_indel', ctypes.c_uint64),
                ('seq_bytes', ctypes.c_uint64), ('aligned_bases', ctypes.c_uint64)]


class DenJob(ctypes.Structure):
    _fields_ = [('ref_id', ctypes.c_uint32), ('tig_id', ctypes.c_uint32), ('ref_pos', ctypes.c_uint64),
                ('ref_end', ctypes.c_uint64), ('tig_pos', ctypes.c_uint64), ('tig_end', ctypes.c_uint64),
                ('ref_rc', ctypes.c_uint32), ('state_run_smooth', ctypes.c_uint32)]


class DenParams(ctypes.Structure):
    _fields_ = [('k', ctypes.c_int32), ('min_informative', ctypes.c_uint32), ('min_state_count', ctypes.c_uint32),
                ('den_smooth', ctypes.c_double), ('state_run_delta', ctypes.c_double),
                ('max_ref_kmer_count', ctypes.c_uint32), ('kde_mode', ctypes.c_uint32), ('kmer_mode', ctypes.c_uint32),
                ('guard_cap', ctypes.c_uint32), ('guard_rel', ctypes.c_double)]


class DenResult(ctypes.Structure):
    _fields_ = [('status', ctypes.c_int32), ('fail_kind', ctypes.c_int32), ('n_rows', ctypes.c_uint32),
                ('n_runs', ctypes.c_uint32), ('max_count', ctypes.c_uint32), ('n_sample', ctypes.c_uint32),
                ('max_kmer', ctypes.c_uint64), ('state_count', ctypes.c_uint32 * 3), ('n_near_tie', ctypes.c_uint32),
                ('n_eval', ctypes.c_uint64), ('h', ctypes.c_double * 3), ('n_reeval', ctypes.c_uint32),
                ('n_unresolved', ctypes.c_uint32), ('n_spike_near', ctypes.c_uint32), ('guard_fallback', ctypes.c_uint32)]


class InvAln(ctypes.Structure):
    _fields_ = [('ref_id', ctypes.c_uint32), ('tig_id', ctypes.c_uint32), ('pos', ctypes.c_uint64), ('end', ctypes.c_uint64),
                ('qry_pos', ctypes.c_uint64), ('qry_end', ctypes.c_uint64), ('rev', ctypes.c_uint32), ('pad', ctypes.c_uint32),
                ('index', ctypes.c_int64)]


INV_ALN_DTYPE = np.dtype([('ref_id', '<u4'), ('tig_id', '<u4'), ('pos', '<u8'), ('end', '<u8'), ('qry_pos', '<u8'),
                          ('qry_end', '<u8'), ('rev', '<u4'), ('pad', '<u4'), ('index', '<i8')])
INV_REGION_DTYPE = np.dtype([('ref_id', '<u4'), ('pad', '<u4'), ('pos', '<u8'), ('end', '<u8')])
assert INV_ALN_DTYPE.itemsize == 56 and INV_REGION_DTYPE.itemsize == 24


class Srs(ctypes.Structure):
    _fields_ = [('begin', ctypes.c_double), ('end', ctypes.c_double), ('value', ctypes.c_uint32), ('pad', ctypes.c_uint32)]


class InvParams(ctypes.Structure):
    _fields_ = [('max_region_size', ctypes.c_int64), ('min_exp_count', ctypes.c_int32), ('n_srs', ctypes.c_uint32),
                ('srs', ctypes.POINTER(Srs)), ('den', DenParams), ('lazy_tables', ctypes.c_uint32), ('reserved', ctypes.c_uint32)]


class InvRgn(ctypes.Structure):
    _fields_ = [('seq_id', ctypes.c_uint32), ('is_rev', ctypes.c_uint32), ('pos', ctypes.c_uint64), ('end', ctypes.c_uint64),
                ('n_aln', ctypes.c_uint32 * 2), ('aln_index', (ctypes.c_int64 * 2) * 2)]


class InvResult(ctypes.Structure):
    _fields_ = [('outcome', ctypes.c_int32), ('found', ctypes.c_uint32), ('iterations', ctypes.c_uint32),
                ('n_rows', ctypes.c_uint32), ('svlen', ctypes.c_uint64),
                ('ref_outer', InvRgn), ('ref_inner', InvRgn), ('tig_outer', InvRgn), ('tig_inner', InvRgn),
                ('ref_discovery', InvRgn), ('tig_discovery', InvRgn), ('log_bytes', ctypes.c_uint32),
                ('error_bytes', ctypes.c_uint32), ('n_near_tie', ctypes.c_uint32), ('n_unresolved', ctypes.c_uint32)]


INV_NONE, INV_CALL, INV_ERROR = 0, 1, 2
_INV_RGN_DTYPE = np.dtype([('seq_id', '<u4'), ('is_rev', '<u4'), ('pos', '<u8'), ('end', '<u8'), ('n_aln', '<u4', (2,)),
                           ('aln_index', '<i8', (2, 2))])
INV_RESULT_DTYPE = np.dtype([('outcome', '<i4'), ('found', '<u4'), ('iterations', '<u4'), ('n_rows', '<u4'), ('svlen', '<u8'),
                             ('ref_outer', _INV_RGN_DTYPE), ('ref_inner', _INV_RGN_DTYPE), ('tig_outer', _INV_RGN_DTYPE),
                             ('tig_inner', _INV_RGN_DTYPE), ('ref_discovery', _INV_RGN_DTYPE), ('tig_discovery', _INV_RGN_DTYPE),
                             ('log_bytes', '<u4'), ('error_bytes', '<u4'), ('n_near_tie', '<u4'), ('n_unresolved', '<u4')])
assert INV_RESULT_DTYPE.itemsize == ctypes.sizeof(InvResult)
RUN_DTYPE = np.dtype([('state', '<i4'), ('count', '<u4'), ('pos', '<i8'), ('end', '<i8')])
DEN_OK, DEN_UNFINALISED, DEN_FAIL = 0, 1, 125
KDE_RUNS, KDE_DIRECT = 0, 1
KMER_LDS, KMER_HBM = 0, 1


class TableOpts(ctypes.Structure):
    _fields_ = [('hap', ctypes.c_char_p), ('align_index', ctypes.c_void_p), ('trim_pos', ctypes.c_void_p),
                ('trim_end', ctypes.c_void_p), ('snv_path', ctypes.c_char_p), ('insdel_path', ctypes.c_char_p),
                ('gzip_level', ctypes.c_int32), ('threads', ctypes.c_int32), ('call_batch', ctypes.c_void_p)]


FLAG_RGN_DTYPE = np.dtype([('chrom', '<u4'), ('pad', '<u4'), ('pos', '<i8'), ('end', '<i8'), ('count', '<i8')])
FLAG_LOCUS_DTYPE = np.dtype([('chrom', '<u4'), ('type_mask', '<u4'), ('pos', '<i8'), ('end', '<i8'), ('count_indel', '<i8'),
                             ('count_snv', '<i8'), ('try_inv', '<i4'), ('batch', '<i4')])
assert FLAG_RGN_DTYPE.itemsize == 32 and FLAG_LOCUS_DTYPE.itemsize == 48
FLAG_MATCH_SV, FLAG_MATCH_INDEL, FLAG_CLUSTER_INDEL, FLAG_CLUSTER_SNV = 1, 2, 4, 8
SIG_SVINDEL, SIG_SV, SIG_SINGLE_CLUSTER, SIG_NONE = 0, 1, 2, 3
FLAG_TABLES = ('insdel_sv', 'insdel_indel', 'cluster_indel', 'cluster_snv')          # order of pav_flag_merge_loci


TRIM_ROW_DTYPE = np.dtype([('chrom', '<u4'), ('qry_id', '<u4'), ('pos', '<i8'), ('end', '<i8'), ('qry_pos', '<i8'), ('qry_end', '<i8'),
                           ('index', '<i8'), ('rev', '<i4'), ('modified', '<i4'), ('trim_ref_l', '<i8'), ('trim_ref_r', '<i8'),
                           ('trim_qry_l', '<i8'), ('trim_qry_r', '<i8')])
TRIM_COUNT_DTYPE = np.dtype([('ref_bp', '<i8'), ('tig_bp', '<i8'), ('clip_h_l', '<i8'), ('clip_s_l', '<i8'), ('clip_h_r', '<i8'),
                             ('clip_s_r', '<i8'), ('err_kind', '<i4'), ('err_op', '<u4'), ('err_len', '<u8'), ('err_char', '<u4'),
                             ('pad', '<u4')])
assert TRIM_ROW_DTYPE.itemsize == 88 and TRIM_COUNT_DTYPE.itemsize == 72
TRIM_QUERY, TRIM_SUBJECT = 0, 1


class BedInfo(ctypes.Structure):
    _fields_ = [('n_rows', ctypes.c_uint64), ('cigar_bytes', ctypes.c_uint64), ('n_chrom', ctypes.c_uint32), ('n_qry', ctypes.c_uint32),
                ('columns', ctypes.c_uint32), ('pad', ctypes.c_uint32)]


class BedCols(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ('chrom_id', 'qry_id', 'pos', 'end', 'index', 'qry_pos', 'qry_end', 'qry_len', 'mapq',
                                               'call_batch', 'rev', 'cigar_text', 'cigar_off')]


BED_COLUMNS = ('#CHROM', 'POS', 'END', 'INDEX', 'QRY_ID', 'QRY_POS', 'QRY_END', 'QRY_LEN', 'MAPQ', 'REV', 'CIGAR', 'CALL_BATCH')


class TrimErr(ctypes.Structure):
    _fields_ = [('kind', ctypes.c_int32), ('row_l', ctypes.c_uint32), ('row_r', ctypes.c_uint32), ('op_index', ctypes.c_uint32),
                ('op_char', ctypes.c_uint32), ('side', ctypes.c_int32), ('diff_bp', ctypes.c_int64), ('op_len', ctypes.c_uint64)]


class FlagParams(ctypes.Structure):
    _fields_ = [('cluster_win', ctypes.c_int64), ('cluster_min_snv', ctypes.c_int64), ('cluster_min_indel', ctypes.c_int64),
                ('insdel_flank_cluster', ctypes.c_int64), ('insdel_flank_merge', ctypes.c_int64),
                ('insdel_min_svlen', ctypes.c_int64), ('merge_flank', ctypes.c_int64), ('batch_count', ctypes.c_int32),
                ('sig_filter', ctypes.c_int32)]


class FlagResult(ctypes.Structure):
    _fields_ = [('tables', ctypes.c_void_p * 4), ('n', ctypes.c_uint64 * 4), ('loci', ctypes.c_void_p), ('n_loci', ctypes.c_uint64),
                ('n_snv_pass', ctypes.c_uint64), ('n_indel_pass', ctypes.c_uint64)]


class CigarErr(ctypes.Structure):
    _fields_ = [('kind', ctypes.c_int32), ('aln', ctypes.c_uint32), ('op_index', ctypes.c_uint32),
                ('op_char', ctypes.c_uint32), ('pos_ref', ctypes.c_uint32), ('pos_tig', ctypes.c_uint32)]


# Every symbol include/pav_amd.h declares: (restype, argtypes).  tests/test_abi.py checks the header against this.
_P = ctypes.c_void_p
SYMBOLS = {
    'pav_abi_version': (ctypes.c_int, []),
    'pav_device_count': (ctypes.c_int, []),
    'pav_create': (_P, [ctypes.c_int]),
    'pav_destroy': (None, [_P]),
    'pav_last_error': (ctypes.c_char_p, [_P]),
    'pav_device_name': (ctypes.c_int, [_P, ctypes.c_char_p, ctypes.c_int]),
    'pav_seq_load_fasta_path': (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_uint32)]),
    'pav_seq_fetch': (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]),
    'pav_seq_fetch_many': (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    'pav_seq_name': (ctypes.c_char_p, [_P, ctypes.c_int, ctypes.c_uint32]),
    'pav_seq_length': (ctypes.c_uint64, [_P, ctypes.c_int, ctypes.c_uint32]),
    'pav_gzip_buffer': (ctypes.c_int, [_P, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64,
                                       ctypes.POINTER(ctypes.c_uint64)]),
    'pav_bgzf_inflate': (ctypes.c_int, [_P, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]),
    'pav_gzip_buffers': (ctypes.c_int, [_P, ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64), ctypes.c_int,
                                        ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    'pav_device_pci_bus_id': (ctypes.c_int, [_P, ctypes.c_char_p, ctypes.c_int]),
    'pav_sync': (ctypes.c_int, [_P]),
    'pav_mem_info': (ctypes.c_int, [_P, _P, _P]),
    'pav_kde_work': (ctypes.c_int, [_P, _P]),
    'pav_wait_stats': (ctypes.c_int, [_P]),
    'pav_device_pool_trim': (ctypes.c_uint64, [ctypes.c_int]),
    'pav_cigar_verify': (ctypes.c_int, [_P, _P]),
    'pav_seq_load': (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_uint32, _P, _P]),
    'pav_seq_share': (ctypes.c_int, [_P, _P, ctypes.c_int]),
    'pav_seq_pack': (ctypes.c_int, [_P, ctypes.c_int]),
    'pav_seq_count': (ctypes.c_int, [_P, ctypes.c_int, _P, _P]),
    'pav_cigar_load': (ctypes.c_int, [_P, ctypes.c_uint32, _P, _P, _P]),
    'pav_cigar_call': (ctypes.c_int, [_P, _P]),
    'pav_cigar_error': (ctypes.c_int, [_P, _P]),
    'pav_cigar_fetch': (ctypes.c_int, [_P, _P, _P, _P]),
    'pav_cigar_fetch_ops': (ctypes.c_int, [_P, _P, _P]),
    'pav_cigar_write_tables': (ctypes.c_int, [_P, _P, _P, _P]),
    'pav_cigar_write_tables_begin': (ctypes.c_int, [_P, _P]),
    'pav_cigar_write_tables_end': (ctypes.c_int, [_P, _P, _P]),
    'pav_align_index': (ctypes.c_int, [_P, ctypes.c_uint32, _P, _P, _P, _P, _P, _P, _P, _P]),
    'pav_homology': (ctypes.c_int, [_P, ctypes.c_uint32, _P, _P]),
    'pav_density_batch': (ctypes.c_int, [_P, ctypes.c_uint32, _P, _P, _P]),
    'pav_density_runs': (ctypes.c_int, [_P, ctypes.c_uint32, _P]),
    'pav_density_table': (ctypes.c_int, [_P, ctypes.c_uint32, _P, _P, _P, _P, _P, _P, _P]),
    'pav_density_annotate': (ctypes.c_int, [_P, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint64,
                                            ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int64, ctypes.c_int64,
                                            ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, _P, _P]),
    'pav_seq_set_names': (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_uint32, _P]),
    'pav_inv_load_alignments': (ctypes.c_int, [_P, ctypes.c_uint32, _P, _P, _P]),
    'pav_inv_scan_batch': (ctypes.c_int, [_P, ctypes.c_uint32, _P, _P, _P]),
    'pav_inv_text': (ctypes.c_int, [_P, ctypes.c_uint32, ctypes.c_int, ctypes.c_char_p, ctypes.c_uint32]),
    'pav_inv_texts': (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_char_p, ctypes.c_uint64, _P]),
    'pav_inv_table': (ctypes.c_int, [_P, ctypes.c_uint32, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'pav_inv_table_view': (ctypes.c_int, [_P, ctypes.c_uint32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'pav_inv_tables': (ctypes.c_int, [_P, ctypes.c_uint32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'pav_inv_write_tables': (ctypes.c_int, [_P, ctypes.c_uint32, _P, _P, ctypes.c_int, ctypes.c_int]),
    'pav_repr_f64': (ctypes.c_int, [ctypes.c_double, ctypes.c_char_p, ctypes.c_int]),
    'pav_kmer_rev_complement': (ctypes.c_uint64, [ctypes.c_uint64, ctypes.c_int]),
    'pav_kmer_canonical': (ctypes.c_uint64, [ctypes.c_uint64, ctypes.c_int]),
    'pav_fasta_open': (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, _P]),
    'pav_fasta_close': (None, [_P]),
    'pav_fasta_count': (ctypes.c_uint32, [_P]),
    'pav_fasta_name': (ctypes.c_char_p, [_P, ctypes.c_uint32]),
    'pav_fasta_length': (ctypes.c_uint64, [_P, ctypes.c_uint32]),
    'pav_fasta_seq': (_P, [_P, ctypes.c_uint32]),
    'pav_fasta_kind': (ctypes.c_int, [_P]),
    'pav_seq_load_fasta': (ctypes.c_int, [_P, ctypes.c_int, _P, ctypes.c_uint32, _P]),
    'pav_sam_open': (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, _P]),
    'pav_sam_close': (None, [_P]),
    'pav_sam_info': (ctypes.c_int, [_P, _P]),
    'pav_sam_fetch': (ctypes.c_int, [_P, _P]),
    'pav_sam_name': (ctypes.c_char_p, [_P, ctypes.c_int, ctypes.c_uint32]),
    'pav_sam_header': (ctypes.c_int, [_P, _P]),
    'pav_bed_open': (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, _P]),
    'pav_bed_close': (None, [_P]),
    'pav_bed_info': (ctypes.c_int, [_P, _P]),
    'pav_bed_fetch': (ctypes.c_int, [_P, _P]),
    'pav_bed_name': (ctypes.c_char_p, [_P, ctypes.c_int, ctypes.c_uint32]),
    'pav_cigar_load_bed': (ctypes.c_int, [_P, _P, ctypes.c_int64, _P, _P]),
    'pav_trim_load': (ctypes.c_int, [_P, ctypes.c_uint32, _P, _P, _P]),
    'pav_trim_pass': (ctypes.c_int, [_P, ctypes.c_uint32, _P, ctypes.c_int, ctypes.c_int64, ctypes.c_int]),
    'pav_trim_pair': (ctypes.c_int, [_P, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    'pav_trim_error': (ctypes.c_int, [_P, _P]),
    'pav_trim_fetch': (ctypes.c_int, [_P, _P, _P, _P]),
    'pav_trim_fetch_cigar': (ctypes.c_int, [_P, _P, _P]),
    'pav_flag_params_default': (None, [_P]),
    'pav_flag_cluster': (ctypes.c_int, [_P, ctypes.c_uint64, _P, _P, _P, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, _P, _P]),
    'pav_flag_insdel': (ctypes.c_int, [_P, ctypes.c_uint64, _P, _P, _P, ctypes.c_uint64, _P, _P, _P, ctypes.c_int64,
                                       ctypes.c_int64, _P, _P]),
    'pav_flag_merge_loci': (ctypes.c_int, [_P, _P, _P, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, _P, _P]),
    'pav_cigar_flag': (ctypes.c_int, [_P, _P, _P, _P, _P]),
    'pav_prof_enable': (ctypes.c_int, [_P, ctypes.c_int]),
    'pav_prof_reset': (ctypes.c_int, [_P]),
    'pav_prof_count': (ctypes.c_int, [_P]),
    'pav_prof_get': (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, _P, _P]),
}

_LIB = None


def _preload_hip_runtime():
    """If PyTorch-ROCm is installed, make its bundled HIP runtime the process-wide one *before* libpav_amd.so
    resolves ``libamdhip64.so.7``: bench.py uses torch.distributed in the same process and two HIP runtimes in
    one process must be avoided.  Uses find_spec only (no ``import torch``)."""
    try:
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], 'lib', 'libamdhip64.so')
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def source_fingerprint():
    """What the library at LIB_PATH was built from, for the provenance stamp of committed profiles (tools/prof_summary.py writes
    it, bench.py compares it): sha256 (16 hex digits) of every file under csrc/ and of include/pav_amd.h, the kernels each file
    defines, the build flags of __graft_entry__.build_hip, and the git commit when the tree has one (the GPU box's copy has not)."""
    import hashlib
    import re
    import subprocess
    csrc = os.path.join(_HERE, 'csrc')
    files, kernels = {}, {}
    for path in sorted([os.path.join(csrc, f) for f in os.listdir(csrc)] + [os.path.join(os.path.dirname(_HERE), 'include', 'pav_amd.h')]):
        if not os.path.isfile(path):
            continue
        with open(path, 'rb') as fh:
            data = fh.read()
        name = os.path.basename(path)
        files[name] = hashlib.sha256(data).hexdigest()[:16]
        for k in re.findall(rb'__global__(?:\s+__launch_bounds__\s*\([^)]*\))?\s+void\s+(\w+)\s*\(', data):
            kernels[k.decode()] = name
        for k in re.findall(rb'PAV_LAUNCH(?:_ON)?\s*\([^"\n]*"(\w+)"', data):    # launch labels (walk_snv = walk_emit<WALK_SNV>)
            kernels.setdefault(k.decode(), name)
    commit = None
    try:
        r = subprocess.run(['git', '-C', os.path.dirname(_HERE), 'rev-parse', 'HEAD'], capture_output=True, text=True, timeout=10)
        if r.returncode == 0:
            commit = r.stdout.strip()
            d = subprocess.run(['git', '-C', os.path.dirname(_HERE), 'status', '--porcelain', '--', 'pav_amd/csrc', 'include'],
                               capture_output=True, text=True, timeout=10)
            if d.stdout.strip():
                commit += '+dirty'
    except (OSError, subprocess.SubprocessError):
        pass
    whole = hashlib.sha256(''.join(f'{k}:{v};' for k, v in sorted(files.items())).encode()).hexdigest()[:16]
    return {'library_source_sha16': whole, 'files': files, 'kernel_file': kernels, 'commit': commit,
            'build_flags': '--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off'}


def kernel_sources_match(stamp, kernel):
    """True when the committed profile `stamp` (a source_fingerprint() of the run it was taken on) was taken on the same source
    of `kernel` as the library here: the file that defines it, common.h and the public header are unchanged."""
    if not stamp or 'files' not in stamp:
        return False
    now = source_fingerprint()
    f = now['kernel_file'].get(kernel.split('<')[0])
    if f is None:
        return False
    return all(stamp['files'].get(x) == now['files'].get(x) for x in (f, 'common.h', 'pav_amd.h'))


def load():
    """Load libpav_amd.so and declare every prototype.  Raises PavDeviceError if it is not built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise PavDeviceError(
            f'{LIB_PATH} is not built (run: python -c "import __graft_entry__ as g; g.build()"). '
            'pav_amd has no CPU fallback.')
    _preload_hip_runtime()
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as ex:
        raise PavDeviceError(f'cannot load {LIB_PATH}: {ex}') from ex
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _LIB = lib
    return lib


def _ptr(arr):
    return ctypes.c_void_p(arr.ctypes.data) if arr is not None and arr.size else None


class Context:
    """One GPU context (``pav_ctx``).  Use as a context manager or call :meth:`close`."""

    def __init__(self, device_id=0):
        self.lib = load()
        self.handle = self.lib.pav_create(int(device_id))
        if not self.handle:
            msg = self.lib.pav_last_error(None)
            raise PavDeviceError('pav_create(%d) failed: %s' % (device_id, msg.decode() if msg else 'unknown'))
        self.device_id = int(device_id)
        self._seq_names = {PAV_ROLE_REF: [], PAV_ROLE_TIG: []}

    # -- lifecycle ----------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, 'handle', None):
            self.lib.pav_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc == PAV_OK:
            return
        msg = self.lib.pav_last_error(self.handle)
        msg = msg.decode() if msg else ''
        if rc == PAV_E_CIGAR:
            raise CigarDeviceError(msg)
        raise PavDeviceError(f'{what} failed ({rc}): {msg}')

    @property
    def device_name(self):
        buf = ctypes.create_string_buffer(256)
        self._check(self.lib.pav_device_name(self.handle, buf, 256), 'pav_device_name')
        return buf.value.decode()

    @property
    def pci_bus_id(self):
        buf = ctypes.create_string_buffer(64)
        self._check(self.lib.pav_device_pci_bus_id(self.handle, buf, 64), 'pav_device_pci_bus_id')
        return buf.value.decode()

    def gzip_buffer(self, data, level=0):
        """gzip of ``data`` (bytes / uint8 array) made on the device; returns bytes (a complete gzip file)."""
        buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
        n = int(buf.shape[0])
        out = np.empty(n + n // 2 + 4096, dtype=np.uint8)
        got = ctypes.c_uint64(0)
        self._check(self.lib.pav_gzip_buffer(self.handle, buf.ctypes.data if n else None, n, int(level), out.ctypes.data, out.shape[0],
                                             ctypes.byref(got)), 'pav_gzip_buffer')
        return out[:got.value].tobytes()

    def bgzf_inflate(self, data):
        """The text of a BGZF file held in memory (bytes / uint8 array), its members inflated on the device and checked against
        their CRC-32 and ISIZE; returns bytes.  A corrupt member raises, naming it (``pav_bgzf_inflate``)."""
        buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
        n = int(buf.shape[0])
        # the text's length: the sum of the members' ISIZE, read from the footers here when every header is bgzip's own 18 bytes;
        # otherwise asked of the library with no room for the text (PAV_E_LIMIT answers with the length)
        got = ctypes.c_uint64(0)
        at, total, plain = 0, 0, True
        while at < n:
            if at + 28 > n or buf[at] != 0x1f or buf[at + 1] != 0x8b or buf[at + 10] != 6 or buf[at + 11] != 0 or buf[at + 12] != 66 or buf[at + 13] != 67:
                plain = False
                break
            bsize = int(buf[at + 16]) + (int(buf[at + 17]) << 8) + 1
            if at + bsize > n:
                plain = False
                break
            total += int(buf[at + bsize - 4]) | int(buf[at + bsize - 3]) << 8 | int(buf[at + bsize - 2]) << 16 | int(buf[at + bsize - 1]) << 24
            at += bsize
        if plain:
            got.value = total
            if total == 0 and n == 0:
                return b''
        else:
            rc = self.lib.pav_bgzf_inflate(self.handle, buf.ctypes.data if n else None, n, None, 0, ctypes.byref(got))
            if rc == 0 and got.value == 0:
                return b''
            if got.value == 0:
                self._check(rc, 'pav_bgzf_inflate')
        out = np.empty(max(1, int(got.value)), dtype=np.uint8)
        self._check(self.lib.pav_bgzf_inflate(self.handle, buf.ctypes.data, n, out.ctypes.data, out.shape[0], ctypes.byref(got)), 'pav_bgzf_inflate')
        return out[:got.value].tobytes()

    def gzip_buffers(self, texts, level=0):
        """gzip of every item of ``texts`` (bytes), all in one launch set on the device; returns a list of bytes."""
        n = len(texts)
        if n == 0:
            return []
        bufs = [np.frombuffer(t, dtype=np.uint8) for t in texts]
        ptrs = (ctypes.c_void_p * n)(*[b.ctypes.data if b.shape[0] else None for b in bufs])
        lens = (ctypes.c_uint64 * n)(*[int(b.shape[0]) for b in bufs])
        out = np.empty(sum(int(b.shape[0]) + int(b.shape[0]) // 2 + 4096 for b in bufs), dtype=np.uint8)
        off, got = (ctypes.c_uint64 * n)(), (ctypes.c_uint64 * n)()
        self._check(self.lib.pav_gzip_buffers(self.handle, n, ptrs, lens, int(level), out.ctypes.data, out.shape[0], off, got), 'pav_gzip_buffers')
        return [out[int(off[i]):int(off[i]) + int(got[i])].tobytes() for i in range(n)]

    def sync(self):
        self._check(self.lib.pav_sync(self.handle), 'pav_sync')

    def mem_info(self):
        """(free, total) bytes of HBM on the context's GPU."""
        f, t = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._check(self.lib.pav_mem_info(self.handle, ctypes.byref(f), ctypes.byref(t)), 'pav_mem_info')
        return int(f.value), int(t.value)

    def kde_work(self):
        """Cumulative density work of the context: (evaluation points, (point, run) pairs, (point, data point) pairs)."""
        out = (ctypes.c_double * 3)()
        self._check(self.lib.pav_kde_work(self.handle, out), 'pav_kde_work')
        return float(out[0]), float(out[1]), float(out[2])

    def wait_stats(self):
        """(seconds the CALLING thread has spent in the library's host waits, number of waits) since the thread started."""
        out = (ctypes.c_double * 2)()
        self._check(self.lib.pav_wait_stats(out), 'pav_wait_stats')
        return float(out[0]), int(out[1])

    # -- sequences ----------------------------------------------------------------------------------------
    def seq_load(self, role, names, arrays):
        """Upload and pack all records of one role.  ``arrays``: list of contiguous uint8 ASCII arrays."""
        arrays = [np.ascontiguousarray(a, dtype=np.uint8) for a in arrays]
        n = len(arrays)
        ptrs = (ctypes.c_void_p * max(n, 1))(*[a.ctypes.data for a in arrays])
        lens = (ctypes.c_uint64 * max(n, 1))(*[a.shape[0] for a in arrays])
        self._check(self.lib.pav_seq_load(self.handle, role, n, ptrs, lens), 'pav_seq_load')
        if role == PAV_ROLE_REF:
            self._ref_resident = None                     # (cigarcall.load_reference marks a whole reference file as resident)
        self._seq_names[role] = [str(x) for x in names]
        cnames = (ctypes.c_char_p * max(n, 1))(*[x.encode() for x in self._seq_names[role]])
        self._check(self.lib.pav_seq_set_names(self.handle, role, n, cnames), 'pav_seq_set_names')

    def seq_share(self, other, role=PAV_ROLE_REF):
        """Read ``other``'s resident records of ``role`` (same GPU) instead of holding a copy: one context per haplotype, one
        reference for all of them (``pav_seq_share``)."""
        self._check(self.lib.pav_seq_share(self.handle, other.handle, role), 'pav_seq_share')
        if role == PAV_ROLE_REF:
            self._ref_resident = getattr(other, '_ref_resident', None)
        self._seq_names[role] = list(other._seq_names[role])
        n = len(self._seq_names[role])
        cnames = (ctypes.c_char_p * max(n, 1))(*[x.encode() for x in self._seq_names[role]])
        self._check(self.lib.pav_seq_set_names(self.handle, role, n, cnames), 'pav_seq_set_names')

    def seq_load_fasta(self, role, fasta, records):
        """Upload records (indexes into ``fasta``, a :class:`FastaFile`) straight from the library's parse buffers."""
        rec = np.ascontiguousarray(records, dtype=np.uint32)
        self._check(self.lib.pav_seq_load_fasta(self.handle, role, fasta.handle, rec.shape[0], _ptr(rec)), 'pav_seq_load_fasta')
        if role == PAV_ROLE_REF:
            self._ref_resident = None
        self._seq_names[role] = [fasta.names[int(i)] for i in rec]

    def seq_load_fasta_path(self, role, path, threads=0):
        """Every record of the FASTA file ``path`` into the store of ``role`` without a host-side parse (``pav_seq_load_fasta_path``:
        the file is uploaded as it is - plain text, or bgzipped as PAV keeps its FASTA files, the BGZF members then inflated on the
        device and checked against their CRC-32 - header lines and line breaks are removed on the device).  Returns the record names."""
        n = ctypes.c_uint32(0)
        self._check(self.lib.pav_seq_load_fasta_path(self.handle, role, str(path).encode(), int(threads), ctypes.byref(n)), 'pav_seq_load_fasta_path')
        if role == PAV_ROLE_REF:
            self._ref_resident = None
        self._seq_names[role] = [self.lib.pav_seq_name(self.handle, role, i).decode() for i in range(int(n.value))]
        return self._seq_names[role]

    def seq_lengths(self, role):
        return [int(self.lib.pav_seq_length(self.handle, role, i)) for i in range(len(self._seq_names[role]))]

    def seq_fetch(self, role, rec, pos, end):
        """ASCII bytes [pos, end) of resident record number ``rec`` (uint8 array; forward strand, case preserved)."""
        n = max(0, int(end) - int(pos))
        out = np.empty(n, dtype=np.uint8)
        self._check(self.lib.pav_seq_fetch(self.handle, role, int(rec), int(pos), n, out.ctypes.data if n else None), 'pav_seq_fetch')
        return out

    def seq_fetch_many(self, role, slices):
        """``slices``: (record number, pos, end) triples -> list of uint8 arrays, one round trip for all of them."""
        n = len(slices)
        if n == 0:
            return []
        rec = np.ascontiguousarray([x[0] for x in slices], dtype=np.uint32)
        pos = np.ascontiguousarray([x[1] for x in slices], dtype=np.uint64)
        ln = np.ascontiguousarray([max(0, int(x[2]) - int(x[1])) for x in slices], dtype=np.uint64)
        out = np.empty(int(ln.sum()), dtype=np.uint8)
        self._check(self.lib.pav_seq_fetch_many(self.handle, role, n, _ptr(rec), _ptr(pos), _ptr(ln), out.ctypes.data if out.shape[0] else None),
                    'pav_seq_fetch_many')
        ends = np.cumsum(ln)
        return [out[int(e - k):int(e)] for e, k in zip(ends, ln)]

    def seq_pack(self, role):
        self._check(self.lib.pav_seq_pack(self.handle, role), 'pav_seq_pack')

    def seq_names(self, role):
        return self._seq_names[role]

    # -- CIGAR --------------------------------------------------------------------------------------------
    def cigar_load(self, aln, cigar_text, cigar_off):
        aln = np.ascontiguousarray(aln, dtype=ALN_DTYPE)
        cigar_text = np.ascontiguousarray(cigar_text, dtype=np.uint8)
        cigar_off = np.ascontiguousarray(cigar_off, dtype=np.uint64)
        if cigar_off.shape[0] != aln.shape[0] + 1:
            raise ValueError('cigar_off must have n_aln + 1 entries')
        self._check(self.lib.pav_cigar_load(self.handle, aln.shape[0], _ptr(aln), _ptr(cigar_text),
                                            ctypes.c_void_p(cigar_off.ctypes.data)), 'pav_cigar_load')

    def cigar_load_bed(self, bed, call_batch=-1):
        """pav_cigar_load from a parsed table (rows of one CALL_BATCH, all when negative): -> INDEX of the loaded rows."""
        n = ctypes.c_uint32(0)
        index = np.zeros(max(1, bed.n_rows), dtype=np.int64)
        self._check(self.lib.pav_cigar_load_bed(self.handle, bed.handle, int(call_batch), ctypes.byref(n), _ptr(index)),
                    'pav_cigar_load_bed')
        return index[:int(n.value)].copy()

    def cigar_call(self):
        c = CigarCounts()
        rc = self.lib.pav_cigar_call(self.handle, ctypes.byref(c))
        if rc == PAV_E_CIGAR:
            e = CigarErr()
            self.lib.pav_cigar_error(self.handle, ctypes.byref(e))
            raise CigarDeviceError(self.lib.pav_last_error(self.handle).decode(), e)
        self._check(rc, 'pav_cigar_call')
        return c

    def cigar_fetch(self, counts):
        snv = np.empty(counts.n_snv, dtype=SNV_DTYPE)
        indel = np.empty(counts.n_indel, dtype=INDEL_DTYPE)
        blob = np.empty(counts.seq_bytes, dtype=np.uint8)
        self._check(self.lib.pav_cigar_fetch(self.handle, _ptr(snv), _ptr(indel), _ptr(blob)), 'pav_cigar_fetch')
        return snv, indel, blob

    def cigar_write_tables(self, hap, align_index, trim_pos=None, trim_end=None, snv_path=None, insdel_path=None,
                           gzip_level=0, threads=0, call_batch=None, background=False):
        """Write the SNV / INS-DEL tables of the last cigar_call natively (sorted, FILTER, pandas-identical text).
        ``call_batch`` (CALL_BATCH of every alignment row): write the merged tables of rule call_cigar_merge instead.
        ``background``: return when the device part is done (the text and the gzip members are made on a thread of the
        library's own); :meth:`cigar_write_wait` waits for the files and returns the two row counts."""
        cb = None if call_batch is None else np.ascontiguousarray(call_batch, dtype=np.int64)
        align_index = np.ascontiguousarray(align_index, dtype=np.int64)
        tp = None if trim_pos is None else np.ascontiguousarray(trim_pos, dtype=np.int64)
        te = None if trim_end is None else np.ascontiguousarray(trim_end, dtype=np.int64)
        opts = TableOpts(str(hap).encode(), align_index.ctypes.data, None if tp is None else tp.ctypes.data,
                         None if te is None else te.ctypes.data, None if snv_path is None else str(snv_path).encode(),
                         None if insdel_path is None else str(insdel_path).encode(), int(gzip_level), int(threads),
                         None if cb is None else cb.ctypes.data)
        if background:
            self._check(self.lib.pav_cigar_write_tables_begin(self.handle, ctypes.byref(opts)), 'pav_cigar_write_tables_begin')
            return None
        n1, n2 = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._check(self.lib.pav_cigar_write_tables(self.handle, ctypes.byref(opts), ctypes.byref(n1), ctypes.byref(n2)),
                    'pav_cigar_write_tables')
        return int(n1.value), int(n2.value)

    def cigar_write_wait(self):
        """Wait for a write begun with ``cigar_write_tables(..., background=True)``; -> (SNV rows, INS / DEL rows)."""
        n1, n2 = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._check(self.lib.pav_cigar_write_tables_end(self.handle, ctypes.byref(n1), ctypes.byref(n2)), 'pav_cigar_write_tables_end')
        return int(n1.value), int(n2.value)

    # ---- alignment trimming ---------------------------------------------------------------------------------------
    def trim_load(self, rows, cigar_text, cigar_off):
        rows = np.ascontiguousarray(rows, dtype=TRIM_ROW_DTYPE)
        cigar_text = np.ascontiguousarray(cigar_text, dtype=np.uint8)
        cigar_off = np.ascontiguousarray(cigar_off, dtype=np.uint64)
        rc = self.lib.pav_trim_load(self.handle, rows.shape[0], _ptr(rows), _ptr(cigar_text), ctypes.c_void_p(cigar_off.ctypes.data))
        if rc == PAV_E_CIGAR:
            err = CigarErr()
            self.lib.pav_cigar_error(self.handle, ctypes.byref(err))
            raise CigarDeviceError(self.lib.pav_last_error(self.handle).decode(), err)
        self._check(rc, 'pav_trim_load')
        self._trim_n = rows.shape[0]

    def trim_pass(self, order, mode, min_trim_tig_len, match_tig=False):
        """One pair-loop pass; raises TrimDeviceError (with the pav_trim_err detail) where the reference would raise."""
        order = np.ascontiguousarray(order, dtype=np.uint32)
        rc = self.lib.pav_trim_pass(self.handle, order.shape[0], _ptr(order), int(mode), int(min_trim_tig_len), int(bool(match_tig)))
        if rc == PAV_E_TRIM:
            err = TrimErr()
            self.lib.pav_trim_error(self.handle, ctypes.byref(err))
            raise TrimDeviceError(self.lib.pav_last_error(self.handle).decode(), err)
        self._check(rc, 'pav_trim_pass')

    def trim_pair(self, row_l, row_r, mode, rev_l, rev_r):
        rc = self.lib.pav_trim_pair(self.handle, int(row_l), int(row_r), int(mode), int(bool(rev_l)), int(bool(rev_r)))
        if rc == PAV_E_TRIM:
            err = TrimErr()
            self.lib.pav_trim_error(self.handle, ctypes.byref(err))
            raise TrimDeviceError(self.lib.pav_last_error(self.handle).decode(), err)
        self._check(rc, 'pav_trim_pair')

    def trim_fetch(self, with_cigar=True, with_counts=True):
        """-> (rows, count_cigar records or None, CIGAR strings of the modified rows (None elsewhere) or None)."""
        n = self._trim_n
        rows = np.zeros(n, dtype=TRIM_ROW_DTYPE)
        counts = np.zeros(n, dtype=TRIM_COUNT_DTYPE) if with_counts else None
        nbytes = ctypes.c_uint64(0)
        self._check(self.lib.pav_trim_fetch(self.handle, _ptr(rows), _ptr(counts) if with_counts else None,
                                            ctypes.byref(nbytes) if with_cigar else None), 'pav_trim_fetch')
        if not with_cigar:
            return rows, counts, None
        text = np.zeros(int(nbytes.value), dtype=np.uint8)
        off = np.zeros(n + 1, dtype=np.uint64)
        self._check(self.lib.pav_trim_fetch_cigar(self.handle, _ptr(text), ctypes.c_void_p(off.ctypes.data)), 'pav_trim_fetch_cigar')
        raw = text.tobytes()
        return rows, counts, [raw[int(off[i]):int(off[i + 1])].decode() if rows['modified'][i] else None for i in range(n)]

    # ---- inversion-signature flagging ---------------------------------------------------------------------------
    @staticmethod
    def _copy_out(ptr, n, dtype):
        n = int(n)
        if not n:
            return np.empty(0, dtype=dtype)
        buf = (ctypes.c_uint8 * (n * dtype.itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype, count=n).copy()

    def flag_params(self, **kw):
        p = FlagParams()
        self.lib.pav_flag_params_default(ctypes.byref(p))
        for k, v in kw.items():
            if not hasattr(p, k):
                raise TypeError(f'unknown flag parameter {k}')
            setattr(p, k, int(v))
        return p

    def flag_cluster(self, chrom, pos, end, win, win_min, min_count):
        chrom = np.ascontiguousarray(chrom, dtype=np.uint32)
        pos = np.ascontiguousarray(pos, dtype=np.int64)
        end = np.ascontiguousarray(end, dtype=np.int64)
        out, n = ctypes.c_void_p(0), ctypes.c_uint64(0)
        self._check(self.lib.pav_flag_cluster(self.handle, len(chrom), _ptr(chrom), _ptr(pos), _ptr(end), int(win), int(win_min),
                                              int(min_count), ctypes.byref(out), ctypes.byref(n)), 'pav_flag_cluster')
        return self._copy_out(out.value, n.value, FLAG_RGN_DTYPE)

    def flag_insdel(self, ins_chrom, ins_pos, ins_svlen, del_chrom, del_pos, del_end, flank_cluster, flank_merge):
        a = [np.ascontiguousarray(ins_chrom, dtype=np.uint32), np.ascontiguousarray(ins_pos, dtype=np.int64),
             np.ascontiguousarray(ins_svlen, dtype=np.int64), np.ascontiguousarray(del_chrom, dtype=np.uint32),
             np.ascontiguousarray(del_pos, dtype=np.int64), np.ascontiguousarray(del_end, dtype=np.int64)]
        out, n = ctypes.c_void_p(0), ctypes.c_uint64(0)
        self._check(self.lib.pav_flag_insdel(self.handle, len(a[0]), _ptr(a[0]), _ptr(a[1]), _ptr(a[2]), len(a[3]), _ptr(a[3]),
                                             _ptr(a[4]), _ptr(a[5]), int(flank_cluster), int(flank_merge), ctypes.byref(out),
                                             ctypes.byref(n)), 'pav_flag_insdel')
        return self._copy_out(out.value, n.value, FLAG_RGN_DTYPE)

    def flag_merge_loci(self, tables, flank, batch_count, sig_filter):
        """tables: four FLAG_RGN_DTYPE arrays in FLAG_TABLES order."""
        tabs = [np.ascontiguousarray(t, dtype=FLAG_RGN_DTYPE) for t in tables]
        ptrs = (ctypes.c_void_p * 4)(*[t.ctypes.data if len(t) else None for t in tabs])
        ns = (ctypes.c_uint64 * 4)(*[len(t) for t in tabs])
        out, n = ctypes.c_void_p(0), ctypes.c_uint64(0)
        self._check(self.lib.pav_flag_merge_loci(self.handle, ptrs, ns, int(flank), int(batch_count), int(sig_filter), ctypes.byref(out),
                                                 ctypes.byref(n)), 'pav_flag_merge_loci')
        return self._copy_out(out.value, n.value, FLAG_LOCUS_DTYPE)

    def cigar_flag(self, trim_pos, trim_end, params=None):
        """All flag tables + flagged loci from the records of the last cigar_call: -> (dict of tables, loci, counts)."""
        tp = np.ascontiguousarray(trim_pos, dtype=np.int64)
        te = np.ascontiguousarray(trim_end, dtype=np.int64)
        params = params if params is not None else self.flag_params()
        res = FlagResult()
        self._check(self.lib.pav_cigar_flag(self.handle, _ptr(tp), _ptr(te), ctypes.byref(params), ctypes.byref(res)), 'pav_cigar_flag')
        tables = {name: self._copy_out(res.tables[i], res.n[i], FLAG_RGN_DTYPE) for i, name in enumerate(FLAG_TABLES)}
        loci = self._copy_out(res.loci, res.n_loci, FLAG_LOCUS_DTYPE)
        return tables, loci, {'n_snv_pass': int(res.n_snv_pass), 'n_indel_pass': int(res.n_indel_pass)}

    def cigar_verify(self):
        """Verify mode: -> dict(eq_bases, eq_mismatch, x_bases, x_match, first_bad_op) for the last cigar_call."""
        c = (ctypes.c_uint64 * 5)()
        self._check(self.lib.pav_cigar_verify(self.handle, c), 'pav_cigar_verify')
        first = int(c[4])
        return {'eq_bases': int(c[0]), 'eq_mismatch': int(c[1]), 'x_bases': int(c[2]), 'x_match': int(c[3]),
                'first_bad_op': None if first == 0xFFFFFFFFFFFFFFFF else first}

    def cigar_fetch_ops(self, n_ops, n_aln):
        ops = np.empty(n_ops, dtype=np.uint32)
        off = np.empty(n_aln + 1, dtype=np.uint64)
        self._check(self.lib.pav_cigar_fetch_ops(self.handle, _ptr(ops), ctypes.c_void_p(off.ctypes.data)),
                    'pav_cigar_fetch_ops')
        return ops, off

    def align_index(self, row_pos, cigar_text, cigar_off):
        """Device tokenizer + prefix scan for lift-over: -> (ops u32, op_off u64, sub_begin u32, qry_begin u32)."""
        row_pos = np.ascontiguousarray(row_pos, dtype=np.uint32)
        cigar_text = np.ascontiguousarray(cigar_text, dtype=np.uint8)
        cigar_off = np.ascontiguousarray(cigar_off, dtype=np.uint64)
        n = row_pos.shape[0]
        n_ops = ctypes.c_uint64(0)
        rc = self.lib.pav_align_index(self.handle, n, _ptr(row_pos), _ptr(cigar_text), ctypes.c_void_p(cigar_off.ctypes.data),
                                      ctypes.byref(n_ops), None, None, None, None)
        if rc == PAV_E_CIGAR:
            e = CigarErr()
            self.lib.pav_cigar_error(self.handle, ctypes.byref(e))
            raise CigarDeviceError(self.lib.pav_last_error(self.handle).decode(), e)
        self._check(rc, 'pav_align_index')
        ops = np.empty(n_ops.value, dtype=np.uint32)
        op_off = np.empty(n + 1, dtype=np.uint64)
        sub_begin = np.empty(n_ops.value, dtype=np.uint32)
        qry_begin = np.empty(n_ops.value, dtype=np.uint32)
        dummy = np.zeros(1, dtype=np.uint32)
        self._check(self.lib.pav_align_index(self.handle, n, _ptr(row_pos), _ptr(cigar_text), ctypes.c_void_p(cigar_off.ctypes.data),
                                             ctypes.byref(n_ops), ctypes.c_void_p((ops if ops.size else dummy).ctypes.data),
                                             ctypes.c_void_p(op_off.ctypes.data), _ptr(sub_begin), _ptr(qry_begin)), 'pav_align_index')
        return ops, op_off, sub_begin, qry_begin

    def homology(self, queries):
        q = np.ascontiguousarray(queries, dtype=HOM_QUERY_DTYPE)
        out = np.zeros(q.shape[0], dtype=np.uint32)
        self._check(self.lib.pav_homology(self.handle, q.shape[0], _ptr(q), _ptr(out)), 'pav_homology')
        return out

    # -- k-mer state + density scan --------------------------------------------------------------------------
    def density_batch(self, jobs, params):
        """``jobs``: list of DenJob; returns list of DenResult (tables / runs stay resident until the next batch)."""
        n = len(jobs)
        arr = (DenJob * max(n, 1))(*jobs)
        res = (DenResult * max(n, 1))()
        self._check(self.lib.pav_density_batch(self.handle, n, arr, ctypes.byref(params), res), 'pav_density_batch')
        return [res[i] for i in range(n)]

    def density_runs(self, job, n_runs):
        runs = np.zeros(n_runs, dtype=RUN_DTYPE)
        self._check(self.lib.pav_density_runs(self.handle, job, _ptr(runs)), 'pav_density_runs')
        return [(int(r['state']), int(r['count']), int(r['pos']), int(r['end'])) for r in runs]

    def density_table(self, job, n_rows):
        cols = {'INDEX': np.zeros(n_rows, dtype=np.int64), 'STATE_MER': np.zeros(n_rows, dtype=np.int8),
                'STATE': np.zeros(n_rows, dtype=np.int8), 'KERN_FWD': np.zeros(n_rows, dtype=np.float64),
                'KERN_FWDREV': np.zeros(n_rows, dtype=np.float64), 'KERN_REV': np.zeros(n_rows, dtype=np.float64),
                'KMER': np.zeros(n_rows, dtype=np.uint64)}
        self._check(self.lib.pav_density_table(self.handle, job, _ptr(cols['INDEX']), _ptr(cols['STATE_MER']),
                                               _ptr(cols['STATE']), _ptr(cols['KERN_FWD']), _ptr(cols['KERN_FWDREV']),
                                               _ptr(cols['KERN_REV']), _ptr(cols['KMER'])), 'pav_density_table')
        return cols

    def density_annotate(self, job, n_rows, ref_id, ref_up, ref_dn, qry_index_base, tig_up, tig_dn):
        flank = np.zeros(n_rows, dtype=np.uint8)
        match = np.zeros(n_rows, dtype=np.uint8)
        self._check(self.lib.pav_density_annotate(
            self.handle, job, ref_id, ref_up[0], ref_up[1], ref_dn[0], ref_dn[1], qry_index_base, tig_up[0], tig_up[1],
            tig_dn[0], tig_dn[1], _ptr(flank) or ctypes.c_void_p(flank.ctypes.data),
            _ptr(match) or ctypes.c_void_p(match.ctypes.data)), 'pav_density_annotate')
        return flank, match

    # -- native batched inversion scan -------------------------------------------------------------------------
    def inv_load_alignments(self, aln, cigar_text, cigar_off):
        aln = np.ascontiguousarray(aln, dtype=INV_ALN_DTYPE)
        cigar_text = np.ascontiguousarray(cigar_text, dtype=np.uint8)
        cigar_off = np.ascontiguousarray(cigar_off, dtype=np.uint64)
        rc = self.lib.pav_inv_load_alignments(self.handle, aln.shape[0], _ptr(aln), _ptr(cigar_text),
                                              ctypes.c_void_p(cigar_off.ctypes.data))
        if rc == PAV_E_CIGAR:
            e = CigarErr()
            self.lib.pav_cigar_error(self.handle, ctypes.byref(e))
            raise CigarDeviceError(self.lib.pav_last_error(self.handle).decode(), e)
        self._check(rc, 'pav_inv_load_alignments')

    def inv_scan_batch(self, regions, params):
        regions = np.ascontiguousarray(regions, dtype=INV_REGION_DTYPE)
        n = regions.shape[0]
        # the result block is kept between scans of the same size (half a megabyte of fresh zeroed memory per scan of a thousand
        # regions otherwise); like the table views it is valid until the next scan on this context
        cached = getattr(self, '_inv_res_buf', None)
        if cached is None or len(cached) != max(n, 1):
            cached = self._inv_res_buf = (InvResult * max(n, 1))()
        res = cached
        self._inv_generation = getattr(self, '_inv_generation', 0) + 1      # invalidates table views of earlier scans
        self._check(self.lib.pav_inv_scan_batch(self.handle, n, _ptr(regions), ctypes.byref(params), res), 'pav_inv_scan_batch')
        return res

    def inv_table_view(self, region, generation):
        """Zero-copy numpy views of a call's table in the library's pinned host memory (valid until the next scan)."""
        if generation != getattr(self, '_inv_generation', 0) or not self.handle:
            raise PavDeviceError('the density table of this InvCall is no longer resident: read InvCall.df before the next '
                                 'scan on the same context, or scan with eager_tables=True')
        n = ctypes.c_uint32(0)
        ptrs = [ctypes.c_void_p() for _ in range(9)]
        self._check(self.lib.pav_inv_table_view(self.handle, region, ctypes.byref(n), *[ctypes.byref(p) for p in ptrs]),
                    'pav_inv_table_view')
        n = int(n.value)
        types = [ctypes.c_uint32, ctypes.c_int8, ctypes.c_int8, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_uint64,
                 ctypes.c_uint8, ctypes.c_uint8]
        arrs = [np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(t)), shape=(n,)) if n else np.zeros(0, dtype=t)
                for p, t in zip(ptrs, types)]
        for a in arrs:                                  # library memory (a column of zeros is shared between calls): read-only
            a.flags.writeable = False
        cols = dict(zip(['INDEX', 'STATE_MER', 'STATE', 'KERN_FWD', 'KERN_FWDREV', 'KERN_REV', 'KMER'], arrs[:7]))
        return cols, arrs[7], arrs[8]

    def inv_text(self, region, what, n_bytes):
        buf = ctypes.create_string_buffer(n_bytes + 1)
        self._check(self.lib.pav_inv_text(self.handle, region, what, buf, n_bytes + 1), 'pav_inv_text')
        return buf.value.decode()

    def inv_texts(self, what, n_regions, total_bytes=None, joined=False):
        """Log (what 0) / error (what 1) text or 'INV Found' line (what 2) of every region of the last scan: list of str
        (``joined``: one string, the texts in region order).  ``total_bytes`` None: asked from the library first."""
        off = np.zeros(n_regions + 1, dtype=np.uint64)
        if total_bytes is None:
            self._check(self.lib.pav_inv_texts(self.handle, what, None, 0, _ptr(off)), 'pav_inv_texts')
            total_bytes = int(off[n_regions])
        buf = ctypes.create_string_buffer(int(total_bytes) + 1)
        self._check(self.lib.pav_inv_texts(self.handle, what, buf, int(total_bytes), _ptr(off)), 'pav_inv_texts')
        raw = buf.raw
        if joined:
            return [raw[:int(off[n_regions])].decode()]
        o = off.tolist()
        text = raw.decode()
        if len(text) == len(raw):                                   # ASCII: byte offsets are character offsets
            return [text[o[i]:o[i + 1]] for i in range(n_regions)]
        return [raw[o[i]:o[i + 1]].decode() for i in range(n_regions)]

    def inv_write_tables(self, regions, paths, threads=0, gzip_level=0):
        """Write the density tables of calls of the last scan (region numbers of that scan) to ``paths`` as TSV / TSV.gz."""
        rg = np.ascontiguousarray(regions, dtype=np.uint32)
        cp = (ctypes.c_char_p * max(1, rg.shape[0]))(*[str(p).encode() for p in paths])
        self._check(self.lib.pav_inv_write_tables(self.handle, rg.shape[0], _ptr(rg), cp, int(threads), int(gzip_level)),
                    'pav_inv_write_tables')

    def inv_table(self, region, n_rows):
        cols = {'INDEX': np.zeros(n_rows, dtype=np.int64), 'STATE_MER': np.zeros(n_rows, dtype=np.int8),
                'STATE': np.zeros(n_rows, dtype=np.int8), 'KERN_FWD': np.zeros(n_rows, dtype=np.float64),
                'KERN_FWDREV': np.zeros(n_rows, dtype=np.float64), 'KERN_REV': np.zeros(n_rows, dtype=np.float64),
                'KMER': np.zeros(n_rows, dtype=np.uint64)}
        flank = np.zeros(n_rows, dtype=np.uint8)
        match = np.zeros(n_rows, dtype=np.uint8)
        self._check(self.lib.pav_inv_table(self.handle, region, _ptr(cols['INDEX']), _ptr(cols['STATE_MER']), _ptr(cols['STATE']),
                                           _ptr(cols['KERN_FWD']), _ptr(cols['KERN_FWDREV']), _ptr(cols['KERN_REV']),
                                           _ptr(cols['KMER']), _ptr(flank), _ptr(match)), 'pav_inv_table')
        return cols, flank, match

    def inv_tables(self, n_rows_per_region):
        """All call tables of the last scan in one copy: -> (cols dict of concatenated columns, flank, match, row_off)."""
        n_rows = np.asarray(n_rows_per_region, dtype=np.uint64)
        off = np.zeros(n_rows.shape[0] + 1, dtype=np.uint64)
        off[1:] = np.cumsum(n_rows)
        total = int(off[-1])
        cols = {'INDEX': np.empty(total, dtype=np.int64), 'STATE_MER': np.empty(total, dtype=np.int8),
                'STATE': np.empty(total, dtype=np.int8), 'KERN_FWD': np.empty(total, dtype=np.float64),
                'KERN_FWDREV': np.empty(total, dtype=np.float64), 'KERN_REV': np.empty(total, dtype=np.float64),
                'KMER': np.empty(total, dtype=np.uint64)}
        flank = np.empty(total, dtype=np.uint8)
        match = np.empty(total, dtype=np.uint8)
        if total:
            self._check(self.lib.pav_inv_tables(self.handle, n_rows.shape[0], ctypes.c_void_p(off.ctypes.data), _ptr(cols['INDEX']),
                                                _ptr(cols['STATE_MER']), _ptr(cols['STATE']), _ptr(cols['KERN_FWD']),
                                                _ptr(cols['KERN_FWDREV']), _ptr(cols['KERN_REV']), _ptr(cols['KMER']), _ptr(flank),
                                                _ptr(match)), 'pav_inv_tables')
        return cols, flank, match, off

    # -- profiling ----------------------------------------------------------------------------------------
    def prof_enable(self, on=True):
        self._check(self.lib.pav_prof_enable(self.handle, 1 if on else 0), 'pav_prof_enable')

    def prof_reset(self):
        self._check(self.lib.pav_prof_reset(self.handle), 'pav_prof_reset')

    def prof_read(self):
        """{kernel name: (launches, total_ms)} measured with HIP events on the library's stream."""
        n = self.lib.pav_prof_count(self.handle)
        if n < 0:
            self._check(n, 'pav_prof_count')
        out = {}
        for i in range(n):
            name = ctypes.create_string_buffer(128)
            launches = ctypes.c_uint64(0)
            ms = ctypes.c_double(0.0)
            self._check(self.lib.pav_prof_get(self.handle, i, name, 128, ctypes.byref(launches), ctypes.byref(ms)),
                        'pav_prof_get')
            out[name.value.decode()] = (int(launches.value), float(ms.value))
        return out


class FastaFile:
    """A FASTA file parsed by the library (``pav_fasta_open``: plain, gzip or BGZF with parallel inflate).  ``seq(i)`` is a
    zero-copy ``uint8`` view of record i in the library's memory; it stays valid while this object is alive."""

    KINDS = ('plain', 'gzip', 'bgzf')

    def __init__(self, path, threads=0):
        self.lib = load()
        h = ctypes.c_void_p(0)
        rc = self.lib.pav_fasta_open(str(path).encode(), int(threads), ctypes.byref(h))
        if rc != PAV_OK:
            raise PavDeviceError('pav_fasta_open failed ({}): {}'.format(rc, (self.lib.pav_last_error(None) or b'').decode()))
        self.handle = h
        n = int(self.lib.pav_fasta_count(h))
        self.names = [self.lib.pav_fasta_name(h, i).decode() for i in range(n)]
        self.lengths = [int(self.lib.pav_fasta_length(h, i)) for i in range(n)]
        self.kind = self.KINDS[self.lib.pav_fasta_kind(h)]

    def seq(self, i):
        n = self.lengths[i]
        if n == 0:
            return np.zeros(0, dtype=np.uint8)
        buf = (ctypes.c_uint8 * n).from_address(self.lib.pav_fasta_seq(self.handle, i))
        buf._pav_owner = self            # the view keeps the file's memory alive (the library recycles the buffer of a closed file)
        a = np.frombuffer(buf, dtype=np.uint8)
        a.flags.writeable = False
        return a

    def close(self):
        if self.handle:
            self.lib.pav_fasta_close(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class SamInfo(ctypes.Structure):
    _fields_ = [('n_records', ctypes.c_uint64), ('n_rows', ctypes.c_uint64), ('n_ref', ctypes.c_uint32), ('n_qry', ctypes.c_uint32),
                ('cigar_bytes', ctypes.c_uint64), ('tag_bytes', ctypes.c_uint64), ('header_bytes', ctypes.c_uint64)]


_SAM_COLS = [('index', np.int64), ('pos', np.int64), ('end', np.int64), ('chrom_id', np.uint32), ('qry_id', np.uint32),
             ('query_alignment_start', np.int64), ('query_alignment_end', np.int64), ('clip_h', np.int64), ('tig_map_pos', np.int64),
             ('mapq', np.int32), ('flag', np.int32), ('has_m', np.uint8), ('status', np.uint8), ('ref_bp', np.int64), ('tig_bp', np.int64),
             ('err_kind', np.uint32), ('err_op', np.uint32), ('err_len', np.uint32), ('err_char', np.uint32),
             ('cigar_text', None), ('cigar_off', None), ('tag_text', None), ('rg_off', None), ('ao_off', None),
             ('rg_kind', np.uint8), ('ao_kind', np.uint8)]


class SamCols(ctypes.Structure):
    _fields_ = [(name, ctypes.c_void_p) for name, _ in _SAM_COLS]


class SamFile:
    """A SAM file parsed by the library (``pav_sam_open``): what ``pavlib.align.get_align_bed`` reads from pysam, per kept record,
    plus the transformed CIGAR strings.  ``cols`` maps field name -> numpy array; ``cigars`` / ``rg`` / ``ao`` are Python lists."""

    def __init__(self, path, min_mapq=0, threads=0):
        self.lib = load()
        h = ctypes.c_void_p(0)
        rc = self.lib.pav_sam_open(str(path).encode(), int(min_mapq), int(threads), ctypes.byref(h))
        if rc != PAV_OK:
            raise PavDeviceError('pav_sam_open failed ({}): {}'.format(rc, (self.lib.pav_last_error(None) or b'').decode()))
        self.handle = h
        try:
            info = SamInfo()
            self.lib.pav_sam_info(h, ctypes.byref(info))
            n = self.n_rows = int(info.n_rows)
            self.n_records = int(info.n_records)
            self.ref_names = [self.lib.pav_sam_name(h, 0, i).decode() for i in range(info.n_ref)]
            self.qry_names = [self.lib.pav_sam_name(h, 1, i).decode() for i in range(info.n_qry)]
            head = ctypes.create_string_buffer(max(1, int(info.header_bytes)))
            self.lib.pav_sam_header(h, head)
            self.header = head.raw[:int(info.header_bytes)]
            cols = {name: np.zeros(max(n, 1), dtype=dt) for name, dt in _SAM_COLS if dt is not None}
            cols['cigar_text'] = np.zeros(max(1, int(info.cigar_bytes)), dtype=np.uint8)
            cols['tag_text'] = np.zeros(max(1, int(info.tag_bytes)), dtype=np.uint8)
            for name in ('cigar_off', 'rg_off', 'ao_off'):
                cols[name] = np.zeros(n + 1, dtype=np.uint64)
            c = SamCols(**{name: cols[name].ctypes.data for name, _ in _SAM_COLS})
            rc = self.lib.pav_sam_fetch(h, ctypes.byref(c))
            if rc != PAV_OK:
                raise PavDeviceError('pav_sam_fetch failed ({})'.format(rc))
            self.cols = {name: (a[:n] if name not in ('cigar_off', 'rg_off', 'ao_off', 'cigar_text', 'tag_text') else a) for name, a in cols.items()}
            text = cols['cigar_text'].tobytes()
            off = cols['cigar_off'].tolist()
            self.cigars = [text[off[i]:off[i + 1]].decode() for i in range(n)]
            tags = cols['tag_text'].tobytes()
            ro, ao = cols['rg_off'].tolist(), cols['ao_off'].tolist()
            self.rg = [tags[ro[i]:ao[i]].decode() for i in range(n)]
            self.ao = [tags[ao[i]:ro[i + 1]].decode() for i in range(n)]
        finally:
            self.lib.pav_sam_close(h)
            self.handle = None


class BedTable:
    """An alignment table parsed by the library (``pav_bed_open``): numeric columns as numpy arrays, names, CIGAR text."""

    _NUMERIC = {'POS': 'pos', 'END': 'end', 'INDEX': 'index', 'QRY_POS': 'qry_pos', 'QRY_END': 'qry_end', 'QRY_LEN': 'qry_len',
                'MAPQ': 'mapq', 'CALL_BATCH': 'call_batch'}

    def __init__(self, path, with_cigar=True):
        self.lib = load()
        h = ctypes.c_void_p(0)
        rc = self.lib.pav_bed_open(str(path).encode(), 1 if with_cigar else 0, ctypes.byref(h))
        if rc != PAV_OK:
            raise PavDeviceError('pav_bed_open failed ({}): {}'.format(rc, (self.lib.pav_last_error(None) or b'').decode()))
        self.handle = h
        info = BedInfo()
        self.lib.pav_bed_info(self.handle, ctypes.byref(info))
        self.n_rows, self.cigar_bytes = int(info.n_rows), int(info.cigar_bytes)
        self.columns = [c for i, c in enumerate(BED_COLUMNS) if info.columns >> i & 1]
        self.chrom_names = [self.lib.pav_bed_name(self.handle, 0, i).decode() for i in range(info.n_chrom)]
        self.qry_names = [self.lib.pav_bed_name(self.handle, 1, i).decode() for i in range(info.n_qry)]
        self._cols = None

    def close(self):
        if self.handle:
            self.lib.pav_bed_close(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def fetch(self):
        """dict: column name -> numpy array (ids for #CHROM / QRY_ID, bool for REV), plus 'CIGAR_TEXT' / 'CIGAR_OFF'."""
        if self._cols is not None:
            return self._cols
        n = self.n_rows
        out, c = {}, BedCols()
        for name, field in self._NUMERIC.items():
            if name in self.columns:
                out[name] = np.zeros(n, dtype=np.int64)
                setattr(c, field, out[name].ctypes.data)
        if '#CHROM' in self.columns:
            out['#CHROM'] = np.zeros(n, dtype=np.uint32)
            c.chrom_id = out['#CHROM'].ctypes.data
        if 'QRY_ID' in self.columns:
            out['QRY_ID'] = np.zeros(n, dtype=np.uint32)
            c.qry_id = out['QRY_ID'].ctypes.data
        if 'REV' in self.columns:
            out['REV'] = np.zeros(n, dtype=np.uint8)
            c.rev = out['REV'].ctypes.data
        if 'CIGAR' in self.columns and self.cigar_bytes:
            out['CIGAR_TEXT'] = np.zeros(self.cigar_bytes, dtype=np.uint8)
            out['CIGAR_OFF'] = np.zeros(n + 1, dtype=np.uint64)
            c.cigar_text, c.cigar_off = out['CIGAR_TEXT'].ctypes.data, out['CIGAR_OFF'].ctypes.data
        rc = self.lib.pav_bed_fetch(self.handle, ctypes.byref(c))
        if rc != PAV_OK:
            raise PavDeviceError(f'pav_bed_fetch failed ({rc})')
        if 'REV' in out:
            out['REV'] = out['REV'].astype(bool)
        self._cols = out
        return out


class TrimDeviceError(RuntimeError):
    """pav_trim_pass returned PAV_E_TRIM; ``detail`` is the TrimErr record the host mirror turns into the reference's message."""

    def __init__(self, message, detail=None):
        super().__init__(message)
        self.detail = detail


class CigarDeviceError(RuntimeError):
    """The device walk hit what the reference raises on; ``detail`` is the ``pav_cigar_err`` struct."""

    def __init__(self, message, detail=None):
        super().__init__(message)
        self.detail = detail


def device_count():
    return int(load().pav_device_count())


def device_pool_trim(device_id=-1):
    """Give the idle device blocks the library keeps for the next context (include/pav_amd.h: pav_device_pool_trim) back to the
    driver; ``device_id`` < 0: of every GPU.  Returns the bytes freed."""
    return int(load().pav_device_pool_trim(int(device_id)))
