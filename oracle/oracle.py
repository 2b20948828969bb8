"""
ctypes wrapper of the CPU oracle (oracle/_build/libpavoracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never
by the product (pav_amd/).  Record dtypes are byte-identical to the product's so one comparison covers both.
"""

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PAV_ORACLE_LIB: another build of the same sources (tests/test_host_sanitize.py points it at the ASan + UBSan build)
LIB_PATH = os.environ.get('PAV_ORACLE_LIB') or os.path.join(_HERE, '_build', 'libpavoracle.so')

ALN_DTYPE = np.dtype([('ref_id', '<u4'), ('tig_id', '<u4'), ('pos', '<u4'), ('rev', '<u4')])
SNV_DTYPE = np.dtype([('aln', '<u4'), ('pos', '<u4'), ('qry_pos', '<u4'), ('ref', 'u1'), ('alt', 'u1'),
                      ('pad', '<u2')])
INDEL_DTYPE = np.dtype([('aln', '<u4'), ('op_index', '<u4'), ('pos', '<u4'), ('end', '<u4'), ('svlen', '<u4'),
                        ('qry_pos', '<u4'), ('qry_end', '<u4'), ('left_shift', '<u4'),
                        ('hom_ref_l', '<u4'), ('hom_ref_r', '<u4'), ('hom_tig_l', '<u4'), ('hom_tig_r', '<u4'),
                        ('seq_off', '<u8'), ('svtype', 'u1'), ('pad', 'u1', (7,))])


class CigarErr(ctypes.Structure):
    _fields_ = [('kind', ctypes.c_int32), ('aln', ctypes.c_uint32), ('op_index', ctypes.c_uint32),
                ('op_char', ctypes.c_uint32), ('pos_ref', ctypes.c_uint32), ('pos_tig', ctypes.c_uint32)]


_LIB = None


def build():
    subprocess.run(['make', '-C', _HERE, '-s'], check=True)


def load():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            build()
        lib = ctypes.CDLL(LIB_PATH)
        P = ctypes.c_void_p
        lib.orc_cigar_tokenize.restype = ctypes.c_int
        lib.orc_cigar_tokenize.argtypes = [ctypes.c_char_p, ctypes.c_uint64, P, ctypes.c_uint64, P, P, P]
        for f in (lib.orc_left_homology, lib.orc_right_homology):
            f.restype = ctypes.c_int64
            f.argtypes = [ctypes.c_int64, ctypes.c_char_p, ctypes.c_int64, ctypes.c_char_p, ctypes.c_int64]
        lib.orc_cigar_call.restype = P
        lib.orc_cigar_call.argtypes = [P, P, ctypes.c_uint32, P, P, ctypes.c_uint32, P, ctypes.c_uint32,
                                       ctypes.c_char_p, P, P]
        for name in ('orc_calls_n_snv', 'orc_calls_n_indel', 'orc_calls_seq_bytes'):
            getattr(lib, name).restype = ctypes.c_uint64
            getattr(lib, name).argtypes = [P]
        for name in ('orc_calls_snv', 'orc_calls_indel', 'orc_calls_seq'):
            getattr(lib, name).restype = P
            getattr(lib, name).argtypes = [P]
        lib.orc_calls_free.restype = None
        lib.orc_calls_free.argtypes = [P]
        _LIB = lib
    return _LIB


def cigar_tokenize(text):
    """-> (rc, [(len, op_char)...], err_off, err_char)   pavlib/align/align.py:286-322"""
    lib = load()
    b = text.encode() if isinstance(text, str) else bytes(text)
    ops = np.zeros(max(1, len(b)), dtype=np.uint32)
    n = ctypes.c_uint64(0)
    eo, ec = ctypes.c_uint32(0), ctypes.c_uint32(0)
    rc = lib.orc_cigar_tokenize(b, len(b), ops.ctypes.data, ops.shape[0], ctypes.byref(n), ctypes.byref(eo),
                                ctypes.byref(ec))
    ops = ops[:n.value]
    return rc, [(int(o >> 4), 'MIDNSHP=X'[int(o & 15)]) for o in ops], eo.value, ec.value


def left_homology(pos_tig, seq_tig, seq_sv):
    lib = load()
    if seq_tig is None or seq_sv is None:
        return 0
    t, s = seq_tig.encode(), seq_sv.encode()
    return int(lib.orc_left_homology(pos_tig, t, len(t), s, len(s)))


def right_homology(pos_tig, seq_tig, seq_sv):
    lib = load()
    if seq_tig is None or seq_sv is None:
        return 0
    t, s = seq_tig.encode(), seq_sv.encode()
    return int(lib.orc_right_homology(pos_tig, t, len(t), s, len(s)))


def cigar_call(ref_arrays, tig_arrays, aln, cigar_text, cigar_off):
    """Scalar walk over all rows.  -> (snv, indel, seq_blob, err) with err.kind == 0 on success."""
    lib = load()
    ref_arrays = [np.ascontiguousarray(a, dtype=np.uint8) for a in ref_arrays]
    tig_arrays = [np.ascontiguousarray(a, dtype=np.uint8) for a in tig_arrays]
    aln = np.ascontiguousarray(aln, dtype=ALN_DTYPE)
    cigar_off = np.ascontiguousarray(cigar_off, dtype=np.uint64)
    text = bytes(np.ascontiguousarray(cigar_text, dtype=np.uint8).tobytes())
    rp = (ctypes.c_void_p * max(1, len(ref_arrays)))(*[a.ctypes.data for a in ref_arrays])
    rl = (ctypes.c_uint64 * max(1, len(ref_arrays)))(*[a.shape[0] for a in ref_arrays])
    tp = (ctypes.c_void_p * max(1, len(tig_arrays)))(*[a.ctypes.data for a in tig_arrays])
    tl = (ctypes.c_uint64 * max(1, len(tig_arrays)))(*[a.shape[0] for a in tig_arrays])
    err = CigarErr()
    h = lib.orc_cigar_call(rp, rl, len(ref_arrays), tp, tl, len(tig_arrays), aln.ctypes.data, aln.shape[0], text,
                           cigar_off.ctypes.data, ctypes.byref(err))
    try:
        ns, ni, nb = lib.orc_calls_n_snv(h), lib.orc_calls_n_indel(h), lib.orc_calls_seq_bytes(h)
        snv = np.zeros(ns, dtype=SNV_DTYPE)
        indel = np.zeros(ni, dtype=INDEL_DTYPE)
        blob = np.zeros(nb, dtype=np.uint8)
        if ns:
            ctypes.memmove(snv.ctypes.data, lib.orc_calls_snv(h), ns * SNV_DTYPE.itemsize)
        if ni:
            ctypes.memmove(indel.ctypes.data, lib.orc_calls_indel(h), ni * INDEL_DTYPE.itemsize)
        if nb:
            ctypes.memmove(blob.ctypes.data, lib.orc_calls_seq(h), nb)
    finally:
        lib.orc_calls_free(h)
    return snv, indel, blob, err


# ---------------------------------------------------------------------------------------------------------
# k-mer state + density scan (oracle/pav_oracle_density.c)
# ---------------------------------------------------------------------------------------------------------

class DenParams(ctypes.Structure):
    _fields_ = [('k', ctypes.c_int32), ('min_informative', ctypes.c_uint32), ('min_state_count', ctypes.c_uint32),
                ('den_smooth', ctypes.c_double), ('state_run_smooth', ctypes.c_uint32),
                ('state_run_delta', ctypes.c_double), ('max_ref_kmer_count', ctypes.c_uint32)]


class DensityInfo(ctypes.Structure):
    _fields_ = [('status', ctypes.c_int32), ('fail_kind', ctypes.c_int32), ('n', ctypes.c_uint32),
                ('max_count', ctypes.c_uint32), ('max_kmer', ctypes.c_uint64), ('state_count', ctypes.c_uint32 * 3),
                ('h', ctypes.c_double * 3), ('n_eval', ctypes.c_uint64)]


RUN_DTYPE = np.dtype([('state', '<i4'), ('count', '<u4'), ('pos', '<i8'), ('end', '<i8')])


def den_params(k=31, min_informative=2000, min_state_count=20, den_smooth=1.0, state_run_smooth=20,
               state_run_delta=0.005, max_ref_kmer_count=100):
    return DenParams(k, min_informative, min_state_count, den_smooth, state_run_smooth, state_run_delta,
                     max_ref_kmer_count)


def _den_protos(lib):
    if getattr(lib, '_den_ready', False):
        return
    P = ctypes.c_void_p
    lib.orc_density_run.restype = P
    lib.orc_density_run.argtypes = [P, ctypes.c_uint64, P, ctypes.c_uint64, ctypes.c_int, P]
    lib.orc_density_get_info.restype = ctypes.POINTER(DensityInfo)
    lib.orc_density_get_info.argtypes = [P]
    for name in ('orc_density_index', 'orc_density_state_mer', 'orc_density_state', 'orc_density_kmer',
                 'orc_density_interp'):
        getattr(lib, name).restype = P
        getattr(lib, name).argtypes = [P]
    lib.orc_density_kern.restype = P
    lib.orc_density_kern.argtypes = [P, ctypes.c_int]
    lib.orc_density_free.restype = None
    lib.orc_density_free.argtypes = [P]
    lib.orc_density_set_threads.restype = None
    lib.orc_density_set_threads.argtypes = [ctypes.c_int]
    lib.orc_rl_encode.restype = ctypes.c_uint32
    lib.orc_rl_encode.argtypes = [P, P, ctypes.c_uint32, P, ctypes.c_uint32]
    lib.orc_annotate.restype = None
    lib.orc_annotate.argtypes = [P, P, ctypes.c_uint32, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                 ctypes.c_int64, ctypes.c_int64, P, ctypes.c_uint64, P, ctypes.c_uint64, P, P]
    lib.orc_kmer_rc.restype = ctypes.c_uint64
    lib.orc_kmer_rc.argtypes = [ctypes.c_uint64, ctypes.c_int]
    lib.orc_kmer_canonical.restype = ctypes.c_uint64
    lib.orc_kmer_canonical.argtypes = [ctypes.c_uint64, ctypes.c_int]
    lib._den_ready = True


def _copy(ptr, n, dtype):
    out = np.zeros(n, dtype=dtype)
    if n:
        ctypes.memmove(out.ctypes.data, ptr, n * np.dtype(dtype).itemsize)
    return out


def density(ref_seq, tig_seq, ref_rc, params=None, threads=1):
    """scripts/density.py on extracted region sequences (uint8 ASCII arrays).  Returns a dict of arrays + info.
    ``threads``: evaluation points of the KDE in parallel (each summed by one thread in scipy's order: same bits)."""
    lib = load()
    _den_protos(lib)
    lib.orc_density_set_threads(int(threads))
    params = params or den_params()
    ref_seq = np.ascontiguousarray(ref_seq, dtype=np.uint8)
    tig_seq = np.ascontiguousarray(tig_seq, dtype=np.uint8)
    h = lib.orc_density_run(ref_seq.ctypes.data, ref_seq.shape[0], tig_seq.ctypes.data, tig_seq.shape[0],
                            1 if ref_rc else 0, ctypes.byref(params))
    try:
        info = lib.orc_density_get_info(h).contents
        n = int(info.n)
        out = {'status': int(info.status), 'fail_kind': int(info.fail_kind), 'n': n, 'max_count': int(info.max_count),
               'max_kmer': int(info.max_kmer), 'state_count': [int(v) for v in info.state_count],
               'h': [float(v) for v in info.h], 'n_eval': int(info.n_eval)}
        if info.status != 125:
            out['INDEX'] = _copy(lib.orc_density_index(h), n, np.int64)
            out['STATE_MER'] = _copy(lib.orc_density_state_mer(h), n, np.int8)
            out['STATE'] = _copy(lib.orc_density_state(h), n, np.int8)
            out['KMER'] = _copy(lib.orc_density_kmer(h), n, np.uint64)
            out['INTERP'] = _copy(lib.orc_density_interp(h), n, np.uint8)
            for s, name in enumerate(('KERN_FWD', 'KERN_FWDREV', 'KERN_REV')):
                out[name] = _copy(lib.orc_density_kern(h, s), n, np.float64)
    finally:
        lib.orc_density_free(h)
    return out


def rl_encode(state, index):
    lib = load()
    _den_protos(lib)
    state = np.ascontiguousarray(state, dtype=np.int8)
    index = np.ascontiguousarray(index, dtype=np.int64)
    runs = np.zeros(max(1, state.shape[0]), dtype=RUN_DTYPE)
    m = lib.orc_rl_encode(state.ctypes.data, index.ctypes.data, state.shape[0], runs.ctypes.data, runs.shape[0])
    return [(int(r['state']), int(r['count']), int(r['pos']), int(r['end'])) for r in runs[:m]]


def annotate(kmer, index, k, qry_index_base, up, dn, ref_up, ref_dn):
    """pavlib/inv.py:457-561.  up / dn = (pos, end) of the contig duplication regions.  -> (flank, match) codes."""
    lib = load()
    _den_protos(lib)
    kmer = np.ascontiguousarray(kmer, dtype=np.uint64)
    index = np.ascontiguousarray(index, dtype=np.int64)
    ref_up = np.ascontiguousarray(ref_up, dtype=np.uint8)
    ref_dn = np.ascontiguousarray(ref_dn, dtype=np.uint8)
    flank = np.zeros(kmer.shape[0], dtype=np.uint8)
    match = np.zeros(kmer.shape[0], dtype=np.uint8)
    lib.orc_annotate(kmer.ctypes.data, index.ctypes.data, kmer.shape[0], k, qry_index_base, up[0], up[1], dn[0], dn[1],
                     ref_up.ctypes.data, ref_up.shape[0], ref_dn.ctypes.data, ref_dn.shape[0], flank.ctypes.data,
                     match.ctypes.data)
    return flank, match
