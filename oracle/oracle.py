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
LIB_PATH = os.path.join(_HERE, '_build', 'libpavoracle.so')

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
