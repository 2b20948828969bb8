"""
FASTA access for the host side of the hot path.

Replaces the two ``pysam.FastaFile(...).fetch(...)`` uses on the path (pavlib/cigarcall.py:59-66,
pavlib/seq.py:339-351) without pysam: whole records are held as ``numpy.uint8`` ASCII arrays exactly as they
appear in the file (case preserved), because the device library takes plain byte pointers.  The file is read by the
library's native reader (``pav_fasta_open``, csrc/fastaio.cpp): plain, gzip and BGZF (blocks inflated in parallel);
the arrays are zero-copy views of its buffers.
"""

import os
import threading

import numpy as np

from . import _lib

_CACHE = {}
_CACHE_LOCK = threading.Lock()          # lanes of a cohort rank load and forget files from their own threads


class Fasta:
    """All records of one FASTA file, name -> uint8 ASCII array (line breaks removed, case preserved)."""

    def __init__(self, path):
        self.path = str(path)
        self.names = []
        self.seqs = {}
        self._record = {}                              # name -> record number in the native reader
        self._load()

    def record_numbers(self, names):
        """Record numbers of ``names`` for ``Context.seq_load_fasta``."""
        return [self._record[str(n)] for n in names]

    def _load(self):
        self.native = _lib.FastaFile(self.path)
        self.kind = self.native.kind
        for i, name in enumerate(self.native.names):
            self.names.append(name)
            if name not in self.seqs:                  # a repeated name keeps its first record (dict of faidx names)
                self.seqs[name] = self.native.seq(i)
                self._record[name] = i

    def __contains__(self, name):
        return str(name) in self.seqs

    def __getitem__(self, name):
        return self.seqs[str(name)]

    def fetch(self, name, start=None, end=None):
        """Same contract as ``pysam.FastaFile.fetch`` (0-based half-open), returning ``str``."""
        s = self.seqs[str(name)]
        if start is None and end is None:
            return s.tobytes().decode()
        return s[(0 if start is None else int(start)):(s.shape[0] if end is None else int(end))].tobytes().decode()

    def lengths(self):
        return {n: int(self.seqs[n].shape[0]) for n in self.names}


def open_fasta(path, cache=True):
    """Load (and by default memoise per path + mtime) a FASTA file."""
    path = str(path)
    if not cache:
        return Fasta(path)
    key = (os.path.abspath(path), os.path.getmtime(path))
    with _CACHE_LOCK:
        fa = _CACHE.get(key)
    if fa is None:
        fa = Fasta(path)                      # (outside the lock: lanes load different files side by side)
        with _CACHE_LOCK:
            fa = _CACHE.setdefault(key, fa)
    return fa


def forget(path):
    """Drop a memoised file (a cohort's contig files are read once each: 3 GB of host memory per haplotype otherwise)."""
    path = os.path.abspath(str(path))
    with _CACHE_LOCK:
        for key in [k for k in list(_CACHE) if k[0] == path]:
            del _CACHE[key]


_FAI_CACHE = {}


def read_fai(fai_file_name):
    """``svpoplib.ref.get_df_fai`` contract (pavlib/inv.py:201): Series name -> length (memoised per path + mtime)."""
    import pandas as pd
    key = (os.path.abspath(str(fai_file_name)), os.path.getmtime(fai_file_name))
    if key in _FAI_CACHE:
        return _FAI_CACHE[key]
    df = pd.read_csv(fai_file_name, sep='\t', header=None, usecols=[0, 1], names=['CHROM', 'LEN'],
                     dtype={'CHROM': str, 'LEN': np.int64})
    _FAI_CACHE[key] = df.set_index('CHROM')['LEN']
    return _FAI_CACHE[key]
