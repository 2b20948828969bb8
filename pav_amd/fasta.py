"""
Minimal FASTA access for the host side of the hot path.

Replaces the two ``pysam.FastaFile(...).fetch(...)`` uses on the path (pavlib/cigarcall.py:59-66,
pavlib/seq.py:339-351) without pysam: whole records are held as ``numpy.uint8`` ASCII arrays exactly as they
appear in the file (case preserved), because the device library takes plain byte pointers.
Plain and gzip/bgzip FASTA are both accepted.
"""

import gzip
import os

import numpy as np

_CACHE = {}


class Fasta:
    """All records of one FASTA file, name -> uint8 ASCII array (line breaks removed, case preserved)."""

    def __init__(self, path):
        self.path = str(path)
        self.names = []
        self.seqs = {}
        self._load()

    def _load(self):
        opener = gzip.open if self.path.endswith('.gz') else open
        with opener(self.path, 'rb') as fh:
            data = np.frombuffer(fh.read(), dtype=np.uint8)
        if data.size == 0:
            return
        # record starts: '>' at offset 0 or right after a newline
        gt = np.flatnonzero(data == ord('>'))
        gt = gt[(gt == 0) | (data[np.maximum(gt, 1) - 1] == ord('\n'))]
        nl = np.flatnonzero(data == ord('\n'))
        for i, s in enumerate(gt):
            e = gt[i + 1] if i + 1 < gt.size else data.size
            k = np.searchsorted(nl, s)
            hdr_end = nl[k] if k < nl.size and nl[k] < e else e
            name = data[s + 1:hdr_end].tobytes().decode().split()[0] if hdr_end > s + 1 else ''
            body = data[min(hdr_end + 1, e):e]
            n_nl = int(np.count_nonzero(body == ord('\n'))) if body.size else 0
            if n_nl == 1 and body[-1] == ord('\n') and not np.any(body == ord('\r')):
                body = body[:-1]
            elif n_nl > 0 or (body.size and np.any(body == ord('\r'))):
                body = body[(body != ord('\n')) & (body != ord('\r'))]
            self.names.append(name)
            self.seqs[name] = np.ascontiguousarray(body)

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
    if key not in _CACHE:
        _CACHE[key] = Fasta(path)
    return _CACHE[key]


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
