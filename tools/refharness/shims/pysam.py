"""Oracle-harness stand-in for the absent `pysam` module (this container only).

Only what the reference's hot path touches: ``pysam.FastaFile(path).fetch(name, start, end)``
as a context manager (pavlib/cigarcall.py:59-66, pavlib/seq.py:339-351).  Written for this
repo's golden-vector generator; it is never shipped with, nor imported by, the product.
"""
import gzip


class FastaFile:
    _cache = {}

    def __init__(self, path):
        import os
        self.path = str(path)
        self._fai = None
        # a large plain file with an index beside it (the full-size haplotype of gen_golden_fullsize_loci.py: 3 GB per file) is
        # read by seeking, as pysam itself does; everything else is held in memory
        if not self.path.endswith('.gz') and os.path.exists(self.path + '.fai') and os.path.getsize(self.path) > (256 << 20):
            self._fai = {}
            with open(self.path + '.fai') as fh:
                for line in fh:
                    f = line.rstrip('\n').split('\t')
                    self._fai[f[0]] = (int(f[1]), int(f[2]), int(f[3]), int(f[4]))
            self._seqs = self._fai
            return
        if self.path not in FastaFile._cache:
            FastaFile._cache[self.path] = self._read(self.path)
        self._seqs = FastaFile._cache[self.path]

    @staticmethod
    def _read(path):
        opener = gzip.open if path.endswith('.gz') else open
        seqs, name, chunks = {}, None, []
        with opener(path, 'rt') as fh:
            for line in fh:
                line = line.rstrip('\n')
                if line.startswith('>'):
                    if name is not None:
                        seqs[name] = ''.join(chunks)
                    name, chunks = line[1:].split()[0], []
                elif line:
                    chunks.append(line)
        if name is not None:
            seqs[name] = ''.join(chunks)
        return seqs

    @property
    def references(self):
        return list(self._seqs)

    def fetch(self, reference=None, start=None, end=None):
        if self._fai is not None:
            length, offset, line_bases, line_width = self._fai[str(reference)]
            a = 0 if start is None else max(0, int(start))
            b = length if end is None else min(length, int(end))
            if b <= a:
                return ''
            with open(self.path, 'rb') as fh:
                fh.seek(offset + a + (a // line_bases) * (line_width - line_bases))
                first = b - 1
                n_bytes = (first + (first // line_bases) * (line_width - line_bases)) - (a + (a // line_bases) * (line_width - line_bases)) + 1
                return fh.read(n_bytes).replace(b'\n', b'').decode()
        seq = self._seqs[str(reference)]
        if start is None and end is None:
            return seq
        return seq[(0 if start is None else int(start)):(len(seq) if end is None else int(end))]

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


# ---- pysam.AlignmentFile over SAM text ---------------------------------------------------------------------------------
# What pavlib.align.get_align_bed (pavlib/align/align.py:666-794) reads from a record, restated from the SAM specification
# and the htslib / pysam sources (sam.c bam_endpos / bam_cigar2rlen, libcalignedsegment.pyx getQueryStart / getQueryEnd):
# written for the golden-vector generator only (tools/refharness/gen_golden_align.py).  It pins the reference's own logic on
# top of these semantics; pysam itself is not available in this image.
_CIGAR_OPS = 'MIDNSHP=XB'


class AlignedSegment:
    def __init__(self, fields):
        self.query_name = fields[0]
        self.flag = int(fields[1])
        self.reference_name = fields[2]
        self.reference_start = int(fields[3]) - 1
        self.mapping_quality = int(fields[4])
        self._seq_len = 0 if fields[9] == '*' else len(fields[9])
        self.cigartuples = None
        if fields[5] != '*':
            self.cigartuples, num = [], ''
            for ch in fields[5]:
                if ch.isdigit():
                    num += ch
                else:
                    self.cigartuples.append((_CIGAR_OPS.index(ch), int(num)))
                    num = ''
        self._tags = []
        for f in fields[11:]:
            tag, typ, val = f.split(':', 2)
            self._tags.append((tag, int(val) if typ == 'i' else float(val) if typ == 'f' else val))

    @property
    def cigar(self):
        return [] if self.cigartuples is None else list(self.cigartuples)

    @property
    def is_unmapped(self):
        return bool(self.flag & 4)

    @property
    def is_reverse(self):
        return bool(self.flag & 16)

    def get_tags(self):
        return list(self._tags)

    @property
    def reference_end(self):                                            # bam_endpos
        if self.is_unmapped or not self.cigartuples:
            return None
        rlen = sum(n for op, n in self.cigartuples if op in (0, 2, 3, 7, 8))
        return self.reference_start + (rlen if rlen else 1)

    def _query_length(self):
        if self._seq_len:
            return self._seq_len
        return sum(n for op, n in self.cigartuples if op in (0, 1, 4, 7, 8))

    @property
    def query_alignment_start(self):                                    # getQueryStart
        start, qlen = 0, self._query_length()
        for op, n in self.cigartuples or []:
            if op == 5:
                if start != 0 and start != qlen:
                    raise ValueError('Invalid clipping in CIGAR string')
            elif op == 4:
                start += n
            else:
                break
        return start

    @property
    def query_alignment_end(self):                                      # getQueryEnd
        qlen = self._query_length()
        end = qlen
        ops = self.cigartuples or []
        for k in range(len(ops) - 1, 0, -1):
            op, n = ops[k]
            if op == 5:
                if end != qlen:
                    raise ValueError('Invalid clipping in CIGAR string')
            elif op == 4:
                end -= n
            else:
                break
        return end


class AlignmentFile:
    def __init__(self, path, mode='r', **kwargs):
        self.path = str(path)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def __iter__(self):
        with open(self.path, 'rb') as fh:
            magic = fh.read(2)
        opener = gzip.open if magic == b'\x1f\x8b' else open
        with opener(self.path, 'rt') as fh:
            for line in fh:
                line = line.rstrip('\r\n')
                if not line or line.startswith('@'):
                    continue
                yield AlignedSegment(line.split('\t'))
