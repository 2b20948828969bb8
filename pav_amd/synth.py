"""
Deterministic synthetic workloads for tests and bench.py (SURVEY.md section 8(d) profile).

PAV ships no generator or test data; everything here is this repo's own workload
tooling.  The per-base work (random soft-masked sequence, contig mutation with a base-exact =/X/I/D/H CIGAR)
is plain C in ``csrc/synth.c`` (``lib/libpavsynth.so``); this module plans segments, inversions, flagged
regions and produces the tables in the reference's file schemas:

* alignment BED columns: API_ALIGN.md:33-57 (``CALL_BATCH = INDEX % 10``: rules/align.snakefile:163)
* flagged-region BED columns: rules/call_inv.snakefile:420-470

Seeds follow SURVEY.md: config ``n`` uses seed ``1000 + n``; haplotype ``h`` uses ``seed * 64 + h``.
"""

import ctypes
import gzip
import os
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass, field

import numpy as np
import pandas as pd

_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib', 'libpavsynth.so')
        if not os.path.exists(path):
            raise RuntimeError(f'{path} is missing: run `python -c "import __graft_entry__ as g; g.build()"` first')
        lib = ctypes.CDLL(path)
        lib.pavsynth_random_seq.argtypes = [ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_double]
        lib.pavsynth_random_seq.restype = ctypes.c_int
        lib.pavsynth_revcomp.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
        lib.pavsynth_revcomp.restype = None
        lib.pavsynth_contig.argtypes = [
            ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p,
            ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int,
            ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p,
            ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
        lib.pavsynth_contig.restype = ctypes.c_int
        _LIB = lib
    return _LIB


class _Params(ctypes.Structure):
    _fields_ = [('snv_rate', ctypes.c_double), ('indel_rate', ctypes.c_double),
                ('pareto_alpha', ctypes.c_double), ('max_indel', ctypes.c_uint32),
                ('tandem_frac', ctypes.c_double), ('clip', ctypes.c_uint32), ('pair_frac', ctypes.c_double)]


# hg38 no-ALT primary assembly lengths (chr1..22, X, Y); T2T-CHM13v2.0 lengths.
HG38_LENGTHS = {
    'chr1': 248956422, 'chr2': 242193529, 'chr3': 198295559, 'chr4': 190214555, 'chr5': 181538259,
    'chr6': 170805979, 'chr7': 159345973, 'chr8': 145138636, 'chr9': 138394717, 'chr10': 133797422,
    'chr11': 135086622, 'chr12': 133275309, 'chr13': 114364328, 'chr14': 107043718, 'chr15': 101991189,
    'chr16': 90338345, 'chr17': 83257441, 'chr18': 80373285, 'chr19': 58617616, 'chr20': 64444167,
    'chr21': 46709983, 'chr22': 50818468, 'chrX': 156040895, 'chrY': 57227415,
}
CHM13_LENGTHS = {
    'chr1': 248387328, 'chr2': 242696752, 'chr3': 201105948, 'chr4': 193574945, 'chr5': 182045439,
    'chr6': 172126628, 'chr7': 160567428, 'chr8': 146259331, 'chr9': 150617247, 'chr10': 134758134,
    'chr11': 135127769, 'chr12': 133324548, 'chr13': 113566686, 'chr14': 101161492, 'chr15': 99753195,
    'chr16': 96330374, 'chr17': 84276897, 'chr18': 80542538, 'chr19': 61707364, 'chr20': 66210255,
    'chr21': 45090682, 'chr22': 51324926, 'chrX': 154259566, 'chrY': 62460029,
}

_COMP = np.arange(256, dtype=np.uint8)
for _a, _b in zip(b'ACGTRYSWKMBDHVNUacgtryswkmbdhvnu', b'TGCAYRSWMKVHDBNAtgcayrswmkvhdbna'):
    _COMP[_a] = _b


def revcomp(arr):
    """IUPAC-aware, case-preserving reverse complement of a uint8 ASCII array."""
    return _COMP[arr[::-1]]


@dataclass
class Inversion:
    chrom: str
    pos: int          # inverted interval [pos, end) on the reference (includes inverted-repeat flanks)
    end: int
    repeat: int       # length of each inverted-repeat flank (0 = none)


@dataclass
class Reference:
    names: list
    seqs: dict                     # name -> np.uint8 ASCII
    inversions: list = field(default_factory=list)

    @property
    def lengths(self):
        return {n: int(self.seqs[n].shape[0]) for n in self.names}


@dataclass
class Haplotype:
    hap: str
    ref: Reference
    tig_names: list
    tig_seqs: dict                 # name -> np.uint8 ASCII (stored orientation)
    df_align: pd.DataFrame         # trim-none alignment BED
    df_trim: pd.DataFrame          # trim-tigref alignment BED
    df_flag: pd.DataFrame          # flagged regions
    stats: dict

    @property
    def tig_lengths(self):
        return pd.Series({n: int(self.tig_seqs[n].shape[0]) for n in self.tig_names}, dtype=np.int64)


def make_reference(seed, lengths, case_run=300.0, n_every=50_000_000, n_len=10_000,
                   inv_every=25_000_000, inv_min=2_000, inv_max=200_000, threads=8):
    """Random soft-masked reference with N runs and planted inverted repeats for half the inversion loci."""
    lib = _lib()
    names = list(lengths)
    seqs = {n: np.empty(int(lengths[n]), dtype=np.uint8) for n in names}

    def fill(i):
        n = names[i]
        lib.pavsynth_random_seq(seed * 1000003 + i, seqs[n].ctypes.data, seqs[n].shape[0], float(case_run))

    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(fill, range(len(names))))

    rng = np.random.default_rng(seed)
    inversions = []
    for n in names:
        L = seqs[n].shape[0]
        # N runs
        n_runs = int(L // n_every) if n_every else 0
        for _ in range(n_runs):
            p = int(rng.integers(0, max(1, L - n_len)))
            seqs[n][p:p + n_len] = ord('N')
        # inversions
        n_inv = int(L // inv_every) if inv_every else 0
        taken = []
        for _ in range(n_inv):
            ln = int(np.exp(rng.uniform(np.log(inv_min), np.log(inv_max))))
            rep = 0
            if rng.random() < 0.5:
                rep = int(rng.integers(500, 5001))
                rep = min(rep, ln // 4)
            margin = 3 * ln + 20_000
            if L < ln + 2 * margin:
                continue
            p = int(rng.integers(margin, L - ln - margin))
            if any(p < e + margin and s < p + ln + margin for s, e in taken):
                continue
            if np.any(seqs[n][p - 10_000:p + ln + 10_000] == ord('N')):
                continue
            taken.append((p, p + ln))
            if rep:
                seqs[n][p + ln - rep:p + ln] = revcomp(seqs[n][p:p + rep])
            inversions.append(Inversion(n, p, p + ln, rep))
    inversions.sort(key=lambda v: (names.index(v.chrom), v.pos))
    return Reference(names, seqs, inversions)


def _plan_segments(rng, L, median, sigma, cap, zones=(), gap_min=1_000, gap_max=10_000, min_len=2_000):
    """Tile [0, L) with non-overlapping aligned segments separated by small unaligned gaps.

    ``zones`` are [start, end) intervals (an inversion plus the flank its scan may expand into) that no
    segment boundary may cut: a boundary falling inside one is pushed past its end.
    """
    def push(x):
        for zs, ze in zones:
            if zs <= x < ze:
                return ze
        return x

    segs = []
    p = push(int(rng.integers(0, gap_max)))
    while p < L - min_len:
        ln = int(min(cap, max(min_len, rng.lognormal(np.log(median), sigma))))
        e = min(L, push(p + ln))
        if e - p >= min_len:
            segs.append((p, e))
        p = push(e + int(rng.integers(gap_min, gap_max)))
    return segs


def make_haplotype(ref, seed, hap='h1', snv_rate=1.0e-3, indel_rate=2.0e-4, pareto_alpha=1.2, max_indel=5000,
                   tandem_frac=0.5, clip=100, seg_median=1_000_000, seg_sigma=1.4, seg_cap=150_000_000,
                   rev_frac=0.5, decoys_per_inv=9, min_decoys=0, flag_batches=60, threads=8, segments=None,
                   zone_factor=3, zone_pad=20_000, pair_frac=0.0):
    """One haplotype: contigs, alignment BEDs (trim-none and trim-tigref) and flagged regions."""
    lib = _lib()
    rng = np.random.default_rng(seed)
    params = _Params(snv_rate, indel_rate, pareto_alpha, max_indel, tandem_frac, clip, pair_frac)

    # Plan alignment rows: (chrom, pos, end, rev, inversions inside)
    plan = []
    for n in ref.names:
        L = ref.seqs[n].shape[0]
        invs = [v for v in ref.inversions if v.chrom == n]
        zones = [(max(0, v.pos - zone_factor * (v.end - v.pos) - zone_pad),
                  min(L, v.end + zone_factor * (v.end - v.pos) + zone_pad)) for v in invs]
        segs = segments[n] if segments is not None and n in segments else \
            _plan_segments(rng, L, seg_median, seg_sigma, seg_cap, zones)
        for s, e in segs:
            inside = [v for v, (zs, ze) in zip(invs, zones) if s <= zs and ze <= e]
            plan.append((n, s, e, bool(rng.random() < rev_frac), inside))

    n_rows = len(plan)
    tig_names = [f'tig{i:06d}' for i in range(n_rows)]
    tig_seqs = [None] * n_rows
    cigars = [None] * n_rows
    counts = np.zeros((n_rows, 6), dtype=np.uint64)

    def build(i):
        chrom, s, e, rev, inside = plan[i]
        seg = ref.seqs[chrom][s:e]
        L = e - s
        cap_t = int(L * 1.02) + 2 * clip + 200_000
        cap_c = max(4096, int(L * (snv_rate + indel_rate) * 40) + 16 * sum(v.end - v.pos for v in inside) + 65536)
        inv_arr = np.asarray([x for v in inside for x in (v.pos - s, v.end - s)], dtype=np.uint64)
        while True:
            tig = np.empty(cap_t, dtype=np.uint8)
            cig = ctypes.create_string_buffer(cap_c)
            tl = ctypes.c_uint64(0)
            rc = lib.pavsynth_contig(
                seed * 1000003 + i, seg.ctypes.data, L, ctypes.byref(params),
                inv_arr.ctypes.data if inv_arr.size else None, len(inside), int(rev),
                tig.ctypes.data, cap_t, ctypes.byref(tl), cig, cap_c, counts[i].ctypes.data)
            if rc == 0:
                break
            cap_t *= 2
            cap_c *= 2
        tig_seqs[i] = tig[:tl.value].copy()
        cigars[i] = cig.value.decode()

    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(build, range(n_rows)))

    rows = []
    for i, (chrom, s, e, rev, inside) in enumerate(plan):
        tl = int(tig_seqs[i].shape[0])
        rows.append((chrom, s, e, i, tig_names[i], clip, tl - clip, tl, 'NA', 'NA', 60, rev,
                     '0x0010' if rev else '0x0000', hap, cigars[i], i % 10))
    df_align = pd.DataFrame(rows, columns=[
        '#CHROM', 'POS', 'END', 'INDEX', 'QRY_ID', 'QRY_POS', 'QRY_END', 'QRY_LEN', 'RG', 'AO', 'MAPQ', 'REV',
        'FLAGS', 'HAP', 'CIGAR', 'CALL_BATCH'])
    # get_align_bed order: pavlib/align/align.py:786
    chrom_rank = {n: k for k, n in enumerate(sorted(ref.names))}
    df_align = df_align.iloc[np.lexsort((df_align['QRY_ID'].values, -df_align['END'].values,
                                         df_align['POS'].values,
                                         df_align['#CHROM'].map(chrom_rank).values))].reset_index(drop=True)

    # trim-tigref: same rows; every 7th row loses a few flank bases (exercises FILTER=TRIM), every 13th row is
    # dropped (reindex fill -1 => TRIM, rules/call.snakefile:813-842).
    df_trim = df_align.copy()
    for c in ('TRIM_REF_L', 'TRIM_REF_R', 'TRIM_QRY_L', 'TRIM_QRY_R'):
        df_trim[c] = 0
    # Only the POS/END columns are consumed on the hot path (call.snakefile:813); shrink them directly.
    shrink = (df_trim['INDEX'] % 7 == 3)
    df_trim.loc[shrink, 'POS'] += 700
    df_trim.loc[shrink, 'END'] -= 900
    df_trim.loc[shrink, 'TRIM_REF_L'] = 700
    df_trim.loc[shrink, 'TRIM_REF_R'] = 900
    df_trim = df_trim.loc[df_trim['INDEX'] % 13 != 11].reset_index(drop=True)

    # Flagged regions: planted inversions + decoys
    flag = []
    for (chrom, s, e, rev, inside) in plan:
        for v in inside:
            flag.append((chrom, v.pos, v.end, 'CLUSTER_SNV'))
    n_decoy = max(min_decoys, decoys_per_inv * len(flag))
    for _ in range(n_decoy):
        chrom, s, e, rev, inside = plan[int(rng.integers(0, n_rows))]
        ln = int(rng.integers(500, 5001))
        if e - s <= ln + 2:
            continue
        p = int(rng.integers(s, e - ln))
        flag.append((chrom, p, p + ln, 'CLUSTER_INDEL,MATCH_INDEL'))
    flag.sort(key=lambda t: (chrom_rank[t[0]], t[1]))
    df_flag = pd.DataFrame(
        [(c, p, e, f'{c}-{p}-RGN-{e - p}', 'RGN', e - p, t, 0, 0, True, k % flag_batches)
         for k, (c, p, e, t) in enumerate(flag)],
        columns=['#CHROM', 'POS', 'END', 'ID', 'SVTYPE', 'SVLEN', 'TYPE', 'COUNT_INDEL', 'COUNT_SNV', 'TRY_INV',
                 'BATCH'])

    stats = dict(n_aln=n_rows, n_ops=int(counts[:, 0].sum()), n_snv=int(counts[:, 1].sum()),
                 n_ins=int(counts[:, 2].sum()), n_del=int(counts[:, 3].sum()),
                 aligned_bp=int(counts[:, 4].sum()), n_inv=sum(len(p[4]) for p in plan), n_flag=len(flag))
    return Haplotype(hap, ref, tig_names, dict(zip(tig_names, tig_seqs)), df_align, df_trim, df_flag, stats)


def write_fasta(path, names, seqs, line=0):
    """Write FASTA (+ .fai).  ``line=0`` writes each record on one line.  ``.gz`` paths are gzip'd."""
    opener = gzip.open if path.endswith('.gz') else open
    off = 0
    fai = []
    with opener(path, 'wb') as fh:
        for n in names:
            hdr = f'>{n}\n'.encode()
            fh.write(hdr)
            off += len(hdr)
            s = seqs[n]
            L = int(s.shape[0])
            w = L if not line else line
            fai.append(f'{n}\t{L}\t{off}\t{w}\t{w + 1}\n')
            if not line:
                fh.write(s.tobytes())
                fh.write(b'\n')
                off += L + 1
            else:
                for i in range(0, L, line):
                    fh.write(s[i:i + line].tobytes())
                    fh.write(b'\n')
                off += L + (L + line - 1) // line
    with open(path + '.fai', 'w') as fh:
        fh.writelines(fai)


def bgzip(src, dst, level=6, threads=8, block=65280):
    """``src`` rewritten as BGZF (SAM specification 4.1: gzip members of ``block`` bytes of text each, the end-of-file member last) -
    the form PAV keeps its FASTA files in (``bgzip``; rules/align.snakefile).  zlib releases the GIL: the members are made by
    ``threads`` threads, a few hundred at a time."""
    import struct
    import zlib
    from concurrent.futures import ThreadPoolExecutor

    def member(chunk):
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = c.compress(chunk) + c.flush()
        return (b'\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00' + struct.pack('<H', len(body) + 25) + body
                + struct.pack('<II', zlib.crc32(chunk), len(chunk)))

    with open(src, 'rb') as fi, open(dst, 'wb') as fo, ThreadPoolExecutor(max(1, threads)) as pool:
        while True:
            buf = fi.read(block * 64 * max(1, threads))
            if not buf:
                break
            for m in pool.map(member, [buf[a:a + block] for a in range(0, len(buf), block)]):
                fo.write(m)
        fo.write(member(b''))


def scaled_lengths(lengths, scale):
    return {n: max(20_000, int(L * scale)) for n, L in lengths.items()}


# ---------------------------------------------------------------------------------------------------------
# Config presets (BASELINE.json:configs, restated in SURVEY.md section 8(d))
# ---------------------------------------------------------------------------------------------------------

def config1(seed=1001):
    """One 1 Mb contig vs a 1 Mb "chr20 slice"; 1 row, ~2.4 k ops (plumbing case)."""
    ref = make_reference(seed, {'chr20': 1_000_000}, n_every=0, inv_every=0, threads=1)
    hap = make_haplotype(ref, seed * 64, 'h1', segments={'chr20': [(0, 1_000_000)]}, rev_frac=0.0, threads=1,
                         min_decoys=0)
    return hap


def large_inversions(seed=1049):
    """Two chromosomes with one large planted inversion each, aligned end to end by one contig each - the case behind
    tests/golden/inv_large (regions of 337 / 462 kbp after two expansion rounds, `pavlib/inv.py:310-351`).
    Returns (ref, hap, flags); flags = [(chrom, pos, end, type, scan kwargs)] - the flagged regions sit INSIDE the
    inversions so that the scan has to expand.

      chrA 1.0 Mbp  inversion 400 000 - 550 000 (150 kb, 3 kb inverted-repeat flanks), contig stored forward,
                    flagged at its centre: 54 k -> 135 k -> 337 kbp
      chrB 1.3 Mbp  inversion 500 000 - 700 000 (200 kb, no repeat), a 5 kb N run at 360 000, contig stored
                    reverse-complemented, flagged off centre (the contig carries the mirror image of the interval, so a
                    region shares k-mers with it only where it overlaps its own mirror image about 600 000):
                    74 k (REV only) -> 185 k (FWD flank on one side: states [0, 2]) -> 462 kbp
    """
    lengths = {'chrA': 1_000_000, 'chrB': 1_300_000}
    ref = make_reference(seed, lengths, n_every=0, inv_every=0, threads=1)
    ref.seqs['chrB'][360_000:365_000] = ord('N')
    plan = [Inversion('chrA', 400_000, 550_000, 3_000), Inversion('chrB', 500_000, 700_000, 0)]
    for v in plan:
        s = ref.seqs[v.chrom]
        if v.repeat:
            s[v.end - v.repeat:v.end] = revcomp(s[v.pos:v.pos + v.repeat])
    ref.inversions = plan
    # the strand of a row is drawn from the haplotype's generator; seed 1049 gives chrA forward, chrB reverse
    hap = make_haplotype(ref, seed * 64, 'h1', segments={n: [(0, L)] for n, L in lengths.items()}, rev_frac=0.5,
                         threads=1, decoys_per_inv=0, zone_factor=1, zone_pad=1_000)
    assert hap.df_align['REV'].tolist() == [False, True], 'large_inversions: the generator no longer draws these strands'
    flags = [('chrA', 450_000, 500_000, 'CLUSTER_SNV', None),
             ('chrB', 535_000, 605_000, 'CLUSTER_SNV', None),
             ('chrB', 520_000, 560_000, 'CLUSTER_SNV', None),         # shares no k-mer with its mirror image
             ('chrB', 358_000, 361_000, 'CLUSTER_INDEL', None)]       # beside / inside the N run, FWD only
    return ref, hap, flags


def config2(seed=1002, scale=1.0, hap_index=0, ref=None, lengths=None, threads=8, n_runs=True, **kw):
    """One haplotype vs an hg38-shaped reference (24 sequences); ``scale`` shrinks every length.  ``n_runs=False``: a
    reference without N runs (a T2T assembly)."""
    if ref is None:
        n_every = int(50_000_000 * max(scale, 0.02)) if scale < 1 else 50_000_000
        ref = make_reference(seed, scaled_lengths(lengths or HG38_LENGTHS, scale), threads=threads,
                             n_every=n_every if n_runs else 0,
                             inv_every=int(25_000_000 * max(scale, 0.02)) if scale < 1 else 25_000_000)
    med = max(20_000, int(1_000_000 * min(1.0, scale * 10)))
    return make_haplotype(ref, seed * 64 + hap_index, f'h{hap_index + 1}', seg_median=med, threads=threads, **kw)


def config5(seed=1005, scale=1.0, hap_index=0, ref=None, threads=8, **kw):
    """BASELINE.json configs[4]: one haplotype of the synthetic cohort against a
    T2T-CHM13-shaped reference - 24 sequences with the CHM13v2.0 lengths (3.1 Gbp), no N runs; the cohort is
    ``hap_index`` = 0..63 (32 samples x h1 / h2), batched 8 per GPU against ONE resident reference."""
    return config2(seed=seed, scale=scale, hap_index=hap_index, ref=ref, lengths=CHM13_LENGTHS, threads=threads,
                   n_runs=False, **kw)


# ---------------------------------------------------------------------------------------------------------
# Untrimmed alignment tables with overlapping records (input of the trimming rules, rules/align.snakefile:54-97).
# Trimming reads coordinates and CIGAR strings only, so no sequence is generated.
# ---------------------------------------------------------------------------------------------------------

def _random_cigar_ops(rng, qry_len, snv_rate, indel_rate, max_indel, ragged):
    """CIGAR operations (len, op) consuming exactly ``qry_len`` query bases; returns (ops, ref_bp)."""
    ops, q, ref = [], 0, 0
    first = True
    while q < qry_len:
        left = qry_len - q
        run = int(min(left, rng.geometric(min(0.5, snv_rate + indel_rate))))
        if first and ragged and rng.random() < 0.5:
            run = 0                                                   # alignment starts on a variant
        first = False
        if run:
            ops.append((run, '='))
            q += run
            ref += run
        left = qry_len - q
        if not left:
            break
        u = rng.random() * (snv_rate + indel_rate)
        if u < snv_rate:
            n = int(min(left, 1 + rng.integers(0, 3)))
            ops.append((n, 'X'))
            q += n
            ref += n
        elif rng.random() < 0.5:
            n = int(min(left, 1 + rng.integers(0, max_indel)))
            ops.append((n, 'I'))
            q += n
        else:
            n = int(1 + rng.integers(0, max_indel))
            ops.append((n, 'D'))
            ref += n
    while not ragged and ops and ops[-1][1] in 'ID' and len(ops) > 1:    # aligners do not end on an indel
        n, op = ops.pop()
        if op == 'I':
            ops.append((n, '='))
            ref += n
        else:
            ref -= n
    merged = []
    for n, op in ops:
        if merged and merged[-1][1] == op:
            merged[-1] = (merged[-1][0] + n, op)
        else:
            merged.append((n, op))
    return merged, ref


def make_overlap_table(seed, n_tigs=40, chroms=('chr1', 'chr10', 'chr2'), tig_len=(40_000, 200_000), snv_rate=2e-3,
                       indel_rate=1e-3, max_indel=60, max_overlap=4_000, short_frac=0.08, soft_frac=0.15, hap='h1'):
    """Alignment table (trim-none schema + TRIM_* = 0, rules/align.snakefile:166-169) whose records overlap in query and in
    reference space the way split alignments around SVs do: neighbouring records of a contig share up to ``max_overlap``
    query bases (same or different chromosome / strand, sometimes one inside the other), records of different contigs pile
    up on the same reference intervals, some records are shorter than the minimum aligned length.  Returns
    ``(DataFrame, Series of contig lengths)``."""
    import pandas as pd
    rng = np.random.default_rng(seed)
    rows, tig_lens = [], {}
    placed = []                                                        # (chrom, pos, end) of earlier records
    index = 0
    for t in range(n_tigs):
        tig = 'tig%06d' % t
        L = int(rng.integers(tig_len[0], tig_len[1]))
        tig_lens[tig] = L
        k = int(rng.integers(1, 6))
        edge = sorted(int(x) for x in rng.integers(200, L - 200, size=k - 1)) if k > 1 else []
        cuts = [int(rng.integers(0, 150))] + edge + [L - int(rng.integers(0, 150))]
        prev = None
        for s in range(k):
            qs, qe = cuts[s], cuts[s + 1]
            if qe - qs < 50:
                continue
            ov_l = int(rng.integers(0, max_overlap)) if s > 0 and rng.random() < 0.8 else 0
            ov_r = int(rng.integers(0, max_overlap)) if s + 1 < k and rng.random() < 0.8 else 0
            if rng.random() < 0.1:
                ov_l *= 8                                              # may swallow the neighbour (containment)
            qs, qe = max(0, qs - ov_l), min(L, qe + ov_r)
            if rng.random() < short_frac:
                qe = min(qe, qs + int(rng.integers(60, 990)))
            ragged = rng.random() < 0.25
            ops, ref_bp = _random_cigar_ops(rng, qe - qs, snv_rate, indel_rate, max_indel, ragged)
            mode = rng.random()
            if prev is not None and mode < 0.45:                       # same chromosome and strand, next to the previous record
                chrom, rev = prev['#CHROM'], prev['REV']
                shift = int(rng.integers(-max_overlap, max_overlap))
                pos = (prev['END'] + shift) if not rev else (prev['POS'] - ref_bp - shift)
            elif prev is not None and mode < 0.6:                      # same chromosome, other strand
                chrom, rev = prev['#CHROM'], not prev['REV']
                pos = prev['POS'] + int(rng.integers(-20_000, 20_000))
            elif placed and mode < 0.8:                                # on top of a record of another contig
                chrom, p0, p1 = placed[int(rng.integers(0, len(placed)))]
                rev = bool(rng.integers(0, 2))
                pos = p0 + int(rng.integers(-ref_bp // 2, max(1, (p1 - p0) // 2 + 1)))
            else:
                chrom, rev = chroms[int(rng.integers(0, len(chroms)))], bool(rng.integers(0, 2))
                pos = int(rng.integers(0, 5_000_000))
            pos = max(0, int(pos))
            lead, trail = (L - qe, qs) if rev else (qs, L - qe)
            clip = 'S' if rng.random() < soft_frac else 'H'
            cigar = ('%d%s' % (lead, clip) if lead else '') + ''.join('%d%s' % o for o in ops) + ('%d%s' % (trail, clip) if trail else '')
            row = {'#CHROM': chrom, 'POS': pos, 'END': pos + ref_bp, 'INDEX': index, 'QRY_ID': tig, 'QRY_POS': qs, 'QRY_END': qe,
                   'QRY_LEN': L, 'RG': 'NA', 'AO': 'NA', 'MAPQ': 60, 'REV': bool(rev), 'FLAGS': '0x0000' if s == 0 else '0x0800',
                   'HAP': hap, 'CIGAR': cigar, 'CALL_BATCH': t % 10, 'TRIM_REF_L': 0, 'TRIM_REF_R': 0, 'TRIM_QRY_L': 0, 'TRIM_QRY_R': 0}
            rows.append(row)
            placed.append((chrom, pos, pos + ref_bp))
            prev = row
            index += 1
    df = pd.DataFrame(rows)
    df = df.sort_values(['#CHROM', 'POS']).reset_index(drop=True)      # get_align_bed order (align.py:280)
    return df, pd.Series(tig_lens, name='LEN')


def split_overlaps(df_align, seed, max_overlap_ops=60, pieces=(2, 4)):
    """Untrimmed-looking table from a clean one: every record is cut into 2..4 consecutive records whose CIGARs share up to
    ``max_overlap_ops`` operations (the aligner's view of a repeat-mediated breakpoint: same chromosome and strand,
    overlapping in contig *and* reference space - the "try both trim orders" branch, trim.py:128-197).  Works on the CIGAR
    text (numpy), so it scales to the bench haplotype.  Adds TRIM_* = 0; INDEX is renumbered."""
    import pandas as pd
    from .align.cigar import tokenize
    rng = np.random.default_rng(seed)
    out = []
    for _, row in df_align.iterrows():
        text = row['CIGAR']
        b = np.frombuffer(text.encode(), dtype=np.uint8)
        lens, ops = tokenize(text)
        op_end = np.flatnonzero((b < 48) | (b > 57)) + 1                # byte offset one past each operation
        op_start = np.concatenate(([0], op_end[:-1]))
        q_adv = np.where(np.isin(ops, np.frombuffer(b'=XIM', dtype=np.uint8)), lens, 0)
        r_adv = np.where(np.isin(ops, np.frombuffer(b'=XDM', dtype=np.uint8)), lens, 0)
        clip = np.isin(ops, np.frombuffer(b'SH', dtype=np.uint8))
        body = np.flatnonzero(~clip)
        lo, hi = int(body[0]), int(body[-1]) + 1                       # operations [lo, hi) are the aligned part
        lead = int(lens[:lo].sum())
        trail = int(lens[hi:].sum())
        qc = np.concatenate(([0], np.cumsum(q_adv[lo:hi])))           # query / reference consumed before body op i
        rc = np.concatenate(([0], np.cumsum(r_adv[lo:hi])))
        nb = hi - lo
        k = int(rng.integers(pieces[0], pieces[1] + 1))
        if nb < 40 * k:
            k = 1
        cuts = sorted(int(x) for x in rng.integers(10, nb - 10, size=k - 1)) if k > 1 else []
        bounds = [0] + cuts + [nb]
        is_match = ops[lo:hi] == ord('=')
        for p in range(k):
            a, e = bounds[p], bounds[p + 1]
            if p > 0:
                a = max(1, a - int(rng.integers(0, max_overlap_ops // 2 + 1)))
            if p + 1 < k:
                e = min(nb - 1, e + int(rng.integers(0, max_overlap_ops // 2 + 1)))
            while a > 0 and not is_match[a]:                            # records start and end on '=' like the aligner's
                a -= 1
            while e < nb and not is_match[e - 1]:
                e += 1
            q0, q1, r0, r1 = int(qc[a]), int(qc[e]), int(rc[a]), int(rc[e])
            new = row.copy()
            h_lead, h_trail = lead + q0, trail + int(qc[nb]) - q1
            new['CIGAR'] = ('%dH' % h_lead if h_lead else '') + text[int(op_start[lo + a]):int(op_end[lo + e - 1])] + ('%dH' % h_trail if h_trail else '')
            new['POS'], new['END'] = int(row['POS']) + r0, int(row['POS']) + r1
            if bool(row['REV']):
                new['QRY_POS'], new['QRY_END'] = int(row['QRY_END']) - q1, int(row['QRY_END']) - q0
            else:
                new['QRY_POS'], new['QRY_END'] = int(row['QRY_POS']) + q0, int(row['QRY_POS']) + q1
            out.append(new)
    df = pd.DataFrame(out).reset_index(drop=True)
    df['INDEX'] = np.arange(df.shape[0])
    for c in ('TRIM_REF_L', 'TRIM_REF_R', 'TRIM_QRY_L', 'TRIM_QRY_R'):
        df[c] = 0
    return df.sort_values(['#CHROM', 'POS']).reset_index(drop=True)


def make_truncating_table(hap, seed, sv_split_prob=0.7, low_mapq_frac=0.1):
    """Alignment table in which large events truncate alignments, the input of the large-SV caller
    (pavlib/lgsv.py:31-642): every record of ``hap.df_align`` is broken at some of its >= 50 bp insertions / deletions
    (the aligner's split around a large SV: reference gap = deletion, contig gap = insertion) and around planted
    inversions (+,-,+ triples with the middle record on the other strand, or just the two flanks).  Pure Python on the
    operation lists: meant for test-sized haplotypes."""
    import pandas as pd
    from .align.cigar import tokenize
    rng = np.random.default_rng(seed)
    inv_by_chrom = {}
    for iv in hap.ref.inversions:
        inv_by_chrom.setdefault(iv.chrom, []).append(iv)
    out = []

    def emit(row, ops, r0, q0, q1, rev, qlen, total_q, lead, trail, mapq=None):
        """One record from body operations ``ops`` that start r0 reference / q0 query bases into the parent's body."""
        r_len = sum(n for n, o in ops if o in '=XD')
        new = row.copy()
        h_lead, h_trail = lead + q0, trail + total_q - q1
        new['CIGAR'] = ('%dH' % h_lead if h_lead else '') + ''.join('%d%s' % o for o in ops) + ('%dH' % h_trail if h_trail else '')
        new['POS'], new['END'] = int(row['POS']) + r0, int(row['POS']) + r0 + r_len
        if rev:
            new['QRY_POS'], new['QRY_END'] = int(row['QRY_END']) - q1, int(row['QRY_END']) - q0
        else:
            new['QRY_POS'], new['QRY_END'] = int(row['QRY_POS']) + q0, int(row['QRY_POS']) + q1
        if mapq is not None:
            new['MAPQ'] = mapq
        out.append(new)

    for _, row in hap.df_align.iterrows():
        lens, codes = tokenize(row['CIGAR'])
        ops = [(int(n), chr(c)) for n, c in zip(lens, codes)]
        lead = sum(n for n, o in ops[:1] if o in 'SH')
        trail = sum(n for n, o in ops[-1:] if o in 'SH') if len(ops) > 1 else 0
        body = [o for o in ops if o[1] not in 'SH']
        rev, qlen = bool(row['REV']), int(row['QRY_LEN'])
        total_q = sum(n for n, o in body if o in '=XI')
        # cut points: (body op index, kind); inversion zones are split at reference offsets first
        zones = []
        for iv in inv_by_chrom.get(row['#CHROM'], []):
            if iv.pos - 200 >= int(row['POS']) and iv.end + 200 <= int(row['END']):
                zones.append((iv.pos - int(row['POS']), iv.end - int(row['POS']), float(rng.random())))
        zones.sort()
        pieces, cur, r, q = [], [], 0, 0                           # pieces: (ops, r0, q0, kind)
        r_start, q_start = 0, 0
        zi = 0
        in_zone = False
        stack = list(reversed(body))
        while stack:
            n, o = stack.pop()
            r_adv, q_adv = (n if o in '=XD' else 0), (n if o in '=XI' else 0)
            bound = None
            if zi < len(zones):
                bound = zones[zi][1] if in_zone else zones[zi][0]
            if bound is not None and o in '=X' and r < bound < r + n:        # split the operation at the zone boundary
                stack.append((r + n - bound, o))
                n = bound - r
                r_adv = q_adv = n
            if bound is not None and r == bound and (cur or in_zone):
                pieces.append((cur, r_start, q_start, 'zone' if in_zone else 'flank'))
                if in_zone:
                    zi += 1
                in_zone = not in_zone
                cur, r_start, q_start = [], r, q
            big = o in 'ID' and n >= 50 and not in_zone and cur and stack and rng.random() < sv_split_prob
            if big:
                pieces.append((cur, r_start, q_start, 'flank'))
                r += r_adv
                q += q_adv
                cur, r_start, q_start = [], r, q
                continue
            cur.append((n, o))
            r += r_adv
            q += q_adv
        if cur:
            pieces.append((cur, r_start, q_start, 'zone' if in_zone else 'flank'))
        zone_rank = iter(z[2] for z in zones)
        for p_ops, r0, q0, kind in pieces:
            while p_ops and p_ops[0][1] in 'ID':                  # records do not start / end on an indel
                n, o = p_ops.pop(0)
                r0 += n if o == 'D' else 0
                q0 += n if o == 'I' else 0
            while p_ops and p_ops[-1][1] in 'ID':
                p_ops.pop()
            if not p_ops:
                continue
            q1 = q0 + sum(n for n, o in p_ops if o in '=XI')
            mapq = 20 if rng.random() < low_mapq_frac else None
            if kind == 'flank':
                emit(row, p_ops, r0, q0, q1, rev, qlen, total_q, lead, trail, mapq)
                continue
            u = next(zone_rank)
            if u < 0.25:                                          # left aligned through, as the generator made it
                emit(row, p_ops, r0, q0, q1, rev, qlen, total_q, lead, trail, mapq)
            elif u < 0.5:                                         # the aligner dropped the inverted part: two flanks only
                continue
            else:                                                 # +,-,+: the middle record on the other strand
                mid = row.copy()
                span = q1 - q0
                if rev:
                    qa, qb = int(row['QRY_END']) - q1, int(row['QRY_END']) - q0
                else:
                    qa, qb = int(row['QRY_POS']) + q0, int(row['QRY_POS']) + q1
                m_rev = not rev
                h_lead, h_trail = (qlen - qb, qa) if m_rev else (qa, qlen - qb)
                mid['CIGAR'] = ('%dH' % h_lead if h_lead else '') + '%d=' % span + ('%dH' % h_trail if h_trail else '')
                mid['POS'], mid['END'] = int(row['POS']) + r0, int(row['POS']) + r0 + span
                mid['QRY_POS'], mid['QRY_END'], mid['REV'] = qa, qb, m_rev
                out.append(mid)
    df = pd.DataFrame(out).reset_index(drop=True)
    df['INDEX'] = np.arange(df.shape[0])
    for c in ('TRIM_REF_L', 'TRIM_REF_R', 'TRIM_QRY_L', 'TRIM_QRY_R'):
        df[c] = 0
    return df.sort_values(['#CHROM', 'POS']).reset_index(drop=True)
