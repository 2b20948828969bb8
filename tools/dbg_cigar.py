import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import util
from pav_amd import _lib, cigarcall
d, df_align, df_trim = util.golden_case(sys.argv[1] if len(sys.argv) > 1 else 'cigar_synth')
ctx = _lib.Context(0)
ref_fa, tig_fa = util.seq_arrays(d, None)
ctx.seq_load(_lib.PAV_ROLE_REF, ref_fa.names, [ref_fa[n] for n in ref_fa.names])
ctx.seq_load(_lib.PAV_ROLE_TIG, tig_fa.names, [tig_fa[n] for n in tig_fa.names])
aln, text, off = cigarcall.pack_alignments(df_align, ref_fa.names, tig_fa.names)
print('T', text.shape, 'rows', aln.shape, off)
ctx.cigar_load(aln, text, off)
os.environ['PAV_CIGAR_STAGE'] = 'scan'
c = ctx.cigar_call()
print('counts', c.n_ops, c.n_snv, c.n_indel, c.seq_bytes, c.aligned_bases)
ops, op_off = ctx.cigar_fetch_ops(c.n_ops, aln.shape[0])
from oracle import oracle
o_snv, o_indel, o_blob, err = util.oracle_records(ref_fa.names, [ref_fa[n] for n in ref_fa.names], tig_fa.names, [tig_fa[n] for n in tig_fa.names], df_align)
print('oracle', o_snv.shape, o_indel.shape, o_blob.shape)
exp = []
for r in range(aln.shape[0]):
    rc, tuples, _, _ = oracle.cigar_tokenize(text[off[r]:off[r+1]].tobytes().decode())
    exp.append(tuples)
codes = 'MIDNSHP=X'
flat = [ (l << 4) | codes.index(o) for tp in exp for l, o in tp]
print('ops equal', np.array_equal(ops, np.array(flat, dtype=np.uint32)), len(flat))
print('op_off', op_off, np.cumsum([0] + [len(t) for t in exp]))
del os.environ['PAV_CIGAR_STAGE']
os.environ['PAV_SYNC_EACH'] = '1'
os.environ['PAV_CIGAR_STAGE'] = 'indel'
c = ctx.cigar_call()
snv, indel, blob = ctx.cigar_fetch(c)
print('snv equal', snv.tobytes() == o_snv.tobytes())
for f in ('aln', 'op_index', 'svlen', 'seq_off', 'svtype'):
    bad = np.flatnonzero(indel[f] != o_indel[f])
    print('stub', f, 'mismatches', bad.size, bad[:5], indel[f][bad[:5]], o_indel[f][bad[:5]])
print('stub pos range', indel['pos'].min(), indel['pos'].max(), 'qry', indel['qry_pos'].min(), indel['qry_pos'].max(), 'cap', indel['left_shift'].max())
dels = indel['svtype'] == 1
print('DEL pos equal', np.array_equal(indel['pos'][dels], o_indel['pos'][dels]))
del os.environ['PAV_CIGAR_STAGE']
c = ctx.cigar_call()
print('full call ok')
snv, indel, blob = ctx.cigar_fetch(c)
print('snv equal', snv.tobytes() == o_snv.tobytes(), 'blob equal', blob.tobytes() == o_blob.tobytes())
for f in o_indel.dtype.names:
    if f != 'pad' and not np.array_equal(indel[f], o_indel[f]):
        bad = np.flatnonzero(indel[f] != o_indel[f])
        print('indel', f, 'differs at', bad[:5], indel[f][bad[:5]], o_indel[f][bad[:5]])
if snv.tobytes() != o_snv.tobytes():
    for f in o_snv.dtype.names:
        bad = np.flatnonzero(snv[f] != o_snv[f])
        if bad.size: print('snv', f, bad.size, bad[:5], snv[f][bad[:5]], o_snv[f][bad[:5]])
