import os, sys, time, threading
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
import __graft_entry__ as g
g.build_cpu_side()
from pav_amd import _lib, synth, cigarcall, fasta as pavfasta
import tempfile
work = tempfile.mkdtemp(prefix='pav_seq_')
hap = synth.config2(seed=1002, scale=1.0, threads=16, pair_frac=0.009)
ref_fa, tig_fa = os.path.join(work, 'ref.fa'), os.path.join(work, 'tig.fa')
synth.write_fasta(ref_fa, hap.ref.names, hap.ref.seqs, line=80)
synth.write_fasta(tig_fa, hap.tig_names, hap.tig_seqs, line=80)
df = hap.df_align.copy(); df['CALL_BATCH'] = df['INDEX'] % 10
bed, bed_trim = os.path.join(work, 'a.bed.gz'), os.path.join(work, 't.bed.gz')
df.to_csv(bed, sep='\t', index=False, compression={'method': 'gzip', 'compresslevel': 1})
hap.df_trim.to_csv(bed_trim, sep='\t', index=False, compression={'method': 'gzip', 'compresslevel': 1})
del hap
def T(name, f):
    t0 = time.perf_counter(); r = f(); print(f'{name}: {(time.perf_counter() - t0) * 1e3:.1f} ms', flush=True); return r
with _lib.Context(0) as ctx:
    for rep in range(2):
        print('--- rep', rep)
        t = T('BedTable(bed)', lambda: _lib.BedTable(bed, with_cigar=True))
        t2 = T('BedTable(trim)', lambda: _lib.BedTable(bed_trim, with_cigar=True))
        T('table.fetch', lambda: t.fetch())
        fa_t = T('open_fasta(tig)', lambda: pavfasta.open_fasta(tig_fa, cache=False))
        fa_r = T('open_fasta(ref)', lambda: pavfasta.open_fasta(ref_fa, cache=False))
        T('seq_load_fasta(ref)', lambda: ctx.seq_load_fasta(_lib.PAV_ROLE_REF, fa_r.native, fa_r.record_numbers(fa_r.names)))
        T('seq_load_fasta(tig)', lambda: ctx.seq_load_fasta(_lib.PAV_ROLE_TIG, fa_t.native, fa_t.record_numbers(fa_t.names)))
        T('close tables', lambda: (t.close(), t2.close()))
        def drop():
            global fa_t, fa_r
            fa_t = fa_r = None
        T('drop fasta', drop)
