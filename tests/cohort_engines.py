"""TEST INFRASTRUCTURE: a stand-in for pav_amd.cohort.DeviceEngine that needs no GPU, so that the plumbing of run_cohort - the
plan, the child processes, the stage order with its barriers, the merges of the reference's rules, the manifests - runs under
`-m "not gpu"` with two gloo ranks.  The work itself comes from the CPU oracle (CIGAR calls: oracle/pav_oracle_cigar.c; scans:
tests/oracle_scan.py) and, for the flag tables (which have no CPU restatement), from the golden tables the reference's rule
bodies wrote for the same inputs.  Imported by tests only."""
import gzip
import os
import shutil

import pandas as pd

import oracle_scan
import util
from pav_amd import rules
from pav_amd.fasta import open_fasta

GOLD = util.GOLD


class OracleEngine:
    def __init__(self, rank, device_id, ref_fa, config):
        self.rank, self.ref_fa, self.config = rank, ref_fa, dict(config or {})
        self.log = []

    def open(self):
        os.environ['PAV_INV_DRIVER'] = 'python'                 # the Python scan state machine, answered by the oracle below

    def close(self):
        pass

    def _golden_dir(self, job):
        return os.path.dirname(job.bed)

    def call_cigar_batches(self, job, P, batches):
        df_align = rules.read_align_bed(job.bed)
        df_trim = rules.read_trim_bed(job.bed_trim)
        for b in batches:
            df_snv, df_insdel = util.oracle_frames(self._golden_dir(job), df_align.loc[df_align['CALL_BATCH'] == b], df_trim, hap=job.hap)
            df_insdel.to_csv(P['cigar_batch_insdel'][b], sep='\t', index=False, compression='gzip')
            df_snv.to_csv(P['cigar_batch_snv'][b], sep='\t', index=False, compression='gzip')

    def flag_tables(self, job, P):
        for name in rules.FLAG_OUTPUTS:
            with open(os.path.join(self._golden_dir(job), name + '.tsv'), 'rb') as src, gzip.open(P[name], 'wb') as dst:
                shutil.copyfileobj(src, dst)
        if self.config.get('inv_sig_filter') == 'single_cluster':
            # _call_inv_accept_flagged_region with allow_single_cluster and no match_any accepts every locus; the batches are
            # dealt round in table order (rules/call_inv.snakefile:56-79, 458-466)
            df = pd.read_csv(P['flagged_regions'], sep='\t', keep_default_na=False)
            df['TRY_INV'] = True
            df['BATCH'] = [i % int(self.config.get('inv_sig_batch_count', 60)) for i in range(df.shape[0])]
            df.to_csv(P['flagged_regions'], sep='\t', index=False, compression='gzip')

    def _scan_ctx(self, job):
        ref, tig = open_fasta(self.ref_fa), open_fasta(job.tig_fa)
        ctx = oracle_scan.OracleScanContext(ref.names, {n: ref[n] for n in ref.names}, tig.names, {n: tig[n] for n in tig.names})
        ctx._inv_loaded = (str(self.ref_fa), str(job.tig_fa))
        ctx.handle = None
        return ctx

    def call_inv_batches(self, job, P, batches):
        ctx = self._scan_ctx(job)
        for b in batches:
            rules.call_inv_batch(P['flagged_regions'], job.bed_trim, job.tig_fa, job.tig_fa + '.fai', self.ref_fa, job.hap, b,
                                 bed_out=P['inv_batch'][b], log_path=P['inv_log'][b], density_out_dir=P['density_dir'], ctx=ctx)

    def call_haplotype(self, job, out_dir):
        """The unshared route: the same four stages on one rank."""
        batch_count = int(self.config.get('inv_sig_batch_count', 60))
        P = rules.haplotype_paths(out_dir, job.asm_name, job.hap, batch_count)
        rules._makedirs_for(P)
        self.call_cigar_batches(job, P, list(range(rules.CALL_CIGAR_BATCH_COUNT)))
        rules.call_cigar_merge(P['cigar_batch_insdel'], P['cigar_batch_snv'], P['insdel'], P['snv'])
        self.flag_tables(job, P)
        self.call_inv_batches(job, P, list(range(batch_count)))
        df = rules.call_inv_batch_merge(P['inv_batch'], P['inv'])
        return {'asm_name': job.asm_name, 'hap': job.hap, 'inv_calls': int(df.shape[0]),
                'files': {k: P[k] for k in ('snv', 'insdel', 'flagged_regions', 'inv')}}


def oracle_engine(rank, device_id, ref_fa, config):
    return OracleEngine(rank, device_id, ref_fa, config)


def golden_job(case, asm_name, hap='h1'):
    from pav_amd.cohort import HaplotypeJob
    d = os.path.join(GOLD, case)
    return HaplotypeJob(asm_name, hap, os.path.join(d, 'tig.fa'), os.path.join(d, 'align.tsv'), os.path.join(d, 'trim.tsv'))


def tree_text(root):
    """{relative file name: text} of every file under root (gzip files inflated: deflate streams differ between writers)."""
    out = {}
    for base, _, files in os.walk(root):
        for f in files:
            p = os.path.join(base, f)
            with open(p, 'rb') as fh:
                raw = fh.read()
            out[os.path.relpath(p, root)] = gzip.decompress(raw) if raw[:2] == b'\x1f\x8b' else raw
    return out
