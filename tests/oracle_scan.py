"""TEST INFRASTRUCTURE: a stand-in for the device context that answers the density calls of ``pav_amd.inv``'s Python scan
state machine (``_Scan`` / ``_drive``, the mirror of pavlib/inv.py:203-454, itself pinned on the reference's logs and calls)
with the CPU oracle's tables (oracle/pav_oracle_density.c, pinned on the reference's tables).  The pair is a scan of flagged
regions that never touches the GPU: an independent statement of what the native scan driver + the density kernels must
produce - calls, regions, logs - at any region size.  Only tests/ and tools that make golden digests import this."""
import io
import os
import tempfile
from concurrent.futures import ThreadPoolExecutor
from types import SimpleNamespace

from pav_amd import _lib, density as pavden, inv as pavinv


class OracleScanContext:
    """Implements the slice of ``_lib.Context`` that ``pav_amd.inv._drive`` / ``_Scan.characterise`` use."""

    def __init__(self, ref_names, ref_seqs, tig_names, tig_seqs, threads=1, pool=1):
        self.names = {_lib.PAV_ROLE_REF: list(ref_names), _lib.PAV_ROLE_TIG: list(tig_names)}
        self.ref = [ref_seqs[n] for n in ref_names]
        self.tig = [tig_seqs[n] for n in tig_names]
        self.threads, self.pool = threads, pool
        self.tables = []
        self.iterations = []              # every density job the scans asked for, in order, with what came back

    def seq_names(self, role):
        return self.names[role]

    def _one(self, job, params):
        from oracle import oracle
        p = oracle.den_params(k=params.k, min_informative=params.min_informative, min_state_count=params.min_state_count,
                              den_smooth=params.den_smooth, state_run_smooth=job.state_run_smooth,
                              state_run_delta=params.state_run_delta, max_ref_kmer_count=params.max_ref_kmer_count)
        return oracle.density(self.ref[job.ref_id][job.ref_pos:job.ref_end], self.tig[job.tig_id][job.tig_pos:job.tig_end],
                              bool(job.ref_rc), p, threads=self.threads)

    def density_batch(self, jobs, params):
        from oracle import oracle
        self.k = int(params.k)
        if self.pool > 1 and len(jobs) > 1:             # many small regions: one region per thread (ctypes drops the GIL)
            self.threads = 1
            with ThreadPoolExecutor(self.pool) as ex:
                self.tables = list(ex.map(lambda j: self._one(j, params), jobs))
        else:
            self.tables = [self._one(j, params) for j in jobs]
        out = []
        for job, o in zip(jobs, self.tables):
            fail = o['status'] == 125
            n = 0 if fail else o['n']
            o['runs'] = [] if fail or n == 0 else oracle.rl_encode(o['STATE'], o['INDEX'])
            out.append(SimpleNamespace(status=_lib.DEN_FAIL if fail else (_lib.DEN_OK if o['status'] == 0 else _lib.DEN_UNFINALISED),
                                       fail_kind=o['fail_kind'], n_rows=n, n_runs=len(o['runs']), max_count=o['max_count'],
                                       max_kmer=o['max_kmer'], n_unresolved=0, n_near_tie=0))
            self.iterations.append((job.ref_id, int(job.ref_pos), int(job.ref_end), job.tig_id, int(job.tig_pos), int(job.tig_end),
                                    int(job.ref_rc), n, [list(r) for r in o['runs']]))
        return out

    def density_runs(self, j, n_runs):
        assert n_runs == len(self.tables[j]['runs'])
        return list(self.tables[j]['runs'])

    def density_table(self, j, n_rows):
        o = self.tables[j]
        assert n_rows == o['n']
        return {c: o[c] for c in pavden.DENSITY_COLUMNS}

    def density_annotate(self, j, n_rows, ref_id, ref_up, ref_dn, qry_index_base, tig_up, tig_dn):
        from oracle import oracle
        o = self.tables[j]
        chrom = self.ref[ref_id]
        return oracle.annotate(o['KMER'], o['INDEX'], getattr(self, 'k', 31), qry_index_base, tig_up, tig_dn,
                               chrom[ref_up[0]:ref_up[1]], chrom[ref_dn[0]:ref_dn[1]])


def write_fai(path, names, lengths):
    with open(path, 'w') as fh:
        for n in names:
            fh.write(f'{n}\t{int(lengths[n])}\t0\t0\t0\n')


def oracle_scan(region_flags, ref_names, ref_seqs, tig_names, tig_seqs, align_lift, k_util, threads=1, pool=1, **scan_kwargs):
    """scan_for_inv_batch(native=False) with the oracle behind it.  -> (list of InvCall / None / RuntimeError, log text,
    the context with every iteration it served)."""
    ctx = OracleScanContext(ref_names, ref_seqs, tig_names, tig_seqs, threads=threads, pool=pool)
    d = tempfile.mkdtemp(prefix='pav_oracle_scan_')
    ref_fa = os.path.join(d, 'ref.fa')
    write_fai(ref_fa + '.fai', ref_names, {n: ref_seqs[n].shape[0] for n in ref_names})
    ref_index = {n: i for i, n in enumerate(ref_names)}
    tig_index = {n: i for i, n in enumerate(tig_names)}
    logs = [io.StringIO() for _ in region_flags]
    scans = [pavinv._Scan(rf, ref_fa, None, align_lift, k_util, scan_kwargs.get('n_tree'), scan_kwargs.get('max_region_size'),
                          logs[i], scan_kwargs.get('srs_tree'), scan_kwargs.get('min_exp_count', pavinv.DEFAULT_MIN_EXP_COUNT),
                          ref_index, tig_index) for i, rf in enumerate(region_flags)]
    pavinv._drive(ctx, scans, pavden.den_params(k=k_util.k_size))
    return [sc.error if sc.error is not None else sc.result for sc in scans], ''.join(b.getvalue() for b in logs), ctx


def call_record(call):
    """What a digest keeps of an InvCall: id + the six regions."""
    def rg(r):
        return [r.chrom, int(r.pos), int(r.end), bool(r.is_rev)]
    return {'id': call.id, 'svlen': int(call.svlen), 'ref_outer': rg(call.region_ref_outer), 'ref_inner': rg(call.region_ref_inner),
            'tig_outer': rg(call.region_tig_outer), 'tig_inner': rg(call.region_tig_inner),
            'ref_discovery': rg(call.region_ref_discovery), 'tig_discovery': rg(call.region_tig_discovery)}
