"""
A cohort on the GPUs of one node: ``run_cohort(jobs, n_gpus, out_dir, ref_fa)``.

PAV treats haplotypes independently (README.md:83-86) and, inside a haplotype, its CALL_BATCH alignment batches (rule
call_cigar, rules/call.snakefile:792-846) and its flagged-region BATCHes (rule call_inv_batch, rules/call_inv.snakefile:115-311);
Snakemake runs them as separate jobs and exchanges files.  Here there is one process per GPU and no data-path collective
(SURVEY.md section 8(e)):

* ``len(jobs) >= n_gpus`` - BASELINE configs[3] (16 haplotypes -> 8 GPUs in two waves) and configs[4] (64 haplotypes, 8 per GPU):
  whole haplotypes are dealt to the ranks longest-processing-time first (cost = size of the alignment table, i.e. CIGAR text) and
  every rank runs :func:`pav_amd.rules.call_haplotype` on its haplotypes against ONE resident reference - one after the other, or
  ``config['pav_amd_lanes']`` of them at a time on contexts of their own (threads of the rank's process; configs[4]).
* fewer haplotypes than GPUs (or ``split=True``): the ranks that share a haplotype split its CALL_BATCH jobs, the lead rank
  merges them as rule call_cigar_merge does (:765-786) and flags the merged tables (the five flag rules), the ranks split the
  flagged-region BATCH jobs, the lead rank merges them as rule call_inv_batch_merge does (call_inv.snakefile:101-112, incl.
  ``drop_duplicates('ID')``).  Files are the exchange medium, as in the reference.

``torch.distributed`` carries the barriers between the stages and the manifests (host objects); RCCL / xGMI move no data.  The
ranks are either the processes of a launcher (``torch.distributed.run``: RANK / WORLD_SIZE in the environment) or started here,
as child processes, before anything in this process touches a GPU.

The work of a rank is done by an *engine* (``DeviceEngine``: the library on one GPU).  The engine is a parameter so that the
plumbing - planning, stage order, merges, manifests - can be driven on a machine without a GPU by a stand-in (tests); the
product default fails loudly without the HIP library and a GPU.
"""
import os
import socket
from dataclasses import dataclass

from . import rules, shard


SHARE_GPU_MAX_BYTES = 8 << 20       # run_cohort(share_gpu=True): alignment tables above this are refused (test mode only)


@dataclass
class HaplotypeJob:
    asm_name: str
    hap: str
    tig_fa: str                 # temp/{asm}/align/contigs_{hap}.fa.gz (+ .fai)
    bed: str                    # results/{asm}/align/trim-none/aligned_tig_{hap}.bed.gz
    bed_trim: str               # results/{asm}/align/trim-tigref/aligned_tig_{hap}.bed.gz
    cost: float = 0.0           # 0: the size of `bed` (CIGAR text dominates it)

    def weight(self):
        return float(self.cost) if self.cost else float(os.path.getsize(self.bed))


def plan(jobs, world, split=False):
    """-> ``world`` lists of work items ``(job number, part, n_parts, lead rank)``.  Whole haplotypes (n_parts == 1) by LPT when
    there are at least as many as ranks and ``split`` is off; otherwise every haplotype gets a contiguous group of ranks (in
    proportion to its weight, at least one) whose first member is its lead."""
    out = [[] for _ in range(world)]
    if not jobs:
        return out
    w = [j.weight() for j in jobs]
    if len(jobs) >= world and not split:
        for r, mine in enumerate(shard.assign_lpt(w, world)):
            out[r] = [(j, 0, 1, r) for j in mine]
        return out
    # groups of ranks per haplotype: largest-remainder apportionment of the ranks, every haplotype at least one; with more
    # haplotypes than ranks (split forced) the haplotypes are dealt round and share ranks whole
    if len(jobs) > world:
        for r, mine in enumerate(shard.assign_lpt(w, world)):
            out[r] = [(j, 0, 1, r) for j in mine]
        return out
    total = sum(w) or 1.0
    share = [max(1, int(world * x / total)) for x in w]
    while sum(share) > world:
        share[max(range(len(jobs)), key=lambda i: (share[i], -w[i]))] -= 1
    while sum(share) < world:
        share[max(range(len(jobs)), key=lambda i: w[i] / share[i])] += 1
    r0 = 0
    for j, n in enumerate(share):
        for p in range(n):
            out[r0 + p].append((j, p, n, r0))
        r0 += n
    return out


class DeviceEngine:
    """One rank's worker: the reference resident on one GPU and ``lanes`` contexts that read it (``pav_seq_share``) - a rank calls
    up to ``lanes`` of its haplotypes at the same time, one host thread each (BASELINE configs[4]: several haplotypes resident per
    GPU; the file stages of one haplotype - FASTA parse, text, deflate - run beside the kernels of another)."""

    def __init__(self, device_id, ref_fa, config=None, threads=0, gzip_level=0, lanes=1):
        self.device_id, self.ref_fa, self.config = int(device_id), ref_fa, dict(config or {})
        self.threads, self.gzip_level, self.lanes = threads, gzip_level, max(1, int(lanes))
        self.ctx, self.lane_ctx = None, []

    def open(self):
        from . import _lib, cigarcall
        self.ctx = _lib.Context(self.device_id)                  # raises without the HIP library / a GPU: there is no CPU path
        cigarcall.load_reference(self.ctx, self.ref_fa)
        self.lane_ctx = [self.ctx]
        for _ in range(1, self.lanes):
            c = _lib.Context(self.device_id)
            c.seq_share(self.ctx, _lib.PAV_ROLE_REF)             # the same planes in HBM; the mark of the resident file travels with them
            self.lane_ctx.append(c)

    def close(self):
        for c in self.lane_ctx[::-1]:
            c.close()
        self.ctx, self.lane_ctx = None, []
        from . import _lib
        _lib.device_pool_trim(self.device_id)                  # the rank is done with the GPU: nothing idle stays behind for other processes

    # ---- whole haplotypes ---------------------------------------------------------------------------------------------
    def call_haplotype(self, job, out_dir, ctx=None):
        return rules.call_haplotype(job.bed, job.bed_trim, job.tig_fa, self.ref_fa, job.asm_name, job.hap, out_dir, ctx=ctx or self.ctx,
                                    config=self.config, threads=self.threads, gzip_level=self.gzip_level)

    def call_haplotypes(self, jobs, out_dir):
        """All of a rank's whole haplotypes, ``lanes`` at a time; manifests in the order of ``jobs``."""
        if self.lanes == 1 or len(jobs) <= 1:
            return [self.call_haplotype(j, out_dir) for j in jobs]
        import queue
        import threading
        free = queue.Queue()
        for c in self.lane_ctx:
            free.put(c)
        out, errs = [None] * len(jobs), []

        def work(i):
            c = free.get()
            try:
                out[i] = self.call_haplotype(jobs[i], out_dir, ctx=c)
            except BaseException as ex:                          # noqa: BLE001 - raised again on the rank's main thread
                errs.append(ex)
            finally:
                free.put(c)
        sem = threading.Semaphore(self.lanes)

        def run(i):
            with sem:
                work(i)
        ths = [threading.Thread(target=run, args=(i,)) for i in range(len(jobs))]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if errs:
            raise errs[0]
        return out

    # ---- the jobs of a haplotype that several ranks share -------------------------------------------------------------------
    def call_cigar_batches(self, job, P, batches):
        for b in batches:
            rules.call_cigar_files(job.bed, job.bed_trim, job.tig_fa, self.ref_fa, job.hap, b, P['cigar_batch_insdel'][b],
                                   P['cigar_batch_snv'][b], ctx=self.ctx, threads=self.threads)

    def flag_tables(self, job, P):
        cfg = self.config
        rules.call_inv_cluster([P['insdel']], 'indel', P['cluster_indel'], ctx=self.ctx,
                               cluster_win=cfg.get('inv_sig_cluster_win', 200), cluster_min_snv=cfg.get('inv_sig_cluster_snv_min', 20),
                               cluster_min_indel=cfg.get('inv_sig_cluster_indel_min', 10))
        rules.call_inv_cluster([P['snv']], 'snv', P['cluster_snv'], ctx=self.ctx,
                               cluster_win=cfg.get('inv_sig_cluster_win', 200), cluster_min_snv=cfg.get('inv_sig_cluster_snv_min', 20),
                               cluster_min_indel=cfg.get('inv_sig_cluster_indel_min', 10))
        for vartype in ('sv', 'indel'):
            rules.call_inv_flag_insdel_cluster(P['insdel'], vartype, P['insdel_' + vartype], ctx=self.ctx,
                                               flank_cluster=cfg.get('inv_sig_insdel_cluster_flank', 2),
                                               flank_merge=cfg.get('inv_sig_insdel_merge_flank', 2000),
                                               cluster_min_svlen=cfg.get('inv_sig_cluster_svlen_min', 4))
        rules.call_inv_merge_flagged_loci(P['insdel_sv'], P['insdel_indel'], P['cluster_indel'], P['cluster_snv'], P['flagged_regions'],
                                          ctx=self.ctx, flank=cfg.get('inv_sig_merge_flank', 500),
                                          batch_count=cfg.get('inv_sig_batch_count', 60), inv_sig_filter=cfg.get('inv_sig_filter', 'svindel'))

    def call_inv_batches(self, job, P, batches):
        cfg = self.config
        for b in batches:
            rules.call_inv_batch(P['flagged_regions'], job.bed_trim, job.tig_fa, job.tig_fa + '.fai', self.ref_fa, job.hap, b,
                                 bed_out=P['inv_batch'][b], log_path=P['inv_log'][b], density_out_dir=P['density_dir'],
                                 k_size=cfg.get('inv_k_size', 31), inv_region_limit=cfg.get('inv_region_limit'),
                                 inv_min_expand=cfg.get('inv_min_expand'), srs_list=cfg.get('srs_list'), ctx=self.ctx)


def _default_engine(rank, device_id, ref_fa, config):
    return DeviceEngine(device_id, ref_fa, config, lanes=int((config or {}).get('pav_amd_lanes', 1)))


def _stage(world, rank, name, fn, sync):
    """Run one stage of a rank; with ``sync`` the ranks then exchange whether it failed anywhere (the exchange is the stage's
    barrier).  The failing rank raises its own exception, the others a RuntimeError that names it."""
    err = None
    try:
        fn()
    except BaseException as ex:                                  # noqa: BLE001 - exchanged, then raised again below
        err = ex
        if not (sync and world > 1) or not isinstance(ex, Exception):
            raise
    if sync and world > 1:
        import torch.distributed as dist
        seen = [None] * world
        dist.all_gather_object(seen, None if err is None else f'{type(err).__name__}: {err}')
        if err is not None:
            raise err
        bad = [(r, m) for r, m in enumerate(seen) if m is not None]
        if bad:
            raise RuntimeError(f"run_cohort: stage '{name}' failed on rank {bad[0][0]}: {bad[0][1]} (this is rank {rank})")


class _GpuTurn:
    """``share_gpu`` (several rank PROCESSES on GPU 0: tests and dry runs only): the ranks take turns with the device, one stage
    of one rank at a time, through a lock file (in the temporary directory, named after the output directory).  Two processes that keep one GPU busy at the same time
    are time-sliced by the driver and a pass can take 40 x as long (profiles/r04_cohort_two_ranks_one_gpu.json); a production
    run has one process per GPU (lanes are threads of it) and takes no lock."""

    def __init__(self, on, out_dir):
        import hashlib
        import tempfile
        key = hashlib.sha1(os.path.abspath(str(out_dir)).encode()).hexdigest()[:16]
        self.path = os.path.join(tempfile.gettempdir(), f'pav_amd_gpu_turn_{key}.lock') if on else None
        self.fh = None

    def __enter__(self):
        if self.path:
            import fcntl
            self.fh = open(self.path, 'a')
            fcntl.flock(self.fh, fcntl.LOCK_EX)
        return self

    def __exit__(self, *exc):
        if self.fh:
            import fcntl
            fcntl.flock(self.fh, fcntl.LOCK_UN)
            self.fh.close()
            self.fh = None
        return False


def run_rank(rank, world, jobs, out_dir, ref_fa, config=None, engine_factory=None, split=False, share_gpu=False):
    """The work of one rank (the process group, if any, is up).  Returns this rank's manifests: one per whole haplotype it ran,
    one per shared haplotype it leads."""
    import time
    t_rank = time.time()
    config = dict(config or {})
    batch_count = int(config.get('inv_sig_batch_count', 60))
    items = plan(jobs, world, split)
    mine = items[rank]
    engine = (engine_factory or _default_engine)(rank, 0 if share_gpu else rank, ref_fa, config)
    manifests = []
    shared = sorted({(j, n, lead) for its in items for (j, p, n, lead) in its if n > 1})
    turn = _GpuTurn(share_gpu and world > 1, out_dir)
    sync = True                                                # every stage ends with the ranks' error state exchanged (_stage)
    try:
        def opened():
            with turn:
                engine.open()                                  # (inside the try: contexts created before a failure are closed)
        _stage(world, rank, 'open', opened, sync)
        whole = [jobs[j] for (j, p, n, lead) in mine if n == 1]
        many = getattr(engine, 'call_haplotypes', None)
        done = []

        def whole_haplotypes():
            with turn:
                done.extend(many(whole, out_dir) if many else [engine.call_haplotype(j, out_dir) for j in whole])
        _stage(world, rank, 'whole haplotypes', whole_haplotypes, sync)
        for m in done:
            m['rank'], m['mode'] = rank, 'whole haplotype'
            manifests.append(m)
        # the shared haplotypes advance stage by stage; every rank meets every barrier (a rank without a part just passes)
        my_part = {j: (p, n) for (j, p, n, lead) in mine if n > 1}
        paths = {j: rules.haplotype_paths(out_dir, jobs[j].asm_name, jobs[j].hap, batch_count) for (j, n, lead) in shared}
        for j in my_part:
            rules._makedirs_for(paths[j])
        # (every stage ends with _stage's exchange of the ranks' error state in place of a bare barrier: a rank that fails
        #  takes its peers down with a message instead of leaving them in the barrier until the group times out)
        def stage1():                                                           # rule call_cigar, batches p, p + n, ...
            for j, (p, n) in my_part.items():
                with turn:
                    engine.call_cigar_batches(jobs[j], paths[j], list(range(p, rules.CALL_CIGAR_BATCH_COUNT, n)))

        def stage2():                                                           # (lead) call_cigar_merge + the five flag rules
            for (j, n, lead) in shared:
                if lead == rank:
                    P = paths[j]
                    rules.call_cigar_merge(P['cigar_batch_insdel'], P['cigar_batch_snv'], P['insdel'], P['snv'])
                    with turn:
                        engine.flag_tables(jobs[j], P)

        def stage3():                                                           # rule call_inv_batch, batches p, p + n, ...
            for j, (p, n) in my_part.items():
                with turn:
                    engine.call_inv_batches(jobs[j], paths[j], list(range(p, batch_count, n)))

        def stage4():                                                           # (lead) call_inv_batch_merge
            for (j, n, lead) in shared:
                if lead == rank:
                    P = paths[j]
                    df = rules.call_inv_batch_merge(P['inv_batch'], P['inv'])
                    manifests.append({'asm_name': jobs[j].asm_name, 'hap': jobs[j].hap, 'rank': rank, 'mode': f'shared by {n} ranks',
                                      'inv_calls': int(df.shape[0]), 'files': {k: P[k] for k in ('snv', 'insdel', 'flagged_regions', 'inv')}})
        if shared:
            for name, fn in (('call_cigar batches', stage1), ('call_cigar_merge + flagging', stage2),
                             ('call_inv_batch batches', stage3), ('call_inv_batch_merge', stage4)):
                _stage(world, rank, name, fn, True)
        # every manifest says which GPU its rank drove and how long the rank worked (a scaling line shows N distinct devices)
        dev = getattr(engine, 'ctx', None)
        info = {'rank_wall_s': round(time.time() - t_rank, 3), 'rank_cores': shard.effective_cpus(),
                'device_name': getattr(dev, 'device_name', '') if dev is not None else '',
                'pci_bus_id': getattr(dev, 'pci_bus_id', '') if dev is not None else ''}
        for m in manifests:
            m.update(info)
    finally:
        engine.close()
    return manifests


def _gather(manifests, world):
    if world <= 1:
        return manifests
    import torch.distributed as dist
    buf = [None] * world
    dist.all_gather_object(buf, manifests)
    return [m for part in buf for m in part]


def _free_port():
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def _child(rank, world, port, backend, args, queue):
    os.environ.update({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'RANK': str(rank), 'WORLD_SIZE': str(world),
                       'LOCAL_RANK': str(rank)})
    import torch.distributed as dist
    jobs, out_dir, ref_fa, config, engine_factory, split, share_gpu = args
    shard.first_contact(backend, rank, world, rank, share_gpu)     # group up, one all_reduce answered, one GPU per rank
    try:
        ms = _gather(run_rank(rank, world, jobs, out_dir, ref_fa, config, engine_factory, split, share_gpu), world)
        dist.barrier()
        if rank == 0:
            queue.put(ms)
    finally:
        dist.destroy_process_group()


def run_cohort(jobs, n_gpus, out_dir, ref_fa, config=None, engine_factory=None, split=False, backend=None, share_gpu=False,
               timeout=None):
    """Call every haplotype of ``jobs`` (:class:`HaplotypeJob`) on ``n_gpus`` GPUs of this node; files under ``out_dir`` with the
    reference's relative names (:func:`pav_amd.rules.haplotype_paths`).  Returns the manifests of all haplotypes (on every rank
    when the ranks were started by a launcher).

    ``backend``: process-group backend of the control plane - default ``'nccl'`` (RCCL) with one GPU per rank, ``'gloo'`` with
    ``share_gpu`` (tests: every rank on GPU 0) or without a GPU (stand-in engines).  ``engine_factory(rank, device_id, ref_fa,
    config)`` must be importable by the child processes (a module-level function).  The children are started with
    ``multiprocessing``'s *spawn* method: a script that calls this with ``n_gpus > 1`` needs the usual
    ``if __name__ == '__main__':`` guard around its top-level code."""
    jobs = [j if isinstance(j, HaplotypeJob) else HaplotypeJob(**j) for j in jobs]
    if share_gpu and n_gpus > 1:
        # several rank PROCESSES on one GPU are time-sliced by the driver: measured 504 s for what one process does in 12.4 s
        # (profiles/r04_cohort_two_ranks_one_gpu.json).  A mode for tests and dry runs: refused above a toy size.
        size = sum(os.path.getsize(j.bed) for j in jobs if os.path.exists(j.bed))
        if size > SHARE_GPU_MAX_BYTES and os.environ.get('PAV_AMD_SHARE_GPU_ANYWAY') != '1':
            raise ValueError(f'run_cohort(share_gpu=True): the alignment tables add up to {size} bytes (> {SHARE_GPU_MAX_BYTES}); rank '
                             f'processes that share one GPU are time-sliced and run ~40 x slower than one process with lanes '
                             f"(config['pav_amd_lanes']).  share_gpu is a test mode; PAV_AMD_SHARE_GPU_ANYWAY=1 overrides")
    os.makedirs(out_dir, exist_ok=True)
    if 'RANK' in os.environ and 'WORLD_SIZE' in os.environ and int(os.environ['WORLD_SIZE']) > 1:
        import torch.distributed as dist                       # the ranks exist already (torch.distributed.run)
        rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
        if world != n_gpus:
            raise ValueError(f'run_cohort(n_gpus={n_gpus}) inside a job of WORLD_SIZE={world}')
        if not dist.is_initialized():
            be = backend or ('gloo' if share_gpu else 'nccl')
            shard.first_contact(be, rank, world, int(os.environ.get('LOCAL_RANK', rank)), share_gpu)
        return _gather(run_rank(rank, world, jobs, out_dir, ref_fa, config, engine_factory, split, share_gpu), world)
    if n_gpus <= 1:
        return run_rank(0, 1, jobs, out_dir, ref_fa, config, engine_factory, split, share_gpu)
    import torch.multiprocessing as mp
    be = backend or ('gloo' if share_gpu else 'nccl')
    mctx = mp.get_context('spawn')                             # fresh interpreters: nothing of this process's GPU state is inherited
    queue = mctx.Queue()
    port = _free_port()
    args = (jobs, out_dir, ref_fa, config, engine_factory, split, share_gpu)
    procs = [mctx.Process(target=_child, args=(r, n_gpus, port, be, args, queue)) for r in range(n_gpus)]
    for p in procs:
        p.start()
    try:
        result = None
        import queue as _q
        waited = 0.0
        while result is None:
            try:
                result = queue.get(timeout=1.0)
            except _q.Empty:
                waited += 1.0
                dead = [p for p in procs if p.exitcode not in (None, 0)]
                if dead:
                    raise RuntimeError(f'run_cohort: rank process(es) failed with exit code(s) {[p.exitcode for p in dead]}')
                if all(p.exitcode == 0 for p in procs) and queue.empty():
                    raise RuntimeError('run_cohort: every rank process has ended (exit code 0) and none has sent the manifests')
                if timeout is not None and waited > timeout:
                    raise TimeoutError(f'run_cohort: no result after {timeout} s')
        for p in procs:
            p.join(timeout=120)
        bad = [p.exitcode for p in procs if p.exitcode != 0]
        if bad:
            raise RuntimeError(f'run_cohort: rank process(es) ended with exit code(s) {bad}')
        return result
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()                                   # the exact processes started above
