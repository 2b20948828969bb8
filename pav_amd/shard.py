"""
Multi-GPU sharding for the hot path: one process per GPU, no data-path collective.

Units are independent at every level the reference already splits on:
  * haplotypes (README.md:83-86)                       -> one haplotype per rank when there are enough of them
  * ``CALL_BATCH`` alignment batches (cigarcall.py:21)  -> rules ``call_cigar`` jobs
  * flagged-region ``BATCH``es (call_inv.snakefile:96)  -> rules ``call_inv_batch`` jobs
Work is assigned longest-processing-time first on a cost estimate (CIGAR text length / region length); results are
gathered as Python objects on rank 0 (``torch.distributed.gather_object``; gloo on CPU, RCCL-backed nccl groups can use
the same call through a gloo side group) and merged exactly like ``call_cigar_merge`` / ``call_inv_batch_merge``.
"""

import numpy as np
import pandas as pd


def effective_cpus():
    """CPUs this process can really use: the affinity mask, capped by the cgroup CPU quota (containers report the host's core
    count in os.cpu_count() while cpu.max allows a fraction of it)."""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path in ('/sys/fs/cgroup/cpu.max',):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != 'max' and int(period) > 0:
                n = min(n, max(1, int(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
    try:                                                                   # cgroup v1
        q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        if q > 0 and per > 0:
            n = min(n, max(1, q // per))
    except (OSError, ValueError):
        pass
    return max(1, n)


def first_contact(backend, rank, world, local_rank, share_gpu=False, init=True):
    """Bring the process group of a multi-GPU run up and PROVE it before any work is queued: enough visible GPUs for the ranks
    of this node, a one-element ``all_reduce`` whose sum must be the world size (the first RCCL exchange of the job: a fabric /
    IPC problem shows here, with a message, not as a hang inside the timed region), and - one GPU per rank - every rank on a
    device of its own (PCI bus ids gathered and compared).  Returns ``{'device_name', 'pci_bus_id'}`` of this rank's GPU (empty
    strings without one).  ``init=False``: the group exists already, only the checks run."""
    import torch
    import torch.distributed as dist
    info = {'device_name': '', 'pci_bus_id': ''}
    on_gpu = backend == 'nccl'
    if on_gpu:
        n_dev = torch.cuda.device_count()
        need = 1 if share_gpu else local_rank + 1
        if n_dev < need:
            raise RuntimeError(f'rank {rank}: local rank {local_rank} needs GPU {need - 1} but only {n_dev} HIP device(s) are visible '
                               f'(HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES = {__import__("os").environ.get("HIP_VISIBLE_DEVICES")!r} / '
                               f'{__import__("os").environ.get("ROCR_VISIBLE_DEVICES")!r}); one process per GPU, no collective moves data')
        dev = 0 if share_gpu else local_rank
        torch.cuda.set_device(dev)
        props = torch.cuda.get_device_properties(dev)
        info['device_name'] = props.name
        bus = [getattr(props, a, None) for a in ('pci_domain_id', 'pci_bus_id', 'pci_device_id')]
        if all(b is not None for b in bus):
            info['pci_bus_id'] = '{:04x}:{:02x}:{:02x}.0'.format(*bus)
    if world <= 1:
        return info
    if init and not dist.is_initialized():
        if on_gpu:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', 0 if share_gpu else local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    one = torch.ones(1, dtype=torch.float32, device='cuda' if on_gpu else 'cpu')
    try:
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        got = float(one.item())
    except Exception as ex:                                               # noqa: BLE001 - re-raised with what the user needs to know
        raise RuntimeError(f'rank {rank}/{world}: the first all_reduce over the {backend} group failed ({ex}); on ROCm check that '
                           f'HSA_ENABLE_IPC_MODE_LEGACY=0 is exported and that MASTER_ADDR is 127.0.0.1 on a single node') from ex
    if got != float(world):
        raise RuntimeError(f'rank {rank}/{world}: the first all_reduce returned {got}, expected {world}')
    seen = [None] * world
    dist.all_gather_object(seen, info['pci_bus_id'])
    if on_gpu and not share_gpu and all(seen) and len(set(seen)) != world:
        raise RuntimeError(f'{world} ranks but only {len(set(seen))} distinct GPU(s): {seen} - every rank must drive a device of its own')
    return info


def assign_lpt(costs, n_ranks):
    """Greedy longest-processing-time assignment.  Returns ``n_ranks`` lists of item indices (deterministic)."""
    costs = np.asarray(costs, dtype=np.float64)
    order = np.argsort(-costs, kind='stable')
    load = np.zeros(n_ranks, dtype=np.float64)
    out = [[] for _ in range(n_ranks)]
    for i in order.tolist():
        r = int(np.argmin(load))
        out[r].append(i)
        load[r] += costs[i]
    for lst in out:
        lst.sort()
    return out


def shard_alignments(df_align, rank, world, by='row'):
    """Rows of the alignment table this rank walks.  ``by='batch'`` keeps the reference's CALL_BATCH groups whole."""
    if world <= 1:
        return df_align
    if by == 'batch':
        batches = sorted(df_align['CALL_BATCH'].unique())
        cost = [int(df_align.loc[df_align['CALL_BATCH'] == b, 'CIGAR'].str.len().sum()) for b in batches]
        mine = {batches[i] for i in assign_lpt(cost, world)[rank]}
        return df_align.loc[df_align['CALL_BATCH'].isin(mine)]
    cost = df_align['CIGAR'].str.len().to_numpy()
    return df_align.iloc[assign_lpt(cost, world)[rank]]


def shard_regions(df_flag, rank, world):
    """Flagged regions this rank scans (cost = region length + the initial 4 kb expansion)."""
    if world <= 1:
        return df_flag
    cost = (df_flag['END'] - df_flag['POS']).to_numpy() + 4000
    return df_flag.iloc[assign_lpt(cost, world)[rank]]


def shard_haplotypes(n_haplotypes, rank, world, costs=None):
    """Haplotype indices processed by this rank (config 4: 16 haplotypes over 8 GPUs, two waves)."""
    costs = np.ones(n_haplotypes) if costs is None else costs
    return assign_lpt(costs, world)[rank]


def gather_frames(df_local, dst=0, group=None):
    """Gather per-rank DataFrames on ``dst`` and concatenate (None on the other ranks).  No tensor collective: the
    tables are host objects, exactly like the per-batch files the reference exchanges through the file system."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return df_local
    rank = dist.get_rank(group)
    buf = [None] * dist.get_world_size(group) if rank == dst else None
    dist.gather_object(df_local, buf, dst=dst, group=group)
    if rank != dst:
        return None
    return pd.concat(buf, axis=0)


def merge_cigar_tables(df_snv, df_insdel):
    """Order of rule call_cigar_merge (rules/call.snakefile:763-786)."""
    return (df_snv.reset_index(drop=True).sort_values(['#CHROM', 'POS']),
            df_insdel.reset_index(drop=True).sort_values(['#CHROM', 'POS', 'END', 'ID']))
