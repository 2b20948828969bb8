"""N > 1 plumbing on CPU: two gloo ranks shard the alignment table / flagged regions, produce their tables with the
oracle (standing in for the GPU walk, which needs no collective), gather on rank 0 and merge; the result must equal
the unsharded tables.  Also checks the LPT assignment."""
import io
import os
import socket

import numpy as np
import pandas as pd
import pytest

import util
from pav_amd import shard


def test_assign_lpt_is_balanced_and_complete():
    rng = np.random.default_rng(0)
    costs = rng.pareto(1.2, 200) + 1
    parts = shard.assign_lpt(costs, 8)
    assert sorted(i for p in parts for i in p) == list(range(200))
    loads = [costs[p].sum() for p in parts]
    assert max(loads) <= costs.sum() / 8 + costs.max()
    assert shard.assign_lpt(costs, 8) == parts                      # deterministic


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        d, df_align, df_trim = util.golden_case(case)
        mine = shard.shard_alignments(df_align, rank, world)
        df_snv, df_insdel = util.oracle_frames(d, mine, df_trim)
        g_snv = shard.gather_frames(df_snv)
        g_ins = shard.gather_frames(df_insdel)
        flag = pd.DataFrame({'#CHROM': ['c'] * 9, 'POS': np.arange(9) * 1000, 'END': np.arange(9) * 1000 + [10, 5000, 20, 30, 4000, 1, 2, 3, 4]})
        regions = shard.shard_regions(flag, rank, world)
        all_regions = shard.gather_frames(regions)
        dist.barrier()
        if rank == 0:
            s, i = shard.merge_cigar_tables(g_snv, g_ins)
            q.put((util.frame_text(s), util.frame_text(i), sorted(all_regions['POS'].tolist()), len(mine)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_shard_gather_merge(built):
    import torch.multiprocessing as mp
    case, world = 'cigar_synth', 2
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    snv_text, ins_text, region_pos, n_mine = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d, df_align, df_trim = util.golden_case(case)
    assert 0 < n_mine < df_align.shape[0]
    df_snv, df_insdel = util.oracle_frames(d, df_align, df_trim)
    s, i = shard.merge_cigar_tables(df_snv, df_insdel)
    assert snv_text == util.frame_text(s)
    assert ins_text == util.frame_text(i)
    assert region_pos == [k * 1000 for k in range(9)]
