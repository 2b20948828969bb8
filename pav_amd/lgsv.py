"""
Large alignment-truncating variants: mirror of ``pavlib.lgsv.scan_for_events`` (pavlib/lgsv.py:31-642; rule
``call_lg_discover``, rules/call_lg.snakefile:40-104).

The control flow over (chromosome, contig) pairs with several alignment records is the reference's; what ran as Python
per-base loops and subprocesses runs on the GPU: the breakpoint homology of every INS / DEL (``pav_homology``, one batched
call for all events of the table: the event list does not depend on the homology values, see ``match_bp`` below) and the
inversion scans (``pav_amd.inv.scan_for_inv``).  Sequences are uploaded once per context instead of the reference's
per-call whole-chromosome ``SeqCache`` reloads (lgsv.py:645-695).
"""

import collections
import os
import sys

import numpy as np
import pandas as pd

from . import _lib, inv, seq
from .align import AlignLift
from .kmer import KmerUtil

MAX_QRY_DIST_PROP = 1       # lgsv.py:18
MAX_REF_DIST_PROP = 3       # lgsv.py:19
DIST_PROP_LEN_MAPQ = (20000, 40)   # lgsv.py:21
CALL_SOURCE = 'ALNTRUNC'
CALL_SOURCE_INV_DENSITY = 'ALNTRUNC-DEN'
CALL_SOURCE_INV_NO_DENSITY = 'ALNTRUNC-NODEN'

INSDEL_COLUMNS = ['#CHROM', 'POS', 'END', 'ID', 'SVTYPE', 'SVLEN', 'HAP', 'QRY_REGION', 'QRY_STRAND', 'CI', 'ALIGN_INDEX', 'LEFT_SHIFT',
                  'HOM_REF', 'HOM_TIG', 'CALL_SOURCE', 'FILTER', 'SEQ']
INV_COLUMNS = ['#CHROM', 'POS', 'END', 'ID', 'SVTYPE', 'SVLEN', 'HAP', 'QRY_REGION', 'QRY_STRAND', 'CI', 'RGN_REF_INNER', 'RGN_QRY_INNER',
               'RGN_REF_DISC', 'RGN_QRY_DISC', 'FLAG_ID', 'FLAG_TYPE', 'ALIGN_INDEX', 'CALL_SOURCE', 'FILTER', 'SEQ']


def match_bp(record, right_end):
    """``pavlib.align.match_bp`` (align.py:325-361) compares the operation *characters* of ``cigar_str_to_tuples`` with BAM
    integer codes, so no operation ever counts as a match: the first one breaks the loop and the result is 0 for every
    record with a non-empty CIGAR.  The left shift of alignment-truncating INS / DEL is therefore always 0 in PAV 2.4.6;
    kept as is."""
    return 0


def _overlap_error(row1, row2):
    return RuntimeError(
        'Contig ranges overlap for two alignment records (should not occur after alignment trimming): '
        f'Index {row1["INDEX"]} ({row1["QRY_ID"]}:{row1["QRY_POS"]}-{row1["QRY_END"]}) and '
        f'Index {row2["INDEX"]} ({row2["QRY_ID"]}:{row2["QRY_POS"]}-{row2["QRY_END"]})')


def _inv_series(inv_call, hap, is_rev, align_index, call_source, tig_fa_name):
    seq_str = seq.region_seq_fasta(inv_call.region_tig_outer, tig_fa_name, rev_compl=is_rev)
    return pd.Series([
        inv_call.region_ref_outer.chrom, inv_call.region_ref_outer.pos, inv_call.region_ref_outer.end, inv_call.id, 'INV', inv_call.svlen,
        hap, inv_call.region_tig_outer.to_base1_string(), '-' if is_rev else '+', 0,
        inv_call.region_ref_inner.to_base1_string(), inv_call.region_tig_inner.to_base1_string(),
        inv_call.region_ref_discovery.to_base1_string(), inv_call.region_tig_discovery.to_base1_string(),
        inv_call.region_flag.region_id(), 'ALNTRUNC', align_index, call_source, 'PASS', seq_str], index=INV_COLUMNS)


def scan_for_events(df, df_tig_fai, hap, ref_fa_name, tig_fa_name, k_size, n_tree=None, threads=1, log=sys.stdout,
                    density_out_dir=None, max_tig_dist_prop=None, max_ref_dist_prop=None, srs_tree=None, max_region_size=None,
                    version_id=True, ctx=None, device_id=0):
    """Same arguments and return value as ``pavlib.lgsv.scan_for_events``: ``(df_ins, df_del, df_inv)``."""
    if version_id:
        raise NotImplementedError('version_id=True needs svpoplib.variant.version_id (un-vendored submodule); '
                                  'rule call_lg_discover passes version_id=False (call_lg.snakefile:98)')
    max_tig_dist_prop = max_tig_dist_prop if max_tig_dist_prop is not None else MAX_QRY_DIST_PROP
    max_ref_dist_prop = max_ref_dist_prop if max_ref_dist_prop is not None else MAX_REF_DIST_PROP

    df = df.copy()
    df['QRY_LEN'] = df['END'] - df['POS']                                 # lgsv.py:70 (the column is reused for the reference span)

    own = ctx is None
    if own:
        ctx = _lib.Context(device_id)
    try:
        inv.ensure_sequences(ctx, ref_fa_name, tig_fa_name)
        ref_index, tig_index = inv._seq_index(ctx)
        align_lift = AlignLift(df, df_tig_fai)
        k_util = KmerUtil(k_size)

        events = []                    # INS / DEL: dict of row fields; homology filled in after the loop
        queries = []                   # four pav_hom_query per event
        inv_list = []
        inv_id_set = set()

        def hom(role, name, rev, pos, sv_role, sv_name, sv_rev, sv_pos, svlen, direction):
            names = ref_index if role == _lib.PAV_ROLE_REF else tig_index
            sv_names = ref_index if sv_role == _lib.PAV_ROLE_REF else tig_index
            queries.append((role, names[name], 1 if rev else 0, 0, pos, sv_role, sv_names[sv_name], 1 if sv_rev else 0, 0, sv_pos, svlen, direction))

        tig_map_count = collections.Counter(df[['#CHROM', 'QRY_ID']].apply(tuple, axis=1)) if df.shape[0] else {}
        tig_map_count = [(chrom, tig_id) for (chrom, tig_id), count in tig_map_count.items() if count > 1]

        for chrom, tig_id in tig_map_count:
            tig_index_list = list(df.loc[(df['#CHROM'] == chrom) & (df['QRY_ID'] == tig_id)].index)
            tig_index_list_len = len(tig_index_list)
            tig_len = int(df_tig_fai[tig_id])

            for subindex1 in range(len(tig_index_list) - 1):
                subindex2 = subindex1 + 1
                row1 = df.loc[tig_index_list[subindex1]]
                is_rev = row1['REV']

                while subindex2 < tig_index_list_len:
                    row2 = df.loc[tig_index_list[subindex2]]

                    if row2['REV'] == is_rev:
                        # INS / DEL / 2-record INV (lgsv.py:116-437)
                        if row1['QRY_POS'] < row2['QRY_POS']:
                            if row2['QRY_POS'] < row1['QRY_END']:
                                raise _overlap_error(row1, row2)
                            query_pos, query_end = row1['QRY_END'], row2['QRY_POS']
                        else:
                            if row1['QRY_POS'] < row2['QRY_END']:
                                raise _overlap_error(row1, row2)
                            query_pos, query_end = row2['QRY_END'], row1['QRY_POS']

                        dist_tig = query_end - query_pos
                        dist_ref = row2['POS'] - row1['END']
                        if dist_tig < 0:
                            raise RuntimeError(
                                f'Contig query positions are out of order (program bug): Contig distance is negative ({dist_tig}): '
                                f'Index {row1["INDEX"]} ({row1["QRY_ID"]}:{row1["QRY_POS"]}-{row1["QRY_END"]}) and '
                                f'Index {row2["INDEX"]} ({row2["QRY_ID"]}:{row2["QRY_POS"]}-{row2["QRY_END"]})')

                        min_aln_len = np.min([row1['QRY_LEN'], row2['QRY_LEN']])
                        min_mapq = np.min([row1['MAPQ'], row2['MAPQ']])
                        if min_aln_len < DIST_PROP_LEN_MAPQ[0] or min_mapq < DIST_PROP_LEN_MAPQ[1]:
                            if (np.abs(dist_tig) / min_aln_len > max_tig_dist_prop) or (np.abs(dist_ref) / min_aln_len > max_ref_dist_prop):
                                subindex2 += 1
                                continue

                        if dist_ref >= 50 and dist_tig < 50:
                            # DEL (lgsv.py:173-258)
                            svlen = dist_ref
                            pos_ref, end_ref = row1['END'], row2['POS']
                            pos_tig = query_pos
                            end_tig = pos_tig + 1
                            left_shift = np.min([match_bp(row1, True), 0])          # see match_bp: always 0, nothing moves
                            seq_str = seq.region_seq_fasta(seq.Region(chrom, pos_ref, end_ref), ref_fa_name)
                            sv_id = '{}-{}-DEL-{}'.format(chrom, pos_ref, svlen)
                            log.write('DEL: {}\n'.format(sv_id))
                            log.flush()
                            R, T = _lib.PAV_ROLE_REF, _lib.PAV_ROLE_TIG
                            sv = (R, chrom, False, int(pos_ref), int(svlen))
                            hom(R, chrom, False, int(pos_ref) - 1, *sv, 0)
                            hom(R, chrom, False, int(end_ref), *sv, 1)
                            hom(T, tig_id, is_rev, int(pos_tig) - 1, *sv, 0)
                            hom(T, tig_id, is_rev, int(pos_tig), *sv, 1)
                            events.append(('DEL', [chrom, pos_ref, end_ref, sv_id, 'DEL', svlen, hap, f'{tig_id}:{pos_tig + 1}-{end_tig}',
                                                   '-' if row1['REV'] else '+', dist_tig, '{},{}'.format(row1['INDEX'], row2['INDEX']),
                                                   left_shift, None, None, CALL_SOURCE, 'PASS', seq_str]))
                            break

                        elif dist_ref < 50 and dist_tig >= 50:
                            # INS (lgsv.py:261-348)
                            pos_ref = row1['END']
                            end_ref = pos_ref + 1
                            pos_tig, end_tig = query_pos, query_end
                            svlen = dist_tig
                            tig_region = seq.Region(tig_id, pos_tig, end_tig, is_rev=is_rev)
                            seq_str = seq.region_seq_fasta(tig_region, tig_fa_name, rev_compl=is_rev)
                            left_shift = np.min([match_bp(row1, True), 0])
                            sv_id = '{}-{}-INS-{}'.format(chrom, pos_ref, svlen)
                            log.write('INS: {}\n'.format(sv_id))
                            log.flush()
                            R, T = _lib.PAV_ROLE_REF, _lib.PAV_ROLE_TIG
                            # the SV sequence is contig[pos_tig:end_tig], reverse-complemented with the alignment
                            sv = (T, tig_id, bool(is_rev), (tig_len - int(end_tig)) if is_rev else int(pos_tig), int(svlen))
                            hom(R, chrom, False, int(pos_ref) - 1, *sv, 0)
                            hom(R, chrom, False, int(pos_ref), *sv, 1)
                            hom(T, tig_id, is_rev, int(pos_tig) - 1, *sv, 0)       # contig coordinates on the oriented contig, as lgsv.py:309-310
                            hom(T, tig_id, is_rev, int(end_tig), *sv, 1)
                            events.append(('INS', [chrom, pos_ref, end_ref, sv_id, 'INS', svlen, hap, tig_region.to_base1_string(),
                                                   '-' if is_rev else '+', dist_ref, '{},{}'.format(row1['INDEX'], row2['INDEX']),
                                                   left_shift, None, None, CALL_SOURCE, 'PASS', seq_str]))
                            break

                        elif dist_ref >= 50 and dist_tig >= 50:
                            # INV from two records (lgsv.py:351-434)
                            region_flag = seq.Region(chrom, row1['END'], row2['POS'], is_rev=row1['REV'])
                            inv_call = inv.scan_for_inv(region_flag, ref_fa_name, tig_fa_name, align_lift, k_util,
                                                        max_region_size=max_region_size, threads=threads, n_tree=n_tree, srs_tree=srs_tree,
                                                        log=log, min_exp_count=1, ctx=ctx)
                            if inv_call is not None and inv_call.id not in inv_id_set:
                                log.write('INV (2-tig): {}\n'.format(inv_call))
                                log.flush()
                                inv_list.append(_inv_series(inv_call, hap, is_rev, '{},{}'.format(row1['INDEX'], row2['INDEX']),
                                                            CALL_SOURCE_INV_DENSITY, tig_fa_name))
                                inv_id_set.add(inv_call.id)
                                if density_out_dir is not None:
                                    inv_call.df.to_csv(os.path.join(density_out_dir, 'density_{}_{}.tsv.gz'.format(inv_call.id, hap)),
                                                       sep='\t', index=False, compression='gzip')
                                break

                        subindex2 += 1

                    elif subindex2 + 1 < tig_index_list_len:
                        # INV from three records, middle one on the other strand (lgsv.py:440-558)
                        subindex3 = subindex2 + 1
                        row3 = df.loc[tig_index_list[subindex3]]
                        mid = (row2['QRY_POS'] + row2['QRY_END']) // 2
                        if (row3['REV'] == row1['REV']) and (
                            (not row1['REV'] and (row1['QRY_END'] < mid < row3['QRY_POS'])) or
                            (row1['REV'] and (row3['QRY_POS'] < mid < row1['QRY_END']))
                        ):
                            region_flag = seq.Region(chrom, row1['END'], row3['POS'], is_rev=row1['REV'])
                            inv_call = inv.scan_for_inv(region_flag, ref_fa_name, tig_fa_name, align_lift, k_util,
                                                        max_region_size=max_region_size, threads=threads, n_tree=n_tree, srs_tree=srs_tree,
                                                        log=log, min_exp_count=1, ctx=ctx)
                            if inv_call is None and subindex2 == subindex1 + 1 and subindex3 == subindex1 + 2:
                                region_ref = seq.Region(chrom, row2['POS'], row2['END'])
                                region_tig = seq.Region(row2['QRY_ID'], row2['QRY_POS'], row2['QRY_END'])
                                inv_call = inv.InvCall(region_ref, region_ref, region_tig, region_tig, region_ref, region_tig, region_ref, None)
                                call_source = CALL_SOURCE_INV_NO_DENSITY
                            else:
                                call_source = CALL_SOURCE_INV_DENSITY
                            if inv_call is not None and inv_call.id not in inv_id_set:
                                log.write('INV (3-tig): {}\n'.format(inv_call))
                                log.flush()
                                inv_list.append(_inv_series(inv_call, hap, is_rev,
                                                            '{},{},{}'.format(row1['INDEX'], row2['INDEX'], row3['INDEX']), call_source,
                                                            tig_fa_name))
                                inv_id_set.add(inv_call.id)
                                if density_out_dir is not None and inv_call.df is not None:
                                    inv_call.df.to_csv(os.path.join(density_out_dir, 'density_{}_{}.tsv.gz'.format(inv_call.id, hap)),
                                                       sep='\t', index=False, compression='gzip')
                                break

                    subindex2 += 1

        # breakpoint homology of all INS / DEL in one device call
        if queries:
            q = np.zeros(len(queries), dtype=_lib.HOM_QUERY_DTYPE)
            for i, rec in enumerate(queries):
                q[i] = rec
            h = ctx.homology(q)
        ins_list, del_list = [], []
        for e, (kind, fields) in enumerate(events):
            fields[12] = '{},{}'.format(int(h[4 * e]), int(h[4 * e + 1]))
            fields[13] = '{},{}'.format(int(h[4 * e + 2]), int(h[4 * e + 3]))
            (ins_list if kind == 'INS' else del_list).append(pd.Series(fields, index=INSDEL_COLUMNS))
    finally:
        if own:
            ctx.close()

    def table(rows, columns):
        if len(rows) > 0:
            out = pd.concat(rows, axis=1).T
            out.sort_values(['#CHROM', 'POS', 'END', 'ID'], inplace=True)
            return out
        return pd.DataFrame([], columns=columns)

    return table(ins_list, INSDEL_COLUMNS), table(del_list, INSDEL_COLUMNS), table(inv_list, INV_COLUMNS)
