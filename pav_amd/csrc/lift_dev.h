// Point lifts through the alignment table on the device (lift_dev.hip): what AlignLift.lift_to_qry / lift_to_sub
// (pavlib/align/lift.py:51-272, 333-378) answer, for a batch of positions, from the operation tables pav_align_index leaves in
// HBM.  The scan driver (invscan.cpp) asks twice per flagged region and scan round; with the tables on the device a haplotype's
// lift-over index costs the tokenizer + two scans (0.2 ms) instead of a 109 MB copy to the host and the host's lookup tables.
#pragma once

#include <cstdint>

#include "common.h"

namespace pav {

struct LiftRowDev {                 // one alignment record
    uint32_t ref_id, tig_id;
    int64_t pos, end, qry_pos, qry_end;
    int64_t index;                  // INDEX column
    uint64_t op_a, op_b;            // its operations: [op_a, op_b) of the operation arrays
    uint64_t tig_len;               // length of its contig record
    int32_t rev;
    uint32_t bad;                   // 0: the record has no N (3) / P (6) operation (lift.py:463-471); else the code of the first one in the
};                                  // low four bits (row_bad_code) under its inverted ordinal; written by k_lift_row_bad
__host__ __device__ inline uint32_t row_bad_code(uint32_t bad) { return bad & 15u; }

struct LiftQuery { int32_t axis /* 0: reference position -> contig (lift_to_qry), 1: contig position -> reference (lift_to_sub) */,
                   seq, gap, pad; int64_t pos; };
enum { LIFT_NONE = 0, LIFT_OK = 1, LIFT_ERR_OP = 2, LIFT_ERR_NO_MATCH_QRY = 3, LIFT_ERR_NO_MATCH_SUB = 4 };
struct LiftAnswer {
    int32_t status, id;             // LIFT_*; record number on the other axis
    int64_t pos;
    int32_t rev, rev_none, n_idx;
    uint32_t row;                   // the record the position was lifted through (errors name it)
    int64_t idx[2];                 // INDEX of the record(s)
};

struct LiftTables {                 // device pointers
    const uint32_t *ops, *sub, *qry;
    LiftRowDev *rows; uint32_t n_rows;
    // per axis (0: by reference record, 1: by contig record): the records of sequence s are entries [seq_off[s], seq_off[s + 1]) of
    // seq_rows (sorted by their begin on that axis), with begin / end / running maximum of end beside them
    const uint32_t *seq_off[2], *seq_rows[2];
    const int64_t *begin[2], *end[2], *max_end[2];
    const uint32_t *tig_table;      // the contig axis again, records in table order (subject_gap breaks ties by it)
    uint32_t n_seq[2];
};

int lift_row_flags(pav_ctx *ctx, const LiftTables &T, uint64_t n_ops);                                             // queued on ctx->stream
int lift_points(pav_ctx *ctx, const LiftTables &T, const LiftQuery *d_q, LiftAnswer *d_a, uint32_t n);   // queued on ctx->stream

}  // namespace pav
