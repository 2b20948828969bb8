// Device side of the lift-over (lift_dev.h): one lane per point query.  The control is the host driver's (invscan.cpp, itself the
// restatement of pavlib/align/lift.py), statement by statement; every search is a bisection over arrays in HBM - a few thousand
// queries per scan round, ~20 dependent loads each, all of them in flight at once.
#include "lift_dev.h"

namespace pav {

namespace {

// records of sequence `seq` on `axis` whose [begin, end) contains pos: the count, and the last one met (the only one when count == 1)
__device__ int containing(const LiftTables &T, int axis, int32_t seq, int64_t pos, uint32_t &hit) {
    if (seq < 0 || (uint32_t)seq >= T.n_seq[axis]) return 0;
    const uint32_t a = T.seq_off[axis][seq], b = T.seq_off[axis][seq + 1];
    uint32_t lo = a, hi = b;                                            // first entry whose begin is above pos
    while (lo < hi) { const uint32_t mid = lo + (hi - lo) / 2; if (T.begin[axis][mid] <= pos) lo = mid + 1; else hi = mid; }
    int count = 0;
    uint32_t i = lo;
    while (i > a && T.max_end[axis][i - 1] > pos) {
        --i;
        if (T.end[axis][i] > pos) { ++count; hit = T.seq_rows[axis][i]; }
    }
    return count;
}

// operation of `row` that holds position pos on `axis`, or -1 (invscan.cpp Driver::op_at)
__device__ int64_t op_at(const LiftTables &T, const LiftRowDev &r, int axis, int64_t pos) {
    if (pos < 0) return -1;
    const uint32_t *beg = axis == 0 ? T.sub : T.qry;
    const uint32_t v = (uint32_t)(pos < 0xFFFFFFFFll ? pos : 0xFFFFFFFFll);
    uint64_t lo = r.op_a, hi = r.op_b;                                  // first operation that begins above v
    while (lo < hi) { const uint64_t mid = lo + (hi - lo) / 2; if (beg[mid] <= v) lo = mid + 1; else hi = mid; }
    if (lo == r.op_a) return -1;
    const uint64_t k = lo - 1;
    const uint32_t code = T.ops[k] & 15u, len = T.ops[k] >> 4;
    const bool match = code == 7 || code == 8 || code == 0;
    if (!(match || code == (axis == 0 ? 2u : 1u))) return -1;
    return pos < (int64_t)beg[k] + (int64_t)len ? (int64_t)k : -1;
}

// (begin, end, d0, d1) of operation k on `axis` (pavlib/align/lift.py:437-461)
__device__ void op_interval(const LiftTables &T, uint64_t k, int axis, int64_t &begin, int64_t &end, int64_t &d0, int64_t &d1) {
    const uint32_t code = T.ops[k] & 15u; const int64_t len = T.ops[k] >> 4;
    const bool match = code == 7 || code == 8 || code == 0;
    begin = axis == 0 ? T.sub[k] : T.qry[k];
    end = begin + len;
    d0 = axis == 0 ? T.qry[k] : T.sub[k];
    d1 = match ? d0 + len : d0 + 1;
}

// AlignLift._get_subject_gap (lift.py:333-378)
__device__ void subject_gap(const LiftTables &T, int32_t tig, int64_t pos, LiftAnswer &out) {
    if (tig < 0 || (uint32_t)tig >= T.n_seq[1]) return;
    const uint32_t a = T.seq_off[1][tig], b = T.seq_off[1][tig + 1];
    int64_t best_l = -1, best_r = -1, lv = 0, rv = 0;
    for (uint32_t i = a; i < b; ++i) {                                  // table order: the largest QRY_END below pos (ties: the last), the
        const uint32_t r = T.tig_table[i];                              // smallest QRY_POS above it (ties: the first)
        const LiftRowDev &x = T.rows[r];
        if (x.qry_end < pos && (best_l < 0 || x.qry_end >= lv)) { best_l = r; lv = x.qry_end; }
        if (x.qry_pos > pos && (best_r < 0 || x.qry_pos < rv)) { best_r = r; rv = x.qry_pos; }
    }
    if (best_l < 0 || best_r < 0) return;
    const LiftRowDev &L = T.rows[best_l], &R = T.rows[best_r];
    if (L.ref_id != R.ref_id) return;
    out.status = LIFT_OK; out.id = (int32_t)L.ref_id;
    out.pos = (int64_t)((double)(L.qry_end + R.qry_pos) / 2.0);        // int((a + b) / 2)
    out.rev = L.rev; out.rev_none = L.rev != R.rev;
    out.idx[0] = L.index; out.idx[1] = R.index; out.n_idx = 2;
}

__global__ __launch_bounds__(64) void k_lift_points(LiftTables T, const LiftQuery *__restrict__ q, LiftAnswer *__restrict__ out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const LiftQuery Q = q[i];
    LiftAnswer A{};
    A.status = LIFT_NONE; A.id = -1;
    if (Q.axis < 0 || Q.axis > 1) { out[i] = A; return; }              // an empty slot of a round's query block (k_round_decide)
    uint32_t row = 0;
    const int hits = containing(T, Q.axis, Q.seq, Q.pos, row);
    if (Q.axis == 0) {                                                  // AlignLift.lift_to_qry (lift.py:187-272)
        if (hits == 1) {
            const LiftRowDev r = T.rows[row];
            A.row = row;
            if (r.bad) { A.status = LIFT_ERR_OP; A.id = (int32_t)row_bad_code(r.bad); }
            else {
                const int64_t k = op_at(T, r, 0, Q.pos);
                if (k < 0) A.status = LIFT_ERR_NO_MATCH_QRY;
                else {
                    int64_t b, e, d0, d1;
                    op_interval(T, (uint64_t)k, 0, b, e, d0, d1);
                    int64_t p = d1 - d0 > 1 ? d0 + (Q.pos - b) : d1;
                    if (r.rev) p = (int64_t)r.tig_len - p;
                    A.status = LIFT_OK; A.id = (int32_t)r.tig_id; A.pos = p; A.rev = r.rev; A.idx[0] = r.index; A.n_idx = 1;
                }
            }
        }
    } else {                                                            // AlignLift.lift_to_sub (lift.py:51-185)
        if (hits == 0 && Q.gap) subject_gap(T, Q.seq, Q.pos, A);
        else if (hits == 1) {
            const LiftRowDev r = T.rows[row];
            A.row = row;
            if (r.bad) { A.status = LIFT_ERR_OP; A.id = (int32_t)row_bad_code(r.bad); }
            else {
                const int64_t pos = r.rev ? (int64_t)r.tig_len - Q.pos : Q.pos;
                int64_t k = op_at(T, r, 1, pos);
                int64_t b = 0, e = 0, d0 = 0, d1 = 0;
                bool bad = false;
                if (k < 0) {
                    k = op_at(T, r, 1, pos - 1);
                    if (k >= 0) op_interval(T, (uint64_t)k, 1, b, e, d0, d1);
                    if (k < 0 || e != pos) bad = true;
                }
                if (bad) A.status = LIFT_ERR_NO_MATCH_SUB;
                else {
                    op_interval(T, (uint64_t)k, 1, b, e, d0, d1);
                    A.status = LIFT_OK; A.id = (int32_t)r.ref_id; A.pos = d1 - d0 > 1 ? d0 + (pos - b) : d1; A.rev = r.rev;
                    A.idx[0] = r.index; A.n_idx = 1;
                }
            }
        }
    }
    out[i] = A;
}

// first N / P operation of every record (lift.py:463-471): a flat grid over ALL operations of the table (a 150 Mb contig is one
// record of half a million operations: a wave per record walked it 64 at a time, 2.1 ms; this is one read of the array).  An N / P
// operation is an exception, so the lane that meets one looks its record up by bisection and keeps the EARLIEST one with an
// atomicMax of ((2^28 - 1 - ordinal within the record) << 4 | code); readers take `bad & 15` (row_bad_code).
constexpr uint32_t BAD_OPS_PER_LANE = 8;
__global__ __launch_bounds__(256) void k_lift_row_bad(LiftTables T, uint64_t n_ops) {
    const uint64_t k0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * BAD_OPS_PER_LANE;
    if (k0 >= n_ops) return;
    uint32_t code[BAD_OPS_PER_LANE];
    if (k0 + BAD_OPS_PER_LANE <= n_ops && (reinterpret_cast<uintptr_t>(T.ops + k0) & 15u) == 0) {
        const uint4 a = *reinterpret_cast<const uint4 *>(T.ops + k0), b = *reinterpret_cast<const uint4 *>(T.ops + k0 + 4);
        code[0] = a.x; code[1] = a.y; code[2] = a.z; code[3] = a.w; code[4] = b.x; code[5] = b.y; code[6] = b.z; code[7] = b.w;
    } else {
        for (uint32_t j = 0; j < BAD_OPS_PER_LANE; ++j) code[j] = k0 + j < n_ops ? T.ops[k0 + j] : 0u;
    }
    for (uint32_t j = 0; j < BAD_OPS_PER_LANE; ++j) {
        const uint32_t c = code[j] & 15u;
        if (c != 3u && c != 6u) continue;
        const uint64_t k = k0 + j;
        uint32_t lo = 0, hi = T.n_rows;                                // last record whose operations begin at or before k
        while (hi - lo > 1) { const uint32_t mid = lo + (hi - lo) / 2; if (T.rows[mid].op_a <= k) lo = mid; else hi = mid; }
        while (lo < T.n_rows && T.rows[lo].op_b <= k) ++lo;            // records without operations share their op_a with the next
        if (lo >= T.n_rows || k < T.rows[lo].op_a) continue;
        const uint64_t rel = k - T.rows[lo].op_a;
        const uint32_t ord = rel < 0x0FFFFFFFull ? (uint32_t)rel : 0x0FFFFFFFu;
        atomicMax(&T.rows[lo].bad, (0x0FFFFFFFu - ord) << 4 | c);
    }
}

}  // namespace

int lift_row_flags(pav_ctx *ctx, const LiftTables &T, uint64_t n_ops) {
    if (!T.n_rows) return PAV_OK;
    if (!n_ops) return PAV_OK;
    const uint64_t per_block = 256ull * BAD_OPS_PER_LANE;
    PAV_LAUNCH(ctx, "k_lift_row_bad", k_lift_row_bad, (uint32_t)((n_ops + per_block - 1) / per_block), 256, 0, T, n_ops);
    return PAV_OK;
}

int lift_points(pav_ctx *ctx, const LiftTables &T, const LiftQuery *d_q, LiftAnswer *d_a, uint32_t n) {
    if (!n) return PAV_OK;
    PAV_LAUNCH(ctx, "k_lift_points", k_lift_points, (n + 63) / 64, 64, 0, T, d_q, d_a, n);
    return PAV_OK;
}

}  // namespace pav
