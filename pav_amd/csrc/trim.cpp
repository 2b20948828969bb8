// Alignment trimming: pavlib.align.trim_alignments / trim_alignment_record /
// find_cut_sites / trace_cigar_to_zero (pavlib/align/trim.py:11-917) behind rules align_trim_tig / align_trim_tigref
// (rules/align.snakefile:54-97).
//
// The reference re-tokenises both CIGAR strings of every overlapping pair (O(pairs x CIGAR length) in Python) and walks
// the table with df.loc.  Here every CIGAR is tokenised once, on the device (the tokenizer of the call path), and a record's
// current CIGAR is a window into that operation array plus the clipping operations trimming adds at either end, so a pair
// costs only the operations inside the overlap.  The pair loop itself is sequential by definition (a trimmed record is
// the input of the next pair) and runs on the host inside the library.
#include "common.h"
#include "trim_dev.h"

#include <algorithm>
#include <atomic>
#include <string>
#include <thread>

namespace pav {
namespace {

constexpr uint8_t OP_M = 0, OP_I = 1, OP_D = 2, OP_S = 4, OP_H = 5, OP_EQ = 7, OP_X = 8;
const char OP_CHARS[] = "MIDNSHP=X";

struct Op { uint64_t len; uint8_t code; };

// Current CIGAR of one record: pre + ops[a, b) (first / last length overridden) + post.
struct Cigar {
    std::vector<Op> pre, post;
    uint64_t a = 0, b = 0;
    uint64_t len_first = 0, len_last = 0;
    bool modified = false;
    size_t size() const { return pre.size() + (size_t)(b - a) + post.size(); }
};

struct Row {
    pav_trim_row f{};
    Cigar c;
};

struct TrimState {
    std::vector<uint32_t> ops;              // tokenised operations of the loaded table, len << 4 | BAM code
    std::vector<uint64_t> op_off;
    std::vector<Row> rows;
    pav_trim_err err{};
    std::string text;                       // CIGAR strings of the last pav_trim_fetch
    std::vector<uint64_t> text_off;
    // device side of the passes (trim_dev.hip): the operations stay resident, rows / order / scratch go up per pass
    DevBuf d_ops, d_rows, d_order, d_group, d_scratch, d_meta, d_err;
    uint64_t n_dev_passes = 0;
    // a pass that failed leaves the rows undefined: the sequential host loops have trimmed every pair in front of the failing
    // one, the device pass copies nothing back - so after PAV_E_TRIM from pav_trim_pass the table answers PAV_E_STATE on both
    // paths until the next pav_trim_load (the reference raises there as well and never returns a table)
    bool undefined = false;
    ~TrimState() { for (DevBuf *b : {&d_ops, &d_rows, &d_order, &d_group, &d_scratch, &d_meta, &d_err}) b->release(); }
};

TrimState *tstate(pav_ctx *ctx) {
    if (!ctx->trim) ctx->trim = new TrimState();
    return static_cast<TrimState *>(ctx->trim);
}

inline Op cigar_get(const TrimState &S, const Cigar &c, size_t i) {
    if (i < c.pre.size()) return c.pre[i];
    i -= c.pre.size();
    const size_t w = (size_t)(c.b - c.a);
    if (i < w) {
        const uint32_t o = S.ops[c.a + i];
        uint64_t len = o >> 4;
        if (i == 0) len = c.len_first;
        if (i + 1 == w) len = (w == 1) ? c.len_first : c.len_last;
        return Op{len, (uint8_t)(o & 15)};
    }
    return c.post[i - w];
}

// trace_cigar_to_zero record (trim.py:779-800)
struct Trace {
    size_t index;            // TC_INDEX: index in the oriented operation list
    uint64_t op_len;         // TC_OP_LEN
    uint8_t op_code;         // TC_OP_CODE
    int64_t diff_cum, diff;  // TC_DIFF_CUM, TC_DIFF
    int64_t event_cum, event;   // TC_EVENT_CUM, TC_EVENT
    int64_t sub_bp, qry_bp;  // TC_SUB_BP, TC_QRY_BP
    int64_t clip_s, clip_h;  // TC_CLIPS_BP, TC_CLIPH_BP
};

struct TrimFail { int kind = 0; uint32_t op_index = 0; uint64_t op_len = 0; uint32_t op_char = 0; int64_t diff_bp = 0; int side = 0; };

// trim.py:779-917.  `rev`: the list is traversed from its end.
bool trace_cigar_to_zero(const TrimState &S, const Cigar &c, bool rev, int64_t diff_bp, bool diff_query, std::vector<Trace> &out,
                         TrimFail &fail) {
    out.clear();
    const size_t n = c.size();
    size_t index = 0;
    int64_t diff_cumulative = 0, event_cumulative = 0, sub_bp_sum = 0, qry_bp_sum = 0, clip_s_sum = 0, clip_h_sum = 0;
    bool last_no_match = false;
    while (index < n && (diff_cumulative <= diff_bp || last_no_match || out.empty())) {
        const Op o = cigar_get(S, c, rev ? n - 1 - index : index);
        const int64_t len = (int64_t)o.len;
        int64_t event_count = 0, sub_bp = 0, qry_bp = 0;
        switch (o.code) {
            case OP_EQ: sub_bp = len; qry_bp = len; last_no_match = false; break;
            case OP_X: event_count = len; sub_bp = len; qry_bp = len; last_no_match = true; break;
            case OP_I: event_count = 1; qry_bp = len; last_no_match = true; break;
            case OP_D: event_count = 1; sub_bp = len; last_no_match = true; break;
            case OP_S: clip_s_sum += len; last_no_match = true; break;
            case OP_H: clip_h_sum += len; last_no_match = true; break;
            default:
                fail.kind = PAV_TRIM_ERR_ILLEGAL_OP; fail.op_index = (uint32_t)index; fail.op_len = o.len; fail.op_char = (uint32_t)OP_CHARS[o.code];
                return false;
        }
        const int64_t diff_change = diff_query ? qry_bp : sub_bp;
        if (o.code == OP_EQ || o.code == OP_X)
            out.push_back(Trace{index, o.len, o.code, diff_cumulative, diff_change, event_cumulative, event_count, sub_bp_sum, qry_bp_sum,
                                clip_s_sum, clip_h_sum});
        diff_cumulative += diff_change;
        event_cumulative += event_count;
        sub_bp_sum += sub_bp;
        qry_bp_sum += qry_bp;
        ++index;
    }
    return true;
}

// trim.py:602-776.  Returns false when no cut site exists.
bool find_cut_sites(const std::vector<Trace> &tl, const std::vector<Trace> &tr, int64_t diff_bp, size_t &cut_l, size_t &cut_r) {
    const int64_t len_r = (int64_t)tr.size();
    int64_t tc_idx_r = 0;
    bool have = false, have_diff = false;
    int64_t max_event = 0, max_diff_optimal = 0;
    for (int64_t tc_idx_l = (int64_t)tl.size() - 1; tc_idx_l >= 0; --tc_idx_l) {
        bool have_part = false, have_diff_part = false;
        size_t part_l = 0, part_r = 0;
        int64_t max_event_part = 0, max_diff_optimal_part = 0;
        const Trace &L = tl[(size_t)tc_idx_l];
        const int64_t min_bp_l = L.diff_cum, max_bp_l = L.diff_cum + L.diff - 1;
        while (tc_idx_r + 1 < len_r && max_bp_l + tr[(size_t)tc_idx_r].diff_cum + tr[(size_t)tc_idx_r].diff - 1 < diff_bp) ++tc_idx_r;
        const int64_t tc_idx_r_start = tc_idx_r;
        while (tc_idx_r < len_r && (min_bp_l + tr[(size_t)tc_idx_r].diff_cum <= diff_bp || tc_idx_r == tc_idx_r_start)) {
            const Trace &R = tr[(size_t)tc_idx_r];
            const int64_t max_bp = max_bp_l + R.diff_cum + R.diff - 1;
            const int64_t diff_min = diff_bp - max_bp;
            int64_t event_count = L.event_cum + R.event_cum, diff_optimal;
            if (diff_min <= 0) {
                event_count += std::min<int64_t>(diff_bp - diff_min, L.event + R.event - (L.event > 0 ? 1 : 0) - (R.event > 0 ? 1 : 0));
                diff_optimal = 0;
            } else {
                diff_optimal = diff_min;
            }
            if (event_count > max_event_part || (event_count == max_event_part && (!have_diff_part || diff_optimal < max_diff_optimal_part))) {
                have_part = true; part_l = (size_t)tc_idx_l; part_r = (size_t)tc_idx_r;
                max_event_part = event_count; max_diff_optimal_part = diff_optimal; have_diff_part = true;
            }
            ++tc_idx_r;
        }
        // `max_diff_optimal_part` is never None here when the right trace is non-empty: the inner loop runs at least once
        if (max_event_part > max_event || (max_event_part == max_event && (!have_diff || (have_diff_part && max_diff_optimal_part < max_diff_optimal)))) {
            have = have_part; cut_l = part_l; cut_r = part_r;
            max_event = max_event_part; max_diff_optimal = max_diff_optimal_part; have_diff = have_diff_part;
        }
        tc_idx_r = tc_idx_r_start;
    }
    return have;
}

// Apply one cut to a record (trim.py:497-593): drop the operations before `cut` in the oriented list, shorten the surviving
// one by `trim`, put the accumulated clipping in front, fix coordinates and TRIM_* counters.
void apply_cut(const TrimState &S, Row &r, bool rev, const Trace &cut, int64_t trim) {
    const int64_t cut_sub = cut.sub_bp + trim, cut_qry = cut.qry_bp + trim;
    if (rev) {
        r.f.end -= cut_sub;
        if (r.f.rev) r.f.qry_pos += cut_qry; else r.f.qry_end -= cut_qry;
        r.f.trim_ref_r += cut_sub;
        r.f.trim_qry_r += cut_qry;
    } else {
        r.f.pos += cut_sub;
        if (r.f.rev) r.f.qry_end -= cut_qry; else r.f.qry_pos += cut_qry;
        r.f.trim_ref_l += cut_sub;
        r.f.trim_qry_l += cut_qry;
    }
    std::vector<Op> clip;                                         // in the oriented direction: H, then S
    if (cut.clip_h > 0) clip.push_back(Op{(uint64_t)cut.clip_h, OP_H});
    const int64_t clip_s = cut.clip_s + cut.qry_bp + trim;
    if (clip_s > 0) clip.push_back(Op{(uint64_t)clip_s, OP_S});
    Cigar &c = r.c;
    const size_t n = c.size();
    // the surviving operation is an '=' / 'X', hence inside the window
    const size_t fwd = rev ? n - 1 - cut.index : cut.index;       // index in stored orientation
    const size_t w_i = fwd - c.pre.size();                        // index within the window
    const uint64_t new_len = cut.op_len - (uint64_t)trim;
    (void)S;
    if (!rev) {
        const bool single = (c.b - c.a) - w_i == 1;
        c.a += w_i;
        c.len_first = new_len;
        if (single) c.len_last = new_len;
        c.pre = clip;
    } else {
        const bool single = w_i == 0;
        c.b = c.a + w_i + 1;
        c.len_last = new_len;
        if (single) c.len_first = new_len;
        std::reverse(clip.begin(), clip.end());
        c.post = clip;
    }
    c.modified = true;
}

// trim_alignment_record (trim.py:357-599).  `l`, `r` are modified copies on success.
bool trim_record(const TrimState &S, Row &l, Row &r, bool query, bool rev_l, bool rev_r, TrimFail &fail) {
    int64_t diff_bp;
    if (query) {
        if (l.f.qry_pos < r.f.qry_pos) diff_bp = l.f.qry_end - r.f.qry_pos;
        else diff_bp = r.f.qry_end - l.f.qry_pos;
        if (diff_bp <= 0) { fail.kind = PAV_TRIM_ERR_NEGATIVE; fail.diff_bp = diff_bp; return false; }
    } else {
        if (l.f.pos > r.f.pos) { fail.kind = PAV_TRIM_ERR_ORDER; return false; }
        diff_bp = l.f.end - r.f.pos;
        if (diff_bp <= 0) { fail.kind = PAV_TRIM_ERR_NEGATIVE; fail.diff_bp = diff_bp; return false; }
    }
    std::vector<Trace> trace_l, trace_r;
    if (!trace_cigar_to_zero(S, l.c, rev_l, diff_bp, query, trace_l, fail)) { fail.side = 0; return false; }
    if (!trace_cigar_to_zero(S, r.c, rev_r, diff_bp, query, trace_r, fail)) { fail.side = 1; return false; }
    size_t ci_l = 0, ci_r = 0;
    if (!find_cut_sites(trace_l, trace_r, diff_bp, ci_l, ci_r)) { fail.kind = PAV_TRIM_ERR_NO_CUT; return false; }
    const Trace &cut_l = trace_l[ci_l], &cut_r = trace_r[ci_r];
    // mid-record cuts: left-align, mismatches first (trim.py:475-494)
    int64_t residual = diff_bp - (cut_l.diff_cum + cut_r.diff_cum), trim_l = 0, trim_r = 0;
    if (residual > 0 && cut_r.op_code == OP_X) { trim_r += std::min<int64_t>(residual, (int64_t)cut_r.op_len - 1); residual -= trim_r; }
    if (residual > 0 && cut_l.op_code == OP_X) { trim_l += std::min<int64_t>(residual, (int64_t)cut_l.op_len - 1); residual -= trim_l; }
    if (residual > 0 && cut_l.op_code == OP_EQ) { trim_l += std::min<int64_t>(residual, (int64_t)cut_l.op_len - 1); residual -= trim_l; }
    if (residual > 0 && cut_r.op_code == OP_EQ) { trim_r += std::min<int64_t>(residual, (int64_t)cut_r.op_len - 1); residual -= trim_r; }
    apply_cut(S, l, rev_l, cut_l, trim_l);
    apply_cut(S, r, rev_r, cut_r, trim_r);
    return true;
}

int record_fail(pav_ctx *ctx, TrimState *S, const TrimFail &f, uint32_t row_l, uint32_t row_r) {
    S->err = pav_trim_err{f.kind, row_l, row_r, f.op_index, f.op_char, f.side, f.diff_bp, f.op_len};
    return fail(ctx, PAV_E_TRIM, "alignment trimming failed (kind %d, records %u / %u)", f.kind, row_l, row_r);
}

inline int64_t qlen(const Row &r) { return r.f.qry_end - r.f.qry_pos; }

// Contig-space pass (trim.py:61-256) over the rows in `order` (sorted by QRY_ID, QRY_LEN descending).
int pass_query(pav_ctx *ctx, TrimState *S, const std::vector<uint32_t> &order, int64_t min_len) {
    const size_t n = order.size();
    std::vector<Row> &rows = S->rows;
    for (size_t il = 0; il < n; ++il) {
        for (size_t ir = il + 1; ir < n && rows[order[il]].f.qry_id == rows[order[ir]].f.qry_id; ++ir) {
            uint32_t index_l, index_r;
            if (rows[order[il]].f.qry_pos <= rows[order[ir]].f.qry_pos) { index_l = order[il]; index_r = order[ir]; }
            else { index_l = order[ir]; index_r = order[il]; }
            if (rows[index_l].f.index < 0 || rows[index_r].f.index < 0) continue;
            if (rows[index_r].f.qry_pos >= rows[index_l].f.qry_end) continue;
            if (rows[index_r].f.qry_end <= rows[index_l].f.qry_end) { rows[index_r].f.index = -1; continue; }
            bool rev_l = !rows[index_l].f.rev, rev_r = rows[index_r].f.rev != 0;
            bool ref_overlap = false;
            const pav_trim_row &fl = rows[index_l].f, &fr = rows[index_r].f;
            if (!(rev_l == rev_r || fl.chrom != fr.chrom)) {
                if (fl.pos < fr.pos) ref_overlap = fr.pos < fl.end;
                else if (fr.pos < fl.pos) ref_overlap = fl.pos < fr.end;
            }
            Row record_l, record_r;
            TrimFail tf;
            if (ref_overlap) {
                // try both trim orders and keep the one that left-aligns best (trim.py:128-197)
                Row la = rows[index_l], ra = rows[index_r];
                if (!trim_record(*S, la, ra, true, rev_l, rev_r, tf)) return record_fail(ctx, S, tf, index_l, index_r);
                Row lb = rows[index_r], rb = rows[index_l];
                if (!trim_record(*S, lb, rb, true, rev_r, rev_l, tf)) return record_fail(ctx, S, tf, index_r, index_l);
                int keep = 0;                                              // 0 = undecided, 1 = a, 2 = b
                const bool rm_l_a = qlen(la) < min_len, rm_l_b = qlen(lb) < min_len, rm_r_a = qlen(ra) < min_len, rm_r_b = qlen(rb) < min_len;
                const bool rm_any_a = rm_l_a || rm_r_a, rm_any_b = rm_l_b || rm_r_b;
                if (rm_any_a && !rm_any_b) { if (!rm_l_a && rm_r_a) keep = 1; }
                else if (rm_any_b && !rm_any_a) { if (!rm_l_b && rm_r_b) keep = 2; }
                if (!keep && rm_any_a) keep = 1;
                if (!keep && rm_any_b) keep = 2;
                if (!keep) {
                    const int64_t trim_pos_l_a = !la.f.rev ? la.f.end : la.f.pos, trim_pos_l_b = !lb.f.rev ? lb.f.end : lb.f.pos;
                    keep = trim_pos_l_a <= trim_pos_l_b ? 1 : 2;
                }
                if (keep == 1) { record_l = std::move(la); record_r = std::move(ra); }
                else { record_l = std::move(rb); record_r = std::move(lb); }
            } else {
                if (fl.chrom == fr.chrom && rev_l != rev_r) {
                    const int64_t trim_pos_l = !fl.rev ? fl.end : fl.pos, trim_pos_r = !fr.rev ? fr.pos : fr.end;
                    if (trim_pos_r < trim_pos_l) { std::swap(rev_l, rev_r); std::swap(index_l, index_r); }
                }
                record_l = rows[index_l]; record_r = rows[index_r];
                if (!trim_record(*S, record_l, record_r, true, rev_l, rev_r, tf)) return record_fail(ctx, S, tf, index_l, index_r);
            }
            if (qlen(record_l) >= min_len) rows[index_l] = std::move(record_l); else rows[index_l].f.index = -1;
            if (qlen(record_r) >= min_len) rows[index_r] = std::move(record_r); else rows[index_r].f.index = -1;
        }
    }
    return PAV_OK;
}

// Reference-space pass (trim.py:264-333) over the rows in `order` (sorted by #CHROM, END - POS descending).
int pass_subject(pav_ctx *ctx, TrimState *S, const std::vector<uint32_t> &order, int64_t min_len, bool match_tig) {
    const size_t n = order.size();
    std::vector<Row> &rows = S->rows;
    for (size_t il = 0; il < n; ++il) {
        for (size_t ir = il + 1; ir < n && rows[order[il]].f.chrom == rows[order[ir]].f.chrom; ++ir) {
            if (rows[order[il]].f.index < 0 || rows[order[ir]].f.index < 0) continue;
            if (match_tig && rows[order[il]].f.qry_id != rows[order[ir]].f.qry_id) continue;
            uint32_t index_l, index_r;
            if (rows[order[il]].f.pos <= rows[order[ir]].f.pos) { index_l = order[il]; index_r = order[ir]; }
            else { index_l = order[ir]; index_r = order[il]; }
            if (!(rows[index_r].f.pos < rows[index_l].f.end)) continue;
            if (rows[index_r].f.end <= rows[index_l].f.end) { rows[index_r].f.index = -1; continue; }
            Row record_l = rows[index_l], record_r = rows[index_r];
            TrimFail tf;
            if (!trim_record(*S, record_l, record_r, false, true, false, tf)) return record_fail(ctx, S, tf, index_l, index_r);
            if (qlen(record_l) >= min_len) rows[index_l] = std::move(record_l); else rows[index_l].f.index = -1;
            if (qlen(record_r) >= min_len) rows[index_r] = std::move(record_r); else rows[index_r].f.index = -1;
        }
    }
    return PAV_OK;
}

void put_dec(std::string &s, uint64_t v) {
    char buf[24];
    int n = 0;
    do { buf[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) s += buf[--n];
}

// count_cigar (pavlib/align/align.py:534-664) of one record: spans, clipping and the structural checks, first failure wins.
pav_trim_count count_cigar(const TrimState &S, const Cigar &c) {
    pav_trim_count cnt{};
    const size_t n = c.size();
    bool lead = true;
    for (size_t k = 0; k < n && !cnt.err_kind; ++k) {
        const Op o = cigar_get(S, c, k);
        auto bad = [&](int kind) { cnt.err_kind = kind; cnt.err_op = (uint32_t)k; cnt.err_len = o.len; cnt.err_char = (uint32_t)OP_CHARS[o.code]; };
        if (lead && (o.code == OP_S || o.code == OP_H)) {
            if (o.code == OP_S) { if (cnt.clip_s_l > 0) bad(PAV_TRIM_CHECK_DUP_S_L); else cnt.clip_s_l = (int64_t)o.len; }
            else if (cnt.clip_h_l > 0) bad(PAV_TRIM_CHECK_DUP_H_L);
            else if (cnt.clip_s_l > 0) bad(PAV_TRIM_CHECK_S_BEFORE_H_L);
            else cnt.clip_h_l = (int64_t)o.len;
            continue;
        }
        lead = false;
        const bool clipped = cnt.clip_s_r > 0 || cnt.clip_h_r > 0;
        switch (o.code) {
            case OP_EQ: case OP_X: if (clipped) bad(PAV_TRIM_CHECK_CLIP_INSIDE); else { cnt.ref_bp += (int64_t)o.len; cnt.tig_bp += (int64_t)o.len; } break;
            case OP_I: if (clipped) bad(PAV_TRIM_CHECK_CLIP_INSIDE); else cnt.tig_bp += (int64_t)o.len; break;
            case OP_D: if (clipped) bad(PAV_TRIM_CHECK_CLIP_INSIDE); else cnt.ref_bp += (int64_t)o.len; break;
            case OP_S:
                if (cnt.clip_s_r > 0) bad(PAV_TRIM_CHECK_DUP_S_R);
                else if (cnt.clip_h_r > 0) bad(PAV_TRIM_CHECK_H_BEFORE_S_R);
                else cnt.clip_s_r = (int64_t)o.len;
                break;
            case OP_H: if (cnt.clip_h_r > 0) bad(PAV_TRIM_CHECK_DUP_H_R); else cnt.clip_h_r = (int64_t)o.len; break;
            case OP_M: bad(PAV_TRIM_CHECK_M); break;
            default: bad(PAV_TRIM_CHECK_BAD_OP); break;
        }
    }
    return cnt;
}

// ---- device passes (trim_dev.hip) -----------------------------------------------------------------------------------------
RowDev to_dev(const Row &r) {
    RowDev d{};
    d.f = r.f;
    d.c.a = r.c.a; d.c.b = r.c.b; d.c.len_first = r.c.len_first; d.c.len_last = r.c.len_last;
    d.c.n_pre = (uint8_t)r.c.pre.size(); d.c.n_post = (uint8_t)r.c.post.size(); d.c.modified = r.c.modified ? 1 : 0;
    for (size_t k = 0; k < r.c.pre.size() && k < 2; ++k) { d.c.pre_len[k] = r.c.pre[k].len; d.c.pre_code[k] = r.c.pre[k].code; }
    for (size_t k = 0; k < r.c.post.size() && k < 2; ++k) { d.c.post_len[k] = r.c.post[k].len; d.c.post_code[k] = r.c.post[k].code; }
    return d;
}
void from_dev(const RowDev &d, Row &r) {
    r.f = d.f;
    r.c.a = d.c.a; r.c.b = d.c.b; r.c.len_first = d.c.len_first; r.c.len_last = d.c.len_last; r.c.modified = d.c.modified != 0;
    r.c.pre.clear(); r.c.post.clear();
    for (uint32_t k = 0; k < d.c.n_pre; ++k) r.c.pre.push_back(Op{d.c.pre_len[k], d.c.pre_code[k]});
    for (uint32_t k = 0; k < d.c.n_post; ++k) r.c.post.push_back(Op{d.c.post_len[k], d.c.post_code[k]});
}
TrimFail fail_of(const TrimFailDev &f) {
    TrimFail t;
    t.kind = f.kind; t.op_index = f.op_index; t.op_len = f.op_len;
    // (the operation fields describe an illegal operation only; the host loops leave them zero for the other kinds)
    t.op_char = f.kind == PAV_TRIM_ERR_ILLEGAL_OP ? (uint32_t)OP_CHARS[f.op_code < 9 ? f.op_code : 0] : 0u;
    t.diff_bp = f.diff_bp; t.side = f.side;
    return t;
}
bool host_passes() { const char *e = getenv("PAV_TRIM_HOST"); return e && *e == '1'; }

// One pass on the device: the groups (records of one contig / one chromosome, consecutive in `ord`) get a wave each.
int pass_device(pav_ctx *ctx, TrimState *S, const std::vector<uint32_t> &ord, int mode, int64_t min_len, bool match_tig) {
    PAV_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = ord.size(), n_rows = S->rows.size();
    if (n == 0) return PAV_OK;
    for (const Row &r : S->rows) if (r.c.pre.size() > 2 || r.c.post.size() > 2) return fail(ctx, PAV_E_STATE, "pav_trim_pass: a record carries more than two clipping operations at one end");
    std::vector<uint32_t> group_off;
    std::vector<unsigned long long> cap, off;
    unsigned long long bytes = 0;
    for (size_t i = 0; i < n; ++i) {
        const pav_trim_row &f = S->rows[ord[i]].f;
        const bool new_group = i == 0 || (mode == PAV_TRIM_QUERY ? f.qry_id != S->rows[ord[i - 1]].f.qry_id : f.chrom != S->rows[ord[i - 1]].f.chrom);
        if (new_group) { group_off.push_back((uint32_t)i); cap.push_back(0); }
        const Cigar &c = S->rows[ord[i]].c;
        cap.back() = std::max<unsigned long long>(cap.back(), (c.b - c.a) + 8);
    }
    group_off.push_back((uint32_t)n);
    const uint32_t n_groups = (uint32_t)cap.size();
    for (uint32_t g = 0; g < n_groups; ++g) { cap[g] = (cap[g] + 1) & ~1ull; off.push_back(bytes); bytes += 2ull * cap[g] * 56ull; }
    std::vector<RowDev> dev(n_rows);
    for (size_t i = 0; i < n_rows; ++i) dev[i] = to_dev(S->rows[i]);
    hipStream_t st = ctx->stream;
    PAV_HIP(ctx, S->d_rows.reserve(sizeof(RowDev) * n_rows));
    PAV_HIP(ctx, S->d_order.reserve(4 * n));
    PAV_HIP(ctx, S->d_group.reserve(4 * ((size_t)n_groups + 1)));
    PAV_HIP(ctx, S->d_meta.reserve(16ull * n_groups));
    PAV_HIP(ctx, S->d_scratch.reserve(bytes + 64));
    PAV_HIP(ctx, S->d_err.reserve(8 + (sizeof(TrimFailDev) + 8) * (size_t)n_groups));
    unsigned long long *d_key = S->d_err.as<unsigned long long>();
    unsigned long long *d_slot_key = d_key + 1;
    TrimFailDev *d_slots = reinterpret_cast<TrimFailDev *>(d_slot_key + n_groups);
    PAV_HIP(ctx, hipMemcpyAsync(S->d_rows.p, dev.data(), sizeof(RowDev) * n_rows, hipMemcpyHostToDevice, st));
    PAV_HIP(ctx, hipMemcpyAsync(S->d_order.p, ord.data(), 4 * n, hipMemcpyHostToDevice, st));
    PAV_HIP(ctx, hipMemcpyAsync(S->d_group.p, group_off.data(), 4 * group_off.size(), hipMemcpyHostToDevice, st));
    PAV_HIP(ctx, hipMemcpyAsync(S->d_meta.p, off.data(), 8ull * n_groups, hipMemcpyHostToDevice, st));
    PAV_HIP(ctx, hipMemcpyAsync(S->d_meta.as<unsigned long long>() + n_groups, cap.data(), 8ull * n_groups, hipMemcpyHostToDevice, st));
    PAV_HIP(ctx, hipMemsetAsync(d_key, 0xFF, 8 + 8ull * n_groups, st));
    TrimPassArgs A;
    A.ops = S->d_ops.as<uint32_t>(); A.rows = S->d_rows.as<RowDev>(); A.order = S->d_order.as<uint32_t>(); A.group_off = S->d_group.as<uint32_t>();
    A.scratch = S->d_scratch.as<uint8_t>(); A.scratch_off = S->d_meta.as<unsigned long long>(); A.scratch_cap = A.scratch_off + n_groups;
    A.min_len = min_len; A.mode = mode; A.match_tig = match_tig ? 1 : 0;
    A.err_key = d_key; A.err_slots = d_slots; A.err_slot_key = d_slot_key;
    { const int rcl = trim_launch_pass(ctx, A, n_groups); if (rcl != PAV_OK) return rcl; }
    unsigned long long key = ~0ull;
    PAV_HIP(ctx, hipMemcpyAsync(&key, d_key, 8, hipMemcpyDeviceToHost, st));
    PAV_HIP(ctx, hipMemcpyAsync(dev.data(), S->d_rows.p, sizeof(RowDev) * n_rows, hipMemcpyDeviceToHost, st));
    PAV_HIP(ctx, hipStreamSynchronize(st));
    S->n_dev_passes += 1;
    if (key != ~0ull) {                                              // the pair the sequential loop would have failed on first
        std::vector<unsigned long long> keys(n_groups);
        std::vector<TrimFailDev> slots(n_groups);
        PAV_HIP(ctx, hipMemcpy(keys.data(), d_slot_key, 8ull * n_groups, hipMemcpyDeviceToHost));
        PAV_HIP(ctx, hipMemcpy(slots.data(), d_slots, sizeof(TrimFailDev) * n_groups, hipMemcpyDeviceToHost));
        // the rows come back as they are: the two of the error record stand as the failing pair met them (the pairs in front of
        // it in their group are done, the pair itself changed nothing) - what the reference prints in its message, and what the
        // host loops leave; every other row is unspecified and the table is marked undefined by the caller
        for (size_t i = 0; i < n_rows; ++i) from_dev(dev[i], S->rows[i]);
        for (uint32_t g = 0; g < n_groups; ++g)
            if (keys[g] == key) return record_fail(ctx, S, fail_of(slots[g]), slots[g].row_l, slots[g].row_r);
        return fail(ctx, PAV_E_STATE, "pav_trim_pass: a device pass failed without a record of the failure");
    }
    for (size_t i = 0; i < n_rows; ++i) from_dev(dev[i], S->rows[i]);
    return PAV_OK;
}

int pair_device(pav_ctx *ctx, TrimState *S, uint32_t row_l, uint32_t row_r, int mode, int rev_l, int rev_r) {
    PAV_HIP(ctx, hipSetDevice(ctx->device));
    RowDev dev[2] = {to_dev(S->rows[row_l]), to_dev(S->rows[row_r])};
    const unsigned long long cap1 = (std::max(S->rows[row_l].c.b - S->rows[row_l].c.a, S->rows[row_r].c.b - S->rows[row_r].c.a) + 9) & ~1ull;
    const unsigned long long meta[2] = {0ull, cap1};
    hipStream_t st = ctx->stream;
    PAV_HIP(ctx, S->d_rows.reserve(sizeof(RowDev) * 2));
    PAV_HIP(ctx, S->d_meta.reserve(16));
    PAV_HIP(ctx, S->d_scratch.reserve(2ull * cap1 * 56ull + 64));
    PAV_HIP(ctx, S->d_err.reserve(8 + sizeof(TrimFailDev) + 8));
    unsigned long long *d_key = S->d_err.as<unsigned long long>();
    PAV_HIP(ctx, hipMemcpyAsync(S->d_rows.p, dev, sizeof dev, hipMemcpyHostToDevice, st));
    PAV_HIP(ctx, hipMemcpyAsync(S->d_meta.p, meta, sizeof meta, hipMemcpyHostToDevice, st));
    PAV_HIP(ctx, hipMemsetAsync(d_key, 0xFF, 16, st));
    TrimPassArgs A{};
    A.ops = S->d_ops.as<uint32_t>(); A.rows = S->d_rows.as<RowDev>();
    A.scratch = S->d_scratch.as<uint8_t>(); A.scratch_off = S->d_meta.as<unsigned long long>(); A.scratch_cap = A.scratch_off + 1;
    A.mode = mode; A.err_key = d_key; A.err_slot_key = d_key + 1; A.err_slots = reinterpret_cast<TrimFailDev *>(d_key + 2);
    { const int rcl = trim_launch_pair(ctx, A, rev_l, rev_r); if (rcl != PAV_OK) return rcl; }
    unsigned long long key = ~0ull;
    TrimFailDev tf{};
    PAV_HIP(ctx, hipMemcpyAsync(&key, d_key, 8, hipMemcpyDeviceToHost, st));
    PAV_HIP(ctx, hipMemcpyAsync(&tf, d_key + 2, sizeof tf, hipMemcpyDeviceToHost, st));
    PAV_HIP(ctx, hipMemcpyAsync(dev, S->d_rows.p, sizeof dev, hipMemcpyDeviceToHost, st));
    PAV_HIP(ctx, hipStreamSynchronize(st));
    if (key != ~0ull) return record_fail(ctx, S, fail_of(tf), row_l, row_r);
    from_dev(dev[0], S->rows[row_l]);
    from_dev(dev[1], S->rows[row_r]);
    return PAV_OK;
}

template <class F> void parallel_rows(size_t n, F &&fn) {
    const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    const size_t nt = std::min<size_t>(hw, (n + 63) / 64);
    if (nt <= 1) { for (size_t i = 0; i < n; ++i) fn(i); return; }
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    for (size_t t = 0; t < nt; ++t)
        pool.emplace_back([&] { for (size_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i); });
    for (auto &th : pool) th.join();
}

}  // namespace
}  // namespace pav

using namespace pav;

extern "C" {

void pav_trim_release(pav_ctx *ctx) {
    if (!ctx || !ctx->trim) return;
    delete static_cast<TrimState *>(ctx->trim);
    ctx->trim = nullptr;
}

int pav_trim_load(pav_ctx *ctx, uint32_t n, const pav_trim_row *rows, const uint8_t *cigar_text, const uint64_t *cigar_off) {
    if (!ctx || (n && (!rows || !cigar_text || !cigar_off))) return PAV_E_ARG;
    TrimState *S = tstate(ctx);
    S->rows.clear();
    S->undefined = false;
    S->err = pav_trim_err{};
    // tokenise every CIGAR once, on the device (same kernels as the call path)
    std::vector<uint32_t> zero_pos(n, 0u);
    uint64_t n_ops = 0;
    int rc = pav_align_index(ctx, n, zero_pos.data(), cigar_text, cigar_off, &n_ops, nullptr, nullptr, nullptr, nullptr);
    if (rc != PAV_OK) return rc;
    S->ops.resize(n_ops);
    S->op_off.assign((size_t)n + 1, 0);
    rc = pav_align_index(ctx, n, zero_pos.data(), cigar_text, cigar_off, &n_ops, S->ops.data(), S->op_off.data(), nullptr, nullptr);
    if (rc != PAV_OK) return rc;
    // the passes run on the device (trim_dev.hip): the operations stay resident there
    PAV_HIP(ctx, S->d_ops.reserve(sizeof(uint32_t) * (n_ops + 16)));
    if (n_ops) PAV_HIP(ctx, hipMemcpy(S->d_ops.p, S->ops.data(), sizeof(uint32_t) * n_ops, hipMemcpyHostToDevice));
    S->rows.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        Row &r = S->rows[i];
        r.f = rows[i];
        r.f.modified = 0;
        r.c.a = S->op_off[i]; r.c.b = S->op_off[i + 1];
        if (r.c.b > r.c.a) { r.c.len_first = S->ops[r.c.a] >> 4; r.c.len_last = S->ops[r.c.b - 1] >> 4; }
    }
    return PAV_OK;
}

int pav_trim_pass(pav_ctx *ctx, uint32_t n_order, const uint32_t *order, int mode, int64_t min_trim_tig_len, int match_tig) {
    if (!ctx || (n_order && !order) || (mode != PAV_TRIM_QUERY && mode != PAV_TRIM_SUBJECT)) return PAV_E_ARG;
    TrimState *S = tstate(ctx);
    std::vector<uint32_t> ord(order, order + n_order);
    std::vector<uint8_t> seen(S->rows.size(), 0);
    for (uint32_t i : ord) {
        if (i >= S->rows.size() || seen[i]) return fail(ctx, PAV_E_ARG, "pav_trim_pass: order is not a set of loaded rows");
        seen[i] = 1;
    }
    if (S->undefined) return fail(ctx, PAV_E_STATE, "pav_trim_pass: the table is undefined after a failed pass; call pav_trim_load again");
    S->err = pav_trim_err{};
    // PAV_TRIM_HOST=1: the same loops on the host (one thread) - kept as the cross-check of the device pass
    const int rc = host_passes()
        ? (mode == PAV_TRIM_QUERY ? pass_query(ctx, S, ord, min_trim_tig_len) : pass_subject(ctx, S, ord, min_trim_tig_len, match_tig != 0))
        : pass_device(ctx, S, ord, mode, min_trim_tig_len, match_tig != 0);
    if (rc != PAV_OK) S->undefined = true;
    return rc;
}

int pav_trim_pair(pav_ctx *ctx, uint32_t row_l, uint32_t row_r, int mode, int rev_l, int rev_r) {
    if (!ctx || (mode != PAV_TRIM_QUERY && mode != PAV_TRIM_SUBJECT)) return PAV_E_ARG;
    TrimState *S = tstate(ctx);
    if (row_l >= S->rows.size() || row_r >= S->rows.size() || row_l == row_r) return fail(ctx, PAV_E_ARG, "pav_trim_pair: no such pair of loaded rows");
    if (S->undefined) return fail(ctx, PAV_E_STATE, "pav_trim_pair: the table is undefined after a failed pass; call pav_trim_load again");
    S->err = pav_trim_err{};                                        // (a pair that fails leaves both rows as they were, on both paths)
    if (!host_passes()) return pair_device(ctx, S, row_l, row_r, mode, rev_l, rev_r);
    Row l = S->rows[row_l], r = S->rows[row_r];
    TrimFail tf;
    if (!trim_record(*S, l, r, mode == PAV_TRIM_QUERY, rev_l != 0, rev_r != 0, tf)) return record_fail(ctx, S, tf, row_l, row_r);
    S->rows[row_l] = std::move(l);
    S->rows[row_r] = std::move(r);
    return PAV_OK;
}

int pav_trim_error(const pav_ctx *ctx, pav_trim_err *err) {
    if (!ctx || !err || !ctx->trim) return PAV_E_ARG;
    *err = static_cast<const TrimState *>(ctx->trim)->err;
    return PAV_OK;
}

int pav_trim_fetch(pav_ctx *ctx, pav_trim_row *rows, pav_trim_count *counts, uint64_t *cigar_bytes) {
    if (!ctx) return PAV_E_ARG;
    TrimState *S = tstate(ctx);
    // after a failed pass only the plain rows may be read (the two of the error record are as the failing pair met them: the
    // mirror prints their coordinates in the reference's message); counts and CIGAR strings of an undefined table are refused
    if (S->undefined && (counts || cigar_bytes))
        return fail(ctx, PAV_E_STATE, "pav_trim_fetch: the table is undefined after a failed pass; call pav_trim_load again");
    const size_t n = S->rows.size();
    if (rows) for (size_t i = 0; i < n; ++i) { rows[i] = S->rows[i].f; rows[i].modified = S->rows[i].c.modified ? 1 : 0; }
    if (counts) parallel_rows(n, [&](size_t i) { counts[i] = count_cigar(*S, S->rows[i].c); });
    if (cigar_bytes) {
        // CIGAR strings of the records trimming changed (the others kept their input string: empty here)
        std::vector<std::string> parts(n);
        parallel_rows(n, [&](size_t i) {
            const Cigar &c = S->rows[i].c;
            if (!c.modified) return;
            std::string &t = parts[i];
            const size_t m = c.size();
            t.reserve(m * 5 + 16);
            for (size_t k = 0; k < m; ++k) { const Op o = cigar_get(*S, c, k); put_dec(t, o.len); t += OP_CHARS[o.code]; }
        });
        S->text.clear();
        S->text_off.assign(n + 1, 0);
        size_t total = 0;
        for (size_t i = 0; i < n; ++i) { S->text_off[i] = total; total += parts[i].size(); }
        S->text_off[n] = total;
        S->text.resize(total);
        parallel_rows(n, [&](size_t i) { if (!parts[i].empty()) memcpy(&S->text[S->text_off[i]], parts[i].data(), parts[i].size()); });
        *cigar_bytes = total;
    }
    return PAV_OK;
}

int pav_trim_fetch_cigar(pav_ctx *ctx, uint8_t *text, uint64_t *off) {
    if (!ctx || !off) return PAV_E_ARG;
    TrimState *S = tstate(ctx);
    if (S->text_off.size() != S->rows.size() + 1) return fail(ctx, PAV_E_STATE, "pav_trim_fetch_cigar: call pav_trim_fetch with cigar_bytes first");
    if (!S->text.empty()) { if (!text) return PAV_E_ARG; memcpy(text, S->text.data(), S->text.size()); }
    memcpy(off, S->text_off.data(), sizeof(uint64_t) * S->text_off.size());
    return PAV_OK;
}

}  // extern "C"
