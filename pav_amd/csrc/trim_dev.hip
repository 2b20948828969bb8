// Alignment trimming on the device: the pair loops of pavlib.align.trim_alignments and
// trim_alignment_record / trace_cigar_to_zero / find_cut_sites (pavlib/align/trim.py:11-917) as ONE launch per pass.
//
// What is parallel and what is not.  A trimmed record is the input of the next pair that touches it, so the pair loop of a
// group - the records of one contig in the contig-space pass (trim.py:61-256), of one chromosome in the reference-space
// pass (trim.py:264-333) - is sequential by definition; but groups share no record, so every group gets a wave of its own,
// and inside a pair the two things that cost time are data-parallel:
//   * trace_cigar_to_zero (trim.py:779-917): six running sums over the operations from the overlapping end of a record, up to
//     the first operation at which the overlap is used up - a wave scan over 64 operations per step; the '=' / 'X' operations
//     in front of that point are the trace (compacted with ballot / prefix counts into the group's scratch);
//   * find_cut_sites (trim.py:602-776): the best pair of cut sites over (left trace entry) x (the right entries that can close
//     the overlap with it) - lane = one left entry, its window of right entries found by two binary searches (the sequential
//     code moves one pointer monotonically: the same window), wave arg-max in the reference's order of preference: more events
//     removed, then fewer bases over-trimmed, then the first met by the reference's loops (left entries from the last, right
//     entries upwards).
// The control around them (which pairs overlap, both trim orders when the records also overlap on the reference, contained
// and too-short records) is the reference's, executed uniformly by all lanes of the group's wave.
// A record's CIGAR is a window [a, b) into the tokenised operation array plus at most two clipping operations trimming adds at
// either end (host struct Cigar in trim.cpp; here CigDev).  gfx950 only.
#include "common.h"
#include "trim_dev.h"

namespace pav {
namespace {

constexpr uint32_t T_I = 1, T_D = 2, T_S = 4, T_H = 5, T_EQ = 7, T_X = 8;

struct TOp { uint64_t len; uint32_t code; };

__device__ __forceinline__ uint32_t cig_size(const CigDev &c) { return c.n_pre + (uint32_t)(c.b - c.a) + c.n_post; }

__device__ __forceinline__ TOp cig_get(const uint32_t *__restrict__ ops, const CigDev &c, uint32_t i) {
    if (i < c.n_pre) return TOp{c.pre_len[i], c.pre_code[i]};
    i -= c.n_pre;
    const uint32_t w = (uint32_t)(c.b - c.a);
    if (i < w) {
        const uint32_t o = ops[c.a + i];
        uint64_t len = o >> 4;
        if (i == 0) len = c.len_first;
        if (i + 1 == w) len = (w == 1) ? c.len_first : c.len_last;
        return TOp{len, o & 15u};
    }
    return TOp{c.post_len[i - w], c.post_code[i - w]};
}

// The trace of one record in the group's scratch (structure of arrays; entry t = the t-th '=' / 'X' operation met)
struct TraceRef {
    uint32_t *idx;          // TC_INDEX: index in the oriented operation list
    uint32_t *opw;          // length << 4 | code (the diff of an '=' / 'X' entry is its length; its events: the length of an 'X')
    long long *diff_cum, *event_cum, *sub_bp, *qry_bp, *clip_s, *clip_h;
    uint32_t n;
};

struct TraceEnt { uint32_t idx, code; long long len, diff_cum, event_cum, sub_bp, qry_bp, clip_s, clip_h; };

__device__ __forceinline__ TraceEnt trace_ent(const TraceRef &t, uint32_t i) {
    const uint32_t w = t.opw[i];
    return TraceEnt{t.idx[i], w & 15u, (long long)(w >> 4), t.diff_cum[i], t.event_cum[i], t.sub_bp[i], t.qry_bp[i], t.clip_s[i], t.clip_h[i]};
}

__device__ __forceinline__ long long wave_incl_scan(long long v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const long long y = __shfl_up(v, d); if (lane >= d) v += y; }
    return v;
}

// trace_cigar_to_zero, trim.py:779-917.  Every lane returns the same values.  false: illegal operation (fail filled in).
__device__ bool trace_to_zero(const uint32_t *__restrict__ ops, const CigDev &c, bool rev, long long diff_bp, bool diff_query,
                              TraceRef &out, TrimFailDev &fail) {
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const uint32_t n = cig_size(c);
    long long diff_c = 0, event_c = 0, sub_c = 0, qry_c = 0, cs_c = 0, ch_c = 0;     // sums over the operations of the chunks before
    bool prev_no_match = false;                                                        // last_no_match after the operation in front
    uint32_t n_out = 0;
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t i = base + (uint32_t)lane;
        const bool live = i < n;
        TOp o{0, T_EQ};
        if (live) o = cig_get(ops, c, rev ? n - 1 - i : i);
        const long long len = (long long)o.len;
        long long ev = 0, sub = 0, qry = 0, cs = 0, ch = 0;
        bool no_match = true, legal = true;
        switch (o.code) {
            case T_EQ: sub = len; qry = len; no_match = false; break;
            case T_X: ev = len; sub = len; qry = len; break;
            case T_I: ev = 1; qry = len; break;
            case T_D: ev = 1; sub = len; break;
            case T_S: cs = len; break;
            case T_H: ch = len; break;
            default: legal = false; break;
        }
        if (!live) { ev = sub = qry = cs = ch = 0; legal = true; }
        const long long dchange = diff_query ? qry : sub;
        const bool is_match = live && (o.code == T_EQ || o.code == T_X);
        // exclusive running values in front of operation i
        const long long d_in = wave_incl_scan(dchange, lane), e_in = wave_incl_scan(ev, lane), s_in = wave_incl_scan(sub, lane),
                        q_in = wave_incl_scan(qry, lane), cs_in = wave_incl_scan(cs, lane), ch_in = wave_incl_scan(ch, lane);
        const long long d_ex = diff_c + d_in - dchange, e_ex = event_c + e_in - ev, s_ex = sub_c + s_in - sub, q_ex = qry_c + q_in - qry;
        // clipping counts the operation itself (clip_s_sum / clip_h_sum are advanced before the record is appended, :872-878; an
        // '=' / 'X' adds nothing to them)
        const long long cs_at = cs_c + cs_in, ch_at = ch_c + ch_in;
        const unsigned long long m_match = __ballot(is_match);
        const uint32_t matches_before = n_out + (uint32_t)__popcll(m_match & lt);
        // last_no_match in front of operation i: of the lane below, or carried in
        const int below_nm = __shfl_up((int)no_match, 1);
        const bool lnm = lane ? below_nm != 0 : prev_no_match;
        // the loop condition in front of operation i (:851): keep going while the overlap is not used up, or the last operation was
        // not a match, or nothing has been traced yet
        const bool go = live && (d_ex <= diff_bp || lnm || matches_before == 0);
        const unsigned long long m_stop = __ballot(live && !go);
        const uint32_t stop_lane = m_stop ? (uint32_t)__ffsll((long long)m_stop) - 1 : 64u;      // first operation that is not processed
        const bool processed = live && (uint32_t)lane < stop_lane;
        const unsigned long long m_bad = __ballot(processed && !legal);
        if (m_bad) {
            const int bl = __ffsll((long long)m_bad) - 1;
            fail.kind = PAV_TRIM_ERR_ILLEGAL_OP;
            fail.op_index = base + (uint32_t)bl;
            fail.op_len = (unsigned long long)__shfl((long long)o.len, bl);
            fail.op_code = (uint32_t)__shfl((int)o.code, bl);
            return false;
        }
        if (processed && is_match) {
            const uint32_t t = matches_before;
            out.idx[t] = i; out.opw[t] = (uint32_t)(o.len << 4) | o.code;
            out.diff_cum[t] = d_ex; out.event_cum[t] = e_ex; out.sub_bp[t] = s_ex; out.qry_bp[t] = q_ex; out.clip_s[t] = cs_at; out.clip_h[t] = ch_at;
        }
        const unsigned long long m_proc = stop_lane >= 64 ? ~0ull : ((1ull << stop_lane) - 1ull);
        n_out += (uint32_t)__popcll(m_match & m_proc);
        if (stop_lane < 64) break;
        diff_c += __shfl(d_in, 63); event_c += __shfl(e_in, 63); sub_c += __shfl(s_in, 63); qry_c += __shfl(q_in, 63);
        cs_c += __shfl(cs_in, 63); ch_c += __shfl(ch_in, 63);
        prev_no_match = __shfl((int)no_match, 63) != 0;
    }
    out.n = n_out;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // the entries are read by other lanes of this wave below
    __builtin_amdgcn_wave_barrier();
    return true;
}

// find_cut_sites, trim.py:602-776.  false: no cut site.
struct CutKey { long long event; long long diff_opt; uint32_t order_l, r; bool have; };

__device__ __forceinline__ bool cut_better(const CutKey &a, const CutKey &b) {      // a strictly preferred over b
    if (!a.have) return false;
    if (!b.have) return true;
    if (a.event != b.event) return a.event > b.event;
    if (a.diff_opt != b.diff_opt) return a.diff_opt < b.diff_opt;
    if (a.order_l != b.order_l) return a.order_l < b.order_l;                       // met earlier by the outer loop (it starts at the last left entry)
    return a.r < b.r;
}

__device__ bool find_cuts(const TraceRef &tl, const TraceRef &tr, long long diff_bp, uint32_t &cut_l, uint32_t &cut_r) {
    const int lane = threadIdx.x & 63;
    const uint32_t len_l = tl.n, len_r = tr.n;
    if (len_l == 0 || len_r == 0) return false;
    CutKey best{0, 0, 0, 0, false};
    for (uint32_t base = 0; base < len_l; base += 64) {
        const uint32_t ol = base + (uint32_t)lane;                                  // position in the reference's outer loop
        CutKey mine{0, 0, ol, 0, false};
        if (ol < len_l) {
            const uint32_t il = len_l - 1 - ol;
            const uint32_t lw = tl.opw[il];
            const long long l_diff = (long long)(lw >> 4), l_event = (lw & 15u) == T_X ? l_diff : 0;
            const long long l_dc = tl.diff_cum[il], l_ec = tl.event_cum[il];
            const long long min_bp_l = l_dc, max_bp_l = l_dc + l_diff - 1;
            // first right entry that can close the overlap with this one - or the last one (:680-687; the reference advances one
            // pointer over the outer loop: max_bp_l falls, the start never moves back)
            uint32_t lo = 0, hi = len_r - 1;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                const long long rdiff = (long long)(tr.opw[mid] >> 4);
                if (max_bp_l + tr.diff_cum[mid] + rdiff - 1 < diff_bp) lo = mid + 1; else hi = mid;
            }
            const uint32_t r0 = lo;
            for (uint32_t r = r0; r < len_r; ++r) {
                const long long r_dc = tr.diff_cum[r];
                if (!(min_bp_l + r_dc <= diff_bp || r == r0)) break;
                const uint32_t rw = tr.opw[r];
                const long long r_diff = (long long)(rw >> 4), r_event = (rw & 15u) == T_X ? r_diff : 0;
                const long long max_bp = max_bp_l + r_dc + r_diff - 1;
                const long long diff_min = diff_bp - max_bp;
                long long event_count = l_ec + tr.event_cum[r], diff_optimal;
                if (diff_min <= 0) {
                    const long long cap = l_event + r_event - (l_event > 0 ? 1 : 0) - (r_event > 0 ? 1 : 0);
                    const long long room = diff_bp - diff_min;
                    event_count += room < cap ? room : cap;
                    diff_optimal = 0;
                } else diff_optimal = diff_min;
                const CutKey cand{event_count, diff_optimal, ol, r, true};
                if (cut_better(cand, mine)) mine = cand;
            }
        }
        // wave arg-max, then against the best of the chunks before (they come earlier in the reference's order)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            CutKey o;
            o.event = __shfl_xor(mine.event, d); o.diff_opt = __shfl_xor(mine.diff_opt, d);
            o.order_l = (uint32_t)__shfl_xor((int)mine.order_l, d); o.r = (uint32_t)__shfl_xor((int)mine.r, d);
            o.have = __shfl_xor((int)mine.have, d) != 0;
            if (cut_better(o, mine)) mine = o;
        }
        if (cut_better(mine, best)) best = mine;
    }
    if (!best.have) return false;
    cut_l = len_l - 1 - best.order_l; cut_r = best.r;
    return true;
}

// Apply one cut to a record (trim.py:497-593): drop the operations before `cut` in the oriented list, shorten the surviving one by
// `trim`, put the accumulated clipping in front, fix coordinates and TRIM_* counters.
__device__ void apply_cut(RowDev &r, bool rev, const TraceEnt &cut, long long trim) {
    const long long cut_sub = cut.sub_bp + trim, cut_qry = cut.qry_bp + trim;
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
    // clipping in the oriented direction: H, then S
    unsigned long long cl_len[2]; uint8_t cl_code[2]; uint32_t n_cl = 0;
    if (cut.clip_h > 0) { cl_len[n_cl] = (unsigned long long)cut.clip_h; cl_code[n_cl] = (uint8_t)T_H; ++n_cl; }
    const long long clip_s = cut.clip_s + cut.qry_bp + trim;
    if (clip_s > 0) { cl_len[n_cl] = (unsigned long long)clip_s; cl_code[n_cl] = (uint8_t)T_S; ++n_cl; }
    CigDev &c = r.c;
    const uint32_t n = cig_size(c);
    const uint32_t fwd = rev ? n - 1 - cut.idx : cut.idx;         // index in stored orientation; an '=' / 'X': inside the window
    const uint32_t w_i = fwd - c.n_pre;
    const unsigned long long new_len = (unsigned long long)(cut.len - trim);
    if (!rev) {
        const bool single = (c.b - c.a) - w_i == 1;
        c.a += w_i;
        c.len_first = new_len;
        if (single) c.len_last = new_len;
        c.n_pre = (uint8_t)n_cl;
        for (uint32_t k = 0; k < n_cl; ++k) { c.pre_len[k] = cl_len[k]; c.pre_code[k] = cl_code[k]; }
    } else {
        const bool single = w_i == 0;
        c.b = c.a + w_i + 1;
        c.len_last = new_len;
        if (single) c.len_first = new_len;
        c.n_post = (uint8_t)n_cl;                                    // stored orientation: the oriented list reversed
        for (uint32_t k = 0; k < n_cl; ++k) { c.post_len[k] = cl_len[n_cl - 1 - k]; c.post_code[k] = cl_code[n_cl - 1 - k]; }
    }
    c.modified = 1;
}

struct Scratch { TraceRef l, r; };

__device__ __forceinline__ TraceRef trace_at(uint8_t *base, uint64_t cap, uint32_t which) {
    // two traces per group, `cap` entries each: idx | opw | six running sums
    uint8_t *p = base + (uint64_t)which * cap * 56ull;
    TraceRef t;
    t.idx = reinterpret_cast<uint32_t *>(p); t.opw = t.idx + cap;
    t.diff_cum = reinterpret_cast<long long *>(p + 8ull * cap);
    t.event_cum = t.diff_cum + cap; t.sub_bp = t.event_cum + cap; t.qry_bp = t.sub_bp + cap; t.clip_s = t.qry_bp + cap; t.clip_h = t.clip_s + cap;
    t.n = 0;
    return t;
}

// trim_alignment_record (trim.py:357-599).  `l`, `r` are modified copies on success.
__device__ bool trim_record(const uint32_t *__restrict__ ops, Scratch &sc, RowDev &l, RowDev &r, bool query, bool rev_l, bool rev_r,
                            TrimFailDev &fail) {
    long long diff_bp;
    if (query) {
        if (l.f.qry_pos < r.f.qry_pos) diff_bp = l.f.qry_end - r.f.qry_pos;
        else diff_bp = r.f.qry_end - l.f.qry_pos;
        if (diff_bp <= 0) { fail.kind = PAV_TRIM_ERR_NEGATIVE; fail.diff_bp = diff_bp; return false; }
    } else {
        if (l.f.pos > r.f.pos) { fail.kind = PAV_TRIM_ERR_ORDER; return false; }
        diff_bp = l.f.end - r.f.pos;
        if (diff_bp <= 0) { fail.kind = PAV_TRIM_ERR_NEGATIVE; fail.diff_bp = diff_bp; return false; }
    }
    if (!trace_to_zero(ops, l.c, rev_l, diff_bp, query, sc.l, fail)) { fail.side = 0; return false; }
    if (!trace_to_zero(ops, r.c, rev_r, diff_bp, query, sc.r, fail)) { fail.side = 1; return false; }
    uint32_t ci_l = 0, ci_r = 0;
    if (!find_cuts(sc.l, sc.r, diff_bp, ci_l, ci_r)) { fail.kind = PAV_TRIM_ERR_NO_CUT; return false; }
    const TraceEnt cut_l = trace_ent(sc.l, ci_l), cut_r = trace_ent(sc.r, ci_r);
    // mid-record cuts: left-align, mismatches first (trim.py:475-494)
    long long residual = diff_bp - (cut_l.diff_cum + cut_r.diff_cum), trim_l = 0, trim_r = 0;
    auto take = [&](long long &t, long long len) { const long long x = residual < len - 1 ? residual : len - 1; t += x; residual -= t; };
    if (residual > 0 && cut_r.code == T_X) take(trim_r, cut_r.len);
    if (residual > 0 && cut_l.code == T_X) take(trim_l, cut_l.len);
    if (residual > 0 && cut_l.code == T_EQ) take(trim_l, cut_l.len);
    if (residual > 0 && cut_r.code == T_EQ) take(trim_r, cut_r.len);
    apply_cut(l, rev_l, cut_l, trim_l);
    apply_cut(r, rev_r, cut_r, trim_r);
    return true;
}

__device__ __forceinline__ long long qlen(const RowDev &r) { return r.f.qry_end - r.f.qry_pos; }

__device__ void report(TrimPassArgs &A, const TrimFailDev &f, uint32_t il, uint32_t ir, uint32_t row_l, uint32_t row_r) {   // il, ir: positions in `order`
    if ((threadIdx.x & 63) != 0) return;
    const unsigned long long key = (unsigned long long)il << 32 | ir;                   // the sequential loop meets the smallest first
    const unsigned long long old = atomicMin(A.err_key, key);
    if (key < old) {                                                                    // (one wave per group, one error per wave)
        TrimFailDev g = f;
        g.row_l = row_l; g.row_r = row_r;
        A.err_slots[blockIdx.x] = g;
        A.err_slot_key[blockIdx.x] = key;
    }
}

// One pair of the contig-space pass (the body of the loops of trim.py:61-256).  false: trimming failed (reported).
__device__ bool pair_query(TrimPassArgs &A, Scratch &sc, uint32_t il, uint32_t ir, long long min_len) {
    RowDev *rows = A.rows;
    const uint32_t *order = A.order;
    const bool lane0 = (threadIdx.x & 63) == 0;
    auto store = [&](uint32_t idx, const RowDev &v, bool keep) {                        // rows[idx] = v, or drop the record
        if (lane0) { if (keep) rows[idx] = v; else rows[idx].f.index = -1; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    };
    uint32_t index_l, index_r;
    if (rows[order[il]].f.qry_pos <= rows[order[ir]].f.qry_pos) { index_l = order[il]; index_r = order[ir]; }
    else { index_l = order[ir]; index_r = order[il]; }
    if (rows[index_l].f.index < 0 || rows[index_r].f.index < 0) return true;
    if (rows[index_r].f.qry_pos >= rows[index_l].f.qry_end) return true;
    if (rows[index_r].f.qry_end <= rows[index_l].f.qry_end) { store(index_r, rows[index_r], false); return true; }
    bool rev_l = !rows[index_l].f.rev, rev_r = rows[index_r].f.rev != 0;
    bool ref_overlap = false;
    const pav_trim_row fl = rows[index_l].f, fr = rows[index_r].f;
    if (!(rev_l == rev_r || fl.chrom != fr.chrom)) {
        if (fl.pos < fr.pos) ref_overlap = fr.pos < fl.end;
        else if (fr.pos < fl.pos) ref_overlap = fl.pos < fr.end;
    }
    RowDev record_l, record_r;
    TrimFailDev tf{};
    if (ref_overlap) {
        // try both trim orders and keep the one that left-aligns best (trim.py:128-197)
        RowDev la = rows[index_l], ra = rows[index_r];
        if (!trim_record(A.ops, sc, la, ra, true, rev_l, rev_r, tf)) { report(A, tf, il, ir, index_l, index_r); return false; }
        RowDev lb = rows[index_r], rb = rows[index_l];
        if (!trim_record(A.ops, sc, lb, rb, true, rev_r, rev_l, tf)) { report(A, tf, il, ir, index_r, index_l); return false; }
        int keep = 0;                                              // 0 = undecided, 1 = a, 2 = b
        const bool rm_l_a = qlen(la) < min_len, rm_l_b = qlen(lb) < min_len, rm_r_a = qlen(ra) < min_len, rm_r_b = qlen(rb) < min_len;
        const bool rm_any_a = rm_l_a || rm_r_a, rm_any_b = rm_l_b || rm_r_b;
        if (rm_any_a && !rm_any_b) { if (!rm_l_a && rm_r_a) keep = 1; }
        else if (rm_any_b && !rm_any_a) { if (!rm_l_b && rm_r_b) keep = 2; }
        if (!keep && rm_any_a) keep = 1;
        if (!keep && rm_any_b) keep = 2;
        if (!keep) {
            const long long trim_pos_l_a = !la.f.rev ? la.f.end : la.f.pos, trim_pos_l_b = !lb.f.rev ? lb.f.end : lb.f.pos;
            keep = trim_pos_l_a <= trim_pos_l_b ? 1 : 2;
        }
        if (keep == 1) { record_l = la; record_r = ra; }
        else { record_l = rb; record_r = lb; }
    } else {
        if (fl.chrom == fr.chrom && rev_l != rev_r) {
            const long long trim_pos_l = !fl.rev ? fl.end : fl.pos, trim_pos_r = !fr.rev ? fr.pos : fr.end;
            if (trim_pos_r < trim_pos_l) { const bool t = rev_l; rev_l = rev_r; rev_r = t; const uint32_t u = index_l; index_l = index_r; index_r = u; }
        }
        record_l = rows[index_l]; record_r = rows[index_r];
        if (!trim_record(A.ops, sc, record_l, record_r, true, rev_l, rev_r, tf)) { report(A, tf, il, ir, index_l, index_r); return false; }
    }
    store(index_l, record_l, qlen(record_l) >= min_len);
    store(index_r, record_r, qlen(record_r) >= min_len);
    return true;
}

// One pair of the reference-space pass (trim.py:264-333).
__device__ bool pair_subject(TrimPassArgs &A, Scratch &sc, uint32_t il, uint32_t ir, long long min_len) {
    RowDev *rows = A.rows;
    const uint32_t *order = A.order;
    const bool lane0 = (threadIdx.x & 63) == 0;
    auto store = [&](uint32_t idx, const RowDev &v, bool keep) {
        if (lane0) { if (keep) rows[idx] = v; else rows[idx].f.index = -1; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    };
    if (rows[order[il]].f.index < 0 || rows[order[ir]].f.index < 0) return true;
    if (A.match_tig && rows[order[il]].f.qry_id != rows[order[ir]].f.qry_id) return true;
    uint32_t index_l, index_r;
    if (rows[order[il]].f.pos <= rows[order[ir]].f.pos) { index_l = order[il]; index_r = order[ir]; }
    else { index_l = order[ir]; index_r = order[il]; }
    if (!(rows[index_r].f.pos < rows[index_l].f.end)) return true;
    if (rows[index_r].f.end <= rows[index_l].f.end) { store(index_r, rows[index_r], false); return true; }
    RowDev record_l = rows[index_l], record_r = rows[index_r];
    TrimFailDev tf{};
    if (!trim_record(A.ops, sc, record_l, record_r, false, true, false, tf)) { report(A, tf, il, ir, index_l, index_r); return false; }
    store(index_l, record_l, qlen(record_l) >= min_len);
    store(index_r, record_r, qlen(record_r) >= min_len);
    return true;
}

}  // namespace

// One wave per group: order[g0 .. g1) are the group's rows in the reference's iteration order.  The pair loop is the
// reference's; what the wave does in parallel is LOOK for the next pair of an outer row that needs work: lane = one candidate
// inner row, tested against the outer row's current coordinates (most pairs of a chromosome's rows do not overlap); the pairs
// found are handled one after the other, in order, and the candidates behind a handled pair are tested again, because
// trimming has changed the outer row.
__global__ __launch_bounds__(64) void trim_pass_kernel(TrimPassArgs A) {
    const uint32_t g = blockIdx.x;
    const uint32_t g0 = A.group_off[g], g1 = A.group_off[g + 1];
    const int lane = threadIdx.x & 63;
    Scratch sc;
    sc.l = trace_at(A.scratch + A.scratch_off[g], A.scratch_cap[g], 0);
    sc.r = trace_at(A.scratch + A.scratch_off[g], A.scratch_cap[g], 1);
    RowDev *rows = A.rows;
    const uint32_t *order = A.order;
    const bool query = A.mode == PAV_TRIM_QUERY;
    for (uint32_t il = g0; il < g1; ++il) {
        const uint32_t row_o = order[il];
        for (uint32_t base = il + 1; base < g1; base += 64) {
            const uint32_t ir = base + (uint32_t)lane;
            const bool live = ir < g1;
            const uint32_t row_i = live ? order[ir] : row_o;
            uint32_t from = 0;                                         // candidates in front of this lane have been dealt with
            while (true) {
                // does the sequential loop do anything for (il, ir) with the rows as they are now?  (exactly its skip tests)
                const pav_trim_row fo = rows[row_o].f, fi = rows[row_i].f;
                bool work = live && (uint32_t)lane >= from && fo.index >= 0 && fi.index >= 0;
                if (work) {
                    if (query) {
                        const bool o_first = fo.qry_pos <= fi.qry_pos;
                        const long long l_end = o_first ? fo.qry_end : fi.qry_end, r_pos = o_first ? fi.qry_pos : fo.qry_pos;
                        work = r_pos < l_end;                          // overlap in contig space: contained (dropped) or trimmed
                    } else {
                        if (A.match_tig && fo.qry_id != fi.qry_id) work = false;
                        else {
                            const bool o_first = fo.pos <= fi.pos;
                            const long long l_end = o_first ? fo.end : fi.end, r_pos = o_first ? fi.pos : fo.pos;
                            work = r_pos < l_end;
                        }
                    }
                }
                const unsigned long long m = __ballot(work);
                if (!m) break;
                const uint32_t sel = (uint32_t)__ffsll((long long)m) - 1;
                const bool ok = query ? pair_query(A, sc, il, base + sel, A.min_len) : pair_subject(A, sc, il, base + sel, A.min_len);
                if (!ok) return;
                from = sel + 1;
            }
        }
    }
}

// trim_alignment_record on one pair (pav_trim_pair): rows[0] = record_l, rows[1] = record_r, both replaced on success.
__global__ __launch_bounds__(64) void trim_pair_kernel(TrimPassArgs A, int rev_l, int rev_r) {
    Scratch sc;
    sc.l = trace_at(A.scratch, A.scratch_cap[0], 0);
    sc.r = trace_at(A.scratch, A.scratch_cap[0], 1);
    RowDev l = A.rows[0], r = A.rows[1];
    TrimFailDev tf{};
    if (!trim_record(A.ops, sc, l, r, A.mode == PAV_TRIM_QUERY, rev_l != 0, rev_r != 0, tf)) { report(A, tf, 0, 1, 0, 1); return; }
    if ((threadIdx.x & 63) == 0) { A.rows[0] = l; A.rows[1] = r; }
}

int trim_launch_pass(pav_ctx *ctx, const TrimPassArgs &A, uint32_t n_groups) {
    PAV_LAUNCH(ctx, "trim_pass_kernel", trim_pass_kernel, n_groups, 64, 0, A);
    return PAV_OK;
}
int trim_launch_pair(pav_ctx *ctx, const TrimPassArgs &A, int rev_l, int rev_r) {
    PAV_LAUNCH(ctx, "trim_pair_kernel", trim_pair_kernel, 1, 64, 0, A, rev_l, rev_r);
    return PAV_OK;
}

}  // namespace pav
