// Native batched driver for the inversion scan: the control flow of pavlib.inv.scan_for_inv (pavlib/inv.py:149-454)
// for many flagged regions in lock-step, with lift-over (pavlib/align/lift.py) and region arithmetic
// (pavlib/seq.py:112-188) on the host in C++ and every scan iteration one batched device call (density.hip).
// The Python mirror (pav_amd/inv.py) implements the same state machine; this driver exists because at ~1k regions per
// haplotype the interpreter, not the GPU, bounded throughput (bench.py --workload cigar+inv).  Log lines, soft / hard
// failure behaviour and every coordinate are identical to the reference; tests compare both drivers with the golden
// vectors the reference produced.
#include "common.h"
#include "textio.h"
#include "pool.h"
#include "lift_dev.h"
#include "textdev.h"

#include <sys/mman.h>
#include <new>

#include <algorithm>
#include <cmath>
#include <map>
#include <chrono>
#include <memory>

namespace pav {

struct LiftRow { uint32_t ref_id, tig_id; int64_t pos, end, qry_pos, qry_end; int rev; int64_t index; };

struct InvTable {           // density table of one call: views into a pinned host block filled when the call is made
    uint32_t n = 0;
    uint32_t *index = nullptr; int8_t *state_mer = nullptr, *state = nullptr; double *kern[3] = {nullptr, nullptr, nullptr};
    uint64_t *kmer = nullptr; uint8_t *flank = nullptr, *match = nullptr;
};

struct Rgn {                                                  // pavlib.seq.Region (only what the scan uses)
    int chrom = -1; int role = 0;
    int64_t pos = 0, end = 0;
    bool is_rev = false;
    int64_t aln[2][2] = {{0, 0}, {0, 0}}; int n_aln[2] = {0, 0};   // pos_aln_index / end_aln_index (flattened)
    int64_t len() const { return end - pos; }
};

struct Scan {                       // per flagged region
    Rgn flag, region_ref, region_tig;
    int expansion_count = 0;
    bool done = false;
    std::vector<pav_run> state_rl;
    uint32_t n_rows = 0;
};

// The lift tables of a haplotype (operations and their begins on both axes: 3 x 36 MB) are read at random - two point queries per
// flagged region and scan round, each three dependent misses.  On 4 KiB pages every miss is a TLB miss as well (and with six
// resident haplotypes in one process the page tables themselves fall out of the caches: the same lookups took twice as long as
// in a process with one); the tables sit on 2 MiB pages.
template <class T> class HugeArray {
public:
    HugeArray() = default;
    HugeArray(const HugeArray &) = delete;
    HugeArray &operator=(const HugeArray &) = delete;
    ~HugeArray() { free(p_); }
    void resize(size_t n) {                                  // contents are not kept
        if (n > cap_) {
            free(p_);
            const size_t bytes = (std::max<size_t>(n, 1) * sizeof(T) + (2u << 20) - 1) & ~((size_t)(2u << 20) - 1);
            p_ = static_cast<T *>(aligned_alloc(2u << 20, bytes));
            if (!p_) throw std::bad_alloc();
            if (!getenv("PAV_NO_HUGE")) (void)madvise(p_, bytes, MADV_HUGEPAGE);
            cap_ = bytes / sizeof(T);
        }
        n_ = n;
    }
    T *data() { return p_; }
    const T *data() const { return p_; }
    size_t size() const { return n_; }
    T &operator[](size_t i) { return p_[i]; }
    const T &operator[](size_t i) const { return p_[i]; }
private:
    T *p_ = nullptr; size_t n_ = 0, cap_ = 0;
};

struct InvState {
    // lift-over index (pav_inv_load_alignments)
    std::vector<LiftRow> rows;
    HugeArray<uint32_t> ops, sub_begin, qry_begin;
    // first level of op_at, per axis: a record's begins are cut into position buckets of 2^shift bases (about 16 operations per
    // bucket on average); bucket[bucket_off[row] + b] = operations of the record (relative to its first) that begin before
    // base + (b << shift).  One read bounds the search in the 36 MB begin array to a few lines.
    struct Buckets { std::vector<uint32_t> first; std::vector<uint64_t> off; std::vector<uint8_t> shift; std::vector<uint32_t> base; } buckets[2];
    std::vector<uint64_t> op_off;
    std::vector<std::vector<uint32_t>> by_ref, by_tig;      // rows per reference / contig record, sorted by start
    std::vector<int64_t> by_ref_maxlen, by_tig_maxlen;
    // the same rows as columns (start, end, largest end so far), for the point queries of the lifts
    struct Spans { std::vector<int64_t> begin, end, max_end; };
    std::vector<Spans> span_ref, span_tig;
    std::vector<std::string> names[2];
    std::vector<uint8_t> row_bad;                           // first N (3) / P (6) operation code of a record, 0: none (lift.py:463-471)
    // The same index on the device (lift_dev.hip; the default): the operation tables stay where pav_align_index built them, the
    // point lifts of a scan round are one kernel.  PAV_LIFT_HOST=1 keeps the host tables above instead (the cross-check).
    bool on_device = false;
    DevBuf d_ops, d_begin, d_small, d_q, d_a;
    DevBuf d_rin, d_rq, d_ra;                               // a round's decisions on the device: job descriptors in, queries + answers out
    void *h_round = nullptr; size_t h_round_cap = 0;        // pinned: the descriptors up, the queries and answers down
    LiftTables dev{};
    void *h_qa = nullptr; size_t h_qa_cap = 0;              // pinned: queries up, answers down
    std::unique_ptr<HostPool> pool;                         // helper threads of the per-region host loops (pav_inv_scan_batch)
    std::vector<Scan> scans;                                // per-region state of the running scan (kept: its vectors keep their memory)
    bool loaded = false;
    uint64_t n_round_dev = 0, n_round_host = 0;             // rounds whose lifts came with the batch / were asked for after it (PAV_TIMING)
    // last scan
    std::vector<pav_inv_result> results;
    std::vector<std::string> logs;
    std::vector<std::string> errors;
    std::vector<std::string> found;                          // 'INV Found: ...' line of a region (inv.py:408), or empty
    std::vector<std::unique_ptr<InvTable>> tables;
    // Pinned host memory for the call tables: blocks persist across scans (pinning is expensive) and are bump-allocated.
    // Two arenas alternate between scans: the tables of a scan live until the next pav_inv_scan_batch, and that next scan
    // fills the other arena, so it never waits for copies of the scan before it that are still crossing PCIe.
    struct PinBlock { void *p; size_t cap, used; };
    std::vector<PinBlock> pinned_sets[2];
    int cur_set = 0;
    // KERN column of a state without k-mers: every table points at a shared block of zeros instead of receiving 8 B per row
    std::vector<std::unique_ptr<double[]>> zero_blocks;
    size_t zero_rows = 0;
    const double *zeros(size_t n) {
        if (n > zero_rows) {                                  // older blocks stay alive: earlier tables point into them
            zero_rows = std::max(n, 2 * zero_rows);
            zero_blocks.emplace_back(new double[zero_rows]());
        }
        return zero_blocks.back().get();
    }
    // Device side of the same tables: one packed block per scan round, in the set of the current scan.  With lazy tables
    // (pav_inv_params.lazy_tables) nothing crosses PCIe during the scan; the first reader of a table of this scan brings the
    // blocks over (materialise).
    std::vector<std::unique_ptr<CallStage>> stage_sets[2];
    size_t stage_used = 0;
    CallStage &next_stage() {
        auto &set = stage_sets[cur_set];
        if (stage_used == set.size()) set.emplace_back(new CallStage());
        return *set[stage_used++];
    }
    void next_pinned() { cur_set ^= 1; for (auto &b : pinned_sets[cur_set]) b.used = 0; stage_used = 0; }
    void free_pinned() {
        if (h_qa) { (void)hipHostFree(h_qa); h_qa = nullptr; h_qa_cap = 0; }
        for (DevBuf *b : {&d_ops, &d_begin, &d_small, &d_q, &d_a, &d_rin, &d_rq, &d_ra}) b->release();
        if (h_round) { (void)hipHostFree(h_round); h_round = nullptr; h_round_cap = 0; }
        for (auto &set : pinned_sets) { for (auto &b : set) (void)hipHostFree(b.p); set.clear(); }
        for (auto &set : stage_sets) { for (auto &st : set) st->release(); set.clear(); }
    }
    void *pin_alloc(size_t bytes) {
        auto &pinned = pinned_sets[cur_set];
        for (auto &b : pinned) if (b.cap - b.used >= bytes) { void *r = static_cast<uint8_t *>(b.p) + b.used; b.used += bytes; return r; }
        const size_t cap = std::max<size_t>(bytes + bytes / 4, (size_t)64 << 20);
        void *p = nullptr;
        if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) return nullptr;
        pinned.push_back(PinBlock{p, cap, bytes});
        return p;
    }
};

static InvState *istate(pav_ctx *ctx) {
    std::call_once(ctx->invscan_once, [ctx] { ctx->invscan = new InvState(); });   // (both roles' loaders name their records here first)
    return static_cast<InvState *>(ctx->invscan);
}

// ---- Python-compatible formatting --------------------------------------------------------------------------
static std::string fmt_i(int64_t v) { return std::to_string(v); }
static std::string fmt_commas(int64_t v) {                   // '{:,d}'
    std::string s = std::to_string(v < 0 ? -v : v), out;
    for (size_t i = 0; i < s.size(); ++i) { if (i && (s.size() - i) % 3 == 0) out += ','; out += s[i]; }
    return v < 0 ? "-" + out : out;
}
static std::string fmt_f2(double v) { char b[64]; snprintf(b, sizeof b, "%.2f", v); return b; }   // '{:.2f}'


struct Lifted { bool ok = false; int id = -1; int64_t pos = 0; int rev = 0; bool rev_none = false; int64_t idx[2] = {0, 0}; int n_idx = 0; };

class Driver {
public:
    Driver(pav_ctx *c, InvState *s) : ctx(c), S(s) {}
    pav_ctx *ctx; InvState *S;

    std::string name(int role, int id) const {
        if ((size_t)id < S->names[role].size() && !S->names[role][(size_t)id].empty()) return S->names[role][(size_t)id];
        return (role == PAV_ROLE_REF ? "ref" : "tig") + std::to_string(id);
    }
    std::string base1(const Rgn &r) const { return name(r.role, r.chrom) + ":" + fmt_i(r.pos + 1) + "-" + fmt_i(r.end); }   // to_base1_string
    std::string region_id(const Rgn &r) const { return name(r.role, r.chrom) + "-" + fmt_i(r.pos) + "-RGN-" + fmt_i(r.end - r.pos); }

    // Region.expand(expand_bp, min_pos=0, max_end=fai, shift=True, balance)   pavlib/seq.py:112-188
    void expand(Rgn &r, int64_t expand_bp, double balance) const {
        const int64_t expand_pos = (int64_t)((double)expand_bp * balance);          // int(expand_bp * balance)
        const int64_t expand_end = std::max<int64_t>(0, expand_bp - expand_pos);
        int64_t new_pos = r.pos - expand_pos, new_end = r.end + expand_end;
        const int64_t min_pos = 0;
        if (new_pos < min_pos) { new_end += min_pos - new_pos; new_pos = min_pos; }
        const int64_t max_end = (int64_t)ctx->seq[PAV_ROLE_REF].len[(size_t)r.chrom];
        if (new_end > max_end) {
            new_pos -= new_end - max_end;
            if (new_pos < min_pos) new_pos = min_pos;
            new_end = max_end;
        }
        if (new_end < new_pos) new_end = new_pos = (new_end + new_pos) / 2;       // over-contraction (negative expand)
        r.pos = new_pos; r.end = new_end;
    }

    // records of one sequence whose [begin, end) contains pos, by rising begin: the rows that begin at or before pos, walked from
    // the last one back until no earlier row reaches pos (a walk over every row that begins within the longest record of the
    // sequence before pos was a third of a microsecond per query)
    void containing(const std::vector<uint32_t> &recs, const InvState::Spans &sp, int64_t pos, std::vector<uint32_t> &hits) const {
        hits.clear();
        size_t i = (size_t)(std::upper_bound(sp.begin.begin(), sp.begin.end(), pos) - sp.begin.begin());
        while (i > 0 && sp.max_end[i - 1] > pos) {
            --i;
            if (sp.end[i] > pos) hits.push_back(recs[i]);
        }
        std::reverse(hits.begin(), hits.end());
    }

    // the begins of one record's operation table that bound `v` (see op_at): [lo, hi) holds the first begin above v, or hi is it
    void window(uint32_t row, int axis, uint32_t v, const uint32_t *&lo, const uint32_t *&hi) const {
        const uint64_t a = S->op_off[row], b = S->op_off[row + 1];
        const uint32_t *beg = (axis == 0 ? S->sub_begin.data() : S->qry_begin.data());
        const InvState::Buckets &B = S->buckets[axis];
        lo = hi = beg + a;
        if (a == b || v < B.base[row]) return;                               // nothing begins at or before v
        const uint64_t nb = B.off[row + 1] - B.off[row] - 1;                 // buckets of the record (one more entry closes the last)
        const uint64_t q = std::min<uint64_t>((uint64_t)(v - B.base[row]) >> B.shift[row], nb - 1);
        const uint32_t *e = B.first.data() + B.off[row] + q;
        lo = beg + a + e[0]; hi = beg + a + e[1];
    }
    // point lookup in one record's operation table; axis 0 = subject, 1 = query.  Returns op index or -1.
    int64_t op_at(uint32_t row, int axis, int64_t pos) const {
        const uint32_t *beg = (axis == 0 ? S->sub_begin.data() : S->qry_begin.data());
        const uint32_t *first = beg + S->op_off[row];
        if (pos < 0) return -1;
        // two levels: a position bucket bounds the search in the 36 MB array to a few lines - a plain binary search missed the
        // cache at most of its ~20 probes (0.3 -> 1.6 ms per scan round between a warm and a cold host cache), and so did the
        // search of a sampled array in front of it, at the last-level cache
        const uint32_t v = (uint32_t)std::min<int64_t>(pos, 0xFFFFFFFFll);
        const uint32_t *wlo, *whi;
        window(row, axis, v, wlo, whi);
        const uint32_t *it = std::upper_bound(wlo, whi, v);
        if (it == first) return -1;
        const uint64_t k = (uint64_t)(it - beg) - 1;
        const uint32_t code = S->ops[k] & 15u, len = S->ops[k] >> 4;
        const bool match = code == 7 || code == 8 || code == 0;
        if (!(match || code == (axis == 0 ? 2u : 1u))) return -1;
        return pos < (int64_t)beg[k] + (int64_t)len ? (int64_t)k : -1;
    }
    // (begin, end, d0, d1) of operation k on `axis` (pavlib/align/lift.py:437-461)
    void op_interval(uint64_t k, int axis, int64_t &begin, int64_t &end, int64_t &d0, int64_t &d1) const {
        const uint32_t code = S->ops[k] & 15u; const int64_t len = S->ops[k] >> 4;
        const bool match = code == 7 || code == 8 || code == 0;
        begin = axis == 0 ? S->sub_begin[k] : S->qry_begin[k];
        end = begin + len;
        d0 = axis == 0 ? S->qry_begin[k] : S->sub_begin[k];
        d1 = match ? d0 + len : d0 + 1;
    }
    bool check_row_ops(uint32_t row, std::string &err) const {       // errors _add_align raises (lift.py:463-471), on every use
        const uint32_t code = S->row_bad[row];
        if (!code) return true;
        err = std::string("Unhandled CIGAR operation: ") + (code == 3 ? "N" : "P") + ": Alignment " +
              name(PAV_ROLE_REF, (int)S->rows[row].ref_id) + ":" + fmt_i(S->rows[row].pos) + " (" + name(PAV_ROLE_TIG, (int)S->rows[row].tig_id) + ")";
        return false;
    }

    // AlignLift.lift_to_qry (lift.py:187-272).  false + err on RuntimeError.
    bool lift_to_qry(int ref_id, int64_t pos, Lifted &out, std::string &err) const {
        out = Lifted();
        std::vector<uint32_t> hits;
        if ((size_t)ref_id < S->by_ref.size()) containing(S->by_ref[(size_t)ref_id], S->span_ref[(size_t)ref_id], pos, hits);
        if (hits.size() != 1) return true;
        const uint32_t row = hits[0];
        if (!check_row_ops(row, err)) return false;
        const int64_t k = op_at(row, 0, pos);
        if (k < 0) {
            err = "Program bug: Found no matches in a lift-tree for a record withing a global to-query tree: " + name(PAV_ROLE_REF, ref_id) + ":" +
                  fmt_i(pos) + " (index=" + fmt_i(S->rows[row].index) + ")";
            return false;
        }
        int64_t b, e, d0, d1;
        op_interval((uint64_t)k, 0, b, e, d0, d1);
        int64_t q = d1 - d0 > 1 ? d0 + (pos - b) : d1;
        const LiftRow &r = S->rows[row];
        if (r.rev) q = (int64_t)ctx->seq[PAV_ROLE_TIG].len[r.tig_id] - q;
        out.ok = true; out.id = (int)r.tig_id; out.pos = q; out.rev = r.rev; out.idx[0] = r.index; out.n_idx = 1;
        return true;
    }

    // AlignLift._get_subject_gap (lift.py:333-378)
    void subject_gap(int tig_id, int64_t pos, Lifted &out) const {
        out = Lifted();
        if ((size_t)tig_id >= S->by_tig.size()) return;
        const auto &recs = S->by_tig[(size_t)tig_id];
        int64_t best_l = -1, best_r = -1; int64_t lv = 0, rv = 0;
        // row_l: largest QRY_END < pos (ties: last in table order after a stable sort); row_r: smallest QRY_POS > pos (first)
        std::vector<uint32_t> order(recs); std::sort(order.begin(), order.end());      // table order
        for (uint32_t r : order) {
            const LiftRow &x = S->rows[r];
            if (x.qry_end < pos && (best_l < 0 || x.qry_end >= lv)) { best_l = r; lv = x.qry_end; }
            if (x.qry_pos > pos && (best_r < 0 || x.qry_pos < rv)) { best_r = r; rv = x.qry_pos; }
        }
        if (best_l < 0 || best_r < 0) return;
        const LiftRow &L = S->rows[(size_t)best_l], &R = S->rows[(size_t)best_r];
        if (L.ref_id != R.ref_id) return;
        out.ok = true; out.id = (int)L.ref_id;
        out.pos = (int64_t)((double)(L.qry_end + R.qry_pos) / 2.0);                    // int((a + b) / 2)
        out.rev = L.rev; out.rev_none = L.rev != R.rev;
        out.idx[0] = L.index; out.idx[1] = R.index; out.n_idx = 2;
    }

    // AlignLift.lift_to_sub (lift.py:51-185)
    bool lift_to_sub(int tig_id, int64_t pos, bool gap, Lifted &out, std::string &err) const {
        out = Lifted();
        const int64_t pos_org = pos;
        std::vector<uint32_t> hits;
        if ((size_t)tig_id < S->by_tig.size()) containing(S->by_tig[(size_t)tig_id], S->span_tig[(size_t)tig_id], pos, hits);
        if (hits.size() == 0 && gap) { subject_gap(tig_id, pos, out); return true; }
        if (hits.size() != 1) return true;
        const uint32_t row = hits[0];
        if (!check_row_ops(row, err)) return false;
        const LiftRow &r = S->rows[row];
        if (r.rev) pos = (int64_t)ctx->seq[PAV_ROLE_TIG].len[r.tig_id] - pos;
        int64_t k = op_at(row, 1, pos);
        int64_t b, e, d0, d1;
        if (k < 0) {
            k = op_at(row, 1, pos - 1);
            if (k >= 0) op_interval((uint64_t)k, 1, b, e, d0, d1);
            if (k < 0 || e != pos) {
                err = "Found no matches in a lift-tree for a record within a global to-subject tree: " + name(PAV_ROLE_TIG, tig_id) + ":" +
                      fmt_i(pos_org) + " (index=" + fmt_i(r.index) + ", gap=" + (gap ? "True" : "False") + ")";
                return false;
            }
        }
        op_interval((uint64_t)k, 1, b, e, d0, d1);
        const int64_t p = d1 - d0 > 1 ? d0 + (pos - b) : d1;
        out.ok = true; out.id = (int)r.ref_id; out.pos = p; out.rev = r.rev; out.idx[0] = r.index; out.n_idx = 1;
        return true;
    }

    // ---- point lifts in batches: the device index (lift_dev.hip) or, with PAV_LIFT_HOST=1, the host functions above ----------
    struct Point { LiftAnswer a; std::string err; };                   // a.status >= LIFT_ERR_OP: `err` is the RuntimeError text
    std::string lift_error(const LiftQuery &q, const LiftAnswer &a) const {
        const LiftRow &r = S->rows[a.row];
        if (a.status == LIFT_ERR_OP) {
            uint32_t code = 3;
            if (!S->on_device) code = S->row_bad[a.row];
            else code = (uint32_t)a.id;                                // (the kernel leaves the record's flag in `id` for this status)
            return std::string("Unhandled CIGAR operation: ") + (code == 3 ? "N" : "P") + ": Alignment " +
                   name(PAV_ROLE_REF, (int)r.ref_id) + ":" + fmt_i(r.pos) + " (" + name(PAV_ROLE_TIG, (int)r.tig_id) + ")";
        }
        if (a.status == LIFT_ERR_NO_MATCH_QRY)
            return "Program bug: Found no matches in a lift-tree for a record withing a global to-query tree: " + name(PAV_ROLE_REF, q.seq) + ":" +
                   fmt_i(q.pos) + " (index=" + fmt_i(r.index) + ")";
        return "Found no matches in a lift-tree for a record within a global to-subject tree: " + name(PAV_ROLE_TIG, q.seq) + ":" +
               fmt_i(q.pos) + " (index=" + fmt_i(r.index) + ", gap=" + (q.gap ? "True" : "False") + ")";
    }
    int lift_batch(const std::vector<LiftQuery> &q, std::vector<Point> &out) const {
        out.assign(q.size(), Point{});
        if (q.empty()) return PAV_OK;
        if (!S->on_device) {
            for (size_t i = 0; i < q.size(); ++i) {
                Lifted l; std::string err;
                const bool fine = q[i].axis == 0 ? lift_to_qry(q[i].seq, q[i].pos, l, err) : lift_to_sub(q[i].seq, q[i].pos, q[i].gap != 0, l, err);
                LiftAnswer &a = out[i].a;
                if (!fine) { a.status = LIFT_ERR_OP; out[i].err = err; continue; }
                a.status = l.ok ? LIFT_OK : LIFT_NONE; a.id = l.id; a.pos = l.pos; a.rev = l.rev; a.rev_none = l.rev_none ? 1 : 0;
                a.n_idx = l.n_idx; a.idx[0] = l.idx[0]; a.idx[1] = l.idx[1];
            }
            return PAV_OK;
        }
        const size_t nq = q.size(), bytes_q = sizeof(LiftQuery) * nq, bytes_a = sizeof(LiftAnswer) * nq;
        if (bytes_q + bytes_a > S->h_qa_cap) {
            if (S->h_qa) (void)hipHostFree(S->h_qa);
            S->h_qa = nullptr; S->h_qa_cap = 0;
            const size_t cap = (bytes_q + bytes_a) * 2 + 4096;
            PAV_HIP(ctx, hipHostMalloc(&S->h_qa, cap, hipHostMallocDefault));
            S->h_qa_cap = cap;
        }
        PAV_HIP(ctx, S->d_q.reserve(bytes_q));
        PAV_HIP(ctx, S->d_a.reserve(bytes_a));
        uint8_t *h = static_cast<uint8_t *>(S->h_qa);
        memcpy(h, q.data(), bytes_q);
        hipStream_t st = ctx->stream;
        PAV_HIP(ctx, hipMemcpyAsync(S->d_q.p, h, bytes_q, hipMemcpyHostToDevice, st));
        const int rc = lift_points(ctx, S->dev, S->d_q.as<LiftQuery>(), S->d_a.as<LiftAnswer>(), (uint32_t)nq);
        if (rc != PAV_OK) return rc;
        PAV_HIP(ctx, hipMemcpyAsync(h + bytes_q, S->d_a.p, bytes_a, hipMemcpyDeviceToHost, st));
        PAV_HIP(ctx, hipStreamSynchronize(st));
        const LiftAnswer *a = reinterpret_cast<const LiftAnswer *>(h + bytes_q);
        for (size_t i = 0; i < nq; ++i) {
            out[i].a = a[i];
            if (a[i].status >= LIFT_ERR_OP) out[i].err = lift_error(q[i], a[i]);
        }
        return PAV_OK;
    }
    // lift_region_to_qry (lift.py:304-331) from the answers for region.pos and region.end; false + err on RuntimeError
    bool region_from_qry_points(const Point &pa, const Point &pb, Rgn &q, bool &ok, std::string &err) const {
        ok = false;
        if (pa.a.status >= LIFT_ERR_OP) { err = pa.err; return false; }
        if (pb.a.status >= LIFT_ERR_OP) { err = pb.err; return false; }
        const LiftAnswer &a = pa.a, &b = pb.a;
        if (a.status != LIFT_OK || b.status != LIFT_OK || a.id != b.id || a.rev != b.rev) return true;
        q = Rgn(); q.role = PAV_ROLE_TIG; q.chrom = a.id; q.pos = a.pos; q.end = b.pos; q.is_rev = a.rev != 0;
        q.aln[0][0] = a.idx[0]; q.aln[1][0] = b.idx[0]; q.n_aln[0] = q.n_aln[1] = 1;
        if (q.pos > q.end) { std::swap(q.pos, q.end); std::swap(q.aln[0], q.aln[1]); }   // Region swaps reversed coordinates
        ok = true;
        return true;
    }
    // lift_region_to_sub (lift.py:274-302) from the two answers
    bool region_from_sub_points(const Point &pa, const Point &pb, Rgn &s, bool &ok, std::string &err) const {
        ok = false;
        if (pa.a.status >= LIFT_ERR_OP) { err = pa.err; return false; }
        if (pb.a.status >= LIFT_ERR_OP) { err = pb.err; return false; }
        const LiftAnswer &a = pa.a, &b = pb.a;
        if (a.status != LIFT_OK || b.status != LIFT_OK) return true;
        if (a.id != b.id || (!a.rev_none && !b.rev_none && a.rev != b.rev)) return true;
        s = Rgn(); s.role = PAV_ROLE_REF; s.chrom = a.id; s.pos = a.pos; s.end = b.pos; s.is_rev = false;
        for (int i = 0; i < a.n_idx; ++i) s.aln[0][i] = a.idx[i];
        for (int i = 0; i < b.n_idx; ++i) s.aln[1][i] = b.idx[i];
        s.n_aln[0] = a.n_idx; s.n_aln[1] = b.n_idx;
        if (s.pos > s.end) { std::swap(s.pos, s.end); std::swap(s.aln[0], s.aln[1]); std::swap(s.n_aln[0], s.n_aln[1]); }
        ok = true;
        return true;
    }

    // lift_region_to_qry (lift.py:304-331): ok=false => None
    bool region_to_qry(const Rgn &r, Rgn &q, bool &ok, std::string &err) const {
        Lifted a, b; ok = false;
        if (!lift_to_qry(r.chrom, r.pos, a, err) || !lift_to_qry(r.chrom, r.end, b, err)) return false;
        if (!a.ok || !b.ok || a.id != b.id || a.rev != b.rev) return true;
        q = Rgn(); q.role = PAV_ROLE_TIG; q.chrom = a.id; q.pos = a.pos; q.end = b.pos; q.is_rev = a.rev != 0;
        q.aln[0][0] = a.idx[0]; q.aln[1][0] = b.idx[0]; q.n_aln[0] = q.n_aln[1] = 1;
        if (q.pos > q.end) { std::swap(q.pos, q.end); std::swap(q.aln[0], q.aln[1]); }   // Region swaps reversed coordinates
        ok = true;
        return true;
    }
    // lift_region_to_sub (lift.py:274-302)
    bool region_to_sub(const Rgn &r, bool gap, Rgn &s, bool &ok, std::string &err) const {
        Lifted a, b; ok = false;
        if (!lift_to_sub(r.chrom, r.pos, gap, a, err) || !lift_to_sub(r.chrom, r.end, gap, b, err)) return false;
        if (!a.ok || !b.ok) return true;
        if (a.id != b.id || (!a.rev_none && !b.rev_none && a.rev != b.rev)) return true;
        s = Rgn(); s.role = PAV_ROLE_REF; s.chrom = a.id; s.pos = a.pos; s.end = b.pos; s.is_rev = false;
        for (int i = 0; i < a.n_idx; ++i) s.aln[0][i] = a.idx[i];
        for (int i = 0; i < b.n_idx; ++i) s.aln[1][i] = b.idx[i];
        s.n_aln[0] = a.n_idx; s.n_aln[1] = b.n_idx;
        if (s.pos > s.end) { std::swap(s.pos, s.end); std::swap(s.aln[0], s.aln[1]); std::swap(s.n_aln[0], s.n_aln[1]); }
        ok = true;
        return true;
    }
};

const std::vector<std::string> &seq_names(pav_ctx *ctx, int role) { return istate(ctx)->names[role]; }
int pav_seq_set_names_internal(pav_ctx *ctx, int role, const std::vector<std::string> &names) { istate(ctx)->names[role] = names; return PAV_OK; }

// Give a round its pinned host block (whole-round columns: K0 | K1 | K2 | KMER | INDEX | STATE_MER | STATE | FLANK | MATCH, each
// holding the calls one after the other) and point the tables of its calls into it.
static int bind_round(pav_ctx *ctx, InvState *S, CallStage &stg) {
    if (stg.host || stg.rows == 0) return PAV_OK;
    const size_t rows = stg.rows;
    const size_t total = (rows * 40 + 64 + 63) / 64 * 64;             // blocks stay 64-byte aligned
    void *blk = S->pin_alloc(total);
    if (!blk) return fail(ctx, PAV_E_HIP, "pav_inv_scan_batch: cannot pin %zu bytes of host memory", total);
    stg.host = blk;
    double *c_k0 = static_cast<double *>(blk), *c_k1 = c_k0 + rows, *c_k2 = c_k1 + rows;
    uint64_t *c_kmer = reinterpret_cast<uint64_t *>(c_k2 + rows);
    uint32_t *c_index = reinterpret_cast<uint32_t *>(c_kmer + rows);
    int8_t *c_sm = reinterpret_cast<int8_t *>(c_index + rows), *c_st = c_sm + rows;
    uint8_t *c_fl = reinterpret_cast<uint8_t *>(c_st + rows), *c_ma = c_fl + rows;
    for (const CallStage::Entry &e : stg.entries) {
        InvTable *tab = S->tables[e.owner].get();
        if (!tab) continue;
        const size_t o = e.row0;
        tab->kern[0] = c_k0 + o; tab->kern[1] = c_k1 + o; tab->kern[2] = c_k2 + o; tab->kmer = c_kmer + o; tab->index = c_index + o;
        tab->state_mer = c_sm + o; tab->state = c_st + o; tab->flank = c_fl + o; tab->match = c_ma + o;
        // a call without FWDREV k-mers has a KERN_FWDREV column of zeros (scripts/density.py:313-323): it is not copied, the
        // table reads a shared block of zeros (read-only for every consumer)
        if (!e.has_k1) tab->kern[1] = const_cast<double *>(S->zeros(e.n));
    }
    return PAV_OK;
}

// Host copies of the call tables of the last scan: bind and queue whatever a lazy scan left in HBM, then wait for the copy stream.
static int tables_on_host(pav_ctx *ctx, InvState *S) {
    bool any = false;
    for (size_t r = 0; r < S->stage_used; ++r) any = any || S->stage_sets[S->cur_set][r]->n_copies > 0;
    if (any) {
        PAV_HIP(ctx, hipSetDevice(ctx->device));
        PAV_HIP(ctx, hipStreamSynchronize(ctx->stream));              // the gather kernels of the scan
        for (size_t r = 0; r < S->stage_used; ++r) {
            CallStage &stg = *S->stage_sets[S->cur_set][r];
            int rc = bind_round(ctx, S, stg);
            if (rc == PAV_OK) rc = stage_copy(ctx, stg);
            if (rc != PAV_OK) return rc;
        }
    }
    return wait_tables(ctx);
}


}  // namespace pav

using namespace pav;

extern "C" {

void pav_invscan_release(pav_ctx *ctx) {
    if (!ctx || !ctx->invscan) return;
    static_cast<InvState *>(ctx->invscan)->free_pinned();
    delete static_cast<InvState *>(ctx->invscan);
    ctx->invscan = nullptr;
}

// name of record i of `role` as set by pav_seq_set_names / a FASTA loader (NULL: no such record or no names)
const char *pav_seq_name(const pav_ctx *ctx, int role, uint32_t i) {
    if (!ctx || !ctx->invscan || (role != PAV_ROLE_REF && role != PAV_ROLE_TIG)) return nullptr;
    const InvState *S = static_cast<const InvState *>(ctx->invscan);
    return i < S->names[role].size() ? S->names[role][i].c_str() : nullptr;
}

int pav_seq_set_names(pav_ctx *ctx, int role, uint32_t n, const char *const *names) {
    if (!ctx || (role != PAV_ROLE_REF && role != PAV_ROLE_TIG) || (n && !names)) return PAV_E_ARG;
    InvState *S = istate(ctx);
    S->names[role].assign(n, std::string());
    for (uint32_t i = 0; i < n; ++i) S->names[role][i] = names[i] ? names[i] : "";
    return PAV_OK;
}

int pav_inv_load_alignments(pav_ctx *ctx, uint32_t n, const pav_inv_aln *aln, const uint8_t *cigar_text, const uint64_t *cigar_off) {
    if (!ctx || (n && (!aln || !cigar_off))) return fail(ctx, PAV_E_ARG, "pav_inv_load_alignments: null input");
    InvState *S = istate(ctx);
    S->loaded = false;
    S->rows.resize(n);
    std::vector<uint32_t> row_pos(n);
    std::map<int64_t, int> seen;
    for (uint32_t i = 0; i < n; ++i) {
        const pav_inv_aln &a = aln[i];
        if (a.ref_id >= ctx->seq[PAV_ROLE_REF].n || a.tig_id >= ctx->seq[PAV_ROLE_TIG].n)
            return fail(ctx, PAV_E_ARG, "pav_inv_load_alignments: row %u references a sequence that is not loaded", i);
        S->rows[i] = LiftRow{a.ref_id, a.tig_id, (int64_t)a.pos, (int64_t)a.end, (int64_t)a.qry_pos, (int64_t)a.qry_end, a.rev != 0, a.index};
        row_pos[i] = (uint32_t)a.pos;
    }
    uint64_t n_ops = 0;
    int rc = pav_align_index(ctx, n, row_pos.data(), cigar_text, cigar_off, &n_ops, nullptr, nullptr, nullptr, nullptr);
    if (rc != PAV_OK) return rc;
    const char *e_host = getenv("PAV_LIFT_HOST");
    const bool host_tables = e_host && e_host[0] == '1';
    S->on_device = !host_tables;
    S->op_off.resize((size_t)n + 1);
    if (S->on_device) {
        // the tables stay in HBM: the context's index buffers are scratch (the next pav_align_index overwrites them), so they are
        // copied device to device (109 MB for a haplotype: 0.05 ms); only the per-record operation ranges come to the host
        hipStream_t st = ctx->stream;
        PAV_HIP(ctx, hipSetDevice(ctx->device));
        PAV_HIP(ctx, S->d_ops.reserve(4 * (n_ops + 16)));
        PAV_HIP(ctx, S->d_begin.reserve(8 * (n_ops + 16)));
        if (n_ops) {
            PAV_HIP(ctx, hipMemcpyAsync(S->d_ops.p, ctx->ix_ops.p, 4 * n_ops, hipMemcpyDeviceToDevice, st));
            PAV_HIP(ctx, hipMemcpyAsync(S->d_begin.p, ctx->ix_begin.p, 8 * n_ops, hipMemcpyDeviceToDevice, st));
        }
        if (n) PAV_HIP(ctx, hipMemcpyAsync(S->op_off.data(), ctx->ix_op_off.p, sizeof(uint64_t) * ((size_t)n + 1), hipMemcpyDeviceToHost, st));
        PAV_HIP(ctx, hipStreamSynchronize(st));
        if (n == 0) S->op_off.assign(1, 0);
        S->ops.resize(0); S->sub_begin.resize(0); S->qry_begin.resize(0);
    } else {
    S->ops.resize(n_ops + 1); S->sub_begin.resize(n_ops + 1); S->qry_begin.resize(n_ops + 1);
    rc = pav_align_index(ctx, n, row_pos.data(), cigar_text, cigar_off, &n_ops, S->ops.data(), S->op_off.data(), S->sub_begin.data(), S->qry_begin.data());
    if (rc != PAV_OK) return rc;
    if (n == 0) S->op_off.assign(1, 0);
    for (int axis = 0; axis < 2; ++axis) {
        const HugeArray<uint32_t> &src = axis == 0 ? S->sub_begin : S->qry_begin;
        InvState::Buckets &B = S->buckets[axis];
        B.first.clear(); B.off.assign((size_t)n + 1, 0); B.shift.assign(n, 0); B.base.assign(n, 0);
        for (uint32_t r = 0; r < n; ++r) {
            const uint64_t a = S->op_off[r], b = S->op_off[r + 1];
            B.off[r] = B.first.size();
            if (a == b) continue;
            const uint32_t base = src[a];
            const uint64_t span = (uint64_t)src[b - 1] - base + 1, want = (b - a + 15) / 16;
            unsigned sh = 0;
            while (((span - 1) >> sh) + 1 > want) ++sh;
            const uint64_t nb = ((span - 1) >> sh) + 1;
            B.shift[r] = (uint8_t)sh; B.base[r] = base;
            uint64_t k = a;
            for (uint64_t q = 0; q < nb; ++q) {                           // operations that begin before the bucket's first base
                while (k < b && (uint64_t)(src[k] - base) < (q << sh)) ++k;
                B.first.push_back((uint32_t)(k - a));
            }
            B.first.push_back((uint32_t)(b - a));
        }
        B.off[n] = B.first.size();
    }
    }
    S->by_ref.assign(ctx->seq[PAV_ROLE_REF].n, {}); S->by_tig.assign(ctx->seq[PAV_ROLE_TIG].n, {});
    S->by_ref_maxlen.assign(ctx->seq[PAV_ROLE_REF].n, 0); S->by_tig_maxlen.assign(ctx->seq[PAV_ROLE_TIG].n, 0);
    for (uint32_t i = 0; i < n; ++i) {
        const LiftRow &r = S->rows[i];
        if (r.end > r.pos) { S->by_ref[r.ref_id].push_back(i); S->by_ref_maxlen[r.ref_id] = std::max(S->by_ref_maxlen[r.ref_id], r.end - r.pos); }
        if (r.qry_end > r.qry_pos) { S->by_tig[r.tig_id].push_back(i); S->by_tig_maxlen[r.tig_id] = std::max(S->by_tig_maxlen[r.tig_id], r.qry_end - r.qry_pos); }
    }
    for (auto &v : S->by_ref) std::stable_sort(v.begin(), v.end(), [&](uint32_t a, uint32_t b) { return S->rows[a].pos < S->rows[b].pos; });
    for (auto &v : S->by_tig) std::stable_sort(v.begin(), v.end(), [&](uint32_t a, uint32_t b) { return S->rows[a].qry_pos < S->rows[b].qry_pos; });
    for (int axis = 0; axis < 2; ++axis) {
        const auto &by = axis == 0 ? S->by_ref : S->by_tig;
        auto &spans = axis == 0 ? S->span_ref : S->span_tig;
        spans.assign(by.size(), {});
        for (size_t q = 0; q < by.size(); ++q) {
            InvState::Spans &sp = spans[q];
            int64_t run = INT64_MIN;
            for (uint32_t r : by[q]) {
                const LiftRow &x = S->rows[r];
                sp.begin.push_back(axis == 0 ? x.pos : x.qry_pos);
                sp.end.push_back(axis == 0 ? x.end : x.qry_end);
                run = std::max(run, sp.end.back());
                sp.max_end.push_back(run);
            }
        }
    }
    S->row_bad.assign(n, 0);
    if (!S->on_device) {
        for (uint32_t r = 0; r < n; ++r)
            for (uint64_t q = S->op_off[r]; q < S->op_off[r + 1]; ++q) {
                const uint32_t code = S->ops[q] & 15u;
                if (code == 3 || code == 6) { S->row_bad[r] = (uint8_t)code; break; }
            }
    } else {
        // the small tables of the device index in one block: records | per axis: sequence offsets, records by begin, begin, end,
        // running maximum of end | contig records in table order
        const uint32_t n_seq[2] = {ctx->seq[PAV_ROLE_REF].n, ctx->seq[PAV_ROLE_TIG].n};
        std::vector<LiftRowDev> rows(n);
        for (uint32_t i = 0; i < n; ++i) {
            const LiftRow &r = S->rows[i];
            rows[i] = LiftRowDev{r.ref_id, r.tig_id, r.pos, r.end, r.qry_pos, r.qry_end, r.index, S->op_off[i], S->op_off[i + 1],
                                 ctx->seq[PAV_ROLE_TIG].len[r.tig_id], r.rev, 0u};
        }
        auto pad = [](size_t x) { return (x + 63) / 64 * 64; };
        size_t at = 0, o_rows = at; at += pad(sizeof(LiftRowDev) * n);
        size_t o_off[2], o_rw[2], o_b[2], o_e[2], o_m[2];
        size_t cnt[2] = {0, 0};
        for (int axis = 0; axis < 2; ++axis) { const auto &by = axis == 0 ? S->by_ref : S->by_tig; for (const auto &v : by) cnt[axis] += v.size(); }
        for (int axis = 0; axis < 2; ++axis) {
            o_off[axis] = at; at += pad(4 * ((size_t)n_seq[axis] + 1));
            o_rw[axis] = at; at += pad(4 * cnt[axis]);
            o_b[axis] = at; at += pad(8 * cnt[axis]); o_e[axis] = at; at += pad(8 * cnt[axis]); o_m[axis] = at; at += pad(8 * cnt[axis]);
        }
        const size_t o_tab = at; at += pad(4 * cnt[1]);
        std::vector<uint8_t> blk(at + 64, 0);
        if (n) memcpy(blk.data() + o_rows, rows.data(), sizeof(LiftRowDev) * n);
        for (int axis = 0; axis < 2; ++axis) {
            const auto &by = axis == 0 ? S->by_ref : S->by_tig;
            const auto &spans = axis == 0 ? S->span_ref : S->span_tig;
            uint32_t *off = reinterpret_cast<uint32_t *>(blk.data() + o_off[axis]), *rw = reinterpret_cast<uint32_t *>(blk.data() + o_rw[axis]);
            int64_t *b = reinterpret_cast<int64_t *>(blk.data() + o_b[axis]), *e = reinterpret_cast<int64_t *>(blk.data() + o_e[axis]),
                    *m = reinterpret_cast<int64_t *>(blk.data() + o_m[axis]);
            uint32_t run = 0;
            for (size_t q = 0; q < by.size(); ++q) {
                off[q] = run;
                for (size_t t = 0; t < by[q].size(); ++t) { rw[run + t] = by[q][t]; b[run + t] = spans[q].begin[t]; e[run + t] = spans[q].end[t]; m[run + t] = spans[q].max_end[t]; }
                if (axis == 1) {
                    uint32_t *tab = reinterpret_cast<uint32_t *>(blk.data() + o_tab) + run;
                    std::copy(by[q].begin(), by[q].end(), tab);
                    std::sort(tab, tab + by[q].size());                   // table order
                }
                run += (uint32_t)by[q].size();
            }
            off[by.size()] = run;
        }
        PAV_HIP(ctx, hipSetDevice(ctx->device));
        PAV_HIP(ctx, S->d_small.reserve(blk.size()));
        PAV_HIP(ctx, hipMemcpyAsync(S->d_small.p, blk.data(), blk.size(), hipMemcpyHostToDevice, ctx->stream));
        uint8_t *d = S->d_small.as<uint8_t>();
        LiftTables &T = S->dev;
        T.ops = S->d_ops.as<uint32_t>(); T.sub = S->d_begin.as<uint32_t>(); T.qry = S->d_begin.as<uint32_t>() + n_ops;
        T.rows = reinterpret_cast<LiftRowDev *>(d + o_rows); T.n_rows = n;
        for (int axis = 0; axis < 2; ++axis) {
            T.seq_off[axis] = reinterpret_cast<const uint32_t *>(d + o_off[axis]); T.seq_rows[axis] = reinterpret_cast<const uint32_t *>(d + o_rw[axis]);
            T.begin[axis] = reinterpret_cast<const int64_t *>(d + o_b[axis]); T.end[axis] = reinterpret_cast<const int64_t *>(d + o_e[axis]);
            T.max_end[axis] = reinterpret_cast<const int64_t *>(d + o_m[axis]);
            T.n_seq[axis] = n_seq[axis];
        }
        T.tig_table = reinterpret_cast<const uint32_t *>(d + o_tab);
        rc = lift_row_flags(ctx, T, n_ops);
        if (rc != PAV_OK) return rc;
        PAV_HIP(ctx, hipStreamSynchronize(ctx->stream));                     // `blk` is a local buffer
    }
    if (!S->pool) {
        // PAV_HOST_THREADS = n: n - 1 helper threads next to the caller's.  Default none: on the bench box (a CPU quota shared
        // with the runtime's own threads) three helpers made a one-lane pass slower, 3.2 -> 3.5 ms - their spinning between
        // the loops of a round costs more than the loops (0.1 - 0.2 ms each) gain
        const char *e = getenv("PAV_HOST_THREADS");
        const int helpers = e ? std::max(0, atoi(e) - 1) : 0;
        S->pool = std::make_unique<HostPool>(helpers);
    }
    S->loaded = true;
    return PAV_OK;
}

int pav_inv_scan_batch(pav_ctx *ctx, uint32_t n_regions, const pav_inv_region *regions, const pav_inv_params *pp,
                       pav_inv_result *results) {
    if (!ctx || !pp || (n_regions && (!regions || !results))) return fail(ctx, PAV_E_ARG, "pav_inv_scan_batch: null argument");
    InvState *S = istate(ctx);
    if (!S->loaded) return fail(ctx, PAV_E_STATE, "pav_inv_scan_batch: pav_inv_load_alignments has not been called");
    PAV_HIP(ctx, hipSetDevice(ctx->device));                           // the first HIP work below is the lift batch's own
    const double t_entry = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    S->pool->wake();                                                     // the first per-region loop is a few microseconds away
    Driver D(ctx, S);
    auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tl[6] = {0};
    const int k = pp->den.k;
    const int64_t max_region_size = pp->max_region_size;
    const int min_exp_count = pp->min_exp_count;
    S->results.assign(n_regions, pav_inv_result{});
    // (texts and per-region state are emptied, not rebuilt: a thousand regions append five lines each, and a string that grows
    // from nothing is reallocated five times on the way)
    for (std::vector<std::string> *v : {&S->logs, &S->errors, &S->found}) { v->resize(n_regions); for (std::string &x : *v) x.clear(); }
    tl[0] = tnow();
    // this scan fills the arena of the scan before the last one: only *its* copies must have landed (they have, long ago);
    // the copies of the last scan may still be travelling into the other arena
    std::swap(ctx->tables_done, ctx->tables_done_prev);
    std::swap(ctx->tables_pending, ctx->tables_pending_prev);
    { const int rcw = wait_tables(ctx); if (rcw != PAV_OK) return rcw; }
    tl[1] = tnow();
    S->tables.clear(); S->tables.resize(n_regions);
    S->next_pinned();
    tl[2] = tnow();
    std::vector<Scan> &scans = S->scans;
    scans.resize(n_regions);
    for (Scan &sc : scans) { sc.flag = sc.region_ref = sc.region_tig = Rgn(); sc.expansion_count = 0; sc.done = false; sc.state_rl.clear(); sc.n_rows = 0; }
    tl[3] = tnow();
    auto log = [&](uint32_t i, const std::string &m) { S->logs[i] += m; S->logs[i] += '\n'; };
    auto srs_of = [&](int64_t len) -> uint32_t {
        for (uint32_t i = 0; i < pp->n_srs; ++i) if ((double)len >= pp->srs[i].begin && (double)len < pp->srs[i].end) return pp->srs[i].value;
        return 20;
    };
    auto finish = [&](uint32_t i, int outcome) { scans[i].done = true; S->results[i].outcome = outcome; };
    auto set_rgn = [&](pav_inv_rgn &o, const Rgn &r) {
        o.seq_id = (uint32_t)r.chrom; o.pos = (uint64_t)r.pos; o.end = (uint64_t)r.end; o.is_rev = r.is_rev ? 1 : 0;
        for (int e = 0; e < 2; ++e) { o.n_aln[e] = (uint32_t)r.n_aln[e]; for (int q = 0; q < 2; ++q) o.aln_index[e][q] = r.aln[e][q]; }
    };

    // The loops over regions below do per-region work only (lifts through the alignment table, log lines): they run on the
    // caller's thread and the helpers of S->pool, and whatever depends on the order of the regions follows in a serial pass.
    HostPool &pool = *S->pool;
    constexpr size_t CHUNK = 32;
    std::vector<uint32_t> live;
    for (uint32_t i = 0; i < n_regions; ++i)
        if (regions[i].ref_id >= ctx->seq[PAV_ROLE_REF].n) return fail(ctx, PAV_E_ARG, "pav_inv_scan_batch: region %u: unknown reference record", i);
    pool.run(n_regions, CHUNK, [&](size_t ii) {
        const uint32_t i = (uint32_t)ii;
        Scan &sc = scans[i];
        sc.flag.role = PAV_ROLE_REF; sc.flag.chrom = (int)regions[i].ref_id; sc.flag.pos = (int64_t)regions[i].pos; sc.flag.end = (int64_t)regions[i].end;
        if (sc.flag.pos > sc.flag.end) { std::swap(sc.flag.pos, sc.flag.end); sc.flag.is_rev = true; }
        sc.region_ref = sc.flag;
        D.expand(sc.region_ref, 4000, 0.5);                                          // INITIAL_EXPAND, inv.py:203-204
    });
    for (uint32_t i = 0; i < n_regions; ++i) live.push_back(i);
    tl[4] = tnow();

    // bases (reference + contig) of the regions one density batch takes; the regions behind the budget wait for the next batch
    // of the same round.  PAV_SCAN_BATCH_BP: tests shrink it to run that path on small cases.
    const uint64_t max_batch_bp = [] { const char *e = getenv("PAV_SCAN_BATCH_BP"); const long long v = e ? atoll(e) : 0; return v > 0 ? (uint64_t)v : 64000000ull; }();
    const bool timing = getenv("PAV_TIMING") != nullptr;
    double t_batch = 0, t_table = 0, t_pre = 0, t_post = 0, t_lift = 0;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_start = now();
    // point lifts of the regions of the NEXT round, asked for together with the breakpoint lifts of this one
    std::vector<Driver::Point> carried; std::vector<uint32_t> carried_first; bool carried_valid = false;
    auto queue_region_lifts = [&](const std::vector<uint32_t> &rgns, std::vector<LiftQuery> &lq, std::vector<uint32_t> &first) {
        first.assign(rgns.size(), ~0u);
        lq.reserve(lq.size() + 2 * rgns.size());
        for (size_t q = 0; q < rgns.size(); ++q) {
            const Scan &sc = scans[rgns[q]];
            if (0 < max_region_size && max_region_size < sc.region_ref.len()) continue;      // too large: no lift (reported in the round)
            first[q] = (uint32_t)lq.size();
            lq.push_back(LiftQuery{0, sc.region_ref.chrom, 0, 0, sc.region_ref.pos});
            lq.push_back(LiftQuery{0, sc.region_ref.chrom, 0, 0, sc.region_ref.end});
        }
    };
    // "Found no inverted k-mer states ..." of the regions a round has settled (a thousand of them in the first round): a region that
    // ends gets no further line, so its last one is written while the next round's kernels run, or behind the last round
    std::vector<std::pair<uint32_t, int>> notes;                          // (region, expansions)
    auto flush_notes = [&]() {
        pool.run(notes.size(), CHUNK, [&](size_t q) { log(notes[q].first, "Found no inverted k-mer states after " + fmt_i(notes[q].second) + " expansion(s)"); });
        notes.clear();
    };
    while (!live.empty()) {
        std::vector<pav_den_job> jobs; std::vector<uint32_t> owners, rest;
        uint64_t budget = 0;
        const double t_p0 = now();
        // top of the while-loop, inv.py:223-260.  Lifts first (every live region, no side effects) ...
        struct Pre { uint8_t kind; uint32_t job; Rgn tig; std::string err; };   // kind 0: too large, 1: lift raised, 2: not liftable, 3: lifted
        std::vector<Pre> pre(live.size());
        {   // both ends of every live region through the alignment table: point lifts on the device (lift_dev.hip; PAV_LIFT_HOST=1:
            // the host tables).  The first round asks for them here; later rounds find them done - they travelled with the
            // breakpoint lifts of the round before (one round trip to the device per round, not two).
            std::vector<uint32_t> first;
            if (!carried_valid) {
                std::vector<LiftQuery> lq;
                queue_region_lifts(live, lq, first);
                const int rcl = D.lift_batch(lq, carried);
                if (rcl != PAV_OK) return rcl;
            } else first.swap(carried_first);
            carried_valid = false;
            const std::vector<Driver::Point> &lp = carried;
            pool.run(live.size(), CHUNK, [&](size_t q) {
                if (first[q] == ~0u) { pre[q].kind = 0; return; }
                Pre &x = pre[q];
                bool ok = false;
                x.kind = !D.region_from_qry_points(lp[first[q]], lp[first[q] + 1], x.tig, ok, x.err) ? 1 : ok ? 3 : 2;
            });
        }
        t_lift += now() - t_p0;
        // ... then which of them this round takes (a budget of bases per batch, in region order) ...
        std::vector<uint32_t> taken;                                              // positions in `live`
        for (uint32_t q = 0; q < live.size(); ++q) {
            if (budget > max_batch_bp && !owners.empty()) { rest.push_back(live[q]); continue; }
            taken.push_back(q);
            if (pre[q].kind != 3) continue;
            pre[q].job = (uint32_t)owners.size();
            owners.push_back(live[q]);
            budget += (uint64_t)scans[live[q]].region_ref.len() + (uint64_t)pre[q].tig.len();
        }
        jobs.assign(owners.size(), pav_den_job{});
        // ... then their log lines and job descriptors
        pool.run(taken.size(), CHUNK, [&](size_t t) {
            const uint32_t q = taken[t], i = live[q];
            Scan &sc = scans[i];
            Pre &x = pre[q];
            if (x.kind == 0) {
                log(i, "Region size exceeds max: " + D.base1(sc.region_ref) + " (" + fmt_i(sc.region_ref.len()) + " > " + fmt_i(max_region_size) + ")");
                finish(i, PAV_INV_NONE); return;
            }
            if (x.kind == 1) { S->errors[i] = x.err; finish(i, PAV_INV_ERROR); return; }
            if (x.kind == 2) { log(i, "Could not lift reference region onto contigs: " + D.base1(sc.region_ref)); finish(i, PAV_INV_NONE); return; }
            sc.region_tig = x.tig;
            sc.expansion_count += 1;                                        // (its "Scanning region" line is written while the batch runs)
            pav_den_job &j = jobs[x.job];
            j.ref_id = (uint32_t)sc.region_ref.chrom; j.tig_id = (uint32_t)sc.region_tig.chrom;
            j.ref_pos = (uint64_t)sc.region_ref.pos; j.ref_end = (uint64_t)sc.region_ref.end;
            j.tig_pos = (uint64_t)sc.region_tig.pos; j.tig_end = (uint64_t)sc.region_tig.end;
            j.ref_rc = sc.region_tig.is_rev ? 1 : 0; j.state_run_smooth = srs_of(sc.region_tig.len());
        });
        if (jobs.empty()) { live.swap(rest); continue; }
        std::vector<pav_den_result> res(jobs.size());
        double t0 = now();
        t_pre += t0 - t_p0;
        if (timing) fprintf(stderr, "[pav timing]   scan round: %zu jobs, lift + jobs %.2f ms\n", jobs.size(), (t0 - t_p0) * 1e3);
        ctx->den_scan_only = true;                                          // a call needs REV k-mers: tables of FWD-only regions are never asked for
        // Texts that nothing in this round depends on are written while the round's kernels run (pav_density_batch calls back once,
        // between its last launch and its synchronisation): the decision lines of the round before, then this round's first lines.
        bool overlapped = false;
        auto texts = [&]() {
            overlapped = true;
            flush_notes();
            pool.run(owners.size(), CHUNK, [&](size_t j) { const uint32_t i = owners[j]; log(i, "Scanning region: " + D.base1(scans[i].region_ref)); });
        };
        ctx->den_overlap = texts;
        // The round's decisions are taken on the device too (density.hip k_round_decide), right behind the batch's last kernel: the
        // breakpoint queries of the flanked regions and the ends of the expanded ones, lifted in the same launch set; queries and
        // answers come back with the batch.  Below, an answer is used where its query equals the one this driver would ask.
        RoundHook hook;
        const size_t nj = jobs.size();
        const bool dev_round = S->on_device && getenv("PAV_ROUND_HOST") == nullptr && nj > 0;
        const LiftQuery *h_rq = nullptr; const LiftAnswer *h_ra = nullptr;
        if (dev_round) {
            const size_t b_in = sizeof(RoundJobIn) * nj, b_q = sizeof(LiftQuery) * 4 * nj, b_a = sizeof(LiftAnswer) * 4 * nj;
            if (b_in + b_q + b_a + 256 > S->h_round_cap) {
                if (S->h_round) (void)hipHostFree(S->h_round);
                S->h_round = nullptr; S->h_round_cap = 0;
                const size_t cap = (b_in + b_q + b_a) * 2 + 4096;
                PAV_HIP(ctx, hipHostMalloc(&S->h_round, cap, hipHostMallocDefault));
                S->h_round_cap = cap;
            }
            PAV_HIP(ctx, S->d_rin.reserve(b_in)); PAV_HIP(ctx, S->d_rq.reserve(b_q)); PAV_HIP(ctx, S->d_ra.reserve(b_a));
            uint8_t *h = static_cast<uint8_t *>(S->h_round);
            RoundJobIn *rin = reinterpret_cast<RoundJobIn *>(h);
            for (size_t j = 0; j < nj; ++j) {
                const Scan &sc = scans[owners[j]];
                rin[j] = RoundJobIn{sc.region_ref.chrom, sc.region_tig.chrom, sc.region_tig.is_rev ? 1 : 0, sc.expansion_count,
                                    sc.region_ref.pos, sc.region_ref.end, sc.region_tig.pos, (int64_t)ctx->seq[PAV_ROLE_REF].len[(size_t)sc.region_ref.chrom]};
            }
            PAV_HIP(ctx, hipMemcpyAsync(S->d_rin.p, h, b_in, hipMemcpyHostToDevice, ctx->stream));
            const size_t o_q = (b_in + 63) / 64 * 64, o_a = o_q + (b_q + 63) / 64 * 64;
            h_rq = reinterpret_cast<const LiftQuery *>(h + o_q); h_ra = reinterpret_cast<const LiftAnswer *>(h + o_a);
            hook.d_in = S->d_rin.as<RoundJobIn>(); hook.d_queries = S->d_rq.p; hook.n_jobs = (uint32_t)nj;
            hook.min_exp_count = min_exp_count; hook.k = k;
            hook.after = [&, h, o_q, o_a, b_q, b_a, nj]() -> int {
                const int rcl = lift_points(ctx, S->dev, S->d_rq.as<LiftQuery>(), S->d_ra.as<LiftAnswer>(), (uint32_t)(4 * nj));
                if (rcl != PAV_OK) return rcl;
                PAV_HIP(ctx, hipMemcpyAsync(h + o_q, S->d_rq.p, b_q, hipMemcpyDeviceToHost, ctx->stream));
                PAV_HIP(ctx, hipMemcpyAsync(h + o_a, S->d_ra.p, b_a, hipMemcpyDeviceToHost, ctx->stream));
                return PAV_OK;
            };
            ctx->den_round = &hook;
        }
        int rc = pav_density_batch(ctx, (uint32_t)jobs.size(), jobs.data(), &pp->den, res.data());
        ctx->den_round = nullptr;
        ctx->den_overlap = nullptr;
        ctx->den_scan_only = false;
        if (!overlapped && rc == PAV_OK) texts();                           // (a batch that took the host-planned path from the start)
        t_batch += now() - t0;
        if (rc != PAV_OK) return rc;
        const double t_q0 = now();
        auto inv_bounds = [&](const Scan &sc, Rgn &t_outer, Rgn &t_inner) -> int {   // 0: bounds set; 1: no inverted state; 2: run too short (max_run in t_outer.pos)
            const auto &rl = sc.state_rl;
            bool any_inv = false; uint32_t max_run = 0;
            const pav_run *inv_first = nullptr, *inv_last = nullptr;
            for (const auto &x : rl) if (x.state == 2) { any_inv = true; max_run = std::max(max_run, x.count); if (!inv_first) inv_first = &x; inv_last = &x; }
            if (!any_inv) return 1;
            if (max_run < 100) { t_outer.pos = max_run; return 2; }               // MIN_INV_KMER_RUN
            t_outer.role = t_inner.role = PAV_ROLE_TIG; t_outer.chrom = t_inner.chrom = sc.region_tig.chrom;
            t_outer.is_rev = t_inner.is_rev = sc.region_tig.is_rev;
            t_outer.pos = rl[1].pos + sc.region_tig.pos; t_outer.end = rl[rl.size() - 2].end + sc.region_tig.pos + k;
            t_inner.pos = inv_first->pos + sc.region_tig.pos; t_inner.end = inv_last->end + sc.region_tig.pos + k;
            if (t_outer.pos > t_outer.end) std::swap(t_outer.pos, t_outer.end);
            if (t_inner.pos > t_inner.end) std::swap(t_inner.pos, t_inner.end);
            return 0;
        };
        // after the density call, per region (inv.py:268-351) ...
        struct Dec { uint8_t what = 0, k1 = 0; int note = 0; CallFetch cf{}; };                   // what 0: finished, 1: expanded, 2: call, 3: internal error,
        std::vector<Dec> dec(jobs.size());                                           //      4: flanked, its breakpoints wait for their lifts
        struct Cand { Rgn t_outer, t_inner; };
        std::vector<Cand> cand(jobs.size());
        pool.run(jobs.size(), CHUNK, [&](size_t jj) {
            const uint32_t j = (uint32_t)jj;
            const uint32_t i = owners[j];
            Scan &sc = scans[i];
            const pav_den_result &r = res[j];
            S->results[i].iterations = (uint32_t)sc.expansion_count;
            S->results[i].n_near_tie += r.n_near_tie; S->results[i].n_unresolved += r.n_unresolved;
            // after the density call, inv.py:268-351
            if (r.status == PAV_DEN_FAIL) {
                std::string stderr_text;
                if (r.fail_kind == 2) {
                    std::string kmer(k, 'A');
                    for (int b = 0; b < k; ++b) kmer[(size_t)b] = "ACGT"[(r.max_kmer >> (2 * (k - 1 - b))) & 3];
                    stderr_text = "K-mer count exceeds max: " + fmt_i(r.max_count) + " > " + fmt_i(pp->den.max_ref_kmer_count) + " (" + kmer + "): " +
                                  D.base1(sc.region_ref) + "\n";
                }
                log(i, "Received return code 125 from scripts/density.py for region " + D.base1(sc.region_ref) + ":\n" + stderr_text);
                finish(i, PAV_INV_NONE); return;
            }
            if (r.n_rows == 0) { log(i, "No informative reference k-mers in forward or reverse orientation in region"); finish(i, PAV_INV_NONE); return; }
            sc.state_rl.resize(r.n_runs);
            if (r.n_runs && pav_density_runs(ctx, j, sc.state_rl.data()) != PAV_OK) { dec[j].what = 3; return; }
            sc.n_rows = r.n_rows;
            const auto &rl = sc.state_rl;
            if (rl.size() == 1 && (rl[0].state == 0 || rl[0].state == -1) && sc.expansion_count >= min_exp_count) {
                dec[j].note = sc.expansion_count;                                   // most regions end here: the line is written later (flush_notes)
                finish(i, PAV_INV_NONE); return;
            }
            if (rl.size() > 2 && rl.front().state == 0 && rl.back().state == 0) {
                // ---- characterise, inv.py:353-454 ----------------------------------------------------------
                Rgn t_outer, t_inner;
                const int why = inv_bounds(sc, t_outer, t_inner);
                if (why == 1) { log(i, "No inverted states found"); finish(i, PAV_INV_NONE); return; }
                if (why == 2) {
                    log(i, "Longest run of strictly inverted k-mers (" + fmt_i(t_outer.pos) + ") does not meet the minimum threshold (100)");
                    finish(i, PAV_INV_NONE); return;
                }
                cand[j].t_outer = t_outer; cand[j].t_inner = t_inner;
                dec[j].what = 4;                                                    // lifted below, all flanked regions of the round at once
                return;
            }
            // Expand, inv.py:309-342
            const int64_t last_len = sc.region_ref.len();
            const int64_t expand_bp = (int64_t)(int32_t)((double)last_len * 1.5);        // np.int32(len * EXPAND_FACTOR)
            double balance = 0.5;
            if (rl.size() > 2) { if (rl.front().state == 0) balance = 0.25; else if (rl.back().state == 0) balance = 0.75; }
            D.expand(sc.region_ref, expand_bp, balance);
            if (sc.region_ref.len() == last_len) { log(i, "Reached reference limits, cannot expand"); finish(i, PAV_INV_NONE); return; }
            dec[j].what = 1;
        });
        // the regions that go on, in region order (known now: a flanked region ends in this round, as a call or not)
        std::vector<uint32_t> next;
        for (uint32_t j = 0; j < jobs.size(); ++j) if (dec[j].what == 1) next.push_back(owners[j]);
        next.insert(next.end(), rest.begin(), rest.end());
        {   // the breakpoint regions of the flanked jobs -> reference (inv.py:393-406): outer ends without, inner ends with the gap rule;
            // behind them in the same batch the region ends of the next round
            std::vector<LiftQuery> lq;
            std::vector<uint32_t> pend;
            for (uint32_t j = 0; j < jobs.size(); ++j) {
                if (dec[j].what != 4) continue;
                pend.push_back(j);
                const Cand &c = cand[j];
                lq.push_back(LiftQuery{1, c.t_outer.chrom, 0, 0, c.t_outer.pos}); lq.push_back(LiftQuery{1, c.t_outer.chrom, 0, 0, c.t_outer.end});
                lq.push_back(LiftQuery{1, c.t_inner.chrom, 1, 0, c.t_inner.pos}); lq.push_back(LiftQuery{1, c.t_inner.chrom, 1, 0, c.t_inner.end});
            }
            const size_t n_cand_q = lq.size();
            queue_region_lifts(next, lq, carried_first);
            std::vector<Driver::Point> lp;
            // the device's queries, job by job: a flanked job's four, a continuing job's two.  Every query asked here must be there
            // (a region that waited for this batch - `rest` - was not part of it: then the batch of lifts is asked as before)
            bool from_device = hook.ran && rest.empty();
            if (from_device) {
                std::vector<uint32_t> slot(lq.size(), ~0u);
                for (size_t pq = 0; pq < pend.size(); ++pq) for (uint32_t t = 0; t < 4; ++t) slot[4 * pq + t] = 4 * pend[pq] + t;
                {
                    size_t nx = 0;
                    for (uint32_t j = 0; j < jobs.size(); ++j) {
                        if (dec[j].what != 1) continue;
                        const uint32_t f = carried_first[nx++];
                        if (f != ~0u) { slot[f] = 4 * j; slot[f + 1] = 4 * j + 1; }
                    }
                }
                for (size_t i = 0; i < lq.size() && from_device; ++i) {
                    if (slot[i] == ~0u) { from_device = false; break; }
                    const LiftQuery &a = lq[i], &b = h_rq[slot[i]];
                    from_device = a.axis == b.axis && a.seq == b.seq && a.gap == b.gap && a.pos == b.pos;
                }
                if (from_device) {
                    lp.assign(lq.size(), Driver::Point{});
                    for (size_t i = 0; i < lq.size(); ++i) {
                        lp[i].a = h_ra[slot[i]];
                        if (lp[i].a.status >= LIFT_ERR_OP) lp[i].err = D.lift_error(lq[i], lp[i].a);
                    }
                    S->n_round_dev += 1;
                }
            }
            if (!from_device) {
                if (hook.ran) S->n_round_host += 1;
                const int rcl = D.lift_batch(lq, lp);
                if (rcl != PAV_OK) return rcl;
            }
            carried.assign(lp.begin() + (ptrdiff_t)n_cand_q, lp.end());
            for (uint32_t &f : carried_first) if (f != ~0u) f -= (uint32_t)n_cand_q;
            carried_valid = true;
            pool.run(pend.size(), CHUNK, [&](size_t pp_) {
                const uint32_t j = pend[pp_];
                const uint32_t i = owners[j];
                Scan &sc = scans[i];
                const pav_den_result &r = res[j];
                const Rgn &t_outer = cand[j].t_outer, &t_inner = cand[j].t_inner;
                dec[j].what = 0;
                Rgn r_outer, r_inner; bool ok = false; std::string err;
                if (!D.region_from_sub_points(lp[4 * pp_], lp[4 * pp_ + 1], r_outer, ok, err)) { S->errors[i] = err; finish(i, PAV_INV_ERROR); return; }
                if (!ok) { log(i, "Failed lifting outer INV region to reference: " + D.base1(t_outer)); finish(i, PAV_INV_NONE); return; }
                bool ok2 = false;
                if (!D.region_from_sub_points(lp[4 * pp_ + 2], lp[4 * pp_ + 3], r_inner, ok2, err)) { S->errors[i] = err; finish(i, PAV_INV_ERROR); return; }
                if (!ok2) r_inner = r_outer;
                pav_inv_result &out = S->results[i];
                set_rgn(out.tig_outer, t_outer); set_rgn(out.tig_inner, t_inner); set_rgn(out.ref_outer, r_outer); set_rgn(out.ref_inner, r_inner);
                set_rgn(out.ref_discovery, sc.region_ref); set_rgn(out.tig_discovery, sc.region_tig);
                out.found = 1;                                                      // the caller prints the line (inv.py:408)
                S->found[i] = "INV Found: outer=" + D.base1(t_outer) + ", inner=" + D.base1(t_inner) + " (ref outer=" + D.base1(r_outer) +
                              ", inner=" + D.base1(r_inner) + ")\n";
                if ((double)r_outer.len() < (double)t_outer.len() * 0.6) {
                    log(i, "Reference region too short: Reference region length (" + fmt_commas(r_outer.len()) + ") is not within " + fmt_f2(60.0) +
                           "% of the contig region length (" + fmt_commas(t_outer.len()) + ")");
                    finish(i, PAV_INV_NONE); return;
                }
                if ((double)t_outer.len() < (double)r_outer.len() * 0.6) {
                    log(i, "Contig region too short: Contig region length (" + fmt_commas(t_outer.len()) + ") is not within " + fmt_f2(60.0) +
                           "% of the reference region length (" + fmt_commas(r_outer.len()) + ")");
                    finish(i, PAV_INV_NONE); return;
                }
                // density table + FLANK / MATCH (inv.py:440-442, 457-561): queued for the end of this round, while the batch
                // is still resident.  The annotation receives the *reference* discovery start (inv.py:441).
                const uint32_t n = r.n_rows;
                CallFetch cf{};
                cf.job = j; cf.n = n; cf.ref_id = (uint32_t)r_outer.chrom;
                cf.ref_up_pos = (uint64_t)r_outer.pos; cf.ref_up_end = (uint64_t)r_inner.pos;
                cf.ref_dn_pos = (uint64_t)r_inner.end; cf.ref_dn_end = (uint64_t)r_outer.end;
                cf.base = sc.region_ref.pos;
                cf.tig_up_pos = std::min(t_outer.pos, t_inner.pos); cf.tig_up_end = std::max(t_outer.pos, t_inner.pos);
                cf.tig_dn_pos = std::min(t_inner.end, t_outer.end); cf.tig_dn_end = std::max(t_inner.end, t_outer.end);
                dec[j].what = 2; dec[j].cf = cf; dec[j].k1 = r.state_count[1] != 0;
                out.n_rows = n;
                out.svlen = (uint64_t)r_outer.len();
                log(i, "Found inversion: " + D.name(PAV_ROLE_REF, r_outer.chrom) + "-" + fmt_i(r_outer.pos + 1) + "-INV-" + fmt_i(r_outer.len()));
                finish(i, PAV_INV_CALL);
            });
        }
        // ... and, in region order, the calls of the round and the regions that go on
        std::vector<CallFetch> round_calls; std::vector<uint32_t> round_owner; std::vector<uint8_t> round_k1;   // round_k1: the call has FWDREV k-mers
        for (uint32_t j = 0; j < jobs.size(); ++j) {
            if (dec[j].what == 3) return fail(ctx, PAV_E_STATE, "pav_inv_scan_batch: the state runs of job %u could not be read", j);
            if (dec[j].what == 2) { round_calls.push_back(dec[j].cf); round_owner.push_back(owners[j]); round_k1.push_back(dec[j].k1); }
            if (dec[j].note) notes.emplace_back(owners[j], dec[j].note);
        }
        t_post += now() - t_q0;
        if (timing) fprintf(stderr, "[pav timing]   scan round: decisions %.2f ms, %zu calls\n", (now() - t_q0) * 1e3, round_calls.size());
        if (!round_calls.empty()) {                                          // one pinned block, one synchronisation per round
            double t0 = now();
            // one pinned block per round, column-major over the whole round: K0 | K1 | K2 | KMER | INDEX | STATE_MER | STATE |
            // FLANK | MATCH, each column holding the calls one after the other (nine bulk device-to-host copies)
            // calls without FWDREV k-mers (KERN_FWDREV all zeros) go last: their part of that column stays on the device
            {
                std::vector<size_t> order(round_calls.size());
                for (size_t c = 0; c < order.size(); ++c) order[c] = c;
                std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return round_k1[a] > round_k1[b]; });
                std::vector<CallFetch> rc2; std::vector<uint32_t> ro2; std::vector<uint8_t> rk2;
                for (size_t c : order) { rc2.push_back(round_calls[c]); ro2.push_back(round_owner[c]); rk2.push_back(round_k1[c]); }
                round_calls.swap(rc2); round_owner.swap(ro2); round_k1.swap(rk2);
            }
            size_t rows = 0, k1_rows = 0;
            for (size_t c = 0; c < round_calls.size(); ++c) { rows += round_calls[c].n; if (round_k1[c]) k1_rows += round_calls[c].n; }
            CallStage &stg = S->next_stage();
            stg.entries.clear();
            stg.host = nullptr;
            (void)rows;
            rc = density_fetch_calls(ctx, round_calls, k1_rows, stg);     // queued; says where every call's rows sit (stg.row0, stg.rows)
            if (rc != PAV_OK) return rc;
            for (size_t c = 0; c < round_calls.size(); ++c) {
                stg.entries.push_back(CallStage::Entry{round_owner[c], round_calls[c].n, stg.row0[c], round_k1[c] != 0});
                auto tab = std::make_unique<InvTable>();            // rows known now; the columns are bound with the host block
                tab->n = round_calls[c].n;
                S->tables[round_owner[c]] = std::move(tab);
            }
            if (!pp->lazy_tables) {
                rc = bind_round(ctx, S, stg);
                if (rc == PAV_OK) rc = density_copy_now(ctx, stg);
            }
            t_table += now() - t0;
            if (rc != PAV_OK) return rc;
        }
        live.swap(next);
    }
    flush_notes();
    if (timing) fprintf(stderr, "[pav timing] inv_scan_batch setup %.2f ms: texts %.3f, wait tables %.3f, tables %.3f, scans %.3f, first lines %.3f\n", (t_start - t_entry) * 1e3,
                        (tl[0] - t_entry) * 1e3, (tl[1] - tl[0]) * 1e3, (tl[2] - tl[1]) * 1e3, (tl[3] - tl[2]) * 1e3, (tl[4] - tl[3]) * 1e3);
    if (timing) fprintf(stderr, "[pav timing] inv_scan_batch %.1f ms: density_batch %.1f, call tables + annotate %.1f, lift + jobs %.2f (lifting %.2f), decisions %.2f; "
                        "rounds whose lifts came with the batch / were asked for afterwards so far: %llu / %llu\n",
                        (now() - t_start) * 1e3, t_batch * 1e3, t_table * 1e3, t_pre * 1e3, t_lift * 1e3, t_post * 1e3,
                        (unsigned long long)S->n_round_dev, (unsigned long long)S->n_round_host);
    // the first line of every region's log (inv.py:194-199) is written now, with the last kernels of the scan still running: it
    // depends on the flagged region alone, and a thousand of them were 0.08 ms in front of the first round
    pool.run(n_regions, CHUNK, [&](size_t i) {
        const Scan &sc = scans[i];
        S->logs[i].insert(0, "Scanning for inversions in flagged region: " + D.base1(sc.flag) + " (flagged region record id = " + D.region_id(sc.flag) + ")\n");
    });
    for (uint32_t i = 0; i < n_regions; ++i) {
        S->results[i].log_bytes = (uint32_t)S->logs[i].size();
        S->results[i].error_bytes = (uint32_t)S->errors[i].size();
        results[i] = S->results[i];
    }
    return PAV_OK;
}

int pav_inv_text(pav_ctx *ctx, uint32_t region, int what, char *buf, uint32_t buf_len) {
    if (!ctx || !buf) return PAV_E_ARG;
    InvState *S = istate(ctx);
    if (region >= S->logs.size()) return fail(ctx, PAV_E_STATE, "pav_inv_text: no such region in the last scan");
    const std::string &s = what == 0 ? S->logs[region] : S->errors[region];
    if (buf_len < s.size() + 1) return fail(ctx, PAV_E_ARG, "pav_inv_text: buffer too small");
    memcpy(buf, s.data(), s.size());
    buf[s.size()] = 0;
    return PAV_OK;
}

int pav_inv_texts(pav_ctx *ctx, int what, char *buf, uint64_t buf_len, uint64_t *off) {
    if (!ctx || !off || what < 0 || what > 2) return PAV_E_ARG;
    InvState *S = istate(ctx);
    const std::vector<std::string> &src = what == 0 ? S->logs : what == 1 ? S->errors : S->found;
    uint64_t at = 0;
    for (size_t i = 0; i < src.size(); ++i) {
        off[i] = at;
        if (buf) {                                                       // no buffer: only the offsets (the sizes) are wanted
            if (at + src[i].size() > buf_len) return fail(ctx, PAV_E_ARG, "pav_inv_texts: buffer too small");
            if (!src[i].empty()) memcpy(buf + at, src[i].data(), src[i].size());
        }
        at += src[i].size();
    }
    off[src.size()] = at;
    return PAV_OK;
}

int pav_inv_table(pav_ctx *ctx, uint32_t region, int64_t *index, int8_t *state_mer, int8_t *state, double *kern_fwd, double *kern_fwdrev,
                  double *kern_rev, uint64_t *kmer, uint8_t *flank, uint8_t *match) {
    if (!ctx) return PAV_E_ARG;
    InvState *S = istate(ctx);
    if (region >= S->tables.size() || !S->tables[region]) return fail(ctx, PAV_E_STATE, "pav_inv_table: region has no call in the last scan");
    { const int rcw = tables_on_host(ctx, S); if (rcw != PAV_OK) return rcw; }
    const InvTable &t = *S->tables[region];
    const size_t n = t.n;
    if (index) for (size_t i = 0; i < n; ++i) index[i] = t.index[i];
    if (state_mer) memcpy(state_mer, t.state_mer, n);
    if (state) memcpy(state, t.state, n);
    if (kern_fwd) memcpy(kern_fwd, t.kern[0], 8 * n);
    if (kern_fwdrev) memcpy(kern_fwdrev, t.kern[1], 8 * n);
    if (kern_rev) memcpy(kern_rev, t.kern[2], 8 * n);
    if (kmer) memcpy(kmer, t.kmer, 8 * n);
    if (flank) memcpy(flank, t.flank, n);
    if (match) memcpy(match, t.match, n);
    return PAV_OK;
}

int pav_inv_table_view(pav_ctx *ctx, uint32_t region, uint32_t *n_rows, const uint32_t **index, const int8_t **state_mer,
                       const int8_t **state, const double **kern_fwd, const double **kern_fwdrev, const double **kern_rev,
                       const uint64_t **kmer, const uint8_t **flank, const uint8_t **match) {
    if (!ctx || !n_rows) return PAV_E_ARG;
    InvState *S = istate(ctx);
    if (region >= S->tables.size() || !S->tables[region]) return fail(ctx, PAV_E_STATE, "pav_inv_table_view: region has no call in the last scan");
    { const int rcw = tables_on_host(ctx, S); if (rcw != PAV_OK) return rcw; }
    const InvTable &t = *S->tables[region];
    *n_rows = t.n;
    if (index) *index = t.index;
    if (state_mer) *state_mer = t.state_mer;
    if (state) *state = t.state;
    if (kern_fwd) *kern_fwd = t.kern[0];
    if (kern_fwdrev) *kern_fwdrev = t.kern[1];
    if (kern_rev) *kern_rev = t.kern[2];
    if (kmer) *kmer = t.kmer;
    if (flank) *flank = t.flank;
    if (match) *match = t.match;
    return PAV_OK;
}

// Density tables of rule call_inv_batch (rules/call_inv.snakefile:287-291: call.df.to_csv(path, sep='\t', index=False,
// compression='gzip')) written straight from the host copies of the last scan: same text as pandas, formatted on host
// threads, ".gz" names as concatenated gzip members compressed in parallel.
int pav_inv_write_tables(pav_ctx *ctx, uint32_t n, const uint32_t *regions, const char *const *paths, int threads, int gzip_level) {
    if (!ctx || (n && (!regions || !paths))) return fail(ctx, PAV_E_ARG, "pav_inv_write_tables: null argument");
    InvState *S = istate(ctx);
    const int level = gzip_level > 0 ? gzip_level : 6;
    if (device_writer_enabled()) {
        // The device writer (textdev.hip): rows -> text -> gzip in HBM, straight from the column blocks the scan left resident
        // (CallStage::buf, one per round: K0 | K1 | K2 | KMER | INDEX | STATE_MER | STATE | FLANK | MATCH); the tables never come to
        // the host.  A region whose block is not resident any more falls through to the host writer below.
        std::vector<DenTableDev> tabs; std::vector<std::string> ps;
        bool all = true;
        for (uint32_t q = 0; q < n && all; ++q) {
            if (!paths[q]) return fail(ctx, PAV_E_ARG, "pav_inv_write_tables: null path");
            if (regions[q] >= S->tables.size() || !S->tables[regions[q]])
                return fail(ctx, PAV_E_STATE, "pav_inv_write_tables: region %u has no call in the last scan", regions[q]);
            bool found = false;
            for (size_t r = 0; r < S->stage_used && !found; ++r) {
                const CallStage &stg = *S->stage_sets[S->cur_set][r];
                if (!stg.buf.p || !stg.rows) continue;
                for (const CallStage::Entry &e : stg.entries) {
                    if (e.owner != regions[q]) continue;
                    const size_t rows = stg.rows;
                    const double *k0 = stg.buf.as<double>(), *k1 = k0 + rows, *k2 = k1 + rows;
                    const unsigned long long *kmer = reinterpret_cast<const unsigned long long *>(k2 + rows);
                    const uint32_t *index = reinterpret_cast<const uint32_t *>(kmer + rows);
                    const int8_t *sm = reinterpret_cast<const int8_t *>(index + rows), *stt = sm + rows;
                    const uint8_t *fl = reinterpret_cast<const uint8_t *>(stt + rows), *ma = fl + rows;
                    const size_t o = e.row0;
                    tabs.push_back(DenTableDev{k0 + o, e.has_k1 ? k1 + o : nullptr, k2 + o, kmer + o, index + o, sm + o, stt + o, fl + o, ma + o, e.n});
                    ps.emplace_back(paths[q]);
                    found = true;
                    break;
                }
            }
            all = all && found;
        }
        if (all) {
            PAV_HIP(ctx, hipSetDevice(ctx->device));
            PAV_HIP(ctx, hipStreamSynchronize(ctx->stream));          // the scan's last gather / annotate kernels
            const int rcd = text_density_tables(ctx, tabs, ps, level);
            if (rcd != PAV_OK) return fail(ctx, rcd, "%s", pav_last_error(nullptr));
            return PAV_OK;
        }
    }
    { const int rcw = tables_on_host(ctx, S); if (rcw != PAV_OK) return rcw; }
    if (threads <= 0) threads = (int)std::min<unsigned>(16, std::max<unsigned>(1, std::thread::hardware_concurrency()));
    static const char *FLANK_TEXT[3] = {"", "UP", "DN"};
    static const char *MATCH_TEXT[4] = {"", "SAME", "OTHER", ""};      // 3 = NaN, written as the empty na_rep
    const std::string header = "INDEX\tSTATE_MER\tSTATE\tKERN_FWD\tKERN_FWDREV\tKERN_REV\tKMER\tFLANK\tMATCH\n";
    // one pool over the (table, chunk of rows) pairs of all tables: calls range from 10^4 to 10^6 rows
    const std::string gz_suffix = ".gz";
    constexpr uint64_t CHUNK_ROWS = 1 << 15;
    struct Task { uint32_t q; uint64_t a, b; };
    std::vector<Task> tasks;
    std::vector<uint32_t> first_task(n + 1, 0);
    for (uint32_t q = 0; q < n; ++q) {
        const uint32_t region = regions[q];
        if (region >= S->tables.size() || !S->tables[region])
            return fail(ctx, PAV_E_STATE, "pav_inv_write_tables: region %u has no call in the last scan", region);
        if (!paths[q]) return fail(ctx, PAV_E_ARG, "pav_inv_write_tables: null path");
        const uint64_t rows = S->tables[region]->n;
        first_task[q] = (uint32_t)tasks.size();
        for (uint64_t a = 0; a < rows || a == 0; a += CHUNK_ROWS) { tasks.push_back(Task{q, a, std::min(rows, a + CHUNK_ROWS)}); if (a + CHUNK_ROWS >= rows) break; }
    }
    first_task[n] = (uint32_t)tasks.size();
    std::vector<std::string> done(tasks.size());
    std::atomic<size_t> next{0};
    std::atomic<bool> ok{true};
    auto work = [&]() {
        std::string text;
        for (size_t k; (k = next.fetch_add(1)) < tasks.size();) {
            const Task &tk = tasks[k];
            const InvTable &t = *S->tables[regions[tk.q]];
            const std::string p(paths[tk.q]);
            const bool gz = p.size() > 3 && p.compare(p.size() - 3, 3, gz_suffix) == 0;
            text.clear();
            text.reserve((size_t)(tk.b - tk.a) * 110 + header.size());
            if (tk.a == 0) text = header;
            for (uint64_t i = tk.a; i < tk.b; ++i) {
                put_u64(text, t.index[i]); text.push_back('\t');
                put_i64(text, t.state_mer[i]); text.push_back('\t');
                put_i64(text, t.state[i]); text.push_back('\t');
                put_f64_repr(text, t.kern[0][i]); text.push_back('\t');
                put_f64_repr(text, t.kern[1][i]); text.push_back('\t');
                put_f64_repr(text, t.kern[2][i]); text.push_back('\t');
                put_u64(text, t.kmer[i]); text.push_back('\t');
                text += FLANK_TEXT[t.flank[i] < 3 ? t.flank[i] : 0]; text.push_back('\t');
                text += MATCH_TEXT[t.match[i] & 3]; text.push_back('\n');
            }
            if (gz) { if (!gz_member(text, level, done[k])) ok = false; } else done[k].swap(text);
        }
    };
    {
        std::vector<std::thread> pool;
        for (int w = 1; w < threads; ++w) pool.emplace_back(work);
        work();
        for (auto &th : pool) th.join();
    }
    if (!ok) return fail(ctx, PAV_E_ARG, "pav_inv_write_tables: zlib failed");
    for (uint32_t q = 0; q < n; ++q) {
        FILE *fh = fopen(paths[q], "wb");
        if (!fh) return fail(ctx, PAV_E_ARG, "pav_inv_write_tables: cannot open %s", paths[q]);
        for (uint32_t k = first_task[q]; k < first_task[q + 1]; ++k)
            if (!done[k].empty() && fwrite(done[k].data(), 1, done[k].size(), fh) != done[k].size()) {
                fclose(fh);
                return fail(ctx, PAV_E_ARG, "pav_inv_write_tables: short write to %s", paths[q]);
            }
        fclose(fh);
        for (uint32_t k = first_task[q]; k < first_task[q + 1]; ++k) std::string().swap(done[k]);
    }
    return PAV_OK;
}

// repr() of a float64 as DataFrame.to_csv writes it (text helper of the writers; exposed for the unit tests).
int pav_repr_f64(double value, char *out, int out_len) {
    std::string s;
    put_f64_repr(s, value);
    if (!out || out_len <= (int)s.size()) return PAV_E_ARG;
    memcpy(out, s.c_str(), s.size() + 1);
    return (int)s.size();
}

int pav_inv_tables(pav_ctx *ctx, uint32_t n_regions, const uint64_t *row_off, int64_t *index, int8_t *state_mer, int8_t *state,
                   double *kern_fwd, double *kern_fwdrev, double *kern_rev, uint64_t *kmer, uint8_t *flank, uint8_t *match) {
    if (!ctx || !row_off) return PAV_E_ARG;
    InvState *S = istate(ctx);
    if (n_regions != S->tables.size()) return fail(ctx, PAV_E_STATE, "pav_inv_tables: region count does not match the last scan");
    { const int rcw = tables_on_host(ctx, S); if (rcw != PAV_OK) return rcw; }
    for (uint32_t i = 0; i < n_regions; ++i) {
        if (!S->tables[i]) continue;
        const uint64_t o = row_off[i];
        int rc = pav_inv_table(ctx, i, index ? index + o : nullptr, state_mer ? state_mer + o : nullptr, state ? state + o : nullptr,
                               kern_fwd ? kern_fwd + o : nullptr, kern_fwdrev ? kern_fwdrev + o : nullptr, kern_rev ? kern_rev + o : nullptr,
                               kmer ? kmer + o : nullptr, flank ? flank + o : nullptr, match ? match + o : nullptr);
        if (rc != PAV_OK) return rc;
    }
    return PAV_OK;
}

}  // extern "C"
