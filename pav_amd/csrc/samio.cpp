// Native SAM reader for the alignment ingest in front of the hot path: the record loop of pavlib.align.get_align_bed
// (pavlib/align/align.py:666-794; rule align_get_read_bed, rules/align.snakefile:101-171) without pysam.  SAM text (plain, gzip,
// BGZF) -> the per-record quantities that function reads from pysam's AlignedSegment, the CIGAR with soft clipping folded
// into hard clipping (clip_soft_to_hard, align.py:797-831) and count_cigar of it (align.py:534-663, for check_record).
// The host mirror (pav_amd/align/ingest.py) turns the columns into the reference's table.  Host code; no GPU needed.
//
// pysam semantics restated from the SAM specification and htslib / pysam sources (pysam itself is not in this image):
//   reference_start = POS - 1; reference_end = reference_start + max(1, sum of M D N = X lengths) (bam_endpos)
//   query_alignment_start = leading S after optional leading H; query_alignment_end = (sum of M I S = X) - trailing S
//   is_unmapped = FLAG & 4, is_reverse = FLAG & 16; cigartuples empty when CIGAR is '*'
#include "common.h"

#include "fileio.h"

#include <unordered_map>

struct pav_sam {
    uint64_t n_records = 0;
    std::string header;
    std::vector<std::string> names[2];                        // RNAME / QNAME of the kept records, order of first appearance
    struct Row {
        int64_t index, pos, end, qas, qae, clip_h, tig_map_pos, ref_bp, tig_bp;
        uint32_t chrom_id, qry_id, err_kind, err_op, err_len, err_char;
        int32_t mapq, flag;
        uint8_t has_m, status, rg_kind, ao_kind;
        std::string cigar, rg, ao;
    };
    std::vector<Row> rows;
};

namespace pav {
namespace {

constexpr const char *OPS = "MIDNSHP=XB";

int op_code(uint8_t c) {
    switch (c) {
        case 'M': return 0; case 'I': return 1; case 'D': return 2; case 'N': return 3; case 'S': return 4;
        case 'H': return 5; case 'P': return 6; case '=': return 7; case 'X': return 8; case 'B': return 9;
        default: return -1;
    }
}

struct Op { int code; int64_t len; };

// count_cigar (align.py:534-663) on an operation list; error kinds as pav_trim_count (pav_amd/align/trim.py _CHECK_TEXT).
void count_cigar(const std::vector<Op> &ops, pav_sam::Row &r) {
    int64_t ref_bp = 0, tig_bp = 0, s_l = 0, h_l = 0, s_r = 0, h_r = 0;
    size_t i = 0;
    auto err = [&](uint32_t kind, size_t at) { r.err_kind = kind; r.err_op = (uint32_t)at; r.err_len = (uint32_t)ops[at].len; r.err_char = (uint8_t)OPS[ops[at].code]; };
    for (; i < ops.size() && (ops[i].code == 4 || ops[i].code == 5); ++i) {
        if (ops[i].code == 4) { if (s_l > 0) return err(1, i); s_l = ops[i].len; }
        else { if (h_l > 0) return err(2, i); if (s_l > 0) return err(3, i); h_l = ops[i].len; }
    }
    for (; i < ops.size(); ++i) {
        const int c = ops[i].code;
        const bool clipped = s_r > 0 || h_r > 0;
        if (c == 7 || c == 8) { if (clipped) return err(4, i); ref_bp += ops[i].len; tig_bp += ops[i].len; }
        else if (c == 1) { if (clipped) return err(4, i); tig_bp += ops[i].len; }
        else if (c == 2) { if (clipped) return err(4, i); ref_bp += ops[i].len; }
        else if (c == 4) { if (s_r > 0) return err(5, i); if (h_r > 0) return err(6, i); s_r = ops[i].len; }
        else if (c == 5) { if (h_r > 0) return err(7, i); h_r = ops[i].len; }
        else if (c == 0) return err(8, i);
        else return err(9, i);
    }
    r.ref_bp = ref_bp; r.tig_bp = tig_bp;
}

bool parse_int(const uint8_t *b, const uint8_t *e, int64_t &v) {
    if (b == e) return false;
    bool neg = false;
    if (*b == '-') { neg = true; if (++b == e) return false; }
    int64_t x = 0;
    for (; b < e; ++b) {
        if (*b < '0' || *b > '9') return false;
        if (x > (INT64_MAX - 9) / 10) return false;                    // does not fit: refused like any other non-number
        x = x * 10 + (*b - '0');
    }
    v = neg ? -x : x;
    return true;
}

struct Parsed { bool keep = false; bool bad = false; std::string error; std::string rname, qname; pav_sam::Row row; };

void parse_line(const uint8_t *b, const uint8_t *e, int64_t index, int min_mapq, Parsed &out) {
    const uint8_t *f[12];
    int nf = 0;
    f[nf++] = b;
    for (const uint8_t *p = b; nf < 12;) {
        const uint8_t *t = static_cast<const uint8_t *>(memchr(p, '\t', (size_t)(e - p)));
        if (!t) break;
        f[nf++] = t + 1;
        p = t + 1;
    }
    auto fe = [&](int i) { return i + 1 < nf ? f[i + 1] - 1 : e; };
    if (nf < 11) { out.bad = true; out.error = "fewer than 11 fields"; return; }
    int64_t flag = 0, pos = 0, mapq = 0;
    if (!parse_int(f[1], fe(1), flag) || !parse_int(f[3], fe(3), pos) || !parse_int(f[4], fe(4), mapq)) { out.bad = true; out.error = "FLAG / POS / MAPQ is not a number"; return; }
    const uint8_t *cb = f[5], *ce = fe(5);
    const bool no_cigar = ce - cb == 1 && *cb == '*';
    if ((flag & 4) || mapq < min_mapq || no_cigar || cb == ce) return;              // align.py:695-696
    pav_sam::Row &r = out.row;
    r = pav_sam::Row{};
    std::vector<Op> ops;
    for (const uint8_t *p = cb; p < ce;) {
        int64_t len = 0;
        const uint8_t *d = p;
        while (p < ce && *p >= '0' && *p <= '9') { len = len * 10 + (*p++ - '0'); if (len >= (int64_t)1 << 28) break; }
        // (a length of 2^28 or more does not exist in BAM and is refused by the device tokenizer as well: PAV_E_LIMIT)
        if (p == d || p == ce || op_code(*p) < 0) { out.bad = true; out.error = "malformed CIGAR"; return; }
        ops.push_back(Op{op_code(*p++), len});
    }
    out.keep = true;
    out.qname.assign(reinterpret_cast<const char *>(f[0]), (size_t)(fe(0) - f[0]));
    out.rname.assign(reinterpret_cast<const char *>(f[2]), (size_t)(fe(2) - f[2]));
    r.index = index; r.flag = (int32_t)flag; r.mapq = (int32_t)mapq;
    r.pos = pos - 1;
    int64_t rlen = 0, qlen = 0;
    for (const Op &o : ops) {
        if (o.code == 0 || o.code == 2 || o.code == 3 || o.code == 7 || o.code == 8) rlen += o.len;
        if (o.code == 0 || o.code == 1 || o.code == 4 || o.code == 7 || o.code == 8) qlen += o.len;
        if (o.code == 0) r.has_m = 1;
    }
    r.end = r.pos + std::max<int64_t>(rlen, 1);
    // pysam getQueryStart / getQueryEnd
    {
        int64_t start = 0;
        for (const Op &o : ops) {
            if (o.code == 5) { if (start != 0 && start != qlen) r.status = 1; }
            else if (o.code == 4) start += o.len;
            else break;
        }
        int64_t end = qlen;
        for (size_t i = ops.size(); i-- > 1;) {
            const Op &o = ops[i];
            if (o.code == 5) { if (end != qlen) r.status = 1; }
            else if (o.code == 4) end -= o.len;
            else break;
        }
        r.qas = start; r.qae = end;
    }
    r.clip_h = ops[0].code == 5 ? ops[0].len : 0;                                     // align.py:708-711
    // clip_soft_to_hard (align.py:797-831)
    size_t a = 0, z = ops.size();
    int64_t front = 0, back = 0;
    while (a < z && (ops[a].code == 4 || ops[a].code == 5)) front += ops[a++].len;
    while (z > a && (ops[z - 1].code == 4 || ops[z - 1].code == 5)) back += ops[--z].len;
    if (a == z) { r.status = 2; return; }
    std::vector<Op> t;
    if (front > 0) t.push_back(Op{5, front});
    t.insert(t.end(), ops.begin() + (long)a, ops.begin() + (long)z);
    if (back > 0) t.push_back(Op{5, back});
    r.tig_map_pos = t[0].code == 5 ? t[0].len : 0;
    for (const Op &o : t) { r.cigar += std::to_string(o.len); r.cigar.push_back(OPS[o.code]); }
    count_cigar(t, r);
    // optional fields: RG and AO (align.py:702, 758-759)
    for (const uint8_t *p = nf > 11 ? f[11] : e; p < e;) {
        const uint8_t *t2 = static_cast<const uint8_t *>(memchr(p, '\t', (size_t)(e - p)));
        const uint8_t *q = t2 ? t2 : e;
        if (q - p >= 5 && p[2] == ':' && p[4] == ':') {
            const bool rg = p[0] == 'R' && p[1] == 'G', ao = p[0] == 'A' && p[1] == 'O';
            if (rg || ao) {
                const uint8_t kind = p[3] == 'i' ? 1 : p[3] == 'f' ? 3 : 2;
                std::string val(reinterpret_cast<const char *>(p + 5), (size_t)(q - p - 5));
                if (rg) { r.rg_kind = kind; r.rg = val; } else { r.ao_kind = kind; r.ao = val; }
            }
        }
        p = q + 1;
    }
}

}  // namespace
}  // namespace pav

using namespace pav;

extern "C" {

int pav_sam_open(const char *path, int min_mapq, int threads, pav_sam **out) {
    if (!path || !out) return PAV_E_ARG;
    *out = nullptr;
    if (threads <= 0) threads = default_host_threads();
    FileText ft;
    std::string err;
    if (!read_file_text(path, threads, ft, err)) return fail(nullptr, PAV_E_ARG, "pav_sam_open: %s", err.c_str());
    auto sam = new pav_sam();
    const uint8_t *text = ft.text, *end = ft.text + ft.n;
    // header: '@' lines (everywhere in the file for the parser, the leading block for pav_sam_header)
    std::vector<std::pair<const uint8_t *, const uint8_t *>> lines;
    bool in_head = true;
    for (const uint8_t *p = text; p < end;) {
        const uint8_t *nl = static_cast<const uint8_t *>(memchr(p, '\n', (size_t)(end - p)));
        const uint8_t *e = nl ? nl : end;
        const uint8_t *le = e > p && e[-1] == '\r' ? e - 1 : e;
        if (le > p && *p == '@') {
            if (in_head) sam->header.append(reinterpret_cast<const char *>(p), (size_t)((nl ? nl + 1 : end) - p));
        } else if (le > p) {
            in_head = false;
            lines.emplace_back(p, le);
        }
        p = nl ? nl + 1 : end;
    }
    sam->n_records = lines.size();
    std::vector<Parsed> parsed(lines.size());
    parallel_for(lines.size(), threads, [&](size_t i) { parse_line(lines[i].first, lines[i].second, (int64_t)i, min_mapq, parsed[i]); });
    std::unordered_map<std::string, uint32_t> maps[2];
    for (size_t i = 0; i < parsed.size(); ++i) {
        Parsed &p = parsed[i];
        if (p.bad) { const std::string m = p.error; delete sam; return fail(nullptr, PAV_E_ARG, "pav_sam_open: %s: alignment record %zu: %s", path, i, m.c_str()); }
        if (!p.keep) continue;
        const std::string *nm[2] = {&p.rname, &p.qname};
        uint32_t ids[2];
        for (int w = 0; w < 2; ++w) {
            auto it = maps[w].find(*nm[w]);
            if (it == maps[w].end()) { it = maps[w].emplace(*nm[w], (uint32_t)sam->names[w].size()).first; sam->names[w].push_back(*nm[w]); }
            ids[w] = it->second;
        }
        p.row.chrom_id = ids[0]; p.row.qry_id = ids[1];
        sam->rows.push_back(std::move(p.row));
    }
    *out = sam;
    return PAV_OK;
}

void pav_sam_close(pav_sam *sam) { delete sam; }

int pav_sam_info(const pav_sam *sam, pav_sam_info_t *info) {
    if (!sam || !info) return PAV_E_ARG;
    info->n_records = sam->n_records;
    info->n_rows = sam->rows.size();
    info->n_ref = (uint32_t)sam->names[0].size();
    info->n_qry = (uint32_t)sam->names[1].size();
    info->cigar_bytes = info->tag_bytes = 0;
    for (const auto &r : sam->rows) { info->cigar_bytes += r.cigar.size(); info->tag_bytes += r.rg.size() + r.ao.size(); }
    info->header_bytes = sam->header.size();
    return PAV_OK;
}

const char *pav_sam_name(const pav_sam *sam, int which, uint32_t id) {
    if (!sam || which < 0 || which > 1 || id >= sam->names[which].size()) return nullptr;
    return sam->names[which][id].c_str();
}

int pav_sam_header(const pav_sam *sam, uint8_t *buf) {
    if (!sam || (!buf && !sam->header.empty())) return PAV_E_ARG;
    if (!sam->header.empty()) memcpy(buf, sam->header.data(), sam->header.size());
    return PAV_OK;
}

int pav_sam_fetch(const pav_sam *sam, const pav_sam_cols *c) {
    if (!sam || !c) return PAV_E_ARG;
    uint64_t co = 0, to = 0;
    const size_t n = sam->rows.size();
    for (size_t i = 0; i < n; ++i) {
        const pav_sam::Row &r = sam->rows[i];
        if (c->index) c->index[i] = r.index;
        if (c->pos) c->pos[i] = r.pos;
        if (c->end) c->end[i] = r.end;
        if (c->chrom_id) c->chrom_id[i] = r.chrom_id;
        if (c->qry_id) c->qry_id[i] = r.qry_id;
        if (c->query_alignment_start) c->query_alignment_start[i] = r.qas;
        if (c->query_alignment_end) c->query_alignment_end[i] = r.qae;
        if (c->clip_h) c->clip_h[i] = r.clip_h;
        if (c->tig_map_pos) c->tig_map_pos[i] = r.tig_map_pos;
        if (c->mapq) c->mapq[i] = r.mapq;
        if (c->flag) c->flag[i] = r.flag;
        if (c->has_m) c->has_m[i] = r.has_m;
        if (c->status) c->status[i] = r.status;
        if (c->ref_bp) c->ref_bp[i] = r.ref_bp;
        if (c->tig_bp) c->tig_bp[i] = r.tig_bp;
        if (c->err_kind) c->err_kind[i] = r.err_kind;
        if (c->err_op) c->err_op[i] = r.err_op;
        if (c->err_len) c->err_len[i] = r.err_len;
        if (c->err_char) c->err_char[i] = r.err_char;
        if (c->cigar_off) c->cigar_off[i] = co;
        if (c->cigar_text && !r.cigar.empty()) memcpy(c->cigar_text + co, r.cigar.data(), r.cigar.size());
        co += r.cigar.size();
        if (c->rg_kind) c->rg_kind[i] = r.rg_kind;
        if (c->ao_kind) c->ao_kind[i] = r.ao_kind;
        if (c->rg_off) c->rg_off[i] = to;
        if (c->tag_text && !r.rg.empty()) memcpy(c->tag_text + to, r.rg.data(), r.rg.size());
        to += r.rg.size();
        if (c->ao_off) c->ao_off[i] = to;
        if (c->tag_text && !r.ao.empty()) memcpy(c->tag_text + to, r.ao.data(), r.ao.size());
        to += r.ao.size();
    }
    if (c->cigar_off) c->cigar_off[n] = co;
    if (c->rg_off) c->rg_off[n] = to;
    if (c->ao_off) c->ao_off[n] = to;
    return PAV_OK;
}

}  // extern "C"
