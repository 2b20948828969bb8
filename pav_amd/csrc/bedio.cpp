// Native reader of PAV alignment tables (API_ALIGN.md:31-64; what rule call_cigar reads with pandas,
// rules/call.snakefile:805,813-816): gzip / plain TSV -> columns + one CIGAR text block, and a direct hand-off of the rows of
// one CALL_BATCH to the variant caller (reader half).  Host code; no GPU needed to parse.
#include "common.h"

#include <zlib.h>

#include <algorithm>
#include <string>
#include <unordered_map>

struct pav_bed {
    uint64_t n = 0;
    std::vector<int64_t> pos, end, index, qry_pos, qry_end, qry_len, mapq, call_batch;
    std::vector<uint8_t> rev;
    std::vector<uint32_t> chrom_id, qry_id;
    std::vector<std::string> chrom_names, qry_names;          // in order of first appearance
    std::string cigar;
    std::vector<uint64_t> cigar_off;
    uint32_t have = 0;                                        // bit per known column present in the header
};

namespace pav {

const std::vector<std::string> &seq_names(pav_ctx *ctx, int role);   // invscan.cpp

namespace {

enum Col { C_CHROM, C_POS, C_END, C_INDEX, C_QRY_ID, C_QRY_POS, C_QRY_END, C_QRY_LEN, C_MAPQ, C_REV, C_CIGAR, C_CALL_BATCH, C_N };
const char *COL_NAMES[C_N] = {"#CHROM", "POS", "END", "INDEX", "QRY_ID", "QRY_POS", "QRY_END", "QRY_LEN", "MAPQ", "REV", "CIGAR", "CALL_BATCH"};

bool read_all(const char *path, std::string &out, std::string &err) {
    gzFile f = gzopen(path, "rb");                              // transparent for uncompressed files
    if (!f) { err = std::string("cannot open ") + path; return false; }
    (void)gzbuffer(f, 1 << 20);
    out.clear();
    std::vector<char> buf(8 << 20);
    for (;;) {
        const int n = gzread(f, buf.data(), (unsigned)buf.size());
        if (n < 0) { int e = 0; err = std::string("read error in ") + path + ": " + gzerror(f, &e); gzclose(f); return false; }
        if (n == 0) break;
        out.append(buf.data(), (size_t)n);
    }
    gzclose(f);
    return true;
}

bool parse_i64(const char *b, const char *e, int64_t &v) {
    if (b == e) return false;
    bool neg = false;
    if (*b == '-') { neg = true; ++b; if (b == e) return false; }
    int64_t x = 0;
    for (; b < e; ++b) {
        if (*b < '0' || *b > '9') return false;
        if (x > (INT64_MAX - 9) / 10) return false;                    // does not fit: refused like any other non-number
        x = x * 10 + (*b - '0');
    }
    v = neg ? -x : x;
    return true;
}

uint32_t intern(std::unordered_map<std::string, uint32_t> &map, std::vector<std::string> &names, const char *b, const char *e) {
    std::string s(b, e);
    auto it = map.find(s);
    if (it != map.end()) return it->second;
    const uint32_t id = (uint32_t)names.size();
    names.push_back(s);
    map.emplace(std::move(s), id);
    return id;
}

}  // namespace
}  // namespace pav

using namespace pav;

extern "C" {

int pav_bed_open(const char *path, int with_cigar, pav_bed **out) {
    if (!path || !out) return PAV_E_ARG;
    *out = nullptr;
    std::string text, err;
    if (!read_all(path, text, err)) return fail(nullptr, PAV_E_ARG, "pav_bed_open: %s", err.c_str());
    auto bed = new pav_bed();
    const char *p = text.data(), *end = p + text.size();
    // header
    const char *eol = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
    if (!eol) eol = end;
    int col_of[64];
    int n_cols = 0;
    std::fill(col_of, col_of + 64, -1);
    for (const char *b = p; b <= eol && n_cols < 64;) {
        const char *t = static_cast<const char *>(memchr(b, '\t', (size_t)(eol - b)));
        if (!t) t = eol;
        const char *fe = t;
        if (fe > b && fe[-1] == '\r') --fe;
        for (int c = 0; c < C_N; ++c)
            if ((size_t)(fe - b) == strlen(COL_NAMES[c]) && memcmp(b, COL_NAMES[c], (size_t)(fe - b)) == 0) { col_of[n_cols] = c; bed->have |= 1u << c; }
        ++n_cols;
        if (t == eol) break;
        b = t + 1;
    }
    if (with_cigar && !(bed->have & (1u << C_CIGAR))) { delete bed; return fail(nullptr, PAV_E_ARG, "pav_bed_open: %s has no CIGAR column", path); }
    std::unordered_map<std::string, uint32_t> chrom_map, qry_map;
    bed->cigar_off.push_back(0);
    uint64_t line_no = 1;
    for (p = eol < end ? eol + 1 : end; p < end;) {
        eol = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
        if (!eol) eol = end;
        ++line_no;
        if (eol == p) { p = eol + 1; continue; }               // blank line (pandas skips them)
        int c = 0;
        for (const char *b = p; c < n_cols; ++c) {
            const char *t = static_cast<const char *>(memchr(b, '\t', (size_t)(eol - b)));
            if (!t) t = eol;
            const char *fe = t;
            if (t == eol && fe > b && fe[-1] == '\r') --fe;
            const int k = col_of[c];
            int64_t v = 0;
            bool ok = true;
            switch (k) {
                case C_CHROM: bed->chrom_id.push_back(intern(chrom_map, bed->chrom_names, b, fe)); break;
                case C_QRY_ID: bed->qry_id.push_back(intern(qry_map, bed->qry_names, b, fe)); break;
                case C_POS: ok = parse_i64(b, fe, v); bed->pos.push_back(v); break;
                case C_END: ok = parse_i64(b, fe, v); bed->end.push_back(v); break;
                case C_INDEX: ok = parse_i64(b, fe, v); bed->index.push_back(v); break;
                case C_QRY_POS: ok = parse_i64(b, fe, v); bed->qry_pos.push_back(v); break;
                case C_QRY_END: ok = parse_i64(b, fe, v); bed->qry_end.push_back(v); break;
                case C_QRY_LEN: ok = parse_i64(b, fe, v); bed->qry_len.push_back(v); break;
                case C_MAPQ: ok = parse_i64(b, fe, v); bed->mapq.push_back(v); break;
                case C_CALL_BATCH: ok = parse_i64(b, fe, v); bed->call_batch.push_back(v); break;
                case C_REV:
                    if (fe - b == 4 && memcmp(b, "True", 4) == 0) bed->rev.push_back(1);
                    else if (fe - b == 5 && memcmp(b, "False", 5) == 0) bed->rev.push_back(0);
                    else ok = false;
                    break;
                case C_CIGAR:
                    if (with_cigar) { bed->cigar.append(b, fe); bed->cigar_off.push_back(bed->cigar.size()); }
                    break;
                default: break;
            }
            if (!ok) {
                const std::string bad(b, fe);
                delete bed;
                return fail(nullptr, PAV_E_ARG, "pav_bed_open: %s line %llu: cannot parse '%s' in column %s", path, (unsigned long long)line_no,
                            bad.c_str(), COL_NAMES[k]);
            }
            if (t == eol) { ++c; break; }
            b = t + 1;
        }
        if (c != n_cols) { delete bed; return fail(nullptr, PAV_E_ARG, "pav_bed_open: %s line %llu has %d fields, the header %d", path, (unsigned long long)line_no, c, n_cols); }
        ++bed->n;
        p = eol < end ? eol + 1 : end;
    }
    *out = bed;
    return PAV_OK;
}

void pav_bed_close(pav_bed *bed) { delete bed; }

int pav_bed_info(const pav_bed *bed, pav_bed_info_t *info) {
    if (!bed || !info) return PAV_E_ARG;
    info->n_rows = bed->n;
    info->cigar_bytes = bed->cigar.size();
    info->n_chrom = (uint32_t)bed->chrom_names.size();
    info->n_qry = (uint32_t)bed->qry_names.size();
    info->columns = bed->have;
    info->pad = 0;
    return PAV_OK;
}

const char *pav_bed_name(const pav_bed *bed, int which, uint32_t id) {
    if (!bed) return nullptr;
    const std::vector<std::string> &v = which == 0 ? bed->chrom_names : bed->qry_names;
    return id < v.size() ? v[id].c_str() : nullptr;
}

int pav_bed_fetch(const pav_bed *bed, const pav_bed_cols *c) {
    if (!bed || !c) return PAV_E_ARG;
    auto put = [](void *dst, const auto &src) { if (dst && !src.empty()) memcpy(dst, src.data(), sizeof(src[0]) * src.size()); };
    put(c->chrom_id, bed->chrom_id); put(c->qry_id, bed->qry_id);
    put(c->pos, bed->pos); put(c->end, bed->end); put(c->index, bed->index); put(c->qry_pos, bed->qry_pos); put(c->qry_end, bed->qry_end);
    put(c->qry_len, bed->qry_len); put(c->mapq, bed->mapq); put(c->call_batch, bed->call_batch); put(c->rev, bed->rev);
    if (c->cigar_text && !bed->cigar.empty()) memcpy(c->cigar_text, bed->cigar.data(), bed->cigar.size());
    if (c->cigar_off) memcpy(c->cigar_off, bed->cigar_off.data(), sizeof(uint64_t) * bed->cigar_off.size());
    return PAV_OK;
}

int pav_cigar_load_bed(pav_ctx *ctx, const pav_bed *bed, int64_t call_batch, uint32_t *n_rows, int64_t *index_out) {
    if (!ctx || !bed) return PAV_E_ARG;
    const uint32_t need = (1u << C_CHROM) | (1u << C_POS) | (1u << C_QRY_ID) | (1u << C_REV) | (1u << C_CIGAR);
    if ((bed->have & need) != need || bed->cigar_off.size() != bed->n + 1)
        return fail(ctx, PAV_E_ARG, "pav_cigar_load_bed: the table lacks #CHROM / POS / QRY_ID / REV / CIGAR (open it with with_cigar)");
    if (call_batch >= 0 && !(bed->have & (1u << C_CALL_BATCH))) return fail(ctx, PAV_E_ARG, "pav_cigar_load_bed: the table has no CALL_BATCH column");
    const std::vector<std::string> &rn = seq_names(ctx, PAV_ROLE_REF), &tn = seq_names(ctx, PAV_ROLE_TIG);
    std::unordered_map<std::string, uint32_t> rmap, tmap;
    for (uint32_t i = 0; i < rn.size(); ++i) rmap.emplace(rn[i], i);
    for (uint32_t i = 0; i < tn.size(); ++i) tmap.emplace(tn[i], i);
    std::vector<int64_t> chrom_to(bed->chrom_names.size(), -1), qry_to(bed->qry_names.size(), -1);
    for (size_t i = 0; i < chrom_to.size(); ++i) { auto it = rmap.find(bed->chrom_names[i]); if (it != rmap.end()) chrom_to[i] = it->second; }
    for (size_t i = 0; i < qry_to.size(); ++i) { auto it = tmap.find(bed->qry_names[i]); if (it != tmap.end()) qry_to[i] = it->second; }
    std::vector<pav_aln> aln;
    std::vector<uint64_t> off{0};
    std::string text;
    uint32_t k = 0;
    for (uint64_t i = 0; i < bed->n; ++i) {
        if (call_batch >= 0 && bed->call_batch[i] != call_batch) continue;
        const int64_t r = chrom_to[bed->chrom_id[i]], t = qry_to[bed->qry_id[i]];
        if (r < 0) return fail(ctx, PAV_E_ARG, "pav_cigar_load_bed: reference sequence '%s' is not loaded", bed->chrom_names[bed->chrom_id[i]].c_str());
        if (t < 0) return fail(ctx, PAV_E_ARG, "pav_cigar_load_bed: contig '%s' is not loaded", bed->qry_names[bed->qry_id[i]].c_str());
        if (bed->pos[i] < 0 || bed->pos[i] > 0xffffffffll) return fail(ctx, PAV_E_LIMIT, "pav_cigar_load_bed: POS out of range in row %llu", (unsigned long long)i);
        aln.push_back(pav_aln{(uint32_t)r, (uint32_t)t, (uint32_t)bed->pos[i], bed->rev[i]});
        text.append(bed->cigar, bed->cigar_off[i], bed->cigar_off[i + 1] - bed->cigar_off[i]);
        off.push_back(text.size());
        if (index_out && (bed->have & (1u << C_INDEX))) index_out[k] = bed->index[i];
        ++k;
    }
    if (n_rows) *n_rows = k;
    return pav_cigar_load(ctx, k, aln.data(), reinterpret_cast<const uint8_t *>(text.data()), off.data());
}

}  // extern "C"
