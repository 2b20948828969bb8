// Text / gzip helpers shared by the table writers (tables.hip: rule call_cigar; invscan.cpp: density tables of rule
// call_inv_batch).  Everything here reproduces what pandas.DataFrame.to_csv(sep='\t', index=False) writes.
#pragma once

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <cmath>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

namespace pav {

// ---- text ---------------------------------------------------------------------------------------------------
static inline void put_u64(std::string &s, uint64_t v) {
    char b[24]; int n = 0;
    do { b[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) s.push_back(b[--n]);
}
static inline void put_i64(std::string &s, int64_t v) { if (v < 0) { s.push_back('-'); put_u64(s, 0 - (uint64_t)v); } else put_u64(s, (uint64_t)v); }
// csv.QUOTE_MINIMAL with delimiter '\t' and quotechar '"' (what DataFrame.to_csv uses)
[[maybe_unused]] static std::string csv_field(const std::string &f) {
    if (f.find_first_of("\t\"\n\r") == std::string::npos) return f;
    std::string q = "\"";
    for (char c : f) { if (c == '"') q += '"'; q += c; }
    return q + "\"";
}

static bool gz_member(const std::string &in, int level, std::string &out) {
    z_stream zs; memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, level, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    out.resize(deflateBound(&zs, (uLong)in.size()) + 64);
    zs.next_in = (Bytef *)in.data(); zs.avail_in = (uInt)in.size();
    zs.next_out = (Bytef *)&out[0]; zs.avail_out = (uInt)out.size();
    const int rc = deflate(&zs, Z_FINISH);
    out.resize(zs.total_out);
    deflateEnd(&zs);
    return rc == Z_STREAM_END;
}

// Format `n_rows` rows in chunks on `threads` workers, compress each chunk as one gzip member when the name ends in
// ".gz", and write the chunks in order.
template <class RowFn>
static int write_table(pav_ctx *ctx, const char *path, const std::string &header, uint64_t n_rows, int threads, int level, RowFn row) {
    const std::string p(path);
    const bool gz = p.size() > 3 && p.compare(p.size() - 3, 3, ".gz") == 0;
    const uint64_t chunk_rows = 1 << 16;
    const uint64_t n_chunks = std::max<uint64_t>(1, (n_rows + chunk_rows - 1) / chunk_rows);
    std::vector<std::string> done(n_chunks);
    std::atomic<uint64_t> next{0};
    std::atomic<bool> ok{true};
    auto work = [&]() {
        std::string text;
        for (uint64_t c; (c = next.fetch_add(1)) < n_chunks;) {
            text.clear();
            if (c == 0) text = header;
            const uint64_t a = c * chunk_rows, b = std::min(n_rows, a + chunk_rows);
            text.reserve((size_t)(b - a) * 160 + header.size());
            for (uint64_t i = a; i < b; ++i) row(i, text);
            if (gz) { if (!gz_member(text, level, done[c])) ok = false; } else done[c].swap(text);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    if (!ok) return fail(ctx, PAV_E_ARG, "table writer: zlib failed for %s", path);
    FILE *fh = fopen(path, "wb");
    if (!fh) return fail(ctx, PAV_E_ARG, "table writer: cannot open %s", path);
    for (const std::string &s : done) if (!s.empty() && fwrite(s.data(), 1, s.size(), fh) != s.size()) { fclose(fh); return fail(ctx, PAV_E_ARG, "table writer: short write to %s", path); }
    fclose(fh);
    return PAV_OK;
}


// repr(float) as CPython / numpy print a float64 (what to_csv writes for a float column): the shortest digit string that
// round-trips, positional when -4 < decimal exponent <= 16 (with ".0" for integers), else d.ddde[+-]XX; NaN -> "" (na_rep).
static inline void put_f64_repr(std::string &s, double v) {
    if (std::isnan(v)) return;
    if (std::isinf(v)) { s += v < 0 ? "-inf" : "inf"; return; }
    char buf[40];
    const auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::scientific);     // shortest round-trip digits
    const char *p = buf, *end = r.ptr;
    if (*p == '-') { s.push_back('-'); ++p; }
    char digits[24]; int nd = 0;
    while (p < end && *p != 'e') { if (*p != '.') digits[nd++] = *p; ++p; }
    int exp10 = 0; bool neg = false;
    if (p < end) { ++p; if (*p == '-') { neg = true; ++p; } else if (*p == '+') ++p; for (; p < end; ++p) exp10 = exp10 * 10 + (*p - '0'); }
    if (neg) exp10 = -exp10;
    const int decpt = exp10 + 1;
    if (decpt > -4 && decpt <= 16) {
        if (decpt <= 0) { s += "0."; s.append((size_t)(-decpt), '0'); s.append(digits, (size_t)nd); }
        else if (decpt >= nd) { s.append(digits, (size_t)nd); s.append((size_t)(decpt - nd), '0'); s += ".0"; }
        else { s.append(digits, (size_t)decpt); s.push_back('.'); s.append(digits + decpt, (size_t)(nd - decpt)); }
    } else {
        s.push_back(digits[0]);
        if (nd > 1) { s.push_back('.'); s.append(digits + 1, (size_t)(nd - 1)); }
        s.push_back('e');
        int e = decpt - 1;
        s.push_back(e < 0 ? '-' : '+');
        if (e < 0) e = -e;
        if (e < 10) s.push_back('0');
        put_u64(s, (uint64_t)e);
    }
}

}  // namespace pav
