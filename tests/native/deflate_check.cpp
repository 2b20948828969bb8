// Host run of pav_amd/csrc/deflate_dev.h (the serial parts of the device deflate encoder): a scalar encoder built from the same
// functions - one greedy hash match finder, dynamic blocks, stored-block byte alignment between blocks, gzip framing with joined
// CRCs - whose output zlib must inflate back to the input.  Test infrastructure (tests/test_host_deflate.py); the product's
// encoder is the kernel in deflate.hip, which uses these functions for its trees, headers and checksums.
//   deflate_check self                  symbol tables, CRC algebra, length limits, round trips of generated inputs
//   deflate_check file <in> <out.gz>    compress a file (blocks of 64 KiB)
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../pav_amd/csrc/deflate_dev.h"

using namespace pav::dfl;

struct Token { uint32_t lit_len, dist; };       // dist 0: literal

static std::vector<Token> greedy(const uint8_t *p, size_t hist, size_t n) {      // p points at the block; p[-hist..) is history
    std::vector<Token> out;
    std::vector<int64_t> head(1 << 15, -1);
    auto h4 = [&](const uint8_t *q) { uint32_t w; memcpy(&w, q, 4); return (w * 2654435761u) >> 17; };
    const uint8_t *base = p - hist;
    for (size_t i = 0; i + 4 <= hist; ++i) head[h4(base + i)] = (int64_t)i;
    size_t i = 0;
    while (i < n) {
        uint32_t best = 0, bd = 0;
        if (i + 4 <= n) {
            const uint32_t h = h4(p + i);
            const int64_t c = head[h];
            if (c >= 0) {
                const size_t cur = hist + i, d = cur - (size_t)c;
                if (d >= 1 && d <= 32768) {
                    uint32_t l = 0;
                    while (l < 258 && i + l < n && base[(size_t)c + l] == p[i + l]) ++l;
                    if (l >= 4) { best = l; bd = (uint32_t)d; }
                }
            }
            head[h] = (int64_t)(hist + i);
        }
        if (best) { out.push_back(Token{best, bd}); for (size_t k = 1; k < best && i + k + 4 <= n; ++k) head[h4(p + i + k)] = (int64_t)(hist + i + k); i += best; }
        else { out.push_back(Token{p[i], 0}); ++i; }
    }
    return out;
}

struct Bits {
    std::vector<uint8_t> bytes; uint64_t acc = 0; int n = 0;
    void put(uint32_t v, int b) { acc |= (uint64_t)v << n; n += b; while (n >= 8) { bytes.push_back((uint8_t)acc); acc >>= 8; n -= 8; } }
    void align() { if (n) { bytes.push_back((uint8_t)acc); acc = 0; n = 0; } }
};

static void build(const uint32_t *freq_in, int n, int limit, uint8_t *len, uint32_t *code) {
    std::vector<uint32_t> freq(freq_in, freq_in + n);
    int used = 0; for (int s = 0; s < n; ++s) used += freq[s] != 0;
    for (int s = 0; s < n && used < 2; ++s) if (!freq[s]) { freq[s] = 1; ++used; }       // as deflate.hip does
    HuffWork W; W.n_used = (uint32_t)used;
    for (int s = 0; s < n; ++s) if (freq[s]) W.order[huff_rank(freq.data(), n, s)] = (uint16_t)s;
    huff_lengths(freq.data(), n, limit, len, W);
    uint32_t count[MAX_BITS + 2], next[MAX_BITS + 2];
    huff_codes(len, n, code, count, next);
}

static void encode_block(Bits &out, const std::vector<Token> &toks, bool final_block) {
    uint32_t f_ll[N_LL] = {0}, f_d[N_D] = {0};
    for (const Token &t : toks) {
        if (!t.dist) f_ll[t.lit_len]++;
        else { uint32_t c, eb, ev; len_symbol(t.lit_len, c, eb, ev); f_ll[c]++; dist_symbol(t.dist, c, eb, ev); f_d[c]++; }
    }
    f_ll[256]++;
    uint8_t ll_len[N_LL], d_len[N_D]; uint32_t ll_code[N_LL], d_code[N_D];
    build(f_ll, N_LL, MAX_BITS, ll_len, ll_code);
    build(f_d, N_D, MAX_BITS, d_len, d_code);
    std::vector<uint32_t> words(400, 0);
    BitSink sink{words.data(), 0};
    HeaderWork hw;
    block_header(sink, final_block, ll_len, d_len, hw);
    for (uint32_t b = 0; b < sink.n_bits; ++b) out.put((words[b >> 5] >> (b & 31)) & 1u, 1);
    for (const Token &t : toks) {
        if (!t.dist) out.put(ll_code[t.lit_len] & 0xFFFF, (int)(ll_code[t.lit_len] >> 16));
        else {
            uint32_t c, eb, ev;
            len_symbol(t.lit_len, c, eb, ev); out.put(ll_code[c] & 0xFFFF, (int)(ll_code[c] >> 16)); out.put(ev, (int)eb);
            dist_symbol(t.dist, c, eb, ev); out.put(d_code[c] & 0xFFFF, (int)(d_code[c] >> 16)); out.put(ev, (int)eb);
        }
    }
    out.put(ll_code[256] & 0xFFFF, (int)(ll_code[256] >> 16));
    if (final_block) out.align();
    else { out.put(0, 3); out.align(); out.put(0x0000, 16); out.put(0xFFFF, 16); }     // empty stored block: byte alignment
}

static uint32_t crc_bytes(const uint8_t *p, size_t n) {
    static uint32_t tab[256]; static bool init = false;
    if (!init) { for (uint32_t i = 0; i < 256; ++i) tab[i] = crc_table_entry(i); init = true; }
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; ++i) c = tab[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

static std::vector<uint8_t> gzip_of(const std::vector<uint8_t> &text, size_t block, size_t hist_max) {
    Bits out;
    const uint8_t hdr[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 0xff};
    out.bytes.assign(hdr, hdr + 10);
    uint32_t crc = 0;
    const size_t n_blocks = text.empty() ? 1 : (text.size() + block - 1) / block;
    for (size_t b = 0; b < n_blocks; ++b) {
        const size_t a = b * block, e = std::min(text.size(), a + block), hist = std::min(a, hist_max);
        encode_block(out, greedy(text.data() + a, hist, e - a), b + 1 == n_blocks);
        crc = crc_join(crc, crc_bytes(text.data() + a, e - a), e - a);
    }
    for (int k = 0; k < 4; ++k) out.bytes.push_back((uint8_t)(crc >> (8 * k)));
    for (int k = 0; k < 4; ++k) out.bytes.push_back((uint8_t)((uint32_t)text.size() >> (8 * k)));
    return out.bytes;
}

static bool inflate_equals(const std::vector<uint8_t> &gz, const std::vector<uint8_t> &text) {
    z_stream zs; memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, 15 + 16) != Z_OK) return false;
    std::vector<uint8_t> out(text.size() + 64);
    zs.next_in = const_cast<Bytef *>(gz.data()); zs.avail_in = (uInt)gz.size();
    zs.next_out = out.data(); zs.avail_out = (uInt)out.size();
    const int rc = inflate(&zs, Z_FINISH);
    const bool ok = rc == Z_STREAM_END && zs.total_out == text.size() && zs.avail_in == 0 && (text.empty() || memcmp(out.data(), text.data(), text.size()) == 0);
    if (!ok) fprintf(stderr, "inflate: rc %d (%s), %lu of %zu bytes, %u input bytes left\n", rc, zs.msg ? zs.msg : "", zs.total_out, text.size(), zs.avail_in);
    inflateEnd(&zs);
    return ok;
}

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "%s:%d: %s failed\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main(int argc, char **argv) {
    if (argc >= 4 && !strcmp(argv[1], "file")) {
        FILE *fh = fopen(argv[2], "rb"); if (!fh) return 2;
        std::vector<uint8_t> text; uint8_t buf[1 << 16]; size_t r;
        while ((r = fread(buf, 1, sizeof buf, fh)) > 0) text.insert(text.end(), buf, buf + r);
        fclose(fh);
        const std::vector<uint8_t> gz = gzip_of(text, 1 << 16, 8192);
        if (!inflate_equals(gz, text)) return 1;
        fh = fopen(argv[3], "wb"); fwrite(gz.data(), 1, gz.size(), fh); fclose(fh);
        printf("ok %zu -> %zu\n", text.size(), gz.size());
        return 0;
    }
    // ---- symbol tables against the table of RFC 1951 3.2.5 ----
    {
        static const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        for (int i = 0; i < 19; ++i) CHECK(cl_order(i) == (uint32_t)order[i]);
        static const uint16_t lbase[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
        static const uint8_t lext[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
        static const uint16_t dbase[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
        static const uint8_t dext[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
        for (uint32_t len = 3; len <= 258; ++len) {
            uint32_t c, eb, ev; len_symbol(len, c, eb, ev);
            CHECK(c >= 257 && c <= 285 && eb == lext[c - 257] && lbase[c - 257] + ev == len && ev < (1u << eb) + (eb ? 0 : 1));
            if (len == 258) CHECK(c == 285);
        }
        for (uint32_t d = 1; d <= 32768; ++d) {
            uint32_t c, eb, ev; dist_symbol(d, c, eb, ev);
            CHECK(c < 30 && eb == dext[c] && dbase[c] + ev == d && (eb == 0 ? ev == 0 : ev < (1u << eb)));
        }
    }
    // ---- CRC algebra against zlib ----
    {
        std::mt19937_64 rng(11);
        std::vector<uint8_t> a(70001), b(12345);
        for (auto &x : a) x = (uint8_t)rng();
        for (auto &x : b) x = (uint8_t)rng();
        CHECK(crc_bytes(a.data(), a.size()) == crc32(0, a.data(), (uInt)a.size()));
        std::vector<uint8_t> ab(a); ab.insert(ab.end(), b.begin(), b.end());
        CHECK(crc_join(crc_bytes(a.data(), a.size()), crc_bytes(b.data(), b.size()), b.size()) == crc32(0, ab.data(), (uInt)ab.size()));
        CHECK(crc_join(crc_bytes(a.data(), a.size()), 0, 0) == crc_bytes(a.data(), a.size()));
        CHECK(crc_join(0, crc_bytes(b.data(), b.size()), b.size()) == crc_bytes(b.data(), b.size()));
        uint32_t c = 0; size_t at = 0;                                       // many pieces of uneven size, one of them empty
        for (size_t piece : {1u, 0u, 7u, 4096u, 65536u, 333u}) { const size_t e = std::min(ab.size(), at + piece); c = crc_join(c, crc_bytes(ab.data() + at, e - at), e - at); at = e; }
        c = crc_join(c, crc_bytes(ab.data() + at, ab.size() - at), ab.size() - at);
        CHECK(c == crc32(0, ab.data(), (uInt)ab.size()));
    }
    // ---- length limits: Fibonacci counts make the deepest possible tree ----
    {
        for (int n : {2, 3, 19, 30, 40, 64, 200, 286}) {
            std::vector<uint32_t> f((size_t)n);
            uint64_t a = 1, b = 1;
            for (int i = 0; i < n; ++i) { f[(size_t)i] = (uint32_t)std::min<uint64_t>(a, 1u << 30); const uint64_t t = a + b; a = b; b = t; }
            for (int limit : {7, 15}) {
                if (limit == 7 && n > 19) continue;
                uint8_t len[288]; uint32_t code[288];
                build(f.data(), n, limit, len, code);
                uint64_t kraft = 0;
                for (int s = 0; s < n; ++s) { CHECK(len[s] >= 1 && len[s] <= limit); kraft += 1ull << (limit - len[s]); }
                CHECK(kraft == 1ull << limit);                                  // complete: what inflate demands of the two main trees
                for (int s = 1; s < n; ++s) CHECK(len[s] <= len[s - 1]);        // rarer symbols never get shorter codes
                for (int s = 0; s < n; ++s) for (int t = 0; t < s; ++t) {       // prefix-free (codes are stored reversed)
                    const int ls = len[s], lt = len[t], m = std::min(ls, lt);
                    CHECK((bit_reverse(code[s] & 0xFFFF, ls) >> (ls - m)) != (bit_reverse(code[t] & 0xFFFF, lt) >> (lt - m)));
                }
            }
        }
        uint32_t f1[N_D] = {0}; f1[7] = 5;                                     // one used symbol: a second one is forced in
        uint8_t len[N_D]; uint32_t code[N_D];
        build(f1, N_D, MAX_BITS, len, code);
        int used = 0; for (int s = 0; s < N_D; ++s) used += len[s] != 0;
        CHECK(used == 2 && len[7] == 1);
    }
    // ---- round trips ----
    {
        std::mt19937_64 rng(3);
        auto trip = [&](const std::vector<uint8_t> &text, size_t block, size_t hist) { return inflate_equals(gzip_of(text, block, hist), text); };
        CHECK(trip({}, 65536, 8192));
        CHECK(trip({'a'}, 65536, 8192));
        CHECK(trip(std::vector<uint8_t>(100000, 'x'), 65536, 8192));          // one distance code, length 258 matches
        CHECK(trip(std::vector<uint8_t>(5, 0), 65536, 8192));
        for (int kind = 0; kind < 6; ++kind)
            for (size_t n : {1u, 2u, 3u, 4u, 5u, 63u, 64u, 65u, 1000u, 65535u, 65536u, 65537u, 200000u}) {
                std::vector<uint8_t> t(n);
                for (size_t i = 0; i < n; ++i) {
                    switch (kind) {
                        case 0: t[i] = (uint8_t)rng(); break;                                              // incompressible
                        case 1: t[i] = "ACGT"[rng() & 3]; break;
                        case 2: t[i] = (uint8_t)('0' + rng() % 10); break;
                        case 3: t[i] = (uint8_t)(i % 97 < 90 ? "chr1\t123456\tSNV\tA\tG\tPASS\n"[i % 26] : '0' + rng() % 10); break;   // rows
                        case 4: t[i] = (uint8_t)(rng() % 1000 ? 'a' : rng()); break;                       // skewed counts
                        default: t[i] = (uint8_t)(i & 0xFF); break;
                    }
                }
                CHECK(trip(t, 65536, 8192));
                CHECK(trip(t, 4096, 32768));
                CHECK(trip(t, 1000, 0));
            }
    }
    printf("ok self\n");
    return 0;
}
