// Host trial of pav_amd/csrc/inflate_dev.h (test infrastructure): the lane-serial token decoder of the device inflate, run here
// with plain arrays for its tables, its tokens resolved by a scalar loop, against zlib's own inflate of the same raw deflate stream.
//   inflate_check self              streams made here by zlib at every level / strategy, stored and fixed blocks, corrupt streams
//   inflate_check file IN OUT LEVEL text file -> raw deflate by zlib -> tokens -> text written to OUT (compared by the caller)
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <random>
#include <string>
#include <vector>

#include <zlib.h>

#include "../../pav_amd/csrc/deflate_dev.h"
#include "../../pav_amd/csrc/inflate_dev.h"

using namespace pav::ifl;

#define CHECK(x) do { if (!(x)) { fprintf(stderr, "CHECK failed at line %d: %s\n", __LINE__, #x); return 1; } } while (0)

struct HostTab {
    uint16_t l[LIT_TAB], ll[LIT_LONG];
    uint8_t d[DIST_TAB], ds[DIST_SYMS];
    uint16_t lit(uint32_t e) const { return l[e]; }
    uint16_t lit_long(uint32_t i) const { return ll[i]; }
    uint8_t dist(uint32_t e) const { return d[e]; }
    uint8_t dist_sym(uint32_t i) const { return ds[i]; }
    void set_lit(uint32_t e, uint16_t v) { l[e] = v; }
    void set_lit_long(uint32_t i, uint16_t v) { ll[i] = v; }
    void set_dist(uint32_t e, uint8_t v) { d[e] = v; }
    void set_dist_sym(uint32_t i, uint8_t v) { ds[i] = v; }
};

static std::vector<uint8_t> raw_deflate(const std::vector<uint8_t> &text, int level, int strategy, int mem_level = 8) {
    z_stream z; memset(&z, 0, sizeof z);
    deflateInit2(&z, level, Z_DEFLATED, -15, mem_level, strategy);
    std::vector<uint8_t> out(deflateBound(&z, (uLong)text.size()) + 64);
    z.next_in = const_cast<Bytef *>(text.data()); z.avail_in = (uInt)text.size();
    z.next_out = out.data(); z.avail_out = (uInt)out.size();
    deflate(&z, Z_FINISH);
    out.resize(z.total_out);
    deflateEnd(&z);
    return out;
}

// tokens -> text, the obvious way
static bool resolve(const std::vector<uint32_t> &tok, uint32_t n, std::vector<uint8_t> &text) {
    text.clear();
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t t = tok[i];
        if (t & 3u) { for (uint32_t k = 0; k < (t & 3u); ++k) text.push_back((uint8_t)(t >> (8 + 8 * k))); }
        else {
            const uint32_t len = tok_bytes(t), dist = tok_dist(t);
            if (dist == 0 || dist > text.size() || len < 3 || len > 258) return false;
            for (uint32_t k = 0; k < len; ++k) text.push_back(text[text.size() - dist]);
        }
    }
    return true;
}

static int decode(const std::vector<uint8_t> &stream, uint32_t text_len, std::vector<uint8_t> &text, uint32_t *n_tok_out = nullptr,
                  std::vector<uint32_t> *tok_out = nullptr, uint8_t pad = 0) {
    std::vector<uint8_t> in(stream); in.resize(in.size() + 16, pad);         // (the reader looks a few bytes beyond the stream: 12 at most)
    std::vector<uint32_t> tok(tok_capacity(text_len));
    HostTab T; LaneScratch S; memset(&T, 0, sizeof T); memset(&S, 0, sizeof S);
    uint32_t n = 0;
    const int rc = inflate_tokens(in.data(), (uint32_t)stream.size(), text_len, tok.data(), (uint32_t)tok.size(), &n, T, &S);
    if (rc != IFL_OK) return rc;
    if (n > tok.size()) return 100;
    if (!resolve(tok, n, text)) return 101;
    if (n_tok_out) *n_tok_out = n;
    if (tok_out) *tok_out = tok;
    return 0;
}

static std::vector<uint8_t> dna(std::mt19937_64 &rng, size_t n, int line) {
    std::vector<uint8_t> t; t.reserve(n + n / 60 + 64);
    static const char *A = "ACGTacgtN";
    size_t col = 0;
    std::vector<uint8_t> unit;
    while (t.size() < n) {
        const uint64_t r = rng();
        uint8_t c;
        if ((r & 0xFFFF) < 40 && !unit.empty()) {                            // a tandem repeat, a run of N, a soft-masked stretch
            const size_t rep = 20 + (r >> 16) % 600;
            for (size_t k = 0; k < rep && t.size() < n; ++k) { t.push_back(unit[k % unit.size()]); if (++col == (size_t)line) { t.push_back('\n'); col = 0; } }
            unit.clear();
            continue;
        }
        c = (uint8_t)A[(r >> 20) % ((r & 0x3FF) < 6 ? 9 : 4)];
        unit.push_back(c); if (unit.size() > 1 + (r >> 40) % 12) unit.erase(unit.begin());
        t.push_back(c);
        if (++col == (size_t)line) { t.push_back('\n'); col = 0; }
    }
    t.resize(n);
    return t;
}

int main(int argc, char **argv) {
    if (argc >= 5 && !strcmp(argv[1], "file")) {
        FILE *fh = fopen(argv[2], "rb"); if (!fh) return 2;
        std::vector<uint8_t> text; uint8_t buf[1 << 16]; size_t r;
        while ((r = fread(buf, 1, sizeof buf, fh)) > 0) text.insert(text.end(), buf, buf + r);
        fclose(fh);
        fh = fopen(argv[3], "wb"); if (!fh) return 2;
        size_t tokens = 0, literals = 0, copies = 0;
        for (size_t at = 0; at < text.size(); at += 65280) {                 // BGZF-sized members
            std::vector<uint8_t> part(text.begin() + (long)at, text.begin() + (long)std::min(text.size(), at + 65280));
            std::vector<uint8_t> back; uint32_t n = 0;
            std::vector<uint32_t> toks;
            const int rc = decode(raw_deflate(part, atoi(argv[4]), Z_DEFAULT_STRATEGY), (uint32_t)part.size(), back, &n, &toks);
            if (rc) { fprintf(stderr, "member at %zu: error %d\n", at, rc); return 1; }
            for (uint32_t i = 0; i < n; ++i) { if (toks[i] & 3u) literals += toks[i] & 3u; else ++copies; }
            fwrite(back.data(), 1, back.size(), fh);
            tokens += n;
        }
        fclose(fh);
        printf("ok %zu bytes, %zu tokens: %zu literals, %zu copies\n", text.size(), tokens, literals, copies);
        return 0;
    }
    std::mt19937_64 rng(5);
    // ---- length / distance symbols against the table of RFC 1951 3.2.5 (the decoder computes them) ----
    {
        static const uint16_t lbase[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
        static const uint8_t lext[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
        static const uint16_t dbase[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
        static const uint8_t dext[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
        for (uint32_t s = 0; s < 29; ++s) {
            uint32_t len, eb;
            if (s < 8) { len = 3 + s; eb = 0; } else if (s == 28) { len = 258; eb = 0; } else { eb = (s >> 2) - 1; len = 3 + ((4 + (s & 3)) << eb); }
            CHECK(len == lbase[s] && eb == lext[s]);
        }
        for (uint32_t d = 0; d < 30; ++d) {
            uint32_t dist, eb;
            if (d < 4) { dist = 1 + d; eb = 0; } else { eb = (d >> 1) - 1; dist = 1 + ((2 + (d & 1)) << eb); }
            CHECK(dist == dbase[d] && eb == dext[d]);
        }
        CHECK(tok_bytes(tok_match(258, 32768)) == 258 && tok_dist(tok_match(258, 32768)) == 32768);
        CHECK(tok_bytes(tok_match(3, 1)) == 3 && tok_dist(tok_match(3, 1)) == 1);
    }
    // ---- the checksum as k_inflate_resolve takes it: zeros in front of the text up to 256 L bytes, 256 pieces of L bytes run from a
    //      zero register, joined in pairs by x^(8 L), x^(16 L), ...; the initial value's term all-ones * x^(8 n) and the final inversion ----
    {
        using namespace pav::dfl;
        uint32_t tab[256]; for (uint32_t i = 0; i < 256; ++i) tab[i] = crc_table_entry(i);
        for (uint32_t n : {1u, 2u, 255u, 256u, 257u, 4097u, 65279u, 65280u, 65535u, 65536u, 31337u}) {
            std::vector<uint8_t> t(n); for (auto &c : t) c = (uint8_t)rng();
            const uint32_t L = (n + 255) / 256, pad = 256 * L - n;
            std::vector<uint32_t> v(256);
            for (uint32_t p = 0; p < 256; ++p) {
                uint32_t r = 0;
                for (uint32_t i = 0; i < L; ++i) { const uint32_t q = p * L + i; const uint32_t b = q >= pad ? t[q - pad] : 0u; r = tab[(r ^ b) & 0xFF] ^ (r >> 8); }
                v[p] = r;
            }
            uint32_t f = gf_xpow8(L);
            for (uint32_t width = 256; width > 1; width /= 2, f = gf_mul(f, f))
                for (uint32_t p = 0; p < width / 2; ++p) v[p] = gf_mul(f, v[2 * p]) ^ v[2 * p + 1];
            const uint32_t crc = v[0] ^ gf_mul(0xFFFFFFFFu, gf_xpow8(n)) ^ 0xFFFFFFFFu;
            CHECK(crc == crc32(0, t.data(), (uInt)n));
        }
    }
    // ---- round trips: every level and strategy zlib has, the texts the loader meets ----
    {
        std::vector<std::vector<uint8_t>> texts;
        texts.push_back({});                                                 // an empty member (the BGZF end-of-file marker)
        texts.push_back({'A'});
        texts.push_back({'A', 'C', 'G'});
        texts.push_back(std::vector<uint8_t>(65280, 'N'));                   // one long run: copies at distance 1
        texts.push_back(std::vector<uint8_t>(65536, 0));
        texts.push_back(dna(rng, 65280, 60));
        texts.push_back(dna(rng, 65280, 80));
        texts.push_back(dna(rng, 40000, 0));
        texts.push_back(dna(rng, 777, 70));
        { std::vector<uint8_t> t(65536); for (auto &c : t) c = (uint8_t)rng(); texts.push_back(t); }     // incompressible: stored blocks
        { std::vector<uint8_t> t(65536); for (size_t i = 0; i < t.size(); ++i) t[i] = (uint8_t)(i * 7 + (i >> 8)); texts.push_back(t); }
        { std::string s; for (int i = 0; s.size() < 60000; ++i) s += ">tig" + std::to_string(i) + " len=" + std::to_string(i * 977) + "\nACGTTGCA\n"; texts.push_back(std::vector<uint8_t>(s.begin(), s.end())); }
        { std::vector<uint8_t> t(65536); for (size_t i = 0; i < t.size(); ++i) t[i] = (uint8_t)(rng() % 200 < 199 ? 'a' + rng() % 3 : rng()); texts.push_back(t); }   // long codes for the rare bytes
        { std::vector<uint8_t> t; uint32_t a = 1, b = 1;                      // counts that grow like Fibonacci numbers: codes of every length up to 15 bits
          for (int i = 0; i < 24; ++i) { t.insert(t.end(), a, (uint8_t)(33 + i)); const uint32_t c = a + b; a = b; b = c; }
          std::shuffle(t.begin(), t.end(), rng); t.resize(65000); texts.push_back(t); }
        const int strategies[] = {Z_DEFAULT_STRATEGY, Z_FILTERED, Z_HUFFMAN_ONLY, Z_RLE, Z_FIXED};
        for (const auto &t : texts) for (int level = 0; level <= 9; ++level) for (int st : strategies) {
            if (level == 0 && st != Z_DEFAULT_STRATEGY) continue;
            const std::vector<uint8_t> z = raw_deflate(t, level, st, (level & 1) ? 8 : 3);     // (a small memLevel: many blocks per member)
            std::vector<uint8_t> back;
            const int rc = decode(z, (uint32_t)t.size(), back);
            if (rc || back != t) { fprintf(stderr, "round trip failed: text of %zu bytes, level %d, strategy %d: rc %d\n", t.size(), level, st, rc); return 1; }
        }
    }
    // ---- streams that are not what they say ----
    {
        const std::vector<uint8_t> t = dna(rng, 30000, 60);
        const std::vector<uint8_t> z = raw_deflate(t, 6, Z_DEFAULT_STRATEGY);
        std::vector<uint8_t> back;
        CHECK(decode(z, (uint32_t)t.size() + 1, back) == IFL_E_TEXT);         // ISIZE too large
        CHECK(decode(z, (uint32_t)t.size() - 1, back) == IFL_E_TEXT);         // too small
        { std::vector<uint8_t> cut(z.begin(), z.begin() + (long)z.size() / 2); const int rc = decode(cut, (uint32_t)t.size(), back); CHECK(rc != 0); }
        { std::vector<uint8_t> bad = {0x07}; CHECK(decode(bad, 0, back) == IFL_E_BTYPE); }      // final block of type 3
        { std::vector<uint8_t> bad = {0x01, 0x05, 0x00, 0x00, 0x00, 'a'}; CHECK(decode(bad, 5, back) == IFL_E_STORED); }
        { std::vector<uint8_t> bad = {0x01, 0x05, 0x00, 0xFA, 0xFF, 'a'}; CHECK(decode(bad, 5, back) == IFL_E_INPUT); }
        // a last member whose bits go on as literals for ever (fixed block, then 1-bits = literal 255 on and on; the bytes behind the
        // member look the same): the reader must stop at the member's end, not at ISIZE (ASan is the judge: 16 bytes of padding)
        { std::vector<uint8_t> bad(10, 0xFF); bad[0] = 0x03 | 0xF8; CHECK(decode(bad, 65536, back, nullptr, nullptr, 0xFF) == IFL_E_INPUT); }
        { std::vector<uint8_t> bad(3000, 0xFF); bad[0] = 0x03 | 0xF8; CHECK(decode(bad, 65536, back, nullptr, nullptr, 0xFF) == IFL_E_INPUT); }
        // a copy that reaches in front of the text: fixed block, length 3 at distance 1 as the first symbol (257 = 0000001, distance 0 = 00000)
        { std::vector<uint8_t> bad = {0x03 | (0x00 << 3), 0x02, 0x00, 0x00}; const int rc = decode(bad, 3, back); CHECK(rc == IFL_E_DISTANCE); }
        // every single-bit corruption of a short stream ends in an error or in some text - never out of bounds (ASan is the judge)
        const std::vector<uint8_t> s = dna(rng, 3000, 60);
        const std::vector<uint8_t> zs = raw_deflate(s, 6, Z_DEFAULT_STRATEGY);
        for (size_t bit = 0; bit < zs.size() * 8; ++bit) {
            std::vector<uint8_t> c(zs); c[bit >> 3] ^= (uint8_t)(1u << (bit & 7));
            (void)decode(c, (uint32_t)s.size(), back);
        }
        for (int trial = 0; trial < 2000; ++trial) {
            std::vector<uint8_t> c(64 + rng() % 400); for (auto &x : c) x = (uint8_t)rng();
            (void)decode(c, (uint32_t)(rng() % 65537), back);
        }
    }
    printf("ok self\n");
    return 0;
}
