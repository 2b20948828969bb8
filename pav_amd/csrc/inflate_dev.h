// Serial half of the DEVICE inflate (inflate.hip): one deflate stream (RFC 1951) -> a list of tokens, decoded by ONE lane.
// PAV keeps its FASTA files bgzipped - rule call_cigar reads `temp/{asm_name}/align/contigs_{hap}.fa.gz` and `data/ref/ref.fa.gz`
// through pysam.FastaFile (rules/call.snakefile:796, pavlib/cigarcall.py:59-64) - and BGZF (SAM specification 4.1) is a series of
// independent gzip members of at most 64 KiB of text: a lane per member walks the Huffman codes (nothing in a deflate stream says
// where a symbol starts but the symbol before it), a wave per member then resolves the tokens' copies in LDS (inflate.hip).
//
// Everything here is __host__ __device__ and free of wave intrinsics; the tables a lane decodes with are reached through a small
// accessor type (LDS, interleaved over the lanes of a wave on the device; plain arrays in tests/native/inflate_check.cpp, which runs
// this code on the host against zlib's inflate - test infrastructure).
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#define PAV_IHD __host__ __device__ __forceinline__
#define PAV_IUNROLL _Pragma("unroll")
#else
#define PAV_IHD inline
#define PAV_IUNROLL
#endif

namespace pav {
namespace ifl {

constexpr int LIT_BITS = 8;          // first-level table of the literal / length code: 2^8 entries of (symbol << 4 | code length), 16 bits each
constexpr int DIST_BITS = 7;         // of the distance code: 2^7 entries of (symbol << 3 | code length), 8 bits each
constexpr uint32_t LIT_TAB = 1u << LIT_BITS, DIST_TAB = 1u << DIST_BITS;
constexpr uint32_t LIT_LONG = 80;    // symbols of the literal / length codes longer than LIT_BITS kept beside the table (canonical order);
                                     // the ones beyond that are read from the lane's scratch.  All 30 distance symbols are kept (DIST_SYMS).
constexpr uint32_t DIST_SYMS = 32;
constexpr uint32_t LIT_CNT_BITS = 9, DIST_CNT_BITS = 5;   // a LongCode packs the number of codes of each length above the table's: up to 286, up to 30
constexpr uint32_t MAX_MEMBER_TEXT = 65536;   // BGZF: a member holds at most 64 KiB of text

// ---- tokens -------------------------------------------------------------------------------------------------------------------
// bits [1:0] = c > 0: c literals, in bits [15:8], [23:16], [31:24];  c == 0: a copy, length in bits [10:2], distance in [26:11]
PAV_IHD uint32_t tok_match(uint32_t len, uint32_t dist) { return (len << 2) | (dist << 11); }
PAV_IHD uint32_t tok_bytes(uint32_t t) { return (t & 3u) ? (t & 3u) : ((t >> 2) & 0x1FFu); }
PAV_IHD uint32_t tok_dist(uint32_t t) { return t >> 11; }
// most tokens a member of `text` bytes can decode to: a token gives a byte at least, a copy three, a literal token is closed
// early only by a copy behind it - two tokens give four bytes or more, save the last
PAV_IHD uint32_t tok_capacity(uint32_t text) { return text / 2u + 2u; }

enum : int {
    IFL_OK = 0,
    IFL_E_BTYPE = 1,       // block type 3
    IFL_E_STORED = 2,      // LEN / NLEN of a stored block disagree
    IFL_E_LENGTHS = 3,     // code lengths: a repeat without a previous length, too many lengths, an over-subscribed code, no end-of-block code
    IFL_E_SYMBOL = 4,      // bits that are no code of the block's tree, or literal / length symbols 286, 287, distance symbols 30, 31
    IFL_E_DISTANCE = 5,    // a copy from before the start of the member's text
    IFL_E_TEXT = 6,        // more text than ISIZE says, or less
    IFL_E_INPUT = 7,       // the stream runs past the end of the member
    IFL_E_TOKENS = 8       // (token list full: cannot happen within tok_capacity)
};

// Per-lane scratch in HBM (L2-resident: a kilobyte): code lengths while a block's header is read, the symbols of both codes in
// canonical order and the number of codes of each length - what the long codes (rare: above LIT_BITS / DIST_BITS bits) decode from.
struct LaneScratch {
    uint8_t lens[344];               // [0, 19) the code-length code's lengths; from 19: [0, n_lit) literal / length code, then n_dist of the distance code
    uint16_t lit_sym[288];
    uint16_t dist_sym[32];
    uint16_t lit_count[16], dist_count[16];
    uint16_t offs[16];
    uint16_t cl_count[8], cl_sym[20];
};

struct BitReader {
    const uint8_t *base; uint64_t pos;   // next byte to go into `bits`
    uint64_t bits; uint32_t n;           // n valid bits in `bits` (bit 0 = next bit of the stream)
    uint32_t ahead;                      // the four bytes at `pos`, loaded one refill early: a lane has nothing else to do while a load is under way
};
// (unaligned four-byte load: gfx950 global loads take any address; the host test assembles bytes)
PAV_IHD uint32_t load32(const uint8_t *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return *reinterpret_cast<const uint32_t *>(p);
#else
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
#endif
}
PAV_IHD void reader_at(BitReader &B, uint64_t pos) { B.pos = pos; B.bits = 0; B.n = 0; B.ahead = load32(B.base + pos); }
PAV_IHD void refill(BitReader &B) {                      // at least 32 bits afterwards (the buffer behind the stream is padded)
    if (B.n < 32u) { B.bits |= (uint64_t)B.ahead << B.n; B.pos += 4; B.n += 32u; B.ahead = load32(B.base + B.pos); }
}
PAV_IHD uint32_t take(BitReader &B, uint32_t k) {        // k <= 16, k bits present
    const uint32_t v = (uint32_t)B.bits & ((1u << k) - 1u);
    B.bits >>= k; B.n -= k;
    return v;
}
PAV_IHD uint64_t bits_used(const BitReader &B) { return B.pos * 8u - B.n; }

PAV_IHD uint32_t reverse_bits(uint32_t v, uint32_t n) {  // the low n bits of v, reversed
    uint32_t r = 0;
    for (uint32_t i = 0; i < n; ++i) { r = (r << 1) | (v & 1u); v >>= 1; }
    return r;
}

// Where the canonical walk of a code stands behind the first-level table: the first code of the next length, the number of symbols
// with shorter codes, the number of codes of each longer length (packed, CNT_BITS each).  A code longer than the table is decoded
// from these three registers; only the symbol itself is fetched.
struct LongCode { uint32_t first, index; uint64_t counts, counts2; };      // counts: the first seven longer lengths, counts2: the ones above

// One Huffman code from its lengths: count[] per length and sym[] in canonical order (the lane's scratch), the first-level table and
// the symbols of the longer codes through the lane's tables T, the walk's state in L.  Returns false for an over-subscribed set of
// lengths.  An incomplete code is accepted: the bit patterns it leaves out decode to IFL_E_SYMBOL when they are met (zlib refuses
// the block up front; a stream zlib accepts decodes the same here).
template <bool LIT, class Tab>
PAV_IHD bool build_code(const uint8_t *lens, uint32_t n, uint16_t *count, uint16_t *sym, uint16_t *offs, Tab &T, LongCode &L) {
    constexpr uint32_t tab_bits = LIT ? LIT_BITS : DIST_BITS, cnt_bits = LIT ? LIT_CNT_BITS : DIST_CNT_BITS;
    for (uint32_t l = 0; l < 16; ++l) count[l] = 0;
    for (uint32_t s = 0; s < n; ++s) count[lens[s]] = (uint16_t)(count[lens[s]] + 1);
    int32_t left = 1;
    for (uint32_t l = 1; l < 16; ++l) { left <<= 1; left -= (int32_t)count[l]; if (left < 0) return false; }
    offs[0] = 0; offs[1] = 0;
    for (uint32_t l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
    for (uint32_t s = 0; s < n; ++s) { const uint32_t l = lens[s]; if (l) { sym[offs[l]] = (uint16_t)s; offs[l] = (uint16_t)(offs[l] + 1); } }
    const uint32_t tab = 1u << tab_bits;
    for (uint32_t e = 0; e < tab; ++e) { if (LIT) T.set_lit(e, 0); else T.set_dist(e, 0); }
    uint32_t code = 0, idx = 0;
    for (uint32_t l = 1; l <= tab_bits; ++l) {
        const uint32_t c = count[l];
        for (uint32_t i = 0; i < c; ++i, ++idx, ++code) {
            const uint32_t v = LIT ? ((uint32_t)sym[idx] << 4 | l) : ((uint32_t)sym[idx] << 3 | l);
            for (uint32_t e = reverse_bits(code, l); e < tab; e += 1u << l) { if (LIT) T.set_lit(e, (uint16_t)v); else T.set_dist(e, (uint8_t)v); }
        }
        code <<= 1;
    }
    L.first = code; L.index = idx; L.counts = 0; L.counts2 = 0;   // (code: the first code of length tab_bits + 1, as the loop leaves it)
    uint32_t used = idx;
    for (uint32_t l = tab_bits + 1; l < 16; ++l) {
        const uint32_t at = l - tab_bits - 1;
        if (at < 7) L.counts |= (uint64_t)count[l] << (at * cnt_bits); else L.counts2 |= (uint64_t)count[l] << ((at - 7) * cnt_bits);
        used += count[l];
    }
    if (LIT) { for (uint32_t i = idx; i < used && i - idx < LIT_LONG; ++i) T.set_lit_long(i - idx, sym[i]); }
    else { for (uint32_t i = 0; i < used && i < DIST_SYMS; ++i) T.set_dist_sym(i, (uint8_t)sym[i]); }
    return true;
}

// A code longer than the first-level table: the canonical walk from the table's length on, in registers.  Returns the symbol's place
// in canonical order (the bits consumed), -1 when the bits are no code.
template <bool LIT>
PAV_IHD int decode_long(BitReader &B, const LongCode &L) {
    constexpr uint32_t tab_bits = LIT ? LIT_BITS : DIST_BITS, cnt_bits = LIT ? LIT_CNT_BITS : DIST_CNT_BITS;
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t code = (__brev((uint32_t)B.bits) >> (32 - tab_bits)) << 1;
#else
    uint32_t code = reverse_bits((uint32_t)B.bits, tab_bits) << 1;
#endif
    uint32_t first = L.first, index = L.index;
    PAV_IUNROLL
    for (uint32_t l = tab_bits + 1; l < 16; ++l) {
        code |= (uint32_t)(B.bits >> (l - 1)) & 1u;
        const uint32_t at = l - tab_bits - 1;
        const uint32_t c = (uint32_t)(at < 7 ? L.counts >> (at * cnt_bits) : L.counts2 >> ((at - 7) * cnt_bits)) & ((1u << cnt_bits) - 1u);
        if (code >= first && code - first < c) { B.bits >>= l; B.n -= l; return (int)(index + (code - first)); }
        index += c; first = (first + c) << 1; code <<= 1;
    }
    return -1;
}

// Tables of a lane (LDS on the device): lit(e) / set_lit(e, v) the first-level entries of the literal / length code, lit_long(i) /
// set_lit_long(i, v) the first LIT_LONG symbols of its longer codes; dist(e) / set_dist(e, v), dist_sym(i) / set_dist_sym(i, v) the same
// for the distance code (every symbol kept).
template <class Tab>
PAV_IHD int read_dynamic_header(BitReader &B, Tab &T, LaneScratch *S, uint32_t &n_lit, uint32_t &n_dist) {
    refill(B);
    n_lit = take(B, 5) + 257u; n_dist = take(B, 5) + 1u;
    const uint32_t n_cl = take(B, 4) + 4u;
    if (n_lit > 286u || n_dist > 30u) return IFL_E_LENGTHS;
    // the code-length code: 19 symbols of up to 7 bits, decoded by the canonical walk (a few hundred symbols per block)
    uint8_t *cl = S->lens;                               // (its 19 lengths sit at the front of lens[] until the code is built)
    for (uint32_t i = 0; i < 19; ++i) cl[i] = 0;
    for (uint32_t i = 0; i < n_cl; ++i) {
        refill(B);
        const uint32_t at = i < 3 ? 16u + i : (i == 3 ? 0u : ((i & 1u) ? 7u - (i - 5u) / 2u : 8u + (i - 4u) / 2u));   // 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
        cl[at] = (uint8_t)take(B, 3);
    }
    for (uint32_t l = 0; l < 8; ++l) S->cl_count[l] = 0;
    for (uint32_t s = 0; s < 19; ++s) S->cl_count[cl[s]] = (uint16_t)(S->cl_count[cl[s]] + 1);
    { int32_t left = 1; for (uint32_t l = 1; l < 8; ++l) { left <<= 1; left -= (int32_t)S->cl_count[l]; if (left < 0) return IFL_E_LENGTHS; } }
    { uint32_t at = 0; for (uint32_t l = 1; l < 8; ++l) for (uint32_t s = 0; s < 19; ++s) if (cl[s] == l) S->cl_sym[at++] = (uint16_t)s; }
    const uint32_t total = n_lit + n_dist;
    uint32_t i = 0, prev = 0;
    while (i < total) {
        refill(B);
        uint32_t code = 0, first = 0, index = 0, sym = 0xFFFFu;
        for (uint32_t l = 1; l < 8; ++l) {
            code |= (uint32_t)(B.bits >> (l - 1)) & 1u;
            const uint32_t c = S->cl_count[l];
            if (code < first + c) { B.bits >>= l; B.n -= l; sym = S->cl_sym[index + (code - first)]; break; }
            index += c; first += c; first <<= 1; code <<= 1;
        }
        if (sym == 0xFFFFu) return IFL_E_SYMBOL;
        if (sym < 16u) { S->lens[19 + i] = (uint8_t)sym; prev = sym; ++i; continue; }   // (lens[] proper starts behind the code-length code's 19)
        uint32_t rep, val = 0;
        if (sym == 16u) { if (i == 0) return IFL_E_LENGTHS; val = prev; rep = 3u + take(B, 2); }
        else if (sym == 17u) rep = 3u + take(B, 3);
        else rep = 11u + take(B, 7);
        if (i + rep > total) return IFL_E_LENGTHS;
        for (uint32_t k = 0; k < rep; ++k) S->lens[19 + i + k] = (uint8_t)val;
        i += rep; prev = val;
    }
    if (S->lens[19 + 256] == 0) return IFL_E_LENGTHS;    // no end-of-block code
    (void)T;
    return IFL_OK;
}

// The stream `in` (deflate blocks up to the final one; `in_len` bytes, readable a few bytes beyond) -> tok[0 .. *n_tok), whose bytes
// add up to text_len exactly.  Returns IFL_OK or the error.
template <class Tab>
PAV_IHD int inflate_tokens(const uint8_t *in, uint32_t in_len, uint32_t text_len, uint32_t *tok, uint32_t tok_cap, uint32_t *n_tok,
                           Tab &T, LaneScratch *S) {
    BitReader B; B.base = in; reader_at(B, 0);
    uint32_t k = 0, out = 0, lit_acc = 0, lit_n = 0;
    const uint64_t in_bits = (uint64_t)in_len * 8u;
    // tokens leave four at a time (one 16-byte store: a lane waits for its last store whenever it waits for a load)
    uint32_t q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    auto push = [&](uint32_t t) {
        q0 = q1; q1 = q2; q2 = q3; q3 = t;
        ++k;
        if ((k & 3u) == 0 && k <= tok_cap) { tok[k - 4] = q0; tok[k - 3] = q1; tok[k - 2] = q2; tok[k - 1] = q3; }
    };
    auto flush = [&]() { if (lit_n) { push(lit_acc | lit_n); lit_acc = 0; lit_n = 0; } };
    LongCode LL{0, 0, 0, 0}, LD{0, 0, 0, 0};
    auto literal = [&](uint32_t b) {
        lit_acc |= b << (8u * lit_n + 8u);
        if (++lit_n == 3u) flush();
        ++out;
    };
    for (;;) {
        refill(B);
        if (bits_used(B) + 3u > in_bits) return IFL_E_INPUT;
        const uint32_t last = take(B, 1), type = take(B, 2);
        if (type == 3u) return IFL_E_BTYPE;
        if (type == 0u) {                                // stored: to the byte boundary, LEN, NLEN, the bytes
            take(B, B.n & 7u);
            refill(B);
            const uint32_t len = take(B, 16), nlen = take(B, 16);
            if ((len ^ nlen) != 0xFFFFu) return IFL_E_STORED;
            uint64_t at = bits_used(B) >> 3;             // (whole bytes are left in the reader: dropped, the copy reads the stream itself)
            if (at + len > in_len) return IFL_E_INPUT;
            if (out + len > text_len) return IFL_E_TEXT;
            for (uint32_t i = 0; i < len; ++i) literal(in[at + i]);
            reader_at(B, at + len);
        } else {
            uint32_t n_lit = 288, n_dist = 30;
            uint8_t *lens = S->lens + 19;
            if (type == 1u) {
                for (uint32_t s = 0; s < 288; ++s) lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
                for (uint32_t s = 0; s < 30; ++s) lens[288 + s] = 5;
            } else {
                const int rc = read_dynamic_header(B, T, S, n_lit, n_dist);
                if (rc != IFL_OK) return rc;
            }
            if (!build_code<true>(lens, n_lit, S->lit_count, S->lit_sym, S->offs, T, LL)) return IFL_E_LENGTHS;
            if (!build_code<false>(lens + n_lit, n_dist, S->dist_count, S->dist_sym, S->offs, T, LD)) return IFL_E_LENGTHS;
            for (;;) {
                refill(B);
                // a run of literals never reaches the check behind a copy: the reader itself must stay within the member (a valid
                // stream keeps pos <= in_len + 8: up to 64 buffered bits; the load ahead then ends at in_len + 12 - the buffer's padding
                // covers it; a dynamic block's header, the other loop without a copy, is at most ~600 bytes)
                if (B.pos > (uint64_t)in_len + 8u) return IFL_E_INPUT;
                uint32_t sym;
                { const uint32_t e = T.lit((uint32_t)B.bits & (LIT_TAB - 1u));
                  if (e & 15u) { sym = e >> 4; B.bits >>= (e & 15u); B.n -= (e & 15u); }
                  else {
                      const int at = decode_long<true>(B, LL);
                      if (at < 0) return IFL_E_SYMBOL;
                      const uint32_t j = (uint32_t)at - LL.index;
                      sym = j < LIT_LONG ? T.lit_long(j) : S->lit_sym[at];
                  } }
                if (sym < 256u) { if (out >= text_len) return IFL_E_TEXT; literal(sym); continue; }
                if (sym == 256u) break;
                if (sym > 285u) return IFL_E_SYMBOL;
                uint32_t len;
                { const uint32_t s = sym - 257u;
                  if (s < 8u) len = 3u + s;
                  else if (s == 28u) len = 258u;
                  else { const uint32_t eb = (s >> 2) - 1u; len = 3u + ((4u + (s & 3u)) << eb) + take(B, eb); } }
                refill(B);
                uint32_t dsym;
                { const uint32_t e = T.dist((uint32_t)B.bits & (DIST_TAB - 1u));
                  if (e & 7u) { dsym = e >> 3; B.bits >>= (e & 7u); B.n -= (e & 7u); }
                  else { const int at = decode_long<false>(B, LD); if (at < 0) return IFL_E_SYMBOL; dsym = T.dist_sym((uint32_t)at); } }
                if (dsym > 29u) return IFL_E_SYMBOL;
                uint32_t dist;
                if (dsym < 4u) dist = 1u + dsym;
                else { const uint32_t eb = (dsym >> 1) - 1u; dist = 1u + ((2u + (dsym & 1u)) << eb) + take(B, eb); }
                if (dist > out) return IFL_E_DISTANCE;
                if (out + len > text_len) return IFL_E_TEXT;
                flush();
                push(tok_match(len, dist));
                out += len;
                if (bits_used(B) > in_bits) return IFL_E_INPUT;
            }
        }
        if (bits_used(B) > in_bits) return IFL_E_INPUT;
        if (last) break;
    }
    flush();
    if (out != text_len) return IFL_E_TEXT;
    if (k > tok_cap) return IFL_E_TOKENS;
    { const uint32_t r = k & 3u;                          // the last one to three tokens
      if (r >= 3u) tok[k - 3] = q1;
      if (r >= 2u) tok[k - 2] = q2;
      if (r >= 1u) tok[k - 1] = q3; }
    *n_tok = k;
    return IFL_OK;
}

}  // namespace ifl
}  // namespace pav
