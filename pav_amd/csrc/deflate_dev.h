// Serial building blocks of the DEVICE deflate encoder (deflate.hip): canonical length-limited Huffman codes from symbol counts,
// the dynamic-block header of RFC 1951 (section 3.2.7), the symbol / extra-bit split of match lengths and distances (3.2.5), and
// the CRC-32 algebra that joins per-segment checksums into the gzip trailer (RFC 1952).  What rule call_cigar / call_inv_batch
// get from DataFrame.to_csv(compression='gzip') (rules/call.snakefile:845-846, rules/call_inv.snakefile:279-291) is a gzip file
// with this content; any conforming deflate stream of the same text is equivalent for every reader.
//
// Everything here is __host__ __device__ and free of wave intrinsics: a wave runs these on its LDS arrays (a few lanes or one),
// tests/native/deflate_check.cpp runs the same code on the host and inflates the result with zlib (test infrastructure).
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#define PAV_DHD __host__ __device__ __forceinline__
#else
#define PAV_DHD inline
#endif

namespace pav {
namespace dfl {

constexpr int N_LL = 286;            // literal / length alphabet (0..255 literals, 256 end of block, 257..285 lengths)
constexpr int N_D = 30;              // distance alphabet
constexpr int N_CL = 19;             // code-length alphabet
constexpr int MAX_BITS = 15, MAX_CL_BITS = 7;
constexpr int MIN_MATCH = 3, MAX_MATCH = 258;

// ---- symbols of a match (RFC 1951 3.2.5) --------------------------------------------------------------------------------
// length 3..258 -> code 257..285, number of extra bits, their value
PAV_DHD void len_symbol(uint32_t len, uint32_t &code, uint32_t &ebits, uint32_t &eval) {
    const uint32_t l3 = len - 3;
    if (l3 < 8) { code = 257 + l3; ebits = 0; eval = 0; return; }
    if (len == 258) { code = 285; ebits = 0; eval = 0; return; }
    uint32_t lg = 3; while ((l3 >> (lg + 1)) != 0) ++lg;                 // floor(log2(l3)), 3..7
    code = 257 + 4 * (lg - 1) + ((l3 >> (lg - 2)) & 3u);
    ebits = lg - 2;
    eval = l3 & ((1u << ebits) - 1u);
}
// distance 1..32768 -> code 0..29, extra bits, value
PAV_DHD void dist_symbol(uint32_t dist, uint32_t &code, uint32_t &ebits, uint32_t &eval) {
    const uint32_t d1 = dist - 1;
    if (d1 < 4) { code = d1; ebits = 0; eval = 0; return; }
    uint32_t lg = 2; while ((d1 >> (lg + 1)) != 0) ++lg;                 // floor(log2(d1)), 2..14
    code = 2 * lg + ((d1 >> (lg - 1)) & 1u);
    ebits = lg - 1;
    eval = d1 & ((1u << ebits) - 1u);
}

// ---- Huffman code lengths ---------------------------------------------------------------------------------------------
// Scratch of one tree (LDS on the device) for up to N symbols.
template <int N> struct HuffWorkT {
    uint16_t order[N];               // used symbols, ascending by (count, symbol)
    uint32_t weight[2 * N];          // leaves in that order, then the internal nodes in the order they are made
    uint16_t parent[2 * N];
    uint16_t depth[2 * N];
    uint32_t bl_count[MAX_BITS + 2];
    uint32_t n_used;
};
using HuffWork = HuffWorkT<288>;     // the literal / length and distance trees
using HuffWorkCl = HuffWorkT<20>;    // the code-length tree

// Position of symbol s among the used symbols in (count, symbol) order - the caller runs this for every used symbol (one lane
// each, or a loop) and stores order[rank] = s: a rank sort, n_used^2 compares spread over the lanes.
PAV_DHD uint32_t huff_rank(const uint32_t *freq, int n, int s) {
    uint32_t r = 0;
    const uint32_t fs = freq[s];
    for (int t = 0; t < n; ++t) {
        const uint32_t ft = freq[t];
        r += (ft != 0 && (ft < fs || (ft == fs && t < s))) ? 1u : 0u;
    }
    return r;
}

// Lengths from the sorted leaves (W.order, W.n_used filled): the two-queue construction (leaves and internal nodes are both met
// in ascending weight), depths top-down, then the length limit on the COUNTS per length - codes deeper than `limit` are folded
// into it and the Kraft sum is paid back by lengthening the deepest shorter code, one step at a time - and the lengths are dealt
// to the symbols from the rarest up.  One lane.  len[] of unused symbols = 0.
template <class Work> PAV_DHD void huff_lengths(const uint32_t *freq, int n, int limit, uint8_t *len, Work &W) {
    const int m = (int)W.n_used;
    for (int s = 0; s < n; ++s) len[s] = 0;
    if (m == 0) return;
    if (m == 1) { len[W.order[0]] = 1; return; }
    for (int i = 0; i < m; ++i) W.weight[i] = freq[W.order[i]];
    int i = 0, j = m, k = m;
    while (k < 2 * m - 1) {
        int a, b;
        if (i < m && (j >= k || W.weight[i] <= W.weight[j])) a = i++; else a = j++;
        if (i < m && (j >= k || W.weight[i] <= W.weight[j])) b = i++; else b = j++;
        W.weight[k] = W.weight[a] + W.weight[b];
        W.parent[a] = (uint16_t)k; W.parent[b] = (uint16_t)k;
        ++k;
    }
    W.depth[2 * m - 2] = 0;
    for (int x = 2 * m - 3; x >= m; --x) W.depth[x] = (uint16_t)(W.depth[W.parent[x]] + 1);
    for (int d = 0; d <= MAX_BITS + 1; ++d) W.bl_count[d] = 0;
    for (int x = 0; x < m; ++x) {
        int d = W.depth[W.parent[x]] + 1;
        if (d > limit) d = limit;
        W.bl_count[d]++;
    }
    uint32_t total = 0;
    for (int d = 1; d <= limit; ++d) total += W.bl_count[d] << (limit - d);
    while (total > (1u << limit)) {
        W.bl_count[limit]--;
        for (int d = limit - 1; d >= 1; --d)
            if (W.bl_count[d]) { W.bl_count[d]--; W.bl_count[d + 1] += 2; break; }
        --total;
    }
    int x = 0;
    for (int d = limit; d >= 1; --d)
        for (uint32_t c = 0; c < W.bl_count[d]; ++c) len[W.order[x++]] = (uint8_t)d;
}

PAV_DHD uint32_t bit_reverse(uint32_t v, int bits) {
#if defined(__HIP_DEVICE_COMPILE__)
    return bits ? __brev(v) >> (32 - bits) : 0u;
#else
    uint32_t r = 0;
    for (int i = 0; i < bits; ++i) { r = (r << 1) | (v & 1u); v >>= 1; }
    return r;
#endif
}

// Canonical codes (3.2.2) of the lengths, stored the way the bit stream takes them: reversed, so that the first bit of the code
// is bit 0.  code_len[s] = code | length << 16 (0 for unused symbols).  One lane; `count` / `next`: MAX_BITS + 2 words of scratch
// each (the caller's, so that a kernel indexes LDS and not private memory).
PAV_DHD void huff_codes(const uint8_t *len, int n, uint32_t *code_len, uint32_t *count, uint32_t *next) {
    for (int d = 0; d <= MAX_BITS; ++d) count[d] = 0;
    for (int s = 0; s < n; ++s) count[len[s]]++;
    count[0] = 0;
    uint32_t c = 0;
    next[0] = 0;
    for (int d = 1; d <= MAX_BITS; ++d) { c = (c + count[d - 1]) << 1; next[d] = c; }
    for (int s = 0; s < n; ++s) {
        const int d = len[s];
        code_len[s] = d ? (bit_reverse(next[d]++, d) | (uint32_t)d << 16) : 0u;
    }
}

// ---- the header of a dynamic block ------------------------------------------------------------------------------------
struct BitSink {                     // serial bit writer into 32-bit words (LDS or host memory), LSB first
    uint32_t *words; uint32_t n_bits;
    PAV_DHD void put(uint32_t value, uint32_t bits) {
        if (!bits) return;
        const uint32_t w = n_bits >> 5, sh = n_bits & 31u;
        if (sh == 0) words[w] = 0;
        words[w] |= value << sh;
        if (sh + bits > 32) words[w + 1] = value >> (32 - sh);
        else if (sh + bits == 32) { /* the next put clears its word */ }
        n_bits += bits;
    }
};

struct ClItem { uint8_t sym, extra; };

// Run-length form of the code lengths (3.2.7): items[] (at most n entries) and the counts of the 19 code-length symbols.
PAV_DHD int cl_sequence(const uint8_t *lens, int n, ClItem *items, uint32_t *cl_freq) {
    for (int i = 0; i < N_CL; ++i) cl_freq[i] = 0;
    int out = 0, i = 0;
    while (i < n) {
        const uint8_t v = lens[i];
        int run = 1;
        while (i + run < n && lens[i + run] == v) ++run;
        i += run;
        if (v == 0) {
            while (run >= 11) { const int r = run > 138 ? 138 : run; items[out++] = ClItem{18, (uint8_t)(r - 11)}; cl_freq[18]++; run -= r; }
            if (run >= 3) { items[out++] = ClItem{17, (uint8_t)(run - 3)}; cl_freq[17]++; run = 0; }
            while (run-- > 0) { items[out++] = ClItem{0, 0}; cl_freq[0]++; }
        } else {
            items[out++] = ClItem{v, 0}; cl_freq[v]++; --run;               // the length itself, then repeats of it
            while (run >= 3) { const int r = run > 6 ? 6 : run; items[out++] = ClItem{16, (uint8_t)(r - 3)}; cl_freq[16]++; run -= r; }
            while (run-- > 0) { items[out++] = ClItem{v, 0}; cl_freq[v]++; }
        }
    }
    return out;
}

// Scratch of the header (LDS on the device)
struct HeaderWork {
    uint32_t cl_freq[N_CL], cl_code[N_CL], count[MAX_BITS + 2], next[MAX_BITS + 2];
    uint8_t cl_len[N_CL + 1];
    uint8_t all_len[N_LL + N_D + 4];
    ClItem items[N_LL + N_D + 4];
    HuffWorkCl tree;
};

// the order the lengths of the code-length code are sent in (3.2.7): 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15, five bits
// per entry in two constants (no table in memory)
constexpr uint64_t cl_pack(int a, int b, int c, int d, int e, int f, int g = 0, int h = 0, int i = 0, int j = 0, int k = 0, int l = 0) {
    return (uint64_t)a | (uint64_t)b << 5 | (uint64_t)c << 10 | (uint64_t)d << 15 | (uint64_t)e << 20 | (uint64_t)f << 25 | (uint64_t)g << 30 |
           (uint64_t)h << 35 | (uint64_t)i << 40 | (uint64_t)j << 45 | (uint64_t)k << 50 | (uint64_t)l << 55;
}
PAV_DHD uint32_t cl_order(int i) {
    constexpr uint64_t lo = cl_pack(16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4), hi = cl_pack(12, 3, 13, 2, 14, 1, 15);
    return (uint32_t)((i < 12 ? lo >> (5 * i) : hi >> (5 * (i - 12))) & 31u);
}

// Header of one dynamic block into `sink`: BFINAL, BTYPE = 2, HLIT, HDIST, HCLEN, the code-length code, the two trees.
// ll_len / d_len: lengths of the two alphabets (at least two used symbols each, see deflate.hip).
PAV_DHD void block_header(BitSink &sink, bool final_block, const uint8_t *ll_len, const uint8_t *d_len, HeaderWork &X) {
    int hlit = N_LL, hdist = N_D;
    while (hlit > 257 && ll_len[hlit - 1] == 0) --hlit;
    while (hdist > 1 && d_len[hdist - 1] == 0) --hdist;
    for (int i = 0; i < hlit; ++i) X.all_len[i] = ll_len[i];
    for (int i = 0; i < hdist; ++i) X.all_len[hlit + i] = d_len[i];
    const int n_items = cl_sequence(X.all_len, hlit + hdist, X.items, X.cl_freq);
    X.tree.n_used = 0;
    for (int s = 0; s < N_CL; ++s) if (X.cl_freq[s]) X.tree.n_used++;
    for (int s = 0; s < N_CL && X.tree.n_used < 2; ++s) if (!X.cl_freq[s]) { X.cl_freq[s] = 1; X.tree.n_used++; }   // a complete code needs two symbols
    for (int s = 0; s < N_CL; ++s) if (X.cl_freq[s]) X.tree.order[huff_rank(X.cl_freq, N_CL, s)] = (uint16_t)s;
    huff_lengths(X.cl_freq, N_CL, MAX_CL_BITS, X.cl_len, X.tree);
    huff_codes(X.cl_len, N_CL, X.cl_code, X.count, X.next);
    int hclen = N_CL;
    while (hclen > 4 && X.cl_len[cl_order(hclen - 1)] == 0) --hclen;
    sink.put(final_block ? 1u : 0u, 1); sink.put(2u, 2);
    sink.put((uint32_t)(hlit - 257), 5); sink.put((uint32_t)(hdist - 1), 5); sink.put((uint32_t)(hclen - 4), 4);
    for (int i = 0; i < hclen; ++i) sink.put(X.cl_len[cl_order(i)], 3);
    for (int i = 0; i < n_items; ++i) {
        const ClItem it = X.items[i];
        sink.put(X.cl_code[it.sym] & 0xFFFFu, X.cl_code[it.sym] >> 16);
        if (it.sym == 16) sink.put(it.extra, 2);
        else if (it.sym == 17) sink.put(it.extra, 3);
        else if (it.sym == 18) sink.put(it.extra, 7);
    }
}

// ---- CRC-32 (reflected, polynomial 0xEDB88320) ------------------------------------------------------------------------
constexpr uint32_t CRC_POLY = 0xEDB88320u;
PAV_DHD uint32_t crc_table_entry(uint32_t i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ CRC_POLY : c >> 1; return c; }
// product of two polynomials modulo the CRC polynomial, reflected form (bit 31 is x^0)
PAV_DHD uint32_t gf_mul(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int i = 0; i < 32; ++i) {
        if (a & (0x80000000u >> i)) p ^= b;
        b = (b & 1u) ? (b >> 1) ^ CRC_POLY : b >> 1;                     // b * x
    }
    return p;
}
PAV_DHD uint32_t gf_xpow8(uint64_t n_bytes) {                             // x^(8 n) mod p
    uint32_t r = 0x80000000u, base = 0x40000000u;                         // 1, x
    uint64_t e = n_bytes * 8;                                            // (n < 2^61)
    while (e) { if (e & 1) r = gf_mul(r, base); base = gf_mul(base, base); e >>= 1; }
    return r;
}
// crc(A || B) from crc(A), crc(B), |B|
PAV_DHD uint32_t crc_join(uint32_t crc_a, uint32_t crc_b, uint64_t len_b) { return gf_mul(gf_xpow8(len_b), crc_a) ^ crc_b; }

}  // namespace dfl
}  // namespace pav
