// BGZF -> text on the device.  PAV keeps its FASTA files bgzipped (rules/call.snakefile:796 `contigs_{hap}.fa.gz`, `data/ref/ref.fa.gz`,
// read through pysam.FastaFile: pavlib/cigarcall.py:59-64): 3.1 GB of reference text are 0.9 GB on disk - and 0.9 GB across PCIe when
// the members are inflated here instead of on sixteen host threads (0.5 s of a haplotype's 0.5 s files to files otherwise).
//
// A BGZF file is a series of independent gzip members of at most 64 KiB of text (SAM specification 4.1).  Inside a member nothing
// is parallel at first sight - a Huffman symbol starts where the one before it ends - so the work is split by what IS independent:
//   k_bgzf_footers     a lane per member: ISIZE and CRC-32 from the eight bytes behind the payload (a prefix sum of ISIZE is the
//                      member's place in the text: scan_dev.h)
//   k_inflate_tokens   a LANE per member walks the member's deflate blocks (inflate_dev.h: stored, fixed, dynamic) into tokens -
//                      up to three literals, or a copy (length, distance) - with its two first-level decode tables and the symbols
//                      of the longer codes in LDS, interleaved over the 64 lanes (entry e of lane l at e * 64 + l), 52 KiB per wave,
//                      three waves per CU: every member of a 3 GB file is in flight at once.  A code longer than its table is
//                      walked in registers (ifl::LongCode); tokens leave as 16-byte stores, the stream is read a word ahead -
//                      a lane that waits for a load waits for every store before it
//   k_inflate_resolve_ring   a WAVE per member turns the tokens into text in a 36 KiB LDS ring (32 KiB of history + the step's own
//                      text), flushed to HBM as it is made, four waves a CU: 64 tokens per step, places by a prefix sum of their byte
//                      counts, literals stored at once, copies eight bytes at a time by the lanes whose source is complete (the first
//                      pending copy always is), the long ones - runs of N, tandem repeats - by the whole wave
//                      (k_inflate_resolve, PAV_INFLATE_WINDOW=full: the first version, the member's whole 64 KiB in LDS, two waves a CU)
//   k_bgzf_crc         a wave per member: CRC-32 of the text in HBM against the footer (a lane 64 bytes of every 4 KiB tile; chains,
//                      tiles and lanes joined by the checksum's algebra)
// The text is what zlib's inflate gives for the same members (tests/test_gpu_bgzf.py; the serial decoder alone against zlib on the
// host: tests/native/inflate_check.cpp).
#include "inflatedev.h"

#include "deflate_dev.h"    // the CRC-32 algebra
#include "crc_wave.h"
#include "inflate_dev.h"
#include "scan_dev.h"

#include <algorithm>
#include <chrono>
#include <cstring>
#include <mutex>
#include <utility>
#include <vector>

namespace pav {

namespace {

// crc_init, xp_sub: two powers of x modulo the CRC polynomial that the member's checksum needs (k_inflate_resolve), found by the lane
// that reads the footer: the term of the register's initial value, all-ones * x^(8 text_len), and x^(8 L), L = ceil(text_len / 256)
struct BgzfMember { uint64_t in_off; uint64_t out_off; uint32_t in_len, text_len, crc, crc_init, xp_sub, pad; };

constexpr uint32_t TOK_STRIDE = 32784;                  // tokens a member may decode to (ifl::tok_capacity(65536) = 32770), a multiple of 16
constexpr uint32_t BATCH_MEMBERS = 49152;               // members per launch of the two inflate kernels: 768 waves of lanes, three on every CU
                                                        // (a 3 GB file in one launch; its token lists: 6.4 GB)
constexpr uint32_t ST_ISIZE = 20, ST_CRC = 21;          // status of a member beyond the decoder's own (ifl::IFL_E_*)
constexpr uint32_t RESOLVE_LDS = 65536 + 32;

__global__ __launch_bounds__(256) void k_bgzf_footers(const uint8_t *__restrict__ comp, BgzfMember *__restrict__ mem, uint32_t n, uint32_t *__restrict__ text_len,
                                                      uint32_t *__restrict__ status) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint8_t *f = comp + mem[i].in_off + mem[i].in_len;
    const uint32_t crc = (uint32_t)f[0] | (uint32_t)f[1] << 8 | (uint32_t)f[2] << 16 | (uint32_t)f[3] << 24;
    const uint32_t isize = (uint32_t)f[4] | (uint32_t)f[5] << 8 | (uint32_t)f[6] << 16 | (uint32_t)f[7] << 24;
    const bool ok = isize <= ifl::MAX_MEMBER_TEXT;
    mem[i].crc = crc; mem[i].text_len = ok ? isize : 0u;
    mem[i].crc_init = ok ? dfl::gf_mul(0xFFFFFFFFu, dfl::gf_xpow8(isize)) : 0u;
    mem[i].xp_sub = ok ? dfl::gf_xpow8((isize + 255u) / 256u) : 0u;
    text_len[i] = ok ? isize : 0u;
    status[i] = ok ? 0u : ST_ISIZE;
}

__global__ __launch_bounds__(256) void k_bgzf_places(BgzfMember *__restrict__ mem, const uint64_t *__restrict__ out_off, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) mem[i].out_off = out_off[i];
}

// A lane's tables, interleaved with the other lanes' (entry e of lane l at e * 64 + l): the 16-bit entries of the literal / length
// table and of its long codes' symbols, then the 8-bit entries of the distance table and the distance symbols - 832 bytes a lane.
constexpr uint32_t TAB16 = (ifl::LIT_TAB + ifl::LIT_LONG) * 64, TAB8 = (ifl::DIST_TAB + ifl::DIST_SYMS) * 64;
struct LdsTab {
    uint16_t *t16; uint8_t *t8; uint32_t lane;
    __device__ __forceinline__ uint16_t lit(uint32_t e) const { return t16[e * 64u + lane]; }
    __device__ __forceinline__ uint16_t lit_long(uint32_t i) const { return t16[(ifl::LIT_TAB + i) * 64u + lane]; }
    __device__ __forceinline__ uint8_t dist(uint32_t e) const { return t8[e * 64u + lane]; }
    __device__ __forceinline__ uint8_t dist_sym(uint32_t i) const { return t8[(ifl::DIST_TAB + i) * 64u + lane]; }
    __device__ __forceinline__ void set_lit(uint32_t e, uint16_t v) { t16[e * 64u + lane] = v; }
    __device__ __forceinline__ void set_lit_long(uint32_t i, uint16_t v) { t16[(ifl::LIT_TAB + i) * 64u + lane] = v; }
    __device__ __forceinline__ void set_dist(uint32_t e, uint8_t v) { t8[e * 64u + lane] = v; }
    __device__ __forceinline__ void set_dist_sym(uint32_t i, uint8_t v) { t8[(ifl::DIST_TAB + i) * 64u + lane] = v; }
};

__global__ __launch_bounds__(64) void k_inflate_tokens(const uint8_t *__restrict__ comp, const BgzfMember *__restrict__ mem, uint32_t n,
                                                       uint32_t *__restrict__ tok, uint32_t *__restrict__ n_tok, uint32_t *__restrict__ status,
                                                       ifl::LaneScratch *__restrict__ scratch) {
    __shared__ uint16_t tab16[TAB16];
    __shared__ uint8_t tab8[TAB8];
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    n_tok[i] = 0;
    if (status[i]) return;
    const BgzfMember M = mem[i];
    LdsTab T{tab16, tab8, threadIdx.x};
    uint32_t nt = 0;
    const int rc = ifl::inflate_tokens(comp + M.in_off, M.in_len, M.text_len, tok + (size_t)i * TOK_STRIDE, ifl::tok_capacity(M.text_len), &nt, T, scratch + i);
    if (rc != ifl::IFL_OK) { status[i] = (uint32_t)rc; return; }
    n_tok[i] = nt;
}

extern __shared__ __align__(16) uint8_t resolve_lds[];

// inclusive sum over the 64 lanes in six DPP steps (rows of 16 by row_shr, the rows joined by row_bcast:15 / :31)
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}

constexpr uint32_t TOK_GROUP = 512;                     // tokens fetched ahead of the steps that use them
constexpr uint32_t WAVE_COPY_LEN = 32;                  // a copy this long is made by the whole wave, the shorter ones by their lanes

// PROF (PAV_INFLATE_PROFILE=1): the wave's clock at the seams of the kernel and a few counts, summed over a member, left in prof[16 m ..]
// - [0] steps of 64 tokens, [1] trips of the copy loop, [2] iterations of the lanes' own copies, [3] copies made by the whole wave,
// cycles: [4] before the first step, [5] token fetch + places + literals, [6] copies, [7] CRC-32, [8] window -> HBM.
template <bool PROF>
__global__ __launch_bounds__(64) void k_inflate_resolve(const uint32_t *__restrict__ tok, const uint32_t *__restrict__ n_tok, const BgzfMember *__restrict__ mem,
                                                        uint32_t *__restrict__ status, uint8_t *__restrict__ out, unsigned long long *__restrict__ prof) {
    unsigned long long pc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, clk = PROF ? __builtin_readcyclecounter() : 0ull;
    auto lap = [&](int i) { if (PROF) { const unsigned long long now = __builtin_readcyclecounter(); pc[i] += now - clk; clk = now; } };
    __shared__ uint32_t crc_tab[256];
    __shared__ uint32_t tok_buf[2][TOK_GROUP];
    const uint32_t m = blockIdx.x, lane = threadIdx.x;
    if (status[m]) return;
    for (uint32_t i = lane; i < 256; i += 64) crc_tab[i] = dfl::crc_table_entry(i);
    const BgzfMember M = mem[m];
    const uint32_t shift = (uint32_t)(M.out_off & 15u);  // the window sits in LDS as it will sit in HBM, modulo 16
    uint8_t *W = resolve_lds + shift;
    const uint32_t nt = n_tok[m];
    const uint32_t *tp = tok + (size_t)m * TOK_STRIDE;
    uint32_t done = 0;
    // The tokens come through a small LDS buffer, 512 at a time: the loads of the next 512 are under way while these are worked on
    // (a step of 64 tokens takes less than a trip to HBM: one load a step in flight was the kernel's whole time).
    uint32_t pre[TOK_GROUP / 64];
#pragma unroll
    for (uint32_t j = 0; j < TOK_GROUP / 64; ++j) { const uint32_t i = j * 64 + lane; pre[j] = i < nt ? tp[i] : 0u; }
#pragma unroll
    for (uint32_t j = 0; j < TOK_GROUP / 64; ++j) tok_buf[0][j * 64 + lane] = pre[j];
    lap(4);
    for (uint32_t b = 0; b < nt; b += 64) {
        if (PROF) ++pc[0];
        const uint32_t half = (b / TOK_GROUP) & 1u, in_group = b % TOK_GROUP;
        if (in_group == 0) {
#pragma unroll
            for (uint32_t j = 0; j < TOK_GROUP / 64; ++j) { const uint32_t i = b + TOK_GROUP + j * 64 + lane; pre[j] = i < nt ? tp[i] : 0u; }
        }
        const uint32_t t = tok_buf[half][in_group + lane];
        if (in_group == TOK_GROUP - 64) {                 // the group's last step has its tokens: the next group's go to the other half
#pragma unroll
            for (uint32_t j = 0; j < TOK_GROUP / 64; ++j) tok_buf[half ^ 1u][j * 64 + lane] = pre[j];
        }
        const uint32_t bytes = b + lane < nt ? ifl::tok_bytes(t) : 0u;
        const uint32_t incl = wave_incl_sum(bytes);
        const uint32_t start = done + incl - bytes;
        const uint32_t c = t & 3u;
        if (bytes && c) {
            W[start] = (uint8_t)(t >> 8);
            if (c > 1) W[start + 1] = (uint8_t)(t >> 16);
            if (c > 2) W[start + 2] = (uint8_t)(t >> 24);
        }
        bool pending = bytes && !c;
        const uint32_t len = bytes, dist = ifl::tok_dist(t);
        const uint32_t src = start - dist, period = min(dist, len);
        const uint32_t tiny = (dist < 8u && dist < len) ? 1u : 0u;       // a copy whose source wraps inside eight bytes
        bool swept = false;
        lap(5);
        for (;;) {
            const unsigned long long pm = __ballot(pending);
            if (!pm) break;
            if (PROF) ++pc[1];
            const int first = __ffsll((long long)pm) - 1;
            const uint32_t ready = (uint32_t)__builtin_amdgcn_readlane((int)start, first);   // everything in front of the first pending copy is written
            const uint32_t f_len = (uint32_t)__builtin_amdgcn_readlane((int)len, first);
            if (f_len >= WAVE_COPY_LEN || __builtin_amdgcn_readlane((int)tiny, first)) {
                // a long copy (runs of N, tandem repeats: chains of 258-byte copies each reading the one before) or one at a distance
                // under eight: the wave writes it, 64 bytes a step
                const uint32_t f_src = (uint32_t)__builtin_amdgcn_readlane((int)src, first), f_per = (uint32_t)__builtin_amdgcn_readlane((int)period, first);
                const float inv = 1.0f / (float)f_per;
                for (uint32_t k = lane; k < f_len; k += 64) {
                    uint32_t r = k - (uint32_t)((float)k * inv) * f_per;    // k mod f_per (k < 258: the quotient is off by one at most)
                    if ((int)r < 0) r += f_per;
                    if (r >= f_per) r -= f_per;
                    W[ready + k] = W[f_src + r];
                }
                if ((int)lane == first) pending = false;
                if (PROF) ++pc[3];
                continue;
            }
            // First sweep: every copy whose source ends in front of the first pending one - all but a few (a FASTA line break makes a
            // short copy from a line or two above: a source inside the step's own text).  After it the few that are left go as soon as
            // their source touches no text that is still pending, which for most is the next sweep.
            bool go = pending && len < WAVE_COPY_LEN && !tiny;
            if (!swept || __popcll(pm) > 8) go = go && src + period <= ready;
            else {
                unsigned long long rest = pm;
                while (rest) {
                    const int i = __ffsll((long long)rest) - 1;
                    rest &= rest - 1;
                    const uint32_t s_i = (uint32_t)__builtin_amdgcn_readlane((int)start, i), e_i = s_i + (uint32_t)__builtin_amdgcn_readlane((int)len, i);
                    if (i != (int)lane && src < e_i && src + period > s_i) go = false;
                }
            }
            // The lanes' own copies, eight bytes an LDS access (any alignment).  The source of such a copy does not wrap inside eight
            // bytes (the few that do - a distance under eight, shorter than the length - are the wave's, above), and what a copy reads
            // of its own output was written an iteration ago.  A copy of eight bytes or more ends with its last eight bytes written
            // again rather than fewer than eight; a shorter one is two overlapping words, or three bytes.
            if (go) {                                     // the first eight bytes of every copy - all there is of most
                if (PROF) ++pc[2];
                uint64_t v;
                __builtin_memcpy(&v, W + src, 8);
                if (len >= 8u) __builtin_memcpy(W + start, &v, 8);
                else if (len >= 4u) {
                    const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> (8u * (len - 4u)));
                    __builtin_memcpy(W + start, &lo, 4);
                    __builtin_memcpy(W + start + len - 4u, &hi, 4);
                } else {
                    const uint16_t lo = (uint16_t)v;
                    __builtin_memcpy(W + start, &lo, 2);
                    W[start + 2] = (uint8_t)(v >> 16);
                }
            }
            uint32_t k = 8;
            while (__ballot(go && k < len)) {             // the rest of the longer ones: a read and a write an iteration
                if (PROF) ++pc[2];
                if (go && k < len) {
                    const uint32_t kk = min(k, len - 8u);
                    uint64_t v;
                    __builtin_memcpy(&v, W + src + kk, 8);
                    __builtin_memcpy(W + start + kk, &v, 8);
                    k = kk + 8u;
                }
            }
            pending = pending && !go;
            swept = true;
        }
        done += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        lap(6);
    }
    const uint32_t n = M.text_len;
    // CRC-32 of the text, read where it lies: the text, zeros in front of it up to 256 L bytes, is 256 pieces of L bytes - four to a
    // lane, four independent chains of table look-ups.  The register is run from zero (zeros in front change nothing then) and the
    // pieces are joined in pairs, x^(8 L) doubling its exponent at every level; the initial value's term is added at the end.
    if (n) {
        __syncthreads();
        const uint32_t L = (n + 255u) / 256u, pad = 256u * L - n, q0 = 4u * L * lane;
        uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
        for (uint32_t i = 0; i < L; ++i) {
            const uint32_t a = q0 + i, b = a + L, c = b + L, d = c + L;
            const uint32_t ba = a >= pad ? W[a - pad] : 0u, bb = b >= pad ? W[b - pad] : 0u, bc = c >= pad ? W[c - pad] : 0u, bd = d >= pad ? W[d - pad] : 0u;
            r0 = crc_tab[(r0 ^ ba) & 0xFFu] ^ (r0 >> 8); r1 = crc_tab[(r1 ^ bb) & 0xFFu] ^ (r1 >> 8);
            r2 = crc_tab[(r2 ^ bc) & 0xFFu] ^ (r2 >> 8); r3 = crc_tab[(r3 ^ bd) & 0xFFu] ^ (r3 >> 8);
        }
        uint32_t f = M.xp_sub;
        const uint32_t lo = dfl::gf_mul(f, r0) ^ r1, hi = dfl::gf_mul(f, r2) ^ r3;
        f = dfl::gf_mul(f, f);
        uint32_t v = dfl::gf_mul(f, lo) ^ hi;
        for (uint32_t d = 1; d < 64; d <<= 1) {
            f = dfl::gf_mul(f, f);
            const uint32_t o = (uint32_t)__shfl_xor((int)v, (int)d);
            v = (lane & d) ? dfl::gf_mul(f, o) ^ v : dfl::gf_mul(f, v) ^ o;
        }
        if (lane == 0 && (v ^ M.crc_init ^ 0xFFFFFFFFu) != M.crc) status[m] = ST_CRC;
    } else if (lane == 0 && M.crc != 0) status[m] = ST_CRC;
    lap(7);
    // the window -> HBM: the bytes up to the first 16-byte boundary, whole 16-byte pieces, the rest
    uint8_t *dst = out + M.out_off;
    const uint32_t head = min(n, (16u - shift) & 15u);
    if (lane < head) dst[lane] = W[lane];
    const uint32_t body = (n - head) / 16u;
    for (uint32_t q = lane; q < body; q += 64) *reinterpret_cast<uint4 *>(dst + head + q * 16u) = *reinterpret_cast<const uint4 *>(W + head + q * 16u);
    const uint32_t tail = head + body * 16u;
    if (tail + lane < n) dst[tail + lane] = W[tail + lane];
    if (PROF) { lap(8); if (lane == 0) for (int i = 0; i < 9; ++i) prof[16ull * m + i] = pc[i]; }
}

// ---- the resolve kernel with a ring for a window ---------------------------------------------------------------------------------
// k_inflate_resolve above holds a member's whole text in LDS: two waves a CU, each alone on its SIMD, issuing in half of its cycles
// and parked in s_waitcnt for the other half (profiles/r05_bgzf_loader_pmc_sq.txt).  A copy reaches back 32 KiB at most, so the
// window need only hold that much history and the step's own text: a ring of 36 KiB, the text flushed to HBM as it is made, four
// waves a CU.  A step is cut short where its text would pass 4 080 bytes (runs of long copies; a step of FASTA text is ~600 bytes).
// Positions map to the ring as (position + the text's alignment in HBM) mod RING, so whole 16-byte pieces of HBM are whole pieces
// of the ring.  An 8-byte access may start up to seven bytes before the ring's end: the first 16 bytes of the ring have a copy
// behind its end, kept by every store (ring_store); a store that runs past the end is repeated at the front.
// The CRC-32 is taken afterwards from the text in HBM (k_bgzf_crc).
constexpr uint32_t RING = 36864, RING_FRONT = 16, RING_BACK = 32, RING_SPAN = 4080, RING_TOK = 512;
constexpr uint32_t RING_LDS = RING_FRONT + RING + RING_BACK + 4 * RING_TOK;
__device__ __forceinline__ uint32_t ring_ix(uint32_t q) { return q >= RING ? q - RING : q; }             // q < 2 RING
template <class T> __device__ __forceinline__ void ring_store(uint8_t *R0, uint32_t r, T v) {
    __builtin_memcpy(R0 + r, &v, sizeof(T));
    if (r + (uint32_t)sizeof(T) > RING) __builtin_memcpy(R0 + r - RING, &v, sizeof(T));          // (R0 has RING_FRONT bytes in front of it)
    else if (r < 16u) __builtin_memcpy(R0 + r + RING, &v, sizeof(T));
}

template <bool PROF>
__global__ __launch_bounds__(64) void k_inflate_resolve_ring(const uint32_t *__restrict__ tok, const uint32_t *__restrict__ n_tok, const BgzfMember *__restrict__ mem,
                                                             const uint32_t *__restrict__ status, uint8_t *__restrict__ out, unsigned long long *__restrict__ prof) {
    unsigned long long pc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, clk = PROF ? __builtin_readcyclecounter() : 0ull;
    auto lap = [&](int i) { if (PROF) { const unsigned long long now = __builtin_readcyclecounter(); pc[i] += now - clk; clk = now; } };
    const uint32_t m = blockIdx.x, lane = threadIdx.x;
    if (status[m]) return;
    uint8_t *R0 = resolve_lds + RING_FRONT;
    uint32_t *tbuf = reinterpret_cast<uint32_t *>(resolve_lds + RING_FRONT + RING + RING_BACK);
    const BgzfMember M = mem[m];
    const uint32_t shift = (uint32_t)(M.out_off & 15u);
    uint8_t *dst = out + M.out_off - shift;                // dst + q: the byte of ring position q (q = text position + shift)
    const uint32_t nt = n_tok[m];
    const uint32_t *tp = tok + (size_t)m * TOK_STRIDE;
    // tokens through a circular buffer of 512: the next 256 are loaded into registers a step before they are stored over the 256
    // that the steps have left behind
#pragma unroll
    for (uint32_t j = 0; j < RING_TOK / 64; ++j) { const uint32_t i = j * 64 + lane; tbuf[i] = i < nt ? tp[i] : 0u; }
    uint32_t loaded = RING_TOK, pre_at = 0, pre[4] = {0, 0, 0, 0};
    bool pre_full = false;
    uint32_t done = 0, flushed = 0, b = 0;
    lap(4);
    while (b < nt) {
        if (PROF) ++pc[0];
        if (pre_full) {
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) tbuf[(pre_at + j * 64 + lane) & (RING_TOK - 1)] = pre[j];
            loaded += 256; pre_full = false;
        }
        if (loaded < nt && loaded - b <= 256u) {
            pre_at = loaded;
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) { const uint32_t i = loaded + j * 64 + lane; pre[j] = i < nt ? tp[i] : 0u; }
            pre_full = true;
        }
        const uint32_t t = b + lane < nt ? tbuf[(b + lane) & (RING_TOK - 1)] : 0u;
        uint32_t bytes = b + lane < nt ? ifl::tok_bytes(t) : 0u;
        const uint32_t incl = wave_incl_sum(bytes);
        const bool in = b + lane < nt && incl <= RING_SPAN;            // the step's tokens: a prefix of the lanes (one token at least)
        const uint32_t n_take = (uint32_t)__popcll(__ballot(in));
        if (!in) bytes = 0;
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)n_take - 1);
        const uint32_t start = done + incl - bytes;                    // (of the lanes in the step)
        const uint32_t qs = start + shift;
        const uint32_t c = t & 3u;
        if (bytes && c) {
            ring_store<uint8_t>(R0, ring_ix(qs), (uint8_t)(t >> 8));
            if (c > 1) ring_store<uint8_t>(R0, ring_ix(qs + 1), (uint8_t)(t >> 16));
            if (c > 2) ring_store<uint8_t>(R0, ring_ix(qs + 2), (uint8_t)(t >> 24));
        }
        bool pending = bytes && !c;
        const uint32_t len = bytes, dist = ifl::tok_dist(t);
        const uint32_t src = start - dist, period = min(dist, len), qsrc = qs - dist;
        const uint32_t tiny = (dist < 8u && dist < len) ? 1u : 0u;
        bool swept = false;
        lap(5);
        for (;;) {
            const unsigned long long pm = __ballot(pending);
            if (!pm) break;
            if (PROF) ++pc[1];
            const int first = __ffsll((long long)pm) - 1;
            const uint32_t ready = (uint32_t)__builtin_amdgcn_readlane((int)start, first);
            const uint32_t f_len = (uint32_t)__builtin_amdgcn_readlane((int)len, first);
            if (f_len >= WAVE_COPY_LEN || __builtin_amdgcn_readlane((int)tiny, first)) {
                const uint32_t f_qsrc = (uint32_t)__builtin_amdgcn_readlane((int)qsrc, first), f_per = (uint32_t)__builtin_amdgcn_readlane((int)period, first);
                const uint32_t f_qs = ready + shift;
                const float inv = 1.0f / (float)f_per;
                for (uint32_t k = lane; k < f_len; k += 64) {
                    uint32_t r = k - (uint32_t)((float)k * inv) * f_per;
                    if ((int)r < 0) r += f_per;
                    if (r >= f_per) r -= f_per;
                    ring_store<uint8_t>(R0, ring_ix(f_qs + k), R0[ring_ix(f_qsrc + r)]);
                }
                if ((int)lane == first) pending = false;
                if (PROF) ++pc[3];
                continue;
            }
            bool go = pending && len < WAVE_COPY_LEN && !tiny;
            if (!swept || __popcll(pm) > 8) go = go && src + period <= ready;
            else {
                unsigned long long rest = pm;
                while (rest) {
                    const int i = __ffsll((long long)rest) - 1;
                    rest &= rest - 1;
                    const uint32_t s_i = (uint32_t)__builtin_amdgcn_readlane((int)start, i), e_i = s_i + (uint32_t)__builtin_amdgcn_readlane((int)len, i);
                    if (i != (int)lane && src < e_i && src + period > s_i) go = false;
                }
            }
            if (go) {
                if (PROF) ++pc[2];
                uint64_t v;
                __builtin_memcpy(&v, R0 + ring_ix(qsrc), 8);
                if (len >= 8u) ring_store<uint64_t>(R0, ring_ix(qs), v);
                else if (len >= 4u) {
                    ring_store<uint32_t>(R0, ring_ix(qs), (uint32_t)v);
                    ring_store<uint32_t>(R0, ring_ix(qs + len - 4u), (uint32_t)(v >> (8u * (len - 4u))));
                } else {
                    ring_store<uint16_t>(R0, ring_ix(qs), (uint16_t)v);
                    ring_store<uint8_t>(R0, ring_ix(qs + 2u), (uint8_t)(v >> 16));
                }
            }
            uint32_t k = 8;
            while (__ballot(go && k < len)) {
                if (PROF) ++pc[2];
                if (go && k < len) {
                    const uint32_t kk = min(k, len - 8u);
                    uint64_t v;
                    __builtin_memcpy(&v, R0 + ring_ix(qsrc + kk), 8);
                    ring_store<uint64_t>(R0, ring_ix(qs + kk), v);
                    k = kk + 8u;
                }
            }
            pending = pending && !go;
            swept = true;
        }
        done += total;
        b += n_take;
        lap(6);
        // the finished text -> HBM, in whole 16-byte pieces (2 KiB of it or more at a time; everything after the last step)
        const bool last = b >= nt;
        if (done - flushed >= 2048u || last) {
            uint32_t q0 = flushed + shift;
            const uint32_t qd = done + shift;
            const uint32_t qa = min((q0 + 15u) & ~15u, qd);            // byte by byte up to the first boundary (the text's first bytes)
            if (q0 + lane < qa) dst[q0 + lane] = R0[ring_ix(q0 + lane)];
            q0 = qa;
            const uint32_t qe = max(q0, qd & ~15u);
            for (uint32_t q = q0 + 16u * lane; q + 16u <= qe; q += 1024u) *reinterpret_cast<uint4 *>(dst + q) = *reinterpret_cast<const uint4 *>(R0 + ring_ix(q));
            q0 = qe;
            if (last) { if (q0 + lane < qd) dst[q0 + lane] = R0[ring_ix(q0 + lane)]; q0 = qd; }
            flushed = q0 - shift;
            lap(8);
        }
    }
    if (PROF) { if (lane == 0) for (int i = 0; i < 9; ++i) prof[16ull * m + i] = pc[i]; }
}

// CRC-32 of every member's text in HBM against the footer: a wave a member, a lane 64 bytes of every 4 KiB tile (four table chains of 16
// bytes); the chains of a tile are joined by x^(8 16), a lane's tiles by x^(8 4096), the lanes by x^(8 64) doubling its exponent - the
// text counted from its END (zeros in front of it change nothing in a register run from zero), the initial value's term added last.
__global__ __launch_bounds__(64) void k_bgzf_crc(const uint8_t *__restrict__ text, const BgzfMember *__restrict__ mem, uint32_t n_mem, uint32_t *__restrict__ status,
                                                 CrcPowers X) {
    __shared__ uint32_t tab[256];
    const uint32_t lane = threadIdx.x, m = blockIdx.x;
    for (uint32_t i = lane; i < 256; i += 64) tab[i] = dfl::crc_table_entry(i);
    __syncthreads();
    if (m >= n_mem || status[m]) return;
    const BgzfMember M = mem[m];
    const uint32_t n = M.text_len;
    if (!n) { if (lane == 0 && M.crc != 0) status[m] = ST_CRC; return; }
    const uint32_t v = wave_crc_raw(tab, text + M.out_off, n, X);              // (crc_wave.h: the layout and the algebra)
    if (lane == 0 && (v ^ M.crc_init ^ 0xFFFFFFFFu) != M.crc) status[m] = ST_CRC;
}

struct InflateDev {
    DevBuf mem, text_len, out_off, bsum, status, n_tok, tok, scratch, prof;
    void *pin = nullptr; size_t pin_cap = 0;
    bool lds_set = false;
};

double wall() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

const char *status_text(uint32_t s) {
    switch (s) {
        case ifl::IFL_E_BTYPE: return "a deflate block of type 3";
        case ifl::IFL_E_STORED: return "a stored block whose LEN and NLEN disagree";
        case ifl::IFL_E_LENGTHS: return "code lengths that make no Huffman code";
        case ifl::IFL_E_SYMBOL: return "bits that are no code of the block";
        case ifl::IFL_E_DISTANCE: return "a copy from before the member's text";
        case ifl::IFL_E_TEXT: return "the text is not the ISIZE bytes the member's footer gives";
        case ifl::IFL_E_INPUT: return "the deflate stream runs past the member's end";
        case ifl::IFL_E_TOKENS: return "token list full";
        case ST_ISIZE: return "ISIZE above 64 KiB";
        case ST_CRC: return "CRC-32 of the text differs from the footer's";
    }
    return "unknown error";
}

}  // namespace

hipError_t scratch_take(int device, size_t bytes, DevBuf &b) {
    (void)device;                                       // (the block list of common.h is per current device)
    return b.reserve_exact(bytes);
}

void scratch_give(int device, DevBuf &b) {
    (void)device;
    b.release();
}

void inflate_release(void **state) {
    if (!state || !*state) return;
    InflateDev *I = static_cast<InflateDev *>(*state);
    for (DevBuf *b : {&I->mem, &I->text_len, &I->out_off, &I->bsum, &I->status, &I->n_tok, &I->tok, &I->scratch, &I->prof}) b->release();
    if (I->pin) (void)hipHostFree(I->pin);
    delete I;
    *state = nullptr;
}

int bgzf_inflate_device(pav_ctx *ctx, hipStream_t st, void **state, const uint8_t *d_comp, const BgzfMembers &M, DevBuf &out, uint64_t *n_text,
                        const char *what) {
    if (!*state) *state = new InflateDev();
    InflateDev *I = static_cast<InflateDev *>(*state);
    const bool timing = getenv("PAV_TIMING") != nullptr;
    const double t0 = wall();
    const uint32_t n = (uint32_t)M.in_off.size();
    *n_text = 0;
    if (!n) { PAV_HIP(ctx, scratch_take(ctx->device, 4096, out)); return PAV_OK; }
    const uint32_t batch = std::min(n, BATCH_MEMBERS);
    PAV_HIP(ctx, I->mem.reserve(sizeof(BgzfMember) * (size_t)n));
    PAV_HIP(ctx, I->text_len.reserve(4ull * n));
    PAV_HIP(ctx, I->out_off.reserve(8ull * (n + 8)));
    PAV_HIP(ctx, I->bsum.reserve(8ull * (n / SCAN_TILE + 8)));
    PAV_HIP(ctx, I->status.reserve(4ull * n));
    PAV_HIP(ctx, I->n_tok.reserve(4ull * batch));
    PAV_HIP(ctx, scratch_take(ctx->device, 4ull * TOK_STRIDE * batch, I->tok));   // (back on the list when this function returns, error or not)
    struct TokBack { pav_ctx *ctx; InflateDev *I; hipStream_t st; ~TokBack() { (void)hipStreamSynchronize(st); scratch_give(ctx->device, I->tok); } } tok_back{ctx, I, st};
    PAV_HIP(ctx, I->scratch.reserve(sizeof(ifl::LaneScratch) * (size_t)batch));
    const size_t pin_need = std::max<size_t>(sizeof(BgzfMember) * (size_t)n, 4096);
    if (I->pin_cap < pin_need) {
        if (I->pin) (void)hipHostFree(I->pin);
        I->pin = nullptr; I->pin_cap = 0;
        PAV_HIP(ctx, hipHostMalloc(&I->pin, pin_need + pin_need / 4, hipHostMallocDefault));
        I->pin_cap = pin_need + pin_need / 4;
    }
    if (!I->lds_set) {
        PAV_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_inflate_resolve<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RESOLVE_LDS));
        PAV_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_inflate_resolve<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RESOLVE_LDS));
        PAV_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_inflate_resolve_ring<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RING_LDS));
        PAV_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_inflate_resolve_ring<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RING_LDS));
        I->lds_set = true;
    }
    const bool profile = getenv("PAV_INFLATE_PROFILE") != nullptr;
    const bool full_window = [] { const char *e = getenv("PAV_INFLATE_WINDOW"); return e && !strcmp(e, "full"); }();
    if (profile) { PAV_HIP(ctx, I->prof.reserve(128ull * n)); PAV_HIP(ctx, hipMemsetAsync(I->prof.p, 0, 128ull * n, st)); }
    BgzfMember *hm = static_cast<BgzfMember *>(I->pin);
    for (uint32_t i = 0; i < n; ++i) hm[i] = BgzfMember{M.in_off[i], 0, M.in_len[i], 0, 0, 0, 0, 0};
    PAV_HIP(ctx, hipMemcpyAsync(I->mem.p, hm, sizeof(BgzfMember) * (size_t)n, hipMemcpyHostToDevice, st));
    PAV_LAUNCH_ON(ctx, st, "k_bgzf_footers", k_bgzf_footers, (n + 255) / 256, 256, 0, d_comp, I->mem.as<BgzfMember>(), n, I->text_len.as<uint32_t>(), I->status.as<uint32_t>());
    { const int rc = scan_u32_to_u64(st, I->text_len.as<uint32_t>(), n, I->bsum.as<uint64_t>(), I->out_off.as<uint64_t>());
      if (rc != PAV_OK) return fail(ctx, rc, "%s", pav_last_error(nullptr)); }
    PAV_LAUNCH_ON(ctx, st, "k_bgzf_places", k_bgzf_places, (n + 255) / 256, 256, 0, I->mem.as<BgzfMember>(), I->out_off.as<uint64_t>(), n);
    // the text's length: the last member's place + its ISIZE
    PAV_HIP(ctx, hipStreamSynchronize(st));             // (the pinned member table has crossed: the same block takes the answer)
    PAV_HIP(ctx, hipMemcpyAsync(hm, I->mem.as<BgzfMember>() + (n - 1), sizeof(BgzfMember), hipMemcpyDeviceToHost, st));
    PAV_HIP(ctx, hipStreamSynchronize(st));
    const uint64_t total = hm[0].out_off + hm[0].text_len;
    PAV_HIP(ctx, scratch_take(ctx->device, total + 4096, out));
    const double t1 = wall();
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    if (timing) { for (hipEvent_t &e : ev) PAV_HIP(ctx, hipEventCreate(&e)); PAV_HIP(ctx, hipEventRecord(ev[0], st)); }
    float ms_tok = 0, ms_res = 0;
    for (uint32_t at = 0; at < n; at += batch) {
        const uint32_t nb = std::min(batch, n - at);
        PAV_LAUNCH_ON(ctx, st, "k_inflate_tokens", k_inflate_tokens, (nb + 63) / 64, 64, 0, d_comp, I->mem.as<BgzfMember>() + at, nb, I->tok.as<uint32_t>(),
                      I->n_tok.as<uint32_t>(), I->status.as<uint32_t>() + at, I->scratch.as<ifl::LaneScratch>());
        if (timing) PAV_HIP(ctx, hipEventRecord(ev[1], st));
        // PAV_INFLATE_WINDOW=full: the first version of the kernel - the member's whole text in a 64 KiB window, two waves a CU, the CRC-32
        // taken from the window; the default holds a ring of 36 KiB, four waves a CU, and the CRC-32 is k_bgzf_crc's
        unsigned long long *pp = profile ? I->prof.as<unsigned long long>() + 16ull * at : nullptr;
        if (full_window) {
            if (profile) PAV_LAUNCH_ON(ctx, st, "k_inflate_resolve", k_inflate_resolve<true>, nb, 64, RESOLVE_LDS, I->tok.as<uint32_t>(), I->n_tok.as<uint32_t>(),
                                       I->mem.as<BgzfMember>() + at, I->status.as<uint32_t>() + at, out.as<uint8_t>(), pp);
            else PAV_LAUNCH_ON(ctx, st, "k_inflate_resolve", k_inflate_resolve<false>, nb, 64, RESOLVE_LDS, I->tok.as<uint32_t>(), I->n_tok.as<uint32_t>(),
                               I->mem.as<BgzfMember>() + at, I->status.as<uint32_t>() + at, out.as<uint8_t>(), pp);
        } else {
            if (profile) PAV_LAUNCH_ON(ctx, st, "k_inflate_resolve_ring", k_inflate_resolve_ring<true>, nb, 64, RING_LDS, I->tok.as<uint32_t>(), I->n_tok.as<uint32_t>(),
                                       I->mem.as<BgzfMember>() + at, I->status.as<uint32_t>() + at, out.as<uint8_t>(), pp);
            else PAV_LAUNCH_ON(ctx, st, "k_inflate_resolve_ring", k_inflate_resolve_ring<false>, nb, 64, RING_LDS, I->tok.as<uint32_t>(), I->n_tok.as<uint32_t>(),
                               I->mem.as<BgzfMember>() + at, I->status.as<uint32_t>() + at, out.as<uint8_t>(), pp);
        }
        if (timing) {                                   // (PAV_TIMING: the two kernels of every batch timed by events, the stream drained per batch)
            PAV_HIP(ctx, hipEventRecord(ev[2], st)); PAV_HIP(ctx, hipEventSynchronize(ev[2]));
            float a = 0, b = 0;
            PAV_HIP(ctx, hipEventElapsedTime(&a, ev[0], ev[1])); PAV_HIP(ctx, hipEventElapsedTime(&b, ev[1], ev[2]));
            ms_tok += a; ms_res += b;
            PAV_HIP(ctx, hipEventRecord(ev[0], st));
        }
    }
    float ms_crc = 0;
    if (!full_window) {

        if (timing) PAV_HIP(ctx, hipEventRecord(ev[0], st));
        PAV_LAUNCH_ON(ctx, st, "k_bgzf_crc", k_bgzf_crc, n, 64, 0, out.as<uint8_t>(), I->mem.as<BgzfMember>(), n, I->status.as<uint32_t>(), crc_powers());
        if (timing) { PAV_HIP(ctx, hipEventRecord(ev[1], st)); PAV_HIP(ctx, hipEventSynchronize(ev[1])); PAV_HIP(ctx, hipEventElapsedTime(&ms_crc, ev[0], ev[1])); }
    }
    uint32_t *hs = static_cast<uint32_t *>(I->pin);
    PAV_HIP(ctx, hipMemcpyAsync(hs, I->status.p, 4ull * n, hipMemcpyDeviceToHost, st));
    PAV_HIP(ctx, hipStreamSynchronize(st));
    for (uint32_t i = 0; i < n; ++i)
        if (hs[i]) return fail(ctx, PAV_E_ARG, "%s: corrupt BGZF member %u of %u (payload at byte %llu): %s", what, i, n, (unsigned long long)M.in_off[i], status_text(hs[i]));
    *n_text = total;
    if (timing) {
        fprintf(stderr, "[pav timing] bgzf_inflate_device: %u members, %.1f MB of text; places %.1f ms, inflate + crc %.1f ms (%.1f GB/s): k_inflate_tokens %.2f ms, %s %.2f ms, k_bgzf_crc %.2f ms\n",
                n, (double)total / 1e6, (t1 - t0) * 1e3, (wall() - t1) * 1e3, (double)total / 1e9 / std::max(1e-9, wall() - t1), ms_tok,
                full_window ? "k_inflate_resolve (+ CRC-32)" : "k_inflate_resolve_ring", ms_res, ms_crc);
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    }
    if (profile) {
        std::vector<unsigned long long> hp(16ull * n);
        PAV_HIP(ctx, hipMemcpy(hp.data(), I->prof.p, 128ull * n, hipMemcpyDeviceToHost));
        double sum[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (uint32_t i = 0; i < n; ++i) for (int k = 0; k < 9; ++k) sum[k] += (double)hp[16ull * i + k];
        fprintf(stderr, "[pav profile] k_inflate_resolve, mean of %u members: %.0f steps, %.0f copy-loop trips, %.0f lane-copy iterations, %.0f wave copies; cycles: "
                        "start %.0f, fetch + places + literals %.0f, copies %.0f, CRC-32 %.0f, window -> HBM %.0f\n", n, sum[0] / n, sum[1] / n, sum[2] / n, sum[3] / n,
                sum[4] / n, sum[5] / n, sum[6] / n, sum[7] / n, sum[8] / n);
    }
    return PAV_OK;
}

}  // namespace pav
