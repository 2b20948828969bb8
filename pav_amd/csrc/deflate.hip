// gzip on the device: text resident in HBM -> conforming gzip members (RFC 1951 / 1952), the output format of
// DataFrame.to_csv(compression='gzip') in rules call_cigar, call_cigar_merge and call_inv_batch (rules/call.snakefile:845-846,
// rules/call_inv.snakefile:279-291).  The reference compresses on one host core inside pandas; the host writers of rounds 1-4 did it
// with zlib on sixteen (1.5 s per haplotype at level 6: 2.2 GB of text).  Here:
//
//   * the text of every file is cut into 64 KiB segments; ONE WAVE encodes a segment, start to finish, with no barrier against any
//     other wave: a zlib-shaped sliding window in its LDS (a ring of 2 KiB of text, hash heads, chain links), 64 positions per step;
//   * matches may reach back into the text in front of the segment (it is all in HBM: the window is primed from it), so the file is
//     one deflate stream; a segment is one dynamic-Huffman block, followed - except the last - by an empty stored block that brings
//     the stream to a byte boundary (the way parallel gzip implementations join their pieces);
//   * chain insertion is exact in position order: the 64 lanes of a step link to the head before the step, the lanes that share a
//     hash inside the step are chained to each other by a ballot loop over the distinct duplicated hashes;
//   * match search walks the chain per lane; the lazy rule of zlib (take the match at p unless p + 1 has a longer one) is a
//     function of the per-position lengths, so the parse is a scalar walk over a step's `next position` values;
//   * tokens go to a scratch of the wave in HBM, counts to LDS; the trees, the block header (deflate_dev.h) and the bit stream -
//     per 64 tokens a prefix sum of code lengths and LDS atomic ORs - follow in the same wave;
//   * CRC-32 per segment in a second small kernel (64 lane-chunks joined by the checksum's own algebra), joined per file on the host;
//   * the segments' outputs are packed into one buffer on the device and cross PCIe once.
#include "common.h"
#include "deflate_dev.h"
#include "crc_wave.h"
#include "devgz.h"

#include <algorithm>
#include <mutex>
#include <chrono>

namespace pav {

namespace {

struct GzSegment {
    uint64_t text_off;        // arena offset of the segment's first byte (a multiple of 256)
    uint32_t len;             // bytes of text (0 only for an empty file)
    uint32_t hist;            // bytes of the same file in front of it that matches may reach (a multiple of 256)
    uint32_t last;            // last segment of its file
    uint32_t file;
};
struct GzSegOut { uint32_t bytes, crc; };      // bytes = 0xFFFFFFFF: the slot was too small (never seen; reported)

struct DeflateArgs {
    const uint8_t *text; uint64_t text_alloc;
    const GzSegment *segs; uint32_t n_segs;
    uint32_t *counter;
    uint32_t *tok; uint32_t tok_per_wave;
    uint8_t *slots; uint32_t slot_bytes;
    GzSegOut *out;
    uint32_t chain, lazy, nice, good;
};

__device__ __forceinline__ uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t byte_shift) { return __builtin_amdgcn_alignbyte(hi, lo, byte_shift); }

template <int WBITS, int HBITS>
__global__ __launch_bounds__(64) void k_deflate(DeflateArgs A) {
    constexpr uint32_t W = 1u << WBITS, WM = W - 1u, H = 1u << HBITS;
    constexpr uint32_t AHEAD = 64 + 320;                 // a step reads up to 258 + 3 bytes past its last position
    constexpr uint32_t MAXDIST = W - AHEAD - 256;        // the ring holds [loaded - W, loaded), loaded < step + AHEAD + 256
    // the window: ring of text | chain links | hash heads.  Behind the match phase it is dead: trees and header are built in its place
    constexpr uint32_t LZ_BYTES = W + 2 * W + 2 * H;
    static_assert(sizeof(dfl::HuffWork) + sizeof(dfl::HeaderWork) + 4 * 192 <= LZ_BYTES, "the tree scratch overlays the window");
    __shared__ __attribute__((aligned(16))) uint8_t lz[LZ_BYTES];
    uint32_t *T = reinterpret_cast<uint32_t *>(lz);
    uint16_t *prev = reinterpret_cast<uint16_t *>(lz + W);
    uint16_t *head = reinterpret_cast<uint16_t *>(lz + 3 * W);
    __shared__ uint32_t f_ll[288], f_d[32];
    __shared__ uint8_t ll_len[288], d_len[32];
    // The codes of both trees and the staging words of the bit stream live where the tree builder's scratch (HuffWork) was: they are
    // made after BOTH trees have their lengths.  9 792 bytes of LDS a wave instead of 11 584: sixteen waves a CU - what the registers
    // allow - instead of fourteen.
    static_assert(sizeof(dfl::HuffWork) >= 4 * (288 + 32 + 128), "codes and bit staging overlay the tree builder's scratch");
    uint32_t *c_ll = reinterpret_cast<uint32_t *>(lz), *c_d = c_ll + 288, *stage = c_d + 32;
    dfl::HuffWork &HW = *reinterpret_cast<dfl::HuffWork *>(lz);
    dfl::HeaderWork &XW = *reinterpret_cast<dfl::HeaderWork *>(lz + sizeof(dfl::HuffWork));
    uint32_t *hdr = reinterpret_cast<uint32_t *>(lz + sizeof(dfl::HuffWork) + sizeof(dfl::HeaderWork));

    const uint32_t lane = threadIdx.x;
    const uint64_t lane_lt = (1ull << lane) - 1ull;
    uint32_t *tok = A.tok + (size_t)blockIdx.x * A.tok_per_wave;

    auto get4 = [&](uint32_t pos) -> uint32_t {          // bytes pos .. pos + 3 of the stream, from the ring
        const uint32_t i = pos & WM, w = i >> 2;
        return alignbyte(T[(w + 1) & (W / 4 - 1)], T[w], i & 3u);
    };

    for (;;) {
        uint32_t seg = 0;
        if (lane == 0) seg = atomicAdd(A.counter, 1u);
        seg = (uint32_t)__builtin_amdgcn_readfirstlane((int)seg);
        if (seg >= A.n_segs) break;
        const GzSegment sg = A.segs[seg];
        const uint64_t base = sg.text_off - sg.hist;                    // stream position 0 (256-aligned in the arena)
        const uint32_t S = sg.hist, E = sg.hist + sg.len;
        for (uint32_t i = lane; i < LZ_BYTES / 4; i += 64) reinterpret_cast<uint32_t *>(lz)[i] = 0;
        for (uint32_t i = lane; i < 288; i += 64) f_ll[i] = 0;
        if (lane < 32) f_d[lane] = 0;
        __syncthreads();

        uint32_t loaded = 0, next_free = S, n_tok = 0;
        uint32_t pv_len = 0, pv_dist = 0, pv_byte = 0, pv_g = 0;
        bool have_pv = false;

        // tokens of the step at G from its per-position (length, distance, byte); next0 = the length found at G + 64
        auto parse = [&](uint32_t G, uint32_t len, uint32_t dist, uint32_t byte, uint32_t next0) {
            uint32_t nl = (uint32_t)__shfl_down((int)len, 1);
            if (lane == 63) nl = next0;
            const bool take = len >= 4u && !(len < A.lazy && nl > len);
            const uint32_t step = take ? len : 1u;
            uint32_t cur = next_free - G;
            uint64_t mask = 0;
            while (cur < 64u) {
                mask |= 1ull << cur;
                cur += (uint32_t)__builtin_amdgcn_readlane((int)step, (int)cur);
            }
            next_free = G + cur;
            if (E - G < 64u) mask &= (1ull << (E - G)) - 1ull;          // positions behind the segment's end are nobody's
            if (!mask) return;
            if ((mask >> lane) & 1ull) {
                const uint32_t at = n_tok + (uint32_t)__popcll(mask & lane_lt);
                if (take) {
                    tok[at] = 0x80000000u | len << 16 | dist;
                    uint32_t c, eb, ev;
                    dfl::len_symbol(len, c, eb, ev); atomicAdd(&f_ll[c], 1u);
                    dfl::dist_symbol(dist, c, eb, ev); atomicAdd(&f_d[c], 1u);
                } else {
                    tok[at] = byte;
                    atomicAdd(&f_ll[byte], 1u);
                }
            }
            n_tok += (uint32_t)__popcll(mask);
        };

        for (uint32_t g = 0; g < E; g += 64) {
            if (loaded < g + AHEAD) {
                while (loaded < g + AHEAD) {                            // 256 bytes per load, a dword per lane
                    const uint64_t off = base + loaded + 4ull * lane;
                    T[((loaded >> 2) + lane) & (W / 4 - 1)] = off + 4 <= A.text_alloc ? *reinterpret_cast<const uint32_t *>(A.text + off) : 0u;
                    loaded += 256;
                }
                __syncthreads();
            }
            const uint32_t p = g + lane, p16 = p & 0xFFFFu;
            const bool hashable = p + 4u <= E;
            const uint32_t w0 = get4(p);
            const uint32_t h = (w0 * 2654435761u) >> (32 - HBITS);
            // ---- link: nearest earlier position with the same hash -------------------------------------------------------
            uint32_t cand = p16;                                         // distance 0 = none
            if (hashable) cand = head[h];
            __syncthreads();
            if (hashable) head[h] = (uint16_t)p16;
            __syncthreads();
            const bool lost = hashable && head[h] != p16;                // somebody else of this step has my hash
            uint64_t dm = __ballot(lost);
            while (dm) {
                const int l = __ffsll((long long)dm) - 1;
                const uint32_t hl = (uint32_t)__builtin_amdgcn_readlane((int)h, l);
                const bool in = hashable && h == hl;
                const uint64_t eq = __ballot(in);
                if (in) {
                    const uint64_t lower = eq & lane_lt;
                    if (lower) cand = (g + 63u - (uint32_t)__clzll((long long)lower)) & 0xFFFFu;
                    if ((eq >> lane) == 1ull) head[h] = (uint16_t)p16;   // the last of them is the head the next step sees
                }
                dm &= ~eq;
            }
            if (hashable) prev[p & WM] = (uint16_t)cand;
            __syncthreads();
            if (g + 64 <= S) continue;                                   // the text in front of the segment only primes the window
            // ---- longest match along the chain ---------------------------------------------------------------------------
            // One flat loop, every lane in one of two states - at a link of its chain, or extending a candidate four bytes at a
            // time - so the wave's time is the longest lane's own work, not the sum over chain depths of the longest extension at
            // each depth.  A candidate is extended only if its first four bytes agree and so do the four bytes that END one past the
            // best length so far (it cannot win otherwise: zlib's scan_end test); links cost more once a good match is in hand.
            uint32_t best_len = 0, best_dist = 0;
            {
                bool active = p >= S && hashable, extending = false;
                const uint32_t maxlen = min(258u, E - p);
                uint32_t cur = cand, last = 0, budget = A.chain, q = 0, len = 0, qdist = 0;
                while (__any(active)) {
                    if (active) {
                        if (!extending) {
                            const uint32_t dist = (p - cur) & 0xFFFFu;
                            if (budget == 0 || dist == 0 || dist > MAXDIST || dist > p || dist <= last) active = false;
                            else {
                                last = dist; q = p - dist; qdist = dist;
                                budget -= best_len >= A.good ? min(budget, 4u) : 1u;
                                cur = prev[q & WM];
                                bool ok = get4(q) == w0;
                                if (ok && best_len >= 4u && best_len < maxlen) ok = get4(q + best_len - 3u) == get4(p + best_len - 3u);
                                if (ok) { extending = true; len = 4; }
                            }
                        } else {
                            bool done = len >= maxlen;
                            if (!done) {
                                const uint32_t x = get4(p + len) ^ get4(q + len);
                                if (x) { len += (uint32_t)(__ffs((int)x) - 1) >> 3; done = true; }
                                else len += 4;
                            }
                            if (done) {
                                len = min(len, maxlen);
                                if (len > best_len) { best_len = len; best_dist = qdist; if (len >= A.nice || len >= maxlen) active = false; }
                                extending = false;
                            }
                        }
                    }
                }
                // a four-byte match far away costs more bits than four literals of this kind of text
                if (best_len == 4u && best_dist > 2048u) best_len = 0;
            }
            if (have_pv) parse(pv_g, pv_len, pv_dist, pv_byte, (uint32_t)__builtin_amdgcn_readlane((int)best_len, 0));
            pv_g = g; pv_len = best_len; pv_dist = best_dist; pv_byte = w0 & 0xFFu; have_pv = true;
        }
        if (have_pv) parse(pv_g, pv_len, pv_dist, pv_byte, 0u);
        __syncthreads();
        if (lane == 0) f_ll[256] += 1u;                                  // end of block
        __syncthreads();

        // ---- the two trees ------------------------------------------------------------------------------------------
        for (int tree = 0; tree < 2; ++tree) {
            uint32_t *freq = tree == 0 ? f_ll : f_d;
            const int n = tree == 0 ? dfl::N_LL : dfl::N_D;
            uint32_t used = 0;
            for (int s0 = 0; s0 < n; s0 += 64) used += (uint32_t)__popcll(__ballot(s0 + (int)lane < n && freq[s0 + lane] != 0));
            if (used < 2 && lane == 0) {                                 // a complete code needs two symbols (RFC 1951 3.2.7)
                for (int s = 0; s < n && used < 2; ++s) if (!freq[s]) { freq[s] = 1; ++used; }
            }
            used = max(used, 2u);
            __syncthreads();
            if (lane == 0) HW.n_used = used;
            for (int s = (int)lane; s < n; s += 64) if (freq[s]) HW.order[dfl::huff_rank(freq, n, s)] = (uint16_t)s;
            __syncthreads();
            if (lane == 0) dfl::huff_lengths(freq, n, dfl::MAX_BITS, tree == 0 ? ll_len : d_len, HW);
            __syncthreads();
        }
        if (lane == 0) {                                                 // (HuffWork is dead: the codes go where it was)
            dfl::huff_codes(ll_len, dfl::N_LL, c_ll, XW.count, XW.next);
            dfl::huff_codes(d_len, dfl::N_D, c_d, XW.count, XW.next);
        }
        __syncthreads();
        // ---- header ---------------------------------------------------------------------------------------------------
        uint32_t hdr_bits = 0;
        if (lane == 0) {
            dfl::BitSink sink{hdr, 0};
            dfl::block_header(sink, sg.last != 0, ll_len, d_len, XW);
            if ((sink.n_bits & 31u) == 0) hdr[sink.n_bits >> 5] = 0;
            hdr_bits = sink.n_bits;
        }
        hdr_bits = (uint32_t)__builtin_amdgcn_readfirstlane((int)hdr_bits);
        __syncthreads();
        uint32_t *out_w = reinterpret_cast<uint32_t *>(A.slots + (size_t)seg * A.slot_bytes);
        const uint32_t slot_words = A.slot_bytes / 4;
        uint32_t words_done = hdr_bits >> 5, fill = hdr_bits & 31u;
        bool overflow = false;
        for (uint32_t i = lane; i < words_done; i += 64) out_w[i] = hdr[i];
        for (uint32_t i = lane; i < 128; i += 64) stage[i] = 0;
        __syncthreads();
        if (lane == 0) stage[0] = fill ? hdr[words_done] : 0u;
        __syncthreads();
        // ---- the bit stream: 64 tokens per step ---------------------------------------------------------------------------
        for (uint32_t t0 = 0; t0 <= n_tok; t0 += 64) {
            const uint32_t t = t0 + lane;
            uint64_t bits = 0; uint32_t nbits = 0;
            if (t < n_tok) {
                const uint32_t tk = tok[t];
                if (tk & 0x80000000u) {
                    const uint32_t len = (tk >> 16) & 0x1FFu, dist = tk & 0xFFFFu;
                    uint32_t c, eb, ev;
                    dfl::len_symbol(len, c, eb, ev);
                    const uint32_t cl = c_ll[c];
                    bits = cl & 0xFFFFu; nbits = cl >> 16;
                    bits |= (uint64_t)ev << nbits; nbits += eb;
                    dfl::dist_symbol(dist, c, eb, ev);
                    const uint32_t cd = c_d[c];
                    bits |= (uint64_t)(cd & 0xFFFFu) << nbits; nbits += cd >> 16;
                    bits |= (uint64_t)ev << nbits; nbits += eb;
                } else {
                    const uint32_t cl = c_ll[tk & 0xFFu];
                    bits = cl & 0xFFFFu; nbits = cl >> 16;
                }
            } else if (t == n_tok) {
                const uint32_t cl = c_ll[256];
                bits = cl & 0xFFFFu; nbits = cl >> 16;
            }
            uint32_t incl = nbits;
            for (int d = 1; d < 64; d <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, d); if ((int)lane >= d) incl += v; }
            const uint32_t total = fill + (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (nbits) {
                const uint32_t off = fill + incl - nbits, wi = off >> 5, sh = off & 31u;
                const uint64_t lo = bits << sh;
                const uint32_t hi = sh ? (uint32_t)(bits >> (64u - sh)) : 0u;
                if ((uint32_t)lo) atomicOr(&stage[wi], (uint32_t)lo);
                if ((uint32_t)(lo >> 32)) atomicOr(&stage[wi + 1], (uint32_t)(lo >> 32));
                if (hi) atomicOr(&stage[wi + 2], hi);
            }
            __syncthreads();
            const uint32_t nfull = total >> 5;                           // <= (31 + 64 * 48) / 32 = 96
            if (words_done + nfull + 4 > slot_words) { overflow = true; break; }
            for (uint32_t i = lane; i < nfull; i += 64) out_w[words_done + i] = stage[i];
            const uint32_t carry = stage[nfull];
            __syncthreads();
            for (uint32_t i = lane; i < nfull + 3 && i < 128; i += 64) stage[i] = 0;
            __syncthreads();
            if (lane == 0) stage[0] = carry;
            __syncthreads();
            words_done += nfull; fill = total & 31u;
        }
        // ---- end of the segment: the last one pads to a byte, the others append an empty stored block (000, pad, 00 00 FF FF) ------
        if (!overflow) {
            if (sg.last) fill = (fill + 7u) & ~7u;
            else {
                fill = (fill + 3u + 7u) & ~7u;
                if (lane == 0) {
                    const uint64_t v = (uint64_t)0xFFFF0000u << (fill & 31u);
                    stage[fill >> 5] |= (uint32_t)v;
                    stage[(fill >> 5) + 1] |= (uint32_t)(v >> 32);
                }
                fill += 32u;
            }
            __syncthreads();
            const uint32_t nw = (fill + 31u) >> 5;                       // <= 3
            if (lane < nw) out_w[words_done + lane] = stage[lane];
        }
        if (lane == 0) A.out[seg].bytes = overflow ? 0xFFFFFFFFu : words_done * 4u + (fill >> 3);
        __syncthreads();
    }
}

// CRC-32 of every segment, a wave a segment in the tile layout of crc_wave.h: a lane 64 bytes of every 4 KiB tile, every line of the
// text fetched once.  (Round 5 gave every lane one 1 KiB piece of the segment: a load of the wave touched 64 lines and every line came
// back sixteen times - 4.7 GB fetched for 0.85 GB of text, 26 GB/s.)
__global__ __launch_bounds__(64) void k_crc_segments(const uint8_t *__restrict__ text, const GzSegment *__restrict__ segs, uint32_t n_segs,
                                                     GzSegOut *__restrict__ out, CrcPowers X) {
    __shared__ uint32_t tab[256];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 256; i += 64) tab[i] = dfl::crc_table_entry(i);
    __syncthreads();
    const uint32_t seg = blockIdx.x;
    if (seg >= n_segs) return;
    const GzSegment sg = segs[seg];
    if (!sg.len) { if (lane == 0) out[seg].crc = 0u; return; }
    const uint32_t v = wave_crc_raw(tab, text + sg.text_off, sg.len, X);
    if (lane == 0) out[seg].crc = v ^ dfl::gf_mul(0xFFFFFFFFu, dfl::gf_xpow8(sg.len)) ^ 0xFFFFFFFFu;
}

// the segments' bytes, one behind the other where the host wants them (dst: byte offsets into `packed`)
__global__ __launch_bounds__(256) void k_gz_pack(const uint8_t *__restrict__ slots, uint32_t slot_bytes, const GzSegOut *__restrict__ so,
                                                 const uint64_t *__restrict__ dst, uint32_t n_segs, uint8_t *__restrict__ packed) {
    const uint32_t seg = blockIdx.x;
    if (seg >= n_segs) return;
    const uint32_t n = so[seg].bytes;
    const uint8_t *s = slots + (size_t)seg * slot_bytes;
    uint8_t *d = packed + dst[seg];
    for (uint32_t i = threadIdx.x; i < n; i += 256) d[i] = s[i];
}

struct GzState {
    DevBuf segs, out, tok, slots, counter, dst, packed, text;   // text: the uploads of pav_gzip_buffers
    void *h_out = nullptr; size_t h_out_cap = 0;          // pinned: per-segment sizes + checksums
    void *h_packed = nullptr; size_t h_packed_cap = 0;    // the files (ordinary memory: host_reserve)
    std::vector<GzSegment> h_segs;
    std::vector<uint64_t> h_dst;
    int waves = 0, waves_wbits = 0;
};

// The files' bytes come back into ordinary memory: a device-to-host copy into it runs at the speed of one into pinned memory on
// this platform (56 GB/s, tools/ubench/pin_cost.hip), and 300 MB of pinned memory cost 50 ms to get and 25 ms to give back - per
// context.  The blocks outlive their context (a process-wide list, at most 1 GiB kept): pages touched once stay touched.
struct HostBlocks { std::mutex mu; std::vector<std::pair<void *, size_t>> idle; };
HostBlocks &host_blocks() { static HostBlocks *P = new HostBlocks(); return *P; }
void host_give(void *&p, size_t &cap) {
    if (!p) return;
    HostBlocks &H = host_blocks();
    {
        std::lock_guard<std::mutex> lk(H.mu);
        size_t held = 0;
        for (auto &b : H.idle) held += b.second;
        if (held + cap <= (1ull << 30)) { H.idle.emplace_back(p, cap); p = nullptr; cap = 0; return; }
    }
    free(p); p = nullptr; cap = 0;
}
int host_reserve(void *&p, size_t &cap, size_t bytes) {
    if (bytes <= cap) return PAV_OK;
    host_give(p, cap);
    HostBlocks &H = host_blocks();
    {
        std::lock_guard<std::mutex> lk(H.mu);
        size_t best = H.idle.size();
        for (size_t i = 0; i < H.idle.size(); ++i)
            if (H.idle[i].second >= bytes && (best == H.idle.size() || H.idle[i].second < H.idle[best].second)) best = i;
        if (best < H.idle.size()) { p = H.idle[best].first; cap = H.idle[best].second; H.idle.erase(H.idle.begin() + (long)best); return PAV_OK; }
    }
    const size_t want = bytes + bytes / 4 + 4096;
    void *q = nullptr;
    if (posix_memalign(&q, 4096, want) != 0 || !q) return fail(nullptr, PAV_E_LIMIT, "gz_files: out of host memory (%zu bytes)", want);
    p = q; cap = want;
    return PAV_OK;
}

int pin_reserve(void *&p, size_t &cap, size_t bytes) {
    if (bytes <= cap) return PAV_OK;
    if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
    const size_t want = bytes + bytes / 4 + 4096;
    W_HIP(hipHostMalloc(&p, want, hipHostMallocDefault));
    cap = want;
    return PAV_OK;
}

}  // namespace

void gz_release_slot(pav_ctx *ctx, void **slot) {
    if (!ctx || !slot || !*slot) return;
    GzState *G = static_cast<GzState *>(*slot);
    (void)hipSetDevice(ctx->device);
    for (DevBuf *b : {&G->segs, &G->out, &G->tok, &G->slots, &G->counter, &G->dst, &G->packed, &G->text}) b->release();
    if (G->h_out) (void)hipHostFree(G->h_out);
    host_give(G->h_packed, G->h_packed_cap);
    delete G;
    *slot = nullptr;
}
void gz_release(pav_ctx *ctx) { if (ctx) gz_release_slot(ctx, &ctx->gz); }

int gz_files(pav_ctx *ctx, void **slot, hipStream_t st, const uint8_t *d_text, uint64_t text_alloc, const std::vector<GzFile> &files, int level,
             GzOut &out) {
    out.host = nullptr; out.off.assign(files.size(), 0); out.len.assign(files.size(), 0);
    if (files.empty()) return PAV_OK;
    if (!*slot) *slot = new GzState();
    GzState *G = static_cast<GzState *>(*slot);
    const bool timing = getenv("PAV_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_a = now();
    // window of the match finder: 2 KiB of text behind a position (the rows of these tables repeat their neighbours, not text 30 KiB
    // back) keeps a wave's LDS at 11.6 KB - thirteen waves per CU; PAV_GZ_WBITS=12 / 13: 4 / 8 KiB (eight / four waves per CU)
    const int wbits = [] { const char *e = getenv("PAV_GZ_WBITS"); const int v = e ? atoi(e) : 0; return v == 13 ? 13 : (v == 12 ? 12 : 11); }();
    const uint32_t HIST = wbits == 13 ? 4096u : (wbits == 12 ? 3328u : 1280u);   // text in front of a segment that primes its window (< the window's reach)
    std::vector<GzSegment> &segs = G->h_segs;
    segs.clear();
    std::vector<uint32_t> first(files.size() + 1, 0);
    for (size_t f = 0; f < files.size(); ++f) {
        const GzFile &F = files[f];
        if (F.text_off % GZ_TEXT_ALIGN) return fail(nullptr, PAV_E_ARG, "gz_files: file %zu does not start on a %llu-byte boundary", f, (unsigned long long)GZ_TEXT_ALIGN);
        if (F.text_off + F.text_len + GZ_TEXT_PAD > text_alloc) return fail(nullptr, PAV_E_ARG, "gz_files: file %zu ends less than %llu bytes before the end of the arena", f, (unsigned long long)GZ_TEXT_PAD);
        first[f] = (uint32_t)segs.size();
        const uint64_t n = std::max<uint64_t>(1, (F.text_len + GZ_SEGMENT - 1) / GZ_SEGMENT);
        for (uint64_t s = 0; s < n; ++s) {
            const uint64_t a = s * GZ_SEGMENT, b = std::min<uint64_t>(F.text_len, a + GZ_SEGMENT);
            segs.push_back(GzSegment{F.text_off + a, (uint32_t)(b - a), (uint32_t)std::min<uint64_t>(a, HIST), s + 1 == n ? 1u : 0u, (uint32_t)f});
        }
    }
    first[files.size()] = (uint32_t)segs.size();
    const uint32_t n_segs = (uint32_t)segs.size();
    if (level <= 0) level = 6;
    DeflateArgs A{};
    // how hard to look: chain steps per position, the length from which a match is taken without looking at the next position,
    // the length that ends a chain walk (zlib's max_chain / max_lazy / nice_length, scaled to what a lock-step wave can afford)
    // (measured on 200 MB of SNV rows / 150 MB of density rows, profiles/r05_gzip_variants.json.  Window 2 KiB - 11.6 KB of LDS,
    //  thirteen waves per CU - chain 4 / 6 / 8: 14.3 / 13.0 / 11.8 GB/s on SNV rows at 0.985 / 0.977 / 0.970 of zlib level 6's size,
    //  19.0 / 16.9 / 15.2 GB/s on density rows at 0.948 / 0.943 / 0.941; window 4 KiB - eight waves per CU - is 15 - 35 % slower at
    //  the same sizes; 8 KiB slower still.  Chain 1 / 2 / 3: 20.3 / 16.1 / 15.1 GB/s on SNV rows at 1.016 / 0.998 / 0.990 of zlib
    //  level 6's size, 26.3 / 22.4 / 20.6 GB/s on density rows at 0.979 / 0.959 / 0.952.)
    // Level 6 = three chain steps: 1 - 5 % below zlib-6's size on the haplotype's own rows and a sixth to a fifth faster than the six
    // steps of the first version, which bought 1 % of size (two steps: on rows of random positions - tests/test_gpu_gzip.py - 10 %
    // above zlib-6).  Levels below it: one step (zlib-1 .. 5 are 10 - 28 % larger than that); 7 and above: thirty-two.
    if (level <= 5) { A.chain = 1; A.lazy = 16; A.nice = 64; A.good = 16; }
    else if (level <= 6) { A.chain = 3; A.lazy = 32; A.nice = 128; A.good = 32; }
    else { A.chain = 32; A.lazy = 258; A.nice = 258; A.good = 64; }
    if (const char *e = getenv("PAV_GZ_CHAIN")) A.chain = (uint32_t)std::max(1, atoi(e));
    if (const char *e = getenv("PAV_GZ_NICE")) A.nice = (uint32_t)std::max(4, atoi(e));

    W_HIP(hipSetDevice(ctx->device));
    if (!G->waves || G->waves_wbits != wbits) {
        int per_cu = 0;
        if (wbits == 13) W_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_deflate<13, 12>, 64, 0));
        else if (wbits == 12) W_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_deflate<12, 11>, 64, 0));
        else W_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_deflate<11, 10>, 64, 0));
        // The waves are persistent and fill their CUs for the whole launch.  With every register of a SIMD taken (sixteen waves a
        // CU: 4 x 112 of 512 registers a lane, 64 left) no other kernel's wave fits beside them - the text kernel of the next group of
        // tables, the CRC kernel, another haplotype's kernels wait for the launch to end: with sixteen the density tables of the bench
        // took 0.18 s instead of 0.10.  Fourteen leave a wave's worth of registers on two SIMDs of every CU.  PAV_GZ_WAVES_PER_CU.
        static const int cap = [] { const char *e = getenv("PAV_GZ_WAVES_PER_CU"); return e ? std::max(1, atoi(e)) : 14; }();
        per_cu = std::min(per_cu, cap);
        G->waves = std::max(1, per_cu) * std::max(1, ctx->n_cu);
        G->waves_wbits = wbits;
        if (timing) fprintf(stderr, "[pav timing] gz_files: window 2^%d bytes, %d waves per CU, %d CUs\n", wbits, per_cu, ctx->n_cu);
    }
    const uint32_t waves = std::min<uint32_t>((uint32_t)G->waves, n_segs);
    const uint32_t slot_bytes = GZ_SEGMENT + GZ_SEGMENT / 2 + 1024;      // a match can cost more bits than its bytes as literals
    W_HIP(G->segs.reserve(sizeof(GzSegment) * n_segs));
    W_HIP(G->out.reserve(sizeof(GzSegOut) * n_segs));
    W_HIP(G->dst.reserve(8ull * n_segs));
    W_HIP(G->tok.reserve(4ull * GZ_SEGMENT * waves));
    W_HIP(G->slots.reserve((size_t)slot_bytes * n_segs));
    W_HIP(G->counter.reserve(64));
    { int rc = pin_reserve(G->h_out, G->h_out_cap, sizeof(GzSegOut) * n_segs); if (rc != PAV_OK) return rc; }
    W_HIP(hipMemcpyAsync(G->segs.p, segs.data(), sizeof(GzSegment) * n_segs, hipMemcpyHostToDevice, st));
    W_HIP(hipMemsetAsync(G->counter.p, 0, 64, st));
    A.text = d_text; A.text_alloc = text_alloc; A.segs = G->segs.as<GzSegment>(); A.n_segs = n_segs; A.counter = G->counter.as<uint32_t>();
    A.tok = G->tok.as<uint32_t>(); A.tok_per_wave = GZ_SEGMENT; A.slots = G->slots.as<uint8_t>(); A.slot_bytes = slot_bytes;
    A.out = G->out.as<GzSegOut>();
    if (wbits == 13) W_LAUNCH(st, (k_deflate<13, 12>), waves, 64, 0, A);
    else if (wbits == 12) W_LAUNCH(st, (k_deflate<12, 11>), waves, 64, 0, A);
    else W_LAUNCH(st, (k_deflate<11, 10>), waves, 64, 0, A);
    W_LAUNCH(st, k_crc_segments, n_segs, 64, 0, d_text, G->segs.as<GzSegment>(), n_segs, G->out.as<GzSegOut>(), crc_powers());
    W_HIP(hipMemcpyAsync(G->h_out, G->out.p, sizeof(GzSegOut) * n_segs, hipMemcpyDeviceToHost, st));
    W_HIP(hipStreamSynchronize(st));
    const GzSegOut *so = static_cast<const GzSegOut *>(G->h_out);
    const double t_b = now();
    // layout of the files: 10-byte header | the segments | CRC-32, ISIZE
    std::vector<uint64_t> &dst = G->h_dst;
    dst.resize(n_segs);
    uint64_t at = 0;
    std::vector<uint32_t> crc(files.size(), 0);
    for (size_t f = 0; f < files.size(); ++f) {
        at = (at + 63) / 64 * 64;
        out.off[f] = at;
        at += 10;
        for (uint32_t s = first[f]; s < first[f + 1]; ++s) {
            if (so[s].bytes == 0xFFFFFFFFu) return fail(nullptr, PAV_E_LIMIT, "gz_files: segment %u of file %zu outgrew its slot of %u bytes", s - first[f], f, slot_bytes);
            dst[s] = at; at += so[s].bytes;
            crc[f] = dfl::crc_join(crc[f], so[s].crc, segs[s].len);
        }
        at += 8;
        out.len[f] = at - out.off[f];
    }
    W_HIP(G->packed.reserve(at + 64));
    { int rc = host_reserve(G->h_packed, G->h_packed_cap, at + 64); if (rc != PAV_OK) return rc; }
    W_HIP(hipMemcpyAsync(G->dst.p, dst.data(), 8ull * n_segs, hipMemcpyHostToDevice, st));
    W_LAUNCH(st, k_gz_pack, n_segs, 256, 0, G->slots.as<uint8_t>(), slot_bytes, G->out.as<GzSegOut>(), G->dst.as<uint64_t>(),
                  n_segs, G->packed.as<uint8_t>());
    W_HIP(hipMemcpyAsync(G->h_packed, G->packed.p, at, hipMemcpyDeviceToHost, st));
    W_HIP(hipStreamSynchronize(st));
    uint8_t *hp = static_cast<uint8_t *>(G->h_packed);
    for (size_t f = 0; f < files.size(); ++f) {
        static const uint8_t hdr[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 0xff};   // deflate, no name, no time, unknown system
        memcpy(hp + out.off[f], hdr, 10);
        uint8_t *tr = hp + out.off[f] + out.len[f] - 8;
        const uint32_t isize = (uint32_t)files[f].text_len;
        for (int k = 0; k < 4; ++k) { tr[k] = (uint8_t)(crc[f] >> (8 * k)); tr[4 + k] = (uint8_t)(isize >> (8 * k)); }
    }
    out.host = hp;
    if (timing) {
        uint64_t tin = 0; for (const GzFile &F : files) tin += F.text_len;
        fprintf(stderr, "[pav timing] gz_files: %zu files, %u segments, %.1f MB -> %.1f MB; deflate + crc %.1f ms (%.1f GB/s), pack + copy %.1f ms\n", files.size(), n_segs,
                (double)tin / 1e6, (double)at / 1e6, (t_b - t_a) * 1e3, (double)tin / 1e9 / std::max(1e-9, t_b - t_a), (now() - t_b) * 1e3);
    }
    return PAV_OK;
}

}  // namespace pav

using namespace pav;

extern "C" {

// gzip of host buffers on the device (the table writers' compressor, for tests and for callers with text of their own): uploads
// the n texts, encodes them in one launch set, returns the n gzip members one behind the other in `out` (out_off / out_len say
// where; PAV_E_LIMIT when out_cap is too small: sum of the lengths x 1.5 + 4096 per text always suffices).
int pav_gzip_buffers(pav_ctx *ctx, uint32_t n, const uint8_t *const *texts, const uint64_t *lens, int level, uint8_t *out, uint64_t out_cap,
                     uint64_t *out_off, uint64_t *out_len) {
    if (!ctx || (n && (!texts || !lens || !out || !out_off || !out_len))) return fail(ctx, PAV_E_ARG, "pav_gzip_buffers: null argument");
    if (!n) return PAV_OK;
    PAV_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->gz) ctx->gz = new GzState();
    GzState *G = static_cast<GzState *>(ctx->gz);
    std::vector<GzFile> files(n);
    uint64_t at = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (lens[i] && !texts[i]) return fail(ctx, PAV_E_ARG, "pav_gzip_buffers: text %u is null", i);
        at = (at + GZ_TEXT_ALIGN - 1) / GZ_TEXT_ALIGN * GZ_TEXT_ALIGN;
        files[i] = GzFile{at, lens[i]};
        at += lens[i];
    }
    const uint64_t alloc = (at + GZ_TEXT_PAD + 255) / 256 * 256;
    PAV_HIP(ctx, G->text.reserve(alloc));
    PAV_HIP(ctx, hipMemsetAsync(G->text.p, 0, alloc, ctx->stream));
    for (uint32_t i = 0; i < n; ++i)
        if (lens[i]) PAV_HIP(ctx, hipMemcpyAsync(G->text.as<uint8_t>() + files[i].text_off, texts[i], lens[i], hipMemcpyHostToDevice, ctx->stream));
    GzOut g;
    const int rc = gz_files(ctx, &ctx->gz, ctx->stream, G->text.as<uint8_t>(), alloc, files, level, g);
    if (rc != PAV_OK) return fail(ctx, rc, "%s", pav_last_error(nullptr));
    uint64_t o = 0;
    for (uint32_t i = 0; i < n; ++i) {
        out_off[i] = o; out_len[i] = g.len[i];
        if (o + g.len[i] > out_cap) return fail(ctx, PAV_E_LIMIT, "pav_gzip_buffers: the output needs more than the %llu bytes given", (unsigned long long)out_cap);
        memcpy(out + o, g.host + g.off[i], g.len[i]);
        o += g.len[i];
    }
    return PAV_OK;
}

int pav_gzip_buffer(pav_ctx *ctx, const uint8_t *text, uint64_t n, int level, uint8_t *out, uint64_t out_cap, uint64_t *out_len) {
    if (!ctx || (n && !text) || !out || !out_len) return fail(ctx, PAV_E_ARG, "pav_gzip_buffer: null argument");
    uint64_t off = 0;
    const uint8_t *texts[1] = {text};
    return pav_gzip_buffers(ctx, 1, texts, &n, level, out, out_cap, &off, out_len);
}

}  // extern "C"
