// CIGAR variant calling on gfx950: device tokenizer, flat prefix-scan walk, SNV / INDEL emission with
// deterministic (row, op, base) ordering, breakpoint-homology kernel, SEQ gather.
//
// Replaces (file:line in the PAV 2.4.6 snapshot):
//   pavlib/align/align.py:286-322    cigar_str_to_tuples          -> tok_count / tok_emit / row_ops
//   pavlib/cigarcall.py:78-93,286    position bookkeeping          -> walk_reduce / walk_chunks / row_base
//   pavlib/cigarcall.py:95-139       SNV rows                      -> walk_emit (X ops)
//   pavlib/cigarcall.py:141-282      INS / DEL rows                -> walk_emit (I/D ops) + homology_kernel
//   pavlib/call.py:542-647           left / right homology         -> left_hom / right_hom
// Integer / byte work only: no MFMA.  All kernels are launched on the context's stream.
#include "common.h"

namespace pav {

constexpr int TOK_CHUNK = 4096;          // text bytes per workgroup (256 lanes x 16 B)
constexpr int OPS_PER_LANE = 8;
constexpr int WALK_CHUNK = 256 * OPS_PER_LANE;   // CIGAR ops per workgroup
constexpr int NQ = 6;                    // scanned quantities: ref_adv, tig_adv, n_snv, n_indel, seq_bytes, aligned


__device__ __forceinline__ bool is_digit(uint32_t c) { return c - (uint32_t)'0' <= 9u; }

__device__ __forceinline__ int op_code_of(uint32_t c) {
    switch (c) {
        case 'M': return 0; case 'I': return 1; case 'D': return 2; case 'N': return 3; case 'S': return 4;
        case 'H': return 5; case 'P': return 6; case '=': return 7; case 'X': return 8; default: return -1;
    }
}

// ---- block-wide exclusive scan of N u64 values per lane (256 lanes = 4 waves) ------------------------------
template <int N, int NW = 4>
__device__ __forceinline__ void block_excl_scan(uint64_t (&v)[N], uint64_t (&total)[N], uint64_t *lds /* NW*N */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t inc[N];
#pragma unroll
    for (int q = 0; q < N; ++q) {
        uint64_t x = v[q];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint64_t y = __shfl_up(x, d);
            if (lane >= d) x += y;
        }
        inc[q] = x;
        if (lane == 63) lds[wave * N + q] = x;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < N; ++q) {
        uint64_t base = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            uint64_t s = lds[w * N + q];
            if (w < wave) base += s;
            tot += s;
        }
        total[q] = tot;
        v[q] = base + inc[q] - v[q];
    }
    __syncthreads();
}

// ---- tokenizer -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tok_count(const uint4 *__restrict__ text, uint32_t *__restrict__ chunk_cnt) {
    __shared__ uint32_t s_cnt[4];
    const uint4 v = text[(uint64_t)blockIdx.x * 256 + threadIdx.x];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int b = 0; b < 4; ++b) c += !is_digit((w[k] >> (8 * b)) & 0xFFu);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) chunk_cnt[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// Single-workgroup exclusive scan of u32 counts into u64 prefixes; out[n] = total.  The scans sit on the critical path of a
// step between two kernels, beside the HBM-saturating pack: what they cost is dependent round trips to memory.  A tile of 8192
// counts is fetched with 32 independent coalesced loads per lane into LDS, every lane then owns 32 consecutive counts (one
// block scan per tile instead of one per 256 counts: 43 -> ~8 us for the 6.4 k counts of a haplotype).  256 lanes on purpose:
// a 1024-lane workgroup is not scheduled before the pack drains (it needs four free wave slots on every SIMD of one CU).
constexpr int SCAN_PER = 32;
__global__ __launch_bounds__(256) void scan_counts(const uint32_t *__restrict__ in, uint64_t *__restrict__ out,
                                                   uint32_t n, volatile uint64_t *host_total) {
    __shared__ uint64_t lds[4];
    __shared__ uint32_t tile[256 * (SCAN_PER + 1)];                       // + 1: a lane's run starts in its own bank
    uint64_t carry = 0;
    for (uint32_t base = 0; base < n; base += 256 * SCAN_PER) {
        uint32_t c[SCAN_PER];
#pragma unroll
        for (int k = 0; k < SCAN_PER; ++k) { const uint32_t i = base + k * 256 + threadIdx.x; c[k] = i < n ? in[i] : 0u; }
#pragma unroll
        for (int k = 0; k < SCAN_PER; ++k) { const uint32_t j = k * 256 + threadIdx.x; tile[j + j / SCAN_PER] = c[k]; }
        __syncthreads();
        uint64_t v[1] = {0}, tot[1];
#pragma unroll
        for (int k = 0; k < SCAN_PER; ++k) { c[k] = tile[threadIdx.x * (SCAN_PER + 1) + k]; v[0] += c[k]; }
        block_excl_scan<1>(v, tot, lds);
        uint64_t run = carry + v[0];
#pragma unroll
        for (int k = 0; k < SCAN_PER; ++k) {
            const uint32_t i = base + threadIdx.x * SCAN_PER + k;
            if (i < n) out[i] = run;
            run += c[k];
        }
        carry += tot[0];
    }
    if (threadIdx.x == 0) {
        out[n] = carry;
        if (host_total) *host_total = carry;     // straight into the host's pinned word: see walk_chunks
    }
}

// Each op character parses the digits in front of it and writes ops[ordinal] = len << 4 | code.
// Tokenizer errors: atomicMin of (byte position of the token start << 3 | kind); the smallest position is the
// first error the sequential reference would hit (rows are concatenated in table order).
__global__ __launch_bounds__(256) void tok_emit(const uint8_t *__restrict__ text, const uint64_t *__restrict__ chunk_pre,
                                                uint32_t *__restrict__ ops, unsigned long long *__restrict__ tok_err) {
    __shared__ uint64_t lds[4];
    __shared__ __attribute__((aligned(16))) uint8_t s_txt[32 + TOK_CHUNK];   // 32 B halo: digits of a token that starts
    const uint64_t b0 = (uint64_t)blockIdx.x * TOK_CHUNK;                     // in the previous workgroup's text
    const uint64_t p0 = b0 + (uint64_t)threadIdx.x * 16;
    const uint4 v = *reinterpret_cast<const uint4 *>(text + p0);
    *reinterpret_cast<uint4 *>(s_txt + 32 + threadIdx.x * 16) = v;
    if (threadIdx.x < 2) {
        uint4 h = make_uint4(0x2a2a2a2a, 0x2a2a2a2a, 0x2a2a2a2a, 0x2a2a2a2a);     // '*': not a digit
        if (blockIdx.x > 0) h = *reinterpret_cast<const uint4 *>(text + b0 - 32 + threadIdx.x * 16);
        *reinterpret_cast<uint4 *>(s_txt + threadIdx.x * 16) = h;
    }
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t flags = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) flags |= (uint32_t)!is_digit((w[j >> 2] >> (8 * (j & 3))) & 0xFFu) << j;
    uint64_t cnt[1] = {(uint64_t)__popc(flags)}, tot[1];
    block_excl_scan<1>(cnt, tot, lds);                  // its barriers also publish s_txt
    uint64_t ord = chunk_pre[blockIdx.x] + cnt[0];
    while (flags) {
        const int j = __ffs((int)flags) - 1;
        flags &= flags - 1;
        const int lp = 32 + (int)threadIdx.x * 16 + j;  // position of the op character in s_txt
        const uint32_t ch = s_txt[lp];
        // digits in front of the op character: nine of them fit 32-bit arithmetic (64-bit multiplies are several instructions
        // each) and are more than the 2^28 limit allows; any non-zero digit beyond them is an overflow as well
        uint32_t val = 0, mul = 1;
        int nd = 0;
        bool over = false;
        int q = lp - 1;
        while (q >= 0 && is_digit(s_txt[q])) {
            if (nd < 9) { val += (uint32_t)(s_txt[q] - '0') * mul; mul *= 10; } else if (s_txt[q] != '0') over = true;
            ++nd; --q;
        }
        int64_t g = (int64_t)b0 - 32 + q;               // global position of the first non-digit before the token
        if (q < 0 && blockIdx.x > 0) {                  // more than 32 + j digits: keep walking in global memory
            while (g >= 0 && is_digit(text[g])) { if (text[g] != '0') over = true; ++nd; --g; }
        } else if (q < 0) g = -1;
        const int code = op_code_of(ch);
        int kind = 0;
        if (nd == 0) kind = PAV_CIGAR_ERR_MISSING_LEN;              // align.py:310 (checked before the op set)
        else if (code < 0) kind = PAV_CIGAR_ERR_UNKNOWN_OP;         // align.py:315
        else if (over || val >= (1u << 28)) kind = PAV_CIGAR_ERR_LEN_OVERFLOW;
        if (kind) atomicMin(tok_err, (unsigned long long)(((uint64_t)(g + 1)) << 3 | (uint64_t)kind));
        ops[ord] = kind ? 0x7u /* a zero-length '=' keeps the stream well formed; the call fails anyway */ : (val << 4 | (uint32_t)code);
        ++ord;
    }
}

// One wave per row boundary r in [0, n_aln]: op_off[r] = number of op characters before text_off[r];
// also flags rows whose text ends inside a length (IndexError in the reference).
__global__ __launch_bounds__(256) void row_ops(const uint8_t *__restrict__ text, const uint64_t *__restrict__ text_off,
                                               const uint64_t *__restrict__ chunk_pre, uint64_t *__restrict__ op_off,
                                               uint32_t n_aln, unsigned long long *__restrict__ tok_err) {
    const uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r > n_aln) return;
    const int lane = threadIdx.x & 63;
    const uint64_t p = text_off[r];
    const uint64_t c = p / TOK_CHUNK;
    uint32_t cnt = 0;
    for (uint64_t q = c * TOK_CHUNK + lane; q < p; q += 64) cnt += !is_digit(text[q]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);
    if (lane != 0) return;
    op_off[r] = chunk_pre[c] + cnt;
    if (r < n_aln) {
        const uint64_t e = text_off[r + 1];
        if (e > p && is_digit(text[e - 1])) {
            uint64_t q = e - 1;
            while (q > p && is_digit(text[q - 1])) --q;
            atomicMin(tok_err, (unsigned long long)(q << 3 | (uint64_t)PAV_CIGAR_ERR_TRUNCATED));
        }
    }
}

__device__ __forceinline__ void op_contrib(uint32_t op, uint64_t (&c)[NQ]) {
    const uint32_t code = op & 15u;
    const uint64_t len = op >> 4;
    const bool eq = code == 7, x = code == 8, ins = code == 1, del = code == 2, clip = code == 4 || code == 5;
    c[0] = (eq || x || del) ? len : 0;            // reference advance  (cigarcall.py:92,138,282)
    c[1] = (eq || x || ins || clip) ? len : 0;    // query advance      (cigarcall.py:93,139,213,287)
    c[2] = x ? len : 0;                           // SNV rows
    c[3] = (ins || del) ? 1 : 0;                  // INS/DEL rows
    c[4] = (ins || del) ? len : 0;                // SEQ bytes
    c[5] = (eq || x) ? len : 0;                   // aligned bases (metric numerator)
}

// ---- tile tokenizer (pav_cigar_call) ------------------------------------------------------------------------
// The call path tokenises and sums in ONE pass and has no host-visible intermediate: a 4096-byte text tile holds at most 2048
// valid operations (>= one digit + the operation character each), so tile t owns the 2048 operation slots
// [t * 2048, (t + 1) * 2048) of a PADDED operation array - its operations first, OP_PAD (no operation) behind them - and the
// 2048-operation chunk every walk / verify workgroup works on IS the tile: nothing has to be counted before an operation can be
// stored, and the exclusive prefix a chunk needs is the prefix over the tiles before it (tile_scan, one workgroup per
// quantity).  "Slot ordinal" s = t * 2048 + i is monotone in walk order; the real ordinal of the operation in slot s is
// tile_pre[NQ][t] + i.  (Round 2: tok_count -> scan_counts -> host sync -> tok_emit -> row_ops -> walk_reduce -> walk_chunks ->
// row_base -> row_check = eight launches and a synchronisation for what tok_tiles + tile_scan do; a single-pass scan with
// decoupled look-back was not taken: a hand-off between workgroups costs 1 - 3 us on this chip (per-XCD L2s, agent-scope
// release / acquire) and 8.5 k tiles advance 64 per hop at best - slower than one more kernel boundary.)
constexpr uint32_t OP_PAD = 0xFu;        // code 15, length 0: contributes nothing, is no operation (op_contrib, walk_emit)
constexpr int NT = NQ + 1;               // per-tile sums: the NQ walk quantities + [NQ] the operation count
static_assert(TOK_CHUNK == 2 * WALK_CHUNK, "a text tile owns one chunk of operation slots");

constexpr uint64_t op_lut(bool hi) {     // nibble (c & 15) of the half selected by (c & 16): BAM code of the character, 15 = none
    uint64_t v = ~0ull;
    const char ops[] = "MIDNSHP=X";
    for (int k = 0; k < 9; ++k) {
        const int idx = ops[k] & 31;
        if ((idx >= 16) == hi) { const int sh = 4 * (idx & 15); v = (v & ~(0xFull << sh)) | ((uint64_t)k << sh); }
    }
    return v;
}
__device__ __forceinline__ uint32_t op_code_lut(uint32_t c) {      // the nine characters differ in (c & 31); '=' is the one below 64
    const uint64_t lut = (c & 16u) ? op_lut(true) : op_lut(false);
    const uint32_t code = (uint32_t)(lut >> (4u * (c & 15u))) & 15u;
    return (c >> 5) == (code == 7u ? 1u : 2u) ? code : 15u;
}
// bit i (0..3): byte i of x is not an ASCII digit
__device__ __forceinline__ uint32_t nondigit4(uint32_t x) {
    const uint32_t t = x ^ 0x30303030u;                                // digits become 0..9
    uint32_t nz = ((((t & 0x7F7F7F7Fu) + 0x76767676u) | t) & 0x80808080u) >> 7;
    return (nz | (nz >> 7) | (nz >> 14) | (nz >> 21)) & 0xFu;
}
__device__ __forceinline__ uint32_t nondigit16(const uint4 &v) {
    return nondigit4(v.x) | nondigit4(v.y) << 4 | nondigit4(v.z) << 8 | nondigit4(v.w) << 12;
}
struct TokArgs {
    const uint8_t *text; uint64_t T;                 // text bytes (the buffer is padded with '0' to whole tiles)
    const uint64_t *text_off; uint32_t n_aln, n_tiles;
    uint32_t *ops;                                   // [n_tiles * 2048] padded operations
    uint64_t *tile_agg;                              // [NT][n_tiles + 1]
    uint32_t *tile_last;                             // [n_tiles] last operation of the tile (OP_PAD: none)
    uint32_t *chunk_row;                             // [n_tiles] a row at or before the row of the tile's first operation
    uint64_t *op_slot;                               // [n_aln + 1] slot ordinal of the row's first operation
    uint32_t *row_tile, *row_i;                      // [n_aln + 1] that slot as (tile, index)
    uint64_t *row_local;                             // [2 (n_aln + 1)] reference / query advance of the tile's operations before it
    unsigned long long *tok_err, *err_op;
};

// Token error at the operation character at global byte position pos (rare path): the key is the byte position of the
// token's first digit, as the sequential tokenizer would report it.
__device__ __noinline__ void tok_error(const uint8_t *__restrict__ text, uint64_t pos, int kind, unsigned long long *tok_err) {
    uint64_t g = pos;
    while (g > 0 && is_digit(text[g - 1])) --g;
    atomicMin(tok_err, (unsigned long long)(g << 3 | (uint64_t)kind));
}
// Exact length of the token whose operation character is at pos, however many digits it has (rare path: > 9 digits behind it)
__device__ __noinline__ uint32_t tok_long_value(const uint8_t *__restrict__ text, uint64_t pos) {
    uint64_t g = pos;
    while (g > 0 && is_digit(text[g - 1])) --g;
    uint32_t val = 0;
    for (; g < pos; ++g) { const uint32_t nv = val * 10u + (uint32_t)(text[g] - '0'); val = val >= (1u << 28) ? val : nv; }
    return val;
}

// First index in [0, n) with off[idx] >= key (n when none): 64-ary search by one wave, uniform result; a handful of L2 reads.
__device__ __forceinline__ uint32_t row_lower_bound(const uint64_t *__restrict__ off, uint32_t n, uint64_t key, uint32_t lane) {
    uint32_t lo = 0, hi = n;                            // off[i] < key for i < lo, off[i] >= key for i >= hi
    while (hi > lo) {
        const uint32_t step = (hi - lo + 63) / 64;
        const uint32_t idx = lo + lane * step;
        const bool below = idx < hi && off[idx] < key;
        const uint32_t nb = (uint32_t)__popcll(__ballot(below));      // probes below the key: a prefix, off is sorted
        if (step == 1) return lo + nb;
        const uint32_t nhi = nb < 64 && lo + nb * step < hi ? lo + nb * step : hi;
        lo = nb ? lo + (nb - 1) * step + 1 : lo;
        hi = nb ? nhi : lo;
    }
    return lo;
}

__global__ __launch_bounds__(256) void tok_tiles(TokArgs A) {
    __shared__ uint32_t s_ops[TOK_CHUNK + 8];          // the tile's operations, compact (a malformed tile can hold 4096 characters)
    __shared__ uint32_t s_wave[4];
    __shared__ uint16_t s_mask[256], s_pre[256];       // per lane: non-digit bits of its 16 bytes, operations before them
    __shared__ uint64_t s_red[4 * NT];
    __shared__ uint64_t s_adv[2 * 256];                // per lane: reference / query advance of the tile's operations before its eight
    const uint32_t t = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t b0 = (uint64_t)t * TOK_CHUNK, p0 = b0 + (uint64_t)threadIdx.x * 16;
    const uint4 v = *reinterpret_cast<const uint4 *>(A.text + p0);
    uint4 pv = make_uint4(0x2a2a2a2au, 0x2a2a2a2au, 0x2a2a2a2au, 0x2a2a2a2au);      // '*': in front of the text
    if (p0) pv = *reinterpret_cast<const uint4 *>(A.text + p0 - 16);
    const uint32_t ndm = nondigit16(v), ndp = nondigit16(pv);
    // ---- operations before this lane's bytes (tile-local) ----
    uint32_t cnt = __popc(ndm), inc = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(inc, d); if ((int)lane >= d) inc += y; }
    if (lane == 63) s_wave[wave] = inc;
    s_mask[threadIdx.x] = (uint16_t)ndm;
    __syncthreads();
    uint32_t base = inc - cnt, n_t = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const uint32_t x = s_wave[w]; if (w < (int)wave) base += x; n_t += x; }
    s_pre[threadIdx.x] = (uint16_t)base;
    // ---- carry: the digits that end the 16 bytes in front (at most the last nine count; more: the long path below) ----
    const uint32_t nd0 = ndp ? (uint32_t)__clz((int)ndp) - 16u : 16u;      // trailing digit bytes of pv
    const uint32_t pw[4] = {pv.x, pv.y, pv.z, pv.w}, w4[4] = {v.x, v.y, v.z, v.w};
    uint32_t val = 0;
#pragma unroll
    for (int i = 7; i < 16; ++i) {
        const uint32_t d = ((pw[i >> 2] >> (8 * (i & 3))) & 0xFFu) - 48u;
        const uint32_t nv = (val << 3) + (val << 1) + d;
        val = (uint32_t)i >= 16u - nd0 ? (val >= (1u << 28) ? val : nv) : 0u;
    }
    bool any = nd0 != 0, lng = nd0 > 9;                 // digits seen since the last operation; more of them than the carry holds
    uint32_t k = base, bad = 0;                         // bad: bit j = the operation at byte j needs the error path
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t c = (w4[j >> 2] >> (8 * (j & 3))) & 0xFFu;
        if (ndm >> j & 1u) {
            const uint32_t code = op_code_lut(c);
            if (lng) val = tok_long_value(A.text, p0 + j);
            const bool err = !any || code == 15u || val >= (1u << 28);
            bad |= (uint32_t)err << j;
            s_ops[k] = err ? 0x7u /* a zero-length '=' keeps the stream well formed; the call fails anyway */ : (val << 4 | code);
            ++k; val = 0; any = false; lng = false;
        } else {
            const uint32_t d = c - 48u, nv = (val << 3) + (val << 1) + d;
            val = val >= (1u << 28) ? val : nv;
            any = true;
        }
    }
    if (bad) {                                          // first error of every malformed token, in the reference's order of checks
        for (uint32_t m = bad; m; m &= m - 1) {
            const int j = __ffs((int)m) - 1;
            const uint64_t pos = p0 + (uint64_t)j;
            const uint32_t c = A.text[pos];
            int kind = PAV_CIGAR_ERR_LEN_OVERFLOW;
            if (pos == 0 || !is_digit(A.text[pos - 1])) kind = PAV_CIGAR_ERR_MISSING_LEN;          // align.py:310 (checked before the op set)
            else if (op_code_lut(c) == 15u) kind = PAV_CIGAR_ERR_UNKNOWN_OP;                        // align.py:315
            tok_error(A.text, pos, kind, A.tok_err);
        }
    }
    __syncthreads();
    // ---- eight consecutive operations per lane: sums, first illegal operation; the padded tile goes out ----
    const uint32_t n_real = n_t < (uint32_t)WALK_CHUNK ? n_t : (uint32_t)WALK_CHUNK;    // more: a token without digits, the call fails
    uint32_t o[OPS_PER_LANE];
    const uint32_t i0 = threadIdx.x * OPS_PER_LANE;
#pragma unroll
    for (int j = 0; j < OPS_PER_LANE; ++j) o[j] = i0 + j < n_real ? s_ops[i0 + j] : OP_PAD;
    uint4 *out = reinterpret_cast<uint4 *>(A.ops + (uint64_t)t * WALK_CHUNK + i0);
    out[0] = make_uint4(o[0], o[1], o[2], o[3]);
    out[1] = make_uint4(o[4], o[5], o[6], o[7]);
    uint64_t run[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) run[q] = 0;
    unsigned long long ill = ~0ull;                     // first M / N / P of the lane (cigarcall.py:289-307), as a slot ordinal
#pragma unroll
    for (int j = 0; j < OPS_PER_LANE; ++j) {
        uint64_t c6[NQ];
        op_contrib(o[j], c6);
#pragma unroll
        for (int q = 0; q < NQ; ++q) run[q] += c6[q];
        const uint32_t code = o[j] & 15u;
        if ((code == 0u || code == 3u || code == 6u) && ill == ~0ull) ill = (uint64_t)t * WALK_CHUNK + i0 + j;
    }
    if (__ballot(ill != ~0ull)) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const unsigned long long y = __shfl_xor(ill, d); ill = y < ill ? y : ill; }
        if (lane == 0 && ill != ~0ull) atomicMin(A.err_op, ill);
    }
    // rows that start in this tile (text_off in [b0, b0 + 4096)): [rlo, rhi); most tiles have none
    const uint32_t rlo = row_lower_bound(A.text_off, A.n_aln + 1, b0, lane);
    const uint32_t rhi = row_lower_bound(A.text_off, A.n_aln + 1, b0 + TOK_CHUNK, lane);
    if (rhi > rlo) {                                    // (uniform)
        // exclusive block scan of the advances: a row's base is the tile prefix + the advance of the tile's operations before it
        uint64_t ex[2] = {run[0], run[1]}, tt[2];
        block_excl_scan<2>(ex, tt, s_red);
        s_adv[2 * threadIdx.x] = ex[0]; s_adv[2 * threadIdx.x + 1] = ex[1];
        __syncthreads();
        for (uint32_t r = rlo + threadIdx.x; r < rhi; r += 256) {
            const uint32_t off = (uint32_t)(A.text_off[r] - b0);
            const uint32_t L = off >> 4, jb = off & 15u;
            uint32_t i = (uint32_t)s_pre[L] + __popc((uint32_t)s_mask[L] & ((1u << jb) - 1u));     // operations of the tile before the row
            if (i > n_real) i = n_real;
            uint32_t rt = t, ri = i;
            uint64_t a = 0, b = 0;
            if (i >= n_real) { rt = t + 1; ri = 0; }    // behind the tile's last operation: the first slot of the next tile
            else {                                      // the owner lane's base + the advance of operations [8 (i / 8), i)
                const uint32_t own = i >> 3;
                a = s_adv[2 * own]; b = s_adv[2 * own + 1];
                for (uint32_t x = own * 8; x < i; ++x) {
                    uint64_t c6[NQ];
                    op_contrib(s_ops[x], c6);
                    a += c6[0]; b += c6[1];
                }
            }
            A.row_tile[r] = rt; A.row_i[r] = ri;
            A.op_slot[r] = (uint64_t)rt * WALK_CHUNK + ri;
            A.row_local[2ull * r] = a; A.row_local[2ull * r + 1] = b;
            // a row whose text ends inside a length: IndexError in the reference (align.py:307)
            if (r > 0) {
                const uint64_t e = A.text_off[r], pb = A.text_off[r - 1];
                if (e > pb && is_digit(A.text[e - 1])) {
                    uint64_t q = e - 1;
                    while (q > pb && is_digit(A.text[q - 1])) --q;
                    atomicMin(A.tok_err, (unsigned long long)(q << 3 | (uint64_t)PAV_CIGAR_ERR_TRUNCATED));
                }
            }
        }
    }
    // ---- tile sums ----
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) run[q] += __shfl_xor(run[q], d);
    }
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int q = 0; q < NQ; ++q) s_red[wave * NT + q] = run[q];
    __syncthreads();
    if (threadIdx.x < NQ)
        A.tile_agg[(uint64_t)threadIdx.x * (A.n_tiles + 1) + t] =
            s_red[threadIdx.x] + s_red[NT + threadIdx.x] + s_red[2 * NT + threadIdx.x] + s_red[3 * NT + threadIdx.x];
    else if (threadIdx.x == NQ) {
        A.tile_agg[(uint64_t)NQ * (A.n_tiles + 1) + t] = n_real;
        A.tile_last[t] = n_real ? s_ops[n_real - 1] : OP_PAD;
        const uint32_t before = rlo ? rlo - 1 : 0;        // its operations may end in front of this tile: the walk moves on from there
        A.chunk_row[t] = A.n_aln && before >= A.n_aln ? A.n_aln - 1 : before;
    }
}

// One workgroup per tile quantity: exclusive prefix over the tiles, in place (tile_agg[q][n_tiles] = the total, which also goes
// straight into the host's pinned status block: no copy, no second readback).  Then the per-row values that need a prefix:
// workgroup NQ: the real operation ordinal of every row's first operation (op_off); 0 / 1: the rows' reference / query base
// (rowbase) and the first row whose advance does not fit the record it names (the reference raises IndexError at the first X base
// past the end, cigarcall.py:104-105; here such a row is refused before any kernel reads past a record); 2: the error words.
struct ScanArgs {
    uint64_t *tile_agg; uint32_t n_tiles, n_aln;
    const uint32_t *row_tile, *row_i; const uint64_t *row_local;
    uint64_t *op_off, *rowbase;
    const pav_aln *aln; SeqView ref, tig;
    const unsigned long long *tok_err, *err_op;
    volatile uint64_t *host_status;                  // [0, NT) totals, [NT] token error key, [NT + 1] illegal operation slot,
};                                                   // [NT + 2] / [NT + 3] first row that overruns its reference / query record

__global__ __launch_bounds__(256) void tile_scan(ScanArgs A) {
    constexpr int PER = 16;
    __shared__ uint64_t lds[4];
    __shared__ uint64_t tile[256 * (PER + 1)];
    const uint32_t q = blockIdx.x, n = A.n_tiles;
    uint64_t *x = A.tile_agg + (uint64_t)q * (n + 1);
    uint64_t carry = 0;
    for (uint32_t base = 0; base < n; base += 256 * PER) {
        uint64_t c[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) { const uint32_t i = base + k * 256 + threadIdx.x; c[k] = i < n ? x[i] : 0ull; }
#pragma unroll
        for (int k = 0; k < PER; ++k) { const uint32_t j = k * 256 + threadIdx.x; tile[j + j / PER] = c[k]; }
        __syncthreads();
        uint64_t v[1] = {0}, tot[1];
#pragma unroll
        for (int k = 0; k < PER; ++k) { c[k] = tile[threadIdx.x * (PER + 1) + k]; v[0] += c[k]; }
        block_excl_scan<1>(v, tot, lds);
        uint64_t run = carry + v[0];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const uint32_t i = base + threadIdx.x * PER + k;
            if (i < n) x[i] = run;
            run += c[k];
        }
        carry += tot[0];
    }
    if (threadIdx.x == 0) { x[n] = carry; A.host_status[q] = carry; }
    if (q == 2 && threadIdx.x == 0) { A.host_status[NT] = *A.tok_err; A.host_status[NT + 1] = *A.err_op; }
    if (q != (uint32_t)NQ && q > 1) return;
    // the prefixes this workgroup has just written are read back past its L1 (agent-scope loads are served by the L2)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    auto pre = [&](uint32_t i) { return __hip_atomic_load(x + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    if (q == (uint32_t)NQ) {
        for (uint32_t r = threadIdx.x; r <= A.n_aln; r += 256) A.op_off[r] = pre(A.row_tile[r]) + A.row_i[r];
        return;
    }
    __shared__ unsigned long long red[4];
    unsigned long long bad = ~0ull;
    for (uint32_t r = threadIdx.x; r <= A.n_aln; r += 256) {
        const uint64_t here = pre(A.row_tile[r]) + A.row_local[2ull * r + q];
        A.rowbase[2ull * r + q] = here;
        if (r < A.n_aln) {
            const uint64_t adv = pre(A.row_tile[r + 1]) + A.row_local[2ull * (r + 1) + q] - here;
            const pav_aln a = A.aln[r];
            const bool over = q == 0 ? (uint64_t)a.pos + adv > A.ref.len[a.ref_id] : adv > A.tig.len[a.tig_id];
            if (over && r < bad) bad = r;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const unsigned long long y = __shfl_xor(bad, d); bad = y < bad ? y : bad; }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = bad;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) bad = red[w] < bad ? red[w] : bad;
        A.host_status[NT + 2 + q] = bad;
    }
}

// Contiguous operation array (pav_cigar_fetch_ops, the error path): tile t's operations go to their real ordinals.
__global__ __launch_bounds__(256) void ops_compact(const uint32_t *__restrict__ ops_pad, const uint64_t *__restrict__ pre_n,
                                                   uint32_t *__restrict__ out) {
    const uint32_t t = blockIdx.x;
    const uint64_t a = pre_n[t], n = pre_n[t + 1] - a;
    for (uint32_t i = threadIdx.x; i < n; i += 256) out[a + i] = ops_pad[(uint64_t)t * WALK_CHUNK + i];
}

// ---- the walk --------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_ops(const uint32_t *__restrict__ ops, uint64_t n_ops, uint64_t first,
                                         uint32_t (&o)[OPS_PER_LANE]) {
    if (first + OPS_PER_LANE <= n_ops) {
        const uint4 a = *reinterpret_cast<const uint4 *>(ops + first);
        const uint4 b = *reinterpret_cast<const uint4 *>(ops + first + 4);
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    } else {
#pragma unroll
        for (int j = 0; j < OPS_PER_LANE; ++j) o[j] = first + j < n_ops ? ops[first + j] : 0x5u /* 0H */;
    }
}

struct WalkArgs {
    const uint32_t *ops;                 // padded: tile t owns slots [t * 2048, (t + 1) * 2048)
    const uint64_t *op_slot;             // [n_aln + 1] slot ordinal of every row's first operation
    const uint64_t *op_off;              // ... and its real ordinal
    const pav_aln *aln; uint32_t n_aln, n_tiles;
    const uint64_t *tile_pre;            // [NT][n_tiles + 1] exclusive prefix of the tile sums
    const uint64_t *rowbase; const uint32_t *chunk_row, *tile_last;
    SeqView ref, tig;
    pav_snv *snv; pav_indel *indel;
};

// Emit SNV rows and INDEL stubs.  One workgroup per tile of operation slots; each lane owns 8 consecutive slots; the block scan
// on top of the tile prefix gives every operation its running positions and its output slots, so the output order is exactly the
// reference's (row, op, base) order.  INDEL stubs (one per I / D op) are written where they are met.  SNV rows are written
// flat: every op leaves its first output slot and positions in LDS, then lane t of the workgroup builds rows t, t + 256, ... of
// the tile (binary search for the owning 'X' run) - complete rows with REF / ALT read from the ASCII planes
// (cigarcall.py:104-105), stored 16 B per lane to consecutive addresses.
// Two instances of the same walk: WALK_INDEL writes the stubs - little traffic, on the critical path of the homology scans;
// WALK_SNV writes the SNV rows - 13 M isolated sector fetches per haplotype - on the side stream beside the homology scans.
constexpr int WALK_INDEL = 1, WALK_SNV = 2;
// (the stub walk keeps 152 registers - three waves a SIMD; asked to fit four it spills 100 bytes a lane and takes 0.080 instead of
//  0.054 ms: measured in round 6, left alone)
template <int MODE>
__global__ __launch_bounds__(256) void walk_emit(WalkArgs A) {
    constexpr int NSLOT = MODE == WALK_SNV ? WALK_CHUNK : 1;
    // One block of LDS: s_pre | d_pos | d_q0 | d_row, 32 KiB for the SNV walk - five workgroups a CU (160 KiB; the registers allow five).
    // With 208 bytes more (an unused sentinel entry and a scan scratch of its own) it was four.  The scan's scratch lies over the
    // front of s_pre: block_excl_scan ends with a barrier and the arrays are written after it.
    constexpr int SMEM_WORDS = 4 * NSLOT > 8 * NQ ? 4 * NSLOT : 8 * NQ;
    __shared__ __attribute__((aligned(16))) uint32_t smem[SMEM_WORDS];
#ifdef PAV_LDS_AB                     // tuning build: 256 bytes more, the workgroup count per CU of round 5
    __shared__ uint32_t ab_pad[64];
    if (MODE == WALK_SNV && threadIdx.x == 0) reinterpret_cast<volatile uint32_t *>(ab_pad)[0] = 1;
#endif
    uint64_t *lds = reinterpret_cast<uint64_t *>(smem);
    uint32_t *s_pre = smem;                             // first SNV row of the op, relative to the tile's first row
    uint32_t *d_pos = smem + NSLOT, *d_q0 = smem + 2 * NSLOT, *d_row = smem + 3 * NSLOT;   // 'X' ops: POS, stored contig position of base 0, row | rev << 31
    const uint32_t t = blockIdx.x;
    const uint64_t first = (uint64_t)t * WALK_CHUNK + (uint64_t)threadIdx.x * OPS_PER_LANE;      // slot ordinal
    uint32_t o[OPS_PER_LANE];
    {
        const uint4 a = *reinterpret_cast<const uint4 *>(A.ops + first), b = *reinterpret_cast<const uint4 *>(A.ops + first + 4);
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    }
    const uint64_t stride = (uint64_t)A.n_tiles + 1;
    uint64_t run[NQ] = {0, 0, 0, 0, 0, 0}, tot[NQ];
#pragma unroll
    for (int j = 0; j < OPS_PER_LANE; ++j) {
        uint64_t c[NQ];
        op_contrib(o[j], c);
#pragma unroll
        for (int q = 0; q < NQ; ++q) run[q] += c[q];
    }
    block_excl_scan<NQ>(run, tot, lds);
    const uint64_t snv_base = A.tile_pre[2 * stride + t];
    const uint32_t n_rows = (uint32_t)tot[2];            // SNV rows of this tile
    if (MODE == WALK_SNV && n_rows == 0) return;         // (uniform)
    if (MODE == WALK_INDEL && tot[3] == 0) return;
#pragma unroll
    for (int q = 0; q < NQ; ++q) if (q != 2) run[q] += A.tile_pre[(uint64_t)q * stride + t];
    const uint64_t real0 = A.tile_pre[(uint64_t)NQ * stride + t] + (uint64_t)threadIdx.x * OPS_PER_LANE;   // real ordinal of this lane's first slot
                                                                                                             // (only meaningful for operations: they lead the tile)
    // row of this lane's first op: a row at or before the tile's first op comes from tok_tiles (chunk_row), lanes walk on
    const uint32_t row0 = A.chunk_row[t];
    const pav_aln al0 = A.aln[row0];
    const uint64_t roff0 = A.ref.off[al0.ref_id], toff0 = A.tig.off[al0.tig_id];
    const uint32_t slot0 = threadIdx.x * OPS_PER_LANE;
    {
        uint32_t row = row0;
        uint64_t row_end = A.op_slot[row + 1];
        pav_aln al = al0;
        if (row_end <= first && row + 1 < A.n_aln) {           // (slots behind the last row's operations stay with the last row: padding)
            do { ++row; row_end = A.op_slot[row + 1]; } while (row_end <= first && row + 1 < A.n_aln);     // rows without ops are skipped
            al = A.aln[row];
        }
        uint64_t rb_ref = A.rowbase[2ull * row], rb_tig = A.rowbase[2ull * row + 1];
        uint64_t tlen = A.tig.len[al.tig_id];
        uint64_t row_begin = A.op_off[row];                    // real ordinal; kept in a register: no load inside the per-op branches below
        // the stubs carry what the homology scans need to know about the row's records (offsets, lengths): one load there, not four
        uint64_t s_roff = 0, s_toff = 0, s_rlen = 0;
        if constexpr (MODE == WALK_INDEL) { s_roff = A.ref.off[al.ref_id]; s_toff = A.tig.off[al.tig_id]; s_rlen = A.ref.len[al.ref_id]; }
        // last_op / last_oplen carried across lanes: the operation in the slot in front, or the last one of the tile in front
        uint32_t prev = threadIdx.x ? A.ops[first - 1] : (t ? A.tile_last[t - 1] : OP_PAD);
#pragma unroll
        for (int j = 0; j < OPS_PER_LANE; ++j) {
            const uint64_t k = first + j;
            if constexpr (MODE == WALK_SNV) s_pre[slot0 + j] = (uint32_t)run[2];
            while (k >= row_end && row + 1 < A.n_aln) {        // next row (rows without ops are skipped)
                ++row; row_end = A.op_slot[row + 1];
                al = A.aln[row];
                rb_ref = A.rowbase[2ull * row]; rb_tig = A.rowbase[2ull * row + 1];
                tlen = A.tig.len[al.tig_id];
                row_begin = A.op_off[row];
                if constexpr (MODE == WALK_INDEL) { s_roff = A.ref.off[al.ref_id]; s_toff = A.tig.off[al.tig_id]; s_rlen = A.ref.len[al.ref_id]; }
            }
            const uint32_t code = o[j] & 15u, len = o[j] >> 4;
            const int64_t pos_ref = (int64_t)al.pos + (int64_t)(run[0] - rb_ref);
            const int64_t pos_tig = (int64_t)(run[1] - rb_tig);
            const int rev = al.rev != 0;
            if (code == 8) {                                                   // 'X'  cigarcall.py:95-139
                if constexpr (MODE == WALK_SNV) {
                d_pos[slot0 + j] = (uint32_t)pos_ref;
                d_q0[slot0 + j] = (uint32_t)(rev ? (int64_t)tlen - pos_tig - 1 : pos_tig);   // cigarcall.py:108-109; base i: -i / +i
                d_row[slot0 + j] = row | (rev ? 0x80000000u : 0u);
                }
            } else if (code == 1 || code == 2) {                               // 'I' / 'D' stub
                if constexpr (MODE == WALK_INDEL) {
                pav_indel r;
                r.aln = row;
                r.op_index = (uint32_t)(real0 + j - row_begin) + 1;            // cigar_index, cigarcall.py:89
                r.pos = (uint32_t)pos_ref;                                     // un-shifted; finalised by homology_kernel
                r.end = (uint32_t)s_rlen;                                      // stub only: length of the reference record
                r.svlen = len;
                r.qry_pos = (uint32_t)pos_tig;                                 // oriented, un-shifted
                r.qry_end = (uint32_t)tlen;                                    // stub only: length of the contig
                // last_op / last_oplen (cigarcall.py:149-151,310-311): previous op of the same row
                const bool has_prev = real0 + j > row_begin;                   // first op of a row: last_op is None
                r.left_shift = (has_prev && (prev & 15u) == 7u) ? (prev >> 4) : 0u;   // shift cap; 0 when last_op != '='
                r.hom_ref_l = (uint32_t)s_roff; r.hom_ref_r = (uint32_t)(s_roff >> 32);       // stub only: arena offsets of the two records
                r.hom_tig_l = (uint32_t)s_toff; r.hom_tig_r = (uint32_t)(s_toff >> 32);
                r.seq_off = run[4];
                r.svtype = code == 1 ? 0 : 1;
#pragma unroll
                for (int b = 0; b < 7; ++b) r.pad[b] = 0;
                r.pad[0] = (uint8_t)rev;                                       // stub only: strand of the row
                A.indel[run[3]] = r;
                }
            }                                                                  // M, N, P: reported by tok_tiles before any row is emitted
            uint64_t c[NQ];
            op_contrib(o[j], c);
#pragma unroll
            for (int q = 0; q < NQ; ++q) run[q] += c[q];
            prev = o[j];
        }
    }
    if constexpr (MODE == WALK_INDEL) return;
    __syncthreads();

    // flat SNV rows: two rows per lane in flight; lanes past the last row repeat it and do not store (no branch around the loads)
    // (round 6, measured: four or eight rows in flight per lane change nothing - 0.342 / 0.354 against 0.349 ms beside the scans: the
    //  memory system is full of this kernel's fetches already, about a million outstanding; fewer resident workgroups make it slower)
    constexpr int EMIT_U = 2;
    uint4 *out = reinterpret_cast<uint4 *>(A.snv + snv_base);              // pav_snv = {aln, pos, qry_pos, ref | alt << 8 | pad << 16}
    static_assert(sizeof(pav_snv) == sizeof(uint4), "pav_snv is stored as one 16-byte vector");
    for (uint32_t s0 = 0; s0 < n_rows; s0 += 256 * EMIT_U) {
        uint32_t at[EMIT_U], rw[EMIT_U], pos[EMIT_U], qp[EMIT_U];
        uint64_t roff[EMIT_U], toff[EMIT_U];
        uint8_t br[EMIT_U], bt[EMIT_U];
        bool live[EMIT_U];
#pragma unroll
        for (int u = 0; u < EMIT_U; ++u) {
            const uint32_t want = s0 + u * 256 + threadIdx.x;
            live[u] = want < n_rows;
            const uint32_t s = live[u] ? want : n_rows - 1;
            at[u] = s;
            uint32_t lo = 0, hi = WALK_CHUNK;               // largest op with s_pre <= s: the 'X' run that owns row s
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (s_pre[mid] <= s) lo = mid; else hi = mid; }
            const uint32_t i = s - s_pre[lo];
            rw[u] = d_row[lo];
            pos[u] = d_pos[lo] + i;
            qp[u] = (rw[u] >> 31) ? d_q0[lo] - i : d_q0[lo] + i;
            roff[u] = roff0; toff[u] = toff0;
            if ((rw[u] & 0x7FFFFFFFu) != row0) {            // tile that crosses into another row
                const pav_aln al = A.aln[rw[u] & 0x7FFFFFFFu];
                roff[u] = A.ref.off[al.ref_id]; toff[u] = A.tig.off[al.tig_id];
            }
        }
#pragma unroll
        for (int u = 0; u < EMIT_U; ++u) {
            // non-temporal: a line is touched once (tools/ubench/gather_rate.hip: isolated lines come 14 % faster that way)
            br[u] = __builtin_nontemporal_load(A.ref.ascii + roff[u] + pos[u]);
            bt[u] = __builtin_nontemporal_load(A.tig.ascii + toff[u] + qp[u]);     // stored contig; complemented when the row is reversed
        }
#pragma unroll
        for (int u = 0; u < EMIT_U; ++u) {
            const uint32_t alt = (rw[u] >> 31) ? comp_ascii(bt[u]) : bt[u];
            if (live[u]) out[at[u]] = make_uint4(rw[u] & 0x7FFFFFFFu, pos[u], qp[u], (uint32_t)br[u] | alt << 8);
        }
    }
}

// ---- verify mode ----------------------------------------------------------------------------------------------------------
// Streams both packed sequences along every '=' and 'X' operation and checks what the aligner claimed: every base of an '='
// run equal, every base of an 'X' run different (SURVEY.md section 8(d) "verify mode"; the reference trusts the CIGAR -
// pavlib/cigarcall.py:91-93 skips '=' runs without looking at them).  The streaming member of the call path: 0.25 B of 2-bit
// plane per reference base + 0.25 B per contig base (non-ACGT planes only where SeqView::dirty marks a block).  Operation positions come from the same block scan as walk_emit; the
// workgroup then cuts its runs into 64-base pieces, one lane per piece (one unaligned 64-base window of each 2-bit plane,
// reversed and complemented for reverse-strand rows).
struct VerifyArgs {
    const uint32_t *ops;                 // padded tiles of operation slots, as WalkArgs
    const uint64_t *op_slot; const pav_aln *aln; uint32_t n_aln, n_tiles;
    const uint64_t *tile_pre; const uint64_t *rowbase; const uint32_t *chunk_row;
    SeqView ref, tig;
    unsigned long long *cnt;             // [0] '=' bases, [1] of them different, [2] 'X' bases, [3] of them equal, [4] first bad op
};

typedef unsigned __int128 u128;
constexpr uint32_t VPIECE = 64;                                        // bases per lane and step
constexpr int VSPLIT = 1;                                              // workgroups per tile of 2048 operation slots (round 3: the padded tiles of
                                                                       // tok_tiles are about half full - a tile is the work half a chunk was; 2: 0.41 ms)

// Five dwords from the dword that holds base a (4-byte aligned 16-byte load + 1 dword): the 64-base window is bits [2 (a & 15), + 128).
struct W5 { uint32_t w[5]; };
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ W5 window_dw(const uint32_t *__restrict__ two, uint64_t a) {
    const uint32_t *p = two + (a >> 4);
    const u32x4_a4 v = *reinterpret_cast<const u32x4_a4 *>(p);
    return W5{{v.x, v.y, v.z, v.w, p[4]}};
}
__device__ __forceinline__ void align_dw(const W5 &x, uint64_t a, uint32_t (&out)[4]) {
    const uint32_t sh = ((uint32_t)a & 15u) * 2u;
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = __builtin_amdgcn_alignbit(x.w[i + 1], x.w[i], sh);
}
__device__ __forceinline__ uint64_t window64(const uint32_t *__restrict__ mask, uint64_t a) {         // 64 mask bits from position a
    const uint64_t *m64 = reinterpret_cast<const uint64_t *>(mask);
    const uint64_t w = a >> 6;
    const int b = (int)(a & 63);
    const uint64_t v0 = m64[w], v1 = m64[w + 1];
    return b ? (v0 >> b | v1 << (64 - b)) : v0;
}
__device__ __forceinline__ uint64_t low_bits(uint32_t n) { return n >= 64 ? ~0ull : ((1ull << n) - 1ull); }
__device__ __forceinline__ bool dirty_span(const uint8_t *__restrict__ dirty, uint64_t a) {              // any non-ACGT in [a, a + 64)?
    const uint64_t d0 = a >> DIRTY_SHIFT, d1 = (a + 63) >> DIRTY_SHIFT;
    uint32_t d = dirty[d0];
    if (d1 != d0) d |= dirty[d1];
    return d != 0;
}
__device__ __forceinline__ uint64_t spread32(uint32_t x32) {                                          // bit i -> bit 2i
    uint64_t x = x32;
    x = (x | x << 16) & 0x0000FFFF0000FFFFull; x = (x | x << 8) & 0x00FF00FF00FF00FFull; x = (x | x << 4) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | x << 2) & 0x3333333333333333ull; x = (x | x << 1) & 0x5555555555555555ull;
    return x;
}
__device__ __forceinline__ u128 spread64(uint64_t m) { return (u128)spread32((uint32_t)(m >> 32)) << 64 | spread32((uint32_t)m); }
__device__ __forceinline__ int popc128(u128 x) { return __popcll((uint64_t)x) + __popcll((uint64_t)(x >> 64)); }

// Positions are arena offsets in bases.  Arenas below 2^32 bases (a human genome is 3.1 G) run the kernel on 32-bit
// positions: half the descriptor bytes in LDS, and every address is a scalar base + a 32-bit lane offset (one VALU op
// instead of 64-bit shifts and adds).  WIDE keeps the high words beside them.
// Where the time goes (bench haplotype, 0.30 ms, SQ counters in DESIGN.md section 5): 111 M vector instructions at four cycles
// each = 0.18 ms of issue on 1024 SIMDs, 1.63 GB of HBM = 0.25 ms at the streaming rate of pack_kernel; four waves per SIMD
// overlap the two to 0.30 ms.  With the loads of the loop compiled out the kernel takes 0.255 ms, with the loop compiled out
// 0.07 ms (the scan + descriptors).
// Tried and dropped: 5 waves / SIMD via amdgpu_waves_per_eu (96 VGPRs: +5 %), one unaligned 2-byte load of the dirty pair
// (+5 %), non-temporal window loads (no change), one or four pieces per lane and step, one workgroup per chunk (no change).
template <bool WIDE> struct VPos { typedef uint32_t type; };
template <> struct VPos<true> { typedef uint64_t type; };

template <typename P>
__device__ __forceinline__ W5 window_at(const uint32_t *__restrict__ two, P a) {
    // byte offset of the dword that holds base a; for 32-bit positions it stays a 32-bit lane offset on a scalar base
    const P byte = (a >> 4) << 2;
    const uint32_t *p = reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(two) + byte);
    const u32x4_a4 v = *reinterpret_cast<const u32x4_a4 *>(p);
    return W5{{v.x, v.y, v.z, v.w, p[4]}};
}

template <bool WIDE>
__global__ __launch_bounds__(256) void verify_kernel(VerifyArgs A) {
    typedef typename VPos<WIDE>::type pos_t;
    __shared__ uint64_t lds[4 * 3];
    // VSPLIT workgroups share a 2048-operation chunk: all repeat its (cheap) scan, each keeps the descriptors of one part - a
    // fraction of the LDS per workgroup, more waves per CU for the gather below.
    constexpr int VSLOTS = WALK_CHUNK / VSPLIT;
    // One descriptor per '=' / 'X' run of this part, in operation order, runs without bases left out ("compact slots"):
    // ref position, contig position, len << 2 | rev << 1 | is 'X', first piece; slot n_slots is a sentinel of length 0.
    __shared__ uint32_t d_ref[VSLOTS + 1], d_tig[VSLOTS + 1], d_len[VSLOTS + 1], s_pre[VSLOTS + 1];
    __shared__ uint2 d_hi[WIDE ? VSLOTS + 1 : 1];                        // high words of the two positions
    __shared__ uint16_t d_op[VSLOTS + 1];                                // slot -> operation of the part (first_bad_op)
    constexpr int VU = 2;                                                // pieces per lane and step
    __shared__ uint8_t mark[4][VU * VPIECE];                              // per wave: which of the 128 pieces of a step start a run
    __shared__ uint4 km_tbl[VPIECE + 1];                                 // n bases -> compare mask of the four dwords
    constexpr uint32_t EVEN = 0x55555555u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t chunk = blockIdx.x / VSPLIT, slot0 = (blockIdx.x % VSPLIT) * VSLOTS;
    const uint64_t first = (uint64_t)chunk * WALK_CHUNK + (uint64_t)threadIdx.x * OPS_PER_LANE;      // slot ordinal
    uint32_t o[OPS_PER_LANE];
    {
        const uint4 a = *reinterpret_cast<const uint4 *>(A.ops + first), b = *reinterpret_cast<const uint4 *>(A.ops + first + 4);
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    }
    const bool mine_half = threadIdx.x * OPS_PER_LANE >= slot0 && threadIdx.x * OPS_PER_LANE < slot0 + VSLOTS;
    const uint32_t my0 = threadIdx.x * OPS_PER_LANE - slot0;             // this lane's first operation within the part (its 8 ops lie in one part)
    // one block scan: the two positions, and (runs | pieces) of this part packed in one word.  The runs are cut into 64-base
    // pieces (one lane, one window of each plane) - an 'X' of one base costs one lane, not one wave.
    constexpr int PK_SHIFT = 44;                                         // pieces below, runs above (<= 1024 runs, < 2^38 / 64 pieces)
    uint64_t run[3] = {0, 0, 0}, tot[3];
    uint32_t np[OPS_PER_LANE], len_eq = 0, len_x = 0;
#pragma unroll
    for (int j = 0; j < OPS_PER_LANE; ++j) {
        uint64_t c[NQ];
        op_contrib(o[j], c);
        run[0] += c[0]; run[1] += c[1];
        const uint32_t code = o[j] & 15u, len = o[j] >> 4;
        np[j] = (mine_half && (code == 7 || code == 8)) ? (len + VPIECE - 1) / VPIECE : 0u;
        if (np[j]) run[2] += (1ull << PK_SHIFT) + np[j];
        if (np[j]) { if (code == 8) len_x += len; else len_eq += len; }    // the bases checked are counted here, not per piece
    }
    block_excl_scan<3>(run, tot, lds);
    run[0] += A.tile_pre[chunk];
    run[1] += A.tile_pre[(uint64_t)A.n_tiles + 1 + chunk];
    const uint32_t n_slots = (uint32_t)(tot[2] >> PK_SHIFT), n_pieces = (uint32_t)(tot[2] & ((1ull << PK_SHIFT) - 1));
    if (threadIdx.x <= VPIECE) {
        uint32_t m[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int have = (int)threadIdx.x - 16 * i;                  // bases of the piece in dword i
            m[i] = have >= 16 ? EVEN : (have <= 0 ? 0u : EVEN & ((1u << (2 * have)) - 1u));
        }
        km_tbl[threadIdx.x] = make_uint4(m[0], m[1], m[2], m[3]);
    }
#pragma unroll
    for (int u = 0; u < VU; ++u) mark[wave][u * VPIECE + lane] = 0;
    if (threadIdx.x == 0) {
        d_ref[n_slots] = 0; d_tig[n_slots] = 0; d_len[n_slots] = 0; s_pre[n_slots] = n_pieces;
        if (WIDE) d_hi[n_slots] = make_uint2(0u, 0u);
        d_op[n_slots] = 0;
    }
    if (mine_half && n_slots) {                                          // (n_slots: uniform; a tile of padding has no runs)
        uint32_t row = A.chunk_row[chunk];
        uint64_t row_end = A.op_slot[row + 1];
        while (row_end <= first && row + 1 < A.n_aln) { ++row; row_end = A.op_slot[row + 1]; }
        // per row: ra = r0 + (reference bases before the op), ta = t0 +/- (contig bases before the op).  Forward rows: first
        // base of the run; reverse rows: the stored base that is oriented base 0 of the run (the run goes downwards)
        uint64_t r0, t0;
        bool rev;
        auto enter_row = [&](uint32_t rw) {
            const pav_aln al = A.aln[rw];
            const uint64_t rb_ref = A.rowbase[2ull * rw], rb_tig = A.rowbase[2ull * rw + 1];
            rev = al.rev != 0;
            r0 = A.ref.off[al.ref_id] + (uint64_t)al.pos - rb_ref;
            t0 = rev ? A.tig.off[al.tig_id] + A.tig.len[al.tig_id] - 1 + rb_tig : A.tig.off[al.tig_id] - rb_tig;
        };
        enter_row(row);
        uint32_t slot = (uint32_t)(run[2] >> PK_SHIFT), at = (uint32_t)(run[2] & ((1ull << PK_SHIFT) - 1));
#pragma unroll
        for (int j = 0; j < OPS_PER_LANE; ++j) {
            const uint64_t k = first + j;
            if (k >= row_end && row + 1 < A.n_aln) {
                do { ++row; row_end = A.op_slot[row + 1]; } while (k >= row_end && row + 1 < A.n_aln);
                enter_row(row);
            }
            const uint32_t code = o[j] & 15u, len = o[j] >> 4;
            if (np[j]) {
                const uint64_t ra = r0 + run[0], ta = rev ? t0 - run[1] : t0 + run[1];
                d_ref[slot] = (uint32_t)ra; d_tig[slot] = (uint32_t)ta;
                d_len[slot] = len << 2 | (rev ? 2u : 0u) | (code == 8 ? 1u : 0u);
                if (WIDE) d_hi[slot] = make_uint2((uint32_t)(ra >> 32), (uint32_t)(ta >> 32));
                s_pre[slot] = at;
                d_op[slot] = (uint16_t)(my0 + j);
                ++slot; at += np[j];
            }
            uint64_t c[NQ];
            op_contrib(o[j], c);
            run[0] += c[0]; run[1] += c[1];
        }
    }
    __syncthreads();
    // Every wave takes one contiguous quarter of the pieces, 128 per step (two per lane, the loads of both issued before the
    // first compare); consecutive lanes read consecutive windows of the same run.  The run that owns a piece is not searched
    // for: c0 = run of the step's first piece; the lanes look at the first pieces of the 128 runs after c0 (more cannot
    // start within 128 pieces), mark the pieces of this step that start a run, and a lane's run = c0 + the marks at or
    // before its piece (ballot + mbcnt).
    uint32_t bad_tot = 0, bad_x = 0, bad_slot = ~0u;
    const uint32_t per_wave = ((n_pieces + 4 * VU * VPIECE - 1) / (4 * VU * VPIECE)) * VU * VPIECE;
    // (wave id, piece range, c0 in scalar registers: the loop control and the owner arithmetic stay off the vector ALU)
    const uint32_t swave = __builtin_amdgcn_readfirstlane((uint32_t)wave);
    const uint32_t p0 = swave * per_wave, p1 = min(p0 + per_wave, n_pieces);
    uint32_t c0 = 0;
    if (p0 < p1) {                                                       // last run that starts at or before piece p0
        uint32_t hi = n_slots;
        while (hi - c0 > 1) {
            const uint32_t mid = (c0 + hi) >> 1;
            if ((uint32_t)__builtin_amdgcn_readfirstlane(s_pre[mid]) <= p0) c0 = mid; else hi = mid;
        }
    }
    // The owners of a step's pieces (LDS only: run starts, marks, ballot) are worked out one step ahead, after the loads of the
    // current step have been issued: four dependent LDS round trips per step leave the chain that the four waves of a SIMD have
    // to hide.  No branch around it - past the last step it reads clamped slots and marks nothing.
    auto owners = [&](const uint32_t base, uint32_t (&q)[VU]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < VU; ++u) {                                   // run c0 + 1 + i starts at piece base + 1 + s
            const uint32_t s = s_pre[min(c0 + 1 + (uint32_t)u * VPIECE + (uint32_t)lane, n_slots)] - base - 1;
            if (s < VU * VPIECE) mark[wave][s] = 1;
        }
        // mark[i]: a run starts at piece base + 1 + i; piece base + l belongs to run c0 + (marks below l)
#pragma unroll
        for (int u = 0; u < VU; ++u) {
            const uint32_t m = mark[wave][u * VPIECE + lane];
            mark[wave][u * VPIECE + lane] = 0;
            const uint64_t M = __ballot(m != 0);
            q[u] = __builtin_amdgcn_mbcnt_hi((uint32_t)(M >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)M, c0));
            c0 += (uint32_t)__popcll(M);
        }
    };
    uint32_t q[VU], q_next[VU];
    if (p0 < p1) owners(p0, q);
    for (uint32_t base = p0; base < p1; base += VU * VPIECE) {
        uint32_t n[VU], flags[VU], lsh[VU];
        uint8_t dr[VU], dt[VU];                                             // summary bytes of the blocks a window touches (used after all loads are out)
        pos_t pr[VU], pt[VU];
        W5 xr[VU], xt[VU];
#pragma unroll
        for (int u = 0; u < VU; ++u) {
            // lanes past the last piece sit on the sentinel (length 0, position 0): no branch around the loads, so nothing of
            // piece u has to be waited for before the loads of piece u + 1 are issued
            const uint32_t piece = base + (uint32_t)u * VPIECE + (uint32_t)lane;
            const bool live = piece < n_pieces;
            pos_t ra = d_ref[q[u]], ta = d_tig[q[u]];
            if (WIDE) { const uint2 H = d_hi[q[u]]; ra |= (pos_t)((uint64_t)H.x << 32); ta |= (pos_t)((uint64_t)H.y << 32); }
            const uint32_t dl = d_len[q[u]], len = dl >> 2;
            flags[u] = dl & 3u;                                          // bit 1 reverse row, bit 0 'X'
            const uint32_t first_piece = s_pre[q[u]];
            const uint32_t b = live ? (piece - first_piece) * VPIECE : 0u;
            n[u] = live ? min(VPIECE, len - b) : 0u;
            pr[u] = ra + b;
            // forward rows: the window starts at the piece's first base.  Reverse rows: it ends there (stored bases ta-b-63 .. ta-b);
            // at the very start of the arena it starts at 0 and the reversed window is moved down by the missing bases (lsh)
            pos_t s0 = ta + b;
            lsh[u] = 0;
            if (flags[u] & 2u) {
                const pos_t e = ta - b;
                lsh[u] = e < 63 ? (uint32_t)(63 - e) : 0u;
                s0 = e < 63 ? (pos_t)0 : e - 63;
            }
            pt[u] = s0;
            // (two byte loads per plane: one unaligned 2-byte load of the block and its neighbour measured 5 % slower)
            dr[u] = A.ref.dirty[pr[u] >> DIRTY_SHIFT] | A.ref.dirty[(pr[u] + 63) >> DIRTY_SHIFT];
            dt[u] = A.tig.dirty[s0 >> DIRTY_SHIFT] | A.tig.dirty[(s0 + 63) >> DIRTY_SHIFT];
            xr[u] = window_at<pos_t>(A.ref.two, pr[u]);
            xt[u] = window_at<pos_t>(A.tig.two, s0);
        }
        owners(base + VU * VPIECE, q_next);                             // while the loads above are in flight
#pragma unroll
        for (int u = 0; u < VU; ++u) {
            uint32_t r[4], t[4];
            align_dw(xr[u], (uint64_t)pr[u], r);
            align_dw(xt[u], (uint64_t)pt[u], t);
            uint64_t rm = 0, tm = 0;
            if (dr[u] | dt[u]) {                                         // rare: a non-ACGT base somewhere near
                const uint64_t keep = n[u] == 64 ? ~0ull : (1ull << n[u]) - 1ull;
                rm = window64(A.ref.mask, (uint64_t)pr[u]) & keep;
                tm = window64(A.tig.mask, (uint64_t)pt[u]);
                if (flags[u] & 2u) tm = __brevll(tm) >> lsh[u];
                tm &= keep;
            }
            if (flags[u] & 2u) {
                // bit reversal: bases in scan order with the two bits of every base swapped; swap them back and complement
                uint32_t v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t x = __brev(t[3 - i]);
                    v[i] = ~(((x & ~EVEN) >> 1) | ((x & EVEN) << 1));
                }
                if (lsh[u]) {                                            // window clipped at arena position 0 (first bases of record 0)
                    u128 w = (u128)v[3] << 96 | (u128)v[2] << 64 | (u128)v[1] << 32 | v[0];
                    w >>= 2 * lsh[u];
                    v[0] = (uint32_t)w; v[1] = (uint32_t)(w >> 32); v[2] = (uint32_t)(w >> 64); v[3] = (uint32_t)(w >> 96);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) t[i] = v[i];
            }
            // bit 2j of dword i: base 16 i + j differs; km keeps the n bases of the piece; 'X' pieces count the equal ones
            const uint32_t xm = (flags[u] & 1u) ? EVEN : 0u;
            const uint4 K = km_tbl[n[u]];
            const uint32_t km[4] = {K.x, K.y, K.z, K.w};
            uint32_t bad = 0;
            uint32_t neq[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t d = r[i] ^ t[i];
                neq[i] = (d | d >> 1) & km[i];
                bad += __popc((neq[i] ^ xm) & km[i]);
            }
            if (rm | tm) {
                const u128 nq = (u128)neq[3] << 96 | (u128)neq[2] << 64 | (u128)neq[1] << 32 | neq[0];
                const u128 kmask = (u128)km[3] << 96 | (u128)km[2] << 64 | (u128)km[1] << 32 | km[0];
                const u128 one_n = spread64(rm ^ tm), both_n = spread64(rm & tm);           // non-ACGT on one / on both sides
                // '=' : wrong when the codes differ (unless both are non-ACGT) or exactly one side is non-ACGT
                // 'X' : wrong when both are ACGT and equal, or both are non-ACGT (the planes cannot tell N from N)
                const u128 wrong = (flags[u] & 1u) ? ((~nq & kmask & ~one_n) | both_n) : ((nq & ~both_n) | one_n);
                bad = (uint32_t)popc128(wrong);
            }
            bad_tot += bad;                                              // 32-bit counters: < 2^38 / 256 bases per lane; '=' = all - 'X'
            if (flags[u] & 1u) bad_x += bad;
            if (bad) bad_slot = min(bad_slot, q[u]);
        }
#pragma unroll
        for (int u = 0; u < VU; ++u) q[u] = q_next[u];
    }
    unsigned long long n_eq = len_eq, bad_eq = bad_tot - bad_x, nx = len_x, bx = bad_x;
    // real ordinal of the operation: the tile's operations lead its slots
    unsigned long long first_bad = bad_slot == ~0u ? ~0ull : A.tile_pre[(uint64_t)NQ * (A.n_tiles + 1) + chunk] + slot0 + (unsigned long long)d_op[bad_slot];
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) {
        n_eq += __shfl_xor(n_eq, dd); bad_eq += __shfl_xor(bad_eq, dd); nx += __shfl_xor(nx, dd); bx += __shfl_xor(bx, dd);
        first_bad = min(first_bad, (unsigned long long)__shfl_xor(first_bad, dd));
    }
    // one atomic per workgroup and counter
    __shared__ unsigned long long part[4][5];
    if (lane == 0) { part[wave][0] = n_eq; part[wave][1] = bad_eq; part[wave][2] = nx; part[wave][3] = bx; part[wave][4] = first_bad; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const unsigned long long v = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        if (v) atomicAdd(&A.cnt[threadIdx.x], v);
    } else if (threadIdx.x == 4) {
        const unsigned long long v = min(min(part[0][4], part[1][4]), min(part[2][4], part[3][4]));
        if (v != ~0ull) atomicMin(&A.cnt[4], v);
    }
}

// ---- lift-over tables (pavlib/align/lift.py:380-476) ------------------------------------------------------------
// Per operation: subject position where it starts (absolute) and query position where it starts (alignment
// orientation, clips included).  Advance rules are AlignLift's: M, =, X move both axes, I / S / H the query,
// D the subject.  Same flat scan as the variant walk, two quantities.
__device__ __forceinline__ void lift_contrib(uint32_t op, uint64_t &sub, uint64_t &qry) {
    const uint32_t code = op & 15u;
    const uint64_t len = op >> 4;
    const bool match = code == 7 || code == 8 || code == 0;
    sub = (match || code == 2) ? len : 0;
    qry = (match || code == 1 || code == 4 || code == 5) ? len : 0;
}

__global__ __launch_bounds__(256) void lift_reduce(const uint32_t *__restrict__ ops, uint64_t n_ops,
                                                   uint64_t *__restrict__ chunk_sum /* [n_chunks][2] */) {
    __shared__ uint64_t lds[8];
    uint32_t o[OPS_PER_LANE];
    load_ops(ops, n_ops, (uint64_t)blockIdx.x * WALK_CHUNK + (uint64_t)threadIdx.x * OPS_PER_LANE, o);
    uint64_t a = 0, b = 0;
#pragma unroll
    for (int j = 0; j < OPS_PER_LANE; ++j) { uint64_t x, y; lift_contrib(o[j], x, y); a += x; b += y; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d); b += __shfl_xor(b, d); }
    if ((threadIdx.x & 63) == 0) { lds[(threadIdx.x >> 6) * 2] = a; lds[(threadIdx.x >> 6) * 2 + 1] = b; }
    __syncthreads();
    if (threadIdx.x < 2) chunk_sum[(uint64_t)blockIdx.x * 2 + threadIdx.x] = lds[threadIdx.x] + lds[2 + threadIdx.x] + lds[4 + threadIdx.x] + lds[6 + threadIdx.x];
}

__global__ __launch_bounds__(256) void lift_chunks(const uint64_t *__restrict__ chunk_sum, uint64_t *__restrict__ chunk_pre, uint32_t n_chunks) {
    constexpr int PER = 8, WORDS = PER * 2;                               // tiles of 2048 chunks through LDS, see scan_counts
    __shared__ uint64_t lds[8];
    __shared__ uint64_t tile[256 * WORDS];
    uint64_t carry[2] = {0, 0};
    const uint64_t n_words = 2ull * n_chunks;
    for (uint32_t base = 0; base < n_chunks; base += 256 * PER) {
        uint64_t w[WORDS];
#pragma unroll
        for (int k = 0; k < WORDS; ++k) {
            const uint64_t i = 2ull * base + (uint64_t)k * 256 + threadIdx.x;
            w[k] = i < n_words ? chunk_sum[i] : 0ull;
        }
#pragma unroll
        for (int k = 0; k < WORDS; ++k) tile[k * 256 + threadIdx.x] = w[k];
        __syncthreads();
        uint64_t v[2] = {0, 0}, tot[2];
#pragma unroll
        for (int k = 0; k < WORDS; ++k) { w[k] = tile[threadIdx.x * WORDS + k]; v[k & 1] += w[k]; }
        block_excl_scan<2>(v, tot, lds);
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const uint32_t i = base + threadIdx.x * PER + j;
            if (i < n_chunks) { chunk_pre[2ull * i] = carry[0] + v[0]; chunk_pre[2ull * i + 1] = carry[1] + v[1]; }
            v[0] += w[2 * j]; v[1] += w[2 * j + 1];
        }
        carry[0] += tot[0]; carry[1] += tot[1];
    }
    if (threadIdx.x < 2) chunk_pre[2ull * n_chunks + threadIdx.x] = threadIdx.x ? carry[1] : carry[0];   // rows starting at n_ops
}

__global__ __launch_bounds__(256) void lift_row_base(const uint32_t *__restrict__ ops, const uint64_t *__restrict__ op_off,
                                                     const uint64_t *__restrict__ chunk_pre, uint64_t *__restrict__ rowbase, uint32_t n_aln) {
    const uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_aln) return;
    const int lane = threadIdx.x & 63;
    const uint64_t first = op_off[r];
    const uint64_t c = first / WALK_CHUNK;
    uint64_t a = 0, b = 0;
    for (uint64_t i = c * WALK_CHUNK + lane; i < first; i += 64) { uint64_t x, y; lift_contrib(ops[i], x, y); a += x; b += y; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d); b += __shfl_xor(b, d); }
    if (lane == 0) {
        rowbase[2ull * r] = chunk_pre[2ull * c] + a;          // chunk_pre[n_chunks] = grand total: rows that start at n_ops
        rowbase[2ull * r + 1] = chunk_pre[2ull * c + 1] + b;
    }
}

__global__ __launch_bounds__(256) void lift_positions(const uint32_t *__restrict__ ops, uint64_t n_ops,
                                                      const uint64_t *__restrict__ op_off, const uint32_t *__restrict__ row_pos,
                                                      uint32_t n_aln, const uint64_t *__restrict__ chunk_pre,
                                                      const uint64_t *__restrict__ rowbase, uint32_t *__restrict__ sub_begin,
                                                      uint32_t *__restrict__ qry_begin) {
    __shared__ uint64_t lds[8];
    const uint64_t first = (uint64_t)blockIdx.x * WALK_CHUNK + (uint64_t)threadIdx.x * OPS_PER_LANE;
    uint32_t o[OPS_PER_LANE];
    load_ops(ops, n_ops, first, o);
    uint64_t run[2] = {0, 0}, tot[2];
#pragma unroll
    for (int j = 0; j < OPS_PER_LANE; ++j) { uint64_t x, y; lift_contrib(o[j], x, y); run[0] += x; run[1] += y; }
    block_excl_scan<2>(run, tot, lds);
    run[0] += chunk_pre[2ull * blockIdx.x]; run[1] += chunk_pre[2ull * blockIdx.x + 1];
    if (first >= n_ops) return;
    uint32_t lo = 0, hi = n_aln;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (op_off[mid] <= first) lo = mid; else hi = mid; }
    uint32_t row = lo;
    uint64_t row_end = op_off[row + 1];
    uint64_t rb0 = rowbase[2ull * row], rb1 = rowbase[2ull * row + 1];
    uint32_t pos0 = row_pos[row];
#pragma unroll
    for (int j = 0; j < OPS_PER_LANE; ++j) {
        const uint64_t k = first + j;
        if (k >= n_ops) break;
        while (k >= row_end) { ++row; row_end = op_off[row + 1]; rb0 = rowbase[2ull * row]; rb1 = rowbase[2ull * row + 1]; pos0 = row_pos[row]; }
        sub_begin[k] = pos0 + (uint32_t)(run[0] - rb0);
        qry_begin[k] = (uint32_t)(run[1] - rb1);
        uint64_t x, y; lift_contrib(o[j], x, y); run[0] += x; run[1] += y;
    }
}

// ---- breakpoint homology (pavlib/call.py:542-647) -----------------------------------------------------------
// The reference walks one base at a time; here a lane compares up to 32 bases per step straight from the packed
// planes: an unaligned 64-bit window of the 2-bit plane (and 32 bits of the non-ACGT plane) for the scanned
// sequence against the SV sequence laid out in scan order.  The circular index into seq_sv (call.py:579,637)
// becomes a periodic pattern: for svlen <= 32 one period is fetched once and replicated so that every step
// starts at phase 0; longer SVs are compared window by window up to the wrap point.
struct SeqRef {                 // one oriented record of a store
    const uint32_t *two, *mask; // planes of a packed store; of a store whose planes are packed on demand (contigs, ctx.hip
    const uint8_t *dirty;       // "lazy contig pack"): two = nullptr, dirty = the ASCII arena - fetch_run decodes the bytes
    uint64_t off, len;
    int rev;
};
__device__ __forceinline__ SeqRef seq_ref(const SeqView &v, uint32_t id, int rev) {
    if (v.packed) return SeqRef{v.two, v.mask, v.dirty, v.off[id], v.len[id], rev};
    return SeqRef{nullptr, nullptr, v.ascii, v.off[id], v.len[id], rev};
}

__device__ __forceinline__ uint64_t low_bits64(int n) { return n >= 64 ? ~0ull : ((1ull << n) - 1ull); }

// Reverse the order of the n 2-bit groups held in the low 2n bits of x.
__device__ __forceinline__ uint64_t reverse_groups(uint64_t x, int n) {
    uint64_t r = __brevll(x) >> (64 - 2 * n);
    return ((r & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((r & 0x5555555555555555ull) << 1);
}

// n (1..32) bases of s starting at oriented position p and continuing in direction dir (+1 / -1), returned in
// scan order: base j in bits [2j, 2j+1] of codes, non-ACGT flag in bit j of bad.
__device__ __forceinline__ void fetch_run(const SeqRef &s, int64_t p, int dir, int n, uint64_t &codes, uint32_t &bad) {
    const int64_t st = s.rev ? (int64_t)s.len - 1 - p : p;          // stored position of the first scanned base
    const int sdir = s.rev ? -dir : dir;                              // direction in stored coordinates
    const uint64_t abs = s.off + (uint64_t)(sdir > 0 ? st : st - (n - 1));
    uint64_t x;
    uint32_t m = 0;
    if (!s.two) {
        // ASCII arena: nine dwords from the dword that holds byte abs, moved into place with v_alignbyte, four bases per
        // dword through the pack's own conversion (pad blocks and anything that is not ACGT / acgt come out as non-ACGT bits)
        const uint32_t *a4 = reinterpret_cast<const uint32_t *>(s.dirty + (abs & ~3ull));
        const u32x4_a4 v0 = *reinterpret_cast<const u32x4_a4 *>(a4), v1 = *reinterpret_cast<const u32x4_a4 *>(a4 + 4);
        const uint32_t w[9] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, a4[8]};
        const uint32_t sh = (uint32_t)abs & 3u;
        x = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint32_t c8, b4;
            pack4(__builtin_amdgcn_alignbyte(w[i + 1], w[i], sh), c8, b4);
            x |= (uint64_t)c8 << (8 * i);
            m |= b4 << (4 * i);
        }
    } else {
        const uint64_t *two64 = reinterpret_cast<const uint64_t *>(s.two);
        const uint64_t *mask64 = reinterpret_cast<const uint64_t *>(s.mask);
        const uint64_t w = abs >> 5;
        const int b = (int)(abs & 31) * 2;
        x = two64[w] >> b;
        if (b) x |= two64[w + 1] << (64 - b);
        // non-ACGT bits: only blocks the summary marks hold any (SeqView::dirty)
        const uint64_t d0 = abs >> DIRTY_SHIFT, d1 = (abs + (uint64_t)(n - 1)) >> DIRTY_SHIFT;
        uint32_t dirty = s.dirty[d0];
        if (d1 != d0) dirty |= s.dirty[d1];
        if (dirty) {
            const uint64_t w2 = abs >> 6;
            const int b2 = (int)(abs & 63);
            uint64_t y = mask64[w2] >> b2;
            if (b2) y |= mask64[w2 + 1] << (64 - b2);
            m = (uint32_t)y;
        }
    }
    if (n < 32) { x &= low_bits64(2 * n); m &= (1u << n) - 1u; }
    if (sdir < 0) { x = reverse_groups(x, n); m = __brev(m) >> (32 - n); }
    if (s.rev) x ^= low_bits64(2 * n);                                // complement every base
    codes = x;
    bad = m;
}

// First scan offset (0..m-1) where the bases differ or either side is non-ACGT; m when none.
__device__ __forceinline__ int first_stop(uint64_t a, uint64_t b, uint32_t bad, int m) {
    uint64_t d = a ^ b;
    d = (d | (d >> 1)) & 0x5555555555555555ull & low_bits64(2 * m);
    if (m < 32) bad &= (1u << m) - 1u;
    const int pd = d ? (__ffsll((long long)d) - 1) >> 1 : 64;
    const int pm = bad ? __ffs((int)bad) - 1 : 64;
    const int st = pd < pm ? pd : pm;
    return st < m ? st : m;
}

// Shared scan: t is walked from t_pos in direction dir for at most avail bases; seq_sv = sv[sv_pos, sv_pos+svlen)
// is walked circularly from its last base backwards (dir < 0) or its first base forwards (dir > 0).
// At most max_steps windows are examined; `done` tells whether the scan ended (mismatch, non-ACGT or edge).
// One period of seq_sv in forward order (svlen <= 32), fetched once per INS / DEL and shared by its scans: the scan that walks
// backwards reads it reversed.
struct Period { uint64_t c; uint32_t m; };
__device__ __forceinline__ void period_in_scan_order(const Period &per, int dir, int L, uint64_t &pc, uint32_t &pm) {
    if (dir > 0) { pc = per.c; pm = per.m; }
    else { pc = reverse_groups(per.c, L); pm = __brev(per.m) >> (32 - L); }
}

__device__ uint32_t hom_scan(const SeqRef &t, int64_t t_pos, int dir, int64_t avail, const SeqRef &sv, int64_t sv_pos,
                             int64_t svlen, int max_steps, bool &done, bool has_per = false, Period per = Period{0, 0}) {
    done = true;
    if (svlen <= 0 || avail <= 0) return 0;
    int64_t h = 0;
    if (svlen <= 32) {
        const int L = (int)svlen;
        uint64_t pc; uint32_t pm;
        if (has_per) period_in_scan_order(per, dir, L, pc, pm);
        else fetch_run(sv, dir < 0 ? sv_pos + svlen - 1 : sv_pos, dir, L, pc, pm);
        const int reps = 32 / L, n = reps * L;
        uint64_t P = pc; uint32_t M = pm;
        for (int r = 1; r < reps; ++r) { P |= pc << (2 * L * r); M |= pm << (L * r); }
        for (int step = 0; h < avail; ++step) {
            if (step == max_steps) { done = false; return (uint32_t)h; }
            const int m = (int)(avail - h < n ? avail - h : n);
            uint64_t tc; uint32_t tm;
            fetch_run(t, t_pos + dir * h, dir, m, tc, tm);
            const int st = first_stop(tc, P, tm | M, m);
            if (st < m) return (uint32_t)(h + st);
            h += m;
        }
        return (uint32_t)h;
    }
    int64_t idx = dir < 0 ? svlen - 1 : 0;
    for (int step = 0; h < avail; ++step) {
        if (step == max_steps) { done = false; return (uint32_t)h; }
        const int64_t to_wrap = dir < 0 ? idx + 1 : svlen - idx;
        int64_t mm = avail - h < 32 ? avail - h : 32;
        if (to_wrap < mm) mm = to_wrap;
        const int m = (int)mm;
        uint64_t tc, sc; uint32_t tm, sm;
        fetch_run(t, t_pos + dir * h, dir, m, tc, tm);
        fetch_run(sv, sv_pos + idx, dir, m, sc, sm);
        const int st = first_stop(tc, sc, tm | sm, m);
        if (st < m) return (uint32_t)(h + st);
        h += m;
        idx += dir * m;
        if (idx < 0) idx = svlen - 1; else if (idx >= svlen) idx = 0;
    }
    return (uint32_t)h;
}

__device__ __forceinline__ int64_t bcast64(int64_t v, int src) { return (int64_t)__shfl((long long)v, src); }

// Wave-cooperative scan.  Every lane first walks its own scan for a few windows (most homologies are short); scans
// still running (long tandem copies, kilobases) are then finished one at a time by the whole wave, lane c taking
// the c-th window, so a 5 kb homology costs a handful of steps instead of ~160 dependent ones.
// Must be called by all 64 lanes; `active` masks lanes without work.  `h0` resumes a scan known to match up to h0.
__device__ uint32_t wave_hom_scan(bool active, const SeqRef &t, int64_t t_pos, int dir, int64_t avail, const SeqRef &sv,
                                  int64_t sv_pos, int64_t svlen, bool has_per = false, Period per = Period{0, 0}) {
    const int lane = threadIdx.x & 63;
    bool done = true;
    uint32_t res = 0;
    if (active) res = hom_scan(t, t_pos, dir, avail, sv, sv_pos, svlen, 3, done, has_per, per);
    unsigned long long pending = __ballot(active && !done);
    while (pending) {
        const int src = __ffsll((long long)pending) - 1;
        pending &= pending - 1;
        SeqRef bt{t.two, t.mask, t.dirty, (uint64_t)bcast64((int64_t)t.off, src), (uint64_t)bcast64((int64_t)t.len, src), __shfl(t.rev, src)};
        SeqRef bs{sv.two, sv.mask, sv.dirty, (uint64_t)bcast64((int64_t)sv.off, src), (uint64_t)bcast64((int64_t)sv.len, src), __shfl(sv.rev, src)};
        // the two planes differ between stores (reference / contig): broadcast the pointers as well
        bt.two = (const uint32_t *)bcast64((int64_t)t.two, src); bt.mask = (const uint32_t *)bcast64((int64_t)t.mask, src);
        bs.two = (const uint32_t *)bcast64((int64_t)sv.two, src); bs.mask = (const uint32_t *)bcast64((int64_t)sv.mask, src);
        bt.dirty = (const uint8_t *)bcast64((int64_t)t.dirty, src); bs.dirty = (const uint8_t *)bcast64((int64_t)sv.dirty, src);
        const int64_t b_pos = bcast64(t_pos, src), b_avail = bcast64(avail, src), b_svpos = bcast64(sv_pos, src),
                      b_svlen = bcast64(svlen, src);
        const int b_dir = __shfl(dir, src);
        int64_t h = (int64_t)__shfl(res, src);
        uint64_t P = 0; uint32_t M = 0;
        int n = 32;
        const bool periodic = b_svlen <= 32;
        if (periodic) {
            const int L = (int)b_svlen;
            uint64_t pc; uint32_t pm;
            fetch_run(bs, b_dir < 0 ? b_svpos + b_svlen - 1 : b_svpos, b_dir, L, pc, pm);
            const int reps = 32 / L;
            n = reps * L;
            P = pc; M = pm;
            for (int r = 1; r < reps; ++r) { P |= pc << (2 * L * r); M |= pm << (L * r); }
        }
        int64_t found_at = -1;
        while (found_at < 0) {
            const int64_t my_h = h + (int64_t)lane * n;
            int64_t mm = b_avail - my_h;
            if (mm > n) mm = n;
            if (mm < 0) mm = 0;
            const int m = (int)mm;
            int st = m;
            if (m > 0) {
                uint64_t tc, sc; uint32_t tm, sm;
                fetch_run(bt, b_pos + b_dir * my_h, b_dir, m, tc, tm);
                if (periodic) { sc = P; sm = M; }
                else {
                    // h is a multiple of... nothing in general: index of the SV base matching scan offset my_h
                    int64_t idx = b_dir < 0 ? (b_svlen - 1 - (my_h % b_svlen)) : (my_h % b_svlen);
                    const int64_t to_wrap = b_dir < 0 ? idx + 1 : b_svlen - idx;
                    const int m1 = (int)(to_wrap < m ? to_wrap : m);
                    fetch_run(bs, b_svpos + idx, b_dir, m1, sc, sm);
                    if (m1 < m) {
                        uint64_t sc2; uint32_t sm2;
                        fetch_run(bs, b_svpos + (b_dir < 0 ? b_svlen - 1 : 0), b_dir, m - m1, sc2, sm2);
                        sc |= sc2 << (2 * m1); sm |= sm2 << m1;
                    }
                }
                st = first_stop(tc, sc, tm | sm, m);
            }
            const bool stop_here = st < m || m < n;               // mismatch / non-ACGT in my window, or the edge is in it
            const unsigned long long hit = __ballot(stop_here);
            if (hit) {
                const int first = __ffsll((long long)hit) - 1;
                found_at = bcast64(my_h + st, first);
            } else {
                h += 64ll * n;
            }
        }
        if (lane == src) res = (uint32_t)found_at;
    }
    return res;
}

// The four breakpoint scans of one INS / DEL (cigarcall.py:178-182, 247-251) share seq_sv and are independent of each other:
// they advance in lockstep, so that a lane has the windows of all its unfinished scans in flight at once instead of one
// dependent load chain per scan (the kernel is latency-bound: 88 % of its wave cycles are spent parked in s_waitcnt).
// Scan s walks t[s] from t_pos[s] in direction dir[s] (-1 left_homology / +1 right_homology) for at most avail[s] bases.
// Up to three windows per scan and lane; scans still running are finished by the whole wave as in wave_hom_scan.
struct ScanSide { SeqRef t; int64_t t_pos, avail; int dir; };

__device__ void wave_hom_scan4(bool active, const ScanSide (&sc)[4], const SeqRef &sv, int64_t sv_pos, int64_t svlen, uint32_t (&res)[4],
                               const Period &per) {
    const int lane = threadIdx.x & 63;
    int64_t h[4] = {0, 0, 0, 0};
    bool done[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) done[s] = !active || svlen <= 0 || sc[s].avail <= 0;
    const bool periodic = svlen <= 32;
    // one period of seq_sv in either scan order, replicated to a whole number of periods (phase 0 at every window start)
    uint64_t P[2] = {0, 0}; uint32_t M[2] = {0, 0};
    int n = 32;
    if (active && periodic && svlen > 0) {
        const int L = (int)svlen;
        uint64_t pc[2]; uint32_t pm[2];
        period_in_scan_order(per, -1, L, pc[0], pm[0]);
        period_in_scan_order(per, +1, L, pc[1], pm[1]);
        const int reps = 32 / L;
        n = reps * L;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            uint64_t x = pc[d]; uint32_t y = pm[d];
            for (int have = 1; have < reps; ) {                    // doubling: have periods -> min(2 have, reps)
                const int add = have < reps - have ? have : reps - have;
                x |= (x & low_bits64(2 * L * add)) << (2 * L * have);
                y |= (y & (uint32_t)low_bits64(L * add)) << (L * have);
                have += add;
            }
            P[d] = x; M[d] = y;
        }
    }
    for (int step = 0; step < 3; ++step) {
        int m[4]; uint64_t tc[4], pc[4]; uint32_t tm[4], pm[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {                               // every load of this step is issued before any is used
            m[s] = 0;
            if (done[s]) continue;
            const int dir = sc[s].dir;
            int64_t mm = sc[s].avail - h[s] < n ? sc[s].avail - h[s] : n;
            if (periodic) { pc[s] = P[dir > 0]; pm[s] = M[dir > 0]; }
            else {
                const int64_t idx = dir < 0 ? svlen - 1 - (h[s] % svlen) : h[s] % svlen;
                const int64_t to_wrap = dir < 0 ? idx + 1 : svlen - idx;
                if (to_wrap < mm) mm = to_wrap;
                fetch_run(sv, sv_pos + idx, dir, (int)mm, pc[s], pm[s]);
            }
            m[s] = (int)mm;
            fetch_run(sc[s].t, sc[s].t_pos + dir * h[s], dir, m[s], tc[s], tm[s]);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (done[s]) continue;
            const int st = first_stop(tc[s], pc[s], tm[s] | pm[s], m[s]);
            h[s] += st;
            if (st < m[s] || h[s] >= sc[s].avail) done[s] = true;
        }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        res[s] = (uint32_t)h[s];
        unsigned long long pending = __ballot(!done[s]);
        while (pending) {                                           // long scan: lane c takes the c-th window from h on
            const int src = __ffsll((long long)pending) - 1;
            pending &= pending - 1;
            SeqRef bt, bs;
            bt.two = (const uint32_t *)bcast64((int64_t)sc[s].t.two, src); bt.mask = (const uint32_t *)bcast64((int64_t)sc[s].t.mask, src);
            bt.off = (uint64_t)bcast64((int64_t)sc[s].t.off, src); bt.len = (uint64_t)bcast64((int64_t)sc[s].t.len, src); bt.rev = __shfl(sc[s].t.rev, src);
            bs.two = (const uint32_t *)bcast64((int64_t)sv.two, src); bs.mask = (const uint32_t *)bcast64((int64_t)sv.mask, src);
            bt.dirty = (const uint8_t *)bcast64((int64_t)sc[s].t.dirty, src); bs.dirty = (const uint8_t *)bcast64((int64_t)sv.dirty, src);
            bs.off = (uint64_t)bcast64((int64_t)sv.off, src); bs.len = (uint64_t)bcast64((int64_t)sv.len, src); bs.rev = __shfl(sv.rev, src);
            const int64_t b_pos = bcast64(sc[s].t_pos, src), b_avail = bcast64(sc[s].avail, src), b_svpos = bcast64(sv_pos, src),
                          b_svlen = bcast64(svlen, src);
            const int b_dir = sc[s].dir;                            // the direction of scan s is the same in every lane
            const bool b_periodic = b_svlen <= 32;
            const int b_n = __shfl(n, src);
            const uint64_t bP = (uint64_t)bcast64((int64_t)P[b_dir > 0], src);
            const uint32_t bM = (uint32_t)__shfl((int)M[b_dir > 0], src);
            int64_t hh = bcast64(h[s], src);
            int64_t found_at = -1;
            while (found_at < 0) {
                const int64_t my_h = hh + (int64_t)lane * b_n;
                int64_t mm = b_avail - my_h;
                if (mm > b_n) mm = b_n;
                if (mm < 0) mm = 0;
                const int mw = (int)mm;
                int st = mw;
                if (mw > 0) {
                    uint64_t tcw, scw; uint32_t tmw, smw;
                    fetch_run(bt, b_pos + b_dir * my_h, b_dir, mw, tcw, tmw);
                    if (b_periodic) { scw = bP; smw = bM; }
                    else {
                        const int64_t idx = b_dir < 0 ? (b_svlen - 1 - (my_h % b_svlen)) : (my_h % b_svlen);
                        const int64_t to_wrap = b_dir < 0 ? idx + 1 : b_svlen - idx;
                        const int m1 = (int)(to_wrap < mw ? to_wrap : mw);
                        fetch_run(bs, b_svpos + idx, b_dir, m1, scw, smw);
                        if (m1 < mw) {
                            uint64_t sc2; uint32_t sm2;
                            fetch_run(bs, b_svpos + (b_dir < 0 ? b_svlen - 1 : 0), b_dir, mw - m1, sc2, sm2);
                            scw |= sc2 << (2 * m1); smw |= sm2 << m1;
                        }
                    }
                    st = first_stop(tcw, scw, tmw | smw, mw);
                }
                const bool stop_here = st < mw || mw < b_n;
                const unsigned long long hit = __ballot(stop_here);
                if (hit) found_at = bcast64(my_h + st, __ffsll((long long)hit) - 1);
                else hh += 64ll * b_n;
            }
            if (lane == src) res[s] = (uint32_t)found_at;
        }
    }
}

// left_homology(pos_tig, seq_tig, seq_sv)   call.py:542-592: walks upstream from pos while hom_len <= pos_tig
__device__ __forceinline__ uint32_t left_hom(const SeqRef &t, int64_t pos, const SeqRef &sv, int64_t sv_pos, int64_t svlen) {
    bool done;
    return hom_scan(t, pos, -1, pos + 1, sv, sv_pos, svlen, 0x7FFFFFFF, done);
}
__device__ __forceinline__ uint32_t wave_left_hom(bool active, const SeqRef &t, int64_t pos, const SeqRef &sv, int64_t sv_pos, int64_t svlen,
                                                  bool has_per, Period per) {      // (by value: a pointer to the caller's copy kept it in scratch memory)
    return wave_hom_scan(active, t, pos, -1, pos + 1, sv, sv_pos, svlen, has_per, per);
}

// right_homology(pos_tig, seq_tig, seq_sv)  call.py:595-647: walks downstream while hom_len < len - pos_tig
__device__ __forceinline__ uint32_t right_hom(const SeqRef &t, int64_t pos, const SeqRef &sv, int64_t sv_pos, int64_t svlen) {
    bool done;
    return hom_scan(t, pos, +1, (int64_t)t.len - pos, sv, sv_pos, svlen, 0x7FFFFFFF, done);
}

// One lane per INS/DEL stub: left shift, then the four breakpoint homologies in lockstep (wave_hom_scan4), then the final
// coordinates.  The scans are wave-uniform calls (long scans are finished cooperatively), so no lane leaves early.
// (Four waves per SIMD instead of three - amdgpu_waves_per_eu(4, 4): 128 VGPRs, 112 B of scratch per lane - measured 9 % slower.)
// Epilogue: the SEQ column (cigarcall.py:145,163,221).  The records of a wave are consecutive and so are their sequences in the
// blob: the wave copies the span byte by byte, lane = output byte (coalesced stores), the record that owns a byte is found among
// the wave's 64 by a search over shuffled offsets - and for an INS the bytes come from the contig lines the scans above have just
// fetched.  (Round 2 had a separate seq_gather kernel that re-read every record: one launch and 64 B per record more.)
__global__ __launch_bounds__(64) void homology_kernel(pav_indel *__restrict__ indel, uint64_t n_indel,
                                                      const pav_aln *__restrict__ aln, SeqView R, SeqView T,
                                                      uint8_t *__restrict__ blob) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool active = i < n_indel;
    pav_indel r = indel[active ? i : n_indel - 1];
    // everything about the row's records comes with the stub (walk_indel): no look-ups of the alignment row and the record tables
    const int rev = r.pad[0] != 0;
    (void)aln;
    SeqRef ref, tig;
    ref.off = (uint64_t)r.hom_ref_l | (uint64_t)r.hom_ref_r << 32; ref.len = r.end; ref.rev = 0;
    tig.off = (uint64_t)r.hom_tig_l | (uint64_t)r.hom_tig_r << 32; tig.len = r.qry_end; tig.rev = rev;
    if (R.packed) { ref.two = R.two; ref.mask = R.mask; ref.dirty = R.dirty; } else { ref.two = nullptr; ref.mask = nullptr; ref.dirty = R.ascii; }
    if (T.packed) { tig.two = T.two; tig.mask = T.mask; tig.dirty = T.dirty; } else { tig.two = nullptr; tig.mask = nullptr; tig.dirty = T.ascii; }
    r.pad[0] = 0;
    const int64_t pos_ref = r.pos, pos_tig = r.qry_pos, oplen = r.svlen, tig_len = (int64_t)tig.len;
    const bool ins = r.svtype == 0;
    const SeqRef svs = ins ? tig : ref;                // seq = seq_tig[pos_tig:+oplen] / seq_ref[pos_ref:+oplen]
    int64_t sv_at = ins ? pos_tig : pos_ref;
    // last_op == '=' (cigarcall.py:149-155 / :225-231): shift = min(last_oplen, left_homology(pos_ref - 1, REF, SEQ))
    // a short SV sequence is one period of the circular comparison: fetched once, shared by the five scans (it used to be
    // fetched by the shift scan and twice more - in either order - by the breakpoint scans: for an INS three windows of the ASCII arena)
    const bool periodic = active && oplen > 0 && oplen <= 32;
    Period per{0, 0};
    if (periodic) fetch_run(svs, sv_at, +1, (int)oplen, per.c, per.m);
    const uint32_t hs = wave_left_hom(active && r.left_shift != 0, ref, pos_ref - 1, svs, sv_at, oplen, periodic, per);
    const int64_t shift = r.left_shift ? ((int64_t)hs < (int64_t)r.left_shift ? (int64_t)hs : (int64_t)r.left_shift) : 0;
    const int64_t sv_pos_ref = pos_ref - shift, sv_pos_tig = pos_tig - shift;
    if (ins && shift) {                                // INS: seq re-sliced at the shifted position (:162-163)
        sv_at = sv_pos_tig;
        if (periodic) fetch_run(svs, sv_at, +1, (int)oplen, per.c, per.m);
    }
    // INS: :178-182;  DEL: :247-251
    const int64_t p_rr = ins ? sv_pos_ref : sv_pos_ref + oplen, p_tr = ins ? sv_pos_tig + oplen : sv_pos_tig;
    const ScanSide sides[4] = {{ref, sv_pos_ref - 1, sv_pos_ref, -1}, {ref, p_rr, (int64_t)ref.len - p_rr, +1},
                               {tig, sv_pos_tig - 1, sv_pos_tig, -1}, {tig, p_tr, tig_len - p_tr, +1}};
    uint32_t hom[4];
    wave_hom_scan4(active, sides, svs, sv_at, oplen, hom, per);
    const uint64_t seq_off = r.seq_off;
    if (active) {
        r.hom_ref_l = hom[0]; r.hom_ref_r = hom[1]; r.hom_tig_l = hom[2]; r.hom_tig_r = hom[3];
        if (ins) {
            r.pos = (uint32_t)sv_pos_ref; r.end = (uint32_t)(sv_pos_ref + 1);           // :157-158
            if (rev) { r.qry_end = (uint32_t)(tig_len - sv_pos_tig); r.qry_pos = r.qry_end - (uint32_t)oplen; }   // :167-169
            else { r.qry_pos = (uint32_t)sv_pos_tig; r.qry_end = (uint32_t)(sv_pos_tig + oplen); }                // :171-173
        } else {
            r.pos = (uint32_t)pos_ref; r.end = (uint32_t)(pos_ref + oplen);             // :258 (un-shifted)
            const int64_t q = rev ? tig_len - sv_pos_tig : sv_pos_tig;                  // :239-242
            r.qry_pos = (uint32_t)q; r.qry_end = (uint32_t)(q + 1);
        }
        r.left_shift = (uint32_t)shift;
        indel[i] = r;
    }
    // ---- SEQ bytes of the wave's records: INS seq_tig[sv_pos_tig : +oplen] (oriented), DEL seq_ref[pos_ref : +oplen] ----
    // byte k of this lane's sequence sits at arena position s0 + sdir * k of the contig (complemented on reverse rows) / reference
    const int flags = (ins ? 1 : 0) | ((ins && rev) ? 2 : 0);
    const int64_t s0 = ins ? (int64_t)tig.off + (rev ? tig_len - 1 - sv_at : sv_at) : (int64_t)ref.off + pos_ref;
    // (the owner search below takes a different path in every lane: the wave's 64 offsets go through LDS, not through shuffles,
    //  which would read lanes that have left the loop)
    __shared__ uint64_t s_rel[64];                  // 64 sequences of up to 2^28 bases each: 2^34, more than 32 bits hold
    __shared__ int64_t s_src[64];
    __shared__ int s_fl[64];
    const uint64_t span0 = (uint64_t)__shfl((long long)seq_off, 0);                     // lane 0 is active in every launched wave
    const uint32_t n_act = n_indel - (uint64_t)blockIdx.x * 64 < 64 ? (uint32_t)(n_indel - (uint64_t)blockIdx.x * 64) : 64u;
    s_rel[lane] = active ? seq_off - span0 : ~0ull;
    s_src[lane] = s0; s_fl[lane] = flags;
    __syncthreads();
    const uint64_t span = s_rel[n_act - 1] + (uint64_t)(uint32_t)__shfl((int)oplen, (int)n_act - 1);
    for (uint64_t b = (uint64_t)lane; b < span; b += 64) {
        uint32_t lo = 0, hi = n_act;                    // last record whose sequence starts at or before byte b
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (s_rel[mid] <= b) lo = mid; else hi = mid; }
        const uint64_t k = b - s_rel[lo];
        const int64_t base = s_src[lo];
        const int fl = s_fl[lo];
        const uint8_t *plane = (fl & 1) ? T.ascii : R.ascii;
        const uint8_t c = plane[(fl & 2) ? base - (int64_t)k : base + (int64_t)k];
        blob[span0 + b] = (fl & 2) ? comp_ascii(c) : c;
    }
}

__global__ void homology_query_kernel(const pav_hom_query *__restrict__ q, uint32_t n, SeqView R, SeqView T,
                                      uint32_t *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const pav_hom_query h = q[i];
    // (one store or the other, field by field: a reference picked at run time would keep both views in scratch memory)
    auto pick = [&](bool is_ref, uint32_t id, int rev) {
        const bool packed = (is_ref ? R.packed : T.packed) != 0;
        const uint64_t off = is_ref ? R.off[id] : T.off[id], len = is_ref ? R.len[id] : T.len[id];
        if (packed) return SeqRef{is_ref ? R.two : T.two, is_ref ? R.mask : T.mask, is_ref ? R.dirty : T.dirty, off, len, rev};
        return SeqRef{nullptr, nullptr, is_ref ? R.ascii : T.ascii, off, len, rev};
    };
    const SeqRef t = pick(h.role == PAV_ROLE_REF, (uint32_t)h.seq_id, h.rev != 0), sv = pick(h.sv_role == PAV_ROLE_REF, (uint32_t)h.sv_seq_id, h.sv_rev != 0);
    out[i] = h.dir == 0 ? left_hom(t, h.pos, sv, h.sv_pos, h.svlen) : right_hom(t, h.pos, sv, h.sv_pos, h.svlen);
}

}  // namespace pav

using namespace pav;

static uint64_t round_up(uint64_t x, uint64_t m) { return (x + m - 1) / m * m; }

extern "C" {

int pav_cigar_load(pav_ctx *ctx, uint32_t n_aln, const pav_aln *aln, const uint8_t *cigar_text,
                   const uint64_t *cigar_off) {
    if (!ctx) return PAV_E_ARG;
    if (n_aln && (!aln || !cigar_off)) return fail(ctx, PAV_E_ARG, "pav_cigar_load: null input");
    table_writer_quiesce(ctx);                                          // a device table write still reads the rows and the records
    PAV_HIP(ctx, hipSetDevice(ctx->device));
    { const int rch = wait_homology(ctx); if (rch != PAV_OK) return rch; }     // the scans of the last call read the alignment rows
    const uint64_t T = n_aln ? cigar_off[n_aln] : 0;
    if (T && !cigar_text) return fail(ctx, PAV_E_ARG, "pav_cigar_load: null CIGAR text");
    for (uint32_t r = 0; r < n_aln; ++r) {
        if (cigar_off[r + 1] < cigar_off[r]) return fail(ctx, PAV_E_ARG, "pav_cigar_load: cigar_off not monotone at row %u", r);
        if (aln[r].ref_id >= ctx->seq[PAV_ROLE_REF].n || aln[r].tig_id >= ctx->seq[PAV_ROLE_TIG].n)
            return fail(ctx, PAV_E_ARG, "pav_cigar_load: row %u references a sequence that is not loaded", r);
    }
    ctx->n_aln = n_aln;
    ctx->text_bytes = T;
    ctx->cigar_called = false;
    const uint64_t Tpad = round_up(T + 1, TOK_CHUNK);
    PAV_HIP(ctx, ctx->d_text.reserve(Tpad));
    PAV_HIP(ctx, ctx->d_text_off.reserve(sizeof(uint64_t) * ((size_t)n_aln + 1)));
    PAV_HIP(ctx, ctx->d_aln.reserve(sizeof(pav_aln) * ((size_t)n_aln + 1)));
    PAV_HIP(ctx, hipMemsetAsync(ctx->d_text.p, '0', Tpad, ctx->stream));          // padding: digits, never ops
    if (T) PAV_HIP(ctx, hipMemcpyAsync(ctx->d_text.p, cigar_text, T, hipMemcpyHostToDevice, ctx->stream));
    uint64_t zero = 0;
    PAV_HIP(ctx, hipMemcpyAsync(ctx->d_text_off.p, n_aln ? cigar_off : &zero, sizeof(uint64_t) * ((size_t)n_aln + 1),
                                hipMemcpyHostToDevice, ctx->stream));
    if (n_aln) PAV_HIP(ctx, hipMemcpyAsync(ctx->d_aln.p, aln, sizeof(pav_aln) * n_aln, hipMemcpyHostToDevice, ctx->stream));
    PAV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->cigar_loaded = true;
    return PAV_OK;
}

static int pav_cigar_fetch_ops_unchecked(pav_ctx *ctx, uint32_t *ops, uint64_t *op_off);

namespace {
// Device buffers of the tile tokenizer + walk, carved out of the context's grow-only buffers.  Every size follows from the text
// size and the row count, so nothing has to be read back before the kernels can be queued.
struct CallBufs {
    uint32_t n_tiles = 0;
    uint32_t *ops = nullptr;                 // [n_tiles * 2048] padded operation slots
    uint64_t *tile_pre = nullptr;            // [NT][n_tiles + 1]
    uint32_t *tile_last = nullptr, *chunk_row = nullptr;
    uint64_t *op_off = nullptr, *op_slot = nullptr, *rowbase = nullptr, *row_local = nullptr;
    uint32_t *row_tile = nullptr, *row_i = nullptr;
    unsigned long long *tok_err = nullptr, *err_op = nullptr;
};

int call_bufs(pav_ctx *ctx, CallBufs &B, bool reserve) {
    const uint64_t Tpad = (ctx->text_bytes + 1 + TOK_CHUNK - 1) / TOK_CHUNK * TOK_CHUNK;
    const size_t n_tiles = (size_t)(Tpad / TOK_CHUNK), rows = (size_t)ctx->n_aln + 2;
    if (n_tiles >= 0x7FFFFFFFull / WALK_CHUNK * 2) return fail(ctx, PAV_E_LIMIT, "pav_cigar_call: CIGAR text too large");
    if (reserve) {
        PAV_HIP(ctx, ctx->d_ops.reserve(sizeof(uint32_t) * (n_tiles * WALK_CHUNK + 16)));
        PAV_HIP(ctx, ctx->d_chunk.reserve(sizeof(uint64_t) * NT * (n_tiles + 1)));
        PAV_HIP(ctx, ctx->d_chunk2.reserve(2 * sizeof(uint32_t) * (n_tiles + 1)));
        PAV_HIP(ctx, ctx->d_op_off.reserve(2 * sizeof(uint64_t) * rows));
        PAV_HIP(ctx, ctx->d_rowbase.reserve((4 * sizeof(uint64_t) + 2 * sizeof(uint32_t)) * rows));
        PAV_HIP(ctx, ctx->d_totals.reserve(2 * sizeof(uint64_t)));
    }
    B.n_tiles = (uint32_t)n_tiles;
    B.ops = ctx->d_ops.as<uint32_t>();
    B.tile_pre = ctx->d_chunk.as<uint64_t>();
    B.tile_last = ctx->d_chunk2.as<uint32_t>(); B.chunk_row = B.tile_last + n_tiles + 1;
    B.op_off = ctx->d_op_off.as<uint64_t>(); B.op_slot = B.op_off + rows;
    B.rowbase = ctx->d_rowbase.as<uint64_t>(); B.row_local = B.rowbase + 2 * rows;
    B.row_tile = reinterpret_cast<uint32_t *>(B.row_local + 2 * rows); B.row_i = B.row_tile + rows;
    B.tok_err = ctx->d_totals.as<unsigned long long>(); B.err_op = B.tok_err + 1;
    return PAV_OK;
}
}  // namespace

int pav_cigar_call(pav_ctx *ctx, pav_cigar_counts *counts) {
    if (!ctx) return PAV_E_ARG;
    if (!ctx->cigar_loaded) return fail(ctx, PAV_E_STATE, "pav_cigar_call: pav_cigar_load has not been called");
    table_writer_quiesce(ctx);                                          // a device table write still reads the records this call rewrites
    PAV_HIP(ctx, hipSetDevice(ctx->device));
    { const int rch = wait_homology(ctx); if (rch != PAV_OK) return rch; }     // the scans of the last call: their records are rewritten
    memset(&ctx->counts, 0, sizeof ctx->counts);
    memset(&ctx->cigar_err, 0, sizeof ctx->cigar_err);
    ctx->cigar_called = false;
    const uint32_t n_aln = ctx->n_aln;
    CallBufs B;
    { const int rcb = call_bufs(ctx, B, true); if (rcb != PAV_OK) return rcb; }
    uint64_t *h_status = ctx->h_status;

    // --- tokenise + sum (one pass over the text), prefix over the tiles: two launches, one synchronisation ------------------
    PAV_HIP(ctx, hipMemsetAsync(B.tok_err, 0xFF, 2 * sizeof(uint64_t), ctx->stream));
    TokArgs TA;
    TA.text = ctx->d_text.as<uint8_t>(); TA.T = ctx->text_bytes; TA.text_off = ctx->d_text_off.as<uint64_t>();
    TA.n_aln = n_aln; TA.n_tiles = B.n_tiles;
    TA.ops = B.ops; TA.tile_agg = B.tile_pre; TA.tile_last = B.tile_last; TA.chunk_row = B.chunk_row;
    TA.op_slot = B.op_slot; TA.row_tile = B.row_tile; TA.row_i = B.row_i; TA.row_local = B.row_local;
    TA.tok_err = B.tok_err; TA.err_op = B.err_op;
    PAV_LAUNCH(ctx, "tok_tiles", tok_tiles, B.n_tiles, 256, 0, TA);
    ScanArgs SA;
    SA.tile_agg = B.tile_pre; SA.n_tiles = B.n_tiles; SA.n_aln = n_aln;
    SA.row_tile = B.row_tile; SA.row_i = B.row_i; SA.row_local = B.row_local; SA.op_off = B.op_off; SA.rowbase = B.rowbase;
    SA.aln = ctx->d_aln.as<pav_aln>(); SA.ref = ctx->seq[PAV_ROLE_REF].view(); SA.tig = ctx->seq[PAV_ROLE_TIG].view();
    SA.tok_err = B.tok_err; SA.err_op = B.err_op; SA.host_status = h_status;
    PAV_LAUNCH(ctx, "tile_scan", tile_scan, NT, 256, 0, SA);
    PAV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    uint64_t totals[NQ];
    for (int q = 0; q < NQ; ++q) totals[q] = h_status[q];
    const uint64_t n_ops = h_status[NQ];
    ctx->n_ops = n_ops;
    const uint64_t errs[2] = {h_status[NT], h_status[NT + 1]};
    const uint64_t range_row = h_status[NT + 2] < h_status[NT + 3] ? h_status[NT + 2] : h_status[NT + 3];
    // Any error is known by now (malformed token, illegal operation, a row that does not fit its records): nothing is emitted,
    // the kernels below would walk garbage or read past a record.
    const bool refuse = errs[0] != ~0ull || errs[1] != ~0ull || range_row != ~0ull;

    ctx->counts.n_ops = n_ops;
    ctx->counts.n_snv = totals[2];
    ctx->counts.n_indel = totals[3];
    ctx->counts.seq_bytes = totals[4];
    ctx->counts.aligned_bases = totals[5];

    // --- emit + homology (+ SEQ column) ---------------------------------------------------------------------
    const char *stage = getenv("PAV_CIGAR_STAGE");      // debugging: "scan" stops behind the tokenizer / prefix kernels
    if (n_ops && !refuse && !(stage && !strcmp(stage, "scan"))) {
        PAV_HIP(ctx, ctx->d_snv.reserve(sizeof(pav_snv) * (totals[2] + 1)));
        PAV_HIP(ctx, ctx->d_indel.reserve(sizeof(pav_indel) * (totals[3] + 1)));
        PAV_HIP(ctx, ctx->d_seqblob.reserve(round_up(totals[4] + 16, 16)));
        WalkArgs A;
        A.ops = B.ops; A.op_slot = B.op_slot; A.op_off = B.op_off; A.aln = ctx->d_aln.as<pav_aln>(); A.n_aln = n_aln;
        A.n_tiles = B.n_tiles; A.tile_pre = B.tile_pre; A.rowbase = B.rowbase; A.chunk_row = B.chunk_row; A.tile_last = B.tile_last;
        A.ref = SA.ref; A.tig = SA.tig;
        A.snv = ctx->d_snv.as<pav_snv>(); A.indel = ctx->d_indel.as<pav_indel>();
        // The stubs first; then the homology scans (+ SEQ) on the side stream and the SNV rows on this one, next to each other
        // (13 M isolated line fetches; the scans draw on the same budget).  Everything queued on this stream after the call is
        // ordered behind the scans (wait_homology below): they left-shift an insertion through its homology and write POS / END /
        // QRY_POS / QRY_END of every INS / DEL row - the stubs of walk_indel carry record offsets in those fields - so the
        // flagging, which reads POS / END, must not start before them.  (Letting it start early was tried: the race showed as
        // one failure in six runs of the eight-lane CHM13 test.)
        if (totals[3]) PAV_LAUNCH(ctx, "walk_indel", walk_emit<WALK_INDEL>, B.n_tiles, 256, 0, A);
        const bool skip_snv = stage && !strcmp(stage, "hom");     // debugging: the homology scans with nothing beside them
        if (totals[3] && !(stage && !strcmp(stage, "indel"))) {
            { int rcw = wait_planes(ctx); if (rcw != PAV_OK) return rcw; }   // the packed planes may still be in flight
            PAV_HIP(ctx, hipEventRecord(ctx->snv_ready, ctx->stream));       // (the stubs are written, the planes packed)
            PAV_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ctx->snv_ready, 0));
            // one wave per workgroup: the wave lifetimes are heavy-tailed (one long tandem repeat keeps a wave for tens of
            // microseconds), and a 256-lane workgroup holds its CU slot until the slowest of its four waves is done (0.34 -> 0.22 ms)
            PAV_LAUNCH_ON(ctx, ctx->stream2, "homology_kernel", homology_kernel, (uint32_t)((totals[3] + 63) / 64), 64, 0,
                          ctx->d_indel.as<pav_indel>(), totals[3], ctx->d_aln.as<pav_aln>(), A.ref, A.tig, ctx->d_seqblob.as<uint8_t>());
            PAV_HIP(ctx, hipEventRecord(ctx->hom_done, ctx->stream2));
            ctx->hom_pending = true;
        }
        if (totals[2] && !skip_snv) PAV_LAUNCH(ctx, "walk_snv", walk_emit<WALK_SNV>, B.n_tiles, 256, 0, A);
        { const int rch = wait_homology(ctx); if (rch != PAV_OK) return rch; }       // later readers of the INS / DEL rows use this stream
    }
    if (counts) *counts = ctx->counts;
    ctx->cigar_called = true;

    // --- errors: the first one the sequential walk would have hit ------------------------------------------
    if (refuse) {
        // Resolve on the host (error path only): fetch op offsets and, for an illegal op, the row's ops.
        std::vector<uint64_t> op_off((size_t)n_aln + 1), text_off((size_t)n_aln + 1);
        PAV_HIP(ctx, hipMemcpy(op_off.data(), B.op_off, sizeof(uint64_t) * op_off.size(), hipMemcpyDeviceToHost));
        PAV_HIP(ctx, hipMemcpy(text_off.data(), ctx->d_text_off.p, sizeof(uint64_t) * text_off.size(), hipMemcpyDeviceToHost));
        auto row_of = [&](const std::vector<uint64_t> &off, uint64_t x) {
            uint32_t lo = 0, hi = n_aln;
            while (hi - lo > 1) { uint32_t mid = (lo + hi) / 2; if (off[mid] <= x) lo = mid; else hi = mid; }
            while (lo + 1 < n_aln && off[lo + 1] <= x) ++lo;
            return lo;
        };
        pav_cigar_err tok{}, ill{};
        uint64_t tok_ord = ~0ull, ill_ord = ~0ull;     // (row << 32 | ordinal-in-row) walk order keys
        if (errs[0] != ~0ull) {
            const uint64_t bytepos = errs[0] >> 3;
            tok.kind = (int32_t)(errs[0] & 7);
            tok.aln = row_of(text_off, bytepos);
            tok.op_index = (uint32_t)(bytepos - text_off[tok.aln]);
            std::vector<uint8_t> rowtext((size_t)(text_off[tok.aln + 1] - text_off[tok.aln]));
            if (!rowtext.empty())
                PAV_HIP(ctx, hipMemcpy(rowtext.data(), ctx->d_text.as<uint8_t>() + text_off[tok.aln], rowtext.size(),
                                       hipMemcpyDeviceToHost));
            tok.op_char = tok.op_index < rowtext.size() ? rowtext[tok.op_index] : 0;
            uint32_t before = 0;
            for (uint32_t i = 0; i < tok.op_index && i < rowtext.size(); ++i) before += !(rowtext[i] >= '0' && rowtext[i] <= '9');
            tok_ord = ((uint64_t)tok.aln << 32) | (uint64_t)(before + 1);
        }
        if (errs[1] != ~0ull) {
            // the illegal operation is known by its slot: real ordinal = operations before its tile + its index in the tile
            const uint64_t tile = errs[1] / WALK_CHUNK;
            uint64_t pre_n = 0;
            PAV_HIP(ctx, hipMemcpy(&pre_n, B.tile_pre + (size_t)NQ * (B.n_tiles + 1) + tile, sizeof pre_n, hipMemcpyDeviceToHost));
            const uint64_t k = pre_n + errs[1] % WALK_CHUNK;
            ill.aln = row_of(op_off, k);
            const uint64_t first = op_off[ill.aln];
            std::vector<uint32_t> all((size_t)n_ops);
            std::vector<uint64_t> dummy((size_t)n_aln + 1);
            { const int rcf = pav_cigar_fetch_ops_unchecked(ctx, all.data(), dummy.data()); if (rcf != PAV_OK) return rcf; }
            std::vector<uint32_t> rops(all.begin() + (ptrdiff_t)first, all.begin() + (ptrdiff_t)k + 1);
            std::vector<pav_aln> al(1);
            PAV_HIP(ctx, hipMemcpy(al.data(), ctx->d_aln.as<pav_aln>() + ill.aln, sizeof(pav_aln), hipMemcpyDeviceToHost));
            uint64_t pr = al[0].pos, pt = 0;
            for (size_t i = 0; i + 1 < rops.size(); ++i) {
                const uint32_t c = rops[i] & 15u, l = rops[i] >> 4;
                if (c == 7 || c == 8 || c == 2) pr += l;
                if (c == 7 || c == 8 || c == 1 || c == 4 || c == 5) pt += l;
            }
            const uint32_t code = rops.back() & 15u;
            ill.kind = code == 0 ? PAV_CIGAR_ERR_M : PAV_CIGAR_ERR_OP;
            ill.op_index = (uint32_t)(k - first) + 1;
            ill.op_char = (uint32_t)"MIDNSHP=X???????"[code];
            ill.pos_ref = (uint32_t)pr; ill.pos_tig = (uint32_t)pt;
            ill_ord = ((uint64_t)ill.aln << 32) | (uint64_t)ill.op_index;
        }
        ctx->cigar_err = (ill_ord < tok_ord) ? ill : tok;
        // a row that does not fit its records is reported unless a token / operation error comes first in walk order (same
        // row included: the reference meets those while it still walks inside the sequence)
        if (range_row != ~0ull && (std::min(ill_ord, tok_ord) >> 32) > range_row) {
            pav_cigar_err rg{};
            rg.kind = PAV_CIGAR_ERR_RANGE;
            rg.aln = (uint32_t)range_row;
            ctx->cigar_err = rg;
        }
        ctx->cigar_called = false;
        return fail(ctx, PAV_E_CIGAR, "CIGAR error kind %d at alignment row %u", ctx->cigar_err.kind, ctx->cigar_err.aln);
    }
    return PAV_OK;
}

// The operations live in padded tiles on the device (tok_tiles); callers get them contiguous, at their real ordinals.
static int pav_cigar_fetch_ops_unchecked(pav_ctx *ctx, uint32_t *ops, uint64_t *op_off) {
    PAV_HIP(ctx, hipSetDevice(ctx->device));
    CallBufs B;
    { const int rcb = call_bufs(ctx, B, false); if (rcb != PAV_OK) return rcb; }
    if (ops && ctx->n_ops) {
        PAV_HIP(ctx, ctx->d_tmp.reserve(sizeof(uint32_t) * (ctx->n_ops + 16)));
        PAV_LAUNCH(ctx, "ops_compact", ops_compact, B.n_tiles, 256, 0, B.ops, B.tile_pre + (size_t)NQ * (B.n_tiles + 1), ctx->d_tmp.as<uint32_t>());
        PAV_HIP(ctx, hipMemcpyAsync(ops, ctx->d_tmp.p, sizeof(uint32_t) * ctx->n_ops, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (op_off)
        PAV_HIP(ctx, hipMemcpyAsync(op_off, B.op_off, sizeof(uint64_t) * ((size_t)ctx->n_aln + 1), hipMemcpyDeviceToHost, ctx->stream));
    PAV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PAV_OK;
}

int pav_cigar_error(const pav_ctx *ctx, pav_cigar_err *err) {
    if (!ctx || !err) return PAV_E_ARG;
    *err = ctx->cigar_err;
    return PAV_OK;
}

int pav_cigar_fetch(pav_ctx *ctx, pav_snv *snv, pav_indel *indel, uint8_t *seq_blob) {
    if (!ctx) return PAV_E_ARG;
    if (!ctx->cigar_called) return fail(ctx, PAV_E_STATE, "pav_cigar_fetch: no successful pav_cigar_call to fetch from");
    PAV_HIP(ctx, hipSetDevice(ctx->device));
    { const int rch = wait_homology(ctx); if (rch != PAV_OK) return rch; }
    if (snv && ctx->counts.n_snv)
        PAV_HIP(ctx, hipMemcpyAsync(snv, ctx->d_snv.p, sizeof(pav_snv) * ctx->counts.n_snv, hipMemcpyDeviceToHost, ctx->stream));
    if (indel && ctx->counts.n_indel)
        PAV_HIP(ctx, hipMemcpyAsync(indel, ctx->d_indel.p, sizeof(pav_indel) * ctx->counts.n_indel, hipMemcpyDeviceToHost, ctx->stream));
    if (seq_blob && ctx->counts.seq_bytes)
        PAV_HIP(ctx, hipMemcpyAsync(seq_blob, ctx->d_seqblob.p, ctx->counts.seq_bytes, hipMemcpyDeviceToHost, ctx->stream));
    PAV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PAV_OK;
}

int pav_cigar_verify(pav_ctx *ctx, pav_verify_counts *out) {
    if (!ctx || !out) return PAV_E_ARG;
    if (!ctx->cigar_called) return fail(ctx, PAV_E_STATE, "pav_cigar_verify: pav_cigar_call has not been called");
    PAV_HIP(ctx, hipSetDevice(ctx->device));
    memset(out, 0, sizeof *out);
    out->first_bad_op = ~0ull;
    if (!ctx->n_ops) return PAV_OK;
    CallBufs B;
    { const int rcb = call_bufs(ctx, B, false); if (rcb != PAV_OK) return rcb; }
    PAV_HIP(ctx, ctx->d_tmp.reserve(64));
    unsigned long long *d_cnt = ctx->d_tmp.as<unsigned long long>();
    PAV_HIP(ctx, hipMemsetAsync(d_cnt, 0, 32, ctx->stream));
    PAV_HIP(ctx, hipMemsetAsync(d_cnt + 4, 0xFF, 8, ctx->stream));
    { int rcw = need_planes_full(ctx, PAV_ROLE_TIG); if (rcw != PAV_OK) return rcw; }   // verify streams both arenas: the contig planes are packed here if they are not
    VerifyArgs A;
    A.ops = B.ops; A.op_slot = B.op_slot; A.aln = ctx->d_aln.as<pav_aln>(); A.n_aln = ctx->n_aln; A.n_tiles = B.n_tiles;
    A.tile_pre = B.tile_pre; A.rowbase = B.rowbase; A.chunk_row = B.chunk_row;
    A.ref = ctx->seq[PAV_ROLE_REF].view(); A.tig = ctx->seq[PAV_ROLE_TIG].view();
    A.cnt = d_cnt;
    // 32-bit positions when both arenas (padding included) stay below 2^32 bases; PAV_VERIFY_WIDE=1 forces the 64-bit kernel (tests)
    const char *force_wide = getenv("PAV_VERIFY_WIDE");
    const bool wide = (force_wide && *force_wide == '1') || ctx->seq[PAV_ROLE_REF].arena + 128 >= (1ull << 32) ||
                      ctx->seq[PAV_ROLE_TIG].arena + 128 >= (1ull << 32);
    if (wide) PAV_LAUNCH(ctx, "verify_kernel", verify_kernel<true>, VSPLIT * B.n_tiles, 256, 0, A);
    else      PAV_LAUNCH(ctx, "verify_kernel", verify_kernel<false>, VSPLIT * B.n_tiles, 256, 0, A);
    unsigned long long h[5];
    PAV_HIP(ctx, hipMemcpyAsync(h, d_cnt, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    PAV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    out->eq_bases = h[0]; out->eq_mismatch = h[1]; out->x_bases = h[2]; out->x_match = h[3]; out->first_bad_op = h[4];
    if (prof_flush(ctx) != PAV_OK) return PAV_E_HIP;
    return PAV_OK;
}

int pav_cigar_fetch_ops(pav_ctx *ctx, uint32_t *ops, uint64_t *op_off) {
    if (!ctx) return PAV_E_ARG;
    if (!ctx->cigar_loaded) return fail(ctx, PAV_E_STATE, "pav_cigar_fetch_ops: nothing loaded");
    return pav_cigar_fetch_ops_unchecked(ctx, ops, op_off);
}

int pav_align_index(pav_ctx *ctx, uint32_t n_aln, const uint32_t *row_pos, const uint8_t *cigar_text, const uint64_t *cigar_off,
                    uint64_t *n_ops_out, uint32_t *ops, uint64_t *op_off, uint32_t *sub_begin, uint32_t *qry_begin) {
    if (!ctx || !n_ops_out) return PAV_E_ARG;
    if (n_aln && (!row_pos || !cigar_off)) return fail(ctx, PAV_E_ARG, "pav_align_index: null input");
    PAV_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint64_t T = n_aln ? cigar_off[n_aln] : 0;
    const bool upload = ops == nullptr;                 // first call (ops == NULL): tokenise and return the op count
    const uint64_t Tpad = round_up(T + 1, TOK_CHUNK);
    const uint32_t n_tchunks = (uint32_t)(Tpad / TOK_CHUNK);
    if (upload) {
        memset(&ctx->cigar_err, 0, sizeof ctx->cigar_err);
        PAV_HIP(ctx, ctx->ix_text.reserve(Tpad));
        PAV_HIP(ctx, ctx->ix_off.reserve(sizeof(uint64_t) * ((size_t)n_aln + 1)));
        PAV_HIP(ctx, ctx->ix_pos.reserve(sizeof(uint32_t) * ((size_t)n_aln + 1)));
        PAV_HIP(ctx, ctx->ix_err.reserve(16));
        PAV_HIP(ctx, hipMemsetAsync(ctx->ix_text.p, '0', Tpad, st));
        PAV_HIP(ctx, hipMemsetAsync(ctx->ix_err.p, 0xFF, 16, st));
        if (T) PAV_HIP(ctx, hipMemcpyAsync(ctx->ix_text.p, cigar_text, T, hipMemcpyHostToDevice, st));
        uint64_t zero = 0;
        PAV_HIP(ctx, hipMemcpyAsync(ctx->ix_off.p, n_aln ? cigar_off : &zero, sizeof(uint64_t) * ((size_t)n_aln + 1), hipMemcpyHostToDevice, st));
        if (n_aln) PAV_HIP(ctx, hipMemcpyAsync(ctx->ix_pos.p, row_pos, sizeof(uint32_t) * n_aln, hipMemcpyHostToDevice, st));
        PAV_HIP(ctx, ctx->ix_chunk.reserve(sizeof(uint32_t) * n_tchunks + sizeof(uint64_t) * ((size_t)n_tchunks + 1) + 64));
        uint32_t *d_tcnt = ctx->ix_chunk.as<uint32_t>();
        uint64_t *d_tpre = reinterpret_cast<uint64_t *>(ctx->ix_chunk.as<uint8_t>() + round_up(sizeof(uint32_t) * n_tchunks, 16));
        unsigned long long *d_tok_err = ctx->ix_err.as<unsigned long long>();
        PAV_LAUNCH(ctx, "tok_count", tok_count, n_tchunks, 256, 0, ctx->ix_text.as<uint4>(), d_tcnt);
        PAV_LAUNCH(ctx, "scan_counts", scan_counts, 1, 256, 0, d_tcnt, d_tpre, n_tchunks, (volatile uint64_t *)nullptr);
        uint64_t n_ops = 0;
        PAV_HIP(ctx, hipMemcpyAsync(&n_ops, d_tpre + n_tchunks, sizeof n_ops, hipMemcpyDeviceToHost, st));
        PAV_HIP(ctx, hipStreamSynchronize(st));
        PAV_HIP(ctx, ctx->ix_ops.reserve(sizeof(uint32_t) * (n_ops + 16)));
        PAV_HIP(ctx, ctx->ix_op_off.reserve(sizeof(uint64_t) * ((size_t)n_aln + 1)));
        PAV_LAUNCH(ctx, "tok_emit", tok_emit, n_tchunks, 256, 0, ctx->ix_text.as<uint8_t>(), d_tpre, ctx->ix_ops.as<uint32_t>(), d_tok_err);
        PAV_LAUNCH(ctx, "row_ops", row_ops, (n_aln + 1 + 3) / 4, 256, 0, ctx->ix_text.as<uint8_t>(), ctx->ix_off.as<uint64_t>(), d_tpre,
                   ctx->ix_op_off.as<uint64_t>(), n_aln, d_tok_err);
        const uint32_t n_wchunks = (uint32_t)((n_ops + WALK_CHUNK - 1) / WALK_CHUNK);
        PAV_HIP(ctx, ctx->ix_begin.reserve(2 * sizeof(uint32_t) * (n_ops + 16)));
        if (n_wchunks) {
            PAV_HIP(ctx, ctx->ix_chunk2.reserve(4 * sizeof(uint64_t) * ((size_t)n_wchunks + 1)));
            uint64_t *d_csum = ctx->ix_chunk2.as<uint64_t>(), *d_cpre = d_csum + 2ull * (n_wchunks + 1);
            PAV_HIP(ctx, ctx->ix_rowbase.reserve(2 * sizeof(uint64_t) * ((size_t)n_aln + 1)));
            PAV_LAUNCH(ctx, "lift_reduce", lift_reduce, n_wchunks, 256, 0, ctx->ix_ops.as<uint32_t>(), n_ops, d_csum);
            PAV_LAUNCH(ctx, "lift_chunks", lift_chunks, 1, 256, 0, d_csum, d_cpre, n_wchunks);
            PAV_LAUNCH(ctx, "lift_row_base", lift_row_base, (n_aln + 3) / 4, 256, 0, ctx->ix_ops.as<uint32_t>(), ctx->ix_op_off.as<uint64_t>(),
                       d_cpre, ctx->ix_rowbase.as<uint64_t>(), n_aln);
            PAV_LAUNCH(ctx, "lift_positions", lift_positions, n_wchunks, 256, 0, ctx->ix_ops.as<uint32_t>(), n_ops, ctx->ix_op_off.as<uint64_t>(),
                       ctx->ix_pos.as<uint32_t>(), n_aln, d_cpre, ctx->ix_rowbase.as<uint64_t>(), ctx->ix_begin.as<uint32_t>(),
                       ctx->ix_begin.as<uint32_t>() + n_ops);
        }
        uint64_t terr = ~0ull;
        PAV_HIP(ctx, hipMemcpyAsync(&terr, ctx->ix_err.p, sizeof terr, hipMemcpyDeviceToHost, st));
        PAV_HIP(ctx, hipStreamSynchronize(st));
        *n_ops_out = n_ops;
        ctx->ix_n_ops = n_ops; ctx->ix_n_aln = n_aln;
        if (terr != ~0ull) {
            const uint64_t bytepos = terr >> 3;
            uint32_t lo = 0, hi = n_aln;
            while (hi - lo > 1) { uint32_t mid = (lo + hi) / 2; if (cigar_off[mid] <= bytepos) lo = mid; else hi = mid; }
            while (lo + 1 < n_aln && cigar_off[lo + 1] <= bytepos) ++lo;
            ctx->cigar_err.kind = (int32_t)(terr & 7); ctx->cigar_err.aln = lo;
            ctx->cigar_err.op_index = (uint32_t)(bytepos - cigar_off[lo]);
            ctx->cigar_err.op_char = cigar_text[bytepos];
            return fail(ctx, PAV_E_CIGAR, "CIGAR tokenizer error kind %d at alignment row %u", ctx->cigar_err.kind, lo);
        }
        return PAV_OK;
    }
    // second call: copy the tables out
    if (ctx->ix_n_aln != n_aln) return fail(ctx, PAV_E_STATE, "pav_align_index: fetch does not match the indexed table");
    const uint64_t n_ops = ctx->ix_n_ops;
    *n_ops_out = n_ops;
    if (n_ops) {
        PAV_HIP(ctx, hipMemcpyAsync(ops, ctx->ix_ops.p, sizeof(uint32_t) * n_ops, hipMemcpyDeviceToHost, st));
        if (sub_begin) PAV_HIP(ctx, hipMemcpyAsync(sub_begin, ctx->ix_begin.p, sizeof(uint32_t) * n_ops, hipMemcpyDeviceToHost, st));
        if (qry_begin) PAV_HIP(ctx, hipMemcpyAsync(qry_begin, ctx->ix_begin.as<uint32_t>() + n_ops, sizeof(uint32_t) * n_ops, hipMemcpyDeviceToHost, st));
    }
    if (op_off) PAV_HIP(ctx, hipMemcpyAsync(op_off, ctx->ix_op_off.p, sizeof(uint64_t) * ((size_t)n_aln + 1), hipMemcpyDeviceToHost, st));
    PAV_HIP(ctx, hipStreamSynchronize(st));
    return PAV_OK;
}

int pav_homology(pav_ctx *ctx, uint32_t n, const pav_hom_query *q, uint32_t *out) {
    if (!ctx || (n && (!q || !out))) return PAV_E_ARG;
    if (!n) return PAV_OK;
    PAV_HIP(ctx, hipSetDevice(ctx->device));
    for (uint32_t i = 0; i < n; ++i) {
        const pav_hom_query &h = q[i];
        if (h.role < 0 || h.role > 1 || h.sv_role < 0 || h.sv_role > 1 || h.seq_id < 0 || h.sv_seq_id < 0 ||
            (uint32_t)h.seq_id >= ctx->seq[h.role].n || (uint32_t)h.sv_seq_id >= ctx->seq[h.sv_role].n)
            return fail(ctx, PAV_E_ARG, "pav_homology: query %u references a sequence that is not loaded", i);
    }
    PAV_HIP(ctx, ctx->d_tmp.reserve(sizeof(pav_hom_query) * n + sizeof(uint32_t) * n + 64));
    pav_hom_query *dq = ctx->d_tmp.as<pav_hom_query>();
    uint32_t *dout = reinterpret_cast<uint32_t *>(dq + n);
    PAV_HIP(ctx, hipMemcpyAsync(dq, q, sizeof(pav_hom_query) * n, hipMemcpyHostToDevice, ctx->stream));
    { int rcw = wait_planes(ctx); if (rcw != PAV_OK) return rcw; }
    PAV_LAUNCH(ctx, "homology_query_kernel", homology_query_kernel, (n + 255) / 256, 256, 0, dq, n,
               ctx->seq[PAV_ROLE_REF].view(), ctx->seq[PAV_ROLE_TIG].view(), dout);
    PAV_HIP(ctx, hipMemcpyAsync(out, dout, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, ctx->stream));
    PAV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PAV_OK;
}

}  // extern "C"
