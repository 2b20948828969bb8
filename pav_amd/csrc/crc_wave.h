// CRC-32 of a byte range by ONE wave (gzip members: rules/call.snakefile:845-846 writes them, pavlib/cigarcall.py:59-64 reads BGZF).
// A lane takes 64 bytes of every 4 KiB tile - four table chains of 16 bytes, so a wave's loads cover whole lines and every line is
// fetched once; the chains of a tile are joined by x^(8 16), a lane's tiles by x^(8 4096), the lanes by x^(8 64) doubling its
// exponent.  The text is counted from its END: zeros in front of it change nothing in a register run from zero, so the first tile is
// the ragged one.  Returns the register run from ZERO (every lane holds it); the checksum of the range is
//     wave_crc_raw(...) ^ gf_mul(0xFFFFFFFF, x^(8 n)) ^ 0xFFFFFFFF.
// Used by k_bgzf_crc (inflate.hip: members checked against their footers) and k_crc_segments (deflate.hip: members written).
#pragma once

#include "deflate_dev.h"

namespace pav {

struct CrcPowers { uint32_t x16, x64, x4096; };
inline CrcPowers crc_powers() { static const CrcPowers P{dfl::gf_xpow8(16), dfl::gf_xpow8(64), dfl::gf_xpow8(4096)}; return P; }

// tab: the 256-entry table in LDS (filled and synchronised by the caller); p .. p + n: the bytes (n > 0)
__device__ __forceinline__ uint32_t wave_crc_raw(const uint32_t *tab, const uint8_t *__restrict__ p, uint32_t n, const CrcPowers X) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t tiles = (n + 4095u) / 4096u, pad = tiles * 4096u - n;       // `pad` zero bytes in front
    uint32_t acc = 0;
    for (uint32_t t = 0; t < tiles; ++t) {
        const uint32_t base = t * 4096u + lane * 64u;                          // padded position of the lane's 64 bytes
        uint32_t r[4] = {0, 0, 0, 0};
        if (base >= pad && (((size_t)(p + (base - pad))) & 3u) == 0) {        // the common tile: all 64 bytes are text, words can be loaded
            const uint32_t *w = reinterpret_cast<const uint32_t *>(p + (base - pad));
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) {
                uint32_t v[4];
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) v[j] = w[j * 4 + i];
#pragma unroll
                for (uint32_t b8 = 0; b8 < 4; ++b8)
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) { r[j] = tab[(r[j] ^ v[j]) & 0xFFu] ^ (r[j] >> 8); v[j] >>= 8; }
            }
        } else if (base + 64u > pad) {
            for (uint32_t i = 0; i < 16; ++i)
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) { const uint32_t q = base + j * 16u + i; const uint32_t by = q >= pad ? p[q - pad] : 0u; r[j] = tab[(r[j] ^ by) & 0xFFu] ^ (r[j] >> 8); }
        }
        const uint32_t tile = dfl::gf_mul(X.x16, dfl::gf_mul(X.x16, dfl::gf_mul(X.x16, r[0]) ^ r[1]) ^ r[2]) ^ r[3];
        acc = dfl::gf_mul(X.x4096, acc) ^ tile;
    }
    uint32_t f = X.x64, v = acc;
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, (int)d);
        v = (lane & d) ? dfl::gf_mul(f, o) ^ v : dfl::gf_mul(f, v) ^ o;
        f = dfl::gf_mul(f, f);
    }
    return v;
}

}  // namespace pav
