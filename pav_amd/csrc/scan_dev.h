// Exclusive prefix sum of u32 values into u64 offsets on the device (three small kernels, tiles of 2048): the row offsets of the
// table text (textdev.hip) and the line-break counts of a FASTA file's raw text (fastadev.hip).  Header-only: the kernels are
// static, every translation unit that includes this gets its own copy.
#pragma once

#include <algorithm>

#include "devgz.h"      // W_HIP / W_LAUNCH

namespace pav {

// ---- exclusive prefix sum of the row lengths (u32 -> u64), tiles of 2048 --------------------------------------------------
constexpr uint32_t SCAN_TILE = 2048;
static __device__ __forceinline__ uint64_t block_exclusive(uint64_t v, uint64_t *sh /* 8 */, uint64_t &block_total) {   // 256 lanes
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint64_t incl = v;
    for (int d = 1; d < 64; d <<= 1) { const uint64_t o = (uint64_t)__shfl_up((long long)incl, d); if ((int)lane >= d) incl += o; }
    if (lane == 63) sh[wave] = incl;
    __syncthreads();
    uint64_t base = 0;
    for (uint32_t w = 0; w < wave; ++w) base += sh[w];
    block_total = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return base + incl - v;
}
static __global__ __launch_bounds__(256) void k_scan_sum(const uint32_t *__restrict__ len, uint64_t n, uint64_t *__restrict__ bsum) {
    __shared__ uint64_t sh[8];
    const uint64_t t0 = (uint64_t)blockIdx.x * SCAN_TILE;
    uint64_t v = 0;
    for (uint32_t j = 0; j < SCAN_TILE / 256; ++j) { const uint64_t i = t0 + j * 256 + threadIdx.x; if (i < n) v += len[i]; }
    uint64_t total;
    (void)block_exclusive(v, sh, total);
    if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}
static __global__ __launch_bounds__(256) void k_scan_top(uint64_t *bsum, uint32_t nb) {        // one block; bsum[nb] = everything
    __shared__ uint64_t sh[8];
    uint64_t carry = 0;
    for (uint32_t b0 = 0; b0 < nb; b0 += 256) {
        const uint32_t b = b0 + threadIdx.x;
        const uint64_t v = b < nb ? bsum[b] : 0;
        uint64_t total;
        const uint64_t ex = block_exclusive(v, sh, total);
        if (b < nb) bsum[b] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) bsum[nb] = carry;
}
static __global__ __launch_bounds__(256) void k_scan_apply(const uint32_t *__restrict__ len, uint64_t n, const uint64_t *__restrict__ bsum, uint32_t nb,
                                                    uint64_t *__restrict__ roff) {
    __shared__ uint64_t sh[8];
    const uint64_t i0 = (uint64_t)blockIdx.x * SCAN_TILE + threadIdx.x * 8u;
    uint32_t x[8]; uint64_t v = 0;
    for (uint32_t j = 0; j < 8; ++j) { x[j] = i0 + j < n ? len[i0 + j] : 0u; v += x[j]; }
    uint64_t total;
    uint64_t at = bsum[blockIdx.x] + block_exclusive(v, sh, total);
    for (uint32_t j = 0; j < 8; ++j) { if (i0 + j < n) roff[i0 + j] = at; at += x[j]; }
    if (blockIdx.x == 0 && threadIdx.x == 0) roff[n] = bsum[nb];
}

// off[0 .. n] from val[0 .. n) on `st`; bsum: scratch of (n / 2048 + 2) u64
static inline int scan_u32_to_u64(hipStream_t st, const uint32_t *val, uint64_t n, uint64_t *bsum, uint64_t *off) {
    const uint32_t nb = (uint32_t)((n + SCAN_TILE - 1) / SCAN_TILE);
    if (nb) W_LAUNCH(st, k_scan_sum, nb, 256, 0, val, n, bsum);
    W_LAUNCH(st, k_scan_top, 1, 256, 0, bsum, nb);
    W_LAUNCH(st, k_scan_apply, std::max(1u, nb), 256, 0, val, n, bsum, nb, off);
    return PAV_OK;
}

}  // namespace pav
