// gzip on the device (deflate.hip): the text of whole files, resident in HBM, becomes conforming gzip files in pinned host memory.
// The table writers (tables.hip: rule call_cigar / call_cigar_merge, rules/call.snakefile:845-846; invscan.cpp: the density tables
// of rule call_inv_batch, rules/call_inv.snakefile:279-291) format their rows on the device (textdev.hip) and hand the text here.
#pragma once

#include <cstdint>
#include <vector>

#include "common.h"

namespace pav {

constexpr uint64_t GZ_TEXT_ALIGN = 256;        // every file's text starts on such a boundary of the arena
constexpr uint32_t GZ_SEGMENT = 1u << 16;      // bytes of text per deflate block (one wave encodes one)
constexpr uint64_t GZ_TEXT_PAD = 4096;         // readable bytes the arena must have behind the last text byte

struct GzFile { uint64_t text_off, text_len; };

struct GzOut {                                 // where the files are when gz_files returns (valid until the next call on the context)
    const uint8_t *host = nullptr;             // pinned
    std::vector<uint64_t> off, len;            // per file: the whole gzip member, header and trailer included
};

// Queues everything on `st`, waits for it, fills `out`.  level 1..9 (0 = 6): how hard the match finder looks.  `slot`: the caller's
// scratch (device buffers, pinned memory; created on first use) - one per writer, writers run side by side.  Errors go to the
// calling thread's message (pav_last_error(NULL)).
int gz_files(pav_ctx *ctx, void **slot, hipStream_t st, const uint8_t *d_text, uint64_t text_alloc, const std::vector<GzFile> &files, int level,
             GzOut &out);
void gz_release_slot(pav_ctx *ctx, void **slot);
void gz_release(pav_ctx *ctx);                 // the context's own slot (pav_gzip_buffer)

// The writers run on threads of their own beside the context's caller (tables.hip, rules.call_haplotype): errors go to the
// thread's own message (pav_last_error(NULL)), launches are not event-profiled (the context's profile is its caller's).
#define W_HIP(call)                                                                                              \
    do {                                                                                                         \
        hipError_t e__ = (call);                                                                                 \
        if (e__ != hipSuccess) return pav::fail(nullptr, PAV_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)
#define W_LAUNCH(st, kernel, grid, block, shmem, ...)                                                            \
    do {                                                                                                         \
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), (shmem), (st), __VA_ARGS__);                         \
        W_HIP(hipGetLastError());                                                                                \
        if (pav::sync_each()) { fprintf(stderr, "[pav launch] %s grid %u\n", #kernel, (unsigned)(grid)); W_HIP(hipStreamSynchronize(st)); } \
    } while (0)

}  // namespace pav
