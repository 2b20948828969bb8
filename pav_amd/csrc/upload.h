// Host-to-device plumbing shared by the sequence loaders (ctx.hip: pav_seq_load; fastadev.hip: pav_seq_load_fasta_path).
#pragma once

#include <functional>
#include <vector>

#include "common.h"

namespace pav {

// Large uploads from pageable host memory (a FASTA file's records: 3 GB per store).  hipMemcpyAsync from pageable memory is staged
// by the runtime on one thread (10 - 13 GB/s measured: 0.2 s per store, most of the "sequences" stage of a haplotype); here the
// bytes go through a ring of pinned slots filled by several threads while the slots before them cross PCIe.
struct UploadRing {
    static constexpr int SLOTS = 4;
    static constexpr size_t SLOT_BYTES = 32u << 20;
    void *slot[SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev[SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    bool busy[SLOTS] = {false, false, false, false};
    int next = 0;
    bool ok = false;
};

UploadRing *upload_ring(pav_ctx *ctx, int which = 0);       // ring 0 / 1 of the context (created on first use; ok == false: no pinned memory)
void upload_release(pav_ctx *ctx);
void fastadev_release(pav_ctx *ctx);                        // fastadev.hip
// dst[0, bytes) on the device <- src (pageable host memory), queued on `st`; returns when every byte has left `src`
int staged_upload(pav_ctx *ctx, hipStream_t st, uint8_t *dst, const uint8_t *src, uint64_t bytes, int which = 0);
int seq_store_load(pav_ctx *ctx, int role, uint32_t n_seq, const uint64_t *len, const char *what,
                   const std::function<int(uint8_t *, const std::vector<uint64_t> &)> &fill);

}  // namespace pav
