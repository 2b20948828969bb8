// BGZF text inflated on the device (inflate.hip): the members of a bgzipped file - PAV's FASTA files, `contigs_{hap}.fa.gz` and
// `data/ref/ref.fa.gz` (rules/call.snakefile:796, pavlib/cigarcall.py:59-64) - cross PCIe compressed and become text in HBM.
#pragma once

#include <cstdint>
#include <vector>

#include "common.h"

namespace pav {

// The members of a file as the host's walk over the headers found them: the deflate payload of each, [in_off, in_off + in_len)
// (the CRC-32 and ISIZE of the member are the eight bytes behind it).
struct BgzfMembers {
    std::vector<uint64_t> in_off;
    std::vector<uint32_t> in_len;
};

// d_comp: the file's bytes in HBM (readable 16 bytes beyond the last member).  The text of all members, one behind the other, is left
// in `out` (grown as needed, readable 4 KiB beyond), its length in *n_text; every member's CRC-32 and ISIZE are checked.  `state`:
// scratch that lives between calls (inflate_release frees it).  Runs on `st`, returns with the stream drained.
int bgzf_inflate_device(pav_ctx *ctx, hipStream_t st, void **state, const uint8_t *d_comp, const BgzfMembers &M, DevBuf &out, uint64_t *n_text,
                        const char *what);
void inflate_release(void **state);

// Large device scratch that outlives a context (a process-wide list per GPU): the file's bytes, its text and the token lists of a
// 3 GB FASTA are 10 GB a role - a haplotype per context, allocated and freed each time, had the driver clear tens of GB of HBM per
// haplotype, and every few contexts an allocation waited seconds for it.  scratch_take gives a buffer of at least `bytes` (one from
// the list when it fits without wasting more than it holds, a new one otherwise); scratch_give returns it (at most 48 GB are kept per
// GPU; the rest is freed).  The buffer must be idle: no kernel or copy that uses it still queued.
hipError_t scratch_take(int device, size_t bytes, DevBuf &b);
void scratch_give(int device, DevBuf &b);

}  // namespace pav
