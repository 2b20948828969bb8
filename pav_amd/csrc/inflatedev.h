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

// The loaders' large scratch - the file's bytes, its text, the token lists: 10 GB a role - is taken for a load and given back after it
// (the process-wide block list of common.h keeps it for the next load, the other role's or the next haplotype's).  The buffer must be
// idle when it is given back.
hipError_t scratch_take(int device, size_t bytes, DevBuf &b);
void scratch_give(int device, DevBuf &b);

}  // namespace pav
