// Device side of the alignment trimming (trim_dev.hip) as trim.cpp sees it.
#pragma once
#include "common.h"

namespace pav {

// Current CIGAR of one record: pre + ops[a, b) (first / last length overridden) + post; trimming adds at most two clipping
// operations (H, S) at either end.
struct CigDev {
    unsigned long long a, b, len_first, len_last;
    unsigned long long pre_len[2], post_len[2];
    uint8_t pre_code[2], post_code[2];
    uint8_t n_pre, n_post, modified, pad;
};
struct RowDev { pav_trim_row f; CigDev c; };

struct TrimFailDev {
    int32_t kind; uint32_t op_index; unsigned long long op_len; uint32_t op_code; int32_t side; long long diff_bp;
    uint32_t row_l, row_r;
};

struct TrimPassArgs {
    const uint32_t *ops;                 // tokenised operations of the loaded table (device copy)
    RowDev *rows;                        // every loaded row
    const uint32_t *order;               // the pass's iteration order (row numbers), groups one after the other
    const uint32_t *group_off;           // [n_groups + 1] first position of every group in `order`
    uint8_t *scratch; const unsigned long long *scratch_off, *scratch_cap;   // per group: two traces of scratch_cap entries (56 B each)
    long long min_len; int32_t mode, match_tig;
    unsigned long long *err_key;         // smallest (il << 32 | ir) of a failing pair, ~0 when none
    TrimFailDev *err_slots; unsigned long long *err_slot_key;   // [n_groups] the failure of a group and its key
};

int trim_launch_pass(pav_ctx *ctx, const TrimPassArgs &A, uint32_t n_groups);
int trim_launch_pair(pav_ctx *ctx, const TrimPassArgs &A, int rev_l, int rev_r);

}  // namespace pav
