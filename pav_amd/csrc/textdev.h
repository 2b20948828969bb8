// Table text on the device (textdev.hip): the rows of the SNV / INS-DEL tables of rule call_cigar(_merge) and of the density
// tables of rule call_inv_batch become TSV bytes in HBM - what DataFrame.to_csv(sep='\t', index=False) writes
// (rules/call.snakefile:845-846, rules/call_inv.snakefile:287-291) - and go to the device gzip (devgz.h) from there.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "common.h"

namespace pav {

// ---- the two tables of rule call_cigar / call_cigar_merge -------------------------------------------------------------------
// Everything the writer needs besides the records resident in the context (d_snv, d_indel, d_seqblob, d_aln): owned by the job,
// so the write may run on a thread and a stream of its own while the context goes on (tables.hip).
struct CigarTextJob {
    std::vector<std::string> rnames, tnames;           // record names of the two stores (no character that to_csv would quote)
    std::vector<uint16_t> rank;                        // [n_ref] position of the name in byte order (#CHROM sorts as str)
    std::vector<uint8_t> batch8;                       // [n_aln] CALL_BATCH, empty: the tables of one job
    std::vector<int64_t> align_index, trim_pos, trim_end;   // [n_aln]; trim_*: empty = no FILTER column
    std::string hap, snv_path, insdel_path;
    bool have_snv = false, have_insdel = false, filter = false;   // filter: trim_* given, the tables get a FILTER column
    int level = 6;
    uint64_t n_snv = 0, n_ind = 0;
    hipEvent_t ready = nullptr;                        // recorded on the context's stream behind the call (and the homology scans)
};
// Runs the whole write on the writer's own stream: order + FILTER, row text, gzip (names ending in ".gz") or plain text, the files.
int text_cigar_tables(pav_ctx *ctx, CigarTextJob &job, std::string &err);

// ---- the density tables of rule call_inv_batch ----------------------------------------------------------------------------
struct DenTableDev {                                   // one call's table, columns resident in HBM (a CallStage block)
    const double *k0, *k1, *k2;                        // k1 == nullptr: the call has no FWDREV k-mers, the column is 0.0
    const unsigned long long *kmer; const uint32_t *index; const int8_t *state_mer, *state; const uint8_t *flank, *match;
    uint32_t n;
};
int text_density_tables(pav_ctx *ctx, const std::vector<DenTableDev> &tables, const std::vector<std::string> &paths, int level);

bool device_writer_enabled();                          // PAV_WRITER=host switches the writers back to host threads + zlib
void textdev_release(pav_ctx *ctx);

}  // namespace pav
