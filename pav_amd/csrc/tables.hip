// Native writer for the two tables rule call_cigar produces (rules/call.snakefile:813-846): FILTER, the
// (#CHROM, POS, END, ID) order of pavlib/cigarcall.py:320,343 and the TSV text pandas.to_csv writes - straight from the
// record streams resident in HBM.  The Python mirror (pav_amd/cigarcall.records_to_frames) builds all-object DataFrames
// like the reference does, which costs ~13 us per row (~90 s per haplotype); this writer produces byte-identical text at
// memory speed and gzips it in parallel as concatenated members.
//   device : sort keys of the SNV rows, stable radix sort (rocPRIM primitive), gather + FILTER
//   host   : INDEL order (stable sort; ID tie-break on TYPE then on the decimal *string* of SVLEN), text, gzip (zlib)
#include "common.h"
#include "textdev.h"
#include "textio.h"

#include <rocprim/rocprim.hpp>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <thread>

namespace pav {

struct SnvOut { uint32_t aln, pos, qry_pos; uint8_t ref, alt, pass, pad; };

__device__ __forceinline__ uint8_t up8(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; }

// key = chrom rank | POS | REF.upper() | ALT.upper(): the order of (#CHROM, POS, END = POS + 1, ID) for SNV rows
// merged tables (call_batch given): chrom rank (12 bits) | POS | CALL_BATCH (4 bits) | REF.upper() | ALT.upper() - the batch
// files concatenated in batch order and stable-sorted by (#CHROM, POS) (rules/call.snakefile:777-786)
__global__ __launch_bounds__(256) void snv_sort_keys(const pav_snv *__restrict__ snv, uint64_t n, const pav_aln *__restrict__ aln,
                                                     const uint16_t *__restrict__ chrom_rank, const uint8_t *__restrict__ batch,
                                                     unsigned long long *__restrict__ keys, uint32_t *__restrict__ vals) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const pav_snv s = snv[i];
    const uint64_t rank = chrom_rank[aln[s.aln].ref_id];
    if (batch) keys[i] = rank << 52 | (uint64_t)s.pos << 20 | (uint64_t)batch[s.aln] << 16 | (uint64_t)up8(s.ref) << 8 | up8(s.alt);
    else keys[i] = rank << 48 | (uint64_t)s.pos << 16 | (uint64_t)up8(s.ref) << 8 | up8(s.alt);
    vals[i] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void snv_gather(const pav_snv *__restrict__ snv, const uint32_t *__restrict__ order, uint64_t n,
                                                  const long long *__restrict__ trim_pos, const long long *__restrict__ trim_end,
                                                  SnvOut *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const pav_snv s = snv[order[i]];
    SnvOut o;
    o.aln = s.aln; o.pos = s.pos; o.qry_pos = s.qry_pos; o.ref = s.ref; o.alt = s.alt; o.pad = 0;
    o.pass = 1;
    if (trim_pos) o.pass = ((long long)s.pos > trim_pos[s.aln] && (long long)s.pos + 1 < trim_end[s.aln]) ? 1 : 0;   // call.snakefile:826-828
    out[i] = o;
}

const std::vector<std::string> &seq_names(pav_ctx *ctx, int role);   // invscan.cpp

}  // namespace pav

namespace pav {

// Everything the host phase of the writer reads, owned by the job: it may run behind the call that queued it
// (pav_cigar_write_tables_begin / _end), on a thread of its own, while the context goes on with the flagging and the scan.
struct TableWork {
    std::vector<pav_aln> aln; std::vector<SnvOut> snv; std::vector<pav_indel> ind; std::vector<uint8_t> blob, batch8;
    std::vector<uint16_t> rank; std::vector<int64_t> align_index, trim_pos, trim_end;
    std::vector<std::string> rnames, tnames;
    std::string hap, snv_path, insdel_path, err;
    bool have_snv = false, have_insdel = false, with_filter = false, merged = false;
    int threads = 1, level = 6;
    uint64_t n_snv = 0, n_ind = 0;
    // the device writer (textdev.hip; the default): text and gzip are made in HBM on a stream of the writer's own, from the records
    // resident in the context - nothing but the files' bytes comes to the host.  `job` then holds what the write needs.
    bool device = false;
    pav_ctx *ctx = nullptr;
    CigarTextJob job;
};
struct TableJob { std::shared_ptr<TableWork> work; std::thread th; int rc = PAV_OK; bool pending = false; };

static int table_host_phase(TableWork &W) {
    if (W.device) return text_cigar_tables(W.ctx, W.job, W.err);
    const std::vector<pav_aln> &aln = W.aln; const std::vector<SnvOut> &snv = W.snv; const std::vector<pav_indel> &ind = W.ind;
    const std::vector<uint8_t> &blob = W.blob, &batch8 = W.batch8; const std::vector<uint16_t> &rank = W.rank;
    const std::vector<std::string> &rnames = W.rnames, &tnames = W.tnames;
    const uint32_t n_ref = (uint32_t)rnames.size(), n_tig = (uint32_t)tnames.size();
    const uint64_t n_snv = W.n_snv, n_ind = W.n_ind;
    const bool with_filter = W.with_filter, merged = W.merged;
    const int threads = W.threads, level = W.level;
    const std::string hap = csv_field(W.hap);
    std::vector<std::string> chrom_f(n_ref), tig_f(n_tig);          // quoted forms are only needed when a name has odd characters
    bool odd_names = false;
    for (uint32_t i = 0; i < n_ref; ++i) { chrom_f[i] = csv_field(rnames[i]); odd_names |= chrom_f[i] != rnames[i]; }
    for (uint32_t i = 0; i < n_tig; ++i) { tig_f[i] = csv_field(tnames[i]); odd_names |= tig_f[i] != tnames[i]; }
    auto field = [&](std::string &s, const std::string &plain) { if (odd_names) s += csv_field(plain); else s += plain; };

    int rc = PAV_OK;
    if (W.have_snv) {
        std::string header = "#CHROM\tPOS\tEND\tID\tSVTYPE\tSVLEN\tREF\tALT\tHAP\tQRY_REGION\tQRY_STRAND\tCI\tALIGN_INDEX\tCALL_SOURCE";
        header += with_filter ? "\tFILTER\n" : "\n";
        rc = write_table(nullptr, W.snv_path.c_str(), header, n_snv, threads, level, [&](uint64_t i, std::string &s) {
            const SnvOut &r = snv[i];
            const pav_aln &a = aln[r.aln];
            const std::string &chrom = rnames[a.ref_id];
            s += chrom_f[a.ref_id]; s += '\t'; put_u64(s, r.pos); s += '\t'; put_u64(s, (uint64_t)r.pos + 1); s += '\t';
            std::string id = chrom; id += '-'; put_u64(id, (uint64_t)r.pos + 1); id += "-SNV-";
            id += (char)((r.ref >= 'a' && r.ref <= 'z') ? r.ref - 32 : r.ref); id += (char)((r.alt >= 'a' && r.alt <= 'z') ? r.alt - 32 : r.alt);
            field(s, id);
            s += "\tSNV\t1\t"; s += (char)r.ref; s += '\t'; s += (char)r.alt; s += '\t'; s += hap; s += '\t';
            std::string q = tnames[a.tig_id]; q += ':'; put_u64(q, (uint64_t)r.qry_pos + 1); q += '-'; put_u64(q, (uint64_t)r.qry_pos + 1);
            field(s, q);
            s += a.rev ? "\t-\t0\t" : "\t+\t0\t";
            put_i64(s, W.align_index[r.aln]);
            s += "\tCIGAR";
            if (with_filter) s += r.pass ? "\tPASS" : "\tTRIM";
            s += '\n';
        });
        if (rc != PAV_OK) { W.err = pav_last_error(nullptr); return rc; }
    }
    if (W.have_insdel) {
        std::vector<uint32_t> order(n_ind);
        for (uint64_t i = 0; i < n_ind; ++i) order[i] = (uint32_t)i;
        auto dec = [](uint32_t v) { char b[16]; int n = snprintf(b, sizeof b, "%u", v); return std::string(b, (size_t)n); };
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
            const pav_indel &p = ind[x], &q = ind[y];
            const uint16_t rp = rank[aln[p.aln].ref_id], rq = rank[aln[q.aln].ref_id];
            if (rp != rq) return rp < rq;
            if (p.pos != q.pos) return p.pos < q.pos;
            if (p.end != q.end) return p.end < q.end;
            if (p.svtype != q.svtype) return p.svtype > q.svtype;            // 'DEL' < 'INS' (svtype 1 = DEL)
            if (p.svlen != q.svlen) return dec(p.svlen) < dec(q.svlen);      // ID compares the decimal strings
            if (merged && batch8[p.aln] != batch8[q.aln]) return batch8[p.aln] < batch8[q.aln];   // equal keys keep concat order
            return false;
        });
        std::string header = "#CHROM\tPOS\tEND\tID\tSVTYPE\tSVLEN\tHAP\tQRY_REGION\tQRY_STRAND\tCI\tALIGN_INDEX\tLEFT_SHIFT\tHOM_REF\tHOM_TIG\tCALL_SOURCE\tSEQ";
        header += with_filter ? "\tFILTER\n" : "\n";
        rc = write_table(nullptr, W.insdel_path.c_str(), header, n_ind, threads, level, [&](uint64_t i, std::string &s) {
            const pav_indel &r = ind[order[i]];
            const pav_aln &a = aln[r.aln];
            const std::string &chrom = rnames[a.ref_id];
            const char *type = r.svtype == 0 ? "INS" : "DEL";
            s += chrom_f[a.ref_id]; s += '\t'; put_u64(s, r.pos); s += '\t'; put_u64(s, r.end); s += '\t';
            std::string id = chrom; id += '-'; put_u64(id, (uint64_t)r.pos + 1); id += '-'; id += type; id += '-'; put_u64(id, r.svlen);
            field(s, id);
            s += '\t'; s += type; s += '\t'; put_u64(s, r.svlen); s += '\t'; s += hap; s += '\t';
            std::string q = tnames[a.tig_id]; q += ':'; put_u64(q, (uint64_t)r.qry_pos + 1); q += '-'; put_u64(q, r.qry_end);
            field(s, q);
            s += a.rev ? "\t-\t0\t" : "\t+\t0\t";
            put_i64(s, W.align_index[r.aln]); s += '\t';
            put_u64(s, r.left_shift); s += '\t';
            put_u64(s, r.hom_ref_l); s += ','; put_u64(s, r.hom_ref_r); s += '\t';
            put_u64(s, r.hom_tig_l); s += ','; put_u64(s, r.hom_tig_r);
            s += "\tCIGAR\t";
            s.append(reinterpret_cast<const char *>(blob.data()) + r.seq_off, r.svlen);
            if (with_filter) {
                const bool pass = (long long)r.pos > W.trim_pos[r.aln] && (long long)r.end < W.trim_end[r.aln];   // call.snakefile:838-840
                s += pass ? "\tPASS" : "\tTRIM";
            }
            s += '\n';
        });
        if (rc != PAV_OK) { W.err = pav_last_error(nullptr); return rc; }
    }
    return PAV_OK;
}

// a device write reads the records resident in the context: whoever is about to replace them waits for it first
void table_writer_quiesce(pav_ctx *ctx) {
    TableJob *J = static_cast<TableJob *>(ctx->table_writer);
    if (J && J->th.joinable()) J->th.join();
}

void table_writer_release(pav_ctx *ctx) {
    TableJob *J = static_cast<TableJob *>(ctx->table_writer);
    if (!J) return;
    if (J->th.joinable()) J->th.join();
    delete J;
    ctx->table_writer = nullptr;
}

}  // namespace pav

using namespace pav;

// Device phase: order + FILTER of the SNV rows on the device, every record stream to the host (W owns the copies).
static int table_device_phase(pav_ctx *ctx, const pav_table_opts *o, TableWork &W) {
    if (!ctx || !o || !o->hap || !o->align_index) return fail(ctx, PAV_E_ARG, "pav_cigar_write_tables: null argument");
    if (!ctx->cigar_called) return fail(ctx, PAV_E_STATE, "pav_cigar_write_tables: no successful pav_cigar_call to write");
    if ((o->trim_pos == nullptr) != (o->trim_end == nullptr)) return fail(ctx, PAV_E_ARG, "pav_cigar_write_tables: trim_pos and trim_end go together");
    PAV_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint32_t n_aln = ctx->n_aln;
    const uint64_t n_snv = ctx->counts.n_snv, n_ind = ctx->counts.n_indel;
    const std::vector<std::string> &rnames = seq_names(ctx, PAV_ROLE_REF), &tnames = seq_names(ctx, PAV_ROLE_TIG);
    const uint32_t n_ref = ctx->seq[PAV_ROLE_REF].n, n_tig = ctx->seq[PAV_ROLE_TIG].n;
    if (rnames.size() != n_ref || tnames.size() != n_tig) return fail(ctx, PAV_E_STATE, "pav_cigar_write_tables: pav_seq_set_names has not been called for both stores");
    if (n_ref > 65535) return fail(ctx, PAV_E_LIMIT, "pav_cigar_write_tables: more than 65535 reference records");
    const bool with_filter = o->trim_pos != nullptr;
    const bool merged = o->call_batch != nullptr;
    std::vector<uint8_t> &batch8 = W.batch8;
    batch8.assign(merged ? n_aln : 0, 0);
    if (merged) {
        if (n_ref > 4096) return fail(ctx, PAV_E_LIMIT, "pav_cigar_write_tables: merged tables need <= 4096 reference records");
        for (uint32_t i = 0; i < n_aln; ++i) {
            if (o->call_batch[i] < 0 || o->call_batch[i] > 15) return fail(ctx, PAV_E_LIMIT, "pav_cigar_write_tables: CALL_BATCH %lld of row %u is outside 0..15", (long long)o->call_batch[i], i);
            batch8[i] = (uint8_t)o->call_batch[i];
        }
        for (uint32_t i = 0; i < n_ref; ++i) {
            char *e = nullptr;
            (void)strtod(rnames[i].c_str(), &e);
            if (!rnames[i].empty() && e && *e == 0)
                return fail(ctx, PAV_E_LIMIT, "pav_cigar_write_tables: reference name '%s' reads as a number: rule call_cigar_merge re-reads its "
                            "inputs without a string dtype and would order such names numerically", rnames[i].c_str());
        }
    }
    int threads = o->threads > 0 ? o->threads : (int)std::min<unsigned>(16, std::max<unsigned>(1, std::thread::hardware_concurrency()));
    const int level = o->gzip_level > 0 ? o->gzip_level : 6;

    // #CHROM compares as Python str: rank of each record name in byte order
    std::vector<uint32_t> by_name(n_ref);
    for (uint32_t i = 0; i < n_ref; ++i) by_name[i] = i;
    std::sort(by_name.begin(), by_name.end(), [&](uint32_t a, uint32_t b) { return rnames[a] < rnames[b]; });
    std::vector<uint16_t> &rank = W.rank;
    rank.assign(n_ref, 0);
    for (uint32_t i = 0; i < n_ref; ++i) rank[by_name[i]] = (uint16_t)(i && rnames[by_name[i]] == rnames[by_name[i - 1]] ? rank[by_name[i - 1]] : i);

    W.n_snv = n_snv; W.n_ind = n_ind;
    // ---- the device writer: no record leaves the GPU -----------------------------------------------------------------------------
    bool plain = csv_field(o->hap) == o->hap;                       // (a name to_csv would quote: the host writer knows how)
    for (const std::string &nm : rnames) plain = plain && csv_field(nm) == nm;
    for (const std::string &nm : tnames) plain = plain && csv_field(nm) == nm;
    if (plain && device_writer_enabled()) {
        CigarTextJob &J = W.job;
        J.rnames = rnames; J.tnames = tnames; J.rank = rank; J.batch8 = batch8;
        J.align_index.assign(o->align_index, o->align_index + n_aln);
        J.filter = with_filter;
        if (with_filter) { J.trim_pos.assign(o->trim_pos, o->trim_pos + n_aln); J.trim_end.assign(o->trim_end, o->trim_end + n_aln); }
        J.hap = o->hap; J.have_snv = o->snv_path != nullptr; J.have_insdel = o->insdel_path != nullptr;
        if (o->snv_path) J.snv_path = o->snv_path;
        if (o->insdel_path) J.insdel_path = o->insdel_path;
        J.level = level; J.n_snv = n_snv; J.n_ind = n_ind;
        { const int rch = wait_homology(ctx); if (rch != PAV_OK) return rch; }        // the SEQ column and the homology columns
        if (!ctx->writer_ready) PAV_HIP(ctx, hipEventCreateWithFlags(&ctx->writer_ready, hipEventDisableTiming));
        PAV_HIP(ctx, hipEventRecord(ctx->writer_ready, st));
        J.ready = ctx->writer_ready;
        W.device = true; W.ctx = ctx;
        return PAV_OK;
    }
    std::vector<pav_aln> &aln = W.aln;
    aln.resize(n_aln);
    if (n_aln) PAV_HIP(ctx, hipMemcpyAsync(aln.data(), ctx->d_aln.p, sizeof(pav_aln) * n_aln, hipMemcpyDeviceToHost, st));

    // ---- SNV rows: device sort + gather + FILTER ------------------------------------------------------------
    std::vector<SnvOut> &snv = W.snv;
    snv.resize(n_snv);
    if (n_snv && o->snv_path) {
        size_t tmp_bytes = 0;
        unsigned long long *kin = nullptr, *kout = nullptr; uint32_t *vin = nullptr, *vout = nullptr;
        PAV_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tmp_bytes, kin, kout, vin, vout, (size_t)n_snv, 0, 64, st));
        const size_t need = 2 * 8 * n_snv + 2 * 4 * n_snv + tmp_bytes + sizeof(SnvOut) * n_snv + 2 * n_ref + 17 * (size_t)n_aln + 1024;
        PAV_HIP(ctx, ctx->d_tmp.reserve(need));
        uint8_t *p = ctx->d_tmp.as<uint8_t>();
        kin = reinterpret_cast<unsigned long long *>(p); p += 8 * n_snv;
        kout = reinterpret_cast<unsigned long long *>(p); p += 8 * n_snv;
        SnvOut *d_out = reinterpret_cast<SnvOut *>(p); p += sizeof(SnvOut) * n_snv;
        long long *d_tp = reinterpret_cast<long long *>(p); p += 8 * (size_t)n_aln;
        long long *d_te = reinterpret_cast<long long *>(p); p += 8 * (size_t)n_aln;
        vin = reinterpret_cast<uint32_t *>(p); p += 4 * n_snv;
        vout = reinterpret_cast<uint32_t *>(p); p += 4 * n_snv;
        uint16_t *d_rank = reinterpret_cast<uint16_t *>(p); p += (2 * (size_t)n_ref + 15) / 16 * 16;
        uint8_t *d_batch = p; p += ((size_t)n_aln + 15) / 16 * 16;
        void *d_sort_tmp = p;
        if (merged && n_aln) PAV_HIP(ctx, hipMemcpyAsync(d_batch, batch8.data(), n_aln, hipMemcpyHostToDevice, st));
        PAV_HIP(ctx, hipMemcpyAsync(d_rank, rank.data(), 2 * (size_t)n_ref, hipMemcpyHostToDevice, st));
        if (with_filter) {
            PAV_HIP(ctx, hipMemcpyAsync(d_tp, o->trim_pos, 8 * (size_t)n_aln, hipMemcpyHostToDevice, st));
            PAV_HIP(ctx, hipMemcpyAsync(d_te, o->trim_end, 8 * (size_t)n_aln, hipMemcpyHostToDevice, st));
        }
        PAV_LAUNCH(ctx, "snv_sort_keys", snv_sort_keys, (uint32_t)((n_snv + 255) / 256), 256, 0, ctx->d_snv.as<pav_snv>(), n_snv,
                   ctx->d_aln.as<pav_aln>(), d_rank, merged ? d_batch : nullptr, kin, vin);
        {
            int tok = prof_begin(ctx, "rocprim::radix_sort_pairs");
            hipError_t e = rocprim::radix_sort_pairs(d_sort_tmp, tmp_bytes, kin, kout, vin, vout, (size_t)n_snv, 0, 64, st);
            prof_end(ctx, tok);
            PAV_HIP(ctx, e);
        }
        PAV_LAUNCH(ctx, "snv_gather", snv_gather, (uint32_t)((n_snv + 255) / 256), 256, 0, ctx->d_snv.as<pav_snv>(), vout, n_snv,
                   with_filter ? d_tp : nullptr, with_filter ? d_te : nullptr, d_out);
        PAV_HIP(ctx, hipMemcpyAsync(snv.data(), d_out, sizeof(SnvOut) * n_snv, hipMemcpyDeviceToHost, st));
    }
    // ---- INDEL rows: records + SEQ blob to the host, stable sort there ------------------------------------------
    { const int rch = wait_homology(ctx); if (rch != PAV_OK) return rch; }
    std::vector<pav_indel> &ind = W.ind;
    ind.resize(n_ind);
    std::vector<uint8_t> &blob = W.blob;
    blob.resize(ctx->counts.seq_bytes + 1);
    if (n_ind && o->insdel_path) {
        PAV_HIP(ctx, hipMemcpyAsync(ind.data(), ctx->d_indel.p, sizeof(pav_indel) * n_ind, hipMemcpyDeviceToHost, st));
        if (ctx->counts.seq_bytes) PAV_HIP(ctx, hipMemcpyAsync(blob.data(), ctx->d_seqblob.p, ctx->counts.seq_bytes, hipMemcpyDeviceToHost, st));
    }
    PAV_HIP(ctx, hipStreamSynchronize(st));

    W.n_snv = n_snv; W.n_ind = n_ind; W.with_filter = with_filter; W.merged = merged; W.threads = threads; W.level = level;
    W.hap = o->hap; W.have_snv = o->snv_path != nullptr; W.have_insdel = o->insdel_path != nullptr;
    if (o->snv_path) W.snv_path = o->snv_path;
    if (o->insdel_path) W.insdel_path = o->insdel_path;
    W.rnames = rnames; W.tnames = tnames;
    W.align_index.assign(o->align_index, o->align_index + n_aln);
    if (with_filter) { W.trim_pos.assign(o->trim_pos, o->trim_pos + n_aln); W.trim_end.assign(o->trim_end, o->trim_end + n_aln); }
    return PAV_OK;
}

extern "C" {

// The two tables of the last pav_cigar_call; returns when the files are written.
int pav_cigar_write_tables(pav_ctx *ctx, const pav_table_opts *o, uint64_t *n_snv_rows, uint64_t *n_insdel_rows) {
    if (ctx && ctx->table_writer && static_cast<TableJob *>(ctx->table_writer)->pending)
        return fail(ctx, PAV_E_STATE, "pav_cigar_write_tables: a write begun with pav_cigar_write_tables_begin is pending");
    TableWork W;
    int rc = table_device_phase(ctx, o, W);
    if (rc != PAV_OK) return rc;
    rc = table_host_phase(W);
    if (rc != PAV_OK) return fail(ctx, rc, "%s", W.err.c_str());
    if (n_snv_rows) *n_snv_rows = W.n_snv;
    if (n_insdel_rows) *n_insdel_rows = W.n_ind;
    return PAV_OK;
}

// The same in two halves: _begin runs the device phase (sort, FILTER, the record streams to host memory the job owns) and
// starts the host phase - text, gzip members, the files - on a thread of its own; the context is free for the stages that
// follow (flagging, scan: they read the resident records, which the writer no longer touches).  _end waits for the files.
int pav_cigar_write_tables_begin(pav_ctx *ctx, const pav_table_opts *o) {
    if (!ctx) return PAV_E_ARG;
    TableJob *J = static_cast<TableJob *>(ctx->table_writer);
    if (J && J->pending) return fail(ctx, PAV_E_STATE, "pav_cigar_write_tables_begin: the write before this one has not been waited for");
    if (!J) { J = new TableJob(); ctx->table_writer = J; }
    J->work = std::make_shared<TableWork>();
    const int rc = table_device_phase(ctx, o, *J->work);
    if (rc != PAV_OK) { J->work.reset(); return rc; }
    J->pending = true; J->rc = PAV_OK;
    std::shared_ptr<TableWork> w = J->work;
    try {
        J->th = std::thread([J, w] {
            // (an exception that left this thread would end the whole rank process, not this haplotype's write)
            try { J->rc = table_host_phase(*w); }
            catch (const std::exception &ex) { J->rc = PAV_E_STATE; w->err = std::string("table writer: ") + ex.what(); }
            catch (...) { J->rc = PAV_E_STATE; w->err = "table writer: unknown exception"; }
        });
    } catch (const std::exception &ex) {
        J->pending = false; J->work.reset();
        return fail(ctx, PAV_E_STATE, "pav_cigar_write_tables_begin: cannot start the writer thread: %s", ex.what());
    }
    return PAV_OK;
}

int pav_cigar_write_tables_end(pav_ctx *ctx, uint64_t *n_snv_rows, uint64_t *n_insdel_rows) {
    if (!ctx) return PAV_E_ARG;
    TableJob *J = static_cast<TableJob *>(ctx->table_writer);
    if (!J || !J->pending) return fail(ctx, PAV_E_STATE, "pav_cigar_write_tables_end: no write has been begun");
    if (J->th.joinable()) J->th.join();
    J->pending = false;
    std::shared_ptr<TableWork> w = std::move(J->work);
    if (J->rc != PAV_OK) return fail(ctx, J->rc, "%s", w->err.c_str());
    if (n_snv_rows) *n_snv_rows = w->n_snv;
    if (n_insdel_rows) *n_insdel_rows = w->n_ind;
    return PAV_OK;
}

}  // extern "C"
