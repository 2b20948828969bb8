// Table text on the device: one lane formats one row, twice - once for its length, once, behind a prefix sum over the lengths,
// into its place in the file's text - with the same row function and two sinks, so the two passes cannot disagree.  The text
// stays in HBM and is compressed there (deflate.hip); only the gzip files cross PCIe.  The host writers this replaces formatted
// 2.2 GB of text per haplotype on sixteen threads and ran zlib over it (tables.hip, invscan.cpp; round 4: 1.3 - 3.1 s per haplotype
// with the GPU idle).  The bytes are pandas': DataFrame.to_csv(sep='\t', index=False) of the frames of pavlib/cigarcall.py:125-134,
// 199-209 (columns and their order), rules/call.snakefile:813-846 (FILTER), scripts/density.py:329-342 + pavlib/inv.py:457-561
// (density table columns); integers in decimal, floats as repr (fmt_dev.h), NaN as the empty field.
#include "textdev.h"

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "devgz.h"
#include "fmt_dev.h"
#include "scan_dev.h"

namespace pav {

namespace {

// ---- sinks --------------------------------------------------------------------------------------------------------------
struct LenSink {
    uint32_t n = 0;
    __device__ __forceinline__ void ch(uint8_t) { ++n; }
    __device__ __forceinline__ void lit(const char *, uint32_t k) { n += k; }
    __device__ __forceinline__ void bytes(const uint8_t *, uint32_t k) { n += k; }
    __device__ __forceinline__ void u64(uint64_t v) { n += fmt::dec_len(v); }
    __device__ __forceinline__ void i64(int64_t v) { n += fmt::i64_len(v); }
    __device__ __forceinline__ void f64(double v) { n += fmt::f64_repr_len(v); }
    __device__ __forceinline__ void blob(uint64_t, uint32_t k) { n += k; }
};
struct LongCopy { uint64_t src, dst; uint32_t len, pad; };
struct MemSink {
    uint8_t *p;
    uint8_t *text;                       // start of the text arena (long copies are queued by offset)
    const uint8_t *seq;                  // SEQ blob
    LongCopy *longs; uint32_t *n_longs;  // copies of more than INLINE_SEQ bytes are left to k_long_copies (a wave each)
    __device__ __forceinline__ void ch(uint8_t c) { *p++ = c; }
    __device__ __forceinline__ void lit(const char *s, uint32_t k) { for (uint32_t i = 0; i < k; ++i) p[i] = (uint8_t)s[i]; p += k; }
    __device__ __forceinline__ void bytes(const uint8_t *s, uint32_t k) { for (uint32_t i = 0; i < k; ++i) p[i] = s[i]; p += k; }
    __device__ __forceinline__ void u64(uint64_t v) { p += fmt::put_u64(p, v); }
    __device__ __forceinline__ void i64(int64_t v) { p += fmt::put_i64(p, v); }
    __device__ __forceinline__ void f64(double v) { p += fmt::put_f64_repr(p, v); }
    __device__ __forceinline__ void blob(uint64_t off, uint32_t k) {
        constexpr uint32_t INLINE_SEQ = 48;
        if (k <= INLINE_SEQ) { for (uint32_t i = 0; i < k; ++i) p[i] = seq[off + i]; }
        else { const uint32_t at = atomicAdd(n_longs, 1u); longs[at] = LongCopy{off, (uint64_t)(p - text), k, 0}; }
        p += k;
    }
};

struct SnvOut { uint32_t aln, pos, qry_pos; uint8_t ref, alt, pass, pad; };
__device__ __forceinline__ uint8_t up8(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; }

struct Names { const uint8_t *blob; const uint32_t *off; };          // name i = blob[off[i] .. off[i + 1])

struct CigarRows {                       // what a row of either table reads
    const pav_aln *aln; Names ref, tig;
    const uint8_t *hap; uint32_t hap_len;
    const long long *align_index, *trim_pos, *trim_end;               // trim_*: nullptr = no FILTER column
    const SnvOut *snv;
    const pav_indel *ind; const uint32_t *ind_order;
};

// pavlib/cigarcall.py:125-134 + FILTER (rules/call.snakefile:826-828)
template <class S> __device__ __forceinline__ void snv_row(S &s, const CigarRows &C, uint64_t i) {
    const SnvOut r = C.snv[i];
    const pav_aln a = C.aln[r.aln];
    const uint8_t *chrom = C.ref.blob + C.ref.off[a.ref_id]; const uint32_t cl = C.ref.off[a.ref_id + 1] - C.ref.off[a.ref_id];
    const uint8_t *tig = C.tig.blob + C.tig.off[a.tig_id]; const uint32_t tl = C.tig.off[a.tig_id + 1] - C.tig.off[a.tig_id];
    s.bytes(chrom, cl); s.ch('\t'); s.u64(r.pos); s.ch('\t'); s.u64((uint64_t)r.pos + 1); s.ch('\t');
    s.bytes(chrom, cl); s.ch('-'); s.u64((uint64_t)r.pos + 1); s.lit("-SNV-", 5); s.ch(up8(r.ref)); s.ch(up8(r.alt));
    s.lit("\tSNV\t1\t", 7); s.ch(r.ref); s.ch('\t'); s.ch(r.alt); s.ch('\t'); s.bytes(C.hap, C.hap_len); s.ch('\t');
    s.bytes(tig, tl); s.ch(':'); s.u64((uint64_t)r.qry_pos + 1); s.ch('-'); s.u64((uint64_t)r.qry_pos + 1);
    s.lit(a.rev ? "\t-\t0\t" : "\t+\t0\t", 5); s.i64(C.align_index[r.aln]); s.lit("\tCIGAR", 6);
    if (C.trim_pos) s.lit(r.pass ? "\tPASS" : "\tTRIM", 5);
    s.ch('\n');
}

// pavlib/cigarcall.py:199-209, 268-278 + FILTER (rules/call.snakefile:838-840)
template <class S> __device__ __forceinline__ void indel_row(S &s, const CigarRows &C, uint64_t i) {
    const pav_indel r = C.ind[C.ind_order[i]];
    const pav_aln a = C.aln[r.aln];
    const uint8_t *chrom = C.ref.blob + C.ref.off[a.ref_id]; const uint32_t cl = C.ref.off[a.ref_id + 1] - C.ref.off[a.ref_id];
    const uint8_t *tig = C.tig.blob + C.tig.off[a.tig_id]; const uint32_t tl = C.tig.off[a.tig_id + 1] - C.tig.off[a.tig_id];
    const char *type = r.svtype == 0 ? "INS" : "DEL";
    s.bytes(chrom, cl); s.ch('\t'); s.u64(r.pos); s.ch('\t'); s.u64(r.end); s.ch('\t');
    s.bytes(chrom, cl); s.ch('-'); s.u64((uint64_t)r.pos + 1); s.ch('-'); s.lit(type, 3); s.ch('-'); s.u64(r.svlen);
    s.ch('\t'); s.lit(type, 3); s.ch('\t'); s.u64(r.svlen); s.ch('\t'); s.bytes(C.hap, C.hap_len); s.ch('\t');
    s.bytes(tig, tl); s.ch(':'); s.u64((uint64_t)r.qry_pos + 1); s.ch('-'); s.u64(r.qry_end);
    s.lit(a.rev ? "\t-\t0\t" : "\t+\t0\t", 5); s.i64(C.align_index[r.aln]); s.ch('\t');
    s.u64(r.left_shift); s.ch('\t'); s.u64(r.hom_ref_l); s.ch(','); s.u64(r.hom_ref_r); s.ch('\t');
    s.u64(r.hom_tig_l); s.ch(','); s.u64(r.hom_tig_r); s.lit("\tCIGAR\t", 7);
    s.blob(r.seq_off, r.svlen);
    if (C.trim_pos) {
        const bool pass = (long long)r.pos > C.trim_pos[r.aln] && (long long)r.end < C.trim_end[r.aln];
        s.lit(pass ? "\tPASS" : "\tTRIM", 5);
    }
    s.ch('\n');
}

struct DenRows { const DenTableDev *tab; const uint64_t *row0; uint32_t n_tab; };   // row0[t] = first row of table t, row0[n_tab] = all
__device__ __forceinline__ uint32_t den_table_of(const DenRows &D, uint64_t i) {
    uint32_t lo = 0, hi = D.n_tab;                                       // last table whose first row is at or before i
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (D.row0[mid] <= i) lo = mid; else hi = mid; }
    return lo;
}
// scripts/density.py:329-342 (INDEX ... KMER), pavlib/inv.py:524-561 (FLANK, MATCH)
template <class S> __device__ __forceinline__ void den_row(S &s, const DenRows &D, uint32_t t, uint64_t i) {
    const DenTableDev T = D.tab[t];
    const uint64_t r = i - D.row0[t];
    s.u64(T.index[r]); s.ch('\t'); s.i64(T.state_mer[r]); s.ch('\t'); s.i64(T.state[r]); s.ch('\t');
    s.f64(T.k0[r]); s.ch('\t'); s.f64(T.k1 ? T.k1[r] : 0.0); s.ch('\t'); s.f64(T.k2[r]); s.ch('\t');
    s.u64(T.kmer[r]); s.ch('\t');
    const uint32_t f = T.flank[r], m = T.match[r] & 3u;
    if (f == 1) s.lit("UP", 2); else if (f == 2) s.lit("DN", 2);
    s.ch('\t');
    if (m == 1) s.lit("SAME", 4); else if (m == 2) s.lit("OTHER", 5);     // 3 = NaN: the empty na_rep
    s.ch('\n');
}

enum { ROWS_SNV = 0, ROWS_INDEL = 1 };
template <int KIND> __global__ __launch_bounds__(256) void k_cigar_rowlen(CigarRows C, uint64_t n, uint32_t *__restrict__ len) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    LenSink s;
    if (KIND == ROWS_SNV) snv_row(s, C, i); else indel_row(s, C, i);
    len[i] = s.n;
}
template <int KIND> __global__ __launch_bounds__(256) void k_cigar_rowtext(CigarRows C, uint64_t n, const uint64_t *__restrict__ roff, uint64_t base,
                                                                         uint8_t *text, const uint8_t *seq, LongCopy *longs, uint32_t *n_longs) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    MemSink s{text + base + roff[i], text, seq, longs, n_longs};
    if (KIND == ROWS_SNV) snv_row(s, C, i); else indel_row(s, C, i);
}
__global__ __launch_bounds__(256) void k_den_rowlen(DenRows D, uint64_t n, uint32_t *__restrict__ len) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    LenSink s;
    den_row(s, D, den_table_of(D, i), i);
    len[i] = s.n;
}
// fbase[t] = where row row0[t] of table t goes - the table's text offset + its header - minus roff[row0[t]]
__global__ __launch_bounds__(256) void k_den_rowtext(DenRows D, uint64_t n, const uint64_t *__restrict__ roff, const uint64_t *__restrict__ fbase,
                                                     uint8_t *text) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t t = den_table_of(D, i);
    MemSink s{text + fbase[t] + roff[i], text, nullptr, nullptr, nullptr};
    den_row(s, D, t, i);
}
// the header line at the start of every file's text
__global__ __launch_bounds__(64) void k_headers(uint8_t *text, const uint64_t *__restrict__ file_off, uint32_t n_files, const uint8_t *__restrict__ hdr,
                                                uint32_t hdr_len) {
    if (blockIdx.x >= n_files) return;
    uint8_t *d = text + file_off[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < hdr_len; i += 64) d[i] = hdr[i];
}
// SEQ columns too long for the row's lane: a wave per copy
__global__ __launch_bounds__(64) void k_long_copies(uint8_t *text, const uint8_t *__restrict__ seq, const LongCopy *__restrict__ longs,
                                                    const uint32_t *__restrict__ n_longs) {
    const uint32_t n = *n_longs;
    for (uint32_t e = blockIdx.x; e < n; e += gridDim.x) {
        const LongCopy c = longs[e];
        for (uint32_t i = threadIdx.x; i < c.len; i += 64) text[c.dst + i] = seq[c.src + i];
    }
}

// ---- order of the two tables --------------------------------------------------------------------------------------------
// SNV rows: key = chrom rank | POS | REF.upper() | ALT.upper() - the order of (#CHROM, POS, END = POS + 1, ID); merged tables:
// chrom rank (12 bits) | POS | CALL_BATCH (4 bits) | REF | ALT - the batch files concatenated in batch order and stable-sorted by
// (#CHROM, POS) (rules/call.snakefile:777-786)
__global__ __launch_bounds__(256) void k_snv_keys(const pav_snv *__restrict__ snv, uint64_t n, const pav_aln *__restrict__ aln,
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
__global__ __launch_bounds__(256) void k_snv_gather(const pav_snv *__restrict__ snv, const uint32_t *__restrict__ order, uint64_t n,
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
// INS / DEL rows: sort_values(['#CHROM', 'POS', 'END', 'ID']) (pavlib/cigarcall.py:343; merged: rules/call.snakefile:786) with the
// ID compared as a STRING: chrom-(POS+1)-TYPE-SVLEN ties on TYPE ('DEL' < 'INS') and then on the decimal string of SVLEN
// ('10' < '9'); equal keys keep the order of the records (merged tables: batch order).  Three stable radix passes, last key first:
//   pass[0] = TYPE | SVLEN as a string (its digits left-aligned in nine places, then the digit count) | CALL_BATCH
//   pass[1] = END        pass[2] = chrom rank | POS
__global__ __launch_bounds__(256) void k_indel_keys(const pav_indel *__restrict__ ind, const uint32_t *__restrict__ order, uint64_t n, int pass,
                                                    const pav_aln *__restrict__ aln, const uint16_t *__restrict__ chrom_rank,
                                                    const uint8_t *__restrict__ batch, unsigned long long *__restrict__ keys, uint32_t *__restrict__ vals) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t src = order ? order[i] : (uint32_t)i;
    const pav_indel r = ind[src];
    unsigned long long k;
    if (pass == 0) {
        const uint32_t nd = fmt::dec_len(r.svlen);
        uint64_t str = r.svlen;
        for (uint32_t d = nd; d < 10; ++d) str *= 10;                    // svlen < 2^32: ten places
        k = (unsigned long long)(r.svtype ? 0u : 1u) << 44 | (unsigned long long)str << 8 | (unsigned long long)nd << 4 | (batch ? batch[r.aln] : 0u);
    } else if (pass == 1) k = r.end;
    else k = (unsigned long long)chrom_rank[aln[r.aln].ref_id] << 32 | r.pos;
    keys[i] = k; vals[i] = src;
}

// ---- per-writer device state ------------------------------------------------------------------------------------------
struct TextDev {
    hipStream_t st = nullptr;
    DevBuf len, roff, bsum, text, small, sort_a, sort_b, sort_tmp, snv_out, longs;
    void *gz = nullptr, *gz2 = nullptr;                   // gzip scratch; the second one: groups of files compressed while the group before is written
    void *pin = nullptr; size_t pin_cap = 0;              // descriptors up, totals down
    void *pin_text = nullptr; size_t pin_text_cap = 0;    // the text of files written without gzip
    int open(pav_ctx *ctx) {
        (void)ctx;
        if (!st) W_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        return PAV_OK;
    }
    void release(pav_ctx *ctx) {
        for (DevBuf *b : {&len, &roff, &bsum, &text, &small, &sort_a, &sort_b, &sort_tmp, &snv_out, &longs}) b->release();
        gz_release_slot(ctx, &gz);
        gz_release_slot(ctx, &gz2);
        if (pin) (void)hipHostFree(pin);
        if (pin_text) (void)hipHostFree(pin_text);
        if (st) (void)hipStreamDestroy(st);
        pin = pin_text = nullptr; pin_cap = pin_text_cap = 0; st = nullptr;
    }
};
struct TextDevState { TextDev cigar, density; };

TextDevState *tstate(pav_ctx *ctx) {
    std::call_once(ctx->textdev_once, [ctx] { ctx->textdev = new TextDevState(); });   // (the CIGAR-table and density-table writer threads)
    return static_cast<TextDevState *>(ctx->textdev);
}

int pin_grow(void *&p, size_t &cap, size_t bytes) {
    if (bytes <= cap) return PAV_OK;
    if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
    const size_t want = bytes + bytes / 4 + 4096;
    W_HIP(hipHostMalloc(&p, want, hipHostMallocDefault));
    cap = want;
    return PAV_OK;
}
int pin_reserve(pav_ctx *ctx, TextDev &D, size_t bytes) { (void)ctx; return pin_grow(D.pin, D.pin_cap, bytes); }

// roff[0 .. n] from len[0 .. n) on D.st
int scan_lengths(pav_ctx *ctx, TextDev &D, uint64_t n) {
    (void)ctx;
    W_HIP(D.roff.reserve(8 * (n + 2)));
    W_HIP(D.bsum.reserve(8 * ((size_t)(n / SCAN_TILE) + 4)));
    return scan_u32_to_u64(D.st, D.len.as<uint32_t>(), n, D.bsum.as<uint64_t>(), D.roff.as<uint64_t>());
}

double wall_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

bool ends_gz(const std::string &p) { return p.size() > 3 && p.compare(p.size() - 3, 3, ".gz") == 0; }

int write_file(const std::string &path, const uint8_t *data, uint64_t n) {
    FILE *fh = fopen(path.c_str(), "wb");
    if (!fh) return fail(nullptr, PAV_E_ARG, "table writer: cannot open %s", path.c_str());
    const bool ok = n == 0 || fwrite(data, 1, n, fh) == n;
    if (fclose(fh) != 0 || !ok) return fail(nullptr, PAV_E_ARG, "table writer: short write to %s", path.c_str());
    return PAV_OK;
}

// The files of `files` (text resident at D.text): gzip on the device for names ending in ".gz", the text itself otherwise.
int emit_files(pav_ctx *ctx, TextDev &D, uint64_t text_alloc, const std::vector<GzFile> &files, const std::vector<std::string> &paths, int level) {
    std::vector<GzFile> gz; std::vector<size_t> gz_ix;
    for (size_t f = 0; f < files.size(); ++f) if (ends_gz(paths[f])) { gz.push_back(files[f]); gz_ix.push_back(f); }
    if (!gz.empty()) {
        if (getenv("PAV_TIMING")) W_HIP(hipStreamSynchronize(D.st));    // (so that the gzip's own time is what gz_files prints)
        // Many files (the density tables of a haplotype: a hundred, 0.85 GB of text): in two groups (PAV_WRITER_GROUPS; a group's compression is whole rounds of the persistent waves: more groups, more part-filled rounds), the files of a group written by
        // a thread of their own while the next group is compressed (two scratch sets take turns).  One file, or little text: one call.
        uint64_t total = 0;
        for (const GzFile &g : gz) total += g.text_len;
        std::vector<size_t> cut{0};
        static const uint64_t groups = [] { const char *e = getenv("PAV_WRITER_GROUPS"); const int v = e ? atoi(e) : 0; return (uint64_t)(v > 0 ? v : 2); }();
        if (gz.size() >= 8 && total >= (64ull << 20) && groups > 1) {
            uint64_t acc = 0;
            for (size_t q = 0; q < gz.size(); ++q) {
                acc += gz[q].text_len;
                if (acc >= total / groups * cut.size() && cut.size() < groups && q + 1 < gz.size()) cut.push_back(q + 1);
            }
        }
        cut.push_back(gz.size());
        std::thread writer[2];
        int rc_w[2] = {PAV_OK, PAV_OK};
        std::string err_w[2];
        void **slot[2] = {&D.gz, &D.gz2};
        const double t0 = wall_now();
        int rc = PAV_OK;
        for (size_t g = 0; g + 1 < cut.size() && rc == PAV_OK; ++g) {
            const int s2 = (int)(g & 1);
            if (writer[s2].joinable()) writer[s2].join();          // the group that used this scratch before is on disk
            if (rc_w[s2] != PAV_OK) break;
            const std::vector<GzFile> part(gz.begin() + (long)cut[g], gz.begin() + (long)cut[g + 1]);
            GzOut out;
            rc = gz_files(ctx, slot[s2], D.st, D.text.as<uint8_t>(), text_alloc, part, level, out);
            if (rc != PAV_OK) break;
            const size_t a = cut[g];
            writer[s2] = std::thread([&, s2, a, out] {
                // (a hundred files of 3 MB each: four threads take every fourth - the page cache takes 9 GB/s from one)
                const size_t n_files = out.off.size(), n_thr = n_files >= 8 ? 4 : 1;
                std::vector<int> rc_t(n_thr, PAV_OK);
                std::vector<std::string> err_t(n_thr);
                auto part = [&](size_t k) {
                    for (size_t q = k; q < n_files && rc_t[k] == PAV_OK; q += n_thr) rc_t[k] = write_file(paths[gz_ix[a + q]], out.host + out.off[q], out.len[q]);
                    if (rc_t[k] != PAV_OK) err_t[k] = pav_last_error(nullptr);
                };
                std::vector<std::thread> sub;
                for (size_t k = 1; k < n_thr; ++k) sub.emplace_back(part, k);
                part(0);
                for (std::thread &t : sub) t.join();
                for (size_t k = 0; k < n_thr; ++k) if (rc_t[k] != PAV_OK && rc_w[s2] == PAV_OK) { rc_w[s2] = rc_t[k]; err_w[s2] = err_t[k]; }
            });
        }
        for (std::thread &w : writer) if (w.joinable()) w.join();
        if (rc != PAV_OK) return rc;
        for (int s2 = 0; s2 < 2; ++s2) if (rc_w[s2] != PAV_OK) return fail(nullptr, rc_w[s2], "%s", err_w[s2].c_str());
        if (getenv("PAV_TIMING")) fprintf(stderr, "[pav timing] %zu table files compressed and written in %.1f ms (%zu group(s))\n", gz.size(), (wall_now() - t0) * 1e3, cut.size() - 1);
    }
    for (size_t f = 0; f < files.size(); ++f) {
        if (ends_gz(paths[f])) continue;
        int rc = pin_grow(D.pin_text, D.pin_text_cap, files[f].text_len + 64);
        if (rc != PAV_OK) return rc;
        if (files[f].text_len) W_HIP(hipMemcpyAsync(D.pin_text, D.text.as<uint8_t>() + files[f].text_off, files[f].text_len, hipMemcpyDeviceToHost, D.st));
        W_HIP(hipStreamSynchronize(D.st));
        rc = write_file(paths[f], static_cast<const uint8_t *>(D.pin_text), files[f].text_len);
        if (rc != PAV_OK) return rc;
    }
    return PAV_OK;
}

uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

struct NameBlob { std::vector<uint8_t> bytes; std::vector<uint32_t> off; };
NameBlob pack_names(const std::vector<std::string> &names) {
    NameBlob b; b.off.reserve(names.size() + 1);
    for (const std::string &s : names) { b.off.push_back((uint32_t)b.bytes.size()); b.bytes.insert(b.bytes.end(), s.begin(), s.end()); }
    b.off.push_back((uint32_t)b.bytes.size());
    return b;
}

}  // namespace

bool device_writer_enabled() {
    const char *e = getenv("PAV_WRITER");
    return !(e && std::string(e) == "host");
}

void textdev_release(pav_ctx *ctx) {
    if (!ctx || !ctx->textdev) return;
    TextDevState *T = static_cast<TextDevState *>(ctx->textdev);
    (void)hipSetDevice(ctx->device);
    T->cigar.release(ctx); T->density.release(ctx);
    delete T;
    ctx->textdev = nullptr;
}

#define TD_CHECK(call) do { const int rc__ = (call); if (rc__ != PAV_OK) return rc__; } while (0)

static int text_cigar_tables_impl(pav_ctx *ctx, CigarTextJob &J) {
    TextDev &D = tstate(ctx)->cigar;
    W_HIP(hipSetDevice(ctx->device));
    TD_CHECK(D.open(ctx));
    hipStream_t st = D.st;
    if (J.ready) W_HIP(hipStreamWaitEvent(st, J.ready, 0));
    const uint32_t n_aln = (uint32_t)J.align_index.size(), n_ref = (uint32_t)J.rnames.size();
    const uint64_t n_snv = J.have_snv ? J.n_snv : 0, n_ind = J.have_insdel ? J.n_ind : 0;
    const bool merged = !J.batch8.empty();
    // ---- the small tables: names, ranks, INDEX, trim coordinates, CALL_BATCH, HAP, the two header lines -----------------------
    const NameBlob rn = pack_names(J.rnames), tn = pack_names(J.tnames);
    std::string h_snv = "#CHROM\tPOS\tEND\tID\tSVTYPE\tSVLEN\tREF\tALT\tHAP\tQRY_REGION\tQRY_STRAND\tCI\tALIGN_INDEX\tCALL_SOURCE";
    std::string h_ind = "#CHROM\tPOS\tEND\tID\tSVTYPE\tSVLEN\tHAP\tQRY_REGION\tQRY_STRAND\tCI\tALIGN_INDEX\tLEFT_SHIFT\tHOM_REF\tHOM_TIG\tCALL_SOURCE\tSEQ";
    const bool filt = J.filter;
    h_snv += filt ? "\tFILTER\n" : "\n"; h_ind += filt ? "\tFILTER\n" : "\n";
    std::vector<uint8_t> blk;
    auto put = [&](const void *p, size_t bytes) {                      // p == nullptr: `bytes` zeros
        const size_t at = align_up(blk.size(), 16);
        blk.resize(at + bytes, 0);
        if (bytes && p) memcpy(blk.data() + at, p, bytes);
        return at;
    };
    const size_t o_rblob = put(rn.bytes.data(), rn.bytes.size()), o_roff = put(rn.off.data(), 4 * rn.off.size());
    const size_t o_tblob = put(tn.bytes.data(), tn.bytes.size()), o_toff = put(tn.off.data(), 4 * tn.off.size());
    const size_t o_rank = put(J.rank.data(), 2 * (size_t)n_ref), o_ai = put(J.align_index.data(), 8 * (size_t)n_aln);
    const size_t o_tp = put(J.trim_pos.data(), 8 * J.trim_pos.size()), o_te = put(J.trim_end.data(), 8 * J.trim_end.size());
    const size_t o_batch = put(J.batch8.data(), J.batch8.size()), o_hap = put(J.hap.data(), J.hap.size());
    const size_t o_hs = put(h_snv.data(), h_snv.size()), o_hi = put(h_ind.data(), h_ind.size());
    const size_t o_foff = put(nullptr, 16), o_nlong = put(nullptr, 16);
    W_HIP(D.small.reserve(blk.size() + 64));
    TD_CHECK(pin_reserve(ctx, D, blk.size() + 256));
    memcpy(D.pin, blk.data(), blk.size());
    W_HIP(hipMemcpyAsync(D.small.p, D.pin, blk.size(), hipMemcpyHostToDevice, st));
    uint8_t *sm = D.small.as<uint8_t>();
    const uint16_t *d_rank = reinterpret_cast<const uint16_t *>(sm + o_rank);
    const uint8_t *d_batch = merged ? sm + o_batch : nullptr;
    const long long *d_tp = filt ? reinterpret_cast<const long long *>(sm + o_tp) : nullptr, *d_te = filt ? reinterpret_cast<const long long *>(sm + o_te) : nullptr;
    CigarRows C{};
    C.aln = ctx->d_aln.as<pav_aln>();
    C.ref = Names{sm + o_rblob, reinterpret_cast<const uint32_t *>(sm + o_roff)};
    C.tig = Names{sm + o_tblob, reinterpret_cast<const uint32_t *>(sm + o_toff)};
    C.hap = sm + o_hap; C.hap_len = (uint32_t)J.hap.size();
    C.align_index = reinterpret_cast<const long long *>(sm + o_ai); C.trim_pos = d_tp; C.trim_end = d_te;
    C.ind = ctx->d_indel.as<pav_indel>();

    // ---- order ------------------------------------------------------------------------------------------------------------
    const uint64_t n_max = std::max(n_snv, n_ind);
    size_t tmp_bytes = 0;
    if (n_max) {
        unsigned long long *k0 = nullptr; uint32_t *v0 = nullptr;
        W_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, k0, k0, v0, v0, (size_t)n_max, 0, 64, st));
        W_HIP(D.sort_a.reserve(12 * n_max + 64));
        W_HIP(D.sort_b.reserve(12 * n_max + 64));
        W_HIP(D.sort_tmp.reserve(tmp_bytes + 64));
    }
    unsigned long long *ka = D.sort_a.as<unsigned long long>(), *kb = D.sort_b.as<unsigned long long>();
    uint32_t *va = reinterpret_cast<uint32_t *>(ka + n_max), *vb = reinterpret_cast<uint32_t *>(kb + n_max);
    if (n_snv) {
        W_HIP(D.snv_out.reserve(sizeof(SnvOut) * n_snv));
        W_LAUNCH(st, k_snv_keys, (uint32_t)((n_snv + 255) / 256), 256, 0, ctx->d_snv.as<pav_snv>(), n_snv, C.aln, d_rank, d_batch, ka, va);
        W_HIP(rocprim::radix_sort_pairs(D.sort_tmp.p, tmp_bytes, ka, kb, va, vb, (size_t)n_snv, 0, 64, st));
        W_LAUNCH(st, k_snv_gather, (uint32_t)((n_snv + 255) / 256), 256, 0, ctx->d_snv.as<pav_snv>(), vb, n_snv, d_tp, d_te,
                      D.snv_out.as<SnvOut>());
    }
    C.snv = D.snv_out.as<SnvOut>();

    // ---- SNV table: lengths, offsets ----------------------------------------------------------------------------------------------
    uint64_t snv_text = h_snv.size(), ind_text = h_ind.size();
    uint64_t *h_tot = reinterpret_cast<uint64_t *>(static_cast<uint8_t *>(D.pin) + align_up(blk.size(), 64));
    W_HIP(D.len.reserve(4 * (n_max + 8)));
    uint64_t snv_rows_bytes = 0;
    if (J.have_snv) {
        if (n_snv) W_LAUNCH(st, (k_cigar_rowlen<ROWS_SNV>), (uint32_t)((n_snv + 255) / 256), 256, 0, C, n_snv, D.len.as<uint32_t>());
        TD_CHECK(scan_lengths(ctx, D, n_snv));
        W_HIP(hipMemcpyAsync(h_tot, D.roff.as<uint64_t>() + n_snv, 8, hipMemcpyDeviceToHost, st));
        W_HIP(hipStreamSynchronize(st));
        snv_rows_bytes = h_tot[0];
        snv_text += snv_rows_bytes;
    }
    // (the two tables share the writer's scratch - lengths, offsets, text arena: the SNV file is finished before the INS / DEL rows
    //  are sorted and laid out)
    std::vector<GzFile> files; std::vector<std::string> paths;
    uint64_t text_alloc = 0;
    const uint64_t snv_off = 0;
    if (J.have_snv) {
        text_alloc = align_up(snv_text, GZ_TEXT_ALIGN) + GZ_TEXT_PAD;
        W_HIP(D.text.reserve(text_alloc));
        W_HIP(hipMemsetAsync(D.text.as<uint8_t>() + snv_text, 0, text_alloc - snv_text, st));
        W_HIP(hipMemsetAsync(sm + o_foff, 0, 16, st));
        W_LAUNCH(st, k_headers, 1, 64, 0, D.text.as<uint8_t>(), reinterpret_cast<const uint64_t *>(sm + o_foff), 1u, sm + o_hs, (uint32_t)h_snv.size());
        if (n_snv) W_LAUNCH(st, (k_cigar_rowtext<ROWS_SNV>), (uint32_t)((n_snv + 255) / 256), 256, 0, C, n_snv, D.roff.as<uint64_t>(),
                                 snv_off + h_snv.size(), D.text.as<uint8_t>(), (const uint8_t *)nullptr, (LongCopy *)nullptr, (uint32_t *)nullptr);
        files.push_back(GzFile{snv_off, snv_text}); paths.push_back(J.snv_path);
        TD_CHECK(emit_files(ctx, D, text_alloc, files, paths, J.level));
        files.clear(); paths.clear();
    }
    // ---- INS / DEL table -------------------------------------------------------------------------------------------------------------
    if (J.have_insdel) {
        const uint32_t *order = nullptr;
        if (n_ind) {
            const uint32_t grid = (uint32_t)((n_ind + 255) / 256);
            unsigned long long *kin = ka, *kout = kb; uint32_t *vin = va, *vout = vb;
            for (int pass = 0; pass < 3; ++pass) {
                W_LAUNCH(st, k_indel_keys, grid, 256, 0, C.ind, order, n_ind, pass, C.aln, d_rank, d_batch, kin, vin);
                const int bits = pass == 0 ? 45 : (pass == 1 ? 32 : 48);
                W_HIP(rocprim::radix_sort_pairs(D.sort_tmp.p, tmp_bytes, kin, kout, vin, vout, (size_t)n_ind, 0, bits, st));
                order = vout;
                std::swap(kin, kout); std::swap(vin, vout);                  // the next pass takes the order from (now) vin and rewrites it in
                                                                             // place (vals[i] = order[i], same lane), keys beside it
            }
        }
        C.ind_order = order;
        if (n_ind) W_LAUNCH(st, (k_cigar_rowlen<ROWS_INDEL>), (uint32_t)((n_ind + 255) / 256), 256, 0, C, n_ind, D.len.as<uint32_t>());
        TD_CHECK(scan_lengths(ctx, D, n_ind));
        W_HIP(hipMemcpyAsync(h_tot, D.roff.as<uint64_t>() + n_ind, 8, hipMemcpyDeviceToHost, st));
        W_HIP(hipStreamSynchronize(st));
        ind_text += h_tot[0];
        text_alloc = align_up(ind_text, GZ_TEXT_ALIGN) + GZ_TEXT_PAD;
        W_HIP(D.text.reserve(text_alloc));
        W_HIP(D.longs.reserve(sizeof(LongCopy) * (n_ind + 1)));
        W_HIP(hipMemsetAsync(D.text.as<uint8_t>() + ind_text, 0, text_alloc - ind_text, st));
        W_HIP(hipMemsetAsync(sm + o_foff, 0, 16, st));
        W_HIP(hipMemsetAsync(sm + o_nlong, 0, 16, st));
        W_LAUNCH(st, k_headers, 1, 64, 0, D.text.as<uint8_t>(), reinterpret_cast<const uint64_t *>(sm + o_foff), 1u, sm + o_hi, (uint32_t)h_ind.size());
        if (n_ind) {
            W_LAUNCH(st, (k_cigar_rowtext<ROWS_INDEL>), (uint32_t)((n_ind + 255) / 256), 256, 0, C, n_ind, D.roff.as<uint64_t>(),
                          (uint64_t)h_ind.size(), D.text.as<uint8_t>(), ctx->d_seqblob.as<uint8_t>(), D.longs.as<LongCopy>(), reinterpret_cast<uint32_t *>(sm + o_nlong));
            W_LAUNCH(st, k_long_copies, 2048, 64, 0, D.text.as<uint8_t>(), ctx->d_seqblob.as<uint8_t>(), D.longs.as<LongCopy>(),
                          reinterpret_cast<const uint32_t *>(sm + o_nlong));
        }
        files.push_back(GzFile{0, ind_text}); paths.push_back(J.insdel_path);
        TD_CHECK(emit_files(ctx, D, text_alloc, files, paths, J.level));
    }
    return PAV_OK;
}

int text_cigar_tables(pav_ctx *ctx, CigarTextJob &job, std::string &err) {
    const double t0 = wall_now();
    const int rc = text_cigar_tables_impl(ctx, job);
    if (getenv("PAV_TIMING")) fprintf(stderr, "[pav timing] device writer, SNV + INS / DEL tables: %.1f ms\n", (wall_now() - t0) * 1e3);
    if (rc != PAV_OK) err = pav_last_error(nullptr);
    return rc;
}

int text_density_tables(pav_ctx *ctx, const std::vector<DenTableDev> &tables, const std::vector<std::string> &paths, int level) {
    if (tables.empty()) return PAV_OK;
    TextDev &D = tstate(ctx)->density;
    W_HIP(hipSetDevice(ctx->device));
    TD_CHECK(D.open(ctx));
    hipStream_t st = D.st;
    const uint32_t n_tab = (uint32_t)tables.size();
    const double t_in = wall_now();
    static const std::string header = "INDEX\tSTATE_MER\tSTATE\tKERN_FWD\tKERN_FWDREV\tKERN_REV\tKMER\tFLANK\tMATCH\n";
    std::vector<uint64_t> row0(n_tab + 1, 0);
    for (uint32_t t = 0; t < n_tab; ++t) row0[t + 1] = row0[t] + tables[t].n;
    const uint64_t n = row0[n_tab];
    // small block: descriptors | first rows | header | (later) file offsets, row bases
    const size_t o_tab = 0, o_row0 = align_up(sizeof(DenTableDev) * n_tab, 64), o_hdr = o_row0 + align_up(8 * ((size_t)n_tab + 1), 64);
    const size_t o_foff = o_hdr + align_up(header.size(), 64), o_fbase = o_foff + align_up(8 * (size_t)n_tab, 64), o_bound = o_fbase + align_up(8 * (size_t)n_tab, 64);
    const size_t small_bytes = o_bound + align_up(8 * ((size_t)n_tab + 1), 64);
    W_HIP(D.small.reserve(small_bytes + 64));
    TD_CHECK(pin_reserve(ctx, D, 2 * small_bytes + 256));
    uint8_t *hp = static_cast<uint8_t *>(D.pin);
    memcpy(hp + o_tab, tables.data(), sizeof(DenTableDev) * n_tab);
    memcpy(hp + o_row0, row0.data(), 8 * ((size_t)n_tab + 1));
    memcpy(hp + o_hdr, header.data(), header.size());
    W_HIP(hipMemcpyAsync(D.small.p, hp, o_foff, hipMemcpyHostToDevice, st));
    uint8_t *sm = D.small.as<uint8_t>();
    DenRows R{reinterpret_cast<const DenTableDev *>(sm + o_tab), reinterpret_cast<const uint64_t *>(sm + o_row0), n_tab};
    W_HIP(D.len.reserve(4 * (n + 8)));
    if (n) W_LAUNCH(st, k_den_rowlen, (uint32_t)((n + 255) / 256), 256, 0, R, n, D.len.as<uint32_t>());
    TD_CHECK(scan_lengths(ctx, D, n));
    // the text offset of every table's first row (and of the end): n_tab + 1 values of roff, picked by a strided copy
    uint64_t *h_bound = reinterpret_cast<uint64_t *>(hp + small_bytes + 64);
    for (uint32_t t = 0; t <= n_tab; ++t)
        W_HIP(hipMemcpyAsync(h_bound + t, D.roff.as<uint64_t>() + row0[t], 8, hipMemcpyDeviceToHost, st));
    W_HIP(hipStreamSynchronize(st));
    std::vector<GzFile> files(n_tab);
    std::vector<uint64_t> foff(n_tab), fbase(n_tab);
    uint64_t at = 0;
    for (uint32_t t = 0; t < n_tab; ++t) {
        at = align_up(at, GZ_TEXT_ALIGN);
        const uint64_t bytes = header.size() + (h_bound[t + 1] - h_bound[t]);
        files[t] = GzFile{at, bytes};
        foff[t] = at; fbase[t] = at + header.size() - h_bound[t];
        at += bytes;
    }
    const uint64_t text_alloc = align_up(at, GZ_TEXT_ALIGN) + GZ_TEXT_PAD;
    W_HIP(D.text.reserve(text_alloc));
    memcpy(hp + o_foff, foff.data(), 8 * (size_t)n_tab);
    memcpy(hp + o_fbase, fbase.data(), 8 * (size_t)n_tab);
    W_HIP(hipMemcpyAsync(sm + o_foff, hp + o_foff, o_bound - o_foff, hipMemcpyHostToDevice, st));
    W_HIP(hipMemsetAsync(D.text.p, 0, text_alloc, st));          // the gaps between the files and the pad behind them read as zeros
    W_LAUNCH(st, k_headers, n_tab, 64, 0, D.text.as<uint8_t>(), reinterpret_cast<const uint64_t *>(sm + o_foff), n_tab, sm + o_hdr,
                  (uint32_t)header.size());
    if (n) W_LAUNCH(st, k_den_rowtext, (uint32_t)((n + 255) / 256), 256, 0, R, n, D.roff.as<uint64_t>(),
                         reinterpret_cast<const uint64_t *>(sm + o_fbase), D.text.as<uint8_t>());
    const double t1 = wall_now();
    const int rce = emit_files(ctx, D, text_alloc, files, paths, level);
    if (getenv("PAV_TIMING")) fprintf(stderr, "[pav timing] device writer, %u density tables (%llu rows): lengths + offsets %.1f ms (incl. waiting for the stream), gzip + files %.1f ms\n",
                                      n_tab, (unsigned long long)n, (t1 - t_in) * 1e3, (wall_now() - t1) * 1e3);
    return rce;
}

}  // namespace pav
