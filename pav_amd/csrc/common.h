// Internal shared declarations of libpav_amd.so (gfx950 only).  Public ABI: include/pav_amd.h.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <atomic>
#include <mutex>
#include <memory>
#include <string>
#include <vector>

#include "../../include/pav_amd.h"

namespace pav {

constexpr int WAVE = 64;                 // CDNA wavefront
constexpr int DIRTY_SHIFT = 10;          // granularity of SeqView::dirty (1024 bases = what one wave packs per load)
constexpr uint64_t SEQ_ALIGN = 256;      // every record starts on a 256-base boundary of the arena

// ---- device blocks that outlive their context (ctx.hip) --------------------------------------------------------------------
// A haplotype per context - allocated, used, destroyed - is how Snakemake-style drivers and tools/bench_e2e.py use the library, and a
// context's large buffers are 25 GB (two sequence stores, the file text and token lists of the loaders, the writers' text).  Freed HBM
// is cleared by the driver before it is handed out again: every other context of a run waited 0.4 - 0.65 s in hipMalloc for the clearing
// of what the context before it had freed (`profiles/r05_e2e_soak.txt`).  Blocks of 32 MB and more therefore go back to a process-wide
// list per GPU instead of to the driver (at most PAV_DEVICE_POOL_GB = 24 GB and a quarter of the device's memory kept, the rest freed), and come from it.  dev_block_put waits for the device
// first, as hipFree does.  PAV_DEVICE_POOL=0: plain hipMalloc / hipFree.
hipError_t dev_block_get(void **p, size_t *cap, size_t want);
void dev_block_put(void *p, size_t cap);
size_t dev_pool_trim(int device);        // idle blocks of the device (< 0: all devices) back to the driver; bytes freed

// ---- grow-only device buffer ------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool view = false;                    // p points into another buffer's allocation (alias): never freed from here
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (view) { p = nullptr; cap = 0; view = false; }                     // outgrown: from here on an allocation of its own
        if (p) { dev_block_put(p, cap); p = nullptr; cap = 0; }
        return dev_block_get(&p, &cap, bytes + bytes / 8 + 256);
    }
    // At least `bytes`, and when it has to grow `bytes` without the margin (buffers that change hands must not outgrow each other in turn)
    hipError_t reserve_exact(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (view) { p = nullptr; cap = 0; view = false; }
        if (p) { dev_block_put(p, cap); p = nullptr; cap = 0; }
        return dev_block_get(&p, &cap, bytes);
    }
    // Make this buffer `bytes` of another allocation (several small buffers laid out in one arena travel in one copy and are
    // cleared by one fill).  An allocation of its own is given up.
    void alias(void *q, size_t bytes) {
        if (p && !view) dev_block_put(p, cap);
        p = q; cap = bytes; view = true;
    }
    void release() { if (p && !view) dev_block_put(p, cap); p = nullptr; cap = 0; view = false; }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Four ASCII bases (one dword) -> 8 bits of 2-bit codes (base 0 lowest) and 4 non-ACGT bits.
__device__ __forceinline__ void pack4(uint32_t x, uint32_t &codes8, uint32_t &bad4) {
    const uint32_t t = ((x >> 1) ^ (x >> 2)) & 0x03030303u;          // A0 C1 G2 T3, either case
    uint32_t c = t | (t >> 6);
    codes8 = (c | (c >> 12)) & 0xFFu;
    const uint32_t lo = t & 0x01010101u, hi = (t >> 1) & 0x01010101u, both = lo & hi;
    const uint32_t expect = 0x41414141u + lo * 2u + hi * 6u + both * 11u;   // 'A','C','G','T' for the code
    const uint32_t d = (x & 0xDFDFDFDFu) ^ expect;                    // non-zero byte <=> not ACGT/acgt
    uint32_t nz = ((((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d) & 0x80808080u) >> 7;
    bad4 = (nz | (nz >> 7) | (nz >> 14) | (nz >> 21)) & 0xFu;
}

// ---- sequence store (one per role) ------------------------------------------------------------------------
struct SeqView {                          // passed by value to kernels
    const uint8_t *ascii;                 // arena bytes
    const uint32_t *two;                  // 2-bit plane, 16 bases / word, base i at bits 2*(i&15)
    const uint32_t *mask;                 // non-ACGT plane, 32 bases / word, base i at bit (i&31)
    const uint8_t *dirty;                 // one byte per 2^DIRTY_SHIFT bases: non-zero when the block holds a non-ACGT base
    const uint64_t *off;                  // per record: first base in the arena (multiple of SEQ_ALIGN)
    const uint64_t *len;                  // per record: length
    uint32_t n;
    uint32_t packed;                      // 1: the three planes hold the whole arena; 0: only the spans packed on demand
};                                        //    (contigs, pav_seq_pack / pav_seq_load) - scattered readers then decode the ASCII

struct SeqStore {
    uint32_t n = 0;
    uint64_t arena = 0;                   // bases incl. padding
    uint64_t total = 0;                   // bases excl. padding
    std::vector<uint64_t> off, len;
    DevBuf d_ascii, d_two, d_mask, d_dirty, d_off, d_len;
    int device = -1;                      // where the planes live (shared stores: all users sit on this device)
    bool planes_full = false;             // a pack of the whole arena has been QUEUED since the ASCII last changed; it has run
                                          // only behind pack_event: every reader's stream waits for it (pav::wait_planes)
    hipEvent_t pack_event = nullptr;      // recorded behind the last full pack, on whichever context's stream ran it
    uint64_t pack_gen = 0;                // counts full packs; a context remembers the last one its main stream waited for
    SeqView view() const {
        return SeqView{d_ascii.as<uint8_t>(), d_two.as<uint32_t>(), d_mask.as<uint32_t>(), d_dirty.as<uint8_t>(),
                       d_off.as<uint64_t>(), d_len.as<uint64_t>(), n, planes_full ? 1u : 0u};
    }
    void release() {
        for (DevBuf *b : {&d_ascii, &d_two, &d_mask, &d_dirty, &d_off, &d_len}) b->release();
        if (pack_event) { (void)hipEventDestroy(pack_event); pack_event = nullptr; }
        n = 0; arena = total = 0; planes_full = false;
    }
    SeqStore() = default;
    SeqStore(const SeqStore &) = delete;
    SeqStore &operator=(const SeqStore &) = delete;
    ~SeqStore() { if (device >= 0) (void)hipSetDevice(device); release(); }
};

// The two stores of a context.  A store can be shared between contexts on the same GPU (pav_seq_share): a cohort's
// haplotypes - one context each, driven from their own host threads - read ONE resident reference.
struct SeqSlots {
    std::shared_ptr<SeqStore> p[2];
    SeqSlots() { p[0] = std::make_shared<SeqStore>(); p[1] = std::make_shared<SeqStore>(); }
    SeqStore &operator[](int role) { return *p[role]; }
    const SeqStore &operator[](int role) const { return *p[role]; }
};

// ---- profiling --------------------------------------------------------------------------------------------
struct ProfEntry { std::string name; uint64_t launches = 0; double ms = 0.0; };
struct ProfPending { int entry; hipEvent_t a, b; };

}  // namespace pav

namespace pav {
// The decisions of a scan round on the device (density.hip k_round_decide; invscan.cpp): what the scan driver knows of a job's
// region when the density batch is queued, and the hook through which the batch - behind its last kernel and in front of its one
// read-back - derives every job's next lift queries (pavlib/inv.py:297-351, 378-406) and lets the driver queue their lifts.
struct RoundJobIn { int32_t ref_chrom, tig_chrom, tig_rev, expansion_count; int64_t ref_pos, ref_end, tig_pos, chrom_len; };
struct RoundHook {
    const RoundJobIn *d_in = nullptr;     // [n_jobs] on the device
    void *d_queries = nullptr;            // LiftQuery[4 n_jobs] on the device: a job's breakpoint queries, or its next region's two ends
    uint32_t n_jobs = 0;
    int32_t min_exp_count = 1, k = 31;
    bool ran = false;                     // the batch took the device-planned path and the queries were derived
    std::function<int()> after;           // queues the lifts of the queries and their way to the host on the context's stream
};
}  // namespace pav

struct pav_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;        // side stream: contig re-pack overlaps the tokenizer / walk kernels
    hipStream_t stream3 = nullptr;        // copy stream: call tables of the inversion scan travel to the host behind the scan
    hipEvent_t tables_done = nullptr;     // recorded on stream3 after the last queued table copy of the current scan
    bool tables_pending = false;
    hipEvent_t tables_done_prev = nullptr;   // same for the scan before it: its tables live in the other pinned arena, so a new
    bool tables_pending_prev = false;        // scan does not wait for them (the two are swapped when a scan starts)
    std::function<void()> den_overlap;    // set by the inversion-scan driver around pav_density_batch: host work that does not depend on
                                          // the batch (log texts), run once while the batch's kernels are executing
    pav::RoundHook *den_round = nullptr;  // set by the inversion-scan driver around pav_density_batch (see RoundHook)
    bool den_scan_only = false;           // set by the inversion-scan driver around pav_density_batch: only run lists and the tables of
                                          // regions that can become calls will be read (density.hip, fwd_only)
    hipEvent_t hom_done = nullptr;        // pav_cigar_call: recorded behind the homology scans on stream2 (wait_homology)
    std::atomic<bool> hom_pending{false};   // (atomic: read and cleared by both loader threads, see err_mu)
    hipEvent_t snv_ready = nullptr, snv_done = nullptr;   // pav_cigar_call: the SNV rows are written on stream2 (behind the pack),
                                                          // next to the homology scans of the main stream
    hipEvent_t pack_done[2] = {nullptr, nullptr};   // pav_seq_pack: orders the side stream's pack behind the main stream
    bool pack_pending[2] = {false, false};
    uint64_t seen_pack_gen[2] = {0, 0};   // SeqStore::pack_gen of the last full pack the main stream waits behind (per role)
    std::string err;
    std::mutex err_mu;                    // fail(): the two roles of one context may be loaded from two threads at the same time
    std::mutex prof_mu;                   // ... and both launch kernels through PAV_LAUNCH: the event bookkeeping below is shared
    std::once_flag invscan_once, textdev_once;   // lazily made state that both loader / writer threads reach first (invscan.cpp, textdev.hip)
    char dev_name[256] = {0};
    int n_cu = 0;

    pav::SeqSlots seq;

    // CIGAR state
    uint32_t n_aln = 0;
    uint64_t text_bytes = 0;
    std::atomic<bool> cigar_loaded{false}, cigar_called{false};   // (atomic: both loader threads clear them)
    pav_cigar_counts counts{};
    pav_cigar_err cigar_err{};
    uint64_t *h_status = nullptr;         // pinned host words for the small device-to-host readbacks of pav_cigar_call
    pav::DevBuf d_aln, d_text, d_text_off, d_ops, d_op_off, d_chunk, d_chunk2, d_rowbase, d_totals;
    pav::DevBuf d_snv, d_indel, d_seqblob, d_tmp;
    double kde_work[3] = {0, 0, 0};       // pav_kde_work: evaluation points, (point, run) pairs, (point, data point) pairs
    pav::DevBuf d_spans;                  // need_planes_spans: block runs of the spans being packed
    std::vector<uint8_t> span_host;       // ... and their host copy (alive until the upload has run)
    pav::DevBuf ix_text, ix_off, ix_pos, ix_ops, ix_op_off, ix_chunk, ix_chunk2, ix_rowbase, ix_begin, ix_err;   // pav_align_index
    uint64_t n_ops = 0;
    uint64_t ix_n_ops = 0; uint32_t ix_n_aln = 0;

    // density state lives in density.hip (opaque here)
    void *density = nullptr;
    void *invscan = nullptr;              // native scan driver state (invscan.cpp)
    void *flag = nullptr;                 // flagging scratch + results (flag.hip)
    void *trim = nullptr;                 // alignment trimming state (trim.cpp)
    void *table_writer = nullptr;         // a table write in two halves (tables.hip: pav_cigar_write_tables_begin / _end)
    hipEvent_t writer_ready = nullptr;    // recorded behind the calls when a device table write begins (tables.hip)
    void *upload = nullptr;               // pinned staging rings of the large host-to-device uploads (ctx.hip: staged_upload)
    void *fa_dev = nullptr;               // device FASTA loader scratch (fastadev.hip)
    void *gz = nullptr;                   // device gzip scratch (deflate.hip)
    void *textdev = nullptr;              // device table text scratch (textdev.hip)

    // profiling
    bool prof_on = false;
    std::vector<pav::ProfEntry> prof;
    std::vector<pav::ProfPending> prof_pending;
    std::vector<hipEvent_t> ev_pool;
};

namespace pav {

extern thread_local std::string g_err;   // pav_last_error(NULL)

int fail(pav_ctx *ctx, int code, const char *fmt, ...);
void table_writer_release(pav_ctx *ctx);                             // tables.hip: joins a pending writer thread
void table_writer_quiesce(pav_ctx *ctx);                             // tables.hip: waits for a write that still reads the resident records

// Every host wait for a stream in the library goes through stream_wait (the macro below routes the runtime's name to it): how a host
// thread waits decides what a lane costs.  The runtime's own wait spins on the completion signal - with more lanes than cores the
// spinning threads take the cores from the threads that have work (six lanes on two cores: 1.60 Tbp/s against 2.36 on sixteen).
//   PAV_WAIT=spin   the runtime's hipStreamSynchronize
//   PAV_WAIT=yield  an event behind the stream's work, polled with sched_yield() between the polls: as quick as spinning while
//                   cores are free, and a waiting lane gives its core to any thread that can run
//   PAV_WAIT=block  the same event, created with hipEventBlockingSync: the thread sleeps until the interrupt
//   Default: yield (measured, six lanes: 2.16 Tbp/s on two cores, 2.40 on sixteen; spin 1.62 / 2.43; block 1.57 / 2.40; one lane alike).
hipError_t stream_wait(hipStream_t st);
hipError_t event_wait(hipEvent_t ev);
#define hipStreamSynchronize(st) ::pav::stream_wait(st)

#define PAV_HIP(ctx, call)                                                                         \
    do {                                                                                           \
        hipError_t e__ = (call);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return pav::fail((ctx), PAV_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                             __FILE__, __LINE__);                                                  \
    } while (0)

// Profiled launch: records a HIP event pair around the launch on ctx->stream when profiling is on.
int prof_begin(pav_ctx *ctx, const char *name, hipStream_t st = nullptr);
void prof_end(pav_ctx *ctx, int token, hipStream_t st = nullptr);
int wait_tables(pav_ctx *ctx);            // host waits until every queued call-table copy has landed
int wait_homology(pav_ctx *ctx);          // make ctx->stream wait for the homology scans of the last pav_cigar_call (stream2)
int wait_planes(pav_ctx *ctx);            // make ctx->stream wait for any pack still running on stream2
// Contig planes are packed on demand (ctx.hip "lazy contig pack"): a consumer that streams the whole arena calls
// need_planes_full, one that reads spans [abs, abs + len) calls need_planes_spans; both order the pack before what the caller
// launches next on ctx->stream.  The reference store is always packed in full.
int need_planes_full(pav_ctx *ctx, int role);
struct PlaneSpan { uint64_t abs, len; };
int need_planes_spans(pav_ctx *ctx, int role, const std::vector<PlaneSpan> &spans);
int prof_flush(pav_ctx *ctx);

// PAV_SYNC_EACH=1 (debugging): wait for every kernel and name it on stderr, so that a memory fault points at its launch
bool sync_each();
#define PAV_LAUNCH_ON(ctx, st, name, kernel, grid, block, shmem, ...)                               \
    do {                                                                                           \
        int tok__ = pav::prof_begin((ctx), name, (st));                                            \
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), (shmem), (st), __VA_ARGS__);           \
        pav::prof_end((ctx), tok__, (st));                                                         \
        PAV_HIP((ctx), hipGetLastError());                                                         \
        if (pav::sync_each()) {                                                                    \
            fprintf(stderr, "[pav launch] %s grid %u x %u ... ", name, (unsigned)(grid), (unsigned)(block)); fflush(stderr); \
            PAV_HIP((ctx), hipStreamSynchronize(st));                                              \
            fprintf(stderr, "done\n");                                                             \
        }                                                                                          \
    } while (0)

#define PAV_LAUNCH(ctx, name, kernel, grid, block, shmem, ...) PAV_LAUNCH_ON(ctx, (ctx)->stream, name, kernel, grid, block, shmem, __VA_ARGS__)

// ---- density.hip internals used by the native scan driver ---------------------------------------------------------
struct CallFetch {              // one call of the batch still resident on the device
    uint32_t job, n, ref_id;
    uint64_t ref_up_pos, ref_up_end, ref_dn_pos, ref_dn_end;
    int64_t base, tig_up_pos, tig_up_end, tig_dn_pos, tig_dn_end;
};
// The packed call tables of one scan round on the device (+ the flank k-mer sets and descriptors in front of / behind them)
// and the copies that bring them to the host block of the round.
struct CallStage {
    DevBuf buf;                           // the columns: packed rows of the round's calls, or - a round whose calls make up most of the
                                          // batch - the batch's own column block, handed over without a copy (density_fetch_calls)
    DevBuf aux;                           // flank k-mer sets and descriptors
    std::vector<uint8_t> desc_host;
    struct Copy { const void *src; size_t dst_off; size_t bytes; };   // dst_off: into the round's host block
    Copy copies[2];
    int n_copies = 0;
    uint64_t rows = 0;                    // rows per column of the block (packed: the calls' rows; handed over: the batch's rows); host
    void *host = nullptr;                 // block = whole-round columns K0 | K1 | K2 | KMER | INDEX | STATE_MER | STATE | FLANK | MATCH
                                          // (40 bytes per row), bound by the caller before stage_copy
    std::vector<uint64_t> row0;           // first row of every call in the columns, in the order of the calls given (density_fetch_calls)
    struct Entry { uint32_t owner; uint32_t n; uint64_t row0; bool has_k1; };   // the calls of the round
    std::vector<Entry> entries;
    void release() { buf.release(); aux.release(); }
};
// Density tables + FLANK / MATCH of all calls of the last batch, put into `stage` in the host block's column order; fills
// stage.rows and stage.row0.  k1_rows (packed rounds): rows of the leading calls whose KERN_FWDREV column is wanted; the column of
// the calls behind them (no FWDREV k-mers: all zeros, scripts/density.py:313-323) is not sent over PCIe.  = all rows to copy
// everything.  The copies that bring the block to stage.host are left in stage.copies: the tables stay in HBM until a reader asks
// for them (stage_copy), or the caller binds stage.host and calls density_copy_now.
int density_fetch_calls(pav_ctx *ctx, const std::vector<CallFetch> &calls, uint64_t k1_rows, CallStage &stage);
int density_copy_now(pav_ctx *ctx, CallStage &stage);     // the copy stream waits for the stage's kernels, then stage_copy
int stage_copy(pav_ctx *ctx, CallStage &stage);

// ---- device helpers shared by kernels ---------------------------------------------------------------------

// Base code at oriented position p of record (off,len): 0..3 = A,C,G,T (complemented when rev), 4 = non-ACGT.
__device__ __forceinline__ uint32_t base_at(const uint32_t *__restrict__ two, const uint32_t *__restrict__ mask,
                                            uint64_t off, uint64_t len, int rev, int64_t p) {
    const uint64_t a = off + (uint64_t)(rev ? (int64_t)len - 1 - p : p);
    const uint32_t m = (mask[a >> 5] >> (a & 31)) & 1u;
    uint32_t c = (two[a >> 4] >> ((a & 15) * 2)) & 3u;
    if (rev) c ^= 3u;
    return m ? 4u : c;
}

// IUPAC-aware, case-preserving complement of one ASCII base (Bio.Seq.reverse_complement semantics).
__device__ __forceinline__ uint8_t comp_ascii(uint8_t c) {
    switch (c) {
        case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
        case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a';
        case 'R': return 'Y'; case 'Y': return 'R'; case 'K': return 'M'; case 'M': return 'K';
        case 'r': return 'y'; case 'y': return 'r'; case 'k': return 'm'; case 'm': return 'k';
        case 'B': return 'V'; case 'V': return 'B'; case 'D': return 'H'; case 'H': return 'D';
        case 'b': return 'v'; case 'v': return 'b'; case 'd': return 'h'; case 'h': return 'd';
        case 'U': return 'A'; case 'u': return 'a';
        default: return c;                     // S, W, N and anything else map to themselves
    }
}

__device__ __forceinline__ uint8_t ascii_at(const uint8_t *__restrict__ ascii, uint64_t off, uint64_t len, int rev,
                                            int64_t p) {
    if (rev) return comp_ascii(ascii[off + (uint64_t)((int64_t)len - 1 - p)]);
    return ascii[off + (uint64_t)p];
}

}  // namespace pav
