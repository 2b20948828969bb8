// Context, sequence store (ASCII arena -> 2-bit + non-ACGT planes) and HIP-event profiling.
// gfx950 only; public ABI in include/pav_amd.h.
#include "common.h"
#include "devgz.h"
#include "upload.h"
#include "textdev.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

namespace pav {

thread_local std::string g_err;

int fail(pav_ctx *ctx, int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) { std::lock_guard<std::mutex> lk(ctx->err_mu); ctx->err = buf; }
    g_err = buf;
    return code;
}

bool sync_each() { const char *e = getenv("PAV_SYNC_EACH"); return e && *e == '1'; }

// ---- host waits (common.h: stream_wait) ---------------------------------------------------------------------
namespace {
enum WaitMode { WAIT_SPIN = 0, WAIT_YIELD = 1, WAIT_BLOCK = 2 };
std::atomic<int> g_wait_mode{-1};
int wait_mode() {
    int m = g_wait_mode.load(std::memory_order_relaxed);
    if (m >= 0) return m;
    const char *e = getenv("PAV_WAIT");
    if (e && !strcmp(e, "spin")) m = WAIT_SPIN;
    else if (e && !strcmp(e, "yield")) m = WAIT_YIELD;
    else if (e && !strcmp(e, "block")) m = WAIT_BLOCK;
    else m = WAIT_YIELD;
    g_wait_mode.store(m, std::memory_order_relaxed);
    return m;
}
// one event per (thread, device, kind): waits are made by the thread that owns the lane, so nothing is shared
struct WaitEvents {
    static constexpr int MAX_DEV = 64;
    hipEvent_t ev[2][MAX_DEV] = {};
    // a thread that ends (the library's writer and loader threads come and go with every haplotype) gives its events back; the
    // process's first thread keeps them: it ends with the process, when the runtime may already be gone
    ~WaitEvents() {
        if ((long)getpid() == (long)syscall(SYS_gettid)) return;
        for (auto &row : ev) for (hipEvent_t e : row) if (e) (void)hipEventDestroy(e);
    }
};
hipEvent_t wait_event(int mode) {
    constexpr int MAX_DEV = WaitEvents::MAX_DEV;
    thread_local WaitEvents mine;
    auto &ev = mine.ev;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
    hipEvent_t &e = ev[mode == WAIT_BLOCK][dev];
    if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming | (mode == WAIT_BLOCK ? hipEventBlockingSync : 0u)) != hipSuccess) e = nullptr;
    return e;
}
}  // namespace

thread_local double t_waited = 0.0;       // pav_wait_stats
thread_local uint64_t n_waits = 0;
struct WaitClock {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    ~WaitClock() { t_waited += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); ++n_waits; }
};

static hipError_t event_wait_inner(hipEvent_t ev) {
    if (wait_mode() != WAIT_YIELD) return hipEventSynchronize(ev);
    // Poll, and give the core away between the polls: a lane alone waits as quickly as a spinning one (the yield comes back at once),
    // with more lanes than cores the waiting ones take turns behind the threads that have work.  (Sleeping 30 / 80 us between polls
    // once a yield has taken a while - another thread ran - was measured too: six lanes on one core 1.83 / 1.92 Tbp/s against 2.15
    // with the plain yield, on two cores 2.30 / 2.14 against 2.38.)
    for (;;) {
        const hipError_t q = hipEventQuery(ev);
        if (q != hipErrorNotReady) return q;
        sched_yield();
    }
}

hipError_t event_wait(hipEvent_t ev) {
    WaitClock clock;
    return event_wait_inner(ev);
}

hipError_t stream_wait(hipStream_t st) {
    WaitClock clock;
    const int mode = wait_mode();
    if (mode == WAIT_SPIN) return (hipStreamSynchronize)(st);
    hipEvent_t ev = wait_event(mode);
    if (!ev) return (hipStreamSynchronize)(st);
    const hipError_t r = hipEventRecord(ev, st);
    if (r != hipSuccess) return r;
    return event_wait_inner(ev);
}

// ---- profiling ------------------------------------------------------------------------------------------
int prof_begin(pav_ctx *ctx, const char *name, hipStream_t st) {
    if (!st) st = ctx->stream;
    if (!ctx->prof_on) return -1;
    std::lock_guard<std::mutex> lk(ctx->prof_mu);
    int entry = -1;
    for (size_t i = 0; i < ctx->prof.size(); ++i)
        if (ctx->prof[i].name == name) { entry = (int)i; break; }
    if (entry < 0) { ctx->prof.push_back(ProfEntry{name, 0, 0.0}); entry = (int)ctx->prof.size() - 1; }
    hipEvent_t a, b;
    if (ctx->ev_pool.size() >= 2) {
        a = ctx->ev_pool.back(); ctx->ev_pool.pop_back();
        b = ctx->ev_pool.back(); ctx->ev_pool.pop_back();
    } else {
        if (hipEventCreate(&a) != hipSuccess) return -1;
        if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); return -1; }
    }
    (void)hipEventRecord(a, st);
    ctx->prof_pending.push_back(ProfPending{entry, a, b});
    return (int)ctx->prof_pending.size() - 1;
}

void prof_end(pav_ctx *ctx, int token, hipStream_t st) {
    if (token < 0) return;
    hipEvent_t b;
    { std::lock_guard<std::mutex> lk(ctx->prof_mu); b = ctx->prof_pending[(size_t)token].b; }
    (void)hipEventRecord(b, st ? st : ctx->stream);
}

// Everything the caller launches next on ctx->stream runs behind the last full pack of both stores - whichever context
// queued it, on whichever stream (a shared store is packed by one of its users; SeqStore::planes_full only says that the
// pack has been queued).
int wait_planes(pav_ctx *ctx) {
    for (int r = 0; r < 2; ++r) {
        SeqStore &s = ctx->seq[r];
        if (s.pack_event && s.pack_gen != ctx->seen_pack_gen[r]) {
            PAV_HIP(ctx, hipStreamWaitEvent(ctx->stream, s.pack_event, 0));
            ctx->seen_pack_gen[r] = s.pack_gen;
        }
        ctx->pack_pending[r] = false;
    }
    return PAV_OK;
}

int wait_homology(pav_ctx *ctx) {
    if (!ctx->hom_pending) return PAV_OK;
    PAV_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->hom_done, 0));
    ctx->hom_pending = false;
    return PAV_OK;
}

int wait_tables(pav_ctx *ctx) {
    if (!ctx->tables_pending) return PAV_OK;
    PAV_HIP(ctx, event_wait(ctx->tables_done));
    ctx->tables_pending = false;
    return PAV_OK;
}

int prof_flush(pav_ctx *ctx) {
    { std::lock_guard<std::mutex> lk(ctx->prof_mu); if (ctx->prof_pending.empty()) return PAV_OK; }
    PAV_HIP(ctx, hipStreamSynchronize(ctx->stream3));
    PAV_HIP(ctx, hipStreamSynchronize(ctx->stream2));
    PAV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::lock_guard<std::mutex> lk(ctx->prof_mu);
    for (auto &p : ctx->prof_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            ctx->prof[(size_t)p.entry].launches += 1;
            ctx->prof[(size_t)p.entry].ms += (double)ms;
        }
        ctx->ev_pool.push_back(p.a);
        ctx->ev_pool.push_back(p.b);
    }
    ctx->prof_pending.clear();
    return PAV_OK;
}

// ---- pack kernel ----------------------------------------------------------------------------------------
// One lane packs 16 bases (one 16-byte load): a u32 of 2-bit codes and 16 non-ACGT bits; lane pairs merge
// their halves of the 32-base mask word with one cross-lane move.  Pure streaming: 1 B/base in,
// 0.25 + 0.125 B/base out; HBM-bound.
// Tuned on MI355X (tools/ubench/pack_variants.hip): four independent 16-byte loads in flight per lane, non-temporal
// loads and stores (every byte is touched once), one 16 KiB tile per workgroup with an exact grid (a capped
// grid-stride launch was 20 % slower): 6.0 TB/s of algorithmic traffic vs 4.5 TB/s for the first version.
// A wave also writes one summary byte per 1024 bases (dirty: the block holds a non-ACGT base).  Non-ACGT bases are rare, so
// the random-access readers (breakpoint homology, verify mode) look at the summary - a few MB that stay in L2 - and fetch a
// line of the mask plane only for dirty blocks: half of their cache-line fetches otherwise.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int PACK_U = 4;

__global__ __launch_bounds__(256) void pack_kernel(const uint4 *__restrict__ ascii, uint32_t *__restrict__ two,
                                                   uint32_t *__restrict__ mask, uint8_t *__restrict__ dirty, uint64_t n16) {
    const uint64_t base = (uint64_t)blockIdx.x * (256 * PACK_U) + threadIdx.x;
    u32x4 v[PACK_U];
#pragma unroll
    for (int u = 0; u < PACK_U; ++u) {
        const uint64_t i = base + (uint64_t)u * 256;
        if (i < n16) v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(ascii) + i);
        else v[u] = u32x4{0x4e4e4e4eu, 0x4e4e4e4eu, 0x4e4e4e4eu, 0x4e4e4e4eu};
    }
#pragma unroll
    for (int u = 0; u < PACK_U; ++u) {
        const uint64_t i = base + (uint64_t)u * 256;
        uint32_t c0, c1, c2, c3, b0, b1, b2, b3;
        pack4(v[u].x, c0, b0); pack4(v[u].y, c1, b1); pack4(v[u].z, c2, b2); pack4(v[u].w, c3, b3);
        const uint32_t m16 = b0 | (b1 << 4) | (b2 << 8) | (b3 << 12);
        const uint32_t other = __shfl_xor(m16, 1);                    // n16 is even: the partner always exists
        const unsigned long long any_bad = __ballot(m16 != 0);
        if ((threadIdx.x & 63) == 0) dirty[(uint64_t)blockIdx.x * 16 + u * 4 + (threadIdx.x >> 6)] = any_bad != 0;
        if (i < n16) {
            __builtin_nontemporal_store(c0 | (c1 << 8) | (c2 << 16) | (c3 << 24), &two[i]);
            if ((threadIdx.x & 1) == 0) __builtin_nontemporal_store(m16 | (other << 16), &mask[i >> 1]);
        }
    }
}

static int run_pack(pav_ctx *ctx, SeqStore &s, hipStream_t st) {
    if (s.arena == 0) { s.planes_full = true; return PAV_OK; }
    const uint64_t n16 = s.arena / 16;
    const uint64_t blocks = (n16 + 256 * PACK_U - 1) / (256 * PACK_U);
    PAV_LAUNCH_ON(ctx, st, "pack_kernel", pack_kernel, (uint32_t)blocks, 256, 0, s.d_ascii.as<uint4>(),
                  s.d_two.as<uint32_t>(), s.d_mask.as<uint32_t>(), s.d_dirty.as<uint8_t>(), n16);
    // the planes are whole behind this event; readers on other streams (this context's main stream, other contexts that
    // share the store) wait for it in wait_planes
    if (!s.pack_event) PAV_HIP(ctx, hipEventCreateWithFlags(&s.pack_event, hipEventDisableTiming));
    PAV_HIP(ctx, hipEventRecord(s.pack_event, st));
    s.pack_gen += 1;
    s.planes_full = true;
    return PAV_OK;
}

// ---- lazy contig pack -------------------------------------------------------------------------------------
// The contig planes have three readers: the breakpoint-homology scans (windows of <= 32 bases at ~0.6 M scattered places),
// the k-mer scans of the flagged regions (a few % of the contigs) and verify mode (everything).  Packing the whole arena for
// them was the largest kernel of a pass (4.2 of the ~5.7 GB it moved).  So pav_seq_load / pav_seq_pack of the contig role
// only mark the planes stale; the homology scans decode their windows from the ASCII arena (cigar.hip fetch_run), the k-mer
// scans pack the 1024-base blocks of their regions first (pack_spans_kernel: the body of pack_kernel, one wave per block),
// verify mode packs everything.  PAV_EAGER_PACK=1 restores the full pack at load time (A/B measurements, tests).
// The reference is packed once, in full, when it is loaded.
struct SpanDev { uint32_t first_block, pre; };                          // 1024-base blocks of a span; blocks of the spans before it

__global__ __launch_bounds__(256) void pack_spans_kernel(const uint4 *__restrict__ ascii, uint32_t *__restrict__ two,
                                                         uint32_t *__restrict__ mask, uint8_t *__restrict__ dirty,
                                                         const SpanDev *__restrict__ spans, uint32_t n_spans, uint32_t n_blocks,
                                                         uint64_t n16) {
    const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);              // one wave, one block of 1024 bases
    if (w >= n_blocks) return;
    uint32_t lo = 0, hi = n_spans;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (spans[mid].pre <= w) lo = mid; else hi = mid; }
    const uint64_t blk = (uint64_t)spans[lo].first_block + (w - spans[lo].pre);
    const uint64_t i = blk * 64 + (threadIdx.x & 63);                    // the lane's 16 bases
    u32x4 v = u32x4{0x4e4e4e4eu, 0x4e4e4e4eu, 0x4e4e4e4eu, 0x4e4e4e4eu};   // past the arena (its last block may be partial): 'N'
    if (i < n16) v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(ascii) + i);
    uint32_t c0, c1, c2, c3, b0, b1, b2, b3;
    pack4(v.x, c0, b0); pack4(v.y, c1, b1); pack4(v.z, c2, b2); pack4(v.w, c3, b3);
    const uint32_t m16 = b0 | (b1 << 4) | (b2 << 8) | (b3 << 12);
    const uint32_t other = __shfl_xor(m16, 1);                           // n16 is even: the partner always exists
    const unsigned long long any_bad = __ballot(m16 != 0);
    if ((threadIdx.x & 63) == 0) dirty[blk] = any_bad != 0;
    if (i < n16) {
        two[i] = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
        if ((threadIdx.x & 1) == 0) mask[i >> 1] = m16 | (other << 16);
    }
}

static bool eager_pack() { const char *e = getenv("PAV_EAGER_PACK"); return e && *e == '1'; }   // read per load / pack call

int need_planes_full(pav_ctx *ctx, int role) {
    SeqStore &s = ctx->seq[role];
    if (!s.planes_full) { const int rc = run_pack(ctx, s, ctx->stream); if (rc != PAV_OK) return rc; }
    return wait_planes(ctx);
}

int need_planes_spans(pav_ctx *ctx, int role, const std::vector<PlaneSpan> &spans) {
    SeqStore &s = ctx->seq[role];
    if (s.planes_full || s.arena == 0) return wait_planes(ctx);
    // spans -> runs of 1024-base blocks (pack_kernel's tile per wave; the arena is a multiple of 256 bases, its planes and
    // the ASCII buffer are allocated in whole pack tiles of 16384)
    const uint64_t arena_blocks = (s.arena + 1023) >> DIRTY_SHIFT;
    std::vector<std::pair<uint64_t, uint64_t>> runs;                     // [first block, last block]
    runs.reserve(spans.size());
    for (const PlaneSpan &sp : spans) {
        if (sp.len == 0 || (sp.abs >> DIRTY_SHIFT) >= arena_blocks) continue;
        const uint64_t b1 = (sp.abs + sp.len - 1) >> DIRTY_SHIFT;
        runs.emplace_back(sp.abs >> DIRTY_SHIFT, b1 < arena_blocks ? b1 : arena_blocks - 1);
    }
    std::sort(runs.begin(), runs.end());
    std::vector<SpanDev> sd;
    uint64_t total = 0;
    for (size_t i = 0; i < runs.size();) {                               // overlapping / adjacent runs become one
        uint64_t b0 = runs[i].first, b1 = runs[i].second;
        for (++i; i < runs.size() && runs[i].first <= b1 + 1; ++i) b1 = runs[i].second > b1 ? runs[i].second : b1;
        sd.push_back(SpanDev{(uint32_t)b0, (uint32_t)total});
        total += b1 - b0 + 1;
    }
    if (total == 0) return wait_planes(ctx);
    if (total >= 0xFFFFFFFFull) return need_planes_full(ctx, role);
    // more than a third of the arena asked for: the streaming kernel is cheaper than scattered blocks
    if (3 * total > arena_blocks) return need_planes_full(ctx, role);
    ctx->span_host.assign(reinterpret_cast<const uint8_t *>(sd.data()), reinterpret_cast<const uint8_t *>(sd.data()) + sizeof(SpanDev) * sd.size());
    PAV_HIP(ctx, ctx->d_spans.reserve(sizeof(SpanDev) * sd.size()));
    PAV_HIP(ctx, hipMemcpyAsync(ctx->d_spans.p, ctx->span_host.data(), sizeof(SpanDev) * sd.size(), hipMemcpyHostToDevice, ctx->stream));
    PAV_LAUNCH(ctx, "pack_spans_kernel", pack_spans_kernel, (uint32_t)((total + 3) / 4), 256, 0, s.d_ascii.as<uint4>(),
               s.d_two.as<uint32_t>(), s.d_mask.as<uint32_t>(), s.d_dirty.as<uint8_t>(), ctx->d_spans.as<SpanDev>(),
               (uint32_t)sd.size(), (uint32_t)total, s.arena / 16);
    return wait_planes(ctx);
}

}  // namespace pav

using namespace pav;

namespace pav {

struct UploadRings { UploadRing r[2]; };               // one per role: the two stores of a context may be loaded side by side

}  // namespace pav

// ---- the process-wide list of idle device blocks (common.h) -------------------------------------------------------------------
// A block remembers the device it was allocated on (not whatever device is current when it comes back); a device's idle blocks are
// capped at PAV_DEVICE_POOL_GB (default 24: three haplotypes' worth of stores and tables) and at a quarter of the device's memory,
// so that other processes on the GPU - share_gpu ranks, RCCL, torch - are not starved by memory nobody uses; pav_device_pool_trim
// gives everything back.
namespace {
struct BlockPool {
    std::mutex mu;
    struct B { int dev; void *p; size_t cap; };
    std::vector<B> idle;
    std::unordered_map<void *, int> owner;             // every live block of >= BLOCK_MIN: the device it was allocated on
};
BlockPool &block_pool() { static BlockPool *P = new BlockPool(); return *P; }    // (never destroyed: the HIP runtime may be gone by then)
constexpr size_t BLOCK_MIN = 32ull << 20;
bool block_pool_on() { static const bool on = [] { const char *e = getenv("PAV_DEVICE_POOL"); return !(e && e[0] == '0'); }(); return on; }
size_t block_keep(int dev) {
    static const size_t env_cap = [] { const char *e = getenv("PAV_DEVICE_POOL_GB"); const double gb = e ? atof(e) : 24.0; return (size_t)(gb > 0 ? gb * (double)(1ull << 30) : 0.0); }();
    static std::mutex mu;
    static std::unordered_map<int, size_t> quarter;
    std::lock_guard<std::mutex> lk(mu);
    auto it = quarter.find(dev);
    if (it == quarter.end()) {
        hipDeviceProp_t prop;
        const size_t total = hipGetDeviceProperties(&prop, dev) == hipSuccess ? (size_t)prop.totalGlobalMem : (size_t)64 << 30;
        it = quarter.emplace(dev, total / 4).first;
    }
    return std::min(env_cap, it->second);
}
// what hipFree would have waited for - nothing queued on the block's device still uses it - with the caller's device restored
void sync_device_of(int dev) {
    int cur = dev;
    (void)hipGetDevice(&cur);
    if (cur != dev) (void)hipSetDevice(dev);
    (void)hipDeviceSynchronize();
    if (cur != dev) (void)hipSetDevice(cur);
}
}  // namespace

hipError_t pav::dev_block_get(void **p, size_t *cap, size_t want) {
    *p = nullptr; *cap = 0;
    int dev = 0;
    const bool pooled = want >= BLOCK_MIN && block_pool_on() && hipGetDevice(&dev) == hipSuccess;
    if (pooled) {
        BlockPool &P = block_pool();
        std::lock_guard<std::mutex> lk(P.mu);
        size_t best = P.idle.size();
        for (size_t i = 0; i < P.idle.size(); ++i)           // the smallest idle block that holds it without wasting more than it holds
            if (P.idle[i].dev == dev && P.idle[i].cap >= want && P.idle[i].cap <= 2 * want + (256ull << 20)
                && (best == P.idle.size() || P.idle[i].cap < P.idle[best].cap)) best = i;
        if (best < P.idle.size()) { *p = P.idle[best].p; *cap = P.idle[best].cap; P.idle.erase(P.idle.begin() + (long)best); return hipSuccess; }
    }
    hipError_t e = hipMalloc(p, want);
    if (e != hipSuccess && block_pool_on()) {               // out of memory with idle blocks held: let them go and try again
        (void)hipGetLastError();
        (void)hipGetDevice(&dev);
        pav::dev_pool_trim(dev);
        e = hipMalloc(p, want);
    }
    if (e != hipSuccess) { *p = nullptr; return e; }
    *cap = want;
    if (pooled) { BlockPool &P = block_pool(); std::lock_guard<std::mutex> lk(P.mu); P.owner[*p] = dev; }
    return hipSuccess;
}

void pav::dev_block_put(void *p, size_t cap) {
    if (!p) return;
    if (cap >= BLOCK_MIN && block_pool_on()) {
        BlockPool &P = block_pool();
        int dev = -1;
        { std::lock_guard<std::mutex> lk(P.mu); auto it = P.owner.find(p); if (it != P.owner.end()) dev = it->second; }
        if (dev >= 0) {
            sync_device_of(dev);
            const size_t keep = block_keep(dev);
            std::lock_guard<std::mutex> lk(P.mu);
            size_t held = 0;
            for (auto &b : P.idle) if (b.dev == dev) held += b.cap;
            if (held + cap <= keep) { P.idle.push_back(BlockPool::B{dev, p, cap}); return; }
            P.owner.erase(p);
        }
    }
    (void)hipFree(p);
}

// every idle block of the device (< 0: of every device) back to the driver; returns the bytes freed
size_t pav::dev_pool_trim(int device) {
    std::vector<BlockPool::B> drop;
    { BlockPool &P = block_pool(); std::lock_guard<std::mutex> lk(P.mu);
      for (auto &b : P.idle) if (device < 0 || b.dev == device) { drop.push_back(b); P.owner.erase(b.p); }
      P.idle.erase(std::remove_if(P.idle.begin(), P.idle.end(), [&](const BlockPool::B &b) { return device < 0 || b.dev == device; }), P.idle.end()); }
    size_t bytes = 0;
    for (auto &b : drop) { (void)hipFree(b.p); bytes += b.cap; }
    return bytes;
}

namespace pav {

// Rings that outlive their context (a process-wide list): 128 MB of pinned memory cost 21 ms to get and 10 ms to give back
// (tools/ubench/pin_cost.hip: 0.165 ms per MB), twice per context - a tenth of a haplotype's files to files when every haplotype has a
// context of its own.  A context takes its rings from the list and returns them; at most four idle rings are kept.
namespace {
struct RingPool { std::mutex mu; std::vector<std::pair<int, UploadRing>> idle; };
RingPool &ring_pool() { static RingPool *P = new RingPool(); return *P; }
void ring_free(UploadRing &R) {
    for (int i = 0; i < UploadRing::SLOTS; ++i) { if (R.slot[i]) (void)hipHostFree(R.slot[i]); if (R.ev[i]) (void)hipEventDestroy(R.ev[i]); R.slot[i] = nullptr; R.ev[i] = nullptr; }
}
}  // namespace

UploadRing *upload_ring(pav_ctx *ctx, int which) {
    {
        static std::mutex first;                       // (the two roles' loads may be the first users of the context at the same moment)
        std::lock_guard<std::mutex> lk(first);
        if (!ctx->upload) ctx->upload = new UploadRings();
    }
    UploadRing *R = &static_cast<UploadRings *>(ctx->upload)->r[which & 1];
    if (R->slot[0] || R->ok) return R;
    {
        RingPool &P = ring_pool();
        std::lock_guard<std::mutex> lk(P.mu);
        for (size_t i = 0; i < P.idle.size(); ++i)
            if (P.idle[i].first == ctx->device) { *R = P.idle[i].second; P.idle.erase(P.idle.begin() + (long)i); return R; }
    }
    R->ok = true;
    for (int i = 0; i < UploadRing::SLOTS && R->ok; ++i)
        R->ok = hipHostMalloc(&R->slot[i], UploadRing::SLOT_BYTES, hipHostMallocDefault) == hipSuccess &&
                hipEventCreateWithFlags(&R->ev[i], hipEventDisableTiming) == hipSuccess;
    return R;
}

void upload_release(pav_ctx *ctx) {
    if (!ctx || !ctx->upload) return;
    auto *P = static_cast<UploadRings *>(ctx->upload);
    for (UploadRing &R : P->r) {
        if (!R.slot[0]) continue;
        bool kept = false;
        if (R.ok) {
            for (int i = 0; i < UploadRing::SLOTS; ++i) if (R.busy[i]) { (void)hipEventSynchronize(R.ev[i]); R.busy[i] = false; }
            R.next = 0;
            RingPool &G = ring_pool();
            std::lock_guard<std::mutex> lk(G.mu);
            if (G.idle.size() < 4) { G.idle.emplace_back(ctx->device, R); kept = true; }
        }
        if (!kept) ring_free(R);
    }
    delete P;
    ctx->upload = nullptr;
}

// dst[0, bytes) on the device <- src (pageable), queued on `st`; returns when every byte has left `src` (not when it has arrived)
int staged_upload(pav_ctx *ctx, hipStream_t st, uint8_t *dst, const uint8_t *src, uint64_t bytes, int which) {
    UploadRing *R = upload_ring(ctx, which);
    static const int threads = [] { const char *e = getenv("PAV_UPLOAD_THREADS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 8; }();
    if (!R->ok || bytes < (4u << 20) || threads <= 1) {
        PAV_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
        return PAV_OK;
    }
    for (uint64_t at = 0; at < bytes; at += UploadRing::SLOT_BYTES) {
        const uint64_t n = std::min<uint64_t>(UploadRing::SLOT_BYTES, bytes - at);
        const int k = R->next; R->next = (k + 1) % UploadRing::SLOTS;
        if (R->busy[k]) { PAV_HIP(ctx, hipEventSynchronize(R->ev[k])); R->busy[k] = false; }
        uint8_t *stage = static_cast<uint8_t *>(R->slot[k]);
        const int use = n >= (8u << 20) ? threads : 1;                 // (a thread is not worth starting for a small record)
        const uint64_t piece = (n + (uint64_t)use - 1) / (uint64_t)use;
        std::vector<std::thread> pool;
        for (int t = 1; t < use; ++t) {
            const uint64_t a = std::min<uint64_t>(n, piece * (uint64_t)t), b = std::min<uint64_t>(n, a + piece);
            if (b > a) pool.emplace_back([=] { memcpy(stage + a, src + at + a, b - a); });
        }
        memcpy(stage, src + at, std::min<uint64_t>(n, piece));
        for (auto &th : pool) th.join();
        PAV_HIP(ctx, hipMemcpyAsync(dst + at, stage, n, hipMemcpyHostToDevice, st));
        PAV_HIP(ctx, hipEventRecord(R->ev[k], st));
        R->busy[k] = true;
    }
    return PAV_OK;
}

}  // namespace pav

namespace pav {

// The records of one role into its store: layout (every record on a 256-base boundary, one pad block behind it), buffers, the
// arena filled with 'N', then `fill(arena, off)` queues whatever brings the ASCII bytes of record i to arena + off[i] on the
// context's stream (pav_seq_load: uploads from host arrays; pav_seq_load_fasta_path: a strip kernel over the file's raw text),
// then offsets, lengths, the pack of a reference, and the wait.
int seq_store_load(pav_ctx *ctx, int role, uint32_t n_seq, const uint64_t *len, const char *what,
                   const std::function<int(uint8_t *, const std::vector<uint64_t> &)> &fill) {
    // lengths are checked and the layout is planned before the store is touched: a refused call leaves it as it was
    std::vector<uint64_t> off(n_seq, 0);
    uint64_t a = 0, total = 0;
    for (uint32_t i = 0; i < n_seq; ++i) {
        if (len[i] >= 0xFFFFFF00ull)
            return fail(ctx, PAV_E_LIMIT, "%s: record %u has %llu bases (limit 2^32 - 256)", what, i, (unsigned long long)len[i]);
        off[i] = a;
        a += (len[i] + SEQ_ALIGN - 1) / SEQ_ALIGN * SEQ_ALIGN + SEQ_ALIGN;   // one pad block between records
        total += len[i];
    }
    PAV_HIP(ctx, hipSetDevice(ctx->device));
    { const int rch = wait_homology(ctx); if (rch != PAV_OK) return rch; }     // the scans of the last call read the arenas
    PAV_HIP(ctx, hipStreamSynchronize(ctx->stream2));   // a re-pack of the old arena may still be running
    PAV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->pack_pending[role] = false;
    if (ctx->seq.p[role].use_count() > 1) ctx->seq.p[role] = std::make_shared<SeqStore>();   // other contexts keep the shared one
    SeqStore &s = ctx->seq[role];
    s.device = ctx->device;
    // from here on the old content is gone: results and tables that refer to it are invalid, and so is the store until the
    // uploads and the pack have succeeded
    ctx->cigar_called = false;
    s.n = 0; s.arena = s.total = 0;
    s.off.clear(); s.len.clear();
    struct Guard {                                   // any early return below leaves an empty store and no loaded table
        pav_ctx *c; bool armed = true;
        ~Guard() { if (armed) c->cigar_loaded = false; }
    } guard{ctx};
    if (a == 0) { guard.armed = false; s.n = n_seq; s.off = off; s.len.assign(len, len + n_seq); return PAV_OK; }
    const bool timing = getenv("PAV_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_a = now();
    PAV_HIP(ctx, s.d_ascii.reserve(a));
    PAV_HIP(ctx, s.d_two.reserve(a / 4));
    PAV_HIP(ctx, s.d_mask.reserve(a / 8));
    PAV_HIP(ctx, s.d_dirty.reserve((a / 16 + 256 * PACK_U - 1) / (256 * PACK_U) * 16 + 16));   // 16 summary bytes per pack workgroup (+ slack: verify_kernel reads pairs)
    PAV_HIP(ctx, s.d_off.reserve(sizeof(uint64_t) * n_seq));
    PAV_HIP(ctx, s.d_len.reserve(sizeof(uint64_t) * n_seq));
    const double t_b = now();
    PAV_HIP(ctx, hipMemsetAsync(s.d_ascii.p, 'N', a, ctx->stream));          // padding reads as non-ACGT
    { const int rcf = fill(s.d_ascii.as<uint8_t>(), off); if (rcf != PAV_OK) return rcf; }
    PAV_HIP(ctx, hipMemcpyAsync(s.d_off.p, off.data(), sizeof(uint64_t) * n_seq, hipMemcpyHostToDevice, ctx->stream));
    PAV_HIP(ctx, hipMemcpyAsync(s.d_len.p, len, sizeof(uint64_t) * n_seq, hipMemcpyHostToDevice, ctx->stream));
    const double t_c = now();
    s.n = n_seq; s.arena = a; s.total = total;                               // run_pack reads the layout from the store
    s.off = off; s.len.assign(len, len + n_seq);
    // contigs: planes on demand ("lazy contig pack" above); the reference is packed now
    s.planes_full = false;
    int rc = (role == PAV_ROLE_REF || eager_pack()) ? run_pack(ctx, s, ctx->stream) : PAV_OK;
    if (rc == PAV_OK && hipStreamSynchronize(ctx->stream) != hipSuccess)     // inputs are borrowed only for the duration of the call
        rc = fail(ctx, PAV_E_HIP, "%s: upload / pack failed", what);
    if (rc != PAV_OK) { s.n = 0; s.arena = s.total = 0; s.off.clear(); s.len.clear(); return rc; }
    if (timing) fprintf(stderr, "[pav timing] %s role %d: %.2f GB; device buffers %.1f ms, fill queued in %.1f ms (%.1f GB/s), pack + drain %.1f ms\n", what, role,
                        (double)total / 1e9, (t_b - t_a) * 1e3, (t_c - t_b) * 1e3, (double)total / 1e9 / std::max(1e-9, t_c - t_b), (now() - t_c) * 1e3);
    guard.armed = false;
    return PAV_OK;
}

}  // namespace pav

extern "C" {

int pav_abi_version(void) { return PAV_ABI_VERSION; }

uint64_t pav_device_pool_trim(int device_id) { return (uint64_t)pav::dev_pool_trim(device_id); }

int pav_wait_stats(double out[2]) {
    if (!out) return PAV_E_ARG;
    out[0] = pav::t_waited; out[1] = (double)pav::n_waits;
    return PAV_OK;
}

int pav_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *pav_last_error(const pav_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

// The side stream (contig pack, SNV rows).  PAV_PACK_CUS = n (experiment): restrict it to n compute units with a CU mask, so
// that the short kernels of the main stream always find a free CU next to the HBM-saturating pack; priorities cannot be
// combined with a mask, the stream then has the default one.
static hipError_t create_side_stream(hipStream_t *st, int prio, int n_cu) {
    const char *env = getenv("PAV_PACK_CUS");
    const int want = env ? atoi(env) : 0;
    if (want <= 0 || want >= n_cu) return hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio);
    std::vector<uint32_t> mask((size_t)(n_cu + 31) / 32, 0u);
    const char *mode = getenv("PAV_PACK_CU_SPREAD");
    for (int i = 0; i < n_cu; ++i) {
        // spread: leave every (n_cu / (n_cu - want))-th unit free; otherwise the low `want` units
        const bool on = mode ? ((long long)(i + 1) * (n_cu - want) / n_cu == (long long)i * (n_cu - want) / n_cu) : i < want;
        if (on) mask[(size_t)i / 32] |= 1u << (i % 32);
    }
    return hipExtStreamCreateWithCUMask(st, (uint32_t)mask.size(), mask.data());
}

// PAV_PRIO = main | equal (experiment): which of the two streams gets the high priority (default: the side stream).
static int main_prio(int lo, int hi) { const char *e = getenv("PAV_PRIO"); return e && !strcmp(e, "main") ? hi : lo; }
static int side_prio(int lo, int hi) { const char *e = getenv("PAV_PRIO"); return e && (!strcmp(e, "main") || !strcmp(e, "equal")) ? lo : hi; }

pav_ctx *pav_create(int device_id) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        fail(nullptr, PAV_E_NODEV, "no HIP device visible (%s): libpav_amd has no CPU fallback",
             e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        return nullptr;
    }
    if (device_id < 0 || device_id >= n) {
        fail(nullptr, PAV_E_ARG, "device_id %d out of range [0, %d)", device_id, n);
        return nullptr;
    }
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess) {
        fail(nullptr, PAV_E_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
        return nullptr;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fail(nullptr, PAV_E_NODEV, "device %d is %s; libpav_amd is built for gfx950 (MI355X) only", device_id,
             prop.gcnArchName);
        return nullptr;
    }
    pav_ctx *ctx = new pav_ctx();
    ctx->device = device_id;
    ctx->n_cu = prop.multiProcessorCount;
    snprintf(ctx->dev_name, sizeof ctx->dev_name, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, ctx->n_cu);
    int prio_lo = 0, prio_hi = 0;
    if ((e = hipSetDevice(device_id)) != hipSuccess ||
        (e = hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi)) != hipSuccess ||
        // The side stream (contig pack, then the SNV rows) gets the high priority, the main stream with its chain of short
        // latency-bound kernels the low one: the pack is the one HBM-bound kernel of a step and runs at its stand-alone rate
        // that way (0.68 ms = 0.78 of the HBM peak inside the step; 0.78 ms = 0.68 with the priorities the other way round).
        // The whole path runs at the same rate either way (the step is bound by the call tables' PCIe time); a CIGAR-call-only
        // step costs 1.31 instead of 1.19 ms, because its chain now waits for the pack more often.
        (e = hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, main_prio(prio_lo, prio_hi))) != hipSuccess ||
        (e = create_side_stream(&ctx->stream2, side_prio(prio_lo, prio_hi), ctx->n_cu)) != hipSuccess ||
        (e = hipStreamCreateWithPriority(&ctx->stream3, hipStreamNonBlocking, prio_lo)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->tables_done, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->tables_done_prev, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->hom_done, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->snv_ready, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->snv_done, hipEventDisableTiming)) != hipSuccess ||
        (e = hipHostMalloc(reinterpret_cast<void **>(&ctx->h_status), 256, hipHostMallocDefault)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->pack_done[0], hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->pack_done[1], hipEventDisableTiming)) != hipSuccess) {
        fail(nullptr, PAV_E_HIP, "device init: %s", hipGetErrorString(e));
        delete ctx;
        return nullptr;
    }
    return ctx;
}

void pav_density_release(pav_ctx *ctx);   // density.hip
void pav_invscan_release(pav_ctx *ctx);   // invscan.cpp
void pav_flag_release(pav_ctx *ctx);      // flag.hip
void pav_trim_release(pav_ctx *ctx);      // trim.cpp

void pav_destroy(pav_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream3);
    (void)hipStreamSynchronize(ctx->stream2);
    (void)hipStreamSynchronize(ctx->stream);
    table_writer_release(ctx);
    pav::upload_release(ctx);
    pav::fastadev_release(ctx);
    pav::gz_release(ctx);
    pav::textdev_release(ctx);
    pav_density_release(ctx);
    pav_invscan_release(ctx);
    pav_flag_release(ctx);
    pav_trim_release(ctx);
    for (auto &p : ctx->prof_pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto ev : ctx->ev_pool) (void)hipEventDestroy(ev);
    for (int r = 0; r < 2; ++r) ctx->seq.p[r].reset();              // the planes go with their last user
    DevBuf *bufs[] = {&ctx->d_aln, &ctx->d_text, &ctx->d_text_off, &ctx->d_ops, &ctx->d_op_off, &ctx->d_chunk,
                      &ctx->d_chunk2, &ctx->d_rowbase, &ctx->d_totals, &ctx->d_snv, &ctx->d_indel,
                      &ctx->d_seqblob, &ctx->d_tmp, &ctx->d_spans, &ctx->ix_text, &ctx->ix_off, &ctx->ix_pos, &ctx->ix_ops, &ctx->ix_op_off,
                      &ctx->ix_chunk, &ctx->ix_chunk2, &ctx->ix_rowbase, &ctx->ix_begin, &ctx->ix_err};
    for (DevBuf *b : bufs) b->release();
    (void)hipStreamDestroy(ctx->stream);
    (void)hipStreamDestroy(ctx->stream2);
    (void)hipStreamDestroy(ctx->stream3);
    (void)hipEventDestroy(ctx->tables_done);
    (void)hipEventDestroy(ctx->tables_done_prev);
    (void)hipEventDestroy(ctx->hom_done);
    (void)hipEventDestroy(ctx->snv_ready);
    (void)hipEventDestroy(ctx->snv_done);
    if (ctx->writer_ready) (void)hipEventDestroy(ctx->writer_ready);
    if (ctx->h_status) (void)hipHostFree(ctx->h_status);
    (void)hipEventDestroy(ctx->pack_done[0]);
    (void)hipEventDestroy(ctx->pack_done[1]);
    delete ctx;
}

int pav_device_name(const pav_ctx *ctx, char *buf, int buf_len) {
    if (!ctx || !buf || buf_len <= 0) return PAV_E_ARG;
    snprintf(buf, (size_t)buf_len, "%s", ctx->dev_name);
    return PAV_OK;
}

// "domain:bus:device.function" of the context's GPU (hipDeviceGetPCIBusId): the ranks of a multi-GPU run print it, so that a
// scaling line shows N distinct devices
int pav_device_pci_bus_id(const pav_ctx *ctx, char *buf, int buf_len) {
    if (!ctx || !buf || buf_len < 16) return PAV_E_ARG;
    buf[0] = 0;
    if (hipDeviceGetPCIBusId(buf, buf_len, ctx->device) != hipSuccess) { buf[0] = 0; return PAV_E_HIP; }
    return PAV_OK;
}

int pav_sync(pav_ctx *ctx) {
    if (!ctx) return PAV_E_ARG;
    PAV_HIP(ctx, hipSetDevice(ctx->device));
    PAV_HIP(ctx, hipStreamSynchronize(ctx->stream2));
    PAV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PAV_HIP(ctx, hipStreamSynchronize(ctx->stream3));
    ctx->tables_pending = false;
    ctx->tables_pending_prev = false;
    ctx->hom_pending = false;
    return PAV_OK;
}

int pav_seq_load(pav_ctx *ctx, int role, uint32_t n_seq, const uint8_t *const *ascii, const uint64_t *len) {
    if (!ctx || (role != PAV_ROLE_REF && role != PAV_ROLE_TIG)) return fail(ctx, PAV_E_ARG, "pav_seq_load: bad role");
    if (n_seq && (!ascii || !len)) return fail(ctx, PAV_E_ARG, "pav_seq_load: null input");
    return pav::seq_store_load(ctx, role, n_seq, len, "pav_seq_load", [&](uint8_t *arena, const std::vector<uint64_t> &off) {
        for (uint32_t i = 0; i < n_seq; ++i)
            if (len[i]) {
                const int rcu = pav::staged_upload(ctx, ctx->stream, arena + off[i], ascii[i], len[i], role);
                if (rcu != PAV_OK) return rcu;
            }
        return (int)PAV_OK;
    });
}

int pav_seq_share(pav_ctx *ctx, const pav_ctx *from, int role) {
    if (!ctx || !from || (role != PAV_ROLE_REF && role != PAV_ROLE_TIG)) return fail(ctx, PAV_E_ARG, "pav_seq_share: bad argument");
    if (ctx->device != from->device) return fail(ctx, PAV_E_ARG, "pav_seq_share: the two contexts are on different devices");
    PAV_HIP(ctx, hipSetDevice(ctx->device));
    PAV_HIP(ctx, hipStreamSynchronize(ctx->stream2));
    PAV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->hom_pending = false;
    ctx->pack_pending[role] = false;
    ctx->cigar_loaded = ctx->cigar_called = false;
    ctx->seq.p[role] = from->seq.p[role];
    ctx->seen_pack_gen[role] = 0;
    // a shared store is read from several streams: its planes are made whole here, once, and stay so (pav_seq_pack
    // re-packs a shared store in full; the lazy contig pack is for stores with one user)
    SeqStore &s = ctx->seq[role];
    if (!s.planes_full) {
        // A store that is packed on demand (contigs) becomes shared: this context packs it in full.  `from` must be idle
        // during this call (include/pav_amd.h) - its kernels queued earlier may still decode the ASCII arena, which the pack
        // only reads; the ones it launches after this call see planes_full and wait for pack_event.
        const int rc = run_pack(ctx, s, ctx->stream);
        if (rc != PAV_OK) return rc;
    }
    // a pack the owner (or another sharer) has queued but not finished - e.g. an asynchronous pav_seq_pack on its side
    // stream - is waited for here, and again by every plane reader of this context through wait_planes
    if (s.pack_event) PAV_HIP(ctx, hipEventSynchronize(s.pack_event));
    return PAV_OK;
}

int pav_seq_pack(pav_ctx *ctx, int role) {
    if (!ctx || (role != PAV_ROLE_REF && role != PAV_ROLE_TIG)) return fail(ctx, PAV_E_ARG, "pav_seq_pack: bad role");
    PAV_HIP(ctx, hipSetDevice(ctx->device));
    // Asynchronous on the side stream: everything queued so far on the main stream may still read the old planes, so
    // the pack first waits for the main stream; consumers of the planes wait for pack_done (pav::wait_planes).
    if (role == PAV_ROLE_TIG && !eager_pack() && ctx->seq.p[role].use_count() == 1) {
        ctx->seq[role].planes_full = false;           // readers pack what they need (need_planes_full / need_planes_spans)
        return PAV_OK;
    }
    hipEvent_t *ev = &ctx->pack_done[role];
    PAV_HIP(ctx, hipEventRecord(*ev, ctx->stream));
    PAV_HIP(ctx, hipStreamWaitEvent(ctx->stream2, *ev, 0));
    int rc = run_pack(ctx, ctx->seq[role], ctx->stream2);   // records the store's pack_event: what every reader waits for
    if (rc != PAV_OK) return rc;
    ctx->pack_pending[role] = true;
    return PAV_OK;
}

int pav_mem_info(pav_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes) {
    if (!ctx) return PAV_E_ARG;
    PAV_HIP(ctx, hipSetDevice(ctx->device));
    size_t f = 0, t = 0;
    PAV_HIP(ctx, hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return PAV_OK;
}

int pav_kde_work(const pav_ctx *ctx, double out[3]) {
    if (!ctx || !out) return PAV_E_ARG;
    for (int i = 0; i < 3; ++i) out[i] = ctx->kde_work[i];
    return PAV_OK;
}

int pav_seq_count(const pav_ctx *ctx, int role, uint32_t *n_seq, uint64_t *total_bases) {
    if (!ctx || (role != PAV_ROLE_REF && role != PAV_ROLE_TIG)) return PAV_E_ARG;
    if (n_seq) *n_seq = ctx->seq[role].n;
    if (total_bases) *total_bases = ctx->seq[role].total;
    return PAV_OK;
}

int pav_prof_enable(pav_ctx *ctx, int on) {
    if (!ctx) return PAV_E_ARG;
    int rc = prof_flush(ctx);
    ctx->prof_on = on != 0;
    return rc;
}

int pav_prof_reset(pav_ctx *ctx) {
    if (!ctx) return PAV_E_ARG;
    int rc = prof_flush(ctx);
    ctx->prof.clear();
    return rc;
}

int pav_prof_count(pav_ctx *ctx) {
    if (!ctx) return PAV_E_ARG;
    if (prof_flush(ctx) != PAV_OK) return PAV_E_HIP;
    return (int)ctx->prof.size();
}

int pav_prof_get(pav_ctx *ctx, int i, char *name, int name_len, uint64_t *launches, double *total_ms) {
    if (!ctx || i < 0 || (size_t)i >= ctx->prof.size()) return PAV_E_ARG;
    const ProfEntry &e = ctx->prof[(size_t)i];
    if (name && name_len > 0) snprintf(name, (size_t)name_len, "%s", e.name.c_str());
    if (launches) *launches = e.launches;
    if (total_ms) *total_ms = e.ms;
    return PAV_OK;
}

}  // extern "C"
