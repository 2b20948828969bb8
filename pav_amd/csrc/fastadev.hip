// A FASTA file straight into a context's sequence store: the file's text crosses PCIe as it is and loses its header lines and
// line breaks ON THE DEVICE.  What this replaces is pysam.FastaFile(...).fetch(name) of whole records (pavlib/cigarcall.py:59-66,
// pavlib/seq.py:339-351) - and, inside this library, pav_fasta_open + pav_seq_load_fasta (fastaio.cpp): five host passes over a
// 3 GB file (read, find records, count line breaks, copy without them, stage for the upload), 1.9 CPU-seconds per file - most of
// the 0.43 s "sequences" stage of a haplotype on a box whose process may use sixteen cores.  Here the host reads the file once, in
// parallel pieces, into the pinned slots of the upload ring (a BGZF file - the form PAV keeps its FASTA files in - as it is on disk:
// its members are inflated on the device, inflate.hip; other gzip streams are inflated on the host first), and the device
//   k_fa_marks    counts the line-break bytes of every 256-byte tile (a wave a tile, four bytes a lane) and lists the '>' that start a line,
//   k_fa_hdr_end  finds where each header line ends,
//   (prefix sum of the tile counts: scan_dev.h)
//   k_fa_records  turns every record's body [start, end) into its number of kept bytes,
//   k_fa_strip    moves every kept byte to its place in the arena: its record's offset + its distance from the body's start
//                 - the line breaks between the two (tile prefix + four ballots inside the tile); a lane whose four bytes are
//                 all kept stores them as one word.
// The host sees the header list (a few thousand entries), reads the names from the file, lays the records out as pav_seq_load
// does (seq_store_load) and sets the names.  Same bytes in the arena as the host parser produces (tests/test_gpu_fasta.py).
#include "common.h"
#include "fileio.h"
#include "inflatedev.h"
#include "scan_dev.h"
#include "upload.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>

namespace pav {

int pav_seq_set_names_internal(pav_ctx *ctx, int role, const std::vector<std::string> &names);   // invscan.cpp

namespace {

constexpr uint32_t FA_TILE = 256;

struct FaRec { uint64_t body, body_end, breaks_before, arena_off; };     // breaks_before: line-break bytes in raw[0, body)

__device__ __forceinline__ bool is_break(uint8_t c) { return c == '\n' || c == '\r'; }

// Four bytes a lane: a wave covers one tile of 256 bytes (FA_TILE), a workgroup four.
__device__ __forceinline__ uint32_t break_bits(uint32_t w) {               // bit j: byte j of w is a line-break byte
    uint32_t m = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) { const uint32_t c = (w >> (8 * j)) & 0xFFu; m |= (c == '\n' || c == '\r') ? 1u << j : 0u; }
    return m;
}
__device__ __forceinline__ uint32_t load_tile_word(const uint8_t *__restrict__ raw, uint64_t n, uint64_t x) {   // bytes [x, x + 4) of the text, 'A' beyond its end
    if (x + 4 <= n) return *reinterpret_cast<const uint32_t *>(raw + x);    // (x is a multiple of four, the buffer is an allocation of its own)
    uint32_t w = 0x41414141u;
    for (uint32_t j = 0; j < 4; ++j) if (x + j < n) w = (w & ~(0xFFu << (8 * j))) | (uint32_t)raw[x + j] << (8 * j);
    return w;
}

constexpr uint32_t FA_TPW = 4;                          // tiles a wave takes: their loads are under way together

__global__ __launch_bounds__(256) void k_fa_marks(const uint8_t *__restrict__ raw, uint64_t n, uint32_t *__restrict__ tile_cnt,
                                                  uint64_t *__restrict__ hdr_pos, uint32_t *__restrict__ n_hdr, uint32_t hdr_cap) {
    const uint64_t tile0 = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * FA_TPW;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t ws[FA_TPW];
#pragma unroll
    for (uint32_t q = 0; q < FA_TPW; ++q) { const uint64_t x = (tile0 + q) * FA_TILE + 4ull * lane; ws[q] = (tile0 + q) * FA_TILE < n ? load_tile_word(raw, n, x) : 0x41414141u; }
#pragma unroll
    for (uint32_t q = 0; q < FA_TPW; ++q) {
        const uint64_t tile = tile0 + q, x = tile * FA_TILE + 4ull * lane;
        if (tile * FA_TILE >= n) break;                                    // (whole waves: the reduction below is among live lanes)
        const uint32_t w = ws[q];
        uint32_t cnt = (uint32_t)__popc(break_bits(w));
        for (int d = 32; d >= 1; d >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, d);
        if (lane == 0) tile_cnt[tile] = cnt;
        // a '>' that starts a line: the byte in front of it is '\n' (or it is the file's first byte)
        uint32_t gt = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) gt |= ((w >> (8 * j)) & 0xFFu) == '>' ? 1u << j : 0u;
        if (gt) {
            for (uint32_t j = 0; j < 4; ++j) {
                if (!((gt >> j) & 1u) || x + j >= n) continue;
                const uint32_t prev = j ? (w >> (8 * (j - 1))) & 0xFFu : (x ? (uint32_t)raw[x - 1] : (uint32_t)'\n');
                if (prev != '\n') continue;
                const uint32_t at = atomicAdd(n_hdr, 1u);
                if (at < hdr_cap) hdr_pos[at] = x + j;
            }
        }
    }
}

// hdr[i] = start of a header line ('>'); end[i] = position of the '\n' that ends it (n when the file ends first)
__global__ __launch_bounds__(64) void k_fa_hdr_end(const uint8_t *__restrict__ raw, uint64_t n, const uint64_t *__restrict__ hdr, uint32_t n_hdr,
                                                   uint64_t *__restrict__ end) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n_hdr) return;
    uint64_t p = hdr[i];
    while (p < n && raw[p] != '\n') ++p;
    end[i] = p;
}

// line-break bytes in raw[0, x): the prefix of x's tile + the ones in the tile before x
__device__ __forceinline__ uint64_t breaks_before(const uint8_t *__restrict__ raw, const uint64_t *__restrict__ tile_pre, uint64_t x) {
    const uint64_t t0 = x / FA_TILE * FA_TILE;
    uint64_t b = tile_pre[x / FA_TILE];
    for (uint64_t p = t0; p < x; ++p) b += is_break(raw[p]) ? 1u : 0u;
    return b;
}

__global__ __launch_bounds__(64) void k_fa_records(const uint8_t *__restrict__ raw, const uint64_t *__restrict__ tile_pre, FaRec *__restrict__ rec,
                                                   uint32_t n_rec, uint64_t *__restrict__ kept) {
    const uint32_t r = blockIdx.x * 64 + threadIdx.x;
    if (r >= n_rec) return;
    const uint64_t a = rec[r].body, b = rec[r].body_end;
    const uint64_t ba = breaks_before(raw, tile_pre, a), bb = breaks_before(raw, tile_pre, b);
    rec[r].breaks_before = ba;
    kept[r] = (b - a) - (bb - ba);
}

__global__ __launch_bounds__(256) void k_fa_strip(const uint8_t *__restrict__ raw, uint64_t n, const uint64_t *__restrict__ tile_pre,
                                                  const FaRec *__restrict__ rec, uint32_t n_rec, uint8_t *__restrict__ arena) {
    const uint64_t tile0 = ((uint64_t)blockIdx.x * 4 + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))) * FA_TPW;   // (a scalar: the search below
                                                                                                                                   // loads through the scalar cache)
    const uint32_t lane = threadIdx.x & 63;
    if (tile0 * FA_TILE >= n || n_rec == 0) return;
    uint32_t ws[FA_TPW];
#pragma unroll
    for (uint32_t q = 0; q < FA_TPW; ++q) { const uint64_t x = (tile0 + q) * FA_TILE + 4ull * lane; ws[q] = (tile0 + q) * FA_TILE < n ? load_tile_word(raw, n, x) : 0x0a0a0a0au; }
    // last record whose body starts at or before the first tile's first byte (the wave's lanes search together: scalar loads)
    uint32_t lo = 0, hi = n_rec;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (rec[mid].body <= tile0 * FA_TILE) lo = mid; else hi = mid; }
#pragma unroll
    for (uint32_t q = 0; q < FA_TPW; ++q) {
        const uint64_t tile = tile0 + q, t0 = tile * FA_TILE, x = t0 + 4ull * lane;
        if (t0 >= n) break;
        const uint32_t w = ws[q];
        uint32_t brk = break_bits(w);
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) if (x + j >= n) brk |= 1u << j;    // (beyond the text: nothing to keep)
        // line-break bytes of the tile in front of this lane's four: a ballot per byte position
        uint32_t before = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) before += (uint32_t)__popcll(__ballot((brk >> j) & 1u) & ((1ull << lane) - 1ull));
        while (lo + 1 < n_rec && rec[lo + 1].body <= t0) ++lo;             // (the wave's record moves on with the tiles)
        uint32_t r = lo;
        while (r + 1 < n_rec && rec[r + 1].body <= x) ++r;                 // a tile holds the end of one record and the start of the next at most a few times
        FaRec R = rec[r];
        const uint64_t brk_x = tile_pre[tile] + before;
        if (brk == 0 && x >= R.body && x + 4 <= R.body_end) {              // four kept bytes of one record: one store (any alignment)
            *reinterpret_cast<uint32_t *>(arena + R.arena_off + (x - R.body) - (brk_x - R.breaks_before)) = w;
            continue;
        }
        uint32_t seen = 0;
        for (uint32_t j = 0; j < 4; ++j) {
            const uint64_t p = x + j;
            if ((brk >> j) & 1u) { ++seen; continue; }
            while (r + 1 < n_rec && rec[r + 1].body <= p) { ++r; R = rec[r]; }
            if (p < R.body || p >= R.body_end) continue;                   // header lines, the bytes in front of the first record
            arena[R.arena_off + (p - R.body) - (brk_x + seen - R.breaks_before)] = (uint8_t)(w >> (8 * j));
        }
    }
}

// the header lines [hdr[i], end[i]) one behind the other at dst + off[i] (a bgzipped file's text is in HBM only: the names come from here)
__global__ __launch_bounds__(64) void k_fa_hdr_copy(const uint8_t *__restrict__ raw, const uint64_t *__restrict__ hdr, const uint64_t *__restrict__ end,
                                                    const uint64_t *__restrict__ off, uint32_t n_hdr, uint8_t *__restrict__ dst) {
    const uint32_t i = blockIdx.x;
    if (i >= n_hdr) return;
    const uint64_t a = hdr[i], n = end[i] - a;
    for (uint64_t k = threadIdx.x; k < n; k += 64) dst[off[i] + k] = raw[a + k];
}

struct FaDev {
    DevBuf raw, tile_cnt, tile_pre, bsum, hdr, hdr_end, rec, kept, counter;
    DevBuf comp, hdr_off, hdr_text;                      // a bgzipped file: its bytes as they are on disk; the header lines packed for the host
    void *inflate = nullptr;                             // scratch of the device inflate (inflate.hip)
    void *pin = nullptr; size_t pin_cap = 0;
    hipStream_t st = nullptr;                            // the role's own stream: the two files of a haplotype are loaded side by side, one's
                                                         // upload and inflate beside the other's (their kernels fill a CU's LDS half each)
};

struct FaDevPair {
    FaDev role[2]; FaDev other;
    std::mutex wire;        // one BGZF file crosses PCIe at a time: the first is being inflated while the second crosses, instead of both
};                          // arriving late and the device idle till then        // one scratch per role: the two stores of a context may be loaded side by side
                                                         // (other: pav_bgzf_inflate)

// The members of a BGZF file, found while its bytes pass through the upload ring: a member's header says how long the member is,
// the next header follows it.  A header that straddles the end of a slot is read from the file.
struct BgzfWalk {
    uint64_t next = 0, size = 0;
    bool bad = false;
    BgzfMembers M;
    void feed(const uint8_t *stage, uint64_t at, uint64_t m, int fd) {
        while (!bad && next < at + m) {
            uint8_t tmp[512];
            const uint8_t *h; uint64_t have;
            if (next + sizeof(tmp) <= at + m || fd < 0) { h = stage + (next - at); have = std::min<uint64_t>(sizeof(tmp), at + m - next); }
            else {
                const ssize_t got = pread(fd, tmp, sizeof(tmp), (off_t)next);
                if (got <= 0) { bad = true; break; }
                h = tmp; have = (uint64_t)got;
            }
            // (bgzf_block wants the whole member within `avail`: the header is what is looked at, the file's size is the bound)
            uint64_t bsize = 0, hdr = 0;
            if (have < 18 || 12ull + (h[10] | (uint32_t)h[11] << 8) > have                    // (extra fields beyond 500 bytes: not bgzip's)
                || !bgzf_block(h, (size_t)std::max<uint64_t>(have, size - next), bsize, hdr)) { bad = true; break; }
            M.in_off.push_back(next + hdr); M.in_len.push_back((uint32_t)(bsize - hdr - 8));
            next += bsize;
        }
    }
};

FaDev *fstate(pav_ctx *ctx, int role) {
    static std::mutex first;                             // (the two roles' loads may be the first users of the context at the same moment)
    std::lock_guard<std::mutex> lk(first);
    if (!ctx->fa_dev) ctx->fa_dev = new FaDevPair();
    return &static_cast<FaDevPair *>(ctx->fa_dev)->role[role];
}

double wall() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// the whole file -> d_raw, through the pinned ring: `threads` readers fill a slot, the slot crosses PCIe while the next is filled
int stream_file(pav_ctx *ctx, hipStream_t st, int role, int fd, uint64_t n, uint8_t *d_raw, int threads, BgzfWalk *walk = nullptr) {
    UploadRing *R = upload_ring(ctx, role);
    if (!R->ok) return fail(ctx, PAV_E_HIP, "pav_seq_load_fasta_path: no pinned memory for the upload ring");
    // PAV_FA_MMAP=1: the pieces are copied out of a mapping of the file instead of being read (a page fault per 4 KiB and an munmap of
    // the whole file - on a thread of its own - against the kernel's copy_to_user)
    const uint8_t *map = nullptr;
    if (const char *e = getenv("PAV_FA_MMAP")) if (e[0] == '1' && n) {
        void *p = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (p != MAP_FAILED) map = static_cast<const uint8_t *>(p);
    }
    struct Unmap { const uint8_t *p; uint64_t n; ~Unmap() { if (p) { const uint8_t *q = p; const uint64_t m = n; std::thread([q, m] { munmap(const_cast<uint8_t *>(q), m); }).detach(); } } } unmap{map, n};
    for (uint64_t at = 0; at < n; at += UploadRing::SLOT_BYTES) {
        const uint64_t m = std::min<uint64_t>(UploadRing::SLOT_BYTES, n - at);
        const int k = R->next; R->next = (k + 1) % UploadRing::SLOTS;
        if (R->busy[k]) { PAV_HIP(ctx, hipEventSynchronize(R->ev[k])); R->busy[k] = false; }
        uint8_t *stage = static_cast<uint8_t *>(R->slot[k]);
        const int use = m >= (4u << 20) ? threads : 1;
        const uint64_t piece = ((m + (uint64_t)use - 1) / (uint64_t)use + 4095) & ~4095ull;
        std::atomic<int> bad{0};
        auto read_piece = [&](uint64_t a, uint64_t b) {
            if (map) { memcpy(stage + a, map + at + a, b - a); return; }
            while (a < b) {
                const ssize_t got = pread(fd, stage + a, b - a, (off_t)(at + a));
                if (got <= 0) { bad = 1; return; }
                a += (uint64_t)got;
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < use; ++t) {
            const uint64_t a = std::min<uint64_t>(m, piece * (uint64_t)t), b = std::min<uint64_t>(m, a + piece);
            if (b > a) pool.emplace_back(read_piece, a, b);
        }
        read_piece(0, std::min<uint64_t>(m, piece));
        for (auto &th : pool) th.join();
        if (bad) return fail(ctx, PAV_E_ARG, "pav_seq_load_fasta_path: read error");
        if (walk) walk->feed(stage, at, m, fd);
        PAV_HIP(ctx, hipMemcpyAsync(d_raw + at, stage, m, hipMemcpyHostToDevice, st));
        PAV_HIP(ctx, hipEventRecord(R->ev[k], st));
        R->busy[k] = true;
    }
    return PAV_OK;
}

}  // namespace

void fastadev_release(pav_ctx *ctx) {
    if (!ctx || !ctx->fa_dev) return;
    FaDevPair *P = static_cast<FaDevPair *>(ctx->fa_dev);
    for (FaDev *F : {&P->role[0], &P->role[1], &P->other}) {
        scratch_give(ctx->device, F->raw);
        scratch_give(ctx->device, F->comp);
        for (DevBuf *b : {&F->raw, &F->tile_cnt, &F->tile_pre, &F->bsum, &F->hdr, &F->hdr_end, &F->rec, &F->kept, &F->counter, &F->comp, &F->hdr_off, &F->hdr_text}) b->release();
        inflate_release(&F->inflate);
        if (F->pin) (void)hipHostFree(F->pin);
        if (F->st) (void)hipStreamDestroy(F->st);
    }
    delete P;
    ctx->fa_dev = nullptr;
}

}  // namespace pav

using namespace pav;

extern "C" {

int pav_seq_load_fasta_path(pav_ctx *ctx, int role, const char *path, int threads, uint32_t *n_records) {
    if (!ctx || !path || (role != PAV_ROLE_REF && role != PAV_ROLE_TIG)) return fail(ctx, PAV_E_ARG, "pav_seq_load_fasta_path: bad argument");
    if (threads <= 0) threads = std::min(8, default_host_threads());
    const bool timing = getenv("PAV_TIMING") != nullptr;
    const double t0 = wall();
    PAV_HIP(ctx, hipSetDevice(ctx->device));
    FaDev *F = fstate(ctx, role);
    if (!F->st) PAV_HIP(ctx, hipStreamCreateWithFlags(&F->st, hipStreamNonBlocking));
    hipStream_t st = F->st;
    // whichever way this call ends, the file's bytes and its text go back on the process's list of idle scratch (an error used to
    // leave gigabytes attached to the context until pav_destroy); the stream is drained first: nothing queued may still read them
    struct ScratchBack {
        pav_ctx *ctx; FaDev *F; hipStream_t st;
        ~ScratchBack() { (void)hipStreamSynchronize(st); (void)hipStreamSynchronize(ctx->stream); scratch_give(ctx->device, F->raw); scratch_give(ctx->device, F->comp); }
    } scratch_back{ctx, F, st};
    // ---- the text of the file into HBM --------------------------------------------------------------------------------
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(ctx, PAV_E_ARG, "pav_seq_load_fasta_path: cannot open %s", path);
    struct FdGuard { int fd; ~FdGuard() { if (fd >= 0) close(fd); } } fdg{fd};
    struct stat sb;
    if (fstat(fd, &sb) != 0) return fail(ctx, PAV_E_ARG, "pav_seq_load_fasta_path: cannot stat %s", path);
    uint8_t magic[2] = {0, 0};
    const bool compressed = pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
    FileText ft;                                             // gzip files that are not BGZF: inflated on the host first
    uint64_t n = (uint64_t)sb.st_size;
    // the previous load's kernels have run (seq_store_load waits), so the scratch can be reused
    PAV_HIP(ctx, hipStreamSynchronize(st));
    // BGZF - what PAV's own files are: the file crosses PCIe as it is, its members are inflated on the device (inflate.hip).
    // PAV_FASTA_INFLATE=host: the members are inflated by host threads as before (the A/B switch of tools/bench_e2e.py).
    bool on_device = false;
    if (compressed) {
        const char *e = getenv("PAV_FASTA_INFLATE");
        uint8_t h[512];
        const ssize_t got = pread(fd, h, sizeof h, 0);
        uint64_t bsize = 0, hdr = 0;
        if (!(e && !strcmp(e, "host")) && got >= 18 && 12ull + (h[10] | (uint32_t)h[11] << 8) <= (uint64_t)got && bgzf_block(h, (size_t)sb.st_size, bsize, hdr)) {
            BgzfWalk walk; walk.size = (uint64_t)sb.st_size;
            PAV_HIP(ctx, scratch_take(ctx->device, (size_t)sb.st_size + 4096, F->comp));
            int rc;
            { std::lock_guard<std::mutex> wire(static_cast<FaDevPair *>(ctx->fa_dev)->wire);
              // (one file at a time: its readers may be twice the usual eight - 0.081 -> 0.072 s per 3 GB file)
              rc = stream_file(ctx, st, role, fd, (uint64_t)sb.st_size, F->comp.as<uint8_t>(), std::max(threads, std::min(16, default_host_threads())), &walk); }
            if (rc != PAV_OK) return rc;
            if (!walk.bad && walk.next == (uint64_t)sb.st_size) {
                const int ri = bgzf_inflate_device(ctx, st, &F->inflate, F->comp.as<uint8_t>(), walk.M, F->raw, &n, path);
                if (ri != PAV_OK) return ri;
                on_device = true;
            } else if (timing || getenv("PAV_VERBOSE"))      // (a file that starts as BGZF and goes on as something else: the host reader decides)
                fprintf(stderr, "[pav] %s starts as BGZF but its members do not tile the file (walk stopped at byte %llu of %llu): read again and "
                                "inflated by the host reader\n", path, (unsigned long long)walk.next, (unsigned long long)sb.st_size);
        }
        if (!on_device) {
            std::string err;
            if (!read_file_text(path, default_host_threads(), ft, err)) return fail(ctx, PAV_E_ARG, "pav_seq_load_fasta_path: %s", err.c_str());
            n = ft.n;
        }
    }
    const uint32_t n_tiles = (uint32_t)((n + FA_TILE - 1) / FA_TILE);
    const uint32_t hdr_cap = 1u << 22;
    PAV_HIP(ctx, scratch_take(ctx->device, n + 4096, F->raw));
    PAV_HIP(ctx, F->tile_cnt.reserve(4ull * (n_tiles + 8)));
    PAV_HIP(ctx, F->tile_pre.reserve(8ull * (n_tiles + 8)));
    PAV_HIP(ctx, F->bsum.reserve(8ull * (n_tiles / SCAN_TILE + 8)));
    PAV_HIP(ctx, F->hdr.reserve(8ull * hdr_cap));
    PAV_HIP(ctx, F->counter.reserve(64));
    if (!F->pin) { PAV_HIP(ctx, hipHostMalloc(&F->pin, 1 << 20, hipHostMallocDefault)); F->pin_cap = 1 << 20; }
    if (on_device) {}                                        // (the text is there)
    else if (compressed) { const int rc = staged_upload(ctx, st, F->raw.as<uint8_t>(), ft.text, n, role); if (rc != PAV_OK) return rc; }
    else { const int rc = stream_file(ctx, st, role, fd, n, F->raw.as<uint8_t>(), threads); if (rc != PAV_OK) return rc; }
    const double t1 = wall();
    // ---- records ------------------------------------------------------------------------------------------------------
    PAV_HIP(ctx, hipMemsetAsync(F->counter.p, 0, 64, st));
    if (n_tiles) PAV_LAUNCH_ON(ctx, st, "k_fa_marks", k_fa_marks, (n_tiles + 4 * FA_TPW - 1) / (4 * FA_TPW), 256, 0, F->raw.as<uint8_t>(), n, F->tile_cnt.as<uint32_t>(), F->hdr.as<uint64_t>(),
                            F->counter.as<uint32_t>(), hdr_cap);
    { const int rc = scan_u32_to_u64(st, F->tile_cnt.as<uint32_t>(), n_tiles, F->bsum.as<uint64_t>(), F->tile_pre.as<uint64_t>());
      if (rc != PAV_OK) return fail(ctx, rc, "%s", pav_last_error(nullptr)); }
    uint32_t *h_n = static_cast<uint32_t *>(F->pin);
    PAV_HIP(ctx, hipMemcpyAsync(h_n, F->counter.p, 4, hipMemcpyDeviceToHost, st));
    PAV_HIP(ctx, hipStreamSynchronize(st));
    const uint32_t n_hdr = *h_n;
    if (n_hdr > hdr_cap) return fail(ctx, PAV_E_LIMIT, "pav_seq_load_fasta_path: %u records in %s (limit %u)", n_hdr, path, hdr_cap);
    std::vector<uint64_t> hdr(n_hdr), hend(n_hdr);
    if (n_hdr) {
        PAV_HIP(ctx, F->hdr_end.reserve(8ull * n_hdr));
        // (the list is in the order the workgroups met the headers: sorted here, then their ends are looked up in file order)
        PAV_HIP(ctx, hipMemcpy(hdr.data(), F->hdr.p, 8ull * n_hdr, hipMemcpyDeviceToHost));
        std::sort(hdr.begin(), hdr.end());
        PAV_HIP(ctx, hipMemcpyAsync(F->hdr.p, hdr.data(), 8ull * n_hdr, hipMemcpyHostToDevice, st));
        PAV_LAUNCH_ON(ctx, st, "k_fa_hdr_end", k_fa_hdr_end, (n_hdr + 63) / 64, 64, 0, F->raw.as<uint8_t>(), n, F->hdr.as<uint64_t>(), n_hdr, F->hdr_end.as<uint64_t>());
        PAV_HIP(ctx, hipMemcpyAsync(hend.data(), F->hdr_end.p, 8ull * n_hdr, hipMemcpyDeviceToHost, st));
        PAV_HIP(ctx, hipStreamSynchronize(st));
    }
    // names: the first whitespace-delimited word of the header line, read from the file (or the inflated text)
    std::vector<std::string> names(n_hdr);
    std::vector<FaRec> rec(n_hdr);
    std::vector<uint8_t> line;
    std::vector<uint64_t> hoff(n_hdr + 1, 0);                // text inflated on the device: the header lines come back packed, in one copy
    std::vector<uint8_t> htext;
    if (on_device && n_hdr) {
        for (uint32_t i = 0; i < n_hdr; ++i) hoff[i + 1] = hoff[i] + (hend[i] - hdr[i]);
        PAV_HIP(ctx, F->hdr_off.reserve(8ull * n_hdr));
        PAV_HIP(ctx, F->hdr_text.reserve(hoff[n_hdr] + 64));
        PAV_HIP(ctx, hipMemcpyAsync(F->hdr_off.p, hoff.data(), 8ull * n_hdr, hipMemcpyHostToDevice, st));
        PAV_LAUNCH_ON(ctx, st, "k_fa_hdr_copy", k_fa_hdr_copy, n_hdr, 64, 0, F->raw.as<uint8_t>(), F->hdr.as<uint64_t>(), F->hdr_end.as<uint64_t>(), F->hdr_off.as<uint64_t>(),
                   n_hdr, F->hdr_text.as<uint8_t>());
        htext.resize((size_t)hoff[n_hdr] + 1);
        PAV_HIP(ctx, hipMemcpyAsync(htext.data(), F->hdr_text.p, hoff[n_hdr], hipMemcpyDeviceToHost, st));
        PAV_HIP(ctx, hipStreamSynchronize(st));
    }
    for (uint32_t i = 0; i < n_hdr; ++i) {
        const uint64_t a = hdr[i] + 1, b = hend[i];
        line.resize((size_t)(b > a ? b - a : 0));
        if (!line.empty()) {
            if (on_device) memcpy(line.data(), htext.data() + hoff[i] + 1, line.size());
            else if (compressed) memcpy(line.data(), ft.text + a, line.size());
            else if (pread(fd, line.data(), line.size(), (off_t)a) != (ssize_t)line.size()) return fail(ctx, PAV_E_ARG, "pav_seq_load_fasta_path: read error in %s", path);
        }
        size_t p = 0, q;
        while (p < line.size() && (line[p] == ' ' || line[p] == '\t' || line[p] == '\r')) ++p;
        for (q = p; q < line.size() && line[q] != ' ' && line[q] != '\t' && line[q] != '\r'; ++q) {}
        names[i].assign(reinterpret_cast<const char *>(line.data()) + p, q - p);
        rec[i].body = std::min<uint64_t>(hend[i] + 1, n);
        rec[i].body_end = i + 1 < n_hdr ? hdr[i + 1] : n;
        rec[i].breaks_before = 0; rec[i].arena_off = 0;
    }
    std::vector<uint64_t> kept(n_hdr, 0);
    if (n_hdr) {
        PAV_HIP(ctx, F->rec.reserve(sizeof(FaRec) * n_hdr));
        PAV_HIP(ctx, F->kept.reserve(8ull * n_hdr));
        PAV_HIP(ctx, hipMemcpyAsync(F->rec.p, rec.data(), sizeof(FaRec) * n_hdr, hipMemcpyHostToDevice, st));
        PAV_LAUNCH_ON(ctx, st, "k_fa_records", k_fa_records, (n_hdr + 63) / 64, 64, 0, F->raw.as<uint8_t>(), F->tile_pre.as<uint64_t>(), F->rec.as<FaRec>(), n_hdr,
                   F->kept.as<uint64_t>());
        PAV_HIP(ctx, hipMemcpyAsync(kept.data(), F->kept.p, 8ull * n_hdr, hipMemcpyDeviceToHost, st));
        PAV_HIP(ctx, hipStreamSynchronize(st));
    }
    const double t2 = wall();
    // ---- the records into the store: the layout of pav_seq_load, the bytes by the strip kernel -----------------------------------
    const int rc = seq_store_load(ctx, role, n_hdr, kept.data(), "pav_seq_load_fasta_path", [&](uint8_t *arena, const std::vector<uint64_t> &off) {
        if (!n_hdr) return (int)PAV_OK;
        // arena offsets into the device records (breaks_before was written there by k_fa_records: only this column goes up)
        std::vector<FaRec> up(n_hdr);
        PAV_HIP(ctx, hipMemcpy(up.data(), F->rec.p, sizeof(FaRec) * n_hdr, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < n_hdr; ++i) up[i].arena_off = off[i];
        PAV_HIP(ctx, hipMemcpy(F->rec.p, up.data(), sizeof(FaRec) * n_hdr, hipMemcpyHostToDevice));
        PAV_LAUNCH(ctx, "k_fa_strip", k_fa_strip, (n_tiles + 4 * FA_TPW - 1) / (4 * FA_TPW), 256, 0, F->raw.as<uint8_t>(), n, F->tile_pre.as<uint64_t>(), F->rec.as<FaRec>(), n_hdr, arena);
        return (int)PAV_OK;
    });
    if (rc != PAV_OK) return rc;
    const int rcn = pav_seq_set_names_internal(ctx, role, names);
    if (rcn != PAV_OK) return rcn;
    if (n_records) *n_records = n_hdr;
    // the store is filled and the stream drained (seq_store_load): the file's bytes and its text go back on the process's list of
    // idle scratch when this function returns (scratch_back) - the other role's load, or the next haplotype's, takes them from there
    if (timing) fprintf(stderr, "[pav timing] seq_load_fasta_path role %d: %.2f GB of text%s; file -> HBM %.1f ms (%.1f GB/s), records %.1f ms, store %.1f ms (%s)\n", role,
                        (double)n / 1e9, on_device ? " (BGZF, inflated on the device)" : "", (t1 - t0) * 1e3, (double)n / 1e9 / std::max(1e-9, t1 - t0), (t2 - t1) * 1e3, (wall() - t2) * 1e3, path);
    return PAV_OK;
}

// A BGZF file held in host memory -> its text (the members inflated on the device, every CRC-32 and ISIZE checked).  *out_len is the
// text's length also when out_cap is too small for it (PAV_E_LIMIT then).
int pav_bgzf_inflate(pav_ctx *ctx, const uint8_t *in, uint64_t n_in, uint8_t *out, uint64_t out_cap, uint64_t *out_len) {
    if (!ctx || (n_in && !in) || !out_len) return fail(ctx, PAV_E_ARG, "pav_bgzf_inflate: bad argument");
    *out_len = 0;
    PAV_HIP(ctx, hipSetDevice(ctx->device));
    (void)fstate(ctx, 0);
    FaDev *F = &static_cast<FaDevPair *>(ctx->fa_dev)->other;
    BgzfWalk walk; walk.size = n_in;
    walk.feed(in, 0, n_in, -1);
    if (walk.bad || walk.next != n_in) return fail(ctx, PAV_E_ARG, "pav_bgzf_inflate: not a series of BGZF members (member %zu, byte %llu)", walk.M.in_off.size(),
                                                  (unsigned long long)walk.next);
    PAV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PAV_HIP(ctx, scratch_take(ctx->device, (size_t)n_in + 4096, F->comp));
    if (n_in) PAV_HIP(ctx, hipMemcpyAsync(F->comp.p, in, n_in, hipMemcpyHostToDevice, ctx->stream));
    struct ScratchBack {
        pav_ctx *ctx; FaDev *F;
        ~ScratchBack() { (void)hipStreamSynchronize(ctx->stream); scratch_give(ctx->device, F->raw); scratch_give(ctx->device, F->comp); }
    } scratch_back{ctx, F};
    uint64_t n = 0;
    const int rc = bgzf_inflate_device(ctx, ctx->stream, &F->inflate, F->comp.as<uint8_t>(), walk.M, F->raw, &n, "pav_bgzf_inflate");
    if (rc != PAV_OK) return rc;
    *out_len = n;
    if (n > out_cap) return fail(ctx, PAV_E_LIMIT, "pav_bgzf_inflate: %llu bytes of text, room for %llu", (unsigned long long)n, (unsigned long long)out_cap);
    if (n) PAV_HIP(ctx, hipMemcpy(out, F->raw.p, n, hipMemcpyDeviceToHost));
    return PAV_OK;
}

// ASCII bytes [pos, pos + n) of record `rec` as they stand in the store (case preserved, forward strand): a device-to-host copy
int pav_seq_fetch(pav_ctx *ctx, int role, uint32_t rec, uint64_t pos, uint64_t n, uint8_t *out) {
    if (!ctx || (role != PAV_ROLE_REF && role != PAV_ROLE_TIG) || (n && !out)) return fail(ctx, PAV_E_ARG, "pav_seq_fetch: bad argument");
    const SeqStore &s = ctx->seq[role];
    if (rec >= s.n) return fail(ctx, PAV_E_ARG, "pav_seq_fetch: record %u of %u", rec, s.n);
    if (pos > s.len[rec] || n > s.len[rec] - pos) return fail(ctx, PAV_E_ARG, "pav_seq_fetch: [%llu, +%llu) is outside the record's %llu bases",
                                                                (unsigned long long)pos, (unsigned long long)n, (unsigned long long)s.len[rec]);
    if (!n) return PAV_OK;
    PAV_HIP(ctx, hipSetDevice(ctx->device));
    PAV_HIP(ctx, hipMemcpy(out, s.d_ascii.as<uint8_t>() + s.off[rec] + pos, n, hipMemcpyDeviceToHost));
    return PAV_OK;
}

// n slices at once: slice i = [pos[i], pos[i] + len[i]) of record rec[i], written to out + (sum of len[0 .. i)).  The copies are queued
// on the context's copy stream into pinned memory and waited for once (the SEQ columns of a haplotype's INV calls: a hundred slices).
int pav_seq_fetch_many(pav_ctx *ctx, int role, uint32_t n, const uint32_t *rec, const uint64_t *pos, const uint64_t *len, uint8_t *out) {
    if (!ctx || (role != PAV_ROLE_REF && role != PAV_ROLE_TIG) || (n && (!rec || !pos || !len || !out))) return fail(ctx, PAV_E_ARG, "pav_seq_fetch_many: bad argument");
    const SeqStore &s = ctx->seq[role];
    uint64_t total = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (rec[i] >= s.n || pos[i] > s.len[rec[i]] || len[i] > s.len[rec[i]] - pos[i]) return fail(ctx, PAV_E_ARG, "pav_seq_fetch_many: slice %u is outside its record", i);
        total += len[i];
    }
    if (!total) return PAV_OK;
    PAV_HIP(ctx, hipSetDevice(ctx->device));
    FaDev *F = fstate(ctx, role);
    if (F->pin_cap < total) {
        if (F->pin) (void)hipHostFree(F->pin);
        F->pin = nullptr; F->pin_cap = 0;
        PAV_HIP(ctx, hipHostMalloc(&F->pin, total + total / 4 + (1 << 20), hipHostMallocDefault));
        F->pin_cap = total + total / 4 + (1 << 20);
    }
    uint8_t *stage = static_cast<uint8_t *>(F->pin);
    uint64_t at = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (len[i]) PAV_HIP(ctx, hipMemcpyAsync(stage + at, s.d_ascii.as<uint8_t>() + s.off[rec[i]] + pos[i], len[i], hipMemcpyDeviceToHost, ctx->stream3));
        at += len[i];
    }
    PAV_HIP(ctx, hipStreamSynchronize(ctx->stream3));
    memcpy(out, stage, total);
    return PAV_OK;
}

const char *pav_seq_name(const pav_ctx *ctx, int role, uint32_t i);       // invscan.cpp (the names live with the scan driver's state)

uint64_t pav_seq_length(const pav_ctx *ctx, int role, uint32_t i) {
    if (!ctx || (role != PAV_ROLE_REF && role != PAV_ROLE_TIG) || i >= ctx->seq[role].n) return 0;
    return ctx->seq[role].len[i];
}

}  // extern "C"
