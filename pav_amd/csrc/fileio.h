// Whole-file text loading shared by the host-side readers (fastaio.cpp, samio.cpp): plain files are mapped, gzip streams are
// inflated, BGZF files (a series of independent <= 64 KiB gzip members, SAM specification 4.1) are inflated block-parallel.
#pragma once

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

namespace pav {

template <class F> static void parallel_for(size_t n, int threads, F &&body) {
    threads = (int)std::max<size_t>(1, std::min<size_t>((size_t)threads, n));
    if (threads == 1) { for (size_t i = 0; i < n; ++i) body(i); return; }
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&] { for (size_t i; (i = next.fetch_add(1)) < n;) body(i); });
    for (auto &th : pool) th.join();
}

// Host threads for the parallel readers / writers: the CPUs this process may really use - hardware_concurrency() reports the
// host's cores inside a container whose cgroup quota (cpu.max) allows a fraction of them - at most 32.
static inline int default_host_threads() {
    unsigned n = std::max<unsigned>(std::thread::hardware_concurrency(), 1);
    if (FILE *fh = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32] = {0};
        long long period = 0;
        if (fscanf(fh, "%31s %lld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0) {
            const long long q = atoll(quota) / period;
            if (q >= 1) n = std::min<unsigned>(n, (unsigned)q);
        }
        fclose(fh);
    }
    return (int)std::min<unsigned>(n, 32);
}

// Gigabyte host buffers (the text of a file, the records of a FASTA file) are recycled through a small process-wide pool
// (fastaio.cpp): a fresh 3 GB buffer costs 760 k page faults on the way in and 0.1 - 0.2 s of unmapping on the way out (measured on
// the MI355X box), and a cohort rank opens one contig file per haplotype.
uint8_t *big_take(uint64_t want, uint64_t &cap);         // a buffer of at least `want` bytes (2 MiB aligned), or nullptr
void big_give(uint8_t *p, uint64_t cap);                 // back to the pool (or freed, on a thread of its own)

struct FileText {
    const uint8_t *text = nullptr;           // the decoded bytes of the file
    uint64_t n = 0;
    int kind = 0;                            // 0 plain, 1 gzip (one stream), 2 BGZF (blocks inflated in parallel)
    // owners
    const uint8_t *map_p = nullptr; size_t map_n = 0; int fd = -1;
    uint8_t *heap = nullptr;
    uint8_t *pooled = nullptr; uint64_t pooled_cap = 0;
    std::vector<uint8_t> inflated;
    FileText() = default;
    FileText(const FileText &) = delete;
    FileText &operator=(const FileText &) = delete;
    ~FileText() { release(); }
    void release() {
        if (map_p && map_n) munmap(const_cast<uint8_t *>(map_p), map_n);
        if (fd >= 0) close(fd);
        free(heap);
        if (pooled) big_give(pooled, pooled_cap);
        map_p = nullptr; map_n = 0; fd = -1; heap = nullptr; pooled = nullptr; pooled_cap = 0; text = nullptr; n = 0;
        std::vector<uint8_t>().swap(inflated);
    }
};

// BGZF block header: gzip member with FEXTRA carrying subfield 'B','C' = total block size - 1.
static inline bool bgzf_block(const uint8_t *p, size_t avail, uint64_t &bsize, uint64_t &hdr) {
    if (avail < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return false;
    const uint32_t xlen = p[10] | (uint32_t)p[11] << 8;
    if (avail < 12ull + xlen) return false;
    for (uint32_t x = 0; x + 4 <= xlen;) {
        const uint8_t *f = p + 12 + x;
        const uint32_t slen = f[2] | (uint32_t)f[3] << 8;
        if (f[0] == 'B' && f[1] == 'C' && slen == 2 && x + 6 <= xlen) {
            bsize = (uint64_t)(f[4] | (uint32_t)f[5] << 8) + 1;
            hdr = 12ull + xlen;
            return bsize >= hdr + 8 && bsize <= avail;
        }
        x += 4 + slen;
    }
    return false;
}

// One gzip stream, possibly several concatenated members (what `gzip` and `cat a.gz b.gz` produce).
static inline bool inflate_stream(const uint8_t *in, size_t n, std::vector<uint8_t> &out, std::string &err) {
    z_stream z{};
    if (inflateInit2(&z, 15 + 16) != Z_OK) { err = "inflateInit2 failed"; return false; }
    out.resize(std::max<size_t>(n * 4, 1 << 20));
    size_t produced = 0;
    z.next_in = const_cast<Bytef *>(in);
    size_t left = n;
    for (;;) {
        z.avail_in = (uInt)std::min<size_t>(left, 1u << 30);
        const size_t fed = z.avail_in;
        if (out.size() - produced < (1u << 20)) out.resize(out.size() * 2);
        z.next_out = out.data() + produced;
        z.avail_out = (uInt)std::min<size_t>(out.size() - produced, 1u << 30);
        const size_t room = z.avail_out;
        const int rc = inflate(&z, Z_NO_FLUSH);
        left -= fed - z.avail_in;
        produced += room - z.avail_out;
        if (rc == Z_STREAM_END) {
            if (left == 0) break;
            if (inflateReset(&z) != Z_OK) { err = "inflateReset failed"; inflateEnd(&z); return false; }
            continue;
        }
        if (rc != Z_OK && rc != Z_BUF_ERROR) { err = std::string("inflate: ") + (z.msg ? z.msg : "corrupt data"); inflateEnd(&z); return false; }
        if (rc == Z_BUF_ERROR && left == 0 && z.avail_out != 0) { err = "truncated gzip stream"; inflateEnd(&z); return false; }
    }
    inflateEnd(&z);
    out.resize(produced);
    return true;
}

// Load `path` into `ft`; false + message on failure.
static inline bool read_file_text(const char *path, int threads, FileText &ft, std::string &err) {
    ft.release();
    ft.fd = open(path, O_RDONLY);
    if (ft.fd < 0) { err = std::string("cannot open ") + path; return false; }
    struct stat sb;
    if (fstat(ft.fd, &sb) != 0) { err = std::string("cannot stat ") + path; return false; }
    ft.map_n = (size_t)sb.st_size;
    {   // A plain-text file of some size is READ, in parallel pieces, into a recycled buffer: mapping it instead cost a page fault
        // per 4 KiB on the way through and 0.06 - 0.14 s of munmap at the end (3 GB FASTA, MI355X box), and the unmapping holds the
        // process's memory-map lock against every other thread.  (Compressed files are small and stay mapped.)
        uint8_t magic[2] = {0, 0};
        const bool gz = pread(ft.fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
        if (!gz && ft.map_n >= (64u << 20)) {
            const uint64_t n = ft.map_n, want = (n + (2ull << 20)) & ~((2ull << 20) - 1);
            ft.pooled = big_take(want, ft.pooled_cap);
            if (!ft.pooled) {
                ft.pooled = static_cast<uint8_t *>(aligned_alloc(2ull << 20, want));
                ft.pooled_cap = want;
                if (ft.pooled) (void)madvise(ft.pooled, want, MADV_HUGEPAGE);
            }
            if (ft.pooled) {
                constexpr uint64_t PIECE = 16ull << 20;
                std::atomic<int> bad{0};
                uint8_t *dst = ft.pooled; const int fd = ft.fd;
                parallel_for((size_t)((n + PIECE - 1) / PIECE), threads, [&](size_t c) {
                    uint64_t at = (uint64_t)c * PIECE; const uint64_t end = std::min(n, at + PIECE);
                    while (at < end) {
                        const ssize_t got = pread(fd, dst + at, end - at, (off_t)at);
                        if (got <= 0) { bad = 1; return; }
                        at += (uint64_t)got;
                    }
                });
                if (bad) { err = std::string("cannot read ") + path; return false; }
                ft.map_n = 0;
                ft.text = ft.pooled; ft.n = n; ft.kind = 0;
                return true;
            }
        }
    }
    if (ft.map_n) {
        void *p = mmap(nullptr, ft.map_n, PROT_READ, MAP_PRIVATE, ft.fd, 0);
        if (p == MAP_FAILED) { ft.map_n = 0; err = std::string("cannot map ") + path; return false; }
        ft.map_p = static_cast<const uint8_t *>(p);
    }
    ft.text = ft.map_p; ft.n = ft.map_n; ft.kind = 0;
    if (!(ft.map_n >= 2 && ft.map_p[0] == 0x1f && ft.map_p[1] == 0x8b)) return true;
    struct Block { uint64_t in_off, in_len, out_off, out_len; };   // deflate payload of one BGZF block, its place in the text
    std::vector<Block> blocks;
    uint64_t at = 0, total = 0, bsize = 0, hdr = 0;
    bool bgzf = true;
    while (at < ft.map_n) {
        if (!bgzf_block(ft.map_p + at, ft.map_n - at, bsize, hdr)) { bgzf = false; break; }
        const uint8_t *tail = ft.map_p + at + bsize - 4;
        const uint64_t isize = tail[0] | (uint64_t)tail[1] << 8 | (uint64_t)tail[2] << 16 | (uint64_t)tail[3] << 24;
        blocks.push_back(Block{at + hdr, bsize - hdr - 8, total, isize});
        total += isize;
        at += bsize;
    }
    if (bgzf) {
        ft.kind = 2;
        ft.heap = static_cast<uint8_t *>(malloc(std::max<uint64_t>(total, 1)));
        if (!ft.heap) { err = "out of memory (" + std::to_string(total) + " bytes of text)"; return false; }
        std::atomic<int> bad{0};
        constexpr size_t STRIPE = 64;                        // blocks per work item
        const uint8_t *src = ft.map_p;
        uint8_t *dst = ft.heap;
        parallel_for((blocks.size() + STRIPE - 1) / STRIPE, threads, [&](size_t s) {
            z_stream z{};
            if (inflateInit2(&z, -15) != Z_OK) { bad = 1; return; }
            for (size_t b = s * STRIPE; b < std::min(blocks.size(), (s + 1) * STRIPE); ++b) {
                const Block &k = blocks[b];
                z.next_in = const_cast<Bytef *>(src + k.in_off); z.avail_in = (uInt)k.in_len;
                z.next_out = dst + k.out_off; z.avail_out = (uInt)k.out_len;
                const int rc = k.out_len || k.in_len > 2 ? inflate(&z, Z_FINISH) : Z_STREAM_END;
                if (rc != Z_STREAM_END || z.avail_out != 0) bad = 1;
                inflateReset(&z);
            }
            inflateEnd(&z);
        });
        if (bad) { err = std::string("corrupt BGZF block in ") + path; return false; }
        ft.text = ft.heap; ft.n = total;
        return true;
    }
    ft.kind = 1;
    std::string ierr;
    if (!inflate_stream(ft.map_p, ft.map_n, ft.inflated, ierr)) { err = std::string(path) + ": " + ierr; return false; }
    ft.text = ft.inflated.data(); ft.n = ft.inflated.size();
    return true;
}

}  // namespace pav
