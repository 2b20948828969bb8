// Native FASTA reader for the sequences the hot path works on: replaces the two pysam.FastaFile(...).fetch(...) uses of the
// reference (pavlib/cigarcall.py:59-66, pavlib/seq.py:339-351) and the Python gzip + numpy loader that stood in for them.
// PAV stores its FASTA files bgzipped (rules/align.snakefile: contigs_{hap}.fa.gz, data/ref/ref.fa.gz): BGZF is a series of
// independent <= 64 KiB gzip members, so the blocks are inflated in parallel; plain gzip (one stream) and uncompressed files
// are read too.  Records come back as contiguous ASCII byte arrays (line breaks removed, case preserved) - the layout
// pav_seq_load takes - and can be handed to a context directly (pav_seq_load_fasta).  Host code; no GPU needed to parse.
#include "common.h"

#include "fileio.h"

#include <chrono>
#include <memory>
#include <mutex>
#include <thread>
#include <sys/mman.h>

// Pool of gigabyte host buffers (fileio.h big_take / big_give): the text of a large plain file and the records of a FASTA file are
// recycled instead of being mapped and unmapped per file.  At most four buffers wait here (two files' text + records: a rank
// parses a contig file beside the reference's); PAV_FASTA_POOL=0: no pool (buffers are freed on a thread of their own).
namespace pav {
namespace {
struct BigPool {
    std::mutex mu;
    static constexpr int N = 4;
    struct Buf { uint8_t *p; uint64_t cap; } held[N] = {{nullptr, 0}, {nullptr, 0}, {nullptr, 0}, {nullptr, 0}};
    static bool on() { static const bool v = [] { const char *e = getenv("PAV_FASTA_POOL"); return !(e && e[0] == '0'); }(); return v; }
};
BigPool g_big;
}  // namespace

uint8_t *big_take(uint64_t want, uint64_t &cap) {
    std::lock_guard<std::mutex> g(g_big.mu);
    int best = -1;
    for (int i = 0; i < BigPool::N; ++i)
        if (g_big.held[i].p && g_big.held[i].cap >= want && (best < 0 || g_big.held[i].cap < g_big.held[best].cap)) best = i;
    if (best < 0) return nullptr;
    uint8_t *p = g_big.held[best].p; cap = g_big.held[best].cap; g_big.held[best] = BigPool::Buf{nullptr, 0};
    return p;
}

void big_give(uint8_t *p, uint64_t cap) {
    if (!p) return;
    uint8_t *drop = p;
    if (BigPool::on()) {
        std::lock_guard<std::mutex> g(g_big.mu);
        int slot = -1;
        for (int i = 0; i < BigPool::N; ++i) if (!g_big.held[i].p) { slot = i; break; }
        if (slot < 0) {                                                  // full: the smallest one makes room if this one is larger
            slot = 0;
            for (int i = 1; i < BigPool::N; ++i) if (g_big.held[i].cap < g_big.held[slot].cap) slot = i;
            if (g_big.held[slot].cap >= cap) slot = -1;
        }
        if (slot >= 0) { drop = g_big.held[slot].p; g_big.held[slot] = BigPool::Buf{p, cap}; }
    }
    if (drop) std::thread([drop] { free(drop); }).detach();              // (unmapping gigabytes takes 0.1 - 0.2 s: not on the caller's time)
}
}  // namespace pav

struct pav_fasta {
    std::vector<std::string> names;
    std::vector<uint64_t> off, len;          // record i = seq[off[i], off[i] + len[i])
    uint8_t *seq = nullptr;
    uint64_t bytes = 0, cap = 0;
    int kind = 0;                            // 0 plain, 1 gzip (one stream), 2 BGZF (blocks inflated in parallel)
    ~pav_fasta() { pav::big_give(seq, cap); }
};

using namespace pav;

extern "C" {

int pav_fasta_open(const char *path, int threads, pav_fasta **out) {
    if (!path || !out) return PAV_E_ARG;
    *out = nullptr;
    if (threads <= 0) threads = default_host_threads();
    const bool timing = getenv("PAV_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_mark = now();
    auto lap = [&](const char *what) { if (timing) { const double t = now(); fprintf(stderr, "[pav timing] fasta %-10s %.3f s (%s)\n", what, t - t_mark, path); t_mark = t; } };
    // (the mapping of the file is let go on a thread of its own when the records have been copied: see the end)
    std::unique_ptr<FileText> ftp(new FileText());
    FileText &ft = *ftp;
    std::string err;
    if (!read_file_text(path, threads, ft, err)) return fail(nullptr, PAV_E_ARG, "pav_fasta_open: %s", err.c_str());
    lap("read");
    auto fa = new pav_fasta();
    fa->kind = ft.kind;
    const uint8_t *text = ft.text;
    const uint64_t n_text = ft.n;

    // ---- records: '>' at the start of a line -------------------------------------------------------------------
    struct Rec { uint64_t body, body_end; };
    std::vector<Rec> recs;
    // header positions: every chunk of the text is searched by its own thread (one memchr over 3 GB was 0.17 s), the headers are
    // then read in file order
    std::vector<uint64_t> heads;
    {
        constexpr uint64_t SCAN = 64ull << 20;
        const size_t n_scan = (size_t)((n_text + SCAN - 1) / SCAN);
        std::vector<std::vector<uint64_t>> found(n_scan);
        parallel_for(n_scan, threads, [&](size_t c) {
            const uint64_t a = (uint64_t)c * SCAN, b = std::min(n_text, a + SCAN);
            for (uint64_t at = a; at < b;) {
                const uint8_t *g = static_cast<const uint8_t *>(memchr(text + at, '>', b - at));
                if (!g) break;
                const uint64_t s0 = (uint64_t)(g - text);
                if (s0 == 0 || text[s0 - 1] == '\n') found[c].push_back(s0);
                at = s0 + 1;
            }
        });
        for (const auto &v : found) heads.insert(heads.end(), v.begin(), v.end());
    }
    for (const uint64_t s0 : heads) {
        const uint8_t *nl = static_cast<const uint8_t *>(memchr(text + s0, '\n', n_text - s0));
        const uint64_t hdr_end = nl ? (uint64_t)(nl - text) : n_text;
        uint64_t a = s0 + 1, b = a;                              // name = first whitespace-delimited word of the header
        while (a < hdr_end && (text[a] == ' ' || text[a] == '\t' || text[a] == '\r')) ++a;
        for (b = a; b < hdr_end && text[b] != ' ' && text[b] != '\t' && text[b] != '\r'; ++b) {}
        fa->names.emplace_back(reinterpret_cast<const char *>(text + a), (size_t)(b - a));
        if (!recs.empty()) recs.back().body_end = s0;
        recs.push_back(Rec{std::min<uint64_t>(hdr_end + 1, n_text), n_text});
    }
    lap("records");
    // ---- bodies without line breaks: count per chunk, prefix, copy (both passes in parallel) -------------------
    constexpr uint64_t CHUNK = 4ull << 20;
    struct Piece { uint32_t rec; uint64_t a, b, keep, out; };
    std::vector<Piece> pieces;
    for (uint32_t r = 0; r < recs.size(); ++r)
        for (uint64_t a = recs[r].body; a < recs[r].body_end || a == recs[r].body; a += CHUNK) {
            pieces.push_back(Piece{r, a, std::min(a + CHUNK, recs[r].body_end), 0, 0});
            if (a + CHUNK >= recs[r].body_end) break;
        }
    parallel_for(pieces.size(), threads, [&](size_t i) {
        Piece &pc = pieces[i];
        uint64_t brk = 0;
        for (uint64_t q = pc.a; q < pc.b; ++q) brk += (text[q] == '\n') | (text[q] == '\r');
        pc.keep = pc.b - pc.a - brk;
    });
    lap("count");
    fa->off.assign(recs.size(), 0);
    fa->len.assign(recs.size(), 0);
    uint64_t total = 0;
    for (size_t i = 0; i < pieces.size(); ++i) {
        Piece &pc = pieces[i];
        if (i == 0 || pieces[i - 1].rec != pc.rec) { total = (total + 63) & ~63ull; fa->off[pc.rec] = total; }   // records start 64 B aligned
        pc.out = total;
        total += pc.keep;
        fa->len[pc.rec] += pc.keep;
    }
    fa->bytes = total;
    {   // 2 MiB alignment + MADV_HUGEPAGE: the copy pass below touches every page of a fresh 3 GB buffer for the first time - with
        // 4 KiB pages that is 760 k page faults per genome, and they, not the copying, were most of the pass (0.52 s of 0.75 s)
        const uint64_t want = (std::max<uint64_t>(total, 1) + (2ull << 20) - 1) & ~((2ull << 20) - 1);
        fa->seq = big_take(want, fa->cap);
        if (!fa->seq) {
            fa->seq = static_cast<uint8_t *>(aligned_alloc(2ull << 20, want));
            fa->cap = want;
            if (fa->seq) (void)madvise(fa->seq, want, MADV_HUGEPAGE);
        }
    }
    if (!fa->seq) { delete fa; return fail(nullptr, PAV_E_ARG, "pav_fasta_open: out of memory (%llu sequence bytes)", (unsigned long long)total); }
    parallel_for(pieces.size(), threads, [&](size_t i) {
        const Piece &pc = pieces[i];
        uint8_t *o = fa->seq + pc.out;
        for (uint64_t q = pc.a; q < pc.b;) {
            const uint8_t *nl = static_cast<const uint8_t *>(memchr(text + q, '\n', pc.b - q));
            uint64_t e = nl ? (uint64_t)(nl - text) : pc.b;
            uint64_t span_end = e;
            // '\r' is dropped wherever it stands (CRLF files; same as the byte filter of the numpy loader this replaces)
            for (uint64_t s = q; s < span_end;) {
                const uint8_t *cr = static_cast<const uint8_t *>(memchr(text + s, '\r', span_end - s));
                const uint64_t c = cr ? (uint64_t)(cr - text) : span_end;
                memcpy(o, text + s, c - s);
                o += c - s;
                s = c + 1;
            }
            q = e + 1;
        }
    });
    lap("copy");
    ftp.reset();                                             // the text goes back to the pool (a mapped file: unmapped)
    lap("release");
    *out = fa;
    return PAV_OK;
}

uint32_t pav_fasta_count(const pav_fasta *fa) { return fa ? (uint32_t)fa->names.size() : 0; }
const char *pav_fasta_name(const pav_fasta *fa, uint32_t i) { return fa && i < fa->names.size() ? fa->names[i].c_str() : nullptr; }
uint64_t pav_fasta_length(const pav_fasta *fa, uint32_t i) { return fa && i < fa->len.size() ? fa->len[i] : 0; }
const uint8_t *pav_fasta_seq(const pav_fasta *fa, uint32_t i) { return fa && i < fa->off.size() ? fa->seq + fa->off[i] : nullptr; }
int pav_fasta_kind(const pav_fasta *fa) { return fa ? fa->kind : -1; }
void pav_fasta_close(pav_fasta *fa) {
    const bool timing = getenv("PAV_TIMING") != nullptr;
    const double t0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    delete fa;
    if (timing) fprintf(stderr, "[pav timing] fasta close      %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0);
}

int pav_seq_load_fasta(pav_ctx *ctx, int role, const pav_fasta *fa, uint32_t n, const uint32_t *records) {
    if (!ctx || !fa || (n && !records)) return fail(ctx, PAV_E_ARG, "pav_seq_load_fasta: null argument");
    std::vector<const uint8_t *> ptr(std::max<uint32_t>(n, 1));
    std::vector<uint64_t> len(std::max<uint32_t>(n, 1));
    std::vector<const char *> name(std::max<uint32_t>(n, 1));
    for (uint32_t i = 0; i < n; ++i) {
        if (records[i] >= fa->names.size()) return fail(ctx, PAV_E_ARG, "pav_seq_load_fasta: record %u does not exist", records[i]);
        ptr[i] = fa->seq + fa->off[records[i]];
        len[i] = fa->len[records[i]];
        name[i] = fa->names[records[i]].c_str();
    }
    const int rc = pav_seq_load(ctx, role, n, ptr.data(), len.data());
    if (rc != PAV_OK) return rc;
    return pav_seq_set_names(ctx, role, n, name.data());
}

}  // extern "C"
