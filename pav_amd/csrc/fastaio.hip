// Native FASTA reader for the sequences the hot path works on: replaces the two pysam.FastaFile(...).fetch(...) uses of the
// reference (pavlib/cigarcall.py:59-66, pavlib/seq.py:339-351) and the Python gzip + numpy loader that stood in for them.
// PAV stores its FASTA files bgzipped (rules/align.snakefile: contigs_{hap}.fa.gz, data/ref/ref.fa.gz): BGZF is a series of
// independent <= 64 KiB gzip members, so the blocks are inflated in parallel; plain gzip (one stream) and uncompressed files
// are read too.  Records come back as contiguous ASCII byte arrays (line breaks removed, case preserved) - the layout
// pav_seq_load takes - and can be handed to a context directly (pav_seq_load_fasta).  Host code; no GPU needed to parse.
#include "common.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <thread>

struct pav_fasta {
    std::vector<std::string> names;
    std::vector<uint64_t> off, len;          // record i = seq[off[i], off[i] + len[i])
    uint8_t *seq = nullptr;
    uint64_t bytes = 0;
    int kind = 0;                            // 0 plain, 1 gzip (one stream), 2 BGZF (blocks inflated in parallel)
    ~pav_fasta() { free(seq); }
};

namespace pav {
namespace {

struct Mapped {
    const uint8_t *p = nullptr;
    size_t n = 0;
    int fd = -1;
    ~Mapped() { if (p && n) munmap(const_cast<uint8_t *>(p), n); if (fd >= 0) close(fd); }
};

struct Block { uint64_t in_off, in_len, out_off, out_len; };   // deflate payload of one BGZF block, its place in the text

// BGZF block header (SAM specification 4.1): gzip member with FEXTRA carrying subfield 'B','C' = total block size - 1.
bool bgzf_block(const uint8_t *p, size_t avail, uint64_t &bsize, uint64_t &hdr) {
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

template <class F> void parallel_for(size_t n, int threads, F &&body) {
    threads = (int)std::max<size_t>(1, std::min<size_t>((size_t)threads, n));
    if (threads == 1) { for (size_t i = 0; i < n; ++i) body(i); return; }
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&] { for (size_t i; (i = next.fetch_add(1)) < n;) body(i); });
    for (auto &th : pool) th.join();
}

// One gzip stream, possibly several concatenated members (what `gzip` and `cat a.gz b.gz` produce).
bool inflate_stream(const uint8_t *in, size_t n, std::vector<uint8_t> &out, std::string &err) {
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

}  // namespace
}  // namespace pav

using namespace pav;

extern "C" {

int pav_fasta_open(const char *path, int threads, pav_fasta **out) {
    if (!path || !out) return PAV_E_ARG;
    *out = nullptr;
    if (threads <= 0) threads = (int)std::min<unsigned>(std::max<unsigned>(std::thread::hardware_concurrency(), 1), 32);
    Mapped m;
    m.fd = open(path, O_RDONLY);
    if (m.fd < 0) return fail(nullptr, PAV_E_ARG, "pav_fasta_open: cannot open %s", path);
    struct stat sb;
    if (fstat(m.fd, &sb) != 0) return fail(nullptr, PAV_E_ARG, "pav_fasta_open: cannot stat %s", path);
    m.n = (size_t)sb.st_size;
    if (m.n) {
        void *p = mmap(nullptr, m.n, PROT_READ, MAP_PRIVATE, m.fd, 0);
        if (p == MAP_FAILED) { m.n = 0; return fail(nullptr, PAV_E_ARG, "pav_fasta_open: cannot map %s", path); }
        m.p = static_cast<const uint8_t *>(p);
    }
    auto fa = new pav_fasta();

    // ---- text of the file ------------------------------------------------------------------------------------
    const uint8_t *text = m.p;
    uint64_t n_text = m.n;
    std::vector<uint8_t> inflated;
    uint8_t *bg_text = nullptr;
    if (m.n >= 2 && m.p[0] == 0x1f && m.p[1] == 0x8b) {
        std::vector<Block> blocks;
        uint64_t at = 0, total = 0, bsize = 0, hdr = 0;
        bool bgzf = true;
        while (at < m.n) {
            if (!bgzf_block(m.p + at, m.n - at, bsize, hdr)) { bgzf = false; break; }
            const uint8_t *tail = m.p + at + bsize - 4;
            const uint64_t isize = tail[0] | (uint64_t)tail[1] << 8 | (uint64_t)tail[2] << 16 | (uint64_t)tail[3] << 24;
            blocks.push_back(Block{at + hdr, bsize - hdr - 8, total, isize});
            total += isize;
            at += bsize;
        }
        if (bgzf) {
            fa->kind = 2;
            bg_text = static_cast<uint8_t *>(malloc(std::max<uint64_t>(total, 1)));
            if (!bg_text) { delete fa; return fail(nullptr, PAV_E_ARG, "pav_fasta_open: out of memory (%llu bytes of text)", (unsigned long long)total); }
            std::atomic<int> bad{0};
            constexpr size_t STRIPE = 64;                        // blocks per work item
            parallel_for((blocks.size() + STRIPE - 1) / STRIPE, threads, [&](size_t s) {
                z_stream z{};
                if (inflateInit2(&z, -15) != Z_OK) { bad = 1; return; }
                for (size_t b = s * STRIPE; b < std::min(blocks.size(), (s + 1) * STRIPE); ++b) {
                    const Block &k = blocks[b];
                    z.next_in = const_cast<Bytef *>(m.p + k.in_off); z.avail_in = (uInt)k.in_len;
                    z.next_out = bg_text + k.out_off; z.avail_out = (uInt)k.out_len;
                    const int rc = k.out_len || k.in_len > 2 ? inflate(&z, Z_FINISH) : Z_STREAM_END;
                    if (rc != Z_STREAM_END || z.avail_out != 0) bad = 1;
                    inflateReset(&z);
                }
                inflateEnd(&z);
            });
            if (bad) { free(bg_text); delete fa; return fail(nullptr, PAV_E_ARG, "pav_fasta_open: corrupt BGZF block in %s", path); }
            text = bg_text; n_text = total;
        } else {
            fa->kind = 1;
            std::string err;
            if (!inflate_stream(m.p, m.n, inflated, err)) { delete fa; return fail(nullptr, PAV_E_ARG, "pav_fasta_open: %s: %s", path, err.c_str()); }
            text = inflated.data(); n_text = inflated.size();
        }
    }

    // ---- records: '>' at the start of a line -------------------------------------------------------------------
    struct Rec { uint64_t body, body_end; };
    std::vector<Rec> recs;
    for (uint64_t at = 0; at < n_text;) {
        const uint8_t *g = static_cast<const uint8_t *>(memchr(text + at, '>', n_text - at));
        if (!g) break;
        const uint64_t s = (uint64_t)(g - text);
        at = s + 1;
        if (s != 0 && text[s - 1] != '\n') continue;
        const uint8_t *nl = static_cast<const uint8_t *>(memchr(g, '\n', n_text - s));
        const uint64_t hdr_end = nl ? (uint64_t)(nl - text) : n_text;
        uint64_t a = s + 1, b = a;                               // name = first whitespace-delimited word of the header
        while (a < hdr_end && (text[a] == ' ' || text[a] == '\t' || text[a] == '\r')) ++a;
        for (b = a; b < hdr_end && text[b] != ' ' && text[b] != '\t' && text[b] != '\r'; ++b) {}
        fa->names.emplace_back(reinterpret_cast<const char *>(text + a), (size_t)(b - a));
        if (!recs.empty()) recs.back().body_end = s;
        recs.push_back(Rec{std::min<uint64_t>(hdr_end + 1, n_text), n_text});
        at = hdr_end;
    }

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
    fa->seq = static_cast<uint8_t *>(aligned_alloc(64, (std::max<uint64_t>(total, 1) + 63) & ~63ull));
    if (!fa->seq) { free(bg_text); delete fa; return fail(nullptr, PAV_E_ARG, "pav_fasta_open: out of memory (%llu sequence bytes)", (unsigned long long)total); }
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
    free(bg_text);
    *out = fa;
    return PAV_OK;
}

uint32_t pav_fasta_count(const pav_fasta *fa) { return fa ? (uint32_t)fa->names.size() : 0; }
const char *pav_fasta_name(const pav_fasta *fa, uint32_t i) { return fa && i < fa->names.size() ? fa->names[i].c_str() : nullptr; }
uint64_t pav_fasta_length(const pav_fasta *fa, uint32_t i) { return fa && i < fa->len.size() ? fa->len[i] : 0; }
const uint8_t *pav_fasta_seq(const pav_fasta *fa, uint32_t i) { return fa && i < fa->off.size() ? fa->seq + fa->off[i] : nullptr; }
int pav_fasta_kind(const pav_fasta *fa) { return fa ? fa->kind : -1; }
void pav_fasta_close(pav_fasta *fa) { delete fa; }

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
