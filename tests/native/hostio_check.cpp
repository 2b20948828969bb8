// Host-only check of the file readers of libpav_amd (fastaio.cpp, bedio.cpp, samio.cpp) - the ~700 lines that parse files the
// library does not control.  tests/test_host_sanitize.py builds this driver TOGETHER WITH those three sources with
// g++ -fsanitize=address,undefined (CPU build; no GPU, no HIP call is made) and runs it on well-formed and on damaged files:
// a reader must either parse the file or refuse it with a message - never read or write outside its buffers.
//   hostio_check fasta|bed|sam <path>   ->  one line: "ok <digest fields>" or "error <message>"; exit status 0 in both cases
// (a sanitizer report ends the process with a non-zero status).  The digests are compared with a Python reading of the same
// file by the test.
#include "../../include/pav_amd.h"

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <zlib.h>

// ---- what the three sources expect from the rest of the library (ctx.hip / cigar.hip): never reached by this driver ------
struct pav_ctx { std::string err; };
namespace pav {
static std::string g_err;
int fail(pav_ctx *ctx, int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    g_err = buf;
    return code;
}
const std::vector<std::string> &seq_names(pav_ctx *, int) { static std::vector<std::string> none; return none; }
}  // namespace pav
extern "C" {
const char *pav_last_error(const pav_ctx *ctx) { return ctx ? ctx->err.c_str() : pav::g_err.c_str(); }
int pav_seq_load(pav_ctx *, int, uint32_t, const uint8_t *const *, const uint64_t *) { return PAV_E_STATE; }
int pav_seq_set_names(pav_ctx *, int, uint32_t, const char *const *) { return PAV_E_STATE; }
int pav_cigar_load(pav_ctx *, uint32_t, const pav_aln *, const uint8_t *, const uint64_t *) { return PAV_E_STATE; }
}

static int check_fasta(const char *path) {
    pav_fasta *fa = nullptr;
    const int rc = pav_fasta_open(path, 4, &fa);
    if (rc != PAV_OK) { printf("error %s\n", pav_last_error(nullptr)); return 0; }
    const uint32_t n = pav_fasta_count(fa);
    uint64_t total = 0;
    uLong crc = crc32(0L, Z_NULL, 0), ncrc = crc32(0L, Z_NULL, 0);
    for (uint32_t i = 0; i < n; ++i) {
        const char *name = pav_fasta_name(fa, i);
        const uint64_t len = pav_fasta_length(fa, i);
        const uint8_t *s = pav_fasta_seq(fa, i);
        ncrc = crc32(ncrc, (const Bytef *)name, (uInt)strlen(name) + 1);
        for (uint64_t a = 0; a < len; a += 1u << 30) crc = crc32(crc, s + a, (uInt)std::min<uint64_t>(len - a, 1u << 30));
        total += len;
    }
    printf("ok kind=%d records=%u bases=%llu names_crc=%08lx seq_crc=%08lx\n", pav_fasta_kind(fa), n, (unsigned long long)total, ncrc, crc);
    pav_fasta_close(fa);
    return 0;
}

static int check_bed(const char *path) {
    pav_bed *bed = nullptr;
    const int rc = pav_bed_open(path, 1, &bed);
    if (rc != PAV_OK) { printf("error %s\n", pav_last_error(nullptr)); return 0; }
    pav_bed_info_t info;
    pav_bed_info(bed, &info);
    const size_t n = (size_t)info.n_rows;
    std::vector<uint32_t> chrom(n), qry(n);
    std::vector<int64_t> pos(n), end(n), index(n), qpos(n), qend(n), qlen(n), mapq(n), batch(n);
    std::vector<uint8_t> rev(n), text((size_t)info.cigar_bytes + 1);
    std::vector<uint64_t> off(n + 1);
    pav_bed_cols cols = {chrom.data(), qry.data(), pos.data(), end.data(), index.data(), qpos.data(), qend.data(), qlen.data(),
                         mapq.data(), batch.data(), rev.data(), text.data(), off.data()};
    if (pav_bed_fetch(bed, &cols) != PAV_OK) { printf("error %s\n", pav_last_error(nullptr)); pav_bed_close(bed); return 0; }
    long long spos = 0, send = 0;
    for (size_t i = 0; i < n; ++i) { spos += pos[i]; send += end[i]; }
    uLong ncrc = crc32(0L, Z_NULL, 0);
    for (uint32_t i = 0; i < info.n_chrom; ++i) { const char *s = pav_bed_name(bed, 0, i); ncrc = crc32(ncrc, (const Bytef *)s, (uInt)strlen(s) + 1); }
    for (uint32_t i = 0; i < info.n_qry; ++i) { const char *s = pav_bed_name(bed, 1, i); ncrc = crc32(ncrc, (const Bytef *)s, (uInt)strlen(s) + 1); }
    const uLong ccrc = crc32(crc32(0L, Z_NULL, 0), text.data(), (uInt)info.cigar_bytes);
    printf("ok rows=%llu chroms=%u qrys=%u columns=%x pos=%lld end=%lld cigar_bytes=%llu cigar_crc=%08lx names_crc=%08lx\n",
           (unsigned long long)info.n_rows, info.n_chrom, info.n_qry, info.columns, spos, send, (unsigned long long)info.cigar_bytes, ccrc, ncrc);
    pav_bed_close(bed);
    return 0;
}

static int check_sam(const char *path) {
    pav_sam *sam = nullptr;
    const int rc = pav_sam_open(path, 0, 4, &sam);
    if (rc != PAV_OK) { printf("error %s\n", pav_last_error(nullptr)); return 0; }
    pav_sam_info_t info;
    pav_sam_info(sam, &info);
    const size_t n = (size_t)info.n_rows;
    std::vector<int64_t> index(n), pos(n), end(n), qas(n), qae(n), clip(n), tmp(n), refbp(n), tigbp(n);
    std::vector<uint32_t> chrom(n), qry(n), ek(n), eo(n), el(n), ec(n);
    std::vector<int32_t> mapq(n), flag(n);
    std::vector<uint8_t> has_m(n), status(n), text((size_t)info.cigar_bytes + 1), tags((size_t)info.tag_bytes + 1), rgk(n), aok(n),
        head((size_t)info.header_bytes + 1);
    std::vector<uint64_t> off(n + 1), rgo(n + 1), aoo(n + 1);
    pav_sam_cols cols = {index.data(), pos.data(), end.data(), chrom.data(), qry.data(), qas.data(), qae.data(), clip.data(), tmp.data(),
                         mapq.data(), flag.data(), has_m.data(), status.data(), refbp.data(), tigbp.data(), ek.data(), eo.data(),
                         el.data(), ec.data(), text.data(), off.data(), tags.data(), rgo.data(), aoo.data(), rgk.data(), aok.data()};
    if (pav_sam_fetch(sam, &cols) != PAV_OK) { printf("error %s\n", pav_last_error(nullptr)); pav_sam_close(sam); return 0; }
    pav_sam_header(sam, head.data());
    long long spos = 0, send = 0, sref = 0;
    for (size_t i = 0; i < n; ++i) { spos += pos[i]; send += end[i]; sref += refbp[i]; }
    uLong ncrc = crc32(0L, Z_NULL, 0);
    for (uint32_t i = 0; i < info.n_ref; ++i) { const char *s = pav_sam_name(sam, 0, i); ncrc = crc32(ncrc, (const Bytef *)s, (uInt)strlen(s) + 1); }
    for (uint32_t i = 0; i < info.n_qry; ++i) { const char *s = pav_sam_name(sam, 1, i); ncrc = crc32(ncrc, (const Bytef *)s, (uInt)strlen(s) + 1); }
    printf("ok records=%llu rows=%llu refs=%u qrys=%u pos=%lld end=%lld ref_bp=%lld cigar_bytes=%llu header_bytes=%llu names_crc=%08lx\n",
           (unsigned long long)info.n_records, (unsigned long long)info.n_rows, info.n_ref, info.n_qry, spos, send, sref,
           (unsigned long long)info.cigar_bytes, (unsigned long long)info.header_bytes, ncrc);
    pav_sam_close(sam);
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: hostio_check fasta|bed|sam <path>...\n"); return 2; }
    for (int i = 2; i < argc; ++i) {
        if (!strcmp(argv[1], "fasta")) check_fasta(argv[i]);
        else if (!strcmp(argv[1], "bed")) check_bed(argv[i]);
        else if (!strcmp(argv[1], "sam")) check_sam(argv[i]);
        else return 2;
        fflush(stdout);
    }
    return 0;
}
