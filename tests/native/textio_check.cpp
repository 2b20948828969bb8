// Host-only check of the text / gzip helpers of the table writers (pav_amd/csrc/textio.h: what pav_cigar_write_tables and
// pav_inv_write_tables format their rows with).  tests/test_host_sanitize.py builds it with ASan + UBSan and with TSan (the writer
// formats and deflates chunks on worker threads) and compares its files with what Python writes for the same values.
//   textio_check <values.bin> <out.tsv> <out.tsv.gz> <threads>
// values.bin: uint64 n, then n x { int64, float64, uint32 text length, text bytes }.
#include "../../pav_amd/csrc/textio.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>

// common.h declares the context; this driver never makes one (ctx == nullptr) and no HIP call is reached
namespace pav {
thread_local std::string g_err;
int fail(pav_ctx *ctx, int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    (void)ctx;
    g_err = buf;
    return code;
}
}  // namespace pav

int main(int argc, char **argv) {
    if (argc < 5) return 2;
    FILE *fh = fopen(argv[1], "rb");
    if (!fh) return 2;
    uint64_t n = 0;
    if (fread(&n, 8, 1, fh) != 1) return 2;
    std::vector<int64_t> iv(n); std::vector<double> fv(n); std::vector<std::string> tv(n);
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t len = 0;
        if (fread(&iv[i], 8, 1, fh) != 1 || fread(&fv[i], 8, 1, fh) != 1 || fread(&len, 4, 1, fh) != 1) return 2;
        tv[i].resize(len);
        if (len && fread(&tv[i][0], 1, len, fh) != len) return 2;
    }
    fclose(fh);
    const int threads = atoi(argv[4]);
    auto row = [&](uint64_t i, std::string &s) {
        pav::put_u64(s, i); s.push_back('\t');
        pav::put_i64(s, iv[i]); s.push_back('\t');
        pav::put_f64_repr(s, fv[i]); s.push_back('\t');
        s += pav::csv_field(tv[i]); s.push_back('\n');
    };
    const std::string header = "ROW\tINT\tFLOAT\tTEXT\n";
    if (pav::write_table(nullptr, argv[2], header, n, threads, 6, row) != PAV_OK) { fprintf(stderr, "%s\n", pav::g_err.c_str()); return 1; }
    if (pav::write_table(nullptr, argv[3], header, n, threads, 1, row) != PAV_OK) { fprintf(stderr, "%s\n", pav::g_err.c_str()); return 1; }
    if (pav::write_table(nullptr, "/nonexistent-dir/x.tsv", header, n, threads, 1, row) == PAV_OK) return 1;     // refused with a message
    printf("ok %llu rows\n", (unsigned long long)n);
    return 0;
}
