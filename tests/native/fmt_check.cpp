// Host build of pav_amd/csrc/fmt_dev.h (the device table writer's text primitives) against the host writer's formatter
// (textio.h put_f64_repr = std::to_chars shortest round trip) - test infrastructure, run by tests/test_host_fmt.py.
//   fmt_check sweep <n> <seed>       n random doubles of every kind (uniform bit patterns, values near powers of ten, KERN-like
//                                    magnitudes, subnormals, integers); exits 1 on the first difference
//   fmt_check repr                   reads doubles as hex bit patterns from stdin, prints the repr of each
//   fmt_check ints                   integer formatting against snprintf
#include <charconv>
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>

#include "../../pav_amd/csrc/fmt_dev.h"

static std::string host_repr(double v) {                  // textio.h put_f64_repr, restated on std::to_chars
    std::string s;
    if (std::isnan(v)) return s;
    if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
    char buf[40];
    const auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::scientific);
    const char *p = buf, *end = r.ptr;
    if (*p == '-') { s.push_back('-'); ++p; }
    char digits[24]; int nd = 0;
    while (p < end && *p != 'e') { if (*p != '.') digits[nd++] = *p; ++p; }
    int exp10 = 0; bool neg = false;
    if (p < end) { ++p; if (*p == '-') { neg = true; ++p; } else if (*p == '+') ++p; for (; p < end; ++p) exp10 = exp10 * 10 + (*p - '0'); }
    if (neg) exp10 = -exp10;
    const int decpt = exp10 + 1;
    if (decpt > -4 && decpt <= 16) {
        if (decpt <= 0) { s += "0."; s.append((size_t)(-decpt), '0'); s.append(digits, (size_t)nd); }
        else if (decpt >= nd) { s.append(digits, (size_t)nd); s.append((size_t)(decpt - nd), '0'); s += ".0"; }
        else { s.append(digits, (size_t)decpt); s.push_back('.'); s.append(digits + decpt, (size_t)(nd - decpt)); }
    } else {
        s.push_back(digits[0]);
        if (nd > 1) { s.push_back('.'); s.append(digits + 1, (size_t)(nd - 1)); }
        s.push_back('e');
        int e = decpt - 1;
        s.push_back(e < 0 ? '-' : '+');
        if (e < 0) e = -e;
        char eb[16]; snprintf(eb, sizeof eb, "%02d", e); s += eb;
    }
    return s;
}

static std::string dev_repr(double v) {
    uint8_t out[40];
    memset(out, '#', sizeof out);
    const uint32_t n = pav::fmt::put_f64_repr(out, v);
    if (n > 24 || out[n] != '#' || pav::fmt::f64_repr_len(v) != n) { printf("length: put %u, len %u\n", n, pav::fmt::f64_repr_len(v)); exit(1); }
    return std::string(reinterpret_cast<char *>(out), n);
}

static double from_bits(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const std::string mode = argv[1];
    if (mode == "repr") {
        char line[64];
        while (fgets(line, sizeof line, stdin)) {
            const uint64_t b = strtoull(line, nullptr, 16);
            printf("%s\n", dev_repr(from_bits(b)).c_str());
        }
        return 0;
    }
    if (mode == "ints") {
        std::mt19937_64 rng(7);
        uint8_t out[32]; char ref[32];
        for (int i = 0; i < 2000000; ++i) {
            uint64_t v = rng() >> (rng() % 64);
            if (i < 64) v = i < 20 ? (uint64_t)i : ~0ull >> (i - 20);
            uint32_t n = pav::fmt::put_u64(out, v);
            int m = snprintf(ref, sizeof ref, "%" PRIu64, v);
            if ((int)n != m || memcmp(out, ref, n) || pav::fmt::dec_len(v) != n) { printf("u64 %" PRIu64 " differs\n", v); return 1; }
            const int64_t s = (int64_t)v;
            n = pav::fmt::put_i64(out, s);
            m = snprintf(ref, sizeof ref, "%" PRId64, s);
            if ((int)n != m || memcmp(out, ref, n) || pav::fmt::i64_len(s) != n) { printf("i64 %" PRId64 " differs\n", s); return 1; }
        }
        for (uint64_t p = 1, k = 0; k < 20; ++k, p *= 10) {                  // powers of ten and their neighbours
            for (uint64_t v : {p - 1, p, p + 1}) {
                const uint32_t n = pav::fmt::put_u64(out, v);
                const int m = snprintf(ref, sizeof ref, "%" PRIu64, v);
                if ((int)n != m || memcmp(out, ref, n)) { printf("u64 %" PRIu64 " differs\n", v); return 1; }
            }
            if (k == 19) break;
        }
        printf("ok ints\n");
        return 0;
    }
    if (mode == "sweep") {
        const uint64_t n = strtoull(argv[2], nullptr, 10);
        std::mt19937_64 rng(strtoull(argv[3], nullptr, 10));
        std::uniform_real_distribution<double> uni(0.0, 1.0);
        uint64_t checked = 0;
        auto check = [&](double v) {
            const std::string a = dev_repr(v), b = host_repr(v);
            ++checked;
            if (a != b) { uint64_t bits; memcpy(&bits, &v, 8); printf("differs: bits %016" PRIx64 " device-side '%s' host '%s'\n", bits, a.c_str(), b.c_str()); exit(1); }
        };
        for (double v : {0.0, -0.0, 1.0, -1.0, 0.1, 0.5, 1e16, 1e17, 9999999999999998.0, 1e-4, 1e-5, 0.0001, 0.00001234, 123456789012345680.0,
                         5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, 4.9406564584124654e-324, 1e22, 1e23, 9007199254740993.0,
                         (double)INFINITY, -(double)INFINITY, (double)NAN, 0.3, 2.0 / 3.0, 1e-300, 123.456, 1e15, 1e-7, 299792458.0})
            check(v);
        for (int e = -324; e <= 308; ++e)                                    // powers of ten and the doubles beside them
            for (int d = 1; d <= 9; ++d) {
                char t[32]; snprintf(t, sizeof t, "%de%d", d, e);
                const double v = strtod(t, nullptr);
                check(v); check(std::nextafter(v, 0.0)); check(std::nextafter(v, INFINITY)); check(-v);
            }
        for (uint32_t ex = 0; ex < 2047; ++ex)                               // every binade: smallest, largest, power of two
            for (uint64_t m : {0ull, 1ull, (1ull << 52) - 1, 1ull << 51, 0x5555555555555ull}) check(from_bits((uint64_t)ex << 52 | m));
        for (uint64_t i = 0; i < n; ++i) {
            switch (i & 7) {
                case 0: case 1: { const double v = from_bits(rng()); check(v); break; }                   // any bit pattern
                case 2: check(uni(rng)); break;                                                            // [0, 1)
                case 3: check(std::exp(-uni(rng) * 700.0)); break;                                         // KERN-like tails
                case 4: check(std::ldexp(uni(rng), (int)(rng() % 2100) - 1074)); break;                    // any magnitude
                case 5: check((double)(rng() >> (rng() % 64))); break;                                     // integers
                case 6: check(from_bits(rng() & ((1ull << 52) - 1))); break;                               // subnormals
                case 7: { char t[40]; snprintf(t, sizeof t, "%" PRIu64 "e%d", (uint64_t)(rng() % 100000000ull), (int)(rng() % 600) - 300); check(strtod(t, nullptr)); break; }   // short decimals
            }
        }
        printf("ok %" PRIu64 "\n", checked);
        return 0;
    }
    return 2;
}
