// Text primitives of the DEVICE table writer (textdev.hip): decimal integers and repr(float) - the bytes pandas.DataFrame.to_csv
// writes for an int64 / float64 column (rules/call.snakefile:845-846, rules/call_inv.snakefile:287-291) - as functions a lane calls
// for its own row.  The float formatter is the Ryu algorithm (Adams, "Ryu: fast float-to-string conversion", PLDI 2018): the
// shortest digit string that reads back as the same double, the closest one among those, laid out as CPython's float_repr_style
// 'short' lays it out (positional for -4 < decimal point <= 16, else d.ddde+XX).  Its two 128-bit tables are generated
// (tools/gen/gen_ryu_tables.py -> ryu_tables.h).
//
// Every function is __host__ __device__: the host build is what tests/test_host_fmt.py runs against std::to_chars and Python's
// repr on millions of values (test infrastructure; the product calls these from kernels only).
#pragma once

#include <cstdint>

#include "ryu_tables.h"

#if defined(__HIPCC__)
#define PAV_HD __host__ __device__ __forceinline__
#else
#define PAV_HD inline
#endif

namespace pav {
namespace fmt {

#if defined(__HIPCC__)
__device__ const uint64_t d_ryu_inv[PAV_RYU_N_INV][2] = {PAV_RYU_INV_ROWS};
__device__ const uint64_t d_ryu_pow[PAV_RYU_N_POW][2] = {PAV_RYU_POW_ROWS};
#endif
static const uint64_t h_ryu_inv[PAV_RYU_N_INV][2] = {PAV_RYU_INV_ROWS};
static const uint64_t h_ryu_pow[PAV_RYU_N_POW][2] = {PAV_RYU_POW_ROWS};

PAV_HD const uint64_t *ryu_inv(uint32_t i) {
#if defined(__HIP_DEVICE_COMPILE__)
    return d_ryu_inv[i];
#else
    return h_ryu_inv[i];
#endif
}
PAV_HD const uint64_t *ryu_pow(uint32_t i) {
#if defined(__HIP_DEVICE_COMPILE__)
    return d_ryu_pow[i];
#else
    return h_ryu_pow[i];
#endif
}

// ---- integers -----------------------------------------------------------------------------------------------------
PAV_HD uint32_t dec_len(uint64_t v) {                 // digits of v (1 for 0)
    uint32_t n = 1;
    if (v >= 10000000000000000ull) { v /= 10000000000000000ull; n += 16; }
    if (v >= 100000000ull) { v /= 100000000ull; n += 8; }
    if (v >= 10000ull) { v /= 10000ull; n += 4; }
    if (v >= 100ull) { v /= 100ull; n += 2; }
    if (v >= 10ull) n += 1;
    return n;
}
// writes the decimal digits of v at out, returns their number
PAV_HD uint32_t put_u64(uint8_t *out, uint64_t v) {
    const uint32_t n = dec_len(v);
    for (uint32_t i = n; i-- > 0;) { out[i] = (uint8_t)('0' + v % 10); v /= 10; }
    return n;
}
PAV_HD uint32_t i64_len(int64_t v) { return v < 0 ? 1 + dec_len(0 - (uint64_t)v) : dec_len((uint64_t)v); }
PAV_HD uint32_t put_i64(uint8_t *out, int64_t v) {
    if (v < 0) { out[0] = '-'; return 1 + put_u64(out + 1, 0 - (uint64_t)v); }
    return put_u64(out, (uint64_t)v);
}

// ---- Ryu: the shortest decimal (digits, exponent) of a double ---------------------------------------------------------------
PAV_HD uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}
// ((m * mul) >> j) for the 128-bit mul = {lo, hi}, 64 < j < 128 + 64, m < 2^55: the paper's mulShift
PAV_HD uint64_t mul_shift(uint64_t m, const uint64_t *mul, int32_t j) {
    const uint64_t b0_hi = mulhi64(m, mul[0]);
    const uint64_t b2_lo = m * mul[1], b2_hi = mulhi64(m, mul[1]);
    const uint64_t s_lo = b0_hi + b2_lo;
    const uint64_t s_hi = b2_hi + (s_lo < b0_hi ? 1u : 0u);
    const int32_t sh = j - 64;                       // 0 < sh < 64
    return (s_hi << (64 - sh)) | (s_lo >> sh);
}
PAV_HD uint32_t pow5_factor(uint64_t v) { uint32_t c = 0; for (;;) { const uint64_t q = v / 5; if (v - 5 * q != 0) break; v = q; ++c; } return c; }
PAV_HD bool multiple_of_pow5(uint64_t v, uint32_t p) { return pow5_factor(v) >= p; }
PAV_HD bool multiple_of_pow2(uint64_t v, uint32_t p) { return (v & ((1ull << p) - 1)) == 0; }
PAV_HD int32_t pow5bits(int32_t e) { return (int32_t)(((uint32_t)e * 1217359u) >> 19) + 1; }   // bit length of 5^e, 0 <= e <= 3528
PAV_HD int32_t log10_pow2(int32_t e) { return (int32_t)(((uint32_t)e * 78913u) >> 18); }       // floor(log10(2^e)), 0 <= e <= 1650
PAV_HD int32_t log10_pow5(int32_t e) { return (int32_t)(((uint32_t)e * 732923u) >> 20); }      // floor(log10(5^e)), 0 <= e <= 2620

struct Dec { uint64_t digits; int32_t exp; };       // value = digits * 10^exp, digits without trailing zeros only by chance

// finite, non-zero |v| given by its IEEE fields
PAV_HD Dec shortest(uint64_t ieee_mant, uint32_t ieee_exp) {
    int32_t e2; uint64_t m2;
    if (ieee_exp == 0) { e2 = 1 - 1023 - 52 - 2; m2 = ieee_mant; }
    else { e2 = (int32_t)ieee_exp - 1023 - 52 - 2; m2 = (1ull << 52) | ieee_mant; }
    const bool accept = (m2 & 1) == 0;               // round-to-even reads the interval's ends back as v
    const uint64_t mv = 4 * m2;
    const uint32_t mm_shift = (ieee_mant != 0 || ieee_exp <= 1) ? 1u : 0u;
    uint64_t vr, vp, vm; int32_t e10;
    bool vm_tz = false, vr_tz = false;
    if (e2 >= 0) {
        const uint32_t q = (uint32_t)(log10_pow2(e2) - (e2 > 3));
        e10 = (int32_t)q;
        const int32_t k = PAV_RYU_BITS + pow5bits((int32_t)q) - 1;
        const int32_t i = -e2 + (int32_t)q + k;
        const uint64_t *mul = ryu_inv(q);
        vr = mul_shift(4 * m2, mul, i); vp = mul_shift(4 * m2 + 2, mul, i); vm = mul_shift(4 * m2 - 1 - mm_shift, mul, i);
        if (q <= 21) {                                // only then can mv, mp or mm be a multiple of 5^q
            const uint32_t mv_mod5 = (uint32_t)(mv - 5 * (mv / 5));
            if (mv_mod5 == 0) vr_tz = multiple_of_pow5(mv, q);
            else if (accept) vm_tz = multiple_of_pow5(mv - 1 - mm_shift, q);
            else vp -= multiple_of_pow5(mv + 2, q) ? 1u : 0u;
        }
    } else {
        const uint32_t q = (uint32_t)(log10_pow5(-e2) - (-e2 > 1));
        e10 = (int32_t)q + e2;
        const int32_t i = -e2 - (int32_t)q;
        const int32_t k = pow5bits(i) - PAV_RYU_BITS;
        const int32_t j = (int32_t)q - k;
        const uint64_t *mul = ryu_pow((uint32_t)i);
        vr = mul_shift(4 * m2, mul, j); vp = mul_shift(4 * m2 + 2, mul, j); vm = mul_shift(4 * m2 - 1 - mm_shift, mul, j);
        if (q <= 1) {
            vr_tz = true;                              // mv = 4 m2 always has two trailing zero bits
            if (accept) vm_tz = mm_shift == 1;
            else --vp;
        } else if (q < 63) {
            vr_tz = multiple_of_pow2(mv, q);
        }
    }
    int32_t removed = 0; uint32_t last = 0; uint64_t out;
    if (vm_tz || vr_tz) {                             // rare: exact ties and interval ends need the removed digits' history
        for (;;) {
            const uint64_t vp10 = vp / 10, vm10 = vm / 10;
            if (vp10 <= vm10) break;
            const uint32_t vm_mod = (uint32_t)(vm - 10 * vm10);
            const uint64_t vr10 = vr / 10; const uint32_t vr_mod = (uint32_t)(vr - 10 * vr10);
            vm_tz &= vm_mod == 0; vr_tz &= last == 0; last = vr_mod;
            vr = vr10; vp = vp10; vm = vm10; ++removed;
        }
        if (vm_tz) {
            for (;;) {
                const uint64_t vm10 = vm / 10; const uint32_t vm_mod = (uint32_t)(vm - 10 * vm10);
                if (vm_mod != 0) break;
                const uint64_t vp10 = vp / 10, vr10 = vr / 10; const uint32_t vr_mod = (uint32_t)(vr - 10 * vr10);
                vr_tz &= last == 0; last = vr_mod;
                vr = vr10; vp = vp10; vm = vm10; ++removed;
            }
        }
        if (vr_tz && last == 5 && vr % 2 == 0) last = 4;      // exactly half: to even
        out = vr + (((vr == vm && (!accept || !vm_tz)) || last >= 5) ? 1u : 0u);
    } else {
        bool up = false;
        const uint64_t vp100 = vp / 100, vm100 = vm / 100;
        if (vp100 > vm100) {
            const uint64_t vr100 = vr / 100; const uint32_t vr_mod = (uint32_t)(vr - 100 * vr100);
            up = vr_mod >= 50; vr = vr100; vp = vp100; vm = vm100; removed += 2;
        }
        for (;;) {
            const uint64_t vp10 = vp / 10, vm10 = vm / 10;
            if (vp10 <= vm10) break;
            const uint64_t vr10 = vr / 10; const uint32_t vr_mod = (uint32_t)(vr - 10 * vr10);
            up = vr_mod >= 5; vr = vr10; vp = vp10; vm = vm10; ++removed;
        }
        out = vr + ((vr == vm || up) ? 1u : 0u);
    }
    return Dec{out, e10 + removed};
}

// ---- repr(float) ------------------------------------------------------------------------------------------------------
// What textio.h put_f64_repr writes (pandas' to_csv of a float64 column: repr; NaN -> the empty na_rep).  Two functions with one
// layout rule: the length, and the bytes (at most 24) - every digit is stored straight at its place, no buffer in between (a
// kernel that indexed a private array would spill it to scratch memory).
struct ReprShape { uint64_t digits; int32_t nd, decpt; uint32_t kind; };   // kind 0 NaN, 1 inf, 2 zero, 3 positional, 4 exponent
PAV_HD ReprShape repr_shape(double v, bool &neg) {
    union { double d; uint64_t u; } cv; cv.d = v;
    const uint64_t bits = cv.u;
    neg = (bits >> 63) != 0;
    const uint64_t mant = bits & ((1ull << 52) - 1);
    const uint32_t ex = (uint32_t)((bits >> 52) & 0x7FFu);
    if (ex == 0x7FFu) return ReprShape{0, 0, 0, mant ? 0u : 1u};
    if (ex == 0 && mant == 0) return ReprShape{0, 1, 1, 2u};
    const Dec d = shortest(mant, ex);
    const int32_t nd = (int32_t)dec_len(d.digits), decpt = d.exp + nd;
    return ReprShape{d.digits, nd, decpt, (decpt > -4 && decpt <= 16) ? 3u : 4u};
}
PAV_HD uint32_t f64_repr_len(double v) {
    bool neg; const ReprShape r = repr_shape(v, neg);
    const uint32_t sign = neg ? 1u : 0u;
    if (r.kind == 0) return 0;
    if (r.kind <= 2) return sign + 3;                                    // inf, 0.0
    if (r.kind == 3) {
        if (r.decpt <= 0) return sign + 2u + (uint32_t)(-r.decpt) + (uint32_t)r.nd;
        if (r.decpt >= r.nd) return sign + (uint32_t)r.decpt + 2u;
        return sign + (uint32_t)r.nd + 1u;
    }
    int32_t e = r.decpt - 1; if (e < 0) e = -e;
    return sign + (uint32_t)r.nd + (r.nd > 1 ? 1u : 0u) + 2u + (e < 10 ? 2u : dec_len((uint64_t)e));
}
PAV_HD uint32_t put_f64_repr(uint8_t *out, double v) {
    bool neg; const ReprShape r = repr_shape(v, neg);
    uint32_t n = 0;
    if (r.kind == 0) return 0;                                           // NaN: empty field
    if (neg) out[n++] = '-';
    if (r.kind == 1) { out[n++] = 'i'; out[n++] = 'n'; out[n++] = 'f'; return n; }
    if (r.kind == 2) { out[n++] = '0'; out[n++] = '.'; out[n++] = '0'; return n; }
    uint64_t dv = r.digits;
    const int32_t nd = r.nd, decpt = r.decpt;
    if (r.kind == 3) {
        if (decpt <= 0) {                                                // 0.000ddd
            out[n] = '0'; out[n + 1] = '.';
            for (int32_t i = 0; i < -decpt; ++i) out[n + 2 + i] = '0';
            const uint32_t at = n + 2u + (uint32_t)(-decpt);
            for (int32_t i = nd; i-- > 0;) { out[at + i] = (uint8_t)('0' + dv % 10); dv /= 10; }
            return at + (uint32_t)nd;
        }
        if (decpt >= nd) {                                               // ddd000.0
            for (int32_t i = nd; i-- > 0;) { out[n + i] = (uint8_t)('0' + dv % 10); dv /= 10; }
            for (int32_t i = nd; i < decpt; ++i) out[n + i] = '0';
            out[n + decpt] = '.'; out[n + decpt + 1] = '0';
            return n + (uint32_t)decpt + 2u;
        }
        for (int32_t i = nd; i-- > 0;) { out[n + i + (i >= decpt ? 1 : 0)] = (uint8_t)('0' + dv % 10); dv /= 10; }   // dd.ddd
        out[n + decpt] = '.';
        return n + (uint32_t)nd + 1u;
    }
    for (int32_t i = nd; i-- > 1;) { out[n + 1 + i] = (uint8_t)('0' + dv % 10); dv /= 10; }          // d.ddde+XX
    out[n] = (uint8_t)('0' + dv);
    if (nd > 1) { out[n + 1] = '.'; n += (uint32_t)nd + 1u; } else n += 1u;
    out[n++] = 'e';
    int32_t e = decpt - 1;
    out[n++] = e < 0 ? '-' : '+';
    if (e < 0) e = -e;
    if (e < 10) out[n++] = '0';
    n += put_u64(out + n, (uint64_t)e);
    return n;
}

}  // namespace fmt
}  // namespace pav
