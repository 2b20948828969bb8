/*
 * pav_oracle_density.c - scalar CPU restatement of PAV's k-mer state + density scan.
 * TEST INFRASTRUCTURE ONLY (see pav_oracle.h).  Follows:
 *   pavlib/seq.py:305-325                 ref_kmers (k-mer Counter of the reference region)
 *   scripts/density.py:500-545            main: low-complexity gate, reference set, orientation, contig stream
 *   scripts/density.py:154-342            get_smoothed_density (STATE_MER, compaction, KDE, interpolation, arg-max)
 *   scipy/stats/_kde.py + _stats.pyx      gaussian_kde: bandwidth, data*(1/h) scaling, loop order, libm exp
 *   numpy arr_interp                      slope*(x - x0) + y0
 *   pavlib/density.py:330-361             rl_encoder
 *   pavlib/inv.py:457-561                 annotate_inv_dup_mers (FLANK / MATCH)
 * K-mer integers use the encoding assumed for the absent kanapy (A0 C1 G2 T3, first base most significant).
 */
#include "pav_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- k-mers ------------------------------------------------------------------------------------------------ */
static inline int base2(uint8_t c) {
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2;
                 case 'T': case 't': return 3; default: return -1; }
}
static inline uint64_t kmask(int k) { return k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1ull); }

uint64_t orc_kmer_rc(uint64_t kmer, int k) {
    uint64_t rc = 0;
    for (int i = 0; i < k; ++i) { rc = (rc << 2) | ((kmer & 3ull) ^ 3ull); kmer >>= 2; }
    return rc;
}
uint64_t orc_kmer_canonical(uint64_t kmer, int k) { uint64_t rc = orc_kmer_rc(kmer, k); return kmer <= rc ? kmer : rc; }

/* kanapy.util.kmer.stream: every k-mer without a non-ACGT base, with the offset of its first base */
static uint64_t kmer_stream(const uint8_t *seq, uint64_t len, int k, uint64_t *kmers, int64_t *index) {
    uint64_t n = 0, kmer = 0, mask = kmask(k);
    int load = 0;
    for (uint64_t i = 0; i < len; ++i) {
        int c = base2(seq[i]);
        if (c < 0) { kmer = 0; load = 0; continue; }
        kmer = ((kmer << 2) | (uint64_t)c) & mask;
        if (++load >= k) { kmers[n] = kmer; if (index) index[n] = (int64_t)(i - (uint64_t)k + 1); ++n; }
    }
    return n;
}

/* open-addressing table: key -> (count, first position of insertion) */
typedef struct { uint64_t *key; uint32_t *cnt; uint64_t *first; uint8_t *used; uint64_t cap; } kset;
static uint64_t mix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
static void kset_init(kset *s, uint64_t n) {
    uint64_t cap = 64; while (cap < 2 * n + 2) cap <<= 1;
    s->cap = cap; s->key = calloc(cap, 8); s->cnt = calloc(cap, 4); s->first = calloc(cap, 8); s->used = calloc(cap, 1);
}
static void kset_free(kset *s) { free(s->key); free(s->cnt); free(s->first); free(s->used); }
static uint64_t kset_slot(const kset *s, uint64_t k) {
    uint64_t h = mix64(k) & (s->cap - 1);
    while (s->used[h] && s->key[h] != k) h = (h + 1) & (s->cap - 1);
    return h;
}
static void kset_add(kset *s, uint64_t k, uint64_t pos) {
    uint64_t h = kset_slot(s, k);
    if (!s->used[h]) { s->used[h] = 1; s->key[h] = k; s->first[h] = pos; }
    s->cnt[h]++;
}
static int kset_has(const kset *s, uint64_t k) { return s->used[kset_slot(s, k)]; }

/* ---- density ----------------------------------------------------------------------------------------------- */
struct orc_density {
    orc_density_info info;
    int64_t *index; int8_t *state_mer; int8_t *state; double *kern[3]; uint64_t *kmer; uint8_t *interp;
};
const orc_density_info *orc_density_get_info(const orc_density *d) { return &d->info; }
const int64_t *orc_density_index(const orc_density *d) { return d->index; }
const int8_t *orc_density_state_mer(const orc_density *d) { return d->state_mer; }
const int8_t *orc_density_state(const orc_density *d) { return d->state; }
const double *orc_density_kern(const orc_density *d, int s) { return d->kern[s]; }
const uint64_t *orc_density_kmer(const orc_density *d) { return d->kmer; }
const uint8_t *orc_density_interp(const orc_density *d) { return d->interp; }
void orc_density_free(orc_density *d) {
    if (!d) return;
    free(d->index); free(d->state_mer); free(d->state); free(d->kmer); free(d->interp);
    for (int s = 0; s < 3; ++s) free(d->kern[s]);
    free(d);
}

typedef struct { uint64_t n; const double *p_scaled; double inv_h, norm, w, count; } kde1;

/* Evaluation points are independent of each other, so large regions (tests of 0.1 - 1.2 Mbp) may spread them over threads;
 * the sum of ONE point is always accumulated by one thread in scipy's order, so the table does not depend on the count. */
static int g_threads = 1;
void orc_density_set_threads(int n) { g_threads = n < 1 ? 1 : n; }

/* scipy gaussian_kernel_estimate for d = 1: for one evaluation point, data ascending (the loop order in which
 * scipy accumulates estimate[j]).  Data points further than 39 bandwidths from the point are stepped over: their terms are
 * exp(-760.5) or less, which is exactly 0.0 in IEEE double (the smallest subnormal is exp(-744.4)), and est + 0.0 == est
 * bit for bit - the scaled positions ascend, so the points that matter are one range found by bisection. */
static uint64_t first_ge(const double *p, uint64_t n, double v) {
    uint64_t lo = 0, hi = n;
    while (lo < hi) { uint64_t mid = lo + (hi - lo) / 2; if (p[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}
static double kde_eval(const kde1 *kd, double x) {
    if (kd->n == 0) return 0.0;                                /* density.py:84,92,100: zeros for an absent state */
    const double xs = x * kd->inv_h;
    double est = 0.0;
    const uint64_t i0 = first_ge(kd->p_scaled, kd->n, xs - 39.0), i1 = first_ge(kd->p_scaled, kd->n, xs + 39.0);
    for (uint64_t i = i0; i < i1; ++i) {
        const double r = kd->p_scaled[i] - xs;
        const double arg = exp(-(r * r) / 2) * kd->norm;
        est += kd->w * arg;
    }
    return est * kd->count;                                    /* density.py:110-115: kernel(val) * sum_state */
}

static int argmax3(double a, double b, double c) { int m = 0; double v = a; if (b > v) { v = b; m = 1; } if (c > v) m = 2; return m; }

orc_density *orc_density_run(const uint8_t *ref_seq, uint64_t ref_len, const uint8_t *tig_seq, uint64_t tig_len,
                             int ref_rc, const orc_den_params *pp) {
    orc_density *d = calloc(1, sizeof *d);
    const int k = pp->k;

    /* reference k-mer counts (seq.py:305-325) */
    uint64_t *rk = malloc(8 * (ref_len + 1));
    uint64_t n_ref = kmer_stream(ref_seq, ref_len, k, rk, NULL);
    if (n_ref == 0) { d->info.status = 125; d->info.fail_kind = 1; free(rk); return d; }     /* density.py:510-513 */
    kset cnt; kset_init(&cnt, n_ref);
    for (uint64_t i = 0; i < n_ref; ++i) kset_add(&cnt, rk[i], i);
    uint32_t max_count = 0;
    for (uint64_t h = 0; h < cnt.cap; ++h) if (cnt.used[h] && cnt.cnt[h] > max_count) max_count = cnt.cnt[h];
    d->info.max_count = max_count;
    if (max_count > pp->max_ref_kmer_count) {                                                /* density.py:516-527 */
        uint64_t best = ~0ull, bk = 0;                      /* first k-mer (insertion order) with the max count */
        for (uint64_t h = 0; h < cnt.cap; ++h)
            if (cnt.used[h] && cnt.cnt[h] == max_count && cnt.first[h] < best) { best = cnt.first[h]; bk = cnt.key[h]; }
        d->info.status = 125; d->info.fail_kind = 2; d->info.max_kmer = bk;
        kset_free(&cnt); free(rk); return d;
    }
    kset ref_set; kset_init(&ref_set, n_ref);                                                /* density.py:536-539 */
    for (uint64_t h = 0; h < cnt.cap; ++h)
        if (cnt.used[h]) kset_add(&ref_set, ref_rc ? orc_kmer_rc(cnt.key[h], k) : cnt.key[h], 0);
    kset_free(&cnt); free(rk);

    /* contig k-mer stream with index (density.py:543-545) and STATE_MER (:165-175) */
    uint64_t *tk = malloc(8 * (tig_len + 1)); int64_t *ti = malloc(8 * (tig_len + 1));
    uint64_t n_tig = kmer_stream(tig_seq, tig_len, k, tk, ti);
    int8_t *sm = malloc(n_tig + 1);
    uint64_t state_count[3] = {0, 0, 0};
    static const int8_t M[2][2] = {{-1, 2}, {0, 1}};                                         /* density.py:38-43 */
    for (uint64_t i = 0; i < n_tig; ++i) {
        sm[i] = M[kset_has(&ref_set, tk[i])][kset_has(&ref_set, orc_kmer_rc(tk[i], k))];
        if (sm[i] >= 0) state_count[sm[i]]++;
    }
    kset_free(&ref_set);
    /* informative sites; low-count states removed (density.py:178-190) */
    int keep[3];
    for (int s = 0; s < 3; ++s) keep[s] = state_count[s] >= pp->min_state_count;
    uint64_t n = 0;
    for (uint64_t i = 0; i < n_tig; ++i) if (sm[i] >= 0 && keep[sm[i]]) ++n;
    d->info.n = (uint32_t)n;
    d->index = malloc(8 * (n + 1)); d->state_mer = malloc(n + 1); d->state = malloc(n + 1); d->kmer = malloc(8 * (n + 1));
    d->interp = calloc(n + 1, 1);
    for (int s = 0; s < 3; ++s) d->kern[s] = calloc(n + 1, 8);
    uint64_t j = 0;
    for (uint64_t i = 0; i < n_tig; ++i)
        if (sm[i] >= 0 && keep[sm[i]]) { d->index[j] = ti[i]; d->state_mer[j] = sm[i]; d->kmer[j] = tk[i]; d->state[j] = -1; ++j; }
    free(tk); free(ti); free(sm);
    if (n < pp->min_informative || n == 0) { d->info.status = 1; return d; }                           /* density.py:193-195 */

    /* bandwidth and per-state KDE set-up (density.py:198, 69-102; scipy _kde.py:_compute_covariance) */
    const double bandwidth = pow((double)n, -1.0 / 5.0) * pp->den_smooth;
    kde1 kd[3]; double *scaled[3];
    for (int s = 0; s < 3; ++s) {
        uint64_t m = 0; unsigned __int128 s1 = 0, s2 = 0;
        for (uint64_t i = 0; i < n; ++i) if (d->state_mer[i] == s) { ++m; s1 += i; s2 += (unsigned __int128)i * i; }
        d->info.state_count[s] = (uint32_t)m;
        scaled[s] = malloc(8 * (m + 1));
        kd[s].n = m; kd[s].p_scaled = scaled[s]; kd[s].count = (double)m;
        if (m == 0) { kd[s].inv_h = kd[s].norm = kd[s].w = 0; d->info.h[s] = 0; continue; }
        /* unbiased variance of the integer positions, exact numerator: (m*S2 - S1^2) / (m*(m-1)) */
        const unsigned __int128 num = (unsigned __int128)m * s2 - s1 * s1;
        const double var = (double)num / ((double)m * (double)(m - 1));
        const double h = sqrt(var) * bandwidth;                 /* cho_cov = cholesky(cov) * factor */
        d->info.h[s] = h;
        kd[s].inv_h = 1.0 / h;                                  /* solve_triangular == multiply by the reciprocal */
        kd[s].norm = pow(2 * 3.14159265358979323846, -0.5) / h;
        kd[s].w = 1.0 / (double)m;                              /* weights = ones(n) / n */
        uint64_t q = 0;
        for (uint64_t i = 0; i < n; ++i) if (d->state_mer[i] == s) scaled[s][q++] = (double)i * kd[s].inv_h;
    }

    /* sampled sites (density.py:206-214), density + state there (:238-255) */
    const uint64_t srs = pp->state_run_smooth;
    uint64_t n_samp = (n + srs - 1) / srs;
    uint64_t *samp = malloc(8 * (n_samp + 2));
    for (uint64_t q = 0; q < n_samp; ++q) samp[q] = q * srs;
    if (n_samp == 0 || samp[n_samp - 1] != n - 1) samp[n_samp++] = n - 1;
    uint64_t n_eval = n_samp;
    #pragma omp parallel for schedule(dynamic, 16) num_threads(g_threads)
    for (uint64_t q = 0; q < n_samp; ++q) {
        const uint64_t x = samp[q];
        for (int s = 0; s < 3; ++s) d->kern[s][x] = kde_eval(&kd[s], (double)x);
        d->state[x] = (int8_t)argmax3(d->kern[0][x], d->kern[1][x], d->kern[2][x]);
    }
    /* windows between sampled sites (density.py:257-286): interpolate or compute.  A window reads the two sampled sites at
     * its ends and writes the rows between them: windows are independent. */
    const uint64_t n_win = n_samp - 1;                          /* n_samp >= 1 */
    #pragma omp parallel for schedule(dynamic, 4) reduction(+ : n_eval) num_threads(g_threads)
    for (uint64_t q = 0; q < n_win; ++q) {
        const uint64_t a = samp[q], b = samp[q + 1];
        if (b == a + 1) continue;
        int state_change = d->state[a] != d->state[b];
        for (uint64_t i = a + 1; i <= b && !state_change; ++i) if (d->state_mer[i] != d->state_mer[a]) state_change = 1;
        double dmax = 0;
        for (int s = 0; s < 3; ++s) { double df = fabs(d->kern[s][a] - d->kern[s][b]); if (df > dmax) dmax = df; }
        if (state_change || dmax > pp->state_run_delta) {
            for (uint64_t x = a + 1; x < b; ++x) {
                for (int s = 0; s < 3; ++s) d->kern[s][x] = kde_eval(&kd[s], (double)x);
                n_eval += 1;
            }
        } else {                                                /* np.interp: slope * (x - x0) + y0 (density.py:121-151) */
            for (int s = 0; s < 3; ++s) {
                const double slope = (d->kern[s][b] - d->kern[s][a]) / ((double)b - (double)a);
                for (uint64_t x = a + 1; x < b; ++x) d->kern[s][x] = slope * ((double)x - (double)a) + d->kern[s][a];
            }
            for (uint64_t x = a + 1; x < b; ++x) d->interp[x] = 1;
        }
    }
    d->info.n_eval = n_eval;
    /* spike penalty and arg-max (density.py:329-338) */
    for (uint64_t x = 0; x < n; ++x) {
        for (int s = 0; s < 3; ++s) if (d->kern[s][x] > 1.0) d->kern[s][x] = 1 / d->kern[s][x];
        d->state[x] = (int8_t)argmax3(d->kern[0][x], d->kern[1][x], d->kern[2][x]);
    }
    for (int s = 0; s < 3; ++s) free(scaled[s]);
    free(samp);
    d->info.status = 0;
    return d;
}

/* ---- pavlib/density.py:330-361 ----------------------------------------------------------------------------- */
uint32_t orc_rl_encode(const int8_t *state, const int64_t *index, uint32_t n, orc_run *runs, uint32_t cap) {
    uint32_t m = 0;
    int have = 0; int8_t cur = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (have && state[i] == cur) {                                      /* density.py:347-349 */
            if (m - 1 < cap) { runs[m - 1].count++; runs[m - 1].end = index[i]; }
        } else {                                                            /* :351-357 */
            if (m < cap) { runs[m].state = state[i]; runs[m].count = 1; runs[m].pos = runs[m].end = index[i]; }
            ++m; cur = state[i]; have = 1;
        }
    }
    return m;
}

/* ---- pavlib/inv.py:457-561 --------------------------------------------------------------------------------- */
void orc_annotate(const uint64_t *kmer, const int64_t *index, uint32_t n, int k, int64_t qry_index_base,
                  int64_t up_pos, int64_t up_end, int64_t dn_pos, int64_t dn_end,
                  const uint8_t *ref_up, uint64_t ref_up_len, const uint8_t *ref_dn, uint64_t ref_dn_len,
                  uint8_t *flank, uint8_t *match) {
    uint64_t *buf = malloc(8 * ((ref_up_len > ref_dn_len ? ref_up_len : ref_dn_len) + 1));
    kset up, dn;
    uint64_t m = kmer_stream(ref_up, ref_up_len, k, buf, NULL);            /* inv.py:507-509 canonical k-mers */
    kset_init(&up, m); for (uint64_t i = 0; i < m; ++i) kset_add(&up, orc_kmer_canonical(buf[i], k), 0);
    m = kmer_stream(ref_dn, ref_dn_len, k, buf, NULL);                     /* inv.py:511-513 */
    kset_init(&dn, m); for (uint64_t i = 0; i < m; ++i) kset_add(&dn, orc_kmer_canonical(buf[i], k), 0);
    static const uint8_t LOC[2][2] = {{3, 2}, {1, 3}};                     /* inv.py:46-51: NA, OTHER / SAME, NA */
    for (uint32_t i = 0; i < n; ++i) {
        const int64_t q = index[i] + qry_index_base;                       /* inv.py:519 */
        uint8_t f = 0;
        if (q >= up_pos && q < up_end - k) f = 1;                          /* inv.py:524-527 */
        if (q >= dn_pos && q < dn_end - k) f = 2;                          /* inv.py:529-532 */
        flank[i] = f;
        uint8_t mt = 0;
        if (f == 1) mt = LOC[kset_has(&up, kmer[i])][kset_has(&dn, kmer[i])];   /* raw KMER vs canonical sets (:537-544) */
        if (f == 2) mt = LOC[kset_has(&dn, kmer[i])][kset_has(&up, kmer[i])];   /* :546-553 */
        match[i] = mt;                                                      /* 0 '', 1 SAME, 2 OTHER, 3 NaN */
    }
    kset_free(&up); kset_free(&dn); free(buf);
}
