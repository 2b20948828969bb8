/*
 * Synthetic workload generator (host, plain C): random soft-masked reference sequence and contigs that are
 * mutated copies of reference segments with a base-exact =/X/I/D/H CIGAR, following the profile of
 * SURVEY.md section 8(d).  Workload tooling for tests/ and bench.py - not on the product's compute path and
 * not derived from the reference (PAV has no generator).
 *
 * Determinism: every stream is a xoshiro256** seeded by splitmix64(seed).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint64_t s[4]; } rng_t;

static uint64_t splitmix64(uint64_t *x) {
    uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static void rng_seed(rng_t *r, uint64_t seed) {
    for (int i = 0; i < 4; ++i) r->s[i] = splitmix64(&seed);
}
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static inline uint64_t rng_next(rng_t *r) {
    uint64_t *s = r->s, result = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return result;
}
static inline double rng_unit(rng_t *r) { return (double)(rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }
/* geometric gap >= 1 with success probability p */
static inline uint64_t rng_geom(rng_t *r, double p) {
    if (p <= 0.0) return UINT64_MAX / 4;
    if (p >= 1.0) return 1;
    double u = rng_unit(r);
    if (u <= 0.0) u = 1e-300;
    return (uint64_t)(log(u) / log1p(-p)) + 1;
}

static const char UP[4] = {'A', 'C', 'G', 'T'};
static const char LO[4] = {'a', 'c', 'g', 't'};

static inline int base_code(uint8_t c) {
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1;
                 case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return -1; }
}
static uint8_t COMP[256];
static int comp_init = 0;
static void init_comp(void) {
    if (comp_init) return;
    for (int i = 0; i < 256; ++i) COMP[i] = (uint8_t)i;
    const char *a = "ACGTRYSWKMBDHVNUacgtryswkmbdhvnu", *b = "TGCAYRSWMKVHDBNAtgcayrswmkvhdbna";
    for (int i = 0; a[i]; ++i) COMP[(uint8_t)a[i]] = (uint8_t)b[i];
    comp_init = 1;
}

/* Fill out[0..len) with i.i.d. uniform ACGT; case alternates in geometric runs of mean `case_run`
 * (0 = all upper).  Returns 0. */
int pavsynth_random_seq(uint64_t seed, uint8_t *out, uint64_t len, double case_run) {
    rng_t r; rng_seed(&r, seed);
    int lower = 0;
    uint64_t next_flip = case_run > 0 ? rng_geom(&r, 1.0 / case_run) : UINT64_MAX;
    uint64_t i = 0;
    while (i < len) {
        uint64_t w = rng_next(&r);
        for (int k = 0; k < 32 && i < len; ++k, ++i, w >>= 2) {
            if (i == next_flip) { lower ^= 1; next_flip = i + rng_geom(&r, 1.0 / case_run); }
            out[i] = (uint8_t)(lower ? LO[w & 3] : UP[w & 3]);
        }
    }
    return 0;
}

/* In-place reverse complement (IUPAC-aware, case-preserving). */
void pavsynth_revcomp(uint8_t *s, uint64_t len) {
    init_comp();
    uint64_t i = 0, j = len;
    while (i + 1 < j) { --j; uint8_t a = COMP[s[i]], b = COMP[s[j]]; s[i] = b; s[j] = a; ++i; }
    if (i < j && j - i == 1) s[i] = COMP[s[i]];
}

typedef struct {
    double snv_rate;      /* per aligned reference base */
    double indel_rate;    /* per aligned reference base, 50/50 INS/DEL */
    double pareto_alpha;  /* indel length = min(floor(Pareto(alpha)), max_indel) */
    uint32_t max_indel;
    double tandem_frac;   /* fraction of insertions that copy the upstream bases */
    uint32_t clip;        /* hard-clipped bases on each end of the contig */
    double pair_frac;     /* fraction of indel events emitted as a matched DEL + INS of equal length (signature the
                             inversion flagging of rules/call_inv.snakefile:476-600 looks for); 0 keeps older seeds' output */
} pavsynth_params;

typedef struct { char *p; uint64_t n, cap; int overflow; } sbuf;
static void sb_op(sbuf *b, uint64_t len, char op) {
    char tmp[32]; int k = snprintf(tmp, sizeof tmp, "%llu%c", (unsigned long long)len, op);
    if (b->n + (uint64_t)k + 1 > b->cap) { b->overflow = 1; return; }
    memcpy(b->p + b->n, tmp, (size_t)k); b->n += (uint64_t)k; b->p[b->n] = 0;
}
typedef struct { sbuf *sb; char op; uint64_t len; uint64_t n_ops; } oprun;
static void op_push(oprun *o, char op, uint64_t len) {
    if (!len) return;
    if (o->op == op) { o->len += len; return; }
    if (o->len) { sb_op(o->sb, o->len, o->op); o->n_ops++; }
    o->op = op; o->len = len;
}
static void op_flush(oprun *o) { if (o->len) { sb_op(o->sb, o->len, o->op); o->n_ops++; } o->op = 0; o->len = 0; }

/*
 * Build one contig from reference segment ref[0..ref_len).
 *   inv[2*i], inv[2*i+1]: sorted, disjoint [start,end) intervals (segment coordinates) that are inverted in
 *   the contig and aligned straight through as =/X.
 * Outputs: tig (stored orientation: reverse-complemented when is_rev), its length, the CIGAR text in
 * reference orientation, and counts[0..5] = {n_ops, n_snv, n_ins, n_del, aligned_bases(=,X), tig_aligned}.
 * Returns 0, or -1 when a buffer is too small.
 */
int pavsynth_contig(uint64_t seed, const uint8_t *ref, uint64_t ref_len, const pavsynth_params *pp,
                    const uint64_t *inv, uint32_t n_inv, int is_rev,
                    uint8_t *tig, uint64_t tig_cap, uint64_t *tig_len,
                    char *cigar, uint64_t cigar_cap, uint64_t *counts) {
    init_comp();
    rng_t r; rng_seed(&r, seed);
    sbuf sb = {cigar, 0, cigar_cap, 0};
    if (cigar_cap) cigar[0] = 0;
    oprun o = {&sb, 0, 0, 0};
    uint64_t t = 0, n_snv = 0, n_ins = 0, n_del = 0, n_aligned = 0;
    const double p_evt = pp->snv_rate + pp->indel_rate;
#define PUT(c) do { if (t >= tig_cap) return -1; tig[t++] = (uint8_t)(c); } while (0)
    for (uint32_t i = 0; i < pp->clip; ++i) PUT(UP[rng_next(&r) & 3]);
    op_push(&o, 'H', pp->clip);

    uint64_t p = 0;
    uint32_t ii = 0;
    int force_match = 1;                    /* first aligned base and the base after an indel are '=' */
    uint64_t next_evt = rng_geom(&r, p_evt);
    while (p < ref_len) {
        if (ii < n_inv && p == inv[2 * ii]) {           /* inverted interval, aligned through */
            uint64_t a = inv[2 * ii], b = inv[2 * ii + 1];
            for (uint64_t q = a; q < b; ++q) {
                uint8_t c = COMP[ref[a + b - 1 - q]];
                PUT(c);
                int c1 = base_code(c), c2 = base_code(ref[q]);
                int same = (c1 >= 0 && c1 == c2) || (c1 < 0 && c2 < 0);
                op_push(&o, same ? '=' : 'X', 1);
                if (!same) n_snv++;
                n_aligned++;
            }
            p = b; ++ii; force_match = 1;
            next_evt = p + rng_geom(&r, p_evt);
            continue;
        }
        uint64_t stop = ref_len;
        if (ii < n_inv && inv[2 * ii] < stop) stop = inv[2 * ii];
        if (force_match || next_evt > p) {                /* run of matches up to the next event */
            uint64_t e = next_evt < stop ? next_evt : stop;
            if (e <= p) e = p + 1;
            if (force_match && e == p) e = p + 1;
            if (t + (e - p) > tig_cap) return -1;
            memcpy(tig + t, ref + p, (size_t)(e - p));
            t += e - p; op_push(&o, '=', e - p); n_aligned += e - p;
            p = e; force_match = 0;
            if (next_evt < p) next_evt = p;
            continue;
        }
        /* event at p (p < stop, previous op is '=') */
        next_evt = p + rng_geom(&r, p_evt);
        int last_base = (p + 1 >= stop);
        int code = base_code(ref[p]);
        double u = rng_unit(&r) * p_evt;
        if (code < 0 || last_base) {                      /* never mutate N or the last base before a boundary */
            PUT(ref[p]); op_push(&o, '=', 1); n_aligned++; ++p;
            continue;
        }
        if (u < pp->snv_rate) {                           /* SNV */
            int alt = (code + 1 + (int)(rng_next(&r) % 3)) & 3;
            int lower = (ref[p] >= 'a');
            PUT(lower ? LO[alt] : UP[alt]); op_push(&o, 'X', 1); n_snv++; n_aligned++; ++p;
            continue;
        }
        /* indel: length = min(floor(Pareto(alpha)), max_indel), Pareto scale 1 */
        double up = rng_unit(&r); if (up <= 0.0) up = 1e-300;
        double lf = floor(pow(up, -1.0 / pp->pareto_alpha));
        uint64_t len = lf > (double)pp->max_indel ? pp->max_indel : (uint64_t)lf;
        if (len < 1) len = 1;
        if (pp->pair_frac > 0.0 && rng_unit(&r) < pp->pair_frac && p + len + 4 < stop) {   /* matched DEL + INS */
            p += len; op_push(&o, 'D', len); n_del++;
            const uint64_t gap = 1 + (rng_next(&r) & 1);
            if (t + gap > tig_cap) return -1;
            memcpy(tig + t, ref + p, (size_t)gap); t += gap; op_push(&o, '=', gap); n_aligned += gap; p += gap;
            for (uint64_t k = 0; k < len; ++k) PUT(UP[rng_next(&r) & 3]);
            op_push(&o, 'I', len); n_ins++;
            force_match = 1;
            if (next_evt <= p) next_evt = p + 1;
            continue;
        }
        if (rng_next(&r) & 1) {                           /* INS */
            uint64_t aligned_t = t - pp->clip;
            if (rng_unit(&r) < pp->tandem_frac && aligned_t >= len) {
                if (t + len > tig_cap) return -1;
                for (uint64_t k = 0; k < len; ++k) tig[t + k] = tig[t - len + k];
                t += len;
            } else {
                for (uint64_t k = 0; k < len; ++k) PUT(UP[rng_next(&r) & 3]);
            }
            op_push(&o, 'I', len); n_ins++;
        } else {                                          /* DEL */
            if (p + len + 1 > stop) len = stop - p - 1;
            if (len < 1) { PUT(ref[p]); op_push(&o, '=', 1); n_aligned++; ++p; continue; }
            p += len; op_push(&o, 'D', len); n_del++;
        }
        force_match = 1;
        if (next_evt <= p) next_evt = p + 1;
    }
    for (uint32_t i = 0; i < pp->clip; ++i) PUT(UP[rng_next(&r) & 3]);
    op_push(&o, 'H', pp->clip);
    op_flush(&o);
#undef PUT
    if (sb.overflow) return -1;
    if (is_rev) pavsynth_revcomp(tig, t);
    *tig_len = t;
    counts[0] = o.n_ops; counts[1] = n_snv; counts[2] = n_ins; counts[3] = n_del;
    counts[4] = n_aligned; counts[5] = t - 2ull * pp->clip;
    return 0;
}
