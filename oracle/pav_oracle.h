/*
 * pav_oracle.h - CPU restatement of PAV's CIGAR-call + k-mer inversion-density hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call it, and only as the
 * checker / reported CPU baseline.  The product (pav_amd/, include/pav_amd.h) never links or imports it.
 *
 * Every function cites the reference file:line it restates (paths relative to the PAV 2.4.6 snapshot).
 * Pinning: checked against golden vectors produced by importing the reference itself in the build container
 * (tools/refharness/gen_golden_*.py -> tests/golden/), see tests/test_oracle_*.py.
 * Unpinned: the numeric k-mer encoding (kanapy is absent from the snapshot).
 */
#ifndef PAV_ORACLE_H
#define PAV_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Record layouts are byte-identical to include/pav_amd.h so tests can compare with one numpy dtype. */
typedef struct {
    uint32_t aln;        /* row number in the submitted alignment table                              */
    uint32_t pos;        /* POS (END = POS + 1)                                                      */
    uint32_t qry_pos;    /* 0-based position on the stored (forward) contig                          */
    uint8_t ref, alt;    /* case-preserved bases (alt in reference orientation)                      */
    uint16_t pad;
} orc_snv;

typedef struct {
    uint32_t aln;
    uint32_t op_index;   /* 1-based CIGAR operation index                                            */
    uint32_t pos, end;   /* POS, END                                                                 */
    uint32_t svlen;
    uint32_t qry_pos, qry_end;   /* QRY_REGION = tig:(qry_pos+1)-qry_end                             */
    uint32_t left_shift;
    uint32_t hom_ref_l, hom_ref_r, hom_tig_l, hom_tig_r;
    uint64_t seq_off;    /* offset of SEQ (svlen bytes) in the SEQ blob                              */
    uint8_t svtype;      /* 0 = INS, 1 = DEL                                                         */
    uint8_t pad[7];
} orc_indel;

typedef struct {
    uint32_t ref_id, tig_id;
    uint32_t pos;        /* POS of the alignment row                                                 */
    uint32_t rev;        /* REV                                                                      */
} orc_aln;

/* error kinds shared with the product's pav_cigar_err */
enum {
    ORC_OK = 0,
    ORC_ERR_CIGAR_M = 1,          /* pavlib/cigarcall.py:292-299 */
    ORC_ERR_CIGAR_OP = 2,         /* pavlib/cigarcall.py:301-307 (N, P) */
    ORC_ERR_TOK_MISSING_LEN = 3,  /* pavlib/align/align.py:310-313 */
    ORC_ERR_TOK_UNKNOWN_OP = 4,   /* pavlib/align/align.py:315-318 */
    ORC_ERR_TOK_TRUNCATED = 5     /* IndexError at pavlib/align/align.py:307 (text ends inside a length) */
};

typedef struct {
    int32_t kind;
    uint32_t aln;        /* row */
    uint32_t op_index;   /* 1-based op index (CIGAR errors) or byte offset in the row's text (tokenizer) */
    uint32_t op_char;    /* offending character                                                      */
    uint32_t pos_ref, pos_tig;
} orc_cigar_err;

/* pavlib/align/align.py:286-322.  ops[i] = len << 4 | BAM opcode (M0 I1 D2 N3 S4 H5 P6 =7 X8). */
int orc_cigar_tokenize(const char *text, uint64_t len, uint32_t *ops, uint64_t cap, uint64_t *n_ops,
                       uint32_t *err_off, uint32_t *err_char);

/* pavlib/call.py:542-592 and :595-647, verbatim semantics: inputs must already be upper case (lower-case
 * bases never match), exactly like the reference functions; orc_cigar_call passes upper-cased copies as
 * pavlib/cigarcall.py:74-75,146,176 do. */
int64_t orc_left_homology(int64_t pos_tig, const uint8_t *seq_tig, int64_t tig_len, const uint8_t *seq_sv, int64_t svlen);
int64_t orc_right_homology(int64_t pos_tig, const uint8_t *seq_tig, int64_t tig_len, const uint8_t *seq_sv, int64_t svlen);

typedef struct orc_calls orc_calls;

/* pavlib/cigarcall.py:24-311 (rows in table order, records in emission order; the caller sorts). */
orc_calls *orc_cigar_call(const uint8_t *const *ref_seq, const uint64_t *ref_len, uint32_t n_ref,
                          const uint8_t *const *tig_seq, const uint64_t *tig_len, uint32_t n_tig,
                          const orc_aln *aln, uint32_t n_aln, const char *cigar_text, const uint64_t *cigar_off,
                          orc_cigar_err *err);
uint64_t orc_calls_n_snv(const orc_calls *);
uint64_t orc_calls_n_indel(const orc_calls *);
uint64_t orc_calls_seq_bytes(const orc_calls *);
const orc_snv *orc_calls_snv(const orc_calls *);
const orc_indel *orc_calls_indel(const orc_calls *);
const uint8_t *orc_calls_seq(const orc_calls *);
void orc_calls_free(orc_calls *);

/* ---- k-mer state + density scan ---------------------------------------------------------------------------- */
typedef struct {
    int32_t k;                      /* -k (31)                                    scripts/density.py:438          */
    uint32_t min_informative;       /* --mininf (2000)                            :449                            */
    uint32_t min_state_count;       /* --minstatecount (20)                       :462                            */
    double den_smooth;              /* --densmooth (1)                            :455                            */
    uint32_t state_run_smooth;      /* --staterunsmooth (20)                      :468                            */
    double state_run_delta;         /* --staterundelta (0.005)                    :476                            */
    uint32_t max_ref_kmer_count;    /* MAX_REF_KMER_COUNT (100)                   :47                             */
} orc_den_params;

typedef struct {
    int32_t status;                 /* 0 finalised table; 1 un-finalised (< mininf rows, STATE = -1); 125 soft fail */
    int32_t fail_kind;              /* status 125: 1 = no reference k-mers, 2 = k-mer count above the limit        */
    uint32_t n;                     /* table rows                                                                   */
    uint32_t max_count;             /* largest reference k-mer count                                                */
    uint64_t max_kmer;              /* first-inserted k-mer with that count (fail_kind 2; message text)             */
    uint32_t state_count[3];
    double h[3];                    /* KDE bandwidth (cho_cov) per state                                            */
    uint64_t n_eval;                /* evaluation points actually computed (sampled + filled)                       */
} orc_density_info;

typedef struct { int32_t state; uint32_t count; int64_t pos, end; } orc_run;   /* rl_encoder tuple */

typedef struct orc_density orc_density;

/* scripts/density.py main + get_smoothed_density on an already extracted reference / contig region (ASCII). */
orc_density *orc_density_run(const uint8_t *ref_seq, uint64_t ref_len, const uint8_t *tig_seq, uint64_t tig_len,
                             int ref_rc, const orc_den_params *pp);
/* Threads for the KDE of the calls that follow (default 1: the scalar port; tests of 0.1 - 1.2 Mbp regions raise it).  Every
 * evaluation point is summed by one thread in scipy's order, so the result does not depend on the count. */
void orc_density_set_threads(int n);
const orc_density_info *orc_density_get_info(const orc_density *);
const int64_t *orc_density_index(const orc_density *);
const int8_t *orc_density_state_mer(const orc_density *);
const int8_t *orc_density_state(const orc_density *);
const double *orc_density_kern(const orc_density *, int state);
const uint64_t *orc_density_kmer(const orc_density *);
const uint8_t *orc_density_interp(const orc_density *);
void orc_density_free(orc_density *);

/* pavlib/density.py:330-361; returns the number of runs (may exceed cap; only cap are written). */
uint32_t orc_rl_encode(const int8_t *state, const int64_t *index, uint32_t n, orc_run *runs, uint32_t cap);

/* pavlib/inv.py:457-561: flank 0 '' / 1 UP / 2 DN; match 0 '' / 1 SAME / 2 OTHER / 3 NaN. */
void orc_annotate(const uint64_t *kmer, const int64_t *index, uint32_t n, int k, int64_t qry_index_base,
                  int64_t up_pos, int64_t up_end, int64_t dn_pos, int64_t dn_end,
                  const uint8_t *ref_up, uint64_t ref_up_len, const uint8_t *ref_dn, uint64_t ref_dn_len,
                  uint8_t *flank, uint8_t *match);

uint64_t orc_kmer_rc(uint64_t kmer, int k);
uint64_t orc_kmer_canonical(uint64_t kmer, int k);

#ifdef __cplusplus
}
#endif
#endif
