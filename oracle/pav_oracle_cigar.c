/*
 * pav_oracle_cigar.c - scalar CPU restatement of PAV's CIGAR variant caller.
 * TEST INFRASTRUCTURE ONLY (see pav_oracle.h).  Follows, line by line in control flow:
 *   pavlib/align/align.py:286-322   cigar_str_to_tuples
 *   pavlib/call.py:542-592, 595-647 left_homology / right_homology
 *   pavlib/cigarcall.py:24-311      make_insdel_snv_calls (the per-row, per-op walk)
 * Sorting and text formatting (cigarcall.py:313-362) are done by the Python side of the tests.
 */
#include "pav_oracle.h"

#include <stdlib.h>
#include <string.h>

static inline uint8_t up(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; }
static inline int is_acgt_up(uint8_t c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }

/* ---- pavlib/align/align.py:286-322 -------------------------------------------------------------------- */
static int op_code(uint8_t c) {
    switch (c) {                       /* _CIGAR_OP_SET, align.py:10; BAM numbering */
        case 'M': return 0; case 'I': return 1; case 'D': return 2; case 'N': return 3; case 'S': return 4;
        case 'H': return 5; case 'P': return 6; case '=': return 7; case 'X': return 8; default: return -1;
    }
}

int orc_cigar_tokenize(const char *text, uint64_t len, uint32_t *ops, uint64_t cap, uint64_t *n_ops,
                       uint32_t *err_off, uint32_t *err_char) {
    uint64_t pos = 0, n = 0;
    while (pos < len) {                                                   /* align.py:303 */
        uint64_t len_pos = pos;
        while (len_pos < len && text[len_pos] >= '0' && text[len_pos] <= '9') ++len_pos;   /* :307 */
        if (len_pos >= len) {                                             /* cigar[len_pos] -> IndexError */
            *n_ops = n; *err_off = (uint32_t)pos; *err_char = 0; return ORC_ERR_TOK_TRUNCATED;
        }
        if (len_pos == pos) {                                             /* :310-313 */
            *n_ops = n; *err_off = (uint32_t)pos; *err_char = (uint8_t)text[pos]; return ORC_ERR_TOK_MISSING_LEN;
        }
        int code = op_code((uint8_t)text[len_pos]);
        if (code < 0) {                                                   /* :315-318 (reports cigar[pos]) */
            *n_ops = n; *err_off = (uint32_t)pos; *err_char = (uint8_t)text[pos]; return ORC_ERR_TOK_UNKNOWN_OP;
        }
        uint64_t v = 0;
        for (uint64_t i = pos; i < len_pos; ++i) v = v * 10 + (uint64_t)(text[i] - '0');
        if (n < cap) ops[n] = (uint32_t)(v << 4) | (uint32_t)code;
        ++n;
        pos = len_pos + 1;                                                /* :322 */
    }
    *n_ops = n;
    return ORC_OK;
}

/* ---- pavlib/call.py:542-592 --------------------------------------------------------------------------- */
int64_t orc_left_homology(int64_t pos_tig, const uint8_t *seq_tig, int64_t tig_len, const uint8_t *seq_sv, int64_t svlen) {
    (void)tig_len;
    if (!seq_sv || !seq_tig || svlen <= 0) return 0;
    int64_t hom_len = 0;
    while (hom_len <= pos_tig) {                                          /* call.py:572 */
        uint8_t b = seq_tig[pos_tig - hom_len];
        if (!is_acgt_up(b)) break;                                        /* :576 (upper case only) */
        int64_t m = (hom_len + 1) % svlen;                                /* seq_sv[-m]; -0 == 0  (:579) */
        uint8_t s = seq_sv[m == 0 ? 0 : svlen - m];
        if (s != b) break;
        ++hom_len;
    }
    return hom_len;
}

/* ---- pavlib/call.py:595-647 --------------------------------------------------------------------------- */
int64_t orc_right_homology(int64_t pos_tig, const uint8_t *seq_tig, int64_t tig_len, const uint8_t *seq_sv, int64_t svlen) {
    if (!seq_sv || !seq_tig || svlen <= 0) return 0;
    int64_t hom_len = 0, limit = tig_len - pos_tig;                       /* call.py:627 */
    while (hom_len < limit) {
        uint8_t b = seq_tig[pos_tig + hom_len];
        if (!is_acgt_up(b)) break;
        if (seq_sv[hom_len % svlen] != b) break;                      /* :637 */
        ++hom_len;
    }
    return hom_len;
}

/* ---- growable outputs --------------------------------------------------------------------------------- */
struct orc_calls {
    orc_snv *snv; uint64_t n_snv, cap_snv;
    orc_indel *indel; uint64_t n_indel, cap_indel;
    uint8_t *seq; uint64_t n_seq, cap_seq;
};
static void *grow(void *p, uint64_t *cap, uint64_t need, size_t el) {
    if (need <= *cap) return p;
    uint64_t c = *cap ? *cap : 1024;
    while (c < need) c *= 2;
    *cap = c;
    return realloc(p, (size_t)c * el);
}
uint64_t orc_calls_n_snv(const orc_calls *c) { return c->n_snv; }
uint64_t orc_calls_n_indel(const orc_calls *c) { return c->n_indel; }
uint64_t orc_calls_seq_bytes(const orc_calls *c) { return c->n_seq; }
const orc_snv *orc_calls_snv(const orc_calls *c) { return c->snv; }
const orc_indel *orc_calls_indel(const orc_calls *c) { return c->indel; }
const uint8_t *orc_calls_seq(const orc_calls *c) { return c->seq; }
void orc_calls_free(orc_calls *c) { if (c) { free(c->snv); free(c->indel); free(c->seq); free(c); } }

static const uint8_t *comp_table(void) {
    static uint8_t t[256]; static int init = 0;
    if (!init) {
        for (int i = 0; i < 256; ++i) t[i] = (uint8_t)i;
        const char *a = "ACGTRYSWKMBDHVNUacgtryswkmbdhvnu", *b = "TGCAYRSWMKVHDBNAtgcayrswmkvhdbna";
        for (int i = 0; a[i]; ++i) t[(uint8_t)a[i]] = (uint8_t)b[i];
        init = 1;
    }
    return t;
}

/* ---- pavlib/cigarcall.py:24-311 ----------------------------------------------------------------------- */
orc_calls *orc_cigar_call(const uint8_t *const *ref_seq, const uint64_t *ref_len, uint32_t n_ref,
                          const uint8_t *const *tig_seq, const uint64_t *tig_len, uint32_t n_tig,
                          const orc_aln *aln, uint32_t n_aln, const char *cigar_text, const uint64_t *cigar_off,
                          orc_cigar_err *err) {
    (void)n_ref; (void)n_tig;
    orc_calls *out = (orc_calls *)calloc(1, sizeof *out);
    memset(err, 0, sizeof *err);
    const uint8_t *comp = comp_table();

    uint8_t *rc_buf = NULL; uint64_t rc_cap = 0;         /* reverse-complemented contig, cigarcall.py:63-72 */
    uint8_t *ref_up = NULL; uint64_t ref_up_cap = 0;     /* seq_ref.upper(), cigarcall.py:74 (cached per record) */
    uint8_t *tig_up = NULL; uint64_t tig_up_cap = 0;     /* seq_tig.upper(), cigarcall.py:75 */
    int64_t cur_tig = -1, cur_ref = -1; int cur_rev = -1;
    const uint8_t *seq_tig = NULL;

    uint32_t *ops = NULL; uint64_t ops_cap = 0;

    for (uint32_t r = 0; r < n_aln; ++r) {                                /* cigarcall.py:50 */
        const orc_aln *a = &aln[r];
        const int is_rev = a->rev != 0;
        const uint8_t *seq_ref = ref_seq[a->ref_id];                      /* :58-61 */
        const int64_t seq_ref_len = (int64_t)ref_len[a->ref_id];
        const int64_t seq_tig_len = (int64_t)tig_len[a->tig_id];
        if (cur_tig != (int64_t)a->tig_id || cur_rev != is_rev) {         /* :63-72 */
            if (is_rev) {
                rc_buf = (uint8_t *)grow(rc_buf, &rc_cap, (uint64_t)seq_tig_len + 1, 1);
                const uint8_t *src = tig_seq[a->tig_id];
                for (int64_t i = 0; i < seq_tig_len; ++i) rc_buf[i] = comp[src[seq_tig_len - 1 - i]];
                seq_tig = rc_buf;
            } else {
                seq_tig = tig_seq[a->tig_id];
            }
            cur_tig = a->tig_id; cur_rev = is_rev;
            tig_up = (uint8_t *)grow(tig_up, &tig_up_cap, (uint64_t)seq_tig_len + 1, 1);
            for (int64_t i = 0; i < seq_tig_len; ++i) tig_up[i] = up(seq_tig[i]);
        }
        if (cur_ref != (int64_t)a->ref_id) {
            ref_up = (uint8_t *)grow(ref_up, &ref_up_cap, (uint64_t)seq_ref_len + 1, 1);
            for (int64_t i = 0; i < seq_ref_len; ++i) ref_up[i] = up(seq_ref[i]);
            cur_ref = a->ref_id;
        }

        /* tokenise this row (generator in the reference; errors surface when the walk reaches them) */
        const char *ctext = cigar_text + cigar_off[r];
        const uint64_t clen = cigar_off[r + 1] - cigar_off[r];
        uint64_t n_ops = 0; uint32_t e_off = 0, e_chr = 0;
        ops = (uint32_t *)grow(ops, &ops_cap, clen / 2 + 2, sizeof *ops);
        int tok_rc = orc_cigar_tokenize(ctext, clen, ops, ops_cap, &n_ops, &e_off, &e_chr);

        int64_t pos_ref = a->pos, pos_tig = 0;                            /* :78-79 */
        uint32_t cigar_index = 0;
        int last_op = -1; int64_t last_oplen = 0;                         /* :83-84 */

        for (uint64_t k = 0; k < n_ops; ++k) {                            /* :86 */
            const int op = (int)(ops[k] & 15);
            const int64_t oplen = (int64_t)(ops[k] >> 4);
            ++cigar_index;                                                /* :89 */
            if (op == 7) {                                                /* '=' :91-93 */
                pos_ref += oplen; pos_tig += oplen;
            } else if (op == 8) {                                         /* 'X' :95-139 */
                out->snv = (orc_snv *)grow(out->snv, &out->cap_snv, out->n_snv + (uint64_t)oplen, sizeof(orc_snv));
                for (int64_t i = 0; i < oplen; ++i) {
                    int64_t pr = pos_ref + i, pt = pos_tig + i;
                    orc_snv *s = &out->snv[out->n_snv++];
                    memset(s, 0, sizeof *s);
                    s->aln = r; s->pos = (uint32_t)pr;
                    s->ref = seq_ref[pr]; s->alt = seq_tig[pt];           /* :104-105 */
                    s->qry_pos = (uint32_t)(is_rev ? seq_tig_len - pt - 1 : pt);   /* :108-109 */
                }
                pos_ref += oplen; pos_tig += oplen;
            } else if (op == 1 || op == 2) {                              /* 'I' :141-213, 'D' :217-282 */
                const int is_ins = (op == 1);
                const uint8_t *seq = is_ins ? seq_tig + pos_tig : seq_ref + pos_ref;   /* :145 / :221 */
                const uint8_t *seq_u = is_ins ? tig_up + pos_tig : ref_up + pos_ref;   /* seq.upper() :146 / :222 */
                int64_t left_shift = 0;
                if (last_op == 7) {                                       /* :149-155 / :225-231 */
                    int64_t h = orc_left_homology(pos_ref - 1, ref_up, seq_ref_len, seq_u, oplen);
                    left_shift = h < last_oplen ? h : last_oplen;
                }
                const int64_t sv_pos_ref = pos_ref - left_shift;
                const int64_t sv_pos_tig = pos_tig - left_shift;
                int64_t hrl, hrr, htl, htr;
                orc_indel rec; memset(&rec, 0, sizeof rec);
                rec.aln = r; rec.op_index = cigar_index; rec.svlen = (uint32_t)oplen;
                rec.left_shift = (uint32_t)left_shift;
                if (is_ins) {
                    const int64_t sv_end_tig = sv_pos_tig + oplen;
                    if (left_shift != 0) { seq = seq_tig + sv_pos_tig; seq_u = tig_up + sv_pos_tig; }   /* :162-163,176 */
                    hrl = orc_left_homology(sv_pos_ref - 1, ref_up, seq_ref_len, seq_u, oplen);    /* :178 */
                    hrr = orc_right_homology(sv_pos_ref, ref_up, seq_ref_len, seq_u, oplen);       /* :179 */
                    htl = orc_left_homology(sv_pos_tig - 1, tig_up, seq_tig_len, seq_u, oplen);    /* :181 */
                    htr = orc_right_homology(sv_end_tig, tig_up, seq_tig_len, seq_u, oplen);       /* :182 */
                    rec.svtype = 0;
                    rec.pos = (uint32_t)sv_pos_ref; rec.end = (uint32_t)(sv_pos_ref + 1);         /* :157-158 */
                    if (is_rev) {                                                                  /* :167-173 */
                        rec.qry_end = (uint32_t)(seq_tig_len - sv_pos_tig);
                        rec.qry_pos = rec.qry_end - (uint32_t)oplen;
                    } else {
                        rec.qry_pos = (uint32_t)sv_pos_tig; rec.qry_end = (uint32_t)(sv_pos_tig + oplen);
                    }
                } else {
                    const int64_t sv_end_ref = sv_pos_ref + oplen;
                    hrl = orc_left_homology(sv_pos_ref - 1, ref_up, seq_ref_len, seq_u, oplen);    /* :247 */
                    hrr = orc_right_homology(sv_end_ref, ref_up, seq_ref_len, seq_u, oplen);       /* :248 */
                    htl = orc_left_homology(sv_pos_tig - 1, tig_up, seq_tig_len, seq_u, oplen);    /* :250 */
                    htr = orc_right_homology(sv_pos_tig, tig_up, seq_tig_len, seq_u, oplen);       /* :251 */
                    rec.svtype = 1;
                    rec.pos = (uint32_t)pos_ref; rec.end = (uint32_t)(pos_ref + oplen);           /* :258 un-shifted */
                    int64_t q = is_rev ? seq_tig_len - sv_pos_tig : sv_pos_tig;                   /* :239-242 */
                    rec.qry_pos = (uint32_t)q; rec.qry_end = (uint32_t)(q + 1);
                }
                rec.hom_ref_l = (uint32_t)hrl; rec.hom_ref_r = (uint32_t)hrr;
                rec.hom_tig_l = (uint32_t)htl; rec.hom_tig_r = (uint32_t)htr;
                rec.seq_off = out->n_seq;
                out->seq = (uint8_t *)grow(out->seq, &out->cap_seq, out->n_seq + (uint64_t)oplen + 1, 1);
                memcpy(out->seq + out->n_seq, seq, (size_t)oplen);        /* SEQ, case preserved (:197 / :266) */
                out->n_seq += (uint64_t)oplen;
                out->indel = (orc_indel *)grow(out->indel, &out->cap_indel, out->n_indel + 1, sizeof(orc_indel));
                out->indel[out->n_indel++] = rec;
                if (is_ins) pos_tig += oplen; else pos_ref += oplen;      /* :213 / :282 */
            } else if (op == 4 || op == 5) {                              /* 'S','H' :286-287 */
                pos_tig += oplen;
            } else {                                                      /* :289-307 */
                err->kind = (op == 0) ? ORC_ERR_CIGAR_M : ORC_ERR_CIGAR_OP;
                err->aln = r; err->op_index = cigar_index; err->op_char = (uint32_t)"MIDNSHP=X"[op];
                err->pos_ref = (uint32_t)pos_ref; err->pos_tig = (uint32_t)pos_tig;
                goto done;
            }
            last_op = op; last_oplen = oplen;                             /* :310-311 */
        }
        if (tok_rc != ORC_OK) {                  /* the generator raises after the valid tokens were consumed */
            err->kind = tok_rc; err->aln = r; err->op_index = e_off; err->op_char = e_chr;
            err->pos_ref = (uint32_t)pos_ref; err->pos_tig = (uint32_t)pos_tig;
            goto done;
        }
    }
done:
    free(rc_buf); free(ops); free(ref_up); free(tig_up);
    return out;
}
