/*
 * pav_amd.h - C ABI of libpav_amd.so: the MI355X (gfx950) variant-calling core for PAV's hot path.
 *
 * This is the drop-in boundary.  PAV (EichlerLab/pav 2.4.6) is pure Python and has no FFI of its own, so each
 * entry point below cites the Python interface it replaces (file:line in the reference snapshot); the ctypes
 * binding a PAV maintainer would add is shown in INTEGRATION.md and implemented in pav_amd/_lib.py.
 *
 * Conventions
 *   - plain C types, caller-owned host buffers; no torch / numpy types.
 *   - every function returns 0 (PAV_OK) or a negative PAV_E* code; pav_last_error() gives the message.
 *   - one pav_ctx per (process, GPU).  Calls on one context are serialised on its HIP stream; contexts are
 *     independent, so N Snakemake jobs / N ranks drive N GPUs with no collective.
 *   - there is NO CPU fallback: without a usable gfx950 device pav_create() fails.
 *   - coordinates are 0-based half-open (BED) unless stated; sequence lengths must be < 2^32 - 256 bases
 *     and CIGAR operation lengths < 2^28 (the BAM limit).
 */
#ifndef PAV_AMD_H
#define PAV_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PAV_ABI_VERSION 2

enum {
    PAV_OK = 0,
    PAV_E_ARG = -1,        /* bad argument                                                             */
    PAV_E_HIP = -2,        /* HIP runtime error (message has the HIP error string)                     */
    PAV_E_NODEV = -3,      /* no usable gfx950 device                                                  */
    PAV_E_CIGAR = -4,      /* illegal / malformed CIGAR: details via pav_cigar_error()                 */
    PAV_E_STATE = -5,      /* call order violated (e.g. call before load)                              */
    PAV_E_LIMIT = -6,      /* a documented size limit was exceeded                                     */
    PAV_E_TRIM = -7        /* alignment trimming would have raised: details via pav_trim_error()      */
};

enum { PAV_ROLE_REF = 0, PAV_ROLE_TIG = 1 };

typedef struct pav_ctx pav_ctx;

/* ---- context ------------------------------------------------------------------------------------------ */
int pav_abi_version(void);
int pav_device_count(void);                       /* number of visible HIP devices (0 if none / no driver) */
pav_ctx *pav_create(int device_id);               /* NULL on failure; then pav_last_error(NULL) explains   */
void pav_destroy(pav_ctx *ctx);
const char *pav_last_error(const pav_ctx *ctx);
int pav_device_name(const pav_ctx *ctx, char *buf, int buf_len);
int pav_device_pci_bus_id(const pav_ctx *ctx, char *buf, int buf_len);   /* "0000:c5:00.0" (buf_len >= 16): which GPU a rank drives */
int pav_sync(pav_ctx *ctx);                       /* hipStreamSynchronize on the context's stream          */
int pav_mem_info(pav_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes);   /* HBM of the context's GPU (hipMemGetInfo) */
/* Density work done by the context since it was created (for roofline accounting, bench.py): out[0] evaluation points of
 * the kernel densities (sampled + filled), out[1] (point, run of consecutive INDEX values) pairs the closed-form sums went
 * over, out[2] (point, data point) pairs - what scipy's double loop would have gone over (~25 flop each). */
int pav_kde_work(const pav_ctx *ctx, double out[3]);
/* Host waits of the CALLING thread since it started (every wait of the library for a stream or an event goes through one function):
 * out[0] seconds spent waiting, out[1] number of waits.  wall time of a pass - its waits = what the lane's host thread computes
 * (bench.py `host`; the reference has no counterpart: pavlib runs its stages as separate processes, rules/call_inv.snakefile:145-196).
 * PAV_WAIT = yield | spin | block picks how a thread waits (default yield: polls an event with sched_yield() in between, so a waiting
 * lane gives its core to any thread that can run). */
int pav_wait_stats(double out[2]);
/* Large device blocks that a context gives up (sequence arenas, record and table buffers of a destroyed context) wait on a
 * process-wide list per GPU for the next context - a context per haplotype then pays neither hipMalloc nor the driver's clearing of
 * freed memory.  At most PAV_DEVICE_POOL_GB (default 24) and a quarter of the device's memory are kept per GPU; PAV_DEVICE_POOL=0
 * turns the list off.  pav_device_pool_trim gives the idle blocks of one device (device_id < 0: of all) back to the driver and
 * returns the bytes freed - for a process that goes on to use the GPU with something else. */
uint64_t pav_device_pool_trim(int device_id);

/* ---- sequence store ----------------------------------------------------------------------------------- *
 * Replaces: pysam.FastaFile.fetch of whole records + str.upper() of the whole chromosome / contig per
 * alignment row (pavlib/cigarcall.py:58-75) and Bio reverse_complement (:70).  All records of one role are
 * uploaded as ASCII (kept in HBM for case-exact REF/ALT/SEQ output) and packed on the device into a 2-bit
 * plane (A0 C1 G2 T3, case folded) plus a 1-bit non-ACGT plane: the "upper-case view" the reference builds
 * with str.upper().  Reverse-complemented contigs are never materialised; kernels index them in place.
 * The reference (PAV_ROLE_REF) is packed in full by this call.  The contig planes (PAV_ROLE_TIG) are filled on demand:
 * the breakpoint-homology scans of pav_cigar_call / pav_homology decode their windows from the ASCII arena, a density
 * batch packs the blocks under its regions first, pav_cigar_verify and pav_seq_share pack everything (identical results;
 * the full pack is the largest kernel of a pass otherwise).  PAV_EAGER_PACK=1 in the environment packs contigs here too.
 */
int pav_seq_load(pav_ctx *ctx, int role, uint32_t n_seq, const uint8_t *const *ascii, const uint64_t *len);
/* Use the store `from` holds for `role` (no copy: both contexts read the same planes in HBM; they must be on the same GPU).
 * This is how several haplotypes are kept resident against ONE reference (BASELINE.json configs[3], [4]; the reference
 * treats haplotypes as independent jobs, README.md:83-86, rules/call.snakefile:755-786): one context per haplotype - each
 * with its own streams, contigs, alignment tables and results, drivable from its own host thread - all sharing the
 * reference context's PAV_ROLE_REF store.  The planes are freed with their last user.  A later pav_seq_load for that
 * role gives the context a store of its own again; re-packing a shared store (pav_seq_pack) while others use it is
 * the caller's race.  Record names (pav_seq_set_names) are per context and are not copied.
 * Ordering: a pack `from` has queued and not finished (an asynchronous pav_seq_pack) is waited for here, and every reader of
 * the planes on either context orders itself behind the store's last full pack, whichever context queued it.  A store whose
 * planes are still on demand (PAV_ROLE_TIG after pav_seq_load) is packed in full by this call: `from` must not be driven
 * from another thread while it runs. */
int pav_seq_share(pav_ctx *ctx, const pav_ctx *from, int role);

/* Native FASTA reader (host only; no context needed).  Replaces pysam.FastaFile(path).fetch(name) of the reference
 * (pavlib/cigarcall.py:59-66, pavlib/seq.py:339-351) for whole records: plain, gzip and BGZF files (PAV bgzips its
 * FASTA files, rules/align.snakefile:32-50; BGZF blocks are inflated in parallel).  Records are contiguous ASCII byte
 * arrays with the line breaks removed and the case kept; names are the first word of the header line, as in faidx.
 * threads <= 0: one per host core (at most 32).  Errors: PAV_E_ARG, message via pav_last_error(NULL). */
typedef struct pav_fasta pav_fasta;
int pav_fasta_open(const char *path, int threads, pav_fasta **out);
void pav_fasta_close(pav_fasta *fa);
uint32_t pav_fasta_count(const pav_fasta *fa);
const char *pav_fasta_name(const pav_fasta *fa, uint32_t record);
uint64_t pav_fasta_length(const pav_fasta *fa, uint32_t record);
const uint8_t *pav_fasta_seq(const pav_fasta *fa, uint32_t record);     /* valid until pav_fasta_close              */
int pav_fasta_kind(const pav_fasta *fa);                                 /* 0 plain text, 1 gzip stream, 2 BGZF      */
/* pav_seq_load + pav_seq_set_names of the chosen records, in the order given. */
int pav_seq_load_fasta(pav_ctx *ctx, int role, const pav_fasta *fa, uint32_t n_records, const uint32_t *records);
/* Every record of a FASTA file (plain, gzip, BGZF) into the store of `role`, names set - WITHOUT the host-side parse: the file's text
 * crosses PCIe as it is - a plain file's text; a BGZF file's members, the form PAV keeps its FASTA files in (rules/call.snakefile:796
 * contigs_{hap}.fa.gz, data/ref/ref.fa.gz), which are inflated ON THE DEVICE (inflate.hip; PAV_FASTA_INFLATE=host: by host threads); a gzip
 * stream that is not BGZF is inflated on the host first - read in parallel pieces straight into pinned memory, and the text
 * loses its header lines and line breaks on the device (fastadev.hip).  Same arena bytes as pav_fasta_open + pav_seq_load_fasta of all
 * records; *n_records = their number.  The two roles of one context may be loaded from two threads at the same time.
 * Replaces pysam.FastaFile(fa).fetch(name) of whole records (pavlib/cigarcall.py:59-66, pavlib/seq.py:339-351). */
int pav_seq_load_fasta_path(pav_ctx *ctx, int role, const char *path, int threads, uint32_t *n_records);
/* A BGZF file held in host memory -> its text: the members (independent gzip members of at most 64 KiB of text, SAM specification 4.1)
 * are decoded on the device - a lane per member walks the Huffman codes into tokens, a wave per member resolves the copies in LDS - and
 * every member's CRC-32 and ISIZE are checked (a corrupt member: PAV_E_ARG, the member named).  *out_len = the text's length, also when
 * out_cap is too small for it (PAV_E_LIMIT).  The inflate step of pav_seq_load_fasta_path, exposed for tests and for callers with BGZF
 * data of their own; what pysam.FastaFile does through htslib's bgzf reader (pavlib/cigarcall.py:59-64). */
int pav_bgzf_inflate(pav_ctx *ctx, const uint8_t *in, uint64_t n_in, uint8_t *out, uint64_t out_cap, uint64_t *out_len);
/* ASCII bytes [pos, pos + n) of record `rec` as they stand in the store (case preserved, forward strand): what
 * pysam.FastaFile.fetch(name, pos, pos + n) returns for a resident record (pavlib/seq.py:339-351, the SEQ column of rule
 * call_inv_batch, rules/call_inv.snakefile:203-282). */
int pav_seq_fetch(pav_ctx *ctx, int role, uint32_t rec, uint64_t pos, uint64_t n, uint8_t *out);
/* n slices in one round trip: slice i = [pos[i], pos[i] + len[i]) of record rec[i], at out + len[0] + ... + len[i - 1]. */
int pav_seq_fetch_many(pav_ctx *ctx, int role, uint32_t n, const uint32_t *rec, const uint64_t *pos, const uint64_t *len, uint8_t *out);
const char *pav_seq_name(const pav_ctx *ctx, int role, uint32_t i);      /* NULL: no such record / no names set */
uint64_t pav_seq_length(const pav_ctx *ctx, int role, uint32_t i);
/* The resident ASCII of `role` has changed (or a new haplotype's pass begins): its planes are stale.  PAV_ROLE_TIG with one
 * user: marks them so - readers fill what they need (pav_seq_load above).  PAV_ROLE_REF, a shared store, or PAV_EAGER_PACK=1:
 * re-runs the pack kernel over the whole arena, asynchronously on a side stream - it overlaps whatever the next calls queue
 * that does not read the packed planes (CIGAR tokenizer, walk, SNV emission); kernels that do read them wait for it on the
 * device.  pav_sync() waits for both streams. */
int pav_seq_pack(pav_ctx *ctx, int role);
int pav_seq_count(const pav_ctx *ctx, int role, uint32_t *n_seq, uint64_t *total_bases);

/* ---- CIGAR variant calling ---------------------------------------------------------------------------- *
 * Replaces pavlib.cigarcall.make_insdel_snv_calls (pavlib/cigarcall.py:24-311) together with
 * pavlib.align.cigar_str_to_tuples (pavlib/align/align.py:286-322) and pavlib.call.left_homology /
 * right_homology (pavlib/call.py:542-647).  The host keeps names/strings; the device returns integer
 * records in the reference's emission order (row, op, base) plus case-exact bases / SEQ bytes.
 */
typedef struct {          /* one alignment BED row (API_ALIGN.md:33-57), only what the walk reads          */
    uint32_t ref_id;      /* index of #CHROM in the PAV_ROLE_REF store                                      */
    uint32_t tig_id;      /* index of QRY_ID in the PAV_ROLE_TIG store                                      */
    uint32_t pos;         /* POS                                                                            */
    uint32_t rev;         /* REV (0 / 1)                                                                    */
} pav_aln;

typedef struct {          /* one SNV row (pavlib/cigarcall.py:95-135); END = pos + 1, SVLEN = 1             */
    uint32_t aln;         /* row number in the table given to pav_cigar_load                                */
    uint32_t pos;         /* POS                                                                            */
    uint32_t qry_pos;     /* 0-based position on the stored contig: QRY_REGION = tig:(qry_pos+1)-(qry_pos+1) */
    uint8_t ref, alt;     /* REF, ALT exactly as in the FASTA (alt in reference orientation)                */
    uint16_t pad;
} pav_snv;

typedef struct {          /* one INS / DEL row (pavlib/cigarcall.py:141-282)                                */
    uint32_t aln;
    uint32_t op_index;    /* 1-based CIGAR operation index within the row                                   */
    uint32_t pos, end;    /* POS, END (INS: shifted, END = POS + 1; DEL: un-shifted, cigarcall.py:258)      */
    uint32_t svlen;
    uint32_t qry_pos, qry_end;   /* QRY_REGION = tig:(qry_pos+1)-qry_end                                    */
    uint32_t left_shift;  /* LEFT_SHIFT                                                                     */
    uint32_t hom_ref_l, hom_ref_r, hom_tig_l, hom_tig_r;   /* HOM_REF = "l,r", HOM_TIG = "l,r"              */
    uint64_t seq_off;     /* offset of SEQ (svlen bytes) in the SEQ blob                                    */
    uint8_t svtype;       /* 0 = INS, 1 = DEL                                                               */
    uint8_t pad[7];
} pav_indel;

typedef struct {
    uint64_t n_ops;       /* CIGAR operations tokenised                                                     */
    uint64_t n_snv;
    uint64_t n_indel;
    uint64_t seq_bytes;   /* size of the SEQ blob                                                           */
    uint64_t aligned_bases;   /* sum of '=' and 'X' lengths: the numerator of the Gbp/s metric              */
} pav_cigar_counts;

enum {                    /* pav_cigar_err.kind                                                             */
    PAV_CIGAR_ERR_NONE = 0,
    PAV_CIGAR_ERR_M = 1,             /* 'M' op: RuntimeError of pavlib/cigarcall.py:292-299                 */
    PAV_CIGAR_ERR_OP = 2,            /* 'N' / 'P' op: RuntimeError of pavlib/cigarcall.py:301-307           */
    PAV_CIGAR_ERR_MISSING_LEN = 3,   /* pavlib/align/align.py:310-313                                       */
    PAV_CIGAR_ERR_UNKNOWN_OP = 4,    /* pavlib/align/align.py:315-318                                       */
    PAV_CIGAR_ERR_TRUNCATED = 5,     /* text ends inside a length: IndexError at align.py:307               */
    PAV_CIGAR_ERR_LEN_OVERFLOW = 6,  /* operation length >= 2^28 (not representable in BAM either)          */
    PAV_CIGAR_ERR_RANGE = 7          /* the row (POS + its = / X / D lengths, or its query lengths) does not fit the
                                        reference record / contig it names: pavlib indexes Python strings and raises
                                        IndexError at the first X base past the end (pavlib/cigarcall.py:104-105); here
                                        the whole row is refused before any kernel reads past a record; aln = the row   */
};

typedef struct {
    int32_t kind;
    uint32_t aln;         /* row                                                                            */
    uint32_t op_index;    /* kinds 1,2: 1-based op index; kinds 3-6: byte offset of the token in the row    */
    uint32_t op_char;     /* offending character as the reference prints it                                 */
    uint32_t pos_ref;     /* subject position when the walk reached the op (kinds 1,2)                      */
    uint32_t pos_tig;     /* query position (reference orientation) when the walk reached the op            */
} pav_cigar_err;

/* Upload the alignment table and the concatenated CIGAR text (row r = cigar_text[cigar_off[r] .. cigar_off[r+1]));
 * inputs are borrowed for the duration of the call only. */
int pav_cigar_load(pav_ctx *ctx, uint32_t n_aln, const pav_aln *aln, const uint8_t *cigar_text,
                   const uint64_t *cigar_off);
/* Tokenise + walk + homology on the device.  Results stay in HBM until fetched.  PAV_E_CIGAR when the
 * reference would have raised (first error in row order, then operation order, like the sequential walk). */
int pav_cigar_call(pav_ctx *ctx, pav_cigar_counts *counts);
int pav_cigar_error(const pav_ctx *ctx, pav_cigar_err *err);
/* Copy results to caller buffers sized from pav_cigar_counts (any pointer may be NULL to skip it). */
int pav_cigar_fetch(pav_ctx *ctx, pav_snv *snv, pav_indel *indel, uint8_t *seq_blob);
/* Verify mode: stream the packed reference and contig along every '=' / 'X' operation of the
 * last pav_cigar_call and count the bases that contradict the CIGAR - the reference trusts the aligner and never looks
 * at '=' runs (pavlib/cigarcall.py:91-93).  A base of an '=' run is wrong when the two codes differ or exactly one side
 * is non-ACGT; a base of an 'X' run is wrong when both sides are the same ACGT base or both are non-ACGT (case is folded,
 * contigs of reverse rows are read reverse-complemented).  first_bad_op: smallest global operation ordinal with a wrong
 * base (see pav_cigar_fetch_ops), ~0 when none.  One streaming kernel: 0.5 B of 2-bit planes per aligned base
 * (the non-ACGT planes are read only where the pack's per-1024-base summary marks a block). */
typedef struct { uint64_t eq_bases, eq_mismatch, x_bases, x_match, first_bad_op; } pav_verify_counts;
int pav_cigar_verify(pav_ctx *ctx, pav_verify_counts *counts);
/* Tokenised operations of the last pav_cigar_call: ops[i] = len << 4 | BAM opcode; op_off has n_aln + 1 entries. */
int pav_cigar_fetch_ops(pav_ctx *ctx, uint32_t *ops, uint64_t *op_off);

/* Native writer of the two tables rule call_cigar produces (rules/call.snakefile:813-846), straight from the records of
 * the last pav_cigar_call: FILTER (PASS iff POS > trim.POS and END < trim.END of the row's ALIGN_INDEX), the row order of
 * sort_values(['#CHROM','POS','END','ID']) (pavlib/cigarcall.py:320,343; SNV rows are sorted on the device) and the TSV
 * text DataFrame.to_csv(sep='\t', index=False) writes, byte for byte.  Names ending in ".gz" are written as concatenated
 * gzip members compressed in parallel.  Needs pav_seq_set_names for both stores. */
typedef struct {
    const char *hap;              /* HAP column                                                                    */
    const int64_t *align_index;   /* [n_aln] INDEX of every row of the table given to pav_cigar_load               */
    const int64_t *trim_pos;      /* [n_aln] POS of that INDEX in the trim-tigref table, -1 when absent            */
    const int64_t *trim_end;      /* [n_aln] END ...; both NULL: no FILTER column (the function-level tables)      */
    const char *snv_path;         /* NULL = skip                                                                   */
    const char *insdel_path;      /* NULL = skip                                                                   */
    int32_t gzip_level;           /* 1..9, 0 = 6                                                                   */
    int32_t threads;              /* formatting / compression threads, 0 = auto                                    */
    const int64_t *call_batch;    /* NULL: the tables of one call_cigar job.  [n_aln] CALL_BATCH of every row (0..15): the
                                     *merged* tables of rule call_cigar_merge (rules/call.snakefile:755-786) when all rows of
                                     a haplotype were called at once - the batch files are concatenated in batch order and
                                     stable-sorted by (#CHROM, POS, END, ID) (INS / DEL) resp. (#CHROM, POS) only (SNV), so
                                     equal keys keep batch order; needs <= 4096 reference records with non-numeric names
                                     (pandas would re-read numeric names as integers and order them numerically)         */
} pav_table_opts;
int pav_cigar_write_tables(pav_ctx *ctx, const pav_table_opts *opts, uint64_t *n_snv_rows, uint64_t *n_insdel_rows);
/* The same write in two halves.  _begin does the device part (order, FILTER, record streams to host memory the library
 * owns; every array of `opts` is copied) and starts the host part - text, gzip members, the files - on a thread of its own;
 * the context may go on with pav_cigar_flag / the inversion scan meanwhile (they read the resident records, the writer no
 * longer does).  _end waits for the files and reports the writer's error, if any.  One write at a time per context;
 * pav_destroy waits for a write that was never ended.  (rules/call.snakefile:765-786 is a job of its own in the reference:
 * nothing downstream of it on this path reads the two tables.) */
int pav_cigar_write_tables_begin(pav_ctx *ctx, const pav_table_opts *opts);
int pav_cigar_write_tables_end(pav_ctx *ctx, uint64_t *n_snv_rows, uint64_t *n_insdel_rows);

/* gzip of a host buffer on the device: the compressor of the table writers above (deflate.hip - 64 KiB of text per wave, a sliding
 * window in LDS, dynamic Huffman blocks joined on byte boundaries into ONE member), exposed for tests and for callers with text
 * of their own.  `out` receives a complete gzip file (RFC 1952) that inflates to `text`; *out_len its size, PAV_E_LIMIT when
 * out_cap is smaller (n + n / 2 + 4096 always suffices).  level 1..9 as zlib's (0 = 6): how far the match finder looks.
 * Replaces the gzip step of DataFrame.to_csv(compression='gzip') (rules/call.snakefile:845-846, rules/call_inv.snakefile:279-291). */
int pav_gzip_buffer(pav_ctx *ctx, const uint8_t *text, uint64_t n, int level, uint8_t *out, uint64_t out_cap, uint64_t *out_len);
/* The same for n texts in one launch set (the sixty per-batch INV tables of a haplotype): member i at out + out_off[i], out_len[i]
 * bytes; out_cap >= sum over i of (lens[i] + lens[i] / 2 + 4096) always suffices. */
int pav_gzip_buffers(pav_ctx *ctx, uint32_t n, const uint8_t *const *texts, const uint64_t *lens, int level, uint8_t *out, uint64_t out_cap,
                     uint64_t *out_off, uint64_t *out_len);

/* Lift-over tables for pavlib.align.AlignLift (pavlib/align/lift.py:380-476, `_add_align`): tokenises every row's
 * CIGAR on the device and returns, per operation, the subject position where it starts (absolute, row POS included)
 * and the query position where it starts (alignment orientation, clipping included).  Advance rules are AlignLift's:
 * M / = / X move both axes, I / S / H the query, D the subject; N and P move nothing (the host raises on use).
 * Two-call protocol: first with ops == NULL (tokenise, *n_ops_out = total operations; PAV_E_CIGAR + pav_cigar_error on a
 * malformed row), then with buffers of n_ops_out entries (op_off: n_aln + 1) to copy the tables out. */
int pav_align_index(pav_ctx *ctx, uint32_t n_aln, const uint32_t *row_pos, const uint8_t *cigar_text,
                    const uint64_t *cigar_off, uint64_t *n_ops_out, uint32_t *ops, uint64_t *op_off, uint32_t *sub_begin,
                    uint32_t *qry_begin);

/* Direct entry to the device homology routines (unit parity tests against pavlib/call.py:542-647).
 * Sequence `seq_id` / `sv_seq_id` index the store of the given role; `rev` views that record reverse-
 * complemented (pavlib/cigarcall.py:69-70).  dir 0 = left_homology, 1 = right_homology. */
typedef struct {
    int32_t role, seq_id, rev;          /* the sequence scanned (seq_tig argument)                          */
    int64_t pos;                        /* pos_tig argument                                                 */
    int32_t sv_role, sv_seq_id, sv_rev; /* where seq_sv lives                                               */
    int64_t sv_pos;                     /* start of seq_sv in that (oriented) record                        */
    uint32_t svlen;
    int32_t dir;
} pav_hom_query;
int pav_homology(pav_ctx *ctx, uint32_t n, const pav_hom_query *q, uint32_t *out);

/* ---- k-mer state + density scan (inversion caller) -------------------------------------------------------- *
 * Replaces the subprocess `scripts/density.py` that pavlib.inv.scan_for_inv spawns once per scan iteration
 * (pavlib/inv.py:246-288): pavlib.seq.ref_kmers (pavlib/seq.py:305-325), the low-complexity gate and
 * orientation handling of scripts/density.py:508-545, get_smoothed_density (scripts/density.py:154-342:
 * STATE_MER, compaction, scipy gaussian_kde x3, interpolate / fill, spike rule, arg-max STATE) and
 * pavlib.density.rl_encoder (pavlib/density.py:330-361).  A batch of independent (reference region, contig
 * region) jobs is processed per call; tables stay in HBM and are fetched only for regions that become calls.
 * k <= 32 (k = 32: the k-mer sets live in HBM tables - pav_den_params.kmer_mode is ignored - because an LDS slot keeps two
 * orientation bits above a canonical k-mer of at most 62 bits; the reference takes any k, rules/call_inv.snakefile:131, its default
 * is 31).  Float columns follow scipy's arithmetic order (data ascending per evaluation point) with the
 * device's exp(); they agree with the reference to ~1e-13 relative, STATE / STATE_MER / INDEX / runs exactly.
 */
typedef struct {
    uint32_t ref_id, tig_id;        /* records in the REF / TIG stores                                      */
    uint64_t ref_pos, ref_end;      /* region_ref  [pos, end)   (--refregion, BED coordinates)              */
    uint64_t tig_pos, tig_end;      /* region_tig  [pos, end) on the stored contig (--tigregion)            */
    uint32_t ref_rc;                /* -r true: reverse-complement the reference k-mer set (region_tig.is_rev) */
    uint32_t state_run_smooth;      /* --staterunsmooth for this region (pavlib/inv.py:259 srs_tree lookup) */
} pav_den_job;

typedef struct {
    int32_t k;                      /* -k               (31)     scripts/density.py:438                     */
    uint32_t min_informative;       /* --mininf         (2000)   :449                                       */
    uint32_t min_state_count;       /* --minstatecount  (20)     :462                                       */
    double den_smooth;              /* --densmooth      (1)      :455                                       */
    double state_run_delta;         /* --staterundelta  (0.005)  :476                                       */
    uint32_t max_ref_kmer_count;    /* MAX_REF_KMER_COUNT (100)  :47                                        */
    uint32_t kde_mode;              /* PAV_KDE_RUNS (default) or PAV_KDE_DIRECT, see below                  */
    uint32_t kmer_mode;             /* PAV_KMER_LDS (default) or PAV_KMER_HBM, see below                    */
    uint32_t guard_cap;             /* near-tie guard: entries of the re-evaluation list (0 = 1 << 20); when more sites
                                       are flagged the whole batch is evaluated again in PAV_KDE_DIRECT order           */
    double guard_rel;               /* near-tie guard: relative margin below which a float decision is re-evaluated in
                                       scipy's accumulation order; 0 = default 1e-9, < 0 = guard off (see below)        */
} pav_den_params;

/* Density evaluation.  Both give the reference's KERN_* to ~1e-13 relative and identical STATE on every test vector.
 *   PAV_KDE_DIRECT  one exp() per (evaluation point, data point) pair, data accumulated in ascending order: the
 *                   arithmetic order of scipy's gaussian_kernel_estimate.  O(N_eval x N_data).
 *   PAV_KDE_RUNS    the data points of a state are runs of consecutive integers (INDEX_DEN = 0..n-1); the sum of the
 *                   Gaussian over a run is evaluated in closed form (Euler-Maclaurin: erf/erfc integral + endpoint
 *                   and odd-derivative corrections, remainder < 1e-15 for bandwidth >= 32; short runs and small
 *                   bandwidths fall back to direct terms).  O(N_eval x N_runs). */
enum { PAV_KDE_RUNS = 0, PAV_KDE_DIRECT = 1 };

/* Near-tie guard (hard part 2).  Three decisions of scripts/density.py depend on float64 densities:
 *   (a) the arg-max at the sampled sites (:250-255), which feeds `state_change` of the windows (:273-276);
 *   (b) `density_change`: max |delta KERN| > --staterundelta between two sampled sites (:277-281);
 *   (c) the final arg-max STATE of every row (:335-338), taken after the spike rule KERN > 1.0 -> 1 / KERN (:330-332).
 * The closed-form run sums of PAV_KDE_RUNS agree with scipy to ~1e-12 relative, so a decision whose margin is smaller than
 * guard_rel - (max - second) / max for an arg-max, | max|delta| - delta | / delta for (b) - is not trusted: the sampled sites
 * (and, for rows of evaluated windows, the rows) it depends on are evaluated again with one exp() per (site, data point)
 * pair accumulated in ascending data order - scipy's gaussian_kernel_estimate order - and everything downstream (windows,
 * interpolation, fill, spike rule, arg-max, runs) is redone from those values, until no new site is flagged.
 * The spike rule itself is continuous (|1/v - v| <= 2 |v - 1|: a row the reference inverts and we do not, or the reverse,
 * differs by at most twice its distance from 1.0), so rows with |KERN - 1| < guard_rel are only counted (n_spike_near: in
 * the interior of a long run KERN = 1.0 to the last bit and the branch is decided by rounding in the reference as well);
 * its discrete consequence, the arg-max taken afterwards, is covered by (c).
 * After re-evaluation the only differences from scipy left are the last bit of exp() and of np.cov's BLAS dot product
 * (machine dependent in the reference itself); decisions still within 1e-13 are reported as n_unresolved. */

/* Where the reference k-mer set of a region lives while STATE_MER is computed (identical results):
 *   PAV_KMER_LDS    partitioned by hash, one workgroup builds each partition in a 32 KiB LDS table and answers the
 *                   contig k-mers that hash to it (regions up to ~1.8 Mbp; larger ones use HBM tables).
 *   PAV_KMER_HBM    one open-addressing table per region in HBM, device-scope atomics.  Also used per region when a
 *                   partition overflows (kilobases of one repeated k-mer) and for the exact count of the count-limit
 *                   failure.  The environment variable PAV_KMER_HBM (any value) forces this mode for every call. */
enum { PAV_KMER_LDS = 0, PAV_KMER_HBM = 1 };

enum { PAV_DEN_OK = 0, PAV_DEN_UNFINALISED = 1, PAV_DEN_FAIL = 125 };   /* 125 = pavlib.constants.ERR_INV_FAIL */

typedef struct {
    int32_t status;                 /* PAV_DEN_OK: full table; PAV_DEN_UNFINALISED: < mininf rows, STATE = -1
                                       (scripts/density.py:193-195); PAV_DEN_FAIL: soft failure                 */
    int32_t fail_kind;              /* PAV_DEN_FAIL: 1 = no reference k-mers (:510-513), 2 = k-mer count (:516-527) */
    uint32_t n_rows;                /* informative k-mers = table rows                                        */
    uint32_t n_runs;                /* rl_encoder tuples                                                      */
    uint32_t max_count;             /* largest reference k-mer count                                          */
    uint32_t n_sample;              /* sampled sites                                                          */
    uint64_t max_kmer;              /* fail_kind 2: first k-mer (insertion order) with max_count, kanapy encoding */
    uint32_t state_count[3];        /* informative k-mers per STATE_MER after the min-state-count rule        */
    uint32_t n_near_tie;            /* float decisions (a)-(c) whose margin was below guard_rel               */
    uint64_t n_eval;                /* density evaluation points computed (sampled + filled)                  */
    double h[3];                    /* KDE bandwidth per state (scipy cho_cov)                                */
    uint32_t n_reeval;              /* sites evaluated again in scipy's accumulation order because of them    */
    uint32_t n_unresolved;          /* decisions still within 1e-13 after that: ambiguous in the reference too */
    uint32_t n_spike_near;          /* table values with |KERN - 1.0| < guard_rel (continuous branch, counted only) */
    uint32_t guard_fallback;        /* 1: the re-evaluation list overflowed, the batch was redone in PAV_KDE_DIRECT */
} pav_den_result;

typedef struct { int32_t state; uint32_t count; int64_t pos, end; } pav_run;   /* rl_encoder (state,count,pos,end) */

int pav_density_batch(pav_ctx *ctx, uint32_t n_jobs, const pav_den_job *jobs, const pav_den_params *params,
                      pav_den_result *results);
/* rl_encoder tuples of one job of the last batch (results[job].n_runs entries). */
int pav_density_runs(pav_ctx *ctx, uint32_t job, pav_run *runs);
/* Density table of one job of the last batch, results[job].n_rows entries each; any pointer may be NULL.
 * Columns: INDEX, STATE_MER, STATE, KERN_FWD, KERN_FWDREV, KERN_REV, KMER (scripts/density.py:341). */
int pav_density_table(pav_ctx *ctx, uint32_t job, int64_t *index, int8_t *state_mer, int8_t *state, double *kern_fwd,
                      double *kern_fwdrev, double *kern_rev, uint64_t *kmer);
/* pavlib.inv.annotate_inv_dup_mers (pavlib/inv.py:457-561) on the resident table of one job:
 * flank[i] 0 '' / 1 UP / 2 DN; match[i] 0 '' / 1 SAME / 2 OTHER / 3 NaN.  up_* / dn_* are the duplication
 * regions on the reference record ref_id (BED) and on the contig; qry_index_base is added to INDEX (:519). */
int pav_density_annotate(pav_ctx *ctx, uint32_t job, uint32_t ref_id, uint64_t ref_up_pos, uint64_t ref_up_end,
                         uint64_t ref_dn_pos, uint64_t ref_dn_end, int64_t qry_index_base, int64_t tig_up_pos,
                         int64_t tig_up_end, int64_t tig_dn_pos, int64_t tig_dn_end, uint8_t *flank, uint8_t *match);
/* K-mer helpers in the (assumed) kanapy integer encoding: A0 C1 G2 T3, first base most significant. */
uint64_t pav_kmer_rev_complement(uint64_t kmer, int k);
uint64_t pav_kmer_canonical(uint64_t kmer, int k);

/* ---- batched inversion scan (native driver) ----------------------------------------------------------------- *
 * Replaces the per-region loop of rule call_inv_batch (rules/call_inv.snakefile:185-200) around
 * pavlib.inv.scan_for_inv (pavlib/inv.py:149-454): flagged regions advance in lock-step, every scan iteration is one
 * pav_density_batch call; lift-over (pavlib/align/lift.py), region expansion (pavlib/seq.py:112-188), the stop /
 * expand rules, breakpoint regions, size checks and annotate_inv_dup_mers are done inside the library.  Log lines
 * and errors are the reference's, byte for byte.  pav_amd/inv.py holds the same state machine in Python (used when
 * a foreign AlignLift object or an N-tree is passed); both are tested against the reference's golden vectors. */
typedef struct {            /* one row of the trim-tigref alignment BED (AlignLift input, lift.py:20-49)            */
    uint32_t ref_id, tig_id;
    uint64_t pos, end;      /* POS, END                                                                            */
    uint64_t qry_pos, qry_end;   /* QRY_POS, QRY_END                                                               */
    uint32_t rev;           /* REV                                                                                 */
    uint32_t pad;
    int64_t index;          /* INDEX                                                                               */
} pav_inv_aln;

typedef struct { uint32_t ref_id; uint32_t pad; uint64_t pos, end; } pav_inv_region;   /* flagged region (BED)     */
typedef struct { double begin, end; uint32_t value; uint32_t pad; } pav_srs;      /* state-run-smooth interval     */

typedef struct {
    int64_t max_region_size;        /* MAX_REGION_SIZE / inv_region_limit; 0 = unlimited (inv.py:226)               */
    int32_t min_exp_count;          /* inv_min_expand / DEFAULT_MIN_EXP_COUNT (inv.py:297)                          */
    uint32_t n_srs;                 /* get_srs_tree intervals (inv.py:564-620)                                      */
    const pav_srs *srs;
    pav_den_params den;
    uint32_t lazy_tables;           /* 0: the density tables of the calls are copied to the library's pinned host memory
                                       behind the scan (copy stream).  1: they stay packed in HBM; the first pav_inv_table* /
                                       pav_inv_write_tables call that needs them brings them over (like the SNV / INDEL
                                       records of pav_cigar_call, which stay in HBM until pav_cigar_fetch)              */
    uint32_t reserved;
} pav_inv_params;

typedef struct {            /* pavlib.seq.Region as the scan reports it                                            */
    uint32_t seq_id;        /* record in the REF store (ref_*) or the TIG store (tig_*)                            */
    uint32_t is_rev;
    uint64_t pos, end;
    uint32_t n_aln[2];      /* entries of pos_aln_index / end_aln_index (1, or 2 after a gap lift)                 */
    int64_t aln_index[2][2];
} pav_inv_rgn;

enum { PAV_INV_NONE = 0, PAV_INV_CALL = 1, PAV_INV_ERROR = 2 };   /* None / InvCall / RuntimeError               */

typedef struct {
    int32_t outcome;
    uint32_t found;         /* 1 when 'INV Found: ...' is printed (inv.py:408), even if a size check then rejects  */
    uint32_t iterations;
    uint32_t n_rows;        /* density table rows of the call                                                      */
    uint64_t svlen;
    pav_inv_rgn ref_outer, ref_inner, tig_outer, tig_inner, ref_discovery, tig_discovery;
    uint32_t log_bytes, error_bytes;
    uint32_t n_near_tie, n_unresolved;   /* near-tie guard, summed over the region's scan iterations (pav_den_result)       */
} pav_inv_result;

int pav_seq_set_names(pav_ctx *ctx, int role, uint32_t n, const char *const *names);   /* record names for log text */
int pav_inv_load_alignments(pav_ctx *ctx, uint32_t n, const pav_inv_aln *aln, const uint8_t *cigar_text,
                            const uint64_t *cigar_off);
int pav_inv_scan_batch(pav_ctx *ctx, uint32_t n_regions, const pav_inv_region *regions, const pav_inv_params *params,
                       pav_inv_result *results);
/* what 0: the region's log lines (log_bytes + 1 buffer); what 1: the RuntimeError text (error_bytes + 1). */
int pav_inv_text(pav_ctx *ctx, uint32_t region, int what, char *buf, uint32_t buf_len);
/* The same texts for every region of the last scan at once: region i is buf[off[i] .. off[i + 1]) (no terminators); off has
 * n_regions + 1 entries; buf_len >= the sum of log_bytes (what 0) / error_bytes (what 1).  what 2: the line scan_for_inv prints
 * for a region with an inversion signature ('INV Found: outer=..., inner=... (ref outer=..., inner=...)' + newline,
 * pavlib/inv.py:408), empty for the other regions.  buf NULL: only off is filled (the sizes). */
int pav_inv_texts(pav_ctx *ctx, int what, char *buf, uint64_t buf_len, uint64_t *off);
/* Density table of a call incl. FLANK (0 '' / 1 UP / 2 DN) and MATCH (0 '' / 1 SAME / 2 OTHER / 3 NaN). */
int pav_inv_table(pav_ctx *ctx, uint32_t region, int64_t *index, int8_t *state_mer, int8_t *state, double *kern_fwd,
                  double *kern_fwdrev, double *kern_rev, uint64_t *kmer, uint8_t *flank, uint8_t *match);

/* Zero-copy access: pointers to the library's pinned host copy of a call's table (INDEX as uint32).  Valid until the
 * next pav_inv_scan_batch / pav_destroy on this context. */
int pav_inv_table_view(pav_ctx *ctx, uint32_t region, uint32_t *n_rows, const uint32_t **index, const int8_t **state_mer,
                       const int8_t **state, const double **kern_fwd, const double **kern_fwdrev, const double **kern_rev,
                       const uint64_t **kmer, const uint8_t **flank, const uint8_t **match);
/* All call tables of the last scan at once: rows of region i are written at row_off[i] of every column (regions without
 * a call are skipped); the caller sizes the columns as the sum of n_rows. */
int pav_inv_tables(pav_ctx *ctx, uint32_t n_regions, const uint64_t *row_off, int64_t *index, int8_t *state_mer,
                   int8_t *state, double *kern_fwd, double *kern_fwdrev, double *kern_rev, uint64_t *kmer, uint8_t *flank,
                   uint8_t *match);
/* Density tables of rule call_inv_batch (rules/call_inv.snakefile:287-291: call.df.to_csv(path, sep='\t', index=False,
 * compression='gzip')) for calls of the last scan, written from the library's host copies: the text pandas writes (float
 * columns as repr(), NaN as empty), names ending in ".gz" as concatenated gzip members compressed in parallel.
 * threads <= 0: up to 16 host threads; gzip_level <= 0: 6. */
int pav_inv_write_tables(pav_ctx *ctx, uint32_t n, const uint32_t *regions, const char *const *paths, int threads,
                         int gzip_level);
/* repr() of a float64 as DataFrame.to_csv writes it (shortest round-trip digits, Python's positional / exponent rule);
 * returns the length, out is NUL-terminated.  Text helper of the writers, exposed for unit tests. */
int pav_repr_f64(double value, char *out, int out_len);

/* ---- alignment ingest: SAM -> alignment table ----------------------------- *
 * The record loop of pavlib.align.get_align_bed (pavlib/align/align.py:666-794; rule align_get_read_bed,
 * rules/align.snakefile:101-171) without pysam: SAM text (plain, gzip, BGZF) is parsed on host threads into the quantities
 * that function reads from pysam's AlignedSegment (restated from the SAM specification / htslib: reference_start = POS - 1,
 * reference_end = start + max(1, M+D+N+=+X), query_alignment_start / _end from the soft clips, flags), the CIGAR with
 * soft clipping folded into hard clipping (clip_soft_to_hard, align.py:797-831) and count_cigar of that CIGAR
 * (align.py:534-663; err_kind as pav_trim_count).  Records that are unmapped, below min_mapq or without CIGAR are
 * dropped but counted: INDEX is the ordinal of the alignment line in the file (align.py:692).  Host only. */
typedef struct pav_sam pav_sam;
typedef struct {
    uint64_t n_records;             /* alignment lines in the file                                            */
    uint64_t n_rows;                /* records kept                                                           */
    uint32_t n_ref, n_qry;          /* distinct RNAME / QNAME among them                                      */
    uint64_t cigar_bytes, tag_bytes, header_bytes;
} pav_sam_info_t;
typedef struct {                    /* caller-allocated, n_rows entries (offset arrays n_rows + 1); any may be NULL */
    int64_t *index, *pos, *end;
    uint32_t *chrom_id, *qry_id;
    int64_t *query_alignment_start, *query_alignment_end;
    int64_t *clip_h;                /* leading H of the SAM CIGAR (align.py:708-711)                          */
    int64_t *tig_map_pos;           /* leading clipping of the transformed CIGAR (:715)                       */
    int32_t *mapq, *flag;
    uint8_t *has_m;                 /* an M operation is present (:725-729)                                   */
    uint8_t *status;                /* 0 ok; 1 clipping order pysam rejects; 2 clipping operations only       */
    int64_t *ref_bp, *tig_bp;       /* count_cigar of the transformed CIGAR                                   */
    uint32_t *err_kind, *err_op, *err_len, *err_char;
    uint8_t *cigar_text; uint64_t *cigar_off;
    uint8_t *tag_text;              /* RG value at [rg_off[i], ao_off[i]), AO value at [ao_off[i], rg_off[i + 1]) */
    uint64_t *rg_off, *ao_off;
    uint8_t *rg_kind, *ao_kind;     /* 0 absent, 1 integer (type i), 2 text, 3 float                          */
} pav_sam_cols;
int pav_sam_open(const char *path, int min_mapq, int threads, pav_sam **out);   /* message via pav_last_error(NULL) */
void pav_sam_close(pav_sam *sam);
int pav_sam_info(const pav_sam *sam, pav_sam_info_t *info);
int pav_sam_fetch(const pav_sam *sam, const pav_sam_cols *cols);
const char *pav_sam_name(const pav_sam *sam, int which /* 0 RNAME, 1 QNAME */, uint32_t id);
int pav_sam_header(const pav_sam *sam, uint8_t *buf /* header_bytes: the leading '@' lines */);

/* ---- alignment tables: native reader (reader half) -------------------------- *
 * The tables of results/{asm}/align/trim-{none,tig,tigref}/aligned_tig_{hap}.bed.gz (API_ALIGN.md:31-64) as rule call_cigar
 * reads them with pandas (rules/call.snakefile:805, 813-816): gzip or plain TSV with a header line; columns are found by
 * name, unknown columns are skipped.  Parsing needs no GPU and no context.  Fields are taken verbatim (no quoting rules:
 * PAV writes none for these tables); REV is the text True / False. */
typedef struct pav_bed pav_bed;
typedef struct {
    uint64_t n_rows, cigar_bytes;
    uint32_t n_chrom, n_qry;        /* distinct #CHROM / QRY_ID values, numbered in order of first appearance                */
    uint32_t columns;               /* bit i set: column i of (#CHROM, POS, END, INDEX, QRY_ID, QRY_POS, QRY_END, QRY_LEN,    */
    uint32_t pad;                   /* MAPQ, REV, CIGAR, CALL_BATCH) is present                                              */
} pav_bed_info_t;
typedef struct {                    /* destinations, n_rows entries each (cigar_off: n_rows + 1); NULL = not wanted         */
    uint32_t *chrom_id, *qry_id;
    int64_t *pos, *end, *index, *qry_pos, *qry_end, *qry_len, *mapq, *call_batch;
    uint8_t *rev;
    uint8_t *cigar_text;
    uint64_t *cigar_off;
} pav_bed_cols;
int pav_bed_open(const char *path, int with_cigar, pav_bed **out);      /* message via pav_last_error(NULL)                  */
void pav_bed_close(pav_bed *bed);
int pav_bed_info(const pav_bed *bed, pav_bed_info_t *info);
int pav_bed_fetch(const pav_bed *bed, const pav_bed_cols *cols);
const char *pav_bed_name(const pav_bed *bed, int which /* 0 #CHROM, 1 QRY_ID */, uint32_t id);
/* pav_cigar_load straight from a parsed table: the rows with CALL_BATCH == call_batch (all rows when negative), names mapped
 * to the stores through pav_seq_set_names.  index_out (n_rows of the table is an upper bound) receives their INDEX. */
int pav_cigar_load_bed(pav_ctx *ctx, const pav_bed *bed, int64_t call_batch, uint32_t *n_rows, int64_t *index_out);

/* ---- alignment trimming ------------------------------------------------ *
 * Replaces pavlib.align.trim_alignments and its helpers trim_alignment_record, find_cut_sites, trace_cigar_to_zero
 * (pavlib/align/trim.py:11-917; rules align_trim_tig / align_trim_tigref, rules/align.snakefile:54-97).  Every CIGAR is
 * tokenised once on the device; a record's CIGAR is then a window into that operation array plus the clipping trimming
 * adds, so a pair of overlapping records costs the operations inside the overlap.  The host mirror keeps the DataFrame,
 * does the reference's sorts and calls one pass per mode with the iteration order. */
typedef struct {            /* one alignment row: what the trimming loops read and write                            */
    uint32_t chrom;         /* #CHROM, compared for equality only                                                  */
    uint32_t qry_id;        /* QRY_ID, compared for equality only                                                  */
    int64_t pos, end, qry_pos, qry_end;
    int64_t index;          /* INDEX; set to -1 when the record is dropped (contained / shorter than the minimum)  */
    int32_t rev;            /* REV                                                                                 */
    int32_t modified;       /* out: the CIGAR string changed                                                       */
    int64_t trim_ref_l, trim_ref_r, trim_qry_l, trim_qry_r;    /* TRIM_REF_L ... (rules/align.snakefile:166-169)    */
} pav_trim_row;

enum { PAV_TRIM_QUERY = 0, PAV_TRIM_SUBJECT = 1 };             /* match_coord 'query' / 'subject'                  */
enum {                                                         /* pav_trim_err.kind: the RuntimeError of ...       */
    PAV_TRIM_ERR_NONE = 0,
    PAV_TRIM_ERR_NEGATIVE = 1,       /* 'Cannot trim to negative distance'        trim.py:428-434, 445-451        */
    PAV_TRIM_ERR_ORDER = 2,          /* 'Contigs are incorrectly ordered in subject space'  trim.py:436-441       */
    PAV_TRIM_ERR_ILLEGAL_OP = 3,     /* 'Illegal operation in contig alignment while trimming'  trim.py:880-883   */
    PAV_TRIM_ERR_NO_CUT = 4          /* 'Program bug: Found no cut-sites'          trim.py:465-470                 */
};
typedef struct {
    int32_t kind;
    uint32_t row_l, row_r;  /* record_l / record_r of the failing trim_alignment_record call (loaded row numbers)   */
    uint32_t op_index;      /* kind 3: operation index in trimming orientation ('CIGAR operation #')               */
    uint32_t op_char;       /* kind 3                                                                              */
    int32_t side;           /* kind 3: 0 = the operation is in record_l, 1 = record_r                              */
    int64_t diff_bp;        /* kind 1                                                                              */
    uint64_t op_len;        /* kind 3                                                                              */
} pav_trim_err;

enum {                      /* pav_trim_count.err_kind: the RuntimeError count_cigar raises (align.py:560-664)       */
    PAV_TRIM_CHECK_OK = 0, PAV_TRIM_CHECK_DUP_S_L = 1, PAV_TRIM_CHECK_DUP_H_L = 2, PAV_TRIM_CHECK_S_BEFORE_H_L = 3,
    PAV_TRIM_CHECK_CLIP_INSIDE = 4, PAV_TRIM_CHECK_DUP_S_R = 5, PAV_TRIM_CHECK_H_BEFORE_S_R = 6, PAV_TRIM_CHECK_DUP_H_R = 7,
    PAV_TRIM_CHECK_M = 8, PAV_TRIM_CHECK_BAD_OP = 9
};
typedef struct {            /* count_cigar of a record's current CIGAR (input of check_record, align.py:364-509)      */
    int64_t ref_bp, tig_bp, clip_h_l, clip_s_l, clip_h_r, clip_s_r;
    int32_t err_kind;
    uint32_t err_op;        /* operation index of the failure                                                      */
    uint64_t err_len;
    uint32_t err_char;
    uint32_t pad;
} pav_trim_count;

/* Upload the table and tokenise the CIGAR strings (PAV_E_CIGAR + pav_cigar_error on a malformed string). */
int pav_trim_load(pav_ctx *ctx, uint32_t n, const pav_trim_row *rows, const uint8_t *cigar_text, const uint64_t *cigar_off);
/* One pass of the pair loop over the rows listed in `order` (loaded row numbers in the reference's iteration order:
 * QRY_ID / QRY_LEN descending for PAV_TRIM_QUERY, trim.py:64-66; #CHROM / END - POS descending for PAV_TRIM_SUBJECT,
 * trim.py:267-274).  PAV_E_TRIM + pav_trim_error when the reference would have raised; the loaded table is undefined from then
 * on (the reference returns none either): pav_trim_pass / pav_trim_pair / pav_trim_fetch with counts or cigar_bytes answer
 * PAV_E_STATE until the next pav_trim_load; pav_trim_fetch of the rows alone still works - the two rows of the error record
 * stand as the failing pair met them (the coordinates the reference's message prints), every other row is unspecified.
 * (A failing pav_trim_pair leaves both rows as they were.) */
int pav_trim_pass(pav_ctx *ctx, uint32_t n_order, const uint32_t *order, int mode, int64_t min_trim_tig_len, int match_tig);
/* trim_alignment_record (trim.py:357-599) on two loaded rows: record_l = row_l, record_r = row_r, rev_l / rev_r as in
 * the reference (trim that record from its downstream end).  Both rows are replaced by their trimmed versions. */
int pav_trim_pair(pav_ctx *ctx, uint32_t row_l, uint32_t row_r, int mode, int rev_l, int rev_r);
int pav_trim_error(const pav_ctx *ctx, pav_trim_err *err);
/* Current state of every loaded row, count_cigar of its CIGAR, and the total size of the CIGAR strings of the rows whose
 * `modified` flag is set (any pointer may be NULL: that part is skipped); then the strings themselves: row i =
 * text[off[i] .. off[i + 1]), empty for unmodified rows (their CIGAR is the input string). */
int pav_trim_fetch(pav_ctx *ctx, pav_trim_row *rows, pav_trim_count *counts, uint64_t *cigar_bytes);
int pav_trim_fetch_cigar(pav_ctx *ctx, uint8_t *text, uint64_t *off);

/* ---- inversion-signature flagging --------------------------------------- *
 * Replaces the run: bodies of rules call_inv_cluster (rules/call_inv.snakefile:603-692),
 * call_inv_flag_insdel_cluster (:480-599) and call_inv_merge_flagged_loci (:321-474).  The per-variant sweeps (cluster
 * boundaries over millions of SNV / indel rows, the INS-against-DEL interval queries) run on the device; the merges of
 * the few thousand resulting regions run on the host inside the library.  #CHROM is passed as a rank: the position of
 * the chromosome name in Python str order, so that (chrom, POS) order here is sort_values(['#CHROM', 'POS']) there.
 * Result arrays are owned by the library and stay valid until the next pav_flag_* / pav_cigar_flag call on the context. */
typedef struct {            /* one row of a flag table: cluster_{snv,indel} (count = COUNT) or insdel_{sv,indel} (0)  */
    uint32_t chrom;
    uint32_t pad;
    int64_t pos, end;
    int64_t count;
} pav_flag_rgn;

enum { PAV_FLAG_MATCH_SV = 1, PAV_FLAG_MATCH_INDEL = 2, PAV_FLAG_CLUSTER_INDEL = 4, PAV_FLAG_CLUSTER_SNV = 8 };   /* TYPE */
enum { PAV_SIG_SVINDEL = 0, PAV_SIG_SV = 1, PAV_SIG_SINGLE_CLUSTER = 2, PAV_SIG_NONE = 3 };   /* inv_sig_filter (:337-355) */

typedef struct {            /* one row of flagged_regions_{hap}.bed.gz (:409-420)                                     */
    uint32_t chrom;
    uint32_t type_mask;     /* TYPE: PAV_FLAG_* bits                                                                  */
    int64_t pos, end;       /* SVLEN = end - pos (may be <= 0: END is the END of the last row merged, :397)          */
    int64_t count_indel, count_snv;
    int32_t try_inv;        /* TRY_INV (_call_inv_accept_flagged_region, :56-79)                                      */
    int32_t batch;          /* BATCH: round-robin over accepted rows, -1 otherwise (:458-466)                         */
} pav_flag_locus;

typedef struct {            /* config keys with the reference's defaults                                              */
    int64_t cluster_win;            /* inv_sig_cluster_win 200 (:609); also the minimum span: the rule reads          */
                                    /* params.cluster_win for cluster_win_min (:619), inv_sig_cluster_win_min is dead */
    int64_t cluster_min_snv;        /* inv_sig_cluster_snv_min 20 (:611)                                              */
    int64_t cluster_min_indel;      /* inv_sig_cluster_indel_min 10 (:612)                                            */
    int64_t insdel_flank_cluster;   /* inv_sig_insdel_cluster_flank 2 (:486)                                          */
    int64_t insdel_flank_merge;     /* inv_sig_insdel_merge_flank 2000 (:487)                                         */
    int64_t insdel_min_svlen;       /* inv_sig_cluster_svlen_min 4 (:488); the SV table always uses 50 (:497)         */
    int64_t merge_flank;            /* inv_sig_merge_flank 500 (:332)                                                 */
    int32_t batch_count;            /* inv_sig_batch_count 60 (:81, :333)                                             */
    int32_t sig_filter;             /* inv_sig_filter: PAV_SIG_*                                                      */
} pav_flag_params;
void pav_flag_params_default(pav_flag_params *p);

/* Cluster sweep of rule call_inv_cluster (:646-684) over n rows given in the rule's iteration order (FILTER == PASS
 * rows, sort_values(['#CHROM','POS']) on the original POS): midpoint = (END + POS) // 2; a row joins the open cluster
 * when chrom matches and midpoint < previous midpoint + win; clusters with count >= min_count and
 * last midpoint - first midpoint >= win_min are reported in order. */
int pav_flag_cluster(pav_ctx *ctx, uint64_t n, const uint32_t *chrom, const int64_t *pos, const int64_t *end, int64_t win,
                     int64_t win_min, int64_t min_count, const pav_flag_rgn **out, uint64_t *n_out);
/* Matched INS / DEL of rule call_inv_flag_insdel_cluster (:517-596): for every INS the DELs of its chromosome that
 * overlap [POS - SVLEN * flank_cluster, POS + SVLEN * flank_cluster) give one (min DEL POS, max DEL END) interval; the
 * intervals are sorted and merged with flank_merge exactly as the rule does (the interval still open at the end of the
 * loop is not written, :575-583).  Rows in any order. */
int pav_flag_insdel(pav_ctx *ctx, uint64_t n_ins, const uint32_t *ins_chrom, const int64_t *ins_pos, const int64_t *ins_svlen,
                    uint64_t n_del, const uint32_t *del_chrom, const int64_t *del_pos, const int64_t *del_end,
                    int64_t flank_cluster, int64_t flank_merge, const pav_flag_rgn **out, uint64_t *n_out);
/* Rule call_inv_merge_flagged_loci (:357-466) over the four flag tables, in the rule's concat order: insdel_sv,
 * insdel_indel, cluster_indel, cluster_snv.  Host only. */
int pav_flag_merge_loci(pav_ctx *ctx, const pav_flag_rgn *const tables[4], const uint64_t n[4], int64_t flank,
                        int32_t batch_count, int32_t sig_filter, const pav_flag_locus **out, uint64_t *n_out);

/* All three rules at once, straight from the records of the last pav_cigar_call (no tables in between): FILTER from the
 * trim table as in pav_table_opts, chromosome ranks from pav_seq_set_names.  tables[] order as pav_flag_merge_loci. */
typedef struct {
    const pav_flag_rgn *tables[4];
    uint64_t n[4];
    const pav_flag_locus *loci;
    uint64_t n_loci;
    uint64_t n_snv_pass, n_indel_pass;      /* rows with FILTER == PASS                                              */
} pav_flag_result;
int pav_cigar_flag(pav_ctx *ctx, const int64_t *trim_pos, const int64_t *trim_end, const pav_flag_params *params,
                   pav_flag_result *result);

/* ---- profiling ---------------------------------------------------------------------------------------- *
 * HIP-event timing of every kernel the library launches on its stream (bench.py's roofline leg).         */
int pav_prof_enable(pav_ctx *ctx, int on);
int pav_prof_reset(pav_ctx *ctx);
int pav_prof_count(pav_ctx *ctx);                                     /* number of distinct kernels seen    */
int pav_prof_get(pav_ctx *ctx, int i, char *name, int name_len, uint64_t *launches, double *total_ms);

#ifdef __cplusplus
}
#endif
#endif /* PAV_AMD_H */
