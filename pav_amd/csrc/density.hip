// K-mer state + density scan on gfx950 (inversion caller): batched over independent (reference region, contig
// region) jobs.  Replaces the `scripts/density.py` subprocess of pavlib.inv.scan_for_inv:
//   pavlib/seq.py:305-325            ref_kmers                      -> k_ref_insert   (hash set with counts in HBM)
//   scripts/density.py:510-539       low-complexity gate, -r        -> k_ref_insert / host gate
//   scripts/density.py:165-203       STATE_MER, informative subset  -> k_tig_state, k_compact_*
//   scripts/density.py:69-115,206-255  gaussian_kde at sampled sites -> k_kde_eval (mode 0)
//   scripts/density.py:257-323       change test, np.interp / fill  -> k_windows, k_kde_eval (mode 1)
//   scripts/density.py:329-338       spike rule, arg-max STATE      -> k_finalize
//   pavlib/density.py:330-361        rl_encoder                     -> k_heads + host assembly
//   pavlib/inv.py:457-561            annotate_inv_dup_mers          -> k_canon_insert, k_annotate
// Integer work is exact.  The KDE is FP64 vector work (no MFMA: exp of a difference is not a contraction); every
// evaluation point accumulates its data points in ascending order like scipy's gaussian_kernel_estimate, so the
// only difference from the reference is the device exp() (<= 1 ulp per term).  Built with -ffp-contract=off.
#include "common.h"
#include "lift_dev.h"

#ifndef PAV_LDS_SLOTS                 // slots / threads of a k_kmer_lds workgroup (tuning builds override them)
#define PAV_LDS_SLOTS 4096
#define PAV_LDS_THREADS 512
#endif

#include <algorithm>
#include <chrono>
#include <cmath>

namespace pav {

constexpr int DTILE = 2048;               // positions per tile (256 lanes x 8); job arenas are tile aligned
constexpr uint64_t EMPTY_KEY = ~0ull;

struct JobDev {
    uint64_t ref_abs;                     // arena position of the first base of region_ref
    uint64_t tig_abs;                     // arena position of the first base of region_tig
    uint32_t ref_len, tig_len;
    uint32_t ref_rc, srs;
    uint64_t ht_off;                      // first slot of this job's hash table
    uint32_t ht_mask;                     // capacity - 1 (power of two)
    uint32_t first_tile;                  // first tile of the job in the contig-position arena
    uint64_t rpos_off;                    // job start in the reference-position arena
    uint64_t tpos_off;                    // job start in the contig-position / row arena
    uint32_t n_parts;                     // > 0: LDS-resident k-mer set split into n_parts partitions; 0: hash table in HBM
    uint32_t bucket_off;                  // first of the job's 2 x n_parts list counters (reference, contig)
    uint64_t list_off_r, list_off_t;      // first entry of the job's n_parts reference lists / of its n_parts contig lists
    uint32_t cap_r, cap_t;                // entries per list
};

struct JobStat {                          // written by kernels, zeroed per batch
    uint32_t n_ref_valid, max_count;
    uint32_t st_count[3];                 // STATE_MER counts before the min-state-count rule
    uint32_t n_rows;
    uint32_t m[3], fill_n;
    uint32_t lds_flags, ones_count;       // k_kmer_lds: LDS_EXCEED (a count passed the limit), LDS_OVERFLOW (partition table full);
                                          // HBM tables, k = 32: occurrences of the one k-mer whose 64 bits are the tables' EMPTY word
                                          // (thirty-two T as the reference spells it: poly-A / poly-T tracts are real) - kept here
    uint32_t inv_first, last1;            // k_state_combine: 0xFFFFFFFF - first / 1 + last contig position with a FWD k-mer (scan-only batches)
    uint32_t n_near, n_reeval, n_unres, n_spike;   // near-tie guard (include/pav_amd.h)
    unsigned long long s1[3], s2[3];      // sum / sum of squares of the row numbers of each state
    unsigned long long max_key;           // packed (first position << 32 | slot) of the max-count k-mer (failure path)
};

struct JobKde {                           // host -> device after the first readback
    uint32_t finalised, n, n_samp, srs;
    uint32_t m[3], use_runs;              // use_runs: closed-form run sums (PAV_KDE_RUNS) for this job
    uint32_t samp_off, ps_mask;           // first entry of the job in the compact arrays of sampled sites (ks / ss); states
                                          // whose density is summed term by term and therefore need the scaled positions
    uint32_t all_direct, pad;             // every present state is summed term by term in ascending order: scipy's order
    uint32_t run_off[3], n_run[3];        // per-state slices of the run arena
    uint32_t heads_off, n_heads;          // device-planned batches: the job's slice of the run arena (upper bound), its runs in all
    double inv_h[3], norm[3], w[3], cnt[3], h[3];
};

// A STATE_MER run head as k_compact_scatter leaves it in a tile's slots (device-planned batches): row (26 bits) | state << 26.
constexpr uint32_t HEAD_ROW_MASK = (1u << 26) - 1u;
constexpr uint32_t HEADS_PER_TILE = 32;   // slots per compaction tile; a tile with more changes sends the batch to the host-planned path

struct RunDev { uint32_t a, b; };         // rows a..b (inclusive) are consecutive data points of one state

constexpr double KDE_RUNS_MIN_H = 32.0;   // Euler-Maclaurin remainder < 1e-15 of the peak for h >= 32
constexpr uint32_t KDE_RUNS_MAX = 1u << 16;   // jobs with more STATE_MER runs use the direct kernel

struct EvalTile { uint32_t job, first, count, mode; };   // mode 0: sampled sites, 1: fill list

struct HeadEvent { uint32_t job, row; int32_t state; uint32_t index, prev_index; uint32_t pad; };

struct DensityState {
    DevBuf jobs, stat, kde, tile_job_r, tile_job_t, keys, cnt, keys_x, cnt_x, lists, bcount, items;
    DevBuf st_tmp, tile_sum, tile_pre, index, state_mer, state, kmer, kern[3], list[3], fill_list;
    DevBuf tiles, events, ev_count, scratch, run_arena, win_fill, ks[3], ss;
    DevBuf guard, guard_entries, samp_flag, row_flag, ftiles; // near-tie guard; evaluation tiles of the fill list
    DevBuf tile_heads, tile_head_cnt, heads, plan_flags, pow_tab, fin_blocks;   // device-planned batches (k_plan); fin_blocks: k_finalize_blocks' list
    // The small per-batch buffers are slices of two arenas: what the host sends (jobs, tile -> job maps, partition items, KDE
    // descriptors, evaluation tiles) travels in ONE copy out of a pinned staging block; what the kernels count into (event
    // count, guard, plan flags, statistics, event list, bucket counts, guard flags) is cleared by two fills and its front
    // comes back in ONE copy.  (Separate pageable copies and fills cost ~10 us each with the stream idle in between: a dozen per
    // scan round.)
    DevBuf in_arena, zero_arena;
    // The columns of the density tables (KERN x 3, KMER, INDEX, STATE_MER, STATE, and FLANK / MATCH of the calls) are slices of one
    // block in the order of a round's host block: a round whose calls are most of the batch hands the block to its stage as it is.
    DevBuf table_block, flank, match;
    uint64_t block_rows = 0;
    void *pin_in = nullptr; size_t pin_in_cap = 0;
    void *pinned_in(size_t bytes) {
        if (bytes <= pin_in_cap) return pin_in;
        if (pin_in) (void)hipHostFree(pin_in);
        pin_in = nullptr; pin_in_cap = 0;
        if (hipHostMalloc(&pin_in, bytes + bytes / 2 + 4096, hipHostMallocDefault) != hipSuccess) { pin_in = nullptr; return nullptr; }
        pin_in_cap = bytes + bytes / 2 + 4096;
        return pin_in;
    }
    std::vector<double> h_pow;                                // pow(n, -1/5), n = 0 .. size - 1 (libm, computed once and extended)
    uint64_t n_fast = 0, n_fallback = 0;                      // batches planned on the device / sent back to the host-planned path
    std::vector<JobDev> h_jobs;
    std::vector<JobKde> h_kde;
    void *pin = nullptr; size_t pin_cap = 0;                  // pinned host scratch for the small readbacks (pageable targets are
    void *pinned(size_t bytes) {                              // staged by the runtime and cost tens of microseconds each)
        if (bytes <= pin_cap) return pin;
        if (pin) (void)hipHostFree(pin);
        pin = nullptr; pin_cap = 0;
        if (hipHostMalloc(&pin, bytes + bytes / 2 + 4096, hipHostMallocDefault) != hipSuccess) { pin = nullptr; return nullptr; }
        pin_cap = bytes + bytes / 2 + 4096;
        return pin;
    }
    hipEvent_t gathered = nullptr;        // the packed call tables of a round are complete (main stream)
    std::vector<pav_den_result> results;
    std::vector<std::vector<pav_run>> runs;
    std::vector<uint8_t> no_table;                            // per job: its table was not built (scan-only batch, FWD k-mers only)
    pav_den_params params{};
    uint32_t n_jobs = 0;
    uint64_t arena_t = 0;
    bool valid = false;
    void release() {
        DevBuf *all[] = {&jobs, &stat, &kde, &tile_job_r, &tile_job_t, &keys, &cnt, &keys_x, &cnt_x, &lists, &bcount, &items, &st_tmp, &tile_sum, &tile_pre,
                         &index, &state_mer, &state, &kmer, &kern[0], &kern[1], &kern[2], &list[0], &list[1], &list[2],
                         &fill_list, &tiles, &events, &ev_count, &scratch, &run_arena, &win_fill, &ks[0], &ks[1], &ks[2], &ss,
                         &guard, &guard_entries, &samp_flag, &row_flag, &ftiles, &tile_heads, &tile_head_cnt, &heads, &plan_flags, &pow_tab, &fin_blocks,
                         &in_arena, &zero_arena, &table_block, &flank, &match};
        for (DevBuf *b : all) b->release();
        if (pin) { (void)hipHostFree(pin); pin = nullptr; pin_cap = 0; }
        if (pin_in) { (void)hipHostFree(pin_in); pin_in = nullptr; pin_in_cap = 0; }
        if (gathered) { (void)hipEventDestroy(gathered); gathered = nullptr; }
    }
};

// ---- k-mer access ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}
__device__ __forceinline__ uint64_t kmer_mask(int k) { return k >= 32 ? ~0ull : (1ull << (2 * k)) - 1ull; }   // k <= 32

// Reverse the k 2-bit groups in the low 2k bits: window order (first base lowest) <-> kanapy order (first base highest)
__device__ __forceinline__ uint64_t rev_groups(uint64_t x, int k) {
    uint64_t r = __brevll(x) >> (64 - 2 * k);
    return ((r & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((r & 0x5555555555555555ull) << 1);
}

// K-mer window starting at arena position a: x = bases in window order (base j at bits 2j); returns false when any of
// the k bases is non-ACGT (kanapy's stream skips those windows).
__device__ __forceinline__ bool kmer_window(const uint32_t *__restrict__ two, const uint32_t *__restrict__ mask, uint64_t a,
                                            int k, uint64_t &x) {
    const uint64_t *two64 = reinterpret_cast<const uint64_t *>(two);
    const uint64_t *mask64 = reinterpret_cast<const uint64_t *>(mask);
    const uint64_t w = a >> 5;
    const int b = (int)(a & 31) * 2;
    uint64_t v = two64[w] >> b;
    if (b) v |= two64[w + 1] << (64 - b);
    x = v & kmer_mask(k);
    const uint64_t w2 = a >> 6;
    const int b2 = (int)(a & 63);
    uint64_t y = mask64[w2] >> b2;
    if (b2) y |= mask64[w2 + 1] << (64 - b2);
    return (y & ((1ull << k) - 1ull)) == 0;
}
// In kanapy order: kmer = rev_groups(x), reverse complement = x ^ mask (see DESIGN.md "k-mer encoding").
// The same window in two steps, for loops that want the loads of several windows in flight: kmer_words() only loads (no branch,
// no use of the data - a window fetched inside an `if` is waited for before the next one is issued), kmer_from_words() shifts.
struct KmerWords { uint64_t v0, v1; };
__device__ __forceinline__ KmerWords kmer_words(const uint32_t *__restrict__ two, uint64_t a) {
    const uint64_t *two64 = reinterpret_cast<const uint64_t *>(two);
    const uint64_t w = a >> 5;
    return KmerWords{two64[w], two64[w + 1]};                       // the word after the last base of a record is padding
}
// all k bases from a are ACGT (the rare second step of a window whose summary bytes mark a non-ACGT base nearby)
__device__ __forceinline__ bool kmer_clean(const uint32_t *__restrict__ mask, uint64_t a, int k) {
    const uint64_t *mask64 = reinterpret_cast<const uint64_t *>(mask);
    const uint64_t w2 = a >> 6;
    const int b2 = (int)(a & 63);
    uint64_t y = mask64[w2] >> b2;
    if (b2) y |= mask64[w2 + 1] << (64 - b2);
    return (y & ((1ull << k) - 1ull)) == 0;
}
__device__ __forceinline__ uint64_t kmer_from_words(const KmerWords &kw, uint64_t a, int k) {
    const int b = (int)(a & 31) * 2;
    return (b ? (kw.v0 >> b | kw.v1 << (64 - b)) : kw.v0) & kmer_mask(k);
}


__device__ __forceinline__ bool table_has(const unsigned long long *__restrict__ keys, uint64_t off, uint32_t hmask,
                                          uint64_t key) {
    if (key == EMPTY_KEY) return false;          // (k = 32: the word of a free slot is never IN a table - k_ref_insert keeps that k-mer's
                                                 //  count aside, and it is no canonical k-mer, so the flank sets never hold it)
    uint32_t s = (uint32_t)mix64(key) & hmask;
    while (true) {
        const uint64_t cur = keys[off + s];
        if (cur == key) return true;
        if (cur == EMPTY_KEY) return false;
        s = (s + 1) & hmask;
    }
}

__device__ __forceinline__ uint32_t table_insert(unsigned long long *__restrict__ keys, uint64_t off, uint32_t hmask,
                                                 uint64_t key) {
    uint32_t s = (uint32_t)mix64(key) & hmask;
    while (true) {
        const unsigned long long old = atomicCAS(&keys[off + s], (unsigned long long)EMPTY_KEY, (unsigned long long)key);
        if (old == EMPTY_KEY || old == key) return s;
        s = (s + 1) & hmask;
    }
}

// Insert; `dup` tells whether the key was already there.
__device__ __forceinline__ uint32_t table_insert_dup(unsigned long long *__restrict__ keys, uint64_t off, uint32_t hmask,
                                                     uint64_t key, bool &dup) {
    uint32_t s = (uint32_t)mix64(key) & hmask;
    while (true) {
        const unsigned long long old = atomicCAS(&keys[off + s], (unsigned long long)EMPTY_KEY, (unsigned long long)key);
        if (old == EMPTY_KEY) { dup = false; return s; }
        if (old == key) { dup = true; return s; }
        s = (s + 1) & hmask;
    }
}

__device__ __forceinline__ uint32_t table_slot(const unsigned long long *__restrict__ keys, uint64_t off, uint32_t hmask, uint64_t key) {
    uint32_t s = (uint32_t)mix64(key) & hmask;
    while (keys[off + s] != key) s = (s + 1) & hmask;                  // the key is known to be present
    return s;
}

// ---- reference k-mers -> hash set with counts (pavlib/seq.py:305-325; -r: scripts/density.py:538-539) ----------
// One atomic per k-mer in the common case: cnt[] holds (occurrences - 1) and is only touched by repeats.
__global__ __launch_bounds__(256) void k_ref_insert(const JobDev *__restrict__ jobs, const uint32_t *__restrict__ tile_job,
                                                    SeqView R, int k, uint32_t limit, unsigned long long *__restrict__ keys,
                                                    uint32_t *__restrict__ cnt, JobStat *__restrict__ stat, uint32_t block0,
                                                    int force) {
    const uint64_t ap = ((uint64_t)blockIdx.x + block0) * 256 + threadIdx.x;
    const uint32_t j = tile_job[ap / DTILE];
    const JobDev jd = jobs[j];
    if (jd.n_parts && !force) return;                                  // this job's set lives in LDS (k_kmer_lds)
    const uint64_t i = ap - jd.rpos_off;
    bool valid = false;
    if (i + (uint64_t)k <= jd.ref_len) {
        uint64_t x;
        valid = kmer_window(R.two, R.mask, jd.ref_abs + i, k, x);
        if (valid) {
            const uint64_t key = jd.ref_rc ? (x ^ kmer_mask(k)) : rev_groups(x, k);   // set of rc(kmer) when -r true
            uint32_t c;
            if (key == EMPTY_KEY) c = atomicAdd(&stat[j].ones_count, 1u) + 1u;        // (k = 32 only: the word that marks a free slot)
            else {
                bool dup;
                const uint32_t s = table_insert_dup(keys, jd.ht_off, jd.ht_mask, key, dup);
                c = dup ? atomicAdd(&cnt[jd.ht_off + s], 1u) + 2u : 1u;
            }
            if (c > limit) atomicMax(&stat[j].max_count, c);
        }
    }
    const unsigned long long b = __ballot(valid);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(&stat[j].n_ref_valid, (uint32_t)__popcll(b));
}

// Failure path only: first-inserted k-mer among those with the maximum count (message of scripts/density.py:519-526):
// the smallest reference position whose k-mer reached the maximum.
__global__ void k_max_kmer(const JobDev *__restrict__ jobs, uint32_t j, SeqView R, int k, const unsigned long long *__restrict__ keys,
                           const uint32_t *__restrict__ cnt, JobStat *__restrict__ stat) {
    const JobDev jd = jobs[j];
    const uint32_t mx = stat[j].max_count;
    for (uint64_t i = blockIdx.x * blockDim.x + threadIdx.x; i + (uint64_t)k <= jd.ref_len; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t x;
        if (!kmer_window(R.two, R.mask, jd.ref_abs + i, k, x)) continue;
        const uint64_t key = jd.ref_rc ? (x ^ kmer_mask(k)) : rev_groups(x, k);
        if (key == EMPTY_KEY) {                                          // (k = 32: counted in the job's statistics, slot 0xFFFFFFFF stands for it)
            if (stat[j].ones_count == mx) atomicMin(&stat[j].max_key, ((unsigned long long)i << 32) | 0xFFFFFFFFull);
            continue;
        }
        const uint32_t s = table_slot(keys, jd.ht_off, jd.ht_mask, key);
        if (cnt[jd.ht_off + s] + 1u == mx) atomicMin(&stat[j].max_key, ((unsigned long long)i << 32) | s);
    }
}

// ---- contig k-mers -> STATE_MER (scripts/density.py:165-175) -------------------------------------------------
__global__ __launch_bounds__(256) void k_tig_state(const JobDev *__restrict__ jobs, const uint32_t *__restrict__ tile_job,
                                                   SeqView T, int k, const unsigned long long *__restrict__ keys,
                                                   int8_t *__restrict__ st_tmp, JobStat *__restrict__ stat, uint32_t block0) {
    const uint64_t ap = ((uint64_t)blockIdx.x + block0) * 256 + threadIdx.x;
    const uint32_t j = tile_job[ap / DTILE];
    const JobDev jd = jobs[j];
    if (jd.n_parts) return;                                            // k_kmer_lds + k_state_combine
    const uint64_t i = ap - jd.tpos_off;
    int st = -1;
    if (i + (uint64_t)k <= jd.tig_len) {
        uint64_t x;
        if (kmer_window(T.two, T.mask, jd.tig_abs + i, k, x)) {
            const uint64_t kf = rev_groups(x, k), kr = x ^ kmer_mask(k);
            const bool in_f = kf == EMPTY_KEY ? stat[j].ones_count != 0 : table_has(keys, jd.ht_off, jd.ht_mask, kf);
            const bool in_r = kr == EMPTY_KEY ? stat[j].ones_count != 0 : table_has(keys, jd.ht_off, jd.ht_mask, kr);
            st = in_f ? (in_r ? 1 : 0) : (in_r ? 2 : -1);            // KMER_ORIENTATION_STATE, density.py:38-43
        }
    }
    st_tmp[ap] = (int8_t)st;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const unsigned long long b = __ballot(st == s);
        if ((threadIdx.x & 63) == 0 && b) atomicAdd(&stat[j].st_count[s], (uint32_t)__popcll(b));
    }
}

// ---- LDS-resident k-mer sets ------------------------------------------------------------------------------------
// The reference k-mers of a region are split by a hash into n_parts partitions of ~1.8 k k-mers; one workgroup owns one
// (job, partition): it builds that partition's set in a 32 KiB LDS table and answers the membership questions of the contig
// k-mers that hash to it - no table atomic ever leaves the CU.  (Table size: with the GPU to itself a 128 KiB table - one
// 1024-lane workgroup per CU - and a 32 KiB one run the same, 0.146 / 0.118 ms per launch; beside the kernels of three other
// resident haplotypes the big workgroup waits for a whole free CU, and the step went from 3.85 to 3.55 ms with the small one.)
//   k_bucket_ref / k_bucket_tig  one workgroup per 2048-position tile: every k-mer is hashed once and its position is
//                                appended to the list of its partition (ranks from an LDS histogram, one global atomic
//                                per (tile, partition) to reserve the slots)
//   k_kmer_lds                   reads its two lists densely: inserts, then probes and writes STATE_MER of the positions whose k-mer is in a set
//   k_state_combine              per-tile and per-job counts of the states
// The sets are keyed by the CANONICAL k-mer - the smaller of a k-mer and its reverse complement - with one occurrence count
// per orientation: "the contig k-mer is in the reference set" and "its reverse complement is" are then ONE question (same
// canonical key; same / other orientation), so a contig k-mer goes to one list and is looked up once.  (Round 2 kept the
// oriented reference k-mers and asked twice: two list entries and two probes per contig position.)  For even k a k-mer can
// be its own reverse complement: both answers are then the same.
// Counts (scripts/density.py:516-527) are bytes holding occurrences, per orientation; a count that passes the limit only raises a flag
// and the exact maximum and its k-mer come from the HBM-table kernels on that failure path.
constexpr int LDS_SLOTS = PAV_LDS_SLOTS;
constexpr int LDS_THREADS = PAV_LDS_THREADS;
#ifndef PAV_LDS_STEP
#define PAV_LDS_STEP 2048
#endif
constexpr int KU = PAV_LDS_STEP / LDS_THREADS;        // list entries a lane of k_kmer_lds has in flight: a step covers 2048 entries, about a partition
constexpr uint32_t LDS_FILL = LDS_SLOTS * 7 / 16;   // k-mers per partition aimed at (load factor 0.44)
constexpr uint32_t LDS_MAX_PARTS = 1024;             // histogram size of the bucket kernels (regions up to 1.8 Mbp; MAX_REGION_SIZE is 1.2 Mbp)
constexpr uint32_t LDS_MAX_LIMIT = 250;              // byte counts: the limit must stay below the wrap
constexpr uint32_t LDS_EXCEED = 1, LDS_OVERFLOW = 2;

struct PartItem { uint32_t job, part; };

__device__ __forceinline__ uint32_t khash(uint64_t key) {
    uint32_t h = (uint32_t)key * 0x9E3779B1u ^ (uint32_t)(key >> 32) * 0x85EBCA77u;
    h ^= h >> 15; h *= 0xC2B2AE3Du; h ^= h >> 13;
    return h;
}
__device__ __forceinline__ uint32_t kpart(uint32_t h, uint32_t n_parts) { return ((h >> 16) * n_parts) >> 16; }

// x: the k bases of a window in window order (kmer_from_words).  The k-mer as the reference spells it (kanapy order) is
// rev_groups(x), its reverse complement x ^ mask; the canonical key is the smaller one, `other` says that it is the reverse
// complement (the k-mer is in the orientation opposite to its canonical form).
__device__ __forceinline__ uint64_t canon_key(uint64_t x, int k, bool *other = nullptr, bool *self_rc = nullptr) {
    const uint64_t f = rev_groups(x, k), c = x ^ kmer_mask(k);
    if (other) *other = c < f;
    if (self_rc) *self_rc = c == f;
    return c < f ? c : f;
}

// List capacity per (job, partition): the mean plus 25 % (> 20 sigma at 7 k entries) plus slack for small means.
static uint32_t bucket_cap(uint64_t n_pos, uint32_t n_parts) {
    if (n_parts <= 1) return (uint32_t)std::max<uint64_t>(n_pos, 1);
    return (uint32_t)(n_pos / n_parts + n_pos / n_parts / 4 + 256);
}

// Output stage of the bucket kernels.  hist[p] = entries of this tile for partition p, base[p] = their first slot in the partition's
// list (reserved with one global atomic).  The entries are put in partition order in LDS and leave run by run: consecutive lanes
// store consecutive slots of one list (a lane storing its eight entries to eight unrelated lists kept the waves issue-stalled 40 %
// of their time: every such store instruction is 64 separate lines).
struct BucketStage { uint32_t lstart[LDS_MAX_PARTS + 1]; uint32_t pos[DTILE]; uint16_t pid[DTILE]; uint32_t wsum[4]; };
__device__ __forceinline__ bool bucket_flush(BucketStage &S, uint32_t P, const uint32_t *hist, const uint32_t *base, const uint32_t (&pid)[8],
                                             const uint32_t (&rank)[8], uint32_t pos0, uint32_t *__restrict__ list0, uint32_t cap) {
    // exclusive prefix of hist over the partitions: four per lane
    const uint32_t p0 = threadIdx.x * 4;
    uint32_t h[4], mine = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { h[q] = p0 + q < P ? hist[p0 + q] : 0u; mine += h[q]; }
    uint32_t inc = mine;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(inc, d); if (lane >= d) inc += y; }
    if (lane == 63) S.wsum[wave] = inc;
    __syncthreads();
    uint32_t run = inc - mine, total = 0;
    for (int w = 0; w < 4; ++w) { if (w < wave) run += S.wsum[w]; total += S.wsum[w]; }
#pragma unroll
    for (int q = 0; q < 4; ++q) { if (p0 + q < P) S.lstart[p0 + q] = run; run += h[q]; }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        if (pid[t] == ~0u) continue;
        const uint32_t at = S.lstart[pid[t]] + rank[t];
        S.pos[at] = pos0 + t * 256 + threadIdx.x;
        S.pid[at] = (uint16_t)pid[t];
    }
    __syncthreads();
    bool over = false;
    for (uint32_t r = threadIdx.x; r < total; r += 256) {
        const uint32_t p = S.pid[r], slot = base[p] + (r - S.lstart[p]);
        if (slot < cap) list0[(uint64_t)p * cap + slot] = S.pos[r]; else over = true;
    }
    return over;
}

__global__ __launch_bounds__(256, 6) void k_bucket_ref(const JobDev *__restrict__ jobs, const uint32_t *__restrict__ tile_job,
                                                    SeqView R, int k, uint32_t *__restrict__ lists, uint32_t *__restrict__ bcount,
                                                    JobStat *__restrict__ stat) {
    __shared__ uint32_t hist[LDS_MAX_PARTS], base[LDS_MAX_PARTS];
    const uint32_t j = tile_job[blockIdx.x];
    const JobDev jd = jobs[j];
    const uint32_t P = jd.n_parts;
    if (!P) return;
    for (uint32_t p = threadIdx.x; p < P; p += 256) hist[p] = 0;
    __syncthreads();
    const uint64_t i0 = (uint64_t)blockIdx.x * DTILE - jd.rpos_off;
    uint32_t pid[8], rank[8];
    bool any = false;
    // the eight windows of a lane and the summary bytes of their blocks are fetched before any is used (no load inside a branch);
    // the non-ACGT plane itself only where a summary byte is set
    KmerWords kw[8]; uint64_t at[8]; uint32_t dirt[8]; bool inside[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const uint64_t i = i0 + t * 256 + threadIdx.x;
        inside[t] = i + (uint64_t)k <= jd.ref_len;
        at[t] = jd.ref_abs + (inside[t] ? i : 0);
        kw[t] = kmer_words(R.two, at[t]);
        dirt[t] = (uint32_t)R.dirty[at[t] >> DIRTY_SHIFT] | R.dirty[(at[t] + (uint64_t)k - 1) >> DIRTY_SHIFT];
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const uint64_t x = kmer_from_words(kw[t], at[t], k);
        pid[t] = ~0u;
        if (inside[t] && (!dirt[t] || kmer_clean(R.mask, at[t], k))) {
            pid[t] = kpart(khash(canon_key(x, k)), P);
            rank[t] = atomicAdd(&hist[pid[t]], 1u);
            any = true;
        }
    }
    const int block_any = __syncthreads_or(any);
    for (uint32_t p = threadIdx.x; p < P; p += 256)
        if (hist[p]) base[p] = atomicAdd(&bcount[jd.bucket_off + p], hist[p]);
    __syncthreads();
    __shared__ BucketStage stage;
    const bool over = bucket_flush(stage, P, hist, base, pid, rank, (uint32_t)i0, lists + jd.list_off_r, jd.cap_r);
    if (over) atomicOr(&stat[j].lds_flags, LDS_OVERFLOW);
    if (block_any && threadIdx.x == 0) stat[j].n_ref_valid = 1;       // only "any" matters (scripts/density.py:510-513)
}

__global__ __launch_bounds__(256, 6) void k_bucket_tig(const JobDev *__restrict__ jobs, const uint32_t *__restrict__ tile_job,
                                                    SeqView T, int k, uint32_t *__restrict__ lists, uint32_t *__restrict__ bcount,
                                                    int8_t *__restrict__ st_tmp, JobStat *__restrict__ stat) {
    __shared__ uint32_t hist[LDS_MAX_PARTS], base[LDS_MAX_PARTS];
    const uint32_t j = tile_job[blockIdx.x];
    const JobDev jd = jobs[j];
    const uint32_t P = jd.n_parts;
    if (!P) return;
    for (uint32_t p = threadIdx.x; p < P; p += 256) hist[p] = 0;
    __syncthreads();
    const uint64_t i0 = (uint64_t)blockIdx.x * DTILE - jd.tpos_off;
    uint32_t pf[8], rf[8];
    KmerWords kw[8]; uint64_t at[8]; uint32_t dirt[8]; bool inside[8];          // as in k_bucket_ref
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const uint64_t i = i0 + t * 256 + threadIdx.x;
        inside[t] = i + (uint64_t)k <= jd.tig_len;
        at[t] = jd.tig_abs + (inside[t] ? i : 0);
        kw[t] = kmer_words(T.two, at[t]);
        dirt[t] = (uint32_t)T.dirty[at[t] >> DIRTY_SHIFT] | T.dirty[(at[t] + (uint64_t)k - 1) >> DIRTY_SHIFT];
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const uint64_t x = kmer_from_words(kw[t], at[t], k);
        pf[t] = ~0u;
        if (inside[t] && (!dirt[t] || kmer_clean(T.mask, at[t], k))) {
            pf[t] = kpart(khash(canon_key(x, k)), P);
            rf[t] = atomicAdd(&hist[pf[t]], 1u);
        }
        st_tmp[(uint64_t)blockIdx.x * DTILE + t * 256 + threadIdx.x] = (int8_t)-1;      // no window here, or a k-mer in neither set: k_kmer_lds writes the others
    }
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < P; p += 256)
        if (hist[p]) base[p] = atomicAdd(&bcount[jd.bucket_off + P + p], hist[p]);
    __syncthreads();
    __shared__ BucketStage stage;
    const bool over = bucket_flush(stage, P, hist, base, pf, rf, (uint32_t)i0, lists + jd.list_off_t, jd.cap_t);
    if (over) atomicOr(&stat[j].lds_flags, LDS_OVERFLOW);
}

// Slot layout of k_kmer_lds: bits 0..61 the canonical k-mer (k <= 31), bit 62 / 63: the set holds it in its canonical / in the
// other orientation.  A probe of the answer phase is then ONE LDS read (round 3: key, then the count bytes of the slot).
#ifndef PAV_FIN_R                     // table rows a lane of k_finalize_blocks takes
#define PAV_FIN_R 2
#endif
#ifndef PAV_KMER_V                    // 0: round 5's probes; 1: counts touched by repeats only; 2: ... and pair reads in the answer phase
#define PAV_KMER_V 2
#endif
constexpr uint32_t PROBE_START_MASK = PAV_KMER_V >= 2 ? ~1u : ~0u;   // pair reads: every walk starts at an even slot
constexpr unsigned long long KEY_BITS = (1ull << 62) - 1ull;
constexpr int KEY_O_SHIFT = 62;

#ifdef PAV_KMER_PROF                  // tuning build: cycles of wave 0 of every workgroup per phase of k_kmer_lds (printed by pav_density_release)
__device__ unsigned long long g_kmer_prof[16];
#define KPROF_LAP(i) do { if ((threadIdx.x >> 6) == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); \
                                                         if ((threadIdx.x & 63) == 0) atomicAdd(&g_kmer_prof[i], now_ - kp_clk); kp_clk = now_; } } while (0)
#else
#define KPROF_LAP(i) do { } while (0)
#endif
// ABL / ABL_W: ablations for the tuning build (PAV_TUNING, tools/build_variant.sh); the product kernel is the <0, 0> body
template <int ABL, int ABL_W>
__device__ __forceinline__ void kmer_lds_body(const PartItem *__restrict__ items, const JobDev *__restrict__ jobs,
                                                          SeqView R, SeqView T, int k, uint32_t limit,
                                                          const uint32_t *__restrict__ lists, const uint32_t *__restrict__ bcount,
                                                          int8_t *__restrict__ st_tmp, JobStat *__restrict__ stat) {
    __shared__ __attribute__((aligned(16))) unsigned long long keys[LDS_SLOTS];   // canonical k-mers + orientation bits
    __shared__ uint32_t cnt2[LDS_SLOTS / 2];                           // per slot two bytes: occurrences in the canonical / the other orientation
#ifdef PAV_LDS_AB                     // tuning build: three workgroups a CU as in round 5
    __shared__ uint32_t ab_pad[2];
    if (threadIdx.x == 0) reinterpret_cast<volatile uint32_t *>(ab_pad)[0] = 1;
#endif
    // (40 960 bytes exactly: FOUR workgroups a CU.  A flag word beside the two arrays made it 40 968 and three - found in round 6 from
    //  the code object's notes; the rare flags go to the job's statistics straight from the waves that raise them.)
#ifdef PAV_KMER_PROF
    unsigned long long kp_clk = __builtin_readcyclecounter();
#endif
    const PartItem it = items[blockIdx.x];
    if (it.job == ~0u) return;                                          // padding of the XCD-grouped order
    const JobDev jd = jobs[it.job];
    const uint32_t P = jd.n_parts;
    const uint32_t n_ref = min(bcount[jd.bucket_off + it.part], jd.cap_r), n_tig = min(bcount[jd.bucket_off + P + it.part], jd.cap_t);
    const uint32_t *list_r = lists + jd.list_off_r + (uint64_t)it.part * jd.cap_r;
    const uint32_t *list_t = lists + jd.list_off_t + (uint64_t)it.part * jd.cap_t;
    constexpr uint32_t M = LDS_SLOTS - 1;

    // The kernel is a chain of latencies - list entry -> window of the plane -> LDS table - once for the reference k-mers and once
    // for the contig k-mers, and a partition is about one step of the workgroup in either list (KU entries per lane).  The first
    // step of BOTH lists is therefore put in flight before the table is cleared: its positions, then the reference windows, wait
    // in registers while the table is cleared; the contig windows are fetched behind the inserts.
    uint32_t pos_r[KU], pos_t[KU]; bool ok_r[KU], ok_t[KU];
    KmerWords kw_r[KU], kw_t[KU];
#pragma unroll
    for (int u = 0; u < KU; ++u) {
        const uint32_t e = threadIdx.x + u * LDS_THREADS;
        ok_r[u] = e < n_ref; pos_r[u] = n_ref ? list_r[ok_r[u] ? e : n_ref - 1] : 0u;
        ok_t[u] = e < n_tig; pos_t[u] = n_tig ? list_t[ok_t[u] ? e : n_tig - 1] : 0u;
    }
#pragma unroll
    for (int u = 0; u < KU; ++u) {
        if constexpr (ABL_W) kw_r[u] = KmerWords{(uint64_t)pos_r[u] * 0x9E3779B97F4A7C15ull, (uint64_t)pos_r[u] * 0xC2B2AE3D27D4EB4Full};
        else kw_r[u] = kmer_words(R.two, jd.ref_abs + pos_r[u]);
    }

    KPROF_LAP(0);                                                       // items -> job -> counts; list and window loads issued
    for (int s = threadIdx.x; s < LDS_SLOTS; s += LDS_THREADS) keys[s] = EMPTY_KEY;
    for (int s = threadIdx.x; s < LDS_SLOTS / 2; s += LDS_THREADS) cnt2[s] = 0;
    __syncthreads();
    KPROF_LAP(1);                                                       // table cleared, barrier
#ifdef PAV_KMER_PROF
    { uint64_t acc_ = 0; for (int u = 0; u < KU; ++u) acc_ += kw_r[u].v0 ^ kw_r[u].v1; if (acc_ == 0x123456789ull) st_tmp[0] = 7; }
    KPROF_LAP(2);                                                       // the reference windows have arrived
#endif

    // One entry after the other, each with its own probe loop.  Tuning builds of round 4 (profiles/r04_kmer_ablation.txt) put the
    // first probe of all KU entries in flight, or probed in rounds over the KU entries until the wave's last entry had its slot:
    // both are SLOWER (0.25 - 0.35 ms per pass against 0.20) - with 32 waves per CU the latency of a probe is covered already,
    // and the extra compare-and-swaps of entries that are done cost more than the waiting they remove.
    uint32_t my_flags = 0;
    auto insert_step = [&](const KmerWords (&kw)[KU], const uint32_t (&pos)[KU], const bool (&ok)[KU]) __attribute__((always_inline)) {
        if constexpr (ABL == 3 || ABL == 5) {
            uint64_t acc = 0;
            for (int u = 0; u < KU; ++u) acc += kw[u].v0 ^ kw[u].v1 ^ pos[u];
            if (acc == 0x123456789ull && ok[0]) my_flags |= 4;
            return;
        }
#pragma unroll
        for (int u = 0; u < KU; ++u) if (ok[u]) {
            bool other, self_rc;
            const uint64_t key = canon_key(kmer_from_words(kw[u], jd.ref_abs + pos[u], k), k, &other, &self_rc);
            // the set holds the region's k-mers as they are (-r: reverse-complemented): which orientation of the key that is
            const uint32_t o = self_rc ? 0u : (uint32_t)(other != (jd.ref_rc != 0));
            const unsigned long long want = key | 1ull << (KEY_O_SHIFT + o);
            uint32_t s = khash(key) & M & PROBE_START_MASK;
            int probes = 0;
            while (true) {
                const unsigned long long od = atomicCAS(&keys[s], (unsigned long long)EMPTY_KEY, want);
#if PAV_KMER_V >= 1
                // One LDS atomic per k-mer in the common case: the lane that makes the slot (or sets the orientation bit of a slot that
                // only had the other one) is that orientation's first occurrence and counts nothing; every later one adds 1, so the
                // byte holds occurrences - 1.  Exactly one lane is first: the compare-and-swap / the fetch-or say which.
                if (od == EMPTY_KEY) break;
                if (((od ^ want) & KEY_BITS) == 0) {
                    const unsigned long long fl = want & ~KEY_BITS;
                    bool repeat = (od & fl) != 0;
                    if (!repeat) repeat = (atomicOr(&keys[s], fl) & fl) != 0;
                    if (repeat) {
                        const uint32_t sh = 16 * (s & 1) + 8 * o;
                        const uint32_t prev = (atomicAdd(&cnt2[s >> 1], 1u << sh) >> sh) & 0xFFu;            // repeats so far: this is occurrence prev + 2
                        if (prev + 2 > limit) my_flags |= LDS_EXCEED;                                        // (the limit of scripts/density.py:516-527)
                    }
                    break;
                }
#else
                if (od == EMPTY_KEY || ((od ^ want) & KEY_BITS) == 0) {
                    if (od != EMPTY_KEY && !(od & want & ~KEY_BITS)) atomicOr(&keys[s], want & ~KEY_BITS);   // seen before, in the other orientation only
                    const uint32_t sh = 16 * (s & 1) + 8 * o;
                    const uint32_t prev = (atomicAdd(&cnt2[s >> 1], 1u << sh) >> sh) & 0xFFu;                // occurrences so far (the limit of
                    if (prev + 1 > limit) my_flags |= LDS_EXCEED;                                            // scripts/density.py:516-527)
                    break;
                }
#endif
                s = (s + 1) & M;
                if (++probes >= LDS_SLOTS) { my_flags |= LDS_OVERFLOW; break; }
            }
        }
    };
    insert_step(kw_r, pos_r, ok_r);
    for (uint32_t e0 = threadIdx.x + KU * LDS_THREADS; e0 - threadIdx.x < n_ref; e0 += KU * LDS_THREADS) {   // a partition longer than one step
        uint32_t pos[KU]; bool ok[KU]; KmerWords kw[KU];
#pragma unroll
        for (int u = 0; u < KU; ++u) { const uint32_t e = e0 + u * LDS_THREADS; ok[u] = e < n_ref; pos[u] = list_r[ok[u] ? e : n_ref - 1]; }
#pragma unroll
        for (int u = 0; u < KU; ++u) kw[u] = kmer_words(R.two, jd.ref_abs + pos[u]);
        insert_step(kw, pos, ok);
    }
#pragma unroll
    for (int u = 0; u < KU; ++u) {
        if constexpr (ABL_W) kw_t[u] = KmerWords{(uint64_t)pos_t[u] * 0x9E3779B97F4A7C15ull, (uint64_t)pos_t[u] * 0xC2B2AE3D27D4EB4Full};
        else kw_t[u] = kmer_words(T.two, jd.tig_abs + pos_t[u]);
    }
    KPROF_LAP(3);                                                       // inserts of wave 0
    if (__ballot(my_flags != 0)) {                                      // (rare: a count above the limit, a full table)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) my_flags |= (uint32_t)__shfl_xor((int)my_flags, d);
        if ((threadIdx.x & 63) == 0) atomicOr(&stat[it.job].lds_flags, my_flags);
    }
    __syncthreads();
    KPROF_LAP(4);                                                       // barrier behind the inserts (the slowest wave)
#ifdef PAV_KMER_PROF
    { uint64_t acc_ = 0; for (int u = 0; u < KU; ++u) acc_ += kw_t[u].v0 ^ kw_t[u].v1; if (acc_ == 0x123456789ull) st_tmp[0] = 7; }
    KPROF_LAP(5);                                                       // the contig windows have arrived
#endif

    // contig k-mers: "is it in the set" and "is its reverse complement" are the two orientation bits of its canonical key.
    // STATE_MER (scripts/density.py:38-43,165-175: KMER_ORIENTATION_STATE) goes straight to the tile array - k_bucket_tig has
    // put -1 at every position, a k-mer in neither set leaves it there.
    auto answer_step = [&](const KmerWords (&kw)[KU], const uint32_t (&pos)[KU], const bool (&ok)[KU]) __attribute__((always_inline)) {
        if constexpr (ABL == 2 || ABL == 5) {
            uint64_t acc = 0;
            for (int u = 0; u < KU; ++u) acc += kw[u].v0 ^ kw[u].v1 ^ pos[u];
            if (acc == 0x123456789ull && ok[0]) st_tmp[0] = 1;
            return;
        }
#pragma unroll
        for (int u = 0; u < KU; ++u) if (ok[u]) {
            bool other, self_rc;
            const uint64_t key = canon_key(kmer_from_words(kw[u], jd.tig_abs + pos[u], k), k, &other, &self_rc);
            uint32_t s = khash(key) & M & PROBE_START_MASK;
#if PAV_KMER_V >= 2
            // a probe reads the PAIR of slots at an even index with one 16-byte LDS load (the inserts start at even slots too and walk
            // on slot by slot, so the first empty slot of the walk ends it): half the trips round the divergent loop
            for (int probes = 0; probes < LDS_SLOTS; probes += 2) {
                const ulonglong2 two = *reinterpret_cast<const ulonglong2 *>(&keys[s]);
                const bool hit0 = (two.x & KEY_BITS) == key, hit1 = two.x != EMPTY_KEY && (two.y & KEY_BITS) == key;
                if (hit0 || hit1) {                                         // (EMPTY has all 62 bits set: no k-mer is that key)
                    const unsigned long long cur = hit0 ? two.x : two.y;
                    const bool c0 = (cur >> KEY_O_SHIFT) & 1ull, c1 = cur >> 63;
                    const bool same = other ? c1 : c0, opposite = self_rc ? c0 : (other ? c0 : c1);
                    if constexpr (ABL != 1) st_tmp[jd.tpos_off + pos[u]] = (int8_t)(same ? (opposite ? 1 : 0) : 2);
                    else if (same && opposite && pos[u] == 0xFFFFFFF0u) st_tmp[0] = 1;
                    break;
                }
                if (two.x == EMPTY_KEY || two.y == EMPTY_KEY) break;
                s = (s + 2) & M;
            }
#else
            for (int probes = 0; probes < LDS_SLOTS; ++probes) {
                const unsigned long long cur = keys[s];
                if ((cur & KEY_BITS) == key) {                              // (EMPTY has all 62 bits set: no k-mer is that key)
                    const bool c0 = (cur >> KEY_O_SHIFT) & 1ull, c1 = cur >> 63;
                    const bool same = other ? c1 : c0, opposite = self_rc ? c0 : (other ? c0 : c1);
                    if constexpr (ABL != 1) st_tmp[jd.tpos_off + pos[u]] = (int8_t)(same ? (opposite ? 1 : 0) : 2);
                    else if (same && opposite && pos[u] == 0xFFFFFFF0u) st_tmp[0] = 1;
                    break;
                }
                if (cur == EMPTY_KEY) break;
                s = (s + 1) & M;
            }
#endif
        }
    };
    answer_step(kw_t, pos_t, ok_t);
    for (uint32_t e0 = threadIdx.x + KU * LDS_THREADS; e0 - threadIdx.x < n_tig; e0 += KU * LDS_THREADS) {
        uint32_t pos[KU]; bool ok[KU]; KmerWords kw[KU];
#pragma unroll
        for (int u = 0; u < KU; ++u) { const uint32_t e = e0 + u * LDS_THREADS; ok[u] = e < n_tig; pos[u] = list_t[ok[u] ? e : n_tig - 1]; }
#pragma unroll
        for (int u = 0; u < KU; ++u) kw[u] = kmer_words(T.two, jd.tig_abs + pos[u]);
        answer_step(kw, pos, ok);
    }
    KPROF_LAP(6);                                                       // look-ups + STATE_MER stores issued
#ifdef PAV_KMER_PROF
    if (threadIdx.x == 0) atomicAdd(&g_kmer_prof[15], 1ull);
#endif
}

__global__ __launch_bounds__(LDS_THREADS) void k_kmer_lds(const PartItem *__restrict__ items, const JobDev *__restrict__ jobs,
                                                          SeqView R, SeqView T, int k, uint32_t limit,
                                                          const uint32_t *__restrict__ lists, const uint32_t *__restrict__ bcount,
                                                          int8_t *__restrict__ st_tmp, JobStat *__restrict__ stat) {
    kmer_lds_body<0, 0>(items, jobs, R, T, k, limit, lists, bcount, st_tmp, stat);
}
#ifdef PAV_TUNING
template <int ABL, int ABL_W>
__global__ __launch_bounds__(LDS_THREADS) void k_kmer_abl(const PartItem *__restrict__ items, const JobDev *__restrict__ jobs,
                                                          SeqView R, SeqView T, int k, uint32_t limit,
                                                          const uint32_t *__restrict__ lists, const uint32_t *__restrict__ bcount,
                                                          int8_t *__restrict__ st_tmp, JobStat *__restrict__ stat) {
    kmer_lds_body<ABL, ABL_W>(items, jobs, R, T, k, limit, lists, bcount, st_tmp, stat);
}
#endif

// Per-tile and per-job counts of the STATE_MER values k_kmer_lds has written (+ the span of FWD k-mers for scan-only batches),
// one workgroup per tile, 8 positions per lane; jobs with HBM tables were done by k_tig_state.
__global__ __launch_bounds__(256) void k_state_combine(const JobDev *__restrict__ jobs, const uint32_t *__restrict__ tile_job,
                                                       const int8_t *__restrict__ st_tmp, JobStat *__restrict__ stat,
                                                       uint32_t *__restrict__ tile_cnt /* [tiles][4]: [1 + s] rows of state s; null: not wanted */,
                                                       int want_span = 0) {
    __shared__ uint32_t red[4][3];
    const uint32_t j = tile_job[blockIdx.x];
    if (!jobs[j].n_parts) return;
    uint32_t fwd_lo = ~0u, fwd_hi = 0;
    const uint64_t at = (uint64_t)blockIdx.x * DTILE + (uint64_t)threadIdx.x * 8;
    const uint64_t s8 = *reinterpret_cast<const uint64_t *>(st_tmp + at);
    uint32_t n[3] = {0, 0, 0};
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int st = (int)(int8_t)(uint8_t)(s8 >> (8 * t));
        if (st >= 0) n[st]++;
        if (st == 0) { if (fwd_lo == ~0u) fwd_lo = (uint32_t)t; fwd_hi = (uint32_t)t; }
    }
    // first / last position of the region with a FWD k-mer (scan-only batches): lanes hold rising positions, so a wave's first
    // is its lowest lane's and its last its highest lane's; one pair of atomics per workgroup
    __shared__ uint32_t span[4][2];
    if (want_span) {
        const unsigned long long has = __ballot(fwd_lo != ~0u);
        const int lane = threadIdx.x & 63;
        if (has) {
            const int lo_lane = __ffsll((long long)has) - 1, hi_lane = 63 - __clzll((long long)has);
            const uint32_t i0 = (uint32_t)(at - jobs[j].tpos_off);
            if (lane == lo_lane) span[threadIdx.x >> 6][0] = 0xFFFFFFFFu - (i0 + fwd_lo);
            if (lane == hi_lane) span[threadIdx.x >> 6][1] = i0 + fwd_hi + 1u;
        } else if (lane == 0) { span[threadIdx.x >> 6][0] = 0u; span[threadIdx.x >> 6][1] = 0u; }
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) n[s] += __shfl_xor(n[s], d);
    }
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int s = 0; s < 3; ++s) red[threadIdx.x >> 6][s] = n[s];
    __syncthreads();
    if (threadIdx.x < 3) {
        const uint32_t v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (v) atomicAdd(&stat[j].st_count[threadIdx.x], v);
        if (tile_cnt) tile_cnt[(uint64_t)blockIdx.x * 4 + 1 + threadIdx.x] = v;
    }
    if (want_span && threadIdx.x >= 64 && threadIdx.x < 66) {         // (a lane of another wave than the one busy with the counts)
        const int q = threadIdx.x - 64;
        const uint32_t v = max(max(span[0][q], span[1][q]), max(span[2][q], span[3][q]));
        if (v) atomicMax(q ? &stat[j].last1 : &stat[j].inv_first, v);
    }
}

// Device-planned batches: the compaction prefix straight from k_state_combine's per-tile state counts - a state whose job total
// is below the minimum is dropped (density.py:181-190) - one workgroup per counter ([0] rows kept, [1 + s] rows of state s kept),
// 4096 tiles per step.  Replaces k_compact_reduce + the single-workgroup k_scan_tiles4 of the host-planned path.
__global__ __launch_bounds__(256) void k_scan_tiles_keep(const uint32_t *__restrict__ tile_cnt, const uint32_t *__restrict__ tile_job,
                                                        const JobStat *__restrict__ stat, uint32_t min_state_count,
                                                        unsigned long long *__restrict__ tile_pre, uint32_t n_tiles) {
    constexpr int PER = 16;
    __shared__ uint32_t lds[4];
    __shared__ uint32_t tile[256 * (PER + 1)];
    const uint32_t q = blockIdx.x;
    unsigned long long carry = 0;
    for (uint32_t base = 0; base < n_tiles; base += 256 * PER) {
        uint32_t c[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const uint32_t i = base + k * 256 + threadIdx.x;
            uint32_t v = 0;
            if (i < n_tiles) {
                const JobStat &js = stat[tile_job[i]];
                if (q) v = js.st_count[q - 1] >= min_state_count ? tile_cnt[(uint64_t)i * 4 + q] : 0u;
                else
#pragma unroll
                    for (int s3 = 0; s3 < 3; ++s3) v += js.st_count[s3] >= min_state_count ? tile_cnt[(uint64_t)i * 4 + 1 + s3] : 0u;
            }
            c[k] = v;
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) { const uint32_t jx = k * 256 + threadIdx.x; tile[jx + jx / PER] = c[k]; }
        __syncthreads();
        uint32_t sum = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) { c[k] = tile[threadIdx.x * (PER + 1) + k]; sum += c[k]; }
        uint32_t inc = sum;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(inc, d); if (lane >= d) inc += y; }
        if (lane == 63) lds[wave] = inc;
        __syncthreads();
        uint32_t before = inc - sum, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { if (w < wave) before += lds[w]; tot += lds[w]; }
        unsigned long long run = carry + before;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const uint32_t i = base + threadIdx.x * PER + k;
            if (i < n_tiles) tile_pre[(uint64_t)i * 4 + q] = run;
            run += c[k];
        }
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_pre[(uint64_t)n_tiles * 4 + q] = carry;
}

// ---- compaction to informative rows (scripts/density.py:178-203) ----------------------------------------------
// Scanned quantities per position: [0] row kept, [1..3] row of state 0/1/2 kept.
__device__ __forceinline__ void block_scan4(uint32_t (&v)[4], uint32_t (&total)[4], uint32_t *lds /* 16 */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t x = v[q];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { uint32_t y = __shfl_up(x, d); if (lane >= d) x += y; }
        inc[q] = x;
        if (lane == 63) lds[wave * 4 + q] = x;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t base = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { uint32_t s = lds[w * 4 + q]; if (w < wave) base += s; tot += s; }
        total[q] = tot;
        v[q] = base + inc[q] - v[q];
    }
    __syncthreads();
}

// A region whose k-mers are all FWD once the low-count states are dropped (1 025 of the 1 126 flagged regions of the bench
// haplotype: a cluster of SNVs or indels, not an inversion).  Its STATE column is 0 in every row whatever the densities are -
// the other two kernel columns are zero, arg-max takes the first (scripts/density.py:300-342) - so the scan driver knows its
// outcome from the counts alone: one run of state 0 over its rows, "Found no inverted k-mer states".  In a batch the driver
// marks scan-only (pav_ctx::den_scan_only) such a region is neither compacted nor evaluated; its run comes from the first and
// last contig position with a FWD k-mer (k_state_combine).  Its table does not exist: pav_density_table refuses.
__device__ __host__ __forceinline__ bool fwd_only(const JobStat &js, uint32_t min_state_count) {
    return js.st_count[1] < min_state_count && js.st_count[2] < min_state_count;
}

// The three STATE_MER counts of a region in registers.  (The kernels used to copy the whole JobStat and index st_count[] with the
// row's state: a dynamically indexed private array lives in scratch memory, so every lane wrote 128 B of JobStat through the caches
// to HBM - WRITE_SIZE of k_compact_scatter was 235 MB per launch for 120 MB of rows.)
struct StateCounts { uint32_t c0, c1, c2; };
__device__ __forceinline__ StateCounts state_counts(const JobStat *stat, uint32_t j) {
    return StateCounts{stat[j].st_count[0], stat[j].st_count[1], stat[j].st_count[2]};
}
__device__ __forceinline__ bool fwd_only(const StateCounts &sc, uint32_t min_state_count) {
    return sc.c1 < min_state_count && sc.c2 < min_state_count;
}
__device__ __forceinline__ int keep_state(int st, const StateCounts &sc, uint32_t min_state_count) {
    if (st < 0) return -1;
    const uint32_t n = st == 0 ? sc.c0 : st == 1 ? sc.c1 : sc.c2;
    return n >= min_state_count ? st : -1;                             // low-count states are dropped (density.py:181-190)
}

__global__ __launch_bounds__(256) void k_compact_reduce(const uint32_t *__restrict__ tile_job, const JobStat *__restrict__ stat,
                                                        const int8_t *__restrict__ st_tmp, uint32_t min_state_count,
                                                        uint32_t *__restrict__ tile_sum /* [tiles][4] */) {
    __shared__ uint32_t lds[16];
    const uint32_t j = tile_job[blockIdx.x];
    const StateCounts js = state_counts(stat, j);
    const uint64_t base = (uint64_t)blockIdx.x * DTILE + (uint64_t)threadIdx.x * 8;
    const uint64_t packed = *reinterpret_cast<const uint64_t *>(st_tmp + base);
    uint32_t c[4] = {0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int st = keep_state((int)(int8_t)(packed >> (8 * t)), js, min_state_count);
        if (st >= 0) { c[0]++; c[1 + st]++; }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) c[q] += __shfl_xor(c[q], d);
    }
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int q = 0; q < 4; ++q) lds[(threadIdx.x >> 6) * 4 + q] = c[q];
    __syncthreads();
    if (threadIdx.x < 4)
        tile_sum[(uint64_t)blockIdx.x * 4 + threadIdx.x] = lds[threadIdx.x] + lds[4 + threadIdx.x] + lds[8 + threadIdx.x] + lds[12 + threadIdx.x];
}

// Single workgroup: exclusive scan over tiles of the 4 counters.
__global__ __launch_bounds__(256) void k_scan_tiles4(const uint32_t *__restrict__ tile_sum, unsigned long long *__restrict__ tile_pre,
                                                     uint32_t n_tiles) {
    __shared__ uint32_t lds[16];
    __shared__ unsigned long long carry[4];
    if (threadIdx.x < 4) carry[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_tiles; base += 256) {
        const uint32_t i = base + threadIdx.x;
        uint32_t v[4], tot[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = i < n_tiles ? tile_sum[(uint64_t)i * 4 + q] : 0;
        block_scan4(v, tot, lds);
        if (i < n_tiles)
#pragma unroll
            for (int q = 0; q < 4; ++q) tile_pre[(uint64_t)i * 4 + q] = carry[q] + v[q];
        __syncthreads();
        if (threadIdx.x == 0)
#pragma unroll
            for (int q = 0; q < 4; ++q) carry[q] += tot[q];
        __syncthreads();
    }
    if (threadIdx.x < 4) tile_pre[(uint64_t)n_tiles * 4 + threadIdx.x] = carry[threadIdx.x];
}

struct CompactArgs {
    const JobDev *jobs; const uint32_t *tile_job; JobStat *stat; const int8_t *st_tmp;
    const unsigned long long *tile_pre; SeqView T; int k; uint32_t min_state_count;
    uint32_t *index; int8_t *state_mer; int8_t *state; unsigned long long *kmer; uint32_t *list[3];
    HeadEvent *events; uint32_t ev_cap; uint32_t *ev_count;   // run heads of STATE_MER (closed-form run sums); null: not wanted
    uint32_t *tile_heads, *tile_head_cnt;                     // device-planned batches instead: HEADS_PER_TILE ordered slots per tile
    int scan_only;                                            // tables of regions with FWD k-mers only are not wanted (fwd_only below)
};

#ifndef PAV_COMPACT_WAVES
#define PAV_COMPACT_WAVES 4            // waves per SIMD the compiler is asked to fit the kernel into (129 VGPRs without: 3)
#endif
__global__ __launch_bounds__(256, PAV_COMPACT_WAVES) void k_compact_scatter(CompactArgs A) {
    __shared__ uint32_t lds[16];
    __shared__ unsigned long long red[4][7];
    // the tile's kept rows are staged in LDS in row order and leave with coalesced stores
    __shared__ unsigned long long s_kmer[DTILE];
    // the three per-state lists share ONE array (a row belongs to one state: their entries add up to the tile's rows): state s
    // starts behind the rows of the states in front of it - 34 KiB of LDS per workgroup instead of 50, four workgroups per CU
    // (round 6: INDEX is staged as the 16-bit offset of the position in its tile - 31 KiB, FIVE workgroups a CU, which is what the
    //  registers allow; with a 32-bit INDEX it was 34.4 KiB and four)
    __shared__ uint32_t s_list1[DTILE];
    __shared__ uint16_t s_off[DTILE];
    __shared__ int8_t s_mer[DTILE];
    const uint32_t j = A.tile_job[blockIdx.x];
    const JobDev jd = A.jobs[j];
    const StateCounts js = state_counts(A.stat, j);
    if (A.scan_only && fwd_only(js, A.min_state_count)) return;       // nobody will read this region's rows
    const uint64_t base = (uint64_t)blockIdx.x * DTILE + (uint64_t)threadIdx.x * 8;
    const uint64_t packed = *reinterpret_cast<const uint64_t *>(A.st_tmp + base);
    int st[8];
    uint32_t c[4] = {0, 0, 0, 0}, tot[4];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        st[t] = keep_state((int)(int8_t)(packed >> (8 * t)), js, A.min_state_count);
        if (st[t] >= 0) { c[0]++; c[1 + st[t]]++; }
    }
    block_scan4(c, tot, lds);                                          // c[] = exclusive prefixes inside the tile
    // job-relative bases of the tile: global tile prefix minus the prefix at the job's first tile
    unsigned long long tile0[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        tile0[q] = A.tile_pre[(uint64_t)blockIdx.x * 4 + q] - A.tile_pre[(uint64_t)jd.first_tile * 4 + q];
    unsigned long long s1[3] = {0, 0, 0}, s2[3] = {0, 0, 0};
    const uint32_t lbase[3] = {0u, tot[1], tot[1] + tot[2]};
    // the lane's eight k-mers start at consecutive bases: one 64-base window of the 2-bit plane (a 4-byte-aligned 16-byte load +
    // one dword), fetched before the loop - a window per kept k-mer inside the branch cost eight dependent rounds of loads, and
    // the non-ACGT plane is not needed (a kept k-mer is a valid one)
    const uint64_t i0 = base - jd.tpos_off;
    uint64_t w_lo, w_hi;
    {
        const uint64_t a0 = jd.tig_abs + (i0 < jd.tig_len ? i0 : jd.tig_len);      // past the region: stay inside the record's pad block
        const uint32_t *p = A.T.two + (a0 >> 4);
        typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
        const u32x4_a4 v = *reinterpret_cast<const u32x4_a4 *>(p);
        const uint32_t v4 = p[4], sh = ((uint32_t)a0 & 15u) * 2u;
        w_lo = (uint64_t)__builtin_amdgcn_alignbit(v.z, v.y, sh) << 32 | __builtin_amdgcn_alignbit(v.y, v.x, sh);
        w_hi = (uint64_t)__builtin_amdgcn_alignbit(v4, v.w, sh) << 32 | __builtin_amdgcn_alignbit(v.w, v.z, sh);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        if (st[t] < 0) continue;
        const uint32_t lr = c[0]++;                                    // row inside the tile
        const uint64_t row = tile0[0] + lr;
        const uint64_t x = (t ? (w_lo >> (2 * t) | w_hi << (64 - 2 * t)) : w_lo) & kmer_mask(A.k);
        s_off[lr] = (uint16_t)(threadIdx.x * 8 + t);                   // INDEX = the tile's first offset in region_tig + this
        s_mer[lr] = (int8_t)st[t];
        s_kmer[lr] = rev_groups(x, A.k);
        s_list1[lbase[st[t]] + c[1 + st[t]]++] = (uint32_t)row;        // INDEX_DEN of the state's data points, ascending
        s1[st[t]] += row; s2[st[t]] += row * row;
    }
    __syncthreads();
    const uint64_t o0 = jd.tpos_off + tile0[0];
    const uint32_t tile_i0 = (uint32_t)((uint64_t)blockIdx.x * DTILE - jd.tpos_off);
    for (uint32_t r = threadIdx.x; r < tot[0]; r += 256) {
        A.index[o0 + r] = tile_i0 + s_off[r];
        A.state_mer[o0 + r] = s_mer[r];
        A.state[o0 + r] = -1;                                          // df['STATE'] = -1 (density.py:163)
        A.kmer[o0 + r] = s_kmer[r];
        // runs of equal STATE_MER: every change inside the tile is a head; the tile's first row is reported as well (pad = 1)
        // and the host drops it when the run merely continues from the tile before
        if (A.events && (r == 0 || s_mer[r] != s_mer[r - 1])) {
            const uint32_t e = atomicAdd(A.ev_count, 1u);
            if (e < A.ev_cap) A.events[e] = HeadEvent{j, (uint32_t)(tile0[0] + r), (int32_t)s_mer[r], 0u, 0u, r == 0 ? 1u : 0u};
        }
    }
#pragma unroll
    for (int s = 0; s < 3; ++s)
        for (uint32_t r = threadIdx.x; r < tot[1 + s]; r += 256) A.list[s][jd.tpos_off + tile0[1 + s] + r] = s_list1[lbase[s] + r];
    if (A.tile_heads) {
        // Device-planned batches: the tile's run heads of STATE_MER in row order, at fixed slots (no atomics, nothing to sort:
        // tiles are ordered by (job, row)).  Row r of the tile (staged order) is a head when the state changes in front of it;
        // the tile's first row always is one - k_plan drops it when the run continues from the tile before.
        __shared__ uint32_t s_hcnt[DTILE / 256][4];
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        unsigned long long hb[DTILE / 256];
#pragma unroll
        for (int it = 0; it < DTILE / 256; ++it) {
            const uint32_t r = it * 256 + threadIdx.x;
            const bool head = r < tot[0] && (r == 0 || s_mer[r] != s_mer[r - 1]);
            hb[it] = __ballot(head);
            if (lane == 0) s_hcnt[it][wave] = (uint32_t)__popcll(hb[it]);
        }
        __syncthreads();
        uint32_t before = 0;
#pragma unroll
        for (int it = 0; it < DTILE / 256; ++it) {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const uint32_t c4 = s_hcnt[it][w];
                if (w == wave) {
                    const uint32_t r = it * 256 + threadIdx.x;
                    if (hb[it] >> lane & 1ull) {
                        const uint32_t rank = before + (uint32_t)__popcll(hb[it] & ((1ull << lane) - 1ull));
                        if (rank < HEADS_PER_TILE)
                            A.tile_heads[(uint64_t)blockIdx.x * HEADS_PER_TILE + rank] = (uint32_t)(tile0[0] + r) | (uint32_t)s_mer[r] << 26;
                    }
                }
                before += c4;
            }
        }
        if (threadIdx.x == 0) A.tile_head_cnt[blockIdx.x] = before;      // (may exceed the slots: k_plan flags it)
    }
    // block reduction of the per-state moments, one atomic per block and quantity
#pragma unroll
    for (int s = 0; s < 3; ++s) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { s1[s] += __shfl_xor(s1[s], d); s2[s] += __shfl_xor(s2[s], d); }
    }
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        for (int s = 0; s < 3; ++s) { red[w][s] = s1[s]; red[w][3 + s] = s2[s]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const unsigned long long v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (v) atomicAdd(threadIdx.x < 3 ? &A.stat[j].s1[threadIdx.x] : &A.stat[j].s2[threadIdx.x - 3], v);
    }
    if (threadIdx.x == 0) {
        if (tot[0]) atomicAdd(&A.stat[j].n_rows, tot[0]);
        for (int s = 0; s < 3; ++s) if (tot[1 + s]) atomicAdd(&A.stat[j].m[s], tot[1 + s]);
    }
}

// ---- device-side planning of the density stage (one wave per job) ------------------------------------------------------------
// What the host used to do between two synchronisations of a batch (status of every region, bandwidths, sampled sites, the run
// lists of the closed-form sums) happens here, so that the kernels of a scan round are queued back to back and the host reads
// ONE block of results at the end (pav_density_batch, "device-planned").  The arithmetic is the host's: exact integer moments,
// n^(-1/5) from a table the host filled with libm's pow (scipy's factor, scripts/density.py:198), IEEE sqrt and division.
struct PlanArgs {
    const JobDev *jobs; const JobStat *stat; JobKde *kde; uint32_t n_jobs;
    const uint32_t *tile_heads, *tile_head_cnt; RunDev *runs;          // per-tile head slots -> per-job, per-state run lists
    const double *pow_tab; uint32_t pow_n;                            // pow_tab[n] = pow((double)n, -1.0 / 5.0)
    uint32_t min_informative, max_ref_kmer_count; double den_smooth, norm0;   // norm0 = pow(2 pi, -0.5) as the host's libm returns it
    uint32_t *flags;                                                  // [0] != 0: the batch needs the host-planned path
    struct FinBlockOut { uint32_t job, row0; } *fin_blocks; uint32_t *n_fin_blocks; uint32_t fin_rows;   // blocks of table rows for k_finalize_blocks
};
constexpr uint32_t PLAN_TILE_OVERFLOW = 1, PLAN_POW_RANGE = 2;

// (double) of an unsigned 128-bit integer, round to nearest even - what the host's conversion of the exact variance numerator does
__device__ __forceinline__ double u128_to_double(unsigned __int128 v) {
    const uint64_t hi = (uint64_t)(v >> 64), lo = (uint64_t)v;
    if (hi == 0) return (double)lo;                                    // (the 64-bit conversion rounds to nearest even)
    const int lz = __clzll((long long)hi);                             // top bit at 127 - lz
    const int sh = 64 - lz;                                            // bits below a 64-bit window that holds the top bit
    uint64_t top = (uint64_t)(v >> sh);                                // 64 significant bits
    const bool sticky = (v & (((unsigned __int128)1 << sh) - 1)) != 0;
    top |= sticky ? 1ull : 0ull;                                       // the conversion of `top` rounds on bit 11 .. 0 with this sticky bit
    return ldexp((double)top, sh);
}

__global__ __launch_bounds__(256) void k_plan(PlanArgs A) {
    __shared__ uint32_t s_stage[4][64 * HEADS_PER_TILE];               // a chunk of 64 tiles' heads, flattened in order (per wave)
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t j = blockIdx.x * 4 + wave;
    if (j >= A.n_jobs) return;
    const JobStat st = A.stat[j];
    const JobDev jd = A.jobs[j];
    JobKde kd = A.kde[j];                                              // samp_off / heads_off / srs come from the host (upper bounds)
    kd.finalised = 0; kd.n = 0; kd.n_samp = 0; kd.use_runs = 0; kd.ps_mask = 0; kd.all_direct = 0; kd.n_heads = 0;
    for (int q = 0; q < 3; ++q) { kd.m[q] = 0; kd.run_off[q] = 0; kd.n_run[q] = 0; kd.inv_h[q] = kd.norm[q] = kd.w[q] = kd.cnt[q] = kd.h[q] = 0.0; }
    const bool fail = st.n_ref_valid == 0 || st.max_count > A.max_ref_kmer_count;                   // density.py:510-527
    const bool fin = !fail && st.n_rows != 0 && st.n_rows >= A.min_informative;                      // :193-195
    if (fin) {
        const uint32_t n = st.n_rows;
        kd.finalised = 1; kd.n = n;
        kd.n_samp = (n + kd.srs - 1) / kd.srs;
        if ((uint64_t)(kd.n_samp - 1) * kd.srs != n - 1) kd.n_samp += 1;                            // :213-214
        if (n >= A.pow_n) { if (lane == 0) atomicOr(A.flags, PLAN_POW_RANGE); }
        const double bandwidth = A.pow_tab[n < A.pow_n ? n : 0] * A.den_smooth;                      // :198
        for (int q = 0; q < 3; ++q) {
            const uint64_t m = st.m[q];
            kd.m[q] = (uint32_t)m;
            kd.cnt[q] = (double)m;
            if (m == 0) continue;
            const unsigned __int128 num = (unsigned __int128)m * st.s2[q] - (unsigned __int128)st.s1[q] * st.s1[q];
            const double var = u128_to_double(num) / ((double)m * (double)(m - 1));
            const double h = sqrt(var) * bandwidth;
            kd.h[q] = h;
            kd.inv_h[q] = 1.0 / h;
            kd.norm[q] = A.norm0 / h;
            kd.w[q] = 1.0 / (double)m;
        }
        // ---- run lists of STATE_MER, one per state (what kde_state_runs reads): the tiles' head slots flattened in row order;
        //      a run that continues across a tile edge is one run.  Pass 0 counts the runs of every state, pass 1 writes them
        //      behind one another: [run_off[s], run_off[s] + n_run[s]) of the job's slice of the run arena.
        const uint32_t n_tiles = (uint32_t)((std::max<uint64_t>(jd.tig_len, 1) + DTILE - 1) / DTILE);
        uint32_t out = 0, cnt_s[3] = {0, 0, 0};
        bool over = false;
        for (int pass = 0; pass < 2; ++pass) {
            uint32_t base_s[3] = {0, 0, 0}, carry_st = 3, pend = ~0u;          // pend: slot of the run that is still open
            if (pass) { base_s[0] = kd.heads_off; base_s[1] = base_s[0] + cnt_s[0]; base_s[2] = base_s[1] + cnt_s[1]; }
            for (uint32_t c0 = 0; c0 < n_tiles; c0 += 64) {
                const uint32_t tile = jd.first_tile + c0 + lane;
                uint32_t cnt = c0 + lane < n_tiles ? A.tile_head_cnt[tile] : 0u;
                if (cnt > HEADS_PER_TILE) { over = true; cnt = HEADS_PER_TILE; }
                uint32_t inc = cnt;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(inc, d); if ((int)lane >= d) inc += y; }
                const uint32_t total = __shfl(inc, 63), at = inc - cnt;
                for (uint32_t e = 0; e < cnt; ++e) s_stage[wave][at + e] = A.tile_heads[(uint64_t)tile * HEADS_PER_TILE + e];
                __builtin_amdgcn_wave_barrier();
                for (uint32_t b = 0; b < total; b += 64) {
                    const uint32_t i = b + lane;
                    const bool live = i < total;
                    const uint32_t wv = live ? s_stage[wave][i] : 0u;
                    const uint32_t stt = (wv >> 26) & 3u, row = wv & HEAD_ROW_MASK;
                    const uint32_t prev = i ? ((live ? s_stage[wave][i - 1] : 0u) >> 26) & 3u : carry_st;
                    const bool head = live && stt != prev;
                    const unsigned long long mh = __ballot(head);
                    const unsigned long long lt = (1ull << lane) - 1ull;
                    uint32_t slot = 0;
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const unsigned long long mq = __ballot(head && stt == (uint32_t)q);
                        if (stt == (uint32_t)q) slot = base_s[q] + (uint32_t)__popcll(mq & lt);
                        base_s[q] += (uint32_t)__popcll(mq);
                    }
                    if (pass && mh) {
                        // a head opens its run and closes the run of the head in front of it (this chunk's, or the open one carried in)
                        const unsigned long long below = mh & lt;
                        const int pl = below ? 63 - __clzll((long long)below) : 0;
                        const uint32_t pslot = (uint32_t)__shfl((int)slot, pl);
                        if (head) {
                            A.runs[slot].a = row;
                            const uint32_t close = below ? pslot : pend;
                            if (close != ~0u) A.runs[close].b = row - 1u;
                        }
                        pend = (uint32_t)__shfl((int)slot, 63 - __clzll((long long)mh));
                    }
                    out += pass ? 0u : (uint32_t)__popcll(mh);
                    const uint32_t last = total - b < 64 ? total - b - 1 : 63;
                    carry_st = (uint32_t)__shfl((int)stt, (int)last);
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (!pass) { for (int q = 0; q < 3; ++q) cnt_s[q] = base_s[q]; }
            else if (pend != ~0u && lane == 0) A.runs[pend].b = n - 1u;        // the last run ends with the table
        }
        if (over && lane == 0) atomicOr(A.flags, PLAN_TILE_OVERFLOW);
        kd.n_heads = out;
        kd.run_off[0] = kd.heads_off; kd.run_off[1] = kd.run_off[0] + cnt_s[0]; kd.run_off[2] = kd.run_off[1] + cnt_s[1];
        for (int q = 0; q < 3; ++q) kd.n_run[q] = cnt_s[q];
        kd.use_runs = out <= KDE_RUNS_MAX ? 1u : 0u;                    // very fragmented region: direct kernel
        kd.all_direct = kd.use_runs ? 0u : 1u;
        for (int q = 0; q < 3; ++q)
            if (kd.m[q] && !(kd.use_runs && kd.h[q] >= KDE_RUNS_MIN_H)) kd.ps_mask |= 1u << q;
    }
    if (lane == 0) A.kde[j] = kd;
    if (A.fin_blocks) {                                                 // the blocks of this region that hold table rows (k_finalize_blocks)
        const uint32_t rows = kd.finalised ? kd.n : st.n_rows;
        const uint32_t nbk = (rows + A.fin_rows - 1) / A.fin_rows;
        uint32_t base = 0;
        if (lane == 0 && nbk) base = atomicAdd(A.n_fin_blocks, nbk);
        base = (uint32_t)__shfl((int)base, 0);
        for (uint32_t i = lane; i < nbk; i += 64) A.fin_blocks[base + i] = PlanArgs::FinBlockOut{j, i * A.fin_rows};
    }
}

// Evaluation tiles of the windows k_windows queued (device-planned batches): 64 fill points each, in job order.
struct FillPlanArgs { const JobStat *stat; const JobKde *kde; uint32_t n_jobs; EvalTile *ftiles; uint32_t *n_ftiles; uint32_t cap; };
__global__ __launch_bounds__(1024) void k_plan_fill(FillPlanArgs A) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t j0 = 0; j0 < A.n_jobs; j0 += 1024) {
        const uint32_t j = j0 + threadIdx.x;
        const uint32_t fill = j < A.n_jobs && A.kde[j].finalised ? A.stat[j].fill_n : 0u;
        const uint32_t nt = (fill + 63) / 64;
        uint32_t inc = nt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(inc, d); if ((int)lane >= d) inc += y; }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        uint32_t at = s_carry + inc - nt, tot = 0;
        for (int w = 0; w < 16; ++w) { if (w < (int)wave) at += s_wave[w]; tot += s_wave[w]; }
        for (uint32_t f = 0; f < nt; ++f)
            if (at + f < A.cap) A.ftiles[at + f] = EvalTile{j, f * 64, fill - f * 64 < 64 ? fill - f * 64 : 64u, 1u};
        __syncthreads();
        if (threadIdx.x == 0) s_carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *A.n_ftiles = s_carry < A.cap ? s_carry : A.cap;
}

// ---- KDE -----------------------------------------------------------------------------------------------------
__device__ __forceinline__ int argmax3(double a, double b, double c) {   // np.argmax: first maximum wins
    int m = 0; double v = a;
    if (b > v) { v = b; m = 1; }
    if (c > v) m = 2;
    return m;
}


// ---- near-tie guard (include/pav_amd.h; SURVEY.md section 7, hard part 2) -----------------------------------------
// Float decisions of scripts/density.py - arg-max at the sampled sites (:250-255), density_change (:277-281), final arg-max
// (:335-338) - whose margin is below G.rel are not trusted when they rest on closed-form run sums: the kernels that take
// the decisions append the sites they depend on to a list; k_redo evaluates those in scipy's accumulation order and the
// host redoes everything downstream (pav_density_batch).  Decisions that are still within G.unres afterwards are counted.
constexpr double GUARD_REL_DEFAULT = 1e-9;
constexpr double GUARD_UNRESOLVED = 1e-13;
constexpr uint32_t GUARD_CAP_DEFAULT = 1u << 20;
constexpr uint32_t SF_EXACT = 1, SF_PENDING = 2, SF_WIN_COUNTED = 4;  // samp_flag[]: one word per sampled site
constexpr uint8_t RF_EXACT = 1, RF_PENDING = 2, RF_COUNTED = 4;       // row_flag[]: one byte per table row
constexpr unsigned long long GE_ROW = 0x80000000ull;                  // list entry: job << 32 | GE_ROW? | sampled site / row

struct GuardDev { uint32_t n_entries, overflow, pad[2]; };
struct GuardArgs {
    double rel, unres;                    // rel <= 0: guard off
    uint32_t pass, cap;                   // pass 0: the regular evaluation; >= 1: redone from re-evaluated sites
    uint32_t *samp_flag; uint8_t *row_flag;
    GuardDev *g; unsigned long long *entries;
    JobStat *stat;
};

// (max - second) / max < rel.  All zero (absent states, underflow): no tie to resolve, np.argmax takes the first.
__device__ __forceinline__ bool near_argmax(double a, double b, double c, double rel) {
    const int m = argmax3(a, b, c);
    const double mx = m == 0 ? a : (m == 1 ? b : c);
    if (!(mx > 0.0)) return false;
    const double second = m == 0 ? fmax(b, c) : (m == 1 ? fmax(a, c) : fmax(a, b));
    return mx - second < rel * mx;
}

__device__ __forceinline__ void guard_append(const GuardArgs &G, unsigned long long entry) {
    const uint32_t e = atomicAdd(&G.g->n_entries, 1u);
    if (e < G.cap) G.entries[e] = entry; else G.g->overflow = 1;
}
__device__ __forceinline__ void guard_flag_sample(const GuardArgs &G, uint32_t job, uint32_t q, uint64_t so) {
    const uint32_t old = atomicOr(&G.samp_flag[so], SF_PENDING);
    if (!(old & (SF_PENDING | SF_EXACT))) guard_append(G, (unsigned long long)job << 32 | q);
}
__device__ __forceinline__ void guard_flag_row(const GuardArgs &G, uint32_t job, uint32_t row, uint64_t ap) {
    const uint8_t old = G.row_flag[ap];                                // only the row's own lane (k_finalize) gets here
    if (old & (RF_PENDING | RF_EXACT)) return;
    G.row_flag[ap] = old | RF_PENDING;
    guard_append(G, (unsigned long long)job << 32 | GE_ROW | row);
}

// exp(-r^2 / 2) is exactly 0.0 in float64 from |r| = 38.61 on (r^2 / 2 > 745.2 = -log of half the smallest subnormal), so a data
// point further than this from an evaluation point adds +0.0 to scipy's sum: leaving it out changes no bit of the result.
constexpr double KDE_ZERO_R = 38.7;

__device__ __forceinline__ double uniform_f64(double v) {          // the first lane's value, in scalar registers
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// points_ = data * (1 / h) - what scipy's solve_triangular computes for a 1x1 cho_cov - made where it is read: the positions of a
// state's rows times the state's 1 / h (round 3 kept them as arrays of doubles, written by a kernel of their own, k_pscale:
// 8 B per row and state, and a launch per round that most jobs left at its first branch).
struct ScaledList {
    const uint32_t *l; double inv_h;
    __device__ __forceinline__ double operator[](uint32_t i) const { return (double)l[i] * inv_h; }
};

// First index i in [0, m) with ps[i] >= v (ps ascending; wave-uniform arguments: scalar loads).
__device__ __forceinline__ uint32_t ps_lower_bound(const ScaledList ps, uint32_t m, double v) {
    uint32_t lo = 0, hi = m;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (ps[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}

// gaussian_kernel_estimate (scipy/stats/_stats.pyx) for one state at one evaluation point; `ps` is wave-uniform so
// the data stream goes through the scalar cache.  Accumulation order = data ascending, as in scipy.  [lo, hi): the data
// points that can contribute to any point of the wave (everything outside adds exact zeros).
__device__ __forceinline__ double kde_state(const ScaledList ps, uint32_t lo, uint32_t hi, double xs, double norm, double w) {
    double est = 0.0;
    uint32_t i = lo;
    for (; i + 4 <= hi; i += 4) {
        const double a0 = ps[i], a1 = ps[i + 1], a2 = ps[i + 2], a3 = ps[i + 3];
        const double r0 = a0 - xs, r1 = a1 - xs, r2 = a2 - xs, r3 = a3 - xs;
        est += w * (exp(-(r0 * r0) / 2) * norm);
        est += w * (exp(-(r1 * r1) / 2) * norm);
        est += w * (exp(-(r2 * r2) / 2) * norm);
        est += w * (exp(-(r3 * r3) / 2) * norm);
    }
    for (; i < hi; ++i) {
        const double r = ps[i] - xs;
        est += w * (exp(-(r * r) / 2) * norm);
    }
    return est;
}

// Sum of exp(-((i - x)/h)^2 / 2) over the consecutive integers i = a..b in closed form (Euler-Maclaurin):
//   integral (erf / erfc difference, chosen so that tails keep their relative accuracy)
//   + (f(a) + f(b)) / 2 + (f1(b) - f1(a)) / 12 - (f3(b) - f3(a)) / 720 + (f5(b) - f5(a)) / 30240,
// where fm is the m-th derivative: fm(t) = (-1)^m He_m(u) f / h^m, u = (t - x)/h.  The next term is below
// 27 / (1.2e6 h^7): < 1e-15 for h >= 32.
__device__ __forceinline__ double run_sum_em(double ua, double ub, double h, double inv_h, bool more) {
    const double fa = exp(-(ua * ua) / 2), fb = exp(-(ub * ub) / 2);
    const double s = 0.70710678118654752440;
    double integ;
    const double wd = ub - ua, u_near = fmax(1.0, ua >= 0.0 ? ua : (ub <= 0.0 ? -ub : 0.0));
    if (wd * u_near < 0.0625) {
        // A run much shorter than the bandwidth: the erfc difference cancels - relative error ~ eps / (wd max(1, |u|)) - while a
        // four-point Gauss-Legendre rule has relative error 5.6e-10 wd^8 max(105, u^8) < 1e-17 under the same condition
        // (f^(8) = He_8(u) f).  Such runs used to be summed term by term, up to h / 16 exp each: the longest tiles of a launch.
        const double c = 0.5 * (ua + ub), r = 0.5 * wd;
        const double x1 = r * 0.33998104358485626480, x2 = r * 0.86113631159405257522;
        const double g1 = exp(-((c - x1) * (c - x1)) / 2) + exp(-((c + x1) * (c + x1)) / 2);
        const double g2 = exp(-((c - x2) * (c - x2)) / 2) + exp(-((c + x2) * (c + x2)) / 2);
        integ = r * (0.65214515486254614263 * g1 + 0.34785484513745385737 * g2) * h;
    } else {
        if (ua >= 0.0) integ = erfc(ua * s) - erfc(ub * s);
        else if (ub <= 0.0) integ = erfc(-ub * s) - erfc(-ua * s);
        else integ = erf(ub * s) - erf(ua * s);
        integ *= h * 1.25331413731550025121;                           // sqrt(pi / 2)
    }
    const double ua2 = ua * ua, ub2 = ub * ub;
    const double h3a = ua * (ua2 - 3.0), h3b = ub * (ub2 - 3.0);
    const double h5a = ua * (ua2 * (ua2 - 10.0) + 15.0), h5b = ub * (ub2 * (ub2 - 10.0) + 15.0);
    const double ih2 = inv_h * inv_h;
    const double d1 = (ua * fa - ub * fb) * inv_h;                     // f1(b) - f1(a)
    const double d3 = (h3a * fa - h3b * fb) * inv_h * ih2;             // f3(b) - f3(a)
    const double d5 = (h5a * fa - h5b * fb) * inv_h * ih2 * ih2;       // f5(b) - f5(a)
    double sum = integ + 0.5 * (fa + fb) + d1 / 12.0 - d3 / 720.0 + d5 / 30240.0;
    if (more) {
        // On a flank of the kernel consecutive terms fall by e^(-a) per element, a = |u| / h, and the series converges like
        // (a / 2 pi)^(2k): the terms up to f5 reach 1e-15 for a < 0.08 only.  B8 .. B14 (f7 .. f13, Hermite recurrence) carry it
        // to a = 0.65 - relative remainder 2 (a / 2 pi)^16 < 4e-16 - which covers every flank for h >= 60.
        double pa = h5a, pb = h5b, qa = ua2 * (ua2 - 6.0) + 3.0, qb = ub2 * (ub2 - 6.0) + 3.0;   // He_5, He_4
        double ihm = inv_h * ih2 * ih2;                                                          // h^-5
        const double c[4] = {-1.0 / 1209600.0, 1.0 / 47900160.0, -691.0 / 1307674368000.0, 1.0 / 74724249600.0};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int m = 5 + 2 * k;                                   // He_m, He_(m-1) -> He_(m+2), He_(m+1)
            const double ea = ua * pa - (double)m * qa, eb = ub * pb - (double)m * qb;               // He_(m+1)
            const double oa = ua * ea - (double)(m + 1) * pa, ob = ub * eb - (double)(m + 1) * pb;   // He_(m+2)
            pa = oa; pb = ob; qa = ea; qb = eb;
            ihm *= ih2;
            sum += c[k] * ((pa * fa - pb * fb) * ihm);
        }
    }
    return sum;
}

// Tail of a run seen from far away with a narrow kernel: consecutive terms fall by exp(-u/h) per step, so the sum is
// its first few hundred terms at most (the Euler-Maclaurin series converges in u/h and is not used there).
// Terms as scipy forms them: fl(i / h) - fl(x / h).  first / step: nearest element of the run and the direction away from x.
__device__ __forceinline__ double run_sum_tail(uint32_t first, int step, uint32_t len, double inv_h, double xs) {
    double sum = 0.0;
    for (uint32_t k = 0; k < len; ++k) {
        const double u = (double)(first + (uint32_t)(step * (int)k)) * inv_h - xs;
        const double t = exp(-(u * u) / 2);
        sum += t;
        if (t <= 1e-19 * sum) break;                                   // also when everything so far underflowed to zero
    }
    return sum;
}

// scipy subtracts the scaled positions, fl(i / h) - fl(x / h) (gaussian_kernel_estimate: points_[i] - xi_[j]); the rounding of
// x / h is common to every term of an evaluation point and shifts a far, concentrated state's density by up to 1e-11
// relative, so the closed form takes its arguments relative to the same rounded xs: u = fma(i, 1 / h, -xs).
// Partial sum over the runs first, first + stride, ...: the waves of a workgroup share the runs of an evaluation point.
__device__ __forceinline__ double kde_state_runs(const RunDev *__restrict__ runs, uint32_t n_run, uint32_t first, uint32_t stride,
                                                 uint32_t xi, double h, double inv_h) {
    const double short_len = 16.0;                                     // short runs: direct terms
    const double x = (double)xi, xs = x * inv_h;
    double sum = 0.0;
    for (uint32_t r = first; r < n_run; r += stride) {
        const RunDev rn = runs[r];
        const double da = (double)rn.a - x, db = (double)rn.b - x;
        const double near = da >= 0.0 ? da : (db <= 0.0 ? -db : 0.0);  // distance of the nearest run element
        if (near * inv_h > 38.8) continue;                             // exp(-u^2 / 2) is exactly 0.0 from u = 38.6 on: every term of
                                                                       // the run is zero in scipy's sum as well
        if ((double)(rn.b - rn.a) < short_len) {
            for (uint32_t i = rn.a; i <= rn.b; ++i) { const double u = (double)i * inv_h - xs; sum += exp(-(u * u) / 2); }
        } else if (near * inv_h * inv_h > 0.65) {                      // |u| / h > 0.65: steep flank of a narrow kernel (h < 60)
            sum += da >= 0.0 ? run_sum_tail(rn.a, 1, rn.b - rn.a + 1, inv_h, xs) : run_sum_tail(rn.b, -1, rn.b - rn.a + 1, inv_h, xs);
        } else {
            sum += run_sum_em(fma((double)rn.a, inv_h, -xs), fma((double)rn.b, inv_h, -xs), h, inv_h, near * inv_h * inv_h > 0.05);
        }
    }
    return sum;
}

#ifndef PAV_KDE_WAVES                 // waves per evaluation tile (tuning builds override it): 1 / 2 / 4 / 8 waves gave
#define PAV_KDE_WAVES 2               // 0.104 / 0.097 / 0.111 / 0.148 ms per launch - a tile's lifetime is a chain of dependent loads,
#endif                                // and small workgroups let more tiles hide each other's latency
constexpr int KDE_WAVES = PAV_KDE_WAVES;
struct KdeArgs {
    const JobDev *jobs; const JobKde *kde; const EvalTile *tiles; const uint32_t *fill_list;
    const uint32_t *list[3]; double *kern[3]; int8_t *state; const RunDev *runs;      // list: the rows of each state (positions)
    double *ks[3]; int8_t *ss;            // sampled sites, compact: entry samp_off + q of job j = row min(q * srs, n - 1)
    GuardArgs G;
    uint32_t n_tiles;                     // tiles in the list; the workgroups go round them (grid = n_tiles: one each)
    const uint32_t *n_tiles_dev;          // device-planned batches: ... or the count k_plan_fill left on the device
    uint32_t dyn;                         // device-planned batches: the tile counts of the sampled sites come from the job (the
                                          // list holds upper bounds)
};

// One workgroup per tile of 64 evaluation points of one job.  Lane l of every wave stands for point l; the waves share the
// runs of a state (wave w takes runs w, w + KDE_WAVES, ...) and wave 0 adds their partial sums in wave order - a fixed order,
// so the result does not depend on scheduling; a region whose states alternate thousands of times does not hang on one lane.
// States summed term by term in scipy's order (PAV_KDE_DIRECT) stay on wave 0.
#ifndef PAV_KDE_MIN_WAVES             // waves per SIMD asked of the compiler for k_kde_eval (0: no request - 155 registers, three waves)
#define PAV_KDE_MIN_WAVES 0
#endif
#if PAV_KDE_MIN_WAVES
__global__ __launch_bounds__(64 * KDE_WAVES, PAV_KDE_MIN_WAVES) void k_kde_eval(KdeArgs A) {
#else
__global__ __launch_bounds__(64 * KDE_WAVES) void k_kde_eval(KdeArgs A) {
#endif
    __shared__ double part[KDE_WAVES][3][64];
    const uint32_t n_tiles = A.n_tiles_dev ? *A.n_tiles_dev : A.n_tiles;
    for (uint32_t tile_id = blockIdx.x; tile_id < n_tiles; tile_id += gridDim.x) {
    EvalTile t = A.tiles[tile_id];
    const JobKde kd = A.kde[t.job];
    if (A.dyn && t.mode == 0) t.count = kd.finalised && t.first < kd.n_samp ? (kd.n_samp - t.first < 64 ? kd.n_samp - t.first : 64u) : 0u;
    if (t.count == 0) continue;                                        // (uniform) an upper-bound tile behind the job's last site
    const uint64_t off = A.jobs[t.job].tpos_off;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool active = lane < t.count;
    uint32_t x = 0;
    if (active) {
        if (t.mode == 0) {                                             // sampled sites (density.py:211-214)
            const uint64_t xx = ((uint64_t)t.first + lane) * kd.srs;
            x = xx > kd.n - 1 ? kd.n - 1 : (uint32_t)xx;
        } else {
            x = A.fill_list[off + t.first + lane];
        }
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        double v = 0.0;
        const bool by_runs = kd.use_runs && kd.h[s] >= KDE_RUNS_MIN_H;                     // (uniform over the workgroup)
        if (kd.m[s] && by_runs) {
            if (active) v = kde_state_runs(A.runs + kd.run_off[s], kd.n_run[s], wave, KDE_WAVES, x, kd.h[s], kd.inv_h[s]);
        } else if (kd.m[s] && (wave == 0 || kd.use_runs)) {
            // term by term over the data points within KDE_ZERO_R of any point of the wave (bounds from the smallest / largest
            // scaled position among the wave's points).  PAV_KDE_DIRECT (use_runs == 0): wave 0 alone, strictly ascending -
            // scipy's order.  A narrow-bandwidth state of a run-sum job: the eight waves take consecutive slices of the window
            // (the sum is then not in scipy's order; the near-tie guard treats it like a run sum).  One wave walking up to
            // 77 bandwidths + the tile's span of data points (~4 k exp) was the longest workgroup of every launch.
            const double xs = (double)x * kd.inv_h[s];
            double xlo = active ? xs : INFINITY, xhi = active ? xs : -INFINITY;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { xlo = fmin(xlo, __shfl_xor(xlo, d)); xhi = fmax(xhi, __shfl_xor(xhi, d)); }
            const ScaledList ps{A.list[s] + off, kd.inv_h[s]};
            uint32_t lo = __builtin_amdgcn_readfirstlane(ps_lower_bound(ps, kd.m[s], uniform_f64(xlo) - KDE_ZERO_R));
            uint32_t hi = __builtin_amdgcn_readfirstlane(ps_lower_bound(ps, kd.m[s], uniform_f64(xhi) + KDE_ZERO_R));
            if (kd.use_runs) {
                const uint32_t len = hi - lo, w0 = __builtin_amdgcn_readfirstlane(wave);
                hi = lo + (uint32_t)((uint64_t)len * (w0 + 1) / KDE_WAVES);
                lo = lo + (uint32_t)((uint64_t)len * w0 / KDE_WAVES);
            }
            if (active) v = kde_state(ps, lo, hi, xs, kd.norm[s], kd.w[s]);
        }
        part[wave][s][lane] = v;
    }
    __syncthreads();
    if (wave == 0 && active) {
    double val[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        if (kd.m[s] == 0) { val[s] = 0.0; continue; }                  // density.py:84,92,100
        double est = part[0][s][lane];
        if (kd.use_runs) {                                             // partial sums of the eight waves, in wave order
#pragma unroll
            for (int w = 1; w < KDE_WAVES; ++w) est += part[w][s][lane];
            if (kd.h[s] >= KDE_RUNS_MIN_H) est = kd.w[s] * (est * kd.norm[s]);
        }
        val[s] = est * kd.cnt[s];                                      // density.py:110-115
    }
    if (t.mode == 0) {
        // sampled sites go to compact arrays (coalesced here and in k_windows; k_finalize puts them into the table rows)
        const uint64_t so = (uint64_t)kd.samp_off + t.first + lane;
#pragma unroll
        for (int s = 0; s < 3; ++s) A.ks[s][so] = val[s];
        A.ss[so] = (int8_t)argmax3(val[0], val[1], val[2]);            // density.py:250-255
        if (A.G.rel > 0.0 && near_argmax(val[0], val[1], val[2], A.G.rel)) {
            atomicAdd(&A.G.stat[t.job].n_near, 1u);                    // sampled sites are evaluated here in pass 0 only
            if (!kd.all_direct) guard_flag_sample(A.G, t.job, t.first + lane, so);
            else if (near_argmax(val[0], val[1], val[2], A.G.unres)) atomicAdd(&A.G.stat[t.job].n_unres, 1u);
        }
    } else {
#pragma unroll
        for (int s = 0; s < 3; ++s) A.kern[s][off + x] = val[s];
    }
    }
    __syncthreads();                                                   // `part` is written again by the next tile
    }
}

// Windows between consecutive sampled sites (scripts/density.py:257-323), two launches.  k_windows: one lane per window
// decides whether the states or the densities change inside it; if so its inner sites are queued for full evaluation.
// The inner sites of the quiet windows are interpolated by k_finalize (one lane per row, coalesced stores).
__global__ __launch_bounds__(64) void k_windows(const JobDev *__restrict__ jobs, const EvalTile *__restrict__ tiles,
                                                const JobKde *__restrict__ kde, const int8_t *__restrict__ state_mer,
                                                const int8_t *__restrict__ ss, const double *__restrict__ s0,
                                                const double *__restrict__ s1, const double *__restrict__ s2, double delta,
                                                uint32_t *__restrict__ fill_list, uint8_t *__restrict__ win_fill,
                                                JobStat *__restrict__ stat, GuardArgs G) {
    const EvalTile t = tiles[blockIdx.x];                              // the tiles of the sampled sites: one wave per 64 windows
    if (threadIdx.x >= t.count) return;
    const uint32_t j = t.job;
    const JobKde kd = kde[j];
    const uint64_t off = jobs[j].tpos_off;
    const uint64_t q = (uint64_t)t.first + threadIdx.x;
    const uint64_t ap = off + q;
    if (q + 1 >= kd.n_samp) return;
    const uint64_t a = q * kd.srs;
    uint64_t b = (q + 1) * kd.srs;
    if (b > kd.n - 1) b = kd.n - 1;
    if (b == a + 1) { win_fill[ap] = 1; return; }                      // density.py:270-271: nothing in between
    const uint64_t so = (uint64_t)kd.samp_off + q;                     // the two sampled sites of the window, compact arrays
    bool change = ss[so] != ss[so + 1];
    const int8_t sm = state_mer[off + a];
    for (uint64_t i = a + 1; i <= b && !change; ++i) change = state_mer[off + i] != sm;      // :273-275
    const double *kk[3] = {s0, s1, s2};
    double dmax = 0.0;
#pragma unroll
    for (int s = 0; s < 3; ++s) { const double d = fabs(kk[s][so] - kk[s][so + 1]); if (d > dmax) dmax = d; }
    const bool fill = change || dmax > delta;                          // :277-283
    if (G.rel > 0.0 && !change && fabs(dmax - delta) < G.rel * delta) {          // density_change decided by a hair
        if (G.pass == 0) atomicAdd(&stat[j].n_near, 1u);
        const bool e0 = kd.all_direct || (G.samp_flag[so] & SF_EXACT), e1 = kd.all_direct || (G.samp_flag[so + 1] & SF_EXACT);
        if (!e0) guard_flag_sample(G, j, (uint32_t)q, so);
        if (!e1) guard_flag_sample(G, j, (uint32_t)q + 1, so + 1);
        if (e0 && e1 && !(atomicOr(&G.samp_flag[so], SF_WIN_COUNTED) & SF_WIN_COUNTED) && fabs(dmax - delta) < G.unres * delta)
            atomicAdd(&stat[j].n_unres, 1u);
    }
    win_fill[ap] = fill ? 1 : 0;
    if (fill) {
        const uint32_t cnt = (uint32_t)(b - a - 1);
        const uint32_t at = atomicAdd(&stat[j].fill_n, cnt);
        for (uint32_t t = 0; t < cnt; ++t) fill_list[off + at + t] = (uint32_t)(a + 1 + t);
    }
}

// ---- one pass over the table rows: interpolation, spike rule, arg-max, run heads (scripts/density.py:289-338, pavlib/density.py:330-361)
struct FinArgs {
    const JobDev *jobs; const uint32_t *tile_job; const JobKde *kde; const JobStat *stat;
    const uint8_t *win_fill; const double *ks[3]; double *kern[3]; int8_t *state; const uint32_t *index;
    HeadEvent *events; uint32_t ev_cap; uint32_t *ev_count;
    GuardArgs G; uint32_t *blk_spike; int spike_add = 0;
};

// Densities of row x of a finalised job before the spike rule: a sampled site comes from the compact arrays, a row of an
// evaluated window from the table (k_kde_eval mode 1 / k_redo wrote it), a row of a quiet window is interpolated between the
// two sampled sites around it (np.interp: slope * (x - x0) + y0).
__device__ __forceinline__ void row_raw(const FinArgs &A, const JobKde &kd, uint64_t off, uint32_t x, double (&v)[3]) {
    const uint32_t q = x / kd.srs, a = q * kd.srs;
    const uint64_t so = (uint64_t)kd.samp_off + q;
    if (x == a) {
#pragma unroll
        for (int s = 0; s < 3; ++s) v[s] = A.ks[s][so];
        return;
    }
    uint32_t b = a + kd.srs;                                           // x > a, so the window has an end: q + 1 < n_samp
    if (b > kd.n - 1) b = kd.n - 1;
    if (x == b) {                                                      // the last sampled site (n - 1, not a multiple of srs)
#pragma unroll
        for (int s = 0; s < 3; ++s) v[s] = A.ks[s][so + 1];
        return;
    }
    if (A.win_fill[off + q]) {
#pragma unroll
        for (int s = 0; s < 3; ++s) v[s] = A.kern[s][off + x];
        return;
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const double ya = A.ks[s][so], yb = A.ks[s][so + 1];
        const double slope = (yb - ya) / ((double)b - (double)a);
        v[s] = slope * ((double)x - (double)a) + ya;
    }
}
__device__ __forceinline__ int spike_argmax(double (&v)[3]) {         // KERN > 1.0 -> 1 / KERN, then np.argmax (:329-338)
#pragma unroll
    for (int s = 0; s < 3; ++s) if (v[s] > 1.0) v[s] = 1 / v[s];
    return argmax3(v[0], v[1], v[2]);
}

// One lane per table row.  Finalised jobs: the row's densities (row_raw), spike rule, arg-max, the three KERN columns and
// STATE written once; near-tie guard of the arg-max - the sites a doubtful row rests on (itself when it is a sampled site or
// a row of an evaluated window, the two ends of its window when it was interpolated) are queued for evaluation in scipy's
// order unless they already are; blk_spike: rows of the block with a value within G.rel of the spike threshold 1.0 (counted
// only: the branch is continuous, see include/pav_amd.h).  Every job (un-finalised tables have STATE = -1 throughout): one
// event per row that starts a run of equal STATE and one end marker - the input of rl_encoder; the state of the row in front
// of a block comes from re-deriving that row (its own lane may be rewriting its spiked value meanwhile: v and 1 / v give the
// same state).  Failed jobs have no rows.
__global__ __launch_bounds__(256) void k_finalize(FinArgs A) {
    __shared__ uint32_t s_spike[4];
    __shared__ int8_t s_state[256];
    const GuardArgs &G = A.G;
    const uint64_t ap = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t j = A.tile_job[(uint64_t)blockIdx.x * 256 / DTILE];    // a workgroup lies in one tile: job and descriptor through the scalar cache
    const JobKde kd = A.kde[j];
    const uint64_t off = A.jobs[j].tpos_off;
    const uint32_t n = kd.finalised ? kd.n : A.stat[j].n_rows;
    const uint32_t x = (uint32_t)(ap - off);
    const bool live = ap - off < n;
    bool spike_near = false;
    int st = -1;
    if (live && kd.finalised) {
        double v[3];
        row_raw(A, kd, off, x, v);
        if (G.rel > 0.0) spike_near = fabs(v[0] - 1.0) < G.rel || fabs(v[1] - 1.0) < G.rel || fabs(v[2] - 1.0) < G.rel;
        st = spike_argmax(v);
#pragma unroll
        for (int s = 0; s < 3; ++s) A.kern[s][ap] = v[s];
        A.state[ap] = (int8_t)st;
        if (G.rel > 0.0 && near_argmax(v[0], v[1], v[2], G.rel)) {
            if (G.pass == 0) atomicAdd(&G.stat[j].n_near, 1u);
            const uint32_t q = x / kd.srs;
            bool exact;
            if (x == q * kd.srs || x == kd.n - 1) {                    // a sampled site (n - 1 is site q + 1 unless a multiple of srs)
                const uint32_t qs = x == q * kd.srs ? q : q + 1;
                const uint64_t so = (uint64_t)kd.samp_off + qs;
                exact = kd.all_direct || (G.samp_flag[so] & SF_EXACT);
                if (!exact) guard_flag_sample(G, j, qs, so);
            } else if (A.win_fill[off + q]) {                          // evaluated row
                exact = kd.all_direct || (G.row_flag[ap] & RF_EXACT);
                if (!exact) guard_flag_row(G, j, x, ap);
            } else {                                                   // interpolated between the ends of window q
                const uint64_t so = (uint64_t)kd.samp_off + q;
                const bool e0 = kd.all_direct || (G.samp_flag[so] & SF_EXACT), e1 = kd.all_direct || (G.samp_flag[so + 1] & SF_EXACT);
                if (!e0) guard_flag_sample(G, j, q, so);
                if (!e1) guard_flag_sample(G, j, q + 1, so + 1);
                exact = e0 && e1;
            }
            if (exact && !(G.row_flag[ap] & RF_COUNTED)) {
                G.row_flag[ap] |= RF_COUNTED;
                if (near_argmax(v[0], v[1], v[2], G.unres)) atomicAdd(&G.stat[j].n_unres, 1u);
            }
        }
    }
    s_state[threadIdx.x] = (int8_t)st;
    const unsigned long long bal = __ballot(spike_near);
    if ((threadIdx.x & 63) == 0) s_spike[threadIdx.x >> 6] = (uint32_t)__popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0 && (A.blk_spike || A.spike_add)) {
        const uint32_t c = s_spike[0] + s_spike[1] + s_spike[2] + s_spike[3];
        // device-planned batches run this kernel once: the (rare) counts are added where they are read; the host-planned path may
        // repeat it and sums the blocks' plain stores afterwards (k_spike_sum)
        if (A.spike_add) { if (c) atomicAdd(&G.stat[j].n_spike, c); } else A.blk_spike[blockIdx.x] = c;
    }
    if (!live) return;
    // run heads
    bool head = x == 0;
    if (!head) {
        int prev;
        if (threadIdx.x > 0) prev = s_state[threadIdx.x - 1];
        else if (!kd.finalised) prev = -1;
        else { double pv[3]; row_raw(A, kd, off, x - 1, pv); prev = spike_argmax(pv); }
        head = prev != st;
    }
    if (head) {
        const uint32_t e = atomicAdd(A.ev_count, 1u);
        if (e < A.ev_cap) A.events[e] = HeadEvent{j, x, (int32_t)st, A.index[ap], x ? A.index[ap - 1] : 0u, 0u};
    }
    if (x == n - 1) {
        const uint32_t e = atomicAdd(A.ev_count, 1u);
        if (e < A.ev_cap) A.events[e] = HeadEvent{j, n, -2, 0u, A.index[ap], 0u};
    }
}

// ---- k_finalize over a list of live blocks (device-planned batches) -------------------------------------------------------------
// k_finalize above is launched over every 256 positions of the batch's arena.  In a scan round most of them hold no table row - the
// regions that are settled from their counts have none, a region's rows fill only the front of its span - and a workgroup finds that
// out two dependent loads after it has been placed: two fifths of the kernel's waves did nothing, and what the lanes regime runs out
// of is wave slots (the SQ counters of a pass add up to ~15 resident waves per CU on average, profiles/r06_full_path_pmc_sq.txt).
// k_plan, which knows every region's row count, therefore leaves a list of the blocks that HAVE rows; FIN_R rows per lane, their
// loads issued together (sampled sites, window flag, then the table rows of evaluated windows), so that a wave's dependent round
// trips are paid once per FIN_R * 64 rows; a fixed grid goes round the list.  Same arithmetic, same events as k_finalize.
constexpr int FIN_R = PAV_FIN_R;
constexpr uint32_t FIN_ROWS = 256u * FIN_R;
struct FinBlock { uint32_t job, row0; };

__global__ __launch_bounds__(256) void k_finalize_blocks(FinArgs A, const FinBlock *__restrict__ blocks, const uint32_t *__restrict__ n_blocks) {
    __shared__ uint32_t s_spike[4];
    __shared__ int8_t s_state[FIN_ROWS];
    const GuardArgs &G = A.G;
    const uint32_t nb = *n_blocks;
    for (uint32_t e = blockIdx.x; e < nb; e += gridDim.x) {
        const FinBlock fb = blocks[e];
        const uint32_t j = fb.job;
        const JobKde kd = A.kde[j];
        const uint64_t off = A.jobs[j].tpos_off;
        const uint32_t n = kd.finalised ? kd.n : A.stat[j].n_rows;
        uint32_t x[FIN_R]; bool live[FIN_R]; int st[FIN_R];
        uint32_t spike_cnt = 0;                                        // rows of this lane within G.rel of the spike threshold
#pragma unroll
        for (int r = 0; r < FIN_R; ++r) { x[r] = fb.row0 + (uint32_t)r * 256u + threadIdx.x; live[r] = x[r] < n; st[r] = -1; }
        if (kd.finalised) {
            // phase 1: what every row needs - the two sampled sites around it and its window's flag (rows past the end: the last row's)
            uint32_t q[FIN_R], a[FIN_R], b[FIN_R]; bool at_a[FIN_R], at_b[FIN_R];
            double ya[FIN_R][3], yb[FIN_R][3]; uint8_t wf[FIN_R];
#pragma unroll
            for (int r = 0; r < FIN_R; ++r) {
                const uint32_t xx = live[r] ? x[r] : n - 1u;
                q[r] = xx / kd.srs; a[r] = q[r] * kd.srs;
                b[r] = a[r] + kd.srs; if (b[r] > kd.n - 1u) b[r] = kd.n - 1u;
                at_a[r] = xx == a[r]; at_b[r] = !at_a[r] && xx == b[r];
                const uint64_t so = (uint64_t)kd.samp_off + q[r];
#pragma unroll
                for (int s = 0; s < 3; ++s) { ya[r][s] = A.ks[s][so]; yb[r][s] = A.ks[s][so + 1]; }   // (so + 1: inside the arrays also behind a job's last site)
                wf[r] = A.win_fill[off + q[r]];
            }
            // phase 2: the table rows of evaluated windows (the others ask for the job's first row: one line a wave)
            double kv[FIN_R][3];
#pragma unroll
            for (int r = 0; r < FIN_R; ++r) {
                const bool need = live[r] && !at_a[r] && !at_b[r] && wf[r];
                const uint64_t at = need ? off + x[r] : off;
#pragma unroll
                for (int s = 0; s < 3; ++s) kv[r][s] = A.kern[s][at];
                wf[r] = need ? 1 : 0;
            }
#pragma unroll
            for (int r = 0; r < FIN_R; ++r) {
                if (!live[r]) continue;
                const uint64_t ap = off + x[r];
                double v[3];
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    if (at_a[r]) v[s] = ya[r][s];
                    else if (at_b[r]) v[s] = yb[r][s];
                    else if (wf[r]) v[s] = kv[r][s];
                    else {                                             // np.interp (row_raw above: the same expression)
                        const double slope = (yb[r][s] - ya[r][s]) / ((double)b[r] - (double)a[r]);
                        v[s] = slope * ((double)x[r] - (double)a[r]) + ya[r][s];
                    }
                }
                if (G.rel > 0.0 && (fabs(v[0] - 1.0) < G.rel || fabs(v[1] - 1.0) < G.rel || fabs(v[2] - 1.0) < G.rel)) ++spike_cnt;
                st[r] = spike_argmax(v);
#pragma unroll
                for (int s = 0; s < 3; ++s) A.kern[s][ap] = v[s];
                A.state[ap] = (int8_t)st[r];
                if (G.rel > 0.0 && near_argmax(v[0], v[1], v[2], G.rel)) {          // (rare: as in k_finalize)
                    if (G.pass == 0) atomicAdd(&G.stat[j].n_near, 1u);
                    bool exact;
                    if (at_a[r] || x[r] == kd.n - 1u) {
                        const uint32_t qs = at_a[r] ? q[r] : q[r] + 1u;
                        const uint64_t so = (uint64_t)kd.samp_off + qs;
                        exact = kd.all_direct || (G.samp_flag[so] & SF_EXACT);
                        if (!exact) guard_flag_sample(G, j, qs, so);
                    } else if (A.win_fill[off + q[r]]) {
                        exact = kd.all_direct || (G.row_flag[ap] & RF_EXACT);
                        if (!exact) guard_flag_row(G, j, x[r], ap);
                    } else {
                        const uint64_t so = (uint64_t)kd.samp_off + q[r];
                        const bool e0 = kd.all_direct || (G.samp_flag[so] & SF_EXACT), e1 = kd.all_direct || (G.samp_flag[so + 1] & SF_EXACT);
                        if (!e0) guard_flag_sample(G, j, q[r], so);
                        if (!e1) guard_flag_sample(G, j, q[r] + 1u, so + 1);
                        exact = e0 && e1;
                    }
                    if (exact && !(G.row_flag[ap] & RF_COUNTED)) {
                        G.row_flag[ap] |= RF_COUNTED;
                        if (near_argmax(v[0], v[1], v[2], G.unres)) atomicAdd(&G.stat[j].n_unres, 1u);
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < FIN_R; ++r) s_state[r * 256 + threadIdx.x] = (int8_t)st[r];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) spike_cnt += (uint32_t)__shfl_xor((int)spike_cnt, d);
        if ((threadIdx.x & 63) == 0) s_spike[threadIdx.x >> 6] = spike_cnt;
        __syncthreads();
        if (threadIdx.x == 0 && A.spike_add) {
            const uint32_t c = s_spike[0] + s_spike[1] + s_spike[2] + s_spike[3];
            if (c) atomicAdd(&G.stat[j].n_spike, c);
        }
        // run heads
#pragma unroll
        for (int r = 0; r < FIN_R; ++r) {
            if (!live[r]) continue;
            const uint64_t ap = off + x[r];
            const uint32_t i = (uint32_t)r * 256u + threadIdx.x;
            bool head = x[r] == 0;
            if (!head) {
                int prev;
                if (i > 0) prev = s_state[i - 1];
                else if (!kd.finalised) prev = -1;
                else { double pv[3]; row_raw(A, kd, off, x[r] - 1, pv); prev = spike_argmax(pv); }
                head = prev != st[r];
            }
            if (head) {
                const uint32_t ev = atomicAdd(A.ev_count, 1u);
                if (ev < A.ev_cap) A.events[ev] = HeadEvent{j, x[r], (int32_t)st[r], A.index[ap], x[r] ? A.index[ap - 1] : 0u, 0u};
            }
            if (x[r] == n - 1) {
                const uint32_t ev = atomicAdd(A.ev_count, 1u);
                if (ev < A.ev_cap) A.events[ev] = HeadEvent{j, n, -2, 0u, A.index[ap], 0u};
            }
        }
        __syncthreads();                                                // s_state is the next block's
    }
}

// Per job: sum of its blocks' spike counts (a plain store: the same value whenever k_finalize is repeated).
__global__ __launch_bounds__(256) void k_spike_sum(const JobDev *__restrict__ jobs, const JobKde *__restrict__ kde,
                                                   const uint32_t *__restrict__ blk_spike, JobStat *__restrict__ stat) {
    __shared__ uint32_t red[4];
    const uint32_t j = blockIdx.x;
    const JobKde kd = kde[j];
    uint32_t v = 0;
    if (kd.finalised) {
        const uint64_t b0 = jobs[j].tpos_off / 256, nb = ((uint64_t)kd.n + 255) / 256;
        for (uint64_t b = threadIdx.x; b < nb; b += 256) v += blk_spike[b0 + b];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) stat[j].n_spike = red[0] + red[1] + red[2] + red[3];
}

__global__ void k_reset_fill(JobStat *__restrict__ stat, uint32_t n_jobs) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n_jobs) stat[j].fill_n = 0;
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// One wave per queued site: its three densities with one exp() per data point, accumulated in ascending data order like
// scipy's gaussian_kernel_estimate (the lanes compute 64 terms at a time, the sum takes them one after the other: the same
// additions in the same order as kde_state()).  kinds: bit 0 sampled sites of the list slice, bit 1 rows.
struct RedoArgs {
    const JobDev *jobs; const JobKde *kde; const uint32_t *list[3]; double *ks[3]; int8_t *ss; double *kern[3];
    const uint8_t *win_fill; GuardArgs G; uint32_t first, kinds;
};
__global__ __launch_bounds__(64) void k_redo(RedoArgs A) {
    const unsigned long long e = A.G.entries[A.first + blockIdx.x];
    const uint32_t j = (uint32_t)(e >> 32), idx = (uint32_t)(e & 0x7FFFFFFFull);
    const bool is_row = (e & GE_ROW) != 0;
    if (!((A.kinds >> (is_row ? 1 : 0)) & 1u)) return;
    const JobKde kd = A.kde[j];
    const uint64_t off = A.jobs[j].tpos_off;
    uint32_t x;
    uint64_t so = 0;
    if (is_row) {
        x = idx;
        if (!A.win_fill[off + x / kd.srs]) return;                    // interpolated now: k_finalize looks at the window's ends
    } else {
        so = (uint64_t)kd.samp_off + idx;
        if (A.G.samp_flag[so] & SF_EXACT) return;
        const uint64_t xx = (uint64_t)idx * kd.srs;
        x = xx > kd.n - 1 ? kd.n - 1 : (uint32_t)xx;
    }
    const int lane = threadIdx.x;
    double val[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const uint32_t m = kd.m[s];
        if (m == 0) { val[s] = 0.0; continue; }
        const double inv_h = kd.inv_h[s], norm = kd.norm[s], w = kd.w[s];
        const double xs = (double)x * inv_h;
        const uint32_t *ls = A.list[s] + off;
        double est = 0.0;
        // data points further than KDE_ZERO_R add exact zeros: start at the first one that can contribute, stop behind the last
        uint32_t lo = 0, hi = m;
        {
            uint32_t a = 0, b = m;
            while (a < b) { const uint32_t mid = (a + b) >> 1; if ((double)ls[mid] * inv_h < xs - KDE_ZERO_R) a = mid + 1; else b = mid; }
            lo = a; b = m;
            while (a < b) { const uint32_t mid = (a + b) >> 1; if ((double)ls[mid] * inv_h <= xs + KDE_ZERO_R) a = mid + 1; else b = mid; }
            hi = a;
        }
        for (uint32_t i0 = lo; i0 < hi; i0 += 64) {
            const uint32_t i = i0 + lane;
            double t = 0.0;
            if (i < hi) {
                const double r = (double)ls[i] * inv_h - xs;          // points_[i] - xi_[j], points_ = data * (1 / h)
                t = w * (exp(-(r * r) / 2) * norm);
            }
            const int cnt = (int)min(64u, hi - i0);
            for (int l = 0; l < cnt; ++l) est += readlane_f64(t, l);
        }
        val[s] = est * kd.cnt[s];
    }
    if (lane != 0) return;
    if (is_row) {
        const uint64_t ap = off + x;
#pragma unroll
        for (int s = 0; s < 3; ++s) A.kern[s][ap] = val[s];
        const uint8_t old = A.G.row_flag[ap];
        if (!(old & RF_EXACT)) atomicAdd(&A.G.stat[j].n_reeval, 1u);
        A.G.row_flag[ap] = (uint8_t)((old & ~RF_PENDING) | RF_EXACT);
    } else {
#pragma unroll
        for (int s = 0; s < 3; ++s) A.ks[s][so] = val[s];
        A.ss[so] = (int8_t)argmax3(val[0], val[1], val[2]);
        A.G.samp_flag[so] = (A.G.samp_flag[so] & ~SF_PENDING) | SF_EXACT;
        atomicAdd(&A.G.stat[j].n_reeval, 1u);
        if (near_argmax(val[0], val[1], val[2], A.G.unres)) atomicAdd(&A.G.stat[j].n_unres, 1u);
    }
}

// ---- the decisions of a scan round, on the device -----------------------------------------------------------------------
// pavlib/inv.py:297-351 decides from the run-length encoding of STATE what becomes of a region - finished, expanded (:309-342), or
// flanked by FWD runs and characterised (:353-406) - and the two latter need lifts through the alignment table: the four breakpoint
// positions of a flanked region, the two ends of an expanded one.  The scan driver (invscan.cpp) took those decisions on the host and
// then asked the device for the lifts: one more round trip per round.  Here a wave per job gathers the job's run heads from the event
// list k_finalize has just written, orders them by row, applies the same rules and writes the job's queries (four slots per job,
// axis -1 = empty); k_lift_points answers them in the same launch set and queries and answers travel with the batch's one read-back.
// The driver still takes its decisions itself (it writes the reference's log text from them) and uses an answer only when the query
// beside it IS the query it would have asked: the lift is a pure function of the query, so nothing here has to be trusted.
constexpr uint32_t ROUND_MAX_HEADS = 512;
__global__ __launch_bounds__(64) void k_round_decide(const JobStat *__restrict__ stat, const HeadEvent *__restrict__ ev, const uint32_t *__restrict__ ev_count,
                                                     uint32_t ev_cap, const RoundJobIn *__restrict__ in, LiftQuery *__restrict__ q, uint32_t n_jobs,
                                                     uint32_t min_state_count, uint32_t min_informative, uint32_t max_ref_kmer_count, int min_exp_count, int k) {
    __shared__ HeadEvent s_ev[ROUND_MAX_HEADS];
    __shared__ uint16_t s_order[ROUND_MAX_HEADS];
    __shared__ uint32_t s_n;
    const uint32_t j = blockIdx.x, lane = threadIdx.x;
    if (j >= n_jobs) return;
    if (lane < 4) { LiftQuery e{}; e.axis = -1; q[4ull * j + lane] = e; }
    const JobStat S = stat[j];
    const RoundJobIn I = in[j];
    const bool failed = S.n_ref_valid == 0 || S.max_count > max_ref_kmer_count;
    const bool no_table = fwd_only(S, min_state_count);                  // (the driver's batches are scan-only: such a region has no rows)
    const uint32_t n_rows = no_table ? (S.st_count[0] >= min_state_count ? S.st_count[0] : 0u) : S.n_rows;
    if (failed || n_rows == 0) return;                                   // inv.py:268-296: the region is finished
    // what the rules look at: number of runs, the states at both ends, rl[1].pos, rl[-2].end, the REV runs
    uint32_t n_runs = 0, max_inv = 0;
    int first_state = 0, last_state = 0;
    long long r1_pos = 0, rn2_end = 0, inv_first_pos = 0, inv_last_end = 0;
    bool any_inv = false;
    if (no_table) { n_runs = n_rows >= min_informative ? 1u : 0u; }
    else {
        const uint32_t n_all = *ev_count;
        if (n_all > ev_cap) return;                                      // (the batch goes to the host-planned path)
        if (lane == 0) s_n = 0;
        __syncthreads();
        for (uint32_t e = lane; e < n_all; e += 64)
            if (ev[e].job == j) { const uint32_t at = atomicAdd(&s_n, 1u); if (at < ROUND_MAX_HEADS) s_ev[at] = ev[e]; }
        __syncthreads();
        const uint32_t n = s_n;
        if (n > ROUND_MAX_HEADS) return;                                 // too many runs for this shortcut: the driver asks for its lifts itself
        for (uint32_t a = lane; a < n; a += 64) {                        // order by row (rank sort; rows of a job's heads are distinct)
            uint32_t rank = 0;
            const uint32_t ra = s_ev[a].row;
            for (uint32_t b = 0; b < n; ++b) rank += (s_ev[b].row < ra || (s_ev[b].row == ra && b < a)) ? 1u : 0u;
            s_order[rank] = (uint16_t)a;
        }
        __syncthreads();
        if (lane != 0) return;
        long long prev_end = 0;
        for (uint32_t e = 0; e + 1 < n; ++e) {                           // a run = a head and the head behind it (the last one ends the rows)
            const HeadEvent h = s_ev[s_order[e]], nx = s_ev[s_order[e + 1]];
            if (h.state == -2) continue;
            const long long pos = h.index, end = nx.prev_index;
            const uint32_t count = nx.row - h.row;
            if (n_runs == 0) first_state = h.state;
            if (n_runs == 1) r1_pos = pos;
            rn2_end = prev_end; prev_end = end;                          // when the loop ends: the end of the run before the last
            last_state = h.state;
            if (h.state == 2) {
                if (!any_inv) inv_first_pos = pos;
                inv_last_end = end; any_inv = true;
                max_inv = count > max_inv ? count : max_inv;
            }
            ++n_runs;
        }
    }
    if (lane != 0) return;
    if (n_runs == 1 && (first_state == 0 || first_state == -1) && I.expansion_count >= min_exp_count) return;     // inv.py:297-307
    LiftQuery out[4];
    for (int t = 0; t < 4; ++t) { out[t] = LiftQuery{}; out[t].axis = -1; }
    if (n_runs > 2 && first_state == 0 && last_state == 0) {             // flanked: inv.py:353-406
        if (!any_inv || max_inv < 100) return;
        long long o_pos = r1_pos + I.tig_pos, o_end = rn2_end + I.tig_pos + k;
        long long i_pos = inv_first_pos + I.tig_pos, i_end = inv_last_end + I.tig_pos + k;
        if (o_pos > o_end) { const long long t = o_pos; o_pos = o_end; o_end = t; }
        if (i_pos > i_end) { const long long t = i_pos; i_pos = i_end; i_end = t; }
        out[0] = LiftQuery{1, I.tig_chrom, 0, 0, o_pos}; out[1] = LiftQuery{1, I.tig_chrom, 0, 0, o_end};
        out[2] = LiftQuery{1, I.tig_chrom, 1, 0, i_pos}; out[3] = LiftQuery{1, I.tig_chrom, 1, 0, i_end};
    } else {                                                             // expand: inv.py:309-342, Region.expand (seq.py:112-188)
        const long long last_len = I.ref_end - I.ref_pos;
        const long long expand_bp = (long long)(int)((double)last_len * 1.5);
        double balance = 0.5;
        if (n_runs > 2) { if (first_state == 0) balance = 0.25; else if (last_state == 0) balance = 0.75; }
        const long long expand_pos = (long long)((double)expand_bp * balance);
        const long long expand_end = expand_bp - expand_pos > 0 ? expand_bp - expand_pos : 0;
        long long new_pos = I.ref_pos - expand_pos, new_end = I.ref_end + expand_end;
        if (new_pos < 0) { new_end += -new_pos; new_pos = 0; }
        if (new_end > I.chrom_len) { new_pos -= new_end - I.chrom_len; if (new_pos < 0) new_pos = 0; new_end = I.chrom_len; }
        if (new_end < new_pos) new_end = new_pos = (new_end + new_pos) / 2;
        if (new_end - new_pos == last_len) return;                       // reached the reference's limits
        out[0] = LiftQuery{0, I.ref_chrom, 0, 0, new_pos}; out[1] = LiftQuery{0, I.ref_chrom, 0, 0, new_end};
    }
    for (int t = 0; t < 4; ++t) q[4ull * j + t] = out[t];
}


// Run heads for rl_encoder (pavlib/density.py:330-361): one event per row that starts a run, plus one end marker.
__global__ __launch_bounds__(256) void k_heads(const JobDev *__restrict__ jobs, const uint32_t *__restrict__ tile_job,
                                               const JobStat *__restrict__ stat, const int8_t *__restrict__ state,
                                               const uint32_t *__restrict__ index, HeadEvent *__restrict__ events,
                                               uint32_t cap, uint32_t *__restrict__ ev_count) {
    const uint64_t ap = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t j = tile_job[(uint64_t)blockIdx.x * 256 / DTILE];
    const uint32_t n = stat[j].n_rows;
    const uint64_t r = ap - jobs[j].tpos_off;
    if (r >= n) return;
    const bool head = r == 0 || state[ap] != state[ap - 1];
    if (head) {
        const uint32_t e = atomicAdd(ev_count, 1u);
        if (e < cap) events[e] = HeadEvent{j, (uint32_t)r, (int32_t)state[ap], index[ap], r ? index[ap - 1] : 0u, 0u};
    }
    if (r == n - 1) {
        const uint32_t e = atomicAdd(ev_count, 1u);
        if (e < cap) events[e] = HeadEvent{j, n, -2, 0u, index[ap], 0u};
    }
}

// ---- annotate_inv_dup_mers (pavlib/inv.py:457-561) --------------------------------------------------------------
__global__ __launch_bounds__(256) void k_canon_insert(SeqView R, uint64_t abs0, uint32_t len, int k,
                                                      unsigned long long *__restrict__ keys, uint32_t hmask) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i + (uint64_t)k > len) return;
    uint64_t x;
    if (!kmer_window(R.two, R.mask, abs0 + i, k, x)) return;
    const uint64_t f = rev_groups(x, k), r = x ^ kmer_mask(k);
    table_insert(keys, 0, hmask, f <= r ? f : r);                      // canonical_complement = numeric min
}

__global__ __launch_bounds__(256) void k_annotate(const uint32_t *__restrict__ index, const unsigned long long *__restrict__ kmer,
                                                  uint64_t off, uint32_t n, int k, int64_t base, int64_t up_pos, int64_t up_end,
                                                  int64_t dn_pos, int64_t dn_end, const unsigned long long *__restrict__ keys_up,
                                                  uint32_t mask_up, const unsigned long long *__restrict__ keys_dn,
                                                  uint32_t mask_dn, uint8_t *__restrict__ flank, uint8_t *__restrict__ match) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t q = (int64_t)index[off + i] + base;                 // QRY_INDEX (inv.py:519)
    uint8_t f = 0;
    if (q >= up_pos && q < up_end - k) f = 1;                          // inv.py:524-527
    if (q >= dn_pos && q < dn_end - k) f = 2;                          // inv.py:529-532
    uint8_t m = 0;
    if (f) {
        const uint64_t km = kmer[off + i];                             // raw KMER against canonical sets (inv.py:537-553)
        const bool in_up = table_has(keys_up, 0, mask_up, km), in_dn = table_has(keys_dn, 0, mask_dn, km);
        const bool same = f == 1 ? in_up : in_dn, other = f == 1 ? in_dn : in_up;
        m = same ? (other ? 3 : 1) : (other ? 2 : 3);                  // KMER_LOC_STATE: NA / OTHER / SAME / NA
    }
    flank[i] = f;
    match[i] = m;
}

static uint32_t pow2_at_least(uint64_t n) {
    uint64_t c = 64;
    while (c < n) c <<= 1;
    return (uint32_t)c;
}

static DensityState *dstate(pav_ctx *ctx) {
    if (!ctx->density) ctx->density = new DensityState();
    return static_cast<DensityState *>(ctx->density);
}

// ---- call tables of a round: flank k-mer sets, FLANK / MATCH and the packed columns, three launches in all -------------
struct CanonJob { uint64_t abs0, key_off; uint32_t len, hmask, tile0, pad; };      // one flank of one call
__global__ __launch_bounds__(256) void k_canon_insert_batch(SeqView R, const CanonJob *__restrict__ jobs, uint32_t n_jobs, int k,
                                                            unsigned long long *__restrict__ keys) {
    uint32_t lo = 0, hi = n_jobs;                        // last job with tile0 <= blockIdx.x
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (jobs[mid].tile0 <= blockIdx.x) lo = mid; else hi = mid; }
    const CanonJob j = jobs[lo];
    const uint64_t i = (uint64_t)(blockIdx.x - j.tile0) * 256 + threadIdx.x;
    if (i + (uint64_t)k > j.len) return;
    uint64_t x;
    if (!kmer_window(R.two, R.mask, j.abs0 + i, k, x)) return;
    const uint64_t f = rev_groups(x, k), r = x ^ kmer_mask(k);
    table_insert(keys + j.key_off, 0, j.hmask, f <= r ? f : r);
}

// Pack the table rows of all calls of a round into contiguous columns and annotate them (annotate_inv_dup_mers,
// pavlib/inv.py:457-561), so that the host copy is one large transfer per round.
struct GatherCall {
    uint64_t src_off, key_up, key_dn;
    int64_t base, up_pos, up_end, dn_pos, dn_end;
    uint32_t dst_off, n, mask_up, mask_dn;
};
__global__ __launch_bounds__(256) void k_gather_calls(const GatherCall *__restrict__ calls, uint32_t n_calls, uint32_t total, int k,
                                                      const uint32_t *__restrict__ index, const int8_t *__restrict__ state_mer,
                                                      const int8_t *__restrict__ state, const double *__restrict__ k0,
                                                      const double *__restrict__ k1, const double *__restrict__ k2,
                                                      const unsigned long long *__restrict__ kmer,
                                                      const unsigned long long *__restrict__ keys, uint32_t *__restrict__ o_index,
                                                      int8_t *__restrict__ o_sm, int8_t *__restrict__ o_st, double *__restrict__ o_k0,
                                                      double *__restrict__ o_k1, double *__restrict__ o_k2,
                                                      unsigned long long *__restrict__ o_kmer, uint8_t *__restrict__ o_flank,
                                                      uint8_t *__restrict__ o_match) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    // last call with dst_off <= t.  The calls of the workgroup's first and last row are found with uniform arguments (scalar
    // loads); nearly always they are the same call and every lane takes its 80-byte descriptor through the scalar cache
    const uint32_t t_first = blockIdx.x * 256, t_last = min(t_first + 255u, total - 1);
    uint32_t lo = 0, hi = n_calls, lo_last = 0;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (calls[mid].dst_off <= t_first) lo = mid; else hi = mid; }
    hi = n_calls; lo_last = lo;
    while (hi - lo_last > 1) { const uint32_t mid = (lo_last + hi) >> 1; if (calls[mid].dst_off <= t_last) lo_last = mid; else hi = mid; }
    auto row = [&](const GatherCall &c) __attribute__((always_inline)) {
        const uint64_t src = c.src_off + (t - c.dst_off);
        const uint32_t ix = index[src];
        const unsigned long long km = kmer[src];
        o_index[t] = ix; o_sm[t] = state_mer[src]; o_st[t] = state[src];
        o_k0[t] = k0[src]; o_k1[t] = k1[src]; o_k2[t] = k2[src]; o_kmer[t] = km;
        const int64_t q = (int64_t)ix + c.base;                            // QRY_INDEX (inv.py:519)
        uint8_t f = 0;
        if (q >= c.up_pos && q < c.up_end - k) f = 1;                      // inv.py:524-527
        if (q >= c.dn_pos && q < c.dn_end - k) f = 2;                      // inv.py:529-532
        uint8_t m = 0;
        if (f) {                                                           // raw KMER against canonical sets (inv.py:537-553)
            const bool in_up = table_has(keys + c.key_up, 0, c.mask_up, km), in_dn = table_has(keys + c.key_dn, 0, c.mask_dn, km);
            const bool same = f == 1 ? in_up : in_dn, other = f == 1 ? in_dn : in_up;
            m = same ? (other ? 3 : 1) : (other ? 2 : 3);                  // KMER_LOC_STATE: NA / OTHER / SAME / NA
        }
        o_flank[t] = f;
        o_match[t] = m;
    };
    if (lo_last == lo) {
        row(calls[lo]);
    } else {                                             // a call boundary inside the workgroup: every lane searches for itself
        hi = lo_last + 1;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (calls[mid].dst_off <= t) lo = mid; else hi = mid; }
        row(calls[lo]);
    }
}

// FLANK / MATCH of the calls' rows where they are (a round whose column block goes to the stage as a whole): what k_gather_calls
// does, without the copy.  A lane takes four consecutive rows of the block (one 16-byte load of INDEX, two 4-byte stores);
// `calls` is sorted by src_off here, rows of the block that belong to no call are left alone.
__global__ __launch_bounds__(256) void k_annotate_calls(const GatherCall *__restrict__ calls, uint32_t n_calls, uint64_t block_rows, int k,
                                                        const uint32_t *__restrict__ index, const unsigned long long *__restrict__ kmer,
                                                        const unsigned long long *__restrict__ keys, uint8_t *__restrict__ flank,
                                                        uint8_t *__restrict__ match) {
    const uint64_t p0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (p0 >= block_rows) return;
    // the call of the workgroup's first and last row with uniform arguments (scalar loads); nearly always the same call
    const uint64_t w_first = (uint64_t)blockIdx.x * 1024, w_last = min(w_first + 1023, block_rows - 1);
    uint32_t lo = 0, hi = n_calls, lo_last = 0;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (calls[mid].src_off <= w_first) lo = mid; else hi = mid; }
    hi = n_calls; lo_last = lo;
    while (hi - lo_last > 1) { const uint32_t mid = (lo_last + hi) >> 1; if (calls[mid].src_off <= w_last) lo_last = mid; else hi = mid; }
    auto one = [&](const GatherCall &c, uint32_t ix, uint64_t src, uint8_t &f, uint8_t &m) __attribute__((always_inline)) {
        const int64_t q = (int64_t)ix + c.base;                            // QRY_INDEX (inv.py:519)
        f = 0; m = 0;
        if (q >= c.up_pos && q < c.up_end - k) f = 1;                      // inv.py:524-527
        if (q >= c.dn_pos && q < c.dn_end - k) f = 2;                      // inv.py:529-532
        if (f) {                                                           // raw KMER against canonical sets (inv.py:537-553)
            const unsigned long long km = kmer[src];
            const bool in_up = table_has(keys + c.key_up, 0, c.mask_up, km), in_dn = table_has(keys + c.key_dn, 0, c.mask_dn, km);
            const bool same = f == 1 ? in_up : in_dn, other = f == 1 ? in_dn : in_up;
            m = same ? (other ? 3 : 1) : (other ? 2 : 3);                  // KMER_LOC_STATE: NA / OTHER / SAME / NA
        }
    };
    if (lo_last == lo) {
        const GatherCall &c = calls[lo];
        if (p0 >= c.src_off && p0 + 3 < c.src_off + c.n) {                 // four rows of one call: the common case
            const uint4 ix = *reinterpret_cast<const uint4 *>(index + p0);
            uint8_t f[4], m[4];
            one(c, ix.x, p0, f[0], m[0]); one(c, ix.y, p0 + 1, f[1], m[1]); one(c, ix.z, p0 + 2, f[2], m[2]); one(c, ix.w, p0 + 3, f[3], m[3]);
            *reinterpret_cast<uint32_t *>(flank + p0) = (uint32_t)f[0] | (uint32_t)f[1] << 8 | (uint32_t)f[2] << 16 | (uint32_t)f[3] << 24;
            *reinterpret_cast<uint32_t *>(match + p0) = (uint32_t)m[0] | (uint32_t)m[1] << 8 | (uint32_t)m[2] << 16 | (uint32_t)m[3] << 24;
            return;
        }
    }
    for (int r = 0; r < 4; ++r) {                                          // an edge of a call, or a boundary inside the workgroup
        const uint64_t p = p0 + r;
        if (p >= block_rows) break;
        uint32_t a = lo; hi = lo_last + 1;
        while (hi - a > 1) { const uint32_t mid = (a + hi) >> 1; if (calls[mid].src_off <= p) a = mid; else hi = mid; }
        const GatherCall &c = calls[a];
        if (p < c.src_off || p >= c.src_off + c.n) continue;
        uint8_t f, m;
        one(c, index[p], p, f, m);
        flank[p] = f; match[p] = m;
    }
}

int density_fetch_calls(pav_ctx *ctx, const std::vector<CallFetch> &calls, uint64_t k1_rows, CallStage &stage) {
    stage.n_copies = 0;
    stage.row0.clear();
    stage.rows = 0;
    if (calls.empty()) return PAV_OK;
    DensityState *D = dstate(ctx);
    if (!D->valid) return fail(ctx, PAV_E_STATE, "density_fetch_calls: no batch resident");
    const SeqStore &RS = ctx->seq[PAV_ROLE_REF];
    const SeqView RV = RS.view();
    const int k = D->params.k;
    hipStream_t st = ctx->stream;
    std::vector<GatherCall> gc(calls.size());
    std::vector<CanonJob> cj;
    uint64_t keys = 0, total = 0;
    uint32_t tiles = 0;
    for (size_t c = 0; c < calls.size(); ++c) {
        const CallFetch &f = calls[c];
        uint64_t a = f.ref_up_pos, b = f.ref_up_end, x = f.ref_dn_pos, y = f.ref_dn_end;
        if (a > b) std::swap(a, b);                                   // Region() swaps reversed coordinates
        if (x > y) std::swap(x, y);
        if (f.ref_id >= RS.n || b > RS.len[f.ref_id] || y > RS.len[f.ref_id]) return fail(ctx, PAV_E_ARG, "density_fetch_calls: region outside the reference record");
        const uint32_t up_len = (uint32_t)(b - a), dn_len = (uint32_t)(y - x);
        const uint32_t cap_up = pow2_at_least(2ull * up_len + 2), cap_dn = pow2_at_least(2ull * dn_len + 2);
        GatherCall &g = gc[c];
        g.src_off = D->h_jobs[f.job].tpos_off; g.dst_off = (uint32_t)total; g.n = f.n;
        g.base = f.base; g.up_pos = f.tig_up_pos; g.up_end = f.tig_up_end; g.dn_pos = f.tig_dn_pos; g.dn_end = f.tig_dn_end;
        g.key_up = keys; g.key_dn = keys + cap_up; g.mask_up = cap_up - 1; g.mask_dn = cap_dn - 1;
        if (up_len >= (uint32_t)k) { cj.push_back(CanonJob{RS.off[f.ref_id] + a, g.key_up, up_len, g.mask_up, tiles, 0}); tiles += (up_len + 255) / 256; }
        if (dn_len >= (uint32_t)k) { cj.push_back(CanonJob{RS.off[f.ref_id] + x, g.key_dn, dn_len, g.mask_dn, tiles, 0}); tiles += (dn_len + 255) / 256; }
        keys += (uint64_t)cap_up + cap_dn;
        total += f.n;
    }
    // A round whose calls are most of the batch (the second round of a haplotype: 46 regions grown to ~200 kbp, all of them calls)
    // keeps its tables where the batch wrote them: the batch's column block changes hands with the stage's (the one a scan before
    // the last handed over), and only FLANK / MATCH are computed.  Packing such a round read and wrote 80 bytes per row - 0.15 ms,
    // the largest kernel of the scan after the k-mer sets.  PAV_CALL_GATHER=1: always pack.
    const uint64_t a_t = D->block_rows;
    const bool dense = total * 2 >= a_t && getenv("PAV_CALL_GATHER") == nullptr;
    // device staging, owned by the caller for as long as the tables may be asked for: aux = [hash keys][descriptors],
    // buf = the columns in the host block's order
    const uint64_t col_bytes = (dense ? a_t : total) * 40;
    const uint64_t desc_bytes = sizeof(GatherCall) * gc.size() + sizeof(CanonJob) * cj.size();
    if (!D->gathered) PAV_HIP(ctx, hipEventCreateWithFlags(&D->gathered, hipEventDisableTiming));
    PAV_HIP(ctx, stage.aux.reserve(8ull * keys + desc_bytes + 512));
    unsigned long long *d_keys = stage.aux.as<unsigned long long>();
    GatherCall *d_gc = reinterpret_cast<GatherCall *>(d_keys + keys);
    CanonJob *d_cj = reinterpret_cast<CanonJob *>(d_gc + gc.size());
    std::vector<uint8_t> &desc_host = stage.desc_host;                 // stays alive until the upload has run
    desc_host.resize(desc_bytes);
    if (dense) {                                                       // k_annotate_calls walks the block: the descriptors by first row
        std::vector<GatherCall> by_row(gc);
        std::sort(by_row.begin(), by_row.end(), [](const GatherCall &x, const GatherCall &y) { return x.src_off < y.src_off; });
        memcpy(desc_host.data(), by_row.data(), sizeof(GatherCall) * by_row.size());
    } else memcpy(desc_host.data(), gc.data(), sizeof(GatherCall) * gc.size());
    if (!cj.empty()) memcpy(desc_host.data() + sizeof(GatherCall) * gc.size(), cj.data(), sizeof(CanonJob) * cj.size());
    PAV_HIP(ctx, hipMemsetAsync(d_keys, 0xFF, 8ull * keys, st));
    PAV_HIP(ctx, hipMemcpyAsync(d_gc, desc_host.data(), desc_bytes, hipMemcpyHostToDevice, st));
    { int rcw = wait_planes(ctx); if (rcw != PAV_OK) return rcw; }
    if (tiles) PAV_LAUNCH(ctx, "k_canon_insert", k_canon_insert_batch, tiles, 256, 0, RV, d_cj, (uint32_t)cj.size(), k, d_keys);
    // Host side (invscan.cpp): the round's block is laid out as whole-round columns, K0 | K1 | K2 | KMER | INDEX | STATE_MER |
    // STATE | FLANK | MATCH, exactly like the device columns: one copy (two when the tail of K1 stays behind).
    if (dense) {
        PAV_LAUNCH(ctx, "k_annotate_calls", k_annotate_calls, (uint32_t)((a_t + 1023) / 1024), 256, 0, d_gc, (uint32_t)gc.size(), a_t, k,
                   D->index.as<uint32_t>(), D->kmer.as<unsigned long long>(), d_keys, D->flank.as<uint8_t>(), D->match.as<uint8_t>());
        // the blocks change hands (host side only: the kernels above and the views of the batch keep their addresses); the block
        // the batch gets is never smaller than the one it gives, so the next batch does not allocate
        PAV_HIP(ctx, stage.buf.reserve_exact(D->table_block.cap));
        std::swap(stage.buf, D->table_block);
        stage.rows = a_t;
        for (const GatherCall &g : gc) stage.row0.push_back(g.src_off);
        stage.copies[stage.n_copies++] = CallStage::Copy{stage.buf.p, 0, col_bytes};
    } else {
        PAV_HIP(ctx, stage.buf.reserve(col_bytes + 64));
        uint8_t *d_cols = stage.buf.as<uint8_t>();
        double *g_k0 = reinterpret_cast<double *>(d_cols), *g_k1 = g_k0 + total, *g_k2 = g_k1 + total;
        unsigned long long *g_kmer = reinterpret_cast<unsigned long long *>(g_k2 + total);
        uint32_t *g_index = reinterpret_cast<uint32_t *>(g_kmer + total);
        int8_t *g_sm = reinterpret_cast<int8_t *>(g_index + total), *g_st = g_sm + total;
        uint8_t *g_fl = reinterpret_cast<uint8_t *>(g_st + total), *g_ma = g_fl + total;
        PAV_LAUNCH(ctx, "k_gather_calls", k_gather_calls, (uint32_t)((total + 255) / 256), 256, 0, d_gc, (uint32_t)gc.size(), (uint32_t)total, k,
                   D->index.as<uint32_t>(), D->state_mer.as<int8_t>(), D->state.as<int8_t>(), D->kern[0].as<double>(), D->kern[1].as<double>(),
                   D->kern[2].as<double>(), D->kmer.as<unsigned long long>(), d_keys, g_index, g_sm, g_st, g_k0, g_k1, g_k2, g_kmer, g_fl, g_ma);
        stage.rows = total;
        for (const GatherCall &g : gc) stage.row0.push_back(g.dst_off);
        if (k1_rows >= total) {
            stage.copies[stage.n_copies++] = CallStage::Copy{d_cols, 0, col_bytes};
        } else {                                                        // K0 | leading part of K1, then K2 | ... | MATCH
            stage.copies[stage.n_copies++] = CallStage::Copy{d_cols, 0, 8ull * (total + k1_rows)};
            stage.copies[stage.n_copies++] = CallStage::Copy{g_k2, 16ull * total, col_bytes - 16ull * total};
        }
    }
    if (getenv("PAV_TIMING")) fprintf(stderr, "[pav timing]   call tables: %zu calls, %llu rows (%.1f MB %s), %llu hash slots, %u insert tiles\n",
                                      calls.size(), (unsigned long long)total, (double)col_bytes / 1e6,
                                      dense ? "resident, the batch's own block" : "packed, resident", (unsigned long long)keys, tiles);
    return PAV_OK;                                                  // the tables stay in HBM until somebody asks (stage_copy)
}

// The copy runs on the copy stream behind the scan (wait_tables() before the host reads the block).
// (a hand-rolled 64-workgroup copy kernel was tried instead of the runtime's copy to keep the chip free for the next
// step: 16.6 ms per step against 13.9 ms, the runtime's copy is the better one)
int density_copy_now(pav_ctx *ctx, CallStage &stage) {
    DensityState *D = dstate(ctx);
    if (!D->gathered) PAV_HIP(ctx, hipEventCreateWithFlags(&D->gathered, hipEventDisableTiming));
    PAV_HIP(ctx, hipEventRecord(D->gathered, ctx->stream));
    PAV_HIP(ctx, hipStreamWaitEvent(ctx->stream3, D->gathered, 0));
    return stage_copy(ctx, stage);
}

// Queue the device-to-host copies of a packed round on the copy stream (the caller has made that stream wait for the gather).
int stage_copy(pav_ctx *ctx, CallStage &stage) {
    if (stage.n_copies && !stage.host) return fail(ctx, PAV_E_STATE, "stage_copy: the round has no host block");
    for (int c = 0; c < stage.n_copies; ++c)
        PAV_HIP(ctx, hipMemcpyAsync(static_cast<uint8_t *>(stage.host) + stage.copies[c].dst_off, stage.copies[c].src, stage.copies[c].bytes,
                                    hipMemcpyDeviceToHost, ctx->stream3));
    if (stage.n_copies) {
        PAV_HIP(ctx, hipEventRecord(ctx->tables_done, ctx->stream3));
        ctx->tables_pending = true;
    }
    stage.n_copies = 0;
    return PAV_OK;
}

}  // namespace pav

using namespace pav;

extern "C" {

void pav_density_release(pav_ctx *ctx) {
    if (!ctx || !ctx->density) return;
#ifdef PAV_KMER_PROF
    { unsigned long long h[16] = {0};
      if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_kmer_prof), sizeof h) == hipSuccess && h[15]) {
          static const char *nm[7] = {"header+loads issued", "clear+barrier", "ref windows wait", "inserts", "barrier", "tig windows wait", "look-ups"};
          fprintf(stderr, "[k_kmer_lds profile] %llu workgroups; cycles of wave 0 per workgroup:", h[15]);
          for (int i = 0; i < 7; ++i) fprintf(stderr, " %s %.0f;", nm[i], (double)h[i] / (double)h[15]);
          fprintf(stderr, "\n");
      } }
#endif
    DensityState *D = static_cast<DensityState *>(ctx->density);
    D->release();
    delete D;
    ctx->density = nullptr;
}

uint64_t pav_kmer_rev_complement(uint64_t kmer, int k) {
    uint64_t rc = 0;
    for (int i = 0; i < k; ++i) { rc = (rc << 2) | ((kmer & 3ull) ^ 3ull); kmer >>= 2; }
    return rc;
}

uint64_t pav_kmer_canonical(uint64_t kmer, int k) {
    const uint64_t rc = pav_kmer_rev_complement(kmer, k);
    return kmer <= rc ? kmer : rc;
}

int pav_density_batch(pav_ctx *ctx, uint32_t n_jobs, const pav_den_job *jobs, const pav_den_params *pp,
                      pav_den_result *results) {
    if (!ctx || !pp || (n_jobs && (!jobs || !results))) return fail(ctx, PAV_E_ARG, "pav_density_batch: null argument");
    if (pp->k < 1 || pp->k > 32) return fail(ctx, PAV_E_LIMIT, "pav_density_batch: k = %d is outside 1..32", pp->k);
    PAV_HIP(ctx, hipSetDevice(ctx->device));
    DensityState *D = dstate(ctx);
    const bool timing = getenv("PAV_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    double t_mark = t_begin;
    auto lap = [&](const char *what) { if (timing) { const double t = now(); fprintf(stderr, "[pav timing]   density %-10s %.2f ms\n", what, (t - t_mark) * 1e3); t_mark = t; } };
    D->valid = false;
    D->n_jobs = n_jobs;
    D->params = *pp;
    D->results.assign(n_jobs, pav_den_result{});
    D->runs.resize(n_jobs);                                              // (cleared, not rebuilt: a thousand small vectors keep their blocks)
    for (auto &r : D->runs) r.clear();
    D->no_table.assign(n_jobs, 0);
    if (n_jobs == 0) { D->valid = true; return PAV_OK; }
    const SeqStore &RS = ctx->seq[PAV_ROLE_REF], &TS = ctx->seq[PAV_ROLE_TIG];
    const int k = pp->k;

    // ---- plan arenas --------------------------------------------------------------------------------------
    D->h_jobs.assign(n_jobs, JobDev{});
    uint64_t a_r = 0, a_t = 0, a_h = 0;
    std::vector<uint32_t> tile_job_r, tile_job_t;
    // k-mer sets in LDS unless asked otherwise (params, env), the count limit does not fit a byte, or a region needs
    // more partitions than the bucket kernels' histograms hold (> 1.8 Mbp; the reference's MAX_REGION_SIZE is 1.2 Mbp)
    // ... or k = 32: an LDS slot keeps two orientation bits above the 62 bits of a canonical 31-mer; a 32-mer fills the word, its
    // sets live in the HBM tables (the reference takes any k, rules/call_inv.snakefile:131; PAV's default is 31)
    const bool lds_sets = pp->kmer_mode != PAV_KMER_HBM && pp->max_ref_kmer_count <= LDS_MAX_LIMIT && pp->k <= 31 &&
                          getenv("PAV_KMER_HBM") == nullptr;
    std::vector<PartItem> items;
    uint32_t n_hbm_jobs = 0;
    uint64_t n_lists = 0, n_bcount = 0;
    uint64_t total_parts = 0, max_parts = 0, samp_ub = 0, n_eval_tiles_ub = 0;
    D->h_kde.assign(n_jobs, JobKde{});
    for (uint32_t j = 0; j < n_jobs; ++j) {
        const pav_den_job &q = jobs[j];
        if (q.ref_id >= RS.n || q.tig_id >= TS.n) return fail(ctx, PAV_E_ARG, "pav_density_batch: job %u references a sequence that is not loaded", j);
        if (q.ref_end < q.ref_pos || q.ref_end > RS.len[q.ref_id] || q.tig_end < q.tig_pos || q.tig_end > TS.len[q.tig_id])
            return fail(ctx, PAV_E_ARG, "pav_density_batch: job %u region outside its record", j);
        if (q.state_run_smooth < 1) return fail(ctx, PAV_E_ARG, "pav_density_batch: job %u state_run_smooth must be >= 1", j);
        JobDev &jd = D->h_jobs[j];
        jd.ref_abs = RS.off[q.ref_id] + q.ref_pos;
        jd.tig_abs = TS.off[q.tig_id] + q.tig_pos;
        jd.ref_len = (uint32_t)(q.ref_end - q.ref_pos);
        jd.tig_len = (uint32_t)(q.tig_end - q.tig_pos);
        jd.ref_rc = q.ref_rc ? 1 : 0;
        jd.srs = q.state_run_smooth;
        const uint64_t n_ref_kmers = jd.ref_len >= (uint32_t)k ? (uint64_t)jd.ref_len - k + 1 : 0;
        const uint64_t parts = std::max<uint64_t>(1, (n_ref_kmers + LDS_FILL - 1) / LDS_FILL);
        if (lds_sets && parts <= LDS_MAX_PARTS) {
            jd.n_parts = (uint32_t)parts;
            jd.bucket_off = (uint32_t)n_bcount; n_bcount += 2 * parts;
            jd.cap_r = bucket_cap(jd.ref_len, jd.n_parts);
            jd.cap_t = bucket_cap(jd.tig_len, jd.n_parts);
            jd.list_off_r = n_lists; n_lists += parts * jd.cap_r;
            jd.list_off_t = n_lists; n_lists += parts * jd.cap_t;
            total_parts += parts; max_parts = std::max<uint64_t>(max_parts, parts);
        } else {
            const uint32_t cap = pow2_at_least(2ull * jd.ref_len + 2);
            jd.ht_off = a_h; jd.ht_mask = cap - 1; a_h += cap;
            ++n_hbm_jobs;
        }
        jd.rpos_off = a_r; jd.tpos_off = a_t;
        jd.first_tile = (uint32_t)(a_t / DTILE);
        const uint64_t tr = (std::max<uint64_t>(jd.ref_len, 1) + DTILE - 1) / DTILE, tt = (std::max<uint64_t>(jd.tig_len, 1) + DTILE - 1) / DTILE;
        tile_job_r.insert(tile_job_r.end(), tr, j);
        tile_job_t.insert(tile_job_t.end(), tt, j);
        a_r += tr * DTILE; a_t += tt * DTILE;
        // what is known before anything runs (device-planned batches: upper bounds; rows <= contig positions of the region)
        JobKde &kd = D->h_kde[j];
        kd.srs = jd.srs;
        kd.samp_off = (uint32_t)samp_ub;
        kd.heads_off = jd.first_tile * HEADS_PER_TILE;
        const uint32_t ns = (std::max<uint32_t>(jd.tig_len, 1) + jd.srs - 1) / jd.srs + 1;
        samp_ub += ns;
        n_eval_tiles_ub += (ns + 63) / 64;
    }
    D->arena_t = a_t;
    const uint32_t n_tiles_r = (uint32_t)tile_job_r.size(), n_tiles_t = (uint32_t)tile_job_t.size();
    const bool have_lds = total_parts > 0;
    // The partition items and the evaluation tiles are planned BEHIND the launch of the bucket kernels, which do not read them
    // (plan_behind below): their slices of the input arena are sized from bounds here.  XCD-grouped items are padded: the longest
    // of the eight queues is at most total / 8 + the largest job.
    const size_t items_ub = have_lds ? (size_t)(total_parts + 8 * max_parts) : 0;
    auto plan_items = [&]() {   // One workgroup per (job, partition), the long scans first - and all partitions of a job on ONE XCD (workgroup b runs on
        // XCD b % 8, observed; used for speed only): the workgroups of a job answer into the same byte arrays, one byte per
        // contig k-mer at scattered positions, and read the same windows of the planes.  Spread over the eight L2s every line of
        // the answers left the chip once per XCD that had touched it (205 MB written per launch for 20 MB of answers); in one L2
        // the lines fill up before they go.  PAV_XCD_GROUP=0: the plain order.
        // (largest region first, ties in job order: one 64-bit key per job - size above, job number inverted below - and a plain sort)
        std::vector<uint64_t> keyed;
        keyed.reserve(n_jobs);
        for (uint32_t j = 0; j < n_jobs; ++j)
            if (D->h_jobs[j].n_parts) keyed.push_back((((uint64_t)D->h_jobs[j].ref_len + D->h_jobs[j].tig_len) << 31) | (uint64_t)(0x7FFFFFFFu - j));
        std::sort(keyed.begin(), keyed.end(), std::greater<uint64_t>());
        std::vector<uint32_t> order(keyed.size());
        for (size_t o = 0; o < keyed.size(); ++o) order[o] = 0x7FFFFFFFu - (uint32_t)(keyed[o] & 0x7FFFFFFFull);
        static const bool group = [] { const char *e = getenv("PAV_XCD_GROUP"); return !(e && e[0] == '0'); }();
        if (!group) {
            for (uint32_t j : order)
                for (uint32_t p = 0; p < D->h_jobs[j].n_parts; ++p) items.push_back(PartItem{j, p});
        } else {
            constexpr int XCDS = 8;
            // longest first, each to the shortest queue; queue x is the items at x, x + 8, x + 16, ... (written in place: a job's
            // partitions go to consecutive rounds of its queue)
            size_t fill[XCDS] = {0, 0, 0, 0, 0, 0, 0, 0};
            std::vector<std::pair<uint8_t, uint32_t>> place(order.size());     // (queue, first round) of every job, in `order`
            for (size_t o = 0; o < order.size(); ++o) {
                int best = 0;
                for (int x = 1; x < XCDS; ++x) if (fill[x] < fill[best]) best = x;
                place[o] = {(uint8_t)best, (uint32_t)fill[best]};
                fill[best] += D->h_jobs[order[o]].n_parts;
            }
            size_t longest = 0;
            for (int x = 0; x < XCDS; ++x) longest = std::max(longest, fill[x]);
            items.assign(longest * XCDS, PartItem{~0u, 0u});                     // ~0: nothing to do
            for (size_t o = 0; o < order.size(); ++o) {
                const uint32_t j = order[o], np = D->h_jobs[j].n_parts;
                PartItem *at = items.data() + (size_t)place[o].second * XCDS + place[o].first;
                for (uint32_t p = 0; p < np; ++p, at += XCDS) *at = PartItem{j, p};
            }
            while (!items.empty() && items.back().job == ~0u) items.pop_back();
        }
    };

    const bool want_runs = pp->kde_mode != PAV_KDE_DIRECT;
    bool fast = want_runs && n_hbm_jobs == 0 && have_lds && getenv("PAV_DENSITY_HOST") == nullptr && samp_ub <= 0xFFFFFFFFull;
    // the caller reads run lists only, and tables of calls: regions with FWD k-mers only are settled from their counts (fwd_only)
    const bool scan_only = fast && ctx->den_scan_only && getenv("PAV_SCAN_FULL") == nullptr;
    std::vector<EvalTile> tiles_ub;
    const size_t tiles_cap = fast ? (size_t)n_eval_tiles_ub : 0;
    auto plan_tiles = [&]() {
        if (!fast) return;
        tiles_ub.reserve(tiles_cap);
        for (uint32_t j = 0; j < n_jobs; ++j) {
            const JobDev &jd = D->h_jobs[j];
            const uint32_t ns = (std::max<uint32_t>(jd.tig_len, 1) + jd.srs - 1) / jd.srs + 1;
            for (uint32_t f = 0; f < ns; f += 64) tiles_ub.push_back(EvalTile{j, f, 64u, 0u});
        }
    };
    // the event list as large as either path asks for (prepare_events)
    uint32_t ev_cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(65536, (a_t + 16ull * n_tiles_t) / 16 + 4ull * n_jobs), 0x7FFFFFFF);

    // ---- the two arenas of small buffers ------------------------------------------------------------------------------------
    struct Slice { DevBuf *buf; const void *src; size_t bytes, at; };
    auto lay = [](std::vector<Slice> &v, size_t at0) { size_t at = at0; for (Slice &x : v) { x.at = at; at += (x.bytes + 255) / 256 * 256; } return at; };
    // (the first four go up in front of the bucket kernels, the last two behind their launch)
    std::vector<Slice> in_sl = {{&D->jobs, D->h_jobs.data(), sizeof(JobDev) * n_jobs, 0}, {&D->tile_job_r, tile_job_r.data(), 4ull * n_tiles_r, 0},
                                {&D->tile_job_t, tile_job_t.data(), 4ull * n_tiles_t, 0},
                                {&D->kde, fast ? D->h_kde.data() : nullptr, sizeof(JobKde) * n_jobs, 0},
                                {&D->items, nullptr, sizeof(PartItem) * items_ub, 0},
                                {&D->tiles, nullptr, sizeof(EvalTile) * tiles_cap, 0}};
    const size_t in_bytes = lay(in_sl, 0);
    const size_t in_front = in_sl[4].at;
    // front of the zero arena = what comes back: event count (64) | guard (64) | plan flags (64) | statistics | event list
    constexpr size_t ZH = 192;
    std::vector<Slice> z_sl = {{&D->stat, nullptr, sizeof(JobStat) * n_jobs, 0}, {&D->events, nullptr, sizeof(HeadEvent) * (size_t)ev_cap, 0},
                               {&D->bcount, nullptr, 4 * n_bcount, 0}, {&D->samp_flag, nullptr, 4 * (samp_ub + 2), 0}, {&D->row_flag, nullptr, a_t, 0}};
    const size_t z_bytes = lay(z_sl, ZH);
    PAV_HIP(ctx, D->in_arena.reserve(in_bytes + 256));
    PAV_HIP(ctx, D->zero_arena.reserve(z_bytes + 256));
    uint8_t *h_in = static_cast<uint8_t *>(D->pinned_in(in_bytes + 256));
    if (!h_in) return fail(ctx, PAV_E_HIP, "pav_density_batch: cannot pin host memory for the batch descriptors");
    for (const Slice &x : in_sl) {
        x.buf->alias(D->in_arena.as<uint8_t>() + x.at, (x.bytes + 255) / 256 * 256);
        if (x.src && x.bytes) memcpy(h_in + x.at, x.src, x.bytes);
    }
    for (const Slice &x : z_sl) x.buf->alias(D->zero_arena.as<uint8_t>() + x.at, (x.bytes + 255) / 256 * 256);
    D->ev_count.alias(D->zero_arena.as<uint8_t>(), 64);
    D->guard.alias(D->zero_arena.as<uint8_t>() + 64, 64);
    D->plan_flags.alias(D->zero_arena.as<uint8_t>() + 128, 64);
    static_assert(sizeof(GuardDev) <= 64, "the guard block has 64 bytes in the zero arena");

    PAV_HIP(ctx, D->keys.reserve(8 * a_h));
    PAV_HIP(ctx, D->cnt.reserve(4 * a_h));
    if (have_lds) {
        if (n_bcount > 0xFFFFFFFFull) return fail(ctx, PAV_E_LIMIT, "pav_density_batch: too many k-mer partitions in one batch");
        PAV_HIP(ctx, D->lists.reserve(4 * n_lists));
    }
    PAV_HIP(ctx, D->st_tmp.reserve(a_t + 64));
    PAV_HIP(ctx, D->tile_sum.reserve(16ull * n_tiles_t));
    PAV_HIP(ctx, D->tile_pre.reserve(32ull * (n_tiles_t + 1)));
    PAV_HIP(ctx, D->table_block.reserve(40ull * a_t + 64));          // K0 | K1 | K2 | KMER | INDEX | STATE_MER | STATE | FLANK | MATCH, a_t rows each
    {
        uint8_t *tb = D->table_block.as<uint8_t>();
        for (int s = 0; s < 3; ++s) D->kern[s].alias(tb + 8ull * s * a_t, 8ull * a_t);
        D->kmer.alias(tb + 24ull * a_t, 8ull * a_t);
        D->index.alias(tb + 32ull * a_t, 4ull * a_t);
        D->state_mer.alias(tb + 36ull * a_t, a_t);
        D->state.alias(tb + 37ull * a_t, a_t);
        D->flank.alias(tb + 38ull * a_t, a_t);
        D->match.alias(tb + 39ull * a_t, a_t);
        D->block_rows = a_t;
    }
    PAV_HIP(ctx, D->fill_list.reserve(4 * a_t));
    PAV_HIP(ctx, D->win_fill.reserve(a_t));
    for (int s = 0; s < 3; ++s) {
        PAV_HIP(ctx, D->list[s].reserve(4 * a_t));
    }
    hipStream_t st = ctx->stream;
    lap("plan+alloc");
    PAV_HIP(ctx, hipMemcpyAsync(D->in_arena.p, h_in, in_front, hipMemcpyHostToDevice, st));
    PAV_HIP(ctx, hipMemsetAsync(D->zero_arena.p, 0, z_sl[1].at, st));                                  // event count .. statistics
    PAV_HIP(ctx, hipMemsetAsync(D->zero_arena.as<uint8_t>() + z_sl[2].at, 0, z_bytes - z_sl[2].at, st));   // bucket counts, guard flags
    if (a_h) {
        PAV_HIP(ctx, hipMemsetAsync(D->keys.p, 0xFF, 8 * a_h, st));
        PAV_HIP(ctx, hipMemsetAsync(D->cnt.p, 0, 4 * a_h, st));
    }

    const JobDev *d_jobs = D->jobs.as<JobDev>();
    JobStat *d_stat = D->stat.as<JobStat>();
    const uint32_t *d_tjr = D->tile_job_r.as<uint32_t>(), *d_tjt = D->tile_job_t.as<uint32_t>();
    unsigned long long *const d_keys = D->keys.as<unsigned long long>();      // HBM tables (jobs without LDS sets, failure path)
    const SeqView RV = RS.view(), TV = TS.view();

    // ---- k-mer states and compaction -------------------------------------------------------------------------
    {   // the contig planes are packed on demand (ctx.hip "lazy contig pack"): the blocks under the regions of this batch, plus
        // what the 64-base windows of the scans read past a region's end
        std::vector<PlaneSpan> spans;
        if (!TS.planes_full) {
            spans.reserve(n_jobs);
            for (uint32_t j = 0; j < n_jobs; ++j) spans.push_back(PlaneSpan{D->h_jobs[j].tig_abs, (uint64_t)D->h_jobs[j].tig_len + 128});
        }
        const int rcw = need_planes_spans(ctx, PAV_ROLE_TIG, spans);
        if (rcw != PAV_OK) return rcw;
    }
    lap("copies+spans");
    if (have_lds) {
        PAV_LAUNCH(ctx, "k_bucket_ref", k_bucket_ref, n_tiles_r, 256, 0, d_jobs, d_tjr, RV, k, D->lists.as<uint32_t>(),
                   D->bcount.as<uint32_t>(), d_stat);
        PAV_LAUNCH(ctx, "k_bucket_tig", k_bucket_tig, n_tiles_t, 256, 0, d_jobs, d_tjt, TV, k, D->lists.as<uint32_t>(),
                   D->bcount.as<uint32_t>(), D->st_tmp.as<int8_t>(), d_stat);
    }
    lap("buckets out");
    {   // plan_behind: the partition items (XCD order) and the evaluation tiles, while the bucket kernels run
        plan_items();
        lap("items");
        plan_tiles();
        lap("tiles");
        if (items.size() > items_ub || tiles_ub.size() > tiles_cap) return fail(ctx, PAV_E_STATE, "pav_density_batch: plan bounds (%zu / %zu items, %zu / %zu tiles)", items.size(), items_ub, tiles_ub.size(), tiles_cap);
        if (!items.empty()) memcpy(h_in + in_sl[4].at, items.data(), sizeof(PartItem) * items.size());
        if (!tiles_ub.empty()) memcpy(h_in + in_sl[5].at, tiles_ub.data(), sizeof(EvalTile) * tiles_ub.size());
        if (in_bytes > in_front) PAV_HIP(ctx, hipMemcpyAsync(D->in_arena.as<uint8_t>() + in_front, h_in + in_front, in_bytes - in_front, hipMemcpyHostToDevice, st));
        lap("plan behind");
    }
    if (have_lds) {
#ifdef PAV_TUNING             // an ablated instance in front of the real one, into a scratch array: what each part of the kernel costs
        if (const char *abl = getenv("PAV_KMER_ABL")) {
            static DevBuf scratch, scratch_stat;                                  // (flags of the ablated run go nowhere that is read)
            PAV_HIP(ctx, scratch.reserve(a_t + 64));
            PAV_HIP(ctx, scratch_stat.reserve(sizeof(JobStat) * n_jobs));
            JobStat *no_stat = scratch_stat.as<JobStat>();
#define PAV_ABL_LAUNCH(A, W) PAV_LAUNCH(ctx, "k_kmer_abl", (k_kmer_abl<A, W>), (uint32_t)items.size(), LDS_THREADS, 0, D->items.as<PartItem>(), d_jobs, RV, TV, k, \
                   pp->max_ref_kmer_count, D->lists.as<uint32_t>(), D->bcount.as<uint32_t>(), scratch.as<int8_t>(), no_stat)
            const std::string a = abl;
            if (a == "0") PAV_ABL_LAUNCH(0, 0); else if (a == "1") PAV_ABL_LAUNCH(1, 0); else if (a == "2") PAV_ABL_LAUNCH(2, 0);
            else if (a == "3") PAV_ABL_LAUNCH(3, 0); else if (a == "5") PAV_ABL_LAUNCH(5, 0); else if (a == "w") PAV_ABL_LAUNCH(0, 1);
            else if (a == "5w") PAV_ABL_LAUNCH(5, 1); else if (a == "1w") PAV_ABL_LAUNCH(1, 1);
        }
#endif
        PAV_LAUNCH(ctx, "k_kmer_lds", k_kmer_lds, (uint32_t)items.size(), LDS_THREADS, 0, D->items.as<PartItem>(), d_jobs, RV, TV, k,
                   pp->max_ref_kmer_count, D->lists.as<uint32_t>(), D->bcount.as<uint32_t>(), D->st_tmp.as<int8_t>(), d_stat);
        PAV_LAUNCH(ctx, "k_state_combine", k_state_combine, n_tiles_t, 256, 0, d_jobs, d_tjt, D->st_tmp.as<int8_t>(), d_stat,
                   D->tile_sum.as<uint32_t>(), scan_only ? 1 : 0);
    }
    if (n_hbm_jobs) {
        PAV_LAUNCH(ctx, "k_ref_insert", k_ref_insert, (uint32_t)(a_r / 256), 256, 0, d_jobs, d_tjr, RV, k, pp->max_ref_kmer_count,
                   d_keys, D->cnt.as<uint32_t>(), d_stat, 0u, 0);
        PAV_LAUNCH(ctx, "k_tig_state", k_tig_state, (uint32_t)(a_t / 256), 256, 0, d_jobs, d_tjt, TV, k, d_keys,
                   D->st_tmp.as<int8_t>(), d_stat, 0u);
    }
    CompactArgs CA;
    CA.jobs = d_jobs; CA.tile_job = d_tjt; CA.stat = d_stat; CA.st_tmp = D->st_tmp.as<int8_t>();
    CA.tile_pre = D->tile_pre.as<unsigned long long>(); CA.T = TV; CA.k = k; CA.min_state_count = pp->min_state_count;
    CA.index = D->index.as<uint32_t>(); CA.state_mer = D->state_mer.as<int8_t>(); CA.state = D->state.as<int8_t>();
    CA.kmer = D->kmer.as<unsigned long long>();
    for (int s = 0; s < 3; ++s) CA.list[s] = D->list[s].as<uint32_t>();
    CA.events = nullptr; CA.ev_cap = 0; CA.ev_count = nullptr; CA.tile_heads = nullptr; CA.tile_head_cnt = nullptr; CA.scan_only = 0;
    std::vector<JobStat> hs(n_jobs);
    constexpr uint32_t EV_PREFETCH = 16384;                            // head events copied together with their count
    // pinned readback area: [event count | first events] [per-job statistics] [guard counters]
    const size_t pin_stat_off = (sizeof(HeadEvent) * (size_t)EV_PREFETCH + 64 + 63) / 64 * 64;
    const size_t pin_guard_off = pin_stat_off + (sizeof(JobStat) * (size_t)n_jobs + 63) / 64 * 64;
    // (the device-planned path reads the front of the zero arena into the same block)
    const uint32_t ev_first = std::min<uint32_t>(ev_cap, std::max<uint32_t>(4096u, 4u * n_jobs));   // run heads that come back with their count
    uint8_t *h_pin = static_cast<uint8_t *>(D->pinned(std::max(pin_guard_off, z_sl[1].at + sizeof(HeadEvent) * (size_t)ev_first) + 64));
    if (!h_pin) return fail(ctx, PAV_E_HIP, "pav_density_batch: cannot pin host memory for the readbacks");
    auto queue_stats = [&]() -> int {                                  // in front of a synchronisation that follows
        PAV_HIP(ctx, hipMemcpyAsync(h_pin + pin_stat_off, d_stat, sizeof(JobStat) * n_jobs, hipMemcpyDeviceToHost, st));
        return PAV_OK;
    };
    auto take_stats = [&]() { memcpy(hs.data(), h_pin + pin_stat_off, sizeof(JobStat) * n_jobs); };
    auto read_stats = [&]() -> int {
        const int rcq = queue_stats();
        if (rcq != PAV_OK) return rcq;
        PAV_HIP(ctx, hipStreamSynchronize(st));
        take_stats();
        return PAV_OK;
    };
    // Run heads of a per-row state array: events sorted by (job, row).  The kernels that produce the arrays write the events
    // themselves (STATE_MER: k_compact_scatter, STATE: k_finalize); k_heads only repeats the work when the event buffer was too small.
    auto prepare_events = [&](uint64_t hint, bool cleared = false) -> int {     // cleared: the count is zero already (the batch's fill)
        ev_cap = std::max(ev_cap, (uint32_t)std::min<uint64_t>(std::max<uint64_t>(65536, hint / 16 + 4ull * n_jobs), 0x7FFFFFFF));
        PAV_HIP(ctx, D->events.reserve(sizeof(HeadEvent) * (size_t)ev_cap));
        PAV_HIP(ctx, D->ev_count.reserve(16));
        if (!cleared) PAV_HIP(ctx, hipMemsetAsync(D->ev_count.p, 0, 4, st));
        return PAV_OK;
    };
    auto launch_heads = [&](const int8_t *d_state) -> int {
        PAV_LAUNCH(ctx, "k_heads", k_heads, (uint32_t)(a_t / 256), 256, 0, d_jobs, d_tjt, d_stat, d_state,
                   D->index.as<uint32_t>(), D->events.as<HeadEvent>(), ev_cap, D->ev_count.as<uint32_t>());
        return PAV_OK;
    };
    // d_state: the array k_heads walks again should the events not fit
    bool ev_overflow = false;                                          // read_events(nullptr, ...): too many events, nothing was read
    auto read_events = [&](const int8_t *d_state, std::vector<HeadEvent> &ev) -> int {
        ev_overflow = false;
        while (true) {
            // the count and the first EV_PREFETCH events come back together (a batch of 1 000 regions has ~2 000 heads)
            const uint32_t pre = std::min(ev_cap, EV_PREFETCH);
            PAV_HIP(ctx, hipMemcpyAsync(h_pin, D->ev_count.p, 4, hipMemcpyDeviceToHost, st));
            PAV_HIP(ctx, hipMemcpyAsync(h_pin + 64, D->events.p, sizeof(HeadEvent) * pre, hipMemcpyDeviceToHost, st));
            PAV_HIP(ctx, hipStreamSynchronize(st));
            uint32_t n_ev = 0;
            memcpy(&n_ev, h_pin, 4);
            if (n_ev > ev_cap && !d_state) { ev_overflow = true; ev.clear(); return PAV_OK; }
            if (n_ev > ev_cap) {
                ev_cap = n_ev + 1024;
                int rc = prepare_events(0);
                if (rc == PAV_OK) rc = launch_heads(d_state);
                if (rc != PAV_OK) return rc;
                continue;
            }
            ev.resize(n_ev);
            if (n_ev) memcpy(ev.data(), h_pin + 64, sizeof(HeadEvent) * std::min(n_ev, pre));
            if (n_ev > pre)
                PAV_HIP(ctx, hipMemcpy(ev.data() + pre, D->events.as<HeadEvent>() + pre, sizeof(HeadEvent) * (n_ev - pre), hipMemcpyDeviceToHost));
            break;
        }
        std::sort(ev.begin(), ev.end(), [](const HeadEvent &a, const HeadEvent &b) { return a.job != b.job ? a.job < b.job : a.row < b.row; });
        return PAV_OK;
    };
    auto collect_heads = [&](const int8_t *d_state, std::vector<HeadEvent> &ev, uint64_t hint) -> int {
        int rc = prepare_events(hint);
        if (rc == PAV_OK) rc = launch_heads(d_state);
        if (rc == PAV_OK) rc = read_events(d_state, ev);
        return rc;
    };
    // compaction to the informative rows, then readback 1: per-job counts and moments -> status, bandwidths (host, libm:
    // same arithmetic as scipy)
    std::vector<HeadEvent> mev;                                        // run heads of STATE_MER (closed-form run sums)
    auto compact_and_read = [&]() -> int {
        CA.events = nullptr; CA.ev_cap = 0; CA.ev_count = nullptr;
        if (want_runs) {                                               // every change of STATE_MER + one event per tile
            const int rce = prepare_events(a_t + 16ull * n_tiles_t);
            if (rce != PAV_OK) return rce;
            CA.events = D->events.as<HeadEvent>(); CA.ev_cap = ev_cap; CA.ev_count = D->ev_count.as<uint32_t>();
        }
        PAV_LAUNCH(ctx, "k_compact_reduce", k_compact_reduce, n_tiles_t, 256, 0, d_tjt, d_stat, D->st_tmp.as<int8_t>(),
                   pp->min_state_count, D->tile_sum.as<uint32_t>());
        PAV_LAUNCH(ctx, "k_scan_tiles4", k_scan_tiles4, 1, 256, 0, D->tile_sum.as<uint32_t>(), D->tile_pre.as<unsigned long long>(),
                   n_tiles_t);
        PAV_LAUNCH(ctx, "k_compact_scatter", k_compact_scatter, n_tiles_t, 256, 0, CA);
        if (!want_runs) return read_stats();
        const int rcq = queue_stats();                                 // one synchronisation for the statistics and the events
        if (rcq != PAV_OK) return rcq;
        const int rcv = read_events(nullptr, mev);
        take_stats();
        return rcv;
    };
    // ---- device-planned batch: every kernel of the density stage queued back to back, ONE readback at the end ------------------
    // (k_plan / k_plan_fill do what the host does between the three synchronisations of the path below).  Taken for run-sum
    // batches whose k-mer sets all live in LDS; whatever it cannot finish - a partition overflow or a count above the limit
    // (HBM tables), a near-tie the guard wants re-evaluated, more STATE_MER changes in one tile than its slots hold - sends
    // the whole batch through the host-planned path below, which is the same arithmetic with every rare path in it.
    if (fast) {
        uint32_t max_len = 0;
        for (uint32_t j = 0; j < n_jobs; ++j) max_len = std::max(max_len, D->h_jobs[j].tig_len);
        if (D->h_pow.size() < (size_t)max_len + 2) {                   // libm's pow, as scipy's factor is computed (density.py:198)
            const size_t old_n = D->h_pow.size(), new_n = std::max<size_t>((size_t)max_len + 2, 2 * old_n);
            D->h_pow.resize(new_n);
            for (size_t n = old_n; n < new_n; ++n) D->h_pow[n] = std::pow((double)n, -1.0 / 5.0);
            PAV_HIP(ctx, hipStreamSynchronize(st));                    // (a kernel of an earlier batch may still read the old table)
            PAV_HIP(ctx, D->pow_tab.reserve(sizeof(double) * new_n));
            PAV_HIP(ctx, hipMemcpyAsync(D->pow_tab.p, D->h_pow.data(), sizeof(double) * new_n, hipMemcpyHostToDevice, st));
        }
        {
            const size_t ft_cap = (size_t)(a_t / 64 + n_jobs + 1);
            PAV_HIP(ctx, D->tile_heads.reserve(4ull * HEADS_PER_TILE * n_tiles_t));
            PAV_HIP(ctx, D->tile_head_cnt.reserve(4ull * n_tiles_t));
            PAV_HIP(ctx, D->heads.reserve(sizeof(RunDev) * ((size_t)HEADS_PER_TILE * n_tiles_t + 1)));
            PAV_HIP(ctx, D->ftiles.reserve(sizeof(EvalTile) * ft_cap));
            for (int q = 0; q < 3; ++q) PAV_HIP(ctx, D->ks[q].reserve(8 * (samp_ub + 1)));
            PAV_HIP(ctx, D->ss.reserve(samp_ub + 1));
            PAV_HIP(ctx, D->scratch.reserve(4ull * (a_t / 256 + 1)));
            GuardArgs G{};
            G.rel = pp->guard_rel == 0.0 ? GUARD_REL_DEFAULT : pp->guard_rel;
            G.unres = GUARD_UNRESOLVED;
            G.cap = pp->guard_cap ? pp->guard_cap : GUARD_CAP_DEFAULT;
            G.stat = d_stat;
            if (G.rel > 0.0) {                                           // guard block and flags: slices of the zero arena, cleared
                PAV_HIP(ctx, D->guard_entries.reserve(8ull * G.cap));
                G.g = D->guard.as<GuardDev>(); G.entries = D->guard_entries.as<unsigned long long>();
                G.samp_flag = D->samp_flag.as<uint32_t>(); G.row_flag = D->row_flag.as<uint8_t>();
            }
            CA.tile_heads = D->tile_heads.as<uint32_t>(); CA.tile_head_cnt = D->tile_head_cnt.as<uint32_t>(); CA.scan_only = scan_only ? 1 : 0;
            PAV_LAUNCH(ctx, "k_scan_tiles_keep", k_scan_tiles_keep, 4, 256, 0, D->tile_sum.as<uint32_t>(), d_tjt, d_stat, pp->min_state_count,
                       D->tile_pre.as<unsigned long long>(), n_tiles_t);
            PAV_LAUNCH(ctx, "k_compact_scatter", k_compact_scatter, n_tiles_t, 256, 0, CA);
            CA.tile_heads = nullptr; CA.tile_head_cnt = nullptr; CA.scan_only = 0;
            PlanArgs PA;
            PA.jobs = d_jobs; PA.stat = d_stat; PA.kde = D->kde.as<JobKde>(); PA.n_jobs = n_jobs;
            PA.tile_heads = D->tile_heads.as<uint32_t>(); PA.tile_head_cnt = D->tile_head_cnt.as<uint32_t>(); PA.runs = D->heads.as<RunDev>();
            PA.pow_tab = D->pow_tab.as<double>(); PA.pow_n = (uint32_t)D->h_pow.size();
            PA.min_informative = pp->min_informative; PA.max_ref_kmer_count = pp->max_ref_kmer_count; PA.den_smooth = pp->den_smooth;
            PA.norm0 = std::pow(2 * 3.14159265358979323846, -0.5);
            PA.flags = D->plan_flags.as<uint32_t>();
            // the blocks of FIN_ROWS table rows k_finalize_blocks goes round: at most a_t / FIN_ROWS + one ragged block per region; the
            // count is the third word of the plan flags (zero arena: cleared with the batch)
            static const bool fin_list = [] { const char *e = getenv("PAV_FINALIZE_LIST"); return !(e && e[0] == '0'); }();
            PAV_HIP(ctx, D->fin_blocks.reserve(sizeof(FinBlock) * ((size_t)(a_t / FIN_ROWS) + n_jobs + 1)));
            PA.fin_blocks = fin_list ? D->fin_blocks.as<PlanArgs::FinBlockOut>() : nullptr;
            PA.n_fin_blocks = D->plan_flags.as<uint32_t>() + 2; PA.fin_rows = FIN_ROWS;
            PAV_LAUNCH(ctx, "k_plan", k_plan, (n_jobs + 3) / 4, 256, 0, PA);
            const JobKde *d_kde = D->kde.as<JobKde>();
            KdeArgs KA;
            KA.jobs = d_jobs; KA.kde = d_kde; KA.fill_list = D->fill_list.as<uint32_t>(); KA.state = D->state.as<int8_t>();
            KA.runs = D->heads.as<RunDev>(); KA.dyn = 1;
            for (int q = 0; q < 3; ++q) { KA.list[q] = D->list[q].as<uint32_t>(); KA.kern[q] = D->kern[q].as<double>(); KA.ks[q] = D->ks[q].as<double>(); }
            KA.ss = D->ss.as<int8_t>();
            G.pass = 0;
            KA.G = G;
            KA.tiles = D->tiles.as<EvalTile>(); KA.n_tiles = (uint32_t)tiles_ub.size(); KA.n_tiles_dev = nullptr;
            PAV_LAUNCH(ctx, "k_kde_eval", k_kde_eval, (uint32_t)tiles_ub.size(), 64 * KDE_WAVES, 0, KA);
            PAV_LAUNCH(ctx, "k_windows", k_windows, (uint32_t)tiles_ub.size(), 64, 0, d_jobs, D->tiles.as<EvalTile>(), d_kde, D->state_mer.as<int8_t>(),
                       D->ss.as<int8_t>(), D->ks[0].as<double>(), D->ks[1].as<double>(), D->ks[2].as<double>(),
                       pp->state_run_delta, D->fill_list.as<uint32_t>(), D->win_fill.as<uint8_t>(), d_stat, G);
            FillPlanArgs FP;
            FP.stat = d_stat; FP.kde = d_kde; FP.n_jobs = n_jobs; FP.ftiles = D->ftiles.as<EvalTile>();
            FP.n_ftiles = D->plan_flags.as<uint32_t>() + 1; FP.cap = (uint32_t)std::min<size_t>(ft_cap, 0xFFFFFFFFu);
            PAV_LAUNCH(ctx, "k_plan_fill", k_plan_fill, 1, 1024, 0, FP);
            KA.tiles = D->ftiles.as<EvalTile>(); KA.n_tiles = 0; KA.n_tiles_dev = FP.n_ftiles; KA.dyn = 0;
            PAV_LAUNCH(ctx, "k_kde_eval", k_kde_eval, (uint32_t)std::min<size_t>(ft_cap, 8192), 64 * KDE_WAVES, 0, KA);
            { const int rce = prepare_events(a_t, true); if (rce != PAV_OK) return rce; }
            FinArgs FA;
            FA.jobs = d_jobs; FA.tile_job = d_tjt; FA.kde = d_kde; FA.stat = d_stat; FA.win_fill = D->win_fill.as<uint8_t>();
            FA.state = D->state.as<int8_t>(); FA.index = D->index.as<uint32_t>();
            for (int q = 0; q < 3; ++q) { FA.ks[q] = D->ks[q].as<double>(); FA.kern[q] = D->kern[q].as<double>(); }
            FA.G = G; FA.ev_cap = ev_cap; FA.events = D->events.as<HeadEvent>(); FA.ev_count = D->ev_count.as<uint32_t>();
            FA.blk_spike = nullptr; FA.spike_add = G.rel > 0.0 ? 1 : 0;
            if (fin_list) {
                const uint32_t grid = (uint32_t)std::min<size_t>((size_t)(a_t / FIN_ROWS) + n_jobs + 1, (size_t)ctx->n_cu * 16);
                PAV_LAUNCH(ctx, "k_finalize", k_finalize_blocks, grid, 256, 0, FA, D->fin_blocks.as<FinBlock>(), D->plan_flags.as<uint32_t>() + 2);
            } else PAV_LAUNCH(ctx, "k_finalize", k_finalize, (uint32_t)(a_t / 256), 256, 0, FA);
            if (ctx->den_round && ctx->den_round->n_jobs == n_jobs && scan_only) {   // the round's next lifts, derived and answered in this launch set
                RoundHook &H = *ctx->den_round;
                PAV_LAUNCH(ctx, "k_round_decide", k_round_decide, n_jobs, 64, 0, d_stat, D->events.as<HeadEvent>(), D->ev_count.as<uint32_t>(), ev_cap, H.d_in,
                           static_cast<LiftQuery *>(H.d_queries), n_jobs, pp->min_state_count, pp->min_informative, pp->max_ref_kmer_count, H.min_exp_count, H.k);
                if (H.after) { const int rch = H.after(); if (rch != PAV_OK) return rch; }
                H.ran = true;
            }
            lap("enqueue");
            // the one readback: the front of the zero arena - event count, guard counters, plan flags, statistics, run heads of STATE
            const uint32_t pre = ev_first;
            const size_t rb_events = z_sl[1].at, rb_bytes = rb_events + sizeof(HeadEvent) * (size_t)pre;
            uint8_t *h_rb = h_pin;
            PAV_HIP(ctx, hipMemcpyAsync(h_rb, D->zero_arena.p, rb_bytes, hipMemcpyDeviceToHost, st));
            if (ctx->den_overlap) { auto f = std::move(ctx->den_overlap); ctx->den_overlap = nullptr; f(); lap("overlapped"); }   // the caller's deferred host work
            PAV_HIP(ctx, hipStreamSynchronize(st));
            lap("device plan");
            if (timing) fprintf(stderr, "[pav timing]   device-planned batches so far: %llu, sent back to the host-planned path: %llu\n",
                                (unsigned long long)D->n_fast + 1, (unsigned long long)D->n_fallback);
            if (timing) {
                double rows_all = 0, rows_fwd_only = 0, rows_norev = 0; uint32_t jobs_fwd_only = 0;
                const JobStat *q = reinterpret_cast<const JobStat *>(h_rb + z_sl[0].at);
                for (uint32_t j = 0; j < n_jobs; ++j) {
                    rows_all += q[j].n_rows;
                    if (q[j].m[2] == 0) rows_norev += q[j].n_rows;
                    if (q[j].m[1] == 0 && q[j].m[2] == 0) { rows_fwd_only += q[j].n_rows; ++jobs_fwd_only; }
                }
                fprintf(stderr, "[pav timing]   rows %.3g: %.0f %% in jobs without REV k-mers, %.0f %% in %u of %u jobs with FWD k-mers only\n", rows_all,
                        rows_all ? 100 * rows_norev / rows_all : 0.0, rows_all ? 100 * rows_fwd_only / rows_all : 0.0, jobs_fwd_only, n_jobs);
            }
            uint32_t n_ev = 0;
            GuardDev h_guard{};
            uint32_t h_flags[4] = {0, 0, 0, 0};
            memcpy(&n_ev, h_rb, 4);
            if (G.g) memcpy(&h_guard, h_rb + 64, sizeof h_guard);
            memcpy(h_flags, h_rb + 128, sizeof h_flags);
            memcpy(hs.data(), h_rb + z_sl[0].at, sizeof(JobStat) * n_jobs);
            lap("rl: stats");
            std::vector<HeadEvent> ev;
            const bool ev_lost = n_ev > ev_cap;                                  // more run heads than the list holds: the other path
            if (!ev_lost) {
                ev.resize(n_ev);
                if (n_ev) memcpy(ev.data(), h_rb + rb_events, sizeof(HeadEvent) * std::min(n_ev, pre));
                if (n_ev > pre)
                    PAV_HIP(ctx, hipMemcpy(ev.data() + pre, D->events.as<HeadEvent>() + pre, sizeof(HeadEvent) * (n_ev - pre), hipMemcpyDeviceToHost));
                // by (job, row): a counting pass over the jobs, then the handful of heads of each job by row (a comparison sort of
                // the whole list was 0.05 ms of a thousand-region round)
                bool in_range = true;
                std::vector<uint32_t> at(n_jobs + 1, 0);
                for (const HeadEvent &h : ev) { if (h.job >= n_jobs) { in_range = false; break; } at[h.job + 1]++; }
                if (in_range) {
                    for (uint32_t j = 0; j < n_jobs; ++j) at[j + 1] += at[j];
                    std::vector<HeadEvent> by(ev.size());
                    std::vector<uint32_t> put(at.begin(), at.end() - 1);
                    for (const HeadEvent &h : ev) by[put[h.job]++] = h;
                    for (uint32_t j = 0; j < n_jobs; ++j) {
                        HeadEvent *b = by.data() + at[j], *e = by.data() + at[j + 1];
                        if (e - b > 1) std::sort(b, e, [](const HeadEvent &x, const HeadEvent &y) { return x.row < y.row; });
                    }
                    ev.swap(by);
                } else std::sort(ev.begin(), ev.end(), [](const HeadEvent &x, const HeadEvent &y) { return x.job != y.job ? x.job < y.job : x.row < y.row; });
            }
            lap("rl: events");
            bool redo = ev_lost || h_flags[0] != 0 || h_guard.n_entries != 0 || h_guard.overflow != 0 || h_flags[1] >= FP.cap;
            for (uint32_t j = 0; j < n_jobs && !redo; ++j) redo = hs[j].lds_flags != 0;
            if (!redo) {
                D->n_fast += 1;
                if (scan_only)                                             // regions settled from their counts: rows = their FWD k-mers
                    for (uint32_t j = 0; j < n_jobs; ++j) {
                        JobStat &q = hs[j];
                        if (!fwd_only(q, pp->min_state_count)) continue;
                        const uint32_t n = q.st_count[0] >= pp->min_state_count ? q.st_count[0] : 0u;
                        q.n_rows = n; q.m[0] = n; q.m[1] = q.m[2] = 0; q.fill_n = 0;
                        D->no_table[j] = 1;
                    }
                // results: the host's own arithmetic on the same statistics (what k_plan did on the device)
                double pairs = 0, points = 0, data_pairs = 0;
                for (uint32_t j = 0; j < n_jobs; ++j) {
                    pav_den_result &r = D->results[j];
                    const JobStat &q = hs[j];
                    r.n_rows = q.n_rows;
                    for (int t3 = 0; t3 < 3; ++t3) r.state_count[t3] = q.m[t3];
                    r.max_count = q.max_count;
                    if (q.n_ref_valid == 0) { r.status = PAV_DEN_FAIL; r.fail_kind = 1; r.n_rows = 0; continue; }
                    if (q.max_count > pp->max_ref_kmer_count) { r.status = PAV_DEN_FAIL; r.fail_kind = 2; r.n_rows = 0; continue; }   // (not reached: LDS_EXCEED)
                    if (q.n_rows < pp->min_informative || q.n_rows == 0) { r.status = PAV_DEN_UNFINALISED; continue; }
                    r.status = PAV_DEN_OK;
                    const uint32_t n = q.n_rows, srs = D->h_jobs[j].srs;
                    if (D->no_table[j]) {                                // one run of state 0 over its rows (INDEX of the first and the last)
                        D->runs[j].push_back(pav_run{0, n, (int64_t)(0xFFFFFFFFu - q.inv_first), (int64_t)q.last1 - 1});
                        continue;
                    }
                    uint32_t n_samp = (n + srs - 1) / srs;
                    if ((uint64_t)(n_samp - 1) * srs != n - 1) n_samp += 1;
                    r.n_sample = n_samp;
                    r.n_eval = (uint64_t)n_samp + q.fill_n;
                    const double bandwidth = D->h_pow[n] * pp->den_smooth;
                    double runs_total = 0;
                    for (int t3 = 0; t3 < 3; ++t3) {
                        const uint64_t m = q.m[t3];
                        if (m == 0) continue;
                        const unsigned __int128 num = (unsigned __int128)m * q.s2[t3] - (unsigned __int128)q.s1[t3] * q.s1[t3];
                        r.h[t3] = std::sqrt((double)num / ((double)m * (double)(m - 1))) * bandwidth;
                    }
                    (void)runs_total;
                    points += (double)r.n_eval;
                    data_pairs += (double)r.n_eval * ((double)q.m[0] + q.m[1] + q.m[2]);
                    r.n_near_tie = q.n_near; r.n_reeval = q.n_reeval; r.n_unresolved = q.n_unres; r.n_spike_near = q.n_spike;
                    r.guard_fallback = 0;
                }
                lap("rl: results");
                pairs = points;                                          // (run pairs are not counted on this path; at least one run per point)
                ctx->kde_work[0] += points; ctx->kde_work[1] += pairs; ctx->kde_work[2] += data_pairs;
                for (size_t e = 0; e < ev.size(); ++e) {
                    const HeadEvent &h = ev[e];
                    if (h.state == -2 || D->results[h.job].status == PAV_DEN_FAIL) continue;
                    const HeadEvent &nx = ev[e + 1];
                    D->runs[h.job].push_back(pav_run{h.state, nx.row - h.row, (int64_t)h.index, (int64_t)nx.prev_index});
                }
                for (uint32_t j = 0; j < n_jobs; ++j) {
                    D->results[j].n_runs = (uint32_t)D->runs[j].size();
                    results[j] = D->results[j];
                }
                lap("rl");
                D->valid = true;
                return PAV_OK;
            }
            // back to the host-planned path: the compaction is repeated there, its counters start from zero
            D->n_fallback += 1;
            for (uint32_t j = 0; j < n_jobs; ++j) {
                JobStat &q = hs[j];
                q.n_rows = 0; q.fill_n = 0; q.n_near = q.n_reeval = q.n_unres = q.n_spike = 0;
                for (int t3 = 0; t3 < 3; ++t3) { q.m[t3] = 0; q.s1[t3] = 0; q.s2[t3] = 0; }
            }
            PAV_HIP(ctx, hipMemcpyAsync(d_stat, hs.data(), sizeof(JobStat) * n_jobs, hipMemcpyHostToDevice, st));
            D->results.assign(n_jobs, pav_den_result{});
            D->runs.assign(n_jobs, {});
        }
    }
    { const int rcc = compact_and_read(); if (rcc != PAV_OK) return rcc; }
    lap("kmer+compact");
    std::vector<uint8_t> in_x(n_jobs, 0);                              // the job's HBM table lives in keys_x / cnt_x
    {
        std::vector<uint32_t> overflow, exceed;
        for (uint32_t j = 0; j < n_jobs; ++j) {
            if (hs[j].lds_flags & LDS_OVERFLOW) overflow.push_back(j);
            else if (hs[j].lds_flags & LDS_EXCEED) exceed.push_back(j);
        }
        if (!overflow.empty() || !exceed.empty()) {
            // Both sets get an HBM table of their region (keys_x / cnt_x):
            //  * overflow - a partition list or LDS table of the region overflowed: low-complexity sequence sends thousands of
            //    copies of one k-mer to one partition (or, in theory, a hash imbalance far beyond the 0.44 load aimed at).
            //    These regions alone are redone with the HBM-table kernels; every other region keeps its rows.
            //  * exceed - failure path (scripts/density.py:516-527): the exact largest count and its k-mer.
            uint64_t extra = 0;
            for (const std::vector<uint32_t> *set : {&overflow, &exceed})
                for (uint32_t j : *set) {
                    JobDev &jd = D->h_jobs[j];
                    const uint32_t cap = pow2_at_least(2ull * jd.ref_len + 2);
                    jd.ht_off = extra; jd.ht_mask = cap - 1; extra += cap;
                    in_x[j] = 1;
                }
            for (uint32_t j : overflow) D->h_jobs[j].n_parts = 0;
            PAV_HIP(ctx, D->keys_x.reserve(8 * extra));
            PAV_HIP(ctx, D->cnt_x.reserve(4 * extra));
            PAV_HIP(ctx, hipMemsetAsync(D->keys_x.p, 0xFF, 8 * extra, st));
            PAV_HIP(ctx, hipMemsetAsync(D->cnt_x.p, 0, 4 * extra, st));
            PAV_HIP(ctx, hipMemcpyAsync(D->jobs.p, D->h_jobs.data(), sizeof(JobDev) * n_jobs, hipMemcpyHostToDevice, st));
            if (!overflow.empty()) {
                // statistics: the overflowed regions start from zero, the compaction results of all regions are recomputed
                for (uint32_t j : overflow) hs[j] = JobStat{};
                for (uint32_t j = 0; j < n_jobs; ++j) {
                    JobStat &q = hs[j];
                    q.n_rows = 0; q.fill_n = 0;
                    for (int t = 0; t < 3; ++t) { q.m[t] = 0; q.s1[t] = 0; q.s2[t] = 0; }
                }
                PAV_HIP(ctx, hipMemcpyAsync(d_stat, hs.data(), sizeof(JobStat) * n_jobs, hipMemcpyHostToDevice, st));
            }
            for (const std::vector<uint32_t> *set : {&overflow, &exceed})
                for (uint32_t j : *set) {
                    const JobDev &jd = D->h_jobs[j];
                    const uint32_t blocks_r = (uint32_t)((std::max<uint64_t>(jd.ref_len, 1) + DTILE - 1) / DTILE * (DTILE / 256));
                    if (set == &exceed) PAV_HIP(ctx, hipMemsetAsync(&d_stat[j].n_ref_valid, 0, 4, st));     // counted again by the insert
                    PAV_LAUNCH(ctx, "k_ref_insert", k_ref_insert, blocks_r, 256, 0, d_jobs, d_tjr, RV, k, pp->max_ref_kmer_count,
                               D->keys_x.as<unsigned long long>(), D->cnt_x.as<uint32_t>(), d_stat, (uint32_t)(jd.rpos_off / 256), 1);
                    if (set == &overflow) {
                        const uint32_t blocks_t = (uint32_t)((std::max<uint64_t>(jd.tig_len, 1) + DTILE - 1) / DTILE * (DTILE / 256));
                        PAV_LAUNCH(ctx, "k_tig_state", k_tig_state, blocks_t, 256, 0, d_jobs, d_tjt, TV, k,
                                   D->keys_x.as<unsigned long long>(), D->st_tmp.as<int8_t>(), d_stat, (uint32_t)(jd.tpos_off / 256));
                    }
                }
            if (!overflow.empty()) {
                const int rcc = compact_and_read();
                if (rcc != PAV_OK) return rcc;
            } else {
                const int rcs = read_stats();
                if (rcs != PAV_OK) return rcs;
            }
        }
    }

    uint64_t total_rows = 0;
    for (uint32_t j = 0; j < n_jobs; ++j) total_rows += hs[j].n_rows;
    std::vector<std::vector<RunDev>> mer_runs;                         // [job * 3 + state]
    if (want_runs) {
        if (ev_overflow) {                                             // more events than the buffer held: walk the column again
            int rc = collect_heads(D->state_mer.as<int8_t>(), mev, total_rows);
            if (rc != PAV_OK) return rc;
        } else {
            // k_compact_scatter's events: a tile's first row is a head only when the state really changes there; every job's
            // events end with a marker at its row count (what k_heads writes)
            std::vector<HeadEvent> kept;
            kept.reserve(mev.size() + n_jobs);
            for (size_t e = 0; e < mev.size(); ++e) {
                const HeadEvent &h = mev[e];
                const bool same_job = !kept.empty() && kept.back().job == h.job;
                if (!kept.empty() && !same_job) kept.push_back(HeadEvent{kept.back().job, hs[kept.back().job].n_rows, -2, 0u, 0u, 0u});
                if (same_job && kept.back().state == h.state) continue;
                kept.push_back(h);
            }
            if (!kept.empty()) kept.push_back(HeadEvent{kept.back().job, hs[kept.back().job].n_rows, -2, 0u, 0u, 0u});
            mev.swap(kept);
        }
        mer_runs.assign((size_t)n_jobs * 3, {});
        for (size_t e = 0; e + 1 < mev.size(); ++e) {
            const HeadEvent &h = mev[e];
            if (h.state < 0 || h.state > 2) continue;                 // end markers
            mer_runs[(size_t)h.job * 3 + h.state].push_back(RunDev{h.row, mev[e + 1].row - 1});
        }
    }
    lap("mer_runs");
    D->h_kde.assign(n_jobs, JobKde{});
    std::vector<EvalTile> tiles;
    uint64_t total_samp = 0;
    for (uint32_t j = 0; j < n_jobs; ++j) {
        pav_den_result &r = D->results[j];
        JobKde &kd = D->h_kde[j];
        const JobStat &s = hs[j];
        r.n_rows = s.n_rows;
        for (int q = 0; q < 3; ++q) r.state_count[q] = s.m[q];
        r.max_count = s.max_count;
        if (s.n_ref_valid == 0) { r.status = PAV_DEN_FAIL; r.fail_kind = 1; r.n_rows = 0; continue; }          // density.py:510-513
        if (s.max_count > pp->max_ref_kmer_count) { r.status = PAV_DEN_FAIL; r.fail_kind = 2; r.n_rows = 0; continue; }   // :516-527
        if (s.n_rows < pp->min_informative || s.n_rows == 0) { r.status = PAV_DEN_UNFINALISED; continue; }     // :193-195
        r.status = PAV_DEN_OK;
        const uint32_t n = s.n_rows;
        kd.finalised = 1; kd.n = n; kd.srs = D->h_jobs[j].srs;
        kd.n_samp = (n + kd.srs - 1) / kd.srs;
        if ((uint64_t)(kd.n_samp - 1) * kd.srs != n - 1) kd.n_samp += 1;                                         // :213-214
        r.n_sample = kd.n_samp;
        kd.samp_off = (uint32_t)total_samp;
        total_samp += kd.n_samp;
        const double bandwidth = std::pow((double)n, -1.0 / 5.0) * pp->den_smooth;                               // :198
        for (int q = 0; q < 3; ++q) {
            const uint64_t m = s.m[q];
            kd.m[q] = (uint32_t)m;
            kd.cnt[q] = (double)m;
            if (m == 0) continue;
            // unbiased variance of the integer row numbers, exact numerator (m*S2 - S1^2) / (m*(m-1))
            const unsigned __int128 num = (unsigned __int128)m * s.s2[q] - (unsigned __int128)s.s1[q] * s.s1[q];
            const double var = (double)num / ((double)m * (double)(m - 1));
            const double h = std::sqrt(var) * bandwidth;                 // cho_cov = cholesky(cov) * factor
            r.h[q] = h;
            kd.inv_h[q] = 1.0 / h;
            kd.norm[q] = std::pow(2 * 3.14159265358979323846, -0.5) / h;
            kd.w[q] = 1.0 / (double)m;
        }
        for (int q = 0; q < 3; ++q) kd.h[q] = r.h[q];
        for (uint32_t f = 0; f < kd.n_samp; f += 64) tiles.push_back(EvalTile{j, f, std::min<uint32_t>(64, kd.n_samp - f), 0});
    }
    // run arena for the closed-form evaluation
    if (want_runs) {
        std::vector<RunDev> arena;
        for (uint32_t j = 0; j < n_jobs; ++j) {
            JobKde &kd = D->h_kde[j];
            if (!kd.finalised) continue;
            const size_t nr = mer_runs[3 * j].size() + mer_runs[3 * j + 1].size() + mer_runs[3 * j + 2].size();
            if (nr > KDE_RUNS_MAX) continue;                           // very fragmented region: direct kernel
            kd.use_runs = 1;
            for (int q = 0; q < 3; ++q) {
                kd.run_off[q] = (uint32_t)arena.size();
                kd.n_run[q] = (uint32_t)mer_runs[3 * j + q].size();
                arena.insert(arena.end(), mer_runs[3 * j + q].begin(), mer_runs[3 * j + q].end());
            }
        }
        PAV_HIP(ctx, D->run_arena.reserve(sizeof(RunDev) * (arena.size() + 1)));
        if (!arena.empty())
            PAV_HIP(ctx, hipMemcpyAsync(D->run_arena.p, arena.data(), sizeof(RunDev) * arena.size(), hipMemcpyHostToDevice, st));
        PAV_HIP(ctx, hipStreamSynchronize(st));                        // `arena` is a local buffer
    }
    // failure path: the k-mer named in the message of scripts/density.py:519-526
    for (uint32_t j = 0; j < n_jobs; ++j) {
        if (D->results[j].fail_kind != 2) continue;
        PAV_HIP(ctx, hipMemsetAsync(&d_stat[j].max_key, 0xFF, sizeof(unsigned long long), st));
        // regions with LDS sets got their HBM table above (keys_x / cnt_x)
        const unsigned long long *jk = in_x[j] ? D->keys_x.as<unsigned long long>() : d_keys;
        const uint32_t *jc = in_x[j] ? D->cnt_x.as<uint32_t>() : D->cnt.as<uint32_t>();
        PAV_LAUNCH(ctx, "k_max_kmer", k_max_kmer, 64, 256, 0, d_jobs, j, RV, k, jk, jc, d_stat);
        unsigned long long packed = 0, key = 0;
        PAV_HIP(ctx, hipMemcpyAsync(&packed, &d_stat[j].max_key, sizeof packed, hipMemcpyDeviceToHost, st));
        PAV_HIP(ctx, hipStreamSynchronize(st));
        if ((packed & 0xFFFFFFFFull) == 0xFFFFFFFFull) key = ~0ull;       // k = 32: the k-mer that is not in the table (JobStat::ones_count)
        else PAV_HIP(ctx, hipMemcpy(&key, jk + D->h_jobs[j].ht_off + (packed & 0xFFFFFFFFull), sizeof key, hipMemcpyDeviceToHost));
        // the set holds rc(k-mer) when -r is set; the counter in the reference is keyed by the forward k-mer
        D->results[j].max_kmer = D->h_jobs[j].ref_rc ? pav_kmer_rev_complement(key, k) : key;
    }
    const JobKde *d_kde = D->kde.as<JobKde>();
    if (total_samp > 0xFFFFFFFFull) return fail(ctx, PAV_E_LIMIT, "pav_density_batch: too many sampled sites in one batch");
    for (int s = 0; s < 3; ++s) PAV_HIP(ctx, D->ks[s].reserve(8 * (total_samp + 1)));
    PAV_HIP(ctx, D->ss.reserve(total_samp + 1));

    // near-tie guard (include/pav_amd.h): flags per sampled site / table row, the list of sites to evaluate again
    GuardArgs G{};
    G.rel = pp->guard_rel == 0.0 ? GUARD_REL_DEFAULT : pp->guard_rel;
    G.unres = GUARD_UNRESOLVED;
    G.cap = pp->guard_cap ? pp->guard_cap : GUARD_CAP_DEFAULT;
    G.stat = d_stat;
    if (G.rel > 0.0 && !tiles.empty()) {
        PAV_HIP(ctx, D->guard.reserve(sizeof(GuardDev)));
        PAV_HIP(ctx, D->guard_entries.reserve(8ull * G.cap));
        PAV_HIP(ctx, D->samp_flag.reserve(4 * (total_samp + 2)));
        PAV_HIP(ctx, D->row_flag.reserve(a_t));
        G.g = D->guard.as<GuardDev>(); G.entries = D->guard_entries.as<unsigned long long>();
        G.samp_flag = D->samp_flag.as<uint32_t>(); G.row_flag = D->row_flag.as<uint8_t>();
    }
    PAV_HIP(ctx, D->scratch.reserve(4ull * (a_t / 256 + 1)));         // spike counts per block of k_finalize
    GuardDev h_guard{};
    auto read_guard = [&](uint8_t *dst) -> int {                        // queued in front of a synchronisation that follows
        if (G.rel > 0.0 && G.g) PAV_HIP(ctx, hipMemcpyAsync(dst, G.g, sizeof(GuardDev), hipMemcpyDeviceToHost, st));
        return PAV_OK;
    };
    uint8_t *h_guard_pin = h_pin + pin_guard_off;

    KdeArgs KA;
    KA.jobs = d_jobs; KA.kde = d_kde; KA.fill_list = D->fill_list.as<uint32_t>(); KA.state = D->state.as<int8_t>();
    KA.runs = D->run_arena.as<RunDev>(); KA.dyn = 0; KA.n_tiles_dev = nullptr; KA.n_tiles = 0;
    for (int s = 0; s < 3; ++s) { KA.list[s] = D->list[s].as<uint32_t>(); KA.kern[s] = D->kern[s].as<double>(); KA.ks[s] = D->ks[s].as<double>(); }
    KA.ss = D->ss.as<int8_t>();
    RedoArgs RA;
    RA.jobs = d_jobs; RA.kde = d_kde; RA.ss = D->ss.as<int8_t>(); RA.win_fill = D->win_fill.as<uint8_t>();
    for (int s = 0; s < 3; ++s) { RA.list[s] = D->list[s].as<uint32_t>(); RA.ks[s] = D->ks[s].as<double>(); RA.kern[s] = D->kern[s].as<double>(); }

    FinArgs FA;
    FA.jobs = d_jobs; FA.tile_job = d_tjt; FA.kde = d_kde; FA.stat = d_stat; FA.win_fill = D->win_fill.as<uint8_t>();
    FA.state = D->state.as<int8_t>(); FA.index = D->index.as<uint32_t>();
    for (int s = 0; s < 3; ++s) { FA.ks[s] = D->ks[s].as<double>(); FA.kern[s] = D->kern[s].as<double>(); }
    std::vector<HeadEvent> ev;
    std::vector<EvalTile> ftiles;                                      // alive until the synchronisation of collect_heads
    if (!tiles.empty()) {
        PAV_HIP(ctx, D->tiles.reserve(sizeof(EvalTile) * tiles.size()));
        PAV_HIP(ctx, hipMemcpyAsync(D->tiles.p, tiles.data(), sizeof(EvalTile) * tiles.size(), hipMemcpyHostToDevice, st));
    }
    // The density stage runs once in the regular case.  It is repeated with every job summed term by term (force_direct) only
    // when the guard's list overflowed.
    bool fallback = false;
    double work_add[3] = {0, 0, 0};
    for (int attempt = 0; attempt < 2; ++attempt) {
        const bool force_direct = attempt == 1;
        for (uint32_t j = 0; j < n_jobs; ++j) {                        // same choice as k_kde_eval makes per state
            JobKde &kd = D->h_kde[j];
            if (!kd.finalised) continue;
            if (force_direct) kd.use_runs = 0;
            kd.ps_mask = 0; kd.all_direct = kd.use_runs ? 0 : 1;       // scipy's order throughout: only without run sums
            for (int q = 0; q < 3; ++q) {
                if (!kd.m[q]) continue;
                if (!(kd.use_runs && kd.h[q] >= KDE_RUNS_MIN_H)) kd.ps_mask |= 1u << q;
            }
        }
        if (attempt == 0) lap("kde host");
        PAV_HIP(ctx, hipMemcpyAsync(D->kde.p, D->h_kde.data(), sizeof(JobKde) * n_jobs, hipMemcpyHostToDevice, st));
        if (tiles.empty()) break;
        if (G.g) {
            PAV_HIP(ctx, hipMemsetAsync(G.g, 0, sizeof(GuardDev), st));
            PAV_HIP(ctx, hipMemsetAsync(G.samp_flag, 0, 4 * (total_samp + 2), st));
            PAV_HIP(ctx, hipMemsetAsync(G.row_flag, 0, a_t, st));
        }
        if (force_direct) {                                            // counters of the abandoned attempt
            for (uint32_t j = 0; j < n_jobs; ++j) { JobStat &q = hs[j]; q.fill_n = 0; q.n_near = q.n_reeval = q.n_unres = q.n_spike = 0; }
            PAV_HIP(ctx, hipMemcpyAsync(d_stat, hs.data(), sizeof(JobStat) * n_jobs, hipMemcpyHostToDevice, st));
        }
        G.pass = 0;
        KA.G = G;
        KA.tiles = D->tiles.as<EvalTile>(); KA.n_tiles = (uint32_t)tiles.size();
        PAV_LAUNCH(ctx, "k_kde_eval", k_kde_eval, (uint32_t)tiles.size(), 64 * KDE_WAVES, 0, KA);
        uint32_t processed = 0;                                        // list entries whose sampled sites have been evaluated again
        bool overflow = false;
        for (uint32_t pass = 0; ; ++pass) {
            if (pass > 64) {                                             // windows flipping between quiet and filled: does not settle
                if (force_direct) return fail(ctx, PAV_E_STATE, "pav_density_batch: the near-tie guard did not settle in direct mode");
                overflow = true;                                         // same way out as a full list: everything in scipy's order
                break;
            }
            G.pass = pass;
            KA.G = G; RA.G = G;
            // ---- windows between sampled sites: interpolate or queue for evaluation (readback 2: fill counts) ----------
            if (pass) PAV_LAUNCH(ctx, "k_reset_fill", k_reset_fill, (n_jobs + 255) / 256, 256, 0, d_stat, n_jobs);
            PAV_LAUNCH(ctx, "k_windows", k_windows, (uint32_t)tiles.size(), 64, 0, d_jobs, D->tiles.as<EvalTile>(), d_kde, D->state_mer.as<int8_t>(),
                       D->ss.as<int8_t>(), D->ks[0].as<double>(), D->ks[1].as<double>(), D->ks[2].as<double>(),
                       pp->state_run_delta, D->fill_list.as<uint32_t>(), D->win_fill.as<uint8_t>(), d_stat, G);
            if (pass == 0) lap("kde queue 1");
            { const int rcs = read_stats(); if (rcs != PAV_OK) return rcs; }
            if (pass == 0) lap("kde eval 1");
            ftiles.clear();
            for (uint32_t j = 0; j < n_jobs; ++j) {
                if (!D->h_kde[j].finalised) continue;
                D->results[j].n_eval = (uint64_t)D->h_kde[j].n_samp + hs[j].fill_n;
                for (uint32_t f = 0; f < hs[j].fill_n; f += 64) ftiles.push_back(EvalTile{j, f, std::min<uint32_t>(64, hs[j].fill_n - f), 1});
            }
            if (pass == 0) {                                             // work counters (pav_kde_work)
                double pairs = 0, points = 0, big = 0, data_pairs = 0, rows_all = 0, rows_norev = 0;
                for (uint32_t j = 0; j < n_jobs; ++j) {
                    const JobKde &kd = D->h_kde[j];
                    if (!kd.finalised) continue;
                    rows_all += kd.n; if (hs[j].m[2] == 0) rows_norev += kd.n;
                    const double pts = (double)kd.n_samp + hs[j].fill_n, runs = (double)kd.n_run[0] + kd.n_run[1] + kd.n_run[2];
                    points += pts; pairs += pts * runs;
                    data_pairs += pts * ((double)hs[j].m[0] + hs[j].m[1] + hs[j].m[2]);
                    if (runs > 256) big += pts * runs;
                }
                work_add[0] = points; work_add[1] = pairs; work_add[2] = data_pairs;   // of the attempt that is kept (added below)
                if (timing) fprintf(stderr, "[pav timing]   kde work: %.3g evaluation points, %.3g (point, run) pairs (%.1f runs per point; %.0f %% of the pairs in jobs with > 256 runs); %.3g table rows, %.0f %% in jobs without REV k-mers\n",
                        points, pairs, points ? pairs / points : 0.0, pairs ? 100.0 * big / pairs : 0.0, rows_all, rows_all ? 100.0 * rows_norev / rows_all : 0.0);
            }
            if (!ftiles.empty()) {
                PAV_HIP(ctx, D->ftiles.reserve(sizeof(EvalTile) * ftiles.size()));
                PAV_HIP(ctx, hipMemcpyAsync(D->ftiles.p, ftiles.data(), sizeof(EvalTile) * ftiles.size(), hipMemcpyHostToDevice, st));
                KA.tiles = D->ftiles.as<EvalTile>(); KA.n_tiles = (uint32_t)ftiles.size();
                PAV_LAUNCH(ctx, "k_kde_eval", k_kde_eval, (uint32_t)ftiles.size(), 64 * KDE_WAVES, 0, KA);
            }
            if (processed) {                                           // rows queued earlier: scipy's order overrides the run sums
                RA.first = 0; RA.kinds = 2;
                PAV_LAUNCH(ctx, "k_redo", k_redo, processed, 64, 0, RA);
            }
            { const int rce = prepare_events(total_rows); if (rce != PAV_OK) return rce; }
            FA.G = G; FA.ev_cap = ev_cap; FA.events = D->events.as<HeadEvent>(); FA.ev_count = D->ev_count.as<uint32_t>();
            FA.blk_spike = G.rel > 0.0 ? D->scratch.as<uint32_t>() : nullptr;
            PAV_LAUNCH(ctx, "k_finalize", k_finalize, (uint32_t)(a_t / 256), 256, 0, FA);
            if (G.rel > 0.0)
                PAV_LAUNCH(ctx, "k_spike_sum", k_spike_sum, n_jobs, 256, 0, d_jobs, d_kde, D->scratch.as<uint32_t>(), d_stat);
            if (pass == 0) lap("kde");
            // ---- rl_encoder: run heads -> host (readback 3, with the guard's counters) -----------------------------------
            { const int rcg = read_guard(h_guard_pin); if (rcg != PAV_OK) return rcg; }
            { const int rcq = queue_stats(); if (rcq != PAV_OK) return rcq; }      // guard counters of every job ride along
            { const int rch = read_events(D->state.as<int8_t>(), ev); if (rch != PAV_OK) return rch; }
            take_stats();
            if (!G.g) break;
            memcpy(&h_guard, h_guard_pin, sizeof h_guard);
            if (h_guard.overflow || h_guard.n_entries > G.cap) { overflow = true; break; }
            if (h_guard.n_entries == processed) break;                 // nothing new was flagged: settled
            // sampled sites flagged in this pass are evaluated in scipy's order; rows are applied behind the fill evaluation
            RA.first = processed; RA.kinds = 1;
            PAV_LAUNCH(ctx, "k_redo", k_redo, h_guard.n_entries - processed, 64, 0, RA);
            processed = h_guard.n_entries;
        }
        if (!overflow) break;
        if (force_direct) return fail(ctx, PAV_E_STATE, "pav_density_batch: guard list overflow in direct mode");
        fallback = true;
    }
    for (int q = 0; q < 3; ++q) ctx->kde_work[q] += work_add[q];
    if (G.g) {                                                         // guard counters of every job (read with the run heads)
        for (uint32_t j = 0; j < n_jobs; ++j) {
            pav_den_result &r = D->results[j];
            r.n_near_tie = hs[j].n_near; r.n_reeval = hs[j].n_reeval; r.n_unresolved = hs[j].n_unres; r.n_spike_near = hs[j].n_spike;
            r.guard_fallback = fallback ? 1 : 0;
        }
    }
    if (tiles.empty()) {
        const int rch = collect_heads(D->state.as<int8_t>(), ev, total_rows);
        if (rch != PAV_OK) return rch;
    }
    for (size_t e = 0; e < ev.size(); ++e) {
        const HeadEvent &h = ev[e];
        if (h.state == -2 || D->results[h.job].status == PAV_DEN_FAIL) continue;
        const HeadEvent &nx = ev[e + 1];                               // next head or the end marker of the same job
        D->runs[h.job].push_back(pav_run{h.state, nx.row - h.row, (int64_t)h.index, (int64_t)nx.prev_index});
    }
    for (uint32_t j = 0; j < n_jobs; ++j) {
        D->results[j].n_runs = (uint32_t)D->runs[j].size();
        results[j] = D->results[j];
    }
    lap("rl");
    D->valid = true;
    return PAV_OK;
}

int pav_density_runs(pav_ctx *ctx, uint32_t job, pav_run *runs) {
    if (!ctx) return PAV_E_ARG;
    DensityState *D = dstate(ctx);
    if (!D->valid || job >= D->n_jobs) return fail(ctx, PAV_E_STATE, "pav_density_runs: no such job in the last batch");
    if (!D->runs[job].empty()) {
        if (!runs) return PAV_E_ARG;
        memcpy(runs, D->runs[job].data(), sizeof(pav_run) * D->runs[job].size());
    }
    return PAV_OK;
}

int pav_density_table(pav_ctx *ctx, uint32_t job, int64_t *index, int8_t *state_mer, int8_t *state, double *kern_fwd,
                      double *kern_fwdrev, double *kern_rev, uint64_t *kmer) {
    if (!ctx) return PAV_E_ARG;
    DensityState *D = dstate(ctx);
    if (job < D->no_table.size() && D->no_table[job]) return fail(ctx, PAV_E_STATE, "the table of job %u was not built (scan-only batch, a region with FWD k-mers only)", job);
    if (!D->valid || job >= D->n_jobs) return fail(ctx, PAV_E_STATE, "pav_density_table: no such job in the last batch");
    PAV_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t n = D->results[job].n_rows;
    if (n == 0 || D->results[job].status == PAV_DEN_FAIL) return PAV_OK;
    const uint64_t off = D->h_jobs[job].tpos_off;
    hipStream_t st = ctx->stream;
    std::vector<uint32_t> idx32;
    if (index) {
        idx32.resize(n);
        PAV_HIP(ctx, hipMemcpyAsync(idx32.data(), D->index.as<uint32_t>() + off, 4ull * n, hipMemcpyDeviceToHost, st));
    }
    if (state_mer) PAV_HIP(ctx, hipMemcpyAsync(state_mer, D->state_mer.as<int8_t>() + off, n, hipMemcpyDeviceToHost, st));
    if (state) PAV_HIP(ctx, hipMemcpyAsync(state, D->state.as<int8_t>() + off, n, hipMemcpyDeviceToHost, st));
    if (kmer) PAV_HIP(ctx, hipMemcpyAsync(kmer, D->kmer.as<uint64_t>() + off, 8ull * n, hipMemcpyDeviceToHost, st));
    double *outs[3] = {kern_fwd, kern_fwdrev, kern_rev};
    const bool fin = D->results[job].status == PAV_DEN_OK;
    for (int s = 0; s < 3; ++s)
        if (outs[s] && fin) PAV_HIP(ctx, hipMemcpyAsync(outs[s], D->kern[s].as<double>() + off, 8ull * n, hipMemcpyDeviceToHost, st));
    PAV_HIP(ctx, hipStreamSynchronize(st));
    if (index) for (uint32_t i = 0; i < n; ++i) index[i] = idx32[i];
    if (!fin) for (int s = 0; s < 3; ++s) if (outs[s]) for (uint32_t i = 0; i < n; ++i) outs[s][i] = NAN;
    return PAV_OK;
}

int pav_density_annotate(pav_ctx *ctx, uint32_t job, uint32_t ref_id, uint64_t ref_up_pos, uint64_t ref_up_end,
                         uint64_t ref_dn_pos, uint64_t ref_dn_end, int64_t qry_index_base, int64_t tig_up_pos,
                         int64_t tig_up_end, int64_t tig_dn_pos, int64_t tig_dn_end, uint8_t *flank, uint8_t *match) {
    if (!ctx || !flank || !match) return PAV_E_ARG;
    DensityState *D = dstate(ctx);
    if (job < D->no_table.size() && D->no_table[job]) return fail(ctx, PAV_E_STATE, "the table of job %u was not built (scan-only batch, a region with FWD k-mers only)", job);
    if (!D->valid || job >= D->n_jobs) return fail(ctx, PAV_E_STATE, "pav_density_annotate: no such job in the last batch");
    const SeqStore &RS = ctx->seq[PAV_ROLE_REF];
    if (ref_id >= RS.n || ref_up_end > RS.len[ref_id] || ref_dn_end > RS.len[ref_id])
        return fail(ctx, PAV_E_ARG, "pav_density_annotate: region outside the reference record");
    PAV_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t n = D->results[job].n_rows;
    if (n == 0) return PAV_OK;
    const int k = D->params.k;
    // Region(chrom, pos, end) swaps reversed coordinates (pavlib/seq.py:54-66); an empty region has no k-mers
    if (ref_up_pos > ref_up_end) std::swap(ref_up_pos, ref_up_end);
    if (ref_dn_pos > ref_dn_end) std::swap(ref_dn_pos, ref_dn_end);
    const uint32_t up_len = (uint32_t)(ref_up_end - ref_up_pos), dn_len = (uint32_t)(ref_dn_end - ref_dn_pos);
    const uint32_t cap_up = pow2_at_least(2ull * up_len + 2), cap_dn = pow2_at_least(2ull * dn_len + 2);
    hipStream_t st = ctx->stream;
    PAV_HIP(ctx, D->scratch.reserve(8ull * (cap_up + cap_dn) + 2ull * n + 64));
    unsigned long long *k_up = D->scratch.as<unsigned long long>(), *k_dn = k_up + cap_up;
    uint8_t *d_flank = reinterpret_cast<uint8_t *>(k_dn + cap_dn), *d_match = d_flank + n;
    PAV_HIP(ctx, hipMemsetAsync(k_up, 0xFF, 8ull * (cap_up + cap_dn), st));
    { int rcw = wait_planes(ctx); if (rcw != PAV_OK) return rcw; }
    const SeqView RV = RS.view();
    if (up_len >= (uint32_t)k)
        PAV_LAUNCH(ctx, "k_canon_insert", k_canon_insert, (up_len + 255) / 256, 256, 0, RV, RS.off[ref_id] + ref_up_pos, up_len, k, k_up, cap_up - 1);
    if (dn_len >= (uint32_t)k)
        PAV_LAUNCH(ctx, "k_canon_insert", k_canon_insert, (dn_len + 255) / 256, 256, 0, RV, RS.off[ref_id] + ref_dn_pos, dn_len, k, k_dn, cap_dn - 1);
    PAV_LAUNCH(ctx, "k_annotate", k_annotate, (n + 255) / 256, 256, 0, D->index.as<uint32_t>(), D->kmer.as<unsigned long long>(),
               D->h_jobs[job].tpos_off, n, k, qry_index_base, tig_up_pos, tig_up_end, tig_dn_pos, tig_dn_end, k_up, cap_up - 1, k_dn,
               cap_dn - 1, d_flank, d_match);
    PAV_HIP(ctx, hipMemcpyAsync(flank, d_flank, n, hipMemcpyDeviceToHost, st));
    PAV_HIP(ctx, hipMemcpyAsync(match, d_match, n, hipMemcpyDeviceToHost, st));
    PAV_HIP(ctx, hipStreamSynchronize(st));
    return PAV_OK;
}

}  // extern "C"
