// Inversion-signature flagging: rules call_inv_cluster, call_inv_flag_insdel_cluster and
// call_inv_merge_flagged_loci of rules/call_inv.snakefile:321-692.
//
// The reference walks every SNV / indel row with DataFrame.iterrows(); here the per-row work is data parallel:
//   * cluster sweep: "does row j open a new cluster" depends only on rows j-1 and j (the open cluster's end is always the
//     previous row's midpoint, :653-672), so every row that closes a cluster finds its opening row by a backward search
//     (serial for short clusters, one wave per cluster for long ones) and reports the cluster if it qualifies;
//   * INS / DEL matching: the interval-tree query of :517-534 becomes two binary searches over the DELs sorted by
//     (chrom, POS) and a running maximum of END (the rank in the upper bits makes the maximum restart per chromosome).
// The merges of the resulting few thousand intervals are sequential by definition and run on the host, quirks included.
#include "common.h"

#include <chrono>

#include <rocprim/rocprim.hpp>

#include <algorithm>

namespace pav {

const std::vector<std::string> &seq_names(pav_ctx *ctx, int role);   // invscan.cpp

namespace {

constexpr int CM_SHIFT = 40;                                  // cluster key = chrom rank << 40 | midpoint
constexpr uint64_t CM_MID = (1ull << CM_SHIFT) - 1;

struct ClusterHit { uint64_t start; int64_t pos, end, count; uint32_t chrom, pad; };   // 40 B
struct MatchHit { uint64_t key; int64_t end; };                                        // key = chrom << 32 | POS

struct FlagState {
    DevBuf a, b, c, d, tmp, hits, cnt, small, start;
    DevBuf split[2][4];                                       // pav_cigar_flag: DEL keys / ENDs, INS keys / lengths of the two vartypes
    DevBuf hits_b[4];                                         // pav_cigar_flag: hits of the four branches (snv / indel sweep, two matches)
    std::vector<pav_flag_rgn> table[4];                       // insdel_sv, insdel_indel, cluster_indel, cluster_snv
    std::vector<pav_flag_rgn> single;                         // result of the array-level entry points
    std::vector<pav_flag_locus> loci;
    // device-planned pav_cigar_flag: INS / DEL rows of both vartypes, pinned result block, what the small inputs held last time
    DevBuf plan, delmax;                                      // delmax: tile maxima of the running maximum (k_delmax_*)
    void *pin = nullptr;
    std::vector<uint8_t> small_seen;
    const void *small_at = nullptr;
};

FlagState *fstate(pav_ctx *ctx) {
    if (!ctx->flag) ctx->flag = new FlagState();
    return static_cast<FlagState *>(ctx->flag);
}

// ---- cluster sweep ---------------------------------------------------------------------------------------------------

// True when row j (>= 1) does not join the cluster of row j - 1 (:653).
__device__ __forceinline__ bool opens_cluster(const uint64_t *__restrict__ cm, uint64_t j, int64_t win) {
    const uint64_t a = cm[j - 1], b = cm[j];
    return (a >> CM_SHIFT) != (b >> CM_SHIFT) || (int64_t)(b & CM_MID) >= (int64_t)(a & CM_MID) + win;
}

// The cluster a row belongs to opens at the last row <= it that does not join its predecessor: an inclusive maximum
// scan over (row opens a cluster ? row : 0).  The reference's sequential loop (:646-684) in two data-parallel passes.
struct OpenRow {
    const uint64_t *cm;
    int64_t win;
    __device__ unsigned long long operator()(unsigned long long i) const { return (i == 0 || opens_cluster(cm, i, win)) ? i : 0ull; }
};

__global__ __launch_bounds__(256) void k_cluster_emit(const uint64_t *__restrict__ cm, const unsigned long long *__restrict__ start,
                                                      uint64_t n, int64_t win, int64_t win_min, int64_t min_count,
                                                      ClusterHit *__restrict__ hits, unsigned long long *__restrict__ n_hits, uint64_t cap) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !(i + 1 == n || opens_cluster(cm, i + 1, win))) return;       // only the last row of a cluster reports
    const uint64_t j = start[i];
    const int64_t count = (int64_t)(i - j) + 1;
    const int64_t pos = (int64_t)(cm[j] & CM_MID), end = (int64_t)(cm[i] & CM_MID);
    if (count >= min_count && end - pos >= win_min) {
        const unsigned long long slot = atomicAdd(n_hits, 1ull);
        if (slot < cap) hits[slot] = ClusterHit{j, pos, end, count, (uint32_t)(cm[i] >> CM_SHIFT), 0};
    }
}

// ---- keys from the device-resident call records ---------------------------------------------------------------------


// One atomic per workgroup for a per-thread partial count: thousands of waves adding to one word serialise at ~11 ns each
// (k_indel_keys spent 0.18 of its 0.20 ms there with one atomic per wave).  All 256 threads of the block must call it.
__device__ __forceinline__ void block_add(unsigned long long v, unsigned long long *counter, uint32_t *block_total = nullptr) {
    __shared__ unsigned long long part[4];
    for (int o = WAVE / 2; o; o >>= 1) v += __shfl_down(v, o);
    __syncthreads();                                                    // `part` may still be read from a previous call
    if ((threadIdx.x & (WAVE - 1)) == 0) part[threadIdx.x / WAVE] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = part[0] + part[1] + part[2] + part[3];
        if (t) atomicAdd(counter, t);
        if (block_total) *block_total = (uint32_t)t;
    }
}

// The key kernels work on tiles of KEY_TILE rows, KEY_PER per lane (all their loads issued before the first use): keys (sentinel = row filtered out) go to keys[], the number of real keys of the tile to tile_cnt[] - what the
// compaction below needs.  (A grid-stride loop with one 16-byte load in flight per lane ran at a third of this rate, and
// rocprim::select needed 0.09 ms for the 6.5 M SNV keys.)
constexpr int KEY_PER = 8, KEY_TILE = 256 * KEY_PER;

// cluster key of every SNV row: rank << 40 | POS (the midpoint of [POS, POS + 1) is POS).
__global__ __launch_bounds__(256) void k_snv_keys(const pav_snv *__restrict__ snv, uint64_t n, const pav_aln *__restrict__ aln,
                                                  const uint16_t *__restrict__ rank, const long long *__restrict__ tpos,
                                                  const long long *__restrict__ tend, unsigned long long *__restrict__ keys,
                                                  unsigned long long *__restrict__ n_pass, unsigned long long sentinel,
                                                  uint32_t *__restrict__ tile_cnt) {
    const uint64_t i0 = (uint64_t)blockIdx.x * KEY_TILE + threadIdx.x;     // rows i0, i0 + 256, ...: consecutive lanes, consecutive rows
    pav_snv s[KEY_PER];
#pragma unroll
    for (int u = 0; u < KEY_PER; ++u) s[u] = snv[i0 + 256 * u < n ? i0 + 256 * u : n - 1];
    unsigned long long mine = 0;
#pragma unroll
    for (int u = 0; u < KEY_PER; ++u) {
        const uint64_t i = i0 + 256 * u;
        if (i >= n) break;
        const bool pass = (long long)s[u].pos > tpos[s[u].aln] && (long long)s[u].pos + 1 < tend[s[u].aln];
        keys[i] = pass ? ((unsigned long long)rank[aln[s[u].aln].ref_id] << CM_SHIFT | s[u].pos) : sentinel;
        mine += pass;
    }
    block_add(mine, n_pass, tile_cnt + blockIdx.x);
}

// cluster key of every indel row < 50 bp: rank << 38 | POS << 6 | (END - POS): the (#CHROM, POS, END) order of the table.
__global__ __launch_bounds__(256) void k_indel_keys(const pav_indel *__restrict__ ind, uint64_t n, const pav_aln *__restrict__ aln,
                                                    const uint16_t *__restrict__ rank, const long long *__restrict__ tpos,
                                                    const long long *__restrict__ tend, unsigned long long *__restrict__ keys,
                                                    unsigned long long *__restrict__ counters, unsigned long long tag,
                                                    unsigned long long sentinel, uint32_t *__restrict__ tile_cnt) {
    const uint64_t i0 = (uint64_t)blockIdx.x * KEY_TILE + threadIdx.x;
    unsigned long long n_pass = 0, n_small = 0;
    uint32_t pos[KEY_PER], end[KEY_PER], svlen[KEY_PER], al[KEY_PER];
#pragma unroll
    for (int u = 0; u < KEY_PER; ++u) {                                  // 64-byte records: only the fields the key needs
        const pav_indel &v = ind[i0 + 256 * u < n ? i0 + 256 * u : n - 1];
        pos[u] = v.pos; end[u] = v.end; svlen[u] = v.svlen; al[u] = v.aln;
    }
#pragma unroll
    for (int u = 0; u < KEY_PER; ++u) {
        const uint64_t i = i0 + 256 * u;
        if (i >= n) break;
        const bool pass = (long long)pos[u] > tpos[al[u]] && (long long)end[u] < tend[al[u]];
        const bool small = pass && svlen[u] < 50;
        keys[i] = small ? (tag | (unsigned long long)rank[aln[al[u]].ref_id] << 38 | (unsigned long long)pos[u] << 6 | (end[u] - pos[u]))
                             : sentinel;
        n_pass += pass;
        n_small += small;
    }
    block_add(n_pass, counters);
    block_add(n_small, counters + 1, tile_cnt + blockIdx.x);
}

// Exclusive scan of the tile counts of one table per workgroup (blockIdx.x = table): tile_off[t] = real keys before tile t.
struct TileScanArgs { const uint32_t *cnt[2]; uint64_t *off[2]; uint32_t n_tiles[2]; };
__global__ __launch_bounds__(1024) void k_key_tile_scan(TileScanArgs A) {
    __shared__ uint64_t wsum[16];
    const uint32_t *cnt = A.cnt[blockIdx.x];
    uint64_t *off = A.off[blockIdx.x];
    const uint32_t n = A.n_tiles[blockIdx.x], per = (n + 1023) / 1024;
    const uint32_t t0 = threadIdx.x * per, t1 = t0 + per < n ? t0 + per : n;
    uint64_t mine = 0;
    for (uint32_t t = t0; t < t1; ++t) mine += cnt[t];
    uint64_t inc = mine;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint64_t y = __shfl_up(inc, d); if (lane >= d) inc += y; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint64_t base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    uint64_t run = base + inc - mine;
    for (uint32_t t = t0; t < t1; ++t) { off[t] = run; run += cnt[t]; }
}

// Stable compaction of the real keys (table order) of both tables: blockIdx.x < tiles[0] -> SNV tile, else indel tile.
struct CompactKeysArgs {
    const unsigned long long *in[2]; unsigned long long *out[2]; const uint64_t *off[2]; uint64_t n[2]; uint32_t tiles0;
    unsigned long long low;                                             // RealKey
};
__global__ __launch_bounds__(256) void k_compact_keys(CompactKeysArgs A) {
    __shared__ uint32_t wsum[4];
    const int t = blockIdx.x < A.tiles0 ? 0 : 1;
    const uint32_t tile = blockIdx.x - (t ? A.tiles0 : 0u);
    const uint64_t n = A.n[t], i0 = (uint64_t)tile * KEY_TILE + (uint64_t)threadIdx.x * KEY_PER;
    const unsigned long long *in = A.in[t];
    unsigned long long k[KEY_PER];
    uint32_t mine = 0;
    if (i0 + KEY_PER <= n) {
        typedef unsigned long long u64x2_a8 __attribute__((ext_vector_type(2), aligned(8)));   // the indel keys start at an odd key when n_snv is odd
        const u64x2_a8 *p = reinterpret_cast<const u64x2_a8 *>(in + i0);
#pragma unroll
        for (int u = 0; u < KEY_PER / 2; ++u) { const u64x2_a8 v = p[u]; k[2 * u] = v.x; k[2 * u + 1] = v.y; }
    } else {
#pragma unroll
        for (int u = 0; u < KEY_PER; ++u) k[u] = i0 + u < n ? in[i0 + u] : ~0ull;   // ~0: a sentinel for every table
    }
#pragma unroll
    for (int u = 0; u < KEY_PER; ++u) mine += (k[u] & A.low) != A.low;
    uint32_t inc = mine;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(inc, d); if (lane >= d) inc += y; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (int w = 0; w < 4; ++w) { if (w < wave) base += wsum[w]; total += wsum[w]; }
    // the tile's real keys are put in order in LDS and leave with coalesced stores (a lane storing its up to eight keys one
    // after the other kept the store queue full: the waves were issue-stalled two thirds of their time)
    __shared__ unsigned long long stage[KEY_TILE];
    uint32_t at = base + inc - mine;
#pragma unroll
    for (int u = 0; u < KEY_PER; ++u) if ((k[u] & A.low) != A.low) stage[at++] = k[u];
    __syncthreads();
    unsigned long long *out = A.out[t] + A.off[t][tile];
    for (uint32_t r = threadIdx.x; r < total; r += 256) out[r] = stage[r];
}

__global__ __launch_bounds__(256) void k_indel_mid(unsigned long long *__restrict__ keys, uint64_t n, unsigned long long tag) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = keys[i] & ~tag;
    const unsigned long long pos = (k >> 6) & 0xffffffffull, dlen = k & 63;
    keys[i] = (k >> 38) << CM_SHIFT | ((2 * pos + dlen) >> 1);            // (END + POS) // 2, :643
}

// INS queries and DEL intervals of one vartype (svlen in [lo, hi)), FILTER == PASS.
__global__ __launch_bounds__(256) void k_insdel_split(const pav_indel *__restrict__ ind, uint64_t n, const pav_aln *__restrict__ aln,
                                                      const uint16_t *__restrict__ rank, const long long *__restrict__ tpos,
                                                      const long long *__restrict__ tend, uint32_t svlen_lo, uint32_t svlen_hi,
                                                      unsigned long long *__restrict__ del_key, unsigned long long *__restrict__ del_end,
                                                      unsigned long long *__restrict__ ins_key, unsigned long long *__restrict__ ins_len,
                                                      unsigned long long *__restrict__ counters) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & (WAVE - 1);
    pav_indel v{};
    bool keep = false;
    if (i < n) {
        v = ind[i];
        keep = (long long)v.pos > tpos[v.aln] && (long long)v.end < tend[v.aln] && v.svlen >= svlen_lo && v.svlen < svlen_hi;
    }
    const unsigned long long r = keep ? rank[aln[v.aln].ref_id] : 0;
    // block-aggregated append (the DEL rows are sorted afterwards and the INS order does not matter): ranks inside the block
    // from wave ballots + the waves' totals in LDS, one returning atomic per block and kind
    __shared__ uint32_t wtot[2][4];
    __shared__ unsigned long long base[2];
    const int wave = threadIdx.x / WAVE;
    unsigned long long m[2];
    for (int t = 0; t < 2; ++t) {
        const bool mine = keep && v.svtype == (t == 0 ? 1 : 0);
        m[t] = __ballot(mine);
        if (lane == 0) wtot[t][wave] = (uint32_t)__popcll(m[t]);
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        const uint32_t tot = wtot[threadIdx.x][0] + wtot[threadIdx.x][1] + wtot[threadIdx.x][2] + wtot[threadIdx.x][3];
        base[threadIdx.x] = tot ? atomicAdd(counters + threadIdx.x, (unsigned long long)tot) : 0ull;
    }
    __syncthreads();
    for (int t = 0; t < 2; ++t) {
        const bool mine = keep && v.svtype == (t == 0 ? 1 : 0);
        if (!mine) continue;
        uint32_t before = 0;
        for (int w = 0; w < wave; ++w) before += wtot[t][w];
        const unsigned long long s = base[t] + before + __popcll(m[t] & ((1ull << lane) - 1));
        if (t == 0) { del_key[s] = r << 32 | v.pos; del_end[s] = r << 32 | v.end; }
        else { ins_key[s] = r << 32 | v.pos; ins_len[s] = v.svlen; }
    }
}

// first index in [lo, hi) with a[i] >= x
__device__ __forceinline__ uint64_t lower_bound_u64(const unsigned long long *__restrict__ a, uint64_t lo, uint64_t hi, unsigned long long x) {
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (a[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// del_key sorted; del_max[i] = max over rows <= i of (rank << 32 | END).  One INS per thread (:520-534).
__global__ __launch_bounds__(256) void k_ins_match(const unsigned long long *__restrict__ ins_key, const unsigned long long *__restrict__ ins_len,
                                                   uint64_t n_ins, const unsigned long long *__restrict__ del_key,
                                                   const unsigned long long *__restrict__ del_max, uint64_t n_del, long long flank_cluster,
                                                   MatchHit *__restrict__ hits, unsigned long long *__restrict__ n_hits) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_ins) return;
    const unsigned long long r = ins_key[i] >> 32;
    const long long pos = (long long)(ins_key[i] & 0xffffffffull), flank = (long long)ins_len[i] * flank_cluster;
    const long long lo = pos - flank, hi = pos + flank;        // DEL overlaps when DEL.POS < hi and DEL.END > lo
    if (hi <= lo || hi <= 0) return;
    const uint64_t seg = lower_bound_u64(del_key, 0, n_del, r << 32);
    const uint64_t j = hi > 0xffffffffll ? lower_bound_u64(del_key, seg, n_del, (r + 1) << 32)
                                         : lower_bound_u64(del_key, seg, n_del, r << 32 | (unsigned long long)hi);
    if (j == seg) return;
    const long long max_end = (long long)(del_max[j - 1] & 0xffffffffull);
    if (max_end <= lo) return;
    uint64_t first = seg;
    if (lo >= 0) first = lower_bound_u64(del_max, seg, j, (r << 32 | (unsigned long long)lo) + 1);   // first running max > lo
    const unsigned long long s = atomicAdd(n_hits, 1ull);
    hits[s] = MatchHit{r << 32 | (del_key[first] & 0xffffffffull), max_end};
}

// ---- host: the sequential merges ---------------------------------------------------------------------------------------

// :538-596.  Sorted by (chrom, POS); the interval open when the loop ends is never appended.
void merge_matches(std::vector<MatchHit> &m, int64_t flank_merge, std::vector<pav_flag_rgn> &out) {
    out.clear();
    std::sort(m.begin(), m.end(), [](const MatchHit &x, const MatchHit &y) { return x.key != y.key ? x.key < y.key : x.end < y.end; });
    bool have = false;
    uint32_t chrom = 0;
    int64_t pos = 0, end = 0;
    for (const MatchHit &h : m) {
        const uint32_t c = (uint32_t)(h.key >> 32);
        const int64_t p = (int64_t)(h.key & 0xffffffffull);
        if (!have || chrom != c) {
            if (have) out.push_back(pav_flag_rgn{chrom, 0, pos, end, 0});
            have = true; chrom = c; pos = p; end = h.end;
        }
        if (p - flank_merge <= end) {
            end = std::max(end, h.end);
        } else {
            out.push_back(pav_flag_rgn{chrom, 0, pos, end, 0});
            pos = p; end = h.end;
        }
    }
}

// :357-466
void merge_loci(const pav_flag_rgn *const tables[4], const uint64_t n[4], int64_t flank, int32_t batch_count, int32_t sig_filter,
                std::vector<pav_flag_locus> &out) {
    static const uint32_t type_of[4] = {PAV_FLAG_MATCH_SV, PAV_FLAG_MATCH_INDEL, PAV_FLAG_CLUSTER_INDEL, PAV_FLAG_CLUSTER_SNV};
    struct Row { uint32_t chrom, type; int64_t pos, end, n_indel, n_snv; };
    std::vector<Row> rows;
    for (int t = 0; t < 4; ++t)
        for (uint64_t i = 0; i < n[t]; ++i) {
            const pav_flag_rgn &r = tables[t][i];
            rows.push_back(Row{r.chrom, type_of[t], r.pos, r.end, t == 2 ? r.count : 0, t == 3 ? r.count : 0});
        }
    std::stable_sort(rows.begin(), rows.end(), [](const Row &x, const Row &y) { return x.chrom != y.chrom ? x.chrom < y.chrom : x.pos < y.pos; });
    out.clear();
    bool have_chrom = false;
    uint32_t chrom = 0, type = 0;
    int64_t pos = 0, end = 0, n_indel = 0, n_snv = 0;
    auto flush = [&]() { if (type) out.push_back(pav_flag_locus{chrom, type, pos, end, n_indel, n_snv, 0, -1}); };
    for (const Row &r : rows) {
        if (r.pos < end + flank && have_chrom && r.chrom == chrom) {
            type |= r.type;
            end = r.end;                                       // not the maximum (:397)
            n_indel += r.n_indel;
            n_snv += r.n_snv;
        } else {
            flush();
            type = r.type; pos = r.pos; end = r.end; chrom = r.chrom; have_chrom = true;
            n_indel = r.n_indel; n_snv = r.n_snv;
        }
    }
    flush();
    // df_merged.sort_values(['#CHROM', 'POS']) (:449): already in that order, and stable
    const bool allow_single = sig_filter == PAV_SIG_SINGLE_CLUSTER;
    const uint32_t match_any = sig_filter == PAV_SIG_SVINDEL ? (PAV_FLAG_MATCH_SV | PAV_FLAG_MATCH_INDEL)
                             : sig_filter == PAV_SIG_SV ? PAV_FLAG_MATCH_SV : 0u;
    int32_t batch = 0;
    for (pav_flag_locus &l : out) {
        bool ok = true;
        if (!allow_single && (l.type_mask == PAV_FLAG_CLUSTER_SNV || l.type_mask == PAV_FLAG_CLUSTER_INDEL)) ok = false;
        if (ok && match_any && !(l.type_mask & match_any)) ok = false;
        l.try_inv = ok;
        if (ok) { l.batch = batch; batch = (batch + 1) % batch_count; }
    }
}

// Keys in table order are already in the rules' iteration order when the alignment table is what get_align_bed writes
// (sorted by #CHROM as str, then POS - pavlib/align/align.py:280) and its rows do not overlap on the reference (trim-tigref):
// the walk emits the records of a row by rising position.  Then the sort is a stable compaction of the real keys.
struct RealKey {
    unsigned long long low;                  // tag - 1: every sentinel has all of these bits set, no real key has
    __device__ bool operator()(unsigned long long k) const { return (k & low) != low; }
};
// *unsorted |= the first *n_ptr keys do not rise
__global__ __launch_bounds__(256) void k_check_sorted(const unsigned long long *__restrict__ keys, const unsigned long long *__restrict__ n_ptr,
                                                      unsigned long long *__restrict__ unsorted) {
    const unsigned long long n = *n_ptr;
    bool bad = false;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i + 1 < n; i += (uint64_t)gridDim.x * 256) bad |= keys[i] > keys[i + 1];
    if (__ballot(bad) && (threadIdx.x & 63) == 0) *unsorted = 1;
}

// ---- pav_cigar_flag planned on the device (round 3) ----------------------------------------------------------------------
// Nothing in the stage waits for a count: every launch is sized by what the host knows beforehand (all SNV / INS-DEL rows), the
// real counts stay on the device and every kernel reads them there.  The four INS / DEL row sets are compacted in table order
// (tile counts from k_indel_keys_cls, one scan, one ordered scatter), so no sort is needed as long as the table rises; the two
// cluster sweeps are one scan + one emit over both key tables; the two matches one scan + one launch.  One synchronisation at
// the end.  A table that does not rise shows in that read-back (d_cnt[3]) and the stage runs again on the general path.

constexpr int CLS = 4;                                        // DEL sv, INS sv, DEL indel, INS indel (d_cnt words 8 .. 11)
constexpr uint32_t FIRST_HITS = 4096;                         // hits per list written straight to pinned host memory

struct FlagPin {                                              // the pinned result block
    unsigned long long cnt[16];
    ClusterHit sweep[2][FIRST_HITS];
    MatchHit match[2][FIRST_HITS];
};

__device__ __forceinline__ int insdel_class(bool pass, uint32_t svlen, uint32_t svtype, uint32_t lo_indel) {
    if (!pass) return -1;
    const int t = svlen >= 50 ? 0 : (svlen >= lo_indel ? 1 : -1);       // :497, :504-505
    return t < 0 ? -1 : 2 * t + (svtype == 0 ? 1 : 0);
}

// k_indel_keys + the tile's rows per INS / DEL class (four 16-bit fields of one word: a tile has 2048 rows)
__global__ __launch_bounds__(256) void k_indel_keys_cls(const pav_indel *__restrict__ ind, uint64_t n, const pav_aln *__restrict__ aln,
                                                        const uint16_t *__restrict__ rank, const long long *__restrict__ tpos,
                                                        const long long *__restrict__ tend, unsigned long long *__restrict__ keys,
                                                        unsigned long long *__restrict__ counters, unsigned long long tag,
                                                        unsigned long long sentinel, uint32_t *__restrict__ tile_cnt,
                                                        unsigned long long *__restrict__ tile_cls, uint32_t lo_indel) {
    const uint64_t i0 = (uint64_t)blockIdx.x * KEY_TILE + threadIdx.x;
    unsigned long long n_pass = 0, n_small = 0, cls = 0;
    uint32_t pos[KEY_PER], end[KEY_PER], svlen[KEY_PER], al[KEY_PER], ty[KEY_PER];
#pragma unroll
    for (int u = 0; u < KEY_PER; ++u) {
        const pav_indel &v = ind[i0 + 256 * u < n ? i0 + 256 * u : n - 1];
        pos[u] = v.pos; end[u] = v.end; svlen[u] = v.svlen; al[u] = v.aln; ty[u] = v.svtype;
    }
#pragma unroll
    for (int u = 0; u < KEY_PER; ++u) {
        const uint64_t i = i0 + 256 * u;
        if (i >= n) break;
        const bool pass = (long long)pos[u] > tpos[al[u]] && (long long)end[u] < tend[al[u]];
        const bool small = pass && svlen[u] < 50;
        keys[i] = small ? (tag | (unsigned long long)rank[aln[al[u]].ref_id] << 38 | (unsigned long long)pos[u] << 6 | (end[u] - pos[u]))
                             : sentinel;
        n_pass += pass;
        n_small += small;
        const int c = insdel_class(pass, svlen[u], ty[u], lo_indel);
        if (c >= 0) cls += 1ull << (16 * c);
    }
    block_add(n_pass, counters);
    block_add(n_small, counters + 1, tile_cnt + blockIdx.x);
    // the class fields cannot carry into each other (2048 rows per tile at most in all four together)
    __shared__ unsigned long long part[4];
    for (int o = WAVE / 2; o; o >>= 1) cls += __shfl_down(cls, o);
    if ((threadIdx.x & (WAVE - 1)) == 0) part[threadIdx.x / WAVE] = cls;
    __syncthreads();
    if (threadIdx.x == 0) tile_cls[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// k_key_tile_scan for six sequences: workgroups 0 / 1 the key tables, 2 .. 5 the INS / DEL classes (totals to cls_total)
struct TileScan6Args {
    const uint32_t *cnt[2]; uint64_t *off[2]; uint32_t n_tiles[2];
    const unsigned long long *tile_cls; uint32_t *cls_off[CLS]; unsigned long long *cls_total;
};
__global__ __launch_bounds__(1024) void k_key_tile_scan6(TileScan6Args A) {
    __shared__ uint64_t wsum[16];
    const int q = blockIdx.x, c = q - 2;
    const uint32_t n = A.n_tiles[q < 2 ? q : 1], per = (n + 1023) / 1024;
    const uint32_t t0 = threadIdx.x * per < n ? threadIdx.x * per : n, t1 = t0 + per < n ? t0 + per : n;
    auto at = [&](uint32_t t) -> uint64_t { return q < 2 ? A.cnt[q][t] : (A.tile_cls[t] >> (16 * c)) & 0xffffull; };
    uint64_t mine = 0;
    for (uint32_t t = t0; t < t1; ++t) mine += at(t);
    uint64_t inc = mine;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint64_t y = __shfl_up(inc, d); if (lane >= d) inc += y; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint64_t base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    uint64_t run = base + inc - mine;
    if (q < 2) for (uint32_t t = t0; t < t1; ++t) { A.off[q][t] = run; run += at(t); }
    else {
        for (uint32_t t = t0; t < t1; ++t) { A.cls_off[c][t] = (uint32_t)run; run += at(t); }
        if (threadIdx.x == 1023) A.cls_total[c] = run;
    }
}

// The rows of the four INS / DEL classes in table order: tile t writes behind cls_off[c][t].  Lane l owns rows 8 l .. 8 l + 7 of
// the tile (the 64-byte records are read field-wise either way: one line per record).
struct SplitArgs {
    const pav_indel *ind; uint64_t n; const pav_aln *aln; const uint16_t *rank; const long long *tpos, *tend;
    const uint32_t *cls_off[CLS];
    unsigned long long *key[CLS], *val[CLS];                  // DEL: rank << 32 | POS, rank << 32 | END; INS: rank << 32 | POS, SVLEN
    uint32_t lo_indel;
};
__global__ __launch_bounds__(256) void k_insdel_split_ordered(SplitArgs A) {
    __shared__ unsigned long long wsum[4];
    const uint64_t i0 = (uint64_t)blockIdx.x * KEY_TILE + (uint64_t)threadIdx.x * KEY_PER;
    uint32_t pos[KEY_PER], end[KEY_PER], svlen[KEY_PER], al[KEY_PER], ty[KEY_PER];
#pragma unroll
    for (int u = 0; u < KEY_PER; ++u) {
        const pav_indel &v = A.ind[i0 + u < A.n ? i0 + u : A.n - 1];
        pos[u] = v.pos; end[u] = v.end; svlen[u] = v.svlen; al[u] = v.aln; ty[u] = v.svtype;
    }
    int cls[KEY_PER];
    unsigned long long mine = 0;
#pragma unroll
    for (int u = 0; u < KEY_PER; ++u) {
        cls[u] = -1;
        if (i0 + u < A.n && svlen[u] >= (A.lo_indel < 50 ? A.lo_indel : 50u)) {
            const bool pass = (long long)pos[u] > A.tpos[al[u]] && (long long)end[u] < A.tend[al[u]];
            cls[u] = insdel_class(pass, svlen[u], ty[u], A.lo_indel);
        }
        if (cls[u] >= 0) mine += 1ull << (16 * cls[u]);
    }
    unsigned long long inc = mine;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long y = __shfl_up(inc, d); if (lane >= d) inc += y; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    if (!mine) return;
    unsigned long long before = inc - mine;
    for (int w = 0; w < wave; ++w) before += wsum[w];
#pragma unroll
    for (int u = 0; u < KEY_PER; ++u) {
        const int c = cls[u];
        if (c < 0) continue;
        const uint64_t s = (uint64_t)A.cls_off[c][blockIdx.x] + ((before >> (16 * c)) & 0xffffull);
        before += 1ull << (16 * c);
        const unsigned long long r = (unsigned long long)A.rank[A.aln[al[u]].ref_id] << 32;
        A.key[c][s] = r | pos[u];
        A.val[c][s] = (c & 1) ? (unsigned long long)svlen[u] : (r | end[u]);
    }
}

// Both compacted key tables as the sweep sees them: rows [0, n_snv) hold SNV keys (cnt[0] of them real), rows [n_snv, ...) the
// INS-DEL keys < 50 bp (cnt[2] real).  Rows behind the real ones are never read.
struct KeyView {
    const unsigned long long *k; uint64_t n_snv; unsigned long long tag; const unsigned long long *cnt;
    __device__ __forceinline__ bool real(uint64_t i, uint64_t &local, uint64_t &n_real) const {
        const bool ind = i >= n_snv;
        local = ind ? i - n_snv : i;
        n_real = cnt[ind ? 2 : 0];
        return local < n_real;
    }
    __device__ __forceinline__ void at(uint64_t i, uint32_t &chrom, int64_t &mid) const {
        unsigned long long x = k[i];
        if (i >= n_snv) {                                                     // k_indel_mid: (END + POS) // 2, :643
            x &= ~tag;
            chrom = (uint32_t)(x >> 38);
            mid = (int64_t)((2 * ((x >> 6) & 0xffffffffull) + (x & 63)) >> 1);
        } else { chrom = (uint32_t)(x >> CM_SHIFT); mid = (int64_t)(x & CM_MID); }
    }
    __device__ __forceinline__ bool opens(uint64_t i, int64_t win) const {   // rows i - 1 and i of one table, both real
        uint32_t ca, cb; int64_t ma, mb;
        at(i - 1, ca, ma); at(i, cb, mb);
        return ca != cb || mb >= ma + win;
    }
};
// Both cluster sweeps (rule call_inv_cluster, :646-684) without a scan over all rows.  k_cluster_sweep notes, for every 256 rows, the
// last row that opens a cluster, and lets the last row of each cluster find the cluster's first row (k_cluster_resolve for the long ones): it walks back
// while fewer than min_count rows are behind it (nearly every cluster is a single row and ends the walk at once; a cluster
// that short is not reported anyway), and the clusters that pass are finished by the whole wave - the rest of the row's own
// block 64 rows per step, then 64 block notes per step (16 K rows): the inverted stretches this stage exists to find make
// clusters of tens of thousands of rows.  (The max-scan over "row opens a cluster" this replaces wrote and re-read 8 bytes per
// row: 0.10 ms for the two launches of the scan and the emit.)
struct SweepArgs {
    KeyView V; uint64_t n; int64_t win, win_min, min_count[2]; uint32_t *last_open;
    ClusterHit *hits[2]; uint64_t cap[2]; unsigned long long *n_hits, *unsorted; FlagPin *pin;
    unsigned long long *pending; unsigned long long *n_pending; uint64_t pending_cap;   // closers of long clusters: (row, where the walk stands)
};
__device__ __forceinline__ void sweep_report(const SweepArgs &A, uint64_t i, uint64_t j) {
    const int t = i >= A.V.n_snv;
    const int64_t count = (int64_t)(i - j) + 1;
    uint32_t cj, ci; int64_t pos, end;
    A.V.at(j, cj, pos); A.V.at(i, ci, end);
    if (count >= A.min_count[t] && end - pos >= A.win_min) {
        const unsigned long long slot = atomicAdd(A.n_hits + t, 1ull);
        const ClusterHit h{j, pos, end, count, ci, 0};
        if (slot < A.cap[t]) A.hits[t][slot] = h;
        if (slot < FIRST_HITS) A.pin->sweep[t][slot] = h;
    }
}
// Pass 1: every row.  Leaves the block notes (last row of every 256 that opens a cluster - a by-product of the bits the rows need
// anyway), reports the clusters whose first row it finds within min_count rows, and hands the closers of longer clusters to pass 2.
__global__ __launch_bounds__(256) void k_cluster_sweep(SweepArgs A) {
    __shared__ uint32_t wbest[4];
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint64_t local = 0, n_real = 0;
    const bool real = i < A.n && A.V.real(i, local, n_real);
    const uint64_t first = i - local;                                         // first row of the table
    const int t = i >= A.V.n_snv;
    // the row, the one behind it and the one in front are asked for together: whether the row closes a cluster and whether it
    // opens one (a single-row cluster: nearly every row) are then answered by one round trip to memory, not by two in a row
    uint32_t c_prev = 0, c_own = 0, c_next = 0; int64_t m_prev = 0, m_own = 0, m_next = 0;
    const bool has_prev = real && local != 0, has_next = real && local + 1 != n_real;
    if (real) A.V.at(i, c_own, m_own);
    if (has_prev) { A.V.at(i - 1, c_prev, m_prev); if (A.V.k[i - 1] > A.V.k[i]) *A.unsorted = 1; }   // table order is not the rules' iteration order
    if (has_next) A.V.at(i + 1, c_next, m_next);
    const bool closes = real && (!has_next || c_own != c_next || m_next >= m_own + A.win);
    const bool opens_own = !has_prev || c_prev != c_own || m_own >= m_prev + A.win;
    // The rows of a wave are consecutive: where the row's cluster starts inside the wave is in the wave's "opens a cluster" bits -
    // no load, no walk (a third of the SNV rows sit in clusters of two to five).  Only a cluster that began before the wave's
    // first row is walked, and only while it is still too short to report.
    const unsigned long long ob = __ballot(real && opens_own);
    if (lane == 0) wbest[threadIdx.x >> 6] = ob ? (uint32_t)(threadIdx.x + 64 - __clzll((long long)ob)) : 0u;     // 1 + thread of the wave's last opening row
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t best = 0;
        for (int w = 0; w < 4; ++w) if (wbest[w]) best = wbest[w];
        A.last_open[blockIdx.x] = best;                                        // 1 + offset in the block, 0: no row of the block opens a cluster
    }
    if (!closes) return;
    const int64_t need = A.min_count[t] > 1 ? A.min_count[t] : 1;
    const unsigned long long upto = ob & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
    uint64_t j;
    int64_t behind;                                                            // rows j .. i
    bool start_known;
    if (upto) { const int sl = 63 - __clzll((long long)upto); j = i - (uint64_t)(lane - sl); behind = lane - sl + 1; start_known = true; }
    else { j = i - (uint64_t)lane; behind = lane + 1; start_known = false; }   // the wave's first row: it does not open the cluster
    while (!start_known && behind < need) {
        --j; ++behind;
        start_known = j == first || A.V.opens(j, A.win);
    }
    if (start_known) { if (behind >= need) sweep_report(A, i, j); return; }    // (a cluster with fewer rows is not reported)
    const unsigned long long slot = atomicAdd(A.n_pending, 1ull);              // long, and its first row not found yet: pass 2
    if (slot < A.pending_cap) { A.pending[2 * slot] = i; A.pending[2 * slot + 1] = j; }
}
// Pass 2: one wave per long cluster.  No row in (j, i] opens a cluster; the wave looks at the rest of j's block 64 rows per step,
// then at 64 block notes (16 K rows) per step - the inverted stretches this stage exists to find make clusters of 10^4 rows.
__global__ __launch_bounds__(64) void k_cluster_resolve(SweepArgs A) {
    const int lane = threadIdx.x;
    const unsigned long long n_pend = min(*A.n_pending, (unsigned long long)A.pending_cap);
    for (unsigned long long e = blockIdx.x; e < n_pend; e += gridDim.x) {
        const uint64_t i = A.pending[2 * e];
        uint64_t at = A.pending[2 * e + 1];
        uint64_t local, n_real;
        (void)A.V.real(i, local, n_real);
        const uint64_t lo = i - local;
        uint64_t got = ~0ull;
        const uint64_t blk0 = at & ~255ull;                                    // rest of the block of `at`
        while (got == ~0ull) {
            const bool valid = at >= lo + (uint64_t)lane && at - (uint64_t)lane >= blk0;
            const uint64_t row = at - (uint64_t)lane;
            const bool op = valid && (row == lo || A.V.opens(row, A.win));
            const unsigned long long m = __ballot(op);
            if (m) { got = at - (uint64_t)(__ffsll((long long)m) - 1); break; }
            if (at < blk0 + 64) break;
            at -= 64;
        }
        if (got == ~0ull) {                                                    // the blocks before: the nearest with a note holds the row
            uint64_t b = blk0 / 256;                                           // (the table's first row opens, so there is one)
            for (;;) {
                const bool valid = b >= (uint64_t)lane + 1;
                const uint32_t note = valid ? A.last_open[b - 1 - (uint64_t)lane] : 0u;
                const unsigned long long m = __ballot(note != 0);
                if (m) {
                    const int src = __ffsll((long long)m) - 1;
                    got = (b - 1 - (uint64_t)src) * 256 + (uint64_t)__shfl(note, src) - 1;
                    break;
                }
                b -= 64;
            }
        }
        if (lane == 0) sweep_report(A, i, got);
    }
}

// Input of the running maximum of rank << 32 | END over the DEL rows of both vartypes, one sequence: vartype 1 follows vartype 0
// n_ind rows later and carries bit 63, so the maximum restarts there; rows behind the real ones count as 0.  Looks at the order
// of the DEL keys on the way.
constexpr unsigned long long VT1 = 1ull << 63;
struct DelEndPlanned {
    const unsigned long long *key[2], *end[2]; uint64_t n_ind; const unsigned long long *cnt; unsigned long long *unsorted;
    __device__ unsigned long long operator()(unsigned long long i) const {
        const int t = i >= n_ind;
        const uint64_t local = t ? i - n_ind : i, n_real = cnt[8 + 2 * t];
        if (local >= n_real) return t ? VT1 : 0ull;
        if (local && key[t][local - 1] > key[t][local]) *unsorted = 1;
        return end[t][local] | (t ? VT1 : 0ull);
    }
};

// The running maximum itself, two launches (the planned path; rocprim::inclusive_scan - an init kernel and a look-back scan,
// 0.032 ms - did this in round 3): k_delmax_tiles reduces tiles of 2048 elements to their maxima; k_delmax_apply takes the
// maximum of the tiles in front of its own as carry-in, scans its tile on top of it and writes the values.  The input is
// read twice: 16 B per element, 1.2 M elements.
constexpr int DM_TILE = 2048;                                    // 8 rows of 256 consecutive elements, one per lane
struct DelMaxArgs { DelEndPlanned in; unsigned long long *out, *tile_max; uint64_t n; uint32_t n_tiles; };

__device__ __forceinline__ unsigned long long umax64(unsigned long long a, unsigned long long b) { return a > b ? a : b; }
__device__ __forceinline__ unsigned long long wave_incl_max(unsigned long long v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long y = (unsigned long long)__shfl_up((long long)v, d);
        if (lane >= d) v = umax64(v, y);
    }
    return v;
}

__global__ __launch_bounds__(256) void k_delmax_tiles(DelMaxArgs A) {
    __shared__ unsigned long long wmax[4];
    const uint64_t i0 = (uint64_t)blockIdx.x * DM_TILE + threadIdx.x;            // element q * 256 + lane of the tile: coalesced
    unsigned long long m = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) if (i0 + q * 256 < A.n) m = umax64(m, A.in(i0 + q * 256));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = umax64(m, (unsigned long long)__shfl_xor((long long)m, d));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) A.tile_max[blockIdx.x] = umax64(umax64(wmax[0], wmax[1]), umax64(wmax[2], wmax[3]));
}

__global__ __launch_bounds__(256) void k_delmax_apply(DelMaxArgs A) {
    __shared__ unsigned long long wsum[8][4];
    const uint64_t i0 = (uint64_t)blockIdx.x * DM_TILE + threadIdx.x;            // element q * 256 + lane of the tile
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long inc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const unsigned long long v = i0 + q * 256 < A.n ? A.in(i0 + q * 256) : 0ull;
        inc[q] = wave_incl_max(v);                                        // inclusive over the lanes of this wave, row q
        if (lane == 63) wsum[q][wave] = inc[q];
    }
    // the tile's carry-in: the maximum of the tiles in front of it (a few hundred values from the L2: every block reduces its own
    // prefix - a carry kernel between the two launches, or a last-block-done pass with its device-scope fences, costs more)
    __shared__ unsigned long long cmax[4];
    unsigned long long c = 0;
    for (uint32_t t = threadIdx.x; t < blockIdx.x; t += 256) c = umax64(c, A.tile_max[t]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c = umax64(c, (unsigned long long)__shfl_xor((long long)c, d));
    if (lane == 0) cmax[wave] = c;
    __syncthreads();
    unsigned long long before = umax64(umax64(cmax[0], cmax[1]), umax64(cmax[2], cmax[3]));   // everything in front of (row q, this wave)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        unsigned long long mine = before;
        for (int w = 0; w < wave; ++w) mine = umax64(mine, wsum[q][w]);
        if (i0 + q * 256 < A.n) A.out[i0 + q * 256] = umax64(mine, inc[q]);
        before = umax64(umax64(before, wsum[q][0]), umax64(umax64(wsum[q][1], wsum[q][2]), wsum[q][3]));
    }
}

// k_ins_match for both vartypes (first / second half of the grid), counts on the device; del_max from the scan above (vartype 1 carries VT1)
struct MatchArgs {
    const unsigned long long *ins_key[2], *ins_len[2], *del_key[2], *del_max[2]; const unsigned long long *cnt;
    long long flank_cluster; MatchHit *hits[2]; unsigned long long *n_hits; FlagPin *pin; uint32_t blocks_per_t;
};
__global__ __launch_bounds__(256) void k_ins_match_planned(MatchArgs A) {
    const int t = blockIdx.x >= A.blocks_per_t;
    const uint64_t i = (uint64_t)(blockIdx.x - (t ? A.blocks_per_t : 0u)) * 256 + threadIdx.x, n_del = A.cnt[8 + 2 * t], n_ins = A.cnt[9 + 2 * t];
    if (i >= n_ins || n_del == 0) return;
    const unsigned long long *del_key = A.del_key[t], *del_max = A.del_max[t], vt = t ? VT1 : 0ull;
    const unsigned long long r = A.ins_key[t][i] >> 32;
    const long long pos = (long long)(A.ins_key[t][i] & 0xffffffffull), flank = (long long)A.ins_len[t][i] * A.flank_cluster;
    const long long lo = pos - flank, hi = pos + flank;
    if (hi <= lo || hi <= 0) return;
    const uint64_t seg = lower_bound_u64(del_key, 0, n_del, r << 32);
    const uint64_t j = hi > 0xffffffffll ? lower_bound_u64(del_key, seg, n_del, (r + 1) << 32)
                                         : lower_bound_u64(del_key, seg, n_del, r << 32 | (unsigned long long)hi);
    if (j == seg) return;
    const long long max_end = (long long)(del_max[j - 1] & 0xffffffffull);
    if (max_end <= lo) return;
    uint64_t first = seg;
    if (lo >= 0) first = lower_bound_u64(del_max, seg, j, (vt | r << 32 | (unsigned long long)lo) + 1);
    const unsigned long long s = atomicAdd(A.n_hits + t, 1ull);
    const MatchHit h{r << 32 | (del_key[first] & 0xffffffffull), max_end};
    A.hits[t][s] = h;
    if (s < FIRST_HITS) A.pin->match[t][s] = h;
}

// ---- device drivers ----------------------------------------------------------------------------------------------------

int sort_keys(pav_ctx *ctx, FlagState *S, unsigned long long *in, unsigned long long *out, uint64_t n, unsigned end_bit) {
    size_t bytes = 0;
    PAV_HIP(ctx, rocprim::radix_sort_keys(nullptr, bytes, in, out, (size_t)n, 0, end_bit, ctx->stream));
    PAV_HIP(ctx, S->tmp.reserve(bytes + 16));
    const int tok = prof_begin(ctx, "rocprim::radix_sort_keys");
    const hipError_t e = rocprim::radix_sort_keys(S->tmp.p, bytes, in, out, (size_t)n, 0, end_bit, ctx->stream);
    prof_end(ctx, tok);
    PAV_HIP(ctx, e);
    return PAV_OK;
}

// Sweep over n device-resident cluster keys, queued on the stream: qualifying clusters are appended to d_hits (capacity
// sweep_cap), their number to *d_n_hits (zeroed by the caller).
uint64_t sweep_cap(uint64_t n, int64_t min_count) { return n / (uint64_t)std::max<int64_t>(min_count, 1) + 1; }

int sweep_launch(pav_ctx *ctx, FlagState *S, const unsigned long long *d_cm, uint64_t n, int64_t win, int64_t win_min, int64_t min_count,
                 ClusterHit *d_hits, unsigned long long *d_n_hits) {
    if (!n) return PAV_OK;
    PAV_HIP(ctx, S->start.reserve(8 * n));
    auto opens = rocprim::make_transform_iterator(rocprim::counting_iterator<unsigned long long>(0), OpenRow{(const uint64_t *)d_cm, win});
    size_t bytes = 0;
    PAV_HIP(ctx, rocprim::inclusive_scan(nullptr, bytes, opens, S->start.as<unsigned long long>(), (size_t)n,
                                         rocprim::maximum<unsigned long long>(), ctx->stream));
    PAV_HIP(ctx, S->tmp.reserve(bytes + 16));
    {
        const int tok = prof_begin(ctx, "rocprim::inclusive_scan");
        const hipError_t e = rocprim::inclusive_scan(S->tmp.p, bytes, opens, S->start.as<unsigned long long>(), (size_t)n,
                                                     rocprim::maximum<unsigned long long>(), ctx->stream);
        prof_end(ctx, tok);
        PAV_HIP(ctx, e);
    }
    PAV_LAUNCH(ctx, "k_cluster_emit", k_cluster_emit, (uint32_t)((n + 255) / 256), 256, 0, (const uint64_t *)d_cm,
               S->start.as<unsigned long long>(), n, win, win_min, min_count, d_hits, d_n_hits, sweep_cap(n, min_count));
    return PAV_OK;
}

// Hits copied off the device -> result rows in sweep order.
void sweep_collect(std::vector<ClusterHit> &hits, std::vector<pav_flag_rgn> &out) {
    std::sort(hits.begin(), hits.end(), [](const ClusterHit &x, const ClusterHit &y) { return x.start < y.start; });
    out.clear();
    out.reserve(hits.size());
    for (const ClusterHit &h : hits) out.push_back(pav_flag_rgn{h.chrom, 0, h.pos, h.end, h.count});
}

int run_sweep(pav_ctx *ctx, FlagState *S, const unsigned long long *d_cm, uint64_t n, int64_t win, int64_t win_min, int64_t min_count,
              std::vector<pav_flag_rgn> &out) {
    out.clear();
    if (!n) return PAV_OK;
    const uint64_t cap = sweep_cap(n, min_count);
    PAV_HIP(ctx, S->hits.reserve(sizeof(ClusterHit) * cap));
    PAV_HIP(ctx, S->cnt.reserve(64));
    PAV_HIP(ctx, hipMemsetAsync(S->cnt.p, 0, 64, ctx->stream));
    const int rc = sweep_launch(ctx, S, d_cm, n, win, win_min, min_count, S->hits.as<ClusterHit>(), S->cnt.as<unsigned long long>());
    if (rc != PAV_OK) return rc;
    unsigned long long n_hits = 0;
    PAV_HIP(ctx, hipMemcpyAsync(&n_hits, S->cnt.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    PAV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (n_hits > cap) return fail(ctx, PAV_E_LIMIT, "flag: cluster output overflow (%llu > %llu)", n_hits, (unsigned long long)cap);
    std::vector<ClusterHit> hits(n_hits);
    if (n_hits) PAV_HIP(ctx, hipMemcpy(hits.data(), S->hits.p, sizeof(ClusterHit) * n_hits, hipMemcpyDeviceToHost));
    sweep_collect(hits, out);
    return PAV_OK;
}

// DELs (unsorted keys / rank-tagged ENDs in k_in / v_in, buffers sized 2 x n_del) against INS (keys, lengths), queued on the
// stream: hits are appended to d_hits (capacity n_ins), their number to *d_n_hits (zeroed by the caller).
int match_launch(pav_ctx *ctx, FlagState *S, unsigned long long *k_in, unsigned long long *v_in, const unsigned long long *ins_key,
                 const unsigned long long *ins_len, uint64_t n_del, uint64_t n_ins, int64_t flank_cluster, MatchHit *d_hits,
                 unsigned long long *d_n_hits) {
    if (!n_del || !n_ins) return PAV_OK;
    hipStream_t st = ctx->stream;
    // sort DELs by (chrom, POS), carrying END; then the running maximum of rank << 32 | END
    unsigned long long *k_out = k_in + n_del, *v_out = v_in + n_del;
    size_t bytes = 0, bytes2 = 0;
    PAV_HIP(ctx, rocprim::radix_sort_pairs(nullptr, bytes, k_in, k_out, v_in, v_out, (size_t)n_del, 0, 48, st));
    PAV_HIP(ctx, rocprim::inclusive_scan(nullptr, bytes2, v_out, v_in, (size_t)n_del, rocprim::maximum<unsigned long long>(), st));
    PAV_HIP(ctx, S->tmp.reserve(std::max(bytes, bytes2) + 16));
    {
        const int tok = prof_begin(ctx, "rocprim::radix_sort_pairs");
        const hipError_t e = rocprim::radix_sort_pairs(S->tmp.p, bytes, k_in, k_out, v_in, v_out, (size_t)n_del, 0, 48, st);
        prof_end(ctx, tok);
        PAV_HIP(ctx, e);
    }
    {
        const int tok = prof_begin(ctx, "rocprim::inclusive_scan");
        const hipError_t e = rocprim::inclusive_scan(S->tmp.p, bytes2, v_out, v_in, (size_t)n_del, rocprim::maximum<unsigned long long>(), st);
        prof_end(ctx, tok);
        PAV_HIP(ctx, e);
    }
    PAV_LAUNCH(ctx, "k_ins_match", k_ins_match, (uint32_t)((n_ins + 255) / 256), 256, 0, ins_key, ins_len, n_ins, k_out, v_in, n_del,
               (long long)flank_cluster, d_hits, d_n_hits);
    return PAV_OK;
}

// DELs (S->a / S->b) against INS (S->c keys, S->d lengths); result merged on the host.
int run_match(pav_ctx *ctx, FlagState *S, uint64_t n_del, uint64_t n_ins, int64_t flank_cluster, int64_t flank_merge,
              std::vector<pav_flag_rgn> &out) {
    out.clear();
    if (!n_del || !n_ins) return PAV_OK;
    hipStream_t st = ctx->stream;
    PAV_HIP(ctx, S->hits.reserve(sizeof(MatchHit) * n_ins));
    PAV_HIP(ctx, S->cnt.reserve(64));
    PAV_HIP(ctx, hipMemsetAsync(S->cnt.p, 0, 64, st));
    const int rc = match_launch(ctx, S, S->a.as<unsigned long long>(), S->b.as<unsigned long long>(), S->c.as<unsigned long long>(),
                                S->d.as<unsigned long long>(), n_del, n_ins, flank_cluster, S->hits.as<MatchHit>(),
                                S->cnt.as<unsigned long long>());
    if (rc != PAV_OK) return rc;
    unsigned long long n_hits = 0;
    PAV_HIP(ctx, hipMemcpyAsync(&n_hits, S->cnt.p, 8, hipMemcpyDeviceToHost, st));
    PAV_HIP(ctx, hipStreamSynchronize(st));
    std::vector<MatchHit> hits(n_hits);
    if (n_hits) PAV_HIP(ctx, hipMemcpy(hits.data(), S->hits.p, sizeof(MatchHit) * n_hits, hipMemcpyDeviceToHost));
    merge_matches(hits, flank_merge, out);
    return PAV_OK;
}

// pav_cigar_flag planned on the device (see above).  d_cnt zeroed, trim table and ranks uploaded by the caller.  *again: the tables
// do not rise in table order - nothing was produced, the caller takes the general path.
int flag_planned(pav_ctx *ctx, FlagState *S, const pav_flag_params *P, unsigned long long *d_cnt, const long long *d_tp, const long long *d_te,
                 const uint16_t *d_rank, unsigned long long tag, unsigned long long pass_counts[2], bool *again) {
    hipStream_t st = ctx->stream;
    const uint64_t n_snv = ctx->counts.n_snv, n_ind = ctx->counts.n_indel, n_keys = n_snv + n_ind;
    const pav_aln *d_aln = ctx->d_aln.as<pav_aln>();
    *again = false;
    if (!S->pin) PAV_HIP(ctx, hipHostMalloc(&S->pin, sizeof(FlagPin), hipHostMallocDefault));
    FlagPin *pin = static_cast<FlagPin *>(S->pin);
    const uint32_t tiles_snv = (uint32_t)((n_snv + KEY_TILE - 1) / KEY_TILE), tiles_ind = (uint32_t)((n_ind + KEY_TILE - 1) / KEY_TILE);
    const uint32_t lo_indel = (uint32_t)std::min<int64_t>(std::max<int64_t>(P->insdel_min_svlen, 0), 0xffffffffll);

    PAV_HIP(ctx, S->a.reserve(8 * n_keys)); PAV_HIP(ctx, S->b.reserve(8 * n_keys));
    unsigned long long *k_in = S->a.as<unsigned long long>(), *k_sorted = S->b.as<unsigned long long>();
    // tile arrays: offsets of the key tiles, their counts, class word and class offsets of the INS-DEL tiles
    PAV_HIP(ctx, S->c.reserve(12 * ((size_t)tiles_snv + tiles_ind) + (8 + 4 * CLS) * (size_t)tiles_ind + 64));
    uint64_t *d_toff = S->c.as<uint64_t>();
    unsigned long long *d_tcls = reinterpret_cast<unsigned long long *>(d_toff + tiles_snv + tiles_ind);
    uint32_t *d_tcnt = reinterpret_cast<uint32_t *>(d_tcls + tiles_ind);
    uint32_t *d_cls_off = d_tcnt + tiles_snv + tiles_ind;
    // rows of the four classes (keys, values: n_ind each), then the running maximum over the DEL rows of both vartypes (2 n_ind)
    PAV_HIP(ctx, S->plan.reserve(8 * (2 * CLS + 2) * (size_t)n_ind + 64));
    unsigned long long *cls_key[CLS], *cls_val[CLS], *del_max = S->plan.as<unsigned long long>() + 2 * CLS * n_ind;
    for (int c = 0; c < CLS; ++c) { cls_key[c] = S->plan.as<unsigned long long>() + 2 * c * n_ind; cls_val[c] = cls_key[c] + n_ind; }
    const uint64_t cap[2] = {sweep_cap(n_snv, P->cluster_min_snv), sweep_cap(n_ind, P->cluster_min_indel)};
    PAV_HIP(ctx, S->hits_b[0].reserve(sizeof(ClusterHit) * cap[0])); PAV_HIP(ctx, S->hits_b[1].reserve(sizeof(ClusterHit) * cap[1]));
    PAV_HIP(ctx, S->hits_b[2].reserve(sizeof(MatchHit) * (n_ind + 1))); PAV_HIP(ctx, S->hits_b[3].reserve(sizeof(MatchHit) * (n_ind + 1)));

    const KeyView V{k_sorted, n_snv, tag, d_cnt};
    const DelEndPlanned DE{{cls_key[0], cls_key[2]}, {cls_val[0], cls_val[2]}, n_ind, d_cnt, d_cnt + 3};
    const uint32_t dm_tiles = (uint32_t)((2 * n_ind + DM_TILE - 1) / DM_TILE);      // the running maximum's tile maxima
    PAV_HIP(ctx, S->delmax.reserve(8ull * dm_tiles + 64));

    if (n_snv)
        PAV_LAUNCH(ctx, "k_snv_keys", k_snv_keys, tiles_snv, 256, 0, ctx->d_snv.as<pav_snv>(), n_snv, d_aln, d_rank, d_tp, d_te, k_in, d_cnt,
                   tag - 1, d_tcnt);
    if (n_ind)
        PAV_LAUNCH(ctx, "k_indel_keys_cls", k_indel_keys_cls, tiles_ind, 256, 0, ctx->d_indel.as<pav_indel>(), n_ind, d_aln, d_rank, d_tp, d_te,
                   k_in + n_snv, d_cnt + 1, tag, tag | (tag - 1), d_tcnt + tiles_snv, d_tcls, lo_indel);
    TileScan6Args TS;
    TS.cnt[0] = d_tcnt; TS.cnt[1] = d_tcnt + tiles_snv; TS.off[0] = d_toff; TS.off[1] = d_toff + tiles_snv;
    TS.n_tiles[0] = tiles_snv; TS.n_tiles[1] = tiles_ind; TS.tile_cls = d_tcls; TS.cls_total = d_cnt + 8;
    for (int c = 0; c < CLS; ++c) TS.cls_off[c] = d_cls_off + (size_t)c * tiles_ind;
    PAV_LAUNCH(ctx, "k_key_tile_scan6", k_key_tile_scan6, 2 + CLS, 1024, 0, TS);
    CompactKeysArgs CK;
    CK.in[0] = k_in; CK.in[1] = k_in + n_snv; CK.out[0] = k_sorted; CK.out[1] = k_sorted + n_snv;
    CK.off[0] = d_toff; CK.off[1] = d_toff + tiles_snv; CK.n[0] = n_snv; CK.n[1] = n_ind; CK.tiles0 = tiles_snv; CK.low = tag - 1;
    PAV_LAUNCH(ctx, "k_compact_keys", k_compact_keys, tiles_snv + tiles_ind, 256, 0, CK);
    if (n_ind) {
        SplitArgs SA;
        SA.ind = ctx->d_indel.as<pav_indel>(); SA.n = n_ind; SA.aln = d_aln; SA.rank = d_rank; SA.tpos = d_tp; SA.tend = d_te; SA.lo_indel = lo_indel;
        for (int c = 0; c < CLS; ++c) { SA.cls_off[c] = TS.cls_off[c]; SA.key[c] = cls_key[c]; SA.val[c] = cls_val[c]; }
        PAV_LAUNCH(ctx, "k_insdel_split_ordered", k_insdel_split_ordered, tiles_ind, 256, 0, SA);
    }
    {   // both cluster sweeps (rule call_inv_cluster)
        const uint32_t n_blocks = (uint32_t)((n_keys + 255) / 256);
        const uint64_t pend_cap = n_keys / 2 + 64;                             // (a long cluster has at least two rows)
        PAV_HIP(ctx, S->start.reserve(4ull * n_blocks + 64 + 16 * pend_cap));
        SweepArgs SW;
        SW.V = V; SW.n = n_keys; SW.win = SW.win_min = P->cluster_win; SW.last_open = S->start.as<uint32_t>();
        SW.pending = reinterpret_cast<unsigned long long *>(S->start.as<uint8_t>() + ((4ull * n_blocks + 63) & ~63ull));
        SW.n_pending = d_cnt + 6; SW.pending_cap = pend_cap;
        SW.min_count[0] = P->cluster_min_snv; SW.min_count[1] = P->cluster_min_indel;
        SW.hits[0] = S->hits_b[0].as<ClusterHit>(); SW.hits[1] = S->hits_b[1].as<ClusterHit>(); SW.cap[0] = cap[0]; SW.cap[1] = cap[1];
        SW.n_hits = d_cnt + 4; SW.unsorted = d_cnt + 3; SW.pin = pin;
        PAV_LAUNCH(ctx, "k_cluster_sweep", k_cluster_sweep, n_blocks, 256, 0, SW);
        PAV_LAUNCH(ctx, "k_cluster_resolve", k_cluster_resolve, 1024, 64, 0, SW);
    }
    if (n_ind) {   // both matches (rule call_inv_flag_insdel_cluster)
        DelMaxArgs DM;
        DM.in = DE; DM.out = del_max; DM.n = 2 * n_ind; DM.n_tiles = dm_tiles;
        DM.tile_max = S->delmax.as<unsigned long long>();
        PAV_LAUNCH(ctx, "k_delmax_tiles", k_delmax_tiles, dm_tiles, 256, 0, DM);
        PAV_LAUNCH(ctx, "k_delmax_apply", k_delmax_apply, dm_tiles, 256, 0, DM);
        MatchArgs MA;
        for (int t = 0; t < 2; ++t) {
            MA.del_key[t] = cls_key[2 * t]; MA.ins_key[t] = cls_key[2 * t + 1]; MA.ins_len[t] = cls_val[2 * t + 1];
            MA.del_max[t] = del_max + (size_t)t * n_ind; MA.hits[t] = S->hits_b[2 + t].as<MatchHit>();
        }
        MA.cnt = d_cnt; MA.flank_cluster = P->insdel_flank_cluster; MA.n_hits = d_cnt + 12; MA.pin = pin;
        MA.blocks_per_t = (uint32_t)((n_ind + 255) / 256);
        PAV_LAUNCH(ctx, "k_ins_match_planned", k_ins_match_planned, 2 * MA.blocks_per_t, 256, 0, MA);
    }
    PAV_HIP(ctx, hipMemcpyAsync(pin->cnt, d_cnt, 128, hipMemcpyDeviceToHost, st));
    PAV_HIP(ctx, hipStreamSynchronize(st));                                                 // the one synchronisation of the stage
    const unsigned long long *c = pin->cnt;
    if (c[3]) { *again = true; return PAV_OK; }
    pass_counts[0] = c[0]; pass_counts[1] = c[1];
    if (c[4] > cap[0] || c[5] > cap[1]) return fail(ctx, PAV_E_LIMIT, "flag: cluster output overflow (%llu / %llu hits)", c[4], c[5]);
    for (int q = 0; q < 2; ++q) {
        std::vector<ClusterHit> sh(c[4 + q]);
        if (c[4 + q] > FIRST_HITS) PAV_HIP(ctx, hipMemcpy(sh.data(), S->hits_b[q].p, sizeof(ClusterHit) * sh.size(), hipMemcpyDeviceToHost));
        else std::copy(pin->sweep[q], pin->sweep[q] + sh.size(), sh.begin());
        sweep_collect(sh, S->table[q == 0 ? 3 : 2]);
        std::vector<MatchHit> mh(c[12 + q]);
        if (c[12 + q] > FIRST_HITS) PAV_HIP(ctx, hipMemcpy(mh.data(), S->hits_b[2 + q].p, sizeof(MatchHit) * mh.size(), hipMemcpyDeviceToHost));
        else std::copy(pin->match[q], pin->match[q] + mh.size(), mh.begin());
        merge_matches(mh, P->insdel_flank_merge, S->table[q]);
    }
    return PAV_OK;
}

}  // namespace
}  // namespace pav

using namespace pav;

extern "C" {

void pav_flag_release(pav_ctx *ctx) {
    if (!ctx || !ctx->flag) return;
    FlagState *S = static_cast<FlagState *>(ctx->flag);
    DevBuf *bufs[] = {&S->a, &S->b, &S->c, &S->d, &S->tmp, &S->hits, &S->cnt, &S->small, &S->start,
                      &S->split[0][0], &S->split[0][1], &S->split[0][2], &S->split[0][3], &S->split[1][0], &S->split[1][1],
                      &S->split[1][2], &S->split[1][3], &S->hits_b[0], &S->hits_b[1], &S->hits_b[2], &S->hits_b[3], &S->plan, &S->delmax};
    for (DevBuf *b : bufs) b->release();
    if (S->pin) (void)hipHostFree(S->pin);
    delete S;
    ctx->flag = nullptr;
}

void pav_flag_params_default(pav_flag_params *p) {
    if (!p) return;
    p->cluster_win = 200;
    p->cluster_min_snv = 20;
    p->cluster_min_indel = 10;
    p->insdel_flank_cluster = 2;
    p->insdel_flank_merge = 2000;
    p->insdel_min_svlen = 4;
    p->merge_flank = 500;
    p->batch_count = 60;
    p->sig_filter = PAV_SIG_SVINDEL;
}

int pav_flag_cluster(pav_ctx *ctx, uint64_t n, const uint32_t *chrom, const int64_t *pos, const int64_t *end, int64_t win,
                     int64_t win_min, int64_t min_count, const pav_flag_rgn **out, uint64_t *n_out) {
    if (!ctx || !out || !n_out || (n && (!chrom || !pos || !end))) return PAV_E_ARG;
    PAV_HIP(ctx, hipSetDevice(ctx->device));
    FlagState *S = fstate(ctx);
    std::vector<unsigned long long> cm(n);
    for (uint64_t i = 0; i < n; ++i) {
        const int64_t s = end[i] + pos[i];
        const int64_t mid = s >= 0 ? s / 2 : -((-s + 1) / 2);                 // Python floor division
        if (mid < 0 || (uint64_t)mid > CM_MID || chrom[i] >= (1u << 24))
            return fail(ctx, PAV_E_LIMIT, "pav_flag_cluster: row %llu outside the supported coordinate range", (unsigned long long)i);
        cm[i] = (unsigned long long)chrom[i] << CM_SHIFT | (unsigned long long)mid;
    }
    if (n) {
        PAV_HIP(ctx, S->a.reserve(8 * n));
        PAV_HIP(ctx, hipMemcpyAsync(S->a.p, cm.data(), 8 * n, hipMemcpyHostToDevice, ctx->stream));
    }
    const int rc = run_sweep(ctx, S, S->a.as<unsigned long long>(), n, win, win_min, min_count, S->single);
    if (rc != PAV_OK) return rc;
    if (prof_flush(ctx) != PAV_OK) return PAV_E_HIP;
    *out = S->single.data();
    *n_out = S->single.size();
    return PAV_OK;
}

int pav_flag_insdel(pav_ctx *ctx, uint64_t n_ins, const uint32_t *ins_chrom, const int64_t *ins_pos, const int64_t *ins_svlen,
                    uint64_t n_del, const uint32_t *del_chrom, const int64_t *del_pos, const int64_t *del_end,
                    int64_t flank_cluster, int64_t flank_merge, const pav_flag_rgn **out, uint64_t *n_out) {
    if (!ctx || !out || !n_out || (n_ins && (!ins_chrom || !ins_pos || !ins_svlen)) || (n_del && (!del_chrom || !del_pos || !del_end)))
        return PAV_E_ARG;
    PAV_HIP(ctx, hipSetDevice(ctx->device));
    FlagState *S = fstate(ctx);
    const int64_t lim = 0xffffffffll;
    std::vector<unsigned long long> dk(n_del), de(n_del), ik(n_ins), il(n_ins);
    for (uint64_t i = 0; i < n_del; ++i) {
        if (del_pos[i] < 0 || del_pos[i] > lim || del_end[i] < 0 || del_end[i] > lim)
            return fail(ctx, PAV_E_LIMIT, "pav_flag_insdel: DEL row %llu outside the supported coordinate range", (unsigned long long)i);
        dk[i] = (unsigned long long)del_chrom[i] << 32 | (unsigned long long)del_pos[i];
        de[i] = (unsigned long long)del_chrom[i] << 32 | (unsigned long long)del_end[i];
    }
    for (uint64_t i = 0; i < n_ins; ++i) {
        if (ins_pos[i] < 0 || ins_pos[i] > lim || ins_svlen[i] < 0 || ins_svlen[i] > lim)
            return fail(ctx, PAV_E_LIMIT, "pav_flag_insdel: INS row %llu outside the supported coordinate range", (unsigned long long)i);
        ik[i] = (unsigned long long)ins_chrom[i] << 32 | (unsigned long long)ins_pos[i];
        il[i] = (unsigned long long)ins_svlen[i];
    }
    if (n_del && n_ins) {
        PAV_HIP(ctx, S->a.reserve(16 * n_del)); PAV_HIP(ctx, S->b.reserve(16 * n_del));
        PAV_HIP(ctx, S->c.reserve(8 * n_ins)); PAV_HIP(ctx, S->d.reserve(8 * n_ins));
        PAV_HIP(ctx, hipMemcpyAsync(S->a.p, dk.data(), 8 * n_del, hipMemcpyHostToDevice, ctx->stream));
        PAV_HIP(ctx, hipMemcpyAsync(S->b.p, de.data(), 8 * n_del, hipMemcpyHostToDevice, ctx->stream));
        PAV_HIP(ctx, hipMemcpyAsync(S->c.p, ik.data(), 8 * n_ins, hipMemcpyHostToDevice, ctx->stream));
        PAV_HIP(ctx, hipMemcpyAsync(S->d.p, il.data(), 8 * n_ins, hipMemcpyHostToDevice, ctx->stream));
    }
    const int rc = run_match(ctx, S, n_del, n_ins, flank_cluster, flank_merge, S->single);
    if (rc != PAV_OK) return rc;
    if (prof_flush(ctx) != PAV_OK) return PAV_E_HIP;
    *out = S->single.data();
    *n_out = S->single.size();
    return PAV_OK;
}

int pav_flag_merge_loci(pav_ctx *ctx, const pav_flag_rgn *const tables[4], const uint64_t n[4], int64_t flank, int32_t batch_count,
                        int32_t sig_filter, const pav_flag_locus **out, uint64_t *n_out) {
    if (!ctx || !tables || !n || !out || !n_out) return PAV_E_ARG;
    if (batch_count <= 0) return fail(ctx, PAV_E_ARG, "pav_flag_merge_loci: batch_count must be positive");
    if (sig_filter < PAV_SIG_SVINDEL || sig_filter > PAV_SIG_NONE) return fail(ctx, PAV_E_ARG, "pav_flag_merge_loci: unknown sig_filter %d", sig_filter);
    for (int t = 0; t < 4; ++t) if (n[t] && !tables[t]) return PAV_E_ARG;
    FlagState *S = fstate(ctx);
    merge_loci(tables, n, flank, batch_count, sig_filter, S->loci);
    *out = S->loci.data();
    *n_out = S->loci.size();
    return PAV_OK;
}

int pav_cigar_flag(pav_ctx *ctx, const int64_t *trim_pos, const int64_t *trim_end, const pav_flag_params *P, pav_flag_result *res) {
    if (!ctx || !P || !res) return PAV_E_ARG;
    const bool timing = getenv("PAV_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_mark = now();
    auto lap = [&](const char *what) { if (timing) { const double t = now(); fprintf(stderr, "[pav timing]   flag %-14s %.2f ms\n", what, (t - t_mark) * 1e3); t_mark = t; } };
    if (ctx->n_aln && (!trim_pos || !trim_end)) return PAV_E_ARG;
    if (!ctx->cigar_called) return fail(ctx, PAV_E_STATE, "pav_cigar_flag: no pav_cigar_call results on this context");
    if (P->batch_count <= 0 || P->sig_filter < PAV_SIG_SVINDEL || P->sig_filter > PAV_SIG_NONE) return fail(ctx, PAV_E_ARG, "pav_cigar_flag: bad parameters");
    PAV_HIP(ctx, hipSetDevice(ctx->device));
    FlagState *S = fstate(ctx);
    hipStream_t st = ctx->stream;
    const uint32_t n_aln = ctx->n_aln, n_ref = ctx->seq[PAV_ROLE_REF].n;
    const uint64_t n_snv = ctx->counts.n_snv, n_ind = ctx->counts.n_indel;
    const std::vector<std::string> &rnames = seq_names(ctx, PAV_ROLE_REF);
    if (rnames.size() != n_ref) return fail(ctx, PAV_E_STATE, "pav_cigar_flag: pav_seq_set_names has not been called for the reference store");
    if (n_ref > 65535) return fail(ctx, PAV_E_LIMIT, "pav_cigar_flag: more than 65535 reference records");
    memset(res, 0, sizeof(*res));
    for (auto &t : S->table) t.clear();
    S->loci.clear();

    // #CHROM compares as Python str: rank of each record name in byte order
    std::vector<uint32_t> by_name(n_ref);
    for (uint32_t i = 0; i < n_ref; ++i) by_name[i] = i;
    std::sort(by_name.begin(), by_name.end(), [&](uint32_t x, uint32_t y) { return rnames[x] < rnames[y]; });
    std::vector<uint16_t> rank(n_ref);
    for (uint32_t i = 0; i < n_ref; ++i)
        rank[by_name[i]] = (uint16_t)(i && rnames[by_name[i]] == rnames[by_name[i - 1]] ? rank[by_name[i - 1]] : i);

    // small inputs: counters, trim table, ranks
    PAV_HIP(ctx, S->small.reserve(128 + 16 * (size_t)n_aln + 2 * (size_t)n_ref + 64));
    unsigned long long *d_cnt = S->small.as<unsigned long long>();
    long long *d_tp = reinterpret_cast<long long *>(S->small.as<uint8_t>() + 128), *d_te = d_tp + n_aln;
    uint16_t *d_rank = reinterpret_cast<uint16_t *>(d_te + n_aln);
    PAV_HIP(ctx, hipMemsetAsync(d_cnt, 0, 128, st));
    // the trim table and the ranks go up when they differ from what is there (a caller that flags pass after pass sends the same)
    std::vector<uint8_t> up(16 * (size_t)n_aln + 2 * (size_t)n_ref);
    if (n_aln) { memcpy(up.data(), trim_pos, 8 * (size_t)n_aln); memcpy(up.data() + 8 * (size_t)n_aln, trim_end, 8 * (size_t)n_aln); }
    if (n_ref) memcpy(up.data() + 16 * (size_t)n_aln, rank.data(), 2 * (size_t)n_ref);
    if (S->small_at != S->small.p || up != S->small_seen) {
        S->small_at = nullptr;
        if (!up.empty()) PAV_HIP(ctx, hipMemcpyAsync(d_tp, up.data(), up.size(), hipMemcpyHostToDevice, st));
        PAV_HIP(ctx, hipStreamSynchronize(st));                          // `up` is pageable: the copy has read it when this returns
        S->small_seen.swap(up);
        S->small_at = S->small.p;
    }
    const pav_aln *d_aln = ctx->d_aln.as<pav_aln>();
    unsigned long long cnt[8] = {0};
    int rc;

    // ---- keys of both cluster tables, sorted into the rules' iteration order --------------------------------------------
    // only the key bits in use are sorted: the rank field is as wide as needed to keep all-ones (the sentinels) above every
    // real rank - 24 chromosomes: 45 instead of 56 bits, six radix passes instead of seven
    unsigned rank_bits = 1;
    while ((1ull << rank_bits) < (uint64_t)n_ref + 1) ++rank_bits;
    // One radix sort for both tables: the indel keys carry a tag bit above the widest key, so the sorted array holds the SNV keys
    // (their sentinels last) and behind them the indel keys (a separate sort of the 0.6 M indel keys went through rocPRIM's merge
    // sort: twenty small launches, 0.15 ms).
    const unsigned long long tag = 1ull << (CM_SHIFT + rank_bits);
    const uint64_t n_keys = n_snv + n_ind;
    unsigned long long *k_in = nullptr, *k_sorted = nullptr;
    bool planned = false;
    if (n_keys && getenv("PAV_FLAG_HOST") == nullptr) {                  // round 3: planned on the device, one synchronisation
        bool again = false;
        unsigned long long pc[2] = {0, 0};
        if ((rc = flag_planned(ctx, S, P, d_cnt, d_tp, d_te, d_rank, tag, pc, &again)) != PAV_OK) return rc;
        lap("planned");
        planned = !again;
        if (planned) { res->n_snv_pass = pc[0]; res->n_indel_pass = pc[1]; }
        else {
            for (auto &t : S->table) t.clear();
            PAV_HIP(ctx, hipMemsetAsync(d_cnt, 0, 128, st));
        }
    }
    if (!planned) {
    if (n_keys) {
        PAV_HIP(ctx, S->a.reserve(8 * n_keys)); PAV_HIP(ctx, S->b.reserve(8 * n_keys));
        k_in = S->a.as<unsigned long long>(); k_sorted = S->b.as<unsigned long long>();
        // keys of both tables (sentinels for the rows the FILTER drops), then - the common case first - the real keys of both
        // compacted in table order (tile counts -> one scan -> scatter) and a look whether they rise (d_cnt[3])
        const uint32_t tiles_snv = (uint32_t)((n_snv + KEY_TILE - 1) / KEY_TILE), tiles_ind = (uint32_t)((n_ind + KEY_TILE - 1) / KEY_TILE);
        PAV_HIP(ctx, S->tmp.reserve(12 * ((size_t)tiles_snv + tiles_ind) + 64));
        uint64_t *d_toff = S->tmp.as<uint64_t>();                                       // tile offsets (snv | indel), then tile counts
        uint32_t *d_tcnt = reinterpret_cast<uint32_t *>(d_toff + tiles_snv + tiles_ind);
        if (n_snv)
            PAV_LAUNCH(ctx, "k_snv_keys", k_snv_keys, tiles_snv, 256, 0, ctx->d_snv.as<pav_snv>(), n_snv, d_aln, d_rank,
                       d_tp, d_te, k_in, d_cnt, tag - 1, d_tcnt);
        if (n_ind)
            PAV_LAUNCH(ctx, "k_indel_keys", k_indel_keys, tiles_ind, 256, 0, ctx->d_indel.as<pav_indel>(), n_ind, d_aln, d_rank,
                       d_tp, d_te, k_in + n_snv, d_cnt + 1, tag, tag | (tag - 1), d_tcnt + tiles_snv);
        TileScanArgs TS;
        TS.cnt[0] = d_tcnt; TS.cnt[1] = d_tcnt + tiles_snv; TS.off[0] = d_toff; TS.off[1] = d_toff + tiles_snv;
        TS.n_tiles[0] = tiles_snv; TS.n_tiles[1] = tiles_ind;
        PAV_LAUNCH(ctx, "k_key_tile_scan", k_key_tile_scan, 2, 1024, 0, TS);
        CompactKeysArgs CK;
        CK.in[0] = k_in; CK.in[1] = k_in + n_snv; CK.out[0] = k_sorted; CK.out[1] = k_sorted + n_snv;
        CK.off[0] = d_toff; CK.off[1] = d_toff + tiles_snv; CK.n[0] = n_snv; CK.n[1] = n_ind; CK.tiles0 = tiles_snv; CK.low = tag - 1;
        PAV_LAUNCH(ctx, "k_compact_keys", k_compact_keys, tiles_snv + tiles_ind, 256, 0, CK);
        for (int t = 0; t < 2; ++t) {
            const uint64_t n_t = t == 0 ? n_snv : n_ind, at = t == 0 ? 0 : n_snv;
            if (!n_t) continue;
            PAV_LAUNCH(ctx, "k_check_sorted", k_check_sorted, (uint32_t)std::min<uint64_t>((n_t + 255) / 256, 4u * (uint32_t)ctx->n_cu), 256, 0,
                       k_sorted + at, d_cnt + (t == 0 ? 0 : 2), d_cnt + 3);
        }
    }
    PAV_HIP(ctx, hipMemcpyAsync(cnt, d_cnt, 64, hipMemcpyDeviceToHost, st));
    PAV_HIP(ctx, hipStreamSynchronize(st));
    if (cnt[3] && n_keys) {                                         // a table in another order: the general path, one radix sort
        if ((rc = sort_keys(ctx, S, k_in, k_sorted, n_keys, CM_SHIFT + rank_bits + 1)) != PAV_OK) return rc;
    }
    lap("keys+sort");
    const uint64_t snv_pass = cnt[0], ind_pass = cnt[1], ind_small = cnt[2];
    res->n_snv_pass = snv_pass;
    res->n_indel_pass = ind_pass;

    // ---- the four tables: everything that only needs the counts above is queued behind one another, then read back together:
    //      cluster_snv, cluster_indel (rule call_inv_cluster), the INS / DEL split of both vartypes (rule
    //      call_inv_flag_insdel_cluster); the two matches follow once the split sizes are known.  Four readbacks in all.
    // d_cnt words: 4 / 5 sweep hits (snv, indel); 8,9 / 10,11 DEL and INS rows of vartype sv / indel; 12 / 13 match hits
    const uint64_t cap_snv = sweep_cap(snv_pass, P->cluster_min_snv), cap_ind = sweep_cap(ind_small, P->cluster_min_indel);
    if (snv_pass) PAV_HIP(ctx, S->hits_b[0].reserve(sizeof(ClusterHit) * cap_snv));
    if (ind_small) PAV_HIP(ctx, S->hits_b[1].reserve(sizeof(ClusterHit) * cap_ind));
    PAV_HIP(ctx, hipMemsetAsync(d_cnt, 0, 128, st));
    if ((rc = sweep_launch(ctx, S, k_sorted, snv_pass, P->cluster_win, P->cluster_win, P->cluster_min_snv,
                           S->hits_b[0].as<ClusterHit>(), d_cnt + 4)) != PAV_OK) return rc;
    if (ind_small) PAV_LAUNCH(ctx, "k_indel_mid", k_indel_mid, (uint32_t)((ind_small + 255) / 256), 256, 0, k_sorted + n_snv, ind_small, tag);
    if ((rc = sweep_launch(ctx, S, k_sorted + n_snv, ind_small, P->cluster_win, P->cluster_win, P->cluster_min_indel,
                           S->hits_b[1].as<ClusterHit>(), d_cnt + 5)) != PAV_OK) return rc;
    if (ind_pass) {
        for (int t = 0; t < 2; ++t) {
            const int64_t lo64 = t == 0 ? 50 : P->insdel_min_svlen;                   // :497
            const uint32_t lo = (uint32_t)std::min<int64_t>(std::max<int64_t>(lo64, 0), 0xffffffffll);
            const uint32_t hi = t == 0 ? 0xffffffffu : 50u;                           // :504-505 (svlen is below 2^28)
            PAV_HIP(ctx, S->split[t][0].reserve(16 * n_ind)); PAV_HIP(ctx, S->split[t][1].reserve(16 * n_ind));
            PAV_HIP(ctx, S->split[t][2].reserve(8 * n_ind)); PAV_HIP(ctx, S->split[t][3].reserve(8 * n_ind));
            PAV_LAUNCH(ctx, "k_insdel_split", k_insdel_split, (uint32_t)((n_ind + 255) / 256), 256, 0, ctx->d_indel.as<pav_indel>(), n_ind, d_aln,
                       d_rank, d_tp, d_te, lo, hi, S->split[t][0].as<unsigned long long>(), S->split[t][1].as<unsigned long long>(),
                       S->split[t][2].as<unsigned long long>(), S->split[t][3].as<unsigned long long>(), d_cnt + 8 + 2 * t);
        }
    }
    unsigned long long c2[16] = {0};
    PAV_HIP(ctx, hipMemcpyAsync(c2, d_cnt, 128, hipMemcpyDeviceToHost, st));
    lap("queue 2");
    PAV_HIP(ctx, hipStreamSynchronize(st));                                             // readback 2
    lap("sweeps+split");
    if (c2[4] > cap_snv || c2[5] > cap_ind)
        return fail(ctx, PAV_E_LIMIT, "flag: cluster output overflow (%llu / %llu hits)", c2[4], c2[5]);
    std::vector<ClusterHit> sweep_hits[2];
    sweep_hits[0].resize(c2[4]); sweep_hits[1].resize(c2[5]);
    for (int q = 0; q < 2; ++q)
        if (!sweep_hits[q].empty())
            PAV_HIP(ctx, hipMemcpyAsync(sweep_hits[q].data(), S->hits_b[q].p, sizeof(ClusterHit) * sweep_hits[q].size(), hipMemcpyDeviceToHost, st));
    for (int t = 0; t < 2; ++t) {
        const uint64_t n_del = c2[8 + 2 * t], n_ins = c2[9 + 2 * t];
        if (!n_del || !n_ins) continue;
        PAV_HIP(ctx, S->hits_b[2 + t].reserve(sizeof(MatchHit) * n_ins));
        if ((rc = match_launch(ctx, S, S->split[t][0].as<unsigned long long>(), S->split[t][1].as<unsigned long long>(),
                               S->split[t][2].as<unsigned long long>(), S->split[t][3].as<unsigned long long>(), n_del, n_ins,
                               P->insdel_flank_cluster, S->hits_b[2 + t].as<MatchHit>(), d_cnt + 12 + t)) != PAV_OK) return rc;
    }
    unsigned long long c3[2] = {0, 0};
    PAV_HIP(ctx, hipMemcpyAsync(c3, d_cnt + 12, 16, hipMemcpyDeviceToHost, st));
    lap("queue 3");
    PAV_HIP(ctx, hipStreamSynchronize(st));                                             // readback 3 (sweep hits have landed too)
    lap("matches");
    sweep_collect(sweep_hits[0], S->table[3]);
    sweep_collect(sweep_hits[1], S->table[2]);
    std::vector<MatchHit> match_hits[2];
    for (int t = 0; t < 2; ++t) {
        match_hits[t].resize(c3[t]);
        if (c3[t]) PAV_HIP(ctx, hipMemcpyAsync(match_hits[t].data(), S->hits_b[2 + t].p, sizeof(MatchHit) * c3[t], hipMemcpyDeviceToHost, st));
    }
    if (c3[0] || c3[1]) PAV_HIP(ctx, hipStreamSynchronize(st));                         // readback 4
    for (int t = 0; t < 2; ++t) merge_matches(match_hits[t], P->insdel_flank_merge, S->table[t]);
    lap("hits+merge");
    }

    // ---- flagged regions (rule call_inv_merge_flagged_loci) ----------------------------------------------------------------
    for (int t = 0; t < 4; ++t) { res->tables[t] = S->table[t].data(); res->n[t] = S->table[t].size(); }
    merge_loci(res->tables, res->n, P->merge_flank, P->batch_count, P->sig_filter, S->loci);
    res->loci = S->loci.data();
    res->n_loci = S->loci.size();
    if (prof_flush(ctx) != PAV_OK) return PAV_E_HIP;
    lap("loci");
    return PAV_OK;
}

}  // extern "C"
