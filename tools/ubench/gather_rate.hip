// Isolated-sector rate of HBM on MI355X (development tool, run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_rate tools/ubench/gather_rate.hip && /tmp/gather_rate
// walk_snv reads, per SNV row, ONE byte of the reference ASCII plane and ONE byte of the contig ASCII plane; the SNVs of the
// bench haplotype lie ~470 bases apart (6.5 M rows over 3.08 Gbp), so every byte sits in a 32 B sector of its own: the kernel is
// bound by how many isolated sectors the memory system delivers per second, not by bytes.  This tool measures that rate with
// nothing else in the kernel, and - round 3 - looks for the MAXIMUM over the things a kernel can choose, so that the number is a
// roof (round 2's single configuration read 39.7 G sectors/s, and walk_snv itself beat it: both arenas were walked at the same
// offsets from two allocations of equal size, i.e. every pair of loads hit the same channel).  Variants:
//   positions sorted (a Poisson process along the sequence, as SNVs are) / shuffled between waves;
//   one arena / two arenas walked in step / two arenas walked at unrelated offsets (the contig is not the reference: the
//   alignment rows start anywhere);  1, 2, 4, 8 independent rows in flight per lane;  16 B stored per row or nothing stored.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int U, int ARENAS, bool STORE, bool NT = false>
__global__ __launch_bounds__(256) void gather(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b,
                                              const uint32_t *__restrict__ pa, const uint32_t *__restrict__ pb,
                                              uint4 *__restrict__ out, uint64_t n, unsigned long long *sink, uint32_t never) {
    const uint64_t base = ((uint64_t)blockIdx.x * 256) * U + threadIdx.x;
    uint32_t xa[U], xb[U];
    uint8_t va[U], vb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint64_t i = base + (uint64_t)u * 256;
        const uint64_t j = i < n ? i : n - 1;
        xa[u] = pa[j]; xb[u] = ARENAS > 1 ? pb[j] : 0u;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        va[u] = NT ? __builtin_nontemporal_load(a + xa[u]) : a[xa[u]];
        vb[u] = ARENAS > 1 ? (NT ? __builtin_nontemporal_load(b + xb[u]) : b[xb[u]]) : (uint8_t)0;
    }
    uint32_t acc = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint64_t i = base + (uint64_t)u * 256;
        if (STORE) { if (i < n) out[i] = make_uint4(xa[u], xb[u], (uint32_t)i, (uint32_t)va[u] | (uint32_t)vb[u] << 8); }
        else acc += (uint32_t)va[u] + vb[u];
    }
    if (!STORE && acc == never) atomicAdd(sink, 1ull);        // `never` is a run-time value: the loads stay
}

int main(int argc, char **argv) {
    const uint64_t arena = argc > 1 ? strtoull(argv[1], 0, 10) : 3080000000ull;
    const uint64_t n = argc > 2 ? strtoull(argv[2], 0, 10) : 6532292ull;
    uint8_t *d_a, *d_b, *d_gap; uint32_t *d_pa, *d_pb, *d_pc, *d_ps; uint4 *d_out; unsigned long long *d_sink;
    CK(hipMalloc(&d_a, arena));
    CK(hipMalloc(&d_gap, 777 * 1048576 + 12345 * 64));           // so that the two arenas do not sit at like offsets of like allocations
    CK(hipMalloc(&d_b, arena));
    CK(hipMemset(d_a, 'A', arena)); CK(hipMemset(d_b, 'C', arena));
    CK(hipMalloc(&d_pa, 4 * n)); CK(hipMalloc(&d_pb, 4 * n)); CK(hipMalloc(&d_pc, 4 * n)); CK(hipMalloc(&d_ps, 4 * n));
    CK(hipMalloc(&d_out, 16 * n)); CK(hipMalloc(&d_sink, 8));
    // sorted positions with gaps of the right mean (a Poisson-like process along the sequence)
    std::vector<uint32_t> pa(n), pb(n), pc(n), ps(n);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    const double mean = (double)arena / (double)n;
    double xa = 0, xb = 0;
    for (uint64_t i = 0; i < n; ++i) {
        xa += 1.0 + mean * 2.0 * (double)(rnd() >> 11) / 9007199254740992.0 * 0.999;
        xb += 1.0 + mean * 2.0 * (double)(rnd() >> 11) / 9007199254740992.0 * 0.999;
        pa[i] = (uint32_t)(xa < (double)arena - 1 ? xa : (double)arena - 1);
        pb[i] = (uint32_t)(xb < (double)arena - 1 ? xb : (double)arena - 1);
    }
    // "unrelated offsets": the second arena is walked in 2048 pieces taken in a shuffled order (alignment rows start anywhere)
    {
        const uint64_t pieces = 2048, per = n / pieces;
        std::vector<uint32_t> order(pieces);
        for (uint32_t i = 0; i < pieces; ++i) order[i] = i;
        for (uint32_t i = pieces - 1; i > 0; --i) std::swap(order[i], order[rnd() % (i + 1)]);
        for (uint64_t i = 0; i < n; ++i) { const uint64_t p = i / per < pieces ? i / per : pieces - 1; const uint64_t src = (uint64_t)order[p] * per + (i - p * per); pc[i] = pb[src < n ? src : n - 1]; }
        // shuffled between waves: blocks of 64 rows in a random order (every wave still reads 64 ascending positions)
        const uint64_t wv = n / 64;
        std::vector<uint32_t> wo(wv);
        for (uint64_t i = 0; i < wv; ++i) wo[i] = (uint32_t)i;
        for (uint64_t i = wv - 1; i > 0; --i) std::swap(wo[i], wo[rnd() % (i + 1)]);
        for (uint64_t i = 0; i < n; ++i) { const uint64_t w = i / 64; ps[i] = w < wv ? pa[(uint64_t)wo[w] * 64 + i % 64] : pa[i]; }
    }
    CK(hipMemcpy(d_pa, pa.data(), 4 * n, hipMemcpyHostToDevice)); CK(hipMemcpy(d_pb, pb.data(), 4 * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_pc, pc.data(), 4 * n, hipMemcpyHostToDevice)); CK(hipMemcpy(d_ps, ps.data(), 4 * n, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double best_rate = 0;
    auto run = [&](const char *name, int sectors_per_row, auto launch) {
        for (int w = 0; w < 2; ++w) launch();
        CK(hipDeviceSynchronize());
        float best = 1e9f, tot = 0;
        for (int r = 0; r < 10; ++r) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; tot += ms; }
        CK(hipGetLastError());
        const double rate = sectors_per_row * (double)n / (tot / 10) / 1e6;
        if (rate > best_rate) best_rate = rate;
        printf("%-58s avg %.4f ms  best %.4f ms  -> %5.1f G isolated sectors / s\n", name, tot / 10, best, rate);
    };
#define LAUNCH(U, AR, ST, PA, PB) [&] { hipLaunchKernelGGL((gather<U, AR, ST>), dim3((unsigned)((n + 256 * U - 1) / (256 * U))), dim3(256), 0, 0, d_a, d_b, PA, PB, d_out, n, d_sink, 0xFFFFFFFFu); }
#define LAUNCH_NT(U, AR, ST, PA, PB) [&] { hipLaunchKernelGGL((gather<U, AR, ST, true>), dim3((unsigned)((n + 256 * U - 1) / (256 * U))), dim3(256), 0, 0, d_a, d_b, PA, PB, d_out, n, d_sink, 0xFFFFFFFFu); }
    {   // distinct 32 B sectors / 64 B lines among the positions of one arena (what the memory side sees of a "row")
        uint64_t u32 = 0, u64 = 0; uint32_t l32 = ~0u, l64 = ~0u;
        for (uint64_t i = 0; i < n; ++i) { if (pa[i] >> 5 != l32) { ++u32; l32 = pa[i] >> 5; } if (pa[i] >> 6 != l64) { ++u64; l64 = pa[i] >> 6; } }
        printf("%llu rows, arenas of %.2f GB, mean spacing %.0f bytes; per arena %.3f distinct 32 B sectors and %.3f distinct 64 B lines per row\n",
               (unsigned long long)n, (double)arena / 1e9, mean, (double)u32 / n, (double)u64 / n);
    }
    run("two arenas in step (round 2's setup), 16 B stored, U=4", 2, LAUNCH(4, 2, true, d_pa, d_pb));
    run("two arenas, unrelated offsets, 16 B stored, U=1", 2, LAUNCH(1, 2, true, d_pa, d_pc));
    run("two arenas, unrelated offsets, 16 B stored, U=2", 2, LAUNCH(2, 2, true, d_pa, d_pc));
    run("two arenas, unrelated offsets, 16 B stored, U=4", 2, LAUNCH(4, 2, true, d_pa, d_pc));
    run("two arenas, unrelated offsets, 16 B stored, U=8", 2, LAUNCH(8, 2, true, d_pa, d_pc));
    run("two arenas, unrelated offsets, nothing stored, U=4", 2, LAUNCH(4, 2, false, d_pa, d_pc));
    run("two arenas, unrelated offsets, nothing stored, U=8", 2, LAUNCH(8, 2, false, d_pa, d_pc));
    run("one arena, sorted, nothing stored, U=4", 1, LAUNCH(4, 1, false, d_pa, d_pb));
    run("one arena, sorted, nothing stored, U=8", 1, LAUNCH(8, 1, false, d_pa, d_pb));
    run("one arena, waves in shuffled order, nothing stored, U=4", 1, LAUNCH(4, 1, false, d_ps, d_pb));
    run("one arena, waves in shuffled order, nothing stored, U=8", 1, LAUNCH(8, 1, false, d_ps, d_pb));
    run("two arenas, unrelated offsets, 16 B stored, U=4, nt loads", 2, LAUNCH_NT(4, 2, true, d_pa, d_pc));
    run("one arena, sorted, nothing stored, U=4, nt loads", 1, LAUNCH_NT(4, 1, false, d_pa, d_pb));
    printf("maximum over the variants: %.1f G isolated sectors / s\n", best_rate);
    return 0;
}
