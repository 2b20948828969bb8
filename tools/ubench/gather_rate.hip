// Scattered-sector roofline of the CIGAR-call path (development tool, run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_rate tools/ubench/gather_rate.hip && /tmp/gather_rate
// walk_snv reads, per SNV row, ONE byte of the reference ASCII plane and ONE byte of the contig ASCII plane; SNVs of the
// bench haplotype lie ~470 bases apart (6.5 M rows over 3.08 Gbp), so every byte sits in a 32 B sector of its own: the kernel
// is bound by how many isolated sectors HBM delivers per second, not by bytes.  This tool measures that rate with nothing
// else in the kernel: N sorted positions with the same mean spacing over two 3.1 GB arenas, one byte fetched from each,
// 16 B stored per row (the SNV record) - the memory traffic of walk_snv without its scan, searches and op decoding.
// Variants: rows in flight per lane (memory-level parallelism), and 2-bit-plane reads (0.25 B / base: the sectors are the
// same number, only 4x closer - it buys nothing, which is why the rows are not read from the packed planes).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int U>
__global__ __launch_bounds__(256) void gather2(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b,
                                               const uint32_t *__restrict__ pa, const uint32_t *__restrict__ pb,
                                               uint4 *__restrict__ out, uint64_t n, int shift) {
    const uint64_t base = ((uint64_t)blockIdx.x * 256) * U + threadIdx.x;
    uint32_t xa[U], xb[U];
    uint8_t va[U], vb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint64_t i = base + (uint64_t)u * 256;
        const uint64_t j = i < n ? i : n - 1;
        xa[u] = pa[j]; xb[u] = pb[j];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { va[u] = a[xa[u] >> shift]; vb[u] = b[xb[u] >> shift]; }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint64_t i = base + (uint64_t)u * 256;
        if (i < n) out[i] = make_uint4(xa[u], xb[u], (uint32_t)i, (uint32_t)va[u] | (uint32_t)vb[u] << 8);
    }
}

int main(int argc, char **argv) {
    const uint64_t arena = argc > 1 ? strtoull(argv[1], 0, 10) : 3080000000ull;
    const uint64_t n = argc > 2 ? strtoull(argv[2], 0, 10) : 6532292ull;
    uint8_t *d_a, *d_b; uint32_t *d_pa, *d_pb; uint4 *d_out;
    CK(hipMalloc(&d_a, arena)); CK(hipMalloc(&d_b, arena));
    CK(hipMemset(d_a, 'A', arena)); CK(hipMemset(d_b, 'C', arena));
    CK(hipMalloc(&d_pa, 4 * n)); CK(hipMalloc(&d_pb, 4 * n)); CK(hipMalloc(&d_out, 16 * n));
    // sorted positions with geometric gaps of the right mean (a Poisson process along the sequence)
    std::vector<uint32_t> pa(n), pb(n);
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
    CK(hipMemcpy(d_pa, pa.data(), 4 * n, hipMemcpyHostToDevice)); CK(hipMemcpy(d_pb, pb.data(), 4 * n, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto launch) {
        for (int w = 0; w < 2; ++w) launch();
        CK(hipDeviceSynchronize());
        float best = 1e9f, tot = 0;
        for (int r = 0; r < 10; ++r) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; tot += ms; }
        CK(hipGetLastError());
        printf("%-34s avg %.4f ms  best %.4f ms  -> %.1f G isolated sectors / s (2 per row)\n", name, tot / 10, best, 2.0 * (double)n / (tot / 10) / 1e6);
    };
    printf("%llu rows, two arenas of %.2f GB, mean spacing %.0f bytes\n", (unsigned long long)n, (double)arena / 1e9, mean);
    run("ASCII planes, 1 row / lane", [&] { hipLaunchKernelGGL(gather2<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_a, d_b, d_pa, d_pb, d_out, n, 0); });
    run("ASCII planes, 2 rows / lane", [&] { hipLaunchKernelGGL(gather2<2>, dim3((unsigned)((n + 511) / 512)), dim3(256), 0, 0, d_a, d_b, d_pa, d_pb, d_out, n, 0); });
    run("ASCII planes, 4 rows / lane", [&] { hipLaunchKernelGGL(gather2<4>, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, d_a, d_b, d_pa, d_pb, d_out, n, 0); });
    run("ASCII planes, 8 rows / lane", [&] { hipLaunchKernelGGL(gather2<8>, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, 0, d_a, d_b, d_pa, d_pb, d_out, n, 0); });
    run("2-bit planes (pos / 4), 4 rows", [&] { hipLaunchKernelGGL(gather2<4>, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, d_a, d_b, d_pa, d_pb, d_out, n, 2); });
    return 0;
}
