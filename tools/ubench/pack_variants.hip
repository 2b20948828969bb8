// Micro-benchmark of pack-kernel variants (development tool, run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pack_variants tools/ubench/pack_variants.hip && /tmp/pack_variants
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void pack4(uint32_t x, uint32_t &codes8, uint32_t &bad4) {
    const uint32_t t = ((x >> 1) ^ (x >> 2)) & 0x03030303u;
    uint32_t c = t | (t >> 6);
    codes8 = (c | (c >> 12)) & 0xFFu;
    const uint32_t lo = t & 0x01010101u, hi = (t >> 1) & 0x01010101u, both = lo & hi;
    const uint32_t expect = 0x41414141u + lo * 2u + hi * 6u + both * 11u;
    const uint32_t d = (x & 0xDFDFDFDFu) ^ expect;
    uint32_t nz = ((((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d) & 0x80808080u) >> 7;
    bad4 = (nz | (nz >> 7) | (nz >> 14) | (nz >> 21)) & 0xFu;
}
__device__ __forceinline__ void pack16(const uint4 v, uint32_t &two, uint32_t &m16) {
    uint32_t c0, c1, c2, c3, b0, b1, b2, b3;
    pack4(v.x, c0, b0); pack4(v.y, c1, b1); pack4(v.z, c2, b2); pack4(v.w, c3, b3);
    two = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
    m16 = b0 | (b1 << 4) | (b2 << 8) | (b3 << 12);
}

// V0: shipped kernel
__global__ __launch_bounds__(256) void v0(const uint4 *__restrict__ a, uint32_t *__restrict__ two, uint32_t *__restrict__ mask, uint64_t n16) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        uint32_t t, m; pack16(a[i], t, m);
        two[i] = t;
        const uint32_t o = __shfl_xor(m, 1);
        if ((threadIdx.x & 1) == 0) mask[i >> 1] = m | (o << 16);
    }
}
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 nt_load(const uint4 *p) { u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p)); return make_uint4(v.x, v.y, v.z, v.w); }
// V1: U independent 16 B loads per lane per iteration (stride = block), same stores
template <int U, bool NT>
__global__ __launch_bounds__(256) void v1(const uint4 *__restrict__ a, uint32_t *__restrict__ two, uint32_t *__restrict__ mask, uint64_t n16) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * U;
    for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x * U + threadIdx.x; base < n16; base += stride) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t i = base + (uint64_t)u * 256;
            if (i < n16) v[u] = NT ? nt_load(&a[i]) : a[i]; else v[u] = make_uint4(0x4e4e4e4e, 0x4e4e4e4e, 0x4e4e4e4e, 0x4e4e4e4e);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t i = base + (uint64_t)u * 256;
            uint32_t t, m; pack16(v[u], t, m);
            const uint32_t o = __shfl_xor(m, 1);
            if (i < n16) {
                if (NT) __builtin_nontemporal_store(t, &two[i]); else two[i] = t;
                if ((threadIdx.x & 1) == 0) { if (NT) __builtin_nontemporal_store(m | (o << 16), &mask[i >> 1]); else mask[i >> 1] = m | (o << 16); }
            }
        }
    }
}
// V2: lane owns 64 contiguous bases (4 x 16 B loads), stores 16 B of codes + 8 B of mask
template <bool NT>
__global__ __launch_bounds__(256) void v2(const uint4 *__restrict__ a, uint4 *__restrict__ two, uint2 *__restrict__ mask, uint64_t n64) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n64; i += stride) {
        uint4 v0 = a[4 * i], v1_ = a[4 * i + 1], v2_ = a[4 * i + 2], v3 = a[4 * i + 3];
        uint32_t t0, t1, t2, t3, m0, m1, m2, m3;
        pack16(v0, t0, m0); pack16(v1_, t1, m1); pack16(v2_, t2, m2); pack16(v3, t3, m3);
        two[i] = make_uint4(t0, t1, t2, t3);
        mask[i] = make_uint2(m0 | (m1 << 16), m2 | (m3 << 16));
    }
}
// V3: wave-cooperative transposition through LDS: coalesced 16 B loads (4 per lane), coalesced 16 B code stores
__global__ __launch_bounds__(256) void v3(const uint4 *__restrict__ a, uint32_t *__restrict__ two, uint32_t *__restrict__ mask, uint64_t n16) {
    // each block handles 1024 x 16 B = 16 KiB per iteration: lane t loads rows t, t+256, t+512, t+768
    const uint64_t per_iter = 1024;
    const uint64_t stride = (uint64_t)gridDim.x * per_iter;
    for (uint64_t base = (uint64_t)blockIdx.x * per_iter; base < n16; base += stride) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const uint64_t i = base + threadIdx.x + 256 * u; v[u] = i < n16 ? a[i] : make_uint4(0x4e4e4e4e, 0x4e4e4e4e, 0x4e4e4e4e, 0x4e4e4e4e); }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint64_t i = base + threadIdx.x + 256 * u;
            uint32_t t, m; pack16(v[u], t, m);
            const uint32_t o = __shfl_xor(m, 1);
            if (i < n16) { two[i] = t; if ((threadIdx.x & 1) == 0) mask[i >> 1] = m | (o << 16); }
        }
    }
}

int main(int argc, char **argv) {
    const uint64_t bases = argc > 1 ? strtoull(argv[1], 0, 10) : 3080000000ull / 256 * 256;
    uint8_t *d_a; uint32_t *d_two, *d_mask;
    CK(hipMalloc(&d_a, bases)); CK(hipMalloc(&d_two, bases / 4)); CK(hipMalloc(&d_mask, bases / 8));
    std::vector<uint8_t> h(1 << 24);
    const char *al = "ACGTacgtNnACGTACGTACGTACGTACGTAC";
    for (size_t i = 0; i < h.size(); ++i) h[i] = al[(i * 2654435761u >> 13) & 31];
    for (uint64_t o = 0; o < bases; o += h.size()) CK(hipMemcpy(d_a + o, h.data(), std::min<uint64_t>(h.size(), bases - o), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint64_t n16 = bases / 16;
    const double gb = bases * 1.375 / 1e9;
    auto run = [&](const char *name, auto launch) {
        for (int w = 0; w < 2; ++w) launch();
        CK(hipDeviceSynchronize());
        float best = 1e9, tot = 0;
        for (int r = 0; r < 10; ++r) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms); tot += ms; }
        printf("%-28s avg %.4f ms  best %.4f ms  -> %.0f GB/s (avg)  %.0f GB/s (best)\n", name, tot / 10, best, gb / (tot / 10) * 1e3, gb / best * 1e3);
        CK(hipGetLastError());
    };
    run("v0 grid exact", [&] { hipLaunchKernelGGL(v0, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, (const uint4 *)d_a, d_two, d_mask, n16); });
    run("v1 U2 exact", [&] { hipLaunchKernelGGL((v1<2, false>), dim3((unsigned)((n16 + 511) / 512)), dim3(256), 0, 0, (const uint4 *)d_a, d_two, d_mask, n16); });
    run("v1 U4 exact", [&] { hipLaunchKernelGGL((v1<4, false>), dim3((unsigned)((n16 + 1023) / 1024)), dim3(256), 0, 0, (const uint4 *)d_a, d_two, d_mask, n16); });
    run("v1 U4 NT exact", [&] { hipLaunchKernelGGL((v1<4, true>), dim3((unsigned)((n16 + 1023) / 1024)), dim3(256), 0, 0, (const uint4 *)d_a, d_two, d_mask, n16); });
    run("v1 U8 exact", [&] { hipLaunchKernelGGL((v1<8, false>), dim3((unsigned)((n16 + 2047) / 2048)), dim3(256), 0, 0, (const uint4 *)d_a, d_two, d_mask, n16); });
    run("v1 U8 NT exact", [&] { hipLaunchKernelGGL((v1<8, true>), dim3((unsigned)((n16 + 2047) / 2048)), dim3(256), 0, 0, (const uint4 *)d_a, d_two, d_mask, n16); });
    run("v1 U16 exact", [&] { hipLaunchKernelGGL((v1<16, false>), dim3((unsigned)((n16 + 4095) / 4096)), dim3(256), 0, 0, (const uint4 *)d_a, d_two, d_mask, n16); });
    run("v2 exact", [&] { hipLaunchKernelGGL((v2<false>), dim3((unsigned)((n16 / 4 + 255) / 256)), dim3(256), 0, 0, (const uint4 *)d_a, (uint4 *)d_two, (uint2 *)d_mask, n16 / 4); });
    run("v3 exact", [&] { hipLaunchKernelGGL(v3, dim3((unsigned)((n16 + 1023) / 1024)), dim3(256), 0, 0, (const uint4 *)d_a, d_two, d_mask, n16); });
    run("v1 U4 exact (again)", [&] { hipLaunchKernelGGL((v1<4, false>), dim3((unsigned)((n16 + 1023) / 1024)), dim3(256), 0, 0, (const uint4 *)d_a, d_two, d_mask, n16); });
    // reference point: plain device-to-device copy of the same read volume
    run("hipMemcpy D2D (ascii->ascii) *", [&] { CK(hipMemcpyAsync(d_a + bases / 2 / 256 * 256, d_a, bases / 2 / 256 * 256, hipMemcpyDeviceToDevice, 0)); });
    return 0;
}
