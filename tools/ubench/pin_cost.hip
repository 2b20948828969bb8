// What pinned host memory costs on this box: hipHostMalloc / hipHostFree of the sizes the library asks for (the upload ring's 32 MB
// slots, the writers' packed-output buffers of 100 - 300 MB), and a device-to-host copy into pinned against pageable memory.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/pin_cost tools/ubench/pin_cost.hip && /tmp/pin_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    (void)hipFree(nullptr);
    void *d = nullptr; (void)hipMalloc(&d, 512u << 20); (void)hipMemset(d, 1, 512u << 20); (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep)
        for (size_t mb : {1, 32, 128, 300}) {
            void *p = nullptr;
            double t0 = now(); hipError_t e = hipHostMalloc(&p, mb << 20, hipHostMallocDefault); double t1 = now();
            if (e != hipSuccess) { printf("hipHostMalloc %zu MB failed\n", mb); continue; }
            (void)hipMemcpy(p, d, mb << 20, hipMemcpyDeviceToHost); double t2 = now();
            (void)hipMemcpy(p, d, mb << 20, hipMemcpyDeviceToHost); double t3 = now();
            (void)hipHostFree(p); double t4 = now();
            void *q = malloc(mb << 20);
            double t5 = now(); (void)hipMemcpy(q, d, mb << 20, hipMemcpyDeviceToHost); double t6 = now();
            (void)hipMemcpy(q, d, mb << 20, hipMemcpyDeviceToHost); double t7 = now();
            double t8 = now(); (void)hipMemcpy(d, q, mb << 20, hipMemcpyHostToDevice); double t9 = now();
            (void)hipMemcpy(d, q, mb << 20, hipMemcpyHostToDevice); double t10 = now();
            void *p2 = nullptr; (void)hipHostMalloc(&p2, mb << 20, hipHostMallocDefault); memset(p2, 1, mb << 20);
            double t11 = now(); (void)hipMemcpy(d, p2, mb << 20, hipMemcpyHostToDevice); double t12 = now();
            (void)hipHostFree(p2);
            free(q);
            printf("%4zu MB: H2D from malloc'd memory %.2f / %.2f ms (%.1f GB/s), from pinned %.2f ms (%.1f GB/s)\n", mb, (t9 - t8) * 1e3, (t10 - t9) * 1e3,
                   (double)(mb << 20) / 1e9 / (t10 - t9), (t12 - t11) * 1e3, (double)(mb << 20) / 1e9 / (t12 - t11));
            printf("%4zu MB: hipHostMalloc %.2f ms, D2H into it %.2f / %.2f ms (%.1f GB/s), hipHostFree %.2f ms; D2H into malloc'd memory %.2f / %.2f ms (%.1f GB/s)\n", mb,
                   (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (double)(mb << 20) / 1e9 / (t3 - t2), (t4 - t3) * 1e3, (t6 - t5) * 1e3, (t7 - t6) * 1e3,
                   (double)(mb << 20) / 1e9 / (t7 - t6));
        }
    return 0;
}
