// Host-side check of pav_amd/csrc/pool.h (tests/test_pool.py builds and runs it; no GPU): every index of every loop is visited
// exactly once, loops of all sizes follow each other without a pause (helpers that wake late must skip a loop that is over),
// and a pool that is destroyed while its helpers sleep or spin comes down.
#include "../../pav_amd/csrc/pool.h"

#include <cstdio>
#include <cstdlib>
#include <numeric>

int main(int argc, char **argv) {
    const int helpers = argc > 1 ? atoi(argv[1]) : 3;
    const int rounds = argc > 2 ? atoi(argv[2]) : 2000;
    unsigned long long checksum = 0;
    for (int life = 0; life < 3; ++life) {
        pav::HostPool pool(helpers);
        std::vector<std::atomic<int>> hit(5000);
        for (int r = 0; r < rounds; ++r) {
            const size_t n = (size_t)((r * 37) % 4999) + 1, chunk = (size_t)(r % 3 == 0 ? 1 : (r % 3 == 1 ? 32 : 700));
            for (size_t i = 0; i < n; ++i) hit[i].store(0);
            if (r % 50 == 0) pool.wake();
            std::atomic<unsigned long long> sum{0};
            pool.run(n, chunk, [&](size_t i) { hit[i].fetch_add(1); sum.fetch_add(i + 1); });
            for (size_t i = 0; i < n; ++i) if (hit[i].load() != 1) { fprintf(stderr, "round %d: index %zu visited %d times\n", r, i, hit[i].load()); return 1; }
            if (sum.load() != (unsigned long long)n * (n + 1) / 2) { fprintf(stderr, "round %d: sum\n", r); return 1; }
            checksum += sum.load();
            if (r % 400 == 399) std::this_thread::sleep_for(std::chrono::milliseconds(2));     // let the helpers fall asleep
        }
    }
    printf("ok %llu\n", checksum);
    return 0;
}
