// Host-side check of pav_amd/csrc/pool.h (tests/test_pool.py builds and runs it; no GPU): every index of every loop is visited
// exactly once, loops of all sizes follow each other without a pause (helpers that wake late must skip a loop that is over),
// a pool that is destroyed while its helpers sleep or spin comes down (also right after a loop: a helper may still be between
// two looks at the generation), and an exception thrown inside a loop - by the caller's share or by a helper's - arrives at the
// caller with no helper left inside the loop.  tests/test_pool.py runs it plain and under ThreadSanitizer.
#include "../../pav_amd/csrc/pool.h"

#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <stdexcept>

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
    // exceptions: thrown at a few indices of loops of every shape; the pool must stay usable afterwards
    {
        pav::HostPool pool(helpers);
        int caught = 0;
        for (int r = 0; r < 300; ++r) {
            const size_t n = (size_t)((r * 53) % 3000) + 2, chunk = (size_t)(r % 2 ? 1 : 16), bad = (size_t)(r * 7919) % n;
            std::vector<int> scratch(n, 0);                            // dies with this iteration: a helper still inside would show
            try {
                pool.run(n, chunk, [&](size_t i) { scratch[i] = 1; if (i == bad || i == n - 1) throw std::runtime_error("boom"); });
            } catch (const std::runtime_error &) { ++caught; }
            std::atomic<size_t> cnt{0};
            pool.run(n, chunk, [&](size_t) { cnt.fetch_add(1); });
            if (cnt.load() != n) { fprintf(stderr, "loop after an exception: %zu of %zu\n", cnt.load(), n); return 1; }
        }
        if (caught != 300) { fprintf(stderr, "exceptions caught: %d of 300\n", caught); return 1; }
    }
    // short lives: constructed, one loop, destroyed at once
    for (int life = 0; life < 200; ++life) {
        pav::HostPool pool(helpers);
        std::atomic<size_t> cnt{0};
        pool.run(100, 1, [&](size_t) { cnt.fetch_add(1); });
        if (cnt.load() != 100) return 1;
    }
    printf("ok %llu\n", checksum);
    return 0;
}
