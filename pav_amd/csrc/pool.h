// A few helper threads for the host loops that stand between two device stages (lifting a thousand regions through the
// alignment table, formatting their log lines): per-region work with no order between regions, a fraction of a millisecond
// in all - but the device waits for it.  The caller's thread takes part; helpers that wake up too late for a loop skip it (the
// caller never waits for a sleeping thread), and helpers spin for a moment after a loop because the next one follows at once.
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace pav {

class HostPool {
public:
    explicit HostPool(int helpers) {
        for (int t = 0; t < helpers; ++t) threads_.emplace_back([this] { loop(); });
    }
    ~HostPool() {
        { std::lock_guard<std::mutex> lk(m_); stop_.store(true); gen_.fetch_add(1); }
        cv_.notify_all();
        for (auto &t : threads_) t.join();
    }
    HostPool(const HostPool &) = delete;
    HostPool &operator=(const HostPool &) = delete;

    // A loop follows within the helpers' spin time: wake them now, so that they are spinning when it arrives
    void wake() {
        if (threads_.empty()) return;
        { std::lock_guard<std::mutex> lk(m_); gen_.fetch_add(1); }
        cv_.notify_all();
    }

    // f(i) for every i in [0, n), in chunks of `chunk` taken by whoever is free; returns when all calls have returned.
    // An exception thrown by f - on the caller's thread or on a helper - ends the loop early (indices not yet taken are
    // skipped) and is rethrown here, on the caller's thread, once no helper is inside the loop any more.
    template <class F> void run(size_t n, size_t chunk, F &&f) {
        if (threads_.empty() || n <= chunk) { for (size_t i = 0; i < n; ++i) f(i); return; }
        fn_ = [&f](size_t i) { f(i); };
        n_ = n; chunk_ = chunk;
        next_.store(0);
        const uint64_t g = gen_.load() + 1;
        open_.store(g);                                               // everything above is in place before a helper may look
        { std::lock_guard<std::mutex> lk(m_); gen_.store(g); }
        cv_.notify_all();
        work();
        open_.store(0);
        while (active_.load() != 0) cpu_relax();                      // helpers inside work(); one that arrives now sees open_ == 0
        fn_ = nullptr;                                                // the functor refers to the caller's frame
        if (failed_.load()) {
            std::exception_ptr e;
            { std::lock_guard<std::mutex> lk(m_); e = err_; err_ = nullptr; }
            failed_.store(false);
            std::rethrow_exception(e);
        }
    }

private:
    static void cpu_relax() {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
    }
    void work() noexcept {
        try {
            for (;;) {
                const size_t a = next_.fetch_add(chunk_);
                if (a >= n_) break;
                const size_t b = a + chunk_ < n_ ? a + chunk_ : n_;
                for (size_t i = a; i < b; ++i) fn_(i);
            }
        } catch (...) {
            next_.store(n_);                                          // nobody takes another chunk
            std::lock_guard<std::mutex> lk(m_);
            if (!err_) err_ = std::current_exception();               // the first one is reported
            failed_.store(true);
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            // a short spin (the loops of one scan round follow each other within microseconds), then sleep
            const auto t0 = std::chrono::steady_clock::now();
            while (gen_.load() == seen) {
                cpu_relax();
                if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) {
                    std::unique_lock<std::mutex> lk(m_);
                    cv_.wait(lk, [&] { return gen_.load() != seen; });
                }
            }
            seen = gen_.load();
            if (stop_.load()) return;
            active_.fetch_add(1);
            if (open_.load() == seen) work();                         // too late for loop `seen`: its state may be gone already
            active_.fetch_sub(1);
        }
    }

    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_;
    std::atomic<uint64_t> gen_{0};
    std::atomic<size_t> next_{0};
    std::atomic<int> active_{0};
    std::atomic<uint64_t> open_{0};                                   // generation whose loop may be joined, 0: none
    std::atomic<bool> stop_{false};
    std::atomic<bool> failed_{false};
    std::exception_ptr err_;                                          // guarded by m_
    std::function<void(size_t)> fn_;
    size_t n_ = 0, chunk_ = 1;
};

}  // namespace pav
