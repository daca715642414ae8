// g2048_side_launcher.h -- the second launch thread of a two-chain rollout (g2048_set_chains).  HIP-free on purpose:
// tests/side_launcher/ builds this header alone with g++ -fsanitize=thread and runs its protocol from several threads.
//
// A rollout that runs as TWO CHAINS issues the launches of the upper half of the batch from this thread on the device's
// side stream while the calling thread issues the lower half's on the caller's stream.  One host thread issues a launch
// every ~3.3 us; fed by two threads, both hardware queues always have a kernel waiting, and the head of one half-batch
// kernel (loads in flight, nothing to compute yet) overlaps the tail of the other's (tools/ubench/overlap.hip: 9.4 -> 8.2
// us per step at 2^20 boards).
//
// Protocol (one poster at a time -- SideChain::use serialises the engines of a device):
//   post(fn)  hands the thread a job and returns its ticket;  wait(ticket) spins until the job has RUN (its launches are
//             issued; nothing waits for the device) and returns its status;  error[] holds the job's message;
//   nudge()   wakes a sleeping thread without a job, so that a rollout that follows shortly finds it spinning;
//   stop()    ends the thread (joins it).
// After a job or a nudge the thread SPINS for spin_us, then sleeps on a condition variable (waking it costs ~10 us of
// latency, plus 50-100 us of scheduling when the core was given away).  spin_us defaults to kDefaultSpinUs = 200: an
// environment library must not burn a core behind its caller's back; a caller that runs rollouts back to back opts into a
// longer window (G2048_SIDE_SPIN_US, read when the device's side chain is created; bench.py asks for 2 000).
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace g2048 {

inline void cpu_relax()
{
#if defined(__x86_64__)
    _mm_pause();
#else
    std::this_thread::yield();
#endif
}

constexpr long kDefaultSpinUs = 200, kMaxSpinUs = 100000;

// G2048_SIDE_SPIN_US, clamped to [0, kMaxSpinUs]; kDefaultSpinUs when unset or not a number
inline long side_spin_us_from_env()
{
    const char *v = std::getenv("G2048_SIDE_SPIN_US");
    if (!v || !*v)
        return kDefaultSpinUs;
    char *end = nullptr;
    const long us = std::strtol(v, &end, 10);
    if (end == v)
        return kDefaultSpinUs;
    return us < 0 ? 0 : (us > kMaxSpinUs ? kMaxSpinUs : us);
}

struct SideLauncher {
    std::thread thread;
    std::mutex m;
    std::condition_variable cv;
    std::atomic<uint64_t> posted{0}, finished{0}, nudges{0};
    std::atomic<bool> sleeping{false}, quit{false};
    std::atomic<uint64_t> sleeps{0};  // how often the thread went to sleep (tests, diagnostics)
    std::function<int()> job;         // written by post() before `posted` moves, read by run() after it saw the move
    int result = 0;                   // written by run() before `finished` moves, read by wait() after it saw the move
    char error[512] = "";             // the job's message, same ordering as `result`
    long spin_us = kDefaultSpinUs;    // set before start()

    void start()
    {
        thread = std::thread([this] { run(); });
    }

    void run()
    {
        uint64_t seen = 0, seen_nudges = 0;
        uint32_t spins = 0;
        auto idle_since = std::chrono::steady_clock::now();
        for (;;) {
            if (quit.load())
                return;
            const uint64_t now_posted = posted.load();
            if (now_posted != seen) {
                seen = now_posted;
                result = job();
                finished.store(seen);
                idle_since = std::chrono::steady_clock::now();
                continue;
            }
            cpu_relax();
            if ((++spins & 0x3fu) == 0u && std::chrono::steady_clock::now() - idle_since >= std::chrono::microseconds(spin_us)) {
                std::unique_lock<std::mutex> lock(m);
                sleeping.store(true);
                sleeps.fetch_add(1);
                cv.wait(lock, [&] { return posted.load() != seen || quit.load() || nudges.load() != seen_nudges; });
                sleeping.store(false);
                seen_nudges = nudges.load();
                idle_since = std::chrono::steady_clock::now(); // awake again: spin for another window
            }
        }
    }

    void nudge()
    {
        nudges.fetch_add(1);
        std::lock_guard<std::mutex> lock(m);
        cv.notify_one();
    }

    uint64_t post(std::function<int()> fn)
    {
        job = std::move(fn);
        const uint64_t ticket = posted.fetch_add(1) + 1;
        // a thread that is going to sleep re-checks `posted` under the lock before it waits, and one that already waits is
        // woken here: the job cannot be missed either way
        if (sleeping.load()) {
            std::lock_guard<std::mutex> lock(m);
            cv.notify_one();
        }
        return ticket;
    }

    int wait(uint64_t ticket)
    {
        while (finished.load() < ticket)
            cpu_relax();
        return result;
    }

    void stop()
    {
        if (!thread.joinable())
            return;
        {
            std::lock_guard<std::mutex> lock(m);
            quit.store(true);
        }
        cv.notify_one();
        thread.join();
    }
};

} // namespace g2048
