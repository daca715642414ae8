// side_launcher_test.cpp -- the launch thread of a two-chain rollout (gym-2048_amd/csrc/g2048_side_launcher.h) driven the
// way g2048_api.hip drives it, without HIP: built with g++ -fsanitize=thread by tests/test_side_launcher.py.
//
//   side_launcher_test <cycles> <poster threads> <spin_us>
//
// Every poster thread runs `cycles` rounds of: [sometimes nudge, as a rollout too short to split does, WITHOUT the lock]
// -> lock `use` (SideChain::use) -> post a job that touches plain, unsynchronised state and writes error[] -> do some
// "launches" of its own -> wait -> read result / error / the state -> unlock -> [sometimes pause across the thread's
// sleep / wake window].  The plain state is how ThreadSanitizer sees whether post -> run -> wait really orders the job
// against its poster.  Exit code 0 and "ok ..." on stdout when every job ran exactly once, in order.
#include "../../gym-2048_amd/csrc/g2048_side_launcher.h"

#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

int main(int argc, char **argv)
{
    const long cycles = argc > 1 ? std::atol(argv[1]) : 100000;
    const int posters = argc > 2 ? std::atoi(argv[2]) : 2;
    const long spin_us = argc > 3 ? std::atol(argv[3]) : 20;
    g2048::SideLauncher w;
    w.spin_us = spin_us;
    w.start();
    std::mutex use;
    unsigned long long plain_counter = 0; // written by the jobs, read by the posters: no atomics on purpose
    unsigned long long last_seen_by_job = 0;
    std::atomic<long> failures{0};
    std::vector<std::thread> threads;
    for (int p = 0; p < posters; ++p) {
        threads.emplace_back([&, p] {
            std::mt19937 rng(1234u + static_cast<unsigned>(p));
            for (long c = 0; c < cycles; ++c) {
                if (rng() % 7u == 0u)
                    w.nudge();
                {
                    std::lock_guard<std::mutex> lock(use);
                    const unsigned long long before = plain_counter;
                    const int want = static_cast<int>(rng() % 1000u);
                    const uint64_t ticket = w.post([&plain_counter, &last_seen_by_job, &w, before, want]() -> int {
                        last_seen_by_job = before;
                        plain_counter = before + 1;
                        snprintf(w.error, sizeof w.error, "job %llu", before + 1);
                        return want;
                    });
                    for (volatile int k = 0; k < static_cast<int>(rng() % 64u); ++k) // the caller's own launches
                        ;
                    const int got = w.wait(ticket);
                    char expect[64];
                    snprintf(expect, sizeof expect, "job %llu", before + 1);
                    if (got != want || plain_counter != before + 1 || last_seen_by_job != before || std::strcmp(w.error, expect) != 0)
                        failures.fetch_add(1);
                }
                const unsigned r = rng() % 1000u;
                if (r < 3u) // cross the sleep / wake transition
                    std::this_thread::sleep_for(std::chrono::microseconds(spin_us + static_cast<long>(rng() % 60u)));
            }
        });
    }
    for (auto &t : threads)
        t.join();
    const unsigned long long sleeps = w.sleeps.load();
    w.stop();
    w.stop(); // idempotent
    const unsigned long long want_total = static_cast<unsigned long long>(cycles) * static_cast<unsigned long long>(posters);
    if (failures.load() != 0 || plain_counter != want_total) {
        printf("FAILED: %ld bad rounds, %llu of %llu jobs ran\n", failures.load(), plain_counter, want_total);
        return 1;
    }
    printf("ok: %llu jobs from %d threads, the launcher slept %llu times (spin %ld us)\n", plain_counter, posters, sleeps, spin_us);
    // a launcher that was never started stops cleanly, and the environment knob parses as documented
    g2048::SideLauncher idle;
    idle.stop();
    setenv("G2048_SIDE_SPIN_US", "2000", 1);
    if (g2048::side_spin_us_from_env() != 2000) return 2;
    setenv("G2048_SIDE_SPIN_US", "-5", 1);
    if (g2048::side_spin_us_from_env() != 0) return 2;
    setenv("G2048_SIDE_SPIN_US", "99999999", 1);
    if (g2048::side_spin_us_from_env() != g2048::kMaxSpinUs) return 2;
    setenv("G2048_SIDE_SPIN_US", "x", 1);
    if (g2048::side_spin_us_from_env() != g2048::kDefaultSpinUs) return 2;
    unsetenv("G2048_SIDE_SPIN_US");
    if (g2048::side_spin_us_from_env() != g2048::kDefaultSpinUs || g2048::kDefaultSpinUs > 200) return 2;
    return 0;
}
