// overlap.hip -- measurement tool: do step kernels of two HIP streams (each stream = one half of the batch, its
// own chain of dependent launches) overlap on the chip?  Run under `rocprofv3 --kernel-trace` and feed the trace
// to tools/overlap_timeline.py.   Usage: overlap [log2_boards] [streams] [launches]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <thread>
#include <vector>

#include "../../gym-2048_amd/csrc/g2048_kernels.hip"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

int main(int argc, char **argv)
{
    const int lg = argc > 1 ? atoi(argv[1]) : 20;
    const int S = argc > 2 ? atoi(argv[2]) : 2;
    const int launches = argc > 3 ? atoi(argv[3]) : 32;
    const uint32_t n = 1u << lg;
    g2048::StepArgs a{};
    CHECK(hipMalloc(&a.st.boards, (size_t)n * 16));
    CHECK(hipMalloc(&a.st.last_record, (size_t)n * 16));
    CHECK(hipMalloc(&a.st.ep_counters, (size_t)(n / 64 + 16) * 32));
    CHECK(hipMemset(a.st.last_record, 0, (size_t)n * 16));
    CHECK(hipMemset(a.st.ep_counters, 0, (size_t)(n / 64 + 16) * 32));
    uint8_t *actions, *term; float *reward;
    CHECK(hipMalloc(&actions, (size_t)n * launches));
    CHECK(hipMalloc(&term, (size_t)n * launches));
    CHECK(hipMalloc(&reward, (size_t)n * launches * 4));
    CHECK(hipMemset(term, 0, (size_t)n * launches));
    CHECK(hipMemset(reward, 0, (size_t)n * launches * 4));
    a.n = n; a.seed_lo = 42; a.auto_reset = 1;
    CHECK(g2048::launch_fill_actions(actions, n, 0, 42, 0, 1, launches, 0));
    CHECK(g2048::launch_reset(a, 0, nullptr, 0));
    a.k_steps = 64; a.t_lo = 1;
    CHECK(g2048::launch_rollout_random(a, 0));
    CHECK(hipDeviceSynchronize());
    hipStream_t ss[8];
    for (int q = 0; q < S; ++q) CHECK(hipStreamCreateWithFlags(&ss[q], hipStreamNonBlocking));
    g2048::StepArgs part[8];
    for (int q = 0; q < S; ++q) {
        part[q] = a;
        part[q].n = n / S; part[q].board_offset = q * (n / S);
        part[q].st.boards = a.st.boards + (size_t)q * (n / S);
        part[q].st.last_record = a.st.last_record + (size_t)q * (n / S);
        part[q].st.ep_counters = a.st.ep_counters + (size_t)q * (n / S / 64) * g2048::kSlotWords;
    }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, ss[0]));
        for (int j = 0; j < launches; ++j)
            for (int q = 0; q < S; ++q) {
                part[q].t_lo = 100 + j;
                part[q].actions = actions + (size_t)j * n + (size_t)q * (n / S);
                part[q].reward = reward + (size_t)j * n + (size_t)q * (n / S);
                part[q].terminated = term + (size_t)j * n + (size_t)q * (n / S);
                CHECK(g2048::launch_step(part[q], 1, ss[q]));
            }
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e1, ss[0]));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("streams %d, boards 2^%d: %.2f us per step (events, incl. the final sync)\n", S, lg, ms * 1e3f / launches);
    }
    // ---- round 4: ONE HOST THREAD PER CHAIN (a single thread issues a launch every ~3.3 us, so S chains of
    //      quarter-batch kernels were host-bound in round 2); wall clock from the first launch to the last chain's end
    {
        a.st.last_record = nullptr; // (round-4 bench configuration: no terminal records)
        for (int q = 0; q < S; ++q) part[q].st.last_record = nullptr;
        for (int rep = 0; rep < 5; ++rep) {
            CHECK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (int q = 0; q < S; ++q)
                th.emplace_back([&, q] {
                    g2048::StepArgs p = part[q];
                    for (int j = 0; j < launches; ++j) {
                        p.t_lo = 100 + j;
                        p.actions = actions + (size_t)j * n + (size_t)q * (n / S);
                        p.reward = reward + (size_t)j * n + (size_t)q * (n / S);
                        p.terminated = term + (size_t)j * n + (size_t)q * (n / S);
                        (void)g2048::launch_step(p, 1, ss[q]);
                    }
                    (void)hipStreamSynchronize(ss[q]);
                });
            for (auto &t : th) t.join();
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            printf("%d chains, one host thread each, boards 2^%d: %.2f us per step (wall, incl. thread start and final syncs)\n", S, lg, us / launches);
        }
    }
    // ---- the same S chains as ONE hipGraph (fork / join by events during capture): no per-launch host cost
    {
        hipGraph_t graph; hipGraphExec_t exec;
        hipEvent_t fork, join[8];
        CHECK(hipEventCreate(&fork));
        for (int q = 0; q < S; ++q) CHECK(hipEventCreate(&join[q]));
        CHECK(hipStreamBeginCapture(ss[0], hipStreamCaptureModeGlobal));
        CHECK(hipEventRecord(fork, ss[0]));
        for (int q = 1; q < S; ++q) CHECK(hipStreamWaitEvent(ss[q], fork, 0));
        for (int j = 0; j < launches; ++j)
            for (int q = 0; q < S; ++q) {
                part[q].t_lo = 100 + j;
                part[q].actions = actions + (size_t)j * n + (size_t)q * (n / S);
                part[q].reward = reward + (size_t)j * n + (size_t)q * (n / S);
                part[q].terminated = term + (size_t)j * n + (size_t)q * (n / S);
                CHECK(g2048::launch_step(part[q], 1, ss[q]));
            }
        for (int q = 1; q < S; ++q) { CHECK(hipEventRecord(join[q], ss[q])); CHECK(hipStreamWaitEvent(ss[0], join[q], 0)); }
        CHECK(hipStreamEndCapture(ss[0], &graph));
        CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        for (int rep = 0; rep < 6; ++rep) {
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, ss[0]));
            CHECK(hipGraphLaunch(exec, ss[0]));
            CHECK(hipEventRecord(e1, ss[0]));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("hipGraph, %d chains, boards 2^%d: %.2f us per step\n", S, lg, ms * 1e3f / launches);
        }
    }
    return 0;
}
