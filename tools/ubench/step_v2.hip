// step_v2.hip -- A/B harness (measurement tool, not product): the round-1 step kernel (score array,
// transposing move, per-wave DPP bookkeeping; sources taken from git at build time into _v1/, see
// tools/ubench/build.sh) against the current record-layout kernel, timed interleaved in one process
// (HIP events, median over rounds).  Also cross-checks that both kernels play the same games.
// Usage: step_v2 [log2_boards] [rounds]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "_v1/g2048_kernels.hip"
#include "../../gym-2048_amd/csrc/g2048_kernels.hip"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Variant { std::string name; std::function<void(uint32_t t)> launch; };

int main(int argc, char **argv)
{
    const int lg = argc > 1 ? atoi(argv[1]) : 20;
    const int rounds = argc > 2 ? atoi(argv[2]) : 15;
    const uint32_t n = 1u << lg;
    const int launches = lg >= 23 ? 16 : 64;

    // ---- v1 state
    g2048v1::StepArgs a1{};
    CHECK(hipMalloc(&a1.st.boards, (size_t)n * 16));
    CHECK(hipMalloc(&a1.st.score, (size_t)n * 4));
    CHECK(hipMalloc(&a1.st.last_score, (size_t)n * 4));
    CHECK(hipMalloc(&a1.st.wave_stats, (size_t)(n / 64 + 16) * sizeof(g2048v1::WaveStats)));
    CHECK(hipMemset(a1.st.score, 0, (size_t)n * 4));
    CHECK(hipMemset(a1.st.last_score, 0, (size_t)n * 4));
    CHECK(hipMemset(a1.st.wave_stats, 0, (size_t)(n / 64 + 16) * sizeof(g2048v1::WaveStats)));
    // ---- v2 state
    g2048::StepArgs a2{};
    CHECK(hipMalloc(&a2.st.boards, (size_t)n * 16));
    CHECK(hipMalloc(&a2.st.last_record, (size_t)n * 16));
    CHECK(hipMalloc(&a2.st.ep_counters, (size_t)(n / 64 + 16) * 16));
    CHECK(hipMemset(a2.st.last_record, 0, (size_t)n * 16));
    CHECK(hipMemset(a2.st.ep_counters, 0, (size_t)(n / 64 + 16) * 16));
    uint8_t *actions, *term; float *reward;
    CHECK(hipMalloc(&actions, (size_t)n * launches));
    CHECK(hipMalloc(&term, (size_t)n * launches));
    CHECK(hipMalloc(&reward, (size_t)n * launches * 4));
    CHECK(hipMemset(term, 0, (size_t)n * launches));
    CHECK(hipMemset(reward, 0, (size_t)n * launches * 4));
    a1.n = a2.n = n; a1.seed_lo = a2.seed_lo = 42; a1.auto_reset = a2.auto_reset = 1;
    CHECK(g2048::launch_fill_actions(actions, n, 0, 42, 0, 1, launches, 0));
    a1.t_lo = a2.t_lo = 0;
    CHECK(g2048v1::launch_reset(a1, 0, nullptr, 0));
    CHECK(g2048::launch_reset(a2, 0, nullptr, 0));
    CHECK(hipDeviceSynchronize());

    // ---- cross-check: 48 steps, same boards / scores / last_score
    {
        uint32_t *plain; int32_t *sc2;
        CHECK(hipMalloc(&plain, (size_t)n * 16)); CHECK(hipMalloc(&sc2, (size_t)n * 4));
        std::vector<uint32_t> h1((size_t)n * 4), h2((size_t)n * 4); std::vector<int32_t> s1(n), s2(n), l1(n), l2(n);
        for (int j = 0; j < 48; ++j) {
            a1.t_lo = a2.t_lo = 1 + j;
            a1.actions = a2.actions = actions + (size_t)(j % launches) * n;
            a1.reward = a2.reward = nullptr; a1.terminated = a2.terminated = nullptr;
            CHECK(g2048v1::launch_step(a1, 1, 0));
            CHECK(g2048::launch_step(a2, 1, 0));
        }
        CHECK(g2048::launch_export_boards(a2.st.boards, n, reinterpret_cast<uint4 *>(plain), 0));
        CHECK(g2048::launch_export_scores(a2.st.boards, n, sc2, 0));
        CHECK(hipMemcpy(h1.data(), a1.st.boards, (size_t)n * 16, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(h2.data(), plain, (size_t)n * 16, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(s1.data(), a1.st.score, (size_t)n * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(s2.data(), sc2, (size_t)n * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(l1.data(), a1.st.last_score, (size_t)n * 4, hipMemcpyDeviceToHost));
        CHECK(g2048::launch_export_last_scores(a2.st, n, sc2, 0));
        CHECK(hipMemcpy(l2.data(), sc2, (size_t)n * 4, hipMemcpyDeviceToHost));
        printf("cross-check after 48 steps: boards %s, scores %s, last_score %s\n", h1 == h2 ? "equal" : "DIFFER",
               s1 == s2 ? "equal" : "DIFFER", l1 == l2 ? "equal" : "DIFFER");
        CHECK(hipFree(plain)); CHECK(hipFree(sc2));
    }

    std::vector<Variant> vs;
    auto io1 = [&](uint32_t j) { a1.t_lo = 100 + j; a1.actions = actions + (size_t)j * n; a1.reward = reward + (size_t)j * n; a1.terminated = term + (size_t)j * n; };
    auto io2 = [&](uint32_t j) { a2.t_lo = 100 + j; a2.actions = actions + (size_t)j * n; a2.reward = reward + (size_t)j * n; a2.terminated = term + (size_t)j * n; };
    vs.push_back({"v1  round-1 kernel (score array, owner reset, DPP stats), block 256", [&](uint32_t j) { io1(j); (void)g2048v1::launch_step(a1, 1, 0); }});
    vs.push_back({"v3  product step_kernel<1> (records, per-wave LDS tables, in-lane reset)", [&](uint32_t j) { io2(j); (void)g2048::launch_step(a2, 1, 0); }});
    vs.push_back({"v3  product step_kernel<0> (synthetic actions)", [&](uint32_t j) { io2(j); (void)g2048::launch_step(a2, 0, 0); }});
    vs.push_back({"v3  step_kernel<1>, no outputs", [&](uint32_t j) { io2(j); a2.reward = nullptr; a2.terminated = nullptr; (void)g2048::launch_step(a2, 1, 0); }});

    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<std::vector<float>> us(vs.size());
    for (int r = 0; r < rounds + 1; ++r) {
        for (size_t v = 0; v < vs.size(); ++v) {
            CHECK(hipEventRecord(e0, 0));
            for (int j = 0; j < launches; ++j) vs[v].launch(j);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) us[v].push_back(ms * 1e3f / launches);
        }
    }
    printf("boards 2^%d, %d rounds x %d launches; us per launch (median, min)\n", lg, rounds, launches);
    for (size_t v = 0; v < vs.size(); ++v) {
        std::sort(us[v].begin(), us[v].end());
        const float med = us[v][us[v].size() / 2], mn = us[v][0];
        printf("%-82s %8.2f %8.2f   -> algorithmic %5.0f GB/s\n", vs[v].name.c_str(), med, mn, 38.0 * n / (med * 1e-6) / 1e9);
    }
    return 0;
}
