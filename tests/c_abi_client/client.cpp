// client.cpp -- TEST: a caller of libg2048_hip.so that knows nothing but include/g2048.h (no Python, no torch).
// It plays the benchmark's random-policy rollout through the C ABI with raw hipMalloc'ed I/O buffers and checks
// every board, reward, flag, score and episodic return against the CPU oracle (oracle/g2048_oracle.h, linked
// here as the checker); then the observation-returning step and the host-resident step, the same way.
// Built by __graft_entry__.build() with hipcc; run by tests/test_gpu_abi_surface.py.
//   usage: client [n_boards] [steps] [seed] [board_offset]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/g2048.h"
#include "../../oracle/g2048_oracle.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("FAIL %s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define G_OK(x) do { int rc_ = (x); if (rc_ != G2048_OK) { std::printf("FAIL %s: %d %s\n", #x, rc_, g2048_last_error()); return 3; } } while (0)

int main(int argc, char **argv)
{
    const uint64_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 5000;
    const uint32_t steps = argc > 2 ? static_cast<uint32_t>(std::atoi(argv[2])) : 40;
    const uint64_t seed = argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 42;
    const uint64_t offset = argc > 4 ? std::strtoull(argv[4], nullptr, 10) : 0;

    g2048_engine *eng = nullptr;
    G_OK(g2048_create(n, 0, seed, offset, &eng));
    G_OK(g2048_set_illegal_move_reward(eng, -1.0f));
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    uint8_t *d_actions, *d_term;
    float *d_reward;
    HIP_OK(hipMalloc(&d_actions, n * steps));
    HIP_OK(hipMalloc(&d_term, n * steps));
    HIP_OK(hipMalloc(&d_reward, n * steps * sizeof(float)));
    G_OK(g2048_reset(eng, 0, 0, nullptr, stream));
    G_OK(g2048_fill_random_actions(eng, 1, steps, d_actions, stream)); // transactions 1 .. steps
    g2048_step_io io{};
    io.actions = d_actions;
    io.action_dtype = G2048_ACT_U8;
    io.reward = d_reward;
    io.terminated = d_term;
    G_OK(g2048_rollout(eng, steps, &io, n, 1, stream));                // k launches, [k][n] buffers, auto-reset

    std::vector<uint8_t> boards(n * 16), term(n * steps), actions(n * steps);
    std::vector<int32_t> scores(n), returns(n);
    std::vector<float> reward(n * steps);
    G_OK(g2048_get_boards(eng, boards.data(), stream));                // host buffers: synchronous
    G_OK(g2048_get_scores(eng, scores.data(), stream));
    G_OK(g2048_get_last_scores(eng, returns.data(), stream));
    HIP_OK(hipMemcpy(term.data(), d_term, n * steps, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(actions.data(), d_actions, n * steps, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(reward.data(), d_reward, n * steps * sizeof(float), hipMemcpyDeviceToHost));
    g2048_stats st{};
    G_OK(g2048_episode_stats(eng, &st, stream));

    // ---- the checker: the CPU oracle on the same seed, offset and actions
    std::vector<uint8_t> o_boards(n * 16), o_term(n), o_ill(n);
    std::vector<int32_t> o_score(n), o_last(n), o_len(n);
    std::vector<uint32_t> o_start(n), o_count(n);
    std::vector<float> o_reward(n);
    g2048o_batch b{};
    b.boards = o_boards.data(); b.score = o_score.data(); b.ep_start = o_start.data();
    b.reward = o_reward.data(); b.terminated = o_term.data(); b.illegal = o_ill.data();
    b.last_score = o_last.data(); b.last_len = o_len.data(); b.ep_count = o_count.data();
    g2048o_reset_batch(&b, n, seed, 0, offset, 0, 0);
    uint64_t episodes = 0, illegal_ends = 0;
    int64_t return_sum = 0; // final score of every finished episode, added when it ends (the oracle writes last_score then)
    for (uint32_t j = 0; j < steps; ++j) {
        b.actions = actions.data() + static_cast<size_t>(j) * n;
        g2048o_step_batch(&b, n, seed, 1 + j, offset, -1.0f, 0, 1, 0);
        for (uint64_t i = 0; i < n; ++i) {
            if (b.actions[i] != g2048o_random_action(seed, 1 + j, static_cast<uint32_t>(offset + i))) {
                std::printf("FAIL action step %u board %llu\n", j, (unsigned long long)i);
                return 1;
            }
            if (reward[static_cast<size_t>(j) * n + i] != o_reward[i] || term[static_cast<size_t>(j) * n + i] != o_term[i]) {
                std::printf("FAIL reward/terminated step %u board %llu\n", j, (unsigned long long)i);
                return 1;
            }
            episodes += o_term[i];
            illegal_ends += o_term[i] && o_ill[i];
            if (o_term[i])
                return_sum += o_last[i];
        }
    }
    if (std::memcmp(boards.data(), o_boards.data(), n * 16) || std::memcmp(scores.data(), o_score.data(), n * 4) ||
        std::memcmp(returns.data(), o_last.data(), n * 4)) {
        std::printf("FAIL final boards / scores / returns differ\n");
        return 1;
    }
    if (st.episodes != episodes || st.illegal_ends != illegal_ends) {
        std::printf("FAIL episode counters %llu/%llu vs %llu/%llu\n", (unsigned long long)st.episodes,
                    (unsigned long long)st.illegal_ends, (unsigned long long)episodes, (unsigned long long)illegal_ends);
        return 1;
    }
    if (st.return_sum != return_sum) {
        std::printf("FAIL return_sum %lld vs the oracle's %lld\n", (long long)st.return_sum, (long long)return_sum);
        return 1;
    }
    // ---- the returns-only summary (what a multi-GPU job ships once per rollout): the same three numbers, and the last_*
    //      members say "not computed" -- last_score_max == -1 -- so that a C caller cannot take the zeros for a mean of 0
    {
        g2048_stats *d_sum;
        HIP_OK(hipMalloc(&d_sum, sizeof(g2048_stats)));
        HIP_OK(hipMemset(d_sum, 0xee, sizeof(g2048_stats)));
        G_OK(g2048_returns_summary_async(eng, d_sum, stream));
        HIP_OK(hipStreamSynchronize(stream));
        g2048_stats ro;
        HIP_OK(hipMemcpy(&ro, d_sum, sizeof ro, hipMemcpyDeviceToHost));
        HIP_OK(hipFree(d_sum));
        if (ro.last_score_max != -1 || ro.last_count != 0 || ro.last_score_sum != 0) {
            std::printf("FAIL returns-only summary: last_score_max %d (want -1 = not computed), last_count %llu, last_score_sum %lld\n",
                        ro.last_score_max, (unsigned long long)ro.last_count, (long long)ro.last_score_sum);
            return 1;
        }
        if (ro.episodes != episodes || ro.illegal_ends != illegal_ends || ro.return_sum != return_sum) {
            std::printf("FAIL returns-only summary differs from the oracle's books\n");
            return 1;
        }
        if (st.last_score_max < 0) { // the full reduction on an engine that keeps terminal records DOES compute them
            std::printf("FAIL g2048_episode_stats left last_score_max = %d on an engine with terminal records\n", st.last_score_max);
            return 1;
        }
    }
    // ---- strict actions through the plain C path: a 4 in the buffer is played as its low two bits (0 = up) and reported by
    //      the NEXT call, once, as G2048_ERR_INVALID; with the switch off again it is silent
    {
        g2048_engine *se = nullptr;
        G_OK(g2048_create(256, 0, seed, 0, &se));
        G_OK(g2048_set_strict_actions(se, 1));
        G_OK(g2048_reset(se, 0, 0, nullptr, stream));
        std::vector<uint8_t> a(256, 1);
        a[200] = 4;
        uint8_t *sd_actions, *sd_term;
        float *sd_reward;
        HIP_OK(hipMalloc(&sd_actions, 256));
        HIP_OK(hipMalloc(&sd_term, 256));
        HIP_OK(hipMalloc(&sd_reward, 256 * sizeof(float)));
        std::vector<int32_t> s_scores(256);
        HIP_OK(hipMemcpy(sd_actions, a.data(), 256, hipMemcpyHostToDevice));
        g2048_step_io s3{};
        s3.actions = sd_actions;
        s3.action_dtype = G2048_ACT_U8;
        s3.reward = sd_reward;
        s3.terminated = sd_term;
        G_OK(g2048_step(se, &s3, 1, stream));
        HIP_OK(hipStreamSynchronize(stream));
        uint64_t clk = 0;
        const int rc = g2048_get_scores(se, s_scores.data(), stream);
        if (rc != G2048_ERR_INVALID || !std::strstr(g2048_last_error(), "board 200")) {
            std::printf("FAIL strict actions: rc %d, message '%s'\n", rc, g2048_last_error());
            return 1;
        }
        G_OK(g2048_get_scores(se, s_scores.data(), stream)); // reported once
        G_OK(g2048_get_clock(se, &clk));
        if (clk != 1) {
            std::printf("FAIL strict actions: the offending step must still have been played (clock %llu)\n", (unsigned long long)clk);
            return 1;
        }
        G_OK(g2048_destroy(se));
        HIP_OK(hipFree(sd_actions));
        HIP_OK(hipFree(sd_term));
        HIP_OK(hipFree(sd_reward));
    }
    // ---- phase 2: the step that returns its observation (g2048_step_io.obs, one launch) and the host-resident step
    //      (g2048_host_io_map / g2048_step_host: no hipMemcpy, no stream synchronisation by the caller), 12 more steps
    uint8_t *d_obs;
    HIP_OK(hipMalloc(&d_obs, n * 256));
    std::vector<uint8_t> obs(n * 256), want(n * 256);
    g2048_host_io hio{};
    G_OK(g2048_host_io_map(eng, &hio));
    for (uint32_t j = 0; j < 12; ++j) {
        const uint64_t t = 1 + steps + j;
        std::vector<uint8_t> a(n);
        for (uint64_t i = 0; i < n; ++i)
            a[i] = g2048o_random_action(seed, t, static_cast<uint32_t>(offset + i));
        b.actions = a.data();
        if (j % 2 == 0) { // device buffers + fused uint8 observation
            HIP_OK(hipMemcpy(d_actions, a.data(), n, hipMemcpyHostToDevice));
            g2048_step_io s2{};
            s2.actions = d_actions;
            s2.action_dtype = G2048_ACT_U8;
            s2.reward = d_reward;
            s2.terminated = d_term;
            s2.obs = d_obs;
            s2.obs_dtype = G2048_OBS_U8;
            G_OK(g2048_step(eng, &s2, 1, stream));
            HIP_OK(hipStreamSynchronize(stream));
            HIP_OK(hipMemcpy(obs.data(), d_obs, n * 256, hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(reward.data(), d_reward, n * sizeof(float), hipMemcpyDeviceToHost));
            g2048o_step_batch(&b, n, seed, t, offset, -1.0f, 0, 1, 0);
            g2048o_onehot_batch(o_boards.data(), n, want.data());
            if (std::memcmp(obs.data(), want.data(), n * 256) || std::memcmp(reward.data(), o_reward.data(), n * sizeof(float))) {
                std::printf("FAIL fused observation / reward at extra step %u\n", j);
                return 1;
            }
        } else {          // host-resident I/O
            for (uint64_t i = 0; i < n; ++i)
                hio.actions[i] = a[i];
            G_OK(g2048_step_host(eng, 1, stream));
            g2048o_step_batch(&b, n, seed, t, offset, -1.0f, 0, 1, 0);
            if (std::memcmp(hio.boards, o_boards.data(), n * 16) || std::memcmp(hio.reward, o_reward.data(), n * sizeof(float)) ||
                std::memcmp(hio.terminated, o_term.data(), n) || std::memcmp(hio.illegal, o_ill.data(), n)) {
                std::printf("FAIL host-resident step at extra step %u\n", j);
                return 1;
            }
        }
    }
    G_OK(g2048_fetch_host(eng, stream));
    if (std::memcmp(hio.boards, o_boards.data(), n * 16) || std::memcmp(hio.scores, o_score.data(), n * 4)) {
        std::printf("FAIL g2048_fetch_host boards / scores\n");
        return 1;
    }
    G_OK(g2048_destroy(eng));
    std::printf("OK %llu boards x %u steps through the C ABI == oracle; %llu episodes; + fused observation and "
                "host-resident steps\n", (unsigned long long)n, steps, (unsigned long long)episodes);
    return 0;
}
