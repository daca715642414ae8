// two_stream.hip -- does splitting the batch over S streams (independent halves, no cross dependency)
// hide launch/head/tail overheads of the per-step kernel?  Measurement tool, not product.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "_v1/g2048_kernels.hip" // the ROUND-1 kernels (git e7a8169, namespace g2048v1; tools/ubench/build.sh): this harness documents round 1
using namespace g2048v1;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char **argv)
{
    const int lg = argc > 1 ? atoi(argv[1]) : 20;
    const int K = argc > 2 ? atoi(argv[2]) : 500;
    const uint32_t n = 1u << lg;
    StepArgs a{};
    CHECK(hipMalloc(&a.st.boards, (size_t)n * 16)); CHECK(hipMalloc(&a.st.score, (size_t)n * 4)); CHECK(hipMalloc(&a.st.last_score, (size_t)n * 4));
    CHECK(hipMalloc(&a.st.wave_stats, ((size_t)n / 64 + 64) * sizeof(WaveStats)));
    CHECK(hipMemset(a.st.wave_stats, 0, ((size_t)n / 64 + 64) * sizeof(WaveStats))); CHECK(hipMemset(a.st.score, 0, (size_t)n * 4));
    uint8_t *actions, *terminated; float *reward;
    CHECK(hipMalloc(&actions, (size_t)n * K)); CHECK(hipMalloc(&reward, (size_t)n * K * 4)); CHECK(hipMalloc(&terminated, (size_t)n * K));
    CHECK(hipMemset(reward, 0, (size_t)n * K * 4)); CHECK(hipMemset(terminated, 0, (size_t)n * K));
    a.n = n; a.seed_lo = 42; a.auto_reset = 1; a.t_lo = 0;
    CHECK(launch_reset(a, 0, nullptr, 0));
    CHECK(launch_fill_actions(actions, n, 0, 42, 0, 1, K, 0));
    CHECK(hipDeviceSynchronize());
    for (int S : {1, 2, 4, 8}) {
        std::vector<hipStream_t> st(S);
        for (auto &s : st) CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        const uint32_t part = n / S;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            for (int j = 0; j < K; ++j) {
                for (int s = 0; s < S; ++s) {
                    StepArgs p = a;
                    const size_t off = (size_t)s * part;
                    p.st.boards += off; p.st.score += off; p.st.last_score += off; p.st.wave_stats += off / 64;
                    p.n = part; p.board_offset = (uint32_t)off; p.t_lo = 1 + j;
                    p.actions = actions + (size_t)j * n + off; p.reward = reward + (size_t)j * n + off; p.terminated = terminated + (size_t)j * n + off;
                    launch_step(p, 1, st[s]);
                }
            }
            CHECK(hipDeviceSynchronize());
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (rep) printf("2^%d boards, %d streams: %.2f us per step  -> %.3e steps/s, algorithmic %.0f GB/s\n", lg, S, us / K, n / (us / K * 1e-6), 38.0 * n / (us / K * 1e-6) / 1e9);
        }
        for (auto &s : st) CHECK(hipStreamDestroy(s));
    }
    return 0;
}
