// chain_floor.hip -- measurement tool: how far is the two-chain rollout from what the memory system allows?
// The product's step kernel and a MEMORY-ONLY kernel of the same traffic (16 B record in + 1 B action in, 16 B record +
// 4 B reward + 1 B terminated out per board, nontemporal like the product's, one xor per dword of arithmetic) are run as
// one chain (one whole-batch launch per step) and as two chains of half-batch launches on two streams from two host
// threads -- the product's two-chain form without its fork / join.  Wall clock per step over `launches` steps.
//   Usage: chain_floor [log2_boards=20] [launches=400]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../../gym-2048_amd/csrc/g2048_kernels.hip"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) mem_only_kernel(uint4 *boards, const uint8_t *actions, float *reward, uint8_t *term, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n)
        return;
    g2048::Board r = g2048::load_board_nt(boards, i);
    const uint32_t a = __builtin_nontemporal_load(actions + i);
    r.r[0] ^= a; r.r[1] ^= a << 8; r.r[2] ^= a << 16; r.r[3] ^= a << 24;
    g2048::store_board_nt(boards, i, r);
    __builtin_nontemporal_store(static_cast<float>(r.r[0] & 0xffu), reward + i);
    __builtin_nontemporal_store(static_cast<uint8_t>(r.r[1] & 1u), term + i);
}

int main(int argc, char **argv)
{
    const int lg = argc > 1 ? atoi(argv[1]) : 20;
    const int launches = argc > 2 ? atoi(argv[2]) : 400;
    const uint32_t n = 1u << lg;
    g2048::StepArgs a{};
    CHECK(hipMalloc(&a.st.boards, (size_t)n * 16));
    CHECK(hipMalloc(&a.st.ep_counters, (size_t)(n / 64 + 16) * 32));
    CHECK(hipMemset(a.st.ep_counters, 0, (size_t)(n / 64 + 16) * 32));
    const int ring = 32; // [ring][n] rollout buffers, reused round-robin (the bench's K x n buffers at K = 400 would be 2.4 GiB at 2^20: fine, but 2^22 ...)
    uint8_t *actions, *term; float *reward;
    CHECK(hipMalloc(&actions, (size_t)n * ring));
    CHECK(hipMalloc(&term, (size_t)n * ring));
    CHECK(hipMalloc(&reward, (size_t)n * ring * 4));
    a.n = n; a.seed_lo = 42; a.auto_reset = 1;
    CHECK(g2048::launch_fill_actions(actions, n, 0, 42, 0, 1, ring, 0));
    CHECK(g2048::launch_reset(a, 0, nullptr, 0));
    a.k_steps = 64; a.t_lo = 1;
    CHECK(g2048::launch_rollout_random(a, 0));
    CHECK(hipDeviceSynchronize());
    hipStream_t ss[2];
    int least = 0, greatest = 0;
    CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    CHECK(hipStreamCreateWithFlags(&ss[0], hipStreamNonBlocking));
    CHECK(hipStreamCreateWithPriority(&ss[1], hipStreamNonBlocking, greatest));
    for (int kind = 0; kind < 2; ++kind)          // 0: the product's step kernel, 1: memory-only
        for (int S = 1; S <= 2; ++S) {
            double best = 1e30;
            for (int rep = 0; rep < 6; ++rep) {
                CHECK(hipDeviceSynchronize());
                const auto t0 = std::chrono::steady_clock::now();
                std::vector<std::thread> th;
                for (int q = 0; q < S; ++q)
                    th.emplace_back([&, q] {
                        const uint32_t cnt = n / S, first = q * cnt;
                        g2048::StepArgs p = a;
                        p.n = cnt; p.board_offset = first; p.st.boards = a.st.boards + first;
                        p.st.ep_counters = a.st.ep_counters + (size_t)(first / 64) * g2048::kSlotWords;
                        for (int j = 0; j < launches; ++j) {
                            const size_t off = (size_t)(j % ring) * n + first;
                            if (kind == 0) {
                                p.t_lo = 100 + j; p.actions = actions + off; p.reward = reward + off; p.terminated = term + off;
                                (void)g2048::launch_step(p, 1, ss[q]);
                            } else {
                                hipLaunchKernelGGL(mem_only_kernel, dim3(cnt / 256), dim3(256), 0, ss[q], a.st.boards + first,
                                                   actions + off, reward + off, term + off, cnt);
                            }
                        }
                        (void)hipStreamSynchronize(ss[q]);
                    });
                for (auto &t : th) t.join();
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / launches;
                if (rep > 0 && us < best) best = us;
            }
            printf("2^%d boards, %s, %d chain%s: %.2f us per step (best of 5, wall over %d steps) = %.2f TB/s on 38 B\n", lg,
                   kind == 0 ? "step kernel" : "memory-only", S, S > 1 ? "s" : " ", best, launches, 38.0 * n / best / 1e6);
        }
    return 0;
}
