// r4_probe.hip -- round-4 A/B harness (measurement tool, not product): what the return accounting costs the step.
// Interleaved in one process, 2^lg boards, HIP events around `launches` back-to-back launches over a ring of [R][n]
// action / reward / terminated buffers, median over rounds:
//   r3        the round-3 product kernel (commit 1a3ea8e: 2-word episode slots, no gain sum)
//   r4        the current product kernel step_kernel<1,true,true,false> (4-word slots, six-DPP-add gain sum, always-store)
//   r4-nosum  the same step with the 8-dword slot but no gain sum at all (what the sum itself costs)
//   r4-noterm the product step without the sparse 16-byte terminal-record store (what per-board last returns cost)
// (A variant with the gain sum as one ds_add_u32 per lane to a zeroed LDS word measured 10.60 us against 10.36 for the
// DPP form at 2^20 boards and its sums did not cross-check: dropped, profiles/r04_b_r4_probe_2p20.txt.)
//   r4-noterm + touch ...: first-generation wavefronts pull the next generation's records into their XCD's L2
// Usage: r4_probe [log2_boards] [rounds] [launches] [touch distance in boards = 2048 * 256]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "_r3/g2048r3_kernels.hip"
#include "../../gym-2048_amd/csrc/g2048_kernels.hip"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

using namespace g2048;

// MODE 0: no gain sum; 1: DPP (= product); 3: DPP, no terminal-record store
template <int MODE>
__global__ void __launch_bounds__(kBlock)
probe_step(uint4 *boards, const void *actions, unsigned long long *ep_counters, uint32_t board_offset, uint32_t seed_lo,
           uint32_t seed_hi, uint32_t t_lo, uint32_t t_hi, uint32_t n, float *reward, const StepTail tail)
{
    __shared__ WaveTables s_tables[kBlock / 64];
    StepArgs p{};
    p.st.boards = boards;
    p.st.last_record = tail.last_record;
    p.st.ep_counters = ep_counters;
    p.actions = actions;
    p.reward = reward;
    p.terminated = tail.terminated;
    p.n = n;
    p.auto_reset = tail.auto_reset;
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    uint32_t dep = 0;
    Words w_early{};
    if constexpr (MODE == 6) { // odd blocks: the Philox block BEFORE the loads are issued (their requests leave ~0.7 us later)
        if (blockIdx.x & 1u) {
            w_early = philox4x32_10(t_lo, t_hi, board_offset + i, 0u, seed_lo, seed_hi);
            dep = w_early.w[0] & 0u; // a data dependency the compiler cannot see through
            asm volatile("" : "+v"(dep));
        }
    }
    if constexpr (MODE == 7) {
        if (blockIdx.x & 1u)
            __builtin_amdgcn_s_sleep(16); // 16 * 64 cycles ~ 0.43 us
    }
    Board rec = load_board_nt(p.st.boards, i + dep);
    const uint2 tables_piece = load_tables_piece();
    // MODE 4 / 5: a first-generation wavefront touches the records (MODE 5: and the actions) of the block that will run in
    // its wave slots NEXT (same XCD: block + 8 * 256 slots-per-XCD), so that they sit in that XCD's L2 when it starts
    uint32_t touched = 0;
    if constexpr (MODE >= 4) {
        const uint32_t partner = i + tail.max_exp; // (max_exp is unused in this probe: carries the prefetch distance in boards)
        if (partner < n) {
            touched = reinterpret_cast<const uint32_t *>(p.st.boards + partner)[0];
            if constexpr (MODE == 5)
                touched ^= static_cast<const uint8_t *>(p.actions)[partner];
        }
    }
    const EpisodeCounters counters = load_episode_counters(p, i);
    Words w;
    if (MODE == 6 && (blockIdx.x & 1u))
        w = w_early;
    else
        w = philox4x32_10(t_lo, t_hi, board_offset + i, 0u, seed_lo, seed_hi);
    const uint32_t action = load_action<1>(p.actions, i + dep, w.w[3]);
    const LdsTables tb = stage_tables(s_tables, use_after(tables_piece, w.w[0]));
    const StepOut o = play_record(rec, action, w, 0u, tb);
    uint32_t episodes = 0, illegal_ends = 0;
    unsigned long long done;
    if constexpr (MODE >= 3) {
        done = __builtin_amdgcn_ballot_w64(o.terminated);
        episodes = static_cast<uint32_t>(__popcll(done));
        illegal_ends = static_cast<uint32_t>(__popcll(__builtin_amdgcn_ballot_w64(o.terminated && !o.legal)));
    } else {
        done = record_episode_ends(p, i, o.terminated, !o.legal, rec, episodes, illegal_ends);
    }
    uint32_t wave_gain = 0;
    if constexpr (MODE != 0)
        wave_gain = wave_sum_lane63(o.gain);
    const unsigned long long pending = pending_after_step(done, p.auto_reset);
    if (o.terminated && p.auto_reset != 0)
        reset_record(rec, o, w, tb);
    store_board_nt(p.st.boards, i, rec);
    __builtin_nontemporal_store(o.legal ? static_cast<float>(o.gain) : tail.illegal_reward, p.reward + i);
    __builtin_nontemporal_store(static_cast<uint8_t>(o.terminated ? 1 : 0), p.terminated + i);
    flush_episode_counts(counters, episodes, illegal_ends, wave_gain, pending);
    if constexpr (MODE >= 4)
        asm volatile("" ::"v"(touched)); // the touch is consumed here, after everything else
}

struct Variant { std::string name; std::function<void(uint32_t j, hipStream_t s)> launch; };

int main(int argc, char **argv)
{
    const int lg = argc > 1 ? atoi(argv[1]) : 20;
    const int rounds = argc > 2 ? atoi(argv[2]) : 15;
    const int launches = argc > 3 ? atoi(argv[3]) : 200;
    const uint32_t n = 1u << lg, R = lg >= 24 ? 4 : 32;
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    auto state = [&](uint4 *&boards, uint4 *&last, unsigned long long *&ctr) {
        CHECK(hipMalloc(&boards, (size_t)n * 16)); CHECK(hipMalloc(&last, (size_t)n * 16)); CHECK(hipMalloc(&ctr, (size_t)(n / 64 + 16) * 32));
        CHECK(hipMemset(boards, 0, (size_t)n * 16)); CHECK(hipMemset(last, 0, (size_t)n * 16)); CHECK(hipMemset(ctr, 0, (size_t)(n / 64 + 16) * 32));
    };
    const int NV = 8;
    uint4 *boards[NV], *last[NV]; unsigned long long *ctr[NV];
    for (int v = 0; v < NV; ++v) state(boards[v], last[v], ctr[v]);
    uint8_t *actions, *term; float *reward;
    CHECK(hipMalloc(&actions, (size_t)R * n)); CHECK(hipMalloc(&term, (size_t)R * n)); CHECK(hipMalloc(&reward, (size_t)R * n * 4));
    CHECK(launch_fill_actions(actions, n, 0, 42u, 0u, 1000, R, s));
    // initial state: reset + 64 aged steps with the product kernel, copied to every variant
    StepArgs a{};
    a.st.boards = boards[1]; a.st.last_record = last[1]; a.st.ep_counters = ctr[1];
    a.n = n; a.seed_lo = 42u; a.t_lo = 0; a.auto_reset = 1; a.k_steps = 64;
    CHECK(launch_reset(a, 0, nullptr, s));
    a.t_lo = 1;
    CHECK(launch_rollout_random(a, s));
    CHECK(hipStreamSynchronize(s));
    for (int v = 0; v < NV; ++v)
        if (v != 1) {
            CHECK(hipMemcpy(boards[v], boards[1], (size_t)n * 16, hipMemcpyDeviceToDevice));
            if (v != 0) // (the round-3 kernel has its own 2-word slot layout)
                CHECK(hipMemcpy(ctr[v], ctr[1], (size_t)(n / 64 + 16) * 32, hipMemcpyDeviceToDevice));
        }
    const dim3 g(n / 256), b(256);
    auto tail_of = [&](int v, uint32_t j) { return StepTail{term + (size_t)(j % R) * n, last[v], nullptr, nullptr, nullptr, 0.0f, 0u, 1u, nullptr, 0u, nullptr, nullptr, 0ull}; };
    std::vector<Variant> vs;
    vs.push_back({"r3 product (2-word slot, no sum)", [&](uint32_t j, hipStream_t st) {
        const g2048r3::StepTail t{term + (size_t)(j % R) * n, last[0], nullptr, nullptr, nullptr, 0.0f, 0u, 1u, nullptr, 0u, nullptr, nullptr, 0ull};
        hipLaunchKernelGGL((g2048r3::step_kernel<1, true, true, false>), g, b, 0, st, boards[0], (const void *)(actions + (size_t)(j % R) * n), ctr[0], 0u, 42u, 0u, 100u + j, 0u, n, reward + (size_t)(j % R) * n, t); }});
    vs.push_back({"r4 product (DPP sum, 3-dword store)", [&](uint32_t j, hipStream_t st) {
        hipLaunchKernelGGL((step_kernel<1, true, true, false>), g, b, 0, st, boards[1], (const void *)(actions + (size_t)(j % R) * n), ctr[1], 0u, 42u, 0u, 100u + j, 0u, n, reward + (size_t)(j % R) * n, tail_of(1, j)); }});
    vs.push_back({"r4-noterm (no terminal-record store)", [&](uint32_t j, hipStream_t st) {
        hipLaunchKernelGGL((probe_step<3>), g, b, 0, st, boards[2], (const void *)(actions + (size_t)(j % R) * n), ctr[2], 0u, 42u, 0u, 100u + j, 0u, n, reward + (size_t)(j % R) * n, tail_of(2, j)); }});
    auto tail_pf = [&](int v, uint32_t j, uint32_t dist) { StepTail t = tail_of(v, j); t.max_exp = dist; return t; };
    const uint32_t dist = argc > 4 ? (uint32_t)atoi(argv[4]) : 2048u * 256u; // boards between a block and the one that follows it in its wave slots
    vs.push_back({"r4-noterm + touch next block's records", [&](uint32_t j, hipStream_t st) {
        hipLaunchKernelGGL((probe_step<4>), g, b, 0, st, boards[4], (const void *)(actions + (size_t)(j % R) * n), ctr[4], 0u, 42u, 0u, 100u + j, 0u, n, reward + (size_t)(j % R) * n, tail_pf(4, j, dist)); }});
    vs.push_back({"r4-noterm + touch records and actions", [&](uint32_t j, hipStream_t st) {
        hipLaunchKernelGGL((probe_step<5>), g, b, 0, st, boards[5], (const void *)(actions + (size_t)(j % R) * n), ctr[5], 0u, 42u, 0u, 100u + j, 0u, n, reward + (size_t)(j % R) * n, tail_pf(5, j, dist)); }});
    vs.push_back({"r4-noterm, odd blocks Philox before loads", [&](uint32_t j, hipStream_t st) {
        hipLaunchKernelGGL((probe_step<6>), g, b, 0, st, boards[6], (const void *)(actions + (size_t)(j % R) * n), ctr[6], 0u, 42u, 0u, 100u + j, 0u, n, reward + (size_t)(j % R) * n, tail_of(6, j)); }});
    vs.push_back({"r4-noterm, odd blocks sleep 0.4 us first", [&](uint32_t j, hipStream_t st) {
        hipLaunchKernelGGL((probe_step<7>), g, b, 0, st, boards[7], (const void *)(actions + (size_t)(j % R) * n), ctr[7], 0u, 42u, 0u, 100u + j, 0u, n, reward + (size_t)(j % R) * n, tail_of(7, j)); }});
    vs.push_back({"r4-nosum (slot, no gain sum)", [&](uint32_t j, hipStream_t st) {
        hipLaunchKernelGGL((probe_step<0>), g, b, 0, st, boards[3], (const void *)(actions + (size_t)(j % R) * n), ctr[3], 0u, 42u, 0u, 100u + j, 0u, n, reward + (size_t)(j % R) * n, tail_of(3, j)); }});
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<std::vector<double>> us(vs.size());
    uint32_t j = 0;
    for (int r = -2; r < rounds; ++r) {
        for (size_t v = 0; v < vs.size(); ++v) {
            CHECK(hipEventRecord(e0, s));
            for (int l = 0; l < launches; ++l)
                vs[v].launch(j + l, s);
            CHECK(hipEventRecord(e1, s));
            CHECK(hipStreamSynchronize(s));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 0)
                us[v].push_back(ms * 1e3 / launches);
        }
        j += launches; // every variant plays the same transactions
    }
    // cross-check: all variants must have played the same games; the three r4 flavours the same slots
    std::vector<uint4> h0(n), h1(n);
    CHECK(hipMemcpy(h0.data(), boards[1], (size_t)n * 16, hipMemcpyDeviceToHost));
    for (int v = 0; v < NV; ++v) {
        CHECK(hipMemcpy(h1.data(), boards[v], (size_t)n * 16, hipMemcpyDeviceToHost));
        printf("boards of variant %d %s the product's\n", v, memcmp(h0.data(), h1.data(), (size_t)n * 16) == 0 ? "==" : "DIFFER FROM");
    }
    std::vector<unsigned long long> c1((size_t)n / 64 * 4), c2((size_t)n / 64 * 4);
    CHECK(hipMemcpy(c1.data(), ctr[1], c1.size() * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(c2.data(), ctr[2], c2.size() * 8, hipMemcpyDeviceToHost));
    printf("slots of r4-noterm %s the product's\n", c1 == c2 ? "==" : "DIFFER FROM");
    for (size_t v = 0; v < vs.size(); ++v) {
        std::sort(us[v].begin(), us[v].end());
        printf("%-36s 2^%d boards: median %7.3f us  min %7.3f  max %7.3f   (38 B/board: %.0f GB/s)\n", vs[v].name.c_str(), lg,
               us[v][us[v].size() / 2], us[v].front(), us[v].back(), 38.0 * n / us[v][us[v].size() / 2] * 1e-3);
    }
    return 0;
}
