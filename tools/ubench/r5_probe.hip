// r5_probe.hip -- round-5 A/B harness (measurement tool, not product).
//   (1) the step kernel of this round against round 4's (commit 8baa390, sources in _r4/ with every identifier renamed):
//       same games? how many microseconds per launch?
//   (2) BASELINE configs[1] (65 536 boards): is the LAUNCH SHAPE right there?  One host thread issues a launch every ~3.3 us
//       and the kernel takes ~2.7 us, so the per-step path is host-issue-bound at that size.  Measured here: the product's
//       256-lane blocks (256 workgroups at 2^16: one per CU), 128- and 64-lane blocks (512 / 1 024 workgroups: every SIMD of
//       every CU busy), each as stream launches and as a hipGraph of the same launch train (the host out of the loop).
// HIP events around `launches` back-to-back launches over a ring of [R][n] action / reward / terminated buffers, median
// over rounds; the graph variants replay ONE captured train (the same transactions every round: timing only, their
// boards are not compared).
// Usage: r5_probe [log2_boards] [rounds] [launches] [order of the variants within a round, e.g. 103245]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "_r4/g2048r4_kernels.hip"
#include "../../gym-2048_amd/csrc/g2048_kernels.hip"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

using namespace g2048;

// the body of step_kernel<1, true, true, false> (standard outputs, whole blocks, no terminal records) with the block size as
// a parameter
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK)
probe_step(uint4 *boards, const void *actions, unsigned long long *ep_counters, uint32_t board_offset, uint32_t seed_lo,
           uint32_t seed_hi, uint32_t t_lo, uint32_t t_hi, uint32_t n, float *reward, const StepTail tail)
{
    __shared__ WaveTables s_tables[BLOCK / 64];
    StepArgs p{};
    p.st.boards = boards;
    p.st.last_record = tail.last_record;
    p.st.ep_counters = ep_counters;
    p.actions = actions;
    p.reward = reward;
    p.terminated = tail.terminated;
    p.n = n;
    p.auto_reset = tail.auto_reset;
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    Board rec = load_board_nt(p.st.boards, i);
    const uint2 tables_piece = load_tables_piece();
    const EpisodeCounters counters = load_episode_counters(p, i);
    const Words w = philox4x32_10(t_lo, t_hi, board_offset + i, 0u, seed_lo, seed_hi);
    const uint32_t action = load_action<1>(p.actions, i, w.w[3]);
    const LdsTables tb = stage_tables(s_tables, use_after(tables_piece, w.w[0]));
    const StepOut o = play_record(rec, action, w, 0u, tb);
    uint32_t episodes = 0, illegal_ends = 0;
    const unsigned long long done = record_episode_ends(p, i, o.terminated, !o.legal, rec, episodes, illegal_ends);
    const uint32_t wave_gain = wave_sum_lane63(o.gain);
    const unsigned long long pending = pending_after_step(done, p.auto_reset);
    if (o.terminated && p.auto_reset != 0)
        reset_record(rec, o, w, tb);
    store_board_nt(p.st.boards, i, rec);
    __builtin_nontemporal_store(o.legal ? static_cast<float>(o.gain) : tail.illegal_reward, p.reward + i);
    __builtin_nontemporal_store(static_cast<uint8_t>(o.terminated ? 1 : 0), p.terminated + i);
    flush_episode_counts(counters, episodes, illegal_ends, wave_gain, pending);
}

// the same body with the transaction counter read from DEVICE MEMORY (t = *t_ptr + j): what a cached hipGraph of a launch
// train needs to be replayed for another rollout -- only the one-lane set_t node in front changes between replays
__global__ void set_t_kernel(unsigned long long *t_ptr, unsigned long long value)
{
    if (threadIdx.x == 0)
        *t_ptr = value;
}

template <int BLOCK>
__global__ void __launch_bounds__(BLOCK)
probe_step_tind(uint4 *boards, const void *actions, unsigned long long *ep_counters, uint32_t board_offset, uint32_t seed_lo,
                uint32_t seed_hi, const unsigned long long *t_ptr, uint32_t n, uint32_t j, float *reward, const StepTail tail)
{
    __shared__ WaveTables s_tables[BLOCK / 64];
    StepArgs p{};
    p.st.boards = boards;
    p.st.last_record = tail.last_record;
    p.st.ep_counters = ep_counters;
    p.actions = actions;
    p.reward = reward;
    p.terminated = tail.terminated;
    p.n = n;
    p.auto_reset = tail.auto_reset;
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    Board rec = load_board_nt(p.st.boards, i);
    const uint2 tables_piece = load_tables_piece();
    const EpisodeCounters counters = load_episode_counters(p, i);
    const unsigned long long t = *t_ptr + j; // uniform address: one s_load through the scalar cache
    const Words w = philox4x32_10(static_cast<uint32_t>(t), static_cast<uint32_t>(t >> 32), board_offset + i, 0u, seed_lo, seed_hi);
    const uint32_t action = load_action<1>(p.actions, i, w.w[3]);
    const LdsTables tb = stage_tables(s_tables, use_after(tables_piece, w.w[0]));
    const StepOut o = play_record(rec, action, w, 0u, tb);
    uint32_t episodes = 0, illegal_ends = 0;
    const unsigned long long done = record_episode_ends(p, i, o.terminated, !o.legal, rec, episodes, illegal_ends);
    const uint32_t wave_gain = wave_sum_lane63(o.gain);
    const unsigned long long pending = pending_after_step(done, p.auto_reset);
    if (o.terminated && p.auto_reset != 0)
        reset_record(rec, o, w, tb);
    store_board_nt(p.st.boards, i, rec);
    __builtin_nontemporal_store(o.legal ? static_cast<float>(o.gain) : tail.illegal_reward, p.reward + i);
    __builtin_nontemporal_store(static_cast<uint8_t>(o.terminated ? 1 : 0), p.terminated + i);
    flush_episode_counts(counters, episodes, illegal_ends, wave_gain, pending);
}

struct Variant { std::string name; std::function<void(uint32_t j, hipStream_t s)> launch; bool graph; int state; };

int main(int argc, char **argv)
{
    const int lg = argc > 1 ? atoi(argv[1]) : 20;
    const int rounds = argc > 2 ? atoi(argv[2]) : 15;
    const int launches = argc > 3 ? atoi(argv[3]) : 200;
    const uint32_t n = 1u << lg, R = lg >= 24 ? 4 : 32;
    hipStream_t s;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto state = [&](uint4 *&boards, unsigned long long *&ctr) {
        CHECK(hipMalloc(&boards, (size_t)n * 16)); CHECK(hipMalloc(&ctr, (size_t)(n / 64 + 16) * 32));
        CHECK(hipMemset(boards, 0, (size_t)n * 16)); CHECK(hipMemset(ctr, 0, (size_t)(n / 64 + 16) * 32));
    };
    const int NV = 4; // 0: r4 kernel, 1: product, 2: 64-lane blocks, 3: 128-lane blocks (graph variants replay on copies 1 / 2)
    uint4 *boards[NV + 3]; unsigned long long *ctr[NV + 3];
    for (int v = 0; v < NV + 3; ++v) state(boards[v], ctr[v]);
    unsigned long long *t_dev;
    CHECK(hipMalloc(&t_dev, 256)); CHECK(hipMemset(t_dev, 0, 256));
    uint8_t *actions, *term; float *reward;
    CHECK(hipMalloc(&actions, (size_t)R * n)); CHECK(hipMalloc(&term, (size_t)R * n)); CHECK(hipMalloc(&reward, (size_t)R * n * 4));
    CHECK(hipDeviceSynchronize());
    CHECK(launch_fill_actions(actions, n, 0, 42u, 0u, 1000, R, s));
    // initial state: reset + 64 aged steps with the product kernel, copied to every variant.  (The r4 kernel plays by round
    // 4's spawn rule -- ABI 13 -- so its games differ from this round's by design; it gets its own aged state.)
    StepArgs a{};
    a.st.boards = boards[1]; a.st.ep_counters = ctr[1];
    a.n = n; a.seed_lo = 42u; a.t_lo = 0; a.auto_reset = 1; a.k_steps = 64;
    CHECK(launch_reset(a, 0, nullptr, s));
    a.t_lo = 1;
    CHECK(launch_rollout_random(a, s));
    CHECK(hipStreamSynchronize(s));
    for (int v = 0; v < NV + 3; ++v)
        if (v != 1) {
            CHECK(hipMemcpy(boards[v], boards[1], (size_t)n * 16, hipMemcpyDeviceToDevice));
            CHECK(hipMemcpy(ctr[v], ctr[1], (size_t)(n / 64 + 16) * 32, hipMemcpyDeviceToDevice));
        }
    auto tail_of = [&](uint32_t j) { return StepTail{term + (size_t)(j % R) * n, nullptr, nullptr, nullptr, nullptr, 0.0f, 0u, 1u, nullptr, 0u, nullptr, nullptr, 0ull}; };
    auto act_of = [&](uint32_t j) { return (const void *)(actions + (size_t)(j % R) * n); };
    auto rew_of = [&](uint32_t j) { return reward + (size_t)(j % R) * n; };
    std::vector<Variant> vs;
    vs.push_back({"r4 product kernel (commit 8baa390)", [&](uint32_t j, hipStream_t st) {
        const g2048r4::StepTail t{term + (size_t)(j % R) * n, nullptr, nullptr, nullptr, nullptr, 0.0f, 0u, 1u, nullptr, 0u, nullptr, nullptr, 0ull};
        hipLaunchKernelGGL((g2048r4::step_kernel<1, true, true, false>), dim3(n / 256), dim3(256), 0, st, boards[0], act_of(j), ctr[0], 0u, 42u, 0u, 100u + j, 0u, n, rew_of(j), t); }, false, 0});
    vs.push_back({"r5 product kernel, 256-lane blocks", [&](uint32_t j, hipStream_t st) {
        hipLaunchKernelGGL((step_kernel<1, true, true, false>), dim3(n / 256), dim3(256), 0, st, boards[1], act_of(j), ctr[1], 0u, 42u, 0u, 100u + j, 0u, n, rew_of(j), tail_of(j)); }, false, 1});
    vs.push_back({"r5 body, 64-lane blocks", [&](uint32_t j, hipStream_t st) {
        hipLaunchKernelGGL((probe_step<64>), dim3(n / 64), dim3(64), 0, st, boards[2], act_of(j), ctr[2], 0u, 42u, 0u, 100u + j, 0u, n, rew_of(j), tail_of(j)); }, false, 2});
    vs.push_back({"r5 body, 128-lane blocks", [&](uint32_t j, hipStream_t st) {
        hipLaunchKernelGGL((probe_step<128>), dim3(n / 128), dim3(128), 0, st, boards[3], act_of(j), ctr[3], 0u, 42u, 0u, 100u + j, 0u, n, rew_of(j), tail_of(j)); }, false, 3});
    vs.push_back({"r5 product, 256-lane, hipGraph replay", [&](uint32_t j, hipStream_t st) {
        hipLaunchKernelGGL((step_kernel<1, true, true, false>), dim3(n / 256), dim3(256), 0, st, boards[4], act_of(j), ctr[4], 0u, 42u, 0u, 100u + j, 0u, n, rew_of(j), tail_of(j)); }, true, 4});
    vs.push_back({"r5 body, 64-lane, hipGraph replay", [&](uint32_t j, hipStream_t st) {
        hipLaunchKernelGGL((probe_step<64>), dim3(n / 64), dim3(64), 0, st, boards[5], act_of(j), ctr[5], 0u, 42u, 0u, 100u + j, 0u, n, rew_of(j), tail_of(j)); }, true, 5});
    // variant 6: an EXPLICIT graph [set_t] -> K x probe_step_tind<256>, built once; every replay updates the set_t node's
    // value (hipGraphExecKernelNodeSetParams, one call) and plays the rollout's real transactions: boards must equal the
    // product's
    hipGraph_t tgraph;
    hipGraphExec_t texec;
    hipGraphNode_t set_node;
    unsigned long long t_value = 0;
    void *set_args[2] = {&t_dev, &t_value};
    hipKernelNodeParams set_params{};
    {
        CHECK(hipGraphCreate(&tgraph, 0));
        set_params.func = reinterpret_cast<void *>(set_t_kernel);
        set_params.gridDim = dim3(1); set_params.blockDim = dim3(64); set_params.sharedMemBytes = 0;
        set_params.kernelParams = set_args; set_params.extra = nullptr;
        CHECK(hipGraphAddKernelNode(&set_node, tgraph, nullptr, 0, &set_params));
        hipGraphNode_t prev = set_node;
        const auto b0 = std::chrono::steady_clock::now();
        for (int l = 0; l < launches; ++l) {
            uint4 *bp = boards[6]; const void *ap = act_of((uint32_t)l); unsigned long long *cp = ctr[6];
            uint32_t off = 0u, slo = 42u, shi = 0u, nn = n, jj = (uint32_t)l; const unsigned long long *tp = t_dev;
            float *rp = rew_of((uint32_t)l); StepTail tl = tail_of((uint32_t)l);
            void *args[11] = {&bp, &ap, &cp, &off, &slo, &shi, &tp, &nn, &jj, &rp, &tl};
            hipKernelNodeParams kp{};
            kp.func = reinterpret_cast<void *>(probe_step_tind<256>);
            kp.gridDim = dim3(n / 256); kp.blockDim = dim3(256); kp.sharedMemBytes = 0; kp.kernelParams = args; kp.extra = nullptr;
            hipGraphNode_t node;
            CHECK(hipGraphAddKernelNode(&node, tgraph, &prev, 1, &kp));
            prev = node;
        }
        const auto b1 = std::chrono::steady_clock::now();
        CHECK(hipGraphInstantiate(&texec, tgraph, nullptr, nullptr, 0));
        const auto b2 = std::chrono::steady_clock::now();
        printf("explicit graph of %d step nodes: build %.0f us, instantiate %.0f us\n", launches,
               std::chrono::duration<double, std::micro>(b1 - b0).count(), std::chrono::duration<double, std::micro>(b2 - b1).count());
    }
    double set_params_us = 0; int set_params_calls = 0;
    vs.push_back({"r5 body, t from memory, cached graph", [&](uint32_t j, hipStream_t st) {
        t_value = 100ull + j; // (the ring slots of the graph's nodes are l % R: the trains start at multiples of `launches`, and R divides it or not -- see main)
        const auto c0 = std::chrono::steady_clock::now();
        CHECK(hipGraphExecKernelNodeSetParams(texec, set_node, &set_params));
        set_params_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - c0).count();
        ++set_params_calls;
        CHECK(hipGraphLaunch(texec, st)); }, false, 6});
    // capture the graph variants' trains once
    std::vector<hipGraphExec_t> execs(vs.size(), nullptr);
    for (size_t v = 0; v < vs.size(); ++v) {
        if (!vs[v].graph)
            continue;
        hipGraph_t graph;
        CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int l = 0; l < launches; ++l)
            vs[v].launch((uint32_t)l, s);
        CHECK(hipStreamEndCapture(s, &graph));
        CHECK(hipGraphInstantiate(&execs[v], graph, nullptr, nullptr, 0));
        CHECK(hipGraphDestroy(graph));
    }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<std::vector<double>> us(vs.size()), wall(vs.size());
    uint32_t j = 0;
    std::vector<size_t> order;
    if (argc > 4) {
        for (const char *c = argv[4]; *c; ++c)
            if (*c >= '0' && (size_t)(*c - '0') < vs.size())
                order.push_back((size_t)(*c - '0'));
    } else {
        for (size_t v = 0; v < vs.size(); ++v)
            order.push_back(v);
    }
    for (int r = -2; r < rounds; ++r) {
        for (size_t v : order) {
            CHECK(hipStreamSynchronize(s));
            const auto w0 = std::chrono::steady_clock::now();
            CHECK(hipEventRecord(e0, s));
            if (vs[v].graph) {
                CHECK(hipGraphLaunch(execs[v], s));
            } else if (vs[v].state == 6) {
                vs[v].launch(j, s);
            } else {
                for (int l = 0; l < launches; ++l)
                    vs[v].launch(j + l, s);
            }
            CHECK(hipEventRecord(e1, s));
            CHECK(hipStreamSynchronize(s));
            const double wus = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 0) {
                us[v].push_back(ms * 1e3 / launches);
                wall[v].push_back(wus / launches);
            }
        }
        j += launches; // every stream variant plays the same transactions
    }
    std::vector<uint4> h0(n), h1(n);
    CHECK(hipMemcpy(h0.data(), boards[1], (size_t)n * 16, hipMemcpyDeviceToHost));
    if (set_params_calls)
        printf("hipGraphExecKernelNodeSetParams: %.2f us per call (%d calls)\n", set_params_us / set_params_calls, set_params_calls);
    for (int v : {2, 3, 6}) {
        if (v == 6 && (launches % (int)R) != 0)
            continue; // (the cached graph's ring slots only line up with the stream variants' when R divides the train length)
        CHECK(hipMemcpy(h1.data(), boards[v], (size_t)n * 16, hipMemcpyDeviceToHost));
        printf("boards of variant %d %s the product's\n", v, memcmp(h0.data(), h1.data(), (size_t)n * 16) == 0 ? "==" : "DIFFER FROM");
    }
    std::vector<unsigned long long> c1((size_t)n / 64 * 4), c2((size_t)n / 64 * 4);
    CHECK(hipMemcpy(c1.data(), ctr[1], c1.size() * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(c2.data(), ctr[2], c2.size() * 8, hipMemcpyDeviceToHost));
    printf("slots of the 64-lane variant %s the product's\n", c1 == c2 ? "==" : "DIFFER FROM");
    for (size_t v = 0; v < vs.size(); ++v) {
        if (us[v].empty())
            continue;
        std::sort(us[v].begin(), us[v].end());
        std::sort(wall[v].begin(), wall[v].end());
        printf("%-40s 2^%d boards: events median %7.3f us  min %7.3f  max %7.3f | wall median %7.3f   (38 B/board: %.0f GB/s)\n",
               vs[v].name.c_str(), lg, us[v][us[v].size() / 2], us[v].front(), us[v].back(), wall[v][wall[v].size() / 2],
               38.0 * n / us[v][us[v].size() / 2] * 1e-3);
    }
    return 0;
}
