// step_variants.hip -- A/B harness for the step kernel structure (measurement tool, not product).
// Builds several variants of the per-step kernel around the SAME device header and times them
// interleaved in one process (HIP events, median over rounds).  Usage: step_variants [log2_boards] [rounds]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "_v1/g2048_kernels.hip" // the ROUND-1 kernels (git e7a8169, namespace g2048v1; tools/ubench/build.sh): this harness documents round 1 // the product kernels themselves (variant "prod")

using namespace g2048v1;
static StepArgs g_prod;   // product-kernel arguments, filled in main()
static int g_prod_outputs = 1;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Args {
    uint4 *boards; int32_t *score; const uint8_t *actions; float *reward; uint8_t *terminated; int32_t *last_score;
    uint32_t n, seed_lo, seed_hi, t_lo;
};

template <typename T> __device__ __forceinline__ T *at(T *base, uint32_t idx)
{
    return reinterpret_cast<T *>(reinterpret_cast<char *>(const_cast<std::remove_const_t<T> *>(base)) + (uint64_t)(idx * (uint32_t)sizeof(T)));
}

enum { F_SCORE = 1, F_RECORD = 2, F_COMPUTE = 4, F_STORE = 8, F_PIPE = 16, F_IDX64 = 32, F_PRIO8 = 64, F_PRIO11 = 128, F_PRIOW = 256, F_TWO = 512, F_TWO_LATE = 1024 };

template <int FLAGS>
__device__ __forceinline__ void do_board(const Args &p, uint32_t i, uint4 v, int32_t score_in, uint32_t action)
{
    Board bd{{v.x, v.y, v.z, v.w}};
    int32_t score = score_in;
    StepResult r;
    r.reward = 0; r.terminated = false; r.illegal = false; r.terminal = bd; r.terminal_score = 0;
    if (FLAGS & F_COMPUTE) {
        const Words w = philox4x32_10(p.t_lo, 0u, i, 0u, p.seed_lo, p.seed_hi);
        r = step_env(bd, score, action & 3u, w, 0.0f, 0u, true);
    } else {
        bd.r[0] ^= action; // keep the data flowing
    }
    if (FLAGS & F_STORE) {
        if (FLAGS & F_IDX64) {
            p.boards[i] = make_uint4(bd.r[0], bd.r[1], bd.r[2], bd.r[3]);
            if (FLAGS & F_SCORE) p.score[i] = score;
            p.reward[i] = r.reward;
            p.terminated[i] = r.terminated;
        } else {
            *at(p.boards, i) = make_uint4(bd.r[0], bd.r[1], bd.r[2], bd.r[3]);
            if (FLAGS & F_SCORE) *at(p.score, i) = score;
            *at(p.reward, i) = r.reward;
            *at(p.terminated, i) = r.terminated;
        }
    } else {
        if (bd.r[0] == 0x12345678u && score == 77) *at(p.boards, i) = make_uint4(bd.r[0], bd.r[1], bd.r[2], bd.r[3]);
    }
    if (FLAGS & F_RECORD) {
        const unsigned long long done = __ballot(r.terminated);
        if (done) {
            if (r.terminated) *at(p.last_score, i) = r.terminal_score;
        }
    }
}

template <int FLAGS, int BS>
__global__ void __launch_bounds__(BS) kern_bs(const Args p)
{
    const uint32_t i = blockIdx.x * BS + threadIdx.x;
    if (i >= p.n) return;
    const uint4 v = p.boards[i]; const int32_t sc = p.score[i]; const uint32_t a = p.actions[i];
    do_board<FLAGS>(p, i, v, sc, a);
}
// cache-policy variants: NT loads and/or NT stores (nt=1 hint)
template <int NTL, int NTS, int SCORE = 1>
__global__ void __launch_bounds__(256) kern_nt(const Args p)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.n) return;
    uint4 v; int32_t sc; uint32_t a;
    if (NTL) {
        const uint32_t *bp = reinterpret_cast<const uint32_t *>(p.boards + i);
        v.x = __builtin_nontemporal_load(bp); v.y = __builtin_nontemporal_load(bp + 1);
        v.z = __builtin_nontemporal_load(bp + 2); v.w = __builtin_nontemporal_load(bp + 3);
        sc = SCORE ? __builtin_nontemporal_load(p.score + i) : 0; a = __builtin_nontemporal_load(p.actions + i);
    } else { v = p.boards[i]; sc = p.score[i]; a = p.actions[i]; }
    Board bd{{v.x, v.y, v.z, v.w}};
    int32_t score = sc;
    const Words w = philox4x32_10(p.t_lo, 0u, i, 0u, p.seed_lo, p.seed_hi);
    const StepResult r = step_env(bd, score, a & 3u, w, 0.0f, 0u, true);
    if (NTS) {
        uint32_t *bp = reinterpret_cast<uint32_t *>(p.boards + i);
        __builtin_nontemporal_store(bd.r[0], bp); __builtin_nontemporal_store(bd.r[1], bp + 1);
        __builtin_nontemporal_store(bd.r[2], bp + 2); __builtin_nontemporal_store(bd.r[3], bp + 3);
        if (SCORE) __builtin_nontemporal_store(score, p.score + i);
        __builtin_nontemporal_store(r.reward, p.reward + i);
        __builtin_nontemporal_store((uint8_t)r.terminated, p.terminated + i);
    } else {
        p.boards[i] = make_uint4(bd.r[0], bd.r[1], bd.r[2], bd.r[3]); p.score[i] = score; p.reward[i] = r.reward; p.terminated[i] = r.terminated;
    }
    if (__ballot(r.terminated) && r.terminated) p.last_score[i] = r.terminal_score;
}
// phase-shifted start: blocks of one "kind" sleep before issuing their loads, so that on every SIMD half
// of the resident waves compute while the other half loads/stores
template <int SHIFT, int DELAY>
__global__ void __launch_bounds__(256) kern_phase(const Args p)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.n) return;
    if ((blockIdx.x >> SHIFT) & 1u) {
#pragma unroll 1
        for (int d = 0; d < DELAY; ++d) __builtin_amdgcn_s_sleep(8); // ~512 cycles per iteration
    }
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p.boards) + i);
    int32_t score = __builtin_nontemporal_load(p.score + i);
    const uint32_t a = __builtin_nontemporal_load(p.actions + i);
    Board bd{{v.x, v.y, v.z, v.w}};
    const Words w = philox4x32_10(p.t_lo, 0u, i, 0u, p.seed_lo, p.seed_hi);
    const StepResult r = step_env(bd, score, a & 3u, w, 0.0f, 0u, true);
    const u32x4 o = {bd.r[0], bd.r[1], bd.r[2], bd.r[3]};
    __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(p.boards) + i);
    __builtin_nontemporal_store(score, p.score + i);
    __builtin_nontemporal_store(r.reward, p.reward + i);
    __builtin_nontemporal_store((uint8_t)r.terminated, p.terminated + i);
    if (__ballot(r.terminated) && r.terminated) p.last_score[i] = r.terminal_score;
}
template <int SHIFT, int DELAY> void launch_phase(const Args &a, uint32_t blocks) { hipLaunchKernelGGL((kern_phase<SHIFT, DELAY>), dim3(blocks), dim3(256), 0, 0, a); }

// two boards per lane (i and i + n/2), all nt, both loads up front
template <int NB>
__global__ void __launch_bounds__(256) kern_nt_multi(const Args p)
{
    const uint32_t i0 = blockIdx.x * 256 + threadIdx.x;
    const uint32_t part = p.n / NB;
    if (i0 >= part) return;
    u32x4 v[NB]; int32_t sc[NB]; uint32_t a[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const uint32_t i = i0 + k * part;
        v[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p.boards) + i);
        sc[k] = __builtin_nontemporal_load(p.score + i); a[k] = __builtin_nontemporal_load(p.actions + i);
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const uint32_t i = i0 + k * part;
        Board bd{{v[k].x, v[k].y, v[k].z, v[k].w}};
        int32_t score = sc[k];
        const Words w = philox4x32_10(p.t_lo, 0u, i, 0u, p.seed_lo, p.seed_hi);
        const StepResult r = step_env(bd, score, a[k] & 3u, w, 0.0f, 0u, true);
        const u32x4 o = {bd.r[0], bd.r[1], bd.r[2], bd.r[3]};
        __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(p.boards) + i);
        __builtin_nontemporal_store(score, p.score + i);
        __builtin_nontemporal_store(r.reward, p.reward + i);
        __builtin_nontemporal_store((uint8_t)r.terminated, p.terminated + i);
        if (__ballot(r.terminated) && r.terminated) p.last_score[i] = r.terminal_score;
    }
}
template <int NB> void launch_nt_multi(const Args &a, uint32_t blocks) { hipLaunchKernelGGL((kern_nt_multi<NB>), dim3((blocks + NB - 1) / NB), dim3(256), 0, 0, a); }

template <int NTL, int NTS, int SCORE = 1> void launch_nt(const Args &a, uint32_t blocks) { hipLaunchKernelGGL((kern_nt<NTL, NTS, SCORE>), dim3(blocks), dim3(256), 0, 0, a); }

template <int FLAGS, int BS> void launch_bs(const Args &a, uint32_t) { hipLaunchKernelGGL((kern_bs<FLAGS, BS>), dim3((a.n + BS - 1) / BS), dim3(BS), 0, 0, a); }

template <int FLAGS>
__global__ void __launch_bounds__(256) kern(const Args p)
{
    const uint32_t stride = gridDim.x * 256;
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.n) return;
    if (FLAGS & (F_PRIO8 | F_PRIO11 | F_PRIOW)) {
        const uint32_t h = (FLAGS & F_PRIO8) ? (blockIdx.x >> 8) : (FLAGS & F_PRIO11) ? (blockIdx.x >> 11) : blockIdx.x;
        switch (h & 3u) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
        }
    }
    if (FLAGS & (F_TWO | F_TWO_LATE)) {
        // straight-line two boards per lane: board i and board i + n/2 (grid covers n/2 lanes)
        const uint32_t i2 = i + p.n / 2;
        if (i >= p.n / 2) return;
        uint4 va, vb; int32_t sa, sb; uint32_t aa, ab;
        if (FLAGS & F_IDX64) {
            va = p.boards[i]; sa = p.score[i]; aa = p.actions[i];
            vb = p.boards[i2]; sb = p.score[i2]; ab = p.actions[i2];
            do_board<FLAGS>(p, i, va, sa, aa);
            do_board<FLAGS>(p, i2, vb, sb, ab);
            return;
        }
        va = *at(p.boards, i); sa = *at(p.score, i); aa = *at(p.actions, i);
        if (FLAGS & F_TWO) { vb = *at(p.boards, i2); sb = *at(p.score, i2); ab = *at(p.actions, i2); }
        if (FLAGS & F_TWO_LATE) {
            // request B only once A has arrived (keeps the first generation's loads short)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            vb = *at(p.boards, i2); sb = *at(p.score, i2); ab = *at(p.actions, i2);
        }
        do_board<FLAGS>(p, i, va, sa, aa);
        do_board<FLAGS>(p, i2, vb, sb, ab);
        return;
    }
    if (!(FLAGS & F_PIPE)) {
        for (; i < p.n; i += stride) {
            uint4 v; int32_t sc = 0; uint32_t a;
            if (FLAGS & F_IDX64) { v = p.boards[i]; if (FLAGS & F_SCORE) sc = p.score[i]; a = p.actions[i]; }
            else { v = *at(p.boards, i); if (FLAGS & F_SCORE) sc = *at(p.score, i); a = *at(p.actions, i); }
            do_board<FLAGS>(p, i, v, sc, a);
        }
    } else {
        uint4 v = *at(p.boards, i); int32_t sc = (FLAGS & F_SCORE) ? *at(p.score, i) : 0; uint32_t a = *at(p.actions, i);
        for (;;) {
            const uint32_t in = i + stride;
            const bool has_next = in < p.n;
            uint4 vn = v; int32_t scn = sc; uint32_t an = a;
            if (has_next) { vn = *at(p.boards, in); if (FLAGS & F_SCORE) scn = *at(p.score, in); an = *at(p.actions, in); }
            do_board<FLAGS>(p, i, v, sc, a);
            if (!has_next) break;
            v = vn; sc = scn; a = an; i = in;
        }
    }
}

__global__ void init_boards(uint4 *boards, int32_t *score, uint8_t *actions, uint32_t n, uint32_t seed)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const Words w = philox4x32_10(0u, 0u, i, 0u, seed, 0u);
    const Board b = fresh_board(w.w[0], w.w[1]);
    boards[i] = make_uint4(b.r[0], b.r[1], b.r[2], b.r[3]);
    score[i] = 0;
    actions[i] = w.w[3] >> 30;
}

struct Variant { const char *name; void (*launch)(const Args &, uint32_t blocks); uint32_t blocks; };
static uint32_t g_lds = 0;

template <int FLAGS> void launch(const Args &a, uint32_t blocks) { hipLaunchKernelGGL(kern<FLAGS>, dim3(blocks), dim3(256), g_lds, 0, a); }
template <int FLAGS, int LDS> void launch_lds(const Args &a, uint32_t blocks) { hipLaunchKernelGGL(kern<FLAGS>, dim3(blocks), dim3(256), LDS, 0, a); }

int main(int argc, char **argv)
{
    const int lg = argc > 1 ? atoi(argv[1]) : 20;
    const int rounds = argc > 2 ? atoi(argv[2]) : 30;
    const uint32_t n = 1u << lg;
    const int K = argc > 3 ? atoi(argv[3]) : 64; // launches per timing sample, each with its own action/reward/terminated slice
    Args a{};
    a.n = n; a.seed_lo = 42; a.seed_hi = 0; a.t_lo = 1;
    uint8_t *actions; float *reward; uint8_t *terminated;
    CHECK(hipMalloc(&a.boards, (size_t)n * 16)); CHECK(hipMalloc(&a.score, (size_t)n * 4)); CHECK(hipMalloc(&a.last_score, (size_t)n * 4));
    CHECK(hipMalloc(&actions, (size_t)n * K)); CHECK(hipMalloc(&reward, (size_t)n * K * 4)); CHECK(hipMalloc(&terminated, (size_t)n * K));
    CHECK(hipMemset(reward, 0, (size_t)n * K * 4)); CHECK(hipMemset(terminated, 0, (size_t)n * K)); CHECK(hipMemset(a.last_score, 0, (size_t)n * 4));
    hipLaunchKernelGGL(init_boards, dim3((n + 255) / 256), dim3(256), 0, 0, a.boards, a.score, actions, n, 42u);
    for (int j = 1; j < K; ++j) CHECK(hipMemcpy(actions + (size_t)j * n, actions, n, hipMemcpyDeviceToDevice));
    CHECK(hipDeviceSynchronize());

    const uint32_t full = (n + 255) / 256;
    const int ALL = F_SCORE | F_RECORD | F_COMPUTE | F_STORE;
    std::vector<Variant> vs = {
        {"v0 full, 1 board/lane, 64-bit idx", launch<ALL | F_IDX64>, full},
        {"v1 full, 1 board/lane, 32-bit idx", launch<ALL>, full},
        {"v2 full, stride loop 2048 blocks", launch<ALL>, 2048},
        {"v3 full, pipelined  2048 blocks", launch<ALL | F_PIPE>, 2048},
        {"v3b full, pipelined 1792 blocks", launch<ALL | F_PIPE>, 1792},
        {"v3c full, pipelined 1024 blocks", launch<ALL | F_PIPE>, 1024},
        {"v4 no score array, 1 board/lane", launch<(ALL & ~F_SCORE)>, full},
        {"v5 no record, 1 board/lane", launch<(ALL & ~F_RECORD)>, full},
        {"v6 compute only (no stores)", launch<F_COMPUTE | F_SCORE>, full},
        {"v7 memory only (no compute)", launch<F_SCORE | F_STORE>, full},
        {"v7b memory only, pipelined 2048", launch<F_SCORE | F_STORE | F_PIPE>, 2048},
        {"v8 no score, no record, pipelined 2048", launch<F_COMPUTE | F_STORE | F_PIPE>, 2048},
        {"v10a full, LDS cap 4 blocks/CU", launch_lds<ALL | F_IDX64, 40000>, full},
        {"v10b full, LDS cap 5 blocks/CU", launch_lds<ALL | F_IDX64, 32000>, full},
        {"v10c full, LDS cap 6 blocks/CU", launch_lds<ALL | F_IDX64, 26000>, full},
        {"v10d full, LDS cap 3 blocks/CU", launch_lds<ALL | F_IDX64, 53000>, full},
        {"v10e full, LDS cap 2 blocks/CU", launch_lds<ALL | F_IDX64, 80000>, full},
        {"v10f compute only, LDS cap 4 blocks/CU", launch_lds<F_COMPUTE | F_SCORE, 40000>, full},
        {"v10g compute only, LDS cap 2 blocks/CU", launch_lds<F_COMPUTE | F_SCORE, 80000>, full},
        {"v12a two boards/lane, both loads up front", launch<ALL | F_TWO>, full / 2},
        {"v12b two boards/lane, B requested when A arrives", launch<ALL | F_TWO_LATE>, full / 2},
        {"v12c two boards/lane, up front, 64-bit idx", launch<ALL | F_TWO | F_IDX64>, full / 2},
        {"v14 plain loads, plain stores", launch_nt<0, 0>, full},
        {"v14 nt loads, plain stores", launch_nt<1, 0>, full},
        {"v14 plain loads, nt stores", launch_nt<0, 1>, full},
        {"v14 nt loads, nt stores", launch_nt<1, 1>, full},
        {"v14 nt loads, nt stores, NO score array", launch_nt<1, 1, 0>, full},
        {"v16 phase shift blk&1, ~0.25us", launch_phase<0, 1>, full},
        {"v16 phase shift blk&1, ~0.5us", launch_phase<0, 2>, full},
        {"v16 phase shift blk&1, ~1us", launch_phase<0, 4>, full},
        {"v16 phase shift blk&1, ~1.5us", launch_phase<0, 6>, full},
        {"v16 phase shift blk&1, ~2us", launch_phase<0, 8>, full},
        {"v16 phase shift (blk>>8)&1, ~1us", launch_phase<8, 4>, full},
        {"v16 phase shift (blk>>8)&1, ~1.5us", launch_phase<8, 6>, full},
        {"v16 phase shift (blk>>3)&1, ~1us", launch_phase<3, 4>, full},
        {"v16 phase shift (blk>>3)&1, ~1.5us", launch_phase<3, 6>, full},
        {"v15 nt, 2 boards per lane", launch_nt_multi<2>, full},
        {"v15 nt, 4 boards per lane", launch_nt_multi<4>, full},
        {"v13 block 64", launch_bs<ALL | F_IDX64, 64>, full},
        {"v13 block 128", launch_bs<ALL | F_IDX64, 128>, full},
        {"v13 block 256", launch_bs<ALL | F_IDX64, 256>, full},
        {"v13 block 512", launch_bs<ALL | F_IDX64, 512>, full},
        {"v13 block 1024", launch_bs<ALL | F_IDX64, 1024>, full},
        {"v9a full, prio (blk>>8)&3", launch<ALL | F_PRIO8>, full},
        {"v9b full, prio (blk>>11)&3", launch<ALL | F_PRIO11>, full},
        {"v9c full, prio blk&3", launch<ALL | F_PRIOW>, full},
        {"v9d compute only, prio (blk>>8)&3", launch<F_COMPUTE | F_SCORE | F_PRIO8>, full},
    };
    // the product kernel through its own launcher
    {
        g_prod = StepArgs{};
        g_prod.st.boards = a.boards; g_prod.st.score = a.score; g_prod.st.last_score = a.last_score;
        CHECK(hipMalloc(&g_prod.st.wave_stats, ((size_t)n / 64 + 4) * sizeof(WaveStats)));
        CHECK(hipMemset(g_prod.st.wave_stats, 0, ((size_t)n / 64 + 4) * sizeof(WaveStats)));
        g_prod.n = n; g_prod.seed_lo = 42; g_prod.auto_reset = 1;
    }
    vs.push_back({"prod step_kernel<1> (u8 actions, reward+terminated)", [](const Args &aj, uint32_t) {
        StepArgs p = g_prod; p.actions = aj.actions; p.reward = aj.reward; p.terminated = aj.terminated; p.t_lo = aj.t_lo;
        launch_step(p, 1, 0); }, full});
    vs.push_back({"prod step_kernel<0> (synthetic actions)", [](const Args &aj, uint32_t) {
        StepArgs p = g_prod; p.reward = aj.reward; p.terminated = aj.terminated; p.t_lo = aj.t_lo;
        launch_step(p, 0, 0); }, full});
    vs.push_back({"prod step_kernel<1>, no outputs", [](const Args &aj, uint32_t) {
        StepArgs p = g_prod; p.actions = aj.actions; p.t_lo = aj.t_lo;
        launch_step(p, 1, 0); }, full});
    std::vector<std::vector<float>> times(vs.size());
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int r = 0; r < rounds + 2; ++r) {
        for (size_t v = 0; v < vs.size(); ++v) {
            CHECK(hipEventRecord(e0));
            for (int j = 0; j < K; ++j) {
                Args aj = a; aj.actions = actions + (size_t)j * n; aj.reward = reward + (size_t)j * n; aj.terminated = terminated + (size_t)j * n;
                aj.t_lo = 1 + r * K + j;
                vs[v].launch(aj, vs[v].blocks);
            }
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) times[v].push_back(ms * 1e3f / K);
        }
    }
    printf("boards 2^%d, %d rounds x %d launches; us per launch (median, min)\n", lg, rounds, K);
    for (size_t v = 0; v < vs.size(); ++v) {
        std::sort(times[v].begin(), times[v].end());
        const float med = times[v][times[v].size() / 2];
        printf("%-42s %8.2f %8.2f   -> %.3e steps/s, algorithmic %.0f GB/s\n", vs[v].name, med, times[v][0], n / (med * 1e-6), 38.0 * n / (med * 1e-6) / 1e9);
    }
    return 0;
}
