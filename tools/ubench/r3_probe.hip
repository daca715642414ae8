// r3_probe.hip -- round-3 A/B harness (measurement tool, not product).  Includes the product's kernel file so that
// every variant is built from the same device functions.  All timings: HIP events around `launches` back-to-back
// launches, median over rounds, variants interleaved in one process.
//
//  part A  the plain step at 2^lg boards: product kernel, a MEMORY-ONLY kernel with the same loads and stores (no
//          arithmetic beyond a dependency), a COMPUTE-ONLY kernel (all arithmetic, no global traffic), and the
//          two-half-batches-on-two-streams overlap with occupancy capped by dynamic LDS (4 / 2 waves per SIMD) so
//          that consecutive kernels can co-reside.
//  part B  the step that also writes its one-hot observation: product kernel (LDS-transposed, coalesced), the
//          two-kernel form (step + onehot), a memory-only kernel of the same traffic shape, per-lane stores without
//          the LDS transpose, plain instead of nt stores, observation before the record store, other block sizes,
//          and a grouped form in which every wavefront walks G groups of 64 boards with the next group's record and
//          action loads issued before the current group's observation stores.
//  part C  the product's observation-writing kernel, u8 / f16 / f32 observations, at different occupancy caps.
// Usage: r3_probe [log2_boards] [rounds] [part: any of a b c]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../../gym-2048_amd/csrc/g2048_kernels.hip"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

using namespace g2048;

struct Variant { std::string name; double bytes_per_board; std::function<void(uint32_t j, hipStream_t s)> launch; };

struct Flat {
    uint4 *boards; const uint8_t *actions; unsigned long long *ep_counters; uint32_t seed_lo, t_lo, n; float *reward;
    uint8_t *terminated; uint4 *last_record; void *obs;
};

// ------------------------------------------------------------------------------------------------ part A
// the same global loads and stores as step_kernel<1,true,true,false>, nothing else (outputs depend on inputs)
__global__ void __launch_bounds__(256) mem_only_kernel(const Flat p)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    Board rec = load_board_nt(p.boards, i);
    const uint32_t a = __builtin_nontemporal_load(p.actions + i);
    rec.r[0] ^= a;
    store_board_nt(p.boards, i, rec);
    __builtin_nontemporal_store(static_cast<float>(rec.r[1] & 0xffu), p.reward + i);
    __builtin_nontemporal_store(static_cast<uint8_t>(rec.r[2] & 1u), p.terminated + i);
}

// all of the step's arithmetic on a synthetic record; a store that never happens keeps it alive
__global__ void __launch_bounds__(256) compute_only_kernel(const Flat p)
{
    __shared__ WaveTables s_tables[4];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const uint2 piece = load_tables_piece();
    const Words w = philox4x32_10(p.t_lo, 0u, i, 0u, p.seed_lo, 0u);
    Board rec = fresh_record(w.w[1], w.w[2]);
    rec.r[1] |= (w.w[0] >> 28) & 0x03030303u;
    const LdsTables tb = stage_tables(s_tables, use_after(piece, w.w[0]));
    const StepOut o = play_record(rec, w.w[3] >> 30, w, 0u, tb);
    if (o.terminated)
        reset_record(rec, o, w, tb);
    if ((rec.r[0] ^ rec.r[1] ^ rec.r[2] ^ rec.r[3] ^ o.gain) == 0x12345678u) // never
        store_board_nt(p.boards, i, rec);
}

// ------------------------------------------------------------------------------------------------ part B
enum { O_MEMONLY = 1, O_PERLANE = 2, O_PLAIN = 4, O_OBS_FIRST = 8, O_NO_OBS = 16, O_PHILOX_FIRST = 32 };

__device__ __forceinline__ void store_chunk(uint4 *out, uint64_t g, uint32_t a, uint32_t b, uint32_t c, uint32_t d, bool plain)
{
    if (plain)
        out[g] = make_uint4(a, b, c, d);
    else
        store_chunk_nt(out, g, a, b, c, d);
}

template <int X>
__device__ __forceinline__ void emit_u8(uint4 *wave_recs, const Board &rec, void *obs, uint32_t wave_first)
{
    const uint32_t lane = threadIdx.x & 63u;
    if (X & O_PERLANE) { // every lane writes its own board's 256 bytes: 16 stores at a 256-byte lane stride
        uint4 *out = static_cast<uint4 *>(obs) + static_cast<uint64_t>(wave_first + lane) * 16u;
        const uint32_t r2 = rec.r[2] & kCellBits, r3 = rec.r[3] & kCellBits;
#pragma unroll
        for (uint32_t c = 0; c < 16u; ++c) {
            const uint32_t splat = c * 0x01010101u;
            store_chunk(out, c, eq_ones(rec.r[0], splat), eq_ones(rec.r[1], splat), eq_ones(r2, splat), eq_ones(r3, splat), X & O_PLAIN);
        }
        return;
    }
    uint4 *out = static_cast<uint4 *>(obs) + static_cast<uint64_t>(wave_first) * 16u;
    if (X & O_MEMONLY) {
#pragma unroll
        for (uint32_t s = 0; s < 16u; ++s)
            store_chunk(out, s * 64u + lane, rec.r[0], rec.r[1], rec.r[2], s, X & O_PLAIN);
        return;
    }
    wave_recs[lane] = make_uint4(rec.r[0], rec.r[1], rec.r[2] & kCellBits, rec.r[3] & kCellBits);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t splat = (lane & 15u) * 0x01010101u;
#pragma unroll
    for (uint32_t s = 0; s < 16u; ++s) {
        const uint4 v = wave_recs[s * 4u + (lane >> 4)];
        store_chunk(out, s * 64u + lane, eq_ones(v.x, splat), eq_ones(v.y, splat), eq_ones(v.z, splat), eq_ones(v.w, splat), X & O_PLAIN);
    }
}

// G groups of 64 boards per wavefront; group g of block b = boards [(b * G + g) * BS, +BS)
template <int X, int BS, int G>
__global__ void __launch_bounds__(BS) obs_kernel(const Flat p)
{
    __shared__ WaveTables s_tables[BS / 64];
    __shared__ uint4 s_recs[BS];
    const uint2 piece = load_tables_piece();
    uint32_t i = blockIdx.x * (BS * G) + threadIdx.x;
    Board rec_next = load_board_nt(p.boards, i);
    uint32_t act_next = __builtin_nontemporal_load(p.actions + i);
    StepArgs sp{};
    sp.st.last_record = p.last_record;
    sp.st.ep_counters = p.ep_counters;
    LdsTables tb{};
    bool staged = false;
    Words w_all[G];
    if (X & O_PHILOX_FIRST) { // every group's Philox block before any loaded value is needed: work for the ramp
#pragma unroll
        for (int g = 0; g < G; ++g)
            w_all[g] = philox4x32_10(p.t_lo, 0u, i + g * BS, 0u, p.seed_lo, 0u);
    }
#pragma unroll
    for (int g = 0; g < G; ++g, i += BS) {
        Board rec = rec_next;
        const uint32_t action = act_next & 3u;
        if (g + 1 < G) {
            rec_next = load_board_nt(p.boards, i + BS);
            act_next = __builtin_nontemporal_load(p.actions + i + BS);
        }
        const EpisodeCounters counters = load_episode_counters(sp, i);
        const Words w = (X & O_PHILOX_FIRST) ? w_all[g] : philox4x32_10(p.t_lo, 0u, i, 0u, p.seed_lo, 0u);
        if (!staged) {
            tb = stage_tables(s_tables, use_after(piece, w.w[0]));
            staged = true;
        }
        StepOut o{};
        if (!(X & O_MEMONLY)) {
            o = play_record(rec, action, w, 0u, tb);
            uint32_t episodes = 0, illegal_ends = 0;
            record_episode_ends(sp, i, o.terminated, !o.legal, rec, episodes, illegal_ends);
            if (o.terminated)
                reset_record(rec, o, w, tb);
            flush_episode_counts(counters, episodes, illegal_ends, wave_sum_lane63(o.gain), 0ull);
        } else {
            rec.r[0] ^= action;
            o.gain = rec.r[1] & 0xffu;
            o.legal = true;
            o.terminated = rec.r[2] & 1u;
        }
        if ((X & O_OBS_FIRST) && !(X & O_NO_OBS))
            emit_u8<X>(s_recs + (threadIdx.x & ~63u), rec, p.obs, i & ~63u);
        store_board_nt(p.boards, i, rec);
        __builtin_nontemporal_store(static_cast<float>(o.gain), p.reward + i);
        __builtin_nontemporal_store(static_cast<uint8_t>(o.terminated ? 1 : 0), p.terminated + i);
        if (!(X & O_OBS_FIRST) && !(X & O_NO_OBS))
            emit_u8<X>(s_recs + (threadIdx.x & ~63u), rec, p.obs, i & ~63u);
    }
}

template <int X, int BS, int G>
static void launch_obs(const Flat &f, hipStream_t s, uint32_t dyn_lds_kib = 0)
{
    static uint32_t raised = 0;
    if (dyn_lds_kib > 48 && raised < dyn_lds_kib) {
        const hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(&obs_kernel<X, BS, G>), hipFuncAttributeMaxDynamicSharedMemorySize, dyn_lds_kib * 1024);
        if (err != hipSuccess) { printf("(hipFuncSetAttribute(%u KiB) -> %s)\n", dyn_lds_kib, hipGetErrorString(err)); (void)hipGetLastError(); }
        raised = dyn_lds_kib;
    }
    hipLaunchKernelGGL((obs_kernel<X, BS, G>), dim3(f.n / (BS * G)), dim3(BS), dyn_lds_kib * 1024u, s, f);
}

int main(int argc, char **argv)
{
    const int lg = argc > 1 ? atoi(argv[1]) : 20;
    const int rounds = argc > 2 ? atoi(argv[2]) : 11;
    const char *part = argc > 3 ? argv[3] : "ab";
    const uint32_t n = 1u << lg;
    const int launches = lg >= 23 ? 8 : 20;

    StepArgs a{};
    CHECK(hipMalloc(&a.st.boards, (size_t)n * 16));
    CHECK(hipMalloc(&a.st.last_record, (size_t)n * 16));
    CHECK(hipMalloc(&a.st.ep_counters, (size_t)(n / 64 + 16) * 32));
    CHECK(hipMemset(a.st.last_record, 0, (size_t)n * 16));
    CHECK(hipMemset(a.st.ep_counters, 0, (size_t)(n / 64 + 16) * 32));
    uint8_t *actions, *term; float *reward; uint8_t *obs;
    CHECK(hipMalloc(&actions, (size_t)n * launches));
    CHECK(hipMalloc(&term, (size_t)n * launches));
    CHECK(hipMalloc(&reward, (size_t)n * launches * 4));
    CHECK(hipMalloc(&obs, (size_t)n * launches * 256));
    CHECK(hipMemset(term, 0, (size_t)n * launches));
    CHECK(hipMemset(reward, 0, (size_t)n * launches * 4));
    CHECK(hipMemset(obs, 0, (size_t)n * launches * 256));
    a.n = n; a.seed_lo = 42; a.auto_reset = 1;
    CHECK(launch_fill_actions(actions, n, 0, 42, 0, 1, launches, 0));
    CHECK(launch_reset(a, 0, nullptr, 0));
    a.k_steps = 64; a.t_lo = 1;
    CHECK(launch_rollout_random(a, 0)); // steady-state mix of episode ages
    CHECK(hipDeviceSynchronize());

    hipStream_t s0, s1;
    CHECK(hipStreamCreate(&s0)); CHECK(hipStreamCreate(&s1));
    {
        int max_lds = 0, cu_lds = 0;
        CHECK(hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, 0));
        CHECK(hipDeviceGetAttribute(&cu_lds, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, 0));
        printf("LDS: %d bytes per block max, %d per CU\n", max_lds, cu_lds);
    }
    auto io = [&](uint32_t j) { a.t_lo = 100 + j; a.actions = actions + (size_t)j * n; a.reward = reward + (size_t)j * n; a.terminated = term + (size_t)j * n; a.obs = nullptr; };
    auto flat = [&](uint32_t j) { return Flat{a.st.boards, actions + (size_t)j * n, a.st.ep_counters, 42u, 100u + j, n, reward + (size_t)j * n, term + (size_t)j * n, a.st.last_record, obs + (size_t)j * n * 256}; };

    std::vector<Variant> vs;
    if (strchr(part, 'a')) {
        vs.push_back({"A  product step_kernel<1,FULL,STD>", 38, [&](uint32_t j, hipStream_t s) { io(j); (void)launch_step(a, 1, s); }});
        vs.push_back({"A  memory-only: the same loads and stores, no arithmetic", 38, [&](uint32_t j, hipStream_t s) { hipLaunchKernelGGL(mem_only_kernel, dim3(n / 256), dim3(256), 0, s, flat(j)); }});
        vs.push_back({"A  compute-only: all of the arithmetic, no global traffic", 38, [&](uint32_t j, hipStream_t s) { hipLaunchKernelGGL(compute_only_kernel, dim3(n / 256), dim3(256), 0, s, flat(j)); }});
        vs.push_back({"A  harness copy of the step without the observation (sanity: ~ product)", 38, [&](uint32_t j, hipStream_t s) { launch_obs<O_NO_OBS, 256, 1>(flat(j), s); }});
        vs.push_back({"A  grouped step: 2 groups per wavefront, next group's loads issued before this group's arithmetic", 38, [&](uint32_t j, hipStream_t s) { launch_obs<O_NO_OBS, 256, 2>(flat(j), s); }});
        vs.push_back({"A  grouped step: 4 groups per wavefront", 38, [&](uint32_t j, hipStream_t s) { launch_obs<O_NO_OBS, 256, 4>(flat(j), s); }});
        vs.push_back({"A  grouped step: 2 groups per wavefront, 64-lane blocks", 38, [&](uint32_t j, hipStream_t s) { launch_obs<O_NO_OBS, 64, 2>(flat(j), s); }});
        vs.push_back({"A  grouped step, 2 groups, BOTH Philox blocks before the first record is needed", 38, [&](uint32_t j, hipStream_t s) { launch_obs<O_NO_OBS | O_PHILOX_FIRST, 256, 2>(flat(j), s); }});
        vs.push_back({"A  grouped step, 4 groups, all Philox blocks first", 38, [&](uint32_t j, hipStream_t s) { launch_obs<O_NO_OBS | O_PHILOX_FIRST, 256, 4>(flat(j), s); }});
        vs.push_back({"A  grouped step, 2 groups, Philox first, 128-lane blocks", 38, [&](uint32_t j, hipStream_t s) { launch_obs<O_NO_OBS | O_PHILOX_FIRST, 128, 2>(flat(j), s); }});
        vs.push_back({"A  grouped step, 2 groups, Philox first, 512-lane blocks", 38, [&](uint32_t j, hipStream_t s) { launch_obs<O_NO_OBS | O_PHILOX_FIRST, 512, 2>(flat(j), s); }});
        vs.push_back({"A  1 group, Philox first (= product order; sanity)", 38, [&](uint32_t j, hipStream_t s) { launch_obs<O_NO_OBS | O_PHILOX_FIRST, 256, 1>(flat(j), s); }});
        vs.push_back({"A  grouped memory-only: 2 groups per wavefront", 38, [&](uint32_t j, hipStream_t s) { launch_obs<O_NO_OBS | O_MEMONLY, 256, 2>(flat(j), s); }});
    }
    if (strchr(part, 'b')) {
        vs.push_back({"B  product step_kernel<1,FULL,STD,HAS_OBS> (LDS transpose, coalesced nt stores)", 294, [&](uint32_t j, hipStream_t s) { io(j); a.obs = obs + (size_t)j * n * 256; a.obs_dtype = 0; (void)launch_step(a, 1, s); }});
        vs.push_back({"B  two kernels: step_kernel + onehot_kernel<u8>", 310, [&](uint32_t j, hipStream_t s) { io(j); (void)launch_step(a, 1, s); (void)launch_onehot(a.st.boards, n, obs + (size_t)j * n * 256, 0, s); }});
        vs.push_back({"B  onehot_kernel<u8> alone", 272, [&](uint32_t j, hipStream_t s) { (void)launch_onehot(a.st.boards, n, obs + (size_t)j * n * 256, 0, s); }});
        vs.push_back({"B  harness copy of the fused kernel (sanity: = product)", 294, [&](uint32_t j, hipStream_t s) { launch_obs<0, 256, 1>(flat(j), s); }});
        vs.push_back({"B  memory-only kernel of the same traffic shape (no arithmetic, no LDS)", 294, [&](uint32_t j, hipStream_t s) { launch_obs<O_MEMONLY, 256, 1>(flat(j), s); }});
        vs.push_back({"B  per-lane stores (no LDS transpose): 16 x dwordx4 at a 256-byte lane stride", 294, [&](uint32_t j, hipStream_t s) { launch_obs<O_PERLANE, 256, 1>(flat(j), s); }});
        vs.push_back({"B  plain (cacheable) observation stores", 294, [&](uint32_t j, hipStream_t s) { launch_obs<O_PLAIN, 256, 1>(flat(j), s); }});
        vs.push_back({"B  observation before the record / reward / terminated stores", 294, [&](uint32_t j, hipStream_t s) { launch_obs<O_OBS_FIRST, 256, 1>(flat(j), s); }});
        vs.push_back({"B  64-lane blocks", 294, [&](uint32_t j, hipStream_t s) { launch_obs<0, 64, 1>(flat(j), s); }});
        vs.push_back({"B  128-lane blocks", 294, [&](uint32_t j, hipStream_t s) { launch_obs<0, 128, 1>(flat(j), s); }});
        vs.push_back({"B  512-lane blocks", 294, [&](uint32_t j, hipStream_t s) { launch_obs<0, 512, 1>(flat(j), s); }});
        vs.push_back({"B  grouped: 2 groups per wavefront, next group's loads before this group's stores", 294, [&](uint32_t j, hipStream_t s) { launch_obs<0, 256, 2>(flat(j), s); }});
        vs.push_back({"B  grouped: 4 groups per wavefront", 294, [&](uint32_t j, hipStream_t s) { launch_obs<0, 256, 4>(flat(j), s); }});
        vs.push_back({"B  grouped: 8 groups per wavefront", 294, [&](uint32_t j, hipStream_t s) { launch_obs<0, 256, 8>(flat(j), s); }});
        vs.push_back({"B  grouped: 4 groups, 64-lane blocks", 294, [&](uint32_t j, hipStream_t s) { launch_obs<0, 64, 4>(flat(j), s); }});
        vs.push_back({"B  occupancy capped: 4 blocks of 256 per CU = 4 waves/SIMD (34 KiB dynamic LDS)", 294, [&](uint32_t j, hipStream_t s) { launch_obs<0, 256, 1>(flat(j), s, 34); }});
        vs.push_back({"B  occupancy capped: 3 blocks per CU (46 KiB)", 294, [&](uint32_t j, hipStream_t s) { launch_obs<0, 256, 1>(flat(j), s, 46); }});
        vs.push_back({"B  occupancy capped: 2 blocks per CU = 2 waves/SIMD (58 KiB)", 294, [&](uint32_t j, hipStream_t s) { launch_obs<0, 256, 1>(flat(j), s, 58); }});
        vs.push_back({"B  512-lane blocks, 2 blocks per CU = 4 waves/SIMD (52 KiB)", 294, [&](uint32_t j, hipStream_t s) { launch_obs<0, 512, 1>(flat(j), s, 52); }});
        vs.push_back({"B  512-lane blocks, 3 blocks per CU = 6 waves/SIMD (36 KiB)", 294, [&](uint32_t j, hipStream_t s) { launch_obs<0, 512, 1>(flat(j), s, 36); }});
        vs.push_back({"B  1024-lane blocks", 294, [&](uint32_t j, hipStream_t s) { launch_obs<0, 1024, 1>(flat(j), s); }});
        vs.push_back({"B  memory-only shape, 2 blocks per CU (58 KiB)", 294, [&](uint32_t j, hipStream_t s) { launch_obs<O_MEMONLY, 256, 1>(flat(j), s, 58); }});
        vs.push_back({"B  grouped memory-only: 4 groups per wavefront", 294, [&](uint32_t j, hipStream_t s) { launch_obs<O_MEMONLY, 256, 4>(flat(j), s); }});
    }

    hipEvent_t e0, e1, f1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&f1));
    std::vector<std::vector<float>> us(vs.size());
    for (int r = 0; r < rounds + 1; ++r)
        for (size_t v = 0; v < vs.size(); ++v) {
            CHECK(hipEventRecord(e0, s0));
            for (int j = 0; j < launches; ++j) vs[v].launch(j, s0);
            CHECK(hipEventRecord(e1, s0));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) us[v].push_back(ms * 1e3f / launches);
        }
    printf("boards 2^%d, %d rounds x %d launches; us per launch (median, min)\n", lg, rounds, launches);
    for (size_t v = 0; v < vs.size(); ++v) {
        std::sort(us[v].begin(), us[v].end());
        const float med = us[v][us[v].size() / 2];
        printf("%-92s %8.2f %8.2f   -> %5.0f GB/s on %3.0f B/board\n", vs[v].name.c_str(), med, us[v][0],
               vs[v].bytes_per_board * n / (med * 1e-6) / 1e9, vs[v].bytes_per_board);
    }

    // ---- part C: the product's observation-writing kernel for every observation dtype at different occupancy caps
    //      (dynamic LDS per workgroup on top of the 6 KiB it uses: 0 -> 8 workgroups per CU by VGPRs, 26 KiB -> 5,
    //      34 -> 4, 42 -> 3 [what the product launches with], 58 -> 2)
    if (strchr(part, 'c')) {
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&step_kernel<1, true, true, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        uint8_t *big;
        const int lc = launches < 8 ? launches : 8;
        CHECK(hipMalloc(&big, (size_t)n * lc * 1024));
        CHECK(hipMemset(big, 0, (size_t)n * lc * 1024));
        for (uint32_t dt = 0; dt < 3; ++dt)
            for (uint32_t pad : {0u, 26u, 34u, 42u, 58u}) {
                std::vector<float> t;
                for (int r = 0; r < rounds + 1; ++r) {
                    CHECK(hipEventRecord(e0, s0));
                    for (int j = 0; j < lc; ++j) {
                        io(j);
                        const StepTail tail{a.terminated, a.st.last_record, nullptr, nullptr, nullptr, 0.0f, 0u, 1u,
                                            big + ((size_t)j * n << (8 + dt)), dt, nullptr, nullptr, 0ull};
                        hipLaunchKernelGGL((step_kernel<1, true, true, true>), dim3(n / 256), dim3(256), pad * 1024u, s0, a.st.boards,
                                           a.actions, a.st.ep_counters, a.board_offset, a.seed_lo, a.seed_hi, a.t_lo, a.t_hi, a.n, a.reward, tail);
                    }
                    CHECK(hipEventRecord(e1, s0));
                    CHECK(hipEventSynchronize(e1));
                    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                    if (r > 0) t.push_back(ms * 1e3f / lc);
                }
                std::sort(t.begin(), t.end());
                const double bytes = 38.0 + (256 << dt);
                printf("C  product HAS_OBS kernel, %s observation, %2u KiB pad: %8.2f us per launch (median) -> %5.0f GB/s on %4.0f B/board\n",
                       dt == 0 ? "u8 " : (dt == 1 ? "f16" : "f32"), pad, t[t.size() / 2], bytes * n / (t[t.size() / 2] * 1e-6) / 1e9, bytes);
            }
        CHECK(hipFree(big));
    }

    // ---- part A, overlap: the two halves of the batch as independent chains on two streams, the kernels' occupancy
    //      capped with dynamic LDS so that a half-batch kernel does NOT fill every wave slot of the chip
    if (strchr(part, 'a')) {
        for (uint32_t dyn_kib : {0u, 18u, 38u, 60u}) { // 0: uncapped; 18 KiB + 2 static -> 8 blocks/CU = 8 waves/SIMD ... 60 -> 2 blocks/CU
            StepArgs h[2];
            for (int q = 0; q < 2; ++q) {
                h[q] = a;
                h[q].n = n / 2; h[q].board_offset = q * (n / 2);
                h[q].st.boards = a.st.boards + (size_t)q * (n / 2);
                h[q].st.last_record = a.st.last_record + (size_t)q * (n / 2);
                h[q].st.ep_counters = a.st.ep_counters + (size_t)q * (n / 2 / 64) * g2048::kSlotWords;
            }
            auto launch_half = [&](const StepArgs &x, hipStream_t s) {
                const StepTail tail{x.terminated, x.st.last_record, nullptr, nullptr, nullptr, 0.0f, 0u, 1u, nullptr, 0u};
                hipLaunchKernelGGL((step_kernel<1, true, true, false>), dim3(x.n / 256), dim3(256), dyn_kib * 1024u, s, x.st.boards,
                                   x.actions, x.st.ep_counters, x.board_offset, x.seed_lo, x.seed_hi, x.t_lo, x.t_hi, x.n, x.reward, tail);
            };
            if (dyn_kib > 48)
                CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&step_kernel<1, true, true, false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, dyn_kib * 1024));
            std::vector<float> two, one;
            for (int r = 0; r < rounds + 1; ++r) {
                // one stream, whole batch, same cap
                CHECK(hipEventRecord(e0, s0));
                for (int j = 0; j < launches; ++j) {
                    io(j);
                    const StepTail tail{a.terminated, a.st.last_record, nullptr, nullptr, nullptr, 0.0f, 0u, 1u, nullptr, 0u};
                    hipLaunchKernelGGL((step_kernel<1, true, true, false>), dim3(n / 256), dim3(256), dyn_kib * 1024u, s0, a.st.boards,
                                       a.actions, a.st.ep_counters, a.board_offset, a.seed_lo, a.seed_hi, a.t_lo, a.t_hi, a.n, a.reward, tail);
                }
                CHECK(hipEventRecord(e1, s0));
                CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (r > 0) one.push_back(ms * 1e3f / launches);
                // two streams, half a batch each
                CHECK(hipEventRecord(e0, s0));
                CHECK(hipStreamWaitEvent(s1, e0, 0));
                for (int j = 0; j < launches; ++j)
                    for (int q = 0; q < 2; ++q) {
                        h[q].t_lo = 100 + j;
                        h[q].actions = actions + (size_t)j * n + (size_t)q * (n / 2);
                        h[q].reward = reward + (size_t)j * n + (size_t)q * (n / 2);
                        h[q].terminated = term + (size_t)j * n + (size_t)q * (n / 2);
                        launch_half(h[q], q ? s1 : s0);
                    }
                CHECK(hipEventRecord(f1, s1));
                CHECK(hipStreamWaitEvent(s0, f1, 0));
                CHECK(hipEventRecord(e1, s0));
                CHECK(hipEventSynchronize(e1));
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (r > 0) two.push_back(ms * 1e3f / launches);
            }
            std::sort(two.begin(), two.end()); std::sort(one.begin(), one.end());
            printf("A  dynamic LDS %2u KiB per block: one stream whole batch %6.2f us per step | two streams, half the batch each %6.2f us per step (median of %d-launch trains)\n",
                   dyn_kib, one[one.size() / 2], two[two.size() / 2], launches);
        }
    }
    return 0;
}
