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
#include <type_traits>
#include <string>
#include <vector>

#include "_v1/g2048_kernels.hip"
#include "../../gym-2048_amd/csrc/g2048_kernels.hip"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Variant { std::string name; std::function<void(uint32_t t)> launch; };

// ---- experiment copy of step_kernel (same device functions) with I/O switches
enum { X_ACT_PACKED = 1, X_TERM_PACKED = 2, X_NO_REWARD = 4, X_NO_TERM = 8, X_NO_ACT = 16, X_NO_LASTREC = 32, X_NO_COUNT = 64,
       X_REWARD_U16 = 128, X_COUNT_RMW = 256, X_LASTREC_32 = 512, X_LASTREC_64 = 1024, X_PREFETCH = 2048,
       X_COUNT_SLOAD = 4096, X_LASTREC_DENSE = 8192, X_LASTREC_4B = 16384, X_LASTREC_RING = 32768,
       X_PREFETCH_BLOCKS = 65536, X_OFF32 = 131072, X_TERM_FIRST = 262144, X_REC_PLAIN = 524288, X_OUT_PLAIN = 1048576, X_CHEAP_RNG = 2097152, X_LASTREC_SCORE = 4194304, X_PRE_RESET = 8388608, X_LASTREC_NT = 16777216, X_LASTREC_SC1 = 33554432 };

template <class T> __device__ __forceinline__ T *off32(T *base, uint32_t i)
{
    return reinterpret_cast<T *>(reinterpret_cast<char *>(const_cast<std::remove_const_t<T> *>(base)) + i * static_cast<uint32_t>(sizeof(T)));
}

template <int X>
__global__ void __launch_bounds__(256) kern_x(const g2048::StepArgs p)
{
    using namespace g2048;
    __shared__ WaveTables s_tables[4];
    uint32_t block = blockIdx.x;
    if (X & X_PREFETCH_BLOCKS) {
        // the FIRST n/32768 blocks of the grid only touch the next step's action bytes (one dword per lane,
        // one per 128-byte line) and retire; the lines are in the Infinity Cache when the next launch reads them
        const uint32_t pf = (p.n + 32767u) / 32768u;
        if (block < pf) {
            const uint32_t line = block * 256u + threadIdx.x;
            if (line * 128u < p.n) {
                uint32_t v;
                const uint32_t *src = reinterpret_cast<const uint32_t *>(static_cast<const uint8_t *>(p.actions) + p.n) + line * 32u;
                asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(src) : "memory");
            }
            return;
        }
        block -= pf;
    }
    const uint32_t i_raw = block * 256 + threadIdx.x;
    const bool valid = i_raw < p.n;
    const uint32_t i = valid ? i_raw : p.n - 1u;
    const uint32_t lane = threadIdx.x & 63u;
    Board rec;
    if (X & X_OFF32) {
        const u32x4 v = __builtin_nontemporal_load(off32(reinterpret_cast<const u32x4 *>(p.st.boards), i));
        rec = Board{{v.x, v.y, v.z, v.w}};
    } else if (X & X_REC_PLAIN)
        rec = load_board(p.st.boards, i);
    else
        rec = load_board_nt(p.st.boards, i);
    const uint2 tables_piece = load_tables_piece();
    // lanes 0,1 of the block touch the NEXT step's 256 action bytes of this block (one dword per 128-byte
    // line), issued with the board load: the line is in the Infinity Cache when the next launch wants it
    uint32_t touched = 0;
    if ((X & X_PREFETCH) && threadIdx.x < 2u)
        touched = *(reinterpret_cast<const uint32_t *>(static_cast<const uint8_t *>(p.actions) + p.n + (size_t)block * 256u) + threadIdx.x * 32u);
    // the wave's counter pair through the scalar cache (uniform address)
    const uint32_t wave_id = __builtin_amdgcn_readfirstlane(i_raw >> 6);
    unsigned long long old_ep = 0, old_ill = 0;
    {
        const unsigned long long *c = p.st.ep_counters + 2u * wave_id;
        old_ep = c[0];
        old_ill = c[1];
    }
    uint32_t packed_actions = 0;
    if ((X & X_ACT_PACKED) && lane < 16u)
        packed_actions = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(static_cast<const uint8_t *>(p.actions) + (i_raw & ~63u)) + lane);
    Words w;
    if (X & X_CHEAP_RNG) { // NOT the spawn stream: a two-multiply hash, to see what the Philox block costs
        const uint32_t h = (p.board_offset + i) * 0x9E3779B1u + p.t_lo * 0x85EBCA77u + p.seed_lo;
        const uint32_t g = (h ^ (h >> 15)) * 0xC2B2AE3Du;
        w = Words{{g ^ (g >> 13), g * 0x27D4EB2Fu, h ^ g, h}};
    } else
        w = philox4x32_10(p.t_lo, p.t_hi, p.board_offset + i, 0u, p.seed_lo, p.seed_hi);
    uint32_t action;
    if (X & X_NO_ACT)
        action = w.w[3] >> 30;
    else if (X & X_ACT_PACKED) {
        const uint32_t d = __builtin_amdgcn_ds_bpermute((lane >> 2) << 2, packed_actions); // dword of lane / 4
        action = (d >> (8u * (lane & 3u))) & 3u;
    } else
        action = load_action<1>(p.actions, i, 0u);
    const LdsTables tb = stage_tables(s_tables, use_after(tables_piece, w.w[0]));
    uint32_t episodes = 0, illegal_ends = 0;
    StepOut o;
    if (X & X_PRE_RESET) {
        // both candidate fresh records (after a legal / an illegal last move) computed from the Philox words BEFORE the
        // board record is needed -- i.e. while its load is in flight; the reset itself is four selects
        Board fa = fresh_record_lut(w.w[1], w.w[2], tb), fb = fresh_record_lut(w.w[0], w.w[1], tb);
        asm volatile("" : "+v"(rec.r[0]), "+v"(rec.r[1]), "+v"(rec.r[2]), "+v"(rec.r[3])
                     : "v"(fa.r[0]), "v"(fa.r[1]), "v"(fa.r[2]), "v"(fa.r[3]), "v"(fb.r[0]), "v"(fb.r[1]), "v"(fb.r[2]), "v"(fb.r[3]));
        o = play_record(rec, action, w, p.max_exp, tb);
        record_episode_ends(p, i, o.terminated && valid, !o.legal, rec, episodes, illegal_ends);
        if (o.terminated && p.auto_reset != 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                rec.r[k] = bfi(o.legal_mask, fa.r[k], fb.r[k]);
        }
    } else if (X & X_TERM_FIRST) {
        // terminal record stored BEFORE the reset overwrites it in place: no second copy of the record is live
        o = step_record(rec, action, w, p.max_exp, false, tb);
        record_episode_ends(p, i, o.terminated && valid, !o.legal, rec, episodes, illegal_ends);
        if (o.terminated && p.auto_reset != 0) {
            const uint32_t lm = lanemask(o.legal);
            rec = fresh_record_lut(bfi(lm, w.w[1], w.w[0]), bfi(lm, w.w[2], w.w[1]), tb);
        }
    } else
        o = step_record(rec, action, w, p.max_exp, p.auto_reset != 0, tb);
    if (valid && (X & X_OFF32)) {
        const u32x4 v = {rec.r[0], rec.r[1], rec.r[2], rec.r[3]};
        __builtin_nontemporal_store(v, off32(reinterpret_cast<u32x4 *>(p.st.boards), i));
        __builtin_nontemporal_store(o.legal ? static_cast<float>(o.gain) : p.illegal_reward, off32(p.reward, i));
        __builtin_nontemporal_store(static_cast<uint8_t>(o.terminated ? 1 : 0), off32(p.terminated, i));
    } else if (valid) {
        if (X & X_REC_PLAIN)
            store_board(p.st.boards, i, rec);
        else
            store_board_nt(p.st.boards, i, rec);
        if (X & X_OUT_PLAIN) {
            p.reward[i] = o.legal ? static_cast<float>(o.gain) : p.illegal_reward;
            p.terminated[i] = static_cast<uint8_t>(o.terminated ? 1 : 0);
        } else
        if (!(X & X_NO_REWARD)) {
            if (X & X_REWARD_U16)
                __builtin_nontemporal_store(static_cast<uint16_t>(o.gain), reinterpret_cast<uint16_t *>(p.reward) + i);
            else
                __builtin_nontemporal_store(o.legal ? static_cast<float>(o.gain) : p.illegal_reward, p.reward + i);
        }
        if (!(X & (X_NO_TERM | X_TERM_PACKED | X_OUT_PLAIN)))
            __builtin_nontemporal_store(static_cast<uint8_t>(o.terminated ? 1 : 0), p.terminated + i);
    }
    if (X & X_TERM_PACKED) {
        const unsigned long long m = __ballot(o.terminated);
        if (lane < 16u) {
            const uint32_t nib = static_cast<uint32_t>(m >> (4u * lane)) & 0xfu;
            const uint32_t bytes = (nib * 0x00204081u) & 0x01010101u;
            __builtin_nontemporal_store(bytes, reinterpret_cast<uint32_t *>(p.terminated + (i_raw & ~63u)) + lane);
        }
    }
    if ((X & X_PREFETCH) && touched == 0x12345677u)
        p.st.ep_counters[0] = touched; // keeps the touch alive
    if (X & (X_LASTREC_NT | X_LASTREC_SC1)) {
        const bool fin = o.terminated && valid;
        const unsigned long long done = __ballot(fin);
        if (done) {
            if (fin) {
                if (X & X_LASTREC_NT)
                    store_board_nt(p.st.last_record, i, o.terminal);
                else {
                    const u32x4 v = {o.terminal.r[0], o.terminal.r[1], o.terminal.r[2], o.terminal.r[3]};
                    uint4 *dst = p.st.last_record + i;
                    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
                }
            }
            episodes += (uint32_t)__popcll(done);
            illegal_ends += (uint32_t)__popcll(__ballot(fin && !o.legal));
        }
    } else if (X & X_LASTREC_SCORE) {
        // what a 4-byte "final score" bookkeeping would cost: the potential (and so the score) of the finished
        // boards computed in-lane, ONE dword stored per finished board
        const bool fin = o.terminated && valid;
        const unsigned long long done = __ballot(fin);
        if (done) {
            if (fin)
                reinterpret_cast<uint32_t *>(p.st.last_record)[i] = record_score(o.terminal);
            episodes += (uint32_t)__popcll(done);
            illegal_ends += (uint32_t)__popcll(__ballot(fin && !o.legal));
        }
    } else if (X & (X_LASTREC_DENSE | X_LASTREC_4B | X_LASTREC_RING)) {
        const bool fin = o.terminated && valid;
        const unsigned long long done = __ballot(fin);
        if (done) {
            if (fin) {
                const uint4 v = make_uint4(o.terminal.r[0], o.terminal.r[1], o.terminal.r[2], o.terminal.r[3]);
                if (X & X_LASTREC_4B)
                    reinterpret_cast<uint32_t *>(p.st.last_record)[i] = v.x;      // one dword, board-indexed (sparse)
                else if (X & X_LASTREC_RING) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(done >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)done, 0u));
                    const uint32_t slot = ((uint32_t)old_ep + rank) & 63u;
                    p.st.last_record[(size_t)wave_id * 64u + slot] = make_uint4(v.x | (lane << 5 & 0xe0u) | ((lane >> 3) << 13), v.y, v.z, v.w);
                } else
                    p.st.last_record[(size_t)(i & ~63u) + (lane & 7u)] = v;       // dense dummy: 8 slots per wave (memory-effect probe)
            }
            episodes += (uint32_t)__popcll(done);
            illegal_ends += (uint32_t)__popcll(__ballot(fin && !o.legal));
        }
    } else if (X & (X_LASTREC_32 | X_LASTREC_64)) {
        const bool fin = o.terminated && valid;
        const unsigned long long done = __ballot(fin);
        if (done) {
            if (fin) {
                constexpr uint32_t STRIDE = (X & X_LASTREC_64) ? 4 : 2;
                uint4 *dst = p.st.last_record + (size_t)i * STRIDE;
                const uint4 v = make_uint4(o.terminal.r[0], o.terminal.r[1], o.terminal.r[2], o.terminal.r[3]);
                for (uint32_t q = 0; q < STRIDE; ++q)
                    dst[q] = v;
            }
            episodes += (uint32_t)__popcll(done);
            illegal_ends += (uint32_t)__popcll(__ballot(fin && !o.legal));
        }
    } else if (X & X_OFF32) {
        const bool fin = o.terminated && valid;
        const unsigned long long done = __ballot(fin);
        if (done) {
            if (fin)
                *off32(p.st.last_record, i) = make_uint4(o.terminal.r[0], o.terminal.r[1], o.terminal.r[2], o.terminal.r[3]);
            episodes += (uint32_t)__popcll(done);
            illegal_ends += (uint32_t)__popcll(__ballot(fin && !o.legal));
        }
    } else if (!(X & (X_NO_LASTREC | X_TERM_FIRST | X_LASTREC_SCORE | X_PRE_RESET | X_LASTREC_NT | X_LASTREC_SC1)))
        record_episode_ends(p, i, o.terminated && valid, !o.legal, o.terminal, episodes, illegal_ends);
    if (!(X & (X_NO_COUNT | X_COUNT_RMW))) {
        if (episodes != 0u && lane == 0u) {
            ulonglong2 *c = reinterpret_cast<ulonglong2 *>(p.st.ep_counters + 2u * wave_id);
            ulonglong2 v;
            v.x = old_ep + episodes; v.y = old_ill + illegal_ends;
            *c = v;
        }
    } else if (X & X_COUNT_RMW) {
        if (episodes != 0u && lane == 0u) {
            ulonglong2 *c = reinterpret_cast<ulonglong2 *>(p.st.ep_counters + 2u * (i_raw >> 6));
            ulonglong2 v = *c;
            v.x += episodes; v.y += illegal_ends;
            *c = v;
        }
    }
}

// ---- the product kernel body with FLAT kernel arguments, so that the first 16 dwords can be preloaded into
//      SGPRs by the command processor (-mllvm -amdgpu-kernarg-preload-count=16): no s_load + wait before the
//      board load can be addressed
template <int BS>
__global__ void __launch_bounds__(BS) kern_flat(uint4 *boards, const void *actions, float *reward, uint8_t *terminated,
                                                 uint4 *last_record, unsigned long long *ep_counters, uint32_t t_lo,
                                                 uint32_t seed_lo, uint32_t seed_hi, uint32_t board_offset, uint32_t t_hi,
                                                 float illegal_reward, uint32_t max_exp, uint32_t auto_reset, uint32_t n)
{
    using namespace g2048;
    __shared__ WaveTables s_tables[BS / 64];
    StepArgs p{};
    p.st.boards = boards; p.st.last_record = last_record; p.st.ep_counters = ep_counters;
    p.actions = actions; p.reward = reward; p.terminated = terminated;
    p.n = n; p.board_offset = board_offset; p.seed_lo = seed_lo; p.seed_hi = seed_hi; p.t_lo = t_lo; p.t_hi = t_hi;
    p.illegal_reward = illegal_reward; p.max_exp = max_exp; p.auto_reset = auto_reset;
    const uint32_t i_raw = blockIdx.x * BS + threadIdx.x;
    const uint32_t i = i_raw;
    Board rec = load_board_nt(p.st.boards, i);
    const uint2 tables_piece = load_tables_piece();
    const EpisodeCounters counters = load_episode_counters(p, i_raw);
    const Words w = philox4x32_10(p.t_lo, p.t_hi, p.board_offset + i, 0u, p.seed_lo, p.seed_hi);
    const uint32_t action = load_action<1>(p.actions, i, w.w[3]);
    const LdsTables tb = stage_tables(s_tables, use_after(tables_piece, w.w[0]));
    const StepOut o = play_record(rec, action, w, p.max_exp, tb);
    uint32_t episodes = 0, illegal_ends = 0;
    record_episode_ends(p, i, o.terminated, !o.legal, rec, episodes, illegal_ends);
    if (o.terminated && p.auto_reset != 0)
        reset_record(rec, o, w, tb);
    store_board_nt(p.st.boards, i, rec);
    __builtin_nontemporal_store(o.legal ? static_cast<float>(o.gain) : p.illegal_reward, p.reward + i);
    __builtin_nontemporal_store(static_cast<uint8_t>(o.terminated ? 1 : 0), p.terminated + i);
    flush_episode_counts(counters, episodes, illegal_ends, wave_sum_lane63(o.gain), 0ull);
}

template <int BS>
static void launch_flat(const g2048::StepArgs &a)
{
    hipLaunchKernelGGL(kern_flat<BS>, dim3(a.n / BS), dim3(BS), 0, 0, a.st.boards, a.actions, a.reward, a.terminated, a.st.last_record,
                       a.st.ep_counters, a.t_lo, a.seed_lo, a.seed_hi, a.board_offset, a.t_hi, a.illegal_reward, a.max_exp,
                       a.auto_reset, a.n);
}

template <int X>
static void launch_x(const g2048::StepArgs &a)
{
    hipLaunchKernelGGL((kern_x<X>), dim3((a.n + 255) / 256 + ((X & X_PREFETCH_BLOCKS) ? (a.n + 32767) / 32768 : 0)), dim3(256), 0, 0, a);
}

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
    CHECK(hipMalloc(&a2.st.last_record, (size_t)n * 64));
    CHECK(hipMalloc(&a2.st.ep_counters, (size_t)(n / 64 + 16) * 32));
    CHECK(hipMemset(a2.st.last_record, 0, (size_t)n * 16));
    CHECK(hipMemset(a2.st.ep_counters, 0, (size_t)(n / 64 + 16) * 32));
    uint8_t *actions, *term; float *reward;
    CHECK(hipMalloc(&actions, (size_t)n * (launches + 1)));
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

    vs.push_back({"x   copy of the product kernel (sanity: = v3 <1>)", [&](uint32_t j) { io2(j); launch_x<0>(a2); }});
    vs.push_back({"f   product body, flat kernel arguments (first 16 dwords preloaded into SGPRs when built with -mllvm -amdgpu-kernarg-preload-count=16)", [&](uint32_t j) { io2(j); launch_flat<256>(a2); }});
    vs.push_back({"f   the same, 64-lane blocks", [&](uint32_t j) { io2(j); launch_flat<64>(a2); }});
    vs.push_back({"f   the same, 128-lane blocks", [&](uint32_t j) { io2(j); launch_flat<128>(a2); }});
    vs.push_back({"f   the same, 512-lane blocks", [&](uint32_t j) { io2(j); launch_flat<512>(a2); }});
    vs.push_back({"f   the same, 1024-lane blocks", [&](uint32_t j) { io2(j); launch_flat<1024>(a2); }});
    vs.push_back({"x   a two-multiply hash instead of the Philox block (NOT the stream: cost probe)", [&](uint32_t j) { io2(j); launch_x<X_CHEAP_RNG>(a2); }});
    vs.push_back({"x   both fresh records precomputed while the board load is in flight; reset = 4 selects", [&](uint32_t j) { io2(j); launch_x<X_PRE_RESET>(a2); }});
    vs.push_back({"x   terminal record stored nt", [&](uint32_t j) { io2(j); launch_x<X_LASTREC_NT>(a2); }});
    vs.push_back({"x   terminal record stored sc1 (write-through)", [&](uint32_t j) { io2(j); launch_x<X_LASTREC_SC1>(a2); }});
    vs.push_back({"x   final SCORE (potential computed in-lane) stored as one dword instead of the terminal record", [&](uint32_t j) { io2(j); launch_x<X_LASTREC_SCORE>(a2); }});
    vs.push_back({"x   records with plain (cacheable) loads/stores, outputs nt", [&](uint32_t j) { io2(j); launch_x<X_REC_PLAIN>(a2); }});
    vs.push_back({"x   records nt, outputs plain", [&](uint32_t j) { io2(j); launch_x<X_OUT_PLAIN>(a2); }});
    vs.push_back({"x   everything plain", [&](uint32_t j) { io2(j); launch_x<X_REC_PLAIN | X_OUT_PLAIN>(a2); }});
    vs.push_back({"x   terminal record stored before the in-place reset (no live copy)", [&](uint32_t j) { io2(j); launch_x<X_TERM_FIRST>(a2); }});
    vs.push_back({"x   actions: 16 lanes load a dword, ds_bpermute", [&](uint32_t j) { io2(j); launch_x<X_ACT_PACKED>(a2); }});
    vs.push_back({"x   terminated: ballot -> 16 lanes store a dword", [&](uint32_t j) { io2(j); launch_x<X_TERM_PACKED>(a2); }});
    vs.push_back({"x   both packed", [&](uint32_t j) { io2(j); launch_x<X_ACT_PACKED | X_TERM_PACKED>(a2); }});
    vs.push_back({"x   no reward store", [&](uint32_t j) { io2(j); launch_x<X_NO_REWARD>(a2); }});
    vs.push_back({"x   no terminated store", [&](uint32_t j) { io2(j); launch_x<X_NO_TERM>(a2); }});
    vs.push_back({"x   no action load (synthetic)", [&](uint32_t j) { io2(j); launch_x<X_NO_ACT>(a2); }});
    vs.push_back({"x   no last_record store, no counters", [&](uint32_t j) { io2(j); launch_x<X_NO_LASTREC | X_NO_COUNT>(a2); }});
    vs.push_back({"x   no counters (atomics)", [&](uint32_t j) { io2(j); launch_x<X_NO_COUNT>(a2); }});
    vs.push_back({"x   reward stored as u16 (not the ABI; traffic probe)", [&](uint32_t j) { io2(j); launch_x<X_REWARD_U16>(a2); }});
    vs.push_back({"x   counters: plain load-add-store by lane 0 (late load)", [&](uint32_t j) { io2(j); launch_x<X_COUNT_RMW>(a2); }});
    vs.push_back({"x   last_record 32 B per board (full-sector store)", [&](uint32_t j) { io2(j); launch_x<X_LASTREC_32>(a2); }});
    vs.push_back({"x   last_record 64 B per board", [&](uint32_t j) { io2(j); launch_x<X_LASTREC_64>(a2); }});
    vs.push_back({"x   lanes 0,1 of each block touch the next step's actions", [&](uint32_t j) { io2(j); launch_x<X_PREFETCH>(a2); }});
    vs.push_back({"x   counters: scalar load early, lane-0 store late", [&](uint32_t j) { io2(j); launch_x<X_COUNT_SLOAD>(a2); }});
    vs.push_back({"x   last_record store to a dense dummy slot (memory-effect probe)", [&](uint32_t j) { io2(j); launch_x<X_LASTREC_DENSE>(a2); }});
    vs.push_back({"x   last_record store of ONE dword, board-indexed", [&](uint32_t j) { io2(j); launch_x<X_LASTREC_4B>(a2); }});
    vs.push_back({"x   per-wave ring of terminal records (cursor = episode count) + scalar counters", [&](uint32_t j) { io2(j); launch_x<X_LASTREC_RING>(a2); }});
    vs.push_back({"x   ring + scalar counters + action touch", [&](uint32_t j) { io2(j); launch_x<X_LASTREC_RING | X_PREFETCH>(a2); }});
    vs.push_back({"x   scalar counters + action touch", [&](uint32_t j) { io2(j); launch_x<X_COUNT_SLOAD | X_PREFETCH>(a2); }});
    vs.push_back({"x   scalar counters (= product now)", [&](uint32_t j) { io2(j); launch_x<X_COUNT_SLOAD>(a2); }});
    vs.push_back({"x   32-bit byte offsets (SGPR base + VGPR offset addressing), n <= 2^28 only", [&](uint32_t j) { io2(j); launch_x<X_OFF32>(a2); }});
    vs.push_back({"x   scalar counters + prefetch blocks at the head of the grid", [&](uint32_t j) { io2(j); launch_x<X_COUNT_SLOAD | X_PREFETCH_BLOCKS>(a2); }});

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
    // ---- the same 64 launches of the product kernel replayed as ONE hipGraph (is the dependent-kernel
    //      boundary any shorter?) and split over two streams (halves of the batch leapfrogging)
    {
        hipStream_t cs, s2;
        CHECK(hipStreamCreate(&cs)); CHECK(hipStreamCreate(&s2));
        hipGraph_t graph; hipGraphExec_t exec;
        CHECK(hipStreamBeginCapture(cs, hipStreamCaptureModeGlobal));
        for (int j = 0; j < launches; ++j) { io2(j); CHECK(g2048::launch_step(a2, 1, cs)); }
        CHECK(hipStreamEndCapture(cs, &graph));
        CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        std::vector<float> g_us, t_us;
        for (int r = 0; r < rounds + 1; ++r) {
            CHECK(hipEventRecord(e0, cs));
            CHECK(hipGraphLaunch(exec, cs));
            CHECK(hipEventRecord(e1, cs));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) g_us.push_back(ms * 1e3f / launches);
        }
        // S streams: boards [q n/S, (q+1) n/S) on stream q, each its own chain of launches
        hipEvent_t f1[8];
        hipStream_t ss[8];
        ss[0] = cs; ss[1] = s2;
        for (int q = 2; q < 8; ++q) CHECK(hipStreamCreate(&ss[q]));
        for (int q = 0; q < 8; ++q) CHECK(hipEventCreate(&f1[q]));
        std::vector<float> s_us[9];
        for (int S : {2, 4, 8}) {
            g2048::StepArgs part[8];
            for (int q = 0; q < S; ++q) {
                part[q] = a2;
                part[q].n = n / S; part[q].board_offset = q * (n / S);
                part[q].st.boards = a2.st.boards + (size_t)q * (n / S);
                part[q].st.last_record = a2.st.last_record + (size_t)q * (n / S);
                part[q].st.ep_counters = a2.st.ep_counters + (size_t)q * (n / S / 64) * 2;
            }
            for (int r = 0; r < rounds + 1; ++r) {
                CHECK(hipEventRecord(e0, cs));
                for (int q = 1; q < S; ++q) CHECK(hipStreamWaitEvent(ss[q], e0, 0));
                for (int j = 0; j < launches; ++j)
                    for (int q = 0; q < S; ++q) {
                        part[q].t_lo = 100 + j;
                        part[q].actions = actions + (size_t)j * n + (size_t)q * (n / S);
                        part[q].reward = reward + (size_t)j * n + (size_t)q * (n / S);
                        part[q].terminated = term + (size_t)j * n + (size_t)q * (n / S);
                        CHECK(g2048::launch_step(part[q], 1, ss[q]));
                    }
                for (int q = 1; q < S; ++q) { CHECK(hipEventRecord(f1[q], ss[q])); CHECK(hipStreamWaitEvent(cs, f1[q], 0)); }
                CHECK(hipEventRecord(e1, cs));
                CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (r > 0) s_us[S].push_back(ms * 1e3f / launches);
            }
            std::sort(s_us[S].begin(), s_us[S].end());
        }
        t_us = s_us[2];
        printf("4 streams, a quarter of the batch each:  %.2f us per step   (median), %.2f (min)\n", s_us[4][s_us[4].size() / 2], s_us[4][0]);
        printf("8 streams, an eighth of the batch each:  %.2f us per step   (median), %.2f (min)\n", s_us[8][s_us[8].size() / 2], s_us[8][0]);
        std::sort(g_us.begin(), g_us.end()); std::sort(t_us.begin(), t_us.end());
        printf("hipGraph replay of %d product launches: %.2f us per launch (median), %.2f (min)\n", launches, g_us[g_us.size() / 2], g_us[0]);
        printf("two streams, half the batch each:        %.2f us per step   (median), %.2f (min)\n", t_us[t_us.size() / 2], t_us[0]);
    }
    printf("boards 2^%d, %d rounds x %d launches; us per launch (median, min)\n", lg, rounds, launches);
    for (size_t v = 0; v < vs.size(); ++v) {
        std::sort(us[v].begin(), us[v].end());
        const float med = us[v][us[v].size() / 2], mn = us[v][0];
        printf("%-82s %8.2f %8.2f   -> algorithmic %5.0f GB/s\n", vs[v].name.c_str(), med, mn, 38.0 * n / (med * 1e-6) / 1e9);
    }
    return 0;
}
