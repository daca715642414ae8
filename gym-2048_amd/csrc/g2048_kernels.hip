// g2048_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the batched 2048 environment.
//
// Mapping: ONE BOARD PER LANE.  A wavefront owns 64 consecutive boards = 1 KiB of board state, read
// and written with one global_load/store_dwordx4 per lane (fully coalesced, 16 B/lane).  The whole
// step -- slide/merge sweep, score, spawn, done detection, auto-reset -- runs out of VGPRs with
// byte-parallel integer ops (g2048_device.h); there is no cross-lane traffic, no LDS and no
// per-board RNG state: the spawn randomness of (transaction t, board b) is one Philox4x32-10 block.
//
// HBM bytes per env-step of step_kernel (the roofline figure in DESIGN.md):
//   algorithmic 38 B = board in 16 + action 1 + board out 16 + reward 4 + terminated 1
//   plus the episodic score state (4 B in + 4 B out) and, only for boards that terminate, the
//   episode record.
#include "g2048_kernels.h"

#include "g2048_device.h"
#include "g2048_pcg64.h"


namespace g2048 {

constexpr int kBlock = 256;

// Streaming accesses of the step kernels carry the non-temporal hint (nt=1): every byte is touched
// exactly once per launch, so it should not displace anything in L2 / Infinity Cache.  Measured on
// MI355X (tools/ubench/step_variants.hip, v14): 2^24 boards 181 -> 128 us per launch, 2^20 boards
// 10.9 -> 10.5 us.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ Board load_board_nt(const uint4 *boards, uint32_t i)
{
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(boards) + i);
    return Board{{v.x, v.y, v.z, v.w}};
}

__device__ __forceinline__ void store_board_nt(uint4 *boards, uint32_t i, const Board &b)
{
    const u32x4 v = {b.r[0], b.r[1], b.r[2], b.r[3]};
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(boards) + i);
}

template <int ACT>
__device__ __forceinline__ uint32_t load_action(const void *actions, uint32_t i, uint32_t w3)
{
    if constexpr (ACT == 0)
        return w3 >> 30;
    else if constexpr (ACT == 1)
        return __builtin_nontemporal_load(static_cast<const uint8_t *>(actions) + i) & 3u;
    else if constexpr (ACT == 2)
        return static_cast<uint32_t>(__builtin_nontemporal_load(static_cast<const int32_t *>(actions) + i)) & 3u;
    else
        return static_cast<uint32_t>(__builtin_nontemporal_load(static_cast<const long long *>(actions) + i)) & 3u;
}

// Wave-wide sum and max of a non-negative per-lane value, result valid in lane 63.  Seven DPP steps
// each (row_shr 1,2,3,4,8 then row_bcast 15, 31): pure VALU, no LDS, no scalar loop.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ int dpp_shift(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, BANK_MASK, true);
}

__device__ __forceinline__ int wave_sum(int v) // result in lane 63
{
    int s = v + dpp_shift<0x111, 0xf, 0xf>(v) + dpp_shift<0x112, 0xf, 0xf>(v) + dpp_shift<0x113, 0xf, 0xf>(v);
    s += dpp_shift<0x114, 0xf, 0xe>(s);
    s += dpp_shift<0x118, 0xf, 0xc>(s);
    s += dpp_shift<0x142, 0xa, 0xf>(s); // row_bcast:15 into rows 1 and 3
    s += dpp_shift<0x143, 0xc, 0xf>(s); // row_bcast:31 into rows 2 and 3
    return s;
}

__device__ __forceinline__ int wave_max(int v) // non-negative v; result broadcast to the whole wave
{
    int m = max(max(v, dpp_shift<0x111, 0xf, 0xf>(v)), max(dpp_shift<0x112, 0xf, 0xf>(v), dpp_shift<0x113, 0xf, 0xf>(v)));
    m = max(m, dpp_shift<0x114, 0xf, 0xe>(m));
    m = max(m, dpp_shift<0x118, 0xf, 0xc>(m));
    m = max(m, dpp_shift<0x142, 0xa, 0xf>(m));
    m = max(m, dpp_shift<0x143, 0xc, 0xf>(m));
    return __builtin_amdgcn_readlane(m, 63);
}

// Episode bookkeeping.  Boards that ended an episode write their final score (and, if asked, their
// terminal board); the wave's totals are accumulated in lane 63.  The whole block is skipped by a
// wave-uniform branch when no lane terminated.
struct WaveAcc {
    unsigned int episodes = 0, illegal_ends = 0;
    unsigned long long score_sum = 0; // valid in lane 63 only
    int max_score = 0;                // wave-uniform
};

// `best_so_far`: the wave's recorded best final score (uniform).  The max reduction only runs when
// some lane beats it, which after the first few hundred episodes of a wave practically never happens.
__device__ __forceinline__ void record_episodes(const StepArgs &p, uint32_t i, const StepResult &r, WaveAcc &acc,
                                                int best_so_far)
{
    const unsigned long long done = __ballot(r.terminated);
    if (done == 0)
        return;
    // plain (cached) stores on purpose: these are sparse 4/16-byte writes, which L2 merges into
    // lines; with the nt hint they cost 12 % of the launch at 2^24 boards
    if (r.terminated) {
        p.st.last_score[i] = r.terminal_score;
        if (p.terminal_boards)
            p.terminal_boards[i] = make_uint4(r.terminal.r[0], r.terminal.r[1], r.terminal.r[2], r.terminal.r[3]);
    }
    acc.episodes += static_cast<unsigned int>(__popcll(done));
    acc.illegal_ends += static_cast<unsigned int>(__popcll(__ballot(r.terminated && r.illegal)));
    const int v = r.terminated ? r.terminal_score : 0;
    acc.score_sum += static_cast<unsigned long long>(static_cast<long long>(wave_sum(v)));
    if (__ballot(v > max(best_so_far, acc.max_score)) != 0ull)
        acc.max_score = max(acc.max_score, wave_max(v));
}

// The wave's slot is private to it (one wave per slot per launch, launches are stream-ordered), so
// the accumulators are updated with a plain load-add-store by ONE lane (63, where the DPP
// reductions land).  `old` is loaded at kernel entry, together with the board, so its latency is
// never exposed.  Requires full wavefronts: the launchers pad the tail wave's bookkeeping by
// running it with all 64 lanes active (boards beyond n are never touched).
__device__ __forceinline__ void flush_wave_stats(const StepArgs &p, uint32_t i, const WaveStats &old, const WaveAcc &acc)
{
    if (acc.episodes == 0 || (threadIdx.x & 63u) != 63u)
        return;
    WaveStats ws;
    ws.episodes = old.episodes + acc.episodes;
    ws.illegal_ends = old.illegal_ends + acc.illegal_ends;
    ws.score_sum = old.score_sum + acc.score_sum;
    ws.max_score = max(old.max_score, acc.max_score);
    ws.pad = 0;
    p.st.wave_stats[i >> 6] = ws;
}

// ---------------------------------------------------------------------------------- step
// Game2048Env.step for every board (game2048_env.py:76-100), one launch per environment step.
//
// One board per lane, one pass: load (board 16 B, score 4 B, action) -> ~310 VALU instructions ->
// store.  The Philox block does not depend on the loaded data, so its ~55 instructions run while
// the loads are in flight.  Measured alternatives that were NOT faster on MI355X at 2^20 boards
// (tools/ubench/step_variants.hip): grid-stride loops with the next board prefetched, per-block
// s_setprio staggering, 32-bit offset addressing.
template <int ACT>
__global__ void __launch_bounds__(kBlock) step_kernel(const StepArgs p)
{
    // Lanes past the end stay active (the DPP reductions and the lane-63 flush need whole
    // wavefronts): they recompute board n-1 and write nothing.
    const uint32_t i_raw = blockIdx.x * kBlock + threadIdx.x;
    const bool valid = i_raw < p.n;
    const uint32_t i = valid ? i_raw : p.n - 1u;
    Board bd = load_board_nt(p.st.boards, i);
    int32_t score = __builtin_nontemporal_load(p.st.score + i);
    const WaveStats old_stats = p.st.wave_stats[i_raw >> 6]; // same address in all lanes: one request

    const Words w = philox4x32_10(p.t_lo, p.t_hi, p.board_offset + i, 0u, p.seed_lo, p.seed_hi);
    const uint32_t action = load_action<ACT>(p.actions, i, w.w[3]);

    StepResult r = step_env(bd, score, action, w, p.illegal_reward, p.max_exp, p.auto_reset != 0);

    if (valid) {
        store_board_nt(p.st.boards, i, bd);
        __builtin_nontemporal_store(score, p.st.score + i);
        if (p.reward)
            __builtin_nontemporal_store(r.reward, p.reward + i);
        if (p.terminated)
            __builtin_nontemporal_store(static_cast<uint8_t>(r.terminated ? 1 : 0), p.terminated + i);
        if (p.illegal)
            __builtin_nontemporal_store(static_cast<uint8_t>(r.illegal ? 1 : 0), p.illegal + i);
        if (p.highest)
            __builtin_nontemporal_store(static_cast<uint8_t>(highest(r.terminal)), p.highest + i); // :97
    }
    r.terminated = r.terminated && valid;
    WaveAcc acc;
    record_episodes(p, i, r, acc, old_stats.max_score);
    flush_wave_stats(p, i_raw, old_stats, acc);
}

// ------------------------------------------------------------------------- fused rollout
// k steps of the synthetic random policy in ONE launch; the board never leaves registers.
__global__ void __launch_bounds__(kBlock) rollout_random_kernel(const StepArgs p)
{
    const uint32_t i_raw = blockIdx.x * kBlock + threadIdx.x;
    const bool valid = i_raw < p.n;
    const uint32_t i = valid ? i_raw : p.n - 1u;
    const uint4 v = p.st.boards[i];
    Board bd{{v.x, v.y, v.z, v.w}};
    int32_t score = p.st.score[i];
    uint64_t t = (static_cast<uint64_t>(p.t_hi) << 32) | p.t_lo; // transaction of the first step
    WaveAcc acc;
    for (uint32_t j = 0; j < p.k_steps; ++j, ++t) {
        const Words w = philox4x32_10(static_cast<uint32_t>(t), static_cast<uint32_t>(t >> 32), p.board_offset + i, 0u,
                                      p.seed_lo, p.seed_hi);
        StepResult r = step_env(bd, score, w.w[3] >> 30, w, p.illegal_reward, p.max_exp, true);
        r.terminated = r.terminated && valid;
        record_episodes(p, i, r, acc, 0);
    }
    if (valid) {
        p.st.boards[i] = make_uint4(bd.r[0], bd.r[1], bd.r[2], bd.r[3]);
        p.st.score[i] = score;
    }
    flush_wave_stats(p, i_raw, p.st.wave_stats[i_raw >> 6], acc);
}

// ---------------------------------------------------------------- fused rollout with per-step I/O
// The same k steps g2048_rollout performs with k launches, in ONE launch: boards and scores stay in
// registers, each step reads action[j][i] and writes reward[j][i] / terminated[j][i] (stride = elements
// between consecutive steps).  Bit-identical outputs; 6 B of traffic per env-step instead of 46.
template <int ACT>
__global__ void __launch_bounds__(kBlock) rollout_fused_kernel(const StepArgs p, uint64_t stride)
{
    const uint32_t i_raw = blockIdx.x * kBlock + threadIdx.x;
    const bool valid = i_raw < p.n;
    const uint32_t i = valid ? i_raw : p.n - 1u;
    Board bd = load_board_nt(p.st.boards, i);
    int32_t score = p.st.score[i];
    const WaveStats old_stats = p.st.wave_stats[i_raw >> 6];
    uint64_t t = (static_cast<uint64_t>(p.t_hi) << 32) | p.t_lo;
    WaveAcc acc;
    for (uint32_t j = 0; j < p.k_steps; ++j, ++t) {
        const size_t o = static_cast<size_t>(j) * stride + i;
        const Words w = philox4x32_10(static_cast<uint32_t>(t), static_cast<uint32_t>(t >> 32), p.board_offset + i, 0u,
                                      p.seed_lo, p.seed_hi);
        uint32_t action;
        if constexpr (ACT == 0)
            action = w.w[3] >> 30;
        else if constexpr (ACT == 1)
            action = __builtin_nontemporal_load(static_cast<const uint8_t *>(p.actions) + o) & 3u;
        else if constexpr (ACT == 2)
            action = static_cast<uint32_t>(__builtin_nontemporal_load(static_cast<const int32_t *>(p.actions) + o)) & 3u;
        else
            action = static_cast<uint32_t>(__builtin_nontemporal_load(static_cast<const long long *>(p.actions) + o)) & 3u;
        StepResult r = step_env(bd, score, action, w, p.illegal_reward, p.max_exp, p.auto_reset != 0);
        if (valid) {
            if (p.reward)
                __builtin_nontemporal_store(r.reward, p.reward + o);
            if (p.terminated)
                __builtin_nontemporal_store(static_cast<uint8_t>(r.terminated ? 1 : 0), p.terminated + o);
            if (p.illegal)
                __builtin_nontemporal_store(static_cast<uint8_t>(r.illegal ? 1 : 0), p.illegal + o);
            if (p.highest)
                __builtin_nontemporal_store(static_cast<uint8_t>(highest(r.terminal)), p.highest + o);
        }
        r.terminated = r.terminated && valid;
        record_episodes(p, i, r, acc, max(old_stats.max_score, acc.max_score));
    }
    if (valid) {
        store_board_nt(p.st.boards, i, bd);
        p.st.score[i] = score;
    }
    flush_wave_stats(p, i_raw, old_stats, acc);
}

// ------------------------------------------------------------------------- numpy-RNG mode
// Same step / reset / add_tile, drawing from each board's own PCG64 exactly as numpy would
// (g2048_pcg64.h).  RNG state: five coalesced 8-byte planes.
__device__ __forceinline__ Pcg64 load_rng(const uint64_t *planes, uint32_t n, uint32_t i)
{
    return Pcg64{planes[i], planes[n + i], planes[2ull * n + i], planes[3ull * n + i], planes[4ull * n + i]};
}

__device__ __forceinline__ void store_rng(uint64_t *planes, uint32_t n, uint32_t i, const Pcg64 &r)
{
    planes[i] = r.state_lo;
    planes[n + i] = r.state_hi;
    planes[4ull * n + i] = r.buf; // inc never changes
}

template <int ACT>
__global__ void __launch_bounds__(kBlock) step_numpy_kernel(const StepArgs p)
{
    const uint32_t i_raw = blockIdx.x * kBlock + threadIdx.x;
    const bool valid = i_raw < p.n;
    const uint32_t i = valid ? i_raw : p.n - 1u;
    const uint4 v = p.st.boards[i];
    Board bd{{v.x, v.y, v.z, v.w}};
    int32_t score = p.st.score[i];
    Pcg64 rng = load_rng(p.st.rng, p.n, i);
    const WaveStats old_stats = p.st.wave_stats[i_raw >> 6];
    uint32_t action;
    if constexpr (ACT == 0)
        action = philox4x32_10(p.t_lo, p.t_hi, p.board_offset + i, 0u, p.seed_lo, p.seed_hi).w[3] >> 30;
    else
        action = load_action<ACT>(p.actions, i, 0u);

    StepResult r = step_env_numpy(bd, score, action, rng, p.illegal_reward, p.max_exp, p.auto_reset != 0);

    if (valid) {
        p.st.boards[i] = make_uint4(bd.r[0], bd.r[1], bd.r[2], bd.r[3]);
        p.st.score[i] = score;
        store_rng(p.st.rng, p.n, i, rng);
        if (p.reward)
            p.reward[i] = r.reward;
        if (p.terminated)
            p.terminated[i] = r.terminated ? 1 : 0;
        if (p.illegal)
            p.illegal[i] = r.illegal ? 1 : 0;
        if (p.highest)
            p.highest[i] = static_cast<uint8_t>(highest(r.terminal));
    }
    r.terminated = r.terminated && valid;
    WaveAcc acc;
    record_episodes(p, i, r, acc, old_stats.max_score);
    flush_wave_stats(p, i_raw, old_stats, acc);
}

// numpy's PCG64(SeedSequence(base_seed + global board index)) for every board, computed on the device.
__global__ void __launch_bounds__(kBlock) seed_numpy_kernel(uint64_t *planes, uint32_t n, uint64_t first_seed)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n)
        return;
    const Pcg64 r = pcg64_from_seed(first_seed + i);
    planes[i] = r.state_lo;
    planes[n + i] = r.state_hi;
    planes[2ull * n + i] = r.inc_lo;
    planes[3ull * n + i] = r.inc_hi;
    planes[4ull * n + i] = r.buf;
}

__global__ void __launch_bounds__(kBlock) reset_numpy_kernel(const StepArgs p, const uint8_t *mask)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= p.n || (mask && mask[i] == 0))
        return;
    Pcg64 rng = load_rng(p.st.rng, p.n, i);
    Board bd{{0u, 0u, 0u, 0u}};   // game2048_env.py:104
    add_tile_numpy(bd, rng);      // :108
    add_tile_numpy(bd, rng);      // :109
    store_rng(p.st.rng, p.n, i, rng);
    p.st.boards[i] = make_uint4(bd.r[0], bd.r[1], bd.r[2], bd.r[3]);
    p.st.score[i] = 0;            // :105
}

__global__ void __launch_bounds__(kBlock) add_tile_numpy_kernel(const StepArgs p)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= p.n)
        return;
    const uint4 v = p.st.boards[i];
    Board bd{{v.x, v.y, v.z, v.w}};
    if (count_empty(bd) == 0)
        return;
    Pcg64 rng = load_rng(p.st.rng, p.n, i);
    add_tile_numpy(bd, rng);
    store_rng(p.st.rng, p.n, i, rng);
    p.st.boards[i] = make_uint4(bd.r[0], bd.r[1], bd.r[2], bd.r[3]);
}

// ---------------------------------------------------------------------------------- reset
// Game2048Env.reset for every (masked) board (game2048_env.py:102-111).
__global__ void __launch_bounds__(kBlock) reset_kernel(const StepArgs p, uint32_t first_slot, const uint8_t *mask)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= p.n)
        return;
    if (mask && mask[i] == 0)
        return;
    const uint32_t b = p.board_offset + i;
    const Words w = philox4x32_10(p.t_lo, p.t_hi, b, first_slot >> 2, p.seed_lo, p.seed_hi);
    const uint32_t s = first_slot & 3u;
    const uint32_t w1 = select_word(w, s);
    uint32_t w2;
    if (s == 3u) // the second spawn lives in the next block of this transaction
        w2 = philox4x32_10(p.t_lo, p.t_hi, b, (first_slot >> 2) + 1u, p.seed_lo, p.seed_hi).w[0];
    else
        w2 = select_word(w, s + 1u);
    const Board bd = fresh_board(w1, w2);
    p.st.boards[i] = make_uint4(bd.r[0], bd.r[1], bd.r[2], bd.r[3]);
    p.st.score[i] = 0; // :105
}

// ------------------------------------------------------------------------- game primitives
// Game2048Env.move alone (game2048_env.py:194-241): no spawn, no score, no clock.
template <int ACT>
__global__ void __launch_bounds__(kBlock) move_kernel(uint4 *boards, uint32_t n, const void *actions, uint32_t trial,
                                                      int32_t *score_out, uint8_t *legal_out)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n)
        return;
    const uint4 v = boards[i];
    Board bd{{v.x, v.y, v.z, v.w}};
    uint32_t gain;
    const bool legal = move(bd, load_action<ACT>(actions, i, 0u), gain);
    if (score_out)
        score_out[i] = legal ? static_cast<int32_t>(gain) : 0;
    if (legal_out)
        legal_out[i] = legal ? 1 : 0;
    if (!trial && legal)
        boards[i] = make_uint4(bd.r[0], bd.r[1], bd.r[2], bd.r[3]);
}

// Game2048Env.isend (game2048_env.py:262-280) and highest (:190-192).
__global__ void __launch_bounds__(kBlock) query_kernel(const uint4 *boards, uint32_t n, uint32_t max_exp,
                                                       uint8_t *isend_out, uint8_t *highest_out)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n)
        return;
    const uint4 v = boards[i];
    const Board bd{{v.x, v.y, v.z, v.w}};
    if (isend_out)
        isend_out[i] = is_end(bd, max_exp) ? 1 : 0;
    if (highest_out)
        highest_out[i] = static_cast<uint8_t>(highest(bd));
}

// Game2048Env.add_tile (game2048_env.py:166-176) from spawn slot `slot` of transaction t.
__global__ void __launch_bounds__(kBlock) add_tile_kernel(const StepArgs p, uint32_t slot)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= p.n)
        return;
    const uint4 v = p.st.boards[i];
    Board bd{{v.x, v.y, v.z, v.w}};
    if (count_empty(bd) == 0) // the reference asserts here (:176)
        return;
    const Words w = philox4x32_10(p.t_lo, p.t_hi, p.board_offset + i, slot >> 2, p.seed_lo, p.seed_hi);
    add_tile(bd, select_word(w, slot & 3u));
    p.st.boards[i] = make_uint4(bd.r[0], bd.r[1], bd.r[2], bd.r[3]);
}

// ------------------------------------------------------------------------ synthetic policy
__global__ void __launch_bounds__(kBlock) fill_actions_kernel(uint8_t *out, uint32_t n, uint32_t board_offset,
                                                              uint32_t seed_lo, uint32_t seed_hi, uint64_t t_first,
                                                              uint32_t k_steps)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t j = blockIdx.y;
    if (i >= n || j >= k_steps)
        return;
    const uint64_t t = t_first + j;
    const Words w = philox4x32_10(static_cast<uint32_t>(t), static_cast<uint32_t>(t >> 32), board_offset + i, 0u,
                                  seed_lo, seed_hi);
    out[static_cast<size_t>(j) * n + i] = static_cast<uint8_t>(w.w[3] >> 30);
}

// ---------------------------------------------------------------------------------- onehot
// stack() (game2048_env.py:17-32): board -> (16,4,4), channel c = (exponent == c).  One lane writes
// one 16-byte chunk of the output, so every store is a coalesced dwordx4:
//   u8 : chunk = one channel (16 cells)            16 chunks / board
//   f16: chunk = half a channel (8 cells)          32 chunks / board
//   f32: chunk = one row of one channel (4 cells)  64 chunks / board
template <int OBS>
__global__ void __launch_bounds__(kBlock) onehot_kernel(const uint4 *__restrict__ boards, uint64_t chunks,
                                                        uint4 *__restrict__ out)
{
    const uint64_t g = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (g >= chunks)
        return;
    const uint32_t *cells = reinterpret_cast<const uint32_t *>(boards);
    if constexpr (OBS == 0) {
        const uint64_t board = g >> 4;
        const uint32_t splat = static_cast<uint32_t>(g & 15u) * 0x01010101u;
        const uint4 v = boards[board];
        out[g] = make_uint4(z80(v.x ^ splat) >> 7, z80(v.y ^ splat) >> 7, z80(v.z ^ splat) >> 7,
                            z80(v.w ^ splat) >> 7);
    } else if constexpr (OBS == 1) {
        const uint64_t board = g >> 5;
        const uint32_t c = static_cast<uint32_t>(g >> 1) & 15u, half = static_cast<uint32_t>(g) & 1u;
        const uint32_t r0 = cells[board * 4 + half * 2], r1 = cells[board * 4 + half * 2 + 1];
        uint32_t h[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            h[k] = (((r0 >> (8 * k)) & 0xffu) == c) ? 0x3c00u : 0u; // fp16 1.0
            h[4 + k] = (((r1 >> (8 * k)) & 0xffu) == c) ? 0x3c00u : 0u;
        }
        out[g] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    } else {
        const uint64_t board = g >> 6;
        const uint32_t c = static_cast<uint32_t>(g >> 2) & 15u, row = static_cast<uint32_t>(g) & 3u;
        const uint32_t r = cells[board * 4 + row];
        uint32_t f[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            f[k] = (((r >> (8 * k)) & 0xffu) == c) ? 0x3f800000u : 0u; // fp32 1.0
        out[g] = make_uint4(f[0], f[1], f[2], f[3]);
    }
}

// ---------------------------------------------------------------------------- augmentation
// training_data.augment() (training_data.py:257-299) for board pairs on the device: the eight
// symmetries [orig, hflip, rot1(orig), rot1(hflip), rot2(..), rot2(..), rot3(..), rot3(..)] with the
// action remapped (hflip swaps 1 <-> 3, :262-267; a clockwise quarter turn adds 1 mod 4, :276).
// One lane per (variant, transition); every access is a coalesced 16-byte load/store.
__device__ __forceinline__ Board hflip_board(const Board &b) // np.flip(x, 2): reverse every row
{
    return Board{{__builtin_bswap32(b.r[0]), __builtin_bswap32(b.r[1]), __builtin_bswap32(b.r[2]), __builtin_bswap32(b.r[3])}};
}

__device__ __forceinline__ Board rotate_board(const Board &b, uint32_t k) // np.rot90(x, k, axes=(2, 1)): clockwise
{
    const Board t = transpose(b);
    if (k == 1u) // out[r][c] = in[3-c][r]
        return Board{{__builtin_bswap32(t.r[0]), __builtin_bswap32(t.r[1]), __builtin_bswap32(t.r[2]), __builtin_bswap32(t.r[3])}};
    if (k == 2u) // out[r][c] = in[3-r][3-c]
        return Board{{__builtin_bswap32(b.r[3]), __builtin_bswap32(b.r[2]), __builtin_bswap32(b.r[1]), __builtin_bswap32(b.r[0])}};
    if (k == 3u) // out[r][c] = in[c][3-r]
        return Board{{t.r[3], t.r[2], t.r[1], t.r[0]}};
    return b;
}

__global__ void __launch_bounds__(kBlock) augment_kernel(const uint4 *__restrict__ boards, const uint4 *__restrict__ next_boards,
                                                         const uint8_t *__restrict__ actions, uint32_t n,
                                                         uint4 *__restrict__ boards_out, uint4 *__restrict__ next_out,
                                                         uint8_t *__restrict__ actions_out)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t variant = blockIdx.y; // 0..7
    if (i >= n)
        return;
    const uint32_t flip = variant & 1u, k = variant >> 1;
    uint4 v = boards[i];
    Board b{{v.x, v.y, v.z, v.w}};
    uint32_t a = actions[i];
    if (flip) {
        b = hflip_board(b);
        a = (a == 1u) ? 3u : (a == 3u ? 1u : a);
    }
    b = rotate_board(b, k);
    a = (a + k) & 3u;
    const size_t o = static_cast<size_t>(variant) * n + i;
    boards_out[o] = make_uint4(b.r[0], b.r[1], b.r[2], b.r[3]);
    actions_out[o] = static_cast<uint8_t>(a);
    if (next_boards) {
        v = next_boards[i];
        Board nb{{v.x, v.y, v.z, v.w}};
        if (flip)
            nb = hflip_board(nb);
        nb = rotate_board(nb, k);
        next_out[o] = make_uint4(nb.r[0], nb.r[1], nb.r[2], nb.r[3]);
    }
}

// ----------------------------------------------------------------------------------- stats
// Reduce the per-wave accumulators (and the highest tile on any board) to one StatsOut.
// Block-level tree in LDS first, then ONE set of atomics per block (a single hot word serialises at
// ~88 atomics/us on this chip, so per-wave atomics to one address would take hundreds of us).
__global__ void __launch_bounds__(kBlock) stats_kernel(const DeviceState st, uint32_t n, uint32_t n_waves,
                                                       StatsOut *out)
{
    __shared__ unsigned long long s_ep[kBlock], s_ill[kBlock], s_sum[kBlock];
    __shared__ int s_max[kBlock], s_exp[kBlock];
    unsigned long long episodes = 0, illegal = 0, score_sum = 0;
    int max_score = 0, max_exp = 0;
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t wv = blockIdx.x * kBlock + threadIdx.x; wv < n_waves; wv += stride) {
        const WaveStats ws = st.wave_stats[wv];
        episodes += ws.episodes;
        illegal += ws.illegal_ends;
        score_sum += ws.score_sum;
        max_score = max(max_score, ws.max_score);
    }
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const uint4 v = st.boards[i];
        max_exp = max(max_exp, static_cast<int>(highest(Board{{v.x, v.y, v.z, v.w}})));
    }
    const uint32_t tid = threadIdx.x;
    s_ep[tid] = episodes; s_ill[tid] = illegal; s_sum[tid] = score_sum; s_max[tid] = max_score; s_exp[tid] = max_exp;
    __syncthreads();
    for (uint32_t off = kBlock / 2; off > 0; off >>= 1) {
        if (tid < off) {
            s_ep[tid] += s_ep[tid + off];
            s_ill[tid] += s_ill[tid + off];
            s_sum[tid] += s_sum[tid + off];
            s_max[tid] = max(s_max[tid], s_max[tid + off]);
            s_exp[tid] = max(s_exp[tid], s_exp[tid + off]);
        }
        __syncthreads();
    }
    if (tid == 0) {
        atomicAdd(&out->episodes, s_ep[0]);
        atomicAdd(&out->illegal_ends, s_ill[0]);
        atomicAdd(&out->score_sum, s_sum[0]);
        atomicMax(&out->max_score, s_max[0]);
        atomicMax(&out->max_exp, static_cast<unsigned int>(s_exp[0]));
    }
}

// -------------------------------------------------------------------------------- launchers
static inline dim3 grid_for(uint64_t items) { return dim3(static_cast<uint32_t>((items + kBlock - 1) / kBlock)); }

hipError_t launch_reset(const StepArgs &a, uint32_t first_slot, const uint8_t *mask, hipStream_t s)
{
    if (a.n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(reset_kernel, grid_for(a.n), dim3(kBlock), 0, s, a, first_slot, mask);
    return hipGetLastError();
}

hipError_t launch_step(const StepArgs &a, int action_dtype, hipStream_t s)
{
    if (a.n == 0)
        return hipSuccess;
    const dim3 g = grid_for(a.n), b(kBlock);
    switch (action_dtype) {
    case 0: hipLaunchKernelGGL(step_kernel<0>, g, b, 0, s, a); break;
    case 1: hipLaunchKernelGGL(step_kernel<1>, g, b, 0, s, a); break;
    case 2: hipLaunchKernelGGL(step_kernel<2>, g, b, 0, s, a); break;
    case 3: hipLaunchKernelGGL(step_kernel<3>, g, b, 0, s, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_move(uint4 *boards, uint32_t n, const void *actions, int action_dtype, bool trial,
                       int32_t *score_out, uint8_t *legal_out, hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    const dim3 g = grid_for(n), b(kBlock);
    const uint32_t tr = trial ? 1u : 0u;
    switch (action_dtype) {
    case 1: hipLaunchKernelGGL(move_kernel<1>, g, b, 0, s, boards, n, actions, tr, score_out, legal_out); break;
    case 2: hipLaunchKernelGGL(move_kernel<2>, g, b, 0, s, boards, n, actions, tr, score_out, legal_out); break;
    case 3: hipLaunchKernelGGL(move_kernel<3>, g, b, 0, s, boards, n, actions, tr, score_out, legal_out); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_query(const uint4 *boards, uint32_t n, uint32_t max_exp, uint8_t *isend_out, uint8_t *highest_out,
                        hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(query_kernel, grid_for(n), dim3(kBlock), 0, s, boards, n, max_exp, isend_out, highest_out);
    return hipGetLastError();
}

hipError_t launch_add_tile(const StepArgs &a, uint32_t slot, hipStream_t s)
{
    if (a.n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(add_tile_kernel, grid_for(a.n), dim3(kBlock), 0, s, a, slot);
    return hipGetLastError();
}

hipError_t launch_seed_numpy(uint64_t *planes, uint32_t n, uint64_t first_seed, hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(seed_numpy_kernel, grid_for(n), dim3(kBlock), 0, s, planes, n, first_seed);
    return hipGetLastError();
}

hipError_t launch_reset_numpy(const StepArgs &a, const uint8_t *mask, hipStream_t s)
{
    if (a.n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(reset_numpy_kernel, grid_for(a.n), dim3(kBlock), 0, s, a, mask);
    return hipGetLastError();
}

hipError_t launch_step_numpy(const StepArgs &a, int action_dtype, hipStream_t s)
{
    if (a.n == 0)
        return hipSuccess;
    const dim3 g = grid_for(a.n), b(kBlock);
    switch (action_dtype) {
    case 0: hipLaunchKernelGGL(step_numpy_kernel<0>, g, b, 0, s, a); break;
    case 1: hipLaunchKernelGGL(step_numpy_kernel<1>, g, b, 0, s, a); break;
    case 2: hipLaunchKernelGGL(step_numpy_kernel<2>, g, b, 0, s, a); break;
    case 3: hipLaunchKernelGGL(step_numpy_kernel<3>, g, b, 0, s, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_add_tile_numpy(const StepArgs &a, hipStream_t s)
{
    if (a.n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(add_tile_numpy_kernel, grid_for(a.n), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_rollout_fused(const StepArgs &a, int action_dtype, uint64_t stride, hipStream_t s)
{
    if (a.n == 0 || a.k_steps == 0)
        return hipSuccess;
    const dim3 g = grid_for(a.n), b(kBlock);
    switch (action_dtype) {
    case 0: hipLaunchKernelGGL(rollout_fused_kernel<0>, g, b, 0, s, a, stride); break;
    case 1: hipLaunchKernelGGL(rollout_fused_kernel<1>, g, b, 0, s, a, stride); break;
    case 2: hipLaunchKernelGGL(rollout_fused_kernel<2>, g, b, 0, s, a, stride); break;
    case 3: hipLaunchKernelGGL(rollout_fused_kernel<3>, g, b, 0, s, a, stride); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_rollout_random(const StepArgs &a, hipStream_t s)
{
    if (a.n == 0 || a.k_steps == 0)
        return hipSuccess;
    hipLaunchKernelGGL(rollout_random_kernel, grid_for(a.n), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_fill_actions(uint8_t *out, uint32_t n, uint32_t board_offset, uint32_t seed_lo, uint32_t seed_hi,
                               uint64_t t_first, uint32_t k_steps, hipStream_t s)
{
    if (n == 0 || k_steps == 0)
        return hipSuccess;
    // gridDim.y is limited to 65535: walk the step axis in slabs.
    for (uint32_t j0 = 0; j0 < k_steps; j0 += 32768u) {
        const uint32_t kj = (k_steps - j0 < 32768u) ? (k_steps - j0) : 32768u;
        dim3 g(grid_for(n).x, kj);
        hipLaunchKernelGGL(fill_actions_kernel, g, dim3(kBlock), 0, s, out + static_cast<size_t>(j0) * n, n,
                           board_offset, seed_lo, seed_hi, t_first + j0, kj);
    }
    return hipGetLastError();
}

hipError_t launch_onehot(const uint4 *boards, uint32_t n, void *out, int obs_dtype, hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    uint4 *o = static_cast<uint4 *>(out);
    switch (obs_dtype) {
    case 0: {
        const uint64_t chunks = static_cast<uint64_t>(n) * 16;
        hipLaunchKernelGGL(onehot_kernel<0>, grid_for(chunks), dim3(kBlock), 0, s, boards, chunks, o);
        break;
    }
    case 1: {
        const uint64_t chunks = static_cast<uint64_t>(n) * 32;
        hipLaunchKernelGGL(onehot_kernel<1>, grid_for(chunks), dim3(kBlock), 0, s, boards, chunks, o);
        break;
    }
    case 2: {
        const uint64_t chunks = static_cast<uint64_t>(n) * 64;
        hipLaunchKernelGGL(onehot_kernel<2>, grid_for(chunks), dim3(kBlock), 0, s, boards, chunks, o);
        break;
    }
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_augment(const uint4 *boards, const uint4 *next_boards, const uint8_t *actions, uint32_t n,
                          uint4 *boards_out, uint4 *next_out, uint8_t *actions_out, hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(augment_kernel, dim3(grid_for(n).x, 8), dim3(kBlock), 0, s, boards, next_boards, actions, n,
                       boards_out, next_out, actions_out);
    return hipGetLastError();
}

hipError_t launch_stats(const DeviceState &st, uint32_t n, StatsOut *dev_out, hipStream_t s)
{
    hipError_t err = hipMemsetAsync(dev_out, 0, sizeof(StatsOut), s);
    if (err != hipSuccess || n == 0)
        return err;
    uint32_t blocks = grid_for(n).x;
    if (blocks > 512u)
        blocks = 512u;
    hipLaunchKernelGGL(stats_kernel, dim3(blocks), dim3(kBlock), 0, s, st, n, (n + 63u) / 64u, dev_out);
    return hipGetLastError();
}

} // namespace g2048
