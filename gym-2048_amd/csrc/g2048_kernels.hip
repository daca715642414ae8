// g2048_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the batched 2048 environment.
//
// Mapping: ONE BOARD PER LANE.  A wavefront owns 64 consecutive boards = 1 KiB of board records, read
// and written with one global_load/store_dwordx4 per lane (fully coalesced, 16 B/lane).  The whole
// step -- slide/merge sweep, score, spawn, done detection -- runs out of VGPRs with byte-parallel
// integer ops (g2048_device.h); there is no per-board RNG state: the spawn randomness of
// (transaction t, board b) is one Philox4x32-10 block.
//
// HBM bytes per env-step of step_kernel (the roofline figure in DESIGN.md):
//   algorithmic 38 B = board in 16 + action 1 + board out 16 + reward 4 + terminated 1
// and that is also all a step moves for a board whose episode goes on: the episodic score travels
// inside the 16-byte record (g2048_device.h "board RECORD").  A board whose episode ENDS additionally
// writes its terminal record to last_record (sparse 16-byte store; the episode's return is computed
// from it when somebody asks) unless the engine was told not to keep them (g2048_set_last_records: 0.85 us
// of a 10.2 us launch at 2^20 boards, tools/ubench/r4_probe.hip), and every wavefront adds its counts and the
// merge scores of its 64 moves to its 32-byte slot (12 bytes stored per launch).
#include "g2048_kernels.h"

#include "g2048_device.h"
#include "g2048_pcg64.h"

#include <atomic>
#include <type_traits>


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

// one 16-byte piece of an observation: written once and read by somebody else's kernel, so a streaming (nt) store
__device__ __forceinline__ void store_chunk_nt(uint4 *out, uint64_t g, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    const u32x4 v = {a, b, c, d};
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(out) + g);
}

__device__ __forceinline__ Board load_board(const uint4 *boards, uint32_t i)
{
    const uint4 v = boards[i];
    return Board{{v.x, v.y, v.z, v.w}};
}

__device__ __forceinline__ void store_board(uint4 *boards, uint32_t i, const Board &b)
{
    boards[i] = make_uint4(b.r[0], b.r[1], b.r[2], b.r[3]);
}

// Per-WAVEFRONT lookup tables in LDS (512 B): the per-action selector rows of move_sel and the 16
// one-tile boards of fresh_record_lut (g2048_device.h).  Each wavefront stages its own copy -- every lane
// fetches ONE 8-byte piece of the 512-byte image below (global load, L2-resident) and writes it to LDS at
// byte offset threadIdx.x * 8 -- and only reads what it wrote itself, so no workgroup barrier is involved.
struct WaveTables {
    uint32_t move[32]; // 4 actions x 8 words (6 used)
    uint32_t cell[64]; // 16 cells x 4 words: entry p = the board whose only tile is a 2 (exponent 1) in cell p
    uint32_t pad[32];  // 64 lanes x 8 bytes
};

// word j of entry p of the one-tile table (= onehot_cell_word(p, j) of g2048_device.h, as a constant expression)
#define G2048_CELL_ENTRY(p) ((p) / 4 == 0 ? 1u << (8 * ((p) % 4)) : 0u), ((p) / 4 == 1 ? 1u << (8 * ((p) % 4)) : 0u), \
                            ((p) / 4 == 2 ? 1u << (8 * ((p) % 4)) : 0u), ((p) / 4 == 3 ? 1u << (8 * ((p) % 4)) : 0u)
__device__ const uint32_t kWaveTablesImage[128] __attribute__((aligned(16))) = {
    G2048_MOVE_LUT_WORDS,
    G2048_CELL_ENTRY(0), G2048_CELL_ENTRY(1), G2048_CELL_ENTRY(2), G2048_CELL_ENTRY(3),
    G2048_CELL_ENTRY(4), G2048_CELL_ENTRY(5), G2048_CELL_ENTRY(6), G2048_CELL_ENTRY(7),
    G2048_CELL_ENTRY(8), G2048_CELL_ENTRY(9), G2048_CELL_ENTRY(10), G2048_CELL_ENTRY(11),
    G2048_CELL_ENTRY(12), G2048_CELL_ENTRY(13), G2048_CELL_ENTRY(14), G2048_CELL_ENTRY(15),
    /* pad[32] = 0 */
};
#undef G2048_CELL_ENTRY

// `x`, but not before `dep` has been computed (pins the s_waitcnt of a load below independent work).
__device__ __forceinline__ uint2 use_after(uint2 x, uint32_t dep)
{
    unsigned long long both = (static_cast<unsigned long long>(x.y) << 32) | x.x; // ONE 64-bit operand: the pair stays a pair
    asm volatile("" : "+v"(both) : "v"(dep));
    return make_uint2(static_cast<uint32_t>(both), static_cast<uint32_t>(both >> 32));
}

struct LdsTables {
    const WaveTables *t;
    __device__ __forceinline__ MoveSel move_sel(uint32_t action) const
    {
        const uint4 a = *reinterpret_cast<const uint4 *>(t->move + action * 8u);
        const uint2 b = *reinterpret_cast<const uint2 *>(t->move + action * 8u + 4u);
        return MoveSel{a.x, a.y, a.z, a.w, b.x, b.y};
    }
    __device__ __forceinline__ Board onehot_cell(uint32_t p) const
    {
        const uint4 v = *reinterpret_cast<const uint4 *>(t->cell + p * 4u);
        return Board{{v.x, v.y, v.z, v.w}};
    }
};

// Two halves so that the global load is issued early and waited for late (after the Philox block).
__device__ __forceinline__ uint2 load_tables_piece()
{
    return reinterpret_cast<const uint2 *>(kWaveTablesImage)[threadIdx.x & 63u];
}

__device__ __forceinline__ LdsTables stage_tables(WaveTables *all, uint2 piece)
{
    // wave w's copy starts at w * 512 bytes and lane l writes bytes [8l, 8l + 8): together threadIdx.x * 8
    reinterpret_cast<uint2 *>(all)[threadIdx.x] = piece;
    // same-wave LDS writes are visible to the wave's later reads in program order; the fence only
    // keeps the compiler from moving those reads up
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return LdsTables{all + (threadIdx.x >> 6)};
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

// Strict actions (g2048_set_strict_actions; game2048_env.py:49 declares Discrete(4), :210-212 happens to play 4 as "down" and
// -1 as "right"): the same load, and whether the RAW value lies outside 0..3.  The move always uses the low two bits.
template <int ACT>
__device__ __forceinline__ uint32_t load_action_checked(const void *actions, uint32_t i, bool &bad, uint32_t &raw_low)
{
    if constexpr (ACT == 1) {
        const uint32_t v = __builtin_nontemporal_load(static_cast<const uint8_t *>(actions) + i);
        bad = v > 3u;
        raw_low = v;
        return v & 3u;
    } else if constexpr (ACT == 2) {
        const uint32_t v = static_cast<uint32_t>(__builtin_nontemporal_load(static_cast<const int32_t *>(actions) + i));
        bad = v > 3u;
        raw_low = v;
        return v & 3u;
    } else {
        const unsigned long long v = static_cast<unsigned long long>(__builtin_nontemporal_load(static_cast<const long long *>(actions) + i));
        bad = v > 3ull;
        raw_low = static_cast<uint32_t>(v);
        return static_cast<uint32_t>(v) & 3u;
    }
}

// One lane of the wavefront (the first offender) publishes {bit 63, low byte of the action, global board index} to the
// engine's pinned error word; the host finds it at the entry of its next call on the engine.  Whole wavefronts call this.
__device__ __forceinline__ void report_bad_action(unsigned long long *action_err, bool bad, uint32_t raw_low, uint32_t global_index)
{
    const unsigned long long offenders = __builtin_amdgcn_ballot_w64(bad);
    if (offenders == 0ull)
        return;
    if ((threadIdx.x & 63u) == static_cast<uint32_t>(__builtin_ctzll(offenders)))
        __hip_atomic_store(action_err, (1ull << 63) | (static_cast<unsigned long long>(raw_low & 0xffu) << 32) | global_index,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Episode bookkeeping, shared by every step kernel.  A lane whose episode ended stores the record it
// ended on (sparse 16-byte store; plain, not nt: L2 merges these into lines) and, if asked, the plain
// terminal board; the wave's two counts go to its slot (flush_episode_counts).  Skipped
// entirely by a wave-uniform branch when no lane terminated.
// Returns the lane mask of the boards whose episode ended in this step.
__device__ __forceinline__ unsigned long long record_episode_ends(const StepArgs &p, uint32_t i, bool fin, bool illegal,
                                                                  const Board &terminal, uint32_t &episodes, uint32_t &illegal_ends)
{
    const unsigned long long done = __builtin_amdgcn_ballot_w64(fin);
    if (done == 0ull)
        return 0ull;
    if (fin) {
        if (p.st.last_record) // (wave-uniform; NULL = the engine does not keep terminal records, g2048_set_last_records)
            store_board(p.st.last_record, i, terminal);
        if (p.terminal_boards)
            store_board(p.terminal_boards, i, record_cells(terminal));
    }
    episodes += static_cast<uint32_t>(__popcll(done));
    illegal_ends += static_cast<uint32_t>(__popcll(__builtin_amdgcn_ballot_w64(fin && illegal)));
    return done;
}

// ---- wave-wide sums.  gfx9 DPP: an inclusive scan inside each row of 16 lanes (row_shr 1, 2, 4, 8), then lane 15 of
// rows 0 / 2 is added to rows 1 / 3 (row_bcast:15) and lane 31 to rows 2, 3 (row_bcast:31): six v_add_u32_dpp, the
// wave's total ends up in LANE 63 (the other lanes hold partial sums).  Every lane of the wavefront must be active.
__device__ __forceinline__ uint32_t wave_sum_lane63(uint32_t v)
{
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x111, 0xf, 0xf, false)); // row_shr:1
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x112, 0xf, 0xf, false)); // row_shr:2
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x114, 0xf, 0xf, false)); // row_shr:4
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x118, 0xf, 0xf, false)); // row_shr:8
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x142, 0xa, 0xf, false)); // row_bcast:15 -> rows 1, 3
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x143, 0xc, 0xf, false)); // row_bcast:31 -> rows 2, 3
    return v;
}

// The same for 64-bit lane values (the gains a lane accumulated over a fused rollout): three 22-bit pieces, each of
// whose 64-lane sums fits 32 bits, recombined in lane 63.  Once per launch.
__device__ __forceinline__ unsigned long long wave_sum64_lane63(unsigned long long v)
{
    const uint32_t s0 = wave_sum_lane63(static_cast<uint32_t>(v) & 0x3fffffu);
    const uint32_t s1 = wave_sum_lane63(static_cast<uint32_t>(v >> 22) & 0x3fffffu);
    const uint32_t s2 = wave_sum_lane63(static_cast<uint32_t>(v >> 44));
    return static_cast<unsigned long long>(s0) + (static_cast<unsigned long long>(s1) << 22) +
           (static_cast<unsigned long long>(s2) << 44);
}

// ---- the wavefront's episode SLOT: 32 bytes per 64 boards (g2048_kernels.h kSlotWords), four 64-bit quantities
//   episodes      finished episodes
//   illegal_ends  ... of which ended on an illegal move
//   gain_sum      G: merge score of every move of these 64 boards since the statistics were cleared (+ the scores
//                 they held then, -/+ what g2048_reset / g2048_set_scores took away or put in).  Conservation:
//                 every point scored is either still on a live board or belongs to a finished episode, so
//                     sum of the final scores of ALL finished episodes  =  G - sum of the live boards' scores
//                 which the statistics kernel evaluates (g2048_stats.return_sum) -- exact, and no per-episode
//                 work in the step: the step only adds its 64 gains (six DPP adds)
//   pending       bit l: board l's episode has ended and the board has NOT been reset (auto_reset == 0): its score
//                 belongs to a finished episode although it is still in the record
// stored as EIGHT DWORDS with the four LOW halves first: {episodes.lo, illegal_ends.lo, gain_sum.lo, pending.lo,
// episodes.hi, illegal_ends.hi, gain_sum.hi, pending.hi}.  Sparse stores cost per DWORD on this chip (the 16-byte
// terminal-record store of 7 % of the lanes costs 0.57 us per launch at 2^20 boards, a 4-byte one 0.11 us,
// profiles/r02_g_ubench_2p20.txt; storing all eight dwords of the slot per launch measured +0.32 us against the two-word
// slot of round 3, tools/ubench/r4_probe.hip), so a launch stores the THREE low counter dwords with one
// global_store_dwordx3 and touches the rest only when it changes: the high halves on a carry (once in 2^32 counts), the
// pending mask when it differs from the one that came in (never with auto_reset).
// The slot is private to its wavefront (one wavefront per slot per launch, launches are stream-ordered), so it is
// updated without atomics: the old values come in through the SCALAR cache (uniform address, one s_load_dwordx8 at
// kernel entry -- no VALU, no vector-memory instruction, latency never exposed) and ONE lane -- lane 63, where the DPP
// sum of the gains lands -- stores the new ones.  Measured against 64-bit atomics: -0.5 us per launch at 2^20 boards
// (profiles/r02_g_ubench_2p20.txt).
struct EpisodeCounters {
    uint32_t *slot;
    uint32_t ep_lo, ill_lo, gain_lo, pend_lo, ep_hi, ill_hi, gain_hi, pend_hi;
};

__device__ __forceinline__ uint32_t *slot_of(unsigned long long *ep_counters, uint32_t wave_id)
{
    return reinterpret_cast<uint32_t *>(ep_counters + kSlotWords * wave_id);
}

__device__ __forceinline__ EpisodeCounters load_episode_counters(const StepArgs &p, uint32_t i_raw)
{
    uint32_t *slot = slot_of(p.st.ep_counters, __builtin_amdgcn_readfirstlane(i_raw >> 6));
    const uint4 lo = *reinterpret_cast<const uint4 *>(slot), hi = *reinterpret_cast<const uint4 *>(slot + 4);
    return EpisodeCounters{slot, lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
}

// 64-bit views of a slot for the kernels off the hot path
__device__ __forceinline__ unsigned long long slot_get(const uint32_t *slot, uint32_t k)
{
    return static_cast<unsigned long long>(slot[k]) | (static_cast<unsigned long long>(slot[4u + k]) << 32);
}

__device__ __forceinline__ void slot_set(uint32_t *slot, uint32_t k, unsigned long long v)
{
    slot[k] = static_cast<uint32_t>(v);
    slot[4u + k] = static_cast<uint32_t>(v >> 32);
}

enum : uint32_t { kSlotEpisodes = 0, kSlotIllegal = 1, kSlotGain = 2, kSlotPending = 3 };

// gain_lane63: the wave's summed merge score of this launch, valid in lane 63 (wave_sum_lane63 / wave_sum64_lane63);
// pending: the new pending mask (wave-uniform).
__device__ __forceinline__ void flush_episode_counts(const EpisodeCounters &c, uint32_t episodes, uint32_t illegal_ends,
                                                     unsigned long long gain_lane63, unsigned long long pending)
{
    if ((threadIdx.x & 63u) != 63u)
        return;
    const unsigned long long ep = (static_cast<unsigned long long>(c.ep_hi) << 32 | c.ep_lo) + episodes;      // scalar
    const unsigned long long ill = (static_cast<unsigned long long>(c.ill_hi) << 32 | c.ill_lo) + illegal_ends; // scalar
    const unsigned long long gain = (static_cast<unsigned long long>(c.gain_hi) << 32 | c.gain_lo) + gain_lane63;
    typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
    const u32x3 lo = {static_cast<uint32_t>(ep), static_cast<uint32_t>(ill), static_cast<uint32_t>(gain)};
    *reinterpret_cast<u32x3 *>(c.slot) = lo;                                           // the common case: 12 bytes
    const uint32_t ep_hi = static_cast<uint32_t>(ep >> 32), ill_hi = static_cast<uint32_t>(ill >> 32),
                   gain_hi = static_cast<uint32_t>(gain >> 32);
    if (ep_hi != c.ep_hi || ill_hi != c.ill_hi || gain_hi != c.gain_hi) {             // a carry: once in 2^32 counts
        const u32x3 hi = {ep_hi, ill_hi, gain_hi};
        *reinterpret_cast<u32x3 *>(c.slot + 4) = hi;
    }
    const uint32_t pend_lo = static_cast<uint32_t>(pending), pend_hi = static_cast<uint32_t>(pending >> 32);
    if (pend_lo != c.pend_lo || pend_hi != c.pend_hi) {                                // only without auto_reset
        c.slot[3] = pend_lo;
        c.slot[7] = pend_hi;
    }
}

// "episode ended and not reset" mask of a step launch from the mask of the lanes whose episode ended: every board was
// stepped, so older marks are gone, and with auto_reset nothing stays pending.  Scalar arithmetic, no branch.
__device__ __forceinline__ unsigned long long pending_after_step(unsigned long long done, uint32_t auto_reset)
{
    return done & (0ull - static_cast<unsigned long long>(auto_reset == 0u ? 1u : 0u));
}

// What a reset / score import does to the slot of its wavefront (not hot; whole wavefronts must call it):
//   reset of a board whose episode is still running   -> the episode is ABANDONED: its score leaves G
//   reset of a board whose episode has ended (pending) -> the mark is cleared, its score stays in G (a finished episode)
// `delta`: signed change of the board's live score caused by this call (0 for lanes that do nothing), `clear`: the lane
// clears its pending mark.
__device__ __forceinline__ void adjust_slot(unsigned long long *ep_counters, uint32_t i_raw, int32_t delta, bool clear)
{
    const uint32_t total = wave_sum_lane63(static_cast<uint32_t>(delta)); // |sum| <= 64 * 2^24: fits int32
    const unsigned long long cleared = __builtin_amdgcn_ballot_w64(clear);
    if ((threadIdx.x & 63u) == 63u) {
        uint32_t *slot = slot_of(ep_counters, i_raw >> 6);
        slot_set(slot, kSlotGain, slot_get(slot, kSlotGain) + static_cast<unsigned long long>(static_cast<long long>(static_cast<int32_t>(total))));
        slot_set(slot, kSlotPending, slot_get(slot, kSlotPending) & ~cleared);
    }
}

__device__ __forceinline__ bool is_pending(unsigned long long *ep_counters, uint32_t i_raw)
{
    const unsigned long long pending = slot_get(slot_of(ep_counters, __builtin_amdgcn_readfirstlane(i_raw >> 6)), kSlotPending);
    return ((pending >> (threadIdx.x & 63u)) & 1ull) != 0ull;
}

// ------------------------------------------------------------ observation fused into the step
// stack() (game2048_env.py:17-32) of the record a step leaves behind -- what Game2048Env.step returns first
// (game2048_env.py:100) -- written by the step kernel itself: no second launch and no re-read of the records.
//
// The 64 boards of a wavefront are consecutive, so their observations are ONE contiguous piece of the output
// (16 / 32 / 64 KiB for u8 / f16 / f32).  The wave parks its 64 records in LDS (1 KiB) and writes that piece
// with fully coalesced 16-byte streaming stores: in store s, lane l writes chunk s * 64 + l.  Per dtype a chunk is
//   u8 : one channel (16 cells) of one board      -> needs the whole record     (ds_read_b128, 4 boards per store)
//   f16: half a channel (8 cells)                 -> two words of the record    (ds_read_b64,  2 boards per store)
//   f32: one row of one channel (4 cells)         -> one word of the record     (ds_read_b32,  1 board per store)
// The byte compares (eq_ones / onehot4_f16 / onehot4_f32) live in g2048_device.h, where the host unit test reaches them.
// Occupancy of the kernels that write observations is CAPPED at 2 workgroups (8 wavefronts) per CU by launching
// them with this much unused dynamic LDS (6 KiB static + 58 KiB = 64 KiB per workgroup of the CU's 160 KiB).  The
// kernel is bound by DRAM writes, and every resident wavefront owns its own 16-64 KiB output region: with all 32
// wave slots of a CU filled the chip writes into 8 192 regions at once (128 MiB of open write window at u8) and the
// memory system runs at 5.1 TB/s; narrowing the window to 2 048 regions brings 5.9-6.1 TB/s (u8, 2^20 and 2^22
// boards: 60.5 -> 52.6 us and 239 -> 201 us; f16 103 -> 99 us; f32 is insensitive; tools/ubench/r3_probe.hip parts B
// and C, profiles/r03_d_probe_*.txt, r03_l_probe_c_*.txt; at 2^24 boards 3-4 workgroups per CU are ~3 % better
// than 2, all of them 4-9 % better than uncapped).  The arithmetic needs a sixth of the issue slots, so nothing is
// lost on the compute side.
constexpr uint32_t kObsOccupancyPad = 58u * 1024u;

// The pad plus the kernel's static LDS (the per-wave tables and the parked records) must fit the 64 KiB a workgroup may
// own; two such workgroups then fill 128 of the CU's 160 KiB and a third does not fit.
static_assert(sizeof(WaveTables) * (kBlock / 64) + sizeof(uint4) * kBlock + kObsOccupancyPad <= 64u * 1024u,
              "static LDS + occupancy pad exceed the 64 KiB workgroup limit");

// Dynamic LDS above 48 KiB is opt-in per kernel and per device.  Returns the pad to launch with: kObsOccupancyPad where
// the device granted it, 0 where it did not (a part with less LDS, a runtime that refuses) -- the kernel is then merely
// not occupancy-capped, instead of every observation-writing launch failing.
template <class Kernel>
static uint32_t obs_pad_for(Kernel kernel, std::atomic<signed char> (&state)[64])
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
        dev = 0;
    // (launches come from several threads -- two chains, LocalShards: the per-device answer is an atomic; two threads that
    //  both find it unset both ask the runtime, which is idempotent)
    signed char known = state[dev].load(std::memory_order_acquire);
    if (known == 0) {
        const hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   static_cast<int>(kObsOccupancyPad));
        if (err != hipSuccess)
            (void)hipGetLastError(); // not sticky: the launch below goes ahead without the pad
        known = err == hipSuccess ? 1 : -1;
        state[dev].store(known, std::memory_order_release);
    }
    return known > 0 ? kObsOccupancyPad : 0u;
}

template <int OBS, bool FULL>
__device__ __forceinline__ void emit_onehot_as(const Cells16 *recs, uint4 *out, uint32_t lane, uint32_t n_here)
{
    constexpr uint32_t kStores = 16u << OBS; // 16 / 32 / 64 KiB per wavefront
    constexpr uint32_t kUnroll = OBS == 2 ? 16u : kStores;
#pragma unroll kUnroll
    for (uint32_t s = 0; s < kStores; ++s) {
        uint32_t b;
        const Chunk16 c = onehot_chunk<OBS>(recs, s, lane, b);
        if (FULL || b < n_here)
            store_chunk_nt(out, s * 64u + lane, c.w[0], c.w[1], c.w[2], c.w[3]);
    }
}

template <bool FULL>
__device__ __forceinline__ void emit_onehot(uint4 *wave_recs, const Board &rec, void *obs, uint32_t obs_dtype,
                                            uint32_t wave_first, uint32_t n)
{
    const uint32_t lane = threadIdx.x & 63u;
    wave_recs[lane] = make_uint4(rec.r[0], rec.r[1], rec.r[2] & kCellBits, rec.r[3] & kCellBits);
    // same-wave LDS accesses execute in program order; the fences keep the compiler from reordering them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // boards of this wave that exist (ragged last block only)
    const uint32_t n_here = FULL ? 64u : (wave_first < n ? (n - wave_first < 64u ? n - wave_first : 64u) : 0u);
    const Cells16 *recs = reinterpret_cast<const Cells16 *>(wave_recs);
    // the wave's piece starts at board wave_first: 16 << dtype chunks of 16 bytes per board
    uint4 *out = static_cast<uint4 *>(obs) + (static_cast<uint64_t>(wave_first) << (4u + obs_dtype));
    if (obs_dtype == 0u)
        emit_onehot_as<0, FULL>(recs, out, lane, n_here);
    else if (obs_dtype == 1u)
        emit_onehot_as<1, FULL>(recs, out, lane, n_here);
    else
        emit_onehot_as<2, FULL>(recs, out, lane, n_here);
}

// ------------------------------------------------------------------- host-visible completion
// g2048_step_host / g2048_fetch_host: the outputs of a launch live in mapped, coherent pinned HOST memory and the host
// polls one word instead of calling into the runtime (hipStreamSynchronize costs ~5 us more per call than a poll,
// tools/ubench/host_latency.hip).  A ONE-block launch publishes the word itself: every wave makes its stores visible
// at system scope, the block meets, one lane stores the value with release semantics.  Larger launches are followed
// by signal_kernel on the same stream (the kernel boundary orders it behind the outputs).
__device__ __forceinline__ void signal_done(unsigned long long *done_seq, unsigned long long value)
{
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0 && gridDim.x == 1)
        __hip_atomic_store(done_seq, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void __launch_bounds__(64) signal_kernel(unsigned long long *done_seq, unsigned long long value)
{
    if (threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(done_seq, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ------------------------------------------------------------------- stream-to-stream ordering by flags
// The two chains of a rollout (g2048_set_chains) are ordered against the caller's stream with a pair of one-wave kernels
// instead of HIP events: flag_set_kernel publishes a ticket in device memory (agent-scope release: it runs behind the
// work it stands for, in stream order), flag_wait_kernel on the other stream spins until the ticket is there (agent-scope
// acquire) and the kernels behind it follow in stream order.  An event record + hipStreamWaitEvent pair costs this
// runtime ~15 us of latency per crossing; the flag pair a few us (tools/chain_fixed_cost.py).  The wait is BOUNDED
// (2^26 polls of ~1 us, one to two minutes -- longer than any work a caller could reasonably have queued ahead of the
// ticket): a ticket that never comes must not hang the device for good.  A wait that runs out is LOUD: it publishes the
// ticket it gave up on in `timed_out` -- a word of pinned, coherent host memory the library checks at the entry of every
// call on the engine, which then fails with G2048_ERR_HIP -- because the kernels behind it run unordered against the
// other chain from there on.
__global__ void __launch_bounds__(64) flag_set_kernel(unsigned long long *flag, unsigned long long value)
{
    if (threadIdx.x == 0)
        __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void __launch_bounds__(64) flag_wait_kernel(const unsigned long long *flag, unsigned long long value,
                                                       unsigned long long *timed_out, uint32_t max_polls)
{
    if (threadIdx.x != 0)
        return;
    for (uint32_t spin = 0; spin < max_polls; ++spin) {
        if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= value)
            return;
        __builtin_amdgcn_s_sleep(16); // ~1 us between polls
    }
    if (timed_out)
        __hip_atomic_store(timed_out, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---------------------------------------------------------------------------------- step
// Game2048Env.step for every board (game2048_env.py:76-100), one launch per environment step.
//
// One board per lane, one pass: load the 16-byte record + the action -> Philox block (independent of
// the loads, so it runs while they are in flight) -> play_record (move through the lane's selector row,
// spawn, done detection, deficit update, auto-reset through the one-tile table) -> store record, reward,
// terminated.  No barrier, no cross-wave traffic.
// FULL: the batch is a whole number of blocks (n % 256 == 0), no lane is past the end.
//
// Kernel arguments are FLAT and ordered by need: the library is built with -mllvm -amdgpu-kernarg-preload-count=16,
// so the command processor hands the first 14 dwords (what the loads and the Philox block need) to every
// wavefront in SGPRs -- the board load is addressed without an s_load + wait on the kernarg segment.  The rest
// (outputs, written last) comes in as one by-reference struct.  -0.25 us per launch at 2^20 boards against the
// single-struct signature (profiles/r02_v_ubench_2p20.txt).
struct StepTail {
    uint8_t *terminated;
    uint4 *last_record;
    uint8_t *illegal;
    uint8_t *highest;
    uint4 *terminal_boards;
    float illegal_reward;
    uint32_t max_exp;
    uint32_t auto_reset;
    void *obs;          // HAS_OBS kernels only
    uint32_t obs_dtype;
    uint4 *boards_out;  // !STD kernels only
    unsigned long long *done_seq;
    unsigned long long done_value;
    unsigned long long *action_err; // !STD kernels only: strict actions (NULL = not checked)
};

// STD: the standard configuration -- reward and terminated present, no illegal / highest / terminal_boards, no
// max_tile -- so none of the "is this wanted" branches exists (a taken scalar branch costs a wavefront ~20 cycles).
// HAS_OBS: the step also returns its observation (game2048_env.py:100 `return stack(self.Matrix), ...`): the
// one-hot of the record it leaves behind is written by emit_onehot (+256 / 512 / 1 024 B per board; the dtype is a
// wave-uniform switch, one taken branch against hundreds of store-bound instructions).
template <int ACT, bool FULL, bool STD, bool HAS_OBS>
__device__ __forceinline__ void step_body(WaveTables *s_tables, uint4 *s_recs, uint4 *boards, const void *actions,
                                          unsigned long long *ep_counters, uint32_t board_offset, uint32_t seed_lo, uint32_t seed_hi,
                                          uint32_t t_lo, uint32_t t_hi, uint32_t n, float *reward, const StepTail &tail)
{
    StepArgs p{};
    p.st.boards = boards;
    p.st.last_record = tail.last_record;
    p.st.ep_counters = ep_counters;
    p.actions = actions;
    p.reward = reward;
    p.terminated = tail.terminated;
    p.illegal = STD ? nullptr : tail.illegal;
    p.highest = STD ? nullptr : tail.highest;
    p.terminal_boards = STD ? nullptr : tail.terminal_boards;
    p.n = n;
    p.board_offset = board_offset;
    p.seed_lo = seed_lo;
    p.seed_hi = seed_hi;
    p.t_lo = t_lo;
    p.t_hi = t_hi;
    p.illegal_reward = tail.illegal_reward;
    p.max_exp = tail.max_exp;
    p.auto_reset = tail.auto_reset;
    // Lanes past the end stay active (whole wavefronts for the ballots): they recompute board n-1
    // and write nothing.
    const uint32_t i_raw = blockIdx.x * kBlock + threadIdx.x;
    const bool valid = FULL || i_raw < p.n;
    const uint32_t i = valid ? i_raw : p.n - 1u;
    Board rec = load_board_nt(p.st.boards, i);
    const uint2 tables_piece = load_tables_piece();
    const EpisodeCounters counters = load_episode_counters(p, i_raw);

    const Words w = philox4x32_10(p.t_lo, p.t_hi, p.board_offset + i, 0u, p.seed_lo, p.seed_hi);
    uint32_t action, raw_action = 0;
    bool bad_action = false;
    if constexpr (!STD && ACT != 0)
        action = load_action_checked<ACT>(p.actions, i, bad_action, raw_action);
    else
        action = load_action<ACT>(p.actions, i, w.w[3]);
    const LdsTables tb = stage_tables(s_tables, use_after(tables_piece, w.w[0]));

    const StepOut o = play_record(rec, action, w, STD ? 0u : p.max_exp, tb);

    // episode ends first: the terminal record goes out before the reset overwrites it in place
    uint32_t episodes = 0, illegal_ends = 0;
    const unsigned long long done = record_episode_ends(p, i, o.terminated && valid, !o.legal, rec, episodes, illegal_ends);
    // :86 self.score += score -- the wave's 64 gains go to its slot (G of the return accounting; an illegal move gains 0)
    const uint32_t wave_gain = wave_sum_lane63((FULL || valid) ? o.gain : 0u);
    const unsigned long long pending = pending_after_step(done, p.auto_reset);
    uint32_t top = 0;
    if (p.highest)
        top = highest(record_cells(rec));                                                        // :97
    if (o.terminated && p.auto_reset != 0)
        reset_record(rec, o, w, tb);

    if (valid) {
        store_board_nt(p.st.boards, i, rec);
        if (STD || p.reward)                                       // :90 / :95
            __builtin_nontemporal_store(o.legal ? static_cast<float>(o.gain) : p.illegal_reward, p.reward + i);
        if (STD || p.terminated)
            __builtin_nontemporal_store(static_cast<uint8_t>(o.terminated ? 1 : 0), p.terminated + i);
        if (p.illegal)
            __builtin_nontemporal_store(static_cast<uint8_t>(o.legal ? 0 : 1), p.illegal + i);
        if (p.highest)
            __builtin_nontemporal_store(static_cast<uint8_t>(top), p.highest + i);
        if (!STD && tail.boards_out)
            store_board(tail.boards_out, i, record_cells(rec));
    }
    flush_episode_counts(counters, episodes, illegal_ends, wave_gain, pending);
    if constexpr (!STD && ACT != 0) {
        if (tail.action_err) // (wave-uniform)
            report_bad_action(tail.action_err, bad_action && valid, raw_action, p.board_offset + i);
    }
    if constexpr (HAS_OBS)
        emit_onehot<FULL>(s_recs + (threadIdx.x & ~63u), rec, tail.obs, tail.obs_dtype, i_raw & ~63u, n);
    if (!STD && tail.done_seq)
        signal_done(tail.done_seq, tail.done_value);
}

template <int ACT, bool FULL, bool STD, bool HAS_OBS>
__global__ void __launch_bounds__(kBlock)
step_kernel(uint4 *boards, const void *actions, unsigned long long *ep_counters, uint32_t board_offset, uint32_t seed_lo,
            uint32_t seed_hi, uint32_t t_lo, uint32_t t_hi, uint32_t n, float *reward, const StepTail tail)
{
    __shared__ WaveTables s_tables[kBlock / 64];
    __shared__ uint4 s_recs[HAS_OBS ? kBlock : 1];
    step_body<ACT, FULL, STD, HAS_OBS>(s_tables, s_recs, boards, actions, ep_counters, board_offset, seed_lo, seed_hi, t_lo, t_hi, n,
                                       reward, tail);
}

// The same step with the transaction counter read from DEVICE MEMORY: t = *t_ptr + j.  This is the node of a CACHED
// hipGraph of a k-step launch train (launch_rollout_graph below): the graph's kernel arguments are frozen when it is
// built, so everything that differs between two rollouts over the same buffers -- only t -- comes through memory, written
// by a one-lane set_clock_kernel launch in front of the graph.  Standard configuration only (reward + terminated, no
// optional output).  The s_load of t (uniform address, scalar cache) is in flight together with the board load.
__global__ void __launch_bounds__(64) set_clock_kernel(unsigned long long *t_ptr, unsigned long long value)
{
    if (threadIdx.x == 0)
        *t_ptr = value;
}

template <int ACT, bool FULL>
__global__ void __launch_bounds__(kBlock)
step_graph_kernel(uint4 *boards, const void *actions, unsigned long long *ep_counters, uint32_t board_offset, uint32_t seed_lo,
                  uint32_t seed_hi, const unsigned long long *t_ptr, uint32_t n, uint32_t j, float *reward, const StepTail tail)
{
    __shared__ WaveTables s_tables[kBlock / 64];
    const unsigned long long t = *t_ptr + j;
    step_body<ACT, FULL, true, false>(s_tables, nullptr, boards, actions, ep_counters, board_offset, seed_lo, seed_hi,
                                      static_cast<uint32_t>(t), static_cast<uint32_t>(t >> 32), n, reward, tail);
}

// ------------------------------------------------------------------------- fused rollout
// k steps of the synthetic random policy in ONE launch; the record never leaves registers.
constexpr uint32_t kGainFold = 1024;
__global__ void __launch_bounds__(kBlock) rollout_random_kernel(const StepArgs p)
{
    __shared__ WaveTables s_tables[kBlock / 64];
    const uint32_t i_raw = blockIdx.x * kBlock + threadIdx.x;
    const bool valid = i_raw < p.n;
    const uint32_t i = valid ? i_raw : p.n - 1u;
    Board rec = load_board(p.st.boards, i);
    const LdsTables tb = stage_tables(s_tables, load_tables_piece());
    const EpisodeCounters counters = load_episode_counters(p, i_raw);
    uint64_t t = (static_cast<uint64_t>(p.t_hi) << 32) | p.t_lo; // transaction of the first step
    uint32_t episodes = 0, illegal_ends = 0;
    // this lane's merge scores over the k steps (reduced over the wave once, at the end): one 32-bit add per step, folded
    // into 64 bits every kGainFold steps (a move gains less than the sum of the tiles, < 2^22: 1 024 of them fit 32 bits)
    unsigned long long gained = 0;
    for (uint32_t j0 = 0; j0 < p.k_steps; j0 += kGainFold) {
        const uint32_t j1 = p.k_steps - j0 < kGainFold ? p.k_steps : j0 + kGainFold;
        uint32_t gained32 = 0;
        for (uint32_t j = j0; j < j1; ++j, ++t) {
            const Words w = philox4x32_10(static_cast<uint32_t>(t), static_cast<uint32_t>(t >> 32), p.board_offset + i, 0u,
                                          p.seed_lo, p.seed_hi);
            const StepOut o = play_record(rec, w.w[3] >> 30, w, p.max_exp, tb);
            gained32 += o.gain;
            record_episode_ends(p, i, o.terminated && valid, !o.legal, rec, episodes, illegal_ends);
            if (o.terminated)
                reset_record(rec, o, w, tb);
        }
        gained += gained32;
    }
    if (valid)
        store_board(p.st.boards, i, rec);
    flush_episode_counts(counters, episodes, illegal_ends, wave_sum64_lane63(valid ? gained : 0ull), 0ull);
}

// ---------------------------------------------------------------- fused rollout with per-step I/O
// The same k steps g2048_rollout performs with k launches, in ONE launch: the record stays in
// registers, each step reads action[j][i] and writes reward[j][i] / terminated[j][i] (stride = elements
// between consecutive steps).  Bit-identical outputs; 6 B of traffic per env-step instead of 38.
//
// The only memory latency on a step's critical path is its action byte, and the only thing a step must not wait
// for is the write acknowledgement of the previous steps' stores.  So the action loads run kPrefetch steps ahead
// of the arithmetic, as a software pipeline in registers: the step loop is unrolled 2 * kPrefetch times over TWO
// register sets -- steps of the first half consume set A and refill set B, the second half consumes B and refills
// A -- so every in-flight action has a fixed register whose old value is dead when the new load is issued (rotating
// registers with moves, or reloading the slot that is being consumed, makes the compiler wait for everything at the
// loop latch).  Step j then waits with s_waitcnt vmcnt(n > 0) for a load issued kPrefetch steps -- thousands of
// cycles -- earlier, and the younger loads and stores stay in flight.  The I/O pointers are __restrict__, which
// lets the compiler issue a step's load ahead of the previous steps' stores.
constexpr uint32_t kPrefetch = 4;

// the in-flight actions keep their memory type (a conversion next to the load would be a use of its result: the
// wait would land there instead of kPrefetch steps later)
template <int ACT> struct RawAction { typedef uint32_t type; };
template <> struct RawAction<1> { typedef uint8_t type; };
template <> struct RawAction<2> { typedef int32_t type; };
template <> struct RawAction<3> { typedef long long type; };

template <int ACT>
__device__ __forceinline__ typename RawAction<ACT>::type load_action_at(const void *__restrict__ actions, size_t idx)
{
    if constexpr (ACT == 0)
        return 0u;
    else
        return __builtin_nontemporal_load(static_cast<const typename RawAction<ACT>::type *>(actions) + idx);
}

template <int ACT>
__global__ void __launch_bounds__(kBlock) rollout_fused_kernel(const StepArgs p, uint64_t stride)
{
    __shared__ WaveTables s_tables[kBlock / 64];
    const uint32_t i_raw = blockIdx.x * kBlock + threadIdx.x;
    const bool valid = i_raw < p.n;
    const uint32_t i = valid ? i_raw : p.n - 1u;
    const void *__restrict__ actions = p.actions;
    float *__restrict__ reward = p.reward;
    uint8_t *__restrict__ terminated = p.terminated;
    uint8_t *__restrict__ illegal = p.illegal;
    uint8_t *__restrict__ highest_out = p.highest;
    const uint32_t k = p.k_steps;
    // the first kPrefetch actions (register set A) are requested before anything else
    typename RawAction<ACT>::type ahead[kPrefetch];
#pragma unroll
    for (uint32_t q = 0; q < kPrefetch; ++q) {
        ahead[q] = load_action_at<ACT>(actions, static_cast<size_t>(q < k ? q : k - 1u) * stride + i);
        asm volatile("" ::: "memory"); // issue order = consumption order (the scheduler reverses them otherwise, and
                                       // the loop's first wait would then have to cover all of them)
    }
    Board rec = load_board_nt(p.st.boards, i);
    const LdsTables tb = stage_tables(s_tables, load_tables_piece());
    const EpisodeCounters counters = load_episode_counters(p, i_raw);
    uint64_t t = (static_cast<uint64_t>(p.t_hi) << 32) | p.t_lo;
    uint32_t episodes = 0, illegal_ends = 0;
    unsigned long long gained = 0; // this lane's merge scores over the k steps: 32-bit adds, folded once per double group
    uint32_t gained32 = 0;
    unsigned long long ended_last = 0; // lanes whose episode the LAST step ended (pending marks when auto_reset == 0)
    // step j with its action; t advances with it
    bool bad_action = false;   // strict actions: any of this lane's k actions outside 0..3 (checked where they are fetched)
    uint32_t bad_raw = 0;
    auto one_step = [&](uint32_t j, uint32_t action_in) {
        const size_t o_idx = static_cast<size_t>(j) * stride + i;
        const Words w = philox4x32_10(static_cast<uint32_t>(t), static_cast<uint32_t>(t >> 32), p.board_offset + i, 0u,
                                      p.seed_lo, p.seed_hi);
        ++t;
        const uint32_t action = ACT == 0 ? w.w[3] >> 30 : action_in & 3u;
        const StepOut o = play_record(rec, action, w, p.max_exp, tb);
        gained32 += o.gain;
        ended_last = record_episode_ends(p, i, o.terminated && valid, !o.legal, rec, episodes, illegal_ends);
        if (valid) {
            if (reward)
                __builtin_nontemporal_store(o.legal ? static_cast<float>(o.gain) : p.illegal_reward, reward + o_idx);
            if (terminated)
                __builtin_nontemporal_store(static_cast<uint8_t>(o.terminated ? 1 : 0), terminated + o_idx);
            if (illegal)
                __builtin_nontemporal_store(static_cast<uint8_t>(o.legal ? 0 : 1), illegal + o_idx);
            if (highest_out)
                __builtin_nontemporal_store(static_cast<uint8_t>(highest(record_cells(rec))), highest_out + o_idx);
        }
        if (o.terminated && p.auto_reset != 0)
            reset_record(rec, o, w, tb);
    };
    // past the end the last step's action is fetched again and never used (an unconditional load: a conditional one
    // would be waited for where the branches join)
    auto fetch = [&](uint32_t j) { return load_action_at<ACT>(actions, static_cast<size_t>(j < k ? j : k - 1u) * stride + i); };
    // (every action that is played passes through `check` right before its step)
    auto check = [&](typename RawAction<ACT>::type raw) {
        if constexpr (ACT != 0) {
            typedef typename std::make_unsigned<typename RawAction<ACT>::type>::type U;
            if (static_cast<U>(raw) > static_cast<U>(3)) {
                bad_action = true;
                bad_raw = static_cast<uint32_t>(raw);
            }
        }
        return static_cast<uint32_t>(raw);
    };
    typename RawAction<ACT>::type set_b[kPrefetch];
    uint32_t j0 = 0;
    for (; j0 + 2u * kPrefetch <= k; j0 += 2u * kPrefetch) { // whole double groups
#pragma unroll
        for (uint32_t q = 0; q < kPrefetch; ++q) {
            if constexpr (ACT != 0)
                set_b[q] = fetch(j0 + q + kPrefetch);
            one_step(j0 + q, check(ahead[q]));
        }
#pragma unroll
        for (uint32_t q = 0; q < kPrefetch; ++q) {
            if constexpr (ACT != 0)
                ahead[q] = fetch(j0 + q + 2u * kPrefetch);
            one_step(j0 + kPrefetch + q, check(set_b[q]));
        }
        gained += gained32;
        gained32 = 0;
    }
    // the last k % (2 * kPrefetch) steps: the first kPrefetch of them have their actions in flight already
#pragma unroll
    for (uint32_t q = 0; q < kPrefetch; ++q)
        if (j0 + q < k)
            one_step(j0 + q, check(ahead[q]));
    for (uint32_t j = j0 + kPrefetch; j < k; ++j)
        one_step(j, check(fetch(j)));
    gained += gained32; // (at most 2 * kPrefetch - 1 tail steps)
    if (valid)
        store_board_nt(p.st.boards, i, rec);
    flush_episode_counts(counters, episodes, illegal_ends, wave_sum64_lane63(valid ? gained : 0ull),
                         pending_after_step(ended_last, p.auto_reset));
    if constexpr (ACT != 0) {
        if (p.action_err)
            report_bad_action(p.action_err, bad_action && valid, bad_raw, p.board_offset + i);
    }
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

// Game2048Env.step + the caller's `if terminated: env.reset()` in ONE launch.  A spawn in this mode is ~1 000 VALU
// instructions (15 accepted draws of the Fisher-Yates shuffle over 32-bit halves of a 128-bit LCG, in lockstep per
// wavefront), and a reset is two of them: run in-lane it would be executed by 99 % of the wavefronts for the 7 % of the
// boards that need it.  Until round 6 the finished boards were listed per wavefront and reset by a SECOND, compacted
// kernel (19 us behind a 42.6 us step at 2^20 boards: two waves per SIMD, each a serial chain of 128-bit multiplies).
// Now the compaction happens inside the launch, per 512-lane block: a lane whose episode ended hands its generator over
// through LDS instead of storing it (slot = the block's running count + its rank among the wavefront's finished lanes),
// and behind ONE barrier the block's first `total` lanes -- under a random policy ~36 of 512: part of one wavefront,
// the other seven retire -- each reset one listed board and store its fresh record and generator.  The reset work of
// a block overlaps the step work of the other blocks on the CU.
// (512 lanes: 45.0-46.1 us per step at 2^20 boards against 48.2 / 48.7 / 48.4 for 256 / 768 / 1 024, one box,
//  profiles/r06_d_numpy_block_ab.txt; forcing 8 waves per SIMD -- 64 VGPRs, five dwords spilled -- changes nothing.
//  Measured "no", same round: a software pipeline over 2 / 4 / 8 tiles per block with the next tile's records and generators
//  prefetched into registers and a ninth, reset-only wavefront: 65 / 61 / 59 us -- 89 VGPRs leave four stepping wavefronts
//  per SIMD and the serial 128-bit multiply chains need six to fill the VALU, profiles/r06_g_numpy_tiles_ab.txt; starting
//  the odd blocks 3.4 us late to pull the load / compute / store phases of co-resident blocks apart: +10 us,
//  profiles/r06_h_numpy_stagger_ab.txt.  What does help is not touching memory per step at all: rollout_fused_numpy_kernel.)
constexpr uint32_t kNumpyBlock = 512;
static_assert(kNumpyBlock == kSlotBlockLanes, "the slot array is sized for whole blocks of this kernel");

template <int ACT>
__global__ void __launch_bounds__(kNumpyBlock) step_numpy_kernel(const StepArgs p)
{
    __shared__ uint64_t s_rng[5][kNumpyBlock]; // the generators handed over (plane-major: conflict-free), 20 KiB
    __shared__ uint16_t s_who[kNumpyBlock];    // ... whose they are (lane index in the block) ...
    __shared__ uint8_t s_first[kNumpyBlock];   // ... and the first tile of the reset where it is drawn already: cell | four << 4, else 0xff
    __shared__ uint32_t s_count;
    if (threadIdx.x == 0u)
        s_count = 0u;
    const uint32_t i_raw = blockIdx.x * kNumpyBlock + threadIdx.x;
    const bool valid = i_raw < p.n;
    const uint32_t i = valid ? i_raw : p.n - 1u;
    Board rec = load_board(p.st.boards, i);
    Pcg64 rng = load_rng(p.st.rng, p.n, i);
    const EpisodeCounters counters = load_episode_counters(p, i_raw);
    uint32_t action;
    if constexpr (ACT == 0)
        action = philox4x32_10(p.t_lo, p.t_hi, p.board_offset + i, 0u, p.seed_lo, p.seed_hi).w[3] >> 30;
    else {
        bool bad_action = false;
        uint32_t raw_action = 0;
        action = load_action_checked<ACT>(p.actions, i, bad_action, raw_action);
        if (p.action_err) // strict actions (wave-uniform)
            report_bad_action(p.action_err, bad_action && valid, raw_action, p.board_offset + i);
    }
    if (p.auto_reset != 0u)
        __syncthreads(); // s_count = 0 is visible before the first hand-over (placed here: the loads above are in flight)

    // rec: the terminal record where the episode ended.  A lane whose move is illegal draws the first tile of its reset here
    const NumpyStepOut o = play_record_numpy(rec, action, rng, p.max_exp, p.auto_reset != 0u);
    const uint32_t wave_gain = wave_sum_lane63(valid ? o.gain : 0u);         // :86

    const bool fin = o.terminated && valid;
    const bool hand_over = fin && p.auto_reset != 0u;
    if (valid) {
        if (!hand_over) { // (a board that is reset below is stored there, once)
            store_board(p.st.boards, i, rec);
            store_rng(p.st.rng, p.n, i, rng);
        }
        if (p.reward)
            p.reward[i] = o.legal ? static_cast<float>(o.gain) : p.illegal_reward;  // :90 / :95
        if (p.terminated)
            p.terminated[i] = o.terminated ? 1 : 0;
        if (p.illegal)
            p.illegal[i] = o.legal ? 0 : 1;
        if (p.highest)
            p.highest[i] = static_cast<uint8_t>(o.top);
    }
    uint32_t episodes = 0, illegal_ends = 0;
    const unsigned long long ended = record_episode_ends(p, i, fin, !o.legal, rec, episodes, illegal_ends);
    flush_episode_counts(counters, episodes, illegal_ends, wave_gain, pending_after_step(ended, p.auto_reset));
    if (p.auto_reset == 0u) // (kernel-uniform)
        return;
    // ---- `if terminated: env.reset()` (game2048_env.py:102-111), compacted over the block
    const unsigned long long done = __builtin_amdgcn_ballot_w64(hand_over);
    if (done != 0ull) { // (wave-uniform)
        uint32_t base = 0;
        if ((threadIdx.x & 63u) == 0u)
            base = atomicAdd(&s_count, static_cast<uint32_t>(__popcll(done)));
        base = __builtin_amdgcn_readfirstlane(base);
        if (hand_over) {
            const uint32_t e = base + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(done >> 32),
                                                                __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(done), 0u));
            s_who[e] = static_cast<uint16_t>(threadIdx.x);
            s_first[e] = o.have_first ? static_cast<uint8_t>(o.first_cell | (o.first_four ? 16u : 0u)) : static_cast<uint8_t>(0xffu);
            s_rng[0][e] = rng.state_lo;
            s_rng[1][e] = rng.state_hi;
            s_rng[2][e] = rng.inc_lo;
            s_rng[3][e] = rng.inc_hi;
            s_rng[4][e] = rng.buf;
        }
    }
    __syncthreads();
    const uint32_t total = s_count;
    if ((threadIdx.x & ~63u) >= total) // (whole wavefronts stay or go: finish_record_numpy votes)
        return;
    const bool mine = threadIdx.x < total;
    const uint32_t e = mine ? threadIdx.x : 0u;
    const uint32_t j = blockIdx.x * kNumpyBlock + s_who[e];
    const uint32_t first = s_first[e];
    Pcg64 r2{s_rng[0][e], s_rng[1][e], s_rng[2][e], s_rng[3][e], s_rng[4][e]};
    const Board fresh = finish_record_numpy(r2, first != 0xffu || !mine, first & 15u, (first & 16u) != 0u);
    if (mine) {
        store_board(p.st.boards, j, fresh);
        store_rng(p.st.rng, p.n, j, r2);
    }
}

// ---- k steps of numpy-RNG mode in ONE launch, record and generator in registers (g2048_rollout_fused / _random)
// A reset in this mode is two ~1 000-instruction spawns that only the 7 % of the lanes whose episode ended need: the
// per-step kernel above compacts them over the block, which a kernel that keeps its boards in registers cannot do without
// shipping generators through LDS twice per step.  Instead the lanes of a wavefront are allowed to DRIFT IN TIME: every
// trip of the loop is one lockstep spawn, and each lane spends it on what IT needs next --
//   * a lane that is stepping plays its own step j (its own row of the [k][n] buffers) and spawns that step's tile;
//   * a lane whose move was illegal spawns nothing in its step (game2048_env.py:91-95), so it draws the FIRST tile of its
//     reset in the same trip, and the second one in the next trip while its neighbours play their next step;
//   * a lane whose episode ended on a legal move (no moves left / max_tile) spends the next two trips on its reset;
// and the loop ends when every lane has played k steps and finished its resets.  A board's generator is consumed in exactly
// the reference's order (step spawn, then the reset's two), whatever its neighbours do.  Under a random policy a lane
// resets ~0.07 times per step, nearly always after an illegal move: ~1.1 trips per step instead of the 3 an in-lane reset
// would cost every wavefront.  The price is I/O that is no longer coalesced once the lanes have drifted apart (each lane
// reads / writes ITS row j): 6 B per env-step through partially written lines, which the L2 and the Infinity Cache merge.
template <int ACT>
__global__ void __launch_bounds__(kBlock) rollout_fused_numpy_kernel(const StepArgs p, uint64_t stride)
{
    const uint32_t i_raw = blockIdx.x * kBlock + threadIdx.x;
    const bool valid = i_raw < p.n;
    const uint32_t i = valid ? i_raw : p.n - 1u;
    Board rec = load_board(p.st.boards, i);
    Pcg64 rng = load_rng(p.st.rng, p.n, i);
    const EpisodeCounters counters = load_episode_counters(p, i_raw);
    const uint64_t t0 = (static_cast<uint64_t>(p.t_hi) << 32) | p.t_lo; // transaction of this lane's step 0
    const uint32_t k = p.k_steps;
    uint32_t j = 0;        // the step this lane plays next
    uint32_t phase = 0;    // 0: stepping; 1: reset pending, no tile drawn yet; 2: reset pending, first tile drawn
    Board fresh{{0u, 0u, 0u, 0u}};
    uint32_t fresh_fours = 0;
    uint32_t episodes = 0, illegal_ends = 0, gained32 = 0;
    unsigned long long gained = 0;
    bool last_ended = false; // this lane's step k - 1 ended its episode (the pending mark when auto_reset == 0)
    bool bad_action = false;
    uint32_t bad_raw = 0;
    for (;;) {
        const bool stepping = phase == 0u && j < k;
        if (!g2048_any(stepping || phase != 0u)) // (wave-uniform)
            break;
        // ---- the step's move (lanes that are stepping)
        uint32_t gain = 0;
        bool legal = false;
        Board cells = record_cells(rec);
        const size_t io_idx = static_cast<size_t>(j < k ? j : 0u) * stride + i;
        if (stepping) {
            uint32_t action;
            if constexpr (ACT == 0) {
                const uint64_t t = t0 + j;
                action = philox4x32_10(static_cast<uint32_t>(t), static_cast<uint32_t>(t >> 32), p.board_offset + i, 0u, p.seed_lo,
                                       p.seed_hi).w[3] >> 30;
            } else {
                const typename RawAction<ACT>::type v = load_action_at<ACT>(p.actions, io_idx);
                typedef typename std::make_unsigned<typename RawAction<ACT>::type>::type U;
                if (static_cast<U>(v) > static_cast<U>(3)) {
                    bad_action = true;
                    bad_raw = static_cast<uint32_t>(v);
                }
                action = static_cast<uint32_t>(v) & 3u;
            }
            legal = move(cells, action, gain);                                  // :85
        }
        // ---- the trip's ONE lockstep spawn: the step's tile, or a tile of the lane's reset
        const bool early = stepping && !legal && p.auto_reset != 0u;            // illegal move: first reset tile now
        const bool spawn = (stepping && legal) || early || phase != 0u;
        Board target = (stepping && legal) ? cells : (phase == 2u ? fresh : Board{{0u, 0u, 0u, 0u}});
        bool four = false;
        if (spawn)
            four = add_tile_numpy(target, rng);                                 // :88 / :108 / :109
        // ---- what the spawn was for
        bool fin = false;
        if (stepping) {
            bool end = false;
            if (legal) {
                cells = target;
                end = is_end(cells, p.max_exp);                                 // :89
            }
            const bool terminated = legal ? end : true;                         // :89, :94
            record_update(rec, cells, (legal && four) ? 0x80u : 0u);            // a spawned 4: deficit += 4
            gained32 += gain;                                                   // :86
            if (valid) {
                if (p.reward)
                    __builtin_nontemporal_store(legal ? static_cast<float>(gain) : p.illegal_reward, p.reward + io_idx);
                if (p.terminated)
                    __builtin_nontemporal_store(static_cast<uint8_t>(terminated ? 1 : 0), p.terminated + io_idx);
                if (p.illegal)
                    __builtin_nontemporal_store(static_cast<uint8_t>(legal ? 0 : 1), p.illegal + io_idx);
                if (p.highest)
                    __builtin_nontemporal_store(static_cast<uint8_t>(highest(cells)), p.highest + io_idx);
            }
            fin = terminated && valid;
            if (j + 1u == k)
                last_ended = fin;
            ++j;
            if (terminated && p.auto_reset != 0u) {                             // the caller's `if terminated: env.reset()`
                phase = early ? 2u : 1u;
                fresh = target;                                                 // (early: the empty board + its first tile)
                fresh_fours = (early && four) ? 1u : 0u;
            }
        } else if (phase == 1u) {
            fresh = target;
            fresh_fours = four ? 1u : 0u;
            phase = 2u;
        } else if (phase == 2u) {
            fresh_fours += four ? 1u : 0u;
            rec = target;                                                       // :104-109: score 0, so deficit = 4 per spawned 4
            rec.r[2] |= fresh_fours == 1u ? 0x80u : (fresh_fours == 2u ? 0x2000u : 0u);
            phase = 0u;
        }
        // episode ends of this trip (terminal record first: rec still holds it -- a reset only lands two trips later)
        (void)record_episode_ends(p, i, fin, !legal, rec, episodes, illegal_ends);
        if ((gained32 >> 30) != 0u) { // (a lane gains < 2^18 per step: folded long before it could wrap)
            gained += gained32;
            gained32 = 0;
        }
    }
    gained += gained32;
    if (valid) {
        store_board(p.st.boards, i, rec);
        store_rng(p.st.rng, p.n, i, rng);
    }
    const unsigned long long ended_last = __builtin_amdgcn_ballot_w64(last_ended);
    flush_episode_counts(counters, episodes, illegal_ends, wave_sum64_lane63(valid ? gained : 0ull),
                         pending_after_step(ended_last, p.auto_reset));
    if constexpr (ACT != 0) {
        if (p.action_err)
            report_bad_action(p.action_err, bad_action && valid, bad_raw, p.board_offset + i);
    }
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

// Return accounting of an explicit reset (g2048_reset; whole wavefronts): see adjust_slot.
__device__ __forceinline__ void account_reset(const StepArgs &p, uint32_t i_raw, uint32_t i, bool doit)
{
    const bool was_pending = is_pending(p.st.ep_counters, i_raw);
    int32_t delta = 0;
    if (doit && !was_pending) // an episode that is still running is abandoned: its score leaves the books
        delta = -static_cast<int32_t>(record_score(load_board(p.st.boards, i)));
    adjust_slot(p.st.ep_counters, i_raw, delta, doit);
}

__global__ void __launch_bounds__(kBlock) reset_numpy_kernel(const StepArgs p, const uint8_t *mask)
{
    const uint32_t i_raw = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t i = i_raw < p.n ? i_raw : p.n - 1u;
    const bool doit = i_raw < p.n && !(mask && mask[i] == 0);
    account_reset(p, i_raw, i, doit);
    if (!doit)
        return;
    Pcg64 rng = load_rng(p.st.rng, p.n, i);
    const Board fresh = fresh_record_numpy(rng); // game2048_env.py:104-109
    store_rng(p.st.rng, p.n, i, rng);
    store_board(p.st.boards, i, fresh);
}

__global__ void __launch_bounds__(kBlock) add_tile_numpy_kernel(const StepArgs p)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= p.n)
        return;
    const Board raw = load_board(p.st.boards, i);
    Board bd = record_cells(raw);
    if (count_empty(bd) == 0)
        return;
    Pcg64 rng = load_rng(p.st.rng, p.n, i);
    const bool four = add_tile_numpy(bd, rng);
    store_rng(p.st.rng, p.n, i, rng);
    Board rec = raw;
    record_update(rec, bd, four ? 0x80u : 0u); // a spawned 4 raises the potential without scoring: deficit += 4
    store_board(p.st.boards, i, rec);
}

// ---------------------------------------------------------------------------------- reset
// Game2048Env.reset for every (masked) board (game2048_env.py:102-111).
__global__ void __launch_bounds__(kBlock) reset_kernel(const StepArgs p, uint32_t first_slot, const uint8_t *mask)
{
    const uint32_t i_raw = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t i = i_raw < p.n ? i_raw : p.n - 1u;
    const bool doit = i_raw < p.n && !(mask && mask[i] == 0);
    account_reset(p, i_raw, i, doit);
    if (!doit)
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
    store_board(p.st.boards, i, fresh_record(w1, w2)); // score 0 (:105) is part of the record
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
    const Board raw = load_board(boards, i);
    Board bd = record_cells(raw);
    uint32_t gain;
    const bool legal = move(bd, load_action<ACT>(actions, i, 0u), gain);
    if (score_out)
        score_out[i] = legal ? static_cast<int32_t>(gain) : 0;
    if (legal_out)
        legal_out[i] = legal ? 1 : 0;
    if (!trial && legal)
        store_board(boards, i, make_record(bd, record_score(raw))); // self.score is not touched by move()
}

// Game2048Env.isend (game2048_env.py:262-280) and highest (:190-192).
__global__ void __launch_bounds__(kBlock) query_kernel(const uint4 *boards, uint32_t n, uint32_t max_exp,
                                                       uint8_t *isend_out, uint8_t *highest_out)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n)
        return;
    const Board bd = record_cells(load_board(boards, i));
    if (isend_out)
        isend_out[i] = is_end(bd, max_exp) ? 1 : 0;
    if (highest_out)
        highest_out[i] = static_cast<uint8_t>(highest(bd));
}

// The four trial moves of isend (game2048_env.py:273-280) as a mask: bit d = move d changes the board.
__global__ void __launch_bounds__(kBlock) legal_mask_kernel(const uint4 *boards, uint32_t n, uint8_t *mask_out)
{
    __shared__ WaveTables s_tables[kBlock / 64];
    const LdsTables tb = stage_tables(s_tables, load_tables_piece());
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n)
        return;
    const Board cells = record_cells(load_board(boards, i));
    uint32_t mask = 0;
#pragma unroll
    for (uint32_t d = 0; d < 4u; ++d) {
        Board trial = cells;                                   // trial=True: the board itself is left alone (:224,:236)
        uint32_t gain;
        if (move_sel(trial, tb.move_sel(d), gain))
            mask |= 1u << d;
    }
    mask_out[i] = static_cast<uint8_t>(mask);
}

// Game2048Env.add_tile (game2048_env.py:166-176) from spawn slot `slot` of transaction t.
__global__ void __launch_bounds__(kBlock) add_tile_kernel(const StepArgs p, uint32_t slot)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= p.n)
        return;
    const Board raw = load_board(p.st.boards, i);
    Board bd = record_cells(raw);
    if (count_empty(bd) == 0) // the reference asserts here (:176)
        return;
    const Words w = philox4x32_10(p.t_lo, p.t_hi, p.board_offset + i, slot >> 2, p.seed_lo, p.seed_hi);
    add_tile(bd, select_word(w, slot & 3u));
    store_board(p.st.boards, i, make_record(bd, record_score(raw)));
}

// ------------------------------------------------------------- records <-> plain boards / scores
// get_board / set_board (game2048_env.py:282-288) and self.score for the whole batch.
__global__ void __launch_bounds__(kBlock) export_boards_kernel(const uint4 *records, uint32_t n, uint4 *cells_out)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n)
        store_board(cells_out, i, record_cells(load_board(records, i)));
}

__global__ void __launch_bounds__(kBlock) import_boards_kernel(uint4 *records, uint32_t n, const uint4 *cells_in)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n)
        return;
    const Board in = load_board(cells_in, i);
    const Board cells{{in.r[0] & kCellBits, in.r[1] & kCellBits, in.r[2] & kCellBits, in.r[3] & kCellBits}};
    store_board(records, i, make_record(cells, record_score(load_board(records, i)))); // score untouched (:286-288)
}

__global__ void __launch_bounds__(kBlock) export_scores_kernel(const uint4 *records, uint32_t n, int32_t *scores_out)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n)
        scores_out[i] = static_cast<int32_t>(record_score(load_board(records, i)));
}

__global__ void __launch_bounds__(kBlock) import_scores_kernel(uint4 *records, uint32_t n, const int32_t *scores_in,
                                                               unsigned long long *ep_counters)
{
    const uint32_t i_raw = blockIdx.x * kBlock + threadIdx.x;
    const bool valid = i_raw < n;
    const uint32_t i = valid ? i_raw : n - 1u;
    const Board raw = load_board(records, i);
    const uint32_t score = static_cast<uint32_t>(scores_in[i]) & kScoreMask;
    // return accounting: the live score of a running episode changes by (new - old); a board whose episode has ended
    // keeps its place among the finished ones
    const bool live = valid && !is_pending(ep_counters, i_raw);
    adjust_slot(ep_counters, i_raw, live ? static_cast<int32_t>(score) - static_cast<int32_t>(record_score(raw)) : 0, false);
    if (valid)
        store_board(records, i, make_record(record_cells(raw), score));
}

// g2048_seed: forget the finished episodes (game2048_env.py:103 restarts the stream).  Every slot restarts with
// G = the scores its 64 boards hold right now (so "G - live scores" = 0: no finished episode) and no pending marks.
__global__ void __launch_bounds__(kBlock) clear_stats_kernel(const DeviceState st, uint32_t n)
{
    const uint32_t i_raw = blockIdx.x * kBlock + threadIdx.x;
    const bool valid = i_raw < n;
    uint32_t score = 0;
    if (valid) {
        if (st.last_record)
            st.last_record[i_raw] = make_uint4(0u, 0u, 0u, 0u);
        score = record_score(load_board(st.boards, i_raw));
    }
    const uint32_t total = wave_sum_lane63(score); // <= 64 * (2^24 - 1)
    if ((threadIdx.x & 63u) == 63u) {
        uint32_t *slot = slot_of(st.ep_counters, i_raw >> 6);
        slot_set(slot, kSlotEpisodes, 0ull);
        slot_set(slot, kSlotIllegal, 0ull);
        slot_set(slot, kSlotGain, total);
        slot_set(slot, kSlotPending, 0ull);
    }
}

// final score of each board's most recent finished episode (0 before the first one)
__global__ void __launch_bounds__(kBlock) export_last_scores_kernel(const uint4 *last_record, uint32_t n, int32_t *out)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n)
        out[i] = static_cast<int32_t>(record_score(load_board(last_record, i)));
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
// stack() (game2048_env.py:17-32): board record -> (16,4,4), channel c = (exponent == c).  One lane writes
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
        store_chunk_nt(out, g, z80(v.x ^ splat) >> 7, z80(v.y ^ splat) >> 7, z80((v.z & kCellBits) ^ splat) >> 7,
                       z80((v.w & kCellBits) ^ splat) >> 7);
    } else if constexpr (OBS == 1) {
        const uint64_t board = g >> 5;
        const uint32_t c = static_cast<uint32_t>(g >> 1) & 15u, half = static_cast<uint32_t>(g) & 1u;
        const uint32_t r0 = cells[board * 4 + half * 2] & kCellBits, r1 = cells[board * 4 + half * 2 + 1] & kCellBits;
        uint32_t h[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            h[k] = (((r0 >> (8 * k)) & 0xffu) == c) ? 0x3c00u : 0u; // fp16 1.0
            h[4 + k] = (((r1 >> (8 * k)) & 0xffu) == c) ? 0x3c00u : 0u;
        }
        store_chunk_nt(out, g, h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    } else {
        const uint64_t board = g >> 6;
        const uint32_t c = static_cast<uint32_t>(g >> 2) & 15u, row = static_cast<uint32_t>(g) & 3u;
        const uint32_t r = cells[board * 4 + row] & kCellBits;
        uint32_t f[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            f[k] = (((r >> (8 * k)) & 0xffu) == c) ? 0x3f800000u : 0u; // fp32 1.0
        store_chunk_nt(out, g, f[0], f[1], f[2], f[3]);
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

// ------------------------------------------------------------------------- canonicalisation
// Symmetric-board canonicalisation: of the eight symmetries training_data.augment()
// generates (training_data.py:257-299, same order: variant = 2 * quarter_turns + hflip), keep the one
// whose 16 cells, read row-major as bytes, are lexicographically smallest (ties: the lowest variant
// index); the action is remapped and the next board transformed by the same symmetry.  In place.
// Row-major byte order = compare r[0] first, byte 0 most significant: bswap makes that an integer compare.
__device__ __forceinline__ bool board_less(const Board &a, const Board &b)
{
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t x = __builtin_bswap32(a.r[k]), y = __builtin_bswap32(b.r[k]);
        if (x != y)
            return x < y;
    }
    return false;
}

__device__ __forceinline__ Board symmetry(const Board &b, uint32_t variant)
{
    return rotate_board((variant & 1u) ? hflip_board(b) : b, variant >> 1);
}

__global__ void __launch_bounds__(kBlock) canonicalize_kernel(uint4 *boards, uint4 *next_boards, uint8_t *actions,
                                                              uint32_t n, uint8_t *sym_out)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n)
        return;
    const Board b = load_board(boards, i);
    Board best = b;
    uint32_t best_v = 0;
#pragma unroll
    for (uint32_t v = 1; v < 8; ++v) {
        const Board c = symmetry(b, v);
        if (board_less(c, best)) {
            best = c;
            best_v = v;
        }
    }
    store_board(boards, i, best);
    if (next_boards)
        store_board(next_boards, i, symmetry(load_board(next_boards, i), best_v));
    if (actions) {
        uint32_t a = actions[i] & 3u;
        if (best_v & 1u)
            a = (a == 1u) ? 3u : (a == 3u ? 1u : a); // hflip swaps right <-> left (:262-267)
        actions[i] = static_cast<uint8_t>((a + (best_v >> 1)) & 3u); // a clockwise quarter turn adds 1 (:276)
    }
    if (sym_out)
        sym_out[i] = static_cast<uint8_t>(best_v);
}

// ----------------------------------------------------------------------------------- stats
// Reduce to one StatsOut: the episode counters, the exact sum of the final scores of ALL finished episodes
// (return_sum = sum of the slots' G - sum of the scores of the boards whose episode is still running, see the slot
// layout above), the returns of the boards' most recent finished episodes (score of last_record: count / sum / max),
// the highest tile on any board and the histogram of the boards' highest tiles (what ppo_train.py:77-81 tallies per
// finished episode, here for the live boards).  Two stages and NO global atomics: a single hot cache line serialises at
// ~88 atomics/us on this chip (a one-stage version spent 55 us at 2^20 boards in its ~5 000 final atomics).
//   stage 1: every block reduces its grid-stride share (registers, then a tree in LDS) to kStatsFields
//            numbers, stored field-major: partials[f * kStatsBlocks + block].  The histogram is counted by
//            per-wave ballots (the counts are wave-uniform: they live in SGPRs);
//   stage 2: kStatsFields blocks, block f reduces field f over the partials (coalesced reads, tree in LDS).
// Cross-lane shuffles are avoided on purpose: a 64-bit __shfl_xor chain costs ~100 cycles per step and the
// reductions here are latency-, not throughput-bound.
constexpr int kStatsScalars = 7; // episodes, illegal_ends, last_count, last_score_sum, last_score_max, max_exp, return_sum
constexpr int kStatsFields = kStatsScalars + 32; // + hist[32]
static_assert(kStatsFields * kStatsBlocks == kStatsPartialWords, "partials buffer size");

// (The RETURNS-ONLY flavour -- what a multi-GPU job all-gathers once per rollout -- is returns_summary_kernel below: one
// launch.  Until round 6 it was this kernel pair without the terminal records and the histogram: 11.5 us per call at 2^20
// boards against 7.4, profiles/r06_b_stats_probe_*.txt.)
__global__ void __launch_bounds__(kBlock) stats_kernel(const DeviceState st, uint32_t n, uint32_t n_waves,
                                                       unsigned long long *partials)
{
    __shared__ unsigned long long s_ep[kBlock], s_ill[kBlock], s_sum[kBlock], s_cnt[kBlock], s_ret[kBlock];
    __shared__ unsigned int s_max[kBlock], s_exp[kBlock];
    __shared__ unsigned int s_hist[32];
    unsigned long long episodes = 0, illegal = 0, score_sum = 0, count = 0;
    unsigned long long ret = 0; // sum of G over this thread's slots - sum of the live scores of its boards (mod 2^64)
    unsigned int max_score = 0, max_exp = 0;
    uint32_t hist[32];
#pragma unroll
    for (int b = 0; b < 32; ++b)
        hist[b] = 0u;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    if (tid < 32u)
        s_hist[tid] = 0u;
    __syncthreads();
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t wv = blockIdx.x * kBlock + tid; wv < n_waves; wv += stride) {
        const uint32_t *slot = slot_of(st.ep_counters, wv);
        const uint4 lo = *reinterpret_cast<const uint4 *>(slot), hi = *reinterpret_cast<const uint4 *>(slot + 4);
        episodes += static_cast<unsigned long long>(hi.x) << 32 | lo.x;
        illegal += static_cast<unsigned long long>(hi.y) << 32 | lo.y;
        ret += static_cast<unsigned long long>(hi.z) << 32 | lo.z;
    }
    // whole wavefronts iterate together (the ballots need every lane's vote): lanes past the end vote "none"
    for (uint64_t base = blockIdx.x * kBlock + tid - lane; base < n; base += stride) { // 64-bit: n may be near 2^32
        const uint64_t i = base + lane;
        // boards whose episode has ended but which were not reset (auto_reset == 0): their score is a finished episode's
        const unsigned long long pending = slot_get(slot_of(st.ep_counters, static_cast<uint32_t>(base >> 6)), kSlotPending); // uniform: scalar loads
        uint32_t h = 0xffu;
        if (i < n) {
            const Board live = load_board_nt(st.boards, static_cast<uint32_t>(i));
            if (((pending >> lane) & 1ull) == 0ull)
                ret -= record_score(live);
            h = highest(record_cells(live));
            max_exp = max(max_exp, h);
            Board last{{0u, 0u, 0u, 0u}};
            if (st.last_record) // NULL: terminal records are not kept, last_* stay zero
                last = load_board_nt(st.last_record, static_cast<uint32_t>(i));
            if ((last.r[0] | last.r[1] | last.r[2] | last.r[3]) != 0u) { // a terminal board is never empty
                const unsigned int sc = record_score(last);
                count += 1;
                score_sum += sc;
                max_score = max(max_score, sc);
            }
        }
#pragma unroll
        for (uint32_t b = 0; b < 32u; ++b)
            hist[b] += static_cast<uint32_t>(__popcll(__builtin_amdgcn_ballot_w64(h == b)));
    }
    s_ep[tid] = episodes; s_ill[tid] = illegal; s_sum[tid] = score_sum; s_cnt[tid] = count; s_ret[tid] = ret;
    s_max[tid] = max_score; s_exp[tid] = max_exp;
    if (lane == 0u) {
#pragma unroll
        for (int b = 0; b < 32; ++b)
            if (hist[b] != 0u)
                atomicAdd(&s_hist[b], hist[b]);
    }
    __syncthreads();
    for (uint32_t off = kBlock / 2; off > 0; off >>= 1) {
        if (tid < off) {
            s_ep[tid] += s_ep[tid + off];
            s_ill[tid] += s_ill[tid + off];
            s_sum[tid] += s_sum[tid + off];
            s_cnt[tid] += s_cnt[tid + off];
            s_ret[tid] += s_ret[tid + off];
            s_max[tid] = max(s_max[tid], s_max[tid + off]);
            s_exp[tid] = max(s_exp[tid], s_exp[tid + off]);
        }
        __syncthreads();
    }
    unsigned long long *mine = partials + blockIdx.x;
    if (tid == 0) {
        mine[0 * kStatsBlocks] = s_ep[0];
        mine[1 * kStatsBlocks] = s_ill[0];
        mine[2 * kStatsBlocks] = s_cnt[0];
        mine[3 * kStatsBlocks] = s_sum[0];
        mine[4 * kStatsBlocks] = s_max[0];
        mine[5 * kStatsBlocks] = s_exp[0];
        mine[6 * kStatsBlocks] = s_ret[0];
    }
    if (tid < 32u)
        mine[(kStatsScalars + tid) * kStatsBlocks] = s_hist[tid];
}

// block f: field f of every partial -> the matching member of *out (fields 4 and 5 are maxima, the rest sums mod 2^64)
// last_known: the terminal records were read (the full flavour on an engine that keeps them).  Otherwise last_count and
// last_score_sum are 0 and last_score_max is -1 -- "not computed", which no score can be -- so that a reader of the struct
// (parse_stats, a peer rank behind an all-gather) cannot mistake the zeros for a measured mean of 0.
__global__ void __launch_bounds__(kBlock) stats_merge_kernel(const unsigned long long *partials, uint32_t n_partials,
                                                             StatsOut *out, bool last_known)
{
    __shared__ unsigned long long s_v[kBlock];
    const uint32_t f = blockIdx.x, tid = threadIdx.x;
    const bool is_max = f == 4u || f == 5u;
    unsigned long long v = 0;
    for (uint32_t q = tid; q < n_partials; q += kBlock) {
        const unsigned long long x = partials[f * kStatsBlocks + q];
        v = is_max ? (x > v ? x : v) : v + x;
    }
    s_v[tid] = v;
    __syncthreads();
    for (uint32_t off = kBlock / 2; off > 0; off >>= 1) {
        if (tid < off) {
            const unsigned long long a = s_v[tid], b = s_v[tid + off];
            s_v[tid] = is_max ? (b > a ? b : a) : a + b;
        }
        __syncthreads();
    }
    if (tid != 0)
        return;
    v = s_v[0];
    switch (f) {
    case 0: out->episodes = v; break;
    case 1: out->illegal_ends = v; break;
    case 2: out->last_count = v; break;
    case 3: out->last_score_sum = v; break;
    case 4: out->last_score_max = last_known ? static_cast<int>(v) : -1; break;
    case 5: out->max_exp = static_cast<unsigned int>(v); break;
    case 6: out->return_sum = static_cast<long long>(v); break;
    default: out->highest_hist[f - kStatsScalars] = static_cast<unsigned int>(v); break;
    }
}

// ------------------------------------------------------------- the returns summary in ONE launch
// What the once-per-rollout exchange of a multi-GPU job ships (first option): episodes, illegal_ends and the
// exact return_sum, from the episode slots and the live records -- the returns-only flavour of the reduction above, which
// as a kernel PAIR cost 11-13 us at 2^20 boards behind the last step of a rollout (two launches, a 2 048-column partials
// buffer in between).  Here: at most kSummaryBlocks blocks of 1 024 lanes (one per CU, sixteen wavefronts each), every
// wavefront walks whole 64-board groups = whole slots (slot through the scalar cache, the pending mask with it; four
// record loads in flight per lane), and the merge happens in the SAME launch by "last one out".
//
// Every word that crosses a block boundary travels by a device-scope read-modify-write and nothing else: a block PUBLISHES
// its three sums with atomic exchanges into its own 32-byte partial, waits for their return values (an RMW returns after it
// has been performed at the point of coherence -- the partial of a block on another XCD never sits in that XCD's L2), then
// counts itself in.  Counting is two-level -- one counter per group of kSummaryGroup blocks and one for the groups, each on
// its own cache line -- because a single hot line serialises at ~88 atomics/us on this chip (256 arrivals on one line:
// up to 2.9 us; 16: 0.2).  The block that finds itself last of the last group READS the partials back with RMWs as well
// (one lane per block, all in flight together), so there is no cache write-back or invalidate on the path at all (an
// agent-scope fence on gfx950 is a buffer_wbl2 + buffer_inv of the whole L2: the first version of this kernel, with three
// fence pairs and one lane walking the partials, took 15.8 us at 2^20 boards against 11.6 for the kernel pair,
// profiles/r06_a_stats_probe_*.txt).  The counters are left at zero for the next launch (calls on one engine are
// stream-ordered).
constexpr uint32_t kSummaryThreads = 1024, kSummaryGroup = 16;
constexpr uint32_t kSummaryGroups = kSummaryBlocks / kSummaryGroup;
constexpr uint32_t kSumCounterBase = kSummaryBlocks * 4;   // uint64 words: the block partials, then the counters (one 64-byte line each)
static_assert(kSumCounterBase + (kSummaryGroups + 1) * 8 <= kSummaryScratchWords, "summary scratch size");
static_assert(kSummaryBlocks % kSummaryGroup == 0 && kSummaryBlocks <= kSummaryThreads, "whole groups; one lane per partial");

// sum over the 16 cells of (e - 1) * 2^e as three instructions per cell: an empty cell contributes (0 - 1) << 0 = -1,
// put back by count_empty.  (g2048_device.h potential() keeps two sums: five instructions per cell.)
__device__ __forceinline__ uint32_t potential_lean(const Board &cells)
{
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const uint32_t e = (cells.r[i] >> (8 * l)) & 0xffu;
            acc += (e - 1u) << (e & 31u);
        }
    }
    return acc + count_empty(cells);
}

__device__ __forceinline__ uint32_t record_score_lean(const Board &raw)
{
    return (potential_lean(record_cells(raw)) - record_deficit(raw)) & kScoreMask;
}

__device__ __forceinline__ unsigned long long rmw_exchange(unsigned long long *p, unsigned long long v)
{
    return __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// a READ as a read-modify-write: `zero` is 0 in a form the compiler cannot see through (it would turn an add of a
// literal 0 into a plain sc1 load, which an L2 is allowed to serve)
__device__ __forceinline__ unsigned long long rmw_read(unsigned long long *p, unsigned long long zero)
{
    return __hip_atomic_fetch_add(p, zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the value has arrived (s_waitcnt) before anything behind this point is issued
__device__ __forceinline__ void wait_for(unsigned long long v) { asm volatile("" ::"v"(v) : "memory"); }

__global__ void __launch_bounds__(kSummaryThreads) returns_summary_kernel(const DeviceState st, uint32_t n, uint32_t n_waves,
                                                                          unsigned long long *scratch, StatsOut *out)
{
    constexpr uint32_t kWaves = kSummaryThreads / 64u, kUnroll = 4u;
    __shared__ unsigned long long s_part[kWaves][3];
    __shared__ uint32_t s_last;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave_in_block = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t wave_stride = gridDim.x * kWaves;
    unsigned long long episodes = 0, illegal = 0, gain = 0; // wave-uniform: the slots' counters
    unsigned long long live = 0;                            // this lane's share of the scores of the RUNNING episodes
    for (uint32_t w0 = blockIdx.x * kWaves + wave_in_block; w0 < n_waves; w0 += kUnroll * wave_stride) {
        Board rec[kUnroll];
        bool counts[kUnroll];
#pragma unroll
        for (uint32_t u = 0; u < kUnroll; ++u) {
            const uint32_t w = w0 + u * wave_stride;
            const bool have = w < n_waves;                  // uniform
            const uint32_t wc = have ? w : w0;
            const uint32_t *slot = slot_of(st.ep_counters, wc);
            const uint4 lo = *reinterpret_cast<const uint4 *>(slot), hi = *reinterpret_cast<const uint4 *>(slot + 4);
            if (have) {
                episodes += static_cast<unsigned long long>(hi.x) << 32 | lo.x;
                illegal += static_cast<unsigned long long>(hi.y) << 32 | lo.y;
                gain += static_cast<unsigned long long>(hi.z) << 32 | lo.z;
            }
            // a board whose episode has ended but which was not reset (auto_reset == 0) holds a FINISHED episode's score
            const unsigned long long pending = static_cast<unsigned long long>(hi.w) << 32 | lo.w;
            const uint64_t i = static_cast<uint64_t>(wc) * 64u + lane;
            counts[u] = have && i < n && ((pending >> lane) & 1ull) == 0ull;
            rec[u] = load_board_nt(st.boards, static_cast<uint32_t>(i < n ? i : n - 1u));
        }
#pragma unroll
        for (uint32_t u = 0; u < kUnroll; ++u)
            live += counts[u] ? record_score_lean(rec[u]) : 0u;
    }
    const unsigned long long live_wave = wave_sum64_lane63(live); // (every lane is here: the loop bounds are uniform)
    if (lane == 63u) {
        s_part[wave_in_block][0] = episodes;
        s_part[wave_in_block][1] = illegal;
        s_part[wave_in_block][2] = gain - live_wave;            // mod 2^64
    }
    __syncthreads();
    const uint32_t blocks = gridDim.x;
    if (tid == 0u) {
        unsigned long long ep = 0, ill = 0, ret = 0;
#pragma unroll
        for (uint32_t w = 0; w < kWaves; ++w) {
            ep += s_part[w][0];
            ill += s_part[w][1];
            ret += s_part[w][2];
        }
        uint32_t last = 0u;
        if (blocks == 1u) { // nothing to merge: wave 0 below reads these back from LDS
            s_part[0][0] = ep;
            s_part[0][1] = ill;
            s_part[0][2] = ret;
            last = 2u;
        } else {
            unsigned long long *mine = scratch + 4u * blockIdx.x;
            const unsigned long long r0 = rmw_exchange(mine + 0, ep), r1 = rmw_exchange(mine + 1, ill), r2 = rmw_exchange(mine + 2, ret);
            wait_for(r0 | r1 | r2);                             // published (performed at the point of coherence) ...
            const uint32_t group = blockIdx.x / kSummaryGroup, groups = (blocks + kSummaryGroup - 1u) / kSummaryGroup;
            const uint32_t members = min(kSummaryGroup, blocks - group * kSummaryGroup);
            unsigned int *group_count = reinterpret_cast<unsigned int *>(scratch + kSumCounterBase + 8u * group);
            unsigned int *top_count = reinterpret_cast<unsigned int *>(scratch + kSumCounterBase + 8u * kSummaryGroups);
            if (atomicAdd(group_count, 1u) == members - 1u) {   // ... before this block counts itself in
                atomicExch(group_count, 0u);                    // (nobody else touches it again in this launch)
                if (atomicAdd(top_count, 1u) == groups - 1u) {
                    atomicExch(top_count, 0u);
                    last = 1u;
                }
            }
        }
        s_last = last;
    }
    __syncthreads();
    const uint32_t last = s_last;                               // block-uniform
    if (last == 0u || tid >= 64u)
        return;
    // the last block out: lane b of wavefront 0 ... reads partial b, b + 64, ... (all RMWs in flight together)
    unsigned long long ep = 0, ill = 0, ret = 0;
    if (last == 2u) {
        if (lane == 0u) {
            ep = s_part[0][0];
            ill = s_part[0][1];
            ret = s_part[0][2];
        }
    } else {
        unsigned long long v[3 * (kSummaryBlocks / 64u)];
#pragma unroll
        for (uint32_t q = 0; q < kSummaryBlocks / 64u; ++q) {
            const uint32_t b = q * 64u + lane;
            unsigned long long *theirs = scratch + 4u * (b < blocks ? b : 0u);
#pragma unroll
            for (uint32_t f = 0; f < 3u; ++f)
                v[3u * q + f] = b < blocks ? rmw_read(theirs + f, last >> 8) : 0ull; // (last is 1 here)
        }
#pragma unroll
        for (uint32_t q = 0; q < kSummaryBlocks / 64u; ++q) {
            ep += v[3u * q + 0];
            ill += v[3u * q + 1];
            ret += v[3u * q + 2];
        }
    }
    ep = wave_sum64_lane63(ep);
    ill = wave_sum64_lane63(ill);
    ret = wave_sum64_lane63(ret); // (a sum mod 2^64: a block's gain - live may be "negative")
    if (lane != 63u)
        return;
    // the terminal records were not read and nothing was counted for the histogram: "not computed" (last_score_max = -1,
    // which no score can be), so that a reader of the struct cannot mistake the zeros for a measured mean of 0
    out->episodes = ep;
    out->illegal_ends = ill;
    out->last_count = 0;
    out->last_score_sum = 0;
    out->last_score_max = -1;
    out->max_exp = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k)
        out->highest_hist[k] = 0u;
    out->return_sum = static_cast<long long>(ret);
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
    const bool full = a.n % kBlock == 0;
    const StepTail tail{a.terminated, a.st.last_record, a.illegal, a.highest, a.terminal_boards, a.illegal_reward, a.max_exp,
                        a.auto_reset, a.obs, a.obs_dtype, a.boards_out, a.done_seq, a.done_value, a.action_err};
#define G2048_STEP_LAUNCH(ACT, FULL, STD, OBS)                                                                          \
    do {                                                                                                                \
        uint32_t pad = 0u;                                                                                              \
        if (OBS) {                                                                                                      \
            static std::atomic<signed char> pad_state[64] = {};                                                                      \
            pad = obs_pad_for(&step_kernel<ACT, FULL, STD, OBS>, pad_state);                                            \
        }                                                                                                               \
        hipLaunchKernelGGL((step_kernel<ACT, FULL, STD, OBS>), g, b, pad, s, a.st.boards,                               \
                           a.actions, a.st.ep_counters, a.board_offset, a.seed_lo, a.seed_hi, a.t_lo, a.t_hi, a.n,      \
                           a.reward, tail);                                                                             \
    } while (0)
#define G2048_STEP(ACT, FULL)                                                                                           \
    do {                                                                                                                \
        if (standard && a.obs)                                                                                          \
            G2048_STEP_LAUNCH(ACT, FULL, true, true);                                                                   \
        else if (standard)                                                                                              \
            G2048_STEP_LAUNCH(ACT, FULL, true, false);                                                                  \
        else if (a.obs)                                                                                                 \
            G2048_STEP_LAUNCH(ACT, FULL, false, true);                                                                  \
        else                                                                                                            \
            G2048_STEP_LAUNCH(ACT, FULL, false, false);                                                                 \
    } while (0)
    const bool standard = a.reward && a.terminated && !a.illegal && !a.highest && !a.terminal_boards && a.max_exp == 0 &&
                          !a.boards_out && !a.done_seq && !(a.action_err && action_dtype != 0);
    switch (action_dtype * 2 + (full ? 1 : 0)) {
    case 0: G2048_STEP(0, false); break;
    case 1: G2048_STEP(0, true); break;
    case 2: G2048_STEP(1, false); break;
    case 3: G2048_STEP(1, true); break;
    case 4: G2048_STEP(2, false); break;
    case 5: G2048_STEP(2, true); break;
    case 6: G2048_STEP(3, false); break;
    case 7: G2048_STEP(3, true); break;
#undef G2048_STEP
#undef G2048_STEP_LAUNCH
    default: return hipErrorInvalidValue;
    }
    if (a.done_seq && g.x > 1) // a one-block launch has published the word itself
        hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(64), 0, s, a.done_seq, a.done_value);
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

hipError_t launch_legal_mask(const uint4 *boards, uint32_t n, uint8_t *mask_out, hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(legal_mask_kernel, grid_for(n), dim3(kBlock), 0, s, boards, n, mask_out);
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

// ---- a k-step launch train as a CACHED hipGraph (small batches)
// At 65 536 boards a step kernel takes ~2.4 us on the device and one host thread issues a launch every 3.0-4.6 us: the
// per-step path is host-issue-bound (tools/ubench/r5_probe.hip: 3.0-4.6 us per step by stream launches, 2.43-2.45 us as a
// graph replay; 2^17 boards 2.9-3.6 vs 2.95-2.99; from 2^18 boards on the kernel is the longer of the two and the forms tie).
// The graph is k x step_graph_kernel, explicit nodes in a chain; building and instantiating it costs ~1.8 us per node (less
// than launching the same train once), replaying it one set_clock_kernel launch (the new clock value) + one hipGraphLaunch.
bool rollout_graph_supported(const StepArgs &a)
{
    return a.n != 0 && a.reward && a.terminated && !a.illegal && !a.highest && !a.terminal_boards && a.max_exp == 0 && !a.boards_out &&
           !a.done_seq && !a.obs && !a.st.rng && !a.action_err;
}

static void *step_graph_function(int action_dtype, bool full)
{
    switch (action_dtype * 2 + (full ? 1 : 0)) {
    case 0: return reinterpret_cast<void *>(step_graph_kernel<0, false>);
    case 1: return reinterpret_cast<void *>(step_graph_kernel<0, true>);
    case 2: return reinterpret_cast<void *>(step_graph_kernel<1, false>);
    case 3: return reinterpret_cast<void *>(step_graph_kernel<1, true>);
    case 4: return reinterpret_cast<void *>(step_graph_kernel<2, false>);
    case 5: return reinterpret_cast<void *>(step_graph_kernel<2, true>);
    case 6: return reinterpret_cast<void *>(step_graph_kernel<3, false>);
    case 7: return reinterpret_cast<void *>(step_graph_kernel<3, true>);
    default: return nullptr;
    }
}

void destroy_rollout_graph(RolloutGraph &g)
{
    if (g.exec)
        (void)hipGraphExecDestroy(g.exec);
    if (g.graph)
        (void)hipGraphDestroy(g.graph);
    g.exec = nullptr;
    g.graph = nullptr;
}

// `first`: the arguments of step 0 (its t is ignored); step j reads / writes its I/O j * stride elements further on.
hipError_t build_rollout_graph(const StepArgs &first, int action_dtype, uint32_t k_steps, uint64_t stride, unsigned long long *t_dev,
                               RolloutGraph *out)
{
    static const size_t act_bytes[4] = {0, 1, 4, 8};
    void *fn = step_graph_function(action_dtype, first.n % kBlock == 0);
    if (!fn || !rollout_graph_supported(first) || !t_dev || !out)
        return hipErrorInvalidValue;
    RolloutGraph g{};
    g.t_dev = t_dev;
    hipError_t err = hipGraphCreate(&g.graph, 0);
    if (err != hipSuccess)
        return err;
    hipGraphNode_t prev = nullptr;
    for (uint32_t j = 0; j < k_steps && err == hipSuccess; ++j) {
        const size_t off = static_cast<size_t>(j) * stride;
        uint4 *boards = first.st.boards;
        const void *actions = first.actions ? static_cast<const char *>(first.actions) + off * act_bytes[action_dtype] : nullptr;
        unsigned long long *ep = first.st.ep_counters;
        uint32_t board_offset = first.board_offset, seed_lo = first.seed_lo, seed_hi = first.seed_hi, n = first.n, jj = j;
        const unsigned long long *t_ptr = t_dev;
        float *reward = first.reward + off;
        StepTail tail{first.terminated + off, first.st.last_record, nullptr, nullptr, nullptr, first.illegal_reward, 0u, first.auto_reset,
                      nullptr, 0u, nullptr, nullptr, 0ull, nullptr};
        void *args[11] = {&boards, &actions, &ep, &board_offset, &seed_lo, &seed_hi, &t_ptr, &n, &jj, &reward, &tail};
        hipKernelNodeParams kp{};
        kp.func = fn;
        kp.gridDim = grid_for(first.n);
        kp.blockDim = dim3(kBlock);
        kp.kernelParams = args;
        hipGraphNode_t node = nullptr;
        err = hipGraphAddKernelNode(&node, g.graph, prev ? &prev : nullptr, prev ? 1 : 0, &kp);
        prev = node;
    }
    if (err == hipSuccess)
        err = hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0);
    if (err != hipSuccess) {
        destroy_rollout_graph(g);
        return err;
    }
    *out = g;
    return hipSuccess;
}

// Replay: step j of the train plays transaction t_first + j.  The clock is written by an ORDINARY launch in front of the
// graph -- its value is captured when it is enqueued, so two replays enqueued back to back each see their own (an executable
// graph's node parameters must not be rewritten while an earlier launch of it may still be waiting to run).
hipError_t launch_rollout_graph(RolloutGraph &g, unsigned long long t_first, hipStream_t s)
{
    hipLaunchKernelGGL(set_clock_kernel, dim3(1), dim3(64), 0, s, g.t_dev, t_first);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess)
        return err;
    return hipGraphLaunch(g.exec, s);
}

hipError_t launch_step_numpy(const StepArgs &a, int action_dtype, hipStream_t s)
{
    if (a.n == 0)
        return hipSuccess;
    const dim3 g((a.n + kNumpyBlock - 1u) / kNumpyBlock), b(kNumpyBlock);
    switch (action_dtype) {
    case 0: hipLaunchKernelGGL(step_numpy_kernel<0>, g, b, 0, s, a); break;
    case 1: hipLaunchKernelGGL(step_numpy_kernel<1>, g, b, 0, s, a); break;
    case 2: hipLaunchKernelGGL(step_numpy_kernel<2>, g, b, 0, s, a); break;
    case 3: hipLaunchKernelGGL(step_numpy_kernel<3>, g, b, 0, s, a); break;
    default: return hipErrorInvalidValue;
    }
    if (a.obs) // the observation of this mode comes from the stand-alone kernel (the block's resets run after its steps)
        return launch_onehot(a.st.boards, a.n, a.obs, static_cast<int>(a.obs_dtype), s);
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
    if (a.st.rng) { // numpy-RNG mode: the lanes of a wavefront drift in time (rollout_fused_numpy_kernel)
        switch (action_dtype) {
        case 0: hipLaunchKernelGGL(rollout_fused_numpy_kernel<0>, g, b, 0, s, a, stride); break;
        case 1: hipLaunchKernelGGL(rollout_fused_numpy_kernel<1>, g, b, 0, s, a, stride); break;
        case 2: hipLaunchKernelGGL(rollout_fused_numpy_kernel<2>, g, b, 0, s, a, stride); break;
        case 3: hipLaunchKernelGGL(rollout_fused_numpy_kernel<3>, g, b, 0, s, a, stride); break;
        default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
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

hipError_t launch_stats(const DeviceState &st, uint32_t n, unsigned long long *partials, unsigned long long *summary_scratch,
                        StatsOut *dev_out, bool returns_only, hipStream_t s)
{
    uint32_t blocks = grid_for(n).x;
    if (blocks > kStatsBlocks)
        blocks = kStatsBlocks;
    if (blocks == 0)
        blocks = 1; // n == 0: one block writes an all-zero partial
    if (returns_only) {
        // ONE launch: at most kSummaryBlocks blocks of sixteen wavefronts, merged by "last one out" (see the kernel)
        const uint32_t n_waves = (n + 63u) / 64u;
        uint32_t sblocks = (n_waves + kSummaryThreads / 64u - 1u) / (kSummaryThreads / 64u);
        sblocks = sblocks == 0u ? 1u : (sblocks > kSummaryBlocks ? kSummaryBlocks : sblocks);
        hipLaunchKernelGGL(returns_summary_kernel, dim3(sblocks), dim3(kSummaryThreads), 0, s, st, n, n_waves, summary_scratch, dev_out);
    } else {
        hipLaunchKernelGGL(stats_kernel, dim3(blocks), dim3(kBlock), 0, s, st, n, (n + 63u) / 64u, partials);
        hipLaunchKernelGGL(stats_merge_kernel, dim3(kStatsFields), dim3(kBlock), 0, s, partials, blocks, dev_out,
                           st.last_record != nullptr);
    }
    return hipGetLastError();
}

hipError_t launch_export_boards(const uint4 *records, uint32_t n, uint4 *cells_out, hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(export_boards_kernel, grid_for(n), dim3(kBlock), 0, s, records, n, cells_out);
    return hipGetLastError();
}

hipError_t launch_import_boards(uint4 *records, uint32_t n, const uint4 *cells_in, hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(import_boards_kernel, grid_for(n), dim3(kBlock), 0, s, records, n, cells_in);
    return hipGetLastError();
}

hipError_t launch_export_scores(const uint4 *records, uint32_t n, int32_t *scores_out, hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(export_scores_kernel, grid_for(n), dim3(kBlock), 0, s, records, n, scores_out);
    return hipGetLastError();
}

hipError_t launch_import_scores(uint4 *records, uint32_t n, const int32_t *scores_in, unsigned long long *ep_counters,
                                hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(import_scores_kernel, grid_for(n), dim3(kBlock), 0, s, records, n, scores_in, ep_counters);
    return hipGetLastError();
}

hipError_t launch_clear_stats(const DeviceState &st, uint32_t n, hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(clear_stats_kernel, grid_for(n), dim3(kBlock), 0, s, st, n);
    return hipGetLastError();
}

hipError_t launch_export_last_scores(const DeviceState &st, uint32_t n, int32_t *out, hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(export_last_scores_kernel, grid_for(n), dim3(kBlock), 0, s, st.last_record, n, out);
    return hipGetLastError();
}

// get_board + self.score of every board in one launch, for the host-resident path
__global__ void __launch_bounds__(kBlock) fetch_kernel(const uint4 *records, uint32_t n, uint4 *cells_out, int32_t *scores_out,
                                                       unsigned long long *done_seq, unsigned long long done_value)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) {
        const Board raw = load_board(records, i);
        store_board(cells_out, i, record_cells(raw));
        scores_out[i] = static_cast<int32_t>(record_score(raw));
    }
    if (done_seq)
        signal_done(done_seq, done_value);
}

hipError_t launch_fetch(const uint4 *records, uint32_t n, uint4 *cells_out, int32_t *scores_out, unsigned long long *done_seq,
                        unsigned long long done_value, hipStream_t s)
{
    const dim3 g = grid_for(n);
    hipLaunchKernelGGL(fetch_kernel, g, dim3(kBlock), 0, s, records, n, cells_out, scores_out, done_seq, done_value);
    if (done_seq && g.x > 1)
        hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(64), 0, s, done_seq, done_value);
    return hipGetLastError();
}

hipError_t launch_flag_set(unsigned long long *flag, unsigned long long value, hipStream_t s)
{
    hipLaunchKernelGGL(flag_set_kernel, dim3(1), dim3(64), 0, s, flag, value);
    return hipGetLastError();
}

hipError_t launch_flag_wait(const unsigned long long *flag, unsigned long long value, unsigned long long *timed_out,
                            uint32_t max_polls, hipStream_t s)
{
    hipLaunchKernelGGL(flag_wait_kernel, dim3(1), dim3(64), 0, s, flag, value, timed_out, max_polls);
    return hipGetLastError();
}

hipError_t launch_signal(unsigned long long *done_seq, unsigned long long done_value, hipStream_t s)
{
    hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(64), 0, s, done_seq, done_value);
    return hipGetLastError();
}

hipError_t launch_canonicalize(uint4 *boards, uint4 *next_boards, uint8_t *actions, uint32_t n, uint8_t *sym_out,
                               hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(canonicalize_kernel, grid_for(n), dim3(kBlock), 0, s, boards, next_boards, actions, n, sym_out);
    return hipGetLastError();
}

} // namespace g2048
