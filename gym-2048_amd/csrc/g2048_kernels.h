// g2048_kernels.h -- launch interface between the C ABI (g2048_api.hip) and the gfx950 kernels
// (g2048_kernels.hip).  Internal; the public contract is include/g2048.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace g2048 {

// Engine-owned device state (one slab).
struct DeviceState {
    uint4 *boards;      // [n] 16-byte board RECORDS: 16 x 5-bit exponents + the 24-bit score deficit in the spare
                        //     bits of bytes 8..15 (g2048_device.h "board RECORD"); there is no separate score
                        //     array -- self.score (game2048_env.py:86) = potential - deficit
    uint4 *last_record; // [n] record a board's most recent episode ENDED on (all-zero: none yet); its score is
                        //     that episode's return.  Written only by lanes whose episode ended.
    unsigned long long *ep_counters; // one 32-byte SLOT per 64 boards: finished episodes, of which ended on an illegal move,
                                     // G = summed merge scores (return accounting), pending mask -- as eight dwords, the
                                     // four low halves first; updated by ONE lane of the wavefront that owns the boards
                                     // (old values via the scalar cache); g2048_kernels.hip "episode SLOT"
    uint64_t *rng;      // numpy-RNG mode only: [5][n] planes (state_lo, state_hi, inc_lo, inc_hi, buf); else NULL
};

constexpr uint32_t kSlotWords = 4; // uint64 per wavefront slot of ep_counters

// bytes of the numpy-RNG allocation for n boards: 5 planes of uint64
inline size_t numpy_rng_bytes(uint64_t n) { return static_cast<size_t>(n * 40); }
// launch blocks never straddle the end of the slot array: slots exist for whole blocks of the LARGEST block size in use
// (the 512 lanes of step_numpy_kernel)
constexpr uint32_t kSlotBlockLanes = 512;

struct StepArgs {
    DeviceState st;
    const void *actions;
    float *reward;
    uint8_t *terminated;
    uint8_t *illegal;
    uint8_t *highest;
    uint4 *terminal_boards;
    uint4 *boards_out;  // [n][16] plain cells of the board AFTER the step (and its auto-reset), or NULL
    unsigned long long *done_seq; // host-visible completion word (mapped pinned memory) or NULL: a launch of ONE block writes
    unsigned long long done_value; //   done_value there after all of its outputs (system-scope release); see g2048_step_host
    unsigned long long *action_err; // strict actions (g2048_set_strict_actions): host-visible word that receives a report when a
                                    //   lane reads an action outside 0..3 (game2048_env.py:49 Discrete(4)); NULL = not checked
    void *obs;          // [n][16][4][4] one-hot observation of the board AFTER the step (and its auto-reset), or NULL
    uint32_t obs_dtype; // G2048_OBS_*
    uint32_t n;
    uint32_t board_offset;
    uint32_t seed_lo, seed_hi;
    uint32_t t_lo, t_hi;
    float illegal_reward;
    uint32_t max_exp;
    uint32_t auto_reset;
    uint32_t k_steps; // fused rollout only
};

struct StatsOut {
    unsigned long long episodes;      // all finished episodes since create / seed
    unsigned long long illegal_ends;
    unsigned long long last_count;    // boards that have finished at least one episode
    unsigned long long last_score_sum; // sum / max of those boards' most recent final scores
    int last_score_max;
    unsigned int max_exp;
    unsigned int highest_hist[32]; // boards whose highest tile is 2^k right now (game2048_env.py:190-192)
    long long return_sum;          // sum of the final scores of ALL finished episodes (exact)
};

hipError_t launch_reset(const StepArgs &a, uint32_t first_slot, const uint8_t *mask, hipStream_t s);
hipError_t launch_step(const StepArgs &a, int action_dtype, hipStream_t s);
hipError_t launch_rollout_random(const StepArgs &a, hipStream_t s);
// a k-step launch train as a cached hipGraph: k x step (t read from *t_dev, written by a launch in front); see g2048_kernels.hip
struct RolloutGraph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    unsigned long long *t_dev = nullptr;
};
bool rollout_graph_supported(const StepArgs &a); // the standard configuration: reward + terminated, nothing optional, spawn stream
hipError_t build_rollout_graph(const StepArgs &first, int action_dtype, uint32_t k_steps, uint64_t stride, unsigned long long *t_dev,
                               RolloutGraph *out);
hipError_t launch_rollout_graph(RolloutGraph &g, unsigned long long t_first, hipStream_t s);
void destroy_rollout_graph(RolloutGraph &g);
hipError_t launch_rollout_fused(const StepArgs &a, int action_dtype, uint64_t stride, hipStream_t s);
hipError_t launch_move(uint4 *boards, uint32_t n, const void *actions, int action_dtype, bool trial,
                       int32_t *score_out, uint8_t *legal_out, hipStream_t s);
hipError_t launch_query(const uint4 *boards, uint32_t n, uint32_t max_exp, uint8_t *isend_out, uint8_t *highest_out,
                        hipStream_t s);
hipError_t launch_legal_mask(const uint4 *boards, uint32_t n, uint8_t *mask_out, hipStream_t s);
hipError_t launch_add_tile(const StepArgs &a, uint32_t slot, hipStream_t s);
// the same three in numpy-RNG mode (a.st.rng != NULL)
hipError_t launch_seed_numpy(uint64_t *planes, uint32_t n, uint64_t first_seed, hipStream_t s);
hipError_t launch_reset_numpy(const StepArgs &a, const uint8_t *mask, hipStream_t s);
hipError_t launch_step_numpy(const StepArgs &a, int action_dtype, hipStream_t s);
hipError_t launch_add_tile_numpy(const StepArgs &a, hipStream_t s);
hipError_t launch_fill_actions(uint8_t *out, uint32_t n, uint32_t board_offset, uint32_t seed_lo, uint32_t seed_hi,
                               uint64_t t_first, uint32_t k_steps, hipStream_t s);
hipError_t launch_onehot(const uint4 *boards, uint32_t n, void *out, int obs_dtype, hipStream_t s);
hipError_t launch_augment(const uint4 *boards, const uint4 *next_boards, const uint8_t *actions, uint32_t n,
                          uint4 *boards_out, uint4 *next_out, uint8_t *actions_out, hipStream_t s);
// partials: kStatsPartialWords uint64 of device scratch (stage 1 -> stage 2; field-major, one column per block)
constexpr uint32_t kStatsBlocks = 2048;
constexpr uint32_t kStatsPartialWords = (7 + 32) * kStatsBlocks;
// summary_scratch: kSummaryScratchWords uint64, ZERO when first used (the one-launch returns summary keeps its block / group
// partials and its "last one out" counters there and leaves the counters at zero)
constexpr uint32_t kSummaryBlocks = 256;
constexpr uint32_t kSummaryScratchWords = 2048;
hipError_t launch_stats(const DeviceState &st, uint32_t n, unsigned long long *partials, unsigned long long *summary_scratch,
                        StatsOut *dev_out, bool returns_only, hipStream_t s);
// record <-> plain views (cells uint8[n][16], scores int32[n]); device pointers
hipError_t launch_export_boards(const uint4 *records, uint32_t n, uint4 *cells_out, hipStream_t s);
hipError_t launch_import_boards(uint4 *records, uint32_t n, const uint4 *cells_in, hipStream_t s);
hipError_t launch_export_scores(const uint4 *records, uint32_t n, int32_t *scores_out, hipStream_t s); // also last_record -> returns
hipError_t launch_import_scores(uint4 *records, uint32_t n, const int32_t *scores_in, unsigned long long *ep_counters,
                                hipStream_t s);
hipError_t launch_clear_stats(const DeviceState &st, uint32_t n, hipStream_t s);
hipError_t launch_export_last_scores(const DeviceState &st, uint32_t n, int32_t *out, hipStream_t s);
// host-resident I/O: boards + scores of the current state in one launch; the completion word (see StepArgs::done_seq)
hipError_t launch_fetch(const uint4 *records, uint32_t n, uint4 *cells_out, int32_t *scores_out, unsigned long long *done_seq,
                        unsigned long long done_value, hipStream_t s);
hipError_t launch_signal(unsigned long long *done_seq, unsigned long long done_value, hipStream_t s);
// stream-to-stream ordering by a ticket in device memory (two-chain rollouts): set behind the work it stands for, wait (bounded) before what depends on it
hipError_t launch_flag_set(unsigned long long *flag, unsigned long long value, hipStream_t s);
// `timed_out`: host-visible word that receives `value` when the wait gives up after `max_polls` polls (~1 us each)
hipError_t launch_flag_wait(const unsigned long long *flag, unsigned long long value, unsigned long long *timed_out,
                            uint32_t max_polls, hipStream_t s);
constexpr uint32_t kFlagWaitPolls = 1u << 26;
hipError_t launch_canonicalize(uint4 *boards, uint4 *next_boards, uint8_t *actions, uint32_t n, uint8_t *sym_out,
                               hipStream_t s);

} // namespace g2048
