/*
 * g2048.h -- C ABI of the MI355X-native batched 2048 environment (libg2048_hip.so).
 *
 * The reference (rgal/gym-2048) has no FFI layer: its boundary is the Gymnasium Python API of
 * Game2048Env (/root/reference/env/envs/game2048_env.py, cited below as game2048_env.py:LINE).
 * This header is the boundary a reference-side binding would call instead of running the Python
 * body of those methods; each entry point names the reference method it replaces.  INTEGRATION.md
 * shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *  - Every function returns 0 on success or a negative G2048_ERR_* code; the message of the last
 *    failure on the calling thread is g2048_last_error().  No C++ exception crosses the ABI.
 *  - The engine owns the per-board device state (boards, episodic score, episode bookkeeping).
 *    The CALLER owns every I/O buffer and passes raw device pointers (e.g. torch.Tensor.data_ptr());
 *    no allocation and no host synchronisation happens inside reset/step/rollout/onehot.
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All work is enqueued on
 *    it; results are ready when the stream reaches that point.
 *  - A board is 16 x uint8 exponents, row-major: 0 = empty, k = tile 2^k (reference: int64 tile
 *    values in a 4x4 ndarray, game2048_env.py:104).  Actions: 0 up, 1 right, 2 down, 3 left
 *    (game2048_env.py:196); only the low two bits of an action are used (g2048_set_strict_actions makes a value outside
 *    0..3 an error).
 *  - What the engine keeps per board in HBM is one 16-byte RECORD: bits [4:0] of byte j are the exponent
 *    of cell j, and the three spare bits [7:5] of bytes 8..15 hold the 24-bit "score deficit"
 *    d = (potential - score) mod 2^24, potential = sum over tiles of (e - 1) * 2^e, so that
 *    self.score (game2048_env.py:46,86,105) = potential - d.  A merge raises the potential by exactly
 *    what it scores, so a step only touches d when it spawns a 4 (d += 4), and a step reads and writes
 *    exactly 16 bytes per board -- there is no separate score array.  g2048_get/set_boards and
 *    g2048_get/set_scores convert to and from plain arrays; g2048_records_ptr exposes the records
 *    (cells = byte & 0x1f).  Scores are exact in 0 .. 2^24 - 1.
 *  - An engine is not thread-safe; use one engine per device per process.  All calls on one engine must also be
 *    STREAM-ORDERED: issue them on one stream, or order the streams with events.  The episode counters are
 *    updated without atomics (one wavefront per counter slot per launch), the statistics reduction has one
 *    scratch area per engine, and the get / set calls with host buffers share one staging buffer -- two calls in flight
 *    on different streams can corrupt each other's results.  (The all-gather has its own send buffer.)
 *  - Value ranges: cell exponents 0..30 (set_boards takes them mod 32; a merge of two 2^31 tiles would overflow the
 *    5-bit field into the deficit bits -- unreachable in play, where the largest tile is 2^17); scores
 *    0 .. 2^24 - 1 (host buffers are range-checked, device buffers are reduced mod 2^24 by the import kernel).
 *
 * Randomness: the "spawn stream" (oracle/g2048_oracle.h has the same text and is the checker)
 *    word(seed, t, board, slot) = Philox4x32-10(ctr = (t_lo, t_hi, board, slot >> 2),
 *                                               key = (seed_lo, seed_hi))[slot & 3]
 *  t     transaction counter: 0 for the reset after seeding, +1 for every step;
 *  board global board index = board_offset + local index (so results do not depend on how the
 *        batch is sharded over GPUs);
 *  slot  0 for the step's spawn, then the two spawns of a reset that follows in the same transaction.
 *  A spawn with word w on a board with n empty cells is read off the 64-bit product p = w * n: the new tile goes
 *  into the k-th empty cell (row-major), k = p >> 32, and is a 2 if (uint32_t)p <= 3865470566 else a 4
 *  (game2048_env.py:166-176: `random() < 0.9` with random() = frac(w * n / 2^32), the part of the word the position
 *  did not use -- P(2) within n / 2^32 < 4e-9 of the reference's 0.9, value and position independent to the same
 *  precision; ABI <= 13 compared the word's low 16 bits, P(2) = 0.900009).
 */
#ifndef G2048_H
#define G2048_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define G2048_OK 0
#define G2048_ERR_INVALID (-1) /* bad argument */
#define G2048_ERR_HIP (-2)     /* a HIP runtime call failed (includes "no GPU") */
#define G2048_ERR_NOMEM (-3)

/* dtype codes for actions */
#define G2048_ACT_RANDOM 0 /* no buffer: synthetic uniform policy, action = word(seed,t,board,3) >> 30 */
#define G2048_ACT_U8 1
#define G2048_ACT_I32 2
#define G2048_ACT_I64 3 /* what torch.argmax / multinomial produce */

/* dtype codes for the one-hot observation */
#define G2048_OBS_U8 0
#define G2048_OBS_F16 1
#define G2048_OBS_F32 2

typedef struct g2048_engine g2048_engine;

/* Per-step I/O buffers, all device pointers of n elements; NULL = not wanted.
 * Replaces the tuple returned by Game2048Env.step (game2048_env.py:100). */
typedef struct {
    const void *actions;     /* [n] of action_dtype; ignored for G2048_ACT_RANDOM */
    int32_t action_dtype;    /* G2048_ACT_* */
    float *reward;           /* [n] merge score of the move, or illegal_move_reward (:90,:95) */
    uint8_t *terminated;     /* [n] (:89,:94) */
    uint8_t *illegal;        /* [n] info['illegal_move'] (:79-81,:93) */
    uint8_t *highest;        /* [n] log2(info['highest']) (:97), of the board BEFORE an auto-reset */
    uint8_t *terminal_boards;/* [n][16] written only where terminated: the episode's last board */
    void *obs;               /* [n][16][4][4] of obs_dtype: stack(self.Matrix), the first element of the tuple step()
                              * returns (:100, :17-32), written by the step launch itself (no second kernel, no re-read of
                              * the records): the one-hot of the board AFTER the step -- with auto_reset, of the fresh board
                              * where the episode ended (the terminal one is terminal_boards), as a VecEnv returns it.
                              * 16-byte aligned.  NULL = not wanted */
    int32_t obs_dtype;       /* G2048_OBS_*; ignored when obs is NULL */
    uint8_t *boards_out;     /* [n][16] plain cell exponents of the board AFTER the step (self.Matrix, :104; after the
                              * auto-reset where it applies); 16-byte aligned.  NULL = not wanted */
} g2048_step_io;

/* Host-resident I/O for callers that hold HOST arrays -- the reference's own calling convention: one Python int in,
 * numpy out (game2048_env.py:76-100).  The engine owns one block of pinned, device-mapped, coherent host memory;
 * g2048_step_host's kernel reads the actions from it and writes every output straight into it across the bus, and
 * the call returns when they are visible (the host polls a completion word the device publishes last: no staging
 * copy, no stream-synchronisation call -- about 10 us per call for a handful of boards).  Pointers are HOST
 * addresses, valid until g2048_destroy. */
typedef struct {
    int64_t *actions;         /* [n] IN: fill before g2048_step_host (low two bits used) */
    float *reward;            /* [n] */
    uint8_t *terminated;      /* [n] */
    uint8_t *illegal;         /* [n] */
    uint8_t *highest;         /* [n] exponent of the highest tile (before an auto-reset) */
    uint8_t *boards;          /* [n][16] cells after the step / at g2048_fetch_host */
    uint8_t *terminal_boards; /* [n][16], rows valid where terminated */
    int32_t *scores;          /* [n] self.score: written by g2048_fetch_host (and by g2048_step_host in numpy-RNG mode) */
} g2048_host_io;

/* Episode statistics since create/seed. */
typedef struct {
    uint64_t episodes;       /* finished episodes, all of them */
    uint64_t illegal_ends;   /* ... of which ended on an illegal move (game2048_env.py:91-95) */
    uint64_t last_count;     /* boards that have finished at least one episode */
    int64_t last_score_sum;  /* sum over those boards of the final merge score (game2048_env.py:86) of their MOST
                              * RECENT finished episode (the engine keeps one terminal record per board) */
    int32_t last_score_max;  /* best of those; -1 = NOT COMPUTED: last_count / last_score_sum / last_score_max are only filled by
                              * g2048_episode_stats[_async] on an engine that keeps terminal records (g2048_set_last_records);
                              * g2048_returns_summary_async never reads them (ABI <= 13 wrote zeros there) */
    uint32_t max_exp;        /* highest exponent currently on any board */
    uint32_t highest_hist[32]; /* highest_hist[k] = boards whose highest tile (game2048_env.py:190-192) is 2^k
                                * right now (k = 0: empty board) -- what ppo_train.py:77-81 tallies */
    int64_t return_sum;      /* EXACT sum of the final merge scores (game2048_env.py:86) of ALL `episodes` finished
                              * episodes -- what SB3's Monitor averages as info["episode"]["r"] (ppo_train.py:123), minus
                              * the illegal-move rewards: mean episodic return = (return_sum + illegal_ends *
                              * illegal_move_reward) / episodes.  Kept by conservation, with no per-episode work in the
                              * step: every wavefront adds the merge scores of its 64 moves to a running total G, and
                              * return_sum = sum of G - sum of the scores of the episodes still running.  An episode
                              * counts from the step that ends it (with auto_reset == 0: although its score is still in
                              * the record); an episode ABANDONED by g2048_reset before it ended is in neither `episodes`
                              * nor return_sum; g2048_set_scores moves a running episode's score without touching the
                              * finished ones.  (Stepping a board again after its episode has ended, without a reset,
                              * continues that episode: it is counted once more when it ends again, its score once.) */
} g2048_stats;

typedef struct g2048_comm g2048_comm;
#define G2048_COMM_ID_BYTES 128

const char *g2048_last_error(void);
/* Library / ABI version, bumped on any signature change. */
int g2048_abi_version(void);

/* Game2048Env.__init__ (game2048_env.py:38-58) for n_boards boards on HIP device `device`.
 * Boards are all-empty until g2048_reset.  board_offset = global index of local board 0.
 * 1 <= n_boards <= 2^32 - 256 and board_offset + n_boards <= 2^32. */
int g2048_create(uint64_t n_boards, int device, uint64_t seed, uint64_t board_offset, g2048_engine **out);
int g2048_destroy(g2048_engine *e);

/* gym.Env.reset(seed=...) seeding half (game2048_env.py:103): restart the spawn stream (t = 0) and
 * clear the episode statistics (terminal records, episode counters) with a kernel enqueued on `stream`
 * (ordered against steps already enqueued there; no host synchronisation). */
int g2048_seed(g2048_engine *e, uint64_t seed, void *stream);
int g2048_get_clock(const g2048_engine *e, uint64_t *t);
int g2048_set_clock(g2048_engine *e, uint64_t t);
uint64_t g2048_num_boards(const g2048_engine *e);

/* set_illegal_move_reward (game2048_env.py:61-67) / set_max_tile (:69-73; max_exp = log2(max_tile),
 * 0 = None). */
int g2048_set_illegal_move_reward(g2048_engine *e, float reward);
int g2048_set_max_tile(g2048_engine *e, int max_exp);

/* Strict actions.  The reference declares Discrete(4) (game2048_env.py:49) but never checks: `int(direction / 2)` and
 * `direction % 2` (:210-212) happen to play 4 as "down" and -1 as "right".  This library plays the low two bits of whatever
 * it is given (4 -> "up") -- a different accident -- unless strict actions are ON: then every g2048_step / g2048_rollout /
 * g2048_rollout_fused / g2048_step_host launch whose action buffer (G2048_ACT_U8 / I32 / I64) holds a value outside 0..3 still
 * plays the low two bits but reports it to a per-engine error word in pinned host memory, and the NEXT call on the engine that
 * starts after that launch has completed returns G2048_ERR_INVALID (naming one offending board), does nothing else, and clears
 * the report.  Costs one compare per lane, and the launches use the general step kernel instead of the reward + terminated
 * specialisation (and no cached graph): nothing measurable at 2^20 boards (9.60 against 9.56 us per launch,
 * profiles/r06_i_strict_probe.txt); default OFF, as the reference.  G2048_ACT_RANDOM is never checked.
 * g2048_step_host reads its actions from host memory and refuses BEFORE stepping anything. */
int g2048_set_strict_actions(g2048_engine *e, int enable);
int g2048_get_strict_actions(const g2048_engine *e);

/* Game2048Env.reset (game2048_env.py:102-111) for every board: zero board, score 0, two spawns.
 * new_transaction != 0: t += 1 first (a reset that is not the first use of the stream);
 * first_slot: slot of the first spawn (0 unless continuing a transaction that already spawned).
 * mask: optional device uint8[n]; when non-NULL only boards with mask != 0 are reset. */
int g2048_reset(g2048_engine *e, int new_transaction, uint32_t first_slot, const uint8_t *mask, void *stream);

/* Game2048Env.step (game2048_env.py:76-100) for every board: t += 1, move, score, spawn, done.
 * auto_reset != 0 additionally performs the caller's `if terminated: env.reset()` in the same
 * launch (slots 1,2 after a legal move, 0,1 after an illegal one). */
int g2048_step(g2048_engine *e, const g2048_step_io *io, int auto_reset, void *stream);

/* The host-resident forms (see g2048_host_io).  g2048_host_io_map allocates the block on first use and returns its
 * arrays.  g2048_step_host = g2048_step with actions from / all outputs to those arrays (int64 actions, reward,
 * terminated, illegal, highest, terminal_boards, boards); it BLOCKS until the outputs are in host memory.
 * g2048_fetch_host brings the current boards and scores there (after reset / set_boards / add_tile / move). */
int g2048_host_io_map(g2048_engine *e, g2048_host_io *out);
int g2048_step_host(g2048_engine *e, int auto_reset, void *stream);
int g2048_fetch_host(g2048_engine *e, void *stream);

/* Host-visible completion WITHOUT a runtime synchronisation call -- the mechanism behind g2048_step_host, exported for
 * callers that bracket their own launch trains (a Gymnasium step() returns to the host after every call,
 * game2048_env.py:100; a rollout loop returns once per rollout).  g2048_stream_signal enqueues on `stream` a one-wave
 * kernel that publishes a fresh ticket (monotonically increasing per engine) to the engine's completion word -- 8 bytes
 * of pinned, device-mapped, coherent host memory -- with a system-scope release, i.e. after everything enqueued on
 * `stream` before it has completed and is visible to the host.  g2048_stream_wait spins on that word until it reaches
 * `ticket`: about 5 us less than hipStreamSynchronize per call on MI355X (tools/ubench/host_latency.hip,
 * tail_probe.hip).  The wait checks the stream's health every ~2 s and gives up after G2048_WAIT_TIMEOUT_S (default
 * 120) seconds; neither call may be used on a capturing stream.  Signals of one engine must be stream-ordered. */
int g2048_stream_signal(g2048_engine *e, void *stream, uint64_t *ticket);
int g2048_stream_wait(g2048_engine *e, uint64_t ticket, void *stream);

/* k consecutive g2048_step launches without returning to the caller.  Buffers of step j are the
 * io pointers advanced by j * stride elements (stride = 0 reuses the same buffers, stride = n
 * walks [k][n] rollout buffers; an element of obs is one board's whole [16][4][4] observation). */
int g2048_rollout(g2048_engine *e, uint32_t k_steps, const g2048_step_io *io, uint64_t stride, int auto_reset,
                  void *stream);

/* Two-chain rollouts.  With chains = 2, g2048_rollout cuts the batch in two at a block boundary and runs the halves as
 * two independent chains of launches: the lower half on the caller's stream, issued by the calling thread, the upper
 * half on the device's side stream (highest priority; one per device and process, shared by its engines, created by the
 * first such call), issued by that side chain's own launch thread.  Boards are
 * independent (no cross-board state anywhere in game2048_env.py), so the results are bit-identical to chains = 1; what
 * changes is that two hardware queues always have a kernel waiting, so the head of one half-batch kernel overlaps the
 * tail of the other's: 9.4 -> 8.2 us per step at 2^20 boards, 5.6 -> 4.8 at 2^19, nothing below 2^19 or at 2^24
 * (tools/ubench/overlap.hip).  The side stream is forked from and joined back into `stream` inside every
 * g2048_rollout call -- by one-wave kernels that publish / await a ticket in device memory, which cost this runtime
 * ~17 us less latency per rollout than an event record + hipStreamWaitEvent pair (G2048_CHAIN_SYNC=events selects
 * those) -- so callers see ordinary stream order.  Applies to g2048_rollout in spawn-stream mode with at least 512
 * boards, when the rollout is long enough to pay: from 12 steps while the device's side chain is WARM (it had work
 * within the last ~50 ms: two chains cost ~6 us per rollout and save ~1.2 us per step at 2^20 boards), from 64 steps when
 * it is COLD (a stream that has idled for a few hundred milliseconds starts its first kernels ~40 us late).  Everything
 * else (g2048_step, shorter rollouts, numpy-RNG mode, a capturing stream, a caller's stream created at the device's
 * highest priority -- the side stream's own: two streams of one priority may be given one hardware queue, and the ticket
 * kernels need two) runs as one chain; g2048_get_chains_used tells
 * what the most recent g2048_rollout did.  Engines of one device share the side chain: their rollouts take turns on it.
 * Default: 1 -- the form is OPT-IN: it starts a launch thread per device (which spins for G2048_SIDE_SPIN_US microseconds,
 * default 200, after a job before it sleeps; rollouts of fewer than 6 steps leave it asleep) and can only ever fire for a
 * single g2048_rollout of >= 12 steps whose actions are all supplied up front -- never for a caller that steps one action
 * at a time.  The knobs (G2048_TWO_CHAIN_MIN_STEPS, G2048_CHAIN_SYNC, G2048_CHAIN_ANY_PRIORITY, G2048_FLAG_WAIT_POLLS) are
 * read by THIS call, once.  Failure behaviour: a ticket wait is bounded (2^26 polls of ~1 us); one that runs out reports
 * through pinned host memory and EVERY later call on the engine returns G2048_ERR_HIP (its chains ran unordered); a
 * rollout whose launches fail half-way still enqueues its join and leaves the engine refusing further calls.  A profiler
 * that serialises kernels across queues (rocprofv3 --pmc) makes the tickets wait for each other: use one chain there. */
int g2048_set_chains(g2048_engine *e, int chains);
int g2048_get_chains(const g2048_engine *e);
int g2048_get_chains_used(const g2048_engine *e);
/* Small batches (up to 2^17 boards), standard outputs (reward + terminated only), spawn-stream mode: the SECOND and every
 * further g2048_rollout in a row over the same buffers (same k_steps, pointers, stride, dtype, auto_reset) is replayed from
 * a cached hipGraph of its launch train instead of being launched kernel by kernel -- at that size one host thread cannot
 * issue launches as fast as the device retires them (65 536 boards: 2.45 us per step replayed, 3.0-4.6 us launched).  Same
 * kernels, bit-identical results; nothing to configure (G2048_ROLLOUT_GRAPH=0 in the environment of g2048_create turns
 * it off).  Returns how many rollouts of this engine were served that way. */
uint64_t g2048_get_graph_replays(const g2048_engine *e);
/* "" while the form is available; otherwise why it is off for this engine (G2048_ROLLOUT_GRAPH=0, or the HIP call that failed:
 * a failing graph build / launch turns the form off for the engine's lifetime and its rollouts are launched kernel by kernel). */
const char *g2048_graph_status(const g2048_engine *e);
/* Plan preparation: build the cached graph for exactly this rollout NOW (no step is executed, nothing is enqueued), so that
 * already the first g2048_rollout with these arguments is a replay.  A no-op where the form does not apply.  No stream
 * operation is involved (no allocation, fill or synchronisation: the graph's clock word is part of the engine), so it may be
 * called while the application has a stream capture open; a fifth set of buffers replaces the oldest cached graph only once
 * the stream that one was last replayed on has drained (otherwise that rollout simply is not cached). */
int g2048_rollout_prepare(g2048_engine *e, uint32_t k_steps, const g2048_step_io *io, uint64_t stride, int auto_reset);

/* The same k steps as g2048_rollout -- same actions in, bit-identical reward / terminated / illegal /
 * highest out -- in ONE launch with the boards held in registers (6 B of traffic per env-step instead
 * of 46).  For action sequences that are known in advance (replays, scripted or tree-search rollouts);
 * a policy that needs every observation uses g2048_step / g2048_rollout.  terminal_boards, boards_out and obs
 * must be NULL.  In numpy-RNG mode the generators stay in registers as well, and the lanes of a wavefront drift in time:
 * every trip of the kernel's loop is one lockstep spawn that each lane spends on its own next step or on a tile of its own
 * reset (a reset costs a lane one or two trips instead of costing every wavefront two spawns per step). */
int g2048_rollout_fused(g2048_engine *e, uint32_t k_steps, const g2048_step_io *io, uint64_t stride, int auto_reset,
                        void *stream);

/* ONE launch that plays k steps of the synthetic random policy with boards held in registers
 * (auto-reset on).  Writes only the final state and the episode bookkeeping. */
int g2048_rollout_random(g2048_engine *e, uint32_t k_steps, void *stream);

/* Game2048Env.move alone (game2048_env.py:194-241) for every board: no spawn, no score update, no
 * clock tick.  score_out int32[n] = merge score (0 where illegal); legal_out uint8[n] = 0 where the
 * reference raises IllegalMove (:238-239).  trial != 0 leaves the boards untouched (:224,:236).
 * action_dtype must name a buffer (G2048_ACT_U8/I32/I64). */
int g2048_move(g2048_engine *e, const void *actions, int32_t action_dtype, int trial, int32_t *score_out,
               uint8_t *legal_out, void *stream);

/* Game2048Env.isend (game2048_env.py:262-280, uses the engine's max_tile) and highest (:190-192, as
 * an exponent) for every board; either output may be NULL. */
int g2048_query(const g2048_engine *e, uint8_t *isend_out, uint8_t *highest_out, void *stream);

/* The four trial moves of Game2048Env.isend (game2048_env.py:273-280: `self.move(direction, trial=True)` for
 * direction 0..3, catching IllegalMove) for every board, kept as a mask instead of collapsed to one flag:
 * mask_out uint8[n], bit d set = move d is legal on that board (0 = no move left).  What a legality-aware
 * policy masks its logits with; one launch instead of four g2048_move(trial) calls.  Boards are not modified. */
int g2048_legal_actions(const g2048_engine *e, uint8_t *mask_out, void *stream);

/* Game2048Env.add_tile (game2048_env.py:166-176): one spawn from slot `slot` of the current
 * transaction on every board that has an empty cell. */
int g2048_add_tile(g2048_engine *e, uint32_t slot, void *stream);

/* Fill out[j][i] (uint8, j < k_steps, i < n) with the synthetic policy's action of transaction
 * t_first + j for board i -- the same values G2048_ACT_RANDOM would use at those transactions. */
int g2048_fill_random_actions(const g2048_engine *e, uint64_t t_first, uint32_t k_steps, uint8_t *out, void *stream);

/* stack() (game2048_env.py:17-32): out[n][16][4][4] of obs_dtype, channel c = (exponent == c). */
int g2048_onehot(const g2048_engine *e, void *out, int32_t obs_dtype, void *stream);

/* get_board / set_board (game2048_env.py:282-288) for the whole batch; `buf` is [n][16] uint8
 * exponents in host or device memory (device buffers 16-byte aligned).  A conversion kernel runs
 * between the records and `buf` (through an engine-owned staging buffer when `buf` is host memory).
 * Host buffers: the copy is synchronous.  Device buffers: everything is enqueued on `stream`, no host
 * synchronisation.  set_boards does not touch the scores; exponents are taken mod 32. */
int g2048_get_boards(const g2048_engine *e, uint8_t *buf, void *stream);
int g2048_set_boards(g2048_engine *e, const uint8_t *buf, void *stream);

/* Episodic merge score self.score (game2048_env.py:46,86,105): int32[n], host or device; computed
 * from / folded into the records by a kernel.  Scores must lie in 0 .. 2^24 - 1 (host buffers are
 * checked, device buffers are taken mod 2^24). */
int g2048_get_scores(const g2048_engine *e, int32_t *buf, void *stream);
int g2048_set_scores(g2048_engine *e, const int32_t *buf, void *stream);

/* Final merge score of each board's most recently finished episode (0 before the first one):
 * int32[n], host or device.  A step that ends an episode stores the 16-byte terminal record; this call
 * turns the records into scores (potential - deficit) with a kernel. */
int g2048_get_last_scores(const g2048_engine *e, int32_t *buf, void *stream);

/* Per-board "last finished episode" bookkeeping on / off (default: on).  On: a step that ends an episode stores the
 * board's 16-byte terminal record (one sparse store per finished episode: 0.85 us of a 10.2 us launch at 2^20 boards
 * under a random policy), which is what g2048_get_last_scores, g2048_last_records_ptr, g2048_allgather_returns and
 * the last_* members of g2048_stats read.  Off: those calls fail with G2048_ERR_INVALID (the pointer is NULL; last_count and
 * last_score_sum are 0 and last_score_max is -1 = NOT COMPUTED, see g2048_stats); episodes, illegal_ends and the exact return_sum -- everything the once-per-rollout exchange of a multi-GPU job
 * needs -- do not depend on it.  Switching it on again clears the stored records (enqueued on `stream`).
 * No counterpart in the reference, whose env keeps no episode history at all (SB3's Monitor does, on the host). */
int g2048_set_last_records(g2048_engine *e, int enable, void *stream);
int g2048_get_last_records(const g2048_engine *e);

/* Raw device pointers of the engine-owned state for zero-copy views: the board RECORDS
 * (uint8[n][16]; cell = byte & 0x1f, see "RECORD" above) and the terminal records of the most recent
 * finished episodes (same format, all-zero = none yet).  Valid until g2048_destroy. */
void *g2048_records_ptr(const g2048_engine *e);
void *g2048_last_records_ptr(const g2048_engine *e);

/* Reduce the episode bookkeeping on the device and copy the result to *out (synchronises `stream`). */
int g2048_episode_stats(const g2048_engine *e, g2048_stats *out, void *stream);
/* The same reduction written to a g2048_stats in DEVICE memory, enqueued on `stream`, no host
 * synchronisation: the per-rank episodic-return summary a multi-GPU job all-gathers. */
int g2048_episode_stats_async(const g2048_engine *e, g2048_stats *device_out, void *stream);
/* The RETURNS-ONLY form of the same call -- what the once-per-rollout exchange of a multi-GPU job needs and nothing
 * else: episodes, illegal_ends and the exact return_sum, from the episode slots and the live records.  The terminal
 * records are not read and the histogram is not counted: last_score_max is written as -1 = NOT COMPUTED (no score can be
 * negative; check it before dividing last_score_sum by last_count), last_count, last_score_sum, max_exp and highest_hist[]
 * as zero.  ONE launch (at most 256 blocks of 1 024 lanes, merged inside the launch by "last block out"): 7.4 us per call
 * at 2^20 boards (6.5 us behind a launch train) against 21 us for the full reduction (profiles/r06_b_stats_probe_one_launch.txt). */
int g2048_returns_summary_async(const g2048_engine *e, g2048_stats *device_out, void *stream);

/* numpy-compatible RNG mode: every board draws from its OWN numpy PCG64 exactly as the reference does
 * through gymnasium's np_random (game2048_env.py:103,168,170: random() < 0.9, then Generator.shuffle of
 * the 16 positions, first empty one).  `planes` = uint64[5][n] in host or device memory: state_lo,
 * state_hi, inc_lo, inc_hi of numpy's PCG64(SeedSequence(seed_i)).state and buf = uinteger |
 * has_uint32 << 32 (the host computes them with numpy; SB3 seeds env i with seed + i).  After this call
 * reset/step/rollout/add_tile consume those generators; board i then plays bit-for-bit the game of the
 * unmodified reference env after reset(seed = seed_i).  planes == NULL returns to the spawn stream.
 * About 5x slower than the spawn stream (a spawn is ~1 000 instructions: a 128-bit LCG and a 15-step shuffle). */
int g2048_set_numpy_rng(g2048_engine *e, const uint64_t *planes, void *stream);
int g2048_get_numpy_rng(const g2048_engine *e, uint64_t *planes, void *stream);
/* The same mode, seeded on the device: board i <- numpy PCG64(SeedSequence(base_seed + board_offset + i))
 * (SeedSequence hashing + PCG64 seeding restated in g2048_pcg64.h, pinned against numpy).  Also does
 * what g2048_seed does (t = 0, statistics cleared). */
int g2048_seed_numpy(g2048_engine *e, uint64_t base_seed, void *stream);

/* training_data.augment() (training_data.py:257-299) on the current device: n transitions
 * (boards[n][16], optional next_boards[n][16], actions uint8[n]; device pointers, 16-byte aligned) ->
 * the eight symmetries, 8n rows in the reference's order [orig, hflip, rot1(orig), rot1(hflip), rot2(..),
 * rot2(..), rot3(..), rot3(..)], actions remapped (hflip swaps 1<->3; a clockwise quarter turn adds 1 mod 4).
 * next_boards / next_out may both be NULL.  Needs no engine. */
int g2048_augment(const uint8_t *boards, const uint8_t *next_boards, const uint8_t *actions, uint64_t n,
                  uint8_t *boards_out, uint8_t *next_out, uint8_t *actions_out, void *stream);

/* Checkpoint / resume of the complete engine state (boards, scores, episode bookkeeping, seed,
 * clock, configuration) as one host blob of g2048_state_bytes() bytes.  The reference checkpoints
 * only models; its env state hooks are get_board/set_board (game2048_env.py:282-288). */
uint64_t g2048_state_bytes(const g2048_engine *e);
int g2048_get_state(const g2048_engine *e, void *host_buf, void *stream);
/* blob_bytes = size of host_buf; magic, board count, size and header fields are validated.  Blobs written before ABI 14
 * are refused (their games were played under the previous spawn rule). */
int g2048_set_state(g2048_engine *e, const void *host_buf, uint64_t blob_bytes, void *stream);

/* Symmetric-board canonicalisation (the counterpart of g2048_augment): every board (plain uint8[n][16]
 * exponents, device pointer, 16-byte aligned) is replaced IN PLACE by the lexicographically smallest
 * (row-major bytes) of its eight symmetries -- the ones training_data.py:257-299 generates, same index
 * order: variant = 2 * clockwise_quarter_turns + hflip; ties keep the lowest variant.  When non-NULL,
 * next_boards[n][16] gets the same symmetry, actions[n] (uint8) are remapped (hflip swaps 1<->3, a
 * clockwise quarter turn adds 1 mod 4) and symmetry_out[n] receives the variant applied.  Needs no engine. */
int g2048_canonicalize(uint8_t *boards, uint8_t *next_boards, uint8_t *actions, uint64_t n, uint8_t *symmetry_out,
                       void *stream);

/* ---- the collective: all-gather of episodic returns, RCCL over xGMI (no counterpart in the reference,
 * which has no multi-device code).  RCCL is loaded on the first g2048_comm_* call.
 *
 * One process per GPU: rank 0 calls g2048_comm_unique_id and hands the 128 bytes to the other ranks by
 * any means (torch.distributed.broadcast, MPI, a file); every rank then calls g2048_comm_create
 * (ncclCommInitRank).  g2048_allgather_returns enqueues, on `stream`, the kernel that turns the engine's
 * terminal records into returns (int32[n], n equal on all ranks) and ONE ncclAllGather of them into
 * out[world * n] (device memory). */
int g2048_comm_unique_id(uint8_t id[G2048_COMM_ID_BYTES]);
int g2048_comm_create(int world, int rank, const uint8_t id[G2048_COMM_ID_BYTES], int device, g2048_comm **out);
int g2048_comm_destroy(g2048_comm *c);
int g2048_allgather_returns(const g2048_engine *e, g2048_comm *c, int32_t *out, void *stream);
/* The once-per-rollout exchange of a multi-GPU job in ONE call (first option): the returns-only summary of
 * this engine's shard (g2048_returns_summary_async: one launch), written straight into row `rank` of out[world] (device
 * memory), and ONE in-place ncclAllGather of the sizeof(g2048_stats)-byte rows -- both enqueued on `stream`, behind the
 * rollout's last step: no send buffer, no hop to a communication stream and back, no host synchronisation (a framework's
 * process-group all-gather of the same bytes costs two event hand-overs more: 22 us per rollout against 11-12 us with a one-rank
 * group on one MI355X, and 26 us in round 5 with the summary as a kernel pair, profiles/r06_b_forced_dist_ab.txt).  out[r] is rank r's summary once `stream` reaches that point, on every rank. */
int g2048_allgather_summary(const g2048_engine *e, g2048_comm *c, g2048_stats *out, void *stream);
int g2048_comm_world(const g2048_comm *c);
int g2048_comm_rank(const g2048_comm *c);
/* One process driving several GPUs.  g2048_comm_local_create builds the communicator set ONCE (ncclCommInitAll over
 * `devices`, distinct; hundreds of milliseconds on 8 GPUs -- not something to pay per rollout) and
 * g2048_allgather_returns_local reuses it: engines[r] must live on devices[r] and hold equal board counts; every
 * engine's last returns are all-gathered into outs[r][n_devices * n] (device memory on devices[r]).  The export
 * kernels and the grouped ncclAllGather are ENQUEUED on streams[r] (NULL array = null streams); nothing is
 * synchronised: outs[r] is complete when streams[r] reaches that point. */
typedef struct g2048_comm_local g2048_comm_local;
#define G2048_COMM_LOCAL_MAX 64
int g2048_comm_local_create(const int *devices, int n_devices, g2048_comm_local **out);
int g2048_comm_local_destroy(g2048_comm_local *c);
int g2048_allgather_returns_local(g2048_comm_local *c, g2048_engine *const *engines, int32_t *const *outs,
                                  void *const *streams);

#ifdef __cplusplus
}
#endif
#endif
