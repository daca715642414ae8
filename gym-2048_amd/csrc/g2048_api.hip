// g2048_api.hip -- the C ABI of include/g2048.h on top of the gfx950 kernels.
// Host-side only: argument checking, the engine object, the transaction clock, launches.
#include "../../include/g2048.h"

#include "g2048_kernels.h"
#include "g2048_side_launcher.h"

#include <rccl/rccl.h> // declarations only: librccl is dlopen()ed by the first g2048_comm_* call

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <functional>
#include <mutex>
#include <new>
#include <thread>

namespace {

thread_local char g_error[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
    return code;
}

#define G2048_HIP(call)                                                                                        \
    do {                                                                                                       \
        hipError_t err_ = (call);                                                                              \
        if (err_ != hipSuccess)                                                                                \
            return fail(G2048_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(err_));                       \
    } while (0)

using g2048::SideLauncher;
using g2048::cpu_relax;

// ONE side chain per device and process, shared by every engine on that device that runs two chains: the side stream
// (highest priority), its launch thread, and the time its last work is expected to end.  Shared so that it stays WARM:
// a stream that has idled for a few hundred milliseconds starts its next kernels late (a forced two-chain 20-step
// rollout takes 175 us after 0.2 ms of idle, 184-186 after 5-50 ms, 223 after 300 ms -- one chain: 195-209;
// tools/chain_gap_probe.py), and e.g. a warm-up on one engine should leave the chain ready for the next engine.  `use`
// serialises the engines' rollouts on it (an engine itself is single-threaded by contract).
struct SideChain {
    SideLauncher launcher;
    hipStream_t stream = nullptr;
    std::mutex use;
    std::atomic<int64_t> busy_until_ns{0}; // steady-clock time the side stream's queued work is expected to have ended + the warm window
    int device = 0, refs = 0;
    int priority = 0; // of `stream`: the highest the device offers
};

std::mutex g_side_mutex;
SideChain *g_side[64] = {};

int64_t steady_now_ns()
{
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// "\0G2048v5": records carry the score, 4-word episode slots for whole 512-lane blocks, the summary scratch behind them.
// v4 (ABI 14) is the same games in a slab of another size; v3 and older were played under the spawn rule before ABI 14
// (g2048.h "Randomness"): a game saved under the old rule would continue differently under the new one.  Both are refused
// with the reason instead of silently resumed.
constexpr uint64_t kStateMagic = 0x3576383430324700ull;

} // namespace

struct g2048_engine {
    uint64_t n = 0;
    int device = 0;
    uint64_t seed = 0;
    uint64_t board_offset = 0;
    uint64_t t = 0;       // transaction counter
    int fresh = 1;        // nothing consumed from the stream since seeding
    float illegal_reward = 0.0f; // game2048_env.py:53
    uint32_t max_exp = 0;        // game2048_env.py:54 (0 = None)
    void *slab = nullptr;
    size_t slab_bytes = 0;
    g2048::DeviceState st{};
    g2048::StatsOut *stats_dev = nullptr;
    unsigned long long *stats_partials = nullptr; // stage-1 output of the statistics reduction
    unsigned long long *summary_scratch = nullptr; // partials + "last one out" counters of the one-launch returns summary (in the slab: zero)
    void *scratch = nullptr; // staging for host-side get/set of boards and scores (16 B per board), lazily
    // host-resident I/O (g2048_host_io_map): one block of pinned, device-mapped, coherent host memory
    void *host_base = nullptr;
    g2048_host_io host_io{};          // host addresses handed to the caller
    g2048_host_io host_io_dev{};      // the same arrays as the device sees them
    // completion word (its own 64-byte pinned, device-mapped, coherent block): published by the device, polled by the host
    unsigned long long *done_host = nullptr, *done_dev = nullptr;
    unsigned long long done_count = 0;
    // two-chain rollouts (g2048_set_chains): the side stream, its fork / join events and its launch thread
    int chains = 1;
    SideChain *side = nullptr; // the device's shared side chain (a reference is held while chains == 2 was ever set)
    hipEvent_t fork_event = nullptr, join_event = nullptr;
    unsigned long long *chain_flags = nullptr; // device memory (256 B): [0] fork ticket, [16] join ticket (own cache lines), [24] scratch
    unsigned long long chain_seq = 0;
    // a ticket wait that ran out (flag_wait_kernel) reports here: 64 bytes of pinned, coherent host memory, checked at
    // the entry of every call on the engine
    unsigned long long *chain_err_host = nullptr, *chain_err_dev = nullptr;
    // the measurement / test knobs of the two-chain form, read from the environment by g2048_set_chains (NOT per rollout)
    uint32_t chain_min_steps = 0;      // G2048_TWO_CHAIN_MIN_STEPS: split every rollout of at least that many steps (0: the warm / cold rule)
    uint32_t chain_wait_polls = g2048::kFlagWaitPolls; // G2048_FLAG_WAIT_POLLS: bound of a ticket wait, ~1 us per poll
    bool chain_any_priority = false;   // G2048_CHAIN_ANY_PRIORITY: also split on a caller's stream of the side stream's priority
    bool chain_by_events = false;      // G2048_CHAIN_SYNC=events: fork / join by HIP events instead of tickets
    // set when a call left the engine in a state its caller cannot know (a two-chain rollout that failed half-way):
    // every later call fails with this message instead of continuing on half-stepped boards
    int poisoned = 0;
    char poison_msg[320] = "";
    int last_rollout_chains = 1; // what the most recent g2048_rollout did (g2048_get_chains_used)
    // A rollout over the SAME buffers as the one before it is replayed from a cached hipGraph when the batch is small
    // enough for the host's launch rate to be the limit (g2048_rollout; kGraphMaxBoards).  The key is everything the
    // graph's frozen kernel arguments depend on; the clock comes through *graph_t_dev.
    struct GraphKey {
        uint32_t k_steps = 0;
        uint64_t stride = 0, seed = 0, board_offset = 0;
        const void *actions = nullptr;
        float *reward = nullptr;
        uint8_t *terminated = nullptr;
        uint4 *last_record = nullptr;
        int32_t action_dtype = 0;
        int auto_reset = 0;
        float illegal_reward = 0.0f;
        bool operator==(const GraphKey &o) const
        {
            return k_steps == o.k_steps && stride == o.stride && seed == o.seed && board_offset == o.board_offset && actions == o.actions &&
                   reward == o.reward && terminated == o.terminated && last_record == o.last_record && action_dtype == o.action_dtype &&
                   auto_reset == o.auto_reset && std::memcmp(&illegal_reward, &o.illegal_reward, sizeof(float)) == 0;
        }
    };
    struct GraphEntry {
        GraphKey key{};
        g2048::RolloutGraph g{};
        hipStream_t last_stream = nullptr; // where it was last replayed (an entry is only evicted once that stream has drained)
    };
    static constexpr int kGraphSlots = 4;  // a trainer alternates between a few sets of rollout buffers at most
    GraphEntry graphs[kGraphSlots];        // the cached graphs, replaced round robin
    int graph_next = 0;
    GraphKey graph_seen{};                 // the key of the previous rollout that was eligible (cached or not)
    unsigned long long *graph_t_dev = nullptr;
    uint32_t graph_max_boards = 1u << 17; // batches up to this size use the form (G2048_GRAPH_MAX_BOARDS, read by g2048_create)
    int graph_enabled = 1;              // G2048_ROLLOUT_GRAPH=0 (read by g2048_create) turns the form off; a failing graph call too
    char graph_off_reason[200] = "";    // ... and says why here (g2048_graph_status)
    uint64_t graph_replays = 0;         // rollouts served from the cached graph (g2048_get_graph_replays)
    // strict actions (g2048_set_strict_actions): 64 bytes of pinned, coherent host memory a step kernel reports an action
    // outside 0..3 to; read (and cleared) at the entry of the next call on the engine
    int strict_actions = 0;
    unsigned long long *action_err_host = nullptr, *action_err_dev = nullptr;
    int track_last = 1;   // keep the terminal record of every board's most recent finished episode (g2048_set_last_records)
    int32_t *returns = nullptr; // send buffer of the all-gather (int32[n]), lazily; NOT the staging buffer: a collective
                                // in flight on one stream must not be clobbered by a get_* call on another
};

namespace {

struct StateHeader {
    uint64_t magic, n, seed, board_offset, t;
    int32_t fresh;
    uint32_t max_exp;
    float illegal_reward;
    uint32_t reserved;
};

size_t align_up(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }

g2048::StepArgs make_args(const g2048_engine *e, const g2048_step_io *io, int auto_reset)
{
    g2048::StepArgs a{};
    a.st = e->st;
    if (!e->track_last)
        a.st.last_record = nullptr; // the kernels skip the terminal-record store
    if (io) {
        a.actions = io->actions;
        a.reward = io->reward;
        a.terminated = io->terminated;
        a.illegal = io->illegal;
        a.highest = io->highest;
        a.terminal_boards = reinterpret_cast<uint4 *>(io->terminal_boards);
        a.obs = io->obs;
        a.obs_dtype = static_cast<uint32_t>(io->obs_dtype);
        a.boards_out = reinterpret_cast<uint4 *>(io->boards_out);
    }
    a.action_err = e->strict_actions ? e->action_err_dev : nullptr;
    a.n = static_cast<uint32_t>(e->n);
    a.board_offset = static_cast<uint32_t>(e->board_offset);
    a.seed_lo = static_cast<uint32_t>(e->seed);
    a.seed_hi = static_cast<uint32_t>(e->seed >> 32);
    a.t_lo = static_cast<uint32_t>(e->t);
    a.t_hi = static_cast<uint32_t>(e->t >> 32);
    a.illegal_reward = e->illegal_reward;
    a.max_exp = e->max_exp;
    a.auto_reset = auto_reset ? 1u : 0u;
    return a;
}

// the device state as the kernels should see it: no terminal-record array when the engine does not keep them
g2048::DeviceState tracked_state(const g2048_engine *e)
{
    g2048::DeviceState st = e->st;
    if (!e->track_last)
        st.last_record = nullptr;
    return st;
}

int need_last_records(const g2048_engine *e, const char *what)
{
    if (e->track_last)
        return G2048_OK;
    return fail(G2048_ERR_INVALID, "%s reads the boards' terminal records, which this engine does not keep "
                                   "(g2048_set_last_records(e, 1, stream) turns them on)", what);
}

size_t obs_board_bytes(int dtype) { return static_cast<size_t>(256) << (dtype < 0 ? 0 : dtype); } // 16 channels x 16 cells

// Every call on an engine starts here: NULL, poisoned (a call that failed half-way), or a two-chain ticket wait that
// ran out on the device (the chains ran unordered since: whatever the engine holds is undefined).
int usable(const g2048_engine *e)
{
    if (!e)
        return fail(G2048_ERR_INVALID, "engine is NULL");
    if (e->poisoned)
        return fail(G2048_ERR_HIP, "%s", e->poison_msg);
    if (e->chain_err_host && __atomic_load_n(e->chain_err_host, __ATOMIC_ACQUIRE) != 0ull)
        return fail(G2048_ERR_HIP, "a two-chain rollout's ordering ticket (%llu) did not arrive within the wait's bound "
                                   "(%u polls of ~1 us): its chains ran unordered from there on, the engine's boards and "
                                   "the rollout's outputs are undefined -- destroy the engine",
                    __atomic_load_n(e->chain_err_host, __ATOMIC_ACQUIRE), e->chain_wait_polls);
    if (e->action_err_host) {
        // strict actions: a step kernel that has COMPLETED by now saw an action outside 0..3.  Reported once, by this call
        // (which does nothing else), and cleared; the step itself played the action's low two bits.
        const unsigned long long w = __atomic_exchange_n(e->action_err_host, 0ull, __ATOMIC_ACQ_REL);
        if (w != 0ull)
            return fail(G2048_ERR_INVALID, "strict actions: an earlier step was given an action outside 0..3 (e.g. global board %u, "
                                           "low byte 0x%02x); that step played the action's low two bits (the reference, "
                                           "game2048_env.py:210-212, would have played 4 as 'down' and -1 as 'right').  "
                                           "This call did nothing; the report is cleared",
                        static_cast<unsigned>(w & 0xffffffffull), static_cast<unsigned>((w >> 32) & 0xffu));
    }
    return G2048_OK;
}

int poison(g2048_engine *e, const char *what)
{
    e->poisoned = 1;
    snprintf(e->poison_msg, sizeof e->poison_msg, "%s; the engine is unusable from here on (a rollout was left half-done): destroy it", what);
    return fail(G2048_ERR_HIP, "%s", e->poison_msg);
}

size_t action_size(int dtype)
{
    switch (dtype) {
    case G2048_ACT_U8: return 1;
    case G2048_ACT_I32: return 4;
    case G2048_ACT_I64: return 8;
    default: return 0;
    }
}

int check_io(const g2048_step_io *io)
{
    if (!io)
        return fail(G2048_ERR_INVALID, "io is NULL");
    if (io->action_dtype < G2048_ACT_RANDOM || io->action_dtype > G2048_ACT_I64)
        return fail(G2048_ERR_INVALID, "unknown action_dtype %d", io->action_dtype);
    if (io->action_dtype != G2048_ACT_RANDOM && !io->actions)
        return fail(G2048_ERR_INVALID, "actions is NULL but action_dtype is %d", io->action_dtype);
    // the kernels use natural-width loads and stores
    if ((reinterpret_cast<uintptr_t>(io->actions) & (action_size(io->action_dtype) ? action_size(io->action_dtype) - 1 : 0)) ||
        (reinterpret_cast<uintptr_t>(io->reward) & 3u) || (reinterpret_cast<uintptr_t>(io->terminal_boards) & 15u) ||
        (reinterpret_cast<uintptr_t>(io->boards_out) & 15u))
        return fail(G2048_ERR_INVALID, "misaligned buffer: actions need their element size, reward 4 bytes, "
                                       "terminal_boards and boards_out 16 bytes");
    if (io->obs) {
        if (io->obs_dtype < G2048_OBS_U8 || io->obs_dtype > G2048_OBS_F32)
            return fail(G2048_ERR_INVALID, "unknown obs_dtype %d", io->obs_dtype);
        if (reinterpret_cast<uintptr_t>(io->obs) & 15u)
            return fail(G2048_ERR_INVALID, "misaligned buffer: obs needs 16 bytes");
    }
    return G2048_OK;
}

} // namespace

extern "C" {

const char *g2048_last_error(void) { return g_error; }

int g2048_abi_version(void) { return 15; }

int g2048_create(uint64_t n_boards, int device, uint64_t seed, uint64_t board_offset, g2048_engine **out)
{
    if (!out)
        return fail(G2048_ERR_INVALID, "out is NULL");
    *out = nullptr;
    // (the launch grid is whole 256-lane blocks indexed in 32 bits, hence the 0xffffff00 cap)
    if (n_boards == 0 || n_boards > 0xffffff00ull || board_offset + n_boards > 0x100000000ull)
        return fail(G2048_ERR_INVALID, "n_boards=%llu board_offset=%llu: global board indices must fit 32 bits",
                    (unsigned long long)n_boards, (unsigned long long)board_offset);
    int count = 0;
    hipError_t err = hipGetDeviceCount(&count);
    if (err != hipSuccess || count == 0)
        return fail(G2048_ERR_HIP, "no HIP device available (%s); this library has no CPU path",
                    err != hipSuccess ? hipGetErrorString(err) : "device count is 0");
    if (device < 0 || device >= count)
        return fail(G2048_ERR_INVALID, "device %d out of range (have %d)", device, count);
    G2048_HIP(hipSetDevice(device));

    g2048_engine *e = new (std::nothrow) g2048_engine;
    if (!e)
        return fail(G2048_ERR_NOMEM, "out of host memory");
    e->n = n_boards;
    e->device = device;
    e->seed = seed;
    e->board_offset = board_offset;
    if (const char *v = std::getenv("G2048_ROLLOUT_GRAPH"))
        e->graph_enabled = std::atoi(v) != 0;
    if (const char *v = std::getenv("G2048_GRAPH_MAX_BOARDS")) { // measurement knob
        const long long m = std::atoll(v);
        if (m >= 0 && m <= 0xffffffffll)
            e->graph_max_boards = static_cast<uint32_t>(m);
    }

    const size_t n = n_boards;
    const size_t off_boards = 0;
    const size_t off_last_record = off_boards + align_up(n * 16);
    const size_t off_counters = off_last_record + align_up(n * 16);
    // one slot of kSlotWords counters per 64 boards (whole launch blocks of the largest block size: a wavefront that lies
    // wholly past the end still loads and stores ITS slot)
    const size_t n_counters = ((n + g2048::kSlotBlockLanes - 1) / g2048::kSlotBlockLanes) * (g2048::kSlotBlockLanes / 64) * g2048::kSlotWords;
    const size_t off_stats = off_counters + align_up(n_counters * sizeof(unsigned long long));
    const size_t off_summary = off_stats + align_up(sizeof(g2048::StatsOut));
    const size_t off_graph_t = off_summary + align_up(g2048::kSummaryScratchWords * sizeof(unsigned long long));
    e->slab_bytes = off_graph_t + 256; // the clock word of the cached rollout graphs: here, so that building one allocates nothing
    err = hipMalloc(&e->slab, e->slab_bytes);
    if (err != hipSuccess) {
        const size_t wanted = e->slab_bytes;
        delete e;
        return fail(G2048_ERR_NOMEM, "hipMalloc(%zu) failed: %s", wanted, hipGetErrorString(err));
    }
    // hipMemset of device memory only ENQUEUES the fill (on the null stream: 3 us for a call that takes 1.4 ms on 8 GiB,
    // tools/memset_probe.py), and a caller's non-blocking stream -- every torch stream but the default one -- is not
    // ordered behind the null stream: without the wait the engine's first kernels could run before the clear.
    err = hipMemset(e->slab, 0, e->slab_bytes);
    if (err == hipSuccess)
        err = hipStreamSynchronize(nullptr);
    if (err != hipSuccess) {
        (void)hipFree(e->slab);
        delete e;
        return fail(G2048_ERR_HIP, "hipMemset failed: %s", hipGetErrorString(err));
    }
    err = hipMalloc(reinterpret_cast<void **>(&e->stats_partials), g2048::kStatsPartialWords * sizeof(unsigned long long));
    if (err != hipSuccess) {
        (void)hipFree(e->slab);
        delete e;
        return fail(G2048_ERR_NOMEM, "hipMalloc of the statistics scratch failed: %s", hipGetErrorString(err));
    }
    char *base = static_cast<char *>(e->slab);
    e->st.boards = reinterpret_cast<uint4 *>(base + off_boards);
    e->st.last_record = reinterpret_cast<uint4 *>(base + off_last_record);
    e->st.ep_counters = reinterpret_cast<unsigned long long *>(base + off_counters);
    e->stats_dev = reinterpret_cast<g2048::StatsOut *>(base + off_stats);
    e->summary_scratch = reinterpret_cast<unsigned long long *>(base + off_summary);
    e->graph_t_dev = reinterpret_cast<unsigned long long *>(base + off_graph_t);
    if (!e->graph_enabled)
        snprintf(e->graph_off_reason, sizeof e->graph_off_reason, "G2048_ROLLOUT_GRAPH=0 in the environment of g2048_create");
    *out = e;
    return G2048_OK;
}

int g2048_destroy(g2048_engine *e)
{
    if (!e)
        return G2048_OK;
    hipError_t err = hipSuccess;
    if (e->side) {
        SideChain *dead = nullptr;
        {
            std::lock_guard<std::mutex> lock(g_side_mutex);
            if (--e->side->refs == 0) {
                dead = e->side;
                g_side[dead->device] = nullptr;
            }
        }
        if (dead) {
            dead->launcher.stop();
            (void)hipSetDevice(dead->device);
            if (dead->stream)
                (void)hipStreamDestroy(dead->stream);
            delete dead;
        }
        e->side = nullptr;
    }
    if (e->slab) {
        (void)hipSetDevice(e->device);
        if (e->fork_event)
            (void)hipEventDestroy(e->fork_event);
        if (e->join_event)
            (void)hipEventDestroy(e->join_event);
        bool any_graph = false;
        for (auto &entry : e->graphs)
            any_graph = any_graph || entry.g.exec;
        if (any_graph)
            (void)hipDeviceSynchronize(); // a replay may still be running: an executable graph is destroyed only at rest (the
                                          // hipFree of the slab below waits for the device anyway)
        for (auto &entry : e->graphs)
            g2048::destroy_rollout_graph(entry.g);
        if (e->chain_flags)
            (void)hipFree(e->chain_flags);
        if (e->chain_err_host)
            (void)hipHostFree(e->chain_err_host);
        if (e->action_err_host)
            (void)hipHostFree(e->action_err_host);
        if (e->st.rng)
            (void)hipFree(e->st.rng);
        if (e->scratch)
            (void)hipFree(e->scratch);
        if (e->returns)
            (void)hipFree(e->returns);
        if (e->host_base)
            (void)hipHostFree(e->host_base);
        if (e->done_host)
            (void)hipHostFree(e->done_host);
        if (e->stats_partials)
            (void)hipFree(e->stats_partials);
        err = hipFree(e->slab);
    }
    delete e;
    if (err != hipSuccess)
        return fail(G2048_ERR_HIP, "hipFree failed: %s", hipGetErrorString(err));
    return G2048_OK;
}

int g2048_seed(g2048_engine *e, uint64_t seed, void *stream)
{
    if (int rc = usable(e))
        return rc;
    e->seed = seed;
    e->t = 0;
    e->fresh = 1;
    // episode statistics restart with the stream: cleared by a kernel ON THE CALLER'S STREAM, so the
    // clear is ordered against step kernels already enqueued there
    G2048_HIP(hipSetDevice(e->device));
    G2048_HIP(g2048::launch_clear_stats(tracked_state(e), static_cast<uint32_t>(e->n), static_cast<hipStream_t>(stream)));
    return G2048_OK;
}

int g2048_set_last_records(g2048_engine *e, int enable, void *stream)
{
    if (int rc = usable(e))
        return rc;
    G2048_HIP(hipSetDevice(e->device));
    if (enable && !e->track_last) // records from before the pause would be stale: start from "none yet"
        G2048_HIP(hipMemsetAsync(e->st.last_record, 0, e->n * 16, static_cast<hipStream_t>(stream)));
    e->track_last = enable ? 1 : 0;
    return G2048_OK;
}

int g2048_get_last_records(const g2048_engine *e)
{
    return e ? e->track_last : 0;
}

int g2048_get_clock(const g2048_engine *e, uint64_t *t)
{
    if (int rc = usable(e))
        return rc;
    if (!t)
        return fail(G2048_ERR_INVALID, "NULL argument");
    *t = e->t;
    return G2048_OK;
}

int g2048_set_clock(g2048_engine *e, uint64_t t)
{
    if (int rc = usable(e))
        return rc;
    e->t = t;
    e->fresh = 0;
    return G2048_OK;
}

uint64_t g2048_num_boards(const g2048_engine *e) { return e ? e->n : 0; }

int g2048_set_illegal_move_reward(g2048_engine *e, float reward)
{
    if (int rc = usable(e))
        return rc;
    e->illegal_reward = reward;
    return G2048_OK;
}

int g2048_set_strict_actions(g2048_engine *e, int enable)
{
    if (int rc = usable(e))
        return rc;
    if (enable && !e->action_err_host) {
        G2048_HIP(hipSetDevice(e->device));
        void *host = nullptr, *dev = nullptr;
        hipError_t err = hipHostMalloc(&host, 64, hipHostMallocMapped | hipHostMallocCoherent);
        if (err != hipSuccess)
            return fail(G2048_ERR_NOMEM, "hipHostMalloc(64) for the action error word failed: %s", hipGetErrorString(err));
        err = hipHostGetDevicePointer(&dev, host, 0);
        if (err != hipSuccess) {
            (void)hipHostFree(host);
            return fail(G2048_ERR_HIP, "hipHostGetDevicePointer failed: %s", hipGetErrorString(err));
        }
        std::memset(host, 0, 64);
        e->action_err_host = static_cast<unsigned long long *>(host);
        e->action_err_dev = static_cast<unsigned long long *>(dev);
    }
    e->strict_actions = enable ? 1 : 0;
    return G2048_OK;
}

int g2048_get_strict_actions(const g2048_engine *e) { return e ? e->strict_actions : 0; }

int g2048_set_max_tile(g2048_engine *e, int max_exp)
{
    if (int rc = usable(e))
        return rc;
    if (max_exp < 0 || max_exp > 31)
        return fail(G2048_ERR_INVALID, "max_exp %d out of range 0..31", max_exp);
    e->max_exp = static_cast<uint32_t>(max_exp);
    return G2048_OK;
}

int g2048_reset(g2048_engine *e, int new_transaction, uint32_t first_slot, const uint8_t *mask, void *stream)
{
    if (int rc = usable(e))
        return rc;
    G2048_HIP(hipSetDevice(e->device));
    if (new_transaction)
        e->t += 1;
    e->fresh = 0;
    const g2048::StepArgs a = make_args(e, nullptr, 0);
    if (e->st.rng)
        G2048_HIP(g2048::launch_reset_numpy(a, mask, static_cast<hipStream_t>(stream)));
    else
        G2048_HIP(g2048::launch_reset(a, first_slot, mask, static_cast<hipStream_t>(stream)));
    return G2048_OK;
}

int g2048_step(g2048_engine *e, const g2048_step_io *io, int auto_reset, void *stream)
{
    if (int rc = usable(e))
        return rc;
    if (int rc = check_io(io))
        return rc;
    G2048_HIP(hipSetDevice(e->device));
    e->t += 1;
    e->fresh = 0;
    const g2048::StepArgs a = make_args(e, io, auto_reset);
    if (e->st.rng) {
        G2048_HIP(g2048::launch_step_numpy(a, io->action_dtype, static_cast<hipStream_t>(stream)));
        if (a.boards_out) // this mode's resets run behind the step kernel: the plain boards come from the export kernel
            G2048_HIP(g2048::launch_export_boards(e->st.boards, a.n, a.boards_out, static_cast<hipStream_t>(stream)));
    } else {
        G2048_HIP(g2048::launch_step(a, io->action_dtype, static_cast<hipStream_t>(stream)));
    }
    return G2048_OK;
}

// The buffers of step j of a rollout: the io pointers advanced by j * stride elements.
static g2048_step_io io_of_step(const g2048_step_io &io, uint32_t j, uint64_t stride)
{
    g2048_step_io s = io;
    const size_t off = static_cast<size_t>(j) * stride;
    if (s.actions) s.actions = static_cast<const char *>(io.actions) + off * action_size(io.action_dtype);
    if (s.reward) s.reward = io.reward + off;
    if (s.terminated) s.terminated = io.terminated + off;
    if (s.illegal) s.illegal = io.illegal + off;
    if (s.highest) s.highest = io.highest + off;
    if (s.terminal_boards) s.terminal_boards = io.terminal_boards + off * 16;
    if (s.obs) s.obs = static_cast<char *>(io.obs) + off * obs_board_bytes(io.obs_dtype);
    if (s.boards_out) s.boards_out = io.boards_out + off * 16;
    return s;
}

// Launch arguments restricted to the boards [first, first + count) of the engine (first is a multiple of 256: whole
// launch blocks and whole episode slots): every per-board pointer advanced by `first`, the global board index with it.
static g2048::StepArgs part_of(g2048::StepArgs a, int action_dtype, uint32_t first, uint32_t count)
{
    a.st.boards += first;
    if (a.st.last_record) a.st.last_record += first;
    a.st.ep_counters += static_cast<size_t>(first / 64u) * g2048::kSlotWords;
    if (a.actions) a.actions = static_cast<const char *>(a.actions) + static_cast<size_t>(first) * action_size(action_dtype);
    if (a.reward) a.reward += first;
    if (a.terminated) a.terminated += first;
    if (a.illegal) a.illegal += first;
    if (a.highest) a.highest += first;
    if (a.terminal_boards) a.terminal_boards += first;
    if (a.boards_out) a.boards_out += first;
    if (a.obs) a.obs = static_cast<char *>(a.obs) + static_cast<size_t>(first) * obs_board_bytes(static_cast<int>(a.obs_dtype));
    a.n = count;
    a.board_offset += first;
    return a;
}

// When do two chains pay?  WARM -- the device's side chain had work within the last ~50 ms -- they cost a fixed ~6 us per
// rollout more than one chain and save ~1.2 us per step at 2^20 boards: HIP-event time of a rollout of k steps, fitted
// over k = 8 .. 128 right behind a 128-step rollout (tools/chain_fixed_cost.py, profiles/r04_v_chain_fixed_cost.txt; two
// boxes): one chain 9.10 us/step + 12.9 us (8.92 + 21.0), two chains 7.91 us/step + 18.6 us (8.07 + 25.0; with HIP events
// instead of the ticket kernels: + 35.3 / + 46.3 us); k = 8 is a tie, k = 16 is 3-8 % faster.  A 20-step two-chain
// rollout after X of idle (tools/chain_gap_probe.py, profiles/r04_ab_chain_gap_probe.txt; one chain: 188-209 us):
// X = 0.2 ms 175 us, 5 ms 184 (the launch thread asleep by then: waking it costs ~10 us), 50 ms 186, 300 ms 223 -- COLD,
// the side stream's first kernels start late.  Hence: warm, two chains from kTwoChainMinSteps; cold, only from
// kTwoChainColdMinSteps, where 40 us are a few per cent.
constexpr uint32_t kTwoChainMinSteps = 12, kTwoChainColdMinSteps = 64;
// Cached-graph replay of a launch train (g2048_kernels.hip "a k-step launch train as a CACHED hipGraph"): up to 2^17 boards
// -- where one host thread cannot issue launches as fast as the device retires them -- and from 8 steps.
constexpr uint32_t kGraphMinSteps = 8; // (the size limit is g2048_engine::graph_max_boards)
constexpr double kSideWarmWindowUs = 50000.0; // after the estimated end of the side chain's last work

static int ensure_side_chain(g2048_engine *e)
{
    if (e->side)
        return G2048_OK;
    if (e->device < 0 || e->device >= 64)
        return fail(G2048_ERR_INVALID, "two chains are available on devices 0..63");
    if (!e->chain_flags) { // the engine's own fork / join tickets (and events, for G2048_CHAIN_SYNC=events)
        G2048_HIP(hipEventCreateWithFlags(&e->fork_event, hipEventDisableTiming));
        G2048_HIP(hipEventCreateWithFlags(&e->join_event, hipEventDisableTiming));
        G2048_HIP(hipMalloc(reinterpret_cast<void **>(&e->chain_flags), 256));
        G2048_HIP(hipMemset(e->chain_flags, 0, 256));
        G2048_HIP(hipStreamSynchronize(nullptr)); // (the fill is only enqueued, and not ordered against non-blocking streams)
    }
    if (!e->chain_err_host) { // where a ticket wait that ran out reports (flag_wait_kernel)
        void *host = nullptr, *dev = nullptr;
        hipError_t err = hipHostMalloc(&host, 64, hipHostMallocMapped | hipHostMallocCoherent);
        if (err != hipSuccess)
            return fail(G2048_ERR_NOMEM, "hipHostMalloc(64) for the chains' error word failed: %s", hipGetErrorString(err));
        err = hipHostGetDevicePointer(&dev, host, 0);
        if (err != hipSuccess) {
            (void)hipHostFree(host);
            return fail(G2048_ERR_HIP, "hipHostGetDevicePointer failed: %s", hipGetErrorString(err));
        }
        std::memset(host, 0, 64);
        e->chain_err_host = static_cast<unsigned long long *>(host);
        e->chain_err_dev = static_cast<unsigned long long *>(dev);
    }
    std::lock_guard<std::mutex> lock(g_side_mutex);
    SideChain *sc = g_side[e->device];
    if (!sc) {
        sc = new (std::nothrow) SideChain;
        if (!sc)
            return fail(G2048_ERR_NOMEM, "out of host memory");
        sc->device = e->device;
        // The side stream is created at the HIGHEST priority the device offers.  In a process that also holds an RCCL
        // communicator (dozens of hardware queues) a normal-priority side queue shares its slot by time slices and the
        // two chains stop overlapping: 10.9-11.1 us per step instead of 8.0 at 2^20 boards, worse than one chain (9.1);
        // with the priority it is 8.0 with or without RCCL (bench.py, forced one-rank process group, K = 400).
        int least = 0, greatest = 0;
        hipError_t err = hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (err == hipSuccess)
            err = hipStreamCreateWithPriority(&sc->stream, hipStreamNonBlocking, greatest);
        if (err != hipSuccess) {
            delete sc;
            return fail(G2048_ERR_HIP, "cannot create the side stream: %s", hipGetErrorString(err));
        }
        sc->priority = greatest;
        SideLauncher *w = &sc->launcher;
        w->spin_us = g2048::side_spin_us_from_env(); // 200 us unless the caller opted into a longer window
        w->start();
        // prime it: a thread's first HIP calls set up per-thread runtime state (tens of microseconds) -- here, not in
        // somebody's first two-chain rollout
        unsigned long long *scratch_flag = e->chain_flags + 24; // (fork = 0 and join = 16 leave words 17..31 of the 256 bytes free)
        const int dev = e->device;
        hipStream_t side_stream = sc->stream;
        const uint64_t ticket = w->post([dev, scratch_flag, side_stream]() -> int {
            if (hipSetDevice(dev) != hipSuccess)
                return G2048_ERR_HIP;
            for (int k = 0; k < 4; ++k)
                (void)g2048::launch_flag_set(scratch_flag, 0ull, side_stream);
            return hipStreamSynchronize(side_stream) == hipSuccess ? G2048_OK : G2048_ERR_HIP;
        });
        if (w->wait(ticket) != G2048_OK) {
            w->stop();
            (void)hipStreamDestroy(sc->stream);
            delete sc;
            return fail(G2048_ERR_HIP, "the side launch thread could not reach device %d", e->device);
        }
        g_side[e->device] = sc;
    }
    ++sc->refs;
    e->side = sc;
    return G2048_OK;
}

int g2048_set_chains(g2048_engine *e, int chains)
{
    if (int rc = usable(e))
        return rc;
    if (chains != 1 && chains != 2)
        return fail(G2048_ERR_INVALID, "chains must be 1 or 2 (got %d)", chains);
    if (chains == 2) {
        G2048_HIP(hipSetDevice(e->device));
        if (int rc = ensure_side_chain(e))
            return rc;
        // The knobs of this form are read HERE, once (not per rollout): a test or a measurement sets the environment and
        // calls g2048_set_chains(e, 2) again.
        const char *v = std::getenv("G2048_TWO_CHAIN_MIN_STEPS");
        const long forced = v ? std::atol(v) : 0l;
        e->chain_min_steps = forced >= 2 ? static_cast<uint32_t>(forced) : 0u;
        v = std::getenv("G2048_FLAG_WAIT_POLLS");
        const long polls = v ? std::atol(v) : 0l;
        e->chain_wait_polls = polls > 0 && polls < 0x7fffffffl ? static_cast<uint32_t>(polls) : g2048::kFlagWaitPolls;
        e->chain_any_priority = std::getenv("G2048_CHAIN_ANY_PRIORITY") != nullptr;
        v = std::getenv("G2048_CHAIN_SYNC");
        e->chain_by_events = v && std::strcmp(v, "events") == 0;
    }
    e->chains = chains;
    return G2048_OK;
}

uint64_t g2048_get_graph_replays(const g2048_engine *e) { return e ? e->graph_replays : 0; }
const char *g2048_graph_status(const g2048_engine *e) { return e ? e->graph_off_reason : "engine is NULL"; }

int g2048_get_chains(const g2048_engine *e) { return e ? e->chains : 0; }
int g2048_get_chains_used(const g2048_engine *e) { return e ? e->last_rollout_chains : 0; }

static g2048_engine::GraphKey graph_key_of(const g2048_engine *e, uint32_t k_steps, const g2048_step_io *io, uint64_t stride,
                                           int auto_reset, const g2048::StepArgs &a0)
{
    g2048_engine::GraphKey key;
    key.k_steps = k_steps; key.stride = stride; key.seed = e->seed; key.board_offset = e->board_offset;
    key.actions = io->actions; key.reward = io->reward; key.terminated = io->terminated; key.last_record = a0.st.last_record;
    key.action_dtype = io->action_dtype; key.auto_reset = auto_reset ? 1 : 0; key.illegal_reward = e->illegal_reward;
    return key;
}

static g2048_engine::GraphEntry *find_graph(g2048_engine *e, const g2048_engine::GraphKey &key)
{
    for (auto &entry : e->graphs)
        if (entry.g.exec && entry.key == key)
            return &entry;
    return nullptr;
}

// Build the cached graph for `key` in the next slot (round robin).  A failure is not fatal: the engine keeps launching by
// stream, and stops trying.
// Builds (or rebuilds) a cached graph.  Nothing here is a stream operation -- no allocation, no memset, no synchronisation
// (the clock word lives in the engine's slab) -- so it is safe while the application has a stream capture open, in any
// capture mode.  An entry is evicted only when the stream it was last replayed on has drained (hipStreamQuery); otherwise
// this rollout simply is not cached (returns NULL, the form stays on).
static g2048_engine::GraphEntry *build_graph(g2048_engine *e, const g2048_engine::GraphKey &key, const g2048::StepArgs &a0,
                                             int action_dtype, uint32_t k_steps, uint64_t stride)
{
    g2048_engine::GraphEntry *slot = &e->graphs[e->graph_next];
    if (slot->g.exec) { // a fifth set of buffers: rare
        if (hipStreamQuery(slot->last_stream) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr; // its last replay may still be running: keep it, launch this rollout kernel by kernel
        }
    }
    e->graph_next = (e->graph_next + 1) % g2048_engine::kGraphSlots;
    g2048::destroy_rollout_graph(slot->g);
    const hipError_t err = g2048::build_rollout_graph(a0, action_dtype, k_steps, stride, e->graph_t_dev, &slot->g);
    if (err != hipSuccess) {
        (void)hipGetLastError();
        e->graph_enabled = 0;
        snprintf(e->graph_off_reason, sizeof e->graph_off_reason, "building the graph of a %u-step rollout failed: %s", k_steps,
                 hipGetErrorString(err));
        return nullptr;
    }
    slot->key = key;
    slot->last_stream = nullptr;
    return slot;
}

int g2048_rollout_prepare(g2048_engine *e, uint32_t k_steps, const g2048_step_io *io, uint64_t stride, int auto_reset)
{
    if (int rc = usable(e))
        return rc;
    if (int rc = check_io(io))
        return rc;
    G2048_HIP(hipSetDevice(e->device));
    if (!e->graph_enabled || e->n > e->graph_max_boards || k_steps < kGraphMinSteps)
        return G2048_OK; // this rollout is launched kernel by kernel whatever happens: nothing to prepare
    g2048_step_io s0 = *io;
    const g2048::StepArgs a0 = make_args(e, &s0, auto_reset);
    if (!g2048::rollout_graph_supported(a0))
        return G2048_OK;
    const g2048_engine::GraphKey key = graph_key_of(e, k_steps, io, stride, auto_reset, a0);
    if (!find_graph(e, key))
        (void)build_graph(e, key, a0, io->action_dtype, k_steps, stride);
    return G2048_OK;
}

int g2048_rollout(g2048_engine *e, uint32_t k_steps, const g2048_step_io *io, uint64_t stride, int auto_reset,
                  void *stream)
{
    if (int rc = usable(e))
        return rc;
    if (int rc = check_io(io))
        return rc;
    G2048_HIP(hipSetDevice(e->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (k_steps == 0)
        return G2048_OK;
    // ---- two chains: the batch is cut at a block boundary and each half gets its own stream and launch thread.
    //      Spawn-stream mode only (the numpy-RNG planes are indexed with the engine's board count), not while the
    //      caller captures a graph (another thread must not launch during a global capture), and only when both halves
    //      are whole blocks of work.  A rollout can only be split when ALL its k_steps actions are supplied up front, and
    //      the split only pays from kTwoChainMinSteps steps: a caller that steps one action at a time (ppo_train.py) never
    //      gets here, whatever g2048_set_chains said.
    const uint32_t n = static_cast<uint32_t>(e->n);
    const uint32_t first_half = (n / 2u) & ~255u;
    bool two = e->chains == 2 && e->side && !e->st.rng && first_half >= 256u && k_steps >= 2;
    // warm: the side chain had work until recently; need: the rollout length from which the split pays right now
    const bool warm = two && steady_now_ns() < e->side->busy_until_ns.load(std::memory_order_relaxed);
    const uint32_t need = e->chain_min_steps ? e->chain_min_steps : warm ? kTwoChainMinSteps : kTwoChainColdMinSteps;
    if (two && k_steps < need) {
        // Too short to split.  A rollout within reach of the threshold still wakes a sleeping launcher, so that a loop of
        // such rollouts finds it spinning when one of them qualifies; short ones (a step at a time) leave it asleep -- an
        // idle core is not spent on a caller that cannot use the second chain.
        if (k_steps >= kTwoChainMinSteps / 2u && e->side->launcher.sleeping.load())
            e->side->launcher.nudge();
        two = false;
    }
    if (two) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &st) != hipSuccess) {
            (void)hipGetLastError();
            two = false;
        } else if (st != hipStreamCaptureStatusNone) {
            two = false;
        }
    }
    if (two) {
        // A caller's stream at the side stream's own priority may be given the SAME hardware queue by the runtime, and
        // streams that share one run in submission order: the caller's wait for the join ticket could then sit in front of
        // the side launches it waits for.  Such a stream gets one chain.  (Streams of other priorities have their own
        // queues; one runtime call per rollout.)
        if (s != nullptr && !e->chain_any_priority) {
            int prio = 0;
            if (hipStreamGetPriority(s, &prio) != hipSuccess) {
                (void)hipGetLastError();
                two = false;
            } else if (prio == e->side->priority) {
                two = false;
            }
        }
    }
    e->last_rollout_chains = two ? 2 : 1;
    const uint64_t t0 = e->t;
    e->t += k_steps;
    e->fresh = 0;
    auto args_of = [e, io, stride, auto_reset, t0](uint32_t j) {
        const g2048_step_io sj = io_of_step(*io, j, stride);
        g2048::StepArgs a = make_args(e, &sj, auto_reset);
        const uint64_t t = t0 + 1u + j; // (make_args read e->t, which already stands at the end of the rollout)
        a.t_lo = static_cast<uint32_t>(t);
        a.t_hi = static_cast<uint32_t>(t >> 32);
        return a;
    };
    if (!two && e->graph_enabled && n <= e->graph_max_boards && k_steps >= kGraphMinSteps) {
        // ---- same buffers as last time (or a prepared plan): replay the cached graph of this launch train
        const g2048::StepArgs a0 = args_of(0);
        if (g2048::rollout_graph_supported(a0)) {
            const g2048_engine::GraphKey key = graph_key_of(e, k_steps, io, stride, auto_reset, a0);
            hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
            const bool capturing = hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone;
            if (capturing)
                (void)hipGetLastError();
            g2048_engine::GraphEntry *have = capturing ? nullptr : find_graph(e, key);
            if (!capturing && !have && key == e->graph_seen) // the second rollout over these buffers in a row: worth a graph
                have = build_graph(e, key, a0, io->action_dtype, k_steps, stride);
            e->graph_seen = key;
            if (have) {
                const hipError_t err = g2048::launch_rollout_graph(have->g, t0 + 1u, s);
                if (err == hipSuccess) {
                    have->last_stream = s;
                    ++e->graph_replays;
                    return G2048_OK;
                }
                // no step was enqueued (the launch of a graph is all or nothing; at most the one-lane clock write went out,
                // which nobody else reads): fall through to stream launches
                (void)hipGetLastError();
                g2048::destroy_rollout_graph(have->g);
                e->graph_enabled = 0;
                snprintf(e->graph_off_reason, sizeof e->graph_off_reason, "hipGraphLaunch of a cached %u-step rollout failed: %s",
                         k_steps, hipGetErrorString(err));
            }
        }
    }
    if (!two) {
        for (uint32_t j = 0; j < k_steps; ++j) {
            const g2048::StepArgs a = args_of(j);
            hipError_t err;
            if (e->st.rng) {
                err = g2048::launch_step_numpy(a, io->action_dtype, s);
                if (err == hipSuccess && a.boards_out)
                    err = g2048::launch_export_boards(e->st.boards, a.n, a.boards_out, s);
            } else {
                err = g2048::launch_step(a, io->action_dtype, s);
            }
            if (err != hipSuccess) {
                // steps 0 .. j-1 are enqueued and will run: the clock says so, and the caller gets the error
                e->t = t0 + j;
                return fail(G2048_ERR_HIP, "launch of step %u of %u failed: %s", j, k_steps, hipGetErrorString(err));
            }
        }
        return G2048_OK;
    }
    // fork / join: the side stream starts where the caller's stream stands, and whatever the caller enqueues next runs
    // after both halves.  By tickets in device memory (flag_set_kernel / flag_wait_kernel) -- or, G2048_CHAIN_SYNC=events,
    // by HIP events (the portable form; ~30 us more latency per rollout on this runtime).
    const bool by_flags = !e->chain_by_events;
    SideChain *sc = e->side;
    std::lock_guard<std::mutex> side_in_use(sc->use); // (another engine of this device may be using the side chain)
    hipStream_t side_stream = sc->stream;
    const unsigned long long seq = ++e->chain_seq;
    unsigned long long *fork_flag = e->chain_flags, *join_flag = e->chain_flags + 16;
    unsigned long long *err_word = e->chain_err_dev;
    const uint32_t polls = e->chain_wait_polls;
    {   // nothing has been launched yet: a failure here leaves the engine as it was
        hipError_t err;
        if (by_flags) {
            err = g2048::launch_flag_set(fork_flag, seq, s); // enqueued BEFORE the side thread can enqueue its wait
        } else {
            err = hipEventRecord(e->fork_event, s);
            if (err == hipSuccess)
                err = hipStreamWaitEvent(side_stream, e->fork_event, 0);
        }
        if (err != hipSuccess) {
            e->t = t0;
            return fail(G2048_ERR_HIP, "cannot fork the two chains: %s", hipGetErrorString(err));
        }
    }
    SideLauncher *w = &sc->launcher;
    const int dtype = io->action_dtype;
    // The caller's chain gets a HEAD START of one launch (~3.3 us of host time, about half a half-batch kernel): two
    // chains that start together run their load phases together, like one big kernel.
    hipError_t mine = g2048::launch_step(part_of(args_of(0), dtype, 0u, first_half), dtype, s);
    const uint64_t ticket = w->post([e, w, args_of, k_steps, dtype, first_half, n, fork_flag, join_flag, seq, side_stream, by_flags,
                                     err_word, polls]() -> int {
        if (hipSetDevice(e->device) != hipSuccess) {
            snprintf(w->error, sizeof w->error, "the side launch thread could not select device %d", e->device);
            return G2048_ERR_HIP;
        }
        hipError_t err = by_flags ? g2048::launch_flag_wait(fork_flag, seq, err_word, polls, side_stream) : hipSuccess;
        for (uint32_t j = 0; j < k_steps && err == hipSuccess; ++j)
            err = g2048::launch_step(part_of(args_of(j), dtype, first_half, n - first_half), dtype, side_stream);
        // the join ticket goes out even after a failed launch: the caller's stream must never wait for a ticket nobody sets
        const hipError_t tail = by_flags ? g2048::launch_flag_set(join_flag, seq, side_stream)
                                         : hipEventRecord(e->join_event, side_stream);
        if (err == hipSuccess)
            err = tail;
        if (err != hipSuccess) {
            snprintf(w->error, sizeof w->error, "launch on the side chain failed: %s", hipGetErrorString(err));
            return G2048_ERR_HIP;
        }
        return G2048_OK;
    });
    for (uint32_t j = 1; j < k_steps && mine == hipSuccess; ++j)
        mine = g2048::launch_step(part_of(args_of(j), dtype, 0u, first_half), dtype, s);
    const int theirs = w->wait(ticket); // (the side thread has ISSUED its launches; nothing waits for the device here)
    // The JOIN is enqueued whatever happened above: side-stream kernels that were launched may still be writing the
    // engine's and the caller's buffers, and whatever the caller enqueues next on `stream` must come after them.
    hipError_t join = by_flags ? g2048::launch_flag_wait(join_flag, seq, err_word, polls, s) : hipStreamWaitEvent(s, e->join_event, 0);
    // the side stream has ~k_steps half-batch kernels ahead of it (they are only enqueued): warm until they are done + a bit
    sc->busy_until_ns.store(steady_now_ns() + static_cast<int64_t>((k_steps * (4.0e-6 * n + 0.5) + kSideWarmWindowUs) * 1000.0),
                            std::memory_order_relaxed);
    if (theirs != G2048_OK || mine != hipSuccess || join != hipSuccess) {
        // some launches of the rollout are enqueued and some are not: the halves of the batch no longer stand at the same
        // step, and nothing the caller could do would repair that
        char what[256];
        if (theirs != G2048_OK)
            snprintf(what, sizeof what, "two-chain rollout: %s", w->error);
        else
            snprintf(what, sizeof what, "two-chain rollout: %s failed: %s", mine != hipSuccess ? "a launch on the caller's stream" : "the join",
                     hipGetErrorString(mine != hipSuccess ? mine : join));
        return poison(e, what);
    }
    return G2048_OK;
}

int g2048_rollout_fused(g2048_engine *e, uint32_t k_steps, const g2048_step_io *io, uint64_t stride, int auto_reset,
                        void *stream)
{
    if (int rc = usable(e))
        return rc;
    if (int rc = check_io(io))
        return rc;
    if (io->terminal_boards || io->obs || io->boards_out)
        return fail(G2048_ERR_INVALID, "g2048_rollout_fused writes neither terminal_boards, boards_out nor obs");
    if (k_steps == 0)
        return G2048_OK;
    G2048_HIP(hipSetDevice(e->device));
    e->t += 1; // transaction of the first fused step
    e->fresh = 0;
    g2048::StepArgs a = make_args(e, io, auto_reset); // (numpy-RNG mode: a.st.rng selects rollout_fused_numpy_kernel)
    a.k_steps = k_steps;
    G2048_HIP(g2048::launch_rollout_fused(a, io->action_dtype, stride, static_cast<hipStream_t>(stream)));
    e->t += k_steps - 1;
    return G2048_OK;
}

int g2048_rollout_random(g2048_engine *e, uint32_t k_steps, void *stream)
{
    if (int rc = usable(e))
        return rc;
    if (k_steps == 0)
        return G2048_OK;
    G2048_HIP(hipSetDevice(e->device));
    e->t += 1; // transaction of the first fused step
    e->fresh = 0;
    g2048::StepArgs a = make_args(e, nullptr, 1);
    a.k_steps = k_steps;
    if (e->st.rng) // numpy-RNG mode: the fused form of that mode with the synthetic policy and no per-step output
        G2048_HIP(g2048::launch_rollout_fused(a, G2048_ACT_RANDOM, 0, static_cast<hipStream_t>(stream)));
    else
        G2048_HIP(g2048::launch_rollout_random(a, static_cast<hipStream_t>(stream)));
    e->t += k_steps - 1;
    return G2048_OK;
}

int g2048_move(g2048_engine *e, const void *actions, int32_t action_dtype, int trial, int32_t *score_out,
               uint8_t *legal_out, void *stream)
{
    if (int rc = usable(e))
        return rc;
    if (!actions)
        return fail(G2048_ERR_INVALID, "NULL argument");
    if (action_dtype < G2048_ACT_U8 || action_dtype > G2048_ACT_I64)
        return fail(G2048_ERR_INVALID, "g2048_move needs an action buffer (dtype %d)", action_dtype);
    if ((reinterpret_cast<uintptr_t>(actions) & (action_size(action_dtype) - 1)) || (reinterpret_cast<uintptr_t>(score_out) & 3u))
        return fail(G2048_ERR_INVALID, "misaligned buffer: actions need their element size, score_out 4 bytes");
    G2048_HIP(hipSetDevice(e->device));
    G2048_HIP(g2048::launch_move(e->st.boards, static_cast<uint32_t>(e->n), actions, action_dtype, trial != 0,
                                 score_out, legal_out, static_cast<hipStream_t>(stream)));
    return G2048_OK;
}

int g2048_query(const g2048_engine *e, uint8_t *isend_out, uint8_t *highest_out, void *stream)
{
    if (int rc = usable(e))
        return rc;
    G2048_HIP(hipSetDevice(e->device));
    G2048_HIP(g2048::launch_query(e->st.boards, static_cast<uint32_t>(e->n), e->max_exp, isend_out, highest_out,
                                  static_cast<hipStream_t>(stream)));
    return G2048_OK;
}

int g2048_legal_actions(const g2048_engine *e, uint8_t *mask_out, void *stream)
{
    if (int rc = usable(e))
        return rc;
    if (!mask_out)
        return fail(G2048_ERR_INVALID, "NULL argument");
    G2048_HIP(hipSetDevice(e->device));
    G2048_HIP(g2048::launch_legal_mask(e->st.boards, static_cast<uint32_t>(e->n), mask_out, static_cast<hipStream_t>(stream)));
    return G2048_OK;
}

int g2048_add_tile(g2048_engine *e, uint32_t slot, void *stream)
{
    if (int rc = usable(e))
        return rc;
    G2048_HIP(hipSetDevice(e->device));
    e->fresh = 0;
    const g2048::StepArgs a = make_args(e, nullptr, 0);
    if (e->st.rng)
        G2048_HIP(g2048::launch_add_tile_numpy(a, static_cast<hipStream_t>(stream)));
    else
        G2048_HIP(g2048::launch_add_tile(a, slot, static_cast<hipStream_t>(stream)));
    return G2048_OK;
}

int g2048_fill_random_actions(const g2048_engine *e, uint64_t t_first, uint32_t k_steps, uint8_t *out, void *stream)
{
    if (int rc = usable(e))
        return rc;
    if (!out)
        return fail(G2048_ERR_INVALID, "NULL argument");
    G2048_HIP(hipSetDevice(e->device));
    G2048_HIP(g2048::launch_fill_actions(out, static_cast<uint32_t>(e->n), static_cast<uint32_t>(e->board_offset),
                                         static_cast<uint32_t>(e->seed), static_cast<uint32_t>(e->seed >> 32),
                                         t_first, k_steps, static_cast<hipStream_t>(stream)));
    return G2048_OK;
}

int g2048_onehot(const g2048_engine *e, void *out, int32_t obs_dtype, void *stream)
{
    if (int rc = usable(e))
        return rc;
    if (!out)
        return fail(G2048_ERR_INVALID, "NULL argument");
    if (obs_dtype < G2048_OBS_U8 || obs_dtype > G2048_OBS_F32)
        return fail(G2048_ERR_INVALID, "unknown obs_dtype %d", obs_dtype);
    if (reinterpret_cast<uintptr_t>(out) & 15u)
        return fail(G2048_ERR_INVALID, "one-hot output must be 16-byte aligned");
    G2048_HIP(hipSetDevice(e->device));
    G2048_HIP(g2048::launch_onehot(e->st.boards, static_cast<uint32_t>(e->n), out, obs_dtype,
                                   static_cast<hipStream_t>(stream)));
    return G2048_OK;
}

static int copy_out(const g2048_engine *e, void *dst, const void *src, size_t bytes, void *stream)
{
    if (int rc = usable(e))
        return rc;
    if (!dst)
        return fail(G2048_ERR_INVALID, "NULL argument");
    G2048_HIP(hipSetDevice(e->device));
    G2048_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, static_cast<hipStream_t>(stream)));
    G2048_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return G2048_OK;
}

static int copy_in(g2048_engine *e, void *dst, const void *src, size_t bytes, void *stream)
{
    if (int rc = usable(e))
        return rc;
    if (!src)
        return fail(G2048_ERR_INVALID, "NULL argument");
    G2048_HIP(hipSetDevice(e->device));
    G2048_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, static_cast<hipStream_t>(stream)));
    G2048_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return G2048_OK;
}

// ------------------------------------------------------------------------- host-resident I/O
static int ensure_done_word(g2048_engine *e)
{
    if (e->done_host)
        return G2048_OK;
    void *host = nullptr, *dev = nullptr;
    hipError_t err = hipHostMalloc(&host, 64, hipHostMallocMapped | hipHostMallocCoherent);
    if (err != hipSuccess)
        return fail(G2048_ERR_NOMEM, "hipHostMalloc(64) for the completion word failed: %s", hipGetErrorString(err));
    err = hipHostGetDevicePointer(&dev, host, 0);
    if (err != hipSuccess) {
        (void)hipHostFree(host);
        return fail(G2048_ERR_HIP, "hipHostGetDevicePointer failed: %s", hipGetErrorString(err));
    }
    std::memset(host, 0, 64);
    e->done_host = static_cast<unsigned long long *>(host);
    e->done_dev = static_cast<unsigned long long *>(dev);
    return G2048_OK;
}

static int ensure_host_io(g2048_engine *e)
{
    if (int rc = ensure_done_word(e))
        return rc;
    if (e->host_base)
        return G2048_OK;
    const size_t n = e->n;
    auto up = [](size_t x) { return (x + 63) & ~static_cast<size_t>(63); };
    const size_t off_act = 0, off_rew = up(off_act + 8 * n), off_term = up(off_rew + 4 * n), off_ill = up(off_term + n),
                 off_high = up(off_ill + n), off_boards = up(off_high + n), off_tb = up(off_boards + 16 * n),
                 off_sc = up(off_tb + 16 * n), bytes = up(off_sc + 4 * n);
    void *host = nullptr, *dev = nullptr;
    hipError_t err = hipHostMalloc(&host, bytes, hipHostMallocMapped | hipHostMallocCoherent);
    if (err != hipSuccess)
        return fail(G2048_ERR_NOMEM, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(err));
    err = hipHostGetDevicePointer(&dev, host, 0);
    if (err != hipSuccess) {
        (void)hipHostFree(host);
        return fail(G2048_ERR_HIP, "hipHostGetDevicePointer failed: %s", hipGetErrorString(err));
    }
    std::memset(host, 0, bytes);
    auto fill = [&](g2048_host_io &io, char *b) {
        io.actions = reinterpret_cast<int64_t *>(b + off_act);
        io.reward = reinterpret_cast<float *>(b + off_rew);
        io.terminated = reinterpret_cast<uint8_t *>(b + off_term);
        io.illegal = reinterpret_cast<uint8_t *>(b + off_ill);
        io.highest = reinterpret_cast<uint8_t *>(b + off_high);
        io.boards = reinterpret_cast<uint8_t *>(b + off_boards);
        io.terminal_boards = reinterpret_cast<uint8_t *>(b + off_tb);
        io.scores = reinterpret_cast<int32_t *>(b + off_sc);
    };
    fill(e->host_io, static_cast<char *>(host));
    fill(e->host_io_dev, static_cast<char *>(dev));
    e->host_base = host;
    return G2048_OK;
}

// A launch whose completion the host will POLL for must really execute: on a capturing stream it would only be
// recorded and the poll would never end.
static int refuse_capture(hipStream_t s, const char *what)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) {
        (void)hipGetLastError();
        return G2048_OK; // cannot tell (e.g. legacy stream semantics): go on
    }
    if (st != hipStreamCaptureStatusNone)
        return fail(G2048_ERR_INVALID, "%s blocks until the device has finished; it cannot be used on a capturing stream", what);
    return G2048_OK;
}

static double wait_limit_seconds()
{
    static const double limit = [] {
        const char *v = std::getenv("G2048_WAIT_TIMEOUT_S");
        const double x = v ? std::atof(v) : 0.0;
        return x > 0.0 ? x : 120.0;
    }();
    return limit;
}

// Poll the completion word (published by the device with system-scope release; tickets on one stream only grow) until
// it reaches `want`.  Liveness: after ~2 s without the word the stream is queried every ~2 s -- a failed stream ends the
// wait with its error, a stream that has drained without publishing is a bug and says so -- and an overall deadline
// (G2048_WAIT_TIMEOUT_S, default 120 s) returns an error instead of spinning forever on a hung device.
static int wait_done(g2048_engine *e, unsigned long long want, hipStream_t s)
{
    using clock = std::chrono::steady_clock;
    clock::time_point started{}, last_check{};
    bool armed = false;
    for (uint64_t spins = 0;; ++spins) {
        if (__atomic_load_n(e->done_host, __ATOMIC_ACQUIRE) >= want)
            return G2048_OK;
        cpu_relax();
        if ((spins & 0xfffffu) == 0xfffffu) { // every ~million polls
            const clock::time_point now = clock::now();
            if (!armed) {
                armed = true;
                started = last_check = now;
            } else if (now - last_check > std::chrono::seconds(2)) {
                last_check = now;
                const hipError_t q = hipStreamQuery(s);
                if (q == hipSuccess) // everything on the stream has finished: the word must be there
                    return __atomic_load_n(e->done_host, __ATOMIC_ACQUIRE) >= want
                               ? G2048_OK
                               : fail(G2048_ERR_HIP, "the device finished without publishing the completion word");
                if (q != hipErrorNotReady)
                    return fail(G2048_ERR_HIP, "stream failed while the host was waiting for the device: %s", hipGetErrorString(q));
                if (std::chrono::duration<double>(now - started).count() > wait_limit_seconds())
                    return fail(G2048_ERR_HIP, "no completion after %.0f s (G2048_WAIT_TIMEOUT_S): the device looks hung",
                                wait_limit_seconds());
            }
        }
    }
}

int g2048_host_io_map(g2048_engine *e, g2048_host_io *out)
{
    if (int rc = usable(e))
        return rc;
    if (!out)
        return fail(G2048_ERR_INVALID, "NULL argument");
    G2048_HIP(hipSetDevice(e->device));
    if (int rc = ensure_host_io(e))
        return rc;
    *out = e->host_io;
    return G2048_OK;
}

int g2048_step_host(g2048_engine *e, int auto_reset, void *stream)
{
    if (int rc = usable(e))
        return rc;
    if (!e->host_base)
        return fail(G2048_ERR_INVALID, "call g2048_host_io_map first (the actions are read from its buffer)");
    G2048_HIP(hipSetDevice(e->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc = refuse_capture(s, "g2048_step_host"))
        return rc;
    if (e->strict_actions) { // the actions are HOST memory here: refused before anything is stepped
        const int64_t *acts = e->host_io.actions;
        for (uint64_t i = 0; i < e->n; ++i)
            if (acts[i] < 0 || acts[i] > 3)
                return fail(G2048_ERR_INVALID, "strict actions: action %lld of board %llu is outside 0..3 (game2048_env.py:49 "
                                               "Discrete(4)); nothing was stepped",
                            static_cast<long long>(acts[i]), static_cast<unsigned long long>(e->board_offset + i));
    }
    const g2048_host_io &d = e->host_io_dev;
    g2048_step_io io{};
    io.actions = d.actions;
    io.action_dtype = G2048_ACT_I64;
    io.reward = d.reward;
    io.terminated = d.terminated;
    io.illegal = d.illegal;
    io.highest = d.highest;
    io.terminal_boards = d.terminal_boards;
    io.boards_out = d.boards;
    e->t += 1;
    e->fresh = 0;
    g2048::StepArgs a = make_args(e, &io, auto_reset);
    const unsigned long long want = ++e->done_count;
    if (e->st.rng) { // numpy-RNG mode: step (+ compacted resets), then boards + scores + the completion word
        a.boards_out = nullptr;
        G2048_HIP(g2048::launch_step_numpy(a, io.action_dtype, s));
        G2048_HIP(g2048::launch_fetch(e->st.boards, a.n, reinterpret_cast<uint4 *>(d.boards), d.scores, e->done_dev, want, s));
    } else {
        a.done_seq = e->done_dev;
        a.done_value = want;
        G2048_HIP(g2048::launch_step(a, io.action_dtype, s));
    }
    return wait_done(e, want, s);
}

int g2048_fetch_host(g2048_engine *e, void *stream)
{
    if (int rc = usable(e))
        return rc;
    if (!e->host_base)
        return fail(G2048_ERR_INVALID, "call g2048_host_io_map first");
    G2048_HIP(hipSetDevice(e->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc = refuse_capture(s, "g2048_fetch_host"))
        return rc;
    const unsigned long long want = ++e->done_count;
    G2048_HIP(g2048::launch_fetch(e->st.boards, static_cast<uint32_t>(e->n), reinterpret_cast<uint4 *>(e->host_io_dev.boards),
                                  e->host_io_dev.scores, e->done_dev, want, s));
    return wait_done(e, want, s);
}

int g2048_stream_signal(g2048_engine *e, void *stream, uint64_t *ticket)
{
    if (int rc = usable(e))
        return rc;
    if (!ticket)
        return fail(G2048_ERR_INVALID, "NULL argument");
    G2048_HIP(hipSetDevice(e->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc = refuse_capture(s, "g2048_stream_signal"))
        return rc;
    if (int rc = ensure_done_word(e))
        return rc;
    const unsigned long long want = ++e->done_count;
    G2048_HIP(g2048::launch_signal(e->done_dev, want, s));
    *ticket = want;
    return G2048_OK;
}

int g2048_stream_wait(g2048_engine *e, uint64_t ticket, void *stream)
{
    if (int rc = usable(e))
        return rc;
    if (!e->done_host || ticket == 0 || ticket > e->done_count)
        return fail(G2048_ERR_INVALID, "ticket %llu was not issued by g2048_stream_signal on this engine",
                    (unsigned long long)ticket);
    return wait_done(e, ticket, static_cast<hipStream_t>(stream));
}

// Is p device-accessible memory of this process (hipMalloc / torch tensor)?  Plain host memory is not
// known to the runtime and makes hipPointerGetAttributes fail.
static bool is_device_ptr(const void *p)
{
    hipPointerAttribute_t attr{};
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError(); // clear the sticky error of the failed query
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

static int ensure_scratch(g2048_engine *e)
{
    if (e->scratch)
        return G2048_OK;
    hipError_t err = hipMalloc(&e->scratch, e->n * 16);
    if (err != hipSuccess) {
        e->scratch = nullptr;
        return fail(G2048_ERR_NOMEM, "hipMalloc(%zu) for the host staging buffer failed: %s", (size_t)(e->n * 16),
                    hipGetErrorString(err));
    }
    return G2048_OK;
}

static int ensure_returns(g2048_engine *e)
{
    if (e->returns)
        return G2048_OK;
    hipError_t err = hipMalloc(reinterpret_cast<void **>(&e->returns), e->n * 4);
    if (err != hipSuccess) {
        e->returns = nullptr;
        return fail(G2048_ERR_NOMEM, "hipMalloc(%zu) for the all-gather send buffer failed: %s", (size_t)(e->n * 4),
                    hipGetErrorString(err));
    }
    return G2048_OK;
}

// Engine-less entry points (g2048_augment, g2048_canonicalize) launch on the device their buffers live on, whatever
// the caller's current device is, and put the caller's device back when they return.  Every buffer must be reachable
// from that one device: device or managed memory OF that device, or pinned / registered host memory (which has a
// device address everywhere).  Plain pageable host memory is refused.
struct DeviceScope {
    int previous = -1;
    ~DeviceScope()
    {
        if (previous >= 0)
            (void)hipSetDevice(previous);
    }
};

static int enter_device_of(DeviceScope &scope, const void *const *bufs, int n_bufs)
{
    int target = -1;
    for (int k = 0; k < n_bufs; ++k) {
        if (!bufs[k])
            continue;
        hipPointerAttribute_t attr{};
        if (hipPointerGetAttributes(&attr, bufs[k]) != hipSuccess) {
            (void)hipGetLastError();
            return fail(G2048_ERR_INVALID, "buffer %p is not memory the device can reach (pageable host memory?)", bufs[k]);
        }
        if (attr.type == hipMemoryTypeHost) {
            if (!attr.devicePointer)
                return fail(G2048_ERR_INVALID, "host buffer %p has no device address (allocate it pinned and mapped)", bufs[k]);
            continue; // reachable from every device
        }
        if (attr.type != hipMemoryTypeDevice && attr.type != hipMemoryTypeManaged) // e.g. hipMemoryTypeUnregistered
            return fail(G2048_ERR_INVALID, "buffer %p is not memory the device can reach (pageable host memory?)", bufs[k]);
        if (target < 0)
            target = attr.device;
        else if (attr.device != target)
            return fail(G2048_ERR_INVALID, "buffers live on different devices (%d and %d)", target, attr.device);
    }
    int current = 0;
    G2048_HIP(hipGetDevice(&current));
    if (target >= 0 && target != current) {
        G2048_HIP(hipSetDevice(target));
        scope.previous = current;
    }
    return G2048_OK;
}

// The engine keeps RECORDS (cells + packed score deficit); the plain views are produced / consumed by
// small kernels, through the staging buffer when the caller's buffer is host memory.
int g2048_get_boards(const g2048_engine *ce, uint8_t *buf, void *stream)
{
    g2048_engine *e = const_cast<g2048_engine *>(ce);
    if (int rc = usable(e))
        return rc;
    if (!buf)
        return fail(G2048_ERR_INVALID, "NULL argument");
    G2048_HIP(hipSetDevice(e->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t n = static_cast<uint32_t>(e->n);
    if (is_device_ptr(buf)) {
        if (reinterpret_cast<uintptr_t>(buf) & 15u)
            return fail(G2048_ERR_INVALID, "device board buffers must be 16-byte aligned");
        G2048_HIP(g2048::launch_export_boards(e->st.boards, n, reinterpret_cast<uint4 *>(buf), s));
        return G2048_OK; // device destination: ready in stream order
    }
    if (int rc = ensure_scratch(e))
        return rc;
    G2048_HIP(g2048::launch_export_boards(e->st.boards, n, static_cast<uint4 *>(e->scratch), s));
    return copy_out(e, buf, e->scratch, e->n * 16, stream);
}

int g2048_set_boards(g2048_engine *e, const uint8_t *buf, void *stream)
{
    if (int rc = usable(e))
        return rc;
    if (!buf)
        return fail(G2048_ERR_INVALID, "NULL argument");
    G2048_HIP(hipSetDevice(e->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t n = static_cast<uint32_t>(e->n);
    const uint4 *src = reinterpret_cast<const uint4 *>(buf);
    if (!is_device_ptr(buf)) {
        if (int rc = ensure_scratch(e))
            return rc;
        if (int rc = copy_in(e, e->scratch, buf, e->n * 16, stream))
            return rc;
        src = static_cast<const uint4 *>(e->scratch);
    } else if (reinterpret_cast<uintptr_t>(buf) & 15u) {
        return fail(G2048_ERR_INVALID, "device board buffers must be 16-byte aligned");
    }
    G2048_HIP(g2048::launch_import_boards(e->st.boards, n, src, s));
    if (src == e->scratch) // the staging buffer must be free again when this returns
        G2048_HIP(hipStreamSynchronize(s));
    return G2048_OK;
}

int g2048_get_scores(const g2048_engine *ce, int32_t *buf, void *stream)
{
    g2048_engine *e = const_cast<g2048_engine *>(ce);
    if (int rc = usable(e))
        return rc;
    if (!buf)
        return fail(G2048_ERR_INVALID, "NULL argument");
    G2048_HIP(hipSetDevice(e->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t n = static_cast<uint32_t>(e->n);
    if (is_device_ptr(buf)) {
        if (reinterpret_cast<uintptr_t>(buf) & 3u)
            return fail(G2048_ERR_INVALID, "device score buffers must be 4-byte aligned");
        G2048_HIP(g2048::launch_export_scores(e->st.boards, n, buf, s));
        return G2048_OK; // device destination: ready in stream order
    }
    if (int rc = ensure_scratch(e))
        return rc;
    G2048_HIP(g2048::launch_export_scores(e->st.boards, n, static_cast<int32_t *>(e->scratch), s));
    return copy_out(e, buf, e->scratch, e->n * 4, stream);
}

int g2048_set_scores(g2048_engine *e, const int32_t *buf, void *stream)
{
    if (int rc = usable(e))
        return rc;
    if (!buf)
        return fail(G2048_ERR_INVALID, "NULL argument");
    G2048_HIP(hipSetDevice(e->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t n = static_cast<uint32_t>(e->n);
    const int32_t *src = buf;
    if (!is_device_ptr(buf)) {
        for (uint64_t i = 0; i < e->n; ++i)
            if (buf[i] < 0 || buf[i] > 0x00ffffff)
                return fail(G2048_ERR_INVALID, "score %d of board %llu is outside 0 .. 2^24-1", buf[i], (unsigned long long)i);
        if (int rc = ensure_scratch(e))
            return rc;
        if (int rc = copy_in(e, e->scratch, buf, e->n * 4, stream))
            return rc;
        src = static_cast<const int32_t *>(e->scratch);
    } else if (reinterpret_cast<uintptr_t>(buf) & 3u) {
        return fail(G2048_ERR_INVALID, "device score buffers must be 4-byte aligned");
    }
    G2048_HIP(g2048::launch_import_scores(e->st.boards, n, src, e->st.ep_counters, s));
    if (src == e->scratch)
        G2048_HIP(hipStreamSynchronize(s));
    return G2048_OK;
}

int g2048_get_last_scores(const g2048_engine *ce, int32_t *buf, void *stream)
{
    g2048_engine *e = const_cast<g2048_engine *>(ce);
    if (int rc = usable(e))
        return rc;
    if (!buf)
        return fail(G2048_ERR_INVALID, "NULL argument");
    if (int rc = need_last_records(e, "g2048_get_last_scores"))
        return rc;
    G2048_HIP(hipSetDevice(e->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t n = static_cast<uint32_t>(e->n);
    if (is_device_ptr(buf)) {
        if (reinterpret_cast<uintptr_t>(buf) & 3u)
            return fail(G2048_ERR_INVALID, "device score buffers must be 4-byte aligned");
        G2048_HIP(g2048::launch_export_last_scores(e->st, n, buf, s));
        return G2048_OK; // device destination: ready in stream order
    }
    if (int rc = ensure_scratch(e))
        return rc;
    G2048_HIP(g2048::launch_export_last_scores(e->st, n, static_cast<int32_t *>(e->scratch), s));
    return copy_out(e, buf, e->scratch, e->n * 4, stream);
}

void *g2048_records_ptr(const g2048_engine *e) { return e ? e->st.boards : nullptr; }
void *g2048_last_records_ptr(const g2048_engine *e) { return e && e->track_last ? e->st.last_record : nullptr; }

int g2048_episode_stats(const g2048_engine *e, g2048_stats *out, void *stream)
{
    if (int rc = usable(e))
        return rc;
    if (!out)
        return fail(G2048_ERR_INVALID, "NULL argument");
    G2048_HIP(hipSetDevice(e->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    G2048_HIP(g2048::launch_stats(tracked_state(e), static_cast<uint32_t>(e->n), e->stats_partials, e->summary_scratch, e->stats_dev, false, s));
    g2048::StatsOut h{};
    G2048_HIP(hipMemcpyAsync(&h, e->stats_dev, sizeof h, hipMemcpyDeviceToHost, s));
    G2048_HIP(hipStreamSynchronize(s));
    out->episodes = h.episodes;
    out->illegal_ends = h.illegal_ends;
    out->last_count = h.last_count;
    out->last_score_sum = static_cast<int64_t>(h.last_score_sum);
    out->last_score_max = h.last_score_max;
    out->max_exp = h.max_exp;
    for (int k = 0; k < 32; ++k)
        out->highest_hist[k] = h.highest_hist[k];
    out->return_sum = h.return_sum;
    return G2048_OK;
}

static int stats_async(const g2048_engine *e, g2048_stats *device_out, bool returns_only, void *stream)
{
    static_assert(sizeof(g2048_stats) == sizeof(g2048::StatsOut), "g2048_stats and the kernel's StatsOut must share one layout");
    if (int rc = usable(e))
        return rc;
    if (!device_out)
        return fail(G2048_ERR_INVALID, "NULL argument");
    if (!is_device_ptr(device_out))
        return fail(G2048_ERR_INVALID, "the asynchronous statistics need a DEVICE buffer (use g2048_episode_stats for a host struct)");
    G2048_HIP(hipSetDevice(e->device));
    G2048_HIP(g2048::launch_stats(tracked_state(e), static_cast<uint32_t>(e->n), e->stats_partials, e->summary_scratch,
                                  reinterpret_cast<g2048::StatsOut *>(device_out),
                                  returns_only, static_cast<hipStream_t>(stream)));
    return G2048_OK;
}

int g2048_episode_stats_async(const g2048_engine *e, g2048_stats *device_out, void *stream)
{
    return stats_async(e, device_out, false, stream);
}

int g2048_returns_summary_async(const g2048_engine *e, g2048_stats *device_out, void *stream)
{
    return stats_async(e, device_out, true, stream);
}

// numpy-RNG mode state: the five PCG64 planes plus the per-wavefront lists of finished boards, one allocation
static int ensure_numpy_rng(g2048_engine *e)
{
    if (e->st.rng)
        return G2048_OK;
    void *p = nullptr;
    const size_t bytes = g2048::numpy_rng_bytes(e->n);
    hipError_t err = hipMalloc(&p, bytes);
    if (err != hipSuccess)
        return fail(G2048_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(err));
    e->st.rng = static_cast<uint64_t *>(p);
    return G2048_OK;
}

int g2048_set_numpy_rng(g2048_engine *e, const uint64_t *planes, void *stream)
{
    if (int rc = usable(e))
        return rc;
    G2048_HIP(hipSetDevice(e->device));
    if (!planes) { // back to the spawn stream
        if (e->st.rng)
            G2048_HIP(hipFree(e->st.rng));
        e->st.rng = nullptr;
        return G2048_OK;
    }
    if (int rc = ensure_numpy_rng(e))
        return rc;
    return copy_in(e, e->st.rng, planes, e->n * 40, stream);
}

int g2048_seed_numpy(g2048_engine *e, uint64_t base_seed, void *stream)
{
    if (int rc = g2048_seed(e, base_seed, stream))
        return rc;
    if (int rc = ensure_numpy_rng(e))
        return rc;
    G2048_HIP(g2048::launch_seed_numpy(e->st.rng, static_cast<uint32_t>(e->n), base_seed + e->board_offset,
                                       static_cast<hipStream_t>(stream)));
    return G2048_OK;
}

int g2048_get_numpy_rng(const g2048_engine *e, uint64_t *planes, void *stream)
{
    if (!e || !e->st.rng)
        return fail(G2048_ERR_INVALID, "engine is not in numpy-RNG mode");
    return copy_out(e, planes, e->st.rng, e->n * 40, stream);
}

int g2048_augment(const uint8_t *boards, const uint8_t *next_boards, const uint8_t *actions, uint64_t n,
                  uint8_t *boards_out, uint8_t *next_out, uint8_t *actions_out, void *stream)
{
    if (!boards || !actions || !boards_out || !actions_out || (next_boards && !next_out))
        return fail(G2048_ERR_INVALID, "NULL argument");
    if (n > 0x1fffffffull)
        return fail(G2048_ERR_INVALID, "n too large");
    if (n == 0)
        return G2048_OK;
    DeviceScope scope;
    const void *bufs[] = {boards, next_boards, actions, boards_out, next_out, actions_out};
    if (int rc = enter_device_of(scope, bufs, 6))
        return rc;
    G2048_HIP(g2048::launch_augment(reinterpret_cast<const uint4 *>(boards), reinterpret_cast<const uint4 *>(next_boards),
                                    actions, static_cast<uint32_t>(n), reinterpret_cast<uint4 *>(boards_out),
                                    reinterpret_cast<uint4 *>(next_out), actions_out, static_cast<hipStream_t>(stream)));
    return G2048_OK;
}

uint64_t g2048_state_bytes(const g2048_engine *e)
{
    return e ? sizeof(StateHeader) + e->slab_bytes + (e->st.rng ? e->n * 40 : 0) : 0;
}

int g2048_get_state(const g2048_engine *e, void *host_buf, void *stream)
{
    if (int rc = usable(e))
        return rc;
    if (!host_buf)
        return fail(G2048_ERR_INVALID, "NULL argument");
    StateHeader h{kStateMagic, e->n, e->seed, e->board_offset, e->t, e->fresh, e->max_exp, e->illegal_reward,
                  (e->st.rng ? 1u : 0u) | (e->track_last ? 0u : 2u)}; // flags: 1 = numpy-RNG planes follow, 2 = no terminal records
    std::memcpy(host_buf, &h, sizeof h);
    char *body = static_cast<char *>(host_buf) + sizeof h;
    if (int rc = copy_out(e, body, e->slab, e->slab_bytes, stream))
        return rc;
    if (e->st.rng)
        return copy_out(e, body + e->slab_bytes, e->st.rng, e->n * 40, stream);
    return G2048_OK;
}

int g2048_set_state(g2048_engine *e, const void *host_buf, uint64_t blob_bytes, void *stream)
{
    if (int rc = usable(e))
        return rc;
    if (!host_buf)
        return fail(G2048_ERR_INVALID, "NULL argument");
    if (blob_bytes < sizeof(StateHeader))
        return fail(G2048_ERR_INVALID, "state blob is too short (%llu bytes)", (unsigned long long)blob_bytes);
    StateHeader h;
    std::memcpy(&h, host_buf, sizeof h);
    if (h.magic != kStateMagic || h.n != e->n)
        return fail(G2048_ERR_INVALID, "state blob does not match this engine (magic %llx%s, n %llu vs %llu)",
                    (unsigned long long)h.magic, h.magic == 0x3376383430324700ull ? " = a blob written before ABI 14: its games were played "
                    "under the old spawn rule and cannot be continued under this one" :
                    (h.magic == 0x3476383430324700ull ? " = a blob written by ABI 14: same games, another slab layout -- save it again "
                    "with g2048_get_boards / g2048_get_scores and re-create" : ""),
                    (unsigned long long)h.n, (unsigned long long)e->n);
    const uint64_t want = sizeof(StateHeader) + e->slab_bytes + ((h.reserved & 1u) ? e->n * 40 : 0);
    if (blob_bytes != want)
        return fail(G2048_ERR_INVALID, "state blob is %llu bytes, this engine's state is %llu",
                    (unsigned long long)blob_bytes, (unsigned long long)want);
    if (h.max_exp > 31u || h.board_offset + h.n > 0x100000000ull)
        return fail(G2048_ERR_INVALID, "state blob header is corrupt (max_exp %u, board_offset %llu)", h.max_exp,
                    (unsigned long long)h.board_offset);
    e->seed = h.seed;
    e->board_offset = h.board_offset;
    e->t = h.t;
    e->fresh = h.fresh;
    e->max_exp = h.max_exp;
    e->illegal_reward = h.illegal_reward;
    e->track_last = (h.reserved & 2u) ? 0 : 1;
    const char *body = static_cast<const char *>(host_buf) + sizeof h;
    if (int rc = copy_in(e, e->slab, body, e->slab_bytes, stream))
        return rc;
    if (h.reserved & 1u) // the blob carries numpy-RNG planes
        return g2048_set_numpy_rng(e, reinterpret_cast<const uint64_t *>(body + e->slab_bytes), stream);
    return g2048_set_numpy_rng(e, nullptr, stream);
}

int g2048_canonicalize(uint8_t *boards, uint8_t *next_boards, uint8_t *actions, uint64_t n, uint8_t *symmetry_out,
                       void *stream)
{
    if (!boards)
        return fail(G2048_ERR_INVALID, "boards is NULL");
    if (n > 0xffffff00ull)
        return fail(G2048_ERR_INVALID, "n too large");
    if ((reinterpret_cast<uintptr_t>(boards) | reinterpret_cast<uintptr_t>(next_boards)) & 15u)
        return fail(G2048_ERR_INVALID, "board buffers must be 16-byte aligned");
    if (n == 0)
        return G2048_OK;
    DeviceScope scope;
    const void *bufs[] = {boards, next_boards, actions, symmetry_out};
    if (int rc = enter_device_of(scope, bufs, 4))
        return rc;
    G2048_HIP(g2048::launch_canonicalize(reinterpret_cast<uint4 *>(boards), reinterpret_cast<uint4 *>(next_boards), actions,
                                         static_cast<uint32_t>(n), symmetry_out, static_cast<hipStream_t>(stream)));
    return G2048_OK;
}

// ------------------------------------------------------------------------------- collective
// The path's only exchange step: the all-gather of episodic returns (scores of last_record) once per rollout.
// RCCL is bound lazily (dlopen) so that the library loads, and everything else works, on a box without
// RCCL or without a GPU; a process that never calls g2048_comm_* never touches it.
} // extern "C"

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;
std::mutex g_rccl_mutex;

int load_rccl()
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle)
        return G2048_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void *h = nullptr;
    for (const char *nm : names) {
        h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (h)
            break;
    }
    if (!h)
        return fail(G2048_ERR_HIP, "cannot load RCCL (librccl.so): %s", dlerror());
    Rccl r;
    r.handle = h;
#define G2048_SYM(field, name)                                                                                 \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name));                                             \
    if (!r.field)                                                                                              \
        return fail(G2048_ERR_HIP, "RCCL has no symbol %s", name);
    G2048_SYM(GetUniqueId, "ncclGetUniqueId")
    G2048_SYM(CommInitRank, "ncclCommInitRank")
    G2048_SYM(CommInitAll, "ncclCommInitAll")
    G2048_SYM(CommDestroy, "ncclCommDestroy")
    G2048_SYM(AllGather, "ncclAllGather")
    G2048_SYM(GroupStart, "ncclGroupStart")
    G2048_SYM(GroupEnd, "ncclGroupEnd")
    G2048_SYM(GetErrorString, "ncclGetErrorString")
#undef G2048_SYM
    g_rccl = r;
    return G2048_OK;
}

#define G2048_NCCL(call)                                                                                       \
    do {                                                                                                       \
        ncclResult_t r_ = (call);                                                                              \
        if (r_ != ncclSuccess)                                                                                 \
            return fail(G2048_ERR_HIP, "%s failed: %s", #call, g_rccl.GetErrorString(r_));                      \
    } while (0)

} // namespace

struct g2048_comm {
    ncclComm_t comm = nullptr;
    int world = 0, rank = 0, device = 0;
};

struct g2048_comm_local {
    int n = 0;
    int devices[G2048_COMM_LOCAL_MAX] = {};
    ncclComm_t comms[G2048_COMM_LOCAL_MAX] = {};
};

extern "C" {

int g2048_comm_unique_id(uint8_t id[G2048_COMM_ID_BYTES])
{
    static_assert(sizeof(ncclUniqueId) <= G2048_COMM_ID_BYTES, "ncclUniqueId does not fit G2048_COMM_ID_BYTES");
    if (!id)
        return fail(G2048_ERR_INVALID, "id is NULL");
    if (int rc = load_rccl())
        return rc;
    ncclUniqueId u;
    G2048_NCCL(g_rccl.GetUniqueId(&u));
    std::memset(id, 0, G2048_COMM_ID_BYTES);
    std::memcpy(id, &u, sizeof u);
    return G2048_OK;
}

int g2048_comm_create(int world, int rank, const uint8_t id[G2048_COMM_ID_BYTES], int device, g2048_comm **out)
{
    if (!out || !id)
        return fail(G2048_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world)
        return fail(G2048_ERR_INVALID, "rank %d / world %d out of range", rank, world);
    if (int rc = load_rccl())
        return rc;
    G2048_HIP(hipSetDevice(device));
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    g2048_comm *c = new (std::nothrow) g2048_comm;
    if (!c)
        return fail(G2048_ERR_NOMEM, "out of host memory");
    c->world = world;
    c->rank = rank;
    c->device = device;
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail(G2048_ERR_HIP, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
    }
    *out = c;
    return G2048_OK;
}

int g2048_comm_destroy(g2048_comm *c)
{
    if (!c)
        return G2048_OK;
    ncclResult_t r = c->comm ? g_rccl.CommDestroy(c->comm) : ncclSuccess;
    delete c;
    if (r != ncclSuccess)
        return fail(G2048_ERR_HIP, "ncclCommDestroy failed: %s", g_rccl.GetErrorString(r));
    return G2048_OK;
}

int g2048_allgather_returns(const g2048_engine *ce, g2048_comm *c, int32_t *out, void *stream)
{
    g2048_engine *e = const_cast<g2048_engine *>(ce);
    if (int rc = usable(e))
        return rc;
    if (!c || !out)
        return fail(G2048_ERR_INVALID, "NULL argument");
    if (c->device != e->device)
        return fail(G2048_ERR_INVALID, "communicator is on device %d, engine on device %d", c->device, e->device);
    if (int rc = need_last_records(e, "g2048_allgather_returns"))
        return rc;
    G2048_HIP(hipSetDevice(e->device));
    if (int rc = ensure_returns(e))
        return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // the returns are the scores of last_record: materialise int32[n] in the engine's send buffer, then ONE
    // all-gather (equal shards: rank r's n returns land at out[r * n]), all on the caller's stream
    int32_t *send = e->returns;
    G2048_HIP(g2048::launch_export_last_scores(e->st, static_cast<uint32_t>(e->n), send, s));
    G2048_NCCL(g_rccl.AllGather(send, out, e->n, ncclInt32, c->comm, s));
    return G2048_OK;
}

int g2048_allgather_summary(const g2048_engine *e, g2048_comm *c, g2048_stats *out, void *stream)
{
    static_assert(sizeof(g2048_stats) % 8 == 0, "g2048_stats is shipped as uint64 words");
    if (int rc = usable(e))
        return rc;
    if (!c || !out)
        return fail(G2048_ERR_INVALID, "NULL argument");
    if (c->device != e->device)
        return fail(G2048_ERR_INVALID, "communicator is on device %d, engine on device %d", c->device, e->device);
    if (!is_device_ptr(out))
        return fail(G2048_ERR_INVALID, "g2048_allgather_summary needs a DEVICE buffer of `world` g2048_stats");
    G2048_HIP(hipSetDevice(e->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    // this rank's summary goes straight into ITS row of the gathered array (one launch), then ONE in-place all-gather of
    // sizeof(g2048_stats) bytes per rank on the same stream: no send buffer, no stream hop, nothing for the host to wait for
    g2048_stats *mine = out + c->rank;
    G2048_HIP(g2048::launch_stats(tracked_state(e), static_cast<uint32_t>(e->n), e->stats_partials, e->summary_scratch,
                                  reinterpret_cast<g2048::StatsOut *>(mine), true, s));
    G2048_NCCL(g_rccl.AllGather(mine, out, sizeof(g2048_stats) / 8, ncclUint64, c->comm, s));
    return G2048_OK;
}

int g2048_comm_world(const g2048_comm *c) { return c ? c->world : 0; }
int g2048_comm_rank(const g2048_comm *c) { return c ? c->rank : -1; }

int g2048_comm_local_create(const int *devices, int n_devices, g2048_comm_local **out)
{
    if (!devices || !out)
        return fail(G2048_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (n_devices < 1 || n_devices > G2048_COMM_LOCAL_MAX)
        return fail(G2048_ERR_INVALID, "n_devices %d out of range 1..%d", n_devices, G2048_COMM_LOCAL_MAX);
    for (int a = 0; a < n_devices; ++a)
        for (int b = a + 1; b < n_devices; ++b)
            if (devices[a] == devices[b])
                return fail(G2048_ERR_INVALID, "device %d listed twice: one communicator per device", devices[a]);
    if (int rc = load_rccl())
        return rc;
    g2048_comm_local *c = new (std::nothrow) g2048_comm_local;
    if (!c)
        return fail(G2048_ERR_NOMEM, "out of host memory");
    c->n = n_devices;
    for (int r = 0; r < n_devices; ++r)
        c->devices[r] = devices[r];
    const ncclResult_t res = g_rccl.CommInitAll(c->comms, n_devices, c->devices); // the expensive part: done ONCE
    if (res != ncclSuccess) {
        delete c;
        return fail(G2048_ERR_HIP, "ncclCommInitAll failed: %s", g_rccl.GetErrorString(res));
    }
    *out = c;
    return G2048_OK;
}

int g2048_comm_local_destroy(g2048_comm_local *c)
{
    if (!c)
        return G2048_OK;
    ncclResult_t first = ncclSuccess;
    for (int r = 0; r < c->n; ++r) {
        const ncclResult_t res = c->comms[r] ? g_rccl.CommDestroy(c->comms[r]) : ncclSuccess;
        if (res != ncclSuccess && first == ncclSuccess)
            first = res;
    }
    delete c;
    if (first != ncclSuccess)
        return fail(G2048_ERR_HIP, "ncclCommDestroy failed: %s", g_rccl.GetErrorString(first));
    return G2048_OK;
}

int g2048_allgather_returns_local(g2048_comm_local *c, g2048_engine *const *engines, int32_t *const *outs,
                                  void *const *streams)
{
    if (!c || !engines || !outs)
        return fail(G2048_ERR_INVALID, "NULL argument");
    const int n_engines = c->n;
    for (int r = 0; r < n_engines; ++r) {
        if (!engines[r] || !outs[r])
            return fail(G2048_ERR_INVALID, "engine or output %d is NULL", r);
        if (engines[r]->device != c->devices[r])
            return fail(G2048_ERR_INVALID, "engine %d is on device %d, the communicator's slot %d on device %d", r,
                        engines[r]->device, r, c->devices[r]);
        if (int rc = need_last_records(engines[r], "g2048_allgather_returns_local"))
            return rc;
        if (engines[r]->n != engines[0]->n)
            return fail(G2048_ERR_INVALID, "engines must hold equal shards (engine %d has %llu boards, engine 0 %llu)", r,
                        (unsigned long long)engines[r]->n, (unsigned long long)engines[0]->n);
    }
    for (int r = 0; r < n_engines; ++r) {
        G2048_HIP(hipSetDevice(c->devices[r]));
        if (int rc = ensure_returns(engines[r]))
            return rc;
        G2048_HIP(g2048::launch_export_last_scores(engines[r]->st, static_cast<uint32_t>(engines[r]->n), engines[r]->returns,
                                                   static_cast<hipStream_t>(streams ? streams[r] : nullptr)));
    }
    ncclResult_t res = g_rccl.GroupStart();
    for (int r = 0; r < n_engines && res == ncclSuccess; ++r) {
        if (hipSetDevice(c->devices[r]) != hipSuccess) {
            res = ncclUnhandledCudaError;
            break;
        }
        res = g_rccl.AllGather(engines[r]->returns, outs[r], engines[r]->n, ncclInt32, c->comms[r],
                               static_cast<hipStream_t>(streams ? streams[r] : nullptr));
    }
    const ncclResult_t end = g_rccl.GroupEnd();
    if (res == ncclSuccess)
        res = end;
    if (res != ncclSuccess)
        return fail(G2048_ERR_HIP, "RCCL all-gather failed: %s", g_rccl.GetErrorString(res));
    return G2048_OK; // enqueued: outs[r] is complete when streams[r] reaches this point
}

} // extern "C"
