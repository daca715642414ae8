// lane16.hip -- the mapping north_star sketches, measured (tool, not product): 16 LANES PER BOARD (one cell per
// lane, 4 boards per wavefront), the board staged through LDS, wavefront ballots + prefix counts for the
// slide/merge compaction and for the empty-cell selection, one Philox block per board (computed by all 16
// lanes of the board -- SIMT executes it for the whole wavefront either way).  It plays the same games as the
// product kernel (cross-checked here on boards / reward / terminated); it keeps NO score state at all (no
// deficit, no last_record, no counters), i.e. it is spared work the product does.
// Usage: lane16 [log2_boards] [rounds]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../gym-2048_amd/csrc/g2048_kernels.hip"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

using namespace g2048;

// 256 lanes = 16 boards per block; LDS: 16 B per board for staging + 16 B per board for the compaction
__global__ void __launch_bounds__(256) lane16_kernel(uint4 *boards, const uint8_t *actions, float *reward, uint8_t *terminated,
                                                     uint32_t n, uint32_t seed_lo, uint32_t seed_hi, uint32_t t_lo)
{
    __shared__ uint8_t s_stage[16][16];
    __shared__ uint8_t s_line[16][4][4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t lb = tid >> 4;              // board within the block
    const uint32_t c = tid & 15u;              // my cell (row-major)
    const uint32_t board = blockIdx.x * 16u + lb;
    const bool valid = board < n;
    const uint32_t bi = valid ? board : n - 1u;
    // ---- stage the board in LDS: the cell-0 lane loads 16 B, everybody picks up its byte
    if (c == 0u)
        *reinterpret_cast<uint4 *>(&s_stage[lb][0]) = boards[bi];
    const uint32_t action = actions[bi] & 3u;
    const Words w = philox4x32_10(t_lo, 0u, bi, 0u, seed_lo, seed_hi);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t v = s_stage[lb][c] & 31u;
    // ---- my line and my position in shift order (game2048_env.py:210-219,230-231)
    const uint32_t row = c >> 2, col = c & 3u;
    const bool hz = action & 1u, rev = ((action ^ (action >> 1)) & 1u) != 0;
    const uint32_t L = hz ? row : col, q = hz ? col : row, P = rev ? 3u - q : q;
    // lanes of my line, in shift order
    const uint32_t base = lane & ~15u;
    uint32_t rank = 0, n_nonzero = 0;
    const unsigned long long nz = __ballot(v != 0u);
#pragma unroll
    for (uint32_t p = 0; p < 4; ++p) {
        const uint32_t qq = rev ? 3u - p : p;
        const uint32_t cell = hz ? 4u * L + qq : 4u * qq + L;
        const uint32_t bit = static_cast<uint32_t>(nz >> (base + cell)) & 1u;
        rank += (p < P) ? bit : 0u;       // non-zero cells before me (:249-251)
        n_nonzero += bit;
    }
    // ---- compaction through LDS: clear the line, then every non-zero cell drops into slot `rank`
    s_line[lb][L][P] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (v != 0u)
        s_line[lb][L][rank] = static_cast<uint8_t>(v);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t packed = *reinterpret_cast<const uint32_t *>(&s_line[lb][L][0]);
    const uint32_t a = packed & 0xffu, b = (packed >> 8) & 0xffu, cc = (packed >> 16) & 0xffu, d = packed >> 24;
    // ---- merge (:252-255), every lane of the line computes the whole line and keeps its own slot
    const bool mab = a != 0u && a == b;
    const bool mbc = !mab && b != 0u && b == cc;
    const bool mcd = !mbc && cc != 0u && cc == d;
    uint32_t o0, o1, o2, o3;
    o0 = a + (mab ? 1u : 0u);
    if (mab) { o1 = mcd ? cc + 1u : cc; o2 = mcd ? 0u : d; o3 = 0u; }
    else if (mbc) { o1 = b + 1u; o2 = d; o3 = 0u; }
    else { o1 = b; o2 = mcd ? cc + 1u : cc; o3 = mcd ? 0u : d; }
    uint32_t nv = P == 0u ? o0 : (P == 1u ? o1 : (P == 2u ? o2 : o3));
    const uint32_t line_gain = (mab ? 2u << a : 0u) + (mbc ? 2u << b : 0u) + (mcd ? 2u << cc : 0u);
    // ---- per-board flags and sums with ballots / DPP-free LDS-free arithmetic on the 16-bit sub-masks
    const unsigned long long chg = __ballot(nv != v);
    const bool legal = ((chg >> base) & 0xffffull) != 0ull;
    // gain: sum over the 4 lines; the P == 0 lane of each line contributes
    uint32_t gain = 0;
#pragma unroll
    for (uint32_t l = 0; l < 4; ++l) {
        const uint32_t cell0 = hz ? 4u * l + (rev ? 3u : 0u) : (rev ? 12u : 0u) + l;     // the P == 0 lane of line l
        gain += __shfl(line_gain, static_cast<int>(base + cell0), 64);
    }
    // ---- spawn (:166-176): ballot of the empty cells, k-th empty by prefix count
    const unsigned long long em = __ballot(nv == 0u);
    const uint32_t emask = static_cast<uint32_t>(em >> base) & 0xffffu;
    const uint32_t n_empty = __popc(emask);
    // the spawn rule of ABI 14 (g2048.h "Randomness"): p = w * n; cell = p >> 32, a 2 iff (uint32_t)p <= 3865470566
    const unsigned long long prod = static_cast<unsigned long long>(w.w[0]) * n_empty;
    const uint32_t k = static_cast<uint32_t>(prod >> 32);
    const uint32_t my_rank = __popc(emask & ((1u << c) - 1u));
    if (legal && nv == 0u && my_rank == k)
        nv = (static_cast<uint32_t>(prod) <= kTwoThreshold) ? 1u : 2u;
    // ---- done detection (:262-280): full board without equal neighbours
    bool end = false;
    if (legal && n_empty == 1u) {
        const uint32_t right = __shfl(nv, static_cast<int>(lane + 1u), 64), down = __shfl(nv, static_cast<int>(lane + 4u), 64);
        const bool eq = (col < 3u && right == nv) || (row < 3u && down == nv);
        end = ((__ballot(eq) >> base) & 0xffffull) == 0ull;
    }
    const bool term = legal ? end : true;
    // ---- auto-reset (:102-111)
    if (term) {
        const uint32_t w1 = legal ? w.w[1] : w.w[0], w2 = legal ? w.w[2] : w.w[1];
        const uint32_t p1 = w1 >> 28, k2 = __umulhi(w2, 15u), p2 = k2 + (k2 >= p1 ? 1u : 0u);
        nv = c == p1 ? (fresh_is_four_16(w1) ? 2u : 1u) : (c == p2 ? (fresh_is_four_15(w2) ? 2u : 1u) : 0u);
    }
    // ---- back through LDS, the cell-0 lane stores the 16 bytes and the step outputs
    s_stage[lb][c] = static_cast<uint8_t>(nv);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (c == 0u && valid) {
        boards[board] = *reinterpret_cast<const uint4 *>(&s_stage[lb][0]);
        reward[board] = legal ? static_cast<float>(gain) : 0.0f;
        terminated[board] = term ? 1 : 0;
    }
}

int main(int argc, char **argv)
{
    const int lg = argc > 1 ? atoi(argv[1]) : 20;
    const int rounds = argc > 2 ? atoi(argv[2]) : 10;
    const uint32_t n = 1u << lg;
    const int launches = 32;
    StepArgs a{};
    uint4 *b16;
    CHECK(hipMalloc(&a.st.boards, (size_t)n * 16)); CHECK(hipMalloc(&b16, (size_t)n * 16));
    CHECK(hipMalloc(&a.st.last_record, (size_t)n * 16)); CHECK(hipMalloc(&a.st.ep_counters, (size_t)(n / 64 + 16) * 32));
    CHECK(hipMemset(a.st.last_record, 0, (size_t)n * 16)); CHECK(hipMemset(a.st.ep_counters, 0, (size_t)(n / 64 + 16) * 32));
    uint8_t *actions, *term, *term16; float *reward, *reward16;
    CHECK(hipMalloc(&actions, (size_t)n * launches)); CHECK(hipMalloc(&term, (size_t)n)); CHECK(hipMalloc(&term16, (size_t)n));
    CHECK(hipMalloc(&reward, (size_t)n * 4)); CHECK(hipMalloc(&reward16, (size_t)n * 4));
    a.n = n; a.seed_lo = 42; a.auto_reset = 1; a.reward = reward; a.terminated = term;
    CHECK(launch_fill_actions(actions, n, 0, 42, 0, 1, launches, 0));
    CHECK(launch_reset(a, 0, nullptr, 0));
    CHECK(launch_export_boards(a.st.boards, n, b16, 0));      // plain cells for the 16-lane kernel
    CHECK(hipDeviceSynchronize());
    // ---- cross-check over `launches` steps
    std::vector<uint32_t> h1((size_t)n * 4), h2((size_t)n * 4); std::vector<float> r1(n), r2(n); std::vector<uint8_t> t1(n), t2(n);
    uint4 *plain; CHECK(hipMalloc(&plain, (size_t)n * 16));
    bool same = true;
    for (int j = 0; j < launches; ++j) {
        a.t_lo = 1 + j; a.actions = actions + (size_t)j * n;
        CHECK(launch_step(a, 1, 0));
        hipLaunchKernelGGL(lane16_kernel, dim3((n + 15) / 16), dim3(256), 0, 0, b16, actions + (size_t)j * n, reward16, term16, n, 42u, 0u, 1u + j);
        if (j % 8 == 7) {
            CHECK(launch_export_boards(a.st.boards, n, plain, 0));
            CHECK(hipMemcpy(h1.data(), plain, (size_t)n * 16, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(h2.data(), b16, (size_t)n * 16, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(r1.data(), reward, (size_t)n * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(r2.data(), reward16, (size_t)n * 4, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(t1.data(), term, n, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(t2.data(), term16, n, hipMemcpyDeviceToHost));
            same = same && h1 == h2 && r1 == r2 && t1 == t2;
        }
    }
    printf("cross-check vs the product kernel over %d steps (boards, reward, terminated): %s\n", launches, same ? "equal" : "DIFFER");
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<float> u1, u2;
    for (int r = 0; r < rounds + 1; ++r) {
        CHECK(hipEventRecord(e0, 0));
        for (int j = 0; j < launches; ++j) { a.t_lo = 100 + j; a.actions = actions + (size_t)j * n; CHECK(launch_step(a, 1, 0)); }
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (r) u1.push_back(ms * 1e3f / launches);
        CHECK(hipEventRecord(e0, 0));
        for (int j = 0; j < launches; ++j)
            hipLaunchKernelGGL(lane16_kernel, dim3((n + 15) / 16), dim3(256), 0, 0, b16, actions + (size_t)j * n, reward16, term16, n, 42u, 0u, 100u + j);
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1)); if (r) u2.push_back(ms * 1e3f / launches);
    }
    std::sort(u1.begin(), u1.end()); std::sort(u2.begin(), u2.end());
    printf("boards 2^%d, %d rounds x %d launches; us per launch (median)\n", lg, rounds, launches);
    printf("product: one board per lane, records in registers            %9.2f  -> algorithmic %5.0f GB/s\n", u1[u1.size() / 2], 38.0 * n / (u1[u1.size() / 2] * 1e-6) / 1e9);
    printf("16 lanes per board, LDS staging, ballot/prefix (no score)    %9.2f  -> algorithmic %5.0f GB/s\n", u2[u2.size() / 2], 38.0 * n / (u2[u2.size() / 2] * 1e-6) / 1e9);
    return 0;
}
