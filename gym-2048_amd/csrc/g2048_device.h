// g2048_device.h -- per-lane 2048 board arithmetic for gfx950 (CDNA4), one board per lane.
//
// A board is 16 cells of int8 exponents (0 = empty, k = tile 2^k), row-major, held in four
// 32-bit VGPRs r[0..3] (r[i] byte j = cell (i, j)).  Every exponent is < 0x80 (really <= 17),
// which makes the classic "SIMD within a register" byte tricks carry-free.
//
// Reference semantics (cited as game2048_env.py:LINE = /root/reference/env/envs/game2048_env.py):
//   move   :194-241   four lines, each through shift(); direction 0 up, 1 right, 2 down, 3 left
//   shift  :243-260   compact non-zeros, merge equal neighbours once, leftmost first
//   add_tile :166-176 value then position (uniform over the empty cells)
//   isend  :262-280   max_tile reached, else any empty -> False, else no legal move
//   highest :190-192
//
// The slide/merge runs on all four lines at once: the board is re-expressed as four registers
// A,B,C,D where byte l of A is the FIRST cell of line l (in shift order), B the second, ... so
// one 32-bit VALU op advances four lines.  For vertical moves A..D are simply the rows (reversed
// for "down"); for horizontal moves they are the columns, obtained with an 8 x v_perm_b32 byte
// transpose.  No LDS, no cross-lane traffic: the whole step lives in ~40 VGPRs.
#pragma once

#include <stdint.h>

#if defined(G2048_HOST_CHECK)
// Host compilation of this header exists ONLY for tests/host_check (a g++-built unit test of the
// SWAR math against the oracle in this GPU-less container).  The product never runs this path.
#define G2048_DEV static inline
static inline uint32_t g2048_perm(uint32_t hi, uint32_t lo, uint32_t sel)
{
    uint64_t src = ((uint64_t)hi << 32) | lo;
    uint32_t out = 0;
    for (int k = 0; k < 4; ++k) {
        uint32_t s = (sel >> (8 * k)) & 0xff;
        uint32_t byte = s < 8 ? (uint32_t)((src >> (8 * s)) & 0xff) : (s == 12 ? 0u : 0xffu);
        out |= byte << (8 * k);
    }
    return out;
}
static inline uint32_t g2048_mulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
static inline uint32_t g2048_popc(uint32_t x) { return (uint32_t)__builtin_popcount(x); }
#else
#include <hip/hip_runtime.h>
#define G2048_DEV __device__ __forceinline__
G2048_DEV uint32_t g2048_perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
G2048_DEV uint32_t g2048_mulhi(uint32_t a, uint32_t b) { return __umulhi(a, b); }
G2048_DEV uint32_t g2048_popc(uint32_t x) { return (uint32_t)__popc(x); }
#endif

namespace g2048 {

struct Board {
    uint32_t r[4];
};

// ------------------------------------------------------------------------------------ Philox
// Philox4x32-10, constants as in rocrand_philox4x32_10.h:62-65.  The spawn stream:
//   word(seed, t, board, slot) = Philox(ctr = (t_lo, t_hi, board, slot >> 2), key = seed)[slot & 3]
// Every batched path needs slots 0..2 (+ word 3 for the synthetic random policy), i.e. ONE block
// per board per step and no per-board RNG state in HBM.
constexpr uint32_t kPhiloxM0 = 0xD2511F53u, kPhiloxM1 = 0xCD9E8D57u;
constexpr uint32_t kPhiloxW0 = 0x9E3779B9u, kPhiloxW1 = 0xBB67AE85u;

struct Words {
    uint32_t w[4];
};

G2048_DEV Words philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const uint64_t p0 = (uint64_t)kPhiloxM0 * c0;
        const uint64_t p1 = (uint64_t)kPhiloxM1 * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += kPhiloxW0;
        k1 += kPhiloxW1;
    }
    return Words{{c0, c1, c2, c3}};
}

// ------------------------------------------------------------------------- SWAR byte helpers
// All inputs have every byte < 0x80.
constexpr uint32_t kLow7 = 0x7f7f7f7fu, kHigh1 = 0x80808080u;

// 0x80 in every non-zero byte.
G2048_DEV uint32_t nz80(uint32_t x) { return (x + kLow7) & kHigh1; }
// 0x80 in every zero byte.
G2048_DEV uint32_t z80(uint32_t x) { return ~(x + kLow7) & kHigh1; }
// 0x80 flags -> 0x7f byte masks (enough to select bytes < 0x80).
G2048_DEV uint32_t mask7(uint32_t f80) { return f80 - (f80 >> 7); }
// (m & a) | (~m & b)
G2048_DEV uint32_t bfi(uint32_t m, uint32_t a, uint32_t b) { return (m & a) | (~m & b); }

// 4x4 byte transpose: rows -> columns (and back; it is an involution).
G2048_DEV Board transpose(const Board &b)
{
    const uint32_t x0 = g2048_perm(b.r[1], b.r[0], 0x05010400u);
    const uint32_t x1 = g2048_perm(b.r[1], b.r[0], 0x07030602u);
    const uint32_t x2 = g2048_perm(b.r[3], b.r[2], 0x05010400u);
    const uint32_t x3 = g2048_perm(b.r[3], b.r[2], 0x07030602u);
    Board t;
    t.r[0] = g2048_perm(x2, x0, 0x05040100u);
    t.r[1] = g2048_perm(x2, x0, 0x07060302u);
    t.r[2] = g2048_perm(x3, x1, 0x05040100u);
    t.r[3] = g2048_perm(x3, x1, 0x07060302u);
    return t;
}

// if a byte of x is zero, pull the byte of y into it (and clear it in y): one bubble step of the
// stable "zeros to the back" compaction (game2048_env.py:249-251 skips zeros).
G2048_DEV void pull(uint32_t &x, uint32_t &y)
{
    const uint32_t keep = mask7(nz80(x));
    x = bfi(keep, x, y);
    y &= keep;
}

// game2048_env.py:243-260 for four lines at once.  a,b,c,d: 1st..4th cell of each line.
// Returns the summed merge score of the four lines.
G2048_DEV uint32_t shift4(uint32_t &a, uint32_t &b, uint32_t &c, uint32_t &d)
{
    // -- compaction (odd-even bubble, 6 steps)
    pull(a, b);
    pull(c, d);
    pull(b, c);
    pull(a, b);
    pull(c, d);
    pull(b, c);
    // -- merge flags: a pair merges when equal and non-zero; leftmost first, each cell once
    const uint32_t eab = z80(a ^ b) & nz80(a);
    const uint32_t ebc = z80(b ^ c) & nz80(b) & ~eab;
    const uint32_t ecd = z80(c ^ d) & nz80(c) & ~ebc;
    const uint32_t iab = eab >> 7, ibc = ebc >> 7, icd = ecd >> 7; // +1 on the exponent
    const uint32_t mab = eab - iab, mbc = ebc - ibc, mcd = ecd - icd; // 0x7f masks
    const uint32_t a1 = a + iab, b1 = b + ibc, c1 = c + icd;
    // -- outputs
    const uint32_t o0 = a1;
    const uint32_t o1 = bfi(mab, c1, b1);
    const uint32_t o2 = bfi(mab, d & ~mcd, bfi(mbc, d, c1));
    const uint32_t o3 = d & ~(mab | mbc | mcd);
    // -- score: sum of 2^e over the merged cells (game2048_env.py:253-254)
    const uint32_t m1 = (a1 & mab) | (b1 & mbc); // first merge of each line (0 if none)
    const uint32_t m2 = c1 & mcd;                // second merge of each line
    uint32_t score = 0;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        score += 1u << ((m1 >> (8 * l)) & 0xff);
        score += 1u << ((m2 >> (8 * l)) & 0xff);
    }
    score -= 8u - g2048_popc(eab | ebc) - g2048_popc(ecd); // the "1 << 0" of non-merged bytes
    a = o0;
    b = o1;
    c = o2;
    d = o3;
    return score;
}

// game2048_env.py:194-241.  Returns true when the board changed (false = IllegalMove).
G2048_DEV bool move(Board &bd, uint32_t action, uint32_t &score)
{
    const bool horizontal = (action & 1u) != 0;          // :211 dir_mod_two
    const bool reversed = ((action ^ (action >> 1)) & 1u) != 0; // :212 shift_direction
    const Board t = transpose(bd);
    uint32_t l0 = horizontal ? t.r[0] : bd.r[0];
    uint32_t l1 = horizontal ? t.r[1] : bd.r[1];
    uint32_t l2 = horizontal ? t.r[2] : bd.r[2];
    uint32_t l3 = horizontal ? t.r[3] : bd.r[3];
    uint32_t a = reversed ? l3 : l0;
    uint32_t b = reversed ? l2 : l1;
    uint32_t c = reversed ? l1 : l2;
    uint32_t d = reversed ? l0 : l3;
    const uint32_t a0 = a, b0 = b, c0 = c, d0 = d;
    score = shift4(a, b, c, d);
    const bool changed = ((a ^ a0) | (b ^ b0) | (c ^ c0) | (d ^ d0)) != 0; // :222,234,238
    Board o;
    o.r[0] = reversed ? d : a;
    o.r[1] = reversed ? c : b;
    o.r[2] = reversed ? b : c;
    o.r[3] = reversed ? a : d;
    const Board ot = transpose(o);
    bd.r[0] = horizontal ? ot.r[0] : o.r[0];
    bd.r[1] = horizontal ? ot.r[1] : o.r[1];
    bd.r[2] = horizontal ? ot.r[2] : o.r[2];
    bd.r[3] = horizontal ? ot.r[3] : o.r[3];
    return changed;
}

// Number of empty cells.
G2048_DEV uint32_t count_empty(const Board &bd)
{
    return g2048_popc(z80(bd.r[0]) | (z80(bd.r[1]) >> 1) | (z80(bd.r[2]) >> 2) | (z80(bd.r[3]) >> 3));
}

// game2048_env.py:166-176 with the injected spawn word w: value 2 (exp 1) if (w & 0xffff) <= 58982
// else 4 (exp 2); position = k-th empty cell in row-major order, k = (w * n_empty) >> 32.
// Precondition: at least one empty cell.
G2048_DEV void add_tile(Board &bd, uint32_t w)
{
    const uint32_t z0 = z80(bd.r[0]), z1 = z80(bd.r[1]), z2 = z80(bd.r[2]), z3 = z80(bd.r[3]);
    const uint32_t c0 = g2048_popc(z0), c1 = c0 + g2048_popc(z1), c2 = c1 + g2048_popc(z2),
                   n = c2 + g2048_popc(z3);
    const uint32_t k = g2048_mulhi(w, n);
    const bool g0 = k >= c0, g1 = k >= c1, g2 = k >= c2;
    const uint32_t zs = g2 ? z3 : (g1 ? z2 : (g0 ? z1 : z0));  // empty flags of the chosen row
    const uint32_t kk = k - (g2 ? c2 : (g1 ? c1 : (g0 ? c0 : 0u))); // rank inside the row
    const uint32_t p0 = g2048_popc(zs & 0x00000080u), p1 = g2048_popc(zs & 0x00008080u),
                   p2 = g2048_popc(zs & 0x00808080u);
    const uint32_t col = (p0 <= kk) + (p1 <= kk) + (p2 <= kk);
    const uint32_t exp = ((w & 0xffffu) <= 58982u) ? 1u : 2u;
    const uint32_t tile = exp << (8u * col);
    bd.r[0] |= g0 ? 0u : tile;
    bd.r[1] |= (g0 && !g1) ? tile : 0u;
    bd.r[2] |= (g1 && !g2) ? tile : 0u;
    bd.r[3] |= g2 ? tile : 0u;
}

// game2048_env.py:102-111: empty board + two spawns from words w1, w2.
G2048_DEV Board fresh_board(uint32_t w1, uint32_t w2)
{
    const uint32_t p1 = w1 >> 28;                 // (w1 * 16) >> 32
    const uint32_t k2 = g2048_mulhi(w2, 15u);
    const uint32_t p2 = k2 + (k2 >= p1 ? 1u : 0u); // k2-th empty cell, skipping p1
    const uint32_t e1 = ((w1 & 0xffffu) <= 58982u) ? 1u : 2u;
    const uint32_t e2 = ((w2 & 0xffffu) <= 58982u) ? 1u : 2u;
    const uint32_t t1 = e1 << (8u * (p1 & 3u)), t2 = e2 << (8u * (p2 & 3u));
    const uint32_t q1 = p1 >> 2, q2 = p2 >> 2;
    Board bd;
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i)
        bd.r[i] = (q1 == i ? t1 : 0u) | (q2 == i ? t2 : 0u);
    return bd;
}

// True when two neighbouring cells (horizontally or vertically) are equal.  Only meaningful on a
// FULL board, where it is exactly "some move is legal" (game2048_env.py:273-279).
G2048_DEV bool has_equal_neighbours(const Board &bd)
{
    uint32_t f = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        f |= z80(bd.r[i] ^ (bd.r[i] >> 8)) & 0x00808080u;
#pragma unroll
    for (int i = 0; i < 3; ++i)
        f |= z80(bd.r[i] ^ bd.r[i + 1]);
    return f != 0;
}

// Bytewise max of two SWAR words.
G2048_DEV uint32_t max4(uint32_t x, uint32_t y)
{
    const uint32_t ge = mask7(((x | kHigh1) - y) & kHigh1); // 0x7f where x >= y
    return bfi(ge, x, y);
}

// game2048_env.py:190-192 as an exponent.
G2048_DEV uint32_t highest(const Board &bd)
{
    uint32_t m = max4(max4(bd.r[0], bd.r[1]), max4(bd.r[2], bd.r[3]));
    m = max4(m, m >> 16);
    m = max4(m, m >> 8);
    return m & 0xffu;
}

// game2048_env.py:262-280.  max_exp: log2(max_tile), 0 = None.
G2048_DEV bool is_end(const Board &bd, uint32_t max_exp)
{
    bool end = false;
    if (count_empty(bd) == 0)                             // :270-271
        end = !has_equal_neighbours(bd);                  // :273-280
    if (max_exp != 0 && highest(bd) == max_exp)           // :267-268
        end = true;
    return end;
}

// ------------------------------------------------------------------------------ one env step
struct StepResult {
    float reward;
    bool terminated;
    bool illegal;
    Board terminal;         // board the episode ended on (valid when terminated)
    int32_t terminal_score; // its final merge score
};

// Word `s` (0..3) of a Philox block without a runtime-indexed array.
G2048_DEV uint32_t select_word(const Words &w, uint32_t s)
{
    return s == 0u ? w.w[0] : (s == 1u ? w.w[1] : (s == 2u ? w.w[2] : w.w[3]));
}

// game2048_env.py:76-100 on one board, followed -- when auto_reset -- by the caller's
// `if terminated: env.reset()` (game2048_env.py:102-111).  w = Philox block of this transaction:
// word 0 = the step's spawn; the reset uses words 1,2 after a legal move and 0,1 after an illegal
// one (an illegal move consumes no randomness, game2048_env.py:91-95).
G2048_DEV StepResult step_env(Board &bd, int32_t &score, uint32_t action, const Words &w, float illegal_reward,
                              uint32_t max_exp, bool auto_reset)
{
    StepResult r;
    Board nb = bd;
    uint32_t gain;
    const bool legal = move(nb, action, gain);            // :85
    // add_tile needs an empty cell; a board that changed always has one (a full board can only
    // change by merging).  When the move was illegal the result is dropped.
    add_tile(nb, w.w[0]);                                 // :88
    const bool end = is_end(nb, max_exp);                 // :89
    r.illegal = !legal;                                   // :91-95
    r.terminated = legal ? end : true;
    r.reward = legal ? (float)gain : illegal_reward;      // :90 / :95
    if (legal) {
        bd = nb;
        score += (int32_t)gain;                           // :86
    }
    r.terminal = bd;
    r.terminal_score = score;
    if (r.terminated && auto_reset) {
        bd = fresh_board(legal ? w.w[1] : w.w[0], legal ? w.w[2] : w.w[1]); // :104,:108-109
        score = 0;                                        // :105
    }
    return r;
}

} // namespace g2048
