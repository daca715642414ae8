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
// for "down"); for horizontal moves they are the columns.  Two formulations of that re-expression:
// move() -- an 8 x v_perm_b32 byte transpose plus lane-mask selects -- and move_sel(), the one the
// step kernels use: a two-stage v_perm network whose byte selectors come from a 4-row table, so the
// direction costs no select instructions at all.  No cross-lane traffic: the step lives in ~42 VGPRs.
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
static inline uint32_t g2048_clz(uint32_t x) { return (uint32_t)__builtin_clz(x); }   // x != 0
static inline uint32_t g2048_funnel_shr(uint32_t hi, uint32_t lo, uint32_t k) // low word of (hi:lo) >> k, 0 <= k <= 31
{
    return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (k & 31u));
}
static inline uint32_t g2048_ctz(uint32_t x) { return (uint32_t)__builtin_ctz(x); }   // x != 0
static inline uint32_t g2048_opaque(uint32_t x) { return x; }
static inline uint32_t g2048_bfi(uint32_t m, uint32_t a, uint32_t b) { return (m & a) | (~m & b); }
static inline bool g2048_any(bool x) { return x; }
static inline uint32_t g2048_xor3(uint32_t a, uint32_t b, uint32_t c) { return a ^ b ^ c; }
template <int K> static inline uint32_t g2048_pow2_byte(uint32_t x, uint32_t) { return 1u << ((x >> (8 * K)) & 31u); }
#else
#include <hip/hip_runtime.h>
#define G2048_DEV __device__ __forceinline__
G2048_DEV uint32_t g2048_perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
G2048_DEV uint32_t g2048_mulhi(uint32_t a, uint32_t b) { return __umulhi(a, b); }
G2048_DEV uint32_t g2048_popc(uint32_t x) { return (uint32_t)__popc(x); }
G2048_DEV uint32_t g2048_clz(uint32_t x) { return (uint32_t)__builtin_clz(x); }   // x != 0: v_ffbh_u32
// low word of (hi:lo) >> k, 0 <= k <= 31: v_alignbit_b32
G2048_DEV uint32_t g2048_funnel_shr(uint32_t hi, uint32_t lo, uint32_t k) { return __builtin_amdgcn_alignbit(hi, lo, k); }
G2048_DEV uint32_t g2048_ctz(uint32_t x) { return (uint32_t)__builtin_ctz(x); }   // x != 0: v_ffbl_b32
// Hides a value's origin from the optimizer.  Used on lane-wide select masks: without it LLVM turns
// "(m & a) | (~m & b)" with m = -(cond) back into v_cndmask_b32_e64 (4 issue cycles) instead of one
// v_bitop3_b32 (2 cycles).
G2048_DEV uint32_t g2048_opaque(uint32_t x)
{
    asm volatile("" : "+v"(x));
    return x;
}
// (m & a) | (~m & b) as ONE v_bitop3_b32 (2 issue cycles; v_bfi_b32 / v_cndmask_e64 take 4).
G2048_DEV uint32_t g2048_bfi(uint32_t m, uint32_t a, uint32_t b) { return __builtin_amdgcn_bitop3_b32(m, a, b, 0xCA); }
// a ^ b ^ c as ONE v_bitop3_b32 (the compiler emits two v_xor_b32 for the Philox rounds otherwise).
G2048_DEV uint32_t g2048_xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
// 1 << (byte K of x, low five bits) as ONE instruction: SDWA selects the byte as the shift operand of
// v_lshlrev_b32 (the compiler emits a v_lshrrev + v_lshl_add pair per byte otherwise).  `one` = a VGPR holding 1.
template <int K> G2048_DEV uint32_t g2048_pow2_byte(uint32_t x, uint32_t one)
{
    uint32_t r;
    if constexpr (K == 0)
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r) : "v"(x), "v"(one));
    else if constexpr (K == 1)
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(x), "v"(one));
    else if constexpr (K == 2)
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r) : "v"(x), "v"(one));
    else
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(r) : "v"(x), "v"(one));
    return r;
}
// true when the predicate holds in any active lane of the wavefront (wave-uniform).  The builtin takes the
// predicate as a lane mask; HIP's __ballot(int) would first materialise it as 0/1 in a VGPR and compare again.
G2048_DEV bool g2048_any(bool x) { return __builtin_amdgcn_ballot_w64(x) != 0ull; }
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
        // At every call site c0, c1, c3 and the key are wave-uniform (t, slot, seed) and only c2 (the board) is
        // per lane.  Rounds 0..2 therefore pair the two uniform terms of each xor3 with plain xors, so they --
        // and round 1's whole M1 * c2 product -- stay on the scalar unit (one v_xor with an SGPR operand per
        // word); the bitop3 builtin is VALU-only and would cost a v_mov per second SGPR operand.
        uint32_t n0, n2;
        if (round == 0) {
            n0 = (uint32_t)(p1 >> 32) ^ (c1 ^ k0);
            n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        } else if (round == 1) {
            n0 = c1 ^ ((uint32_t)(p1 >> 32) ^ k0);
            n2 = (uint32_t)(p0 >> 32) ^ (c3 ^ k1);
        } else if (round == 2) {
            n0 = (uint32_t)(p1 >> 32) ^ (c1 ^ k0);            // c1 = low word of round 1's uniform product
            n2 = g2048_xor3((uint32_t)(p0 >> 32), c3, k1);
        } else {
            n0 = g2048_xor3((uint32_t)(p1 >> 32), c1, k0);
            n2 = g2048_xor3((uint32_t)(p0 >> 32), c3, k1);
        }
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
G2048_DEV uint32_t bfi(uint32_t m, uint32_t a, uint32_t b) { return g2048_bfi(m, a, b); }

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

// 0x7f in every non-zero byte (a byte-select mask; all data bytes are < 0x80).
G2048_DEV uint32_t nzmask(uint32_t x) { return mask7(nz80(x)); }
// all-ones when cond, else 0 (lane-wide select mask; selects then cost one v_bitop3 each).
G2048_DEV uint32_t lanemask(bool cond) { return g2048_opaque(0u - (cond ? 1u : 0u)); }
// the same from bit 0 of an integer, without a compare
G2048_DEV uint32_t lanemask_bit0(uint32_t x) { return g2048_opaque(0u - (x & 1u)); }

// game2048_env.py:243-260 for four lines at once.  a,b,c,d: 1st..4th cell of each line.
// Returns the summed merge score of the four lines.
//
// Issue-cost notes (measured on gfx950, tools/ubench/valu_ubench.hip): VOP2 integer ops and
// v_bitop3_b32 issue in 2 cycles, v_perm / v_bcnt / v_cndmask_e64 / multiplies in 4.  Everything
// below is therefore phrased as three-input boolean functions of byte masks.
G2048_DEV uint32_t shift4(uint32_t &a, uint32_t &b, uint32_t &c, uint32_t &d)
{
    // -- compaction by rank (game2048_env.py:249-251 skips zeros): the j-th output is the j-th
    //    non-zero cell.  ka/kb/kc = "cell is non-zero" byte masks.
    const uint32_t ka = nzmask(a), kb = nzmask(b), kc = nzmask(c);
    const uint32_t one_ab = ka ^ kb;                                  // exactly one of a,b
    const uint32_t all_abc = ka & kb & kc;
    const uint32_t one_abc = (ka ^ kb ^ kc) & ~all_abc;               // exactly one of a,b,c
    const uint32_t two_abc = ((ka & kb) | (kc & (ka | kb))) & ~all_abc; // exactly two of a,b,c
    const uint32_t p0 = bfi(ka, a, bfi(kb, b, bfi(kc, c, d)));        // first non-zero
    const uint32_t p1 = (b & ka) | (c & one_ab) | (d & one_abc);      // second
    const uint32_t p2 = (c & ka & kb) | (d & two_abc);                // third
    const uint32_t p3 = d & all_abc;                                  // fourth
    // -- merge flags (0x80 per byte): a pair merges when equal and non-zero; leftmost first, each
    //    cell once (:252-255).  After compaction p[j+1] != 0 implies p[j] != 0.
    const uint32_t s1 = p1 + kLow7, s2 = p2 + kLow7, s3 = p3 + kLow7; // bit 7 = non-zero
    const uint32_t eab = ~((p0 ^ p1) + kLow7) & s1 & kHigh1;
    const uint32_t ebc = ~((p1 ^ p2) + kLow7) & s2 & kHigh1 & ~eab;
    const uint32_t ecd = ~((p2 ^ p3) + kLow7) & s3 & kHigh1 & ~ebc;
    const uint32_t iab = eab >> 7, ibc = ebc >> 7, icd = ecd >> 7;    // +1 on the exponent
    const uint32_t mab = eab - iab, mbc = ebc - ibc, mcd = ecd - icd; // 0x7f masks
    const uint32_t a1 = p0 + iab, b1 = p1 + ibc, c1 = p2 + icd;
    // -- outputs
    a = a1;
    b = bfi(mab, c1, b1);
    c = bfi(mab, p3 & ~mcd, bfi(mbc, p3, c1));
    d = p3 & ~(mab | mbc | mcd);
    // -- score: sum of 2^e over the merged cells (:253-254).  v_lshlrev uses only the low five bits
    //    of the shift operand, so "1 << (x >> 8l)" needs no byte extraction; bytes without a merge
    //    are set to 31, whose 2^31 terms can only disturb bit 31, which the final mask drops.
    const uint32_t m1 = bfi(mab, a1, bfi(mbc, b1, 0x1f1f1f1fu)); // first merge of each line
    const uint32_t m2 = bfi(mcd, c1, 0x1f1f1f1fu);               // second merge of each line
    const uint32_t one = g2048_opaque(1u);
    const uint32_t score = (g2048_pow2_byte<0>(m1, one) + g2048_pow2_byte<1>(m1, one) + g2048_pow2_byte<2>(m1, one)) +
                           (g2048_pow2_byte<3>(m1, one) + g2048_pow2_byte<0>(m2, one) + g2048_pow2_byte<1>(m2, one)) +
                           (g2048_pow2_byte<2>(m2, one) + g2048_pow2_byte<3>(m2, one));
    return score & 0x7fffffffu;
}

// game2048_env.py:194-241.  Returns true when the board changed (false = IllegalMove).
G2048_DEV bool move(Board &bd, uint32_t action, uint32_t &score)
{
    const uint32_t hz = lanemask_bit0(action);                 // :211 dir_mod_two
    const uint32_t rv = lanemask_bit0(action ^ (action >> 1)); // :212 shift_direction
    const Board t = transpose(bd);
    const uint32_t l0 = bfi(hz, t.r[0], bd.r[0]), l1 = bfi(hz, t.r[1], bd.r[1]);
    const uint32_t l2 = bfi(hz, t.r[2], bd.r[2]), l3 = bfi(hz, t.r[3], bd.r[3]);
    uint32_t a = bfi(rv, l3, l0), b = bfi(rv, l2, l1), c = bfi(rv, l1, l2), d = bfi(rv, l0, l3);
    const uint32_t a0 = a, b0 = b, c0 = c, d0 = d;
    score = shift4(a, b, c, d);
    const bool changed = ((a ^ a0) | (b ^ b0) | (c ^ c0) | (d ^ d0)) != 0; // :222,234,238
    Board o;
    o.r[0] = bfi(rv, d, a);
    o.r[1] = bfi(rv, c, b);
    o.r[2] = bfi(rv, b, c);
    o.r[3] = bfi(rv, a, d);
    const Board ot = transpose(o);
    bd.r[0] = bfi(hz, ot.r[0], o.r[0]);
    bd.r[1] = bfi(hz, ot.r[1], o.r[1]);
    bd.r[2] = bfi(hz, ot.r[2], o.r[2]);
    bd.r[3] = bfi(hz, ot.r[3], o.r[3]);
    return changed;
}

// Number of empty cells.
G2048_DEV uint32_t count_empty(const Board &bd)
{
    return g2048_popc(z80(bd.r[0]) | (z80(bd.r[1]) >> 1) | (z80(bd.r[2]) >> 2) | (z80(bd.r[3]) >> 3));
}

// The spawn word w on a board with n empty cells (game2048_env.py:166-176), read off the 64-bit product p = w * n:
//   position  k = p >> 32: the k-th empty cell in row-major order (= floor(u * n), u = w / 2^32);
//   value     2 (exponent 1) if (uint32_t)p <= kTwoThreshold else 4, i.e. frac(u * n) < 0.9 (:168) -- the part of the
//             word the position did not use: uniform and independent of k up to the word's resolution, so P(2) is
//             within n / 2^32 < 4e-9 of the reference's 0.9.
constexpr uint32_t kTwoThreshold = 3865470566u; // r / 2^32 < 0.9  <=>  r <= 3865470566

// `enable` (all-ones / 0) gates the write.  Returns the number of empty cells BEFORE the spawn; `is2` = "the new
// tile is a 2" (the callers that keep the score deficit need it as well).
G2048_DEV uint32_t add_tile(Board &bd, uint32_t w, uint32_t enable, bool &is2)
{
    const uint32_t z0 = z80(bd.r[0]), z1 = z80(bd.r[1]), z2 = z80(bd.r[2]), z3 = z80(bd.r[3]);
    const uint32_t c0 = g2048_popc(z0), c1 = c0 + g2048_popc(z1), c2 = c1 + g2048_popc(z2),
                   n = c2 + g2048_popc(z3);
    const uint64_t p = static_cast<uint64_t>(w) * n;
    const uint32_t k = static_cast<uint32_t>(p >> 32);
    is2 = static_cast<uint32_t>(p) <= kTwoThreshold;
    // row of the k-th empty cell: g_i = all-ones when k >= c_i (nested: g2 implies g1 implies g0)
    const uint32_t g0 = (uint32_t)((int32_t)(c0 - 1u - k) >> 31), g1 = (uint32_t)((int32_t)(c1 - 1u - k) >> 31),
                   g2 = (uint32_t)((int32_t)(c2 - 1u - k) >> 31);
    const uint32_t zs = bfi(g2, z3, bfi(g1, z2, bfi(g0, z1, z0)));     // empty flags of that row
    const uint32_t base = bfi(g2, c2, bfi(g1, c1, c0 & g0));           // empties before that row
    // inside the row: inclusive prefix count of the empty flags per byte; the target byte is the
    // empty one whose count equals kk + 1
    const uint32_t want = (k - base + 1u) * 0x01010101u;
    const uint32_t prefix = (zs >> 7) * 0x01010101u;
    const uint32_t hit = ~((prefix ^ want) + kLow7) & zs;              // 0x80 at the chosen cell only
    // exponent 1 -> 0x01 at that byte (hit >> 7), exponent 2 -> 0x02 (hit >> 6)
    const uint32_t tile = (hit >> (is2 ? 7u : 6u)) & enable;
    // one-hot row masks from the nested g's (kept as values: one v_bitop3 per row instead of compare + select)
    const uint32_t h0 = g2048_opaque(g0), h1 = g2048_opaque(g1), h2 = g2048_opaque(g2);
    bd.r[0] |= tile & ~h0;
    bd.r[1] |= tile & (h0 ^ h1);
    bd.r[2] |= tile & (h1 ^ h2);
    bd.r[3] |= tile & h2;
    return n;
}

G2048_DEV uint32_t add_tile(Board &bd, uint32_t w, uint32_t enable = 0xffffffffu)
{
    bool is2;
    return add_tile(bd, w, enable, is2);
}

// The two spawns of a reset (game2048_env.py:108-109) on the empty board: the first word meets 16 empty cells
// (k = w1 >> 28, fraction = w1 << 4), the second 15.
G2048_DEV bool fresh_is_four_16(uint32_t w1) { return (w1 << 4) > kTwoThreshold; }
G2048_DEV bool fresh_is_four_15(uint32_t w2) { return w2 * 15u > kTwoThreshold; }

// game2048_env.py:102-111: empty board + two spawns from words w1, w2.
G2048_DEV Board fresh_board(uint32_t w1, uint32_t w2)
{
    const uint32_t p1 = w1 >> 28;                  // (w1 * 16) >> 32: cell of the first tile
    const uint32_t k2 = g2048_mulhi(w2, 15u);
    const uint32_t p2 = k2 + (k2 >= p1 ? 1u : 0u); // k2-th empty cell, skipping p1
    const uint32_t e1 = fresh_is_four_16(w1) ? 2u : 1u;
    const uint32_t e2 = fresh_is_four_15(w2) ? 2u : 1u;
    // place each exponent in a 64-bit half (cells 0-7 / 8-15) with one 64-bit shift
    const uint64_t x1 = (uint64_t)e1 << (8u * (p1 & 7u)), x2 = (uint64_t)e2 << (8u * (p2 & 7u));
    const uint32_t h1 = lanemask_bit0(p1 >> 3), h2 = lanemask_bit0(p2 >> 3); // upper half?
    const uint32_t x1l = (uint32_t)x1, x1h = (uint32_t)(x1 >> 32), x2l = (uint32_t)x2, x2h = (uint32_t)(x2 >> 32);
    Board bd;
    bd.r[0] = (x1l & ~h1) | (x2l & ~h2);
    bd.r[1] = (x1h & ~h1) | (x2h & ~h2);
    bd.r[2] = (x1l & h1) | (x2l & h2);
    bd.r[3] = (x1h & h1) | (x2h & h2);
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

// ------------------------------------------------------------ direction by per-lane selectors
// The same move() without the select network: v_perm_b32 takes its byte selector from a VGPR, so
// a two-stage perm network whose SELECTORS depend on the lane's action maps the board straight to
// the shift-order registers A..D (reversal included) and back -- 16 v_perm, no v_bitop3 selects.
//   stage 1: x0 = perm(r1, r0, sa)  x1 = perm(r1, r0, sb)  x2 = perm(r3, r2, sa)  x3 = perm(r3, r2, sb)
//   stage 2: A  = perm(x2, x0, ta)  C  = perm(x2, x0, tc)  B  = perm(x3, x1, ta)  D  = perm(x3, x1, tc)
// Stage 2 is an involution (the way back uses ta, tc again); stage 1 is undone with va, vb.
// The table (one 32-byte row per action, game2048_env.py:196: 0 up, 1 right, 2 down, 3 left) is
// derived and checked by tests/test_device_math_host.py::test_move_lut.  The kernels keep it in LDS
// and fetch a lane's row with two ds_read (no VALU); the host check indexes the array directly.
struct MoveSel {
    uint32_t sa, sb, ta, tc, va, vb;
};

#define G2048_MOVE_LUT_WORDS                                                                        \
    0x03020100u, 0x07060504u, 0x03020100u, 0x07060504u, 0x03020100u, 0x07060504u, 0u, 0u, /* up */     \
    0x05010703u, 0x04000602u, 0x05040100u, 0x07060302u, 0x00040206u, 0x01050307u, 0u, 0u, /* right */  \
    0x07060504u, 0x03020100u, 0x07060504u, 0x03020100u, 0x07060504u, 0x03020100u, 0u, 0u, /* down */   \
    0x06020400u, 0x07030501u, 0x05040100u, 0x07060302u, 0x06020400u, 0x07030501u, 0u, 0u  /* left */

// game2048_env.py:194-241 with the lane's selector row.  Returns true when the board changed.
G2048_DEV bool move_sel(Board &bd, const MoveSel &s, uint32_t &score)
{
    const uint32_t x0 = g2048_perm(bd.r[1], bd.r[0], s.sa), x1 = g2048_perm(bd.r[1], bd.r[0], s.sb);
    const uint32_t x2 = g2048_perm(bd.r[3], bd.r[2], s.sa), x3 = g2048_perm(bd.r[3], bd.r[2], s.sb);
    uint32_t a = g2048_perm(x2, x0, s.ta), c = g2048_perm(x2, x0, s.tc);
    uint32_t b = g2048_perm(x3, x1, s.ta), d = g2048_perm(x3, x1, s.tc);
    const uint32_t a0 = a, b0 = b, c0 = c, d0 = d;
    score = shift4(a, b, c, d);
    const bool changed = ((a ^ a0) | (b ^ b0) | (c ^ c0) | (d ^ d0)) != 0; // :222,234,238
    const uint32_t y0 = g2048_perm(c, a, s.ta), y2 = g2048_perm(c, a, s.tc);
    const uint32_t y1 = g2048_perm(d, b, s.ta), y3 = g2048_perm(d, b, s.tc);
    bd.r[0] = g2048_perm(y1, y0, s.va);
    bd.r[1] = g2048_perm(y1, y0, s.vb);
    bd.r[2] = g2048_perm(y3, y2, s.va);
    bd.r[3] = g2048_perm(y3, y2, s.vb);
    return changed;
}

// ------------------------------------------------------------------- the 16-byte board RECORD
// What the engine keeps per board in HBM is ONE 16-byte record: bits [4:0] of byte j = exponent of
// cell j (0..31), and the 24-bit SCORE DEFICIT d in the three spare bits [7:5] of bytes 8..15
// (registers r[2], r[3]; bit k of d lives in bit 5 + k % 3 of byte 8 + k / 3).
//   d = (potential(board) - score) mod 2^24,   potential = sum over tiles of (e - 1) * 2^e.
// Why a deficit and not the score: a merge of two 2^e tiles raises the potential by exactly the 2^(e+1)
// it scores (game2048_env.py:253-254), a spawned 2 leaves it alone and a spawned 4 raises it by 4
// without scoring -- so d changes only when a 4 is spawned (d += 4), which in the spread layout is one
// carry-propagating add; the episodic score self.score (game2048_env.py:46,86,105) is recovered as
// potential - d where it is needed (episode end, get_scores).  No separate score array: a step reads
// and writes exactly the 16-byte record.  Scores are exact below 2^24 = 16 777 216 (the largest score
// a 4x4 game can reach is about 3.9 million).
constexpr uint32_t kCellBits = 0x1f1f1f1fu, kSpareBits = 0xe0e0e0e0u, kScoreMask = 0x00ffffffu;

G2048_DEV Board record_cells(const Board &raw)
{
    return Board{{raw.r[0], raw.r[1], raw.r[2] & kCellBits, raw.r[3] & kCellBits}};
}

// sum over the 16 cells of (e - 1) * 2^e (0 for an empty cell); cells must be masked (e <= 31)
G2048_DEV uint32_t potential(const Board &cells)
{
    uint32_t hi = 0, lo = 0; // sum e << e, sum 1 << e
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const uint32_t e = (cells.r[i] >> (8 * l)) & 0xffu;
            hi += e << (e & 31u);
            lo += 1u << (e & 31u);
        }
    }
    return hi - lo + count_empty(cells); // an empty cell contributed 0 - 1
}

// 12 spare bits of one register -> 12 contiguous bits
G2048_DEV uint32_t gather12(uint32_t r)
{
    uint32_t y = (r >> 5) & 0x07070707u;
    y = (y | (y >> 5)) & 0x003f003fu;
    return (y | (y >> 10)) & 0xfffu;
}

// 12 contiguous bits -> the spare bits of one register
G2048_DEV uint32_t scatter12(uint32_t v)
{
    uint32_t y = (v | (v << 10)) & 0x003f003fu;
    y = (y | (y << 5)) & 0x07070707u;
    return y << 5;
}

G2048_DEV uint32_t record_deficit(const Board &raw) { return gather12(raw.r[2]) | (gather12(raw.r[3]) << 12); }

G2048_DEV uint32_t record_score(const Board &raw)
{
    return (potential(record_cells(raw)) - record_deficit(raw)) & kScoreMask;
}

G2048_DEV Board make_record(const Board &cells, uint32_t score)
{
    const uint32_t d = (potential(cells) - score) & kScoreMask;
    return Board{{cells.r[0], cells.r[1], cells.r[2] | scatter12(d & 0xfffu), cells.r[3] | scatter12(d >> 12)}};
}

// d += inc (inc = 0 or 4, given as the spread value 0 / 0x80 for bit 2 of d) without leaving the
// spread layout: the cell bits between the deficit bits are filled with ones so that the carry runs
// through them, then the new cells are put back.
G2048_DEV void record_update(Board &raw, const Board &cells, uint32_t inc_spread)
{
    const uint64_t filled = (static_cast<uint64_t>(raw.r[3] | kCellBits) << 32) | (raw.r[2] | kCellBits);
    const uint64_t sum = filled + inc_spread;
    raw.r[0] = cells.r[0];
    raw.r[1] = cells.r[1];
    raw.r[2] = bfi(kSpareBits, static_cast<uint32_t>(sum), cells.r[2]);
    raw.r[3] = bfi(kSpareBits, static_cast<uint32_t>(sum >> 32), cells.r[3]);
}

// game2048_env.py:102-111 as a record: fresh board, score 0, so d = potential = 4 per spawned 4.
G2048_DEV Board fresh_record(uint32_t w1, uint32_t w2)
{
    Board bd = fresh_board(w1, w2);
    const uint32_t fours = (fresh_is_four_16(w1) ? 1u : 0u) + (fresh_is_four_15(w2) ? 1u : 0u);
    // d = 4 -> bit 2 of d = bit 7 of byte 8; d = 8 -> bit 3 of d = bit 5 of byte 9
    bd.r[2] |= fours == 1u ? 0x80u : (fours == 2u ? 0x2000u : 0u);
    return bd;
}

// The same through a 16-entry table of one-tile boards (entry p = a board whose only tile is a 2,
// exponent 1, in cell p): two table reads, shifts and ORs instead of ~40 VALU of placement logic.
// `Tables` supplies onehot_cell(p) (LDS on the device, an array in the host check).
template <class Tables>
G2048_DEV Board fresh_record_lut(uint32_t w1, uint32_t w2, const Tables &tb)
{
    const uint32_t p1 = w1 >> 28;                  // (w1 * 16) >> 32: cell of the first tile
    const uint32_t k2 = g2048_mulhi(w2, 15u);
    const uint32_t p2 = k2 + (k2 >= p1 ? 1u : 0u); // k2-th empty cell, skipping p1
    const uint32_t s1 = fresh_is_four_16(w1) ? 1u : 0u; // 1: the tile is a 4 (exponent 2 = 1 << 1)
    const uint32_t s2 = fresh_is_four_15(w2) ? 1u : 0u;
    const Board a = tb.onehot_cell(p1), b = tb.onehot_cell(p2);
    const uint32_t fours = s1 + s2;
    Board bd;
    bd.r[0] = (a.r[0] << s1) | (b.r[0] << s2);
    bd.r[1] = (a.r[1] << s1) | (b.r[1] << s2);
    bd.r[2] = (a.r[2] << s1) | (b.r[2] << s2) | ((fours & 1u) << 7) | ((fours >> 1) << 13); // d = 4 * fours
    bd.r[3] = (a.r[3] << s1) | (b.r[3] << s2);
    return bd;
}

// word j (0..3) of entry p of the one-tile table: 1 in byte p & 3 of register p >> 2
G2048_DEV uint32_t onehot_cell_word(uint32_t p, uint32_t j) { return (p >> 2) == j ? 1u << (8u * (p & 3u)) : 0u; }

// --------------------------------------------------------------------- stack() four cells at a time
// game2048_env.py:17-32: channel c of the observation = (cell == 2^c), i.e. (exponent == c).  All of one 32-bit word's
// four cells against one channel: t = cells ^ splat(c) has every byte < 0x40, so 0x80808080 - t has bit 7 set exactly
// in the bytes where t == 0 and no borrow crosses a byte.  An exponent >= 16 matches no channel (all-zero column).
G2048_DEV uint32_t eq_flags(uint32_t cells, uint32_t splat) { return (kHigh1 - (cells ^ splat)) & kHigh1; }             // 0x80 per match
G2048_DEV uint32_t eq_ones(uint32_t cells, uint32_t splat) { return ((kHigh1 - (cells ^ splat)) >> 7) & 0x01010101u; } // uint8 1 per match

// the same four cells as fp16 (two words: cells 0,1 and 2,3; 1.0 = 0x3c00) and fp32 (four words; 1.0 = 0x3f800000):
// the matching bytes become 0x3c / 0x3f, and v_perm puts each one in the top byte of its half / word
struct OneHot4F16 { uint32_t lo, hi; };
struct OneHot4F32 { uint32_t w[4]; };

G2048_DEV OneHot4F16 onehot4_f16(uint32_t cells, uint32_t splat)
{
    const uint32_t f = eq_flags(cells, splat);
    const uint32_t g = (f - (f >> 7)) & 0x3c3c3c3cu;
    return OneHot4F16{g2048_perm(g, g, 0x010c000cu), g2048_perm(g, g, 0x030c020cu)};
}

G2048_DEV OneHot4F32 onehot4_f32(uint32_t cells, uint32_t splat)
{
    const uint32_t f = eq_flags(cells, splat);            // 0x80 in the matching bytes: byte 2 of 1.0f
    const uint32_t g = (f - (f >> 7)) & 0x3f3f3f3fu;      // 0x3f there: byte 3 of 1.0f
    return OneHot4F32{{g2048_perm(g, f, 0x04000c0cu), g2048_perm(g, f, 0x05010c0cu), g2048_perm(g, f, 0x06020c0cu),
                       g2048_perm(g, f, 0x07030c0cu)}};
}

// One 16-byte CHUNK of a wavefront's observation piece.  The 64 boards of a wavefront are consecutive, so their
// observations are one contiguous piece of the output; the wave parks its 64 records (cells masked) in `recs` and
// writes the piece with 16-byte stores, chunk s * 64 + lane in store s (g2048_kernels.hip emit_onehot).  Per dtype:
//   OBS 0  u8 : a chunk = one channel (16 cells) of one board       16 chunks / board, boards 4s .. 4s+3 in store s
//   OBS 1  f16: a chunk = half a channel (8 cells = rows 2h, 2h+1)  32 chunks / board, boards 2s, 2s+1
//   OBS 2  f32: a chunk = one row of one channel (4 cells)          64 chunks / board, board s
// `board` = index (0..63) of the board the chunk belongs to.
struct alignas(16) Cells16 { uint32_t r[4]; };
struct alignas(16) Chunk16 { uint32_t w[4]; };

template <int OBS>
G2048_DEV Chunk16 onehot_chunk(const Cells16 *recs, uint32_t s, uint32_t lane, uint32_t &board)
{
    if constexpr (OBS == 0) {
        board = s * 4u + (lane >> 4);
        const uint32_t splat = (lane & 15u) * 0x01010101u;
        const Cells16 v = recs[board];
        return Chunk16{{eq_ones(v.r[0], splat), eq_ones(v.r[1], splat), eq_ones(v.r[2], splat), eq_ones(v.r[3], splat)}};
    } else if constexpr (OBS == 1) {
        board = s * 2u + (lane >> 5);
        const uint32_t splat = ((lane >> 1) & 15u) * 0x01010101u, half = lane & 1u;
        const OneHot4F16 a = onehot4_f16(recs[board].r[half * 2u], splat), b = onehot4_f16(recs[board].r[half * 2u + 1u], splat);
        return Chunk16{{a.lo, a.hi, b.lo, b.hi}};
    } else {
        board = s;
        const OneHot4F32 h = onehot4_f32(recs[board].r[lane & 3u], (lane >> 2) * 0x01010101u);
        return Chunk16{{h.w[0], h.w[1], h.w[2], h.w[3]}};
    }
}

// ---------------------------------------------------------------------- one env step on a record
struct StepOut {
    uint32_t gain;   // merge score of the move (:85); 0 when illegal
    bool legal;      // false = the reference's IllegalMove (:91)
    bool terminated; // :89 / :94
    uint32_t legal_mask; // all-ones when legal (lane-wide select mask, reused by reset_record)
    Board terminal;  // step_record only: record after move + spawn, before any auto-reset ("terminal" when terminated)
};

// play_record: game2048_env.py:76-100 on one board RECORD (no reset: rec is the terminal record when
// o.terminated).  w = Philox block of this transaction: word 0 = the step's
// spawn; the reset uses words 1,2 after a legal move and 0,1 after an illegal one (an illegal move
// consumes no randomness, :91-95).  The score never appears: a merge moves potential and score together,
// only a spawned 4 touches the deficit.
template <class Tables>
G2048_DEV StepOut play_record(Board &rec, uint32_t action, const Words &w, uint32_t max_exp, const Tables &tb)
{
    StepOut o;
    Board cells = record_cells(rec);
    o.legal = move_sel(cells, tb.move_sel(action), o.gain);        // :85 (illegal: board unchanged, gain 0)
    const uint32_t lm = lanemask(o.legal);
    o.legal_mask = lm;
    // :88 add_tile needs an empty cell; a board that changed always has one (a full board can only
    // change by merging).  After an illegal move nothing is spawned (:91-95).
    bool is2;                                                      // :168 the spawn's value
    const uint32_t n_empty = add_tile(cells, w.w[0], lm, is2);
    // :89 isend(): the board is full after the spawn exactly when it had one empty cell before it
    bool end = false;
    if (n_empty == 1u)                                             // :270-271
        end = !has_equal_neighbours(cells);                        // :273-280
    if (max_exp != 0 && highest(cells) == max_exp)                 // :267-268
        end = true;
    o.terminated = o.legal ? end : true;                           // :89, :94
    // a spawned 4 raises the potential without scoring: deficit += 4 (bit 2 of d = bit 7 of byte 8)
    const uint32_t inc = (o.legal && !is2) ? 0x80u : 0u;
    record_update(rec, cells, inc);
    return o;
}

// The caller's `if terminated: env.reset()` (:102-111) for a lane whose episode just ended in play_record: a
// plain divergent branch at the call site -- only the lanes that reset execute it (exec-masked writes straight
// into the record, no selects), and the wavefront skips it when none does.  The kernels store the terminal
// record BEFORE calling this, so the fresh record overwrites it in place and no second copy is ever live
// (four v_mov per lane otherwise; -0.2 us per launch at 2^20 boards, profiles/r02_s_ubench_2p20.txt).
template <class Tables>
G2048_DEV void reset_record(Board &rec, const StepOut &o, const Words &w, const Tables &tb)
{
    rec = fresh_record_lut(bfi(o.legal_mask, w.w[1], w.w[0]), bfi(o.legal_mask, w.w[2], w.w[1]), tb); // :104-109
}

// play_record + reset_record in one call, keeping the terminal record in o.terminal (host mirror, tools).
template <class Tables>
G2048_DEV StepOut step_record(Board &rec, uint32_t action, const Words &w, uint32_t max_exp, bool auto_reset,
                              const Tables &tb)
{
    StepOut o = play_record(rec, action, w, max_exp, tb);
    o.terminal = rec;
    if (o.terminated && auto_reset)
        reset_record(rec, o, w, tb);
    return o;
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
    uint32_t gain;
    const bool legal = move(bd, action, gain);            // :85 (an illegal move leaves bd unchanged
                                                          //      and merges nothing: gain == 0)
    // :88 add_tile needs an empty cell; a board that changed always has one (a full board can only
    // change by merging).  After an illegal move nothing is spawned (:91-95).
    const uint32_t n_empty = add_tile(bd, w.w[0], lanemask(legal));
    // :89 isend(): the board is full after the spawn exactly when it had one empty cell before it
    bool end = false;
    if (n_empty == 1u)                                    // :270-271
        end = !has_equal_neighbours(bd);                  // :273-280
    if (max_exp != 0 && highest(bd) == max_exp)           // :267-268
        end = true;
    r.illegal = !legal;                                   // :91-95
    r.terminated = legal ? end : true;
    r.reward = legal ? (float)gain : illegal_reward;      // :90 / :95
    score += (int32_t)gain;                               // :86
    r.terminal = bd;
    r.terminal_score = score;
    const bool do_reset = r.terminated && auto_reset;
    if (g2048_any(do_reset)) {                            // wave-uniform: skipped when nobody finished
        const uint32_t lm = lanemask(legal), rm = lanemask(do_reset);
        const Board fb = fresh_board(bfi(lm, w.w[1], w.w[0]), bfi(lm, w.w[2], w.w[1])); // :104,:108-109
        bd.r[0] = bfi(rm, fb.r[0], bd.r[0]);
        bd.r[1] = bfi(rm, fb.r[1], bd.r[1]);
        bd.r[2] = bfi(rm, fb.r[2], bd.r[2]);
        bd.r[3] = bfi(rm, fb.r[3], bd.r[3]);
        score &= (int32_t)~rm;                            // :105
    }
    return r;
}

} // namespace g2048
