// g2048_pcg64.h -- numpy-compatible RNG mode: the reference's OWN randomness on the device.
//
// The reference draws from self.np_random (game2048_env.py:168,170), which gymnasium creates as
// numpy.random.Generator(PCG64(SeedSequence(seed))).  In this mode every board carries the state of
// its own PCG64 (seeded on the host by numpy itself, board i <- seed + i like SB3's make_vec_env) and
// the kernels consume it exactly as numpy would:
//   next64   : state = state * 0x2360ED051FC65DA44385DF649FCCF645 + inc (mod 2^128),
//              out = rotr64(hi ^ lo, state >> 122)                 [numpy random/src/pcg64/pcg64.h]
//   next32   : low half of a fresh next64, the high half is buffered for the following call
//   random() : (next64 >> 11) * 2^-53; "< 0.9" <=> (next64 >> 11) < 8106479329266893 (0.9 * 2^53 exactly)
//   shuffle  : for i = 15 .. 1: j = interval(i); swap(pos[i], pos[j])   [_generator.pyx, untyped path]
//   interval : masked rejection on next32                        [distributions.c random_interval]
// With it, board i plays bit-for-bit the game the unmodified reference env plays after
// reset(seed = s + i) (tests/golden/traj_numpy_*.npz).  It costs ~7x the spawn-stream mode (about
// ten 128-bit LCG steps per spawn, divergent rejection loops, 40 B/board of RNG state per step) and
// exists for fidelity, not for the benchmark.
#pragma once

#include "g2048_device.h"

#if defined(G2048_HOST_CHECK)
static inline uint64_t g2048_mulhi64(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }
#else
G2048_DEV uint64_t g2048_mulhi64(uint64_t a, uint64_t b) { return __umul64hi(a, b); }
#endif

namespace g2048 {

struct Pcg64 {
    uint64_t state_lo, state_hi, inc_lo, inc_hi;
    uint64_t buf; // bits 0..31 buffered high half ("uinteger"), bit 32 = has_uint32
};

constexpr uint64_t kPcgMultHi = 0x2360ED051FC65DA4ull, kPcgMultLo = 0x4385DF649FCCF645ull;
constexpr uint64_t kTwoThreshold53 = 8106479329266893ull; // 0.9 * 2^53 (0.9 as IEEE double)

// One step of the 128-bit LCG and its output, as the two 32-bit halves numpy hands out (low half first).
// state * MULT + inc (mod 2^128) by 32-bit limbs, schoolbook columns: the six products that reach below bit 96 are full
// 32 x 32 + 64 -> 64 multiply-adds (v_mad_u64_u32), the four that only reach limb 3 are low halves (v_mul_lo_u32), and
// every 64-bit addend is kept below 2^33 so that no multiply-add can overflow: (2^32 - 1)^2 + 2^33 - 2 < 2^64.  (Written
// as two 64 x 64 -> 128 products the compiler spends 8 multiply-adds + 6 multiplies and four 64-bit adds on it; the LCG
// step is two fifths of a spawn in this mode.)
G2048_DEV void pcg64_step2x32(Pcg64 &r, uint32_t &out_lo, uint32_t &out_hi)
{
    constexpr uint32_t m0 = (uint32_t)kPcgMultLo, m1 = (uint32_t)(kPcgMultLo >> 32);
    constexpr uint32_t m2 = (uint32_t)kPcgMultHi, m3 = (uint32_t)(kPcgMultHi >> 32);
    const uint32_t s0 = (uint32_t)r.state_lo, s1 = (uint32_t)(r.state_lo >> 32);
    const uint32_t s2 = (uint32_t)r.state_hi, s3 = (uint32_t)(r.state_hi >> 32);
    const uint32_t c0 = (uint32_t)r.inc_lo, c1 = (uint32_t)(r.inc_lo >> 32);
    const uint32_t c2 = (uint32_t)r.inc_hi, c3 = (uint32_t)(r.inc_hi >> 32);
    const uint64_t t0 = (uint64_t)s0 * m0 + c0;                                   // column 0
    const uint64_t u = (uint64_t)s0 * m1 + ((t0 >> 32) + c1);                     // column 1 (addend < 2^33)
    const uint64_t v = (uint64_t)s1 * m0 + (uint32_t)u;
    const uint64_t w1 = (uint64_t)s0 * m2 + ((u >> 32) + (v >> 32));              // column 2 (addend <= 2^33 - 2)
    const uint64_t w2 = (uint64_t)s1 * m1 + ((uint64_t)(uint32_t)w1 + c2);
    const uint64_t w3 = (uint64_t)s2 * m0 + (uint32_t)w2;
    const uint32_t n0 = (uint32_t)t0, n1 = (uint32_t)v, n2 = (uint32_t)w3;
    const uint32_t n3 = s0 * m3 + s1 * m2 + s2 * m1 + s3 * m0 + (uint32_t)(w1 >> 32) + (uint32_t)(w2 >> 32) +
                        (uint32_t)(w3 >> 32) + c3;                                // column 3 (mod 2^32)
    r.state_lo = n0 | ((uint64_t)n1 << 32);
    r.state_hi = n2 | ((uint64_t)n3 << 32);
    // out = rotr64(hi ^ lo, state >> 122): a rotation by >= 32 swaps the halves, the rest is two funnel shifts
    const uint32_t rot = n3 >> 26;
    const uint32_t xl = n0 ^ n2, xh = n1 ^ n3;
    const uint32_t a = (rot & 32u) ? xh : xl, b = (rot & 32u) ? xl : xh;          // rotate (b:a) right by rot & 31
    const uint32_t k = rot & 31u;
    out_lo = g2048_funnel_shr(b, a, k);
    out_hi = g2048_funnel_shr(a, b, k);
}

G2048_DEV uint64_t pcg64_next64(Pcg64 &r)
{
    uint32_t lo, hi;
    pcg64_step2x32(r, lo, hi);
    return lo | ((uint64_t)hi << 32);
}

G2048_DEV uint32_t pcg64_next32(Pcg64 &r)
{
    if (r.buf >> 32) {
        const uint32_t v = (uint32_t)r.buf;
        r.buf = 0;
        return v;
    }
    const uint64_t next = pcg64_next64(r);
    r.buf = (next >> 32) | (1ull << 32);
    return (uint32_t)next;
}

// random_interval(max) for 1 <= max <= 15.
G2048_DEV uint32_t pcg64_interval(Pcg64 &r, uint32_t max)
{
    const uint32_t mask = max >= 8u ? 15u : (max >= 4u ? 7u : (max >= 2u ? 3u : 1u));
    uint32_t v;
    do {
        v = pcg64_next32(r) & mask;
    } while (v > max);
    return v;
}

// 16-bit mask of the empty cells, bit p = cell p (row-major).
G2048_DEV uint32_t empty_mask16(const Board &bd)
{
    // z80 has bit 7 of every empty byte; (z * 0x00204081) >> 28 gathers bits 7,15,23,31 into a nibble
    return ((z80(bd.r[0]) * 0x00204081u) >> 28) | (((z80(bd.r[1]) * 0x00204081u) >> 28) << 4) |
           (((z80(bd.r[2]) * 0x00204081u) >> 28) << 8) | (((z80(bd.r[3]) * 0x00204081u) >> 28) << 12);
}

// game2048_env.py:166-176 with numpy's draws.  Precondition: at least one empty cell.
//
// :170 Generator.shuffle of the 16 positions is a Fisher-Yates pass "for i = 15 .. 1: j = random_interval(i);
// swap(pos[i], pos[j])", random_interval being masked rejection on 32-bit draws (a rejected draw, v > i, leaves i for the
// next draw), and numpy serves 32-bit draws as the two halves of one 64-bit output, low half first, the high half
// buffered.  The draws a lane consumes must be exactly numpy's (the generator goes on), but the board only needs ONE
// thing from the shuffled order: the first empty position (:171-175).  So the loop -- the part that runs in LOCKSTEP on the
// 128-bit LCG, one LCG step and two draws per trip -- only RECORDS the accepted j of every i (15 nibbles, pushed into a
// 64-bit register pair: two instructions), and the permutation is never materialised.  Afterwards, in straight-line code:
//   * where the EMPTY cells end up: the 16-bit mask of indices that hold an empty cell is pushed through the 15 swaps
//     (bit i becomes final at step i; seven instructions per step);
//   * the lowest such index k is the first empty position of the shuffled order; which cell sits there is found by
//     walking index k BACKWARDS through the swaps (i = 1 .. 15: if x == i: x = j_i, else if x == j_i: x = i).
// Round 5 swapped nibbles of a packed permutation inside the loop (eight 64-bit shifts per accepted draw): ~2 700 of the
// ~5 000 issue cycles of a spawn; this form: ~1 000 + ~700.
// Returns true when the new tile is a 4 (the callers that keep the score deficit need it); `where` = its cell.
G2048_DEV bool add_tile_numpy(Board &bd, Pcg64 &r, uint32_t &where)
{
    const uint32_t exp = ((pcg64_next64(r) >> 11) < kTwoThreshold53) ? 1u : 2u; // :168
    uint32_t i = 15;
    uint32_t j_lo = 0, j_hi = 0; // the accepted j's, pushed: after 15 of them nibble (i - 1) of j_hi:j_lo is j_i
    auto consume = [&](uint32_t draw) {
        const uint32_t j = draw & (0xffffffffu >> g2048_clz(i)); // mask = smallest 2^m - 1 >= i (random_interval)
        if (j <= i) {
            j_hi = (j_hi << 4) | (j_lo >> 28);
            j_lo = (j_lo << 4) | j;
            --i;
        }
    };
    if (r.buf >> 32) { // a half that was buffered before the call is used up first
        consume((uint32_t)r.buf);
        r.buf = 0;
    }
    while (i >= 1u) {
        uint32_t lo, hi;
        pcg64_step2x32(r, lo, hi);
        consume(lo);
        if (i >= 1u)
            consume(hi);
        else
            r.buf = hi | (1ull << 32); // a lane that finishes on a low half leaves the high half buffered
    }
    // indices holding an empty cell, pushed through the swaps: step i moves the element at j_i to index i (final from
    // then on) and the one at i to j_i
    uint32_t at = empty_mask16(bd), fin = 0; // pos starts as the identity: index p holds cell p
#pragma unroll
    for (uint32_t s = 15; s >= 1u; --s) {
        const uint32_t j = ((s > 8u ? j_hi : j_lo) >> (4u * ((s - 1u) & 7u))) & 15u;
        const uint32_t a = (at >> s) & 1u, b = (at >> j) & 1u;
        fin |= b << s;
        at = (at & ~(1u << j)) | (a << j);
    }
    fin |= at & 1u;
    uint32_t x = g2048_ctz(fin); // the first empty position of the shuffled order (:171-175) ...
#pragma unroll
    for (uint32_t s = 1; s <= 15u; ++s) { // ... and the cell that the shuffle put there
        const uint32_t j = ((s > 8u ? j_hi : j_lo) >> (4u * ((s - 1u) & 7u))) & 15u;
        x = x == s ? j : (x == j ? s : x);
    }
    const uint32_t p = x;
    const uint32_t tile = exp << (8u * (p & 3u));
    const uint32_t q = p >> 2;
    bd.r[0] |= q == 0u ? tile : 0u;
    bd.r[1] |= q == 1u ? tile : 0u;
    bd.r[2] |= q == 2u ? tile : 0u;
    bd.r[3] |= q == 3u ? tile : 0u;
    where = p;
    return exp == 2u;
}

G2048_DEV bool add_tile_numpy(Board &bd, Pcg64 &r)
{
    uint32_t where;
    return add_tile_numpy(bd, r, where);
}

// a board holding one tile (a 2, or a 4 when `four`) in cell p
G2048_DEV Board one_tile_board(uint32_t p, bool four)
{
    const uint32_t tile = (four ? 2u : 1u) << (8u * (p & 3u)), q = p >> 2;
    return Board{{q == 0u ? tile : 0u, q == 1u ? tile : 0u, q == 2u ? tile : 0u, q == 3u ? tile : 0u}};
}

// game2048_env.py:102-111 on a RECORD in numpy-RNG mode: empty board, score 0, two spawns; a spawned 4 raises the
// potential without scoring, so the fresh record's deficit is 4 per spawned 4 (d = 4: bit 7 of byte 8, d = 8: bit 5 of
// byte 9 -- as fresh_record() in the spawn-stream mode).
G2048_DEV Board fresh_record_numpy(Pcg64 &r)
{
    Board bd{{0u, 0u, 0u, 0u}};                                   // :104, :105 score = 0
    const uint32_t fours = (add_tile_numpy(bd, r) ? 1u : 0u) + (add_tile_numpy(bd, r) ? 1u : 0u); // :108, :109
    bd.r[2] |= fours == 1u ? 0x80u : (fours == 2u ? 0x2000u : 0u);
    return bd;
}

// The same reset when its FIRST spawn (:108) has already been drawn -- tile in cell `first_cell`, a 4 when `first_four` --
// for the lanes where `have_first`; the others draw both.  (The step kernel lets a lane whose move was illegal draw the
// first tile of its reset while the rest of its wavefront draws the step's spawn: an illegal move spawns nothing, :91-95,
// so that lane would idle, and the generator is consumed in the reference's order either way.)  Whole wavefronts call this.
G2048_DEV Board finish_record_numpy(Pcg64 &r, bool have_first, uint32_t first_cell, bool first_four)
{
    Board bd = one_tile_board(first_cell, first_four);
    uint32_t fours = first_four ? 1u : 0u;
    if (g2048_any(!have_first)) { // (wave-uniform: under a random policy nearly every episode ends on an illegal move)
        if (!have_first) {
            bd = Board{{0u, 0u, 0u, 0u}};
            fours = add_tile_numpy(bd, r) ? 1u : 0u;              // :108
        }
    }
    fours += add_tile_numpy(bd, r) ? 1u : 0u;                     // :109
    bd.r[2] |= fours == 1u ? 0x80u : (fours == 2u ? 0x2000u : 0u);
    return bd;
}

// game2048_env.py:76-100 on one board RECORD in numpy-RNG mode, WITHOUT the caller's reset (rec is the terminal record
// when the episode ended): the counterpart of play_record().  A merge moves potential and score together, so the
// score never appears; only a spawned 4 touches the deficit.
struct NumpyStepOut {
    uint32_t gain;   // :85 merge score of the move; 0 when illegal
    bool legal;      // false = IllegalMove (:91)
    bool terminated; // :89 / :94
    uint32_t top;    // exponent of the highest tile after the step (:97)
    bool have_first; // early_reset only: the move was illegal and the FIRST tile of the reset that follows (:108) is drawn:
    uint32_t first_cell; //   its cell
    bool first_four;     //   and whether it is a 4
};

// early_reset (the caller resets a board whose episode ends, :102-111): a lane whose move is illegal spawns nothing in its
// step (:91-95) and would sit out the wavefront's lockstep spawn -- it draws the first tile of its reset there instead
// (same generator, same order: nothing else draws in between).
G2048_DEV NumpyStepOut play_record_numpy(Board &rec, uint32_t action, Pcg64 &rng, uint32_t max_exp, bool early_reset)
{
    NumpyStepOut o;
    Board cells = record_cells(rec);
    o.legal = move(cells, action, o.gain);                // :85 (illegal: board unchanged, gain 0)
    o.have_first = !o.legal && early_reset;
    o.first_cell = 0;
    bool four = false, end = false;
    if (o.legal || o.have_first) {
        Board target = cells;
        if (!o.legal)
            target = Board{{0u, 0u, 0u, 0u}};             // :104
        four = add_tile_numpy(target, rng, o.first_cell); // :88 / :108
        if (o.legal) {
            cells = target;
            end = is_end(cells, max_exp);                 // :89
        }
    }
    o.first_four = four;
    o.terminated = o.legal ? end : true;                  // :89, :94
    o.top = highest(cells);
    record_update(rec, cells, (o.legal && four) ? 0x80u : 0u); // deficit += 4 for a spawned 4
    return o;
}

} // namespace g2048

// ------------------------------------------------------------------------------- device seeding
// numpy.random.SeedSequence(entropy).generate_state(4, uint64) followed by PCG64's seeding, for an
// integer entropy < 2^64 and an empty spawn key [numpy random/bit_generator.pyx: SeedSequence
// mix_entropy / generate_state; _pcg64.pyx _seed -> pcg64_set_seed -> pcg_setseq_128_srandom_r].
// Lets an engine seed a million boards (seed + i) on the device instead of looping over numpy on the
// host.  Pinned against numpy by tests/test_numpy_rng.py.
namespace g2048 {

G2048_DEV uint32_t ss_hashmix(uint32_t value, uint32_t &hash_const)
{
    value ^= hash_const;
    hash_const *= 0x931e8875u; // MULT_A
    value *= hash_const;
    value ^= value >> 16;
    return value;
}

G2048_DEV uint32_t ss_mix(uint32_t x, uint32_t y)
{
    uint32_t r = 0xca01f9ddu * x - 0x4973f715u * y; // MIX_MULT_L, MIX_MULT_R
    r ^= r >> 16;
    return r;
}

G2048_DEV Pcg64 pcg64_from_seed(uint64_t entropy)
{
    // entropy as little-endian uint32 words: one word below 2^32, else two (numpy's _coerce_to_uint32_array)
    const uint32_t e0 = (uint32_t)entropy, e1 = (uint32_t)(entropy >> 32);
    const int n_words = e1 != 0u ? 2 : 1;
    uint32_t pool[4];
    uint32_t hash_const = 0x43b0d7e5u; // INIT_A
    pool[0] = ss_hashmix(e0, hash_const);
    pool[1] = ss_hashmix(n_words > 1 ? e1 : 0u, hash_const);
    pool[2] = ss_hashmix(0u, hash_const);
    pool[3] = ss_hashmix(0u, hash_const);
#pragma unroll
    for (int src = 0; src < 4; ++src)
#pragma unroll
        for (int dst = 0; dst < 4; ++dst)
            if (src != dst)
                pool[dst] = ss_mix(pool[dst], ss_hashmix(pool[src], hash_const));
    // generate_state(4, uint64) = 8 uint32 words cycling through the pool
    uint32_t w[8];
    uint32_t hc = 0x8b51f9ddu; // INIT_B
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint32_t v = pool[i & 3] ^ hc;
        hc *= 0x58f38dedu; // MULT_B
        v *= hc;
        v ^= v >> 16;
        w[i] = v;
    }
    const uint64_t s0 = w[0] | ((uint64_t)w[1] << 32), s1 = w[2] | ((uint64_t)w[3] << 32);
    const uint64_t s2 = w[4] | ((uint64_t)w[5] << 32), s3 = w[6] | ((uint64_t)w[7] << 32);
    // pcg64_set_seed: initstate = (high s0, low s1), initseq = (high s2, low s3)
    Pcg64 r;
    r.inc_lo = (s3 << 1) | 1ull;               // inc = (initseq << 1) | 1
    r.inc_hi = (s2 << 1) | (s3 >> 63);
    r.state_lo = 0;
    r.state_hi = 0;
    r.buf = 0;
    (void)pcg64_next64(r);                     // step
    const uint64_t lo = r.state_lo + s1;       // state += initstate
    r.state_hi += s0 + (lo < r.state_lo ? 1ull : 0ull);
    r.state_lo = lo;
    (void)pcg64_next64(r);                     // step
    return r;
}

} // namespace g2048
