/*
 * g2048_oracle.c -- CPU ORACLE (test infrastructure only; see g2048_oracle.h).
 *
 * Plain-C restatement of /root/reference/env/envs/game2048_env.py.  Every function cites the
 * reference lines it follows.  Like the reference it works on int64 tile values in a row-major
 * 4x4 matrix; the batch driver at the bottom converts from/to the device's uint8 exponents.
 */
#include "g2048_oracle.h"

#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---------------------------------------------------------------------------- Philox4x32-10 */

#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

void g2048o_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; ++round) {
        uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += PHILOX_W0;
        k1 += PHILOX_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

uint32_t g2048o_spawn_word(uint64_t seed, uint64_t t, uint32_t board, uint32_t slot)
{
    uint32_t ctr[4] = { (uint32_t)t, (uint32_t)(t >> 32), board, slot >> 2 };
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    uint32_t out[4];
    g2048o_philox4x32_10(ctr, key, out);
    return out[slot & 3];
}

uint8_t g2048o_random_action(uint64_t seed, uint64_t t, uint32_t board)
{
    return (uint8_t)(g2048o_spawn_word(seed, t, board, 3) >> 30);
}

/* game2048_env.py:168  val = 2 if self.np_random.random() < 0.9 else 4
 * with random() := frac(w * n_empty / 2^32) = ((w * n_empty) mod 2^32) / 2^32 -- what is left of the word once the
 * position k = floor(w * n_empty / 2^32) has been taken out of it -- evaluated here in double exactly as Python would
 * (a 32-bit integer over 2^32 is exact in a double).  P(2) is within n_empty / 2^32 < 4e-9 of the reference's 0.9, and
 * value and position are independent to the same precision. */
int64_t g2048o_spawn_value(uint32_t w, uint32_t n_empty)
{
    double u = (double)(uint32_t)((uint64_t)w * n_empty) / 4294967296.0;
    return u < 0.9 ? 2 : 4;
}

/* game2048_env.py:169-175  shuffle(positions) then first empty: the injected shuffle puts the
 * k-th (row-major) empty cell first, k = floor(w * n_empty / 2^32). */
uint32_t g2048o_spawn_rank(uint32_t w, uint32_t n_empty)
{
    return (uint32_t)(((uint64_t)w * n_empty) >> 32);
}

/* ----------------------------------------------------------------------- reference functions */

/* game2048_env.py:243-260 */
int64_t g2048o_shift(const int64_t row[4], int64_t combined_row[4])
{
    int64_t move_score = 0;
    int output_index = 0;
    int can_merge = 0;
    combined_row[0] = combined_row[1] = combined_row[2] = combined_row[3] = 0; /* :246 */
    for (int i = 0; i < 4; ++i) {                                              /* :249 */
        int64_t val = row[i];
        if (val == 0)                                                          /* :250-251 */
            continue;
        if (can_merge && combined_row[output_index - 1] == val) {              /* :252 */
            combined_row[output_index - 1] *= 2;                               /* :253 */
            move_score += combined_row[output_index - 1];                      /* :254 */
            can_merge = 0;                                                     /* :255 */
        } else {
            combined_row[output_index] = val;                                  /* :257 */
            output_index += 1;                                                 /* :258 */
            can_merge = 1;                                                     /* :259 */
        }
    }
    return move_score;
}

/* game2048_env.py:194-241 */
int g2048o_move(int64_t M[16], int direction, int trial, int64_t *move_score_out)
{
    int changed = 0;
    int64_t move_score = 0;
    int dir_div_two = direction / 2;                   /* :210 int(direction / 2) */
    int dir_mod_two = direction % 2;                   /* :211 */
    int shift_direction = dir_mod_two ^ dir_div_two;   /* :212 0 = towards up/left */

    for (int line = 0; line < 4; ++line) {
        int64_t old[4], new_[4];
        int idx[4];
        for (int k = 0; k < 4; ++k) {
            /* :216-217 column `line`  /  :228-229 row `line`; reversed when shift_direction
             * (:218-219 / :230-231) */
            int kk = shift_direction ? 3 - k : k;
            idx[k] = (dir_mod_two == 0) ? kk * 4 + line : line * 4 + kk;
            old[k] = M[idx[k]];
        }
        move_score += g2048o_shift(old, new_);          /* :220-221 / :232-233 */
        if (memcmp(old, new_, sizeof old) != 0) {       /* :222 / :234 */
            changed = 1;
            if (!trial)                                 /* :224-225 / :236-237 */
                for (int k = 0; k < 4; ++k)
                    M[idx[k]] = new_[k];
        }
    }
    if (move_score_out)
        *move_score_out = move_score;
    return changed;                                     /* :238-239 raise IllegalMove if !changed */
}

/* game2048_env.py:190-192 */
int64_t g2048o_highest(const int64_t M[16])
{
    int64_t h = M[0];
    for (int i = 1; i < 16; ++i)
        if (M[i] > h)
            h = M[i];
    return h;
}

/* game2048_env.py:262-280 */
int g2048o_isend(const int64_t M[16], int64_t max_tile)
{
    if (max_tile != 0 && g2048o_highest(M) == max_tile) /* :267-268 */
        return 1;
    for (int i = 0; i < 16; ++i)                        /* :270-271 */
        if (M[i] == 0)
            return 0;
    for (int direction = 0; direction < 4; ++direction) { /* :273-279 */
        int64_t tmp[16];
        memcpy(tmp, M, sizeof tmp);
        if (g2048o_move(tmp, direction, 1, 0))
            return 0;
    }
    return 1;                                           /* :280 */
}

/* game2048_env.py:166-176 */
int g2048o_add_tile(int64_t M[16], uint32_t w)
{
    uint32_t n_empty = 0;
    for (int i = 0; i < 16; ++i)
        n_empty += (M[i] == 0);
    int64_t val = g2048o_spawn_value(w, n_empty);       /* :168 (drawn before the position, used after it) */
    if (n_empty == 0)
        return -1;                                      /* :176 assert False */
    uint32_t k = g2048o_spawn_rank(w, n_empty);         /* :169-170 injected shuffle */
    for (int i = 0; i < 16; ++i) {                      /* :171-175 first empty in shuffled order */
        if (M[i] == 0) {
            if (k == 0) {
                M[i] = val;
                return i;
            }
            --k;
        }
    }
    return -1;
}

/* game2048_env.py:17-32 */
void g2048o_stack(const int64_t M[16], int64_t out[256])
{
    for (int i = 0; i < 16; ++i)                        /* :25 layer 0 = empty */
        out[i] = (M[i] == 0);
    for (int layer = 1; layer <= 15; ++layer) {         /* :28-30 layers 1..15 = 2^1..2^15 */
        int64_t representation = (int64_t)1 << layer;
        for (int i = 0; i < 16; ++i)
            out[layer * 16 + i] = (M[i] == representation);
    }
}

/* --------------------------------------------------------------------------------- single env */

void g2048o_env_init(g2048o_env *e, uint64_t seed, uint32_t board)
{
    memset(e, 0, sizeof *e);
    e->illegal_move_reward = 0.0;                       /* :53 */
    e->max_tile = 0;                                    /* :54 None */
    e->seed = seed;
    e->board = board;
}

/* game2048_env.py:102-111 */
void g2048o_env_reset(g2048o_env *e, int reseed, uint64_t seed)
{
    if (reseed) {                                       /* :103 super().reset(seed=seed) */
        e->seed = seed;
        e->t = 0;
        e->slot = 0;
    }
    memset(e->M, 0, sizeof e->M);                       /* :104 */
    e->score = 0;                                       /* :105 */
    g2048o_add_tile(e->M, g2048o_spawn_word(e->seed, e->t, e->board, e->slot++)); /* :108 */
    g2048o_add_tile(e->M, g2048o_spawn_word(e->seed, e->t, e->board, e->slot++)); /* :109 */
}

/* game2048_env.py:76-100 */
int g2048o_env_step(g2048o_env *e, int action, double *reward, int *illegal, int64_t *highest)
{
    int terminated;
    int64_t ms = 0;
    e->t += 1;                                          /* new transaction of the spawn stream */
    e->slot = 0;
    if (g2048o_move(e->M, action, 0, &ms)) {            /* :85 */
        e->score += (double)ms;                         /* :86 */
        g2048o_add_tile(e->M, g2048o_spawn_word(e->seed, e->t, e->board, e->slot++)); /* :88 */
        terminated = g2048o_isend(e->M, e->max_tile);   /* :89 */
        *reward = (double)ms;                           /* :90 */
        *illegal = 0;
    } else {                                            /* :91-95 except IllegalMove */
        *illegal = 1;
        terminated = 1;
        *reward = e->illegal_move_reward;
    }
    *highest = g2048o_highest(e->M);                    /* :97 */
    return terminated;
}

/* ----------------------------------------------------------------------------- batch driver */

void g2048o_exp_to_values(const uint8_t b[16], int64_t M[16])
{
    for (int i = 0; i < 16; ++i)
        M[i] = b[i] ? ((int64_t)1 << b[i]) : 0;
}

void g2048o_values_to_exp(const int64_t M[16], uint8_t b[16])
{
    for (int i = 0; i < 16; ++i) {
        uint8_t e = 0;
        int64_t v = M[i];
        while (v > 1) {
            v >>= 1;
            ++e;
        }
        b[i] = e;
    }
}

static void set_threads(int threads)
{
#ifdef _OPENMP
    if (threads > 0)
        omp_set_num_threads(threads);
#else
    (void)threads;
#endif
}

void g2048o_reset_batch(g2048o_batch *s, uint64_t n, uint64_t seed, uint64_t t,
                        uint64_t board_offset, uint32_t first_slot, int threads)
{
    set_threads(threads);
#pragma omp parallel for schedule(static) if (threads != 1)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        int64_t M[16] = { 0 };                          /* :104 */
        uint32_t b = (uint32_t)(board_offset + (uint64_t)i);
        g2048o_add_tile(M, g2048o_spawn_word(seed, t, b, first_slot));     /* :108 */
        g2048o_add_tile(M, g2048o_spawn_word(seed, t, b, first_slot + 1)); /* :109 */
        g2048o_values_to_exp(M, s->boards + 16 * i);
        if (s->score)
            s->score[i] = 0;                            /* :105 */
        if (s->ep_start)
            s->ep_start[i] = (uint32_t)t;
    }
}

void g2048o_step_batch(g2048o_batch *s, uint64_t n, uint64_t seed, uint64_t t,
                       uint64_t board_offset, float illegal_move_reward, int max_exp,
                       int auto_reset, int threads)
{
    const int64_t max_tile = max_exp ? ((int64_t)1 << max_exp) : 0;
    set_threads(threads);
#pragma omp parallel for schedule(static) if (threads != 1)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        int64_t M[16], ms = 0, hi;
        uint32_t b = (uint32_t)(board_offset + (uint64_t)i);
        uint32_t slot = 0;
        int action, terminated, illegal;
        float reward;
        int32_t score = s->score ? s->score[i] : 0;

        g2048o_exp_to_values(s->boards + 16 * i, M);
        action = (s->actions ? s->actions[i] : g2048o_random_action(seed, t, b)) & 3;

        if (g2048o_move(M, action, 0, &ms)) {           /* :85 */
            score += (int32_t)ms;                       /* :86 */
            g2048o_add_tile(M, g2048o_spawn_word(seed, t, b, slot++)); /* :88 */
            terminated = g2048o_isend(M, max_tile);     /* :89 */
            reward = (float)ms;                         /* :90 */
            illegal = 0;
        } else {                                        /* :91-95 */
            illegal = 1;
            terminated = 1;
            reward = illegal_move_reward;
        }
        hi = g2048o_highest(M);                         /* :97 */

        if (s->reward) s->reward[i] = reward;
        if (s->terminated) s->terminated[i] = (uint8_t)terminated;
        if (s->illegal) s->illegal[i] = (uint8_t)illegal;
        if (s->highest) {
            uint8_t e = 0;
            while (hi > 1) { hi >>= 1; ++e; }
            s->highest[i] = e;
        }
        if (terminated) {
            if (s->terminal_boards) g2048o_values_to_exp(M, s->terminal_boards + 16 * i);
            if (s->last_score) s->last_score[i] = score;
            if (s->last_len && s->ep_start) s->last_len[i] = (int32_t)((uint32_t)t - s->ep_start[i]);
            if (s->ep_count) s->ep_count[i] += 1;
            if (auto_reset) {                           /* caller's `if terminated: env.reset()` */
                memset(M, 0, sizeof M);                 /* :104 */
                score = 0;                              /* :105 */
                g2048o_add_tile(M, g2048o_spawn_word(seed, t, b, slot++)); /* :108 */
                g2048o_add_tile(M, g2048o_spawn_word(seed, t, b, slot++)); /* :109 */
                if (s->ep_start) s->ep_start[i] = (uint32_t)t;
            }
        }
        g2048o_values_to_exp(M, s->boards + 16 * i);
        if (s->score) s->score[i] = score;
    }
}

/* ----------------------------------------------------------------- numpy-compatible RNG mode */

typedef unsigned __int128 u128;

uint64_t g2048o_pcg64_next64(g2048o_pcg64 *r)
{
    const u128 mult = ((u128)0x2360ED051FC65DA4ull << 64) | 0x4385DF649FCCF645ull;
    u128 state = ((u128)r->state_hi << 64) | r->state_lo;
    const u128 inc = ((u128)r->inc_hi << 64) | r->inc_lo;
    state = state * mult + inc;                       /* pcg_setseq_128_step_r */
    r->state_lo = (uint64_t)state;
    r->state_hi = (uint64_t)(state >> 64);
    const uint64_t x = r->state_hi ^ r->state_lo;     /* pcg_output_xsl_rr_128_64 */
    const unsigned rot = (unsigned)(r->state_hi >> 58);
    return (x >> rot) | (x << ((64 - rot) & 63));
}

uint32_t g2048o_pcg64_next32(g2048o_pcg64 *r)
{
    if (r->buf >> 32) {                               /* has_uint32 */
        const uint32_t v = (uint32_t)r->buf;
        r->buf = 0;
        return v;
    }
    const uint64_t next = g2048o_pcg64_next64(r);
    r->buf = (next >> 32) | (1ull << 32);
    return (uint32_t)next;
}

uint32_t g2048o_pcg64_interval(g2048o_pcg64 *r, uint32_t max)
{
    uint32_t mask = max, value;
    if (max == 0)
        return 0;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    do {
        value = g2048o_pcg64_next32(r) & mask;
    } while (value > max);
    return value;
}

/* game2048_env.py:166-176 */
int g2048o_add_tile_numpy(int64_t M[16], g2048o_pcg64 *r)
{
    /* :168 random() < 0.9 in double, exactly as Python evaluates it */
    const double u = (double)(g2048o_pcg64_next64(r) >> 11) * (1.0 / 9007199254740992.0);
    const int64_t val = u < 0.9 ? 2 : 4;
    int positions[16];
    for (int i = 0; i < 16; ++i)                       /* :169 row-major (r, c) -> r*4+c */
        positions[i] = i;
    for (int i = 15; i >= 1; --i) {                    /* :170 Generator.shuffle(list) */
        const int j = (int)g2048o_pcg64_interval(r, (uint32_t)i);
        const int tmp = positions[i];
        positions[i] = positions[j];
        positions[j] = tmp;
    }
    for (int i = 0; i < 16; ++i)                       /* :171-175 */
        if (M[positions[i]] == 0) {
            M[positions[i]] = val;
            return positions[i];
        }
    return -1;                                         /* :176 */
}

void g2048o_reset_batch_numpy(g2048o_batch *s, g2048o_pcg64 *rng, uint64_t n, uint64_t t, int threads)
{
    set_threads(threads);
#pragma omp parallel for schedule(static) if (threads != 1)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        int64_t M[16] = { 0 };                          /* :104 */
        g2048o_add_tile_numpy(M, &rng[i]);              /* :108 */
        g2048o_add_tile_numpy(M, &rng[i]);              /* :109 */
        g2048o_values_to_exp(M, s->boards + 16 * i);
        if (s->score) s->score[i] = 0;
        if (s->ep_start) s->ep_start[i] = (uint32_t)t;
    }
}

void g2048o_step_batch_numpy(g2048o_batch *s, g2048o_pcg64 *rng, uint64_t n, uint64_t seed, uint64_t t,
                             uint64_t board_offset, float illegal_move_reward, int max_exp, int auto_reset,
                             int threads)
{
    const int64_t max_tile = max_exp ? ((int64_t)1 << max_exp) : 0;
    set_threads(threads);
#pragma omp parallel for schedule(static) if (threads != 1)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        int64_t M[16], ms = 0, hi;
        int action, terminated, illegal;
        float reward;
        int32_t score = s->score ? s->score[i] : 0;
        g2048o_exp_to_values(s->boards + 16 * i, M);
        action = (s->actions ? s->actions[i]
                             : g2048o_random_action(seed, t, (uint32_t)(board_offset + (uint64_t)i))) & 3;
        if (g2048o_move(M, action, 0, &ms)) {           /* :85 */
            score += (int32_t)ms;                       /* :86 */
            g2048o_add_tile_numpy(M, &rng[i]);          /* :88 */
            terminated = g2048o_isend(M, max_tile);     /* :89 */
            reward = (float)ms;
            illegal = 0;
        } else {                                        /* :91-95 */
            illegal = 1;
            terminated = 1;
            reward = illegal_move_reward;
        }
        hi = g2048o_highest(M);
        if (s->reward) s->reward[i] = reward;
        if (s->terminated) s->terminated[i] = (uint8_t)terminated;
        if (s->illegal) s->illegal[i] = (uint8_t)illegal;
        if (s->highest) {
            uint8_t e = 0;
            while (hi > 1) { hi >>= 1; ++e; }
            s->highest[i] = e;
        }
        if (terminated) {
            if (s->terminal_boards) g2048o_values_to_exp(M, s->terminal_boards + 16 * i);
            if (s->last_score) s->last_score[i] = score;
            if (s->last_len && s->ep_start) s->last_len[i] = (int32_t)((uint32_t)t - s->ep_start[i]);
            if (s->ep_count) s->ep_count[i] += 1;
            if (auto_reset) {                           /* `if terminated: env.reset()` (:102-111) */
                memset(M, 0, sizeof M);
                score = 0;
                g2048o_add_tile_numpy(M, &rng[i]);
                g2048o_add_tile_numpy(M, &rng[i]);
                if (s->ep_start) s->ep_start[i] = (uint32_t)t;
            }
        }
        g2048o_values_to_exp(M, s->boards + 16 * i);
        if (s->score) s->score[i] = score;
    }
}

void g2048o_onehot_batch(const uint8_t *boards, uint64_t n, uint8_t *out)
{
    for (uint64_t i = 0; i < n; ++i) {
        int64_t M[16], st[256];
        g2048o_exp_to_values(boards + 16 * i, M);
        g2048o_stack(M, st);
        for (int j = 0; j < 256; ++j)
            out[256 * i + j] = (uint8_t)st[j];
    }
}
