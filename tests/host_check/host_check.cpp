// host_check.cpp -- TEST HARNESS ONLY: compiles the device header g2048_device.h with g++
// (-DG2048_HOST_CHECK) so that the exact byte-parallel arithmetic the gfx950 kernels run can be
// unit-tested against the oracle in a container that has no GPU.  It mirrors the bodies of
// step_kernel / reset_kernel (g2048_kernels.hip) one board at a time.  Not part of the product.
#define G2048_HOST_CHECK 1
#include "../../gym-2048_amd/csrc/g2048_device.h"
#include "../../gym-2048_amd/csrc/g2048_pcg64.h"
#include "../../oracle/g2048_oracle.h"

#include <cstring>

using namespace g2048;

static Board load_board(const uint8_t *b)
{
    Board bd;
    std::memcpy(bd.r, b, 16);
    return bd;
}
static void store_board(uint8_t *b, const Board &bd) { std::memcpy(b, bd.r, 16); }

extern "C" {

void hostcheck_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    const Words w = philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
    std::memcpy(out, w.w, 16);
}

// one line through shift4 (byte lane `lane` of the four SWAR registers; other lanes get junk rows)
uint32_t hostcheck_shift(const uint8_t row[4], uint8_t out[4], int lane, const uint8_t junk[12])
{
    uint32_t r[4] = {0, 0, 0, 0};
    int j = 0;
    for (int l = 0; l < 4; ++l)
        for (int k = 0; k < 4; ++k)
            r[k] |= (uint32_t)(l == lane ? row[k] : junk[j++ % 12]) << (8 * l);
    // score of the junk lanes alone
    uint32_t rj[4];
    for (int k = 0; k < 4; ++k)
        rj[k] = r[k] & ~(0xffu << (8 * lane));
    uint32_t a = r[0], b = r[1], c = r[2], d = r[3];
    const uint32_t total = shift4(a, b, c, d);
    const uint32_t junk_score = shift4(rj[0], rj[1], rj[2], rj[3]);
    out[0] = (a >> (8 * lane)) & 0xff;
    out[1] = (b >> (8 * lane)) & 0xff;
    out[2] = (c >> (8 * lane)) & 0xff;
    out[3] = (d >> (8 * lane)) & 0xff;
    return total - junk_score;
}

int hostcheck_move(const uint8_t in[16], uint32_t action, uint8_t out[16], uint32_t *score)
{
    Board bd = load_board(in);
    const bool legal = move(bd, action, *score);
    store_board(out, bd);
    return legal;
}

uint32_t hostcheck_highest(const uint8_t in[16]) { return highest(load_board(in)); }
uint32_t hostcheck_count_empty(const uint8_t in[16]) { return count_empty(load_board(in)); }
int hostcheck_has_equal_neighbours(const uint8_t in[16]) { return has_equal_neighbours(load_board(in)); }

void hostcheck_add_tile(uint8_t b[16], uint32_t w)
{
    Board bd = load_board(b);
    add_tile(bd, w);
    store_board(b, bd);
}

void hostcheck_reset_batch(g2048o_batch *s, uint64_t n, uint64_t seed, uint64_t t, uint64_t board_offset,
                           uint32_t first_slot, int)
{
    for (uint64_t i = 0; i < n; ++i) {
        const uint32_t b = (uint32_t)(board_offset + i);
        const Words w = philox4x32_10((uint32_t)t, (uint32_t)(t >> 32), b, first_slot >> 2, (uint32_t)seed,
                                      (uint32_t)(seed >> 32));
        const uint32_t sl = first_slot & 3u;
        const uint32_t w1 = select_word(w, sl);
        uint32_t w2;
        if (sl == 3u)
            w2 = philox4x32_10((uint32_t)t, (uint32_t)(t >> 32), b, (first_slot >> 2) + 1u, (uint32_t)seed,
                               (uint32_t)(seed >> 32)).w[0];
        else
            w2 = select_word(w, sl + 1u);
        store_board(s->boards + 16 * i, fresh_board(w1, w2));
        s->score[i] = 0;
        s->ep_start[i] = (uint32_t)t;
    }
}

void hostcheck_step_batch(g2048o_batch *s, uint64_t n, uint64_t seed, uint64_t t, uint64_t board_offset,
                          float illegal_move_reward, int max_exp, int auto_reset, int)
{
    for (uint64_t i = 0; i < n; ++i) {
        Board bd = load_board(s->boards + 16 * i);
        int32_t score = s->score[i];
        const Words w = philox4x32_10((uint32_t)t, (uint32_t)(t >> 32), (uint32_t)(board_offset + i), 0u,
                                      (uint32_t)seed, (uint32_t)(seed >> 32));
        const uint32_t action = s->actions ? (s->actions[i] & 3u) : (w.w[3] >> 30);
        const StepResult r = step_env(bd, score, action, w, illegal_move_reward, (uint32_t)max_exp, auto_reset != 0);
        if (s->reward) s->reward[i] = r.reward;
        if (s->terminated) s->terminated[i] = r.terminated;
        if (s->illegal) s->illegal[i] = r.illegal;
        if (s->highest) s->highest[i] = (uint8_t)highest(r.terminal);
        if (r.terminated) {
            if (s->terminal_boards) store_board(s->terminal_boards + 16 * i, r.terminal);
            s->last_score[i] = r.terminal_score;
            s->last_len[i] = (int32_t)((uint32_t)t - s->ep_start[i]);
            s->ep_count[i] += 1;
            if (auto_reset) s->ep_start[i] = (uint32_t)t;
        }
        store_board(s->boards + 16 * i, bd);
        s->score[i] = score;
    }
}

// ---- the RECORD layout and the per-step kernel's flow (step_kernel in g2048_kernels.hip), one board
// at a time: state lives in `records` (16 B per board: cells + packed score deficit); the plain boards /
// scores of the batch struct are exported after every call so the test can compare them with the oracle's.
static const uint32_t kLutHost[32] = {G2048_MOVE_LUT_WORDS};

static MoveSel host_move_sel(uint32_t action)
{
    const uint32_t *r = kLutHost + 8 * (action & 3u);
    return MoveSel{r[0], r[1], r[2], r[3], r[4], r[5]};
}

static void export_record(g2048o_batch *s, uint64_t i, const Board &rec)
{
    store_board(s->boards + 16 * i, record_cells(rec));
    s->score[i] = (int32_t)record_score(rec);
}

int hostcheck_move_sel(const uint8_t in[16], uint32_t action, uint8_t out[16], uint32_t *score)
{
    Board bd = load_board(in);
    const bool legal = move_sel(bd, host_move_sel(action), *score);
    store_board(out, bd);
    return legal;
}

uint32_t hostcheck_potential(const uint8_t in[16]) { return potential(load_board(in)); }

void hostcheck_make_record(const uint8_t cells[16], uint32_t score, uint8_t rec[16])
{
    store_board(rec, make_record(load_board(cells), score));
}

uint32_t hostcheck_record_score(const uint8_t rec[16]) { return record_score(load_board(rec)); }
uint32_t hostcheck_record_deficit(const uint8_t rec[16]) { return record_deficit(load_board(rec)); }

// record_update with an arbitrary number of "+4" carries (exercises the ripple through the cell bits)
void hostcheck_record_bump(uint8_t rec[16], uint32_t times)
{
    Board raw = load_board(rec);
    for (uint32_t k = 0; k < times; ++k)
        record_update(raw, record_cells(raw), 0x80u);
    store_board(rec, raw);
}

void hostcheck_reset_records(g2048o_batch *s, uint8_t *records, uint64_t n, uint64_t seed, uint64_t t,
                             uint64_t board_offset, uint32_t first_slot)
{
    for (uint64_t i = 0; i < n; ++i) {
        const uint32_t b = (uint32_t)(board_offset + i);
        const Words w = philox4x32_10((uint32_t)t, (uint32_t)(t >> 32), b, first_slot >> 2, (uint32_t)seed,
                                      (uint32_t)(seed >> 32));
        const uint32_t sl = first_slot & 3u;
        const uint32_t w1 = select_word(w, sl);
        uint32_t w2;
        if (sl == 3u)
            w2 = philox4x32_10((uint32_t)t, (uint32_t)(t >> 32), b, (first_slot >> 2) + 1u, (uint32_t)seed,
                               (uint32_t)(seed >> 32)).w[0];
        else
            w2 = select_word(w, sl + 1u);
        const Board rec = fresh_record(w1, w2);
        store_board(records + 16 * i, rec);
        export_record(s, i, rec);
        s->ep_start[i] = (uint32_t)t;
    }
}

struct HostTables { // what LdsTables is on the device (g2048_kernels.hip)
    MoveSel move_sel(uint32_t action) const { return host_move_sel(action); }
    Board onehot_cell(uint32_t p) const
    {
        return Board{{onehot_cell_word(p, 0), onehot_cell_word(p, 1), onehot_cell_word(p, 2), onehot_cell_word(p, 3)}};
    }
};

void hostcheck_fresh_record_lut(uint32_t w1, uint32_t w2, uint8_t out_lut[16], uint8_t out_ref[16])
{
    store_board(out_lut, fresh_record_lut(w1, w2, HostTables{}));
    store_board(out_ref, fresh_record(w1, w2));
}

// step_kernel's body (g2048_kernels.hip): step_record + outputs + episode bookkeeping; last_records
// [n][16] is the engine's last_record array.
void hostcheck_step_records(g2048o_batch *s, uint8_t *records, uint8_t *last_records, uint64_t n, uint64_t seed, uint64_t t,
                            uint64_t board_offset, float illegal_move_reward, int max_exp, int auto_reset)
{
    for (uint64_t i = 0; i < n; ++i) {
        Board rec = load_board(records + 16 * i);
        const Words w = philox4x32_10((uint32_t)t, (uint32_t)(t >> 32), (uint32_t)(board_offset + i), 0u,
                                      (uint32_t)seed, (uint32_t)(seed >> 32));
        const uint32_t action = s->actions ? (s->actions[i] & 3u) : (w.w[3] >> 30);
        const StepOut o = step_record(rec, action, w, (uint32_t)max_exp, auto_reset != 0, HostTables{});
        store_board(records + 16 * i, rec);
        if (s->reward) s->reward[i] = o.legal ? (float)o.gain : illegal_move_reward;
        if (s->terminated) s->terminated[i] = o.terminated;
        if (s->illegal) s->illegal[i] = !o.legal;
        if (s->highest) s->highest[i] = (uint8_t)highest(record_cells(o.terminal));
        if (o.terminated) {
            store_board(last_records + 16 * i, o.terminal);
            if (s->terminal_boards) store_board(s->terminal_boards + 16 * i, record_cells(o.terminal));
            s->last_score[i] = (int32_t)record_score(load_board(last_records + 16 * i)); // g2048_get_last_scores
            s->last_len[i] = (int32_t)((uint32_t)t - s->ep_start[i]);
            s->ep_count[i] += 1;
            if (auto_reset) s->ep_start[i] = (uint32_t)t;
        }
        export_record(s, i, rec);
    }
}

// the fused kernels' way through the record: unpack once, step_env, pack
void hostcheck_step_records_unpacked(g2048o_batch *s, uint8_t *records, uint8_t *, uint64_t n, uint64_t seed, uint64_t t,
                                     uint64_t board_offset, float illegal_move_reward, int max_exp, int auto_reset)
{
    for (uint64_t i = 0; i < n; ++i) {
        const Board raw = load_board(records + 16 * i);
        Board bd = record_cells(raw);
        int32_t score = (int32_t)record_score(raw);
        const Words w = philox4x32_10((uint32_t)t, (uint32_t)(t >> 32), (uint32_t)(board_offset + i), 0u,
                                      (uint32_t)seed, (uint32_t)(seed >> 32));
        const uint32_t action = s->actions ? (s->actions[i] & 3u) : (w.w[3] >> 30);
        const StepResult r = step_env(bd, score, action, w, illegal_move_reward, (uint32_t)max_exp, auto_reset != 0);
        if (s->reward) s->reward[i] = r.reward;
        if (s->terminated) s->terminated[i] = r.terminated;
        if (s->illegal) s->illegal[i] = r.illegal;
        if (s->highest) s->highest[i] = (uint8_t)highest(r.terminal);
        if (r.terminated) {
            if (s->terminal_boards) store_board(s->terminal_boards + 16 * i, r.terminal);
            s->last_score[i] = r.terminal_score;
            s->last_len[i] = (int32_t)((uint32_t)t - s->ep_start[i]);
            s->ep_count[i] += 1;
            if (auto_reset) s->ep_start[i] = (uint32_t)t;
        }
        const Board rec = make_record(bd, (uint32_t)score);
        store_board(records + 16 * i, rec);
        export_record(s, i, rec);
    }
}

// ---- numpy-compatible RNG mode (g2048_pcg64.h); rng = [n][5] uint64 as in the oracle
static Pcg64 load_rng(const g2048o_pcg64 *r) { return Pcg64{r->state_lo, r->state_hi, r->inc_lo, r->inc_hi, r->buf}; }
static void store_rng(g2048o_pcg64 *r, const Pcg64 &p)
{
    r->state_lo = p.state_lo; r->state_hi = p.state_hi; r->inc_lo = p.inc_lo; r->inc_hi = p.inc_hi; r->buf = p.buf;
}

// The observation piece of ONE wavefront (64 boards, records with their spare bits set as the engine keeps them),
// built exactly as emit_onehot does on the device: records parked with the cells masked, chunk s * 64 + lane of store
// s from onehot_chunk<OBS>.  out: 64 boards x 16 channels x 16 cells of 1 / 2 / 4 bytes.
void hostcheck_onehot_wave(const uint8_t *records, int obs_dtype, uint8_t *out)
{
    Cells16 recs[64];
    for (int l = 0; l < 64; ++l) {
        const Board cells = record_cells(load_board(records + 16 * l));
        for (int k = 0; k < 4; ++k)
            recs[l].r[k] = cells.r[k];
    }
    const uint32_t stores = 16u << obs_dtype;
    for (uint32_t s = 0; s < stores; ++s)
        for (uint32_t lane = 0; lane < 64; ++lane) {
            uint32_t board;
            const Chunk16 c = obs_dtype == 0 ? onehot_chunk<0>(recs, s, lane, board)
                              : obs_dtype == 1 ? onehot_chunk<1>(recs, s, lane, board) : onehot_chunk<2>(recs, s, lane, board);
            std::memcpy(out + (static_cast<size_t>(s) * 64 + lane) * 16, c.w, 16);
        }
}

uint64_t hostcheck_pcg64_next64(g2048o_pcg64 *r) { Pcg64 p = load_rng(r); const uint64_t v = pcg64_next64(p); store_rng(r, p); return v; }
uint32_t hostcheck_pcg64_next32(g2048o_pcg64 *r) { Pcg64 p = load_rng(r); const uint32_t v = pcg64_next32(p); store_rng(r, p); return v; }
uint32_t hostcheck_pcg64_interval(g2048o_pcg64 *r, uint32_t mx) { Pcg64 p = load_rng(r); const uint32_t v = pcg64_interval(p, mx); store_rng(r, p); return v; }
void hostcheck_pcg64_from_seed(uint64_t entropy, g2048o_pcg64 *out) { store_rng(out, pcg64_from_seed(entropy)); }
uint32_t hostcheck_empty_mask16(const uint8_t b[16]) { return empty_mask16(load_board(b)); }

void hostcheck_reset_batch_numpy(g2048o_batch *s, g2048o_pcg64 *rng, uint64_t n, uint64_t t, int)
{
    for (uint64_t i = 0; i < n; ++i) {
        Pcg64 p = load_rng(&rng[i]);
        const Board rec = fresh_record_numpy(p); // what reset_numpy_kernel and the step kernel's in-block reset store
        store_rng(&rng[i], p);
        store_board(s->boards + 16 * i, record_cells(rec));
        s->score[i] = (int32_t)record_score(rec); // 0: the deficit of a fresh record is its potential
        s->ep_start[i] = (uint32_t)t;
    }
}

void hostcheck_step_batch_numpy(g2048o_batch *s, g2048o_pcg64 *rng, uint64_t n, uint64_t seed, uint64_t t,
                                uint64_t board_offset, float illegal_move_reward, int max_exp, int auto_reset, int)
{
    for (uint64_t i = 0; i < n; ++i) {
        Board bd = load_board(s->boards + 16 * i);
        int32_t score = s->score[i];
        Pcg64 p = load_rng(&rng[i]);
        uint32_t action;
        if (s->actions)
            action = s->actions[i] & 3u;
        else
            action = philox4x32_10((uint32_t)t, (uint32_t)(t >> 32), (uint32_t)(board_offset + i), 0u, (uint32_t)seed,
                                   (uint32_t)(seed >> 32)).w[3] >> 30;
        // exactly step_numpy_kernel's sequence, on the RECORD: play_record_numpy, then (auto-reset) fresh_record_numpy
        Board rec = make_record(bd, (uint32_t)score);
        const NumpyStepOut o = play_record_numpy(rec, action, p, (uint32_t)max_exp, auto_reset != 0);
        if (s->reward) s->reward[i] = o.legal ? (float)o.gain : illegal_move_reward;
        if (s->terminated) s->terminated[i] = o.terminated;
        if (s->illegal) s->illegal[i] = !o.legal;
        if (s->highest) s->highest[i] = (uint8_t)o.top;
        if (o.terminated) {
            if (s->terminal_boards) store_board(s->terminal_boards + 16 * i, record_cells(rec));
            s->last_score[i] = (int32_t)record_score(rec);
            s->last_len[i] = (int32_t)((uint32_t)t - s->ep_start[i]);
            s->ep_count[i] += 1;
            if (auto_reset) {
                s->ep_start[i] = (uint32_t)t;
                rec = finish_record_numpy(p, o.have_first, o.first_cell, o.first_four);
            }
        }
        store_rng(&rng[i], p);
        store_board(s->boards + 16 * i, record_cells(rec));
        s->score[i] = (int32_t)record_score(rec);
    }
}

} // extern "C"
