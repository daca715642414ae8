#!/usr/bin/env python
"""Generate the golden vectors in tests/golden/ FROM THE IMPORTED REFERENCE.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py [--validate-steps 1000000]

What it does
------------
1. Imports the *unmodified* reference package ``env`` from /root/reference.  ``gymnasium`` is not
   installed in this image, so a minimal in-memory stand-in for the three things the env touches
   (``gym.Env`` with the seeding ``reset``/``np_random`` property, ``spaces.Discrete/Box`` and
   ``register``) is put into ``sys.modules`` first.  The stand-in is used only by
   this generator; nothing of the reference or of the stand-in is written to the repo.
2. Replaces ``env.np_random`` by ``oracle.cpu_ref.SpawnStream`` (RNG injection) and records what
   the reference computes: shift/move/isend/stack known answers and multi-step trajectories of the
   loop ``obs, r, term, _, info = env.step(a); if term: env.reset()``.
3. Converts the reference's own fixture ``data/test_data.csv`` (848 recorded transitions) to npz.
4. Cross-checks the C oracle and the Python oracle against the reference over ``--validate-steps``
   steps and writes the report to ``VALIDATION.txt``.

The outputs are *data* (inputs + expected outputs); no reference source text is stored.
"""
from __future__ import annotations

import argparse
import hashlib
import itertools
import os
import sys
import time
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import OracleBatch, load as load_oracle  # noqa: E402
from oracle.cpu_ref import RefEnv, SpawnStream, random_action, values_to_exp  # noqa: E402


def install_gymnasium_stand_in():
    gym = types.ModuleType("gymnasium")

    class Env:
        _np_random = None

        def reset(self, *, seed=None, options=None):
            if seed is not None:
                self._np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))

        @property
        def np_random(self):
            if self._np_random is None:
                self._np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence()))
            return self._np_random

        @np_random.setter
        def np_random(self, value):
            self._np_random = value

    spaces = types.ModuleType("gymnasium.spaces")

    class Discrete:
        def __init__(self, n):
            self.n = n

    class Box:
        def __init__(self, low, high, shape, dtype=None):
            self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

    spaces.Discrete, spaces.Box = Discrete, Box
    envs = types.ModuleType("gymnasium.envs")
    registration = types.ModuleType("gymnasium.envs.registration")
    registration.register = lambda **kw: None
    envs.registration = registration
    gym.Env, gym.spaces, gym.envs = Env, spaces, envs
    sys.modules.update({"gymnasium": gym, "gymnasium.spaces": spaces, "gymnasium.envs": envs,
                        "gymnasium.envs.registration": registration})


def import_reference():
    install_gymnasium_stand_in()
    sys.path.insert(0, REFERENCE)
    import env.envs.game2048_env as ref  # noqa
    return ref


class RefDriver:
    """The unmodified reference env + injected spawn stream + the canonical driving loop."""

    def __init__(self, ref, seed, board=0, illegal_move_reward=None, max_tile=None):
        self.ref = ref
        self.env = ref.Game2048Env()
        if illegal_move_reward is not None:
            self.env.set_illegal_move_reward(illegal_move_reward)
        if max_tile is not None:
            self.env.set_max_tile(max_tile)
        self.stream = SpawnStream(self.env, seed, board)
        self.env.np_random = self.stream
        self.env.reset()

    def legal_actions(self):
        out = []
        for d in range(4):
            try:
                self.env.move(d, trial=True)
                out.append(d)
            except self.ref.IllegalMove:
                pass
        return out

    def step(self, action, auto_reset=True):
        self.stream.begin_step()
        obs, reward, terminated, truncated, info = self.env.step(action)
        assert truncated is False
        rec = dict(reward=reward, terminated=bool(terminated), illegal=bool(info["illegal_move"]),
                   highest=int(info["highest"]), terminal_board=self.env.get_board().copy(),
                   terminal_score=self.env.score, obs=obs)
        if terminated and auto_reset:
            self.env.reset()
        rec["board"] = self.env.get_board().copy()
        rec["score"] = self.env.score
        return rec


# ------------------------------------------------------------------------------------- pieces


def gen_shift_table(ref):
    """Exhaustive shift() table over exponents 0..17 (104 976 rows)."""
    e = ref.Game2048Env()
    out = np.zeros((18 ** 4, 4), np.uint8)
    score = np.zeros(18 ** 4, np.int32)
    for i, exps in enumerate(itertools.product(range(18), repeat=4)):
        row = [0 if x == 0 else 1 << x for x in exps]
        new, ms = e.shift(row)
        out[i] = values_to_exp(new)
        score[i] = ms
    return dict(out=out, score=score)


def random_boards(rng, n, max_exp=11):
    """Boards with varied fill levels and lots of equal neighbours (exponents)."""
    boards = np.zeros((n, 16), np.uint8)
    for i in range(n):
        fill = rng.integers(1, 17)
        top = rng.integers(1, max_exp + 1)
        cells = rng.choice(16, size=fill, replace=False)
        boards[i, cells] = rng.integers(1, top + 1, size=fill)
    return boards


def gen_move_table(ref, rng, n=1536):
    e = ref.Game2048Env()
    boards = random_boards(rng, n)
    boards[:64] = random_boards(rng, 64, max_exp=17)           # very high tiles
    boards[64:128] = rng.integers(1, 3, size=(64, 16))         # full boards, many merges
    new = np.zeros((n, 4, 16), np.uint8)
    score = np.zeros((n, 4), np.int32)
    legal = np.zeros((n, 4), np.uint8)
    end = np.zeros(n, np.uint8)
    highest = np.zeros(n, np.uint8)
    for i in range(n):
        vals = np.where(boards[i] > 0, 1 << boards[i].astype(np.int64), 0).reshape(4, 4)
        e.set_board(vals.copy())
        end[i] = e.isend()
        highest[i] = values_to_exp(e.highest())
        for d in range(4):
            e.set_board(vals.copy())
            try:
                score[i, d] = e.move(d)
                legal[i, d] = 1
            except ref.IllegalMove:
                pass
            new[i, d] = values_to_exp(e.get_board()).reshape(16)
    return dict(boards=boards, new=new, score=score, legal=legal, isend=end, highest=highest)


def gen_isend_table(ref, rng, n=512):
    """Full boards (the only place isend's trial moves matter) + max_tile cases."""
    e = ref.Game2048Env()
    boards = np.zeros((n, 16), np.uint8)
    for i in range(n):
        if i % 2 == 0:   # checkerboard-ish full boards with a few forced equal neighbours
            b = np.array([[((r + c) % 2) + 1 + 2 * ((r * 4 + c) % 3) for c in range(4)] for r in range(4)])
            b = b.reshape(16)
            for _ in range(rng.integers(0, 3)):
                j = rng.integers(0, 16)
                b[j] = b[(j + rng.choice([1, 4])) % 16]
            boards[i] = b
        else:
            boards[i] = rng.integers(1, 6, size=16)
    max_exp = rng.integers(0, 12, size=n).astype(np.uint8)      # 0 = no max_tile
    boards[-32:, 3] = 0                                         # some with an empty cell
    end = np.zeros(n, np.uint8)
    for i in range(n):
        e.set_max_tile(None if max_exp[i] == 0 else int(1 << int(max_exp[i])))
        e.set_board(np.where(boards[i] > 0, 1 << boards[i].astype(np.int64), 0).reshape(4, 4))
        end[i] = e.isend()
    return dict(boards=boards, max_exp=max_exp, isend=end)


def gen_stack_table(ref, rng, n=64):
    boards = random_boards(rng, n, max_exp=15)
    boards[0] = np.arange(16)                    # every layer once
    boards[1] = np.arange(2, 18)                 # includes 2^16 and 2^17: all-zero columns
    boards[2] = 0
    out = np.zeros((n, 16, 4, 4), np.uint8)
    for i in range(n):
        vals = np.where(boards[i] > 0, 1 << boards[i].astype(np.int64), 0).reshape(4, 4)
        s = ref.stack(vals)
        assert s.shape == (16, 4, 4) and set(np.unique(s)) <= {0, 1}
        out[i] = s
    return dict(boards=boards, onehot=out)


def greedy_action(legal, random_a, t):
    """'greedy' test policy: the random action on every 41st step (may be illegal -> episode ends
    with the illegal-move reward), the random action if legal on every 7th, else the first legal
    of left, down, right, up.  Gives long episodes, big tiles and full-board endings."""
    if t % 41 == 0 or (t % 7 == 0 and random_a in legal):
        return random_a
    return next((d for d in (3, 2, 1, 0) if d in legal), random_a)


def gen_trajectories(ref, name, seed, n_boards, n_steps, policy, board_offset=0,
                     illegal_move_reward=None, max_tile=None, auto_reset=True):
    """policy: 'random' (the synthetic benchmark policy), 'greedy' (see greedy_action) or 'corner'."""
    shape = (n_boards, n_steps)
    out = dict(actions=np.zeros(shape, np.uint8), reward=np.zeros(shape, np.float32),
               terminated=np.zeros(shape, np.uint8), illegal=np.zeros(shape, np.uint8),
               highest=np.zeros(shape, np.uint8), score=np.zeros(shape, np.int32),
               boards=np.zeros(shape + (16,), np.uint8), terminal_boards=np.zeros(shape + (16,), np.uint8),
               terminal_score=np.zeros(shape, np.int32), initial_boards=np.zeros((n_boards, 16), np.uint8))
    for b in range(n_boards):
        drv = RefDriver(ref, seed, board_offset + b, illegal_move_reward, max_tile)
        out["initial_boards"][b] = values_to_exp(drv.env.get_board()).reshape(16)
        for s in range(n_steps):
            t = s + 1
            a = random_action(seed, t, board_offset + b)
            if policy == "greedy":
                a = greedy_action(drv.legal_actions(), a, t)
            elif policy == "corner":      # always legal: first legal of left, down, right, up -> games end on full boards
                legal = drv.legal_actions()
                a = next((d for d in (3, 2, 1, 0) if d in legal), a)
            rec = drv.step(a, auto_reset)
            out["actions"][b, s] = a
            out["reward"][b, s] = rec["reward"]
            out["terminated"][b, s] = rec["terminated"]
            out["illegal"][b, s] = rec["illegal"]
            out["highest"][b, s] = values_to_exp(rec["highest"])
            out["score"][b, s] = rec["score"]
            out["boards"][b, s] = values_to_exp(rec["board"]).reshape(16)
            out["terminal_boards"][b, s] = values_to_exp(rec["terminal_board"]).reshape(16)
            out["terminal_score"][b, s] = rec["terminal_score"]
    out["meta"] = np.array([seed, board_offset, n_boards, n_steps,
                            0 if max_tile is None else int(np.log2(max_tile)), int(auto_reset)], np.int64)
    out["illegal_move_reward"] = np.array([0.0 if illegal_move_reward is None else illegal_move_reward],
                                          np.float32)
    out["name"] = np.array(name)
    return out


def gen_numpy_trajectories(ref, seed, n_boards, n_steps, illegal_move_reward=None):
    """The UNMODIFIED reference with ITS OWN RNG: env i is seeded ``reset(seed=seed + i)`` (what SB3's
    make_vec_env does), i.e. np_random = Generator(PCG64(SeedSequence(seed + i))) exactly as gymnasium
    creates it.  Actions are inputs: the synthetic policy stream random_action(seed, t, i)."""
    shape = (n_boards, n_steps)
    out = dict(actions=np.zeros(shape, np.uint8), reward=np.zeros(shape, np.float32),
               terminated=np.zeros(shape, np.uint8), illegal=np.zeros(shape, np.uint8),
               highest=np.zeros(shape, np.uint8), score=np.zeros(shape, np.int32),
               boards=np.zeros(shape + (16,), np.uint8), terminal_boards=np.zeros(shape + (16,), np.uint8),
               terminal_score=np.zeros(shape, np.int32), initial_boards=np.zeros((n_boards, 16), np.uint8))
    for b in range(n_boards):
        env = ref.Game2048Env()
        if illegal_move_reward is not None:
            env.set_illegal_move_reward(illegal_move_reward)
        env.reset(seed=seed + b)
        out["initial_boards"][b] = values_to_exp(env.get_board()).reshape(16)
        for s in range(n_steps):
            a = random_action(seed, s + 1, b)
            obs, reward, terminated, truncated, info = env.step(a)
            out["actions"][b, s] = a
            out["reward"][b, s] = reward
            out["terminated"][b, s] = terminated
            out["illegal"][b, s] = info["illegal_move"]
            out["highest"][b, s] = values_to_exp(info["highest"])
            out["terminal_boards"][b, s] = values_to_exp(env.get_board()).reshape(16)
            out["terminal_score"][b, s] = env.score
            if terminated:
                env.reset()
            out["boards"][b, s] = values_to_exp(env.get_board()).reshape(16)
            out["score"][b, s] = env.score
    out["meta"] = np.array([seed, 0, n_boards, n_steps, 0, 1], np.int64)
    out["illegal_move_reward"] = np.array([0.0 if illegal_move_reward is None else illegal_move_reward], np.float32)
    out["numpy_version"] = np.array(np.__version__)
    return out


def gen_training_data_fixtures(report):
    """The reference's data format either side of the step path: training_data.export_csv text and
    training_data.augment() arrays, produced by the reference's own class (training_data.py) from
    32 rows of its fixture data/test_data.csv plus two synthetic episode ends."""
    import training_data as td_mod  # /root/reference/training_data.py (pure numpy)
    raw = np.loadtxt(os.path.join(REFERENCE, "data", "test_data.csv"), delimiter=",", skiprows=1)[:32]
    td = td_mod.training_data()
    for i, row in enumerate(raw):
        td.add(row[:16].astype(int), int(row[16]), float(row[17]), row[18:34].astype(int), done=(i in (11, 31)))
    for name, add_returns in (("transitions_ref.csv", False), ("transitions_ref_returns.csv", True)):
        path = os.path.join(HERE, name)
        td.export_csv(path, add_returns=add_returns)
        report.append(f"{name}: {os.path.getsize(path)} bytes sha256[:16]={hashlib.sha256(open(path, 'rb').read()).hexdigest()[:16]} "
                      f"(written by the reference's training_data.export_csv)")
    base = dict(x=td.get_x().copy(), action=td.get_y_digit().copy(), reward=td.get_reward().copy(),
                next_x=td.get_next_x().copy(), done=td.get_done().copy(),
                returns=td.get_discounted_return().copy())
    td.augment()
    base.update(aug_x=td.get_x(), aug_action=td.get_y_digit(), aug_next_x=td.get_next_x(),
                aug_reward=td.get_reward(), aug_done=td.get_done())
    return base


def gen_render_fixture(ref, rng):
    """render('ansi') text of the reference (game2048_env.py:113-115,156-163) for 24 states: right after
    reset (score is the int 0), during play (score is a float), and set boards with wide tiles."""
    boards, scores, score_is_int, texts = [], [], [], []

    def grab(env):
        boards.append(values_to_exp(env.get_board()).reshape(16))
        scores.append(float(env.score))
        score_is_int.append(isinstance(env.score, int))
        texts.append(env.render(mode="ansi").getvalue())

    d = RefDriver(ref, 123, board=5)
    grab(d.env)                                   # fresh env: "Score: 0"
    for t in range(40):
        legal = d.legal_actions()
        d.step(legal[t % len(legal)] if legal else 0)
        if t % 3 == 0:
            grab(d.env)
    for _ in range(8):                            # arbitrary boards up to 65536, arbitrary float scores
        env = ref.Game2048Env()
        e = rng.integers(0, 17, 16)
        e[rng.random(16) < 0.35] = 0
        env.set_board(np.where(e > 0, 1 << e, 0).reshape(4, 4))
        env.score = float(rng.integers(0, 200000)) * 4.0
        grab(env)
    env = ref.Game2048Env(render_mode="ansi")      # mode taken from render_mode (:114-115), empty board
    env.set_board(np.zeros((4, 4), dtype=int))
    boards.append(np.zeros(16, np.uint8)); scores.append(0.0); score_is_int.append(True)
    texts.append(env.render().getvalue())
    return dict(boards=np.array(boards, np.uint8), scores=np.array(scores), score_is_int=np.array(score_is_int),
                texts=np.array(texts))


def gen_render_rgb_fixture(ref, rng):
    """render('rgb_array') (game2048_env.py:116-154) drawn by the reference's OWN code.  The reference asks Pillow for
    'Arial.ttf' (:140), which this image does not have (its render raises OSError here); for the capture -- and only
    there -- ImageFont.truetype is wrapped so that 'Arial.ttf' resolves to the image's DejaVuSans.ttf at the same
    size.  Everything else (geometry, colour map, text placement through textbbox) is the reference's.  The font
    actually used is recorded; the build's renderer falls back to the same file when Arial is absent."""
    from PIL import ImageFont
    real = ImageFont.truetype

    def truetype(font=None, size=10, *a, **k):
        return real("DejaVuSans.ttf" if font == "Arial.ttf" else font, size, *a, **k)

    boards, images = [], []
    ImageFont.truetype = truetype
    try:
        # tiles up to 512 only: with the wider substitute font a four-digit label exceeds the 70-pixel cell and the
        # reference's own assert (:151) fires; the colours of 1024..4096 are pinned by value in test_host_logic.py
        exps = [np.array([1, 2, 3, 4, 5, 6, 7, 8, 9, 0, 0, 1, 9, 8, 7, 0]),         # nine colours of the map, 1-3 digits
                np.zeros(16, int), np.array([9] * 16), np.array([1] + [0] * 15)]
        for _ in range(4):
            e = rng.integers(0, 10, 16)
            e[rng.random(16) < 0.3] = 0
            exps.append(e)
        for e in exps:
            env = ref.Game2048Env(render_mode="rgb_array")
            env.set_board(np.where(e > 0, 1 << e, 0).reshape(4, 4))
            img = env.render()
            assert img.shape == (280, 280, 3) and img.dtype == np.uint8
            boards.append(e.astype(np.uint8))
            images.append(img.copy())
    finally:
        ImageFont.truetype = real
    return dict(boards=np.array(boards, np.uint8), images=np.array(images, np.uint8),
                font=np.array(os.path.basename(real("DejaVuSans.ttf", 30).path)))


def gen_canonical_table(rng, n=1024):
    """Symmetric-board canonicalisation pinned to the reference's own symmetry code: the eight
    variants of every transition are produced by training_data.hflip()/rotate() (training_data.py:257-280,
    in augment()'s order), and the canonical one is the lexicographically smallest board (row-major), lowest
    variant index on ties.  Includes boards with symmetric duplicates (ties)."""
    import training_data as td_mod
    x = random_boards(rng, n, max_exp=12)
    x[: n // 8] = 0                                     # sparse boards: many ties between symmetries
    x[: n // 8, 0] = rng.integers(1, 5, n // 8)
    x[n // 8: n // 4] = np.repeat(rng.integers(1, 6, (n // 8, 1)), 16, axis=1)   # constant boards: all 8 tie
    nx = random_boards(rng, n, max_exp=12)
    a = rng.integers(0, 4, n)
    vals = lambda e: np.where(e > 0, 1 << e.astype(np.int64), 0)  # noqa: E731
    variants = []
    for flip in (False, True):
        for k in range(4):
            td = td_mod.training_data()
            for i in range(n):
                td.add(vals(x[i]).reshape(4, 4), int(a[i]), 0.0, vals(nx[i]).reshape(4, 4), False)
            if flip:
                td.hflip()
            if k:
                td.rotate(k)
            variants.append((2 * k + int(flip), td.get_x().reshape(n, 16).copy(), td.get_y_digit().reshape(n).copy(),
                             td.get_next_x().reshape(n, 16).copy()))
    variants.sort(key=lambda v: v[0])
    canon_x, canon_a, canon_nx, sym = [], [], [], []
    for i in range(n):
        best = min(range(8), key=lambda v: (tuple(variants[v][1][i]), v))
        sym.append(best)
        canon_x.append(values_to_exp(variants[best][1][i]))
        canon_a.append(variants[best][2][i])
        canon_nx.append(values_to_exp(variants[best][3][i]))
    return dict(boards=x.astype(np.uint8), actions=a.astype(np.uint8), next_boards=nx.astype(np.uint8),
                canon_boards=np.array(canon_x, np.uint8).reshape(n, 16), canon_actions=np.array(canon_a, np.uint8),
                canon_next=np.array(canon_nx, np.uint8).reshape(n, 16), symmetry=np.array(sym, np.uint8))


def gen_rewards_table(ref, rng, n=600):
    """add_rewards_to_training_data.py:55-59 (get_reward_for_state_action): reward of env.step(action) from a
    set board, for random (board, action) pairs incl. illegal moves, with the default and a negative
    illegal_move_reward.  Computed by the reference env itself."""
    boards = random_boards(rng, n, max_exp=12)
    actions = rng.integers(0, 4, n)
    out = {}
    for tag, irw in (("default", None), ("minus1", -1.0)):
        env = ref.Game2048Env()
        if irw is not None:
            env.set_illegal_move_reward(irw)
        rewards = np.zeros(n, np.float64)
        for i in range(n):
            env.reset()
            env.set_board(np.where(boards[i] > 0, 1 << boards[i].astype(np.int64), 0).reshape(4, 4))
            _, reward, _, _, _ = env.step(int(actions[i]))
            rewards[i] = reward
        out["rewards_" + tag] = rewards
    out.update(boards=boards.astype(np.uint8), actions=actions.astype(np.uint8))
    return out


def gen_reference_test_kats(ref):
    """Re-capture, by calling the reference, the values its own unit tests pin
    (test_game2048_env.py:13-34 shift rows, :40-98 move board, :113-151 isend, :165-217 step)."""
    e = ref.Game2048Env()
    rows = [[0, 0, 0, 0], [0, 2, 0, 0], [0, 2, 0, 4], [2, 4, 8, 16], [2, 2, 8, 0], [4, 2, 2, 4],
            [2, 2, 2, 8], [2, 8, 4, 4], [2, 2, 4, 4], [2, 4, 4, 4], [4, 4, 4, 4], [0, 2, 2, 8]]
    shift_out = [e.shift(r) for r in rows]
    board = [[0, 2, 0, 4], [2, 2, 8, 0], [2, 2, 2, 8], [2, 2, 4, 4]]
    move_out = []
    for d in range(4):
        e.set_board(np.array(board))
        sc = e.move(d)
        move_out.append((sc, e.get_board().copy()))
    # repeat-move illegal + follow-on move (test_game2048_env.py:89-98)
    repeat_illegal = False
    try:
        e.move(3)
    except ref.IllegalMove:
        repeat_illegal = True
    follow_score = e.move(2)
    follow_board = e.get_board().copy()
    # step KATs
    e2 = ref.Game2048Env()
    e2.np_random = SpawnStream(e2, 0)
    e2.reset()
    e2.set_board(np.array([[0, 0, 0, 0], [0, 0, 0, 0], [2, 0, 0, 0], [2, 0, 0, 0]]))
    e2.np_random.begin_step()
    _, r1, _, _, _ = e2.step(0)
    e2.set_board(np.array([[0, 0, 0, 0], [0, 0, 0, 0], [4, 0, 0, 0], [4, 0, 0, 0]]))
    e2.np_random.begin_step()
    _, r2, _, _, _ = e2.step(0)
    score_after = e2.score
    e3 = ref.Game2048Env()
    e3.set_illegal_move_reward(-1.0)
    e3.np_random = SpawnStream(e3, 0)
    e3.reset()
    e3.set_board(np.array([[2, 4, 8, 16], [4, 8, 16, 2], [8, 16, 2, 4], [16, 2, 4, 8]]))
    e3.np_random.begin_step()
    _, r3, term3, _, info3 = e3.step(0)
    return dict(
        shift_rows=np.array(rows, np.int64), shift_out=np.array([o[0] for o in shift_out], np.int64),
        shift_score=np.array([o[1] for o in shift_out], np.int64),
        move_board=np.array(board, np.int64), move_score=np.array([m[0] for m in move_out], np.int64),
        move_out=np.array([m[1] for m in move_out], np.int64),
        repeat_illegal=np.array(repeat_illegal), follow_score=np.array(follow_score), follow_board=follow_board,
        step_rewards=np.array([r1, r2]), step_score_after=np.array(score_after),
        illegal_step=np.array([r3, float(term3), float(info3["illegal_move"])]),
    )


def gen_csv_fixture(ref):
    """The reference's own fixture data/test_data.csv (848 human transitions), as arrays, plus what
    the reference's move() computes for every row."""
    raw = np.loadtxt(os.path.join(REFERENCE, "data", "test_data.csv"), delimiter=",", skiprows=1)
    boards = raw[:, :16].astype(np.int64)
    actions = raw[:, 16].astype(np.uint8)
    rewards = raw[:, 17].astype(np.float32)
    nxt = raw[:, 18:34].astype(np.int64)
    done = raw[:, 34].astype(np.uint8)
    e = ref.Game2048Env()
    moved = np.zeros_like(boards)
    ref_score = np.zeros(len(boards), np.int64)
    for i in range(len(boards)):
        e.set_board(boards[i].reshape(4, 4).copy())
        ref_score[i] = e.move(int(actions[i]))
        moved[i] = e.get_board().reshape(16)
    assert np.array_equal(ref_score, rewards.astype(np.int64))
    e.set_max_tile(2048)
    e.set_board(nxt[-1].reshape(4, 4).copy())
    last_isend_max2048 = e.isend()
    return dict(boards=values_to_exp(boards), actions=actions, rewards=rewards, next_boards=values_to_exp(nxt),
                done=done, moved=values_to_exp(moved), last_isend_max2048=np.array(last_isend_max2048))


# ---------------------------------------------------------------------------------- validation


def validate(ref, total_steps, report):
    """C oracle + Python oracle vs the imported reference, step for step."""
    lib = load_oracle()
    n_boards = 64
    per_board = max(1, total_steps // (n_boards * 3))
    configs = [dict(seed=42, policy="random", irw=None, max_tile=None),
               dict(seed=7, policy="greedy", irw=-1.0, max_tile=None),
               dict(seed=2024, policy="greedy", irw=None, max_tile=256)]
    t0 = time.time()
    steps = 0
    episodes = 0
    for cfg in configs:
        drivers = [RefDriver(ref, cfg["seed"], b, cfg["irw"], cfg["max_tile"]) for b in range(n_boards)]
        ob = OracleBatch(n_boards, cfg["seed"])
        ob.illegal_move_reward = 0.0 if cfg["irw"] is None else cfg["irw"]
        ob.max_exp = 0 if cfg["max_tile"] is None else int(np.log2(cfg["max_tile"]))
        ob.reset()
        pys = [RefEnv(cfg["seed"], b) for b in range(n_boards)]
        for p in pys:
            p.illegal_move_reward = ob.illegal_move_reward
            p.max_tile = cfg["max_tile"]
            p.reset()
        for b in range(n_boards):
            assert np.array_equal(values_to_exp(drivers[b].env.get_board()).reshape(16), ob.boards[b])
            assert list(drivers[b].env.get_board().reshape(16)) == pys[b].M
        for s in range(per_board):
            t = s + 1
            acts = np.zeros(n_boards, np.uint8)
            for b in range(n_boards):
                a = random_action(cfg["seed"], t, b)
                assert a == lib.g2048o_random_action(cfg["seed"], t, b)
                if cfg["policy"] == "greedy":
                    a = greedy_action(drivers[b].legal_actions(), a, t)
                acts[b] = a
            ob.step(acts)
            for b in range(n_boards):
                rec = drivers[b].step(int(acts[b]))
                rw, term, ill, hi = pys[b].step(int(acts[b]))
                if term:
                    pys[b].reset()
                # reference vs C oracle
                assert rec["reward"] == ob.reward[b], (cfg, b, s)
                assert rec["terminated"] == bool(ob.terminated[b])
                assert rec["illegal"] == bool(ob.illegal[b])
                assert values_to_exp(rec["highest"]) == ob.highest[b]
                assert np.array_equal(values_to_exp(rec["board"]).reshape(16), ob.boards[b])
                assert rec["score"] == ob.score[b]
                if rec["terminated"]:
                    assert np.array_equal(values_to_exp(rec["terminal_board"]).reshape(16), ob.terminal_boards[b])
                    assert rec["terminal_score"] == ob.last_score[b]
                    episodes += 1
                # reference vs Python oracle
                assert (rw, term, ill, hi) == (rec["reward"], rec["terminated"], rec["illegal"], rec["highest"])
                assert list(rec["board"].reshape(16)) == pys[b].M
                assert rec["score"] == pys[b].score
            steps += n_boards
    dt = time.time() - t0
    report.append(f"validated {steps} env-steps ({episodes} episodes) of the imported reference against the C oracle "
                  f"and the Python oracle: all boards/rewards/terminated/illegal/highest/scores identical "
                  f"({dt:.0f} s)")


def time_reference(ref, report):
    """BASELINE config 1 for orientation: reference vs Python oracle, 1 core, 20 000 steps."""
    n = 20000
    drv = RefDriver(ref, 42)
    acts = [random_action(42, t + 1, 0) for t in range(n)]
    t0 = time.perf_counter()
    for a in acts:
        drv.step(a)
    ref_rate = n / (time.perf_counter() - t0)
    py = RefEnv(42, 0)
    py.reset()
    t0 = time.perf_counter()
    for a in acts:
        if py.step(a)[1]:
            py.reset()
    py_rate = n / (time.perf_counter() - t0)
    report.append(f"reference Game2048Env.step (injected spawn stream, auto-reset loop): {ref_rate:.0f} steps/s; "
                  f"oracle.cpu_ref.RefEnv: {py_rate:.0f} steps/s; ratio RefEnv/reference = {py_rate / ref_rate:.2f} "
                  f"(1 core, this container)")


def eval_policy_weights(rng):
    """Integer weights of the linear stand-in policy used for the evaluation-loop fixture: scores[a] =
    sum(W[a] * onehot) + 0.25 * (3 - a).  Integer sums are exact in float32 on any device and the quarter
    steps break every tie the same way everywhere, so the greedy action never depends on argmax's tie rule."""
    return rng.integers(-8, 9, size=(4, 16, 4, 4)).astype(np.int32)


def gen_eval_table(ref, rng):
    """The reference's own evaluation loop (train.py:127-165 evaluate_episode, driven as train.py:183-200
    evaluate_model does: seed = 456 + i, agent_seed = 123 + i, illegal_move_reward = -1) with a small
    deterministic policy, for epsilon 0 and 0.3; plus the CSV report_evaluation_results writes."""
    import importlib
    import tempfile
    import torch
    train = importlib.import_module("train")
    W = eval_policy_weights(rng)

    class IntPolicy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.from_numpy(W.astype(np.float32)), requires_grad=False)
            self.b = torch.nn.Parameter(torch.tensor([0.75, 0.5, 0.25, 0.0]), requires_grad=False)

        def forward(self, x):                              # x: (B, 16, 4, 4) as model.observation_to_tensor makes it
            return torch.einsum("bcyx,acyx->ba", x, self.w) + self.b

    model = IntPolicy()
    episodes = 24
    out = {"weights": W, "episodes": np.int64(episodes)}
    for tag, eps in (("eps0", 0.0), ("eps03", 0.3)):
        env = ref.Game2048Env()
        env.set_illegal_move_reward(-1.)                   # train.py:184
        rows = []
        for i in range(episodes):                          # train.py:187-200
            total_reward, moves, illegals, highest = train.evaluate_episode(model, env, eps, seed=456 + i,
                                                                            agent_seed=123 + i)
            rows.append({"total_reward": total_reward, "highest": highest, "moves": moves, "illegal_moves": illegals})
        out[f"{tag}_total_reward"] = np.array([r["total_reward"] for r in rows], np.float64)
        out[f"{tag}_highest"] = np.array([r["highest"] for r in rows], np.int64)
        out[f"{tag}_moves"] = np.array([r["moves"] for r in rows], np.int64)
        out[f"{tag}_illegal_moves"] = np.array([r["illegal_moves"] for r in rows], np.int64)
        results = {"Episodes": rows}
        cwd = os.getcwd()
        with tempfile.TemporaryDirectory() as d:           # report_evaluation_results writes into the cwd
            os.chdir(d)
            try:
                train.report_evaluation_results(results, label=tag)
                out[f"{tag}_csv"] = np.frombuffer(open(f"scores_{tag}.csv", "rb").read(), np.uint8)
            finally:
                os.chdir(cwd)
    return out


def load_npz(path):
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def content_hash(d) -> str:
    """sha256[:16] over the fixture's content: sorted keys, dtype, shape and raw bytes of every array."""
    h = hashlib.sha256()
    for k in sorted(d):
        a = np.ascontiguousarray(d[k])
        h.update(f"{k}|{a.dtype.str}|{a.shape}|".encode())
        h.update(a.tobytes())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--validate-steps", type=int, default=1_000_000)
    ap.add_argument("--only-numpy", action="store_true", help="only (re)generate the numpy-RNG trajectories")
    ap.add_argument("--only-data", action="store_true", help="only (re)generate the training_data fixtures")
    ap.add_argument("--only-round2", action="store_true", help="only (re)generate the fixtures added in round 2")
    ap.add_argument("--only-round3", action="store_true", help="only (re)generate the fixtures added in round 3")
    args = ap.parse_args()
    ref = import_reference()
    rng = np.random.default_rng(20480)
    report = []

    def save(name, d):
        # The fixture is only rewritten when its CONTENT changed (a zip archive carries timestamps, so its bytes
        # differ from run to run); the report lists a hash of the content -- keys, dtypes, shapes, array bytes --
        # which tests/test_oracle_golden.py::test_validation_report_covers_every_fixture recomputes.
        path = os.path.join(HERE, name)
        d = {k: np.asarray(v) for k, v in d.items()}
        h = content_hash(d)
        if not (os.path.exists(path) and content_hash(load_npz(path)) == h):
            np.savez_compressed(path, **d)
        report.append(f"{name}: {len(d)} arrays, {sum(v.nbytes for v in d.values())} bytes of data, content sha256[:16]={h}")

    if args.only_data:
        save("training_data_fixture.npz", gen_training_data_fixtures(report))
        print("\n".join(report))
        return
    if args.only_round2:
        save("traj_corner_deep.npz", gen_trajectories(ref, "corner_deep", 31, 48, 1024, "corner"))
        save("rewards_table.npz", gen_rewards_table(ref, np.random.default_rng(9)))
        save("render_ansi.npz", gen_render_fixture(ref, np.random.default_rng(7)))
        save("canonical_table.npz", gen_canonical_table(np.random.default_rng(8)))
        save("eval_table.npz", gen_eval_table(ref, np.random.default_rng(10)))
        print("\n".join(report))
        return
    if args.only_round3:
        save("render_rgb.npz", gen_render_rgb_fixture(ref, np.random.default_rng(11)))
        print("\n".join(report))
        return
    save("training_data_fixture.npz", gen_training_data_fixtures(report))
    save("traj_numpy_seed42.npz", gen_numpy_trajectories(ref, 42, 48, 256))
    save("traj_numpy_seed7_irw.npz", gen_numpy_trajectories(ref, 7, 16, 512, illegal_move_reward=-1.0))
    if args.only_numpy:
        print("\n".join(report))
        return
    save("shift_exhaustive.npz", gen_shift_table(ref))
    save("move_table.npz", gen_move_table(ref, rng))
    save("isend_table.npz", gen_isend_table(ref, rng))
    save("stack_table.npz", gen_stack_table(ref, rng))
    save("reference_test_kats.npz", gen_reference_test_kats(ref))
    save("test_data_csv.npz", gen_csv_fixture(ref))
    save("traj_random_seed42.npz", gen_trajectories(ref, "random", 42, 64, 256, "random"))
    save("traj_random_offset.npz", gen_trajectories(ref, "random_offset", 42, 8, 128, "random",
                                                    board_offset=(1 << 20) - 4))
    save("traj_greedy_irw.npz", gen_trajectories(ref, "greedy_irw", 7, 16, 1024, "greedy",
                                                 illegal_move_reward=-1.0))
    save("traj_greedy_max256.npz", gen_trajectories(ref, "greedy_max256", 2024, 16, 768, "greedy",
                                                    max_tile=256))
    save("traj_noautoreset.npz", gen_trajectories(ref, "noautoreset", 5, 16, 96, "random", auto_reset=False))
    save("traj_corner_deep.npz", gen_trajectories(ref, "corner_deep", 31, 48, 1024, "corner"))
    save("rewards_table.npz", gen_rewards_table(ref, np.random.default_rng(9)))
    save("render_ansi.npz", gen_render_fixture(ref, np.random.default_rng(7)))
    save("canonical_table.npz", gen_canonical_table(np.random.default_rng(8)))
    save("eval_table.npz", gen_eval_table(ref, np.random.default_rng(10)))
    save("render_rgb.npz", gen_render_rgb_fixture(ref, np.random.default_rng(11)))
    validate(ref, args.validate_steps, report)
    time_reference(ref, report)
    import datetime
    oracle_c = os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "g2048_oracle.c")
    report.append(f"generated {datetime.date.today().isoformat()} with numpy {np.__version__}; oracle/g2048_oracle.c "
                  f"sha256[:16]={hashlib.sha256(open(oracle_c, 'rb').read()).hexdigest()[:16]}")
    with open(os.path.join(HERE, "VALIDATION.txt"), "w") as f:
        f.write("\n".join(report) + "\n")
    print("\n".join(report))


if __name__ == "__main__":
    main()
