"""Host-side rendering of one board: the ``render()`` surface of the reference
(game2048_env.py:113-163).  Not on the hot path; the board is copied to the host first."""
from __future__ import annotations

import sys
from io import StringIO

import numpy as np

GRID_SIZE = 70  # game2048_env.py:57


def tile_colour(value: int):
    """Colour ramp of the reference's tile map (game2048_env.py:120-133): red -> green over
    2..512 in steps of 32, then green -> blue-green over 512..4096."""
    k = int(value).bit_length() - 1  # 2 -> 1 ... 4096 -> 12
    if not 1 <= k <= 12:
        raise KeyError(value)  # the reference's dict lookup fails the same way above 4096
    if k <= 9:
        up = 32 * (k - 1)
        return (min(255, 256 - up), min(255, up), 0)
    up = 32 * (k - 9)
    return (0, min(255, 256 - up), up)


def _font(size: int):
    from PIL import ImageFont
    try:
        return ImageFont.truetype("Arial.ttf", size)  # game2048_env.py:140
    except OSError:
        for name in ("DejaVuSans.ttf", "LiberationSans-Regular.ttf"):
            try:
                return ImageFont.truetype(name, size)
            except OSError:
                continue
        return ImageFont.load_default()


def render_rgb(values: np.ndarray) -> np.ndarray:
    """(280, 280, 3) uint8 image of the board (game2048_env.py:116-154)."""
    from PIL import Image, ImageDraw
    g = GRID_SIZE
    img = Image.new("RGB", (g * 4, g * 4))
    draw = ImageDraw.Draw(img)
    draw.rectangle([0, 0, 4 * g, 4 * g], (128, 128, 128))
    font = _font(30)
    for y in range(4):
        for x in range(4):
            v = int(values[y, x])
            if not v:
                continue
            draw.rectangle([x * g, y * g, (x + 1) * g, (y + 1) * g], tile_colour(v))
            box = draw.textbbox((0, 0), str(v), font=font)
            tw, th = box[2] - box[0], box[3] - box[1]
            draw.text((x * g + (g - tw) // 2, y * g + (g - th) // 2), str(v), font=font, fill=(255, 255, 255))
    return np.asarray(img)


def render_board(values: np.ndarray, score, mode: str):
    """``values``: int64 (4,4) tile values.  Modes as game2048_env.py:35,113-163."""
    values = np.asarray(values).reshape(4, 4)
    if mode == "rgb_array":
        return render_rgb(values)
    out = StringIO() if mode == "ansi" else sys.stdout
    out.write("Score: {}\nHighest: {}\n{}\n".format(score, values.max(), values))  # :156-162
    return out
