"""Bake the Pong score-band atlas (build-owned asset, generated once in the build container).

    python competitive_rl_amd/assets/gen_score_atlas.py

What it restates: ``Scoreboard.draw`` (reference pong/base_pong_env.py:474-487) --
``pygame.font.Font("freesansbold.ttf", 20).render("Score = %d : %d", True, BLACK)``
alpha-blitted at (20, 8) onto the white top band.  pygame 1.9.6 / SDL_ttf / FreeType
are third-party and absent, so the glyph pixels are rendered here with PIL+FreeType
from the FreeSansBold.ttf the reference ships (read in place, not copied) and
composited with SDL 1.2's blend ``d + (((s - d) * a) >> 8)``.  Pixel parity with real
pygame is UNPINNED; the oracle and the HIP kernels share this atlas, so
they are bit-identical to each other.

Output: ``pong_score_atlas.npz`` with ``atlas`` u8 [22, 22, 34, 160]: for every
(score_left, score_right) the gray value of rows 0..33 of view 0.
"""
import os

import numpy as np
from PIL import Image, ImageDraw, ImageFont

FONT = "/root/reference/competitive_rl/pong/FreeSansBold.ttf"
HERE = os.path.dirname(os.path.abspath(__file__))
TOP, W = 34, 160


def main():
    font = ImageFont.truetype(FONT, 20, layout_engine=ImageFont.Layout.BASIC)
    atlas = np.full((22, 22, TOP, W), 255, np.uint8)
    for sl in range(22):
        for sr in range(22):
            cov = Image.new("L", (W, 64), 0)
            # anchor "la": x = left, y = ascender line == top of SDL_ttf's text surface
            ImageDraw.Draw(cov).text((0, 0), "Score = %d : %d" % (sl, sr), fill=255, font=font, anchor="la")
            a = np.asarray(cov, dtype=np.int32)  # coverage 0..255 == per-pixel alpha
            band = np.full((TOP, W), 255, np.int32)
            h = min(TOP - 8, a.shape[0])
            w = W - 20
            d = band[8:8 + h, 20:20 + w]
            band[8:8 + h, 20:20 + w] = d + (((0 - d) * a[:h, :w]) >> 8)
            assert not a[h:, :].any(), "text taller than the white band"
            atlas[sl, sr] = band.astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, "pong_score_atlas.npz"), atlas=atlas)
    ink = (atlas < 255)
    rows = np.where(ink.any(axis=(0, 1, 3)))[0]
    cols = np.where(ink.any(axis=(0, 1, 2)))[0]
    print("ink rows", rows.min(), rows.max(), "cols", cols.min(), cols.max(),
          "levels", len(np.unique(atlas)))


if __name__ == "__main__":
    main()
