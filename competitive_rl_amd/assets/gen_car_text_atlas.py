"""Bake the CarRacing reward read-out (build-owned asset, generated once in the build container).

    python competitive_rl_amd/assets/gen_car_text_atlas.py

What it restates: ``draw_text(screen, "%05.0f" % reward, W/100, H - H/20, fonts[5])`` with
``font.render(text, False, (255, 255, 255))`` (reference car_racing/pygame_rendering.py:16-18,
car_racing_multi_players.py:669-670, fonts/COMIC.TTF at 5 px, NOT antialiased).  pygame/SDL_ttf are
absent, so every string "%05.0f" % r for r in [-999, 2000] is rendered here with PIL+FreeType in
1-bit mode from the COMIC.TTF the reference ships (read in place).  Pixel parity with SDL_ttf's mono
renderer is UNPINNED.

Output: ``car_reward_text.npz``: ``bits`` u32 [3001, 10] -- row r of string k (k = 3000 is "-0000"), bit c = pixel (c, r)
lit, relative to the blit position (0, 91); ``r_min`` = -999.
"""
import os

import numpy as np
from PIL import Image, ImageDraw, ImageFont

FONT = "/root/reference/competitive_rl/car_racing/fonts/COMIC.TTF"
HERE = os.path.dirname(os.path.abspath(__file__))
R_MIN, COUNT, ROWS = -999, 3000, 10


def main():
    font = ImageFont.truetype(FONT, 5, layout_engine=ImageFont.Layout.BASIC)
    bits = np.zeros((COUNT + 1, ROWS), np.uint32)
    wmax = hmax = 0
    for k in range(COUNT + 1):
        text = "%05.0f" % float(R_MIN + k) if k < COUNT else "-0000"  # last entry: rewards in (-0.5, 0)
        im = Image.new("1", (32, 16), 0)
        ImageDraw.Draw(im).text((0, 0), text, fill=1, font=font, anchor="la")
        a = np.asarray(im, dtype=np.uint8)
        ys, xs = np.nonzero(a)
        if len(ys):
            hmax, wmax = max(hmax, ys.max() + 1), max(wmax, xs.max() + 1)
        assert not a[ROWS:].any() and not a[:, 32:].any()
        for r in range(ROWS):
            bits[k, r] = sum(int(v) << c for c, v in enumerate(a[r, :32]))
    np.savez_compressed(os.path.join(HERE, "car_reward_text.npz"), bits=bits, r_min=R_MIN)
    print("strings", COUNT, "ink box", wmax, "x", hmax)
    for k in (0, 999, 999 + 12, 999 + 1000):
        print("%05.0f" % float(R_MIN + k))
        for r in range(ROWS):
            print("".join("#" if (int(bits[k, r]) >> c) & 1 else "." for c in range(24)))


if __name__ == "__main__":
    main()
