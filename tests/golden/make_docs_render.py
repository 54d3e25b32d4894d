"""Extracts the 64 pixel columns of the reference's own rendered strip, docs/tutorials/minimal-env/render.png, into
tests/golden/docs_render_columns.npy.

That image is the reference's `r.screen` for `toys.box()`, one agent at (3, 3) heading 0, default Core (64 rays, fov
130), shown through `plotting.plot_images` (= gamma_encode, docs/tutorials/minimal-env/index.rst:106-120). Its textures
and light intensity came from an unseeded RNG, so brightness cannot be reproduced - but which wall every ray lands on,
and the hue each wall is drawn in, can. Only pixel data is stored (64 x 3 mean colours, 0-255).

Run from the repo root:  python tests/golden/make_docs_render.py
"""
import os

import numpy as np
from PIL import Image

SRC = '/root/reference/docs/tutorials/minimal-env/render.png'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'docs_render_columns.npy')


def main():
    im = np.asarray(Image.open(SRC).convert('RGB')).astype(float)
    row = im[80]
    black = np.where((row < 5).all(1))[0]                       # the axes frame
    a, b = black[(black > 30) & (black < 60)].max() + 1, black[black > 900].min() - 1
    assert abs((b - a + 1)/64 - 14.5) < .1, 'expected 64 columns of ~14.5 px'
    cols = []
    for k in range(64):
        c = int(a + (b - a + 1)*(k + .5)/64)
        cols.append(im[60:100, c - 2:c + 3].reshape(-1, 3).mean(0))
    np.save(OUT, np.array(cols, dtype=np.float32))
    print(f'wrote {OUT}')


if __name__ == '__main__':
    main()
