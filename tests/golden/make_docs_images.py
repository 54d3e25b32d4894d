"""Turns two figures of the reference's docs - actual outputs of the reference, published as images - into fixtures.

1. docs/tutorials/minimal-env/render.png -> tests/golden/docs_render_columns.npy: the 64 pixel columns of the
   reference's own rendered strip.

That image is the reference's `r.screen` for `toys.box()`, one agent at (3, 3) heading 0, default Core (64 rays, fov
130), shown through `plotting.plot_images` (= gamma_encode, docs/tutorials/minimal-env/index.rst:106-120). Its textures
and light intensity came from an unseeded RNG, so brightness cannot be reproduced - but which wall every ray lands on,
and the hue each wall is drawn in, can. Only pixel data is stored (64 x 3 mean colours, 0-255).

2. docs/tutorials/geometry/walls-masks.png -> tests/golden/docs_geometry_masks.npy: the (36, 36) array
   `geometry.masks(walls, [corners])` returned for the tutorial's 5 m box (index.rst:9-60), read off the figure cell by
   cell (-1 wall, 0 free, 1 room; row 0 is the top row, as `geometry.centers` has it).

Run from the repo root:  python tests/golden/make_docs_images.py
"""
import os

import numpy as np
from PIL import Image

SRC = '/root/reference/docs/tutorials/minimal-env/render.png'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'docs_render_columns.npy')


MASKS_SRC = '/root/reference/docs/tutorials/geometry/walls-masks.png'
MASKS_OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'docs_geometry_masks.npy')


def masks():
    im = np.asarray(Image.open(MASKS_SRC).convert('RGB')).astype(int)
    row, col = im[400], im[:, 500]
    xs, ys = np.where((row < 30).all(1))[0], np.where((col < 30).all(1))[0]
    x0, x1 = xs.min(), xs[xs < 900].max()                       # the axes frame: x from 0 to 7.2 m ...
    y0, y1 = ys.min(), ys.max()                                 # ... y from 7.2 m (top) down to 0
    palette = {(31, 119, 180): -1, (140, 86, 75): 0, (158, 218, 229): 1}     # the figure's three colours
    keys, vals = np.array(list(palette)), np.array(list(palette.values()))
    n = 36
    grid = np.zeros((n, n), np.int16)
    for i in range(n):
        for j in range(n):
            px, py = int(round(x0 + (x1 - x0)*(j + .5)/n)), int(round(y0 + (y1 - y0)*(i + .5)/n))
            c = im[py - 2:py + 3, px - 2:px + 3].reshape(-1, 3).mean(0)
            grid[i, j] = vals[np.argmin(((keys - c)**2).sum(1))]
    np.save(MASKS_OUT, grid)
    print(f'wrote {MASKS_OUT}')


def main():
    masks()
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
