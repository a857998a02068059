#!/usr/bin/env python3
"""Sizing of "touched-only AdamW with exact catch-up for the sparse grid"  - CPU only, numpy.

torch.optim.AdamW is dense: every step moves every cell's (p, m, v), gradient or not (the moments decay, the momentum keeps
moving p).  An EXACT lazy variant may skip a cell's update at step t only if nobody reads the cell before it is caught up.
Two readers exist: the scatter's flush of step t (cells that received a gradient) and the FORWARD of step t + 1, which reads
p of the 3 x 3 patch of every pixel of batch t + 1.  So the set that must be processed at step t is
    touched(t)  U  read(t + 1)
at the granularity memory moves in: 64-B HBM sectors / 128-B L2 lines of the [T][X][Y][F] tensor (y fastest), not cells.
This script draws two batches with the reference's sampler (dataio.py:104-120: N = 1 245 184 i.i.d. (t, pixel) draws over
1920 x 1080 x 600), maps them to sparse cells as sparsegrid.py:43-69 does and counts.  Output: profiles/r06_size_touched_adamw.txt"""
import numpy as np

rng = np.random.default_rng(0)
N, T, X, Y, H, W = 1245184, 600, 300, 300, 1080, 1920


def cells():
    t = rng.integers(0, T, N)
    pi = rng.integers(0, H * W, N)
    row, col = pi // W, pi % W
    xi = np.clip((np.float32(X - 1) * (row / np.float32(H - 1)).astype(np.float32) + np.float32(0.5)).astype(np.int64), 0, X - 1)
    yi = np.clip((np.float32(Y - 1) * (col / np.float32(W - 1)).astype(np.float32) + np.float32(0.5)).astype(np.int64), 0, Y - 1)
    out = [(t * X + np.clip(xi + dx, 0, X - 1)) * Y + np.clip(yi + dy, 0, Y - 1) for dx in (-1, 0, 1) for dy in (-1, 0, 1)]
    return np.unique(np.concatenate(out))


a, b = cells(), cells()
tot = T * X * Y
print(f"sparse cells that receive a gradient per step: {len(a) / 1e6:.2f} M of {tot / 1e6:.1f} M = {len(a) / tot:.3f}")
rows = np.unique(a // Y)
print(f"(t, x) rows of 300 cells with at least one touched cell: {len(rows)} of {T * X} = {len(rows) / (T * X):.4f}   <- a per-row stamp saves nothing")
for F, name in ((2, "config_nvp_s, F = 2"), (4, "config_nvp_l, F = 4")):
    for seg in (32, 64, 128, 256):
        cps = seg // (4 * F)
        sa, sb = np.unique(a // cps), np.unique(b // cps)
        nseg = (tot + cps - 1) // cps
        u = np.union1d(sa, sb)
        print(f"{name}: {seg:3d}-B segments ({cps:2d} cells): touched(t) {len(sa) / nseg:.3f}   touched(t) U read(t+1) {len(u) / nseg:.3f}"
              f"   -> (p, m, v) bytes saved at best {(1 - len(u) / nseg) * 24 * tot * F / 1e9:.2f} of {24 * tot * F / 1e9:.2f} GB")
