"""Synthetic workloads for the Poisson Gauss-Seidel path (host side, numpy only).

These are the inputs SURVEY §8d defines for the five BASELINE.json configs: the closed-form
5-point Poisson system of ``SolveChannel`` (reference: project/src/PhotoMontage/
PhotoMontage.cpp:541-597) with ``b = A·x_true``, and the photomontage-style irregular
mask whose restricted Laplacian exercises the general CSR path.

Nothing here touches the GPU or the oracle; tests and bench.py share these generators so
the CPU checker and the HIP path always see identical bytes.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def rng(seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.MT19937(seed))


def x_true(n: int, seed: int = 1234) -> np.ndarray:
    """x_true ~ U[0,255) (pixel-valued unknowns)."""
    return rng(seed).uniform(0.0, 255.0, n)


# ---- structured Poisson system (closed form) --------------------------------
def poisson_masks(W: int, H: int):
    """Boolean H×W arrays (has_up, has_left, has_here) and the float diagonal."""
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    cell = lambda x, y: (x >= 0) & (y >= 0) & (x < W - 1) & (y < H - 1)
    up, left, here = cell(xx, yy - 1), cell(xx - 1, yy), cell(xx, yy)
    diag = up.astype(np.float64) + left + 2.0 * here
    diag[0, 0] += 1.0
    return up, left, here, diag


def poisson_apply(W: int, H: int, v: np.ndarray) -> np.ndarray:
    """A·v for the reference Poisson matrix, in applyToVector's accumulation order
    (sparse-matrix.h:382-393: storage order = up, left, diagonal, right, down)."""
    up, left, here, diag = poisson_masks(W, H)
    g = np.asarray(v, dtype=np.float64).reshape(H, W)
    acc = np.zeros((H, W))
    sh = np.zeros((H, W))
    sh[1:, :] = g[:-1, :]
    acc = np.where(up, acc + (-1.0 * sh), acc)
    sh = np.zeros((H, W))
    sh[:, 1:] = g[:, :-1]
    acc = np.where(left, acc + (-1.0 * sh), acc)
    acc = np.where(diag != 0, acc + diag * g, acc)
    sh = np.zeros((H, W))
    sh[:, :-1] = g[:, 1:]
    acc = np.where(here, acc + (-1.0 * sh), acc)
    sh = np.zeros((H, W))
    sh[:-1, :] = g[1:, :]
    acc = np.where(here, acc + (-1.0 * sh), acc)
    return acc.ravel()


def poisson_system(W: int, H: int, seed: int = 1234) -> Tuple[np.ndarray, np.ndarray]:
    """(b, x_true) with b = A·x_true for the W×H reference Poisson matrix."""
    xt = x_true(W * H, seed)
    return poisson_apply(W, H, xt), xt


def poisson_csr(W: int, H: int):
    """Compressed CSR (values f64, col int32, row_offset int32[n+1]) of the reference
    Poisson matrix, built vectorised (what Eigen hands to ConvertFromEigen, utils.cc:5-15)."""
    up, left, here, diag = poisson_masks(W, H)
    n = W * H
    idx = np.arange(n, dtype=np.int64).reshape(H, W)
    has = np.stack([up, left, diag != 0, here, here], axis=-1).reshape(n, 5)
    cols = np.stack([idx - W, idx - 1, idx, idx + 1, idx + W], axis=-1).reshape(n, 5)
    vals = np.stack([-np.ones((H, W)), -np.ones((H, W)), diag, -np.ones((H, W)), -np.ones((H, W))],
                    axis=-1).reshape(n, 5)
    counts = has.sum(axis=1)
    rowp = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(counts, out=rowp[1:])
    return vals[has].astype(np.float64), cols[has].astype(np.int32), rowp.astype(np.int32)


# ---- irregular mask (config 5) ------------------------------------------------------------
def disc_mask(W: int, H: int, seed: int = 4321, n_discs: int = 64, rmin: float = 200.0,
              rmax: float = 900.0, brush: int = 16) -> np.ndarray:
    """Union of seeded discs plus a ``brush``-px wide polyline trail (photomontage-style
    stroke region).  Radii are given for an 8192-px canvas and scale with min(W,H)."""
    g = rng(seed)
    s = min(W, H) / 8192.0
    mask = np.zeros((H, W), dtype=bool)
    yy = np.arange(H)[:, None]
    xx = np.arange(W)[None, :]
    cx = g.uniform(0, W, n_discs)
    cy = g.uniform(0, H, n_discs)
    rr = g.uniform(rmin, rmax, n_discs) * s
    for k in range(n_discs):
        y0, y1 = max(0, int(cy[k] - rr[k]) - 1), min(H, int(cy[k] + rr[k]) + 2)
        x0, x1 = max(0, int(cx[k] - rr[k]) - 1), min(W, int(cx[k] + rr[k]) + 2)
        if y0 >= y1 or x0 >= x1:
            continue
        sub = (yy[y0:y1] - cy[k]) ** 2 + (xx[:, x0:x1] - cx[k]) ** 2 <= rr[k] ** 2
        mask[y0:y1, x0:x1] |= sub
    # brush trail: random polyline, stamped with a square brush
    pts = np.stack([g.uniform(0, W, 9), g.uniform(0, H, 9)], axis=1)
    half = max(1, int(round(brush * max(s, 1.0 / 64) / 2)))
    for a, b in zip(pts[:-1], pts[1:]):
        steps = int(max(abs(b[0] - a[0]), abs(b[1] - a[1]))) + 1
        t = np.linspace(0.0, 1.0, steps)
        px = np.clip((a[0] + t * (b[0] - a[0])).astype(np.int64), 0, W - 1)
        py = np.clip((a[1] + t * (b[1] - a[1])).astype(np.int64), 0, H - 1)
        for dy in range(-half, half + 1):
            for dx in range(-half, half + 1):
                mask[np.clip(py + dy, 0, H - 1), np.clip(px + dx, 0, W - 1)] = True
    return mask


def masked_laplacian_csr(mask: np.ndarray):
    """5-point Laplacian restricted to ``mask`` with Dirichlet boundary: unknowns are the
    masked pixels in raster order, diagonal 4, -1 to each in-mask 4-neighbour.
    Returns (values, col, row_offset[n+1], colour, ys, xs); colour = (x+y)&1."""
    H, W = mask.shape
    ys, xs = np.nonzero(mask)
    n = len(ys)
    ident = -np.ones((H, W), dtype=np.int64)
    ident[ys, xs] = np.arange(n)
    pad = -np.ones((H + 2, W + 2), dtype=np.int64)
    pad[1:-1, 1:-1] = ident
    upn = pad[ys, xs + 1]
    leftn = pad[ys + 1, xs]
    rightn = pad[ys + 1, xs + 2]
    downn = pad[ys + 2, xs + 1]
    me = np.arange(n)
    cols = np.stack([upn, leftn, me, rightn, downn], axis=1)
    vals = np.tile(np.array([-1.0, -1.0, 4.0, -1.0, -1.0]), (n, 1))
    has = cols >= 0
    rowp = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(has.sum(axis=1), out=rowp[1:])
    colour = ((xs + ys) & 1).astype(np.int32)
    return (vals[has].astype(np.float64), cols[has].astype(np.int32), rowp.astype(np.int32),
            colour, ys.astype(np.int32), xs.astype(np.int32))


def csr_apply(values, cols, rowp, v) -> np.ndarray:
    """Row-wise left-to-right CSR product (applyToVector order) for short rows."""
    n = len(rowp) - 1
    out = np.zeros(n)
    cnt = np.diff(rowp)
    for k in range(int(cnt.max()) if n else 0):
        sel = cnt > k
        idx = rowp[:-1][sel] + k
        out[sel] = out[sel] + values[idx] * v[cols[idx]]
    return out
