"""Synthetic workload with the shape of lab8's panorama blend (reference: labs/lab8/src/OpenCVHW1/hw8_pa.cc).

In `stitchImages` (hw8_pa.cc:501-820) every image is warped onto one canvas, the gradient fields of the
warped images are merged span by span under eroded footprint masks (`MergeImage2<float>` :338-383,
`MergeImage<uchar>` :385-440, `MaskImage` :443-465), the gradients along the seam of the union mask are
recomputed from the merged colours (`EnforceGradientBound` :468-498) and `SolveChannel` (:902-986) solves
the Poisson system of the merged field from the merged colours as start vector (:808-810, 50 iterations).

Feature matching and the homographies are out of scope and OpenCV is absent, so this
module generates what they would hand to the merge — footprints of two overlapping warped images, their
eroded masks, the warped colours — and restates the merge itself in numpy, function by function, so that
the solver path gets gradient fields and masks of the reference's shape:

  * full canvas: SolveChannel's matrix (the structured grid path) with the merged field's right-hand side;
  * union region: the 5-point Laplacian restricted to the final union mask (`mask` after :788) with the
    merged colours outside it as Dirichlet values — a second irregular-region workload for the general
    CSR path next to BASELINE configs[4] (it is recognised as a raster region, coursecomputationalphotography_amd/
    csrc/ccp_csr.hip).

Parity note: these restatements are NOT pinned by the reference (hw8_pa.cc needs OpenCV); the solvers they
feed are — tests/golden/lab8_*.npz holds the compiled reference header's results on the small instance.
The reference's row loops run past the end of a row when a mask reaches the last column (`while (*sip == 0
&& *tgp != 0)` has no bound, :364); the generator keeps every mask clear of the canvas border, where the
restatement and the reference agree.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from . import synth


# ---- inputs the out-of-scope front half would produce -------------------------------------------------
def _quad_mask(W: int, H: int, quad: np.ndarray) -> np.ndarray:
    """Pixels inside a convex quadrilateral (4 x 2 corner array, counter-clockwise in image coordinates)."""
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    inside = np.ones((H, W), dtype=bool)
    for k in range(4):
        (x0, y0), (x1, y1) = quad[k], quad[(k + 1) % 4]
        inside &= (x1 - x0) * (yy - y0) - (y1 - y0) * (xx - x0) >= 0
    return inside


def erode_cross(mask: np.ndarray) -> np.ndarray:
    """cv::erode with getStructuringElement(MORPH_CROSS, 3x3) on a 0/255 mask (hw8_pa.cc:718-722)."""
    m = mask != 0
    p = np.pad(m, 1, constant_values=True)         # OpenCV's default erosion border does not erode
    out = p[1:-1, 1:-1] & p[:-2, 1:-1] & p[2:, 1:-1] & p[1:-1, :-2] & p[1:-1, 2:]
    return out.astype(np.uint8) * 255


def smooth_image(W: int, H: int, seed: int) -> np.ndarray:
    """A smooth, textured H x W x 3 u8 image (low-frequency cosines + fine noise)."""
    g = synth.rng(seed)
    xs, ys = np.arange(W), np.arange(H)
    img = np.zeros((H, W, 3), dtype=np.uint8)
    for ch in range(3):
        acc = np.full((H, W), 128.0)
        for _ in range(4):
            fx, fy, ph = g.uniform(0.5, 3.0) / W, g.uniform(0.5, 3.0) / H, g.uniform(0, 6.28)
            ax, by = 6.283 * fx * xs + ph, 6.283 * fy * ys
            # cos(ax + by) by the addition theorem: two outer products instead of H*W cosines
            acc += g.uniform(10, 40) * (np.outer(np.cos(by), np.cos(ax)) - np.outer(np.sin(by), np.sin(ax)))
        acc += g.uniform(-6, 6, (H, W))
        img[..., ch] = np.clip(acc, 0, 255).astype(np.uint8)
    return img


def inputs(W: int, H: int, seed: int = 8) -> Dict[str, np.ndarray]:
    """Two warped images on one W x H canvas: footprints (slightly trapezoid quads overlapping in the middle
    third), warped colours (zero outside the footprint, as BORDER_TRANSPARENT on a zeroed canvas leaves them),
    and the eroded masks of the second image: erode_mask2 = 2 cross erosions, erode_mask = 4 (hw8_pa.cc:718-722)."""
    g = synth.rng(seed)
    j = lambda s: float(g.uniform(-s, s))
    m = max(4.0, 0.04 * min(W, H))
    quad0 = np.array([[m + j(2), m + j(2)], [0.60 * W + j(m), 1.5 * m + j(2)], [0.58 * W + j(m), H - 1.5 * m + j(2)],
                      [1.5 * m + j(2), H - m + j(2)]])
    quad1 = np.array([[0.40 * W + j(m), 1.3 * m + j(2)], [W - m + j(2), m + j(2)], [W - 1.4 * m + j(2), H - m + j(2)],
                      [0.43 * W + j(m), H - 1.6 * m + j(2)]])
    foot0, foot1 = _quad_mask(W, H, quad0), _quad_mask(W, H, quad1)
    scene = smooth_image(W, H, seed + 1)
    img0 = scene * foot0[..., None]
    shade = np.array([1.08, 0.95, 1.03], dtype=np.float32)         # an exposure difference the blend has to hide
    img1 = np.clip(scene.astype(np.float32) * shade + 9.0, 0, 255).astype(np.uint8) * foot1[..., None]
    little = foot1.astype(np.uint8) * 255
    e2 = erode_cross(erode_cross(little))
    e4 = erode_cross(erode_cross(e2))
    return {"img0": img0, "img1": img1, "mask0": foot0.astype(np.uint8) * 255, "erode_mask2": e2, "erode_mask": e4}


# ---- the merge, restated (hw8_pa.cc line references in every function) ---------------------------------
def gradients(img: np.ndarray):
    """Gradients::Gradients(const Mat&) (:606-631) with GradientAt (:318-326): forward differences of the u8
    image as float32 for y < H-1, x < W-1; the reference leaves the last row/column uninitialised — 0 here."""
    v = img.astype(np.int32)
    gx = np.zeros(img.shape, dtype=np.float32)
    gy = np.zeros(img.shape, dtype=np.float32)
    gx[:-1, :-1] = (v[:-1, 1:] - v[:-1, :-1]).astype(np.float32)
    gy[:-1, :-1] = (v[1:, :-1] - v[:-1, :-1]).astype(np.float32)
    return gx, gy


def mask_image(src: np.ndarray, mask: np.ndarray) -> np.ndarray:
    """MaskImage (:443-465): colours outside the mask become 0."""
    return src * (mask != 0)[..., None].astype(src.dtype)


def merge_image2(target: np.ndarray, src: np.ndarray, target_mask: np.ndarray, src_outer: np.ndarray, src_inner: np.ndarray) -> None:
    """MergeImage2<T> (:338-383), in place on `target`, row by row: skip to the first pixel of the source's
    outer mask; then skip on while the inner mask is 0 and the target is already covered; copy from there to
    the end of the outer-mask run."""
    H, W = target_mask.shape

    def first(cond, start):                                          # first index >= start where cond holds, W if none
        hit = np.flatnonzero(cond[start:])
        return start + int(hit[0]) if len(hit) else W

    for i in range(H):
        so, si, tg = src_outer[i], src_inner[i], target_mask[i]
        k = first(so != 0, 0)                                        # :351-359 skip what is not the source image
        k = first(~((si == 0) & (tg != 0)), k)                       # :363-371 (bounded here, see the module note)
        end = first(so == 0, k)                                      # :376-381 to the end of the outer-mask run
        target[i, k:end] = src[i, k:end]                             # :383


def merge_image(target: np.ndarray, src: np.ndarray, target_mask: np.ndarray, src_mask: np.ndarray, skip: float) -> None:
    """MergeImage<T, channel> (:385-440), in place: skip to the first pixel of the source mask; if the target
    is covered there, skip `skip` more pixels; copy to the end of the source-mask run."""
    H, W = target_mask.shape

    def first(cond, start):
        hit = np.flatnonzero(cond[start:])
        return start + int(hit[0]) if len(hit) else W

    for i in range(H):
        sm, tg = src_mask[i], target_mask[i]
        k = first(sm != 0, 0)                                        # :397-404
        if k < W and tg[k] != 0 and skip > 0:                        # :409-429: `skip` pixels, unconditionally
            k = min(W, k + int(np.ceil(skip)))
        end = first(sm == 0, k)                                      # :433-438
        target[i, k:end] = src[i, k:end]                             # :440


def enforce_gradient_bound(dx: np.ndarray, dy: np.ndarray, src: np.ndarray, mask: np.ndarray) -> None:
    """EnforceGradientBound (:468-498), in place: wherever `mask` (the one-pixel rim of the union) is set, the
    gradients at (x, y-1), (x, y) and (x, y+1) are recomputed from the merged colours (GradientAt)."""
    H, W = mask.shape
    v = src.astype(np.int32)
    ys, xs = np.nonzero(mask)
    for off in (0, -1, 1):                                           # :478-480 (every assignment is a pure function of src)
        yy = ys + off
        ok = (yy >= 0) & (yy < H - 1) & (xs < W - 1)
        y, x = yy[ok], xs[ok]
        dx[y, x] = (v[y, x + 1] - v[y, x]).astype(np.float32)
        dy[y, x] = (v[y + 1, x] - v[y, x]).astype(np.float32)


def merge(inp: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """The merge loop of stitchImages for two images (:749-795): returns the merged gradient field (dx, dy,
    float32 H x W x 3), the merged colours `raw` (u8) and the union mask."""
    img0, img1 = inp["img0"], inp["img1"]
    mask = inp["mask0"].copy()
    raw = img0.copy()
    dx, dy = gradients(img0)                                         # :756-759
    e2, e4 = inp["erode_mask2"], inp["erode_mask"]
    gx1, gy1 = gradients(mask_image(img1, e2))                       # :771-773
    merge_image2(dx, gx1, mask, e2, e4)                              # :777
    merge_image2(dy, gy1, mask, e2, e4)                              # :778
    merge_image(raw, img1, mask, e4, 1)                              # :782
    m3 = mask[..., None].copy()
    merge_image(m3, e4[..., None], mask, e4, 0)                      # :787 MergeImage<uchar,1>(mask, erode_mask, mask, erode_mask, 0)
    mask = m3[..., 0]
    rim = ((mask != 0) & (erode_cross(mask) == 0)).astype(np.uint8)  # :797-799 mask - erode(mask)
    enforce_gradient_bound(dx, dy, raw, rim)
    return {"dx": dx, "dy": dy, "raw": raw, "mask": mask}


def region_system(merged: Dict[str, np.ndarray], channel: int):
    """The blend restricted to the union region: unknowns = pixels of `mask` in raster order, 5-point Laplacian
    (diagonal 4, -1 to every neighbour inside), right-hand side = -divergence of the merged field with backward/
    forward differences plus the merged colours of the neighbours OUTSIDE the region (Dirichlet values), start
    vector = the merged colours.  Returns (values, col, row_offset, colour, ys, xs, b, x0)."""
    mask = merged["mask"] != 0
    dx = merged["dx"][..., channel].astype(np.float64)
    dy = merged["dy"][..., channel].astype(np.float64)
    raw = merged["raw"][..., channel].astype(np.float64)
    v, c, r, colour, ys, xs = synth.masked_laplacian_csr(mask)
    H, W = mask.shape
    div = np.zeros((H, W))
    div -= dx + dy                                                   # -gx(y,x) - gy(y,x)
    div[:, 1:] += dx[:, :-1]                                         # + gx(y,x-1)
    div[1:, :] += dy[:-1, :]                                         # + gy(y-1,x)
    pad_m = np.pad(mask, 1)
    pad_r = np.pad(raw, 1)
    outside = np.zeros((H, W))
    for sy, sx in ((0, 1), (2, 1), (1, 0), (1, 2)):
        nb_in = pad_m[sy:sy + H, sx:sx + W]
        outside += np.where(nb_in, 0.0, pad_r[sy:sy + H, sx:sx + W])
    b = (div + outside)[ys, xs]
    return v, c, r, colour, ys, xs, b, raw[ys, xs]
