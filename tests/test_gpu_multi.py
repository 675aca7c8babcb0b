"""GPU: several temporally blocked passes in ONE launch (k_fused_multi, CCP_GS_MULTI=1) give the bits of the same
passes launched one by one — whole images, several channels, odd pass counts, iteration counts that leave passes of
mixed depth, and row blocks whose stored row range shrinks from pass to pass — and never raise the "a wait gave up"
error word.  The in-place half-sweep kernels are the independent third opinion."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from coursecomputationalphotography_amd import capi
    assert capi.device_count() >= 1
    return capi


def solve(capi, monkeypatch, multi, W, H, C, iters, tiling=None, block=None, fused=True):
    monkeypatch.setenv("CCP_GS_MULTI", "1" if multi else "0")
    kw = {}
    if block:
        kw = dict(row_begin=block[0], row_count=block[1], ghost=block[2])
    g = capi.Grid(W, H, C, **kw)
    g.randomize_x(4242, 0.0, 255.0)
    g.b_from_x()
    g.fill_x(1.0)
    g.halo_refreshed()
    if tiling:
        g.set_tiling(*tiling)
    g.set_fused(fused)
    for n in iters:
        g.sweep(n)
    g.synchronize()                       # CCP_ERR_STATE here if a wave of k_fused_multi gave up waiting
    out = [g.get_x_owned(ch) for ch in range(C)]
    launches = g.last_timing()[1]
    g.close()
    return out, launches


@pytest.mark.parametrize("W,H,C,iters,tiling", [
    (4096, 4096, 3, [32], None),              # BASELINE configs[1]: 4 passes of depth 8
    (4096, 4096, 3, [24], (8, 274)),          # 3 passes (odd: the result lands in the other buffer), one round of tall tiles
    (1000, 300, 1, [37, 16], (8, 32)),        # 4x8 + 5: a group and a single; chunks as short as the halo allows
    (257, 131, 3, [16, 16], (4, 16)),         # depth 4, 2 x 4 passes; side strips only (narrow image)
    (16384, 768, 1, [32], (8, 364)),          # full width, one chunk row + short edge chunks
    (130, 4000, 2, [40], (8, 128)),           # tall and narrow: every strip is a side strip
])
def test_multi_pass_launch_gives_the_single_pass_bits(capi, monkeypatch, W, H, C, iters, tiling):
    one, l1 = solve(capi, monkeypatch, False, W, H, C, iters, tiling)
    many, l2 = solve(capi, monkeypatch, True, W, H, C, iters, tiling)
    for a, b in zip(one, many):
        assert np.array_equal(a, b)
    assert l1 == l2                                                  # the same passes were counted
    ref, _ = solve(capi, monkeypatch, False, W, H, C, iters, tiling, fused=False)
    for a, b in zip(ref, many):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("W,H,block,iters", [
    (16384, 16384, (8192, 2048, 64), [32]),      # an interior block of an 8-GPU run: 4 passes, ranges shrinking by 16 rows per side
    (2000, 3000, (0, 1000, 32), [16]),           # top block: only the lower side shrinks
    (2000, 3000, (2000, 1000, 48), [24]),        # bottom block, 3 passes
])
def test_multi_pass_launch_on_row_blocks(capi, monkeypatch, W, H, block, iters):
    one, _ = solve(capi, monkeypatch, False, W, H, 1, iters, None, block)
    many, _ = solve(capi, monkeypatch, True, W, H, 1, iters, None, block)
    assert np.array_equal(one[0], many[0])


@pytest.mark.parametrize("W,H,iters,tiling", [
    (1500, 1100, [28], (7, 96)),          # 4 passes of depth 7; 1100 = 11 x 96 + 44: a remainder chunk
    (1500, 1100, [21], (7, 160)),         # 3 passes; remainder chunk of 140 rows
    (900, 700, [12, 12], (4, 64)),        # depth 4, 3 + 3 passes; 700 = 10 x 64 + 60
    (2000, 1303, [14], (7, 100)),         # remainder chunk of 3 rows: shorter than the halo
    (1500, 1100, [32], (8, 128)),         # depth 8 (the high-word factor window): 4 passes
])
def test_multi_pass_launch_on_dirichlet_mask_grids(capi, monkeypatch, W, H, iters, tiling):
    """Region grids (BASELINE configs[4]'s form): every tile ordinary, dead tiles complete without running."""
    from coursecomputationalphotography_amd import synth
    mask = synth.disc_mask(W, H, seed=11)
    out = {}
    for multi in (False, True):
        monkeypatch.setenv("CCP_GS_MULTI", "1" if multi else "0")
        g = capi.Grid(W, H, 2, mask=mask)
        g.randomize_x(5, 0.0, 255.0)
        g.b_from_x()
        g.fill_x(1.0)
        g.set_tiling(*tiling)
        for n in iters:
            g.sweep(n)
        g.synchronize()
        out[multi] = [g.get_x(ch) for ch in range(2)]
        g.close()
    for a, b in zip(out[False], out[True]):
        assert np.array_equal(a, b)
