"""GPU: the Dirichlet-mask grid (CCP_GRID_DIRICHLET_MASK) and the raster-region dispatch of the general CSR path
(BASELINE configs[4]: the 5-point Laplacian of an irregular pixel region).  Everything is compared bit for bit
with the oracle = the reference gaussSeidel (sparse-matrix.h:350-380) on the colour-major permuted CSR matrix
of the same region, and with the library's own sliced-ELL path (CCP_GS_MASKED=0)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from coursecomputationalphotography_amd import capi
    assert capi.device_count() >= 1
    return capi


def region_system(mask, seed):
    from coursecomputationalphotography_amd import synth
    v, c, r, colour, ys, xs = synth.masked_laplacian_csr(mask)
    n = len(ys)
    rng = np.random.Generator(np.random.MT19937(seed))
    b = rng.uniform(-40.0, 40.0, n)
    x0 = rng.uniform(0.0, 255.0, n)
    return v, c, r, colour, ys, xs, b, x0


def to_canvas(mask, ys, xs, vec):
    out = np.zeros(mask.shape)
    out[ys, xs] = vec
    return out


MASKS = {
    "discs": lambda: __import__("coursecomputationalphotography_amd.synth", fromlist=["x"]).disc_mask(700, 520, seed=9),
    "salt": lambda: np.random.Generator(np.random.MT19937(2)).uniform(size=(300, 411)) < 0.6,
    "edges": lambda: np.ones((130, 257), dtype=bool),          # the region touches every canvas edge
}


@pytest.mark.parametrize("name", list(MASKS))
@pytest.mark.parametrize("tmax,iters", [(1, 5), (2, 4), (3, 9), (4, 8), (5, 11), (6, 13), (7, 15), (7, 30), (8, 16), (8, 35)])
def test_mask_grid_sweeps_equal_oracle(capi, orc, monkeypatch, name, tmax, iters):
    """ccp_grid_sweep on a Dirichlet-mask grid, every depth of the temporally blocked pass (plus the in-place
    kernels for the odd iteration), against the oracle; pixels outside the region stay exactly 0."""
    mask = MASKS[name]()
    v, c, r, colour, ys, xs, b, x0 = region_system(mask, 5)
    H, W = mask.shape
    monkeypatch.setenv("CCP_GS_TMAX", str(tmax))
    monkeypatch.setenv("CCP_GS_CHUNK", "46")
    g = capi.Grid(W, H, 1, mask=mask)
    g.set_b(to_canvas(mask, ys, xs, b))
    g.set_x(to_canvas(mask, ys, xs, x0))
    g.sweep(iters)
    got = g.get_x()
    want, _, _ = orc.multicolour_gauss_seidel(v, c, r, colour, b, 0.0, iters, x0=x0)
    assert np.array_equal(got[ys, xs], want)
    assert not np.any(got[~mask])
    # b := A x and the residual on the same handle
    om = orc.from_csr(v, c, r)
    rr, bb = g.residual_norm2()
    assert abs(np.sqrt(rr[0] / bb[0]) - om.rel_residual(b, want)) <= 1e-12
    g.b_from_x()
    assert np.array_equal(g.get_b()[ys, xs], om.apply_to_vector(want))
    g.close()


def test_mask_grid_stop_rule_and_unfused(capi, orc):
    """The reference loop with its L1 stop rule on a mask grid (checked passes report every sweep's step) and
    the in-place kernels alone (ccp_grid_set_fused(0)): same stop sweep, same iterate."""
    from coursecomputationalphotography_amd import synth
    mask = synth.disc_mask(256, 200, seed=3)
    v, c, r, colour, ys, xs, b, x0 = region_system(mask, 8)
    b *= 1e-3
    H, W = mask.shape
    want, it, eps = orc.multicolour_gauss_seidel(v, c, r, colour, b, 0.5, 3000)
    assert 10 < it < 3000
    for fused in (True, False):
        g = capi.Grid(W, H, 1, mask=mask)
        g.set_fused(fused)
        g.set_b(to_canvas(mask, ys, xs, b))
        g.fill_x(1.0)
        assert np.array_equal(g.get_x(), mask.astype(np.float64))      # x0 = 1 on the region, 0 around it
        rep = g.gauss_seidel(0.5, 3000, 1)[0]
        assert rep.converged == 1 and rep.iterations == it, fused
        assert np.array_equal(g.get_x()[ys, xs], want), fused
        assert abs(rep.last_l1_step - eps) <= 1e-10 * eps
        g.close()


@pytest.mark.parametrize("name", ["discs", "salt"])
def test_csr_upload_reaches_the_region_grid(capi, orc, monkeypatch, name):
    """The unchanged call: upload the CSR arrays, give the colouring, call gaussSeidel.  The matrix is recognised
    as a raster-region Laplacian and swept by the mask grid; results are those of the sliced-ELL path and of the
    oracle, with and without a start vector, with a fixed count and with the stop rule."""
    mask = MASKS[name]()
    v, c, r, colour, ys, xs, b, x0 = region_system(mask, 12)
    m = capi.CsrMatrix().upload_compressed(v, c, r)
    m.set_colouring(colour, 2)
    x, rep = m.gauss_seidel(b, 0.0, 17, check_every=0)
    assert m.last_path().startswith("region grid"), m.last_path()
    want, _, _ = orc.multicolour_gauss_seidel(v, c, r, colour, b, 0.0, 17)
    assert rep.iterations == 17 and np.array_equal(x, want)
    x, _ = m.gauss_seidel(b, 0.0, 6, x0=x0, check_every=0)
    want, _, _ = orc.multicolour_gauss_seidel(v, c, r, colour, b, 0.0, 6, x0=x0)
    assert np.array_equal(x, want)
    got_col, nc = m.get_colouring()
    assert nc == 2 and np.array_equal(got_col, colour)
    # the stop rule: start close enough that the L1 step falls below the loop's initial eps = 10 (sparse-matrix.h:354)
    near = orc.multicolour_gauss_seidel(v, c, r, colour, b, 0.0, 60)[0]
    threshold = orc.multicolour_gauss_seidel(v, c, r, colour, b, 0.0, 37, x0=near)[2] * (1.0 + 1e-9)   # the step of sweep 37, a hair more
    want, it, eps = orc.multicolour_gauss_seidel(v, c, r, colour, b, threshold, 2000, x0=near)
    if threshold < 10.0:
        assert 2 <= it <= 37
        x, rep = m.gauss_seidel(b, threshold, 2000, x0=near, check_every=1)
        assert rep.converged == 1 and rep.iterations == it and np.array_equal(x, want)
    else:                                   # the reference loop never starts: eps = 10 <= epsilon
        x, rep = m.gauss_seidel(b, threshold, 2000, x0=near, check_every=1)
        assert it == 0 and rep.iterations == 0 and np.array_equal(x, near)
    # the reference's own order: the canvas swept in raster order (k_lex_wg, Dirichlet-mask variant)
    x, _ = m.gauss_seidel(b, 0.0, 3, check_every=0, ordering=capi.ORDER_LEXICOGRAPHIC)
    assert m.last_path().startswith("region grid"), m.last_path()
    assert np.array_equal(x, orc.from_csr(v, c, r).gauss_seidel(b, 0.0, 3)[0])
    m.close()
    # the same matrix kept on the general path
    monkeypatch.setenv("CCP_GS_MASKED", "0")
    m = capi.CsrMatrix().upload_compressed(v, c, r)
    m.set_colouring(colour, 2)
    x2, _ = m.gauss_seidel(b, 0.0, 17, check_every=0)
    assert m.last_path() == "sliced ELL"
    assert np.array_equal(x2, orc.multicolour_gauss_seidel(v, c, r, colour, b, 0.0, 17)[0])
    m.close()


@pytest.mark.parametrize("name", list(MASKS))
def test_region_in_the_references_own_order(capi, orc, monkeypatch, name):
    """gaussSeidel in index order (sparse-matrix.h:350-380, the matrix as it is) on a raster-region Laplacian: the
    region grid swept in raster order gives the oracle's bits for sweep counts that leave passes of 8, 4, 2 and 1,
    with a start vector, and stops where the oracle stops; so does the stored-matrix path (CCP_GS_MASKED=0)."""
    mask = MASKS[name]()
    v, c, r, colour, ys, xs, b, x0 = region_system(mask, 21)
    om = orc.from_csr(v, c, r)
    m = capi.CsrMatrix().upload_compressed(v, c, r)
    for k in (1, 8, 15, 26):
        x, rep = m.gauss_seidel(b, 0.0, k, x0=x0, check_every=0, ordering=capi.ORDER_LEXICOGRAPHIC)
        assert m.last_path().startswith("region grid"), m.last_path()
        want = om.gauss_seidel(b, 0.0, k, x0=x0)[0]
        assert rep.iterations == k and np.array_equal(x, want), (name, k, np.abs(x - want).max())
    x, _ = m.gauss_seidel(b, 0.0, 5, check_every=0, ordering=capi.ORDER_LEXICOGRAPHIC)          # default start: all ones
    assert np.array_equal(x, om.gauss_seidel(b, 0.0, 5)[0])
    near = om.gauss_seidel(b, 0.0, 80)[0]
    threshold = om.gauss_seidel(b, 0.0, 21, x0=near)[2] * (1.0 + 1e-9)
    want, it, eps = om.gauss_seidel(b, threshold, 2000, x0=near)
    x, rep = m.gauss_seidel(b, threshold, 2000, x0=near, check_every=1, ordering=capi.ORDER_LEXICOGRAPHIC)
    assert rep.iterations == it and np.array_equal(x, want), (name, it, rep.iterations)
    if it > 0:
        assert rep.converged == 1 and abs(rep.last_l1_step - eps) <= 1e-12 * eps
    # no colouring was given: the colour-ordered sweep of the same handle uses the library's own colouring (which
    # may need a third colour, and then stays on the stored matrix) — whatever it reports is what it sweeps with
    col, nc = m.get_colouring()
    x, _ = m.gauss_seidel(b, 0.0, 4, x0=x0, check_every=0)
    assert np.array_equal(x, orc.multicolour_gauss_seidel(v, c, r, col, b, 0.0, 4, x0=x0)[0]), (name, nc, m.last_path())
    m.close()
    for var, val in (("CCP_GS_LEX_MODE", "planes"), ("CCP_GS_MASKED", "0")):      # an engine that knows no masks; no region grid
        monkeypatch.setenv(var, val)
        m = capi.CsrMatrix().upload_compressed(v, c, r)
        x, _ = m.gauss_seidel(b, 0.0, 8, x0=x0, check_every=0, ordering=capi.ORDER_LEXICOGRAPHIC)
        assert m.last_path() == "sliced ELL" and np.array_equal(x, om.gauss_seidel(b, 0.0, 8, x0=x0)[0]), var
        m.close()
        monkeypatch.delenv(var)


def test_black_first_colouring_and_edit_fall_back(capi, orc):
    """Colour 0 = (x+y) odd: the embedding shifts by one pixel so that the grid still sweeps colour 0 first.  An
    edited matrix leaves the region grid for the general path (and stays exact)."""
    from coursecomputationalphotography_amd import synth
    mask = synth.disc_mask(300, 300, seed=21)
    v, c, r, colour, ys, xs, b, x0 = region_system(mask, 1)
    flipped = 1 - colour
    m = capi.CsrMatrix().upload_compressed(v, c, r)
    m.set_colouring(flipped, 2)
    x, _ = m.gauss_seidel(b, 0.0, 9, check_every=0)
    assert m.last_path().startswith("region grid")
    assert np.array_equal(x, orc.multicolour_gauss_seidel(v, c, r, flipped, b, 0.0, 9)[0])
    row = len(ys) // 2
    m.insert(5.0, row, row)                                    # a heavier diagonal somewhere
    v2 = v.copy()
    sel = np.arange(r[row], r[row + 1])
    v2[sel[c[sel] == row]] = 5.0
    x, _ = m.gauss_seidel(b, 0.0, 9, check_every=0)
    assert m.last_path() == "sliced ELL"
    assert np.array_equal(x, orc.multicolour_gauss_seidel(v2, c, r, flipped, b, 0.0, 9)[0])
    m.close()


def test_chained_solves_of_any_pass_parity_on_one_handle(capi, orc):
    """The CSR entry point's region grid may end a run of passes in either ping-pong buffer (x and its partner swap roles):
    solves with odd and even pass counts, the reference's order in between and an edit at the end, chained on ONE handle,
    every result against the oracle continued from the previous iterate."""
    from coursecomputationalphotography_amd import synth
    mask = synth.disc_mask(900, 640, seed=21)
    v, c, r, colour, ys, xs, b, x0 = region_system(mask, 12)
    m = capi.CsrMatrix().upload_compressed(v, c, r).set_colouring(colour, 2)
    x = x0
    for iters in (5, 8, 13, 50, 1, 9, 24, 7):                 # 1, 1, 2, 7, 1 (in place), 2, 3, 1 passes at depth <= 8
        got, rep = m.gauss_seidel(b, 0.0, iters, x0=x, check_every=0)
        want, _, _ = orc.multicolour_gauss_seidel(v, c, r, colour, b, 0.0, iters, x0=x)
        assert m.last_path().startswith("region grid") and rep.iterations == iters
        assert np.array_equal(got, want), iters
        x = got
    om = orc.from_csr(v, c, r)
    got, _ = m.gauss_seidel(b, 0.0, 3, x0=x, check_every=0, ordering=capi.ORDER_LEXICOGRAPHIC)
    want = om.gauss_seidel(b, 0.0, 3, x)[0]
    assert np.array_equal(got, want)
    got2, _ = m.gauss_seidel(b, 0.0, 11, x0=got, check_every=0)                       # back to the colour order: 2 passes
    want2, _, _ = orc.multicolour_gauss_seidel(v, c, r, colour, b, 0.0, 11, x0=got)
    assert np.array_equal(got2, want2)
    # b := A x and the residual of a recognised region run on its canvas too: the stored-order products, no image of the matrix
    assert np.array_equal(m.apply_to_vector(got2), om.apply_to_vector(got2))
    rr, bb = m.residual_norm2(b, got2)
    assert abs(np.sqrt(rr / bb) - om.rel_residual(b, got2)) <= 1e-12
    assert m.edit_stats()["image_uploads"] == 0
    m.close()


@pytest.mark.parametrize("ask_first", [False, True])
def test_region_without_a_colouring_takes_its_canvas_parity(capi, orc, ask_first):
    """No colouring from the caller (the facade's case): a raster region is laid out first and its canvas parity becomes the
    library's colouring — a proper 2-colouring whichever way the pieces of the mask merge (greedy in row order needs a
    third colour there, and the sweep would stay on the stored matrix) — exported by ccp_csr_get_colouring, so that the
    reference on P A P^T reproduces the iterates."""
    from coursecomputationalphotography_amd import synth
    mask = synth.disc_mask(640, 480, seed=4321, n_discs=40, rmin=300.0, rmax=1400.0)
    v, c, r, colour_xy, ys, xs, b, x0 = region_system(mask, 3)
    m = capi.CsrMatrix().upload_compressed(v, c, r)
    if ask_first:
        col0, nc0 = m.get_colouring()
        assert nc0 == 2
    got, _ = m.gauss_seidel(b, 0.0, 9, x0=x0, check_every=0)
    assert m.last_path().startswith("region grid")
    col, nc = m.get_colouring()
    assert nc == 2 and (not ask_first or np.array_equal(col, col0))
    rows = np.repeat(np.arange(len(r) - 1), np.diff(r))
    off = rows != c
    assert not np.any(col[rows[off]] == col[c[off]])                       # proper
    want, _, _ = orc.multicolour_gauss_seidel(v, c, r, col, b, 0.0, 9, x0=x0)
    assert np.array_equal(got, want)
    m.close()
