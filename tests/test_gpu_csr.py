"""GPU parity of the general slack-CSR path (Gauss-Seidel both orderings, SpMV, residual) —
through the C ABI, against the reference fixtures and the CPU oracle.  Bit-exact bar."""
import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def capi():
    from coursecomputationalphotography_amd import capi
    assert capi.device_count() >= 1, "no HIP device: the product path has no CPU fallback"
    return capi


def dense_to_vector_arrays(orc, dense):
    """The five slack-CSR arrays the reference's initialize(r,c,list) ends with."""
    m = orc.from_dense(dense)
    vals, cols, rb, nnz, _ = m.storage()
    return m, vals, cols, rb, nnz


def test_known_answer_4x4_lexicographic_bit_exact(capi, golden, orc):
    """The reference's only gaussSeidel test (main6.cc:238-249): every iterate k=1..8, the
    default-epsilon stop at k=8, expected x ~ (1, 2, -1, 1)."""
    d = golden("known_answer_4x4.npz")
    _, vals, cols, rb, nnz = dense_to_vector_arrays(orc, d["A"])
    m = capi.CsrMatrix().upload(4, 4, vals, cols, rb, nnz)
    for k in range(1, 9):
        x, rep = m.gauss_seidel(d["b"], 0.0, k, ordering=capi.ORDER_LEXICOGRAPHIC)
        assert np.array_equal(x, d["gs_iterates"][k - 1]), k
        assert rep.iterations == k
    x, rep = m.gauss_seidel(d["b"], 1e-6, 1000, ordering=capi.ORDER_LEXICOGRAPHIC)
    assert rep.iterations == 8 and rep.converged == 1
    assert np.array_equal(x, d["gs_final"])
    assert np.allclose(x, [1, 2, -1, 1], atol=1e-6)
    # multicolour converges to the same solution (different iterates)
    x, rep = m.gauss_seidel(d["b"], 1e-9, 1000, ordering=capi.ORDER_MULTICOLOUR)
    assert rep.converged == 1 and np.allclose(x, [1, 2, -1, 1], atol=1e-8)
    m.close()


@pytest.mark.parametrize("name", ["poisson_8x8.npz", "poisson_17x13.npz", "poisson_64x64.npz"])
def test_poisson_csr_both_orderings(capi, golden, name):
    import oracle
    from coursecomputationalphotography_amd import synth
    d = golden(name)
    W, H = int(d["W"]), int(d["H"])
    v, c, r = synth.poisson_csr(W, H)
    m = capi.CsrMatrix().upload_compressed(v, c, r)
    for k in (1, 2, 10, 50):
        x, _ = m.gauss_seidel(d["b"], 0.0, k, check_every=0, ordering=capi.ORDER_LEXICOGRAPHIC)
        assert np.array_equal(x, d[f"x_lex_k{k}"]), (name, "lex", k)
    m.set_colouring(oracle.grid_colour(W, H), 2)
    for k in (1, 2, 10, 50):
        x, _ = m.gauss_seidel(d["b"], 0.0, k, check_every=0, ordering=capi.ORDER_MULTICOLOUR)
        assert rel_l2(x, d[f"x_rb_k{k}"]) <= TOL
        assert np.array_equal(x, d[f"x_rb_k{k}"]), (name, "rb", k)
    # library-chosen greedy colouring of a full grid is the checkerboard too
    m.set_colouring(None)
    x, _ = m.gauss_seidel(d["b"], 0.0, 10, check_every=0, ordering=capi.ORDER_MULTICOLOUR)
    assert np.array_equal(x, d["x_rb_k10"])
    assert np.array_equal(m.apply_to_vector(d["x_true"]), d["spmv_x_true"])
    rr, bb = m.residual_norm2(d["b"], d["x_rb_k10"])
    want = float(np.sum(d["resid_rb_k10"] ** 2))
    assert abs(rr - want) <= 1e-12 * want and abs(bb - float(np.sum(d["b"] ** 2))) <= 1e-12 * bb
    m.close()


def test_irregular_mask_fixture(capi, golden):
    d = golden("mask_61x47.npz")
    m = capi.CsrMatrix().upload_compressed(d["values"], d["cols"], d["row_offset"])
    assert np.array_equal(m.apply_to_vector(d["x_true"]), d["b"])
    m.set_colouring(d["colour"], 2)
    for k in (1, 2, 10, 50):
        x, _ = m.gauss_seidel(d["b"], 0.0, k, check_every=0, ordering=capi.ORDER_MULTICOLOUR)
        assert np.array_equal(x, d[f"x_rb_k{k}"]), ("rb", k)
        x, _ = m.gauss_seidel(d["b"], 0.0, k, check_every=0, ordering=capi.ORDER_LEXICOGRAPHIC)
        assert np.array_equal(x, d[f"x_lex_k{k}"]), ("lex", k)
    m.close()


def test_slack_ingest_fixture(capi, golden, orc):
    """Matrix ingested with per-row slack / trailing empty rows (sparse-matrix.h:560-619)."""
    d = golden("slack_ingest_12.npz")
    n = 12
    om = orc.from_eigen_row_major(d["values"], d["row_offset"], d["cols"], n, n, d["non_zeros"])
    vals, cols, rb, nnz, _ = om.storage()
    m = capi.CsrMatrix().upload(n, n, vals, cols, rb, nnz)
    assert np.array_equal(m.apply_to_vector(d["xin"]), d["spmv"])
    x, _ = m.gauss_seidel(d["b"], 0.0, 7, check_every=0, ordering=capi.ORDER_LEXICOGRAPHIC)
    assert np.array_equal(x, d["gs_k7"])
    m.close()


def test_bad_colouring_rejected(capi):
    from coursecomputationalphotography_amd import synth
    v, c, r = synth.poisson_csr(6, 5)
    m = capi.CsrMatrix().upload_compressed(v, c, r)
    m.set_colouring(np.zeros(30, dtype=np.int32), 1)
    with pytest.raises(capi.CcpError) as e:
        m.gauss_seidel(np.ones(30), 0.0, 1)
    assert e.value.status == 6
    m.close()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_sparse_vs_oracle(capi, orc, seed):
    """Random diagonally dominant sparse systems, general multi-colouring (greedy)."""
    from coursecomputationalphotography_amd import synth
    g = synth.rng(seed)
    n = 200 + 57 * seed
    dense = np.where(g.uniform(size=(n, n)) < 0.02, g.uniform(-1, 1, (n, n)), 0.0)
    dense = dense + dense.T
    np.fill_diagonal(dense, np.abs(dense).sum(axis=1) + 1.0)
    dense[5, :] = 0.0                      # an empty row: skipped, x stays at its start value
    rows, cols = np.nonzero(dense)
    vals = dense[rows, cols]
    om = orc.from_vector(rows, cols, vals)
    v, c, rb, nnz, _ = om.storage()
    m = capi.CsrMatrix().upload(om.n_rows, om.n_cols, v, c, rb, nnz)
    b = g.uniform(-10, 10, n)
    want, it, eps = om.gauss_seidel(b, 1e-9, 500)
    x, rep = m.gauss_seidel(b, 1e-9, 500, ordering=capi.ORDER_LEXICOGRAPHIC)
    assert np.array_equal(x, want) and rep.iterations == it
    assert abs(rep.last_l1_step - eps) <= 1e-9 * max(eps, 1e-300) + 1e-18
    xm, repm = m.gauss_seidel(b, 1e-9, 500, ordering=capi.ORDER_MULTICOLOUR)
    assert repm.converged == 1 and rel_l2(xm, want) < 1e-8
    assert np.array_equal(m.apply_to_vector(b), om.apply_to_vector(b))
    m.close()


def test_config0_512x512_lexicographic_bit_exact(capi, orc):
    """BASELINE.json configs[0]: 512x512 single-channel Poisson, the reference's own (lexicographic)
    gaussSeidel — reproduced bit for bit on the GPU by level scheduling (1023 levels)."""
    from coursecomputationalphotography_amd import synth
    W = H = 512
    v, c, r = synth.poisson_csr(W, H)
    b, _ = synth.poisson_system(W, H, 1234)
    m = capi.CsrMatrix().upload_compressed(v, c, r)
    x, rep = m.gauss_seidel(b, 0.0, 20, check_every=0, ordering=capi.ORDER_LEXICOGRAPHIC)
    want, _, _ = orc.from_csr(v, c, r).gauss_seidel(b, 0.0, 20)
    assert rep.iterations == 20 and np.array_equal(x, want)
    assert x[-1] == 1.0
    m.close()


@pytest.mark.parametrize("name", ["poisson_8x8.npz", "poisson_17x13.npz", "poisson_64x64.npz"])
def test_conjugate_gradient_vs_oracle(capi, orc, golden, name):
    """conjugateGradient with and without the composite-style initial guess (sparse-matrix.h:396-434,
    call site PhotoMontage.cpp:613).  Tree-ordered reductions: tolerance 1e-9 relative L2 while
    the iteration is far from the rounding floor (north-star bar: 1e-5)."""
    from coursecomputationalphotography_amd import synth
    d = golden(name)
    W, H = int(d["W"]), int(d["H"])
    v, c, r = synth.poisson_csr(W, H)
    om = orc.from_csr(v, c, r)
    m = capi.CsrMatrix().upload_compressed(v, c, r)
    init = synth.x_true(W * H, 7)
    for k in (1, 5, 25):
        for ini in (None, init):
            want, it = om.conjugate_gradient(d["b"], 1e-10, k, ini)
            x, rep = m.conjugate_gradient(d["b"], 1e-10, k, ini)
            assert rep.iterations == it == k
            assert rel_l2(x, want) <= 1e-9, (k, rel_l2(x, want))
    # run to convergence: same stop iteration (+-1) and the true solution (pixel n-1 is free: empty row)
    want, it = om.conjugate_gradient(d["b"], 1e-8, 5000)
    x, rep = m.conjugate_gradient(d["b"], 1e-8, 5000)
    assert rep.converged == 1 and abs(rep.iterations - it) <= 2
    assert np.abs(x[:-1] - d["x_true"][:-1]).max() < 1e-6
    m.close()


def test_stored_matrix_cg_fused_loop_equals_three_pass(capi, orc, monkeypatch):
    """conjugateGradient on the STORED matrix (sliced ELL; recognition switched off): the fused loop (default: x's update
    folded into the next pass over p, k_sell_cg_apply, 12 nnz + 72 B per row) gives the three-pass loop's iterates
    (CCP_GS_CG_FUSED=0) bit for bit — every count, with and without an initial guess, and the same stop iteration — on
    the Poisson matrix, on the Laplacian of an irregular region and on a matrix with rows of very different lengths;
    both agree with the oracle to rounding."""
    import scipy.sparse as sp
    from coursecomputationalphotography_amd import synth
    monkeypatch.setenv("CCP_GS_STRUCTURED", "0")
    monkeypatch.setenv("CCP_GS_MASKED", "0")
    systems = [synth.poisson_csr(61, 47)]
    mask = synth.disc_mask(96, 96, seed=9)
    systems.append(synth.masked_laplacian_csr(mask)[:3])
    rng = np.random.default_rng(5)
    n = 3000
    a = sp.random(n, n, density=0.002, random_state=7, format="csr")
    a = a + a.T
    a = (a + sp.diags(np.asarray(abs(a).sum(axis=1)).ravel() + 1.0)).tocsr()          # symmetric, strictly diagonally dominant
    a.sort_indices()
    systems.append((a.data.astype(np.float64), a.indices.astype(np.int32), a.indptr.astype(np.int32)))
    for v, c, r in systems:
        nrows = len(r) - 1
        x0 = rng.standard_normal(nrows)
        om = orc.from_csr(v, c, r)
        m = capi.CsrMatrix().upload_compressed(v, c, r)
        b = m.apply_to_vector(rng.standard_normal(nrows))       # (consistent: the Poisson matrix has an empty row)
        for k in (1, 2, 7, 40):
            for ini in (None, x0):
                monkeypatch.setenv("CCP_GS_CG_FUSED", "0")
                xa, ra = m.conjugate_gradient(b, 1e-30, k, ini)
                monkeypatch.delenv("CCP_GS_CG_FUSED")
                xb, rb = m.conjugate_gradient(b, 1e-30, k, ini)
                assert ra.iterations == rb.iterations == k and np.array_equal(xa, xb), (nrows, k, ini is not None)
                want, _ = om.conjugate_gradient(b, 1e-30, k, ini)
                assert rel_l2(xb, want) <= 1e-9
        monkeypatch.setenv("CCP_GS_CG_FUSED", "0")
        xa, ra = m.conjugate_gradient(b, 1e-9, 4000)
        monkeypatch.delenv("CCP_GS_CG_FUSED")
        xb, rb = m.conjugate_gradient(b, 1e-9, 4000)
        assert ra.converged == rb.converged == 1 and ra.iterations == rb.iterations and np.array_equal(xa, xb)
        m.close()


def test_known_answer_cg(capi, orc, golden):
    d = golden("known_answer_4x4.npz")
    _, vals, cols, rb, nnz = dense_to_vector_arrays(orc, d["A"])
    m = capi.CsrMatrix().upload(4, 4, vals, cols, rb, nnz)
    x, rep = m.conjugate_gradient(d["b"], 1e-16, 1000)          # main6.cc:251 (lab3 defaults differ only in eps)
    assert np.allclose(x, d["cg_final"], atol=1e-12) and np.allclose(x, [1, 2, -1, 1], atol=1e-12)
    m.close()


def test_conjugate_gradient_reference_fixture(capi, golden):
    """GPU conjugateGradient against the compiled reference's iterates (fixture)."""
    from coursecomputationalphotography_amd import synth
    d = golden("cg_17x13.npz")
    v, c, r = synth.poisson_csr(17, 13)
    m = capi.CsrMatrix().upload_compressed(v, c, r)
    for k in (1, 5, 25):
        x, _ = m.conjugate_gradient(d["b"], 1e-10, k)
        assert rel_l2(x, d[f"x_cg_k{k}"]) <= 1e-9
        x, _ = m.conjugate_gradient(d["b"], 1e-10, k, d["init"])
        assert rel_l2(x, d[f"x_cg_init_k{k}"]) <= 1e-9
    x, rep = m.conjugate_gradient(d["b"], 1e-8, 5000)
    assert rep.converged == 1 and rel_l2(x[:-1], d["x_cg_converged"][:-1]) <= 1e-7
    m.close()


def test_poisson_matrix_is_recognised_and_swept_matrix_free(capi, orc, monkeypatch):
    """The CSR entry point recognises SolveChannel's matrix and takes the grid kernels: the two
    independent GPU implementations (sliced ELL vs matrix-free fused sweep) and the oracle agree
    bit for bit; a perturbed matrix must NOT be recognised."""
    import oracle
    from coursecomputationalphotography_amd import synth
    W, H, k = 300, 211, 24
    v, c, r = synth.poisson_csr(W, H)
    b, _ = synth.poisson_system(W, H, 8)
    x0 = synth.x_true(W * H, 9)
    col = oracle.grid_colour(W, H)
    want, _, _ = orc.multicolour_gauss_seidel(v, c, r, col, b, 0.0, k, x0=x0)
    m = capi.CsrMatrix().upload_compressed(v, c, r)
    x_fast, rep_fast = m.gauss_seidel(b, 0.0, k, x0=x0, check_every=0)           # structured dispatch
    assert np.array_equal(x_fast, want)
    m.set_colouring(col, 2)
    x_fast2, _ = m.gauss_seidel(b, 0.0, k, x0=x0, check_every=0)
    assert np.array_equal(x_fast2, want)
    m.close()
    monkeypatch.setenv("CCP_GS_STRUCTURED", "0")                                    # force the sliced-ELL path
    m = capi.CsrMatrix().upload_compressed(v, c, r)
    x_sell, rep_sell = m.gauss_seidel(b, 0.0, k, x0=x0, check_every=0)
    assert np.array_equal(x_sell, want)
    m.close()
    monkeypatch.delenv("CCP_GS_STRUCTURED")
    # stop rule through the dispatch: same iteration count as the oracle
    bs = b * 1e-3
    want2, it, e = orc.multicolour_gauss_seidel(v, c, r, col, bs, 2.0, 1000)
    m = capi.CsrMatrix().upload_compressed(v, c, r)
    x2, rep2 = m.gauss_seidel(bs, 2.0, 1000)
    assert rep2.iterations == it and rep2.converged == 1 and np.array_equal(x2, want2)
    m.close()
    # one perturbed coefficient: general path, still correct against the oracle
    v2 = v.copy(); v2[1000] = -0.5
    want3, _, _ = orc.multicolour_gauss_seidel(v2, c, r, col, b, 0.0, 5)
    m = capi.CsrMatrix().upload_compressed(v2, c, r)
    x3, _ = m.gauss_seidel(b, 0.0, 5, check_every=0)
    assert np.array_equal(x3, want3)
    m.close()


def test_edge_shapes(capi, orc):
    """Degenerate inputs: 1x1 system, all-empty matrix, rectangular SpMV, zero iterations."""
    m = capi.CsrMatrix().upload(1, 1, [5.0], [0], [0], [1])
    x, rep = m.gauss_seidel([10.0], 1e-12, 50, ordering=capi.ORDER_LEXICOGRAPHIC)
    assert x.tolist() == [2.0] and rep.converged == 1 and rep.iterations == 2     # 2nd sweep: step 0 <= eps
    x, rep = m.gauss_seidel([10.0], 1e-12, 50, ordering=capi.ORDER_MULTICOLOUR)
    assert x.tolist() == [2.0]
    x, rep = m.gauss_seidel([10.0], 0.0, 0)                                        # max_iteration 0: start vector
    assert x.tolist() == [1.0] and rep.iterations == 0
    m.close()
    # every row empty: a_ii == 0 everywhere -> all rows skipped, x stays 1.0, eps = 0 after one sweep
    m = capi.CsrMatrix().upload(3, 3, [0.0], [0], [0, 0, 0], [0, 0, 0])
    x, rep = m.gauss_seidel([1.0, 2.0, 3.0], 1e-6, 10, ordering=capi.ORDER_LEXICOGRAPHIC)
    assert x.tolist() == [1.0, 1.0, 1.0] and rep.iterations == 1 and rep.converged == 1
    assert m.apply_to_vector([4.0, 5.0, 6.0]).tolist() == [0.0, 0.0, 0.0]
    m.close()
    # rectangular matrix: applyToVector only (the reference's gaussSeidel needs a square one)
    dense = np.array([[1.0, 0, 2, 0, 0], [0, 0, 0, 0, 3], [0, 4, 0, 0, 0]])
    rows, cols = np.nonzero(dense)
    om = orc.from_vector(rows, cols, dense[rows, cols])
    v, c, rb, nnz, _ = om.storage()
    m = capi.CsrMatrix().upload(3, 5, v, c, rb, nnz)
    vin = np.array([1.0, 2, 3, 4, 5])
    assert np.array_equal(m.apply_to_vector(vin), dense @ vin)
    with pytest.raises(capi.CcpError) as e:
        m.gauss_seidel(np.ones(5), 0.0, 1)
    assert e.value.status == 6
    m.close()
    # bad arrays are rejected, not dereferenced
    m = capi.CsrMatrix()
    with pytest.raises(capi.CcpError) as e:
        m.upload(2, 2, [1.0, 1.0], [0, 7], [0, 1], [1, 1])        # column 7 of a 2-column matrix
    assert e.value.status == 1
    with pytest.raises(capi.CcpError) as e:
        m.gauss_seidel([1.0, 1.0], 0.0, 1)                          # nothing uploaded
    assert e.value.status == 5
    m.close()


def test_conjugate_gradient_jacobi_vs_oracle_and_fixture(capi, orc, golden):
    """ccp_csr_conjugate_gradient_jacobi = SparseMatrix::conjugateGradientEigen: same iteration counts,
    iterates to reduction rounding (the device sums in a tree, the reference left to right)."""
    from coursecomputationalphotography_amd import synth
    d = golden("cg_jacobi_17x13.npz")
    v, c, r = synth.poisson_csr(17, 13)
    m = capi.CsrMatrix().upload_compressed(v, c, r)
    for k in (1, 5, 25):
        x, rep = m.conjugate_gradient_jacobi(d["b"], 1e-16, k)
        assert rep.iterations == k
        assert np.linalg.norm(x - d[f"x_k{k}"]) <= 1e-9 * np.linalg.norm(d[f"x_k{k}"])
    m.close()
    mv, mc, mr, _, ys, _ = synth.masked_laplacian_csr(synth.disc_mask(61, 47, seed=11))
    mm = capi.CsrMatrix().upload_compressed(mv, mc, mr)
    x, rep = mm.conjugate_gradient_jacobi(d["mask_b"], 1e-9, 5000)
    want, it = orc.from_csr(mv, mc, mr).conjugate_gradient_jacobi(d["mask_b"], 1e-9, 5000)
    assert rep.converged == 1 and abs(rep.iterations - it) <= 1
    assert np.linalg.norm(x - d["mask_x_converged"]) <= 1e-8 * np.linalg.norm(d["mask_x_converged"])
    mm.close()


def _random_spd_csr(seed, n, density):
    """Random symmetric, strictly diagonally dominant sparse matrix (not bipartite: >= 3 colours)."""
    import scipy.sparse as sp
    from coursecomputationalphotography_amd import synth
    g = synth.rng(seed)
    a = sp.random(n, n, density=density, random_state=np.random.RandomState(seed), data_rvs=lambda k: g.uniform(-1, 1, k)).tocsr()
    a = a + a.T
    a.setdiag(0.0)
    a.eliminate_zeros()
    a = (a + sp.diags(np.asarray(abs(a).sum(axis=1)).ravel() + 1.0)).tocsr()
    a.sort_indices()
    return a.data.astype(np.float64), a.indices.astype(np.int32), a.indptr.astype(np.int32), g


@pytest.mark.parametrize("seed,n,density", [(11, 300, 0.03), (12, 1777, 0.004), (13, 40000, 0.0002)])
def test_multicolour_iterates_equal_oracle_with_the_exported_colouring(capi, orc, seed, n, density):
    """the general multi-colour sweep on non-bipartite matrices, iterate for iterate.
    The library colours the rows itself (greedy); ccp_csr_get_colouring exports that colouring, the oracle
    runs the reference gaussSeidel on P A P^T with the rows listed colour by colour (sparse-matrix.h:350-380)
    and every iterate k = 1, 2, 10 must agree bit for bit — as must the stop sweep of the L1 rule."""
    v, c, r, g = _random_spd_csr(seed, n, density)
    b = g.uniform(-10, 10, n)
    m = capi.CsrMatrix().upload_compressed(v, c, r)
    colour, nc = m.get_colouring()
    assert nc >= 3 and colour.min() == 0 and colour.max() == nc - 1
    # a proper colouring: no stored off-diagonal couples two rows of one colour
    rows = np.repeat(np.arange(n), np.diff(r))
    off = rows != c
    assert not np.any(colour[rows[off]] == colour[c[off]])
    for k in (1, 2, 10):
        x, rep = m.gauss_seidel(b, 0.0, k, check_every=0, ordering=capi.ORDER_MULTICOLOUR)
        want, _, _ = orc.multicolour_gauss_seidel(v, c, r, colour, b, 0.0, k)
        assert rep.iterations == k and np.array_equal(x, want), k
    want, it, eps = orc.multicolour_gauss_seidel(v, c, r, colour, b, 1e-7, 400)
    x, rep = m.gauss_seidel(b, 1e-7, 400, check_every=1, ordering=capi.ORDER_MULTICOLOUR)
    assert it < 400 and rep.iterations == it and rep.converged == 1 and np.array_equal(x, want)
    assert abs(rep.last_l1_step - eps) <= 1e-9 * eps + 1e-18
    # the caller's own colouring is what comes back
    mine = (np.arange(n) % (nc + 2)).astype(np.int32)
    m.set_colouring(None)
    m2 = capi.CsrMatrix().upload_compressed(v, c, r)
    try:
        m2.set_colouring(mine, nc + 2)
        got, got_nc = m2.get_colouring()
        assert got_nc == nc + 2 and np.array_equal(got, mine)
    except capi.CcpError as e:                       # arbitrary labels are rarely a proper colouring: rejected, not used
        assert e.status == 6
    m2.close()
    m.close()


def test_apply_and_residual_of_the_poisson_matrix_need_no_image(capi, orc, monkeypatch):
    """applyToVector / the residual on SolveChannel's matrix run matrix-free on the grid twin: the stored-order products
    (bit-identical to the oracle's applyToVector and to the sliced-ELL kernel), and no image of the matrix is built."""
    from coursecomputationalphotography_amd import synth
    W, H = 301, 187
    v, c, r = synth.poisson_csr(W, H)
    xv = synth.x_true(W * H, 5)
    om = orc.from_csr(v, c, r)
    want = om.apply_to_vector(xv)
    m = capi.CsrMatrix().upload_compressed(v, c, r)
    got = m.apply_to_vector(xv)
    assert np.array_equal(got, want)
    b = want + 0.25
    rr, bb = m.residual_norm2(b, xv)
    assert np.isclose(rr, 0.0625 * W * H, rtol=1e-6, atol=0.0) and np.isclose(bb, float(np.dot(b, b)), rtol=1e-12, atol=0.0)
    assert m.edit_stats()["image_uploads"] == 0
    m.close()
    monkeypatch.setenv("CCP_GS_STRUCTURED", "0")                     # the stored-matrix kernel: same bits
    m2 = capi.CsrMatrix().upload_compressed(v, c, r)
    assert np.array_equal(m2.apply_to_vector(xv), want)
    assert m2.edit_stats()["image_uploads"] >= 1
    m2.close()


def test_upload_succeeds_when_the_background_copy_finds_no_room(capi, orc, monkeypatch):
    """ccp_csr_upload starts copying the matrix's structure to the device in the background for the recognition of the
    first solve.  That copy is best effort: a device allocation that fails there must not fail the upload (the copy is
    made when something needs it), and the copy is given back once a grid twin sweeps the matrix."""
    from coursecomputationalphotography_amd import synth
    mask = synth.disc_mask(1400, 1400, seed=4321)
    v, c, r, colour, ys, xs = synth.masked_laplacian_csr(mask)
    n = len(ys)
    assert len(v) >= (1 << 22)                                  # large enough for the background copy
    b = synth.csr_apply(v, c, r, synth.x_true(n, 9))
    want, _, _ = orc.multicolour_gauss_seidel(v, c, r, colour, b, 0.0, 5)
    # (1) the allocation refused (test seam): the upload succeeds, nothing is held, the solve still finds the region
    monkeypatch.setenv("CCP_GS_FAIL_EAGER_ALLOC", "1")
    m = capi.CsrMatrix().upload_compressed(v, c, r)
    monkeypatch.delenv("CCP_GS_FAIL_EAGER_ALLOC")
    assert m.device_footprint() == (0, 1)
    m.set_colouring(colour, 2)
    x, _ = m.gauss_seidel(b, 0.0, 5, check_every=0)
    assert m.last_path().startswith("region grid") and np.array_equal(x, want)
    assert m.device_footprint()[0] == 0                         # the twin has the matrix: no copy of the stored one
    # (2) the ordinary case: structure only (row offsets + columns, no values), given back after the recognition
    m.upload_compressed(v, c, r)
    held, skipped = m.device_footprint()
    assert held == 8 * (n + 1) + 4 * len(v) and skipped == 1
    m.set_colouring(colour, 2)
    x, _ = m.gauss_seidel(b, 0.0, 5, check_every=0)
    assert m.last_path().startswith("region grid") and np.array_equal(x, want)
    assert m.device_footprint()[0] == 0
    # (3) kept on the stored-matrix path, where the images are built from it (values included)
    monkeypatch.setenv("CCP_GS_MASKED", "0")
    m.upload_compressed(v, c, r)
    monkeypatch.delenv("CCP_GS_MASKED")
    m.set_colouring(colour, 2)
    x, _ = m.gauss_seidel(b, 0.0, 5, check_every=0)
    assert m.last_path() == "sliced ELL" and np.array_equal(x, want)
    assert m.device_footprint()[0] == 8 * (n + 1) + 12 * len(v)
    m.close()
    # SolveChannel's matrix is examined on the host and swept on its grid twin: never copied at all
    pv, pc, pr = synth.poisson_csr(1100, 1000)
    assert len(pv) >= (1 << 22)
    p = capi.CsrMatrix().upload_compressed(pv, pc, pr)
    assert p.device_footprint() == (0, 0)
    p.close()
