#!/usr/bin/env python3
"""Randomised parity soak (GPU): the temporally blocked pass against the in-place kernels on random shapes,
iteration counts, tilings and channel counts, every pixel; the reference-order sweep against the oracle on
small random shapes; row-blocked runs against the single block; (round 2) irregular regions through the mask
grid / the raster-region dispatch / the sliced-ELL kernels against the oracle, incremental inserts, the three
reference-order implementations, the RCCL entry points at world size 1.  One line per failure, exit 1 if any."""
import os
import sys
import threading

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from coursecomputationalphotography_amd import capi, rowblock, synth  # noqa: E402


def run_grid(W, H, C, b, x0, iters, env):
    for k in ("CCP_GS_FUSE", "CCP_GS_CHUNK", "CCP_GS_TMAX", "CCP_GS_SIDE_ROWS", "CCP_GS_SHORT_EDGES"):
        os.environ.pop(k, None)
    os.environ.update(env)
    g = capi.Grid(W, H, C)
    for ch in range(C):
        g.set_b(b[ch], ch)
        g.set_x(x0[ch], ch)
    if env.get("SOAK_TUNE"):
        g.tune(int(env["SOAK_TUNE"]))
    g.sweep(iters)
    out = np.stack([g.get_x(ch) for ch in range(C)])
    g.close()
    return out


def run(n, seed):
    """Returns the number of failing cases (each printed)."""
    rng = np.random.Generator(np.random.MT19937(seed))
    bad = 0
    for t in range(n):
        W = int(rng.integers(2, 3000)) if rng.random() < 0.8 else int(rng.integers(3000, 9000))
        H = int(rng.integers(2, 2000))
        C = int(rng.choice([1, 1, 3]))
        iters = int(rng.integers(1, 40))
        env = {}
        if rng.random() < 0.5:
            env["CCP_GS_CHUNK"] = str(int(rng.integers(8, 400)))
        if rng.random() < 0.3:
            env["CCP_GS_TMAX"] = str(int(rng.integers(1, 9)))
        if rng.random() < 0.2:
            env["CCP_GS_SIDE_ROWS"] = str(int(rng.integers(2, 64)))
        if rng.random() < 0.3 and "CCP_GS_CHUNK" not in env:
            env["SOAK_TUNE"] = str(int(rng.integers(1, 9)))       # tuned (depth, chunk height) tables
        b = rng.uniform(-3, 3, (C, H, W))
        x0 = rng.uniform(0, 255, (C, H, W))
        try:
            a = run_grid(W, H, C, b, x0, iters, {**env, "CCP_GS_FUSE": "1"})
            r = run_grid(W, H, C, b, x0, iters, {"CCP_GS_FUSE": "0"})
        except Exception as e:
            bad += 1
            print("ERROR fused/in-place", W, H, C, iters, env, repr(e), flush=True)
            continue
        if not np.array_equal(a, r):
            bad += 1
            print("MISMATCH fused/in-place", W, H, C, iters, env, float(np.abs(a - r).max()), flush=True)
    # reference order against the oracle
    import oracle
    orc = oracle.Oracle()
    for t in range(max(4, n // 4)):
        W, H, iters = int(rng.integers(1, 700)), int(rng.integers(1, 500)), int(rng.integers(1, 12))
        if W * H < 2:
            continue
        b, _ = synth.poisson_system(W, H, int(rng.integers(1, 1000)))
        want, _, _ = orc.from_csr(*synth.poisson_csr(W, H)).gauss_seidel(b, 0.0, iters)
        g = capi.Grid(W, H, 1)
        g.set_b(b)
        g.fill_x(1.0)
        g.gauss_seidel_lexicographic(0.0, iters, 0)
        got = g.get_x().ravel()
        g.close()
        if not np.array_equal(got, want):
            bad += 1
            print("MISMATCH reference order", W, H, iters, float(np.abs(got - want).max()), flush=True)
    # reference order, MANY groups of sweeps: strips that move left from group to group, edge buffers that are used
    # again two groups later, counts that are not multiples of 8, batches of the stop rule that grow (round 4)
    for t in range(max(4, n // 5)):
        W, H = int(rng.integers(1, 1500)), int(rng.integers(1, 400))
        iters = int(rng.integers(13, 260))
        if W * H < 2:
            continue
        iters = max(13, min(iters, int(4e7 // (W * H)) + 13))
        b, _ = synth.poisson_system(W, H, int(rng.integers(1, 1000)))
        om = orc.from_csr(*synth.poisson_csr(W, H))
        want, _, _ = om.gauss_seidel(b, 0.0, iters)
        g = capi.Grid(W, H, 1)
        g.set_b(b)
        g.fill_x(1.0)
        g.gauss_seidel_lexicographic(0.0, iters, 0)
        got = g.get_x().ravel().copy()
        # ... and with the rule after every sweep, stopping at a sweep of the oracle's choosing
        stop = int(rng.integers(1, iters + 1))
        bs = b * 1e-6
        _, _, eps_k = om.gauss_seidel(bs, 0.0, stop)
        want_s, it_s, _ = om.gauss_seidel(bs, eps_k * (1.0 + 1e-9), iters)
        g.set_b(bs)
        g.fill_x(1.0)
        rep = g.gauss_seidel_lexicographic(eps_k * (1.0 + 1e-9), iters, 1)[0]
        got_s = g.get_x().ravel()
        g.close()
        if not np.array_equal(got, want):
            bad += 1
            print("MISMATCH reference order, many groups", W, H, iters, float(np.abs(got - want).max()), flush=True)
        if rep.iterations != it_s or not np.array_equal(got_s, want_s):
            bad += 1
            print("MISMATCH reference order, stop rule", W, H, iters, stop, rep.iterations, it_s, flush=True)
    nontrivial = 0
    # the stop rule: fused passes against the in-place kernels (same stop sweep, same x), random check periods
    for t in range(max(6, n // 4)):
        W, H, C = int(rng.integers(2, 1500)), int(rng.integers(2, 900)), int(rng.choice([1, 3]))
        every = int(rng.choice([1, 1, 2, 3, 8]))
        b = rng.uniform(-3, 3, (C, H, W)) * 1e-8       # keeps the step sums below the start value 10
        x0 = np.zeros((C, H, W))
        stop_at = int(rng.integers(1, 60))
        res = []
        try:
            for fuse in ("0", "1"):
                os.environ["CCP_GS_FUSE"] = fuse
                g = capi.Grid(W, H, C)
                for ch in range(C):
                    g.set_b(b[ch], ch)
                    g.set_x(x0[ch], ch)
                if fuse == "0":
                    # epsilon just above the step of sweep `stop_at` of channel 0 (measured with the in-place kernels)
                    g.sweep(stop_at - 1)
                    eps = float(g.sweep_l1()[0]) * (1.0 + 1e-9)
                    for ch in range(C):
                        g.set_x(x0[ch], ch)
                reps = g.gauss_seidel(eps, 200, every)
                res.append(([(r.iterations, r.converged) for r in reps], np.stack([g.get_x(ch) for ch in range(C)])))
                g.close()
            nontrivial += int(res[0][0][0][0] == stop_at)
            if res[0][0] != res[1][0] or not np.array_equal(res[0][1], res[1][1]):
                bad += 1
                print("MISMATCH stop rule", W, H, C, every, stop_at, res[0][0], res[1][0], flush=True)
        except Exception as e:
            bad += 1
            print("ERROR stop rule", W, H, C, every, stop_at, repr(e), flush=True)
    os.environ.pop("CCP_GS_FUSE", None)
    print("stop-rule cases that stopped at the planted sweep:", nontrivial, flush=True)
    # general matrices (irregular masks), both orders, against the oracle
    for t in range(max(4, n // 8)):
        Wm, Hm, iters = int(rng.integers(20, 400)), int(rng.integers(20, 300)), int(rng.integers(1, 15))
        mask = synth.disc_mask(Wm, Hm, seed=int(rng.integers(1, 10000)), n_discs=int(rng.integers(1, 40)))
        if mask.sum() < 4:
            continue
        v, c, r, colour, ys, xs = synth.masked_laplacian_csr(mask)
        bb = synth.x_true(len(ys), int(rng.integers(1, 1000)))
        om = orc.from_csr(v, c, r)
        try:
            m = capi.CsrMatrix().upload_compressed(v, c, r)
            x, _ = m.gauss_seidel(bb, 0.0, iters, check_every=0, ordering=capi.ORDER_LEXICOGRAPHIC)
            want, _, _ = om.gauss_seidel(bb, 0.0, iters)
            m.set_colouring(colour, 2)
            x2, _ = m.gauss_seidel(bb, 0.0, iters, check_every=0, ordering=capi.ORDER_MULTICOLOUR)
            want2, _, _ = orc.multicolour_gauss_seidel(v, c, r, colour, bb, 0.0, iters)
            m.close()
            if not np.array_equal(x, want) or not np.array_equal(x2, want2):
                bad += 1
                print("MISMATCH csr", Wm, Hm, iters, np.array_equal(x, want), np.array_equal(x2, want2), flush=True)
        except Exception as e:
            bad += 1
            print("ERROR csr", Wm, Hm, iters, repr(e), flush=True)
    # the image side: gradient field + ATb + composite start + clamp, random images and label maps
    for t in range(max(4, n // 8)):
        W, H, K, iters = int(rng.integers(2, 700)), int(rng.integers(2, 500)), int(rng.integers(1, 6)), int(rng.integers(1, 20))
        if rng.random() < 0.15:
            W = int(rng.choice([1, 2]))
        if rng.random() < 0.15:
            H = int(rng.choice([1, 2]))
        if W * H < 2:
            continue
        imgs = [rng.integers(0, 256, (H, W, 3)).astype(np.uint8) for _ in range(K)]
        label = rng.integers(0, K, (H, W)).astype(np.uint8)
        if rng.random() < 0.5:                       # blocky label maps (seams) instead of per-pixel noise
            label = ((np.arange(W)[None, :] * K // W + np.arange(H)[:, None] * K // H) % K).astype(np.uint8)
        try:
            g = capi.Grid(W, H, 3)
            g.assemble_from_images(imgs, label, init_x=True)
            gx, gy = orc.gradient_field(imgs, label)
            v, c, r = synth.poisson_csr(W, H)
            col = oracle.grid_colour(W, H)
            want_img = np.zeros((H, W, 3), dtype=np.uint8)
            ok = True
            for ch in range(3):
                atb = orc.poisson_rhs(gx, gy, ch, int(imgs[0][0, 0, ch]))
                init = orc.composite_init(imgs, label, ch)
                ok &= np.array_equal(g.get_b(ch).ravel(), atb) and np.array_equal(g.get_x(ch).ravel(), init)
                want, _, _ = orc.multicolour_gauss_seidel(v, c, r, col, atb, 0.0, iters, x0=init)
                orc.clamp_store_u8(want, want_img, ch)
            g.sweep(iters)
            ok &= np.array_equal(g.store_u8(), want_img)
            g.close()
            if not ok:
                bad += 1
                print("MISMATCH image side", W, H, K, iters, flush=True)
        except Exception as e:
            bad += 1
            print("ERROR image side", W, H, K, iters, repr(e), flush=True)
    # tiny and degenerate shapes, both orders, stop rule in the reference order, against the oracle
    for t in range(max(8, n // 4)):
        W, H = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        if rng.random() < 0.3:
            W = int(rng.choice([1, 2, 3]))
        if rng.random() < 0.3:
            H = int(rng.choice([1, 2, 3]))
        if W * H < 2:
            continue
        iters = int(rng.integers(1, 80))
        b, _ = synth.poisson_system(W, H, int(rng.integers(1, 1000)))
        b = b * 1e-3
        v, c, r = synth.poisson_csr(W, H)
        om = orc.from_csr(v, c, r)
        try:
            g = capi.Grid(W, H, 1)
            g.set_b(b)
            g.fill_x(1.0)
            g.sweep(iters)
            rb = g.get_x().ravel().copy()
            want_rb, _, _ = orc.multicolour_gauss_seidel(v, c, r, oracle.grid_colour(W, H), b, 0.0, iters)
            _, _, eps_k = om.gauss_seidel(b, 0.0, iters)
            epsilon = eps_k * (1.0 + 1e-9)
            want, it, _ = om.gauss_seidel(b, epsilon, 500)
            g.fill_x(1.0)
            rep = g.gauss_seidel_lexicographic(epsilon, 500, 1)[0]
            lx = g.get_x().ravel()
            g.close()
            if not np.array_equal(rb, want_rb) or rep.iterations != it or not np.array_equal(lx, want):
                bad += 1
                print("MISMATCH tiny", W, H, iters, np.array_equal(rb, want_rb), rep.iterations, it, np.array_equal(lx, want), flush=True)
        except Exception as e:
            bad += 1
            print("ERROR tiny", W, H, iters, repr(e), flush=True)
    # conjugate gradient (plain, from a start vector, Jacobi-preconditioned) against the oracle, to reduction rounding
    for t in range(max(3, n // 12)):
        W, H, iters = int(rng.integers(3, 300)), int(rng.integers(3, 200)), int(rng.integers(1, 60))
        v, c, r = synth.poisson_csr(W, H)
        b, xt = synth.poisson_system(W, H, int(rng.integers(1, 1000)))
        om = orc.from_csr(v, c, r)
        try:
            m = capi.CsrMatrix().upload_compressed(v, c, r)
            init = xt * 0.9
            x, rep = m.conjugate_gradient(b, 1e-300, iters, init=init)
            want, it = om.conjugate_gradient(b, 1e-300, iters, init)
            xj, repj = m.conjugate_gradient_jacobi(b, 1e-300, iters)
            wantj, itj = om.conjugate_gradient_jacobi(b, 1e-300, iters)
            m.close()
            e1 = np.linalg.norm(x - want) / np.linalg.norm(want)
            e2 = np.linalg.norm(xj - wantj) / np.linalg.norm(wantj)
            if rep.iterations != it or repj.iterations != itj or e1 > 1e-8 or e2 > 1e-8:
                bad += 1
                print("MISMATCH cg", W, H, iters, rep.iterations, it, repj.iterations, itj, e1, e2, flush=True)
        except Exception as e:
            bad += 1
            print("ERROR cg", W, H, iters, repr(e), flush=True)
    # ccp_grid_sweep_edges_first directly: any iteration count within the ghost depth, any band height
    for t in range(max(4, n // 10)):
        W, H = int(rng.integers(50, 2500)), int(rng.integers(200, 1500))
        ghost = 2 * int(rng.integers(1, 20))
        rb = int(rng.integers(0, H - 60))
        rc = int(rng.integers(max(ghost, 20), H - rb + 1)) if H - rb > max(ghost, 20) else H - rb
        j = int(rng.integers(1, ghost // 2 + 1))
        edge = int(rng.integers(1, 2 * ghost + 4))
        try:
            whole = capi.Grid(W, H, 1)
            whole.randomize_x(5, 0.0, 255.0)
            whole.b_from_x()
            whole.fill_x(1.0)
            whole.sweep(j)
            want = whole.get_x()[rb:rb + rc]
            whole.close()
            blk = capi.Grid(W, H, 1, rb, rc, ghost, 0)
            blk.randomize_x(5, 0.0, 255.0)
            blk.b_from_x()
            blk.fill_x(1.0)
            blk.sweep_edges_first(j, edge)
            got = blk.get_x_owned()
            blk.close()
            if not np.array_equal(got, want):
                bad += 1
                print("MISMATCH edges_first", W, H, rb, rc, ghost, j, edge, flush=True)
        except Exception as e:
            bad += 1
            print("ERROR edges_first", W, H, rb, rc, ghost, j, edge, repr(e), flush=True)
    # row blocks against the single block
    from test_gpu_fullsize import ThreadDist
    for t in range(max(3, n // 8)):
        W, world = int(rng.integers(100, 3000)), int(rng.integers(2, 5))
        ghost = 2 * int(rng.integers(1, 12))
        H = int(rng.integers(world * (ghost + 2), world * (ghost + 2) + 600))
        iters = int(rng.integers(1, 50))
        for k in ("CCP_GS_FUSE", "CCP_GS_CHUNK", "CCP_GS_TMAX", "CCP_GS_SIDE_ROWS"):
            os.environ.pop(k, None)
        whole = capi.Grid(W, H, 1)
        whole.randomize_x(7, 0.0, 255.0)
        whole.b_from_x()
        whole.fill_x(1.0)
        whole.sweep(iters)
        want = whole.get_x()
        whole.close()
        parts = rowblock.partition_rows(H, world)
        dist = ThreadDist(world)
        out, errs = [None] * world, []
        overlap = bool(rng.random() < 0.5)

        def run(rank):
            try:
                dist.bind(rank)
                rb, rc = parts[rank]
                blk = rowblock.GridBlock(W, H, 1, rb, rc, ghost, 0)
                blk.grid.randomize_x(7, 0.0, 255.0)
                blk.grid.b_from_x()
                blk.grid.fill_x(1.0)
                solver = rowblock.RowBlockSolver(blk, rank, world, ghost, dist, overlap=overlap).set_partition(parts, H)
                solver.exchange_halos()
                solver.sweep(iters)
                out[rank] = blk.grid.get_x_owned()
                blk.grid.close()
            except Exception as e:
                errs.append(e)
        ts = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
        [x.start() for x in ts]
        [x.join(120) for x in ts]
        if errs or any(o is None for o in out) or not np.array_equal(np.concatenate(out), want):
            bad += 1
            print("MISMATCH row blocks", W, H, world, ghost, iters, overlap, errs[:1], flush=True)
    for k in ("CCP_GS_FUSE", "CCP_GS_CHUNK", "CCP_GS_TMAX", "CCP_GS_SIDE_ROWS", "CCP_GS_SHORT_EDGES", "SOAK_TUNE"):
        os.environ.pop(k, None)
    bad += run_round2(n, rng, orc)
    print("soak done, failures:", bad, flush=True)
    return bad


def random_mask(rng, W, H):
    kind = rng.integers(0, 4)
    if kind == 0:
        return synth.disc_mask(W, H, seed=int(rng.integers(1, 10000)), n_discs=int(rng.integers(1, 30)))
    if kind == 1:
        return rng.uniform(size=(H, W)) < rng.uniform(0.3, 0.9)          # salt and pepper: many components, one-pixel bridges
    if kind == 2:
        m = np.zeros((H, W), dtype=bool)                                   # staircases and L-corners: the ambiguous run turns
        for _ in range(int(rng.integers(1, 12))):
            x0, y0 = int(rng.integers(0, W)), int(rng.integers(0, H))
            w, h = int(rng.integers(1, max(2, W // 2))), int(rng.integers(1, max(2, H // 2)))
            m[y0:y0 + h, x0:x0 + w] ^= True
        return m
    return np.ones((H, W), dtype=bool)


def run_round2(n, rng, orc):
    """Region grids and the raster-region dispatch, incremental inserts, the reference-order modes, the RCCL
    entry points at world size 1."""
    import scipy.sparse as sp
    bad = 0
    # ---- irregular regions: mask grid == region dispatch == sliced ELL == oracle --------------------------
    for t in range(max(6, n // 2)):
        W, H = int(rng.integers(3, 500)), int(rng.integers(3, 400))
        mask = random_mask(rng, W, H)
        if mask.sum() < 2:
            continue
        v, c, r, colour, ys, xs = synth.masked_laplacian_csr(mask)
        nn = len(ys)
        iters = int(rng.integers(1, 30))
        b = rng.uniform(-40, 40, nn)
        x0 = rng.uniform(0, 255, nn)
        if rng.random() < 0.3:
            colour = 1 - colour
        want, _, _ = orc.multicolour_gauss_seidel(v, c, r, colour, b, 0.0, iters, x0=x0)
        outs = {}
        for label, env in (("region", "1"), ("ell", "0")):
            os.environ["CCP_GS_MASKED"] = env
            try:
                m = capi.CsrMatrix().upload_compressed(v, c, r)
                m.set_colouring(colour, 2)
                outs[label], _ = m.gauss_seidel(b, 0.0, iters, x0=x0, check_every=0)
                outs[label + "_path"] = m.last_path()
                m.close()
            except Exception as e:
                bad += 1
                print("ERROR region", label, W, H, iters, repr(e), flush=True)
        os.environ.pop("CCP_GS_MASKED", None)
        for label in ("region", "ell"):
            if label in outs and not np.array_equal(outs[label], want):
                bad += 1
                print("MISMATCH region", label, outs.get(label + "_path"), W, H, iters, float(np.abs(outs[label] - want).max()), flush=True)
        # the same region in the reference's own order, with and without a colouring (mask variant of k_lex_wg)
        want_lex = orc.from_csr(v, c, r).gauss_seidel(b, 0.0, iters, x0=x0)[0]
        for with_colouring in (True, False):
            try:
                m = capi.CsrMatrix().upload_compressed(v, c, r)
                if with_colouring:
                    m.set_colouring(colour, 2)
                got, _ = m.gauss_seidel(b, 0.0, iters, x0=x0, check_every=0, ordering=capi.ORDER_LEXICOGRAPHIC)
                path = m.last_path()
                m.close()
                if not np.array_equal(got, want_lex):
                    bad += 1
                    print("MISMATCH region reference order", with_colouring, path, W, H, iters, float(np.abs(got - want_lex).max()), flush=True)
            except Exception as e:
                bad += 1
                print("ERROR region reference order", with_colouring, W, H, iters, repr(e), flush=True)
        # the colour order with NO colouring from the caller (the facade's case): the canvas parity becomes the colouring
        try:
            m = capi.CsrMatrix().upload_compressed(v, c, r)
            got, _ = m.gauss_seidel(b, 0.0, iters, x0=x0, check_every=0)
            colx, ncx = m.get_colouring()
            path = m.last_path()
            m.close()
            wantx, _, _ = orc.multicolour_gauss_seidel(v, c, r, colx, b, 0.0, iters, x0=x0)
            if not np.array_equal(got, wantx):                             # (a tiny or degenerate mask may stay on the stored matrix)
                bad += 1
                print("MISMATCH region without a colouring", path, ncx, W, H, iters, float(np.abs(got - wantx).max()), flush=True)
        except Exception as e:
            bad += 1
            print("ERROR region without a colouring", W, H, iters, repr(e), flush=True)
        if colour[0] == ((xs[0] + ys[0]) & 1):                            # the mask grid directly (its colour 0 is (x+y) even)
            os.environ["CCP_GS_TMAX"] = str(int(rng.integers(1, 8)))
            os.environ["CCP_GS_CHUNK"] = str(int(rng.integers(8, 200)))
            g = capi.Grid(W, H, 1, mask=mask)
            cb, cx = np.zeros((H, W)), np.zeros((H, W))
            cb[ys, xs], cx[ys, xs] = b, x0
            g.set_b(cb)
            g.set_x(cx)
            g.sweep(iters)
            got = g.get_x()
            g.close()
            os.environ.pop("CCP_GS_TMAX", None)
            os.environ.pop("CCP_GS_CHUNK", None)
            if not np.array_equal(got[ys, xs], want) or np.any(got[~mask]):
                bad += 1
                print("MISMATCH mask grid", W, H, iters, flush=True)
    # ---- incremental inserts on random matrices -----------------------------------------------------------
    for t in range(max(3, n // 6)):
        nn = int(rng.integers(20, 900))
        a = sp.random(nn, nn, density=float(rng.uniform(0.005, 0.05)), random_state=np.random.RandomState(int(rng.integers(1, 1 << 30)))).tocsr()
        a = a + a.T
        a.setdiag(0.0)
        a.eliminate_zeros()
        a = (a + sp.diags(np.asarray(abs(a).sum(axis=1)).ravel() + 1.0)).tocsr()
        a.sort_indices()
        b = rng.uniform(-5, 5, nn)
        m = capi.CsrMatrix().upload_compressed(a.data, a.indices.astype(np.int32), a.indptr.astype(np.int32))
        m.gauss_seidel(b, 0.0, 1, check_every=0)
        m.gauss_seidel(b, 0.0, 1, check_every=0, ordering=capi.ORDER_LEXICOGRAPHIC)
        m.apply_to_vector(b)
        lil = a.tolil()
        for _ in range(int(rng.integers(1, 120))):
            i, j = int(rng.integers(0, nn)), int(rng.integers(0, nn))
            kind = rng.random()
            val = 0.0 if (kind < 0.3 and i != j) else float(rng.uniform(0.05, 2.0)) * (1.0 if i != j else 5.0 + nn * 0.05)
            m.insert(val, i, j)
            lil[i, j] = val
        e = lil.tocsr()
        e.eliminate_zeros()
        e.sort_indices()
        ev, ec, er = e.data.astype(np.float64), e.indices.astype(np.int32), e.indptr.astype(np.int32)
        om = orc.from_csr(ev, ec, er)
        k = int(rng.integers(1, 6))
        x, _ = m.gauss_seidel(b, 0.0, k, check_every=0, ordering=capi.ORDER_LEXICOGRAPHIC)
        ok = np.array_equal(x, om.gauss_seidel(b, 0.0, k)[0]) and np.array_equal(m.apply_to_vector(b), om.apply_to_vector(b))
        col, nc = m.get_colouring()
        x, _ = m.gauss_seidel(b, 0.0, k, check_every=0)
        ok = ok and np.array_equal(x, orc.multicolour_gauss_seidel(ev, ec, er, col, b, 0.0, k)[0])
        m.close()
        if not ok:
            bad += 1
            print("MISMATCH insert", nn, flush=True)
    # ---- reference order: the three implementations agree with the oracle ----------------------------------
    for t in range(max(4, n // 4)):
        W, H, C = int(rng.integers(1, 900)), int(rng.integers(1, 600)), int(rng.choice([1, 3]))
        if W * H < 2:
            continue
        iters = int(rng.integers(1, 20))
        bs = [synth.poisson_system(W, H, int(rng.integers(1, 1000)))[0] for _ in range(C)]
        om = orc.from_csr(*synth.poisson_csr(W, H))
        wants = [om.gauss_seidel(b, 0.0, iters)[0] for b in bs]
        for mode in ("wg", "planes"):
            os.environ["CCP_GS_LEX_MODE"] = mode
            g = capi.Grid(W, H, C)
            for ch in range(C):
                g.set_b(bs[ch], ch)
            g.fill_x(1.0)
            g.gauss_seidel_lexicographic(0.0, iters, 0)
            for ch in range(C):
                if not np.array_equal(g.get_x(ch).ravel(), wants[ch]):
                    bad += 1
                    print("MISMATCH reference order", mode, W, H, C, iters, ch, flush=True)
            g.close()
        os.environ.pop("CCP_GS_LEX_MODE", None)
    # ---- RCCL entry points at world size 1 ------------------------------------------------------------------
    comm = capi.Comm(capi.comm_unique_id(), 0, 1, 0)
    for t in range(3):
        W, H, C, iters = int(rng.integers(2, 1500)), int(rng.integers(2, 900)), int(rng.choice([1, 3])), int(rng.integers(1, 40))
        ref, g = capi.Grid(W, H, C), capi.Grid(W, H, C)
        for h in (ref, g):
            h.randomize_x(5, 0.0, 255.0)
            h.b_from_x()
            h.fill_x(1.0)
        g.attach_comm(comm)
        ref.sweep(iters)
        g.gauss_seidel_rowblocked(0.0, iters, 0)
        same = all(np.array_equal(ref.get_x(ch), g.get_x(ch)) for ch in range(C))
        rr, bb = g.residual_norm2_global()
        rr2, bb2 = ref.residual_norm2()
        if not (same and np.array_equal(rr, rr2) and np.array_equal(bb, bb2)):
            bad += 1
            print("MISMATCH rowblocked world 1", W, H, C, iters, flush=True)
        g.attach_comm(None)
        g.close()
        ref.close()
    comm.close()
    return bad


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2024
    sys.exit(1 if run(n, seed) else 0)


if __name__ == "__main__":
    main()
