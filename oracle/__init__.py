"""CPU ORACLE — test infrastructure, NOT product code.

ctypes/numpy front-end to ``oracle/ccp_oracle.c`` (the plain-C restatement of the
reference's Gauss-Seidel path) and, when built, to ``oracle/_ref/*.so`` (the compiled,
unmodified reference headers).  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this package; the product package
``coursecomputationalphotography_amd`` never does.

Parity status: pinned (see ccp_oracle.h header comment).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "_build", "libccp_oracle.so")
REF_PROJECT_SO = os.path.join(_HERE, "_ref", "libccp_ref_project.so")
REF_LAB3_SO = os.path.join(_HERE, "_ref", "libccp_ref_lab3.so")

_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


def build(ref: bool = True) -> None:
    """(Re)build the oracle, and the reference drivers when /root/reference is present."""
    subprocess.check_call(["make", "-C", _HERE] + ([] if ref else ["_build/libccp_oracle.so"]),
                          stdout=subprocess.DEVNULL)


class _Matrix(C.Structure):
    _fields_ = [("values", C.c_void_p), ("col_offset", C.c_void_p), ("row_begin", C.c_void_p),
                ("row_num_nze", C.c_void_p), ("row_space_left", C.c_void_p),
                ("n_rows", C.c_int32), ("n_cols", C.c_int32), ("n_values", C.c_int64)]


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


def _opt(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleMatrix:
    """Slack-CSR matrix held by the C oracle (sparse-matrix.h:670-676 layout)."""

    def __init__(self, lib, m: _Matrix):
        self._lib = lib
        self._m = m

    def __del__(self):
        try:
            self._lib.orc_matrix_free(C.byref(self._m))
        except Exception:
            pass

    @property
    def n_rows(self) -> int:
        return self._m.n_rows

    @property
    def n_cols(self) -> int:
        return self._m.n_cols

    def _arr(self, ptr, n, dtype):
        if n == 0:
            return np.zeros(0, dtype=dtype)
        buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype).copy()

    def storage(self):
        """(values, col_offset, row_begin, row_num_nze, row_space_left) copies."""
        m = self._m
        return (self._arr(m.values, m.n_values, np.float64),
                self._arr(m.col_offset, m.n_values, np.int32),
                self._arr(m.row_begin, m.n_rows, np.int32),
                self._arr(m.row_num_nze, m.n_rows, np.int32),
                self._arr(m.row_space_left, m.n_rows, np.int32))

    def at(self, r: int, c: int) -> float:
        return self._lib.orc_at(C.byref(self._m), r, c)

    def dense(self) -> np.ndarray:
        out = np.zeros((self.n_rows, self.n_cols))
        for r in range(self.n_rows):
            for c in range(self.n_cols):
                out[r, c] = self.at(r, c)
        return out

    def gauss_seidel(self, b, epsilon: float = 1e-6, max_iteration: int = 1000,
                     x0=None) -> Tuple[np.ndarray, int, float]:
        b = _f64(b)
        x = np.empty(self.n_cols, dtype=np.float64)
        it = C.c_int(0)
        eps = C.c_double(0)
        x0a = None if x0 is None else _f64(x0)
        rc = self._lib.orc_gauss_seidel(C.byref(self._m), b, _opt(x0a), epsilon, max_iteration,
                                        x, C.byref(it), C.byref(eps))
        if rc != 0:
            raise MemoryError("orc_gauss_seidel failed")
        return x, it.value, eps.value

    def conjugate_gradient(self, b, epsilon: float = 1e-16, max_iteration: int = 1000, init=None):
        b = _f64(b)
        x = np.empty(self.n_cols, dtype=np.float64)
        it = C.c_int(0)
        ia = None if init is None else _f64(init)
        rc = self._lib.orc_conjugate_gradient(C.byref(self._m), b, _opt(ia), epsilon, max_iteration, x, C.byref(it))
        if rc != 0:
            raise MemoryError("orc_conjugate_gradient failed")
        return x, it.value

    def conjugate_gradient_jacobi(self, b, epsilon: float = 1e-16, max_iteration: int = 180):
        b = _f64(b)
        x = np.empty(self.n_cols, dtype=np.float64)
        it = C.c_int(0)
        rc = self._lib.orc_conjugate_gradient_jacobi(C.byref(self._m), b, epsilon, max_iteration, x, C.byref(it))
        if rc != 0:
            raise MemoryError("orc_conjugate_gradient_jacobi failed")
        return x, it.value

    def apply_to_vector(self, v) -> np.ndarray:
        v = _f64(v)
        out = np.zeros(self.n_rows, dtype=np.float64)
        self._lib.orc_apply_to_vector(C.byref(self._m), v, out)
        return out

    def rel_residual(self, b, x) -> float:
        return self._lib.orc_rel_residual(C.byref(self._m), _f64(b), _f64(x))


class Oracle:
    def __init__(self, path: str = ORACLE_SO):
        if not os.path.exists(path):
            build(ref=False)
        L = C.CDLL(path)
        self.lib = L
        MP = C.POINTER(_Matrix)
        L.orc_matrix_free.argtypes = [MP]
        L.orc_from_eigen_row_major.argtypes = [MP, _f64p, C.c_int32, _i32p, C.c_int32, _i32p,
                                               C.c_int32, C.c_void_p, C.c_int32]
        L.orc_from_vector.argtypes = [MP, _i32p, _i32p, _f64p, C.c_int64]
        L.orc_at.argtypes = [MP, C.c_int32, C.c_int32]
        L.orc_at.restype = C.c_double
        L.orc_gauss_seidel.argtypes = [MP, _f64p, C.c_void_p, C.c_double, C.c_int, _f64p,
                                       C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.orc_conjugate_gradient.argtypes = [MP, _f64p, C.c_void_p, C.c_double, C.c_int, _f64p, C.POINTER(C.c_int)]
        L.orc_conjugate_gradient_jacobi.argtypes = [MP, _f64p, C.c_double, C.c_int, _f64p, C.POINTER(C.c_int)]
        L.orc_conjugate_gradient_jacobi.restype = C.c_int
        L.orc_apply_to_vector.argtypes = [MP, _f64p, _f64p]
        L.orc_apply_to_vector.restype = None
        L.orc_rel_residual.argtypes = [MP, _f64p, _f64p]
        L.orc_rel_residual.restype = C.c_double
        for name in ("orc_manhatton_dist", "orc_dot_prod"):
            getattr(L, name).argtypes = [_f64p, _f64p, C.c_int64]
            getattr(L, name).restype = C.c_double
        L.orc_veclen2.argtypes = [_f64p, C.c_int64]
        L.orc_veclen2.restype = C.c_double
        L.orc_poisson_csr.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_poisson_csr.restype = C.c_int64
        L.orc_poisson_csr_band.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_poisson_csr_band.restype = C.c_int64
        L.orc_poisson_csr_band64.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_poisson_csr_band64.restype = C.c_int64
        L.orc_poisson_row_starts.argtypes = [C.c_int32, C.c_int32, C.c_void_p]
        L.orc_poisson_row_starts.restype = None
        L.orc_poisson_apply_band.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        L.orc_poisson_apply_band.restype = None
        L.orc_poisson_rhs.argtypes = [C.c_int32, C.c_int32, _f32p, _f32p, C.c_int64, C.c_int32,
                                      C.c_int32, C.c_int32, _f64p]
        L.orc_poisson_rhs.restype = None
        L.orc_gradient_field.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.c_int64,
                                         _u8p, C.c_int64, _f32p, _f32p, C.c_int64]
        L.orc_gradient_field.restype = None
        L.orc_clamp_store_u8.argtypes = [C.c_int32, C.c_int32, _f64p, _u8p, C.c_int64, C.c_int32,
                                         C.c_int32]
        L.orc_clamp_store_u8.restype = None
        L.orc_composite_init.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.c_int64,
                                         _u8p, C.c_int64, C.c_int32, _f64p]
        L.orc_composite_init.restype = None
        L.orc_permute_csr.argtypes = [C.c_int32, _f64p, _i32p, _i32p, _i32p, _f64p, _i32p, _i32p]

    # ---- ingest ------------------------------------------------------------------------
    def from_eigen_row_major(self, values, row_offset, col_offset, n_rows: int, n_cols: int,
                             non_zeros=None, n_values: Optional[int] = None) -> OracleMatrix:
        values = _f64(values)
        col_offset = _i32(col_offset)
        row_offset = _i32(row_offset)
        nz = None if non_zeros is None else _i32(non_zeros)
        nv = len(values) if n_values is None else n_values
        m = _Matrix()
        rc = self.lib.orc_from_eigen_row_major(C.byref(m), values, nv, row_offset, n_rows,
                                               col_offset, n_cols, _opt(nz),
                                               0 if nz is None else n_rows)
        if rc != 0:
            raise MemoryError("orc_from_eigen_row_major failed")
        return OracleMatrix(self.lib, m)

    def from_csr(self, values, col_offset, row_offset) -> OracleMatrix:
        """Compressed CSR with n+1 row offsets -> the ConvertFromEigen hand-off (utils.cc:5-15)."""
        n = len(row_offset) - 1
        return self.from_eigen_row_major(values, row_offset, col_offset, n, n)

    def from_vector(self, rows, cols, vals) -> OracleMatrix:
        rows, cols, vals = _i32(rows), _i32(cols), _f64(vals)
        m = _Matrix()
        rc = self.lib.orc_from_vector(C.byref(m), rows, cols, vals, len(vals))
        if rc != 0:
            raise MemoryError("orc_from_vector failed")
        return OracleMatrix(self.lib, m)

    def from_dense(self, dense) -> OracleMatrix:
        """SparseMatrix::initialize(r, c, list) (sparse-matrix.h:332-347)."""
        d = np.asarray(dense, dtype=np.float64)
        r, c = np.meshgrid(np.arange(d.shape[0]), np.arange(d.shape[1]), indexing="ij")
        return self.from_vector(r.ravel(), c.ravel(), d.ravel())

    # ---- helpers -----------------------------------------------------------------------
    def manhatton_dist(self, a, b) -> float:
        a, b = _f64(a), _f64(b)
        return self.lib.orc_manhatton_dist(a, b, len(a))

    def veclen2(self, a) -> float:
        a = _f64(a)
        return self.lib.orc_veclen2(a, len(a))

    def dot_prod(self, a, b) -> float:
        a, b = _f64(a), _f64(b)
        return self.lib.orc_dot_prod(a, b, len(a))

    # ---- Poisson assembly --------------------------------------------------------------
    def poisson_csr(self, W: int, H: int):
        nnz = self.lib.orc_poisson_csr(W, H, None, None, None)
        values = np.empty(nnz, dtype=np.float64)
        cols = np.empty(nnz, dtype=np.int32)
        rowp = np.empty(W * H + 1, dtype=np.int32)
        self.lib.orc_poisson_csr(W, H, values.ctypes.data, cols.ctypes.data, rowp.ctypes.data)
        return values, cols, rowp

    @staticmethod
    def _bands(H: int, threads: int):
        step = -(-H // max(1, threads))
        return [(y, min(H, y + step)) for y in range(0, H, step)]

    def poisson_csr_threaded(self, W: int, H: int, threads: int = 8, index_dtype=np.int32):
        """poisson_csr built by `threads` host threads (row bands; ctypes releases the GIL) into arrays each band
        touches first itself — same bytes as poisson_csr; index_dtype np.int64: the arrays of a 64-bit IndexType."""
        from concurrent.futures import ThreadPoolExecutor
        starts = np.empty(H + 1, dtype=np.int64)
        self.lib.orc_poisson_row_starts(W, H, starts.ctypes.data)
        nnz = int(starts[H])
        if index_dtype == np.int32 and nnz > 2**31 - 1:
            raise ValueError("more stored entries than a 32-bit index holds")
        values = np.empty(nnz, dtype=np.float64)
        cols = np.empty(nnz, dtype=index_dtype)
        rowp = np.empty(W * H + 1, dtype=index_dtype)
        vp, cp, rp = values.ctypes.data, cols.ctypes.data, rowp.ctypes.data
        fill = self.lib.orc_poisson_csr_band if index_dtype == np.int32 else self.lib.orc_poisson_csr_band64
        with ThreadPoolExecutor(max(1, threads)) as ex:
            list(ex.map(lambda b: fill(W, H, b[0], b[1], int(starts[b[0]]), vp, cp, rp), self._bands(H, threads)))
        return values, cols, rowp

    def poisson_apply_threaded(self, W: int, H: int, v, threads: int = 8) -> np.ndarray:
        """A v of the W x H Poisson matrix in applyToVector's order, row bands on `threads` host threads."""
        from concurrent.futures import ThreadPoolExecutor
        v = _f64(v)
        out = np.empty(W * H, dtype=np.float64)
        ip, op = v.ctypes.data, out.ctypes.data
        with ThreadPoolExecutor(max(1, threads)) as ex:
            list(ex.map(lambda b: self.lib.orc_poisson_apply_band(W, H, b[0], b[1], ip, op), self._bands(H, threads)))
        return out

    def poisson_matrix(self, W: int, H: int) -> OracleMatrix:
        v, c, r = self.poisson_csr(W, H)
        return self.from_csr(v, c, r)

    def poisson_rhs(self, gx, gy, channel: int, constraint: int) -> np.ndarray:
        gx = np.ascontiguousarray(gx, dtype=np.float32)
        gy = np.ascontiguousarray(gy, dtype=np.float32)
        H, W, ch = gx.shape
        out = np.empty(W * H, dtype=np.float64)
        self.lib.orc_poisson_rhs(W, H, gx, gy, gx.strides[0], ch, channel, constraint, out)
        return out

    def gradient_field(self, images: Sequence[np.ndarray], label: np.ndarray):
        imgs = [np.ascontiguousarray(i, dtype=np.uint8) for i in images]
        label = np.ascontiguousarray(label, dtype=np.uint8)
        H, W = label.shape
        gx = np.zeros((H, W, 3), dtype=np.float32)
        gy = np.zeros((H, W, 3), dtype=np.float32)
        ptrs = (C.c_void_p * len(imgs))(*[i.ctypes.data for i in imgs])
        self.lib.orc_gradient_field(W, H, ptrs, imgs[0].strides[0], label, label.strides[0],
                                    gx, gy, gx.strides[0])
        return gx, gy

    def clamp_store_u8(self, sol, out: np.ndarray, channel: int) -> None:
        H, W, ch = out.shape
        self.lib.orc_clamp_store_u8(W, H, _f64(sol), out, out.strides[0], ch, channel)

    def composite_init(self, images: Sequence[np.ndarray], label: np.ndarray, channel: int):
        imgs = [np.ascontiguousarray(i, dtype=np.uint8) for i in images]
        label = np.ascontiguousarray(label, dtype=np.uint8)
        H, W = label.shape
        out = np.empty(W * H, dtype=np.float64)
        ptrs = (C.c_void_p * len(imgs))(*[i.ctypes.data for i in imgs])
        self.lib.orc_composite_init(W, H, ptrs, imgs[0].strides[0], label, label.strides[0],
                                    channel, out)
        return out

    # ---- colour-major permutation ------------------------------------------------------
    def permute_csr(self, values, col_offset, row_offset, perm):
        values, col_offset, row_offset, perm = _f64(values), _i32(col_offset), _i32(row_offset), _i32(perm)
        n = len(row_offset) - 1
        pv = np.empty_like(values)
        pc = np.empty_like(col_offset)
        pr = np.empty_like(row_offset)
        rc = self.lib.orc_permute_csr(n, values, col_offset, row_offset, perm, pv, pc, pr)
        if rc != 0:
            raise MemoryError("orc_permute_csr failed")
        return pv, pc, pr

    def multicolour_gauss_seidel(self, values, col_offset, row_offset, colour, b,
                                 epsilon: float = 0.0, max_iteration: int = 1000, x0=None):
        """Reference gaussSeidel on P A P^T with rows grouped colour by colour:
        its index-order sweep is then exactly the multi-colour (red-black) sweep.
        Returns (x in the ORIGINAL ordering, iterations, last L1 step)."""
        colour = np.asarray(colour)
        perm = np.argsort(colour, kind="stable").astype(np.int32)   # perm[new] = old
        pv, pc, pr = self.permute_csr(values, col_offset, row_offset, perm)
        m = self.from_csr(pv, pc, pr)
        bp = _f64(b)[perm]
        x0p = None if x0 is None else _f64(x0)[perm]
        xp, it, eps = m.gauss_seidel(bp, epsilon, max_iteration, x0p)
        x = np.empty_like(xp)
        x[perm] = xp
        return x, it, eps


def grid_colour(W: int, H: int) -> np.ndarray:
    """(x+y)&1 per pixel in raster order; colour 0 ("red") contains pixel 0."""
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    return ((xx + yy) & 1).astype(np.int32).ravel()


class Ref:
    """The compiled, unmodified reference headers (oracle/_ref).  Exists only where
    /root/reference was present at build time."""

    def __init__(self):
        if not (os.path.exists(REF_PROJECT_SO) and os.path.exists(REF_LAB3_SO)):
            raise FileNotFoundError("oracle/_ref not built (needs /root/reference); run make -C oracle")
        P = C.CDLL(REF_PROJECT_SO)
        L3 = C.CDLL(REF_LAB3_SO)
        self.P, self.L3 = P, L3
        eig = [_f64p, C.c_int, _i32p, C.c_int, _i32p, C.c_int, C.c_void_p]
        P.ref_gs_eigen.argtypes = eig + [_f64p, C.c_double, C.c_int, _f64p]
        P.ref_spmv_eigen.argtypes = eig + [_f64p, _f64p]
        P.ref_dense_eigen.argtypes = eig + [_f64p, _i32p]
        P.ref_cg_eigen.argtypes = eig + [_f64p, C.c_double, C.c_int, C.c_void_p, _f64p]
        P.ref_cg_jacobi.argtypes = eig + [_f64p, C.c_double, C.c_int, _f64p]
        P.ref_vector_insert_scenario.argtypes = [_i32p, _i32p, _f64p, C.c_int, _i32p, _i32p, _f64p,
                                                 C.c_int, C.c_int, C.c_int, _f64p]
        P.ref_gs_vector.argtypes = [_i32p, _i32p, _f64p, C.c_int, _f64p, C.c_int, C.c_double,
                                    C.c_int, _f64p]
        for name in ("ref_manhatton_dist", "ref_dot_prod"):
            getattr(P, name).argtypes = [_f64p, _f64p, C.c_int]
            getattr(P, name).restype = C.c_double
        P.ref_veclen2.argtypes = [_f64p, C.c_int]
        P.ref_veclen2.restype = C.c_double
        P.ref_gs_eigen_timed.argtypes = [_f64p, C.c_int, _i32p, C.c_int, _i32p, C.c_int, _f64p,
                                         C.c_int, C.c_void_p]
        P.ref_gs_eigen_timed.restype = C.c_double
        if hasattr(P, "ref_gs_eigen_timed_phases_i64"):
            P.ref_gs_eigen_timed_phases_i64.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                                        C.c_int, C.POINTER(C.c_double), C.c_void_p]
            P.ref_gs_eigen_timed_phases_i64.restype = C.c_double
        if hasattr(P, "ref_gs_eigen_timed_phases"):
            P.ref_gs_eigen_timed_phases.argtypes = [_f64p, C.c_int, _i32p, C.c_int, _i32p, C.c_int, _f64p,
                                                    C.c_int, C.POINTER(C.c_double)]
            P.ref_gs_eigen_timed_phases.restype = C.c_double
        L3.ref_lab3_known_answer.argtypes = [C.c_double, C.c_int, _f64p, C.c_void_p]
        L3.ref_lab3_int_insert_scenario.argtypes = [_i32p, _i32p, _i32p, C.c_int, _i32p, _i32p,
                                                    _i32p, C.c_int, C.c_int, C.c_int, _i32p]
        L3.ref_lab3_gs_vector_int.argtypes = [_i32p, _i32p, _i32p, C.c_int, _f64p, C.c_int,
                                              C.c_double, C.c_int, _f64p]
        L3.ref_lab3_gs_vector_double.argtypes = [_i32p, _i32p, _f64p, C.c_int, _f64p, C.c_int,
                                                 C.c_double, C.c_int, _f64p]
        L3.ref_lab3_spmv_vector_double.argtypes = [_i32p, _i32p, _f64p, C.c_int, _f64p, C.c_int, _f64p]

    # project header ----------------------------------------------------------------------
    def _eig(self, values, col_offset, row_offset, n_rows, n_cols, non_zeros):
        values, col_offset, row_offset = _f64(values), _i32(col_offset), _i32(row_offset)
        nz = None if non_zeros is None else _i32(non_zeros)
        return (values, len(values), row_offset, n_rows, col_offset, n_cols, _opt(nz)), (values, col_offset, row_offset, nz)

    def gs_csr(self, values, col_offset, row_offset, b, epsilon=1e-6, max_iteration=1000,
               non_zeros=None) -> np.ndarray:
        n = len(row_offset) - 1 if non_zeros is None else len(non_zeros)
        args, keep = self._eig(values, col_offset, row_offset, n, n, non_zeros)
        x = np.empty(n, dtype=np.float64)
        self.P.ref_gs_eigen(*args, _f64(b), epsilon, max_iteration, x)
        return x

    def spmv_csr(self, values, col_offset, row_offset, v, non_zeros=None) -> np.ndarray:
        n = len(row_offset) - 1 if non_zeros is None else len(non_zeros)
        args, keep = self._eig(values, col_offset, row_offset, n, n, non_zeros)
        out = np.empty(n, dtype=np.float64)
        self.P.ref_spmv_eigen(*args, _f64(v), out)
        return out

    def dense_eigen(self, values, col_offset, row_offset, n_rows, n_cols, non_zeros=None):
        args, keep = self._eig(values, col_offset, row_offset, n_rows, n_cols, non_zeros)
        d = np.empty((n_rows, n_cols), dtype=np.float64)
        rc = np.zeros(2, dtype=np.int32)
        self.P.ref_dense_eigen(*args, d, rc)
        return d, int(rc[0]), int(rc[1])

    def cg_csr(self, values, col_offset, row_offset, b, epsilon=1e-16, max_iteration=1000, init=None):
        n = len(row_offset) - 1
        args, keep = self._eig(values, col_offset, row_offset, n, n, None)
        x = np.empty(n, dtype=np.float64)
        ia = None if init is None else _f64(init)
        self.P.ref_cg_eigen(*args, _f64(b), epsilon, max_iteration, _opt(ia), x)
        return x

    def cg_jacobi_csr(self, values, col_offset, row_offset, b, epsilon=1e-16, max_iteration=180):
        n = len(row_offset) - 1
        args, keep = self._eig(values, col_offset, row_offset, n, n, None)
        x = np.empty(n, dtype=np.float64)
        self.P.ref_cg_jacobi(*args, _f64(b), epsilon, max_iteration, x)
        return x

    def vector_insert_scenario(self, rows, cols, vals, ops, n_rows, n_cols):
        ops = list(ops)
        orow = _i32([o[1] for o in ops])
        ocol = _i32([o[2] for o in ops])
        oval = _f64([o[0] for o in ops])
        out = np.zeros((len(ops) + 1, n_rows, n_cols), dtype=np.float64)
        self.P.ref_vector_insert_scenario(_i32(rows), _i32(cols), _f64(vals), len(vals), orow, ocol,
                                          oval, len(ops), n_rows, n_cols, out)
        return out

    def gs_vector(self, rows, cols, vals, b, epsilon=1e-6, max_iteration=1000):
        b = _f64(b)
        x = np.empty(len(b), dtype=np.float64)
        self.P.ref_gs_vector(_i32(rows), _i32(cols), _f64(vals), len(vals), b, len(b), epsilon,
                             max_iteration, x)
        return x

    def manhatton_dist(self, a, b):
        a, b = _f64(a), _f64(b)
        return self.P.ref_manhatton_dist(a, b, len(a))

    def veclen2(self, a):
        a = _f64(a)
        return self.P.ref_veclen2(a, len(a))

    def dot_prod(self, a, b):
        a, b = _f64(a), _f64(b)
        return self.P.ref_dot_prod(a, b, len(a))

    def gs_csr_timed(self, values, col_offset, row_offset, b, max_iteration):
        """Seconds spent inside the reference gaussSeidel (epsilon = 0)."""
        n = len(row_offset) - 1
        values, col_offset, row_offset = _f64(values), _i32(col_offset), _i32(row_offset)
        return self.P.ref_gs_eigen_timed(values, len(values), row_offset, n, col_offset, n,
                                         _f64(b), max_iteration, None)

    def gs_csr_timed_phases_i64(self, values, col_offset, row_offset, b, max_iteration, x_out=None):
        """The same through SparseMatrix<double, int64_t> (col_offset, row_offset: int64 arrays) — the header's own
        IndexType parameter, for systems whose entry positions overflow the default int (see orc_poisson_csr_band64)."""
        n = len(row_offset) - 1
        values, b = _f64(values), _f64(b)
        col_offset = np.ascontiguousarray(col_offset, dtype=np.int64)
        row_offset = np.ascontiguousarray(row_offset, dtype=np.int64)
        ingest = C.c_double(0.0)
        secs = self.P.ref_gs_eigen_timed_phases_i64(values.ctypes.data, len(values), row_offset.ctypes.data, n, col_offset.ctypes.data, n,
                                                    b.ctypes.data, max_iteration, C.byref(ingest),
                                                    None if x_out is None else x_out.ctypes.data)
        return ingest.value, secs

    def gs_csr_timed_phases(self, values, col_offset, row_offset, b, max_iteration):
        """(seconds of the ingest, seconds inside the reference gaussSeidel) — epsilon = 0."""
        n = len(row_offset) - 1
        values, col_offset, row_offset = _f64(values), _i32(col_offset), _i32(row_offset)
        ingest = C.c_double(0.0)
        secs = self.P.ref_gs_eigen_timed_phases(values, len(values), row_offset, n, col_offset, n,
                                                _f64(b), max_iteration, C.byref(ingest))
        return ingest.value, secs

    # lab3 header -------------------------------------------------------------------------
    def lab3_known_answer(self, epsilon=1e-6, max_iteration=1000, with_cg=False):
        x = np.empty(4)
        xc = np.empty(4) if with_cg else None
        self.L3.ref_lab3_known_answer(epsilon, max_iteration, x, _opt(xc))
        return (x, xc) if with_cg else x

    def lab3_int_insert_scenario(self, rows, cols, vals, ops, n_rows, n_cols):
        ops = list(ops)
        out = np.zeros((len(ops) + 1, n_rows, n_cols), dtype=np.int32)
        self.L3.ref_lab3_int_insert_scenario(_i32(rows), _i32(cols), _i32(vals), len(vals),
                                             _i32([o[1] for o in ops]), _i32([o[2] for o in ops]),
                                             _i32([o[0] for o in ops]), len(ops), n_rows, n_cols, out)
        return out

    def lab3_modify_bench(self, rows, cols, vals, mods, size):
        """The reference's own insert(0) benchmark on SparseMatrix<int> (main6.cc:92-187): milliseconds for
        the ingest and for the edits, and the sum of the dense scan afterwards."""
        if not hasattr(self.L3, "ref_lab3_modify_bench"):
            raise OSError("oracle/_ref was built before ref_lab3_modify_bench existed")
        mods = np.asarray(mods)
        ms_i, ms_m, chk = C.c_double(), C.c_double(), C.c_longlong()
        self.L3.ref_lab3_modify_bench.argtypes = [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 2 + [C.c_int, C.c_int,
                                                  C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_longlong)]
        rows, cols, vals = _i32(rows), _i32(cols), _i32(vals)
        orow, ocol = _i32(mods[:, 0]), _i32(mods[:, 1])
        self.L3.ref_lab3_modify_bench(rows.ctypes.data, cols.ctypes.data, vals.ctypes.data, len(vals), orow.ctypes.data,
                                      ocol.ctypes.data, len(orow), size, C.byref(ms_i), C.byref(ms_m), C.byref(chk))
        return {"init_ms": ms_i.value, "modify_ms": ms_m.value, "dense_checksum": chk.value, "cores": 1,
                "kind": "reference (lab3 SparseMatrix<int>, compiled header)"}

    def lab3_gs_vector_double(self, rows, cols, vals, b, epsilon=1e-6, max_iteration=1000):
        b = _f64(b)
        x = np.empty(len(b))
        self.L3.ref_lab3_gs_vector_double(_i32(rows), _i32(cols), _f64(vals), len(vals), b, len(b),
                                          epsilon, max_iteration, x)
        return x
