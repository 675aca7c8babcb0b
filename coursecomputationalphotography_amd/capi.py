"""ctypes binding of libccp_gs.so (the C ABI declared in include/ccp_gs.h).

This is plumbing for tests, bench.py and the multi-GPU driver: the product is the shared
library.  There is no CPU fallback — if the library is missing, or no HIP device is usable,
the calls raise (``CcpError``) instead of computing anything on the host.
"""
from __future__ import annotations

import atexit
import ctypes as C
import os
import sys
import weakref
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CCP_GS_LIB: developer override to A/B two builds of the library in one session
LIB_PATH = os.environ.get("CCP_GS_LIB") or os.path.join(_HERE, "lib", "libccp_gs.so")

CCP_OK = 0
GRID_DIRICHLET_MASK = 1
ORDER_LEXICOGRAPHIC = 0
ORDER_MULTICOLOUR = 1

# every symbol include/ccp_gs.h declares (tests check the library exports all of them)
ABI_SYMBOLS = (
    "ccp_status_string", "ccp_abi_version", "ccp_device_count",
    "ccp_csr_create", "ccp_csr_destroy", "ccp_csr_upload", "ccp_csr_upload_rows", "ccp_csr_rows_info", "ccp_csr_set_colouring", "ccp_csr_get_colouring", "ccp_csr_insert", "ccp_csr_insert_many", "ccp_csr_edit_stats", "ccp_csr_device_footprint", "ccp_csr_last_path", "ccp_csr_embed_region_host",
    "ccp_csr_gauss_seidel", "ccp_csr_conjugate_gradient", "ccp_csr_conjugate_gradient_jacobi", "ccp_csr_apply_to_vector", "ccp_csr_residual_norm2",
    "ccp_grid_create", "ccp_grid_destroy", "ccp_grid_get_layout", "ccp_grid_set_stream",
    "ccp_grid_synchronize", "ccp_grid_set_b_host", "ccp_grid_set_x_host", "ccp_grid_get_x_host",
    "ccp_grid_get_b_host", "ccp_grid_set_mask_host", "ccp_grid_fill_x", "ccp_grid_b_from_x", "ccp_grid_randomize_x",
    "ccp_grid_sweep", "ccp_grid_sweep_edges_first", "ccp_grid_stream_wait_edges", "ccp_grid_tune", "ccp_grid_set_fused", "ccp_grid_set_tiling", "ccp_grid_get_tiling", "ccp_grid_sweep_l1", "ccp_grid_halo_refreshed", "ccp_grid_gauss_seidel", "ccp_grid_gauss_seidel_lexicographic", "ccp_debug_lex_tickets", "ccp_grid_conjugate_gradient",
    "ccp_grid_residual_norm2", "ccp_grid_abs_sum", "ccp_grid_assemble_rhs", "ccp_grid_assemble_from_images", "ccp_grid_store_u8",
    "ccp_grid_set_x_u8", "ccp_grid_last_timing", "ccp_grid_region_begin", "ccp_grid_region_end",
    "ccp_comm_probe", "ccp_comm_unique_id", "ccp_comm_create", "ccp_comm_destroy", "ccp_comm_info", "ccp_comm_all_reduce_sum", "ccp_comm_all_reduce_max",
    "ccp_grid_attach_comm", "ccp_grid_set_overlap", "ccp_grid_exchange_halos", "ccp_grid_sweep_rowblocked",
    "ccp_grid_gauss_seidel_rowblocked", "ccp_grid_conjugate_gradient_rowblocked", "ccp_grid_residual_norm2_global", "ccp_grid_comm_stats",
)


class CcpError(RuntimeError):
    def __init__(self, status: int, what: str):
        self.status = status
        super().__init__(f"{what}: status {status} ({status_string(status)})")


class Report(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("converged", C.c_int32),
                ("last_l1_step", C.c_double), ("seconds", C.c_double)]


class GridDesc(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("channels", C.c_int32),
                ("row_begin", C.c_int32), ("row_count", C.c_int32), ("ghost", C.c_int32),
                ("device", C.c_int32), ("flags", C.c_int32)]


class GridLayout(C.Structure):
    _fields_ = [("x_dev", C.c_void_p), ("b_dev", C.c_void_p), ("pitch", C.c_int64),
                ("local_rows", C.c_int32), ("ghost_top", C.c_int32), ("ghost_bottom", C.c_int32),
                ("channels", C.c_int32)]


_lib: Optional[C.CDLL] = None

# Live handles, closed in dependency order (grids before the communicators they are attached to, both
# before the HIP / RCCL runtimes unload) if the program exits without closing them itself.
_live_grids: "weakref.WeakSet" = weakref.WeakSet()
_live_comms: "weakref.WeakSet" = weakref.WeakSet()


def _close_all_at_exit() -> None:
    for pool in (_live_grids, _live_comms):
        for h in list(pool):
            try:
                h.close()
            except Exception:
                pass


atexit.register(_close_all_at_exit)


def _share_rccl_with_torch() -> None:
    """The same for RCCL, which libccp_gs.so binds at run time by soname (ccp_comm.hpp): inside a Python
    process the HIP runtime is torch's, so the collective library must be the one torch ships with it.
    Importing torch (where it is installed) before the first communicator call loads that copy in torch's
    own order; the soname lookup inside the library then finds it.  (Round 2 saw a process that bound torch's
    librccl.so by hand and imported torch LATER abort at exit; root-caused in round 3 to RTLD_GLOBAL symbol
    interposition between librocm_smi64 and libamd_smi and fixed in csrc/ccp_comm.hip — that order is a
    regression test now, tests/test_gpu_rccl.py — so this import is about sharing ONE runtime, not about the exit.)"""
    import importlib.util
    if "torch" in sys.modules or os.environ.get("CCP_GS_NO_TORCH_HIP") or os.environ.get("CCP_GS_RCCL_LIB"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is not None:
        try:
            import torch  # noqa: F401
        except Exception:
            pass


def _share_hip_runtime_with_torch() -> None:
    """PyTorch-ROCm bundles its own libamdhip64.so.7 (same soname as /opt/rocm's).  Whichever
    copy is loaded first serves the whole process; if ours pulled in /opt/rocm's first, a later
    `import torch` would run on a runtime it was not built for and report no GPU.  So when torch
    is installed but not imported yet, load ITS runtime first (no torch import, no GPU init)."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("CCP_GS_NO_TORCH_HIP"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


def load() -> C.CDLL:
    """dlopen libccp_gs.so; raises FileNotFoundError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    _share_hip_runtime_with_torch()
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} is missing: build it with `make -C coursecomputationalphotography_amd/csrc` "
            "(or __graft_entry__.build()).  There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    L.ccp_status_string.restype = C.c_char_p
    L.ccp_status_string.argtypes = [C.c_int]
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    L.ccp_csr_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.ccp_csr_destroy.argtypes = [vp]
    L.ccp_csr_upload.argtypes = [vp, i32, i32, i64, vp, vp, vp, vp]
    L.ccp_csr_upload_rows.argtypes = [vp, vp, i32, i32, i32, i64, vp, vp, vp, vp, vp, i32]
    L.ccp_csr_rows_info.argtypes = [vp] + [C.POINTER(i32)] * 5 + [C.POINTER(i64)] * 2
    L.ccp_csr_set_colouring.argtypes = [vp, vp, i32]
    L.ccp_csr_get_colouring.argtypes = [vp, vp, C.POINTER(i32)]
    L.ccp_csr_insert.argtypes = [vp, i32, i32, dbl]
    L.ccp_csr_insert_many.argtypes = [vp, i64, vp, vp, vp]
    L.ccp_csr_last_path.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i64)]
    L.ccp_csr_edit_stats.argtypes = [vp] + [C.POINTER(i64)] * 5
    L.ccp_csr_device_footprint.argtypes = [vp] + [C.POINTER(i64)] * 2
    L.ccp_csr_gauss_seidel.argtypes = [vp, vp, vp, vp, dbl, i32, i32, i32, C.POINTER(Report)]
    L.ccp_csr_conjugate_gradient.argtypes = [vp, vp, vp, vp, dbl, i32, C.POINTER(Report)]
    L.ccp_csr_conjugate_gradient_jacobi.argtypes = [vp, vp, vp, dbl, i32, C.POINTER(Report)]
    L.ccp_grid_conjugate_gradient.argtypes = [vp, dbl, i32, C.POINTER(Report)]
    L.ccp_csr_apply_to_vector.argtypes = [vp, vp, vp]
    L.ccp_csr_residual_norm2.argtypes = [vp, vp, vp, C.POINTER(dbl), C.POINTER(dbl)]
    L.ccp_grid_create.argtypes = [C.POINTER(GridDesc), C.POINTER(vp)]
    L.ccp_grid_destroy.argtypes = [vp]
    L.ccp_grid_get_layout.argtypes = [vp, C.POINTER(GridLayout)]
    L.ccp_grid_set_stream.argtypes = [vp, vp]
    L.ccp_grid_synchronize.argtypes = [vp]
    for name in ("ccp_grid_set_b_host", "ccp_grid_set_x_host", "ccp_grid_get_x_host", "ccp_grid_get_b_host"):
        getattr(L, name).argtypes = [vp, i32, vp, i32, i32]
    L.ccp_grid_set_mask_host.argtypes = [vp, vp, i64]
    L.ccp_grid_fill_x.argtypes = [vp, dbl]
    L.ccp_grid_b_from_x.argtypes = [vp]
    L.ccp_grid_randomize_x.argtypes = [vp, C.c_uint64, dbl, dbl]
    L.ccp_grid_sweep.argtypes = [vp, i32]
    L.ccp_grid_sweep_edges_first.argtypes = [vp, i32, i32]
    L.ccp_grid_stream_wait_edges.argtypes = [vp, vp]
    L.ccp_grid_sweep_l1.argtypes = [vp, vp]
    L.ccp_grid_tune.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(C.c_float)]
    L.ccp_grid_set_fused.argtypes = [vp, i32]
    L.ccp_grid_set_tiling.argtypes = [vp, i32, i32]
    L.ccp_grid_get_tiling.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.ccp_grid_halo_refreshed.argtypes = [vp]
    L.ccp_grid_gauss_seidel.argtypes = [vp, dbl, i32, i32, C.POINTER(Report)]
    L.ccp_grid_gauss_seidel_lexicographic.argtypes = [vp, dbl, i32, i32, C.POINTER(Report)]
    L.ccp_grid_residual_norm2.argtypes = [vp, vp]
    L.ccp_grid_abs_sum.argtypes = [vp, vp]
    L.ccp_grid_assemble_rhs.argtypes = [vp, vp, vp, i64, vp]
    L.ccp_grid_assemble_from_images.argtypes = [vp, vp, i32, i64, vp, i64, i32]
    L.ccp_grid_store_u8.argtypes = [vp, vp, i64]
    L.ccp_grid_set_x_u8.argtypes = [vp, vp, i64]
    L.ccp_grid_last_timing.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(i32)]
    L.ccp_grid_region_begin.argtypes = [vp]
    L.ccp_grid_region_end.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(i64), C.POINTER(i64)]
    L.ccp_comm_probe.argtypes = [i32]
    L.ccp_comm_unique_id.argtypes = [vp]
    L.ccp_comm_create.argtypes = [vp, i32, i32, i32, C.POINTER(vp)]
    L.ccp_comm_destroy.argtypes = [vp]
    L.ccp_comm_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.ccp_comm_all_reduce_sum.argtypes = [vp, vp, i32]
    L.ccp_comm_all_reduce_max.argtypes = [vp, vp, i32]
    L.ccp_grid_attach_comm.argtypes = [vp, vp]
    L.ccp_grid_set_overlap.argtypes = [vp, i32]
    L.ccp_grid_exchange_halos.argtypes = [vp]
    L.ccp_grid_sweep_rowblocked.argtypes = [vp, i32]
    L.ccp_grid_gauss_seidel_rowblocked.argtypes = [vp, dbl, i32, i32, C.POINTER(Report)]
    L.ccp_grid_conjugate_gradient_rowblocked.argtypes = [vp, dbl, i32, C.POINTER(Report)]
    L.ccp_grid_residual_norm2_global.argtypes = [vp, vp]
    L.ccp_grid_comm_stats.argtypes = [vp, C.POINTER(i64), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    _lib = L
    return L


def status_string(status: int) -> str:
    try:
        return load().ccp_status_string(status).decode()
    except Exception:
        return "?"


def check(status: int, what: str) -> None:
    if status != CCP_OK:
        raise CcpError(status, what)


def device_count() -> int:
    return int(load().ccp_device_count())


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


COMM_ID_BYTES = 128


def comm_probe(device: int = 0) -> None:
    """Raises CcpError unless this process can take part in a communicator on `device` (not collective)."""
    _share_rccl_with_torch()
    check(load().ccp_comm_probe(device), "ccp_comm_probe")


def comm_unique_id() -> bytes:
    """ncclGetUniqueId through the C ABI (rank 0 calls this and hands the bytes to the other ranks)."""
    _share_rccl_with_torch()
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    check(load().ccp_comm_unique_id(buf), "ccp_comm_unique_id")
    return bytes(buf)


class Comm:
    """One rank of an RCCL communicator behind the C ABI (ccp_comm_*).  Creation is collective."""

    def __init__(self, unique_id: bytes, rank: int, world: int, device: int = 0):
        _share_rccl_with_torch()
        self.L = load()
        self.h = C.c_void_p()
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError("unique id must be COMM_ID_BYTES long")
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
        check(self.L.ccp_comm_create(buf, rank, world, device, C.byref(self.h)), "ccp_comm_create")
        self.rank, self.world, self.device = rank, world, device
        _live_comms.add(self)

    def info(self):
        r, w, d, v = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        check(self.L.ccp_comm_info(self.h, C.byref(r), C.byref(w), C.byref(d), C.byref(v)), "ccp_comm_info")
        return {"rank": r.value, "world": w.value, "device": d.value, "rccl_version": v.value}

    def all_reduce_sum(self, values) -> np.ndarray:
        a = _f64(values).copy()
        check(self.L.ccp_comm_all_reduce_sum(self.h, _ptr(a), a.size), "ccp_comm_all_reduce_sum")
        return a

    def all_reduce_max(self, values) -> np.ndarray:
        a = _f64(values).copy()
        check(self.L.ccp_comm_all_reduce_max(self.h, _ptr(a), a.size), "ccp_comm_all_reduce_max")
        return a

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.ccp_comm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass


def embed_region_host(values, col_offset, row_offset, colour):
    """ccp_csr_embed_region_host: (recognised, W, H, x[n], y[n]) — the raster-region recognition alone, on the host."""
    L = load()
    values, col_offset, row_offset, colour = _f64(values), _i32(col_offset), _i32(row_offset), _i32(colour)
    n = len(row_offset) - 1
    ok, w, h = C.c_int32(), C.c_int32(), C.c_int32()
    x, y = np.zeros(max(n, 1), dtype=np.int32), np.zeros(max(n, 1), dtype=np.int32)
    L.ccp_csr_embed_region_host.argtypes = [C.c_int32] + [C.c_void_p] * 4 + [C.POINTER(C.c_int32)] * 3 + [C.c_void_p] * 2
    check(L.ccp_csr_embed_region_host(n, _ptr(row_offset), _ptr(col_offset), _ptr(values), _ptr(colour), C.byref(ok), C.byref(w),
                                      C.byref(h), _ptr(x), _ptr(y)), "ccp_csr_embed_region_host")
    return bool(ok.value), w.value, h.value, x[:n], y[:n]


class CsrMatrix:
    """Device-resident slack-CSR matrix (ccp_csr_*)."""

    def __init__(self, device: int = 0):
        self.L = load()
        self.h = C.c_void_p()
        check(self.L.ccp_csr_create(device, C.byref(self.h)), "ccp_csr_create")
        self.n_rows = self.n_cols = 0

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.ccp_csr_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    def upload(self, n_rows, n_cols, values, col_offset, row_begin, row_num_nze):
        values, col_offset = _f64(values), _i32(col_offset)
        row_begin, row_num_nze = _i32(row_begin), _i32(row_num_nze)
        check(self.L.ccp_csr_upload(self.h, n_rows, n_cols, len(values), _ptr(values), _ptr(col_offset),
                                    _ptr(row_begin), _ptr(row_num_nze)), "ccp_csr_upload")
        self.n_rows, self.n_cols = n_rows, n_cols
        return self

    def upload_rows(self, comm: "Comm", first_row, n_global, values, col_offset, row_begin, row_num_nze, colour, n_colours):
        """COLLECTIVE: this handle becomes the block of rows [first_row, first_row + len(row_begin)) of an n_global-row
        matrix distributed over comm's ranks (global column indices; colour: the owned rows of a proper colouring of the
        whole matrix).  Afterwards gauss_seidel / apply_to_vector / residual_norm2 take and return the block's own rows."""
        values, col_offset = _f64(values), _i32(col_offset)
        row_begin, row_num_nze, colour = _i32(row_begin), _i32(row_num_nze), _i32(colour)
        n_rows = len(row_begin)
        check(self.L.ccp_csr_upload_rows(self.h, comm.h, first_row, n_rows, n_global, len(values), _ptr(values), _ptr(col_offset),
                                         _ptr(row_begin), _ptr(row_num_nze), _ptr(colour), n_colours), "ccp_csr_upload_rows")
        self.n_rows = self.n_cols = n_rows
        return self

    def rows_info(self):
        a = [C.c_int32() for _ in range(5)]
        b = [C.c_int64() for _ in range(2)]
        check(self.L.ccp_csr_rows_info(self.h, *[C.byref(t) for t in a + b]), "ccp_csr_rows_info")
        return dict(zip(("first_row", "n_rows", "n_ghost", "n_peers", "edge_slices", "values_sent", "exchanges"), (t.value for t in a + b)))

    def upload_compressed(self, values, col_offset, row_offset, n_cols=None):
        """Compressed CSR (n+1 offsets) -> the slack arrays with zero slack."""
        row_offset = _i32(row_offset)
        n = len(row_offset) - 1
        return self.upload(n, n if n_cols is None else n_cols, values, col_offset, row_offset[:-1],
                           np.diff(row_offset))

    def set_colouring(self, colour, n_colours=None):
        if colour is None:
            check(self.L.ccp_csr_set_colouring(self.h, None, 0), "ccp_csr_set_colouring")
            return self
        colour = _i32(colour)
        nc = int(colour.max()) + 1 if n_colours is None else n_colours
        check(self.L.ccp_csr_set_colouring(self.h, _ptr(colour), nc), "ccp_csr_set_colouring")
        return self

    def insert(self, val: float, row: int, col: int):
        """SparseMatrix::insert(val, row, col) on the uploaded matrix (applied on the device incrementally)."""
        check(self.L.ccp_csr_insert(self.h, row, col, float(val)), "ccp_csr_insert")

    def last_path(self) -> str:
        """Kernels of the last gauss_seidel: "sliced ELL", "Poisson grid WxH" or "region grid WxH" (canvas)."""
        p, w, h, n = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
        check(self.L.ccp_csr_last_path(self.h, C.byref(p), C.byref(w), C.byref(h), C.byref(n)), "ccp_csr_last_path")
        self.last_sweep_launches = n.value
        return {0: "sliced ELL", 1: f"Poisson grid {w.value}x{h.value}", 2: f"region grid {w.value}x{h.value}"}[p.value]

    def insert_many(self, vals, rows, cols):
        vals, rows, cols = _f64(vals), _i32(rows), _i32(cols)
        check(self.L.ccp_csr_insert_many(self.h, len(vals), _ptr(rows), _ptr(cols), _ptr(vals)), "ccp_csr_insert_many")

    def edit_stats(self):
        v = [C.c_int64() for _ in range(5)]
        check(self.L.ccp_csr_edit_stats(self.h, *[C.byref(t) for t in v]), "ccp_csr_edit_stats")
        return dict(zip(("edits", "image_uploads", "rows_patched", "slices_relocated", "image_rebuilds"), (t.value for t in v)))

    def device_footprint(self):
        """(device bytes held for a copy of the stored matrix, uploads whose background copy was skipped for lack of room)."""
        v = [C.c_int64() for _ in range(2)]
        check(self.L.ccp_csr_device_footprint(self.h, *[C.byref(t) for t in v]), "ccp_csr_device_footprint")
        return v[0].value, v[1].value

    def get_colouring(self):
        """(colour[n_rows], n_colours) the multi-colour sweep uses (caller's or the library's greedy one)."""
        colour = np.empty(max(self.n_rows, 1), dtype=np.int32)
        nc = C.c_int32()
        check(self.L.ccp_csr_get_colouring(self.h, _ptr(colour), C.byref(nc)), "ccp_csr_get_colouring")
        return colour[:self.n_rows], nc.value

    def gauss_seidel(self, b, epsilon=1e-6, max_iteration=1000, x0=None, check_every=1,
                     ordering=ORDER_MULTICOLOUR, out=None):
        """out: a float64 array of n_cols entries to receive x (a buffer that is used again travels at the PCIe rate: the
        HIP runtime registers a host buffer at its first use — 54 GB/s against 10-17 GB/s into fresh pages)."""
        b = _f64(b)
        x0a = None if x0 is None else _f64(x0)
        if out is not None and (out.dtype != np.float64 or out.size != self.n_cols or not out.flags.c_contiguous):
            raise ValueError("out must be a contiguous float64 array of n_cols entries")
        x = np.empty(self.n_cols, dtype=np.float64) if out is None else out
        rep = Report()
        check(self.L.ccp_csr_gauss_seidel(self.h, _ptr(b), _ptr(x0a), _ptr(x), epsilon, max_iteration,
                                          check_every, ordering, C.byref(rep)), "ccp_csr_gauss_seidel")
        return x, rep

    def conjugate_gradient(self, b, epsilon=1e-16, max_iteration=1000, init=None):
        b = _f64(b)
        ia = None if init is None else _f64(init)
        x = np.empty(self.n_cols, dtype=np.float64)
        rep = Report()
        check(self.L.ccp_csr_conjugate_gradient(self.h, _ptr(b), _ptr(ia), _ptr(x), epsilon, max_iteration,
                                                C.byref(rep)), "ccp_csr_conjugate_gradient")
        return x, rep

    def conjugate_gradient_jacobi(self, b, epsilon=1e-16, max_iteration=180):
        """SparseMatrix::conjugateGradientEigen: Jacobi-preconditioned, from x0 = 0."""
        b = _f64(b)
        x = np.empty(self.n_cols, dtype=np.float64)
        rep = Report()
        check(self.L.ccp_csr_conjugate_gradient_jacobi(self.h, _ptr(b), _ptr(x), epsilon, max_iteration, C.byref(rep)),
              "ccp_csr_conjugate_gradient_jacobi")
        return x, rep

    def apply_to_vector(self, v):
        v = _f64(v)
        out = np.empty(self.n_rows, dtype=np.float64)
        check(self.L.ccp_csr_apply_to_vector(self.h, _ptr(v), _ptr(out)), "ccp_csr_apply_to_vector")
        return out

    def residual_norm2(self, b, x):
        b, x = _f64(b), _f64(x)
        rr, bb = C.c_double(), C.c_double()
        check(self.L.ccp_csr_residual_norm2(self.h, _ptr(b), _ptr(x), C.byref(rr), C.byref(bb)),
              "ccp_csr_residual_norm2")
        return rr.value, bb.value


class Grid:
    """Structured Poisson grid block (ccp_grid_*)."""

    def __init__(self, width, height, channels=1, row_begin=0, row_count=None, ghost=0, device=0, mask=None):
        """mask: H x W array (non-zero = unknown) makes this a Dirichlet-mask grid (CCP_GRID_DIRICHLET_MASK)."""
        self.L = load()
        self.h = C.c_void_p()
        row_count = height if row_count is None else row_count
        self.desc = GridDesc(width, height, channels, row_begin, row_count, ghost, device, 0 if mask is None else GRID_DIRICHLET_MASK)
        check(self.L.ccp_grid_create(C.byref(self.desc), C.byref(self.h)), "ccp_grid_create")
        self.layout = GridLayout()
        check(self.L.ccp_grid_get_layout(self.h, C.byref(self.layout)), "ccp_grid_get_layout")
        self.W, self.H, self.C = width, height, channels
        self.row_begin, self.row_count = row_begin, row_count
        self.first_local_row = row_begin - self.layout.ghost_top       # image row of local row 0
        self.local_rows = self.layout.local_rows
        self._comm = None
        self.stream_handle = 0                                   # the null stream until set_stream says otherwise
        _live_grids.add(self)
        if mask is not None:
            self.set_mask(mask)

    def set_mask(self, mask):
        mask = np.ascontiguousarray(np.asarray(mask) != 0, dtype=np.uint8)
        if mask.shape != (self.H, self.W):
            raise ValueError("mask must be H x W")
        check(self.L.ccp_grid_set_mask_host(self.h, _ptr(mask), mask.strides[0]), "ccp_grid_set_mask_host")

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.ccp_grid_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_handle: int):
        check(self.L.ccp_grid_set_stream(self.h, C.c_void_p(stream_handle)), "ccp_grid_set_stream")
        self.stream_handle = int(stream_handle or 0)

    def synchronize(self):
        check(self.L.ccp_grid_synchronize(self.h), "ccp_grid_synchronize")

    def _rows(self, first_row, n_rows):
        if first_row is None:
            first_row = self.first_local_row
        if n_rows is None:
            n_rows = self.local_rows - (first_row - self.first_local_row)
        return first_row, n_rows

    def set_b(self, rows, channel=0, first_row=None):
        rows = _f64(rows).reshape(-1, self.W)
        first_row, _ = self._rows(first_row, None)
        check(self.L.ccp_grid_set_b_host(self.h, channel, _ptr(rows), first_row, rows.shape[0]), "ccp_grid_set_b_host")

    def set_x(self, rows, channel=0, first_row=None):
        rows = _f64(rows).reshape(-1, self.W)
        first_row, _ = self._rows(first_row, None)
        check(self.L.ccp_grid_set_x_host(self.h, channel, _ptr(rows), first_row, rows.shape[0]), "ccp_grid_set_x_host")

    def get_x(self, channel=0, first_row=None, n_rows=None) -> np.ndarray:
        first_row, n_rows = self._rows(first_row, n_rows)
        out = np.empty((n_rows, self.W), dtype=np.float64)
        check(self.L.ccp_grid_get_x_host(self.h, channel, _ptr(out), first_row, n_rows), "ccp_grid_get_x_host")
        return out

    def get_b(self, channel=0, first_row=None, n_rows=None) -> np.ndarray:
        first_row, n_rows = self._rows(first_row, n_rows)
        out = np.empty((n_rows, self.W), dtype=np.float64)
        check(self.L.ccp_grid_get_b_host(self.h, channel, _ptr(out), first_row, n_rows), "ccp_grid_get_b_host")
        return out

    def get_x_owned(self, channel=0) -> np.ndarray:
        return self.get_x(channel, self.row_begin, self.row_count)

    def fill_x(self, value=1.0):
        check(self.L.ccp_grid_fill_x(self.h, value), "ccp_grid_fill_x")

    def randomize_x(self, seed, lo=0.0, hi=255.0):
        check(self.L.ccp_grid_randomize_x(self.h, seed, lo, hi), "ccp_grid_randomize_x")

    def b_from_x(self):
        check(self.L.ccp_grid_b_from_x(self.h), "ccp_grid_b_from_x")

    def sweep(self, iterations):
        check(self.L.ccp_grid_sweep(self.h, iterations), "ccp_grid_sweep")

    def sweep_edges_first(self, iterations, edge_rows):
        """sweep(), with the rows a neighbour block needs finished first in the last pass."""
        check(self.L.ccp_grid_sweep_edges_first(self.h, iterations, edge_rows), "ccp_grid_sweep_edges_first")

    def stream_wait_edges(self, stream_handle: int):
        check(self.L.ccp_grid_stream_wait_edges(self.h, C.c_void_p(stream_handle)), "ccp_grid_stream_wait_edges")

    def tune(self, max_t: int = 8):
        """Time the (depth, rows-per-chunk) candidates on this shape; returns (T, rows, ms/iter)."""
        t, r, ms = C.c_int32(), C.c_int32(), C.c_float()
        check(self.L.ccp_grid_tune(self.h, max_t, C.byref(t), C.byref(r), C.byref(ms)), "ccp_grid_tune")
        return t.value, r.value, ms.value

    def set_fused(self, on: bool):
        """Temporally blocked pass (True, default) or the in-place half-sweep kernels (False)."""
        check(self.L.ccp_grid_set_fused(self.h, 1 if on else 0), "ccp_grid_set_fused")

    def set_tiling(self, max_t: int, rows_per_chunk: int):
        check(self.L.ccp_grid_set_tiling(self.h, max_t, rows_per_chunk), "ccp_grid_set_tiling")

    def get_tiling(self):
        """(max depth, rows per chunk at that depth, tuned?) of the next unchecked sweep."""
        t, r, tuned = C.c_int32(), C.c_int32(), C.c_int32()
        check(self.L.ccp_grid_get_tiling(self.h, C.byref(t), C.byref(r), C.byref(tuned)), "ccp_grid_get_tiling")
        return t.value, r.value, bool(tuned.value)

    def sweep_l1(self) -> np.ndarray:
        out = np.empty(self.C, dtype=np.float64)
        check(self.L.ccp_grid_sweep_l1(self.h, _ptr(out)), "ccp_grid_sweep_l1")
        return out

    def halo_refreshed(self):
        check(self.L.ccp_grid_halo_refreshed(self.h), "ccp_grid_halo_refreshed")

    def gauss_seidel(self, epsilon=1e-6, max_iteration=1000, check_every=1):
        reps = (Report * self.C)()
        check(self.L.ccp_grid_gauss_seidel(self.h, epsilon, max_iteration, check_every, reps), "ccp_grid_gauss_seidel")
        return list(reps)

    def gauss_seidel_lexicographic(self, epsilon=1e-6, max_iteration=1000, check_every=1):
        """The reference's own sweep order (index order), bit-identical iterates; whole-image handles."""
        reps = (Report * self.C)()
        check(self.L.ccp_grid_gauss_seidel_lexicographic(self.h, epsilon, max_iteration, check_every, reps),
              "ccp_grid_gauss_seidel_lexicographic")
        return list(reps)

    def conjugate_gradient(self, epsilon=1e-16, max_iteration=1000):
        reps = (Report * self.C)()
        check(self.L.ccp_grid_conjugate_gradient(self.h, epsilon, max_iteration, reps), "ccp_grid_conjugate_gradient")
        return list(reps)

    def residual_norm2(self):
        out = np.empty(2 * self.C, dtype=np.float64)
        check(self.L.ccp_grid_residual_norm2(self.h, _ptr(out)), "ccp_grid_residual_norm2")
        return out[:self.C].copy(), out[self.C:].copy()

    def abs_sum(self) -> np.ndarray:
        out = np.empty(self.C, dtype=np.float64)
        check(self.L.ccp_grid_abs_sum(self.h, _ptr(out)), "ccp_grid_abs_sum")
        return out

    def assemble_rhs(self, gx: np.ndarray, gy: np.ndarray, constraint):
        gx = np.ascontiguousarray(gx, dtype=np.float32)
        gy = np.ascontiguousarray(gy, dtype=np.float32)
        cons = _i32(constraint)
        check(self.L.ccp_grid_assemble_rhs(self.h, _ptr(gx), _ptr(gy), gx.strides[0], _ptr(cons)), "ccp_grid_assemble_rhs")

    def assemble_from_images(self, images, label: np.ndarray, init_x: bool = False):
        imgs = [np.ascontiguousarray(i, dtype=np.uint8) for i in images]
        label = np.ascontiguousarray(label, dtype=np.uint8)
        ptrs = (C.c_void_p * len(imgs))(*[i.ctypes.data for i in imgs])
        check(self.L.ccp_grid_assemble_from_images(self.h, ptrs, len(imgs), imgs[0].strides[0], _ptr(label),
                                                   label.strides[0], 1 if init_x else 0),
              "ccp_grid_assemble_from_images")

    def store_u8(self) -> np.ndarray:
        out = np.zeros((self.H, self.W, self.C), dtype=np.uint8)
        check(self.L.ccp_grid_store_u8(self.h, _ptr(out), out.strides[0]), "ccp_grid_store_u8")
        return out

    def set_x_u8(self, image: np.ndarray):
        image = np.ascontiguousarray(image, dtype=np.uint8)
        check(self.L.ccp_grid_set_x_u8(self.h, _ptr(image), image.strides[0]), "ccp_grid_set_x_u8")

    def region_begin(self):
        check(self.L.ccp_grid_region_begin(self.h), "ccp_grid_region_begin")

    def region_end(self):
        """(device ms, sweep launches, iterations of the fused passes) since region_begin — HIP events on
        the handle's stream."""
        ms, n, it = C.c_float(), C.c_int64(), C.c_int64()
        check(self.L.ccp_grid_region_end(self.h, C.byref(ms), C.byref(n), C.byref(it)), "ccp_grid_region_end")
        return ms.value, n.value, it.value

    # ---- row blocks over RCCL (ccp_comm_*) ------------------------------------------------------
    def attach_comm(self, comm: "Comm"):
        check(self.L.ccp_grid_attach_comm(self.h, comm.h if comm is not None else None), "ccp_grid_attach_comm")
        self._comm = comm                       # the communicator must outlive the attachment

    def set_overlap(self, on: bool):
        check(self.L.ccp_grid_set_overlap(self.h, 1 if on else 0), "ccp_grid_set_overlap")

    def exchange_halos(self):
        check(self.L.ccp_grid_exchange_halos(self.h), "ccp_grid_exchange_halos")

    def sweep_rowblocked(self, iterations: int):
        check(self.L.ccp_grid_sweep_rowblocked(self.h, iterations), "ccp_grid_sweep_rowblocked")

    def gauss_seidel_rowblocked(self, epsilon=1e-6, max_iteration=1000, check_every=1):
        reps = (Report * self.C)()
        check(self.L.ccp_grid_gauss_seidel_rowblocked(self.h, epsilon, max_iteration, check_every, reps),
              "ccp_grid_gauss_seidel_rowblocked")
        return list(reps)

    def conjugate_gradient_rowblocked(self, epsilon=1e-16, max_iteration=1000):
        reps = (Report * self.C)()
        check(self.L.ccp_grid_conjugate_gradient_rowblocked(self.h, epsilon, max_iteration, reps), "ccp_grid_conjugate_gradient_rowblocked")
        return list(reps)

    def residual_norm2_global(self):
        out = np.empty(2 * self.C, dtype=np.float64)
        check(self.L.ccp_grid_residual_norm2_global(self.h, _ptr(out)), "ccp_grid_residual_norm2_global")
        return out[:self.C].copy(), out[self.C:].copy()

    def comm_stats(self):
        """(exchanges issued, wait mode: 0 hipStreamWaitValue64 / 1 polling kernel / -1 none, rows sent up, down)."""
        n, mode, up, down = C.c_int64(), C.c_int32(), C.c_int32(), C.c_int32()
        check(self.L.ccp_grid_comm_stats(self.h, C.byref(n), C.byref(mode), C.byref(up), C.byref(down)), "ccp_grid_comm_stats")
        return n.value, mode.value, up.value, down.value

    def last_timing(self):
        ms, n = C.c_float(), C.c_int32()
        check(self.L.ccp_grid_last_timing(self.h, C.byref(ms), C.byref(n)), "ccp_grid_last_timing")
        return ms.value, n.value
