"""CPU: the ticket order of the reference-order sweep (k_lex_wg, csrc/ccp_grid.hip: lex_ticket_order) through the host-only
diagnostic entry point ccp_debug_lex_tickets — no GPU needed.

The launch holds as many persistent workgroups as the chip has room for; each takes the next ticket and WAITS inside the
kernel for the strips its strip depends on.  That can only be free of deadlock if everything a strip waits for holds a
smaller ticket (then the holder of the smallest unfinished ticket can always run).  The dependences are restated here
from the kernel's comments (csrc/ccp_grid_lex.hpp, lex_wg_body): the left neighbour (k, s-1), the same strip of the
group before (k-1, s), and the last reader of the strip's edge buffer (k-2, s+1); strips exist where they hold a pixel of
the image: strip s of group k lies at columns 62 s - 2 T k - 2 t ... + 61 for its sweeps t = 0 .. T-1."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
LIB = os.path.join(ROOT, "coursecomputationalphotography_amd", "lib", "libccp_gs.so")
STRIP = 62


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        pytest.skip("libccp_gs.so not built")
    L = C.CDLL(LIB)
    L.ccp_debug_lex_tickets.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    L.ccp_debug_lex_tickets.restype = C.c_int
    return L


def tickets(L, W, T, groups):
    strips, count = C.c_int32(), C.c_int64()
    assert L.ccp_debug_lex_tickets(W, T, groups, None, 0, C.byref(strips), C.byref(count)) == 0
    order = np.zeros(count.value, dtype=np.uint32)
    assert L.ccp_debug_lex_tickets(W, T, groups, order.ctypes.data, order.size, C.byref(strips), C.byref(count)) == 0
    return order, strips.value


def exists(W, T, k, s):
    """does strip s of group k hold a pixel of the image in any of its sweeps?  (restated, not imported)"""
    if k < 0 or s < 0:
        return False
    lo = STRIP * s - 2 * T * k - 2 * (T - 1)          # leftmost column: its last sweep, lane 2
    hi = STRIP * s - 2 * T * k + STRIP - 1            # rightmost column: its first sweep, lane 63
    return hi >= 0 and lo <= W - 1


@pytest.mark.parametrize("T", [8, 4, 2, 1])
@pytest.mark.parametrize("W", [1, 2, 61, 62, 63, 125, 512, 700, 4096, 16384])
def test_everything_a_strip_waits_for_holds_a_smaller_ticket(lib, W, T):
    for groups in (1, 2, 3, 5, 13, 128 if T == 8 else 64):
        if groups * T > 1024:
            continue
        order, S = tickets(lib, W, T, groups)
        rank = {int(v): i for i, v in enumerate(order)}
        assert len(rank) == len(order), "a strip holds two tickets"
        # exactly the strips that exist, and within the slot count
        want = {k * S + s for k in range(groups) for s in range(S) if exists(W, T, k, s)}
        assert set(rank) == want, (W, T, groups, len(rank), len(want))
        assert not any(exists(W, T, k, S) for k in range(groups)), "a strip beyond the slot count holds pixels"
        for v, i in rank.items():
            k, s = divmod(v, S)
            for dk, ds in ((0, -1), (-1, 0), (-2, 1)):
                kk, ss = k + dk, s + ds
                if exists(W, T, kk, ss):
                    assert rank[kk * S + ss] < i, (W, T, groups, (k, s), "waits for", (kk, ss))


@pytest.mark.parametrize("T", [8, 4, 2, 1])
def test_the_strips_of_a_group_cover_every_column_of_every_sweep(lib, T):
    """every pixel column of every sweep t of every group lies in the real lanes (2 .. 63) of exactly one strip"""
    for W in (1, 2, 63, 500, 4096):
        for groups in (1, 4, 31):
            order, S = tickets(lib, W, T, groups)
            have = set(int(v) for v in order)
            for k in (0, groups // 2, groups - 1):
                for t in range(T):
                    owner = np.zeros(W, dtype=np.int32)
                    for s in range(S):
                        if k * S + s not in have:
                            continue
                        lo = STRIP * s - 2 * T * k - 2 * t
                        a, b = max(lo, 0), min(lo + STRIP - 1, W - 1)
                        if a <= b:
                            owner[a:b + 1] += 1
                    assert (owner == 1).all(), (W, T, groups, k, t)


def test_bad_arguments_are_refused(lib):
    strips, count = C.c_int32(), C.c_int64()
    for W, T, G in ((0, 8, 1), (10, 3, 1), (10, 8, 0), (10, 8, 129)):
        assert lib.ccp_debug_lex_tickets(W, T, G, None, 0, C.byref(strips), C.byref(count)) != 0
    small = np.zeros(1, dtype=np.uint32)
    assert lib.ccp_debug_lex_tickets(700, 8, 4, small.ctypes.data, 1, C.byref(strips), C.byref(count)) != 0
