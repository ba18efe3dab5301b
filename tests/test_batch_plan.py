"""The plan by which the batched likelihood kernel deals rows to waves (batch_plan / batch_at, trx_cells.hpp): rows in
batches of `rows per wave`, an XCD's batches consecutive in its eighth of the rows, the last positions of every XCD
tapered to half and a quarter of the rows per wave.  Host code of libtrx.so, checked here without a GPU through
trx_debug_batch_plan: every row in exactly one batch, for every row count around the plan's thresholds.  (A rule like
this one, applied differently by the workgroup exit test and the batch loop, left rows unwritten in round 4.)"""
import ctypes

import numpy as np
import pytest

from triceratops_amd import _lib


def _plan(L, rows, B, taper):
    pos, small = ctypes.c_long(0), ctypes.c_int(0)
    rc = L.trx_debug_batch_plan(ctypes.c_long(rows), B, taper, ctypes.byref(pos), ctypes.byref(small))
    assert rc == 0, (rows, B, taper, L.trx_last_error())
    return pos.value, small.value


def test_every_row_is_in_exactly_one_batch():
    L = _lib.lib()
    L.trx_debug_batch_plan.argtypes = [ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    rng = np.random.default_rng(1)
    counts = [0, 1, 2, 7, 8, 9, 63, 64, 65, 1000, 2047, 2048, 2049, 3199, 3200, 9999, 10000, 30000, 34001, 100000,
              115488, 299999, 1000003] + [int(x) for x in rng.integers(1, 400000, 60)]
    for B in (1, 2, 3, 4, 6, 8, 11, 13, 16, 22):
        for rows in counts:
            for taper in (0, 1):
                pos, small = _plan(L, rows, B, taper)
                assert 8 * pos * B >= rows                    # enough positions
                if not taper or B == 1:
                    assert pos == (((rows + B - 1) // B + 7) // 8)      # the plain plan: ceil(batches / 8) per XCD


def test_the_taper_shortens_the_last_batches_only_of_large_launches():
    L = _lib.lib()
    L.trx_debug_batch_plan.argtypes = [ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    plain, _ = _plan(L, 100_000, 6, 0)
    tapered, small = _plan(L, 100_000, 6, 1)
    assert tapered > plain and small == 1                     # more, smaller positions at the end
    # a tier holds at most 320 positions per XCD and 15 % / 30 % of the rows: few rows -> hardly any taper
    few, _ = _plan(L, 600, 6, 1)
    assert few <= (((600 + 5) // 6 + 7) // 8) + 8


def test_bad_requests_are_refused():
    L = _lib.lib()
    L.trx_debug_batch_plan.argtypes = [ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    assert L.trx_debug_batch_plan(-1, 6, 1, None, None) != 0
    assert L.trx_debug_batch_plan(10, 0, 1, None, None) != 0
    assert L.trx_debug_batch_plan(10, 23, 1, None, None) != 0
