"""Loader and thin typed wrappers for libtrx.so (C ABI in include/trx.h).

The HIP library is the only compute backend of this package: if it is missing or fails to
load, importing the compute entry points raises -- there is no CPU fallback.
torch is used for device memory, streams and torch.distributed only.
"""
import ctypes
import os
import threading

import numpy as np
import torch  # noqa: F401  (imported first so libtrx binds to the HIP runtime torch loaded)

_HERE = os.path.dirname(os.path.abspath(__file__))
# The production library: no switches, no environment variables, no debug exports (include/trx.h).  TRX_LIB: A/B builds.
LIB_PATH = os.environ.get("TRX_LIB") or os.path.join(_HERE, "libtrx.so")
# The testing library (the same sources with -DTRX_TESTING): + include/trx_debug.h.  tests/ and profiles/ switch to it
# with use_testing_library() (or TRX_TESTING=1 in the environment, for old scripts); the product never does.
TESTING_LIB_PATH = os.environ.get("TRX_TESTING_LIB") or os.path.join(_HERE, "libtrx_testing.so")

MODEL_TP, MODEL_EB, MODEL_EB_TWIN, MODEL_RAW = 0, 1, 2, 3
FLAG_COMPANION_IS_HOST, FLAG_SCALAR_K, FLAG_FP32_MODEL, FLAG_EVALUATE_EXCLUDED = 1, 2, 4, 8
# result-neutral per-call choices (include/trx.h)
FLAG_ALL_SUBEXPOSURES, FLAG_NO_STENCIL, FLAG_COUNT_EVALUATIONS, FLAG_FULL_EVALUATION = 16, 32, 64, 128
N_PARAM = {MODEL_TP: 10, MODEL_EB: 11, MODEL_EB_TWIN: 11, MODEL_RAW: 9}
ERR_NTOTAL = 4

# every symbol include/trx.h declares (tests check that libtrx.so exports all of them and nothing of trx_debug.h)
ABI_SYMBOLS = (
    "trx_lnl_batch", "trx_flux_grid", "trx_chi2_grid", "trx_workspace_bytes",
    "trx_log_mean_exp", "trx_lnz_scenario", "trx_lnz_from_halfchi2", "trx_lnl_batch_host", "trx_flux_grid_host",
    "trx_log_mean_exp_host", "trx_skipped_rows", "trx_pruned_rows",
    "trx_draw_scenario", "trx_draw_args_size", "trx_scenario_evidence", "trx_scenario_enqueue", "trx_star_enqueue",
    "trx_scenario_args_size", "trx_release_scratch", "trx_version", "trx_last_error", "trx_device_count",
)
# ... and include/trx_debug.h: the testing library only
DEBUG_SYMBOLS = (
    "trx_set_rows_per_wave", "trx_set_cell_packing_below", "trx_debug_batch_plan", "trx_set_supersample_tiers",
    "trx_set_stencil", "trx_set_skip_excluded", "trx_set_debug_node_counts", "trx_set_kepler_stepping",
    "trx_set_bounded_evaluation", "trx_set_debug_bounded_lnl", "trx_set_debug_poison", "trx_set_debug_bug",
    "trx_set_probe_rows", "trx_set_star_chain", "trx_debug_capture_buffers",
)


class TrxError(RuntimeError):
    pass


# flags OR-ed into every likelihood launch of the lnZ_* / calc_probs layer: set_precision("fp32")
# puts TRX_FLAG_FP32_MODEL here (BASELINE config 5); the plain wrappers below take explicit flags
EXTRA_FLAGS = 0
# library default of trx_set_cell_packing_below (tests and A/B scripts restore it)
CELL_PACKING_BELOW = 320
# work counters of the scenario layer (bench.py reads them): rows and (row, time) cells that
# went through trx_lnz_scenario since the last reset
# native_calls: lnZ_* calls that went through the library's own chain (trx_scenario_enqueue / trx_star_enqueue)
STATS = {"rows": 0, "cells": 0, "launches": 0, "native_calls": 0}
_stats_lock = threading.Lock()


def count_launch(rows, n_time):
    """bench bookkeeping; the worker threads of sharding.threads update it side by side"""
    with _stats_lock:
        STATS["rows"] += rows
        STATS["cells"] += rows * n_time
        STATS["launches"] += 1
# measurement hook (bench.py --mode batch): a list that receives, per trx_lnz_scenario launch,
# (model, flags, n, n_time, start_event, end_event, first rows of the parameter block)
TRACE = None
TRACE_SAMPLE_ROWS = 128


def reset_stats():
    for k in STATS:
        STATS[k] = 0


def set_precision(mode):
    """'fp64' (default) or 'fp32': fp32 Mandel-Agol arithmetic with fp64 orbit, chi^2 and
    log-mean-exp (TRX_FLAG_FP32_MODEL) for every scenario evaluated through lnZ_* / calc_probs"""
    global EXTRA_FLAGS
    if mode not in ("fp64", "fp32"):
        raise ValueError("precision must be 'fp64' or 'fp32'")
    EXTRA_FLAGS = (EXTRA_FLAGS & ~FLAG_FP32_MODEL) | (FLAG_FP32_MODEL if mode == "fp32" else 0)


def set_full_evaluation(on):
    """True: every lnZ_* / calc_probs call evaluates every masked draw to the end (TRX_FLAG_FULL_EVALUATION on every
    scenario call) instead of the bounded evaluation; same lnZ to rounding, same best draws, slower"""
    global EXTRA_FLAGS
    EXTRA_FLAGS = (EXTRA_FLAGS | FLAG_FULL_EVALUATION) if on else (EXTRA_FLAGS & ~FLAG_FULL_EVALUATION)


_vp = ctypes.c_void_p
_lib = None                 # the library in use
_production = None
_testing = None


def lib():
    """The library in use: libtrx.so, loaded once (raises TrxError if the HIP extension was not built) -- or the
    testing library after use_testing_library()."""
    global _lib, _production
    if _lib is not None:
        return _lib
    if os.environ.get("TRX_TESTING") == "1":
        return use_testing_library(True)
    if _production is None:
        _production = _load(LIB_PATH, False)
    _lib = _production
    return _lib


def testing_lib():
    """libtrx_testing.so (include/trx_debug.h), loaded once; does not change the library in use"""
    global _testing
    if _testing is None:
        _testing = _load(TESTING_LIB_PATH, True)
    return _testing


def use_testing_library(on=True):
    """tests/ and profiles/ only: every wrapper of this module calls the testing library (its own scratch, its own
    switches, all at their defaults unless set) until use_testing_library(False).  Returns the library now in use."""
    global _lib, _production
    if on:
        _lib = testing_lib()
    else:
        if _production is None:
            _production = _load(LIB_PATH, False)
        _lib = _production
    return _lib


def _load(path, testing):
    if not os.path.exists(path):
        raise TrxError(
            "triceratops_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback." % path)
    L = ctypes.CDLL(path)
    c_int, c_long, c_double, c_size_t = ctypes.c_int, ctypes.c_long, ctypes.c_double, ctypes.c_size_t
    L.trx_lnl_batch.restype = c_int
    L.trx_lnl_batch.argtypes = [c_int, c_int, _vp, _vp, c_int, c_double, _vp, c_long, c_double,
                                c_int, _vp, _vp]
    L.trx_flux_grid.restype = c_int
    L.trx_flux_grid.argtypes = [c_int, c_int, _vp, c_int, _vp, c_long, c_double, c_int, _vp, _vp, _vp]
    L.trx_chi2_grid.restype = c_int
    L.trx_chi2_grid.argtypes = [_vp, _vp, c_int, c_long, c_double, _vp, _vp]
    L.trx_workspace_bytes.restype = c_size_t
    L.trx_workspace_bytes.argtypes = []
    L.trx_log_mean_exp.restype = c_int
    L.trx_log_mean_exp.argtypes = [_vp, c_long, c_long, _vp, _vp, c_size_t, _vp]
    L.trx_lnz_scenario.restype = c_int
    L.trx_lnz_scenario.argtypes = [c_int, c_int, _vp, _vp, c_int, c_double, _vp, c_long, c_double,
                                   c_int, _vp, c_long, c_double, _vp, _vp, _vp, c_size_t, _vp]
    L.trx_lnz_from_halfchi2.restype = c_int
    L.trx_lnz_from_halfchi2.argtypes = [_vp, _vp, c_long, c_long, c_double, _vp, _vp, c_size_t, _vp]
    L.trx_lnl_batch_host.restype = c_int
    L.trx_lnl_batch_host.argtypes = [c_int, c_int, _vp, _vp, c_int, c_double, _vp, c_long,
                                     c_double, c_int, _vp]
    L.trx_flux_grid_host.restype = c_int
    L.trx_flux_grid_host.argtypes = [c_int, c_int, _vp, c_int, _vp, c_long, c_double, c_int, _vp, _vp]
    L.trx_log_mean_exp_host.restype = c_int
    L.trx_log_mean_exp_host.argtypes = [_vp, c_long, c_long, _vp]
    L.trx_skipped_rows.restype = c_int
    L.trx_skipped_rows.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), c_int]
    L.trx_pruned_rows.restype = c_int
    L.trx_pruned_rows.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), c_int]
    L.trx_version.restype = ctypes.c_char_p
    L.trx_last_error.restype = ctypes.c_char_p
    L.trx_device_count.restype = c_int
    L.trx_testing = bool(testing)
    if testing:
        for name in ("trx_set_rows_per_wave", "trx_set_kepler_stepping", "trx_set_skip_excluded",
                     "trx_set_bounded_evaluation", "trx_set_debug_bounded_lnl", "trx_set_debug_poison",
                     "trx_set_debug_bug", "trx_set_star_chain", "trx_set_probe_rows", "trx_set_stencil",
                     "trx_set_supersample_tiers", "trx_set_debug_node_counts", "trx_set_cell_packing_below"):
            fn = getattr(L, name)
            fn.restype = c_int
            fn.argtypes = [c_int]
        L.trx_debug_capture_buffers.restype = c_int
        L.trx_debug_capture_buffers.argtypes = [ctypes.POINTER(c_long), ctypes.POINTER(c_long)]
    return L


def check(rc):
    if rc:
        msg = lib().trx_last_error().decode()
        if rc == ERR_NTOTAL:
            raise ValueError(msg)
        raise TrxError("libtrx error %d: %s" % (rc, msg))


def version():
    return lib().trx_version().decode()


_gpu_seen = False


def require_gpu():
    """raises unless a GPU and the HIP library are there (checked until it has succeeded once:
    hipGetDeviceCount costs ~0.1 ms a call on this stack, a tenth of a small lnZ_* call)"""
    global _gpu_seen
    if _gpu_seen:
        return
    if not torch.cuda.is_available() or lib().trx_device_count() < 1:
        raise TrxError("triceratops_amd needs an AMD GPU (gfx950); none is visible and there is no "
                       "CPU fallback")
    _gpu_seen = True
    from . import hw_queues
    if hw_queues()["in_effect"] is False:
        import warnings
        warnings.warn("triceratops_amd was imported after the HIP runtime had been initialised: GPU_MAX_HW_QUEUES=16 "
                      "did not take effect (the runtime keeps its default of 4 hardware queues; calc_probs on more "
                      "than two streams runs ~10 % slower).  Import triceratops_amd before the first GPU call or "
                      "export GPU_MAX_HW_QUEUES=16.", RuntimeWarning, stacklevel=3)


def compute_device():
    """the current GPU as a torch.device (raises without one)"""
    require_gpu()
    return torch.device("cuda", torch.cuda.current_device())


# ---------------------------------------------------------------------------------------
# device helpers
_upload_streams = {}
# events of uploads that may still be in flight: every stream about to read library inputs waits for them ON THE
# DEVICE (wait_uploads), the host never blocks
_pending_uploads = {}          # device index -> [(sequence number, event)], oldest first
_pending_lock = threading.Lock()
_upload_seq = {}               # device index -> sequence number of its newest upload
_stream_seen = {}              # (device index, stream handle) -> sequence number of the last upload it already waits for


def dev(x, device=None):
    """float64 contiguous device tensor from array-like / tensor.

    Host arrays go up asynchronously on a stream of their own, through a pinned staging block: a copy from
    pageable memory returns when it has COMPLETED, and with the GPU busy that was 0.5 ms per small table (a
    third of the host time of a 64-target batch step, profiles/r03/f_batch_host_profile.txt).  Nothing waits on
    the host: the upload leaves an event, the stream that is current here waits for it on the device, and so does
    every stream a library call is enqueued on while the event is pending (wait_uploads; cached tables are read
    by calls on other streams a few microseconds later).  A tensor from dev() is therefore ordered for the stream
    that was current at the upload and for every stream that passes through wait_uploads() / _stream() -- all library
    calls do; a torch operator launched on some OTHER stream right after must call wait_uploads(that stream) itself."""
    if isinstance(x, torch.Tensor):
        return x.to(device=device or "cuda", dtype=torch.float64).contiguous()
    a = np.ascontiguousarray(x, dtype=np.float64)
    d = torch.device(device or "cuda")
    if d.type != "cuda" or not torch.cuda.is_available():
        return torch.as_tensor(a).to(d).contiguous()
    if torch.cuda.is_current_stream_capturing():
        # (an upload is eager work on another stream: a captured call would not be ordered behind it, and its event may
        # neither be waited for nor queried while the capture lasts)
        raise TrxError("triceratops_amd: host data cannot be uploaded while the current stream is being captured into "
                       "a graph; stage the inputs with _lib.dev() before the capture begins")
    if d.index is None:
        d = torch.device("cuda", torch.cuda.current_device())
    up = upload_stream(d).stream
    staged = torch.empty(a.shape, dtype=torch.float64, pin_memory=True)    # (torch caches pinned blocks)
    staged.numpy()[...] = a
    with torch.cuda.stream(up):
        t = staged.to(d, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(up)
    cur = torch.cuda.current_stream(d)
    t.record_stream(cur)      # the allocator must not hand the block out early
    with _pending_lock:
        seq = _upload_seq[d.index] = _upload_seq.get(d.index, 0) + 1
        _pending_uploads.setdefault(d.index, []).append((seq, ev))
    wait_uploads(cur)
    return t


def wait_uploads(stream):
    """`stream` waits (on the device) for every upload that has not completed yet -- each of them once: the uploads
    go up on ONE stream, in order, so waiting for the newest pending one covers all before it"""
    pend = _pending_uploads.get(stream.device.index)
    if not pend:
        return
    if torch.cuda.is_current_stream_capturing():
        # Nothing may be asked of an event while a capture lasts: hipEventQuery fails with "operation not permitted when
        # stream is capturing" AND invalidates the capture (found by profiles/r06/graph_stress.py: an upload still in
        # flight at the warm-up call leaves its event on the list, and the captured call that follows queried it --
        # one capture in ~2700).  A captured call is not executed now; torch.cuda.graph() synchronises the device before
        # it begins the capture, so every upload issued before it has landed, and dev() refuses uploads during one.
        return
    # (keyed per device: the default stream is handle 0 on every device, and sequence numbers are per device too)
    key = (stream.device.index, stream.cuda_stream)
    with _pending_lock:
        while pend and pend[0][1].query():
            pend.pop(0)                                 # (completed: nobody needs to wait any more)
        if not pend:
            # nothing in flight: forget who waited for what, so that a stream created later under a recycled handle
            # does not inherit the record of the one destroyed
            for k in [k for k in _stream_seen if k[0] == stream.device.index]:
                del _stream_seen[k]
            return
        seq, ev = pend[-1]
        if _stream_seen.get(key, 0) >= seq:
            return
        _stream_seen[key] = seq
    stream.wait_event(ev)


class upload_stream:
    """`with upload_stream(device):` -- torch work inside runs on the device's upload stream and has completed
    at exit: for small tables that several lnZ_* calls on different streams read afterwards"""

    def __init__(self, device):
        d = torch.device(device)
        self.stream = self.ctx = None
        if d.type != "cuda":                 # (the torch expression of the tests on CPU tensors)
            return
        if d.index is None:
            d = torch.device("cuda", torch.cuda.current_device())
        self.stream = _upload_streams.get(d.index)
        if self.stream is None:
            self.stream = _upload_streams[d.index] = torch.cuda.Stream(d)
        self.ctx = torch.cuda.stream(self.stream)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()
        return self.stream

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
            self.stream.synchronize()
        return False


def _stream(t):
    """the stream a library call is enqueued on (behind the uploads still in flight)"""
    st = torch.cuda.current_stream(t.device)
    wait_uploads(st)
    return st.cuda_stream


_ws = {}


def workspace(device):
    """LME scratch, one per (device, stream): calls on different streams may overlap"""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    if key not in _ws:
        nbytes = lib().trx_workspace_bytes()
        _ws[key] = torch.empty(nbytes // 8 + 1, dtype=torch.float64, device=device)
    return _ws[key]


def pack_params(model, cols, n=None):
    """SoA [n_param][n] host block from per-sample columns (reference argument order)."""
    assert len(cols) == N_PARAM[model], (len(cols), N_PARAM[model])
    if n is None:
        n = max(int(np.size(c)) for c in cols)
    out = np.empty((len(cols), n), dtype=np.float64)
    for i, c in enumerate(cols):
        out[i] = c
    return out


def lnl_batch(model, flags, time_d, flux_d, sigma, params_d, exptime, nsamples, out=None):
    """chi^2/2 per row on the GPU.  All tensors are fp64 CUDA tensors; params_d is [n_param][n]."""
    require_gpu()
    n = params_d.shape[1]
    assert params_d.shape[0] == N_PARAM[model] and params_d.is_contiguous()
    if out is None:
        out = torch.empty(n, dtype=torch.float64, device=params_d.device)
    with torch.cuda.device(params_d.device):
        check(lib().trx_lnl_batch(model, flags, time_d.data_ptr(), flux_d.data_ptr(),
                                  time_d.numel(), float(sigma), params_d.data_ptr(), n,
                                  float(exptime), int(nsamples), out.data_ptr(), _stream(params_d)))
    return out


def flux_grid(model, flags, time_d, params_d, exptime, nsamples, want_secdepth=True):
    require_gpu()
    n = params_d.shape[1]
    assert params_d.shape[0] == N_PARAM[model] and params_d.is_contiguous()
    out = torch.empty((n, time_d.numel()), dtype=torch.float64, device=params_d.device)
    sec = torch.zeros(n, dtype=torch.float64, device=params_d.device) if want_secdepth else None
    with torch.cuda.device(params_d.device):
        check(lib().trx_flux_grid(model, flags, time_d.data_ptr(), time_d.numel(),
                                  params_d.data_ptr(), n, float(exptime), int(nsamples),
                                  out.data_ptr(), sec.data_ptr() if sec is not None else None,
                                  _stream(params_d)))
    return out, sec


def chi2_grid(flux_d, grid_d, sigma):
    require_gpu()
    n, nt = grid_d.shape
    assert grid_d.is_contiguous() and flux_d.numel() == nt
    out = torch.empty(n, dtype=torch.float64, device=grid_d.device)
    with torch.cuda.device(grid_d.device):
        check(lib().trx_chi2_grid(flux_d.data_ptr(), grid_d.data_ptr(), nt, n, float(sigma),
                                  out.data_ptr(), _stream(grid_d)))
    return out


def log_mean_exp(logw_d, n_total):
    """Device log-mean-exp; returns a 1-element device tensor (no sync)."""
    require_gpu()
    ws = workspace(logw_d.device)
    out = torch.empty(1, dtype=torch.float64, device=logw_d.device)
    with torch.cuda.device(logw_d.device):
        check(lib().trx_log_mean_exp(logw_d.data_ptr(), logw_d.numel(), int(n_total),
                                     out.data_ptr(), ws.data_ptr(), ws.numel() * 8,
                                     _stream(logw_d)))
    return out


def lnz_scenario(model, flags, time_d, flux_d, sigma, params_d, exptime, nsamples, lnprior_d,
                 n_total, lnsigma):
    """Fused chi^2/2 -> log-mean-exp.  Returns (halfchi2[n], lnz[1]) device tensors."""
    require_gpu()
    n = params_d.shape[1]
    device = params_d.device
    flags |= EXTRA_FLAGS
    count_launch(n, time_d.numel())
    h = torch.empty(max(n, 1), dtype=torch.float64, device=device)
    out = torch.empty(1, dtype=torch.float64, device=device)
    ws = workspace(device)
    ev = None
    if TRACE is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    with torch.cuda.device(device):
        check(lib().trx_lnz_scenario(model, flags, time_d.data_ptr(), flux_d.data_ptr(),
                                     time_d.numel(), float(sigma), params_d.data_ptr(), n,
                                     float(exptime), int(nsamples),
                                     lnprior_d.data_ptr() if lnprior_d is not None else None,
                                     int(n_total), float(lnsigma), h.data_ptr(), out.data_ptr(),
                                     ws.data_ptr(), ws.numel() * 8, _stream(params_d)))
    if ev is not None:
        ev[1].record()
        keep = params_d[:, :TRACE_SAMPLE_ROWS].clone() if len(TRACE) < 64 else None
        TRACE.append((model, flags, n, time_d.numel(), ev[0], ev[1], keep))
    return h[:n], out


def lnz_from_halfchi2(h_d, lnprior_d, n_total, lnsigma):
    """lnZ (1-element device tensor) from device chi^2/2 values of the masked draws."""
    require_gpu()
    device = h_d.device
    out = torch.empty(1, dtype=torch.float64, device=device)
    ws = workspace(device)
    with torch.cuda.device(device):
        check(lib().trx_lnz_from_halfchi2(h_d.data_ptr() if h_d.numel() else None,
                                          lnprior_d.data_ptr() if lnprior_d is not None else None,
                                          h_d.numel(), int(n_total), float(lnsigma),
                                          out.data_ptr(), ws.data_ptr(), ws.numel() * 8,
                                          _stream(h_d)))
    return out
