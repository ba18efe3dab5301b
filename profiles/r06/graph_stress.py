"""Stress of the capture -> replay promise of include/trx.h.

tests/test_gpu_kernels.py::test_entry_points_capture_into_a_hip_graph_and_replay returned wrong rows once in ~20 runs
of the suite.  This driver repeats that test's sequence CYCLES times under several variants and, on a mismatch, says
what the replay returned: rows of ANOTHER input set (stale inputs), zeros (never written: the test clears the output),
the "never written" mark, or something else, and which rows.

usage: python profiles/r06/graph_stress.py [cycles] [variant ...]
"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import oracle as O                     # noqa: E402  (checker only)
from triceratops_amd import _lib, synth            # noqa: E402

N_TIME, N_ROWS, K_SETS = 300, 2000, 6


def light_curve(n_time, seed=0):
    rng = np.random.default_rng(synth.SEED + seed)
    t = synth.time_grid(n_time)
    curve = O.flux_grid(O.MODEL_TP, t, synth.reference_tp_row())[0][0]
    return rng, t, synth.noisy_light_curve(rng, curve)


def classify(got, wants, k, prev_k):
    want = wants[k]
    same = (got == want) | (np.isnan(got) & np.isnan(want))
    bad = np.flatnonzero(~same)
    if bad.size == 0:
        return None
    out = {"rows": int(bad.size), "first": bad[:8].tolist(), "last": int(bad[-1]),
           "zeros": int((got[bad] == 0).sum()),
           "unwritten_mark": int((got[bad].view(np.uint64) == 0x7ff8dead0badc0de).sum()),
           "max_rel": float(np.nanmax(np.abs(got[bad] - want[bad]) / np.abs(want[bad])))}
    for j, w in enumerate(wants):
        if j != k and np.array_equal(got[bad], w[bad]):
            out["equals_set"] = j
    out["prev_set"] = prev_k
    out["got"] = got[bad[:4]].tolist()
    out["want"] = want[bad[:4]].tolist()
    return out


def main():
    cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    variants = sys.argv[2:] or ["test", "nosync", "side", "busy", "busy-nosync", "torchcopy"]
    _lib.require_gpu()
    rng, t, flux = light_curve(N_TIME)
    t_d, f_d = _lib.dev(t), _lib.dev(flux)
    sets = [synth.tp_rows(rng, N_ROWS, True) for _ in range(K_SETS)]
    prior = rng.uniform(-5, 0, N_ROWS)
    prior_d = _lib.dev(prior)
    # expected values: a plain (uncaptured) call per input set, itself checked against the oracle
    wants, want_z = [], []
    for rows in sets:
        h = _lib.lnl_batch(0, 0, t_d, f_d, synth.SIGMA, _lib.dev(rows), synth.EXPTIME, 20)
        z = _lib.lnz_from_halfchi2(h, prior_d, 20000, np.log(synth.SIGMA))
        torch.cuda.synchronize()
        got = h.cpu().numpy().copy()
        ref = O.lnl_batch(0, t, flux, synth.SIGMA, rows)
        rel = np.max(np.abs(got - ref) / np.abs(ref))
        assert rel < 1e-9, rel
        wants.append(got)
        want_z.append(float(z.cpu()[0]))
    print("expected values: %d input sets, plain calls agree with the oracle" % K_SETS, flush=True)
    # a larger job for the "busy" variant: other streams with scratch of their own growing and shrinking
    big_t = _lib.dev(synth.time_grid(2000))
    big_f = _lib.dev(np.ones(2000))
    big_rows = _lib.dev(synth.tp_rows(rng, 20000, True))
    side_streams = [torch.cuda.Stream() for _ in range(3)]
    # who owns a captured call's scratch, and for how long (trx::capture_scratch: a user object of the captured graph)
    import ctypes
    import gc
    L = _lib.lib()
    if hasattr(L, "trx_debug_capture_buffers"):
        def bufs():
            a, b = ctypes.c_long(), ctypes.c_long()
            L.trx_debug_capture_buffers(ctypes.byref(a), ctypes.byref(b))
            return a.value, b.value
        rows_d = _lib.dev(sets[0])
        h_d = torch.empty(N_ROWS, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        print("capture buffers (owned by a graph, idle) before any capture:", bufs())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            _lib.lnl_batch(0, 0, t_d, f_d, synth.SIGMA, rows_d, synth.EXPTIME, 20, out=h_d)
        print("  after the capture (torch has destroyed the hipGraph_t, the executable graph lives):", bufs())
        g.replay()
        torch.cuda.synchronize()
        print("  after a replay:", bufs(), "replay equals the plain call:", bool(np.array_equal(h_d.cpu().numpy(), wants[0])))
        del g
        gc.collect()
        torch.cuda.synchronize()
        print("  after the executable graph is gone:", bufs(), flush=True)
    failures = {v: [] for v in variants}
    counts = {v: 0 for v in variants}
    t0 = time.time()
    for cyc in range(cycles):
        v = variants[cyc % len(variants)]
        rows_d = _lib.dev(sets[0])
        h_d = torch.empty(N_ROWS, dtype=torch.float64, device="cuda")
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            _lib.lnl_batch(0, 0, t_d, f_d, synth.SIGMA, rows_d, synth.EXPTIME, 20, out=h_d)     # warm-up
            _lib.lnz_from_halfchi2(h_d, prior_d, 20000, np.log(synth.SIGMA))
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            _lib.lnl_batch(0, 0, t_d, f_d, synth.SIGMA, rows_d, synth.EXPTIME, 20, out=h_d)
            lnz_d = _lib.lnz_from_halfchi2(h_d, prior_d, 20000, np.log(synth.SIGMA))
        prev_k = 0
        for rep in range(3):
            k = (cyc * 3 + rep + 1) % K_SETS
            if "busy" in v:
                for st in side_streams:
                    with torch.cuda.stream(st):
                        _lib.lnl_batch(0, 0, big_t, big_f, synth.SIGMA, big_rows, synth.EXPTIME, 20)
            if v == "torchcopy":
                rows_d.copy_(torch.from_numpy(sets[k]).cuda())
            else:
                rows_d.copy_(_lib.dev(sets[k]))              # new inputs in the captured buffers
            h_d.zero_()
            if "nosync" not in v:
                torch.cuda.synchronize()
            if v == "side":
                with torch.cuda.stream(side_streams[0]):
                    side_streams[0].wait_stream(torch.cuda.current_stream())
                    g.replay()
                torch.cuda.current_stream().wait_stream(side_streams[0])
            else:
                g.replay()
            torch.cuda.synchronize()
            got = h_d.cpu().numpy()
            counts[v] += 1
            f = classify(got, wants, k, prev_k)
            z = float(lnz_d.cpu()[0])
            if f is None and z != want_z[k]:
                f = {"lnz": z, "want_lnz": want_z[k]}
            if f is not None:
                f.update(cycle=cyc, rep=rep, set=k)
                failures[v].append(f)
                if sum(len(x) for x in failures.values()) <= 40:
                    print("MISMATCH", v, f, flush=True)
            prev_k = k
        del g
        if cyc % 200 == 199:
            print("cycle %d, %.1f s, failures so far %s" % (cyc + 1, time.time() - t0, {k: len(x) for k, x in failures.items()}),
                  flush=True)
    print("replays per variant:", counts)
    print("failures per variant:", {k: len(x) for k, x in failures.items()})
    return 1 if any(failures.values()) else 0


if __name__ == "__main__":
    sys.exit(main())
