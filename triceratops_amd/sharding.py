"""Scenario-level sharding of calc_probs over the GPUs of one node.

The reference runs its (star, lnZ_* call) units strictly one after the other in one Python
thread (triceratops.py:736-1428).  The units are independent given the light curve, so here
they are dealt to the ranks of a torch.distributed job (one process per GPU; backend "nccl" is
RCCL over xGMI) with a longest-processing-time-first schedule, every rank evaluates its own
units on its own GPU, and ONE all_gather of a small fp64 record table (15 numbers per
scenario: the best-fit row + lnZ) assembles the result on every rank.  No other collective is
on the data path; the message is a few KB, i.e. latency-bound, so a single direct all_gather
is the right primitive (no ring, no bucketing).

Random numbers.  world == 1: the generator of the sampling mode is consumed unit after unit (numpy modes:
identical to the reference under the same np.random.seed), unless per_unit_seed is set.  world > 1: each
unit draws from its own seed, derived from (base of the rank that owns it, unit index).  base is one draw of
the sampling mode's host generator -- torch's CPU generator in "device" mode (torch.manual_seed governs every
path of that mode), numpy's global stream in the numpy modes -- taken on EVERY rank.  Ranks seeded alike
(the way to a reproducible run) draw the same base, and the result then depends neither on the partition nor
on the world size (per_unit_seed = True gives the same numbers on one GPU).  Ranks that were not seeded -- the
reference's normal usage -- draw different bases: each unit is still an independent draw, the run is just
not reproducible, which is what "unseeded" means.  No seed broadcast: the all_gather is the only collective
of a calc_probs; every rank's base rides in its chunk's header row and is kept in `last_seed_bases` (all
equal = the run can be repeated from the seed).
"""
import os

import numpy as np

RECORD_COLS = ("M_s", "R_s", "u1", "u2", "P_orb", "inc", "b", "R_p", "ecc", "argp", "M_EB",
               "R_EB", "fluxratio_EB", "fluxratio_comp", "lnZ")
per_unit_seed = False
# Host threads that evaluate this rank's units side by side, each on its own HIP stream
# (set_sampling("device") with the kernel's own random numbers only).  A lnZ_* call is ~25 small
# launches and two host syncs around one or two large kernels; a second and third call in flight
# fill the GPU while the first waits for Python.  Implies per-unit seeding (the numbers a unit draws
# depend on its seed only, so the result does not depend on the number of threads).
threads = 1
# HIP streams ONE host thread deals the lnZ_* calls of a calc_probs to (set_sampling("device")): a call
# is enqueued without any host synchronisation (trx_scenario_enqueue), so the thread enqueues every
# unit, the streams overlap the small kernels of one call with the large ones of another, and the
# results are read after one wait.  The library keeps ~0.3 GB of scratch per stream at N = 1e6.
# Four: a call is a chain of dependent launches, most of them small, so a stream keeps the chip busy only while its
# likelihood kernel runs, and the other streams fill those gaps.  64 TOIs x 18 scenarios, the stream counts visited in
# turn over seven rounds (profiles/batch_timing.py, profiles/r04/hw_queues.txt): 0.35 s a step on one stream, 0.225 on
# two, 0.19 on three, 0.175 on four, 0.18-0.19 on six and on eight -- with one hardware queue per stream
# (GPU_MAX_HW_QUEUES = 16 since round 6 -- 8 until then --, set by the package: with the runtime's default of 4, streams share queues and three streams
# run slower than two).
streams = int(os.environ.get("TRX_STREAMS", "4"))
# host seconds of the last single-thread pass: enqueueing every call, then waiting for the streams
# (+ the collective, and -- calc_probs_many -- the unit lists and the result tables of all targets)
timing = {"enqueue_s": 0.0, "wait_s": 0.0, "gather_s": 0.0, "prepare_s": 0.0, "finish_s": 0.0, "build_s": 0.0,
          "library_s": 0.0}
# seed bases of the ranks of the last multi-rank run_units (from the all_gather's header rows)
last_seed_bases = None
# what the schedule of the last run_units gave every rank: lnZ_* calls, distinct (job, star)s and distinct jobs
last_share = {"calls": [0], "stars": [0], "jobs": [0]}
last_own_jobs = set()
_warned_bases = False
# Opt-in (advisor, round 5): True makes the first device pass call gc.freeze() once -- every object alive then moves to
# the collector's permanent generation, so that the full collection which follows a held-off pass looks at the pass's own
# objects only (a 64-target step lost 75 ms to one every fourth step, profiles/r05/gc_pause.txt).  freeze() is for good:
# cyclic garbage that forms LATER among the frozen objects is never reclaimed until release() (gc.unfreeze()), which a
# long-lived service may not want decided for it -- so the default leaves the collector's generations alone (bench.py
# and TRX_FREEZE_GC=1 switch it on).  Either way the collector is held off (gc.disable(), process-wide: it affects the
# caller's other threads too) only while a pass enqueues and waits, and is switched back on afterwards.
freeze_gc = os.environ.get("TRX_FREEZE_GC", "0") == "1"
_gc_frozen = False
# Device scratch of one stream of a pass (include/trx.h): a launch chain of c calls holds c x (~0.36 GB of draw-side
# buffers + ~0.16 GB of likelihood scratch per branch, 1.5 branches a call on average) per 1e6 draws -- c = chain_calls
# (12), or fewer when TRX_CHAIN_DRAWS (2.5e7 draws' worth) bounds the chain: ~7 GB per stream at N = 1e6, ~15 GB at
# N = 2-3e6 (advisor, round 5: the round-4 figure of 0.36 GB per stream predates the chains).  The streams of a pass
# are capped so that their scratch together stays near this many bytes: 6 streams at N = 1e6, 3 at 3e6, 2 from ~5e6 on.
# A sixth of an MI355X's 288 GB by default; several ranks sharing one device should divide it (TRX_SCRATCH_GB).
scratch_budget_bytes = float(os.environ.get("TRX_SCRATCH_GB", "48")) * 1e9


def stream_scratch_bytes(n_draws):
    """estimate of the library's scratch per stream for calls of n_draws draws (see scratch_budget_bytes)"""
    per_call = (0.36e9 + 1.5 * 0.16e9) * n_draws / 1e6
    calls = max(1.0, min(float(chain_calls), 2.5e7 / max(n_draws, 1)))
    return calls * per_call


def release():
    """gives back what a pass with freeze_gc kept: the collector's permanent generation is thawed (gc.unfreeze()) and
    the next device pass freezes afresh if freeze_gc is still set"""
    import gc
    global _gc_frozen
    if _gc_frozen:
        gc.unfreeze()
        _gc_frozen = False

# relative cost of a unit by its drop key: EB calls evaluate two branches plus the 25-point
# secondary-eclipse scan; companion/background hosts add per-draw stellar relations
_COST = {"TP": 1.0, "PTP": 1.1, "STP": 1.2, "DTP": 1.1, "BTP": 1.2, "NTP": 1.0,
         "EB": 1.7, "PEB": 1.8, "SEB": 1.9, "DEB": 1.8, "BEB": 1.9, "NEB": 1.7}


def _draw_base():
    """one 31-bit seed base from the host generator that governs the current sampling mode"""
    from . import fused as _fused
    if _fused.threadable():
        import torch
        return int(torch.randint(0, 2 ** 31 - 1, (1,), dtype=torch.int64).item())
    return int(np.random.randint(0, 2 ** 31 - 1))


# True: a process group of ONE rank takes the multi-rank path too -- schedule, per-unit seeds, the record table through
# the group's all_gather_into_tensor (RCCL on the "nccl" backend: device tensors, header rows, padding) -- so that the
# collective branch can be executed and timed on a single GPU (tests/test_gpu_rccl_world1.py; bench.py's `batch` object
# reports its gather_s).  False (default): one rank has nothing to gather and skips it.
collective_at_world_one = os.environ.get("TRX_COLLECTIVE_WORLD1", "0") == "1"


def _dist():
    try:
        import torch.distributed as dist
    except Exception:  # pragma: no cover
        return None
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or collective_at_world_one):
        return dist
    return None


def _lpt(items, cost, load, calls, n_calls):
    """longest-processing-time-first: the dearest item first, each to the rank with the least load so far (ties:
    fewest calls, then rank index)"""
    out = {}
    for it in sorted(items, key=lambda it: (-cost[it], it)):
        r = min(range(len(load)), key=lambda i: (load[i], calls[i], i))
        out[it] = r
        load[r] += cost[it]
        calls[r] += n_calls[it]
    return out


def schedule(costs, world, groups=None):
    """Owner of every unit; every rank computes the same table.

    groups[k] = (job, star) of unit k (calc_probs_many: the TOI and the star of the lnZ_* call).  A rank pays for
    every star it touches on the HOST as well -- the renormalised light curve, the star's constants and tables,
    one trx_star_enqueue per star -- so units are dealt as whole TOIs, then as whole stars, and only what is left
    as single calls (round 4 dealt single calls: 64 like-sized TOIs on 8 ranks gave every rank 16 TOIs x 6 calls
    instead of 8 x 12, and the host path of a rank shrank to a quarter, not an eighth).  At each level the groups
    are dealt longest-first while there are at least two per rank; a group dearer than 3/4 of a rank's fair share
    (the target star of a single crowded field), and the len mod world cheapest ones, go on to the next level,
    where they fill in what the whole groups left uneven.  groups = None: single units (one calc_probs)."""
    n = len(costs)
    load, calls = [0.0] * world, [0] * world
    owner = [0] * n
    if world <= 1 or n == 0:
        return owner
    fair = 0.75 * sum(costs) / world
    todo = list(range(n))
    levels = [] if groups is None else [lambda k: groups[k][0], lambda k: groups[k]]
    for level in levels:
        members = {}
        for k in todo:
            members.setdefault(level(k), []).append(k)
        cost = {g: sum(costs[k] for k in ks) for g, ks in members.items()}
        whole = sorted((g for g in members if cost[g] <= fair), key=lambda g: (-cost[g], members[g][0]))
        if len(whole) < 2 * world:
            continue
        whole = whole[:len(whole) - len(whole) % world]
        dealt = _lpt(whole, cost, load, calls, {g: len(members[g]) for g in whole})
        for g, r in dealt.items():
            for k in members[g]:
                owner[k] = r
        todo = [k for k in todo if level(k) not in dealt]
    dealt = _lpt(todo, {k: costs[k] for k in todo}, load, calls, {k: 1 for k in todo})
    for k, r in dealt.items():
        owner[k] = r
    return owner


def _record(res):
    """(res,) or (res, res_twin) -> (n, 15) array of the best row + lnZ of each dict"""
    dicts = res if isinstance(res, tuple) else (res,)
    out = np.empty((len(dicts), len(RECORD_COLS)))
    for i, d in enumerate(dicts):
        for j, c in enumerate(RECORD_COLS):
            out[i, j] = d[c] if c == "lnZ" else d[c][0]
    return out


def _as_dicts(rec):
    return tuple({c: rec[i, j] for j, c in enumerate(RECORD_COLS)} for i in range(rec.shape[0]))


def run_units(units, verbose=0, as_rows=False, job_done=None):
    """Evaluate the work units of one calc_probs.

    units: list of (first_row, names, star_num, ID, thunk_or_None, key[, weight, draws, (job, star)]); weight scales
    the scenario cost of `key` in the schedule (units of differently sized jobs, calc_probs_many), (job, star) lets
    the schedule deal whole TOIs and whole stars.
    Returns, per unit, None (dropped scenario) or a tuple of per-scenario dicts
    {column: best value, 'lnZ': float} -- with as_rows, the (branches, 15) array of RECORD_COLS instead (what
    target._finish reads: building a dict per scenario and taking it apart again cost 4 ms of a 64-target step, on
    every rank).
    job_done(job, results_of_its_units): called as soon as every unit of a job (the first element of a unit's
    (job, star)) has its records -- while the GPU still works on later jobs -- on one rank with the calls enqueued
    from one host thread; elsewhere never (the caller finishes what is left)."""
    dist = _dist()
    world = dist.get_world_size() if dist else 1
    rank = dist.get_rank() if dist else 0
    live = [k for k, u in enumerate(units) if u[4] is not None]
    owner = {k: 0 for k in live}
    base = None
    if dist:
        base = _draw_base()         # ranks seeded alike draw the same one (module docstring)
        own = schedule([_COST.get(units[k][5], 1.0) * (units[k][6] if len(units[k]) > 6 else 1.0)
                        for k in live], world,
                       [units[k][8] for k in live] if all(len(units[k]) > 8 for k in live) else None)
        owner = {k: own[i] for i, k in enumerate(live)}
    global last_share
    calls, stars, jobs = [0] * world, [set() for _ in range(world)], [set() for _ in range(world)]
    for k in live:                          # (one pass over the units: this runs on every rank for ALL units)
        r = owner[k]
        calls[r] += 1
        stars[r].add(_star_of(units[k]))
        jobs[r].add(_job_of(units[k]))
    last_share = {"calls": calls, "stars": [len(x) for x in stars], "jobs": [len(x) for x in jobs]}
    global last_own_jobs
    last_own_jobs = jobs[rank]              # (calc_probs_many: the targets whose tables this rank fills at once)
    # calc_probs keeps the best draw of every scenario only: with the device generator the fused
    # path then selects it with one argmin instead of a top-100 sort (fused.TABLE_ROWS)
    from . import fused as _fused
    if not dist and (per_unit_seed or (threads > 1 and _fused.threadable())):
        # (threads only apply to the device generator: the numpy modes keep consuming the caller's stream)
        base = _draw_base()
    _fused.TABLE_ROWS = 1
    try:
        return _run_units(units, live, owner, base, dist, world, rank, verbose, as_rows, job_done if not dist else None)
    finally:
        _fused.TABLE_ROWS = _fused.N_BEST


_streams = {}


def _worker_streams(device, n):
    """the worker threads' HIP streams, kept from call to call: the library keeps its scratch per
    stream (include/trx.h), so fresh streams every calc_probs would keep growing new buffers"""
    import torch
    have = _streams.setdefault(device, [])
    while len(have) < n:
        have.append(torch.cuda.Stream(device))
    return have[:n]


# lnZ_* calls per launch chain the dealing aims at (the library takes up to 16 calls / 24 branches per chain)
chain_calls = int(os.environ.get("TRX_CHAIN_CALLS", "12"))


def _pieces(units, mine_k, n_streams):
    """this rank's units as contiguous runs of one target's calls: whole targets when there are at least as many as
    streams, else a target's calls in enough runs to keep the streams busy (a 75-scenario calc_probs: 50 calls in
    five runs of ten)"""
    jobs = []
    for k in mine_k:
        j = _job_of(units[k])
        if jobs and jobs[-1][0] == j:
            jobs[-1][1].append(k)
        else:
            jobs.append((j, [k]))
    out = []
    for _, ks in jobs:
        n = len(ks)
        want = -(-n // max(chain_calls, 1))
        if len(jobs) < n_streams:
            want = max(want, min(n_streams // len(jobs), n // 5))
        want = max(1, min(want, n))
        cost = [_COST.get(units[k][5], 1.0) for k in ks]
        total, acc, at = sum(cost), 0.0, 0
        for p in range(want):
            # contiguous, balanced by cost: piece p ends where the running cost passes (p + 1) / want of the total
            end = at
            while end < n and (p == want - 1 or acc + cost[end] <= total * (p + 1) / want + 1e-9 or end == at):
                acc += cost[end]
                end += 1
            if end > at:
                out.append(ks[at:end])
            at = end
    return out


def _star_of(u):
    """(job, star) of a unit when the caller gave it (target._prepare), else its star's ID"""
    return u[8] if len(u) > 8 else u[3]


def _job_of(u):
    return u[8][0] if len(u) > 8 else 0


def _run_units(units, live, owner, base, dist, world, rank, verbose, as_rows=False, job_done=None):
    rows = {k: len(units[k][1]) for k in live}
    offs, total = {}, 0
    for k in live:
        offs[k] = total
        total += rows[k]
    table = np.full((total, len(RECORD_COLS)), np.nan)
    from . import fused as _fused
    mine_k = [k for k in live if owner[k] == rank]

    pending = []                                  # (unit, fused.Pending): calls in flight

    def one(k):
        j0, names, snum, ID, fn, key = units[k][:6]
        if verbose == 1:
            print("Calculating " + ", ".join(names) + " scenario probabilit"
                  + ("y" if len(names) == 1 else "ies") + " for " + str(ID)
                  + (" [rank %d]" % rank if dist else "") + ".")
        if base is not None:
            unit_seed = (base + 7919 * (k + 1)) % (2 ** 32)
            _fused.set_thread_seed(unit_seed)        # the draw kernel's Philox key (thread-local)
            if n_threads == 1 and not _fused.threadable():
                np.random.seed(unit_seed)            # the numpy sampling modes
                import torch
                torch.manual_seed(unit_seed)         # staged draws from torch's generator
        try:
            res = fn()
        finally:
            _fused.set_thread_seed(None)
        if isinstance(res, _fused.Pending):
            pending.append((k, res))                 # list.append is atomic: worker threads share it
        else:
            table[offs[k]:offs[k] + rows[k]] = _record(res)

    def resolve():
        for k, rec in _fused.records_to_rows(pending).items():
            table[offs[k]:offs[k] + rows[k]] = rec

    on_device = _fused.threadable() or _fused.staged_native()
    if on_device:
        import torch
        on_device = torch.cuda.is_available()          # (without a GPU the thunks fail loudly themselves)
    # (numpy's stream is consumed on this thread, call after call, in the reference's order: one host thread)
    n_threads = min(threads, len(mine_k)) if (base is not None and on_device and _fused.threadable()) else 1
    if n_threads <= 1 and not (on_device and mine_k):
        for k in mine_k:
            one(k)
    elif n_threads <= 1:
        # one host thread, a few streams: every call is enqueued, then one wait per stream
        import torch
        device = torch.cuda.current_device()
        n_draws = max([units[k][7] for k in mine_k if len(units[k]) > 7] + [1])
        cap = max(2, int(scratch_budget_bytes // stream_scratch_bytes(n_draws)))
        pool = _worker_streams(device, max(1, min(streams, cap, len(mine_k))))
        torch.cuda.current_stream().synchronize()    # inputs staged on the caller's stream
        _fused.begin_deferred(len(mine_k))
        drained = False
        # The cyclic garbage collector is held off while the calls are enqueued and awaited: a pass builds a few thousand
        # small objects (argument blocks, closures, Pending records; no reference cycles among them), which now and then
        # tips the collector into a FULL collection -- 75 ms in a process that has imported torch and pandas, inside a
        # 140 ms step (every fourth 64-target step: profiles/r05/gc_pause.txt).  Reference counting frees the pass's
        # objects as before; the collector is switched back on (if it was on) when the pass is over.
        import gc
        global _gc_frozen
        if freeze_gc and not _gc_frozen:
            # (once: the objects alive now -- torch, pandas, the star tables -- move to the collector's permanent
            # generation, so the full collection that follows a held-off pass looks at the pass's objects only)
            gc.collect()
            gc.freeze()
            _gc_frozen = True
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            import time
            t0 = time.perf_counter()
            # The library turns consecutive calls on ONE stream that share a light curve's time stamps and N into one
            # launch chain (trx_star_enqueue, include/trx.h): every kernel once for up to 16 calls.  So the calls go
            # to the streams in PIECES -- contiguous runs of one target's calls, ~chain_calls each -- and a piece to
            # the stream with the least work queued so far (by the schedule's cost weights); one library call per piece.
            load = [0.0] * len(pool)
            done_marks = []                # (event behind a piece, number of pending calls up to and including it)
            t_build = t_lib = 0.0          # of enqueue_s: Python building the argument blocks / inside trx_star_enqueue
            for piece in _pieces(units, mine_k, len(pool)):
                j = min(range(len(pool)), key=lambda i: (load[i], i))
                load[j] += sum(_COST.get(units[k][5], 1.0) * (units[k][6] if len(units[k]) > 6 else 1.0) for k in piece)
                t_a = time.perf_counter()
                with torch.cuda.stream(pool[j]):
                    for k in piece:
                        one(k)
                t_b = time.perf_counter()
                _fused.flush()
                t_build += t_b - t_a
                t_lib += time.perf_counter() - t_b
                ev = torch.cuda.Event()
                ev.record(pool[j])
                done_marks.append((ev, len(pending)))
            timing["build_s"], timing["library_s"] = t_build, t_lib
            timing["enqueue_s"] = time.perf_counter() - t0
            # The records are turned into table rows piece by piece, as the pieces finish, while the GPU works on the
            # later ones (the host is idle for most of the wait: until round 5 it slept through it and converted all
            # records afterwards, 3-5 ms of every step and of every rank).  Pieces on one stream finish in order;
            # across streams the order of enqueueing is a good guess and a wrong one only waits a little longer.
            at = 0
            t_wait = 0.0
            # (a job whose units all have their rows is handed to the caller at once -- its table is filled while
            # the GPU works on the later jobs instead of after the last kernel)
            left, members = {}, {}
            if job_done is not None:
                for k, u in enumerate(units):
                    members.setdefault(_job_of(u), []).append(k)
                for k in mine_k:
                    left[_job_of(units[k])] = left.get(_job_of(units[k]), 0) + 1
                for k in set(mine_k) - {k for k, _ in pending}:         # (calls that returned their records at once)
                    left[_job_of(units[k])] -= 1
            for ev, upto in done_marks:
                t_w = time.perf_counter()
                ev.synchronize()
                t_wait += time.perf_counter() - t_w
                if upto > at:
                    got = _fused.records_to_rows(pending[at:upto])
                    for k, rec in got.items():
                        table[offs[k]:offs[k] + rows[k]] = rec
                    at = upto
                    if job_done is not None:
                        for k in got:
                            j = _job_of(units[k])
                            left[j] -= 1
                            if left[j] == 0:
                                job_done(j, [None if units[m][4] is None else
                                             (table[offs[m]:offs[m] + rows[m]] if as_rows else _as_dicts(table[offs[m]:offs[m] + rows[m]]))
                                             for m in members[j]])
            for st in pool:
                st.synchronize()
            drained = True
            timing["wait_s"] = t_wait
            if at < len(pending):
                for k, rec in _fused.records_to_rows(pending[at:]).items():
                    table[offs[k]:offs[k] + rows[k]] = rec
        finally:
            if not drained:
                # a call failed after others were enqueued: their kernels and record copies still use the
                # pinned block and the cached tables -- wait for them before anything is released
                for st in pool:
                    st.synchronize()
            pending.clear()
            _fused.end_deferred()
            if gc_was_on:
                gc.enable()
    else:
        import queue
        import threading
        import torch
        device = torch.cuda.current_device()
        todo = queue.SimpleQueue()
        cost = {k: _COST.get(units[k][5], 1.0) * (units[k][6] if len(units[k]) > 6 else 1.0) for k in mine_k}
        for k in sorted(mine_k, key=lambda k: (-cost[k], k)):
            todo.put(k)
        errors = []

        pool_streams = _worker_streams(device, n_threads)

        def worker(stream):
            try:
                torch.cuda.set_device(device)        # the current device is thread-local
                _fused.begin_deferred(len(mine_k))
                with torch.cuda.stream(stream):
                    while True:
                        try:
                            k = todo.get_nowait()
                        except queue.Empty:
                            break
                        one(k)
                        _fused.flush()
            except BaseException as exc:              # re-raised in the caller's thread
                errors.append(exc)
            finally:
                stream.synchronize()                  # (also after an error: calls enqueued before it are in flight)
                _fused.end_deferred()

        torch.cuda.current_stream().synchronize()    # inputs staged on the caller's stream
        pool = [threading.Thread(target=worker, args=(pool_streams[i],)) for i in range(n_threads)]
        for t in pool:
            t.start()
        for t in pool:
            t.join()
        if errors:
            pending.clear()
            raise errors[0]
        resolve()

    if dist:
        # ONE collective: every rank contributes the records of its own units (in unit order, padded to the
        # largest share) behind one header row that carries its seed base -- 15 doubles per scenario, a few KB
        # per rank, latency-bound: a direct all_gather, no ring, no bucketing
        import time
        import torch
        t_g = time.perf_counter()
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        ncol = len(RECORD_COLS)
        share = [sum(rows[k] for k in live if owner[k] == r) for r in range(world)]
        chunk = np.full((1 + max(share + [0]), ncol), np.nan)
        chunk[0, 0] = float(base)
        at = 1
        for k in mine_k:
            chunk[at:at + rows[k]] = table[offs[k]:offs[k] + rows[k]]
            at += rows[k]
        mine = torch.as_tensor(chunk).to(dev).reshape(-1)
        gathered = torch.empty(world * mine.numel(), dtype=mine.dtype, device=dev)
        dist.all_gather_into_tensor(gathered, mine)          # the single collective
        g = gathered.cpu().numpy().reshape((world,) + chunk.shape)
        global last_seed_bases, _warned_bases
        last_seed_bases = [int(b) for b in g[:, 0, 0]]
        if len(set(last_seed_bases)) > 1 and not _warned_bases:
            # (relaxed in round 4: unseeded ranks just run -- but a user who seeded only some ranks, or only numpy in
            # "device" mode, should hear that the run cannot be repeated)
            import warnings
            warnings.warn("calc_probs on %d ranks: the ranks drew different seed bases %s, so this run is not reproducible "
                          "and its numbers depend on the partition.  Seed every rank alike (torch.manual_seed in 'device' "
                          "mode, np.random.seed in the numpy modes) for a repeatable, partition-independent run."
                          % (world, last_seed_bases[:4]), RuntimeWarning, stacklevel=4)
            _warned_bases = True
        at = [1] * world
        for k in live:
            r = owner[k]
            table[offs[k]:offs[k] + rows[k]] = g[r, at[r]:at[r] + rows[k]]
            at[r] += rows[k]
        timing["gather_s"] = time.perf_counter() - t_g

    out = []
    for k, u in enumerate(units):
        if u[4] is None:
            out.append(None)
        else:
            rec = table[offs[k]:offs[k] + rows[k]]
            out.append(rec if as_rows else _as_dicts(rec))
    return out
