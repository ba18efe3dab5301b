"""Multi-process path of calc_probs (triceratops_amd/sharding.py) on CPU: world_size 2, gloo.

The work units here are cheap numpy thunks that draw from the global numpy stream, exactly like
the lnZ_* functions do, so the test pins: the LPT schedule is identical on every rank, each unit
runs on exactly one rank, the single all_gather reassembles the record table in unit order, and
per-unit reseeding makes the result independent of the world size.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from triceratops_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_units(n_stars=3, drop=("PEB",)):
    """the unit layout calc_probs builds: 10 target calls + 2 per nearby star"""
    from triceratops_amd.triceratops import _TARGET_CALLS
    units = []

    def thunk(n_res, tag):
        def run():
            out = []
            for r in range(n_res):
                d = {c: np.random.rand(100) + tag for c in sharding.RECORD_COLS if c != "lnZ"}
                d["lnZ"] = float(-50 * np.random.rand() - tag)
                out.append(d)
            return out[0] if n_res == 1 else tuple(out)
        return run

    for key, names, j0, snum in _TARGET_CALLS:
        units.append((j0, names, snum, 111, None if key in drop else thunk(len(names), j0), key))
    for i in range(1, n_stars):
        j0 = 15 + 3 * (i - 1)
        units.append((j0, ("NTP",), 1, 200 + i, thunk(1, j0), "NTP"))
        units.append((j0 + 1, ("NEB", "NEBx2P"), 1, 200 + i, thunk(2, j0 + 1), "NEB"))
    return units


def _flatten(results):
    rows = []
    for r in results:
        if r is None:
            rows.append(None)
        else:
            rows.append(np.array([[d[c] for c in sharding.RECORD_COLS] for d in r]))
    return rows


def _worker(rank, world, port, q, seed=4242):
    import triceratops_amd
    triceratops_amd.set_sampling("numpy")       # (a fresh process starts in the package default, "device")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    np.random.seed(seed if seed else 100 + rank)   # same global state on every rank (sharding verifies it)
    units = _fake_units()
    try:
        res = sharding.run_units(units, verbose=0)
    except RuntimeError as exc:
        q.put((rank, str(exc), None))
        dist.destroy_process_group()
        return
    owners = sharding.schedule([sharding._COST.get(u[5], 1.0) for u in units if u[4] is not None], world)
    q.put((rank, _flatten(res), owners))
    dist.barrier()
    dist.destroy_process_group()


def test_schedule_is_balanced_and_deterministic():
    costs = [1.0, 1.7, 1.1, 1.8, 1.2, 1.9, 1.1, 1.8, 1.2, 1.9] + [1.0, 1.7] * 5
    for world in (1, 2, 4, 8):
        own = sharding.schedule(costs, world)
        assert own == sharding.schedule(list(costs), world)
        load = [sum(c for c, o in zip(costs, own) if o == r) for r in range(world)]
        assert max(load) - min(load) <= max(costs) + 1e-12
        assert set(own) == set(range(min(world, len(costs))))


def test_world2_gloo_matches_single_process():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort(key=lambda t: t[0])
    (_, res0, own0), (_, res1, own1) = got
    assert own0 == own1 and set(own0) == {0, 1}
    # every rank ends with the same full table
    for a, b in zip(res0, res1):
        assert (a is None and b is None) or np.array_equal(a, b)
    # ... which equals a single-process run with per-unit seeding (partition independence)
    np.random.seed(4242)
    sharding.per_unit_seed = True
    try:
        single = _flatten(sharding.run_units(_fake_units(), verbose=0))
    finally:
        sharding.per_unit_seed = False
    for a, b in zip(res0, single):
        assert (a is None and b is None) or np.array_equal(a, b)
    assert res0[3] is None      # the dropped PEB unit stays empty


def test_ranks_seeded_differently_still_agree_on_one_table():
    """the seed base is drawn on every rank instead of being broadcast.  Ranks that were not seeded alike --
    the reference's normal, unseeded usage -- draw different bases: the run completes, a unit is seeded from the
    base of the rank that owns it, every rank ends with the same table, and the bases the all_gather carried
    are on record (round 3 raised here, so an unseeded multi-GPU calc_probs always failed)"""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, 0)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort(key=lambda t: t[0])
    (_, res0, own0), (_, res1, own1) = got
    assert own0 == own1
    for a, b in zip(res0, res1):
        assert (a is None and b is None) or np.array_equal(a, b)
    # rank r's units are what a single process computes from rank r's base (np.random.seed(100 + r))
    units = _fake_units()
    live = [k for k, u in enumerate(units) if u[4] is not None]
    for r in range(world):
        np.random.seed(100 + r)
        sharding.per_unit_seed = True
        try:
            single = _flatten(sharding.run_units(_fake_units(), verbose=0))
        finally:
            sharding.per_unit_seed = False
        mine = [k for i, k in enumerate(live) if own0[i] == r]
        assert mine
        for k in mine:
            assert np.array_equal(res0[k], single[k])


def test_single_process_consumes_the_stream_sequentially():
    """world == 1 without per-unit seeding: unit k sees the stream where unit k-1 left it, as in
    the reference's sequential scenario loop"""
    np.random.seed(7)
    res = _flatten(sharding.run_units(_fake_units(n_stars=1, drop=()), verbose=0))
    np.random.seed(7)
    units = _fake_units(n_stars=1, drop=())
    for u, r in zip(units, res):
        want = sharding._record(u[4]())
        assert np.array_equal(want, r)


def test_calc_probs_end_to_end_on_host_fakes(monkeypatch):
    """calc_probs table layout, scenario order, FPP/NFPP bookkeeping and drop_scenario handling
    (triceratops.py:673-1485), device calls replaced by the CPU stand-in"""
    import pandas as pd
    from helpers import GOLD, gold, install_cpu_device_fakes
    install_cpu_device_fakes(monkeypatch)
    from triceratops_amd.triceratops import target
    g = gold("lnz_cases.npz")
    stars = pd.DataFrame({
        "ID": [111, 222, 333], "Tmag": [10.4, 13.0, 15.5], "Jmag": [9.5, 12.1, 14.6],
        "Hmag": [9.1, 11.7, 14.2], "Kmag": [9.0, 11.6, 14.1], "ra": [10.0, 10.01, 10.02],
        "dec": [-5.0, -5.01, -5.02], "mass": [0.82, 0.6, np.nan], "rad": [0.8, 0.58, np.nan],
        "Teff": [5100.0, 4000.0, np.nan], "plx": [14.2, 3.0, np.nan],
        "fluxratio": [0.95, 0.04, 0.01], "tdepth": [0.0074, 0.17, 0.0]})
    tg = target(111, np.array([1]), stars=stars, trilegal_fname=os.path.join(GOLD, "trilegal_synth.csv"))
    np.random.seed(5)
    tg.calc_probs(g["time"], g["flux"], float(g["sigma"][0]), 3.3, N=400, parallel=True,
                  drop_scenario=["SEB"], verbose=0)
    assert list(tg.probs.scenario) == ["TP", "EB", "EBx2P", "PTP", "PEB", "PEBx2P", "STP", "SEB",
                                      "SEBx2P", "DTP", "DEB", "DEBx2P", "BTP", "BEB", "BEBx2P",
                                      "NTP", "NEB", "NEBx2P"]
    assert len(tg.lnZ) == 3 * 2 + 12 and tg.lnZ[7] == -np.inf and tg.lnZ[8] == -np.inf
    assert list(tg.star_num[:15]) == [1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 2, 2, 2]
    assert list(tg.probs.ID[15:]) == [222, 222, 222]
    assert abs(tg.probs.prob.sum() - 1) < 1e-12
    assert abs(tg.FPP - (1 - tg.probs.prob[[0, 3, 9]].sum())) < 1e-15
    assert abs(tg.NFPP - tg.probs.prob[15:].sum()) < 1e-15
    assert tg.FPP_degenerate is False
    # same seed, sequential stream -> identical to calling the lnZ_* in the reference's order
    from triceratops_amd import marginal_likelihoods as ml
    from triceratops_amd.funcs import renorm_flux
    np.random.seed(5)
    fl, fe = renorm_flux(g["flux"], float(g["sigma"][0]), 0.95)
    r = ml.lnZ_TTP(g["time"], fl, fe, 3.3, 0.82, 0.8, 5100.0, 0.0, 400, True, "TESS", False, 0.00139, 20)
    assert r["lnZ"] == tg.lnZ[0] and r["R_p"][0] == tg.probs.R_p[0]



# ---------------------------------------------------------------------------------------
# calc_probs_many: several targets, one unit list, one all_gather
def _two_jobs():
    import pandas as pd
    from helpers import GOLD, gold
    from triceratops_amd.triceratops import target
    g = gold("lnz_cases.npz")

    def stars(ids, fr):
        n = len(ids)
        return pd.DataFrame({
            "ID": ids, "Tmag": [10.4, 13.0, 15.5][:n], "Jmag": [9.5, 12.1, 14.6][:n],
            "Hmag": [9.1, 11.7, 14.2][:n], "Kmag": [9.0, 11.6, 14.1][:n], "ra": [10.0, 10.01, 10.02][:n],
            "dec": [-5.0, -5.01, -5.02][:n], "mass": [0.82, 0.6, np.nan][:n], "rad": [0.8, 0.58, np.nan][:n],
            "Teff": [5100.0, 4000.0, np.nan][:n], "plx": [14.2, 3.0, np.nan][:n],
            "fluxratio": fr, "tdepth": [0.0074, 0.17, 0.3][:n]})

    tri = os.path.join(GOLD, "trilegal_synth.csv")
    a = target(111, np.array([1]), stars=stars([111, 222, 333], [0.95, 0.04, 0.01]), trilegal_fname=tri)
    b = target(777, np.array([2]), stars=stars([777, 888], [0.9, 0.1]), trilegal_fname=tri)
    sig = float(g["sigma"][0])
    jobs = [(a, dict(time=g["time"], flux_0=g["flux"], flux_err_0=sig, P_orb=3.3, N=300, parallel=True,
                     drop_scenario=["SEB"])),
            (b, dict(time=g["time"][::2], flux_0=g["flux"][::2], flux_err_0=sig, P_orb=[3.0, 3.6], N=200,
                     parallel=True, drop_scenario=["DTP", "DEB", "BTP", "BEB"]))]
    return jobs


def _tables(jobs):
    return [np.concatenate([tg.lnZ, tg.probs.prob.values, [tg.FPP, tg.NFPP], tg.probs.R_p.values])
            for tg, _ in jobs]


def test_calc_probs_many_equals_sequential_calls_on_one_stream(monkeypatch):
    from helpers import install_cpu_device_fakes
    from triceratops_amd.triceratops import calc_probs_many
    install_cpu_device_fakes(monkeypatch)
    jobs = _two_jobs()
    np.random.seed(31)
    out = calc_probs_many(jobs)
    assert out[0] is jobs[0][0] and len(jobs[0][0].lnZ) == 21 and len(jobs[1][0].lnZ) == 18
    many = _tables(jobs)
    seq = _two_jobs()
    np.random.seed(31)
    for tg, kw in seq:
        tg.calc_probs(verbose=0, **kw)
    for m, s in zip(many, _tables(seq)):
        assert np.array_equal(m, s, equal_nan=True)
    assert np.isfinite(many[0][0]) and np.isfinite(many[1][0]) and jobs[1][0].lnZ[9] == -np.inf


def _many_worker(rank, world, port, q):
    import pytest as _pytest
    import triceratops_amd
    triceratops_amd.set_sampling("numpy")
    from helpers import install_cpu_device_fakes
    from triceratops_amd.triceratops import calc_probs_many
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mp_ = _pytest.MonkeyPatch()
    install_cpu_device_fakes(mp_)
    jobs = _two_jobs()
    np.random.seed(77)
    calc_probs_many(jobs)
    units = [u for tg, kw in _two_jobs() for u in tg._prepare(**{k: v for k, v in kw.items()})[0]]
    live = [u for u in units if u[4] is not None]
    owners = sharding.schedule([sharding._COST.get(u[5], 1.0) * u[6] for u in live], world, [u[8] for u in live])
    q.put((rank, _tables(jobs), owners))
    dist.barrier()
    dist.destroy_process_group()
    mp_.undo()


@pytest.mark.parametrize("world", [2, 4])
def test_calc_probs_many_gloo_matches_single_process(monkeypatch, world):
    from helpers import install_cpu_device_fakes
    from triceratops_amd.triceratops import calc_probs_many
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_many_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    t0, own0 = got[0][1], got[0][2]
    assert all(g[2] == own0 for g in got) and set(own0) == set(range(world))
    # the larger job's units weigh more: the ranks' loads are balanced by weight, not by count
    for g in got[1:]:
        for a, b in zip(t0, g[1]):
            assert np.array_equal(a, b, equal_nan=True)
    install_cpu_device_fakes(monkeypatch)
    jobs = _two_jobs()
    np.random.seed(77)
    sharding.per_unit_seed = True
    try:
        calc_probs_many(jobs)
    finally:
        sharding.per_unit_seed = False
    for a, b in zip(t0, _tables(jobs)):
        assert np.array_equal(a, b, equal_nan=True)


@pytest.mark.gpu
def test_calc_probs_many_on_gpu_equals_sequential_calls():
    from triceratops_amd.triceratops import calc_probs_many
    jobs = _two_jobs()
    np.random.seed(31)
    calc_probs_many(jobs)
    seq = _two_jobs()
    np.random.seed(31)
    for tg, kw in seq:
        tg.calc_probs(verbose=0, **kw)
    for m, s in zip(_tables(jobs), _tables(seq)):
        assert np.array_equal(m, s, equal_nan=True)


def test_schedule_balances_the_64_toi_batch_over_8_ranks():
    """BASELINE config 4: 64 TOIs x (9 target calls + 2 nearby calls) = 704 lnZ_* units holding
    64 x 18 = 1152 scenarios, dealt to 8 ranks by cost"""
    keys = ["TP", "EB", "PTP", "PEB", "STP", "SEB", "DTP", "DEB", "BTP", "BEB", "NTP", "NEB"]
    rng = np.random.default_rng(0)
    costs = []
    for _ in range(64):
        w = float(rng.choice([100, 200, 2000])) * 1e6        # n_time x N of that TOI
        costs += [sharding._COST[k] * w for k in keys[:10]] + [sharding._COST[k] * w for k in keys[10:]]
    owner = sharding.schedule(costs, 8)
    assert len(owner) == 64 * 12 and set(owner) == set(range(8))
    load = np.bincount(owner, weights=costs, minlength=8)
    assert load.max() / load.mean() < 1.02                   # LPT: within 2 % of perfect balance
    assert owner == sharding.schedule(costs, 8)               # deterministic: every rank computes the same table
    # one rank: everything on rank 0
    assert set(sharding.schedule(costs, 1)) == {0}
    # with the (job, star) of every unit the ranks get WHOLE TOIs, still balanced: a rank prepares, uploads and
    # enqueues only the light curves and star tables of its own targets
    groups = [(j, 0 if i < 10 else 1) for j in range(64) for i in range(12)]
    owner = sharding.schedule(costs, 8, groups)
    assert owner == sharding.schedule(costs, 8, groups)
    load = np.bincount(owner, weights=costs, minlength=8)
    assert load.max() / load.mean() < 1.03
    for j in range(64):
        assert len(set(owner[12 * j:12 * j + 12])) == 1       # no TOI is split
    assert sorted(np.bincount([owner[12 * j] for j in range(64)], minlength=8)) == [8] * 8 or \
        load.max() / load.mean() < 1.03                        # (sizes differ: counts may, the loads may not)


def test_schedule_deals_like_sized_tois_eight_to_a_rank():
    """BASELINE configs[3] as bench.py builds it: 64 TOIs of one size, two stars each (10 + 2 calls)"""
    keys = ["TP", "EB", "PTP", "PEB", "STP", "SEB", "DTP", "DEB", "BTP", "BEB", "NTP", "NEB"]
    costs = [sharding._COST[k] * 2e8 for _ in range(64) for k in keys]
    groups = [(j, 0 if i < 10 else 1) for j in range(64) for i in range(12)]
    owner = sharding.schedule(costs, 8, groups)
    for r in range(8):
        mine = [k for k, o in enumerate(owner) if o == r]
        assert len(mine) == 96 and len({groups[k][0] for k in mine}) == 8
    # few groups: falls back to stars, then to single calls, and a star dearer than a rank's share is split
    costs = [sharding._COST[k] for k in keys[:10]] + [1.0, 1.7] * 20
    groups = [(0, 0)] * 10 + [(0, 1 + i // 2) for i in range(40)]
    owner = sharding.schedule(costs, 8, groups)
    load = np.bincount(owner, weights=costs, minlength=8)
    assert load.max() / load.mean() < 1.10 and len(set(owner[:10])) > 1


def test_pieces_are_contiguous_runs_of_one_target():
    """what one rank hands to its streams (sharding._pieces): whole targets when there are at least as many as streams,
    else a target's calls in a few contiguous runs balanced by cost -- every unit exactly once, order kept"""
    keys = ["TP", "EB", "PTP", "PEB", "STP", "SEB", "DTP", "DEB", "BTP", "BEB"]
    units = [(0, (), 1, 1, 1, k, 1.0, 1, (0, 0)) for k in keys]
    for i in range(1, 21):
        units += [(0, (), 1, 1, 1, "NTP", 1.0, 1, (0, i)), (0, (), 1, 1, 1, "NEB", 1.0, 1, (0, i))]
    mine = list(range(len(units)))
    pieces = sharding._pieces(units, mine, 4)
    assert [k for p in pieces for k in p] == mine and 4 <= len(pieces) <= 8
    assert max(len(p) for p in pieces) <= 16                      # (the library chains up to 16 calls)
    many = []
    for j in range(6):
        many += [(0, (), 1, 1, 1, k, 1.0, 1, (j, 0)) for k in keys] + [(0, (), 1, 1, 1, "NTP", 1.0, 1, (j, 1)),
                                                                      (0, (), 1, 1, 1, "NEB", 1.0, 1, (j, 1))]
    pieces = sharding._pieces(many, list(range(len(many))), 4)
    assert [len(p) for p in pieces] == [12] * 6                   # one run (one launch chain) per target
    assert sharding._pieces(units[:1], [0], 4) == [[0]] and sharding._pieces(units, [], 4) == []


def _four_jobs():
    a = _two_jobs()
    b = _two_jobs()
    for tg, _ in b:
        tg.stars["ID"] = tg.stars["ID"] + 5000
    return a + b


def _defer_worker(rank, world, port, q):
    import pytest as _pytest
    import triceratops_amd
    triceratops_amd.set_sampling("numpy")
    from helpers import install_cpu_device_fakes
    from triceratops_amd.triceratops import calc_probs_many
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mp_ = _pytest.MonkeyPatch()
    install_cpu_device_fakes(mp_)
    jobs = _four_jobs()
    for tg, _ in jobs:
        tg.FPP = "stale"                      # (a result of an earlier call must not survive a deferred table)
    np.random.seed(77)
    calc_probs_many(jobs)
    pending = [tg.__dict__.get("_pending_finish") is not None for tg, _ in jobs]
    has_fpp = ["FPP" in tg.__dict__ for tg, _ in jobs]
    own = sorted(sharding.last_own_jobs)
    # a target whose table is still to be filled pickles and copies like any other (advisor, round 5: it used to hold the
    # units' closures -- "Can't pickle local object" -- and views into the whole batch's table)
    import copy
    import pickle
    portable = True
    for k, (tg, _) in enumerate(jobs):
        if pending[k]:
            slim = tg.__dict__["_pending_finish"]
            portable = portable and all(not callable(x) for u in slim[0] for x in u) and \
                all(r is None or r.base is None for r in slim[1])
            clone = pickle.loads(pickle.dumps(tg))
            twin = copy.deepcopy(jobs[k][0])
            portable = portable and clone.FPP == tg.FPP and twin.NFPP == tg.NFPP and \
                np.array_equal(clone.lnZ, tg.lnZ, equal_nan=True)
            break
    q.put((rank, _tables(jobs), pending, has_fpp, own, portable))
    dist.barrier()
    dist.destroy_process_group()
    mp_.undo()


def test_tables_of_other_ranks_targets_are_filled_on_first_read(monkeypatch):
    """calc_probs_many on several ranks: after the all_gather every rank holds every record, fills the tables of the
    targets it evaluated itself at once and the others' when one of their results is first read (target.__getattr__:
    filling all 64 tables of a batch was a quarter of a rank's host path on eight ranks).  Four targets on two ranks:
    a target no unit of which the rank evaluated is pending until read, holds no result of an earlier call, and
    reads the same as everywhere else and as in one process."""
    from helpers import install_cpu_device_fakes
    from triceratops_amd.triceratops import calc_probs_many
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_defer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    owns = [set(g[4]) for g in got]
    # (the two larger targets are dearer than 3/4 of a rank's share and are dealt star by star: both ranks hold part of them)
    assert owns[0] | owns[1] == {0, 1, 2, 3} and all(len(o) < 4 for o in owns)
    for r, (_, tables, pending, has_fpp, _own, portable) in enumerate(got):
        assert portable
        assert pending == [j not in owns[r] for j in range(4)]
        assert has_fpp == [j in owns[r] for j in range(4)]
        for a, b in zip(got[0][1], tables):
            assert np.array_equal(a, b, equal_nan=True)
    install_cpu_device_fakes(monkeypatch)
    jobs = _four_jobs()
    np.random.seed(77)
    sharding.per_unit_seed = True
    try:
        calc_probs_many(jobs)
    finally:
        sharding.per_unit_seed = False
    for a, b in zip(got[0][1], _tables(jobs)):
        assert np.array_equal(a, b, equal_nan=True)


def test_deferred_table_warns_where_it_is_deferred_not_where_it_is_read():
    """the reference's RuntimeWarnings for degenerate evidences (triceratops.py:1466-1478) belong to the calc_probs call:
    a table left to its first reader raises them when it is deferred, and reading it later raises nothing more"""
    import warnings
    from triceratops_amd.triceratops import target
    tg = target.__new__(target)
    from triceratops_amd.triceratops import _TARGET_CALLS
    units = [(j0, names, snum, 42, None, key) for key, names, j0, snum in _TARGET_CALLS]
    rec = [np.zeros((len(u[1]), len(sharding.RECORD_COLS))) for u in units]
    for r in rec:
        r[:, -1] = -np.inf
    with pytest.warns(RuntimeWarning, match="All scenario log-evidences are -inf"):
        tg._defer_finish(units, rec, 15)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert tg.FPP_degenerate is True and tg.FPP == 1.0
    rec[1][0, -1] = np.nan
    with pytest.warns(RuntimeWarning, match="Unexpected NaN"):
        tg._defer_finish(units, rec, 15)
    for r in rec:
        r[:, -1] = -3.0
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        tg._defer_finish(units, rec, 15)
        assert tg.FPP_degenerate is False


def test_gc_freeze_is_opt_in_and_can_be_released():
    """advisor, round 5: gc.freeze() is for good -- a library must not decide that for a long-lived service.  Off by
    default; with sharding.freeze_gc the first device pass freezes once and sharding.release() thaws"""
    import gc
    assert sharding.freeze_gc is False or os.environ.get("TRX_FREEZE_GC") == "1"
    before = gc.get_freeze_count()
    sharding._gc_frozen = True            # (as a device pass with freeze_gc leaves it)
    gc.freeze()
    assert gc.get_freeze_count() > before
    sharding.release()
    assert gc.get_freeze_count() == 0 and sharding._gc_frozen is False
    # the stream cap follows the chains' real scratch (7 GB a stream at N = 1e6, 15 GB from 3e6 on)
    assert 6.5e9 < sharding.stream_scratch_bytes(1_000_000) < 8e9
    assert 14e9 < sharding.stream_scratch_bytes(3_000_000) < 16e9 and 14e9 < sharding.stream_scratch_bytes(10_000_000) < 16e9
