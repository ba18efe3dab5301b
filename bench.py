#!/usr/bin/env python3
"""bench.py -- light-curve-point x sample evaluations per second of the marginal-likelihood hot path.

`python bench.py --gpus N --steps K --warmup W`.  With N > 1 and no WORLD_SIZE in the environment
the script starts its own N ranks (python -m torch.distributed.run, one process per GPU, RCCL,
rendezvous on 127.0.0.1) BEFORE anything touches the GPU and exits with their status; under an
external launcher it checks that WORLD_SIZE == --gpus.

--mode grid (default; BASELINE.json configs[1], the configuration the metric is quoted on)
    One step = a synthetic 2000-point light curve x all 18 scenario families (TP/EB/EBx2P x
    T,P,S,D,B,N) x N_samples transiting rows, fp64: for every family the fused likelihood kernel
    (trx_lnl_batch) then the log-mean-exp evidence (trx_lnz_from_halfchi2).  Inputs are resident in
    HBM before the timed region.  N > 1: every rank runs the same per-GPU work on its own rows
    (weak scaling: scenarios x TOIs shard with no data-path collective) and the step ends with ONE
    all_gather of the per-scenario lnZ vector.
--mode batch (BASELINE.json configs[3]: 64 synthetic TOIs x 18 scenarios x N = 1e6)
    One step = calc_probs_many over the whole batch with the scenario pipeline on the device: the
    64 x 11 lnZ_* units (64 x 18 scenarios) are dealt to the ranks by cost (LPT), every rank draws
    and evaluates its own units, ONE all_gather assembles all tables (strong scaling).

One JSON line on rank 0 (contract in the task statement) with these extra objects:
  roofline     dominant kernel = the fused likelihood kernel; it is fp64-VALU bound, not HBM/MFMA
               (SURVEY.md 8d).  achieved = EXECUTED model evaluations (the kernel's own census,
               measured in this run) x the plain operation count of one evaluation / mean launch
               duration measured with events on the launch stream: always < peak.  The figure that
               prices the launch as if all S sub-exposures had been evaluated is reported separately
               (plain_algorithm_*; it can exceed the peak -- that is the Gauss-node shortcut, not
               throughput), and so is the same workload timed in this run with the shortcut off
               (all_subexposures).  traffic = HBM bytes per launch from rocprofv3 PMC passes
               (FETCH_SIZE / WRITE_SIZE, separate passes) collected by this run in child processes.
  kernels      the two HBM-bound reductions (chi^2 over a materialised grid, log-mean-exp).
  shapes       the same 18 families on the reference's REAL operating points, each with its own census
               and fraction: 100 and 200 binned points (batched variant of the kernel) and an irregular
               2000-point grid (every stamp jittered: no centre-value stencil) -- config.representative
               names the irregular figure; the headline workload is BASELINE's uniform grid.
  cpu_baseline the CPU oracle (a port: the reference's pytransit engine is not installable), compiled on
               this box with -O3 -march=native, on a bounded sample of the same rows: all usable host cores
               (value; affinity mask and cgroup quota stated), single thread, the same with the GPU kernels'
               transit-window early-out (like for like with the GPU's algorithm), and the reference-shaped
               numpy pipeline over a materialised (n, n_time) grid.
  e2e          end-to-end calc_probs() wall-clock (the second half of the BASELINE metric) on
               TOI-465.01 (BASELINE configs[2]: real light curve + contrast curve, 1 + 20 stars,
               75 scenarios, N = 1e6): the default sampling mode (device) first, then the validation modes.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

N_TIME = 2000
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK_TF = 78.6       # MI355X fp64 vector peak = 1/2 of the 157.3 TF fp32 vector peak
# algorithmic fp64 operations of ONE model evaluation of the PLAIN restatement (oracle/trx_oracle.c),
# counted one per add/sub/mul/div/sqrt/compare and one per libm call:
F_ORBIT = 100.0                # offset+mean anomaly 8, Kepler (guess 8 + 3 Halley iterations x 22) 74, position 15
F_MA = 200.0                   # case analysis 40, two cel integrals (4 iterations x 14 + 10) x 2, combination 19


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mode", choices=["grid", "batch"], default="grid")
    ap.add_argument("--n-samples", type=int, default=100_000, help="grid mode: rows per scenario family")
    ap.add_argument("--n-time", type=int, default=None, help="light-curve points (grid: 2000, batch: 200)")
    ap.add_argument("--tois", type=int, default=64, help="batch mode: number of synthetic TOIs")
    ap.add_argument("--batch-n", type=int, default=1_000_000, help="batch mode: Monte-Carlo draws per scenario")
    ap.add_argument("--threads", type=int, default=None,
                    help="host threads evaluating scenarios side by side (default 1: one thread already deals the "
                         "calls to sharding.streams HIP streams and waits once)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="budget of each of the three CPU legs")
    ap.add_argument("--no-extras", action="store_true", help="skip the HBM-kernel, census and e2e legs")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-batch-leg", action="store_true",
                    help="grid mode: skip the `batch` object (the configs[3] strong-scaling step measured in the same job)")
    ap.add_argument("--pmc", choices=["auto", "off"], default="auto",
                    help="collect FETCH_SIZE / WRITE_SIZE of the dominant kernel with rocprofv3 child runs (N = 1)")
    # test hook for 1-GPU boxes: run the N>1 control flow (rendezvous, barrier, gather, max-reduce)
    # with every rank on cuda:0 and a gloo process group (RCCL refuses two ranks on one device)
    ap.add_argument("--debug-single-device", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-rccl-world1", action="store_true",
                    help="one GPU: do not set up the one-rank RCCL group that sends one untimed batch step through the collective")
    ap.add_argument("--all-subexposures", action="store_true",
                    help="switch the reduced-node exposure average off for the timed steps")
    ap.add_argument("--fp32-model", action="store_true",
                    help="BASELINE config 5: fp32 Mandel-Agol arithmetic, fp64 orbit/chi^2/log-mean-exp")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------
def launch_ranks_if_needed(args):
    """--gpus N without a launcher: become the launcher.  Runs before torch is imported, so this
    process never initialises the GPU; the ranks are ordinary child processes."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        if int(env_world) != args.gpus:
            sys.exit("bench.py: --gpus %d but WORLD_SIZE=%s: launch with --nproc-per-node %d or drop one of them"
                     % (args.gpus, env_world, args.gpus))
        return
    if args.gpus <= 1:
        return
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def collect_pmc(args):
    """HBM traffic of the dominant kernel: two rocprofv3 --pmc child runs of this same script (one
    counter per pass: FETCH_SIZE and WRITE_SIZE do not fit one pass), each one step of the same
    workload.  Called before this process touches the GPU.  Returns a dict or None."""
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None          # this process is itself being profiled: no nested profiler runs
    child = [sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
             "--no-extras", "--no-batch-leg", "--pmc", "off", "--n-samples", str(args.n_samples), "--n-time", str(args.n_time)]
    if args.fp32_model:
        child.append("--fp32-model")
    if args.all_subexposures:
        child.append("--all-subexposures")
    got = {}
    tmp = tempfile.mkdtemp(prefix="trx_pmc_", dir="/tmp")
    try:
        # third pass: the fp64 instructions the launch issues (wave instructions; one pass holds all four)
        insts = ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64")
        for counter in ("FETCH_SIZE", "WRITE_SIZE", "INSTS"):
            out = os.path.join(tmp, counter)
            names = list(insts) if counter == "INSTS" else [counter]
            cmd = [exe, "--pmc"] + names + ["--kernel-trace", "--output-format", "csv", "-d", out, "--"] + child
            try:
                p = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True,
                                   text=True, timeout=420)
            except subprocess.TimeoutExpired:
                return None
            if p.returncode != 0:
                return None
            # one likelihood launch = rowc_kernel (row constants to scratch) + cells_kernel<lnl> (one or both
            # of its instantiations without / with the centre-value stencil; the one that does not apply
            # returns at once): everything is counted and divided by the launches of the child's one step
            from triceratops_amd import synth as _synth
            launches, seen, totals = len(_synth.FAMILIES), 0, {k: 0.0 for k in names}
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row["Counter_Name"] not in totals:
                            continue
                        name = row["Kernel_Name"].replace(" ", "")
                        if "cells_kernel<0" in name or "rowc_kernel" in name:
                            totals[row["Counter_Name"]] += float(row["Counter_Value"])
                            seen += ("cells_kernel<0" in name) and row["Counter_Name"] == names[0]
            if seen < launches:
                if counter == "INSTS":
                    continue                      # the traffic passes stand on their own
                return None
            if counter == "INSTS":
                got[counter] = {k: v / launches for k, v in totals.items()}
            else:
                got[counter] = (totals[counter] / launches, launches)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    # gfx950: FETCH_SIZE counts 64 B per 128-B request -> x2 on the read side; KB units
    bytes_ = (2.0 * got["FETCH_SIZE"][0] + got["WRITE_SIZE"][0]) * 1024.0
    issued = None
    if "INSTS" in got:
        i_ = got["INSTS"]
        issued = {"fma_f64": i_["SQ_INSTS_VALU_FMA_F64"], "mul_f64": i_["SQ_INSTS_VALU_MUL_F64"],
                  "add_f64": i_["SQ_INSTS_VALU_ADD_F64"], "trans_f64": i_["SQ_INSTS_VALU_TRANS_F64"],
                  # wave instructions x 64 lanes, an fma = 2 flops; idle lanes of a wave are counted as issued
                  "flop_per_launch": 64.0 * (2.0 * i_["SQ_INSTS_VALU_FMA_F64"] + i_["SQ_INSTS_VALU_MUL_F64"]
                                             + i_["SQ_INSTS_VALU_ADD_F64"])}
    return {"bytes_per_launch": bytes_, "fetch_size_kb": got["FETCH_SIZE"][0], "write_size_kb": got["WRITE_SIZE"][0],
            "launches": got["FETCH_SIZE"][1], "issued": issued,
            "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child runs of this command, in this run; "
                      "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 read-side factor, MI355X_MICROARCH.md)"}


def committed_traffic(n_time, n_rows):
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    for rec in json.load(open(path)):
        if rec["n_time"] == n_time and rec["n_samples"] == n_rows:
            return {"bytes_per_launch": (2.0 * rec["fetch_size_kb"] + rec["write_size_kb"]) * 1024.0,
                    "fetch_size_kb": rec["fetch_size_kb"], "write_size_kb": rec["write_size_kb"],
                    "source": "NOT measured in this run: committed profile " + rec["source"]}
    return None


# ---------------------------------------------------------------------------------------------
def main():
    args = parse()
    if args.n_time is None:
        args.n_time = N_TIME if args.mode == "grid" else 200
    launch_ranks_if_needed(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    extras = not args.no_extras
    traffic = None
    if world == 1 and args.mode == "grid" and extras and args.pmc == "auto":
        try:
            traffic = collect_pmc(args)          # child processes, before this one touches the GPU
        except Exception:
            traffic = None

    import torch
    import torch.distributed as dist
    from triceratops_amd import _lib

    debug_one = args.debug_single_device
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(0 if debug_one else local_rank)
        if debug_one:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    _lib.require_gpu()
    device = torch.device("cuda", local_rank if (world > 1 and not debug_one) else 0)
    ctx = dict(args=args, world=world, rank=rank, device=device, debug_one=debug_one, extras=extras,
               traffic=traffic)
    out = run_grid(ctx) if args.mode == "grid" else run_batch(ctx)
    if world > 1:
        dist.barrier()
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line is the LAST thing on stdout: RCCL prints a version banner through C stdio when NCCL_DEBUG is
        # WARN or VERSION (it is on the GPU boxes), and libc's buffer would otherwise be flushed after Python's
        import ctypes
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:                  # noqa: BLE001
            pass
        print(json.dumps(out))
        sys.stdout.flush()


def _sync(ctx):
    import torch
    import torch.distributed as dist
    torch.cuda.synchronize(ctx["device"])
    if ctx["world"] > 1:
        dist.barrier()
    torch.cuda.synchronize(ctx["device"])


def _max_over_ranks(ctx, elapsed):
    import torch
    import torch.distributed as dist
    if ctx["world"] > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if ctx["debug_one"] else ctx["device"])
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te[0])
    return elapsed


def census(rows_of, fams, t_d, n_sample=512, plan_flags=0):
    """per-cell statistics of the model kernel on a sample of every family's rows: the number of
    model evaluations the kernel spends (TRX_FLAG_COUNT_EVALUATIONS of trx_flux_grid) and the occulted fraction;
    plan_flags: the result-neutral flags of the launches being priced (TRX_FLAG_ALL_SUBEXPOSURES / _NO_STENCIL)"""
    from triceratops_amd import _lib, synth
    evals, p_in = [], []
    for i, fam in enumerate(fams):
        blk = rows_of(i)[:, :n_sample].contiguous()
        g, _ = _lib.flux_grid(fam[1], plan_flags, t_d, blk, synth.EXPTIME, synth.NSAMPLES, False)
        p_in.append(float((g < 1.0).double().mean()))
        c, _ = _lib.flux_grid(fam[1], plan_flags | _lib.FLAG_COUNT_EVALUATIONS, t_d, blk, synth.EXPTIME, synth.NSAMPLES, False)
        evals.append(float(c.mean()))
    return float(np.mean(evals)), float(np.mean(p_in))


def hbm_kernels(ctx, f_d, n_time):
    """the two HBM-bound reductions at sizes past the caches"""
    import torch
    from triceratops_amd import _lib, synth
    device = ctx["device"]

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize(device)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize(device)
        return a.elapsed_time(b) * 1e-3 / reps

    kernels = {}
    n_grid = 200_000
    grid = torch.rand((n_grid, n_time), dtype=torch.float64, device=device)
    dt = timed(lambda: _lib.chi2_grid(f_d, grid, synth.SIGMA))
    gb = (grid.numel() * 8 + n_grid * 8) / 1e9
    kernels["chi2_grid_kernel"] = {"bound": "hbm", "bytes": gb * 1e9, "ms": dt * 1e3, "GBps": gb / dt,
                                   "frac": gb / dt / HBM_PEAK_GBS}
    del grid
    big = torch.empty(400_000_000, dtype=torch.float64, device=device).uniform_(-3000.0, -1.0)
    dt = timed(lambda: _lib.log_mean_exp(big, big.numel()))
    gb = big.numel() * 8 / 1e9
    kernels["lme_partial_kernel"] = {"bound": "hbm", "bytes": gb * 1e9, "ms": dt * 1e3, "GBps": gb / dt,
                                     "frac": gb / dt / HBM_PEAK_GBS,
                                     "input": "4e8 log-weights ~ U(-3000, -1) (SURVEY 8d stress vector)"}
    del big
    return kernels


def shapes_leg(ctx, rows_d, fams, reps=2):
    """The 18 families at the reference's real operating points (the same parameter rows; the timed workload
    above is BASELINE's uniform 2000-point grid): 100 and 200 binned points, and 2000 irregular stamps."""
    import torch
    from triceratops_amd import _lib, synth
    device, n_rows = ctx["device"], rows_d[0].shape[1]
    rng = np.random.default_rng(synth.SEED + 77)
    out = {}
    h = torch.empty(n_rows, dtype=torch.float64, device=device)
    for key, n_time, jitter in (("n100", 100, False), ("n200", 200, False), ("n2000_irregular", 2000, True)):
        t = synth.time_grid(n_time)
        if jitter:      # every stamp moved by up to +-0.3 of the spacing (a folded light curve is not a linspace)
            t = np.sort(t + rng.uniform(-0.3, 0.3, n_time) * (t[1] - t[0]))
        t_d = _lib.dev(t, device)
        curve, _ = _lib.flux_grid(_lib.MODEL_TP, 0, t_d, _lib.dev(synth.reference_tp_row(), device), synth.EXPTIME,
                                  synth.NSAMPLES, False)
        f_d = _lib.dev(synth.noisy_light_curve(rng, curve[0].cpu().numpy()), device)

        def one_pass(evs=None):
            for i, (name, model, is_host, has_comp) in enumerate(fams):
                flags = ((_lib.FLAG_COMPANION_IS_HOST if is_host else 0) | (_lib.FLAG_FP32_MODEL if ctx["args"].fp32_model else 0)
                         | _lib.FLAG_EVALUATE_EXCLUDED)
                if evs is not None:
                    evs[i][0].record()
                _lib.lnl_batch(model, flags, t_d, f_d, synth.SIGMA, rows_d[i], synth.EXPTIME, synth.NSAMPLES, out=h)
                if evs is not None:
                    evs[i][1].record()

        one_pass()
        torch.cuda.synchronize(device)
        ms = []
        for _ in range(reps):
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in fams]
            one_pass(evs)
            torch.cuda.synchronize(device)
            ms.append(np.mean([a.elapsed_time(b) for a, b in evs]))
        launch_s = float(np.min(ms)) * 1e-3
        evals_per_cell, p_in = census(lambda i: rows_d[i], fams, t_d)
        cells = float(n_time) * n_rows
        tf = evals_per_cell * (F_ORBIT + F_MA) * cells / launch_s / 1e12
        out[key] = {"n_time": n_time, "rows_per_family": n_rows, "uniform_grid": not jitter,
                    "kernel": "cells_kernel<lnl, %s>" % ("batches of rows per wave" if n_time < 320 else
                                                         "one row per wave, Gauss nodes (irregular stamps: no stencil)"),
                    "mean_launch_ms": launch_s * 1e3, "evals_per_s": cells / launch_s,
                    "model_evaluations_per_cell": evals_per_cell, "p_in": p_in,
                    "achieved_tflops": tf, "frac": tf / FP64_VALU_PEAK_TF}
    return out


# ---------------------------------------------------------------------------------------------
def run_grid(ctx):
    import torch
    import torch.distributed as dist
    from triceratops_amd import _lib, synth
    # every row this mode counts is evaluated: its likelihood calls carry TRX_FLAG_EVALUATE_EXCLUDED, i.e. they
    # do not skip the rows that lnL_EB_p's secondary-eclipse rule excludes anyway (a per-call flag: the
    # process-wide default -- skip them -- stays in force for the e2e leg below)
    EVAL_ALL = _lib.FLAG_EVALUATE_EXCLUDED
    args, world, rank, device = ctx["args"], ctx["world"], ctx["rank"], ctx["device"]
    # result-neutral choices are per-call flags (include/trx.h): the production library has no switches
    PLAN = _lib.FLAG_ALL_SUBEXPOSURES if args.all_subexposures else 0

    def gather(dst, src):
        if ctx["debug_one"]:                      # gloo moves host tensors
            buf = torch.empty(dst.numel(), dtype=dst.dtype)
            dist.all_gather_into_tensor(buf, src.cpu())
            dst.copy_(buf)
        else:
            dist.all_gather_into_tensor(dst, src)

    n_time, n_rows = args.n_time, args.n_samples
    fams = synth.FAMILIES
    # ---- synthetic inputs, generated on the host, resident in HBM before timing
    rng = np.random.default_rng(synth.SEED + 1000 * rank)
    t = synth.time_grid(n_time)
    t_d = _lib.dev(t, device)
    ref_d = _lib.dev(synth.reference_tp_row(), device)
    curve, _ = _lib.flux_grid(_lib.MODEL_TP, 0, t_d, ref_d, synth.EXPTIME, synth.NSAMPLES, False)
    flux = synth.noisy_light_curve(rng, curve[0].cpu().numpy())
    f_d = _lib.dev(flux, device)
    rows_h = [synth.family_rows(rng, fam, n_rows) for fam in fams]
    rows_d = [_lib.dev(r, device) for r in rows_h]
    lnprior_d = [_lib.dev(rng.uniform(-6.0, 0.0, n_rows), device) if fam[3] else None for fam in fams]
    h_d = [torch.empty(n_rows, dtype=torch.float64, device=device) for _ in fams]
    lnz_all = torch.empty(world * len(fams), dtype=torch.float64, device=device)
    lnsigma = float(np.log(synth.SIGMA))
    n_total = 10 * n_rows  # the masked rows are ~10% of the draws of a real lnZ_* call

    def events():
        return [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in fams]

    ev = [events() for _ in range(args.steps)]

    def step(evs=None, collective=True, plan=PLAN):
        lnz = []
        for i, (name, model, is_host, has_comp) in enumerate(fams):
            flags = (_lib.FLAG_COMPANION_IS_HOST if is_host else 0) | (_lib.FLAG_FP32_MODEL if args.fp32_model else 0) | EVAL_ALL | plan
            if evs is not None:
                evs[i][0].record()
            _lib.lnl_batch(model, flags, t_d, f_d, synth.SIGMA, rows_d[i], synth.EXPTIME,
                           synth.NSAMPLES, out=h_d[i])
            if evs is not None:
                evs[i][1].record()
            lnz.append(_lib.lnz_from_halfchi2(h_d[i], lnprior_d[i], n_total, lnsigma))
        mine = torch.cat(lnz)
        if world > 1 and collective:
            gather(lnz_all, mine)          # the single data-path collective (RCCL over xGMI)
        elif world == 1:
            lnz_all.copy_(mine)
        return lnz_all

    for _ in range(args.warmup):
        step()
    _sync(ctx)
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(ev[s])
    _sync(ctx)
    elapsed = _max_over_ranks(ctx, time.perf_counter() - t0)

    lnz_host = lnz_all.cpu().numpy()
    evals_per_step_per_gpu = float(n_time) * n_rows * len(fams)
    value = evals_per_step_per_gpu * world * args.steps / elapsed
    kern_ms = np.array([[a.elapsed_time(b) for (a, b) in ev[s]] for s in range(args.steps)])
    mean_launch_s = float(kern_ms.mean()) * 1e-3
    # the configs[3] step in the same job, on every rank count (collective inside: all ranks take part)
    batch = None
    if not args.no_batch_leg:
        try:
            batch = batch_object(ctx)
        except Exception as exc:                  # the grid line stands on its own (a failure here is the same on
            batch = {"error": repr(exc)}          # every rank: nobody is left waiting in a collective)
    if rank != 0:
        return None

    cells_per_launch = float(n_time) * n_rows
    n_par = np.mean([r.shape[0] for r in rows_h])
    alg_bytes_per_launch = (8.0 * n_par + 8.0) * n_rows + 16.0 * n_time
    roof = {"bound": "fp64_valu", "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s",
            "mean_launch_ms": mean_launch_s * 1e3, "cells_per_launch": cells_per_launch,
            "algorithmic_bytes_per_launch": alg_bytes_per_launch,
            # the row constants' round trip between rowc_kernel and cells_kernel (152 B per row: RowC, 19 doubles written and
            # read once): design traffic on top of the algorithmic bytes, counted in `traffic` (a variant in which the
            # waves derive the constants themselves was built in round 4 and not kept: profiles/experiments/README.md)
            "scratch_round_trip_bytes_per_launch": 2.0 * 152.0 * n_rows,
            "hbm_GBps_algorithmic": alg_bytes_per_launch / mean_launch_s / 1e9}
    kernels = {}
    shapes = None
    if ctx["extras"]:
        evals_per_cell, p_in = census(lambda i: rows_d[i], fams, t_d, plan_flags=PLAN)
        flop_exec = evals_per_cell * (F_ORBIT + F_MA)
        achieved = flop_exec * cells_per_launch / mean_launch_s / 1e12
        plain_flop = synth.NSAMPLES * (F_ORBIT + p_in * F_MA)
        plain_tf = plain_flop * cells_per_launch / mean_launch_s / 1e12
        assert 0.0 < achieved < FP64_VALU_PEAK_TF, "roofline.frac must be a fraction (got %g TFLOP/s)" % achieved
        roof.update({"achieved": achieved, "frac": achieved / FP64_VALU_PEAK_TF,
                     "model_evaluations_per_cell": evals_per_cell, "flop_per_model_evaluation": F_ORBIT + F_MA,
                     "p_in": p_in, "plain_algorithm_flop_per_cell": plain_flop,
                     "plain_algorithm_equivalent_tflops": plain_tf,
                     "plain_algorithm_equivalent_frac": plain_tf / FP64_VALU_PEAK_TF})
        # the same workload with every sub-exposure evaluated, timed here (one pass over the 18 families)
        if not args.all_subexposures and not args.fp32_model:
            step(collective=False, plan=_lib.FLAG_ALL_SUBEXPOSURES)          # rank 0 only from here on: no collective
            torch.cuda.synchronize(device)
            e2 = events()
            step(e2, collective=False, plan=_lib.FLAG_ALL_SUBEXPOSURES)
            torch.cuda.synchronize(device)
            ms_all = float(np.mean([a.elapsed_time(b) for (a, b) in e2]))
            tf_all = plain_flop * cells_per_launch / (ms_all * 1e-3) / 1e12
            roof["all_subexposures"] = {"mean_launch_ms": ms_all, "tflops": tf_all, "frac": tf_all / FP64_VALU_PEAK_TF,
                                        "evals_per_s": cells_per_launch / (ms_all * 1e-3)}
            # ... and with the centre-value stencil off (Gauss nodes in every cell): its own census and time
            ev_g, _ = census(lambda i: rows_d[i], fams, t_d, plan_flags=_lib.FLAG_NO_STENCIL)
            step(collective=False, plan=_lib.FLAG_NO_STENCIL)
            torch.cuda.synchronize(device)
            e3 = events()
            step(e3, collective=False, plan=_lib.FLAG_NO_STENCIL)
            torch.cuda.synchronize(device)
            ms_g = float(np.mean([a.elapsed_time(b) for (a, b) in e3]))
            tf_g = ev_g * (F_ORBIT + F_MA) * cells_per_launch / (ms_g * 1e-3) / 1e12
            roof["gauss_nodes_only"] = {"mean_launch_ms": ms_g, "model_evaluations_per_cell": ev_g, "tflops": tf_g,
                                        "frac": tf_g / FP64_VALU_PEAK_TF, "evals_per_s": cells_per_launch / (ms_g * 1e-3)}
        tr = ctx["traffic"]
        if tr is None:
            tr = committed_traffic(n_time, n_rows)
        roof["traffic"] = tr["bytes_per_launch"] if tr else None
        roof["traffic_detail"] = tr
        if tr and tr.get("issued"):
            # the hardware's view beside the executed-work fraction: fp64 flops the launch ISSUES (PMC child run
            # of this command), whatever they were spent on
            tf_i = tr["issued"]["flop_per_launch"] / mean_launch_s / 1e12
            roof["issued_fp64"] = {"tflops": tf_i, "frac": tf_i / FP64_VALU_PEAK_TF,
                                   "wave_instructions_per_launch": {k: tr["issued"][k] for k in
                                                                    ("fma_f64", "mul_f64", "add_f64", "trans_f64")}}
        kernels = hbm_kernels(ctx, f_d, n_time)
        shapes = shapes_leg(ctx, rows_d, fams)
    else:
        # no census in this run: price every cell at ONE model evaluation (a lower bound of the executed work)
        achieved = (F_ORBIT + F_MA) * cells_per_launch / mean_launch_s / 1e12
        roof.update({"achieved": achieved, "frac": achieved / FP64_VALU_PEAK_TF, "traffic": None})
    roof["note"] = ("dominant kernel cells_kernel<lnl, one row per wave> (+ its prologue rowc_kernel, 2 %% of the launch) is fp64-VALU bound (no MFMA shape, ~0.05 B/eval of HBM "
                    "traffic). achieved = executed model evaluations (census of this run) x %d plain operations / "
                    "launch time. plain_algorithm_* prices the launch as if all %d sub-exposures of every cell "
                    "had been evaluated; the kernel reaches those averages (to 1e-13) from 3-9 Gauss nodes, and on "
                    "this dense uniform grid from ONE evaluation per cell plus a 17-point stencil over the "
                    "neighbours' centre values wherever no limb contact is near -- so that figure is an "
                    "algorithmic-equivalence number, not a roofline fraction, and `frac` falls when a shortcut "
                    "removes executed work (gauss_nodes_only: the same run with the stencil off)"
                    % (int(F_ORBIT + F_MA), synth.NSAMPLES))

    cpu = None
    if not args.no_cpu_baseline and world == 1:
        cpu = cpu_baseline(t, flux, rows_h, fams, args.cpu_seconds)
    e2e = None
    if ctx["extras"] and not args.no_e2e and world == 1:
        e2e = e2e_calc_probs(args.threads or 4)

    return {
        "metric": "light-curve-point x sample evals/sec", "value": value,
        "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 model / f64 orbit+accumulators" if args.fp32_model else "f64", "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[1]: synthetic %d-point light curve, 18 "
                               "scenario families x %d transiting rows, nsamples=%d supersampling, "
                               "fused lnL + log-mean-exp" % (n_time, n_rows, synth.NSAMPLES),
                   "n_time": n_time, "n_samples": n_rows, "n_scenarios": len(fams),
                   "evals_per_step_per_gpu": evals_per_step_per_gpu,
                   "all_subexposures": bool(args.all_subexposures),
                   "representative": "shapes.n2000_irregular (2000 irregular stamps: what a folded, unbinned light "
                                     "curve looks like) and shapes.n100 / n200 (the reference's binned operating "
                                     "point); `value` is BASELINE configs[1]'s uniform grid, on which the "
                                     "centre-value stencil applies",
                   "parallelism": "scenario-sharded x%d, one all_gather of lnZ" % world},
        "roofline": roof, "shapes": shapes, "kernels": kernels, "cpu_baseline": cpu, "e2e": e2e, "batch": batch,
        "lnZ_checksum": float(np.nansum(lnz_host[np.isfinite(lnz_host)])),
    }



def batch_leg(ctx, tois, N, n_time, steps, warmup, fp32=False, before_timed=None):
    """The configs[3] step -- calc_probs_many over `tois` synthetic TOIs x 18 scenarios x N draws, the lnZ_* units
    dealt to the ranks (strong scaling) -- timed like the main loop: barrier + synchronize on both sides, max over
    ranks.  Returns (on every rank) elapsed seconds, the targets of the last step and the per-rank host timings:
    prepare (unit lists), enqueue (argument blocks + library calls of this rank's units), wait (streams), gather
    (the one collective), finish (tables of all targets), other (the rest of the step)."""
    import torch
    import torch.distributed as dist
    import triceratops_amd
    from triceratops_amd import sharding, synth
    world, rank, device = ctx["world"], ctx["rank"], ctx["device"]
    prev = triceratops_amd.get_sampling()
    triceratops_amd.set_sampling("device")
    if fp32:
        triceratops_amd.set_precision("fp32")
    # (a process of its own: the collector's permanent generation may be used -- opt-in since round 6, sharding.freeze_gc)
    sharding.freeze_gc = True
    try:
        tri = os.path.join(GOLD, "trilegal_synth.csv")
        cc = os.path.join(GOLD, "contrast_curve_synth.csv")
        # every rank builds the same jobs (tiny host tables + one light curve per TOI)
        jobs = synth.toi_jobs(tois, n_time=n_time, N=N, seed=synth.SEED, trilegal_fname=tri, contrast_curve_file=cc)
        small = synth.toi_jobs(min(tois, 2 * world), n_time=n_time, N=20000, seed=synth.SEED, trilegal_fname=tri,
                               contrast_curve_file=cc)

        def step(js, seed):
            np.random.seed(seed)
            torch.manual_seed(seed)          # ranks seeded alike: the run can be repeated (sharding's seed base)
            t0 = time.perf_counter()
            out = triceratops_amd.calc_probs_many(js)
            # Equal work at every world size (advisor, round 5): on several ranks calc_probs_many leaves the tables of
            # the targets a rank did not evaluate to their first reader; the step ends when EVERY target's table exists
            # where the results are read -- rank 0 -- as it does on one rank.  What that costs is `deferred_finish_s`.
            t1 = time.perf_counter()
            if rank == 0:
                for tg in out:
                    tg.FPP
            t2 = time.perf_counter()
            return out, t2 - t0, t2 - t1

        step(small, 1)                           # library load, tables, allocator
        for w in range(warmup):
            step(jobs, 10 + w)
        _sync(ctx)
        if before_timed is not None:
            before_timed()
        host = np.zeros(9)
        best_path = np.inf                       # smallest host path of a single step (a ratio of two such is what
        t0 = time.perf_counter()                 # tests compare: the mean carries whatever else the box was doing)
        gathers = sharding._dist() is not None   # (several ranks, or one rank sent through the collective: main())
        for s_ in range(steps):
            out, dt, dfin = step(jobs, 100 + s_)
            tm = sharding.timing
            waits = tm["wait_s"] + (tm["gather_s"] if gathers else 0.0)
            best_path = min(best_path, dt - waits)
            host[:7] += [tm["prepare_s"], tm["enqueue_s"], tm["wait_s"], tm["gather_s"] if gathers else 0.0,
                         tm["finish_s"] + dfin, dt, 0.0]
            host[8] += dfin
        _sync(ctx)
        elapsed = _max_over_ranks(ctx, time.perf_counter() - t0)
        host /= max(steps, 1)
        host[6] = host[5] - host[:5].sum()       # other
        host[7] = best_path
        per_rank = host[None, :]
        if world > 1:
            mine = torch.as_tensor(host, dtype=torch.float64, device="cpu" if ctx["debug_one"] else device)
            allr = torch.empty(world * host.size, dtype=torch.float64, device=mine.device)
            dist.all_gather_into_tensor(allr, mine)
            per_rank = allr.cpu().numpy().reshape(world, -1)
        # (finish_s includes deferred_finish_s: the tables rank 0 fills for the targets other ranks evaluated)
        names = ("prepare_s", "enqueue_s", "wait_s", "gather_s", "finish_s", "step_s", "other_s", "host_path_best_s",
                 "deferred_finish_s")
        timing = {k: [float(v) for v in per_rank[:, i]] for i, k in enumerate(names)}
        # what a rank's host does on the critical path of a step apart from waiting (for its GPU, for the others)
        timing["host_path_s"] = [float(per_rank[r, 0] + per_rank[r, 1] + per_rank[r, 4] + per_rank[r, 6])
                                 for r in range(per_rank.shape[0])]
        timing["calls"] = sharding.last_share["calls"]
        timing["stars"] = sharding.last_share["stars"]
        timing["jobs"] = sharding.last_share["jobs"]
        own_group = False
        saved_stdout = None
        if world == 1 and not ctx["args"].no_rccl_world1 and dist.is_available() and not dist.is_initialized():
            # (RCCL prints a version banner through C stdio when NCCL_DEBUG is WARN or VERSION -- it is on the GPU boxes --:
            # file descriptor 1 points at stderr while the group lives, so that stdout carries the JSON line and nothing else)
            sys.stdout.flush()
            saved_stdout = os.dup(1)
            os.dup2(2, 1)
            # A one-rank RCCL group, set up only NOW: a live RCCL communicator costs the timed steps 6-8 % (64-target step
            # 94-96 ms without one in the process, 102 with: profiles/r06/rccl_group_cost.txt -- it holds queues / CUs of
            # its own), and a single-GPU run has no use for one.
            try:
                store = "/tmp/trx_bench_pg_%d" % os.getpid()
                if os.path.exists(store):
                    os.remove(store)
                dist.init_process_group("nccl", init_method="file://" + store, rank=0, world_size=1,
                                        device_id=torch.device("cuda", 0))
                own_group = True
            except Exception as exc:                  # noqa: BLE001  (the bench line does not depend on it)
                sys.stderr.write("bench: one-rank RCCL group unavailable (%s: %s)\n" % (type(exc).__name__, exc))
        if world == 1 and dist.is_available() and dist.is_initialized():
            # One rank has nothing to gather and the timed steps skip the collective.  So that the RCCL branch of
            # sharding._run_units is executed on the hardware there is -- device tensors, header row, padding, through
            # all_gather_into_tensor of a one-rank "nccl" group -- one more step goes through it, not
            # timed as part of `value`, next to the same step without it (per-unit seeds on both sides: same tables).
            twin = synth.toi_jobs(tois, n_time=n_time, N=N, seed=synth.SEED, trilegal_fname=tri, contrast_curve_file=cc)
            sharding.per_unit_seed = True
            try:
                ref, dt_a, _ = step(twin, 999)
                fa = [float(tg.FPP) for tg in ref]
            finally:
                sharding.per_unit_seed = False
            sharding.collective_at_world_one = True
            try:
                got, dt_b, _ = step(twin, 999)
                fb = [float(tg.FPP) for tg in got]
                timing["rccl_world1"] = {"backend": dist.get_backend(), "gather_s": float(sharding.timing["gather_s"]),
                                         "step_s": dt_b, "step_without_collective_s": dt_a, "tables_equal": fa == fb}
                timing["gather_s"] = [float(sharding.timing["gather_s"])]
            finally:
                sharding.collective_at_world_one = False
                if own_group:
                    # (gone again before the legs that follow -- shapes, e2e -- are timed)
                    torch.cuda.synchronize()
                    dist.destroy_process_group()
        if saved_stdout is not None:
            try:
                import ctypes
                ctypes.CDLL(None).fflush(None)
            except Exception:              # noqa: BLE001
                pass
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
        return elapsed, out, jobs, timing
    finally:
        triceratops_amd.set_sampling(prev)
        if fp32:
            triceratops_amd.set_precision("fp64")


def triceratops_amd_hw_queues():
    """GPU_MAX_HW_QUEUES as the package requested it and whether the HIP runtime can have seen it (rank 0)"""
    import triceratops_amd
    return triceratops_amd.hw_queues()


def batch_object(ctx, steps=3, warmup=2):
    """`batch` object of the grid-mode line: the configs[3] strong-scaling step measured in the same job
    (two untimed steps first: the second step of a process still builds its argument blocks ~10 % slower than the
    fourth, profiles/r05/o_batch_timing.txt)"""
    args = ctx["args"]
    elapsed, out, jobs, timing = batch_leg(ctx, args.tois, args.batch_n, 200, steps, warmup, fp32=args.fp32_model)
    if ctx["rank"] != 0:
        return None
    from triceratops_amd import sharding
    n_scen = sum(len(tg.lnZ) for tg in out)
    return {"workload": "BASELINE.json configs[3]: %d synthetic TOIs x 18 scenarios x N=%d draws, 200-point light curves, "
                        "calc_probs_many, whole TOIs dealt to %d rank(s) by cost, one all_gather of the records"
                        % (args.tois, args.batch_n, ctx["world"]),
            "scaling": "strong", "n_gpus": ctx["world"], "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3, "calc_probs_per_s": args.tois * steps / elapsed,
            "nominal_evals_per_s": float(args.batch_n) * 200 * n_scen * steps / elapsed,
            "per_rank": timing, "streams": sharding.streams, "hw_queues": triceratops_amd_hw_queues(),
            "fpp_checksum": float(np.sum([float(tg.FPP) for tg in out]))}


# ---------------------------------------------------------------------------------------------
def run_batch(ctx):
    """BASELINE configs[3]: `--tois` synthetic TOIs x 18 scenarios x N draws through calc_probs_many"""
    import torch
    import triceratops_amd
    from triceratops_amd import _lib, synth
    args, world, rank, device = ctx["args"], ctx["world"], ctx["rank"], ctx["device"]
    triceratops_amd.set_sampling("device")
    if args.threads is None:
        args.threads = 1          # one host thread deals the calls to sharding.streams HIP streams and waits once
    triceratops_amd.set_threads(args.threads)
    if args.fp32_model:
        triceratops_amd.set_precision("fp32")
    if ctx["debug_one"]:
        import torch.distributed as dist    # noqa: F401  (gloo group: sharding moves host tensors)
    import ctypes
    skipped = ctypes.c_ulonglong(0)

    def clear_counters():
        _lib.reset_stats()
        _lib.check(_lib.lib().trx_skipped_rows(None, 1))          # clears the device counters
        _lib.check(_lib.lib().trx_pruned_rows(None, 1))

    elapsed, out, jobs, timing = batch_leg(ctx, args.tois, args.batch_n, args.n_time, args.steps, args.warmup,
                                           before_timed=clear_counters)

    def step(js, seed):
        np.random.seed(seed)
        torch.manual_seed(seed)
        return triceratops_amd.calc_probs_many(js)

    stats = dict(_lib.STATS)
    # rows whose light curve was not evaluated: lnL_EB_p's secondary-eclipse rule gives them +inf anyway
    _lib.check(_lib.lib().trx_skipped_rows(ctypes.byref(skipped), 1))
    stats["skipped_rows"] = int(skipped.value)
    # ... and the rows the bounded evaluation abandoned (settled from their constants or after ~16 probe cells: they are
    # NOT counted as evaluated)
    _lib.check(_lib.lib().trx_pruned_rows(ctypes.byref(skipped), 1))
    stats["abandoned_rows"] = int(skipped.value)
    stats["settled_cells"] = stats["cells"] - stats["skipped_rows"] * args.n_time
    stats["cells"] -= (stats["skipped_rows"] + stats["abandoned_rows"]) * args.n_time
    fpps = [float(tg.FPP) for tg in out]         # of the last timed step (the targets are updated in place)
    n_scen = sum(len(tg.lnZ) for tg in out)
    # Not timed: one more step with events around every likelihood launch and sampled parameter blocks
    # (the timed steps run a lnZ_* call as ONE library call, trx_scenario_evidence; tracing takes the
    # chain of torch operators around the same kernels, bit-identical results)
    _lib.TRACE = []
    step(jobs, 100)
    _sync(ctx)
    trace, _lib.TRACE = _lib.TRACE, None
    # cells evaluated over all ranks
    cells = torch.tensor([float(stats["cells"]), float(stats["rows"]), float(stats["skipped_rows"]),
                          float(stats["abandoned_rows"]), float(stats["settled_cells"])], dtype=torch.float64,
                         device="cpu" if ctx["debug_one"] or world == 1 else device)
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(cells)
    if rank != 0:
        return None
    nominal = float(args.batch_n) * args.n_time * n_scen * args.steps
    kern_s = sum(a.elapsed_time(b) for (_, _, _, _, a, b, _) in trace) * 1e-3
    cells_rank0 = float(sum(n * nt for (_, _, n, nt, _, _, _) in trace))
    # census on the traced samples (first launches of this rank)
    ev_cells, tot = 0.0, 0.0
    for (model, flags, n, nt, _, _, blk) in trace:
        if blk is None or blk.shape[1] == 0:
            continue
        c, _ = _lib.flux_grid(model, (flags & 3) | _lib.FLAG_COUNT_EVALUATIONS, jobs_time(jobs, device), blk.contiguous(),
                              0.00139, 20, False)
        ev_cells += float(c.sum())
        tot += float(c.numel())
    evals_per_cell = ev_cells / max(tot, 1.0)
    # the traced launches skip the excluded rows too: scale by the evaluated share of the timed steps
    evaluated_share = float(stats["settled_cells"]) / max(float(stats["settled_cells"]) + float(stats["skipped_rows"]) * args.n_time, 1.0)
    achieved = evals_per_cell * (F_ORBIT + F_MA) * cells_rank0 * evaluated_share / max(kern_s, 1e-12) / 1e12
    return {
        "metric": "light-curve-point x sample evals/sec", "value": float(cells[0]) / elapsed,
        "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32 model / f64 orbit+accumulators" if args.fp32_model else "f64",
        "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[3]: %d synthetic TOIs x 18 scenarios x N=%d draws, %d-point "
                               "light curves, calc_probs_many with device-side sampling, %d host thread(s) per rank, every lnZ_* call enqueued without a host sync on one of %d streams; "
                               "value counts the (draw, time) cells of the masked draws that are evaluated TO THE END: not the draws that lnL_EB_p's secondary-eclipse rule excludes (rows_not_evaluated_per_step), not the draws the bounded evaluation abandons (abandoned_rows_per_step: settled from their constants or after ~16 probe cells; with them: settled_cells_per_s)"
                               % (args.tois, args.batch_n, args.n_time, args.threads, __import__("triceratops_amd.sharding", fromlist=["streams"]).streams),
                   "tois": args.tois, "n_scenarios": n_scen, "N": args.batch_n, "n_time": args.n_time,
                   "evaluated_cells_per_step": float(cells[0]) / args.steps,
                   "evaluated_rows_per_step": (float(cells[1]) - float(cells[2]) - float(cells[3])) / args.steps,
                   "rows_not_evaluated_per_step": float(cells[2]) / args.steps,
                   "abandoned_rows_per_step": float(cells[3]) / args.steps,
                   "settled_cells_per_s": float(cells[4]) / elapsed,
                   "nominal_evals_per_s": nominal / elapsed,
                   "calc_probs_per_s": args.tois * args.steps / elapsed,
                   # host seconds per step and rank: unit lists, argument blocks + library calls of the rank's own units,
                   # stream wait, the collective, the tables of all targets, the rest; host_path_s = all but the waits
                   "per_rank": timing, "hw_queues": triceratops_amd_hw_queues(),
                   "parallelism": "whole TOIs, then whole stars, then single lnZ_* calls dealt to %d ranks by cost (LPT), "
                                  "one all_gather of the tables" % world},
        "roofline": {"bound": "fp64_valu", "achieved": achieved, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s",
                     "frac": achieved / FP64_VALU_PEAK_TF, "traffic": None,
                     "kernel_seconds_rank0": kern_s, "kernel_seconds_over_wall": kern_s / (elapsed / args.steps),
                     "model_evaluations_per_cell": evals_per_cell,
                     "note": "likelihood + log-mean-exp launches of rank 0 in ONE extra, untimed step run with events "
                             "around trx_lnz_scenario (summed over the host threads' streams: they overlap, so the sum "
                             "can exceed the wall-clock of a timed step); the rest of a step is the draw kernel, the "
                             "compaction and the result copies"},
        "cpu_baseline": None,
        "fpp_mean": float(np.mean(fpps)), "fpp_checksum": float(np.sum(fpps)),
    }


def jobs_time(jobs, device):
    from triceratops_amd import _lib
    return _lib.dev(jobs[0][1]["time"], device)


# ---------------------------------------------------------------------------------------------
def e2e_calc_probs(threads=3):
    """end-to-end calc_probs() wall-clock on TOI-465.01 (BASELINE configs[2]): the reference's example
    light curve (100 binned points) + contrast curve, target + 20 neighbours (75 scenarios), N = 1e6,
    parallel=True, in the three sampling modes; plus the 15-scenario run of the notebook's own star table."""
    import pandas as pd
    import torch
    import triceratops_amd
    from triceratops_amd.triceratops import target
    path = os.path.join(GOLD, "toi465_calc_probs.npz")
    if not os.path.exists(path):
        return None
    g = np.load(path)
    cols = ("ID", "Tmag", "Jmag", "Hmag", "Kmag", "ra", "dec", "mass", "rad", "Teff", "plx", "fluxratio", "tdepth")
    res = {"workload": "TOI-465.01: examples/TOI465_01_lightcurve.csv binned to 100 points + "
                       "TOI465_01_contrastcurve.csv, P_orb = 3.836169 d, N = 1e6 draws per scenario, parallel=True; "
                       "synthetic TRILEGAL table; reference notebook (unstated laptop): ~61 s per 15-scenario run",
           "seconds": {}, "FPP": {}}
    for tag, modes in (("blend", ("device", "device-threads", "numpy-device", "numpy")),
                       ("real", ("device", "numpy-device"))):
        st = pd.DataFrame({c: g["%s_stars_%s" % (tag, c)] for c in cols})
        st["ID"] = st["ID"].astype(np.int64)
        for mode in modes:
            triceratops_amd.set_sampling(mode.split("-threads")[0])
            triceratops_amd.set_threads(threads if mode.endswith("-threads") else 1)
            try:
                best = None
                for rep in range(2 if mode != "numpy" else 1):
                    tg = target(270380593, np.array([4]), stars=st.copy(),
                                trilegal_fname=os.path.join(GOLD, "trilegal_synth.csv"))
                    if rep == 0 and mode != "numpy":      # warm-up at small N: tables, allocator
                        np.random.seed(1)
                        tg.calc_probs(g["time"], g["flux"], float(g["sigma"][0]), float(g["P_orb"][0]),
                                      contrast_curve_file=os.path.join(GOLD, "toi465_cc.csv"), N=20000,
                                      parallel=True, verbose=0)
                    np.random.seed(465)
                    torch.manual_seed(465)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    tg.calc_probs(g["time"], g["flux"], float(g["sigma"][0]), float(g["P_orb"][0]),
                                  contrast_curve_file=os.path.join(GOLD, "toi465_cc.csv"), N=1_000_000,
                                  parallel=True, verbose=0)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                    best = dt if best is None else min(best, dt)
                key = "%s_%dscen_%s" % (tag, len(tg.lnZ), mode)
                res["seconds"][key] = best
                res["FPP"][key] = float(tg.FPP)
            finally:
                triceratops_amd.set_sampling("device")
                triceratops_amd.set_threads(1)
    res["threads"] = threads
    res["default_mode"] = "device (one host thread, calls enqueued on %d streams, one wait)" % __import__(
        "triceratops_amd.sharding", fromlist=["streams"]).streams
    return res


def usable_cores():
    """(cores this process may use, how that was found): the affinity mask capped by the cgroup CPU quota --
    omp_get_max_threads() counts the machine's cores, which a container does not own"""
    aff = len(os.sched_getaffinity(0))
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except Exception:
            continue
    n = aff if quota is None else max(1, min(aff, int(quota + 0.5)))
    return n, {"sched_getaffinity": aff, "cgroup_cpu_quota": quota, "os_cpu_count": os.cpu_count()}


def native_oracle(window_skip=False):
    """the oracle compiled on THIS box with -O3 -march=native; falls back to the shipped
    -O2 build when there is no compiler.  window_skip: the build with -DTRXO_WINDOW_SKIP (oracle/bench_window_skip.h,
    the GPU kernels' transit-window early-out on the CPU) -- only this function ever builds it, for the
    `window_early_out` legs; returns None when it cannot be built."""
    from oracle import oracle as O
    src = os.path.join(ROOT, "oracle", "trx_oracle.c")
    out = os.path.join(tempfile.mkdtemp(prefix="trx_oracle_", dir="/tmp"),
                       "libtrx_oracle_native%s.so" % ("_window" if window_skip else ""))
    cmd = ["gcc", "-O3", "-march=native", "-fPIC", "-std=c11", "-ffp-contract=off", "-fno-fast-math", "-fopenmp",
           "-shared", "-o", out, src, "-lm"] + (["-DTRXO_WINDOW_SKIP"] if window_skip else [])
    try:
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        O.use_library(out)
        return "gcc -O3 -march=native -fopenmp%s, built on this box" % (" -DTRXO_WINDOW_SKIP" if window_skip else "")
    except Exception:
        return None if window_skip else "shipped build (gcc -O2 -fopenmp): no compiler on this box"


def cpu_baseline(t, flux, rows_h, fams, budget_s):
    """The CPU oracle on a bounded sample of the same rows: all usable host cores (OpenMP over rows), one
    thread, the same two with the GPU kernels' transit-window early-out, and the reference-shaped numpy pipeline
    (materialised (n, n_time) grid + the elementwise passes and row sum of likelihoods.py:352-357, 427-438, 486,
    534-538) around the oracle's pytransit-shaped evaluate_pv."""
    from oracle import oracle as O
    from triceratops_amd import synth
    build = native_oracle()
    cores, how = usable_cores()
    omp_default = O.num_threads()

    def leg(run, n_threads, what):
        O.set_num_threads(n_threads)
        per = 4
        t0 = time.perf_counter()
        for fam, rows in zip(fams[:3], rows_h[:3]):
            run(fam, rows[:, :per * n_threads])
        probe = time.perf_counter() - t0
        rate = 3 * per * n_threads * t.size / probe
        per_fam = int(max(per * n_threads, min(rows_h[0].shape[1], budget_s * rate / t.size / len(fams))))
        t0 = time.perf_counter()
        evals = 0
        for fam, rows in zip(fams, rows_h):
            run(fam, rows[:, :per_fam])
            evals += per_fam * t.size
        dt = time.perf_counter() - t0
        return {"value": evals / dt, "unit": "evals/s", "cores": n_threads, "kind": "port",
                "sample": "first %d rows of each of the 18 families x %d points (%.1f s, %s)"
                          % (per_fam, t.size, dt, what)}

    def fused(fam, rows):
        O.lnl_batch(fam[1], t, flux, synth.SIGMA, rows, companion_is_host=fam[2])

    def numpy_grid(fam, rows):
        # the reference's dataflow: unit conversion in numpy, evaluate_pv -> (n, n_time) fp64 grid,
        # dilution passes over the grid, 0.5 * sum((flux - model)**2 / sigma**2, axis=1)
        name, model, is_host, _ = fam
        Rsun, Rearth = 6.957e10, 6.3781e8
        if model == 0:
            R_p, P, inc, a, R_s, u1, u2, ecc, argp, cfr = rows
            k = R_p * Rearth / (R_s * Rsun)
            F_eb = None
        else:
            R_EB, ebfr, P, inc, a, R_s, u1, u2, ecc, argp, cfr = rows
            k = R_EB / R_s
            k = np.where((k - 1.0) < 1e-6, k * 0.999, k)
            F_eb = (ebfr / (1 - ebfr)).reshape(-1, 1)
        F_comp = (cfr / (1 - cfr)).reshape(-1, 1)
        pvp = np.array([k, np.zeros_like(k), P, a / (R_s * Rsun), inc * np.pi / 180, ecc,
                        (90 - argp) * np.pi / 180]).T
        ldc = np.array([u1, u2]).T
        grid = O.evaluate_pv(t, pvp, ldc, synth.EXPTIME, synth.NSAMPLES)
        if F_eb is None:
            F_d = 1 / F_comp if is_host else F_comp
            grid = (grid + F_d) / (1 + F_d)
        else:
            pvs = pvp.copy()
            ks = R_s / rows[0]
            pvs[:, 0] = np.where((ks - 1.0) < 1e-6, ks * 0.999, ks)
            pvs[:, 6] = (90 - argp + 180) * np.pi / 180
            sec = np.min(O.evaluate_pv(np.linspace(-0.05, 0.05, 25), pvs, ldc, 0.0, 1), axis=1, keepdims=True)
            if is_host:
                x, y, F_d = F_eb / F_comp, F_comp / F_eb, 1 / (F_comp + F_eb)
            else:
                x, y, F_d = F_eb, 1 / F_eb, F_comp / (1 + F_eb)
            grid = (grid + x) / (1 + x)
            sec = (sec + y) / (1 + y)
            grid = (grid + F_d) / (1 + F_d)
            secdepth = 1 - (sec + F_d) / (1 + F_d)
        h = 0.5 * np.sum((flux - grid) ** 2 / synth.SIGMA ** 2, axis=1)
        if model == 1:
            h[(secdepth >= 1.5 * synth.SIGMA).ravel()] = np.inf
        return h

    try:
        allc = leg(fused, cores, "fused C restatement, OpenMP over rows on all usable cores")
        one = leg(fused, 1, "fused C restatement, one thread")
        grid = leg(numpy_grid, cores, "numpy materialised-grid pipeline of the reference around the oracle's "
                                      "evaluate_pv, model on all usable cores, numpy passes on one")
        win = win1 = None
        if native_oracle(window_skip=True) is not None:       # a second library: the checker's build has no such code
            O.set_window_skip(1)
            try:
                win = leg(fused, cores, "fused C restatement + the GPU kernels' transit-window early-out "
                                        "(oracle/bench_window_skip.h, -DTRXO_WINDOW_SKIP), all usable cores")
                win1 = leg(fused, 1, "fused C restatement + transit-window early-out, one thread")
            finally:
                O.set_window_skip(0)
    finally:
        O.set_num_threads(omp_default)
    allc["single_thread"] = one
    allc["window_early_out"] = win
    allc["window_early_out_single_thread"] = win1
    allc["numpy_grid"] = grid
    allc["build"] = build
    allc["cores_detail"] = dict(how, omp_get_max_threads=omp_default, used=cores,
                                scaling_all_over_one=allc["value"] / one["value"])
    allc["note"] = ("a port (the reference's pytransit engine cannot be installed): every sub-exposure of every point "
                    "with a full Kepler solve, like the reference; window_early_out adds the one shortcut of the GPU "
                    "kernels that a CPU implementation would take as well.  cores = affinity mask capped by the cgroup "
                    "quota; round 2 ran omp_get_max_threads() = every core of the machine inside a smaller quota, "
                    "hence its 13x 'scaling' on 128 threads")
    return allc


if __name__ == "__main__":
    main()
