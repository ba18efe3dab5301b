#!/usr/bin/env python3
"""bench.py -- light-curve-point x sample evaluations per second of the marginal-likelihood hot path.

One *step* = one pass of the hot path over BASELINE.json configs[1]: a synthetic 2000-point
light curve, all 18 scenario families (TP/EB/EBx2P x T,P,S,D,B,N), N_samples rows each, fp64:
for every family the fused likelihood kernel (trx_lnl_batch) then the log-mean-exp evidence
(trx_lnz_from_halfchi2).  Inputs are resident in HBM before the timed region.  With N > 1 ranks
(one process per GPU, RCCL) every rank runs the same per-GPU work on its own rows (weak scaling:
scenarios x TOIs shard with no data-path collective) and the step ends with ONE all_gather of the
per-scenario lnZ vector.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     dominant kernel (the fused likelihood kernel): it is fp64-VALU bound, not HBM/MFMA
               (SURVEY.md 8d); achieved = algorithmic fp64 flop / mean launch duration measured
               with events on the launch stream.  Its HBM-side numbers and the two HBM-bound
               reduction kernels are reported under "kernels".
  cpu_baseline the CPU oracle (a port: the reference's pytransit engine is not installable) on
               a bounded sample of the same rows, all host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_TIME = 2000
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK_TF = 78.6       # MI355X fp64 vector peak = 1/2 of the 157.3 TF fp32 vector peak
# algorithmic fp64 operations per sub-exposure of the PLAIN restatement (oracle/trx_oracle.c),
# counted one per add/sub/mul/div/sqrt/compare and one per libm call:
F_ORBIT = 100.0                # offset+mean anomaly 8, Kepler (guess 8 + 3 Halley iterations x 22) 74, position 15
F_MA = 200.0                   # case analysis 40, two cel integrals (4 iterations x 14 + 10) x 2, combination 19


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-samples", type=int, default=100_000, help="rows per scenario family")
    ap.add_argument("--n-time", type=int, default=N_TIME)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    # test hook for 1-GPU boxes: run the N>1 control flow (rendezvous, barrier, gather, max-reduce)
    # with every rank on cuda:0 and a gloo process group (RCCL refuses two ranks on one device)
    ap.add_argument("--debug-single-device", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--all-subexposures", action="store_true",
                    help="switch the reduced-node exposure average off: every cell evaluates all nsamples "
                         "sub-exposures (the plain algorithm's instruction stream; profiles/r01_*_all_sub*.json)")
    ap.add_argument("--fp32-model", action="store_true",
                    help="BASELINE config 5: fp32 Mandel-Agol arithmetic, fp64 orbit/chi^2/log-mean-exp "
                         "(flux within 2e-6, chi^2/2 within 2e-4 relative of the fp64 path)")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from triceratops_amd import _lib, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    debug_one = args.debug_single_device
    if world > 1 and not debug_one:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    elif world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(0)
        dist.init_process_group("gloo")
    else:
        torch.cuda.set_device(0)
    _lib.require_gpu()
    if args.all_subexposures:
        _lib.lib().trx_set_supersample_tiers(0)
    device = torch.device("cuda", local_rank if (world > 1 and not debug_one) else 0)

    def gather(dst, src):
        if debug_one:                      # gloo moves host tensors
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

    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
           for _ in fams] for _ in range(args.steps)]

    def step(events=None):
        lnz = []
        for i, (name, model, is_host, has_comp) in enumerate(fams):
            flags = (_lib.FLAG_COMPANION_IS_HOST if is_host else 0) | (_lib.FLAG_FP32_MODEL if args.fp32_model else 0)
            if events is not None:
                events[i][0].record()
            _lib.lnl_batch(model, flags, t_d, f_d, synth.SIGMA, rows_d[i], synth.EXPTIME,
                           synth.NSAMPLES, out=h_d[i])
            if events is not None:
                events[i][1].record()
            lnz.append(_lib.lnz_from_halfchi2(h_d[i], lnprior_d[i], n_total, lnsigma))
        mine = torch.cat(lnz)
        if world > 1:
            gather(lnz_all, mine)          # the single data-path collective (RCCL over xGMI)
        else:
            lnz_all.copy_(mine)
        return lnz_all

    def sync():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(ev[s])
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if debug_one else device)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te[0])

    lnz_host = lnz_all.cpu().numpy()
    evals_per_step_per_gpu = float(n_time) * n_rows * len(fams)
    value = evals_per_step_per_gpu * world * args.steps / elapsed

    # ---- per-kernel numbers (rank 0, N=1 only adds the extra diagnostics) -------------------
    kern_ms = np.array([[a.elapsed_time(b) for (a, b) in ev[s]] for s in range(args.steps)])
    mean_launch_s = float(kern_ms.mean()) * 1e-3
    out = None
    if rank == 0:
        # fraction of sub-exposures inside the occultation region, from the model grid of a sample
        p_in = []
        for i, fam in enumerate(fams[:3]):
            g, _ = _lib.flux_grid(fam[1], 0, t_d, rows_d[i][:, :512].contiguous(), synth.EXPTIME,
                                  synth.NSAMPLES, False)
            p_in.append(float((g < 1.0).double().mean()))
        p_in = float(np.mean(p_in))
        # model evaluations the kernel actually plans per cell (census knob), same sample
        L_ = _lib.lib()
        L_.trx_set_debug_node_counts(1)
        try:
            evals_per_cell = float(np.mean([
                float(_lib.flux_grid(fam[1], 0, t_d, rows_d[i][:, :512].contiguous(), synth.EXPTIME,
                                     synth.NSAMPLES, False)[0].mean()) for i, fam in enumerate(fams[:3])]))
        finally:
            L_.trx_set_debug_node_counts(0)
        flop_per_eval = synth.NSAMPLES * (F_ORBIT + p_in * F_MA)
        evals_per_launch = float(n_time) * n_rows
        achieved_tf = flop_per_eval * evals_per_launch / mean_launch_s / 1e12
        n_par = np.mean([r.shape[0] for r in rows_h])
        alg_bytes_per_launch = (8.0 * n_par + 8.0) * n_rows + 16.0 * n_time
        kernels = {"rows_kernel<lnl>": {
            "bound": "fp64_valu", "mean_launch_ms": mean_launch_s * 1e3,
            "all_subexposures": bool(args.all_subexposures),
            "evals_per_launch": evals_per_launch, "p_in": p_in, "flop_per_eval": flop_per_eval,
            "model_evaluations_per_cell": evals_per_cell,
            "algorithmic_bytes_per_launch": alg_bytes_per_launch,
            "hbm_GBps": alg_bytes_per_launch / mean_launch_s / 1e9}}

        # HBM-bound reductions at a size past the caches: chi^2 over a materialised grid, LME
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

        n_grid = 200_000
        grid = torch.rand((n_grid, n_time), dtype=torch.float64, device=device)
        dt = timed(lambda: _lib.chi2_grid(f_d, grid, synth.SIGMA))
        gb = (grid.numel() * 8 + n_grid * 8) / 1e9
        kernels["chi2_grid_kernel"] = {"bound": "hbm", "bytes": gb * 1e9, "ms": dt * 1e3,
                                       "GBps": gb / dt, "frac": gb / dt / HBM_PEAK_GBS}
        del grid
        big = torch.empty(400_000_000, dtype=torch.float64, device=device).uniform_(-3000.0, -1.0)
        dt = timed(lambda: _lib.log_mean_exp(big, big.numel()))
        gb = big.numel() * 8 / 1e9
        kernels["lme_partial_kernel"] = {"bound": "hbm", "bytes": gb * 1e9, "ms": dt * 1e3,
                                         "GBps": gb / dt, "frac": gb / dt / HBM_PEAK_GBS}
        del big

        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(t, flux, rows_h, fams, args.cpu_seconds)

        out = {
            "metric": "light-curve-point x sample evals/sec", "value": value,
            "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32 model / f64 orbit+accumulators" if args.fp32_model else "f64", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: synthetic %d-point light curve, 18 "
                                   "scenario families x %d transiting rows, nsamples=%d supersampling, "
                                   "fused lnL + log-mean-exp" % (n_time, n_rows, synth.NSAMPLES),
                       "n_time": n_time, "n_samples": n_rows, "n_scenarios": len(fams),
                       "evals_per_step_per_gpu": evals_per_step_per_gpu,
                       "parallelism": "scenario-sharded x%d, one all_gather of lnZ" % world},
            "roofline": {"bound": "fp64_valu", "achieved": achieved_tf, "peak": FP64_VALU_PEAK_TF,
                         "frac_all_subexposures": all_sub_frac(n_time, n_rows, args),
                         "frac_executed": (evals_per_cell * (F_ORBIT + F_MA) * evals_per_launch / mean_launch_s
                                           / 1e12 / FP64_VALU_PEAK_TF),
                         "unit": "TFLOP/s", "frac": achieved_tf / FP64_VALU_PEAK_TF,
                         "traffic": pmc_traffic(n_time, n_rows),
                         "note": "dominant kernel rows_kernel<lnl> is fp64-VALU bound (no MFMA shape, "
                                 "~0.05 B/eval of HBM traffic); HBM-bound reductions under 'kernels'. "
                                 "achieved = algorithmic flops of the plain S-sub-exposure algorithm / "
                                 "launch time; the kernel reaches the same averages (to 1e-13) from a few "
                                 "Gauss nodes where the exposure is far from the limb contacts, so "
                                 "frac_all_subexposures (shortcut off, every sub-exposure evaluated) is "
                                 "the figure for the instruction stream itself and frac_executed counts "
                                 "only the model evaluations the kernel really runs (x 300 plain flops each)"},
            "kernels": kernels,
            "cpu_baseline": cpu,
            "lnZ_checksum": float(np.nansum(lnz_host[np.isfinite(lnz_host)])),
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def all_sub_frac(n_time, n_rows, args):
    """roofline fraction of the same workload with every sub-exposure evaluated (this run when
    --all-subexposures is given, else the committed run of that mode, profiles/all_subexposures.json)"""
    if args.fp32_model:
        return None
    path = os.path.join(ROOT, "profiles", "all_subexposures.json")
    if args.all_subexposures or not os.path.exists(path):
        return None
    rec = json.load(open(path))
    if rec.get("n_time") == n_time and rec.get("n_samples") == n_rows:
        return rec["frac"]
    return None


def pmc_traffic(n_time, n_rows):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of this
    same command; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 with the gfx950 factor 2 on the read
    side, MI355X_MICROARCH.md section HBM).  None when no profile matches this workload."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    for rec in json.load(open(path)):
        if rec["n_time"] == n_time and rec["n_samples"] == n_rows:
            return (2.0 * rec["fetch_size_kb"] + rec["write_size_kb"]) * 1024.0
    return None


def cpu_baseline(t, flux, rows_h, fams, budget_s):
    """The CPU oracle on a bounded sample of the same rows, all host cores (OpenMP)."""
    from oracle import oracle as O
    from triceratops_amd import synth
    cores = O.num_threads()
    per = 8
    t0 = time.perf_counter()
    for fam, rows in zip(fams[:3], rows_h[:3]):
        O.lnl_batch(fam[1], t, flux, synth.SIGMA, rows[:, :per * cores], companion_is_host=fam[2])
    probe = time.perf_counter() - t0
    rate = 3 * per * cores * t.size / probe
    per_fam = int(max(per * cores, min(rows_h[0].shape[1], budget_s * rate / t.size / len(fams))))
    t0 = time.perf_counter()
    evals = 0
    for fam, rows in zip(fams, rows_h):
        O.lnl_batch(fam[1], t, flux, synth.SIGMA, rows[:, :per_fam], companion_is_host=fam[2])
        evals += per_fam * t.size
    dt = time.perf_counter() - t0
    return {"value": evals / dt, "unit": "evals/s", "cores": cores, "kind": "port",
            "sample": "first %d rows of each of the 18 families x %d points (%.1f s, OpenMP over rows)"
                      % (per_fam, t.size, dt)}


if __name__ == "__main__":
    main()
