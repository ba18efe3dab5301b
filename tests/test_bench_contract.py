"""bench.py's output contract on the GPU box: one JSON line with the driver's keys plus the
`roofline` and `cpu_baseline` objects, and the N > 1 control flow (rendezvous on 127.0.0.1,
barrier, the single all_gather, max-over-ranks timing) exercised with two ranks sharing cuda:0
through the gloo debug hook (RCCL refuses two ranks on one device)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--n-samples", "2048", "--n-time", "256", "--steps", "2", "--warmup", "1", "--no-e2e",
         "--tois", "4", "--batch-n", "50000"]          # (the `batch` object of the line at a reduced size)


def _last_json(out):
    lines = [ln for ln in out.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def _check_common(d, n_gpus):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == n_gpus and d["steps"] == 2 and d["warmup"] == 1
    assert d["unit"] == "evals/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    evals = 256 * 2048 * 18 * n_gpus * 2
    assert abs(d["value"] - evals / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-9
    r = d["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(r)
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1
    # the configs[3] strong-scaling step rides in the same line, with the host timings of every rank
    b = d["batch"]
    assert b["n_gpus"] == n_gpus and b["scaling"] == "strong" and b["ms_per_step"] > 0
    for k in ("prepare_s", "enqueue_s", "wait_s", "gather_s", "finish_s", "other_s", "host_path_s"):
        assert len(b["per_rank"][k]) == n_gpus, k


def test_gpus_flag_must_agree_with_the_launcher():
    """--gpus is checked against WORLD_SIZE before anything is imported (runs without a GPU)"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True,
                       text=True, timeout=120, cwd=ROOT, env=dict(os.environ, WORLD_SIZE="3", RANK="0"))
    assert p.returncode != 0 and "WORLD_SIZE=3" in p.stderr


@pytest.mark.gpu
def test_single_gpu_line():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + ["--cpu-seconds", "1"],
                       capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _last_json(p.stdout)
    _check_common(d, 1)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "evals/s" and c["sample"]
    assert c["single_thread"]["cores"] == 1 and c["single_thread"]["value"] > 0
    assert c["numpy_grid"]["value"] > 0
    # the honest CPU leg: usable cores (affinity mask capped by the cgroup quota), built on this box, and the
    # same restatement with the GPU kernels' transit-window early-out
    assert c["cores"] == c["cores_detail"]["used"] <= c["cores_detail"]["sched_getaffinity"]
    assert "march=native" in c["build"] or "shipped" in c["build"]
    # (no ratio of two timings is asserted on a 1 s CPU sample at this toy size: the numbers are reported)
    assert c["window_early_out"]["value"] > 0 and c["window_early_out_single_thread"]["cores"] == 1
    assert d["value"] > 0
    # the reference's real operating points ride in the same line, each with its own census and fraction
    assert "representative" in d["config"]
    for key, n_time in (("n100", 100), ("n200", 200), ("n2000_irregular", 2000)):
        sh = d["shapes"][key]
        assert sh["n_time"] == n_time and sh["evals_per_s"] > 0 and 0 < sh["frac"] < 1
        assert 1.0 < sh["model_evaluations_per_cell"] < 20.0
    assert d["shapes"]["n2000_irregular"]["uniform_grid"] is False
    r = d["roofline"]
    # executed work is priced below the plain 20-sub-exposure equivalent; the run with the shortcut off is timed
    # in the same process (reported: at this toy size both launches sit near the launch latency)
    assert r["achieved"] < r["plain_algorithm_equivalent_tflops"]
    assert r["all_subexposures"]["mean_launch_ms"] > 0 and r["mean_launch_ms"] > 0
    assert 1.0 < r["model_evaluations_per_cell"] < 20.0
    # traffic comes from PMC child runs of this very command (or is null when rocprofv3 is missing)
    if r["traffic"] is not None:
        assert "in this run" in r["traffic_detail"]["source"]
        assert r["traffic"] > 0.5 * r["algorithmic_bytes_per_launch"]
    for k in ("chi2_grid_kernel", "lme_partial_kernel"):
        assert 0 < d["kernels"][k]["frac"] < 1


@pytest.mark.gpu
def test_two_ranks_control_flow_on_one_device():
    """`python bench.py --gpus 2` with no external launcher: bench.py starts the two ranks itself"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--debug-single-device"]
                       + SMALL, capture_output=True, text=True, timeout=420, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _last_json(p.stdout)
    _check_common(d, 2)
    assert d["cpu_baseline"] is None           # timed on rank 0 at N = 1 only


@pytest.mark.gpu
def test_two_ranks_under_an_external_launcher():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--debug-single-device"] + SMALL,
                       capture_output=True, text=True, timeout=420, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    _check_common(_last_json(p.stdout), 2)


@pytest.mark.gpu
def test_batch_mode_two_ranks():
    """BASELINE configs[3] mode at a reduced size: 4 TOIs x 18 scenarios, dealt to two ranks"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--debug-single-device",
                        "--mode", "batch", "--tois", "4", "--batch-n", "50000", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=420, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _last_json(p.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["n_scenarios"] == 72
    assert d["value"] > 0 and d["roofline"]["frac"] > 0


def _batch_line(gpus, extra=()):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--mode", "batch",
                        "--tois", "64", "--batch-n", "20000", "--steps", "5", "--warmup", "1", *extra],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    return _last_json(p.stdout)


@pytest.mark.gpu
def test_host_path_of_a_rank_shrinks_with_the_world_size():
    """BASELINE configs[3] must strong-scale by construction: what a rank's HOST does per step -- listing the units,
    building the argument blocks of ITS OWN units and handing them to the library, filling the tables -- has to
    shrink with the number of ranks, or eight GPUs wait for eight Pythons (round 3: every rank prepared and
    finished all 64 targets, 0.042 s per step whatever the world size; round 4 dealt single calls, so every rank
    touched 16 of the 64 targets: a quarter).  Eight gloo ranks on this box's one GPU against one rank, at N = 2e4
    so that the one GPU the eight processes share never pushes back on their queues.

    The gate is STRUCTURAL -- what the schedule hands a rank (whole targets, an eighth of them, an eighth of the
    calls) -- plus one RATIO of host times measured in this very test on this very box (the best of five steps on
    either side; round 4 asserted an absolute 0.035 s chosen on another machine, and a slower host failed it).
    Measured on the builder's leases: ratio 0.144 (0.18 before a rank left the tables of the targets it did not evaluate
    to their first reader, 0.24 before the records were turned into rows while the GPU works, 0.28
    then with eight busy-loop processes beside the suite: profiles/r05/suite_on_a_busy_box.txt) -- listing the units of
    all 64 targets and filling all 64 tables is done by every rank, so it is not 1/8 --; the bar is twice the measured
    ratio and more.  The seconds are printed, not judged."""
    one = _batch_line(1)
    eight = _batch_line(8, ("--debug-single-device",))
    pr1, pr8 = one["config"]["per_rank"], eight["config"]["per_rank"]
    assert eight["n_gpus"] == 8 and len(pr8["host_path_s"]) == 8
    assert pr1["calls"] == [64 * 12] and pr1["jobs"] == [64]
    assert pr8["calls"] == [96] * 8 and pr8["jobs"] == [8] * 8 and pr8["stars"] == [16] * 8
    h1 = pr1["host_path_best_s"][0]
    h8 = max(pr8["host_path_best_s"])
    e8 = pr8["enqueue_s"]
    print("\nhost path per step (best of 5): 1 rank %.4f s, 8 ranks (max) %.4f s, ratio %.3f; mean host path %.4f / %.4f; "
          "enqueue of the 8 ranks %s (max/min %.2f)" % (h1, h8, h8 / h1, pr1["host_path_s"][0], max(pr8["host_path_s"]),
                                                       ["%.4f" % v for v in e8], max(e8) / max(min(e8), 1e-9)))
    assert h8 <= 0.45 * h1, (h1, h8)
