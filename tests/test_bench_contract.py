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
SMALL = ["--n-samples", "2048", "--n-time", "256", "--steps", "2", "--warmup", "1"]


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


@pytest.mark.gpu
def test_single_gpu_line():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + ["--cpu-seconds", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _last_json(p.stdout)
    _check_common(d, 1)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "evals/s" and c["sample"]
    assert d["value"] > 50 * c["value"]


@pytest.mark.gpu
def test_two_ranks_control_flow_on_one_device():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--debug-single-device"] + SMALL,
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _last_json(p.stdout)
    _check_common(d, 2)
    assert d["cpu_baseline"] is None           # timed on rank 0 at N = 1 only
