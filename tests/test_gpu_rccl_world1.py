"""The RCCL branch of the sharded path, executed on the hardware there is: a process group
of ONE rank on the "nccl" backend (= RCCL on ROCm) and sharding.collective_at_world_one, so that calc_probs_many takes
the multi-rank path -- schedule, per-unit seeds, the record table as a DEVICE tensor behind its header row, padded,
through all_gather_into_tensor -- on a single MI355X.  Until round 6 that branch (sharding.py: `dev = "cuda" if
backend == "nccl"`) had never run anywhere: gloo moved host tensors in every test.  No 8-GPU curve is claimed here."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import triceratops_amd
import torch, torch.distributed as dist
from triceratops_amd import sharding, synth
from triceratops_amd.triceratops import calc_probs_many
from test_sharding import _two_jobs, _tables
torch.cuda.set_device(0)
store = "/tmp/trx_test_pg_%%d" %% os.getpid()
dist.init_process_group("nccl", init_method="file://" + store, rank=0, world_size=1, device_id=torch.device("cuda", 0))
out = {"backend": dist.get_backend(), "world": dist.get_world_size()}
for mode in ("device", "numpy-device"):
    triceratops_amd.set_sampling(mode)
    def run(collective):
        jobs = _two_jobs()
        np.random.seed(5); torch.manual_seed(5)
        sharding.collective_at_world_one = collective
        sharding.per_unit_seed = not collective        # (the multi-rank path seeds per unit: same numbers either way)
        try:
            calc_probs_many(jobs)
        finally:
            sharding.collective_at_world_one = False
            sharding.per_unit_seed = False
        return _tables(jobs), float(sharding.timing["gather_s"]), list(sharding.last_seed_bases or [])
    plain, _, _ = run(False)
    sharding.timing["gather_s"] = 0.0
    coll, gather_s, bases = run(True)
    out[mode] = {"equal": all(np.array_equal(a, b, equal_nan=True) for a, b in zip(plain, coll)),
                 "gather_s": gather_s, "bases": bases, "finite": bool(np.isfinite(coll[0][0]))}
# the config-3 shape at a reduced size: 8 synthetic TOIs through the collective, tables equal to the plain run
triceratops_amd.set_sampling("device")
gold = os.path.join(%(root)r, "tests", "golden")
def batch(collective):
    jobs = synth.toi_jobs(8, n_time=200, N=50000, seed=synth.SEED, trilegal_fname=os.path.join(gold, "trilegal_synth.csv"),
                          contrast_curve_file=os.path.join(gold, "contrast_curve_synth.csv"))
    torch.manual_seed(9)
    sharding.collective_at_world_one = collective
    sharding.per_unit_seed = not collective
    try:
        tg = calc_probs_many(jobs)
    finally:
        sharding.collective_at_world_one = False
        sharding.per_unit_seed = False
    return [float(t.FPP) for t in tg], [np.array(t.lnZ) for t in tg]
fa, la = batch(False)
fb, lb = batch(True)
out["batch"] = {"equal": fa == fb and all(np.array_equal(a, b, equal_nan=True) for a, b in zip(la, lb)),
                "gather_s": float(sharding.timing["gather_s"]), "share": sharding.last_share}
dist.barrier()
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
"""


def test_calc_probs_many_through_a_one_rank_rccl_group():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run([sys.executable, "-c", _CHILD % {"root": ROOT}], capture_output=True, text=True, timeout=600,
                       cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    import json
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    d = json.loads(line[len("RESULT "):])
    print(d)
    assert d["backend"] == "nccl" and d["world"] == 1
    for mode in ("device", "numpy-device"):
        assert d[mode]["equal"] and d[mode]["finite"], d[mode]
        assert d[mode]["gather_s"] > 0.0 and len(d[mode]["bases"]) == 1       # the collective ran and carried the header row
    assert d["batch"]["equal"] and d["batch"]["gather_s"] > 0.0
    assert d["batch"]["share"]["jobs"] == [8] and d["batch"]["share"]["calls"] == [8 * 12]


def test_bench_under_an_external_launcher_with_one_rank():
    """`torchrun --nproc-per-node 1 bench.py --gpus 1`: WORLD_SIZE = 1 in the environment; the batch object reports the
    gather of the one-rank RCCL group"""
    import json
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--n-samples", "4000",
                        "--no-cpu-baseline", "--no-extras", "--pmc", "off", "--tois", "4", "--batch-n", "50000"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    r = d["batch"]["per_rank"]["rccl_world1"]
    assert r["backend"] == "nccl" and r["gather_s"] > 0.0 and r["tables_equal"] is True
    assert d["batch"]["per_rank"]["gather_s"][0] > 0.0
