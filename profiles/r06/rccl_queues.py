"""Does a live RCCL communicator slow the 64-target step, and do more hardware queues help?
    GPU_MAX_HW_QUEUES=<n> python profiles/r06/rccl_queues.py <0|1: one-rank nccl group first> [steps]"""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import triceratops_amd  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from triceratops_amd import sharding, synth  # noqa: E402

rccl = int(sys.argv[1]) if len(sys.argv) > 1 else 0
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
torch.cuda.set_device(0)
if rccl:
    store = "/tmp/trx_q_pg_%d" % os.getpid()
    dist.init_process_group("nccl", init_method="file://" + store, rank=0, world_size=1, device_id=torch.device("cuda", 0))
    x = torch.ones(8, device="cuda")
    y = torch.empty(8, device="cuda")
    dist.all_gather_into_tensor(y, x)
    torch.cuda.synchronize()
GOLD = os.path.join(ROOT, "tests", "golden")
jobs = synth.toi_jobs(64, n_time=200, N=1_000_000, seed=synth.SEED, trilegal_fname=os.path.join(GOLD, "trilegal_synth.csv"),
                      contrast_curve_file=os.path.join(GOLD, "contrast_curve_synth.csv"))
triceratops_amd.set_sampling("device")
sharding.freeze_gc = True
out = []
for s in range(steps + 2):
    torch.manual_seed(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    triceratops_amd.calc_probs_many(jobs)
    torch.cuda.synchronize()
    out.append(time.perf_counter() - t0)
print("rccl group %d, GPU_MAX_HW_QUEUES %s (%s): steps %s" % (rccl, os.environ.get("GPU_MAX_HW_QUEUES"), triceratops_amd.hw_queues(),
                                                            " ".join("%.4f" % v for v in out[2:])))
if rccl:
    dist.destroy_process_group()
