"""Where round 4's exit-rule bug (trx_set_debug_bug(1)) bites: values of N at which a scenario's masked count and its
count behind the pilot fall on different sides of a step of the rows-per-wave rule.  Call by call the rule asks for 3200
waves a launch, in a launch chain for 3200 / branches a branch, so the steps sit at other counts.
    python profiles/r05/scan_debug_bug.py [chain: 0 | 1]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import triceratops_amd  # noqa: E402,F401
from triceratops_amd import _lib  # noqa: E402
import anchors  # noqa: E402

chain = int(sys.argv[1]) if len(sys.argv) > 1 else 0
L = _lib.lib()
L.trx_set_star_chain(chain)
grid = range(150_000, 420_000, 10_000) if not chain else list(range(4_000, 40_000, 2_000)) + list(range(40_000, 100_000, 10_000))
for case in ("toi411", "toi465_nocc"):
    for N in grid:
        out = []
        for bug in (0, 1):
            L.trx_set_debug_bug(bug)
            try:
                lnZ, prob, fpp, rp = anchors.run(case, 7, N=N, sampling="device")
                out.append("ok FPP %.6f" % fpp)
            except _lib.TrxError:
                out.append("TrxError")
        print("chain", chain, case, N, out, flush=True)
L.trx_set_debug_bug(0)
