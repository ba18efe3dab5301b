"""What the 80 us of a small upload (_lib.dev) are made of, on an idle GPU."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from triceratops_amd import _lib
_lib.require_gpu()
d = torch.device("cuda", 0)
a = np.random.rand(200)
n = 2000
def bench(label, f):
    for _ in range(50): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    dt = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    print("%-46s %6.1f us" % (label, dt * 1e6))
up = _lib.upload_stream(d).stream
bench("_lib.dev(a)", lambda: _lib.dev(a, d))
bench("torch.empty pinned", lambda: torch.empty(a.shape, dtype=torch.float64, pin_memory=True))
st = torch.empty(a.shape, dtype=torch.float64, pin_memory=True)
bench("staged.numpy()[...] = a", lambda: st.numpy().__setitem__(Ellipsis, a))
def ctx():
    with torch.cuda.stream(up):
        pass
bench("with torch.cuda.stream(up): pass", ctx)
def to_():
    with torch.cuda.stream(up):
        return st.to(d, non_blocking=True)
bench("... staged.to(d, non_blocking)", to_)
bench("torch.empty(200, device)", lambda: torch.empty(200, dtype=torch.float64, device=d))
def ev():
    e = torch.cuda.Event(); e.record(up); return e
bench("Event() + record", ev)
bench("torch.cuda.current_stream(d)", lambda: torch.cuda.current_stream(d))
t = st.to(d)
cur = torch.cuda.current_stream(d)
bench("t.record_stream(cur)", lambda: t.record_stream(cur))
e = ev()
bench("cur.wait_event(e)", lambda: cur.wait_event(e))
bench("np.ascontiguousarray", lambda: np.ascontiguousarray(a, dtype=np.float64))
