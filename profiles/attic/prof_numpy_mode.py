import sys, os, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
sys.argv = [sys.argv[0], "numpy", "real", "1"]
import runpy
pr = cProfile.Profile()
pr.enable()
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "e2e_toi465.py"), run_name="__main__")
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
