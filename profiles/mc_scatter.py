"""numpy against device sampling, all ten lnZ_* x 20 seeds at N = 1e6 (the table tests/test_gpu_equivalence.py
asserts on):  python profiles/mc_scatter.py > profiles/r03/mc_scatter.txt"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import test_gpu_equivalence as T
rows, text = T.table(T.collect())
print(text)
