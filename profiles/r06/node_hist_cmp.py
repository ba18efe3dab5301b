import sys
import numpy as np
a, b = np.load(sys.argv[1]).astype(int), np.load(sys.argv[2]).astype(int)
fa, fb = np.load(sys.argv[1] + ".flux.npy"), np.load(sys.argv[2] + ".flux.npy")
print("max |flux difference| %.3g" % np.abs(fa - fb).max())
h = a >= 20
print("cells with >= 20 evaluations before: %.1f per row; after, by count:" % (h.sum() / a.shape[0]))
u, c = np.unique(b[h], return_counts=True)
print(" ".join("%d:%.2f" % (x, y / a.shape[0]) for x, y in zip(u, c)))
print("evaluations in those cells: %.1f -> %.1f per row; elsewhere %.1f -> %.1f" % (a[h].sum() / a.shape[0], b[h].sum() / a.shape[0], a[~h].sum() / a.shape[0], b[~h].sum() / a.shape[0]))
