import numpy as np, sys
a = np.loadtxt(sys.argv[1]).reshape(32, 32, 4, 8)
ok = (a[..., 0] > 0).all(axis=(1, 2)); a = a[ok]
two = a[..., 3] > a[..., 0]   # second pass happened in this chunk (stamp fresh)
d1 = a[..., 1] - a[..., 0]; d2 = a[..., 2] - a[..., 1]
print("pass0: wait+phase1 %.0f  phase2 %.0f" % (d1.mean(), d2.mean()))
m = (a[..., 3] > a[..., 2]) & (a[..., 4] > a[..., 3])
print("pass1: (%.0f%% of cells) wait+phase1 %.0f  phase2 %.0f" % (100*m.mean(), (a[..., 3] - a[..., 2])[m].mean(), (a[..., 4] - a[..., 3])[m].mean()))
print("all passes %.0f" % (a[..., 5] - a[..., 0]).mean())
