import numpy as np, sys
a = np.loadtxt(sys.argv[1]).reshape(32, 32, 4, 8)   # wg, chunk, wave, stamp
ok = (a[..., 0] > 0).all(axis=(1, 2))
a = a[ok]
print("WGs with stamps:", a.shape[0])
d = np.diff(a, axis=-1)
names = ["passes(ph1+ph2)", "prefetch+rmw issue", "barrier1 wait", "win init+bar2", "merge+bar3", "flush", "bar4"]
for k, nm in enumerate(names): print("%-22s mean %8.0f  median %8.0f  max %8.0f" % (nm, d[..., k].mean(), np.median(d[..., k]), d[..., k].max()))
per = a[:, 1:, :, 0] - a[:, :-1, :, 0]
print("chunk period: mean %.0f median %.0f" % (per.mean(), np.median(per)))
gap = a[:, 1:, :, 0] - a[:, :-1, :, 7]
print("gap end->next start: mean %.0f" % gap.mean())
# spread of pass time across the 4 waves of a chunk
pt = d[..., 0]
print("passes: min over waves %.0f, max over waves %.0f" % (pt.min(axis=2).mean(), pt.max(axis=2).mean()))
