import numpy as np, sys
a = np.loadtxt(sys.argv[1]).reshape(32, 32, 4, 8)
ok = (a[..., 0] > 0).all(axis=(1, 2)); a = a[ok]
d = np.diff(a[..., :5], axis=-1)
for k, nm in enumerate(["chunk start -> after first wave_sync (incl. waiting for the particle loads)", "phase 1 compute + LDS stores", "second wave_sync (+ issue of next-pass loads)", "phase 2 (MFMA)"]):
    print("%-80s mean %7.0f median %7.0f" % (nm, d[..., k].mean(), np.median(d[..., k])))
