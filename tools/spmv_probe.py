import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, xpic_amd as X
ctx = X.Context("ecsim", (256, 256, 256), (0.5,) * 3, 1.0)
ctx.vec_set(X.E, 1.0)
for _ in range(2): ctx.matA_apply(X.E, X.W0)
ctx.profile_enable(True); ctx.profile_reset()
for _ in range(10): ctx.matA_apply(X.E, X.W0)
ctx.synchronize()
n, ms = ctx.profile_get("matA_apply")
print("matA_apply", n, ms / n, "ms  ->", (123 * 3 * 8 + 48) * ctx.N / (ms / n * 1e-3) / 1e9, "GB/s")
