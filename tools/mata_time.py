import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import xpic_amd as X
n = 256
ctx = X.Context("ecsim", (n, n, n), (0.5,) * 3, 1.0)
ctx.set_preconditioner(0)
for _ in range(3):
    ctx.matA_apply(X.E, X.W1)
ctx.synchronize()
ctx.profile_enable(True); ctx.profile_reset()
for _ in range(10):
    ctx.matA_apply(X.E, X.W1)
ctx.synchronize()
nl, ms = ctx.profile_get("matA_apply")
print("matA_apply %.3f ms (%d)" % (ms / nl, nl), "halo", ctx.profile_get("halo"))
