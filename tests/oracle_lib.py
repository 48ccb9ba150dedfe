"""ctypes binding of the CPU oracle (oracle/liboracle.so).

Test infrastructure: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# XPIC_ORACLE_SO: another build of the same sources (tests/test_oracle_sanitized.py loads the ASan/UBSan one)
_SO = os.environ.get("XPIC_ORACLE_SO") or os.path.join(ROOT, "oracle", "liboracle.so")

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)


def _dp(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_dp)


def _ip(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_ip)


def build():
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(ROOT, "oracle", "xpic_oracle.cpp")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)


_lib = None


def affinity_cpus():
    """logical CPUs this process may run on"""
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cgroup_cpus():
    """CPU quota of this process's cgroup (cgroup v2 cpu.max / v1 cfs quota), or None when there is none"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, -(-int(q) // int(per)))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return max(1, -(-q // per))
    except (OSError, ValueError):
        pass
    return None


POOL_CPU_SHARE = 16  # CPU share of a one-GPU box of the pool when neither the affinity mask nor a cgroup quota says so


def default_threads():
    """OpenMP team of the oracle = the CPUs this process really has: min(affinity mask, cgroup quota).  The GPU box shows
    every logical CPU of the host in its affinity mask while its share is a fraction of them (an oversubscribed team makes
    the thousands of small parallel regions of the oracle's GMRES crawl): without a quota to read, a mask wider than
    2 x POOL_CPU_SHARE is cut to POOL_CPU_SHARE.  XPIC_ORACLE_THREADS overrides all of it."""
    cap = os.environ.get("XPIC_ORACLE_THREADS")
    if cap:
        return max(1, int(cap))
    n = affinity_cpus()
    q = cgroup_cpus()
    if q is not None:
        return max(1, min(n, q))
    return n if n <= 2 * POOL_CPU_SHARE else POOL_CPU_SHARE


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        build()
    # The GPU box exposes far more logical CPUs than its CPU share: an oversubscribed, spin-waiting OpenMP team
    # makes the thousands of small parallel regions of the oracle's GMRES crawl.  Cap the team and wait passively.
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    L = C.CDLL(_SO)
    L.orc_set_threads.argtypes = [C.c_int]
    L.orc_set_threads(default_threads())
    L.orc_create.restype = C.c_void_p
    L.orc_create.argtypes = [C.c_int] * 4 + [C.c_double] * 4
    L.orc_destroy.argtypes = [C.c_void_p]
    L.orc_add_sort.argtypes = [C.c_void_p, C.c_int] + [C.c_double] * 6
    L.orc_load_maxwell_box.restype = C.c_long
    L.orc_load_maxwell_box.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.orc_add_particles.restype = C.c_long
    L.orc_add_particles.argtypes = [C.c_void_p, C.c_int, C.c_long, c_dp]
    L.orc_count.restype = C.c_long
    L.orc_count.argtypes = [C.c_void_p, C.c_int]
    L.orc_get_particles.restype = C.c_long
    L.orc_get_particles.argtypes = [C.c_void_p, C.c_int, c_dp, c_ip]
    L.orc_clear_particles.argtypes = [C.c_void_p, C.c_int]
    L.orc_set_field.argtypes = [C.c_void_p, C.c_char_p, c_dp]
    L.orc_get_field.argtypes = [C.c_void_p, C.c_char_p, c_dp]
    L.orc_get_sort_current.argtypes = [C.c_void_p, C.c_int, C.c_char_p, c_dp]
    L.orc_rot_apply.argtypes = [C.c_void_p, C.c_int, C.c_double, c_dp, c_dp]
    L.orc_matM_apply.argtypes = [C.c_void_p, c_dp, c_dp]
    L.orc_matL_apply.argtypes = [C.c_void_p, c_dp, c_dp]
    L.orc_div_neg.argtypes = [C.c_void_p, c_dp, c_dp]
    L.orc_lstencil_decode.argtypes = [C.c_int, C.c_int, c_ip, c_ip]
    L.orc_lstencil_encode.argtypes = [C.c_int] * 5
    L.orc_get_matL.argtypes = [C.c_void_p, c_dp]
    L.orc_gather_s2.argtypes = [C.c_void_p, c_dp, c_dp, c_dp, c_dp]
    L.orc_gather_s1.argtypes = [C.c_void_p, c_dp, c_dp, c_dp, c_dp]
    L.orc_esirkepov.argtypes = [C.c_void_p, C.c_long, c_dp, c_dp, C.c_double, c_dp]
    L.orc_basic_push.argtypes = [C.c_void_p]
    L.orc_update_cells.argtypes = [C.c_void_p, C.c_int]
    L.orc_ecsim_first_push.argtypes = [C.c_void_p, C.c_int]
    L.orc_ecsim_fill_current.argtypes = [C.c_void_p]
    L.orc_ecsim_second_push.argtypes = [C.c_void_p, C.c_int]
    L.orc_ecsimcorr_first_push.argtypes = [C.c_void_p, C.c_int]
    L.orc_ecsimcorr_second_push.argtypes = [C.c_void_p, C.c_int]
    L.orc_ecsimcorr_final_update.argtypes = [C.c_void_p, C.c_int]
    L.orc_calculate_energy.restype = C.c_double
    L.orc_calculate_energy.argtypes = [C.c_void_p, C.c_int]
    L.orc_ecsimcorr_scalars.argtypes = [C.c_void_p, C.c_int, c_dp]
    L.orc_solve.argtypes = [C.c_void_p, C.c_int, c_dp, c_dp, C.c_double, C.c_double, C.c_int, c_dp]
    L.orc_set_tolerances.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int]
    L.orc_step.argtypes = [C.c_void_p]
    L.orc_solve_stats.argtypes = [C.c_void_p, c_dp, C.POINTER(C.c_long), C.c_int]
    L.orc_energy.argtypes = [C.c_void_p, c_dp]
    L.orc_charge_collect.argtypes = [C.c_void_p]
    L.orc_charge_columns.argtypes = [C.c_void_p, c_dp]
    L.orc_charge_density.argtypes = [C.c_void_p, C.c_int, c_dp]
    L.orc_momentum.argtypes = [C.c_void_p, c_dp]
    L.orc_moment_density.argtypes = [C.c_void_p, C.c_int, c_dp]
    L.orc_cell_traversal.argtypes = [c_dp, c_dp, c_dp, C.c_int, c_dp]
    L.orc_cell_traversal.restype = C.c_int
    L.orc_implicit_esirkepov_interpolate.argtypes = [C.c_void_p, C.c_long, c_dp, c_dp, c_dp, c_dp]
    L.orc_implicit_esirkepov_decompose.argtypes = [C.c_void_p, C.c_long, c_dp, c_dp, c_dp, c_dp, C.c_char_p]
    L.orc_implicit_esirkepov_decompose.restype = C.c_int
    L.orc_boris_test_trajectory.argtypes = [C.c_int, C.c_char_p, c_dp, C.c_int]
    L.orc_spline.restype = C.c_double
    L.orc_spline.argtypes = [C.c_int, C.c_double]
    L.orc_shape_setup.argtypes = [c_dp, c_dp, c_dp, C.c_int, C.c_double, C.c_int, c_ip, c_ip, c_dp]
    L.orc_update_r.argtypes = [C.c_double, c_dp]
    L.orc_update_vEB.argtypes = [C.c_double, C.c_double, c_dp, c_dp, c_dp]
    L.orc_update_vX.argtypes = [C.c_char, C.c_double, C.c_double, c_dp, c_dp]
    L.orc_set_threads.argtypes = [C.c_int]
    _lib = L
    return L


SCHEMES = {"basic": 0, "ecsim": 1, "ecsimcorr": 2}


class OracleSim:
    """Thin object wrapper over the orc_* C API."""

    def __init__(self, scheme, n, d, dt):
        self.L = lib()
        self.n = tuple(int(v) for v in n)
        self.d = tuple(float(v) for v in d)
        self.dt = float(dt)
        self.scheme = scheme
        self.h = self.L.orc_create(SCHEMES[scheme], *self.n, *self.d, self.dt)
        self.N = self.n[0] * self.n[1] * self.n[2]
        self.nsorts = 0

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_destroy(self.h)
            self.h = None

    def add_sort(self, Np, n, q, m, T=(0.0, 0.0, 0.0)):
        self.nsorts += 1
        return self.L.orc_add_sort(self.h, Np, n, q, m, *[float(t) for t in T])

    def load_maxwell_box(self, sort, tov=True):
        return self.L.orc_load_maxwell_box(self.h, sort, int(tov))

    def add_particles(self, sort, pts):
        pts = np.ascontiguousarray(pts, dtype=np.float64)
        return self.L.orc_add_particles(self.h, sort, pts.shape[0], _dp(pts))

    def count(self, sort):
        return self.L.orc_count(self.h, sort)

    def particles(self, sort):
        n = self.count(sort)
        pts = np.zeros((n, 6))
        cells = np.zeros(n, dtype=np.int32)
        self.L.orc_get_particles(self.h, sort, _dp(pts), _ip(cells))
        return pts, cells

    def clear(self, sort):
        self.L.orc_clear_particles(self.h, sort)

    def fshape(self):
        return (self.n[2], self.n[1], self.n[0], 3)

    def set_field(self, name, v):
        v = np.ascontiguousarray(v, dtype=np.float64).reshape(self.fshape())
        assert self.L.orc_set_field(self.h, name.encode(), _dp(v)) == 0

    def get_field(self, name):
        v = np.zeros(self.fshape())
        assert self.L.orc_get_field(self.h, name.encode(), _dp(v)) == 0
        return v

    def sort_current(self, sort, which):
        v = np.zeros(self.fshape())
        assert self.L.orc_get_sort_current(self.h, sort, which.encode(), _dp(v)) == 0
        return v

    def rot(self, sign, alpha, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros(self.fshape())
        self.L.orc_rot_apply(self.h, sign, alpha, _dp(x), _dp(y))
        return y

    def matM(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros(self.fshape())
        self.L.orc_matM_apply(self.h, _dp(x), _dp(y))
        return y

    def matL_apply(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros(self.fshape())
        self.L.orc_matL_apply(self.h, _dp(x), _dp(y))
        return y

    def matL(self):
        out = np.zeros((self.N * 3, 123))
        self.L.orc_get_matL(self.h, _dp(out))
        return out

    def solve(self, op, rhs, rtol=1e-7, atol=1e-7, maxit=100):
        rhs = np.ascontiguousarray(rhs, dtype=np.float64)
        x = np.zeros(self.fshape())
        rn = np.zeros(1)
        its = self.L.orc_solve(self.h, op, _dp(rhs), _dp(x), rtol, atol, maxit, _dp(rn))
        return x, its, rn[0]

    def set_tolerances(self, rtol, atol, maxit):
        self.L.orc_set_tolerances(self.h, rtol, atol, maxit)

    def step(self):
        return self.L.orc_step(self.h)

    def solve_stats(self, reset=False):
        """(seconds, iterations) spent inside the Krylov solves of step() since the last reset."""
        sec, its = np.zeros(1), C.c_long()
        self.L.orc_solve_stats(self.h, _dp(sec), C.byref(its), int(reset))
        return float(sec[0]), int(its.value)

    def energy(self):
        out = np.zeros(4 + 2 * self.nsorts)
        self.L.orc_energy(self.h, _dp(out))
        return out

    def charge_collect(self):
        self.L.orc_charge_collect(self.h)

    def charge_columns(self):
        out = np.zeros(2 * self.nsorts + 2)
        self.L.orc_charge_columns(self.h, _dp(out))
        return out

    def momentum(self):
        """MomentumConservation::calculate: rows {Px, Py, Pz, QEx, QEy, QEz}, one per sort."""
        out = np.zeros((self.nsorts, 6))
        self.L.orc_momentum(self.h, _dp(out))
        return out

    def cell_traversal(self, end, start, max_pts=8):
        d3 = np.array(self.d, dtype=np.float64)
        pts = np.zeros((max_pts, 3))
        n = self.L.orc_cell_traversal(_dp(d3), _dp(np.ascontiguousarray(end, dtype=np.float64)),
                                      _dp(np.ascontiguousarray(start, dtype=np.float64)), max_pts, _dp(pts))
        return pts[:min(n, max_pts)].copy(), n

    def implicit_esirkepov_interpolate(self, rn, r0):
        rn, r0 = np.ascontiguousarray(rn, dtype=np.float64), np.ascontiguousarray(r0, dtype=np.float64)
        Ep, Bp = np.zeros_like(rn), np.zeros_like(rn)
        self.L.orc_implicit_esirkepov_interpolate(self.h, rn.shape[0], _dp(rn), _dp(r0), _dp(Ep), _dp(Bp))
        return Ep, Bp

    def implicit_esirkepov_decompose(self, alpha, v, rn, r0, field):
        alpha, v = np.ascontiguousarray(alpha, dtype=np.float64), np.ascontiguousarray(v, dtype=np.float64)
        rn, r0 = np.ascontiguousarray(rn, dtype=np.float64), np.ascontiguousarray(r0, dtype=np.float64)
        assert self.L.orc_implicit_esirkepov_decompose(self.h, rn.shape[0], _dp(alpha), _dp(v), _dp(rn), _dp(r0), field.encode()) == 0

    def moment_density(self, sort):
        out = np.zeros(self.fshape()[:3])
        self.L.orc_moment_density(self.h, sort, _dp(out))
        return out

    def charge_density(self, sort):
        rho = np.zeros((self.n[2], self.n[1], self.n[0]))
        self.L.orc_charge_density(self.h, sort, _dp(rho))
        return rho

    def ecsimcorr_scalars(self, sort):
        o = np.zeros(6)
        self.L.orc_ecsimcorr_scalars(self.h, sort, _dp(o))
        return dict(pred_w=o[0], corr_w=o[1], lambda_dK=o[2], pred_dK=o[3], corr_dK=o[4], energy=o[5])


def read_table(path):
    """Reads a reference `temporal/*.txt` table: header line + whitespace separated numbers."""
    with open(path) as f:
        header = f.readline().split()
        rows = [[float(x) for x in line.split()] for line in f if line.strip()]
    return header, np.array(rows)


def boris_trajectory(example, scheme_id, max_rows=4096):
    rows = np.zeros((max_rows, 7))
    n = lib().orc_boris_test_trajectory(example, scheme_id.encode(), _dp(rows), max_rows)
    assert n >= 0
    return rows[:n]
