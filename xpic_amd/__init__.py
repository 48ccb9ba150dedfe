"""xpic_amd -- MI355X (gfx950) implementation of xpic's per-timestep hot path.

This package is a thin ctypes view of the C ABI in include/xpic_hip.h (xpic_amd/libxpic_hip.so, built from
xpic_amd/csrc/*.hip by `make` / `__graft_entry__.build()`).  There is NO CPU fallback: if the shared library
or a HIP device is missing, every entry point raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libxpic_hip.so")

E, B, B0, J, EP, EC, CURRI, CURRJE, W0, W1, W2 = range(11)
BASIC, ECSIM, ECSIMCORR = 0, 1, 2
OP_MATA_GMRES, OP_MATM_GMRES, OP_MATM_CG = 0, 1, 2
LSTENCIL = 123
SCHEMES = {"basic": BASIC, "ecsim": ECSIM, "ecsimcorr": ECSIMCORR}

# every symbol include/xpic_hip.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "xpic_last_error", "xpic_version", "xpic_create", "xpic_destroy", "xpic_synchronize", "xpic_add_sort",
    "xpic_sort_add_particles", "xpic_sort_count", "xpic_sort_get_particles", "xpic_sort_clear",
    "xpic_sort_fill_synthetic", "xpic_sort_load_synthetic", "xpic_sort_occupancy", "xpic_field_set", "xpic_field_get", "xpic_sort_current_get", "xpic_vec_set",
    "xpic_vec_axpy", "xpic_vec_axpby", "xpic_vec_dot", "xpic_vec_norm2", "xpic_rot_apply", "xpic_matM_apply",
    "xpic_matL_apply", "xpic_matA_apply", "xpic_matL_get", "xpic_lstencil_decode", "xpic_ecsim_first_push",
    "xpic_update_cells", "xpic_ecsim_fill_current", "xpic_ecsim_second_push", "xpic_basic_push",
    "xpic_ecsimcorr_first_push", "xpic_ecsimcorr_second_push", "xpic_ecsimcorr_final_update",
    "xpic_calculate_energy", "xpic_ecsimcorr_scalars", "xpic_solve", "xpic_set_tolerances", "xpic_set_preconditioner", "xpic_set_overlap", "xpic_comm_stats", "xpic_set_fill_kernel", "xpic_set_fused_rebin", "xpic_get_fill_variant", "xpic_debug_set", "xpic_step",
    "xpic_energy", "xpic_momentum", "xpic_charge_density", "xpic_moment_density", "xpic_cell_traversal", "xpic_implicit_esirkepov_interpolate",
    "xpic_implicit_esirkepov_decompose", "xpic_charge_collect", "xpic_charge_columns", "xpic_comm_rccl_unique_id", "xpic_comm_init_rccl", "xpic_comm_init_callbacks", "xpic_comm_size", "xpic_comm_peer_export", "xpic_comm_peer_import",
    "xpic_profile_enable", "xpic_profile_reset", "xpic_profile_get", "xpic_probe_copy_bandwidth",
]


def csrc_hash():
    """sha1 over the kernel sources (xpic_amd/csrc/*, sorted by name): the code version a PMC traffic file under
    profiles/ was taken on is stamped with it, and bench.py quotes that file only while the hash still matches."""
    import hashlib

    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    h = hashlib.sha1()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()


class XpicError(RuntimeError):
    pass


class Geometry(C.Structure):
    _fields_ = [("n", C.c_int32 * 3), ("d", C.c_double * 3), ("dt", C.c_double), ("periodic", C.c_int32 * 3),
                ("rank", C.c_int32), ("nranks", C.c_int32), ("device", C.c_int32), ("self_ring", C.c_int32)]


DEBUG_GATHER_WINDOW, DEBUG_PENCIL_LIMIT, DEBUG_SURROGATE_SCALE = 0, 1, 2  # include/xpic_hip.h: xpic_debug_set
PEER_BLOB_BYTES = 256  # include/xpic_hip.h: XPIC_PEER_BLOB_BYTES
VERSION_EXPERIMENT_BIT = 0x40000000  # include/xpic_hip.h: XPIC_VERSION_EXPERIMENT_BIT


class LoadParams(C.Structure):
    _fields_ = [("ppc", C.c_int32), ("profile", C.c_int32), ("vth", C.c_double), ("drift", C.c_double * 3),
                ("profile_param", C.c_double * 4), ("seed", C.c_uint64)]


LOAD_PROFILES = {"poisson": 0, "uniform": 0, "regular": 1, "gradient": 2, "blob": 3}  # include/xpic_hip.h: XPIC_LOAD_*


class SortParams(C.Structure):
    _fields_ = [("Np", C.c_int32), ("n", C.c_double), ("q", C.c_double), ("m", C.c_double)]


_lib = None
c_dp = C.POINTER(C.c_double)


def load_library():
    """Loads libxpic_hip.so; raises if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise XpicError(f"{LIB_PATH} is missing: run `make` (or __graft_entry__.build()) first; "
                        "xpic_amd has no CPU fallback")
    L = C.CDLL(LIB_PATH)
    L.xpic_last_error.restype = C.c_char_p
    if L.xpic_version() & VERSION_EXPERIMENT_BIT and os.environ.get("XPIC_ALLOW_EXPERIMENT") != "1":
        raise XpicError(f"{LIB_PATH} was built with -DXPIC_EXPERIMENT (ablation switches / in-kernel timers: its results "
                        "may be wrong by design); rebuild with `make clean all`, or set XPIC_ALLOW_EXPERIMENT=1 for a "
                        "measurement script")
    _lib = L
    return L


def _dp(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_dp)


class Context:
    """One z-slab context = one `interfaces::Simulation` backend instance."""

    def __init__(self, scheme, n, d, dt, device=0, rank=0, nranks=1, self_ring=False):
        self.L = load_library()
        g = Geometry()
        g.n[:] = [int(v) for v in n]
        g.d[:] = [float(v) for v in d]
        g.dt = float(dt)
        g.periodic[:] = [1, 1, 1]
        g.rank, g.nranks, g.device, g.self_ring = rank, nranks, device, int(self_ring)
        self.n = tuple(int(v) for v in n)
        self.d = tuple(float(v) for v in d)
        self.dt = float(dt)
        self.scheme = scheme
        self.rank, self.nranks = rank, nranks
        self.nzl = self.n[2] // nranks  # planes of this z-slab
        self.z0 = rank * self.nzl
        self.h = C.c_void_p()
        self._ck(self.L.xpic_create(C.byref(g), SCHEMES[scheme], C.byref(self.h)))
        self.N = self.n[0] * self.n[1] * self.nzl  # local cells
        self.nsorts = 0
        self._cb = None

    # ---- z-slab communicator
    def comm_init_rccl(self, id128):
        buf = (C.c_char * 128).from_buffer_copy(bytes(id128))
        self._ck(self.L.xpic_comm_init_rccl(self.h, buf))

    def comm_init_callbacks(self, sendrecv, allreduce_sum):
        """sendrecv(down: bytes, up: bytes, n_from_up, n_from_down) -> (from_up: bytes, from_down: bytes);
        allreduce_sum(np.ndarray[float64]) -> reduces in place.  Used by tests (torch.distributed / gloo)."""
        SR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                         C.c_void_p, C.c_size_t)
        AR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)

        def _sr(user, down, nd, up, nu, fu, nfu, fd, nfd):
            try:
                a, b = sendrecv(C.string_at(down, nd) if nd else b"", C.string_at(up, nu) if nu else b"", nfu, nfd)
                if nfu:
                    C.memmove(fu, a, nfu)
                if nfd:
                    C.memmove(fd, b, nfd)
                return 0
            except Exception as e:  # pragma: no cover
                print("xpic comm callback failed:", e, flush=True)
                return 1

        def _ar(user, buf, n):
            try:
                arr = np.ctypeslib.as_array(buf, shape=(n,))
                allreduce_sum(arr)
                return 0
            except Exception as e:  # pragma: no cover
                print("xpic comm callback failed:", e, flush=True)
                return 1

        class CB(C.Structure):
            _fields_ = [("user", C.c_void_p), ("sendrecv", SR), ("allreduce_sum", AR)]

        self._cb = CB(None, SR(_sr), AR(_ar))
        self._ck(self.L.xpic_comm_init_callbacks(self.h, C.byref(self._cb)))

    def comm_peer_export(self):
        """the blob (bytes) that tells a z-neighbour where this rank receives its matL ghost rows (copy-engine path)"""
        buf = (C.c_char * PEER_BLOB_BYTES)()
        self._ck(self.L.xpic_comm_peer_export(self.h, buf))
        return bytes(buf)

    def comm_peer_import(self, lower_blob, upper_blob):
        lo = (C.c_char * PEER_BLOB_BYTES).from_buffer_copy(bytes(lower_blob))
        up = (C.c_char * PEER_BLOB_BYTES).from_buffer_copy(bytes(upper_blob))
        self._ck(self.L.xpic_comm_peer_import(self.h, lo, up))

    def comm_size(self):
        n = C.c_int()
        self._ck(self.L.xpic_comm_size(self.h, C.byref(n)))
        return n.value

    def _ck(self, rc):
        if rc != 0:
            raise XpicError(f"xpic error {rc}: {self.L.xpic_last_error().decode()}")

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.xpic_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- particles
    def add_sort(self, Np, n, q, m, capacity):
        p = SortParams(int(Np), float(n), float(q), float(m))
        out = C.c_int()
        self._ck(self.L.xpic_add_sort(self.h, C.byref(p), C.c_int64(int(capacity)), C.byref(out)))
        self.nsorts += 1
        return out.value

    def add_particles(self, sort, pts):
        pts = np.ascontiguousarray(pts, dtype=np.float64)
        added = C.c_int64()
        self._ck(self.L.xpic_sort_add_particles(self.h, sort, C.c_int64(pts.shape[0]), _dp(pts), C.byref(added)))
        return added.value

    def count(self, sort):
        n = C.c_int64()
        self._ck(self.L.xpic_sort_count(self.h, sort, C.byref(n)))
        return n.value

    def particles(self, sort):
        n = self.count(sort)
        pts = np.zeros((n, 6))
        cells = np.zeros(n, dtype=np.int32)
        self._ck(self.L.xpic_sort_get_particles(self.h, sort, _dp(pts), cells.ctypes.data_as(C.POINTER(C.c_int32))))
        return pts, cells

    def clear(self, sort):
        self._ck(self.L.xpic_sort_clear(self.h, sort))

    def fill_synthetic(self, sort, ppc, vth, seed=1, regular=False):
        self._ck(self.L.xpic_sort_fill_synthetic(self.h, sort, int(ppc), C.c_double(vth), C.c_uint64(seed), int(regular)))

    def load_synthetic(self, sort, ppc, vth, seed=1, profile="poisson", drift=(0.0, 0.0, 0.0), param=(0.0, 0.0)):
        """fill_synthetic with a drift (MaxwellianMomentum's px, py, pz: an extension of the reference's JSON surface) and
        a density profile: "gradient" (param[0] : 1 along x) or "blob" (fraction param[0] in a Gaussian of param[1] cells)"""
        lp = LoadParams()
        lp.ppc, lp.profile, lp.vth, lp.seed = int(ppc), LOAD_PROFILES[profile], float(vth), int(seed)
        lp.drift[:] = [float(v) for v in drift]
        lp.profile_param[:] = [float(v) for v in param] + [0.0] * (4 - len(param))
        self._ck(self.L.xpic_sort_load_synthetic(self.h, sort, C.byref(lp)))

    def occupancy(self, sort):
        """dict of the cell / pencil occupancy statistics of a sort (include/xpic_hip.h: xpic_sort_occupancy)"""
        o = (C.c_int64 * 8)()
        self._ck(self.L.xpic_sort_occupancy(self.h, sort, o))
        keys = ("max_cell", "cells_over_64", "cells_over_128", "cells_over_bucket", "max_pencil", "min_pencil", "empty_cells",
                "bucket_cap")
        return dict(zip(keys, (int(v) for v in o)))

    # ---- fields
    def fshape(self):
        return (self.nzl, self.n[1], self.n[0], 3)

    def set_field(self, f, v):
        v = np.ascontiguousarray(v, dtype=np.float64).reshape(self.fshape())
        self._ck(self.L.xpic_field_set(self.h, f, _dp(v)))

    def get_field(self, f):
        v = np.zeros(self.fshape())
        self._ck(self.L.xpic_field_get(self.h, f, _dp(v)))
        return v

    def sort_current(self, sort, which):
        v = np.zeros(self.fshape())
        self._ck(self.L.xpic_sort_current_get(self.h, sort, which, _dp(v)))
        return v

    def vec_set(self, y, a):
        self._ck(self.L.xpic_vec_set(self.h, y, C.c_double(a)))

    def vec_axpy(self, y, a, x):
        self._ck(self.L.xpic_vec_axpy(self.h, y, C.c_double(a), x))

    def vec_axpby(self, y, a, b, x):
        self._ck(self.L.xpic_vec_axpby(self.h, y, C.c_double(a), C.c_double(b), x))

    def vec_dot(self, x, y):
        o = C.c_double()
        self._ck(self.L.xpic_vec_dot(self.h, x, y, C.byref(o)))
        return o.value

    def vec_norm2(self, x):
        o = C.c_double()
        self._ck(self.L.xpic_vec_norm2(self.h, x, C.byref(o)))
        return o.value

    # ---- operators
    def rot_apply(self, sign, alpha, x, y, add=False):
        self._ck(self.L.xpic_rot_apply(self.h, sign, C.c_double(alpha), x, y, int(add)))

    def matM_apply(self, x, y, add=False):
        self._ck(self.L.xpic_matM_apply(self.h, x, y, int(add)))

    def matL_apply(self, x, y, add=False):
        self._ck(self.L.xpic_matL_apply(self.h, x, y, int(add)))

    def matA_apply(self, x, y):
        self._ck(self.L.xpic_matA_apply(self.h, x, y))

    def matL(self):
        out = np.zeros((self.N * 3, LSTENCIL))
        self._ck(self.L.xpic_matL_get(self.h, _dp(out)))
        return out

    # ---- phases
    def ecsim_first_push(self, sort):
        self._ck(self.L.xpic_ecsim_first_push(self.h, sort))

    def update_cells(self, sort):
        n = C.c_int64()
        self._ck(self.L.xpic_update_cells(self.h, sort, C.byref(n)))
        return n.value

    def ecsim_fill_current(self):
        self._ck(self.L.xpic_ecsim_fill_current(self.h))

    def ecsim_second_push(self, sort):
        self._ck(self.L.xpic_ecsim_second_push(self.h, sort))

    def basic_push(self, sort):
        self._ck(self.L.xpic_basic_push(self.h, sort))

    def ecsimcorr_first_push(self, sort):
        self._ck(self.L.xpic_ecsimcorr_first_push(self.h, sort))

    def ecsimcorr_second_push(self, sort):
        self._ck(self.L.xpic_ecsimcorr_second_push(self.h, sort))

    def ecsimcorr_final_update(self, sort):
        self._ck(self.L.xpic_ecsimcorr_final_update(self.h, sort))

    def calculate_energy(self, sort):
        o = C.c_double()
        self._ck(self.L.xpic_calculate_energy(self.h, sort, C.byref(o)))
        return o.value

    def ecsimcorr_scalars(self, sort):
        o = np.zeros(6)
        self._ck(self.L.xpic_ecsimcorr_scalars(self.h, sort, _dp(o)))
        return dict(pred_w=o[0], corr_w=o[1], lambda_dK=o[2], pred_dK=o[3], corr_dK=o[4], energy=o[5])

    def solve(self, op, rhs, x, rtol=1e-7, atol=1e-7, maxit=100):
        its, reason, rn = C.c_int(), C.c_int(), C.c_double()
        self._ck(self.L.xpic_solve(self.h, op, rhs, x, C.c_double(rtol), C.c_double(atol), maxit, C.byref(its),
                                   C.byref(reason), C.byref(rn)))
        return its.value, reason.value, rn.value

    def set_tolerances(self, rtol, atol, maxit):
        self._ck(self.L.xpic_set_tolerances(self.h, C.c_double(rtol), C.c_double(atol), maxit))

    def set_preconditioner(self, kind, degree=0):
        self._ck(self.L.xpic_set_preconditioner(self.h, int(kind), int(degree)))

    def comm_stats(self, reset=False):
        """(messages sent, bytes sent, all-reduces, all-reduce payload bytes) of this rank since the last reset"""
        o = (C.c_int64 * 4)()
        self._ck(self.L.xpic_comm_stats(self.h, o, int(reset)))
        return tuple(int(v) for v in o)

    def set_fill_kernel(self, kind):
        """0 (default): classic 4-wave assembly kernel; 1: warp-specialised 16-wave kernel where the grid allows (measured slower)"""
        self._ck(self.L.xpic_set_fill_kernel(self.h, int(kind)))

    def set_fused_rebin(self, on):
        """1 (default): a re-binning's scatter is left to the next kernel that reads every particle (the assembly in ecsim / ecsimcorr, the next push in basic); 2: to ecsim's second push; 0: scatter first"""
        self._ck(self.L.xpic_set_fused_rebin(self.h, int(on)))

    def fill_variant(self):
        """(power-of-two spacings, full-chunk body, warp-specialised body) of the next assembly"""
        o = (C.c_int * 3)()
        self._ck(self.L.xpic_get_fill_variant(self.h, o))
        return bool(o[0]), bool(o[1]), bool(o[2])

    def debug_set(self, what, value):
        """test hooks (include/xpic_hip.h): DEBUG_GATHER_WINDOW, DEBUG_PENCIL_LIMIT"""
        self._ck(self.L.xpic_debug_set(self.h, int(what), C.c_int64(int(value))))

    def set_overlap(self, on):
        self._ck(self.L.xpic_set_overlap(self.h, int(on)))  # bit 0: operator halos, bit 1: matL ghost rows

    def step(self):
        its = C.c_int()
        self._ck(self.L.xpic_step(self.h, C.byref(its)))
        return its.value

    def energy(self):
        out = np.zeros(4 + 2 * self.nsorts)
        self._ck(self.L.xpic_energy(self.h, _dp(out)))
        return out

    def momentum(self):
        out = np.zeros((self.nsorts, 6))
        self._ck(self.L.xpic_momentum(self.h, _dp(out)))
        return out

    def charge_density(self, sort):
        rho = np.zeros((self.nzl, self.n[1], self.n[0]))
        self._ck(self.L.xpic_charge_density(self.h, sort, _dp(rho)))
        return rho

    def moment_density(self, sort):
        out = np.zeros((self.nzl, self.n[1], self.n[0]))
        self._ck(self.L.xpic_moment_density(self.h, sort, _dp(out)))
        return out

    def cell_traversal(self, end, start, max_pts=8):
        end, start = np.ascontiguousarray(end, dtype=np.float64), np.ascontiguousarray(start, dtype=np.float64)
        n = end.shape[0]
        pts = np.zeros((n, max_pts, 3))
        counts = np.zeros(n, dtype=np.int32)
        self._ck(self.L.xpic_cell_traversal(self.h, C.c_int64(n), _dp(end), _dp(start), max_pts, _dp(pts),
                                            counts.ctypes.data_as(C.POINTER(C.c_int))))
        return pts, counts

    def implicit_esirkepov_interpolate(self, rn, r0):
        rn, r0 = np.ascontiguousarray(rn, dtype=np.float64), np.ascontiguousarray(r0, dtype=np.float64)
        Ep, Bp = np.zeros_like(rn), np.zeros_like(rn)
        self._ck(self.L.xpic_implicit_esirkepov_interpolate(self.h, C.c_int64(rn.shape[0]), _dp(rn), _dp(r0), _dp(Ep), _dp(Bp)))
        return Ep, Bp

    def implicit_esirkepov_decompose(self, alpha, v, rn, r0, field):
        alpha, v = np.ascontiguousarray(alpha, dtype=np.float64), np.ascontiguousarray(v, dtype=np.float64)
        rn, r0 = np.ascontiguousarray(rn, dtype=np.float64), np.ascontiguousarray(r0, dtype=np.float64)
        self._ck(self.L.xpic_implicit_esirkepov_decompose(self.h, C.c_int64(rn.shape[0]), _dp(alpha), _dp(v), _dp(rn), _dp(r0), field))

    def charge_collect(self):
        self._ck(self.L.xpic_charge_collect(self.h))

    def charge_columns(self):
        out = np.zeros(2 * self.nsorts + 2)
        self._ck(self.L.xpic_charge_columns(self.h, _dp(out)))
        return out

    def synchronize(self):
        self._ck(self.L.xpic_synchronize(self.h))

    # ---- measurement
    def profile_enable(self, on=True):
        self._ck(self.L.xpic_profile_enable(self.h, int(on)))

    def profile_reset(self):
        self._ck(self.L.xpic_profile_reset(self.h))

    def profile_get(self, name):
        n, ms = C.c_int64(), C.c_double()
        self._ck(self.L.xpic_profile_get(self.h, name.encode(), C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def probe_copy_bandwidth(self, nbytes=1 << 30, reps=10):
        o = C.c_double()
        self._ck(self.L.xpic_probe_copy_bandwidth(self.h, C.c_int64(nbytes), reps, C.byref(o)))
        return o.value


def rccl_unique_id():
    L = load_library()
    buf = (C.c_char * 128)()
    if L.xpic_comm_rccl_unique_id(buf) != 0:
        raise XpicError(L.xpic_last_error().decode())
    return bytes(buf)


def lstencil_decode(c1, k):
    L = load_library()
    c2 = C.c_int()
    d = (C.c_int * 3)()
    L.xpic_lstencil_decode(c1, k, C.byref(c2), d)
    return c2.value, (d[0], d[1], d[2])
