// common.h -- shared declarations of the gfx950 implementation (see include/xpic_hip.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/xpic_hip.h"

// Experiment builds.  The kernels carry ablation switches, in-kernel timers and tunables (some of the switches produce
// wrong physics by design; every tunable changes generated code that no test has seen).  A value for any of them on the
// command line is honoured ONLY under -DXPIC_EXPERIMENT, and an object built that way says so through xpic_version()
// (bit 30 set), which tests/test_abi.py and xpic_amd/__init__.py refuse: a stray EXTRA=-DFILL_KCP=48 does not build a
// library that loads and passes the ABI test.  This header comes first in every translation unit, before the sources
// give the tunables their defaults, so "defined here" means "defined on the command line".
#if !defined(XPIC_EXPERIMENT) && (defined(FILL_EXP) || defined(FILL_STAMPS) || defined(FILL_GA_EXP) || defined(FILL_GA_NOCHAIN) || \
  defined(FILL_KCP) || defined(FILL_OCC) || defined(FILL_PITCH) || defined(FILL_WPITCH) || defined(FILL_WS_KCP) || \
  defined(FILL_WS_PRIO_C) || defined(FILL_WS_PRIO_P) || defined(FILL_WS_PRIO_F) || defined(ESK_EXP) || defined(ESK_STAMPS) || \
  defined(ESK_FLUSH) || defined(ESK_BOX_GENERIC) || defined(ESK_PIPE) || defined(ESK_LANES) || defined(ESK_COLS1) || defined(ESK_SEG0) || defined(ESK_OCC1) || \
  defined(ESK_DRAIN0) || defined(ESK_TILE_LATE) || defined(XPIC_CHEB_MIN_ZC) || defined(BAR_SCHED_SCALED) || defined(BAR_SCHED_GROUP) || \
  defined(XPIC_MAX_PER_Z) || defined(XPIC_SLAB_FIRST_TOUCH) || defined(XPIC_CHEB_M_BOUND) || defined(XPIC_BUCKET_CAP) || \
  defined(XPIC_DEFAULT_FUSED_REBIN) || defined(XPIC_DEFAULT_PRECOND) || defined(XPIC_DEFAULT_FILL_KERNEL) || defined(MATA_GROUP) || \
  defined(MATA_DEPTH) || defined(BAR_TRUNCATE))
#error "a build switch of the kernels was set on the command line: that is an experiment build, add -DXPIC_EXPERIMENT"
#endif
#ifdef XPIC_EXPERIMENT
#define XPIC_TU_EXPERIMENT 1
#else
#define XPIC_TU_EXPERIMENT 0
#endif
#ifndef XPIC_DEFAULT_FUSED_REBIN
#define XPIC_DEFAULT_FUSED_REBIN 1
#endif
#ifndef XPIC_DEFAULT_PRECOND
#define XPIC_DEFAULT_PRECOND 5 // 3: polynomial in matM + <matL>; 4: its rows scaled by the local density (precond.hip: 3x the
// convergence rate, but at 256^3 x 64 its third residual is 1.17e-7 |b| against the tolerance's 1e-7: still 4 iterations, each dearer);
// 5: 3 for a uniform plasma, 4 where the density varies (chosen per solve from the spread of matL's diagonal)
#endif
#ifndef XPIC_DEFAULT_FILL_KERNEL
#define XPIC_DEFAULT_FILL_KERNEL 0 // the assembly body a new context runs (xpic_set_fill_kernel): whichever measures faster at 256^3 x 64
#endif

namespace xpic {

void set_error(const std::string& msg);
// 1 when the translation unit was built with -DXPIC_EXPERIMENT (one per kernel file that has experiment switches)
int experiment_ecsim();
int experiment_ecsim_ws();
int experiment_esirkepov();
int experiment_fields();
int experiment_precond();
int experiment_particles();

#define XPIC_HIP(call)                                                                         \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      ::xpic::set_error(std::string(#call) + ": " + hipGetErrorString(e_) + " at " __FILE__ + \
        ":" + std::to_string(__LINE__));                                                       \
      return 1;                                                                                \
    }                                                                                          \
  } while (0)

#define XPIC_CHECK(cond, msg)                  \
  do {                                         \
    if (!(cond)) {                             \
      ::xpic::set_error(std::string(msg));     \
      return 2;                                \
    }                                          \
  } while (0)

#define XPIC_CALL(call)        \
  do {                         \
    int rc_ = (call);          \
    if (rc_ != 0) return rc_;  \
  } while (0)

constexpr int kLStencil = XPIC_LSTENCIL;
// matL is blocked by 4 in x: one row block (c1, z, y) is stored as [x/4][k][x%4] with the 123 coefficients padded to
// 124, so a 4-wide x-block of a row is 31 whole 128-byte lines and the assembly flushes whole lines (ecsim.hip).
constexpr int kLPad = 124;
constexpr int kLBlock = kLPad * 4; // doubles per (row block, x-block)

// ---------------------------------------------------------------------------------------------
// Grid of one z-slab.  Field vectors live in HBM as structure-of-arrays
//     F[c][zs][y][x],  zs in [0, nzs),  nzs = nzl + 2*G
// x and y are periodic and wrapped in the kernels; z is wrapped in the kernels when the slab is the
// whole box (G == 0, nranks == 1) and otherwise served by G ghost planes filled by the halo exchange.
// ---------------------------------------------------------------------------------------------
struct GridDev {
  int nx, ny, nzl; // local owned planes
  int nzg;         // global nz
  int z0;          // global index of the first owned plane
  int G;           // ghost planes on each side
  int nzs;         // stored planes
  double dx, dy, dz, dt;
  // all three spacings are powers of two: x * inv[a] == x / d[a] bit for bit, and the particle kernels take the
  // multiplications (a wave-uniform branch) instead of three fp64 divisions per position without changing a result
  double inv[3];
  int pow2;
  double Lx, Ly, Lz; // global box (World::set_geometry: geom = n * d, world.cpp:80-91)
  long plane;        // nx*ny
  long cstride;      // nzs*plane
  long nown;         // nzl*plane  (owned nodes per component)

  __host__ __device__ inline int wx(int x) const { return x < 0 ? x + nx : (x >= nx ? x - nx : x); }
  __host__ __device__ inline int wy(int y) const { return y < 0 ? y + ny : (y >= ny ? y - ny : y); }
  // local z (relative to the first owned plane, may be in [-G, nzl+G)) -> stored plane
  __host__ __device__ inline int wz(int z) const
  {
    if (G == 0) return z < 0 ? z + nzl : (z >= nzl ? z - nzl : z);
    return z + G;
  }
  __host__ __device__ inline long node(int x, int y, int zs) const { return ((long)zs * ny + y) * nx + x; }
  // wrapped access: x,y,z are local signed indices
  __host__ __device__ inline long nodew(int x, int y, int z) const { return node(wx(x), wy(y), wz(z)); }
  // matL: x-blocks per row, stored row planes (one ghost row plane on each side with z-neighbours), element offset
  __host__ __device__ inline int nbx() const { return (nx + 3) >> 2; }
  __host__ __device__ inline int nzp() const { return nzl + (G ? 2 : 0); }
  __host__ __device__ inline long lplane() const { return (long)ny * nbx() * kLBlock; }
  __host__ __device__ inline long lindex(int c1, int zp, int y, int x, int k) const
  {
    return ((((long)c1 * nzp() + zp) * ny + y) * nbx() + (x >> 2)) * kLBlock + k * 4 + (x & 3);
  }
};

constexpr int kCellStartPad = 16; // readable ints behind cell_start[ncell]

struct SortDev {
  double* r[3];   // positions  (SoA)
  double* v[3];   // velocities (SoA; `Point::p` holds velocity in these schemes)
  double* r2[3];  // second buffer of the out-of-place sort
  double* v2[3];
  int* cell;      // new local cell of each particle (or -1: dropped)
  int* rank;      // arrival rank of the particle inside its new cell
  int* cell_count;
  int* cell_start; // [ncell+1 (+ kCellStartPad readable)] exclusive prefix of cell_count
  int* src;        // deferred scatter: source index (old order) of the particle that belongs in slot d of the new order
  int* bucket;     // the same, written by the binning itself: bucket[cell * bucket_cap + arrival rank] (+ one flag word at the end:
  int bucket_cap;  // a cell took more arrivals than bucket_cap); 0: no buckets
  long ncell;
  const double* inc; // z-slabs: the records received from the neighbours ({r, v} per record); a source index -1 - i means record i of it
};

struct Sort {
  xpic_sort_params par;
  int64_t cap = 0;
  int64_t n = 0;
  // cell[], rank[], cell_count (and the migration buffers) already hold the binning of r + v * prebinned_step
  bool prebinned = false;
  // Deferred scatter (sort_rebin(.., defer = true)): cell_start / src describe the NEW order while r, v still hold the OLD,
  // un-moved records; the mass-matrix assembly reads them through src, applies the move and writes the sorted records into
  // r2, v2 on its way (ecsim.hip) -- the scatter pass of the re-binning is gone.  Everything else calls sort_materialize first.
  bool bucket_written = false; // the binning whose keys cell[] / rank[] hold also filled the buckets
  bool keys_valid = true;      // false: that binning wrote the buckets and the counts only (particles.hip: rebuild_keys)
  bool bucket_off = false;     // a cell overflowed its bucket once: pre-binnings write the keys again
  int def_n_in = 0; // records of the deferred re-binning that came from the neighbouring slabs (they lie in mig_recv)
  bool deferred = false, def_wrap = false, def_bucket = false; // def_bucket: the assembly reads the binning's buckets, not src
  double def_step = 0;
  int64_t def_n_old = 0;
  double prebinned_step = 0;
  int64_t prebinned_n = 0;
  SortDev d{};
  double* J = nullptr;      // basic: J
  double* currI = nullptr;  // ecsim: currI
  double* currJe = nullptr; // ecsimcorr: currJe
  double* rho = nullptr;    // ChargeConservation: last collected charge density (component 0 of a vector)
  // particle migration between z-slabs (nranks > 1)
  double* mig_send[2] = {nullptr, nullptr};
  double* mig_recv = nullptr;
  int *mig_cell = nullptr, *mig_rank = nullptr, *mig_count = nullptr;
  int mig_cap = 0;
  double energy = 0, pred_w = 0, corr_w = 0, pred_dK = 0, corr_dK = 0, lambda_dK = 0;
};

struct Comm {
  int kind = 0; // 0 single rank, 1 RCCL, 2 host callbacks
  int rank = 0, nranks = 1;
  void* nccl = nullptr;
  xpic_comm_callbacks cb{};
  void* host[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t host_bytes = 0;
  // traffic counters (xpic_comm_stats): point-to-point messages and bytes SENT by this rank, all-reduces and their payload
  int64_t sent_msgs = 0, sent_bytes = 0, allreduces = 0, allreduce_bytes = 0;
};

struct ProfileEntry {
  int64_t launches = 0;
  double total_ms = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

}  // namespace xpic

struct xpic_ctx {
  xpic_geometry geom;
  int scheme;
  xpic::GridDev g;
  hipStream_t stream = nullptr;
  long nvec = 0;    // doubles per stored field vector (3*cstride)
  long ncell = 0;   // local cells
  double* field[XPIC_NFIELDS] = {};
  double* matL = nullptr; // [c1][nzl][ny][123][nx]
  int* ltab = nullptr;    // [1296] block entry -> packed (k, c1, o1) descriptor
  int* fill_err = nullptr; // set by k_ecsim_fill_ws when one of its bounded waits gave up
  int fused_rebin = XPIC_DEFAULT_FUSED_REBIN; // ecsim step: the re-binning's scatter is deferred -- 1: the assembly gathers, moves and writes
                                              // the sorted copy; 2: the assembly only gathers, k_second_push writes the sorted copy; 0: scatter first
  int gather_window = 1 << 28;  // slots the gathering assembly reaches by 32-bit offsets around a pencil's first slot (xpic_debug_set)
  int pencil_limit = 1 << 29;   // x-pencils of this many particles or more are scattered first (the sorted copy's 32-bit offsets)
  int fill_kernel = XPIC_DEFAULT_FILL_KERNEL; // 1: warp-specialised assembly (8-wave workgroups, producer / consumer waves) where the grid allows; 0: classic
  std::vector<xpic::Sort> sorts;
  // Krylov workspace
  double* kry_V = nullptr; // (m+1) vectors
  double* kry_w = nullptr;
  unsigned* esk_tab = nullptr; // precomposed rounds of the Esirkepov pushes (esirkepov.hip: k_esk_rounds), allocated by the first push
  size_t esk_tab_bytes = 0;
  int esk_pre = 1;             // 0: the pushes compose their rounds themselves (XPIC_ESK_PRE=0; the fall-back of a pencil with too many rounds)
  double* kry_t = nullptr; // preconditioner scratch (fp32 copy of its input)
  double* kry_Z = nullptr; // flexible GMRES: the preconditioned basis z_j = P v_j (allocated by the first preconditioned solve)
  double* kry_p[3] = {nullptr, nullptr, nullptr}; // Chebyshev preconditioner work vectors
  int precond = XPIC_DEFAULT_PRECOND; // 0 none; Chebyshev polynomial (right preconditioning) in matM on fp32 (1) / fp64 (2) work vectors,
                       // 3: in matM + the translation average of matL (precond.hip) for the predict solve; 4: rows scaled by the local
                       // density; 5: 3 or 4, chosen per solve
  int num_cus = 256; // hipDeviceAttributeMultiprocessorCount (the colour schedule of the assembly counts workgroup rounds)
  int cheb_degree = 0; // steps of the Chebyshev iteration on matM (set at create from the spectral interval)
  int cheb_degree_M = 0; // the same iteration as the preconditioner of a solve ON matM (ecsimcorr's "correct"): a tighter bound pays there
  int cheb_degree_user = 0; // explicit degree from xpic_set_preconditioner (0: automatic)
  int cheb_degree_auto = 0, cheb_degree_M_auto = 0; // the automatic degrees of xpic_create (restored by degree <= 0)
  float* abar32 = nullptr;    // kind 3: the 3 x 124 coefficients of Abar = matM + <matL>
  double* abar_work = nullptr; // sums, matM's coefficients, fp64 Abar, per-row partials
  float* abar_r = nullptr;     // kind 4: local density ratio of every row (fp32 field layout) + one word for their maximum
  double abar_lo = 0, abar_hi = 0; // spectral interval of Abar
  double abar_gershgorin = 0;      // 2 + Gershgorin lower bound of Lbar: kind 3 is used only while this is positive
  bool abar_valid = false;
  bool abar_proven = false;        // its Gershgorin bound proves the polynomial's interval; otherwise the surrogate runs on probation
  double debug_surrogate_scale = 1.0; // xpic_debug_set(XPIC_DEBUG_SURROGATE_SCALE): <matL> times this in the surrogate (tests)
  bool abar_scaled = false;        // this solve's surrogate has its rows scaled by the local density (kind 4; kind 5 where it pays)
  double* red_partial = nullptr; // reduction partials
  double* red_out = nullptr;     // device results (pinned mirror below)
  double* red_host = nullptr;
  int* scan_tmp = nullptr;
  long scan_tmp_n = 0;
  double rtol = 1e-7, atol = 1e-7;
  int maxit = 100;
  xpic::Comm comm;
  double* halo_buf[4] = {}; // send down, send up, recv from up, recv from down
  double* lrow_buf[2] = {nullptr, nullptr}; // matL ghost rows received from the upper / lower neighbour (3 row planes each; slabs)
  bool lrow_posted = false, lrow_on_comm_stream = false;
  // Copy-engine path of the large messages (comm.hip: xpic_comm_peer_import): the neighbours' receive buffers mapped into this
  // process (peer_lrow[0]: the lower neighbour's "from above" buffer, [1]: the upper neighbour's "from below" buffer); the
  // ghost rows then travel as hipMemcpyAsync on copy_stream -- SDMA engines between two GPUs, no workgroup slot taken from
  // the assembly -- and the NEXT ring exchange with the neighbour is what tells it that they have arrived.
  double* peer_lrow[2] = {nullptr, nullptr};
  void* peer_mapped[2] = {nullptr, nullptr}; // what hipIpcOpenMemHandle returned (closed with the context); null for same-process peers
  bool peer_copy = false;          // xpic_set_overlap bit 2
  bool lrow_by_copy = false;       // this step's ghost rows went by copy
  bool peer_pending = false;       // copies issued on copy_stream that no exchange has been ordered behind yet
  int peer_exchanges = 0;          // ring exchanges since the post (the arrival signal)
  hipStream_t copy_stream = nullptr;
  hipEvent_t copy_ev[2] = {nullptr, nullptr};
  size_t halo_bytes = 0;
  // overlapped operator applies (fields.hip: op_apply_overlapped): RCCL traffic of a posted halo runs on its own stream
  hipStream_t comm_stream = nullptr;
  hipEvent_t comm_ev[2] = {nullptr, nullptr}; // packed (compute -> comm), ghosts in place (comm -> compute)
  bool halo_posted = false;
  bool overlap = false; // operator applies with their halo exchange posted beside the interior rows (xpic_set_overlap bit 0).  Off by
                        // default since round 4: on a self-ring the split apply costs 1.7 ms of a 28.7 ms slab step more than it hides
  bool overlap_lrows = false; // the matL ghost-row exchange beside the assembly's interior colours (xpic_set_overlap bit 1)
  bool overlap_explicit = false; // set by xpic_set_overlap: xpic_comm_init_rccl then leaves the choice alone
  bool profiling = false;
  std::map<std::string, xpic::ProfileEntry> prof;
  std::vector<hipEvent_t> event_pool;
};

namespace xpic {

// RAII-less timer: records an event pair around a launch when profiling is on.
struct Timed {
  xpic_ctx* c;
  ProfileEntry* e = nullptr;
  hipEvent_t a{}, b{};
  Timed(xpic_ctx* ctx, const char* name);
  ~Timed();
};

hipEvent_t get_event(xpic_ctx* c);

constexpr int kRedBlocks = 1024; // partial-sum blocks of the two-stage reductions
constexpr int kMaxDots = 32;

// fields.hip
int vec_set(xpic_ctx* c, double* y, double a);
int vec_copy(xpic_ctx* c, double* y, const double* x);
int vec_axpy(xpic_ctx* c, double* y, double a, const double* x);
int vec_axpby(xpic_ctx* c, double* y, double a, double b, const double* x);          // y = a x + b y
int vec_waxpby(xpic_ctx* c, double* w, double a, const double* x, double b, const double* y); // w = a x + b y
int vec_scale_to(xpic_ctx* c, double* y, double a, const double* x);                 // y = a x
int vec_dot_host(xpic_ctx* c, const double* x, const double* y, double* out);
int vec_mdot_host(xpic_ctx* c, const double* w, const double* V, int nv, double* out); // out[i] = w . V_i
int vec_mdot_ww_host(xpic_ctx* c, const double* w, const double* V, int nv, double* out, double* ww); // + w . w, ONE reduction
int vec_maxpy_norm_host(xpic_ctx* c, double* w, const double* V, int nv, const double* h, double* nrm2); // w -= sum h_i V_i
int vec_maxpy_scaled(xpic_ctx* c, const double* w, const double* V, int nv, const double* h, double* out, double scale,
  double* nrm2); // out = (w - sum h_i V_i) * scale
int vec_maxpy(xpic_ctx* c, double* x, const double* V, int nv, const double* y);     // x += sum y_i V_i
int rot_apply(xpic_ctx* c, int sign, double alpha, const double* x, double* y, bool add);
int matM_apply(xpic_ctx* c, const double* x, double* y, bool add);
int matL_apply(xpic_ctx* c, const double* x, double* y, bool add);
int matA_apply(xpic_ctx* c, const double* x, double* y);
int op_apply_overlapped(xpic_ctx* c, bool with_L, double* x, double* y); // halo exchange of x beside the interior rows
int cheb_matM_inverse(xpic_ctx* c, const double* r, double* out, int degree);
int halo_fill_f32(xpic_ctx* c, float* f, int width);
// precond.hip
int abar_alloc(xpic_ctx* c);                                      // kind 3's buffers (with the context, not inside a solve)
int abar_update(xpic_ctx* c);                                     // Abar = matM + <matL> from the assembled matL
int cheb_abar_inverse(xpic_ctx* c, const double* r, double* out); // out ~ Abar^-1 r
int cg_apply_dot_host(xpic_ctx* c, const double* p, double* Ap, double* pAp);                       // Ap = matM p, p . Ap
int cg_update_host(xpic_ctx* c, double alpha, const double* p, const double* Ap, double* x, double* r, double* rr);
int div_neg_add(xpic_ctx* c, double* v3, double* out_scalar);
int scalar_norm12_host(xpic_ctx* c, const double* f, double* out2);
int field_import(xpic_ctx* c, double* dst_soa, const double* src_aos_host);
int field_export(xpic_ctx* c, const double* src_soa, double* dst_aos_host);
int field_stats_host(xpic_ctx* c, const double* f, double* sumsq, double* mean3);
int halo_fill(xpic_ctx* c, double* f, int width = 3); // ghost planes <- neighbours' owned planes (no-op when G == 0)
int halo_fill2(xpic_ctx* c, double* f0, double* f1, int width = 3); // two vectors, one message per neighbour
int halo_add(xpic_ctx* c, double* f, int width);      // owned planes += neighbours' ghost planes (DMLocalToGlobal ADD)
int matL_exchange_ghost_rows(xpic_ctx* c);  // blocking: post + finish
int matL_ghost_rows_post(xpic_ctx* c);      // ship this slab's two ghost row planes (on the communication stream when overlapping)
int matL_ghost_rows_finish(xpic_ctx* c);    // add the neighbours' rows into the first / last owned row plane

// particles.hip
int sort_alloc(xpic_ctx* c, Sort& s, int64_t cap);
void sort_free(Sort& s);
// defer: 0 the scatter runs here; 1 it is left to the mass-matrix assembly (buckets of source indices where the binning
// wrote them, else the index k_index builds); 2 it is left to the next Esirkepov push (index only)
int sort_rebin(xpic_ctx* c, Sort& s, double step, bool wrap, int defer = 0);
int sort_materialize(xpic_ctx* c, Sort& s); // run the scatter a deferred re-binning left out (no-op otherwise)
void sort_deferred_done(Sort& s);             // the assembly has written the sorted records: swap the buffers
int sort_move(xpic_ctx* c, Sort& s, double step); // r += step*v in place, cells left stale // (optional move by step*v), wrap, bin, scatter
int sort_append_host(xpic_ctx* c, Sort& s, int64_t n, const double* pts6, int64_t* added);
int sort_download(xpic_ctx* c, Sort& s, double* pts6, int32_t* cell_of);
int sort_fill_synthetic(xpic_ctx* c, Sort& s, const xpic_load_params& lp);
int sort_occupancy(xpic_ctx* c, Sort& s, int64_t* out8);
int ecsim_second_push(xpic_ctx* c, Sort& s, const double* E, const double* B, bool prebin = false);
int charge_density(xpic_ctx* c, Sort& s, double* rho_vec);
int moment_density(xpic_ctx* c, Sort& s, double* vec);
int kinetic_sums_host(xpic_ctx* c, Sort& s, double* out5);   // local sums of vx, vy, vz, v^2 and the count
int kinetic_sums_global(xpic_ctx* c, Sort& s, double* out5); // summed over the slabs
int scale_velocities(xpic_ctx* c, Sort& s, double lambda);
int momentum_sums_global(xpic_ctx* c, Sort& s, const double* E, double* out6);

// ecsim.hip
int ecsim_fill_sort(xpic_ctx* c, Sort& s, const double* B, double* currI_sort, double* matL, bool first_sort, bool post_ghost_rows);
int ecsim_fill_check(xpic_ctx* c); // the assembly's device-side error word, agreed on by all slabs (once per assembly)
int build_ltab(xpic_ctx* c);
void ecsim_fill_variant(const xpic_ctx* c, int* p2, int* fx, int* ws);

// esirkepov.hip: mode 0 basic::push, 1 ecsimcorr first_push, 2 ecsimcorr second_push
int esirkepov_push(xpic_ctx* c, Sort& s, int mode, const double* E, const double* B, double* J, double* pred_w_host);
bool esk_table_alloc(xpic_ctx* c); // the precomposed rounds' table (esirkepov.hip): true if it is there

// comm.hip
int comm_ring(xpic_ctx* c, const void* down, size_t ndown, const void* up, size_t nup, void* from_up, size_t nfrom_up,
  void* from_down, size_t nfrom_down);
int comm_allreduce_sum(xpic_ctx* c, double* dbuf, int n);
int comm_allreduce_sum_host(xpic_ctx* c, double* hbuf, int n);
int comm_allreduce_max_host(xpic_ctx* c, double* v); // one non-negative value
void comm_free(xpic_ctx* c);
int ensure_halo_buf(xpic_ctx* c, size_t bytes);
int peer_order(xpic_ctx* c); // an exchange is about to be issued: order it behind the peer copies still in flight

// krylov.hip
int solve(xpic_ctx* c, int op, const double* rhs, double* x, double rtol, double atol, int maxit, int* its,
  int* reason, double* rnorm);

}  // namespace xpic
