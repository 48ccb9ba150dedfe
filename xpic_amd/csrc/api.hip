// api.hip -- the C ABI of include/xpic_hip.h: context life cycle, boundary copies, per-phase entry points
// and the timestep drivers that mirror timestep_implementation() of the reference's schemes.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "lstencil.h"

// first-touch assembly on a z-slab with neighbours (ecsim_fill_current): 0 clears all of matL every step (round 2)
#ifndef XPIC_SLAB_FIRST_TOUCH
#define XPIC_SLAB_FIRST_TOUCH 1
#endif
// Chebyshev degree of the preconditioner of a solve ON matM (ecsimcorr's "correct"): there the polynomial approximates
// the inverse of the operator itself, and a tighter bound (0.25 % instead of 8 %) trades stencil passes on fp32 vectors for
// outer iterations with their Gram-Schmidt passes and all-reduces: at dt = 1, dx = 0.5 degree 12 instead of 6, 3 instead
// of 6 iterations per solve (128^3: 2.41 -> 1.72 ms; degrees 8 / 16 / 20 / 30: 2.32 / 2.14 / 1.69 / 2.32)
#ifndef XPIC_CHEB_M_BOUND
#define XPIC_CHEB_M_BOUND 0.00125
#endif
namespace xpic {

static thread_local std::string g_error;
void set_error(const std::string& msg) { g_error = msg; }

hipEvent_t get_event(xpic_ctx* c)
{
  if (!c->event_pool.empty()) {
    hipEvent_t e = c->event_pool.back();
    c->event_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}

Timed::Timed(xpic_ctx* ctx, const char* name) : c(ctx)
{
  if (!c->profiling) return;
  e = &c->prof[name];
  a = get_event(c);
  b = get_event(c);
  (void)hipEventRecord(a, c->stream);
}

Timed::~Timed()
{
  if (!e) return;
  (void)hipEventRecord(b, c->stream);
  e->pending.emplace_back(a, b);
}

static int resolve_profile(xpic_ctx* c)
{
  XPIC_HIP(hipStreamSynchronize(c->stream));
  for (auto& kv : c->prof) {
    for (auto& pr : kv.second.pending) {
      float ms = 0;
      XPIC_HIP(hipEventElapsedTime(&ms, pr.first, pr.second));
      kv.second.total_ms += ms;
      kv.second.launches += 1;
      c->event_pool.push_back(pr.first);
      c->event_pool.push_back(pr.second);
    }
    kv.second.pending.clear();
  }
  return 0;
}

// matL holds one extra row plane below and above the slab when there are neighbours (rows of their nodes that
// this rank's cells contribute to)
static size_t matL_doubles(const GridDev& g) { return (size_t)3 * g.nzp() * g.lplane(); }

// the 30 preconditioned basis vectors z_j = P v_j of the flexible GMRES exist exactly while a preconditioner is
// selected: allocated with the context, not by the first solve (every slab sizes its memory up front, the same way)
static int ensure_flexible_workspace(xpic_ctx* c)
{
  if (c->precond == 0 || !c->kry_V) {
    if (c->kry_Z) { XPIC_HIP(hipStreamSynchronize(c->stream)); XPIC_HIP(hipFree(c->kry_Z)); c->kry_Z = nullptr; }
    return 0;
  }
  if (c->precond >= 3 && c->matL) XPIC_CALL(abar_alloc(c)); // kinds 3, 4: the surrogate's buffers and matM's coefficients
  if (c->kry_Z) return 0;
  XPIC_HIP(hipMalloc(&c->kry_Z, sizeof(double) * c->nvec * 30));
  XPIC_HIP(hipMemsetAsync(c->kry_Z, 0, sizeof(double) * c->nvec * 30, c->stream));
  return 0;
}

static bool valid_field(int f) { return f >= 0 && f < XPIC_NFIELDS; }

#define CTX_CHECK(c) XPIC_CHECK((c) != nullptr, "null context")
#define FIELD_CHECK(f) XPIC_CHECK(valid_field(f) && ctx->field[f], "unknown or unallocated field id")
#define SORT_CHECK(s) XPIC_CHECK((s) >= 0 && (s) < (int)ctx->sorts.size(), "unknown sort id")

// ---- scheme drivers --------------------------------------------------------------------------------

// ecsim::Simulation::advance_fields(ksp, curr, out)  (src/impls/ecsim/simulation.cpp:255-279)
static int advance_fields(xpic_ctx* c, int op, const double* curr, double* out, int* its)
{
  double* rhs = c->field[XPIC_W0];
  double* Bm = c->field[XPIC_W1];
  XPIC_CALL(vec_waxpby(c, Bm, 1.0, c->field[XPIC_B], -1.0, c->field[XPIC_B0]));   // B - B0        :260
  XPIC_CALL(vec_waxpby(c, rhs, 2.0, c->field[XPIC_E], -c->g.dt, curr));          // 2E - dt curr  :262-263
  XPIC_CALL(halo_fill(c, Bm));
  XPIC_CALL(rot_apply(c, -1, +c->g.dt, Bm, rhs, true));                           // + dt rotB(B)  :264, :554
  int reason;
  double rn;
  Timed t(c, op == XPIC_OP_MATA_GMRES ? "solve_matA" : "solve_matM");
  XPIC_CALL(solve(c, op, rhs, out, c->rtol, c->atol, c->maxit, its, &reason, &rn));
  return 0;
}

// ecsim::Simulation::fill_ecsim_current (src/impls/ecsim/simulation.cpp:336-368, 471-484) incl. the
// clear_sources of :157-172
static int ecsim_fill_current(xpic_ctx* c)
{
  const GridDev& g = c->g;
  // MatZeroEntries (:164).  On a single slab every matL entry has a local writer, so the first species' assembly
  // stores instead of adding (first touch) and no clearing pass is needed.  With z-neighbours some lines of the ghost
  // planes and of the first / last owned plane are written only by the neighbour's ghost-row exchange: those four
  // planes are cleared (below), the rest is first touch too.
  bool any = false;
  for (auto& s : c->sorts) any = any || s.n > 0;
  // (and a y or z extent of 2 folds two row offsets of a pencil onto one stream: ecsim.hip adds those with atomics)
  const bool first_touch = any && (g.G == 0 || XPIC_SLAB_FIRST_TOUCH) && g.ny >= 3 && g.nzl >= 3;
  if (!first_touch) {
    Timed t(c, "matL_zero");
    XPIC_HIP(hipMemsetAsync(c->matL, 0, sizeof(double) * matL_doubles(g), c->stream));
  }
  else if (g.G > 0) {
    // A slab: the lines of the two ghost planes and of the first / last owned plane whose column lies beyond the slab have
    // no local writer (the neighbour's ghost-row exchange adds into them): those four planes are cleared, everything
    // else is first touch as on a single slab (clearing all of matL was 6.6 GB of stores per step on a 32-plane slab)
    Timed t(c, "matL_zero");
    const long per = g.lplane();
    const int nzp = g.nzl + 2;
    for (int c1 = 0; c1 < 3; ++c1) {
      double* base = c->matL + (long)c1 * nzp * per;
      XPIC_HIP(hipMemsetAsync(base, 0, sizeof(double) * 2 * per, c->stream));
      XPIC_HIP(hipMemsetAsync(base + (long)(nzp - 2) * per, 0, sizeof(double) * 2 * per, c->stream));
    }
  }
  XPIC_CALL(vec_set(c, c->field[XPIC_CURRI], 0.0));
  XPIC_HIP(hipMemsetAsync(c->fill_err, 0, sizeof(int), c->stream));
  bool first_sort = first_touch;
  XPIC_CALL(halo_fill(c, c->field[XPIC_B], 1)); // DMGlobalToLocal(B) :474
  // the LAST species completes the ghost rows of matL: its assembly posts their exchange behind its boundary colours
  // (fields.hip: matL_ghost_rows_post), the neighbours' rows are added when every launch is done.  "Last" is the last of
  // the list, not the last with particles on this slab: point-to-point messages are matched per peer in issue order, so
  // the post must sit at the same place of the message sequence on every rank (a slab without particles of that species
  // posts from ecsim_fill_sort's empty-species return)
  const int last = (int)c->sorts.size() - 1;
  for (size_t i = 0; i < c->sorts.size(); ++i) {
    Sort& s = c->sorts[i];
    XPIC_HIP(hipMemsetAsync(s.currI, 0, sizeof(double) * c->nvec, c->stream));
    XPIC_CALL(ecsim_fill_sort(c, s, c->field[XPIC_B], s.currI, c->matL, first_sort && s.n > 0, (int)i == last && g.G > 0));
    if (s.n > 0) first_sort = false;
    XPIC_CALL(halo_add(c, s.currI, 1));                          // DMLocalToGlobal(ADD) particles.cpp:56
    XPIC_CALL(vec_axpy(c, c->field[XPIC_CURRI], 1.0, s.currI)); // particles.cpp:57
  }
  XPIC_CALL(matL_ghost_rows_finish(c)); // (posts first when there is no species at all: the cleared rows travel)
  return ecsim_fill_check(c);
}

// ecsim::Simulation::final_update (src/impls/ecsim/simulation.cpp:241-253).  Ep's ghost planes are valid at both call
// sites (filled in front of the second push, resp. in front of `currI += matL Ec` whose operand becomes Ep by the swap;
// nothing has written Ep since): no exchange here.
static int ecsim_final_update(xpic_ctx* c)
{
  XPIC_CALL(vec_axpby(c, c->field[XPIC_E], 2.0, -1.0, c->field[XPIC_EP]));          // E = 2 Ep - E
  XPIC_CALL(rot_apply(c, +1, -c->g.dt, c->field[XPIC_EP], c->field[XPIC_B], true)); // B -= dt rot(+) Ep
  return 0;
}

// ecsim::Simulation::timestep_implementation (src/impls/ecsim/simulation.cpp:145-253)
static int step_ecsim(xpic_ctx* c, int* its)
{
  // first_push + update_cells :174-189 (the scatter of the re-binning deferred into the assembly's particle loads)
  for (auto& s : c->sorts) XPIC_CALL(sort_rebin(c, s, c->g.dt, true, c->fused_rebin != 0));
  XPIC_CALL(ecsim_fill_current(c));
  XPIC_CALL(advance_fields(c, XPIC_OP_MATA_GMRES, c->field[XPIC_CURRI], c->field[XPIC_EP], its)); // :191-210
  XPIC_CALL(halo_fill2(c, c->field[XPIC_EP], c->field[XPIC_B]));
  // second_push :212-239.  Positions did not change since the re-bin of first_push and are already wrapped,
  // so correct_coordinates() + update_cells() of :227,:233 are the identity here.
  // the next step starts with sort_rebin(dt) of exactly this state: bin it in the same pass (Sort::prebinned)
  for (auto& s : c->sorts) XPIC_CALL(ecsim_second_push(c, s, c->field[XPIC_EP], c->field[XPIC_B], true));
  XPIC_CALL(ecsim_final_update(c));
  return 0;
}

// basic::Simulation::timestep_implementation (src/impls/basic/simulation.cpp:30-100);
// rotE = -(dt/2) rot(+), rotB = +dt rot(-) (:23-24)
static int step_basic(xpic_ctx* c)
{
  const double dt = c->g.dt;
  double *E = c->field[XPIC_E], *B = c->field[XPIC_B], *B0 = c->field[XPIC_B0], *J = c->field[XPIC_J];
  XPIC_CALL(vec_set(c, J, 0.0));
  // push_particles :45-72
  XPIC_CALL(vec_axpy(c, B, -1.0, B0));
  XPIC_CALL(halo_fill(c, E));
  XPIC_CALL(rot_apply(c, +1, -(0.5 * dt), E, B, true));
  XPIC_CALL(vec_axpy(c, B, +1.0, B0));
  XPIC_CALL(halo_fill(c, B));
  for (auto& s : c->sorts) {
    XPIC_HIP(hipMemsetAsync(s.J, 0, sizeof(double) * c->nvec, c->stream));
    XPIC_CALL(esirkepov_push(c, s, 0, E, B, s.J, nullptr)); // sort->push()
    XPIC_CALL(halo_add(c, s.J, 3));                         // DMLocalToGlobal(ADD) particles.cpp:50
    XPIC_CALL(vec_axpy(c, J, 1.0, s.J));                    // VecAXPY(simulation_.J, 1, J) particles.cpp:51
    // sort->update_cells(); the next step's push reads every particle: it gathers through the index and writes the sorted copy
    XPIC_CALL(sort_rebin(c, s, 0.0, true, c->fused_rebin == 1 && c->comm.kind == 0 ? 2 : 0));
  }
  // push_fields :74-100
  XPIC_CALL(vec_axpy(c, B, -1.0, B0));
  XPIC_CALL(rot_apply(c, +1, -(0.5 * dt), E, B, true));
  XPIC_CALL(halo_fill(c, B));
  XPIC_CALL(rot_apply(c, -1, +dt, B, E, true));
  XPIC_CALL(vec_axpy(c, E, -dt, J));
  XPIC_CALL(vec_axpy(c, B, +1.0, B0));
  return 0;
}

static int calc_energy(xpic_ctx* c, Sort& s)
{
  double o[5];
  XPIC_CALL(kinetic_sums_global(c, s, o));
  s.energy = 0.5 * s.par.m * (s.par.n / s.par.Np) * o[3]; // Energy::get_kinetic summed, ecsimcorr/particles.cpp:134-150
  return 0;
}

// ecsimcorr::Particles::final_update (src/impls/ecsimcorr/particles.cpp:93-126)
static int corr_final_update(xpic_ctx* c, Sort& s)
{
  XPIC_CALL(vec_dot_host(c, s.currJe, c->field[XPIC_EC], &s.corr_w));
  const double K0 = s.energy;
  XPIC_CALL(calc_energy(c, s));
  const double K = s.energy;
  const double lambda2 = 1.0 + c->g.dt * (s.corr_w - s.pred_w) / K;
  XPIC_CALL(scale_velocities(c, s, std::sqrt(lambda2)));
  s.lambda_dK = (lambda2 - 1.0) * K;
  s.pred_dK = K - K0;
  s.corr_dK = lambda2 * K - K0;
  s.energy = lambda2 * K;
  return 0;
}

// ecsimcorr::Simulation::timestep_implementation (src/impls/ecsimcorr/simulation.cpp:21-90)
static int step_ecsimcorr(xpic_ctx* c, int* its)
{
  double *Ep = c->field[XPIC_EP], *B = c->field[XPIC_B];
  // clear_sources :34-49
  XPIC_CALL(vec_set(c, c->field[XPIC_CURRJE], 0.0));
  for (auto& s : c->sorts) {
    XPIC_HIP(hipMemsetAsync(s.currJe, 0, sizeof(double) * c->nvec, c->stream));
    XPIC_CALL(calc_energy(c, s));
  }
  // first_push: half move + Esirkepov, re-bin, ECSIM current + matL
  for (auto& s : c->sorts) XPIC_CALL(esirkepov_push(c, s, 1, nullptr, nullptr, s.currJe, nullptr));
  // (the assembly that follows reads every particle: the re-binning's scatter is deferred into it, as in the ecsim step)
  for (auto& s : c->sorts) XPIC_CALL(sort_rebin(c, s, 0.0, true, c->fused_rebin == 1));
  XPIC_CALL(ecsim_fill_current(c));
  int its0 = 0, its1 = 0;
  XPIC_CALL(advance_fields(c, XPIC_OP_MATA_GMRES, c->field[XPIC_CURRI], Ep, &its0)); // KSP "predict"
  // second_push (ecsim/simulation.cpp:212-239 with the ecsimcorr particles)
  XPIC_CALL(halo_fill2(c, Ep, B));
  for (auto& s : c->sorts) {
    XPIC_CALL(esirkepov_push(c, s, 2, Ep, B, s.currJe, &s.pred_w));
    XPIC_CALL(halo_add(c, s.currJe, 3));                          // DMLocalToGlobal(ADD) particles.cpp:88
    XPIC_CALL(vec_axpy(c, c->field[XPIC_CURRJE], 1.0, s.currJe)); // particles.cpp:89
  }
  for (auto& s : c->sorts) XPIC_CALL(sort_rebin(c, s, 0.0, true)); // correct_coordinates + update_cells
  // correct_fields :52-63: KSP "correct" on matM with the Esirkepov current
  XPIC_CALL(advance_fields(c, XPIC_OP_MATM_GMRES, c->field[XPIC_CURRJE], c->field[XPIC_EC], &its1));
  // final_update :65-90
  for (auto& s : c->sorts) XPIC_CALL(corr_final_update(c, s));
  XPIC_CALL(halo_fill(c, c->field[XPIC_EC]));
  XPIC_CALL(matL_apply(c, c->field[XPIC_EC], c->field[XPIC_CURRI], true)); // currI += matL Ec :78
  std::swap(c->field[XPIC_EP], c->field[XPIC_EC]);                         // VecSwap(Ep, Ec) :86
  XPIC_CALL(ecsim_final_update(c));
  *its = its0 + its1;
  return 0;
}

}  // namespace xpic

using namespace xpic;

extern "C" {

const char* xpic_last_error(void) { return g_error.c_str(); }
int xpic_version(void)
{
  // bit 30: some object of this library was built with -DXPIC_EXPERIMENT (ablation switches, in-kernel timers)
  const bool exp = XPIC_TU_EXPERIMENT || experiment_ecsim() || experiment_ecsim_ws() || experiment_esirkepov() || experiment_fields() ||
    experiment_precond() || experiment_particles();
  return XPIC_VERSION | (exp ? XPIC_VERSION_EXPERIMENT_BIT : 0);
}

int xpic_create(const xpic_geometry* geom, int scheme, xpic_ctx** out)
{
  XPIC_CHECK(geom && out, "null argument");
  XPIC_CHECK(geom->periodic[0] && geom->periodic[1] && geom->periodic[2], "only DM_BOUNDARY_PERIODIC is supported");
  // ecsim runs on the reference's own default geometry (config.json: 2 x 2 x 32 cells): the CIC footprints fold
  // periodically once (every offset is within [-2, 2] and the extent >= 2).  The 2nd-order shapes of basic / ecsimcorr
  // need their 6-node tiles (esirkepov.hip).
  XPIC_CHECK(geom->n[0] >= 2 && geom->n[1] >= 2 && geom->n[2] >= 2, "every grid extent must be >= 2 cells");
  XPIC_CHECK(scheme == XPIC_ECSIM || (geom->n[0] >= 4 && geom->n[1] >= 4 && geom->n[2] >= 4),
    "basic and ecsimcorr need every grid extent >= 4 cells (>= 6 for their pushes)");
  XPIC_CHECK(geom->nranks >= 1 && geom->rank >= 0 && geom->rank < geom->nranks, "bad rank / nranks");
  XPIC_CHECK(geom->n[2] % geom->nranks == 0, "nz must be divisible by the number of z-slabs");
  // (self_ring slabs too: halo_add's two sides are one launch and must not overlap, 2 x 3 planes)
  XPIC_CHECK((geom->nranks == 1 && !geom->self_ring) || geom->n[2] / geom->nranks >= 6, "a z-slab must hold at least 6 planes");
  XPIC_CHECK(scheme == XPIC_BASIC || scheme == XPIC_ECSIM || scheme == XPIC_ECSIMCORR, "unknown scheme");
  int ndev = 0;
  XPIC_HIP(hipGetDeviceCount(&ndev));
  XPIC_CHECK(ndev > 0, "no HIP device: the xpic HIP path has no CPU fallback");
  XPIC_HIP(hipSetDevice(geom->device));
  xpic_ctx* c = new xpic_ctx;
  c->geom = *geom;
  c->scheme = scheme;
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, geom->device) == hipSuccess && cus > 0) c->num_cus = cus;
  }
  GridDev& g = c->g;
  g.nx = geom->n[0]; g.ny = geom->n[1]; g.nzg = geom->n[2];
  g.nzl = g.nzg / geom->nranks;
  g.z0 = geom->rank * g.nzl;
  // ceil(shape_radius) + 1 ghost planes (c-2 .. c+3 of the 2nd-order shape pair) when there are z-neighbours.
  // geometry.self_ring keeps the ghost planes and the exchange layer on a single slab (its own neighbour): the
  // way to exercise the RCCL transport on a one-GPU box.
  XPIC_CHECK(!geom->self_ring || geom->nranks == 1, "self_ring is a single-slab option");
  g.G = (geom->nranks > 1 || geom->self_ring) ? 3 : 0;
  g.nzs = g.nzl + 2 * g.G;
  g.dx = geom->d[0]; g.dy = geom->d[1]; g.dz = geom->d[2]; g.dt = geom->dt;
  g.pow2 = 1;
  for (int a = 0; a < 3; ++a) {
    int ex;
    g.inv[a] = 1.0 / geom->d[a];
    if (std::frexp(geom->d[a], &ex) != 0.5) g.pow2 = 0;
  }
  g.Lx = g.nx * g.dx; g.Ly = g.ny * g.dy; g.Lz = g.nzg * g.dz;
  g.plane = (long)g.nx * g.ny;
  g.cstride = g.plane * g.nzs;
  g.nown = g.plane * g.nzl;
  c->nvec = 3 * g.cstride;
  c->ncell = g.nown;
  XPIC_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  for (int f = 0; f < XPIC_NFIELDS; ++f) {
    const bool need = scheme != XPIC_BASIC || (f != XPIC_EP && f != XPIC_EC && f != XPIC_CURRI && f != XPIC_CURRJE);
    if (!need) continue;
    XPIC_HIP(hipMalloc(&c->field[f], sizeof(double) * c->nvec));
    XPIC_HIP(hipMemsetAsync(c->field[f], 0, sizeof(double) * c->nvec, c->stream));
  }
  XPIC_HIP(hipMalloc(&c->red_partial, sizeof(double) * kMaxDots * kRedBlocks));
  XPIC_HIP(hipMalloc(&c->red_out, sizeof(double) * 128));
  XPIC_HIP(hipHostMalloc(&c->red_host, sizeof(double) * 64));
  if (scheme != XPIC_BASIC) {
    XPIC_HIP(hipMalloc(&c->matL, sizeof(double) * matL_doubles(g)));
    XPIC_HIP(hipMemsetAsync(c->matL, 0, sizeof(double) * matL_doubles(g), c->stream));
    XPIC_HIP(hipMalloc(&c->kry_V, sizeof(double) * c->nvec * 31));
    XPIC_HIP(hipMalloc(&c->kry_w, sizeof(double) * c->nvec));
    XPIC_HIP(hipMemsetAsync(c->kry_V, 0, sizeof(double) * c->nvec * 31, c->stream));
    XPIC_HIP(hipMemsetAsync(c->kry_w, 0, sizeof(double) * c->nvec, c->stream));
    XPIC_HIP(hipMalloc(&c->kry_t, sizeof(double) * c->nvec));
    XPIC_HIP(hipMemsetAsync(c->kry_t, 0, sizeof(double) * c->nvec, c->stream));
    for (int i = 0; i < 3; ++i) {
      XPIC_HIP(hipMalloc(&c->kry_p[i], sizeof(double) * c->nvec));
      XPIC_HIP(hipMemsetAsync(c->kry_p[i], 0, sizeof(double) * c->nvec, c->stream));
    }
    {
      // Chebyshev degree: error bound 2 rho^k / (1 + rho^2k) <= 8 % on the spectral interval of matM
      const double kappa = 1.0 + g.dt * g.dt * (1.0 / (g.dx * g.dx) + 1.0 / (g.dy * g.dy) + 1.0 / (g.dz * g.dz));
      const double rho = (std::sqrt(kappa) - 1.0) / (std::sqrt(kappa) + 1.0);
      int k = rho > 0 ? (int)std::ceil(std::log(0.04) / std::log(rho)) : 2;
      c->cheb_degree = c->cheb_degree_auto = k < 2 ? 2 : (k > 32 ? 32 : k);
      int kM = rho > 0 ? (int)std::ceil(std::log(XPIC_CHEB_M_BOUND) / std::log(rho)) : 2;
      c->cheb_degree_M = c->cheb_degree_M_auto = kM < 2 ? 2 : (kM > 48 ? 48 : kM);
    }
    XPIC_CALL(build_ltab(c));
    if (g.G > 0)
      for (int i = 0; i < 2; ++i) { // the neighbours' matL ghost rows (3 row planes each): sized with the context
        XPIC_HIP(hipMalloc(&c->lrow_buf[i], sizeof(double) * 3 * g.lplane()));
        XPIC_HIP(hipMemsetAsync(c->lrow_buf[i], 0, sizeof(double) * 3 * g.lplane(), c->stream));
      }
    XPIC_HIP(hipMalloc(&c->fill_err, sizeof(int)));
    XPIC_HIP(hipMemsetAsync(c->fill_err, 0, sizeof(int), c->stream));
    XPIC_CALL(ensure_flexible_workspace(c));
  }
  // the round table of the Esirkepov pushes is sized with the context too (a push finds it there; without room for it the
  // pushes compose their rounds themselves, on this rank alone: no collective depends on it)
  if (scheme != XPIC_ECSIM && g.nx >= 6 && g.ny >= 6 && g.nzl >= 6) esk_table_alloc(c);
  XPIC_HIP(hipStreamSynchronize(c->stream));
  *out = c;
  return 0;
}

int xpic_destroy(xpic_ctx* ctx)
{
  if (!ctx) return 0;
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& s : ctx->sorts) sort_free(s);
  for (int f = 0; f < XPIC_NFIELDS; ++f) (void)hipFree(ctx->field[f]);
  (void)hipFree(ctx->matL); (void)hipFree(ctx->ltab); (void)hipFree(ctx->fill_err); (void)hipFree(ctx->kry_V); (void)hipFree(ctx->kry_w); (void)hipFree(ctx->esk_tab);
  (void)hipFree(ctx->kry_t); (void)hipFree(ctx->kry_Z); (void)hipFree(ctx->kry_p[0]); (void)hipFree(ctx->kry_p[1]); (void)hipFree(ctx->kry_p[2]);
  (void)hipFree(ctx->red_partial); (void)hipFree(ctx->red_out); (void)hipHostFree(ctx->red_host);
  (void)hipFree(ctx->scan_tmp);
  (void)hipFree(ctx->abar32); (void)hipFree(ctx->abar_work); (void)hipFree(ctx->abar_r);
  (void)hipFree(ctx->lrow_buf[0]); (void)hipFree(ctx->lrow_buf[1]);
  for (int i = 0; i < 4; ++i) (void)hipFree(ctx->halo_buf[i]);
  comm_free(ctx);
  for (auto& kv : ctx->prof)
    for (auto& pr : kv.second.pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return 0;
}

int xpic_synchronize(xpic_ctx* ctx)
{
  CTX_CHECK(ctx);
  XPIC_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

int xpic_add_sort(xpic_ctx* ctx, const xpic_sort_params* p, int64_t capacity, int* sort_out)
{
  CTX_CHECK(ctx);
  XPIC_CHECK(p && p->Np > 0 && p->m != 0.0, "bad sort parameters");
  Sort s;
  s.par = *p;
  XPIC_CALL(sort_alloc(ctx, s, capacity));
  double** cur = ctx->scheme == XPIC_BASIC ? &s.J : &s.currI;
  XPIC_HIP(hipMalloc(cur, sizeof(double) * ctx->nvec));
  XPIC_HIP(hipMemsetAsync(*cur, 0, sizeof(double) * ctx->nvec, ctx->stream));
  if (ctx->scheme == XPIC_ECSIMCORR) {
    XPIC_HIP(hipMalloc(&s.currJe, sizeof(double) * ctx->nvec));
    XPIC_HIP(hipMemsetAsync(s.currJe, 0, sizeof(double) * ctx->nvec, ctx->stream));
  }
  ctx->sorts.push_back(s);
  if (sort_out) *sort_out = (int)ctx->sorts.size() - 1;
  return 0;
}

int xpic_sort_add_particles(xpic_ctx* ctx, int sort, int64_t n, const double* pts6, int64_t* added)
{
  CTX_CHECK(ctx); SORT_CHECK(sort);
  return sort_append_host(ctx, ctx->sorts[sort], n, pts6, added);
}

int xpic_sort_count(xpic_ctx* ctx, int sort, int64_t* count)
{
  CTX_CHECK(ctx); SORT_CHECK(sort);
  *count = ctx->sorts[sort].n;
  return 0;
}

int xpic_sort_get_particles(xpic_ctx* ctx, int sort, double* pts6, int32_t* cell_of)
{
  CTX_CHECK(ctx); SORT_CHECK(sort);
  return sort_download(ctx, ctx->sorts[sort], pts6, cell_of);
}

int xpic_sort_clear(xpic_ctx* ctx, int sort)
{
  CTX_CHECK(ctx); SORT_CHECK(sort);
  Sort& s = ctx->sorts[sort];
  s.n = 0;
  s.prebinned = false;
  XPIC_HIP(hipMemsetAsync(s.d.cell_count, 0, sizeof(int) * (ctx->ncell + 1), ctx->stream));
  XPIC_HIP(hipMemsetAsync(s.d.cell_start, 0, sizeof(int) * (ctx->ncell + 1), ctx->stream));
  return 0;
}

int xpic_sort_fill_synthetic(xpic_ctx* ctx, int sort, int ppc, double vth, uint64_t seed, int regular)
{
  CTX_CHECK(ctx); SORT_CHECK(sort);
  XPIC_CHECK(ppc > 0, "ppc must be positive");
  xpic_load_params lp{};
  lp.ppc = ppc; lp.vth = vth; lp.seed = seed;
  lp.profile = regular ? XPIC_LOAD_REGULAR : XPIC_LOAD_UNIFORM;
  return sort_fill_synthetic(ctx, ctx->sorts[sort], lp);
}

int xpic_sort_load_synthetic(xpic_ctx* ctx, int sort, const xpic_load_params* lp)
{
  CTX_CHECK(ctx); SORT_CHECK(sort);
  XPIC_CHECK(lp && lp->ppc > 0, "ppc must be positive");
  return sort_fill_synthetic(ctx, ctx->sorts[sort], *lp);
}

int xpic_sort_occupancy(xpic_ctx* ctx, int sort, int64_t* out8)
{
  CTX_CHECK(ctx); SORT_CHECK(sort);
  XPIC_CHECK(out8, "null argument");
  return sort_occupancy(ctx, ctx->sorts[sort], out8);
}

int xpic_field_set(xpic_ctx* ctx, int field, const double* v)
{
  CTX_CHECK(ctx); FIELD_CHECK(field);
  return field_import(ctx, ctx->field[field], v);
}

int xpic_field_get(xpic_ctx* ctx, int field, double* v)
{
  CTX_CHECK(ctx); FIELD_CHECK(field);
  return field_export(ctx, ctx->field[field], v);
}

int xpic_sort_current_get(xpic_ctx* ctx, int sort, int which, double* v)
{
  CTX_CHECK(ctx); SORT_CHECK(sort);
  Sort& s = ctx->sorts[sort];
  const double* src = which == XPIC_J ? s.J : (which == XPIC_CURRI ? s.currI : (which == XPIC_CURRJE ? s.currJe : nullptr));
  XPIC_CHECK(src, "this sort does not hold the requested current");
  return field_export(ctx, src, v);
}

int xpic_vec_set(xpic_ctx* ctx, int y, double a) { CTX_CHECK(ctx); FIELD_CHECK(y); return vec_set(ctx, ctx->field[y], a); }
int xpic_vec_axpy(xpic_ctx* ctx, int y, double a, int x)
{
  CTX_CHECK(ctx); FIELD_CHECK(y); FIELD_CHECK(x);
  return vec_axpy(ctx, ctx->field[y], a, ctx->field[x]);
}
int xpic_vec_axpby(xpic_ctx* ctx, int y, double a, double b, int x)
{
  CTX_CHECK(ctx); FIELD_CHECK(y); FIELD_CHECK(x);
  return vec_axpby(ctx, ctx->field[y], a, b, ctx->field[x]);
}
int xpic_vec_dot(xpic_ctx* ctx, int x, int y, double* out)
{
  CTX_CHECK(ctx); FIELD_CHECK(y); FIELD_CHECK(x);
  return vec_dot_host(ctx, ctx->field[x], ctx->field[y], out);
}
int xpic_vec_norm2(xpic_ctx* ctx, int x, double* out)
{
  CTX_CHECK(ctx); FIELD_CHECK(x);
  double d;
  XPIC_CALL(vec_dot_host(ctx, ctx->field[x], ctx->field[x], &d));
  *out = std::sqrt(d);
  return 0;
}

int xpic_rot_apply(xpic_ctx* ctx, int sign, double alpha, int x, int y, int add)
{
  CTX_CHECK(ctx); FIELD_CHECK(y); FIELD_CHECK(x);
  XPIC_CHECK(x != y, "rot_apply cannot work in place");
  XPIC_CALL(halo_fill(ctx, ctx->field[x]));
  return rot_apply(ctx, sign, alpha, ctx->field[x], ctx->field[y], add != 0);
}

int xpic_matM_apply(xpic_ctx* ctx, int x, int y, int add)
{
  CTX_CHECK(ctx); FIELD_CHECK(y); FIELD_CHECK(x);
  XPIC_CHECK(x != y, "matM_apply cannot work in place");
  XPIC_CALL(halo_fill(ctx, ctx->field[x]));
  return matM_apply(ctx, ctx->field[x], ctx->field[y], add != 0);
}

int xpic_matL_apply(xpic_ctx* ctx, int x, int y, int add)
{
  CTX_CHECK(ctx); FIELD_CHECK(y); FIELD_CHECK(x);
  XPIC_CHECK(ctx->matL, "scheme has no matL");
  XPIC_CHECK(x != y, "matL_apply cannot work in place");
  XPIC_CALL(halo_fill(ctx, ctx->field[x]));
  return matL_apply(ctx, ctx->field[x], ctx->field[y], add != 0);
}

int xpic_matA_apply(xpic_ctx* ctx, int x, int y)
{
  CTX_CHECK(ctx); FIELD_CHECK(y); FIELD_CHECK(x);
  XPIC_CHECK(ctx->matL, "scheme has no matL");
  XPIC_CHECK(x != y, "matA_apply cannot work in place");
  XPIC_CALL(halo_fill(ctx, ctx->field[x]));
  return matA_apply(ctx, ctx->field[x], ctx->field[y]);
}

int xpic_matL_get(xpic_ctx* ctx, double* out)
{
  CTX_CHECK(ctx);
  XPIC_CHECK(ctx->matL, "scheme has no matL");
  const GridDev& g = ctx->g;
  const size_t n = matL_doubles(g);
  std::vector<double> tmp(n);
  XPIC_HIP(hipMemcpyAsync(tmp.data(), ctx->matL, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
  XPIC_HIP(hipStreamSynchronize(ctx->stream));
  // device [c1][z][y][k][x] -> boundary [node][c1][k]
  for (int c1 = 0; c1 < 3; ++c1)
    for (int z = 0; z < g.nzl; ++z)
      for (int y = 0; y < g.ny; ++y)
        for (int k = 0; k < kLStencil; ++k)
          for (int x = 0; x < g.nx; ++x) {
            size_t src = (size_t)g.lindex(c1, z + (g.G ? 1 : 0), y, x, k);
            size_t row = (((size_t)z * g.ny + y) * g.nx + x) * 3 + c1;
            out[row * kLStencil + k] = tmp[src];
          }
  return 0;
}

void xpic_lstencil_decode(int c1, int k, int* c2, int* d3)
{
  LEntry e = ldecode(c1, k);
  *c2 = e.c2;
  d3[0] = e.d[0]; d3[1] = e.d[1]; d3[2] = e.d[2];
}

int xpic_ecsim_first_push(xpic_ctx* ctx, int sort)
{
  // the reference's phase boundary, for callers that want it; xpic_step fuses this move into the re-bin
  CTX_CHECK(ctx); SORT_CHECK(sort);
  return sort_move(ctx, ctx->sorts[sort], ctx->g.dt);
}

int xpic_update_cells(xpic_ctx* ctx, int sort, int64_t* count)
{
  CTX_CHECK(ctx); SORT_CHECK(sort);
  XPIC_CALL(sort_rebin(ctx, ctx->sorts[sort], 0.0, true));
  if (count) *count = ctx->sorts[sort].n;
  return 0;
}

int xpic_ecsim_fill_current(xpic_ctx* ctx)
{
  CTX_CHECK(ctx);
  XPIC_CHECK(ctx->scheme != XPIC_BASIC, "basic scheme has no ECSIM current");
  return ecsim_fill_current(ctx);
}

int xpic_ecsim_second_push(xpic_ctx* ctx, int sort)
{
  CTX_CHECK(ctx); SORT_CHECK(sort);
  XPIC_CHECK(ctx->scheme != XPIC_BASIC, "basic scheme has no second_push");
  XPIC_CALL(halo_fill(ctx, ctx->field[XPIC_EP]));
  XPIC_CALL(halo_fill(ctx, ctx->field[XPIC_B]));
  return ecsim_second_push(ctx, ctx->sorts[sort], ctx->field[XPIC_EP], ctx->field[XPIC_B]);
}

int xpic_basic_push(xpic_ctx* ctx, int sort)
{
  CTX_CHECK(ctx); SORT_CHECK(sort);
  XPIC_CHECK(ctx->scheme == XPIC_BASIC, "xpic_basic_push needs the basic scheme");
  Sort& s = ctx->sorts[sort];
  XPIC_CALL(halo_fill(ctx, ctx->field[XPIC_E]));
  XPIC_CALL(halo_fill(ctx, ctx->field[XPIC_B]));
  XPIC_HIP(hipMemsetAsync(s.J, 0, sizeof(double) * ctx->nvec, ctx->stream));
  XPIC_CALL(esirkepov_push(ctx, s, 0, ctx->field[XPIC_E], ctx->field[XPIC_B], s.J, nullptr));
  XPIC_CALL(halo_add(ctx, s.J, 3));
  return vec_axpy(ctx, ctx->field[XPIC_J], 1.0, s.J);
}

int xpic_ecsimcorr_first_push(xpic_ctx* ctx, int sort)
{
  CTX_CHECK(ctx); SORT_CHECK(sort);
  XPIC_CHECK(ctx->scheme == XPIC_ECSIMCORR, "needs the ecsimcorr scheme");
  Sort& s = ctx->sorts[sort];
  XPIC_HIP(hipMemsetAsync(s.currJe, 0, sizeof(double) * ctx->nvec, ctx->stream));
  return esirkepov_push(ctx, s, 1, nullptr, nullptr, s.currJe, nullptr);
}

int xpic_ecsimcorr_second_push(xpic_ctx* ctx, int sort)
{
  CTX_CHECK(ctx); SORT_CHECK(sort);
  XPIC_CHECK(ctx->scheme == XPIC_ECSIMCORR, "needs the ecsimcorr scheme");
  Sort& s = ctx->sorts[sort];
  XPIC_CALL(halo_fill(ctx, ctx->field[XPIC_EP]));
  XPIC_CALL(halo_fill(ctx, ctx->field[XPIC_B]));
  XPIC_CALL(esirkepov_push(ctx, s, 2, ctx->field[XPIC_EP], ctx->field[XPIC_B], s.currJe, &s.pred_w));
  XPIC_CALL(halo_add(ctx, s.currJe, 3));
  return vec_axpy(ctx, ctx->field[XPIC_CURRJE], 1.0, s.currJe);
}

int xpic_ecsimcorr_final_update(xpic_ctx* ctx, int sort)
{
  CTX_CHECK(ctx); SORT_CHECK(sort);
  XPIC_CHECK(ctx->scheme == XPIC_ECSIMCORR, "needs the ecsimcorr scheme");
  return corr_final_update(ctx, ctx->sorts[sort]);
}

int xpic_ecsimcorr_scalars(xpic_ctx* ctx, int sort, double* o)
{
  CTX_CHECK(ctx); SORT_CHECK(sort);
  const Sort& t = ctx->sorts[sort];
  o[0] = t.pred_w; o[1] = t.corr_w; o[2] = t.lambda_dK; o[3] = t.pred_dK; o[4] = t.corr_dK; o[5] = t.energy;
  return 0;
}

int xpic_calculate_energy(xpic_ctx* ctx, int sort, double* energy)
{
  CTX_CHECK(ctx); SORT_CHECK(sort);
  Sort& s = ctx->sorts[sort];
  double o[5];
  XPIC_CALL(kinetic_sums_global(ctx, s, o));
  s.energy = 0.5 * s.par.m * (s.par.n / s.par.Np) * o[3];
  if (energy) *energy = s.energy;
  return 0;
}

int xpic_solve(xpic_ctx* ctx, int op, int rhs, int x, double rtol, double atol, int maxit, int* iterations,
  int* reason, double* rnorm)
{
  CTX_CHECK(ctx); FIELD_CHECK(rhs); FIELD_CHECK(x);
  XPIC_CHECK(ctx->kry_V, "scheme has no Krylov workspace");
  XPIC_CHECK(rhs != x, "rhs and x must differ");
  XPIC_CHECK(op >= 0 && op <= 2, "unknown solve op");
  int its = 0, rs = 0;
  double rn = 0;
  int rc = solve(ctx, op, ctx->field[rhs], ctx->field[x], rtol, atol, maxit, &its, &rs, &rn);
  if (iterations) *iterations = its;
  if (reason) *reason = rs;
  if (rnorm) *rnorm = rn;
  return rc;
}

int xpic_set_preconditioner(xpic_ctx* ctx, int kind, int degree)
{
  CTX_CHECK(ctx);
  XPIC_CHECK(kind >= 0 && kind <= 5, "unknown preconditioner kind");
  ctx->precond = kind;
  ctx->cheb_degree_user = degree > 0 ? (degree > 64 ? 64 : degree) : 0;
  if (degree > 0) ctx->cheb_degree = ctx->cheb_degree_M = ctx->cheb_degree_user;
  else { ctx->cheb_degree = ctx->cheb_degree_auto; ctx->cheb_degree_M = ctx->cheb_degree_M_auto; } // back to the automatic choice
  return ensure_flexible_workspace(ctx);
}

int xpic_set_fused_rebin(xpic_ctx* ctx, int on)
{
  CTX_CHECK(ctx);
  XPIC_CHECK(on >= 0 && on <= 2, "xpic_set_fused_rebin: 0 scatter first, 1 the assembly writes the sorted copy, 2 the second push does");
  ctx->fused_rebin = on;
  return 0;
}

int xpic_set_fill_kernel(xpic_ctx* ctx, int kind)
{
  CTX_CHECK(ctx);
  XPIC_CHECK(kind == 0 || kind == 1, "unknown assembly kernel (0 classic, 1 warp-specialised)");
  ctx->fill_kernel = kind;
  return 0;
}

int xpic_debug_set(xpic_ctx* ctx, int what, int64_t value)
{
  CTX_CHECK(ctx);
  switch (what) {
    case XPIC_DEBUG_GATHER_WINDOW:
      XPIC_CHECK(value >= 1 && value <= (1 << 28), "gather window: 1 .. 2^28 slots");
      ctx->gather_window = (int)value;
      return 0;
    case XPIC_DEBUG_PENCIL_LIMIT:
      XPIC_CHECK(value >= 1 && value <= (1 << 29), "pencil limit: 1 .. 2^29 particles");
      ctx->pencil_limit = (int)value;
      return 0;
    case XPIC_DEBUG_SURROGATE_SCALE:
      ctx->debug_surrogate_scale = (double)value / 1000.0;
      return 0;
    default:
      XPIC_CHECK(false, "xpic_debug_set: unknown knob");
  }
  return 0;
}

int xpic_get_fill_variant(xpic_ctx* ctx, int* out3)
{
  CTX_CHECK(ctx);
  XPIC_CHECK(out3, "null argument");
  ecsim_fill_variant(ctx, &out3[0], &out3[1], &out3[2]);
  return 0;
}

int xpic_set_overlap(xpic_ctx* ctx, int on)
{
  CTX_CHECK(ctx);
  ctx->overlap = (on & 1) != 0;
  ctx->overlap_lrows = (on & 2) != 0;
  ctx->peer_copy = (on & 4) != 0; // takes effect once xpic_comm_peer_import has mapped the neighbours' buffers
  ctx->overlap_explicit = true;
  return 0;
}

int xpic_set_tolerances(xpic_ctx* ctx, double rtol, double atol, int maxit)
{
  CTX_CHECK(ctx);
  ctx->rtol = rtol; ctx->atol = atol; ctx->maxit = maxit;
  return 0;
}

int xpic_step(xpic_ctx* ctx, int* ksp_iterations)
{
  CTX_CHECK(ctx);
  int its = 0;
  int rc;
  switch (ctx->scheme) {
    case XPIC_BASIC: rc = step_basic(ctx); break;
    case XPIC_ECSIM: rc = step_ecsim(ctx, &its); break;
    default: rc = step_ecsimcorr(ctx, &its); break;
  }
  if (ksp_iterations) *ksp_iterations = its;
  return rc;
}

int xpic_energy(xpic_ctx* ctx, double* out)
{
  CTX_CHECK(ctx);
  const GridDev& g = ctx->g;
  const double g3 = (double)g.nx * g.ny * g.nzg;
  for (int f = 0; f < 2; ++f) {
    double sq, mean[3];
    XPIC_CALL(field_stats_host(ctx, ctx->field[f == 0 ? XPIC_E : XPIC_B], &sq, mean));
    const double nrm = std::sqrt(sq);
    const double w = 0.5 * (nrm * nrm);
    out[f] = w;
    out[2 + f] = std::sqrt((w - 0.5 * (mean[0] * mean[0] + mean[1] * mean[1] + mean[2] * mean[2]) / g3) / g3);
  }
  for (size_t i = 0; i < ctx->sorts.size(); ++i) {
    Sort& s = ctx->sorts[i];
    double o[5];
    XPIC_CALL(kinetic_sums_global(ctx, s, o));
    const double frac = 0.5 * s.par.m * (s.par.n / (double)s.par.Np);
    double K = frac * o[3], sK = 0;
    if (o[4] == 0) K = 0;
    else {
      const double sv = o[3] - (o[0] * o[0] + o[1] * o[1] + o[2] * o[2]) / o[4];
      sK = frac * std::sqrt(std::fabs(sv) / o[4]);
    }
    out[4 + 2 * i] = K;
    out[5 + 2 * i] = sK;
  }
  return 0;
}

int xpic_momentum(xpic_ctx* ctx, double* out) // MomentumConservation::calculate, momentum_conservation.cpp:77-131
{
  CTX_CHECK(ctx);
  XPIC_CHECK(out, "null argument");
  XPIC_CALL(halo_fill(ctx, ctx->field[XPIC_E])); // DMGlobalToLocal(da, E, INSERT_VALUES, El) :82
  for (size_t i = 0; i < ctx->sorts.size(); ++i)
    XPIC_CALL(momentum_sums_global(ctx, ctx->sorts[i], ctx->field[XPIC_E], out + 6 * i));
  return 0;
}

static double* sort_current(xpic_ctx* ctx, Sort& s)
{
  return ctx->scheme == XPIC_BASIC ? s.J : (ctx->scheme == XPIC_ECSIM ? s.currI : s.currJe);
}

int xpic_charge_density(xpic_ctx* ctx, int sort, double* rho_zyx)
{
  CTX_CHECK(ctx); SORT_CHECK(sort);
  double* tmp = ctx->field[XPIC_W2];
  XPIC_CALL(charge_density(ctx, ctx->sorts[sort], tmp));
  std::vector<double> v3((size_t)ctx->g.nown * 3);
  XPIC_CALL(field_export(ctx, tmp, v3.data()));
  for (long i = 0; i < ctx->g.nown; ++i) rho_zyx[i] = v3[3 * i];
  return 0;
}

int xpic_moment_density(xpic_ctx* ctx, int sort, double* out_zyx)
{
  CTX_CHECK(ctx); SORT_CHECK(sort);
  double* tmp = ctx->field[XPIC_W2];
  XPIC_CALL(moment_density(ctx, ctx->sorts[sort], tmp));
  std::vector<double> v3((size_t)ctx->g.nown * 3);
  XPIC_CALL(field_export(ctx, tmp, v3.data()));
  for (long i = 0; i < ctx->g.nown; ++i) out_zyx[i] = v3[3 * i];
  return 0;
}

int xpic_charge_collect(xpic_ctx* ctx) // ChargeConservation::initialize, charge_conservation.cpp:117-123
{
  CTX_CHECK(ctx);
  for (auto& s : ctx->sorts) {
    if (!s.rho) XPIC_HIP(hipMalloc(&s.rho, sizeof(double) * ctx->nvec));
    XPIC_CALL(charge_density(ctx, s, s.rho));
  }
  return 0;
}

int xpic_charge_columns(xpic_ctx* ctx, double* out) // ChargeConservation::add_columns, :125-171
{
  CTX_CHECK(ctx);
  double* sum = ctx->field[XPIC_W0];
  double* diff = ctx->field[XPIC_W1];
  double* fresh = ctx->field[XPIC_W2];
  XPIC_CALL(vec_set(ctx, sum, 0.0));
  size_t i = 0;
  for (; i < ctx->sorts.size(); ++i) {
    Sort& s = ctx->sorts[i];
    XPIC_CHECK(s.rho, "xpic_charge_collect must run first");
    XPIC_CALL(charge_density(ctx, s, fresh));
    XPIC_CALL(vec_waxpby(ctx, diff, 1.0 / ctx->g.dt, fresh, -1.0 / ctx->g.dt, s.rho)); // (rho_new - rho_old) / dt
    XPIC_CALL(vec_copy(ctx, s.rho, fresh));
    XPIC_CALL(vec_axpy(ctx, sum, 1.0, diff));
    XPIC_CALL(div_neg_add(ctx, sort_current(ctx, s), diff));
    XPIC_CALL(scalar_norm12_host(ctx, diff, out + 2 * i));
  }
  double* total = ctx->scheme == XPIC_BASIC ? ctx->field[XPIC_J] : (ctx->scheme == XPIC_ECSIM ? ctx->field[XPIC_CURRI] : ctx->field[XPIC_CURRJE]);
  XPIC_CALL(div_neg_add(ctx, total, sum));
  XPIC_CALL(scalar_norm12_host(ctx, sum, out + 2 * i));
  return 0;
}

int xpic_profile_enable(xpic_ctx* ctx, int on) { CTX_CHECK(ctx); ctx->profiling = on != 0; return 0; }

int xpic_profile_reset(xpic_ctx* ctx)
{
  CTX_CHECK(ctx);
  XPIC_CALL(resolve_profile(ctx));
  for (auto& kv : ctx->prof) { kv.second.launches = 0; kv.second.total_ms = 0; }
  return 0;
}

int xpic_profile_get(xpic_ctx* ctx, const char* name, int64_t* launches, double* total_ms)
{
  CTX_CHECK(ctx);
  XPIC_CALL(resolve_profile(ctx));
  auto it = ctx->prof.find(name);
  if (it == ctx->prof.end()) { *launches = 0; *total_ms = 0; return 0; }
  *launches = it->second.launches;
  *total_ms = it->second.total_ms;
  return 0;
}

__global__ void __launch_bounds__(256) k_copy16(const double2* __restrict__ in, double2* __restrict__ out, long n)
{
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = in[i];
}

__global__ void __launch_bounds__(256) k_copy8(const double* __restrict__ in, double* __restrict__ out, long n)
{
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = in[i];
}

int xpic_probe_copy_bandwidth(xpic_ctx* ctx, int64_t bytes, int reps, double* bytes_per_s)
{
  CTX_CHECK(ctx);
  double2 *a = nullptr, *b = nullptr;
  const long n = bytes / 16;
  XPIC_HIP(hipMalloc(&a, n * 16));
  XPIC_HIP(hipMalloc(&b, n * 16));
  XPIC_HIP(hipMemsetAsync(a, 1, n * 16, ctx->stream));
  hipEvent_t e0 = get_event(ctx), e1 = get_event(ctx);
  hipLaunchKernelGGL(k_copy16, dim3(2048), dim3(256), 0, ctx->stream, a, b, n);
  // one 8-byte-per-lane copy of the same buffers: the access width of the SpMV's loads, launched so that the
  // FETCH_SIZE / WRITE_SIZE counters can be calibrated on a known byte count (MI355X_MICROARCH.md, HBM section)
  hipLaunchKernelGGL(k_copy8, dim3(2048), dim3(256), 0, ctx->stream, (const double*)a, (double*)b, 2 * n);
  XPIC_HIP(hipEventRecord(e0, ctx->stream));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_copy16, dim3(2048), dim3(256), 0, ctx->stream, a, b, n);
  XPIC_HIP(hipEventRecord(e1, ctx->stream));
  XPIC_HIP(hipStreamSynchronize(ctx->stream));
  float ms = 0;
  XPIC_HIP(hipEventElapsedTime(&ms, e0, e1));
  *bytes_per_s = 2.0 * n * 16 * reps / (ms * 1e-3);
  ctx->event_pool.push_back(e0);
  ctx->event_pool.push_back(e1);
  XPIC_HIP(hipFree(a));
  XPIC_HIP(hipFree(b));
  return 0;
}

}  // extern "C"
