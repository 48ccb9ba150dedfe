/*
 * xpic_hip.h -- C ABI of the MI355X (gfx950) implementation of xpic's per-timestep hot path.
 *
 * The reference (vakurshakov/xpic) has no FFI: its hot path is reached through the C++ virtual
 * `interfaces::Simulation::timestep_implementation()` (src/interfaces/simulation.h:71-72) and the
 * public members of `interfaces::Simulation` / `interfaces::Particles`.  This header is the boundary a
 * `impls/` backend binds UNDERNEATH those classes: every entry point names the reference function it
 * replaces (file:line relative to the reference checkout).  INTEGRATION.md shows the subclass a
 * maintainer would add on the reference side.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error (slots into `PetscCall`); the message
 *     of the last error of the calling thread is returned by xpic_last_error().
 *   - field vectors cross the boundary in the reference's DMDA layout: double[nz][ny][nx][3]
 *     (x fastest, 3 components interleaved; src/utils/vector3.h:233, src/utils/world.h:35-43).
 *   - particles cross the boundary as `struct Point` records: double[6] = {x,y,z,px,py,pz}
 *     (src/interfaces/point.h:7-35).  Inside, both are re-laid out.
 *   - pointers are HOST pointers unless the parameter is called `dptr` (device pointer).
 *   - all boundaries are periodic (every BASELINE config); anything else is rejected at create.
 */
#ifndef XPIC_HIP_H
#define XPIC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct xpic_ctx xpic_ctx;

/* Geometry: globals dx,dy,dz,dt,geom_n* (src/constants.h:10-28) + World (src/utils/world.h:11-61). */
typedef struct xpic_geometry {
  int32_t n[3];     /* global cells geom_nx, geom_ny, geom_nz */
  double d[3];      /* dx, dy, dz */
  double dt;
  int32_t periodic[3]; /* must be {1,1,1}: DM_BOUNDARY_PERIODIC */
  int32_t rank;     /* z-slab index of this context (DMDA da_processors_z) */
  int32_t nranks;   /* number of z-slabs */
  int32_t device;   /* HIP device ordinal */
  int32_t self_ring; /* nranks == 1 only: keep the ghost planes and run the exchange layer with the slab as its own
                        lower and upper neighbour (exercises the RCCL transport on one GPU); 0 in production */
} xpic_geometry;

/* SortParameters (src/interfaces/sort_parameters.h:7-19) */
typedef struct xpic_sort_params {
  int32_t Np;
  double n, q, m;
} xpic_sort_params;

/* named global vectors of interfaces::Simulation / ecsim::Simulation / ecsimcorr::Simulation
 * (src/interfaces/simulation.h:33-48, src/impls/ecsim/simulation.h:27-28, ecsimcorr/simulation.h:17-18) */
enum xpic_field {
  XPIC_E = 0, XPIC_B = 1, XPIC_B0 = 2, XPIC_J = 3, XPIC_EP = 4, XPIC_EC = 5, XPIC_CURRI = 6,
  XPIC_CURRJE = 7, XPIC_W0 = 8, XPIC_W1 = 9, XPIC_W2 = 10, XPIC_NFIELDS = 11
};

enum xpic_scheme { XPIC_BASIC = 0, XPIC_ECSIM = 1, XPIC_ECSIMCORR = 2 };

/* operator / method selector of xpic_solve */
enum xpic_solve_op {
  XPIC_OP_MATA_GMRES = 0, /* (matL + matM) x = b, GMRES(30): KSP "predict" (ecsim/simulation.cpp:197-201,266) */
  XPIC_OP_MATM_GMRES = 1, /* matM x = b, GMRES(30): KSP "correct" (ecsimcorr/simulation.cpp:133) */
  XPIC_OP_MATM_CG = 2     /* matM x = b, CG (matM is SPD on periodic boundaries) */
};

#define XPIC_LSTENCIL 123 /* couplings per matL row: 27 same-component + 48 + 48 */

const char* xpic_last_error(void);
/* XPIC_VERSION, with XPIC_VERSION_EXPERIMENT_BIT set when any object of the library was built with -DXPIC_EXPERIMENT
 * (ablation switches and in-kernel timers of the kernels; some produce wrong physics by design): refuse such a library
 * for production runs. */
#define XPIC_VERSION 2
#define XPIC_VERSION_EXPERIMENT_BIT 0x40000000
int xpic_version(void);

/* World::initialize + Simulation::initialize_implementation (world.cpp:11-48; ecsim/simulation.cpp:122-143,
 * 517-567; basic/simulation.cpp:8-28): allocates E,B,B0,J,(Ep,Ec,currI,currJe), operators and solver state. */
int xpic_create(const xpic_geometry* geom, int scheme, xpic_ctx** out);
int xpic_destroy(xpic_ctx* ctx); /* Simulation::finalize (ecsim/simulation.cpp:569-590) */
int xpic_synchronize(xpic_ctx* ctx);

/* init_particles -> PartSpec(sim, SortParameters) (src/interfaces/simulation.tpp:43); capacity in particles */
int xpic_add_sort(xpic_ctx* ctx, const xpic_sort_params* p, int64_t capacity, int* sort_out);
/* Particles::add_particle (src/interfaces/particles.cpp:47-67) for n Points: binned by FLOOR_STEP, points
 * outside the local box are dropped; *added = number kept. Appends to what the sort already holds. */
int xpic_sort_add_particles(xpic_ctx* ctx, int sort, int64_t n, const double* points6, int64_t* added);
int xpic_sort_count(xpic_ctx* ctx, int sort, int64_t* count);
/* storage read-back in cell order: points6[count][6], cell_of[count] = local cell index g (world.s_g) */
int xpic_sort_get_particles(xpic_ctx* ctx, int sort, double* points6, int32_t* cell_of);
int xpic_sort_clear(xpic_ctx* ctx, int sort);
/* synthetic plasma generated on the device (bench/smoke only): ppc * (local cells) particles, Maxwellian velocities of
 * thermal spread vth (then v /= sqrt(1+v^2), "tov").  regular == 0: positions uniform over the slab, i.e. Poisson
 * occupancy of the cells, the load CoordinateInBox + SetParticles produce (src/utils/particles_load.cpp:11-18,
 * src/commands/set_particles.cpp:19-43; the device RNG is its own, not mt19937); regular != 0: exactly ppc particles
 * in every cell.  Collective over the z-slabs like xpic_update_cells. */
int xpic_sort_fill_synthetic(xpic_ctx* ctx, int sort, int ppc, double vth, uint64_t seed, int regular);
/* The same with a drift and a density profile.  drift = SortParameters::px, py, pz as MaxwellianMomentum adds them
 * (src/utils/particles_load.cpp:57-76: to the thermal momentum, before `tov`; in units of m c with m = 1).  The
 * reference's JSON surface never reads px / py / pz (src/interfaces/simulation.tpp:24-41), so a drifting Maxwellian -- the
 * two counter-streaming beams of BASELINE configs[1] -- is an EXTENSION of this build's loaders, not reference behaviour.
 * profile: XPIC_LOAD_UNIFORM / XPIC_LOAD_REGULAR as above; XPIC_LOAD_GRADIENT: density falling linearly along x from
 * profile_param[0] : 1 at x = 0 to 1 at x = Lx (same particle total); XPIC_LOAD_BLOB: the fraction profile_param[0] of
 * the particles in a Gaussian clump of sigma = profile_param[1] cells at the centre of the slab, the rest uniform (cells
 * of many times the mean occupancy: the bucket / third-pass / colour-balance fall-backs of the particle kernels). */
#define XPIC_LOAD_UNIFORM 0
#define XPIC_LOAD_REGULAR 1
#define XPIC_LOAD_GRADIENT 2
#define XPIC_LOAD_BLOB 3
typedef struct xpic_load_params {
  int32_t ppc;
  int32_t profile;
  double vth;
  double drift[3];
  double profile_param[4];
  uint64_t seed;
} xpic_load_params;
int xpic_sort_load_synthetic(xpic_ctx* ctx, int sort, const xpic_load_params* params);
/* Occupancy of the local cells (`storage[g].size()`, src/interfaces/particles.h:32, any distribution): out8 = {largest
 * cell, cells of more than 64 particles (a second staging pass in the assembly), of more than 128 (a third), of more than
 * a bucket of the deferred scatter holds (the step then takes the index pass), largest and smallest population of an
 * x-pencil (one workgroup each in the particle kernels: the balance of a colour launch), empty cells, bucket capacity}. */
int xpic_sort_occupancy(xpic_ctx* ctx, int sort, int64_t* out8);

/* Vec access (DMDAVecGetArray / VecGetArray): copies in/out in the [z][y][x][3] layout, local slab */
int xpic_field_set(xpic_ctx* ctx, int field, const double* v);
int xpic_field_get(xpic_ctx* ctx, int field, double* v);
int xpic_sort_current_get(xpic_ctx* ctx, int sort, int which /* XPIC_J | XPIC_CURRI | XPIC_CURRJE */, double* v);

/* ---- BLAS-1 on named vectors (VecSet/VecAXPY/VecAXPBY/VecDot/VecNorm; K14) */
int xpic_vec_set(xpic_ctx* ctx, int y, double alpha);
int xpic_vec_axpy(xpic_ctx* ctx, int y, double alpha, int x);                 /* y += alpha x */
int xpic_vec_axpby(xpic_ctx* ctx, int y, double alpha, double beta, int x);  /* y = alpha x + beta y */
int xpic_vec_dot(xpic_ctx* ctx, int x, int y, double* out);
int xpic_vec_norm2(xpic_ctx* ctx, int x, double* out);

/* ---- operators (K11-K13) on named vectors; `add` != 0 gives MatMultAdd (y += ...) */
/* Rotor (src/utils/operators.cpp:155-215): y (+)= alpha * rot(sign) x; sign +1 = rotE, -1 = rotB */
int xpic_rot_apply(xpic_ctx* ctx, int sign, double alpha, int x, int y, int add);
/* matM = 2 I + 0.5 dt^2 rotB rotE (src/impls/ecsim/simulation.cpp:544-551) */
int xpic_matM_apply(xpic_ctx* ctx, int x, int y, int add);
/* matL as filled by xpic_ecsim_fill_current (MatMultAdd(matL,...) ecsimcorr/simulation.cpp:78) */
int xpic_matL_apply(xpic_ctx* ctx, int x, int y, int add);
/* matA = matL + matM (ecsim/simulation.cpp:197-198) */
int xpic_matA_apply(xpic_ctx* ctx, int x, int y);
/* matL read-back as double[3N][XPIC_LSTENCIL], row = ((z*ny+y)*nx+x)*3+c, k as xpic_lstencil_decode */
int xpic_matL_get(xpic_ctx* ctx, double* out);
void xpic_lstencil_decode(int c1, int k, int* c2, int* d3);

/* ---- per-phase entry points */
/* ecsim::Particles::first_push (src/impls/ecsim/particles.cpp:21-31): r += dt*p */
int xpic_ecsim_first_push(xpic_ctx* ctx, int sort);
/* Particles::update_cells_seq / correct_coordinates (src/interfaces/particles.cpp:79-116,329-339):
 * periodic wrap, re-bin by FLOOR_STEP, drop what falls outside; *count = particles left */
int xpic_update_cells(xpic_ctx* ctx, int sort, int64_t* count);
/* ecsim::Simulation::fill_ecsim_current + Particles::fill_ecsim_current/decompose_ecsim_current
 * (src/impls/ecsim/simulation.cpp:336-368,471-484; particles.cpp:33-173): zeroes then fills currI
 * (per sort and total) and matL from all sorts, gathering B */
int xpic_ecsim_fill_current(xpic_ctx* ctx);
/* ecsim::Particles::second_push (src/impls/ecsim/particles.cpp:175-192): CIC gather of Ep and B, update_vEB(dt) */
int xpic_ecsim_second_push(xpic_ctx* ctx, int sort);
/* basic::Particles::push (src/impls/basic/particles.cpp:17-53): half move, 2nd-order gather, Boris, half move,
 * Esirkepov into the sort's J and the simulation's J */
int xpic_basic_push(xpic_ctx* ctx, int sort);
/* ecsimcorr::Particles::{first_push, second_push, final_update, calculate_energy}
 * (src/impls/ecsimcorr/particles.cpp:27-50, 52-91, 93-126, 134-150) */
int xpic_ecsimcorr_first_push(xpic_ctx* ctx, int sort);
int xpic_ecsimcorr_second_push(xpic_ctx* ctx, int sort);
int xpic_ecsimcorr_final_update(xpic_ctx* ctx, int sort);
int xpic_calculate_energy(xpic_ctx* ctx, int sort, double* energy);
/* pred_w, corr_w, lambda_dK, pred_dK, corr_dK, energy (ecsimcorr/particles.h:44-49) */
int xpic_ecsimcorr_scalars(xpic_ctx* ctx, int sort, double* out6);

/* KSPSolve (src/impls/ecsim/simulation.cpp:266): x0 = 0, preconditioner as set by xpic_set_preconditioner (both GMRES
 * operators; CG is unpreconditioned), converged when the residual ||b - A x|| <= max(rtol ||b||, atol) (the cheap
 * preconditioned XPIC_OP_MATM_GMRES is run to 1e-2 of that).  The norm tested (and returned in *rnorm) is GMRES's
 * RECURRENCE value of the unpreconditioned residual -- right / flexible preconditioning keeps it the residual of A x = b
 * itself, not of a preconditioned system -- as in PETSc's KSPGMRES; no explicit b - A x is formed at exit (one more apply
 * per solve).  It equals the true residual up to the orthogonality of the basis: iterations whose entering residual is
 * above 1e-6 ||b|| take |w - sum h_i v_i| from w.w - sum h_i^2 (one reduction per iteration), later ones and every solve
 * with a tolerance below 1e-8 ||b|| take the norm explicitly (krylov.hip).  *iterations >= 0; *reason > 0 converged,
 * < 0 diverged (maxit).
 * A non-converged solve RETURNS NON-ZERO, like KSPSetErrorIfNotConverged (:562). */
int xpic_solve(xpic_ctx* ctx, int op, int rhs, int x, double rtol, double atol, int maxit, int* iterations,
  int* reason, double* rnorm);
/* KSPSetTolerances used by the step drivers (src/impls/ecsim/simulation.h:15-18: 1e-7,1e-7,100) */
int xpic_set_tolerances(xpic_ctx* ctx, double rtol, double atol, int maxit);

/* PCSetType for the "predict" and "correct" KSPs.  The reference runs PETSc's default ILU(0) (not part of its tree, not a GPU
 * algorithm); here: kind 0 = none, kind 1 / 2 = a fixed Chebyshev polynomial in matM applied from the right
 * (matM = 2 I + 0.5 dt^2 rotB rotE dominates matA and its spectral interval is known in closed form), its work vectors
 * kept in fp32 (kind 1) or fp64 (kind 2); kind 3 = the polynomial in matM + <matL>, the translation average of the assembled
 * mass matrix as one constant-coefficient 123-point stencil (fp32), for the predict solve (the correct solve on matM keeps
 * kind 1); kind 4 = kind 3 with the rows of <matL> scaled by the local density (the ratio of the row's own diagonal
 * entry of matL to the average's): three times the convergence rate per iteration at 64 particles per cell, at twice the
 * polynomial's cost (at the reference's tolerance both need 4 iterations on the uniform 256^3 box: kind 3 is the faster one
 * there; with the density falling 4 : 1 across the box kind 3 needs 6 and kind 4 four); kind 5 (default) = kind 3 or kind 4,
 * chosen per solve from the relative spread of matL's diagonal (above 0.2: kind 4; a uniform Poisson load of 64 per cell
 * has 0.1).  The GMRES around it is the flexible variant (x = x0 + sum y_j P v_j with the
 * P v_j stored): the result does not depend on how exactly P is applied, only the iteration count could.
 * degree <= 0 returns to the automatic choice.  The stopping rule of xpic_solve is unchanged (the residual of A x = b).
 * Kinds 3, 4 and 5 check their surrogate per solve: where 2 + the Gershgorin lower bound of <matL> (times the largest density
 * ratio) is positive the polynomial's interval is proven; otherwise the surrogate runs on probation -- an iteration that does
 * not halve the residual, or is not finite, ends it and the solve goes on with kind 1. */
int xpic_set_preconditioner(xpic_ctx* ctx, int kind, int degree);
/* The mass-matrix assembly (fill_ecsim_current) has two bodies: kind 0 (default) the classic 4-wave kernel (all grids);
 * kind 1 the warp-specialised kernel (one 16-wave workgroup per CU: a producer wave per SIMD runs the per-particle algebra,
 * two consumer waves the matrix-core accumulation, a flusher the window's read-modify-write), available where nx is a
 * multiple of 4 and no extent is below 3, measured slower.  Same matrix up to the summation order of a
 * cell's neighbours.  xpic_get_fill_variant: out3 = {power-of-two spacings, full-chunk body, warp-specialised body} as the
 * next assembly of this context will run. */
int xpic_set_fill_kernel(xpic_ctx* ctx, int kind);
/* update_cells (src/interfaces/particles.cpp:79-116) re-bins the particles in two passes (keys, then an out-of-place
 * scatter).  on = 1 (default) leaves the scatter to the next kernel that reads every particle anyway: in the ecsim step
 * (ecsim/simulation.cpp:174-189) and for the first re-binning of the ecsimcorr step that is the mass-matrix assembly, in the
 * basic step (basic/simulation.cpp:45-72) the next step's push; it gathers the records through source indices, applies the
 * move and the periodic wrap and writes the sorted copy on its way.  on = 2 (ecsim): the assembly only reads through the
 * index and second_push -- which is bound by memory anyway -- writes the sorted copy with the new velocities.  Same
 * particles, same cells, same arithmetic as on = 0 (scatter first); any other reader of a sort (diagnostics, downloads,
 * the phase entry points) resolves a pending deferral by the plain scatter.  On z-slabs the assembly's form is used as well
 * (records received from the neighbours are gathered out of the receive buffer); the basic step and on = 2 defer on a single
 * slab only. */
int xpic_set_fused_rebin(xpic_ctx* ctx, int on);
int xpic_get_fill_variant(xpic_ctx* ctx, int* out3);
/* Test hooks for two size limits of the gathering assembly that no test-sized box reaches on its own (results are the same
 * for every value; only the code path changes).  XPIC_DEBUG_GATHER_WINDOW: old-order records within `value` slots of an
 * x-pencil's first slot are fetched by 32-bit offsets, the others -- at 256^3 x 64 what crossed the periodic z boundary --
 * by 64-bit addresses (default and maximum 2^28).  XPIC_DEBUG_PENCIL_LIMIT: a sort with an x-pencil of `value` particles or
 * more scatters first instead of deferring (default and maximum 2^29, the reach of the sorted copy's 32-bit offsets). */
#define XPIC_DEBUG_GATHER_WINDOW 0
#define XPIC_DEBUG_PENCIL_LIMIT 1
/* XPIC_DEBUG_SURROGATE_SCALE: the preconditioner's surrogate is built from `value` / 1000 times <matL> (1000 = as it is): a
 * deliberately wrong surrogate, to see the probation of an unproven one end in the fall-back (results are unchanged: the
 * stopping rule is the true residual). */
#define XPIC_DEBUG_SURROGATE_SCALE 2
int xpic_debug_set(xpic_ctx* ctx, int what, int64_t value);
/* MatMult on a z-slab with neighbours: on = 1 posts the ghost exchange of the operand (VecScatterBegin), applies
 * the rows of the interior planes meanwhile and the rows of the boundary planes after it (VecScatterEnd), as PETSc's
 * MPIAIJ MatMult does (the reference's KSPSolve, src/impls/ecsim/simulation.cpp:266); on = 0 exchanges first. Same result.
 * Bit 1 of `on` (on = 3) also posts the assembly's ghost-row exchange of matL behind the boundary colours, beside the
 * interior colour launches (off by default: measured slower).
 * Default: 0 -- on the one-GPU self-ring (the only hardware these paths have run on) both overlaps cost more than the
 * exchanges they hide, and over RCCL with more than one rank the second-stream path has not yet run
 * on two distinct GPUs.  XPIC_RCCL_OVERLAP=1 turns bit 0 on at xpic_comm_init_rccl.
 * bit 2: the matL ghost rows by copy engine (see xpic_comm_peer_import below). */
int xpic_set_overlap(xpic_ctx* ctx, int on);

/* timestep_implementation of the context's scheme (basic/simulation.cpp:30-43, ecsim/simulation.cpp:145-155,
 * ecsimcorr/simulation.cpp:21-32); *ksp_iterations = Krylov iterations spent in this step */
int xpic_step(xpic_ctx* ctx, int* ksp_iterations);

/* Energy::calculate_field/calculate_kinetic (src/diagnostics/energy.cpp:43-108):
 * out = {wE, wB, sE, sB, wK_0, sK_0, wK_1, sK_1, ...} */
int xpic_energy(xpic_ctx* ctx, double* out);

/* MomentumConservation::calculate (src/diagnostics/momentum_conservation.cpp:77-131): per sort
 * out[6 i ..] = {Px, Py, Pz, QEx, QEy, QEz}, P = sum (m/Np) v ns and QE = sum (q/Np) E Es over the particle's
 * 2nd-order shape nodes */
int xpic_momentum(xpic_ctx* ctx, double* out);

/* ParticlesChargeDensity::collect of one sort (src/diagnostics/charge_conservation.cpp:67-97) -> rho[z][y][x] */
int xpic_charge_density(xpic_ctx* ctx, int sort, double* rho_zyx);
/* DistributionMoment::collect with moment "density" (src/diagnostics/distribution_moment.cpp:125-216): cell-centred
 * first-order deposit of n/Np -> out[z][y][x].  Uses the scratch vector XPIC_W2. */
int xpic_moment_density(xpic_ctx* ctx, int sort, double* out_zyx);
/* ChargeConservation (charge_conservation.cpp:117-171): xpic_charge_collect() = initialize(); then once per step
 * xpic_charge_columns(): out = {N1dQ_0, N2dQ_0, ..., N1dQ_tot, N2dQ_tot} of (rho_new - rho_old)/dt + div(-) J.
 * Uses the scratch vectors XPIC_W0..W2. */
int xpic_charge_collect(xpic_ctx* ctx);
int xpic_charge_columns(xpic_ctx* ctx, double* out);

/* ---- inner kernels of the `eccapfim` scheme, batch form over n path segments r0 -> rn (host arrays,
 * 3 doubles per point).  The scheme's outer loops are not part of this library.
 * cell_traversal (src/impls/eccapfim/cell_traversal.cpp:3-77): for every segment the points start, face crossings of the
 * node-centred cells, end -> pts[(q * max_pts + i) * 3 ..], counts[q] (may exceed max_pts: then the tail is dropped). */
int xpic_cell_traversal(xpic_ctx* ctx, int64_t n, const double* end3, const double* start3, int max_pts, double* pts,
  int* counts);
/* ImplicitEsirkepov::interpolate (src/algorithms/implicit_esirkepov.cpp:60-90): E_p with the segment's 54-weight shape,
 * B_p with Shape(midpoint) + SimpleInterpolation, from the context's XPIC_E / XPIC_B. */
int xpic_implicit_esirkepov_interpolate(xpic_ctx* ctx, int64_t n, const double* rn3, const double* r03, double* Ep3,
  double* Bp3);
/* ImplicitEsirkepov::decompose (:92-117): field += alpha[q] * v[q] * shape(q), ghost contributions folded like
 * DMLocalToGlobal(ADD_VALUES).  Uses the scratch vector XPIC_W2. */
int xpic_implicit_esirkepov_decompose(xpic_ctx* ctx, int64_t n, const double* alpha, const double* v3, const double* rn3,
  const double* r03, int field);

/* ---- z-slab decomposition (DMDA da_processors_z = nranks; src/utils/world.cpp:36-38).  A context created with
 * nranks > 1 owns planes [rank*nz/nranks, (rank+1)*nz/nranks) and must be given a communicator before any
 * call that moves data between slabs (steps, solves, operator applies, re-binning, energy): those calls are
 * collective over the ranks.  Replaces update_cells_mpi (src/interfaces/particles.cpp:118-248), DMGlobalToLocal /
 * DMLocalToGlobal(ADD) and the all-reduces inside VecDot/VecNorm/KSPSolve. */
/* RCCL over xGMI: rank 0 creates the id, every rank (one process per GPU) passes the same 128 bytes */
int xpic_comm_rccl_unique_id(void* id128);
int xpic_comm_init_rccl(xpic_ctx* ctx, const void* id128);
/* host-staged transport supplied by the caller (tests: torch.distributed/gloo).  Ring semantics: send `down` to
 * rank-1 and `up` to rank+1; receive the upper neighbour's `down` message into from_up and the lower neighbour's
 * `up` message into from_down (with 2 ranks both neighbours are the same peer: messages are matched in this order) */
typedef struct xpic_comm_callbacks {
  void* user;
  int (*sendrecv)(void* user, const void* down, size_t ndown, const void* up, size_t nup, void* from_up,
    size_t nfrom_up, void* from_down, size_t nfrom_down);
  int (*allreduce_sum)(void* user, double* buf, int n);
} xpic_comm_callbacks;
int xpic_comm_init_callbacks(xpic_ctx* ctx, const xpic_comm_callbacks* cb);
/* number of ranks of the attached communicator: ncclCommCount for RCCL, the z-slab count for callbacks, 1 without one
 * (MPI_Comm_size on PETSC_COMM_WORLD, src/utils/world.cpp:40-42) */
/* Copy-engine path for the one large message of a step, the matL ghost rows (C11, MatSetValuesCOO's off-process entries,
 * src/impls/ecsim/simulation.cpp:366): every rank publishes a blob describing the buffers its z-neighbours write into
 * (xpic_comm_peer_export: IPC handles, hipIpcGetMemHandle), the caller carries the blobs to the neighbours over its bootstrap
 * channel (as it carries the ncclUniqueId), and each rank maps its lower and its upper neighbour's buffer
 * (xpic_comm_peer_import: hipIpcOpenMemHandle; ranks that are threads of one process, and a self-ring, use the addresses).
 * With bit 2 of xpic_set_overlap the ghost rows then travel as hipMemcpyAsync on a copy stream -- between two GPUs an SDMA
 * engine: no workgroup slot is taken from the assembly's colour launches they run beside -- and the rank's next ring
 * exchange, issued behind the copies, is the neighbour's arrival signal.  RCCL (or the callbacks) keep every other message. */
#define XPIC_PEER_BLOB_BYTES 256
int xpic_comm_peer_export(xpic_ctx* ctx, void* blob /* XPIC_PEER_BLOB_BYTES */);
int xpic_comm_peer_import(xpic_ctx* ctx, const void* lower_blob, const void* upper_blob);
int xpic_comm_size(xpic_ctx* ctx, int* nranks);
/* traffic of this rank since the last reset: out4 = {point-to-point messages sent, bytes sent, all-reduces, their payload
 * bytes} -- what a step puts on the links (the reference's MPI / PetscSF traffic) */
int xpic_comm_stats(xpic_ctx* ctx, int64_t* out4, int reset);

/* ---- measurement: HIP-event timers around kernel families, on the context's own stream */
int xpic_profile_enable(xpic_ctx* ctx, int on);
int xpic_profile_reset(xpic_ctx* ctx);
/* name: "matA_apply","matL_apply","matM_apply","fill_current","move_bin","scatter","second_push",... */
int xpic_profile_get(xpic_ctx* ctx, const char* name, int64_t* launches, double* total_ms);
/* device copy bandwidth probe (bytes moved / s) measured with a float4-style copy kernel */
int xpic_probe_copy_bandwidth(xpic_ctx* ctx, int64_t bytes, int reps, double* bytes_per_s);

#ifdef __cplusplus
}
#endif
#endif
