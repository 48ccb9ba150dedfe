/*
 * xpic_oracle.h -- CPU restatement of xpic's per-timestep hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product path
 * (xpic_amd/ + include/xpic_hip.h) never links, imports or calls it.
 *
 * The reference itself (C++20 + PETSc + MPI) is unbuildable in this image (no PETSc,
 * no <format>), so this file re-states its algorithm in plain C++17/OpenMP, function by
 * function, each citing the reference file:line it follows.  It is pinned against the
 * reference's own golden files (tests/golden/, copied from the reference's
 * tests/<x>/expected/<x>/temporal/ tables): see tests/test_oracle_golden.py.
 *
 * All citations are relative to the reference checkout (vakurshakov/xpic).
 *
 * Array conventions (identical to the reference's DMDA vectors):
 *   field vector  : double[nz][ny][nx][3]   (x fastest, component interleaved)
 *   particle      : double[6] = {x, y, z, vx, vy, vz}   (struct Point, point.h:7-35)
 *   matL (stencil): double[3N][ORC_LSTENCIL] -- see orc_lstencil_decode()
 */
#ifndef XPIC_ORACLE_H
#define XPIC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_LSTENCIL 123 /* 27 same-component + 48 + 48 cross-component couplings per row */

typedef struct orc_sim orc_sim;

/* ---- single-particle kernels (src/algorithms/boris_push.cpp) ---------------------- */
/* boris_push.cpp:19-22 */
void orc_update_r(double dt, double* point6);
/* boris_push.cpp:48-57 */
void orc_update_vEB(double dt, double qm, const double* E_p, const double* B_p, double* point6);
/* boris_push.cpp:24-46,60-91; kind: 'M','B','1' (C1),'2' (C2) */
void orc_update_vX(char kind, double dt, double qm, const double* B_p, double* point6);

/* Drivers of tests/boris_push/boris_push_ex{1..6}.cpp: runs the whole trajectory of one
 * (example, scheme id) pair and writes the traced rows {t*dt, x, y, z, vx, vy, vz}.
 * Returns number of rows written (<= max_rows), or -1 on unknown id. */
int orc_boris_test_trajectory(int example, const char* scheme_id, double* rows, int max_rows);

/* ---- shape functions (src/interfaces/sort_parameters.cpp:3-78) -------------------- */
double orc_spline(int order, double s);

/* Shape::setup (src/utils/shape.cpp:31-80). pair=0: (r) -> No/Sh; pair=1: (old_r,new_r) -> Old/New.
 * shape_out holds size[0]*size[1]*size[2]*6 doubles in the reference's i_p() order.
 * Returns 0, or 1 if a size exceeds shape_width=4 (the reference would overflow). */
int orc_shape_setup(const double* d3, const double* r1, const double* r2, int pair, double radius,
  int order, int* start3, int* size3, double* shape_out);

/* ---- simulation object ------------------------------------------------------------ */
/* scheme: 0 = basic, 1 = ecsim, 2 = ecsimcorr.  Periodic box, as every BASELINE config. */
orc_sim* orc_create(int scheme, int nx, int ny, int nz, double dx, double dy, double dz, double dt);
void orc_destroy(orc_sim*);
void orc_set_threads(int n);

/* init_particles (src/interfaces/simulation.tpp:7-79): returns sort index */
int orc_add_sort(orc_sim*, int Np, double n, double q, double m, double Tx, double Ty, double Tz);

/* SetParticles::execute (src/commands/set_particles.cpp:19-43) with CoordinateInBox over the
 * whole domain + MaxwellianMomentum(tov) (src/utils/particles_load.cpp:11-18,52-76), drawing from
 * the ONE global default-seeded std::mt19937 (src/utils/random_generator.h:20-35).
 * Returns number of particles added. */
long orc_load_maxwell_box(orc_sim*, int sort, int tov);
void orc_reset_rng(void);

/* add_particle (src/interfaces/particles.cpp:47-67) for n points {x,y,z,vx,vy,vz}; returns #added */
long orc_add_particles(orc_sim*, int sort, long n, const double* pts6);
long orc_count(orc_sim*, int sort);
/* copies particles cell by cell (storage order); cell_of[i] = local cell index g of particle i */
long orc_get_particles(orc_sim*, int sort, double* pts6, int* cell_of);
void orc_clear_particles(orc_sim*, int sort);

/* named vectors: "E","B","B0","Ep","Ec","J","currI","currJe"; per sort "J<s>" handled by
 * orc_get_sort_current().  Layout [z][y][x][3]. */
int orc_set_field(orc_sim*, const char* name, const double* v);
int orc_get_field(orc_sim*, const char* name, double* v);
int orc_get_sort_current(orc_sim*, int sort, const char* which, double* v); /* "J","currI","currJe" */

/* ---- grid operators (src/utils/operators.cpp:155-215) ----------------------------- */
/* y = alpha * rot(+/-) x ; sign=+1 positive Yee shift (rotE), -1 negative (rotB) */
void orc_rot_apply(orc_sim*, int sign, double alpha, const double* x, double* y);
/* y = matM x, matM = 2 I + 0.5 dt^2 rot(-) rot(+)   (src/impls/ecsim/simulation.cpp:544-551) */
void orc_matM_apply(orc_sim*, const double* x, double* y);
/* y = matL x (assembled by orc_ecsim_fill_current) */
void orc_matL_apply(orc_sim*, const double* x, double* y);
/* negative-shift divergence (operators.cpp:275-333) of a 3-dof vector -> scalar [z][y][x] */
void orc_div_neg(orc_sim*, const double* v3, double* out1);

/* (c1, k) -> (c2, dx, dy, dz) of the fixed-stencil matL layout shared with the HIP library */
void orc_lstencil_decode(int c1, int k, int* c2, int* d3);
int orc_lstencil_encode(int c1, int c2, int dx, int dy, int dz);
/* copies matL in stencil form: double[3N][ORC_LSTENCIL], row = ((z*ny+y)*nx+x)*3+c */
void orc_get_matL(orc_sim*, double* out);

/* ---- grid <-> particle kernels on explicit arrays --------------------------------- */
/* SimpleInterpolation (src/algorithms/simple_interpolation.cpp:8-38) with the 2nd order Shape:
 * E and B are global [z][y][x][3] periodic fields; out6 = {E_p, B_p} */
void orc_gather_s2(orc_sim*, const double* E, const double* B, const double* r3, double* out6);
/* interpolate_E_s1 / interpolate_B_s1 (src/impls/ecsim/simulation.cpp:8-118) */
void orc_gather_s1(orc_sim*, const double* E, const double* B, const double* r3, double* out6);
/* EsirkepovDecomposition (src/algorithms/esirkepov_decomposition.cpp:20-103) of n moves
 * old6[i].r -> new6[i].r with alpha; adds into global periodic J [z][y][x][3] */
int orc_esirkepov(orc_sim*, long n, const double* old_r3, const double* new_r3, double alpha, double* J);

/* ---- per-phase entry points (what the HIP library mirrors) ------------------------- */
/* basic::Particles::push (src/impls/basic/particles.cpp:17-53) for every sort, fields E,B of sim */
int orc_basic_push(orc_sim*);
/* interfaces::Particles::update_cells_seq (src/interfaces/particles.cpp:79-116) */
void orc_update_cells(orc_sim*, int sort);
/* ecsim::Particles::first_push (src/impls/ecsim/particles.cpp:21-31) */
void orc_ecsim_first_push(orc_sim*, int sort);
/* ecsim::Simulation::fill_ecsim_current + Particles::fill_ecsim_current/decompose_ecsim_current
 * (src/impls/ecsim/simulation.cpp:336-368,471-484; particles.cpp:33-173): zeroes and fills currI, matL */
void orc_ecsim_fill_current(orc_sim*);
/* ecsim::Particles::second_push (src/impls/ecsim/particles.cpp:175-192) with E_arr=Ep, B_arr=B */
void orc_ecsim_second_push(orc_sim*, int sort);
/* ecsimcorr::Particles::{first_push,second_push,final_update,calculate_energy}
 * (src/impls/ecsimcorr/particles.cpp:27-150) */
int orc_ecsimcorr_first_push(orc_sim*, int sort);
int orc_ecsimcorr_second_push(orc_sim*, int sort);
void orc_ecsimcorr_final_update(orc_sim*, int sort);
double orc_calculate_energy(orc_sim*, int sort);
/* pred_w, corr_w, lambda_dK, pred_dK, corr_dK, energy of a sort (ecsimcorr/particles.h:44-49) */
void orc_ecsimcorr_scalars(orc_sim*, int sort, double* out6);

/* advance_fields(ksp, curr, out) (src/impls/ecsim/simulation.cpp:255-279).
 * op: 0 = matL + matM (GMRES(30)), 1 = matM with GMRES(30), 2 = matM with CG.
 * No preconditioner; x0 = 0; converged when ||r|| <= max(rtol*||b||, atol); maxit as given.
 * Returns iterations, <0 if not converged (the reference aborts: KSPSetErrorIfNotConverged). */
int orc_solve(orc_sim*, int op, const double* rhs, double* x, double rtol, double atol, int maxit,
  double* final_rnorm);
void orc_set_tolerances(orc_sim*, double rtol, double atol, int maxit);
/* wall time and iterations spent inside the Krylov solves of orc_step since the last reset (timing aid of bench.py's
 * cpu_baseline leg; no reference counterpart) */
void orc_solve_stats(orc_sim*, double* seconds, long* iterations, int reset);

/* timestep_implementation of the scheme (basic/simulation.cpp:30-43, ecsim/simulation.cpp:145-155,
 * ecsimcorr/simulation.cpp:21-32).  Returns KSP iterations of the step (sum), <0 on failure. */
int orc_step(orc_sim*);

/* ---- diagnostics used by the golden tables ---------------------------------------- */
/* Energy::calculate_field/calculate_kinetic (src/diagnostics/energy.cpp:43-108):
 * out = {wE, wB, sE, sB, wK_0, sK_0, wK_1, sK_1, ...} */
void orc_energy(orc_sim*, double* out);
/* ChargeConservation (src/diagnostics/charge_conservation.cpp:67-171): collect() rho for all sorts
 * (call once at t=0 = initialize()), then per step add_columns(): out = {N1_0, N2_0, ..., N1_tot, N2_tot} */
void orc_charge_collect(orc_sim*);
void orc_charge_columns(orc_sim*, double* out);
/* ParticlesChargeDensity::collect of one sort into rho[z][y][x] */
void orc_charge_density(orc_sim*, int sort, double* rho);
/* MomentumConservation::calculate (src/diagnostics/momentum_conservation.cpp:77-131):
 * out = {Px, Py, Pz, QEx, QEy, QEz} per sort */
void orc_momentum(orc_sim*, double* out);
/* eccapfim inner kernels.  cell_traversal (src/impls/eccapfim/cell_traversal.cpp:3-77): returns the
 * number of points (start, face crossings, end), the first max_pts of them in pts[3*i..]. */
int orc_cell_traversal(const double* d3, const double* end3, const double* start3, int max_pts, double* pts);
/* ImplicitEsirkepov::interpolate / decompose (src/algorithms/implicit_esirkepov.cpp:60-117) for n segments r0 -> rn */
void orc_implicit_esirkepov_interpolate(orc_sim*, long n, const double* rn3, const double* r03, double* Ep3, double* Bp3);
int orc_implicit_esirkepov_decompose(orc_sim*, long n, const double* alpha, const double* v3, const double* rn3,
  const double* r03, const char* field);
/* DistributionMoment "density" (src/diagnostics/distribution_moment.cpp:125-216) -> out[z][y][x] */
void orc_moment_density(orc_sim*, int sort, double* out);

#ifdef __cplusplus
}
#endif
#endif
