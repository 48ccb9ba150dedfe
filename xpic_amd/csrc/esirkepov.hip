// esirkepov.hip -- the particle phases that deposit the charge-conserving Esirkepov current:
//   MODE 0  basic::Particles::push            (src/impls/basic/particles.cpp:17-53)
//   MODE 1  ecsimcorr::Particles::first_push  (src/impls/ecsimcorr/particles.cpp:27-50)
//   MODE 2  ecsimcorr::Particles::second_push (src/impls/ecsimcorr/particles.cpp:52-91)
// with Shape (src/utils/shape.cpp:31-80), SimpleInterpolation (src/algorithms/simple_interpolation.cpp:8-38)
// and EsirkepovDecomposition (src/algorithms/esirkepov_decomposition.cpp:20-103) for cell-sorted SoA particles.
//
// A workgroup owns one x-pencil of cells and marches along it in chunks of 4 cells (one wave per cell); what a chunk
// needs from global memory -- the first pass of particles of each cell, the gather tile (MODE 0), the CIC
// neighbourhoods (MODE 2) -- is requested one chunk ahead.
//   * phase 1 (lane = particle) moves / pushes the particle.  MODE 0 gathers E, B with the 2nd-order shape out of an
//     LDS tile of the 9 x 6 x 6 nodes the workgroup's particles can reach at mid-step (only the three nodes per axis
//     and weight type that spline_of_2nd_order does not return as exact zeros: the sums are bitwise those of the
//     reference's loop over its 3..4-wide box); MODE 2 gathers with the CIC weights out of the cell's 36 + 54 value
//     neighbourhood in LDS, like k_second_push.
//   * A particle that starts in cell c and ends less than about half a cell away (every particle of the BASELINE
//     workloads) has old and new spline supports inside the nodes c-1 .. c+2 of each axis.  For those the deposit is
//     the dense 4 x 4 x 4 box per component: phase 1 stages the 1-D old / new spline values and the prefix sums of
//     their differences on these 4 nodes; in phase 2 lane = one of the 48 "lines" of the cell (component, two
//     transverse node indices), which adds  P_c[t] * T  to its 4 nodes particle after particle -- the reference's
//     running sum temp_j += W (:57-103) with the sum over the line taken first.  No atomics inside a cell.
//   * A particle that moves further (up to the reference's own limit of one cell, shape.h:18,91-92) deposits its box
//     directly with fp64 atomics (slow path, same arithmetic as the reference's loop).
//   * The chunk's lines are merged in a sliding LDS J window (7 x 4 x 4 nodes per component); the 4 columns the march
//     has passed leave with one fp64 atomic per node.
#include <cstring>

#include "common.h"
#include "device_common.h"

namespace xpic {

namespace {

constexpr int kBW = 4;            // cells (waves) per workgroup along x
// Particles staged per pass and wave.  A pass costs nearly the same whatever it holds (the wave issues the same VALU
// stream for 1 or 64 particles), so the pass is as wide as the LDS allows at the occupancy the registers allow: 48 with
// the gather tile (MODE 0, 2 workgroups/CU), 36 at 3 workgroups/CU (MODE 1), 48 at 2 (MODE 2).  A 32-ppc Poisson cell
// exceeds 48 particles with probability 0.003, 36 with 0.21, 32 with 0.45 (measured: basic 17.8 -> 12.0 ms,
// ecsimcorr pushes 7.0/10.9 -> 6.6/8.9 ms at 128^3 x 32).
#ifndef ESK_BC0
#define ESK_BC0 48
#endif
#ifndef ESK_BC1
#define ESK_BC1 36
#endif
#ifndef ESK_BC2
#define ESK_BC2 48
#endif
template <int MODE> struct StageDim {
  static constexpr int kBC = MODE == 0 ? ESK_BC0 : MODE == 1 ? ESK_BC1 : ESK_BC2;
  // row pitch: the 4 rows x 2 particles one LDS cycle of the phase-2 reads touches fall in 8 distinct bank pairs for
  // pitches 2, 6, 18, 22 (mod 32) -- 34, 38, 50 here
  static constexpr int kBPad = kBC + 2;
  static_assert(kBC % 4 == 0 && (kBPad % 32 == 2 || kBPad % 32 == 6 || kBPad % 32 == 18 || kBPad % 32 == 22), "stage pitch");
};
constexpr int kD = 4;             // deposit box per axis: nodes c-1 .. c+2
constexpr int kJX = kBW + kD - 1; // J tile nodes along x
constexpr int kJN = kJX * kD * kD;
constexpr int kSRows = 36;        // So[3][4], Sn[3][4], P[3][4]
constexpr int kT = 6;             // gather tile per axis (MODE 0): nodes c-2 .. c+3
constexpr int kTX = kBW + kT - 1;
constexpr int kTileN = kTX * kT * kT;
constexpr int kThreadsB = kBW * 64;
constexpr int kLinesB = 3 * kD * kD; // 48 lines of 4 nodes
constexpr int kFtPer = (6 * kTileN + kThreadsB - 1) / kThreadsB; // gather-tile values per thread
constexpr int kMaxNxB = 1024;        // pencils up to this length keep their cell_start row in LDS
static_assert(3 * kJN <= 2 * kThreadsB, "two window elements per thread");

// raw workgroup barrier that only drains LDS traffic: the particle stores, the J atomics and the requests of the next
// chunk stay in flight across it (__syncthreads() would wait for every one of them: a round trip to HBM per chunk)
__device__ inline void lds_barrier_b()
{
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ inline void wave_sync_b()
{
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// spline_of_2nd_order (src/interfaces/sort_parameters.cpp:21-30)
__device__ inline double spline2(double s)
{
  s = fabs(s);
  if (s <= 0.5) return (0.75 - s * s);
  if (0.5 < s && s < 1.5) return 0.5 * (1.5 - s) * (1.5 - s);
  return 0.0;
}

__device__ inline double wave_sum_b(double v)
{
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// value of lane `src` (wave-uniform) for every lane: v_readlane on the two halves
__device__ inline double lane_value(double v, int src)
{
  const long long x = __builtin_bit_cast(long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)x, src);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(x >> 32), src);
  return __builtin_bit_cast(double, ((long long)hi << 32) | lo);
}

// what a wave requests one chunk ahead: its next cell's first pass of particles (and, MODE 2, the cell's CIC
// neighbourhoods; MODE 0, this thread's share of the workgroup's next gather tile)
template <int MODE>
struct Ahead {
  int start, cnt;
  double r[3], v[3];
  double e, b;                       // MODE 2: lane's value of the 36 E / 54 B neighbourhood
  double ft[MODE == 0 ? kFtPer : 1]; // MODE 0: tile values t = thread + k * kThreadsB
};

// P2: power-of-two spacings (exact reciprocals instead of divisions, device_common.h: scaled_position)
template <int MODE, bool P2>
__global__ void __launch_bounds__(kThreadsB, 2) k_esirkepov_push(GridDev g, SortDev s, const double* __restrict__ E,
  const double* __restrict__ B, double* __restrict__ J, double qm, double alpha, double qn_Np, double* pred_w,
  int* bad_count)
{
  // workgroup -> the x-pencil (cy, cz), marched in chunks of kBW cells (one cell per wave)
  constexpr int kBC = StageDim<MODE>::kBC, kBPad = StageDim<MODE>::kBPad;
  const int cy = blockIdx.x % g.ny;
  const int cz = blockIdx.x / g.ny;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;

  __shared__ double stage[kBW][kSRows * kBPad];
  __shared__ double jtile[3 * kJN];
  __shared__ double ftile[MODE == 0 ? 6 * kTileN : 1];   // Ex,Ey,Ez,Bx,By,Bz on the gather tile (basic only)
  __shared__ double nbE[MODE == 2 ? kBW : 1][36], nbB[MODE == 2 ? kBW : 1][54]; // CIC neighbourhoods (second_push only)
  __shared__ double pwsum[kBW];
  __shared__ int cstart[kMaxNxB + 2];

  const long pencil0 = ((long)cz * g.ny + cy) * g.nx;
  const bool cs_lds = g.nx <= kMaxNxB;
  if (cs_lds)
    for (int i = threadIdx.x; i <= g.nx; i += kThreadsB) cstart[i] = s.cell_start[pencil0 + i];
  for (int t = threadIdx.x; t < 3 * kJN; t += kThreadsB) jtile[t] = 0.0;
  __syncthreads();

  // neighbourhood slot of this lane (MODE 2): numbering of the cell's 3 x 12 E nodes and 54 B nodes as in k_second_push
  int ec = 0, eo[3] = {0, 0, 0}, bc = 0, bo[3] = {0, 0, 0};
  if (MODE == 2) {
    if (lane < 36) {
      ec = lane / 12;
      const int l = lane % 12;
      if (ec == 0) { eo[0] = l % 3 - 1; eo[1] = (l / 3) % 2; eo[2] = l / 6; }
      else if (ec == 1) { eo[0] = l % 2; eo[1] = (l / 2) % 3 - 1; eo[2] = l / 6; }
      else { eo[0] = l % 2; eo[1] = (l / 2) % 2; eo[2] = l / 4 - 1; }
    }
    if (lane < 54) {
      if (lane < 18) { bc = 0; bo[0] = lane % 2; bo[1] = (lane / 2) % 3 - 1; bo[2] = lane / 6 - 1; }
      else if (lane < 36) { const int l = lane - 18; bc = 1; bo[0] = l % 3 - 1; bo[1] = (l / 3) % 2; bo[2] = l / 6 - 1; }
      else { const int l = lane - 36; bc = 2; bo[0] = l % 3 - 1; bo[1] = (l / 3) % 3 - 1; bo[2] = l / 9; }
    }
  }

  // the gather tile of chunk j (MODE 0): requested after the arithmetic of the chunk before, so that its registers
  // are not held across phases 1 and 2
  auto request_tile = [&](int j, Ahead<MODE>& pf) {
    const int cxa0 = j * kBW;
    if (MODE == 0 && cxa0 < g.nx) {
      // DMGlobalToLocal(E), (B) (basic/simulation.cpp:56-57) for just the nodes the chunk's particles can gather from
#pragma unroll
      for (int k = 0; k < kFtPer; ++k) {
        const int t = threadIdx.x + k * kThreadsB;
        pf.ft[k] = 0.0;
        if (t < 6 * kTileN) {
          const int f = t / kTileN, n = t % kTileN;
          const int tx = n % kTX, ty = (n / kTX) % kT, tz = n / (kTX * kT);
          const double* F = (f < 3 ? E : B) + (f % 3) * g.cstride;
          pf.ft[k] = F[g.nodew(cxa0 - 2 + tx, cy - 2 + ty, cz - 2 + tz)];
        }
      }
    }
  };
  auto request = [&](int j, Ahead<MODE>& pf) {
    pf.start = 0; pf.cnt = 0; pf.e = 0.0; pf.b = 0.0;
    const int cxa = j * kBW + wave;
    if (cxa >= g.nx) return;
    if (cs_lds) {
      pf.start = __builtin_amdgcn_readfirstlane(cstart[cxa]);
      pf.cnt = __builtin_amdgcn_readfirstlane(cstart[cxa + 1]) - pf.start;
    }
    else {
      pf.start = __builtin_amdgcn_readfirstlane(s.cell_start[pencil0 + cxa]);
      pf.cnt = __builtin_amdgcn_readfirstlane(s.cell_start[pencil0 + cxa + 1]) - pf.start;
    }
    if (MODE == 2 && pf.cnt > 0) {
      if (lane < 36) pf.e = E[ec * g.cstride + g.nodew(cxa + eo[0], cy + eo[1], cz + eo[2])];
      if (lane < 54) pf.b = B[bc * g.cstride + g.nodew(cxa + bo[0], cy + bo[1], cz + bo[2])];
    }
    // MODE 0 (the register-hungry 2nd-order gather) loads its particles when it gets there instead
    if (MODE != 0 && lane < min(kBC, pf.cnt)) {
      const long p = (long)pf.start + lane;
#pragma unroll
      for (int a = 0; a < 3; ++a) { pf.r[a] = s.r[a][p]; pf.v[a] = s.v[a][p]; }
    }
  };

  double pw = 0.0;
  int bad = 0;
  double* st = stage[wave];
  const double dt = g.dt;

  // the line of this lane: component c along its own axis; transverse axes (A, B) with the reference's roles
  //   X: A = y, B = z   get_jx :57-71      Y: A = x, B = z   get_jy :73-87      Z: A = y, B = x   get_jz :89-103
  const bool has_line = lane < kLinesB;
  const int lc = has_line ? lane / (kD * kD) : 0;
  const int liA = (lane % (kD * kD)) % kD, liB = (lane % (kD * kD)) / kD;
  const int axA = lc == 1 ? 0 : 1, axB = lc == 2 ? 0 : 2;
  const double qd = alpha * (lc == 0 ? g.dx : (lc == 1 ? g.dy : g.dz)); // the slow path's line
  const double qdc[3] = {alpha * g.dx, alpha * g.dy, alpha * g.dz};
  const int kk = lane >> 4, qb = (lane >> 2) & 3, qq = lane & 3;         // phase 2: particle of the step, block, row / column

  Ahead<MODE> pf;
  request_tile(0, pf);
  request(0, pf);
  const int nch = (g.nx + kBW - 1) / kBW;
  for (int j = 0; j < nch; ++j) {
    const int cx0 = j * kBW, cx = cx0 + wave;
    const bool active = cx < g.nx;
    const int cc[3] = {cx, cy, cz + g.z0};
    // ---- this chunk's tile / neighbourhoods, requested one chunk ago, go to LDS; then the next chunk's requests
    if (MODE == 0) {
#pragma unroll
      for (int k = 0; k < kFtPer; ++k) {
        const int t = threadIdx.x + k * kThreadsB;
        if (t < 6 * kTileN) ftile[t] = pf.ft[k];
      }
      lds_barrier_b();
    }
    if (MODE == 2) {
      if (lane < 36) nbE[wave][lane] = pf.e;
      if (lane < 54) nbB[wave][lane] = pf.b;
    }
    const int start = pf.start, cnt = pf.cnt;
    double fr[3] = {pf.r[0], pf.r[1], pf.r[2]}, fv[3] = {pf.v[0], pf.v[1], pf.v[2]};
    request(j + 1, pf);

    double acc[3] = {0.0, 0.0, 0.0}; // one 4 x 4 block element per component (phase 2)
    for (int base = 0; base < cnt; base += kBC) {
      const int mcnt = min(kBC, cnt - base);
      wave_sync_b();
      // a particle whose old / new supports leave the cell's 4-node box: deposited by the whole wave further down
      bool slow = false;
      double po[3] = {0, 0, 0}, pn[3] = {0, 0, 0};
      int sst[3] = {0, 0, 0}, ssz[3] = {0, 0, 0};
      if (lane >= mcnt && lane < ((mcnt + 3) & ~3)) {
        // phase 2 works on K = 4 particles per step: the columns that fill up the last step are particles of weight zero
        double* col = st + lane;
#pragma unroll
        for (int e = 0; e < kSRows; ++e) col[e * kBPad] = 0.0;
      }
      if (lane < mcnt) {
        const long p = (long)start + base + lane;
        double r[3] = {fr[0], fr[1], fr[2]}, v[3] = {fv[0], fv[1], fv[2]};
        if (MODE == 0 || base > 0) { // cells beyond one pass (and MODE 0): plain loads
#pragma unroll
          for (int a = 0; a < 3; ++a) { r[a] = s.r[a][p]; v[a] = s.v[a][p]; }
        }
        const double old_r[3] = {r[0], r[1], r[2]};
        double Ep[3] = {0, 0, 0}, Bp[3] = {0, 0, 0};

        if (MODE == 0) {
          // push.update_r(dt / 2) ; shape.setup(point.r) ; interpolation.process   (basic/particles.cpp:31-38)
#pragma unroll
          for (int a = 0; a < 3; ++a) r[a] += v[a] * (dt / 2.0);
          // Shape::make_start / make_end (shape.cpp:12-28) give the box [sst, sst + ssz), ssz = 3 or 4.  Inside it
          // spline_of_2nd_order is an exact zero outside three nodes: for the node-centred weights ("No") those are
          // sst + nN .. sst + nN + 2 with nN = 0 or 1, for the half-shifted ones ("Sh") always sst .. sst + 2.
          int off[3], nN[3];
          double No[3][3], Sh[3][3];
          bool inside = true;
          double prs[3];
          scaled_position<P2>(g, r[0], r[1], r[2], prs);
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            const double pr = prs[a];
            const int sst = (int)round(pr - 1.5);
            const int ssz = (int)floor(pr + 1.5) + 1 - sst;
            inside = inside && sst >= cc[a] - 2 && sst + ssz <= cc[a] + 4;
            nN[a] = (pr - (double)sst) < 1.5 ? 0 : 1; // node sst has |pr - sst| < 1.5, or node sst + 3 may have
            off[a] = sst - (cc[a] - 2);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
              No[a][t] = spline2(pr - (double)(sst + nN[a] + t));  // Shape::fill, :57-80
              Sh[a][t] = spline2(pr - ((double)(sst + t) + 0.5));
            }
          }
          if (inside) {
            off[0] += wave; // tile x origin is cx0 - 2, the cell's own is cx - 2
            // the reference's loop order (x fastest, then y, z), zero terms left out
            const int bN[3] = {off[0] + nN[0], off[1] + nN[1], off[2] + nN[2]};
            // the z loop stays a loop: unrolled, the compiler keeps all 36 weight products of the box live at once
#pragma nounroll
            for (int kz = 0; kz < 3; ++kz) {
              const double nz = No[2][kz], sz = Sh[2][kz];
              const int zN = (bN[2] + kz) * kT, zS = (off[2] + kz) * kT;
#pragma unroll
              for (int jy = 0; jy < 3; ++jy) {
                const int yN = bN[1] + jy, yS = off[1] + jy;
                const double nn = nz * No[1][jy], ns = nz * Sh[1][jy], sn = sz * No[1][jy], ss = sz * Sh[1][jy];
#pragma unroll
                for (int ix = 0; ix < 3; ++ix) {
                  // Shape::electric / magnetic (shape.h:54-72): each component has its own three nodes per axis
                  const int xN = bN[0] + ix, xS = off[0] + ix;
                  Ep[0] += ftile[0 * kTileN + (zN + yN) * kTX + xS] * (nn * Sh[0][ix]);
                  Ep[1] += ftile[1 * kTileN + (zN + yS) * kTX + xN] * (ns * No[0][ix]);
                  Ep[2] += ftile[2 * kTileN + (zS + yN) * kTX + xN] * (sn * No[0][ix]);
                  Bp[0] += ftile[3 * kTileN + (zS + yS) * kTX + xN] * (ss * No[0][ix]);
                  Bp[1] += ftile[4 * kTileN + (zS + yN) * kTX + xS] * (sn * Sh[0][ix]);
                  Bp[2] += ftile[5 * kTileN + (zN + yS) * kTX + xS] * (ns * Sh[0][ix]);
                }
              }
            }
          }
          // else: the particle moved further than the reference itself supports; flagged by the range test below
          update_vEB(dt, qm, Ep, Bp, v);                 // push.update_vEB(dt)  :41
#pragma unroll
          for (int a = 0; a < 3; ++a) r[a] += v[a] * (dt / 2.0); // push.update_r(dt / 2)  :42
        }
        else if (MODE == 1) {
#pragma unroll
          for (int a = 0; a < 3; ++a) r[a] += v[a] * (0.5 * dt); // BorisPush::update_r(0.5 * dt)  ecsimcorr/particles.cpp:39
        }
        else {
          // interpolate_E_s1 / interpolate_B_s1 ; update_vEB(dt) ; update_r(0.5 dt)   (:64-69), gathers out of the
          // cell's LDS neighbourhood in the loop and product order of ecsim/simulation.cpp:8-118
          const double old_v[3] = {v[0], v[1], v[2]};
          const W1T<P2> w(g, r[0], r[1], r[2]);
          const int ox = w.is[0] - w.in[0] + 1, oy = w.is[1] - w.in[1] + 1, oz = w.is[2] - w.in[2] + 1;
          const double* eE = nbE[MODE == 2 ? wave : 0];
          const double* eB = nbB[MODE == 2 ? wave : 0];
#pragma unroll
          for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int i = 0; i < 2; ++i) {
                Ep[0] += eE[(k * 2 + j) * 3 + (ox + i)] * (w.wn[2][k] * w.wn[1][j] * w.ws[0][i]);
                Ep[1] += eE[12 + (k * 3 + (oy + j)) * 2 + i] * (w.wn[2][k] * w.ws[1][j] * w.wn[0][i]);
                Ep[2] += eE[24 + ((oz + k) * 2 + j) * 2 + i] * (w.ws[2][k] * w.wn[1][j] * w.wn[0][i]);
                Bp[0] += eB[((oz + k) * 3 + (oy + j)) * 2 + i] * (w.ws[2][k] * w.ws[1][j] * w.wn[0][i]);
                Bp[1] += eB[18 + ((oz + k) * 2 + j) * 3 + (ox + i)] * (w.ws[2][k] * w.wn[1][j] * w.ws[0][i]);
                Bp[2] += eB[36 + (k * 3 + (oy + j)) * 3 + (ox + i)] * (w.wn[2][k] * w.ws[1][j] * w.ws[0][i]);
              }
          update_vEB(dt, qm, Ep, Bp, v);
#pragma unroll
          for (int a = 0; a < 3; ++a) r[a] += v[a] * (0.5 * dt);
          // pred_w += qn_Np * 0.5 * (old_v + point.p).dot(E_p)   (:77-78)
          pw += qn_Np * 0.5 * ((old_v[0] + v[0]) * Ep[0] + (old_v[1] + v[1]) * Ep[1] + (old_v[2] + v[2]) * Ep[2]);
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          s.r[a][p] = r[a];
          if (MODE != 1) s.v[a][p] = v[a];
        }

        // Shape::setup(old_r, new_r) (shape.cpp:43-54): the box [sst, send) of the pair per axis
        bool ok = true, fast = true;
        scaled_position<P2>(g, old_r[0], old_r[1], old_r[2], po);
        scaled_position<P2>(g, r[0], r[1], r[2], pn);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          sst[a] = (int)round(fmin(po[a], pn[a]) - 1.5);
          const int send = (int)floor(fmax(po[a], pn[a]) + 1.5) + 1;
          ssz[a] = send - sst[a];
          ok = ok && ssz[a] <= 4;                                       // the reference's Shape::shape[] limit
          fast = fast && sst[a] >= cc[a] - 1 && send <= cc[a] + 3;      // inside the cell's dense 4-node box
        }
        double* col = st + lane;
        if (ok && fast) {
          // old / new spline values on the nodes c-1 .. c+2 (exact zeros outside the support, as the reference's loop
          // sees them) and the prefix sums of their differences along each axis
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            double run = 0.0;
#pragma unroll
            for (int t = 0; t < kD; ++t) {
              const double gx = (double)(cc[a] - 1 + t);
              const double so = spline2(po[a] - gx), sn = spline2(pn[a] - gx);
              run += sn - so;
              col[(a * kD + t) * kBPad] = so;
              col[(12 + a * kD + t) * kBPad] = sn;
              col[(24 + a * kD + t) * kBPad] = run;
            }
          }
        }
        else {
#pragma unroll
          for (int e = 0; e < kSRows; ++e) col[e * kBPad] = 0.0; // nothing for phase 2
          if (ok) slow = true;
          else ++bad; // the reference would overflow Shape::shape here; deposit nothing and report
        }
      }
      // ---- slow path (rare): EsirkepovDecomposition::process of ONE particle by the 48 line lanes, on the particle's
      // own box [sst, sst + ssz) (Shape::setup(old, new), shape.cpp:43-54), straight to the global J with fp64 atomics.
      // Same terms and running sums as :57-103.
      for (unsigned long long sm = __ballot(slow); sm; sm &= sm - 1) {
        const int src = __ffsll((long long)sm) - 1;
        double qo[3], qn[3];
        int qs[3], qz[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          qo[a] = lane_value(po[a], src);
          qn[a] = lane_value(pn[a], src);
          qs[a] = __builtin_amdgcn_readlane(sst[a], src);
          qz[a] = __builtin_amdgcn_readlane(ssz[a], src);
        }
        if (has_line && liA < qz[axA] && liB < qz[axB]) {
          const double gA = (double)(qs[axA] + liA), gB = (double)(qs[axB] + liB);
          const double sA_o = spline2(qo[axA] - gA), sA_n = spline2(qn[axA] - gA);
          const double sB_o = spline2(qo[axB] - gB), sB_n = spline2(qn[axB] - gB);
          const double T = sA_n * (2.0 * sB_n + sB_o) + sA_o * (2.0 * sB_o + sB_n);
          double run = 0.0;
          for (int t = 0; t < qz[lc]; ++t) {
            const double gC = (double)(qs[lc] + t);
            run = run + (-qd * (spline2(qn[lc] - gC) - spline2(qo[lc] - gC)) * T);
            int n[3];
            n[lc] = qs[lc] + t; n[axA] = qs[axA] + liA; n[axB] = qs[axB] + liB;
            if (run != 0.0) unsafeAtomicAdd(&J[lc * g.cstride + g.nodew(n[0], n[1], n[2] - g.z0)], run);
          }
        }
      }
      wave_sync_b();

      // ---- phase 2 on the matrix cores: J_c[i][u][w] += sum_p P_c[i] * T_c[u][w],
      //   T_c[u][w] = -qd_c (Sn_A[u] (2 Sn_B[w] + So_B[w]) + So_A[u] (2 So_B[w] + Sn_B[w]))
      // is, per component, a 4 x 16 x P product: one v_mfma_f64_4x4x4_4b_f64 per K = 4 particles with block b = w, rows
      // i, columns u (lane roles A[b][i][k], B[b][k][j] at lane 16 k + 4 b + (i or j); D[b][i][j] at lane 16 i + 4 b + j).
      // The lane forms its own T from the staged 1-D spline values of particle k.
      for (int t0 = 0; t0 < mcnt; t0 += 4) {
        const double* cp = st + t0 + kk;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int aA = c == 1 ? 0 : 1, aB = c == 2 ? 0 : 2; // X: A = y, B = z;  Y: A = x, B = z;  Z: A = y, B = x
          const double sA_o = cp[(aA * kD + qq) * kBPad], sA_n = cp[(12 + aA * kD + qq) * kBPad];
          const double sB_o = cp[(aB * kD + qb) * kBPad], sB_n = cp[(12 + aB * kD + qb) * kBPad];
          const double T = -qdc[c] * (sA_n * (2.0 * sB_n + sB_o) + sA_o * (2.0 * sB_o + sB_n));
          acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(cp[(24 + c * kD + qq) * kBPad], T, acc[c], 0, 0, 0);
        }
      }
    }

    request_tile(j + 1, pf);
    // ---- merge the chunk's cells in the LDS J window: column 0 is node x = cx0 - 1
    if (active) {
      // lane holds J_c[i = lane >> 4][u = lane & 3][w = (lane >> 2) & 3]: own-axis node i, A-index u, B-index w
      const int di = lane >> 4;
      const int tX[3] = {wave + di, wave + qq, wave + qb}, tY[3] = {qq, di, qq}, tZ[3] = {qb, qb, di};
#pragma unroll
      for (int c = 0; c < 3; ++c)
        if (acc[c] != 0.0) unsafeAtomicAdd(&jtile[c * kJN + (tZ[c] * kD + tY[c]) * kJX + tX[c]], acc[c]);
    }
    lds_barrier_b();
    // ---- the next chunk starts at node cx0 + 3: the first kBW columns are final and leave with one fp64 atomic per
    // node (other pencils add to the same nodes); the other three move to the front of the window.  The last chunk
    // flushes everything (its tail wraps periodically onto nodes 0, 1, ...).
    const bool last = j + 1 == nch;
    double keep[2] = {0.0, 0.0};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int t = threadIdx.x + k * kThreadsB;
      if (t < 3 * kJN) {
        const double val = jtile[t];
        const int c = t / kJN, n = t % kJN;
        const int tx = n % kJX, ty = (n / kJX) % kD, tz = n / (kJX * kD);
        if (tx < kBW || last) {
          if (val != 0.0) unsafeAtomicAdd(&J[c * g.cstride + g.nodew(cx0 - 1 + tx, cy - 1 + ty, cz - 1 + tz)], val);
        }
        // what this thread's element holds next: the value kBW columns further on, zero for the new columns
        keep[k] = (tx + kBW < kJX && !last) ? jtile[t + kBW] : 0.0;
      }
    }
    lds_barrier_b();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int t = threadIdx.x + k * kThreadsB;
      if (t < 3 * kJN) jtile[t] = keep[k];
    }
    lds_barrier_b();
  }

  if (MODE == 2) {
    pw = wave_sum_b(pw);
    if (lane == 0) pwsum[wave] = pw;
    __syncthreads();
    // pred_w: one partial per workgroup, summed by k_sum_partials in a fixed order (atomics of every wave on one
    // address would cost more than the whole push)
    if (threadIdx.x == 0) pred_w[blockIdx.x] = (pwsum[0] + pwsum[1]) + (pwsum[2] + pwsum[3]);
  }
  if (bad) atomicAdd(bad_count, bad);
}

// deterministic sum of n doubles by one workgroup
__global__ void __launch_bounds__(1024) k_sum_partials(const double* __restrict__ part, long n, double* out)
{
  double v = 0.0;
  for (long i = threadIdx.x; i < n; i += 1024) v += part[i];
  __shared__ double sm[16];
  v = wave_sum_b(v);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += sm[w];
    *out = t;
  }
}

}  // namespace

int esirkepov_push(xpic_ctx* c, Sort& s, int mode, const double* E, const double* B, double* J, double* pred_w_host)
{
  s.prebinned = false;
  if (pred_w_host) *pred_w_host = 0.0;
  if (s.n == 0) {
    // an empty slab still takes part in the collective (pred_w and the error count)
    double red[2] = {0.0, 0.0};
    XPIC_CALL(comm_allreduce_sum_host(c, red, 2));
    if (pred_w_host) *pred_w_host = red[0];
    if (red[1] != 0.0) {
      set_error(std::to_string((long)red[1]) + " particle(s) moved more than one cell in an Esirkepov step on another z-slab");
      return 6;
    }
    return 0;
  }
  const GridDev& g = c->g;
  XPIC_CHECK(g.nx >= 6 && g.ny >= 6 && g.nzl >= 6, "the Esirkepov tile needs every grid extent >= 6");
  const double qm = s.par.q / s.par.m;
  const double qn_Np = s.par.q * s.par.n / s.par.Np;
  const double alpha = qn_Np / (6.0 * g.dt); // basic/particles.cpp:44, ecsimcorr/particles.cpp:130
  double* scal = c->red_out;                  // [0] pred_w, [1] bad count (as int)
  XPIC_HIP(hipMemsetAsync(scal, 0, sizeof(double) * 2, c->stream));
  const long nblocks = (long)g.ny * g.nzl; // one workgroup per x-pencil
  XPIC_CHECK(nblocks < 2147483647L, "too many pencils for one launch");
  const char* name = mode == 0 ? "basic_push" : (mode == 1 ? "corr_first_push" : "corr_second_push");
  {
    Timed t(c, name);
    dim3 grid((unsigned)nblocks), block(kThreadsB);
#define ESK(M) (g.pow2 ? k_esirkepov_push<M, true> : k_esirkepov_push<M, false>)
    if (mode == 0) hipLaunchKernelGGL(ESK(0), grid, block, 0, c->stream, g, s.d, E, B, J, qm, alpha, qn_Np, scal, (int*)(scal + 1));
    else if (mode == 1) hipLaunchKernelGGL(ESK(1), grid, block, 0, c->stream, g, s.d, E, B, J, qm, alpha, qn_Np, scal, (int*)(scal + 1));
    else {
      // per-workgroup pred_w partials go to the (idle) Krylov work vector: one double per pencil
      XPIC_CHECK(c->kry_w, "second_push needs the ecsimcorr scheme's work vectors");
      hipLaunchKernelGGL(ESK(2), grid, block, 0, c->stream, g, s.d, E, B, J, qm, alpha, qn_Np, c->kry_w, (int*)(scal + 1));
#undef ESK
      hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, c->stream, c->kry_w, nblocks, scal);
    }
    XPIC_HIP(hipGetLastError());
  }
  XPIC_HIP(hipMemcpyAsync(c->red_host, scal, sizeof(double) * 2, hipMemcpyDeviceToHost, c->stream));
  XPIC_HIP(hipStreamSynchronize(c->stream));
  int bad;
  memcpy(&bad, &c->red_host[1], sizeof(int));
  // one collective for both scalars: MPI_Allreduce(pred_w) (ecsimcorr/particles.cpp:85) and the error count, so that
  // every z-slab leaves with the same return code (a rank returning alone would leave its neighbours waiting)
  double red[2] = {c->red_host[0], (double)bad};
  XPIC_CALL(comm_allreduce_sum_host(c, red, 2));
  if (pred_w_host) *pred_w_host = red[0];
  if (red[1] != 0.0) {
    set_error(std::to_string((long)red[1]) + " particle(s) moved more than one cell in an Esirkepov step "
      "(the reference overflows Shape::shape[] here, src/utils/shape.h:18,91-92)");
    return 6;
  }
  return 0;
}

}  // namespace xpic
