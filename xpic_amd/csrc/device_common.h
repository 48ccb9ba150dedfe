// device_common.h -- device-side helpers shared by the particle kernels.
#pragma once

#include "common.h"

namespace xpic {

// g_bound_periodic (src/interfaces/point.cpp:18-26): ONE wrap, s == L is left as is.
__device__ inline double bound_periodic(double s, double L)
{
  if (s < 0.0) return L - (0.0 - s);
  if (s > L) return 0.0 + (s - L);
  return s;
}

// the same values without branches (both folds are formed, one is selected): for kernels that are bound by instruction
// issue, where three divergent branches per particle cost more than six additions
__device__ inline double bound_periodic_sel(double s, double L)
{
  const double lo = L - (0.0 - s), hi = 0.0 + (s - L);
  double r = s > L ? hi : s;
  r = s < 0.0 ? lo : r;
  return r;
}

// (x / dx, y / dy, z / dz).  P2: all three spacings are powers of two (GridDev::pow2) and the hot kernels are
// instantiated for that case with the exact multiplication by the reciprocal: the same bits at a fraction of the cost
// of three fp64 divisions per position.
template <bool P2 = false>
__device__ inline void scaled_position(const GridDev& g, double x, double y, double z, double* pn)
{
  if (P2) { pn[0] = x * g.inv[0]; pn[1] = y * g.inv[1]; pn[2] = z * g.inv[2]; }
  else { pn[0] = x / g.dx; pn[1] = y / g.dy; pn[2] = z / g.dz; }
}

// Local cell of a position: FLOOR_STEP(s, ds) = floor(s / ds) (src/utils/utils.h:78), bounds test of
// update_cells_seq / add_particle (src/interfaces/particles.cpp:47-67, 90-104).  -1 = outside: dropped.
template <bool P2 = false>
__device__ inline int cell_of(const GridDev& g, double x, double y, double z)
{
  double pn[3];
  scaled_position<P2>(g, x, y, z, pn);
  int cx = (int)floor(pn[0]), cy = (int)floor(pn[1]), cz = (int)floor(pn[2]) - g.z0;
  if (cx < 0 || cx >= g.nx || cy < 0 || cy >= g.ny || cz < 0 || cz >= g.nzl) return -1;
  return (cz * g.ny + cy) * g.nx + cx;
}

// ---- CIC weights of ecsim (src/impls/ecsim/simulation.cpp:12-45): node index floor(x/dx), half-shifted
// index floor(x/dx - 1/2), weights of the upper neighbour = fractional part.
template <bool P2>
struct W1T {
  int in[3], is[3];
  double wn[3][2], ws[3][2];
  __device__ inline W1T(const GridDev& g, double x, double y, double z)
  {
    double pn[3];
    scaled_position<P2>(g, x, y, z, pn);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double ps = pn[a] - 0.5;
      in[a] = (int)floor(pn[a]);
      is[a] = (int)floor(ps);
      wn[a][1] = pn[a] - in[a];
      wn[a][0] = 1 - wn[a][1];
      ws[a][1] = ps - is[a];
      ws[a][0] = 1 - ws[a][1];
    }
    in[2] -= g.z0;
    is[2] -= g.z0;
  }
};
using W1 = W1T<false>;

// interpolate_E_s1 / interpolate_B_s1 (src/impls/ecsim/simulation.cpp:8-62, 64-118), same loop order
template <class W>
__device__ inline void gather_s1(const GridDev& g, const double* __restrict__ E, const double* __restrict__ B,
  const W& w, double* Ep, double* Bp)
{
  const double* Ex = E; const double* Ey = E + g.cstride; const double* Ez = E + 2 * g.cstride;
  const double* Bx = B; const double* By = B + g.cstride; const double* Bz = B + 2 * g.cstride;
  Ep[0] = Ep[1] = Ep[2] = 0.0;
  Bp[0] = Bp[1] = Bp[2] = 0.0;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int zn = g.wz(w.in[2] + k), zs = g.wz(w.is[2] + k);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int yn = g.wy(w.in[1] + j), ys = g.wy(w.is[1] + j);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int xn = g.wx(w.in[0] + i), xs = g.wx(w.is[0] + i);
        if (E) {
          Ep[0] += Ex[g.node(xs, yn, zn)] * (w.wn[2][k] * w.wn[1][j] * w.ws[0][i]);
          Ep[1] += Ey[g.node(xn, ys, zn)] * (w.wn[2][k] * w.ws[1][j] * w.wn[0][i]);
          Ep[2] += Ez[g.node(xn, yn, zs)] * (w.ws[2][k] * w.wn[1][j] * w.wn[0][i]);
        }
        Bp[0] += Bx[g.node(xn, ys, zs)] * (w.ws[2][k] * w.ws[1][j] * w.wn[0][i]);
        Bp[1] += By[g.node(xs, yn, zs)] * (w.ws[2][k] * w.wn[1][j] * w.ws[0][i]);
        Bp[2] += Bz[g.node(xs, ys, zn)] * (w.wn[2][k] * w.ws[1][j] * w.ws[0][i]);
      }
    }
  }
}

// BorisPush::update_vEB (src/algorithms/boris_push.cpp:48-57), same operation order; the three divisions by the same
// denominator are one reciprocal and three products (one rounding apart from the reference's quotient: an fp64 division
// is ~15 instructions, and the kernels this sits in are bound by instruction issue)
__device__ inline void update_vEB(double dt, double qm, const double* E, const double* B, double* v)
{
  const double alpha = dt * qm;
  const double a[3] = {alpha * E[0], alpha * E[1], alpha * E[2]};
  const double b[3] = {-alpha * B[0], -alpha * B[1], -alpha * B[2]};
  const double w[3] = {v[0] + 0.5 * a[0], v[1] + 0.5 * a[1], v[2] + 0.5 * a[2]};
  // Vector3::cross (src/utils/vector3.h:217-224)
  const double bw[3] = {+(b[1] * w[2] - b[2] * w[1]), -(b[0] * w[2] - b[2] * w[0]), +(b[0] * w[1] - b[1] * w[0])};
  const double bbw[3] = {+(b[1] * bw[2] - b[2] * bw[1]), -(b[0] * bw[2] - b[2] * bw[0]), +(b[0] * bw[1] - b[1] * bw[0])};
  const double den = 1.0 + 0.25 * (b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
  const double rden = 1.0 / den;
#pragma unroll
  for (int c = 0; c < 3; ++c) v[c] += a[c] + (bw[c] + 0.5 * bbw[c]) * rden;
}


}  // namespace xpic
