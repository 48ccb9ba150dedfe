/*
 * xpic_oracle.cpp -- CPU restatement of xpic's per-timestep hot path (see xpic_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under xpic_amd/ may use this file.
 * Every function cites the reference file:line it follows (paths relative to the reference
 * checkout).  Container layout and loop structure follow the reference on purpose (per-cell
 * std::list<Point>, OpenMP dynamic(16) over cells, omp atomic grid adds), so that the same code
 * doubles as the "port" CPU baseline timed by bench.py.
 */
#include "xpic_oracle.h"

#include <omp.h>

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <list>
#include <random>
#include <string>
#include <vector>

namespace {

constexpr int X = 0, Y = 1, Z = 2;

/* ------------------------------------------------------------------------------------------
 * Vector3R  (src/utils/vector3.h:9-271) -- only the operations the hot path uses, with the same
 * association order as the reference's operators.
 * ---------------------------------------------------------------------------------------- */
struct V3 {
  double d[3];
  V3() : d{0, 0, 0} {}
  V3(double x, double y, double z) : d{x, y, z} {}
  explicit V3(const double* p) : d{p[0], p[1], p[2]} {}
  double& operator[](int i) { return d[i]; }
  const double& operator[](int i) const { return d[i]; }
  V3& operator+=(const V3& o) { d[0] += o[0]; d[1] += o[1]; d[2] += o[2]; return *this; }
  V3& operator*=(double s) { d[0] *= s; d[1] *= s; d[2] *= s; return *this; }
  V3& operator/=(double s) { d[0] /= s; d[1] /= s; d[2] /= s; return *this; }
  V3 operator+(const V3& o) const { return V3(d[0] + o[0], d[1] + o[1], d[2] + o[2]); }
  V3 operator-(const V3& o) const { return V3(d[0] - o[0], d[1] - o[1], d[2] - o[2]); }
  V3 operator/(double s) const { return V3(d[0] / s, d[1] / s, d[2] / s); }
  double dot(const V3& o) const { return d[0] * o[0] + d[1] * o[1] + d[2] * o[2]; }
  double squared() const { return dot(*this); }
  double length() const { return std::hypot(d[0], d[1], d[2]); } /* vector3.h:169-173 */
  V3 normalized() const /* vector3.h:159-167 */
  {
    double l = length();
    if (l > 0) return *this / l;
    return V3();
  }
  V3 cross(const V3& o) const /* vector3.h:217-224 */
  {
    return V3(+(d[1] * o[2] - d[2] * o[1]), -(d[0] * o[2] - d[2] * o[0]), +(d[0] * o[1] - d[1] * o[0]));
  }
  V3 parallel_to(const V3& ref) const; /* vector3.h:199-203 */
  V3 transverse_to(const V3& ref) const { return *this - parallel_to(ref); }
};
inline V3 operator*(const V3& v, double s) { return V3(v[0] * s, v[1] * s, v[2] * s); }
inline V3 operator*(double s, const V3& v) { return v * s; }
V3 V3::parallel_to(const V3& ref) const { return ((*this).dot(ref) * ref) / ref.squared(); }

struct Point { /* src/interfaces/point.h:7-35 */
  V3 r, p;
};

/* ------------------------------------------------------------------------------------------
 * BorisPush  (src/algorithms/boris_push.cpp)
 * ---------------------------------------------------------------------------------------- */
inline void update_r(double dt, Point& pt) /* boris_push.cpp:19-22 */
{
  pt.r += pt.p * dt;
}

inline void update_vEB(double dt, double qm, const V3& E_p, const V3& B_p, Point& pt) /* :48-57 */
{
  double alpha = dt * qm;
  V3 a = +alpha * E_p;
  V3 b = -alpha * B_p;
  V3& v = pt.p;
  V3 w = v + 0.5 * a;
  v += a + (b.cross(w) + 0.5 * b.cross(b.cross(w))) / (1.0 + 0.25 * b.squared());
}

inline double get_theta(double dt, double qm, const V3& B_p) /* :60-63 */
{
  return (-1.0) * qm * B_p.length() * dt;
}

inline void update_v_impl(V3& v, const V3& B_p, double first, double second) /* :85-91 */
{
  V3 b = B_p.normalized();
  V3 v_p = v.parallel_to(b);
  V3 v_t = v.transverse_to(b);
  v = v_p + second * v_t + first * b.cross(v_t);
}

void update_vX(char kind, double dt, double qm, const V3& B_p, Point& pt) /* :24-46,65-83 */
{
  double theta = get_theta(dt, qm, B_p);
  double s, c;
  switch (kind) {
    case 'M': s = std::sin(theta); c = std::cos(theta); break;
    case 'B': {
      double d = (1.0 + 0.25 * (theta * theta));
      s = theta / d;
      c = (1.0 - 0.25 * (theta * theta)) / d;
      break;
    }
    case '1': s = theta * std::sqrt(1.0 - 0.25 * (theta * theta)); c = 1 - 0.5 * (theta * theta); break;
    default: s = theta; c = std::sqrt(1.0 - (theta * theta)); break; /* '2' */
  }
  update_v_impl(pt.p, B_p, s, c);
}

/* ------------------------------------------------------------------------------------------
 * Splines  (src/interfaces/sort_parameters.cpp:3-78)
 * ---------------------------------------------------------------------------------------- */
double spline0(double s) { s = std::abs(s); return (s <= 0.5) ? 1.0 : 0.0; }
double spline1(double s) { s = std::abs(s); return (s <= 1.0) ? 1.0 - s : 0.0; }
double spline2(double s)
{
  s = std::abs(s);
  if (s <= 0.5) return (0.75 - s * s);
  if (0.5 < s && s < 1.5) return 0.5 * (1.5 - s) * (1.5 - s);
  return 0.0;
}
double spline3(double s)
{
  s = std::abs(s);
  double s2 = s * s, s3 = s * s * s;
  if (s < 1.0) return (4. - 6. * s2 + 3. * s3) / 6.;
  if (1.0 <= s && s < 2.0) return (2. - s) * (2. - s) * (2. - s) / 6.;
  return 0.0;
}
double spline4(double s)
{
  s = std::abs(s);
  double s2 = s * s, s3 = s * s * s, s4 = s * s * s * s;
  if (s <= 0.5) return (115. / 192. - 5. / 8. * s2 + 1. / 4. * s4);
  if (0.5 < s && s <= 1.5) return (55. + 20. * s - 120. * s2 + 80. * s3 - 16. * s4) / 96.;
  if (1.5 < s && s < 2.5) return (5. - 2. * s) * (5. - 2. * s) * (5. - 2. * s) * (5. - 2. * s) / 384.;
  return 0.0;
}
double spline5(double s)
{
  s = std::abs(s);
  double s2 = s * s, s3 = s * s * s, s4 = s * s * s * s, s5 = s * s * s * s * s;
  if (s <= 1.0) return (11. / 20. - 0.5 * s2 + 0.25 * s4 - 1. / 12. * s5);
  if (1.0 < s && s <= 2.0)
    return (17. / 40. + 5. / 8. * s - 7. / 4. * s2 + 5. / 4. * s3 - 3. / 8. * s4 + 1. / 24. * s5);
  if (2.0 < s && s < 3.0) return (3. - s) * (3. - s) * (3. - s) * (3. - s) * (3. - s) / 120.;
  return 0.0;
}
typedef double (*sfunc_t)(double);
sfunc_t spline_of(int order)
{
  static const sfunc_t t[6] = {spline0, spline1, spline2, spline3, spline4, spline5};
  return t[order];
}

/* ------------------------------------------------------------------------------------------
 * Shape  (src/utils/shape.h:21-93, shape.cpp:3-80).  PARTICLES_FORM_FACTOR 2 (constants.h:4):
 * shape_radius 1.5, shape_width 4 (sort_parameters.h:46-47,62-63).
 * ---------------------------------------------------------------------------------------- */
constexpr int shape_width = 4;
constexpr int shc = 6;
enum ShapeType { No = 0, Sh = 1, Old = 2, New = 3 };

struct Shape {
  int start[3], size[3];
  double shape[shape_width * shape_width * shape_width * shc];
  bool overflow = false;

  static int i_p(int i, int t, int c) { return i * shc + ((t % 2) * 3 + c); }
  int s_p(int x, int y, int z) const { return (z * size[Y] + y) * size[X] + x; }
  double operator()(int i, int t, int c) const { return shape[i_p(i, t, c)]; }
  int elements() const { return size[0] * size[1] * size[2]; }

  static void make_start(const V3& p_r, double radius, int* out) /* shape.cpp:12-19 */
  {
    for (int c = 0; c < 3; ++c) out[c] = static_cast<int>(std::round(p_r[c] - radius));
  }
  static void make_end(const V3& p_r, double radius, int* out) /* shape.cpp:21-28 */
  {
    for (int c = 0; c < 3; ++c) out[c] = static_cast<int>(std::floor(p_r[c] + radius)) + 1;
  }

  void setup(const double* d3, const V3& r, double radius, sfunc_t sfunc) /* shape.cpp:31-41 */
  {
    V3 p_r(r[X] / d3[X], r[Y] / d3[Y], r[Z] / d3[Z]);
    make_start(p_r, radius, start);
    make_end(p_r, radius, size);
    for (int c = 0; c < 3; ++c) size[c] -= start[c];
    fill(p_r, p_r, No, Sh, sfunc);
  }

  void setup(const double* d3, const V3& old_r, const V3& new_r, double radius, sfunc_t sfunc) /* :43-54 */
  {
    V3 o(old_r[X] / d3[X], old_r[Y] / d3[Y], old_r[Z] / d3[Z]);
    V3 n(new_r[X] / d3[X], new_r[Y] / d3[Y], new_r[Z] / d3[Z]);
    V3 mn(std::min(o[X], n[X]), std::min(o[Y], n[Y]), std::min(o[Z], n[Z]));
    V3 mx(std::max(o[X], n[X]), std::max(o[Y], n[Y]), std::max(o[Z], n[Z]));
    make_start(mn, radius, start);
    make_end(mx, radius, size);
    for (int c = 0; c < 3; ++c) size[c] -= start[c];
    fill(o, n, Old, New, sfunc);
  }

  void fill(const V3& p_r1, const V3& p_r2, int t1, int t2, sfunc_t sfunc) /* shape.cpp:57-80 */
  {
    overflow = size[X] > shape_width || size[Y] > shape_width || size[Z] > shape_width;
    if (overflow) return; /* the reference would write past shape[384] here (shape.h:18,91-92) */
    for (int i = 0; i < elements(); ++i) {
      double g_x = static_cast<double>(start[X] + i % size[X]);
      double g_y = static_cast<double>(start[Y] + (i / size[X]) % size[Y]);
      double g_z = static_cast<double>(start[Z] + (i / size[X]) / size[Y]);
      shape[i_p(i, t1, X)] = sfunc(p_r1[X] - g_x);
      shape[i_p(i, t1, Y)] = sfunc(p_r1[Y] - g_y);
      shape[i_p(i, t1, Z)] = sfunc(p_r1[Z] - g_z);
      if (t2 == Sh) {
        g_x += 0.5;
        g_y += 0.5;
        g_z += 0.5;
      }
      shape[i_p(i, t2, X)] = sfunc(p_r2[X] - g_x);
      shape[i_p(i, t2, Y)] = sfunc(p_r2[Y] - g_y);
      shape[i_p(i, t2, Z)] = sfunc(p_r2[Z] - g_z);
    }
  }

  V3 electric(int i) const /* shape.h:54-61 */
  {
    return V3(shape[i_p(i, No, Z)] * shape[i_p(i, No, Y)] * shape[i_p(i, Sh, X)],
      shape[i_p(i, No, Z)] * shape[i_p(i, Sh, Y)] * shape[i_p(i, No, X)],
      shape[i_p(i, Sh, Z)] * shape[i_p(i, No, Y)] * shape[i_p(i, No, X)]);
  }
  V3 magnetic(int i) const /* shape.h:65-72 */
  {
    return V3(shape[i_p(i, Sh, Z)] * shape[i_p(i, Sh, Y)] * shape[i_p(i, No, X)],
      shape[i_p(i, Sh, Z)] * shape[i_p(i, No, Y)] * shape[i_p(i, Sh, X)],
      shape[i_p(i, No, Z)] * shape[i_p(i, Sh, Y)] * shape[i_p(i, Sh, X)]);
  }
};

/* ------------------------------------------------------------------------------------------
 * Ghosted local array = what DMDAVecGetArray hands out for a local Vec of the reference's DMDA:
 * box stencil of width st = 4 (src/utils/world.h:24-25, world.cpp:36), periodic in x,y,z.
 * ---------------------------------------------------------------------------------------- */
constexpr int ST = 4;

struct Grid {
  int n[3];
  double d[3], L[3];
  long N;   /* cells */
  int g[3]; /* ghosted sizes */
  long G;
  void set(int nx, int ny, int nz, double dx, double dy, double dz)
  {
    n[0] = nx; n[1] = ny; n[2] = nz;
    d[0] = dx; d[1] = dy; d[2] = dz;
    for (int c = 0; c < 3; ++c) {
      L[c] = n[c] * d[c]; /* World::set_geometry(int...) world.cpp:80-91: geom = n * dx */
      g[c] = n[c] + 2 * ST;
    }
    N = (long)nx * ny * nz;
    G = (long)g[0] * g[1] * g[2];
  }
  static int wrap(int i, int n)
  {
    i %= n;
    return i < 0 ? i + n : i;
  }
  /* natural (global) index, vector field */
  long vg(int x, int y, int z, int c) const { return (((long)z * n[1] + y) * n[0] + x) * 3 + c; }
  long vgw(int x, int y, int z, int c) const { return vg(wrap(x, n[0]), wrap(y, n[1]), wrap(z, n[2]), c); }
  long sg(int x, int y, int z) const { return ((long)z * n[1] + y) * n[0] + x; }
  long sgw(int x, int y, int z) const { return sg(wrap(x, n[0]), wrap(y, n[1]), wrap(z, n[2])); }
  /* ghosted local index: x in [-ST, n+ST) */
  long vl(int x, int y, int z, int c) const
  {
    return (((long)(z + ST) * g[1] + (y + ST)) * g[0] + (x + ST)) * 3 + c;
  }
  long sl(int x, int y, int z) const { return ((long)(z + ST) * g[1] + (y + ST)) * g[0] + (x + ST); }
  bool inside_ghost(int x, int y, int z) const
  {
    return x >= -ST && x < n[0] + ST && y >= -ST && y < n[1] + ST && z >= -ST && z < n[2] + ST;
  }
};

/* DMGlobalToLocal(INSERT_VALUES) on the periodic DMDA */
void global_to_local(const Grid& gr, const std::vector<double>& glob, std::vector<double>& loc, int dof)
{
  loc.resize(gr.G * dof);
#pragma omp parallel for collapse(2)
  for (int z = -ST; z < gr.n[2] + ST; ++z)
    for (int y = -ST; y < gr.n[1] + ST; ++y)
      for (int x = -ST; x < gr.n[0] + ST; ++x) {
        long l = gr.sl(x, y, z), g = gr.sgw(x, y, z);
        for (int c = 0; c < dof; ++c) loc[l * dof + c] = glob[g * dof + c];
      }
}

/* DMLocalToGlobal(ADD_VALUES) on the periodic DMDA */
void local_to_global_add(const Grid& gr, const std::vector<double>& loc, std::vector<double>& glob, int dof)
{
  for (int z = -ST; z < gr.n[2] + ST; ++z)
    for (int y = -ST; y < gr.n[1] + ST; ++y)
      for (int x = -ST; x < gr.n[0] + ST; ++x) {
        long l = gr.sl(x, y, z), g = gr.sgw(x, y, z);
        for (int c = 0; c < dof; ++c) glob[g * dof + c] += loc[l * dof + c];
      }
}

/* ------------------------------------------------------------------------------------------
 * SimpleInterpolation::process  (src/algorithms/simple_interpolation.cpp:8-38)
 * E_g/B_g are ghosted local arrays; either may be null.
 * ---------------------------------------------------------------------------------------- */
inline void simple_interpolation(const Grid& gr, const Shape& shape, const double* E_g, const double* B_g,
  V3& E_p, V3& B_p)
{
  for (int i = 0; i < shape.elements(); ++i) {
    int g_x = shape.start[X] + i % shape.size[X];
    int g_y = shape.start[Y] + (i / shape.size[X]) % shape.size[Y];
    int g_z = shape.start[Z] + (i / shape.size[X]) / shape.size[Y];
    long l = gr.vl(g_x, g_y, g_z, 0);
    if (E_g) {
      V3 s = shape.electric(i);
      E_p[X] += E_g[l + X] * s[X];
      E_p[Y] += E_g[l + Y] * s[Y];
      E_p[Z] += E_g[l + Z] * s[Z];
    }
    if (B_g) {
      V3 s = shape.magnetic(i);
      B_p[X] += B_g[l + X] * s[X];
      B_p[Y] += B_g[l + Y] * s[Y];
      B_p[Z] += B_g[l + Z] * s[Z];
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * EsirkepovDecomposition::process / get_jx,jy,jz  (src/algorithms/esirkepov_decomposition.cpp:20-103)
 * J is a ghosted local array, adds are `omp atomic update` as in the reference (:44-51).
 * ---------------------------------------------------------------------------------------- */
inline void esirkepov_process(const Grid& gr, const Shape& shape, double alpha, double* J)
{
  constexpr int j_geom = shape_width * shape_width;
  double temp_j[3 * j_geom];
  std::fill_n(temp_j, 3 * j_geom, 0.0);
  double* temp_jx = temp_j + j_geom * X;
  double* temp_jy = temp_j + j_geom * Y;
  double* temp_jz = temp_j + j_geom * Z;

  for (int i = 0; i < shape.elements(); ++i) {
    int x = i % shape.size[X];
    int y = (i / shape.size[X]) % shape.size[Y];
    int z = (i / shape.size[X]) / shape.size[Y];
    int g_x = shape.start[X] + x, g_y = shape.start[Y] + y, g_z = shape.start[Z] + z;
    int s = shape.s_p(x, y, z);

    /* get_jx :57-71 */
    double qx = alpha * gr.d[X];
    int jx = z * shape_width + y;
    double wx_p = -qx * (shape(s, New, X) - shape(s, Old, X)) *
      (shape(s, New, Y) * (2.0 * shape(s, New, Z) + shape(s, Old, Z)) +
        shape(s, Old, Y) * (2.0 * shape(s, Old, Z) + shape(s, New, Z)));
    temp_jx[jx] = (static_cast<double>(x > 0) * temp_jx[jx]) + wx_p;

    /* get_jy :73-87 */
    double qy = alpha * gr.d[Y];
    int jy = z * shape_width + x;
    double wy_p = -qy * (shape(s, New, Y) - shape(s, Old, Y)) *
      (shape(s, New, X) * (2.0 * shape(s, New, Z) + shape(s, Old, Z)) +
        shape(s, Old, X) * (2.0 * shape(s, Old, Z) + shape(s, New, Z)));
    temp_jy[jy] = (static_cast<double>(y > 0) * temp_jy[jy]) + wy_p;

    /* get_jz :89-103 */
    double qz = alpha * gr.d[Z];
    int jz = y * shape_width + x;
    double wz_p = -qz * (shape(s, New, Z) - shape(s, Old, Z)) *
      (shape(s, New, Y) * (2.0 * shape(s, New, X) + shape(s, Old, X)) +
        shape(s, Old, Y) * (2.0 * shape(s, Old, X) + shape(s, New, X)));
    temp_jz[jz] = (static_cast<double>(z > 0) * temp_jz[jz]) + wz_p;

    long l = gr.vl(g_x, g_y, g_z, 0);
#pragma omp atomic update
    J[l + X] += temp_jx[jx];
#pragma omp atomic update
    J[l + Y] += temp_jy[jy];
#pragma omp atomic update
    J[l + Z] += temp_jz[jz];
  }
}

/* ------------------------------------------------------------------------------------------
 * interpolate_E_s1 / interpolate_B_s1  (src/impls/ecsim/simulation.cpp:8-62, 64-118)
 * ---------------------------------------------------------------------------------------- */
struct W1 { /* the "weights calculator" block repeated at simulation.cpp:12-45, particles.cpp:73-105 */
  int ixn, iyn, izn, ixs, iys, izs;
  double wnx[2], wny[2], wnz[2], wsx[2], wsy[2], wsz[2];
  W1(const Grid& gr, const V3& r)
  {
    double xn = r[X] / gr.d[X], yn = r[Y] / gr.d[Y], zn = r[Z] / gr.d[Z];
    double xs = xn - 0.5, ys = yn - 0.5, zs = zn - 0.5;
    ixn = (int)std::floor(xn); iyn = (int)std::floor(yn); izn = (int)std::floor(zn);
    ixs = (int)std::floor(xs); iys = (int)std::floor(ys); izs = (int)std::floor(zs);
    wnx[1] = (xn - ixn); wny[1] = (yn - iyn); wnz[1] = (zn - izn);
    wnx[0] = 1 - wnx[1]; wny[0] = 1 - wny[1]; wnz[0] = 1 - wnz[1];
    wsx[1] = (xs - ixs); wsy[1] = (ys - iys); wsz[1] = (zs - izs);
    wsx[0] = 1 - wsx[1]; wsy[0] = 1 - wsy[1]; wsz[0] = 1 - wsz[1];
  }
};

inline V3 interpolate_E_s1(const Grid& gr, const double* E_g, const V3& r)
{
  V3 E_p;
  W1 w(gr, r);
  for (int k = 0; k < 2; ++k)
    for (int j = 0; j < 2; ++j)
      for (int i = 0; i < 2; ++i) {
        double sx = w.wnz[k] * w.wny[j] * w.wsx[i];
        double sy = w.wnz[k] * w.wsy[j] * w.wnx[i];
        double sz = w.wsz[k] * w.wny[j] * w.wnx[i];
        E_p[X] += E_g[gr.vl(w.ixs + i, w.iyn + j, w.izn + k, X)] * sx;
        E_p[Y] += E_g[gr.vl(w.ixn + i, w.iys + j, w.izn + k, Y)] * sy;
        E_p[Z] += E_g[gr.vl(w.ixn + i, w.iyn + j, w.izs + k, Z)] * sz;
      }
  return E_p;
}

inline V3 interpolate_B_s1(const Grid& gr, const double* B_g, const V3& r)
{
  V3 B_p;
  W1 w(gr, r);
  for (int k = 0; k < 2; ++k)
    for (int j = 0; j < 2; ++j)
      for (int i = 0; i < 2; ++i) {
        double sx = w.wsz[k] * w.wsy[j] * w.wnx[i];
        double sy = w.wsz[k] * w.wny[j] * w.wsx[i];
        double sz = w.wnz[k] * w.wsy[j] * w.wsx[i];
        B_p[X] += B_g[gr.vl(w.ixn + i, w.iys + j, w.izs + k, X)] * sx;
        B_p[Y] += B_g[gr.vl(w.ixs + i, w.iyn + j, w.izs + k, Y)] * sy;
        B_p[Z] += B_g[gr.vl(w.ixs + i, w.iys + j, w.izn + k, Z)] * sz;
      }
  return B_p;
}

/* ------------------------------------------------------------------------------------------
 * matL fixed-stencil layout.  The reference keeps matL as a PETSc AIJ matrix assembled from
 * per-cell COO blocks (src/impls/ecsim/simulation.cpp:336-469).  Row (node, c1) couples to
 *   c2 == c1 : offsets [-1,1]^3                                  (27)
 *   c2 != c1 : axis c1 in [-1,2], axis c2 in [-2,1], third [-1,1] (48)
 * which is exactly the set of (vg2 - vg1) that fill_matrix_indices (:370-469) can produce with a
 * possibly non-zero value (|d|=2 same-component pairs are structural zeros of decompose_ecsim_current).
 * ---------------------------------------------------------------------------------------- */
struct LRange { int lo[3], n[3]; };
inline LRange lrange(int c1, int c2)
{
  LRange r;
  for (int a = 0; a < 3; ++a) {
    if (c1 == c2) { r.lo[a] = -1; r.n[a] = 3; }
    else if (a == c1) { r.lo[a] = -1; r.n[a] = 4; }
    else if (a == c2) { r.lo[a] = -2; r.n[a] = 4; }
    else { r.lo[a] = -1; r.n[a] = 3; }
  }
  return r;
}
inline int lblock_offset(int c1, int c2)
{
  int off = 0;
  for (int c = 0; c < c2; ++c) off += (c == c1) ? 27 : 48;
  return off;
}
inline int lencode(int c1, int c2, int dx, int dy, int dz)
{
  LRange r = lrange(c1, c2);
  int i = dx - r.lo[0], j = dy - r.lo[1], k = dz - r.lo[2];
  if (i < 0 || i >= r.n[0] || j < 0 || j >= r.n[1] || k < 0 || k >= r.n[2]) return -1;
  return lblock_offset(c1, c2) + (k * r.n[1] + j) * r.n[0] + i;
}
inline void ldecode(int c1, int kk, int* c2, int* d)
{
  for (int c = 0; c < 3; ++c) {
    int sz = (c == c1) ? 27 : 48;
    if (kk < sz) {
      LRange r = lrange(c1, c);
      *c2 = c;
      d[0] = r.lo[0] + kk % r.n[0];
      d[1] = r.lo[1] + (kk / r.n[0]) % r.n[1];
      d[2] = r.lo[2] + (kk / r.n[0]) / r.n[1];
      return;
    }
    kk -= sz;
  }
}

/* (c1, node offset, c2, node offset) of each of the 1296 per-cell block entries, built with the
 * loops of ecsim::Simulation::fill_matrix_indices (src/impls/ecsim/simulation.cpp:408-464). */
struct BlockEntry { int c1, o1[3], c2, o2[3]; };
const BlockEntry* block_entries()
{
  static BlockEntry tab[1296];
  static bool built = false;
  if (!built) {
    for (int c1 = 0; c1 < 3; ++c1) {
      int si1 = (c1 == 0 ? 3 : 2), sj1 = (c1 == 1 ? 3 : 2), sk1 = (c1 == 2 ? 3 : 2);
      int oi1 = (c1 == 0 ? -1 : 0), oj1 = (c1 == 1 ? -1 : 0), ok1 = (c1 == 2 ? -1 : 0);
      for (int k1 = 0; k1 < sk1; ++k1)
        for (int j1 = 0; j1 < sj1; ++j1)
          for (int i1 = 0; i1 < si1; ++i1)
            for (int c2 = 0; c2 < 3; ++c2) {
              int si2 = (c2 == 0 ? 3 : 2), sj2 = (c2 == 1 ? 3 : 2), sk2 = (c2 == 2 ? 3 : 2);
              int oi2 = (c2 == 0 ? -1 : 0), oj2 = (c2 == 1 ? -1 : 0), ok2 = (c2 == 2 ? -1 : 0);
              for (int k2 = 0; k2 < sk2; ++k2)
                for (int j2 = 0; j2 < sj2; ++j2)
                  for (int i2 = 0; i2 < si2; ++i2) {
                    int i = (k1 * sj1 + j1) * si1 + i1;
                    int j = (k2 * sj2 + j2) * si2 + i2;
                    int ind = (c1 * 3 + c2) * 144 + (i * 12 + j);
                    BlockEntry& e = tab[ind];
                    e.c1 = c1; e.o1[0] = i1 + oi1; e.o1[1] = j1 + oj1; e.o1[2] = k1 + ok1;
                    e.c2 = c2; e.o2[0] = i2 + oi2; e.o2[1] = j2 + oj2; e.o2[2] = k2 + ok2;
                  }
            }
    }
    built = true;
  }
  return tab;
}

/* ------------------------------------------------------------------------------------------
 * the one global RNG (src/utils/random_generator.h:8-35; RANDOM_SEED false, constants.h:5)
 * ---------------------------------------------------------------------------------------- */
std::mt19937& rng()
{
  static std::mt19937 gen;
  return gen;
}
std::uniform_real_distribution<double>& dist01()
{
  static std::uniform_real_distribution<double> d(0.0, 1.0);
  return d;
}
inline double random_01() { return dist01()(rng()); }

constexpr double mec2 = 511.0; /* constants.h:30 */

inline double temperature_momentum(double temperature, double mass) /* particles_load.cpp:52-55 */
{
  return std::sqrt(-2.0 * (temperature * mass / mec2) * std::log(random_01()));
}

struct Sort {
  /* SortParameters (src/interfaces/sort_parameters.h:7-19) */
  int Np;
  double n, q, m, Tx, Ty, Tz;
  std::vector<std::list<Point>> storage; /* particles.h:32 */
  std::vector<double> J, J_loc;          /* basic: J ; ecsim: currI ; ecsimcorr: currJe (particles.cpp ctor) */
  std::vector<double> currI, currI_loc;
  std::vector<double> currJe, currJe_loc;
  std::vector<double> rho; /* ParticlesChargeDensity::field_ */
  double energy = 0, pred_w = 0, corr_w = 0, pred_dK = 0, corr_dK = 0, lambda_dK = 0;
  double q_m() const { return q / m; }             /* particles.cpp:275-278 */
  double n_Np() const { return n / Np; }           /* :280-283 */
  double qn_Np() const { return q * n / Np; }      /* :285-288 */
};

}  // namespace

struct orc_sim {
  int scheme;
  Grid gr;
  double dt;
  std::vector<Sort> sorts;
  std::vector<double> E, B, B0, Ep, Ec, J, currI, currJe;
  std::vector<double> E_loc, B_loc;
  std::vector<double> matL; /* [3N][123] */
  double rtol = 1e-7, atol = 1e-7; /* src/impls/ecsim/simulation.h:15-18 */
  int maxit = 100;
  int last_its[2] = {0, 0};
  double solve_seconds = 0; /* wall time spent inside the Krylov solves of orc_step (bench.py's cpu_baseline) */
  long solve_its = 0;
};

namespace {

long cell_of(const Grid& gr, const V3& r, bool* inside)
{
  /* FLOOR_STEP (utils.h:78) */
  int vx = static_cast<int>(std::floor(r[X] / gr.d[X]));
  int vy = static_cast<int>(std::floor(r[Y] / gr.d[Y]));
  int vz = static_cast<int>(std::floor(r[Z] / gr.d[Z]));
  *inside = (0 <= vx && vx < gr.n[0]) && (0 <= vy && vy < gr.n[1]) && (0 <= vz && vz < gr.n[2]);
  return ((long)vz * gr.n[1] + vy) * gr.n[0] + vx; /* world.s_g, computed before the bounds test as in :96 */
}

/* g_bound_periodic (src/interfaces/point.cpp:18-26) on all three axes
 * (Particles::correct_coordinates(Point&), particles.cpp:329-339) */
inline void correct_coordinates(const Grid& gr, Point& pt)
{
  for (int a = 0; a < 3; ++a) {
    double& s = pt.r[a];
    if (s < 0.0)
      s = gr.L[a] - (0.0 - s);
    else if (s > gr.L[a])
      s = 0.0 + (s - gr.L[a]);
  }
}

/* ---- curl operators: Rotor::fill_stencil + values (operators.cpp:155-215) on the periodic grid */
void rot_apply(const Grid& gr, int sign, double alpha, const double* F, double* out, bool add)
{
  const double ix = 1.0 / gr.d[X], iy = 1.0 / gr.d[Y], iz = 1.0 / gr.d[Z];
  const int nx = gr.n[0], ny = gr.n[1], nz = gr.n[2];
#pragma omp parallel for collapse(2)
  for (int z = 0; z < nz; ++z)
    for (int y = 0; y < ny; ++y)
      for (int x = 0; x < nx; ++x) {
        double rx, ry, rz;
        if (sign > 0) {
          int xp = Grid::wrap(x + 1, nx), yp = Grid::wrap(y + 1, ny), zp = Grid::wrap(z + 1, nz);
          rx = +iy * F[gr.vg(x, yp, z, Z)] - iy * F[gr.vg(x, y, z, Z)] - iz * F[gr.vg(x, y, zp, Y)] + iz * F[gr.vg(x, y, z, Y)];
          ry = -ix * F[gr.vg(xp, y, z, Z)] + ix * F[gr.vg(x, y, z, Z)] + iz * F[gr.vg(x, y, zp, X)] - iz * F[gr.vg(x, y, z, X)];
          rz = +ix * F[gr.vg(xp, y, z, Y)] - ix * F[gr.vg(x, y, z, Y)] - iy * F[gr.vg(x, yp, z, X)] + iy * F[gr.vg(x, y, z, X)];
        }
        else {
          int xm = Grid::wrap(x - 1, nx), ym = Grid::wrap(y - 1, ny), zm = Grid::wrap(z - 1, nz);
          rx = +iy * F[gr.vg(x, y, z, Z)] - iy * F[gr.vg(x, ym, z, Z)] - iz * F[gr.vg(x, y, z, Y)] + iz * F[gr.vg(x, y, zm, Y)];
          ry = -ix * F[gr.vg(x, y, z, Z)] + ix * F[gr.vg(xm, y, z, Z)] + iz * F[gr.vg(x, y, z, X)] - iz * F[gr.vg(x, y, zm, X)];
          rz = +ix * F[gr.vg(x, y, z, Y)] - ix * F[gr.vg(xm, y, z, Y)] - iy * F[gr.vg(x, y, z, X)] + iy * F[gr.vg(x, ym, z, X)];
        }
        long o = gr.vg(x, y, z, 0);
        if (add) { out[o + X] += alpha * rx; out[o + Y] += alpha * ry; out[o + Z] += alpha * rz; }
        else { out[o + X] = alpha * rx; out[o + Y] = alpha * ry; out[o + Z] = alpha * rz; }
      }
}

/* matM = 2 I + 0.5 dt^2 rotB rotE (unscaled rotors)  (ecsim/simulation.cpp:544-551) */
void matM_apply(const orc_sim* s, const double* x, double* y)
{
  std::vector<double> t(s->gr.N * 3);
  rot_apply(s->gr, +1, 1.0, x, t.data(), false);
  rot_apply(s->gr, -1, 0.5 * s->dt * s->dt, t.data(), y, false);
  long n = s->gr.N * 3;
#pragma omp parallel for
  for (long i = 0; i < n; ++i) y[i] += 2.0 * x[i];
}

void matL_apply(const orc_sim* s, const double* x, double* y, bool add)
{
  const Grid& gr = s->gr;
  /* matL before its first assembly is the zero matrix (MatZeroEntries, simulation.cpp:164): an apply or a solve on a
   * simulation whose fill_ecsim_current never ran must not index the still empty value array -- every thread of the
   * team below would fault at once */
  if (s->matL.size() != (size_t)gr.N * 3 * ORC_LSTENCIL) {
    if (!add) std::fill(y, y + gr.N * 3, 0.0);
    return;
  }
  static int dec[3][ORC_LSTENCIL][4];
  static bool built = false;
  if (!built) {
    for (int c1 = 0; c1 < 3; ++c1)
      for (int k = 0; k < ORC_LSTENCIL; ++k) ldecode(c1, k, &dec[c1][k][0], &dec[c1][k][1]);
    built = true;
  }
#pragma omp parallel for collapse(2)
  for (int z = 0; z < gr.n[2]; ++z)
    for (int y0 = 0; y0 < gr.n[1]; ++y0)
      for (int x0 = 0; x0 < gr.n[0]; ++x0)
        for (int c1 = 0; c1 < 3; ++c1) {
          long row = gr.vg(x0, y0, z, c1);
          const double* Lr = &s->matL[row * ORC_LSTENCIL];
          double acc = 0.0;
          for (int k = 0; k < ORC_LSTENCIL; ++k) {
            const int* e = dec[c1][k];
            acc += Lr[k] * x[gr.vgw(x0 + e[1], y0 + e[2], z + e[3], e[0])];
          }
          if (add) y[row] += acc; else y[row] = acc;
        }
}

void matA_apply(const orc_sim* s, const double* x, double* y)
{
  matM_apply(s, x, y);
  matL_apply(s, x, y, true);
}

/* ---- BLAS-1 helpers */
double vdot(long n, const double* a, const double* b)
{
  double s = 0;
#pragma omp parallel for reduction(+ : s)
  for (long i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}
void vaxpy(long n, double a, const double* x, double* y)
{
#pragma omp parallel for
  for (long i = 0; i < n; ++i) y[i] += a * x[i];
}

typedef void (*apply_t)(const orc_sim*, const double*, double*);

/* Restarted GMRES(30), classical Gram-Schmidt, zero initial guess, no preconditioner.
 * Mirrors what KSPSolve does for the reference (PETSc default KSPGMRES, restart 30,
 * KSPConvergedDefault: rnorm <= max(rtol*||b||, atol); ecsim/simulation.cpp:558-567) except for
 * the ILU(0) preconditioner, which PETSc (absent from the reference checkout) supplies. */
int gmres(const orc_sim* s, apply_t A, const double* b, double* x, double rtol, double atol, int maxit,
  double* final_rnorm)
{
  const int m = 30;
  const long n = s->gr.N * 3;
  std::vector<std::vector<double>> V(m + 1, std::vector<double>(n));
  std::vector<double> H((m + 1) * m, 0.0), cs(m), sn(m), g(m + 1), w(n), r(n);
  std::fill(x, x + n, 0.0);
  double bnorm = std::sqrt(vdot(n, b, b));
  double tol = std::max(rtol * bnorm, atol);
  int its = 0;
  std::copy(b, b + n, r.begin());
  double rnorm = bnorm;
  if (final_rnorm) *final_rnorm = rnorm;
  if (rnorm <= tol) return 0;
  while (its < maxit) {
    for (long i = 0; i < n; ++i) V[0][i] = r[i] / rnorm;
    std::fill(g.begin(), g.end(), 0.0);
    g[0] = rnorm;
    int j = 0;
    for (; j < m && its < maxit; ++j) {
      A(s, V[j].data(), w.data());
      std::vector<double> h(j + 2, 0.0);
      for (int i = 0; i <= j; ++i) h[i] = vdot(n, w.data(), V[i].data());
      for (int i = 0; i <= j; ++i) vaxpy(n, -h[i], V[i].data(), w.data());
      h[j + 1] = std::sqrt(vdot(n, w.data(), w.data()));
      if (h[j + 1] != 0.0)
        for (long i = 0; i < n; ++i) V[j + 1][i] = w[i] / h[j + 1];
      for (int i = 0; i < j; ++i) {
        double t = cs[i] * h[i] + sn[i] * h[i + 1];
        h[i + 1] = -sn[i] * h[i] + cs[i] * h[i + 1];
        h[i] = t;
      }
      double den = std::hypot(h[j], h[j + 1]);
      cs[j] = h[j] / den;
      sn[j] = h[j + 1] / den;
      h[j] = den;
      g[j + 1] = -sn[j] * g[j];
      g[j] = cs[j] * g[j];
      for (int i = 0; i <= j; ++i) H[i * m + j] = h[i];
      ++its;
      rnorm = std::abs(g[j + 1]);
      if (rnorm <= tol) { ++j; break; }
    }
    /* back substitution, x += V y */
    std::vector<double> yv(j);
    for (int i = j - 1; i >= 0; --i) {
      double t = g[i];
      for (int k = i + 1; k < j; ++k) t -= H[i * m + k] * yv[k];
      yv[i] = t / H[i * m + i];
    }
    for (int i = 0; i < j; ++i) vaxpy(n, yv[i], V[i].data(), x);
    if (final_rnorm) *final_rnorm = rnorm;
    if (rnorm <= tol) return its;
    A(s, x, w.data());
    for (long i = 0; i < n; ++i) r[i] = b[i] - w[i];
    rnorm = std::sqrt(vdot(n, r.data(), r.data()));
  }
  if (final_rnorm) *final_rnorm = rnorm;
  return rnorm <= tol ? its : -its - 1;
}

int cg(const orc_sim* s, apply_t A, const double* b, double* x, double rtol, double atol, int maxit,
  double* final_rnorm)
{
  const long n = s->gr.N * 3;
  std::vector<double> r(b, b + n), p(b, b + n), Ap(n);
  std::fill(x, x + n, 0.0);
  double rr = vdot(n, r.data(), r.data());
  double tol = std::max(rtol * std::sqrt(rr), atol);
  int its = 0;
  while (std::sqrt(rr) > tol && its < maxit) {
    A(s, p.data(), Ap.data());
    double alpha = rr / vdot(n, p.data(), Ap.data());
    vaxpy(n, alpha, p.data(), x);
    vaxpy(n, -alpha, Ap.data(), r.data());
    double rr1 = vdot(n, r.data(), r.data());
    double beta = rr1 / rr;
    rr = rr1;
#pragma omp parallel for
    for (long i = 0; i < n; ++i) p[i] = r[i] + beta * p[i];
    ++its;
  }
  if (final_rnorm) *final_rnorm = std::sqrt(rr);
  return std::sqrt(rr) <= tol ? its : -its - 1;
}

/* ------------------------------------------------------------------------------------------
 * interfaces::Particles::update_cells_seq  (src/interfaces/particles.cpp:79-116)
 * ---------------------------------------------------------------------------------------- */
void update_cells_seq(const Grid& gr, Sort& sort)
{
  for (long g = 0; g < gr.N; ++g) {
    auto it = sort.storage[g].begin();
    while (it != sort.storage[g].end()) {
      correct_coordinates(gr, *it);
      bool inside;
      long ng = cell_of(gr, it->r, &inside);
      if (ng == g) {
        it = std::next(it);
        continue;
      }
      if (inside) sort.storage[ng].emplace_back(std::move(*it));
      it = sort.storage[g].erase(it);
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * basic::Particles::push  (src/impls/basic/particles.cpp:17-53)
 * ---------------------------------------------------------------------------------------- */
int basic_push_sort(orc_sim* s, Sort& sort)
{
  const Grid& gr = s->gr;
  const double dt = s->dt;
  double* J_arr = sort.J_loc.data();
  const double* E_arr = s->E_loc.data();
  const double* B_arr = s->B_loc.data();
  int bad = 0;
  const long ncell = (long)sort.storage.size();
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : bad)
  for (long g = 0; g < ncell; ++g) {
    for (auto& point : sort.storage[g]) {
      const V3 old_r = point.r;
      update_r(dt / 2.0, point);
      Shape shape;
      shape.setup(gr.d, point.r, 1.5, spline2);
      V3 E_p, B_p;
      simple_interpolation(gr, shape, E_arr, B_arr, E_p, B_p);
      update_vEB(dt, sort.q / sort.m, E_p, B_p, point);
      update_r(dt / 2.0, point);
      shape.setup(gr.d, old_r, point.r, 1.5, spline2);
      if (shape.overflow) { ++bad; continue; }
      esirkepov_process(gr, shape, sort.qn_Np() / (6.0 * dt), J_arr);
    }
  }
  local_to_global_add(gr, sort.J_loc, sort.J, 3); /* DMLocalToGlobal ADD :50 */
  vaxpy(gr.N * 3, 1.0, sort.J.data(), s->J.data()); /* VecAXPY :51 */
  return bad;
}

/* ------------------------------------------------------------------------------------------
 * ecsim::Particles::decompose_ecsim_current  (src/impls/ecsim/particles.cpp:62-173)
 * ---------------------------------------------------------------------------------------- */
void decompose_ecsim_current(const orc_sim* s, const Sort& sort, const Point& point, double* currI_arr,
  double* coo_v)
{
  const Grid& gr = s->gr;
  const double dt = s->dt;
  const V3& r = point.r;
  const V3& v = point.p;
  double q = sort.q, m = sort.m;
  double mpw = sort.n / (double)sort.Np;

  W1 w(gr, r);
  int ox = w.ixs - w.ixn + 1, oy = w.iys - w.iyn + 1, oz = w.izs - w.izn + 1;

  V3 b = interpolate_B_s1(gr, s->B_loc.data(), r) * ((0.5 * dt) * q / m);
  V3 I_p = q * mpw / (1. + b.squared()) * (v + v.cross(b) + v.dot(b) * b);
  double A_p = 0.5 * dt * dt * mpw * q * q / m / (1 + b.squared());

  const double matB[3][3]{
    {1.0 + b[X] * b[X], +b[Z] + b[X] * b[Y], -b[Y] + b[X] * b[Z]},
    {-b[Z] + b[Y] * b[X], 1.0 + b[Y] * b[Y], +b[X] + b[Y] * b[Z]},
    {+b[Y] + b[Z] * b[X], -b[X] + b[Z] * b[Y], 1.0 + b[Z] * b[Z]},
  };

  int i[3], j[3];
  double s1[3], s2[3];
  for (int k1 = 0; k1 < 2; ++k1)
    for (int j1 = 0; j1 < 2; ++j1)
      for (int i1 = 0; i1 < 2; ++i1) {
        int in = w.ixn + i1, jn = w.iyn + j1, kn = w.izn + k1;
        int is = w.ixs + i1, js = w.iys + j1, ks = w.izs + k1;

        s1[X] = w.wnz[k1] * w.wny[j1] * w.wsx[i1];
        s1[Y] = w.wnz[k1] * w.wsy[j1] * w.wnx[i1];
        s1[Z] = w.wsz[k1] * w.wny[j1] * w.wnx[i1];

#pragma omp atomic update
        currI_arr[gr.vl(is, jn, kn, X)] += s1[X] * I_p[X];
#pragma omp atomic update
        currI_arr[gr.vl(in, js, kn, Y)] += s1[Y] * I_p[Y];
#pragma omp atomic update
        currI_arr[gr.vl(in, jn, ks, Z)] += s1[Z] * I_p[Z];

        i[X] = (k1 * 2 + j1) * 3 + (ox + i1);
        i[Y] = (k1 * 3 + (oy + j1)) * 2 + i1;
        i[Z] = ((oz + k1) * 2 + j1) * 2 + i1;

        for (int k2 = 0; k2 < 2; ++k2)
          for (int j2 = 0; j2 < 2; ++j2)
            for (int i2 = 0; i2 < 2; ++i2) {
              s2[X] = w.wsx[i2] * w.wny[j2] * w.wnz[k2];
              s2[Y] = w.wnx[i2] * w.wsy[j2] * w.wnz[k2];
              s2[Z] = w.wnx[i2] * w.wny[j2] * w.wsz[k2];

              j[X] = (k2 * 2 + j2) * 3 + (ox + i2);
              j[Y] = (k2 * 3 + (oy + j2)) * 2 + i2;
              j[Z] = ((oz + k2) * 2 + j2) * 2 + i2;

              for (int c1 = 0; c1 < 3; c1++)
                for (int c2 = 0; c2 < 3; c2++) {
                  int ind = (c1 * 3 + c2) * 144 + (i[c1] * 12 + j[c2]);
                  coo_v[ind] += s1[c1] * s2[c2] * A_p * matB[c1][c2];
                }
            }
      }
}

/* ecsim::Simulation::fill_ecsim_current (simulation.cpp:336-368,471-484) +
 * ecsim::Particles::fill_ecsim_current (particles.cpp:33-59).  MatSetValuesCOO(INSERT) sums the
 * duplicate (row,col) pairs of neighbouring cells' blocks (:366); here they are summed straight
 * into the fixed-stencil rows. */
void ecsim_fill_current(orc_sim* s)
{
  const Grid& gr = s->gr;
  const long n3 = gr.N * 3;
  global_to_local(gr, s->B, s->B_loc, 3); /* :474 */
  std::fill(s->currI.begin(), s->currI.end(), 0.0);
  s->matL.assign(n3 * ORC_LSTENCIL, 0.0); /* MatZeroEntries :164 */
  const BlockEntry* tab = block_entries();
  double* L = s->matL.data();

  /* the reference keeps ONE coo_v over all sorts (:363-366); we process cell by cell and sum
   * the sorts inside, which yields the same per-cell block */
  for (auto& sort : s->sorts) {
    std::fill(sort.currI.begin(), sort.currI.end(), 0.0);
    sort.currI_loc.assign(gr.G * 3, 0.0);
  }
#pragma omp parallel for schedule(dynamic, 16)
  for (long g = 0; g < gr.N; ++g) {
    double coo_cv[1296];
    bool any = false;
    for (auto& sort : s->sorts) {
      if (sort.storage[g].empty()) continue;
      if (!any) { std::fill_n(coo_cv, 1296, 0.0); any = true; }
      for (const auto& point : sort.storage[g]) decompose_ecsim_current(s, sort, point, sort.currI_loc.data(), coo_cv);
    }
    if (!any) continue;
    int vgx = (int)(g % gr.n[0]), vgy = (int)((g / gr.n[0]) % gr.n[1]), vgz = (int)((g / gr.n[0]) / gr.n[1]);
    for (int ind = 0; ind < 1296; ++ind) {
      const BlockEntry& e = tab[ind];
      int k = lencode(e.c1, e.c2, e.o2[0] - e.o1[0], e.o2[1] - e.o1[1], e.o2[2] - e.o1[2]);
      if (k < 0) { assert(coo_cv[ind] == 0.0); continue; }
      long row = gr.vgw(vgx + e.o1[0], vgy + e.o1[1], vgz + e.o1[2], e.c1);
#pragma omp atomic update
      L[row * ORC_LSTENCIL + k] += coo_cv[ind];
    }
  }
  for (auto& sort : s->sorts) {
    local_to_global_add(gr, sort.currI_loc, sort.currI, 3); /* particles.cpp:56 */
    vaxpy(n3, 1.0, sort.currI.data(), s->currI.data());       /* :57 */
  }
}

/* ecsim::Simulation::advance_fields(ksp, curr, out)  (simulation.cpp:255-279) */
int advance_fields(orc_sim* s, int op, const std::vector<double>& curr, std::vector<double>& out, int slot)
{
  const long n3 = s->gr.N * 3;
  std::vector<double> rhs(n3), Bm(n3);
  for (long i = 0; i < n3; ++i) Bm[i] = s->B[i] - s->B0[i];             /* VecAXPY(B,-1,B0) :260 */
  for (long i = 0; i < n3; ++i) rhs[i] = 2.0 * s->E[i] - s->dt * curr[i]; /* VecCopy + VecAXPBY :262-263 */
  rot_apply(s->gr, -1, +s->dt, Bm.data(), rhs.data(), true);             /* MatMultAdd(rotB) :264, rotB = +dt rot(-) :554 */
  double rn;
  int its;
  const double t0 = omp_get_wtime();
  if (op == 0) its = gmres(s, matA_apply, rhs.data(), out.data(), s->rtol, s->atol, s->maxit, &rn);
  else if (op == 1) its = gmres(s, matM_apply, rhs.data(), out.data(), s->rtol, s->atol, s->maxit, &rn);
  else its = cg(s, matM_apply, rhs.data(), out.data(), s->rtol, s->atol, s->maxit, &rn);
  s->solve_seconds += omp_get_wtime() - t0;
  if (its > 0) s->solve_its += its;
  s->last_its[slot] = its;
  return its;
}

/* Energy::get_kinetic (src/diagnostics/energy.cpp:187-190) summed as in
 * ecsimcorr::Particles::calculate_energy (src/impls/ecsimcorr/particles.cpp:134-150) */
double calculate_energy(Sort& sort)
{
  double energy = 0.0;
  const double m = sort.m, mpw = sort.n / sort.Np;
  const long ncell = (long)sort.storage.size();
#pragma omp parallel for reduction(+ : energy) schedule(dynamic, 16)
  for (long g = 0; g < ncell; ++g)
    for (auto& point : sort.storage[g]) energy += 0.5 * (m * point.p.squared()) * mpw;
  sort.energy = energy;
  return energy;
}

/* ecsimcorr::Particles::first_push  (src/impls/ecsimcorr/particles.cpp:27-50) */
int ecsimcorr_first_push(orc_sim* s, Sort& sort)
{
  const Grid& gr = s->gr;
  int bad = 0;
  const long ncell = (long)sort.storage.size();
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : bad)
  for (long g = 0; g < ncell; ++g)
    for (auto& point : sort.storage[g]) {
      const V3 old_r = point.r;
      update_r(0.5 * s->dt, point);
      Shape shape;
      shape.setup(gr.d, old_r, point.r, 1.5, spline2);
      if (shape.overflow) { ++bad; continue; }
      esirkepov_process(gr, shape, sort.qn_Np() / (6.0 * s->dt), sort.currJe_loc.data());
    }
  return bad;
}

/* ecsimcorr::Particles::second_push  (src/impls/ecsimcorr/particles.cpp:52-91) */
int ecsimcorr_second_push(orc_sim* s, Sort& sort)
{
  const Grid& gr = s->gr;
  double pred_w = 0.0;
  int bad = 0;
  const long ncell = (long)sort.storage.size();
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : bad)
  for (long g = 0; g < ncell; ++g)
    for (auto& point : sort.storage[g]) {
      const V3 old_r = point.r;
      const V3 old_v = point.p;
      V3 E_p = interpolate_E_s1(gr, s->E_loc.data(), point.r);
      V3 B_p = interpolate_B_s1(gr, s->B_loc.data(), point.r);
      update_vEB(s->dt, sort.q_m(), E_p, B_p, point);
      update_r(0.5 * s->dt, point);
      Shape shape;
      shape.setup(gr.d, old_r, point.r, 1.5, spline2);
      if (shape.overflow) { ++bad; continue; }
      esirkepov_process(gr, shape, sort.qn_Np() / (6.0 * s->dt), sort.currJe_loc.data());
      double dw = sort.qn_Np() * 0.5 * (old_v + point.p).dot(E_p);
#pragma omp atomic update
      pred_w += dw;
    }
  sort.pred_w = pred_w;
  local_to_global_add(gr, sort.currJe_loc, sort.currJe, 3);   /* :88 */
  vaxpy(gr.N * 3, 1.0, sort.currJe.data(), s->currJe.data()); /* :89 */
  return bad;
}

/* ecsimcorr::Particles::final_update  (src/impls/ecsimcorr/particles.cpp:93-126) */
void ecsimcorr_final_update(orc_sim* s, Sort& sort)
{
  sort.corr_w = vdot(s->gr.N * 3, sort.currJe.data(), s->Ec.data());
  double K0 = sort.energy;
  calculate_energy(sort);
  double K = sort.energy;
  double lambda2 = 1.0 + s->dt * (sort.corr_w - sort.pred_w) / K;
  double lambda = std::sqrt(lambda2);
  const long ncell = (long)sort.storage.size();
#pragma omp parallel for schedule(dynamic, 16)
  for (long g = 0; g < ncell; ++g)
    for (auto& point : sort.storage[g]) point.p *= lambda;
  sort.lambda_dK = (lambda2 - 1.0) * K;
  sort.pred_dK = K - K0;
  sort.corr_dK = lambda2 * K - K0;
  sort.energy = lambda2 * K;
}

/* ---- timestep_implementation of the three schemes */

int step_basic(orc_sim* s) /* src/impls/basic/simulation.cpp:30-100 */
{
  const Grid& gr = s->gr;
  const long n3 = gr.N * 3;
  const double dt = s->dt;
  std::fill(s->J.begin(), s->J.end(), 0.0);
  for (auto& sort : s->sorts) {
    std::fill(sort.J.begin(), sort.J.end(), 0.0);
    sort.J_loc.assign(gr.G * 3, 0.0);
  }
  /* push_particles :45-72 ; rotE = -(0.5 dt) rot(+) :23 */
  vaxpy(n3, -1.0, s->B0.data(), s->B.data());
  rot_apply(gr, +1, -(0.5 * dt), s->E.data(), s->B.data(), true);
  vaxpy(n3, +1.0, s->B0.data(), s->B.data());
  int bad = 0;
  if (!s->sorts.empty()) {
    global_to_local(gr, s->E, s->E_loc, 3);
    global_to_local(gr, s->B, s->B_loc, 3);
    for (auto& sort : s->sorts) {
      bad += basic_push_sort(s, sort);
      update_cells_seq(gr, sort);
    }
  }
  /* push_fields :74-100 ; rotB = +dt rot(-) :24 */
  vaxpy(n3, -1.0, s->B0.data(), s->B.data());
  rot_apply(gr, +1, -(0.5 * dt), s->E.data(), s->B.data(), true);
  rot_apply(gr, -1, +dt, s->B.data(), s->E.data(), true);
  vaxpy(n3, -dt, s->J.data(), s->E.data());
  vaxpy(n3, +1.0, s->B0.data(), s->B.data());
  return bad ? -1 : 0;
}

void ecsim_final_update(orc_sim* s) /* src/impls/ecsim/simulation.cpp:241-253 */
{
  const long n3 = s->gr.N * 3;
  for (long i = 0; i < n3; ++i) s->E[i] = 2.0 * s->Ep[i] - s->E[i];       /* VecAXPBY(E, 2, -1, Ep) */
  rot_apply(s->gr, +1, -s->dt, s->Ep.data(), s->B.data(), true);       /* rotE = -dt rot(+) :553 */
}

int step_ecsim(orc_sim* s) /* src/impls/ecsim/simulation.cpp:145-253 */
{
  const Grid& gr = s->gr;
  /* clear_sources :157-172 is folded into ecsim_fill_current (zeroes currI, matL, per-sort currI) */
  for (auto& sort : s->sorts) { /* first_push :174-189 */
    const long ncell = (long)sort.storage.size();
#pragma omp parallel for schedule(dynamic, 16)
    for (long g = 0; g < ncell; ++g)
      for (auto& point : sort.storage[g]) update_r(s->dt, point);
  }
  for (auto& sort : s->sorts) update_cells_seq(gr, sort); /* update_cells_with_assembly :282-334 */
  ecsim_fill_current(s);
  int its = advance_fields(s, 0, s->currI, s->Ep, 0); /* :191-210 */
  if (its < 0) return its;
  /* second_push :212-239 */
  global_to_local(gr, s->Ep, s->E_loc, 3);
  global_to_local(gr, s->B, s->B_loc, 3);
  for (auto& sort : s->sorts) {
    const long ncell = (long)sort.storage.size();
#pragma omp parallel for schedule(dynamic, 16)
    for (long g = 0; g < ncell; ++g)
      for (auto& point : sort.storage[g]) {
        V3 E_p = interpolate_E_s1(gr, s->E_loc.data(), point.r);
        V3 B_p = interpolate_B_s1(gr, s->B_loc.data(), point.r);
        update_vEB(s->dt, sort.q / sort.m, E_p, B_p, point);
        correct_coordinates(gr, point); /* sort->correct_coordinates() :227 */
      }
  }
  for (auto& sort : s->sorts) update_cells_seq(gr, sort); /* :233 */
  ecsim_final_update(s);
  return its;
}

int step_ecsimcorr(orc_sim* s) /* src/impls/ecsimcorr/simulation.cpp:21-90 */
{
  const Grid& gr = s->gr;
  const long n3 = gr.N * 3;
  /* clear_sources :34-49 */
  std::fill(s->currJe.begin(), s->currJe.end(), 0.0);
  for (auto& sort : s->sorts) {
    std::fill(sort.currJe.begin(), sort.currJe.end(), 0.0);
    sort.currJe_loc.assign(gr.G * 3, 0.0);
    calculate_energy(sort);
  }
  /* first_push (ecsim::Simulation::first_push with the virtual ecsimcorr first_push) */
  int bad = 0;
  for (auto& sort : s->sorts) bad += ecsimcorr_first_push(s, sort);
  for (auto& sort : s->sorts) update_cells_seq(gr, sort);
  ecsim_fill_current(s);
  int its0 = advance_fields(s, 0, s->currI, s->Ep, 0); /* predict */
  if (its0 < 0) return its0;
  /* second_push (ecsim::Simulation::second_push :212-239) */
  global_to_local(gr, s->Ep, s->E_loc, 3);
  global_to_local(gr, s->B, s->B_loc, 3);
  for (auto& sort : s->sorts) {
    bad += ecsimcorr_second_push(s, sort);
    const long ncell = (long)sort.storage.size();
#pragma omp parallel for
    for (long g = 0; g < ncell; ++g)
      for (auto& point : sort.storage[g]) correct_coordinates(gr, point);
  }
  for (auto& sort : s->sorts) update_cells_seq(gr, sort);
  /* correct_fields :52-63: KSP "correct" on matM with currJe */
  int its1 = advance_fields(s, 1, s->currJe, s->Ec, 1);
  if (its1 < 0) return its1;
  /* final_update :65-90 */
  for (auto& sort : s->sorts) ecsimcorr_final_update(s, sort);
  matL_apply(s, s->Ec.data(), s->currI.data(), true); /* MatMultAdd(matL, Ec, currI, currI) :78 */
  std::swap(s->Ep, s->Ec);                            /* VecSwap :86 */
  ecsim_final_update(s);
  (void)n3;
  return bad ? -1 : its0 + its1;
}

struct RhoShape { /* ParticlesChargeDensity::Shape (src/diagnostics/charge_conservation.cpp:34-63) */
  static constexpr int shw = 3; /* (PetscInt)(2.0 * 1.5) */
  static constexpr int shm = 27;
  int start[3];
  double cache[shm];
  void setup(const Grid& gr, const V3& r)
  {
    V3 p_r(r[X] / gr.d[X], r[Y] / gr.d[Y], r[Z] / gr.d[Z]);
    for (int c = 0; c < 3; ++c) start[c] = (int)(std::ceil(p_r[c] - 1.5));
    for (int i = 0; i < shm; ++i) {
      double g_x = (double)(start[X] + i % shw);
      double g_y = (double)(start[Y] + (i / shw) % shw);
      double g_z = (double)(start[Z] + (i / shw) / shw);
      cache[i] = spline2(p_r[X] - g_x) * spline2(p_r[Y] - g_y) * spline2(p_r[Z] - g_z);
    }
  }
};


/* ------------------------------------------------------------------------------------------
 * eccapfim inner kernels
 * cell_traversal  (src/impls/eccapfim/cell_traversal.cpp:3-77): the points at which the straight
 * path start -> end crosses the faces of the node-centred cells (cell of r = round(r/d)).
 * ---------------------------------------------------------------------------------------- */
std::vector<V3> cell_traversal(const double* d3, const V3& end, const V3& start)
{
  int curr[3], last[3];
  for (int c = 0; c < 3; ++c) {
    curr[c] = (int)std::round(start[c] / d3[c]);
    last[c] = (int)std::round(end[c] / d3[c]);
  }
  if (curr[0] == last[0] && curr[1] == last[1] && curr[2] == last[2]) return {start, end};
  V3 dir(end[X] - start[X], end[Y] - start[Y], end[Z] - start[Z]);
  int sg[3];
  double nxt[3], tt[3], dtt[3];
  static const double max = std::numeric_limits<double>::max();
  for (int c = 0; c < 3; ++c) {
    sg[c] = dir[c] > 0 ? 1 : -1;
    nxt[c] = (curr[c] + sg[c] * 0.5) * d3[c];
    tt[c] = (dir[c] != 0) ? (nxt[c] - start[c]) / dir[c] : max;
    dtt[c] = (dir[c] != 0) ? d3[c] / dir[c] * sg[c] : 0.0;
  }
  std::vector<V3> points;
  points.push_back(start);
  double t;
  while (!(curr[0] == last[0] && curr[1] == last[1] && curr[2] == last[2])) {
    if (tt[X] < tt[Y]) {
      if (tt[X] < tt[Z]) { t = tt[X]; curr[X] += sg[X]; tt[X] += dtt[X]; }
      else { t = tt[Z]; curr[Z] += sg[Z]; tt[Z] += dtt[Z]; }
    }
    else {
      if (tt[Y] < tt[Z]) { t = tt[Y]; curr[Y] += sg[Y]; tt[Y] += dtt[Y]; }
      else { t = tt[Z]; curr[Z] += sg[Z]; tt[Z] += dtt[Z]; }
    }
    points.push_back(V3(start[X] + dir[X] * t, start[Y] + dir[Y] * t, start[Z] + dir[Z] * t));
    if (points.size() > 64) break; /* the reference loops until curr == last; a guard for degenerate input */
  }
  points.push_back(end);
  return points;
}

/* ImplicitEsirkepov::Shape::setup  (src/algorithms/implicit_esirkepov.cpp:11-57) */
struct ImplicitShape {
  static constexpr int shw1 = 2, shw2 = 3, shm = 3 * 3 * 2 * 3;
  int start[3];
  double cache[shm];
  static double sfunc_1(double s) { return 1.0 - std::abs(s); }
  static double sfunc_21(double s) { s = std::abs(s); return (0.75 - s * s); }
  static double sfunc_22(double s) { s = std::abs(s); return 0.5 * (1.5 - s) * (1.5 - s); }
  static double sfunc_2(int j, double s) { return j == 1 ? sfunc_21(s) : sfunc_22(s); }
  void setup(const double* d3, const V3& rn, const V3& r0)
  {
    double prn[3], pr0[3], prh[3], gc[3], gv[3];
    for (int c = 0; c < 3; ++c) {
      prn[c] = rn[c] / d3[c];
      pr0[c] = r0[c] / d3[c];
      prh[c] = 0.5 * (prn[c] + pr0[c]);
      gc[c] = std::round(prh[c]);
      start[c] = (int)gc[c] - 1;
      gv[c] = gc[c] + 0.5;
    }
    int m = 0;
    static constexpr double sixth = 1.0 / 6.0;
    for (int cx = 0; cx < 3; cx++) {
      int cy = (cx + 1) % 3, cz = (cx + 2) % 3;
      for (int i = 0; i < 2; i++) {
        double shx = sixth * sfunc_1(gv[cx] + (i - 1) - prh[cx]);
        for (int j = 0; j < 3; j++) {
          double sny = sfunc_2(j, gc[cy] + (j - 1) - prn[cy]);
          double s0y = sfunc_2(j, gc[cy] + (j - 1) - pr0[cy]);
          for (int k = 0; k < 3; k++) {
            double snz = sfunc_2(k, gc[cz] + (k - 1) - prn[cz]);
            double s0z = sfunc_2(k, gc[cz] + (k - 1) - pr0[cz]);
            cache[m++] = shx * (sny * (2 * snz + s0z) + s0y * (2 * s0z + snz));
          }
        }
      }
    }
  }
  /* the 54 (node, component) pairs in cache order (:73-88, :99-114) */
  template <class F>
  void for_each(F f) const
  {
    int m = 0, i[3];
    for (int cx = 0; cx < 3; cx++) {
      int cy = (cx + 1) % 3, cz = (cx + 2) % 3;
      for (i[cx] = 0; i[cx] < 2; i[cx]++)
        for (i[cy] = 0; i[cy] < 3; i[cy]++)
          for (i[cz] = 0; i[cz] < 3; i[cz]++) f(start[X] + i[X], start[Y] + i[Y], start[Z] + i[Z], cx, cache[m++]);
    }
  }
};

/* ParticlesChargeDensity::collect  (src/diagnostics/charge_conservation.cpp:67-97) */
void charge_collect(const orc_sim* s, const Sort& sort, std::vector<double>& field)
{
  const Grid& gr = s->gr;
  std::vector<double> local(gr.G, 0.0);
  field.assign(gr.N, 0.0);
  double* arr = local.data();
  const double q = sort.q;
  const long ncell = (long)sort.storage.size();
#pragma omp parallel for schedule(dynamic, 16)
  for (long g = 0; g < ncell; ++g)
    for (auto& point : sort.storage[g]) {
      RhoShape shape;
      shape.setup(gr, point.r);
      for (int i = 0; i < RhoShape::shm; ++i) {
        int g_x = shape.start[X] + i % RhoShape::shw;
        int g_y = shape.start[Y] + (i / RhoShape::shw) % RhoShape::shw;
        int g_z = shape.start[Z] + (i / RhoShape::shw) / RhoShape::shw;
#pragma omp atomic update
        arr[gr.sl(g_x, g_y, g_z)] += q * shape.cache[i] * sort.n_Np();
      }
    }
  local_to_global_add(gr, local, field, 1);
}

/* DistributionMoment::collect with the "density" moment (src/diagnostics/distribution_moment.cpp:125-205,212-216):
 * cell-centred, 2 x 2 x 2 cells from round(r/dx - 1), spline_of_1st_order, value n/Np */
void density_collect(const orc_sim* s, const Sort& sort, std::vector<double>& field)
{
  const Grid& gr = s->gr;
  std::vector<double> local(gr.G, 0.0);
  field.assign(gr.N, 0.0);
  double* arr = local.data();
  const long ncell = (long)sort.storage.size();
#pragma omp parallel for schedule(dynamic, 16)
  for (long g = 0; g < ncell; ++g)
    for (auto& point : sort.storage[g]) {
      const double p_r[3] = {point.r[X] / gr.d[X], point.r[Y] / gr.d[Y], point.r[Z] / gr.d[Z]};
      int start[3];
      for (int a = 0; a < 3; ++a) start[a] = (int)std::round(p_r[a] - 1.0);
      for (int i = 0; i < 8; ++i) {
        int g_x = start[X] + i % 2, g_y = start[Y] + (i / 2) % 2, g_z = start[Z] + (i / 2) / 2;
        double c = spline1(p_r[X] - ((double)g_x + 0.5)) * spline1(p_r[Y] - ((double)g_y + 0.5)) *
          spline1(p_r[Z] - ((double)g_z + 0.5));
        double si = c * sort.n_Np();
#pragma omp atomic update
        arr[gr.sl(g_x, g_y, g_z)] += 1.0 * si;
      }
    }
  local_to_global_add(gr, local, field, 1);
}

/* Divergence, negative Yee shift  (src/utils/operators.cpp:275-333) */
void div_neg(const Grid& gr, const double* v, double* out, bool add)
{
  const double ix = 1.0 / gr.d[X], iy = 1.0 / gr.d[Y], iz = 1.0 / gr.d[Z];
#pragma omp parallel for collapse(2)
  for (int z = 0; z < gr.n[2]; ++z)
    for (int y = 0; y < gr.n[1]; ++y)
      for (int x = 0; x < gr.n[0]; ++x) {
        double dv = +ix * v[gr.vg(x, y, z, X)] - ix * v[gr.vgw(x - 1, y, z, X)] + iy * v[gr.vg(x, y, z, Y)] -
          iy * v[gr.vgw(x, y - 1, z, Y)] + iz * v[gr.vg(x, y, z, Z)] - iz * v[gr.vgw(x, y, z - 1, Z)];
        long o = gr.sg(x, y, z);
        if (add) out[o] += dv; else out[o] = dv;
      }
}

std::vector<double>* named(orc_sim* s, const char* name)
{
  std::string n(name);
  if (n == "E") return &s->E;
  if (n == "B") return &s->B;
  if (n == "B0") return &s->B0;
  if (n == "Ep") return &s->Ep;
  if (n == "Ec") return &s->Ec;
  if (n == "J") return s->scheme == 0 ? &s->J : (s->scheme == 1 ? &s->currI : &s->currJe);
  if (n == "currI") return &s->currI;
  if (n == "currJe") return &s->currJe;
  return nullptr;
}

}  // namespace

/* ============================================================================================
 * C API
 * ========================================================================================== */
extern "C" {

void orc_set_threads(int n) { omp_set_num_threads(n); }

void orc_update_r(double dt, double* p6)
{
  Point pt;
  pt.r = V3(p6); pt.p = V3(p6 + 3);
  update_r(dt, pt);
  for (int c = 0; c < 3; ++c) { p6[c] = pt.r[c]; p6[3 + c] = pt.p[c]; }
}

void orc_update_vEB(double dt, double qm, const double* E_p, const double* B_p, double* p6)
{
  Point pt;
  pt.r = V3(p6); pt.p = V3(p6 + 3);
  update_vEB(dt, qm, V3(E_p), V3(B_p), pt);
  for (int c = 0; c < 3; ++c) p6[3 + c] = pt.p[c];
}

void orc_update_vX(char kind, double dt, double qm, const double* B_p, double* p6)
{
  Point pt;
  pt.r = V3(p6); pt.p = V3(p6 + 3);
  update_vX(kind, dt, qm, V3(B_p), pt);
  for (int c = 0; c < 3; ++c) p6[3 + c] = pt.p[c];
}

/* tests/boris_push/boris_push.h:19-231 (process_* schemes) driven as in boris_push_ex{1..6}.cpp */
int orc_boris_test_trajectory(int example, const char* scheme_id, double* rows, int max_rows)
{
  std::string id(scheme_id);
  static const char* known[] = {"M1A", "M1B", "MLF", "B1A", "B1B", "BLF", "C1A", "C1B", "CLF", "M2A", "M2B",
    "C2A", "B2B", "EB1A", "EB1B", "EBLF", "EB2B"};
  bool ok = false;
  for (auto k : known) ok = ok || id == k;
  if (!ok) return -1;

  double dt;
  long nt;
  int skip;
  V3 r0, v0;
  const double qm = -1.0;
  switch (example) {
    case 1: r0 = V3(0.5, 0, 0); v0 = V3(0, 1, 0); dt = M_PI / 4.0; nt = 100000; skip = 543; break;
    case 2: r0 = V3(0, 0, 0); v0 = V3(0, 0, 2); dt = 0.5; nt = 1000; skip = 5; break;
    case 3: r0 = V3(0, 10, 0); v0 = V3(0.16, 1, 0); dt = 0.16; nt = 1000; skip = 5; break;
    case 4: r0 = V3(0, 0, 0); v0 = V3(0.1, 0, 0.4); dt = 0.1975; nt = 5000; skip = 32; break;
    case 5: r0 = V3(0, 0, 0); v0 = V3(0, 0, 0.1); dt = 0.5; nt = 10000; skip = 54; break;
    case 6: r0 = V3(0, -1, 0); v0 = V3(0.1, 0.01, 0); dt = 2.1 * M_PI; nt = static_cast<long>(std::round(1000 / dt)); skip = 1; break;
    default: return -1;
  }

  auto fields = [&](long t, const V3& r, V3& E_p, V3& B_p) {
    switch (example) {
      case 1: B_p = V3(0, 0, 2.0); break;                              /* ex1.cpp:9,88-91 */
      case 2: B_p = V3(100 - 25 * r[Y], 0.0, 0.0); break;              /* ex2 get_magnetic_field */
      case 3: {                                                         /* ex3 get_magnetic_field */
        V3 cr = r - V3(10, 10, 0);
        double rr = cr.length();
        double ra = std::atan2(cr[Y], cr[X]);
        double B_theta = 800 / rr;
        B_p = V3(-std::sin(ra) * B_theta, +std::cos(ra) * B_theta, 0.0);
        break;
      }
      case 4: E_p = V3(0, 0, 1); B_p = V3(250, 0, 0); break;           /* ex4.cpp:11-12,80-84 */
      case 5: E_p = V3(0, -2, 0) * ((double)t * dt); B_p = V3(100, 0, 0); break; /* ex5 lambda */
      case 6: {                                                         /* ex6 interpolated_fields */
        double rr = r.length();
        E_p = V3(0.1 * r[X] / (rr * rr * rr), 0.1 * r[Y] / (rr * rr * rr), 0.0);
        B_p = V3(0.0, 0.0, 1.0 * rr);
        break;
      }
    }
  };

  Point point;
  point.r = r0;
  point.p = v0;
  bool lf = id.size() >= 2 && id.compare(id.size() - 2, 2, "LF") == 0;
  if (lf) update_r(-dt / 2.0, point);

  auto vkind = [&](char k, double h, long t) {
    V3 E_p, B_p;
    fields(t, point.r, E_p, B_p);
    if (k == 'E') update_vEB(h, qm, E_p, B_p, point);
    else update_vX(k, h, qm, B_p, point);
  };

  int nrows = 0;
  for (long t = 0; t <= nt; ++t) {
    if (t % skip == 0) {
      if (nrows >= max_rows) return nrows;
      double* row = rows + 7 * (long)nrows++;
      row[0] = t * dt;
      for (int c = 0; c < 3; ++c) { row[1 + c] = point.r[c]; row[4 + c] = point.p[c]; }
    }
    /* process_impl, boris_push.h:201-231 */
    char fam = id[0] == 'E' ? 'E' : id[0];
    std::string tail = id.substr(fam == 'E' ? 2 : 1);
    char k = fam;
    if (fam == 'C') k = (tail == "2A") ? '2' : '1';
    if (tail == "1A") { vkind(k, dt, t); update_r(dt, point); }
    else if (tail == "1B" || tail == "LF") { update_r(dt, point); vkind(k, dt, t); }
    else if (tail == "2A") { vkind(k, dt / 2.0, t); update_r(dt, point); vkind(k, dt / 2.0, t); }
    else if (tail == "2B") { update_r(dt / 2.0, point); vkind(k, dt, t); update_r(dt / 2.0, point); }
  }
  return nrows;
}

double orc_spline(int order, double s) { return spline_of(order)(s); }

int orc_shape_setup(const double* d3, const double* r1, const double* r2, int pair, double radius, int order,
  int* start3, int* size3, double* shape_out)
{
  Shape sh;
  if (pair) sh.setup(d3, V3(r1), V3(r2), radius, spline_of(order));
  else sh.setup(d3, V3(r1), radius, spline_of(order));
  for (int c = 0; c < 3; ++c) { start3[c] = sh.start[c]; size3[c] = sh.size[c]; }
  if (sh.overflow) return 1;
  std::memcpy(shape_out, sh.shape, sizeof(double) * sh.elements() * shc);
  return 0;
}

orc_sim* orc_create(int scheme, int nx, int ny, int nz, double dx, double dy, double dz, double dt)
{
  orc_sim* s = new orc_sim;
  s->scheme = scheme;
  s->gr.set(nx, ny, nz, dx, dy, dz);
  s->dt = dt;
  long n3 = s->gr.N * 3;
  for (auto v : {&s->E, &s->B, &s->B0, &s->Ep, &s->Ec, &s->J, &s->currI, &s->currJe}) v->assign(n3, 0.0);
  return s;
}

void orc_destroy(orc_sim* s) { delete s; }

int orc_add_sort(orc_sim* s, int Np, double n, double q, double m, double Tx, double Ty, double Tz)
{
  Sort sort;
  sort.Np = Np; sort.n = n; sort.q = q; sort.m = m; sort.Tx = Tx; sort.Ty = Ty; sort.Tz = Tz;
  sort.storage.resize(s->gr.N);
  long n3 = s->gr.N * 3;
  sort.J.assign(n3, 0.0);
  sort.currI.assign(n3, 0.0);
  sort.currJe.assign(n3, 0.0);
  sort.J_loc.assign(s->gr.G * 3, 0.0);
  sort.currI_loc.assign(s->gr.G * 3, 0.0);
  sort.currJe_loc.assign(s->gr.G * 3, 0.0);
  s->sorts.push_back(std::move(sort));
  return (int)s->sorts.size() - 1;
}

void orc_reset_rng(void)
{
  rng() = std::mt19937();
  dist01().reset();
}

static bool add_particle(orc_sim* s, Sort& sort, const Point& pt) /* particles.cpp:47-67 */
{
  bool inside;
  long g = cell_of(s->gr, pt.r, &inside);
  if (!inside) return false;
  sort.storage[g].emplace_back(pt);
  return true;
}

long orc_load_maxwell_box(orc_sim* s, int isort, int tov)
{
  Sort& sort = s->sorts[isort];
  const Grid& gr = s->gr;
  /* ParticlesBuilder::load_coordinate (src/commands/builders/particles_builder.cpp:17-27) */
  const double frac = sort.Np / (gr.d[X] * gr.d[Y] * gr.d[Z]);
  V3 bmin(0, 0, 0), bmax(gr.L[X], gr.L[Y], gr.L[Z]);
  V3 ext = bmax - bmin;
  int number_of_particles = (ext[X] * ext[Y] * ext[Z]) * frac; /* truncation to PetscInt as in :26 */
  long added = 0;
  for (int p = 0; p < number_of_particles; ++p) { /* SetParticles::execute set_particles.cpp:19-43 */
    Point pt;
    /* CoordinateInBox particles_load.cpp:11-18 */
    double cx = bmin[X] + random_01() * (bmax[X] - bmin[X]);
    double cy = bmin[Y] + random_01() * (bmax[Y] - bmin[Y]);
    double cz = bmin[Z] + random_01() * (bmax[Z] - bmin[Z]);
    pt.r = V3(cx, cy, cz);
    /* MaxwellianMomentum particles_load.cpp:57-76 (px=py=pz=0: never read from JSON, simulation.tpp:24-41).
     * sin(2 pi u) is drawn BEFORE temperature_momentum's u in each product (operand order). */
    double sx = std::sin(2.0 * M_PI * random_01());
    double mx = 0.0 + sx * temperature_momentum(sort.Tx, sort.m);
    double sy = std::sin(2.0 * M_PI * random_01());
    double my = 0.0 + sy * temperature_momentum(sort.Ty, sort.m);
    double sz = std::sin(2.0 * M_PI * random_01());
    double mz = 0.0 + sz * temperature_momentum(sort.Tz, sort.m);
    V3 mom(mx, my, mz);
    if (tov) mom /= std::sqrt(sort.m * sort.m + mom.squared());
    pt.p = mom;
    if (add_particle(s, sort, pt)) ++added;
  }
  return added;
}

long orc_add_particles(orc_sim* s, int isort, long n, const double* pts6)
{
  Sort& sort = s->sorts[isort];
  long added = 0;
  for (long i = 0; i < n; ++i) {
    Point pt;
    pt.r = V3(pts6 + 6 * i);
    pt.p = V3(pts6 + 6 * i + 3);
    if (add_particle(s, sort, pt)) ++added;
  }
  return added;
}

long orc_count(orc_sim* s, int isort)
{
  long c = 0;
  for (auto& cell : s->sorts[isort].storage) c += (long)cell.size();
  return c;
}

long orc_get_particles(orc_sim* s, int isort, double* pts6, int* cell_ids)
{
  long i = 0;
  Sort& sort = s->sorts[isort];
  for (long g = 0; g < (long)sort.storage.size(); ++g)
    for (auto& pt : sort.storage[g]) {
      for (int c = 0; c < 3; ++c) { pts6[6 * i + c] = pt.r[c]; pts6[6 * i + 3 + c] = pt.p[c]; }
      if (cell_ids) cell_ids[i] = (int)g;
      ++i;
    }
  return i;
}

void orc_clear_particles(orc_sim* s, int isort)
{
  for (auto& cell : s->sorts[isort].storage) cell.clear();
}

int orc_set_field(orc_sim* s, const char* name, const double* v)
{
  auto* f = named(s, name);
  if (!f) return 1;
  std::copy(v, v + f->size(), f->begin());
  return 0;
}

int orc_get_field(orc_sim* s, const char* name, double* v)
{
  auto* f = named(s, name);
  if (!f) return 1;
  std::copy(f->begin(), f->end(), v);
  return 0;
}

int orc_get_sort_current(orc_sim* s, int isort, const char* which, double* v)
{
  Sort& sort = s->sorts[isort];
  std::string w(which);
  const std::vector<double>* f = nullptr;
  if (w == "J") f = &sort.J;
  else if (w == "currI") f = &sort.currI;
  else if (w == "currJe") f = &sort.currJe;
  if (!f) return 1;
  std::copy(f->begin(), f->end(), v);
  return 0;
}

void orc_rot_apply(orc_sim* s, int sign, double alpha, const double* x, double* y) { rot_apply(s->gr, sign, alpha, x, y, false); }
void orc_matM_apply(orc_sim* s, const double* x, double* y) { matM_apply(s, x, y); }
void orc_matL_apply(orc_sim* s, const double* x, double* y) { matL_apply(s, x, y, false); }
void orc_div_neg(orc_sim* s, const double* v3, double* out1) { div_neg(s->gr, v3, out1, false); }

void orc_lstencil_decode(int c1, int k, int* c2, int* d3) { ldecode(c1, k, c2, d3); }
int orc_lstencil_encode(int c1, int c2, int dx, int dy, int dz) { return lencode(c1, c2, dx, dy, dz); }
void orc_get_matL(orc_sim* s, double* out)
{
  if (s->matL.empty()) std::fill(out, out + s->gr.N * 3 * ORC_LSTENCIL, 0.0);
  else std::copy(s->matL.begin(), s->matL.end(), out);
}

void orc_gather_s2(orc_sim* s, const double* E, const double* B, const double* r3, double* out6)
{
  std::vector<double> Eg(E, E + s->gr.N * 3), Bg(B, B + s->gr.N * 3), El, Bl;
  global_to_local(s->gr, Eg, El, 3);
  global_to_local(s->gr, Bg, Bl, 3);
  Shape sh;
  sh.setup(s->gr.d, V3(r3), 1.5, spline2);
  V3 E_p, B_p;
  simple_interpolation(s->gr, sh, El.data(), Bl.data(), E_p, B_p);
  for (int c = 0; c < 3; ++c) { out6[c] = E_p[c]; out6[3 + c] = B_p[c]; }
}

void orc_gather_s1(orc_sim* s, const double* E, const double* B, const double* r3, double* out6)
{
  std::vector<double> Eg(E, E + s->gr.N * 3), Bg(B, B + s->gr.N * 3), El, Bl;
  global_to_local(s->gr, Eg, El, 3);
  global_to_local(s->gr, Bg, Bl, 3);
  V3 E_p = interpolate_E_s1(s->gr, El.data(), V3(r3));
  V3 B_p = interpolate_B_s1(s->gr, Bl.data(), V3(r3));
  for (int c = 0; c < 3; ++c) { out6[c] = E_p[c]; out6[3 + c] = B_p[c]; }
}

int orc_esirkepov(orc_sim* s, long n, const double* old_r3, const double* new_r3, double alpha, double* J)
{
  std::vector<double> Jl(s->gr.G * 3, 0.0), Jg(s->gr.N * 3, 0.0);
  int bad = 0;
  for (long i = 0; i < n; ++i) {
    Shape sh;
    sh.setup(s->gr.d, V3(old_r3 + 3 * i), V3(new_r3 + 3 * i), 1.5, spline2);
    if (sh.overflow) { ++bad; continue; }
    esirkepov_process(s->gr, sh, alpha, Jl.data());
  }
  local_to_global_add(s->gr, Jl, Jg, 3);
  for (long i = 0; i < s->gr.N * 3; ++i) J[i] += Jg[i];
  return bad;
}

int orc_basic_push(orc_sim* s)
{
  std::fill(s->J.begin(), s->J.end(), 0.0);
  global_to_local(s->gr, s->E, s->E_loc, 3);
  global_to_local(s->gr, s->B, s->B_loc, 3);
  int bad = 0;
  for (auto& sort : s->sorts) {
    std::fill(sort.J.begin(), sort.J.end(), 0.0);
    sort.J_loc.assign(s->gr.G * 3, 0.0);
    bad += basic_push_sort(s, sort);
  }
  return bad;
}

void orc_update_cells(orc_sim* s, int isort) { update_cells_seq(s->gr, s->sorts[isort]); }

void orc_ecsim_first_push(orc_sim* s, int isort)
{
  Sort& sort = s->sorts[isort];
  const long ncell = (long)sort.storage.size();
#pragma omp parallel for schedule(dynamic, 16)
  for (long g = 0; g < ncell; ++g)
    for (auto& point : sort.storage[g]) update_r(s->dt, point);
}

void orc_ecsim_fill_current(orc_sim* s) { ecsim_fill_current(s); }

void orc_ecsim_second_push(orc_sim* s, int isort)
{
  const Grid& gr = s->gr;
  global_to_local(gr, s->Ep, s->E_loc, 3);
  global_to_local(gr, s->B, s->B_loc, 3);
  Sort& sort = s->sorts[isort];
  const long ncell = (long)sort.storage.size();
#pragma omp parallel for schedule(dynamic, 16)
  for (long g = 0; g < ncell; ++g)
    for (auto& point : sort.storage[g]) {
      V3 E_p = interpolate_E_s1(gr, s->E_loc.data(), point.r);
      V3 B_p = interpolate_B_s1(gr, s->B_loc.data(), point.r);
      update_vEB(s->dt, sort.q / sort.m, E_p, B_p, point);
    }
}

int orc_ecsimcorr_first_push(orc_sim* s, int isort)
{
  Sort& sort = s->sorts[isort];
  std::fill(sort.currJe.begin(), sort.currJe.end(), 0.0);
  sort.currJe_loc.assign(s->gr.G * 3, 0.0);
  return ecsimcorr_first_push(s, sort);
}

int orc_ecsimcorr_second_push(orc_sim* s, int isort)
{
  global_to_local(s->gr, s->Ep, s->E_loc, 3);
  global_to_local(s->gr, s->B, s->B_loc, 3);
  return ecsimcorr_second_push(s, s->sorts[isort]);
}

void orc_ecsimcorr_final_update(orc_sim* s, int isort) { ecsimcorr_final_update(s, s->sorts[isort]); }
double orc_calculate_energy(orc_sim* s, int isort) { return calculate_energy(s->sorts[isort]); }

void orc_ecsimcorr_scalars(orc_sim* s, int isort, double* o)
{
  Sort& t = s->sorts[isort];
  o[0] = t.pred_w; o[1] = t.corr_w; o[2] = t.lambda_dK; o[3] = t.pred_dK; o[4] = t.corr_dK; o[5] = t.energy;
}

void orc_set_tolerances(orc_sim* s, double rtol, double atol, int maxit)
{
  s->rtol = rtol; s->atol = atol; s->maxit = maxit;
}

int orc_solve(orc_sim* s, int op, const double* rhs, double* x, double rtol, double atol, int maxit, double* rn)
{
  if (op == 0) return gmres(s, matA_apply, rhs, x, rtol, atol, maxit, rn);
  if (op == 1) return gmres(s, matM_apply, rhs, x, rtol, atol, maxit, rn);
  return cg(s, matM_apply, rhs, x, rtol, atol, maxit, rn);
}

void orc_solve_stats(orc_sim* s, double* seconds, long* iterations, int reset)
{
  *seconds = s->solve_seconds;
  *iterations = s->solve_its;
  if (reset) { s->solve_seconds = 0; s->solve_its = 0; }
}

int orc_step(orc_sim* s)
{
  switch (s->scheme) {
    case 0: return step_basic(s);
    case 1: return step_ecsim(s);
    default: return step_ecsimcorr(s);
  }
}

void orc_energy(orc_sim* s, double* out) /* src/diagnostics/energy.cpp:43-108 */
{
  const Grid& gr = s->gr;
  const long n3 = gr.N * 3;
  auto field = [&](const std::vector<double>& F, double& w, double& sd) {
    double nrm = std::sqrt(vdot(n3, F.data(), F.data())); /* VecNorm NORM_2 */
    w = 0.5 * (nrm * nrm);
    V3 mean;
    for (long i = 0; i < gr.N; ++i) { mean[X] += F[3 * i]; mean[Y] += F[3 * i + 1]; mean[Z] += F[3 * i + 2]; }
    double g3 = (double)(gr.n[0] * gr.n[1] * gr.n[2]);
    sd = std::sqrt((w - 0.5 * mean.squared() / g3) / g3);
  };
  field(s->E, out[0], out[2]);
  field(s->B, out[1], out[3]);
  for (size_t i = 0; i < s->sorts.size(); ++i) {
    Sort& sort = s->sorts[i];
    double m = sort.m, mpw = sort.n / (double)sort.Np;
    double frac = 0.5 * m * mpw;
    double vx = 0, vy = 0, vz = 0, w = 0;
    long n = 0;
    const long ncell = (long)sort.storage.size();
#pragma omp parallel for reduction(+ : vx, vy, vz, w, n)
    for (long g = 0; g < ncell; ++g)
      for (auto& point : sort.storage[g]) {
        vx += point.p[X]; vy += point.p[Y]; vz += point.p[Z];
        w += point.p.squared();
        n++;
      }
    double K = frac * w, sK = 0;
    if (n == 0) K = 0;
    else {
      double sv = w - (vx * vx + vy * vy + vz * vz) / n;
      sK = frac * std::sqrt(std::abs(sv) / n);
    }
    out[4 + 2 * i] = K;
    out[5 + 2 * i] = sK;
  }
}

void orc_charge_density(orc_sim* s, int isort, double* rho)
{
  std::vector<double> f;
  charge_collect(s, s->sorts[isort], f);
  std::copy(f.begin(), f.end(), rho);
}


int orc_cell_traversal(const double* d3, const double* end3, const double* start3, int max_pts, double* pts)
{
  std::vector<V3> p = cell_traversal(d3, V3(end3), V3(start3));
  int n = (int)p.size();
  for (int i = 0; i < n && i < max_pts; ++i)
    for (int c = 0; c < 3; ++c) pts[3 * i + c] = p[i][c];
  return n;
}

/* ImplicitEsirkepov::interpolate (src/algorithms/implicit_esirkepov.cpp:60-90) on the simulation's E and B */
void orc_implicit_esirkepov_interpolate(orc_sim* s, long n, const double* rn3, const double* r03, double* Ep3, double* Bp3)
{
  const Grid& gr = s->gr;
  std::vector<double> El, Bl;
  global_to_local(gr, s->E, El, 3);
  global_to_local(gr, s->B, Bl, 3);
  for (long q = 0; q < n; ++q) {
    V3 rn(rn3 + 3 * q), r0(r03 + 3 * q), E_p, B_p, none;
    Shape sh_m;
    V3 mid(0.5 * (rn[X] + r0[X]), 0.5 * (rn[Y] + r0[Y]), 0.5 * (rn[Z] + r0[Z]));
    sh_m.setup(gr.d, mid, 1.5, spline_of(2)); /* shape_radius / shape_function, sort_parameters.h:44-46 */
    simple_interpolation(gr, sh_m, nullptr, Bl.data(), none, B_p);
    ImplicitShape sh_e;
    sh_e.setup(gr.d, rn, r0);
    sh_e.for_each([&](int gx, int gy, int gz, int c, double w) { E_p[c] += El[gr.vl(gx, gy, gz, c)] * w; });
    for (int c = 0; c < 3; ++c) { Ep3[3 * q + c] = E_p[c]; Bp3[3 * q + c] = B_p[c]; }
  }
}

/* ImplicitEsirkepov::decompose (:92-117), then DMLocalToGlobal(ADD) into the named vector */
int orc_implicit_esirkepov_decompose(orc_sim* s, long n, const double* alpha, const double* v3, const double* rn3,
  const double* r03, const char* field)
{
  const Grid& gr = s->gr;
  std::vector<double>* F = named(s, field);
  if (!F) return 1;
  std::vector<double> Jl(gr.G * 3, 0.0);
  for (long q = 0; q < n; ++q) {
    ImplicitShape sh_e;
    sh_e.setup(gr.d, V3(rn3 + 3 * q), V3(r03 + 3 * q));
    sh_e.for_each([&](int gx, int gy, int gz, int c, double w) { Jl[gr.vl(gx, gy, gz, c)] += alpha[q] * v3[3 * q + c] * w; });
  }
  local_to_global_add(gr, Jl, *F, 3);
  return 0;
}

void orc_moment_density(orc_sim* s, int isort, double* out)
{
  std::vector<double> f;
  density_collect(s, s->sorts[isort], f);
  std::copy(f.begin(), f.end(), out);
}

/* MomentumConservation::calculate (src/diagnostics/momentum_conservation.cpp:77-131): per sort
 * P = sum_p sum_nodes (m/Np) v ns,  QE = sum_p sum_nodes (q/Np) E[node] Es   with Shape::setup(point.r) */
void orc_momentum(orc_sim* s, double* out)
{
  const Grid& gr = s->gr;
  std::vector<double> El;
  global_to_local(gr, s->E, El, 3);
  for (size_t is = 0; is < s->sorts.size(); ++is) {
    Sort& sort = s->sorts[is];
    const double Np = (double)sort.Np;
    const double m = sort.m / Np, q = sort.q / Np;
    double px = 0, py = 0, pz = 0, qex = 0, qey = 0, qez = 0;
    const long ncell = (long)sort.storage.size();
#pragma omp parallel for reduction(+ : px, py, pz, qex, qey, qez)
    for (long g = 0; g < ncell; ++g)
      for (auto& point : sort.storage[g]) {
        Shape shape;
        shape.setup(gr.d, point.r, 1.5, spline2);
        for (int i = 0; i < shape.elements(); ++i) {
          int gx = shape.start[X] + i % shape.size[X];
          int gy = shape.start[Y] + (i / shape.size[X]) % shape.size[Y];
          int gz = shape.start[Z] + (i / shape.size[X]) / shape.size[Y];
          double ns = shape(i, No, Z) * shape(i, No, Y) * shape(i, No, X);
          V3 Es = shape.electric(i);
          long l = gr.vl(gx, gy, gz, 0);
          px += m * point.p[X] * ns;
          py += m * point.p[Y] * ns;
          pz += m * point.p[Z] * ns;
          qex += q * El[l + X] * Es[X];
          qey += q * El[l + Y] * Es[Y];
          qez += q * El[l + Z] * Es[Z];
        }
      }
    double* o = out + 6 * is;
    o[0] = px; o[1] = py; o[2] = pz; o[3] = qex; o[4] = qey; o[5] = qez;
  }
}

void orc_charge_collect(orc_sim* s) /* ChargeConservation::initialize :117-123 */
{
  for (auto& sort : s->sorts) charge_collect(s, sort, sort.rho);
}

void orc_charge_columns(orc_sim* s, double* out) /* ChargeConservation::add_columns :125-171 */
{
  const Grid& gr = s->gr;
  std::vector<double> sum(gr.N, 0.0), diff(gr.N);
  auto norms = [&](const std::vector<double>& v, double* o) {
    double n1 = 0, n2 = 0;
    for (double x : v) { n1 += std::abs(x); n2 += x * x; }
    o[0] = n1; o[1] = std::sqrt(n2);
  };
  size_t i = 0;
  for (; i < s->sorts.size(); ++i) {
    Sort& sort = s->sorts[i];
    diff = sort.rho;
    charge_collect(s, sort, sort.rho);
    for (long k = 0; k < gr.N; ++k) diff[k] = (-1.0 * diff[k] + sort.rho[k]) * (1.0 / s->dt);
    for (long k = 0; k < gr.N; ++k) sum[k] += diff[k];
    const std::vector<double>& cur = s->scheme == 0 ? sort.J : (s->scheme == 1 ? sort.currI : sort.currJe);
    div_neg(gr, cur.data(), diff.data(), true);
    norms(diff, out + 2 * i);
  }
  const std::vector<double>& cur = s->scheme == 0 ? s->J : (s->scheme == 1 ? s->currI : s->currJe);
  div_neg(gr, cur.data(), sum.data(), true);
  norms(sum, out + 2 * i);
}

}  // extern "C"
