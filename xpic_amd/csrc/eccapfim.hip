// eccapfim.hip -- the inner kernels of the reference's `eccapfim` scheme, as batch operations
// over path segments r0 -> rn:
//   cell_traversal                     src/impls/eccapfim/cell_traversal.cpp:3-77
//   ImplicitEsirkepov::Shape::setup    src/algorithms/implicit_esirkepov.cpp:11-57
//   ImplicitEsirkepov::interpolate     :60-90   (E with the segment shape, B with Shape(midpoint) + SimpleInterpolation)
//   ImplicitEsirkepov::decompose       :92-117
// One lane per segment.  The scheme's outer loops (Crank-Nicolson sub-stepping, SNES) are not part of this build.
#include <cfloat>

#include "common.h"
#include "device_common.h"

namespace xpic {

namespace {

constexpr int kBlock = 256;

__device__ inline double ie_sfunc_1(double s) { return 1.0 - fabs(s); }
__device__ inline double ie_sfunc_2(int j, double s)
{
  s = fabs(s);
  return j == 1 ? (0.75 - s * s) : 0.5 * (1.5 - s) * (1.5 - s);
}
__device__ inline double spline2_e(double s)
{
  s = fabs(s);
  if (s <= 0.5) return (0.75 - s * s);
  if (0.5 < s && s < 1.5) return 0.5 * (1.5 - s) * (1.5 - s);
  return 0.0;
}

struct IEShape {
  int start[3];
  double cache[54];
  __device__ void setup(const GridDev& g, const double* rn, const double* r0)
  {
    const double d[3] = {g.dx, g.dy, g.dz};
    double prn[3], pr0[3], prh[3], gc[3], gv[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      prn[c] = rn[c] / d[c];
      pr0[c] = r0[c] / d[c];
      prh[c] = 0.5 * (prn[c] + pr0[c]);
      gc[c] = round(prh[c]);
      start[c] = (int)gc[c] - 1;
      gv[c] = gc[c] + 0.5;
    }
    int m = 0;
    const double sixth = 1.0 / 6.0;
#pragma unroll
    for (int cx = 0; cx < 3; cx++) {
      const int cy = (cx + 1) % 3, cz = (cx + 2) % 3;
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const double shx = sixth * ie_sfunc_1(gv[cx] + (i - 1) - prh[cx]);
#pragma unroll
        for (int j = 0; j < 3; j++) {
          const double sny = ie_sfunc_2(j, gc[cy] + (j - 1) - prn[cy]);
          const double s0y = ie_sfunc_2(j, gc[cy] + (j - 1) - pr0[cy]);
#pragma unroll
          for (int k = 0; k < 3; k++) {
            const double snz = ie_sfunc_2(k, gc[cz] + (k - 1) - prn[cz]);
            const double s0z = ie_sfunc_2(k, gc[cz] + (k - 1) - pr0[cz]);
            cache[m++] = shx * (sny * (2 * snz + s0z) + s0y * (2 * s0z + snz));
          }
        }
      }
    }
  }
};

// node (gx, gy, gz) in global numbering -> element of component c of a field vector
__device__ inline long ie_node(const GridDev& g, int gx, int gy, int gz)
{
  const int x = ((gx % g.nx) + g.nx) % g.nx, y = ((gy % g.ny) + g.ny) % g.ny;
  const int zl = g.G == 0 ? ((gz % g.nzl) + g.nzl) % g.nzl : gz - g.z0;
  return g.node(x, y, g.wz(zl));
}

__global__ void __launch_bounds__(kBlock) k_ie_interpolate(GridDev g, const double* __restrict__ E,
  const double* __restrict__ B, long n, const double* rn3, const double* r03, double* Ep3, double* Bp3)
{
  const long q = (long)blockIdx.x * kBlock + threadIdx.x;
  if (q >= n) return;
  const double rn[3] = {rn3[3 * q], rn3[3 * q + 1], rn3[3 * q + 2]}, r0[3] = {r03[3 * q], r03[3 * q + 1], r03[3 * q + 2]};
  // B: Shape::setup(midpoint, 1.5, spline_of_2nd_order) + SimpleInterpolation (magnetic products, shape.h:65-72)
  double Bp[3] = {0, 0, 0};
  {
    const double d[3] = {g.dx, g.dy, g.dz};
    int st[3], sz[3];
    double No[3][4], Sh[3][4];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double pr = 0.5 * (rn[a] + r0[a]) / d[a];
      st[a] = (int)round(pr - 1.5);
      sz[a] = (int)floor(pr + 1.5) + 1 - st[a];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const double gx = (double)(st[a] + t);
        No[a][t] = spline2_e(pr - gx);
        Sh[a][t] = spline2_e(pr - (gx + 0.5));
      }
    }
    for (int kz = 0; kz < sz[2]; ++kz)
      for (int jy = 0; jy < sz[1]; ++jy)
        for (int ix = 0; ix < sz[0]; ++ix) {
          const long o = ie_node(g, st[0] + ix, st[1] + jy, st[2] + kz);
          Bp[0] += B[o] * (Sh[2][kz] * Sh[1][jy] * No[0][ix]);
          Bp[1] += B[g.cstride + o] * (Sh[2][kz] * No[1][jy] * Sh[0][ix]);
          Bp[2] += B[2 * g.cstride + o] * (No[2][kz] * Sh[1][jy] * Sh[0][ix]);
        }
  }
  IEShape sh;
  sh.setup(g, rn, r0);
  double Ep[3] = {0, 0, 0};
  int m = 0;
#pragma unroll
  for (int cx = 0; cx < 3; cx++) {
    const int cy = (cx + 1) % 3, cz = (cx + 2) % 3;
    int i[3];
    for (i[cx] = 0; i[cx] < 2; i[cx]++)
      for (i[cy] = 0; i[cy] < 3; i[cy]++)
        for (i[cz] = 0; i[cz] < 3; i[cz]++)
          Ep[cx] += E[cx * g.cstride + ie_node(g, sh.start[0] + i[0], sh.start[1] + i[1], sh.start[2] + i[2])] * sh.cache[m++];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) { Ep3[3 * q + c] = Ep[c]; Bp3[3 * q + c] = Bp[c]; }
}

__global__ void __launch_bounds__(kBlock) k_ie_decompose(GridDev g, double* J, long n, const double* alpha,
  const double* v3, const double* rn3, const double* r03)
{
  const long q = (long)blockIdx.x * kBlock + threadIdx.x;
  if (q >= n) return;
  const double rn[3] = {rn3[3 * q], rn3[3 * q + 1], rn3[3 * q + 2]}, r0[3] = {r03[3 * q], r03[3 * q + 1], r03[3 * q + 2]};
  IEShape sh;
  sh.setup(g, rn, r0);
  int m = 0;
#pragma unroll
  for (int cx = 0; cx < 3; cx++) {
    const int cy = (cx + 1) % 3, cz = (cx + 2) % 3;
    const double av = alpha[q] * v3[3 * q + cx];
    int i[3];
    for (i[cx] = 0; i[cx] < 2; i[cx]++)
      for (i[cy] = 0; i[cy] < 3; i[cy]++)
        for (i[cz] = 0; i[cz] < 3; i[cz]++) {
          const double w = av * sh.cache[m++];
          if (w != 0.0)
            unsafeAtomicAdd(&J[cx * g.cstride + ie_node(g, sh.start[0] + i[0], sh.start[1] + i[1], sh.start[2] + i[2])], w);
        }
  }
}

__global__ void __launch_bounds__(kBlock) k_cell_traversal(GridDev g, long n, const double* end3, const double* start3,
  int max_pts, double* pts, int* counts)
{
  const long q = (long)blockIdx.x * kBlock + threadIdx.x;
  if (q >= n) return;
  const double d[3] = {g.dx, g.dy, g.dz};
  double st[3], en[3], dir[3], tt[3], dtt[3];
  int curr[3], last[3], sg[3];
  bool same = true;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    st[c] = start3[3 * q + c];
    en[c] = end3[3 * q + c];
    curr[c] = (int)round(st[c] / d[c]);
    last[c] = (int)round(en[c] / d[c]);
    same = same && curr[c] == last[c];
  }
  double* out = pts + 3L * max_pts * q;
  int np = 0;
  auto push = [&](double x, double y, double z) {
    if (np < max_pts) { out[3 * np] = x; out[3 * np + 1] = y; out[3 * np + 2] = z; }
    ++np;
  };
  push(st[0], st[1], st[2]);
  if (!same) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      dir[c] = en[c] - st[c];
      sg[c] = dir[c] > 0 ? 1 : -1;
      const double nxt = (curr[c] + sg[c] * 0.5) * d[c];
      tt[c] = (dir[c] != 0) ? (nxt - st[c]) / dir[c] : DBL_MAX;
      dtt[c] = (dir[c] != 0) ? d[c] / dir[c] * sg[c] : 0.0;
    }
    while (!(curr[0] == last[0] && curr[1] == last[1] && curr[2] == last[2]) && np <= 64) {
      double t;
      if (tt[0] < tt[1]) {
        if (tt[0] < tt[2]) { t = tt[0]; curr[0] += sg[0]; tt[0] += dtt[0]; }
        else { t = tt[2]; curr[2] += sg[2]; tt[2] += dtt[2]; }
      }
      else {
        if (tt[1] < tt[2]) { t = tt[1]; curr[1] += sg[1]; tt[1] += dtt[1]; }
        else { t = tt[2]; curr[2] += sg[2]; tt[2] += dtt[2]; }
      }
      push(st[0] + dir[0] * t, st[1] + dir[1] * t, st[2] + dir[2] * t);
    }
  }
  push(en[0], en[1], en[2]);
  counts[q] = np;
}

struct DevBuf { // host array <-> device scratch, freed on scope exit
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
};

}  // namespace

}  // namespace xpic

using namespace xpic;

extern "C" {

int xpic_cell_traversal(xpic_ctx* ctx, int64_t n, const double* end3, const double* start3, int max_pts, double* pts,
  int* counts)
{
  XPIC_CHECK(ctx && end3 && start3 && pts && counts && n >= 0 && max_pts >= 2, "bad argument");
  if (n == 0) return 0;
  DevBuf e, s, p, k;
  XPIC_HIP(hipMalloc(&e.p, 24 * n)); XPIC_HIP(hipMalloc(&s.p, 24 * n));
  XPIC_HIP(hipMalloc(&p.p, 24 * n * max_pts)); XPIC_HIP(hipMalloc(&k.p, 4 * n));
  XPIC_HIP(hipMemcpyAsync(e.p, end3, 24 * n, hipMemcpyHostToDevice, ctx->stream));
  XPIC_HIP(hipMemcpyAsync(s.p, start3, 24 * n, hipMemcpyHostToDevice, ctx->stream));
  XPIC_HIP(hipMemsetAsync(p.p, 0, 24 * n * max_pts, ctx->stream));
  hipLaunchKernelGGL(k_cell_traversal, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, ctx->g, (long)n,
    (const double*)e.p, (const double*)s.p, max_pts, (double*)p.p, (int*)k.p);
  XPIC_HIP(hipGetLastError());
  XPIC_HIP(hipMemcpyAsync(pts, p.p, 24 * n * max_pts, hipMemcpyDeviceToHost, ctx->stream));
  XPIC_HIP(hipMemcpyAsync(counts, k.p, 4 * n, hipMemcpyDeviceToHost, ctx->stream));
  XPIC_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

int xpic_implicit_esirkepov_interpolate(xpic_ctx* ctx, int64_t n, const double* rn3, const double* r03, double* Ep3,
  double* Bp3)
{
  XPIC_CHECK(ctx && rn3 && r03 && Ep3 && Bp3 && n >= 0, "bad argument");
  if (n == 0) return 0;
  XPIC_CALL(halo_fill(ctx, ctx->field[XPIC_E]));
  XPIC_CALL(halo_fill(ctx, ctx->field[XPIC_B]));
  DevBuf a, b, e, m;
  XPIC_HIP(hipMalloc(&a.p, 24 * n)); XPIC_HIP(hipMalloc(&b.p, 24 * n));
  XPIC_HIP(hipMalloc(&e.p, 24 * n)); XPIC_HIP(hipMalloc(&m.p, 24 * n));
  XPIC_HIP(hipMemcpyAsync(a.p, rn3, 24 * n, hipMemcpyHostToDevice, ctx->stream));
  XPIC_HIP(hipMemcpyAsync(b.p, r03, 24 * n, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_ie_interpolate, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, ctx->g,
    ctx->field[XPIC_E], ctx->field[XPIC_B], (long)n, (const double*)a.p, (const double*)b.p, (double*)e.p, (double*)m.p);
  XPIC_HIP(hipGetLastError());
  XPIC_HIP(hipMemcpyAsync(Ep3, e.p, 24 * n, hipMemcpyDeviceToHost, ctx->stream));
  XPIC_HIP(hipMemcpyAsync(Bp3, m.p, 24 * n, hipMemcpyDeviceToHost, ctx->stream));
  XPIC_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

int xpic_implicit_esirkepov_decompose(xpic_ctx* ctx, int64_t n, const double* alpha, const double* v3, const double* rn3,
  const double* r03, int field)
{
  XPIC_CHECK(ctx && alpha && v3 && rn3 && r03 && n >= 0, "bad argument");
  XPIC_CHECK(field >= 0 && field < XPIC_NFIELDS && ctx->field[field], "no such field vector");
  if (n == 0) return 0;
  DevBuf al, v, a, b;
  XPIC_HIP(hipMalloc(&al.p, 8 * n)); XPIC_HIP(hipMalloc(&v.p, 24 * n));
  XPIC_HIP(hipMalloc(&a.p, 24 * n)); XPIC_HIP(hipMalloc(&b.p, 24 * n));
  XPIC_HIP(hipMemcpyAsync(al.p, alpha, 8 * n, hipMemcpyHostToDevice, ctx->stream));
  XPIC_HIP(hipMemcpyAsync(v.p, v3, 24 * n, hipMemcpyHostToDevice, ctx->stream));
  XPIC_HIP(hipMemcpyAsync(a.p, rn3, 24 * n, hipMemcpyHostToDevice, ctx->stream));
  XPIC_HIP(hipMemcpyAsync(b.p, r03, 24 * n, hipMemcpyHostToDevice, ctx->stream));
  // deposit into a zeroed scratch vector (ghost planes included), fold the ghosts, then add: DMLocalToGlobal(ADD)
  double* tmp = ctx->field[XPIC_W2];
  XPIC_CHECK(tmp != ctx->field[field], "XPIC_W2 is the scratch vector of this call");
  XPIC_CALL(vec_set(ctx, tmp, 0.0));
  XPIC_HIP(hipMemsetAsync(tmp, 0, sizeof(double) * ctx->nvec, ctx->stream));
  hipLaunchKernelGGL(k_ie_decompose, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, ctx->g, tmp,
    (long)n, (const double*)al.p, (const double*)v.p, (const double*)a.p, (const double*)b.p);
  XPIC_HIP(hipGetLastError());
  XPIC_CALL(halo_add(ctx, tmp, 3));
  XPIC_CALL(vec_axpy(ctx, ctx->field[field], 1.0, tmp));
  XPIC_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

}  // extern "C"
