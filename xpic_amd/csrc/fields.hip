// fields.hip -- grid-side kernels: BLAS-1 (K14), curl (K11), matM (K12), matL / matA SpMV (K13),
// layout conversion at the boundary, two-stage reductions.  All HBM-bound; see DESIGN.md for the
// algorithmic byte counts each kernel is measured against.
#include <cmath>
#include <utility>

#include "common.h"
#include "lstencil.h"

namespace xpic {

int experiment_fields() { return XPIC_TU_EXPERIMENT; }

namespace {

constexpr int kBlock = 256;

inline dim3 ew_grid(const GridDev& g)
{
  long blocks = (g.nown + kBlock * 2 - 1) / (kBlock * 2);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  return dim3((unsigned)blocks, 3, 1);
}

// ---- element-wise kernels over the owned planes of the 3 components (grid.y = component) ----------
template <class F>
__global__ void __launch_bounds__(kBlock) k_ew(GridDev g, F f)
{
  const long off = (long)blockIdx.y * g.cstride + (long)g.G * g.plane;
  const long n = g.nown;
  const long stride = (long)gridDim.x * kBlock;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) f(off + i);
}

struct FSet { double* y; double a; __device__ void operator()(long i) const { y[i] = a; } };
struct FCopy { double* y; const double* x; __device__ void operator()(long i) const { y[i] = x[i]; } };
struct FAxpy { double* y; double a; const double* x; __device__ void operator()(long i) const { y[i] += a * x[i]; } };
struct FAxpby { double* y; double a, b; const double* x; __device__ void operator()(long i) const { y[i] = a * x[i] + b * y[i]; } };
struct FWaxpby { double* w; double a; const double* x; double b; const double* y; __device__ void operator()(long i) const { w[i] = a * x[i] + b * y[i]; } };
struct FScaleTo { double* y; double a; const double* x; __device__ void operator()(long i) const { y[i] = a * x[i]; } };

template <class F>
int launch_ew(xpic_ctx* c, F f)
{
  hipLaunchKernelGGL(k_ew<F>, ew_grid(c->g), dim3(kBlock), 0, c->stream, c->g, f);
  XPIC_HIP(hipGetLastError());
  return 0;
}

// ---- reductions -------------------------------------------------------------------------------------
__device__ inline double wave_sum(double v)
{
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// block-reduce NV values, thread 0 writes partial[j*nblocks + block]
template <int NV>
__device__ inline void block_reduce_store(double (&acc)[NV], double* partial, int nblocks, int block)
{
  __shared__ double sm[NV][kBlock / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    double v = wave_sum(acc[j]);
    if (lane == 0) sm[j][wave] = v;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    double v = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) v += sm[threadIdx.x][w];
    partial[(long)threadIdx.x * nblocks + block] = v;
  }
}

struct VPtrs { const double* p[8]; };
struct HVals { double h[8]; };

// partial[j][blk] = sum_i w[i] * V_j[i];  WW: one more row, sum_i w[i]^2 (the norm of the Gram-Schmidt step comes out of the
// same reduction: |w - sum h_j V_j|^2 = w.w - sum h_j^2 for an orthonormal V)
template <int NV, bool WW>
__global__ void __launch_bounds__(kBlock) k_mdot(GridDev g, const double* w, VPtrs V, double* partial)
{
  double acc[NV + (WW ? 1 : 0)];
#pragma unroll
  for (int j = 0; j < NV + (WW ? 1 : 0); ++j) acc[j] = 0.0;
  const long off = (long)blockIdx.y * g.cstride + (long)g.G * g.plane;
  const long stride = (long)gridDim.x * kBlock;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < g.nown; i += stride) {
    double wi = w[off + i];
#pragma unroll
    for (int j = 0; j < NV; ++j) acc[j] += wi * V.p[j][off + i];
    if (WW) acc[NV] += wi * wi;
  }
  block_reduce_store<NV + (WW ? 1 : 0)>(acc, partial, gridDim.x * gridDim.y, blockIdx.y * gridDim.x + blockIdx.x);
}

// out = (w - sum_j h_j V_j) * scale (out may be w itself); optionally partial[blk] = sum out^2
template <int NV, bool NORM>
__global__ void __launch_bounds__(kBlock) k_maxpy(GridDev g, const double* w, VPtrs V, HVals h, double* partial, double* out,
  double scale)
{
  double acc[1] = {0.0};
  const long off = (long)blockIdx.y * g.cstride + (long)g.G * g.plane;
  const long stride = (long)gridDim.x * kBlock;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < g.nown; i += stride) {
    double wi = w[off + i];
#pragma unroll
    for (int j = 0; j < NV; ++j) wi -= h.h[j] * V.p[j][off + i];
    wi *= scale;
    out[off + i] = wi;
    if (NORM) acc[0] += wi * wi;
  }
  if (NORM) block_reduce_store<1>(acc, partial, gridDim.x * gridDim.y, blockIdx.y * gridDim.x + blockIdx.x);
}

// sum of squares and per-component sums of one field: partial rows {sq, sum_c} per component block
__global__ void __launch_bounds__(kBlock) k_stats(GridDev g, const double* f, double* partial)
{
  double acc[2] = {0.0, 0.0};
  const long off = (long)blockIdx.y * g.cstride + (long)g.G * g.plane;
  const long stride = (long)gridDim.x * kBlock;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < g.nown; i += stride) {
    double v = f[off + i];
    acc[0] += v * v;
    acc[1] += v;
  }
  block_reduce_store<2>(acc, partial, gridDim.x * gridDim.y, blockIdx.y * gridDim.x + blockIdx.x);
}

// out[j*nseg + s] = sum over segment s of partial[j*nblocks + ...]; one block per (j, s)
__global__ void __launch_bounds__(kBlock) k_reduce_final(const double* partial, int nblocks, int nseg, double* out)
{
  const int j = blockIdx.x / nseg, s = blockIdx.x % nseg;
  const int seg = nblocks / nseg;
  double v = 0;
  for (int i = threadIdx.x; i < seg; i += kBlock) v += partial[(long)j * nblocks + s * seg + i];
  __shared__ double sm[kBlock / 64];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0;
    for (int w = 0; w < kBlock / 64; ++w) t += sm[w];
    out[blockIdx.x] = t;
  }
}

inline dim3 red_grid(const GridDev& g)
{
  long blocks = (g.nown + kBlock * 4 - 1) / (kBlock * 4);
  long cap = kRedBlocks / 3;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return dim3((unsigned)blocks, 3, 1);
}

int finish_reduce(xpic_ctx* c, int nv, int nblocks, int nseg, double* host_out)
{
  hipLaunchKernelGGL(k_reduce_final, dim3(nv * nseg), dim3(kBlock), 0, c->stream, c->red_partial, nblocks, nseg, c->red_out);
  XPIC_HIP(hipGetLastError());
  XPIC_CALL(comm_allreduce_sum(c, c->red_out, nv * nseg)); // the MPI_Allreduce inside VecDot/VecMDot/VecNorm
  XPIC_HIP(hipMemcpyAsync(c->red_host, c->red_out, sizeof(double) * nv * nseg, hipMemcpyDeviceToHost, c->stream));
  XPIC_HIP(hipStreamSynchronize(c->stream));
  for (int i = 0; i < nv * nseg; ++i) host_out[i] = c->red_host[i];
  return 0;
}

// ---- curl: Rotor::fill_stencil + values (src/utils/operators.cpp:155-215) ---------------------------
// positive shift: (rot F)_x = (Fz[y+1]-Fz[y])/dy - (Fy[z+1]-Fy[z])/dz, ... ; negative: backward differences.
template <int SIGN, class T = double> // T: storage type of the vector (arithmetic is fp64)
__device__ inline void rot_at(const GridDev& g, const T* F, int x, int y, int z, double& rx, double& ry, double& rz)
{
  const double ix = g.inv[0], iy = g.inv[1], iz = g.inv[2]; // 1 / d, formed once on the host (the same quotient)
  const T* Fx = F;
  const T* Fy = F + g.cstride;
  const T* Fz = F + 2 * g.cstride;
  const long c0 = g.nodew(x, y, z);
  if (SIGN > 0) {
    const long xp = g.nodew(x + 1, y, z), yp = g.nodew(x, y + 1, z), zp = g.nodew(x, y, z + 1);
    rx = +iy * Fz[yp] - iy * Fz[c0] - iz * Fy[zp] + iz * Fy[c0];
    ry = -ix * Fz[xp] + ix * Fz[c0] + iz * Fx[zp] - iz * Fx[c0];
    rz = +ix * Fy[xp] - ix * Fy[c0] - iy * Fx[yp] + iy * Fx[c0];
  }
  else {
    const long xm = g.nodew(x - 1, y, z), ym = g.nodew(x, y - 1, z), zm = g.nodew(x, y, z - 1);
    rx = +iy * Fz[c0] - iy * Fz[ym] - iz * Fy[c0] + iz * Fy[zm];
    ry = -ix * Fz[c0] + ix * Fz[xm] + iz * Fx[c0] - iz * Fx[zm];
    rz = +ix * Fy[c0] - ix * Fy[xm] - iy * Fx[c0] + iy * Fx[ym];
  }
}

template <int SIGN>
__global__ void __launch_bounds__(kBlock) k_rot(GridDev g, double alpha, const double* F, double* out, int add)
{
  const long n = g.nown;
  const long stride = (long)gridDim.x * kBlock;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    int x = (int)(i % g.nx), y = (int)((i / g.nx) % g.ny), z = (int)(i / g.plane);
    double rx, ry, rz;
    rot_at<SIGN>(g, F, x, y, z, rx, ry, rz);
    const long o = g.node(x, y, g.wz(z));
    if (add) {
      out[o] += alpha * rx;
      out[o + g.cstride] += alpha * ry;
      out[o + 2 * g.cstride] += alpha * rz;
    }
    else {
      out[o] = alpha * rx;
      out[o + g.cstride] = alpha * ry;
      out[o + 2 * g.cstride] = alpha * rz;
    }
  }
}

// (matM x)_c = 2 x_c + 0.5 dt^2 (rot- rot+ x)_c   (src/impls/ecsim/simulation.cpp:544-551)
// rot+ is evaluated on the fly at the 2 neighbours each backward difference needs: a 13-point stencil.
template <class T = double>
__device__ inline void matM_at(const GridDev& g, const T* F, int x, int y, int z, double& mx, double& my, double& mz)
{
  const double ix = g.inv[0], iy = g.inv[1], iz = g.inv[2]; // 1 / d, formed once on the host (the same quotient)
  double ax, ay, az; // rot+ at (x,y,z)
  double bx, by, bz; // rot+ at (x-1,y,z)
  double cx, cy, cz; // rot+ at (x,y-1,z)
  double dx_, dy_, dz_; // rot+ at (x,y,z-1)
  rot_at<+1, T>(g, F, x, y, z, ax, ay, az);
  rot_at<+1, T>(g, F, x - 1, y, z, bx, by, bz);
  rot_at<+1, T>(g, F, x, y - 1, z, cx, cy, cz);
  rot_at<+1, T>(g, F, x, y, z - 1, dx_, dy_, dz_);
  const double s = 0.5 * g.dt * g.dt;
  const long c0 = g.nodew(x, y, z);
  // negative-shift curl of G = rot+ F (operators.cpp:196-213)
  double rx = +iy * az - iy * cz - iz * ay + iz * dy_;
  double ry = -ix * az + ix * bz + iz * ax - iz * dx_;
  double rz = +ix * ay - ix * by - iy * ax + iy * cx;
  mx = 2.0 * (double)F[c0] + s * rx;
  my = 2.0 * (double)F[c0 + g.cstride] + s * ry;
  mz = 2.0 * (double)F[c0 + 2 * g.cstride] + s * rz;
}

// owned planes [z0r, z0r + nzr) (the whole slab, or its interior / boundary planes when the apply runs beside the halo exchange)
__global__ void __launch_bounds__(kBlock) k_matM(GridDev g, const double* F, double* out, int add, int z0r, int nzr)
{
  const long n = (long)nzr * g.plane;
  const long stride = (long)gridDim.x * kBlock;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    int x = (int)(i % g.nx), y = (int)((i / g.nx) % g.ny), z = z0r + (int)(i / g.plane);
    double mx, my, mz;
    matM_at(g, F, x, y, z, mx, my, mz);
    const long o = g.node(x, y, g.wz(z));
    if (add) { out[o] += mx; out[o + g.cstride] += my; out[o + 2 * g.cstride] += mz; }
    else { out[o] = mx; out[o + g.cstride] = my; out[o + 2 * g.cstride] = mz; }
  }
}

// ---- fused CG kernels on matM (krylov.hip: cg).  One CG iteration moves 11 V: these two kernels (2 V + 6 V) and
// the p = r + beta p update (3 V) -- SURVEY 8(d)'s "maximally fused" count -- instead of the 14 V of one BLAS-1 call
// per line of the textbook algorithm.
// Ap = matM p and partial[blk] = sum p . Ap (owned nodes; a node's three components are summed together)
__global__ void __launch_bounds__(kBlock) k_cg_apply_dot(GridDev g, const double* __restrict__ p, double* __restrict__ Ap,
  double* partial)
{
  double acc[1] = {0.0};
  const long stride = (long)gridDim.x * kBlock;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < g.nown; i += stride) {
    int x = (int)(i % g.nx), y = (int)((i / g.nx) % g.ny), z = (int)(i / g.plane);
    double m[3];
    matM_at(g, p, x, y, z, m[0], m[1], m[2]);
    const long o = g.node(x, y, g.wz(z));
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      Ap[o + c * g.cstride] = m[c];
      acc[0] += p[o + c * g.cstride] * m[c];
    }
  }
  block_reduce_store<1>(acc, partial, gridDim.x, blockIdx.x);
}

// x += alpha p ; r -= alpha Ap ; partial[blk] = sum r . r (after the update)
__global__ void __launch_bounds__(kBlock) k_cg_update(GridDev g, double alpha, const double* __restrict__ p,
  const double* __restrict__ Ap, double* __restrict__ x, double* __restrict__ r, double* partial)
{
  double acc[1] = {0.0};
  const long off = (long)blockIdx.y * g.cstride + (long)g.G * g.plane;
  const long stride = (long)gridDim.x * kBlock;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < g.nown; i += stride) {
    x[off + i] += alpha * p[off + i];
    const double rv = r[off + i] - alpha * Ap[off + i];
    r[off + i] = rv;
    acc[0] += rv * rv;
  }
  block_reduce_store<1>(acc, partial, gridDim.x * gridDim.y, blockIdx.y * gridDim.x + blockIdx.x);
}

// ---- Chebyshev step on matM: res = r - matM z_in ; d = cd d + cr res ; z_out = z_in + d.
// FIRST: the start z_0 = d_0 = r / theta is never stored: matM z_0 = (matM r) / theta, everything comes from r.
template <bool FIRST>
__global__ void __launch_bounds__(kBlock) k_cheb(GridDev g, const double* __restrict__ r, const double* __restrict__ zin,
  double* __restrict__ d, double* __restrict__ zout, double cd, double cr, double itheta)
{
  const long n = g.nown;
  const long stride = (long)gridDim.x * kBlock;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    int x = (int)(i % g.nx), y = (int)((i / g.nx) % g.ny), z = (int)(i / g.plane);
    double m[3];
    matM_at(g, FIRST ? r : zin, x, y, z, m[0], m[1], m[2]);
    const long o = g.node(x, y, g.wz(z));
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const long oc = o + c * g.cstride;
      const double rv = r[oc];
      const double z0 = FIRST ? rv * itheta : zin[oc];
      const double dn = cd * (FIRST ? z0 : d[oc]) + cr * (rv - (FIRST ? m[c] * itheta : m[c]));
      d[oc] = dn;
      zout[oc] = z0 + dn;
    }
  }
}

// The same step on fp32 copies of the vectors (fp64 arithmetic): the preconditioner of a flexible GMRES may be inexact,
// and the iteration is pure memory traffic (5 V per step).  FIRST reads the fp64 input and leaves its fp32 copy for the
// later steps; LAST writes the fp64 result.
template <bool FIRST, bool LAST>
__global__ void __launch_bounds__(kBlock) k_cheb32(GridDev g, const double* __restrict__ r64, float* __restrict__ r32,
  const float* __restrict__ zin, float* __restrict__ d, float* __restrict__ zout, double* __restrict__ out64, double cd,
  double cr, double itheta)
{
  const long n = g.nown;
  const long stride = (long)gridDim.x * kBlock;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    int x = (int)(i % g.nx), y = (int)((i / g.nx) % g.ny), z = (int)(i / g.plane);
    double m[3];
    if (FIRST) matM_at<double>(g, r64, x, y, z, m[0], m[1], m[2]);
    else matM_at<float>(g, zin, x, y, z, m[0], m[1], m[2]);
    const long o = g.node(x, y, g.wz(z));
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const long oc = o + c * g.cstride;
      const double rv = FIRST ? r64[oc] : (double)r32[oc];
      const double z0 = FIRST ? rv * itheta : (double)zin[oc];
      const double dn = cd * (FIRST ? z0 : (double)d[oc]) + cr * (rv - (FIRST ? m[c] * itheta : m[c]));
      if (FIRST && !LAST) r32[oc] = (float)rv;
      if (LAST) out64[oc] = z0 + dn;
      else {
        d[oc] = (float)dn;
        zout[oc] = (float)(z0 + dn);
      }
    }
  }
}


// ---- matL / matA SpMV ---------------------------------------------------------------------------
// matL[c1][z][y][x/4][k][x%4] (common.h): one thread per row (c1, node), lane = x.  A wave reads 16 x-blocks; the
// four coefficients k..k+3 of a block are one 128-byte line, consumed by four consecutive loads of the wave.  The
// operand vector comes out of L1/L2 (every element is used by 123 rows).  The stencil (c2, dx, dy, dz) of every k is a compile-time constant: the term list is expanded with
// an integer_sequence so that all address arithmetic folds into immediates and wave-uniform row bases.
// (Two neighbouring rows per lane -- 16-byte coefficient loads, the operand taps of a stencil line as three 16-byte loads,
// 2.2 x fewer vector-memory instructions per row -- was built and measured SLOWER: 11.6 against 10.5 ms per apply.)
constexpr int kRowX = 64, kRowY = 4;
constexpr int kBandY = 4; // y-chunks per band: 16 rows

// The operand's neighbourhood of a workgroup's 4 x 64 rows of component C1, staged in LDS: per column component c2 the
// nodes the rows' stencil reaches (lstencil.h: lrange), (n_z) x (n_y + 3) x (n_x + 63) values.  Before (rounds 2 - 4)
// every lane fetched its 123 operand values through the vector L1 beside the 123 coefficients it streams: measured on
// the kernel itself, the operand loads alone cost 2.0 of its 10.1 ms (the coefficient stream needs that L1 -- four
// consecutive k share a 128-byte line -- and a pure stream of matL in this layout runs at 6.4 TB/s, tools/ubench/matl_layout.hip).
template <int C1>
struct XTile {
  static constexpr int nx(int c2) { return lrange(C1, c2).n[0] + kRowX - 1; }
  static constexpr int ny(int c2) { return lrange(C1, c2).n[1] + kRowY - 1; }
  static constexpr int nz(int c2) { return lrange(C1, c2).n[2]; }
  static constexpr int off(int c2)
  {
    int o = 0;
    for (int c = 0; c < c2; ++c) o += nx(c) * ny(c) * nz(c);
    return o;
  }
  static constexpr int size = off(3);
  // entry of (component c2, node offset d) for the row of thread (0, 0); thread (tx, ty) adds ty * nx(c2) + tx
  static constexpr int at(int c2, int dx, int dy, int dz)
  {
    const LRange r = lrange(C1, c2);
    return off(c2) + ((dz - r.lo[2]) * ny(c2) + (dy - r.lo[1])) * nx(c2) + (dx - r.lo[0]);
  }
};
constexpr int kXTileMax = XTile<0>::size > XTile<1>::size ? (XTile<0>::size > XTile<2>::size ? XTile<0>::size : XTile<2>::size)
                                                          : (XTile<1>::size > XTile<2>::size ? XTile<1>::size : XTile<2>::size);
static_assert(kXTileMax * 8 <= 40960 - 1280, "four workgroups of k_matA per CU: 160 KB of LDS");

// one periodic fold, as GridDev::wx; what is still outside belongs to lanes / rows beyond the grid's edge (a partial
// chunk): nobody reads those entries, any valid index will do
__device__ __forceinline__ int fold_or_zero(int v, int n)
{
  v = v < 0 ? v + n : (v >= n ? v - n : v);
  return v < n ? v : 0;
}

template <int C1, int C2>
__device__ __forceinline__ void xtile_fill_c(const GridDev& g, const double* __restrict__ X, double* tile, int x0, int y0, int z, int tx, int w)
{
  constexpr LRange r = lrange(C1, C2);
  constexpr int NX = XTile<C1>::nx(C2), NY = XTile<C1>::ny(C2), NZ = XTile<C1>::nz(C2), OFF = XTile<C1>::off(C2);
  static_assert(NX > kRowX && NX <= 2 * kRowX, "a tile row is one full and one partial pass of the wave");
  const int gx0 = fold_or_zero(x0 + r.lo[0] + tx, g.nx), gx1 = fold_or_zero(x0 + r.lo[0] + kRowX + tx, g.nx);
  for (int row = w; row < NY * NZ; row += kRowY) { // (w is wave-uniform: scalar arithmetic)
    const int pz = row / NY, py = row - pz * NY;
    const double* src = X + C2 * g.cstride + (long)g.wz(z + r.lo[2] + pz) * g.plane + (long)fold_or_zero(y0 + r.lo[1] + py, g.ny) * g.nx;
    tile[OFF + row * NX + tx] = src[gx0];
    if (tx < NX - kRowX) tile[OFF + row * NX + kRowX + tx] = src[gx1];
  }
}

#ifndef MATA_GROUP
#define MATA_GROUP 12 // (experiment builds: 8 / 12 / 16 / 20 x depth 2: 9.3 / 8.8 / 8.8 / 9.3 ms per apply at 256^3; 8 x 3 and 8 x 4: 9.1; 12 x 3: 8.8; 24 x 1: 9.5)
#endif
#ifndef MATA_DEPTH
#define MATA_DEPTH 2
#endif
constexpr int kLGroup = MATA_GROUP, kLDepth = MATA_DEPTH;

// The 123 terms of a row in groups of kLGroup, the coefficients of kLDepth groups in flight: group j takes the lane's offset
// into the row block from copy j % kLDepth, and that copy passes through an opaque statement together with the sums at
// the end of its group, so that the requests of group j + kLDepth cannot be issued before the products of group j, while
// those of the groups between are on their way.  The tile bases pass through every group's statement: a group's operand
// values are read out of LDS when the group before it is done.  (Left to itself the scheduler requests all 123
// coefficients and all 123 operand values first and spills a thousand registers; a scheduling barrier does not hold it.)
struct RowAddr {
  unsigned xl8[kLDepth]; // byte offset of (x-block, x % 4) inside the row block
  int lb[3];             // ty * nx(c2) + tx
};

template <int C1, int K>
__device__ __forceinline__ void lterm(double (&acc)[2], const char* Lb, RowAddr& a, const double* tile)
{
  constexpr LEntry e = ldecode(C1, K);
  constexpr int toff = XTile<C1>::at(e.c2, e.d[0], e.d[1], e.d[2]);
  constexpr int set = (K / kLGroup) % kLDepth;
  const double xv = tile[a.lb[e.c2] + toff];
  const double lv = *reinterpret_cast<const double*>(Lb + (size_t)K * 32 + a.xl8[set]);
  acc[K & 1] += lv * xv;
  if (K % kLGroup == kLGroup - 1)
    asm volatile("" : "+v"(a.xl8[set]), "+v"(a.lb[0]), "+v"(a.lb[1]), "+v"(a.lb[2]) : "v"(acc[0]), "v"(acc[1]));
}

template <int C1, int... Ks>
__device__ __forceinline__ double row_apply(std::integer_sequence<int, Ks...>, const GridDev& g, const double* __restrict__ L,
  const double* tile, const int (&lb)[3], int x, int y, int z)
{
  const char* Lb = reinterpret_cast<const char*>(L + g.lindex(C1, z + (g.G ? 1 : 0), y, 0, 0)); // wave-uniform: the row block (c1, z, y)
  RowAddr a;
  for (int d = 0; d < kLDepth; ++d) a.xl8[d] = (unsigned)(x >> 2) * (8u * kLBlock) + 8u * (unsigned)(x & 3);
  for (int c = 0; c < 3; ++c) a.lb[c] = lb[c];
  double acc[2] = {0.0, 0.0};
  (lterm<C1, Ks>(acc, Lb, a, tile), ...);
  return acc[0] + acc[1];
}

// one component of matM x = 2 x + 0.5 dt^2 rot(-) rot(+) x, written for component C with the cyclic
// axes A = C+1, B = C+2:  (rot- G)_C = d-_A G_B - d-_B G_A,  G_B = d+_C F_A - d+_A F_C,  G_A = d+_B F_C - d+_C F_B;
// at(comp, ox, oy, oz): the operand's component `comp` at the node offset (ox, oy, oz) from the row's node
template <int C, class At>
__device__ __forceinline__ double matM_comp(const GridDev& g, At at)
{
  constexpr int A = (C + 1) % 3, B = (C + 2) % 3;
  const double ih[3] = {g.inv[0], g.inv[1], g.inv[2]}; // 1 / d, formed once on the host (the same quotient)
  auto sh = [&](int axis, int s, int& ox, int& oy, int& oz) { (axis == 0 ? ox : (axis == 1 ? oy : oz)) += s; };
  // G_comp at offset (ox,oy,oz): forward differences
  auto dplus = [&](int comp, int axis, int ox, int oy, int oz) {
    int px = ox, py = oy, pz = oz;
    sh(axis, +1, px, py, pz);
    return (at(comp, px, py, pz) - at(comp, ox, oy, oz)) * ih[axis];
  };
  auto GB = [&](int ox, int oy, int oz) { return dplus(A, C, ox, oy, oz) - dplus(C, A, ox, oy, oz); };
  auto GA = [&](int ox, int oy, int oz) { return dplus(C, B, ox, oy, oz) - dplus(B, C, ox, oy, oz); };
  int ax = 0, ay = 0, az = 0, bx = 0, by = 0, bz = 0;
  sh(A, -1, ax, ay, az);
  sh(B, -1, bx, by, bz);
  const double r = (GB(0, 0, 0) - GB(ax, ay, az)) * ih[A] - (GA(0, 0, 0) - GA(bx, by, bz)) * ih[B];
  return 2.0 * at(C, 0, 0, 0) + (0.5 * g.dt * g.dt) * r;
}

// the rows of component C1 of one workgroup: stage the operand, then every live thread its row
template <bool WITH_M, int C1>
__device__ __forceinline__ void rows_of(const GridDev& g, const double* __restrict__ L, const double* __restrict__ X,
  double* __restrict__ Y, double* tile, int add, int x0, int y0, int z, int tx, int ty)
{
  xtile_fill_c<C1, 0>(g, X, tile, x0, y0, z, tx, ty);
  xtile_fill_c<C1, 1>(g, X, tile, x0, y0, z, tx, ty);
  xtile_fill_c<C1, 2>(g, X, tile, x0, y0, z, tx, ty);
  __syncthreads();
  const int x = x0 + tx, y = y0 + ty;
  if (x >= g.nx || y >= g.ny) return;
  const int lb[3] = {ty * XTile<C1>::nx(0) + tx, ty * XTile<C1>::nx(1) + tx, ty * XTile<C1>::nx(2) + tx};
  double r = 0.0;
  // (matM's 13 taps lie inside matL's pattern: the same tile serves them)
  if (WITH_M) r = matM_comp<C1>(g, [&](int comp, int ox, int oy, int oz) { return tile[lb[comp] + XTile<C1>::at(comp, ox, oy, oz)]; });
  r += row_apply<C1>(std::make_integer_sequence<int, kLStencil>{}, g, L, tile, lb, x, y, z);
  const long o = C1 * g.cstride + g.node(x, y, g.wz(z));
  if (add) Y[o] += r;
  else Y[o] = r;
}

template <bool WITH_M>
__global__ void __launch_bounds__(kRowX* kRowY, 4) k_matA(GridDev g, const double* __restrict__ L,
  const double* __restrict__ X, double* __restrict__ Y, int add, int z0r, int nzr, int zsplit)
{
  // rows of the owned planes [z0r, z0r + nzr): the whole slab, or its interior / boundary planes (matA_apply_overlapped)
  // 1-D grid.  Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an L2).  XCD r sweeps its own
  // run of z-planes, and inside the run a band of 16 y-rows at a time along z: the operand footprint of a band
  // (20 rows x 5 planes x 3 components) stays in that XCD's 4 MiB L2 while the coefficient streams pass through.
  // Order inside a band position: component fastest (the three rows of a node share their operand footprint), x, y.
  __shared__ double tile[kXTileMax];
  const int nxc = (g.nx + kRowX - 1) / kRowX, nyc = (g.ny + kRowY - 1) / kRowY;
  const int nyt = (nyc + kBandY - 1) / kBandY;         // y-bands
  // the 8 XCDs split the planes zsplit ways and the y-bands 8 / zsplit ways (row_split: whatever leaves the least idle;
  // 8 x 1 for a whole box, 4 x 2 for the 28 interior planes of a 32-plane slab, where 8 runs of 4 planes left one XCD idle)
  const int ysplit = 8 / zsplit;
  const int Pz = (nzr + zsplit - 1) / zsplit;          // planes per XCD run
  const int Pb = (nyt + ysplit - 1) / ysplit;          // y-bands per XCD
  const int per_z = kBandY * nxc * 3, per_band = Pz * per_z;
  const int xcd = blockIdx.x % 8;
  const long q = blockIdx.x / 8;
  const int ytl = (int)(q / per_band);
  const int yt = (xcd / zsplit) * Pb + ytl;
  const int rem = (int)(q % per_band);
  const int zr = (xcd % zsplit) * Pz + rem / per_z;
  const int z = z0r + zr;
  const int rem2 = rem % per_z;
  const int c1 = rem2 % 3;
  const int x0 = ((rem2 / 3) % nxc) * kRowX;
  const int y0 = (yt * kBandY + rem2 / (3 * nxc)) * kRowY;
  if (ytl >= Pb || yt >= nyt || zr >= nzr) return; // (the whole workgroup: nothing here depends on the thread)
  // the wave index is the same for the 64 lanes of a wave (blockDim.x == 64): tell the compiler, so that every row base is
  // scalar arithmetic
  const int tx = threadIdx.x, ty = __builtin_amdgcn_readfirstlane(threadIdx.y);
  if (c1 == 0) rows_of<WITH_M, 0>(g, L, X, Y, tile, add, x0, y0, z, tx, ty);
  else if (c1 == 1) rows_of<WITH_M, 1>(g, L, X, Y, tile, add, x0, y0, z, tx, ty);
  else rows_of<WITH_M, 2>(g, L, X, Y, tile, add, x0, y0, z, tx, ty);
}

// Divergence, negative Yee shift (src/utils/operators.cpp:275-333), added into component 0 of `out`;
// then |.|_1 and |.|_2^2 partials of that scalar
__global__ void __launch_bounds__(kBlock) k_div_neg_add(GridDev g, const double* __restrict__ v, double* out)
{
  const double ix = g.inv[0], iy = g.inv[1], iz = g.inv[2]; // 1 / d, formed once on the host (the same quotient)
  const long stride = (long)gridDim.x * kBlock;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < g.nown; i += stride) {
    int x = (int)(i % g.nx), y = (int)((i / g.nx) % g.ny), z = (int)(i / g.plane);
    const long c0 = g.nodew(x, y, z);
    const double dv = +ix * v[c0] - ix * v[g.nodew(x - 1, y, z)] + iy * v[g.cstride + c0] -
      iy * v[g.cstride + g.nodew(x, y - 1, z)] + iz * v[2 * g.cstride + c0] - iz * v[2 * g.cstride + g.nodew(x, y, z - 1)];
    out[c0] += dv;
  }
}

__global__ void __launch_bounds__(kBlock) k_norm12(GridDev g, const double* f, double* partial)
{
  double acc[2] = {0.0, 0.0};
  const long off = (long)g.G * g.plane; // component 0
  const long stride = (long)gridDim.x * kBlock;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < g.nown; i += stride) {
    const double v = f[off + i];
    acc[0] += fabs(v);
    acc[1] += v * v;
  }
  block_reduce_store<2>(acc, partial, gridDim.x, blockIdx.x);
}

// ---- boundary layout conversion: [z][y][x][3] (reference DMDA order) <-> SoA with ghost planes ----
__global__ void __launch_bounds__(kBlock) k_import(GridDev g, const double* aos, double* soa)
{
  const long n = g.nown * 3;
  const long stride = (long)gridDim.x * kBlock;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    long node = i / 3;
    int c = (int)(i % 3);
    soa[c * g.cstride + (long)g.G * g.plane + node] = aos[i];
  }
}

__global__ void __launch_bounds__(kBlock) k_export(GridDev g, const double* soa, double* aos)
{
  const long n = g.nown * 3;
  const long stride = (long)gridDim.x * kBlock;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    long node = i / 3;
    int c = (int)(i % 3);
    aos[i] = soa[c * g.cstride + (long)g.G * g.plane + node];
  }
}

}  // namespace

int vec_set(xpic_ctx* c, double* y, double a) { return launch_ew(c, FSet{y, a}); }
int vec_copy(xpic_ctx* c, double* y, const double* x) { return launch_ew(c, FCopy{y, x}); }
int vec_axpy(xpic_ctx* c, double* y, double a, const double* x) { return launch_ew(c, FAxpy{y, a, x}); }
int vec_axpby(xpic_ctx* c, double* y, double a, double b, const double* x) { return launch_ew(c, FAxpby{y, a, b, x}); }
int vec_waxpby(xpic_ctx* c, double* w, double a, const double* x, double b, const double* y) { return launch_ew(c, FWaxpby{w, a, x, b, y}); }
int vec_scale_to(xpic_ctx* c, double* y, double a, const double* x) { return launch_ew(c, FScaleTo{y, a, x}); }

// out[i] = w . V_i; with ww != nullptr also *ww = w . w, out of the same (single) reduction
int vec_mdot_ww_host(xpic_ctx* c, const double* w, const double* V, int nv, double* out, double* ww)
{
  Timed t(c, "mdot");
  dim3 grid = red_grid(c->g);
  const int nblocks = grid.x * grid.y;
  for (int j0 = 0; j0 < nv; j0 += 8) {
    int m = nv - j0 < 8 ? nv - j0 : 8;
    const bool last = j0 + 8 >= nv && ww; // the last group carries w.w: its row lands behind the nv dot products
    VPtrs P{};
    for (int j = 0; j < m; ++j) P.p[j] = V + (long)(j0 + j) * c->nvec;
    double* part = c->red_partial + (long)j0 * nblocks;
    switch (m) {
#define CASE(N)                                                                                           \
  case N:                                                                                                 \
    if (last) hipLaunchKernelGGL((k_mdot<N, true>), grid, dim3(kBlock), 0, c->stream, c->g, w, P, part);  \
    else hipLaunchKernelGGL((k_mdot<N, false>), grid, dim3(kBlock), 0, c->stream, c->g, w, P, part);      \
    break;
      CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
    }
    XPIC_HIP(hipGetLastError());
  }
  if (!ww) return finish_reduce(c, nv, nblocks, 1, out);
  double tmp[kMaxDots + 1];
  XPIC_CALL(finish_reduce(c, nv + 1, nblocks, 1, tmp));
  for (int i = 0; i < nv; ++i) out[i] = tmp[i];
  *ww = tmp[nv];
  return 0;
}

int vec_mdot_host(xpic_ctx* c, const double* w, const double* V, int nv, double* out)
{
  return vec_mdot_ww_host(c, w, V, nv, out, nullptr);
}

int vec_dot_host(xpic_ctx* c, const double* x, const double* y, double* out)
{
  return vec_mdot_host(c, x, y, 1, out);
}

// out = (w - sum h_i V_i) * scale; out may be w.  With nrm2 != nullptr: *nrm2 = |out|^2 (one more reduction)
int vec_maxpy_scaled(xpic_ctx* c, const double* w, const double* V, int nv, const double* h, double* out, double scale,
  double* nrm2)
{
  Timed t(c, "maxpy");
  dim3 grid = red_grid(c->g);
  const int nblocks = grid.x * grid.y;
  const double* src = w;
  for (int j0 = 0; j0 < nv; j0 += 8) {
    int m = nv - j0 < 8 ? nv - j0 : 8;
    const bool lastg = j0 + 8 >= nv;
    const bool last = lastg && nrm2;
    VPtrs P{};
    HVals H{};
    for (int j = 0; j < m; ++j) { P.p[j] = V + (long)(j0 + j) * c->nvec; H.h[j] = h[j0 + j]; }
    const double sc = lastg ? scale : 1.0;
    switch (m) {
#define CASE(N)                                                                                                \
  case N:                                                                                                      \
    if (last) hipLaunchKernelGGL((k_maxpy<N, true>), grid, dim3(kBlock), 0, c->stream, c->g, src, P, H, c->red_partial, out, sc); \
    else hipLaunchKernelGGL((k_maxpy<N, false>), grid, dim3(kBlock), 0, c->stream, c->g, src, P, H, c->red_partial, out, sc);     \
    break;
      CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
    }
    XPIC_HIP(hipGetLastError());
    src = out; // later groups continue on the partial result
  }
  if (nrm2) return finish_reduce(c, 1, nblocks, 1, nrm2);
  return 0;
}

int vec_maxpy_norm_host(xpic_ctx* c, double* w, const double* V, int nv, const double* h, double* nrm2)
{
  return vec_maxpy_scaled(c, w, V, nv, h, w, 1.0, nrm2);
}

int vec_maxpy(xpic_ctx* c, double* x, const double* V, int nv, const double* y)
{
  double neg[kMaxDots];
  for (int i = 0; i < nv; ++i) neg[i] = -y[i];
  return vec_maxpy_norm_host(c, x, V, nv, neg, nullptr);
}

int field_stats_host(xpic_ctx* c, const double* f, double* sumsq, double* mean3)
{
  dim3 grid = red_grid(c->g);
  const int nblocks = grid.x * grid.y;
  hipLaunchKernelGGL(k_stats, grid, dim3(kBlock), 0, c->stream, c->g, f, c->red_partial);
  XPIC_HIP(hipGetLastError());
  double out[6];
  XPIC_CALL(finish_reduce(c, 2, nblocks, 3, out)); // out[j*3 + comp]
  *sumsq = out[0] + out[1] + out[2];
  mean3[0] = out[3]; mean3[1] = out[4]; mean3[2] = out[5];
  return 0;
}

// Ap = matM p, *pAp = p . Ap  (p's ghost planes must be filled)
int cg_apply_dot_host(xpic_ctx* c, const double* p, double* Ap, double* pAp)
{
  Timed t(c, "matM_apply");
  long blocks = (c->g.nown + kBlock - 1) / kBlock;
  if (blocks > kRedBlocks) blocks = kRedBlocks;
  hipLaunchKernelGGL(k_cg_apply_dot, dim3((unsigned)blocks), dim3(kBlock), 0, c->stream, c->g, p, Ap, c->red_partial);
  XPIC_HIP(hipGetLastError());
  return finish_reduce(c, 1, (int)blocks, 1, pAp);
}

// x += alpha p ; r -= alpha Ap ; *rr = r . r
int cg_update_host(xpic_ctx* c, double alpha, const double* p, const double* Ap, double* x, double* r, double* rr)
{
  dim3 grid = red_grid(c->g);
  hipLaunchKernelGGL(k_cg_update, grid, dim3(kBlock), 0, c->stream, c->g, alpha, p, Ap, x, r, c->red_partial);
  XPIC_HIP(hipGetLastError());
  return finish_reduce(c, 1, grid.x * grid.y, 1, rr);
}

int rot_apply(xpic_ctx* c, int sign, double alpha, const double* x, double* y, bool add)
{
  Timed t(c, "rot_apply");
  long blocks = (c->g.nown + kBlock - 1) / kBlock;
  if (blocks > 65536) blocks = 65536;
  if (sign > 0) hipLaunchKernelGGL(k_rot<+1>, dim3((unsigned)blocks), dim3(kBlock), 0, c->stream, c->g, alpha, x, y, add ? 1 : 0);
  else hipLaunchKernelGGL(k_rot<-1>, dim3((unsigned)blocks), dim3(kBlock), 0, c->stream, c->g, alpha, x, y, add ? 1 : 0);
  XPIC_HIP(hipGetLastError());
  return 0;
}

static int matM_planes(xpic_ctx* c, const double* x, double* y, bool add, int z0r, int nzr)
{
  if (nzr <= 0) return 0;
  long blocks = ((long)nzr * c->g.plane + kBlock - 1) / kBlock;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(k_matM, dim3((unsigned)blocks), dim3(kBlock), 0, c->stream, c->g, x, y, add ? 1 : 0, z0r, nzr);
  XPIC_HIP(hipGetLastError());
  return 0;
}

int matM_apply(xpic_ctx* c, const double* x, double* y, bool add)
{
  Timed t(c, "matM_apply");
  return matM_planes(c, x, y, add, 0, c->g.nzl);
}

// how the 8 XCDs share the rows of nzr planes: zsplit runs of planes x (8 / zsplit) groups of y-bands, the split with
// the smallest largest share (ties: more z runs -- a run marches along z with its operand footprint in L2)
static int row_split(const GridDev& g, int nzr)
{
  const long nyc = (g.ny + kRowY - 1) / kRowY, nyt = (nyc + kBandY - 1) / kBandY;
  int best = 8;
  long cost = -1;
  for (int zs = 8; zs >= 1; zs >>= 1) {
    const long load = ((nzr + zs - 1) / zs) * ((nyt + 8 / zs - 1) / (8 / zs));
    if (cost < 0 || load < cost) { cost = load; best = zs; }
  }
  return best;
}

static dim3 row_grid(const GridDev& g, int nzr, int zsplit)
{
  const long nxc = (g.nx + kRowX - 1) / kRowX, nyc = (g.ny + kRowY - 1) / kRowY;
  const long nyt = (nyc + kBandY - 1) / kBandY;
  const long Pz = (nzr + zsplit - 1) / zsplit, Pb = (nyt + 8 / zsplit - 1) / (8 / zsplit);
  return dim3((unsigned)(8 * Pb * Pz * kBandY * nxc * 3));
}

int matL_apply(xpic_ctx* c, const double* x, double* y, bool add)
{
  Timed t(c, "matL_apply");
  const int zs = row_split(c->g, c->g.nzl);
  hipLaunchKernelGGL((k_matA<false>), row_grid(c->g, c->g.nzl, zs), dim3(kRowX, kRowY), 0, c->stream, c->g, c->matL, x, y,
    add ? 1 : 0, 0, c->g.nzl, zs);
  XPIC_HIP(hipGetLastError());
  return 0;
}

static int matA_planes(xpic_ctx* c, const double* x, double* y, int z0r, int nzr)
{
  if (nzr <= 0) return 0;
  const int zs = row_split(c->g, nzr);
  hipLaunchKernelGGL((k_matA<true>), row_grid(c->g, nzr, zs), dim3(kRowX, kRowY), 0, c->stream, c->g, c->matL, x, y, 0, z0r,
    nzr, zs);
  XPIC_HIP(hipGetLastError());
  return 0;
}

int matA_apply(xpic_ctx* c, const double* x, double* y)
{
  Timed t(c, "matA_apply");
  return matA_planes(c, x, y, 0, c->g.nzl);
}

int halo_post(xpic_ctx* c, double* f, int width);
int halo_wait(xpic_ctx* c);

// y = op(x) on a z-slab with neighbours: the exchange of x's `width` boundary planes is POSTED (packed on the compute
// stream, shipped and unpacked on the communication stream), the rows of the interior planes -- which read no ghost
// plane -- are computed meanwhile, and the rows of the 2 * width boundary planes follow once the ghosts have arrived
// (the VecScatterBegin / local part / VecScatterEnd / off-process part structure of PETSc's MatMult).
int op_apply_overlapped(xpic_ctx* c, bool with_L, double* x, double* y)
{
  const GridDev& g = c->g;
  const int w = with_L ? 2 : 1;
  Timed t(c, with_L ? "matA_apply" : "matM_apply");
  const int nin = g.nzl - 2 * w;
  XPIC_CALL(halo_post(c, x, w));
  if (with_L) XPIC_CALL(matA_planes(c, x, y, w, nin));
  else XPIC_CALL(matM_planes(c, x, y, false, w, nin));
  XPIC_CALL(halo_wait(c));
  const int nb = nin >= 0 ? w : g.nzl / 2;              // slabs thinner than 2 w planes: two halves, no interior
  const int ztop = nin >= 0 ? g.nzl - w : nb;
  if (with_L) { XPIC_CALL(matA_planes(c, x, y, 0, nb)); XPIC_CALL(matA_planes(c, x, y, ztop, g.nzl - ztop)); }
  else { XPIC_CALL(matM_planes(c, x, y, false, 0, nb)); XPIC_CALL(matM_planes(c, x, y, false, ztop, g.nzl - ztop)); }
  return 0;
}

int halo_fill_f32(xpic_ctx* c, float* f, int width);

// z = p_k(matM) r ~ matM^-1 r: k steps of the Chebyshev iteration on [a, b] = [2, 2 + 2 dt^2 sum 1/h^2], the exact
// spectral interval of matM = 2 I + 0.5 dt^2 rot- rot+ on the periodic Yee grid.  A fixed polynomial: a linear
// operator, no inner products (no all-reduce), one 1-plane halo per step.  `z` ends in out; uses c->kry_p[0..2] and
// c->kry_t.  precond kind 1 keeps the iteration's vectors in fp32 (half the traffic; the flexible GMRES around it does
// not care how exact its preconditioner is, krylov.hip), kind 2 in fp64.
int cheb_matM_inverse(xpic_ctx* c, const double* r, double* out, int degree)
{
  Timed t(c, "precond");
  const GridDev& g = c->g;
  const double a = 2.0, b = 2.0 + 2.0 * g.dt * g.dt * (1.0 / (g.dx * g.dx) + 1.0 / (g.dy * g.dy) + 1.0 / (g.dz * g.dz));
  const double theta = 0.5 * (b + a), delta = 0.5 * (b - a), sigma1 = theta / delta;
  if (degree <= 1) return launch_ew(c, FScaleTo{out, 1.0 / theta, r});
  if (c->profiling) c->prof["cheb_steps"].launches += degree - 1;
  double rho = 1.0 / sigma1;
  long blocks = (g.nown + kBlock - 1) / kBlock;
  if (blocks > 65536) blocks = 65536;
  const dim3 grid((unsigned)blocks), block(kBlock);
  if (c->precond != 2) {
    float* d = (float*)c->kry_p[0];
    float* z0 = (float*)c->kry_p[1];
    float* z1 = (float*)c->kry_p[2];
    float* r32 = (float*)c->kry_t;
    for (int i = 1; i < degree; ++i) {
      const double rho_new = 1.0 / (2.0 * sigma1 - rho);
      const double cd = rho_new * rho, cr = 2.0 * rho_new / delta, it = 1.0 / theta;
      const bool first = i == 1, last = i == degree - 1;
      if (first) XPIC_CALL(halo_fill(c, const_cast<double*>(r), 1));
      else XPIC_CALL(halo_fill_f32(c, z0, 1));
      if (first && last) hipLaunchKernelGGL((k_cheb32<true, true>), grid, block, 0, c->stream, g, r, r32, z0, d, z1, out, cd, cr, it);
      else if (first) hipLaunchKernelGGL((k_cheb32<true, false>), grid, block, 0, c->stream, g, r, r32, z0, d, z1, out, cd, cr, it);
      else if (last) hipLaunchKernelGGL((k_cheb32<false, true>), grid, block, 0, c->stream, g, r, r32, z0, d, z1, out, cd, cr, it);
      else hipLaunchKernelGGL((k_cheb32<false, false>), grid, block, 0, c->stream, g, r, r32, z0, d, z1, out, cd, cr, it);
      XPIC_HIP(hipGetLastError());
      rho = rho_new;
      std::swap(z0, z1);
    }
    return 0;
  }
  double* d = c->kry_p[0];
  double* z0 = c->kry_p[1];
  double* z1 = c->kry_p[2];
  for (int i = 1; i < degree; ++i) {
    const double rho_new = 1.0 / (2.0 * sigma1 - rho);
    double* zout = i == degree - 1 ? out : z1; // the last step lands in the caller's vector
    if (i == 1) {
      XPIC_CALL(halo_fill(c, const_cast<double*>(r), 1));
      hipLaunchKernelGGL(k_cheb<true>, grid, block, 0, c->stream, g, r, r, d, zout,
        rho_new * rho, 2.0 * rho_new / delta, 1.0 / theta);
    }
    else {
      XPIC_CALL(halo_fill(c, z0, 1));
      hipLaunchKernelGGL(k_cheb<false>, grid, block, 0, c->stream, g, r, z0, d, zout,
        rho_new * rho, 2.0 * rho_new / delta, 1.0 / theta);
    }
    XPIC_HIP(hipGetLastError());
    rho = rho_new;
    std::swap(z0, z1);
  }
  return 0;
}

int div_neg_add(xpic_ctx* c, double* v3, double* out_scalar)
{
  XPIC_CALL(halo_fill(c, v3, 1));
  long blocks = (c->g.nown + kBlock - 1) / kBlock;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(k_div_neg_add, dim3((unsigned)blocks), dim3(kBlock), 0, c->stream, c->g, v3, out_scalar);
  XPIC_HIP(hipGetLastError());
  return 0;
}

int scalar_norm12_host(xpic_ctx* c, const double* f, double* out2) // VecNorm(NORM_1_AND_2) of component 0
{
  long blocks = (c->g.nown + kBlock * 4 - 1) / (kBlock * 4);
  if (blocks > kRedBlocks) blocks = kRedBlocks;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_norm12, dim3((unsigned)blocks), dim3(kBlock), 0, c->stream, c->g, f, c->red_partial);
  XPIC_HIP(hipGetLastError());
  XPIC_CALL(finish_reduce(c, 2, (int)blocks, 1, out2));
  out2[1] = std::sqrt(out2[1]);
  return 0;
}

int field_import(xpic_ctx* c, double* dst, const double* src_host)
{
  const long n = c->g.nown * 3;
  double* tmp = nullptr;
  XPIC_HIP(hipMalloc(&tmp, sizeof(double) * n));
  XPIC_HIP(hipMemcpyAsync(tmp, src_host, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
  long blocks = (n + kBlock - 1) / kBlock;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(k_import, dim3((unsigned)blocks), dim3(kBlock), 0, c->stream, c->g, tmp, dst);
  XPIC_HIP(hipGetLastError());
  XPIC_HIP(hipStreamSynchronize(c->stream));
  XPIC_HIP(hipFree(tmp));
  return 0; // ghost planes are refreshed by whoever reads them next (halo_fill is collective over the slabs)
}

int field_export(xpic_ctx* c, const double* src, double* dst_host)
{
  const long n = c->g.nown * 3;
  double* tmp = nullptr;
  XPIC_HIP(hipMalloc(&tmp, sizeof(double) * n));
  long blocks = (n + kBlock - 1) / kBlock;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(k_export, dim3((unsigned)blocks), dim3(kBlock), 0, c->stream, c->g, src, tmp);
  XPIC_HIP(hipGetLastError());
  XPIC_HIP(hipMemcpyAsync(dst_host, tmp, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
  XPIC_HIP(hipStreamSynchronize(c->stream));
  XPIC_HIP(hipFree(tmp));
  return 0;
}

// ---- z-halo exchange (DMGlobalToLocal / DMLocalToGlobal(ADD) of the slab decomposition) --------------
namespace {

// buf[c][w][plane] <-> f[c][zs0 + w][plane]
// blockIdx.y = 0 / 1: the lower / upper side of an exchange in ONE launch (a 32-plane slab runs 32 exchanges per step:
// four launches of a few microseconds each around every one of them were two too many)
template <int OP, class T = double> // 0: pack f -> buf, 1: unpack buf -> f, 2: add buf into f
__global__ void __launch_bounds__(kBlock) k_planes(GridDev g, T* f, T* buf0, int zs00, T* buf1, int zs01, int width)
{
  T* buf = blockIdx.y == 0 ? buf0 : buf1;
  const int zs0 = blockIdx.y == 0 ? zs00 : zs01;
  const long per = (long)width * g.plane;
  const long n = 3 * per;
  const long stride = (long)gridDim.x * kBlock;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const int c = (int)(i / per);
    const long o = c * g.cstride + (long)zs0 * g.plane + (i % per);
    if (OP == 0) buf[i] = f[o];
    else if (OP == 1) f[o] = buf[i];
    else f[o] += buf[i];
  }
}

__global__ void __launch_bounds__(kBlock) k_add_into(double* dst, const double* src, long n)
{
  const long stride = (long)gridDim.x * kBlock;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) dst[i] += src[i];
}

inline unsigned plane_grid(long n)
{
  long b = (n + kBlock - 1) / kBlock;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

int ensure_halo_buf(xpic_ctx* c, size_t bytes)
{
  if (c->halo_bytes >= bytes) return 0;
  for (int i = 0; i < 4; ++i) {
    if (c->halo_buf[i]) XPIC_HIP(hipFree(c->halo_buf[i]));
    XPIC_HIP(hipMalloc(&c->halo_buf[i], bytes));
  }
  c->halo_bytes = bytes;
  return 0;
}

namespace {
template <class T>
int halo_fill_t(xpic_ctx* c, T* f, int width)
{
  const GridDev& g = c->g;
  if (g.G == 0) return 0;
  XPIC_CHECK(width <= g.G && width <= g.nzl, "halo width exceeds the ghost layer");
  Timed t(c, "halo");
  const long n = 3L * width * g.plane;
  const size_t bytes = sizeof(T) * n;
  XPIC_CALL(ensure_halo_buf(c, bytes));
  const unsigned nb = plane_grid(n);
  T* hb[4] = {(T*)c->halo_buf[0], (T*)c->halo_buf[1], (T*)c->halo_buf[2], (T*)c->halo_buf[3]};
  // my bottom owned planes go down, my top owned planes go up
  hipLaunchKernelGGL((k_planes<0, T>), dim3(nb, 2), dim3(kBlock), 0, c->stream, g, f, hb[0], g.G, hb[1], g.G + g.nzl - width, width);
  XPIC_HIP(hipGetLastError());
  XPIC_CALL(comm_ring(c, c->halo_buf[0], bytes, c->halo_buf[1], bytes, c->halo_buf[2], bytes, c->halo_buf[3], bytes));
  // from the upper neighbour: its bottom planes = my upper ghost; from the lower: its top planes = my lower ghost
  hipLaunchKernelGGL((k_planes<1, T>), dim3(nb, 2), dim3(kBlock), 0, c->stream, g, f, hb[2], g.G + g.nzl, hb[3], g.G - width, width);
  XPIC_HIP(hipGetLastError());
  return 0;
}
}  // namespace

int halo_fill(xpic_ctx* c, double* f, int width) { return halo_fill_t<double>(c, f, width); }

// DMGlobalToLocal of two vectors of one phase (E and B in front of a push) as ONE message per neighbour: a slab step is
// bound by the number of its small exchanges, not by their bytes
int halo_fill2(xpic_ctx* c, double* f0, double* f1, int width)
{
  const GridDev& g = c->g;
  if (g.G == 0) return 0;
  XPIC_CHECK(width <= g.G && width <= g.nzl, "halo width exceeds the ghost layer");
  Timed t(c, "halo");
  const long n = 3L * width * g.plane;
  const size_t bytes = sizeof(double) * n;
  XPIC_CALL(ensure_halo_buf(c, 2 * bytes));
  const unsigned nb = plane_grid(n);
  double* hb[4] = {c->halo_buf[0], c->halo_buf[1], c->halo_buf[2], c->halo_buf[3]};
  double* fs[2] = {f0, f1};
  for (int k = 0; k < 2; ++k)
    hipLaunchKernelGGL((k_planes<0, double>), dim3(nb, 2), dim3(kBlock), 0, c->stream, g, fs[k], hb[0] + k * n, g.G, hb[1] + k * n, g.G + g.nzl - width, width);
  XPIC_HIP(hipGetLastError());
  XPIC_CALL(comm_ring(c, hb[0], 2 * bytes, hb[1], 2 * bytes, hb[2], 2 * bytes, hb[3], 2 * bytes));
  for (int k = 0; k < 2; ++k)
    hipLaunchKernelGGL((k_planes<1, double>), dim3(nb, 2), dim3(kBlock), 0, c->stream, g, fs[k], hb[2] + k * n, g.G + g.nzl, hb[3] + k * n, g.G - width, width);
  XPIC_HIP(hipGetLastError());
  return 0;
}

// Split form of halo_fill for the overlapped operator apply.  With RCCL the exchange and the unpack run on the
// context's communication stream between two events; with the host-callback transport (tests) the exchange is
// synchronous and halo_post simply completes it.
int halo_post(xpic_ctx* c, double* f, int width)
{
  const GridDev& g = c->g;
  if (g.G == 0) return 0;
  if (c->comm.kind != 1 || !c->comm_stream) return halo_fill(c, f, width);
  XPIC_CHECK(width <= g.G && width <= g.nzl, "halo width exceeds the ghost layer");
  const long n = 3L * width * g.plane;
  const size_t bytes = sizeof(double) * n;
  XPIC_CALL(ensure_halo_buf(c, bytes));
  const unsigned nb = plane_grid(n);
  double* hb[4] = {c->halo_buf[0], c->halo_buf[1], c->halo_buf[2], c->halo_buf[3]};
  hipLaunchKernelGGL((k_planes<0, double>), dim3(nb, 2), dim3(kBlock), 0, c->stream, g, f, hb[0], g.G, hb[1], g.G + g.nzl - width, width);
  XPIC_HIP(hipGetLastError());
  XPIC_HIP(hipEventRecord(c->comm_ev[0], c->stream));
  XPIC_HIP(hipStreamWaitEvent(c->comm_stream, c->comm_ev[0], 0));
  hipStream_t compute = c->stream;
  c->stream = c->comm_stream; // comm_ring and the unpack below are enqueued on the communication stream
  int rc = comm_ring(c, hb[0], bytes, hb[1], bytes, hb[2], bytes, hb[3], bytes);
  if (rc == 0) {
    hipLaunchKernelGGL((k_planes<1, double>), dim3(nb, 2), dim3(kBlock), 0, c->stream, g, f, hb[2], g.G + g.nzl, hb[3], g.G - width, width);
  }
  c->stream = compute;
  XPIC_CALL(rc);
  XPIC_HIP(hipGetLastError());
  XPIC_HIP(hipEventRecord(c->comm_ev[1], c->comm_stream));
  c->halo_posted = true;
  return 0;
}

int halo_wait(xpic_ctx* c)
{
  if (!c->halo_posted) return 0;
  c->halo_posted = false;
  XPIC_HIP(hipStreamWaitEvent(c->stream, c->comm_ev[1], 0));
  return 0;
}
int halo_fill_f32(xpic_ctx* c, float* f, int width) { return halo_fill_t<float>(c, f, width); }

int halo_add(xpic_ctx* c, double* f, int width)
{
  const GridDev& g = c->g;
  if (g.G == 0) return 0;
  XPIC_CHECK(width <= g.G && width <= g.nzl, "halo width exceeds the ghost layer");
  // both sides are added by ONE launch (plain +=): the bottom and the top plane ranges must not share a plane
  XPIC_CHECK(2 * width <= g.nzl, "halo_add: the slab is thinner than its two halo layers");
  Timed t(c, "halo");
  const long n = 3L * width * g.plane;
  const size_t bytes = sizeof(double) * n;
  XPIC_CALL(ensure_halo_buf(c, bytes));
  const unsigned nb = plane_grid(n);
  // what I deposited below my slab belongs to the lower neighbour's top planes, above -> upper neighbour's bottom
  hipLaunchKernelGGL(k_planes<0>, dim3(nb, 2), dim3(kBlock), 0, c->stream, g, f, c->halo_buf[0], g.G - width, c->halo_buf[1], g.G + g.nzl, width);
  XPIC_HIP(hipGetLastError());
  XPIC_CALL(comm_ring(c, c->halo_buf[0], bytes, c->halo_buf[1], bytes, c->halo_buf[2], bytes, c->halo_buf[3], bytes));
  hipLaunchKernelGGL(k_planes<2>, dim3(nb, 2), dim3(kBlock), 0, c->stream, g, f, c->halo_buf[2], g.G + g.nzl - width, c->halo_buf[3], g.G, width);
  XPIC_HIP(hipGetLastError());
  return 0;
}

// matL rows of the first ghost plane below / above the slab were filled by cells of this rank but belong to the
// neighbours (MatSetValuesCOO ships such entries to the owner, src/impls/ecsim/simulation.cpp:366): send and add.
// 3 x lplane() doubles per neighbour (195 MB on a 256 x 256 slab): the only large message of a step.  It is split in
// two: _post ships the ghost planes as soon as the assembly's boundary colours have finished them (ecsim.hip launches
// those first) -- with RCCL and xpic_set_overlap on a second stream, beside the interior colours -- and _finish adds
// what arrived into the first / last owned row plane once every local launch is done.
int matL_ghost_rows_post(xpic_ctx* c)
{
  const GridDev& g = c->g;
  if (g.G == 0) return 0;
  XPIC_CHECK(c->lrow_buf[0] && c->lrow_buf[1], "matL ghost-row buffers missing");
  XPIC_CHECK(!c->lrow_posted, "matL ghost rows posted twice");
  Timed t(c, "matL_ghost_rows");
  const long per = g.lplane(); // one row plane of one component
  const size_t bytes = sizeof(double) * per;
  const int nzp = g.nzl + 2;
  // Beside the interior colours only on request (xpic_set_overlap bit 1).  Measured on a self-ring, 256 x 256 x 32 slab
  // (profiles/r04_step_slab_selfring.txt, r04_trace_overlap_selfring.txt): with the exchange on the high-priority second
  // stream its RCCL kernels do run beside k_ecsim_fill (0.85 of their 0.86 ms each) and the step takes 3.6 ms LONGER --
  // a colour launch is sized to fill every workgroup slot of the chip exactly once, so each slot an RCCL workgroup holds
  // sends one assembly workgroup into a second round, which doubles the launch.  The blocking exchange costs its own
  // duration and nothing else.
  // Copy-engine form (xpic_set_overlap bit 2, after xpic_comm_peer_import): the rows are written straight into the
  // neighbours' receive buffers by hipMemcpyAsync on the copy stream -- between two GPUs that is an SDMA engine, which
  // takes no workgroup slot from the colour launches still to come (the reason the RCCL form beside the assembly lost).
  // No message of its own tells the neighbour that the rows are there: this rank's NEXT ring exchange is issued behind the
  // copies (comm.hip: peer_order), the neighbour's matching receive completes after it, and matL_ghost_rows_finish makes
  // sure one such exchange lies between post and finish.  The receiver's buffer is free again by then: its finish of the
  // step before was enqueued ahead of every message it has sent since.
  if (c->peer_copy && c->peer_lrow[0] && c->peer_lrow[1] && c->copy_stream) {
    XPIC_HIP(hipEventRecord(c->copy_ev[0], c->stream));
    XPIC_HIP(hipStreamWaitEvent(c->copy_stream, c->copy_ev[0], 0));
    for (int c1 = 0; c1 < 3; ++c1) {
      double* base = c->matL + (long)c1 * nzp * per;
      if (c1 == 2) XPIC_HIP(hipMemcpyAsync(c->peer_lrow[0] + (long)c1 * per, base, bytes, hipMemcpyDefault, c->copy_stream));
      XPIC_HIP(hipMemcpyAsync(c->peer_lrow[1] + (long)c1 * per, base + (long)(nzp - 1) * per, bytes, hipMemcpyDefault, c->copy_stream));
    }
    XPIC_HIP(hipEventRecord(c->copy_ev[1], c->copy_stream));
    c->comm.sent_msgs += 4; c->comm.sent_bytes += (int64_t)(4 * bytes);
    if (c->profiling) c->prof["peer_copies"].launches += 4;
    c->peer_pending = true;
    c->peer_exchanges = 0;
    c->lrow_posted = true; c->lrow_by_copy = true; c->lrow_on_comm_stream = false;
    return 0;
  }
  c->lrow_by_copy = false;
  const bool side = c->overlap_lrows && c->comm.kind == 1 && c->comm_stream;
  hipStream_t compute = c->stream;
  if (side) {
    XPIC_HIP(hipEventRecord(c->comm_ev[0], c->stream));
    XPIC_HIP(hipStreamWaitEvent(c->comm_stream, c->comm_ev[0], 0));
    c->stream = c->comm_stream; // comm_ring enqueues on the context's stream
  }
  int rc = 0;
  for (int c1 = 0; c1 < 3 && rc == 0; ++c1) {
    double* base = c->matL + (long)c1 * nzp * per;
    // the ghost plane BELOW the slab holds rows of the z component only: a cell's X and Y rows sit on its own two node
    // planes (lstencil.h: block_node_offset), never one below -- those two planes are all zeros and stay at home
    const size_t down = c1 == 2 ? bytes : 0;
    rc = comm_ring(c, base, down, base + (long)(nzp - 1) * per, bytes, c->lrow_buf[0] + (long)c1 * per, down,
      c->lrow_buf[1] + (long)c1 * per, bytes);
  }
  c->stream = compute;
  XPIC_CALL(rc);
  if (side) XPIC_HIP(hipEventRecord(c->comm_ev[1], c->comm_stream));
  c->lrow_posted = true;
  c->lrow_on_comm_stream = side;
  return 0;
}

int matL_ghost_rows_finish(xpic_ctx* c)
{
  const GridDev& g = c->g;
  if (g.G == 0) return 0;
  if (!c->lrow_posted) XPIC_CALL(matL_ghost_rows_post(c));
  Timed t(c, "matL_ghost_rows");
  if (c->lrow_on_comm_stream) XPIC_HIP(hipStreamWaitEvent(c->stream, c->comm_ev[1], 0));
  if (c->lrow_by_copy && c->peer_exchanges == 0) {
    // the neighbours' rows came by copy and nothing has been exchanged since they were posted (no current to add back):
    // a one-word exchange is the arrival signal
    double* w = c->red_out + 124;
    XPIC_CALL(comm_ring(c, w, sizeof(double), w + 1, sizeof(double), w + 2, sizeof(double), w + 3, sizeof(double)));
  }
  c->lrow_by_copy = false;
  c->lrow_posted = false;
  const long per = g.lplane();
  const int nzp = g.nzl + 2;
  for (int c1 = 0; c1 < 3; ++c1) {
    double* base = c->matL + (long)c1 * nzp * per;
    // upper neighbour's ghost-below rows (z component only) are my top owned plane; lower neighbour's ghost-above rows my bottom plane
    if (c1 == 2) hipLaunchKernelGGL(k_add_into, dim3(plane_grid(per)), dim3(kBlock), 0, c->stream, base + (long)g.nzl * per, c->lrow_buf[0] + (long)c1 * per, per);
    hipLaunchKernelGGL(k_add_into, dim3(plane_grid(per)), dim3(kBlock), 0, c->stream, base + per, c->lrow_buf[1] + (long)c1 * per, per);
  }
  XPIC_HIP(hipGetLastError());
  return 0;
}

int matL_exchange_ghost_rows(xpic_ctx* c)
{
  XPIC_CALL(matL_ghost_rows_post(c));
  return matL_ghost_rows_finish(c);
}

}  // namespace xpic
