// precond.hip -- the preconditioner of the "predict" solve that sees the mass matrix (kind 3).
//
// matA = matM + matL.  A polynomial in matM alone (kinds 1, 2: fields.hip) leaves spec(matA P) = [1, 1.25] at the
// bench's parameters, i.e. 6 GMRES iterations at rtol 1e-7 whatever its degree: on the null space of the curl matM
// is 2 I at every wavelength while matL runs from ~0.5 (smooth modes) to ~0.02 (oscillatory ones).  What tells
// those modes apart is the smoothing stencil of the mass matrix, so the surrogate here is the TRANSLATION AVERAGE
// of the assembled matL,
//     Lbar[c1][k] = < matL[node][c1][k] >_nodes        (123 constants per row component),
// which for a uniform plasma is the 27 / 48-point CIC overlap stencil times the mean A_p matB_p rotation, and
//     Abar = matM + Lbar
// is ONE constant-coefficient 123-point stencil (matM's 13 points are a subset of the pattern).  P = p_k(Abar), a
// fixed Chebyshev polynomial, is applied matrix-free: no coefficient stream, no inner products, one 2-plane halo per
// step on slabs.  What is left for GMRES is matA - Abar = the particle noise of matL: spec(matA Abar^-1) = [0.98, 1.02]
// at 64 ppc (tools/precond_spectrum.py, dense eigenvalues on a 10^3 box), 4 iterations instead of 6 -- each of
// them costs a 50 GB sweep of matL.  The GMRES around it is flexible (krylov.hip): the polynomial runs on fp32
// copies of its vectors with fp32 arithmetic, the stopping rule stays the true fp64 residual.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#include "common.h"
#include "lstencil.h"

namespace xpic {

int experiment_precond() { return XPIC_TU_EXPERIMENT; }

namespace {

constexpr int kB = 256;
constexpr double kAutoSpread = 0.2; // kind 5: relative spread of matL's diagonal above which the surrogate's rows are scaled
#ifndef XPIC_CHEB_MIN_ZC
#define XPIC_CHEB_MIN_ZC 8 // planes per z-chunk of k_cheb_bar at least (a chunk loads 5 more)
#endif
#ifndef BAR_SCHED_SCALED
#define BAR_SCHED_SCALED 1 // kind 4's body (12 more live values) gets one scheduling region per (c2, dz) group: without it 2.2 KB of spills per lane
#endif
#ifndef BAR_SCHED_GROUP
#define BAR_SCHED_GROUP 0 // scheduling regions of k_cheb_bar: 1 = one per (c2, dz) group of lines, 5 = one per c2, 0 = none (measured equal)
#endif

// ---- matM as a 123-pattern stencil (host): probe 2 I + 0.5 dt^2 rot- rot+ with unit impulses on a 5^3 periodic box,
// same difference formulas as fields.hip: rot_at / matM_at (src/utils/operators.cpp:155-215, ecsim/simulation.cpp:544-551)
void matM_stencil(const GridDev& g, double* co /* [3][kLPad] */)
{
  constexpr int n = 5;
  auto id = [](int x, int y, int z) { return ((z + n) % n * n + (y + n) % n) * n + (x + n) % n; };
  const double ih[3] = {g.inv[0], g.inv[1], g.inv[2]};
  for (int i = 0; i < 3 * kLPad; ++i) co[i] = 0.0;
  for (int c2 = 0; c2 < 3; ++c2) {
    std::vector<double> F(3 * n * n * n, 0.0), G(3 * n * n * n, 0.0), H(3 * n * n * n, 0.0);
    F[c2 * n * n * n + id(2, 2, 2)] = 1.0;
    auto at = [&](const std::vector<double>& v, int c, int x, int y, int z) { return v[c * n * n * n + id(x, y, z)]; };
    for (int z = 0; z < n; ++z)
      for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x) { // G = rot+ F (forward differences)
          G[0 * n * n * n + id(x, y, z)] = ih[1] * (at(F, 2, x, y + 1, z) - at(F, 2, x, y, z)) - ih[2] * (at(F, 1, x, y, z + 1) - at(F, 1, x, y, z));
          G[1 * n * n * n + id(x, y, z)] = -ih[0] * (at(F, 2, x + 1, y, z) - at(F, 2, x, y, z)) + ih[2] * (at(F, 0, x, y, z + 1) - at(F, 0, x, y, z));
          G[2 * n * n * n + id(x, y, z)] = ih[0] * (at(F, 1, x + 1, y, z) - at(F, 1, x, y, z)) - ih[1] * (at(F, 0, x, y + 1, z) - at(F, 0, x, y, z));
        }
    for (int z = 0; z < n; ++z)
      for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x) { // H = rot- G (backward differences)
          H[0 * n * n * n + id(x, y, z)] = ih[1] * (at(G, 2, x, y, z) - at(G, 2, x, y - 1, z)) - ih[2] * (at(G, 1, x, y, z) - at(G, 1, x, y, z - 1));
          H[1 * n * n * n + id(x, y, z)] = -ih[0] * (at(G, 2, x, y, z) - at(G, 2, x - 1, y, z)) + ih[2] * (at(G, 0, x, y, z) - at(G, 0, x, y, z - 1));
          H[2 * n * n * n + id(x, y, z)] = ih[0] * (at(G, 1, x, y, z) - at(G, 1, x - 1, y, z)) - ih[1] * (at(G, 0, x, y, z) - at(G, 0, x, y - 1, z));
        }
    // row (c1, node q) couples to column (c2, node p = (2,2,2)) with offset d = p - q
    for (int c1 = 0; c1 < 3; ++c1)
      for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy)
          for (int dx = -1; dx <= 1; ++dx) {
            const int k = lencode(c1, c2, dx, dy, dz);
            if (k < 0) continue;
            double v = 0.5 * g.dt * g.dt * at(H, c1, 2 - dx, 2 - dy, 2 - dz);
            if (c1 == c2 && dx == 0 && dy == 0 && dz == 0) v += 2.0;
            co[c1 * kLPad + k] = v;
          }
  }
}

// ---- translation average of matL over a sample of its rows ------------------------------------------------------
// stage 1: one workgroup per sampled row block (c1, zp, y): sums the row's x-blocks; thread t = k * 4 + x % 4
__global__ void __launch_bounds__(512) k_lbar_rows(GridDev g, const double* __restrict__ matL, int sy, int sz, int nys,
  double* __restrict__ partial)
{
  const int t = threadIdx.x;
  const int row = blockIdx.x, c1 = blockIdx.y;
  const int y = (row % nys) * sy, z = (row / nys) * sz;
  if (t >= kLBlock) return;
  const double* base = matL + g.lindex(c1, z + (g.G ? 1 : 0), y, 0, 0) + t;
  double s = 0.0;
  const int nb = g.nbx();
  for (int b = 0; b < nb; ++b) s += base[(long)b * kLBlock];
  partial[((long)row * 3 + c1) * kLBlock + t] = s;
}

// stage 2: fixed-order sums over segments of the sampled rows; stage 3: over the segments, then over x % 4 (deterministic)
constexpr int kSegs = 64;
__global__ void __launch_bounds__(512) k_lbar_segments(const double* __restrict__ partial, int nrows, double* __restrict__ segsum)
{
  const int t = threadIdx.x, c1 = blockIdx.x, sg = blockIdx.y;
  if (t >= kLBlock) return;
  const int per = (nrows + kSegs - 1) / kSegs, r0 = sg * per, r1 = min(r0 + per, nrows);
  double s = 0.0;
  for (int r = r0; r < r1; ++r) s += partial[((long)r * 3 + c1) * kLBlock + t];
  segsum[((long)sg * 3 + c1) * kLBlock + t] = s;
}

__global__ void __launch_bounds__(512) k_lbar_final(const double* __restrict__ segsum, double* __restrict__ sums)
{
  __shared__ double sm[kLBlock];
  const int t = threadIdx.x, c1 = blockIdx.x;
  if (t < kLBlock) {
    double s = 0.0;
    for (int sg = 0; sg < kSegs; ++sg) s += segsum[((long)sg * 3 + c1) * kLBlock + t];
    sm[t] = s;
  }
  __syncthreads();
  if (t < kLPad) sums[c1 * kLPad + t] = (sm[4 * t] + sm[4 * t + 1]) + (sm[4 * t + 2] + sm[4 * t + 3]);
}

// Second moment of matL's diagonal over the same sampled rows (kind 5's choice between the plain and the density-scaled
// surrogate): one workgroup per sampled row and component sums d^2 along x in a fixed order; k_diag2_final adds the rows'
// values, again in a fixed order, into the free slot 123 of `sums` (kLPad = 124), so that it is averaged, all-reduced on
// slabs and copied into abar64 with the coefficients.
__global__ void __launch_bounds__(256) k_diag2_rows(GridDev g, const double* __restrict__ matL, int sy, int sz, int nys,
  double* __restrict__ part2)
{
  const int row = blockIdx.x, c1 = blockIdx.y;
  const int y = (row % nys) * sy, z = (row / nys) * sz;
  const int kd = lencode(c1, c1, 0, 0, 0);
  double s = 0.0;
  for (int x = threadIdx.x; x < g.nx; x += 256) {
    const double d = matL[g.lindex(c1, z + (g.G ? 1 : 0), y, x, kd)];
    s += d * d;
  }
  __shared__ double sm[256];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part2[(long)row * 3 + c1] = sm[0];
}
__global__ void __launch_bounds__(256) k_diag2_final(const double* __restrict__ part2, int nrows, double* __restrict__ sums)
{
  const int c1 = blockIdx.x;
  double s = 0.0;
  for (int r = threadIdx.x; r < nrows; r += 256) s += part2[(long)r * 3 + c1];
  __shared__ double sm[256];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) sums[c1 * kLPad + kLStencil] = sm[0];
}

// Abar = matM + <matL>: fp64 copy in the 123-pattern (host: spectral bound) and the fp32 table k_cheb_bar reads, packed
// per stencil line (line_slot / tap_pos; defined below)
__host__ __device__ constexpr bool line_used(int c2, int dy, int dz);
__host__ __device__ constexpr int line_slot(int L);
__host__ __device__ constexpr int tap_pos(int c2, int dz, int dy, int I);
__global__ void k_abar(const double* __restrict__ sums, const double* __restrict__ mco, double inv_count, float* packed,
  double* abar64, double lscale);

// ---- one Chebyshev step on Abar:  res = r - Abar z ; d = cd d + cr res ; z_out = z + d  (fp32 vectors) -------------
// A workgroup owns a (128 x, 4 y) column of nodes and marches along z with a sliding window of 5 planes of the three
// components in LDS (radius 2 in every direction); a lane computes two neighbouring x nodes of one row: every line
// (c2, dy, dz) of the stencil is three 8-byte LDS reads for 2 x 5 taps.  The coefficients are wave-uniform (scalar
// loads).  ~370 fp32 FMAs per node against ~70 LDS reads: bound by FMA issue, and by the 5 V / 2 of vector traffic.
constexpr int kTX = 128, kTY = 4, kH = 2, kPX = kTX + 2 * kH, kPY = kTY + 2 * kH, kZW = 6;
constexpr int kPlaneF = 3 * kPY * kPX; // floats of one window plane

// Which taps of the 123-pattern the polynomial's stencil keeps (round 5).  The translation average of matL is, up to its
// sampling noise, a(c1, c2) times a PRODUCT of per-axis CIC overlaps (checked against the CPU restatement's matL: 3e-3 of
// the largest entry on a 10^3 box, the noise of 1 000 rows): between equally staggered components (1/6, 2/3, 1/6), between
// a component staggered along the axis and one that is not (1/48, 23/48, 23/48, 1/48).  In a block c2 != c1 every tap that
// carries a 1/48 -- d[c1] outside {0, 1} or d[c2] outside {-1, 0} -- is 2 % of its neighbours; together they hold
// 1 - (46/48)^2 = 8 % of a block that is itself the rotation part of the particle matrix (a tenth of the diagonal block at
// the reference's field): 36 of the 48 taps for 0.4 % of the row.  They are dropped and their sum is spread over the 12
// that stay (k_abar), so the block's sum -- what a smooth field sees -- is kept: 51 taps per row instead of 123 in a
// kernel that is short of vector issue slots (profiles/r05_pmc_sq_solve.txt).  The diagonal blocks and matM's taps (a
// subset of the kept pattern) are untouched.
#ifndef BAR_TRUNCATE
#define BAR_TRUNCATE 1
#endif
// (Measured at 256^3 x 64, the solve's residuals after 1 .. 6 iterations are the same to three digits with and without
// the dropped taps -- 3.92e-8 against 3.91e-8 |b| at the fourth -- and the polynomial's 28 steps take 7.5 instead of 10.3 ms.
// Dropping more -- the eight corners of the diagonal blocks, the third axis' outer taps of the others: 27 taps per row --
// left the fourth residual at 4.04e-8 / 4.24e-8 and the time at 7.5 / 7.1 ms: the kernel is at its vector traffic by
// then, and those taps are not negligible for a strong field or a dense plasma.  Not taken.)
__host__ __device__ constexpr bool tap_kept(int c1, int c2, int dx, int dy, int dz)
{
  if (!BAR_TRUNCATE || c1 == c2) return true;
  const int d[3] = {dx, dy, dz};
  return (d[c1] == 0 || d[c1] == 1) && (d[c2] == 0 || d[c2] == -1);
}
// the tap exists in the polynomial's stencil
__host__ __device__ constexpr bool tap_in(int c1, int c2, int dx, int dy, int dz)
{
  return lencode(c1, c2, dx, dy, dz) >= 0 && tap_kept(c1, c2, dx, dy, dz);
}

__host__ __device__ constexpr bool line_used(int c2, int dy, int dz)
{
  for (int c1 = 0; c1 < 3; ++c1)
    for (int dx = -2; dx <= 2; ++dx)
      if (tap_in(c1, c2, dx, dy, dz)) return true;
  return false;
}

typedef float fpair __attribute__((ext_vector_type(2)));
typedef float fquad __attribute__((ext_vector_type(4)));
typedef double dpairv __attribute__((ext_vector_type(2)));

// The coefficients travel through LDS, packed per stencil line: line L = (c2 * 5 + dz + 2) * 5 + dy + 2 owns kCoefPitch
// floats at slot line_slot(L); its taps I = (dx + 2) * 3 + c1 that exist sit at tap_pos(.., I) in that order.  Read as
// wave-uniform 16-byte LDS reads they return in order with the window reads (counted waits); as 369 scalar loads per
// plane they shared the LDS counter, had to be waited for with lgkmcnt(0), and the kernel spent 4/5 of its time there.
constexpr int kCoefPitch = 12; // a line has at most 3 + 4 + 4 taps
__host__ __device__ constexpr int line_slot(int L)
{
  int n = 0;
  for (int l = 0; l < L; ++l) n += line_used(l / 25, l % 5 - 2, (l / 5) % 5 - 2) ? 1 : 0;
  return n;
}
constexpr int kLinesUsed = line_slot(75);
__host__ __device__ constexpr int tap_pos(int c2, int dz, int dy, int I)
{
  int n = 0;
  for (int i = 0; i < I; ++i) n += tap_in(i % 3, c2, i / 3 - 2, dy, dz) ? 1 : 0;
  return n;
}

// ---- kind 4: the surrogate scaled by the LOCAL density.  Abar_r = matM + diag(r) Lbar with r[node][c1] = matL's own
// diagonal entry of the row over Lbar's: the rows of the average follow the particle count of their neighbourhood (about
// half of the variance of matL around Lbar is count noise), spec(matA Abar_r^-1) = [0.992, 1.007] against [0.983, 1.017]
// on the 10^3 box of tools/precond_spectrum.py: 3 GMRES iterations instead of 4 at rtol 1e-7.  Applied by the same
// stencil kernel: matM's 13 taps per row are summed a second time on their own (their window values are already in
// registers), and (Abar_r z) = matM z + r ((Abar z) - matM z).
// matM = 2 I + 0.5 dt^2 rot- rot+: row component c1 couples to itself at the node and its four neighbours across c1's own
// axis, and to c2 != c1 at d[c1] in {0, 1}, d[c2] in {-1, 0} (verified against the probed coefficients in abar_alloc)
__host__ __device__ constexpr bool is_matM_tap(int c1, int c2, int dx, int dy, int dz)
{
  const int d[3] = {dx, dy, dz};
  if (c1 == c2) {
    if (d[c1] != 0) return false;
    int s1 = 0;
    for (int a = 0; a < 3; ++a) s1 += d[a] < 0 ? -d[a] : d[a];
    return s1 <= 1;
  }
  const int c3 = 3 - c1 - c2;
  return (d[c1] == 0 || d[c1] == 1) && (d[c2] == 0 || d[c2] == -1) && d[c3] == 0;
}
__host__ __device__ constexpr bool mline_used(int c2, int dy, int dz)
{
  for (int c1 = 0; c1 < 3; ++c1)
    for (int dx = -2; dx <= 2; ++dx)
      if (lencode(c1, c2, dx, dy, dz) >= 0 && is_matM_tap(c1, c2, dx, dy, dz)) return true;
  return false;
}
constexpr int kMPitch = 8; // matM taps of one stencil line: at most 5
__host__ __device__ constexpr int mline_slot(int L)
{
  int n = 0;
  for (int l = 0; l < L; ++l) n += mline_used(l / 25, l % 5 - 2, (l / 5) % 5 - 2) ? 1 : 0;
  return n;
}
constexpr int kMLinesUsed = mline_slot(75);
__host__ __device__ constexpr int mtap_pos(int c2, int dz, int dy, int I)
{
  int n = 0;
  for (int i = 0; i < I; ++i) n += (lencode(i % 3, c2, i / 3 - 2, dy, dz) >= 0 && is_matM_tap(i % 3, c2, i / 3 - 2, dy, dz)) ? 1 : 0;
  return n;
}
static_assert(mtap_pos(0, 0, 0, 15) <= kMPitch && mtap_pos(1, 0, 0, 15) <= kMPitch && mtap_pos(2, 0, 0, 15) <= kMPitch, "matM line table pitch");

// The stencil is expanded at compile time (integer sequences, as k_matA's term list): tap I = (dx + 2) * 3 + c1 of the
// line (C2, DZ, DY) exists iff lencode() >= 0 -- a constant expression here, not a run-time test.
__global__ void k_abar(const double* __restrict__ sums, const double* __restrict__ mco, double inv_count, float* packed,
  double* abar64, double lscale)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 3 * kLPad) return;
  const int c1 = i / kLPad, k = i % kLPad;
  const double v = mco[i] + sums[i] * inv_count * (k < kLStencil ? lscale : 1.0); // (lscale = 1 but in the probation test)
  abar64[i] = v; // (the full pattern: the host's spectral bounds are taken on it -- the kept taps' absolute sums are no larger)
  if (k >= kLStencil) return;
  const LEntry e = ldecode(c1, k);
  if (!tap_kept(c1, e.c2, e.d[0], e.d[1], e.d[2])) return;
  // the kept taps of a block share the sum of its dropped ones in proportion (tap_kept; matM has no dropped tap)
  double keep = 1.0;
  if (BAR_TRUNCATE && e.c2 != c1) {
    double all = 0.0, kept = 0.0;
    const int k0 = lblock_offset(c1, e.c2);
    for (int kk = k0; kk < k0 + 48; ++kk) {
      const LEntry f = ldecode(c1, kk);
      all += sums[c1 * kLPad + kk];
      if (tap_kept(c1, f.c2, f.d[0], f.d[1], f.d[2])) kept += sums[c1 * kLPad + kk];
    }
    // (a block without weight, or whose kept taps cancel: leave it as it is)
    if (kept != 0.0 && fabs(all) <= 4.0 * fabs(kept)) keep = all / kept;
  }
  const double vk = mco[i] + keep * sums[i] * inv_count * lscale;
  const int L = (e.c2 * 5 + e.d[2] + 2) * 5 + e.d[1] + 2;
  packed[line_slot(L) * kCoefPitch + tap_pos(e.c2, e.d[2], e.d[1], (e.d[0] + 2) * 3 + c1)] = (float)vk;
  // matM's own coefficients, packed per line behind the full table (kind 4)
  if (is_matM_tap(c1, e.c2, e.d[0], e.d[1], e.d[2]))
    packed[kLinesUsed * kCoefPitch + mline_slot(L) * kMPitch + mtap_pos(e.c2, e.d[2], e.d[1], (e.d[0] + 2) * 3 + c1)] = (float)mco[i];
}

// kind 4: r[c1][node] = matL[node][c1][diagonal] / Lbar[c1][diagonal] (1 where the average has no diagonal: vacuum), in the
// layout of a fp32 field vector; the largest ratio goes to *rmax (as the bits of a positive float: integer max)
__global__ void __launch_bounds__(256) k_rscale(GridDev g, const double* __restrict__ matL, const double* __restrict__ abar64,
  const double* __restrict__ mco, float* __restrict__ rsc, unsigned* __restrict__ rmax)
{
  const int c1 = blockIdx.y;
  const int kd = lencode(c1, c1, 0, 0, 0);
  const double lb = abar64[c1 * kLPad + kd] - mco[c1 * kLPad + kd];
  const double inv = lb > 0.0 ? 1.0 / lb : 0.0;
  float mx = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < g.nown; i += (long)gridDim.x * 256) {
    const int x = (int)(i % g.nx), y = (int)((i / g.nx) % g.ny), z = (int)(i / g.plane);
    const double dg = matL[g.lindex(c1, z + (g.G ? 1 : 0), y, x, kd)];
    const float r = lb > 0.0 ? (float)(dg * inv) : 1.f;
    rsc[c1 * g.cstride + (long)g.G * g.plane + i] = r;
    mx = fmaxf(mx, r);
  }
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((threadIdx.x & 63) == 0) atomicMax(rmax, __float_as_uint(mx));
}

template <bool SCALED, int C2, int DZ, int DY, int I>
__device__ __forceinline__ void bar_tap(float (&acc)[2][3], float (&accM)[2][3], const float (&v)[6], const float (&cf)[kCoefPitch],
  const float (&cm)[kMPitch])
{
  constexpr int dx = I / 3 - 2, c1 = I % 3;
  if constexpr (tap_in(c1, C2, dx, DY, DZ)) {
    constexpr int pos = tap_pos(C2, DZ, DY, I);
    const float a = cf[pos];
    acc[0][c1] += a * v[dx + 2];
    acc[1][c1] += a * v[dx + 3];
    if constexpr (SCALED && is_matM_tap(c1, C2, dx, DY, DZ)) {
      constexpr int mpos = mtap_pos(C2, DZ, DY, I);
      const float am = cm[mpos];
      accM[0][c1] += am * v[dx + 2];
      accM[1][c1] += am * v[dx + 3];
    }
  }
}

template <bool SCALED, int C2, int DZ, int DY, int... Is>
__device__ __forceinline__ void bar_line(std::integer_sequence<int, Is...>, float (&acc)[2][3], float (&accM)[2][3], const fpair* tile,
  const int (&sbase)[5], int lbase, const fquad* ctab, const fquad* mtab)
{
  if constexpr (line_used(C2, DY, DZ)) {
    // float-pair units: every offset here is even, and saying so (a pair-typed array) makes the reads ds_read_b64
    const fpair* src = tile + (sbase[DZ + 2] + (C2 * kPY + DY) * kPX + lbase) / 2;
    const fpair v01 = src[0], v23 = src[1], v45 = src[2];
    const float v[6] = {v01.x, v01.y, v23.x, v23.y, v45.x, v45.y}; // x - 2 .. x + 3
    constexpr int L = (C2 * 5 + DZ + 2) * 5 + DY + 2, ntap = tap_pos(C2, DZ, DY, 15);
    constexpr int slot = line_slot(L);
    const fquad* cq = ctab + slot * (kCoefPitch / 4);
    float cf[kCoefPitch] = {};
    { const fquad q = cq[0]; cf[0] = q.x; cf[1] = q.y; cf[2] = q.z; cf[3] = q.w; }
    if constexpr (ntap > 4) { const fquad q = cq[1]; cf[4] = q.x; cf[5] = q.y; cf[6] = q.z; cf[7] = q.w; }
    if constexpr (ntap > 8) { const fquad q = cq[2]; cf[8] = q.x; cf[9] = q.y; cf[10] = q.z; cf[11] = q.w; }
    float cm[kMPitch] = {};
    if constexpr (SCALED && mline_used(C2, DY, DZ)) {
      constexpr int mtap = mtap_pos(C2, DZ, DY, 15);
      constexpr int mslot = mline_slot(L);
      const fquad* mq = mtab + mslot * (kMPitch / 4);
      { const fquad q = mq[0]; cm[0] = q.x; cm[1] = q.y; cm[2] = q.z; cm[3] = q.w; }
      if constexpr (mtap > 4) { const fquad q = mq[1]; cm[4] = q.x; cm[5] = q.y; cm[6] = q.z; cm[7] = q.w; }
    }
    (bar_tap<SCALED, C2, DZ, DY, Is>(acc, accM, v, cf, cm), ...);
  }
  // one scheduling region per (c2, dz) group of lines: left alone the scheduler hoists all reads of a plane to the
  // top (spills); a group is <= 15 window reads in flight over ~150 FMAs, and the other wave of the SIMD covers the rest
  if constexpr (DY == 2 && (BAR_SCHED_GROUP == 1 || (BAR_SCHED_GROUP == 5 && DZ == 2) || (SCALED && BAR_SCHED_SCALED))) __builtin_amdgcn_sched_barrier(0);
}

template <bool SCALED, int... Ls> // L = (c2 * 5 + dz + 2) * 5 + dy + 2
__device__ __forceinline__ void bar_apply(std::integer_sequence<int, Ls...>, float (&acc)[2][3], float (&accM)[2][3], const fpair* tile,
  const int (&sbase)[5], int lbase, const fquad* ctab, const fquad* mtab)
{
  (bar_line<SCALED, Ls / 25, (Ls / 5) % 5 - 2, Ls % 5 - 2>(std::make_integer_sequence<int, 15>{}, acc, accM, tile, sbase, lbase, ctab, mtab), ...);
}

// PAIR: nx is even -- every lane's two nodes (x, x + 1; x even) and every window pair are one aligned 8-byte (fp32) or
// 16-byte (fp64) access; odd nx takes the scalar accesses
// SCALED (kind 4): the rows of Lbar carry the local density ratio rsc (a fp32 vector in the field layout)
template <bool FIRST, bool LAST, bool PAIR, bool SCALED>
__global__ void __launch_bounds__(kB, 2) k_cheb_bar(GridDev g, const float* __restrict__ coef_, const double* __restrict__ r64,
  float* __restrict__ r32, const float* __restrict__ zin, float* __restrict__ d, float* __restrict__ zout,
  double* __restrict__ out64, double cd, double cr, double itheta, int nbx, int nby, int zc, const float* __restrict__ rsc)
{
  __shared__ __attribute__((aligned(16))) fpair tile2[kZW * kPlaneF / 2];
  __shared__ __attribute__((aligned(16))) fquad ctab[kLinesUsed * kCoefPitch / 4];
  __shared__ __attribute__((aligned(16))) fquad mtab[SCALED ? kMLinesUsed * kMPitch / 4 : 1];
  for (int i = threadIdx.x; i < kLinesUsed * kCoefPitch; i += kB) ((float*)ctab)[i] = coef_[i]; // packed by k_abar
  if (SCALED)
    for (int i = threadIdx.x; i < kMLinesUsed * kMPitch; i += kB) ((float*)mtab)[i] = coef_[kLinesUsed * kCoefPitch + i];
  const int lane = threadIdx.x & 63, wy = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bx = blockIdx.x % nbx, by = (blockIdx.x / nbx) % nby, bz = blockIdx.x / (nbx * nby);
  const int x0 = bx * kTX, y0 = by * kTY, z0 = bz * zc;
  const int z1 = min(z0 + zc, g.nzl);
  if (z0 >= z1) return;

  // this thread's share of a window plane: pairs e = thread + k * 256 of [3][kPY][kPX / 2]; their offsets inside a
  // z-plane of the vector (periodic in x and y) are fixed for the whole march
  constexpr int kPairs = kPlaneF / 2, kPerP = (kPairs + kB - 1) / kB;
  int goff[kPerP];
#pragma unroll
  for (int k = 0; k < kPerP; ++k) {
    const int i = threadIdx.x + k * kB;
    goff[k] = -1;
    if (i < kPairs) {
      const int comp = i / (kPY * kPX / 2), rem = i % (kPY * kPX / 2), yy = rem / (kPX / 2), xx = 2 * (rem % (kPX / 2));
      int gx = (x0 - kH + xx) % g.nx, gy = (y0 - kH + yy) % g.ny;
      gx += gx < 0 ? g.nx : 0;
      gy += gy < 0 ? g.ny : 0;
      goff[k] = (int)(comp * g.cstride + (long)gy * g.nx + gx); // a component-major vector has < 2^31 elements
    }
  }
  const int nxm = g.nx; // (odd nx: the pair's second node is the wrapped x + 1)
  auto fetch = [&](int p, fpair (&v)[kPerP]) { // unwrapped owned-plane number p in [-2, nzl + 1] (clamped reads beyond)
    const long zoff = (long)g.wz(min(p, g.nzl + 1)) * g.plane;
#pragma unroll
    for (int k = 0; k < kPerP; ++k) {
      v[k] = fpair{0.f, 0.f};
      if (goff[k] < 0) continue;
      const long o = zoff + goff[k];
      if (PAIR) {
        if (FIRST) { const dpairv t = *reinterpret_cast<const dpairv*>(r64 + o); v[k] = fpair{(float)t.x, (float)t.y}; }
        else v[k] = *reinterpret_cast<const fpair*>(zin + o);
      }
      else {
        // second node: x + 1 wrapped inside its row
        const int gx = (int)((o - zoff) % nxm);
        const long o1 = gx + 1 < nxm ? o + 1 : o + 1 - nxm;
        v[k] = FIRST ? fpair{(float)r64[o], (float)r64[o1]} : fpair{zin[o], zin[o1]};
      }
    }
  };
  auto store = [&](int p, const fpair (&v)[kPerP]) {
    fpair* dst = tile2 + ((p - z0 + 2) % kZW) * kPairs + threadIdx.x;
#pragma unroll
    for (int k = 0; k < kPerP; ++k)
      if (goff[k] >= 0) dst[k * kB] = v[k];
  };

  // The window holds 6 planes: compute(z) reads z - 2 .. z + 2 while plane z + 3 (requested an iteration earlier, kept in
  // registers meanwhile) is written into the slot plane z - 3 left -- ONE barrier per plane, in front of that write, and
  // the global latency of a plane runs under the FMAs of the previous one.
  fpair nxt[kPerP];
  for (int p = z0 - 2; p <= z0 + 2; ++p) { fetch(p, nxt); store(p, nxt); }
  fetch(z0 + 3, nxt);
  const int x = x0 + 2 * lane, y = y0 + wy;
  const bool row_ok = y < g.ny && x < g.nx;
  const bool two = x + 1 < g.nx;
  for (int z = z0; z < z1; ++z) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // every wave is done with plane z - 3 (and sees z + 2)
    store(z + 3, nxt);
    if (z + 1 < z1) fetch(z + 4, nxt);
    if (row_ok) { // (rows beyond ny, lanes beyond nx only help loading the window)
      // the right-hand side and the direction of this lane's two nodes are requested BEFORE the stencil sums: their HBM
      // latency runs under the 738 FMAs instead of behind them
      const long oc0 = (long)g.wz(z) * g.plane + (long)y * g.nx + x;
      fpair rv2[3], dv2[3], rs2[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const long oc = c * g.cstride + oc0;
        if (SCALED) rs2[c] = PAIR ? *reinterpret_cast<const fpair*>(rsc + oc) : fpair{rsc[oc], two ? rsc[oc + 1] : 1.f};
        if (PAIR) {
          if (FIRST) { const dpairv t = *reinterpret_cast<const dpairv*>(r64 + oc); rv2[c] = fpair{(float)t.x, (float)t.y}; }
          else { rv2[c] = *reinterpret_cast<const fpair*>(r32 + oc); dv2[c] = *reinterpret_cast<const fpair*>(d + oc); }
        }
        else {
          if (FIRST) rv2[c] = fpair{(float)r64[oc], two ? (float)r64[oc + 1] : 0.f};
          else { rv2[c] = fpair{r32[oc], two ? r32[oc + 1] : 0.f}; dv2[c] = fpair{d[oc], two ? d[oc + 1] : 0.f}; }
        }
      }
      float acc[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}}, accM[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
      int sbase[5]; // float offset of the window plane of z + dz
#pragma unroll
      for (int dz = -2; dz <= 2; ++dz) sbase[dz + 2] = ((z + dz - z0 + 2) % kZW) * kPlaneF;
      const int lbase = (wy + kH) * kPX + 2 * lane;
      bar_apply<SCALED>(std::make_integer_sequence<int, 75>{}, acc, accM, tile2, sbase, lbase, ctab, mtab);
      // pin the sums here: their only users sit behind the stores below, and the optimizer otherwise sinks all 738 FMAs
      // there while the 135 LDS reads stay in front (270 live VGPRs: spills)
      asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]));
      if (SCALED) {
        asm volatile("" : "+v"(accM[0][0]), "+v"(accM[0][1]), "+v"(accM[0][2]), "+v"(accM[1][0]), "+v"(accM[1][1]), "+v"(accM[1][2]));
        // (Abar_r z) = matM z + r (Lbar z), Lbar z = (Abar z) - (matM z)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          acc[0][c] = accM[0][c] + rs2[c].x * (acc[0][c] - accM[0][c]);
          acc[1][c] = accM[1][c] + rs2[c].y * (acc[1][c] - accM[1][c]);
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const long oc = c * g.cstride + oc0;
        const fpair zc2 = tile2[(sbase[2] + c * kPY * kPX + lbase + kH) / 2]; // the window's centre values
        float dn[2], zn[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float rv = e ? rv2[c].y : rv2[c].x;
          // FIRST: z0 = r / theta and Abar z0 = (Abar r) / theta
          const float zv = FIRST ? rv * (float)itheta : (e ? zc2.y : zc2.x);
          const float m = FIRST ? acc[e][c] * (float)itheta : acc[e][c];
          dn[e] = (float)cd * (FIRST ? zv : (e ? dv2[c].y : dv2[c].x)) + (float)cr * (rv - m);
          zn[e] = zv + dn[e];
        }
        if (PAIR) {
          if (FIRST && !LAST) *reinterpret_cast<fpair*>(r32 + oc) = rv2[c];
          if (LAST) *reinterpret_cast<dpairv*>(out64 + oc) = dpairv{(double)zn[0], (double)zn[1]};
          else { *reinterpret_cast<fpair*>(d + oc) = fpair{dn[0], dn[1]}; *reinterpret_cast<fpair*>(zout + oc) = fpair{zn[0], zn[1]}; }
        }
        else {
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            if (e && !two) continue;
            if (FIRST && !LAST) r32[oc + e] = e ? rv2[c].y : rv2[c].x;
            if (LAST) out64[oc + e] = (double)zn[e];
            else { d[oc + e] = dn[e]; zout[oc + e] = zn[e]; }
          }
        }
      }
    }
  }
}

}  // namespace

int halo_fill_f32(xpic_ctx* c, float* f, int width);

namespace {

// rows of the translation average: every 4th row in y and z of extents that have 16, every 8th of extents that have 64
// (256^3: 3 x 262 144 rows x 256 nodes, the average's own noise 1e-4 of an entry; the pass reads 0.8 GB instead of 3.1)
inline void lbar_rows(const GridDev& g, int* sy, int* sz, int* nys, int* nrows)
{
  *sy = g.ny >= 64 ? 8 : (g.ny >= 16 ? 4 : 1);
  *sz = g.nzl >= 64 ? 8 : (g.nzl >= 16 ? 4 : 1);
  *nys = (g.ny + *sy - 1) / *sy;
  *nrows = *nys * ((g.nzl + *sz - 1) / *sz);
}

// bounds of Lbar = Abar - matM per row component c1 (one wave each): out[c1] = sum_k |Lbar[c1][k]| (widens the top of the
// spectral interval), out[3 + c1] = Lbar's diagonal minus the absolute sum of its other entries (Gershgorin: a lower
// bound of the real parts of Lbar's eigenvalues, rotation part included)
__global__ void __launch_bounds__(64) k_abar_bounds(const double* __restrict__ abar64, const double* __restrict__ mco, double* out)
{
  const int c1 = blockIdx.x, lane = threadIdx.x;
  const int kdiag = lencode(c1, c1, 0, 0, 0);
  double sa = 0.0, diag = 0.0;
  for (int k = lane; k < kLStencil; k += 64) {
    const double v = abar64[c1 * kLPad + k] - mco[c1 * kLPad + k];
    sa += fabs(v);
    if (k == kdiag) diag = v;
  }
  for (int o = 32; o > 0; o >>= 1) { sa += __shfl_xor(sa, o); diag += __shfl_xor(diag, o); }
  if (lane == 0) {
    out[c1] = sa; out[3 + c1] = diag - (sa - fabs(diag));
    // relative variance of matL's diagonal over the sampled rows: <d^2> / <d>^2 - 1 (slot 123 holds <d^2>, see k_diag2_rows)
    const double d2 = abar64[c1 * kLPad + kLStencil];
    out[7 + c1] = diag > 0.0 ? d2 / (diag * diag) - 1.0 : 0.0;
  }
}

}  // namespace

// kind 3's buffers and matM's probed coefficients: sized and uploaded with the context (api.hip:
// ensure_flexible_workspace, at xpic_create / xpic_set_preconditioner time), like kry_Z -- an allocation inside the first
// solve of a step could fail on one slab alone and leave the others waiting in the solve's collectives
int abar_alloc(xpic_ctx* c)
{
  if (c->abar32) return 0;
  const GridDev& g = c->g;
  int sy, sz, nys, nrows;
  lbar_rows(g, &sy, &sz, &nys, &nrows);
  const size_t ntab = (size_t)kLinesUsed * kCoefPitch + (size_t)kMLinesUsed * kMPitch; // full table, then matM's own taps
  XPIC_HIP(hipMalloc(&c->abar32, sizeof(float) * ntab));
  XPIC_HIP(hipMemsetAsync(c->abar32, 0, sizeof(float) * ntab, c->stream));
  XPIC_HIP(hipMalloc(&c->abar_r, sizeof(float) * c->nvec + 16)); // kind 4: local density ratios (fp32 field layout) + the max word
  XPIC_HIP(hipMemsetAsync(c->abar_r, 0, sizeof(float) * c->nvec + 16, c->stream));
  XPIC_HIP(hipMalloc(&c->abar_work, sizeof(double) * (3 * kLPad * 3 + ((size_t)nrows + kSegs) * 3 * kLBlock + (size_t)nrows * 3)));
  double mco[3 * kLPad];
  matM_stencil(g, mco);
  // the compile-time list of matM's taps (is_matM_tap) against the probed coefficients
  for (int c1 = 0; c1 < 3; ++c1)
    for (int k = 0; k < kLStencil; ++k) {
      const LEntry e = ldecode(c1, k);
      XPIC_CHECK((mco[c1 * kLPad + k] != 0.0) == is_matM_tap(c1, e.c2, e.d[0], e.d[1], e.d[2]) || g.dt == 0.0,
        "matM's stencil does not match the tap list of the scaled surrogate");
    }
  XPIC_HIP(hipMemcpyAsync(c->abar_work + 3 * kLPad, mco, sizeof(mco), hipMemcpyHostToDevice, c->stream));
  XPIC_HIP(hipStreamSynchronize(c->stream)); // mco is a stack array
  return 0;
}

// (re)build Abar = matM + <matL>: called once per assembly, before the predict solve.  No allocation, no pageable copy:
// the two bounds the host needs come back through the pinned reduction mirror.
int abar_update(xpic_ctx* c)
{
  const GridDev& g = c->g;
  Timed t(c, "precond_setup");
  XPIC_CHECK(c->abar32 && c->abar_work, "kind-3 preconditioner workspace missing (xpic_set_preconditioner allocates it)");
  int sy, sz, nys, nrows;
  lbar_rows(g, &sy, &sz, &nys, &nrows);
  double* sums = c->abar_work;                  // [3][kLPad]
  double* mco = c->abar_work + 3 * kLPad;       // matM's coefficients
  double* abar64 = c->abar_work + 6 * kLPad;
  double* partial = c->abar_work + 9 * kLPad;
  hipLaunchKernelGGL(k_lbar_rows, dim3(nrows, 3), dim3(512), 0, c->stream, g, c->matL, sy, sz, nys, partial);
  double* segsum = partial + (size_t)nrows * 3 * kLBlock;
  hipLaunchKernelGGL(k_lbar_segments, dim3(3, kSegs), dim3(512), 0, c->stream, partial, nrows, segsum);
  hipLaunchKernelGGL(k_lbar_final, dim3(3), dim3(512), 0, c->stream, segsum, sums);
  if (c->precond == 5) { // (behind k_lbar_final, which writes all 124 slots of a component)
    double* part2 = segsum + (size_t)kSegs * 3 * kLBlock;
    hipLaunchKernelGGL(k_diag2_rows, dim3(nrows, 3), dim3(256), 0, c->stream, g, c->matL, sy, sz, nys, part2);
    hipLaunchKernelGGL(k_diag2_final, dim3(3), dim3(256), 0, c->stream, part2, nrows, sums);
  }
  XPIC_HIP(hipGetLastError());
  XPIC_CALL(comm_allreduce_sum(c, sums, 3 * kLPad)); // the same surrogate on every slab
  const double count = (double)nrows * g.nx * c->comm.nranks;
  hipLaunchKernelGGL(k_abar, dim3(2), dim3(256), 0, c->stream, sums, mco, 1.0 / count, c->abar32, abar64, c->debug_surrogate_scale);
  double* bounds = c->red_out + 64; // [64, 74): behind the reductions' and the host all-reduce's slots
  hipLaunchKernelGGL(k_abar_bounds, dim3(3), dim3(64), 0, c->stream, abar64, mco, bounds);
  double* hb = c->red_host + 48;
  // Kind 5 chooses per solve: the plain surrogate for a plasma that is uniform up to its count noise, the density-scaled
  // one (kind 4's) where the density itself varies.  The measure is the relative spread of matL's diagonal over the sampled
  // rows, sqrt(<d^2> / <d>^2 - 1): 0.1 for a uniform Poisson(64) load (256^3: kind 3 takes 4 iterations of degree 8, kind 4
  // four dearer ones: 184 against 195 ms per step), 0.37 with the density falling 4 : 1 across the box (kind 3: 6
  // iterations, kind 4: 4; 218 against 202 ms per step).  It travels in the all-reduced sums: every slab decides alike.
  // (one synchronisation for the bounds and the spread; a second one only where the rows are scaled, for their largest ratio)
  XPIC_HIP(hipGetLastError());
  XPIC_HIP(hipMemcpyAsync(hb, bounds, sizeof(double) * 10, hipMemcpyDeviceToHost, c->stream));
  XPIC_HIP(hipStreamSynchronize(c->stream));
  bool scaled = c->precond == 4;
  if (c->precond == 5) {
    const double cv2 = std::max(hb[7], std::max(hb[8], hb[9]));
    scaled = std::isfinite(cv2) && cv2 > kAutoSpread * kAutoSpread;
  }
  c->abar_scaled = scaled;
  if (scaled && c->profiling) c->prof["precond_scaled"].launches += 1;
  if (scaled) {
    unsigned* rmax_w = (unsigned*)(c->abar_r + c->nvec);
    XPIC_HIP(hipMemsetAsync(rmax_w, 0, sizeof(unsigned), c->stream));
    const unsigned nb = (unsigned)std::min<long>((g.nown + 255) / 256, 2048);
    hipLaunchKernelGGL(k_rscale, dim3(nb, 3), dim3(256), 0, c->stream, g, c->matL, abar64, mco, c->abar_r, rmax_w);
    XPIC_HIP(hipGetLastError());
    XPIC_HIP(hipMemcpyAsync(hb + 6, rmax_w, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    XPIC_HIP(hipStreamSynchronize(c->stream));
  }
  double rmax = 1.0;
  if (scaled) {
    unsigned bits;
    memcpy(&bits, hb + 6, sizeof(bits));
    float f;
    memcpy(&f, &bits, sizeof(f));
    double loc = (double)f;
    XPIC_CALL(comm_allreduce_max_host(c, &loc)); // the same interval (and the same degree) on every slab
    rmax = std::max(1.0, loc);
  }
  // Spectral interval of Abar for the Chebyshev polynomial.  Top: matM's exact 2 + 2 dt^2 sum 1/h^2 widened by the largest
  // absolute row sum of Lbar.  Bottom: matM's exact 2 -- the Hermitian part of every particle's block is positive
  // semi-definite ((s s^T) o (I + b b^T)) and a translation average keeps that.  Lbar's Gershgorin bound (rotation part of
  // a strong B and the sub-sampled average included) PROVES the interval's positive side when 2 + rmax min_c (diag - sum
  // |off-diagonal|) > 0 (abar_proven).  The bound is crude -- the CIC overlap stencil is positive semi-definite with
  // diag - sum |off| = -0.41 of its row sum -- and fails for any plasma a few times denser than the reference density or
  // with a clump in it (rmax = 160 was measured on a blob whose solve the scaled surrogate brings from 41 iterations to 5):
  // an unproven surrogate is therefore TRIED, on probation (krylov.hip): the first iteration it fails to halve the
  // residual, or returns something that is not finite, the solve drops it for the matM polynomial (kind 1), on every slab
  // alike (the residual norms are all-reduced).
  const double rs = std::max(hb[0], std::max(hb[1], hb[2]));
  const double gl = std::min(hb[3], std::min(hb[4], hb[5]));
  c->abar_lo = 2.0;
  c->abar_hi = 2.0 + 2.0 * g.dt * g.dt * (1.0 / (g.dx * g.dx) + 1.0 / (g.dy * g.dy) + 1.0 / (g.dz * g.dz)) + rmax * rs;
  c->abar_gershgorin = 2.0 + gl;
  c->abar_valid = std::isfinite(rs) && std::isfinite(rmax) && std::isfinite(gl);
  c->abar_proven = c->abar_valid && 2.0 + rmax * std::min(gl, 0.0) > 0.0;
  if (c->debug_surrogate_scale != 1.0) c->abar_proven = false; // (test hook: the surrogate below is deliberately wrong)
  return 0;
}

// out ~ Abar^-1 r: `degree` steps of the Chebyshev iteration on [abar_lo, abar_hi]
int cheb_abar_inverse(xpic_ctx* c, const double* r, double* out)
{
  Timed t(c, "precond");
  const GridDev& g = c->g;
  XPIC_CHECK(c->abar_valid, "the matL surrogate of the preconditioner was not built (abar_update)");
  const double a = c->abar_lo, b = c->abar_hi;
  const double theta = 0.5 * (b + a), delta = 0.5 * (b - a), sigma1 = theta / delta;
  // error bound 2 rho^k / (1 + rho^2k) <= 2.5 % (what is left for GMRES is the 2 % noise of matL around its average);
  // kind 4 leaves 0.8 % and needs the polynomial at 0.25 % to keep its third iteration below the tolerance (degree 12
  // instead of 8 at dt = 1, h = 0.5: tools/precond_spectrum.py)
  const bool scaled = c->abar_scaled;
  const double kappa = b / a, rh = (std::sqrt(kappa) - 1.0) / (std::sqrt(kappa) + 1.0);
  int degree = c->cheb_degree_user > 0 ? c->cheb_degree_user : (int)std::ceil(std::log(scaled ? 0.00125 : 0.0125) / std::log(rh));
  degree = degree < 2 ? 2 : (degree > 64 ? 64 : degree);
  if (c->profiling) c->prof["cheb_steps"].launches += degree - 1; // stencil steps of the polynomial (bench.py: per outer iteration)
  float* d = (float*)c->kry_p[0];
  float* z0 = (float*)c->kry_p[1];
  float* z1 = (float*)c->kry_p[2];
  float* r32 = (float*)c->kry_t;
  const int nbx = (g.nx + kTX - 1) / kTX, nby = (g.ny + kTY - 1) / kTY;
  // z-chunks: enough workgroups for two per CU, chunks of at least 8 planes (5 more planes are loaded per chunk)
  int nzc = (int)((2 * c->num_cus + (long)nbx * nby - 1) / ((long)nbx * nby));
  int zc = (g.nzl + nzc - 1) / nzc;
  if (zc < XPIC_CHEB_MIN_ZC) zc = g.nzl < XPIC_CHEB_MIN_ZC ? g.nzl : XPIC_CHEB_MIN_ZC;
  if (const char* e = getenv("XPIC_CHEB_ZC")) { const int v = atoi(e); if (v > 0) zc = v < g.nzl ? v : g.nzl; } // (measurements)
  nzc = (g.nzl + zc - 1) / zc;
  const dim3 grid((unsigned)(nbx * nby * nzc)), block(kB);
  double rho = 1.0 / sigma1;
  for (int i = 1; i < degree; ++i) {
    const double rho_new = 1.0 / (2.0 * sigma1 - rho);
    const double cd = rho_new * rho, cr = 2.0 * rho_new / delta, it = 1.0 / theta;
    const bool first = i == 1, last = i == degree - 1;
    if (first) XPIC_CALL(halo_fill(c, const_cast<double*>(r), 2));
    else XPIC_CALL(halo_fill_f32(c, z0, 2));
#define LAUNCH2(F, L, P, S)                                                                                                \
  hipLaunchKernelGGL((k_cheb_bar<F, L, P, S>), grid, block, 0, c->stream, g, c->abar32, r, r32, z0, d, z1, out, cd, cr, it, nbx, \
    nby, zc, c->abar_r)
#define LAUNCH(F, L)                                                                                                        \
  do {                                                                                                                      \
    if (g.nx % 2 == 0) { if (scaled) LAUNCH2(F, L, true, true); else LAUNCH2(F, L, true, false); }                          \
    else { if (scaled) LAUNCH2(F, L, false, true); else LAUNCH2(F, L, false, false); }                                      \
  } while (0)
    if (first && last) LAUNCH(true, true);
    else if (first) LAUNCH(true, false);
    else if (last) LAUNCH(false, true);
    else LAUNCH(false, false);
#undef LAUNCH
#undef LAUNCH2
    XPIC_HIP(hipGetLastError());
    rho = rho_new;
    std::swap(z0, z1);
  }
  return 0;
}

}  // namespace xpic
