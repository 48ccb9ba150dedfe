// ecsim.hip -- ecsim::Particles::fill_ecsim_current / decompose_ecsim_current
// (src/impls/ecsim/particles.cpp:33-173) for cell-sorted SoA particles.
//
// All particles of a cell touch the same 3 x 12 Yee nodes, so the cell's contribution to matL is a dense
// 36 x 36 block  M = sum_p A_p * (s_p s_p^T) o matB_p  (the reference's 1296-entry COO block, :145-163)
// and its contribution to currI is the 36-vector sum_p s_p o I_p.
//
// Work decomposition (v2, "pencil march"):
//   * one workgroup (8 waves) owns one x-pencil of cells (fixed cy, cz) and marches along x in chunks of
//     8 cells, one cell per wave;
//   * per cell:  phase 1 (lane = particle)  CIC weights, B gather, b, I_p, A_p*matB -> the wave's LDS stage;
//                phase 2 (lane = 4 x 6 tile) rank-1 updates out of LDS into 24 register accumulators;
//   * per chunk: the 8 cell blocks are merged in an LDS window indexed [matL line][x] (a "line" is one
//     (row component, row y/z offset, k) coefficient stream of the index-free matL), which sums the
//     duplicates of x-neighbouring cells on chip; finished columns are streamed out with plain,
//     64-byte-aligned read-modify-write; two unfinished columns are carried to the next chunk;
//   * pencils whose rows overlap (|dcy| <= 2 and |dcz| <= 2, periodically) never run in the same launch:
//     launches are coloured by (cy mod 3, cz mod 3) (+ remainder colours), so the RMW needs no atomics and
//     the result is bitwise reproducible.  MatSetValuesCOO's duplicate summation (simulation.cpp:366) thus
//     happens in LDS (x) and in launch order (y, z).
// v1 flushed every cell block with ~1200 scattered fp64 atomics: 85 % of the kernel time at 256^3.
#include <algorithm>
#include <array>
#include <cstdint>
#include <map>
#include <vector>

#include "common.h"
#include "device_common.h"
#include "lstencil.h"

namespace xpic {

namespace {

constexpr int kW = 4;             // waves per workgroup = cells per chunk
constexpr int kCP = 40;           // particles staged per pass and wave (cells of 64 +- 8 fit two passes)
constexpr int kPadP = kCP + 2;    // LDS row pitch; 2 * 42 mod 64 = 20: the 16 rows of an MFMA operand fall in 16 distinct bank quads
constexpr int kRows = 50;         // 36 weights + 9 A_p*matB + 3 I_p + a row of ones + a row of zeros
constexpr int kStage = kRows * kPadP;
constexpr int kRowOne = 48, kRowZero = 49;
typedef double mfma_acc __attribute__((ext_vector_type(4)));
constexpr int kMatLines = 816;    // distinct (c1, row dy, row dz, k) streams one pencil can touch
constexpr int kCurLines = 16;     // (c, dy, dz) streams of currI
constexpr int kLines = kMatLines + kCurLines;
constexpr int kThreads = kW * 64;
constexpr int kSlots = kW + 2;    // window columns: kW finished + 2 carried
constexpr int kOwn = (kLines + kThreads - 1) / kThreads; // window lines owned by a thread (init, flush, carry)
constexpr int kMaxNxLds = 1024;   // pencils up to this length keep their cell_start row in LDS

static_assert(kLines * kSlots <= kW * kStage, "the merge window must fit in the (dead) staging area");

__device__ inline void wave_sync()
{
  // the stage of a wave is private to it and a wave's LDS operations complete in order: draining the LDS
  // counter orders its writes before its reads.  No workgroup barrier, and (unlike a fence) no wait on
  // the global prefetches in flight.
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// raw workgroup barrier that only drains LDS traffic: global prefetches stay in flight across it
__device__ inline void lds_barrier()
{
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

struct Prefetch {
  int start, cnt;  // cell_start of the cell this wave handles next
  double p[6];     // x, y, z, vx, vy, vz of lane's particle of the next pass
  double b;        // lane's value of the next cell's 54-value B neighbourhood
};

// B neighbourhood of cell (cx,cy,cz) = every B value a CIC gather from inside that cell can touch:
//   Bx at (xn in cx..cx+1, ys in cy-1..cy+1, zs in cz-1..cz+1)   -> [ 0,18): (kl*3 + jl)*2 + i
//   By at (xs in cx-1..cx+1, yn in cy..cy+1, zs in cz-1..cz+1)   -> [18,36): (kl*2 + j)*3 + il
//   Bz at (xs in cx-1..cx+1, ys in cy-1..cy+1, zn in cz..cz+1)   -> [36,54): (k*3 + jl)*3 + il
__device__ inline double load_bnb(const GridDev& g, const double* __restrict__ B, int lane, int cx, int cy, int cz)
{
  if (lane >= 54) return 0.0;
  int c, ox, oy, oz;
  if (lane < 18) { c = 0; ox = lane % 2; oy = (lane / 2) % 3 - 1; oz = lane / 6 - 1; }
  else if (lane < 36) { const int l = lane - 18; c = 1; ox = l % 3 - 1; oy = (l / 3) % 2; oz = l / 6 - 1; }
  else { const int l = lane - 36; c = 2; ox = l % 3 - 1; oy = (l / 3) % 3 - 1; oz = l / 9; }
  return B[c * g.cstride + g.nodew(cx + ox, cy + oy, cz + oz)];
}

__global__ void __launch_bounds__(kThreads, 2) k_ecsim_fill(GridDev g, SortDev s, const double* __restrict__ B,
  double* currI, double* matL, const int* __restrict__ etab, const int* __restrict__ linetab, const int* __restrict__ cowr, double q, double m,
  double mpw, int cy0, int cystep, int ncy, int cz0, int czstep, int my_order, int ncol_y, int per_y, int per_z, int first_sort)
{
  const int cy = cy0 + (int)(blockIdx.x % ncy) * cystep;
  const int cz = cz0 + (int)(blockIdx.x / ncy) * czstep;

  __shared__ __attribute__((aligned(16))) double sh[kW * kStage];
  __shared__ int cstart[kMaxNxLds + 2];
  __shared__ double bnb[kW][54];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double* st = sh + wave * kStage;
  const double dt = g.dt;

  // Phase 2 runs on the matrix cores.  For a row component c1 the cell block is the GEMM
  //   D_c1[12 x 36] = S_c1^T [12 x P] * ( S [P x 36] o matB_c1 )        (P = particles of the cell)
  // over K = 4 particles per step.  Columns 0..31 go through v_mfma_f64_16x16x4_f64 (A = the 12 (+4 unused) rows of the
  // component, B = two 16-column tiles): 6 instructions.  The last four columns 32..35 and the cell's currI
  // (sum_p s_p I_p[c1], i.e. the product with the stage rows 45..47) would fill a third 16-column tile only to 5/16;
  // they are 18 blocks of 4 x 4 instead and take 5 v_mfma_f64_4x4x4_4b_f64 (4 independent blocks each).
  // 16x16x4 lane roles: operand element (i or j = lane & 15, k = lane >> 4); result rows (lane >> 4) + 4 r, r < 3.
  // 4x4x4 lane roles (probed, tools/ubench/mfma_f64_4x4.hip): A[b][i][k], B[b][k][j] at lane 16 k + 4 b + (i or j);
  // D[b][i][j] at lane 16 i + 4 b + j.
  const int mj = lane & 15, mk = lane >> 4;
  const double* a_ptr = st + min(mj, 11) * kPadP + mk;                 // + c1 * 12 rows
  const double* b_ptr = st + mj * kPadP + mk;                          // + t * 16 rows
  const double* m0_ptr = st + (36 + (mj < 12 ? 0 : 1)) * kPadP + mk;  // tile 0: columns 0..15   (+ c1 * 3 rows)
  const double* m1_ptr = st + (36 + (mj < 8 ? 1 : 2)) * kPadP + mk;   // tile 1: columns 16..31
  // 4x4x4 instructions n = 0..4, block b = (lane >> 2) & 3 of each:
  //   n = 0, 1: row blocks 4 n + b, columns 32..35          n = 2, 3: row blocks 4 (n - 2) + b, currI
  //   n = 4   : b = 0: row block 8, columns 32..35;  b = 1: row block 8, currI;  b = 2, 3: idle (zeros)
  const int qb = (lane >> 2) & 3, qj = lane & 3;
  const double* a4_ptr[3] = {st + mj * kPadP + mk, st + (16 + mj) * kPadP + mk, st + (32 + qj) * kPadP + mk};
  const double* x4_ptr[5]; // first factor of B
  const double* y4_ptr[3]; // second factor of B for n = 0, 1, 4 (the currI operands need none)
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int c1n = (4 * n + qb) / 3; // row component of this block's rows
    x4_ptr[n] = st + (32 + qj) * kPadP + mk;
    y4_ptr[n] = st + (36 + 3 * c1n + 2) * kPadP + mk;
    x4_ptr[2 + n] = st + (qj == c1n ? 45 + qj : kRowZero) * kPadP + mk;
  }
  x4_ptr[4] = st + (qb == 0 ? 32 + qj : (qb == 1 && qj == 2 ? 47 : kRowZero)) * kPadP + mk;
  y4_ptr[2] = st + (qb == 0 ? 44 : (qb == 1 ? kRowOne : kRowZero)) * kPadP + mk;

  // per-lane flush descriptors (line << 2 | row x offset + 1) of the 18 + 5 results, constant over the march; two
  // 16-bit descriptors per register, 0xffff = nothing to add (structural zero, unused row, idle block)
  auto curdesc = [&](int row) {
    const int c = row / 12;
    int o[3];
    block_node_offset(c, row % 12, o);
    const int id = c == 0 ? o[2] * 2 + o[1] : (c == 1 ? 4 + o[2] * 3 + (o[1] + 1) : 10 + (o[2] + 1) * 2 + o[1]);
    return ((kMatLines + id) << 2) | (o[0] + 1);
  };
  constexpr int kDesc = 18 + 5;
  unsigned edesc[(kDesc + 1) / 2];
#pragma unroll
  for (int e = 0; e < (kDesc + 1) / 2; ++e) edesc[e] = 0xffffffffu;
  auto set_desc = [&](int e, int d) {
    edesc[e / 2] = (edesc[e / 2] & ~(0xffffu << (16 * (e & 1)))) | (((unsigned)d & 0xffffu) << (16 * (e & 1)));
  };
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 3; ++r) set_desc((c * 2 + t) * 3 + r, etab[(12 * c + mk + 4 * r) * 36 + 16 * t + mj]);
  {
    const int di = lane >> 4; // D[b][i][j]: i = lane >> 4
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int row = 4 * (4 * n + qb) + di;
      set_desc(18 + n, etab[row * 36 + 32 + qj]);
      set_desc(18 + 2 + n, qj == row / 12 ? curdesc(row) : -1);
    }
    const int row8 = 32 + di;
    set_desc(18 + 4, qb == 0 ? etab[row8 * 36 + 32 + qj] : (qb == 1 && qj == 2 ? curdesc(row8) : -1));
  }

  const long pencil0 = ((long)cz * g.ny + cy) * g.nx;
  const bool cs_lds = g.nx <= kMaxNxLds;
  if (cs_lds)
    for (int i = threadIdx.x; i <= g.nx; i += kThreads) cstart[i] = s.cell_start[pencil0 + i];
  // address of column 0 of the window lines this thread owns (line = thread + mm * kThreads), first-touch flag in bit 0
  uintptr_t lbase[kOwn];
#pragma unroll
  for (int mm = 0; mm < kOwn; ++mm) {
    const int line = threadIdx.x + mm * kThreads;
    lbase[mm] = 0;
    if (line >= kLines) continue;
    const int ld = linetab[line];
    const int ry = g.wy(cy + ((ld >> 2) & 3) - 1);
    const int rz = cz + ((ld >> 4) & 3) - 1;
    // single slab: periodic fold; with z-neighbours matL carries one ghost row plane on each side
    const int rzw = g.G == 0 ? (rz < 0 ? rz + g.nzl : (rz >= g.nzl ? rz - g.nzl : rz)) : rz + 1;
    double* base = line < kMatLines
      ? matL + g.lindex(ld & 3, rzw, ry, 0, ld >> 6)
      : currI + (ld & 3) * g.cstride + g.node(0, ry, g.wz(rz));
    // First touch: if no other pencil that also writes this matL line runs in an EARLIER launch, this workgroup
    // is the first writer of the step and stores instead of read-modify-write (no memset of matL, half the reads).
    // Co-writers sit at pencil offsets (-dy', -dz') listed in cowr[line]; launch order = cz colour * ncol_y + cy colour.
    bool first = first_sort && line < kMatLines;
    if (first) {
      const int bodyy = g.ny - g.ny % per_y, bodyz = g.nzl - g.nzl % per_z;
      for (int e = 0; e < 8 && first; ++e) {
        const int w = cowr[line * 8 + e];
        if (w == 0x7fffffff) break;
        const int oy = (w & 0xff) - 8, oz = ((w >> 8) & 0xff) - 8;
        int py = cy + oy, pz = cz + oz;
        py = py < 0 ? py + g.ny : (py >= g.ny ? py - g.ny : py);
        if (g.G == 0) pz = pz < 0 ? pz + g.nzl : (pz >= g.nzl ? pz - g.nzl : pz);
        else if (pz < 0 || pz >= g.nzl) continue; // no such local pencil: the neighbour rank's rows are its own
        const int ca = py < bodyy ? py % per_y : per_y + (py - bodyy);
        const int cb = g.G == 0 ? (pz < bodyz ? pz % per_z : per_z + (pz - bodyz)) : pz % 3;
        if (cb * ncol_y + ca < my_order) first = false;
      }
    }
    lbase[mm] = (uintptr_t)base | (first ? 1u : 0u);
  }

  // cstart is read by other threads from the first chunk on
  __syncthreads();

  // cells are visited in the order 1, 2, ..., nx-1, 0 so that finished columns start 64-byte aligned
  auto cell_x = [&](int i) { return (i + 1 == g.nx) ? 0 : i + 1; };
  auto prefetch_cell = [&](int i, Prefetch& pf) {
    pf.start = 0; pf.cnt = 0; pf.b = 0.0;
    if (i >= g.nx) return;
    const int cx = cell_x(i);
    if (cs_lds) {
      pf.start = __builtin_amdgcn_readfirstlane(cstart[cx]);
      pf.cnt = __builtin_amdgcn_readfirstlane(cstart[cx + 1]) - pf.start;
    }
    else {
      pf.start = __builtin_amdgcn_readfirstlane(s.cell_start[pencil0 + cx]);
      pf.cnt = __builtin_amdgcn_readfirstlane(s.cell_start[pencil0 + cx + 1]) - pf.start;
    }
    pf.b = load_bnb(g, B, lane, cx, cy, cz);
    if (lane < min(kCP, pf.cnt)) {
      const long p = (long)pf.start + lane;
#pragma unroll
      for (int a = 0; a < 3; ++a) { pf.p[a] = s.r[a][p]; pf.p[3 + a] = s.v[a][p]; }
    }
  };

  Prefetch pf;
  prefetch_cell(wave, pf);
  double carry[kOwn][2];
#pragma unroll
  for (int mm = 0; mm < kOwn; ++mm) carry[mm][0] = carry[mm][1] = 0.0;

  const int nch = (g.nx + kW - 1) / kW;
  for (int j = 0; j < nch; ++j) {
    const int i = j * kW + wave;
    const bool active = i < g.nx;

    mfma_acc acc[3][2];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[c][t] = mfma_acc{0.0, 0.0, 0.0, 0.0};
    double acc4[5] = {0.0, 0.0, 0.0, 0.0, 0.0};

    if (active) {
      const int start = pf.start, cnt = pf.cnt;
      if (lane < 54) bnb[wave][lane] = pf.b;
      double cur[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) cur[a] = pf.p[a];
      for (int base = 0; base < cnt; base += kCP) {
        const int mcnt = min(kCP, cnt - base);
        wave_sync();
        if (lane < ((mcnt + 3) & ~3)) {
          // a pass is padded to whole K = 4 steps with particles of zero weight at the origin: all their A_p*matB
          // and I_p rows are exact zeros
          const bool real = lane < mcnt;
          const double mpw_p = real ? mpw : 0.0;
          const double v[3] = {real ? cur[3] : 0.0, real ? cur[4] : 0.0, real ? cur[5] : 0.0};
          const W1 w(g, real ? cur[0] : 0.0, real ? cur[1] : 0.0, real ? cur[2] : 0.0);
          // interpolate_B_s1 (ecsim/simulation.cpp:64-118) out of the cell's LDS neighbourhood, same loop
          // and product order as the global-memory gather
          const int ox = w.is[0] - w.in[0] + 1, oy = w.is[1] - w.in[1] + 1, oz = w.is[2] - w.in[2] + 1;
          double Bp[3] = {0.0, 0.0, 0.0};
          const double* nb = bnb[wave];
#pragma unroll
          for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
              for (int ii = 0; ii < 2; ++ii) {
                Bp[0] += nb[((oz + k) * 3 + (oy + jj)) * 2 + ii] * (w.ws[2][k] * w.ws[1][jj] * w.wn[0][ii]);
                Bp[1] += nb[18 + ((oz + k) * 2 + jj) * 3 + (ox + ii)] * (w.ws[2][k] * w.wn[1][jj] * w.ws[0][ii]);
                Bp[2] += nb[36 + (k * 3 + (oy + jj)) * 3 + (ox + ii)] * (w.wn[2][k] * w.ws[1][jj] * w.ws[0][ii]);
              }
          // particles.cpp:107-115
          const double f = (0.5 * dt) * q / m;
          const double bx = Bp[0] * f, by = Bp[1] * f, bz = Bp[2] * f;
          const double b2 = bx * bx + by * by + bz * bz;
          const double vb = v[0] * bx + v[1] * by + v[2] * bz;
          const double cxv = +(v[1] * bz - v[2] * by), cyv = -(v[0] * bz - v[2] * bx), czv = +(v[0] * by - v[1] * bx);
          const double iq = q * mpw_p / (1. + b2);
          const double Ip[3] = {iq * (v[0] + cxv + vb * bx), iq * (v[1] + cyv + vb * by), iq * (v[2] + czv + vb * bz)};
          const double A_p = 0.5 * dt * dt * mpw_p * q * q / m / (1 + b2);
          const double AB[9] = {
            A_p * (1.0 + bx * bx), A_p * (+bz + bx * by), A_p * (-by + bx * bz),
            A_p * (-bz + by * bx), A_p * (1.0 + by * by), A_p * (+bx + by * bz),
            A_p * (+by + bz * bx), A_p * (-bx + bz * by), A_p * (1.0 + bz * bz)};
          // staggered-axis weights spread over the cell's 3 node slots (slot = node - cell + 1), :87-89
          double w3[3][3];
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            const int o = w.is[a] - w.in[a] + 1; // ox, oy, oz in {0, 1}
            w3[a][0] = o == 0 ? w.ws[a][0] : 0.0;
            w3[a][1] = o == 0 ? w.ws[a][1] : w.ws[a][0];
            w3[a][2] = o == 0 ? 0.0 : w.ws[a][1];
          }
          double* col = st + lane;
          // X rows: (k*2 + j)*3 + l ; s = wnz[k]*wny[j]*wsx[.]   (:138, :145)
#pragma unroll
          for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
              for (int l = 0; l < 3; ++l) col[((k * 2 + jj) * 3 + l) * kPadP] = w.wn[2][k] * w.wn[1][jj] * w3[0][l];
          // Y rows: 12 + (k*3 + l)*2 + i ; s = wnz[k]*wsy[.]*wnx[i]   (:139, :146)
#pragma unroll
          for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int l = 0; l < 3; ++l)
#pragma unroll
              for (int ii = 0; ii < 2; ++ii) col[(12 + (k * 3 + l) * 2 + ii) * kPadP] = w.wn[2][k] * w3[1][l] * w.wn[0][ii];
          // Z rows: 24 + (l*2 + j)*2 + i ; s = wsz[.]*wny[j]*wnx[i]   (:140, :147)
#pragma unroll
          for (int l = 0; l < 3; ++l)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
              for (int ii = 0; ii < 2; ++ii) col[(24 + (l * 2 + jj) * 2 + ii) * kPadP] = w3[2][l] * w.wn[1][jj] * w.wn[0][ii];
#pragma unroll
          for (int e = 0; e < 9; ++e) col[(36 + e) * kPadP] = AB[e];
#pragma unroll
          for (int e = 0; e < 3; ++e) col[(45 + e) * kPadP] = Ip[e];
          col[kRowOne * kPadP] = 1.0;
          col[kRowZero * kPadP] = 0.0;
        }
        // next pass of this cell: loads in flight during this pass's phase 2
        if (base + kCP + lane < cnt) {
          const long p = (long)start + base + kCP + lane;
#pragma unroll
          for (int a = 0; a < 3; ++a) { cur[a] = s.r[a][p]; cur[3 + a] = s.v[a][p]; }
        }
        wave_sync();

        {
          // K = 4 particles per step: 22 operands, 9 products, 6 + 5 MFMAs; the operands of step s+1 are requested
          // before the MFMAs of step s
          struct Operands { double a[3], b[2], m[6], a4[3], x4[5], y4[3]; };
          auto load = [&](Operands& o, int p0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) o.a[c] = a_ptr[c * 12 * kPadP + p0];
#pragma unroll
            for (int t = 0; t < 2; ++t) o.b[t] = b_ptr[t * 16 * kPadP + p0];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              o.m[c * 2 + 0] = m0_ptr[c * 3 * kPadP + p0];
              o.m[c * 2 + 1] = m1_ptr[c * 3 * kPadP + p0];
            }
#pragma unroll
            for (int n = 0; n < 3; ++n) { o.a4[n] = a4_ptr[n][p0]; o.y4[n] = y4_ptr[n][p0]; }
#pragma unroll
            for (int n = 0; n < 5; ++n) o.x4[n] = x4_ptr[n][p0];
          };
          auto gemm = [&](const Operands& o) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
              for (int t = 0; t < 2; ++t)
                acc[c][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a[c], o.b[t] * o.m[c * 2 + t], acc[c][t], 0, 0, 0);
            acc4[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(o.a4[0], o.x4[0] * o.y4[0], acc4[0], 0, 0, 0);
            acc4[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(o.a4[1], o.x4[1] * o.y4[1], acc4[1], 0, 0, 0);
            acc4[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(o.a4[0], o.x4[2], acc4[2], 0, 0, 0);
            acc4[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(o.a4[1], o.x4[3], acc4[3], 0, 0, 0);
            acc4[4] = __builtin_amdgcn_mfma_f64_4x4x4f64(o.a4[2], o.x4[4] * o.y4[2], acc4[4], 0, 0, 0);
          };
          const int nks = (mcnt + 3) >> 2;
          Operands A, Bo;
          load(A, 0);
          // fully unrolled: every operand address is a base register plus an immediate.  Requests run one step
          // ahead unconditionally (a step past the pass reads stale columns of the stage that nothing consumes).
#pragma unroll
          for (int ks = 0; ks < kCP / 4; ks += 2) {
            if (ks >= nks) break;
            if (ks + 1 < kCP / 4) load(Bo, 4 * (ks + 1));
            gemm(A);
            if (ks + 1 >= nks) break;
            if (ks + 2 < kCP / 4) load(A, 4 * (ks + 2));
            gemm(Bo);
          }
        }
      }
    }

    // next chunk's cell: particle data and B neighbourhood travel while this chunk is merged and flushed
    prefetch_cell(i + kW, pf);

    // ---- the finished columns of this chunk leave by read-modify-write: request their current values NOW, so
    // that the HBM latency runs under the merge below (addresses depend only on the chunk, not on the data)
    const int ndone = min(kW, g.nx - j * kW);
    // matL lines: the kW finished columns are the x-block j of the row, 32 contiguous, aligned bytes;
    // currI lines: kW consecutive doubles, 16-byte aligned when nx is even
    const bool vecL = ndone == kW, vecI = vecL && (g.nx & 1) == 0;
    double old[kOwn][kW];
    double* ptr[kOwn];
    bool fst[kOwn];
#pragma unroll
    for (int mm = 0; mm < kOwn; ++mm) {
      const int line = threadIdx.x + mm * kThreads;
      const uintptr_t lb = lbase[mm];
      fst[mm] = lb & 1;
      const bool vec = line < kMatLines ? vecL : vecI;
      ptr[mm] = lb ? (double*)(lb & ~(uintptr_t)1) + (long)j * (line < kMatLines ? kLBlock : kW) : nullptr;
#pragma unroll
      for (int c = 0; c < kW; ++c) old[mm][c] = 0.0;
      if (ptr[mm] && !fst[mm]) {
        if (vec) {
#pragma unroll
          for (int c = 0; c < kW; c += 2) {
            const double2 v = *(const double2*)(ptr[mm] + c);
            old[mm][c] = v.x; old[mm][c + 1] = v.y;
          }
        }
        else {
#pragma unroll
          for (int c = 0; c < kW; ++c)
            if (c < ndone) old[mm][c] = ptr[mm][c];
        }
      }
    }

    // ---- merge the chunk's cell blocks in the window (aliased over the now dead stages).  A thread owns kOwn
    // lines of the window: it seeds them with the two columns it carried over, and after the merge it streams
    // the finished columns out and keeps the last two in registers.
    lds_barrier();
    double* win = sh; // [kLines][kSlots]
#pragma unroll
    for (int mm = 0; mm < kOwn; ++mm) {
      const int line = threadIdx.x + mm * kThreads;
      if (line < kLines) {
        double2* w = (double2*)(win + line * kSlots);
        w[0] = double2{carry[mm][0], carry[mm][1]};
#pragma unroll
        for (int c = 1; c < kSlots / 2; ++c) w[c] = double2{0.0, 0.0};
      }
    }
    lds_barrier();
    if (active) {
      // row node x of cell (unwrapped) u = i+1 with offset o is column u+o; window column 0 is kW*j
      auto add = [&](int e, double val) {
        const unsigned d = (edesc[e / 2] >> (16 * (e & 1))) & 0xffffu;
        if (d != 0xffffu) unsafeAtomicAdd(&win[(d >> 2) * kSlots + wave + (d & 3)], val);
      };
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 3; ++r) add((c * 2 + t) * 3 + r, acc[c][t][r]);
#pragma unroll
      for (int n = 0; n < 5; ++n) add(18 + n, acc4[n]);
    }
    lds_barrier();
    // ---- stream out the finished columns (plain RMW, kW consecutive doubles per line), keep 2 in registers
#pragma unroll
    for (int mm = 0; mm < kOwn; ++mm) {
      const int line = threadIdx.x + mm * kThreads;
      if (line < kLines) {
        double w[kSlots];
        const double2* wp = (const double2*)(win + line * kSlots);
#pragma unroll
        for (int c = 0; c < kSlots / 2; ++c) { const double2 v = wp[c]; w[2 * c] = v.x; w[2 * c + 1] = v.y; }
        const bool vec = line < kMatLines ? vecL : vecI;
        bool any = fst[mm];
#pragma unroll
        for (int c = 0; c < kW; ++c) any = any || (c < ndone && w[c] != 0.0);
        if (any) {
          if (vec) {
#pragma unroll
            for (int c = 0; c < kW; c += 2)
              *(double2*)(ptr[mm] + c) = double2{old[mm][c] + w[c], old[mm][c + 1] + w[c + 1]};
          }
          else {
#pragma unroll
            for (int c = 0; c < kW; ++c)
              if (c < ndone) ptr[mm][c] = old[mm][c] + w[c];
          }
        }
        if (ndone == kW) { carry[mm][0] = w[kW]; carry[mm][1] = w[kW + 1]; }
        else {
          carry[mm][0] = carry[mm][1] = 0.0;
#pragma unroll
          for (int c = 1; c < kW; ++c)
            if (c == ndone) { carry[mm][0] = w[c]; carry[mm][1] = w[c + 1]; }
        }
      }
    }
    lds_barrier();
  }

  // ---- the two columns still carried are x = nx, nx+1 = 0, 1 (periodic): columns this workgroup has
  // already written, so they are added with atomics (2 of nx columns)
#pragma unroll
  for (int mm = 0; mm < kOwn; ++mm) {
    const int line = threadIdx.x + mm * kThreads;
    if (line < kLines) {
      double* base = (double*)(lbase[mm] & ~(uintptr_t)1);
#pragma unroll
      for (int c = 0; c < 2; ++c)
        if (carry[mm][c] != 0.0) unsafeAtomicAdd(base + g.wx(c), carry[mm][c]);
    }
  }
}

}  // namespace

// Tables.  The (row, col) node pairs of the cell block are those of
// ecsim::Simulation::fill_matrix_indices (src/impls/ecsim/simulation.cpp:408-464).
//   etab[i36*36 + j36] = (line << 2) | (row x offset + 1), or -1 for a structural zero (|d| = 2, same comp.)
//   linetab[line]      = c | (row dy + 1) << 2 | (row dz + 1) << 4 | k << 6
int build_ltab(xpic_ctx* c)
{
  std::vector<int> etab(36 * 36, -1), linetab(kLines, 0);
  // line ids in (c1, row dy, row dz, k) order: consecutive threads of the flush then walk consecutive k of one row
  // block, i.e. consecutive 32-byte pieces of matL
  std::map<int, int> line_of;
  {
    std::vector<std::array<int, 5>> keys;
    for (int i = 0; i < 36; ++i)
      for (int j = 0; j < 36; ++j) {
        int c1 = i / 12, c2 = j / 12, o1[3], o2[3];
        block_node_offset(c1, i % 12, o1);
        block_node_offset(c2, j % 12, o2);
        int k = lencode(c1, c2, o2[0] - o1[0], o2[1] - o1[1], o2[2] - o1[2]);
        if (k < 0) continue;
        keys.push_back({c1, o1[1] + 1, o1[2] + 1, k, c1 | ((o1[1] + 1) << 2) | ((o1[2] + 1) << 4) | (k << 6)});
      }
    std::sort(keys.begin(), keys.end());
    for (auto& q : keys)
      if (!line_of.count(q[4])) {
        const int id = (int)line_of.size();
        XPIC_CHECK(id < kMatLines, "matL line table overflow");
        line_of.emplace(q[4], id);
        linetab[id] = q[4];
      }
  }
  for (int i = 0; i < 36; ++i)
    for (int j = 0; j < 36; ++j) {
      int c1 = i / 12, c2 = j / 12, o1[3], o2[3];
      block_node_offset(c1, i % 12, o1);
      block_node_offset(c2, j % 12, o2);
      int k = lencode(c1, c2, o2[0] - o1[0], o2[1] - o1[1], o2[2] - o1[2]);
      if (k < 0) continue;
      int key = c1 | ((o1[1] + 1) << 2) | ((o1[2] + 1) << 4) | (k << 6);
      auto it = line_of.find(key);
      XPIC_CHECK(it != line_of.end(), "matL line table is incomplete");
      etab[i * 36 + j] = (it->second << 2) | (o1[0] + 1);
    }
  XPIC_CHECK((int)line_of.size() == kMatLines, "unexpected number of matL lines per pencil");
  for (int cidx = 0; cidx < 3; ++cidx)
    for (int l = 0; l < 12; ++l) {
      int o[3];
      block_node_offset(cidx, l, o);
      int id = cidx == 0 ? o[2] * 2 + o[1] : (cidx == 1 ? 4 + o[2] * 3 + (o[1] + 1) : 10 + (o[2] + 1) * 2 + o[1]);
      linetab[kMatLines + id] = cidx | ((o[1] + 1) << 2) | ((o[2] + 1) << 4);
    }
  // co-writers of a matL line (c1, dy, dz, k): the pencils at offset (dy - dy', dz - dz') for every other line
  // (c1, dy', dz', k) of the table -- they add into the same matL row stream
  std::vector<int> cowr(kLines * 8, 0x7fffffff);
  for (int l = 0; l < kMatLines; ++l) {
    int n = 0;
    for (int l2 = 0; l2 < kMatLines; ++l2) {
      if (l2 == l) continue;
      const int a = linetab[l], b = linetab[l2];
      if ((a & 3) != (b & 3) || (a >> 6) != (b >> 6)) continue;
      const int oy = ((a >> 2) & 3) - ((b >> 2) & 3), oz = ((a >> 4) & 3) - ((b >> 4) & 3);
      XPIC_CHECK(n < 8, "too many co-writers of a matL line");
      cowr[l * 8 + n++] = (oy + 8) | ((oz + 8) << 8);
    }
  }
  const size_t total = etab.size() + linetab.size() + cowr.size();
  XPIC_HIP(hipMalloc(&c->ltab, sizeof(int) * total));
  XPIC_HIP(hipMemcpy(c->ltab, etab.data(), sizeof(int) * etab.size(), hipMemcpyHostToDevice));
  XPIC_HIP(hipMemcpy(c->ltab + etab.size(), linetab.data(), sizeof(int) * linetab.size(), hipMemcpyHostToDevice));
  XPIC_HIP(hipMemcpy(c->ltab + etab.size() + linetab.size(), cowr.data(), sizeof(int) * cowr.size(), hipMemcpyHostToDevice));
  return 0;
}

// Colour classes of one periodic axis of n indices.  Same-colour indices must be >= 3 apart (periodically).
// If 3, 4 or 5 divides n the colours are the residues mod that period (all classes equal: no thin launches);
// otherwise residues mod 3 over the first 3*floor(n/3) indices plus one class per trailing index.
static int colour_period(int n)
{
  for (int p = 3; p <= 5; ++p)
    if (n % p == 0) return p;
  return 3;
}

static void colour_class(int n, int period, int colour, int* first, int* step, int* count)
{
  const int body = n - n % period;
  if (colour < period) { *first = colour; *step = period; *count = body / period; }
  else { *first = body + (colour - period); *step = 1; *count = 1; }
}

int ecsim_fill_sort(xpic_ctx* c, Sort& s, const double* B, double* currI_sort, double* matL, bool first_sort)
{
  if (s.n == 0) return 0;
  const GridDev& g = c->g;
  // y is periodic inside the slab; z is periodic only when the slab is the whole box: with z-neighbours the rows
  // below plane 0 / above plane nzl-1 are ghost rows of this rank alone, so plain residues mod 3 suffice
  const int per_y = colour_period(g.ny), per_z = g.G == 0 ? colour_period(g.nzl) : 3;
  const int ncol_y = per_y + g.ny % per_y, ncol_z = g.G == 0 ? per_z + g.nzl % per_z : 3;
  for (int b = 0; b < ncol_z; ++b)
    for (int a = 0; a < ncol_y; ++a) {
      int cy0, cys, ncy, cz0, czs, ncz;
      colour_class(g.ny, per_y, a, &cy0, &cys, &ncy);
      if (g.G == 0) colour_class(g.nzl, per_z, b, &cz0, &czs, &ncz);
      else { cz0 = b; czs = 3; ncz = (g.nzl - b + 2) / 3; }
      if (ncy == 0 || ncz == 0) continue;
      Timed t(c, "fill_current"); // one entry per colour launch: the average is the kernel's own launch duration
      hipLaunchKernelGGL(k_ecsim_fill, dim3((unsigned)(ncy * ncz)), dim3(kThreads), 0, c->stream, g, s.d, B,
        currI_sort, matL, c->ltab, c->ltab + 36 * 36, c->ltab + 36 * 36 + kLines, s.par.q, s.par.m,
        s.par.n / (double)s.par.Np, cy0, cys, ncy, cz0, czs, b * ncol_y + a, ncol_y, per_y, per_z, first_sort ? 1 : 0);
    }
  XPIC_HIP(hipGetLastError());
  return 0;
}

}  // namespace xpic
