// particles.hip -- particle-side kernels: SoA storage, move + periodic wrap + cell binning, the
// out-of-place counting sort that replaces the reference's per-cell std::list splicing, the CIC
// gather + Boris velocity update, reductions over particles, the synthetic loader.
#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "device_common.h"

#ifndef XPIC_BUCKET_CAP
#define XPIC_BUCKET_CAP 128 // source indices a cell's bucket holds (deferred scatter); a fuller cell sends the step through k_index
#endif

namespace xpic {

int experiment_particles() { return XPIC_TU_EXPERIMENT; }

namespace {

constexpr int kBlock = 256;

inline unsigned pgrid(int64_t n, int per_thread = 1)
{
  int64_t b = (n + (int64_t)kBlock * per_thread - 1) / ((int64_t)kBlock * per_thread);
  if (b < 1) b = 1;
  return (unsigned)b;
}

// destination of a (moved, wrapped) position in the slab decomposition:
//   >= 0 local cell, -1 dropped (outside the global box), -2 / -3 owned by the lower / upper z-neighbour,
//   -4 further away than a neighbour (reported as an error)
template <bool P2 = false>
__device__ inline int dest_of(const GridDev& g, int rank, int nranks, double x, double y, double z)
{
  double pn[3];
  scaled_position<P2>(g, x, y, z, pn);
  const int cx = (int)floor(pn[0]), cy = (int)floor(pn[1]), czg = (int)floor(pn[2]);
  if (cx < 0 || cx >= g.nx || cy < 0 || cy >= g.ny || czg < 0 || czg >= g.nzg) return -1;
  const int dest = czg / g.nzl;
  if (dest == rank) return ((czg - g.z0) * g.ny + cy) * g.nx + cx;
  if (dest == (rank + 1) % nranks) return -3;
  if (dest == (rank - 1 + nranks) % nranks) return -2;
  return -4;
}

struct Migr {
  int rank, nranks;
  double* send[2];   // outgoing Point records for the lower / upper neighbour
  int* sendcount;    // [0] down, [1] up, [2] error flags
  int send_cap;
};

// New cell of a (moved, wrapped) particle and its arrival rank in that cell; with MIG also the send side of
// update_cells_mpi.  Every live lane of the wave calls it together (ballots).
// src: the cell the particle sits in now, if the caller knows it (-1 otherwise).
template <bool MIG, bool P2 = false>
__device__ __forceinline__ void bin_particle(const GridDev& g, const SortDev& s, int64_t p, double x, double y, double z,
  double vx, double vy, double vz, const Migr& mg, int src = -1)
{
  int c = MIG ? dest_of<P2>(g, mg.rank, mg.nranks, x, y, z) : cell_of<P2>(g, x, y, z);
  if (MIG && c <= -2) {
    if (c == -4) { atomicOr(&mg.sendcount[2], 1); c = -1; }
    else {
      const int dir = c == -2 ? 0 : 1;
      const int idx = atomicAdd(&mg.sendcount[dir], 1);
      if (idx < mg.send_cap) {
        double* o = mg.send[dir] + 6L * idx;
        o[0] = x; o[1] = y; o[2] = z; o[3] = vx; o[4] = vy; o[5] = vz;
      }
      else atomicOr(&mg.sendcount[2], 2);
      c = -1;
    }
  }
  if (s.cell) s.cell[p] = c; // (null: the buckets alone carry this binning, see ecsim_second_push)
  // Arrival rank inside the new cell.  The input is (nearly) cell-sorted, so a wave sees a handful of distinct
  // cells: the lanes are grouped by cell with ballots, then ONE atomic instruction carries the returning add of
  // every group's first lane (one memory round trip per wave, not one per distinct cell).
  const int lane = threadIdx.x & 63;
  // With the source cell known, only the particles that stay are grouped (a wave holds two or three source cells); the
  // few that change cell (~7 %, up to 26 different neighbours: most of the loop's trips) take a returning atomic each,
  // one instruction for all of them, in flight while the loop runs.
  const bool solo = src >= 0 && c >= 0 && c != src;
  int solo_rank = 0;
  if (solo) solo_rank = atomicAdd(&s.cell_count[c], 1);
  int my_leader = lane, rank_in = 0, gsize = 0;
  unsigned long long todo = __ballot(c >= 0 && !solo);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int lc = __shfl(c, leader, 64);
    const unsigned long long same = __ballot(c == lc) & todo;
    if (c == lc) {
      my_leader = leader;
      rank_in = __popcll(same & ((1ull << lane) - 1ull));
      gsize = __popcll(same);
    }
    todo &= ~same;
  }
  int base = 0;
  if (c >= 0 && !solo && lane == my_leader) base = atomicAdd(&s.cell_count[c], gsize);
  base = __shfl(base, my_leader, 64);
  const int rank = c >= 0 ? (solo ? solo_rank : base + rank_in) : 0;
  if (s.cell) s.rank[p] = rank;
  // the deferred scatter's index, written on the spot: slot `rank` of the cell's bucket (no second pass over the keys; a
  // cell with more arrivals than the bucket holds raises the flag and the step falls back to k_index)
  if (s.bucket_cap > 0 && c >= 0) {
    if (rank < s.bucket_cap) s.bucket[(long)c * s.bucket_cap + rank] = (int)p;
    else atomicOr(s.bucket + s.ncell * s.bucket_cap, 1);
  }
}

// pass 1 of update_cells: (BorisPush::update_r) + correct_coordinates + new cell + arrival rank in it;
// with MIG also the send side of update_cells_mpi (src/interfaces/particles.cpp:118-181)
template <bool MOVE, bool WRAP, bool MIG>
__global__ void __launch_bounds__(kBlock) k_move_bin(GridDev g, SortDev s, int64_t n, double step, Migr mg)
{
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= n) return; // whole trailing lanes drop out together: ballots below only see live lanes
  double x = s.r[0][p], y = s.r[1][p], z = s.r[2][p];
  double vx = 0, vy = 0, vz = 0;
  if (MOVE || MIG) { vx = s.v[0][p]; vy = s.v[1][p]; vz = s.v[2][p]; }
  if (MOVE) {
    x += vx * step;
    y += vy * step;
    z += vz * step;
  }
  if (WRAP) {
    x = bound_periodic(x, g.Lx);
    y = bound_periodic(y, g.Ly);
    z = bound_periodic(z, g.Lz);
  }
  bin_particle<MIG>(g, s, p, x, y, z, vx, vy, vz, mg);
}

// receive side of update_cells_mpi (:226-241): bin what the neighbours sent
__global__ void __launch_bounds__(kBlock) k_bin_incoming(GridDev g, SortDev s, const double* inc, int n, int* inc_cell,
  int* inc_rank, int* err)
{
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int c = cell_of(g, inc[6L * i], inc[6L * i + 1], inc[6L * i + 2]);
  inc_cell[i] = c;
  if (c < 0) { atomicOr(err, 4); return; }
  const int rank = inc_rank[i] = atomicAdd(&s.cell_count[c], 1);
  // (a binning that fills the cells' buckets: record i of the receive buffer is the source index -1 - i, as k_index_incoming
  // writes it into the index)
  if (s.bucket_cap > 0) {
    if (rank < s.bucket_cap) s.bucket[(long)c * s.bucket_cap + rank] = -1 - i;
    else atomicOr(s.bucket + s.ncell * s.bucket_cap, 1);
  }
}

__global__ void __launch_bounds__(kBlock) k_scatter_incoming(SortDev s, const double* inc, int n, const int* inc_cell,
  const int* inc_rank)
{
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int c = inc_cell[i];
  if (c < 0) return;
  const int64_t d = (int64_t)s.cell_start[c] + inc_rank[i];
  for (int a = 0; a < 3; ++a) {
    s.r2[a][d] = inc[6L * i + a];
    s.v2[a][d] = inc[6L * i + 3 + a];
  }
}

// pass 2: recompute the moved + wrapped position (bitwise the same arithmetic) and scatter the record
template <bool MOVE, bool WRAP>
__global__ void __launch_bounds__(kBlock) k_scatter(GridDev g, SortDev s, int64_t n, double step)
{
  // XCD r (workgroups r, r + 8, ...) sweeps its own contiguous eighth of the particles: the six 8-byte records of a
  // particle that changes cell and the stayers of the cell it arrives in (a neighbour in x or y: close by in the
  // array) then pass through the same L2 and leave it as full lines (28.5 -> 25 ms at 256^3 x 64)
  const int64_t nblk = (n + kBlock - 1) / kBlock, chunk = (nblk + 7) / 8;
  const int64_t blk = (int64_t)(blockIdx.x % 8) * chunk + blockIdx.x / 8;
  if (blk >= nblk) return;
  const int64_t p = blk * kBlock + threadIdx.x;
  if (p >= n) return;
  // everything that does not depend on another load is requested together: the key, the arrival rank and the record
  // (the compiler otherwise serialises key -> record -> cell_start -> rank: four dependent round trips per thread)
  const int c = s.cell[p];
  const int rk = s.rank[p];
  double x = s.r[0][p], y = s.r[1][p], z = s.r[2][p];
  const double vx = s.v[0][p], vy = s.v[1][p], vz = s.v[2][p];
  if (c < 0) return;
  const int64_t d = (int64_t)s.cell_start[c] + rk;
  if (MOVE) {
    x += vx * step;
    y += vy * step;
    z += vz * step;
  }
  if (WRAP) {
    x = bound_periodic(x, g.Lx);
    y = bound_periodic(y, g.Ly);
    z = bound_periodic(z, g.Lz);
  }
  s.r2[0][d] = x; s.r2[1][d] = y; s.r2[2][d] = z;
  s.v2[0][d] = vx; s.v2[1][d] = vy; s.v2[2][d] = vz;
}

// deferred scatter: instead of moving the records, note for every slot of the new order where its record sits in the old
__global__ void __launch_bounds__(kBlock) k_index(SortDev s, int64_t n)
{
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= n) return;
  const int c = s.cell[p];
  if (c < 0) return;
  s.src[(int64_t)s.cell_start[c] + s.rank[p]] = (int)p;
}

// largest population of an x-pencil (the gathering assembly addresses the sorted copy by 32-bit byte offsets from the pencil's
// first slot: ecsim.hip); only launched for sorts of 2^29 particles or more
__global__ void __launch_bounds__(kBlock) k_max_pencil(const int* __restrict__ cell_start, int nx, long npencils, int limit, int* out)
{
  const long i = (long)blockIdx.x * kBlock + threadIdx.x;
  if (i >= npencils) return;
  const int pop = cell_start[(i + 1) * nx] - cell_start[i * nx];
  if (pop >= limit) atomicMax(out, pop);
}

// occupancy statistics of a sort (xpic_sort_occupancy): what the per-cell passes, the buckets and the per-pencil workgroups
// of the particle kernels will meet.  One thread per x-pencil.
__global__ void __launch_bounds__(kBlock) k_occupancy(const int* __restrict__ cell_start, int nx, long npencils, int bucket_cap,
  unsigned long long* out)
{
  const long i = (long)blockIdx.x * kBlock + threadIdx.x;
  if (i >= npencils) return;
  const int* cs = cell_start + i * nx;
  int mx = 0, over64 = 0, over128 = 0, overb = 0, empty = 0;
  for (int x = 0; x < nx; ++x) {
    const int c = cs[x + 1] - cs[x];
    mx = c > mx ? c : mx;
    over64 += c > 64; over128 += c > 128; overb += bucket_cap > 0 && c > bucket_cap; empty += c == 0;
  }
  const unsigned long long pop = (unsigned long long)(cs[nx] - cs[0]);
  atomicMax(&out[0], (unsigned long long)mx);
  if (over64) atomicAdd(&out[1], (unsigned long long)over64);
  if (over128) atomicAdd(&out[2], (unsigned long long)over128);
  if (overb) atomicAdd(&out[3], (unsigned long long)overb);
  atomicMax(&out[4], pop);
  atomicMin(&out[5], pop);
  if (empty) atomicAdd(&out[6], (unsigned long long)empty);
}

// ... and the records that arrived from the neighbouring slabs stay in the receive buffer: slot <- -1 - (record number)
__global__ void __launch_bounds__(kBlock) k_index_incoming(SortDev s, int n, const int* __restrict__ inc_cell, const int* __restrict__ inc_rank)
{
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int c = inc_cell[i];
  if (c < 0) return;
  s.src[(int64_t)s.cell_start[c] + inc_rank[i]] = -1 - i;
}

// ---- exclusive scan of the per-cell counts (3 small kernels; N ints, negligible next to particles) ----
constexpr int kScanItems = 8; // per thread
constexpr int kScanTile = kBlock * kScanItems;

__device__ inline int block_excl_scan(int v, int* total)
{
  __shared__ int wsum[kBlock / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int base = 0, tot = 0;
  for (int w = 0; w < kBlock / 64; ++w) {
    if (w < wave) base += wsum[w];
    tot += wsum[w];
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__global__ void __launch_bounds__(kBlock) k_scan_tiles(const int* in, long n, int* tile_sums)
{
  const long base = (long)blockIdx.x * kScanTile + (long)threadIdx.x * kScanItems;
  int v = 0;
  for (int i = 0; i < kScanItems; ++i)
    if (base + i < n) v += in[base + i];
  int tot;
  block_excl_scan(v, &tot);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(kBlock) k_scan_sums(int* tile_sums, int ntiles, int* total_out)
{
  // single workgroup, serial over chunks of kBlock tiles
  __shared__ int carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int b = 0; b < ntiles; b += kBlock) {
    int i = b + threadIdx.x;
    int v = i < ntiles ? tile_sums[i] : 0;
    int tot;
    int ex = block_excl_scan(v, &tot);
    int carry = carry_s;
    if (i < ntiles) tile_sums[i] = carry + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total_out = carry_s;
}

__global__ void __launch_bounds__(kBlock) k_scan_apply(const int* in, long n, const int* tile_sums, int* out)
{
  const long base = (long)blockIdx.x * kScanTile + (long)threadIdx.x * kScanItems;
  int vals[kScanItems];
  int v = 0;
  for (int i = 0; i < kScanItems; ++i) {
    vals[i] = (base + i < n) ? in[base + i] : 0;
    v += vals[i];
  }
  int tot;
  int ex = block_excl_scan(v, &tot) + tile_sums[blockIdx.x];
  for (int i = 0; i < kScanItems; ++i) {
    if (base + i < n) out[base + i] = ex;
    ex += vals[i];
  }
}

// ecsim::Particles::second_push (src/impls/ecsim/particles.cpp:175-192).
// A workgroup owns one x-pencil of cells and marches along it in rounds of 256 CONSECUTIVE PARTICLES, one per thread,
// whatever cells they belong to (the pencil's particles are contiguous): every wave is full, where one wave per cell
// ran its second pass of a 64-particle Poisson cell with a handful of live lanes (1.47 passes per cell on average; the
// instruction stream of a pass costs the same for 1 or 64 lanes, and the gather + Boris arithmetic is 10 of this
// kernel's 27 ms at 256^3 x 64).  A round's particles lie in at most kSPSeg consecutive cells (a round ends early
// where they do not); all they gather from (interpolate_E_s1 / interpolate_B_s1, ecsim/simulation.cpp:8-118) is the
// strip of E, B nodes around those cells: 349 values, fetched once per round into LDS.  The next round's strip and
// particles are requested while the current round is pushed; the strip is double-buffered: one barrier per round.
constexpr int kSPW = 4;      // waves per workgroup
constexpr int kSPRound = kSPW * 64;
constexpr int kSPSeg = 8;    // cells a round can span
// strip layout [z row][y row][x]: Ex (x from c-1: S+2 nodes, y: 2, z: 2), Ey (x from c: S+1, y from cy-1: 3, z: 2),
// Ez (S+1, 2, z from cz-1: 3), Bx (S+1, 3, 3), By (S+2, 2, 3), Bz (S+2, 3, 2)
constexpr int kSPXs = kSPSeg + 2, kSPXn = kSPSeg + 1;
constexpr int kSPoEx = 0, kSPoEy = kSPoEx + kSPXs * 4, kSPoEz = kSPoEy + kSPXn * 6, kSPoBx = kSPoEz + kSPXn * 6;
constexpr int kSPoBy = kSPoBx + kSPXn * 9, kSPoBz = kSPoBy + kSPXs * 6, kSPStrip = kSPoBz + kSPXs * 6;
constexpr int kSPPer = (kSPStrip + kSPRound - 1) / kSPRound; // strip values per thread

using UniformIntsP = const __attribute__((address_space(4))) int*; // wave-uniform reads of cell_start: scalar loads

// GA: the sort's re-binning deferred its scatter and the assembly only read through the index (xpic_set_fused_rebin 2):
// slot p of the new order finds its record at src[p] of the old; it is moved by `step` and wrapped here (k_scatter's
// arithmetic), pushed, and written -- position and new velocity -- to r2 / v2 [p]: this kernel, which is bound by HBM
// anyway, carries the sorted copy's stores (in the assembly they cost 9 ms: its store instructions, not their bytes).
template <bool PREBIN, bool MIG, bool P2, bool GA>
__global__ void __launch_bounds__(kSPRound) k_second_push(GridDev g, SortDev s, const double* __restrict__ E,
  const double* __restrict__ B, double qm, long npencil, long chunk, Migr mg, double step)
{
  // workgroup -> pencil; XCD r sweeps its own contiguous range of pencils (see k_matA)
  const long q = (long)(blockIdx.x % 8) * chunk + blockIdx.x / 8;
  if (blockIdx.x / 8 >= chunk || q >= npencil) return;
  const int cy = (int)(q % g.ny), cz = (int)(q / g.ny);
  UniformIntsP cs = (UniformIntsP)(s.cell_start + q * g.nx);

  __shared__ double strip[2][kSPStrip];

  // this thread's strip entries e = thread + i * 256: the row of the field array the value comes from and its x
  // offset from the round's first cell
  const double* srow[kSPPer];
  int sdx[kSPPer];
#pragma unroll
  for (int i = 0; i < kSPPer; ++i) {
    const int e = threadIdx.x + i * kSPRound;
    int comp, l, X, Y, x0, y0, z0; // component (0..2 E, 3..5 B), local index, row length, rows per plane, origin
    if (e < kSPoEy) { comp = 0; l = e - kSPoEx; X = kSPXs; Y = 2; x0 = -1; y0 = 0; z0 = 0; }
    else if (e < kSPoEz) { comp = 1; l = e - kSPoEy; X = kSPXn; Y = 3; x0 = 0; y0 = -1; z0 = 0; }
    else if (e < kSPoBx) { comp = 2; l = e - kSPoEz; X = kSPXn; Y = 2; x0 = 0; y0 = 0; z0 = -1; }
    else if (e < kSPoBy) { comp = 3; l = e - kSPoBx; X = kSPXn; Y = 3; x0 = 0; y0 = -1; z0 = -1; }
    else if (e < kSPoBz) { comp = 4; l = e - kSPoBy; X = kSPXs; Y = 2; x0 = -1; y0 = 0; z0 = -1; }
    else { comp = 5; l = e - kSPoBz; X = kSPXs; Y = 3; x0 = -1; y0 = -1; z0 = 0; }
    const int ix = l % X, iy = (l / X) % Y, iz = l / (X * Y);
    sdx[i] = x0 + ix;
    srow[i] = e < kSPStrip
      ? (comp < 3 ? E : B) + (comp % 3) * g.cstride + g.node(0, g.wy(cy + y0 + iy), g.wz(cz + z0 + iz))
      : nullptr;
  }

  // a round: particles [P0, P1) in the cells c_lo .. c_lo + kSPSeg - 1; adv = cells the round finishes
  struct Round { int c_lo, P0, P1, adv; };
  auto compose = [&](int c_lo, int P0) {
    Round r;
    r.c_lo = c_lo; r.P0 = P0;
    int cv[kSPSeg + 1];
#pragma unroll
    for (int i = 0; i <= kSPSeg; ++i) cv[i] = cs[min(c_lo + i, g.nx)];
    r.P1 = min(P0 + kSPRound, cv[kSPSeg]);
    r.adv = 0;
#pragma unroll
    for (int i = 0; i < kSPSeg; ++i) r.adv += (c_lo + i < g.nx && cv[i + 1] <= r.P1) ? 1 : 0;
    return r;
  };

  struct Ahead { double r[3], v[3], f[kSPPer]; int src; };
  // GA: the index of the round AFTER next is requested with the records of the next one (no dependent round trip)
  auto request_src = [&](const Round& r, Ahead& pf) {
    const long p = (long)r.P0 + threadIdx.x;
    pf.src = 0;
    if (GA && r.c_lo < g.nx && p < r.P1) pf.src = s.src[p];
  };
  auto request = [&](const Round& r, Ahead& pf) {
#pragma unroll
    for (int i = 0; i < kSPPer; ++i) {
      pf.f[i] = 0.0;
      // (nodes beyond x = nx, the periodic image of node 0, are never gathered from)
      if (srow[i] && r.c_lo + sdx[i] <= g.nx) pf.f[i] = srow[i][g.wx(r.c_lo + sdx[i])];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) { pf.r[a] = 0.0; pf.v[a] = 0.0; }
    const long p = (long)r.P0 + threadIdx.x;
    if (p < r.P1) {
      const long sp = GA ? (long)pf.src : p;
#pragma unroll
      for (int a = 0; a < 3; ++a) { pf.r[a] = s.r[a][sp]; pf.v[a] = s.v[a][sp]; }
    }
  };

  Round nxt = compose(0, cs[0]);
  Ahead pf;
  request_src(nxt, pf);
  request(nxt, pf);
  Round nn = compose(nxt.c_lo + nxt.adv, nxt.P1);
  request_src(nn, pf);
  for (int rd = 0; nxt.c_lo < g.nx; ++rd) {
    const Round cur = nxt;
    double* st = strip[rd & 1];
#pragma unroll
    for (int i = 0; i < kSPPer; ++i) {
      const int e = threadIdx.x + i * kSPRound;
      if (e < kSPStrip) st[e] = pf.f[i];
    }
    double r[3] = {pf.r[0], pf.r[1], pf.r[2]};
    double v[3] = {pf.v[0], pf.v[1], pf.v[2]};
    if (GA) {
      r[0] += v[0] * step; r[1] += v[1] * step; r[2] += v[2] * step;
      r[0] = bound_periodic(r[0], g.Lx); r[1] = bound_periodic(r[1], g.Ly); r[2] = bound_periodic(r[2], g.Lz);
    }
    nxt = GA ? nn : compose(cur.c_lo + cur.adv, cur.P1);
    // raw barrier that only drains LDS traffic (the requests of the next round stay in flight).  One per round: the
    // strip written here was last read two rounds ago, before the previous round's barrier.
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    request(nxt, pf);
    if (GA) {
      nn = compose(nxt.c_lo + nxt.adv, nxt.P1);
      request_src(nn, pf);
    }

    const long p = (long)cur.P0 + threadIdx.x;
    if (p < cur.P1) {
      const W1T<P2> w(g, r[0], r[1], r[2]);
      const int ox = w.is[0] - w.in[0] + 1, oy = w.is[1] - w.in[1] + 1, oz = w.is[2] - w.in[2] + 1;
      const int dn = w.in[0] - cur.c_lo, ds = dn + ox; // x index of the node-centred / half-shifted lower node in the strip
      double Ep[3] = {0, 0, 0}, Bp[3] = {0, 0, 0};
      // loop and product order of ecsim/simulation.cpp:8-118
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            Ep[0] += st[kSPoEx + (k * 2 + j) * kSPXs + (ds + i)] * (w.wn[2][k] * w.wn[1][j] * w.ws[0][i]);
            Ep[1] += st[kSPoEy + (k * 3 + (oy + j)) * kSPXn + (dn + i)] * (w.wn[2][k] * w.ws[1][j] * w.wn[0][i]);
            Ep[2] += st[kSPoEz + ((oz + k) * 2 + j) * kSPXn + (dn + i)] * (w.ws[2][k] * w.wn[1][j] * w.wn[0][i]);
            Bp[0] += st[kSPoBx + ((oz + k) * 3 + (oy + j)) * kSPXn + (dn + i)] * (w.ws[2][k] * w.ws[1][j] * w.wn[0][i]);
            Bp[1] += st[kSPoBy + ((oz + k) * 2 + j) * kSPXs + (ds + i)] * (w.ws[2][k] * w.wn[1][j] * w.ws[0][i]);
            Bp[2] += st[kSPoBz + (k * 3 + (oy + j)) * kSPXs + (ds + i)] * (w.wn[2][k] * w.ws[1][j] * w.ws[0][i]);
          }
      update_vEB(g.dt, qm, Ep, Bp, v);
      if (GA) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { s.r2[a][p] = r[a]; s.v2[a][p] = v[a]; }
      }
      else { s.v[0][p] = v[0]; s.v[1][p] = v[1]; s.v[2][p] = v[2]; }
      if (PREBIN) {
        // the next step opens with first_push + update_cells of exactly this state: r + v dt, wrapped, binned.  Doing
        // the binning here (same expressions as k_move_bin<true, true, .>) saves that pass its 48 B per particle.
        const double x = bound_periodic(r[0] + v[0] * g.dt, g.Lx);
        const double y = bound_periodic(r[1] + v[1] * g.dt, g.Ly);
        const double z = bound_periodic(r[2] + v[2] * g.dt, g.Lz);
        bin_particle<MIG, P2>(g, s, p, x, y, z, v[0], v[1], v[2], mg, (int)(q * g.nx) + w.in[0]);
      }
    }
  }
}

__device__ inline double wave_sum(double v)
{
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// sums of vx, vy, vz, v^2 (Energy::calculate_kinetic, src/diagnostics/energy.cpp:61-108)
__global__ void __launch_bounds__(kBlock) k_kinetic(SortDev s, int64_t n, double* partial, int nblocks)
{
  double a[4] = {0, 0, 0, 0};
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += stride) {
    const double vx = s.v[0][p], vy = s.v[1][p], vz = s.v[2][p];
    a[0] += vx; a[1] += vy; a[2] += vz;
    a[3] += vx * vx + vy * vy + vz * vz;
  }
  __shared__ double sm[4][kBlock / 64];
  for (int j = 0; j < 4; ++j) {
    double v = wave_sum(a[j]);
    if ((threadIdx.x & 63) == 0) sm[j][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    double v = 0;
    for (int w = 0; w < kBlock / 64; ++w) v += sm[threadIdx.x][w];
    partial[(long)threadIdx.x * nblocks + blockIdx.x] = v;
  }
}

__global__ void __launch_bounds__(kBlock) k_sum_rows(const double* partial, int nblocks, double* out)
{
  double v = 0;
  for (int i = threadIdx.x; i < nblocks; i += kBlock) v += partial[(long)blockIdx.x * nblocks + i];
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

// BorisPush::update_r (src/algorithms/boris_push.cpp:19-22) over a whole sort
__global__ void __launch_bounds__(kBlock) k_move(SortDev s, int64_t n, double step)
{
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= n) return;
  s.r[0][p] += s.v[0][p] * step;
  s.r[1][p] += s.v[1][p] * step;
  s.r[2][p] += s.v[2][p] * step;
}

__global__ void __launch_bounds__(kBlock) k_scale_v(SortDev s, int64_t n, double lambda)
{
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= n) return;
  s.v[0][p] *= lambda; s.v[1][p] *= lambda; s.v[2][p] *= lambda;
}

// ParticlesChargeDensity::collect (src/diagnostics/charge_conservation.cpp:34-97): 3 x 3 x 3 nodes from
// ceil(r/dx - 1.5), weight spline2 x spline2 x spline2, value q * n/Np.  Diagnostic, off the hot path: plain atomics.
__device__ inline double spline2_d(double s)
{
  s = fabs(s);
  if (s <= 0.5) return (0.75 - s * s);
  if (0.5 < s && s < 1.5) return 0.5 * (1.5 - s) * (1.5 - s);
  return 0.0;
}

__global__ void __launch_bounds__(kBlock) k_charge_density(GridDev g, SortDev s, int64_t n, double qn, double* rho)
{
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= n) return;
  const double pr[3] = {s.r[0][p] / g.dx, s.r[1][p] / g.dy, s.r[2][p] / g.dz};
  int st[3];
  double w[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    st[a] = (int)ceil(pr[a] - 1.5);
#pragma unroll
    for (int t = 0; t < 3; ++t) w[a][t] = spline2_d(pr[a] - (double)(st[a] + t));
  }
  st[2] -= g.z0;
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        // cache[i] = sfunc(x) * sfunc(y) * sfunc(z); arr += q * cache * n_Np   (:57, :90)
        const int x = ((st[0] + i) % g.nx + g.nx) % g.nx, y = ((st[1] + j) % g.ny + g.ny) % g.ny;
        const double v = w[0][i] * w[1][j] * w[2][k];
        if (v != 0.0) unsafeAtomicAdd(&rho[g.node(x, y, g.wz(st[2] + k))], qn * v);
      }
}

// DistributionMoment::collect, moment "density" (src/diagnostics/distribution_moment.cpp:125-216): cell-centred,
// 2 x 2 x 2 cells from round(r/dx - 1), spline_of_1st_order, value n/Np
__global__ void __launch_bounds__(kBlock) k_moment_density(GridDev g, SortDev s, int64_t n, double n_Np, double* out)
{
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= n) return;
  const double pr[3] = {s.r[0][p] / g.dx, s.r[1][p] / g.dy, s.r[2][p] / g.dz};
  int st[3];
  double w[3][2];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    st[a] = (int)round(pr[a] - 1.0);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const double d = fabs(pr[a] - ((double)(st[a] + t) + 0.5));
      w[a][t] = d <= 1.0 ? 1.0 - d : 0.0;
    }
  }
  st[2] -= g.z0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int ix = i % 2, iy = (i / 2) % 2, iz = i / 4;
    const int x = ((st[0] + ix) % g.nx + g.nx) % g.nx, y = ((st[1] + iy) % g.ny + g.ny) % g.ny;
    const double c = w[0][ix] * w[1][iy] * w[2][iz];
    if (c != 0.0) unsafeAtomicAdd(&out[g.node(x, y, g.wz(st[2] + iz))], c * n_Np);
  }
}

// MomentumConservation::calculate (src/diagnostics/momentum_conservation.cpp:77-131): per particle the 2nd-order
// Shape::setup(point.r) (shape.cpp:31-41), sums of m v ns and q E[node] Es over its nodes.  Diagnostic, off the hot
// path: E straight from global memory; deterministic two-stage reduction (6 rows of block partials).
__global__ void __launch_bounds__(kBlock) k_momentum(GridDev g, SortDev s, int64_t n, const double* __restrict__ E,
  double m, double q, double* partial, int nblocks)
{
  double a[6] = {0, 0, 0, 0, 0, 0};
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  const double dd[3] = {g.dx, g.dy, g.dz};
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += stride) {
    const double r[3] = {s.r[0][p], s.r[1][p], s.r[2][p]};
    const double v[3] = {s.v[0][p], s.v[1][p], s.v[2][p]};
    int st[3], sz[3];
    double No[3][4], Sh[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double pr = r[c] / dd[c];
      st[c] = (int)round(pr - 1.5);              // Shape::make_start
      sz[c] = (int)floor(pr + 1.5) + 1 - st[c];  // Shape::make_end
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const double gx = (double)(st[c] + t);
        No[c][t] = spline2_d(pr - gx);
        Sh[c][t] = spline2_d(pr - (gx + 0.5));
      }
    }
    st[2] -= g.z0;
    for (int kz = 0; kz < sz[2]; ++kz)
      for (int jy = 0; jy < sz[1]; ++jy)
        for (int ix = 0; ix < sz[0]; ++ix) {
          const long node = g.nodew(st[0] + ix, st[1] + jy, st[2] + kz);
          const double ns = No[2][kz] * No[1][jy] * No[0][ix];
          a[0] += m * v[0] * ns;
          a[1] += m * v[1] * ns;
          a[2] += m * v[2] * ns;
          // Shape::electric (shape.h:54-61)
          a[3] += q * E[node] * (No[2][kz] * No[1][jy] * Sh[0][ix]);
          a[4] += q * E[g.cstride + node] * (No[2][kz] * Sh[1][jy] * No[0][ix]);
          a[5] += q * E[2 * g.cstride + node] * (Sh[2][kz] * No[1][jy] * No[0][ix]);
        }
  }
  __shared__ double sm[6][kBlock / 64];
  for (int j = 0; j < 6; ++j) {
    double v = wave_sum(a[j]);
    if ((threadIdx.x & 63) == 0) sm[j][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    double v = 0;
    for (int w = 0; w < kBlock / 64; ++w) v += sm[threadIdx.x][w];
    partial[(long)threadIdx.x * nblocks + blockIdx.x] = v;
  }
}

// AoS Point records (host staging buffer on device) -> SoA tail of the sort
__global__ void __launch_bounds__(kBlock) k_unpack(SortDev s, int64_t at, int64_t n, const double* pts6)
{
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  for (int c = 0; c < 3; ++c) {
    s.r[c][at + i] = pts6[6 * i + c];
    s.v[c][at + i] = pts6[6 * i + 3 + c];
  }
}

__global__ void __launch_bounds__(kBlock) k_pack(GridDev g, SortDev s, int64_t n, double* pts6, int* cell)
{
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  for (int c = 0; c < 3; ++c) {
    pts6[6 * i + c] = s.r[c][i];
    pts6[6 * i + 3 + c] = s.v[c][i];
  }
  cell[i] = cell_of(g, s.r[0][i], s.r[1][i], s.r[2][i]);
}

// ---- synthetic loader (bench / smoke only): counter-based splitmix64 stream per particle -----------
__device__ inline uint64_t splitmix(uint64_t& x)
{
  uint64_t z = (x += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ inline double u01(uint64_t& st) { return ((splitmix(st) >> 11) + 0.5) * (1.0 / 9007199254740992.0); }

// REGULAR: particle p sits in cell p / ppc (exactly ppc per cell, the generated order IS the sorted order);
// otherwise the position is uniform over the whole slab like CoordinateInBox + SetParticles
// (src/utils/particles_load.cpp:11-18, src/commands/set_particles.cpp:19-43): Poisson occupancy of the cells,
// binned afterwards by the ordinary counting sort
// PROFILE (xpic_load_params::profile): 2 = density falling linearly along x from ratio : 1 (x = 0) to 1 (x = Lx), sampled
// through the inverse of its distribution function; 3 = a fraction of the particles in a Gaussian clump at the centre of the
// slab (Box-Muller, folded back periodically), the rest uniform.  drift: MaxwellianMomentum's px, py, pz
// (src/utils/particles_load.cpp:57-76): added to the thermal momentum before `tov`.
struct LoadDev {
  double drift[3];
  double a, b; // profile 2: density ratio; profile 3: clump fraction, clump sigma in cells
};
template <bool REGULAR, int PROFILE = 0>
__global__ void __launch_bounds__(kBlock) k_synthetic(GridDev g, SortDev s, int64_t n, int ppc, double vth, uint64_t seed, LoadDev ld)
{
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= n) return;
  uint64_t st = seed * 0xD1342543DE82EF95ull + (uint64_t)p * 0x2545F4914F6CDD1Dull;
  // strictly inside (0, 1): a generated point is never on the upper face of the box
  const double fx = u01(st) * 0.999999 + 0.0000005, fy = u01(st) * 0.999999 + 0.0000005, fz = u01(st) * 0.999999 + 0.0000005;
  if (REGULAR) {
    const int64_t cell = p / ppc;
    const int cx = (int)(cell % g.nx), cy = (int)((cell / g.nx) % g.ny), cz = (int)(cell / g.plane);
    s.r[0][p] = (cx + fx) * g.dx;
    s.r[1][p] = (cy + fy) * g.dy;
    s.r[2][p] = (cz + g.z0 + fz) * g.dz;
  }
  else {
    double ux = fx, uy = fy, uz = fz;
    if (PROFILE == 2) {
      // n(u) ~ ratio - (ratio - 1) u on [0, 1):  F(u) = (ratio u - (ratio - 1) u^2 / 2) / ((ratio + 1) / 2)
      const double r = ld.a, F = fx * 0.5 * (r + 1.0);
      ux = r > 1.0 ? (r - sqrt(r * r - 2.0 * (r - 1.0) * F)) / (r - 1.0) : fx;
      ux = fmin(fmax(ux, 0.0000005), 0.9999995);
    }
    if (PROFILE == 3) {
      const double pick = u01(st);
      double gs[3];
      for (int c = 0; c < 3; ++c) {
        const double u1 = u01(st), u2 = u01(st);
        gs[c] = ld.b * sqrt(-2.0 * log(u1)) * sin(2.0 * M_PI * u2); // cells
      }
      if (pick < ld.a) {
        ux = 0.5 + gs[0] / g.nx; uy = 0.5 + gs[1] / g.ny; uz = 0.5 + gs[2] / g.nzl;
        ux -= floor(ux); uy -= floor(uy); uz -= floor(uz);
        ux = fmin(fmax(ux, 0.0000005), 0.9999995); uy = fmin(fmax(uy, 0.0000005), 0.9999995); uz = fmin(fmax(uz, 0.0000005), 0.9999995);
      }
    }
    s.r[0][p] = ux * g.nx * g.dx;
    s.r[1][p] = uy * g.ny * g.dy;
    s.r[2][p] = (g.z0 + uz * g.nzl) * g.dz;
  }
  double v[3];
  for (int c = 0; c < 3; ++c) {
    const double u1 = u01(st), u2 = u01(st);
    v[c] = ld.drift[c] + vth * sqrt(-2.0 * log(u1)) * sin(2.0 * M_PI * u2);
  }
  const double gam = sqrt(1.0 + v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  s.v[0][p] = v[0] / gam; s.v[1][p] = v[1] / gam; s.v[2][p] = v[2] / gam;
}

__global__ void __launch_bounds__(kBlock) k_fill_counts(int* count, int* start, long ncell, int ppc)
{
  const long i = (long)blockIdx.x * kBlock + threadIdx.x;
  if (i < ncell) count[i] = ppc;
  if (i <= ncell) start[i] = (int)(i * ppc);
}

int exclusive_scan(xpic_ctx* c, const int* in, long n, int* out, int* total_host)
{
  const int ntiles = (int)((n + kScanTile - 1) / kScanTile);
  if (c->scan_tmp_n < ntiles + 1) {
    if (c->scan_tmp) XPIC_HIP(hipFree(c->scan_tmp));
    XPIC_HIP(hipMalloc(&c->scan_tmp, sizeof(int) * (ntiles + 1)));
    c->scan_tmp_n = ntiles + 1;
  }
  hipLaunchKernelGGL(k_scan_tiles, dim3(ntiles), dim3(kBlock), 0, c->stream, in, n, c->scan_tmp);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kBlock), 0, c->stream, c->scan_tmp, ntiles, c->scan_tmp + ntiles);
  hipLaunchKernelGGL(k_scan_apply, dim3(ntiles), dim3(kBlock), 0, c->stream, in, n, c->scan_tmp, out);
  XPIC_HIP(hipGetLastError());
  // out[n] = total
  XPIC_HIP(hipMemcpyAsync(out + n, c->scan_tmp + ntiles, sizeof(int), hipMemcpyDeviceToDevice, c->stream));
  if (total_host) {
    XPIC_HIP(hipMemcpyAsync(total_host, c->scan_tmp + ntiles, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    XPIC_HIP(hipStreamSynchronize(c->stream));
  }
  return 0;
}

}  // namespace

int sort_alloc(xpic_ctx* c, Sort& s, int64_t cap)
{
  XPIC_CHECK(cap > 0 && cap < (int64_t)2147483000, "sort capacity must be in (0, 2^31)");
  s.cap = cap;
  for (int a = 0; a < 3; ++a) {
    XPIC_HIP(hipMalloc(&s.d.r[a], sizeof(double) * cap));
    XPIC_HIP(hipMalloc(&s.d.v[a], sizeof(double) * cap));
    XPIC_HIP(hipMalloc(&s.d.r2[a], sizeof(double) * cap));
    XPIC_HIP(hipMalloc(&s.d.v2[a], sizeof(double) * cap));
  }
  XPIC_HIP(hipMalloc(&s.d.cell, sizeof(int) * cap));
  XPIC_HIP(hipMalloc(&s.d.rank, sizeof(int) * cap));
  s.d.src = nullptr; s.d.inc = nullptr;
  s.d.bucket = nullptr; s.d.bucket_cap = 0; s.d.ncell = c->ncell;
  XPIC_HIP(hipMalloc(&s.d.src, sizeof(int) * (cap + 1))); // deferred scatter (into the assembly, or into the next Esirkepov push)
  if (c->scheme != XPIC_BASIC) {
    // buckets of XPIC_BUCKET_CAP source indices per cell, where they cost at most as much as the particles' keys
    // (twice the mean occupancy the capacity allows, in steps of 32, at most XPIC_BUCKET_CAP: a Poisson cell never gets there)
    long bcap = ((2 * cap / c->ncell + 32 + 31) / 32) * 32;
    if (bcap > XPIC_BUCKET_CAP) bcap = XPIC_BUCKET_CAP;
    if (bcap > 0) {
      XPIC_HIP(hipMalloc(&s.d.bucket, sizeof(int) * (c->ncell * bcap + 1)));
      XPIC_HIP(hipMemsetAsync(s.d.bucket + c->ncell * bcap, 0, sizeof(int), c->stream));
      s.d.bucket_cap = (int)bcap;
    }
  }
  XPIC_HIP(hipMalloc(&s.d.cell_count, sizeof(int) * (c->ncell + 1)));
  // + kCellStartPad: the pencil kernels read a fixed number of entries ahead of the cell they are at (never used past
  // the pencil's end, but the reads must land in the allocation)
  XPIC_HIP(hipMalloc(&s.d.cell_start, sizeof(int) * (c->ncell + 1 + kCellStartPad)));
  XPIC_HIP(hipMemsetAsync(s.d.cell_start, 0, sizeof(int) * (c->ncell + 1 + kCellStartPad), c->stream));
  XPIC_HIP(hipMemsetAsync(s.d.cell_count, 0, sizeof(int) * (c->ncell + 1), c->stream));
  if (c->g.G > 0) {
    int64_t mc = cap / 8;
    if (mc < 65536) mc = 65536;
    s.mig_cap = (int)mc;
    XPIC_HIP(hipMalloc(&s.mig_send[0], sizeof(double) * 6 * mc));
    XPIC_HIP(hipMalloc(&s.mig_send[1], sizeof(double) * 6 * mc));
    XPIC_HIP(hipMalloc(&s.mig_recv, sizeof(double) * 6 * mc));
    s.d.inc = s.mig_recv;
    XPIC_HIP(hipMalloc(&s.mig_cell, sizeof(int) * mc));
    XPIC_HIP(hipMalloc(&s.mig_rank, sizeof(int) * mc));
    XPIC_HIP(hipMalloc(&s.mig_count, sizeof(int) * 8));
  }
  s.n = 0;
  return 0;
}

void sort_free(Sort& s)
{
  for (int a = 0; a < 3; ++a) {
    (void)hipFree(s.d.r[a]); (void)hipFree(s.d.v[a]); (void)hipFree(s.d.r2[a]); (void)hipFree(s.d.v2[a]);
  }
  (void)hipFree(s.d.cell); (void)hipFree(s.d.rank); (void)hipFree(s.d.src); (void)hipFree(s.d.bucket); (void)hipFree(s.d.cell_count); (void)hipFree(s.d.cell_start);
  (void)hipFree(s.J); (void)hipFree(s.currI); (void)hipFree(s.currJe); (void)hipFree(s.rho);
  (void)hipFree(s.mig_send[0]); (void)hipFree(s.mig_send[1]); (void)hipFree(s.mig_recv);
  (void)hipFree(s.mig_cell); (void)hipFree(s.mig_rank); (void)hipFree(s.mig_count);
  s = Sort{};
}

// (optional) r += step*v, periodic wrap, re-bin, drop what left the box: the whole of
// first_push + update_cells_seq (src/impls/ecsim/particles.cpp:21-31, src/interfaces/particles.cpp:79-116);
// with nranks > 1 also update_cells_mpi (:118-248): leavers go to the z-neighbours, arrivals are binned in.
static Migr make_migr(xpic_ctx* c, Sort& s)
{
  Migr mg{};
  if (c->comm.kind != 0) {
    mg.rank = c->comm.rank; mg.nranks = c->comm.nranks;
    mg.send[0] = s.mig_send[0]; mg.send[1] = s.mig_send[1];
    mg.sendcount = s.mig_count; mg.send_cap = s.mig_cap;
  }
  return mg;
}

static int launch_scatter(xpic_ctx* c, Sort& s, int64_t n_old, double step, bool wrap)
{
  Timed t(c, "scatter");
  const bool move = step != 0.0;
  const unsigned nb = (unsigned)(8 * ((pgrid(n_old) + 7) / 8)); // 8 XCD runs of equal length
#define LAUNCH(M, W) hipLaunchKernelGGL((k_scatter<M, W>), dim3(nb), dim3(kBlock), 0, c->stream, c->g, s.d, n_old, step)
  if (move && wrap) LAUNCH(true, true);
  else if (move) LAUNCH(true, false);
  else if (wrap) LAUNCH(false, true);
  else LAUNCH(false, false);
#undef LAUNCH
  XPIC_HIP(hipGetLastError());
  return 0;
}

// The pre-binning of ecsim_second_push wrote buckets but no keys (cell[], rank[]: 8 B per particle that only the scatter and
// k_index read) and one of those two is needed after all: a cell overflowed its bucket, or the deferral is resolved by the
// plain scatter.  The records still lie un-moved in the old order, so the binning pass is simply run again (its counts equal
// the ones cell_start was scanned from; the ranks are another valid numbering of the same cells).
static int rebuild_keys(xpic_ctx* c, Sort& s, int64_t n_old, double step)
{
  Timed t(c, "move_bin");
  if (c->profiling) c->prof["rebuild_keys"].launches += 1;
  XPIC_HIP(hipMemsetAsync(s.d.cell_count, 0, sizeof(int) * (c->ncell + 1), c->stream));
  SortDev sd = s.d;
  sd.bucket_cap = 0;
  hipLaunchKernelGGL((k_move_bin<true, true, false>), dim3(pgrid(n_old)), dim3(kBlock), 0, c->stream, c->g, sd, n_old, step, Migr{});
  XPIC_HIP(hipGetLastError());
  s.keys_valid = true;
  return 0;
}

// a deferred re-binning whose assembly never came (or was not the gathering kind): do the scatter now
int sort_materialize(xpic_ctx* c, Sort& s)
{
  if (!s.deferred) return 0;
  s.deferred = false;
  if (s.def_n_old > 0) {
    if (!s.keys_valid) XPIC_CALL(rebuild_keys(c, s, s.def_n_old, s.def_step));
    XPIC_CALL(launch_scatter(c, s, s.def_n_old, s.def_step, s.def_wrap));
  }
  if (s.def_n_in > 0) {
    hipLaunchKernelGGL(k_scatter_incoming, dim3(pgrid(s.def_n_in)), dim3(kBlock), 0, c->stream, s.d, s.mig_recv, s.def_n_in, s.mig_cell, s.mig_rank);
    XPIC_HIP(hipGetLastError());
  }
  for (int a = 0; a < 3; ++a) {
    std::swap(s.d.r[a], s.d.r2[a]);
    std::swap(s.d.v[a], s.d.v2[a]);
  }
  return 0;
}

void sort_deferred_done(Sort& s)
{
  s.deferred = false;
  for (int a = 0; a < 3; ++a) {
    std::swap(s.d.r[a], s.d.r2[a]);
    std::swap(s.d.v[a], s.d.v2[a]);
  }
}

int sort_rebin(xpic_ctx* c, Sort& s, double step, bool wrap, int defer)
{
  XPIC_CALL(sort_materialize(c, s));
  const bool move = step != 0.0;
  const bool mig = c->comm.kind != 0;
  // keys, ranks, counts (and the migration send buffers) of exactly this move may already be there (ecsim_second_push)
  const bool prebinned = s.prebinned && move && wrap && step == s.prebinned_step && s.n == s.prebinned_n;
  s.prebinned = false;
  Migr mg = make_migr(c, s);
  if (!prebinned) {
    XPIC_HIP(hipMemsetAsync(s.d.cell_count, 0, sizeof(int) * (c->ncell + 1), c->stream));
    if (s.d.bucket_cap > 0) XPIC_HIP(hipMemsetAsync(s.d.bucket + c->ncell * s.d.bucket_cap, 0, sizeof(int), c->stream));
    if (mig) XPIC_HIP(hipMemsetAsync(s.mig_count, 0, sizeof(int) * 4, c->stream));
  }
  if (!prebinned) {
    s.keys_valid = true; // (k_move_bin below writes them; an un-consumed key-less pre-binning is forgotten)
    // (the buckets are filled only for a binning that a deferred scatter will read -- by the arrivals too, sort or no sort)
    s.bucket_written = defer == 1 && s.d.bucket_cap > 0 && c->fused_rebin == 1;
  }
  if (s.n > 0 && !prebinned) {
    Timed t(c, "move_bin");
    const unsigned nb = pgrid(s.n);
    SortDev sd = s.d;
    if (!s.bucket_written) sd.bucket_cap = 0;
#define LAUNCH(M, W, G) hipLaunchKernelGGL((k_move_bin<M, W, G>), dim3(nb), dim3(kBlock), 0, c->stream, c->g, sd, s.n, step, mg)
    if (mig) {
      if (move && wrap) LAUNCH(true, true, true);
      else if (move) LAUNCH(true, false, true);
      else if (wrap) LAUNCH(false, true, true);
      else LAUNCH(false, false, true);
    }
    else {
      if (move && wrap) LAUNCH(true, true, false);
      else if (move) LAUNCH(true, false, false);
      else if (wrap) LAUNCH(false, true, false);
      else LAUNCH(false, false, false);
    }
#undef LAUNCH
    XPIC_HIP(hipGetLastError());
  }
  // Error state is COLLECTIVE on slabs: a rank that returned early while its neighbours wait in the next exchange
  // would hang the job.  Every rank contributes its local flags to one small all-reduce and all of them leave with
  // the same return code, before the next collective is entered.
  auto agree = [&](const bool (&local)[4], const char* const (&what)[4]) -> int {
    double e[4];
    for (int i = 0; i < 4; ++i) e[i] = local[i] ? 1.0 : 0.0;
    if (mig) XPIC_CALL(comm_allreduce_sum_host(c, e, 4));
    for (int i = 0; i < 4; ++i)
      if (e[i] != 0.0) {
        set_error(std::string(what[i]) + (mig ? " (on " + std::to_string((int)e[i]) + " of the z-slabs)" : ""));
        return 2;
      }
    return 0;
  };
  int n_in = 0;
  if (mig) {
    Timed t(c, "migrate");
    // counts first (MPI_Isend/Irecv of o_num/i_num, particles.cpp:183-197), then the Point records (:199-208)
    int hc[4];
    XPIC_HIP(hipMemcpyAsync(hc, s.mig_count, sizeof(int) * 4, hipMemcpyDeviceToHost, c->stream));
    XPIC_HIP(hipStreamSynchronize(c->stream));
    int* cnt = s.mig_count + 4; // [4],[5] = my down/up counts (copy), [6],[7] = from up / from down
    XPIC_HIP(hipMemcpyAsync(cnt, s.mig_count, sizeof(int) * 2, hipMemcpyDeviceToDevice, c->stream));
    XPIC_CALL(comm_ring(c, cnt, sizeof(int), cnt + 1, sizeof(int), cnt + 2, sizeof(int), cnt + 3, sizeof(int)));
    int hin[2];
    XPIC_HIP(hipMemcpyAsync(hin, cnt + 2, sizeof(int) * 2, hipMemcpyDeviceToHost, c->stream));
    XPIC_HIP(hipStreamSynchronize(c->stream));
    n_in = hin[0] + hin[1];
    {
      const bool bad[4] = {(hc[2] & 1) != 0, (hc[2] & 2) != 0, (long)hin[0] + hin[1] > s.mig_cap, false};
      const char* const what[4] = {"a particle moved further than the neighbouring z-slab in one step",
        "particle migration buffer overflow", "particle migration receive buffer overflow", ""};
      XPIC_CALL(agree(bad, what));
    }
    XPIC_CALL(comm_ring(c, s.mig_send[0], sizeof(double) * 6 * hc[0], s.mig_send[1], sizeof(double) * 6 * hc[1], s.mig_recv,
      sizeof(double) * 6 * hin[0], s.mig_recv + 6L * hin[0], sizeof(double) * 6 * hin[1]));
    if (n_in > 0) {
      SortDev sdi = s.d; // (the arrivals join the buckets of a binning that filled them: this pass's, or the second push's)
      if (!(s.bucket_written && defer == 1 && c->fused_rebin == 1)) sdi.bucket_cap = 0;
      hipLaunchKernelGGL(k_bin_incoming, dim3(pgrid(n_in)), dim3(kBlock), 0, c->stream, c->g, sdi, s.mig_recv, n_in,
        s.mig_cell, s.mig_rank, s.mig_count + 2);
      XPIC_HIP(hipGetLastError());
    }
  }
  int total = 0;
  {
    Timed t(c, "scan");
    XPIC_CALL(exclusive_scan(c, s.d.cell_count, c->ncell, s.d.cell_start, &total));
  }
  {
    int flags = 0; // bit 4: a received particle does not lie in this slab (k_bin_incoming)
    if (mig && n_in > 0) {
      XPIC_HIP(hipMemcpyAsync(&flags, s.mig_count + 2, sizeof(int), hipMemcpyDeviceToHost, c->stream));
      XPIC_HIP(hipStreamSynchronize(c->stream));
    }
    const bool bad[4] = {total > s.cap, (flags & 4) != 0, false, false};
    const char* const what[4] = {"sort capacity exceeded by incoming particles",
      "a particle received from a neighbouring z-slab does not lie in this slab", "", ""};
    XPIC_CALL(agree(bad, what));
  }
  // Deferred (the ecsim step on a single slab): the records stay where they are; slot d of the new order learns its
  // source, and the mass-matrix assembly -- which reads every particle anyway -- moves, wraps and writes it (ecsim.hip).
  // On z-slabs too (the assembly only: defer == 1): what the neighbours sent stays in the receive buffer and gets the source
  // indices -1 - i, in the buckets (k_bin_incoming) or in the index (k_index_incoming).
  // What the deferral's decisions need from the device, fetched with ONE synchronisation: the bucket-overflow flag of the
  // binning, and -- for sorts of 2^29 particles or more only -- whether an x-pencil holds 2^29 or more: that is beyond the
  // 32-bit offsets of the gathering assembly, and such a sort is scattered here, by this rank alone (nothing collective
  // depends on who moves the records).
  int* hflag = (int*)(c->red_host + 61);
  int* hmax = (int*)(c->red_host + 62);
  *hflag = 0; *hmax = 0;
  if (defer && (s.n > 0 || n_in > 0) && s.d.src) {
    bool sync = false;
    if (defer == 1 && total >= c->pencil_limit) {
      int* dmax = (int*)(c->red_out + 120);
      XPIC_HIP(hipMemsetAsync(dmax, 0, sizeof(int), c->stream));
      const long npen = c->ncell / c->g.nx;
      hipLaunchKernelGGL(k_max_pencil, dim3(pgrid(npen)), dim3(kBlock), 0, c->stream, s.d.cell_start, c->g.nx, npen, c->pencil_limit, dmax);
      XPIC_HIP(hipMemcpyAsync(hmax, dmax, sizeof(int), hipMemcpyDeviceToHost, c->stream));
      sync = true;
    }
    if (s.d.bucket_cap > 0) {
      XPIC_HIP(hipMemcpyAsync(hflag, s.d.bucket + c->ncell * s.d.bucket_cap, sizeof(int), hipMemcpyDeviceToHost, c->stream));
      sync = true;
    }
    if (sync) XPIC_HIP(hipStreamSynchronize(c->stream));
    if (*hmax != 0) { defer = 0; s.bucket_off = true; } // (and this sort's pre-binnings write the keys from now on)
  }
  if (defer && (!mig || defer == 1) && (s.n > 0 || n_in > 0) && s.d.src) {
    // the binning filled the cells' buckets (bin_particle) unless a cell overflowed its bucket: then the index is built
    // from the keys (k_index: 12 B per particle)
    const bool use_bucket = s.d.bucket_cap > 0 && defer == 1 && *hflag == 0 && s.bucket_written && c->fused_rebin == 1; // (mode 2's second push reads the index k_index builds)
    if (!use_bucket && !s.keys_valid) {
      // (a cell took more arrivals than a bucket holds: this sort's pre-binnings write the keys from now on)
      s.bucket_off = true;
      XPIC_CALL(rebuild_keys(c, s, s.n, step));
    }
    if (!use_bucket) {
      Timed t(c, "index");
      if (s.n > 0) hipLaunchKernelGGL(k_index, dim3(pgrid(s.n)), dim3(kBlock), 0, c->stream, s.d, s.n);
      if (n_in > 0) hipLaunchKernelGGL(k_index_incoming, dim3(pgrid(n_in)), dim3(kBlock), 0, c->stream, s.d, n_in, s.mig_cell, s.mig_rank);
      XPIC_HIP(hipGetLastError());
    }
    s.deferred = true; s.def_step = step; s.def_wrap = wrap; s.def_n_old = s.n; s.def_bucket = use_bucket; s.def_n_in = n_in;
    s.n = total;
    return 0;
  }
  if (s.n > 0 && !s.keys_valid) XPIC_CALL(rebuild_keys(c, s, s.n, step)); // (a key-less pre-binning that is not deferred after all)
  if (s.n > 0) XPIC_CALL(launch_scatter(c, s, s.n, step, wrap));
  if (n_in > 0) {
    hipLaunchKernelGGL(k_scatter_incoming, dim3(pgrid(n_in)), dim3(kBlock), 0, c->stream, s.d, s.mig_recv, n_in, s.mig_cell, s.mig_rank);
    XPIC_HIP(hipGetLastError());
  }
  for (int a = 0; a < 3; ++a) {
    std::swap(s.d.r[a], s.d.r2[a]);
    std::swap(s.d.v[a], s.d.v2[a]);
  }
  s.n = total;
  return 0;
}

int sort_append_host(xpic_ctx* c, Sort& s, int64_t n, const double* pts6, int64_t* added)
{
  XPIC_CALL(sort_materialize(c, s)); // (a deferred re-binning whose assembly has not run)
  s.prebinned = false;

  XPIC_CHECK(s.n + n <= s.cap, "sort capacity exceeded in add_particles");
  const int64_t before = s.n;
  if (n > 0) {
    double* tmp = nullptr;
    XPIC_HIP(hipMalloc(&tmp, sizeof(double) * 6 * n));
    XPIC_HIP(hipMemcpyAsync(tmp, pts6, sizeof(double) * 6 * n, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_unpack, dim3(pgrid(n)), dim3(kBlock), 0, c->stream, s.d, s.n, n, tmp);
    XPIC_HIP(hipGetLastError());
    XPIC_HIP(hipStreamSynchronize(c->stream));
    XPIC_HIP(hipFree(tmp));
    s.n += n;
  }
  // add_particle never wraps: points outside the box are dropped (particles.cpp:55-56)
  XPIC_CALL(sort_rebin(c, s, 0.0, false));
  if (added) *added = s.n - before;
  return 0;
}

int sort_download(xpic_ctx* c, Sort& s, double* pts6, int32_t* cell_of_out)
{
  XPIC_CALL(sort_materialize(c, s)); // (a deferred re-binning whose assembly has not run)
  if (s.n == 0) return 0;
  double* tmp = nullptr;
  int* tc = nullptr;
  XPIC_HIP(hipMalloc(&tmp, sizeof(double) * 6 * s.n));
  XPIC_HIP(hipMalloc(&tc, sizeof(int) * s.n));
  hipLaunchKernelGGL(k_pack, dim3(pgrid(s.n)), dim3(kBlock), 0, c->stream, c->g, s.d, s.n, tmp, tc);
  XPIC_HIP(hipGetLastError());
  XPIC_HIP(hipMemcpyAsync(pts6, tmp, sizeof(double) * 6 * s.n, hipMemcpyDeviceToHost, c->stream));
  if (cell_of_out) XPIC_HIP(hipMemcpyAsync(cell_of_out, tc, sizeof(int) * s.n, hipMemcpyDeviceToHost, c->stream));
  XPIC_HIP(hipStreamSynchronize(c->stream));
  XPIC_HIP(hipFree(tmp));
  XPIC_HIP(hipFree(tc));
  return 0;
}

int sort_occupancy(xpic_ctx* c, Sort& s, int64_t* out8)
{
  XPIC_CALL(sort_materialize(c, s));
  unsigned long long* d = nullptr;
  unsigned long long h[8] = {0, 0, 0, 0, 0, ~0ull, 0, 0};
  XPIC_HIP(hipMalloc(&d, sizeof(h)));
  XPIC_HIP(hipMemcpyAsync(d, h, sizeof(h), hipMemcpyHostToDevice, c->stream));
  const long npen = c->ncell / c->g.nx;
  hipLaunchKernelGGL(k_occupancy, dim3(pgrid(npen)), dim3(kBlock), 0, c->stream, s.d.cell_start, c->g.nx, npen, s.d.bucket_cap, d);
  XPIC_HIP(hipGetLastError());
  XPIC_HIP(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  XPIC_HIP(hipStreamSynchronize(c->stream));
  XPIC_HIP(hipFree(d));
  for (int i = 0; i < 7; ++i) out8[i] = (int64_t)h[i];
  out8[7] = s.d.bucket_cap;
  return 0;
}

int sort_fill_synthetic(xpic_ctx* c, Sort& s, const xpic_load_params& lp)
{
  const int ppc = lp.ppc;
  const double vth = lp.vth;
  const uint64_t seed = lp.seed;
  const bool regular = lp.profile == XPIC_LOAD_REGULAR;
  LoadDev ld{{lp.drift[0], lp.drift[1], lp.drift[2]}, lp.profile_param[0], lp.profile_param[1]};
  XPIC_CHECK(lp.profile >= 0 && lp.profile <= XPIC_LOAD_BLOB, "unknown load profile");
  XPIC_CHECK(lp.profile != XPIC_LOAD_GRADIENT || lp.profile_param[0] >= 1.0, "gradient load: density ratio >= 1");
  XPIC_CHECK(lp.profile != XPIC_LOAD_BLOB || (lp.profile_param[0] >= 0.0 && lp.profile_param[0] <= 1.0 && lp.profile_param[1] > 0.0),
    "blob load: fraction in [0, 1], sigma > 0 cells");
  XPIC_CALL(sort_materialize(c, s)); // (a deferred re-binning whose assembly has not run)
  s.prebinned = false;
  const int64_t n = (int64_t)c->ncell * ppc;
  XPIC_CHECK(n <= s.cap, "sort capacity exceeded in fill_synthetic");
  if (regular) {
    hipLaunchKernelGGL((k_synthetic<true, 0>), dim3(pgrid(n)), dim3(kBlock), 0, c->stream, c->g, s.d, n, ppc, vth, seed, ld);
    hipLaunchKernelGGL(k_fill_counts, dim3((unsigned)((c->ncell + 1 + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream,
      s.d.cell_count, s.d.cell_start, (long)c->ncell, ppc);
    XPIC_HIP(hipGetLastError());
    s.n = n;
    return 0;
  }
  if (lp.profile == XPIC_LOAD_GRADIENT)
    hipLaunchKernelGGL((k_synthetic<false, 2>), dim3(pgrid(n)), dim3(kBlock), 0, c->stream, c->g, s.d, n, ppc, vth, seed, ld);
  else if (lp.profile == XPIC_LOAD_BLOB)
    hipLaunchKernelGGL((k_synthetic<false, 3>), dim3(pgrid(n)), dim3(kBlock), 0, c->stream, c->g, s.d, n, ppc, vth, seed, ld);
  else
    hipLaunchKernelGGL((k_synthetic<false, 0>), dim3(pgrid(n)), dim3(kBlock), 0, c->stream, c->g, s.d, n, ppc, vth, seed, ld);
  XPIC_HIP(hipGetLastError());
  s.n = n;
  // add_particle's binning (no wrap; nothing falls outside).  With z-neighbours this is collective like any re-bin;
  // every generated point lies inside this slab, so nothing migrates.
  return sort_rebin(c, s, 0.0, false);
}

// prebin: also bin the particles for the first_push + update_cells that opens the next ecsim step (sort_rebin with
// step = dt consumes it if nothing touched the species in between)
int ecsim_second_push(xpic_ctx* c, Sort& s, const double* E, const double* B, bool prebin)
{
  const bool mig = c->comm.kind != 0;
  // a deferred re-binning that the assembly only read through (fused_rebin 2) is resolved HERE: gather, move, push, write
  const bool ga = s.deferred && c->fused_rebin == 2 && !mig && s.def_wrap && s.n > 0;
  if (!ga) XPIC_CALL(sort_materialize(c, s)); // (a deferred re-binning whose assembly has not run)
  s.prebinned = false;
  if (s.n == 0) return 0;
  Timed t(c, "second_push");
  const long npencil = (long)c->g.ny * c->g.nzl; // one workgroup per x-pencil
  const long chunk = (npencil + 7) / 8;
  Migr mg = make_migr(c, s);
  if (prebin) {
    XPIC_HIP(hipMemsetAsync(s.d.cell_count, 0, sizeof(int) * (c->ncell + 1), c->stream));
    if (s.d.bucket_cap > 0) XPIC_HIP(hipMemsetAsync(s.d.bucket + c->ncell * s.d.bucket_cap, 0, sizeof(int), c->stream));
    if (mig) XPIC_HIP(hipMemsetAsync(s.mig_count, 0, sizeof(int) * 4, c->stream));
  }
  // (the pre-binning fills the buckets when the next step's re-binning will defer its scatter into the assembly)
  SortDev sd = s.d;
  const bool wbucket = prebin && s.d.bucket_cap > 0 && c->fused_rebin == 1 && c->scheme == XPIC_ECSIM;
  if (!wbucket) sd.bucket_cap = 0;
  // ... and then the buckets are ALL the next re-binning reads: the keys stay unwritten (8 of the 84 B per particle), unless
  // this sort has overflowed a bucket before or the assembly that follows is not the gathering kind
  const bool keyless = wbucket && !s.bucket_off && c->fill_kernel == 0;
  if (keyless) { sd.cell = nullptr; sd.rank = nullptr; }
#define LAUNCH(P, G, Q, A) hipLaunchKernelGGL((k_second_push<P, G, Q, A>), dim3((unsigned)(8 * chunk)), dim3(kSPRound), 0, c->stream, \
    c->g, sd, E, B, s.par.q / s.par.m, npencil, chunk, mg, s.def_step)
  if (ga) {
    if (c->g.pow2) { if (prebin) LAUNCH(true, false, true, true); else LAUNCH(false, false, true, true); }
    else { if (prebin) LAUNCH(true, false, false, true); else LAUNCH(false, false, false, true); }
  }
  else if (c->g.pow2) { // exact reciprocal spacings: no fp64 division per position (device_common.h: scaled_position)
    if (!prebin) LAUNCH(false, false, true, false);
    else if (mig) LAUNCH(true, true, true, false);
    else LAUNCH(true, false, true, false);
  }
  else {
    if (!prebin) LAUNCH(false, false, false, false);
    else if (mig) LAUNCH(true, true, false, false);
    else LAUNCH(true, false, false, false);
  }
#undef LAUNCH
  XPIC_HIP(hipGetLastError());
  if (ga) sort_deferred_done(s); // r2 / v2 hold the sorted, moved, pushed records: they become the sort
  if (prebin) {
    s.prebinned = true;
    s.prebinned_step = c->g.dt;
    s.prebinned_n = s.n;
    s.bucket_written = wbucket;
    s.keys_valid = !keyless;
  }
  return 0;
}

int charge_density(xpic_ctx* c, Sort& s, double* rho_vec)
{
  XPIC_CALL(sort_materialize(c, s)); // (a deferred re-binning whose assembly has not run)
  XPIC_HIP(hipMemsetAsync(rho_vec, 0, sizeof(double) * c->nvec, c->stream));
  if (s.n > 0) {
    hipLaunchKernelGGL(k_charge_density, dim3(pgrid(s.n)), dim3(kBlock), 0, c->stream, c->g, s.d, s.n,
      s.par.q * (s.par.n / s.par.Np), rho_vec);
    XPIC_HIP(hipGetLastError());
  }
  return halo_add(c, rho_vec, 3); // DMLocalToGlobal(ADD) :95
}

int moment_density(xpic_ctx* c, Sort& s, double* vec)
{
  XPIC_CALL(sort_materialize(c, s)); // (a deferred re-binning whose assembly has not run)
  XPIC_HIP(hipMemsetAsync(vec, 0, sizeof(double) * c->nvec, c->stream));
  if (s.n > 0) {
    hipLaunchKernelGGL(k_moment_density, dim3(pgrid(s.n)), dim3(kBlock), 0, c->stream, c->g, s.d, s.n,
      s.par.n / s.par.Np, vec);
    XPIC_HIP(hipGetLastError());
  }
  return halo_add(c, vec, 3);
}

int kinetic_sums_host(xpic_ctx* c, Sort& s, double* out5)
{
  XPIC_CALL(sort_materialize(c, s)); // (a deferred re-binning whose assembly has not run)
  for (int i = 0; i < 5; ++i) out5[i] = 0;
  out5[4] = (double)s.n;
  if (s.n == 0) return 0;
  int nblocks = (int)pgrid(s.n, 8);
  if (nblocks > kRedBlocks) nblocks = kRedBlocks;
  hipLaunchKernelGGL(k_kinetic, dim3(nblocks), dim3(kBlock), 0, c->stream, s.d, s.n, c->red_partial, nblocks);
  hipLaunchKernelGGL(k_sum_rows, dim3(4), dim3(kBlock), 0, c->stream, c->red_partial, nblocks, c->red_out);
  XPIC_HIP(hipGetLastError());
  XPIC_HIP(hipMemcpyAsync(c->red_host, c->red_out, sizeof(double) * 4, hipMemcpyDeviceToHost, c->stream));
  XPIC_HIP(hipStreamSynchronize(c->stream));
  for (int i = 0; i < 4; ++i) out5[i] = c->red_host[i];
  return 0;
}

// the MPI_Allreduce of Energy::calculate_kinetic / calculate_energy (energy.cpp:92-93, ecsimcorr/particles.cpp:148)
int kinetic_sums_global(xpic_ctx* c, Sort& s, double* out5)
{
  XPIC_CALL(kinetic_sums_host(c, s, out5));
  return comm_allreduce_sum_host(c, out5, 5);
}

// out6 = {Px, Py, Pz, QEx, QEy, QEz} of one sort, summed over the slabs (E must have its ghost planes filled)
int momentum_sums_global(xpic_ctx* c, Sort& s, const double* E, double* out6)
{
  XPIC_CALL(sort_materialize(c, s)); // (a deferred re-binning whose assembly has not run)
  for (int i = 0; i < 6; ++i) out6[i] = 0;
  if (s.n > 0) {
    int nblocks = (int)pgrid(s.n, 4);
    if (nblocks > kRedBlocks) nblocks = kRedBlocks;
    hipLaunchKernelGGL(k_momentum, dim3(nblocks), dim3(kBlock), 0, c->stream, c->g, s.d, s.n, E, s.par.m / s.par.Np,
      s.par.q / s.par.Np, c->red_partial, nblocks);
    hipLaunchKernelGGL(k_sum_rows, dim3(6), dim3(kBlock), 0, c->stream, c->red_partial, nblocks, c->red_out);
    XPIC_HIP(hipGetLastError());
    XPIC_HIP(hipMemcpyAsync(c->red_host, c->red_out, sizeof(double) * 6, hipMemcpyDeviceToHost, c->stream));
    XPIC_HIP(hipStreamSynchronize(c->stream));
    for (int i = 0; i < 6; ++i) out6[i] = c->red_host[i];
  }
  return comm_allreduce_sum_host(c, out6, 6);
}

int sort_move(xpic_ctx* c, Sort& s, double step)
{
  XPIC_CALL(sort_materialize(c, s)); // (a deferred re-binning whose assembly has not run)
  s.prebinned = false;

  if (s.n == 0) return 0;
  Timed t(c, "move");
  hipLaunchKernelGGL(k_move, dim3(pgrid(s.n)), dim3(kBlock), 0, c->stream, s.d, s.n, step);
  XPIC_HIP(hipGetLastError());
  return 0;
}

int scale_velocities(xpic_ctx* c, Sort& s, double lambda)
{
  XPIC_CALL(sort_materialize(c, s)); // (a deferred re-binning whose assembly has not run)
  s.prebinned = false;

  if (s.n == 0) return 0;
  hipLaunchKernelGGL(k_scale_v, dim3(pgrid(s.n)), dim3(kBlock), 0, c->stream, s.d, s.n, lambda);
  XPIC_HIP(hipGetLastError());
  return 0;
}

}  // namespace xpic
