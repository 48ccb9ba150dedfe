// ecsim.hip -- ecsim::Particles::fill_ecsim_current / decompose_ecsim_current
// (src/impls/ecsim/particles.cpp:33-173) for cell-sorted SoA particles.
//
// All particles of a cell touch the same 3 x 12 Yee nodes, so the cell's contribution to matL is a dense
// 36 x 36 block  M = sum_p A_p * (s_p s_p^T) o matB_p  (the reference's 1296-entry COO block, :145-163)
// and its contribution to currI is the 36-vector sum_p s_p o I_p.
//
// Work decomposition ("pencil march"):
//   * one workgroup (4 waves) owns one x-pencil of cells (fixed cy, cz) and marches along x in chunks of 4 cells,
//     one cell per wave; a cell is staged in passes of at most kCP particles;
//   * phase 1 (lane = particle): CIC weights, B gather out of the cell's 54-value LDS neighbourhood, b, I_p,
//     A_p*matB.  The particle's half-cell octant (ox, oy, oz) fixes WHICH 8 of the 12 nodes per component it
//     touches; its 24 non-zero weights + 9 + 3 values go to the wave's LDS stage, compacted by octant;
//   * phase 2 (matrix cores): inside an octant the particle block is the dense 24 x 24 the reference fills
//     (576 products, particles.cpp:149-166): per 4 particles nine v_mfma_f64_4x4x4_4b_f64 (36 blocks of 4 x 4) and
//     two more for currI.  An accumulator belongs to one (component pair, octant bits of that pair): 36 per lane;
//   * per chunk the 4 cell blocks are merged in an LDS window indexed [matL line][x] (a "line" is one
//     (row component, row y/z offset, k) coefficient stream of the index-free matL), which sums the duplicates of
//     x-neighbouring cells on chip; finished columns are streamed out with plain, aligned read-modify-write (first
//     touch: plain stores); two unfinished columns are carried to the next chunk in registers;
//   * pencils whose rows overlap (|dcy| <= 2 and |dcz| <= 2, periodically) never run in the same launch:
//     launches are coloured by (cy mod p, cz mod p) (+ remainder colours), so the RMW needs no atomics and
//     the result is bitwise reproducible.  MatSetValuesCOO's duplicate summation (simulation.cpp:366) thus
//     happens in LDS (x) and in launch order (y, z).
// History: v1 flushed every cell block with ~1200 scattered fp64 atomics (85 % of the kernel time at 256^3); v2-v8
// accumulated the padded dense 36 x 36 block of a cell (VALU rank-1 updates, then v_mfma_f64_16x16x4): 56 % of the
// matrix-pipe cycles multiplied structural zeros.
#include <algorithm>
#include <array>
#include <map>
#include <type_traits>
#include <utility>
#include <vector>

#include "ecsim_fill.h"

#ifdef FILL_STAMPS
// in-kernel section timers (experiment build only): s_memtime deltas of wave 0 summed per section over all workgroups
__device__ unsigned long long g_fill_stamps[16];
#define STAMP(k)                                                                       \
  do {                                                                                 \
    const unsigned long long now_ = __builtin_readcyclecounter();                      \
    stamp_acc_[k] += now_ - stamp_t_;                                                  \
    stamp_t_ = now_;                                                                   \
  } while (0)
extern "C" int xpic_debug_fill_stamps(double* out, int reset)
{
  unsigned long long h[16];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fill_stamps), sizeof(h)) != hipSuccess) return 1;
  for (int i = 0; i < 16; ++i) out[i] = (double)h[i];
  if (reset) {
    unsigned long long z[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_fill_stamps), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
// sub-section timers inside a section (they do not move the section chain's clock)
#define SUB_BEGIN() const unsigned long long sub_t_ = __builtin_readcyclecounter()
#define SUB_END(k) stamp_acc_[k] += __builtin_readcyclecounter() - sub_t_
#else
#define STAMP(k)
#define SUB_BEGIN()
#define SUB_END(k)
#endif

namespace xpic {

int experiment_ecsim() { return XPIC_TU_EXPERIMENT; }

using namespace fill;

namespace {

// gathering assembly: what is requested TWO cells ahead -- the cell's range and the source indices of its first slots
struct PrefetchIdx {
  int start, cnt;
  int src, srcx;   // source index of slot start + lane / start + kCP + lane
};
// (the 32-bit offsets of the gather reach `gwin` slots below and above the pencil's first slot: 2^28 in production -- the
// widest a 32-bit byte offset allows -- and a few hundred in the parity tests of the far arm of the gather:
// xpic_debug_set(ctx, XPIC_DEBUG_GATHER_WINDOW, n))

// P2: power-of-two spacings (exact reciprocals, device_common.h); FX: nx is a multiple of the chunk width, so every
// chunk is full and its flush is the aligned 32-byte form (the partial-chunk code paths compile away)
// GA ("gather"): the sort's re-binning deferred its scatter (particles.hip: sort_rebin(.., defer)): slot d of a cell holds
// the index src[d] of its record in the OLD order; the record is moved by `step`, wrapped (the same arithmetic as k_scatter)
// and written to r2 / v2 [d] on its way into phase 1 -- the scatter pass of the re-binning (56 B read + 48 B written per
// particle, 27 ms of the 256^3 x 64 step) shrinks to the index pass and these stores.
// GAK: 0 the records lie sorted; 1 gathered through the index k_index built (s.src: z-slabs, a cell beyond its bucket);
// 2 through the binning's buckets.  Compile-time: at 256 registers per lane the two gathering forms in one body were 18
// registers in scratch.
template <bool P2, bool FX, int GAK>
__global__ void __launch_bounds__(kThreads, FILL_OCC) k_ecsim_fill(GridDev g, SortDev s, const double* __restrict__ B,
  double* currI, double* matL, const unsigned short* __restrict__ dtab, const int* __restrict__ linetab, const int* __restrict__ cowr, double q, double m,
  double mpw, int cy0, int cystep, int ncy, int cz0, int czstep, int my_order, int ncol_y, int per_y, int per_z, int first_sort,
  int alias_rows, unsigned long long zord, double step, int* __restrict__ gerr, int ga_store, int bucket_cap, int gwin)
{
  constexpr bool GA = GAK != 0;
  const int cy = cy0 + (int)(blockIdx.x % ncy) * cystep;
  const int cz = cz0 + (int)(blockIdx.x / ncy) * czstep;
#ifdef FILL_STAMPS
  unsigned long long stamp_t_ = 0;
  unsigned long long stamp_acc_[16] = {};
#endif

  __shared__ __attribute__((aligned(16))) double sh[kW * kStage];
  __shared__ __attribute__((aligned(16))) double zslot[kPitch]; // the particle of weight zero that fills up a K = 4 step
  __shared__ double bnb[kW][54];
  // the wave index is the same for all lanes: say so (readfirstlane), or the compiler treats every branch and count that
  // depends on it (active, cnt, lastpass, the K-step loops) as divergent and guards them with exec masks and VALU compares
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  double* st = sh + wave * kStage;
  const double dt = g.dt;

  // Phase 2 runs on the matrix cores.  All particles of one octant touch the same 3 x 8 nodes: their block is the
  // dense 24 x 24  D[(c1,n1)][(c2,n2)] = sum_p s_p[c1][n1] * s_p[c2][n2] * (A_p matB_p)[c1][c2],  cut into 36 blocks of
  // 4 x 4 (component pair x (h1, h2)) and issued as nine v_mfma_f64_4x4x4_4b_f64 over K = 4 particles: instruction
  // (c1, c2), block b = (h1, h2).  Lane roles (probed, tools/ubench/mfma_f64_4x4.hip): A[b][i][k], B[b][k][j] at lane
  // 16 k + 4 b + (i or j); D[b][i][j] at lane 16 i + 4 b + j.  Two more instructions give the cell's
  // currI = sum_p s_p[c][n] I_p[c]: blocks (X,h), (Y,h) resp. (Z,h) with the multiplier I_p[c] in column j = 0.
  const int kk = lane >> 4, qb = (lane >> 2) & 3, qj = lane & 3;
  const int offA = qj * 2 + (qb >> 1);                   // + c1 * 8: row weights s[c1][i = qj][h1]
  const int offB = qj * 2 + (qb & 1);                    // + c2 * 8: column weights s[c2][j = qj][h2]
  const int offI18 = 8 * (kOffAB + 9 + (qb >> 1));       // byte offset of I_p[X] for the blocks (X, h), of I_p[Y] for (Y, h)
  const int offA8 = 8 * offA, offB8 = 8 * offB;

  if (threadIdx.x < kPitch) zslot[threadIdx.x] = 0.0;

  const long pencil0 = ((long)cz * g.ny + cy) * g.nx;
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
    // Co-writers sit at pencil offsets (-dy', -dz') listed in cowr[line]; launch order = (position of the cz colour in the
    // launch sequence, zord: 4 bits per colour) * ncol_y + cy colour.
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
        const int cb = g.G == 0 ? (pz < bodyz ? pz % per_z : per_z + (pz - bodyz)) : pz % per_z;
        if ((int)((zord >> (4 * cb)) & 15u) * ncol_y + ca < my_order) first = false; // zord: launch position of z colour cb
      }
    }
    lbase[mm] = (uintptr_t)base | (first ? 1u : 0u);
  }

  __syncthreads(); // (the zero slot)

  // cells are visited in the order 1, 2, ..., nx-1, 0 so that finished columns start 64-byte aligned
  auto cell_x = [&](int i) { return (i + 1 == g.nx) ? 0 : i + 1; };
  int box;
  const double* const brow = bnb_row(g, B, lane, cy, cz, &box); // this lane's value of a cell's B neighbourhood: row, x offset
  auto prefetch_cell = [&](int i, Prefetch& pf) {
    pf.start = 0; pf.cnt = 0; pf.b = 0.0;
    if (i >= g.nx) return;
    const int cx = cell_x(i);
    {
      // wave-uniform index: a scalar load (cell_start does not change while this kernel runs; requesting it a chunk
      // ahead of the particle loads it addresses was measured: no change)
      using UniformInts = const __attribute__((address_space(4))) int*;
      UniformInts cs = (UniformInts)(s.cell_start + pencil0);
      const int cxu = __builtin_amdgcn_readfirstlane(cx);
      pf.start = cs[cxu];
      pf.cnt = cs[cxu + 1] - pf.start;
    }
    pf.b = brow ? brow[g.wx(cx + box)] : 0.0;
    if (lane < min(kCP, pf.cnt)) {
      const long p = (long)pf.start + lane;
#pragma unroll
      for (int a = 0; a < 3; ++a) { pf.p[a] = s.r[a][p]; pf.p[3 + a] = s.v[a][p]; }
    }
  };
  // ---- gathering form.  Addresses: 32-bit byte offsets from two wave-uniform bases per array -- the old-order arrays from
  // `gwin` slots below the pencil's first slot (records further away take 64-bit addresses, lane by lane), the
  // new-order arrays from the pencil's first slot -- so that one shifted index serves the six arrays of a
  // record (64-bit address arithmetic per array cost 24 vector instructions per pass).
  using UniformIntsG = const __attribute__((address_space(4))) int*;
  const long pn0 = GA ? (long)((UniformIntsG)(s.cell_start + pencil0))[0] : 0;
  const long gb = pn0 > gwin ? pn0 - gwin : 0;
  const unsigned near_span = 2u * (unsigned)gwin;
  const unsigned gb32 = (unsigned)gb;
  const char* rb[6];
  char* wb[6];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    rb[a] = reinterpret_cast<const char*>(s.r[a] + gb); rb[3 + a] = reinterpret_cast<const char*>(s.v[a] + gb);
    wb[a] = reinterpret_cast<char*>(s.r2[a] + pn0); wb[3 + a] = reinterpret_cast<char*>(s.v2[a] + pn0);
  }
  // (ONE base per buffer + the pitch between its arrays, the other bases formed by scalar additions where they are used, was
  // measured: 32 of the kernel's 145 v_readlane -- scalar registers spilled into vector lanes -- go, as many scalar additions
  // come, the wave cycles stay)
  const long pencil_pop = GA ? (long)((UniformIntsG)(s.cell_start + pencil0))[g.nx] - pn0 : 0;
  if (GA && pencil_pop >= (1L << 29) && threadIdx.x == 0) atomicOr(gerr, kFillErrPencil); // the sorted copy's 32-bit offsets: the host fails the step
  // A record further than 2^29 slots above `gb` (what crossed the periodic z-boundary arrives from the other end of the
  // array; a particle that jumped dozens of planes) takes a plain 64-bit address: per lane, rare.
  // Every lane of the wave calls it (`on`: this lane wants a record).  The far records are fetched in a branch the WHOLE
  // wave takes or skips, complete with its wait: as the two arms of one per-lane branch the arms' loads shared their
  // destination registers, and the compiler guarded that with s_waitcnt vmcnt(0) in front of the near arm's loads -- the
  // memory counter being in order, a wait for the six stores of the sorted copy issued a phase 1 before (8 ms per assembly).
  // z-slabs: a negative source index -1 - i is record i of what the neighbours sent (SortDev::inc, {r, v} per record, moved
  // and wrapped by its sender); the returned mask names those lanes: settle() must not move them again.
  auto gather = [&](int srcidx, bool on, double (&rec)[6]) -> unsigned long long {
    SUB_BEGIN();
    // 32-bit arithmetic modulo 2^32: gb and every source index lie in [0, 2^31), so an index below gb, and the negative
    // index of a received record, wrap to 2^31 or more -- not near.  The offset is made opaque: knowing its range the
    // compiler widened it and added it to each of the six bases with 64-bit vector arithmetic (7 instructions per gather)
    // instead of handing the 32-bit register to the load as its offset from a scalar base.
    const unsigned rel = (unsigned)srcidx - gb32;
    const bool near = rel < near_span;
    if (on && near) {
      unsigned off8 = rel << 3;
      asm volatile("" : "+v"(off8));
#pragma unroll
      for (int a = 0; a < 6; ++a) rec[a] = *reinterpret_cast<const double*>(rb[a] + (size_t)off8);
    }
    unsigned long long incm = 0ull;
    if (__builtin_expect(__ballot(on && !near) != 0, 0)) {
      const bool inc = on && srcidx < 0;
      incm = __ballot(inc);
      double far[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      if (on && !near && !inc) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { far[a] = s.r[a][srcidx]; far[3 + a] = s.v[a][srcidx]; }
      }
      if (inc) {
        const double* q = s.inc + 6L * (-1 - srcidx);
#pragma unroll
        for (int a = 0; a < 6; ++a) far[a] = q[a];
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
      for (int a = 0; a < 6; ++a) rec[a] = (on && !near) ? far[a] : rec[a];
    }
    SUB_END(11); // the gather's issue (addresses, six loads, the far branch)
    return incm;
  };
  // where slot i of cell cx finds its source index: the binning's bucket of the cell, or the index k_index built
  auto idx_of = [&](int cx, int start, int i) {
    return GAK == 2 ? s.bucket[(pencil0 + cx) * bucket_cap + i] : s.src[(long)start + i];
  };
  // two cells ahead: the cell's range and the source indices of its first 2 kCP slots (the index -> record chain of one
  // cell ahead was 2.6 ms of the assembly)
  auto prefetch_idx = [&](int i, PrefetchIdx& pi) {
    pi.start = 0; pi.cnt = 0; pi.src = 0; pi.srcx = 0;
    if (i >= g.nx) return;
    const int cxu = __builtin_amdgcn_readfirstlane(cell_x(i));
    UniformIntsG cs = (UniformIntsG)(s.cell_start + pencil0);
    pi.start = cs[cxu];
    pi.cnt = cs[cxu + 1] - pi.start;
    if (lane < min(kCP, pi.cnt)) pi.src = idx_of(cxu, pi.start, lane);
    if (lane < min(kCP, pi.cnt - kCP)) pi.srcx = idx_of(cxu, pi.start, kCP + lane);
  };
  auto prefetch_rec = [&](int i, const PrefetchIdx& pi, Prefetch& pf) {
    pf.start = pi.start; pf.cnt = pi.cnt; pf.b = 0.0; pf.srcx = pi.srcx; pf.incm = 0ull;
    if (i >= g.nx) return;
    pf.b = brow ? brow[g.wx(cell_x(i) + box)] : 0.0;
    pf.incm = gather(FILL_GA_NOCHAIN ? pf.start + lane : pi.src, lane < min(kCP, pf.cnt), pf.p);
  };
  // a record that has just arrived from the old order is moved, wrapped (k_scatter's arithmetic, bit for bit) and written
  // to its slot d of the new order.  (Wrapping only in the cells on the box's boundary is wrong: a particle that moves
  // several cells in a step lands further inside.)
  // Every lane runs it (`fresh`: the lane's record has just arrived): a lane whose record is older, or came from a
  // neighbouring slab, moves by v * 0 = nothing, and the fold leaves what lies inside the box as it is -- as a branch around
  // the whole thing the compiler merged the two versions of the record with 22 register moves per pass, and the `||` of the
  // six range tests became five nested branches.
  auto settle = [&](double (&cur)[6], int drel, bool fresh, unsigned long long incm) {
    SUB_BEGIN();
    bool mv = fresh;
    if (__builtin_expect(incm != 0ull, 0)) mv = fresh & !((incm >> lane) & 1ull); // (the sender moved it)
    const double st = mv ? step : 0.0;
    cur[0] += cur[3] * st;
    cur[1] += cur[4] * st;
    cur[2] += cur[5] * st;
    // (the fold's thirty instructions are skipped where no lane of the pass left the box -- all but the cells next to its
    // faces, and not only those: a fast particle lands further inside)
    const bool out = fresh & ((cur[0] < 0.0) | (cur[0] > g.Lx) | (cur[1] < 0.0) | (cur[1] > g.Ly) | (cur[2] < 0.0) | (cur[2] > g.Lz));
    if (__builtin_expect(__ballot(out) != 0, 0)) {
      cur[0] = bound_periodic_sel(cur[0], g.Lx);
      cur[1] = bound_periodic_sel(cur[1], g.Ly);
      cur[2] = bound_periodic_sel(cur[2], g.Lz);
    }
    if (!ga_store) return; // (xpic_set_fused_rebin 2: k_second_push writes the sorted copy)
    SUB_END(9); // move + wrap
#ifdef FILL_STAMPS
    const unsigned long long sub2_t_ = __builtin_readcyclecounter();
#endif
    if (fresh) {
      const unsigned off8 = (unsigned)drel << 3;
#if FILL_GA_EXP != 1
#pragma unroll
      for (int a = 0; a < (FILL_GA_EXP == 2 ? 3 : 6); ++a) {
        *reinterpret_cast<double*>(wb[a] + (size_t)off8) = cur[a]; // (non-temporal stores: measured, no change)
      }
#endif
    }
#ifdef FILL_STAMPS
    stamp_acc_[10] += __builtin_readcyclecounter() - sub2_t_; // the six stores of the sorted copy
#endif
  };

  Prefetch pf;
  PrefetchIdx pi;
  if (GA) {
    prefetch_idx(wave, pi);
    prefetch_rec(wave, pi, pf);
    prefetch_idx(wave + kW, pi);
  }
  else prefetch_cell(wave, pf);
  __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): see the comment in front of the flush
  double carry[kOwn][2];
#pragma unroll
  for (int mm = 0; mm < kOwn; ++mm) carry[mm][0] = carry[mm][1] = 0.0;

#ifdef FILL_STAMPS
  stamp_t_ = __builtin_readcyclecounter();
#endif
  // the cell block of this wave: zeroed here and again right behind the merge that consumes it (zeroing at the top of
  // the chunk made the compiler clear all 36 twice: once for the path around the pass loop and once in front of it)
  double acc[kAcc];
#pragma unroll
  for (int e = 0; e < kAcc; ++e) acc[e] = 0.0;
  const int nch = (g.nx + kW - 1) / kW;
  for (int j = 0; j < nch; ++j) {
    STAMP(0);
    const int i = j * kW + wave;
    const bool active = FX || i < g.nx; // full chunks: every wave has a cell

    if (active) {
      const int start = pf.start, cnt = pf.cnt;
      if (lane < 54) bnb[wave][lane] = pf.b;
      double cur[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) cur[a] = pf.p[a];
      const int srcx_cur = pf.srcx;
      if (GA) settle(cur, (int)((long)start - pn0) + lane, lane < min(kCP, cnt), pf.incm);
      unsigned long long fresh_incm = 0ull;
      int fresh = -1; // GA: slot (relative to the cell's first) of a record this lane requested during the last pass
      // particles.cpp:107-115: the factors that do not depend on the particle
      const double fb = (0.5 * dt) * q / m;
      const double qw = q * mpw;                          // iq  = q * mpw / (1 + b^2)
      const double Aq = 0.5 * dt * dt * mpw * q * q / m;  // A_p = 0.5 dt^2 mpw q^2 / m / (1 + b^2)
#if FILL_EXP >= 6 && FILL_EXP <= 8
      for (int e = 0; e < kAcc; ++e) acc[e] = 1.0 + lane;
#endif
      // A pass stages up to kCP particles.  Phase 2 works on 4 particles of one octant per step; a pass that is not the
      // cell's last leaves the 0..3 particles that do not fill a step of their octant to the next pass (their lanes keep
      // them and stage them again with the newcomers) instead of padding every octant in every pass: a Poisson(64)
      // cell then takes sum_o ceil(n_o / 4) = 19 steps on average instead of 22.4.
      int handed = min(kCP, cnt);     // particles of the cell handed to lanes so far
      bool real = lane < handed;
      while (cnt > 0) {
        const bool lastpass = handed >= cnt;
#if FILL_EXP >= 6 && FILL_EXP <= 8
        real = false;
#endif
        wave_sync(); // the previous pass's operand reads (and the neighbourhood store) are done
        if (GA && __ballot(fresh >= 0) != 0) { settle(cur, (int)((long)start - pn0) + fresh, fresh >= 0, fresh_incm); fresh = -1; }
        // CIC weights and the half-cell octant first: the octant decides the particle's stage slot
        const W1T<P2> w(g, cur[0], cur[1], cur[2]); // lanes without a particle: garbage in, masked by oct = 8
        const int ox = w.is[0] - w.in[0] + 1, oy = w.is[1] - w.in[1] + 1, oz = w.is[2] - w.in[2] + 1;
        const int oct = real ? (ox | (oy << 1) | (oz << 2)) : 8;
        // compaction by octant: the particles of octant o take the stage slots ooff[o] .. ooff[o] + ocnt[o] - 1
        int ocnt[8], ooff[8], slot = 0;
        bool keep = false; // this lane's particle waits for the next pass
        {
          // (rank and size of the lane's own octant leave the loop as lane values and `keep` is formed behind it: assigned
          // inside the predicated block, the flag was carried as a lane mask through five scalar instructions per octant)
          int run = 0, rk_own = 0, full_own = 0;
#pragma unroll
          for (int o = 0; o < 8; ++o) {
            const unsigned long long mk = __ballot(oct == o);
            ocnt[o] = __popcll(mk);
            ooff[o] = run;
            if (oct == o) {
              rk_own = __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
              slot = run + rk_own;
              full_own = ocnt[o] & ~3;
            }
            run += ocnt[o];
          }
          keep = !lastpass && oct < 8 && rk_own >= full_own;
        }
        if (real) {
          // (shifting the odd-ranked particles of an octant by 16 bytes inside their slot, so that the two particles whose
          // operand reads share an LDS cycle in phase 2 sit 16 instead of 12 banks apart, was measured: 98.9 against 98.1 ms)
          double2* dst = (double2*)(st + slot * kPitch);
          // The 24 values of the cell's B neighbourhood this particle's octant touches are requested FIRST (their
          // addresses depend on the octant alone) and consumed last: the LDS latency runs under the 36 weight products
          // and their stores instead of being waited for batch by batch in the middle of the gather.
          const double* nb = bnb[wave];
          double nbv[3][8];
#pragma unroll
          for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
              for (int ii = 0; ii < 2; ++ii) {
                nbv[0][(k * 2 + jj) * 2 + ii] = nb[((oz + k) * 3 + (oy + jj)) * 2 + ii];
                nbv[1][(k * 2 + jj) * 2 + ii] = nb[18 + ((oz + k) * 2 + jj) * 3 + (ox + ii)];
                nbv[2][(k * 2 + jj) * 2 + ii] = nb[36 + (k * 3 + (oy + jj)) * 3 + (ox + ii)];
              }
          asm volatile("" ::: "memory"); // the reads are issued before anything below
          // the 8 weights per component (:138-140, :145-147), [i][h]: i = the 2 x 2 transverse nodes, h = lower /
          // upper node along the component's staggered axis; stored as they are formed
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
              const double tx = w.wn[2][a] * w.wn[1][bb];  // X: i = k * 2 + j
              dst[0 + a * 2 + bb] = double2{tx * w.ws[0][0], tx * w.ws[0][1]};
              // Y: i = k * 2 + ix, s = wnz[k] * wsy[h] * wnx[ix] in the reference's product order
              dst[4 + a * 2 + bb] = double2{w.wn[2][a] * w.ws[1][0] * w.wn[0][bb], w.wn[2][a] * w.ws[1][1] * w.wn[0][bb]};
              // Z: i = j * 2 + ix, s = wsz[h] * wny[j] * wnx[ix]
              dst[8 + a * 2 + bb] = double2{w.ws[2][0] * w.wn[1][a] * w.wn[0][bb], w.ws[2][1] * w.wn[1][a] * w.wn[0][bb]};
            }
          const double v[3] = {cur[3], cur[4], cur[5]};
          // interpolate_B_s1 (ecsim/simulation.cpp:64-118) out of the cell's LDS neighbourhood, same loop
          // and product order as the global-memory gather
          double Bp[3] = {0.0, 0.0, 0.0};
#pragma unroll
          for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
              for (int ii = 0; ii < 2; ++ii) {
                Bp[0] += nbv[0][(k * 2 + jj) * 2 + ii] * (w.ws[2][k] * w.ws[1][jj] * w.wn[0][ii]);
                Bp[1] += nbv[1][(k * 2 + jj) * 2 + ii] * (w.ws[2][k] * w.wn[1][jj] * w.ws[0][ii]);
                Bp[2] += nbv[2][(k * 2 + jj) * 2 + ii] * (w.wn[2][k] * w.ws[1][jj] * w.ws[0][ii]);
              }
          const double bx = Bp[0] * fb, by = Bp[1] * fb, bz = Bp[2] * fb;
          const double b2 = bx * bx + by * by + bz * bz;
          const double vb = v[0] * bx + v[1] * by + v[2] * bz;
          const double cxv = +(v[1] * bz - v[2] * by), cyv = -(v[0] * bz - v[2] * bx), czv = +(v[0] * by - v[1] * bx);
          // one division for both factors (the reference divides twice: :112, :116; one rounding apart)
          const double rb = 1.0 / (1. + b2);
          const double iq = qw * rb;
          const double A_p = Aq * rb;
          dst[12] = double2{A_p * (1.0 + bx * bx), A_p * (+bz + bx * by)};
          dst[13] = double2{A_p * (-by + bx * bz), A_p * (-bz + by * bx)};
          dst[14] = double2{A_p * (1.0 + by * by), A_p * (+bx + by * bz)};
          dst[15] = double2{A_p * (+by + bz * bx), A_p * (-bx + bz * by)};
          dst[16] = double2{A_p * (1.0 + bz * bz), iq * (v[0] + cxv + vb * bx)};
          dst[17] = double2{iq * (v[1] + cyv + vb * by), iq * (v[2] + czv + vb * bz)};
        }
        // next pass of this cell: the free lanes take the next particles, loads in flight during this pass's phase 2
        bool real_next = false;
        if (!lastpass) {
          const unsigned long long km = __ballot(keep);
          const int take = min(cnt - handed, kCP - (int)__popcll(km));
          const unsigned long long fm = ~km;
          const int fr = __builtin_amdgcn_mbcnt_hi((unsigned)(fm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)fm, 0u));
          const bool get = !keep && fr < take;
          if (GA) {
            // the second pass's source indices came with the cell (srcx: slot kCP + lane); the lane that takes slot
            // handed + fr fetches the index from lane handed + fr - kCP (every lane takes part in the shuffle)
            const int q = handed + fr - kCP;
            int srcidx = __shfl(srcx_cur, q & 63, 64);
            // a cell's third pass (slots beyond 2 kCP) reads its indices here: a branch of the whole wave that ends with
            // its own wait (merged into the common path, the load brought a vmcnt(0) -- a wait for this pass's stores -- to
            // every refill)
            const bool late = get && !(q >= 0 && q < kCP);
            if (__builtin_expect(__ballot(late) != 0, 0)) {
              int t = 0;
              if (late) t = idx_of(cell_x(i), start, handed + fr);
              __builtin_amdgcn_s_waitcnt(0x0F70);
              srcidx = late ? t : srcidx;
            }
            if (FILL_GA_NOCHAIN) srcidx = start + handed + fr;
            if (get) fresh = handed + fr;
            fresh_incm = gather(srcidx, get, cur);
          }
          else if (get) {
            const long p = (long)start + handed + fr;
#pragma unroll
            for (int a = 0; a < 3; ++a) { cur[a] = s.r[a][p]; cur[3 + a] = s.v[a][p]; }
          }
          real_next = keep || get;
          handed += take;
        }
        wave_sync();
            STAMP(1);

        // ---- phase 2: octant by octant (compile-time octant = compile-time accumulators), K = 4 particles per step;
        // a step beyond the octant's last particle reads the zero slot.  (A software pipeline over the steps -- one rotating
        // operand set, the next step's ten reads issued between this step's products and its matrix instructions, also
        // across octants -- compiled as intended, 6 register moves per step, and was SLOWER: 115.1 against 108.6 ms per
        // assembly.  The other wave of the SIMD already covers the read latency; what a step costs is issue slots.)
        // wave priority: of the two waves on a SIMD (one per workgroup of the CU) the one with matrix instructions to issue
        // goes first: 97.5 -> 96.0 / 96.3 ms per assembly (priority 2 / 3: 96.3-96.6; raising phase 1 or the chunk's tail
        // instead, or too: 97.2-97.8)
        __builtin_amdgcn_s_setprio(1);
#if FILL_EXP != 1 && !(FILL_EXP >= 6 && FILL_EXP <= 8)
#pragma unroll
        for (int o = 0; o < 8; ++o) {
          const int no = ocnt[o];
          const double* seg = st + ooff[o] * kPitch;
          const int nst = lastpass ? (no + 3) >> 2 : no >> 2;
          // 32-bit LDS addresses: the lane's read positions inside a slot are fixed byte offsets, a step moves 4 slots on,
          // and "is there a particle for my row" is one compare of the lane's slot address with the end of the segment
          LdsBytes spr = (LdsBytes)(const char*)(seg + kk * kPitch);
          const LdsBytes seg_end = (LdsBytes)(const char*)(seg + no * kPitch);
          auto kstep = [&](const LdsBytes sb) {
            const LdsDouble* sp = (const LdsDouble*)sb;
            const LdsDouble* spA = (const LdsDouble*)(sb + offA8);
            const LdsDouble* spB = (const LdsDouble*)(sb + offB8);
            double a[3], b[3];
            // (the compiler pairs these into ds_read2_b64; unpaired ds_read_b64 -- half the LDS cycles on paper -- measured 96.7
            // against 96.1 ms)
#pragma unroll
            for (int c = 0; c < 3; ++c) { a[c] = spA[c * 8]; b[c] = spB[c * 8]; }
            // the step's 9 A_p*matB per particle: the same 72 bytes for the 16 lanes of a particle (a DPP row
            // broadcast of one 8-byte read per lane was measured 6 % slower: the VALU is the scarcer resource here)
            const LdsDouble2* u = (const LdsDouble2*)(sp + kOffAB);
            const dpair u0 = u[0], u1 = u[1], u2 = u[2], u3 = u[3];
            const double ab[9] = {u0.x, u0.y, u1.x, u1.y, u2.x, u2.y, u3.x, u3.y, sp[kOffAB + 8]};
            // the nine column operands first, then the matrix instructions back to back: a product issued between two
            // MFMAs waits for the fp64 pipe to drain and the next MFMA waits for the product
            double bm[9];
#pragma unroll
            for (int e = 0; e < 9; ++e) bm[e] = b[e % 3] * ab[e];
            // currI operands: block b of instruction 1 is (X or Y, h = b & 1), of instruction 2 (Z, h = b & 1): their
            // row weights are column weights already loaded.  The multiplier I_p[c] belongs in column j = 0 only, but a
            // column j of D depends on column j of the B operand alone and the columns j != 0 of these two accumulators
            // (and the blocks 2, 3 of the second) have no target in the merge (they go to the lane's dummy double): every
            // lane passes I_p[c], read at a lane-dependent offset, and nothing has to be selected or zeroed
            const double ai1 = qb < 2 ? b[0] : b[1], ai2 = b[2];
            const double bi1 = *(const LdsDouble*)(sb + offI18), bi2 = sp[kOffAB + 11];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c1 = 0; c1 < 3; ++c1)
#pragma unroll
              for (int c2 = 0; c2 < 3; ++c2)
                acc[acc_main(c1, c2, o)] =
                  __builtin_amdgcn_mfma_f64_4x4x4f64(a[c1], bm[c1 * 3 + c2], acc[acc_main(c1, c2, o)], 0, 0, 0);
            acc[acc_cur1(o)] = __builtin_amdgcn_mfma_f64_4x4x4f64(ai1, bi1, acc[acc_cur1(o)], 0, 0, 0);
            acc[acc_cur2(o)] = __builtin_amdgcn_mfma_f64_4x4x4f64(ai2, bi2, acc[acc_cur2(o)], 0, 0, 0);
          };
          // (full steps without the zero-slot compare / select, the partial step peeled, was measured: 99.2 against 96.1 ms --
          // the octant bodies double; the instruction stream, not its count, pays)
          for (int t = 0; t < nst; ++t, spr += 4 * kPitch * 8) kstep(spr < seg_end ? spr : (LdsBytes)(const char*)zslot);
        }
#endif
        __builtin_amdgcn_s_setprio(0);
        STAMP(2);
        // the next pass's particles are waited for HERE (they had this pass's phase 2 to arrive; on every path out of the
        // pass body, or the compiler keeps its own): left to the compiler the wait sits at the head of EVERY pass, the
        // first one of a chunk included, where -- the memory counter being in order -- it is a wait for the previous
        // chunk's flush stores
        {
          SUB_BEGIN();
          __builtin_amdgcn_s_waitcnt(0x0F70);
          SUB_END(13);
        }
        if (lastpass) break;
        real = real_next;
      }
    }

    STAMP(2);
    // next chunk's cell: particle data and B neighbourhood travel while this chunk is merged and flushed
    // (nothing is outstanding here but the previous chunk's stores: said explicitly, so that the compiler does not guard its
    // re-use of the pass loop's load registers with waits BEHIND the requests below)
    {
      SUB_BEGIN();
      __builtin_amdgcn_s_waitcnt(0x0F70);
      SUB_END(12);
    }
    if (GA) {
      prefetch_rec(i + kW, pi, pf);   // its indices were requested a chunk ago
      // (The cell's range is a scalar load consumed on the spot, and the in-kernel timers put 10 ms of the assembly on its
      // wait -- every fourth chunk's row of cell_start comes from HBM.  Round 5 took the wait away -- the range riding in
      // two lanes of the index load, requested behind the flush -- and the assembly took 108.8 - 109.0 ms against 107.0 -
      // 108.0 on the same box (profiles/r05_fill_idx_ab.txt): the other workgroup's wave on the SIMD issues during the
      // stall; what the kernel is short of is issue slots, not latency hiding.)
      SUB_BEGIN();
      prefetch_idx(i + 2 * kW, pi);
      SUB_END(14);
    }
    else prefetch_cell(i + kW, pf);
#ifdef FILL_STAMPS
    const unsigned long long sub3_t_ = __builtin_readcyclecounter();
#endif

    // ---- the finished columns of this chunk leave by read-modify-write: request their current values NOW, so
    // that the HBM latency runs under the merge below (addresses depend only on the chunk, not on the data)
    const int ndone = FX ? kW : min(kW, g.nx - j * kW);
    // matL lines: the kW finished columns are the x-block j of the row, 32 contiguous, aligned bytes;
    // currI lines: kW consecutive doubles, 16-byte aligned when nx is even
    // alias_rows: a y or z extent of 2 cells folds the row offsets -1 and +1 of a pencil onto the SAME row, so two lines of
    // this workgroup own one stream: their flush adds with atomics instead of the plain read-modify-write (general
    // instantiation only; such boxes are the reference's quasi-1D set-ups, config.json:5-7)
    const bool alias = !FX && alias_rows != 0;
    const bool vecL = FX || (ndone == kW && !alias), vecI = FX || (vecL && (g.nx & 1) == 0);
    double old[kOwn][kW];
    GlobalDouble* ptr[kOwn];
    bool fst[kOwn];
#pragma unroll
    for (int mm = 0; mm < kOwn; ++mm) {
      const int line = threadIdx.x + mm * kThreads;
      const uintptr_t lb = lbase[mm];
      fst[mm] = lb & 1;
      const bool vec = line < kMatLines ? vecL : vecI;
      ptr[mm] = lb ? (GlobalDouble*)(lb & ~(uintptr_t)1) + (long)j * (line < kMatLines ? kLBlock : kW) : nullptr;
#pragma unroll
      for (int c = 0; c < kW; ++c) old[mm][c] = 0.0;
      if (ptr[mm] && !fst[mm] && FILL_EXP != 8 && FILL_EXP != 11 && FILL_EXP != 12) {
        if (vec) {
#pragma unroll
          for (int c = 0; c < kW; c += 2) {
            const dpair v = *(const GlobalPair*)(ptr[mm] + c);
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
    unsigned wdst[kAcc]; // requested now, used after the window is seeded
#pragma unroll
    for (int e = 0; e < kAcc; ++e) wdst[e] = 0;
    {
      // the lane's 36 offsets are 72 contiguous bytes of the table (transposed on the host): five 16-byte loads (as 32-bit
      // words, nine loads and no unpacking: 97.2 against 96.1 ms)
      const uint4* q = reinterpret_cast<const uint4*>(dtab + lane * kDtabPitch);
#pragma unroll
      for (int k = 0; k < kDtabPitch / 8; ++k) {
        const uint4 w4 = q[k];
        const unsigned w[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int h = 0; h < 8; ++h)
          if (k * 8 + h < kAcc) wdst[k * 8 + h] = (w[h >> 1] >> (16 * (h & 1))) & 0xffffu; // byte offset
      }
    }
#ifdef FILL_STAMPS
    stamp_acc_[15] += __builtin_readcyclecounter() - sub3_t_; // RMW requests + the merge's offset table loads
#endif
    STAMP(3);
    lds_barrier();
    STAMP(4);
    double* win = sh; // [kLines][kSlots]
#pragma unroll
    for (int mm = 0; mm < kOwn; ++mm) {
      const int line = threadIdx.x + mm * kThreads;
      if (line < kLines) {
        if (kWP % 2 == 0) {
          double2* w = (double2*)(win + line * kWP);
          w[0] = double2{carry[mm][0], carry[mm][1]};
#pragma unroll
          for (int c = 1; c < kSlots / 2; ++c) w[c] = double2{0.0, 0.0};
        }
        else {
          double* w = win + line * kWP;
          w[0] = carry[mm][0]; w[1] = carry[mm][1];
#pragma unroll
          for (int c = 2; c < kSlots; ++c) w[c] = 0.0;
        }
      }
    }
    lds_barrier();
    STAMP(5);
    if (active && !(FILL_EXP >= 7 && FILL_EXP <= 8)) {
      // row node x of cell (unwrapped) u = i+1 with offset o is column u+o; window column 0 is kW*j: the lane's element
      // of accumulator e goes to window byte wdst[e] (+ 8 * wave); elements without a target (currI instructions,
      // idle blocks) add into a per-lane dummy double behind the window.  No branches, no waits between the atomics.
      char* wv = (char*)(win + wave);
#pragma unroll
      for (int e = 0; e < kAcc; ++e) {
        // (masking out the lanes of the six currI accumulators that have no target, instead of letting them add into their
        // dummy doubles, was measured: no difference)
        unsafeAtomicAdd((double*)(wv + wdst[e]), acc[e]);
        acc[e] = 0.0;
      }
    }
    lds_barrier();
    STAMP(6);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    // (The next chunk's first use of the prefetched registers is guarded by s_waitcnt vmcnt(0): the stores of the flush
    // below sit in divergent branches, so the compiler cannot count them, and every chunk begins by waiting for its
    // predecessor's stores.  Explicit waits remove that one step by step.  1 -- all reads declared complete here and before
    // the loop: 108.5 against 97.7 ms, NOT because of the stores: the compiler then guards its re-use of the pass loop's
    // load registers with vmcnt(3) .. (0) BEHIND the next chunk's requests, a full memory round trip.  2 -- also before
    // those requests: 97.9.  3 -- also at the end of every pass (the head of the pass loop otherwise keeps counted waits
    // that, the counter being in order, drain the stores after all): no vmcnt wait is left between the flush and the end
    // of the next chunk's first pass, and the assembly takes 97.4 ms (A/B/A/B in one call: 98.12, 97.47, 98.15, 97.39).
    // The acknowledgements of the flush stores were never most of what a chunk waits for; all three are in.)
    // ---- stream out the finished columns (plain RMW, kW consecutive doubles per line), keep 2 in registers
#pragma unroll
    for (int mm = 0; mm < kOwn; ++mm) {
      const int line = threadIdx.x + mm * kThreads;
      if (line < kLines) {
        double w[kSlots];
        if (kWP % 2 == 0) {
          const double2* wp = (const double2*)(win + line * kWP);
#pragma unroll
          for (int c = 0; c < kSlots / 2; ++c) { const double2 v = wp[c]; w[2 * c] = v.x; w[2 * c + 1] = v.y; }
        }
        else {
#pragma unroll
          for (int c = 0; c < kSlots; ++c) w[c] = win[line * kWP + c];
        }
        const bool vec = line < kMatLines ? vecL : vecI;
        // lines without a contribution from this chunk are left alone (vacuum stays exactly zero and costs no write);
        // the full-chunk instantiation skips the test: in a plasma every line has one
        bool any = FX || fst[mm];
#pragma unroll
        for (int c = 0; c < kW; ++c) any = any || (c < ndone && w[c] != 0.0);
#if FILL_EXP == 8 || FILL_EXP == 12
        any = false;
#endif
        if (any) {
          if (vec) {
#pragma unroll
            for (int c = 0; c < kW; c += 2)
              *(GlobalPair*)(ptr[mm] + c) = dpair{old[mm][c] + w[c], old[mm][c + 1] + w[c + 1]};
          }
          else if (alias) {
#pragma unroll
            for (int c = 0; c < kW; ++c)
              if (c < ndone && w[c] != 0.0) unsafeAtomicAdd((double*)(ptr[mm] + c), w[c]);
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
    STAMP(7);
    lds_barrier();
    STAMP(8);
  }
#ifdef FILL_STAMPS
  if (threadIdx.x == 0)
    for (int k = 0; k < 16; ++k) atomicAdd(&g_fill_stamps[k], stamp_acc_[k]);
#endif

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
//   dtab[e*64 + lane]  = where the kernel's merge adds lane's element of accumulator e: offset inside the window
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
  // per-lane flush descriptors of the 36 accumulators: D[b][i][j] of a 4x4x4 instruction sits at lane 16 i + 4 b + j
  auto curdesc = [&](int row) {
    const int cc = row / 12;
    int o[3];
    block_node_offset(cc, row % 12, o);
    const int id = cc == 0 ? o[2] * 2 + o[1] : (cc == 1 ? 4 + o[2] * 3 + (o[1] + 1) : 10 + (o[2] + 1) * 2 + o[1]);
    return ((kMatLines + id) << 2) | (o[0] + 1);
  };
  // stored as the offset in doubles inside the merge window [kLines][kSlots] (column = row x offset + 1, the wave adds its
  // own cell's column); elements without a target point at a per-lane dummy double behind the window
  std::vector<unsigned short> dtab(kAcc * 64);
  for (int e = 0; e < kAcc; ++e)
    for (int lane = 0; lane < 64; ++lane) dtab[e * 64 + lane] = (unsigned short)(kLines * kWP + lane);
  auto wbyte = [](int d) { return (unsigned short)((d >> 2) * kWP + (d & 3)); };
  std::vector<int> seen(kAcc, 0);
  for (int o = 0; o < 8; ++o) {
    const int ob[3] = {o & 1, (o >> 1) & 1, (o >> 2) & 1};
    for (int lane = 0; lane < 64; ++lane) {
      const int i = lane >> 4, b = (lane >> 2) & 3, j = lane & 3;
      for (int c1 = 0; c1 < 3; ++c1)
        for (int c2 = 0; c2 < 3; ++c2) {
          const int row = row_of(c1, ob[c1] + (b >> 1), i), col = row_of(c2, ob[c2] + (b & 1), j);
          const int d = etab[row * 36 + col];
          XPIC_CHECK(d >= 0 && d < 0xffff, "an octant block entry has no matL line");
          const int e = acc_main(c1, c2, o);
          XPIC_CHECK(!seen[e] || dtab[e * 64 + lane] == wbyte(d), "accumulator variants are inconsistent");
          dtab[e * 64 + lane] = wbyte(d);
        }
      if (j == 0) {
        const int c1 = b >> 1;
        dtab[acc_cur1(o) * 64 + lane] = wbyte(curdesc(row_of(c1, ob[c1] + (b & 1), i)));
        if (b < 2) dtab[acc_cur2(o) * 64 + lane] = wbyte(curdesc(row_of(2, ob[2] + b, i)));
      }
    }
    for (int c1 = 0; c1 < 3; ++c1)
      for (int c2 = 0; c2 < 3; ++c2) seen[acc_main(c1, c2, o)] = 1;
  }
  static_assert((kLines * kWP + 64) <= kW * kStage, "the dummy doubles must lie inside the stage area");
  { // [lane][kDtabPitch] instead of [e][64], and BYTE offsets (one add per atomic in the kernel)
    static_assert((kLines * kWP + 64 + kW) * 8 <= 0xffff, "window byte offsets must fit 16 bits");
    std::vector<unsigned short> t(64 * kDtabPitch, 0);
    for (int e = 0; e < kAcc; ++e)
      for (int lane = 0; lane < 64; ++lane) t[lane * kDtabPitch + e] = (unsigned short)(8 * dtab[e * 64 + lane]);
    dtab.swap(t);
  }
  const size_t total = linetab.size() + cowr.size() + (dtab.size() + 1) / 2;
  XPIC_HIP(hipMalloc(&c->ltab, sizeof(int) * total));
  XPIC_HIP(hipMemcpy(c->ltab, linetab.data(), sizeof(int) * linetab.size(), hipMemcpyHostToDevice));
  XPIC_HIP(hipMemcpy(c->ltab + linetab.size(), cowr.data(), sizeof(int) * cowr.size(), hipMemcpyHostToDevice));
  XPIC_HIP(hipMemcpy(c->ltab + linetab.size() + cowr.size(), dtab.data(), sizeof(unsigned short) * dtab.size(), hipMemcpyHostToDevice));
  return 0;
}

// Colour classes of one periodic axis of n indices.  Same-colour indices must be >= 3 apart (periodically).
// If 3, 4 or 5 divides n the colours are the residues mod that period (all classes equal: no thin launches);
// otherwise residues mod 3 over the first 3*floor(n/3) indices plus one class per trailing index.
#ifndef XPIC_MAX_PER_Z
#define XPIC_MAX_PER_Z 5 // (3: the round-2 schedule of a slab, plain residues mod 3)
#endif
static int colour_period(int n)
{
  for (int p = 3; p <= 5; ++p)
    if (n % p == 0) return p;
  return 3;
}

// Colour periods (p along y, q along z).  Pencils of one launch must be >= 3 apart along y or z; any p, q >= 3 will do,
// and what they cost is the number of workgroup ROUNDS: a launch of n pencils on S workgroup slots (2 per CU) takes
// ceil(n / S) rounds of one pencil's duration each, whatever n is.  A 256 x 32 slab (BASELINE configs[3] on 8 GPUs) with
// (4, 3) is 12 launches of 704 / 704 / 640 pencils = 24 rounds of 512; with (4, 4) 16 launches of exactly 512 = 16 rounds.
// Periodic extents (y always, z on a single slab) take a period that divides them where one exists (no remainder
// classes: each of those is a launch of a few pencils); a slab's z is not periodic inside the slab (its ghost rows are
// its own), any q >= 3 partitions it.
static void colour_periods(const GridDev& g, int slots, int* per_y, int* per_z)
{
  const int py = colour_period(g.ny);
  if (g.G == 0) { *per_y = py; *per_z = colour_period(g.nzl); return; }
  long best = -1;
  int bq = 3;
  const int ncol_y = py + g.ny % py;
  for (int q = 3; q <= XPIC_MAX_PER_Z; ++q) {
    long rounds = 0;
    for (int b = 0; b < q; ++b) {
      const int ncz = (g.nzl - b + q - 1) / q;
      if (ncz <= 0) continue;
      for (int a = 0; a < ncol_y; ++a) {
        const int ncy = a < py ? (g.ny - g.ny % py) / py : 1;
        rounds += ((long)ncy * ncz + slots - 1) / slots;
      }
    }
    if (best < 0 || rounds < best) { best = rounds; bq = q; }
  }
  *per_y = py; *per_z = bq;
}

static void colour_class(int n, int period, int colour, int* first, int* step, int* count)
{
  const int body = n - n % period;
  if (colour < period) { *first = colour; *step = period; *count = body / period; }
  else { *first = body + (colour - period); *step = 1; *count = 1; }
}

// which body ecsim_fill_sort runs on this context's grid: {power-of-two spacings, full chunks, warp-specialised}
void ecsim_fill_variant(const xpic_ctx* c, int* p2, int* fx, int* ws)
{
  const GridDev& g = c->g;
  const bool alias = g.ny < 3 || (g.G == 0 && g.nzl < 3);
  *p2 = g.pow2 ? 1 : 0;
  *fx = (g.nx % kW == 0 && !alias) ? 1 : 0;
  *ws = (*fx && c->fill_kernel == 1) ? 1 : 0;
}

int ecsim_fill_sort(xpic_ctx* c, Sort& s, const double* B, double* currI_sort, double* matL, bool first_sort, bool post_ghost_rows)
{
  if (s.deferred && s.n == 0) XPIC_CALL(sort_materialize(c, s));
  if (s.n == 0) {
    // nothing to assemble on this slab, but the exchange it was to post is part of every rank's message sequence
    if (post_ghost_rows) XPIC_CALL(matL_ghost_rows_post(c));
    return 0;
  }
  const GridDev& g = c->g;
  int p2, fxi, wsi;
  ecsim_fill_variant(c, &p2, &fxi, &wsi);
  const bool ws = wsi != 0;
  // a deferred re-binning is resolved here: the classic kernel gathers the old-order records through s.d.src, moves and
  // wraps them and writes the sorted copy on its way (the periodic wrap it applies is the deferral's only form); the
  // warp-specialised body has no gathering form: scatter first
  if (s.deferred && (ws || !s.def_wrap)) XPIC_CALL(sort_materialize(c, s));
  const bool ga = s.deferred;
  if (ga && c->profiling) c->prof["fill_gather"].launches += 1; // (what this assembly did, for bench.py's byte counts)
  const bool store_sorted = ga && c->fused_rebin != 2; // 2: the records stay in the old order until k_second_push moves them
  // y is periodic inside the slab; z is periodic only when the slab is the whole box: with z-neighbours the rows
  // below plane 0 / above plane nzl-1 are ghost rows of this rank alone, so plain residues mod per_z suffice
  int per_y, per_z;
  colour_periods(g, (ws ? 1 : 2) * c->num_cus, &per_y, &per_z); // workgroup slots of the chip: the 8-wave kernel owns a CU
  const int ncol_y = per_y + g.ny % per_y, ncol_z = g.G == 0 ? per_z + g.nzl % per_z : per_z;
  // Launch sequence of the z colours.  On a slab the colours of the first and the last plane go first: their pencils are
  // the only ones that write the two ghost row planes of matL, which the neighbours are waiting for -- with
  // `post_ghost_rows` (the last species) the exchange is posted right behind them and travels beside the other colours.
  XPIC_CHECK(ncol_z <= 16, "too many z colours for the launch-order word");
  int seq[16], nseq = 0;
  if (g.G > 0) {
    const int b0 = 0, b1 = (g.nzl - 1) % per_z;
    seq[nseq++] = b0;
    if (b1 != b0) seq[nseq++] = b1;
    for (int b = 0; b < ncol_z; ++b)
      if (b != b0 && b != b1) seq[nseq++] = b;
  }
  else
    for (int b = 0; b < ncol_z; ++b) seq[nseq++] = b;
  unsigned long long zord = 0;
  for (int pos = 0; pos < nseq; ++pos) zord |= (unsigned long long)pos << (4 * seq[pos]);
  const int nboundary = g.G > 0 ? ((g.nzl - 1) % per_z != 0 ? 2 : 1) : 0;
  for (int pos = 0; pos < nseq; ++pos) {
    const int b = seq[pos];
    if (post_ghost_rows && pos == nboundary) XPIC_CALL(matL_ghost_rows_post(c));
    for (int a = 0; a < ncol_y; ++a) {
      int cy0, cys, ncy, cz0, czs, ncz;
      colour_class(g.ny, per_y, a, &cy0, &cys, &ncy);
      if (g.G == 0) colour_class(g.nzl, per_z, b, &cz0, &czs, &ncz);
      else { cz0 = b; czs = per_z; ncz = (g.nzl - b + per_z - 1) / per_z; }
      if (ncy == 0 || ncz == 0) continue;
      Timed t(c, "fill_current"); // one entry per colour launch: the average is the kernel's own launch duration
      const bool alias = g.ny < 3 || (g.G == 0 && g.nzl < 3);
      const bool fx = g.nx % kW == 0 && !alias;
      const unsigned short* dtab = (const unsigned short*)(c->ltab + kLines + kLines * 8);
      if (ws) {
        launch_fill_ws(c, s, (unsigned)(ncy * ncz), B, currI_sort, matL, dtab, c->ltab, c->ltab + kLines, cy0, cys, ncy, cz0, czs,
          pos * ncol_y + a, ncol_y, per_y, per_z, first_sort ? 1 : 0, zord);
        continue;
      }
      const int gak = ga ? (s.def_bucket ? 2 : 1) : 0;
#define FILLK(G) (g.pow2 ? (fx ? k_ecsim_fill<true, true, G> : k_ecsim_fill<true, false, G>) \
                         : (fx ? k_ecsim_fill<false, true, G> : k_ecsim_fill<false, false, G>))
      auto kern = gak == 2 ? FILLK(2) : (gak == 1 ? FILLK(1) : FILLK(0));
#undef FILLK
      hipLaunchKernelGGL(kern, dim3((unsigned)(ncy * ncz)), dim3(kThreads), 0, c->stream, g, s.d, B,
        currI_sort, matL, dtab, c->ltab, c->ltab + kLines, s.par.q, s.par.m,
        s.par.n / (double)s.par.Np, cy0, cys, ncy, cz0, czs, pos * ncol_y + a, ncol_y, per_y, per_z, first_sort && !alias ? 1 : 0,
        alias ? 1 : 0, zord, s.def_step, c->fill_err, store_sorted ? 1 : 0, ga && s.def_bucket ? s.d.bucket_cap : 0, c->gather_window);
    }
  }
  if (post_ghost_rows && nseq <= nboundary) XPIC_CALL(matL_ghost_rows_post(c)); // (a slab whose every colour is a boundary colour)
  if (store_sorted) sort_deferred_done(s); // every cell's records are in r2 / v2 now: they become the sort
  XPIC_HIP(hipGetLastError());
  // (what the kernels raised in c->fill_err is read by ecsim_fill_check, once per assembly and for all slabs together)
  return 0;
}

// The assembly's error word (ecsim_fill.h: kFillErr*), read ONCE behind all species' launches.  A rank that returned early
// while its neighbours sit in the next exchange would hang the job (the rule of sort_rebin's agree()): the kernels only
// raise the flag and run on, every exchange of the assembly is issued regardless, and here all slabs learn the OR of all
// flags through one small all-reduce and leave with the same return code.
int ecsim_fill_check(xpic_ctx* c)
{
  int* herr = (int*)(c->red_host + 60);
  XPIC_HIP(hipMemcpyAsync(herr, c->fill_err, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  XPIC_HIP(hipStreamSynchronize(c->stream));
  double e[2] = {(*herr & kFillErrPencil) ? 1.0 : 0.0, (*herr & kFillErrTimeout) ? 1.0 : 0.0};
  if (c->comm.kind != 0) XPIC_CALL(comm_allreduce_sum_host(c, e, 2));
  XPIC_CHECK(e[0] == 0.0, "an x-pencil holds 2^29 particles or more in the gathering assembly (sort_rebin should have scattered first)");
  XPIC_CHECK(e[1] == 0.0, "k_ecsim_fill_ws: a producer / consumer wait timed out (assembly incomplete)");
  return 0;
}

}  // namespace xpic
