// esirkepov.hip -- the particle phases that deposit the charge-conserving Esirkepov current:
//   MODE 0  basic::Particles::push            (src/impls/basic/particles.cpp:17-53)
//   MODE 1  ecsimcorr::Particles::first_push  (src/impls/ecsimcorr/particles.cpp:27-50)
//   MODE 2  ecsimcorr::Particles::second_push (src/impls/ecsimcorr/particles.cpp:52-91)
// with Shape (src/utils/shape.cpp:31-80), SimpleInterpolation (src/algorithms/simple_interpolation.cpp:8-38)
// and EsirkepovDecomposition (src/algorithms/esirkepov_decomposition.cpp:20-103) for cell-sorted SoA particles.
//
// A workgroup (4 waves) owns one x-pencil of cells and marches along it in ROUNDS.  A round takes the particles of as
// many consecutive cells as fit into its 240 stage columns (160 for MODE 1; up to 8 cells; a cell's columns are padded to a multiple
// of 4; a cell with more particles than a round holds continues in the next round) and hands them out one per thread,
// so that the waves are full whatever the number of particles per cell: the instruction stream of a wave costs the same
// for 1 or 64 live lanes, and with one wave per cell (the previous organisation) a 32-particle cell left half of them
// idle.  What a round needs from global memory -- its particles, the CIC neighbourhoods of its cells (MODE 2) -- is
// requested one round ahead; the round table itself is composed one round ahead from cell_start (scalar loads).
//   * phase 1 (thread = particle) moves / pushes the particle.  MODE 0 gathers E, B with the 2nd-order shape out of an
//     LDS tile of the 13 x 6 x 6 nodes the round's particles can reach at mid-step (only the three nodes per axis and
//     weight type that spline_of_2nd_order does not return as exact zeros, summed in the order of the reference's loop
//     over its 3..4-wide box); the tile shares the LDS of the stage, which is written after every gather is done.
//     MODE 2 gathers with the CIC weights out of the cells' 36 + 54 value neighbourhoods in LDS, like k_second_push.
//   * A particle that starts in cell c and ends less than about half a cell away (every particle of the BASELINE
//     workloads) has old and new spline supports inside the nodes c-1 .. c+2 of each axis.  For those the deposit is
//     the dense 4 x 4 x 4 box per component: phase 1 stages the 1-D old / new spline values and the prefix sums of
//     their differences on these 4 nodes (36 values per particle, one stage column).
//   * phase 2 runs on the matrix cores: per component the box of a cell is  J_c[i][u][w] += sum_p P_c[i] T_c[u][w]
//     (the reference's running sum temp_j += W, :57-103, with the sum over the line taken first), one
//     v_mfma_f64_4x4x4_4b_f64 per 4 particles (a "K step").  The K steps of the round are dealt out evenly to the four
//     waves; a wave that passes from one cell to the next adds its box into the J window.  No atomics inside a cell.
//   * A particle that moves further (up to the reference's own limit of one cell, shape.h:18,91-92) deposits its box
//     directly with fp64 atomics (slow path, same arithmetic as the reference's loop).
//   * The boxes are merged in a circular LDS J window (11 x 4 x 4 nodes per component); the columns the march has
//     passed leave with one fp64 atomic per node.
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "device_common.h"

namespace xpic {

int experiment_esirkepov() { return XPIC_TU_EXPERIMENT; }

namespace {

#ifdef ESK_STAMPS
// in-kernel section timers (experiment build only): s_memtime deltas of wave 0 summed per section over all workgroups
__device__ unsigned long long g_esk_stamps[16];
#define STAMP(k)                                                  \
  do {                                                            \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    stamp_acc_[k] += now_ - stamp_t_;                             \
    stamp_t_ = now_;                                              \
  } while (0)
#else
#define STAMP(k)
#endif
constexpr int kW = 4;                 // waves per workgroup
constexpr int kThreadsB = kW * 64;
constexpr int kSeg = 8;               // cells (segments) per round
// Columns of the stage = particles of a round, each cell's padded to a multiple of 4 (the K of the MFMA step).  A wave
// issues the same instruction stream for 1 or 64 particles, so a round packs the particles of as many cells of the
// pencil as fit: at 32 particles per cell about 7 cells = 224 of the 256 lanes, where one wave per cell filled 32 of 64.
// MODE 1 (the lightest phase 1) does better with 160 columns and three workgroups per CU (6.7 ms against 7.8 with 240).
#ifndef ESK_COLS1
#define ESK_COLS1 160
#endif
#ifndef ESK_EXP
#define ESK_EXP 0
#endif
#ifndef ESK_SEG0
#define ESK_SEG0 16
#endif
#ifndef ESK_LANES
#define ESK_LANES 2 // bit MODE: the round is composed on 8 / 16 lanes (row shifts, ballots) instead of by the scalar loop
#endif
#ifndef ESK_OCC1
#define ESK_OCC1 2 // workgroups per CU the register allocation of MODE 1 must allow
#endif
#ifndef ESK_DRAIN0
#define ESK_DRAIN0 1
#endif
#ifndef ESK_TILE_LATE
#define ESK_TILE_LATE 1
#endif
#ifndef ESK_PIPE
#define ESK_PIPE 6 // bit MODE: phase 2 requests the operands of the next K step before this step's products (MODE 1, 2: -2 %; MODE 0 spills with it)
#endif
#ifndef ESK_FLUSH
#define ESK_FLUSH 1 // the J window's finished nodes leave 1: in aligned groups of 8 (one 64-byte atomic request per row), 0: as each round completes them
#endif
#ifndef ESK_BOX_GENERIC
#define ESK_BOX_GENERIC 0 // 1: the staged box values by four calls of the generic spline per axis and position (round 2)
#endif
template <int MODE> struct StageDim {
  static constexpr int kSegM = MODE == 0 ? ESK_SEG0 : 8; // cells (segments) per round
  static constexpr int kCols = MODE == 1 ? ESK_COLS1 : 240;
  static constexpr int kPitch = kCols + 4; // row pitch: 4 rows x 4 columns of a phase-2 read fall in 16 distinct bank pairs (pitch = 4 mod 8)
  static_assert(kCols % 4 == 0 && kCols <= kThreadsB && kPitch % 8 == 4, "stage geometry");
};
constexpr int kD = 4;                 // deposit box per axis: nodes c-1 .. c+2
// (J window: kSeg + kD - 1 nodes along x; gather tile: kSeg + kT - 1; both sized per MODE inside the kernel)
constexpr int kSRows = 36;            // So[3][4], Sn[3][4], P[3][4]
constexpr int kT = 6;                 // gather tile per axis (MODE 0): nodes c-2 .. c+3
constexpr int kLinesB = 3 * kD * kD;  // 48 lines of 4 nodes
constexpr int kNb = 36 + 54;          // CIC neighbourhood of a cell (MODE 2): 3 x 12 E nodes, 54 B nodes
constexpr int kNbPer = (kSeg * kNb + kThreadsB - 1) / kThreadsB;
static_assert(kSeg <= kCellStartPad, "compose reads kSeg entries ahead");

// raw workgroup barrier that only drains LDS traffic: the particle stores, the J atomics and the requests of the next
// round stay in flight across it (__syncthreads() would wait for every one of them: a round trip to HBM per round)
__device__ inline void lds_barrier_b()
{
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ inline void wave_sync_b()
{
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// spline_of_2nd_order (src/interfaces/sort_parameters.cpp:21-30): 0.75 - s^2 for |s| <= 0.5, 0.5 (1.5 - |s|)^2 for
// 0.5 < |s| < 1.5, else 0 -- the same expressions; the outer zero comes out of the clamp (0.5 * 0 * 0) instead of a
// third branch
__device__ inline double spline2(double s)
{
  s = fabs(s);
  const double t = 1.5 - fmin(s, 1.5);
  return s <= 0.5 ? (0.75 - s * s) : 0.5 * t * t;
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

// One round of a pencil: up to kSeg segments, each the (rest of the) particles of one cell.  A cell with more particles
// than a round holds is continued in the next round (then alone in its round's first segment).
constexpr int kSegMax = 16;
struct RoundTab { // (word order = the order tab_put's threads unpack a RoundPack into)
  int base;            // first cell of the round: the J window's column 0 is node base - 1
  int pstart;          // first particle of the round: its particles are consecutive in the sort, thread t takes pstart + t
  int nseg, ncols, tcount;
  int adv;             // cells completed by the round: the window moves on by this many columns
  int cell[kSegMax];      // cell - base
  int col0[kSegMax + 1];  // first stage column (multiple of 4)
  int toff[kSegMax + 1];  // first thread
};
constexpr int kTabWords = 6 + kSegMax + 2 * (kSegMax + 1);
static_assert(sizeof(RoundTab) == 4 * kTabWords && kTabWords <= 64, "one lane of wave 0 per word");

// The rounds of a pencil depend on cell_start alone, and composing them inside the push was a fifth of its instruction
// stream (a scalar loop over up to 16 cells, or row shifts and ballots on 8 lanes, once per round and wave).  PRE: a small
// kernel composes the rounds of every pencil beforehand (k_esk_rounds: one thread per pencil, the same greedy rule) into
// 64-byte records; the push copies record r + 2 into LDS while round r runs.
//   bytes 0-3 base, 4-7 pstart, 8 nseg, 9 ncols, 10 tcount, 11 adv, 12-19 cell[16] (4 bits each), 20-36 col0[17], 37-53 toff[17]
constexpr int kPackDwords = 16;
constexpr int kPackCell = 12, kPackCol0 = 20, kPackToff = 37;
constexpr int kSentinels = 3; // records behind a pencil's last round (the push reads two records ahead)

template <int SEG, int COLS>
__global__ void __launch_bounds__(256) k_esk_rounds(GridDev g, const int* __restrict__ cell_start, unsigned* __restrict__ tabg,
  int rcap, long npencil, int* overflow)
{
  const long pencil = (long)blockIdx.x * 256 + threadIdx.x;
  if (pencil >= npencil) return;
  const int* cs = cell_start + pencil * g.nx;
  uint4* out = reinterpret_cast<uint4*>(tabg + pencil * (long)rcap * kPackDwords);
  int cur = 0, cur_off = 0, sent = 0;
  for (int r = 0; sent < kSentinels; ++r) {
    if (r >= rcap) { atomicOr(overflow, 1); return; } // (cells of several rounds each: the push composes for itself)
    // compose_scalar's rule (below), word for word
    const int base = cur;
    unsigned w[kPackDwords];
#pragma unroll
    for (int k = 0; k < kPackDwords; ++k) w[k] = 0u;
    auto put = [&](int byte, unsigned v) { w[byte >> 2] |= v << ((byte & 3) * 8); };
    int nseg = 0, cols = 0, tc = 0, pstart = 0;
    bool open = true;
#pragma unroll
    for (int i = 0; i < SEG; ++i) {
      if (open && base + i < g.nx) {
        const int c0 = cs[base + i] + (i == 0 ? cur_off : 0);
        const int rem = cs[base + i + 1] - c0;
        if (rem <= 0) { cur = base + i + 1; cur_off = 0; }
        else {
          const int room = COLS - cols;
          int take = rem;
          bool fits = true;
          if (((rem + 3) & ~3) > room) {
            if (nseg > 0) fits = false;
            else take = room;
          }
          if (!fits) open = false;
          else {
            if (nseg == 0) pstart = c0;
#pragma unroll
            for (int k = 0; k < SEG; ++k) // (nseg as a compile-time index: no scratch)
              if (k == nseg) {
                w[(kPackCell + k / 2) >> 2] |= (unsigned)i << (((kPackCell + k / 2) & 3) * 8 + (k & 1) * 4);
                put(kPackCol0 + k, (unsigned)cols);
                put(kPackToff + k, (unsigned)tc);
              }
            ++nseg;
            cols += (take + 3) & ~3;
            tc += take;
            if (take == rem) { cur = base + i + 1; cur_off = 0; }
            else { cur_off = (i == 0 ? cur_off : 0) + take; cur = base + i; open = false; }
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k <= SEG; ++k)
      if (k == nseg) { put(kPackCol0 + k, (unsigned)cols); put(kPackToff + k, (unsigned)tc); }
    w[0] = (unsigned)base; w[1] = (unsigned)pstart;
    w[2] = (unsigned)nseg | (unsigned)cols << 8 | (unsigned)tc << 16 | (unsigned)(cur - base) << 24;
#pragma unroll
    for (int k = 0; k < kPackDwords / 4; ++k) out[(long)r * (kPackDwords / 4) + k] = uint4{w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]};
    if (base >= g.nx) ++sent;
  }
}

// what a thread requests one round ahead: its particle of the next round (and, MODE 2, its share of the cells' CIC
// neighbourhoods)
template <int MODE>
struct Ahead {
  int p, col, crel, seg; // particle (-1: none), stage column, cell - base, segment
  int idx;               // GA: where the record of slot p lies in the old order
  double r[3], v[3];
  double nb[MODE == 2 ? kNbPer : 1];
};

using UniformInts = const __attribute__((address_space(4))) int*; // wave-uniform reads of cell_start: scalar loads

// P2: power-of-two spacings (exact reciprocals instead of divisions, device_common.h: scaled_position)
// GA ("gather"): the re-binning in front of this push deferred its scatter (sort_rebin(.., defer = 2)): slot p of the new
// order finds its record at src[p] of the old one, still un-wrapped; the push reads it from there, wraps it with k_scatter's
// arithmetic and writes position AND velocity to slot p of the second buffer, which becomes the sort (the push reads
// and writes every particle anyway: the scatter pass, 104 B per particle, is gone).  MODE 0 with the round table only.
template <int MODE, bool P2, bool PRE, bool GA = false>
__global__ void __launch_bounds__(kThreadsB, MODE == 1 ? ESK_OCC1 : 2) k_esirkepov_push(GridDev g, SortDev s, const double* __restrict__ E,
  const double* __restrict__ B, double* __restrict__ J, double qm, double alpha, double qn_Np, double* pred_w,
  int* bad_count, const unsigned* __restrict__ tabg, int rcap)
{
  // PRE: a pencil had more rounds than the table holds (k_esk_rounds raised the flag behind the error count): nothing is
  // touched, the host launches the self-composing form instead
  static_assert(!GA || (MODE == 0 && PRE), "the gathering form exists for the basic push with precomposed rounds");
  if (PRE && bad_count[1] != 0) return;
  // workgroup -> the x-pencil (cy, cz), marched in rounds
  constexpr int kCols = StageDim<MODE>::kCols, kPitch = StageDim<MODE>::kPitch;
  // cells per round.  MODE 0 takes up to 16: BASELINE configs[1] is two species of 16 ppc, and a round of 8 such cells
  // fills half of the 256 lanes (the instruction stream of a round costs the same); the names below shadow the
  // namespace-scope values that MODE 1 and 2 keep.
  constexpr int kSeg = StageDim<MODE>::kSegM;
  // J window: the kSeg + kD - 1 nodes a round's boxes reach + up to kJG - 1 finished nodes that wait for their aligned group
  // of kJG to be complete (the flush at the end of a round)
  // (MODE 1 runs three workgroups per CU and has 832 bytes of LDS to spare for the pending columns, MODE 2 1 160: groups of
  // 4 there; a stage of 232 instead of 240 columns to make room for groups of 8 cost MODE 2 more than they gave)
  constexpr int kJGs = ESK_FLUSH ? (MODE == 0 ? 3 : 2) : 0, kJG = 1 << kJGs, kJX = kSeg + kD - 1 + kJG - 1, kJN = kJX * kD * kD;
  constexpr int kTX = kSeg + kT - 1, kTileN = kTX * kT * kT, kFtPer = (6 * kTileN + kThreadsB - 1) / kThreadsB;
  static_assert(kSeg <= kCellStartPad && kSeg <= kSegMax, "compose reads kSeg entries ahead");
  static_assert(MODE != 0 || 6 * kTileN <= kSRows * kPitch, "the gather tile shares the stage's LDS");
  const int cy = blockIdx.x % g.ny;
  const int cz = blockIdx.x / g.ny;
  // the wave index is wave-uniform: said explicitly, the K-step ranges and segment walks of phase 2 become scalar code
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;

  __shared__ double stage[kSRows * kPitch];
  __shared__ double jtile[3 * kJN];
  __shared__ double nbv[MODE == 2 ? kSeg * kNb : 1]; // CIC neighbourhoods of the round's cells (second_push only)
  __shared__ double pwsum[kW];
  __shared__ RoundTab tab[2];
  double* const ftile = stage; // Ex,Ey,Ez,Bx,By,Bz on the gather tile (basic only): dead before the first stage write

  const long pencil0 = ((long)cz * g.ny + cy) * g.nx;
  UniformInts cs = (UniformInts)(s.cell_start + pencil0);
  for (int t = threadIdx.x; t < 3 * kJN; t += kThreadsB) jtile[t] = 0.0;

  // this thread's entries e = thread + i * kThreadsB of the kSeg x 90 neighbourhood table (MODE 2): segment, field
  // component and node offset, numbered as in k_second_push; packed seg | comp << 4 | isB << 6 | (d + 1) << 8, 10, 12
  int nbd[MODE == 2 ? kNbPer : 1];
  const double* nbrow[MODE == 2 ? kNbPer : 1];
  if (MODE == 2) {
#pragma unroll
    for (int i = 0; i < kNbPer; ++i) {
      const int e = threadIdx.x + i * kThreadsB;
      const int sg = e / kNb, k = e % kNb;
      int comp, isB, d0, d1, d2;
      if (k < 36) {
        isB = 0;
        comp = k / 12;
        const int l = k % 12;
        if (comp == 0) { d0 = l % 3 - 1; d1 = (l / 3) % 2; d2 = l / 6; }
        else if (comp == 1) { d0 = l % 2; d1 = (l / 2) % 3 - 1; d2 = l / 6; }
        else { d0 = l % 2; d1 = (l / 2) % 2; d2 = l / 4 - 1; }
      }
      else {
        isB = 1;
        const int kb = k - 36;
        if (kb < 18) { comp = 0; d0 = kb % 2; d1 = (kb / 2) % 3 - 1; d2 = kb / 6 - 1; }
        else if (kb < 36) { const int l = kb - 18; comp = 1; d0 = l % 3 - 1; d1 = (l / 3) % 2; d2 = l / 6 - 1; }
        else { const int l = kb - 36; comp = 2; d0 = l % 3 - 1; d1 = (l / 3) % 3 - 1; d2 = l / 9; }
      }
      nbd[i] = e < kSeg * kNb ? (sg | comp << 4 | isB << 6 | (d0 + 1) << 8 | (d1 + 1) << 10 | (d2 + 1) << 12) : -1;
      // the row of E / B the entry lies in is fixed for the pencil: only its x position moves with the round
      nbrow[i] = e < kSeg * kNb ? (isB ? B : E) + comp * g.cstride + g.node(0, g.wy(cy + d1), g.wz(cz + d2)) : nullptr;
    }
  }

  // (the J window is circular: see the flush at the end of a round)
  // ---- the next round's composition: whole cells while their padded columns fit (every thread runs the same scalar
  // loop; thread 0 records it)
  int cur = 0, cur_off = 0;
  auto compose_scalar = [&](RoundTab& T) {
    const int base = cur;
    // the cell_start entries the round can need, requested together (one scalar-cache latency instead of one per cell)
    int cv[kSeg + 1];
#pragma unroll
    for (int i = 0; i <= kSeg; ++i) cv[i] = cs[base + i]; // entries past the pencil's end are read (kCellStartPad) but not used
    int nseg = 0, cols = 0, tc = 0;
    bool open = true;
#pragma unroll
    for (int i = 0; i < kSeg; ++i) {
      if (open && base + i < g.nx) {
        const int c0 = cv[i] + (i == 0 ? cur_off : 0);
        const int rem = cv[i + 1] - c0;
        if (rem <= 0) { cur = base + i + 1; cur_off = 0; }
        else {
          const int room = kCols - cols;
          int take = rem;
          bool fits = true;
          if (((rem + 3) & ~3) > room) {
            if (nseg > 0) fits = false;
            else take = room; // a cell of more than kCols particles: a full round of it
          }
          if (!fits) open = false;
          else {
            if (threadIdx.x == 0) { T.cell[nseg] = i; T.col0[nseg] = cols; T.toff[nseg] = tc; if (nseg == 0) T.pstart = c0; }
            ++nseg;
            cols += (take + 3) & ~3;
            tc += take;
            if (take == rem) { cur = base + i + 1; cur_off = 0; }
            else { cur_off = (i == 0 ? cur_off : 0) + take; cur = base + i; open = false; }
          }
        }
      }
    }
    if (threadIdx.x == 0) {
      T.base = base; T.nseg = nseg; T.ncols = cols; T.tcount = tc; T.adv = cur - base;
      T.col0[nseg] = cols; T.toff[nseg] = tc;
    }
  };

  // The same on 8 lanes: lane i < kSeg of every wave looks at cell base + i (all waves keep the same cur / cur_off), row
  // shifts give the running column and thread counts, ballots the cells that still fit, the lanes of wave 0 record the
  // segments.  A fifth of the scalar loop's instructions; pays where registers are not the limit (MODE 1: first_push
  // 6.75 -> 6.1 ms; MODE 0 and 2, at their register limit, lose 3 % with it and keep the scalar loop).
  auto compose_lanes = [&](RoundTab& T) {
    // (said to be uniform: the cell_start reads below are then scalar loads; as vector loads they brought a
    // s_waitcnt vmcnt(0) -- a wait for the previous round's particle stores and J atomics -- to the top of every round)
    const int base = __builtin_amdgcn_readfirstlane(cur);
    int cv[kSeg + 1];
#pragma unroll
    for (int i = 0; i <= kSeg; ++i) cv[i] = cs[base + i];
    int first_p = 0, cnt = 0; // this lane's cell: first particle not yet handed out, particles left
#pragma unroll
    for (int i = 0; i < kSeg; ++i)
      if (lane == i) { first_p = cv[i] + (i == 0 ? cur_off : 0); cnt = cv[i + 1] - first_p; }
    if (lane >= kSeg || base + lane >= g.nx || cnt < 0) cnt = 0;
    const int pad = (cnt + 3) & ~3;
    int P = pad, Q = cnt; // inclusive running sums over the lanes 0 .. kSeg - 1 (one DPP row; zeros are shifted in)
    P += __builtin_amdgcn_update_dpp(0, P, 0x111, 0xf, 0xf, true); Q += __builtin_amdgcn_update_dpp(0, Q, 0x111, 0xf, 0xf, true);
    P += __builtin_amdgcn_update_dpp(0, P, 0x112, 0xf, 0xf, true); Q += __builtin_amdgcn_update_dpp(0, Q, 0x112, 0xf, 0xf, true);
    P += __builtin_amdgcn_update_dpp(0, P, 0x114, 0xf, 0xf, true); Q += __builtin_amdgcn_update_dpp(0, Q, 0x114, 0xf, 0xf, true);
    if constexpr (kSeg > 8) { // sixteen lanes: one more shift (a DPP row is 16 lanes)
      P += __builtin_amdgcn_update_dpp(0, P, 0x118, 0xf, 0xf, true); Q += __builtin_amdgcn_update_dpp(0, Q, 0x118, 0xf, 0xf, true);
    }
    static_assert(kSeg == 8 || kSeg == 16, "three / four row shifts cover eight / sixteen lanes");
    const unsigned some = (unsigned)__ballot(cnt > 0) & ((1u << kSeg) - 1u);
    const int first = some ? __ffs(some) - 1 : kSeg;
    int nseg, ncols, tcount;
    if (some && ((__builtin_amdgcn_readlane(cnt, first & (kSeg - 1)) + 3) & ~3) > kCols) {
      // a cell of more than kCols particles: a full round of it, alone
      nseg = 1; ncols = kCols; tcount = kCols;
      if (wave == 0 && lane == first) { T.cell[0] = first; T.pstart = first_p; T.col0[0] = 0; T.toff[0] = 0; }
      cur_off = (first == 0 ? cur_off : 0) + kCols;
      cur = base + first;
    }
    else {
      const unsigned over = (unsigned)__ballot(P > kCols) & ((1u << kSeg) - 1u); // a prefix property: once over, always over
      const int stop = over ? __ffs(over) - 1 : kSeg;                               // first cell that does not fit
      const unsigned segs = some & ((1u << stop) - 1u);
      nseg = __popc(segs);
      ncols = stop > 0 ? __builtin_amdgcn_readlane(P, (stop - 1) & (kSeg - 1)) : 0;
      tcount = stop > 0 ? __builtin_amdgcn_readlane(Q, (stop - 1) & (kSeg - 1)) : 0;
      if (wave == 0 && lane < stop && cnt > 0) {
        const int k = __popc(segs & ((1u << lane) - 1u));
        T.cell[k] = lane; T.col0[k] = P - pad; T.toff[k] = Q - cnt;
        if (k == 0) T.pstart = first_p;
      }
      cur = min(base + stop, g.nx);
      cur_off = 0;
    }
    if (threadIdx.x == 0) {
      T.base = base; T.nseg = nseg; T.ncols = ncols; T.tcount = tcount; T.adv = cur - base;
      T.col0[nseg] = ncols; T.toff[nseg] = tcount;
    }
  };
  auto compose = [&](RoundTab& T) {
    if constexpr ((ESK_LANES >> MODE) & 1) compose_lanes(T);
    else compose_scalar(T);
  };
  // PRE: thread t < kTabWords owns word t of the table: where it sits in a 64-byte record (dword, shift, mask)
  int twd = -1, tsh = 0;
  unsigned tmask = 0u;
  if (PRE && (int)threadIdx.x < kTabWords) {
    const int t = threadIdx.x;
    int byte, nib = 0;
    if (t < 2) { byte = 4 * t; tmask = 0xffffffffu; }
    else if (t < 6) { byte = 8 + (t - 2); tmask = 0xffu; }
    else if (t < 6 + kSegMax) { byte = kPackCell + (t - 6) / 2; nib = ((t - 6) & 1) * 4; tmask = 0xfu; }
    else if (t < 6 + kSegMax + kSegMax + 1) { byte = kPackCol0 + (t - 6 - kSegMax); tmask = 0xffu; }
    else { byte = kPackToff + (t - 6 - kSegMax - (kSegMax + 1)); tmask = 0xffu; }
    twd = byte >> 2;
    tsh = t < 2 ? 0 : (byte & 3) * 8 + nib;
  }
  const unsigned* const tg = PRE ? tabg + (long)blockIdx.x * rcap * kPackDwords : nullptr;
  auto tab_load = [&](int r) { return twd >= 0 ? tg[(long)r * kPackDwords + twd] : 0u; };
  auto tab_put = [&](RoundTab& T, unsigned w) {
    if (twd >= 0) reinterpret_cast<int*>(&T)[threadIdx.x] = (int)((w >> tsh) & tmask);
  };

  auto request = [&](const RoundTab& T, Ahead<MODE>& pf) {
    pf.p = -1; pf.col = 0; pf.crel = 0; pf.seg = 0; pf.idx = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) { pf.r[a] = 0.0; pf.v[a] = 0.0; }
    const int nseg = T.nseg;
    if ((int)threadIdx.x < T.tcount) {
      int sg = 0;
#pragma unroll
      for (int i = 1; i < kSeg; ++i) sg += (i < nseg && (int)threadIdx.x >= T.toff[i]) ? 1 : 0;
      const int k = (int)threadIdx.x - T.toff[sg];
      pf.seg = sg; pf.p = T.pstart + (int)threadIdx.x; pf.col = T.col0[sg] + k; pf.crel = T.cell[sg];
      if (GA) pf.idx = s.src[pf.p]; // (a round ahead of the record it addresses)
      // MODE 0 (the register-hungry 2nd-order gather) loads its particle when it gets there instead
      if (MODE != 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { pf.r[a] = s.r[a][pf.p]; pf.v[a] = s.v[a][pf.p]; }
      }
    }
    if (MODE == 2) {
      const int base = T.base;
#pragma unroll
      for (int i = 0; i < kNbPer; ++i) {
        pf.nb[i] = 0.0;
        const int d = nbd[i];
        if (d >= 0 && (d & 15) < nseg) {
          const int cx = base + T.cell[d & 15];
          pf.nb[i] = nbrow[i][g.wx(cx + ((d >> 8) & 3) - 1)];
        }
      }
    }
  };

  double pw = 0.0;
  int bad = 0;
  int jflushed = -kJG; // J window: nodes below this one have left (a multiple of kJG; the window starts at node -1)
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

  if constexpr (PRE) {
    const unsigned w0 = tab_load(0), w1 = tab_load(1);
    tab_put(tab[0], w0);
    tab_put(tab[1], w1);
  }
  else compose(tab[0]);
  __syncthreads();
  Ahead<MODE> pf;
  request(tab[0], pf);
#if ESK_DRAIN0
  // The first round's particles are waited for HERE.  Left pending into the loop they make the compiler place the waits
  // of the first iteration (vmcnt(4), (2), (0) at the first uses) into every iteration, where -- the request of the
  // next round being conditional, hence not counted -- they wait for the loads issued a moment ago: the prefetch of a
  // round ahead was a full memory round trip at the start of every phase 1.
  __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0), lgkmcnt / expcnt untouched
#endif
#ifdef ESK_STAMPS
  unsigned long long stamp_t_ = __builtin_readcyclecounter();
  unsigned long long stamp_acc_[10] = {};
#endif
  for (int rd = 0;; ++rd) {
    STAMP(0);
    const RoundTab& T = tab[rd & 1];
    const int base = __builtin_amdgcn_readfirstlane(T.base);
    if (base >= g.nx) break;
    const int nseg = __builtin_amdgcn_readfirstlane(T.nseg), ncols = __builtin_amdgcn_readfirstlane(T.ncols);
    const int adv = __builtin_amdgcn_readfirstlane(T.adv);
    // PRE: the record of round rd + 2 travels during this round and goes to LDS behind phase 2, into this round's table
    unsigned tw = 0u;
    if constexpr (PRE) tw = tab_load(rd + 2);
    // ---- this round's neighbourhoods / tile go to LDS; the next round is composed and requested
    double ft[MODE == 0 ? kFtPer : 1];
    if (MODE == 0) {
      // DMGlobalToLocal(E), (B) (basic/simulation.cpp:56-57) for just the nodes the round's particles can gather from
      // (requesting the tile a round ahead was measured: no gain, the other workgroup of the CU covers the latency);
      // the values go to LDS after the next round is composed: the scalar loop runs under the loads' latency
#pragma unroll
      for (int k = 0; k < kFtPer; ++k) {
        const int t = threadIdx.x + k * kThreadsB;
        ft[k] = 0.0;
        if (t < 6 * kTileN && nseg > 0) {
          const int f = t / kTileN, n = t % kTileN;
          const int tx = n % kTX, ty = (n / kTX) % kT, tz = n / (kTX * kT);
          const double* F = (f < 3 ? E : B) + (f % 3) * g.cstride;
          // (a particle of the pencil's last cell reaches node nx + 2 at most: nothing beyond is gathered from, and
          // nodew folds an index only once)
          if (base - 2 + tx <= g.nx + 2) ft[k] = F[g.nodew(base - 2 + tx, cy - 2 + ty, cz - 2 + tz)];
        }
      }
    }
    if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < kNbPer; ++i) {
        const int e = threadIdx.x + i * kThreadsB;
        if (e < kSeg * kNb) nbv[e] = pf.nb[i];
      }
    }
    const int p = pf.p, col = pf.col, crel = pf.crel, seg = pf.seg;
    double r[3] = {pf.r[0], pf.r[1], pf.r[2]}, v[3] = {pf.v[0], pf.v[1], pf.v[2]};
    if (MODE == 0 && p >= 0) {
      // (MODE 0 cannot hold a particle a round ahead -- the 2nd-order gather takes every register -- but it can ask for
      // it here: the composition of the next round and the barrier pass under the latency)
      const long q = GA ? (long)pf.idx : (long)p;
#pragma unroll
      for (int a = 0; a < 3; ++a) { r[a] = s.r[a][q]; v[a] = s.v[a][q]; }
    }
    STAMP(1);
    if constexpr (!PRE) compose(tab[(rd + 1) & 1]);
    if (MODE == 0) {
#if ESK_TILE_LATE
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int k = 0; k < kFtPer; ++k) {
        const int t = threadIdx.x + k * kThreadsB;
        if (t < 6 * kTileN) ftile[t] = ft[k];
      }
    }
    STAMP(2);
    lds_barrier_b();
    STAMP(3);
    request(tab[(rd + 1) & 1], pf);
    STAMP(4);

    // ---- phase 1: thread = particle
    const int cc[3] = {base + crel, cy, cz + g.z0};
    // a particle whose old / new supports leave the cell's 4-node box: deposited by the whole wave further down
    bool slow = false, fast = false;
    double po[3] = {0, 0, 0}, pn[3] = {0, 0, 0};
    int sst[3] = {0, 0, 0}, ssz[3] = {0, 0, 0};
    if (p >= 0) {
      if (GA) { // correct_coordinates of the deferred re-binning (k_scatter<false, true>'s arithmetic); here, at the record's
        // first use: directly behind the loads it made the wave sit out their whole latency at the top of every round
        // (the fold leaves a coordinate inside [0, L] as it is: skipped where no lane of the wave left the box)
        const bool out = r[0] < 0.0 || r[0] > g.Lx || r[1] < 0.0 || r[1] > g.Ly || r[2] < 0.0 || r[2] > g.Lz;
        if (__builtin_expect(__ballot(out) != 0, 0)) {
          r[0] = bound_periodic_sel(r[0], g.Lx);
          r[1] = bound_periodic_sel(r[1], g.Ly);
          r[2] = bound_periodic_sel(r[2], g.Lz);
        }
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
          const int gst = (int)round(pr - 1.5);
          const int gsz = (int)floor(pr + 1.5) + 1 - gst;
          inside = inside && gst >= cc[a] - 2 && gst + gsz <= cc[a] + 4;
          nN[a] = (pr - (double)gst) < 1.5 ? 0 : 1; // node gst has |pr - gst| < 1.5, or node gst + 3 may have
          off[a] = gst - (cc[a] - 2);
          // Shape::fill (:57-80) on the three nodes of each support.  They are (left, centre, right) of the particle with
          // distances s in [0.5, 1.5], [-0.5, 0.5], [-1.5, -0.5]: spline_of_2nd_order's branch is known per node, and
          // the differences that form its arguments are exact (Sterbenz) away from the cells next to the origin (there
          // they agree to an ulp): 0.5 (1.5 - |s|)^2 and 0.75 - s^2 written out per node replace three calls of the
          // generic function (fabs, clamp, compare, select: 9 instructions each) by 10 operations
          const double sN = pr - (double)(gst + nN[a]), sS = pr - ((double)gst + 0.5);
          const double tNl = 1.5 - sN, tNr = 1.5 + (sN - 2.0), tSl = 1.5 - sS, tSr = 1.5 + (sS - 2.0);
          No[a][0] = 0.5 * tNl * tNl; No[a][1] = 0.75 - (sN - 1.0) * (sN - 1.0); No[a][2] = 0.5 * tNr * tNr;
          Sh[a][0] = 0.5 * tSl * tSl; Sh[a][1] = 0.75 - (sS - 1.0) * (sS - 1.0); Sh[a][2] = 0.5 * tSr * tSr;
        }
        if (inside) {
          off[0] += crel; // tile x origin is base - 2, the cell's own is cx - 2
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
              // the 18 tile values of this (z, y) row pair are requested together and waited for once: left to
              // itself the compiler waits out the LDS latency after every single read
              double f[6][3];
#pragma unroll
              for (int ix = 0; ix < 3; ++ix) {
                // Shape::electric / magnetic (shape.h:54-72): each component has its own three nodes per axis
                const int xN = bN[0] + ix, xS = off[0] + ix;
                f[0][ix] = ftile[0 * kTileN + (zN + yN) * kTX + xS];
                f[1][ix] = ftile[1 * kTileN + (zN + yS) * kTX + xN];
                f[2][ix] = ftile[2 * kTileN + (zS + yN) * kTX + xN];
                f[3][ix] = ftile[3 * kTileN + (zS + yS) * kTX + xN];
                f[4][ix] = ftile[4 * kTileN + (zS + yN) * kTX + xS];
                f[5][ix] = ftile[5 * kTileN + (zN + yS) * kTX + xS];
              }
              __builtin_amdgcn_sched_barrier(0);
              // the x sums first, then one product with the row's (z, y) weight: 28 instead of 40 operations per row
              // pair (the reference multiplies the three 1-D weights per node, shape.h:54-72: same terms, another
              // association -- within the 1e-13 the velocities are held to)
              double tx[6];
#pragma unroll
              for (int q = 0; q < 6; ++q) {
                const double (&wx)[3] = (q == 0 || q >= 4) ? Sh[0] : No[0];
                tx[q] = f[q][0] * wx[0] + f[q][1] * wx[1] + f[q][2] * wx[2];
              }
              Ep[0] += nn * tx[0];
              Ep[1] += ns * tx[1];
              Ep[2] += sn * tx[2];
              Bp[0] += ss * tx[3];
              Bp[1] += sn * tx[4];
              Bp[2] += ns * tx[5];
              __builtin_amdgcn_sched_barrier(0);
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
        const double* eE = nbv + (MODE == 2 ? seg * kNb : 0);
        const double* eB = eE + 36;
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
#if ESK_EXP != 2
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        if (GA) { s.r2[a][p] = r[a]; s.v2[a][p] = v[a]; }
        else {
          s.r[a][p] = r[a];
          if (MODE != 1) s.v[a][p] = v[a];
        }
      }
#endif

      // Shape::setup(old_r, new_r) (shape.cpp:43-54): the box [sst, send) of the pair per axis
      bool ok = true;
      fast = true;
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
      fast = fast && ok;
      if (ok && !fast) slow = true;
      if (!ok) ++bad; // the reference would overflow Shape::shape here; deposit nothing and report
    }
    STAMP(5);
    if (MODE == 0) lds_barrier_b(); // every gather out of the tile is done: its LDS becomes the stage

    if (p >= 0) {
      double* colp = stage + col;
      if (fast) {
        // old / new spline values on the nodes c-1 .. c+2 (exact zeros outside the support, as the reference's loop
        // sees them) and the prefix sums of their differences along each axis
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#if ESK_BOX_GENERIC
          double run = 0.0;
#pragma unroll
          for (int t = 0; t < kD; ++t) {
            const double gx = (double)(cc[a] - 1 + t);
            const double so = spline2(po[a] - gx), sn = spline2(pn[a] - gx);
            run += sn - so;
            colp[(a * kD + t) * kPitch] = so;
            colp[(12 + a * kD + t) * kPitch] = sn;
            colp[(24 + a * kD + t) * kPitch] = run;
          }
#else
          // The support of a position p is its nearest node n and the two beside it; on the fast path n is c or c + 1
          // for the old AND the new position (sst >= c - 1 and send <= c + 3 say exactly that).  With d = p - n the
          // reference's spline(p - g) on those three nodes is 0.5 (0.5 - d)^2, 0.75 - d^2, 0.5 (0.5 + d)^2 -- its own
          // expressions, the arguments 1.5 - |p - (n -+ 1)| and 0.5 -+ d being the same real number rounded once
          // (p - n and p - (n -+ 1) are exact away from the origin's cells) -- and an exact zero on the fourth node:
          // 10 operations + the placement instead of four calls of the generic function (36).  At d = +-0.5 either
          // choice of n gives the same four values.
          double w[2][kD];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const double p = q == 0 ? po[a] : pn[a];
            const double nr = rint(p), d = p - nr;
            const double ta = 0.5 - d, tb = 0.5 + d;
            const double wm = 0.5 * ta * ta, w0 = 0.75 - d * d, wp = 0.5 * tb * tb;
            const bool up = nr > (double)cc[a]; // n = c + 1: the box's nodes 1, 2, 3; else 0, 1, 2
            w[q][0] = up ? 0.0 : wm; w[q][1] = up ? wm : w0; w[q][2] = up ? w0 : wp; w[q][3] = up ? wp : 0.0;
          }
          double run = 0.0;
#pragma unroll
          for (int t = 0; t < kD; ++t) {
            run += w[1][t] - w[0][t];
            colp[(a * kD + t) * kPitch] = w[0][t];
#if ESK_EXP != 3 // (experiment 3: a third of the stage stores dropped -- what the LDS store path costs; results garbage)
            colp[(12 + a * kD + t) * kPitch] = w[1][t];
#endif
            colp[(24 + a * kD + t) * kPitch] = run;
          }
#endif
        }
      }
      else {
#pragma unroll
        for (int e = 0; e < kSRows; ++e) colp[e * kPitch] = 0.0; // nothing for phase 2
      }
    }
    if ((int)threadIdx.x < 3 * nseg) {
      // phase 2 works on K = 4 particles per step: the columns that fill up a cell's last step are particles of weight zero
      const int sg = threadIdx.x / 3;
      const int c = T.col0[sg] + (T.toff[sg + 1] - T.toff[sg]) + (int)threadIdx.x % 3;
      if (c < T.col0[sg + 1]) {
#pragma unroll
        for (int e = 0; e < kSRows; ++e) stage[e * kPitch + c] = 0.0;
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
        const double Tl = sA_n * (2.0 * sB_n + sB_o) + sA_o * (2.0 * sB_o + sB_n);
        double run = 0.0;
        for (int t = 0; t < qz[lc]; ++t) {
          const double gC = (double)(qs[lc] + t);
          run = run + (-qd * (spline2(qn[lc] - gC) - spline2(qo[lc] - gC)) * Tl);
          int n[3];
          n[lc] = qs[lc] + t; n[axA] = qs[axA] + liA; n[axB] = qs[axB] + liB;
          if (run != 0.0) unsafeAtomicAdd(&J[lc * g.cstride + g.nodew(n[0], n[1], n[2] - g.z0)], run);
        }
      }
    }
    STAMP(6);
    lds_barrier_b();
    STAMP(7);

    // ---- phase 2 on the matrix cores: J_c[i][u][w] += sum_p P_c[i] * T_c[u][w],
    //   T_c[u][w] = -qd_c (Sn_A[u] (2 Sn_B[w] + So_B[w]) + So_A[u] (2 So_B[w] + Sn_B[w]))
    // is, per component, a 4 x 16 x P product: one v_mfma_f64_4x4x4_4b_f64 per K = 4 particles with block b = w, rows
    // i, columns u (lane roles A[b][i][k], B[b][k][j] at lane 16 k + 4 b + (i or j); D[b][i][j] at lane 16 i + 4 b + j).
    // The lane forms its own T from the staged 1-D spline values of particle k.  The round's K steps are dealt out
    // evenly to the waves; a wave that passes from one cell to the next merges its box into the J window.
    {
      const int K = ncols >> 2;
      const int k0 = (K * wave) >> 2, k1 = (K * (wave + 1)) >> 2;
      if (k0 < k1) {
        int sg = 0;
        while (sg + 1 < nseg && __builtin_amdgcn_readfirstlane(T.col0[sg + 1]) <= 4 * k0) ++sg;
        int nextcol = __builtin_amdgcn_readfirstlane(T.col0[sg + 1]);
        double acc[3] = {0.0, 0.0, 0.0}; // one 4 x 4 block element per component
        // the lane holds J_c[i = lane >> 4][u = lane & 3][w = (lane >> 2) & 3]: own-axis node i, A-index u, B-index w
        auto merge = [&](int cr) {
          const int di = lane >> 4;
          const int m0 = (base + cr) % kJX; // window column of node x is (x + 1) mod kJX: node cx - 1 + d sits at m0 + d
          int tX[3] = {m0 + di, m0 + qq, m0 + qb};
          const int tY[3] = {qq, di, qq}, tZ[3] = {qb, qb, di};
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            if (tX[c] >= kJX) tX[c] -= kJX;
            if (acc[c] != 0.0) unsafeAtomicAdd(&jtile[c * kJN + (tZ[c] * kD + tY[c]) * kJX + tX[c]], acc[c]);
          }
        };
        if constexpr ((ESK_PIPE >> MODE) & 1) {
          // the same K steps with the operands of step k + 1 requested before the products of step k (two operand sets,
          // steps taken in pairs inside a cell so that no set is ever copied)
          struct Ops { double a[2][2], b[2][2], pr[3]; }; // [x | y at row qq][old | new], [z | x at row qb][old | new], P_c at qq
          auto load = [&](int k, Ops& o) {
            const double* cp = stage + 4 * k + kk;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              o.a[0][h] = cp[(12 * h + 0 * kD + qq) * kPitch];
              o.a[1][h] = cp[(12 * h + 1 * kD + qq) * kPitch];
              o.b[0][h] = cp[(12 * h + 2 * kD + qb) * kPitch];
              o.b[1][h] = cp[(12 * h + 0 * kD + qb) * kPitch];
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) o.pr[c] = cp[(24 + c * kD + qq) * kPitch];
          };
          auto compute = [&](const Ops& o) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const double (&A)[2] = c == 1 ? o.a[0] : o.a[1]; // X: A = y, B = z;  Y: A = x, B = z;  Z: A = y, B = x
              const double (&Bv)[2] = c == 2 ? o.b[1] : o.b[0];
              const double Tc = -qdc[c] * (A[1] * (2.0 * Bv[1] + Bv[0]) + A[0] * (2.0 * Bv[0] + Bv[1]));
              acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(o.pr[c], Tc, acc[c], 0, 0, 0);
            }
          };
          Ops oa, ob;
          int k = k0;
          load(k, oa);
          while (true) {
            const int kend = min(k1, nextcol >> 2); // the steps of this cell that are this wave's
            for (; k + 1 < kend; k += 2) {
              load(k + 1, ob);
              compute(oa);
              if (k + 2 < k1) load(k + 2, oa);
              compute(ob);
            }
            if (k < kend) {
              if (k + 1 < k1) load(k + 1, ob);
              compute(oa);
              oa = ob;
              ++k;
            }
            if (k >= k1) break;
            merge(__builtin_amdgcn_readfirstlane(T.cell[sg]));
            acc[0] = acc[1] = acc[2] = 0.0;
            ++sg;
            nextcol = __builtin_amdgcn_readfirstlane(T.col0[sg + 1]);
          }
        }
        else {
        for (int k = k0; k < k1; ++k) {
            if (4 * k >= nextcol) {
              merge(__builtin_amdgcn_readfirstlane(T.cell[sg]));
              acc[0] = acc[1] = acc[2] = 0.0;
              ++sg;
              nextcol = __builtin_amdgcn_readfirstlane(T.col0[sg + 1]);
            }
            const double* cp = stage + 4 * k + kk;
  #pragma unroll
            for (int c = 0; c < 3; ++c) {
              const int aA = c == 1 ? 0 : 1, aB = c == 2 ? 0 : 2; // X: A = y, B = z;  Y: A = x, B = z;  Z: A = y, B = x
              const double sA_o = cp[(aA * kD + qq) * kPitch], sA_n = cp[(12 + aA * kD + qq) * kPitch];
              const double sB_o = cp[(aB * kD + qb) * kPitch], sB_n = cp[(12 + aB * kD + qb) * kPitch];
              const double Tc = -qdc[c] * (sA_n * (2.0 * sB_n + sB_o) + sA_o * (2.0 * sB_o + sB_n));
              acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(cp[(24 + c * kD + qq) * kPitch], Tc, acc[c], 0, 0, 0);
            }
          }
        }
        merge(__builtin_amdgcn_readfirstlane(T.cell[sg]));
      }
    }
    STAMP(8);
    lds_barrier_b();
#if ESK_DRAIN0
    // the next round's particles (requested before phase 1) and this round's stores are waited for HERE, ahead of the
    // flush's atomics: in order behind those, the wait would be for the atomics' acknowledgements (the compiler puts it at
    // the loop's end, where it copies the requested values into this round's registers)
    __builtin_amdgcn_s_waitcnt(0x0F70);
#endif
    // (every wave is past its last read of this round's table: the barrier above)
    if constexpr (PRE) tab_put(tab[rd & 1], tw);
    // ---- the next round starts at cell base + adv: the window's nodes below base + adv - 1 are final.  They leave in
    // ALIGNED GROUPS of kJG = 8 nodes (4 in MODE 1 and 2), one fp64 atomic per node (other pencils add to the same nodes): float atomics execute
    // at the memory side as 64-byte requests (MI355X_MICROARCH.md, "Global float atomics"), so a row's 8 aligned nodes are
    // ONE request where the `adv` (about 7) nodes a round completes, flushed as they came, straddled two 64-byte segments most of
    // the time -- the J atomics wrote 65 times the field per launch (profiles/r04_pmc_traffic_ecsimcorr_128.txt).  A
    // group's entry is (row = e >> 3, node = e & 7): shifts, where the walk over the whole window (every round, flushed or
    // not) cost three integer divisions per entry and three iterations per thread.  Finished nodes wait in the window for
    // their group (kJX has the room); the flushed columns, zeroed, become the nodes kJX further on (the window is circular,
    // nothing moves).  The last round flushes everything (its tail wraps periodically onto nodes 0, 1).  The next merges
    // are two barriers away.
    {
      const bool last = base + adv >= g.nx;
      const int fin = last ? g.nx + kD - 2 : base + adv - 1; // nodes below `fin` are final (the last round: all of them)
#if !ESK_FLUSH
      (void)fin;
      constexpr int kJPer = (3 * kJN + kThreadsB - 1) / kThreadsB;
      const int b11 = base % kJX;
#pragma unroll
      for (int k = 0; k < kJPer; ++k) {
        const int t = threadIdx.x + k * kThreadsB;
        if (t < 3 * kJN) {
          const int c = t / kJN, n = t % kJN;
          const int tx = n % kJX, ty = (n / kJX) % kD, tz = n / (kJX * kD);
          int j = tx - b11;
          if (j < 0) j += kJX;
          if (j < adv || last) {
            const double val = jtile[t];
            if (val != 0.0) {
              unsafeAtomicAdd(&J[c * g.cstride + g.nodew(base - 1 + j, cy - 1 + ty, cz - 1 + tz)], val);
              jtile[t] = 0.0;
            }
          }
        }
      }
#else
      while (jflushed + (last ? 1 : kJG) <= fin) {
#pragma unroll
        for (int k = 0; k < (kLinesB * kJG + kThreadsB - 1) / kThreadsB; ++k) {
          const int e = threadIdx.x + k * kThreadsB;
          const int row = e >> kJGs, x = jflushed + (e & (kJG - 1)); // row = (c, tz, ty): the window's own row order
          if (row < kLinesB && x >= -1 && x < fin) {
            const int c = row >> 4, tz = (row >> 2) & 3, ty = row & 3;
            int wc = (x + 1) % kJX; // (x + 1 >= 0)
            double* const w = &jtile[c * kJN + (tz * kD + ty) * kJX + wc];
            const double val = *w;
            if (val != 0.0) {
#if ESK_EXP != 1 // (experiment builds: 1 drops the J atomics, 2 the particle stores -- what each costs, results garbage)
              unsafeAtomicAdd(&J[c * g.cstride + g.nodew(x, cy - 1 + ty, cz - 1 + tz)], val);
#endif
              *w = 0.0;
            }
          }
        }
        jflushed += kJG;
      }
#endif
    }
  }

#ifdef ESK_STAMPS
  if (threadIdx.x == 0)
    for (int k = 0; k < 10; ++k) atomicAdd(&g_esk_stamps[k], stamp_acc_[k]);
#endif
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

#ifdef ESK_STAMPS
extern "C" int xpic_debug_esk_stamps(double* out, int reset)
{
  unsigned long long h[16];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_esk_stamps), sizeof(h)) != hipSuccess) return 1;
  for (int k = 0; k < 16; ++k) out[k] = (double)h[k];
  if (reset) {
    unsigned long long z[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_esk_stamps), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
#endif

// rounds a pencil's table holds: one per cell + the records the push reads ahead + a few (cells of several rounds each)
static int esk_rcap(const GridDev& g) { return g.nx + kSentinels + 8; }

bool esk_table_alloc(xpic_ctx* c)
{
  const GridDev& g = c->g;
  const size_t need = (size_t)g.ny * g.nzl * esk_rcap(g) * kPackDwords * sizeof(unsigned);
  if (c->esk_tab_bytes >= need) return true;
  (void)hipFree(c->esk_tab);
  c->esk_tab = nullptr; c->esk_tab_bytes = 0;
  if (hipMalloc(&c->esk_tab, need) != hipSuccess) { (void)hipGetLastError(); return false; }
  c->esk_tab_bytes = need;
  return true;
}

int esirkepov_push(xpic_ctx* c, Sort& s, int mode, const double* E, const double* B, double* J, double* pred_w_host)
{
  static const bool pre_env = !(getenv("XPIC_ESK_PRE") && atoi(getenv("XPIC_ESK_PRE")) == 0);
  bool pre = c->esk_pre != 0 && pre_env;
  // a re-binning that left its scatter to this push (sort_rebin(.., 2) in the basic step): the gathering form, if the round
  // table can be used; everything else resolves the deferral by the plain scatter first
  bool ga = pre && mode == 0 && s.deferred && !s.def_bucket && s.def_wrap && s.def_step == 0.0 && c->comm.kind == 0 && s.n > 0;
  if (!ga) XPIC_CALL(sort_materialize(c, s));
  s.prebinned = false;
  if (pred_w_host) *pred_w_host = 0.0;
  // rank-uniform checks first: every slab fails them alike, BEFORE anyone enters the collective below (a slab that
  // returned from here alone would leave an empty neighbour waiting in its all-reduce)
  const GridDev& g = c->g;
  XPIC_CHECK(g.nx >= 6 && g.ny >= 6 && g.nzl >= 6, "the Esirkepov tile needs every grid extent >= 6");
  const long nblocks = (long)g.ny * g.nzl; // one workgroup per x-pencil
  XPIC_CHECK(nblocks < 2147483647L, "too many pencils for one launch");
  XPIC_CHECK(mode != 2 || c->kry_w, "second_push needs the ecsimcorr scheme's work vectors");
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
  const double qm = s.par.q / s.par.m;
  const double qn_Np = s.par.q * s.par.n / s.par.Np;
  const double alpha = qn_Np / (6.0 * g.dt); // basic/particles.cpp:44, ecsimcorr/particles.cpp:130
  double* scal = c->red_out;                  // [0] pred_w, [1] bad count (as int)
  XPIC_HIP(hipMemsetAsync(scal, 0, sizeof(double) * 2, c->stream));
  const char* name = mode == 0 ? "basic_push" : (mode == 1 ? "corr_first_push" : "corr_second_push");
  // the rounds of every pencil, composed beforehand (PRE); the table is sized for one round per cell + a few
  const int rcap = esk_rcap(g);
  if (pre && !esk_table_alloc(c)) pre = false; // (no room for the table: the pushes compose for themselves)
  if (ga && !pre) { XPIC_CALL(sort_materialize(c, s)); ga = false; }
  int* const flags = (int*)(scal + 1); // [0] particles that moved too far, [1] a pencil with more rounds than the table holds
  auto launch = [&](bool with_table) {
    Timed t(c, name);
    dim3 grid((unsigned)nblocks), block(kThreadsB);
    if (with_table) {
      const unsigned nb = (unsigned)((nblocks + 255) / 256);
      if (mode == 0) hipLaunchKernelGGL((k_esk_rounds<StageDim<0>::kSegM, StageDim<0>::kCols>), dim3(nb), dim3(256), 0, c->stream, g, s.d.cell_start, c->esk_tab, rcap, nblocks, flags + 1);
      else if (mode == 1) hipLaunchKernelGGL((k_esk_rounds<StageDim<1>::kSegM, StageDim<1>::kCols>), dim3(nb), dim3(256), 0, c->stream, g, s.d.cell_start, c->esk_tab, rcap, nblocks, flags + 1);
      else hipLaunchKernelGGL((k_esk_rounds<StageDim<2>::kSegM, StageDim<2>::kCols>), dim3(nb), dim3(256), 0, c->stream, g, s.d.cell_start, c->esk_tab, rcap, nblocks, flags + 1);
    }
#define ESK(M) (with_table ? (g.pow2 ? k_esirkepov_push<M, true, true> : k_esirkepov_push<M, false, true>) \
                           : (g.pow2 ? k_esirkepov_push<M, true, false> : k_esirkepov_push<M, false, false>))
    if (mode == 0 && ga && with_table)
      hipLaunchKernelGGL((g.pow2 ? k_esirkepov_push<0, true, true, true> : k_esirkepov_push<0, false, true, true>), grid, block, 0, c->stream, g,
        s.d, E, B, J, qm, alpha, qn_Np, scal, flags, c->esk_tab, rcap);
    else if (mode == 0) hipLaunchKernelGGL(ESK(0), grid, block, 0, c->stream, g, s.d, E, B, J, qm, alpha, qn_Np, scal, flags, c->esk_tab, rcap);
    else if (mode == 1) hipLaunchKernelGGL(ESK(1), grid, block, 0, c->stream, g, s.d, E, B, J, qm, alpha, qn_Np, scal, flags, c->esk_tab, rcap);
    else {
      // per-workgroup pred_w partials go to the (idle) Krylov work vector: one double per pencil
      hipLaunchKernelGGL(ESK(2), grid, block, 0, c->stream, g, s.d, E, B, J, qm, alpha, qn_Np, c->kry_w, flags, c->esk_tab, rcap);
#undef ESK
      hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, c->stream, c->kry_w, nblocks, scal);
    }
  };
  // (a push that composes its rounds itself -- no room for the table, or a pencil with more rounds than it holds -- is slower
  // by 6 %: counted, so that the caller can see it: xpic_profile_get("esk_self_composed"))
  if (!pre && c->profiling) c->prof["esk_self_composed"].launches += 1;
  launch(pre);
  XPIC_HIP(hipGetLastError());
  XPIC_HIP(hipMemcpyAsync(c->red_host, scal, sizeof(double) * 2, hipMemcpyDeviceToHost, c->stream));
  XPIC_HIP(hipStreamSynchronize(c->stream));
  if (pre) {
    // (the flag travels with the scalars; only a raised flag -- the push then returned at once -- costs a second launch)
    int fl[2];
    memcpy(fl, &c->red_host[1], sizeof(fl));
    if (fl[1] != 0) {
      if (ga) { XPIC_CALL(sort_materialize(c, s)); ga = false; } // (the gathering push returned before touching anything)
      XPIC_HIP(hipMemsetAsync(scal, 0, sizeof(double) * 2, c->stream));
      if (c->profiling) c->prof["esk_self_composed"].launches += 1;
      launch(false);
      XPIC_HIP(hipGetLastError());
      XPIC_HIP(hipMemcpyAsync(c->red_host, scal, sizeof(double) * 2, hipMemcpyDeviceToHost, c->stream));
      XPIC_HIP(hipStreamSynchronize(c->stream));
    }
  }
  if (ga) sort_deferred_done(s); // r2 / v2 hold the sorted, wrapped, pushed records: they become the sort
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
