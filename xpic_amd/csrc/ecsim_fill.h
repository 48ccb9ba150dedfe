// ecsim_fill.h -- what the two bodies of the mass-matrix assembly share (ecsim.hip: k_ecsim_fill, the default;
// ecsim_ws.hip: k_ecsim_fill_ws, the warp-specialised alternative): stage / window geometry, the octant accumulators'
// numbering, LDS helpers, the B neighbourhood table.  See ecsim.hip for the algorithm.
#pragma once

#include <cstdint>

#include "common.h"
#include "device_common.h"
#include "lstencil.h"

// Tunables of the assembly.  Every one of them changes generated code, so a value from the command line is an experiment
// build (common.h refuses it without -DXPIC_EXPERIMENT); the defaults below are what the tests cover.
#ifndef FILL_EXP
#define FILL_EXP 0 // ablations (tools/fill_exp.sh; results garbage): 11 the flush without its read-modify-write loads, 12 without loads
                   // and stores; 1 no phase 2, 6 neither phase 1 nor phase 2, 7 as 6 and no merge, 8 as 7 and no flush
#endif
#ifndef FILL_GA_EXP
#define FILL_GA_EXP 0 // ablations (results garbage): 1 the gathering assembly without its stores, 2 with the position's three only
#endif
#ifndef FILL_GA_NOCHAIN
#define FILL_GA_NOCHAIN 0 // ablation (results garbage): the gathering assembly without its index indirection
#endif
#ifndef FILL_KCP
#define FILL_KCP 64
#endif
#ifndef FILL_OCC
#define FILL_OCC 2
#endif
#ifndef FILL_PITCH
#define FILL_PITCH 38
#endif
#ifndef FILL_WPITCH
#define FILL_WPITCH 7
#endif

namespace xpic {
namespace fill {

constexpr int kW = 4;             // waves per workgroup = cells per chunk
constexpr int kCP = FILL_KCP;     // particles staged per pass and wave: one pass for a cell of up to 64 (53 % of Poisson(64) cells)
// One stage slot = one particle: 24 weights [c][i][h] (i: the 2 x 2 nodes transverse to the component's staggered axis,
// h: lower / upper node along it), 9 A_p*matB, 3 I_p.  Pitch 38 doubles = 76 dwords: the 16-byte stores of 8
// consecutive slots fall in 8 distinct bank quads (76 l mod 32 = 0,12,24,4,16,28,8,20) and the operand reads of the
// two particles that share an LDS cycle are 12 banks apart (pitch 42 measured the same).  Four waves x 64 slots + the rest
// are 79 856 bytes: two workgroups per CU (the per-lane window offsets and the cell_start row are read from global
// memory -- five 16-byte loads per chunk, scalar loads -- instead of LDS copies: those 6.6 KB are what lets a wave stage 64
// particles).  Per assembly at 256^3 x 64: 48 slots 125.4 ms, 56 slots 123.9, 64 slots 121.7.
constexpr int kPitch = FILL_PITCH;
constexpr int kOffAB = 24; // [24, 33): A_p*matB row-major, [33, 36): I_p
constexpr int kStage = kCP * kPitch;
constexpr int kAcc = 36;          // accumulators per lane: 30 matL (component pair x octant bits of the pair) + 6 currI
typedef double mfma_acc __attribute__((ext_vector_type(4)));
typedef double dpair __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) double GlobalDouble; // matL / currI are global memory: a pointer rebuilt from an
// integer is 'flat' to the compiler, and flat loads also count in lgkmcnt -- the LDS-only barriers then wait for HBM
typedef __attribute__((address_space(1))) dpair GlobalPair;
typedef __attribute__((address_space(3))) const char* LdsBytes; // LDS addresses are 32 bits: say so where address arithmetic is hot
typedef __attribute__((address_space(3))) double LdsDouble;
typedef __attribute__((address_space(3))) dpair LdsDouble2;
constexpr int kMatLines = 816;    // distinct (c1, row dy, row dz, k) streams one pencil can touch
constexpr int kCurLines = 16;     // (c, dy, dz) streams of currI
constexpr int kLines = kMatLines + kCurLines;
constexpr int kThreads = kW * 64;
constexpr int kSlots = kW + 2;    // window columns: kW finished + 2 carried
// doubles between two window lines (>= kSlots).  The merge's 36 ds_add_f64 per lane hit lines that lie multiples of 8 apart:
// with the natural pitch of 6 doubles those share their banks (line * 12 dwords mod 32 has period 8), an odd pitch cycles
// through all 16 bank pairs.  Per assembly: pitch 6 106.6 ms, 7 101.5, 9 102.8, 8 138.3 (the seed and the flush give up
// their 16-byte window accesses for it).
constexpr int kWP = FILL_WPITCH;
static_assert(kWP >= kSlots, "window pitch");
constexpr int kOwn = (kLines + kThreads - 1) / kThreads; // window lines owned by a thread (init, flush, carry)
constexpr int kDtabPitch = 40;     // ushorts per lane of the transposed offset table

// accumulator of the block (c1, c2) for a particle of octant o = ox | oy << 1 | oz << 2: the rows depend on the
// octant bit of axis c1 only, the columns on that of axis c2
__host__ __device__ constexpr int acc_main(int c1, int c2, int o)
{
  const int o1 = (o >> c1) & 1, o2 = (o >> c2) & 1;
  if (c1 == c2) return c1 * 2 + o1;
  const int pair = c1 * 2 + (c2 > c1 ? c2 - 1 : c2);
  return 6 + pair * 4 + o1 * 2 + o2;
}
// currI: instruction 1 holds the X and Y rows (bits ox, oy), instruction 2 the Z rows (bit oz)
__host__ __device__ constexpr int acc_cur1(int o) { return 30 + (o & 1) * 2 + ((o >> 1) & 1); }
__host__ __device__ constexpr int acc_cur2(int o) { return 34 + ((o >> 2) & 1); }
// row of the cell block (numbering of lstencil.h: block_node_offset) of node i of group g of component c; g = octant
// bit + h is the node slot along the component's own axis
__host__ __device__ constexpr int row_of(int c, int g, int i)
{
  if (c == 0) return i * 3 + g;                                  // i = k * 2 + j
  if (c == 1) return 12 + ((i >> 1) * 3 + g) * 2 + (i & 1);      // i = k * 2 + ix
  return 24 + (g * 2 + (i >> 1)) * 2 + (i & 1);                  // i = j * 2 + ix
}

static_assert(kLines * kWP <= kW * kStage, "the merge window must fit in the (dead) staging area");

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

// B neighbourhood of cell (cx,cy,cz) = every B value a CIC gather from inside that cell can touch:
//   Bx at (xn in cx..cx+1, ys in cy-1..cy+1, zs in cz-1..cz+1)   -> [ 0,18): (kl*3 + jl)*2 + i
//   By at (xs in cx-1..cx+1, yn in cy..cy+1, zs in cz-1..cz+1)   -> [18,36): (kl*2 + j)*3 + il
//   Bz at (xs in cx-1..cx+1, ys in cy-1..cy+1, zn in cz..cz+1)   -> [36,54): (k*3 + jl)*3 + il
// lane's entry of the table: the row of B it lies in (fixed for a pencil) and its x offset from the cell
__device__ inline const double* bnb_row(const GridDev& g, const double* __restrict__ B, int lane, int cy, int cz, int* ox_out)
{
  *ox_out = 0;
  if (lane >= 54) return nullptr;
  int c, ox, oy, oz;
  if (lane < 18) { c = 0; ox = lane % 2; oy = (lane / 2) % 3 - 1; oz = lane / 6 - 1; }
  else if (lane < 36) { const int l = lane - 18; c = 1; ox = l % 3 - 1; oy = (l / 3) % 2; oz = l / 6 - 1; }
  else { const int l = lane - 36; c = 2; ox = l % 3 - 1; oy = (l / 3) % 3 - 1; oz = l / 9; }
  *ox_out = ox;
  return B + c * g.cstride + g.node(0, g.wy(cy + oy), g.wz(cz + oz));
}

struct Prefetch {
  int start, cnt;  // cell_start of the cell this wave handles next
  double p[6];     // x, y, z, vx, vy, vz of lane's particle of the next pass
  double b;        // lane's value of the next cell's 54-value B neighbourhood
  int srcx;        // gathering assembly: source index of slot start + kCP + lane (the second pass's records)
  unsigned long long incm; // ... and the lanes whose record came from a neighbouring slab (already moved: no first_push for it)
};

// bits of the assembly's error word (xpic_ctx::fill_err): raised on the device, read once per assembly by
// ecsim_fill_current and agreed on by all slabs before anybody returns
constexpr int kFillErrPencil = 1;  // the gathering form met an x-pencil of 2^29 particles or more (sort_rebin rules that out)
constexpr int kFillErrTimeout = 2; // k_ecsim_fill_ws: one of the pipeline's bounded waits gave up (matrix incomplete)

}  // namespace fill

// ecsim_ws.hip: one colour launch of the warp-specialised body (same arguments as the classic launch in ecsim.hip)
void launch_fill_ws(xpic_ctx* c, const Sort& s, unsigned nblocks, const double* B, double* currI_sort, double* matL,
  const unsigned short* dtab, const int* linetab, const int* cowr, int cy0, int cys, int ncy, int cz0, int czs, int my_order,
  int ncol_y, int per_y, int per_z, int first_sort, unsigned long long zord);

}  // namespace xpic
