// ecsim_ws.hip -- k_ecsim_fill_ws, the warp-specialised body of the mass-matrix assembly (round 4; selected by
// xpic_set_fill_kernel(ctx, 1), not the default: DESIGN.md section 5d).  Tables, colours and the launch sequence are those
// of ecsim.hip (ecsim_fill_sort calls launch_fill_ws for each colour launch).
#include <type_traits>
#include <utility>

#include "ecsim_fill.h"

namespace xpic {

int experiment_ecsim_ws() { return XPIC_TU_EXPERIMENT; }

using namespace fill;

namespace {

// =====================================================================================================================
// k_ecsim_fill_ws -- the same assembly with its parts on different waves (round 4).
//
// The kernel above runs every wave through phase 1 (VALU, lane = particle), phase 2 (matrix cores), merge and flush in
// turn, two waves per SIMD: whether a SIMD's two waves complement each other (one feeding the matrix pipe while the other
// issues vector / memory instructions) is left to chance, each wave issues a third of the time, and the merge window
// aliases the stage, so a chunk's merge + flush cannot overlap the next chunk's phase 1.  Here ONE workgroup of 16 waves
// owns the CU (all 160 KB of LDS, 128 registers per lane) and every SIMD holds one wave of each of four fixed roles, which
// work on the same cell of a chunk of 4:
//   * PRODUCER (waves 8..11): cell by cell, pass by pass it loads the particles, runs phase 1 and leaves the
//     octant-compacted operands in one of the TWO stage buffers of its SIMD;
//   * two CONSUMERS (waves 0..3: the octants with oz = 0, waves 4..7: oz = 1) run phase 2 out of the stage buffers -- two
//     matrix-instruction streams per SIMD that cover each other's operand-read latency -- keep their part of the cell block
//     in registers (26 of the 36 accumulators each) and merge it into the window when the cell is complete;
//   * FLUSHER (waves 12..15): owns the window's lines; when all consumers have merged a chunk it adds the four finished
//     columns to matL / currI (read-modify-write, the old values requested a chunk ahead: its memory counter holds nothing
//     else), moves the two unfinished columns to the front and clears the rest.
// The window (46.6 KB) has LDS of its own: a chunk's flush runs beside the next chunk's phases 1 and 2.  Nothing in the
// main loop is a workgroup barrier: stage buffers change hands through sequence numbers in LDS (FULL / FREE), the window
// through two counters (MERGED: consumer waves that merged a chunk, SEEDED: flusher waves that flushed and re-seeded it).
// A consumer never waits for another consumer's cell, only -- a chunk later -- for the flush.  Every wait is a bounded
// spin: a wave that waited longer than any schedule can need raises the ABORT word and leaves, the others follow, and the
// host reports the assembly as failed instead of hanging the GPU.
// Work per cell, colours, first touch, the window's line table and the octant accumulators are those of k_ecsim_fill:
// the matrix it assembles is the same up to the summation order of the window's atomics.
// =====================================================================================================================
#ifndef FILL_WS_KCP
#define FILL_WS_KCP 44
#endif
#ifndef FILL_WS_PRIO_C
#define FILL_WS_PRIO_C 2 // s_setprio of the consumers / the producer / the flusher
#endif
#ifndef FILL_WS_PRIO_P
#define FILL_WS_PRIO_P 1
#endif
#ifndef FILL_WS_PRIO_F
#define FILL_WS_PRIO_F 0
#endif
constexpr int kWsCP = FILL_WS_KCP;            // slots of one stage buffer (two per SIMD: 4 x 2 x 44 x 304 B = 107 KB)
constexpr int kWsStage = kWsCP * kPitch;      // doubles of one stage buffer
constexpr int kWsThreads = 1024;
constexpr int kWsFlush = 256;                 // flusher threads (own the window's lines)
constexpr int kWsOwn = (kLines + kWsFlush - 1) / kWsFlush;
constexpr unsigned kWsSpinLimit = 1u << 22;   // polls of ~100 cycles: three orders of magnitude beyond any legitimate wait
enum { kFlFull = 0, kFlFree = 8 /* [half][w][b] */, kFlMerged = 24, kFlSeeded = 25, kFlAbort = 26, kFlCount = 28 };
static_assert(kWsCP % 4 == 0 && kWsCP <= 64, "a stage buffer holds whole K = 4 steps of at most one wave of particles");
static_assert((kLines * kWP + 64 + kW) * 8 + 8 * kWsStage * 8 + kPitch * 8 + kW * 54 * 8 + kW * 2 * 12 * 4 + kFlCount * 4 + 64 * kDtabPitch * 2 <= 160 * 1024,
  "window + stage buffers + offset table exceed the LDS of a CU");

// is accumulator e touched by the octants of half h (oz = h)?
__host__ __device__ constexpr bool acc_in_half(int e, int h)
{
  for (int o = h * 4; o < h * 4 + 4; ++o) {
    for (int c1 = 0; c1 < 3; ++c1)
      for (int c2 = 0; c2 < 3; ++c2)
        if (acc_main(c1, c2, o) == e) return true;
    if (acc_cur1(o) == e || acc_cur2(o) == e) return true;
  }
  return false;
}

typedef __attribute__((address_space(3))) unsigned LdsWord;

#ifdef FILL_STAMPS
// section timers of the warp-specialised kernel (experiment build): 8 per role, of the role's first wave in every workgroup
__device__ unsigned long long g_fill_ws_stamps[32];
#define WSTAMP(k)                                                 \
  do {                                                            \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    ws_acc_[k] += now_ - ws_t_;                                   \
    ws_t_ = now_;                                                 \
  } while (0)
#define WSTAMP_INIT unsigned long long ws_t_ = __builtin_readcyclecounter(); unsigned long long ws_acc_[8] = {}
#define WSTAMP_DUMP(base, cond)                                                                     \
  do {                                                                                              \
    if ((cond) && lane == 0)                                                                        \
      for (int k_ = 0; k_ < 8; ++k_) atomicAdd(&g_fill_ws_stamps[(base) + k_], ws_acc_[k_]);        \
  } while (0)
}  // namespace
}  // namespace xpic
extern "C" int xpic_debug_fill_ws_stamps(double* out, int reset)
{
  unsigned long long h[32];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(xpic::g_fill_ws_stamps), sizeof(h)) != hipSuccess) return 1;
  for (int i = 0; i < 32; ++i) out[i] = (double)h[i];
  if (reset) {
    unsigned long long z[32] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(xpic::g_fill_ws_stamps), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
namespace xpic {
namespace {
#else
#define WSTAMP(k)
#define WSTAMP_INIT
#define WSTAMP_DUMP(base, cond)
#endif

// LDS words that other waves of the workgroup write: always read / written by explicit DS instructions
__device__ inline unsigned lds_peek(const unsigned* p)
{
  unsigned v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((const LdsWord*)p) : "memory");
  return __builtin_amdgcn_readfirstlane(v);
}
// publish: everything this wave sent to the LDS before is complete (a wave's DS operations finish in order), then lane 0 writes
__device__ inline void lds_post(unsigned* p, unsigned v, int lane)
{
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"((LdsWord*)p), "v"(v) : "memory");
}
__device__ inline void lds_bump(unsigned* p, int lane)
{
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"((LdsWord*)p), "v"(1u) : "memory");
}
// wait until *p has reached `target` (sequence numbers only grow); false: aborted (this wave or another one gave up)
__device__ inline bool lds_wait(const unsigned* p, unsigned target, unsigned* flags)
{
  for (unsigned it = 0;; ++it) {
    if ((int)(lds_peek(p) - target) >= 0) return true;
    if ((it & 63u) == 63u) {
      if (lds_peek(flags + kFlAbort)) return false;
      if (it > kWsSpinLimit) {
        asm volatile("ds_write_b32 %0, %1" ::"v"((LdsWord*)(flags + kFlAbort)), "v"(1u) : "memory");
        return false;
      }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

template <int HALF, int... Es>
__device__ __forceinline__ void ws_merge(std::integer_sequence<int, Es...>, char* wv, const uint4 (&dq)[kDtabPitch / 8], double (&acc)[kAcc])
{
  // (the offsets are read from their LDS table at every merge: kept in registers across the pencil they were spilled, and
  // the merge reloaded them from scratch one by one -- 14 serialized round trips, 45 % of a consumer's time)
  // lane's element of accumulator e goes to window byte offset(e) (16-bit entries of the transposed table) + 8 * w
  auto one = [&](auto tag) {
    constexpr int e = decltype(tag)::value;
    if constexpr (acc_in_half(e, HALF)) {
      const uint4 q4 = dq[e / 8];
      const unsigned word = ((e % 8) >> 1) == 0 ? q4.x : ((e % 8) >> 1) == 1 ? q4.y : ((e % 8) >> 1) == 2 ? q4.z : q4.w;
      const unsigned off = (word >> (16 * (e & 1))) & 0xffffu;
      unsafeAtomicAdd((double*)(wv + off), acc[e]);
      acc[e] = 0.0;
    }
  };
  (one(std::integral_constant<int, Es>{}), ...);
}

template <bool P2>
__global__ void __launch_bounds__(kWsThreads, 1) k_ecsim_fill_ws(GridDev g, SortDev s, const double* __restrict__ B,
  double* currI, double* matL, const unsigned short* __restrict__ dtab, const int* __restrict__ linetab, const int* __restrict__ cowr, double q, double m,
  double mpw, int cy0, int cystep, int ncy, int cz0, int czstep, int my_order, int ncol_y, int per_y, int per_z, int first_sort,
  int* __restrict__ err, unsigned long long zord)
{
  const int cy = cy0 + (int)(blockIdx.x % ncy) * cystep;
  const int cz = cz0 + (int)(blockIdx.x / ncy) * czstep;

  __shared__ __attribute__((aligned(16))) double win[kLines * kWP + 64 + kW]; // + the per-lane dummy targets of the merge
  __shared__ __attribute__((aligned(16))) double stage[kW * 2 * kWsStage];
  __shared__ __attribute__((aligned(16))) double zslot[kPitch];
  __shared__ double bnb[kW][54];
  __shared__ __attribute__((aligned(16))) int hdr[kW][2][12]; // per stage buffer: the 8 octant counts, last-pass flag
  __shared__ unsigned flags[kFlCount];
  __shared__ __attribute__((aligned(16))) unsigned short dtl[64 * kDtabPitch]; // per-lane window offsets of the merge

  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const double dt = g.dt;
  for (int i = threadIdx.x; i < kLines * kWP + 64 + kW; i += kWsThreads) win[i] = 0.0;
  for (int i = threadIdx.x; i < 64 * kDtabPitch; i += kWsThreads) dtl[i] = dtab[i];
  if (threadIdx.x < kPitch) zslot[threadIdx.x] = 0.0;
  if (threadIdx.x < kFlCount) flags[threadIdx.x] = 0u;
  __syncthreads(); // the only workgroup barrier of the kernel
  const int nch = g.nx / kW;
  const long pencil0 = ((long)cz * g.ny + cy) * g.nx;
  const int w = wave & 3; // the SIMD's cell of a chunk

  // =================================================== consumers ====================================================
  auto consumer = [&](auto half_tag) {
    constexpr int HALF = decltype(half_tag)::value;
    const int kk = lane >> 4, qb = (lane >> 2) & 3, qj = lane & 3;
    const int offA8 = 8 * (qj * 2 + (qb >> 1)), offB8 = 8 * (qj * 2 + (qb & 1));
    const int offI18 = 8 * (kOffAB + 9 + (qb >> 1));
    double acc[kAcc];
#pragma unroll
    for (int e = 0; e < kAcc; ++e) acc[e] = 0.0;
    __builtin_amdgcn_s_setprio(FILL_WS_PRIO_C);
    unsigned pass = 0;
    WSTAMP_INIT;
    for (int j = 0; j < nch; ++j) {
      for (;;) {
        const int b = pass & 1;
        if (!lds_wait(flags + kFlFull + w * 2 + b, pass + 1, flags)) { if (lane == 0) atomicOr(err, kFillErrTimeout); return; }
        WSTAMP(0);
        const int4 h0 = *(const int4*)&hdr[w][b][0], h1 = *(const int4*)&hdr[w][b][4];
        const int ocnt[8] = {__builtin_amdgcn_readfirstlane(h0.x), __builtin_amdgcn_readfirstlane(h0.y),
          __builtin_amdgcn_readfirstlane(h0.z), __builtin_amdgcn_readfirstlane(h0.w), __builtin_amdgcn_readfirstlane(h1.x),
          __builtin_amdgcn_readfirstlane(h1.y), __builtin_amdgcn_readfirstlane(h1.z), __builtin_amdgcn_readfirstlane(h1.w)};
        const bool lastpass = __builtin_amdgcn_readfirstlane(hdr[w][b][8]) != 0;
        const double* st = stage + (w * 2 + b) * kWsStage;
        int run = HALF ? ocnt[0] + ocnt[1] + ocnt[2] + ocnt[3] : 0;
        WSTAMP(1);
#pragma unroll
        for (int o = HALF * 4; o < HALF * 4 + 4; ++o) {
          const int no = ocnt[o];
          const double* seg = st + run * kPitch;
          run += no;
          const int nst = lastpass ? (no + 3) >> 2 : no >> 2;
          LdsBytes spr = (LdsBytes)(const char*)(seg + kk * kPitch);
          const LdsBytes seg_end = (LdsBytes)(const char*)(seg + no * kPitch);
          for (int t = 0; t < nst; ++t, spr += 4 * kPitch * 8) {
            const LdsBytes sb = spr < seg_end ? spr : (LdsBytes)(const char*)zslot;
            const LdsDouble* sp = (const LdsDouble*)sb;
            const LdsDouble* spA = (const LdsDouble*)(sb + offA8);
            const LdsDouble* spB = (const LdsDouble*)(sb + offB8);
            double a[3], bb[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) { a[c] = spA[c * 8]; bb[c] = spB[c * 8]; }
            const LdsDouble2* u = (const LdsDouble2*)(sp + kOffAB);
            const dpair u0 = u[0], u1 = u[1], u2 = u[2], u3 = u[3];
            const double ab[9] = {u0.x, u0.y, u1.x, u1.y, u2.x, u2.y, u3.x, u3.y, sp[kOffAB + 8]};
            double bm[9];
#pragma unroll
            for (int e = 0; e < 9; ++e) bm[e] = bb[e % 3] * ab[e];
            const double ai1 = qb < 2 ? bb[0] : bb[1], ai2 = bb[2];
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
          }
        }
        WSTAMP(2);
        lds_post(flags + kFlFree + (HALF * kW + w) * 2 + b, pass + 1, lane); // every operand read of this buffer has returned
        WSTAMP(3);
        ++pass;
        if (lastpass) break;
      }
      // ---- merge this half of the cell block into the window
      if (!lds_wait(flags + kFlSeeded, (unsigned)(kW * j), flags)) { if (lane == 0) atomicOr(err, kFillErrTimeout); return; } // re-seeded after chunk j - 1
      WSTAMP(5);
      uint4 dq[kDtabPitch / 8]; // the lane's window offsets of its accumulator elements: 36 16-bit entries
      {
        const uint4* dp = reinterpret_cast<const uint4*>(dtl + lane * kDtabPitch);
#pragma unroll
        for (int k = 0; k < kDtabPitch / 8; ++k) dq[k] = dp[k];
      }
      ws_merge<HALF>(std::make_integer_sequence<int, kAcc>{}, (char*)(win + w), dq, acc);
      lds_bump(flags + kFlMerged, lane);
      WSTAMP(6);
    }
    WSTAMP_DUMP(HALF * 8, w == 0);
  };
  if (wave < kW) { consumer(std::integral_constant<int, 0>{}); return; }
  if (wave < 2 * kW) { consumer(std::integral_constant<int, 1>{}); return; }

  if (wave >= 3 * kW) {
    // ================================================= flusher ======================================================
    __builtin_amdgcn_s_setprio(FILL_WS_PRIO_F);
    const int t = threadIdx.x - 3 * kW * 64;
    // window lines this thread owns (line = t + mm * kWsFlush): address of their column 0, first-touch flag in bit 0
    uintptr_t lbase[kWsOwn];
#pragma unroll
    for (int mm = 0; mm < kWsOwn; ++mm) {
      const int line = t + mm * kWsFlush;
      lbase[mm] = 0;
      if (line >= kLines) continue;
      const int ld = linetab[line];
      const int ry = g.wy(cy + ((ld >> 2) & 3) - 1);
      const int rz = cz + ((ld >> 4) & 3) - 1;
      const int rzw = g.G == 0 ? (rz < 0 ? rz + g.nzl : (rz >= g.nzl ? rz - g.nzl : rz)) : rz + 1;
      double* base = line < kMatLines
        ? matL + g.lindex(ld & 3, rzw, ry, 0, ld >> 6)
        : currI + (ld & 3) * g.cstride + g.node(0, ry, g.wz(rz));
      bool first = first_sort && line < kMatLines; // no co-writer of this line runs in an earlier launch: store, do not add
      if (first) {
        const int bodyy = g.ny - g.ny % per_y, bodyz = g.nzl - g.nzl % per_z;
        for (int e = 0; e < 8 && first; ++e) {
          const int cw = cowr[line * 8 + e];
          if (cw == 0x7fffffff) break;
          const int oy = (cw & 0xff) - 8, oz = ((cw >> 8) & 0xff) - 8;
          int py = cy + oy, pz = cz + oz;
          py = py < 0 ? py + g.ny : (py >= g.ny ? py - g.ny : py);
          if (g.G == 0) pz = pz < 0 ? pz + g.nzl : (pz >= g.nzl ? pz - g.nzl : pz);
          else if (pz < 0 || pz >= g.nzl) continue;
          const int ca = py < bodyy ? py % per_y : per_y + (py - bodyy);
          const int cb = g.G == 0 ? (pz < bodyz ? pz % per_z : per_z + (pz - bodyz)) : pz % per_z;
          if ((int)((zord >> (4 * cb)) & 15u) * ncol_y + ca < my_order) first = false; // zord: launch position of z colour cb
        }
      }
      lbase[mm] = (uintptr_t)base | (first ? 1u : 0u);
    }
    // old[] = the current values of the next chunk's finished columns, requested a chunk ahead (the addresses depend on
    // the chunk alone; these loads and the flush's stores are all this wave has in its memory counter)
    double old[kWsOwn][kW];
    auto request_old = [&](int jc) {
#pragma unroll
      for (int mm = 0; mm < kWsOwn; ++mm) {
        const int line = t + mm * kWsFlush;
#pragma unroll
        for (int c = 0; c < kW; ++c) old[mm][c] = 0.0;
        const uintptr_t lb = lbase[mm];
        if (!lb || (lb & 1) || jc >= nch) continue;
        const GlobalDouble* ptr = (const GlobalDouble*)(lb & ~(uintptr_t)1) + (long)jc * (line < kMatLines ? kLBlock : kW);
#pragma unroll
        for (int c = 0; c < kW; c += 2) {
          const dpair v = *(const GlobalPair*)(ptr + c);
          old[mm][c] = v.x; old[mm][c + 1] = v.y;
        }
      }
    };
    request_old(0);
    WSTAMP_INIT;
    for (int jc = 0; jc < nch; ++jc) {
      if (!lds_wait(flags + kFlMerged, (unsigned)(2 * kW * (jc + 1)), flags)) { if (lane == 0) atomicOr(err, kFillErrTimeout); return; }
      WSTAMP(0);
      // add the window's four finished columns, store, move the two unfinished columns to the front, clear the rest
#pragma unroll
      for (int mm = 0; mm < kWsOwn; ++mm) {
        const int line = t + mm * kWsFlush;
        const uintptr_t lb = lbase[mm];
        if (!lb) continue;
        double* wl = win + line * kWP;
        double wv[kSlots];
#pragma unroll
        for (int c = 0; c < kSlots; ++c) wv[c] = wl[c];
        GlobalDouble* ptr = (GlobalDouble*)(lb & ~(uintptr_t)1) + (long)jc * (line < kMatLines ? kLBlock : kW);
#pragma unroll
        for (int c = 0; c < kW; c += 2)
          *(GlobalPair*)(ptr + c) = dpair{old[mm][c] + wv[c], old[mm][c + 1] + wv[c + 1]};
        wl[0] = wv[kW]; wl[1] = wv[kW + 1];
#pragma unroll
        for (int c = 2; c < kSlots; ++c) wl[c] = 0.0;
      }
      lds_bump(flags + kFlSeeded, lane);
      WSTAMP(1);
      request_old(jc + 1);
      WSTAMP(2);
    }
    // ---- the two columns left over are x = nx, nx + 1 = 0, 1 (periodic): columns this thread has already written,
    // added with atomics (2 of nx columns)
#pragma unroll
    for (int mm = 0; mm < kWsOwn; ++mm) {
      const int line = t + mm * kWsFlush;
      if (line < kLines && lbase[mm]) {
        double* base = (double*)(lbase[mm] & ~(uintptr_t)1);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const double v = win[line * kWP + c];
          if (v != 0.0) unsafeAtomicAdd(base + g.wx(c), v);
        }
      }
    }
    WSTAMP(3);
    WSTAMP_DUMP(24, w == 0);
    return;
  }

  // ================================================= producer =======================================================
  __builtin_amdgcn_s_setprio(FILL_WS_PRIO_P);
  auto cell_x = [&](int i) { return (i + 1 == g.nx) ? 0 : i + 1; };
  int box;
  const double* const brow = bnb_row(g, B, lane, cy, cz, &box);
  Prefetch pf;
  auto prefetch_cell = [&](int i) {
    pf.start = 0; pf.cnt = 0; pf.b = 0.0;
    if (i >= g.nx) return;
    const int cx = cell_x(i);
    using UniformInts = const __attribute__((address_space(4))) int*;
    UniformInts cs = (UniformInts)(s.cell_start + pencil0);
    const int cxu = __builtin_amdgcn_readfirstlane(cx);
    pf.start = cs[cxu];
    pf.cnt = cs[cxu + 1] - pf.start;
    pf.b = brow ? brow[g.wx(cx + box)] : 0.0;
    if (lane < min(kWsCP, pf.cnt)) {
      const long p = (long)pf.start + lane;
#pragma unroll
      for (int a = 0; a < 3; ++a) { pf.p[a] = s.r[a][p]; pf.p[3 + a] = s.v[a][p]; }
    }
  };
  prefetch_cell(w);

  const double fb = (0.5 * dt) * q / m;
  const double qw = q * mpw;
  const double Aq = 0.5 * dt * dt * mpw * q * q / m;
  unsigned pass = 0;
  WSTAMP_INIT;
  for (int j = 0; j < nch; ++j) {
    const int i = j * kW + w;
    // the particles requested a cell ago are waited for HERE, before the next request goes out: left to the compiler the
    // wait sits at their first use, behind the new request, and -- the memory counter being in order -- covers that too
    // (a full HBM round trip at the head of every cell: 20 % of the producer's time)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    const int start = pf.start, cnt = pf.cnt;
    if (lane < 54) bnb[w][lane] = pf.b;
    double cur[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) cur[a] = pf.p[a];
    asm volatile("" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]), "+v"(cur[5])); // copies made before the request below
    prefetch_cell(i + kW); // the next cell's first particles travel during this cell's passes
    int handed = min(kWsCP, cnt);
    bool real = lane < handed;
    WSTAMP(2);
    for (;;) {
      const bool lastpass = handed >= cnt;
      const int b = pass & 1;
      // both consumers have finished with the buffer's previous content (pass - 2)
      if (pass >= 2) {
        if (!lds_wait(flags + kFlFree + w * 2 + b, pass - 1, flags) ||
            !lds_wait(flags + kFlFree + (kW + w) * 2 + b, pass - 1, flags)) { if (lane == 0) atomicOr(err, kFillErrTimeout); return; }
      }
      WSTAMP(0);
      double* st = stage + (w * 2 + b) * kWsStage;
      wave_sync();
      const W1T<P2> wt(g, cur[0], cur[1], cur[2]);
      const int ox = wt.is[0] - wt.in[0] + 1, oy = wt.is[1] - wt.in[1] + 1, oz = wt.is[2] - wt.in[2] + 1;
      const int oct = real ? (ox | (oy << 1) | (oz << 2)) : 8;
      int ocnt[8], slot = 0;
      bool keep = false;
      {
        int run = 0;
#pragma unroll
        for (int o = 0; o < 8; ++o) {
          const unsigned long long mk = __ballot(oct == o);
          ocnt[o] = __popcll(mk);
          if (oct == o) {
            const int rk = __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
            slot = run + rk;
            keep = !lastpass && rk >= (ocnt[o] & ~3);
          }
          run += ocnt[o];
        }
      }
      // next pass of this cell: the free lanes take the next particles.  Requested NOW, as soon as the lanes that keep
      // their particle are known, so that the loads travel under the rest of this pass (requested at the end of the pass
      // they were a full HBM round trip at the head of the next one)
      // (straight into the lanes' own registers: the position is dead behind the weights above, the velocity is copied first)
      bool real_next = false;
      double v[3] = {cur[3], cur[4], cur[5]};
      asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]));
      if (!lastpass) {
        const unsigned long long km = __ballot(keep);
        const int take = min(cnt - handed, kWsCP - (int)__popcll(km));
        const unsigned long long fm = ~km;
        const int fr = __builtin_amdgcn_mbcnt_hi((unsigned)(fm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)fm, 0u));
        const bool get = !keep && fr < take;
        if (get) {
          const long p = (long)start + handed + fr;
#pragma unroll
          for (int a = 0; a < 3; ++a) { cur[a] = s.r[a][p]; cur[3 + a] = s.v[a][p]; }
        }
        real_next = keep || get;
        handed += take;
      }
      if (real) {
        double2* dst = (double2*)(st + slot * kPitch);
        const double* nb = bnb[w];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int bb = 0; bb < 2; ++bb) {
            const double tx = wt.wn[2][a] * wt.wn[1][bb];
            dst[0 + a * 2 + bb] = double2{tx * wt.ws[0][0], tx * wt.ws[0][1]};
            dst[4 + a * 2 + bb] = double2{wt.wn[2][a] * wt.ws[1][0] * wt.wn[0][bb], wt.wn[2][a] * wt.ws[1][1] * wt.wn[0][bb]};
            dst[8 + a * 2 + bb] = double2{wt.ws[2][0] * wt.wn[1][a] * wt.wn[0][bb], wt.ws[2][1] * wt.wn[1][a] * wt.wn[0][bb]};
          }
        // interpolate_B_s1 out of the cell's LDS neighbourhood, component by component (8 values in flight at a time: the
        // producer has 128 registers and time to spare; all 24 at once were spilled), same product and sum order per component
        double Bp[3] = {0.0, 0.0, 0.0};
        {
          double nv[8];
#pragma unroll
          for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
              for (int ii = 0; ii < 2; ++ii) nv[(k * 2 + jj) * 2 + ii] = nb[((oz + k) * 3 + (oy + jj)) * 2 + ii];
#pragma unroll
          for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
              for (int ii = 0; ii < 2; ++ii) Bp[0] += nv[(k * 2 + jj) * 2 + ii] * (wt.ws[2][k] * wt.ws[1][jj] * wt.wn[0][ii]);
          asm volatile("" ::: "memory");
#pragma unroll
          for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
              for (int ii = 0; ii < 2; ++ii) nv[(k * 2 + jj) * 2 + ii] = nb[18 + ((oz + k) * 2 + jj) * 3 + (ox + ii)];
#pragma unroll
          for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
              for (int ii = 0; ii < 2; ++ii) Bp[1] += nv[(k * 2 + jj) * 2 + ii] * (wt.ws[2][k] * wt.wn[1][jj] * wt.ws[0][ii]);
          asm volatile("" ::: "memory");
#pragma unroll
          for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
              for (int ii = 0; ii < 2; ++ii) nv[(k * 2 + jj) * 2 + ii] = nb[36 + (k * 3 + (oy + jj)) * 3 + (ox + ii)];
#pragma unroll
          for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
              for (int ii = 0; ii < 2; ++ii) Bp[2] += nv[(k * 2 + jj) * 2 + ii] * (wt.wn[2][k] * wt.ws[1][jj] * wt.ws[0][ii]);
        }
        const double bx = Bp[0] * fb, by = Bp[1] * fb, bz = Bp[2] * fb;
        const double b2 = bx * bx + by * by + bz * bz;
        const double vb = v[0] * bx + v[1] * by + v[2] * bz;
        const double cxv = +(v[1] * bz - v[2] * by), cyv = -(v[0] * bz - v[2] * bx), czv = +(v[0] * by - v[1] * bx);
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
      WSTAMP(3);
      if (lane == 0) {
        *(int4*)&hdr[w][b][0] = int4{ocnt[0], ocnt[1], ocnt[2], ocnt[3]};
        *(int4*)&hdr[w][b][4] = int4{ocnt[4], ocnt[5], ocnt[6], ocnt[7]};
        hdr[w][b][8] = lastpass ? 1 : 0;
      }
      lds_post(flags + kFlFull + w * 2 + b, pass + 1, lane); // operands and header are in place
      WSTAMP(4);
      ++pass;
      if (lastpass) break;
      real = real_next;
    }
  }
  WSTAMP_DUMP(16, w == 0);
}

}  // namespace

void launch_fill_ws(xpic_ctx* c, const Sort& s, unsigned nblocks, const double* B, double* currI_sort, double* matL,
  const unsigned short* dtab, const int* linetab, const int* cowr, int cy0, int cys, int ncy, int cz0, int czs, int my_order,
  int ncol_y, int per_y, int per_z, int first_sort, unsigned long long zord)
{
  auto kern = c->g.pow2 ? k_ecsim_fill_ws<true> : k_ecsim_fill_ws<false>;
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(kWsThreads), 0, c->stream, c->g, s.d, B, currI_sort, matL, dtab, linetab, cowr,
    s.par.q, s.par.m, s.par.n / (double)s.par.Np, cy0, cys, ncy, cz0, czs, my_order, ncol_y, per_y, per_z, first_sort, c->fill_err, zord);
}

}  // namespace xpic
