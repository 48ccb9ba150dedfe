// esirkepov.hip -- the particle phases that deposit the charge-conserving Esirkepov current:
//   MODE 0  basic::Particles::push            (src/impls/basic/particles.cpp:17-53)
//   MODE 1  ecsimcorr::Particles::first_push  (src/impls/ecsimcorr/particles.cpp:27-50)
//   MODE 2  ecsimcorr::Particles::second_push (src/impls/ecsimcorr/particles.cpp:52-91)
// with Shape (src/utils/shape.cpp:31-80), SimpleInterpolation (src/algorithms/simple_interpolation.cpp:8-38)
// and EsirkepovDecomposition (src/algorithms/esirkepov_decomposition.cpp:20-103) for cell-sorted SoA particles.
//
// A workgroup owns 4 consecutive cells in x (one wave per cell).  A particle that starts in cell c and moves
// at most one cell (the reference's own limit: a larger move overflows its shape[384] scratch, shape.h:18,91-92)
// touches nodes c-2 .. c+3 per axis only, so
//   * the E/B nodes the 2nd-order gather can need are staged once per workgroup in an LDS tile (9 x 6 x 6 nodes);
//   * phase 1 (lane = particle) moves/pushes the particle and stages its 1-D old/new spline values on those
//     6 nodes per axis (zero outside the support, exactly as spline_of_2nd_order returns);
//   * phase 2 (lane = two of the 108 "lines" of the cell: a line is one (component, two transverse node
//     indices) and runs along the component's own axis) rebuilds the reference's running sums
//     temp_j[line] = temp_j[line] + W (:57-103) in registers, particle after particle: the J of the whole cell
//     accumulates with no atomics at all;
//   * the 4 cells' lines are merged in an LDS J tile and leave with one fp64 atomic per tile node, in x-runs.
#include <cstring>

#include "common.h"
#include "device_common.h"

namespace xpic {

namespace {

constexpr int kBW = 4;            // cells (waves) per workgroup along x
constexpr int kBC = 32;           // particles staged per pass and wave
constexpr int kBPad = kBC + 1;
constexpr int kT = 6;             // nodes per axis a cell's particles can touch: c-2 .. c+3
constexpr int kTX = kBW + kT - 1; // tile nodes along x
constexpr int kSRows = 54;        // So[3][6], Sn[3][6], D[3][6]
constexpr int kThreadsB = kBW * 64;
constexpr int kTileN = kTX * kT * kT; // nodes of the workgroup tile

__device__ inline void wave_sync_b()
{
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// spline_of_2nd_order (src/interfaces/sort_parameters.cpp:21-30)
__device__ inline double spline2(double s)
{
  s = fabs(s);
  if (s <= 0.5) return (0.75 - s * s);
  if (0.5 < s && s < 1.5) return 0.5 * (1.5 - s) * (1.5 - s);
  return 0.0;
}

__device__ inline double wave_sum_b(double v)
{
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

template <int MODE>
__global__ void __launch_bounds__(kThreadsB) k_esirkepov_push(GridDev g, SortDev s, const double* __restrict__ E,
  const double* __restrict__ B, double* __restrict__ J, double qm, double alpha, double qn_Np, double* pred_w,
  int* bad_count)
{
  // workgroup -> (x chunk, cy, cz)
  const int nxc = (g.nx + kBW - 1) / kBW;
  const int xc = blockIdx.x % nxc;
  const int cy = (blockIdx.x / nxc) % g.ny;
  const int cz = blockIdx.x / (nxc * g.ny);
  const int cx0 = xc * kBW;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int cx = cx0 + wave;
  const bool active = cx < g.nx;

  __shared__ double stage[kBW][kSRows * kBPad];
  __shared__ double jtile[3 * kTileN];
  __shared__ double ftile[MODE == 0 ? 6 * kTileN : 1]; // Ex,Ey,Ez,Bx,By,Bz on the tile nodes (basic only)

  for (int t = threadIdx.x; t < 3 * kTileN; t += kThreadsB) jtile[t] = 0.0;
  if (MODE == 0) {
    // DMGlobalToLocal(E), (B) (basic/simulation.cpp:56-57) for just the nodes this workgroup can gather from
    for (int t = threadIdx.x; t < 6 * kTileN; t += kThreadsB) {
      const int f = t / kTileN, n = t % kTileN;
      const int tx = n % kTX, ty = (n / kTX) % kT, tz = n / (kTX * kT);
      const double* F = (f < 3 ? E : B) + (f % 3) * g.cstride;
      ftile[t] = F[g.nodew(cx0 - 2 + tx, cy - 2 + ty, cz - 2 + tz)];
    }
  }
  __syncthreads();

  double acc[2][kT];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int t = 0; t < kT; ++t) acc[r][t] = 0.0;
  double pw = 0.0;
  int bad = 0;
  double* st = stage[wave];
  const double dt = g.dt;
  const double dd[3] = {g.dx, g.dy, g.dz};
  const int cc[3] = {cx, cy, cz + g.z0};

  // the two lines of this lane
  int lcomp[2], lu[2], lw[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int l = lane + r * 54;
    lcomp[r] = l / 36;
    lu[r] = (l % 36) % kT;
    lw[r] = (l % 36) / kT;
  }
  const bool has_lines = lane < 54;

  if (active) {
    const long cell = ((long)cz * g.ny + cy) * g.nx + cx;
    const int start = s.cell_start[cell];
    const int cnt = s.cell_start[cell + 1] - start;
    for (int base = 0; base < cnt; base += kBC) {
      const int mcnt = min(kBC, cnt - base);
      wave_sync_b();
      if (lane < mcnt) {
        const long p = (long)start + base + lane;
        double r[3] = {s.r[0][p], s.r[1][p], s.r[2][p]};
        double v[3] = {s.v[0][p], s.v[1][p], s.v[2][p]};
        const double old_r[3] = {r[0], r[1], r[2]};
        double Ep[3] = {0, 0, 0}, Bp[3] = {0, 0, 0};

        if (MODE == 0) {
          // push.update_r(dt / 2) ; shape.setup(point.r) ; interpolation.process   (basic/particles.cpp:31-38)
#pragma unroll
          for (int a = 0; a < 3; ++a) r[a] += v[a] * (dt / 2.0);
          int sst[3], ssz[3];
          double No[3][4], Sh[3][4];
          bool inside = true;
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            const double pr = r[a] / dd[a];
            sst[a] = (int)round(pr - 1.5);               // Shape::make_start, shape.cpp:12-19
            ssz[a] = (int)floor(pr + 1.5) + 1 - sst[a];  // Shape::make_end, :21-28
            inside = inside && sst[a] >= cc[a] - 2 && sst[a] + ssz[a] <= cc[a] + 4;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const double gx = (double)(sst[a] + t);
              No[a][t] = spline2(pr - gx);               // Shape::fill, :57-80
              Sh[a][t] = spline2(pr - (gx + 0.5));
            }
          }
          if (inside) {
            const int ox = sst[0] - (cx0 - 2), oy = sst[1] - (cy - 2), oz = sst[2] - (cz + g.z0 - 2);
            // the reference's loop order (x fastest, then y, z); at most 4 nodes per axis, fully unrolled so that
            // the weight arrays stay in registers
#pragma unroll
            for (int kz = 0; kz < 4; ++kz)
#pragma unroll
              for (int jy = 0; jy < 4; ++jy)
#pragma unroll
                for (int ix = 0; ix < 4; ++ix) {
                  if (kz < ssz[2] && jy < ssz[1] && ix < ssz[0]) {
                    const int n = ((oz + kz) * kT + (oy + jy)) * kTX + (ox + ix);
                    // Shape::electric / magnetic (shape.h:54-72)
                    Ep[0] += ftile[0 * kTileN + n] * (No[2][kz] * No[1][jy] * Sh[0][ix]);
                    Ep[1] += ftile[1 * kTileN + n] * (No[2][kz] * Sh[1][jy] * No[0][ix]);
                    Ep[2] += ftile[2 * kTileN + n] * (Sh[2][kz] * No[1][jy] * No[0][ix]);
                    Bp[0] += ftile[3 * kTileN + n] * (Sh[2][kz] * Sh[1][jy] * No[0][ix]);
                    Bp[1] += ftile[4 * kTileN + n] * (Sh[2][kz] * No[1][jy] * Sh[0][ix]);
                    Bp[2] += ftile[5 * kTileN + n] * (No[2][kz] * Sh[1][jy] * Sh[0][ix]);
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
          // interpolate_E_s1 / interpolate_B_s1 ; update_vEB(dt) ; update_r(0.5 dt)   (:64-69)
          const double old_v[3] = {v[0], v[1], v[2]};
          const W1 w(g, r[0], r[1], r[2]);
          gather_s1(g, E, B, w, Ep, Bp);
          update_vEB(dt, qm, Ep, Bp, v);
#pragma unroll
          for (int a = 0; a < 3; ++a) r[a] += v[a] * (0.5 * dt);
          // pred_w += qn_Np * 0.5 * (old_v + point.p).dot(E_p)   (:77-78)
          pw += qn_Np * 0.5 * ((old_v[0] + v[0]) * Ep[0] + (old_v[1] + v[1]) * Ep[1] + (old_v[2] + v[2]) * Ep[2]);
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          s.r[a][p] = r[a];
          if (MODE != 1) s.v[a][p] = v[a];
        }

        // Shape::setup(old_r, new_r) (shape.cpp:43-54): range test, then the 1-D spline values on the 6 tile nodes
        bool ok = true;
        double* col = st + lane;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const double po = old_r[a] / dd[a], pn = r[a] / dd[a];
          const int sst = (int)round(fmin(po, pn) - 1.5);
          const int send = (int)floor(fmax(po, pn) + 1.5) + 1;
          ok = ok && (send - sst <= 4) && sst >= cc[a] - 2 && send <= cc[a] + 4;
#pragma unroll
          for (int t = 0; t < kT; ++t) {
            const double gx = (double)(cc[a] - 2 + t);
            const double so = spline2(po - gx), sn = spline2(pn - gx);
            col[(a * kT + t) * kBPad] = so;
            col[(18 + a * kT + t) * kBPad] = sn;
            col[(36 + a * kT + t) * kBPad] = sn - so;
          }
        }
        if (!ok) {
          // the reference would overflow Shape::shape here; deposit nothing and report
          ++bad;
#pragma unroll
          for (int e = 0; e < kSRows; ++e) col[e * kBPad] = 0.0;
        }
      }
      wave_sync_b();

      if (has_lines) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          // component c runs along its own axis c; transverse axes (A, B) with the reference's roles:
          //   X: A = y (index u), B = z (index w)   get_jx :57-71
          //   Y: A = x (index u), B = z (index w)   get_jy :73-87
          //   Z: A = y (index w), B = x (index u)   get_jz :89-103
          const int c = lcomp[rr];
          const int axA = c == 1 ? 0 : 1, axB = c == 2 ? 0 : 2;
          const int iA = c == 2 ? lw[rr] : lu[rr], iB = c == 2 ? lu[rr] : lw[rr];
          const double qd = alpha * (c == 0 ? g.dx : (c == 1 ? g.dy : g.dz));
          const double* soA = st + (axA * kT + iA) * kBPad;
          const double* snA = st + (18 + axA * kT + iA) * kBPad;
          const double* soB = st + (axB * kT + iB) * kBPad;
          const double* snB = st + (18 + axB * kT + iB) * kBPad;
          const double* dC = st + (36 + c * kT) * kBPad;
          for (int p = 0; p < mcnt; ++p) {
            // W = -qd * D_c[t] * T with T = Sn_A (2 Sn_B + So_B) + So_A (2 So_B + Sn_B): the factor common to the six
            // nodes of the line is formed once (one rounding apart from the reference's (-qd * D) * T)
            const double T = -qd * (snA[p] * (2.0 * snB[p] + soB[p]) + soA[p] * (2.0 * soB[p] + snB[p]));
            double run = 0.0;
#pragma unroll
            for (int t = 0; t < kT; ++t) {
              run = run + dC[t * kBPad + p] * T; // temp_j = temp_j + w_p
              acc[rr][t] += run;
            }
          }
        }
      }
    }
  }

  // ---- merge the workgroup's cells in the LDS J tile, then one atomic per tile node
  __syncthreads();
  if (active && has_lines) {
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int c = lcomp[rr];
#pragma unroll
      for (int t = 0; t < kT; ++t) {
        int tx, ty, tz;
        if (c == 0) { tx = wave + t; ty = lu[rr]; tz = lw[rr]; }
        else if (c == 1) { tx = wave + lu[rr]; ty = t; tz = lw[rr]; }
        else { tx = wave + lu[rr]; ty = lw[rr]; tz = t; }
        if (acc[rr][t] != 0.0) unsafeAtomicAdd(&jtile[c * kTileN + (tz * kT + ty) * kTX + tx], acc[rr][t]);
      }
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < 3 * kTileN; t += kThreadsB) {
    const double val = jtile[t];
    if (val == 0.0) continue;
    const int c = t / kTileN, n = t % kTileN;
    const int tx = n % kTX, ty = (n / kTX) % kT, tz = n / (kTX * kT);
    unsafeAtomicAdd(&J[c * g.cstride + g.nodew(cx0 - 2 + tx, cy - 2 + ty, cz - 2 + tz)], val);
  }
  if (MODE == 2) {
    pw = wave_sum_b(pw);
    if (lane == 0 && pw != 0.0) unsafeAtomicAdd(pred_w, pw);
  }
  if (bad) atomicAdd(bad_count, bad);
}

}  // namespace

int esirkepov_push(xpic_ctx* c, Sort& s, int mode, const double* E, const double* B, double* J, double* pred_w_host)
{  s.prebinned = false;

  if (pred_w_host) *pred_w_host = 0.0;
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
  const GridDev& g = c->g;
  XPIC_CHECK(g.nx >= 6 && g.ny >= 6 && g.nzl >= 6, "the Esirkepov tile needs every grid extent >= 6");
  const double qm = s.par.q / s.par.m;
  const double qn_Np = s.par.q * s.par.n / s.par.Np;
  const double alpha = qn_Np / (6.0 * g.dt); // basic/particles.cpp:44, ecsimcorr/particles.cpp:130
  double* scal = c->red_out;                  // [0] pred_w, [1] bad count (as int)
  XPIC_HIP(hipMemsetAsync(scal, 0, sizeof(double) * 2, c->stream));
  const long nxc = (g.nx + kBW - 1) / kBW;
  const long nblocks = nxc * g.ny * g.nzl;
  XPIC_CHECK(nblocks < 2147483647L, "too many cells for one launch");
  const char* name = mode == 0 ? "basic_push" : (mode == 1 ? "corr_first_push" : "corr_second_push");
  {
    Timed t(c, name);
    dim3 grid((unsigned)nblocks), block(kThreadsB);
    if (mode == 0) hipLaunchKernelGGL(k_esirkepov_push<0>, grid, block, 0, c->stream, g, s.d, E, B, J, qm, alpha, qn_Np, scal, (int*)(scal + 1));
    else if (mode == 1) hipLaunchKernelGGL(k_esirkepov_push<1>, grid, block, 0, c->stream, g, s.d, E, B, J, qm, alpha, qn_Np, scal, (int*)(scal + 1));
    else hipLaunchKernelGGL(k_esirkepov_push<2>, grid, block, 0, c->stream, g, s.d, E, B, J, qm, alpha, qn_Np, scal, (int*)(scal + 1));
    XPIC_HIP(hipGetLastError());
  }
  XPIC_HIP(hipMemcpyAsync(c->red_host, scal, sizeof(double) * 2, hipMemcpyDeviceToHost, c->stream));
  XPIC_HIP(hipStreamSynchronize(c->stream));
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
