// lstencil.h -- fixed-stencil layout of the ECSIM mass matrix matL.
//
// The reference assembles matL as a PETSc AIJ matrix from per-cell 36x36 COO blocks
// (src/impls/ecsim/simulation.cpp:336-469, particles.cpp:145-163).  On a periodic Yee grid the set of
// columns a row can couple to is fixed, so here matL is stored index-free: row (node, c1) holds
// XPIC_LSTENCIL = 123 coefficients, ordered by column component c2 = 0,1,2 and inside each block by the
// node offset d = col_node - row_node:
//     c2 == c1 : d in [-1,1]^3                                              27 entries
//     c2 != c1 : d[c1] in [-1,2], d[c2] in [-2,1], third axis in [-1,1]     48 entries
// (row component c1 is staggered along axis c1, so its 2-node CIC footprint starts at floor(x/dx - 1/2),
// one node below or at the cell's node; hence the asymmetric ranges).
#pragma once

namespace xpic {

struct LRange {
  int lo[3], n[3];
};

__host__ __device__ constexpr LRange lrange(int c1, int c2)
{
  LRange r{};
  for (int a = 0; a < 3; ++a) {
    if (c1 == c2) { r.lo[a] = -1; r.n[a] = 3; }
    else if (a == c1) { r.lo[a] = -1; r.n[a] = 4; }
    else if (a == c2) { r.lo[a] = -2; r.n[a] = 4; }
    else { r.lo[a] = -1; r.n[a] = 3; }
  }
  return r;
}

__host__ __device__ constexpr int lblock_offset(int c1, int c2)
{
  int off = 0;
  for (int c = 0; c < c2; ++c) off += (c == c1) ? 27 : 48;
  return off;
}

__host__ __device__ constexpr int lencode(int c1, int c2, int dx, int dy, int dz)
{
  LRange r = lrange(c1, c2);
  int i = dx - r.lo[0], j = dy - r.lo[1], k = dz - r.lo[2];
  if (i < 0 || i >= r.n[0] || j < 0 || j >= r.n[1] || k < 0 || k >= r.n[2]) return -1;
  return lblock_offset(c1, c2) + (k * r.n[1] + j) * r.n[0] + i;
}

struct LEntry {
  int c2, d[3];
};

__host__ __device__ constexpr LEntry ldecode(int c1, int kk)
{
  LEntry e{};
  for (int c = 0; c < 3; ++c) {
    int sz = (c == c1) ? 27 : 48;
    if (kk < sz) {
      LRange r = lrange(c1, c);
      e.c2 = c;
      e.d[0] = r.lo[0] + kk % r.n[0];
      e.d[1] = r.lo[1] + (kk / r.n[0]) % r.n[1];
      e.d[2] = r.lo[2] + (kk / r.n[0]) / r.n[1];
      return e;
    }
    kk -= sz;
  }
  return e;
}

// Per-cell block numbering of the 3 x 12 nodes a cell's particles can touch, identical to the local
// indices of decompose_ecsim_current (src/impls/ecsim/particles.cpp:145-147):
//   X: (k*2 + j)*3 + i, i in 0..2 <-> node offset (i-1, j, k)
//   Y: (k*3 + j)*2 + i, j in 0..2 <-> node offset (i, j-1, k)
//   Z: (k*2 + j)*2 + i, k in 0..2 <-> node offset (i, j, k-1)
__host__ __device__ constexpr void block_node_offset(int c, int l, int* o)
{
  if (c == 0) { o[0] = l % 3 - 1; o[1] = (l / 3) % 2; o[2] = l / 6; }
  else if (c == 1) { o[0] = l % 2; o[1] = (l / 2) % 3 - 1; o[2] = l / 6; }
  else { o[0] = l % 2; o[1] = (l / 2) % 2; o[2] = l / 4 - 1; }
}

}  // namespace xpic
