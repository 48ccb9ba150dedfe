// What the x-block width of matL's layout [x / BS][k][x % BS] costs a row kernel that streams it: every lane owns one row
// (123 coefficients), a wave 64 consecutive x, and reads coefficient k of its row in the k-th load -- BS doubles of
// consecutive x lie together, so a wave's load touches 64 / BS runs of 8 BS bytes.  BS = 4 is the layout the assembly
// writes (32-byte pieces); 16 makes every run a full 128-byte line.  Pure streaming, no operand vector.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int K = 123;
template <int BS>
__global__ void __launch_bounds__(256, 4) k_rows(const double* __restrict__ L, long nrowblocks, double* out)
{
  const int lane = threadIdx.x & 63;
  const long rb = (long)blockIdx.x * 4 + (threadIdx.x >> 6); // one 64-row block per wave
  if (rb >= nrowblocks) return;
  const double* p = L + rb * 64 * K + (long)(lane / BS) * (K * BS) + lane % BS;
  double a0 = 0, a1 = 0;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const double v = p[(long)k * BS];
    if (k & 1) a1 += v; else a0 += v;
  }
  if (a0 + a1 == 12345.678) out[0] = a0;
}
template <int BS>
static void run(const double* L, long nrb, double* out)
{
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const unsigned grid = (unsigned)((nrb + 3) / 4);
  k_rows<BS><<<grid, 256>>>(L, nrb, out); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) k_rows<BS><<<grid, 256>>>(L, nrb, out);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)nrb * 64 * K * 8;
  printf("x-block of %2d: %.2f ms per pass, %.1f GB/s\n", BS, ms / 3, 3.0 * bytes / (ms * 1e-3) / 1e9);
}
int main()
{
  const long rows = 3L * 256 * 256 * 256; // the headline's matL: 49.5 GB
  const long nrb = rows / 64;
  double *L, *out;
  if (hipMalloc(&L, sizeof(double) * rows * K) != hipSuccess) { printf("no memory\n"); return 1; }
  (void)hipMalloc(&out, 8);
  (void)hipMemset(L, 0, sizeof(double) * rows * K);
  for (int rep = 0; rep < 2; ++rep) { run<4>(L, nrb, out); run<8>(L, nrb, out); run<16>(L, nrb, out); run<64>(L, nrb, out); }
  return 0;
}
