// Do stores whose 512-byte pieces (one wave, 64 lanes x 8 B) start off the 128-byte line grid -- the sorted copy the assembly
// and the gathering push write cell by cell -- cost HBM READS (a fetch of the lines a piece covers partly) even though
// neighbouring waves write the rest of those lines?  And do loads that are a permutation inside a wave's 512-byte window
// (the gather through a source index) fetch more than sequential ones?  Run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
// (tools/ubench/partial_lines.sh): kernels k<SHIFT, PERM, STAGGER>.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int SHIFT, bool PERM, int STAGGER>
__global__ void __launch_bounds__(256) k(const double* __restrict__ in, double* __restrict__ out, long n)
{
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
  const long base = wave * 64 + SHIFT;
  if (base + 64 > n) return;
  const int src = PERM ? (lane * 37 + 11) & 63 : lane;
  double v = in[base + src];
  if (STAGGER) { // waves of a workgroup reach their store at different times (as cells of different sizes do)
    const int w = threadIdx.x >> 6;
    for (int i = 0; i < STAGGER * w; ++i) v = v * 1.0000001 + 1e-30;
  }
  out[base + lane] = v;
}
template <int SHIFT, bool PERM, int STAGGER>
static void run(const double* in, double* out, long n)
{
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const unsigned blocks = (unsigned)((n / 64 + 3) / 4);
  k<SHIFT, PERM, STAGGER><<<blocks, 256>>>(in, out, n); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<SHIFT, PERM, STAGGER><<<blocks, 256>>>(in, out, n);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("shift %d perm %d stagger %4d: %.3f ms, %.2f TB/s (read + write of %ld doubles)\n", SHIFT, (int)PERM, STAGGER, ms, 16.0 * n / (ms * 1e-3) / 1e12, n);
}
int main()
{
  const long n = 1L << 28;
  double *in, *out;
  (void)hipMalloc(&in, 8 * (n + 64)); (void)hipMalloc(&out, 8 * (n + 64));
  (void)hipMemset(in, 0, 8 * (n + 64)); (void)hipMemset(out, 0, 8 * (n + 64));
  run<0, false, 0>(in, out, n);
  run<3, false, 0>(in, out, n);
  run<3, false, 2000>(in, out, n);
  run<0, true, 0>(in, out, n);
  run<3, true, 2000>(in, out, n);
  return 0;
}
