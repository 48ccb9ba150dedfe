// How fast does the memory system move the particle kernels' access pattern?  n-stream structure-of-arrays copies
// (read S arrays, write S arrays, 8 B per lane and stream -- the shape of k_scatter without the permutation and of
// k_second_push) against the one-stream copy that DESIGN.md quotes as the device copy rate.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int S, int V>
__global__ void __launch_bounds__(256) k_copy(const double* __restrict__ in, double* __restrict__ out, long n, long pitch)
{
  const long stride = (long)gridDim.x * 256 * V;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * V; i < n; i += stride) {
    double v[S][V];
#pragma unroll
    for (int a = 0; a < S; ++a)
#pragma unroll
      for (int e = 0; e < V; ++e) v[a][e] = in[a * pitch + i + e];
#pragma unroll
    for (int a = 0; a < S; ++a)
#pragma unroll
      for (int e = 0; e < V; ++e) out[a * pitch + i + e] = v[a][e];
  }
}
template <int S, int V>
static void run(const double* in, double* out, long n, long pitch, int blocks)
{
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k_copy<S, V><<<blocks, 256>>>(in, out, n, pitch); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) k_copy<S, V><<<blocks, 256>>>(in, out, n, pitch);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%d streams, %2d B per lane and stream, %6d blocks: %.2f TB/s (read + write)\n", S, 8 * V, blocks, 3.0 * 2 * S * n * 8 / (ms * 1e-3) / 1e12);
}
int main()
{
  const long n = 1L << 28, pitch = n + 4096; // 2 GiB per stream
  double *in, *out;
  (void)hipMalloc(&in, sizeof(double) * 6 * pitch); (void)hipMalloc(&out, sizeof(double) * 6 * pitch);
  (void)hipMemset(in, 0, sizeof(double) * 6 * pitch); (void)hipMemset(out, 0, sizeof(double) * 6 * pitch);
  for (int blocks : {4096, 16384, 65536}) {
    run<1, 1>(in, out, n, pitch, blocks);
    run<1, 2>(in, out, n, pitch, blocks);
    run<3, 1>(in, out, n, pitch, blocks);
    run<6, 1>(in, out, n, pitch, blocks);
    run<6, 2>(in, out, n, pitch, blocks);
  }
  return 0;
}
