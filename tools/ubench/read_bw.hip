// read-only HBM bandwidth ceiling: grid-stride sum of a 16 GiB array with 8- and 16-byte loads per lane
#include <hip/hip_runtime.h>
#include <cstdio>
template <class T>
__global__ void __launch_bounds__(256) k_sum(const T* __restrict__ p, long n, double* out)
{
  double acc = 0;
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const T v = p[i];
    if constexpr (sizeof(T) == 8) acc += v; else acc += v.x + v.y;
  }
  if (acc == 12345.678) out[0] = acc;
}
template <class T>
static void run(const void* p, size_t bytes, double* out, int blocks, const char* name)
{
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const long n = bytes / sizeof(T);
  k_sum<T><<<blocks, 256>>>((const T*)p, n, out); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) k_sum<T><<<blocks, 256>>>((const T*)p, n, out);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%s, %d blocks: %.1f GB/s\n", name, blocks, 3.0 * bytes / (ms * 1e-3) / 1e9);
}
int main()
{
  const size_t bytes = 16ull << 30;
  void* p; double* out;
  (void)hipMalloc(&p, bytes); (void)hipMalloc(&out, 8); (void)hipMemset(p, 0, bytes);
  for (int blocks : {2048, 8192, 32768}) {
    run<double>(p, bytes, out, blocks, "8 B per lane ");
    run<double2>(p, bytes, out, blocks, "16 B per lane");
  }
  return 0;
}
