// issue cost of v_mfma_f64_16x16x4_f64 and of v_fma_f64 on one wave per SIMD (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void k_mfma(double* out, long long* cyc, int n)
{
  double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  long long t0 = clock64();
  for (int i = 0; i < n; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_fma(double* out, long long* cyc, int n)
{
  double c[16];
  for (int j = 0; j < 16; ++j) c[j] = j;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  long long t0 = clock64();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) c[j] = __builtin_fma(a, b, c[j]);
  }
  long long t1 = clock64();
  double s = 0;
  for (int j = 0; j < 16; ++j) s += c[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
  double* out; long long* cyc;
  hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 1 << 16);
  const int n = 20000;
  for (int waves = 1; waves <= 2; ++waves) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    long long h[4];
    k_mfma<<<256 * 4, 64 * waves>>>(out, cyc, n); hipDeviceSynchronize();
    hipEventRecord(e0); k_mfma<<<256 * 4, 64 * waves>>>(out, cyc, n); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
    printf("mfma_f64_16x16x4: %d wave(s)/WG, 4 WG/CU: %.3f ms, %.1f clock64 ticks per MFMA, %.1f TFLOP/s\n", waves, ms,
      (double)h[0] / (4.0 * n), 1024.0 * waves * 4 * 2048.0 * n / (ms * 1e-3) / 1e12);
    k_fma<<<256 * 4, 64 * waves>>>(out, cyc, n); hipDeviceSynchronize();
    hipEventRecord(e0); k_fma<<<256 * 4, 64 * waves>>>(out, cyc, n); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
    printf("v_fma_f64: %d wave(s)/WG: %.3f ms, %.2f ticks per FMA instr, %.1f TFLOP/s\n", waves, ms, (double)h[0] / (16.0 * n),
      1024.0 * waves * 128.0 * 16 * n / (ms * 1e-3) / 1e12);
  }
  return 0;
}
