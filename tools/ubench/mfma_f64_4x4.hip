// issue cost and operand layout of v_mfma_f64_4x4x4_4b_f64 on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_rate(double* out, long long* cyc, int n)
{
  double c[8];
  for (int j = 0; j < 8; ++j) c[j] = j;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  long long t0 = clock64();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[j], 0, 0, 0);
  }
  long long t1 = clock64();
  double s = 0;
  for (int j = 0; j < 8; ++j) s += c[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// layout probe: A = one-hot at lane la, B = one-hot at lane lb -> which D lane becomes non-zero
__global__ void k_layout(int la, int lb, double* d)
{
  const int l = threadIdx.x;
  double a = l == la ? 1.0 : 0.0, b = l == lb ? 1.0 : 0.0;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
}
int main()
{
  double* out; long long* cyc;
  (void)hipMalloc(&out, 1 << 24); (void)hipMalloc(&cyc, 1 << 16);
  const int n = 20000;
  k_rate<<<1024, 64>>>(out, cyc, n); (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0); k_rate<<<1024, 64>>>(out, cyc, n); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("mfma_f64_4x4x4: %.3f ms, %.2f ticks per instruction, %.1f TFLOP/s\n", ms, (double)h / (8.0 * n),
    1024.0 * 8 * n * 512.0 / (ms * 1e-3) / 1e12);
  double* d; (void)hipMalloc(&d, 64 * 8);
  double hd[64];
  // A lane la = (block ba, ?, ?), B lane lb: scan a few combinations inside block 0 and across blocks
  for (int la = 0; la < 64; la += 1) {
    for (int lb = 0; lb < 64; lb += 1) {
      k_layout<<<1, 64>>>(la, lb, d); (void)hipMemcpy(hd, d, sizeof(hd), hipMemcpyDeviceToHost);
      for (int l = 0; l < 64; ++l)
        if (hd[l] != 0.0 && (la < 20 && lb < 20)) printf("A lane %2d x B lane %2d -> D lane %2d\n", la, lb, l);
    }
  }
  return 0;
}
