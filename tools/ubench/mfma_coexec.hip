// do v_mfma_f64_16x16x4_f64 and v_fma_f64 co-execute on one SIMD of gfx950?  8-wave workgroups, one per CU:
// waves 0-3 run role A, waves 4-7 role B (wave k and k+4 share a SIMD).  Roles: 0 idle, 1 MFMA f64, 2 FMA f64, 3 MFMA bf16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ double run_mfma(int n, double a, double b)
{
  double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  for (int i = 0; i < n; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  return c0[0] + c1[1] + c2[2] + c3[3];
}
__device__ double run_fma(int n, double a, double b)
{
  double c[16];
  for (int j = 0; j < 16; ++j) c[j] = j;
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) c[j] = __builtin_fma(a, b, c[j]);
  }
  double s = 0;
  for (int j = 0; j < 16; ++j) s += c[j];
  return s;
}
__device__ double run_fma32(int n, float a, float b)
{
  float c[16];
  for (int j = 0; j < 16; ++j) c[j] = j;
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) c[j] = __builtin_fmaf(a, b, c[j]);
  }
  float s = 0;
  for (int j = 0; j < 16; ++j) s += c[j];
  return s;
}
__global__ void __launch_bounds__(512) k(double* out, int roleA, int roleB, int nA, int nB)
{
  const int wave = threadIdx.x >> 6;
  const int role = wave < 4 ? roleA : roleB, n = wave < 4 ? nA : nB;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4, r = 0;
  if (role == 1) r = run_mfma(n, a, b);
  else if (role == 2) r = run_fma(n, a, b);
  else if (role == 3) r = run_fma32(n, (float)a, (float)b);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
static float timeit(double* out, int ra, int rb, int na, int nb)
{
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<<<256, 512>>>(out, ra, rb, na, nb); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); k<<<256, 512>>>(out, ra, rb, na, nb); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main()
{
  double* out; (void)hipMalloc(&out, 1 << 24);
  const int nm = 20000, nf = 4 * nm * 16 / 16; // MFMA: 4*nm instr of 64 cycles; FMA loop: 16*nf instr of 4+ cycles
  printf("MFMA f64 alone        %.3f ms\n", timeit(out, 1, 0, nm, 0));
  printf("FMA f64 alone         %.3f ms\n", timeit(out, 0, 2, 0, nf));
  printf("MFMA f64 + FMA f64    %.3f ms  (sum = no co-execution, max = full co-execution)\n", timeit(out, 1, 2, nm, nf));
  printf("FMA f32 alone         %.3f ms\n", timeit(out, 0, 3, 0, nf));
  printf("MFMA f64 + FMA f32    %.3f ms\n", timeit(out, 1, 3, nm, nf));
  printf("FMA f64 + FMA f64     %.3f ms\n", timeit(out, 2, 2, nf, nf));
  printf("MFMA f64 + MFMA f64   %.3f ms\n", timeit(out, 1, 1, nm, nm));
  return 0;
}
