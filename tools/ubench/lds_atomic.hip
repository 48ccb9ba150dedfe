// LDS read-modify-write rates on gfx950: what one wave-instruction costs for ds_add_f64 (the assembly's merge),
// ds_add_u64, ds_add_f32, ds_write_b64 and the plain sequence ds_read_b64 / v_add_f64 / ds_write_b64, conflict-free
// (lane l -> byte 8 l of its wave's region) and with the assembly window's stride (56 bytes between lanes), at 1, 4, 8
// and 16 waves per CU.  Cycles are s_memtime ticks of one wave for REPS back-to-back instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int REPS = 512;
template <int MODE>
__global__ void __launch_bounds__(1024) k(int stride8, unsigned long long* out, double* sink)
{
  __shared__ double buf[16 * 64 * 8 + 64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16 * 64 * 8 + 64; i += blockDim.x) buf[i] = 0.0;
  __syncthreads();
  double* p = buf + wave * 64 * (stride8 > 8 ? 1 : 1) * 8 + lane * stride8 % (64 * 8);
  const double v = 1.0 + lane;
  double acc = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 16
  for (int r = 0; r < REPS; ++r) {
    if (MODE == 0) unsafeAtomicAdd(p, v);
    else if (MODE == 1) atomicAdd((unsigned long long*)p, (unsigned long long)lane);
    else if (MODE == 2) unsafeAtomicAdd((float*)p, (float)v);
    else if (MODE == 3) { *(volatile double*)p = v + r; }
    else { const double o = *(volatile double*)p; *(volatile double*)p = o + v; }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
  if (acc == 1234.5) sink[0] = acc + buf[lane];
}
template <int MODE>
static void run(const char* name, int stride8)
{
  unsigned long long* out; double* sink;
  (void)hipMalloc(&out, 256 * 16 * 8); (void)hipMalloc(&sink, 8);
  printf("%-34s stride %2d B:", name, stride8 * 8);
  for (int waves : {1, 4, 8, 16}) {
    k<MODE><<<256, waves * 64>>>(stride8, out, sink);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 16);
    (void)hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; int n = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) { s += (double)h[b * 16 + w]; ++n; }
    printf("  %2d waves %7.1f cyc/instr", waves, s / n / REPS);
  }
  printf("\n");
  (void)hipFree(out); (void)hipFree(sink);
}
int main()
{
  for (int stride8 : {1, 7}) {
    run<0>("ds_add_f64", stride8);
    run<1>("ds_add_u64", stride8);
    run<2>("ds_add_f32", stride8);
    run<3>("ds_write_b64", stride8);
    run<4>("ds_read_b64 + v_add_f64 + ds_write", stride8);
  }
  return 0;
}
