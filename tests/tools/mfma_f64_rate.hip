// Micro-probe: cycles per v_mfma_f64_16x16x4_f64, dependent chain vs 2/4 independent accumulators,
// 1 wave per SIMD (256-thread block on one CU) and 2 waves per SIMD (512 threads).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void rate(double* out, long long* cyc, int iters) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = 1.0 + threadIdx.x * 1e-3, b = 0.5;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// VALU fp64 fma rate for comparison
__global__ void vrate(double* out, long long* cyc, int iters) {
  double x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7, c = 1.0000001, d = 1e-9;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    x0 = fma(x0, c, d); x1 = fma(x1, c, d); x2 = fma(x2, c, d); x3 = fma(x3, c, d);
    x4 = fma(x4, c, d); x5 = fma(x5, c, d); x6 = fma(x6, c, d); x7 = fma(x7, c, d);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  double* out; long long* cyc; long long h[4];
  (void)hipMalloc(&out, 8 * 1024 * 256); (void)hipMalloc(&cyc, 8 * 1024);
  const int iters = 4096;
  // s_memtime counts at a fixed 100 MHz; convert with wall clock of a long run instead: use events
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto run = [&](auto kern, int threads, int nacc, const char* name) {
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, cyc, iters);  // warm
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double per_wave_ns = ms * 1e6 / ((double)iters * nacc);
    printf("%-28s threads/blk %3d : %.1f ns per instr per wave  (%.1f cycles @2.4GHz)\n", name, threads, per_wave_ns, per_wave_ns * 2.4);
  };
  run(rate<1>, 256, 1, "mfma dependent x1");
  run(rate<2>, 256, 2, "mfma 2 accumulators");
  run(rate<4>, 256, 4, "mfma 4 accumulators");
  run(rate<1>, 512, 1, "mfma dependent x1, 2w/SIMD");
  run(rate<4>, 512, 4, "mfma 4 acc, 2w/SIMD");
  run(vrate, 256, 8, "valu fma64 x8 indep");
  run(vrate, 512, 8, "valu fma64, 2w/SIMD");
  run(vrate, 1024, 8, "valu fma64, 4w/SIMD");
  return 0;
}
