// Issue cost of the fp64 / int VALU instructions the spike pass is made of (gfx950), per wave-instruction per SIMD,
// with W waves per SIMD resident (every CU busy).  Each kernel runs ITERS x 16 independent copies of one instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITERS 2000

#define BODY16(STMT) \
  STMT(0) STMT(1) STMT(2) STMT(3) STMT(4) STMT(5) STMT(6) STMT(7) STMT(8) STMT(9) STMT(10) STMT(11) STMT(12) STMT(13) STMT(14) STMT(15)

template <int OP>
__global__ __launch_bounds__(256) void rate(double* out, double seed) {
  double x[16];
  int n[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    x[i] = seed + 1e-3 * (threadIdx.x + 7 * i);
    n[i] = threadIdx.x + i;
  }
  const double c1 = 0.999999, c2 = 1e-7;
  for (int it = 0; it < ITERS; ++it) {
#define FMA(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c1), "v"(c2));
#define MUL(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[i]) : "v"(c1));
#define ADD(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[i]) : "v"(c2));
#define RND(i) asm volatile("v_rndne_f64 %0, %0" : "+v"(x[i]));
#define CVT(i) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(n[i]) : "v"(x[i]));
#define LDX(i) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(x[i]) : "v"(n[i] & 1));
#define AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(n[i]) : "v"(0x7fffffff));
#define ASH(i) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(n[i]));
#define LSA(i) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(n[i]) : "v"(3));
#define MAX(i) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x[i]) : "v"(c2));
#define F32(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(n[i]) : "v"(1.0f), "v"(0.0f));
#define PKF(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c1), "v"(c2));
#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(n[i]));
    if (OP == 0) { BODY16(FMA) }
    if (OP == 1) { BODY16(MUL) }
    if (OP == 2) { BODY16(ADD) }
    if (OP == 3) { BODY16(RND) }
    if (OP == 4) { BODY16(CVT) }
    if (OP == 5) { BODY16(LDX) }
    if (OP == 6) { BODY16(AND) }
    if (OP == 7) { BODY16(ASH) }
    if (OP == 8) { BODY16(LSA) }
    if (OP == 9) { BODY16(MAX) }
    if (OP == 10) { BODY16(F32) }
    if (OP == 11) { BODY16(PKF) }
    if (OP == 12) { BODY16(EXP) }
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i] + n[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, double* out, int wps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * wps;     // 256 CUs, one 4-wave block per SIMD-quad per wave-per-SIMD wanted
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(rate<OP>, dim3(grid), dim3(256), 0, 0, out, 1.0);
  hipEventRecord(e0);
  const int reps = 5;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(rate<OP>, dim3(grid), dim3(256), 0, 0, out, 1.0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double ns = 1e6 * ms / reps;
  const double per = ns / ((double)ITERS * 16 * wps);    // ns per wave-instruction per SIMD
  printf("%-16s %d waves/SIMD: %.3f ns per wave-instruction per SIMD (%.2f cycles @2.4 GHz)\n", name, wps, per, per * 2.4);
}

int main() {
  double* out;
  hipMalloc(&out, 256 * 8 * 256 * 8);
  for (int wps : {1, 4}) {
    run<0>("v_fma_f64", out, wps);
    run<1>("v_mul_f64", out, wps);
    run<2>("v_add_f64", out, wps);
    run<3>("v_rndne_f64", out, wps);
    run<4>("v_cvt_i32_f64", out, wps);
    run<5>("v_ldexp_f64", out, wps);
    run<9>("v_max_f64", out, wps);
    run<6>("v_and_b32", out, wps);
    run<7>("v_ashrrev_i32", out, wps);
    run<8>("v_lshl_add_u32", out, wps);
    run<10>("v_fma_f32", out, wps);
    run<11>("v_pk_fma_f32", out, wps);
    run<12>("v_exp_f32", out, wps);
    run<0>("v_fma_f64 (again)", out, wps);
  }
  return 0;
}
