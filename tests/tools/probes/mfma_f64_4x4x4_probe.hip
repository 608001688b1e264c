// Probe of v_mfma_f64_4x4x4_4b_f64 on gfx950 (diagnosis tool, not part of the library): operand / result lane layout by
// one-hot operands, and issue cost against v_mfma_f64_16x16x4_f64.  hipcc --offload-arch=gfx950 -O2 -o probe probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k_layout(double* out) {             // out[la][lb][ld]
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const double a = (lane == la) ? 1.0 : 0.0, b = (lane == lb) ? 1.0 : 0.0;
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      out[(la * 64 + lb) * 64 + lane] = d;
    }
}

template <int KIND>
__global__ void k_rate(double* out, int iters, long long* cyc) {
  const int lane = threadIdx.x & 63;
  double a = 1.0 + lane * 1e-3, b = 1.0 - lane * 1e-3;
  long long t0 = 0, t1 = 0;
  if (KIND == 0) {
    d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
    }
    t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
  } else {
    double acc[4] = {0, 0, 0, 0};
    t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[u], 0, 0, 0);
    }
    t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[KIND] = t1 - t0;
}

int main() {
  double* d;
  long long* c;
  hipMalloc(&d, 64 * 64 * 64 * sizeof(double));
  hipMalloc(&c, 2 * sizeof(long long));
  k_layout<<<1, 64>>>(d);
  std::vector<double> h(64 * 64 * 64);
  hipMemcpy(h.data(), d, h.size() * sizeof(double), hipMemcpyDeviceToHost);
  // (la, lb) -> ld list
  int n = 0;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb)
      for (int ld = 0; ld < 64; ++ld)
        if (h[(la * 64 + lb) * 64 + ld] != 0.0) {
          if (la < 20 || (la % 16 == 0)) printf("A lane %2d  B lane %2d -> D lane %2d\n", la, lb, ld);
          ++n;
        }
  printf("nonzero triples: %d (expected 4 blocks * 4 k * 4 i * 4 j = 256)\n", n);
  // hypothesis: lane = x + 4 block + 16 k for A (x = i) and B (x = j); D lane = j + 4 block + 16 i
  int bad = 0;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb)
      for (int ld = 0; ld < 64; ++ld) {
        const int ia = la & 3, ba = (la >> 2) & 3, ka = la >> 4, jb = lb & 3, bb = (lb >> 2) & 3, kb = lb >> 4;
        const int jd = ld & 3, bd = (ld >> 2) & 3, id = ld >> 4;
        const bool exp = ba == bb && ka == kb && bd == ba && id == ia && jd == jb;
        if (exp != (h[(la * 64 + lb) * 64 + ld] != 0.0)) ++bad;
      }
  printf("hypothesis lane = x + 4 b + 16 k, D lane = j + 4 b + 16 i: %d mismatches\n", bad);
  const int iters = 20000;
  for (int rep = 0; rep < 2; ++rep) {
    hipEvent_t e0, e1, e2;
    hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    hipEventRecord(e0);
    k_rate<0><<<256 * 4, 256>>>(d, iters, c);
    hipEventRecord(e1);
    k_rate<1><<<256 * 4, 256>>>(d, iters, c);
    hipEventRecord(e2);
    hipDeviceSynchronize();
    float m0, m1;
    hipEventElapsedTime(&m0, e0, e1);
    hipEventElapsedTime(&m1, e1, e2);
    long long hc[2];
    hipMemcpy(hc, c, sizeof(hc), hipMemcpyDeviceToHost);
    // 4 waves per SIMD, each 4 * iters MFMAs -> per SIMD 16 iters MFMAs
    printf("16x16x4: %.3f ms  %.1f ns per MFMA and SIMD  (clock64 per MFMA of one wave %.1f)\n", m0, m0 * 1e6 / (16.0 * iters), (double)hc[0] / (4.0 * iters));
    printf("4x4x4  : %.3f ms  %.1f ns per MFMA and SIMD  (clock64 per MFMA of one wave %.1f)\n", m1, m1 * 1e6 / (16.0 * iters), (double)hc[1] / (4.0 * iters));
  }
  return 0;
}
