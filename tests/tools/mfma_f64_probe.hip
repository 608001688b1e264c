// Probe v_mfma_f64_16x16x4_f64 lane layouts on gfx950 with exact integer data.
// Claims checked:  A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D reg r: row=(l>>4)+4r, col=l&15,
// and "accumulator register r of a 16x16 tile X is the B operand of K-chunk r" (Y = A' X).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void probe(const double* A, const double* B, const double* A2, double* D, double* Y) {
  int l = threadIdx.x;
  // D = A(16x16) * B(16x16) via 4 K-chunks
  d4 acc = {0, 0, 0, 0};
  for (int c = 0; c < 4; ++c) {
    double a = A[(l & 15) * 16 + 4 * c + (l >> 4)];
    double b = B[(4 * c + (l >> 4)) * 16 + (l & 15)];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
  // Y = A2 * D using acc registers directly as B operands
  d4 y = {0, 0, 0, 0};
  for (int c = 0; c < 4; ++c) {
    double a = A2[(l & 15) * 16 + 4 * c + (l >> 4)];
    y = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc[c], y, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) Y[((l >> 4) + 4 * r) * 16 + (l & 15)] = y[r];
}

int main() {
  double hA[256], hB[256], hA2[256], hD[256], hY[256], rD[256], rY[256];
  for (int i = 0; i < 256; ++i) { hA[i] = (i * 7 + 3) % 11 - 5; hB[i] = (i * 5 + 1) % 13 - 6; hA2[i] = (i * 3 + 2) % 7 - 3; }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 16; ++k) s += hA[i*16+k]*hB[k*16+j]; rD[i*16+j] = s; }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 16; ++k) s += hA2[i*16+k]*rD[k*16+j]; rY[i*16+j] = s; }
  double *dA, *dB, *dA2, *dD, *dY;
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dA2, 2048); hipMalloc(&dD, 2048); hipMalloc(&dY, 2048);
  hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice); hipMemcpy(dA2, hA2, 2048, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dA2, dD, dY);
  hipMemcpy(hD, dD, 2048, hipMemcpyDeviceToHost); hipMemcpy(hY, dY, 2048, hipMemcpyDeviceToHost);
  int badD = 0, badY = 0;
  for (int i = 0; i < 256; ++i) { badD += hD[i] != rD[i]; badY += hY[i] != rY[i]; }
  printf("mfma_f64 probe: D mismatches %d, chained Y mismatches %d\n", badD, badY);
  return badD + badY ? 1 : 0;
}
