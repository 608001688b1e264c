/*
 * Torch-free use of the drop-in boundary: plain C + the HIP runtime for device memory, nothing else.
 *
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ examples/c_abi_demo.c -I/opt/rocm/include -Iinclude -Lsvgpfa_amd/csrc -lsvgpfa_hip \
 *       -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/svgpfa_amd/csrc -Wl,-rpath,/opt/rocm/lib -o c_abi_demo
 *   ./c_abi_demo problem.bin
 *
 * problem.bin (written by tests/test_gpu_c_abi.py): int32 header R N K Q S, then per latent int32 M_k, kernel type
 * (0 RBF, 1 periodic); then doubles: reg, quad points R*Q, quad weights R*Q, int64 offsets R+1, spike times S,
 * int32 spike neuron S, then per latent Z (R*M), mean (R*M), cholVecs (R*P), theta (1 or 2), then C (N*K), d (N).
 * Prints "ELBO ELL KL" and the sums of every gradient group (all four requested), one number per line.
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "svgpfa.h"

#define CHECK(call)                                                                 \
  do {                                                                              \
    int rc_ = (call);                                                               \
    if (rc_ != 0) {                                                                 \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, eng ? svgpfa_last_error(eng) : ""); \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

static void* rd(FILE* f, size_t bytes) {
  void* p = malloc(bytes ? bytes : 1);
  if (fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "short read\n"); exit(2); }
  return p;
}
static void* to_dev(const void* h, size_t bytes) {
  void* d = NULL;
  if (hipMalloc(&d, bytes ? bytes : 8) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); exit(3); }
  if (bytes && hipMemcpy(d, h, bytes, hipMemcpyHostToDevice) != hipSuccess) { fprintf(stderr, "H2D failed\n"); exit(3); }
  return d;
}
static double dev_sum(const double* d, size_t n) {
  double* h = (double*)malloc(n * sizeof(double) + 8);
  hipMemcpy(h, d, n * sizeof(double), hipMemcpyDeviceToHost);
  double s = 0.0;
  for (size_t i = 0; i < n; ++i) s += h[i];
  free(h);
  return s;
}

int main(int argc, char** argv) {
  svgpfa_engine* eng = NULL;
  if (argc < 2) { fprintf(stderr, "usage: %s problem.bin\n", argv[0]); return 2; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 2; }
  int32_t* hdr = (int32_t*)rd(f, 5 * sizeof(int32_t));
  const int R = hdr[0], N = hdr[1], K = hdr[2], Q = hdr[3], S = hdr[4];
  int32_t* mk = (int32_t*)rd(f, 2 * K * sizeof(int32_t));
  int32_t M[32], kt[32];
  for (int k = 0; k < K; ++k) { M[k] = mk[2 * k]; kt[k] = mk[2 * k + 1]; }
  double* reg = (double*)rd(f, sizeof(double));
  double* tq = (double*)rd(f, (size_t)R * Q * 8);
  double* wq = (double*)rd(f, (size_t)R * Q * 8);
  int64_t* off = (int64_t*)rd(f, (size_t)(R + 1) * 8);
  double* st = (double*)rd(f, (size_t)S * 8);
  int32_t* sn = (int32_t*)rd(f, (size_t)S * 4);

  svgpfa_dims dims = {R, N, K, Q, M, kt, NULL, NULL, NULL};
  CHECK(svgpfa_create(&dims, 0, &eng));
  CHECK(svgpfa_set_reg(eng, *reg));
  CHECK(svgpfa_set_quad(eng, (double*)to_dev(tq, (size_t)R * Q * 8), (double*)to_dev(wq, (size_t)R * Q * 8)));
  CHECK(svgpfa_set_spikes(eng, off, (double*)to_dev(st, (size_t)S * 8), (int32_t*)to_dev(sn, (size_t)S * 4)));

  const double *Z[32], *mean[32], *chol[32], *theta[32];
  double *g_mean[32], *g_chol[32], *g_Z[32];
  size_t nP[32];
  for (int k = 0; k < K; ++k) {
    const size_t rm = (size_t)R * M[k], P = (size_t)M[k] * (M[k] + 1) / 2, nth = kt[k] == 0 ? 1 : 2;
    nP[k] = (size_t)R * P;
    Z[k] = (double*)to_dev(rd(f, rm * 8), rm * 8);
    mean[k] = (double*)to_dev(rd(f, rm * 8), rm * 8);
    chol[k] = (double*)to_dev(rd(f, nP[k] * 8), nP[k] * 8);
    theta[k] = (double*)to_dev(rd(f, nth * 8), nth * 8);
    g_mean[k] = (double*)to_dev(NULL, 0);
    hipMalloc((void**)&g_mean[k], rm * 8);
    hipMalloc((void**)&g_chol[k], nP[k] * 8);
    hipMalloc((void**)&g_Z[k], rm * 8);
  }
  double* C = (double*)to_dev(rd(f, (size_t)N * K * 8), (size_t)N * K * 8);
  double* d = (double*)to_dev(rd(f, (size_t)N * 8), (size_t)N * 8);
  fclose(f);

  CHECK(svgpfa_build_kernels(eng, Z, theta));
  const int64_t ns = svgpfa_shared_len(eng);
  double* shared = NULL;
  hipMalloc((void**)&shared, (size_t)ns * 8);
  const unsigned mask = SVGPFA_GRAD_VARIATIONAL | SVGPFA_GRAD_EMBEDDING | SVGPFA_GRAD_KERNELS | SVGPFA_GRAD_INDPOINTS;
  CHECK(svgpfa_elbo(eng, mean, chol, C, d, mask, shared, g_mean, g_chol, g_Z));
  CHECK(svgpfa_sync(eng));

  double* hs = (double*)malloc((size_t)ns * 8);
  hipMemcpy(hs, shared, (size_t)ns * 8, hipMemcpyDeviceToHost);
  printf("%.17g\n%.17g\n%.17g\n", hs[0], hs[1], hs[2]);
  double sC = 0.0, sd = 0.0, sth = 0.0;
  for (int i = 0; i < N * K; ++i) sC += hs[3 + i];
  for (int i = 0; i < N; ++i) sd += hs[3 + N * K + i];
  for (int64_t i = 3 + (int64_t)N * K + N; i < ns; ++i) sth += hs[i];
  printf("%.17g\n%.17g\n%.17g\n", sC, sd, sth);
  double sm = 0.0, sc = 0.0, sz = 0.0;
  for (int k = 0; k < K; ++k) {
    sm += dev_sum(g_mean[k], (size_t)R * M[k]);
    sc += dev_sum(g_chol[k], nP[k]);
    sz += dev_sum(g_Z[k], (size_t)R * M[k]);
  }
  printf("%.17g\n%.17g\n%.17g\n", sm, sc, sz);
  svgpfa_destroy(eng);
  return 0;
}
