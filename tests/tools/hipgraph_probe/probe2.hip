// The evaluation's launch sequence (svgpfa_api.hip: enqueue_elbo), first with plain launches into the null stream and then
// captured on a private stream with the SAME events -- as graph_elbo does.   usage: probe2 <plain runs before the capture>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t s_ = (x); if (s_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(s_)); return 2; } } while (0)
__global__ void k(double* p, int i) { p[i] += 1.0; }
static hipEvent_t ev_build, ev_cls, ev_prep, ev_fork, ev_join, ev_spk;
static int seq(hipStream_t o, hipStream_t s1, hipStream_t s2, double* p, bool need_cls) {
  if (need_cls) {
    CK(hipEventRecord(ev_build, o));
    CK(hipStreamWaitEvent(s1, ev_build, 0));
    hipLaunchKernelGGL(k, 1, 1, 0, s1, p, 1);
    CK(hipEventRecord(ev_cls, s1));
  }
  hipLaunchKernelGGL(k, 1, 1, 0, o, p, 0);
  CK(hipEventRecord(ev_prep, o));
  hipLaunchKernelGGL(k, 1, 1, 0, o, p, 2);
  CK(hipStreamWaitEvent(s1, ev_prep, 0));
  hipLaunchKernelGGL(k, 1, 1, 0, s1, p, 3);
  CK(hipEventRecord(ev_fork, s1));
  CK(hipStreamWaitEvent(s2, ev_fork, 0));
  hipLaunchKernelGGL(k, 1, 1, 0, s2, p, 4);
  hipLaunchKernelGGL(k, 1, 1, 0, s1, p, 5);
  CK(hipEventRecord(ev_join, s2));
  CK(hipStreamWaitEvent(s1, ev_join, 0));
  hipLaunchKernelGGL(k, 1, 1, 0, s1, p, 6);
  CK(hipEventRecord(ev_spk, s1));
  CK(hipStreamWaitEvent(o, ev_spk, 0));
  if (need_cls) CK(hipStreamWaitEvent(o, ev_cls, 0));
  hipLaunchKernelGGL(k, 1, 1, 0, o, p, 7);
  CK(hipGetLastError());
  return 0;
}
int main(int argc, char** argv) {
  const int nplain = argc > 1 ? atoi(argv[1]) : 2;
  const bool cls_in_capture = argc > 2 && atoi(argv[2]) != 0;
  hipStream_t o, s1, s2;
  CK(hipStreamCreateWithFlags(&o, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  for (hipEvent_t* x : {&ev_build, &ev_cls, &ev_prep, &ev_fork, &ev_join, &ev_spk}) CK(hipEventCreateWithFlags(x, getenv("PROBE_TIMING") ? 0 : hipEventDisableTiming));
  {
    char line[512];
    FILE* f = fopen("/proc/self/maps", "r");
    while (f && fgets(line, sizeof line, f))
      if (strstr(line, "libamdhip64") && strstr(line, "r-xp")) printf("%s", strchr(line, '/'));
    if (f) fclose(f);
  }
  double* p;
  CK(hipMalloc(&p, 64 * 8));
  CK(hipMemset(p, 0, 64 * 8));
  for (int i = 0; i < nplain; ++i) {
    if (seq(nullptr, s1, s2, p, i == 0)) return 2;
    CK(hipStreamSynchronize(nullptr));
  }
  CK(hipStreamBeginCapture(o, hipStreamCaptureModeRelaxed));
  if (seq(o, s1, s2, p, cls_in_capture)) return 2;
  hipGraph_t g;
  printf("plain %d: ending capture\n", nplain); fflush(stdout);
  CK(hipStreamEndCapture(o, &g));
  hipGraphExec_t x;
  CK(hipGraphInstantiate(&x, g, nullptr, nullptr, 0));
  for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(x, nullptr));
  CK(hipStreamSynchronize(nullptr));
  double h[8];
  CK(hipMemcpy(h, p, sizeof h, hipMemcpyDeviceToHost));
  printf("ok:");
  for (double v : h) printf(" %g", v);
  printf("\n");
  return 0;
}
