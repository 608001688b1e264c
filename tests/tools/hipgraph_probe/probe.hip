// Which multi-stream capture patterns does this HIP runtime accept?  (svgpfa_api.hip: graph_elbo.)  One pattern per run:
// a runtime fault in one must not hide the others.   usage: probe <A|B|C|D|E>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t s_ = (x); if (s_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(s_)); return 2; } } while (0)
__global__ void k(double* p, int i) { p[i] += 1.0; }
int main(int argc, char** argv) {
  const char pat = argc > 1 ? argv[1][0] : 'A';
  hipStream_t o, s1, s2;
  CK(hipStreamCreateWithFlags(&o, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e[6];
  for (auto& x : e) CK(hipEventCreate(&x));
  double* p;
  CK(hipMalloc(&p, 64 * 8));
  CK(hipMemset(p, 0, 64 * 8));
  hipGraph_t g;
  hipGraphExec_t x;
  double h[8];
  CK(hipStreamBeginCapture(o, hipStreamCaptureModeRelaxed));
  if (pat == 'F' || pat == 'G') {                      // the evaluation's own sequence (F), fork before any node (G)
    CK(hipEventRecord(e[0], o));
    CK(hipStreamWaitEvent(s1, e[0], 0));
    hipLaunchKernelGGL(k, 1, 1, 0, s1, p, 1);
    CK(hipEventRecord(e[1], s1));                      // class table ready
    hipLaunchKernelGGL(k, 1, 1, 0, o, p, 0);
    if (pat == 'F') {
      CK(hipEventRecord(e[2], o));
      hipLaunchKernelGGL(k, 1, 1, 0, o, p, 2);
      CK(hipStreamWaitEvent(s1, e[2], 0));
      hipLaunchKernelGGL(k, 1, 1, 0, s1, p, 3);
      CK(hipEventRecord(e[3], s1));
      CK(hipStreamWaitEvent(s2, e[3], 0));
      hipLaunchKernelGGL(k, 1, 1, 0, s2, p, 4);
      hipLaunchKernelGGL(k, 1, 1, 0, s1, p, 5);
      CK(hipEventRecord(e[4], s2));
      CK(hipStreamWaitEvent(s1, e[4], 0));
      hipLaunchKernelGGL(k, 1, 1, 0, s1, p, 6);
    }
    CK(hipEventRecord(e[5], s1));
    CK(hipStreamWaitEvent(o, e[5], 0));
    CK(hipStreamWaitEvent(o, e[1], 0));                // an older event of the side stream
    goto tail;
  }
  hipLaunchKernelGGL(k, 1, 1, 0, o, p, 0);
  CK(hipEventRecord(e[0], o));
  CK(hipStreamWaitEvent(s1, e[0], 0));               // fork
  hipLaunchKernelGGL(k, 1, 1, 0, s1, p, 1);
  if (pat == 'B') {                                    // side waits for a second origin event
    hipLaunchKernelGGL(k, 1, 1, 0, o, p, 2);
    CK(hipEventRecord(e[1], o));
    CK(hipStreamWaitEvent(s1, e[1], 0));
    hipLaunchKernelGGL(k, 1, 1, 0, s1, p, 3);
  }
  if (pat == 'C') {                                    // origin waits for two side events
    CK(hipEventRecord(e[1], s1));
    CK(hipStreamWaitEvent(o, e[1], 0));
    hipLaunchKernelGGL(k, 1, 1, 0, o, p, 2);
    hipLaunchKernelGGL(k, 1, 1, 0, s1, p, 3);
  }
  if (pat == 'D' || pat == 'E') {                      // nested fork: s2 from s1
    CK(hipEventRecord(e[2], s1));
    CK(hipStreamWaitEvent(s2, e[2], 0));
    hipLaunchKernelGGL(k, 1, 1, 0, s2, p, 4);
    hipLaunchKernelGGL(k, 1, 1, 0, s1, p, 5);
    CK(hipEventRecord(e[3], s2));
    if (pat == 'D') CK(hipStreamWaitEvent(s1, e[3], 0));   // joined to its parent
    else CK(hipStreamWaitEvent(o, e[3], 0));               // joined to the origin
    hipLaunchKernelGGL(k, 1, 1, 0, s1, p, 6);
  }
  CK(hipEventRecord(e[4], s1));
  CK(hipStreamWaitEvent(o, e[4], 0));                // join
tail:
  hipLaunchKernelGGL(k, 1, 1, 0, o, p, 7);
  printf("pattern %c: ending capture\n", pat); fflush(stdout);
  CK(hipStreamEndCapture(o, &g));
  CK(hipGraphInstantiate(&x, g, nullptr, nullptr, 0));
  for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(x, o));
  CK(hipStreamSynchronize(o));
  CK(hipMemcpy(h, p, sizeof h, hipMemcpyDeviceToHost));
  printf("pattern %c ok:", pat);
  for (double v : h) printf(" %g", v);
  printf("\n");
  return 0;
}
