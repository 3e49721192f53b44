// Micro-lab: what a dependent kernel boundary costs inside a replayed hipGraph as a function of the grid size of an
// (otherwise empty) kernel -- separates the boundary itself from the dispatch/retire time of the workgroups.
//   hipcc --offload-arch=gfx950 -O3 tools/floor_lab.hip -o /tmp/floor_lab && /tmp/floor_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct Big { float* p[100]; };                       // an 800-byte kernarg like the product's descriptors
__global__ void k_empty(float* p) {}
__global__ void k_empty_big(Big b) {}
__global__ void k_touch(float* p) {                  // one dependent load -> store per thread (L2 / Infinity Cache hop)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  p[i] = p[i] + 1.0f;
}

template <typename F>
static float graph_time(F launch, int nk, int iters) {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < nk; ++i) launch(st);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(st));
  return ms * 1000.f / iters;
}

int main() {
  float* buf;
  CK(hipMalloc(&buf, 4096 * 256 * 4));
  CK(hipMemset(buf, 0, 4096 * 256 * 4));
  Big big;
  for (int i = 0; i < 100; ++i) big.p[i] = buf;
  printf("graph of N dependent launches, replayed 200x; us per kernel = (t(N=64) - t(N=16)) / 48\n");
  for (int wg : {1, 8, 64, 192, 256, 512, 1024}) {
    for (int thr : {64, 256}) {
      auto le = [&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(wg), dim3(thr), 0, s, buf); };
      auto lb = [&](hipStream_t s) { hipLaunchKernelGGL(k_empty_big, dim3(wg), dim3(thr), 0, s, big); };
      auto lt = [&](hipStream_t s) { hipLaunchKernelGGL(k_touch, dim3(wg), dim3(thr), 0, s, buf); };
      const float e = (graph_time(le, 64, 200) - graph_time(le, 16, 200)) / 48.f;
      const float b = (graph_time(lb, 64, 200) - graph_time(lb, 16, 200)) / 48.f;
      const float t = (graph_time(lt, 64, 200) - graph_time(lt, 16, 200)) / 48.f;
      printf("grid %4d x %3d threads: empty %.2f us, empty + 800 B kernarg %.2f us, load->store %.2f us\n", wg, thr, e, b, t);
    }
  }
  return 0;
}
