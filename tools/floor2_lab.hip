// Micro-lab (round 5): what a launch costs that no workgroup sees.  tools/floor_lab.hip priced an EMPTY dependent kernel in
// a replayed hipGraph at 1.52 us whatever its grid.  The product's two update launches carry 4-5 us (ddpg_rows_kernel) and
// ~3 us (dw_adam_her_kernel) that their own in-kernel cycle stamps do not account for (DESIGN 4.3 "Round 4", 7).  This lab
// rebuilds such launches from their ingredients and attributes the time:
//   footprint  light (no LDS, few registers)  |  heavy: 64 KB of dynamic LDS + 211 VGPRs per lane (= ddpg_rows_kernel:
//              two workgroups per CU) -- the ramp of dispatching 256 / 576 such workgroups
//   dirty      every workgroup writes its share of 0 / 3 / 6 MB before it exits -- the release at the end of the kernel
//              (dirty L2 lines written back before the next kernel may start)
//   fetch      every workgroup first reads its share of 0 / 2 / 9 MB its PREDECESSOR wrote (other XCDs' L2s are not
//              coherent: the lines come from the Infinity Cache / HBM) -- the invalidate + cold fetch at the start
// Per configuration a graph of N dependent launches is replayed; besides the time per launch (events around the replays)
// every workgroup stamps s_memrealtime (100 MHz, device-wide) at its start and at its end:
//   span = last end - first start of a launch (what the workgroups see),   gap = first start of launch k+1 - last end of
//   launch k (what nobody sees),   span + gap = time per launch.
//   hipcc --offload-arch=gfx950 -O3 tools/floor2_lab.hip -o tools/floor2_lab && tools/floor2_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct Work {
  float* wr; const float* rd;       // this launch writes `wr`, reads what its predecessor wrote to `rd`
  long long dirty_f4, fetch_f4;     // float4s in all
  unsigned long long* stamps;       // [grid][2] of this launch: start / end of every workgroup (plain stores: a shared
                                    // min / max word would serialise 2 x grid same-address atomics, ~5 us at grid 256)
  float* sink;
};

__device__ __forceinline__ void body(const Work& w) {
  unsigned long long t0 = 0;
  if (threadIdx.x == 0) t0 = __builtin_amdgcn_s_memrealtime();
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long long)gridDim.x * blockDim.x;
  float4 acc = {0.f, 0.f, 0.f, 0.f};
  for (long long i = tid; i < w.fetch_f4; i += nth) {
    const float4 v = reinterpret_cast<const float4*>(w.rd)[i];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  for (long long i = tid; i < w.dirty_f4; i += nth) {
    float4 v = {acc.x + (float)i, acc.y, acc.z, acc.w};
    reinterpret_cast<float4*>(w.wr)[i] = v;
  }
  if (acc.x == 12345.678f) w.sink[0] = acc.y;                // (keeps the reads alive)
  if (threadIdx.x == 0) {
    __builtin_amdgcn_s_waitcnt(0);
    w.stamps[2 * blockIdx.x] = t0;
    w.stamps[2 * blockIdx.x + 1] = (unsigned long long)__builtin_amdgcn_s_memrealtime();
  }
}

__global__ __launch_bounds__(256) void k_light(Work w) { body(w); }
// the footprint of ddpg_rows_kernel: 64 KB of dynamic LDS (two workgroups per CU), 211 VGPRs
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(211))) void k_heavy(Work w) {
  extern __shared__ float lds[];
  if (w.dirty_f4 < 0) lds[threadIdx.x] = 1.f;                // (never: the LDS allocation is what counts)
  body(w);
}

struct Result { float us, span, gap; };

static Result run(bool heavy, int grid, long long dirty_bytes, long long fetch_bytes, float* bufs[2],
                  unsigned long long* stamps_d, float* sink) {
  const int N = 40, iters = 50;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  if (heavy) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_heavy), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int k = 0; k < N; ++k) {
    Work w;
    w.wr = bufs[k & 1]; w.rd = bufs[(k + 1) & 1];
    w.dirty_f4 = dirty_bytes / 16; w.fetch_f4 = fetch_bytes / 16;
    w.stamps = stamps_d + (size_t)2 * 1024 * k; w.sink = sink;
    if (heavy) hipLaunchKernelGGL(k_heavy, dim3(grid), dim3(256), 64 * 1024, st, w);
    else hipLaunchKernelGGL(k_light, dim3(grid), dim3(256), 0, st, w);
  }
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
  // the stamps of the last replay: first start / last end per launch
  std::vector<unsigned long long> raw((size_t)2 * 1024 * N);
  CK(hipMemcpy(raw.data(), stamps_d, sizeof(unsigned long long) * raw.size(), hipMemcpyDeviceToHost));
  std::vector<unsigned long long> s(2 * N);
  for (int k = 0; k < N; ++k) {
    unsigned long long lo = ~0ull, hi = 0;
    for (int b = 0; b < grid; ++b) {
      lo = raw[(size_t)2 * 1024 * k + 2 * b] < lo ? raw[(size_t)2 * 1024 * k + 2 * b] : lo;
      hi = raw[(size_t)2 * 1024 * k + 2 * b + 1] > hi ? raw[(size_t)2 * 1024 * k + 2 * b + 1] : hi;
    }
    s[2 * k] = lo; s[2 * k + 1] = hi;
  }
  double span = 0, gap = 0;
  for (int k = 4; k < N - 1; ++k) {                          // (the head of the graph launch is left out)
    span += (double)(s[2 * k + 1] - s[2 * k]);
    gap += (double)((long long)s[2 * (k + 1)] - (long long)s[2 * k + 1]);
  }
  Result r;
  r.us = ms * 1000.f / (iters * N);
  r.span = (float)(span / (N - 5) * 0.01);                   // 100 MHz ticks -> us
  r.gap = (float)(gap / (N - 5) * 0.01);
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(st));
  return r;
}

int main() {
  float* bufs[2];
  for (int i = 0; i < 2; ++i) { CK(hipMalloc(&bufs[i], 16 << 20)); CK(hipMemset(bufs[i], 0, 16 << 20)); }
  unsigned long long* stamps;
  CK(hipMalloc(&stamps, sizeof(unsigned long long) * 2 * 1024 * 64));
  float* sink;
  CK(hipMalloc(&sink, 64));
  printf("graph of 40 dependent launches, replayed 50x; us per launch | span = last end - first start (s_memrealtime) | "
         "gap = next first start - last end\n");
  const long long MB = 1 << 20;
  for (int heavy = 0; heavy < 2; ++heavy)
    for (int grid : {64, 256, 576}) {
      for (long long dirty : {0ll, 3 * MB, 6 * MB})
        for (long long fetch : {0ll, 2 * MB, 9 * MB}) {
          if (dirty == 0 && fetch != 0) continue;            // (nothing was written: nothing cold to fetch)
          const Result r = run(heavy != 0, grid, dirty, fetch, bufs, stamps, sink);
          printf("%s grid %3d  dirty %lld MB  fetch %lld MB :  %6.2f us per launch   span %6.2f   gap %5.2f\n",
                 heavy ? "heavy (64 KB LDS, 211 VGPR)" : "light                      ", grid, dirty / MB, fetch / MB, r.us,
                 r.span, r.gap);
        }
    }
  return 0;
}
