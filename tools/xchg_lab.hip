// Micro-lab: what it costs two RESIDENT workgroups to hand each other a few KB through the L2 in the middle of a kernel
// (the exchange a layer split over a pair of workgroups would need), for partners on the same XCD (block ids b, b + 8)
// and on different XCDs (b, b + 1).  No fences that write back or invalidate the L2: payload and flag travel as relaxed
// agent-scope atomics (sc1: served by the L2 / coherent across XCDs), ordered by s_waitcnt + a workgroup barrier.
//   hipcc --offload-arch=gfx950 -O3 tools/xchg_lab.hip -o /tmp/xchg_lab && /tmp/xchg_lab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_flag(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t ld_flag(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Every workgroup: `iters` rounds of  publish `words` floats + flag -> wait for the partner's flag -> read its floats.
// mode 0: symmetric exchange (both publish, both read);  dist = block-id distance of the partners (8: same XCD, 1: not)
// work = dummy dependent FMA steps between the rounds (0 = pure latency)
__global__ __launch_bounds__(256) void xchg_kernel(float* buf, uint32_t* flags, int words, int iters, int dist, int work,
                                                   unsigned long long* cycles, uint32_t* xcc, float* sink) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const int pair_lo = (b / (2 * dist)) * (2 * dist) + (b % dist);          // the lower block id of my pair
  const int me = (b >= pair_lo + dist) ? 1 : 0;
  const int pair = (b / (2 * dist)) * dist + (b % dist);
  float* mine = buf + ((size_t)pair * 2 + me) * 2 * 2048;                  // double-buffered by round parity
  const float* theirs = buf + ((size_t)pair * 2 + (me ^ 1)) * 2 * 2048;
  uint32_t* myflag = flags + (pair * 2 + me) * 32;
  const uint32_t* theirflag = flags + (pair * 2 + (me ^ 1)) * 32;
  if (tid == 0) {
    uint32_t id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    xcc[b] = id;
  }
  float acc = (float)tid;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 1; it <= iters; ++it) {
    float* out = mine + (it & 1) * 2048;
    const float* in = theirs + (it & 1) * 2048;
    for (int i = tid; i < words; i += 256) st_agent(out + i, acc + (float)i);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");                 // s_waitcnt vmcnt(0): my stores reached the L2
    __syncthreads();
    if (tid == 0) st_flag(myflag, (uint32_t)it);
    for (int w = 0; w < work; ++w) acc = acc * 1.0000001f + 0.5f;          // independent work that could hide the hop
    if (tid == 0) {
      while (ld_flag(theirflag) < (uint32_t)it) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    float s = 0.f;
    for (int i = tid; i < words; i += 256) s += ld_agent(in + i);
    acc += s * 1e-9f;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (tid == 0) cycles[b] = t1 - t0;
  sink[b * 256 + tid] = acc;
}

int main() {
  const int NB = 256;
  float* buf; uint32_t* flags; unsigned long long* cyc; uint32_t* xcc; float* sink;
  CK(hipMalloc(&buf, (size_t)NB * 2 * 2048 * 4));
  CK(hipMalloc(&flags, NB * 32 * 4));
  CK(hipMalloc(&cyc, NB * 8));
  CK(hipMalloc(&xcc, NB * 4));
  CK(hipMalloc(&sink, NB * 256 * 4));
  unsigned long long hc[NB]; uint32_t hx[NB];
  const int iters = 200;
  printf("symmetric exchange between two resident workgroups, %d rounds; cycles of the 100 MHz-independent shader clock "
         "(s_memtime) per round, avg / max over workgroups\n", iters);
  for (int nb : {16, 128, 256}) {
    for (int dist : {8, 1}) {
      for (int words : {16, 512, 1024}) {
        for (int work : {0, 1500}) {
          CK(hipMemset(flags, 0, NB * 32 * 4));
          hipEvent_t e0, e1;
          CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
          CK(hipEventRecord(e0, 0));
          hipLaunchKernelGGL(xchg_kernel, dim3(nb), dim3(256), 0, 0, buf, flags, words, iters, dist, work, cyc, xcc, sink);
          CK(hipEventRecord(e1, 0));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          CK(hipMemcpy(hc, cyc, nb * 8, hipMemcpyDeviceToHost));
          CK(hipMemcpy(hx, xcc, nb * 4, hipMemcpyDeviceToHost));
          double avg = 0, mx = 0;
          int same = 0;
          for (int b = 0; b < nb; ++b) {
            avg += (double)hc[b] / iters; if ((double)hc[b] / iters > mx) mx = (double)hc[b] / iters;
            const int lo = (b / (2 * dist)) * (2 * dist) + (b % dist);
            if (b == lo && hx[b] == hx[b + dist]) ++same;
          }
          printf("grid %3d, partner +%d (%3d of %3d pairs share an XCD), %4d floats, %4d fma of work: %.0f / %.0f cycles "
                 "per round, %.3f us per round by the event clock\n", nb, dist, same, nb / 2, words, work, avg / nb, mx,
                 ms * 1000.f / iters);
        }
      }
    }
  }
  printf("xcc ids of blocks 0..15:");
  for (int b = 0; b < 16; ++b) printf(" %u", hx[b]);
  printf("\n");
  return 0;
}
