// Micro-lab: does kernarg preloading (user SGPRs filled at wave launch) shorten a small dependent kernel?
// Same 256x256x256 layer kernel with FLAT scalar arguments, built with and without
//   -mllvm -amdgpu-kernarg-preload-count=8
// and timed as a hipGraph of 16 dependent launches (ping-pong buffers).  Not part of the product.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ inline f32x4 ldv(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ inline f32x4 zero4() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }

__global__ __launch_bounds__(256) void k_flat(const float* X, const float* W, const float* bias, float* Y) {
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 64;
  const float* xr = X + (size_t)(m0 + j) * 256;
  const int col = n0 + 4 * j;
  f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
  f32x4 a[4], b[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int kq = (wave + 4 * u) * 16 + 4 * q;
    a[u] = ldv(xr + kq);
#pragma unroll
    for (int s = 0; s < 4; ++s) b[u][s] = ldv(W + (size_t)(kq + s) * 256 + col);
  }
  const f32x4 bv = ldv(bias + n0 + 4 * (tid & 15));
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = MFMA(a[u][s], b[u][s][e], acc[e]);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    f32x4 v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
    *reinterpret_cast<f32x4*>(red + ((wave * 16 + 4 * q + r) * 64 + 4 * j)) = v;
  }
  __syncthreads();
  const int orow = tid >> 4, c4 = tid & 15;
  f32x4 s = ldv(red + (orow * 64 + 4 * c4));
#pragma unroll
  for (int w = 1; w < 4; ++w) s += ldv(red + ((w * 16 + orow) * 64 + 4 * c4));
  s += bv;
#pragma unroll
  for (int e = 0; e < 4; ++e) s[e] = fmaxf(s[e], 0.f) * 0.25f;
  *reinterpret_cast<f32x4*>(Y + (size_t)(m0 + orow) * 256 + n0 + 4 * c4) = s;
}

int main() {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  float *buf[2], *W, *b;
  CK(hipMalloc(&buf[0], 256 * 256 * 4)); CK(hipMalloc(&buf[1], 256 * 256 * 4));
  CK(hipMalloc(&W, 256 * 256 * 4)); CK(hipMalloc(&b, 256 * 4));
  std::vector<float> h(256 * 256, 0.01f);
  CK(hipMemcpy(buf[0], h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(b, h.data(), 256 * 4, hipMemcpyHostToDevice));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int l = 0; l < 16; ++l) hipLaunchKernelGGL(k_flat, dim3(4, 16, 1), dim3(256), 0, st, buf[l & 1], W, b, buf[(l + 1) & 1]);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t ea, eb;
  CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
  for (int i = 0; i < 50; ++i) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(ea, st));
  for (int i = 0; i < 500; ++i) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(eb, st));
  CK(hipEventSynchronize(eb));
  float ms;
  CK(hipEventElapsedTime(&ms, ea, eb));
  printf("%s: %.3f us per dependent layer launch\n", PRELOAD_TAG, 1e3 * ms / 500 / 16);
  return 0;
}
