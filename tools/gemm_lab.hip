// Micro-lab for the 256x256x256 f32 layer GEMM: times back-to-back launches of kernel variants (hipcc
// --offload-arch=gfx950 tools/gemm_lab.hip -o /tmp/gemm_lab && /tmp/gemm_lab).  Not part of the product.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <math.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ inline f32x4 ldv(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ inline f32x4 zero4() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }

struct P { const float* X; const float* W; const float* b; float* Y; int M, N, K; };
struct Args { P p[3]; };

__global__ void k_empty(Args a) {}

// V1: the product kernel's structure: WG = 16x64 tile, 4 waves split K, all loads up front, LDS reduce.
__global__ __launch_bounds__(256) void k_v1(Args args) {
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];
  const P& p = args.p[blockIdx.z];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 64;
  const float* xr = p.X + (size_t)(m0 + j) * p.K;
  const int col = n0 + 4 * j;
  f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
  f32x4 a[4], b[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int kq = (wave + 4 * u) * 16 + 4 * q;
    a[u] = ldv(xr + kq);
#pragma unroll
    for (int s = 0; s < 4; ++s) b[u][s] = ldv(p.W + (size_t)(kq + s) * p.N + col);
  }
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
  s += ldv(p.b + n0 + 4 * c4);
#pragma unroll
  for (int e = 0; e < 4; ++e) s[e] = fmaxf(s[e], 0.f);
  *reinterpret_cast<f32x4*>(p.Y + (size_t)(m0 + orow) * p.N + n0 + 4 * c4) = s;
}

// V2: same math, but no split-K: one wave = 16x64 tile over the whole K in a rolled loop (small code),
// 4 waves per WG = 4 different column tiles (WG tile 16 x 256), grid (1, M/16).
__global__ __launch_bounds__(256) void k_v2(Args args) {
  const P& p = args.p[blockIdx.z];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.y * 16, n0 = wave * 64;
  const float* xr = p.X + (size_t)(m0 + j) * p.K;
  const int col = n0 + 4 * j;
  f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
  for (int c = 0; c < p.K / 16; ++c) {
    const int kq = c * 16 + 4 * q;
    f32x4 a = ldv(xr + kq);
    f32x4 b[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) b[s] = ldv(p.W + (size_t)(kq + s) * p.N + col);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = MFMA(a[s], b[s][e], acc[e]);
  }
  f32x4 bias = ldv(p.b + col);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    f32x4 v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
    v += bias;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    *reinterpret_cast<f32x4*>(p.Y + (size_t)(m0 + 4 * q + r) * p.N + col) = v;
  }
}

// V3: split-K over 8 waves (512 threads), 2 chunks each.
__global__ __launch_bounds__(512) void k_v3(Args args) {
  __shared__ __attribute__((aligned(16))) float red[8 * 16 * 64];
  const P& p = args.p[blockIdx.z];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 64;
  const float* xr = p.X + (size_t)(m0 + j) * p.K;
  const int col = n0 + 4 * j;
  f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
  f32x4 a[2], b[2][4];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int kq = (wave + 8 * u) * 16 + 4 * q;
    a[u] = ldv(xr + kq);
#pragma unroll
    for (int s = 0; s < 4; ++s) b[u][s] = ldv(p.W + (size_t)(kq + s) * p.N + col);
  }
#pragma unroll
  for (int u = 0; u < 2; ++u)
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
  if (tid < 256) {
    const int orow = tid >> 4, c4 = tid & 15;
    f32x4 s = ldv(red + (orow * 64 + 4 * c4));
#pragma unroll
    for (int w = 1; w < 8; ++w) s += ldv(red + ((w * 16 + orow) * 64 + 4 * c4));
    s += ldv(p.b + n0 + 4 * c4);
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] = fmaxf(s[e], 0.f);
    *reinterpret_cast<f32x4*>(p.Y + (size_t)(m0 + orow) * p.N + n0 + 4 * c4) = s;
  }
}

// V4: plain VALU, one thread per output column, 4 rows per WG (row slab), W streamed row by row (coalesced).
__global__ __launch_bounds__(256) void k_v4(Args args) {
  __shared__ float xs[4][256];
  const P& p = args.p[blockIdx.z];
  const int tid = threadIdx.x, m0 = blockIdx.y * 4;
  for (int i = tid; i < 4 * p.K; i += 256) xs[i / p.K][i % p.K] = p.X[(size_t)(m0 + i / p.K) * p.K + i % p.K];
  __syncthreads();
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int k = 0; k < p.K; ++k) {
    float w = p.W[(size_t)k * p.N + tid];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] += xs[r][k] * w;
  }
  float b = p.b[tid];
#pragma unroll
  for (int r = 0; r < 4; ++r) p.Y[(size_t)(m0 + r) * p.N + tid] = fmaxf(acc[r] + b, 0.f);
}

template <typename F>
static double bench(const char* name, F launch, int iters, hipStream_t st) {
  for (int i = 0; i < 20; ++i) launch();
  CK(hipStreamSynchronize(st));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipEventRecord(a, st));
  for (int i = 0; i < iters; ++i) launch();
  CK(hipEventRecord(b, st));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  double us = 1e3 * ms / iters;
  printf("%-44s %8.2f us / launch\n", name, us);
  return us;
}

int main() {
  const int M = 256, N = 256, K = 256;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  std::vector<float> hX(3 * M * K), hW(3 * K * N), hb(3 * N), hY(M * N), ref(M * N);
  for (auto& v : hX) v = (rand() % 2001 - 1000) / 1000.f;
  for (auto& v : hW) v = (rand() % 2001 - 1000) / 4000.f;
  for (auto& v : hb) v = (rand() % 2001 - 1000) / 1000.f;
  float *dX, *dW, *db, *dY;
  CK(hipMalloc(&dX, hX.size() * 4)); CK(hipMalloc(&dW, hW.size() * 4)); CK(hipMalloc(&db, hb.size() * 4));
  CK(hipMalloc(&dY, 3 * M * N * 4));
  CK(hipMemcpy(dX, hX.data(), hX.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  Args a;
  for (int i = 0; i < 3; ++i) a.p[i] = P{dX + i * M * K, dW + i * K * N, db + i * N, dY + i * M * N, M, N, K};
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      double s = hb[n];
      for (int k = 0; k < K; ++k) s += (double)hX[m * K + k] * hW[k * N + n];
      ref[m * N + n] = s > 0 ? s : 0;
    }
  auto check = [&](const char* name) {
    CK(hipMemcpy(hY.data(), dY, M * N * 4, hipMemcpyDeviceToHost));
    double err = 0;
    for (int i = 0; i < M * N; ++i) err = fmax(err, fabs(hY[i] - ref[i]));
    printf("   %s max err %.2e\n", name, err);
    CK(hipMemset(dY, 0, 3 * M * N * 4));
  };
  const int it = 2000;
  for (int nz = 1; nz <= 3; nz += 2) {
    printf("---- %d problem(s) per launch\n", nz);
    bench("empty <<<(4,16,nz),256>>>", [&] { hipLaunchKernelGGL(k_empty, dim3(4, 16, nz), dim3(256), 0, st, a); }, it, st);
    bench("v1 splitK4 16x64 tile, unrolled", [&] { hipLaunchKernelGGL(k_v1, dim3(4, 16, nz), dim3(256), 0, st, a); }, it, st);
    check("v1");
    bench("v2 wave=16x64 over full K, rolled", [&] { hipLaunchKernelGGL(k_v2, dim3(1, 16, nz), dim3(256), 0, st, a); }, it, st);
    check("v2");
    bench("v3 splitK8 (512 thr)", [&] { hipLaunchKernelGGL(k_v3, dim3(4, 16, nz), dim3(512), 0, st, a); }, it, st);
    check("v3");
    bench("v4 VALU row-slab 4 rows/WG", [&] { hipLaunchKernelGGL(k_v4, dim3(1, 64, nz), dim3(256), 0, st, a); }, it, st);
    check("v4");
  }
  // graph of 16 dependent launches (what one update looks like)
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int i = 0; i < 16; ++i) hipLaunchKernelGGL(k_v1, dim3(4, 16, 3), dim3(256), 0, st, a);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  bench("graph of 16 x v1 (per graph)", [&] { CK(hipGraphLaunch(ge, st)); }, 300, st);
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int i = 0; i < 16; ++i) hipLaunchKernelGGL(k_empty, dim3(4, 16, 3), dim3(256), 0, st, a);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  bench("graph of 16 x empty (per graph)", [&] { CK(hipGraphLaunch(ge, st)); }, 300, st);
  return 0;
}
