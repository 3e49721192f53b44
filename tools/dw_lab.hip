// Micro-lab (round 5): what bounds a weight-gradient tile whose reduction runs over many chunks of 256 batch rows (virtual
// ranks: M = V x 256).  The product's hidden tile (csrc/mlp_dw.h dw_hot_tile) -- C[16, 64] += X[M, 16-column strip]^T .
// dY[M, 64-column panel] on v_mfma_f32_16x16x4, 4 waves splitting every 256-row chunk, software-pipelined in two halves --
// is rebuilt here with its ingredients switchable:
//   mode 0  as in the product                     mode 1  no loads of X (a constant instead)
//   mode 2  no loads of dY                        mode 3  no loads at all (matrix instructions only)
//   mode 4  loads only (one matrix instruction per chunk keeps them alive)
// and two alternative tile shapes with fewer operand bytes per flop:
//   mode 5  32 x 64 tile: two X strips per dY panel (8 accumulators)
//   mode 6  16 x 128 tile: two dY panels per X strip
// grid = tiles of 4 matrices [256, 256] (256 tiles of 16 x 64; 128 of the larger shapes), XCD-aware as in the product when
// xcd = 1 (blockIdx & 7 = XCD: matrix = XCD / 2, half = XCD % 2).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/dw_lab.hip -o tools/dw_lab && tools/dw_lab [M]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

struct Args { const float* X[4]; const float* Y[4]; float* C[4]; int M; int mode; int xcd; int S; };   // S: workgroups per tile, a segment of the rows each (no combine: partial tiles, one over the other)

__device__ __forceinline__ float ld_su(const float* u, uint32_t o) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(u) + o);
}
__device__ __forceinline__ f32x4 ld4_su(const float* u, uint32_t o) {
  return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(u) + o);
}

// TK x TN tile: TK = 16 | 32 rows of C (columns of X), TN = 64 | 128 columns
template <int TK, int TN, int MODE>
__device__ __forceinline__ void tile(const Args& A, const int pi, const int t, float* red) {
  constexpr int NA = TK / 16, NB = TN / 64;
  const int nx = 256 / TN;
  const int by = t / nx, bx = t % nx;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, q = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int k0 = by * TK, n0 = bx * TN;
  const float* xu = A.X[pi] + k0;
  const float* yu = A.Y[pi] + n0;
  const uint32_t xo = (uint32_t)(4 * q * 256 + j) * 4u;
  const uint32_t yo = (uint32_t)(4 * q * 256 + 4 * j) * 4u;
  f32x4 acc[NA][NB][4];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][n][e] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto load_half = [&](const int mb, const int h, float (&a)[NA][2][4], f32x4 (&b)[NB][2][4]) {
#pragma unroll
    for (int u2 = 0; u2 < 2; ++u2) {
      const int mu = mb + (wv + 4 * (2 * h + u2)) * 16;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
          a[i][u2][s] = (MODE == 1 || MODE == 3) ? 1.0f : ld_su(xu + (int64_t)(mu + s) * 256 + 16 * i, xo);
#pragma unroll
        for (int n = 0; n < NB; ++n)
          b[n][u2][s] = (MODE == 2 || MODE == 3) ? f32x4{1.f, 1.f, 1.f, 1.f} : ld4_su(yu + (int64_t)(mu + s) * 256 + 64 * n, yo);
      }
    }
  };
  auto mac_half = [&](const float (&a)[NA][2][4], const f32x4 (&b)[NB][2][4]) {
#pragma unroll
    for (int u2 = 0; u2 < 2; ++u2)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (MODE == 4) {
          if (u2 == 0 && s == 0) acc[0][0][0] = MFMA(a[0][u2][s], b[0][u2][s][0], acc[0][0][0]);
          else {
#pragma unroll
            for (int i = 0; i < NA; ++i) asm volatile("" :: "v"(a[i][u2][s]));
#pragma unroll
            for (int n = 0; n < NB; ++n) asm volatile("" :: "v"(b[n][u2][s]));
          }
          continue;
        }
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
          for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][n][e] = MFMA(a[i][u2][s], b[n][u2][s][e], acc[i][n][e]);
      }
  };
  float a0[NA][2][4], a1[NA][2][4];
  f32x4 b0[NB][2][4], b1[NB][2][4];
  const int C = A.M >> 8, seg = blockIdx.y;
  const int m_hi = (((seg + 1) * C) / A.S) << 8;
  int mb = ((seg * C) / A.S) << 8;
  load_half(mb, 0, a0, b0);
  if (MODE == 7) {
    // the loads of the other half go out one pair at a time, each behind the matrix instructions of one pair of this half:
    // a wave that issues 16 loads in a row is held at every one of them until the texture path -- which the CU's 4 waves
    // share -- has taken it, and its matrix instructions wait behind them
    auto one_pair = [&](const int mbl, const int h, const int u2, const int s, float (&a)[NA][2][4], f32x4 (&b)[NB][2][4]) {
      const int mu = mbl + (wv + 4 * (2 * h + u2)) * 16;
#pragma unroll
      for (int i = 0; i < NA; ++i) a[i][u2][s] = ld_su(xu + (int64_t)(mu + s) * 256 + 16 * i, xo);
#pragma unroll
      for (int n = 0; n < NB; ++n) b[n][u2][s] = ld4_su(yu + (int64_t)(mu + s) * 256 + 64 * n, yo);
    };
    auto mac_pair = [&](const int u2, const int s, const float (&a)[NA][2][4], const f32x4 (&b)[NB][2][4]) {
#pragma unroll
      for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][n][e] = MFMA(a[i][u2][s], b[n][u2][s][e], acc[i][n][e]);
    };
    for (; mb < m_hi; mb += 256) {
      const int mbn = (mb + 256 < m_hi) ? mb + 256 : mb;
#pragma unroll
      for (int u2 = 0; u2 < 2; ++u2)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          one_pair(mb, 1, u2, s, a1, b1);
          __builtin_amdgcn_sched_barrier(0);
          mac_pair(u2, s, a0, b0);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
      for (int u2 = 0; u2 < 2; ++u2)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          one_pair(mbn, 0, u2, s, a0, b0);
          __builtin_amdgcn_sched_barrier(0);
          mac_pair(u2, s, a1, b1);
          __builtin_amdgcn_sched_barrier(0);
        }
    }
  } else
  for (; mb < m_hi; mb += 256) {
    load_half(mb, 1, a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    mac_half(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    load_half((mb + 256 < m_hi) ? mb + 256 : mb, 0, a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    mac_half(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
  }
  // partial tiles of the 4 waves -> LDS -> sum -> C (as the product; no optimiser here)
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        f32x4 v = {acc[i][n][0][r], acc[i][n][1][r], acc[i][n][2][r], acc[i][n][3][r]};
        *reinterpret_cast<f32x4*>(red + ((wave * 16 + 4 * q + r) * 64 + 4 * j)) = v;
      }
      __syncthreads();
      const int orow = tid >> 4, c4 = tid & 15;
      f32x4 s = *reinterpret_cast<const f32x4*>(red + ((0 * 16 + orow) * 64 + 4 * c4));
#pragma unroll
      for (int w = 1; w < 4; ++w) s += *reinterpret_cast<const f32x4*>(red + ((w * 16 + orow) * 64 + 4 * c4));
      *reinterpret_cast<f32x4*>(A.C[pi] + (int64_t)(k0 + 16 * i + orow) * 256 + n0 + 64 * n + 4 * c4) = s;
    }
}

template <int TK, int TN, int MODE>
__global__ __launch_bounds__(256) void k_tiles(Args A) {
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];
  constexpr int per = (256 / TK) * (256 / TN);              // tiles per matrix
  int pi, t;
  if (A.xcd) {
    const int x = blockIdx.x & 7, r = blockIdx.x >> 3;
    pi = x >> 1;
    t = (x & 1) * (per / 2) + r;
  } else {
    pi = blockIdx.x / per; t = blockIdx.x % per;
  }
  tile<TK, TN, MODE>(A, pi, t, red);
}

template <int TK, int TN, int MODE>
static float run(const Args& A) {
  constexpr int per = (256 / TK) * (256 / TN);
  const int grid = 4 * per, N = 20, iters = 10;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int k = 0; k < N; ++k) hipLaunchKernelGGL((k_tiles<TK, TN, MODE>), dim3(grid, A.S), dim3(256), 0, st, A);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
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
  return ms * 1000.f / (iters * N);
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 4864;
  Args A;
  A.M = M; A.mode = 0; A.xcd = 1; A.S = 1;
  std::vector<float> h((size_t)M * 256);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 20 & 255) / 256.f - 0.5f;
  for (int p = 0; p < 4; ++p) {
    float *x, *y, *c;
    CK(hipMalloc(&x, sizeof(float) * M * 256)); CK(hipMalloc(&y, sizeof(float) * M * 256)); CK(hipMalloc(&c, sizeof(float) * 65536));
    CK(hipMemcpy(x, h.data(), sizeof(float) * M * 256, hipMemcpyHostToDevice));
    CK(hipMemcpy(y, h.data(), sizeof(float) * M * 256, hipMemcpyHostToDevice));
    A.X[p] = x; A.Y[p] = y; A.C[p] = c;
  }
  const double flop = 4.0 * 2.0 * M * 65536;
  printf("M = %d: 4 matrices [256, 256], %.2f GFLOP per launch (%.1f us at the 157.3 TFLOP/s f32 MFMA peak)\n", M, flop * 1e-9,
         flop / 157.3e12 * 1e6);
  for (int S : {1, 2, 4}) {
    A.xcd = 1; A.S = S;
    printf("xcd map, %d workgroup(s) per tile (segments of the rows)\n", S);
#define RUN(TK, TN, MODE, what) { const float us = run<TK, TN, MODE>(A); printf("  %-44s %7.2f us  %5.1f TFLOP/s\n", what, us, flop / us * 1e-6); }
    RUN(16, 64, 0, "16 x 64 tile as in the product");
    RUN(16, 64, 7, "16 x 64, loads interleaved pair by pair");
    RUN(32, 64, 7, "32 x 64, loads interleaved pair by pair");
    RUN(32, 128, 7, "32 x 128, loads interleaved pair by pair");
    RUN(16, 64, 1, "16 x 64, no loads of X");
    RUN(16, 64, 2, "16 x 64, no loads of dY");
    RUN(16, 64, 3, "16 x 64, matrix instructions only");
    RUN(16, 64, 4, "16 x 64, loads only");
    RUN(32, 64, 0, "32 x 64 tile (two X strips per dY panel)");
    RUN(16, 128, 0, "16 x 128 tile (two dY panels per X strip)");
    RUN(32, 128, 0, "32 x 128 tile");
  }
  return 0;
}
