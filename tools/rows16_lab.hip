// Micro-lab (round 5): a row-local layer chain for batches of MANY rows (virtual ranks) with the waves of a workgroup
// splitting the OUTPUT COLUMNS instead of k.  The product's row-local kernel (csrc/mlp_rows_layers.h) gives a workgroup 4 or
// 8 batch rows on v_mfma_f32_4x4x1: every wave takes a quarter of k for all 256 columns, the four partial tiles meet in LDS,
// and a CU's texture path has to deliver the whole 256 KB of a layer per 8 rows -- at 19 ranks texture time and matrix time
// are equal (8 k cycles per pair of workgroups and layer) and add up to 12.8 k.  Here a workgroup owns 16 rows:
//   v_mfma_f32_16x16x4: A = activations [16 rows x 4 k] from LDS, B = W[4 k x 16 columns]; lane (q, j) loads the 16 bytes
//   W[k + q][c0 + 4 j .. + 3] (a wave instruction = 4 rows of 256 bytes) and feeds 4 instructions (columns 4 j + e);
//   wave w owns columns [64 w, 64 w + 64) for the WHOLE of k: no partial tiles, no reduction through LDS; its finished
//   columns go straight into the next layer's activation rows (two LDS buffers, one barrier per layer).
// Bytes per flop halve against 8 rows (256 KB per 16 rows and layer), weight registers 32 instead of 128 per lane.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/rows16_lab.hip -o tools/rows16_lab && tools/rows16_lab [groups]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define HLD 260                                              // LDS row stride of an activation row (floats)

struct Args { const float* W; const float* bias; const float* x; float* y; int layers; };   // W: [layers][256][256]

// DEPTH: k-steps of 4 whose weight fragments are in flight (each: one 16-byte load per lane)
template <int DEPTH>
__global__ __launch_bounds__(256) void chain16(Args A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* hs0 = lds;
  float* hs1 = lds + 16 * HLD;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, q = lane >> 4;
  const int c0 = 64 * wave;
  // input rows
  for (int i = tid; i < 16 * 256; i += 256) hs0[(i >> 8) * HLD + (i & 255)] = A.x[(int64_t)blockIdx.x * 4096 + i];
  __syncthreads();
  float* cur = hs0;
  float* nxt = hs1;
  f32x4 wb[DEPTH];
  const float* wl = A.W + (int64_t)q * 256 + c0 + 4 * j;     // lane part of every weight address
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) wb[d] = *reinterpret_cast<const f32x4*>(wl + (int64_t)(4 * d) * 256);
  for (int l = 0; l < A.layers; ++l) {
    const float* W = A.W + (int64_t)l * 65536 + (int64_t)q * 256 + c0 + 4 * j;
    const float* Wn = (l + 1 < A.layers) ? W + 65536 : W;
    f32x4 acc[4] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    const f32x4 bv = *reinterpret_cast<const f32x4*>(A.bias + l * 256 + c0 + 4 * j);
    // 64 k-steps of 4; the fragment of step s + DEPTH is requested behind the instructions of step s
#pragma unroll 1
    for (int s0 = 0; s0 < 64; s0 += DEPTH) {
      // activations of DEPTH k-steps: lane (q, j) needs cur[j][4 s + q]
      float a[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) a[d] = cur[j * HLD + 4 * (s0 + d) + q];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const f32x4 b = wb[d];
        const int sn = s0 + d + DEPTH;
        wb[d] = *reinterpret_cast<const f32x4*>((sn < 64 ? W + (int64_t)(4 * sn) * 256 : Wn + (int64_t)(4 * (sn - 64)) * 256));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = MFMA16(a[d], b[e], acc[e]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // lane (q, j): rows 4 q + r, columns c0 + 4 j + e
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      f32x4 v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
      v += bv;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      *reinterpret_cast<f32x4*>(nxt + (4 * q + r) * HLD + c0 + 4 * j) = v;
    }
    __syncthreads();
    float* t = cur; cur = nxt; nxt = t;
  }
  for (int i = tid; i < 16 * 256; i += 256) A.y[(int64_t)blockIdx.x * 4096 + i] = cur[(i >> 8) * HLD + (i & 255)];
}

template <int DEPTH>
static float run(const Args& A, int groups) {
  const size_t ldsb = sizeof(float) * 2 * 16 * HLD;
  const int N = 10, iters = 5;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int k = 0; k < N; ++k) hipLaunchKernelGGL((chain16<DEPTH>), dim3(groups), dim3(256), ldsb, st, A);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 2; ++i) CK(hipGraphLaunch(ge, st));
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
  const int L = 12;
  std::vector<float> hw((size_t)L * 65536), hb(L * 256), hx;
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = ((float)((i * 2654435761u) >> 16 & 1023) / 1024.f - 0.5f) * 0.12f;
  for (size_t i = 0; i < hb.size(); ++i) hb[i] = 0.01f;
  float *W, *b, *x, *y;
  const int maxg = 4096;
  hx.resize((size_t)maxg * 4096);
  for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 40503u) >> 8 & 255) / 256.f;
  CK(hipMalloc(&W, hw.size() * 4)); CK(hipMalloc(&b, hb.size() * 4)); CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&y, hx.size() * 4));
  CK(hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  Args A; A.W = W; A.bias = b; A.x = x; A.y = y; A.layers = L;
  printf("chains of %d layers [256 -> 256], 16 rows per workgroup; time per launch, per layer and 16 rows on one CU "
         "(= us x 256 CUs / (groups x layers)), TFLOP/s\n", L);
  printf("(the product at 19 ranks: 8 rows per workgroup, 12.8 k cycles = 5.3 us per pair of workgroups and layer = 16 rows)\n");
  for (int groups : {256, 512, 768, 1024, 1216, 2048}) {
    if (argc > 1 && atoi(argv[1]) != groups) continue;
    const double flop = 2.0 * groups * 16 * 65536.0 * L;
#define RUN(D) { const float us = run<D>(A, groups); printf("  groups %4d  depth %2d : %8.2f us   %6.3f us per layer and 16 rows per CU   %6.1f TFLOP/s\n", groups, D, us, us * 256.0 / ((double)groups * L), flop / us * 1e-6); }
    RUN(4); RUN(8); RUN(16);
  }
  // checksum (the chain is deterministic)
  std::vector<float> hy(4096);
  CK(hipMemcpy(hy.data(), y, 4096 * 4, hipMemcpyDeviceToHost));
  double s = 0; for (float v : hy) s += v;
  printf("checksum of group 0: %.6f\n", s);
  return 0;
}
