// Micro-lab: a chain of L dependent 256x256x256 f32 layers (3 independent chains, 192 workgroups) as
//   (a) L launches in one hipGraph (what the product does), and
//   (b) ONE persistent launch with a grid-wide barrier between layers (atomic counter, agent scope).
// hipcc --offload-arch=gfx950 -O3 tools/barrier_lab.hip -o /tmp/barrier_lab && /tmp/barrier_lab.  Not part of the product.
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

struct Chain { float* buf[2]; const float* W; const float* b; };
struct Args { Chain c[3]; int L; unsigned* ctr; unsigned base; int mode; };

__device__ inline void layer_tile(const float* X, const float* W, const float* bias, float* Y, float* red) {
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

// one layer per launch; `l` selects the ping-pong side
__global__ __launch_bounds__(256) void k_layer(Args a, int l) {
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];
  const Chain& c = a.c[blockIdx.z];
  layer_tile(c.buf[l & 1], c.W, c.b, c.buf[(l + 1) & 1], red);
}

__device__ inline void grid_barrier(unsigned* ctr, unsigned target, int mode) {
  if (mode == 1) __atomic_thread_fence(__ATOMIC_RELEASE);           // every thread: agent-scope release
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
  if (mode == 1) __atomic_thread_fence(__ATOMIC_ACQUIRE);
}

// all L layers in one launch
__global__ __launch_bounds__(256) void k_persist(Args a) {
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];
  const Chain& c = a.c[blockIdx.z];
  const unsigned nwg = gridDim.x * gridDim.y * gridDim.z;
  for (int l = 0; l < a.L; ++l) {
    layer_tile(c.buf[l & 1], c.W, c.b, c.buf[(l + 1) & 1], red);
    if (l + 1 < a.L) grid_barrier(a.ctr, a.base + (unsigned)(l + 1) * nwg, a.mode);
  }
}

// barrier only (no work) to isolate its cost
__global__ __launch_bounds__(256) void k_barriers(Args a) {
  const unsigned nwg = gridDim.x * gridDim.y * gridDim.z;
  for (int l = 0; l + 1 < a.L; ++l) grid_barrier(a.ctr, a.base + (unsigned)(l + 1) * nwg, a.mode);
}

int main() {
  const int M = 256, N = 256, K = 256, NC = 3;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  std::vector<float> hX(M * K), hW(K * N), hb(N);
  srand(1);
  for (auto& v : hX) v = (rand() % 2001 - 1000) / 1000.f;
  for (auto& v : hW) v = (rand() % 2001 - 1000) / 4000.f;
  for (auto& v : hb) v = (rand() % 2001 - 1000) / 1000.f;
  Args a;
  float* dbuf[NC][2];
  for (int i = 0; i < NC; ++i) {
    float *W, *b;
    CK(hipMalloc(&dbuf[i][0], M * K * 4)); CK(hipMalloc(&dbuf[i][1], M * K * 4));
    CK(hipMalloc(&W, K * N * 4)); CK(hipMalloc(&b, N * 4));
    CK(hipMemcpy(W, hW.data(), K * N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice));
    a.c[i].buf[0] = dbuf[i][0]; a.c[i].buf[1] = dbuf[i][1]; a.c[i].W = W; a.c[i].b = b;
  }
  unsigned* ctr;
  CK(hipMalloc(&ctr, 4));
  CK(hipMemset(ctr, 0, 4));
  a.ctr = ctr; a.base = 0; a.mode = 0;
  auto reset_x = [&] { for (int i = 0; i < NC; ++i) CK(hipMemcpy(dbuf[i][0], hX.data(), M * K * 4, hipMemcpyHostToDevice)); };
  auto reference = [&](int L, std::vector<float>& out) {
    std::vector<float> x = hX, y(M * N);
    for (int l = 0; l < L; ++l) {
      for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
          double s = hb[n];
          for (int k = 0; k < K; ++k) s += (double)x[m * K + k] * hW[k * N + n];
          y[m * N + n] = (float)((s > 0 ? s : 0) * 0.25);
        }
      x = y;
    }
    out = x;
  };
  const dim3 grid(4, 16, NC);
  const unsigned nwg = 4 * 16 * NC;
  unsigned used = 0;   // barrier generations consumed so far (the counter only grows)
  for (int L : {2, 8}) {
    std::vector<float> ref, got(M * N);
    reference(L, ref);
    // (a) graph of L launches
    reset_x();
    for (int l = 0; l < L; ++l) hipLaunchKernelGGL(k_layer, grid, dim3(256), 0, st, a, l);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(got.data(), dbuf[1][L & 1], M * N * 4, hipMemcpyDeviceToHost));
    double e1 = 0; for (int i = 0; i < M * N; ++i) e1 = fmax(e1, fabs(got[i] - ref[i]));
    // (b) persistent, both fence modes
    for (int mode = 0; mode < 2; ++mode) {
      reset_x();
      a.L = L; a.mode = mode; a.base = used;
      hipLaunchKernelGGL(k_persist, grid, dim3(256), 0, st, a);
      CK(hipStreamSynchronize(st));
      used += (unsigned)(L - 1) * nwg;
      CK(hipMemcpy(got.data(), dbuf[1][L & 1], M * N * 4, hipMemcpyDeviceToHost));
      double e2 = 0; for (int i = 0; i < M * N; ++i) e2 = fmax(e2, fabs(got[i] - ref[i]));
      printf("L=%d  max err: launches %.2e, persistent(mode %d) %.2e\n", L, e1, mode, e2);
    }
  }
  // timings
  const int L = 8, reps = 300;
  hipEvent_t ea, eb;
  CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
  float ms;
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int l = 0; l < L; ++l) hipLaunchKernelGGL(k_layer, grid, dim3(256), 0, st, a, l);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(ea, st));
  for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(eb, st));
  CK(hipEventSynchronize(eb));
  CK(hipEventElapsedTime(&ms, ea, eb));
  printf("graph of %d dependent layer launches : %7.2f us per graph  (%.2f us per layer)\n", L, 1e3 * ms / reps, 1e3 * ms / reps / L);
  for (int mode = 0; mode < 2; ++mode) {
    for (int which = 0; which < 2; ++which) {
      a.L = L; a.mode = mode;
      CK(hipStreamSynchronize(st));
      CK(hipEventRecord(ea, st));
      for (int i = 0; i < reps; ++i) {
        a.base = used;
        if (which == 0) hipLaunchKernelGGL(k_persist, grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(k_barriers, grid, dim3(256), 0, st, a);
        used += (unsigned)(L - 1) * nwg;
      }
      CK(hipEventRecord(eb, st));
      CK(hipEventSynchronize(eb));
      CK(hipEventElapsedTime(&ms, ea, eb));
      printf("%s, fence mode %d : %7.2f us per launch  (%.2f us per layer / barrier)\n",
             which == 0 ? "persistent 8 layers " : "8-1 barriers only   ", mode, 1e3 * ms / reps, 1e3 * ms / reps / L);
    }
  }
  return 0;
}
