// Micro-lab (round 4): what does the boundary BETWEEN two row-local layers cost, and does a deeper hand-over of the weight
// stream remove it?  Same chain as tools/rowchain_lab.hip (R = 4 rows per workgroup, 256 x 256 layers, 4 waves split k,
// v_mfma_f32_4x4x1, two 16-row chunk buffers per wave), two disciplines:
//   MODE 0 (the product up to round 4): iteration c issues chunk c + 1 (the successor's chunk 0 at c = 3), then multiplies
//          chunk c.  While a layer's epilogue (partials -> LDS -> barrier -> sums -> barrier) and the next layer's start run,
//          ONE chunk (1 k cycles of the CU's fill path) is in flight: the fill path runs dry.
//   MODE 1: a layer starts with its chunks 0 AND 1 in flight.  After each group of 16 matrix instructions the 4 registers
//          they read are refilled with the same rows of the chunk two ahead (chunk c + 2; the successor's chunks 0 / 1 at
//          c = 2 / 3), so TWO chunks are queued when the epilogue begins.
// EPI: extra s_sleep rounds in the epilogue (64 cycles each): the product's epilogue is longer than the lab's (copies kept
// for the backward pass, stores for the weight gradients, argument fetches at the next layer's start).
//   MODE 4: MODE 0 with the 16 loads of the next chunk issued 4 at a time, in front of each group of 16 matrix instructions.
//   MODE 9: MODE 4, and the successor's chunk 1 is requested INSIDE the epilogue (8 loads in front of the barrier, 8 behind
//          the LDS reads of the partial sums), where the waves wait anyway.
//   MODE 2: the matrix instructions alone (no weight loads);  MODE 3: the weight stream alone (MODE 0's loads, 4 VALU
//          operations per 4 loaded registers instead of 16 matrix instructions).
//   hipcc --offload-arch=gfx950 -O3 tools/rowchain2_lab.hip -o /tmp/rowchain2_lab && /tmp/rowchain2_lab
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, 0)
#define H 256
#define HLD 264

__device__ inline f32x4 ldv(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ inline f32x4 zero4() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }

template <int MODE, int EPI>
__global__ __launch_bounds__(256) void rowchain(const float* __restrict__ X, const float* __restrict__ W,
                                                const float* __restrict__ bias, float* __restrict__ Y, int L, int B) {
  __shared__ __attribute__((aligned(16))) float hs[4 * HLD];
  __shared__ __attribute__((aligned(16))) float part[4 * 4 * H];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r0 = blockIdx.x * 4, chain = blockIdx.y;
  const float* Wc = W + (size_t)chain * L * H * H;
  const float* bc = bias + (size_t)chain * L * H;
  const float* Xc = X + (size_t)chain * B * H;
  float* Yc = Y + (size_t)chain * B * H;
  const unsigned long long t_start = __builtin_readcyclecounter();
  for (int i = tid; i < 4 * H / 4; i += 256) {
    const int r = i / (H / 4), c = (i % (H / 4)) * 4;
    *reinterpret_cast<f32x4*>(hs + r * HLD + c) = ldv(Xc + (size_t)(r0 + r) * H + c);
  }
  f32x4 b[2][16];
  const float* wl = Wc + (size_t)(64 * wave) * H + 4 * lane;         // this wave's k quarter of layer l
#pragma unroll
  for (int i = 0; i < 16; ++i) b[0][i] = ldv(wl + (size_t)i * H);
  if (MODE == 1 || MODE == 9) {
#pragma unroll
    for (int i = 0; i < 16; ++i) b[1][i] = ldv(wl + (size_t)(16 + i) * H);
  }
  __syncthreads();
  for (int l = 0; l < L; ++l) {
    f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
    const float bv = bc[(size_t)l * H + tid];
    const bool more = l + 1 < L;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float* nx4 = (c < 3) ? wl + (size_t)(16 * (c + 1)) * H : wl + (size_t)H * H;
      if (MODE == 0 || MODE == 3) {
        const float* nx = (c < 3) ? wl + (size_t)(16 * (c + 1)) * H : wl + (size_t)H * H;
        if (c < 3 || more) {
#pragma unroll
          for (int i = 0; i < 16; ++i) b[(c + 1) & 1][i] = ldv(nx + (size_t)i * H);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // MODE 1: the chunk two ahead: c + 2 of this layer, or chunk c - 2 of the next one
      const float* rf = (c < 2) ? wl + (size_t)(16 * (c + 2)) * H : wl + (size_t)H * H + (size_t)(16 * (c - 2)) * H;
      const bool refill = MODE == 1 && (c < 2 || more);
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) {
        if (MODE == 4 || (MODE == 9 && c > 0)) {
          if (c < 3 || more) {
#pragma unroll
            for (int i = 0; i < 4; ++i) b[(c + 1) & 1][4 * kq + i] = ldv(nx4 + (size_t)(4 * kq + i) * H);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        const f32x4 a = *reinterpret_cast<const f32x4*>(hs + (lane & 3) * HLD + 64 * wave + 16 * c + 4 * kq);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (MODE == 3) { if (e == 0) acc[s] += b[c & 1][4 * kq + s] * a[s]; }   // loads only: 16 VALU ops per chunk row group
            else acc[e] = MFMA4(a[s], b[c & 1][4 * kq + s][e], acc[e]);
          }
        if (MODE == 4 || MODE == 9) __builtin_amdgcn_sched_barrier(0);
        if (MODE == 1) {
          __builtin_amdgcn_sched_barrier(0);
          if (refill) {
#pragma unroll
            for (int i = 0; i < 4; ++i) b[c & 1][4 * kq + i] = ldv(rf + (size_t)(4 * kq + i) * H);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    wl += (size_t)H * H;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      f32x4 v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
      *reinterpret_cast<f32x4*>(part + ((wave * 4 + r) * H + 4 * lane)) = v;
    }
    if (MODE == 9 && more) {                                         // (wl = the next layer now) its chunk 1, first half
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 8; ++i) b[1][i] = ldv(wl + (size_t)(16 + i) * H);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPI; ++e) __builtin_amdgcn_s_sleep(1);
    float pp[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) pp[r][j] = part[(j * 4 + r) * H + tid];
    if (MODE == 9 && more) {                                         // second half, behind the LDS reads
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 8; i < 16; ++i) b[1][i] = ldv(wl + (size_t)(16 + i) * H);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s = (pp[r][0] + pp[r][1]) + (pp[r][2] + pp[r][3]);
      s = fmaxf(s + bv, 0.f);
      hs[r * HLD + tid] = s;
      if (l == L - 1) Yc[(size_t)(r0 + r) * H + tid] = s;
    }
    __syncthreads();
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0)               // (cycles of workgroup 0, behind the 3 chains' outputs)
    reinterpret_cast<unsigned long long*>(Y + (size_t)3 * B * H)[0] = __builtin_readcyclecounter() - t_start;
}

template <int MODE, int EPI>
static float run(const float* X, const float* W, const float* b, float* Y, int L, int B, int nch, int iters) {
  dim3 grid(B / 4, nch);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((rowchain<MODE, EPI>), grid, dim3(256), 0, 0, X, W, b, Y, L, B);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((rowchain<MODE, EPI>), grid, dim3(256), 0, 0, X, W, b, Y, L, B);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / iters;
}

template <int MODE, int EPI>
static void report(const char* name, const float* X, const float* W, const float* b, float* Y, int B) {
  for (int nch : {1, 3}) {
    const float t2 = run<MODE, EPI>(X, W, b, Y, 2, B, nch, 300);
    const float t8 = run<MODE, EPI>(X, W, b, Y, 8, B, nch, 300);
    const float t14 = run<MODE, EPI>(X, W, b, Y, 14, B, nch, 300);
    unsigned long long cyc = 0;
    CK(hipMemcpy(&cyc, Y + (size_t)3 * B * H, 8, hipMemcpyDeviceToHost));
    printf("%-34s extra epilogue %4d cycles, chains=%d (%3d WGs): L=2 %.2f us, L=8 %.2f us, L=14 %.2f us -> %.2f us per layer; "
           "workgroup 0 at L=14: %llu cycles = %.2f GHz if it lasts the launch - 2 us\n",
           name, 64 * EPI, nch, B / 4 * nch, t2, t8, t14, (t14 - t2) / 12.f, cyc, cyc / ((t14 - 2.0) * 1000.0));
  }
}

void raw_main(const float* W, float* Y);
void ta_main(const float* W, float* Y);
void ld_main(const float* X, const float* W, const float* b, float* Y, int B);
void lr_main(const float* X, const float* W, const float* b, float* Y, int B);
void l7_main(const float* X, const float* W, const float* b, float* Y, int B);
void w8_main(const float* X, const float* W, const float* b, float* Y, int B);
int main() {
  const int B = 256, LMAX = 16, NCH = 3;
  std::vector<float> hX((size_t)NCH * B * H), hW((size_t)NCH * LMAX * H * H), hb((size_t)NCH * LMAX * H);
  srand(1);
  for (auto& v : hX) v = (float)rand() / RAND_MAX - 0.5f;
  const float lim = sqrtf(6.0f / (H + H));
  for (auto& v : hW) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * lim * 1.4f;
  for (auto& v : hb) v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
  float *X, *W, *b, *Y;
  CK(hipMalloc(&X, hX.size() * 4)); CK(hipMalloc(&W, hW.size() * 4)); CK(hipMalloc(&b, hb.size() * 4));
  CK(hipMalloc(&Y, hX.size() * 4 + 64));
  CK(hipMemcpy(X, hX.data(), hX.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  // the two disciplines accumulate in the same order: bit-identical outputs
  {
    std::vector<float> y0((size_t)B * H), y1((size_t)B * H);
    for (int L : {1, 2, 5}) {
      CK(hipMemset(Y, 0, hX.size() * 4));
      hipLaunchKernelGGL((rowchain<0, 0>), dim3(B / 4, 1), dim3(256), 0, 0, X, W, b, Y, L, B);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(y0.data(), Y, y0.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemset(Y, 0, hX.size() * 4));
      hipLaunchKernelGGL((rowchain<1, 0>), dim3(B / 4, 1), dim3(256), 0, 0, X, W, b, Y, L, B);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(y1.data(), Y, y1.size() * 4, hipMemcpyDeviceToHost));
      double amax = 0;
      for (float v : y0) amax = fmax(amax, fabs(v));
      printf("check L=%d: MODE 1 %s MODE 0 bit for bit (max |y| %.3f)\n", L,
             memcmp(y0.data(), y1.data(), y0.size() * 4) == 0 ? "==" : "!=", amax);
    }
  }
  raw_main(W, Y);
  ta_main(W, Y);
  ld_main(X, W, b, Y, B);
  lr_main(X, W, b, Y, B);
  l7_main(X, W, b, Y, B);
  w8_main(X, W, b, Y, B);
  report<0, 0>("one chunk across the boundary", X, W, b, Y, B);
  report<1, 0>("two chunks across the boundary", X, W, b, Y, B);
  report<4, 0>("one chunk, loads 4 by 4", X, W, b, Y, B);
  report<9, 0>("4 by 4 + chunk 1 in the epilogue", X, W, b, Y, B);
  report<9, 8>("4 by 4 + chunk 1 in the epilogue", X, W, b, Y, B);
  report<2, 0>("matrix instructions only", X, W, b, Y, B);
  report<3, 0>("weight stream only (loads first)", X, W, b, Y, B);
  report<0, 8>("one chunk across the boundary", X, W, b, Y, B);
  report<1, 8>("two chunks across the boundary", X, W, b, Y, B);
  report<0, 16>("one chunk across the boundary", X, W, b, Y, B);
  report<1, 16>("two chunks across the boundary", X, W, b, Y, B);
  return 0;
}

// ---- the bare weight stream: every workgroup reads L x 256 KB (the layers' matrices, 1 KB rows, 16 rows per batch and
// wave) and does nothing with it.  WAVES = 4 or 8 (k split over the waves); DMA: global_load_lds_dwordx4 into a per-wave
// LDS area instead of registers.  What a CU can take in, whatever the consumer does.
template <int WAVES, int DMA>
__global__ __launch_bounds__(64 * WAVES) void stream_raw(const float* __restrict__ W, float* __restrict__ Y, int L) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) float ring[WAVES * 16 * 256];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  constexpr int KW = 256 / WAVES;                                 // k rows per wave and layer
  const float* wl = W + (size_t)blockIdx.y * L * H * H + (size_t)(KW * wave) * H + 4 * lane;
  f32x4 keep = zero4();
  for (int l = 0; l < L; ++l) {
    for (int c = 0; c < KW / 16; ++c) {
      const float* p = wl + (size_t)(16 * c) * H;
      if (DMA) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
          __builtin_amdgcn_global_load_lds(p + (size_t)i * H, ring + (wave * 16 + i) * 256, 16, 0, 0);
      } else {
        f32x4 b[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) b[i] = ldv(p + (size_t)i * H);
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" :: "v"(b[i]));
      }
    }
    wl += (size_t)H * H;
  }
  if (DMA) {
    __builtin_amdgcn_s_waitcnt(0);
    keep = *reinterpret_cast<const f32x4*>(ring + wave * 4096 + 4 * lane);
  }
  if (keep[0] == 123.456f) Y[tid] = keep[1];
#endif
}

template <int WAVES, int DMA>
static void report_raw(const char* name, const float* W, float* Y) {
  for (int nch : {1, 3}) {
    float t[2];
    const int Ls[2] = {2, 14};
    for (int k = 0; k < 2; ++k) {
      dim3 grid(64, nch);
      for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((stream_raw<WAVES, DMA>), grid, dim3(64 * WAVES), 0, 0, W, Y, Ls[k]);
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 300; ++i) hipLaunchKernelGGL((stream_raw<WAVES, DMA>), grid, dim3(64 * WAVES), 0, 0, W, Y, Ls[k]);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1));
      t[k] = ms * 1000.f / 300;
    }
    const float per = (t[1] - t[0]) / 12.f;
    printf("%-40s chains=%d (%3d WGs): L=2 %.2f us, L=14 %.2f us -> %.2f us per 256 KB = %.0f GB/s per CU\n", name, nch,
           64 * nch, t[0], t[1], per, 262144.0 / per / 1000.0);
  }
}

void raw_main(const float* W, float* Y) {
  report_raw<4, 0>("bare stream, 4 waves, registers", W, Y);
  report_raw<8, 0>("bare stream, 8 waves, registers", W, Y);
  report_raw<4, 1>("bare stream, 4 waves, LDS-DMA", W, Y);
  report_raw<8, 1>("bare stream, 8 waves, LDS-DMA", W, Y);
}

// ---- MODE 5: loader waves.  512 threads: waves 0-3 compute (the same k quarters, the same accumulation order), waves 4-7
// do nothing but stream: wave 4 + w feeds wave w through a ring of 3 half-chunks (8 rows = 8 KB per wave and slot, 96 KB
// in all) filled by LDS-DMA (global_load_lds_dwordx4, no registers).  One workgroup barrier per half-chunk g orders
// everything: the loader arrives when half-chunk g has landed (counted vmcnt: g + 1 may still be in flight), a compute wave
// when it has consumed g - 1; behind the barrier the loader refills the slot of g - 1 with g + 2.  A wave that only issues
// loads never waits for matrix instructions, and the matrix waves never wait for the texture path to accept a load.
#define SBAR() __builtin_amdgcn_s_barrier()
template <int EPI>
__global__ __launch_bounds__(512) void rowchain_ld(const float* __restrict__ X, const float* __restrict__ W,
                                                   const float* __restrict__ bias, float* __restrict__ Y, int L, int B) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) float hs[4 * HLD];
  __shared__ __attribute__((aligned(16))) float part[4 * 4 * H];
  __shared__ __attribute__((aligned(16))) float ring[3 * 4 * 8 * 256];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int w = wave & 3;
  const int r0 = blockIdx.x * 4, chain = blockIdx.y;
  const float* Wc = W + (size_t)chain * L * H * H;
  const float* bc = bias + (size_t)chain * L * H;
  const float* Xc = X + (size_t)chain * B * H;
  float* Yc = Y + (size_t)chain * B * H;
  const int N = 8 * L;
  if (wave >= 4) {
    // ------------------------------------------------ loader
    const float* src = Wc + (size_t)(64 * w) * H + 4 * lane;        // + g * 8 rows (+ 192 rows more at every layer end)
    auto issue = [&](int g) {
      const int l = g >> 3, h = g & 7;
      const float* p = src + ((size_t)l * H + 8 * h) * H;
      float* dst = ring + ((g % 3) * 4 + w) * 8 * 256;
#pragma unroll
      for (int i = 0; i < 8; ++i) __builtin_amdgcn_global_load_lds(p + (size_t)i * H, dst + i * 256, 16, 0, 0);
    };
    issue(0);
    if (N > 1) issue(1);
    SBAR();                                                          // (the compute waves' input rows)
    for (int g = 0; g < N; ++g) {
      if (g + 1 < N) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      SBAR();                                                        // barrier(g)
      if (g + 2 < N) issue(g + 2);
      if ((g & 7) == 7) SBAR();                                      // the layer's epilogue barrier
    }
    return;
  }
  // -------------------------------------------------- compute
  for (int i = tid; i < 4 * H / 4; i += 256) {
    const int r = i / (H / 4), c = (i % (H / 4)) * 4;
    *reinterpret_cast<f32x4*>(hs + r * HLD + c) = ldv(Xc + (size_t)(r0 + r) * H + c);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  SBAR();
  int g = 0;
  for (int l = 0; l < L; ++l) {
    f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
    const float bv = bc[(size_t)l * H + tid];
#pragma unroll
    for (int h = 0; h < 8; ++h, ++g) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // (hs of the previous layer / nothing)
      SBAR();                                                        // barrier(g)
      const float* slot = ring + ((g % 3) * 4 + w) * 8 * 256 + 4 * lane;
      f32x4 b[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) b[i] = *reinterpret_cast<const f32x4*>(slot + i * 256);
      f32x4 a[2];
#pragma unroll
      for (int kq = 0; kq < 2; ++kq)
        a[kq] = *reinterpret_cast<const f32x4*>(hs + (lane & 3) * HLD + 64 * w + 8 * h + 4 * kq);
      __builtin_amdgcn_sched_barrier(0);                             // (all LDS reads of the half-chunk issued first)
#pragma unroll
      for (int kq = 0; kq < 2; ++kq) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = MFMA4(a[kq][s], b[4 * kq + s][e], acc[e]);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      f32x4 v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
      *reinterpret_cast<f32x4*>(part + ((w * 4 + r) * H + 4 * lane)) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    SBAR();                                                          // the epilogue barrier
#pragma unroll
    for (int e = 0; e < EPI; ++e) __builtin_amdgcn_s_sleep(1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s = (part[(0 * 4 + r) * H + tid] + part[(1 * 4 + r) * H + tid]) +
                (part[(2 * 4 + r) * H + tid] + part[(3 * 4 + r) * H + tid]);
      s = fmaxf(s + bv, 0.f);
      hs[r * HLD + tid] = s;
      if (l == L - 1) Yc[(size_t)(r0 + r) * H + tid] = s;
    }
  }
#endif
}

template <int EPI>
static float run_ld(const float* X, const float* W, const float* b, float* Y, int L, int B, int nch, int iters) {
  dim3 grid(B / 4, nch);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((rowchain_ld<EPI>), grid, dim3(512), 0, 0, X, W, b, Y, L, B);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((rowchain_ld<EPI>), grid, dim3(512), 0, 0, X, W, b, Y, L, B);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / iters;
}
template <int EPI>
static void report_ld(const float* X, const float* W, const float* b, float* Y, int B) {
  for (int nch : {1, 3}) {
    const float t2 = run_ld<EPI>(X, W, b, Y, 2, B, nch, 300);
    const float t8 = run_ld<EPI>(X, W, b, Y, 8, B, nch, 300);
    const float t14 = run_ld<EPI>(X, W, b, Y, 14, B, nch, 300);
    printf("%-34s extra epilogue %4d cycles, chains=%d (%3d WGs): L=2 %.2f us, L=8 %.2f us, L=14 %.2f us -> %.2f us per layer\n",
           "loader waves (LDS-DMA ring)", 64 * EPI, nch, B / 4 * nch, t2, t8, t14, (t14 - t2) / 12.f);
  }
}

void ld_main(const float* X, const float* W, const float* b, float* Y, int B) {
  std::vector<float> y0((size_t)B * H), y1((size_t)B * H);
  for (int L : {1, 2, 5}) {
    CK(hipMemset(Y, 0, (size_t)B * H * 4));
    hipLaunchKernelGGL((rowchain<0, 0>), dim3(B / 4, 1), dim3(256), 0, 0, X, W, b, Y, L, B);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(y0.data(), Y, y0.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemset(Y, 0, (size_t)B * H * 4));
    hipLaunchKernelGGL((rowchain_ld<0>), dim3(B / 4, 1), dim3(512), 0, 0, X, W, b, Y, L, B);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(y1.data(), Y, y1.size() * 4, hipMemcpyDeviceToHost));
    printf("check L=%d: loader waves %s MODE 0 bit for bit\n", L, memcmp(y0.data(), y1.data(), y0.size() * 4) == 0 ? "==" : "!=");
  }
  report_ld<0>(X, W, b, Y, B);
  report_ld<8>(X, W, b, Y, B);
  report_ld<16>(X, W, b, Y, B);
}

// ---- MODE 6: loader waves that stage through REGISTERS.  The LDS-DMA ring above can keep only what fits in LDS in flight
// (64 KB with 3 slots of 32 KB) and a load needs ~1 k cycles from issue to landing: the ring runs dry.  Here a loader wave
// keeps 3 half-chunks (24 rows = 96 registers) in flight in its own registers, whatever the compute waves do, and hands a
// landed half-chunk over with 8 ds_write_b128 into a ring of only 2 slots; one workgroup barrier per half-chunk as before.
template <int EPI>
__global__ __launch_bounds__(512) void rowchain_lr(const float* __restrict__ X, const float* __restrict__ W,
                                                   const float* __restrict__ bias, float* __restrict__ Y, int L, int B) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) float hs[4 * HLD];
  __shared__ __attribute__((aligned(16))) float part[4 * 4 * H];
  __shared__ __attribute__((aligned(16))) float ring[2 * 4 * 8 * 256];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int w = wave & 3;
  const int r0 = blockIdx.x * 4, chain = blockIdx.y;
  const float* Wc = W + (size_t)chain * L * H * H;
  const float* bc = bias + (size_t)chain * L * H;
  const float* Xc = X + (size_t)chain * B * H;
  float* Yc = Y + (size_t)chain * B * H;
  const int N = 8 * L;
  if (wave >= 4) {
    // ------------------------------------------------ loader
    const float* src = Wc + (size_t)(64 * w) * H + 4 * lane;
    f32x4 q[3][8];
#define LR_ISSUE(bank, g)                                                            \
    do {                                                                             \
      const float* p_ = src + ((size_t)((g) >> 3) * H + 8 * ((g) & 7)) * H;          \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) q[bank][i] = ldv(p_ + (size_t)i * H); \
    } while (0)
#define LR_STEP(bank, g)                                                             \
    do {                                                                             \
      if ((g) < N) {                                                                 \
        float* dst_ = ring + ((((g) & 1) * 4 + w) * 8) * 256 + 4 * lane;             \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(dst_ + i * 256) = q[bank][i]; \
        __builtin_amdgcn_sched_barrier(0);                                           \
        if ((g) + 3 < N) LR_ISSUE(bank, (g) + 3);                                    \
        __builtin_amdgcn_sched_barrier(0);                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                           \
        SBAR();                                                                      \
        if (((g) & 7) == 7) SBAR();                                                  \
      }                                                                              \
    } while (0)
    LR_ISSUE(0, 0);
    if (N > 1) LR_ISSUE(1, 1);
    if (N > 2) LR_ISSUE(2, 2);
    SBAR();                                                          // (the compute waves' input rows)
    for (int g = 0; g < N; g += 3) {
      LR_STEP(0, g);
      LR_STEP(1, g + 1);
      LR_STEP(2, g + 2);
    }
    return;
  }
  // -------------------------------------------------- compute
  for (int i = tid; i < 4 * H / 4; i += 256) {
    const int r = i / (H / 4), c = (i % (H / 4)) * 4;
    *reinterpret_cast<f32x4*>(hs + r * HLD + c) = ldv(Xc + (size_t)(r0 + r) * H + c);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  SBAR();
  int g = 0;
  for (int l = 0; l < L; ++l) {
    f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
    const float bv = bc[(size_t)l * H + tid];
#pragma unroll
    for (int h = 0; h < 8; ++h, ++g) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      SBAR();                                                        // barrier(g): half-chunk g is in slot g & 1
      const float* slot = ring + (((g & 1) * 4 + w) * 8) * 256 + 4 * lane;
      f32x4 b[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) b[i] = *reinterpret_cast<const f32x4*>(slot + i * 256);
      f32x4 a[2];
#pragma unroll
      for (int kq = 0; kq < 2; ++kq)
        a[kq] = *reinterpret_cast<const f32x4*>(hs + (lane & 3) * HLD + 64 * w + 8 * h + 4 * kq);
      __builtin_amdgcn_sched_barrier(0);                             // (all LDS reads of the half-chunk issued first)
#pragma unroll
      for (int kq = 0; kq < 2; ++kq) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = MFMA4(a[kq][s], b[4 * kq + s][e], acc[e]);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      f32x4 v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
      *reinterpret_cast<f32x4*>(part + ((w * 4 + r) * H + 4 * lane)) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    SBAR();                                                          // the epilogue barrier
#pragma unroll
    for (int e = 0; e < EPI; ++e) __builtin_amdgcn_s_sleep(1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s = (part[(0 * 4 + r) * H + tid] + part[(1 * 4 + r) * H + tid]) +
                (part[(2 * 4 + r) * H + tid] + part[(3 * 4 + r) * H + tid]);
      s = fmaxf(s + bv, 0.f);
      hs[r * HLD + tid] = s;
      if (l == L - 1) Yc[(size_t)(r0 + r) * H + tid] = s;
    }
  }
#endif
}

template <int EPI>
static float run_lr(const float* X, const float* W, const float* b, float* Y, int L, int B, int nch, int iters) {
  dim3 grid(B / 4, nch);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((rowchain_lr<EPI>), grid, dim3(512), 0, 0, X, W, b, Y, L, B);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((rowchain_lr<EPI>), grid, dim3(512), 0, 0, X, W, b, Y, L, B);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / iters;
}
template <int EPI>
static void report_lr(const float* X, const float* W, const float* b, float* Y, int B) {
  for (int nch : {1, 3}) {
    const float t2 = run_lr<EPI>(X, W, b, Y, 2, B, nch, 300);
    const float t8 = run_lr<EPI>(X, W, b, Y, 8, B, nch, 300);
    const float t14 = run_lr<EPI>(X, W, b, Y, 14, B, nch, 300);
    printf("%-34s extra epilogue %4d cycles, chains=%d (%3d WGs): L=2 %.2f us, L=8 %.2f us, L=14 %.2f us -> %.2f us per layer\n",
           "loader waves (register staged)", 64 * EPI, nch, B / 4 * nch, t2, t8, t14, (t14 - t2) / 12.f);
  }
}

void lr_main(const float* X, const float* W, const float* b, float* Y, int B) {
  std::vector<float> y0((size_t)B * H), y1((size_t)B * H);
  for (int L : {1, 2, 5}) {
    CK(hipMemset(Y, 0, (size_t)B * H * 4));
    hipLaunchKernelGGL((rowchain<0, 0>), dim3(B / 4, 1), dim3(256), 0, 0, X, W, b, Y, L, B);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(y0.data(), Y, y0.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemset(Y, 0, (size_t)B * H * 4));
    hipLaunchKernelGGL((rowchain_lr<0>), dim3(B / 4, 1), dim3(512), 0, 0, X, W, b, Y, L, B);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(y1.data(), Y, y1.size() * 4, hipMemcpyDeviceToHost));
    printf("check L=%d: register-staged loader waves %s MODE 0 bit for bit\n", L,
           memcmp(y0.data(), y1.data(), y0.size() * 4) == 0 ? "==" : "!=");
  }
  report_lr<0>(X, W, b, Y, B);
  report_lr<8>(X, W, b, Y, B);
  report_lr<16>(X, W, b, Y, B);
}

// ---- MODE 7: register-staged loader waves, branch-free, one layer per loop iteration (so that hipcc's own vmcnt
// bookkeeping stays exact: 2 banks of 8 rows per loader wave, always reloaded in the same order -- past the end of the
// chain with the last half-chunk again), ring of 2 slots, compute waves read half-chunk g into registers while they multiply
// g - 1.  Barrier order on both sides:  ... barrier(g7) | E1 (the epilogue's) | barrier(g0') ...; the loader executes E1 at
// the end of its step for g0', i.e. with g0' written to the ring and g1', g2' in flight.
template <int EPI>
__global__ __launch_bounds__(512) void rowchain_l7(const float* __restrict__ X, const float* __restrict__ W,
                                                   const float* __restrict__ bias, float* __restrict__ Y, int L, int B) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) float hs[4 * HLD];
  __shared__ __attribute__((aligned(16))) float part[4 * 4 * H];
  __shared__ __attribute__((aligned(16))) float ring[2 * 4 * 8 * 256];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int w = wave & 3;
  const int r0 = blockIdx.x * 4, chain = blockIdx.y;
  const float* Wc = W + (size_t)chain * L * H * H;
  const float* bc = bias + (size_t)chain * L * H;
  const float* Xc = X + (size_t)chain * B * H;
  float* Yc = Y + (size_t)chain * B * H;
  const int N = 8 * L;
  if (wave >= 4) {
    // ------------------------------------------------ loader
    const float* src = Wc + (size_t)(64 * w) * H + 4 * lane;
    f32x4 q[2][8];
    auto rows = [&](int g) {                                         // (clamped: the tail reloads the last half-chunk)
      const int gc = g < N ? g : N - 1;
      return src + ((size_t)(gc >> 3) * H + 8 * (gc & 7)) * H;
    };
    {
      const float* p0 = rows(0);
      const float* p1 = rows(1);
#pragma unroll
      for (int i = 0; i < 8; ++i) q[0][i] = ldv(p0 + (size_t)i * H);
#pragma unroll
      for (int i = 0; i < 8; ++i) q[1][i] = ldv(p1 + (size_t)i * H);
    }
    SBAR();                                                          // (the compute waves' input rows)
    for (int l = 0; l < L; ++l) {
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        const int g = 8 * l + h;
        float* dst = ring + ((((h & 1) * 4 + w) * 8) * 256) + 4 * lane;
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(dst + i * 256) = q[h & 1][i];
        __builtin_amdgcn_sched_barrier(0);
        const float* p = rows(g + 2);
#pragma unroll
        for (int i = 0; i < 8; ++i) q[h & 1][i] = ldv(p + (size_t)i * H);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (h == 0 && l > 0) SBAR();                                 // E1 of the previous layer
        SBAR();                                                      // barrier(g)
      }
    }
    SBAR();                                                          // E1 of the last layer
    return;
  }
  // -------------------------------------------------- compute
  for (int i = tid; i < 4 * H / 4; i += 256) {
    const int r = i / (H / 4), c = (i % (H / 4)) * 4;
    *reinterpret_cast<f32x4*>(hs + r * HLD + c) = ldv(Xc + (size_t)(r0 + r) * H + c);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  SBAR();
  for (int l = 0; l < L; ++l) {
    f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
    const float bv = bc[(size_t)l * H + tid];
    f32x4 b[2][8], a[2][2];
#pragma unroll
    for (int h = 0; h < 9; ++h) {
      if (h < 8) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // reads of g - 1 done (and hs of the layer before)
        SBAR();                                                      // barrier(g): half-chunk g is in slot h & 1
        const float* slot = ring + ((((h & 1) * 4 + w) * 8) * 256) + 4 * lane;
#pragma unroll
        for (int i = 0; i < 8; ++i) b[h & 1][i] = *reinterpret_cast<const f32x4*>(slot + i * 256);
#pragma unroll
        for (int kq = 0; kq < 2; ++kq)
          a[h & 1][kq] = *reinterpret_cast<const f32x4*>(hs + (lane & 3) * HLD + 64 * w + 8 * h + 4 * kq);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (h > 0) {
#pragma unroll
        for (int kq = 0; kq < 2; ++kq)
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = MFMA4(a[(h - 1) & 1][kq][s], b[(h - 1) & 1][4 * kq + s][e], acc[e]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      f32x4 v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
      *reinterpret_cast<f32x4*>(part + ((w * 4 + r) * H + 4 * lane)) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    SBAR();                                                          // E1
#pragma unroll
    for (int e = 0; e < EPI; ++e) __builtin_amdgcn_s_sleep(1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s = (part[(0 * 4 + r) * H + tid] + part[(1 * 4 + r) * H + tid]) +
                (part[(2 * 4 + r) * H + tid] + part[(3 * 4 + r) * H + tid]);
      s = fmaxf(s + bv, 0.f);
      hs[r * HLD + tid] = s;
      if (l == L - 1) Yc[(size_t)(r0 + r) * H + tid] = s;
    }
  }
#endif
}

template <int EPI>
static float run_l7(const float* X, const float* W, const float* b, float* Y, int L, int B, int nch, int iters) {
  dim3 grid(B / 4, nch);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((rowchain_l7<EPI>), grid, dim3(512), 0, 0, X, W, b, Y, L, B);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((rowchain_l7<EPI>), grid, dim3(512), 0, 0, X, W, b, Y, L, B);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / iters;
}
template <int EPI>
static void report_l7(const float* X, const float* W, const float* b, float* Y, int B) {
  for (int nch : {1, 3}) {
    const float t2 = run_l7<EPI>(X, W, b, Y, 2, B, nch, 300);
    const float t8 = run_l7<EPI>(X, W, b, Y, 8, B, nch, 300);
    const float t14 = run_l7<EPI>(X, W, b, Y, 14, B, nch, 300);
    printf("%-34s extra epilogue %4d cycles, chains=%d (%3d WGs): L=2 %.2f us, L=8 %.2f us, L=14 %.2f us -> %.2f us per layer\n",
           "loader waves (2 banks, pipelined)", 64 * EPI, nch, B / 4 * nch, t2, t8, t14, (t14 - t2) / 12.f);
  }
}

void l7_main(const float* X, const float* W, const float* b, float* Y, int B) {
  std::vector<float> y0((size_t)B * H), y1((size_t)B * H);
  for (int L : {1, 2, 5}) {
    CK(hipMemset(Y, 0, (size_t)B * H * 4));
    hipLaunchKernelGGL((rowchain<0, 0>), dim3(B / 4, 1), dim3(256), 0, 0, X, W, b, Y, L, B);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(y0.data(), Y, y0.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemset(Y, 0, (size_t)B * H * 4));
    hipLaunchKernelGGL((rowchain_l7<0>), dim3(B / 4, 1), dim3(512), 0, 0, X, W, b, Y, L, B);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(y1.data(), Y, y1.size() * 4, hipMemcpyDeviceToHost));
    printf("check L=%d: pipelined loader waves %s MODE 0 bit for bit\n", L,
           memcmp(y0.data(), y1.data(), y0.size() * 4) == 0 ? "==" : "!=");
  }
  report_l7<0>(X, W, b, Y, B);
  report_l7<8>(X, W, b, Y, B);
  report_l7<16>(X, W, b, Y, B);
}

// ---- MODE 8: 8 waves, k split 8 ways (32 rows of every matrix per wave: 2 chunks of 16), everything else as MODE 0.
// Two waves share a SIMD: while one is held at the issue of a load the other issues matrix instructions.  The order of
// the k summation changes (8 partial sums instead of 4).
template <int EPI>
__global__ __launch_bounds__(512) void rowchain_8w(const float* __restrict__ X, const float* __restrict__ W,
                                                   const float* __restrict__ bias, float* __restrict__ Y, int L, int B) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) float hs[4 * HLD];
  __shared__ __attribute__((aligned(16))) float part[8 * 4 * H];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r0 = blockIdx.x * 4, chain = blockIdx.y;
  const float* Wc = W + (size_t)chain * L * H * H;
  const float* bc = bias + (size_t)chain * L * H;
  const float* Xc = X + (size_t)chain * B * H;
  float* Yc = Y + (size_t)chain * B * H;
  for (int i = tid; i < 4 * H / 4; i += 512) {
    const int r = i / (H / 4), c = (i % (H / 4)) * 4;
    *reinterpret_cast<f32x4*>(hs + r * HLD + c) = ldv(Xc + (size_t)(r0 + r) * H + c);
  }
  f32x4 b[2][16];
  const float* wl = Wc + (size_t)(32 * wave) * H + 4 * lane;
#pragma unroll
  for (int i = 0; i < 16; ++i) b[0][i] = ldv(wl + (size_t)i * H);
  __syncthreads();
  const int col = tid & 255, rh = tid >> 8;                          // epilogue: thread -> column, rows 2 rh, 2 rh + 1
  for (int l = 0; l < L; ++l) {
    f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
    const float bv = bc[(size_t)l * H + col];
    const bool more = l + 1 < L;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float* nx = (c < 1) ? wl + (size_t)16 * H : wl + (size_t)H * H;
      if (c < 1 || more) {
#pragma unroll
        for (int i = 0; i < 16; ++i) b[(c + 1) & 1][i] = ldv(nx + (size_t)i * H);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(hs + (lane & 3) * HLD + 32 * wave + 16 * c + 4 * kq);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = MFMA4(a[s], b[c & 1][4 * kq + s][e], acc[e]);
      }
    }
    wl += (size_t)H * H;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      f32x4 v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
      *reinterpret_cast<f32x4*>(part + ((wave * 4 + r) * H + 4 * lane)) = v;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPI; ++e) __builtin_amdgcn_s_sleep(1);
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int r = 2 * rh + rr;
      float s = ((part[(0 * 4 + r) * H + col] + part[(1 * 4 + r) * H + col]) +
                 (part[(2 * 4 + r) * H + col] + part[(3 * 4 + r) * H + col])) +
                ((part[(4 * 4 + r) * H + col] + part[(5 * 4 + r) * H + col]) +
                 (part[(6 * 4 + r) * H + col] + part[(7 * 4 + r) * H + col]));
      s = fmaxf(s + bv, 0.f);
      hs[r * HLD + col] = s;
      if (l == L - 1) Yc[(size_t)(r0 + r) * H + col] = s;
    }
    __syncthreads();
  }
#endif
}

template <int EPI>
static void report_8w(const float* X, const float* W, const float* b, float* Y, int B) {
  for (int nch : {1, 3}) {
    float t[3];
    const int Ls[3] = {2, 8, 14};
    for (int k = 0; k < 3; ++k) {
      dim3 grid(B / 4, nch);
      for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((rowchain_8w<EPI>), grid, dim3(512), 0, 0, X, W, b, Y, Ls[k], B);
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 300; ++i) hipLaunchKernelGGL((rowchain_8w<EPI>), grid, dim3(512), 0, 0, X, W, b, Y, Ls[k], B);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1));
      t[k] = ms * 1000.f / 300;
    }
    printf("%-34s extra epilogue %4d cycles, chains=%d (%3d WGs): L=2 %.2f us, L=8 %.2f us, L=14 %.2f us -> %.2f us per layer\n",
           "8 waves, k split 8 ways", 64 * EPI, nch, B / 4 * nch, t[0], t[1], t[2], (t[2] - t[0]) / 12.f);
  }
}

void w8_main(const float* X, const float* W, const float* b, float* Y, int B) {
  std::vector<float> y0((size_t)B * H), y1((size_t)B * H);
  for (int L : {1, 5}) {
    CK(hipMemset(Y, 0, (size_t)B * H * 4));
    hipLaunchKernelGGL((rowchain<0, 0>), dim3(B / 4, 1), dim3(256), 0, 0, X, W, b, Y, L, B);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(y0.data(), Y, y0.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemset(Y, 0, (size_t)B * H * 4));
    hipLaunchKernelGGL((rowchain_8w<0>), dim3(B / 4, 1), dim3(512), 0, 0, X, W, b, Y, L, B);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(y1.data(), Y, y1.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (size_t i = 0; i < y0.size(); ++i) maxerr = fmax(maxerr, fabs((double)y0[i] - y1[i]));
    printf("check L=%d: 8 waves vs MODE 0: max abs difference %.3e\n", L, maxerr);
  }
  report_8w<0>(X, W, b, Y, B);
  report_8w<8>(X, W, b, Y, B);
}

// ---- what one vector-memory instruction costs the CU's texture path, by shape (4 waves per workgroup, 64 workgroups, data
// L2-resident): PAT 0 = one 1 KB row per wave instruction (global_load_dwordx4, the weight stream); PAT 1 = the dW tiles' X
// operand: dword per lane, 16 lanes x 4 bytes contiguous in each of 4 rows 1 KB apart (4 x 64 B per instruction);
// PAT 2 = the dW tiles' dY operand: dwordx4 per lane, 16 lanes x 16 bytes in each of 4 rows (4 x 256 B per instruction).
template <int PAT>
__global__ __launch_bounds__(256) void ta_cost(const float* __restrict__ W, float* __restrict__ Y, int iters) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, q = lane >> 4;
  const float* base = W + (size_t)blockIdx.x * 4096 + (size_t)wave * 64 * H;
  f32x4 keep = zero4();
  for (int it = 0; it < iters; ++it) {
    const float* p = base + (size_t)((it & 3) * 16) * H;
    if (PAT == 0) {
      f32x4 b[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) b[i] = ldv(p + (size_t)i * H + 4 * lane);
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("" :: "v"(b[i]));
    } else if (PAT == 1) {
      float b[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) b[i] = p[(size_t)((i >> 2) * 16 + 4 * q + (i & 3)) * H + j];
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("" :: "v"(b[i]));
    } else {
      f32x4 b[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) b[i] = ldv(p + (size_t)((i >> 2) * 16 + 4 * q + (i & 3)) * H + 4 * j);
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("" :: "v"(b[i]));
    }
  }
  if (keep[0] == 123.456f) Y[tid] = keep[1];
#endif
}
template <int PAT>
static void report_ta(const char* name, const float* W, float* Y) {
  float t[2];
  const int its[2] = {64, 576};
  for (int k = 0; k < 2; ++k) {
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((ta_cost<PAT>), dim3(64), dim3(256), 0, 0, W, Y, its[k]);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((ta_cost<PAT>), dim3(64), dim3(256), 0, 0, W, Y, its[k]);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    t[k] = ms * 1000.f / 200;
  }
  const double per = (t[1] - t[0]) * 1000.0 / (512.0 * 16 * 4);       // ns per wave instruction and CU (4 waves issue)
  printf("%-44s %.1f ns per wave instruction per CU = %.0f cycles at 2.4 GHz\n", name, per, per * 2.4);
}
void ta_main(const float* W, float* Y) {
  report_ta<0>("dwordx4, one 1 KB row", W, Y);
  report_ta<1>("dword, 4 rows x 64 B (dW tiles' X)", W, Y);
  report_ta<2>("dwordx4, 4 rows x 256 B (dW tiles' dY)", W, Y);
}
