// K10 epilogue: exploration noise, clip and epsilon-greedy mix on the actor output.
//
// Replaces (reference): DDPG.get_actions post-processing ddpg.py:149-152 (+ _random_action ddpg.py:114-115).
// NumPy promotion is reproduced: the float32 action is updated in place with float64 operands, i.e. each
// in-place `+=` computes in float64 and rounds once to float32.
#include "common.h"

#define STREAM_NOISE_N 21u
#define STREAM_NOISE_U 22u

__global__ __launch_bounds__(256) void action_noise_kernel(float* __restrict__ u, int32_t ldu, int32_t n,
                                                          int32_t dimu, double noise_scale, double random_eps,
                                                          double max_u, const double* __restrict__ randn,
                                                          const double* __restrict__ binom,
                                                          const double* __restrict__ unif, uint64_t seed,
                                                          uint64_t counter) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * dimu) return;
  int row = idx / dimu, col = idx % dimu;
  double z, b, ru;
  if (randn) {
    z = randn[idx];
    b = binom[row];
    ru = unif[idx];
  } else {
    Philox4 r = philox4x32((uint32_t)idx, (uint32_t)counter, (uint32_t)(counter >> 32), STREAM_NOISE_N,
                           (uint32_t)seed, (uint32_t)(seed >> 32));
    double u1 = u01_f64(r.x, r.y), u2 = u01_f64(r.z, r.w);
    z = sqrt(-2.0 * log(1.0 - u1)) * cos(6.283185307179586 * u2);   // Box-Muller
    Philox4 q = philox4x32((uint32_t)row, (uint32_t)counter, (uint32_t)(counter >> 32), STREAM_NOISE_U,
                           (uint32_t)seed, (uint32_t)(seed >> 32));
    b = (u01_f64(q.x, q.y) < random_eps) ? 1.0 : 0.0;
    Philox4 w = philox4x32((uint32_t)idx, (uint32_t)counter, (uint32_t)(counter >> 32), STREAM_NOISE_U + 1u,
                           (uint32_t)seed, (uint32_t)(seed >> 32));
    ru = __dadd_rn(-max_u, __dmul_rn(2.0 * max_u, u01_f64(w.x, w.y)));
  }
  float* p = u + (int64_t)row * ldu + col;
  float v = (float)__dadd_rn((double)(*p), __dmul_rn(noise_scale, z));          // ddpg.py:149-150
  v = fclip(v, (float)-max_u, (float)max_u);                                      // ddpg.py:151
  v = (float)__dadd_rn((double)v, __dmul_rn(b, __dsub_rn(ru, (double)v)));       // ddpg.py:152
  *p = v;
}

extern "C" int curious_action_noise(float* u, int32_t ldu, int32_t n, int32_t dimu, double noise_scale,
                                    double random_eps, double max_u, const double* randn, const double* binom,
                                    const double* unif, uint64_t seed, uint64_t counter, curious_stream_t stream) {
  CURIOUS_CHECK(u, "curious_action_noise: NULL argument");
  bool any = randn || binom || unif, all = randn && binom && unif;
  CURIOUS_CHECK(any == all, "curious_action_noise: give all of randn/binom/unif or none");
  if (n <= 0) return 0;
  int total = n * dimu;
  { ProfScope ps__(CK_NOISE, as_stream(stream)); hipLaunchKernelGGL(action_noise_kernel, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), u, ldu, n, dimu,
                     noise_scale, random_eps, max_u, randn, binom, unif, seed, counter); }
  CURIOUS_LAUNCH_CHECK("action_noise_kernel");
  return 0;
}
