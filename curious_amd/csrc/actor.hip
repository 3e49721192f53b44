// K10 epilogue: exploration noise, clip and epsilon-greedy mix on the actor output.
//
// Replaces (reference): DDPG.get_actions post-processing ddpg.py:149-152 (+ _random_action ddpg.py:114-115).
// NumPy promotion is reproduced: the float32 action is updated in place with float64 operands, i.e. each
// in-place `+=` computes in float64 and rounds once to float32.
#include "noise_body.h"

__global__ __launch_bounds__(256) void action_noise_kernel(float* __restrict__ u, int32_t ldu, int32_t n,
                                                          int32_t dimu, double noise_scale, double random_eps,
                                                          double max_u, const double* __restrict__ randn,
                                                          const double* __restrict__ binom,
                                                          const double* __restrict__ unif, uint64_t seed,
                                                          uint64_t counter) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * dimu) return;
  int row = idx / dimu, col = idx % dimu;
  float* p = u + (int64_t)row * ldu + col;
  *p = noise_apply(*p, idx, row, noise_scale, random_eps, max_u, randn, binom, unif, seed, counter);
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
