// Exploration noise / clip / epsilon-greedy on one action component (shared by action_noise_kernel and the fused
// act + env-step kernel).  See actor.hip.
#pragma once
#include "common.h"

#define STREAM_NOISE_N 21u
#define STREAM_NOISE_U 22u

// The three random numbers of one action component (ddpg.py:149-152): standard normal, the row's eps-greedy coin, the
// uniform replacement action.  They do not depend on the policy output, so a fused kernel can draw them while it waits
// for memory (noise_draw) and mix them in at the end (noise_mix); noise_apply = both.
struct NoiseDraw { double z, b, ru; };

__device__ inline NoiseDraw noise_draw(int idx, int row, double random_eps, double max_u, const double* randn,
                                       const double* binom, const double* unif, uint64_t seed, uint64_t counter) {
  NoiseDraw n;
  if (randn) {
    n.z = randn[idx];
    n.b = binom[row];
    n.ru = unif[idx];
  } else {
    Philox4 r = philox4x32((uint32_t)idx, (uint32_t)counter, (uint32_t)(counter >> 32), STREAM_NOISE_N,
                           (uint32_t)seed, (uint32_t)(seed >> 32));
    double u1 = u01_f64(r.x, r.y), u2 = u01_f64(r.z, r.w);
    n.z = sqrt(-2.0 * log(1.0 - u1)) * cos(6.283185307179586 * u2);   // Box-Muller
    Philox4 q = philox4x32((uint32_t)row, (uint32_t)counter, (uint32_t)(counter >> 32), STREAM_NOISE_U,
                           (uint32_t)seed, (uint32_t)(seed >> 32));
    n.b = (u01_f64(q.x, q.y) < random_eps) ? 1.0 : 0.0;
    Philox4 w = philox4x32((uint32_t)idx, (uint32_t)counter, (uint32_t)(counter >> 32), STREAM_NOISE_U + 1u,
                           (uint32_t)seed, (uint32_t)(seed >> 32));
    n.ru = __dadd_rn(-max_u, __dmul_rn(2.0 * max_u, u01_f64(w.x, w.y)));
  }
  return n;
}

__device__ inline float noise_mix(float pi, const NoiseDraw& n, double noise_scale, double max_u) {
  float v = (float)__dadd_rn((double)pi, __dmul_rn(noise_scale, n.z));             // ddpg.py:149-150
  v = fclip(v, (float)-max_u, (float)max_u);                                      // ddpg.py:151
  v = (float)__dadd_rn((double)v, __dmul_rn(n.b, __dsub_rn(n.ru, (double)v)));   // ddpg.py:152
  return v;
}

__device__ inline float noise_apply(float pi, int idx, int row, double noise_scale, double random_eps, double max_u,
                                    const double* randn, const double* binom, const double* unif, uint64_t seed,
                                    uint64_t counter) {
  return noise_mix(pi, noise_draw(idx, row, random_eps, max_u, randn, binom, unif, seed, counter), noise_scale, max_u);
}
