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

// Virtual ranks inside one batched rollout (curious_rank_groups_t): the launch's envs are consecutive groups of `group`
// envs, one group per rank.  Env `row` of the launch is env row % group of group row / group: it draws with that group's
// Philox key at its own row index -- the numbers a process of its own with that key would draw --, and a group whose
// exploit flag is set acts without exploration noise (rollout.py:183-189: every rank decides for itself).
struct RankGroups { int32_t group; int32_t pad_; uint64_t seed_stride; const int32_t* exploit; };
struct RowNoise { uint64_t seed; int row; double noise_scale, random_eps; };
__device__ inline RowNoise row_noise(const RankGroups& g, int row, uint64_t seed, double noise_scale, double random_eps) {
  RowNoise r;
  r.seed = seed; r.row = row; r.noise_scale = noise_scale; r.random_eps = random_eps;
  if (g.group > 0) {
    const int gi = row / g.group;
    r.row = row - gi * g.group;
    r.seed = seed + (uint64_t)gi * g.seed_stride;
    if (g.exploit && g.exploit[gi] != 0) { r.noise_scale = 0.0; r.random_eps = 0.0; }
  }
  return r;
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
