// K12: batched synthetic MultiTaskFetchArm stand-in (reset + step) writing the episode record in place.
//
// Replaces (reference): the per-env Python loop around env.reset/reset_task_goal/env.step and the list
// appends + `change` flag of RolloutWorker rollout.py:107-143,244-303, for the synthetic environment
// specified in oracle/env.py (the reference's MuJoCo envs live in the un-vendored gym_flowers package).
// float32 arithmetic with single-rounding ops and Philox4x32-10 streams: bit-exact with oracle/env.py.
#include "env_body.h"

__global__ void env_reset_kernel(curious_env_cfg_t E, curious_layout_t L, int32_t env_id0,
                                 int32_t* __restrict__ episode, const int32_t* __restrict__ tasks,
                                 const float* __restrict__ goals_raw, int32_t n, float* __restrict__ o,
                                 float* __restrict__ ag, float* __restrict__ g, float* __restrict__ td,
                                 float* __restrict__ staging, float* __restrict__ flags, int64_t* __restrict__ counter,
                                 int64_t delta) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  if (flags && e == 0) flags[n] = 0.0f;                     // the NaN word of the coming rollout (env_body.h)
  if (counter && e == 0) *counter += delta;                 // (curious_env_reset_count)
  const int AG = 3 * E.ntasks;
  float* oe = o + (int64_t)e * E.dimo;
  float* row0 = staging + (int64_t)e * (L.T + 1) * L.row_stride;
  for (int i = 0; i < E.dimo; ++i) oe[i] = 0.0f;
  const int32_t epi = episode[e];
  episode[e] = epi + 1;                                     // episodes started so far (the env's step stream reads it)
  for (int s = 0; s < (AG + 3) / 4; ++s) {
    Philox4 r = philox4x32((uint32_t)(env_id0 + env_of_slot(E, e)), (uint32_t)epi, (uint32_t)s, STREAM_RESET,
                           (uint32_t)E.seed, (uint32_t)(E.seed >> 32));
    uint32_t w[4] = {r.x, r.y, r.z, r.w};
    for (int k = 0; k < 4; ++k) {
      int i = 4 * s + k;
      if (i < AG) {
        float lo = (i < 3) ? -0.1f : -0.6f, wid = (i < 3) ? 0.2f : 1.2f;
        oe[i] = __fadd_rn(lo, __fmul_rn(wid, u01_f32(w[k])));
      }
    }
  }
  const int task = tasks[e];
  for (int i = 0; i < AG; ++i) {
    ag[(int64_t)e * AG + i] = oe[i];
    g[(int64_t)e * AG + i] = 0.0f;
  }
  for (int k = 0; k < 3; ++k) g[(int64_t)e * AG + 3 * task + k] = __fmul_rn(0.5f, goals_raw[e * 3 + k]);
  for (int j = 0; j < E.ntasks; ++j) td[(int64_t)e * E.ntasks + j] = (j == task) ? 1.0f : 0.0f;
  for (int i = 0; i < E.dimo; ++i) row0[L.off_o + i] = oe[i];
  for (int i = 0; i < AG; ++i) row0[L.off_ag + i] = oe[i];
}

extern "C" int curious_env_reset(const curious_env_cfg_t* E, const curious_layout_t* L, int32_t env_id0,
                                 int32_t* episode, const int32_t* tasks, const float* goals_raw, int32_t n,
                                 float* o, float* ag, float* g, float* td, float* staging, float* flags,
                                 curious_stream_t stream) {
  return curious_env_reset_count(E, L, env_id0, episode, tasks, goals_raw, n, o, ag, g, td, staging, flags, nullptr, 0,
                                 stream);
}

extern "C" int curious_env_reset_count(const curious_env_cfg_t* E, const curious_layout_t* L, int32_t env_id0,
                                       int32_t* episode, const int32_t* tasks, const float* goals_raw, int32_t n,
                                       float* o, float* ag, float* g, float* td, float* staging, float* flags,
                                       int64_t* counter, int64_t delta, curious_stream_t stream) {
  CURIOUS_CHECK(E && L && episode && tasks && goals_raw && o && ag && g && td && staging,
                "curious_env_reset: NULL argument");
  CURIOUS_CHECK(E->dimo <= 128, "curious_env_reset: the synthetic env handles observations of at most 128 floats");
  CURIOUS_CHECK(L->dimo == E->dimo && L->dimag == 3 * E->ntasks && L->dimg == 3 * E->ntasks &&
                    L->dimtd == E->ntasks && L->dimu == 4 && E->dimo >= 3 * E->ntasks + 4,
                "curious_env_reset: layout does not match the synthetic env");
  if (n <= 0) return 0;
  { ProfScope ps__(CK_ENV_RESET, as_stream(stream)); hipLaunchKernelGGL(env_reset_kernel, dim3((n + 63) / 64), dim3(64), 0, as_stream(stream), *E, *L, env_id0, episode,
                     tasks, goals_raw, n, o, ag, g, td, staging, flags, counter, delta); }
  CURIOUS_LAUNCH_CHECK("env_reset_kernel");
  return 0;
}

__global__ __launch_bounds__(256) void env_step_kernel(curious_env_cfg_t E, curious_layout_t L, int32_t env_id0,
                                                       const int32_t* __restrict__ episode,
                                                       const int32_t* __restrict__ tasks, const float* __restrict__ u,
                                                       int32_t ldu, int32_t t, int32_t n, float* __restrict__ o,
                                                       float* __restrict__ ag, const float* __restrict__ g,
                                                       const float* __restrict__ td, float* __restrict__ staging,
                                                       int32_t off_change, int32_t off_success, double reward_eps,
                                                       float* __restrict__ flags) {
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (e >= n) return;
  env_step_body(E, L, env_id0, episode, tasks, u + (int64_t)e * ldu, t, o, ag, g, td, staging, off_change,
                off_success, reward_eps, e, lane, flags, n);
}

extern "C" int curious_env_step(const curious_env_cfg_t* E, const curious_layout_t* L, int32_t env_id0,
                                const int32_t* episode, const int32_t* tasks, const float* u, int32_t ldu, int32_t t,
                                int32_t n, float* o, float* ag, const float* g, const float* td, float* staging,
                                int32_t off_change, int32_t off_success, double reward_eps, float* flags,
                                curious_stream_t stream) {
  CURIOUS_CHECK(E && L && episode && tasks && u && o && ag && g && td && staging, "curious_env_step: NULL argument");
  CURIOUS_CHECK(t >= 0 && t < L->T, "curious_env_step: t out of range");
  CURIOUS_CHECK(E->dimo <= 128, "curious_env_step: the synthetic env handles observations of at most 128 floats");
  if (n <= 0) return 0;
  { ProfScope ps__(CK_ENV_STEP, as_stream(stream)); hipLaunchKernelGGL(env_step_kernel, dim3((n + 3) / 4), dim3(256), 0, as_stream(stream), *E, *L, env_id0, episode,
                     tasks, u, ldu, t, n, o, ag, g, td, staging, off_change, off_success, reward_eps, flags); }
  CURIOUS_LAUNCH_CHECK("env_step_kernel");
  return 0;
}

// *p += delta on the device (one thread): the noise-counter base of a captured rollout advances inside the graph.
__global__ void counter_add_kernel(int64_t* p, int64_t delta) { *p += delta; }

extern "C" int curious_counter_add(int64_t* p, int64_t delta, curious_stream_t stream) {
  CURIOUS_CHECK(p, "curious_counter_add: NULL argument");
  { ProfScope ps__(CK_COUNTER_ADD, as_stream(stream));
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, as_stream(stream), p, delta); }
  CURIOUS_LAUNCH_CHECK("counter_add_kernel");
  return 0;
}
