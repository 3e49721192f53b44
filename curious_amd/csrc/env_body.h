// Device body of one synthetic-env step for one environment handled by one wavefront (shared by env_step_kernel and
// the fused act + env-step kernel).  See env.hip.
#pragma once
#include "common.h"

#define STREAM_RESET 1u
#define STREAM_DISTRACT 2u

// One wavefront per environment; lane l owns observation entry l (and l + 64 when dimo > 64: those entries lie beyond
// AG + 3 and never change).  Same arithmetic, operation by operation, as the per-env reference loop in oracle/env.py.
//
// Split in two so that a kernel that walks an env through all steps of an episode (policy_rows_kernel, multi-step)
// fetches what does not change during an episode ONCE and carries the observation in registers:
//   env_consts()    -- task, episode counter, the lane's goal / task-descriptor / initial-achieved-goal entries
//   env_step_core() -- one step from the lane's old observation entry `v` to the new one (returned), with all the
//                      stores of the step (env state, episode record row t and the head of row t + 1, flags)
//   env_step_body() -- consts + the old observation from memory + core: the single-step form
struct EnvConsts {
  int task; uint32_t ep_ctr;
  float ge_i, td_i, ag0_i;            // g[e][lane], td[e][lane], achieved goal of row 0 [lane]  (where lane is in range)
  float goal[3];                      // the goal of the env's own task
  float v_hi;                         // o[e][lane + 64] (constant during the episode) when lane + 64 < dimo
};

// slot -> env (curious_env_cfg_t.wrap: a batch of slots that are the SAME envs at different episodes)
__device__ __forceinline__ int env_of_slot(const curious_env_cfg_t& E, const int e) { return E.wrap > 0 ? e % E.wrap : e; }

__device__ inline EnvConsts env_consts(const curious_env_cfg_t& E, const curious_layout_t& L,
                                       const int32_t* __restrict__ episode, const int32_t* __restrict__ tasks,
                                       const float* __restrict__ o, const float* __restrict__ g,
                                       const float* __restrict__ td, const float* __restrict__ staging, const int e,
                                       const int lane) {
  EnvConsts C;
  const int AG = 3 * E.ntasks;
  const float* ep0 = staging + (int64_t)e * (L.T + 1) * L.row_stride;
  const float* ge = g + (int64_t)e * AG;
  C.task = tasks[e];
  C.ep_ctr = (uint32_t)(episode[e] - 1);
  C.ge_i = (lane < AG) ? ge[lane] : 0.f;
  C.td_i = (lane < E.ntasks) ? td[(int64_t)e * E.ntasks + lane] : 0.f;
  C.ag0_i = (lane < AG) ? ep0[L.off_ag + lane] : 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) C.goal[k] = ge[3 * C.task + k];
  C.v_hi = (lane + 64 < E.dimo) ? o[(int64_t)e * E.dimo + lane + 64] : 0.f;
  return C;
}

// v: o[e][lane] before the step (lanes >= dimo: ignored).  Returns o[e][lane] after the step.
// STORE = false: the step is computed (return value, `next_in`) but nothing is written to global memory -- the members of
// a workgroup group that step the same envs redundantly (mlp_rows_res.h) leave the stores to one of them.
// what the policy's input normalisation does to an observation entry on its way into the next step's input row
// (actor_critic.py:76-83); mean == NULL: nothing
// rel_off >= 0: relative goals (ddpg.py:119-124): the goal part of the input row, at column rel_off, is g - ag of the NEW
// achieved goal (clipped, then normalised with g_mean / g_std when given) -- it changes with every step
struct InNorm {
  const float* mean; const float* stdv; float nclip;
  int rel_off = -1; const float* g_mean = nullptr; const float* g_std = nullptr;
};

template <bool STORE = true>
__device__ inline float env_step_core(const curious_env_cfg_t& E, const curious_layout_t& L, int32_t env_id0,
                                      const EnvConsts& C, const float* ue /* the env's 4 action values (global or LDS) */,
                                      int32_t t, const float v, float* __restrict__ o, float* __restrict__ ag,
                                      float* __restrict__ staging, int32_t off_change, int32_t off_success,
                                      double reward_eps, const int e, const int lane, float* __restrict__ flags,
                                      const int n, float* next_in, const float in_clip,
                                      const InNorm nrm = InNorm{nullptr, nullptr, 0.f}) {
  const int AG = 3 * E.ntasks;
  float* oe = o + (int64_t)e * E.dimo;
  float* ep0 = staging + (int64_t)e * (L.T + 1) * L.row_stride;
  float* row = ep0 + (int64_t)t * L.row_stride;
  float* nxt = row + L.row_stride;
  // every lane needs the gripper, its displacement and the gripper command
  float uc[4], grip[3], ng[3], delta[3];
#pragma unroll
  for (int k = 0; k < 4; ++k) uc[k] = fclip(ue[k], -1.0f, 1.0f);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    grip[k] = __shfl(v, k);
    ng[k] = fclip(__fadd_rn(grip[k], __fmul_rn(0.05f, uc[k])), -1.0f, 1.0f);
    delta[k] = __fsub_rn(ng[k], grip[k]);
  }
  const int i = lane;
  const int jt = i / 3, k = i - 3 * jt;
  // old position of the object this lane's entry belongs to (entries 3 jt .. 3 jt + 2 live in lanes 3 jt ..)
  const int ob = (i < AG) ? 3 * jt : 0;
  const float obj0 = __shfl(v, ob), obj1 = __shfl(v, ob + 1), obj2 = __shfl(v, ob + 2);
  float nv = v;
  if (i < E.dimo) {
    if (i < 3) {
      nv = ng[i];
    } else if (i < AG) {
      if (jt < 4) {
        float d = fmaxf(fmaxf(fabsf(__fsub_rn(grip[0], obj0)), fabsf(__fsub_rn(grip[1], obj1))),
                        fabsf(__fsub_rn(grip[2], obj2)));
        if (d < 0.1f && uc[3] < 0.0f) nv = fclip(__fadd_rn(v, delta[k]), -1.0f, 1.0f);
      } else {
        Philox4 r = philox4x32((uint32_t)(env_id0 + env_of_slot(E, e)), C.ep_ctr, (uint32_t)(t * E.ntasks + jt), STREAM_DISTRACT,
                               (uint32_t)E.seed, (uint32_t)(E.seed >> 32));
        uint32_t wv = (k == 0) ? r.x : ((k == 1) ? r.y : r.z);
        float st = __fmul_rn(0.01f, __fsub_rn(__fmul_rn(2.0f, u01_f32(wv)), 1.0f));
        nv = fclip(__fadd_rn(v, st), -1.0f, 1.0f);
      }
    } else if (i < AG + 3) {
      nv = delta[i - AG];
    } else if (i == AG + 3) {
      nv = uc[3];
    }
    if (next_in) {
      float w = (in_clip > 0.f) ? fclip(nv, -in_clip, in_clip) : nv;
      if (nrm.mean) w = fclip(fdiv(__fsub_rn(w, nrm.mean[i]), nrm.stdv[i]), -nrm.nclip, nrm.nclip);
      next_in[i] = w;
      if (nrm.rel_off >= 0 && i < AG) {                     // (the achieved goal = the first AG observation entries)
        float r = __fsub_rn(C.ge_i, nv);                                                  // ddpg.py:121-123
        if (in_clip > 0.f) r = fclip(r, -in_clip, in_clip);                               // ddpg.py:126
        if (nrm.g_mean) r = fclip(fdiv(__fsub_rn(r, nrm.g_mean[i]), nrm.g_std[i]), -nrm.nclip, nrm.nclip);
        next_in[nrm.rel_off + i] = r;
      }
    }
  }
  if (STORE && i < E.dimo) {
    oe[i] = nv;
    nxt[L.off_o + i] = nv;
    if (i < AG) {
      ag[(int64_t)e * AG + i] = nv;
      nxt[L.off_ag + i] = nv;
      row[off_change + i] = (fabsf(__fsub_rn(C.ag0_i, nv)) > 1e-3f) ? 1.0f : 0.0f;             // rollout.py:284
      row[L.off_g + i] = C.ge_i;
    }
    if (i < L.dimu) row[L.off_u + i] = ue[i];
    if (i < E.ntasks) row[L.off_td + i] = C.td_i;
  }
  if (STORE && i + 64 < E.dimo) nxt[L.off_o + i + 64] = C.v_hi;        // entries beyond AG + 3: unchanged
  // is_success for the env's own task: the new coordinates of the task's slots sit in lanes 3*task .. 3*task+2
  double d2 = 0.0;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    double d = __dsub_rn((double)__shfl(nv, 3 * C.task + q), (double)C.goal[q]);
    d2 = __dadd_rn(d2, __dmul_rn(d, d));
  }
  const float succ = (sqrt(d2) > reward_eps) ? 0.0f : 1.0f;
  if (STORE && lane == 0) row[off_success] = succ;
  // rollout flags (rollout.py:268-271,306): flags[e] = is_success of the final step, flags[n] = 1 when an observation
  // of any env ended up NaN.  flags[n] is cleared by curious_env_reset -- a launch in front of the rollout: inside the
  // multi-step kernel env 0's workgroup may well run after another workgroup has finished all its steps -- and only set
  // (every writer stores the same value) at t = T - 1.
  if (STORE && flags) {
    if (t == L.T - 1) {
      const bool bad = __any((i < E.dimo) && (nv != nv));
      if (lane == 0) {
        flags[e] = succ;
        if (bad) flags[n] = 1.0f;
      }
    }
  }
  return nv;
}

__device__ inline void env_step_body(const curious_env_cfg_t& E, const curious_layout_t& L, int32_t env_id0,
                                     const int32_t* __restrict__ episode, const int32_t* __restrict__ tasks,
                                     const float* ue /* the env's 4 action values (global or LDS) */, int32_t t,
                                     float* __restrict__ o, float* __restrict__ ag, const float* __restrict__ g,
                                     const float* __restrict__ td, float* __restrict__ staging, int32_t off_change,
                                     int32_t off_success, double reward_eps, const int e, const int lane,
                                     float* __restrict__ flags = nullptr, const int n = 0) {
  const EnvConsts C = env_consts(E, L, episode, tasks, o, g, td, staging, e, lane);
  const float v = (lane < E.dimo) ? o[(int64_t)e * E.dimo + lane] : 0.f;
  (void)env_step_core(E, L, env_id0, C, ue, t, v, o, ag, staging, off_change, off_success, reward_eps, e, lane, flags, n,
                      nullptr, 0.f);
}
