// Device body of one synthetic-env step for one environment handled by one wavefront (shared by env_step_kernel and
// the fused act + env-step kernel).  See env.hip.
#pragma once
#include "common.h"

#define STREAM_RESET 1u
#define STREAM_DISTRACT 2u

// One wavefront per environment; lane l owns observation entry l (and l + 64, ...).  Same arithmetic, operation by
// operation, as the per-env reference loop in oracle/env.py.
__device__ inline void env_step_body(const curious_env_cfg_t& E, const curious_layout_t& L, int32_t env_id0,
                                     const int32_t* __restrict__ episode, const int32_t* __restrict__ tasks,
                                     const float* ue /* the env's 4 action values (global or LDS) */, int32_t t,
                                     float* __restrict__ o, float* __restrict__ ag, const float* __restrict__ g,
                                     const float* __restrict__ td, float* __restrict__ staging, int32_t off_change,
                                     int32_t off_success, double reward_eps, const int e, const int lane,
                                     float* __restrict__ flags = nullptr, const int n = 0,
                                     float* next_in = nullptr /* LDS: receives clip(new o, +-in_clip), the policy's next
                                                                 input row (multi-step rollout kernel) */,
                                     const float in_clip = 0.f) {
  const int AG = 3 * E.ntasks;
  float* oe = o + (int64_t)e * E.dimo;
  float* ep0 = staging + (int64_t)e * (L.T + 1) * L.row_stride;
  float* row = ep0 + (int64_t)t * L.row_stride;
  float* nxt = row + L.row_stride;
  const float* ge = g + (int64_t)e * AG;
  const float* tde = td + (int64_t)e * E.ntasks;
  // every lane needs the gripper, its displacement and the gripper command
  float uc[4], grip[3], ng[3], delta[3];
#pragma unroll
  for (int k = 0; k < 4; ++k) uc[k] = fclip(ue[k], -1.0f, 1.0f);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    grip[k] = oe[k];
    ng[k] = fclip(__fadd_rn(grip[k], __fmul_rn(0.05f, uc[k])), -1.0f, 1.0f);
    delta[k] = __fsub_rn(ng[k], grip[k]);
  }
  const int task = tasks[e];
  const uint32_t ep_ctr = (uint32_t)(episode[e] - 1);
  float nv_first = 0.f;
  for (int i = lane; i < E.dimo; i += 64) {
    float v = oe[i];
    float nv;
    if (i < 3) {
      nv = ng[i];
    } else if (i < AG) {
      const int jt = i / 3, k = i - 3 * jt;
      nv = v;
      if (jt < 4) {
        const float* obj = oe + 3 * jt;                      // old object position (read before anyone writes)
        float d = fmaxf(fmaxf(fabsf(__fsub_rn(grip[0], obj[0])), fabsf(__fsub_rn(grip[1], obj[1]))),
                        fabsf(__fsub_rn(grip[2], obj[2])));
        if (d < 0.1f && uc[3] < 0.0f) nv = fclip(__fadd_rn(v, delta[k]), -1.0f, 1.0f);
      } else {
        Philox4 r = philox4x32((uint32_t)(env_id0 + e), ep_ctr, (uint32_t)(t * E.ntasks + jt), STREAM_DISTRACT,
                               (uint32_t)E.seed, (uint32_t)(E.seed >> 32));
        uint32_t wv = (k == 0) ? r.x : ((k == 1) ? r.y : r.z);
        float st = __fmul_rn(0.01f, __fsub_rn(__fmul_rn(2.0f, u01_f32(wv)), 1.0f));
        nv = fclip(__fadd_rn(v, st), -1.0f, 1.0f);
      }
    } else if (i < AG + 3) {
      nv = delta[i - AG];
    } else if (i == AG + 3) {
      nv = uc[3];
    } else {
      nv = v;
    }
    __builtin_amdgcn_wave_barrier();
    // all lanes have read the old state they need (the loop has one trip for dimo <= 64; for larger dimo the
    // entries >= 64 are beyond AG + 3 and unchanged)
    oe[i] = nv;
    nxt[L.off_o + i] = nv;
    if (next_in) next_in[i] = (in_clip > 0.f) ? fclip(nv, -in_clip, in_clip) : nv;
    if (i < AG) {
      ag[(int64_t)e * AG + i] = nv;
      nxt[L.off_ag + i] = nv;
      row[off_change + i] = (fabsf(__fsub_rn(ep0[L.off_ag + i], nv)) > 1e-3f) ? 1.0f : 0.0f;   // rollout.py:284
      row[L.off_g + i] = ge[i];
    }
    if (i < L.dimu) row[L.off_u + i] = ue[i];
    if (i < E.ntasks) row[L.off_td + i] = tde[i];
    if (i == lane) nv_first = nv;
  }
  // is_success for the env's own task: the new coordinates of the task's slots sit in lanes 3*task .. 3*task+2
  // (AG <= 48 < 64, so they were produced in the first trip)
  double d2 = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    double d = __dsub_rn((double)__shfl(nv_first, 3 * task + k), (double)ge[3 * task + k]);
    d2 = __dadd_rn(d2, __dmul_rn(d, d));
  }
  const float succ = (sqrt(d2) > reward_eps) ? 0.0f : 1.0f;
  if (lane == 0) row[off_success] = succ;
  // rollout flags (rollout.py:268-271,306): flags[e] = is_success of the final step, flags[n] = 1 when an observation
  // of any env ended up NaN.  flags[n] is cleared by env 0 at t = 0 and only set (to the same value) at t = T - 1 --
  // different launches of one stream, no ordering problem.
  if (flags) {
    if (t == 0 && e == 0 && lane == 0) flags[n] = 0.0f;
    if (t == L.T - 1) {
      const bool bad = __any(nv_first != nv_first);
      if (lane == 0) {
        flags[e] = succ;
        if (bad) flags[n] = 1.0f;
      }
    }
  }
}

