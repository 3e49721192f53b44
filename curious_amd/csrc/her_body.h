// Device body of the HER gather, shared by her_sample_kernel (her_sample.hip) and the fused Adam + next-sample launch
// (optim.hip).  See her_sample.hip for the description.
#pragma once
#include "common.h"

#define SPB 4                 // samples per block pass (= waves per block)
struct DwStamp { unsigned long long t[4]; };   // lab: cycle stamps a block collects in registers (mlp_lean_gemm.h)
#ifndef DW_STAMP
#ifdef DW_STAMPS
#define DW_STAMP(st, k) do { if (st) (st)->t[k] = __builtin_readcyclecounter(); } while (0)
#else
#define DW_STAMP(st, k) do { } while (0)
#endif
#endif
#define STREAM_SAMPLE_A 11u
#define STREAM_SAMPLE_B 12u

struct HerArgs {
  const float* storage;
  int64_t buf_stride;
  curious_layout_t L;
  curious_tasks_t tasks;
  curious_sample_params_t P;
  curious_sample_plan_t plan;
  curious_sample_rng_t rng;
  int32_t use_rng;
  int32_t n;
  float* batch;
  curious_batch_layout_t BL;
};

#define TAB_INTS (CURIOUS_MAX_TASKS * (1 + 2 * CURIOUS_MAX_TASK_DIMS))   // len | g_id | ag_id, contiguous in the struct

// Dependent global round trips are what this kernel costs (its data volume is ~1 KB per transition), so it is
// organised as exactly two: (1) everything needed to decide WHERE to read -- the task tables (into LDS), the
// sampling tables / host plan and the step counter, all issued together; (2) the three row segments of the
// transition.  Relabelling, reward and clipping then run out of LDS / registers.
// Past the one barrier that publishes the task tables a wave works on its transition alone (round 4): its own LDS slot,
// no further workgroup barrier -- with two __syncthreads() every wave of a block waited twice for the slowest of four random HBM reads
// (tools/dw_stamps.py: 6 k cycles between "rows in LDS" and the exit of a gather block).  LDS operations of one wave
// execute in order, so between a wave's own writes and its reads only the compiler has to be kept from reordering.
// eo / seed_add: batched experts (mlp_common.h "Ex"): the staged batch, the sampling tables and the step counter of
// expert e live eo floats behind expert 0's, its Philox key is seed + seed_add; the replay storage is shared.
// step_add: added to the step counter read from memory (the gather that rides in ddpg_rows_kernel runs BEFORE the
// update's increment of the counter, mlp_rows.h: + 1 gives the key the gather after it would have used).
__device__ __forceinline__ void her_sample_body(const HerArgs& a, const int block, float* lds, const int64_t eo = 0,
                                                const uint64_t seed_add = 0, const int64_t step_add = 0,
                                                DwStamp* stamps = nullptr) {
  const curious_layout_t& L = a.L;
  const curious_batch_layout_t& BL = a.BL;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int head = L.dimo + L.dimag;                    // (o, ag) head of a record row (off_o = 0, off_ag = dimo)
  const int slot_floats = L.row_stride + head + L.dimag + L.dimg + L.dimtd + 4;
  float* slot = lds + wave * slot_floats;
  float* s_row = slot;                                  // record row t
  float* s_next = s_row + L.row_stride;                 // o_2 | ag_2
  float* s_fut = s_next + head;                         // future ag
  float* s_g = s_fut + L.dimag;                         // relabelled goal
  float* s_td = s_g + L.dimg;                           // relabelled task descriptor
  // task tables: ONE copy per workgroup in LDS, fetched with round trip 1 (reads of the kernel's arguments are not
  // cached: read where they are needed they would cost a full round trip each).  The one workgroup barrier that
  // publishes the copy sits in front of round trip 2, where the four waves have all just waited for the same tables.
  __shared__ int32_t s_tab[TAB_INTS];
  const int32_t* s_len = s_tab;
  const int32_t* s_gid = s_tab + CURIOUS_MAX_TASKS;
  const int32_t* s_agid = s_gid + CURIOUS_MAX_TASKS * CURIOUS_MAX_TASK_DIMS;

  // ---- round trip 1
  {
    const int32_t* src = a.tasks.len;                   // len, g_id, ag_id are contiguous after ntasks
    for (int i = threadIdx.x; i < TAB_INTS; i += 256) s_tab[i] = src[i];
  }
  const int gi = block * SPB + wave;
  const bool valid = gi < a.n;
  const int gic = valid ? gi : a.n - 1;
  int buf, ep, t, ttr, out_row;
  double u_her, u_off;
  if (!a.use_rng) {
    buf = a.plan.buf ? a.plan.buf[gic] : 0;
    ep = a.plan.ep[gic];
    t = a.plan.t[gic];
    u_her = a.plan.u_her[gic];
    u_off = a.plan.u_off[gic];
    ttr = a.plan.task_to_replay ? a.plan.task_to_replay[gic] : -1;
    out_row = a.plan.out_row ? a.plan.out_row[gic] : gic;
  } else {
    const curious_sample_rng_t& R = a.rng;
    // virtual ranks (curious_sample_rng_t.rank_rows): this wave's sample is sample gl of rank vr -- that rank's tables,
    // that rank's Philox key, the index it has within the rank's own batch; it lands in row gic of the joint batch
    int gl = gic;
    int64_t to = eo;
    uint64_t seed = R.seed + seed_add;
    if (R.rank_rows > 0) {
      const int vr = gic / R.rank_rows;
      gl = gic - vr * R.rank_rows;
      to += (int64_t)vr * R.rank_tab_stride;
      seed += (uint64_t)vr * R.rank_seed_stride;
    }
    // lane b looks at logical buffer b: all table entries are fetched in one batch
    const int b = min(lane, R.nbuf - 1);
    const int pe = R.prop_prefix[to + b + 1];
    const int al = R.buf_alias ? R.buf_alias[to + b] : b;
    const int tk = R.buf_task ? R.buf_task[to + b] : -1;
    const int cs = R.cur_size[to + b];
    const int64_t step = (R.step_ctr
        ? *reinterpret_cast<const int64_t*>(reinterpret_cast<const float*>(R.step_ctr) + eo) : R.step_host) + step_add;
    const unsigned long long beyond = __ballot(lane < R.nbuf && gl >= pe);
    int lb = __popcll(beyond);
    if (lb >= R.nbuf) lb = R.nbuf - 1;
    buf = __shfl(al, lb);
    ttr = __shfl(tk, lb);
    const uint32_t E = (uint32_t)__shfl(cs, lb);
    Philox4 r1 = philox4x32((uint32_t)gl, (uint32_t)step, (uint32_t)(step >> 32), STREAM_SAMPLE_A,
                            (uint32_t)seed, (uint32_t)(seed >> 32));
    Philox4 r2 = philox4x32((uint32_t)gl, (uint32_t)step, (uint32_t)(step >> 32), STREAM_SAMPLE_B,
                            (uint32_t)seed, (uint32_t)(seed >> 32));
    ep = (int)(((uint64_t)r1.x * E) >> 32);
    t = (int)(((uint64_t)r1.y * (uint32_t)L.T) >> 32);
    u_her = u01_f64(r1.z, r1.w);
    u_off = u01_f64(r2.x, r2.y);
    out_row = gic;
  }
  // her.py:115-118 in float64 / truncation toward zero
  const bool her = __builtin_amdgcn_readfirstlane((int)(u_her < a.P.future_p)) != 0;   // (wave-uniform)
  const int off = (int)(u_off * (double)(L.T - t));
  const int future_t = t + 1 + off;
  const int64_t ep_base = (int64_t)buf * a.buf_stride + (int64_t)ep * (L.T + 1) * L.row_stride;
  DW_STAMP(stamps, 1);
  __syncthreads();                                      // the task tables are in LDS (the only workgroup barrier)

  // ---- round trip 2: row t, the (o, ag) head of row t+1 (replay_buffer.py:47-48) and the future achieved goal
  {
    const float* src = a.storage + ep_base + (int64_t)t * L.row_stride;
    const float* fut = a.storage + ep_base + (int64_t)future_t * L.row_stride + L.off_ag;
    const int n1 = L.row_stride + head;                 // rows t and t+1 are adjacent: one contiguous span
    if (n1 <= 256 && L.dimag <= 64) {
      // every load of the wave is issued before the first LDS write: ONE round trip (a load -> ds_write loop is one
      // round trip per 64 floats, tools/dw_stamps.py: 5.9 k cycles for a 140-float span).  Indices are clamped, not
      // predicated, so that the loads stay unconditional and back to back.
      const int last = n1 - 1;
      const float r0 = src[min(lane, last)];
      const float r1 = src[min(lane + 64, last)];
      const float r2 = src[min(lane + 128, last)];
      const float r3 = src[min(lane + 192, last)];
      const float f0 = fut[min(lane, L.dimag - 1)];
      if (lane < n1) s_row[lane] = r0;
      if (lane + 64 < n1) s_row[lane + 64] = r1;
      if (lane + 128 < n1) s_row[lane + 128] = r2;
      if (lane + 192 < n1) s_row[lane + 192] = r3;
      if (lane < L.dimag) s_fut[lane] = f0;
    } else {
      for (int i = lane; i < n1; i += 64) s_row[i] = src[i];
      for (int i = lane; i < L.dimag; i += 64) s_fut[i] = fut[i];
    }
  }
  __builtin_amdgcn_wave_barrier();
  DW_STAMP(stamps, 2);

  const int mode = a.P.relabel_mode;
  // current task of the sampled transition = position of the 1 in task_descr (her.py:133,159)
  int cur = 0;
  for (int jj = 1; jj < L.dimtd; ++jj)
    if (s_row[L.off_td + jj] > s_row[L.off_td + cur]) cur = jj;
  int rt = cur;
  if (mode == CURIOUS_RELABEL_BUFFER_TASK || mode == CURIOUS_RELABEL_GIVEN_TASK) rt = (ttr >= 0) ? ttr : cur;
  cur = __builtin_amdgcn_readfirstlane(cur);            // (the same on every lane of the wave: one transition per wave)
  rt = __builtin_amdgcn_readfirstlane(rt);
  const int ntasks = a.tasks.ntasks;
  for (int i = lane; i < L.dimg; i += 64) {
    float v = s_row[L.off_g + i];
    if (her) {
      if (mode == CURIOUS_RELABEL_FLAT) {
        int p = 0;
        for (int tt = 0; tt < ntasks; ++tt)
          for (int k = 0; k < s_len[tt]; ++k, ++p)
            if (p == i) v = s_fut[s_agid[tt * CURIOUS_MAX_TASK_DIMS + k]];          // her.py:43-47
      } else {
        if (mode != CURIOUS_RELABEL_CURRENT_TASK) v = 0.0f;                          // her.py:151
        for (int k = 0; k < s_len[rt]; ++k)
          if (s_gid[rt * CURIOUS_MAX_TASK_DIMS + k] == i) v = s_fut[s_agid[rt * CURIOUS_MAX_TASK_DIMS + k]];   // her.py:154 / :164
      }
    }
    s_g[i] = v;
  }
  // task descriptor after relabelling (her.py:152,155); its argmax is the reward's task
  const bool retask = her && (mode == CURIOUS_RELABEL_BUFFER_TASK || mode == CURIOUS_RELABEL_GIVEN_TASK);
  for (int i = lane; i < L.dimtd; i += 64) s_td[i] = retask ? ((i == rt) ? 1.0f : 0.0f) : s_row[L.off_td + i];
  const int rtask = retask ? rt : cur;                  // first maximum of the (one-hot) descriptor
  __builtin_amdgcn_wave_barrier();

  // reward (oracle/reward.py): float64, sequential, no FMA, correctly rounded sqrt; computed redundantly by all lanes
  const float* ag2 = s_next + L.dimo;
  double d2 = 0.0;
  if (a.P.flat_reward) {
    for (int tt = 0; tt < ntasks; ++tt)
      for (int k = 0; k < s_len[tt]; ++k) {
        double d = __dsub_rn((double)ag2[s_agid[tt * CURIOUS_MAX_TASK_DIMS + k]],
                             (double)s_g[s_gid[tt * CURIOUS_MAX_TASK_DIMS + k]]);
        d2 = __dadd_rn(d2, __dmul_rn(d, d));
      }
  } else {
    for (int k = 0; k < s_len[rtask]; ++k) {
      double d = __dsub_rn((double)ag2[s_agid[rtask * CURIOUS_MAX_TASK_DIMS + k]],
                           (double)s_g[s_gid[rtask * CURIOUS_MAX_TASK_DIMS + k]]);
      d2 = __dadd_rn(d2, __dmul_rn(d, d));
    }
  }
  const float reward = (sqrt(d2) > a.P.reward_eps) ? -1.0f : 0.0f;

  if (valid) {
    float* out = a.batch + eo + (int64_t)out_row * BL.stride;
    const float c = a.P.clip_obs;
    const bool rel = a.P.relative_goals != 0;
    const float* ag = s_row + L.off_ag;
    for (int i = lane; i < L.dimo; i += 64) {
      out[BL.off_o + i] = fclip(s_row[L.off_o + i], -c, c);      // ddpg.py:125
      out[BL.off_o2 + i] = fclip(s_next[i], -c, c);
    }
    for (int i = lane; i < L.dimtd; i += 64) out[BL.off_td + i] = s_td[i];
    for (int i = lane; i < L.dimu; i += 64) out[BL.off_u + i] = s_row[L.off_u + i];
    for (int i = lane; i < L.dimg; i += 64) {
      float g = s_g[i];
      float g1 = rel ? fsub(g, ag[i]) : g;                        // ddpg.py:119-124 (dimg == dimag there)
      float g2 = rel ? fsub(g, ag2[i]) : g;
      out[BL.off_g + i] = fclip(g1, -c, c);                       // ddpg.py:126
      out[BL.off_g2 + i] = fclip(g2, -c, c);                      // ddpg.py:353
    }
    for (int i = lane; i < L.dimag; i += 64) {
      out[BL.off_ag + i] = ag[i];
      out[BL.off_ag2 + i] = ag2[i];
    }
    for (int i = lane; i < L.dimextra; i += 64) out[BL.off_extra + i] = s_row[L.off_extra + i];
    if (lane == 0) out[BL.off_r] = reward;
  }
}


// LDS floats one workgroup of the gather needs
static inline size_t her_lds_bytes(const curious_layout_t* L) {
  const int head = L->dimo + L->dimag;
  const int slot_floats = L->row_stride + head + L->dimag + L->dimg + L->dimtd + 4;
  return (size_t)SPB * slot_floats * sizeof(float);
}

// validates the arguments and fills HerArgs; returns 0 on success
static inline int her_fill_args(HerArgs& a, const float* storage, int64_t buf_stride, const curious_layout_t* L,
                                const curious_tasks_t* tasks, const curious_sample_params_t* P,
                                const curious_sample_plan_t* plan, const curious_sample_rng_t* rng, int32_t n,
                                float* batch, const curious_batch_layout_t* BL) {
  CURIOUS_CHECK(storage && L && tasks && P && batch && BL, "curious_her_sample: NULL argument");
  CURIOUS_CHECK((plan != nullptr) != (rng != nullptr), "curious_her_sample: exactly one of plan / rng must be given");
  CURIOUS_CHECK(n >= 0, "curious_her_sample: negative n");
  CURIOUS_CHECK(L->off_o == 0 && L->off_ag == L->dimo, "curious_her_sample: record rows must start with [o | ag]");
  CURIOUS_CHECK(tasks->ntasks <= CURIOUS_MAX_TASKS, "curious_her_sample: too many tasks");
  CURIOUS_CHECK(!rng || rng->nbuf <= 64, "curious_her_sample: at most 64 logical buffers");
  CURIOUS_CHECK(!rng || rng->rank_rows == 0 || (rng->rank_rows > 0 && n % rng->rank_rows == 0 && rng->rank_tab_stride >= 0),
                "curious_her_sample: n must be a whole number of ranks of rank_rows samples");
  CURIOUS_CHECK(!P->relative_goals || L->dimg == L->dimag, "relative_goals needs dimg == dimag (config.py:177-179)");
  memset(&a, 0, sizeof(a));
  a.storage = storage;
  a.buf_stride = buf_stride;
  a.L = *L;
  a.tasks = *tasks;
  a.P = *P;
  if (plan) {
    CURIOUS_CHECK(plan->ep && plan->t && plan->u_her && plan->u_off, "curious_her_sample: incomplete plan");
    a.plan = *plan;
  } else {
    CURIOUS_CHECK(rng->prop_prefix && rng->cur_size && rng->nbuf > 0, "curious_her_sample: incomplete rng plan");
    a.rng = *rng;
    a.use_rng = 1;
  }
  a.n = n;
  a.batch = batch;
  a.BL = *BL;
  CURIOUS_CHECK(her_lds_bytes(L) <= 64 * 1024, "curious_her_sample: record row too large for the LDS slot");
  return 0;
}
