// K1-K3: HER transition sampling as one coalesced, LDS-staged gather kernel.
//
// Replaces (reference): ReplayBuffer.sample replay_buffer.py:37-55, _sample_her_transitions
// her.py:99-183 (multi-task) / her.py:20-66 (flat), the multi-buffer concat+shuffle ddpg.py:326-345 and
// the clip of ddpg.py:350-353.  Index math (HER mask, future offset) is float64/int like NumPy's; goal/task
// relabel and the reward threshold are bit-exact with oracle/her.py + oracle/reward.py.
//
// Mapping: one 256-thread workgroup = 4 waves = 4 sampled transitions per pass; each wave pulls the
// transition's record row t, the (o, ag) head of row t+1 and the future achieved goal into its LDS slot with
// lane-contiguous dword loads (a record row is a few hundred contiguous bytes), relabels in LDS, and
// streams the staged batch row out lane-contiguously.  Algorithmic traffic: SURVEY 8d (1 034 B per
// transition at Arm4 dims).
#include "common.h"

#define SPB 4                 // samples per block pass (= waves per block)
#define STREAM_SAMPLE_A 11u
#define STREAM_SAMPLE_B 12u

struct SampleMeta {
  int64_t row;        // float offset of record row t in storage
  int64_t fut;        // float offset of record row future_t
  int32_t her;        // 1 = HER-relabelled
  int32_t rtask;      // task to replay (<0: keep own / none)
  int32_t out_row;
  int32_t valid;
};

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
  int32_t capacity_rows;   // unused (kept for debugging bounds)
};

__device__ inline void make_meta(const HerArgs& a, int gi, SampleMeta& m) {
  const curious_layout_t& L = a.L;
  int buf, ep, t, ttr, out_row;
  double u_her, u_off;
  if (!a.use_rng) {
    buf = a.plan.buf ? a.plan.buf[gi] : 0;
    ep = a.plan.ep[gi];
    t = a.plan.t[gi];
    u_her = a.plan.u_her[gi];
    u_off = a.plan.u_off[gi];
    ttr = a.plan.task_to_replay ? a.plan.task_to_replay[gi] : -1;
    out_row = a.plan.out_row ? a.plan.out_row[gi] : gi;
  } else {
    const curious_sample_rng_t& R = a.rng;
    int64_t step = R.step_ctr ? *R.step_ctr : R.step_host;
    int lb = 0;
    for (int b = 0; b < R.nbuf; ++b)
      if (gi >= R.prop_prefix[b + 1]) lb = b + 1;
    if (lb >= R.nbuf) lb = R.nbuf - 1;
    buf = R.buf_alias ? R.buf_alias[lb] : lb;
    ttr = R.buf_task ? R.buf_task[lb] : -1;
    uint32_t E = (uint32_t)R.cur_size[buf];
    Philox4 r1 = philox4x32((uint32_t)gi, (uint32_t)step, (uint32_t)(step >> 32), STREAM_SAMPLE_A,
                            (uint32_t)R.seed, (uint32_t)(R.seed >> 32));
    Philox4 r2 = philox4x32((uint32_t)gi, (uint32_t)step, (uint32_t)(step >> 32), STREAM_SAMPLE_B,
                            (uint32_t)R.seed, (uint32_t)(R.seed >> 32));
    ep = (int)(((uint64_t)r1.x * E) >> 32);
    t = (int)(((uint64_t)r1.y * (uint32_t)L.T) >> 32);
    u_her = u01_f64(r1.z, r1.w);
    u_off = u01_f64(r2.x, r2.y);
    out_row = gi;
  }
  // her.py:115-118 in float64 / truncation toward zero
  int her = u_her < a.P.future_p;
  int off = (int)(u_off * (double)(L.T - t));
  int future_t = t + 1 + off;
  int64_t ep_base = (int64_t)buf * a.buf_stride + (int64_t)ep * (L.T + 1) * L.row_stride;
  m.row = ep_base + (int64_t)t * L.row_stride;
  m.fut = ep_base + (int64_t)future_t * L.row_stride;
  m.her = her;
  m.rtask = ttr;
  m.out_row = out_row;
  m.valid = 1;
}

__global__ __launch_bounds__(256) void her_sample_kernel(HerArgs a) {
  extern __shared__ float lds[];
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
  float* s_r = s_td + L.dimtd;                          // reward
  __shared__ SampleMeta meta[SPB];

  const int gi = blockIdx.x * SPB + wave;
  if (lane == 0) {
    if (gi < a.n) make_meta(a, gi, meta[wave]);
    else meta[wave].valid = 0;
  }
  __syncthreads();
  const SampleMeta m = meta[wave];
  if (m.valid) {
    const float* src = a.storage + m.row;
    for (int i = lane; i < L.row_stride; i += 64) s_row[i] = src[i];
    const float* nxt = src + L.row_stride;              // row t+1 (replay_buffer.py:47-48)
    for (int i = lane; i < head; i += 64) s_next[i] = nxt[i];
    const float* fut = a.storage + m.fut + L.off_ag;
    for (int i = lane; i < L.dimag; i += 64) s_fut[i] = fut[i];
  }
  __syncthreads();
  if (m.valid) {
    const curious_tasks_t& T = a.tasks;
    const int mode = a.P.relabel_mode;
    // current task of the sampled transition = position of the 1 in task_descr (her.py:133,159)
    int cur = 0;
    for (int j = 1; j < L.dimtd; ++j)
      if (s_row[L.off_td + j] > s_row[L.off_td + cur]) cur = j;
    int rt = cur;
    if (mode == CURIOUS_RELABEL_BUFFER_TASK || mode == CURIOUS_RELABEL_GIVEN_TASK)
      rt = (m.rtask >= 0) ? m.rtask : cur;
    const bool relabel = m.her != 0;
    for (int i = lane; i < L.dimg; i += 64) {
      float v = s_row[L.off_g + i];
      if (relabel) {
        if (mode == CURIOUS_RELABEL_FLAT) {
          int p = 0;
          for (int t = 0; t < T.ntasks; ++t)
            for (int k = 0; k < T.len[t]; ++k, ++p)
              if (p == i) v = s_fut[T.ag_id[t][k]];      // her.py:43-47
        } else {
          if (mode != CURIOUS_RELABEL_CURRENT_TASK) v = 0.0f;   // her.py:151
          for (int k = 0; k < T.len[rt]; ++k)
            if (T.g_id[rt][k] == i) v = s_fut[T.ag_id[rt][k]];  // her.py:154 / :164
        }
      }
      s_g[i] = v;
    }
    for (int i = lane; i < L.dimtd; i += 64) {
      float v = s_row[L.off_td + i];
      if (relabel && (mode == CURIOUS_RELABEL_BUFFER_TASK || mode == CURIOUS_RELABEL_GIVEN_TASK))
        v = (i == rt) ? 1.0f : 0.0f;                     // her.py:152,155
      s_td[i] = v;
    }
  }
  __syncthreads();
  if (m.valid && lane == 0) {
    // reward (oracle/reward.py): float64, sequential, no FMA, correctly rounded sqrt
    const curious_tasks_t& T = a.tasks;
    const float* ag2 = s_next + L.dimo;
    double d2 = 0.0;
    if (a.P.flat_reward) {
      for (int t = 0; t < T.ntasks; ++t)
        for (int k = 0; k < T.len[t]; ++k) {
          double d = __dsub_rn((double)ag2[T.ag_id[t][k]], (double)s_g[T.g_id[t][k]]);
          d2 = __dadd_rn(d2, __dmul_rn(d, d));
        }
    } else {
      int task = 0;
      for (int j = 1; j < L.dimtd; ++j)
        if (s_td[j] > s_td[task]) task = j;              // np.argmax: first maximum
      for (int k = 0; k < T.len[task]; ++k) {
        double d = __dsub_rn((double)ag2[T.ag_id[task][k]], (double)s_g[T.g_id[task][k]]);
        d2 = __dadd_rn(d2, __dmul_rn(d, d));
      }
    }
    s_r[0] = (sqrt(d2) > a.P.reward_eps) ? -1.0f : 0.0f;
  }
  __syncthreads();
  if (m.valid) {
    float* out = a.batch + (int64_t)m.out_row * BL.stride;
    const float c = a.P.clip_obs;
    const bool rel = a.P.relative_goals != 0;
    const float* ag = s_row + L.off_ag;
    const float* ag2 = s_next + L.dimo;
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
    if (lane == 0) out[BL.off_r] = s_r[0];
  }
}

extern "C" int curious_her_sample(const float* storage, int64_t buf_stride, const curious_layout_t* L,
                                  const curious_tasks_t* tasks, const curious_sample_params_t* P,
                                  const curious_sample_plan_t* plan, const curious_sample_rng_t* rng, int32_t n,
                                  float* batch, const curious_batch_layout_t* BL, curious_stream_t stream) {
  CURIOUS_CHECK(storage && L && tasks && P && batch && BL, "curious_her_sample: NULL argument");
  CURIOUS_CHECK((plan != nullptr) != (rng != nullptr), "curious_her_sample: exactly one of plan / rng must be given");
  CURIOUS_CHECK(n >= 0, "curious_her_sample: negative n");
  CURIOUS_CHECK(L->off_o == 0 && L->off_ag == L->dimo, "curious_her_sample: record rows must start with [o | ag]");
  CURIOUS_CHECK(tasks->ntasks <= CURIOUS_MAX_TASKS, "curious_her_sample: too many tasks");
  CURIOUS_CHECK(!P->relative_goals || L->dimg == L->dimag, "relative_goals needs dimg == dimag (config.py:177-179)");
  if (n == 0) return 0;
  HerArgs a;
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
  const int head = L->dimo + L->dimag;
  const int slot_floats = L->row_stride + head + L->dimag + L->dimg + L->dimtd + 4;
  size_t shmem = (size_t)SPB * slot_floats * sizeof(float);
  CURIOUS_CHECK(shmem <= 64 * 1024, "curious_her_sample: record row too large for the LDS slot");
  int blocks = (n + SPB - 1) / SPB;
  { ProfScope ps__(CK_HER_SAMPLE, as_stream(stream)); hipLaunchKernelGGL(her_sample_kernel, dim3(blocks), dim3(256), shmem, as_stream(stream), a); }
  CURIOUS_LAUNCH_CHECK("her_sample_kernel");
  return 0;
}
