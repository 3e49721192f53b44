// Row-local acting pass (included by mlp.hip after mlp_rows.h): actor forward (+ critic forward, or + exploration noise
// and one step of the GPU-resident env) for 4 envs / batch rows per workgroup in ONE launch.
//
// Replaces, per env step of a batched rollout (rollout.py:226-263 for every env), the 3 launches fwd_l01 -> fwd_hot<DOT>
// -> act_step of the tiled route; and, for DDPG.get_actions (ddpg.py:129-146), fwd_l01 / fwd_hot / head_fwd (+ the same
// again for Q).  Both entry points share this kernel, so the fused acting step stays bit-identical to
// curious_policy_forward + curious_action_noise + curious_env_step.  curious_policy_rollout runs ALL steps of a
// rollout in one launch of it (the envs of a workgroup do not depend on any other env).
#pragma once

struct ActRowsArgs {
  RowsNet pi, q;                       // main (or target) actor; critic only when out_Q
  const float* o; const float* td; const float* g;
  int32_t ldo, ldtd, ldg;
  float clip;                          // clip_obs (ddpg.py:118-127); <= 0: none
  int32_t n, nl, dimo, dimtd, dimg;
  float max_u;
  float* out_pi; int32_t ldpi;         // plain forward: actions without noise
  float* out_Q;                        // optional [n]
  int32_t fused;                       // != 0: noise + clip + eps-greedy + env step (the fields of ActStepArgs below)
  double noise_scale, random_eps, max_u_d;
  uint64_t seed, counter;
  const int64_t* counter_base;
  float* u_out; int32_t ldu;
  curious_env_cfg_t E; curious_layout_t L;
  int32_t env_id0, t, nsteps, off_change, off_success;   // steps t .. t + nsteps - 1 of the episode in this launch
  const int32_t* episode; const int32_t* tasks;
  float* eo; float* eag; float* staging;          // env state (o is also the network input `o`), episode records
  double reward_eps;
  float* flags;                                   // optional rollout flags (env_step_body)
  int32_t noise_lds;                              // != 0: the launch has the LDS area for pre-drawn noise
  // input normalisation (curious_policy_forward, curious_policy_*_stats): mean / std of the normalisers or NULL
  const float *o_mean, *o_std, *g_mean, *g_std;
  float nclip;
  // relative goals (ddpg.py:119-124): g - ag before the clip, or NULL.  Fused steps: ag = the env's achieved goals
  const float* ag; int32_t ldag;
  RankGroups rg;                                  // virtual ranks (noise_body.h); group == 0: one rank
};

// + the pre-drawn exploration noise of a multi-step launch: 4 envs x nsteps x 4 components x (z, coin, uniform) doubles
static inline size_t act_rows_lds_floats(int nsteps = 1) {
  return 4 * RLD + 4 * 4 * 256 + 4 * XLD + 64 + (nsteps > 1 ? (size_t)4 * nsteps * 4 * 3 * 2 : 0);
}

__global__ __launch_bounds__(256) void policy_rows_kernel(ActRowsArgs a) {
  extern __shared__ __attribute__((aligned(16))) float rows_lds[];
  RCtx x;
  x.hs = rows_lds;
  x.part = x.hs + 4 * RLD;
  x.xin = x.part + 4 * 4 * 256;
  x.sm = x.xin + 4 * XLD;
  x.keep = nullptr; x.kb = 0; x.dbg = nullptr;
  x.tid = threadIdx.x; x.wave = x.tid >> 6; x.lane = x.tid & 63;
  x.r0 = blockIdx.x * ROWS_R;
  const int Sa = a.dimo + a.dimtd, Sc = Sa + 4, G = a.dimg;
  const int m = x.r0 + x.wave;                              // the env / row this wave finishes
  const float* pp = a.pi.th;
  f32x4 wb[2][16];
  rows_l0_load(wb[0], pp + a.pi.W0, Sa, pp + a.pi.Wg, Sa + G, x.wave, x.lane, 0);
  const HeadW4 wpi = rows_head4_w(pp + a.pi.Wout, x.lane);
  const float bpi = pp[a.pi.bout + (x.lane & 3)];
  const float b0_pi = pp[a.pi.b0 + x.tid];
  // ---- inputs: xin[i] = [clip(o) | td | action slot | clip(g)]
  {
    const int tot = Sc + G;
    const float c = (a.clip > 0.f) ? a.clip : INFINITY;
    for (int idx = x.tid; idx < 4 * tot; idx += 256) {
      const int i = idx / tot, k = idx - i * tot;
      const int64_t r = x.r0 + i;
      float v;
      if (k < a.dimo) {
        v = fclip(a.o[r * a.ldo + k], -c, c);
        if (a.o_mean) v = fclip(fdiv(__fsub_rn(v, a.o_mean[k]), a.o_std[k]), -a.nclip, a.nclip);     // actor_critic.py:76-83
      } else if (k < Sa) {
        v = a.td[r * a.ldtd + (k - a.dimo)];
      } else if (k < Sc) {
        v = 0.f;
      } else {
        v = a.g[r * a.ldg + (k - Sc)];
        if (a.ag) v = __fsub_rn(v, a.ag[r * a.ldag + (k - Sc)]);
        v = fclip(v, -c, c);
        if (a.g_mean) v = fclip(fdiv(__fsub_rn(v, a.g_mean[k - Sc]), a.g_std[k - Sc]), -a.nclip, a.nclip);
      }
      x.xin[i * XLD + k] = v;
    }
  }
  if (a.fused) {
    // ---- acting steps t .. t + nsteps - 1 of the workgroup's 4 envs (rollout.py:226-303 per env): the envs are
    // independent of each other, so the whole T-step loop of a rollout runs inside one launch -- the new observation
    // goes straight from the env step into the policy's LDS input row, the layer-0 weights of the next step are fetched
    // while the env steps.  Same numbers as nsteps launches with one step each.
    const uint64_t ctr0 = a.counter + (a.counter_base ? (uint64_t)*a.counter_base : 0ull);
    // what does not change during the episode, and the env's observation (lane l: entry l) carried in a register
    const EnvConsts ec = env_consts(a.E, a.L, a.episode, a.tasks, a.eo, a.g, a.td, a.staging, m, x.lane);
    float ov = (x.lane < a.E.dimo) ? a.eo[(int64_t)m * a.E.dimo + x.lane] : 0.f;
    // The exploration noise does not depend on the policy output.  Drawing it inside the step would occupy the wave with
    // 4 active lanes (Philox + float64 Box-Muller) once per step; instead all 64 lanes draw the noise of ALL steps of
    // this wave's env up front -- draw q = 4 s + d: step s, action component d -- into LDS.  (Same (index, counter)
    // pairs, same numbers.)  Too many steps for the LDS area: drawn per step as before.
    double* nz = reinterpret_cast<double*>(x.sm + 64) + (size_t)x.wave * 3 * 4 * a.nsteps;
    const bool predrawn = a.nsteps > 1 && a.noise_lds;
    // (the env of a wave is wave-uniform: the group's key and noise switches live in scalar registers)
    const RowNoise rn = row_noise(a.rg, __builtin_amdgcn_readfirstlane(m), a.seed, a.noise_scale, a.random_eps);
    if (predrawn) {
      for (int q = x.lane; q < 4 * a.nsteps; q += 64) {
        const NoiseDraw d = noise_draw(rn.row * 4 + (q & 3), rn.row, rn.random_eps, a.max_u_d, nullptr, nullptr, nullptr,
                                       rn.seed, ctr0 + (uint64_t)(q >> 2));
        nz[3 * q] = d.z; nz[3 * q + 1] = d.b; nz[3 * q + 2] = d.ru;
      }
    }
    for (int s = 0; s < a.nsteps; ++s) {
      NoiseDraw nd;
      nd.z = nd.b = nd.ru = 0.0;
      if (!predrawn && x.lane < 4)
        nd = noise_draw(rn.row * 4 + x.lane, rn.row, rn.random_eps, a.max_u_d, nullptr, nullptr, nullptr, rn.seed,
                        ctr0 + (uint64_t)s);
      __syncthreads();                                       // input rows of all 4 envs are in LDS
      rows_l0_fwd(x, wb, pp + a.pi.W0, Sa, pp + a.pi.Wg, G, Sc, b0_pi, nullptr, -1, nullptr,
                  rnext(RN_FWD, pp + a.pi.W[1]));
      const RNext again = (s + 1 < a.nsteps) ? rnext(RN_L0, pp + a.pi.W0, Sa, pp + a.pi.Wg, Sa + G)
                                             : rnext(RN_NONE, nullptr);
      for (int l = 1; l < a.nl; ++l)
        rows_big_fwd(x, wb, pp + a.pi.W[l], pp + a.pi.b[l], -1, nullptr,
                     (l + 1 < a.nl) ? rnext(RN_FWD, pp + a.pi.W[l + 1]) : again);
      float z[4];
      rows_head4(x, wpi, z);
      float v = 0.f;
#pragma unroll
      for (int d = 0; d < 4; ++d) v = ((x.lane & 3) == d) ? z[d] : v;
      v = a.max_u * tanhf(v + bpi);                                                          // actor_critic.py:89
      // exploration noise, clip, eps-greedy (ddpg.py:149-152) and one env step, one wavefront per env
      float* s_u = x.sm + 8 * x.wave;
      if (x.lane < 4) {
        if (predrawn) {
          const int q = 4 * s + x.lane;
          nd.z = nz[3 * q]; nd.b = nz[3 * q + 1]; nd.ru = nz[3 * q + 2];
        }
        v = noise_mix(v, nd, rn.noise_scale, a.max_u_d);
        s_u[x.lane] = v;
        a.u_out[(int64_t)m * a.ldu + x.lane] = v;
      }
      // (the network inputs were copied to LDS before layer 0: the env arrays they came from may be overwritten now)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      ov = env_step_core(a.E, a.L, a.env_id0, ec, s_u, a.t + s, ov, a.eo, a.eag, a.staging, a.off_change,
                         a.off_success, a.reward_eps, m, x.lane, a.flags, a.n,
                         (s + 1 < a.nsteps) ? x.xin + x.wave * XLD : nullptr, a.clip,
                         InNorm{a.o_mean, a.o_std, a.nclip, a.ag ? Sc : -1, a.g_mean, a.g_std});
    }
    return;
  }
  __syncthreads();
  rows_l0_fwd(x, wb, pp + a.pi.W0, Sa, pp + a.pi.Wg, G, Sc, b0_pi, nullptr, -1, nullptr,
              rnext(RN_FWD, pp + a.pi.W[1]));
  const float* qp = a.q.th;
  const RNext after = a.out_Q ? rnext(RN_L0, qp + a.q.W0, Sc, qp + a.q.Wg, Sc + G) : rnext(RN_NONE, nullptr);
  for (int l = 1; l < a.nl; ++l)
    rows_big_fwd(x, wb, pp + a.pi.W[l], pp + a.pi.b[l], -1, nullptr,
                 (l + 1 < a.nl) ? rnext(RN_FWD, pp + a.pi.W[l + 1]) : after);
  float z[4];
  rows_head4(x, wpi, z);
  float v = 0.f;
#pragma unroll
  for (int d = 0; d < 4; ++d) v = ((x.lane & 3) == d) ? z[d] : v;
  v = a.max_u * tanhf(v + bpi);                                                            // actor_critic.py:89
  if (x.lane < 4) {
    a.out_pi[(int64_t)m * a.ldpi + x.lane] = v;
    x.xin[x.wave * XLD + Sa + x.lane] = fdiv(v, a.max_u);                                  // actor_critic.py:93
  }
  if (!a.out_Q) return;
  const f32x4 wq = ldv(qp + a.q.Wout + 4 * x.lane);
  const float bq = qp[a.q.bout];
  __syncthreads();
  rows_l0_fwd(x, wb, qp + a.q.W0, Sc, qp + a.q.Wg, G, Sc, qp[a.q.b0 + x.tid], nullptr, -1, nullptr,
              rnext(RN_FWD, qp + a.q.W[1]));
  for (int l = 1; l < a.nl; ++l)
    rows_big_fwd(x, wb, qp + a.q.W[l], qp + a.q.b[l], -1, nullptr,
                 (l + 1 < a.nl) ? rnext(RN_FWD, qp + a.q.W[l + 1]) : rnext(RN_NONE, nullptr));
  const float Q = rows_head1(x, wq) + bq;
  if (x.lane == 0) a.out_Q[m] = Q;
}

// ---- the plain forward (no env step) for SIXTEEN rows per workgroup (mlp_rows16.h): batches of >= FWD16_MIN rows -- the
// evaluator's Q values, one actor + critic forward over the [n x (T + 1)] recorded rows of its rollouts
// (DDPG.rollout_q_sum: 13 056 rows per rollout of 256 envs).  The 4-row form above pulls every layer's 256 KB for 4 rows
// (4 096 rows: 62 us); here the waves split the output columns of 16 rows on v_mfma_f32_16x16x4 and the output layers run
// on the matrix unit too -- the target group of ddpg_rows16_kernel with the main networks and two outputs.  The hidden
// layers sum over k in the order of mlp_rows16.h: the oracle's values at 1e-5, not the bits of the 4-row form.
#define FWD16_MIN 1024
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ROWS16_WAVES, ROWS16_WAVES)))
void policy_fwd16_kernel(ActRowsArgs a) {
  extern __shared__ __attribute__((aligned(16))) float rows_lds[];
  constexpr int R = ROWS_R3;
  RCtx x;
  x.dbg = nullptr;
  x.hs = rows_lds;
  x.hn = x.hs + R * RLD;
  x.part = x.part2 = nullptr;
  x.xin = x.hn + R * RLD;
  x.sm = x.xin + R * XLD;
  x.keep = nullptr; x.kb = x.kb2 = 0;
  x.tid = threadIdx.x; x.wave = x.tid >> 6; x.lane = x.tid & 63;
  x.r0 = blockIdx.x * R;
  const int Sa = a.dimo + a.dimtd, Sc = Sa + 4, G = a.dimg;
  const float* pp = a.pi.th;
  const float* qp = a.q.th;
  f32x4 wb[2][16];
  r16_l0_load8(wb[0], pp + a.pi.W0, Sa, pp + a.pi.Wg, Sa + G, x.wave, x.lane, 0);
  const float bpi = pp[a.pi.bout + (x.lane & 3)];
  {
    const int tot = Sc + G;
    const float c = (a.clip > 0.f) ? a.clip : INFINITY;
    for (int idx = x.tid; idx < R * tot; idx += 256) {
      const int i = idx / tot, k = idx - i * tot;
      const int64_t r = x.r0 + i;
      float v;
      if (k < a.dimo) {
        v = fclip(a.o[r * a.ldo + k], -c, c);
        if (a.o_mean) v = fclip(fdiv(__fsub_rn(v, a.o_mean[k]), a.o_std[k]), -a.nclip, a.nclip);     // actor_critic.py:76-83
      } else if (k < Sa) {
        v = a.td[r * a.ldtd + (k - a.dimo)];
      } else if (k < Sc) {
        v = 0.f;
      } else {
        v = a.g[r * a.ldg + (k - Sc)];
        if (a.ag) v = __fsub_rn(v, a.ag[r * a.ldag + (k - Sc)]);
        v = fclip(v, -c, c);
        if (a.g_mean) v = fclip(fdiv(__fsub_rn(v, a.g_mean[k - Sc]), a.g_std[k - Sc]), -a.nclip, a.nclip);
      }
      x.xin[i * XLD + k] = v;
    }
  }
  __syncthreads();
  r16_l0_fwd(x, wb, pp + a.pi.W0, Sa, pp + a.pi.Wg, G, Sc, pp + a.pi.b0, -1, nullptr, rnext(RN_FWD, pp + a.pi.W[1]));
  const RNext after = a.out_Q ? rnext(RN_L0, qp + a.q.W0, Sc, qp + a.q.Wg, Sc + G) : rnext(RN_NONE, nullptr);
  for (int l = 1; l < a.nl; ++l)
    r16_big_fwd(x, wb, pp + a.pi.W[l], pp + a.pi.b[l], -1, nullptr,
                (l + 1 < a.nl) ? rnext(RN_FWD, pp + a.pi.W[l + 1]) : after);
  const int row = 4 * x.wave + ((x.lane & 15) >> 2);          // lane L < 16 of wave w finishes row 4 w + (L >> 2) (r16_thin_sum)
  {
    float bf[16];
    r16_frag_cols4(bf, pp + a.pi.Wout, x.wave, x.lane);
    const float z = r16_thin_sum(x, r16_thin(x, bf));
    if (x.lane < 16) {
      const float v = a.max_u * tanhf(z + bpi);                                              // actor_critic.py:89
      a.out_pi[(int64_t)(x.r0 + row) * a.ldpi + (x.lane & 3)] = v;
      x.xin[row * XLD + Sa + (x.lane & 3)] = fdiv(v, a.max_u);                               // actor_critic.py:93
    }
  }
  if (!a.out_Q) return;
  const float bq = qp[a.q.bout];
  __syncthreads();
  r16_l0_fwd(x, wb, qp + a.q.W0, Sc, qp + a.q.Wg, G, Sc, qp + a.q.b0, -1, nullptr, rnext(RN_FWD, qp + a.q.W[1]));
  for (int l = 1; l < a.nl; ++l)
    r16_big_fwd(x, wb, qp + a.q.W[l], qp + a.q.b[l], -1, nullptr,
                (l + 1 < a.nl) ? rnext(RN_FWD, qp + a.q.W[l + 1]) : rnext(RN_NONE, nullptr));
  float bf[16];
  r16_frag_rows(bf, qp + a.q.Wout, 1, x.wave, x.lane);
  const float Q = r16_thin_sum(x, r16_thin(x, bf)) + bq;
  if (x.lane < 16 && (x.lane & 3) == 0) a.out_Q[x.r0 + row] = Q;
}

