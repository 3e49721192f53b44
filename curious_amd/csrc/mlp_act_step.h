// Tiled route, acting: the fused step behind the actor's last hidden layer (included by mlp.hip after mlp_heads.h).
#pragma once

// ------------------------------------------------------------------ fused acting step of the batched rollout
// Actor output layer + max_u*tanh + exploration noise + clip + eps-greedy + one synthetic-env step, one wavefront per
// environment: replaces head_fwd_kernel + action_noise_kernel + env_step_kernel (3 dependent launches -> 1 per env step).
struct ActStepArgs {
  const float* part;        // PART: [4][n][4] column-tile partials of a_last . Wout (dot epilogue of the last layer)
  const float* a_last;      // actor last hidden activation [n, H]
  const float* Wout; const float* bout;
  int32_t H, U, n;
  float max_u_f;
  double noise_scale, random_eps, max_u;
  uint64_t seed, counter;
  const int64_t* counter_base;               // optional device-resident offset of the noise counter (graph replay)
  float* u_out; int32_t ldu;                 // actions as given to the env (also recorded in the episode row)
  curious_env_cfg_t E; curious_layout_t L;
  int32_t env_id0, t, off_change, off_success;
  const int32_t* episode; const int32_t* tasks;
  float* o; float* ag; const float* g; const float* td; float* staging;
  double reward_eps;
  float* flags;                              // optional rollout flags (env_step_body)
  RankGroups rg;                             // virtual ranks (noise_body.h); group == 0: one rank
};

template <bool PART>
__global__ __launch_bounds__(256) void act_step_kernel(ActStepArgs a) {
  __shared__ float s_u[4][MAX_U];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int e = blockIdx.x * 4 + wave;
  if (e >= a.n) return;
  float o_[MAX_U];
#pragma unroll
  for (int d = 0; d < MAX_U; ++d) o_[d] = 0.f;
  if (PART) {
    // dimu == 4: lane d sums the 4 partials of output d
    float pp[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) pp[t] = a.part[((int64_t)t * a.n + e) * 4 + (lane & 3)];
    const float sv = (pp[0] + pp[1]) + (pp[2] + pp[3]);
#pragma unroll
    for (int d = 0; d < 4; ++d) o_[d] = sv;                    // only o_[lane] of lanes 0..3 is used below
  } else {
    const float* hrow = a.a_last + (int64_t)e * a.H;
    const bool al = (((uintptr_t)hrow | (uintptr_t)a.Wout) & 15) == 0;
    if (al && a.U == 4) row_dot_fast<4>(hrow, a.Wout, a.H, lane, o_);
    else row_dot(hrow, a.Wout, a.H, a.U, lane, o_);
  }
  if (lane < a.U) {
    float v = 0.f;
#pragma unroll
    for (int d = 0; d < MAX_U; ++d)
      if (d == lane) v = o_[d];
    v = a.max_u_f * tanhf(v + a.bout[lane]);                  // actor_critic.py:89
    const uint64_t ctr = a.counter + (a.counter_base ? (uint64_t)*a.counter_base : 0ull);
    const RowNoise rn = row_noise(a.rg, e, a.seed, a.noise_scale, a.random_eps);
    v = noise_apply(v, rn.row * a.U + lane, rn.row, rn.noise_scale, rn.random_eps, a.max_u, nullptr, nullptr, nullptr,
                    rn.seed, ctr);                            // ddpg.py:149-152
    s_u[wave][lane] = v;
    a.u_out[(int64_t)e * a.ldu + lane] = v;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  env_step_body(a.E, a.L, a.env_id0, a.episode, a.tasks, s_u[wave], a.t, a.o, a.ag, a.g, a.td, a.staging,
                a.off_change, a.off_success, a.reward_eps, e, lane, a.flags, a.n);
}
