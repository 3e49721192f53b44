// Batched rollout with the actor's hidden matrices RESIDENT in LDS for the whole episode (included by mlp.hip after
// mlp_rows_act.h).
//
// policy_rows_kernel gives 4 envs to one workgroup, which pulls the actor's 590 KB through its CU's L1 at EVERY step of
// the episode: 8.5 us per step at 256 envs, 64 of the 256 CUs busy, the same weights streamed 50 times
// (rollout.py:226-303 for every env; the weights cannot change inside a rollout).  Here a GROUP of 4 workgroups -- 4 CUs
// of one XCD -- serves the 4 envs: member c keeps columns [64 c, 64 c + 64) of every hidden matrix in its LDS (64 KB per
// matrix) from the first step to the last, and per step
//   layer 0      every member computes all 256 columns (57 KB of weights, re-fetched from the L2 while the member waits
//                for x3 of the previous step -- they do not depend on anything)
//   hidden 1     each member its 64 columns out of LDS;  x2: the members all-gather their slices (4 x 64 floats each)
//   hidden 2     each member its 64 columns out of LDS
//   output layer each member the partial sums over its 64 hidden units -- exactly one 16-lane DPP row of the wave-wide
//                reduction of rows_head4;  x3: the members exchange the 4 x 4 partials and add them in that reduction's
//                order
//   noise / clip / eps-greedy / env step: every member does it for all 4 envs (a few hundred flops); member 0 stores.
// With 2 layers per network there is one hidden matrix and no x2.  Same numbers, bit for bit, as policy_rows_kernel: the
// k order of every accumulation and the order of every reduction are kept (tests: the rollout == one launch per step ==
// the unfused path).  tools/rollout_lab.hip (profiles/r03_rollout_lab.txt): 5.3 us per step against 8.5.
//
// Exchange: 64-bit words (tag << 32 | bits) through the L2, relaxed agent-scope atomics, no fences, two buffers per
// member selected by the parity of the exchange's sequence number q: a member publishes q only after it has consumed
// every peer's q - 1, which the peers published after consuming q - 2 -- the slot it overwrites has been read by all.
// Tags = 2 x (Philox noise counter of the launch) + q: unique per launch because (seed, counter) pairs never repeat on
// one agent, so words left in the buffer by an earlier rollout are never mistaken for fresh ones.
// The 4 members of a group spin on each other: ALL workgroups of the launch must be resident at once (one per CU: 157 KB
// of LDS).  The host takes this route only for grids that fit the device's CU count; a member that polls `spins` times
// without an answer gives up, the launch ends, and flags[n] = 2 tells the host (BatchedSyntheticArm.wait_flags raises).
#pragma once

#define RES_NOISE_CH 16          // steps of pre-drawn exploration noise held in LDS at a time

struct ResX {
  unsigned long long* xbuf;      // [groups][2][4 members][4 x 64] tagged words (the head of the acting workspace)
  int32_t xmap;                  // 1: grid % 32 == 0 -> the 4 members of a group share an XCD (block b lands on XCD b % 8)
  int32_t spins;
  int32_t inject;                // > 0 (tests, option "fault_inject"): member 1 of group inject - 1 exits at once -- a
                                 // workgroup that was never scheduled, as far as its peers can tell
  unsigned long long* stamps;    // lab only (option "lab_res_stamps", tools/res_stamps.py): [8] summed cycles of block 0:
                                 // layer 0 | slice 1 | x2 | slice 2 + h2s | output partials | x3 + noise | env step | steps
};
#define RES_STAMP(i)                                                                          \
  do {                                                                                        \
    if (rx.stamps && blockIdx.x == 0 && x.tid == 0) {                                         \
      const unsigned long long n__ = __builtin_readcyclecounter();                            \
      rx.stamps[i] += n__ - last_stamp; last_stamp = n__;                                     \
    }                                                                                         \
  } while (0)

static inline size_t res_lds_floats(int nl) {
  return (size_t)(nl - 1) * 64 * 256 + 4 * RLD + 4 * 4 * 256 + 4 * XLD + 64 + 4 * 64 + (size_t)4 * 3 * 4 * RES_NOISE_CH * 2;
}
static inline size_t res_xbuf_floats(int n) { return (size_t)(n / 4) * 2 * 4 * 256 * 2; }

__device__ __forceinline__ void res_put(unsigned long long* p, uint32_t tag, float v) {
  __hip_atomic_store(p, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
// N words polled together: the loads of a round are all in flight at once (one round trip per round, not N)
template <int N>
__device__ __forceinline__ bool res_take(const unsigned long long* const (&p)[N], uint32_t tag, int max_spins,
                                         float (&out)[N]) {
  unsigned long long w[N];
  int spins = 0;
  bool ok;
  for (;;) {
#pragma unroll
    for (int i = 0; i < N; ++i) w[i] = __hip_atomic_load(p[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ok = true;
#pragma unroll
    for (int i = 0; i < N; ++i) ok = ok && ((uint32_t)(w[i] >> 32) == tag);
    if (ok || ++spins > max_spins) break;
    __builtin_amdgcn_s_sleep(1);
  }
#pragma unroll
  for (int i = 0; i < N; ++i) out[i] = __uint_as_float((uint32_t)(w[i] & 0xffffffffull));
  return ok;
}

// this member's 64 columns of one hidden layer out of LDS: ws[(k >> 2) * 256 + col * 4 + (k & 3)]; wave w takes k in
// [64 w, 64 w + 64) in ascending order (the order of rows_fw_mac).  Returns relu(sum + bias) of (row tid >> 6, column
// tid & 63) -- the partial tiles of the 4 waves meet in x.part, summed like rows_fw_finish.
__device__ __forceinline__ float res_slice(const RCtx& x, const float* ws, const float bias) {
  f32x4 acc = zero4();
#pragma unroll 4
  for (int kq = 0; kq < 16; ++kq) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(ws + (16 * x.wave + kq) * 256 + x.lane * 4);
    const f32x4 a = *reinterpret_cast<const f32x4*>(x.hs + (x.lane & 3) * RLD + 64 * x.wave + 4 * kq);
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = MFMA4(a[s], b[s], acc);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) x.part[(x.wave * 4 + r) * 64 + x.lane] = acc[r];
  __syncthreads();
  const int r = x.tid >> 6, c = x.tid & 63;
  const float s = (x.part[(0 * 4 + r) * 64 + c] + x.part[(1 * 4 + r) * 64 + c]) +
                  (x.part[(2 * 4 + r) * 64 + c] + x.part[(3 * 4 + r) * 64 + c]);
  return fmaxf(s + bias, 0.f);
}

// grid (4 * n / 4): 4 members per group of 4 envs; a.fused, a.nsteps >= 1, a.nl in {2, 3}
__global__ __launch_bounds__(256) void policy_resident_kernel(ActRowsArgs a, ResX rx) {
  extern __shared__ __attribute__((aligned(16))) float rows_lds[];
  const int nh = a.nl - 1;                                   // hidden matrices
  float* w1s = rows_lds;
  float* w2s = w1s + 64 * 256;                               // (nl == 3)
  RCtx x;
  x.hs = rows_lds + (size_t)nh * 64 * 256;
  x.part = x.hs + 4 * RLD;
  x.xin = x.part + 4 * 4 * 256;
  x.sm = x.xin + 4 * XLD;
  float* h2s = x.sm + 64;                                    // [4][64]: this member's slice of the last hidden layer
  double* nzb = reinterpret_cast<double*>(h2s + 4 * 64);
  x.keep = nullptr; x.kb = 0; x.dbg = nullptr;
  x.tid = threadIdx.x; x.wave = x.tid >> 6; x.lane = x.tid & 63;
  int group, member;
  if (rx.xmap) {
    const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    member = slot & 3;
    group = xcd * ((int)gridDim.x >> 5) + (slot >> 2);
  } else {
    group = (int)blockIdx.x >> 2;
    member = (int)blockIdx.x & 3;
  }
  if (rx.inject > 0 && group == rx.inject - 1 && member == 1) return;
  x.r0 = group * ROWS_R;
  const int Sa = a.dimo + a.dimtd, Sc = Sa + 4, G = a.dimg;
  const int m = x.r0 + x.wave;                              // the env this wave finishes (every member: the same 4 envs)
  const float* pp = a.pi.th;
  f32x4 wb[2][16];
  rows_l0_load(wb[0], pp + a.pi.W0, Sa, pp + a.pi.Wg, Sa + G, x.wave, x.lane, 0);
  // output-layer rows of this member's 64 hidden units: lane j < 16 <-> lane 16 member + j of rows_head4
  HeadW4 wpi;
  {
    const int hl = 16 * member + (x.lane & 15);
#pragma unroll
    for (int e = 0; e < 4; ++e) wpi.w[e] = ldv(pp + a.pi.Wout + (int64_t)(4 * hl + e) * 4);
  }
  const float bpi = pp[a.pi.bout + (x.lane & 3)];
  const float b0_pi = pp[a.pi.b0 + x.tid];
  const float bias1 = pp[a.pi.b[1] + 64 * member + (x.tid & 63)];
  const float bias2 = (nh > 1) ? pp[a.pi.b[2] + 64 * member + (x.tid & 63)] : 0.f;
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
        // (relative goals, ddpg.py:119-124: g - ag -- every member holds the whole input row of its 4 envs and runs the
        //  env step itself, so the goal part is re-derived from the new achieved goal on every member, env_step_core)
        v = a.g[r * a.ldg + (k - Sc)];
        if (a.ag) v = __fsub_rn(v, a.ag[r * a.ldag + (k - Sc)]);
        v = fclip(v, -c, c);
        if (a.g_mean) v = fclip(fdiv(__fsub_rn(v, a.g_mean[k - Sc]), a.g_std[k - Sc]), -a.nclip, a.nclip);
      }
      x.xin[i * XLD + k] = v;
    }
  }
  // the resident slices (read once per launch)
  for (int h = 0; h < nh; ++h) {
    const float* W = pp + a.pi.W[h + 1] + 64 * member;
    float* ws = h ? w2s : w1s;
    for (int i = x.tid; i < 256 * 64; i += 256) {
      const int k = i >> 6, c = i & 63;
      ws[(k >> 2) * 256 + c * 4 + (k & 3)] = W[(int64_t)k * 256 + c];
    }
  }
  const uint64_t ctr0 = a.counter + (a.counter_base ? (uint64_t)*a.counter_base : 0ull);
  const EnvConsts ec = env_consts(a.E, a.L, a.episode, a.tasks, a.eo, a.g, a.td, a.staging, m, x.lane);
  float ov = (x.lane < a.E.dimo) ? a.eo[(int64_t)m * a.E.dimo + x.lane] : 0.f;
  double* nz = nzb + (size_t)x.wave * 3 * 4 * RES_NOISE_CH;
  const RowNoise rn = row_noise(a.rg, __builtin_amdgcn_readfirstlane(m), a.seed, a.noise_scale, a.random_eps);
  unsigned long long* xg = rx.xbuf + (size_t)group * 2 * 4 * 256;
  uint32_t q = (uint32_t)(2ull * ctr0) + 1u;                 // tag of the next exchange
  bool lost = false;
  int max_spins = rx.spins;                                  // (after a lost exchange: no more waiting, the rollout is void)
  unsigned long long last_stamp = __builtin_readcyclecounter();
  for (int s = 0; s < a.nsteps; ++s) {
    if ((s % RES_NOISE_CH) == 0) {
      // exploration noise of the next RES_NOISE_CH steps of this wave's env, drawn by all 64 lanes (mlp_rows_act.h)
      const int left = min(RES_NOISE_CH, a.nsteps - s);
      for (int i = x.lane; i < 4 * left; i += 64) {
        const NoiseDraw d = noise_draw(rn.row * 4 + (i & 3), rn.row, rn.random_eps, a.max_u_d, nullptr, nullptr, nullptr,
                                       rn.seed, ctr0 + (uint64_t)(s + (i >> 2)));
        nz[3 * i] = d.z; nz[3 * i + 1] = d.b; nz[3 * i + 2] = d.ru;
      }
    }
    __syncthreads();                                         // input rows (and, first step, the slices) are in LDS
    if (rx.stamps && blockIdx.x == 0 && x.tid == 0) last_stamp = __builtin_readcyclecounter();
    rows_l0_fwd(x, wb, pp + a.pi.W0, Sa, pp + a.pi.Wg, G, Sc, b0_pi, nullptr, -1, nullptr, rnext(RN_NONE, nullptr));
    RES_STAMP(0);
    float v1 = res_slice(x, w1s, bias1);
    RES_STAMP(1);
    if (nh > 1) {
      // x2: all-gather the slices of the first hidden layer into hs
      const int r = x.tid >> 6, c = x.tid & 63;
      res_put(xg + ((q & 1) * 4 + member) * 256 + x.tid, q, v1);
      x.hs[r * RLD + 64 * member + c] = v1;                  // (every wave is past its reads of hs: barrier in res_slice)
      const unsigned long long* ps[3];
#pragma unroll
      for (int p = 1; p < 4; ++p) ps[p - 1] = xg + ((q & 1) * 4 + ((member + p) & 3)) * 256 + x.tid;
      float pv[3];
      if (!res_take<3>(ps, q, max_spins, pv)) { lost = true; max_spins = 0; }
#pragma unroll
      for (int p = 1; p < 4; ++p) x.hs[r * RLD + 64 * ((member + p) & 3) + c] = pv[p - 1];
      ++q;
      __syncthreads();
      RES_STAMP(2);
      v1 = res_slice(x, w2s, bias2);
    }
    h2s[x.tid] = v1;
    __syncthreads();
    RES_STAMP(3);
    // output layer: the partial sums over this member's 64 hidden units = DPP row `member` of rows_head4's reduction
    {
      const f32x4 h4 = *reinterpret_cast<const f32x4*>(h2s + x.wave * 64 + 4 * (x.lane & 15));
      float pd = 0.f;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const float t = row16_sum(h4[0] * wpi.w[0][d] + h4[1] * wpi.w[1][d] + h4[2] * wpi.w[2][d] + h4[3] * wpi.w[3][d]);
        pd = (x.lane == d) ? t : pd;
      }
      if (x.lane < 4) res_put(xg + ((q & 1) * 4 + member) * 256 + 4 * x.wave + x.lane, q, pd);
    }
    RES_STAMP(4);
    // the next step's layer-0 weights: in flight while this member waits for the partials
    if (s + 1 < a.nsteps) rows_l0_load(wb[0], pp + a.pi.W0, Sa, pp + a.pi.Wg, Sa + G, x.wave, x.lane, 0);
    float* s_u = x.sm + 8 * x.wave;
    if (x.lane < 4) {
      const unsigned long long* ps[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) ps[p] = xg + ((q & 1) * 4 + p) * 256 + 4 * x.wave + x.lane;
      float pv[4];
      if (!res_take<4>(ps, q, max_spins, pv)) { lost = true; max_spins = 0; }
      float v = (pv[0] + pv[1]) + (pv[2] + pv[3]);           // wave_sum: (row 0 + row 1) + (row 2 + row 3)
      v = a.max_u * tanhf(v + bpi);                                                            // actor_critic.py:89
      NoiseDraw nd;
      const int i = 4 * (s % RES_NOISE_CH) + x.lane;
      nd.z = nz[3 * i]; nd.b = nz[3 * i + 1]; nd.ru = nz[3 * i + 2];
      v = noise_mix(v, nd, rn.noise_scale, a.max_u_d);                                         // ddpg.py:149-152
      s_u[x.lane] = v;
      if (member == 0) a.u_out[(int64_t)m * a.ldu + x.lane] = v;
    }
    ++q;
    RES_STAMP(5);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float* nin = (s + 1 < a.nsteps) ? x.xin + x.wave * XLD : nullptr;
    if (member == 0)
      ov = env_step_core<true>(a.E, a.L, a.env_id0, ec, s_u, a.t + s, ov, a.eo, a.eag, a.staging, a.off_change,
                               a.off_success, a.reward_eps, m, x.lane, a.flags, a.n, nin, a.clip,
                               InNorm{a.o_mean, a.o_std, a.nclip, a.ag ? Sc : -1, a.g_mean, a.g_std});
    else
      ov = env_step_core<false>(a.E, a.L, a.env_id0, ec, s_u, a.t + s, ov, a.eo, a.eag, a.staging, a.off_change,
                                a.off_success, a.reward_eps, m, x.lane, a.flags, a.n, nin, a.clip,
                                InNorm{a.o_mean, a.o_std, a.nclip, a.ag ? Sc : -1, a.g_mean, a.g_std});
    RES_STAMP(6);
    if (rx.stamps && blockIdx.x == 0 && x.tid == 0) rx.stamps[7] += 1;
  }
  // a member that gave up on a peer: the rollout is void, the host has to know
  if (__any(lost) && x.lane == 0 && a.flags) a.flags[a.n] = 2.0f;
}
