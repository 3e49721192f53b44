// Host side of the network launches, part 2 (included by mlp.hip): the acting entry points -- curious_policy_forward
// (ddpg.py:129-146), the fused act + env-step launch and the whole-rollout launch of the batched env (rollout.py:226-303)
// -- and the choice between the row-local kernels (streaming / weights-resident) and the tiled route.
#pragma once

static bool rows_enabled() { return curious_options().rows != 0; }

static RowsNet rows_net(const float* th, const NetOff& o, int nl) {
  RowsNet n;
  memset(&n, 0, sizeof(n));
  n.th = th; n.W0 = (int32_t)o.W0; n.b0 = (int32_t)o.b0; n.Wg = (int32_t)o.Wg; n.Wout = (int32_t)o.Wout;
  n.bout = (int32_t)o.bout;
  for (int l = 1; l < nl; ++l) { n.W[l] = (int32_t)o.W[l]; n.b[l] = (int32_t)o.b[l]; }
  return n;
}

// with_stats: the caller can hand the normalisers' statistics to the kernel (the plain forward; the fused acting entry
// points carry none)
// relative: goals relative to the achieved goal, which only the plain forward can compute (it is handed ag)
static bool act_rows_ok(const curious_net_cfg_t* c, int n, bool relative, const float* theta, bool with_stats = false) {
  return rows_enabled() && c->modular && c->layers >= 2 && c->layers <= ROWS_MAXL && c->hidden == 256 && c->dimu == 4 &&
         (n % ROWS_R == 0) && (!c->normalize_obs || with_stats) && !relative && c->dimo + c->dimtd + 4 + c->dimg <= ROWS_MAXIN &&
         aligned16(theta);
}

static int device_cu_count() {
  static int cus = -1;
  if (cus < 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
              ? prop.multiProcessorCount : 0;
  }
  return cus;
}

// Multi-step rollouts with the hidden matrices resident in LDS (mlp_rows_res.h): 4 workgroups per 4 envs that spin on
// each other, so every workgroup of the launch must be resident at once -- one per CU (157 KB of LDS each).
// The kernel needs more dynamic LDS than the 64 KB default: the device must have it and the attribute call must succeed
// (checked once per process; a device or partition without 160 KB of LDS per workgroup takes the streaming kernel).
static bool resident_lds_ok() {
  static int ok = -1;
  if (ok < 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    const size_t need = res_lds_floats(3) * sizeof(float);
    ok = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
        (size_t)prop.maxSharedMemoryPerMultiProcessor >= need) {
      ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&policy_resident_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
    }
    (void)hipGetLastError();
  }
  return ok == 1;
}
static bool resident_ok(const ActRowsArgs& a, int n, const float* workspace) {
  return curious_options().resident && a.fused && a.nsteps >= 4 && (a.nl == 2 || a.nl == 3) && n >= 4 &&
         n <= device_cu_count() && workspace != nullptr && resident_lds_ok();
}

static int launch_policy_resident(ActRowsArgs& a, int n, float* workspace, int64_t ws_floats, hipStream_t st) {
  ResX rx;
  rx.xbuf = reinterpret_cast<unsigned long long*>(workspace);
  rx.xmap = (n % 32 == 0) ? 1 : 0;
  rx.spins = curious_options().res_spins;
  rx.inject = curious_options().fault_inject;
  // lab: per-phase cycle stamps of block 0 (8 x 64 bit) behind the exchange buffer, when the workspace has room for them
  rx.stamps = (curious_options().lab_res_stamps && (int64_t)res_xbuf_floats(n) + 16 <= ws_floats)
                  ? reinterpret_cast<unsigned long long*>(workspace + res_xbuf_floats(n)) : nullptr;
  const size_t lds = res_lds_floats(a.nl) * sizeof(float);
  { ProfScope ps__(CK_ACT_RES, st);
    hipLaunchKernelGGL(policy_resident_kernel, dim3(n), dim3(256), lds, st, a, rx); }
  CURIOUS_LAUNCH_CHECK("policy_resident_kernel");
  return 0;
}

static int launch_policy_rows(ActRowsArgs& a, int n, hipStream_t st) {
  const int nsteps = a.fused ? a.nsteps : 1;
  size_t lds = act_rows_lds_floats(nsteps) * sizeof(float);
  a.noise_lds = (nsteps > 1 && lds <= 150 * 1024) ? 1 : 0;
  if (!a.noise_lds) lds = act_rows_lds_floats(1) * sizeof(float);
  static int lds_big = -1;
  if (lds_big < 0) {                                         // > 64 KB of dynamic LDS has to be allowed once per kernel
    lds_big = hipFuncSetAttribute(reinterpret_cast<const void*>(&policy_rows_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
    (void)hipGetLastError();
  }
  if (!lds_big && lds > 64 * 1024) {                         // refused: the per-step noise form fits the default limit
    a.noise_lds = 0;
    lds = act_rows_lds_floats(1) * sizeof(float);
  }
  CURIOUS_CHECK(lds_big || lds <= 64 * 1024, "policy_rows_kernel: the device refused %zu bytes of dynamic LDS", lds);
  if (!a.fused && n >= FWD16_MIN && n % ROWS_R3 == 0 && curious_options().fwd16 && curious_options().rows16 > 0 && a.nl >= 2) {
    // big plain forwards on request (option fwd16: the evaluator's Q pass): 16 rows per workgroup (mlp_rows_act.h
    // policy_fwd16_kernel); without the option every forward keeps the bits of the fused acting kernels
    ProfScope ps__(CK_ACT_ROWS, st);
    hipLaunchKernelGGL(policy_fwd16_kernel, dim3(n / ROWS_R3), dim3(256), rows_lds_floats(ROWS_R3, a.nl) * sizeof(float), st, a);
    CURIOUS_LAUNCH_CHECK("policy_fwd16_kernel");
    return 0;
  }
  { ProfScope ps__(CK_ACT_ROWS, st);
    hipLaunchKernelGGL(policy_rows_kernel, dim3(n / ROWS_R), dim3(256), lds, st, a); }
  CURIOUS_LAUNCH_CHECK("policy_rows_kernel");
  return 0;
}

static void fill_obs_stats(const curious_net_cfg_t* cfg, ObsIn& in, const float* o_stats, const float* g_stats) {
  in.nclip = cfg->norm_clip;
  if (cfg->normalize_obs) {
    in.o_mean = o_stats + 2 * cfg->dimo + 1; in.o_std = o_stats + 3 * cfg->dimo + 1;
    in.g_mean = g_stats + 2 * cfg->dimg + 1; in.g_std = g_stats + 3 * cfg->dimg + 1;
  }
}

extern "C" int curious_policy_forward(const curious_net_cfg_t* cfg, const float* theta, const float* o, int32_t ldo,
                                      const float* ag, int32_t ldag, const float* g, int32_t ldg, const float* td,
                                      int32_t ldtd, int32_t n, float clip_obs, int32_t relative_goals,
                                      const float* o_stats, const float* g_stats, float* workspace, float* out_pi,
                                      float* out_Q, curious_stream_t stream) {
  if (check_cfg(cfg)) return -1;
  CURIOUS_CHECK(theta && o && g && workspace && out_pi, "curious_policy_forward: NULL argument");
  CURIOUS_CHECK(!cfg->modular || cfg->dimtd == 0 || td, "curious_policy_forward: task_descr required");
  CURIOUS_CHECK(!relative_goals || ag, "curious_policy_forward: relative goals need ag");
  CURIOUS_CHECK(!cfg->normalize_obs || (o_stats && g_stats), "curious_policy_forward: normalize_obs needs stats");
  if (n <= 0) return 0;
  hipStream_t st = as_stream(stream);
  Ws w = carve(cfg, n, workspace);
  NetOff offQ = net_off(cfg, true), offPi = net_off(cfg, false);
  const int H = cfg->hidden, nl = cfg->layers;
  const float* thPi = theta + pi_offset(cfg);
  if (act_rows_ok(cfg, n, relative_goals != 0 && !ag, theta, o_stats && g_stats) && aligned16(thPi)) {
    ActRowsArgs a;
    memset(&a, 0, sizeof(a));
    if (cfg->normalize_obs) {
      ObsIn st_in;
      memset(&st_in, 0, sizeof(st_in));
      fill_obs_stats(cfg, st_in, o_stats, g_stats);
      a.o_mean = st_in.o_mean; a.o_std = st_in.o_std; a.g_mean = st_in.g_mean; a.g_std = st_in.g_std;
      a.nclip = st_in.nclip;
    }
    if (relative_goals) { a.ag = ag; a.ldag = ldag; }
    a.pi = rows_net(thPi, offPi, nl); a.q = rows_net(theta, offQ, nl);
    a.o = o; a.td = td; a.g = g; a.ldo = ldo; a.ldtd = ldtd; a.ldg = ldg; a.clip = clip_obs;
    a.n = n; a.nl = nl; a.dimo = cfg->dimo; a.dimtd = cfg->dimtd; a.dimg = cfg->dimg; a.max_u = cfg->max_u;
    a.out_pi = out_pi; a.ldpi = cfg->dimu; a.out_Q = out_Q;
    return launch_policy_rows(a, n, st);
  }
  ObsIn in;
  memset(&in, 0, sizeof(in));
  in.o = o; in.ldo = ldo; in.td = td; in.ldtd = ldtd; in.g = g; in.ldg = ldg; in.ag = ag; in.ldag = ldag;
  in.clip = clip_obs; in.relative = relative_goals;
  fill_obs_stats(cfg, in, o_stats, g_stats);
  Chain a;
  a.theta = thPi; a.off = offPi; a.in = in; a.critic = false; a.act = w.act[2]; a.store_h0 = false;
  if (forward_chains(cfg, &a, 1, n, st)) return -2;
  HeadFwdArgs ha;
  memset(&ha, 0, sizeof(ha));
  ha.nprob = 1;
  ha.p[0] = head_prob(w.act[2][nl - 1], H, thPi + offPi.Wout, thPi + offPi.bout, out_pi, n, cfg->dimu, 2, cfg->max_u);
  if (launch_head_fwd(ha, n, st)) return -2;
  if (out_Q) {
    Chain qc;
    qc.theta = theta; qc.off = offQ; qc.in = in; qc.in.u = out_pi; qc.in.ldu = cfg->dimu; qc.critic = true; qc.store_h0 = false;
    qc.act = w.act[4];
    if (forward_chains(cfg, &qc, 1, n, st)) return -2;
    memset(&ha, 0, sizeof(ha));
    ha.nprob = 1;
    ha.p[0] = head_prob(w.act[4][nl - 1], H, theta + offQ.Wout, theta + offQ.bout, out_Q, n, 1, 0, cfg->max_u);
    if (launch_head_fwd(ha, n, st)) return -2;
  }
  return 0;
}

// steps t .. t + nsteps - 1 of every env: one launch on the row-local route, else nsteps x (forward chain + act_step)
static int policy_act_env_steps(const curious_net_cfg_t* cfg, const float* theta, int32_t n, float clip_obs,
                                float* workspace, double noise_scale, double random_eps, uint64_t seed,
                                uint64_t counter, const int64_t* counter_base, float* u_out, int32_t ldu,
                                const curious_env_cfg_t* E, const curious_layout_t* L, int32_t env_id0,
                                const int32_t* episode, const int32_t* tasks, int32_t t, int32_t nsteps, float* o,
                                float* ag, const float* g, const float* td, float* staging, int32_t off_change,
                                int32_t off_success, double reward_eps, float* flags, curious_stream_t stream,
                                const float* o_stats = nullptr, const float* g_stats = nullptr,
                                int32_t relative_goals = 0, const curious_rank_groups_t* rgp = nullptr) {
  if (check_cfg(cfg)) return -1;
  RankGroups rg;
  memset(&rg, 0, sizeof(rg));
  if (rgp && rgp->group > 0) { rg.group = rgp->group; rg.seed_stride = rgp->seed_stride; rg.exploit = rgp->exploit; }
  CURIOUS_CHECK(theta && workspace && u_out && E && L && episode && tasks && o && ag && g && td && staging,
                "curious_policy_act_env_step: NULL argument");
  CURIOUS_CHECK(cfg->modular, "curious_policy_act_env_step: modular nets only (use curious_policy_forward otherwise)");
  CURIOUS_CHECK(!cfg->normalize_obs || (o_stats && g_stats),
                "curious_policy_act_env_step: input normalisation needs the statistics (curious_policy_*_stats)");
  CURIOUS_CHECK(cfg->dimu == 4 && L->dimu == 4 && cfg->dimo == E->dimo && cfg->dimtd == E->ntasks &&
                    cfg->dimg == 3 * E->ntasks, "curious_policy_act_env_step: network / env dimensions differ");
  CURIOUS_CHECK(t >= 0 && nsteps >= 1 && t + nsteps <= L->T, "curious_policy_act_env_step: t out of range");
  CURIOUS_CHECK(E->dimo <= 128, "curious_policy_act_env_step: the synthetic env handles observations of at most 128 floats");
  if (n <= 0) return 0;
  hipStream_t st = as_stream(stream);
  Ws w = carve(cfg, n, workspace);
  NetOff offPi = net_off(cfg, false);
  const int H = cfg->hidden, nl = cfg->layers;
  const float* thPi = theta + pi_offset(cfg);
  ObsIn st_in;
  memset(&st_in, 0, sizeof(st_in));
  fill_obs_stats(cfg, st_in, o_stats, g_stats);
  if (act_rows_ok(cfg, n, false, theta, true) && aligned16(thPi)) {
    ActRowsArgs a;
    memset(&a, 0, sizeof(a));
    if (relative_goals) { a.ag = ag; a.ldag = 3 * E->ntasks; }   // (dimag == dimg == 3 ntasks in this env)
    a.o_mean = st_in.o_mean; a.o_std = st_in.o_std; a.g_mean = st_in.g_mean; a.g_std = st_in.g_std; a.nclip = st_in.nclip;
    a.pi = rows_net(thPi, offPi, nl);
    a.o = o; a.td = td; a.g = g; a.ldo = E->dimo; a.ldtd = E->ntasks; a.ldg = 3 * E->ntasks; a.clip = clip_obs;
    a.n = n; a.nl = nl; a.dimo = cfg->dimo; a.dimtd = cfg->dimtd; a.dimg = cfg->dimg; a.max_u = cfg->max_u;
    a.fused = 1;
    a.noise_scale = noise_scale; a.random_eps = random_eps; a.max_u_d = (double)cfg->max_u;
    a.seed = seed; a.counter = counter; a.counter_base = counter_base; a.u_out = u_out; a.ldu = ldu;
    a.E = *E; a.L = *L; a.env_id0 = env_id0; a.t = t; a.nsteps = nsteps; a.off_change = off_change;
    a.off_success = off_success;
    a.episode = episode; a.tasks = tasks; a.eo = o; a.eag = ag; a.staging = staging; a.reward_eps = reward_eps;
    a.flags = flags;
    a.rg = rg;
    // the exchange buffer of the resident form is the head of the workspace (the row-local routes use nothing else of it)
    if (resident_ok(a, n, workspace) &&
        (int64_t)res_xbuf_floats(n) <= curious_workspace_floats(cfg, n) && aligned16(workspace))
      return launch_policy_resident(a, n, workspace, curious_workspace_floats(cfg, n), st);
    return launch_policy_rows(a, n, st);
  }
  if (nsteps > 1) {
    for (int s = 0; s < nsteps; ++s) {
      const int rc = policy_act_env_steps(cfg, theta, n, clip_obs, workspace, noise_scale, random_eps, seed, counter + s,
                                          counter_base, u_out, ldu, E, L, env_id0, episode, tasks, t + s, 1, o, ag, g, td,
                                          staging, off_change, off_success, reward_eps, flags, stream, o_stats, g_stats,
                                          relative_goals, rgp);
      if (rc) return rc;
    }
    return 0;
  }
  ObsIn in;
  memset(&in, 0, sizeof(in));
  in.o = o; in.ldo = E->dimo; in.td = td; in.ldtd = E->ntasks; in.g = g; in.ldg = 3 * E->ntasks;
  in.clip = clip_obs;
  if (relative_goals) { in.ag = ag; in.ldag = 3 * E->ntasks; in.relative = 1; }
  fill_obs_stats(cfg, in, o_stats, g_stats);
  Chain a;
  a.theta = thPi; a.off = offPi; a.in = in; a.critic = false; a.act = w.act[2]; a.store_h0 = false;
  // output layer as a dot epilogue of the last hidden layer when that layer runs on the lean kernel
  const bool part = nl >= 3 && H == 256 && hot_ok(n, H, H) && aligned16(thPi) && aligned16(workspace) &&
                    aligned16(thPi + offPi.Wout);
  if (part) { a.dot_mode = 2; a.dot_w = thPi + offPi.Wout; a.dot_out = w.part[1]; }
  if (forward_chains(cfg, &a, 1, n, st)) return -2;
  ActStepArgs k;
  memset(&k, 0, sizeof(k));
  k.part = part ? w.part[1] : nullptr;
  k.a_last = w.act[2][nl - 1]; k.Wout = thPi + offPi.Wout; k.bout = thPi + offPi.bout;
  k.H = H; k.U = cfg->dimu; k.n = n; k.max_u_f = cfg->max_u;
  k.noise_scale = noise_scale; k.random_eps = random_eps; k.max_u = (double)cfg->max_u;
  k.seed = seed; k.counter = counter; k.counter_base = counter_base; k.u_out = u_out; k.ldu = ldu;
  k.E = *E; k.L = *L; k.env_id0 = env_id0; k.t = t; k.off_change = off_change; k.off_success = off_success;
  k.episode = episode; k.tasks = tasks; k.o = o; k.ag = ag; k.g = g; k.td = td; k.staging = staging;
  k.reward_eps = reward_eps;
  k.flags = flags;
  k.rg = rg;
  { ProfScope ps__(CK_ACT_STEP, st);
    if (part) hipLaunchKernelGGL(act_step_kernel<true>, dim3((n + 3) / 4), dim3(256), 0, st, k);
    else hipLaunchKernelGGL(act_step_kernel<false>, dim3((n + 3) / 4), dim3(256), 0, st, k); }
  CURIOUS_LAUNCH_CHECK("act_step_kernel");
  return 0;
}

extern "C" int curious_policy_act_env_step(const curious_net_cfg_t* cfg, const float* theta, int32_t n, float clip_obs,
                                           float* workspace, double noise_scale, double random_eps, uint64_t seed,
                                           uint64_t counter, const int64_t* counter_base, float* u_out, int32_t ldu,
                                           const curious_env_cfg_t* E,
                                           const curious_layout_t* L, int32_t env_id0, const int32_t* episode,
                                           const int32_t* tasks, int32_t t, float* o, float* ag, const float* g,
                                           const float* td, float* staging, int32_t off_change, int32_t off_success,
                                           double reward_eps, float* flags, curious_stream_t stream) {
  return policy_act_env_steps(cfg, theta, n, clip_obs, workspace, noise_scale, random_eps, seed, counter, counter_base,
                              u_out, ldu, E, L, env_id0, episode, tasks, t, 1, o, ag, g, td, staging, off_change,
                              off_success, reward_eps, flags, stream);
}

extern "C" int curious_policy_rollout(const curious_net_cfg_t* cfg, const float* theta, int32_t n, float clip_obs,
                                      float* workspace, double noise_scale, double random_eps, uint64_t seed,
                                      uint64_t counter, const int64_t* counter_base, float* u_out, int32_t ldu,
                                      const curious_env_cfg_t* E, const curious_layout_t* L, int32_t env_id0,
                                      const int32_t* episode, const int32_t* tasks, int32_t t0, int32_t nsteps, float* o,
                                      float* ag, const float* g, const float* td, float* staging, int32_t off_change,
                                      int32_t off_success, double reward_eps, float* flags, curious_stream_t stream) {
  CURIOUS_CHECK(nsteps >= 1, "curious_policy_rollout: nsteps must be positive");
  return policy_act_env_steps(cfg, theta, n, clip_obs, workspace, noise_scale, random_eps, seed, counter, counter_base,
                              u_out, ldu, E, L, env_id0, episode, tasks, t0, nsteps, o, ag, g, td, staging, off_change,
                              off_success, reward_eps, flags, stream);
}

extern "C" int curious_policy_act_env_step_stats(const curious_net_cfg_t* cfg, const float* theta, int32_t n,
                                                 float clip_obs, float* workspace, double noise_scale, double random_eps,
                                                 uint64_t seed, uint64_t counter, const int64_t* counter_base,
                                                 float* u_out, int32_t ldu, const curious_env_cfg_t* E,
                                                 const curious_layout_t* L, int32_t env_id0, const int32_t* episode,
                                                 const int32_t* tasks, int32_t t, float* o, float* ag, const float* g,
                                                 const float* td, float* staging, int32_t off_change,
                                                 int32_t off_success, double reward_eps, float* flags,
                                                 int32_t relative_goals, const float* o_stats, const float* g_stats,
                                                 curious_stream_t stream) {
  return policy_act_env_steps(cfg, theta, n, clip_obs, workspace, noise_scale, random_eps, seed, counter, counter_base,
                              u_out, ldu, E, L, env_id0, episode, tasks, t, 1, o, ag, g, td, staging, off_change,
                              off_success, reward_eps, flags, stream, o_stats, g_stats, relative_goals);
}

extern "C" int curious_policy_rollout_stats(const curious_net_cfg_t* cfg, const float* theta, int32_t n, float clip_obs,
                                            float* workspace, double noise_scale, double random_eps, uint64_t seed,
                                            uint64_t counter, const int64_t* counter_base, float* u_out, int32_t ldu,
                                            const curious_env_cfg_t* E, const curious_layout_t* L, int32_t env_id0,
                                            const int32_t* episode, const int32_t* tasks, int32_t t0, int32_t nsteps,
                                            float* o, float* ag, const float* g, const float* td, float* staging,
                                            int32_t off_change, int32_t off_success, double reward_eps, float* flags,
                                            int32_t relative_goals, const float* o_stats, const float* g_stats,
                                            curious_stream_t stream) {
  CURIOUS_CHECK(nsteps >= 1, "curious_policy_rollout: nsteps must be positive");
  return policy_act_env_steps(cfg, theta, n, clip_obs, workspace, noise_scale, random_eps, seed, counter, counter_base,
                              u_out, ldu, E, L, env_id0, episode, tasks, t0, nsteps, o, ag, g, td, staging, off_change,
                              off_success, reward_eps, flags, stream, o_stats, g_stats, relative_goals);
}

extern "C" int curious_policy_rollout_ranks(const curious_net_cfg_t* cfg, const float* theta, int32_t n, float clip_obs,
                                            float* workspace, double noise_scale, double random_eps, uint64_t seed,
                                            uint64_t counter, const int64_t* counter_base, float* u_out, int32_t ldu,
                                            const curious_env_cfg_t* E, const curious_layout_t* L, int32_t env_id0,
                                            const int32_t* episode, const int32_t* tasks, int32_t t0, int32_t nsteps,
                                            float* o, float* ag, const float* g, const float* td, float* staging,
                                            int32_t off_change, int32_t off_success, double reward_eps, float* flags,
                                            int32_t relative_goals, const float* o_stats, const float* g_stats,
                                            const curious_rank_groups_t* groups, curious_stream_t stream) {
  CURIOUS_CHECK(nsteps >= 1, "curious_policy_rollout: nsteps must be positive");
  CURIOUS_CHECK(!groups || groups->group >= 0, "curious_policy_rollout_ranks: negative group size");
  return policy_act_env_steps(cfg, theta, n, clip_obs, workspace, noise_scale, random_eps, seed, counter, counter_base,
                              u_out, ldu, E, L, env_id0, episode, tasks, t0, nsteps, o, ag, g, td, staging, off_change,
                              off_success, reward_eps, flags, stream, o_stats, g_stats, relative_goals, groups);
}

// What follows the gradients in curious_ddpg_update: Adam (+ the gather of the next batch).
