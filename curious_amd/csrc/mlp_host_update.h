// Host side of the network launches, part 4 (included by mlp.hip): the weight-gradient launch with the optimiser in its tile
// epilogues (+ the HER gather of the next batch), and the entry points curious_ddpg_grads / curious_ddpg_update and their
// batched-experts forms (ddpg.py:235-248, mpi_adam.py:29-35, train.py:65-121).
#pragma once

int DdpgPass::weight_grads(const UpdateTail* tail) {
  // ---- weight/bias gradients: problem lists for the lean kernels (launched after the actor's backward chain)
  const bool dw_hot = dx_hot && (B % 256 == 0) && (nl - 1) <= 4 && (!cfg->normalize_obs || xn_rows);
  LossFin fin;
  fin.rows = w.rows; fin.out = out_losses; fin.B = B; fin.Bl = Bl; fin.U = U; fin.action_l2 = cfg->action_l2;
  fin.step_ctr = gather_in_rows ? step_ctr : nullptr;
  // gradients only (the all-reduce of several ranks follows): the fault word rides along as a padding element
  // (the flag element must be PADDING: with a critic whose parameter count is a multiple of 64 there is none in front of
  //  theta_pi and grad[pi_offset - 1] would be the gradient of the critic's output bias -- then there is no collective
  //  flag, curious_ddpg_transposed reports fault_flag = 0 and the guard stays rank-local)
  const bool has_pad = pi_offset(cfg) > offQ.total;
  fin.fault = (tail || !has_pad) ? nullptr : w.fault;
  fin.flag = (tail || !has_pad) ? nullptr : grad + pi_offset(cfg) - 1;
  // the hidden matrices as 64 x 64 tiles staged through LDS (mlp_dw.h dw_hot_tile64), each reduced by up to 8 workgroups:
  // batches of >= `dw64` rows of a single agent (the split reduction sums in another order than the 16 x 64 tiles)
  const bool t64 = dw_hot && curious_options().dw64 > 0 && B >= curious_options().dw64 && xd.nex == 1 && H == 256 &&
                   w.split64 != nullptr;
  auto build_net = [&](bool critic, DwHotArgs& hw, int& tiles, DwSmallArgs& sm, int& stiles) -> bool {
    int nh = hw.nprob, ns_ = sm.nprob;                              // append to what the other network queued
    bool ok = true;
    const NetOff& off = critic ? offQ : offPi;
    float* g = critic ? gQ : gPi;
    const int chain = critic ? 1 : 2;
    float** dact = w.dact[critic ? 0 : 2];
    auto add_small = [&](const Seg& x, const float* dY, int lddy, int N, float* dW, float* db) {
      if (ns_ >= MAX_DW_SMALL || x.sub || x.mean || x.clip > 0.0f || !(N % 4 == 0 || N == 1) || (N == 1 && x.div != 1.0f) ||
          !(N == 1 || (aligned16(dY) && lddy % 4 == 0 && aligned16(dW)))) { ok = false; return; }
      DwSmall& p = sm.p[ns_++];
      p.x = x.x; p.ldx = x.ld; p.w = x.w; p.div = x.div; p.dY = dY; p.lddy = lddy; p.N = N; p.dW = dW; p.db = db;
      stiles = std::max(stiles, ((x.w + 15) / 16) * ((N + 63) / 64));   // -> slots per problem
    };
    add_small(make_seg(w.act[chain][nl - 1], H, H, nullptr), critic ? w.dQ : w.dz, critic ? 1 : U, off.D,
              g + off.Wout, g + off.bout);
    for (int l = nl - 1; l >= 1; --l) {
      GemmHot& p = hw.p[nh];
      p.A = w.act[chain][l - 1]; p.lda = H; p.B = dact[l]; p.ldb = H; p.C = g + off.W[l]; p.ldc = H;
      p.aux_out = g + off.b[l]; p.M = B; p.N = H; p.K = H;
      p.dot_out = copies_kept ? w.wT[critic ? 0 : 1][l] : nullptr;
      tiles += t64 ? (H / 64) * (H / 64) : (H / 16) * (H / 64);
      ++nh;
    }
    hw.nprob = nh;
    Seg seg[MAX_SEG];
    int ns;
    if (xn_rows) {
      // input normalisation: the row-local launch left the normalised rows [o | td | u / max_u | g] in the workspace
      const float* xr = w.xn[critic ? 0 : 1];
      const int Sa = cfg->dimo + cfg->dimtd;
      ObsIn in;
      memset(&in, 0, sizeof(in));
      in.o = xr; in.td = xr + cfg->dimo; in.u = xr + Sa; in.g = xr + Sa + 4;
      in.ldo = in.ldtd = in.ldu = in.ldg = XLD;
      ns = l0_segments(cfg, off, nullptr, in, critic, 1.0f, seg);
    } else {
      ns = l0_segments(cfg, off, nullptr, cur, critic, cfg->max_u, seg);
    }
    // Segments that are neighbours in memory AND in W0 become one problem: the staged batch row is [o | td | u | g | ..], so
    // the critic's o (40) | td (4) | u (4) are one 48-wide operand -- three strips of 16 rows instead of 3 + 1 + 1, every
    // one of them a tile that costs what a hidden-layer tile costs (Arm4: 76 -> 64 small tiles per launch).  An element's
    // sum over the batch rows does not depend on the tile it sits in: the same bits.
    for (int s = 0; s + 1 < ns - (cfg->modular ? 1 : 0);) {
      Seg& a = seg[s];
      const Seg& b = seg[s + 1];
      if (b.x == a.x + a.w && b.ld == a.ld && b.div == a.div && !a.sub && !b.sub && !a.mean && !b.mean && a.clip == b.clip) {
        a.w += b.w;
        for (int t = s + 1; t + 1 < ns; ++t) seg[t] = seg[t + 1];
        --ns;
      } else {
        ++s;
      }
    }
    int64_t r = 0;
    for (int s = 0; s < ns; ++s) {
      const bool goal_branch = cfg->modular && s == ns - 1;
      float* dW = goal_branch ? g + off.Wg : g + off.W0 + r * H;
      add_small(seg[s], dact[0], H, H, dW, (s == 0) ? g + off.b0 : nullptr);
      if (!goal_branch) r += seg[s].w;
    }
    sm.nprob = ns_; sm.M = B;
    return ok;
  };
  // (Measured: running the critic's gradient kernels on a forked side stream -- a parallel branch of the captured
  //  graph -- made every update 70 % SLOWER on this stack, and slowed unrelated eager launches once a second hardware
  //  queue was active; everything therefore stays on the caller's stream.)
  DwAllArgs dwAll;
  memset(&dwAll, 0, sizeof(dwAll));
  DwHotArgs& hwAll = dwAll.hot;
  DwSmallArgs& smAll = dwAll.small;
  int tAll = 0, stAll = 0;
  bool lean_dw = dw_hot && 2 * (nl - 1) <= 4;
  if (lean_dw) {
    lean_dw = build_net(true, hwAll, tAll, smAll, stAll);
    lean_dw = lean_dw && build_net(false, hwAll, tAll, smAll, stAll);
  }
  CURIOUS_CHECK(xd.nex == 1 || lean_dw, "batched experts need the lean weight-gradient launch");
  if (lean_dw) {
    smAll.fin = fin;
    dwAll.n_hot = tAll;
    hwAll.tiles_per = t64 ? (H / 64) * (H / 64) : (H / 16) * (H / 64);
    smAll.slots = stAll > 0 ? stAll : 1;
    const int nsmall = smAll.nprob * smAll.slots;           // + 1 block for the loss finalisation
    // XCD-aware placement of the launch's blocks (mlp_lean_gemm.h DwMap; option "dw_xcd"): applies when the hidden
    // matrices divide the 8 XCDs evenly (2 or 4 of them: 4 or 2 XCDs each)
    DwMap map;
    memset(&map, 0, sizeof(map));
    // several virtual ranks: the small problems in halves, dealt evenly over the XCDs (mlp_dw.h dw_role; option "dw_bal")
    const bool xcd_ok = curious_options().dw_xcd && (hwAll.nprob == 2 || hwAll.nprob == 4) &&
                        (hwAll.tiles_per == 64 || hwAll.tiles_per == 16);
    // (measured, us per launch with -- in 3 segments -- / without: 19 ranks 41.9 / 47.8, 8 ranks 21.8 / 25.0, 7 ranks 20.7 / 24.9,
    //  6 ranks 18.5 / 22.4, 5 ranks 17.8 / 19.3, 4 ranks 17.1 / 16.3, 3 ranks 15.6 / 14.8: from 5 ranks = 1 280 rows on)
    const bool bal = xcd_ok && curious_options().dw_bal > 0 && B >= curious_options().dw_bal && xd.nex == 1;
    if (bal) {
      auto ntiles = [](const DwSmall& q) { return ((q.w + 15) / 16) * ((q.N + 63) / 64); };
      std::stable_sort(smAll.p, smAll.p + smAll.nprob, [&](const DwSmall& a, const DwSmall& b) { return ntiles(a) > ntiles(b); });
    }
    auto dw_grid = [&](int n_her, int S_hot = 1, int S_small = 1) -> int {
      const int np = hwAll.nprob;
      if (xcd_ok) {
        map.units = 8 / np;
        map.r_hot = hwAll.tiles_per / map.units;
        map.r_her = (n_her + 7) / 8;
        const int r_small = bal ? ((smAll.slots + 1) / 2) * ((2 * smAll.nprob + 1 + 7) / 8)
                                : smAll.slots * ((smAll.nprob + 1 + 7) / 8);      // + 1: the loss finalisation
        const int gx_ = 8 * (map.r_her + map.r_hot * S_hot + r_small * S_small);
        if (B > 256) map.r_her = -map.r_her;                  // several virtual ranks: the gather blocks come last (DwMap)
        return gx_;
      }
      return n_her + tAll * S_hot + (nsmall + 1) * S_small;
    };
    if (curious_options().lab_dw_stamps) dwAll.stamps = reinterpret_cast<unsigned long long*>(w.part[0]);
    // The split reduction (mlp_dw.h DwSplit): by default only the SMALL tiles are split (layer-0 segments and output layers:
    // as many chunks as a hidden tile, and they are what the launch ends with); option "dw_split" = 10 S_hot + S_small
    // overrides (A/B)
    int S_hot = 1, S_small = 1;
    if (w.split && xd.nex == 1 && B % 256 == 0 && tAll + nsmall <= DW_SPLIT_TILES) {
      const int C = B / 256, opt = curious_options().dw_split;
      S_small = std::min(4, C / 2);
      // (dealt in halves, an XCD holds 8 + 2 small tiles: in 3 segments that is 30 workgroups on its 32 CUs, one beside each
      //  hidden tile -- with 4 segments 8 CUs carried two; 19 ranks 45.9 -> 41.9 us per launch, 8 ranks 24.9 -> 21.8)
      if (bal) S_small = 3;
      if (opt > 0) { S_hot = opt / 10; S_small = opt % 10; }
      if (t64) S_hot = DW_SPLIT_MAX;                          // 64 tiles x 8 = 512 workgroups
      S_hot = std::max(1, std::min(S_hot, std::min(DW_SPLIT_MAX, C)));
      S_small = std::max(1, std::min(S_small, std::min(DW_SPLIT_MAX, C)));
    }
    dwAll.split.pbuf = w.split; dwAll.split.cnt = w.split_cnt;
    dwAll.pbuf64 = t64 ? w.split64 : nullptr;
    if (tail && (!tail->her || her_lds_bytes(&tail->h.L) <= sizeof(float) * 4 * 16 * 64)) {
      const int n_her = tail->her ? (tail->h.n + SPB - 1) / SPB : 0;
      const int gx = dw_grid(n_her, S_hot, S_small);
      const int units_s = map.units | (S_hot << 8) | (S_small << 16) | ((bal && map.units) ? 1 << 24 : 0);
      if (dwAll.stamps && (int64_t)gx * 8 * 2 > 6 * 16 * (int64_t)B) dwAll.stamps = nullptr;   // (room: part[0..5])
      { ProfScope ps__(CK_DW_ADAM_HER, st);
        const AdamFuse& af = tail->adam;
        // (never NULL in the kernel: a block loads both words before it knows whether it will need them)
        const int32_t* fault0 = af.fault ? af.fault : reinterpret_cast<const int32_t*>(af.theta);
        const int64_t* ctr0 = af.alpha_tab ? af.step_ctr : reinterpret_cast<const int64_t*>(af.theta);
        // batches of several chunks of 256 rows (virtual ranks): the pipelined form of the tiles (mlp_dw.h PIPE)
#define DW_LAUNCH(...)                                                                                         \
  hipLaunchKernelGGL((dw_adam_her_kernel<__VA_ARGS__>), dim3(gx, xd.nex), dim3(256), 0, st, hwAll.tiles_per,   \
                     hwAll.nprob, smAll.slots, smAll.nprob, n_her, map.r_her, map.r_hot, units_s, fault0, ctr0, \
                     (int64_t)xd.stride, dwAll, tail->adam, tail->h, (int64_t)xd.gstride, seed_stride)
        if (B <= 256) DW_LAUNCH(false);
        else if (dwAll.pbuf64) DW_LAUNCH(true, true);
        else DW_LAUNCH(true);
#undef DW_LAUNCH
      }
      CURIOUS_LAUNCH_CHECK("dw_adam_her_kernel");
      return 0;
    }
    dwAll.stamps = nullptr;
    CURIOUS_CHECK(xd.nex == 1 || !tail, "batched experts need the fused update tail");
    if (!tail && gather_in_dw && !dwAll.pbuf64 && her_lds_bytes(&her_rows.L) <= sizeof(float) * 4 * 16 * 64) {
      // gradients only + the gather of the next batch (dw_adam_her_kernel<true, false, false>, mlp_dw.h)
      CURIOUS_CHECK(!copies_kept, "internal: the transposed copies are not maintained on this route");
      const int n_her = (her_rows.n + SPB - 1) / SPB;
      const int gx = dw_grid(n_her, S_hot, S_small);
      const int units_s = map.units | (S_hot << 8) | (S_small << 16) | ((bal && map.units) ? 1 << 24 : 0);
      AdamFuse none;
      memset(&none, 0, sizeof(none));
      { ProfScope ps__(CK_DW, st);
#define DW_GRADS_HER(PIPE_)                                                                                      \
  hipLaunchKernelGGL((dw_adam_her_kernel<PIPE_, false, false>), dim3(gx, xd.nex), dim3(256), 0, st, hwAll.tiles_per,  \
                     hwAll.nprob, smAll.slots, smAll.nprob, n_her, map.r_her, map.r_hot, units_s,                 \
                     reinterpret_cast<const int32_t*>(theta_main), reinterpret_cast<const int64_t*>(theta_main),  \
                     (int64_t)xd.stride, dwAll, none, her_rows, (int64_t)xd.gstride, seed_stride)
        if (B <= 256) DW_GRADS_HER(false);
        else DW_GRADS_HER(true);
#undef DW_GRADS_HER
      }
      CURIOUS_LAUNCH_CHECK("dw_adam_her_kernel (gradients + gather)");
      gather_done = true;
      return 0;
    }
    CURIOUS_CHECK(!copies_kept, "internal: the transposed copies are not maintained on this route");
    { ProfScope ps__(CK_DW, st);
      const int gx = dw_grid(0, S_hot, S_small);
      const int units_s = map.units | (S_hot << 8) | (S_small << 16) | ((bal && map.units) ? 1 << 24 : 0);
      if (B <= 256)
        hipLaunchKernelGGL(dw_all_kernel<false>, dim3(gx, xd.nex), dim3(256), 0, st, hwAll.tiles_per, hwAll.nprob,
                           smAll.slots, smAll.nprob, 0, map.r_her, map.r_hot, units_s, (int64_t)xd.stride, dwAll,
                           (int64_t)xd.gstride);
      else if (dwAll.pbuf64)
        hipLaunchKernelGGL((dw_all_kernel<true, true>), dim3(gx, xd.nex), dim3(256), 0, st, hwAll.tiles_per, hwAll.nprob,
                           smAll.slots, smAll.nprob, 0, map.r_her, map.r_hot, units_s, (int64_t)xd.stride, dwAll,
                           (int64_t)xd.gstride);
      else
        hipLaunchKernelGGL(dw_all_kernel<true>, dim3(gx, xd.nex), dim3(256), 0, st, hwAll.tiles_per, hwAll.nprob,
                           smAll.slots, smAll.nprob, 0, map.r_her, map.r_hot, units_s, (int64_t)xd.stride, dwAll,
                           (int64_t)xd.gstride); }
    CURIOUS_LAUNCH_CHECK("dw_all_kernel");
  } else {
    CURIOUS_CHECK(!copies_kept, "internal: the transposed copies are not maintained on this route");
    // generic path: every weight/bias gradient + the loss finalisation in one grouped launch
    DwArgs wa;
    memset(&wa, 0, sizeof(wa));
    int np = 0, maxw = 0;
    auto add = [&](const Seg& x, const float* dY, int lddy, int N, float* dW, float* db) {
      DwProb& p = wa.p[np++];
      p.x = x; p.dY = dY; p.lddy = lddy; p.dW = dW; p.db = db; p.M = B; p.N = N;
      p.yvec = (lddy % 4 == 0) && (N % 4 == 0) && aligned16(dY);
      p.fast = p.yvec && N >= 4 && !x.sub;
      if (x.w > maxw) maxw = x.w;
    };
    for (int net = 0; net < 2; ++net) {
      const bool critic = (net == 0);
      const NetOff& off = critic ? offQ : offPi;
      float* g = critic ? gQ : gPi;
      const int chain = critic ? 1 : 2;
      float** dact = w.dact[critic ? 0 : 2];
      add(make_seg(w.act[chain][nl - 1], H, H, nullptr), critic ? w.dQ : w.dz, critic ? 1 : U, off.D, g + off.Wout,
          g + off.bout);
      for (int l = nl - 1; l >= 1; --l)
        add(make_seg(w.act[chain][l - 1], H, H, nullptr), dact[l], H, H, g + off.W[l], g + off.b[l]);
      Seg seg[MAX_SEG];
      int ns = l0_segments(cfg, off, nullptr, cur, critic, cfg->max_u, seg);
      int64_t r = 0;
      for (int s2 = 0; s2 < ns; ++s2) {
        const bool goal_branch = cfg->modular && s2 == ns - 1;
        float* dW = goal_branch ? g + off.Wg : g + off.W0 + r * H;
        add(seg[s2], dact[0], H, H, dW, (s2 == 0) ? g + off.b0 : nullptr);
        if (!goal_branch) r += seg[s2].w;
      }
    }
    CURIOUS_CHECK(np <= MAX_DW, "curious_ddpg_grads: too many gradient problems");
    wa.nprob = np;
    wa.fin = fin;
    dim3 grid((H + 63) / 64, (maxw + 15) / 16, np + 1);
    { ProfScope ps__(CK_DW_SMALL, st); hipLaunchKernelGGL(dw_kernel, grid, dim3(256), 0, st, wa); }
    CURIOUS_LAUNCH_CHECK("dw_kernel");
  }
  if (tail) {
    // the lean gradient launch was not applicable: same result from the stand-alone optimiser (+ gather) launch
    const curious_adam_state_t* a = tail->st;
    const float ah[2] = {a->alpha_Q, a->alpha_pi};
    const int64_t n_Q = pi_offset(cfg);
    curious_transposed_t kp;                                 // no copies kept on this route; the fault word still guards
    memset(&kp, 0, sizeof(kp));
    kp.fault = w.fault;
    if (tail->her) {
      const curious_next_batch_t* nx = tail->next;
      return curious_adam_update_and_sample(const_cast<float*>(theta_main), a->m, a->v, grad, n_Q, tail->n_pi,
                                            a->alpha_tab, step_ctr, a->tab_base, a->tab_len, a->alpha_tab ? nullptr : ah,
                                            a->beta1, a->one_minus_beta1, a->beta2, a->one_minus_beta2, a->epsilon,
                                            nx->storage, nx->buf_stride, nx->L, nx->tasks, nx->P, nx->rng, B, nx->batch,
                                            BL, &kp, (curious_stream_t)st);
    }
    return curious_adam_update(const_cast<float*>(theta_main), a->m, a->v, grad, n_Q, tail->n_pi, a->alpha_tab, step_ctr,
                               a->tab_base, a->tab_len, a->alpha_tab ? nullptr : ah, a->beta1, a->one_minus_beta1,
                               a->beta2, a->one_minus_beta2, a->epsilon, &kp, (curious_stream_t)st);
  }
  return 0;
}

// next (without a tail only): the device-drawn HER gather of the NEXT update's batch as part of this call -- inside the
// row-local launch where that route applies, as a launch of its own behind the gradients otherwise.
static int ddpg_grads_impl(const curious_net_cfg_t* cfg, const float* theta_main, const float* theta_target,
                           const float* batch, const curious_batch_layout_t* BL, int32_t B, const float* o_stats,
                           const float* g_stats, float* workspace, float* grad, float* out_losses, float* out_Q_pi,
                           int64_t* step_ctr, curious_stream_t stream, const UpdateTail* tail,
                           const ExDim& xd = ExDim(), uint64_t seed_stride = 0, bool params_unchanged = false,
                           const curious_next_batch_t* next = nullptr) {
  DdpgPass p;
  p.xd = xd; p.seed_stride = seed_stride;
  p.cfg = cfg; p.theta_main = theta_main; p.theta_target = theta_target; p.batch = batch; p.BL = BL; p.B = B;
  p.o_stats = o_stats; p.g_stats = g_stats; p.workspace = workspace; p.grad = grad; p.out_losses = out_losses;
  p.out_Q_pi = out_Q_pi; p.step_ctr = step_ctr;
  int rc = p.setup(stream);
  if (!rc && next) {
    CURIOUS_CHECK(!tail, "internal: a fused update carries its own gather");
    CURIOUS_CHECK(next->batch && next->batch != batch, "curious_ddpg_grads: the next batch needs its own staging buffer");
    CURIOUS_CHECK(next->rng && next->rng->step_ctr == step_ctr && step_ctr,
                  "curious_ddpg_grads: the next batch must be keyed by this call's step counter");
    if (her_fill_args(p.her_rows, next->storage, next->buf_stride, next->L, next->tasks, next->P, nullptr, next->rng, B,
                      next->batch, BL)) return -1;
    p.gather_in_rows = p.rows_route() && her_lds_bytes(next->L) <= rows_lds_floats(ROWS_R, cfg->layers) * sizeof(float) &&
                       her_lds_bytes(next->L) <= rows_lds_floats(ROWS_R3, cfg->layers) * sizeof(float) &&
                       (B % (ROWS_R * 4) == 0) && SPB == ROWS_R;
    // batches of the 16-row form (several virtual ranks): the gather rides in the weight-gradient launch (mlp_dw.h)
    if (p.gather_in_rows && xd.nex == 1 && curious_options().gather_dw > 0 && B >= curious_options().gather_dw && B % 256 == 0) {
      p.gather_in_rows = false;
      p.gather_in_dw = true;
    }
  }
  if (!rc && p.rows_route()) {
    // the copies are kept current by this pass's own optimiser tail (maintained), or -- without a tail -- by the
    // caller's stand-alone optimiser call (curious_adam_update* with `keep`), as the caller asserts
    const bool maintained = p.keeps_copies(tail);
    rc = p.rows_pass(!((maintained || !tail) && params_unchanged), maintained);
  } else {
    if (!rc) rc = p.forward();
    if (!rc) rc = p.critic_backward();
    if (!rc) rc = p.actor_backward();
  }
  if (!rc) rc = p.weight_grads(tail);
  if (!rc && next && !p.gather_in_rows && !p.gather_done) {
    if (xd.nex > 1) { curious_set_error("batched experts need the lean route (row-local kernels) for the gather of the next batch"); return -1; }
    rc = curious_her_sample(next->storage, next->buf_stride, next->L, next->tasks, next->P, nullptr, next->rng, B,
                            next->batch, BL, stream);
  }
  return rc;
}

extern "C" int curious_ddpg_grads(const curious_net_cfg_t* cfg, const float* theta_main, const float* theta_target,
                                  const float* batch, const curious_batch_layout_t* BL, int32_t B,
                                  const float* o_stats, const float* g_stats, float* workspace, float* grad,
                                  float* out_losses, float* out_Q_pi, int64_t* step_ctr, int32_t params_unchanged,
                                  const curious_next_batch_t* next, curious_stream_t stream) {
  return ddpg_grads_impl(cfg, theta_main, theta_target, batch, BL, B, o_stats, g_stats, workspace, grad, out_losses,
                         out_Q_pi, step_ctr, stream, nullptr, ExDim(), 0, params_unchanged != 0, next);
}

static int ddpg_update_impl(const curious_net_cfg_t* cfg, float* theta_main, const float* theta_target,
                            const float* batch, const curious_batch_layout_t* BL, int32_t B,
                            const float* o_stats, const float* g_stats, float* workspace, float* grad,
                            float* out_losses, float* out_Q_pi, int64_t* step_ctr,
                            const curious_adam_state_t* adam, const curious_next_batch_t* next,
                            curious_stream_t stream, const ExDim& xd, uint64_t seed_stride) {
  if (check_cfg(cfg)) return -1;
  CURIOUS_CHECK(adam && adam->m && adam->v, "curious_ddpg_update: NULL optimiser state");
  CURIOUS_CHECK(!adam->alpha_tab || (step_ctr && adam->tab_len > 0), "curious_ddpg_update: step-size table needs step_ctr");
  UpdateTail t;
  memset(&t, 0, sizeof(t));
  t.st = adam; t.next = next;
  t.n_pi = curious_param_total(cfg) - pi_offset(cfg);
  AdamFuse& A = t.adam;
  A.theta = theta_main; A.m = adam->m; A.v = adam->v; A.grad = grad; A.n_Q = pi_offset(cfg);
  A.alpha_tab = adam->alpha_tab; A.step_ctr = step_ctr; A.tab_base = adam->tab_base; A.tab_len = adam->tab_len;
  A.a_Q = adam->alpha_Q; A.a_pi = adam->alpha_pi;
  A.b1 = adam->beta1; A.omb1 = adam->one_minus_beta1; A.b2 = adam->beta2; A.omb2 = adam->one_minus_beta2;
  A.eps = adam->epsilon;
  A.fault = carve(cfg, B, workspace).fault;
  if (next) {
    CURIOUS_CHECK(next->batch && next->batch != batch, "curious_ddpg_update: the next batch needs its own staging buffer");
    if (her_fill_args(t.h, next->storage, next->buf_stride, next->L, next->tasks, next->P, nullptr, next->rng, B,
                      next->batch, BL)) return -1;
    t.her = true;
  }
  return ddpg_grads_impl(cfg, theta_main, theta_target, batch, BL, B, o_stats, g_stats, workspace, grad, out_losses,
                         out_Q_pi, step_ctr, stream, &t, xd, seed_stride, adam->params_unchanged != 0);
}

extern "C" int curious_ddpg_update(const curious_net_cfg_t* cfg, float* theta_main, const float* theta_target,
                                   const float* batch, const curious_batch_layout_t* BL, int32_t B,
                                   const float* o_stats, const float* g_stats, float* workspace, float* grad,
                                   float* out_losses, float* out_Q_pi, int64_t* step_ctr,
                                   const curious_adam_state_t* adam, const curious_next_batch_t* next,
                                   curious_stream_t stream) {
  return ddpg_update_impl(cfg, theta_main, theta_target, batch, BL, B, o_stats, g_stats, workspace, grad, out_losses,
                          out_Q_pi, step_ctr, adam, next, stream, ExDim(), 0);
}

static int check_experts(const curious_net_cfg_t* cfg, int32_t n_experts, int64_t expert_stride, int64_t grad_stride) {
  CURIOUS_CHECK(n_experts >= 1 && n_experts <= 64, "batched experts: n_experts must be in 1..64");
  CURIOUS_CHECK(n_experts == 1 || (expert_stride > 0 && expert_stride % 64 == 0),
                "batched experts: expert_stride must be a positive multiple of 64 floats");
  CURIOUS_CHECK(cfg && (n_experts == 1 || (grad_stride >= curious_param_total(cfg) && grad_stride % 64 == 0)),
                "batched experts: grad_stride must be a multiple of 64 floats >= curious_param_total()");
  return 0;
}

extern "C" int curious_ddpg_grads_experts(const curious_net_cfg_t* cfg, int32_t n_experts, int64_t expert_stride,
                                          int64_t grad_stride, const float* theta_main, const float* theta_target,
                                          const float* batch, const curious_batch_layout_t* BL, int32_t B,
                                          const float* o_stats, const float* g_stats, float* workspace, float* grad,
                                          float* out_losses, float* out_Q_pi, int64_t* step_ctr,
                                          int32_t params_unchanged, uint64_t seed_stride,
                                          const curious_next_batch_t* next, curious_stream_t stream) {
  if (check_experts(cfg, n_experts, expert_stride, grad_stride)) return -1;
  CURIOUS_CHECK(!cfg->normalize_obs || (o_stats && g_stats),
                "curious_ddpg_grads_experts: input normalisation needs the experts' statistics");
  ExDim xd;
  xd.nex = n_experts; xd.stride = expert_stride; xd.gstride = grad_stride;
  return ddpg_grads_impl(cfg, theta_main, theta_target, batch, BL, B, o_stats, g_stats, workspace, grad, out_losses,
                         out_Q_pi, step_ctr, stream, nullptr, xd, seed_stride, params_unchanged != 0, next);
}

extern "C" int curious_ddpg_update_experts(const curious_net_cfg_t* cfg, int32_t n_experts, int64_t expert_stride,
                                           int64_t grad_stride, uint64_t seed_stride, float* theta_main,
                                           const float* theta_target, const float* batch,
                                           const curious_batch_layout_t* BL, int32_t B, const float* o_stats,
                                           const float* g_stats, float* workspace, float* grad, float* out_losses,
                                           float* out_Q_pi, int64_t* step_ctr, const curious_adam_state_t* adam,
                                           const curious_next_batch_t* next, curious_stream_t stream) {
  if (check_experts(cfg, n_experts, expert_stride, grad_stride)) return -1;
  CURIOUS_CHECK(cfg && (!cfg->normalize_obs || (o_stats && g_stats)),
                "curious_ddpg_update_experts: input normalisation needs the experts' statistics");
  CURIOUS_CHECK(step_ctr && adam && adam->alpha_tab && next,
                "curious_ddpg_update_experts: device step counter, step-size table and next batch are required");
  ExDim xd;
  xd.nex = n_experts; xd.stride = expert_stride; xd.gstride = grad_stride;
  return ddpg_update_impl(cfg, theta_main, theta_target, batch, BL, B, o_stats, g_stats, workspace, grad, out_losses,
                          out_Q_pi, step_ctr, adam, next, stream, xd, seed_stride);
}
