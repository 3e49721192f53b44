// Host side of the network launches, part 3 (included by mlp.hip): one pass of DDPG._grads (ddpg.py:235-243,419-449) as a
// launch sequence -- DdpgPass: set-up and carving, the row-local route (one launch) or the tiled forward / critic backward /
// actor backward launches.  The weight gradients and the optimiser tail follow in mlp_host_update.h.
#pragma once

struct UpdateTail {
  AdamFuse adam;
  bool her;
  HerArgs h;
  const curious_adam_state_t* st;
  const curious_next_batch_t* next;
  int64_t n_pi;
};

// One pass of DDPG._grads (+ the optimiser tail of curious_ddpg_update): the launch sequence of DESIGN.md section 4.
// setup() validates and carves, then forward() -> critic_backward() -> actor_backward() -> weight_grads() enqueue the
// launches: the row-local route (rows_pass: one launch) or the 8 tiled launches (or their generic equivalents), then
// the weight-gradient launch.
struct DdpgPass {
  // arguments
  const curious_net_cfg_t* cfg; const float* theta_main; const float* theta_target; const float* batch;
  const curious_batch_layout_t* BL; int32_t B; const float* o_stats; const float* g_stats; float* workspace;
  float* grad; float* out_losses; float* out_Q_pi; int64_t* step_ctr;
  ExDim xd; uint64_t seed_stride = 0;     // batched experts: every pointer above is expert 0's
  // derived
  hipStream_t st; int H, nl, U, ld;
  int Bl;                        // rows per (virtual) rank: cfg->loss_rows, or B
  Ws w; NetOff offQ, offPi;
  const float *thQ, *thPi, *ttQ, *ttPi; float *gQ, *gPi;
  ObsIn cur, nxt;
  Chain ch[3], cb[2];            // level A: target actor, main critic(u), main actor; level B: target critic(pi'), critic(pi)
  int64_t urow; L0Prob pre[2];
  bool fuse_pi, use_part, dx_hot, fuse_crit;

  int setup(curious_stream_t stream);
  bool rows_route() const;
  bool keeps_copies(const UpdateTail* tail) const;
  int rows_pass(bool refresh, bool maintained);
  bool copies_kept = false;   // this pass's optimiser tail has to write the transposed copies next to the parameters
  // curious_ddpg_grads* with a `next` batch: its HER gather rides in the row-local launch (spare workgroups), the step
  // counter's increment moves to the weight-gradient launch (mlp_rows.h RowsArgs.n_her)
  bool gather_in_rows = false;
  HerArgs her_rows;
  bool gather_in_dw = false;      // (gradients only, big batches: the gather rides in the weight-gradient launch; her_rows holds it)
  bool gather_done = false;       // ... and that launch took it
  bool xn_rows = false;       // the row-local launch of this pass keeps the NORMALISED layer-0 input rows in w.xn
  RowsArgs ra;
  size_t ra_lds = 0;
  int ra_R = ROWS_R;                                         // batch rows per workgroup of the row-local launch (4 | 8)
  int launch_rows();
  int forward();
  int critic_backward();
  int actor_backward();
  int weight_grads(const UpdateTail* tail);
};

int DdpgPass::setup(curious_stream_t stream) {
  if (check_cfg(cfg)) return -1;
  CURIOUS_CHECK(theta_main && theta_target && batch && BL && workspace && grad && out_losses && out_Q_pi,
                "curious_ddpg_grads: NULL argument");
  CURIOUS_CHECK(B > 0, "curious_ddpg_grads: empty batch");
  Bl = cfg->loss_rows > 0 ? cfg->loss_rows : B;
  CURIOUS_CHECK(B % Bl == 0, "curious_ddpg_grads: the batch (%d rows) is not a whole number of ranks of loss_rows = %d rows", B, Bl);
  CURIOUS_CHECK(!cfg->normalize_obs || (o_stats && g_stats), "curious_ddpg_grads: normalize_obs needs stats");
  st = as_stream(stream);
  H = cfg->hidden; nl = cfg->layers; U = cfg->dimu;
  w = carve(cfg, B, workspace);
  offQ = net_off(cfg, true); offPi = net_off(cfg, false);
  thQ = theta_main;
  thPi = theta_main + pi_offset(cfg);
  ttQ = theta_target;
  ttPi = theta_target + pi_offset(cfg);
  gQ = grad;
  gPi = grad + pi_offset(cfg);
  ld = BL->stride;

  memset(&cur, 0, sizeof(cur));
  cur.o = batch + BL->off_o; cur.ldo = ld;
  cur.td = batch + BL->off_td; cur.ldtd = ld;
  cur.u = batch + BL->off_u; cur.ldu = ld;
  cur.g = batch + BL->off_g; cur.ldg = ld;
  fill_obs_stats(cfg, cur, o_stats, g_stats);
  nxt = cur;
  nxt.o = batch + BL->off_o2;            // target nets see (o_2, g_2) (ddpg.py:427-431)
  nxt.g = batch + BL->off_g2;
  return 0;
}

bool DdpgPass::rows_route() const {
  return rows_enabled() && cfg->modular && nl >= 2 && nl <= ROWS_MAXL && H == 256 && U == 4 && (B % 16 == 0) &&
         cfg->dimo + cfg->dimtd + 4 + cfg->dimg <= ROWS_MAXIN && aligned16(thQ) && aligned16(thPi) &&
         aligned16(ttQ) && aligned16(ttPi) && aligned16(workspace) && (offQ.Wout % 4 == 0) && (offPi.Wout % 4 == 0);
}

// The backward layers of the row-local pass run on transposed copies of the main networks' hidden matrices (workspace
// w.wT).  The fused optimiser tail on the lean weight-gradient tiles keeps them current (weight_grads checks that it
// really ran); on every other route they are rebuilt from the parameters at the head of the pass.
// (the same conditions under which weight_grads() takes the fused dw_adam_her launch: lean tiles for at most 4 hidden
//  matrices -- with 4 layers per network the generic gradient launch + the stand-alone optimiser run, which do not
//  write the copies)
bool DdpgPass::keeps_copies(const UpdateTail* tail) const {
  const bool dx_ok = hot_ok(B, H, H) && aligned16(thQ) && aligned16(thPi) && aligned16(workspace);
  return tail && dx_ok && (B % 256 == 0) && (nl - 1) <= 4 && 2 * (nl - 1) <= 4 &&
         (!tail->her || her_lds_bytes(&tail->h.L) <= sizeof(float) * 4 * 16 * 64);
}

int DdpgPass::rows_pass(bool refresh, bool maintained) {
  const Ex ex = make_ex(xd, 1);
  if (refresh) {
    RowsTransposeArgs t;
    memset(&t, 0, sizeof(t));
    int n = 0;
    for (int l = 1; l < nl; ++l) { t.src[n] = thQ + offQ.W[l]; t.dst[n++] = w.wT[0][l]; }
    for (int l = 1; l < nl; ++l) { t.src[n] = thPi + offPi.W[l]; t.dst[n++] = w.wT[1][l]; }
    { ProfScope ps__(CK_ROWS_T, st);
      if (xd.nex > 1) hipLaunchKernelGGL((rows_transpose_kernel<true>), dim3(16, n, xd.nex), dim3(256), 0, st, t, ex);
      else hipLaunchKernelGGL((rows_transpose_kernel<false>), dim3(16, n, 1), dim3(256), 0, st, t, ex); }
    CURIOUS_LAUNCH_CHECK("rows_transpose_kernel");
  }
  copies_kept = maintained;
  RowsArgs a;
  memset(&a, 0, sizeof(a));
  a.tQ = rows_net(ttQ, offQ, nl); a.tPi = rows_net(ttPi, offPi, nl);
  a.mQ = rows_net(thQ, offQ, nl); a.mPi = rows_net(thPi, offPi, nl);
  a.batch = batch; a.ld = ld;
  a.off_o = BL->off_o; a.off_td = BL->off_td; a.off_u = BL->off_u; a.off_g = BL->off_g; a.off_o2 = BL->off_o2;
  a.off_g2 = BL->off_g2; a.off_r = BL->off_r;
  for (int l = 0; l < nl; ++l) {
    a.actc[l] = w.act[1][l]; a.dactc[l] = w.dact[0][l];
    a.acta[l] = w.act[2][l]; a.dacta[l] = w.dact[2][l];
    a.wTq[l] = w.wT[0][l]; a.wTpi[l] = w.wT[1][l];
  }
  a.dQ = w.dQ; a.dz = w.dz; a.rows = w.rows; a.out_Qpi = out_Q_pi; a.step_ctr = step_ctr;
  a.qt = reinterpret_cast<unsigned long long*>(w.qt);
  a.B = B; a.Bl = Bl; a.nl = nl; a.dimo = cfg->dimo; a.dimtd = cfg->dimtd; a.dimg = cfg->dimg;
  // option "rows_xcd" = 0: the plain block-id order (A/B); batched experts fill the chip several times over: plain order
  a.xmap = (curious_options().rows_xcd && xd.nex == 1) ? 1 : 0;
  a.fault = w.fault; a.inject = curious_options().fault_inject; a.spins = curious_options().qt_spins;
  a.lab_no_target = curious_options().lab_no_target;
  if (curious_options().lab_rows_stamps && xd.nex == 1 && (int64_t)(B / 4) * 3 * 16 * 2 <= 6 * 16 * (int64_t)B) {
    a.stamps = reinterpret_cast<unsigned long long*>(w.part[0]);      // (room: part[0..5], as for lab_dw_stamps)
    a.stamp_all = 1;
  }
  xn_rows = cfg->normalize_obs != 0;
  if (cfg->normalize_obs) {
    a.o_mean = cur.o_mean; a.o_std = cur.o_std; a.g_mean = cur.g_mean; a.g_std = cur.g_std; a.nclip = cur.nclip;
    a.xn_c = w.xn[0]; a.xn_a = w.xn[1];
  }
  a.gamma = cfg->gamma; a.clip_lo = -cfg->clip_return; a.clip_hi = cfg->clip_pos_returns ? 0.0f : INFINITY;
  a.max_u = cfg->max_u;
  a.l2c = cfg->action_l2 * 2.0f / (cfg->max_u * cfg->max_u * (float)(Bl * U));
  // 8 rows per workgroup once every CU holds several row groups (mlp_rows.h ROWS_R2): the rows of a batch are independent,
  // so the split changes no row's arithmetic
  // (batched experts: the rows of all experts count -- what matters is whether the row groups of 4 outnumber the chip's
  //  workgroup slots; 4 experts x 256 rows: 51.1 -> 49.5 us per launch, 8.12 -> 7.82 ms per cycle)
  ra_R = (curious_options().rows8 && (int64_t)B * xd.nex >= ROWS_R2_MIN && B % (4 * ROWS_R2) == 0) ? ROWS_R2 : ROWS_R;
  // 16 rows per workgroup, waves splitting the output columns on v_mfma_f32_16x16x4 (mlp_rows16.h), once the chip is full
  // several times over: the weight stream per row halves again and the matrix unit, not the texture path, bounds a layer.
  // Another order of summation over k than the 4- / 8-row forms (parity against the oracle, not their bits)
  // (single agent only: a bank of batched experts promises the bits of its experts updated one by one, and those -- 256 rows
  //  each -- take the 4-row form: test_batched_experts_random_banks)
  if (curious_options().rows16 > 0 && xd.nex == 1 && B >= curious_options().rows16 && B % (4 * ROWS_R3) == 0)
    ra_R = ROWS_R3;
  // The XCD map gives every actor-side group -- the longest chain -- a CU of XCDs 0-3 for itself; with more row groups than
  // those 128 CUs some of them hold two and the launch ends with those: in plain order (actor side first, then target, then
  // main critic) every CU takes the next group as a slot frees up (tools/rows_stamps.py; bench, rows kernel in us, map | plain:
  // 16 rows: 8 ranks = 128 groups 53 | 68, 10 ranks 86 | 69, 12 ranks 94 | 86, 19 ranks 132 | 113; 8 rows: 4 ranks = 128 groups
  // 42 | 50, 5 ranks 60 | 51; 4 rows: 2 ranks = 128 groups 30.6 | 36.0)
  if (B / ra_R > 128) a.xmap = 0;
  const size_t lds = rows_lds_floats(ra_R, nl) * sizeof(float);
  static bool lds_set = false;
  if (!lds_set) {                                            // > 64 KB of dynamic LDS has to be allowed once per kernel
    // (the kernel also has ~1 KB of static LDS -- the task tables of its gather blocks: dynamic + static must stay <= 80 KB
    //  for two workgroups to share a CU)
    const int max_dyn = (int)(rows_lds_floats(ROWS_R2, ROWS_MAXL) * sizeof(float));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ddpg_rows_kernel<true, ROWS_R2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, max_dyn);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ddpg_rows_kernel<false, ROWS_R2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, max_dyn);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ddpg_rows_her_kernel<true, ROWS_R2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, max_dyn);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ddpg_rows_her_kernel<false, ROWS_R2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, max_dyn);
    const int max16 = (int)(rows_lds_floats(ROWS_R3, ROWS_MAXL) * sizeof(float));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ddpg_rows16_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, max16);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ddpg_rows16_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, max16);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ddpg_rows16_her_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, max16);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ddpg_rows16_her_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, max16);
    (void)hipGetLastError();                                 // a refusal here must not be mistaken for a failed launch
    lds_set = true;
  }
  a.n_her = gather_in_rows ? B / ROWS_R : 0;                 // gather blocks of ROWS_R (= SPB) transitions each; a spare
                                                             // workgroup of the 8-row grid runs two of them
  ra = a;
  ra_lds = lds;
  if (launch_rows()) return -1;
  // what weight_grads() reads of the tiled route's state
  dx_hot = hot_ok(B, H, H) && aligned16(thQ) && aligned16(thPi) && aligned16(workspace);
  fuse_pi = use_part = fuse_crit = false;
  urow = cfg->dimo + (cfg->modular ? cfg->dimtd : cfg->dimg);
  return 0;
}

int DdpgPass::launch_rows() {
  const Ex ex = make_ex(xd, 1);
  const RowsArgs& a = ra;
  const size_t lds = ra_lds;
  dim3 grid((a.xmap || gather_in_rows ? 4 : 3) * (B / ra_R), 1, xd.nex);
  // the leading arguments (mlp_rows.h RowsPre: in scalar registers when the wave starts)
  const int Sa = a.dimo + a.dimtd, Sc = Sa + 4;
  const auto fits16 = [](int v) { return v >= 0 && v < 65536; };
  const bool pre_ok = xd.nex == 1 && a.xmap && !a.o_mean && !a.g_mean && curious_options().rows_pre &&
                      fits16(B) && fits16(a.ld) && fits16(a.off_o) && fits16(a.off_td) && fits16(a.off_u) &&
                      fits16(a.off_g) && fits16(a.off_o2) && fits16(a.off_g2) && a.dimo < 256 && a.dimtd < 256 &&
                      a.dimg < 256 && offPi.Wg == offPi.W0 + (int64_t)(Sa + 1) * H && offQ.Wg == offQ.W0 + (int64_t)(Sc + 1) * H;
  const float* pw0a = a.mPi.th + a.mPi.W0;
  const float* pw0t = a.tPi.th + a.tPi.W0;
  const float* pw0c = a.mQ.th + a.mQ.W0;
  const uint32_t k0 = (uint32_t)a.ld | ((uint32_t)a.off_o << 16), k1 = (uint32_t)a.off_td | ((uint32_t)a.off_g << 16);
  const uint32_t k2 = (uint32_t)a.off_o2 | ((uint32_t)a.off_g2 << 16);
  const uint32_t k3 = (uint32_t)a.off_u | ((uint32_t)a.dimo << 16) | ((uint32_t)a.dimtd << 24);
  const uint32_t k4 = pre_ok ? ((uint32_t)B | ((uint32_t)a.dimg << 16) | (1u << 25)) : 0u, k5 = 0u;
#define ROWS_LAUNCH(kernel, EXF, RR, ...)                                                                      \
  hipLaunchKernelGGL((kernel<EXF, RR>), grid, dim3(256), lds, st, pw0a, pw0t, pw0c, a.batch, k0, k1, k2, k3, k4, k5, a, ex, \
                     ##__VA_ARGS__)
#define ROWS_LAUNCH16(kernel, EXF, ...)                                                                        \
  hipLaunchKernelGGL((kernel<EXF>), grid, dim3(256), lds, st, pw0a, pw0t, pw0c, a.batch, k0, k1, k2, k3, k4, k5, a, ex, \
                     ##__VA_ARGS__)
#define ROWS_LAUNCH_R(kernel, EXF, ...)                                                                         \
  do {                                                                                                          \
    if (ra_R == ROWS_R2) ROWS_LAUNCH(kernel, EXF, ROWS_R2, ##__VA_ARGS__);                                      \
    else ROWS_LAUNCH(kernel, EXF, ROWS_R, ##__VA_ARGS__);                                                       \
  } while (0)
  if (gather_in_rows) {
    ProfScope ps__(CK_ROWS_HER, st);
    if (ra_R == ROWS_R3) {
      if (xd.nex > 1) ROWS_LAUNCH16(ddpg_rows16_her_kernel, true, her_rows, seed_stride);
      else ROWS_LAUNCH16(ddpg_rows16_her_kernel, false, her_rows, seed_stride);
    } else if (xd.nex > 1) ROWS_LAUNCH_R(ddpg_rows_her_kernel, true, her_rows, seed_stride);
    else ROWS_LAUNCH_R(ddpg_rows_her_kernel, false, her_rows, seed_stride);
  } else {
    ProfScope ps__(CK_ROWS, st);
    if (ra_R == ROWS_R3) {
      if (xd.nex > 1) ROWS_LAUNCH16(ddpg_rows16_kernel, true);
      else ROWS_LAUNCH16(ddpg_rows16_kernel, false);
    } else if (xd.nex > 1) ROWS_LAUNCH_R(ddpg_rows_kernel, true);
    else ROWS_LAUNCH_R(ddpg_rows_kernel, false);
  }
#undef ROWS_LAUNCH16
#undef ROWS_LAUNCH_R
#undef ROWS_LAUNCH
  CURIOUS_LAUNCH_CHECK("ddpg_rows_kernel");
  return 0;
}

int DdpgPass::forward() {
  // ---- forward level A: hidden layers of target actor, main critic(u), main actor
  ch[0].theta = ttPi; ch[0].off = offPi; ch[0].in = nxt; ch[0].critic = false; ch[0].act = w.act[0];
  ch[1].theta = thQ; ch[1].off = offQ; ch[1].in = cur; ch[1].critic = true; ch[1].act = w.act[1];
  ch[2].theta = thPi; ch[2].off = offPi; ch[2].in = cur; ch[2].critic = false; ch[2].act = w.act[2];
  // level B = hidden layers of target critic(pi_target), main critic(pi)
  cb[0].theta = ttQ; cb[0].off = offQ; cb[0].in = nxt; cb[0].in.u = w.pi_t; cb[0].in.ldu = U; cb[0].critic = true;
  cb[0].act = w.act[3];
  cb[1].theta = thQ; cb[1].off = offQ; cb[1].in = cur; cb[1].in.u = w.pi; cb[1].in.ldu = U; cb[1].critic = true;
  cb[1].act = w.act[4];
  urow = cfg->dimo + (cfg->modular ? cfg->dimtd : cfg->dimg);                 // first action row of W0
  // Lean route: the action-independent part of level B's layer 0 rides on level A's layer-0 launch, the actor output
  // layers and the action rows are folded into level B's layer-1 launch (fwd_pi_kernel).
  memset(pre, 0, sizeof(pre));
  fuse_pi = nl >= 2 && H == 256 && U == 4 && (B % 16 == 0) && aligned16(thQ) && aligned16(thPi) && aligned16(ttQ) &&
                 aligned16(ttPi) && aligned16(workspace) && aligned16(thPi + offPi.Wout) &&
                 aligned16(thQ + offQ.W0 + urow * H);
  if (fuse_pi) {
    L0Prob tmp;
    for (int i = 0; i < 3 && fuse_pi; ++i) fuse_pi = l0_lean_prob(cfg, ch[i], ch[i].critic, true, ch[i].act[0], B, tmp);
    for (int i = 0; i < 2 && fuse_pi; ++i) fuse_pi = l0_lean_prob(cfg, cb[i], false, false, w.zp[i], B, pre[i]);
  }
  // With >= 3 layers the last hidden layer of every chain runs on the lean kernel, whose dot epilogue leaves the
  // output-layer products as 4 column-tile partials: the fused prologues downstream then add 4 numbers per row
  // instead of contracting 256-wide rows.  part[]: 0 pi_target, 1 pi, 2 Q, 3 Q_target, 4 Q_pi, 5 dz.
  use_part = fuse_pi && nl >= 3;
  CURIOUS_CHECK(xd.nex == 1 || use_part,
                "batched experts need the lean route (hidden 256, >= 3 layers, dimu 4, batch % 256 == 0)");
  if (use_part) {
    ch[0].dot_mode = 2; ch[0].dot_w = ttPi + offPi.Wout; ch[0].dot_out = w.part[0];
    ch[1].dot_mode = 1; ch[1].dot_w = thQ + offQ.Wout; ch[1].dot_out = w.part[2];
    ch[2].dot_mode = 2; ch[2].dot_w = thPi + offPi.Wout; ch[2].dot_out = w.part[1];
    cb[0].dot_mode = 1; cb[0].dot_w = ttQ + offQ.Wout; cb[0].dot_out = w.part[3];
    cb[1].dot_mode = 1; cb[1].dot_w = thQ + offQ.Wout; cb[1].dot_out = w.part[4];
  }
  if (fuse_pi) {
    if (forward_chains(cfg, ch, 3, B, st, 0, pre, 2, xd)) return -2;
    FwdPiArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.max_u = cfg->max_u; fa.B = B;
    for (int i = 0; i < 2; ++i) {
      FwdPiProb& p = fa.p[i];
      const float* tq = (i == 0) ? ttQ : thQ;
      const float* tp = (i == 0) ? ttPi : thPi;
      p.part = w.part[i];
      p.a_last = w.act[i == 0 ? 0 : 2][nl - 1]; p.WoutPi = tp + offPi.Wout; p.boutPi = tp + offPi.bout;
      p.zp = w.zp[i]; p.Wu = tq + offQ.W0 + urow * H; p.W1 = tq + offQ.W[1]; p.b1 = tq + offQ.b[1];
      p.pi_out = (i == 0) ? nullptr : w.pi;                 // pi_target is consumed here only
      p.h0_out = (i == 0) ? nullptr : w.act[4][0];          // relu mask of the actor-loss backward pass
      p.C = w.act[i == 0 ? 3 : 4][1];
    }
    dim3 grid(H / 64, B / 16, 2 * xd.nex);
    const Ex ex = make_ex(xd, 2);
    { ProfScope ps__(CK_FWD_PI, st);
      if (use_part) {
        if (xd.nex > 1) hipLaunchKernelGGL((fwd_pi_kernel<true, true>), grid, dim3(256), 0, st, fa, ex);
        else hipLaunchKernelGGL((fwd_pi_kernel<true, false>), grid, dim3(256), 0, st, fa, ex);
      } else {
        hipLaunchKernelGGL((fwd_pi_kernel<false, false>), grid, dim3(256), 0, st, fa, ex);
      } }
    CURIOUS_LAUNCH_CHECK("fwd_pi_kernel");
    if (forward_chains(cfg, cb, 2, B, st, 2, nullptr, 0, xd)) return -2;
  } else {
    if (forward_chains(cfg, ch, 3, B, st)) return -2;
    // ---- actor output layers: pi_target, pi
    HeadFwdArgs ha;
    memset(&ha, 0, sizeof(ha));
    ha.nprob = 2;
    ha.p[0] = head_prob(w.act[0][nl - 1], H, ttPi + offPi.Wout, ttPi + offPi.bout, w.pi_t, B, U, 2, cfg->max_u);
    ha.p[1] = head_prob(w.act[2][nl - 1], H, thPi + offPi.Wout, thPi + offPi.bout, w.pi, B, U, 2, cfg->max_u);
    if (launch_head_fwd(ha, B, st)) return -2;
    // ---- forward level B
    if (forward_chains(cfg, cb, 2, B, st)) return -2;
  }
  return 0;
}

int DdpgPass::critic_backward() {
  // ---- critic output layers, per-row loss terms, backward through the output layers (fused with the first hidden
  //      backward level when the lean kernels apply)
  dx_hot = hot_ok(B, H, H) && aligned16(thQ) && aligned16(thPi) && aligned16(workspace);
  fuse_crit = dx_hot && nl >= 2 && H == 256;
  CURIOUS_CHECK(!use_part || fuse_crit, "curious_ddpg_grads: inconsistent lean-path conditions");
  CURIOUS_CHECK(xd.nex == 1 || (fuse_crit && use_part), "batched experts need the lean route");
  if (fuse_crit) {
    DxCritArgs a;
    memset(&a, 0, sizeof(a));
    a.partQ = w.part[2]; a.partQt = w.part[3]; a.partQpi = w.part[4];
    const int l = nl - 1;
    a.hl[0] = w.act[1][l]; a.hl[1] = w.act[4][l];
    a.hprev[0] = w.act[1][l - 1]; a.hprev[1] = w.act[4][l - 1];
    a.dY[0] = w.dact[0][l]; a.dY[1] = w.dact[1][l];
    a.dX[0] = w.dact[0][l - 1]; a.dX[1] = w.dact[1][l - 1];
    a.W = thQ + offQ.W[l];
    a.WoutQ = thQ + offQ.Wout; a.boutQ = thQ + offQ.bout;
    a.e2 = w.act[3][l]; a.WoutQt = ttQ + offQ.Wout; a.boutQt = ttQ + offQ.bout;
    a.r = batch + BL->off_r; a.ldr = ld; a.pi = w.pi; a.ldpi = U;
    a.B = B; a.Bl = Bl; a.H = H; a.U = U;
    a.gamma = cfg->gamma; a.clip_lo = -cfg->clip_return; a.clip_hi = cfg->clip_pos_returns ? 0.0f : INFINITY;
    a.max_u = cfg->max_u;
    a.dQ = w.dQ; a.rows = w.rows; a.out_Qpi = out_Q_pi; a.step_ctr = step_ctr;
    dim3 grid(H / 64, B / 16, 2 * xd.nex);
    const Ex ex = make_ex(xd, 2);
    { ProfScope ps__(CK_CRITIC_HEAD, st);
      if (use_part) {
        if (xd.nex > 1) hipLaunchKernelGGL((dx_crit_kernel<true, true>), grid, dim3(256), 0, st, a, ex);
        else hipLaunchKernelGGL((dx_crit_kernel<true, false>), grid, dim3(256), 0, st, a, ex);
      } else {
        hipLaunchKernelGGL((dx_crit_kernel<false, false>), grid, dim3(256), 0, st, a, ex);
      } }
    CURIOUS_LAUNCH_CHECK("dx_crit_kernel");
  } else
  {
    CriticHeadArgs a;
    a.c2 = w.act[1][nl - 1]; a.d2 = w.act[4][nl - 1]; a.e2 = w.act[3][nl - 1];
    a.WoutQ = thQ + offQ.Wout; a.boutQ = thQ + offQ.bout;
    a.WoutQt = ttQ + offQ.Wout; a.boutQt = ttQ + offQ.bout;
    a.r = batch + BL->off_r; a.ldr = ld; a.pi = w.pi; a.ldpi = U;
    a.B = B; a.Bl = Bl; a.H = H; a.U = U;
    a.gamma = cfg->gamma; a.clip_lo = -cfg->clip_return; a.clip_hi = cfg->clip_pos_returns ? 0.0f : INFINITY;
    a.max_u = cfg->max_u;
    a.dc2 = w.dact[0][nl - 1]; a.dd2 = w.dact[1][nl - 1]; a.dQ = w.dQ; a.rows = w.rows; a.out_Qpi = out_Q_pi;
    a.step_ctr = step_ctr;
    { ProfScope ps__(CK_CRITIC_HEAD_GENERIC, st);
      hipLaunchKernelGGL(critic_head_kernel, dim3((B + 3) / 4), dim3(256), 0, st, a); }
    CURIOUS_LAUNCH_CHECK("critic_head_kernel");
  }
  // ---- hidden layers of the two critic passes: dact[k][l-1] = (dact[k][l] . W_l^T) * relu'(act[l-1])
  for (int l = fuse_crit ? nl - 2 : nl - 1; l >= 1; --l) {
    if (dx_hot) {
      HotArgs ha;
      memset(&ha, 0, sizeof(ha));
      for (int k = 0; k < 2; ++k) {
        GemmHot& p = ha.p[k];
        const int chain = (k == 0) ? 1 : 4;
        p.A = w.dact[k][l]; p.lda = H; p.B = thQ + offQ.W[l]; p.ldb = H; p.aux = w.act[chain][l - 1];
        p.C = w.dact[k][l - 1]; p.ldc = H; p.M = B; p.N = H; p.K = H;
        p.dot_w = p.B;
        if (use_part && l == 1 && k == 1) {                  // d pi_loss / d(action slot): dd0 . Wu^T as partials
          p.dot_mode = 3; p.dot_w = thQ + offQ.W0 + urow * H; p.dot_out = w.part[5]; p.dot_ld = H;
        }
      }
      dim3 grid(H / 64, B / 16, 2 * xd.nex);
      const Ex ex = make_ex(xd, 2);
      { ProfScope ps__(CK_DX, st);
        const int xr = (xd.nex == 1 && B == 256 && H == 256) ? xcd_rows() : 0;
        if (use_part && l == 1) {
          if (xd.nex > 1) hipLaunchKernelGGL((dx_hot_kernel<true, true>), grid, dim3(256), 0, st, ha, ex);
          else if (xr == 8) hipLaunchKernelGGL((dx_hot_kernel<true, false, 8>), xcd_grid<8>(2), dim3(256), 0, st, ha, ex);
          else if (xr == 4) hipLaunchKernelGGL((dx_hot_kernel<true, false, 4>), xcd_grid<4>(2), dim3(256), 0, st, ha, ex);
          else hipLaunchKernelGGL((dx_hot_kernel<true, false>), grid, dim3(256), 0, st, ha, ex);
        } else {
          if (xd.nex > 1) hipLaunchKernelGGL((dx_hot_kernel<false, true>), grid, dim3(256), 0, st, ha, ex);
          else if (xr == 8) hipLaunchKernelGGL((dx_hot_kernel<false, false, 8>), xcd_grid<8>(2), dim3(256), 0, st, ha, ex);
          else if (xr == 4) hipLaunchKernelGGL((dx_hot_kernel<false, false, 4>), xcd_grid<4>(2), dim3(256), 0, st, ha, ex);
          else hipLaunchKernelGGL((dx_hot_kernel<false, false>), grid, dim3(256), 0, st, ha, ex);
        } }
      CURIOUS_LAUNCH_CHECK("dx_hot_kernel");
      continue;
    }
    DxArgs da;
    memset(&da, 0, sizeof(da));
    da.nprob = 2;
    for (int k = 0; k < 2; ++k) {
      DxProb& p = da.p[k];
      const int chain = (k == 0) ? 1 : 4;
      p.dY = w.dact[k][l]; p.lddy = H; p.W = thQ + offQ.W[l]; p.ldw = H;
      p.H = w.act[chain][l - 1]; p.ldh = H; p.dX = w.dact[k][l - 1]; p.lddx = H;
      p.M = B; p.N = H; p.K = H; p.vec = aligned16(p.W) ? 1 : 0;
      p.fast = p.vec && aligned16(p.dY) && H >= 4;
    }
    dim3 grid((H + 63) / 64, (B + 15) / 16, 2);
    { ProfScope ps__(CK_DX_GENERIC, st); hipLaunchKernelGGL(dx_kernel, grid, dim3(256), 0, st, da); }
    CURIOUS_LAUNCH_CHECK("dx_kernel");
  }
  return 0;
}

int DdpgPass::actor_backward() {
  // ---- into the action slot of critic(pi), through tanh + l2 term -> dz; backward through the actor output layer
  //      (fused with the actor's first hidden backward level when the lean kernels apply)
  const float l2c = cfg->action_l2 * 2.0f / (cfg->max_u * cfg->max_u * (float)(Bl * U));
  const bool fuse_actor = fuse_crit && U == 4 && aligned16(w.pi) && aligned16(w.dz) && aligned16(thQ + offQ.W0 + urow * H) &&
                          aligned16(thPi + offPi.Wout);
  CURIOUS_CHECK(!use_part || fuse_actor, "curious_ddpg_grads: inconsistent lean-path conditions");
  if (fuse_actor) {
    DxActorArgs a;
    const int l = nl - 1;
    a.part = w.part[5];
    a.dd0 = w.dact[1][0]; a.Wu = thQ + offQ.W0 + urow * H; a.pi = w.pi;
    a.a2 = w.act[2][l]; a.WoutPi = thPi + offPi.Wout; a.hprev = w.act[2][l - 1]; a.W = thPi + offPi.W[l];
    a.dz = w.dz; a.da2 = w.dact[2][l]; a.dX = w.dact[2][l - 1];
    a.B = B; a.max_u = cfg->max_u; a.l2c = l2c;
    dim3 grid(H / 64, B / 16, xd.nex);
    const Ex ex = make_ex(xd, 1);
    { ProfScope ps__(CK_ACTOR_DZ, st);
      if (use_part) {
        if (xd.nex > 1) hipLaunchKernelGGL((dx_actor_kernel<true, true>), grid, dim3(256), 0, st, a, ex);
        else hipLaunchKernelGGL((dx_actor_kernel<true, false>), grid, dim3(256), 0, st, a, ex);
      } else {
        hipLaunchKernelGGL((dx_actor_kernel<false, false>), grid, dim3(256), 0, st, a, ex);
      } }
    CURIOUS_LAUNCH_CHECK("dx_actor_kernel");
  } else {
    ActorDzArgs a;
    a.dd0 = w.dact[1][0]; a.Wu = thQ + offQ.W0 + urow * H; a.pi = w.pi; a.ldpi = U;
    a.a2 = w.act[2][nl - 1]; a.WoutPi = thPi + offPi.Wout; a.dz = w.dz; a.da2 = w.dact[2][nl - 1];
    a.B = B; a.H = H; a.U = U; a.max_u = cfg->max_u;
    a.l2c = l2c;
    { ProfScope ps__(CK_ACTOR_DZ_GENERIC, st);
      hipLaunchKernelGGL(actor_dz_kernel, dim3((B + 3) / 4), dim3(256), 0, st, a); }
    CURIOUS_LAUNCH_CHECK("actor_dz_kernel");
  }
  for (int l = fuse_actor ? nl - 2 : nl - 1; l >= 1; --l) {
    if (dx_hot) {
      HotArgs ha;
      memset(&ha, 0, sizeof(ha));
      GemmHot& p = ha.p[0];
      p.A = w.dact[2][l]; p.lda = H; p.B = thPi + offPi.W[l]; p.ldb = H; p.aux = w.act[2][l - 1];
      p.C = w.dact[2][l - 1]; p.ldc = H; p.M = B; p.N = H; p.K = H;
      dim3 grid(H / 64, B / 16, xd.nex);
      const Ex ex = make_ex(xd, 1);
      const int xr = (xd.nex == 1 && B == 256 && H == 256) ? xcd_rows() : 0;
      { ProfScope ps__(CK_DX, st);
        if (xd.nex > 1) hipLaunchKernelGGL((dx_hot_kernel<false, true>), grid, dim3(256), 0, st, ha, ex);
        else if (xr == 8) hipLaunchKernelGGL((dx_hot_kernel<false, false, 8>), xcd_grid<8>(1), dim3(256), 0, st, ha, ex);
        else if (xr == 4) hipLaunchKernelGGL((dx_hot_kernel<false, false, 4>), xcd_grid<4>(1), dim3(256), 0, st, ha, ex);
        else hipLaunchKernelGGL((dx_hot_kernel<false, false>), grid, dim3(256), 0, st, ha, ex); }
      CURIOUS_LAUNCH_CHECK("dx_hot_kernel(actor)");
      continue;
    }
    DxArgs da;
    memset(&da, 0, sizeof(da));
    da.nprob = 1;
    DxProb& p = da.p[0];
    p.dY = w.dact[2][l]; p.lddy = H; p.W = thPi + offPi.W[l]; p.ldw = H;
    p.H = w.act[2][l - 1]; p.ldh = H; p.dX = w.dact[2][l - 1]; p.lddx = H;
    p.M = B; p.N = H; p.K = H; p.vec = aligned16(p.W) ? 1 : 0;
    p.fast = p.vec && aligned16(p.dY) && H >= 4;
    dim3 grid((H + 63) / 64, (B + 15) / 16, 1);
    { ProfScope ps__(CK_DX_GENERIC, st); hipLaunchKernelGGL(dx_kernel, grid, dim3(256), 0, st, da); }
    CURIOUS_LAUNCH_CHECK("dx_kernel(actor)");
  }
  return 0;
}

