// Host side of the network launches, part 1 (included by mlp.hip): where the inputs of a network pass live (ObsIn), the
// layer-0 segments of a modular / flat network, and the forward chains of the TILED route (one launch per layer level;
// the row-local route of mlp_rows.h walks all layers in one launch and uses none of this but the descriptors).
#pragma once

static Seg make_seg(const float* x, int ld, int w, const float* W) {
  Seg s;
  memset(&s, 0, sizeof(s));
  s.x = x; s.ld = ld; s.w = w; s.W = W;
  s.div = 1.0f; s.clip = 0.0f;
  s.vec = (ld % 4 == 0) && aligned16(x);
  return s;
}

struct ObsIn {   // where the network inputs of one pass live
  const float *o, *td, *u, *g, *ag;
  int ldo, ldtd, ldu, ldg, ldag;
  float clip;            // acting path: clip_obs
  int relative;          // acting path: relative goals
  const float *o_mean, *o_std, *g_mean, *g_std;
  float nclip;
};

// layer-0 segments of a network: modular [o | td | (u)] -> W0, g -> Wg ; flat [o | g | (u)] -> W0
static int l0_segments(const curious_net_cfg_t* c, const NetOff& n, const float* theta, const ObsIn& in,
                       bool critic, float max_u, Seg* seg) {
  int k = 0;
  const int64_t H = c->hidden;
  auto obs_seg = [&](const float* W) {
    Seg s = make_seg(in.o, in.ldo, c->dimo, W);
    s.clip = in.clip;
    s.mean = in.o_mean; s.stdv = in.o_std; s.nclip = in.nclip;
    return s;
  };
  auto goal_seg = [&](const float* W) {
    Seg s = make_seg(in.g, in.ldg, c->dimg, W);
    if (in.relative) { s.sub = in.ag; s.ldsub = in.ldag; }
    s.clip = in.clip;
    s.mean = in.g_mean; s.stdv = in.g_std; s.nclip = in.nclip;
    return s;
  };
  const float* W0 = theta ? theta + n.W0 : nullptr;
  int64_t r = 0;
  seg[k++] = obs_seg(W0);
  r += c->dimo;
  if (c->modular) {
    if (c->dimtd > 0) {
      seg[k++] = make_seg(in.td, in.ldtd, c->dimtd, W0 ? W0 + r * H : nullptr);
      r += c->dimtd;
    }
  } else {
    seg[k++] = goal_seg(W0 ? W0 + r * H : nullptr);
    r += c->dimg;
  }
  if (critic) {
    Seg s = make_seg(in.u, in.ldu, c->dimu, W0 ? W0 + r * H : nullptr);
    s.div = max_u;
    seg[k++] = s;
    r += c->dimu;
  }
  if (c->modular) seg[k++] = goal_seg(theta ? theta + n.Wg : nullptr);
  return k;
}

// One chain = one network applied to one set of inputs; forward_chains runs the hidden layers of up to 3
// independent chains, one launch per layer level.
struct Chain {
  const float* theta;   // base of this network's parameters
  NetOff off;
  ObsIn in;
  bool critic;
  float** act;          // [layers] activations out
  bool store_h0 = true; // layer-0 activations are needed later (backward pass); acting passes drop them
  int dot_mode = 0;     // dot epilogue on the LAST hidden layer (GemmHot::dot_*)
  const float* dot_w = nullptr;
  float* dot_out = nullptr;
};

static bool hot_ok(int M, int N, int K) { return (M % 16 == 0) && (N % 64 == 0) && (K % 256 == 0); }

// XCD-aware block placement of the 256 x 256 hidden-layer launches (mlp_lean_gemm.h tile_ids): rows-per-unit 0 (plain
// grid), 4 or 8; CURIOUS_XCD_MAP overrides the default for A/B measurements.
static int xcd_rows() { return curious_options().xcd_map; }
template <int XR> static dim3 xcd_grid(int nprob) { return dim3(8, 4 * XR, (nprob * (16 / XR) + 7) / 8); }

// Batched experts (mlp_common.h "Ex"): nex agents per launch, slabs `stride` floats apart.
// gstride: floats between the experts' GRADIENT vectors (they live in a contiguous [N, P] block of their own).
struct ExDim { int nex = 1; int64_t stride = 0; int64_t gstride = 0; };
static Ex make_ex(const ExDim& d, int nprob) {
  Ex e;
  e.stride = d.stride; e.nprob = nprob; e.zmul = (uint32_t)((65536 + nprob - 1) / nprob);
  return e;
}

// Fills the lean layer-0 descriptor of one chain; false when the lean kernel does not apply.
static bool l0_lean_prob(const curious_net_cfg_t* c, const Chain& C, bool with_u, bool relu, float* Y, int M,
                         L0Prob& p) {
  Seg seg[MAX_SEG];
  const int H = c->hidden;
  const int ns = l0_segments(c, C.off, C.theta, C.in, with_u, c->max_u, seg);
  bool lean = (H % 64 == 0);
  int ktot = 0;
  for (int s = 0; s < ns; ++s) {
    const Seg& sg = seg[s];
    if (!sg.vec || sg.w % 4 != 0 || sg.sub || sg.mean || !aligned16(sg.W)) lean = false;
    p.seg[s].x = sg.x; p.seg[s].W = sg.W; p.seg[s].ld = sg.ld; p.seg[s].w = sg.w; p.seg[s].div = sg.div;
    p.seg[s].clip = sg.clip > 0.0f ? sg.clip : 0.0f;
    ktot += sg.w;
  }
  if (ktot > 128 || !aligned16(Y) || !aligned16(C.theta + C.off.b0)) lean = false;
  p.nseg = ns; p.bias = C.theta + C.off.b0; p.Y = Y; p.M = M; p.N = H; p.ldy = H; p.relu = relu ? 1 : 0;
  p.ktot = ktot;
  return lean;
}

// `pre`/`npre`: extra layer-0 problems (pre-activations without the action rows, see fwd_pi_kernel) that ride on the
// layer-0 launch; only valid when the caller has verified that the lean layer-0 kernel applies to every problem.
static int forward_chains(const curious_net_cfg_t* c, Chain* ch, int nch, int M, hipStream_t st, int l_begin = 0,
                          const L0Prob* pre = nullptr, int npre = 0, const ExDim& xd = ExDim()) {
  const int H = c->hidden;
  const bool exb = xd.nex > 1;
  for (int l = l_begin; l < c->layers; ++l) {
    bool hot = (l >= 1) && hot_ok(M, H, H);
    for (int i = 0; i < nch; ++i)
      if (!aligned16(ch[i].theta) || !aligned16(ch[i].act[0])) hot = false;
    if (l == 0 && c->layers >= 2 && H == 256 && hot_ok(M, H, H) && nch <= 3 && npre <= 2) {
      // layers 0 and 1 in one launch
      L01Args fa;
      memset(&fa, 0, sizeof(fa));
      bool lean = true;
      for (int i = 0; i < nch && lean; ++i) {
        Chain& C = ch[i];
        lean = aligned16(C.theta) && aligned16(C.act[0]) &&
               l0_lean_prob(c, C, C.critic, true, C.store_h0 ? C.act[0] : nullptr, M, fa.p[i].l0);
        if (!C.store_h0) lean = lean && aligned16(C.theta + C.off.b0);
        fa.p[i].W1 = C.theta + C.off.W[1]; fa.p[i].b1 = C.theta + C.off.b[1]; fa.p[i].C = C.act[1];
      }
      if (lean) {
        int kmax = 0;
        for (int i = 0; i < nch; ++i) kmax = std::max(kmax, (int)fa.p[i].l0.ktot);
        for (int i = 0; i < npre; ++i) { fa.pre[i] = pre[i]; kmax = std::max(kmax, (int)pre[i].ktot); }
        fa.n01 = nch;
        dim3 grid(H / 64, M / 16, (nch + npre) * xd.nex);
        const Ex ex = make_ex(xd, nch + npre);
        { ProfScope ps__(CK_FWD_L01, st);
          if (kmax <= 64) {
            if (exb) hipLaunchKernelGGL((fwd_l01_kernel<1, true>), grid, dim3(256), 0, st, fa, ex);
            else hipLaunchKernelGGL((fwd_l01_kernel<1, false>), grid, dim3(256), 0, st, fa, ex);
          } else {
            if (exb) hipLaunchKernelGGL((fwd_l01_kernel<2, true>), grid, dim3(256), 0, st, fa, ex);
            else hipLaunchKernelGGL((fwd_l01_kernel<2, false>), grid, dim3(256), 0, st, fa, ex);
          } }
        CURIOUS_LAUNCH_CHECK("fwd_l01_kernel");
        ++l;                                        // layer 1 is done as well
        continue;
      }
    }
    const bool last = (l == c->layers - 1);
    bool want_dot = false;
    for (int i = 0; i < nch; ++i) want_dot = want_dot || (last && ch[i].dot_mode != 0);
    CURIOUS_CHECK(!want_dot || hot, "forward_chains: dot epilogue needs the lean hidden-layer kernel");
    if (hot) {
      HotArgs a;
      memset(&a, 0, sizeof(a));
      for (int i = 0; i < nch; ++i) {
        GemmHot& p = a.p[i];
        Chain& C = ch[i];
        p.A = C.act[l - 1]; p.lda = H; p.B = C.theta + C.off.W[l]; p.ldb = H; p.aux = C.theta + C.off.b[l];
        p.C = C.act[l]; p.ldc = H; p.M = M; p.N = H; p.K = H;
        p.dot_w = p.B;
        if (last && C.dot_mode) { p.dot_mode = C.dot_mode; p.dot_w = C.dot_w; p.dot_out = C.dot_out; p.dot_ld = H; }
      }
      dim3 grid(H / 64, M / 16, nch * xd.nex);
      const Ex ex = make_ex(xd, nch);
      { ProfScope ps__(CK_FWD_LAYER, st);
        const int xr = (!exb && M == 256 && H == 256) ? xcd_rows() : 0;
        if (want_dot) {
          if (exb) hipLaunchKernelGGL((fwd_hot_kernel<true, true>), grid, dim3(256), 0, st, a, ex);
          else if (xr == 8) hipLaunchKernelGGL((fwd_hot_kernel<true, false, 8>), xcd_grid<8>(nch), dim3(256), 0, st, a, ex);
          else if (xr == 4) hipLaunchKernelGGL((fwd_hot_kernel<true, false, 4>), xcd_grid<4>(nch), dim3(256), 0, st, a, ex);
          else hipLaunchKernelGGL((fwd_hot_kernel<true, false>), grid, dim3(256), 0, st, a, ex);
        } else {
          if (exb) hipLaunchKernelGGL((fwd_hot_kernel<false, true>), grid, dim3(256), 0, st, a, ex);
          else if (xr == 8) hipLaunchKernelGGL((fwd_hot_kernel<false, false, 8>), xcd_grid<8>(nch), dim3(256), 0, st, a, ex);
          else if (xr == 4) hipLaunchKernelGGL((fwd_hot_kernel<false, false, 4>), xcd_grid<4>(nch), dim3(256), 0, st, a, ex);
          else hipLaunchKernelGGL((fwd_hot_kernel<false, false>), grid, dim3(256), 0, st, a, ex);
        } }
      CURIOUS_LAUNCH_CHECK("fwd_hot_kernel");
      continue;
    }
    CURIOUS_CHECK(!exb, "batched experts need the lean route (hidden 256, >= 3 layers, batch % 256 == 0)");
    if (l == 0 && (H % 64 == 0)) {
      L0Args la;
      memset(&la, 0, sizeof(la));
      bool lean = true;
      for (int i = 0; i < nch && lean; ++i) lean = l0_lean_prob(c, ch[i], ch[i].critic, true, ch[i].act[0], M, la.p[i]);
      if (lean) {
        int kmax = 0;
        for (int i = 0; i < nch; ++i) kmax = std::max(kmax, (int)la.p[i].ktot);
        for (int i = 0; i < npre; ++i) { la.p[nch + i] = pre[i]; kmax = std::max(kmax, (int)pre[i].ktot); }
        dim3 grid(H / 64, (M + 15) / 16, nch + npre);
        { ProfScope ps__(CK_FWD_LAYER0, st);
          if (kmax <= 64) hipLaunchKernelGGL(fwd_l0_kernel<1>, grid, dim3(256), 0, st, la);
          else hipLaunchKernelGGL(fwd_l0_kernel<2>, grid, dim3(256), 0, st, la); }
        CURIOUS_LAUNCH_CHECK("fwd_l0_kernel");
        continue;
      }
      CURIOUS_CHECK(npre == 0, "forward_chains: lean layer-0 kernel expected");
    }
    FwdArgs a;
    memset(&a, 0, sizeof(a));
    a.nprob = nch;
    for (int i = 0; i < nch; ++i) {
      FwdProb& p = a.p[i];
      Chain& C = ch[i];
      p.M = M; p.N = H; p.Y = C.act[l]; p.ldy = H; p.act = 1;
      if (l == 0) {
        p.nseg = l0_segments(c, C.off, C.theta, C.in, C.critic, c->max_u, p.seg);
        p.bias = C.theta + C.off.b0;
      } else {
        p.seg[0] = make_seg(C.act[l - 1], H, H, C.theta + C.off.W[l]);
        p.nseg = 1;
        p.bias = C.theta + C.off.b[l];
      }
      p.wvec = (H % 4 == 0) && aligned16(C.theta) ? 1 : 0;
      for (int s = 0; s < p.nseg; ++s)
        if (!aligned16(p.seg[s].W)) p.wvec = 0;
      p.fast = p.wvec && H >= 4;
      for (int s = 0; s < p.nseg; ++s) {
        const Seg& sg = p.seg[s];
        if (!sg.vec || sg.w % 4 != 0 || sg.w < 4 || sg.sub) p.fast = 0;
        if (sg.mean && (!aligned16(sg.mean) || !aligned16(sg.stdv))) p.fast = 0;
      }
    }
    dim3 grid((H + 63) / 64, (M + 15) / 16, nch);
    { ProfScope ps__(CK_FWD_GENERIC, st); hipLaunchKernelGGL(fwd_layer_kernel, grid, dim3(256), 0, st, a); }
    CURIOUS_LAUNCH_CHECK("fwd_layer_kernel");
  }
  return 0;
}

static int launch_head_fwd(HeadFwdArgs& ha, int M, hipStream_t st) {
  dim3 grid((M + 3) / 4, 1, ha.nprob);
  { ProfScope ps__(CK_HEAD_FWD, st); hipLaunchKernelGGL(head_fwd_kernel, grid, dim3(256), 0, st, ha); }
  CURIOUS_LAUNCH_CHECK("head_fwd_kernel");
  return 0;
}

static HeadFwdProb head_prob(const float* h, int H, const float* W, const float* b, float* out, int M, int D, int act,
                             float max_u) {
  HeadFwdProb p;
  p.h = h; p.ldh = H; p.W = W; p.b = b; p.out = out; p.ldo = D; p.M = M; p.H = H; p.D = D; p.act = act;
  p.max_u = max_u;
  return p;
}

// The row-local routes (mlp_rows.h, mlp_rows_act.h).  Option "rows" = 0 (curious_set_option; initial value from
// CURIOUS_ROWS) keeps the tiled multi-launch routes: A/B measurements, the reference point of the parity checks between
// the two, and the route of shapes the row-local kernels refuse.  Read per call, so one process can run both.
