// K5-K7 (+K6 losses): actor/critic MLP forward, DDPG losses and flat gradients on fp32 MFMA.
//
// Replaces (reference): the TensorFlow-1 graph of DDPG._create_network ddpg.py:419-449 evaluated by
// DDPG._grads ddpg.py:235-243 -- MultiTaskActorCritic actor_critic.py:51-98 / nn_modular_her util.py:73-107
// (or the flat ActorCritic actor_critic.py:5-48 / nn util.py:56-71), the losses ddpg.py:436-441, tf.gradients
// ddpg.py:442-443 and flatten_grads util.py:49-53 -- plus the acting forward of DDPG.get_actions
// ddpg.py:129-146.
//
// Shape of the problem: batch 256, hidden 256 -> twenty 256^3 GEMMs per update that form chains of dependent
// layers.  They are latency bound, so the design minimises time-to-result of ONE small GEMM rather than FLOP/s:
//   * one 256-thread workgroup owns a 16 x 64 output tile; its 4 waves split the reduction dimension (split-K
//     inside the workgroup) -> 64 workgroups per 256x256 GEMM, 64 MFMAs per wave;
//   * v_mfma_f32_16x16x4_f32 (exact f32 FMA chains); each wave keeps 4 independent accumulators (the four
//     16x16 tiles made of output columns {4j+e}), so no MFMA waits on its predecessor;
//   * the k index inside a 16-wide chunk is permuted (lane group q supplies k = 4q+s at MFMA step s) and the
//     output columns are interleaved (accumulator e owns columns 4j+e) so that EVERY operand fragment is one
//     16-byte row-contiguous load per lane, straight from L2 into registers; all loads of a wave's whole K share
//     are issued before the first MFMA;
//   * partial tiles meet in LDS (16 KB), the epilogue (bias, ReLU / tanh, ReLU mask) runs on the reduced tile
//     and stores 16 bytes per lane.
// Independent chains (target actor / main critic / main actor ...) are grouped into one launch (blockIdx.z).
// Output layers (N = 1 or dimu) and everything elementwise around the losses run in "one wave per batch row"
// kernels.  Gradients are written directly at their offset of the [Q_grad | pad | pi_grad] vector.
#include <math.h>
#include <stdlib.h>
#include <algorithm>

#include "common.h"
#include "env_body.h"
#include "her_body.h"
#include "noise_body.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// device code, one section per header (all part of this translation unit)
#include "mlp_common.h"
#include "mlp_generic.h"
#include "mlp_lean_gemm.h"
#include "mlp_layer0.h"
#include "mlp_heads.h"
#include "mlp_rows.h"
#include "mlp_step.h"
#include "mlp_rows_act.h"
#include "mlp_rows_res.h"

// ================================================================== host side
struct NetOff {
  int32_t modular, nl, S, G, H, D;
  int64_t W0, b0, Wg;            // layer 0 (Wg = -1 for flat nets; W0 then has S+G rows, goal rows after the o rows)
  int64_t W[MAX_LAYERS], b[MAX_LAYERS];   // hidden layers 1..nl-1
  int64_t Wout, bout, total;
};

static NetOff net_off(const curious_net_cfg_t* c, bool critic) {
  NetOff n;
  memset(&n, 0, sizeof(n));
  n.modular = c->modular; n.nl = c->layers; n.H = c->hidden; n.G = c->dimg;
  n.D = critic ? 1 : c->dimu;
  n.S = c->dimo + (c->modular ? c->dimtd : 0) + (critic ? c->dimu : 0);
  int64_t off = 0;
  const int64_t H = c->hidden;
  if (c->modular) {
    n.W0 = off; off += (int64_t)n.S * H;
    n.b0 = off; off += H;
    n.Wg = off; off += (int64_t)n.G * H;
  } else {
    n.W0 = off; off += (int64_t)(n.S + n.G) * H;
    n.b0 = off; off += H;
    n.Wg = -1;
  }
  for (int l = 1; l < c->layers; ++l) {
    n.W[l] = off; off += H * H;
    n.b[l] = off; off += H;
  }
  n.Wout = off; off += H * n.D;
  n.bout = off; off += n.D;
  n.total = off;
  return n;
}

extern "C" int64_t curious_param_count_Q(const curious_net_cfg_t* cfg) { return net_off(cfg, true).total; }
extern "C" int64_t curious_param_count_pi(const curious_net_cfg_t* cfg) { return net_off(cfg, false).total; }
// theta_pi starts on a 256-byte boundary so that actor weight rows can be read with 16-byte loads
static int64_t pi_offset(const curious_net_cfg_t* cfg) { return (net_off(cfg, true).total + 63) & ~(int64_t)63; }
extern "C" int64_t curious_param_offset_pi(const curious_net_cfg_t* cfg) { return pi_offset(cfg); }
extern "C" int64_t curious_param_total(const curious_net_cfg_t* cfg) {
  return pi_offset(cfg) + ((net_off(cfg, false).total + 63) & ~(int64_t)63);
}

static int check_cfg(const curious_net_cfg_t* c) {
  CURIOUS_CHECK(c, "net cfg is NULL");
  CURIOUS_CHECK(c->layers >= 1 && c->layers <= MAX_LAYERS, "layers must be in 1..%d", MAX_LAYERS);
  CURIOUS_CHECK(c->hidden >= 1 && c->dimo >= 1 && c->dimg >= 0 && c->dimu >= 1 && c->dimu <= MAX_U,
                "bad network dimensions (dimu must be <= %d)", MAX_U);
  CURIOUS_CHECK(c->modular || c->dimtd == 0, "flat networks take no task descriptor");
  CURIOUS_CHECK(c->hidden % 4 == 0, "hidden must be a multiple of 4");
  return 0;
}

struct Ws {   // workspace carve-up
  float* act[5][MAX_LAYERS];   // chains: 0 target actor, 1 main critic(u), 2 main actor, 3 target critic, 4 main critic(pi)
  float* dact[3][MAX_LAYERS];  // gradient wrt hidden activations: 0 critic(u), 1 critic(pi), 2 actor
  float *pi_t, *pi, *dQ, *dz, *rows;
  float* zp[2];                // layer-0 pre-activations of target critic / main critic without the action term
  float* part[6];              // dot-epilogue partials [4 tiles][B][<=4]: pi_target, pi, Q, Q_target, Q_pi, dz
  float* qt;                   // hand-off words of the row-local pass (mlp_rows.h): [B] x 64 bit
  float* xn[2];                // input normalisation on the row-local route: normalised layer-0 input rows [B][XLD] of the
                               // main critic(u) / main actor passes (mlp_rows.h)
  float* wT[2][MAX_LAYERS];    // transposed copies of the hidden matrices of main critic / main actor (mlp_rows.h)
  int32_t* fault;              // fault word (mlp_rows.h): consumers of Q' that gave up; sticky until the host clears it.
                               // wT and fault are the ONLY parts of the workspace that carry state between calls
  int64_t total;
};

static Ws carve(const curious_net_cfg_t* c, int32_t B, float* base) {
  Ws w;
  int64_t off = 0;
  auto take = [&](int64_t n) {
    float* p = base ? base + off : nullptr;
    off += (n + 63) & ~(int64_t)63;
    return p;
  };
  const int64_t BH = (int64_t)B * c->hidden;
  for (int ch = 0; ch < 5; ++ch)
    for (int l = 0; l < c->layers; ++l) w.act[ch][l] = take(BH);
  for (int ch = 0; ch < 3; ++ch)
    for (int l = 0; l < c->layers; ++l) w.dact[ch][l] = take(BH);
  w.pi_t = take((int64_t)B * c->dimu);
  w.pi = take((int64_t)B * c->dimu);
  w.dz = take((int64_t)B * c->dimu);
  w.dQ = take(B);
  w.rows = take(3 * (int64_t)B);
  w.zp[0] = take(BH);
  w.zp[1] = take(BH);
  for (int i = 0; i < 6; ++i) w.part[i] = take(16 * (int64_t)B);
  w.qt = take(2 * (int64_t)B);
  for (int i = 0; i < 2; ++i) w.xn[i] = c->normalize_obs ? take((int64_t)B * XLD) : nullptr;
  for (int net = 0; net < 2; ++net)
    for (int l = 0; l < c->layers; ++l) w.wT[net][l] = (l >= 1) ? take((int64_t)c->hidden * c->hidden) : nullptr;
  w.fault = reinterpret_cast<int32_t*>(take(64));
  w.total = off;
  return w;
}

extern "C" int64_t curious_workspace_floats(const curious_net_cfg_t* cfg, int32_t B) {
  if (!cfg || B <= 0) return 0;
  return carve(cfg, B, nullptr).total;
}
extern "C" int64_t curious_workspace_fault_offset(const curious_net_cfg_t* cfg, int32_t B) {
  if (!cfg || B <= 0) return -1;
  static float dummy[1];
  return (int64_t)(reinterpret_cast<float*>(carve(cfg, B, dummy).fault) - dummy);
}

extern "C" int64_t curious_workspace_stamps_offset(const curious_net_cfg_t* cfg, int32_t B) {
  if (!cfg || B <= 0) return -1;
  static float dummy[1];
  return (int64_t)(carve(cfg, B, dummy).part[0] - dummy);
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

extern "C" int curious_ddpg_transposed(const curious_net_cfg_t* cfg, int32_t B, float* workspace,
                                       curious_transposed_t* out) {
  CURIOUS_CHECK(cfg && workspace && out && B > 0, "curious_ddpg_transposed: bad argument");
  if (check_cfg(cfg)) return -1;
  memset(out, 0, sizeof(*out));
  const Ws w = carve(cfg, B, workspace);
  out->fault = w.fault;
  out->fault_flag = pi_offset(cfg);                          // 1 + (offset of theta_pi - 1): always padding (P_Q % 4 == 1)
  if (cfg->hidden != 256 || cfg->layers < 2 || 2 * (cfg->layers - 1) > 8) return 0;    // nothing is kept for this shape
  const NetOff offQ = net_off(cfg, true), offPi = net_off(cfg, false);
  out->dim = cfg->hidden;
  for (int l = 1; l < cfg->layers; ++l) { out->src_off[out->n] = offQ.W[l]; out->dst[out->n++] = w.wT[0][l]; }
  for (int l = 1; l < cfg->layers; ++l) {
    out->src_off[out->n] = pi_offset(cfg) + offPi.W[l];
    out->dst[out->n++] = w.wT[1][l];
  }
  return 0;
}

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
    // (relative goals: the streaming kernel -- the resident form does not carry the goal part through its exchanges)
    if (!relative_goals && resident_ok(a, n, workspace) &&
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
  // one-launch update (mlp_step.h ddpg_step_kernel): rows_pass() only prepares the row-local launch, weight_grads()
  // enqueues it together with its tiles -- or, should the tile lists not qualify, on its own first (launch_rows)
  bool xn_rows = false;       // the row-local launch of this pass keeps the NORMALISED layer-0 input rows in w.xn
  bool one_launch = false;
  bool rows_pending = false;
  RowsArgs ra;
  size_t ra_lds = 0;
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
  xn_rows = cfg->normalize_obs != 0;
  if (cfg->normalize_obs) {
    a.o_mean = cur.o_mean; a.o_std = cur.o_std; a.g_mean = cur.g_mean; a.g_std = cur.g_std; a.nclip = cur.nclip;
    a.xn_c = w.xn[0]; a.xn_a = w.xn[1];
  }
  a.gamma = cfg->gamma; a.clip_lo = -cfg->clip_return; a.clip_hi = cfg->clip_pos_returns ? 0.0f : INFINITY;
  a.max_u = cfg->max_u;
  a.l2c = cfg->action_l2 * 2.0f / (cfg->max_u * cfg->max_u * (float)(Bl * U));
  const size_t lds = rows_lds_floats(nl) * sizeof(float);
  static bool lds_set = false;
  if (!lds_set) {                                            // > 64 KB of dynamic LDS has to be allowed once per kernel
    // (the kernel also has ~1 KB of static LDS -- the task tables of its gather blocks: dynamic + static must stay <= 160 KB)
    const int max_dyn = (int)(rows_lds_floats(ROWS_MAXL) * sizeof(float));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ddpg_rows_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, max_dyn);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ddpg_rows_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, max_dyn);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ddpg_rows_her_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, max_dyn);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ddpg_rows_her_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, max_dyn);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ddpg_step_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, max_dyn);
    (void)hipGetLastError();                                 // a refusal here must not be mistaken for a failed launch
    lds_set = true;
  }
  a.n_her = gather_in_rows ? B / ROWS_R : 0;                 // (SPB == ROWS_R: as many gather blocks as row groups)
  ra = a;
  ra_lds = lds;
  rows_pending = true;
  if (!one_launch && launch_rows()) return -1;
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
  rows_pending = false;
  dim3 grid((a.xmap || gather_in_rows ? 4 : 3) * (B / ROWS_R), 1, xd.nex);
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
  if (gather_in_rows) {
    ProfScope ps__(CK_ROWS_HER, st);
    if (xd.nex > 1) hipLaunchKernelGGL((ddpg_rows_her_kernel<true>), grid, dim3(256), lds, st, pw0a, pw0t, pw0c, a.batch,
                                       k0, k1, k2, k3, k4, k5, a, ex, her_rows, seed_stride);
    else hipLaunchKernelGGL((ddpg_rows_her_kernel<false>), grid, dim3(256), lds, st, pw0a, pw0t, pw0c, a.batch,
                            k0, k1, k2, k3, k4, k5, a, ex, her_rows, seed_stride);
  } else {
    ProfScope ps__(CK_ROWS, st);
    if (xd.nex > 1) hipLaunchKernelGGL((ddpg_rows_kernel<true>), grid, dim3(256), lds, st, pw0a, pw0t, pw0c, a.batch,
                                       k0, k1, k2, k3, k4, k5, a, ex);
    else hipLaunchKernelGGL((ddpg_rows_kernel<false>), grid, dim3(256), lds, st, pw0a, pw0t, pw0c, a.batch,
                            k0, k1, k2, k3, k4, k5, a, ex);
  }
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

int DdpgPass::weight_grads(const UpdateTail* tail) {
  // ---- weight/bias gradients: problem lists for the lean kernels (launched after the actor's backward chain)
  const bool dw_hot = dx_hot && (B % 256 == 0) && (nl - 1) <= 4 && (!cfg->normalize_obs || xn_rows);
  LossFin fin;
  fin.rows = w.rows; fin.out = out_losses; fin.B = B; fin.Bl = Bl; fin.U = U; fin.action_l2 = cfg->action_l2;
  fin.step_ctr = gather_in_rows ? step_ctr : nullptr;
  // gradients only (the all-reduce of several ranks follows): the fault word rides along as a padding element
  fin.fault = tail ? nullptr : w.fault;
  fin.flag = tail ? nullptr : grad + pi_offset(cfg) - 1;
  auto build_net = [&](bool critic, DwHotArgs& hw, int& tiles, DwSmallArgs& sm, int& stiles) -> bool {
    int nh = hw.nprob, ns_ = sm.nprob;                              // append to what the other network queued
    bool ok = true;
    const NetOff& off = critic ? offQ : offPi;
    float* g = critic ? gQ : gPi;
    const int chain = critic ? 1 : 2;
    float** dact = w.dact[critic ? 0 : 2];
    auto add_small = [&](const Seg& x, const float* dY, int lddy, int N, float* dW, float* db) {
      if (ns_ >= MAX_DW_SMALL || x.sub || x.mean || x.clip > 0.0f || !(N % 4 == 0 || N == 1) ||
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
      tiles += (H / 16) * (H / 64);
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
  int n_small_critic = 0;
  if (lean_dw) {
    lean_dw = build_net(true, hwAll, tAll, smAll, stAll);
    n_small_critic = smAll.nprob;
    lean_dw = lean_dw && build_net(false, hwAll, tAll, smAll, stAll);
  }
  CURIOUS_CHECK(xd.nex == 1 || lean_dw, "batched experts need the lean weight-gradient launch");
  if (lean_dw) {
    smAll.fin = fin;
    dwAll.n_hot = tAll;
    hwAll.tiles_per = (H / 16) * (H / 64);
    smAll.slots = stAll > 0 ? stAll : 1;
    const int nsmall = smAll.nprob * smAll.slots;           // + 1 block for the loss finalisation
    if (rows_pending && tail && copies_kept) {
      // the whole update in one launch of 4 * nrg workgroups: row groups (the spare quarter gathers the next batch), each
      // followed by its share of the tiles (mlp_step.h ddpg_step_kernel)
      RowsArgs a = ra;
      const int nrg = B / ROWS_R;
      a.n_her = tail->her ? nrg : 0;
      a.sync = w.fault + STEP_SYNC_OFFSET;
      a.n_tickets = 4 * nrg;
      a.lab_step = curious_options().lab_step;
      if (a.lab_step & 8) a.stamps = reinterpret_cast<unsigned long long*>(w.part[0]);   // lab: [256] 64-bit stamps
      AdamFuse A = tail->adam;
      A.step_add = 1;                                        // nothing advances the counter while the launch runs
      smAll.fin.step_ctr = nullptr;                          // (the last ticket does)
      StepPlan plan;
      plan.hot_c = plan.hot_a = (nl - 1) * hwAll.tiles_per;  // build_net(critic) queued its hidden matrices first
      smAll.tile0[0] = 0;
      for (int i = 0; i < smAll.nprob; ++i)
        smAll.tile0[i + 1] = smAll.tile0[i] + ((smAll.p[i].w + 15) / 16) * ((smAll.p[i].N + 63) / 64);
      smAll.n_crit = n_small_critic;
      plan.small_c = smAll.tile0[n_small_critic];
      plan.small_a = smAll.tile0[smAll.nprob] - plan.small_c;
      rows_pending = false;
      { ProfScope ps__(CK_STEP, st);
        hipLaunchKernelGGL(ddpg_step_kernel, dim3(4 * nrg), dim3(256), ra_lds, st, a, make_ex(xd, 1), dwAll, A, tail->h,
                           plan); }
      CURIOUS_LAUNCH_CHECK("ddpg_step_kernel");
      return 0;
    }
    if (rows_pending && launch_rows()) return -1;
    // XCD-aware placement of the launch's blocks (mlp_lean_gemm.h DwMap; option "dw_xcd"): applies when the hidden
    // matrices divide the 8 XCDs evenly (2 or 4 of them: 4 or 2 XCDs each)
    DwMap map;
    memset(&map, 0, sizeof(map));
    auto dw_grid = [&](int n_her) -> int {
      const int np = hwAll.nprob;
      if (curious_options().dw_xcd && (np == 2 || np == 4) && hwAll.tiles_per == 64) {
        map.units = 8 / np;
        map.r_hot = hwAll.tiles_per / map.units;
        map.r_her = (n_her + 7) / 8;
        const int r_small = smAll.slots * ((smAll.nprob + 1 + 7) / 8);      // + 1: the loss finalisation
        const int gx_ = 8 * (map.r_her + map.r_hot + r_small);
        if (B > 256) map.r_her = -map.r_her;                  // several virtual ranks: the gather blocks come last (DwMap)
        return gx_;
      }
      return n_her + tAll + nsmall + 1;
    };
    if (curious_options().lab_dw_stamps) dwAll.stamps = reinterpret_cast<unsigned long long*>(w.part[0]);
    if (tail && (!tail->her || her_lds_bytes(&tail->h.L) <= sizeof(float) * 4 * 16 * 64)) {
      const int n_her = tail->her ? (tail->h.n + SPB - 1) / SPB : 0;
      const int gx = dw_grid(n_her);
      if (dwAll.stamps && (int64_t)gx * 8 * 2 > 6 * 16 * (int64_t)B) dwAll.stamps = nullptr;   // (room: part[0..5])
      { ProfScope ps__(CK_DW_ADAM_HER, st);
        const AdamFuse& af = tail->adam;
        // (never NULL in the kernel: a block loads both words before it knows whether it will need them)
        const int32_t* fault0 = af.fault ? af.fault : reinterpret_cast<const int32_t*>(af.theta);
        const int64_t* ctr0 = af.alpha_tab ? af.step_ctr : reinterpret_cast<const int64_t*>(af.theta);
        hipLaunchKernelGGL(dw_adam_her_kernel, dim3(gx, xd.nex), dim3(256), 0, st, hwAll.tiles_per, hwAll.nprob,
                           smAll.slots, smAll.nprob, n_her, map.r_her, map.r_hot, map.units, fault0, ctr0,
                           (int64_t)xd.stride, dwAll, tail->adam, tail->h, (int64_t)xd.gstride, seed_stride); }
      CURIOUS_LAUNCH_CHECK("dw_adam_her_kernel");
      return 0;
    }
    dwAll.stamps = nullptr;
    CURIOUS_CHECK(xd.nex == 1 || !tail, "batched experts need the fused update tail");
    CURIOUS_CHECK(!copies_kept, "internal: the transposed copies are not maintained on this route");
    { ProfScope ps__(CK_DW, st);
      const int gx = dw_grid(0);
      hipLaunchKernelGGL(dw_all_kernel, dim3(gx, xd.nex), dim3(256), 0, st, hwAll.tiles_per, hwAll.nprob, smAll.slots,
                         smAll.nprob, 0, map.r_her, map.r_hot, map.units, (int64_t)xd.stride, dwAll,
                         (int64_t)xd.gstride); }
    CURIOUS_LAUNCH_CHECK("dw_all_kernel");
  } else {
    if (rows_pending && launch_rows()) return -1;
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
    p.gather_in_rows = p.rows_route() && her_lds_bytes(next->L) <= rows_lds_floats(cfg->layers) * sizeof(float) &&
                       (B % (ROWS_R * 4) == 0) && SPB == ROWS_R;
  }
  if (!rc && p.rows_route()) {
    // the copies are kept current by this pass's own optimiser tail (maintained), or -- without a tail -- by the
    // caller's stand-alone optimiser call (curious_adam_update* with `keep`), as the caller asserts
    const bool maintained = p.keeps_copies(tail);
    // option "one_launch" (default 0: it measured slower): a fused single-agent update whose tile lists
    // qualify (the conditions of keeps_copies) runs as ONE launch; the gather of the next batch then sits in the spare
    // row-group slots
    p.one_launch = tail && maintained && curious_options().one_launch && curious_options().rows_xcd && xd.nex == 1 &&
                   B % (ROWS_R * 4) == 0 && SPB == ROWS_R && B <= device_cu_count() &&
                   (!tail->her || (her_lds_bytes(&tail->h.L) <= rows_lds_floats(cfg->layers) * sizeof(float) &&
                                   (!tail->next->rng->step_ctr || tail->next->rng->step_ctr == step_ctr)));
    rc = p.rows_pass(!((maintained || !tail) && params_unchanged), maintained);
  } else {
    if (!rc) rc = p.forward();
    if (!rc) rc = p.critic_backward();
    if (!rc) rc = p.actor_backward();
  }
  if (!rc) rc = p.weight_grads(tail);
  if (!rc && next && !p.gather_in_rows) {
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
