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
#include "mlp_adam_fuse.h"
#include "mlp_lean_gemm.h"
#include "mlp_dw.h"
#include "mlp_layer0.h"
#include "mlp_heads.h"
#include "mlp_act_step.h"
#include "mlp_rows.h"
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
  float* split64;              // ... of the 64 x 64 tiles (mlp_dw.h dw_hot_tile64)
  float* split;                // batches of >= DW_SPLIT_MIN_B rows: partial tiles of the split weight-gradient reduction
  int32_t* split_cnt;          // (mlp_dw.h DwSplit) and their ticket counters -- zero between launches
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
  const bool split = B >= DW_SPLIT_MIN_B;
  w.split = split ? take((int64_t)DW_SPLIT_TILES * DW_SPLIT_MAX * DW_PART) : nullptr;
  w.split64 = split ? take((int64_t)64 * DW_SPLIT_MAX * DW_PART64) : nullptr;     // 4 hidden matrices x 16 tiles of 64 x 64
  w.split_cnt = split ? reinterpret_cast<int32_t*>(take(DW_SPLIT_TILES)) : nullptr;
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
  // 1 + (offset of theta_pi - 1): the last padding element in front of theta_pi -- when there is one (P_Q is no multiple of
  // 64: every modular / flat network of the reference has P_Q % 4 == 1); 0 = no collective flag otherwise
  out->fault_flag = (pi_offset(cfg) > net_off(cfg, true).total) ? pi_offset(cfg) : 0;
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


// the host side, one header per concern (all part of this translation unit)
#include "mlp_host_forward.h"
#include "mlp_host_acting.h"
#include "mlp_host_pass.h"
#include "mlp_host_update.h"
