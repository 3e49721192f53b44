// K5-K7 (+K6 losses): actor/critic MLP forward, DDPG losses and flat gradients on fp32 MFMA.
//
// Replaces (reference): the TensorFlow-1 graph of DDPG._create_network ddpg.py:419-449 evaluated by
// DDPG._grads ddpg.py:235-243 -- MultiTaskActorCritic actor_critic.py:51-98 / nn_modular_her util.py:73-107
// (or the flat ActorCritic actor_critic.py:5-48 / nn util.py:56-71), the losses ddpg.py:436-441, tf.gradients
// ddpg.py:442-443 and flatten_grads util.py:49-53 -- plus the acting forward of DDPG.get_actions
// ddpg.py:129-146.
//
// Arithmetic: v_mfma_f32_16x16x4_f32 (exact f32 FMA chains, MI355X_MICROARCH "Matrix cores"); one wave owns
// one 16x16 output tile and streams its operand fragments straight from L2 (weights 0.6 MB/net and the
// 256-row activations are L2 resident; no reuse inside a wave that LDS staging could add at this tile size).
// The k index inside a 16-wide chunk is permuted (lane group q supplies k = 4q+s at MFMA step s) so that
// the row-major operand is read with one 16-byte load per lane.
// Gradients are written directly at their offset of the flat [Q_grad | pi_grad] vector (no flatten pass).
#include <math.h>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MAX_SEG 4
#define MAX_PROB 8
#define MAX_LAYERS 8

// ------------------------------------------------------------------ operand descriptors
struct Seg {             // one column segment of a layer input (virtual concatenation along k)
  const float* x;
  int32_t ld, w;
  const float* sub;      // relative goals: x - sub            (ddpg.py:119-124), acting path only
  int32_t ldsub;
  float clip;            // clip to +-clip first               (ddpg.py:125-126), acting path only; <=0: off
  const float* mean;     // input normalisation                (actor_critic.py:76-83, normalizer.py:72-77)
  const float* stdv;
  float nclip;
  float div;             // divide by max_u                    (actor_critic.py:93,96)
  const float* W;        // weight rows of this segment [w, N] row-major (forward only)
  int32_t vec;           // 16-byte loads legal
};

struct FwdProb {
  Seg seg[MAX_SEG];
  int32_t nseg;
  const float* bias;
  float* Y;
  int32_t ldy, M, N;
  int32_t act;           // 0 linear, 1 relu, 2 max_u*tanh
  float max_u;
  int32_t wvec;          // unused
};

struct DxProb {          // dX[M,K] = (dY[M,N] . W[K,N]^T) (.) mask
  const float* dY; int32_t lddy;
  const float* W;  int32_t ldw;
  const float* H;  int32_t ldh;      // relu mask source (post-activation of the layer below); NULL: none
  float* dX; int32_t lddx;
  int32_t M, N, K;
  int32_t epi;           // 0: mask only; 1: dz epilogue (actor output layer, through tanh and the l2 term)
  const float* pi; int32_t ldpi;
  float max_u, l2c;      // l2c = action_l2 * 2 / (max_u^2 * B * dimu)
  int32_t vec;
};

struct DwProb {          // dW[w,N] = X[M,w]^T . dY[M,N];  db[N] = colsum(dY)
  Seg x;
  const float* dY; int32_t lddy;
  float* dW;
  float* db;             // nullable
  int32_t M, N;
};

struct FwdArgs { FwdProb p[3]; int32_t nprob; };
struct DxArgs { DxProb p[3]; int32_t nprob; };
struct DwArgs { DwProb p[MAX_PROB * 2]; int32_t nprob; };

__device__ inline float seg_xform(const Seg& s, float v, int row, int col) {
  if (s.sub) v = __fsub_rn(v, s.sub[(int64_t)row * s.ldsub + col]);
  if (s.clip > 0.0f) v = fclip(v, -s.clip, s.clip);
  if (s.mean) v = fclip(fdiv(__fsub_rn(v, s.mean[col]), s.stdv[col]), -s.nclip, s.nclip);
  if (s.div != 1.0f) v = fdiv(v, s.div);
  return v;
}

// four consecutive columns (col .. col+3) of row `row`; out-of-range -> 0
__device__ inline f32x4 seg_load4(const Seg& s, int row, int col, bool row_ok) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (!row_ok || col >= s.w) return v;
  const float* p = s.x + (int64_t)row * s.ld + col;
  if (s.vec && col + 3 < s.w) {
    v = *reinterpret_cast<const f32x4*>(p);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (col + e < s.w) v[e] = p[e];
  }
  const bool plain = !s.sub && s.clip <= 0.0f && !s.mean && s.div == 1.0f;
  if (!plain) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (col + e < s.w) v[e] = seg_xform(s, v[e], row, col + e);
  }
  return v;
}

__device__ inline float seg_load1(const Seg& s, int row, int col, bool ok) {
  if (!ok || col >= s.w) return 0.f;
  return seg_xform(s, s.x[(int64_t)row * s.ld + col], row, col);
}

// ------------------------------------------------------------------ forward layer
// grid: x = ceil(N/64) (4 waves = 4 n-tiles), y = ceil(M/16), z = problem
__global__ __launch_bounds__(256) void fwd_layer_kernel(FwdArgs args) {
  const FwdProb& P = args.p[blockIdx.z];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.y * 16, n0 = (blockIdx.x * 4 + wave) * 16;
  if (m0 >= P.M || n0 >= P.N) return;
  const int row = m0 + i, col = n0 + i;
  const bool row_ok = row < P.M, col_ok = col < P.N;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  for (int sidx = 0; sidx < P.nseg; ++sidx) {
    const Seg& S = P.seg[sidx];
    for (int k0 = 0; k0 < S.w; k0 += 16) {
      const int kq = k0 + 4 * q;
      f32x4 a = seg_load4(S, row, kq, row_ok);
      float b[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) b[s] = (col_ok && kq + s < S.w) ? S.W[(int64_t)(kq + s) * P.N + col] : 0.f;
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc1, 0, 0, 0);
    }
  }
  if (!col_ok) return;
  const float bias = P.bias ? P.bias[col] : 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int orow = m0 + 4 * q + r;
    if (orow < P.M) {
      float v = acc0[r] + acc1[r] + bias;
      if (P.act == 1) v = fmaxf(v, 0.f);
      else if (P.act == 2) v = P.max_u * tanhf(v);
      P.Y[(int64_t)orow * P.ldy + col] = v;
    }
  }
}

// ------------------------------------------------------------------ backward: input gradient
// dX[m][k] = sum_n dY[m][n] W[k][n], masked by relu'(H).  grid: x = ceil(K/64), y = ceil(M/16), z = problem
__global__ __launch_bounds__(256) void dx_kernel(DxArgs args) {
  const DxProb& P = args.p[blockIdx.z];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.y * 16, k0 = (blockIdx.x * 4 + wave) * 16;
  if (m0 >= P.M || k0 >= P.K) return;
  const int row = m0 + i, kcol = k0 + i;
  const bool row_ok = row < P.M, k_ok = kcol < P.K;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const float* dyr = P.dY + (int64_t)row * P.lddy;
  const float* wr = P.W + (int64_t)kcol * P.ldw;
  for (int n0 = 0; n0 < P.N; n0 += 16) {
    const int nq = n0 + 4 * q;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    if (P.vec && nq + 3 < P.N) {
      if (row_ok) a = *reinterpret_cast<const f32x4*>(dyr + nq);
      if (k_ok) b = *reinterpret_cast<const f32x4*>(wr + nq);
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s)
        if (nq + s < P.N) {
          if (row_ok) a[s] = dyr[nq + s];
          if (k_ok) b[s] = wr[nq + s];
        }
    }
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc1, 0, 0, 0);
  }
  if (!k_ok) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int orow = m0 + 4 * q + r;
    if (orow < P.M) {
      float v = acc0[r] + acc1[r];
      if (P.H) v = (P.H[(int64_t)orow * P.ldh + kcol] > 0.f) ? v : 0.f;
      if (P.epi == 1) {
        // d pi_loss / d z  (ddpg.py:440-441 through pi = max_u * tanh(z), actor_critic.py:89)
        float pi = P.pi[(int64_t)orow * P.ldpi + kcol];
        float th = pi / P.max_u;
        float dpi = v / P.max_u + P.l2c * pi;
        v = dpi * P.max_u * (1.0f - th * th);
      }
      P.dX[(int64_t)orow * P.lddx + kcol] = v;
    }
  }
}

// ------------------------------------------------------------------ backward: weight gradient
// dW[k][n] = sum_m X[m][k] dY[m][n]; db[n] = sum_m dY[m][n].  grid: x = ceil(N/64), y = ceil(w/16), z = problem
__global__ __launch_bounds__(256) void dw_kernel(DwArgs args) {
  const DwProb& P = args.p[blockIdx.z];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = lane & 15, q = lane >> 4;
  const int k0 = blockIdx.y * 16, n0 = (blockIdx.x * 4 + wave) * 16;
  if (k0 >= P.x.w || n0 >= P.N) return;
  const int kcol = k0 + i, col = n0 + i;
  const bool k_ok = kcol < P.x.w, col_ok = col < P.N;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  for (int mm = 0; mm < P.M; mm += 16) {
    const int mq = mm + 4 * q;
    float a[4], b[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bool mok = mq + s < P.M;
      a[s] = seg_load1(P.x, mq + s, kcol, mok && k_ok);
      b[s] = (mok && col_ok) ? P.dY[(int64_t)(mq + s) * P.lddy + col] : 0.f;
      bsum += b[s];
    }
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc1, 0, 0, 0);
  }
  if (P.db && blockIdx.y == 0) {
    bsum += __shfl_xor(bsum, 16);
    bsum += __shfl_xor(bsum, 32);
    if (q == 0 && col_ok) P.db[col] = bsum;
  }
  if (!col_ok) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int orow = k0 + 4 * q + r;
    if (orow < P.x.w) P.dW[(int64_t)orow * P.N + col] = acc0[r] + acc1[r];
  }
}

// ------------------------------------------------------------------ output-layer backward (tiny N = D)
// For a chain with last hidden activation Hl[M,H], output weights Wout[H,D] and output gradient dOut[M,D]:
//   dH[m][n]  = (sum_j dOut[m][j] Wout[n][j]) * (Hl[m][n] > 0)
//   dWout[n][j] = sum_m Hl[m][n] dOut[m][j]      (optional)
//   dbout[j]    = sum_m dOut[m][j]               (optional)
struct HeadProb {
  const float* Hl; int32_t ldh;
  const float* Wout;
  const float* dOut; int32_t lddo;
  float* dH; int32_t lddh;
  float* dWout; float* dbout;        // nullable
  int32_t M, H, D;
};
struct HeadArgs { HeadProb p[3]; int32_t nprob; };

// grid: x = ceil(H/256), y = row blocks (16 rows each) + 1 extra block row for dWout/dbout, z = problem
__global__ __launch_bounds__(256) void head_bwd_kernel(HeadArgs args) {
  const HeadProb& P = args.p[blockIdx.z];
  const int n = blockIdx.x * 256 + threadIdx.x;
  const int nrb = (P.M + 15) / 16;
  if ((int)blockIdx.y < nrb) {
    if (n >= P.H) return;
    float w[8];
    for (int j = 0; j < P.D; ++j) w[j] = P.Wout[(int64_t)n * P.D + j];
    const int r0 = blockIdx.y * 16, r1 = min(P.M, r0 + 16);
    for (int m = r0; m < r1; ++m) {
      float s = 0.f;
      for (int j = 0; j < P.D; ++j) s += P.dOut[(int64_t)m * P.lddo + j] * w[j];
      P.dH[(int64_t)m * P.lddh + n] = (P.Hl[(int64_t)m * P.ldh + n] > 0.f) ? s : 0.f;
    }
  } else if (P.dWout) {
    if (n < P.H) {
      float acc[8];
      for (int j = 0; j < P.D; ++j) acc[j] = 0.f;
      for (int m = 0; m < P.M; ++m) {
        float h = P.Hl[(int64_t)m * P.ldh + n];
        for (int j = 0; j < P.D; ++j) acc[j] += h * P.dOut[(int64_t)m * P.lddo + j];
      }
      for (int j = 0; j < P.D; ++j) P.dWout[(int64_t)n * P.D + j] = acc[j];
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < P.D) {
      float s = 0.f;
      for (int m = 0; m < P.M; ++m) s += P.dOut[(int64_t)m * P.lddo + threadIdx.x];
      P.dbout[threadIdx.x] = s;
    }
  }
}

// ------------------------------------------------------------------ losses (single workgroup, fixed-order sums)
struct LossArgs {
  const float* Qt;      // target critic on (o_2, g_2, pi_target)   [B]
  const float* Q;       // main critic on u                         [B]
  const float* Qpi;     // main critic on pi                        [B]
  const float* pi; int32_t ldpi;   // main actor output [B, U]
  const float* r; int32_t ldr;     // reward column of the batch
  int32_t B, U;
  float gamma, clip_lo, clip_hi, max_u, action_l2;
  float* dQ;            // [B]  d Q_loss / d Q
  float* dQpi;          // [B]  d pi_loss / d Q_pi = -1/B
  float* out_losses;    // [2]
  float* out_Qpi;       // [B] copy of Qpi
  int64_t* step_ctr;    // nullable
};

__global__ __launch_bounds__(256) void loss_kernel(LossArgs a) {
  __shared__ float s_q[256], s_p[256], s_l[256];
  float lq = 0.f, lp = 0.f, ll = 0.f;
  const float invB = 1.0f / (float)a.B;
  for (int m = threadIdx.x; m < a.B; m += 256) {
    float target = fclip(a.r[(int64_t)m * a.ldr] + a.gamma * a.Qt[m], a.clip_lo, a.clip_hi);   // ddpg.py:437-438
    float diff = target - a.Q[m];
    lq += diff * diff;                                            // ddpg.py:439
    a.dQ[m] = -2.0f * invB * diff;
    float qp = a.Qpi[m];
    lp += qp;                                                     // ddpg.py:440
    a.dQpi[m] = -invB;
    a.out_Qpi[m] = qp;
    for (int j = 0; j < a.U; ++j) {
      float t = a.pi[(int64_t)m * a.ldpi + j] / a.max_u;
      ll += t * t;                                                // ddpg.py:441
    }
  }
  s_q[threadIdx.x] = lq; s_p[threadIdx.x] = lp; s_l[threadIdx.x] = ll;
  __syncthreads();
  for (int h = 128; h >= 1; h >>= 1) {
    if ((int)threadIdx.x < h) {
      s_q[threadIdx.x] += s_q[threadIdx.x + h];
      s_p[threadIdx.x] += s_p[threadIdx.x + h];
      s_l[threadIdx.x] += s_l[threadIdx.x + h];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    a.out_losses[0] = s_q[0] * invB;
    a.out_losses[1] = -s_p[0] * invB + a.action_l2 * s_l[0] / (float)(a.B * a.U);
    if (a.step_ctr) *a.step_ctr += 1;
  }
}

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
  CURIOUS_CHECK(c->hidden >= 1 && c->dimo >= 1 && c->dimg >= 0 && c->dimu >= 1 && c->dimu <= 8,
                "bad network dimensions (dimu must be <= 8)");
  CURIOUS_CHECK(c->modular || c->dimtd == 0, "flat networks take no task descriptor");
  return 0;
}

struct Ws {   // workspace carve-up
  float* act[5][MAX_LAYERS];   // chains: 0 target actor, 1 main critic(u), 2 main actor, 3 target critic, 4 main critic(pi)
  float* dact[3][MAX_LAYERS];  // gradient wrt hidden activations: 0 critic(u), 1 critic(pi), 2 actor
  float *pi_t, *pi, *Qt, *Q, *Qpi, *dQ, *dQpi, *dz;
  int64_t total;
};

static Ws carve(const curious_net_cfg_t* c, int32_t B, float* base) {
  Ws w;
  int64_t off = 0;
  auto take = [&](int64_t n) {
    float* p = base ? base + off : nullptr;
    off += (n + 3) & ~(int64_t)3;
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
  w.Qt = take(B); w.Q = take(B); w.Qpi = take(B); w.dQ = take(B); w.dQpi = take(B);
  w.total = off;
  return w;
}

extern "C" int64_t curious_workspace_floats(const curious_net_cfg_t* cfg, int32_t B) {
  if (!cfg || B <= 0) return 0;
  return carve(cfg, B, nullptr).total;
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

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

static void launch_fwd(const FwdArgs& a, hipStream_t st) {
  int maxM = 0, maxN = 0;
  for (int i = 0; i < a.nprob; ++i) {
    if (a.p[i].M > maxM) maxM = a.p[i].M;
    if (a.p[i].N > maxN) maxN = a.p[i].N;
  }
  dim3 grid((maxN + 63) / 64, (maxM + 15) / 16, a.nprob);
  hipLaunchKernelGGL(fwd_layer_kernel, grid, dim3(256), 0, st, a);
}

// Forward of `nch` independent chains through all layers (one launch per layer level).
struct Chain {
  const float* theta;   // base of this network's parameters
  NetOff off;
  ObsIn in;
  bool critic;
  float** act;          // [layers] activations out
  float* out;           // output [M, D]
  int act_out;          // 0 linear, 2 tanh
};

static int forward_chains(const curious_net_cfg_t* c, Chain* ch, int nch, int M, hipStream_t st) {
  const int H = c->hidden;
  for (int l = 0; l <= c->layers; ++l) {
    FwdArgs a;
    memset(&a, 0, sizeof(a));
    a.nprob = nch;
    for (int i = 0; i < nch; ++i) {
      FwdProb& p = a.p[i];
      Chain& C = ch[i];
      p.M = M;
      p.max_u = c->max_u;
      if (l == 0) {
        p.nseg = l0_segments(c, C.off, C.theta, C.in, C.critic, c->max_u, p.seg);
        p.bias = C.theta + C.off.b0;
      } else {
        const float* W = (l < c->layers) ? C.theta + C.off.W[l] : C.theta + C.off.Wout;
        p.seg[0] = make_seg(C.act[l - 1], H, H, W);
        p.nseg = 1;
        p.bias = (l < c->layers) ? C.theta + C.off.b[l] : C.theta + C.off.bout;
      }
      if (l < c->layers) {
        p.N = H; p.Y = C.act[l]; p.ldy = H; p.act = 1;
      } else {
        p.N = C.off.D; p.Y = C.out; p.ldy = C.off.D; p.act = C.act_out;
      }
    }
    launch_fwd(a, st);
    CURIOUS_LAUNCH_CHECK("fwd_layer_kernel");
  }
  return 0;
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
  ObsIn in;
  memset(&in, 0, sizeof(in));
  in.o = o; in.ldo = ldo; in.td = td; in.ldtd = ldtd; in.g = g; in.ldg = ldg; in.ag = ag; in.ldag = ldag;
  in.clip = clip_obs; in.relative = relative_goals; in.nclip = cfg->norm_clip;
  if (cfg->normalize_obs) {
    in.o_mean = o_stats + 2 * cfg->dimo + 1; in.o_std = o_stats + 3 * cfg->dimo + 1;
    in.g_mean = g_stats + 2 * cfg->dimg + 1; in.g_std = g_stats + 3 * cfg->dimg + 1;
  }
  Chain a;
  a.theta = theta + pi_offset(cfg); a.off = offPi; a.in = in; a.critic = false; a.act = w.act[2]; a.out = out_pi;
  a.act_out = 2;
  if (forward_chains(cfg, &a, 1, n, st)) return -2;
  if (out_Q) {
    Chain q;
    q.theta = theta; q.off = offQ; q.in = in; q.in.u = out_pi; q.in.ldu = cfg->dimu; q.critic = true;
    q.act = w.act[4]; q.out = out_Q; q.act_out = 0;
    if (forward_chains(cfg, &q, 1, n, st)) return -2;
  }
  return 0;
}

extern "C" int curious_ddpg_grads(const curious_net_cfg_t* cfg, const float* theta_main, const float* theta_target,
                                  const float* batch, const curious_batch_layout_t* BL, int32_t B,
                                  const float* o_stats, const float* g_stats, float* workspace, float* grad,
                                  float* out_losses, float* out_Q_pi, int64_t* step_ctr, curious_stream_t stream) {
  if (check_cfg(cfg)) return -1;
  CURIOUS_CHECK(theta_main && theta_target && batch && BL && workspace && grad && out_losses && out_Q_pi,
                "curious_ddpg_grads: NULL argument");
  CURIOUS_CHECK(B > 0, "curious_ddpg_grads: empty batch");
  CURIOUS_CHECK(!cfg->normalize_obs || (o_stats && g_stats), "curious_ddpg_grads: normalize_obs needs stats");
  hipStream_t st = as_stream(stream);
  const int H = cfg->hidden, nl = cfg->layers, U = cfg->dimu;
  Ws w = carve(cfg, B, workspace);
  NetOff offQ = net_off(cfg, true), offPi = net_off(cfg, false);
  const float* thQ = theta_main;
  const float* thPi = theta_main + pi_offset(cfg);
  const float* ttQ = theta_target;
  const float* ttPi = theta_target + pi_offset(cfg);
  float* gQ = grad;
  float* gPi = grad + pi_offset(cfg);
  const int ld = BL->stride;

  ObsIn cur, nxt;
  memset(&cur, 0, sizeof(cur));
  cur.o = batch + BL->off_o; cur.ldo = ld;
  cur.td = batch + BL->off_td; cur.ldtd = ld;
  cur.u = batch + BL->off_u; cur.ldu = ld;
  cur.g = batch + BL->off_g; cur.ldg = ld;
  cur.nclip = cfg->norm_clip;
  if (cfg->normalize_obs) {
    cur.o_mean = o_stats + 2 * cfg->dimo + 1; cur.o_std = o_stats + 3 * cfg->dimo + 1;
    cur.g_mean = g_stats + 2 * cfg->dimg + 1; cur.g_std = g_stats + 3 * cfg->dimg + 1;
  }
  nxt = cur;
  nxt.o = batch + BL->off_o2;            // target nets see (o_2, g_2) (ddpg.py:427-431)
  nxt.g = batch + BL->off_g2;

  // ---- forward level A: target actor, main critic(u), main actor
  Chain ch[3];
  ch[0].theta = ttPi; ch[0].off = offPi; ch[0].in = nxt; ch[0].critic = false; ch[0].act = w.act[0];
  ch[0].out = w.pi_t; ch[0].act_out = 2;
  ch[1].theta = thQ; ch[1].off = offQ; ch[1].in = cur; ch[1].critic = true; ch[1].act = w.act[1];
  ch[1].out = w.Q; ch[1].act_out = 0;
  ch[2].theta = thPi; ch[2].off = offPi; ch[2].in = cur; ch[2].critic = false; ch[2].act = w.act[2];
  ch[2].out = w.pi; ch[2].act_out = 2;
  if (forward_chains(cfg, ch, 3, B, st)) return -2;
  // ---- forward level B: target critic(pi_target), main critic(pi)
  Chain cb[2];
  cb[0].theta = ttQ; cb[0].off = offQ; cb[0].in = nxt; cb[0].in.u = w.pi_t; cb[0].in.ldu = U; cb[0].critic = true;
  cb[0].act = w.act[3]; cb[0].out = w.Qt; cb[0].act_out = 0;
  cb[1].theta = thQ; cb[1].off = offQ; cb[1].in = cur; cb[1].in.u = w.pi; cb[1].in.ldu = U; cb[1].critic = true;
  cb[1].act = w.act[4]; cb[1].out = w.Qpi; cb[1].act_out = 0;
  if (forward_chains(cfg, cb, 2, B, st)) return -2;

  // ---- losses
  LossArgs la;
  la.Qt = w.Qt; la.Q = w.Q; la.Qpi = w.Qpi; la.pi = w.pi; la.ldpi = U;
  la.r = batch + BL->off_r; la.ldr = ld; la.B = B; la.U = U;
  la.gamma = cfg->gamma; la.clip_lo = -cfg->clip_return;
  la.clip_hi = cfg->clip_pos_returns ? 0.0f : INFINITY;
  la.max_u = cfg->max_u; la.action_l2 = cfg->action_l2;
  la.dQ = w.dQ; la.dQpi = w.dQpi; la.out_losses = out_losses; la.out_Qpi = out_Q_pi; la.step_ctr = step_ctr;
  hipLaunchKernelGGL(loss_kernel, dim3(1), dim3(256), 0, st, la);
  CURIOUS_LAUNCH_CHECK("loss_kernel");

  // ---- backward through the output layers of the two critic passes
  {
    HeadArgs ha;
    memset(&ha, 0, sizeof(ha));
    ha.nprob = 2;
    HeadProb& c0 = ha.p[0];   // critic(u): also dWout, dbout of main/Q
    c0.Hl = w.act[1][nl - 1]; c0.ldh = H; c0.Wout = thQ + offQ.Wout; c0.dOut = w.dQ; c0.lddo = 1;
    c0.dH = w.dact[0][nl - 1]; c0.lddh = H; c0.dWout = gQ + offQ.Wout; c0.dbout = gQ + offQ.bout;
    c0.M = B; c0.H = H; c0.D = 1;
    HeadProb& c1 = ha.p[1];   // critic(pi): input gradient only
    c1 = c0;
    c1.Hl = w.act[4][nl - 1]; c1.dOut = w.dQpi; c1.dH = w.dact[1][nl - 1]; c1.dWout = nullptr; c1.dbout = nullptr;
    dim3 grid((H + 255) / 256, (B + 15) / 16 + 1, 2);
    hipLaunchKernelGGL(head_bwd_kernel, grid, dim3(256), 0, st, ha);
    CURIOUS_LAUNCH_CHECK("head_bwd_kernel");
  }
  // ---- hidden layers of the critic passes: dact[ch][l-1] = (dact[ch][l] . W_l^T) * relu'(act[l-1])
  for (int l = nl - 1; l >= 1; --l) {
    DxArgs da;
    memset(&da, 0, sizeof(da));
    da.nprob = 2;
    for (int k = 0; k < 2; ++k) {
      DxProb& p = da.p[k];
      const int chain = (k == 0) ? 1 : 4;
      p.dY = w.dact[k][l]; p.lddy = H; p.W = thQ + offQ.W[l]; p.ldw = H;
      p.H = w.act[chain][l - 1]; p.ldh = H; p.dX = w.dact[k][l - 1]; p.lddx = H;
      p.M = B; p.N = H; p.K = H; p.epi = 0; p.vec = (H % 4 == 0);
    }
    dim3 grid((H + 63) / 64, (B + 15) / 16, 2);
    hipLaunchKernelGGL(dx_kernel, grid, dim3(256), 0, st, da);
    CURIOUS_LAUNCH_CHECK("dx_kernel");
  }
  // ---- through layer 0 of critic(pi) into the action slot, then through tanh + l2 term -> dz
  {
    DxArgs da;
    memset(&da, 0, sizeof(da));
    da.nprob = 1;
    DxProb& p = da.p[0];
    const int64_t urow = cfg->dimo + (cfg->modular ? cfg->dimtd : cfg->dimg);   // first action row of W0
    p.dY = w.dact[1][0]; p.lddy = H; p.W = thQ + offQ.W0 + urow * H; p.ldw = H;
    p.H = nullptr; p.dX = w.dz; p.lddx = U; p.M = B; p.N = H; p.K = U; p.epi = 1;
    p.pi = w.pi; p.ldpi = U; p.max_u = cfg->max_u;
    p.l2c = cfg->action_l2 * 2.0f / (cfg->max_u * cfg->max_u * (float)(B * U));
    p.vec = (H % 4 == 0);
    dim3 grid(1, (B + 15) / 16, 1);
    hipLaunchKernelGGL(dx_kernel, grid, dim3(256), 0, st, da);
    CURIOUS_LAUNCH_CHECK("dx_kernel(dz)");
  }
  // ---- actor output layer
  {
    HeadArgs ha;
    memset(&ha, 0, sizeof(ha));
    ha.nprob = 1;
    HeadProb& a0 = ha.p[0];
    a0.Hl = w.act[2][nl - 1]; a0.ldh = H; a0.Wout = thPi + offPi.Wout; a0.dOut = w.dz; a0.lddo = U;
    a0.dH = w.dact[2][nl - 1]; a0.lddh = H; a0.dWout = gPi + offPi.Wout; a0.dbout = gPi + offPi.bout;
    a0.M = B; a0.H = H; a0.D = U;
    dim3 grid((H + 255) / 256, (B + 15) / 16 + 1, 1);
    hipLaunchKernelGGL(head_bwd_kernel, grid, dim3(256), 0, st, ha);
    CURIOUS_LAUNCH_CHECK("head_bwd_kernel(actor)");
  }
  for (int l = nl - 1; l >= 1; --l) {
    DxArgs da;
    memset(&da, 0, sizeof(da));
    da.nprob = 1;
    DxProb& p = da.p[0];
    p.dY = w.dact[2][l]; p.lddy = H; p.W = thPi + offPi.W[l]; p.ldw = H;
    p.H = w.act[2][l - 1]; p.ldh = H; p.dX = w.dact[2][l - 1]; p.lddx = H;
    p.M = B; p.N = H; p.K = H; p.epi = 0; p.vec = (H % 4 == 0);
    dim3 grid((H + 63) / 64, (B + 15) / 16, 1);
    hipLaunchKernelGGL(dx_kernel, grid, dim3(256), 0, st, da);
    CURIOUS_LAUNCH_CHECK("dx_kernel(actor)");
  }
  // ---- every weight/bias gradient of the hidden and input layers in one grouped launch
  {
    DwArgs wa;
    memset(&wa, 0, sizeof(wa));
    int np = 0, maxw = 0;
    auto add = [&](const Seg& x, const float* dY, float* dW, float* db) {
      DwProb& p = wa.p[np++];
      p.x = x; p.dY = dY; p.lddy = H; p.dW = dW; p.db = db; p.M = B; p.N = H;
      if (x.w > maxw) maxw = x.w;
    };
    for (int net = 0; net < 2; ++net) {
      const bool critic = (net == 0);
      const NetOff& off = critic ? offQ : offPi;
      float* g = critic ? gQ : gPi;
      const int chain = critic ? 1 : 2;
      float** dact = w.dact[critic ? 0 : 2];
      for (int l = nl - 1; l >= 1; --l)
        add(make_seg(w.act[chain][l - 1], H, H, nullptr), dact[l], g + off.W[l], g + off.b[l]);
      Seg seg[MAX_SEG];
      int ns = l0_segments(cfg, off, nullptr, cur, critic, cfg->max_u, seg);
      int64_t r = 0;
      for (int s = 0; s < ns; ++s) {
        const bool goal_branch = cfg->modular && s == ns - 1;
        float* dW = goal_branch ? g + off.Wg : g + off.W0 + r * H;
        add(seg[s], dact[0], dW, (s == 0) ? g + off.b0 : nullptr);
        if (!goal_branch) r += seg[s].w;
      }
    }
    CURIOUS_CHECK(np <= MAX_PROB * 2, "curious_ddpg_grads: too many gradient problems");
    wa.nprob = np;
    dim3 grid((H + 63) / 64, (maxw + 15) / 16, np);
    hipLaunchKernelGGL(dw_kernel, grid, dim3(256), 0, st, wa);
    CURIOUS_LAUNCH_CHECK("dw_kernel");
  }
  return 0;
}
