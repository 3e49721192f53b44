// Output-layer kernels and the fused prologue kernels: head_fwd, fwd_pi, critic_head, dx_crit, actor_dz, dx_actor,
// act_step (included by mlp.hip).
#pragma once

// ------------------------------------------------------------------ one-wave-per-row kernels

// out[m][d] = f(sum_k h[m][k] W[k][d] + b[d]),  D <= MAX_U
struct HeadFwdProb {
  const float* h; int32_t ldh;
  const float* W; const float* b;
  float* out; int32_t ldo;
  int32_t M, H, D, act;      // act 2: max_u * tanh
  float max_u;
};
struct HeadFwdArgs { HeadFwdProb p[3]; int32_t nprob; };

__device__ inline void row_dot(const float* hrow, const float* W, int H, int D, int lane, float* out /*[MAX_U]*/) {
  float acc[MAX_U];
#pragma unroll
  for (int d = 0; d < MAX_U; ++d) acc[d] = 0.f;
  const bool hv = ((uintptr_t)hrow & 15) == 0;
  for (int k = 4 * lane; k < H; k += 256) {
    f32x4 h4 = ldg4(hrow + k, H - k, hv);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (k + e < H) {
        const float* wr = W + (int64_t)(k + e) * D;
#pragma unroll
        for (int d = 0; d < MAX_U; ++d)
          if (d < D) acc[d] += h4[e] * wr[d];
      }
    }
  }
#pragma unroll
  for (int d = 0; d < MAX_U; ++d) out[d] = (d < D) ? wave_sum(acc[d]) : 0.f;
}

// Branch-free specialisations (H % 4 == 0 is guaranteed by check_cfg): D = 1 (critic) and D = 4 (Fetch actor).
// Rows k..k+3 of W[H, D] are 4*D contiguous floats, read with D unconditional 16-byte loads.
template <int D>
__device__ inline void row_dot_fast(const float* hrow, const float* W, int H, int lane, float* out /*[D]*/) {
  float acc[D];
#pragma unroll
  for (int d = 0; d < D; ++d) acc[d] = 0.f;
  const int trips = (H + 255) >> 8;
  for (int t = 0; t < trips; ++t) {
    const int k = 256 * t + 4 * lane;
    const bool ok = k < H;
    const int kc = min(k, H - 4);
    f32x4 h4 = sel4(ok, ldv(hrow + kc));
    if (D == 1) {
      f32x4 w = ldv(W + kc);
      acc[0] += h4[0] * w[0] + h4[1] * w[1] + h4[2] * w[2] + h4[3] * w[3];
    } else {
      f32x4 w[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = ldv(W + (int64_t)(kc + e) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int d = 0; d < 4; ++d) acc[d] += h4[e] * w[e][d];
    }
  }
#pragma unroll
  for (int d = 0; d < D; ++d) out[d] = wave_sum(acc[d]);
}

// grid: x = ceil(M/4) (one wave per row), z = problem
__global__ __launch_bounds__(256) void head_fwd_kernel(HeadFwdArgs args) {
  const HeadFwdProb& P = args.p[blockIdx.z];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + wave;
  if (m >= P.M) return;
  float o[MAX_U];
#pragma unroll
  for (int d = 0; d < MAX_U; ++d) o[d] = 0.f;
  const float* hrow = P.h + (int64_t)m * P.ldh;
  const bool al = (((uintptr_t)hrow | (uintptr_t)P.W) & 15) == 0;
  if (al && P.D == 4) row_dot_fast<4>(hrow, P.W, P.H, lane, o);
  else if (al && P.D == 1) row_dot_fast<1>(hrow, P.W, P.H, lane, o);
  else row_dot(hrow, P.W, P.H, P.D, lane, o);
  if (lane < P.D) {
    float v = 0.f;
#pragma unroll
    for (int d = 0; d < MAX_U; ++d)
      if (d == lane) v = o[d];
    v += P.b[lane];
    if (P.act == 2) v = P.max_u * tanhf(v);                  // actor_critic.py:89
    P.out[(int64_t)m * P.ldo + lane] = v;
  }
}

// Layer 1 of the two critic(pi) passes with the actor output layer and the critic's layer 0 folded into its prologue
// (replaces head_fwd_kernel + a second fwd_l0_kernel: two ~4.5 us dependent stages per update).  The first layer-0
// launch already produced zp = [o | td] . W0 + g . Wg + b0 (everything but the action rows, no relu).  Every workgroup
// recomputes pi = max_u * tanh(a_last . Wout + bout) for its 16 batch rows (a wavefront per row, the arithmetic and
// order of head_fwd_kernel), builds its A operand  h0[m][k] = relu(zp[m][k] + sum_d (pi[m][d] / max_u) * Wu[d][k])
// on the fly and runs the usual split-K tile.  Column-tile 0 writes pi and h0 for the backward pass.  H == 256, dimu == 4.
struct FwdPiProb {
  const float* part;       // PART: [4][B][4] partial products a_last . Wout of the producing launch's column tiles
  const float* a_last; const float* WoutPi; const float* boutPi;
  const float* zp; const float* Wu; const float* W1; const float* b1;
  float* pi_out; float* h0_out; float* C;
};
struct FwdPiArgs { FwdPiProb p[2]; float max_u; int32_t B; };

template <bool PART, bool EX>
__global__ __launch_bounds__(256) void fwd_pi_kernel(FwdPiArgs args, Ex ex) {
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];
  __shared__ __attribute__((aligned(16))) float s_pi[16 * 4];
  int64_t eo;
  const FwdPiProb& P = args.p[ex_decode<EX>(ex, blockIdx.z, eo)];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 64;
  const int H = 256;
  const int pm = m0 + 4 * wave;
  // ---- all loads
  f32x4 pr_h[4], wp[4];
  float pp[4] = {0.f, 0.f, 0.f, 0.f};
  if (PART) {
    // thread t < 64 finishes pi[m0 + t/4][t%4] from the 4 column-tile partials
#pragma unroll
    for (int t = 0; t < 4; ++t) pp[t] = P.part[eo + ((int64_t)t * args.B + m0) * 4 + (tid & 63)];
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) pr_h[r] = ldv(P.a_last + eo + (int64_t)(pm + r) * H + 4 * lane);
#pragma unroll
    for (int e = 0; e < 4; ++e) wp[e] = ldv(P.WoutPi + eo + (int64_t)(4 * lane + e) * 4);
  }
  const float bo = P.boutPi[eo + (lane & 3)];
  const float* xr = P.zp + eo + (int64_t)(m0 + j) * H;
  const float* wc = P.W1 + eo + n0 + 4 * j;
  const f32x4 bias = ldv(P.b1 + eo + n0 + 4 * (tid & 15));
  f32x4 z[4], wu[4][4], b[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int kq = (wave + 4 * u) * 16 + 4 * q;
    z[u] = ldv(xr + kq);
#pragma unroll
    for (int d = 0; d < 4; ++d) wu[u][d] = ldv(P.Wu + eo + (int64_t)d * H + kq);
#pragma unroll
    for (int s = 0; s < 4; ++s) b[u][s] = ldv(wc + (int64_t)(kq + s) * H);
  }
  LOADS_FIRST();
  // ---- prologue: actor output layer of the 16 rows
  if (PART) {
    if (tid < 64) {
      const float pv = args.max_u * tanhf(((pp[0] + pp[1]) + (pp[2] + pp[3])) + bo);   // actor_critic.py:89
      s_pi[tid] = pv;
      if (blockIdx.x == 0 && P.pi_out) P.pi_out[eo + (int64_t)m0 * 4 + tid] = pv;
    }
  } else {
    // lane 4r+d of wave w finishes pi[pm + r][d]
    float sums[16];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc += pr_h[r][e] * wp[e][d];
        sums[4 * r + d] = wave_sum(acc);
      }
    const float mine = pick16(sums, lane);
    if (lane < 16) {
      const float pv = args.max_u * tanhf(mine + bo);           // actor_critic.py:89
      s_pi[16 * wave + lane] = pv;
      if (blockIdx.x == 0 && P.pi_out) P.pi_out[eo + (int64_t)pm * 4 + lane] = pv;
    }
  }
  __syncthreads();
  f32x4 ud = *reinterpret_cast<const f32x4*>(s_pi + 4 * j);
#pragma unroll
  for (int d = 0; d < 4; ++d) ud[d] = fdiv(ud[d], args.max_u);  // actor_critic.py:93 (pi_tf / max_u)
  f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    f32x4 av;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float t = ud[0] * wu[u][0][s] + ud[1] * wu[u][1][s] + ud[2] * wu[u][2][s] + ud[3] * wu[u][3][s];
      av[s] = fmaxf(z[u][s] + t, 0.f);
    }
    if (blockIdx.x == 0 && P.h0_out) {
      const int kq = (wave + 4 * u) * 16 + 4 * q;
      *reinterpret_cast<f32x4*>(P.h0_out + eo + (int64_t)(m0 + j) * H + kq) = av;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = MFMA(av[s], b[u][s][e], acc[e]);
  }
  int orow, c4;
  f32x4 v = reduce_tile(red, acc, wave, q, j, tid, orow, c4);
  v += bias;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
  *reinterpret_cast<f32x4*>(P.C + eo + (int64_t)(m0 + orow) * H + n0 + 4 * c4) = v;
}


// Critic output layers of the three critic passes + losses' per-row terms + backward through those output layers.
struct CriticHeadArgs {
  const float *c2, *d2, *e2;      // last hidden activations: main critic(u), main critic(pi), target critic   [B,H]
  const float* WoutQ; const float* boutQ;          // main/Q output layer
  const float* WoutQt; const float* boutQt;        // target/Q output layer
  const float* r; int32_t ldr;
  const float* pi; int32_t ldpi;
  int32_t B, H, U;
  int32_t Bl;                     // rows per (virtual) rank (curious_net_cfg_t.loss_rows): divisor of the loss means
  float gamma, clip_lo, clip_hi, max_u;
  float *dc2, *dd2;               // gradients wrt c2 / d2                                                      [B,H]
  float* dQ;                      // [B] d Q_loss / d Q   (feeds dWout/dbout of main/Q)
  float* rows;                    // [3][B] per-row loss terms
  float* out_Qpi;                 // [B]
  int64_t* step_ctr;
};

__global__ __launch_bounds__(256) void critic_head_kernel(CriticHeadArgs a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + wave;
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.step_ctr) *a.step_ctr += 1;
  if (m >= a.B) return;
  const float* c2 = a.c2 + (int64_t)m * a.H;
  const float* d2 = a.d2 + (int64_t)m * a.H;
  const float* e2 = a.e2 + (int64_t)m * a.H;
  float q_[MAX_U], qp_[MAX_U], qt_[MAX_U];
  row_dot_fast<1>(c2, a.WoutQ, a.H, lane, q_);
  row_dot_fast<1>(d2, a.WoutQ, a.H, lane, qp_);
  row_dot_fast<1>(e2, a.WoutQt, a.H, lane, qt_);
  const float Q = q_[0] + a.boutQ[0], Qpi = qp_[0] + a.boutQ[0], Qt = qt_[0] + a.boutQt[0];
  const float invB = 1.0f / (float)a.Bl;
  const float target = fclip(a.r[(int64_t)m * a.ldr] + a.gamma * Qt, a.clip_lo, a.clip_hi);   // ddpg.py:437-438
  const float diff = target - Q;
  const float dQ = -2.0f * invB * diff;                      // d mean((target-Q)^2) / dQ
  const float dQpi = -invB;                                  // d (-mean(Q_pi)) / dQ_pi
  if (lane == 0) {
    float l2 = 0.f;
    for (int jj = 0; jj < a.U; ++jj) {
      float t = a.pi[(int64_t)m * a.ldpi + jj] / a.max_u;
      l2 += t * t;                                           // ddpg.py:441
    }
    a.rows[m] = diff * diff;                                 // ddpg.py:439
    a.rows[a.B + m] = Qpi;                                   // ddpg.py:440
    a.rows[2 * a.B + m] = l2;
    a.dQ[m] = dQ;
    a.out_Qpi[m] = Qpi;
  }
  // backward through the (shared) output layer: dH = dOut * Wout^T, masked by relu'
  const int trips = (a.H + 255) >> 8;
  for (int t = 0; t < trips; ++t) {
    const int k = 256 * t + 4 * lane;
    const int kc = min(k, a.H - 4);
    f32x4 w = ldv(a.WoutQ + kc);
    f32x4 hc = ldv(c2 + kc), hd = ldv(d2 + kc);
    f32x4 gc, gd;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      gc[e] = (hc[e] > 0.f) ? dQ * w[e] : 0.f;
      gd[e] = (hd[e] > 0.f) ? dQpi * w[e] : 0.f;
    }
    if (k < a.H) {
      *reinterpret_cast<f32x4*>(a.dc2 + (int64_t)m * a.H + k) = gc;
      *reinterpret_cast<f32x4*>(a.dd2 + (int64_t)m * a.H + k) = gd;
    }
  }
}


// ------------------------------------------------------------------ fused: critic heads + losses + first backward level
// critic_head_kernel + dx_hot_kernel(level nl-1) in one launch (saves one ~4.5 us dependent stage per update).
// Every workgroup recomputes, for its 16 batch rows, what it needs of the output layers (Q, target Q -> dQ; the
// critic(pi) pass needs no head at all: dQ_pi = -1/B), builds the A operand dY[m][n] = dOut[m] * Wout[n] * relu'(h[m][n])
// on the fly and runs the usual split-K tile.  Column-tile 0 also writes what later kernels read: dY itself (for the
// weight gradients), dQ, the per-row loss terms and Q_pi.
struct DxCritArgs {
  const float* partQ; const float* partQt; const float* partQpi;   // PART: [4][B] column-tile partials of the heads
  const float* hl[2];      // last hidden activations: critic(u), critic(pi)            [B,H]
  const float* hprev[2];   // activations one layer below (relu mask of the result)       [B,H]
  float* dY[2];            // gradient wrt hl (written by column-tile 0)                  [B,H]
  float* dX[2];            // gradient wrt hprev's pre-activation                         [B,H]
  const float* W;          // main/Q kernel of layer nl-1                                 [H,H]
  const float* WoutQ; const float* boutQ;
  const float* e2; const float* WoutQt; const float* boutQt;
  const float* r; int32_t ldr;
  const float* pi; int32_t ldpi;
  int32_t B, H, U;
  int32_t Bl;                     // rows per (virtual) rank: divisor of the loss means
  float gamma, clip_lo, clip_hi, max_u;
  float* dQ; float* rows; float* out_Qpi; int64_t* step_ctr;
};

__device__ inline float dot_row(const float* a, const float* b, int H, int lane) {
  float acc = 0.f;
  for (int k = 4 * lane; k < H; k += 256) {
    f32x4 x = ldv(a + k), y = ldv(b + k);
    acc += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
  }
  return wave_sum(acc);
}

// H == 256 only (one 16-byte fragment per lane covers a row): every global load of the kernel -- the 4 rows of the
// prologue, the output-layer weights and the main loop's 24 fragments -- is issued in one batch before any use.
template <bool PART, bool EX>
__global__ __launch_bounds__(256) void dx_crit_kernel(DxCritArgs a, Ex ex) {
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];
  __shared__ float s_dq[16];
  int64_t eo;
  const int ch = ex_decode<EX>(ex, blockIdx.z, eo);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.y * 16, k0 = blockIdx.x * 64;
  const int H = 256;
  if (blockIdx.x == 0 && blockIdx.y == 0 && ch == 0 && tid == 0 && a.step_ctr) *ex_i64(a.step_ctr, eo) += 1;
  const float* hrow = a.hl[ch] + eo + (int64_t)(m0 + j) * H;
  const float* wr = a.W + eo + (int64_t)(k0 + 4 * j) * H;
  const float invB = 1.0f / (float)a.Bl;
  const bool need_dots = (ch == 0) || (blockIdx.x == 0);
  // ---- all loads
  f32x4 pr_h[4], pr_e[4];                                   // prologue rows m0 + 4*wave + r, this lane's 4 columns
  const int pm = m0 + 4 * wave;
  f32x4 wq = zero4(), wt = zero4();
  const float bq = a.boutQ[eo], bt = a.boutQt[eo];
  float rew[4], l2v[4];
  float pq[4] = {0.f, 0.f, 0.f, 0.f}, pt[4] = {0.f, 0.f, 0.f, 0.f}, rew_j = 0.f;
  f32x4 pi_j = zero4();
  if (PART) {
    // every lane finishes the heads of its own row m0 + j from the 4 column-tile partials
    const float* p1 = (ch == 0) ? a.partQ : a.partQpi;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      pq[t] = p1[eo + (int64_t)t * a.B + m0 + j];
      pt[t] = a.partQt[eo + (int64_t)t * a.B + m0 + j];
    }
    rew_j = a.r[eo + (int64_t)(m0 + j) * a.ldr];
    pi_j = ldv(a.pi + eo + (int64_t)(m0 + j) * 4);
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      pr_h[r] = ldv(a.hl[ch] + eo + (int64_t)(pm + r) * H + 4 * lane);
      pr_e[r] = ldv(a.e2 + eo + (int64_t)(pm + r) * H + 4 * lane);
    }
    wq = ldv(a.WoutQ + eo + 4 * lane); wt = ldv(a.WoutQt + eo + 4 * lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      rew[r] = a.r[eo + (int64_t)(pm + r) * a.ldr];
      // sum_j (pi_j / max_u)^2 of row pm + r: lanes 0..U-1 hold one term each (ddpg.py:441)
      const float pv = (lane < a.U) ? a.pi[eo + (int64_t)(pm + r) * a.ldpi + lane] : 0.f;
      l2v[r] = (ch == 1) ? pv : 0.f;
    }
  }
  const int64_t o = eo + (int64_t)(m0 + (tid >> 4)) * H + k0 + 4 * (tid & 15);
  const f32x4 hm = ldv(a.hprev[ch] + o);
  f32x4 hv[4], wo[4], b[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int nq = (wave + 4 * u) * 16 + 4 * q;
    hv[u] = ldv(hrow + nq);
    wo[u] = ldv(a.WoutQ + eo + nq);
#pragma unroll
    for (int e = 0; e < 4; ++e) b[u][e] = ldv(wr + (int64_t)e * H + nq);
  }
  LOADS_FIRST();
  float* const rows_ = a.rows + eo; float* const dQ_ = a.dQ + eo; float* const Qpi_ = a.out_Qpi + eo;
  float dq;
  if (PART) {
    const bool writer = blockIdx.x == 0 && wave == 0 && q == 0;        // lanes 0..15 <-> rows m0 + j
    const int m = m0 + j;
    const float d1 = ((pq[0] + pq[1]) + (pq[2] + pq[3]));
    if (ch == 0) {
      const float Q = d1 + bq, Qt = ((pt[0] + pt[1]) + (pt[2] + pt[3])) + bt;
      const float target = fclip(rew_j + a.gamma * Qt, a.clip_lo, a.clip_hi);     // ddpg.py:437-438
      const float diff = target - Q;
      dq = -2.0f * invB * diff;
      if (writer) {
        rows_[m] = diff * diff;                               // ddpg.py:439
        dQ_[m] = dq;
      }
    } else {
      dq = -invB;                                              // d(-mean(Q_pi)) / dQ_pi
      if (writer) {
        const float Qpi = d1 + bq;
        float l2 = 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const float tt = pi_j[d] / a.max_u;
          l2 += tt * tt;                                       // ddpg.py:441
        }
        rows_[a.B + m] = Qpi;                                 // ddpg.py:440
        rows_[2 * a.B + m] = l2;
        Qpi_[m] = Qpi;
      }
    }
  } else {
  // ---- prologue: output-layer values of this wave's 4 rows
  if (need_dots) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = pm + r;
      float d1 = pr_h[r][0] * wq[0] + pr_h[r][1] * wq[1] + pr_h[r][2] * wq[2] + pr_h[r][3] * wq[3];
      float d2 = pr_e[r][0] * wt[0] + pr_e[r][1] * wt[1] + pr_e[r][2] * wt[2] + pr_e[r][3] * wt[3];
      d1 = wave_sum(d1);
      d2 = wave_sum(d2);
      if (ch == 0) {
        const float Q = d1 + bq, Qt = d2 + bt;
        const float target = fclip(rew[r] + a.gamma * Qt, a.clip_lo, a.clip_hi);   // ddpg.py:437-438
        const float diff = target - Q;
        const float dq = -2.0f * invB * diff;
        if (lane == 0) {
          s_dq[4 * wave + r] = dq;
          if (blockIdx.x == 0) {
            rows_[m] = diff * diff;                         // ddpg.py:439
            dQ_[m] = dq;
          }
        }
      } else {
        const float Qpi = d1 + bq;
        const float tt = l2v[r] / a.max_u;
        const float l2 = wave_sum(tt * tt);
        if (lane == 0) {
          rows_[a.B + m] = Qpi;                             // ddpg.py:440
          rows_[2 * a.B + m] = l2;
          Qpi_[m] = Qpi;
        }
      }
    }
  }
  if (ch == 1 && lane < 4) s_dq[4 * wave + lane] = -invB;    // d(-mean(Q_pi)) / dQ_pi
  __syncthreads();
  dq = s_dq[j];
  }
  f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    f32x4 av;
#pragma unroll
    for (int s = 0; s < 4; ++s) av[s] = (hv[u][s] > 0.f) ? dq * wo[u][s] : 0.f;
    if (blockIdx.x == 0) {
      const int nq = (wave + 4 * u) * 16 + 4 * q;
      *reinterpret_cast<f32x4*>(a.dY[ch] + eo + (int64_t)(m0 + j) * H + nq) = av;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = MFMA(av[s], b[u][e][s], acc[e]);
  }
  int orow, c4;
  f32x4 v = reduce_tile(red, acc, wave, q, j, tid, orow, c4);
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = (hm[e] > 0.f) ? v[e] : 0.f;
  *reinterpret_cast<f32x4*>(a.dX[ch] + o) = v;
}

// d pi_loss / dz through the critic's action slot, tanh and the l2 term; then backward through the actor output layer.
struct ActorDzArgs {
  const float* dd0;               // gradient wrt critic(pi) layer-0 pre-activation (already relu-masked)      [B,H]
  const float* Wu;                // rows of main/Q layer-0 kernel that multiply the action: [U, H]
  const float* pi; int32_t ldpi;
  const float* a2;                // actor last hidden activation                                               [B,H]
  const float* WoutPi;            // [H, U]
  float* dz;                      // [B, U]
  float* da2;                     // [B, H]
  int32_t B, H, U;
  float max_u, l2c;               // l2c = action_l2 * 2 / (max_u^2 * B * U)
};

// branch-free body for dimu == 4 (H % 4 == 0): every load is an unconditional 16-byte load
__device__ inline void actor_dz_fast4(const ActorDzArgs& a, int m, int lane) {
  const float* g = a.dd0 + (int64_t)m * a.H;
  const float* h = a.a2 + (int64_t)m * a.H;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const int trips = (a.H + 255) >> 8;
  for (int t = 0; t < trips; ++t) {
    const int k = 256 * t + 4 * lane;
    const int kc = min(k, a.H - 4);
    f32x4 g4 = sel4(k < a.H, ldv(g + kc));
    f32x4 w[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) w[d] = ldv(a.Wu + (int64_t)d * a.H + kc);
#pragma unroll
    for (int d = 0; d < 4; ++d) acc[d] += g4[0] * w[d][0] + g4[1] * w[d][1] + g4[2] * w[d][2] + g4[3] * w[d][3];
  }
  f32x4 pi4 = ldv(a.pi + (int64_t)m * a.ldpi);
  float dz[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    float v = wave_sum(acc[d]);
    float th = pi4[d] / a.max_u;
    float dpi = v / a.max_u + a.l2c * pi4[d];                // ddpg.py:440-441
    dz[d] = dpi * a.max_u * (1.0f - th * th);                // through pi = max_u * tanh(z)
  }
  if (lane == 0) {
    f32x4 o = {dz[0], dz[1], dz[2], dz[3]};
    *reinterpret_cast<f32x4*>(a.dz + (int64_t)m * 4) = o;
  }
  for (int t = 0; t < trips; ++t) {
    const int k = 256 * t + 4 * lane;
    const int kc = min(k, a.H - 4);
    f32x4 h4 = ldv(h + kc);
    f32x4 w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = ldv(a.WoutPi + (int64_t)(kc + e) * 4);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float sv = dz[0] * w[e][0] + dz[1] * w[e][1] + dz[2] * w[e][2] + dz[3] * w[e][3];
      o[e] = (h4[e] > 0.f) ? sv : 0.f;
    }
    if (k < a.H) *reinterpret_cast<f32x4*>(a.da2 + (int64_t)m * a.H + k) = o;
  }
}

__global__ __launch_bounds__(256) void actor_dz_kernel(ActorDzArgs a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + wave;
  if (m >= a.B) return;
  if (a.U == 4 && a.ldpi == 4 && ((((uintptr_t)a.Wu | (uintptr_t)a.WoutPi | (uintptr_t)a.pi | (uintptr_t)a.dz)) & 15) == 0) {
    actor_dz_fast4(a, m, lane);
    return;
  }
  const float* g = a.dd0 + (int64_t)m * a.H;
  float acc[MAX_U];
#pragma unroll
  for (int d = 0; d < MAX_U; ++d) acc[d] = 0.f;
  for (int k = 4 * lane; k < a.H; k += 256) {
    f32x4 g4 = ldg4(g + k, a.H - k, true);
#pragma unroll
    for (int d = 0; d < MAX_U; ++d)
      if (d < a.U) {
        f32x4 w = ldg4(a.Wu + (int64_t)d * a.H + k, a.H - k, true);
        acc[d] += g4[0] * w[0] + g4[1] * w[1] + g4[2] * w[2] + g4[3] * w[3];
      }
  }
  float dz[MAX_U];
#pragma unroll
  for (int d = 0; d < MAX_U; ++d) {
    dz[d] = 0.f;
    if (d < a.U) {
      float v = wave_sum(acc[d]);
      float pi = a.pi[(int64_t)m * a.ldpi + d];
      float th = pi / a.max_u;
      float dpi = v / a.max_u + a.l2c * pi;                  // ddpg.py:440-441
      dz[d] = dpi * a.max_u * (1.0f - th * th);              // through pi = max_u * tanh(z)
    }
  }
  if (lane < a.U) {
    float v = 0.f;
#pragma unroll
    for (int d = 0; d < MAX_U; ++d)
      if (d == lane) v = dz[d];
    a.dz[(int64_t)m * a.U + lane] = v;
  }
  const float* h = a.a2 + (int64_t)m * a.H;
  for (int k = 4 * lane; k < a.H; k += 256) {
    f32x4 h4 = ldg4(h + k, a.H - k, true);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float s = 0.f;
      if (k + e < a.H) {
        const float* wr = a.WoutPi + (int64_t)(k + e) * a.U;
#pragma unroll
        for (int d = 0; d < MAX_U; ++d)
          if (d < a.U) s += dz[d] * wr[d];
      }
      o[e] = (h4[e] > 0.f) ? s : 0.f;
    }
    if (k + 3 < a.H) {
      *reinterpret_cast<f32x4*>(a.da2 + (int64_t)m * a.H + k) = o;
    } else {
      for (int e = 0; e < 4 && k + e < a.H; ++e) a.da2[(int64_t)m * a.H + k + e] = o[e];
    }
  }
}


// actor_dz_kernel + dx_hot_kernel(actor level nl-1) in one launch, same idea as dx_crit_kernel: every workgroup
// recomputes dz for its 16 batch rows (a wavefront per row, 4 rows per wave: the arithmetic and its order are those of
// actor_dz_fast4, so the results are bit-identical), builds the A operand
//   da2[m][n] = (sum_d dz[m][d] * WoutPi[n][d]) * relu'(a2[m][n])
// on the fly and runs the split-K tile against main/pi's layer nl-1 kernel.  Column-tile 0 writes dz and da2, which
// the weight-gradient launch reads.  H == 256, dimu == 4.
struct DxActorArgs {
  const float* part;       // PART: [4][B][4] column-tile partials of dd0 . Wu^T
  const float* dd0; const float* Wu; const float* pi; const float* a2; const float* WoutPi;
  const float* hprev;      // actor activations one layer below a2 (relu mask of the result)   [B,H]
  const float* W;          // main/pi kernel of layer nl-1                                     [H,H]
  float* dz; float* da2; float* dX;
  int32_t B;
  float max_u, l2c;
};

template <bool PART, bool EX>
__global__ __launch_bounds__(256) void dx_actor_kernel(DxActorArgs a, Ex ex) {
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];
  __shared__ __attribute__((aligned(16))) float s_dz[16 * 4];
  int64_t eo;
  (void)ex_decode<EX>(ex, blockIdx.z, eo);                   // one problem per expert
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.y * 16, k0 = blockIdx.x * 64;
  const int H = 256;
  const int pm = m0 + 4 * wave;
  // ---- all loads
  f32x4 g4[4], wu[4];
  float pp[4] = {0.f, 0.f, 0.f, 0.f};
  float pim;
  if (PART) {
    // thread t < 64 finishes dz[m0 + t/4][t%4] from the 4 column-tile partials
#pragma unroll
    for (int t = 0; t < 4; ++t) pp[t] = a.part[eo + ((int64_t)t * a.B + m0) * 4 + (tid & 63)];
    pim = a.pi[eo + (int64_t)m0 * 4 + (tid & 63)];
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) g4[r] = ldv(a.dd0 + eo + (int64_t)(pm + r) * H + 4 * lane);
    pim = a.pi[eo + (int64_t)pm * 4 + (lane & 15)];                // lane 4r+d: pi[pm + r][d]
#pragma unroll
    for (int d = 0; d < 4; ++d) wu[d] = ldv(a.Wu + eo + (int64_t)d * H + 4 * lane);
  }
  const int64_t o = eo + (int64_t)(m0 + (tid >> 4)) * H + k0 + 4 * (tid & 15);
  const f32x4 hm = ldv(a.hprev + o);
  const float* hrow = a.a2 + eo + (int64_t)(m0 + j) * H;
  const float* wr = a.W + eo + (int64_t)(k0 + 4 * j) * H;
  f32x4 hv[4], wo[4][4], b[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int nq = (wave + 4 * u) * 16 + 4 * q;
    hv[u] = ldv(hrow + nq);
#pragma unroll
    for (int s = 0; s < 4; ++s) wo[u][s] = ldv(a.WoutPi + eo + (int64_t)(nq + s) * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) b[u][e] = ldv(wr + (int64_t)e * H + nq);
  }
  LOADS_FIRST();
  // ---- prologue: dz of the 16 rows
  if (PART) {
    if (tid < 64) {
      const float v = (pp[0] + pp[1]) + (pp[2] + pp[3]);
      const float th = pim / a.max_u;
      const float dpi = v / a.max_u + a.l2c * pim;              // ddpg.py:440-441
      const float dz = dpi * a.max_u * (1.0f - th * th);        // through pi = max_u * tanh(z)
      s_dz[tid] = dz;
      if (blockIdx.x == 0) a.dz[eo + (int64_t)m0 * 4 + tid] = dz;
    }
  } else {
    // lane 4r+d of wave w finishes dz[pm + r][d]
    float sums[16];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        float acc = 0.f;
        acc += g4[r][0] * wu[d][0] + g4[r][1] * wu[d][1] + g4[r][2] * wu[d][2] + g4[r][3] * wu[d][3];
        sums[4 * r + d] = wave_sum(acc);
      }
    const float v = pick16(sums, lane);
    if (lane < 16) {
      const float th = pim / a.max_u;
      const float dpi = v / a.max_u + a.l2c * pim;              // ddpg.py:440-441
      const float dz = dpi * a.max_u * (1.0f - th * th);        // through pi = max_u * tanh(z)
      s_dz[16 * wave + lane] = dz;
      if (blockIdx.x == 0) a.dz[eo + (int64_t)pm * 4 + lane] = dz;
    }
  }
  __syncthreads();
  const f32x4 dzr = *reinterpret_cast<const f32x4*>(s_dz + 4 * j);
  f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    f32x4 av;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float sv = dzr[0] * wo[u][s][0] + dzr[1] * wo[u][s][1] + dzr[2] * wo[u][s][2] + dzr[3] * wo[u][s][3];
      av[s] = (hv[u][s] > 0.f) ? sv : 0.f;
    }
    if (blockIdx.x == 0) {
      const int nq = (wave + 4 * u) * 16 + 4 * q;
      *reinterpret_cast<f32x4*>(a.da2 + eo + (int64_t)(m0 + j) * H + nq) = av;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = MFMA(av[s], b[u][e][s], acc[e]);
  }
  int orow, c4;
  f32x4 v = reduce_tile(red, acc, wave, q, j, tid, orow, c4);
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = (hm[e] > 0.f) ? v[e] : 0.f;
  *reinterpret_cast<f32x4*>(a.dX + o) = v;
}
