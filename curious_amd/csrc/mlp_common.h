// Descriptors of the generic kernels, MFMA / load helpers, DPP wavefront reductions (included by mlp.hip).
#pragma once


#define MAX_SEG 4
#define MAX_LAYERS 8
#define MAX_DW 24
#define MAX_U 8

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
  int32_t vec;           // 16-byte loads of x legal
};

struct FwdProb {
  Seg seg[MAX_SEG];
  int32_t nseg;
  const float* bias;
  float* Y;
  int32_t ldy, M, N;
  int32_t act;           // 0 linear, 1 relu
  int32_t wvec;          // 16-byte loads of W rows legal (N % 4 == 0, aligned)
  int32_t fast;          // every segment 16-byte loadable with w % 4 == 0, no `sub`: branch-free main loop
};

struct DxProb {          // dX[M,K] = (dY[M,N] . W[K,N]^T) (.) relu'(H)
  const float* dY; int32_t lddy;
  const float* W;  int32_t ldw;
  const float* H;  int32_t ldh;
  float* dX; int32_t lddx;
  int32_t M, N, K;
  int32_t vec;
  int32_t fast;          // vec && N % 4 == 0: branch-free main loop
};

struct DwProb {          // dW[w,N] = X[M,w]^T . dY[M,N];  db[N] = colsum(dY)
  Seg x;
  const float* dY; int32_t lddy;
  float* dW;
  float* db;             // nullable
  int32_t M, N;
  int32_t yvec;          // 16-byte loads of dY rows legal
  int32_t fast;          // yvec && no `sub` on x: branch-free main loop
};

struct FwdArgs { FwdProb p[3]; int32_t nprob; };
struct DxArgs { DxProb p[2]; int32_t nprob; };
struct LossFin {         // final, fixed-order reduction of the per-row loss terms (rides on the dW launch)
  const float* rows;     // [3][B]: (target-Q)^2, Q_pi, sum_j (pi_j/max_u)^2
  float* out;            // [B / Bl][2]: Q_loss, pi_loss of every rank
  int32_t B, U;
  int32_t Bl;            // rows per (virtual) rank (curious_net_cfg_t.loss_rows; == B for one rank): every block of Bl rows
                         // has its own means
  float action_l2;
  int64_t* step_ctr;     // non-NULL: the update's increment of the step counter happens HERE (deferred from the gradient
                         // launch, whose gather blocks read the counter: mlp_rows.h)
  const int32_t* fault;  // gradients-only calls (several ranks): the workspace's fault word and the padding element of the
  float* flag;           // gradient vector that carries it through the all-reduce (curious_transposed_t.fault_flag), or NULL
};
__device__ inline void loss_fin_flag(const LossFin& F, const int64_t eo, const int64_t eg) {
  if (F.flag) F.flag[eg] = (*reinterpret_cast<const int32_t*>(reinterpret_cast<const float*>(F.fault) + eo) != 0) ? 1.0f : 0.0f;
}
struct DwArgs { DwProb p[MAX_DW]; int32_t nprob; LossFin fin; };

__device__ inline float seg_xform(const Seg& s, float v, int row, int col) {
  if (s.sub) v = __fsub_rn(v, s.sub[(int64_t)row * s.ldsub + col]);
  if (s.clip > 0.0f) v = fclip(v, -s.clip, s.clip);
  if (s.mean) v = fclip(fdiv(__fsub_rn(v, s.mean[col]), s.stdv[col]), -s.nclip, s.nclip);
  if (s.div != 1.0f) v = fdiv(v, s.div);
  return v;
}

__device__ inline f32x4 zero4() {
  f32x4 z = {0.f, 0.f, 0.f, 0.f};
  return z;
}

// four consecutive floats p[0..3]; element e is valid when e < nvalid; out-of-range -> 0
__device__ inline f32x4 ldg4(const float* p, int nvalid, bool vec) {
  f32x4 v = zero4();
  if (nvalid <= 0) return v;
  if (vec && nvalid >= 4) return *reinterpret_cast<const f32x4*>(p);
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (e < nvalid) v[e] = p[e];
  return v;
}

// four consecutive columns (col .. col+3) of row `row` of a segment, with its input transforms
__device__ inline f32x4 seg_load4(const Seg& s, int row, int col, bool row_ok) {
  if (!row_ok) return zero4();
  f32x4 v = ldg4(s.x + (int64_t)row * s.ld + col, s.w - col, s.vec != 0);
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

// ------------------------------------------------------------------ batched experts (structure='task_experts')
// N agents with identical shapes go through ONE launch sequence (curious_ddpg_update_experts): every piece of
// per-expert state -- parameters, target, gradient, Adam moments, step counter and step-size ring, workspace, staged
// batches, sampling tables, loss outputs -- sits at the same offset of a per-expert slab, the slabs are `stride`
// floats apart, so expert e's pointer is the expert-0 pointer of the descriptor + e * stride.  Only the replay
// storage is shared.  Kernels are instantiated twice: EX = false is the single-agent path (no extra scalar work, the
// descriptor index is blockIdx.z itself); EX = true decodes blockIdx.z = e * nprob + problem.
struct Ex {
  int64_t stride;          // floats between the slabs of consecutive experts
  int32_t nprob;           // problems per expert in this launch (grid.z = n_experts * nprob)
  uint32_t zmul;           // ceil(65536 / nprob): e = (z * zmul) >> 16 (exact for z < 8192)
};
template <bool EX>
__device__ __forceinline__ int ex_decode(const Ex& ex, int z, int64_t& eo) {
  if (!EX) { eo = 0; return z; }
  const int e = (int)(((uint32_t)z * ex.zmul) >> 16);
  eo = (int64_t)e * ex.stride;
  return z - e * ex.nprob;
}
__device__ __forceinline__ int64_t* ex_i64(int64_t* p, int64_t eo) {      // an int64 living in the float slab
  return reinterpret_cast<int64_t*>(reinterpret_cast<float*>(p) + eo);
}
__device__ __forceinline__ const int64_t* ex_i64(const int64_t* p, int64_t eo) {
  return reinterpret_cast<const int64_t*>(reinterpret_cast<const float*>(p) + eo);
}

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// Scheduling fence: everything above (the loads of a wave's whole K share) is issued before anything below (the
// MFMAs).  Without it hipcc -O3 interleaves load -> s_waitcnt -> 4 MFMA per fragment to save registers (27-40 VGPRs)
// and exposes the full L2/Infinity-Cache latency sixteen times per wave (measured: 7.6 us vs 4 us per layer kernel).
#define LOADS_FIRST() __builtin_amdgcn_sched_barrier(0)

// Unconditional 16-byte load.  The fast paths below never branch around a load: addresses are clamped into range and
// invalid contributions are zeroed with selects afterwards, so that hipcc can issue every load of a wave's share
// before the first MFMA instead of waiting vmcnt(0) per guarded element (cdna_hip_programming.md 5, trap (c)).
__device__ inline f32x4 ldv(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ inline f32x4 sel4(bool ok, f32x4 v) {
  f32x4 r;
  r[0] = ok ? v[0] : 0.f; r[1] = ok ? v[1] : 0.f; r[2] = ok ? v[2] : 0.f; r[3] = ok ? v[3] : 0.f;
  return r;
}
// segment transforms without the relative-goal subtraction (clip, normalise, divide), on a whole float4
__device__ inline f32x4 seg_xform4(const Seg& s, f32x4 v, int col) {
  if (s.clip > 0.0f) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fclip(v[e], -s.clip, s.clip);
  }
  if (s.mean) {
    f32x4 mu = ldv(s.mean + col), sd = ldv(s.stdv + col);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fclip(fdiv(__fsub_rn(v[e], mu[e]), sd[e]), -s.nclip, s.nclip);
  }
  if (s.div != 1.0f) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fdiv(v[e], s.div);
  }
  return v;
}

// Partial 16x64 tiles of the 4 waves -> LDS -> summed tile.  acc[e][r] is element (row 4q+r, column 4j+e).
// Returns the reduced float4 (columns 4*c4 .. 4*c4+3 of row `orow`) owned by this thread.
__device__ inline f32x4 reduce_tile(float* red, const f32x4 acc[4], int wave, int q, int j, int tid, int& orow,
                                    int& c4) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    f32x4 v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
    *reinterpret_cast<f32x4*>(red + ((wave * 16 + 4 * q + r) * 64 + 4 * j)) = v;
  }
  __syncthreads();
  orow = tid >> 4;
  c4 = tid & 15;
  f32x4 s = *reinterpret_cast<const f32x4*>(red + ((0 * 16 + orow) * 64 + 4 * c4));
#pragma unroll
  for (int w = 1; w < 4; ++w) {
    f32x4 t = *reinterpret_cast<const f32x4*>(red + ((w * 16 + orow) * 64 + 4 * c4));
    s += t;
  }
  return s;
}

// Sum over the 64 lanes of a wavefront, result uniform.  Four DPP steps (quad xor 1, quad xor 2, half-row mirror, row
// mirror) leave the sum of each 16-lane row in all its lanes -- plain VALU moves, no LDS crossbar round trips; the four
// row sums are then read as scalars.  (The ds_bpermute butterfly this replaces cost ~30 ns per dependent step; the
// fused prologues below run 16 of these reductions.)
template <int CTRL>
__device__ inline float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);     // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);     // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);    // row_half_mirror
  v += dpp_mov<0x140>(v);    // row_mirror
  return v;
}
__device__ inline float lane_read(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  return (lane_read(v, 0) + lane_read(v, 16)) + (lane_read(v, 32) + lane_read(v, 48));
}
// lane i < 16 gets vals[i] (uniform inputs): lets ONE lane per value do the expensive scalar math of a prologue
__device__ __forceinline__ float pick16(const float (&vals)[16], int lane) {
  float m = vals[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) m = (lane == i) ? vals[i] : m;
  return m;
}

// Ranks beyond the first: one wave per rank, each lane sums its rows in ascending order, then the wave-wide sum.
// (The first rank keeps the workgroup-wide tree of the one-rank form -- see the callers --, so a batch of ONE rank is
// finalised exactly as before; the per-rank losses are outputs only, no gradient depends on their order of summation.)
__device__ inline void loss_fin_ranks(const LossFin& F, const float* rows, float* out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int R = F.B / F.Bl;
  for (int r = 1 + wave; r < R; r += 4) {
    float lq = 0.f, lp = 0.f, ll = 0.f;
    for (int i = lane; i < F.Bl; i += 64) {
      const int m = r * F.Bl + i;
      lq += rows[m]; lp += rows[F.B + m]; ll += rows[2 * F.B + m];
    }
    lq = wave_sum(lq); lp = wave_sum(lp); ll = wave_sum(ll);
    if (lane == 0) {
      const float invB = 1.0f / (float)F.Bl;
      out[2 * r] = lq * invB;
      out[2 * r + 1] = -lp * invB + F.action_l2 * ll / (float)(F.Bl * F.U);
    }
  }
}
