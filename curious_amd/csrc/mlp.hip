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
  float* out;            // [2]: Q_loss, pi_loss
  int32_t B, U;
  float action_l2;
};
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

// ------------------------------------------------------------------ forward layer
// Y[M,N] = act(sum_seg X_seg . W_seg + bias).  grid: x = ceil(N/64), y = ceil(M/16), z = problem
__global__ __launch_bounds__(256) void fwd_layer_kernel(FwdArgs args) {
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];
  const FwdProb& P = args.p[blockIdx.z];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int j = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 64;
  if (m0 >= P.M || n0 >= P.N) return;                       // uniform per workgroup
  const int row = m0 + j;
  const bool row_ok = row < P.M;
  const int col = n0 + 4 * j;                               // this lane's 4 output columns: col + e
  f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
  int ci = 0;                                               // running chunk index over the virtual K
  if (P.fast) {
    const int rowc = min(row, P.M - 1), colc = min(col, P.N - 4);
    for (int sidx = 0; sidx < P.nseg; ++sidx) {
      const Seg& S = P.seg[sidx];
      const int nch = (S.w + 15) >> 4;
      const float* xr = S.x + (int64_t)rowc * S.ld;
      for (int c0 = ((wave - ci) & 3); c0 < nch; c0 += 16) {
        f32x4 a[4], b[4][4];
        int kqs[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int kq = (c0 + 4 * u) * 16 + 4 * q;
          ok[u] = row_ok && (kq < S.w);
          kqs[u] = min(kq, S.w - 4);
          a[u] = ldv(xr + kqs[u]);
#pragma unroll
          for (int s = 0; s < 4; ++s) b[u][s] = ldv(S.W + (int64_t)(kqs[u] + s) * P.N + colc);
        }
        LOADS_FIRST();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          a[u] = sel4(ok[u], seg_xform4(S, a[u], kqs[u]));
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = MFMA(a[u][s], b[u][s][e], acc[e]);
        }
      }
      ci += nch;
    }
  } else {
    for (int sidx = 0; sidx < P.nseg; ++sidx) {
      const Seg& S = P.seg[sidx];
      const int nch = (S.w + 15) >> 4;
      // this wave's chunks of the segment: (ci + c) % 4 == wave; up to 4 chunks are loaded before any MFMA
      for (int c0 = ((wave - ci) & 3); c0 < nch; c0 += 16) {
        f32x4 a[4], b[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int kq = (c0 + 4 * u) * 16 + 4 * q;
          const bool cok = (c0 + 4 * u) < nch;
          a[u] = cok ? seg_load4(S, row, kq, row_ok) : zero4();
#pragma unroll
          for (int s = 0; s < 4; ++s)
            b[u][s] = (cok && kq + s < S.w) ? ldg4(S.W + (int64_t)(kq + s) * P.N + col, P.N - col, P.wvec != 0)
                                            : zero4();
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = MFMA(a[u][s], b[u][s][e], acc[e]);
      }
      ci += nch;
    }
  }
  int orow, c4;
  f32x4 v = reduce_tile(red, acc, wave, q, j, tid, orow, c4);
  const int grow = m0 + orow, gcol = n0 + 4 * c4;
  if (grow >= P.M || gcol >= P.N) return;
  f32x4 bias = P.bias ? ldg4(P.bias + gcol, P.N - gcol, P.wvec != 0) : zero4();
  v += bias;
  if (P.act == 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
  }
  float* dst = P.Y + (int64_t)grow * P.ldy + gcol;
  if (gcol + 3 < P.N && (P.ldy & 3) == 0) {
    *reinterpret_cast<f32x4*>(dst) = v;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (gcol + e < P.N) dst[e] = v[e];
  }
}

// ------------------------------------------------------------------ backward: input gradient
// dX[m][k] = sum_n dY[m][n] W[k][n], masked by relu'(H).  grid: x = ceil(K/64), y = ceil(M/16), z = problem
__global__ __launch_bounds__(256) void dx_kernel(DxArgs args) {
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];
  const DxProb& P = args.p[blockIdx.z];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int j = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.y * 16, k0 = blockIdx.x * 64;
  if (m0 >= P.M || k0 >= P.K) return;
  const int row = m0 + j;
  const bool row_ok = row < P.M;
  const int kc = k0 + 4 * j;                                 // output columns kc + e  <->  weight rows kc + e
  f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
  const int nch = (P.N + 15) >> 4;
  if (P.fast) {
    const float* dyr = P.dY + (int64_t)min(row, P.M - 1) * P.lddy;
    const float* wr[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) wr[e] = P.W + (int64_t)min(kc + e, P.K - 1) * P.ldw;
    for (int c0 = wave; c0 < nch; c0 += 16) {
      f32x4 a[4], b[4][4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int nq = (c0 + 4 * u) * 16 + 4 * q;
        ok[u] = row_ok && (nq < P.N);
        const int nqc = min(nq, P.N - 4);
        a[u] = ldv(dyr + nqc);
#pragma unroll
        for (int e = 0; e < 4; ++e) b[u][e] = ldv(wr[e] + nqc);
      }
      LOADS_FIRST();
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = sel4(ok[u], a[u]);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = MFMA(a[u][s], b[u][e][s], acc[e]);
      }
    }
  } else {
    const float* dyr = P.dY + (int64_t)row * P.lddy;
    for (int c0 = wave; c0 < nch; c0 += 16) {
      f32x4 a[4], b[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int nq = (c0 + 4 * u) * 16 + 4 * q;
        const bool cok = (c0 + 4 * u) < nch;
        a[u] = (cok && row_ok) ? ldg4(dyr + nq, P.N - nq, P.vec != 0) : zero4();
#pragma unroll
        for (int e = 0; e < 4; ++e)
          b[u][e] = (cok && kc + e < P.K) ? ldg4(P.W + (int64_t)(kc + e) * P.ldw + nq, P.N - nq, P.vec != 0)
                                          : zero4();
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = MFMA(a[u][s], b[u][e][s], acc[e]);
    }
  }
  int orow, c4;
  f32x4 v = reduce_tile(red, acc, wave, q, j, tid, orow, c4);
  const int grow = m0 + orow, gcol = k0 + 4 * c4;
  if (grow >= P.M || gcol >= P.K) return;
  if (P.H) {
    f32x4 h = ldg4(P.H + (int64_t)grow * P.ldh + gcol, P.K - gcol, (P.ldh & 3) == 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (h[e] > 0.f) ? v[e] : 0.f;
  }
  float* dst = P.dX + (int64_t)grow * P.lddx + gcol;
  if (gcol + 3 < P.K && (P.lddx & 3) == 0) {
    *reinterpret_cast<f32x4*>(dst) = v;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (gcol + e < P.K) dst[e] = v[e];
  }
}

// ------------------------------------------------------------------ backward: weight gradient (grouped)
// dW[k][n] = sum_m X[m][k] dY[m][n]; db[n] = sum_m dY[m][n].  grid: x = ceil(N/64), y = ceil(w/16), z = problem;
// slice z == nprob finalises the losses.
__global__ __launch_bounds__(256) void dw_kernel(DwArgs args) {
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if ((int)blockIdx.z == args.nprob) {
    // losses (ddpg.py:439-441) from the per-row terms, summed in a fixed order by one workgroup
    if (blockIdx.x != 0 || blockIdx.y != 0) return;
    const LossFin& F = args.fin;
    float lq = 0.f, lp = 0.f, ll = 0.f;
    for (int m = tid; m < F.B; m += 256) {
      lq += F.rows[m];
      lp += F.rows[F.B + m];
      ll += F.rows[2 * F.B + m];
    }
    red[tid] = lq; red[256 + tid] = lp; red[512 + tid] = ll;
    __syncthreads();
    for (int h = 128; h >= 1; h >>= 1) {
      if (tid < h) {
        red[tid] += red[tid + h];
        red[256 + tid] += red[256 + tid + h];
        red[512 + tid] += red[512 + tid + h];
      }
      __syncthreads();
    }
    if (tid == 0) {
      const float invB = 1.0f / (float)F.B;
      F.out[0] = red[0] * invB;
      F.out[1] = -red[256] * invB + F.action_l2 * red[512] / (float)(F.B * F.U);
    }
    return;
  }
  const DwProb& P = args.p[blockIdx.z];
  const int j = lane & 15, q = lane >> 4;
  const int k0 = blockIdx.y * 16, n0 = blockIdx.x * 64;
  if (k0 >= P.x.w || n0 >= P.N) return;
  const int krow = k0 + j;                                   // A operand row index = weight row
  const bool k_ok = krow < P.x.w;
  const int col = n0 + 4 * j;
  f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
  f32x4 bsum = zero4();
  const int nch = (P.M + 15) >> 4;
  if (P.fast) {
    const Seg& X = P.x;
    const int krc = min(krow, X.w - 1), colc = min(col, P.N - 4);
    const bool plain = X.clip <= 0.0f && !X.mean && X.div == 1.0f;
    const float mu = X.mean ? X.mean[krc] : 0.f, sd = X.mean ? X.stdv[krc] : 1.f;
    for (int c0 = wave; c0 < nch; c0 += 16) {
      float a[4][4];
      f32x4 b[4][4];
      bool ok[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int mq = (c0 + 4 * u) * 16 + 4 * q;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          ok[u][s] = (mq + s) < P.M;
          const int mc = min(mq + s, P.M - 1);
          a[u][s] = X.x[(int64_t)mc * X.ld + krc];
          b[u][s] = ldv(P.dY + (int64_t)mc * P.lddy + colc);
        }
      }
      LOADS_FIRST();
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          float av = a[u][s];
          if (!plain) {
            if (X.clip > 0.0f) av = fclip(av, -X.clip, X.clip);
            if (X.mean) av = fclip(fdiv(__fsub_rn(av, mu), sd), -X.nclip, X.nclip);
            if (X.div != 1.0f) av = fdiv(av, X.div);
          }
          av = (ok[u][s] && k_ok) ? av : 0.f;
          f32x4 bv = sel4(ok[u][s], b[u][s]);
          bsum += bv;
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = MFMA(av, bv[e], acc[e]);
        }
    }
  } else {
    for (int c0 = wave; c0 < nch; c0 += 16) {
      float a[4][4];
      f32x4 b[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int mq = (c0 + 4 * u) * 16 + 4 * q;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const bool mok = ((c0 + 4 * u) < nch) && (mq + s < P.M);
          a[u][s] = seg_load1(P.x, mq + s, krow, mok && k_ok);
          b[u][s] = mok ? ldg4(P.dY + (int64_t)(mq + s) * P.lddy + col, P.N - col, P.yvec != 0) : zero4();
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          bsum += b[u][s];
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = MFMA(a[u][s], b[u][s][e], acc[e]);
        }
    }
  }
  int orow, c4;
  f32x4 v = reduce_tile(red, acc, wave, q, j, tid, orow, c4);
  const int grow = k0 + orow, gcol = n0 + 4 * c4;
  if (grow < P.x.w && gcol < P.N) {
    float* dst = P.dW + (int64_t)grow * P.N + gcol;
    if (gcol + 3 < P.N && (P.N & 3) == 0 && (((uintptr_t)P.dW) & 15) == 0) {
      *reinterpret_cast<f32x4*>(dst) = v;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (gcol + e < P.N) dst[e] = v[e];
    }
  }
  if (P.db && blockIdx.y == 0) {
    // bias gradient: every lane summed its (q, s, chunk) share of 4 columns; fold q-groups, then the 4 waves
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = bsum[e];
      t += __shfl_xor(t, 16);
      t += __shfl_xor(t, 32);
      bsum[e] = t;
    }
    __syncthreads();                                         // `red` is free again
    if (q == 0) *reinterpret_cast<f32x4*>(red + wave * 64 + 4 * j) = bsum;
    __syncthreads();
    if (tid < 64 && n0 + tid < P.N) P.db[n0 + tid] = red[tid] + red[64 + tid] + red[128 + tid] + red[192 + tid];
  }
}



// ================================================================== lean kernels for the hot 256-wide layers
// Adam applied where the gradient is produced (curious_ddpg_update, single-rank): the workgroup that finishes a tile
// of dW / db owns the matching elements of theta, m and v, so the optimiser needs no launch of its own.  Arithmetic and
// step-size lookup are those of optim.hip's adam_body (mpi_adam.py:29-35), bit for bit.
struct AdamFuse {
  float* theta; float* m; float* v;
  const float* grad;              // base of the gradient vector: (gradient pointer - grad) = parameter index
  int64_t n_Q;
  const float* alpha_tab; const int64_t* step_ctr; int64_t tab_base; int32_t tab_len;
  float a_Q, a_pi, b1, omb1, b2, omb2, eps;
};

__device__ inline void adam_alphas(const AdamFuse& A, float& aQ, float& aPi) {
  aQ = A.a_Q; aPi = A.a_pi;
  if (A.alpha_tab) {
    int64_t idx = ((*A.step_ctr) - 1 - A.tab_base) % A.tab_len;
    if (idx < 0) idx += A.tab_len;
    aQ = A.alpha_tab[2 * idx];
    aPi = A.alpha_tab[2 * idx + 1];
  }
}

__device__ inline float adam_elem(const AdamFuse& A, float na, float g, float& m, float& v, float th) {
  m = __fadd_rn(__fmul_rn(A.b1, m), __fmul_rn(A.omb1, g));                       // mpi_adam.py:31
  v = __fadd_rn(__fmul_rn(A.b2, v), __fmul_rn(A.omb2, __fmul_rn(g, g)));         // mpi_adam.py:32
  const float step = fdiv(__fmul_rn(na, m), __fadd_rn(sqrtf(v), A.eps));         // mpi_adam.py:33
  return __fadd_rn(th, step);                                                    // mpi_adam.py:34
}

struct AdamPre4 { f32x4 m, v, th; };
__device__ inline AdamPre4 adam_prefetch4(const AdamFuse& A, int64_t i) {
  AdamPre4 p;
  p.m = ldv(A.m + i); p.v = ldv(A.v + i); p.th = ldv(A.theta + i);
  return p;
}
__device__ inline void adam_apply4(const AdamFuse& A, float na, int64_t i, const f32x4& g, AdamPre4& p) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float m = p.m[e], v = p.v[e];
    p.th[e] = adam_elem(A, na, g[e], m, v, p.th[e]);
    p.m[e] = m; p.v[e] = v;
  }
  *reinterpret_cast<f32x4*>(A.m + i) = p.m;
  *reinterpret_cast<f32x4*>(A.v + i) = p.v;
  *reinterpret_cast<f32x4*>(A.theta + i) = p.th;
}
__device__ inline void adam_apply1(const AdamFuse& A, float na, int64_t i, float g) {
  float m = A.m[i], v = A.v[i];
  const float th = adam_elem(A, na, g, m, v, A.theta[i]);
  A.m[i] = m; A.v[i] = v; A.theta[i] = th;
}

// Same tiling as the generic kernels above, but with 56-byte problem descriptors, no bounds checks and no segment
// machinery: tools/gemm_lab.hip measures 3.7 us per launch inside a hipGraph for this form (2.0 us of which is the
// launch floor of an empty kernel) against 5.5-7 us for the generic form with its 1 KB kernarg.
// Preconditions (checked on the host, else the generic kernel runs): M % 16 == 0, N % 64 == 0, reduction dim % 256
// == 0, all pointers 16-byte aligned, leading dims % 4 == 0.
struct GemmHot {
  const float* A; const float* B; const float* aux; float* C; float* aux_out;
  int32_t lda, ldb, ldc, M, N, K;
  // optional epilogue (DOT kernels): partial products of the output tile with a narrow matrix that the NEXT launch
  // would otherwise have to contract over whole rows (output layers, the critic's action rows):
  //   dot_out[tile][m][d] = sum_{c in this 64-column tile} C[m][c] * w(c, d)
  // dot_mode 1: D = 1, w = dot_w[c];  2: D = 4, w = dot_w[c * 4 + d];  3: D = 4, w = dot_w[d * dot_ld + c]
  const float* dot_w; float* dot_out; int32_t dot_mode, dot_ld;
};
struct HotArgs { GemmHot p[3]; };

__device__ inline void hot_store(float* red, const f32x4 acc[4], int wave, int q, int j, int tid, f32x4& v, int& orow,
                                 int& c4) {
  v = reduce_tile(red, acc, wave, q, j, tid, orow, c4);
}

struct DotW { f32x4 w[4]; };
__device__ inline DotW dot_prefetch(const GemmHot& P, int c) {
  // branch-free (4 unconditional loads at selected addresses): a branch on dot_mode here would make every load that
  // follows in program order wait for the scalar load of dot_mode.  dot_w is a valid address for every problem of a
  // DOT launch (the host points it at the weight matrix when dot_mode == 0).
  DotW d;
  const int m = P.dot_mode;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int64_t off = (m == 2) ? (int64_t)(c + e) * 4 : (m == 3) ? (int64_t)e * P.dot_ld + c : (m == 1) ? c : 0;
    d.w[e] = ldv(P.dot_w + off);
  }
  return d;
}
// v: this thread's 4 consecutive output columns of row `row`; the 16 threads of a row are one DPP row
__device__ inline void dot_epilogue(const GemmHot& P, const DotW& d, const f32x4& v, int row, int tile, int c4) {
  if (P.dot_mode == 0) return;
  f32x4 pd = zero4();
  if (P.dot_mode == 1) {
    pd[0] = v[0] * d.w[0][0] + v[1] * d.w[0][1] + v[2] * d.w[0][2] + v[3] * d.w[0][3];
    pd[0] = row16_sum(pd[0]);
    if (c4 == 0) P.dot_out[(int64_t)tile * P.M + row] = pd[0];
    return;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float t = 0.f;
    if (P.dot_mode == 2) t = v[0] * d.w[0][k] + v[1] * d.w[1][k] + v[2] * d.w[2][k] + v[3] * d.w[3][k];
    else t = v[0] * d.w[k][0] + v[1] * d.w[k][1] + v[2] * d.w[k][2] + v[3] * d.w[k][3];
    pd[k] = row16_sum(t);
  }
  if (c4 == 0) *reinterpret_cast<f32x4*>(P.dot_out + ((int64_t)tile * P.M + row) * 4) = pd;
}

// C[M,N] = relu(A[M,K] . B[K,N] + bias)        grid (N/64, M/16, nprob)
template <bool DOT>
__global__ __launch_bounds__(256) void fwd_hot_kernel(HotArgs args) {
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];
  const GemmHot& P = args.p[blockIdx.z];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 64;
  const float* xr = P.A + (int64_t)(m0 + j) * P.lda;
  const float* wc = P.B + n0 + 4 * j;
  const f32x4 bias = ldv(P.aux + n0 + 4 * (tid & 15));      // epilogue operand, issued with the first batch
  DotW dw;
  if (DOT) dw = dot_prefetch(P, n0 + 4 * (tid & 15));
  f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
  for (int kb = 0; kb < P.K; kb += 256) {
    f32x4 a[4], b[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int kq = kb + (wave + 4 * u) * 16 + 4 * q;
      a[u] = ldv(xr + kq);
#pragma unroll
      for (int s = 0; s < 4; ++s) b[u][s] = ldv(wc + (int64_t)(kq + s) * P.ldb);
    }
    LOADS_FIRST();
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = MFMA(a[u][s], b[u][s][e], acc[e]);
  }
  f32x4 v; int orow, c4;
  hot_store(red, acc, wave, q, j, tid, v, orow, c4);
  v += bias;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
  *reinterpret_cast<f32x4*>(P.C + (int64_t)(m0 + orow) * P.ldc + n0 + 4 * c4) = v;
  if (DOT) dot_epilogue(P, dw, v, m0 + orow, blockIdx.x, c4);
}

// C[M,K'] = (A[M,N] . B[K',N]^T) * relu'(aux[M,K'])    (K' = P.N output columns, reduction over P.K)   grid (K'/64, M/16, nprob)
template <bool DOT>
__global__ __launch_bounds__(256) void dx_hot_kernel(HotArgs args) {
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];
  const GemmHot& P = args.p[blockIdx.z];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.y * 16, k0 = blockIdx.x * 64;
  const float* dyr = P.A + (int64_t)(m0 + j) * P.lda;
  const float* wr = P.B + (int64_t)(k0 + 4 * j) * P.ldb;
  const int64_t o = (int64_t)(m0 + (tid >> 4)) * P.ldc + k0 + 4 * (tid & 15);
  const f32x4 h = ldv(P.aux + o);                           // relu mask source, issued with the first batch
  DotW dw;
  if (DOT) dw = dot_prefetch(P, k0 + 4 * (tid & 15));
  f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
  for (int nb = 0; nb < P.K; nb += 256) {
    f32x4 a[4], b[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int nq = nb + (wave + 4 * u) * 16 + 4 * q;
      a[u] = ldv(dyr + nq);
#pragma unroll
      for (int e = 0; e < 4; ++e) b[u][e] = ldv(wr + (int64_t)e * P.ldb + nq);
    }
    LOADS_FIRST();
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = MFMA(a[u][s], b[u][e][s], acc[e]);
  }
  f32x4 v; int orow, c4;
  hot_store(red, acc, wave, q, j, tid, v, orow, c4);
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = (h[e] > 0.f) ? v[e] : 0.f;
  *reinterpret_cast<f32x4*>(P.C + o) = v;
  if (DOT) dot_epilogue(P, dw, v, m0 + orow, blockIdx.x, c4);
}

// C[K',N] = A[M,K']^T . B[M,N];  aux_out[N] = colsum(B)    (reduction over P.M)     1-D grid over a tile list
struct DwHotArgs { GemmHot p[4]; int32_t tiles_per; int32_t nprob; };   // every problem has tiles_per tiles
template <bool ADAM>
__device__ inline void dw_hot_body(const DwHotArgs& args, const AdamFuse& A, const int bid, float* red) {
  // problem and tile from arithmetic on the block id: the descriptor load below does not wait for another load
  const int pi = bid / args.tiles_per, t = bid - pi * args.tiles_per;
  const GemmHot& P = args.p[pi];
  const int nx = P.N >> 6;
  const int by = t / nx, bx = t - by * nx;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, q = lane >> 4;
  const int k0 = by * 16, n0 = bx * 64;
  const float* xc = P.A + k0 + j;
  const float* yc = P.B + n0 + 4 * j;
  // optimiser operands of the tile element this thread finishes (and of the bias column it finishes when by == 0)
  float* const dst = P.C + (int64_t)(k0 + (tid >> 4)) * P.ldc + n0 + 4 * (tid & 15);
  const int64_t pidx = ADAM ? (int64_t)(dst - A.grad) : 0;
  const int64_t bidx = ADAM ? (int64_t)(P.aux_out + n0 + (tid & 63) - A.grad) : 0;
  AdamPre4 pre;
  float aQ = 0.f, aPi = 0.f, bm = 0.f, bv = 0.f, bth = 0.f;
  if (ADAM) {
    adam_alphas(A, aQ, aPi);
    pre = adam_prefetch4(A, pidx);
    if (by == 0 && tid < 64) { bm = A.m[bidx]; bv = A.v[bidx]; bth = A.theta[bidx]; }
  }
  f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
  f32x4 bsum = zero4();
  for (int mb = 0; mb < P.M; mb += 256) {
    float a[4][4];
    f32x4 b[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int mq = mb + (wave + 4 * u) * 16 + 4 * q;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        a[u][s] = xc[(int64_t)(mq + s) * P.lda];
        b[u][s] = ldv(yc + (int64_t)(mq + s) * P.ldb);
      }
    }
    LOADS_FIRST();
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bsum += b[u][s];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = MFMA(a[u][s], b[u][s][e], acc[e]);
      }
  }
  f32x4 v; int orow, c4;
  hot_store(red, acc, wave, q, j, tid, v, orow, c4);
  *reinterpret_cast<f32x4*>(dst) = v;
  if (ADAM) adam_apply4(A, (pidx < A.n_Q) ? -aQ : -aPi, pidx, v, pre);
  if (by == 0) {
    // column sums of B: 16 partials (4 waves x 4 lane groups) per column through LDS
    __syncthreads();
    *reinterpret_cast<f32x4*>(red + (wave * 4 + q) * 64 + 4 * j) = bsum;
    __syncthreads();
    if (tid < 64) {
      float gb = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) gb += red[r * 64 + tid];
      P.aux_out[n0 + tid] = gb;
      if (ADAM) {
        const float th = adam_elem(A, (bidx < A.n_Q) ? -aQ : -aPi, gb, bm, bv, bth);
        A.m[bidx] = bm; A.v[bidx] = bv; A.theta[bidx] = th;
      }
    }
  }
}


// Small weight gradients (layer-0 segments, output layers) on a compact tile list + the loss finalisation.
//   dW[w,N] = (X[M,w] / div)^T . dY[M,N];  db[N] = colsum(dY)         M % 256 == 0, X and dY plain row matrices
struct DwSmall {
  const float* x; const float* dY; float* dW; float* db;
  int32_t ldx, lddy, w, N;
  float div;
};
#define MAX_DW_SMALL 12
struct DwSmallArgs { DwSmall p[MAX_DW_SMALL]; int32_t nprob, M, slots; LossFin fin; };   // `slots` block ids per problem

// One 16 x 64 tile of a small problem.  YV: N % 4 == 0 (16-byte dY fragments); otherwise N == 1 (the critic's output
// layer): one dY column, one accumulator, a quarter of the MFMAs.  Uniform conditions are hoisted out of the unrolled
// load / MFMA loops (a branch per fragment made this body slower than a full 256-deep hidden-layer tile).
template <bool ADAM, bool YV>
__device__ inline void dw_small_tile(const DwSmall& P, const int M, const AdamFuse& A, const int t, float* red) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int nx = (P.N + 63) >> 6;
  const int by = t / nx, bx = t - by * nx;
  const int j = lane & 15, q = lane >> 4;
  const int k0 = by * 16, n0 = bx * 64;
  const int krow = k0 + j, col = n0 + 4 * j;
  const bool k_ok = krow < P.w;
  const float* xc = P.x + min(krow, P.w - 1);
  const int colc = YV ? min(col, P.N - 4) : 0;
  const float* yc = P.dY + colc;
  // optimiser operands of what this thread finishes, fetched with the first batch of loads
  const int grow = k0 + (tid >> 4), gcol = n0 + 4 * (tid & 15);
  const bool own = grow < P.w && gcol < P.N;
  float* const dst = P.dW + (int64_t)(own ? grow : 0) * P.N + (own ? gcol : 0);
  const int64_t pidx = ADAM ? (int64_t)(dst - A.grad) : 0;
  const bool own_b = P.db && by == 0 && tid < 64 && n0 + tid < P.N;
  const int64_t bidx = (ADAM && own_b) ? (int64_t)(P.db + n0 + tid - A.grad) : 0;
  float aQ = 0.f, aPi = 0.f, bm = 0.f, bv = 0.f, bth = 0.f;
  AdamPre4 pre;
  pre.m = zero4(); pre.v = zero4(); pre.th = zero4();
  if (ADAM) {
    adam_alphas(A, aQ, aPi);
    if (YV) {
      pre = adam_prefetch4(A, own ? pidx : 0);
    } else if (own) {                                       // N == 1: one element per owning thread
      pre.m[0] = A.m[pidx]; pre.v[0] = A.v[pidx]; pre.th[0] = A.theta[pidx];
    }
    if (own_b) { bm = A.m[bidx]; bv = A.v[bidx]; bth = A.theta[bidx]; }
  }
  f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
  f32x4 bsum = zero4();
  for (int mb = 0; mb < M; mb += 256) {
    float a[4][4];
    f32x4 b[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int mq = mb + (wave + 4 * u) * 16 + 4 * q;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        a[u][s] = xc[(int64_t)(mq + s) * P.ldx];
        if (YV) {
          b[u][s] = ldv(yc + (int64_t)(mq + s) * P.lddy);
        } else {
          b[u][s] = zero4();
          b[u][s][0] = yc[(int64_t)(mq + s) * P.lddy];
        }
      }
    }
    LOADS_FIRST();
    if (P.div != 1.0f) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s) a[u][s] = fdiv(a[u][s], P.div);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float av = k_ok ? a[u][s] : 0.f;
        bsum += b[u][s];
        if (YV) {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = MFMA(av, b[u][s][e], acc[e]);
        } else {
          acc[0] = MFMA(av, b[u][s][0], acc[0]);
        }
      }
  }
  int orow, c4;
  f32x4 v = reduce_tile(red, acc, wave, q, j, tid, orow, c4);
  if (own) {
    const float na = (pidx < A.n_Q) ? -aQ : -aPi;
    if (YV) {
      *reinterpret_cast<f32x4*>(dst) = v;
      if (ADAM) adam_apply4(A, na, pidx, v, pre);
    } else {
      dst[0] = v[0];                                        // N == 1 (gcol == 0)
      if (ADAM) {
        float m = pre.m[0], vv = pre.v[0];
        const float th = adam_elem(A, na, v[0], m, vv, pre.th[0]);
        A.m[pidx] = m; A.v[pidx] = vv; A.theta[pidx] = th;
      }
    }
  }
  if (P.db && by == 0) {
    __syncthreads();
    *reinterpret_cast<f32x4*>(red + (wave * 4 + q) * 64 + 4 * j) = bsum;
    __syncthreads();
    if (own_b) {
      float gb = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) gb += red[r * 64 + tid];
      P.db[n0 + tid] = gb;
      if (ADAM) {
        const float th = adam_elem(A, (bidx < A.n_Q) ? -aQ : -aPi, gb, bm, bv, bth);
        A.m[bidx] = bm; A.v[bidx] = bv; A.theta[bidx] = th;
      }
    }
  }
}

template <bool ADAM>
__device__ inline void dw_small_body(const DwSmallArgs& args, const AdamFuse& A, const int bid, float* red) {
  const int tid = threadIdx.x;
  // every problem owns `slots` consecutive block ids (surplus blocks exit at once): problem and tile follow from
  // arithmetic, so the descriptor load does not wait for a search through the table
  const int pi = bid / args.slots, t = bid - pi * args.slots;
  if (pi >= args.nprob) {
    // extra last block (only launched when fin.rows != NULL): losses (ddpg.py:439-441) from the per-row terms,
    // summed in a fixed order
    const LossFin& F = args.fin;
    float lq = 0.f, lp = 0.f, ll = 0.f;
    for (int m = tid; m < F.B; m += 256) {
      lq += F.rows[m];
      lp += F.rows[F.B + m];
      ll += F.rows[2 * F.B + m];
    }
    red[tid] = lq; red[256 + tid] = lp; red[512 + tid] = ll;
    __syncthreads();
    for (int h = 128; h >= 1; h >>= 1) {
      if (tid < h) {
        red[tid] += red[tid + h];
        red[256 + tid] += red[256 + tid + h];
        red[512 + tid] += red[512 + tid + h];
      }
      __syncthreads();
    }
    if (tid == 0) {
      const float invB = 1.0f / (float)F.B;
      F.out[0] = red[0] * invB;
      F.out[1] = -red[256] * invB + F.action_l2 * red[512] / (float)(F.B * F.U);
    }
    return;
  }
  const DwSmall& P = args.p[pi];
  if (t >= ((P.w + 15) >> 4) * ((P.N + 63) >> 6)) return;
  if ((P.N & 3) == 0) dw_small_tile<ADAM, true>(P, args.M, A, t, red);
  else dw_small_tile<ADAM, false>(P, args.M, A, t, red);
}

// Every weight/bias gradient of both networks + the loss finalisation in ONE launch: blocks [0, n_hot) run the
// hidden-layer tiles, the rest the small-problem tile list (the two lists are independent, so splitting them over two
// launches only bought a second ~4.5 us dependent stage).
struct DwAllArgs { DwHotArgs hot; DwSmallArgs small; int32_t n_hot; };
__global__ __launch_bounds__(256) void dw_all_kernel(DwAllArgs args) {
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];
  AdamFuse none;
  if ((int)blockIdx.x < args.n_hot) dw_hot_body<false>(args.hot, none, blockIdx.x, red);
  else dw_small_body<false>(args.small, none, (int)blockIdx.x - args.n_hot, red);
}

// The tail of a whole single-rank update in one launch (curious_ddpg_update): every weight/bias gradient with Adam
// applied in the tile epilogue, the loss finalisation, and -- in the first n_her blocks -- the HER gather of the NEXT
// update's batch (it depends on nothing this update computes; it must target a different staging buffer than the one
// the layer-0 gradient tiles of this launch still read).
__global__ __launch_bounds__(256) void dw_adam_her_kernel(DwAllArgs args, AdamFuse A, HerArgs h, int n_her) {
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];
  const int bid = (int)blockIdx.x - n_her;
  if (bid < 0) her_sample_body(h, blockIdx.x, red);
  else if (bid < args.n_hot) dw_hot_body<true>(args.hot, A, bid, red);
  else dw_small_body<true>(args.small, A, bid - args.n_hot, red);
}

// ------------------------------------------------------------------ lean layer-0 forward (total K <= 64)
// Y[M,N] = relu(sum_seg (clip(X_seg) / div) . W_seg + bias): the input is a virtual concatenation of up to 4 column
// segments of row matrices (batch columns [o | td | u], the actor output, g ...), each a multiple of 4 wide.  With
// K <= 64 every wave owns exactly one 16-wide chunk: 1 + 4 loads and 16 MFMAs per wave.
struct SegL { const float* x; const float* W; int32_t ld, w; float div, clip; };
struct L0Prob { SegL seg[MAX_SEG]; const float* bias; float* Y; int32_t nseg, M, N, ldy, relu, ktot; };
struct L0Args { L0Prob p[5]; };

// Branch-free lookup of the segment holding virtual input columns kv .. kv+3 (a 16-byte group never straddles a
// segment: widths % 4 == 0).  The segment table is read unconditionally (unused entries are zero-width); no divergent
// branches, no dependent scalar loads.
struct SegPick { const float* x; const float* W; int ld, kl; float dv, cl; bool ok; };
__device__ __forceinline__ SegPick seg_pick(const L0Prob& P, int kv) {
  // (fields are read straight from the kernarg struct: copying SegL structs around sent them through scratch memory
  //  and turned every dependent load into a flat load)
  const int e0 = P.seg[0].w, e1 = e0 + P.seg[1].w, e2 = e1 + P.seg[2].w, e3 = e2 + P.seg[3].w;   // exclusive ends
  const bool in0 = kv < e0, in1 = kv < e1, in2 = kv < e2, in3 = kv < e3;
  SegPick p;
  p.x = in0 ? P.seg[0].x : in1 ? P.seg[1].x : in2 ? P.seg[2].x : in3 ? P.seg[3].x : P.seg[0].x;
  p.W = in0 ? P.seg[0].W : in1 ? P.seg[1].W : in2 ? P.seg[2].W : in3 ? P.seg[3].W : P.seg[0].W;
  p.ld = in0 ? P.seg[0].ld : in1 ? P.seg[1].ld : in2 ? P.seg[2].ld : in3 ? P.seg[3].ld : P.seg[0].ld;
  p.kl = in3 ? kv - (in0 ? 0 : in1 ? e0 : in2 ? e1 : e2) : 0;                   // column inside the segment
  p.dv = in0 ? P.seg[0].div : in1 ? P.seg[1].div : in2 ? P.seg[2].div : in3 ? P.seg[3].div : 1.0f;
  p.cl = in0 ? P.seg[0].clip : in1 ? P.seg[1].clip : in2 ? P.seg[2].clip : in3 ? P.seg[3].clip : 0.0f;
  p.ok = in3;
  return p;
}
__device__ __forceinline__ bool seg_any_div(const L0Prob& P) {   // uniform: only the critic's action segment divides
  return (P.seg[0].div != 1.0f) || (P.seg[1].w > 0 && P.seg[1].div != 1.0f) ||
         (P.seg[2].w > 0 && P.seg[2].div != 1.0f) || (P.seg[3].w > 0 && P.seg[3].div != 1.0f);
}
__device__ inline f32x4 seg_prep(f32x4 a, float cl, float dv, bool any_div, bool ok) {
  const float c = (cl > 0.0f) ? cl : INFINITY;
#pragma unroll
  for (int e = 0; e < 4; ++e) a[e] = fclip(a[e], -c, c);
  if (any_div) {
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] = fdiv(a[e], dv);
  }
  return sel4(ok, a);
}

// NC = number of 64-wide k chunks (total K <= 64 * NC): wave w owns the 16-wide group w of every chunk
template <int NC>
__device__ inline void fwd_l0_body(const L0Prob& P, float* red) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 64;
  const int row = min(m0 + j, P.M - 1);
  f32x4 a[NC], b[NC][4];
  float dv[NC], cl[NC];
  bool ok[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const SegPick sp = seg_pick(P, 64 * c + 16 * wave + 4 * q);
    dv[c] = sp.dv; cl[c] = sp.cl; ok[c] = sp.ok;
    a[c] = ldv(sp.x + sp.kl + (int64_t)row * sp.ld);
    const float* wc = sp.W + (int64_t)sp.kl * P.N + n0 + 4 * j;
#pragma unroll
    for (int s = 0; s < 4; ++s) b[c][s] = ldv(wc + (int64_t)s * P.N);
  }
  const f32x4 bias = ldv(P.bias + n0 + 4 * (tid & 15));
  LOADS_FIRST();
  const bool any_div = seg_any_div(P);
  f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const f32x4 av = seg_prep(a[c], cl[c], dv[c], any_div, ok[c] && (m0 + j < P.M));
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = MFMA(av[s], b[c][s][e], acc[e]);
  }
  int orow, c4;
  f32x4 v = reduce_tile(red, acc, wave, q, j, tid, orow, c4);
  if (m0 + orow >= P.M) return;
  v += bias;
  if (P.relu) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
  }
  *reinterpret_cast<f32x4*>(P.Y + (int64_t)(m0 + orow) * P.ldy + n0 + 4 * c4) = v;
}


template <int NC>
__global__ __launch_bounds__(256) void fwd_l0_kernel(L0Args args) {
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];
  fwd_l0_body<NC>(args.p[blockIdx.z], red);
}

// ------------------------------------------------------------------ layers 0 + 1 in one launch
// C[M,256] = relu(relu(X . W0 + b0) . W1 + b1) for H == 256, total layer-0 K <= 64, M % 16 == 0.  Every workgroup
// first computes the FULL layer-0 tile h0[16 rows][256] of its batch rows (wave w: columns 64w..64w+63, all of K: 4 + 16
// loads and 64 MFMAs), parks it in LDS, and then runs the usual split-K layer-1 tile out of LDS.  The 4 column tiles of
// a row block recompute h0 (0.25 MFLOP each) -- that buys one dependent launch (~4.3 us) per forward pass.  Column
// tile 0 stores h0 when the backward pass needs it.  Problems z >= n01 of the same launch are plain layer-0 problems
// (the action-free pre-activations of fwd_pi_kernel).
struct L01Prob { L0Prob l0; const float* W1; const float* b1; float* C; };
struct L01Args { L01Prob p[3]; L0Prob pre[2]; int32_t n01; };
#define H0_LD 260      // LDS row stride of the h0 tile: 260 % 64 = 4 -> the 16 rows of a b128 read hit distinct banks

template <int NC>
__device__ inline void fwd_l01_body(const L01Prob& Q, float* red, float* h0s) {
  const L0Prob& P = Q.l0;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 64;
  const int H = 256;
  // ---- all global loads: 4 input fragments + 16 layer-0 weight fragments, 16 layer-1 weight fragments, biases
  constexpr int NG = 4 * NC;                       // 16-wide k groups of layer 0
  f32x4 xa[NG], w0[NG][4], w1[4][4];
  float dv[NG], cl[NG];
  bool okv[NG];
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    const SegPick sp = seg_pick(P, 16 * i + 4 * q);  // this lane's virtual input columns of k-group i
    dv[i] = sp.dv; cl[i] = sp.cl; okv[i] = sp.ok;
    xa[i] = ldv(sp.x + sp.kl + (int64_t)(m0 + j) * sp.ld);
    const float* wc0 = sp.W + (int64_t)sp.kl * H + 64 * wave + 4 * j;
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) w0[i][s2] = ldv(wc0 + (int64_t)s2 * H);
  }
  const f32x4 bias0 = ldv(P.bias + 64 * wave + 4 * j);
  const float* wc1 = Q.W1 + n0 + 4 * j;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int kq = (wave + 4 * u) * 16 + 4 * q;
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) w1[u][s2] = ldv(wc1 + (int64_t)(kq + s2) * H);
  }
  const f32x4 bias1 = ldv(Q.b1 + n0 + 4 * (tid & 15));
  LOADS_FIRST();
  // ---- layer 0: h0[rows 4q..4q+3][cols 64*wave + 4j + e]
  const bool any_div = seg_any_div(P);
  f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    const f32x4 a = seg_prep(xa[i], cl[i], dv[i], any_div, okv[i]);
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = MFMA(a[s2], w0[i][s2][e], acc[e]);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    f32x4 hv = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
    hv += bias0;
#pragma unroll
    for (int e = 0; e < 4; ++e) hv[e] = fmaxf(hv[e], 0.f);
    *reinterpret_cast<f32x4*>(h0s + (4 * q + r) * H0_LD + 64 * wave + 4 * j) = hv;
    if (blockIdx.x == 0 && P.Y)
      *reinterpret_cast<f32x4*>(P.Y + (int64_t)(m0 + 4 * q + r) * H + 64 * wave + 4 * j) = hv;
  }
  __syncthreads();
  // ---- layer 1 out of LDS
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] = zero4();
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int kq = (wave + 4 * u) * 16 + 4 * q;
    const f32x4 a = *reinterpret_cast<const f32x4*>(h0s + j * H0_LD + kq);
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = MFMA(a[s2], w1[u][s2][e], acc[e]);
  }
  int orow, c4;
  f32x4 v = reduce_tile(red, acc, wave, q, j, tid, orow, c4);
  v += bias1;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
  *reinterpret_cast<f32x4*>(Q.C + (int64_t)(m0 + orow) * H + n0 + 4 * c4) = v;
}

template <int NC>
__global__ __launch_bounds__(256) void fwd_l01_kernel(L01Args args) {
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];
  __shared__ __attribute__((aligned(16))) float h0s[16 * H0_LD];
  if ((int)blockIdx.z < args.n01) fwd_l01_body<NC>(args.p[blockIdx.z], red, h0s);
  else fwd_l0_body<NC>(args.pre[blockIdx.z - args.n01], red);
}

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

template <bool PART>
__global__ __launch_bounds__(256) void fwd_pi_kernel(FwdPiArgs args) {
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];
  __shared__ __attribute__((aligned(16))) float s_pi[16 * 4];
  const FwdPiProb& P = args.p[blockIdx.z];
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
    for (int t = 0; t < 4; ++t) pp[t] = P.part[((int64_t)t * args.B + m0) * 4 + (tid & 63)];
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) pr_h[r] = ldv(P.a_last + (int64_t)(pm + r) * H + 4 * lane);
#pragma unroll
    for (int e = 0; e < 4; ++e) wp[e] = ldv(P.WoutPi + (int64_t)(4 * lane + e) * 4);
  }
  const float bo = P.boutPi[lane & 3];
  const float* xr = P.zp + (int64_t)(m0 + j) * H;
  const float* wc = P.W1 + n0 + 4 * j;
  const f32x4 bias = ldv(P.b1 + n0 + 4 * (tid & 15));
  f32x4 z[4], wu[4][4], b[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int kq = (wave + 4 * u) * 16 + 4 * q;
    z[u] = ldv(xr + kq);
#pragma unroll
    for (int d = 0; d < 4; ++d) wu[u][d] = ldv(P.Wu + (int64_t)d * H + kq);
#pragma unroll
    for (int s = 0; s < 4; ++s) b[u][s] = ldv(wc + (int64_t)(kq + s) * H);
  }
  LOADS_FIRST();
  // ---- prologue: actor output layer of the 16 rows
  if (PART) {
    if (tid < 64) {
      const float pv = args.max_u * tanhf(((pp[0] + pp[1]) + (pp[2] + pp[3])) + bo);   // actor_critic.py:89
      s_pi[tid] = pv;
      if (blockIdx.x == 0 && P.pi_out) P.pi_out[(int64_t)m0 * 4 + tid] = pv;
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
      if (blockIdx.x == 0 && P.pi_out) P.pi_out[(int64_t)pm * 4 + lane] = pv;
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
      *reinterpret_cast<f32x4*>(P.h0_out + (int64_t)(m0 + j) * H + kq) = av;
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
  *reinterpret_cast<f32x4*>(P.C + (int64_t)(m0 + orow) * H + n0 + 4 * c4) = v;
}


// Critic output layers of the three critic passes + losses' per-row terms + backward through those output layers.
struct CriticHeadArgs {
  const float *c2, *d2, *e2;      // last hidden activations: main critic(u), main critic(pi), target critic   [B,H]
  const float* WoutQ; const float* boutQ;          // main/Q output layer
  const float* WoutQt; const float* boutQt;        // target/Q output layer
  const float* r; int32_t ldr;
  const float* pi; int32_t ldpi;
  int32_t B, H, U;
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
  const float invB = 1.0f / (float)a.B;
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
template <bool PART>
__global__ __launch_bounds__(256) void dx_crit_kernel(DxCritArgs a) {
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];
  __shared__ float s_dq[16];
  const int ch = blockIdx.z;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.y * 16, k0 = blockIdx.x * 64;
  const int H = 256;
  if (blockIdx.x == 0 && blockIdx.y == 0 && ch == 0 && tid == 0 && a.step_ctr) *a.step_ctr += 1;
  const float* hrow = a.hl[ch] + (int64_t)(m0 + j) * H;
  const float* wr = a.W + (int64_t)(k0 + 4 * j) * H;
  const float invB = 1.0f / (float)a.B;
  const bool need_dots = (ch == 0) || (blockIdx.x == 0);
  // ---- all loads
  f32x4 pr_h[4], pr_e[4];                                   // prologue rows m0 + 4*wave + r, this lane's 4 columns
  const int pm = m0 + 4 * wave;
  f32x4 wq = zero4(), wt = zero4();
  const float bq = a.boutQ[0], bt = a.boutQt[0];
  float rew[4], l2v[4];
  float pq[4] = {0.f, 0.f, 0.f, 0.f}, pt[4] = {0.f, 0.f, 0.f, 0.f}, rew_j = 0.f;
  f32x4 pi_j = zero4();
  if (PART) {
    // every lane finishes the heads of its own row m0 + j from the 4 column-tile partials
    const float* p1 = (ch == 0) ? a.partQ : a.partQpi;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      pq[t] = p1[(int64_t)t * a.B + m0 + j];
      pt[t] = a.partQt[(int64_t)t * a.B + m0 + j];
    }
    rew_j = a.r[(int64_t)(m0 + j) * a.ldr];
    pi_j = ldv(a.pi + (int64_t)(m0 + j) * 4);
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      pr_h[r] = ldv(a.hl[ch] + (int64_t)(pm + r) * H + 4 * lane);
      pr_e[r] = ldv(a.e2 + (int64_t)(pm + r) * H + 4 * lane);
    }
    wq = ldv(a.WoutQ + 4 * lane); wt = ldv(a.WoutQt + 4 * lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      rew[r] = a.r[(int64_t)(pm + r) * a.ldr];
      // sum_j (pi_j / max_u)^2 of row pm + r: lanes 0..U-1 hold one term each (ddpg.py:441)
      const float pv = (lane < a.U) ? a.pi[(int64_t)(pm + r) * a.ldpi + lane] : 0.f;
      l2v[r] = (ch == 1) ? pv : 0.f;
    }
  }
  const int64_t o = (int64_t)(m0 + (tid >> 4)) * H + k0 + 4 * (tid & 15);
  const f32x4 hm = ldv(a.hprev[ch] + o);
  f32x4 hv[4], wo[4], b[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int nq = (wave + 4 * u) * 16 + 4 * q;
    hv[u] = ldv(hrow + nq);
    wo[u] = ldv(a.WoutQ + nq);
#pragma unroll
    for (int e = 0; e < 4; ++e) b[u][e] = ldv(wr + (int64_t)e * H + nq);
  }
  LOADS_FIRST();
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
        a.rows[m] = diff * diff;                               // ddpg.py:439
        a.dQ[m] = dq;
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
        a.rows[a.B + m] = Qpi;                                 // ddpg.py:440
        a.rows[2 * a.B + m] = l2;
        a.out_Qpi[m] = Qpi;
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
            a.rows[m] = diff * diff;                         // ddpg.py:439
            a.dQ[m] = dq;
          }
        }
      } else {
        const float Qpi = d1 + bq;
        const float tt = l2v[r] / a.max_u;
        const float l2 = wave_sum(tt * tt);
        if (lane == 0) {
          a.rows[a.B + m] = Qpi;                             // ddpg.py:440
          a.rows[2 * a.B + m] = l2;
          a.out_Qpi[m] = Qpi;
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
      *reinterpret_cast<f32x4*>(a.dY[ch] + (int64_t)(m0 + j) * H + nq) = av;
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

template <bool PART>
__global__ __launch_bounds__(256) void dx_actor_kernel(DxActorArgs a) {
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];
  __shared__ __attribute__((aligned(16))) float s_dz[16 * 4];
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
    for (int t = 0; t < 4; ++t) pp[t] = a.part[((int64_t)t * a.B + m0) * 4 + (tid & 63)];
    pim = a.pi[(int64_t)m0 * 4 + (tid & 63)];
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) g4[r] = ldv(a.dd0 + (int64_t)(pm + r) * H + 4 * lane);
    pim = a.pi[(int64_t)pm * 4 + (lane & 15)];                // lane 4r+d: pi[pm + r][d]
#pragma unroll
    for (int d = 0; d < 4; ++d) wu[d] = ldv(a.Wu + (int64_t)d * H + 4 * lane);
  }
  const int64_t o = (int64_t)(m0 + (tid >> 4)) * H + k0 + 4 * (tid & 15);
  const f32x4 hm = ldv(a.hprev + o);
  const float* hrow = a.a2 + (int64_t)(m0 + j) * H;
  const float* wr = a.W + (int64_t)(k0 + 4 * j) * H;
  f32x4 hv[4], wo[4][4], b[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int nq = (wave + 4 * u) * 16 + 4 * q;
    hv[u] = ldv(hrow + nq);
#pragma unroll
    for (int s = 0; s < 4; ++s) wo[u][s] = ldv(a.WoutPi + (int64_t)(nq + s) * 4);
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
      if (blockIdx.x == 0) a.dz[(int64_t)m0 * 4 + tid] = dz;
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
      if (blockIdx.x == 0) a.dz[(int64_t)pm * 4 + lane] = dz;
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
      *reinterpret_cast<f32x4*>(a.da2 + (int64_t)(m0 + j) * H + nq) = av;
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


// ------------------------------------------------------------------ fused acting step of the batched rollout
// Actor output layer + max_u*tanh + exploration noise + clip + eps-greedy + one synthetic-env step, one wavefront per
// environment: replaces head_fwd_kernel + action_noise_kernel + env_step_kernel (3 dependent launches -> 1 per env step).
struct ActStepArgs {
  const float* part;        // PART: [4][n][4] column-tile partials of a_last . Wout (dot epilogue of the last layer)
  const float* a_last;      // actor last hidden activation [n, H]
  const float* Wout; const float* bout;
  int32_t H, U, n;
  float max_u_f;
  double noise_scale, random_eps, max_u;
  uint64_t seed, counter;
  const int64_t* counter_base;               // optional device-resident offset of the noise counter (graph replay)
  float* u_out; int32_t ldu;                 // actions as given to the env (also recorded in the episode row)
  curious_env_cfg_t E; curious_layout_t L;
  int32_t env_id0, t, off_change, off_success;
  const int32_t* episode; const int32_t* tasks;
  float* o; float* ag; const float* g; const float* td; float* staging;
  double reward_eps;
};

template <bool PART>
__global__ __launch_bounds__(256) void act_step_kernel(ActStepArgs a) {
  __shared__ float s_u[4][MAX_U];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int e = blockIdx.x * 4 + wave;
  if (e >= a.n) return;
  float o_[MAX_U];
#pragma unroll
  for (int d = 0; d < MAX_U; ++d) o_[d] = 0.f;
  if (PART) {
    // dimu == 4: lane d sums the 4 partials of output d
    float pp[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) pp[t] = a.part[((int64_t)t * a.n + e) * 4 + (lane & 3)];
    const float sv = (pp[0] + pp[1]) + (pp[2] + pp[3]);
#pragma unroll
    for (int d = 0; d < 4; ++d) o_[d] = sv;                    // only o_[lane] of lanes 0..3 is used below
  } else {
    const float* hrow = a.a_last + (int64_t)e * a.H;
    const bool al = (((uintptr_t)hrow | (uintptr_t)a.Wout) & 15) == 0;
    if (al && a.U == 4) row_dot_fast<4>(hrow, a.Wout, a.H, lane, o_);
    else row_dot(hrow, a.Wout, a.H, a.U, lane, o_);
  }
  if (lane < a.U) {
    float v = 0.f;
#pragma unroll
    for (int d = 0; d < MAX_U; ++d)
      if (d == lane) v = o_[d];
    v = a.max_u_f * tanhf(v + a.bout[lane]);                  // actor_critic.py:89
    const uint64_t ctr = a.counter + (a.counter_base ? (uint64_t)*a.counter_base : 0ull);
    v = noise_apply(v, e * a.U + lane, e, a.noise_scale, a.random_eps, a.max_u, nullptr, nullptr, nullptr, a.seed,
                    ctr);                                     // ddpg.py:149-152
    s_u[wave][lane] = v;
    a.u_out[(int64_t)e * a.ldu + lane] = v;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  env_step_body(a.E, a.L, a.env_id0, a.episode, a.tasks, s_u[wave], a.t, a.o, a.ag, a.g, a.td, a.staging,
                a.off_change, a.off_success, a.reward_eps, e, lane);
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
                          const L0Prob* pre = nullptr, int npre = 0) {
  const int H = c->hidden;
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
        dim3 grid(H / 64, M / 16, nch + npre);
        { ProfScope ps__(CK_FWD_L01, st);
          if (kmax <= 64) hipLaunchKernelGGL(fwd_l01_kernel<1>, grid, dim3(256), 0, st, fa);
          else hipLaunchKernelGGL(fwd_l01_kernel<2>, grid, dim3(256), 0, st, fa); }
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
      dim3 grid(H / 64, M / 16, nch);
      { ProfScope ps__(CK_FWD_LAYER, st);
        if (want_dot) hipLaunchKernelGGL(fwd_hot_kernel<true>, grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(fwd_hot_kernel<false>, grid, dim3(256), 0, st, a); }
      CURIOUS_LAUNCH_CHECK("fwd_hot_kernel");
      continue;
    }
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
    { ProfScope ps__(CK_FWD_LAYER0, st); hipLaunchKernelGGL(fwd_layer_kernel, grid, dim3(256), 0, st, a); }
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
  ObsIn in;
  memset(&in, 0, sizeof(in));
  in.o = o; in.ldo = ldo; in.td = td; in.ldtd = ldtd; in.g = g; in.ldg = ldg; in.ag = ag; in.ldag = ldag;
  in.clip = clip_obs; in.relative = relative_goals;
  fill_obs_stats(cfg, in, o_stats, g_stats);
  const float* thPi = theta + pi_offset(cfg);
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

extern "C" int curious_policy_act_env_step(const curious_net_cfg_t* cfg, const float* theta, int32_t n, float clip_obs,
                                           float* workspace, double noise_scale, double random_eps, uint64_t seed,
                                           uint64_t counter, const int64_t* counter_base, float* u_out, int32_t ldu,
                                           const curious_env_cfg_t* E,
                                           const curious_layout_t* L, int32_t env_id0, const int32_t* episode,
                                           const int32_t* tasks, int32_t t, float* o, float* ag, const float* g,
                                           const float* td, float* staging, int32_t off_change, int32_t off_success,
                                           double reward_eps, curious_stream_t stream) {
  if (check_cfg(cfg)) return -1;
  CURIOUS_CHECK(theta && workspace && u_out && E && L && episode && tasks && o && ag && g && td && staging,
                "curious_policy_act_env_step: NULL argument");
  CURIOUS_CHECK(!cfg->normalize_obs && cfg->modular, "curious_policy_act_env_step: modular nets without input "
                                                     "normalisation only (use curious_policy_forward otherwise)");
  CURIOUS_CHECK(cfg->dimu == 4 && L->dimu == 4 && cfg->dimo == E->dimo && cfg->dimtd == E->ntasks &&
                    cfg->dimg == 3 * E->ntasks, "curious_policy_act_env_step: network / env dimensions differ");
  CURIOUS_CHECK(t >= 0 && t < L->T, "curious_policy_act_env_step: t out of range");
  if (n <= 0) return 0;
  hipStream_t st = as_stream(stream);
  Ws w = carve(cfg, n, workspace);
  NetOff offPi = net_off(cfg, false);
  const int H = cfg->hidden, nl = cfg->layers;
  ObsIn in;
  memset(&in, 0, sizeof(in));
  in.o = o; in.ldo = E->dimo; in.td = td; in.ldtd = E->ntasks; in.g = g; in.ldg = 3 * E->ntasks;
  in.clip = clip_obs;
  const float* thPi = theta + pi_offset(cfg);
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
  { ProfScope ps__(CK_ACT_STEP, st);
    if (part) hipLaunchKernelGGL(act_step_kernel<true>, dim3((n + 3) / 4), dim3(256), 0, st, k);
    else hipLaunchKernelGGL(act_step_kernel<false>, dim3((n + 3) / 4), dim3(256), 0, st, k); }
  CURIOUS_LAUNCH_CHECK("act_step_kernel");
  return 0;
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

static int ddpg_grads_impl(const curious_net_cfg_t* cfg, const float* theta_main, const float* theta_target,
                           const float* batch, const curious_batch_layout_t* BL, int32_t B, const float* o_stats,
                           const float* g_stats, float* workspace, float* grad, float* out_losses, float* out_Q_pi,
                           int64_t* step_ctr, curious_stream_t stream, const UpdateTail* tail) {
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
  fill_obs_stats(cfg, cur, o_stats, g_stats);
  nxt = cur;
  nxt.o = batch + BL->off_o2;            // target nets see (o_2, g_2) (ddpg.py:427-431)
  nxt.g = batch + BL->off_g2;

  // ---- forward level A: hidden layers of target actor, main critic(u), main actor
  Chain ch[3];
  ch[0].theta = ttPi; ch[0].off = offPi; ch[0].in = nxt; ch[0].critic = false; ch[0].act = w.act[0];
  ch[1].theta = thQ; ch[1].off = offQ; ch[1].in = cur; ch[1].critic = true; ch[1].act = w.act[1];
  ch[2].theta = thPi; ch[2].off = offPi; ch[2].in = cur; ch[2].critic = false; ch[2].act = w.act[2];
  // level B = hidden layers of target critic(pi_target), main critic(pi)
  Chain cb[2];
  cb[0].theta = ttQ; cb[0].off = offQ; cb[0].in = nxt; cb[0].in.u = w.pi_t; cb[0].in.ldu = U; cb[0].critic = true;
  cb[0].act = w.act[3];
  cb[1].theta = thQ; cb[1].off = offQ; cb[1].in = cur; cb[1].in.u = w.pi; cb[1].in.ldu = U; cb[1].critic = true;
  cb[1].act = w.act[4];
  const int64_t urow = cfg->dimo + (cfg->modular ? cfg->dimtd : cfg->dimg);   // first action row of W0
  // Lean route: the action-independent part of level B's layer 0 rides on level A's layer-0 launch, the actor output
  // layers and the action rows are folded into level B's layer-1 launch (fwd_pi_kernel).
  L0Prob pre[2];
  memset(pre, 0, sizeof(pre));
  bool fuse_pi = nl >= 2 && H == 256 && U == 4 && (B % 16 == 0) && aligned16(thQ) && aligned16(thPi) && aligned16(ttQ) &&
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
  const bool use_part = fuse_pi && nl >= 3;
  if (use_part) {
    ch[0].dot_mode = 2; ch[0].dot_w = ttPi + offPi.Wout; ch[0].dot_out = w.part[0];
    ch[1].dot_mode = 1; ch[1].dot_w = thQ + offQ.Wout; ch[1].dot_out = w.part[2];
    ch[2].dot_mode = 2; ch[2].dot_w = thPi + offPi.Wout; ch[2].dot_out = w.part[1];
    cb[0].dot_mode = 1; cb[0].dot_w = ttQ + offQ.Wout; cb[0].dot_out = w.part[3];
    cb[1].dot_mode = 1; cb[1].dot_w = thQ + offQ.Wout; cb[1].dot_out = w.part[4];
  }
  if (fuse_pi) {
    if (forward_chains(cfg, ch, 3, B, st, 0, pre, 2)) return -2;
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
    dim3 grid(H / 64, B / 16, 2);
    { ProfScope ps__(CK_FWD_PI, st);
      if (use_part) hipLaunchKernelGGL(fwd_pi_kernel<true>, grid, dim3(256), 0, st, fa);
      else hipLaunchKernelGGL(fwd_pi_kernel<false>, grid, dim3(256), 0, st, fa); }
    CURIOUS_LAUNCH_CHECK("fwd_pi_kernel");
    if (forward_chains(cfg, cb, 2, B, st, 2)) return -2;
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

  // ---- critic output layers, per-row loss terms, backward through the output layers (fused with the first hidden
  //      backward level when the lean kernels apply)
  const bool dx_hot = hot_ok(B, H, H) && aligned16(thQ) && aligned16(thPi) && aligned16(workspace);
  const bool fuse_crit = dx_hot && nl >= 2 && H == 256;
  CURIOUS_CHECK(!use_part || fuse_crit, "curious_ddpg_grads: inconsistent lean-path conditions");
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
    a.B = B; a.H = H; a.U = U;
    a.gamma = cfg->gamma; a.clip_lo = -cfg->clip_return; a.clip_hi = cfg->clip_pos_returns ? 0.0f : INFINITY;
    a.max_u = cfg->max_u;
    a.dQ = w.dQ; a.rows = w.rows; a.out_Qpi = out_Q_pi; a.step_ctr = step_ctr;
    dim3 grid(H / 64, B / 16, 2);
    { ProfScope ps__(CK_CRITIC_HEAD, st);
      if (use_part) hipLaunchKernelGGL(dx_crit_kernel<true>, grid, dim3(256), 0, st, a);
      else hipLaunchKernelGGL(dx_crit_kernel<false>, grid, dim3(256), 0, st, a); }
    CURIOUS_LAUNCH_CHECK("dx_crit_kernel");
  } else
  {
    CriticHeadArgs a;
    a.c2 = w.act[1][nl - 1]; a.d2 = w.act[4][nl - 1]; a.e2 = w.act[3][nl - 1];
    a.WoutQ = thQ + offQ.Wout; a.boutQ = thQ + offQ.bout;
    a.WoutQt = ttQ + offQ.Wout; a.boutQt = ttQ + offQ.bout;
    a.r = batch + BL->off_r; a.ldr = ld; a.pi = w.pi; a.ldpi = U;
    a.B = B; a.H = H; a.U = U;
    a.gamma = cfg->gamma; a.clip_lo = -cfg->clip_return; a.clip_hi = cfg->clip_pos_returns ? 0.0f : INFINITY;
    a.max_u = cfg->max_u;
    a.dc2 = w.dact[0][nl - 1]; a.dd2 = w.dact[1][nl - 1]; a.dQ = w.dQ; a.rows = w.rows; a.out_Qpi = out_Q_pi;
    a.step_ctr = step_ctr;
    { ProfScope ps__(CK_CRITIC_HEAD, st);
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
      dim3 grid(H / 64, B / 16, 2);
      { ProfScope ps__(CK_DX, st);
        if (use_part && l == 1) hipLaunchKernelGGL(dx_hot_kernel<true>, grid, dim3(256), 0, st, ha);
        else hipLaunchKernelGGL(dx_hot_kernel<false>, grid, dim3(256), 0, st, ha); }
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
    { ProfScope ps__(CK_DX, st); hipLaunchKernelGGL(dx_kernel, grid, dim3(256), 0, st, da); }
    CURIOUS_LAUNCH_CHECK("dx_kernel");
  }
  // ---- weight/bias gradients: problem lists for the lean kernels (launched after the actor's backward chain)
  const bool dw_hot = dx_hot && (B % 256 == 0) && (nl - 1) <= 4 && !cfg->normalize_obs;
  LossFin fin;
  fin.rows = w.rows; fin.out = out_losses; fin.B = B; fin.U = U; fin.action_l2 = cfg->action_l2;
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
      tiles += (H / 16) * (H / 64);
      ++nh;
    }
    hw.nprob = nh;
    Seg seg[MAX_SEG];
    int ns = l0_segments(cfg, off, nullptr, cur, critic, cfg->max_u, seg);
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
  if (lean_dw) lean_dw = build_net(true, hwAll, tAll, smAll, stAll) && build_net(false, hwAll, tAll, smAll, stAll);
  // ---- into the action slot of critic(pi), through tanh + l2 term -> dz; backward through the actor output layer
  //      (fused with the actor's first hidden backward level when the lean kernels apply)
  const float l2c = cfg->action_l2 * 2.0f / (cfg->max_u * cfg->max_u * (float)(B * U));
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
    dim3 grid(H / 64, B / 16, 1);
    { ProfScope ps__(CK_ACTOR_DZ, st);
      if (use_part) hipLaunchKernelGGL(dx_actor_kernel<true>, grid, dim3(256), 0, st, a);
      else hipLaunchKernelGGL(dx_actor_kernel<false>, grid, dim3(256), 0, st, a); }
    CURIOUS_LAUNCH_CHECK("dx_actor_kernel");
  } else {
    ActorDzArgs a;
    a.dd0 = w.dact[1][0]; a.Wu = thQ + offQ.W0 + urow * H; a.pi = w.pi; a.ldpi = U;
    a.a2 = w.act[2][nl - 1]; a.WoutPi = thPi + offPi.Wout; a.dz = w.dz; a.da2 = w.dact[2][nl - 1];
    a.B = B; a.H = H; a.U = U; a.max_u = cfg->max_u;
    a.l2c = l2c;
    { ProfScope ps__(CK_ACTOR_DZ, st);
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
      dim3 grid(H / 64, B / 16, 1);
      { ProfScope ps__(CK_DX, st); hipLaunchKernelGGL(dx_hot_kernel<false>, grid, dim3(256), 0, st, ha); }
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
    { ProfScope ps__(CK_DX, st); hipLaunchKernelGGL(dx_kernel, grid, dim3(256), 0, st, da); }
    CURIOUS_LAUNCH_CHECK("dx_kernel(actor)");
  }
  if (lean_dw) {
    smAll.fin = fin;
    dwAll.n_hot = tAll;
    hwAll.tiles_per = (H / 16) * (H / 64);
    smAll.slots = stAll > 0 ? stAll : 1;
    const int nsmall = smAll.nprob * smAll.slots;           // + 1 block for the loss finalisation
    if (tail && (!tail->her || her_lds_bytes(&tail->h.L) <= sizeof(float) * 4 * 16 * 64)) {
      const int n_her = tail->her ? (tail->h.n + SPB - 1) / SPB : 0;
      { ProfScope ps__(CK_DW_ADAM_HER, st);
        hipLaunchKernelGGL(dw_adam_her_kernel, dim3(n_her + tAll + nsmall + 1), dim3(256), 0, st, dwAll, tail->adam,
                           tail->h, n_her); }
      CURIOUS_LAUNCH_CHECK("dw_adam_her_kernel");
      return 0;
    }
    { ProfScope ps__(CK_DW, st);
      hipLaunchKernelGGL(dw_all_kernel, dim3(tAll + nsmall + 1), dim3(256), 0, st, dwAll); }
    CURIOUS_LAUNCH_CHECK("dw_all_kernel");
  } else {
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
    if (tail->her) {
      const curious_next_batch_t* nx = tail->next;
      return curious_adam_update_and_sample(const_cast<float*>(theta_main), a->m, a->v, grad, n_Q, tail->n_pi,
                                            a->alpha_tab, step_ctr, a->tab_base, a->tab_len, a->alpha_tab ? nullptr : ah,
                                            a->beta1, a->one_minus_beta1, a->beta2, a->one_minus_beta2, a->epsilon,
                                            nx->storage, nx->buf_stride, nx->L, nx->tasks, nx->P, nx->rng, B, nx->batch,
                                            BL, stream);
    }
    return curious_adam_update(const_cast<float*>(theta_main), a->m, a->v, grad, n_Q, tail->n_pi, a->alpha_tab, step_ctr,
                               a->tab_base, a->tab_len, a->alpha_tab ? nullptr : ah, a->beta1, a->one_minus_beta1,
                               a->beta2, a->one_minus_beta2, a->epsilon, stream);
  }
  return 0;
}

extern "C" int curious_ddpg_grads(const curious_net_cfg_t* cfg, const float* theta_main, const float* theta_target,
                                  const float* batch, const curious_batch_layout_t* BL, int32_t B,
                                  const float* o_stats, const float* g_stats, float* workspace, float* grad,
                                  float* out_losses, float* out_Q_pi, int64_t* step_ctr, curious_stream_t stream) {
  return ddpg_grads_impl(cfg, theta_main, theta_target, batch, BL, B, o_stats, g_stats, workspace, grad, out_losses,
                         out_Q_pi, step_ctr, stream, nullptr);
}

extern "C" int curious_ddpg_update(const curious_net_cfg_t* cfg, float* theta_main, const float* theta_target,
                                   const float* batch, const curious_batch_layout_t* BL, int32_t B,
                                   const float* o_stats, const float* g_stats, float* workspace, float* grad,
                                   float* out_losses, float* out_Q_pi, int64_t* step_ctr,
                                   const curious_adam_state_t* adam, const curious_next_batch_t* next,
                                   curious_stream_t stream) {
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
  if (next) {
    CURIOUS_CHECK(next->batch && next->batch != batch, "curious_ddpg_update: the next batch needs its own staging buffer");
    if (her_fill_args(t.h, next->storage, next->buf_stride, next->L, next->tasks, next->P, nullptr, next->rng, B,
                      next->batch, BL)) return -1;
    t.her = true;
  }
  return ddpg_grads_impl(cfg, theta_main, theta_target, batch, BL, B, o_stats, g_stats, workspace, grad, out_losses,
                         out_Q_pi, step_ctr, stream, &t);
}
