// K4: Normalizer running-statistics reduction (wavefront/LDS column reduction, float64 accumulate).
//
// Replaces (reference): Normalizer.update normalizer.py:64-70 and recompute_stats normalizer.py:50-61,84-118.
// NumPy semantics kept: v.sum(axis=0) and square(v).sum(axis=0) are float64 sums of float32-representable
// values; `local_sum += <float64 array>` adds in float64 and rounds once to float32.  The summation ORDER
// differs from NumPy's row-sequential loop (tree here), so parity is to ~1e-7 relative, not bit-exact; the
// result is deterministic (fixed tree, no atomics).
#include "common.h"

#define NB_ROWS 256   // rows per partial block

__global__ __launch_bounds__(256) void norm_partial_kernel(const float* __restrict__ rows, int32_t n_rows,
                                                          int32_t stride, int32_t col_off, int32_t dim,
                                                          int32_t cp, double* __restrict__ partial) {
  // thread = (row group, column); consecutive lanes read consecutive floats of one row
  __shared__ double s_sum[256];
  __shared__ double s_sq[256];
  const int col = threadIdx.x % cp, rg = threadIdx.x / cp, nrg = 256 / cp;
  const int r0 = blockIdx.x * NB_ROWS;
  const int r1 = min(n_rows, r0 + NB_ROWS);
  double s = 0.0, q = 0.0;
  if (col < dim) {
    for (int r = r0 + rg; r < r1; r += nrg) {
      double v = (double)rows[(int64_t)r * stride + col_off + col];
      s = __dadd_rn(s, v);
      q = __dadd_rn(q, __dmul_rn(v, v));
    }
  }
  s_sum[threadIdx.x] = s;
  s_sq[threadIdx.x] = q;
  __syncthreads();
  for (int h = nrg >> 1; h >= 1; h >>= 1) {
    if (rg < h) {
      s_sum[threadIdx.x] = __dadd_rn(s_sum[threadIdx.x], s_sum[threadIdx.x + h * cp]);
      s_sq[threadIdx.x] = __dadd_rn(s_sq[threadIdx.x], s_sq[threadIdx.x + h * cp]);
    }
    __syncthreads();
  }
  if (rg == 0 && col < dim) {
    partial[(int64_t)blockIdx.x * 2 * dim + col] = s_sum[col];
    partial[(int64_t)blockIdx.x * 2 * dim + dim + col] = s_sq[col];
  }
}

__global__ void norm_final_kernel(const double* __restrict__ partial, int32_t n_blocks, int32_t dim, int32_t n_rows,
                                  float* __restrict__ acc) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < 2 * dim) {
    double s = 0.0;
    for (int b = 0; b < n_blocks; ++b) s = __dadd_rn(s, partial[(int64_t)b * 2 * dim + c]);
    acc[c] = (float)__dadd_rn((double)acc[c], s);               // normalizer.py:68-69 (f32 += f64 array)
  } else if (c == 2 * dim) {
    acc[c] = __fadd_rn(acc[c], (float)n_rows);                  // normalizer.py:70
  }
}

extern "C" int64_t curious_norm_scratch_doubles(int32_t n_rows, int32_t dim) {
  int64_t nb = (n_rows + NB_ROWS - 1) / NB_ROWS;
  if (nb < 1) nb = 1;
  return nb * 2 * dim;
}

extern "C" int curious_norm_update(const float* rows, int32_t n_rows, int32_t stride, int32_t col_off, int32_t dim,
                                   float* acc, double* scratch, curious_stream_t stream) {
  CURIOUS_CHECK(rows && acc && scratch, "curious_norm_update: NULL argument");
  CURIOUS_CHECK(dim > 0 && dim <= 256, "curious_norm_update: dim must be in 1..256");
  if (n_rows <= 0) return 0;
  int cp = 1;
  while (cp < dim) cp <<= 1;
  int nb = (n_rows + NB_ROWS - 1) / NB_ROWS;
  { ProfScope ps__(CK_NORM_PARTIAL, as_stream(stream)); hipLaunchKernelGGL(norm_partial_kernel, dim3(nb), dim3(256), 0, as_stream(stream), rows, n_rows, stride, col_off,
                     dim, cp, scratch); }
  CURIOUS_LAUNCH_CHECK("norm_partial_kernel");
  int n = 2 * dim + 1;
  { ProfScope ps__(CK_NORM_FINAL, as_stream(stream)); hipLaunchKernelGGL(norm_final_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), scratch, nb, dim,
                     n_rows, acc); }
  CURIOUS_LAUNCH_CHECK("norm_final_kernel");
  return 0;
}

// state = [sum[dim] | sumsq[dim] | count[1] | mean[dim] | std[dim]]
__global__ void norm_recompute_kernel(float* __restrict__ acc, float* __restrict__ state, int32_t dim, float world,
                                      float eps) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  float* sum = state;
  float* sumsq = state + dim;
  float* count = state + 2 * dim;
  float* mean = state + 2 * dim + 1;
  float* stdv = state + 3 * dim + 1;
  float cnt = 0.0f;
  if (c < dim) {
    // normalizer.py:84-94: mean over ranks (buf /= size), float32
    float s_sum = fdiv(acc[c], world);
    float s_sq = fdiv(acc[dim + c], world);
    float s_cnt = fdiv(acc[2 * dim], world);
    cnt = __fadd_rn(count[0], s_cnt);                              // update_op normalizer.py:50-54
    float sm = __fadd_rn(sum[c], s_sum);
    float sq = __fadd_rn(sumsq[c], s_sq);
    float mu = fdiv(sm, cnt);                                 // recompute_op normalizer.py:55-61
    float var = __fsub_rn(fdiv(sq, cnt), __fmul_rn(mu, mu));
    float e2 = __fmul_rn(eps, eps);
    sum[c] = sm;
    sumsq[c] = sq;
    mean[c] = mu;
    stdv[c] = sqrtf(fmaxf(e2, var));
  }
  __syncthreads();
  // every thread of this (single) block has read acc / count before they are reset
  if (c < dim) {
    acc[c] = 0.0f;
    acc[dim + c] = 0.0f;
    if (c == 0) {
      count[0] = cnt;
      acc[2 * dim] = 0.0f;
    }
  }
}

extern "C" int curious_norm_recompute(float* acc, float* state, int32_t dim, float world_size, float eps,
                                      curious_stream_t stream) {
  CURIOUS_CHECK(acc && state, "curious_norm_recompute: NULL argument");
  CURIOUS_CHECK(dim > 0 && dim <= 1024, "curious_norm_recompute: dim must be in 1..1024 (single block)");
  { ProfScope ps__(CK_NORM_RECOMPUTE, as_stream(stream)); hipLaunchKernelGGL(norm_recompute_kernel, dim3(1), dim3(((dim + 63) / 64) * 64), 0, as_stream(stream), acc, state,
                     dim, world_size, eps); }
  CURIOUS_LAUNCH_CHECK("norm_recompute_kernel");
  return 0;
}

// ------------------------------------------------------------------ both normalisers of an agent in two launches
// DDPG.store_episode feeds o_stats and g_stats from the same HER-sampled batch (ddpg.py:216-223): one partial launch
// over the virtual column range [o | g] and one finishing launch -- wavefront (DPP) reductions of the partial sums,
// the float32 accumulator update of normalizer.py:68-70 and, on a single rank, recompute_stats (normalizer.py:96-118)
// -- replace the 6 launches of update / update / recompute / recompute.
#define NP_ROWS 64

__global__ __launch_bounds__(256) void norm_pair_partial_kernel(const float* __restrict__ rows, int32_t n_rows,
                                                               int32_t stride, int32_t off_a, int32_t dim_a,
                                                               int32_t off_b, int32_t dim_b, int32_t cp, int32_t nb,
                                                               double* __restrict__ partial) {
  __shared__ double s_sum[256];
  __shared__ double s_sq[256];
  const int ncol = dim_a + dim_b;
  const int col = threadIdx.x % cp, rg = threadIdx.x / cp, nrg = 256 / cp;
  const int coff = (col < dim_a) ? off_a + col : off_b + (col - dim_a);
  const int r0 = blockIdx.x * NP_ROWS;
  const int r1 = min(n_rows, r0 + NP_ROWS);
  double s = 0.0, q = 0.0;
  if (col < ncol) {
    for (int r = r0 + rg; r < r1; r += nrg) {
      const double v = (double)rows[(int64_t)r * stride + coff];
      s = __dadd_rn(s, v);
      q = __dadd_rn(q, __dmul_rn(v, v));
    }
  }
  s_sum[threadIdx.x] = s;
  s_sq[threadIdx.x] = q;
  __syncthreads();
  for (int h = nrg >> 1; h >= 1; h >>= 1) {
    if (rg < h) {
      s_sum[threadIdx.x] = __dadd_rn(s_sum[threadIdx.x], s_sum[threadIdx.x + h * cp]);
      s_sq[threadIdx.x] = __dadd_rn(s_sq[threadIdx.x], s_sq[threadIdx.x + h * cp]);
    }
    __syncthreads();
  }
  if (rg == 0 && col < ncol) {                              // column-major partials: the finishing wave reads them contiguously
    partial[(int64_t)col * nb + blockIdx.x] = s_sum[col];
    partial[(int64_t)(ncol + col) * nb + blockIdx.x] = s_sq[col];
  }
}

template <int CTRL>
__device__ inline double dpp_mov_f64(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xFFFFFFFFll), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, false);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned int)lo);
}
__device__ inline double lane_read_f64(double v, int l) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFll), l);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned int)lo);
}
// fixed-order sum over the 64 lanes of a wavefront: 4 DPP steps inside each row of 16 lanes, then the 4 row sums
__device__ inline double wave_sum_f64(double v) {
  v = __dadd_rn(v, dpp_mov_f64<0xB1>(v));                   // quad_perm [1,0,3,2]
  v = __dadd_rn(v, dpp_mov_f64<0x4E>(v));                   // quad_perm [2,3,0,1]
  v = __dadd_rn(v, dpp_mov_f64<0x141>(v));                  // row_half_mirror
  v = __dadd_rn(v, dpp_mov_f64<0x140>(v));                  // row_mirror
  return __dadd_rn(__dadd_rn(lane_read_f64(v, 0), lane_read_f64(v, 16)),
                   __dadd_rn(lane_read_f64(v, 32), lane_read_f64(v, 48)));
}

__device__ inline void norm_recompute_elem(float* acc, float* state, int dim, int c, float world, float eps, float& cnt) {
  float* sum = state; float* sumsq = state + dim; float* count = state + 2 * dim;
  float* mean = state + 2 * dim + 1; float* stdv = state + 3 * dim + 1;
  const float s_sum = fdiv(acc[c], world);                  // normalizer.py:84-94: mean over ranks
  const float s_sq = fdiv(acc[dim + c], world);
  const float s_cnt = fdiv(acc[2 * dim], world);
  cnt = __fadd_rn(count[0], s_cnt);                         // normalizer.py:50-54
  const float sm = __fadd_rn(sum[c], s_sum);
  const float sq = __fadd_rn(sumsq[c], s_sq);
  const float mu = fdiv(sm, cnt);                           // normalizer.py:55-61
  const float var = __fsub_rn(fdiv(sq, cnt), __fmul_rn(mu, mu));
  sum[c] = sm; sumsq[c] = sq; mean[c] = mu;
  stdv[c] = sqrtf(fmaxf(__fmul_rn(eps, eps), var));
}

#define NF_COLS 6
// one workgroup of 16 wavefronts: wave w finishes columns w, w + 16, ... of [sum_a | sum_b | sumsq_a | sumsq_b]
__global__ __launch_bounds__(1024) void norm_pair_final_kernel(const double* __restrict__ partial, int32_t nb,
                                                              int32_t dim_a, int32_t dim_b, int32_t n_rows,
                                                              float* __restrict__ acc_a, float* __restrict__ acc_b,
                                                              float* __restrict__ state_a, float* __restrict__ state_b,
                                                              float eps_a, float eps_b, const float* __restrict__ skip) {
  if (skip && *skip != 0.0f) return;                        // a NaN rollout (rollout.py:268-271) feeds no statistics
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ncol = dim_a + dim_b;
  // NF_COLS columns of a wave at a time: their loads are independent, so NF_COLS of them are in flight per lane; the
  // order of the additions of one column (lane-strided, then wave_sum_f64) does not depend on NF_COLS
  for (int c0 = wave; c0 < 2 * ncol; c0 += 16 * NF_COLS) {
    double s[NF_COLS];
#pragma unroll
    for (int u = 0; u < NF_COLS; ++u) s[u] = 0.0;
    for (int b = lane; b < nb; b += 64) {
      double v[NF_COLS];
#pragma unroll
      for (int u = 0; u < NF_COLS; ++u) {
        const int c = c0 + 16 * u;
        v[u] = (c < 2 * ncol) ? partial[(int64_t)c * nb + b] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < NF_COLS; ++u) s[u] = __dadd_rn(s[u], v[u]);
    }
#pragma unroll
    for (int u = 0; u < NF_COLS; ++u) {
      const int c = c0 + 16 * u;
      if (c >= 2 * ncol) break;
      const double t = wave_sum_f64(s[u]);
      if (lane == 0) {
        const int kind = c / ncol, vc = c - kind * ncol;
        float* p = (vc < dim_a) ? acc_a + kind * dim_a + vc : acc_b + kind * dim_b + (vc - dim_a);
        *p = (float)__dadd_rn((double)*p, t);               // normalizer.py:68-69 (f32 += f64 array)
      }
    }
  }
  if (threadIdx.x == 0) {
    acc_a[2 * dim_a] = __fadd_rn(acc_a[2 * dim_a], (float)n_rows);             // normalizer.py:70
    acc_b[2 * dim_b] = __fadd_rn(acc_b[2 * dim_b], (float)n_rows);
  }
  if (!state_a) return;
  __threadfence_block();
  __syncthreads();
  // single rank: recompute_stats of both normalisers right here (world size 1)
  const int c = threadIdx.x;
  float cnt_a = 0.f, cnt_b = 0.f;
  if (c < dim_a) norm_recompute_elem(acc_a, state_a, dim_a, c, 1.0f, eps_a, cnt_a);
  else if (c < ncol) norm_recompute_elem(acc_b, state_b, dim_b, c - dim_a, 1.0f, eps_b, cnt_b);
  __syncthreads();
  if (c < dim_a) {
    acc_a[c] = 0.0f; acc_a[dim_a + c] = 0.0f;
    if (c == 0) { state_a[2 * dim_a] = cnt_a; acc_a[2 * dim_a] = 0.0f; }
  } else if (c < ncol) {
    const int d = c - dim_a;
    acc_b[d] = 0.0f; acc_b[dim_b + d] = 0.0f;
    if (d == 0) { state_b[2 * dim_b] = cnt_b; acc_b[2 * dim_b] = 0.0f; }
  }
}

extern "C" int64_t curious_norm_pair_scratch_doubles(int32_t n_rows, int32_t dim_a, int32_t dim_b) {
  int64_t nb = (n_rows + NP_ROWS - 1) / NP_ROWS;
  if (nb < 1) nb = 1;
  return nb * 2 * (dim_a + dim_b);
}

extern "C" int curious_norm_update_pair(const float* rows, int32_t n_rows, int32_t stride, int32_t off_a, int32_t dim_a,
                                        int32_t off_b, int32_t dim_b, float* acc_a, float* acc_b, float* state_a,
                                        float* state_b, float eps_a, float eps_b, double* scratch, const float* skip,
                                        curious_stream_t stream) {
  CURIOUS_CHECK(rows && acc_a && acc_b && scratch, "curious_norm_update_pair: NULL argument");
  CURIOUS_CHECK((state_a == nullptr) == (state_b == nullptr), "curious_norm_update_pair: both states or none");
  CURIOUS_CHECK(dim_a > 0 && dim_b > 0 && dim_a + dim_b <= 256, "curious_norm_update_pair: dim_a + dim_b must be <= 256");
  if (n_rows <= 0) return 0;
  int cp = 1;
  while (cp < dim_a + dim_b) cp <<= 1;
  const int nb = (n_rows + NP_ROWS - 1) / NP_ROWS;
  { ProfScope ps__(CK_NORM_PAIR_PARTIAL, as_stream(stream));
    hipLaunchKernelGGL(norm_pair_partial_kernel, dim3(nb), dim3(256), 0, as_stream(stream), rows, n_rows, stride, off_a,
                       dim_a, off_b, dim_b, cp, nb, scratch); }
  CURIOUS_LAUNCH_CHECK("norm_pair_partial_kernel");
  { ProfScope ps__(CK_NORM_PAIR_FINAL, as_stream(stream));
    hipLaunchKernelGGL(norm_pair_final_kernel, dim3(1), dim3(1024), 0, as_stream(stream), scratch, nb, dim_a, dim_b,
                       n_rows, acc_a, acc_b, state_a, state_b, eps_a, eps_b, skip); }
  CURIOUS_LAUNCH_CHECK("norm_pair_final_kernel");
  return 0;
}
