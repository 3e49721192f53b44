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
