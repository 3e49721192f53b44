// K8 fused Adam over [theta_Q | theta_pi], K9 Polyak, and the parameter checksum for check_synced.
//
// Replaces (reference): MpiAdam.update arithmetic mpi_adam.py:29-35 (after the all-reduce), target-net
// init/update ddpg.py:459-462, MpiAdam.check_synced mpi_adam.py:42-50.  Every float32 operation is a single
// correctly-rounded op (no FMA contraction) so results equal NumPy's float32 arithmetic bit for bit
// (oracle/optim.py, nep50=False).  HBM traffic: 28 B/param (read g,m,v,theta; write m,v,theta).
#include "common.h"
#include "her_body.h"

struct AdamArgs {
  float* theta;
  float* m;
  float* v;
  const float* grad;
  int64_t n_Q, n;
  const float* alpha_tab;
  const int64_t* step_ctr;
  int64_t tab_base;
  int32_t tab_len;
  float a_Q, a_pi;
  float b1, omb1, b2, omb2, eps;
};

__device__ __forceinline__ void adam_body(const AdamArgs& a, const int block, const int nblocks) {
  float aQ = a.a_Q, aPi = a.a_pi;
  if (a.alpha_tab) {
    // the step counter was advanced by ddpg_grads; the table is a ring refilled by the host every tab_len steps
    int64_t idx = ((*a.step_ctr) - 1 - a.tab_base) % a.tab_len;
    if (idx < 0) idx += a.tab_len;
    aQ = a.alpha_tab[2 * idx];
    aPi = a.alpha_tab[2 * idx + 1];
  }
  for (int64_t i = (int64_t)block * 256 + threadIdx.x; i < a.n; i += (int64_t)nblocks * 256) {
    float g = a.grad[i];
    float m = __fadd_rn(__fmul_rn(a.b1, a.m[i]), __fmul_rn(a.omb1, g));              // mpi_adam.py:31
    float v = __fadd_rn(__fmul_rn(a.b2, a.v[i]), __fmul_rn(a.omb2, __fmul_rn(g, g)));  // mpi_adam.py:32
    float na = (i < a.n_Q) ? -aQ : -aPi;
    float step = fdiv(__fmul_rn(na, m), __fadd_rn(sqrtf(v), a.eps));        // mpi_adam.py:33
    a.m[i] = m;
    a.v[i] = v;
    a.theta[i] = __fadd_rn(a.theta[i], step);                                          // mpi_adam.py:34
  }
}

__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) { adam_body(a, blockIdx.x, gridDim.x); }

// Adam + the HER gather of the NEXT update in one launch: the gather does not depend on the parameters, so its
// workgroups (the first n_her blocks) ride along with the optimiser's instead of being a dependent launch of their
// own at the head of the next update (~7 us per update).  Stream order guarantees that every reader of the previous
// staged batch (layer-0 forward, layer-0 weight gradients) has finished before this launch starts.
__global__ __launch_bounds__(256) void adam_her_kernel(AdamArgs a, HerArgs h, int n_her) {
  extern __shared__ float lds[];
  if ((int)blockIdx.x < n_her) her_sample_body(h, blockIdx.x, lds);
  else adam_body(a, blockIdx.x - n_her, gridDim.x - n_her);
}

static int fill_adam(AdamArgs& a, float* theta, float* m, float* v, const float* grad, int64_t n_Q, int64_t n_pi,
                     const float* alpha_tab, const int64_t* step_ctr, int64_t tab_base, int32_t tab_len,
                     const float* alpha_host, float beta1, float one_minus_beta1, float beta2, float one_minus_beta2,
                     float epsilon) {
  CURIOUS_CHECK(theta && m && v && grad, "curious_adam_update: NULL argument");
  CURIOUS_CHECK((alpha_tab && step_ctr && tab_len > 0) || alpha_host, "curious_adam_update: no step size given");
  a.theta = theta; a.m = m; a.v = v; a.grad = grad;
  a.n_Q = n_Q; a.n = n_Q + n_pi;
  a.alpha_tab = alpha_tab; a.step_ctr = step_ctr; a.tab_base = tab_base; a.tab_len = tab_len;
  a.a_Q = alpha_host ? alpha_host[0] : 0.f;
  a.a_pi = alpha_host ? alpha_host[1] : 0.f;
  a.b1 = beta1; a.omb1 = one_minus_beta1; a.b2 = beta2; a.omb2 = one_minus_beta2; a.eps = epsilon;
  return 0;
}

extern "C" int curious_adam_update_and_sample(float* theta, float* m, float* v, const float* grad, int64_t n_Q,
                                              int64_t n_pi, const float* alpha_tab, const int64_t* step_ctr,
                                              int64_t tab_base, int32_t tab_len, const float* alpha_host, float beta1,
                                              float one_minus_beta1, float beta2, float one_minus_beta2, float epsilon,
                                              const float* storage, int64_t buf_stride, const curious_layout_t* L,
                                              const curious_tasks_t* tasks, const curious_sample_params_t* P,
                                              const curious_sample_rng_t* rng, int32_t n, float* batch,
                                              const curious_batch_layout_t* BL, curious_stream_t stream) {
  AdamArgs a;
  if (fill_adam(a, theta, m, v, grad, n_Q, n_pi, alpha_tab, step_ctr, tab_base, tab_len, alpha_host, beta1,
                one_minus_beta1, beta2, one_minus_beta2, epsilon)) return -1;
  HerArgs h;
  if (her_fill_args(h, storage, buf_stride, L, tasks, P, nullptr, rng, n, batch, BL)) return -1;
  int n_her = (n + SPB - 1) / SPB;
  int blocks = (int)((a.n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  { ProfScope ps__(CK_ADAM, as_stream(stream));
    hipLaunchKernelGGL(adam_her_kernel, dim3(n_her + blocks), dim3(256), her_lds_bytes(L), as_stream(stream), a, h,
                       n_her); }
  CURIOUS_LAUNCH_CHECK("adam_her_kernel");
  return 0;
}

extern "C" int curious_adam_update(float* theta, float* m, float* v, const float* grad, int64_t n_Q, int64_t n_pi,
                                   const float* alpha_tab, const int64_t* step_ctr, int64_t tab_base,
                                   int32_t tab_len, const float* alpha_host, float beta1, float one_minus_beta1,
                                   float beta2, float one_minus_beta2, float epsilon, curious_stream_t stream) {
  AdamArgs a;
  if (fill_adam(a, theta, m, v, grad, n_Q, n_pi, alpha_tab, step_ctr, tab_base, tab_len, alpha_host, beta1,
                one_minus_beta1, beta2, one_minus_beta2, epsilon)) return -1;
  if (a.n <= 0) return 0;
  int blocks = (int)((a.n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  { ProfScope ps__(CK_ADAM, as_stream(stream));
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), a); }
  CURIOUS_LAUNCH_CHECK("adam_kernel");
  return 0;
}

__global__ __launch_bounds__(256) void polyak_kernel(float* __restrict__ target, const float* __restrict__ main_,
                                                    int64_t n, float p, float q) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float mv = main_[i];
    target[i] = (p == 0.0f && q == 1.0f) ? mv : __fadd_rn(__fmul_rn(p, target[i]), __fmul_rn(q, mv));  // ddpg.py:462
  }
}

extern "C" int curious_polyak_update(float* target, const float* main_, int64_t n, float polyak,
                                     float one_minus_polyak, curious_stream_t stream) {
  CURIOUS_CHECK(target && main_, "curious_polyak_update: NULL argument");
  if (n <= 0) return 0;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  { ProfScope ps__(CK_POLYAK, as_stream(stream)); hipLaunchKernelGGL(polyak_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), target, main_, n, polyak,
                     one_minus_polyak); }
  CURIOUS_LAUNCH_CHECK("polyak_kernel");
  return 0;
}

__global__ __launch_bounds__(256) void checksum_kernel(const uint32_t* __restrict__ bits, int64_t n,
                                                      unsigned long long* __restrict__ out) {
  unsigned long long s = 0, x = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    unsigned long long h = ((unsigned long long)bits[i] + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1));
    h ^= h >> 31;
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 29;
    s += h;
    x ^= h;
  }
  for (int off = 32; off >= 1; off >>= 1) {
    s += __shfl_xor(s, off);
    x ^= __shfl_xor(x, off);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&out[0], s);       // integer add / xor commute: order-independent, deterministic
    atomicXor(&out[1], x);
  }
}

extern "C" int curious_param_checksum(const float* theta, int64_t n, uint64_t* out, curious_stream_t stream) {
  CURIOUS_CHECK(theta && out, "curious_param_checksum: NULL argument");
  hipError_t e = hipMemsetAsync(out, 0, 2 * sizeof(uint64_t), as_stream(stream));
  CURIOUS_CHECK(e == hipSuccess, "curious_param_checksum: memset failed: %s", hipGetErrorString(e));
  if (n <= 0) return 0;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  { ProfScope ps__(CK_CHECKSUM, as_stream(stream)); hipLaunchKernelGGL(checksum_kernel, dim3(blocks), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const uint32_t*>(theta), n, reinterpret_cast<unsigned long long*>(out)); }
  CURIOUS_LAUNCH_CHECK("checksum_kernel");
  return 0;
}
