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
  curious_transposed_t keep;      // square matrices whose transposed copies are kept current (keep.n == 0: none);
                                  // keep.fault: fault word of the gradient workspace (non-zero = the launch leaves
                                  // theta / m / v / the copies alone) or NULL
};
// Batched experts (curious_adam_update_and_sample_experts): blockIdx.y = expert; eo = its slab offset (parameters,
// moments, step counter, step-size ring, transposed copies, fault word), eg = its offset in the contiguous gradient block.
__device__ __forceinline__ const int64_t* opt_i64(const int64_t* p, int64_t eo) {
  return reinterpret_cast<const int64_t*>(reinterpret_cast<const float*>(p) + eo);
}
// The optimiser leaves everything alone while the fault word of the gradient workspace is non-zero -- or, on several
// ranks, while ANY rank's was when the gradients were produced: that rank's flag element came through the all-reduce with
// the gradients (curious_transposed_t.fault_flag).  The decision is the same on every rank, so the replicas stay identical;
// one thread per agent then raises the local word as well, which makes the freeze sticky and visible to this rank's host.
__device__ __forceinline__ bool adam_faulted(const AdamArgs& a, const int64_t eo, const int64_t eg) {
  if (!a.keep.fault) return false;
  if (*reinterpret_cast<const int32_t*>(reinterpret_cast<const float*>(a.keep.fault) + eo) != 0) return true;
  return a.keep.fault_flag > 0 && a.grad[eg + a.keep.fault_flag - 1] != 0.0f;
}
__device__ __forceinline__ void adam_note_fault(const AdamArgs& a, const int64_t eo, const int64_t eg) {
  int32_t* word = const_cast<int32_t*>(reinterpret_cast<const int32_t*>(reinterpret_cast<const float*>(a.keep.fault) + eo));
  if (a.keep.fault_flag > 0 && a.grad[eg + a.keep.fault_flag - 1] != 0.0f && *word == 0) atomicAdd(word, 1);
}

// The launch's three scalar inputs -- fault word, collective fault flag, step counter -- without their latency (round 4;
// mlp_lean_gemm.h adam_early / PIN_V): a block used to begin with  load fault word, wait; load step counter, wait; load
// step sizes, wait; -- three dependent round trips through cold caches before its first operand was requested, in a
// kernel of 6 us.  adam_scalars() only ISSUES the loads (branch-free: a NULL pointer reads a valid dummy and is masked
// later); the verdict and the step sizes are taken (adam_verdict) when the block's first operands are in flight.
#define OPT_PIN_V(x) asm volatile("" : "+v"(x))
struct AdamScalars { int32_t fw, lo, hi; float flag; };
__device__ __forceinline__ AdamScalars adam_scalars(const AdamArgs& a, const int64_t eo, const int64_t eg) {
  AdamScalars s;
  const int32_t* fp = a.keep.fault ? reinterpret_cast<const int32_t*>(reinterpret_cast<const float*>(a.keep.fault) + eo)
                                   : reinterpret_cast<const int32_t*>(a.theta);
  const float* gp = (a.keep.fault && a.keep.fault_flag > 0) ? a.grad + eg + a.keep.fault_flag - 1 : a.theta;
  const int64_t* cp = a.alpha_tab ? opt_i64(a.step_ctr, eo) : reinterpret_cast<const int64_t*>(a.theta);
  s.fw = *fp;
  s.flag = *gp;
  const int64_t c = *cp;
  s.lo = (int32_t)c; s.hi = (int32_t)(c >> 32);
  return s;
}
// false: the launch leaves everything alone (adam_faulted); otherwise the step sizes of this update
__device__ __forceinline__ bool adam_verdict(const AdamArgs& a, AdamScalars& s, float& aQ, float& aPi, const int64_t eo) {
  OPT_PIN_V(s.fw); OPT_PIN_V(s.flag); OPT_PIN_V(s.lo); OPT_PIN_V(s.hi);
  aQ = a.a_Q; aPi = a.a_pi;
  if (a.alpha_tab) {
    const int64_t ctr = (int64_t)(((uint64_t)(uint32_t)s.hi << 32) | (uint32_t)s.lo);
    const int64_t v = ctr - 1 - a.tab_base;
    int64_t idx;
    if ((a.tab_len & (a.tab_len - 1)) == 0) {
      idx = v & (int64_t)(a.tab_len - 1);
    } else {
      idx = v % a.tab_len;
      if (idx < 0) idx += a.tab_len;
    }
    aQ = a.alpha_tab[eo + 2 * idx];
    aPi = a.alpha_tab[eo + 2 * idx + 1];
  }
  const bool local = a.keep.fault && s.fw != 0;
  const bool collective = a.keep.fault && a.keep.fault_flag > 0 && s.flag != 0.0f;
  return !(local || collective);
}

__device__ __forceinline__ void adam_alphas(const AdamArgs& a, float& aQ, float& aPi, const int64_t eo) {
  aQ = a.a_Q; aPi = a.a_pi;
  if (a.alpha_tab) {
    // the step counter was advanced by ddpg_grads; the table is a ring refilled by the host every tab_len steps
    int64_t idx = ((*opt_i64(a.step_ctr, eo)) - 1 - a.tab_base) % a.tab_len;
    if (idx < 0) idx += a.tab_len;
    aQ = a.alpha_tab[eo + 2 * idx];
    aPi = a.alpha_tab[eo + 2 * idx + 1];
  }
}
// the arithmetic of one element (mpi_adam.py:31-34); returns the new parameter
__device__ __forceinline__ float adam_math(const AdamArgs& a, const float na, const float g, float& m, float& v,
                                           const float th) {
  m = __fadd_rn(__fmul_rn(a.b1, m), __fmul_rn(a.omb1, g));                          // mpi_adam.py:31
  v = __fadd_rn(__fmul_rn(a.b2, v), __fmul_rn(a.omb2, __fmul_rn(g, g)));            // mpi_adam.py:32
  const float step = fdiv(__fmul_rn(na, m), __fadd_rn(sqrtf(v), a.eps));            // mpi_adam.py:33
  return __fadd_rn(th, step);                                                       // mpi_adam.py:34
}
__device__ __forceinline__ void adam_one(const AdamArgs& a, const int64_t i, const float aQ, const float aPi,
                                         const int64_t eo, const int64_t eg) {
  float m = a.m[i + eo], v = a.v[i + eo];
  const float th = adam_math(a, (i < a.n_Q) ? -aQ : -aPi, a.grad[i + eg], m, v, a.theta[i + eo]);
  a.m[i + eo] = m;
  a.v[i + eo] = v;
  a.theta[i + eo] = th;
}

typedef float f32x4_o __attribute__((ext_vector_type(4)));
// vec4: n, n_Q, the matrices of `keep` and the four vectors all start on 16-byte boundaries (the padded parameter
// layout of curious_param_total) -- one thread then takes 4 consecutive elements with 16-byte loads and stores
__device__ __forceinline__ void adam_body(const AdamArgs& a, const int block, const int nblocks, const int64_t eo,
                                          const int64_t eg, const bool vec4) {
  AdamScalars sc = adam_scalars(a, eo, eg);
  float aQ = 0.f, aPi = 0.f;
  bool go = true, decided = false;
  const int64_t msize = (int64_t)a.keep.dim * a.keep.dim;
  if (vec4) {
    for (int64_t i = ((int64_t)block * 256 + threadIdx.x) * 4; i < a.n; i += (int64_t)nblocks * 1024) {
      bool tiled = false;                                   // inside a matrix the tile blocks below take care of
      for (int j = 0; j < a.keep.n; ++j) tiled |= (uint64_t)(i - a.keep.src_off[j]) < (uint64_t)msize;
      if (tiled) continue;
      const f32x4_o g = *reinterpret_cast<const f32x4_o*>(a.grad + i + eg);
      f32x4_o m = *reinterpret_cast<const f32x4_o*>(a.m + i + eo), v = *reinterpret_cast<const f32x4_o*>(a.v + i + eo);
      f32x4_o th = *reinterpret_cast<const f32x4_o*>(a.theta + i + eo);
      if (!decided) {                                       // (the first operands are in flight: now the verdict)
        __builtin_amdgcn_sched_barrier(0);
        go = adam_verdict(a, sc, aQ, aPi, eo);
        decided = true;
      }
      if (!go) break;
      const float na = (i < a.n_Q) ? -aQ : -aPi;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float me = m[e], ve = v[e];
        th[e] = adam_math(a, na, g[e], me, ve, th[e]);
        m[e] = me; v[e] = ve;
      }
      *reinterpret_cast<f32x4_o*>(a.m + i + eo) = m;
      *reinterpret_cast<f32x4_o*>(a.v + i + eo) = v;
      *reinterpret_cast<f32x4_o*>(a.theta + i + eo) = th;
    }
    if (!decided) go = adam_verdict(a, sc, aQ, aPi, eo);
    if (!go && block == 0 && threadIdx.x == 0) adam_note_fault(a, eo, eg);
    return;
  }
  go = adam_verdict(a, sc, aQ, aPi, eo);
  if (!go) {
    if (block == 0 && threadIdx.x == 0) adam_note_fault(a, eo, eg);
    return;
  }
  for (int64_t i = (int64_t)block * 256 + threadIdx.x; i < a.n; i += (int64_t)nblocks * 256) {
    bool tiled = false;
    for (int j = 0; j < a.keep.n; ++j) tiled |= (uint64_t)(i - a.keep.src_off[j]) < (uint64_t)msize;
    if (!tiled) adam_one(a, i, aQ, aPi, eo, eg);
  }
}
static inline bool adam_vec4(const AdamArgs& a, int64_t expert_stride, int64_t grad_stride) {
  bool ok = (a.n % 4 == 0) && (a.n_Q % 4 == 0) && (expert_stride % 4 == 0) && (grad_stride % 4 == 0) &&
            (((uintptr_t)a.theta | (uintptr_t)a.m | (uintptr_t)a.v | (uintptr_t)a.grad) & 15) == 0;
  for (int j = 0; j < a.keep.n; ++j) ok = ok && (a.keep.src_off[j] % 4 == 0);
  return ok && (((int64_t)a.keep.dim * a.keep.dim) % 4 == 0);
}

// One 32 x 32 tile of a matrix whose transposed copy is kept: the same arithmetic element by element, the updated tile
// goes through LDS to the copy (WT[n][k] = W[k][n]), both sides in 128-byte row segments.  (4 elements per thread: a
// 64 x 64 tile made these 64 blocks the long pole of the launch, +3 us.)
#define ADAM_TILE 32
__device__ __forceinline__ void adam_tile_body(const AdamArgs& a, const int tb, float (*tile)[ADAM_TILE + 1],
                                               const int64_t eo, const int64_t eg) {
  AdamScalars sc = adam_scalars(a, eo, eg);
  float aQ, aPi;
  const int dim = a.keep.dim, per = dim / ADAM_TILE;
  const int j = tb / (per * per), t = tb - j * per * per;
  const int k0 = (t / per) * ADAM_TILE, n0 = (t % per) * ADAM_TILE;
  const int c = threadIdx.x & 31, r8 = threadIdx.x >> 5;
  const int64_t base = a.keep.src_off[j];
  // all operands of the thread's elements first, then the arithmetic and the stores: element by element the stores
  // would fence the next element's loads
  constexpr int NE = ADAM_TILE / 8;
  float g[NE], m[NE], v[NE], th[NE];
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int64_t e = base + (int64_t)(k0 + 8 * i + r8) * dim + n0 + c;
    g[i] = a.grad[e + eg]; m[i] = a.m[e + eo]; v[i] = a.v[e + eo]; th[i] = a.theta[e + eo];
  }
  __builtin_amdgcn_sched_barrier(0);
  if (!adam_verdict(a, sc, aQ, aPi, eo)) return;            // (uniform over the workgroup: nobody reaches the barrier)
  const float na = (base < a.n_Q) ? -aQ : -aPi;             // a matrix lies inside one network
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int64_t e = base + (int64_t)(k0 + 8 * i + r8) * dim + n0 + c;
    th[i] = adam_math(a, na, g[i], m[i], v[i], th[i]);
    a.m[e + eo] = m[i]; a.v[e + eo] = v[i]; a.theta[e + eo] = th[i];
    tile[8 * i + r8][c] = th[i];
  }
  __syncthreads();
  float* dst = a.keep.dst[j] + eo;
#pragma unroll
  for (int i = 0; i < NE; ++i) dst[(int64_t)(n0 + 8 * i + r8) * dim + k0 + c] = tile[c][8 * i + r8];
}
static inline int adam_tiles(const AdamArgs& a) {
  const int per = a.keep.n ? a.keep.dim / ADAM_TILE : 0;
  return a.keep.n * per * per;
}

__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a, int n_tile, int vec4) {
  __shared__ float tile[ADAM_TILE][ADAM_TILE + 1];
  if ((int)blockIdx.x < n_tile) adam_tile_body(a, blockIdx.x, tile, 0, 0);
  else adam_body(a, blockIdx.x - n_tile, gridDim.x - n_tile, 0, 0, vec4 != 0);
}

// Adam + the HER gather of the NEXT update in one launch: the gather does not depend on the parameters, so its
// workgroups (the first n_her blocks) ride along with the optimiser's instead of being a dependent launch of their
// own at the head of the next update (~7 us per update).  Stream order guarantees that every reader of the previous
// staged batch (layer-0 forward, layer-0 weight gradients) has finished before this launch starts.
// (batched experts: grid.y = expert)
__global__ __launch_bounds__(256) void adam_her_kernel(AdamArgs a, HerArgs h, int n_her, int n_tile, int64_t ex_stride,
                                                       int64_t grad_stride, uint64_t seed_stride, int vec4) {
  extern __shared__ float lds[];
  const int64_t eo = (int64_t)blockIdx.y * ex_stride, eg = (int64_t)blockIdx.y * grad_stride;
  if ((int)blockIdx.x < n_her) her_sample_body(h, blockIdx.x, lds, eo, (uint64_t)blockIdx.y * seed_stride);
  else if ((int)blockIdx.x < n_her + n_tile)
    adam_tile_body(a, blockIdx.x - n_her, reinterpret_cast<float(*)[ADAM_TILE + 1]>(lds), eo, eg);
  else adam_body(a, blockIdx.x - n_her - n_tile, gridDim.x - n_her - n_tile, eo, eg, vec4 != 0);
}

static int fill_adam(AdamArgs& a, float* theta, float* m, float* v, const float* grad, int64_t n_Q, int64_t n_pi,
                     const float* alpha_tab, const int64_t* step_ctr, int64_t tab_base, int32_t tab_len,
                     const float* alpha_host, float beta1, float one_minus_beta1, float beta2, float one_minus_beta2,
                     float epsilon, const curious_transposed_t* keep) {
  CURIOUS_CHECK(theta && m && v && grad, "curious_adam_update: NULL argument");
  memset(&a.keep, 0, sizeof(a.keep));
  if (keep) { a.keep.fault = keep->fault; a.keep.fault_flag = keep->fault_flag; }
  if (keep && keep->n > 0) {
    CURIOUS_CHECK(keep->n <= 8 && keep->dim > 0 && keep->dim % ADAM_TILE == 0,
                  "curious_adam_update: bad description of the transposed copies");
    for (int j = 0; j < keep->n; ++j)
      CURIOUS_CHECK(keep->dst[j] && keep->src_off[j] >= 0 &&
                        keep->src_off[j] + (int64_t)keep->dim * keep->dim <= n_Q + n_pi,
                    "curious_adam_update: bad description of the transposed copies");
    a.keep = *keep;
  }
  CURIOUS_CHECK((alpha_tab && step_ctr && tab_len > 0) || alpha_host, "curious_adam_update: no step size given");
  a.theta = theta; a.m = m; a.v = v; a.grad = grad;
  a.n_Q = n_Q; a.n = n_Q + n_pi;
  a.alpha_tab = alpha_tab; a.step_ctr = step_ctr; a.tab_base = tab_base; a.tab_len = tab_len;
  a.a_Q = alpha_host ? alpha_host[0] : 0.f;
  a.a_pi = alpha_host ? alpha_host[1] : 0.f;
  a.b1 = beta1; a.omb1 = one_minus_beta1; a.b2 = beta2; a.omb2 = one_minus_beta2; a.eps = epsilon;
  return 0;
}

static int adam_and_sample(int32_t n_experts, int64_t expert_stride, int64_t grad_stride, uint64_t seed_stride,
                           float* theta, float* m, float* v, const float* grad, int64_t n_Q, int64_t n_pi,
                           const float* alpha_tab, const int64_t* step_ctr, int64_t tab_base, int32_t tab_len,
                           const float* alpha_host, float beta1, float one_minus_beta1, float beta2,
                           float one_minus_beta2, float epsilon, const float* storage, int64_t buf_stride,
                           const curious_layout_t* L, const curious_tasks_t* tasks, const curious_sample_params_t* P,
                           const curious_sample_rng_t* rng, int32_t n, float* batch, const curious_batch_layout_t* BL,
                           const curious_transposed_t* keep, curious_stream_t stream) {
  AdamArgs a;
  if (fill_adam(a, theta, m, v, grad, n_Q, n_pi, alpha_tab, step_ctr, tab_base, tab_len, alpha_host, beta1,
                one_minus_beta1, beta2, one_minus_beta2, epsilon, keep)) return -1;
  // storage == NULL: no gather rides along (the caller had it done in the gradient launch, curious_ddpg_grads* `next`)
  HerArgs h;
  memset(&h, 0, sizeof(h));
  if (storage && her_fill_args(h, storage, buf_stride, L, tasks, P, nullptr, rng, n, batch, BL)) return -1;
  int n_her = storage ? (n + SPB - 1) / SPB : 0;
  const bool vec4 = adam_vec4(a, expert_stride, grad_stride);
  int blocks = (int)((a.n + (vec4 ? 1023 : 255)) / (vec4 ? 1024 : 256));
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  const int n_tile = adam_tiles(a);
  size_t lds = storage ? her_lds_bytes(L) : 0;
  if (n_tile && lds < sizeof(float) * ADAM_TILE * (ADAM_TILE + 1)) lds = sizeof(float) * ADAM_TILE * (ADAM_TILE + 1);
  { ProfScope ps__(CK_ADAM_HER, as_stream(stream));
    hipLaunchKernelGGL(adam_her_kernel, dim3(n_her + n_tile + blocks, n_experts), dim3(256), lds, as_stream(stream), a,
                       h, n_her, n_tile, expert_stride, grad_stride, seed_stride, vec4 ? 1 : 0); }
  CURIOUS_LAUNCH_CHECK("adam_her_kernel");
  return 0;
}

extern "C" int curious_adam_update_and_sample(float* theta, float* m, float* v, const float* grad, int64_t n_Q,
                                              int64_t n_pi, const float* alpha_tab, const int64_t* step_ctr,
                                              int64_t tab_base, int32_t tab_len, const float* alpha_host, float beta1,
                                              float one_minus_beta1, float beta2, float one_minus_beta2, float epsilon,
                                              const float* storage, int64_t buf_stride, const curious_layout_t* L,
                                              const curious_tasks_t* tasks, const curious_sample_params_t* P,
                                              const curious_sample_rng_t* rng, int32_t n, float* batch,
                                              const curious_batch_layout_t* BL, const curious_transposed_t* keep,
                                              curious_stream_t stream) {
  return adam_and_sample(1, 0, 0, 0, theta, m, v, grad, n_Q, n_pi, alpha_tab, step_ctr, tab_base, tab_len, alpha_host,
                         beta1, one_minus_beta1, beta2, one_minus_beta2, epsilon, storage, buf_stride, L, tasks, P, rng, n,
                         batch, BL, keep, stream);
}

extern "C" int curious_adam_update_and_sample_experts(int32_t n_experts, int64_t expert_stride, int64_t grad_stride,
                                                      uint64_t seed_stride, float* theta, float* m, float* v,
                                                      const float* grad, int64_t n_Q, int64_t n_pi,
                                                      const float* alpha_tab, const int64_t* step_ctr, int64_t tab_base,
                                                      int32_t tab_len, float beta1, float one_minus_beta1, float beta2,
                                                      float one_minus_beta2, float epsilon, const float* storage,
                                                      int64_t buf_stride, const curious_layout_t* L,
                                                      const curious_tasks_t* tasks, const curious_sample_params_t* P,
                                                      const curious_sample_rng_t* rng, int32_t n, float* batch,
                                                      const curious_batch_layout_t* BL, const curious_transposed_t* keep,
                                                      curious_stream_t stream) {
  CURIOUS_CHECK(n_experts >= 1 && n_experts <= 64, "curious_adam_update_and_sample_experts: n_experts must be in 1..64");
  CURIOUS_CHECK(n_experts == 1 || (expert_stride > 0 && expert_stride % 64 == 0 && grad_stride >= n_Q + n_pi),
                "curious_adam_update_and_sample_experts: bad strides");
  CURIOUS_CHECK(alpha_tab && step_ctr, "curious_adam_update_and_sample_experts: device step counter and step-size "
                                       "table are required");
  return adam_and_sample(n_experts, expert_stride, grad_stride, seed_stride, theta, m, v, grad, n_Q, n_pi, alpha_tab,
                         step_ctr, tab_base, tab_len, nullptr, beta1, one_minus_beta1, beta2, one_minus_beta2, epsilon,
                         storage, buf_stride, L, tasks, P, rng, n, batch, BL, keep, stream);
}

extern "C" int curious_adam_update(float* theta, float* m, float* v, const float* grad, int64_t n_Q, int64_t n_pi,
                                   const float* alpha_tab, const int64_t* step_ctr, int64_t tab_base,
                                   int32_t tab_len, const float* alpha_host, float beta1, float one_minus_beta1,
                                   float beta2, float one_minus_beta2, float epsilon,
                                   const curious_transposed_t* keep, curious_stream_t stream) {
  AdamArgs a;
  if (fill_adam(a, theta, m, v, grad, n_Q, n_pi, alpha_tab, step_ctr, tab_base, tab_len, alpha_host, beta1,
                one_minus_beta1, beta2, one_minus_beta2, epsilon, keep)) return -1;
  if (a.n <= 0) return 0;
  const bool vec4 = adam_vec4(a, 0, 0);
  int blocks = (int)((a.n + (vec4 ? 1023 : 255)) / (vec4 ? 1024 : 256));
  if (blocks > 2048) blocks = 2048;
  const int n_tile = adam_tiles(a);
  { ProfScope ps__(CK_ADAM, as_stream(stream));
    hipLaunchKernelGGL(adam_kernel, dim3(n_tile + blocks), dim3(256), 0, as_stream(stream), a, n_tile, vec4 ? 1 : 0); }
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

// 64 workgroups, 16 bytes per lane; the wave sums meet in LDS and ONE thread per workgroup adds the workgroup's pair to
// the result: 64 atomics per word per launch.  (Round 4: 1 024 workgroups x 4 waves x 2 same-address 64-bit atomics made
// this 1.18 MB read a 100 us kernel -- 2 % of every several-rank cycle.)  Integer add / xor commute: the result does not
// depend on the order, i.e. it is deterministic and the same as the old kernel's.
__device__ __forceinline__ unsigned long long checksum_hash(uint32_t bits, int64_t i) {
  unsigned long long h = ((unsigned long long)bits + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1));
  h ^= h >> 31;
  h *= 0xBF58476D1CE4E5B9ull;
  h ^= h >> 29;
  return h;
}
#define CHECKSUM_BLOCKS 64
__global__ __launch_bounds__(256) void checksum_kernel(const uint32_t* __restrict__ bits, int64_t n,
                                                      unsigned long long* __restrict__ out) {
  __shared__ unsigned long long part[2][4];
  unsigned long long s = 0, x = 0;
  const int64_t n4 = (((uintptr_t)bits & 15) == 0) ? (n >> 2) : 0;
  const uint4* b4 = reinterpret_cast<const uint4*>(bits);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const uint4 w = b4[i];
    const unsigned long long h0 = checksum_hash(w.x, 4 * i), h1 = checksum_hash(w.y, 4 * i + 1);
    const unsigned long long h2 = checksum_hash(w.z, 4 * i + 2), h3 = checksum_hash(w.w, 4 * i + 3);
    s += (h0 + h1) + (h2 + h3);
    x ^= (h0 ^ h1) ^ (h2 ^ h3);
  }
  for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long h = checksum_hash(bits[i], i);
    s += h;
    x ^= h;
  }
  for (int off = 32; off >= 1; off >>= 1) {
    s += __shfl_xor(s, off);
    x ^= __shfl_xor(x, off);
  }
  if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = s; part[1][threadIdx.x >> 6] = x; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&out[0], (part[0][0] + part[0][1]) + (part[0][2] + part[0][3]));
    atomicXor(&out[1], (part[1][0] ^ part[1][1]) ^ (part[1][2] ^ part[1][3]));
  }
}

extern "C" int curious_param_checksum(const float* theta, int64_t n, uint64_t* out, curious_stream_t stream) {
  CURIOUS_CHECK(theta && out, "curious_param_checksum: NULL argument");
  hipError_t e = hipMemsetAsync(out, 0, 2 * sizeof(uint64_t), as_stream(stream));
  CURIOUS_CHECK(e == hipSuccess, "curious_param_checksum: memset failed: %s", hipGetErrorString(e));
  if (n <= 0) return 0;
  int blocks = (int)((n + 1023) / 1024);
  if (blocks > CHECKSUM_BLOCKS) blocks = CHECKSUM_BLOCKS;
  { ProfScope ps__(CK_CHECKSUM, as_stream(stream)); hipLaunchKernelGGL(checksum_kernel, dim3(blocks), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const uint32_t*>(theta), n, reinterpret_cast<unsigned long long*>(out)); }
  CURIOUS_LAUNCH_CHECK("checksum_kernel");
  return 0;
}
