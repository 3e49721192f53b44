// The gradient all-reduce of the several-rank update as ONE hand-written kernel per rank over peer-mapped buffers, fused
// with the optimiser: reduce-scatter + Adam + all-gather (curious_allreduce_adam_ipc; product form of the round-3
// prototype tools/ipc_allreduce_lab.hip).
//
// Replaces (reference): MpiAdam.update = Allreduce(SUM) of the flat gradient + Adam on every rank, twice per update
// (mpi_adam.py:21-35 via ddpg.py:246-248), and of this build's default several-rank path the RCCL all-reduce + the
// stand-alone optimiser launch (optim.hip adam_kernel).
//
//   Rank r owns slice r of the fused parameter vector (P / N elements).  After its gradient launches it runs this kernel:
//     1. "my gradients of epoch t are complete" -> a word in every peer's flag block;
//     2. wait for the same word from every peer; read slice r of EVERY rank's gradient vector out of the peers' memory
//        (xGMI reads on a multi-GPU node), add them in rank order 0 .. N-1 -- the same order on every rank, so the
//        replicas stay bit-identical (mpi_adam.py:42-50) --, apply Adam to slice r (the moments of the other slices are
//        never touched on this rank: the optimiser's work divides by N);
//     3. write the new slice into EVERY rank's STAGING vector, then "slice r of epoch t has landed" -> every peer;
//     4. wait until all N slices of the own staging vector have landed (which also means: every peer has finished reading
//        the own gradient vector -- the next gradient launch may overwrite it); copy the staging vector into the local
//        parameter vector;
//     5. rebuild the transposed copies of the hidden matrices the row-local backward layers read (mlp_rows.h) from the
//        new parameters (read from the staging vector), so the next gradient launch may be told params_unchanged.
//   Two hops on the fully connected xGMI mesh instead of a ring's 2 (N - 1); no separate optimiser launch.
//
// Memory types (round 5).  Everything a PEER reads or writes while a kernel of the owner may be running -- the gradient
// vector, the staging vector, the flag block -- lives in one FINE-GRAINED allocation (hipExtMallocWithFlags,
// hipDeviceMallocFinegrained; what RCCL uses for its own peer buffers): HIP promises cross-device visibility of
// coarse-grained memory (plain hipMalloc) only at kernel boundaries, and a device's L2 may keep a stale line of local
// coarse-grained memory that a peer rewrote over xGMI.  The parameter vector itself stays an ordinary (coarse-grained)
// local allocation that no peer ever touches -- it is what the gradient launches stream 12 times per update --: the new
// slices arrive in the staging vector and this rank's own kernel copies them over (step 4), through its own L2.
//
// Tokens.  The hand-shake word of an epoch is a device-resident EPOCH counter private to this kernel (read at its top,
// advanced by its last block), strictly increasing over the job's life, so the flag blocks never need a reset -- NOT the
// Adam step counter, which DDPG.train_batches_guarded rewinds when it replays a faulted run of updates (a replayed wait
// for a token the peers' flags had already passed would fall through before their gradients exist).
//
// The hand-off guard of the row-local update is collective here as well: every rank reads the fault-flag element of
// EVERY rank's gradient vector (curious_transposed_t.fault_flag) behind step 2's wait and all of them skip the
// arithmetic -- but not the signalling -- when any is set.
//
// A wait that exceeds `spins` polls gives up: *err is raised and the rank treats the epoch like a faulted update -- no
// arithmetic, no copy, no rebuilt copies, the signalling goes on so that no peer hangs -- and the host raises at the end
// of the run of updates (DDPG._train_ranks_ipc).
// What a single-GPU box can validate is validated (tests/test_gpu_round4.py: 2 and 4 processes on ONE GPU, buffers
// mapped through hipIpc handles: replicas identical, == the gloo path at 2 ranks); wire time and behaviour under real
// xGMI ordering need a multi-GPU node.  RCCL stays the default (DDPG(_allreduce='ipc') opts in).
#include "common.h"

#define IPC_MAXR CURIOUS_IPC_MAX_RANKS
#define IPC_TILE 32

struct IpcArgs {
  curious_ipc_peers_t p;
  float* m; float* v;                 // local moment vectors (only the owned slice is used)
  int64_t n_Q, n;
  const float* alpha_tab; const int64_t* step_ctr; int64_t tab_base; int32_t tab_len;
  float b1, omb1, b2, omb2, eps;
  float* theta;                       // local parameter vector (ordinary device memory; nobody else touches it)
  uint32_t* epoch;                    // local: epochs completed so far (the token of this launch is *epoch + 1)
  uint32_t* done;                     // local: blocks that have finished writing their share of the slice (monotone)
  int32_t* err;                       // local: a wait gave up
  int32_t spins;
  curious_transposed_t keep;          // local transposed copies + fault word + flag index
};

__device__ __forceinline__ void ipc_store(uint32_t* p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// tokens wrap after 2^32 updates: compared as signed differences
__device__ __forceinline__ bool ipc_wait(const uint32_t* p, uint32_t v, int32_t spins, int32_t* err) {
  int32_t k = 0;
  while ((int32_t)(__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - v) < 0) {
    if (++k > spins) { *err = 1; return false; }
    __builtin_amdgcn_s_sleep(2);
  }
  return true;
}

__global__ __launch_bounds__(256) void allreduce_adam_ipc_kernel(IpcArgs a) {
  __shared__ float tile[IPC_TILE][IPC_TILE + 1];
  __shared__ uint32_t s_last, s_fault;
  const int tid = threadIdx.x, world = a.p.world, rank = a.p.rank;
  // (every block reads the epoch before it adds itself to `done`; the block that completes `done` advances it)
  const uint32_t tok = __hip_atomic_load(a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
  const int64_t step = *a.step_ctr;
  uint32_t* my_flags = a.p.flags[rank];
  if (tid == 0) s_fault = 0;
  __syncthreads();
  // 1. my gradients are complete (written by earlier launches of this stream; release at system scope)
  if (blockIdx.x == 0 && tid < world) ipc_store(a.p.flags[tid] + rank, tok);
  // 2. every rank's gradients are complete (a wait that gives up: the epoch is treated like a faulted update)
  if (tid < world && !ipc_wait(my_flags + tid, tok, a.spins, a.err)) atomicOr(&s_fault, 2u);
  __syncthreads();
  // the collective verdict of the hand-off guard: any rank's flag element, or the local (sticky) word
  if (a.keep.fault) {
    if (tid < world && a.keep.fault_flag > 0 &&
        __builtin_nontemporal_load(a.p.grad[tid] + a.keep.fault_flag - 1) != 0.0f) atomicOr(&s_fault, 1u);
    if (tid == 0 && *a.keep.fault != 0) atomicOr(&s_fault, 1u);
  }
  __syncthreads();
  const bool faulted = s_fault != 0;
  if ((s_fault & 1u) && blockIdx.x == 0 && tid == 0 && *a.keep.fault == 0)
    atomicAdd(const_cast<int32_t*>(a.keep.fault), 1);
  const int64_t slice = a.n / world, s0 = (int64_t)rank * slice;
  if (!faulted) {
    // step sizes of this update (optim.hip adam_alphas: the ring is indexed by the counter the gradient launches advanced)
    int64_t idx = (step - 1 - a.tab_base) % a.tab_len;
    if (idx < 0) idx += a.tab_len;
    const float aQ = a.alpha_tab[2 * idx], aPi = a.alpha_tab[2 * idx + 1];
    for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < slice; i += (int64_t)gridDim.x * 256) {
      const int64_t e = s0 + i;
      float g = 0.f;
      for (int r = 0; r < world; ++r)                         // rank order: the same sum on every rank
        g = __fadd_rn(g, __builtin_nontemporal_load(a.p.grad[r] + e));
      float m = a.m[e], v = a.v[e];
      m = __fadd_rn(__fmul_rn(a.b1, m), __fmul_rn(a.omb1, g));                              // mpi_adam.py:31
      v = __fadd_rn(__fmul_rn(a.b2, v), __fmul_rn(a.omb2, __fmul_rn(g, g)));                // mpi_adam.py:32
      const float na = (e < a.n_Q) ? -aQ : -aPi;
      const float th = a.theta[e];
      const float nt = __fadd_rn(th, fdiv(__fmul_rn(na, m), __fadd_rn(sqrtf(v), a.eps)));   // mpi_adam.py:33-34
      a.m[e] = m; a.v[e] = v;
      // 3. the new slice goes into every rank's staging vector
      for (int r = 0; r < world; ++r) __builtin_nontemporal_store(nt, a.p.stage[r] + e);
    }
  }
  // ... and once ALL blocks of this rank have written, the peers are told
  __threadfence_system();
  __syncthreads();
  if (tid == 0) s_last = ((atomicAdd(a.done, 1u) + 1u) % gridDim.x == 0u) ? 1u : 0u;
  __syncthreads();
  if (s_last && tid == 0) __hip_atomic_store(a.epoch, tok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (s_last && tid < world) ipc_store(a.p.flags[tid] + IPC_MAXR + rank, tok);
  // 4. all slices of my staging vector have landed
  if (tid < world && !ipc_wait(my_flags + IPC_MAXR + tid, tok, a.spins, a.err)) atomicOr(&s_fault, 2u);
  __syncthreads();
  // (a rank that faulted wrote no slice: every rank reads every rank's verdict in step 2, all of them skip alike.  A wait
  //  that gave up is local: this rank keeps its old parameters and the host raises)
  if (faulted || s_fault != 0) return;
  // ... and become the local parameters (16 bytes per lane; the staging vector is read past the caches)
  const float* stg = a.p.stage[rank];
  for (int64_t i = ((int64_t)blockIdx.x * 256 + tid) * 4; i < a.n; i += (int64_t)gridDim.x * 256 * 4) {
    float4 v4;
    v4.x = __builtin_nontemporal_load(stg + i); v4.y = __builtin_nontemporal_load(stg + i + 1);
    v4.z = __builtin_nontemporal_load(stg + i + 2); v4.w = __builtin_nontemporal_load(stg + i + 3);
    *reinterpret_cast<float4*>(a.theta + i) = v4;
  }
  if (a.keep.n == 0) return;
  // 5. the transposed copies of the kept matrices from the new parameters: WT[n][k] = W[k][n], 32 x 32 tiles through LDS
  const int dim = a.keep.dim, per = dim / IPC_TILE, ntile = a.keep.n * per * per;
  const float* th = stg;
  const int c = tid & 31, r8 = tid >> 5;
  for (int tb = blockIdx.x; tb < ntile; tb += gridDim.x) {
    const int j = tb / (per * per), t = tb - j * per * per;
    const int k0 = (t / per) * IPC_TILE, n0 = (t % per) * IPC_TILE;
    const float* src = th + a.keep.src_off[j];
#pragma unroll
    for (int i = 0; i < IPC_TILE / 8; ++i)
      tile[8 * i + r8][c] = __builtin_nontemporal_load(src + (int64_t)(k0 + 8 * i + r8) * dim + n0 + c);
    __syncthreads();
    float* dst = a.keep.dst[j];
#pragma unroll
    for (int i = 0; i < IPC_TILE / 8; ++i) dst[(int64_t)(n0 + 8 * i + r8) * dim + k0 + c] = tile[c][8 * i + r8];
    __syncthreads();
  }
}

extern "C" int curious_allreduce_adam_ipc(const curious_ipc_peers_t* peers, float* theta, float* m, float* v, int64_t n_Q,
                                          int64_t n_pi, const float* alpha_tab, const int64_t* step_ctr, int64_t tab_base,
                                          int32_t tab_len, float beta1, float one_minus_beta1, float beta2,
                                          float one_minus_beta2, float epsilon, uint32_t* epoch, uint32_t* done,
                                          int32_t* err, int32_t spins, const curious_transposed_t* keep,
                                          curious_stream_t stream) {
  CURIOUS_CHECK(peers && theta && m && v && alpha_tab && step_ctr && epoch && done && err && tab_len > 0,
                "curious_allreduce_adam_ipc: NULL argument");
  CURIOUS_CHECK(peers->world >= 1 && peers->world <= IPC_MAXR && peers->rank >= 0 && peers->rank < peers->world,
                "curious_allreduce_adam_ipc: world must be in 1..%d, rank inside it", IPC_MAXR);
  const int64_t n = n_Q + n_pi;
  CURIOUS_CHECK(n > 0 && n % peers->world == 0 && n % 4 == 0 && ((uintptr_t)theta & 15) == 0,
                "curious_allreduce_adam_ipc: the parameter count must divide by the world size (and by 4)");
  for (int r = 0; r < peers->world; ++r)
    CURIOUS_CHECK(peers->grad[r] && peers->stage[r] && peers->flags[r], "curious_allreduce_adam_ipc: rank %d is not mapped", r);
  IpcArgs a;
  memset(&a, 0, sizeof(a));
  a.p = *peers;
  a.theta = theta; a.epoch = epoch;
  a.m = m; a.v = v; a.n_Q = n_Q; a.n = n;
  a.alpha_tab = alpha_tab; a.step_ctr = step_ctr; a.tab_base = tab_base; a.tab_len = tab_len;
  a.b1 = beta1; a.omb1 = one_minus_beta1; a.b2 = beta2; a.omb2 = one_minus_beta2; a.eps = epsilon;
  a.done = done; a.err = err; a.spins = spins > 0 ? spins : (1 << 26);
  if (keep) {
    CURIOUS_CHECK(keep->n >= 0 && keep->n <= 8 && (keep->n == 0 || (keep->dim > 0 && keep->dim % IPC_TILE == 0)),
                  "curious_allreduce_adam_ipc: bad description of the transposed copies");
    a.keep = *keep;
  }
  // 64 workgroups: all resident at once on any partition of the device (they wait for each other across ranks)
  { ProfScope ps__(CK_IPC, as_stream(stream));
    hipLaunchKernelGGL(allreduce_adam_ipc_kernel, dim3(64), dim3(256), 0, as_stream(stream), a); }
  CURIOUS_LAUNCH_CHECK("allreduce_adam_ipc_kernel");
  return 0;
}

// ---- set-up / tear-down of the peer mappings (host side; NOT enqueue-only like the rest of the ABI: called once per job).
// The shared vectors come from the HIP runtime directly, not from the framework's caching allocator: an allocator block
// that was handed out through its own IPC machinery carries reference-counting state that outlives the job's tear-down.
// FINE-GRAINED device memory (see the head of this file): peers poll, read and write the block while kernels of the owner
// run.  CURIOUS_IPC_COARSE=1 falls back to plain hipMalloc (A/B of what the memory type costs; one-GPU functional tests).
extern "C" int curious_ipc_alloc(int64_t bytes, void** out) {
  CURIOUS_CHECK(out && bytes > 0, "curious_ipc_alloc: bad argument");
  void* p = nullptr;
  const char* coarse = getenv("CURIOUS_IPC_COARSE");
  if (coarse && coarse[0] == '1') {
    CURIOUS_CHECK(hipMalloc(&p, (size_t)bytes) == hipSuccess, "curious_ipc_alloc: hipMalloc of %lld bytes failed", (long long)bytes);
  } else {
    CURIOUS_CHECK(hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained) == hipSuccess,
                  "curious_ipc_alloc: hipExtMallocWithFlags(hipDeviceMallocFinegrained) of %lld bytes failed", (long long)bytes);
  }
  CURIOUS_CHECK(hipMemset(p, 0, (size_t)bytes) == hipSuccess && hipDeviceSynchronize() == hipSuccess,
                "curious_ipc_alloc: clearing the block failed");
  *out = p;
  return 0;
}
extern "C" int curious_ipc_free(void* p) {
  CURIOUS_CHECK(!p || hipFree(p) == hipSuccess, "curious_ipc_free: hipFree failed");
  return 0;
}
extern "C" int curious_ipc_export(const void* p, unsigned char* handle64) {
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
  CURIOUS_CHECK(p && handle64, "curious_ipc_export: NULL argument");
  hipIpcMemHandle_t h;
  CURIOUS_CHECK(hipIpcGetMemHandle(&h, const_cast<void*>(p)) == hipSuccess, "curious_ipc_export: hipIpcGetMemHandle failed "
                "(HSA_ENABLE_IPC_MODE_LEGACY=0 is required on this stack)");
  memcpy(handle64, &h, 64);
  return 0;
}
extern "C" int curious_ipc_import(const unsigned char* handle64, void** out) {
  CURIOUS_CHECK(handle64 && out, "curious_ipc_import: NULL argument");
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, 64);
  void* p = nullptr;
  CURIOUS_CHECK(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) == hipSuccess,
                "curious_ipc_import: hipIpcOpenMemHandle failed");
  *out = p;
  return 0;
}
extern "C" int curious_ipc_close(void* p) {
  CURIOUS_CHECK(!p || hipIpcCloseMemHandle(p) == hipSuccess, "curious_ipc_close: hipIpcCloseMemHandle failed");
  return 0;
}
