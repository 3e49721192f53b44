// Prototype (verdict r2 item 4c, DESIGN section 6 "next step"): the gradient all-reduce of the several-rank update as ONE
// hand-written kernel over IPC-mapped peer buffers, fused with the optimiser -- reduce-scatter + Adam + all-gather:
//
//   rank r owns slice r of the parameter vector (P / N elements).  After its gradient launches it runs ONE kernel that
//     1. tells every peer "my gradients of step t are complete"          (flag word in the peer's memory)
//     2. waits for the same word from every peer, then reads slice r of EVERY rank's gradient vector straight out of the
//        peers' buffers (xGMI reads on a multi-GPU node), adds them in rank order 0..N-1 (the same order on every rank:
//        replicas stay bit-identical), applies Adam to slice r -- m, v exist only for the owned slice --
//     3. writes the new theta slice into EVERY rank's parameter vector (xGMI writes), then tells every peer "slice r of
//        step t has landed", and
//     4. waits until all N slices of its own vector have landed.
//   Two hops on the full xGMI mesh, the optimiser's work divided by N, no separate optimiser launch: the
//   B launch of DESIGN section 6 disappears into the collective.  (mpi_adam.py:21-35: Allreduce(SUM) then Adam.)
//
// This program is the FUNCTIONAL test the development boxes allow: N processes share ONE GPU, map each other's buffers
// through hipIpc handles, run K steps and compare every rank's parameters with a host computation of SUM-in-rank-order +
// Adam (exact: the same single-rounding float operations) -- and with each other.  Wire time cannot be measured here
// (all "peers" are the same HBM); the per-step time printed is the latency of the protocol itself on one device.
//   hipcc --offload-arch=gfx950 -O3 tools/ipc_allreduce_lab.hip -o tools/ipc_allreduce_lab && tools/ipc_allreduce_lab [N]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("rank %d: HIP error %s at line %d\n", g_rank, hipGetErrorString(e), __LINE__); fflush(stdout); _exit(1);} } while (0)
static int g_rank = -1;

#define MAXR 8
#define SPIN_MAX (1 << 26)

struct Peers {
  const float* grad[MAXR];            // every rank's gradient vector (mine included)
  float* theta[MAXR];                 // every rank's parameter vector
  unsigned int* flags[MAXR];          // every rank's flag block: [2][MAXR] words (ready[from], landed[from])
};
struct Args {
  Peers p;
  float* m; float* v;                 // moments of the owned slice only [slice]
  unsigned int* done;                 // local: blocks that have finished writing their part of the slice
  int* err;
  int rank, world, n, slice;          // n = padded parameter count, slice = n / world
  unsigned int step;                  // 1, 2, ...
  float alpha, b1, omb1, b2, omb2, eps;
};

__device__ __forceinline__ void sys_store(unsigned int* p, unsigned int v) {
  __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ bool sys_wait(const unsigned int* p, unsigned int v, int* err) {
  int spins = 0;
  while (__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < v) {
    if (++spins > SPIN_MAX) { *err = 1; return false; }
    __builtin_amdgcn_s_sleep(2);
  }
  return true;
}

// grid: any number of blocks of 256 threads; every block takes a strided share of the owned slice
__global__ __launch_bounds__(256) void allreduce_adam_kernel(Args a) {
  const int tid = threadIdx.x;
  unsigned int* my_flags = a.p.flags[a.rank];
  // 1. my gradients are complete (they were written by earlier launches of this stream)
  if (blockIdx.x == 0 && tid < a.world) sys_store(a.p.flags[tid] + a.rank, a.step);
  // 2. every rank's gradients are complete
  if (tid < a.world) (void)sys_wait(my_flags + tid, a.step, a.err);
  __syncthreads();
  const int s0 = a.rank * a.slice;
  for (int i = blockIdx.x * 256 + tid; i < a.slice; i += gridDim.x * 256) {
    float g = 0.f;
    for (int r = 0; r < a.world; ++r)                       // rank order: the same sum on every rank
      g = __fadd_rn(g, __builtin_nontemporal_load(a.p.grad[r] + s0 + i));
    float m = a.m[i], v = a.v[i];
    m = __fadd_rn(__fmul_rn(a.b1, m), __fmul_rn(a.omb1, g));                        // mpi_adam.py:31
    v = __fadd_rn(__fmul_rn(a.b2, v), __fmul_rn(a.omb2, __fmul_rn(g, g)));          // mpi_adam.py:32
    const float th = a.p.theta[a.rank][s0 + i];
    const float nt = __fadd_rn(th, __fmul_rn(-a.alpha, m) / __fadd_rn(sqrtf(v), a.eps));   // mpi_adam.py:33-34
    a.m[i] = m; a.v[i] = v;
    // 3. the new slice goes into every rank's vector
    for (int r = 0; r < a.world; ++r) __builtin_nontemporal_store(nt, a.p.theta[r] + s0 + i);
  }
  // ... and once ALL blocks of this rank have written, the peers are told
  __threadfence_system();
  __syncthreads();
  __shared__ unsigned int last;
  if (tid == 0) last = (atomicAdd(a.done, 1u) == gridDim.x * a.step - 1u) ? 1u : 0u;
  __syncthreads();
  if (last && tid < a.world) sys_store(a.p.flags[tid] + MAXR + a.rank, a.step);
  // 4. all slices of my vector have landed
  if (tid < a.world) (void)sys_wait(my_flags + MAXR + tid, a.step, a.err);
}

__global__ void fill_grad(float* g, int n, int rank, unsigned int step) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const unsigned int h = (unsigned int)i * 2654435761u + (unsigned int)rank * 40503u + step * 7919u;
    g[i] = ((float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.01f;
  }
}

struct Shared {                       // host memory shared by the processes (created before fork)
  volatile int arrived[64];
  hipIpcMemHandle_t grad[MAXR], theta[MAXR], flags[MAXR];
  float theta_out[MAXR][64];
  double us_per_step[MAXR];
  int ok[MAXR];
};
static void barrier(Shared* sh, int world, int phase) {
  __sync_fetch_and_add(&sh->arrived[phase], 1);
  while (sh->arrived[phase] < world) usleep(100);
}

static int child(int rank, int world, Shared* sh, int n, int steps) {
  g_rank = rank;
  CK(hipSetDevice(0));
  float *grad, *theta, *m, *v;
  unsigned int *flags, *done;
  int* err;
  const int slice = n / world;
  CK(hipMalloc(&grad, n * 4)); CK(hipMalloc(&theta, n * 4)); CK(hipMalloc(&m, slice * 4)); CK(hipMalloc(&v, slice * 4));
  CK(hipMalloc(&flags, 2 * MAXR * 4)); CK(hipMalloc(&done, 4)); CK(hipMalloc(&err, 4));
  CK(hipMemset(flags, 0, 2 * MAXR * 4)); CK(hipMemset(done, 0, 4)); CK(hipMemset(err, 0, 4));
  CK(hipMemset(m, 0, slice * 4)); CK(hipMemset(v, 0, slice * 4));
  std::vector<float> h_theta(n);
  for (int i = 0; i < n; ++i) h_theta[i] = 0.001f * (float)((i * 37) % 101) - 0.05f;      // identical on every rank (C3)
  CK(hipMemcpy(theta, h_theta.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipIpcGetMemHandle(&sh->grad[rank], grad));
  CK(hipIpcGetMemHandle(&sh->theta[rank], theta));
  CK(hipIpcGetMemHandle(&sh->flags[rank], flags));
  CK(hipDeviceSynchronize());
  barrier(sh, world, 0);
  Args a;
  memset(&a, 0, sizeof(a));
  for (int r = 0; r < world; ++r) {
    if (r == rank) { a.p.grad[r] = grad; a.p.theta[r] = theta; a.p.flags[r] = flags; continue; }
    void *pg, *pt, *pf;
    CK(hipIpcOpenMemHandle(&pg, sh->grad[r], hipIpcMemLazyEnablePeerAccess));
    CK(hipIpcOpenMemHandle(&pt, sh->theta[r], hipIpcMemLazyEnablePeerAccess));
    CK(hipIpcOpenMemHandle(&pf, sh->flags[r], hipIpcMemLazyEnablePeerAccess));
    a.p.grad[r] = (const float*)pg; a.p.theta[r] = (float*)pt; a.p.flags[r] = (unsigned int*)pf;
  }
  a.m = m; a.v = v; a.done = done; a.err = err; a.rank = rank; a.world = world; a.n = n; a.slice = slice;
  a.b1 = 0.9f; a.omb1 = 1.0f - 0.9f; a.b2 = 0.999f; a.omb2 = 1.0f - 0.999f; a.eps = 1e-8f;
  barrier(sh, world, 1);
  // host model of the same steps: SUM over ranks in rank order, then Adam, element by element in float32
  std::vector<float> hm(n, 0.f), hv(n, 0.f);
  const int blocks = 64;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms_total = 0.f;
  for (int t = 1; t <= steps; ++t) {
    hipLaunchKernelGGL(fill_grad, dim3(256), dim3(256), 0, 0, grad, n, rank, (unsigned int)t);
    a.step = (unsigned int)t;
    a.alpha = (float)(1e-3 * sqrt(1.0 - pow(0.999, t)) / (1.0 - pow(0.9, t)));            // mpi_adam.py:30
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(allreduce_adam_kernel, dim3(blocks), dim3(256), 0, 0, a);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (t > 2) ms_total += ms;
    for (int i = 0; i < n; ++i) {
      float g = 0.f;
      for (int r = 0; r < world; ++r) {
        const unsigned int h = (unsigned int)i * 2654435761u + (unsigned int)r * 40503u + (unsigned int)t * 7919u;
        g = g + ((float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.01f;
      }
      hm[i] = 0.9f * hm[i] + (1.0f - 0.9f) * g;
      hv[i] = 0.999f * hv[i] + (1.0f - 0.999f) * (g * g);
      h_theta[i] = h_theta[i] + (-a.alpha * hm[i]) / (sqrtf(hv[i]) + 1e-8f);
    }
  }
  std::vector<float> out(n);
  CK(hipMemcpy(out.data(), theta, n * 4, hipMemcpyDeviceToHost));
  int herr = 0;
  CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < n; ++i) bad += (out[i] != h_theta[i]);
  for (int i = 0; i < 64; ++i) sh->theta_out[rank][i] = out[(size_t)i * (n / 64)];
  sh->us_per_step[rank] = 1e3 * ms_total / (steps - 2);
  sh->ok[rank] = (bad == 0 && herr == 0) ? 1 : 0;
  if (bad || herr) printf("rank %d: %d of %d parameters differ from the host model%s\n", rank, bad, n, herr ? ", a wait timed out" : "");
  barrier(sh, world, 2);               // nobody unmaps / frees while a peer may still read
  for (int r = 0; r < world; ++r)
    if (r != rank) { (void)hipIpcCloseMemHandle((void*)a.p.grad[r]); (void)hipIpcCloseMemHandle(a.p.theta[r]); (void)hipIpcCloseMemHandle(a.p.flags[r]); }
  barrier(sh, world, 3);
  return sh->ok[rank] ? 0 : 1;
}

int main(int argc, char** argv) {
  const int world = argc > 1 ? atoi(argv[1]) : 2;
  const int steps = 12;
  const int n = 294912;                // P of the Arm4 agent (294 784) rounded up to a multiple of 8 * 256
  if (world < 1 || world > MAXR || n % world) { printf("bad world size\n"); return 2; }
  Shared* sh = (Shared*)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  memset(sh, 0, sizeof(Shared));
  std::vector<pid_t> kids;
  for (int r = 0; r < world; ++r) {    // fork BEFORE anything touches the GPU
    pid_t p = fork();
    if (p == 0) _exit(child(r, world, sh, n, steps));
    kids.push_back(p);
  }
  int fail = 0;
  for (pid_t p : kids) { int st = 0; waitpid(p, &st, 0); fail |= !(WIFEXITED(st) && WEXITSTATUS(st) == 0); }
  bool same = true;
  for (int r = 1; r < world; ++r) same = same && memcmp(sh->theta_out[0], sh->theta_out[r], sizeof(sh->theta_out[0])) == 0;
  printf("%d ranks on one GPU, %d parameters, %d steps of [reduce-scatter over IPC-mapped gradients + Adam on the owned "
         "slice + all-gather of the new slices] in one kernel: %s; replicas %s; %.1f us per step (protocol latency on one "
         "device, no wire)\n", world, n, steps, fail ? "MISMATCH" : "every rank == host model of SUM(rank order) + Adam, bit for bit",
         same ? "identical" : "DIFFER", sh->us_per_step[0]);
  printf("%s\n", (!fail && same) ? "OK" : "FAILED");
  return (!fail && same) ? 0 : 1;
}
