// Micro-lab for the row-local DDPG pass (curious_amd/csrc/mlp_rows.h): runs ddpg_rows_kernel on synthetic Arm4-sized
// networks, prints the launch time (weights left warm in L2 / rewritten before every launch like Adam does) and the
// in-kernel s_memtime stamps of row group 0.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Icurious_amd/csrc tools/rows_lab.hip -o tools/rows_lab
#include <math.h>
#include <stdlib.h>
#include <vector>
#define ROWS_DEBUG 1
#include "common.h"
typedef float f32x4 __attribute__((ext_vector_type(4)));
#include "her_body.h"
#include "mlp_common.h"
#include "mlp_rows.h"
void curious_set_error(const char*, ...) {}
int g_curious_prof_on = 0;
void curious_prof_push(int, hipStream_t, bool) {}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void touch(float* p, size_t n) {   // rewrite the parameters (what the optimiser launch does)
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0f;
}

int main(int argc, char** argv) {
  const int B = (argc > 1) ? atoi(argv[1]) : 256;             // 256: one workgroup per CU; 512: two
  const int H = 256, O = 40, N = 4, G = 12, U = 4, nl = 3, ld = 152;
  const int Sa = O + N, Sc = Sa + U;
  auto net_size = [&](int S, int D) { return S * H + H + G * H + (nl - 1) * (H * H + H) + H * D + D; };
  const int PQ = net_size(Sc, 1), PP = net_size(Sa, U);
  const int offP = (PQ + 63) & ~63, total = offP + ((PP + 63) & ~63);
  std::vector<float> th(2 * (size_t)total), batch((size_t)B * ld);
  srand(3);
  for (auto& v : th) v = ((float)rand() / RAND_MAX - 0.5f) * 0.2f;
  for (auto& v : batch) v = ((float)rand() / RAND_MAX - 0.5f);
  float *dth, *dbatch, *ws;
  unsigned long long* dst;
  CK(hipMalloc(&dth, th.size() * 4)); CK(hipMalloc(&dbatch, batch.size() * 4));
  const size_t BH = (size_t)B * H;
  CK(hipMalloc(&ws, (4 * nl * BH + 16 * B + 4 * (size_t)H * H) * 4));
  CK(hipMemset(ws, 0xff, (4 * nl * BH + 16 * B) * 4));      // the hand-off words need no initialisation: start from garbage
  CK(hipMalloc(&dst, 1024 * 8)); CK(hipMemset(dst, 0, 1024 * 8));
  CK(hipMemcpy(dth, th.data(), th.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dbatch, batch.data(), batch.size() * 4, hipMemcpyHostToDevice));
  auto mk = [&](const float* base, int S, int D) {
    RowsNet n; memset(&n, 0, sizeof(n));
    int off = 0;
    n.th = base; n.W0 = off; off += S * H; n.b0 = off; off += H; n.Wg = off; off += G * H;
    for (int l = 1; l < nl; ++l) { n.W[l] = off; off += H * H; n.b[l] = off; off += H; }
    n.Wout = off; off += H * D; n.bout = off;
    return n;
  };
  RowsArgs a; memset(&a, 0, sizeof(a));
  a.mQ = mk(dth, Sc, 1); a.mPi = mk(dth + offP, Sa, U);
  a.tQ = mk(dth + total, Sc, 1); a.tPi = mk(dth + total + offP, Sa, U);
  a.batch = dbatch; a.ld = ld; a.off_o = 0; a.off_td = 40; a.off_u = 44; a.off_g = 48; a.off_o2 = 60; a.off_g2 = 100;
  a.off_r = 112;
  for (int l = 0; l < nl; ++l) {
    a.actc[l] = ws + (0 * nl + l) * BH; a.dactc[l] = ws + (1 * nl + l) * BH;
    a.acta[l] = ws + (2 * nl + l) * BH; a.dacta[l] = ws + (3 * nl + l) * BH;
  }
  float* tail = ws + 4 * nl * BH;
  a.dQ = tail; a.dz = tail + B; a.rows = tail + 5 * B; a.out_Qpi = tail + 8 * B;
  a.qt = reinterpret_cast<unsigned long long*>(tail + 10 * B);
  a.xmap = 1; a.B = B; a.nl = nl; a.dimo = O; a.dimtd = N; a.dimg = G;
  a.gamma = 0.98f; a.clip_lo = -50.f; a.clip_hi = 0.f; a.max_u = 1.f; a.l2c = 2.0f / (B * U);
  a.stamps = dst;
  Ex ex; ex.stride = 0; ex.nprob = 1; ex.zmul = 0;
  {                                                          // transposed copies of the main networks' hidden matrices
    RowsTransposeArgs t; memset(&t, 0, sizeof(t));
    float* wt = tail + 16 * B;
    int n = 0;
    for (int l = 1; l < nl; ++l) { t.src[n] = a.mQ.th + a.mQ.W[l]; t.dst[n] = wt + (size_t)n * H * H; a.wTq[l] = t.dst[n]; ++n; }
    for (int l = 1; l < nl; ++l) { t.src[n] = a.mPi.th + a.mPi.W[l]; t.dst[n] = wt + (size_t)n * H * H; a.wTpi[l] = t.dst[n]; ++n; }
    hipLaunchKernelGGL((rows_transpose_kernel<false>), dim3(16, n, 1), dim3(256), 0, 0, t, ex);
    CK(hipDeviceSynchronize());
  }
  const size_t lds = rows_lds_floats(ROWS_R, nl) * sizeof(float);
  // (the kernel also has ~1 KB of static LDS: dynamic + static must stay <= 160 KB)
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ddpg_rows_kernel<false>),
                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)(rows_lds_floats(ROWS_R, ROWS_MAXL) * sizeof(float))));
  dim3 grid(4 * (B / ROWS_R), 1, 1);
  printf("B = %d\n", B);
  // the kernel's leading arguments (mlp_rows.h RowsPre), as DdpgPass::launch_rows fills them; argv[2] = 0: flag clear
  const bool use_pre = !(argc > 2 && atoi(argv[2]) == 0);
  const float* pw0a = a.mPi.th + a.mPi.W0;
  const float* pw0t = a.tPi.th + a.tPi.W0;
  const float* pw0c = a.mQ.th + a.mQ.W0;
  const uint32_t k0 = (uint32_t)a.ld | ((uint32_t)a.off_o << 16), k1 = (uint32_t)a.off_td | ((uint32_t)a.off_g << 16);
  const uint32_t k2 = (uint32_t)a.off_o2 | ((uint32_t)a.off_g2 << 16);
  const uint32_t k3 = (uint32_t)a.off_u | ((uint32_t)a.dimo << 16) | ((uint32_t)a.dimtd << 24);
  const uint32_t k4 = use_pre ? ((uint32_t)B | ((uint32_t)a.dimg << 16) | (1u << 25)) : 0u, k5 = 0u;
  printf("leading arguments %s\n", use_pre ? "in use" : "off");
  auto launch = [&]() {
    hipLaunchKernelGGL((ddpg_rows_kernel<false>), grid, dim3(256), lds, 0, pw0a, pw0t, pw0c, a.batch, k0, k1, k2, k3, k4, k5,
                       a, ex);
  };
  for (int i = 0; i < 5; ++i) launch();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int cold = 0; cold < 2; ++cold) {
    float tot = 0.f;
    const int it = 100;
    for (int i = 0; i < it; ++i) {
      if (cold) hipLaunchKernelGGL(touch, dim3((2 * total + 255) / 256), dim3(256), 0, 0, dth, (size_t)2 * total);
      CK(hipEventRecord(e0, 0));
      launch();
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      tot += ms;
    }
    printf("%s weights: %.2f us per launch (event pair around one launch)\n", cold ? "rewritten" : "warm", tot * 1000.f / it);
    unsigned long long st[96];
    CK(hipMemcpy(st, dst, sizeof(st), hipMemcpyDeviceToHost));
    const char* names[3][10] = {{"start", "inputs", "L0 pi", "hidden pi", "head pi", "L0+hidden Q(pi)", "head + dY",
                                 "bwd Q(pi)", "dz + dY", "bwd pi"},
                                {"start", "inputs", "L0 tpi", "hidden tpi", "head pi'", "L0+hidden tQ", "head Q' + publish",
                                 "", "", ""},
                                {"start", "inputs", "L0 mQ", "hidden mQ", "head Q + wait for Q'", "loss + dY", "bwd mQ",
                                 "", "", ""}};
    const int nst[3] = {10, 7, 7};
    const char* kinds[3] = {"actor side ", "target     ", "main critic"};
    printf("  actor-side group 0: %llu cycles from the kernel's first instruction to its role and arguments known\n",
           st[0] - st[31]);
    for (int ty = 0; ty < 3; ++ty) {
      printf("  %s (shader cycles, from the actor group's start: %lld): ", kinds[ty], (long long)(st[ty * 32] - st[0]));
      for (int k = 1; k < nst[ty]; ++k) printf("%s %llu | ", names[ty][k], st[ty * 32 + k] - st[ty * 32 + k - 1]);
      printf("total %llu\n", st[ty * 32 + nst[ty] - 1] - st[ty * 32]);
    }
    {
      unsigned long long d[256];
      CK(hipMemcpy(d, dst + 96, sizeof(d), hipMemcpyDeviceToHost));
      printf("  actor side, end of each forward hidden layer, deltas: ");
      for (int k = 1; k < 40 && d[k]; ++k) printf(" %llu", d[k] - d[k - 1]);
      printf("\n");
    }
  }
  return 0;
}
