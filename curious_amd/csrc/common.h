// Shared device/host helpers for libcurious_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/curious_hip.h"

#define CURIOUS_WAVE 64

void curious_set_error(const char* fmt, ...);

#define CURIOUS_CHECK(cond, ...)          \
  do {                                    \
    if (!(cond)) {                        \
      curious_set_error(__VA_ARGS__);     \
      return -1;                          \
    }                                     \
  } while (0)

#define CURIOUS_LAUNCH_CHECK(name)                                                   \
  do {                                                                               \
    hipError_t e__ = hipGetLastError();                                              \
    if (e__ != hipSuccess) {                                                         \
      curious_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));      \
      return -2;                                                                     \
    }                                                                                \
  } while (0)

// ---------------------------------------------------------------- Philox4x32-10 (matches oracle/env.py)
struct Philox4 {
  uint32_t x, y, z, w;
};

__host__ __device__ inline Philox4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)M0 * c0;
    uint64_t p1 = (uint64_t)M1 * c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0;
    uint32_t n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += W0; k1 += W1;
  }
  Philox4 o = {c0, c1, c2, c3};
  return o;
}

// uint32 -> float32 uniform in [0,1): top 24 bits * 2^-24 (exact)
__host__ __device__ inline float u01_f32(uint32_t r) { return (float)(r >> 8) * 5.9604644775390625e-08f; }
// two uint32 -> float64 uniform in [0,1) with 53 bits (NumPy's random_sample recipe)
__host__ __device__ inline double u01_f64(uint32_t a, uint32_t b) {
  return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

// non-contracted float32 arithmetic (bit parity with NumPy: one rounding per operation)
__device__ inline float fmul(float a, float b) { return __fmul_rn(a, b); }
__device__ inline float fadd(float a, float b) { return __fadd_rn(a, b); }
__device__ inline float fsub(float a, float b) { return __fsub_rn(a, b); }
// IEEE division / sqrt: hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt makes `/` and sqrtf correctly
// rounded; the __fdiv_rn / __fsqrt_rn intrinsics lower to native (approximate) instructions on AMD.
__device__ inline float fdiv(float a, float b) { return a / b; }
__device__ inline float fclip(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

static inline hipStream_t as_stream(curious_stream_t s) { return (hipStream_t)s; }

// ---------------------------------------------------------------- per-kernel HIP-event timing (bench.py roofline)
enum {
  CK_HER_SAMPLE = 0, CK_STORE, CK_ACTIVITY, CK_NORM_PARTIAL, CK_NORM_FINAL, CK_NORM_RECOMPUTE, CK_NORM_PAIR_PARTIAL,
  CK_NORM_PAIR_FINAL, CK_FWD_LAYER0, CK_FWD_GENERIC, CK_FWD_LAYER, CK_DX, CK_DX_GENERIC, CK_DW, CK_DW_SMALL, CK_HEAD_FWD,
  CK_CRITIC_HEAD, CK_CRITIC_HEAD_GENERIC, CK_ACTOR_DZ, CK_ACTOR_DZ_GENERIC, CK_ADAM, CK_ADAM_HER, CK_POLYAK, CK_CHECKSUM,
  CK_NOISE, CK_ENV_RESET, CK_ENV_STEP, CK_COUNTER_ADD, CK_FWD_PI, CK_DW_ADAM_HER, CK_ACT_STEP, CK_FWD_L01, CK_ROWS,
  CK_ACT_ROWS, CK_ROWS_T, CK_ROUTE, CK_ACT_RES, CK_ROWS_HER, CK_IPC, CK_COUNT
};
extern int g_curious_prof_on;

// ---------------------------------------------------------------- run-time options (curious_set_option)
#define DW_BAL_MIN_B 1280       // batches from this size on deal their small weight-gradient problems in halves (mlp_dw.h dw_role)
#define ROWS16_DEFAULT_MIN 1280      // batch rows from which the 16-row form of the row-local update is taken (option rows16)
struct CuriousOptions {
  int rows;            // 1: row-local routes (mlp_rows.h, mlp_rows_act.h); 0: tiled multi-launch routes      [CURIOUS_ROWS]
  int rows_xcd;        // 1: kinds of ddpg_rows_kernel placed by XCD; 0: plain block-id order                 [CURIOUS_ROWS_XCD]
  int xcd_map;         // 0 / 4 / 8: XCD-aware block placement of fwd_hot / dx_hot (tiled route)              [CURIOUS_XCD_MAP]
  int fault_inject;    // > 0: the target group of row group (fault_inject - 1) never publishes Q' (tests)
  int qt_spins;        // polls before a consumer of Q' gives up
  int resident;        // 1: multi-step rollouts keep the actor's hidden matrices in LDS (mlp_rows_res.h) when the grid fits
                       // the device; 0: policy_rows_kernel streams them every step                     [CURIOUS_RESIDENT]
  int res_spins;       // polls before a member of a resident-rollout group gives up on a peer
  int lab_res_stamps;  // LAB ONLY (tools/res_stamps.py): policy_resident_kernel adds per-phase cycle stamps of block 0 to 8
                       // 64-bit words behind its exchange buffer in the workspace
  int dw_xcd;          // 1: blocks of the weight-gradient / optimiser launch placed by XCD (mlp_lean_gemm.h DwMap)  [CURIOUS_DW_XCD]
  int rows_pre;        // 1: the row-local launch's role / input rows / first layer-0 matrix travel as leading kernel arguments
                       //    (mlp_rows.h RowsPre; 0 = fetched from the argument segment as before: A/B)  [CURIOUS_ROWS_PRE]
  int rows8;           // 1: the row-local update gives 8 batch rows to a workgroup for batches of >= 768 rows (virtual ranks);
                       //    0: always 4 (A/B)                                                       [CURIOUS_ROWS8]
  int rows16;          // > 0: from this many batch rows on the row-local update gives 16 rows to a workgroup, the waves
                       //    splitting the output columns on v_mfma_f32_16x16x4 (mlp_rows16.h); 0: never (A/B)   [CURIOUS_ROWS16]
  int dw64;            // > 0 (A/B; default 0 = never): from this many batch rows on the hidden matrices' weight gradients are 64 x 64
                       //    tiles staged through LDS, 8 workgroups per tile (mlp_dw.h dw_hot_tile64) -- built and measured in round
                       //    6, no faster than the 16 x 64 tiles (19 ranks: 53.9 against 51.0 us), DESIGN 4.7          [CURIOUS_DW64]
  int fwd16;           // 1: curious_policy_forward on >= 1 024 rows (a multiple of 16) takes 16 rows per workgroup (mlp_rows_act.h
                       //    policy_fwd16_kernel) -- another order of the sums over k than the 4-row form the fused acting kernels share:
                       //    off by default, DDPG.rollout_q_sum (the evaluator's Q pass) switches it on around its calls
  int gather_dw;       // = N: curious_ddpg_grads with `next` on >= N rows puts the gather into the weight-gradient launch instead of the
                       //    row-local one (mlp_host_update.h); default ROWS16_DEFAULT_MIN, 0 = never                      [CURIOUS_GATHER_DW]
  int dw_bal;          // = N: batches of >= N rows deal their small weight-gradient problems over the XCDs in halves, 3 segments each
                       //    (mlp_dw.h dw_role); default DW_BAL_MIN_B = 1 280, 0 = never                                  [CURIOUS_DW_BAL]
  int dw_split;        // 0: segments per tile of the weight-gradient launch's split reduction chosen by batch size (mlp_dw.h
                       //    DwSplit; batches of >= 1 024 rows); 10 S_hot + S_small: fixed (A/B)       [CURIOUS_DW_SPLIT]
  int lab_rows_stamps; // LAB ONLY (tools/rows_stamps.py): every row group of ddpg_rows_kernel writes its phase stamps into the workspace
  int lab_dw_stamps;   // LAB ONLY (tools/dw_stamps.py): dw_adam_her_kernel writes per-block cycle stamps into the workspace
  int lab_no_target;   // LAB ONLY (tools/update_lab.py): the target groups of ddpg_rows_kernel exit at once and Q' = 0 --
                       // wrong numbers, right timing of an update whose targets were computed elsewhere
};
CuriousOptions& curious_options();

void curious_prof_push(int kid, hipStream_t st, bool start);
// Brackets ONE kernel launch with a pair of events recorded on the launch stream (no-op unless profiling is on).
// Every launch is also counted per kernel id (curious_prof_launch_counts: the GPU test session asserts that no kernel of
// the library went unexercised).
extern int64_t g_curious_launches[];
struct ProfScope {
  int kid; hipStream_t st; bool on;
  ProfScope(int k, hipStream_t s) : kid(k), st(s), on(g_curious_prof_on != 0) {
    ++g_curious_launches[kid];
    if (on) curious_prof_push(kid, st, true);
  }
  ~ProfScope() { if (on) curious_prof_push(kid, st, false); }
};
