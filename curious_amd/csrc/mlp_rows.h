// Row-local DDPG pass: the whole forward / backward chain of one update in ONE launch (included by mlp.hip).
//
// Replaces the 8 dependent launches fwd_l01 -> fwd_hot -> fwd_pi -> fwd_hot -> dx_crit -> dx_hot -> dx_actor -> dx_hot
// (DESIGN.md section 4).  Those kernels split every 256 x 256 layer over 64 workgroups and pay a kernel boundary
// (1.5 us) plus a cold start (~3 us) per layer.  Here a workgroup owns FOUR batch rows and walks all layers of its
// networks by itself: activations never leave the CU (LDS), workgroups exchange nothing but one scalar per batch row (Q',
// below), and the only per-layer cost is streaming the layer's 256 KB of weights L2 -> CU (tools/rowchain_lab.hip: 2.3-2.9 us per layer).
// Three kinds of workgroup per row group (grid.x = 4 * B / 4, a quarter of them exit at once: see the block-id map in the
// kernel; actor-side groups -- the longest chain -- and target groups are dispatched first, then the main-critic groups,
// which consume what the target groups produce):
//   actor side:  main actor -> pi;  main critic(pi) -> Q_pi;  backward through critic(pi) into the action slot -> dz;
//                backward of the actor
//   target:      target actor -> target critic(pi') -> Q'                      (handed to the main-critic group)
//   main critic: main critic(u) -> Q;  waits for Q';  loss terms, dQ;  backward of critic(u)
// The one dependency between workgroups -- Q' of a batch row -- travels as a single 64-bit word per row (tag | value)
// written with one agent-scope atomic store and polled by the consumer, which clears it after reading: no fences, no
// separate flag, nothing to initialise (a word without the tag is "not yet").  Producers never wait and are dispatched
// before their consumers, so the kernel cannot deadlock whatever number of workgroups is resident.
// (ddpg.py:419-449 for the graph, actor_critic.py:51-98 / util.py:73-107 for the networks.)  Everything the weight-
// gradient launch needs -- layer inputs, masked output gradients, dQ, dz, per-row loss terms -- is written to the same
// workspace arrays the tiled kernels fill, so dw_adam_her_kernel / dw_all_kernel run unchanged afterwards.
//
// Matrix instruction: v_mfma_f32_4x4x1_16b_f32 = 16 independent blocks of (4 rows x 4 columns, K = 1), exact f32 FMA.
//   forward  (h . W):   lane l supplies A = h[l & 3][k] and B = W[k][4 l + e]; accumulator e holds columns {4 l + e}
//                       of the 4 rows; the 4 waves split k (64 each) and meet in LDS.  One 16-byte load of W[k][4l..]
//                       feeds 4 instructions; a wave instruction reads one contiguous 1 KB row.
//   backward (dY . W^T): the matrix instruction wants lane l to supply W[k_l][n] for ONE n per instruction, memory offers
//                       W[k][n .. n+3] contiguously (a direct 16-byte load per lane makes every quad of lanes touch 4
//                       different rows: 2.5x the forward layer's time; staging the rows through LDS: 1.2x).  So the
//                       library keeps TRANSPOSED copies of the main networks' hidden matrices in the workspace --
//                       written by the optimiser (epilogue of dw_adam_her_kernel; tile blocks of the stand-alone
//                       adam launch) next to the parameters it updates, rebuilt by rows_transpose_kernel otherwise -- and dY . W^T is the forward product on W^T
//                       (rows_big_bwdT): a backward layer costs what a forward layer costs.
#pragma once

#define ROWS_MAXL 4          // layers per network on this route (layer 0 + up to 3 hidden layers)
#define ROWS_R 4             // batch rows per workgroup ...
#define ROWS_R2 8            // ... or 8: batches of >= ROWS_R2_MIN rows (3 virtual ranks or more, section 4.7 of DESIGN.md: more row groups of 4 than the chip has workgroup slots).  Every CU then
                             // holds several row groups and the launch is bound by the L2 -> CU weight stream in aggregate, not by
                             // one chain's latency: with 8 rows a 16-byte load of W feeds 8 matrix instructions, the stream per
                             // row halves.  (At B = 256 -- one chain per CU -- the 8-row form is the slower one: round 2.)
#define ROWS_R2_MIN 768
#ifndef ROWS16_WAVES
#define ROWS16_WAVES 3       // workgroups per CU the 16-row kernels are compiled for (register budget 512 / 3 -> 168)
#endif
#define RLD 264              // LDS row stride of an activation row (8 mod 64: conflict-free b128 broadcast reads)
#define ROWS_MAXIN 128       // widest layer-0 input [o | td | action | g] the row-local kernels take: two passes of 64 (rows_l0_fwd)
#define XLD 132              // LDS row stride of that input row: 132 mod 64 = 4 puts the 4 rows a
                             // wave reads together on 4 different banks (96 made rows 0/2 and 1/3 collide: 14 % of the LDS cycles)
#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, 0)

struct RowsNet {             // one network: parameter base + float offsets of its pieces (NetOff)
  const float* th;
  int32_t W0, b0, Wg, Wout, bout;
  int32_t W[ROWS_MAXL], b[ROWS_MAXL];
};
struct RowsArgs {
  RowsNet tQ, tPi, mQ, mPi;
  const float* batch;
  int32_t ld, off_o, off_td, off_u, off_g, off_o2, off_g2, off_r;
  float* actc[ROWS_MAXL]; float* dactc[ROWS_MAXL];     // main critic(u): layer outputs, masked output gradients [B,H]
  float* acta[ROWS_MAXL]; float* dacta[ROWS_MAXL];     // main actor
  float* dQ; float* dz; float* rows; float* out_Qpi;
  unsigned long long* qt;         // [B] hand-off words target group -> main-critic group: (ROWS_QT_TAG << 32) | bits(Q')
  const float* wTq[ROWS_MAXL]; const float* wTpi[ROWS_MAXL];   // transposed hidden matrices of main critic / actor
  int64_t* step_ctr;
  int32_t B, nl, dimo, dimtd, dimg, xmap;
  int32_t Bl;                     // rows per (virtual) rank: the loss means divide by it (curious_net_cfg_t.loss_rows; B for one rank)
  float gamma, clip_lo, clip_hi, max_u, l2c;
  unsigned long long* stamps;     // diagnostics (tools/rows_lab.hip): [3][32] s_memtime stamps of row group 0, else NULL
  int32_t* fault;                 // fault word of the workspace: incremented by every consumer wave that gave up on Q';
                                  // the optimiser that follows leaves theta / m / v / the copies alone while it is set
  int32_t inject, spins;          // inject > 0 (tests): the target group of row group inject - 1 never publishes;
                                  // spins: polls before a consumer gives up
  int32_t lab_no_target;          // lab only: no target groups, Q' = 0 (timing of an update with precomputed targets)
  int32_t stamp_all;              // lab (option lab_rows_stamps): every row group stamps (ROWS_STAMP)
  int32_t n_her;                  // > 0: B / 4 spare workgroups of this launch run the HER gather of the NEXT update's
                                  // batch (her_body.h) -- on CUs the three kinds leave idle, hidden behind the chains.
                                  // The step counter is then NOT incremented here (the gather keys its Philox stream on
                                  // it: counter + 1) but by the weight-gradient launch that follows (LossFin.step_ctr).
  // input normalisation (actor_critic.py:76-83, --normalize_obs): mean / std of the observation and goal normalisers or
  // NULL; the normalised input rows [o | td | u / max_u | g] of the main critic(u) and main actor passes are kept in
  // xn_c / xn_a ([B][XLD]) for the layer-0 weight gradients
  const float *o_mean, *o_std, *g_mean, *g_std;
  float nclip;
  float *xn_c, *xn_a;
};
// (stamp_all, option lab_rows_stamps: EVERY row group stamps, 16 words per (row group, kind): [k] cycle counter of stamp k,
//  [10] / [11] the 100 MHz real-time counter at the first / the latest stamp -- tools/rows_stamps.py)
#define ROWS_STAMP(k)                                                                                   \
  do {                                                                                                  \
    if (a.stamps && x.tid == 0) {                                                                       \
      if (a.stamp_all) {                                                                                \
        unsigned long long* sp_ = a.stamps + ((size_t)rgrp * 3 + kind) * 16;                           \
        sp_[(k)] = __builtin_readcyclecounter();                                                        \
        if ((k) == 0) sp_[10] = __builtin_amdgcn_s_memrealtime();                                       \
        sp_[11] = __builtin_amdgcn_s_memrealtime();                                                     \
      } else if (rgrp == 0) a.stamps[kind * 32 + (k)] = __builtin_readcyclecounter();                  \
    }                                                                                                   \
  } while (0)
#define ROWS_QT_TAG 0x51C0FFEEull
// A consumer that has polled RowsArgs.spins times (default 2^22, ~ seconds) gives up: the loss turns NaN instead of a
// hang, and the fault word of the workspace is incremented -- dw_adam_her_kernel / adam_kernel / adam_her_kernel then skip
// the optimiser (theta, m, v and the transposed copies stay as they were) until the host has read and cleared the word
// (DDPG.check_faults raises).  HIP does not promise block-id dispatch order, which is what makes the producers start
// before their consumers: the guard turns a violated assumption into an error instead of silently poisoned parameters.

#ifdef ROWS_DEBUG           // tools/rows_lab.hip: fine-grained stamps inside the layer routines
#define ROWS_DBG(x) do { if ((x).dbg && (x).tid == 0) *(x).dbg++ = __builtin_readcyclecounter(); } while (0)
#else
#define ROWS_DBG(x) do { } while (0)
#endif
#ifdef ROWS_DEBUG2          // (tools/rows_lab.hip -DROWS_DEBUG2: a stamp behind every chunk of the forward hidden layers)
#define ROWS_DBG2(x) ROWS_DBG(x)
#else
#define ROWS_DBG2(x) do { } while (0)
#endif
struct RCtx {
  mutable unsigned long long* dbg;
  mutable float* hs; mutable float* part; mutable float* part2; float* xin; float* sm;
  mutable float* hn;              // 16 rows (mlp_rows16.h): the activation buffer the running layer writes; swaps with hs
  float* keep;                    // relu' masks of the kept layers: 4 rows per workgroup -- the kept activations in LDS;
  mutable uint64_t kb;            // 8 rows -- 8 bits to a layer (rows_keep); 16 rows -- 16 bits to a layer in kb | kb2
  mutable uint64_t kb2;
  int tid, wave, lane, r0;
};
// a result other workgroups read: the weight-gradient launch that follows
// (lab, -DROWS_NT_STORES: written through instead of left dirty in the L2 for the release at the end of the kernel --
//  tools/floor2_lab.hip prices that release at 0.15 us per MB; profiles/r05_floor2_lab.txt)
__device__ __forceinline__ void rows_gst(const RCtx& x, float* p, float v) {
#ifdef ROWS_NT_STORES
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

#include "mlp_rows_layers.h"

// ================================================================== the kernel
// grid (4 * B / R, 1, n_experts); B % (4 R) == 0.
static inline size_t rows_lds_floats(int R, int nl) {
  if (R == ROWS_R3) return (size_t)2 * R * RLD + R * XLD + 96;   // two activation buffers, no partial tiles: 42 KB
  return (size_t)R * RLD + 2 * 4 * R * 256 + R * XLD + 64 +  // (two buffers of partials: rows_fw_finish)
         (R == ROWS_R ? (size_t)2 * nl * 4 * 256 : 0);       // 4 rows: the kept activations (rows_keep); 64 KB | 77 KB
}

// HER: the launch carries the gather of the next batch (ddpg_rows_her_kernel: a kernel of its own, so that the plain
// form keeps its 632-byte kernarg -- HerArgs adds 1.4 KB)
// Returns the workgroup's role as an index: spare blocks [0, nrg), target groups nrg + row group, main-critic groups
// 2 nrg + row group, actor-side groups 3 nrg + row group.
template <bool EX, bool HER, int R>
__device__ __forceinline__ int ddpg_rows_body(const RowsArgs& a, const Ex& ex, const HerArgs* her,
                                              const uint64_t seed_stride, const RowsPre* pre = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float rows_lds[];
  RCtx x;
  x.dbg = nullptr;
  x.hs = rows_lds;
  x.part = x.hs + R * RLD;
  x.part2 = x.part + 4 * R * 256;
  x.xin = x.part2 + 4 * R * 256;
  x.hn = nullptr;
  if constexpr (R == ROWS_R3) {                              // two activation buffers instead of the partial tiles
    x.hn = x.hs + R * RLD;
    x.part = x.part2 = nullptr;
    x.xin = x.hn + R * RLD;
  }
  x.sm = x.xin + R * XLD;
  x.keep = x.sm + 64;                                       // (4 rows) [2 * nl][4 rows][256]: activations kept for relu'
  x.kb = x.kb2 = 0;
  x.tid = threadIdx.x; x.wave = x.tid >> 6; x.lane = x.tid & 63;
  // Kind of workgroup and row group from the block id.  Workgroups are dealt round-robin over the 8 XCDs in block-id
  // order (block b lands on XCD b % 8; speed only, nothing depends on it for correctness).  A layer's time is set by
  // how many workgroups stream weights out of one XCD's L2 at the same time, and
  // the actor-side chain is the longest: it gets XCDs 0-3 for itself, 16 workgroups per XCD at B = 256, while the target
  // and main-critic groups share XCDs 4-7, 32 per XCD -- they have the slack.  grid.x = 4 * nrg: slot = b / 8;
  //   XCD 0-3: slots [0, nrg/4) actor side, the rest exit;   XCD 4-7: slots [0, nrg/4) target, [nrg/4, nrg/2) main critic
  // (lower block ids are dispatched first: the longest chain and the producers start before the consumers).
  // Without the XCD map: grid (3 * nrg, 1, experts), kinds in dispatch order ACROSS the experts: the actor-side groups of
  // all experts, then all target groups, then all main-critic groups (batched experts fill the chip several times over; a
  // main-critic group that is dispatched while its target group still runs would hold a CU just to wait); kind 3 (grid.x =
  // 4 * nrg only with n_her > 0): the gather blocks.
  // The role and what a row group requests FIRST -- its input rows and the first 16 rows of its first layer-0 matrix --
  // come from the leading arguments when the host filled them (RowsPre: nothing is read from the argument segment up to
  // the `sched_barrier` below), else from the arguments proper; ONE site issues the loads either way.
  f32x4 wb[2][16];
  float xraw[ROWS_IN_IT(R)];
  int nrg, kind, rgrp, expert = 0, spare = -1;
  int64_t eo = 0;
  const bool pre_path = !EX && pre != nullptr && ((pre->k[4] >> 25) & 1u);
  // (the arguments proper are read inside `else` branches that end in an empty asm: a plain select between a leading
  //  argument and a fetched one would wait for the fetch on both paths)
  bool xmap = true;
  nrg = (int)(pre_path ? (pre->k[4] & 0xffffu) : 0u) / R;
  if (!pre_path) {
    int bv = a.B, xm = a.xmap;
    asm volatile("" : "+s"(bv), "+s"(xm));
    nrg = bv / R;
    xmap = xm != 0;
  }
  if (xmap) {                                                // single agent, grid.x = 4 * nrg
    const int per = nrg >> 2;
    const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    kind = (xcd < 4) ? 0 : (slot < per) ? 1 : 2;
    rgrp = (xcd & 3) * per + (slot < per ? slot : slot - per);
    if (xcd < 4 && slot >= per) { kind = 3; spare = xcd * per + (slot - per); }   // the spare quarter: the gather, or nothing
  } else {
    const int groups = nrg * (int)gridDim.z;
    const int lin = (int)blockIdx.z * (int)gridDim.x + (int)blockIdx.x;
    kind = lin / groups;
    const int rem = lin - kind * groups;
    expert = rem / nrg;
    rgrp = rem - expert * nrg;
  }
  if (kind == 3) {
    if (xmap) {
      if (HER) {                                             // (a gather block draws ROWS_R transitions: R / 4 of them here)
#pragma unroll 1
        for (int u = 0; u < R / 4; ++u) {
          if ((R / 4) * spare + u < a.n_her) her_sample_body(*her, (R / 4) * spare + u, rows_lds, 0, 0, 1);
          if (R > 4) __syncthreads();
        }
      }
      return spare;
    }
    int64_t ge;
    (void)ex_decode<EX>(ex, expert, ge);
    if (HER) {
#pragma unroll 1
      for (int u = 0; u < R / 4; ++u) {
        if ((R / 4) * rgrp + u < a.n_her) her_sample_body(*her, (R / 4) * rgrp + u, rows_lds, ge, (uint64_t)expert * seed_stride, 1);
        if (R > 4) __syncthreads();
      }
    }
    return rgrp;
  }
  x.r0 = rgrp * R;
  // what every kind derives from the arguments proper -- expanded INSIDE each kind's branch, behind the branch's first
  // loads (rows_first_loads): in front of the branches these lines would wait for the argument fetch before any load went out
#define ROWS_COMMON()                                                                                            \
  (void)ex_decode<EX>(ex, expert, eo);                      /* one problem per expert */                        \
  const int nl = a.nl, Sa = a.dimo + a.dimtd, Sc = Sa + 4, G = a.dimg;                                           \
  const float* batch = a.batch + eo; (void)batch;                                                                \
  /* wave i finishes the output layers of the workgroup's rows i, i + 4, ...: `row` in the loops over hh below */ \
  const float invB = 1.0f / (float)a.Bl; (void)invB;                                                             \
  float* sm_s = x.sm; (void)sm_s;                           /* [R] per-row scalar handed from wave i to the column threads */ \
  float* sm_v = x.sm + 16; (void)sm_v;                      /* [R][4] per-row 4-vectors (pi, dz) */             \
  unsigned long long* qt = reinterpret_cast<unsigned long long*>(reinterpret_cast<float*>(a.qt) + eo) + x.r0; (void)qt; \
  (void)Sc; (void)G; (void)nl

  if (kind == 1) {
    // ================================================= target group: pi' = target actor(o_2, g_2), Q' = target critic
    rows_first_loads<EX, R>(x, a, ex, pre, pre_path, 1, rgrp, expert, wb[0], xraw);
    ROWS_COMMON();
    if (a.lab_no_target) return nrg + rgrp;
    const float* tp = a.tPi.th + eo;
    const float* tq = a.tQ.th + eo;
    ROWS_STAMP(0);
    const float b0_tp = tp[a.tPi.b0 + x.tid];
    rows_inputs_commit<R>(x, a, false, xraw, nullptr, eo);
    __syncthreads();
    ROWS_STAMP(1);
    const HeadW4 wpi_t = rows_head4_w(tp + a.tPi.Wout, x.lane);
    const float bpi_t = tp[a.tPi.bout + (x.lane & 3)];
    const float b0_tq = tq[a.tQ.b0 + x.tid];
    rows_l0_fwd<R>(x, wb, tp + a.tPi.W0, Sa, tp + a.tPi.Wg, G, Sc, b0_tp, tp + a.tPi.b0, -1, nullptr,
                   rnext(RN_FWD, tp + a.tPi.W[1]), true);
    ROWS_STAMP(2);
    rows_hidden_fwd<R>(x, wb, a, a.tPi, tp, -1, 0, eo, rnext(RN_L0, tq + a.tQ.W0, Sc, tq + a.tQ.Wg, Sc + G));
    ROWS_STAMP(3);
    const f32x4 wq_t = ldv(tq + a.tQ.Wout + 4 * x.lane);
    const float bq_t = tq[a.tQ.bout];
    if constexpr (R == ROWS_R3) {                            // (the output layer on the matrix unit: mlp_rows16.h)
      float bf[16];
      r16_frag_cols4(bf, tp + a.tPi.Wout, x.wave, x.lane);
      const float z = r16_thin_sum(x, r16_thin(x, bf));
      if (x.lane < 16) {
        const float v = a.max_u * tanhf(z + bpi_t);                                        // actor_critic.py:89
        x.xin[(4 * x.wave + (x.lane >> 2)) * XLD + Sa + (x.lane & 3)] = fdiv(v, a.max_u);  // actor_critic.py:93
      }
    } else {
#pragma unroll
    for (int hh = 0; hh < R / 4; ++hh) {
      const int row = 4 * hh + x.wave;
      float z[4];
      rows_head4(x, wpi_t, z, row);
      if (x.lane < 4) {
        float v = 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d) v = (x.lane == d) ? z[d] : v;
        v = a.max_u * tanhf(v + bpi_t);                                                    // actor_critic.py:89
        x.xin[row * XLD + Sa + x.lane] = fdiv(v, a.max_u);                                 // actor_critic.py:93
      }
    }
    }
    __syncthreads();
    ROWS_STAMP(4);
    rows_l0_fwd<R>(x, wb, tq + a.tQ.W0, Sc, tq + a.tQ.Wg, G, Sc, b0_tq, tq + a.tQ.b0, -1, nullptr,
                   rnext(RN_FWD, tq + a.tQ.W[1]), true);
    rows_hidden_fwd<R>(x, wb, a, a.tQ, tq, -1, 0, eo, rnext(RN_NONE, nullptr));
    ROWS_STAMP(5);
    if constexpr (R == ROWS_R3) {
      float bf[16];
      r16_frag_rows(bf, tq + a.tQ.Wout, 1, x.wave, x.lane);
      const float Qt = r16_thin_sum(x, r16_thin(x, bf)) + bq_t;                            // ddpg.py:427-431
      const int row = 4 * x.wave + (x.lane >> 2);
      if (x.lane < 16 && (x.lane & 3) == 0 && !(a.inject > 0 && (x.r0 + row) / ROWS_R == a.inject - 1))
        __hip_atomic_store(qt + row, (ROWS_QT_TAG << 32) | (unsigned long long)__float_as_uint(Qt), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    } else {
#pragma unroll
    for (int hh = 0; hh < R / 4; ++hh) {
      const int row = 4 * hh + x.wave;
      const float Qt = rows_head1(x, wq_t, row) + bq_t;                                    // ddpg.py:427-431
      // (fault injection, tests: the target group of row group inject - 1 of the 4-row grid never publishes)
      if (x.lane == 0 && !(a.inject > 0 && (x.r0 + row) / ROWS_R == a.inject - 1))
        __hip_atomic_store(qt + row, (ROWS_QT_TAG << 32) | (unsigned long long)__float_as_uint(Qt), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    }
    ROWS_STAMP(6);
    return nrg + rgrp;
  }

  if (kind == 2) {
    // ================================================= main-critic group: critic(o, g, u), loss, backward
    rows_first_loads<EX, R>(x, a, ex, pre, pre_path, 2, rgrp, expert, wb[0], xraw);
    ROWS_COMMON();
    if (!HER && rgrp == 0 && x.tid == 0 && a.step_ctr) *ex_i64(a.step_ctr, eo) += 1;
    const float* mq = a.mQ.th + eo;
    ROWS_STAMP(0);
    const float b0_mq = mq[a.mQ.b0 + x.tid];
    rows_inputs_commit<R>(x, a, true, xraw, a.xn_c ? a.xn_c + eo : nullptr, eo);
    __syncthreads();
    ROWS_STAMP(1);
    // operands of the head / loss / first backward step, fetched ahead of the hidden layers
    const f32x4 wq_m = ldv(mq + a.mQ.Wout + 4 * x.lane);
    const float bq_m = mq[a.mQ.bout];
    const float wq_col = mq[a.mQ.Wout + x.tid];
    const f32x4 wq_c4 = ldv(mq + a.mQ.Wout + 64 * x.wave + 4 * (x.lane & 15));   // (16 rows: the thread's 4 columns)
    (void)wq_col; (void)wq_c4;
    float rew[R / 4];
#pragma unroll
    for (int hh = 0; hh < R / 4; ++hh) rew[hh] = batch[(int64_t)(x.r0 + 4 * hh + x.wave) * a.ld + a.off_r];
    // (16 rows: lane L < 16 of wave w finishes row 4 w + (L >> 2) -- mlp_rows16.h r16_thin_sum)
    const float rew_l = batch[(int64_t)(x.r0 + 4 * x.wave + ((x.lane & 15) >> 2)) * a.ld + a.off_r];
    (void)rew; (void)rew_l;
    // relu' masks kept for the backward pass (slots 0 .. nl - 1), layer outputs stored for the weight gradients
    rows_l0_fwd<R>(x, wb, mq + a.mQ.W0, Sc, mq + a.mQ.Wg, G, Sc, b0_mq, mq + a.mQ.b0, 0, a.actc[0] + eo,
                   rnext(RN_FWD, mq + a.mQ.W[1]), true);
    ROWS_STAMP(2);
    rows_hidden_fwd<R>(x, wb, a, a.mQ, mq, 0, 1, eo, rows_bwd_first(a, false, eo));
    ROWS_STAMP(3);
    if constexpr (R == ROWS_R3) {
      float bf[16];
      r16_frag_rows(bf, mq + a.mQ.Wout, 1, x.wave, x.lane);
      const float Q = r16_thin_sum(x, r16_thin(x, bf)) + bq_m;
      const int row = 4 * x.wave + ((x.lane & 15) >> 2), m = x.r0 + row;
      if (x.lane < 16 && (x.lane & 3) == 0) {               // one lane per row: Q' of its row from the target group
        unsigned long long word = 0;
        int spins = 0;
        const int max_spins = a.spins > 0 ? a.spins : (1 << 22);
        for (;;) {
          if (a.lab_no_target) { word = ROWS_QT_TAG << 32; break; }
          word = __hip_atomic_load(qt + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((word >> 32) == ROWS_QT_TAG || ++spins > max_spins) break;
          __builtin_amdgcn_s_sleep(1);
        }
        const bool got = (word >> 32) == ROWS_QT_TAG;
        const float Qt = got ? __uint_as_float((unsigned)(word & 0xffffffffull)) : NAN;
        __hip_atomic_store(qt + row, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // consumed
        if (!got && a.fault) atomicAdd(reinterpret_cast<int32_t*>(reinterpret_cast<float*>(a.fault) + eo), 1);
        const float target = got ? fclip(rew_l + a.gamma * Qt, a.clip_lo, a.clip_hi) : NAN;   // ddpg.py:437-438
        const float diff = target - Q;
        const float dq = -2.0f * invB * diff;
        rows_gst(x, a.rows + eo + m, diff * diff);           // ddpg.py:439
        rows_gst(x, a.dQ + eo + m, dq);
        sm_s[row] = dq;
      }
      ROWS_STAMP(4);
    } else {
#pragma unroll
    for (int hh = 0; hh < R / 4; ++hh) {
      const int row = 4 * hh + x.wave, m = x.r0 + row;
      const float Q = rows_head1(x, wq_m, row) + bq_m;
      // Q' of this wave's row from the target group (every lane polls the same word: one request per poll)
      unsigned long long word = 0;
      int spins = 0;
      const int max_spins = a.spins > 0 ? a.spins : (1 << 22);
      for (;;) {
        if (a.lab_no_target) { word = ROWS_QT_TAG << 32; break; }
        word = __hip_atomic_load(qt + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((word >> 32) == ROWS_QT_TAG || ++spins > max_spins) break;
        __builtin_amdgcn_s_sleep(1);
      }
      ROWS_STAMP(4);
      const bool got = (word >> 32) == ROWS_QT_TAG;
      const float Qt = got ? __uint_as_float((unsigned)(word & 0xffffffffull)) : NAN;
      if (x.lane == 0) {
        __hip_atomic_store(qt + row, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // consumed
        if (!got && a.fault) atomicAdd(reinterpret_cast<int32_t*>(reinterpret_cast<float*>(a.fault) + eo), 1);
      }
      // (fminf / fmaxf drop a NaN operand, so the clip alone would turn a missing Q' into the target clip_lo: the NaN is
      //  re-injected behind it -- a faulted update reports a NaN loss as well as the fault word)
      const float target = got ? fclip(rew[hh] + a.gamma * Qt, a.clip_lo, a.clip_hi) : NAN;   // ddpg.py:437-438
      const float diff = target - Q;
      const float dq = -2.0f * invB * diff;                  // d mean((target - Q)^2) / dQ, target is a constant
      if (x.lane == 0) {
        rows_gst(x, a.rows + eo + m, diff * diff);           // ddpg.py:439
        rows_gst(x, a.dQ + eo + m, dq);
        sm_s[row] = dq;
      }
    }
    }
    __syncthreads();
    // ---- backward through the output layer: dY[i][c] = dQ[i] Wout[c] relu'(h[i][c])
    if constexpr (R == ROWS_R3) {
      const f32x4 w = wq_c4;
      r16_seed(x, nl - 1, a.dactc[nl - 1] + eo, [&](int row, int e) { return sm_s[row] * w[e]; });
    } else {
      const int L = nl - 1;
      const float w = wq_col;
      const uint32_t hk = rows_kept<R>(x, L);
      float* g = a.dactc[L] + eo;
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const float v = ((hk >> i) & 1u) ? sm_s[i] * w : 0.f;
        x.hs[i * RLD + x.tid] = v;
        rows_gst(x, g + (int64_t)(x.r0 + i) * 256 + x.tid, v);
      }
    }
    __syncthreads();
    ROWS_STAMP(5);
    rows_hidden_bwd<R>(x, wb, a, 0, 1, eo, rnext(RN_NONE, nullptr));
    ROWS_STAMP(6);
    return 2 * nrg + rgrp;
  }

  // =================================================== actor side
  rows_first_loads<EX, R>(x, a, ex, pre, pre_path, 0, rgrp, expert, wb[0], xraw);
  ROWS_COMMON();
#ifdef ROWS_DEBUG
  if (a.stamps && rgrp == 0) x.dbg = a.stamps + 96;
#endif
  const float* mp = a.mPi.th + eo;
  const float* mq = a.mQ.th + eo;
  const int keepA = 0;                                      // relu' masks: slots of the actor's layers
  const int keepD = nl;                                     // ... of critic(pi)'s
  ROWS_STAMP(0);
  const float b0_mp = mp[a.mPi.b0 + x.tid];
  rows_inputs_commit<R>(x, a, false, xraw, a.xn_a ? a.xn_a + eo : nullptr, eo);
  __syncthreads();
  ROWS_STAMP(1);
  const HeadW4 wpi = rows_head4_w(mp + a.mPi.Wout, x.lane);
  const f32x4 bpi = ldv(mp + a.mPi.bout);
  const float b0_mq = mq[a.mQ.b0 + x.tid];
  rows_l0_fwd<R>(x, wb, mp + a.mPi.W0, Sa, mp + a.mPi.Wg, G, Sc, b0_mp, mp + a.mPi.b0, keepA, a.acta[0] + eo,
                 rnext(RN_FWD, mp + a.mPi.W[1]), true);
  ROWS_STAMP(2);
  rows_hidden_fwd<R>(x, wb, a, a.mPi, mp, keepA, 2, eo, rnext(RN_L0, mq + a.mQ.W0, Sc, mq + a.mQ.Wg, Sc + G));
  ROWS_STAMP(3);
  float pi[R / 4][4];
  float pi_l = 0.f;                                         // (16 rows: pi[row][d] of lane (row, d), mlp_rows16.h r16_thin_sum)
  if constexpr (R == ROWS_R3) {
    float bf[16];
    r16_frag_cols4(bf, mp + a.mPi.Wout, x.wave, x.lane);
    const float z = r16_thin_sum(x, r16_thin(x, bf));
    const int row = 4 * x.wave + ((x.lane & 15) >> 2), d = x.lane & 3, m = x.r0 + row;
    float bd = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) bd = (d == e) ? bpi[e] : bd;
    pi_l = a.max_u * tanhf(z + bd);                                                        // actor_critic.py:89
    float t[4];
    r16_quad(pi_l / a.max_u, t);
    float l2 = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) l2 += t[e] * t[e];                                         // ddpg.py:441
    if (x.lane < 16) {
      x.xin[row * XLD + Sa + d] = fdiv(pi_l, a.max_u);                                     // actor_critic.py:93
      if (d == 0) rows_gst(x, a.rows + eo + 2 * a.B + m, l2);
    }
  } else {
#pragma unroll
  for (int hh = 0; hh < R / 4; ++hh) {
    const int row = 4 * hh + x.wave, m = x.r0 + row;
    float z[4];
    rows_head4(x, wpi, z, row);
    float l2 = 0.f;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      pi[hh][d] = a.max_u * tanhf(z[d] + bpi[d]);                                          // actor_critic.py:89
      const float t = pi[hh][d] / a.max_u;
      l2 += t * t;                                                                         // ddpg.py:441
    }
    if (x.lane < 4) {
      float v = 0.f;
#pragma unroll
      for (int d = 0; d < 4; ++d) v = (x.lane == d) ? pi[hh][d] : v;
      x.xin[row * XLD + Sa + x.lane] = fdiv(v, a.max_u);                                   // actor_critic.py:93
    }
    if (x.lane == 0) rows_gst(x, a.rows + eo + 2 * a.B + m, l2);
  }
  }
  (void)pi; (void)pi_l;
  __syncthreads();
  ROWS_STAMP(4);
  // operands of the critic's head, of the first backward step, of the action-slot product and of the actor's first
  // backward step, fetched ahead of the layers in between
  const f32x4 wq_m = ldv(mq + a.mQ.Wout + 4 * x.lane);
  const float bq_m = mq[a.mQ.bout];
  const float wq_col = mq[a.mQ.Wout + x.tid];
  const f32x4 wq_c4 = ldv(mq + a.mQ.Wout + 64 * x.wave + 4 * (x.lane & 15));     // (16 rows: the thread's 4 columns)
  (void)wq_col; (void)wq_c4;
  // ---- main critic on (o, g, pi) -> Q_pi
  rows_l0_fwd<R>(x, wb, mq + a.mQ.W0, Sc, mq + a.mQ.Wg, G, Sc, b0_mq, mq + a.mQ.b0, keepD, nullptr,
                 rnext(RN_FWD, mq + a.mQ.W[1]), true);
  rows_hidden_fwd<R>(x, wb, a, a.mQ, mq, keepD, 0, eo, rows_bwd_first(a, false, eo));
  ROWS_STAMP(5);
  if constexpr (R == ROWS_R3) {
    float bf[16];
    r16_frag_rows(bf, mq + a.mQ.Wout, 1, x.wave, x.lane);
    const float Qpi = r16_thin_sum(x, r16_thin(x, bf)) + bq_m;
    const int m = x.r0 + 4 * x.wave + ((x.lane & 15) >> 2);
    if (x.lane < 16 && (x.lane & 3) == 0) {
      rows_gst(x, a.rows + eo + a.B + m, Qpi);               // ddpg.py:440
      a.out_Qpi[eo + m] = Qpi;
    }
  } else {
#pragma unroll
  for (int hh = 0; hh < R / 4; ++hh) {
    const int row = 4 * hh + x.wave, m = x.r0 + row;
    const float Qpi = rows_head1(x, wq_m, row) + bq_m;
    if (x.lane == 0) {
      rows_gst(x, a.rows + eo + a.B + m, Qpi);               // ddpg.py:440
      a.out_Qpi[eo + m] = Qpi;
    }
  }
  }
  __syncthreads();
  // ---- backward of -mean(Q_pi) through the critic into the action slot
  if constexpr (R == ROWS_R3) {
    const f32x4 w = wq_c4;
    r16_seed(x, keepD + nl - 1, nullptr, [&](int, int e) { return w[e] * (-invB); });
  } else {
    const float w = wq_col * (-invB);
    const uint32_t hk = rows_kept<R>(x, keepD + nl - 1);
#pragma unroll
    for (int i = 0; i < R; ++i) x.hs[i * RLD + x.tid] = ((hk >> i) & 1u) ? w : 0.f;
  }
  __syncthreads();
  ROWS_STAMP(6);
  f32x4 wu[4];
  {
    const float* Wu = mq + a.mQ.W0 + (int64_t)Sa * 256;
#pragma unroll
    for (int d = 0; d < 4; ++d) wu[d] = ldv(Wu + (int64_t)d * 256 + 4 * x.lane);
  }
  const f32x4 wpi_row = ldv(mp + a.mPi.Wout + 4 * x.tid);
  f32x4 wpi_c4[4];                                          // (16 rows: Wout rows of the thread's 4 columns)
  if constexpr (R == ROWS_R3) {
#pragma unroll
    for (int e = 0; e < 4; ++e) wpi_c4[e] = ldv(mp + a.mPi.Wout + 4 * (64 * x.wave + 4 * (x.lane & 15) + e));
  }
  (void)wpi_row; (void)wpi_c4;
  rows_hidden_bwd<R>(x, wb, a, keepD, 0, eo, rows_bwd_first(a, true, eo));
  ROWS_STAMP(7);
  // (every operand that comes from the critic's parameters is in registers or consumed by now: wu, wq_col above)
  if constexpr (R == ROWS_R3) {
    float bf[16];
    r16_frag_rows(bf, mq + a.mQ.W0 + (int64_t)Sa * 256, 4, x.wave, x.lane);
    const float v = r16_thin_sum(x, r16_thin(x, bf));
    const int row = 4 * x.wave + ((x.lane & 15) >> 2), d = x.lane & 3, m = x.r0 + row;
    const float th = pi_l / a.max_u;
    const float dpi = v / a.max_u + a.l2c * pi_l;
    const float dzl = dpi * a.max_u * (1.0f - th * th);
    if (x.lane < 16) {
      a.dz[eo + (int64_t)m * 4 + d] = dzl;
      sm_v[4 * row + d] = dzl;
    }
  } else {
#pragma unroll
  for (int hh = 0; hh < R / 4; ++hh) {
    // d / d(action slot): dd0 . Wu^T (Wu = the action rows of the critic's layer-0 kernel), then through
    // pi = max_u tanh(z) and the l2 term (ddpg.py:440-441)
    const int row = 4 * hh + x.wave, m = x.r0 + row;
    const f32x4 g4 = *reinterpret_cast<const f32x4*>(x.hs + row * RLD + 4 * x.lane);
    float dz[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const f32x4 w = wu[d];
      const float v = wave_sum(g4[0] * w[0] + g4[1] * w[1] + g4[2] * w[2] + g4[3] * w[3]);
      const float th = pi[hh][d] / a.max_u;
      const float dpi = v / a.max_u + a.l2c * pi[hh][d];
      dz[d] = dpi * a.max_u * (1.0f - th * th);
    }
    if (x.lane == 0) {
      const f32x4 o = {dz[0], dz[1], dz[2], dz[3]};
      *reinterpret_cast<f32x4*>(a.dz + eo + (int64_t)m * 4) = o;
      *reinterpret_cast<f32x4*>(sm_v + 4 * row) = o;
    }
  }
  }
  __syncthreads();
  // ---- backward through the actor's output layer: dY[i][c] = (sum_d dz[i][d] Wout[c][d]) relu'(a[i][c])
  if constexpr (R == ROWS_R3) {
    r16_seed(x, keepA + nl - 1, a.dacta[nl - 1] + eo, [&](int row, int e) {
      const f32x4 dzr = *reinterpret_cast<const f32x4*>(sm_v + 4 * row);
      const f32x4 w = wpi_c4[e];
      return dzr[0] * w[0] + dzr[1] * w[1] + dzr[2] * w[2] + dzr[3] * w[3];
    });
  } else {
    const int L = nl - 1;
    const f32x4 w = wpi_row;
    const uint32_t hk = rows_kept<R>(x, keepA + L);
    float* g = a.dacta[L] + eo;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const f32x4 dzr = *reinterpret_cast<const f32x4*>(sm_v + 4 * i);
      const float sv = dzr[0] * w[0] + dzr[1] * w[1] + dzr[2] * w[2] + dzr[3] * w[3];
      const float v = ((hk >> i) & 1u) ? sv : 0.f;
      x.hs[i * RLD + x.tid] = v;
      rows_gst(x, g + (int64_t)(x.r0 + i) * 256 + x.tid, v);
    }
  }
  __syncthreads();
  ROWS_STAMP(8);
  rows_hidden_bwd<R>(x, wb, a, keepA, 2, eo, rnext(RN_NONE, nullptr));
  ROWS_STAMP(9);
  return 3 * nrg + rgrp;
}

// waves_per_eu(2, 2): left alone the compiler spends 255 + 32 registers on this kernel -- one workgroup per CU.  Held to
// 256 it needs no spill, and two workgroups share a CU: nothing changes for a single agent (B / 4 * 4 workgroups <= CUs,
// one each), but batched experts (3 or 4 times as many workgroups as CUs) gain 17 % per cycle -- while one workgroup sits
// in an epilogue, a head or a hand-off, the other keeps the CU's fill path busy.
// (Not a way to make ONE row group faster: tools/rows_lab.hip B = 512 shows two co-resident groups streaming the same
//  matrices in the time of one, but they share the fetched lines through the L1; a variant of this kernel with 8 waves
//  per row group, 128 KB of each layer per wave quartet, ran every layer 1.3-1.5 x SLOWER -- the CU's fill path, not the
//  number of loads in flight, is what bounds a layer.)
#define RKA_T(o) "s_load_dword %0, %1, " #o "\n"
// Every 64-byte line of RowsArgs requested at once, at the very top of the kernel.  The body fetches its arguments in ~7
// dependent rounds (kind and row group, dimensions, the networks' offsets, ...) before a row group's first operand load,
// and at the start of a launch every one of them misses the scalar cache AND the L2 (~2 k cycles each under the load of
// all workgroups starting together).  Behind this batch the later rounds hit the scalar cache: ddpg_rows_kernel
// 32.05 -> 31.1 us, 4.67 -> 4.565 ms per cycle (A/B on one box, profiles/r04_ab_kernarg.txt).  An asm block because plain
// loads from the constant address space are sunk to their use by the compiler; no wait inside -- the compiler's own wait
// for the first argument it needs follows right behind the scheduling fence and covers these loads (SMEM waits are
// all-or-nothing) -- and the dummy target register stays reserved to the kernel's end.
__device__ __forceinline__ uint32_t rows_kernarg_touch() {
  static_assert(14 * 4 + sizeof(RowsArgs) > 64 * 11 && 14 * 4 + sizeof(RowsArgs) <= 64 * 12, "rows_kernarg_touch: line list");
  const uint64_t ka = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
  uint32_t d;
  asm volatile(RKA_T(0) RKA_T(64) RKA_T(128) RKA_T(192) RKA_T(256) RKA_T(320) RKA_T(384) RKA_T(448) RKA_T(512) RKA_T(576)
               RKA_T(640) RKA_T(704) : "=&s"(d) : "s"(ka) : "memory");
  __builtin_amdgcn_sched_barrier(0);
  return d;
}
template <bool EX, int R = ROWS_R>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void ddpg_rows_kernel(ROWS_PRE_PARAMS, RowsArgs a, Ex ex) {
#ifdef ROWS_DEBUG
  const unsigned long long t_entry = __builtin_readcyclecounter();   // (before the first argument is read)
#endif
  ROWS_PRE_MAKE(pre);
  const uint32_t touched = rows_kernarg_touch();
  ddpg_rows_body<EX, false, R>(a, ex, nullptr, 0, &pre);
  asm volatile("" :: "s"(touched));
#ifdef ROWS_DEBUG
  if (a.stamps && blockIdx.x == 0 && blockIdx.z == 0 && threadIdx.x == 0) a.stamps[31] = t_entry;   // (actor-side group 0)
#endif
}
template <bool EX, int R = ROWS_R>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void ddpg_rows_her_kernel(ROWS_PRE_PARAMS, RowsArgs a, Ex ex, HerArgs her, uint64_t seed_stride) {
  ROWS_PRE_MAKE(pre);
  const uint32_t touched = rows_kernarg_touch();
  ddpg_rows_body<EX, true, R>(a, ex, &her, seed_stride, &pre);
  asm volatile("" :: "s"(touched));
}

// 16 rows per workgroup (mlp_rows16.h): 42 KB of LDS and <= 168 registers -- THREE workgroups share a CU, so that two
// keep the matrix unit busy while the third sits in an epilogue, a head or a hand-off
template <bool EX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ROWS16_WAVES, ROWS16_WAVES)))
void ddpg_rows16_kernel(ROWS_PRE_PARAMS, RowsArgs a, Ex ex) {
  ROWS_PRE_MAKE(pre);
  const uint32_t touched = rows_kernarg_touch();
  ddpg_rows_body<EX, false, ROWS_R3>(a, ex, nullptr, 0, &pre);
  asm volatile("" :: "s"(touched));
}
template <bool EX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ROWS16_WAVES, ROWS16_WAVES)))
void ddpg_rows16_her_kernel(ROWS_PRE_PARAMS, RowsArgs a, Ex ex, HerArgs her, uint64_t seed_stride) {
  ROWS_PRE_MAKE(pre);
  const uint32_t touched = rows_kernarg_touch();
  ddpg_rows_body<EX, true, ROWS_R3>(a, ex, &her, seed_stride, &pre);
  asm volatile("" :: "s"(touched));
}

// ---- (re)build the transposed copies from the parameters: dst[j][n][k] = src[j][k][n], 256 x 256 each.
// grid (16 tiles of 64 x 64, matrices, n_experts)
struct RowsTransposeArgs { const float* src[2 * ROWS_MAXL]; float* dst[2 * ROWS_MAXL]; };
template <bool EX>
__global__ __launch_bounds__(256) void rows_transpose_kernel(RowsTransposeArgs a, Ex ex) {
  __shared__ float tile[64][65];
  int64_t eo;
  (void)ex_decode<EX>(ex, blockIdx.z, eo);
  const float* src = a.src[blockIdx.y] + eo;
  float* dst = a.dst[blockIdx.y] + eo;
  const int k0 = (blockIdx.x >> 2) * 64, n0 = (blockIdx.x & 3) * 64;
  const int c = threadIdx.x & 63, r4 = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) tile[4 * i + r4][c] = src[(int64_t)(k0 + 4 * i + r4) * 256 + n0 + c];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) dst[(int64_t)(n0 + 4 * i + r4) * 256 + k0 + c] = tile[c][4 * i + r4];
}
