// Layer routines of the row-local kernels for SIXTEEN batch rows per workgroup (included by mlp_rows.h; round 6).
//
// Batches of several ranks (DDPG virtual_ranks) fill every CU with several row groups; what bounds the launch
// then is the CUs' texture paths (62 B / clk each) against their matrix units: a row group of 8 rows pulls each layer's
// 256 KB for 4 096 cycles of v_mfma_f32_4x4x1 -- texture time = matrix time, and the two overlap badly (8.2 k + 8.2 k
// cycles per pair of co-resident groups take 12.8 k: profiles/r05_bench_virtual_ranks_19_kernel_stats.csv, 0.45 of the
// f32 MFMA peak).  Here a workgroup owns 16 rows on v_mfma_f32_16x16x4 and its four waves split the OUTPUT COLUMNS:
//   A = activations [16 rows x 4 k] out of LDS, B = W[4 k x 16 columns] straight from memory into registers;
//   wave w owns columns [64 w, 64 w + 64) for the WHOLE of k: no partial tiles, no reduction through LDS, 32 weight
//   registers in flight instead of 128; the stream per row halves again: 256 KB per 8 192 matrix cycles -- the texture
//   path is busy half of the time and the matrix unit decides.
// Lane (q, j) = (lane >> 4, lane & 15):
//   A operand: row j, the k of sub-block q;   B operand: the k of sub-block q, column 4 j + e of the wave's 64 for
//   accumulator e (one 16-byte load W[k][64 w + 4 j ..] feeds four instructions; a wave instruction reads 4 rows of 256 B);
//   accumulator e, element i = out[row 4 q + i][column 64 w + 4 j + e].
// The k of a HIDDEN layer are permuted so that a lane's A operands of four steps are one ds_read_b128: step t (0 .. 63)
// gives sub-block q the index k = 16 (t >> 2) + 4 q + (t & 3); weights and activations agree on it, the sum over k is a
// sum in another order (parity: the oracle at 1e-5, not the bits of the 4- / 8-row forms).  Layer 0 walks the virtual
// concatenation [o | td (| u) | g] in natural order, k = 4 t + q, one ds_read_b32 per step.
// Pipeline: the weights of 8 steps (one "phase") are in flight while the 32 matrix instructions of the previous 8 run,
// ONE load in front of every group of four instructions (a wave that issues loads in a row is held at each until the path
// has taken it: tools/rowchain2_lab.hip); two register sets of 8 x 4 alternate; the successor's first phase is requested
// with the layer's last one.  Activations alternate between two LDS buffers (x.hs / x.hn): one barrier per layer.
// relu' masks: 16 bits per kept layer and thread (its 4 x 4 outputs), in two 64-bit words.
#pragma once

#define ROWS_R3 16           // rows per workgroup of this form
#define ROWS_R3_MIN ROWS16_DEFAULT_MIN   // ... taken from this many batch rows on (common.h; option rows16)

__device__ __forceinline__ f32x4 lds4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
// a row of activations / gradients for the weight-gradient launch (60 MB per launch at 19 ranks)
__device__ __forceinline__ void r16_gst4(float* p, const f32x4 v) {
#ifdef ROWS16_NT_STORES       // (lab: streamed out instead of left dirty in the L2 until the release at the end of the kernel --
                              //  19 ranks: this launch 109.2 -> 108.5 us, the weight-gradient launch that reads them 42.5 -> 45.8)
  __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
#else
  *reinterpret_cast<f32x4*>(p) = v;
#endif
}

// lane part of every weight address of a hidden layer: rows 4 q + .., columns 64 wave + 4 j
__device__ __forceinline__ const float* r16_wl(const float* W, int wave, int lane) {
  return W + (int64_t)(4 * (lane >> 4)) * 256 + 64 * wave + 4 * (lane & 15);
}
// float offset of step i (0 .. 7) of a phase behind the phase's base (wl + 1024 t0 for the phase of steps t0 .. t0 + 7)
#define R16_OFF(i) ((((i) >> 2) * 16 + ((i) & 3)) * 256)

// Weight registers: two sets of 8 steps (wb[0][0 .. 7], wb[1][0 .. 7]); one is multiplied while the other is filled, ONE load
// in front of every group of four matrix instructions; 8-9 loads (9 KB per wave) are in flight.  (Measured and
// dropped: a ring of 16 steps, slot t & 15 refilled with step t + 16 behind its instructions -- 16 KB per wave in flight.
// Nothing gained where the weights come from the L2 (a row group alone on its CU: 10.8 -> 11.2 k cycles per hidden layer)
// and the FIRST network of every row group, which starts on cold caches right after the optimiser rewrote the parameters,
// took twice as long (2 x 11.8 -> 2 x 24 k cycles: 512 cache lines in flight per CU, the waves are held at the issue of
// their loads and their matrix instructions wait behind them); 19 ranks 109 -> 123 us, 8 ranks 53 -> 70 us.)
__device__ __forceinline__ const float* r16_l0_addr(const float* W0, int S, int dg, int nk, int kv, int col) {
  const int kc = (kv < nk) ? kv : 0;
  return W0 + (kc * 256 + ((kc < S) ? 0 : dg) + col);
}
// steps s0 .. s0 + 7 of a layer 0 (natural order: step t = rows 4 t + q of the virtual concatenation [W0 | Wg])
__device__ __forceinline__ void r16_l0_load8(f32x4 (&b)[16], const float* W0, int S, const float* Wg, int nk, int wave,
                                             int lane, int s0) {
  const int dg = (int)(Wg - W0) - S * 256, col = 64 * wave + 4 * (lane & 15), q = lane >> 4;
#pragma unroll
  for (int i = 0; i < 8; ++i) b[i] = ldv(r16_l0_addr(W0, S, dg, nk, 4 * (s0 + i) + q, col));
}
__device__ __forceinline__ void r16_fw_load8(f32x4 (&b)[16], const float* W, int wave, int lane) {
  const float* wl = r16_wl(W, wave, lane);
#pragma unroll
  for (int i = 0; i < 8; ++i) b[i] = ldv(wl + R16_OFF(i));
}
__device__ __forceinline__ void r16_prefetch(f32x4 (&b)[16], const RNext& n, int wave, int lane) {
  if (n.kind == RN_FWD) r16_fw_load8(b, n.W + n.off, wave, lane);
  else if (n.kind == RN_L0) r16_l0_load8(b, n.W, n.S, n.Wg, n.nk, wave, lane, 0);
}

// one phase of a hidden layer: 8 steps out of `use` with the A operands a0 (steps 0 .. 3) / a1 (4 .. 7); FILL: `fill` <-
// fa(i), one load in front of every four instructions; NEXTA: the A operands of the NEXT phase (n0, n1 <- an, an + 16)
// are read behind the first group of instructions (in front of it the wait for this phase's operands would wait for them
// as well: LDS operations return in order)
template <bool FILL, bool NEXTA, class Fill>
__device__ __forceinline__ void r16_phase(const f32x4 (&use)[16], f32x4 (&fill)[16], Fill fa, const f32x4 a0,
                                          const f32x4 a1, f32x4 (&acc)[4], const float* an, f32x4& n0, f32x4& n1) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (FILL) fill[i] = ldv(fa(i));
    if (NEXTA && i == 1) { n0 = lds4(an); n1 = lds4(an + 16); }
    __builtin_amdgcn_sched_barrier(0);
    const float av = (i < 4) ? a0[i & 3] : a1[i & 3];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = MFMA(av, use[i][e], acc[e]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---- the k loop of a 256 x 256 hidden layer (forward, or backward on the transposed copy): wb[0] holds steps 0 .. 7 on
// entry (the predecessor's prefetch) and the successor's first phase on exit
__device__ __forceinline__ void r16_big_steps(const RCtx& x, f32x4 (&wb)[2][16], const float* W, const RNext& next,
                                              f32x4 (&acc)[4]) {
  const int q = x.lane >> 4, j = x.lane & 15;
  const float* wl = r16_wl(W, x.wave, x.lane);
  const float* ar = x.hs + j * RLD + 4 * q;                  // chunk c of 16 k: ar + 16 c
  f32x4 a0 = lds4(ar), a1 = lds4(ar + 16), b0, b1;
#pragma unroll 1
  for (int it = 0; it < 3; ++it) {
    const float* w1 = wl + (int64_t)(16 * it + 8) * 1024;    // steps t0 ..: (t0 >> 2) * 16 * 256 = 1024 t0 floats
    const float* w2 = wl + (int64_t)(16 * it + 16) * 1024;
    r16_phase<true, true>(wb[0], wb[1], [&](int i) { return w1 + R16_OFF(i); }, a0, a1, acc, ar + 64 * it + 32, b0, b1);
    r16_phase<true, true>(wb[1], wb[0], [&](int i) { return w2 + R16_OFF(i); }, b0, b1, acc, ar + 64 * it + 64, a0, a1);
  }
  const float* w1 = wl + (int64_t)56 * 1024;
  r16_phase<true, true>(wb[0], wb[1], [&](int i) { return w1 + R16_OFF(i); }, a0, a1, acc, ar + 224, b0, b1);
  if (next.kind == RN_FWD) {
    const float* wn = r16_wl(next.W + next.off, x.wave, x.lane);
    r16_phase<true, false>(wb[1], wb[0], [&](int i) { return wn + R16_OFF(i); }, b0, b1, acc, nullptr, a0, a1);
  } else if (next.kind == RN_L0) {
    const int dg = (int)(next.Wg - next.W) - next.S * 256, col = 64 * x.wave + 4 * j;
    r16_phase<true, false>(wb[1], wb[0], [&](int i) { return r16_l0_addr(next.W, next.S, dg, next.nk, 4 * i + q, col); },
                           b0, b1, acc, nullptr, a0, a1);
  } else {
    r16_phase<false, false>(wb[1], wb[0], [&](int) { return (const float*)nullptr; }, b0, b1, acc, nullptr, a0, a1);
  }
}

// ---- relu' masks: thread (wave, q, j) finishes out[4 q + r][64 wave + 4 j + e] of every layer -- bit 4 r + e of the slot
__device__ __forceinline__ void r16_keep(const RCtx& x, const int slot, const uint32_t bits) {
  if (slot < 4) x.kb |= (uint64_t)bits << (16 * slot);
  else x.kb2 |= (uint64_t)bits << (16 * (slot - 4));
}
__device__ __forceinline__ uint32_t r16_kept(const RCtx& x, const int slot) {
  return (uint32_t)(((slot < 4) ? x.kb >> (16 * slot) : x.kb2 >> (16 * (slot - 4))) & 0xffffu);
}
// the finished 4 x 4 block of a thread -> the other activation buffer (+ the workspace array the weight-gradient launch
// reads); the buffers swap, ONE barrier
__device__ __forceinline__ void r16_publish(const RCtx& x, const f32x4 (&v)[4], float* gout) {
  const int q = x.lane >> 4, c = 64 * x.wave + 4 * (x.lane & 15);
#pragma unroll
  for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x4*>(x.hn + (4 * q + r) * RLD + c) = v[r];
  if (gout) {
#pragma unroll
    for (int r = 0; r < 4; ++r) r16_gst4(gout + (int64_t)(x.r0 + 4 * q + r) * 256 + c, v[r]);
  }
  float* t = x.hs; x.hs = x.hn; x.hn = t;
  __syncthreads();
}
__device__ __forceinline__ void r16_fw_finish(const RCtx& x, const f32x4 (&acc)[4], const f32x4 bv, const int keep,
                                              float* gout) {
  f32x4 v[4];
  uint32_t bits = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[r][e] = fmaxf(acc[e][r] + bv[e], 0.f);
      bits |= (v[r][e] > 0.f) ? (1u << (4 * r + e)) : 0u;
    }
  if (keep >= 0) r16_keep(x, keep, bits);
  r16_publish(x, v, gout);
}
__device__ __forceinline__ void r16_big_fwd(const RCtx& x, f32x4 (&wb)[2][16], const float* W, const float* bias,
                                            const int keep, float* gout, const RNext& next) {
  f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
  const f32x4 bv = ldv(bias + 64 * x.wave + 4 * (x.lane & 15));
  r16_big_steps(x, wb, W, next, acc);
  r16_fw_finish(x, acc, bv, keep, gout);
}
__device__ __forceinline__ void r16_big_bwdT(const RCtx& x, f32x4 (&wb)[2][16], const float* WT, const int mask,
                                             float* gout, const RNext& next) {
  f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
  r16_big_steps(x, wb, WT, next, acc);
  const uint32_t mk = r16_kept(x, mask);
  f32x4 v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int e = 0; e < 4; ++e) v[r][e] = ((mk >> (4 * r + e)) & 1u) ? acc[e][r] : 0.f;
  r16_publish(x, v, gout);
}

// ---- layer 0: hs <- relu(x . W0 + g . Wg + b0) over the virtual k = 4 t + q of [first S entries of xin | G from gofs on]
// (steps 0 .. 7 are in flight into wb[0]; inputs wider than 32 take further phases of 8 steps; one ds_read_b32 per step)
__device__ __forceinline__ void r16_l0_phase(const RCtx& x, const f32x4 (&use)[16], int S, int nk, int gofs, int s0,
                                             f32x4 (&acc)[4]) {
  const int q = x.lane >> 4, j = x.lane & 15;
  float av[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int kv = 4 * (s0 + i) + q;
    const bool ok = kv < nk;
    const int kc = ok ? kv : 0;
    const float v = x.xin[j * XLD + ((kc < S) ? kc : gofs + (kc - S))];
    av[i] = ok ? v : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = MFMA(av[i], use[i][e], acc[e]);
}
__device__ __forceinline__ void r16_l0_fwd(const RCtx& x, f32x4 (&wb)[2][16], const float* W0, int S, const float* Wg,
                                           int G, int gofs, const float* bias, const int keep, float* gout,
                                           const RNext& next) {
  f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
  const f32x4 bv = ldv(bias + 64 * x.wave + 4 * (x.lane & 15));
  const int nk = S + G, np = (nk + 31) >> 5;                 // phases of 8 steps = 32 inputs
  int p = 0;
  // pairs of phases: wb[0] -> wb[1] -> wb[0]; the successor's first phase follows the last one into wb[0]
#pragma unroll 1
  for (; p + 1 < np; p += 2) {
    r16_l0_load8(wb[1], W0, S, Wg, nk, x.wave, x.lane, 8 * (p + 1));
    __builtin_amdgcn_sched_barrier(0);
    r16_l0_phase(x, wb[0], S, nk, gofs, 8 * p, acc);
    __builtin_amdgcn_sched_barrier(0);
    if (p + 2 < np) r16_l0_load8(wb[0], W0, S, Wg, nk, x.wave, x.lane, 8 * (p + 2));
    else r16_prefetch(wb[0], next, x.wave, x.lane);
    __builtin_amdgcn_sched_barrier(0);
    r16_l0_phase(x, wb[1], S, nk, gofs, 8 * (p + 1), acc);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (p < np) {                                              // an odd number of phases: the prefetch goes out late
    r16_l0_phase(x, wb[0], S, nk, gofs, 8 * p, acc);
    __builtin_amdgcn_sched_barrier(0);
    r16_prefetch(wb[0], next, x.wave, x.lane);
    __builtin_amdgcn_sched_barrier(0);
  }
  r16_fw_finish(x, acc, bv, keep, gout);
}

// ---- a gradient that enters a backward chain through an output layer: thread (wave, q, j) writes its 4 x 4 block
// v(row, e) of hs (the CURRENT buffer: the caller's barriers bracket the site as for the other forms), masked by the kept
// layer `slot`, and stores it to g ([B][256]) when given
template <class F>
__device__ __forceinline__ void r16_seed(const RCtx& x, const int slot, float* g, F f) {
  const int q = x.lane >> 4, c = 64 * x.wave + 4 * (x.lane & 15);
  const uint32_t mk = r16_kept(x, slot);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = ((mk >> (4 * r + e)) & 1u) ? f(4 * q + r, e) : 0.f;
    *reinterpret_cast<f32x4*>(x.hs + (4 * q + r) * RLD + c) = v;
    if (g) r16_gst4(g + (int64_t)(x.r0 + 4 * q + r) * 256 + c, v);
  }
}

// ---- thin layers on the matrix unit: out[16 rows][<= 4 columns] = hs[16][256] . B[256][<= 4] -- the output layers (pi:
// 256 -> 4, Q: 256 -> 1) and the action-slot product of the critic's backward pass (dY0 . Wu^T).  The 4- / 8-row forms give
// every row to a wave and reduce over its 64 lanes with DPP steps (rows_head4 / rows_head1: ~200 vector instructions per
// row); with three workgroups on a CU those instructions queue behind the other workgroups' matrix instructions -- the
// stamps of tools/rows_stamps.py showed the pi head of 16 rows taking as long as a hidden layer (26 k cycles).  Here wave w
// multiplies its quarter of k (16 instructions: columns >= 4 of B are zero), the four partial [16 x 4] tiles meet in the
// idle activation buffer, and lane L < 16 of wave w finishes out[4 w + (L >> 2)][L & 3].
// B fragments: b[4 c + d] = B[64 w + 16 c + 4 q + d][j] for j < 4 (<1: head1), else 0 -- the k order of r16_big_steps.
__device__ __forceinline__ void r16_frag_cols4(float (&b)[16], const float* W, int wave, int lane) {   // W[256][4]
  const int q = lane >> 4, j = lane & 15, jc = j & 3;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float v = W[(64 * wave + 16 * (i >> 2) + 4 * q + (i & 3)) * 4 + jc];
    b[i] = (j < 4) ? v : 0.f;
  }
}
__device__ __forceinline__ void r16_frag_rows(float (&b)[16], const float* Wr, int nrows, int wave, int lane) {   // Wr[nrows][256]: B[k][j] = Wr[j][k]
  const int q = lane >> 4, j = lane & 15, jc = (j < nrows) ? j : 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const f32x4 v = ldv(Wr + (int64_t)jc * 256 + 64 * wave + 16 * c + 4 * q);
#pragma unroll
    for (int d = 0; d < 4; ++d) b[4 * c + d] = (j < nrows) ? v[d] : 0.f;
  }
}
__device__ __forceinline__ f32x4 r16_thin(const RCtx& x, const float (&b)[16]) {
  const float* ar = x.hs + (x.lane & 15) * RLD + 64 * x.wave + 4 * (x.lane >> 4);
  f32x4 acc = zero4();
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const f32x4 a4 = lds4(ar + 16 * c);
#pragma unroll
    for (int d = 0; d < 4; ++d) acc = MFMA(a4[d], b[4 * c + d], acc);
  }
  return acc;
}
// (one barrier inside; the caller's next barrier separates the reads from whoever writes hn next)
__device__ __forceinline__ float r16_thin_sum(const RCtx& x, const f32x4 acc) {
  const int q = x.lane >> 4, j = x.lane & 15;
  if (j < 4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) x.hn[x.wave * 64 + (4 * q + i) * 4 + j] = acc[i];
  }
  __syncthreads();
  const float* p = x.hn + (4 * x.wave + ((x.lane & 15) >> 2)) * 4 + (x.lane & 3);
  return (p[0] + p[64]) + (p[128] + p[192]);
}
// the value of lane (row, d) of a quad broadcast to the quad: t[d] for d = 0 .. 3
__device__ __forceinline__ void r16_quad(const float v, float (&t)[4]) {
  t[0] = dpp_mov<0x00>(v); t[1] = dpp_mov<0x55>(v); t[2] = dpp_mov<0xAA>(v); t[3] = dpp_mov<0xFF>(v);
}
