// Layer routines of the row-local kernels (included by mlp_rows.h): weight-fragment loads, the chunk loop of a 256 x 256
// hidden layer on v_mfma_f32_4x4x1, epilogues, layer 0, output layers, input rows, the leading-argument description.
#pragma once

// ---- weight fragments of one 16-deep k-chunk
__device__ __forceinline__ void rows_fw_load(f32x4 (&b)[16], const float* W, int wave, int lane, int c) {
  const float* p = W + (int64_t)(64 * wave + 16 * c) * 256 + 4 * lane;
#pragma unroll
  for (int i = 0; i < 16; ++i) b[i] = ldv(p + (int64_t)i * 256);
}
// (rows 4 kq .. 4 kq + 3 of the chunk: the hidden layers issue a chunk's loads 4 at a time, in front of each group of 16
//  matrix instructions -- a wave that issues 16 loads in a row is held until the texture path has taken them all, and its
//  matrix instructions wait behind them: tools/rowchain2_lab.hip, 2.34 -> 2.23 us per layer)
__device__ __forceinline__ void rows_fw_load4(f32x4 (&b)[16], const float* W, int wave, int lane, int c, int kq) {
  const float* p = W + (int64_t)(64 * wave + 16 * c + 4 * kq) * 256 + 4 * lane;
#pragma unroll
  for (int i = 0; i < 4; ++i) b[4 * kq + i] = ldv(p + (int64_t)i * 256);
}
// ---- what a layer routine loads ahead for its successor (weights do not depend on activations): the successor's
// first chunk lands in wb[0] while this layer's last chunk is multiplied / its epilogue runs
enum { RN_NONE = 0, RN_FWD = 1, RN_L0 = 3 };
// (off: added to W where the prefetch is issued, not where the descriptor is built -- an offset that was itself just
//  fetched from the arguments is then waited for behind the layer's own loads, rows_hidden_fwd)
struct RNext { int kind; const float* W; int S; const float* Wg; int nk; int64_t off; };     // RN_L0: W = W0
__device__ __forceinline__ RNext rnext(int kind, const float* W, int S = 0, const float* Wg = nullptr, int nk = 0,
                                       int64_t off = 0) {
  RNext n;
  n.kind = kind; n.W = W; n.S = S; n.Wg = Wg; n.nk = nk; n.off = off;
  return n;
}
// layer-0 rows of the virtual k = 4 (t0 + t) + wave, t = 0..15 (rows past the end are clamped to row 0 and ignored)
__device__ __forceinline__ void rows_l0_load(f32x4 (&b)[16], const float* W0, int S, const float* Wg, int nk, int wave,
                                             int lane, int t0) {
  // (row offsets in 32 bits relative to W0 -- both pieces live in one parameter vector --: a 64-bit pointer select per row
  //  cost ~10 vector instructions a row, 1.3 k cycles for the 16 rows a hidden layer requests ahead for a layer 0)
  const int dg = (int)(Wg - W0) - S * 256;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int kv = 4 * (t0 + t) + wave;
    const int kc = (kv < nk) ? kv : 0;
    const int off = kc * 256 + ((kc < S) ? 0 : dg) + 4 * lane;
    b[t] = ldv(W0 + off);
  }
}
__device__ __forceinline__ void rows_l0_load4(f32x4 (&b)[16], const float* W0, int S, const float* Wg, int nk, int wave,
                                              int lane, int t0, int kq) {
  const int dg = (int)(Wg - W0) - S * 256;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = 4 * kq + i;
    const int kv = 4 * (t0 + t) + wave;
    const int kc = (kv < nk) ? kv : 0;
    const int off = kc * 256 + ((kc < S) ? 0 : dg) + 4 * lane;
    b[t] = ldv(W0 + off);
  }
}
__device__ __forceinline__ void rows_prefetch4(f32x4 (&b)[16], const RNext& n, int wave, int lane, int kq) {
  if (n.kind == RN_FWD) rows_fw_load4(b, n.W + n.off, wave, lane, 0, kq);
  else if (n.kind == RN_L0) rows_l0_load4(b, n.W, n.S, n.Wg, n.nk, wave, lane, 0, kq);
}
__device__ __forceinline__ void rows_prefetch(f32x4 (&b)[16], const RNext& n, int wave, int lane) {
  if (n.kind == RN_FWD) rows_fw_load(b, n.W + n.off, wave, lane, 0);
  else if (n.kind == RN_L0) rows_l0_load(b, n.W, n.S, n.Wg, n.nk, wave, lane, 0);
}
#include "mlp_rows16.h"      // 16 rows per workgroup on v_mfma_f32_16x16x4 (the dispatch sits in the routines below)
// R batch rows per workgroup (4, or 8 for batches of several ranks / experts -- mlp_rows.h): one 16-byte load of W feeds
// R instructions; acc[h] are rows 4 h .. 4 h + 3
template <int R>
__device__ __forceinline__ void rows_fw_mac4(const f32x4 (&b)[16], const float* hs, int wave, int lane, int c, int kq,
                                             f32x4 (&acc)[R / 4][4]) {
  f32x4 a[R / 4];
#pragma unroll
  for (int h = 0; h < R / 4; ++h)
    a[h] = *reinterpret_cast<const f32x4*>(hs + (4 * h + (lane & 3)) * RLD + 64 * wave + 16 * c + 4 * kq);
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int h = 0; h < R / 4; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[h][e] = MFMA4(a[h][s], b[4 * kq + s][e], acc[h][e]);
}
// the chunk loop of a hidden layer (forward, or backward on the transposed copy): chunk c + 1 (the successor's first chunk
// behind the last one) is requested in four pieces between the four groups of matrix instructions of chunk c
template <int R>
__device__ __forceinline__ void rows_big_chunks(const RCtx& x, f32x4 (&wb)[2][16], const float* W, const RNext& next,
                                                const bool lean, const int64_t late_off, f32x4 (&acc)[R / 4][4]) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    RNext n = next;
    if (c == 3 && lean) n.off += late_off;                    // (lean <=> another hidden layer of the network follows)
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
      if (c < 3) rows_fw_load4(wb[(c + 1) & 1], W, x.wave, x.lane, c + 1, kq);
      else rows_prefetch4(wb[0], n, x.wave, x.lane, kq);
      __builtin_amdgcn_sched_barrier(0);
      rows_fw_mac4<R>(wb[c & 1], x.hs, x.wave, x.lane, c, kq, acc);
      __builtin_amdgcn_sched_barrier(0);
    }
    ROWS_DBG2(x);
  }
}
// ---- epilogues: partial tiles -> LDS -> finished rows (next layer's input in hs, optional copies)
// forward: acc[e][r] = partial of out[row r][column 4 lane + e] over this wave's k quarter
// bv = bias[tid], loaded by the caller at the start of the layer (not here: its latency would be exposed)
// lean = false: two workgroup barriers (partials published | finished rows published).
// lean (the update chains, round 4): ONE.  Thread tid finishes column tid = 64 wave + lane, and the next layer's
// wave reads exactly the columns [64 wave, 64 wave + 64) of hs as its A operand: the finished rows never cross a wave on
// their way into the next hidden layer, and LDS operations of one wave execute in order.  What would race without the
// second barrier is the NEXT layer's partials against a slow wave still summing this layer's: the partials alternate
// between two buffers (x.part / x.part2), and a buffer comes round again only behind the barrier of the layer in between.
// A layer whose rows ARE read across waves next (an output layer, the action-slot product) keeps both.
// While no wave is held at the issue of a load the CU's fill path idles (tools/rowchain2_lab.hip: every cycle of epilogue
// is a cycle added to the layer), so the epilogue is kept short: all 16 partials are requested before the first sum, the
// copies and stores sit behind one uniform branch each instead of one per row.
template <int R>
__device__ __forceinline__ void rows_finish_sums(const RCtx& x, const f32x4 (&acc)[R / 4][4], float (&s)[R], const bool lean) {
  float* part = x.part;
#pragma unroll
  for (int h = 0; h < R / 4; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const f32x4 v = {acc[h][0][r], acc[h][1][r], acc[h][2][r], acc[h][3][r]};
      *reinterpret_cast<f32x4*>(part + (x.wave * R + 4 * h + r) * 256 + 4 * x.lane) = v;
    }
  __syncthreads();
  float p[R][4];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int j = 0; j < 4; ++j) p[r][j] = part[(j * R + r) * 256 + x.tid];
#pragma unroll
  for (int r = 0; r < R; ++r) s[r] = (p[r][0] + p[r][1]) + (p[r][2] + p[r][3]);
  if (lean) { x.part = x.part2; x.part2 = part; }
}
// ---- relu' masks of the kept layers (slot = layer, + nl for the second network of the actor side).  Thread tid finishes
// column tid of every row and is the only one that ever asks whether (row r, column tid) of a kept layer was positive.
// 4 rows: the kept activations lie in LDS (x.keep: [slot][4][256]; the form the single-rank kernel was tuned with).  8 rows:
// the masks are bits of a per-thread word, 8 to a layer -- 2 nl x 8 KB of kept activations would leave no room for a second
// workgroup on the CU.
template <int R>
__device__ __forceinline__ void rows_keep(const RCtx& x, const int slot, const float (&s)[R]) {
  if (R == 4) {
    float* keep = x.keep + slot * 1024;
#pragma unroll
    for (int r = 0; r < R; ++r) keep[r * 256 + x.tid] = s[r];
    return;
  }
  uint32_t bits = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) bits |= (s[r] > 0.f) ? (1u << r) : 0u;
  x.kb |= (uint64_t)bits << (slot * R);
}
template <int R>
__device__ __forceinline__ uint32_t rows_kept(const RCtx& x, const int slot) {
  if (R == 4) {
    const float* keep = x.keep + slot * 1024;
    float k[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) k[r] = keep[r * 256 + x.tid];
    uint32_t bits = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) bits |= (k[r] > 0.f) ? (1u << r) : 0u;
    return bits;
  }
  return (uint32_t)(x.kb >> (slot * R));
}
// keep: slot of the layer's relu' mask (rows_keep) or -1
template <int R>
__device__ __forceinline__ void rows_fw_finish(const RCtx& x, const f32x4 (&acc)[R / 4][4], const float bv, const int keep,
                                               float* gout, const bool lean = false) {
  float s[R];
  rows_finish_sums<R>(x, acc, s, lean);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    s[r] = fmaxf(s[r] + bv, 0.f);
    x.hs[r * RLD + x.tid] = s[r];
  }
  if (keep >= 0) rows_keep<R>(x, keep, s);
  if (gout) {
#pragma unroll
    for (int r = 0; r < R; ++r) rows_gst(x, gout + (int64_t)(x.r0 + r) * 256 + x.tid, s[r]);
  }
  if (!lean) __syncthreads();
}
// ---- one 256 x 256 hidden layer, forward: hs <- relu(hs . W + bias)
// (the first chunk of W is already in flight into wb[0]: rows_prefetch of the predecessor)
// late_off: added to next.off of a lean layer where the prefetch is issued (a value that may still be on its way from the
// arguments when the layer starts)
template <int R>
__device__ __forceinline__ void rows_acc_zero(f32x4 (&acc)[R / 4][4]) {
#pragma unroll
  for (int h = 0; h < R / 4; ++h)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[h][e] = zero4();
}
template <int R = 4>
__device__ __forceinline__ void rows_big_fwd(const RCtx& x, f32x4 (&wb)[2][16], const float* W, const float* bias,
                                             const int keep, float* gout, const RNext& next, const bool lean = false,
                                             const int64_t late_off = 0) {
  if constexpr (R == ROWS_R3) {
    RNext n = next;
    if (lean) n.off += late_off;
    r16_big_fwd(x, wb, W, bias, keep, gout, n);
    return;
  }
  f32x4 acc[R / 4][4];
  rows_acc_zero<R>(acc);
  const float bv = bias[x.tid];
  rows_big_chunks<R>(x, wb, W, next, lean, late_off, acc);
  rows_fw_finish<R>(x, acc, bv, keep, gout, lean);
  ROWS_DBG(x);                                               // (one stamp per layer: finer ones slow the measured group down)
}
// ---- one 256 x 256 hidden layer, backward on the TRANSPOSED matrix: hs <- (hs . WT) * relu'(mask), WT[n][k] = W[k][n].
// The forward product with another epilogue (no bias; the kept activation of the layer below gates the gradient).
// (mask: the slot of the kept layer below)
template <int R>
__device__ __forceinline__ void rows_big_bwdT(const RCtx& x, f32x4 (&wb)[2][16], const float* WT, const int mask,
                                              float* gout, const RNext& next, const bool lean = false) {
  if constexpr (R == ROWS_R3) {
    r16_big_bwdT(x, wb, WT, mask, gout, next);
    return;
  }
  f32x4 acc[R / 4][4];
  rows_acc_zero<R>(acc);
  rows_big_chunks<R>(x, wb, WT, next, false, 0, acc);
  const uint32_t mk = rows_kept<R>(x, mask);
  float s[R];
  rows_finish_sums<R>(x, acc, s, lean);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    s[r] = ((mk >> r) & 1u) ? s[r] : 0.f;
    x.hs[r * RLD + x.tid] = s[r];
  }
  if (gout) {
#pragma unroll
    for (int r = 0; r < R; ++r) rows_gst(x, gout + (int64_t)(x.r0 + r) * 256 + x.tid, s[r]);
  }
  if (!lean) __syncthreads();
}

// ---- layer 0: hs <- relu(x . W0 + g . Wg + b0).  The input row in LDS is xin[i] = [o | td | action slot | g]: the first
// S entries meet the S rows of W0 (S excludes the action slot for an actor), the G entries from `gofs` on meet Wg
// (util.py:79-92).  Wave w takes the virtual k = 4 t + w of the concatenation.
template <int R>
__device__ __forceinline__ void rows_l0_mac(const RCtx& x, const f32x4 (&b)[16], int S, int nk, int gofs, int t0,
                                            f32x4 (&acc)[R / 4][4]) {
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int kv = 4 * (t0 + t) + x.wave;
    const bool ok = kv < nk;
    const int kc = ok ? kv : 0;
    const int col = (kc < S) ? kc : gofs + (kc - S);
#pragma unroll
    for (int h = 0; h < R / 4; ++h) {
      const float v = x.xin[(4 * h + (x.lane & 3)) * XLD + col];
      const float av = ok ? v : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[h][e] = MFMA4(av, b[t][e], acc[h][e]);
    }
  }
}
// (rows t < 16 are already in flight into wb[0]; a second pass covers inputs wider than 64)
// bv = bias[tid], loaded by the caller ahead of the layer (for a group's first layer: together with its inputs -- the
// parameters were just rewritten by the optimiser, a load issued here would be a second cold round trip)
template <int R = 4>
__device__ __forceinline__ void rows_l0_fwd(const RCtx& x, f32x4 (&wb)[2][16], const float* W0, int S, const float* Wg,
                                            int G, int gofs, const float bv, const float* b0p, const int keep,
                                            float* gout, const RNext& next, const bool lean = false) {
  // (b0p: the bias vector itself -- the 16-row form's threads own 4 columns each and fetch their own)
  if constexpr (R == ROWS_R3) {
    r16_l0_fwd(x, wb, W0, S, Wg, G, gofs, b0p, keep, gout, next);
    return;
  }
  f32x4 acc[R / 4][4];
  rows_acc_zero<R>(acc);
  const int nk = S + G;
  const bool two = nk > 64;
  if (two) rows_l0_load(wb[1], W0, S, Wg, nk, x.wave, x.lane, 16);
  __builtin_amdgcn_sched_barrier(0);
  rows_l0_mac<R>(x, wb[0], S, nk, gofs, 0, acc);
  rows_prefetch(wb[0], next, x.wave, x.lane);
  __builtin_amdgcn_sched_barrier(0);
  if (two) rows_l0_mac<R>(x, wb[1], S, nk, gofs, 16, acc);
  rows_fw_finish<R>(x, acc, bv, keep, gout, lean);
}

// ---- output layers: wave i finishes batch row r0 + i; the result is uniform over the wave.  The output-layer weights
// are fetched into registers BEFORE the network's hidden layers run (rows_head*_w), so that no load latency sits
// between the last hidden layer and the head.
struct HeadW4 { f32x4 w[4]; };
__device__ __forceinline__ HeadW4 rows_head4_w(const float* Wout, int lane) {      // Wout[256][4]: rows 4 lane .. +3
  HeadW4 h;
#pragma unroll
  for (int e = 0; e < 4; ++e) h.w[e] = ldv(Wout + (int64_t)(4 * lane + e) * 4);
  return h;
}
// (row: which of the workgroup's rows -- wave i finishes rows i, i + 4, ...)
__device__ __forceinline__ void rows_head4(const RCtx& x, const HeadW4& h, float (&out)[4], const int row) {
  const f32x4 h4 = *reinterpret_cast<const f32x4*>(x.hs + row * RLD + 4 * x.lane);
#pragma unroll
  for (int d = 0; d < 4; ++d)
    out[d] = wave_sum(h4[0] * h.w[0][d] + h4[1] * h.w[1][d] + h4[2] * h.w[2][d] + h4[3] * h.w[3][d]);
}
__device__ __forceinline__ void rows_head4(const RCtx& x, const HeadW4& h, float (&out)[4]) { rows_head4(x, h, out, x.wave); }
__device__ __forceinline__ float rows_head1(const RCtx& x, const f32x4& w, const int row) {   // w = Wout[4 lane .. +3] of a [256][1]
  const f32x4 h4 = *reinterpret_cast<const f32x4*>(x.hs + row * RLD + 4 * x.lane);
  return wave_sum(h4[0] * w[0] + h4[1] * w[1] + h4[2] * w[2] + h4[3] * w[3]);
}
__device__ __forceinline__ float rows_head1(const RCtx& x, const f32x4& w) { return rows_head1(x, w, x.wave); }

// layer-0 input rows of the workgroup's R batch rows: xin[i] = [o | td | action slot | g]; the action slot receives
// the batch action / max_u (actor_critic.py:96) when with_u, else it is filled later from the actor's output
// keep: where the rows are also stored for the layer-0 weight gradients (input normalisation only: without it those
// read the batch itself), or NULL
// so: the expert's slab offset (batched experts: every expert has its own normalisers, expert_stride floats apart)
// In two halves: the raw loads (rows_inputs_issue: needs nothing but the batch pointer, the row stride, the column offsets
// and the widths -- RowsPre below hands those over in scalar registers, so the loads go out before the first argument has
// arrived from memory) and what is done to the values (rows_inputs_commit: action / max_u, normalisation, the LDS rows).
struct RowsIn { int dimo, dimtd, dimg, ld, off_td, off_u; };
#define ROWS_IN_IT(R) (((R) * ROWS_MAXIN + 255) / 256)        // elements per thread
template <int R>
__device__ __forceinline__ void rows_inputs_issue(const int tid, const int r0, const RowsIn& in, const float* batch,
                                                  int off_o, int off_g, bool with_u, float (&v)[ROWS_IN_IT(R)]) {
  const int Sa = in.dimo + in.dimtd, S = Sa + 4, tot = S + in.dimg;
#pragma unroll
  for (int it = 0; it < ROWS_IN_IT(R); ++it) {
    const int idx = tid + 256 * it;
    v[it] = 0.f;
    if (idx < R * tot) {
      const int i = idx / tot, k = idx - i * tot;
      const float* row = batch + (int64_t)(r0 + i) * in.ld;
      if (k < in.dimo) v[it] = row[off_o + k];
      else if (k < Sa) v[it] = row[in.off_td + (k - in.dimo)];
      else if (k < S) { if (with_u) v[it] = row[in.off_u + (k - Sa)]; }
      else v[it] = row[off_g + (k - S)];
    }
  }
}
template <int R>
__device__ __forceinline__ void rows_inputs_commit(const RCtx& x, const RowsArgs& a, bool with_u,
                                                   const float (&raw)[ROWS_IN_IT(R)], float* keep = nullptr,
                                                   const int64_t so = 0) {
  const int Sa = a.dimo + a.dimtd, S = Sa + 4, tot = S + a.dimg;
#pragma unroll
  for (int it = 0; it < ROWS_IN_IT(R); ++it) {
    const int idx = x.tid + 256 * it;
    if (idx < R * tot) {
      const int i = idx / tot, k = idx - i * tot;
      float v = raw[it];
      if (k < a.dimo) {
        if (a.o_mean) v = fclip(fdiv(__fsub_rn(v, a.o_mean[so + k]), a.o_std[so + k]), -a.nclip, a.nclip);   // normalizer.py:72-77
      } else if (k < Sa) {
      } else if (k < S) {
        v = with_u ? fdiv(v, a.max_u) : 0.f;
      } else {
        if (a.g_mean) v = fclip(fdiv(__fsub_rn(v, a.g_mean[so + k - S]), a.g_std[so + k - S]), -a.nclip, a.nclip);
      }
      x.xin[i * XLD + k] = v;
      if (keep) keep[(int64_t)(x.r0 + i) * XLD + k] = v;
    }
  }
}
__device__ __forceinline__ RowsIn rows_in_of(const RowsArgs& a) {
  RowsIn in;
  in.dimo = a.dimo; in.dimtd = a.dimtd; in.dimg = a.dimg; in.ld = a.ld; in.off_td = a.off_td; in.off_u = a.off_u;
  return in;
}

// ---- the kernel's LEADING arguments (round 4): with -mllvm -amdgpu-kernarg-preload-count they arrive in scalar registers
// with the wave.  A row group needs 1.4 k cycles to get its first argument from memory (tools/rows_lab.hip) and only then
// could request its input rows and its first layer-0 weights -- another cold round trip.  With these 14 dwords it knows its
// role, its rows and its first matrix at once: the two round trips overlap.  Host-checked (flag in k[4]): single agent on
// the XCD-aware grid, no input normalisation, 16-bit offsets, Wg right behind W0's bias row (the parameter layout of
// util.py:79-92 as this library stores it); otherwise the flag is clear and the arguments are fetched as before.
struct RowsPre {
  const float* w0[3];          // layer-0 matrices W0 of: main actor | target actor | main critic
  const float* batch;
  uint32_t k[6];               // ld | off_o << 16,  off_td | off_g << 16,  off_o2 | off_g2 << 16,
                               // off_u | dimo << 16 | dimtd << 24,  B | dimg << 16 | ok << 25,  (spare)
};
#define ROWS_PRE_PARAMS const float* pw0a, const float* pw0t, const float* pw0c, const float* pbatch, const uint32_t pk0, \
                        const uint32_t pk1, const uint32_t pk2, const uint32_t pk3, const uint32_t pk4, const uint32_t pk5
#define ROWS_PRE_MAKE(pre) RowsPre pre; pre.w0[0] = pw0a; pre.w0[1] = pw0t; pre.w0[2] = pw0c; pre.batch = pbatch; \
                           pre.k[0] = pk0; pre.k[1] = pk1; pre.k[2] = pk2; pre.k[3] = pk3; pre.k[4] = pk4; pre.k[5] = pk5

// hidden layers 1 .. nl-1 of a network, forward; which: 0 nothing stored, 1 -> a.actc[l], 2 -> a.acta[l]
// (the kernarg arrays are indexed in place: handing their address around would copy the struct to scratch memory)
// `after` = what follows the network's last hidden layer
// (the network's last hidden layer keeps both barriers: an output layer reads its rows across the waves)
// The offsets of layer l + 1 are fetched while layer l runs and carried in registers: read at the top of a layer, as
// `th + N.W[l]`, they put two dependent scalar-load waits (~300 cycles) between the barrier and the layer's first operand
// load -- with the fill path idle.
// keep0: slot of the network's layer 0 (rows_keep) or -1
template <int R>
__device__ __forceinline__ void rows_hidden_fwd(const RCtx& x, f32x4 (&wb)[2][16], const RowsArgs& a, const RowsNet& N,
                                                const float* th, const int keep0, int which, int64_t eo,
                                                const RNext& after) {
  if (a.nl == 1) __syncthreads();                            // (layer 0 ran with one barrier: rows_fw_finish)
  int Wc = N.W[1], bc = N.b[1];
  for (int l = 1; l < a.nl; ++l) {
    const bool more = l + 1 < a.nl;
    const int ln = more ? l + 1 : l;
    const int Wn = N.W[ln], bn = N.b[ln];                    // (consumed by the prefetch behind this layer's third chunk)
    float* g = (which == 1) ? a.actc[l] + eo : (which == 2) ? a.acta[l] + eo : nullptr;
    const int kp = keep0 >= 0 ? keep0 + l : -1;
    rows_big_fwd<R>(x, wb, th + Wc, th + bc, kp, g, more ? rnext(RN_FWD, th) : after, more, Wn);
    Wc = Wn; bc = bn;
  }
}
// hidden layers nl-1 .. 1 of a network, backward on the transposed copies; which: 0 critic, nothing stored,
// 1 critic -> a.dactc[l-1], 2 actor -> a.dacta[l-1]
template <int R>
__device__ __forceinline__ void rows_hidden_bwd(const RCtx& x, f32x4 (&wb)[2][16], const RowsArgs& a,
                                                const int keep0, int which, int64_t eo, const RNext& after) {
  const float* wt = (which == 2) ? a.wTpi[a.nl - 1] : a.wTq[a.nl - 1];
  for (int l = a.nl - 1; l >= 1; --l) {
    const int ln = (l > 1) ? l - 1 : l;
    const float* wn = (which == 2) ? a.wTpi[ln] : a.wTq[ln];            // (fetched while layer l runs: rows_hidden_fwd)
    float* g = (which == 1) ? a.dactc[l - 1] + eo : (which == 2) ? a.dacta[l - 1] + eo : nullptr;
    rows_big_bwdT<R>(x, wb, wt + eo, keep0 + (l - 1), g, (l > 1) ? rnext(RN_FWD, wn, 0, nullptr, 0, eo) : after, l > 1);
    wt = wn;
  }
}
// what the layer in front of a network's backward pass loads ahead: the first chunk of its top hidden matrix
__device__ __forceinline__ RNext rows_bwd_first(const RowsArgs& a, bool actor, int64_t eo) {
  return rnext(RN_FWD, (actor ? a.wTpi[a.nl - 1] : a.wTq[a.nl - 1]) + eo);
}

// What a row group requests first: the first 16 rows of its first layer-0 matrix and its input rows.  The description comes
// from the leading arguments (pre_path: nothing is read from the argument segment) or from the arguments proper; one load
// sequence either way.  kind: 0 actor side (main actor), 1 target (target actor), 2 main critic.
template <bool EX, int R>
__device__ __forceinline__ void rows_first_loads(const RCtx& x, const RowsArgs& a, const Ex& ex, const RowsPre* pre,
                                                 const bool pre_path, const int kind, const int rgrp, const int expert,
                                                 f32x4 (&wb0)[16], float (&xraw)[ROWS_IN_IT(R)]) {
  const float *fW0, *fbatch;
  int fS, fWg_off, foff_o, foff_g;
  RowsIn fin;
  if (pre_path) {
    fin.ld = (int)(pre->k[0] & 0xffffu); fin.off_td = (int)(pre->k[1] & 0xffffu); fin.off_u = (int)(pre->k[3] & 0xffffu);
    fin.dimo = (int)((pre->k[3] >> 16) & 0xffu); fin.dimtd = (int)(pre->k[3] >> 24); fin.dimg = (int)((pre->k[4] >> 16) & 0xffu);
    fS = fin.dimo + fin.dimtd + ((kind == 2) ? 4 : 0);
    fW0 = (kind == 0) ? pre->w0[0] : (kind == 1) ? pre->w0[1] : pre->w0[2];
    fWg_off = (fS + 1) * 256;
    fbatch = pre->batch;
    foff_o = (kind == 1) ? (int)(pre->k[2] & 0xffffu) : (int)(pre->k[0] >> 16);
    foff_g = (kind == 1) ? (int)(pre->k[2] >> 16) : (int)(pre->k[1] >> 16);
  } else {
    int64_t eo;
    (void)ex_decode<EX>(ex, expert, eo);
    fin = rows_in_of(a);
    fS = fin.dimo + fin.dimtd + ((kind == 2) ? 4 : 0);
    const RowsNet& N = (kind == 0) ? a.mPi : (kind == 1) ? a.tPi : a.mQ;
    fW0 = N.th + eo + N.W0;
    fWg_off = N.Wg - N.W0;
    fbatch = a.batch + eo;
    foff_o = (kind == 1) ? a.off_o2 : a.off_o;
    foff_g = (kind == 1) ? a.off_g2 : a.off_g;
    asm volatile("" : "+s"(fS), "+s"(foff_o), "+s"(foff_g));   // (no select between the two descriptions: see the caller)
  }
  if constexpr (R == ROWS_R3) r16_l0_load8(wb0, fW0, fS, fW0 + fWg_off, fS + fin.dimg, x.wave, x.lane, 0);
  else rows_l0_load(wb0, fW0, fS, fW0 + fWg_off, fS + fin.dimg, x.wave, x.lane, 0);
  rows_inputs_issue<R>(x.tid, rgrp * R, fin, fbatch, foff_o, foff_g, kind == 2, xraw);
  __builtin_amdgcn_sched_barrier(0);
}

