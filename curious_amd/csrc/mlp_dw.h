// Weight-gradient tiles with Adam in the epilogue, the small-problem tiles, the loss finalisation and the update's tail
// launch dw_adam_her_kernel / dw_all_kernel (included by mlp.hip after mlp_lean_gemm.h: GemmHot, hot_store).
#pragma once

// C[K',N] = A[M,K']^T . B[M,N];  aux_out[N] = colsum(B)    (reduction over P.M)     1-D grid over a tile list
struct DwHotArgs { GemmHot p[4]; int32_t tiles_per; int32_t nprob; };   // every problem has tiles_per tiles
// ---- split reduction (batches of many chunks of 256 rows: virtual ranks).  The tile list of a launch does not grow with
// the batch -- 256 hidden tiles + ~80 small ones at Arm4, for 256 CUs -- while every tile's reduction does: at 19 ranks a
// tile is 30-40 us of work, the CUs that were dealt two of them decide the launch's duration and the others idle for half
// of it (tools/dw_stamps.py; tools/dw_lab.hip: the hidden tiles alone 28 us, the launch 71).  With S > 1 (grid.z) a tile
// is worked on by S workgroups, each taking a segment of the chunks: S times as many, shorter work items, which the
// dispatcher spreads evenly.  A workgroup leaves its partial tile (16 x 64 + the 64 column sums) in the workspace, stored
// THROUGH the L2 (agent scope: the L2s of the 8 XCDs are not coherent with each other), waits for the acknowledgements and
// takes a ticket; whoever draws the last ticket of a tile adds the S partial tiles IN SEGMENT ORDER (the same sums whoever
// it is), resets the ticket counter for the next launch and goes on with the tile's epilogue -- store, optimiser, copies --
// as the only workgroup always did.  Nobody waits for anybody.
struct DwSplit { float* pbuf; int32_t* cnt; };               // workspace: [tiles][S][DW_PART] floats, [tiles] tickets (zero between launches)
struct DwSeg { int seg, S; };                                // this workgroup's segment of its tile's S (dw_role)
#define DW_PART 1088                                         // floats of a partial tile: 16 x 64 + 64 column sums
#define DW_SPLIT_MIN_B 1024                                  // batches from this size on have the workspace for it
#define DW_SPLIT_MAX 8                                       // segments per tile at most
#define DW_SPLIT_TILES (4 * 64 + 12 * 16)                    // tiles of a launch at most: 4 hidden matrices + MAX_DW_SMALL x 16 slots
#define DW_SC1 16                                            // aux bit of a raw buffer access: agent scope
typedef unsigned int dw_u32x4 __attribute__((ext_vector_type(4)));
// true: this workgroup finishes the tile (v / gb then hold the whole sums); red: LDS scratch (one word)
__device__ __forceinline__ bool dw_split_combine(const DwSplit& sp, const int gt, const int S, const int seg, const int tid,
                                                 f32x4& v, float& gb, float* red) {
  if (S == 1) return true;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(sp.pbuf + (int64_t)gt * S * DW_PART, 0, 0x7fffffff,
                                                                     0x00020000);
  if (tid < 256) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(dw_u32x4, v), r, (seg * DW_PART + 4 * tid) * 4, 0, DW_SC1);
  if (tid < 64) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, gb), r, (seg * DW_PART + 1024 + tid) * 4, 0, DW_SC1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int* flag = reinterpret_cast<int*>(red);
  if (tid == 0) {
    const int old = __hip_atomic_fetch_add(sp.cnt + gt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = old == S - 1;
    if (last) __hip_atomic_store(sp.cnt + gt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = last;
  }
  __syncthreads();
  const bool last = *flag != 0;
  __syncthreads();
  if (!last) return false;
  f32x4 sv = zero4();
  float sb = 0.f;
  for (int k = 0; k < S; ++k) {                               // (segment order: whoever is last adds the same way)
    if (tid < 256) sv += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (k * DW_PART + 4 * tid) * 4, 0, DW_SC1));
    if (tid < 64) sb += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (k * DW_PART + 1024 + tid) * 4, 0, DW_SC1));
  }
  v = sv; gb = sb;
  return true;
}
// the rows [m_lo, m_hi) segment `seg` of S takes of M = 256 C
__device__ __forceinline__ void dw_split_range(const int M, const int S, const int seg, int& m_lo, int& m_hi) {
  const int C = M >> 8;
  m_lo = ((seg * C) / S) << 8;
  m_hi = (((seg + 1) * C) / S) << 8;
}

// PIPE: the reduction has several chunks and is software-pipelined (below) -- a kernel of its own: the one-chunk kernel of
// a single rank's batch keeps its registers and its three workgroups on a CU.
// (Measured and not kept, round 5: 8 or 16 waves per workgroup splitting the chunks of ONE tile -- two waves of a SIMD then
//  run in lockstep, loads against loads and matrix instructions against matrix instructions: 73.5 -> 71.4 us at 19 ranks.)
template <bool ADAM, bool PIPE>
__device__ DW_INLINE void dw_hot_tile(const GemmHot& P, const AdamFuse& A, const int t, float* red,
                                   const int64_t eo, const int64_t eg, DwStamp* stamps = nullptr,
                                   const AdamEarly* given = nullptr, const DwSplit* sp = nullptr, const int gt = 0,
                                   const DwSeg sg = DwSeg{0, 1}) {
  const int S = PIPE ? sg.S : 1, seg = PIPE ? sg.seg : 0;     // (split reduction: dw_split_combine)
  int m_lo = 0, m_hi = P.M;
  if (S > 1) dw_split_range(P.M, S, seg, m_lo, m_hi);
  const int nx = P.N >> 6;
  int by, bx;
  tile_divmod(t, nx, by, bx);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, q = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int k0 = by * 16, n0 = bx * 64;
  const float* xu = P.A + eo + k0;                           // (uniform parts; lane parts below, ld_su)
  const float* yu = P.B + eo + n0;
  const uint32_t xo = (uint32_t)(4 * q * P.lda + j) * 4u;
  const uint32_t yo = (uint32_t)(4 * q * P.ldb + 4 * j) * 4u;
  // optimiser operands of the tile element this thread finishes (and of the bias column it finishes when by == 0);
  // pidx / bidx = parameter indices (the same for every expert), addressed at index + eo
  const int64_t toff = (int64_t)(k0 + (tid >> 4)) * P.ldc + n0 + 4 * (tid & 15);
  float* const dst = P.C + eg + toff;
  const int64_t pidx = ADAM ? (int64_t)(P.C - A.grad) + toff : 0;
  const int64_t bidx = ADAM ? (int64_t)(P.aux_out + n0 + (tid & 63) - A.grad) : 0;
  AdamPre4 pre;
  float aQ = 0.f, aPi = 0.f, bm = 0.f, bv = 0.f, bth = 0.f;
  bool faulted = false;
  AdamEarly early;
  if (ADAM) {
    early = given ? *given : adam_early(A, eo);
    if (S == 1) {                                             // (split: the workgroup that finishes the tile fetches them then)
      pre = adam_prefetch4(A, pidx + eo);
      if (by == 0 && tid < 64) { bm = A.m[bidx + eo]; bv = A.v[bidx + eo]; bth = A.theta[bidx + eo]; }
    }
  }
  f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
  f32x4 bsum = zero4();
  // half h of a 256-row chunk: this wave's 16-row pieces u = 2 h, 2 h + 1
  auto load_half = [&](const int mb, const int h, float (&a)[2][4], f32x4 (&b)[2][4]) {
#pragma unroll
    for (int u2 = 0; u2 < 2; ++u2) {
      const int mu = mb + (wv + 4 * (2 * h + u2)) * 16;       // (+ 4 q: in the lane offsets)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        a[u2][s] = ld_su(xu + (int64_t)(mu + s) * P.lda, xo);
        b[u2][s] = ld4_su(yu + (int64_t)(mu + s) * P.ldb, yo);
      }
    }
  };
  auto load_pair = [&](const int mb, const int h, const int u2, const int s, float (&a)[2][4], f32x4 (&b)[2][4]) {
    const int mu = mb + (wv + 4 * (2 * h + u2)) * 16;
    a[u2][s] = ld_su(xu + (int64_t)(mu + s) * P.lda, xo);
    b[u2][s] = ld4_su(yu + (int64_t)(mu + s) * P.ldb, yo);
  };
  auto mac_pair = [&](const int u2, const int s, const float (&a)[2][4], const f32x4 (&b)[2][4]) {
    bsum += b[u2][s];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = MFMA(a[u2][s], b[u2][s][e], acc[e]);
  };
  auto mac_half = [&](const float (&a)[2][4], const f32x4 (&b)[2][4]) {
#pragma unroll
    for (int u2 = 0; u2 < 2; ++u2)
#pragma unroll
      for (int s = 0; s < 4; ++s) mac_pair(u2, s, a, b);
  };
  float a0[2][4], a1[2][4];
  f32x4 b0[2][4], b1[2][4];
  const int step = 256;
  int mb = m_lo;
  if (!PIPE) {
    // one chunk (a batch of one rank): everything requested, then everything multiplied
    load_half(mb, 0, a0, b0);
    load_half(mb, 1, a1, b1);
    LOADS_FIRST();
    DW_STAMP(stamps, 1);
    mac_half(a0, b0);
    mac_half(a1, b1);
  } else {
    // several chunks (virtual ranks): while one half is multiplied the other is on its way -- the same registers and the
    // same order of the products as above, but the texture path and the matrix cores work at the same time.  The loads of
    // the other half go out ONE PAIR AT A TIME, each behind the matrix instructions of one pair of this half: the CU's
    // waves share one texture path, a wave that issues 16 loads in a row is held at every one of them until the path has
    // taken it, and its matrix instructions wait behind them -- texture time and matrix time add up instead of overlapping
    // (tools/dw_lab.hip: 3.5 k cycles per chunk = 1.5 k + 2.0 k, whatever the number of workgroups on the CU; two
    // workgroups on a CU 48 -> 33 us for the hidden tiles of 19 ranks).
    if (mb < m_hi) load_half(mb, 0, a0, b0);
    DW_STAMP(stamps, 1);
    for (; mb < m_hi; mb += step) {
      // (behind the last chunk the same rows once more, unused: a branch around those loads makes the number of loads in
      //  flight depend on the path, and the waits in front of the products become vmcnt(0))
      const int mbn = (mb + step < m_hi) ? mb + step : mb;
#pragma unroll
      for (int u2 = 0; u2 < 2; ++u2)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          load_pair(mb, 1, u2, s, a1, b1);
          __builtin_amdgcn_sched_barrier(0);
          mac_pair(u2, s, a0, b0);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
      for (int u2 = 0; u2 < 2; ++u2)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          load_pair(mbn, 0, u2, s, a0, b0);
          __builtin_amdgcn_sched_barrier(0);
          mac_pair(u2, s, a1, b1);
          __builtin_amdgcn_sched_barrier(0);
        }
    }
  }
  DW_STAMP(stamps, 2);
  if (ADAM) {
    // the matrix instructions are issued: the step counter has long arrived, the look-up of the step sizes hides behind
    // the reduction of the tile.  (Outside the loop on purpose: a pinned value redefined inside it becomes loop-carried,
    // and the copy in front of the loop waits for the load -- before the tile's operands are even requested.)
    adam_alphas_late(A, early, aQ, aPi, eo);
    faulted = adam_early_faulted(A, early);
  }
  f32x4 v; int orow, c4;
  hot_store(red, acc, wave, q, j, tid, v, orow, c4);
  float gb = 0.f;
  auto bias_reduce = [&]() {
    // column sums of B: 16 partials (4 waves x 4 lane groups) per column through LDS
    __syncthreads();
    *reinterpret_cast<f32x4*>(red + (wave * 4 + q) * 64 + 4 * j) = bsum;
    __syncthreads();
    if (tid < 64) {
#pragma unroll
      for (int r = 0; r < 16; ++r) gb += red[r * 64 + tid];
    }
  };
  const bool split = PIPE && S > 1;
  if (split) {
    if (by == 0) bias_reduce();
    if (!dw_split_combine(*sp, gt, S, seg, tid, v, gb, red)) return;
    if (ADAM) {
      pre = adam_prefetch4(A, pidx + eo);
      if (by == 0 && tid < 64) { bm = A.m[bidx + eo]; bv = A.v[bidx + eo]; bth = A.theta[bidx + eo]; }
    }
  }
  *reinterpret_cast<f32x4*>(dst) = v;
  if (ADAM && !faulted) {
    adam_apply4(A, (pidx < A.n_Q) ? -aQ : -aPi, pidx + eo, v, pre);
    // transposed copy of the updated tile for the row-local backward layers (mlp_rows.h): WT[n][k] = W[k][n]
    if (P.dot_out) {
      float* t = P.dot_out + eo + (int64_t)(n0 + 4 * (tid & 15)) * P.K + k0 + (tid >> 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) t[(int64_t)e * P.K] = pre.th[e];
    }
  }
  if (by == 0) {
    if (!split) bias_reduce();                                // (behind the tile's own stores: they are what the launch waits for)
    if (tid < 64) {
      P.aux_out[eg + n0 + tid] = gb;
      if (ADAM && !faulted) {
        const float th = adam_elem(A, (bidx < A.n_Q) ? -aQ : -aPi, gb, bm, bv, bth);
        A.m[bidx + eo] = bm; A.v[bidx + eo] = bv; A.theta[bidx + eo] = th;
      }
    }
  }
}


// ---- 64 x 64 tiles of the hidden matrices for batches of many chunks (round 6; option dw64).
// The 16 x 64 tile above reads, per 4 batch rows and wave, 64 B of X four times (half cache lines) and 256 B of dY four times for
// 4 matrix instructions; its four waves split the ROWS, so nothing is shared: 12 cache lines through the CU's one texture
// path per 128 matrix cycles and wave -- the path is busy 75 % of the time with ONE workgroup on the CU (measured: 5.4 k
// cycles per chunk of 256 rows against 2.0 k of matrix time, 34 % SQ_VALU_MFMA_BUSY), and a second workgroup per CU makes the launch slower, not
// faster (option dw_split, 51 -> 55 us at 19 ranks).  Here a workgroup owns 64 x 64 of dW = X^T dY for a SEGMENT of the batch
// rows (DwSplit: S = up to 8 segments per tile, 64 tiles per launch x S = 512 workgroups; the ticket reduction adds the
// partial tiles in segment order): chunks of 32 rows of X[., 64] and dY[., 64] travel global -> registers -> LDS once per
// workgroup (two chunks in flight, two LDS stages, one barrier per chunk), wave (wr, wc) multiplies the 32 x 32 quadrant
// out of LDS: per 4 rows one ds_read_b64 of X and one of dY feed 4 x v_mfma_f32_16x16x4 (rows k' = 2 j + r, columns
// n = 2 j + c of the quadrant: a lane's operands of both 16-wide halves are neighbours).  Texture traffic per matrix
// instruction falls by 2.5 (full lines, each fetched once per workgroup), the LDS serves 32 B / clk per workgroup.
// The finished tile goes through LDS into the layout of four 16 x 64 strips, whose epilogue (store, Adam, transposed copy,
// bias column) is the one above.
#define DW64_LD 96                                           // LDS row stride (floats): 96 mod 64 = 32 -> the two 4-row halves of a
                                                             // ds_read_b64 lane group fall on different banks
#define DW64_ROWS 32                                         // batch rows per chunk
#define DW64_STAGE (2 * DW64_ROWS * DW64_LD)                 // floats per stage: X chunk | dY chunk
#ifndef DW64_PAD
#define DW64_PAD 0
#endif
#define DW64_LDS (2 * DW64_STAGE + DW64_PAD)                 // 48 KB (+ lab padding: workgroups per CU)
#define DW64_TLD 68                                          // row stride of the finished tile in LDS
#define DW_PART64 (64 * 64 + 64)                             // floats of a partial 64 x 64 tile + its 64 column sums
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool ADAM>
__device__ DW_INLINE void dw_hot_tile64(const GemmHot& P, const AdamFuse& A, const int t, float* lds, const int64_t eo,
                                        const int64_t eg, const AdamEarly* given, float* pbuf64, int32_t* cnt,
                                        const int gt, const DwSeg sg) {
  const int S = sg.S, seg = sg.seg;
  int m_lo = 0, m_hi = P.M;
  if (S > 1) {
    // segments in units of 64 rows (two chunks), not of 256 like the 16 x 64 tiles': 19 ranks are 76 units = 9 or 10 per
    // segment (in units of 256 rows: 2 or 3 -- the segments of 3 took 1.5 x as long as the others and the launch waited)
    const int U = P.M >> 6;
    m_lo = ((seg * U) / S) << 6;
    m_hi = (((seg + 1) * U) / S) << 6;
  }
  const int by = t >> 2, bx = t & 3;                          // 4 x 4 tiles of a 256 x 256 matrix
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, q = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  const int k0 = by * 64, n0 = bx * 64;
  AdamEarly early;
  if (ADAM) early = given ? *given : adam_early(A, eo);
  // staging: thread tid carries rows (tid >> 4) and 16 + (tid >> 4) of a chunk, columns 4 (tid & 15) .. + 3
  // (uniform base + 32-bit lane offset, ld4_su: the row part of an address is scalar work and no vector register is written
  //  on the way to a load -- with per-lane 64-bit pointers hipcc computed them INTO the destination registers and waited
  //  vmcnt(0) in front of every fetch for the loads it thought might still be writing them)
  const float* xu = P.A + eo + k0;
  const float* yu = P.B + eo + n0;
  const uint32_t xo = (uint32_t)((tid >> 4) * P.lda + 4 * (tid & 15)) * 4u;
  const uint32_t yo = (uint32_t)((tid >> 4) * P.ldb + 4 * (tid & 15)) * 4u;
  const int so = (tid >> 4) * DW64_LD + 4 * (tid & 15);
  f32x4 sx[2][2], sy[2][2];                                   // [chunk in flight][half]
  auto fetch = [&](const int m, f32x4 (&x)[2], f32x4 (&y)[2]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#ifdef DW64_NO_LOADS        // (lab: ... without its loads: the same rows again and again)
      x[h] = ld4_su(xu + (int64_t)(m_lo + 16 * h) * P.lda, xo);
      y[h] = ld4_su(yu + (int64_t)(m_lo + 16 * h) * P.ldb, yo);
#else
      x[h] = ld4_su(xu + (int64_t)(m + 16 * h) * P.lda, xo);
      y[h] = ld4_su(yu + (int64_t)(m + 16 * h) * P.ldb, yo);
#endif
    }
  };
  auto stage = [&](float* st, const f32x4 (&x)[2], const f32x4 (&y)[2]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      *reinterpret_cast<f32x4*>(st + so + 16 * h * DW64_LD) = x[h];
      *reinterpret_cast<f32x4*>(st + DW64_ROWS * DW64_LD + so + 16 * h * DW64_LD) = y[h];
    }
  };
  f32x4 acc[2][2] = {{zero4(), zero4()}, {zero4(), zero4()}};
  float bs = 0.f;                                             // column sums of dY (tiles of the first tile row): column tid & 63,
  const bool bias = by == 0;                                  // rows 8 (tid >> 6) .. + 7 of every chunk
  const int ao = q * DW64_LD + 32 * wr + 2 * j, bo = DW64_ROWS * DW64_LD + q * DW64_LD + 32 * wc + 2 * j;
  auto chunk_mac = [&](const float* st) {
#pragma unroll
    for (int s4 = 0; s4 < DW64_ROWS / 4; ++s4) {
      const f32x2 a2 = *reinterpret_cast<const f32x2*>(st + ao + 4 * s4 * DW64_LD);
      const f32x2 b2 = *reinterpret_cast<const f32x2*>(st + bo + 4 * s4 * DW64_LD);
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#ifdef DW64_NO_MFMA         // (lab: what the tile costs without its matrix instructions)
        acc[r][c][0] += a2[r] * b2[c];
#else
        acc[r][c] = MFMA(a2[r], b2[c], acc[r][c]);
#endif
      }
    }
    if (bias) {
      const float* yb = st + DW64_ROWS * DW64_LD + 8 * (tid >> 6) * DW64_LD + (tid & 63);
#pragma unroll
      for (int r = 0; r < 8; ++r) bs += yb[r * DW64_LD];
    }
  };
  // pipeline: chunk c is multiplied out of stage c & 1 while chunk c + 1 waits in registers for its turn to be written to
  // the other stage (behind the products: the barrier at the end of the previous half has freed it) and chunk c + 2 is on
  // its way from memory.  Branch-free (a segment is a multiple of 256 rows: an even number of chunks; behind the last
  // chunks the last one is fetched and staged once more, unused): a branch around a fetch makes the number of loads in
  // flight depend on the path, and the waits in front of the stores to LDS become vmcnt(0) (dw_hot_tile).
  const int nch = (m_hi - m_lo) / DW64_ROWS;
  const int last = m_lo + (nch - 1) * DW64_ROWS;
  fetch(m_lo, sx[0], sy[0]);
  fetch(min(m_lo + DW64_ROWS, last), sx[1], sy[1]);
  stage(lds, sx[0], sy[0]);
  __syncthreads();
  for (int m = m_lo; m < m_hi; m += 2 * DW64_ROWS) {
    fetch(min(m + 2 * DW64_ROWS, last), sx[0], sy[0]);         // registers: [1] = chunk c + 1, [0] <- chunk c + 2
    __builtin_amdgcn_sched_barrier(0);
    chunk_mac(lds);
    __builtin_amdgcn_sched_barrier(0);
    stage(lds + DW64_STAGE, sx[1], sy[1]);
    __syncthreads();
    fetch(min(m + 3 * DW64_ROWS, last), sx[1], sy[1]);         // registers: [0] = chunk c + 2, [1] <- chunk c + 3
    __builtin_amdgcn_sched_barrier(0);
    chunk_mac(lds + DW64_STAGE);
    __builtin_amdgcn_sched_barrier(0);
    stage(lds, sx[0], sy[0]);
    __syncthreads();
  }
  float aQ = 0.f, aPi = 0.f;
  bool faulted = false;
  if (ADAM) {
    adam_alphas_late(A, early, aQ, aPi, eo);
    faulted = adam_early_faulted(A, early);
  }
  // ---- the quadrants -> LDS tile T[64][DW64_TLD]; thread tid then owns row 16 s + (tid >> 4), columns 4 (tid & 15) .. of
  // strip s = 0 .. 3 (the ownership of the 16 x 64 tile's epilogue)
  float* T = lds;
  float* bred = lds + 64 * DW64_TLD;                          // [4][64] partial column sums
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x2 v2 = {acc[r][0][i], acc[r][1][i]};
      *reinterpret_cast<f32x2*>(T + (32 * wr + 2 * (4 * q + i) + r) * DW64_TLD + 32 * wc + 2 * j) = v2;
    }
  if (bias) bred[(tid >> 6) * 64 + (tid & 63)] = bs;
  __syncthreads();
  f32x4 v[4];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) v[s4] = *reinterpret_cast<const f32x4*>(T + (16 * s4 + (tid >> 4)) * DW64_TLD + 4 * (tid & 15));
  float gb = 0.f;
  if (bias && tid < 64) gb = (bred[tid] + bred[64 + tid]) + (bred[128 + tid] + bred[192 + tid]);
  if (S > 1) {
    // split reduction (dw_split_combine for a 64 x 64 partial tile)
    __syncthreads();                                          // (T is read: `flag` below reuses the LDS)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(pbuf64 + (int64_t)gt * S * DW_PART64, 0, 0x7fffffff,
                                                                        0x00020000);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(dw_u32x4, v[s4]), rs, (seg * DW_PART64 + 1024 * s4 + 4 * tid) * 4, 0, DW_SC1);
    if (tid < 64) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, gb), rs, (seg * DW_PART64 + 4096 + tid) * 4, 0, DW_SC1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* flag = reinterpret_cast<int*>(lds);
    if (tid == 0) {
      const int old = __hip_atomic_fetch_add(cnt + gt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = old == S - 1;
      if (last) __hip_atomic_store(cnt + gt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *flag = last;
    }
    __syncthreads();
    if (*flag == 0) return;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) v[s4] = zero4();
    gb = 0.f;
    for (int k = 0; k < S; ++k) {                             // (segment order: whoever is last adds the same way)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
        v[s4] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (k * DW_PART64 + 1024 * s4 + 4 * tid) * 4, 0, DW_SC1));
      if (tid < 64) gb += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (k * DW_PART64 + 4096 + tid) * 4, 0, DW_SC1));
    }
  }
  // ---- epilogue of the four strips: gradient store, optimiser, transposed copy (dw_hot_tile)
  AdamPre4 pre[4];
  int64_t pidx[4];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
    const int64_t toff = (int64_t)(k0 + 16 * s4 + (tid >> 4)) * P.ldc + n0 + 4 * (tid & 15);
    pidx[s4] = ADAM ? (int64_t)(P.C - A.grad) + toff : 0;
    if (ADAM) pre[s4] = adam_prefetch4(A, pidx[s4] + eo);
    *reinterpret_cast<f32x4*>(P.C + eg + toff) = v[s4];
  }
  const int64_t bidx = ADAM ? (int64_t)(P.aux_out + n0 + (tid & 63) - A.grad) : 0;
  float bm = 0.f, bv = 0.f, bth = 0.f;
  if (ADAM && bias && tid < 64) { bm = A.m[bidx + eo]; bv = A.v[bidx + eo]; bth = A.theta[bidx + eo]; }
  if (ADAM && !faulted) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      adam_apply4(A, (pidx[s4] < A.n_Q) ? -aQ : -aPi, pidx[s4] + eo, v[s4], pre[s4]);
      if (P.dot_out) {                                        // WT[n][k] = W[k][n] (mlp_rows.h)
        float* tp = P.dot_out + eo + (int64_t)(n0 + 4 * (tid & 15)) * P.K + k0 + 16 * s4 + (tid >> 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) tp[(int64_t)e * P.K] = pre[s4].th[e];
      }
    }
  }
  if (bias && tid < 64) {
    P.aux_out[eg + n0 + tid] = gb;
    if (ADAM && !faulted) {
      const float th = adam_elem(A, (bidx < A.n_Q) ? -aQ : -aPi, gb, bm, bv, bth);
      A.m[bidx + eo] = bm; A.v[bidx + eo] = bv; A.theta[bidx + eo] = th;
    }
  }
}

// Small weight gradients (layer-0 segments, output layers) on a compact tile list + the loss finalisation.
//   dW[w,N] = (X[M,w] / div)^T . dY[M,N];  db[N] = colsum(dY)         M % 256 == 0, X and dY plain row matrices
struct DwSmall {
  const float* x; const float* dY; float* dW; float* db;
  int32_t ldx, lddy, w, N;
  float div;
};
#define MAX_DW_SMALL 12
struct DwSmallArgs {
  DwSmall p[MAX_DW_SMALL]; int32_t nprob, M, slots; LossFin fin;   // `slots` block ids per problem
};

// One 16 x 64 tile of a small problem.  YV: N % 4 == 0 (16-byte dY fragments); otherwise N == 1 (the critic's output
// layer): one dY column, one accumulator, a quarter of the MFMAs.  Uniform conditions are hoisted out of the unrolled
// load / MFMA loops (a branch per fragment made this body slower than a full 256-deep hidden-layer tile).
// SC: the X elements are divided by P.div (the action segment with max_u != 1; a form of its own: with the division behind
// a select in the common form every tile would pay for it -- 16 divisions per wave and chunk, 0.3 us per launch)
template <bool ADAM, bool YV, bool PIPE, bool SC>
__device__ DW_INLINE void dw_small_tile(const DwSmall& P, const int M, const AdamFuse& A, const int t, float* red,
                                     const int64_t eo, const int64_t eg, DwStamp* stamps = nullptr,
                                     const AdamEarly* given = nullptr, const DwSplit* sp = nullptr, const int gt = 0,
                                     const DwSeg sg = DwSeg{0, 1}) {
  const int S = PIPE ? sg.S : 1, seg = PIPE ? sg.seg : 0;     // (split reduction: dw_split_combine)
  int m_lo = 0, m_hi = M;
  if (S > 1) dw_split_range(M, S, seg, m_lo, m_hi);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int nx = (P.N + 63) >> 6;
  int by, bx;
  tile_divmod(t, nx, by, bx);
  const int j = lane & 15, q = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int k0 = by * 16, n0 = bx * 64;
  const int krow = k0 + j, col = n0 + 4 * j;
  const bool k_ok = krow < P.w;
  const int colc = YV ? min(col, P.N - 4) : 0;
  const float* xu = P.x + eo;                                 // (uniform parts; lane parts below, ld_su)
  const float* yu = P.dY + eo;
  const uint32_t xo = (uint32_t)(4 * q * P.ldx + min(krow, P.w - 1)) * 4u;
  const uint32_t yo = (uint32_t)(4 * q * P.lddy + colc) * 4u;
  // optimiser operands of what this thread finishes, fetched with the first batch of loads
  const int grow = k0 + (tid >> 4), gcol = n0 + 4 * (tid & 15);
  const bool own = grow < P.w && gcol < P.N;
  float* const dst = P.dW + eg + (int64_t)(own ? grow : 0) * P.N + (own ? gcol : 0);
  const int64_t pidx = ADAM ? (int64_t)(dst - eg - A.grad) : 0;      // parameter index, addressed at index + eo
  const bool own_b = P.db && by == 0 && tid < 64 && n0 + tid < P.N;
  const int64_t bidx = (ADAM && own_b) ? (int64_t)(P.db + n0 + tid - A.grad) : 0;
  float aQ = 0.f, aPi = 0.f, bm = 0.f, bv = 0.f, bth = 0.f;
  AdamPre4 pre;
  pre.m = zero4(); pre.v = zero4(); pre.th = zero4();
  bool faulted = false;
  AdamEarly early;
  if (ADAM) {
    early = given ? *given : adam_early(A, eo);
  }
  auto fetch_params = [&]() {
    if (YV) {
      pre = adam_prefetch4(A, (own ? pidx : 0) + eo);
    } else if (own) {                                       // N == 1: one element per owning thread
      pre.m[0] = A.m[pidx + eo]; pre.v[0] = A.v[pidx + eo]; pre.th[0] = A.theta[pidx + eo];
    }
    if (own_b) { bm = A.m[bidx + eo]; bv = A.v[bidx + eo]; bth = A.theta[bidx + eo]; }
  };
  if (ADAM && S == 1) fetch_params();                         // (split: the workgroup that finishes the tile fetches them then)
  f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
  f32x4 bsum = zero4();
  auto load_half = [&](const int mb, const int h, float (&a)[2][4], f32x4 (&b)[2][4]) {
#pragma unroll
    for (int u2 = 0; u2 < 2; ++u2) {
      const int mu = mb + (wv + 4 * (2 * h + u2)) * 16;       // (+ 4 q: in the lane offsets)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        a[u2][s] = ld_su(xu + (int64_t)(mu + s) * P.ldx, xo);
        if (YV) {
          b[u2][s] = ld4_su(yu + (int64_t)(mu + s) * P.lddy, yo);
        } else {
          b[u2][s] = zero4();
          b[u2][s][0] = ld_su(yu + (int64_t)(mu + s) * P.lddy, yo);
        }
      }
    }
  };
  auto load_pair = [&](const int mb, const int h, const int u2, const int s, float (&a)[2][4], f32x4 (&b)[2][4]) {
    const int mu = mb + (wv + 4 * (2 * h + u2)) * 16;
    a[u2][s] = ld_su(xu + (int64_t)(mu + s) * P.ldx, xo);
    if (YV) {
      b[u2][s] = ld4_su(yu + (int64_t)(mu + s) * P.lddy, yo);
    } else {
      b[u2][s] = zero4();
      b[u2][s][0] = ld_su(yu + (int64_t)(mu + s) * P.lddy, yo);
    }
  };
  auto mac_pair = [&](const int u2, const int s, const float (&a)[2][4], const f32x4 (&b)[2][4]) {
    const float as = SC ? fdiv(a[u2][s], P.div) : a[u2][s];
    const float av = k_ok ? as : 0.f;
    bsum += b[u2][s];
    if (YV) {
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = MFMA(av, b[u2][s][e], acc[e]);
    } else {
      acc[0] = MFMA(av, b[u2][s][0], acc[0]);
    }
  };
  auto mac_half = [&](const float (&a)[2][4], const f32x4 (&b)[2][4]) {
#pragma unroll
    for (int u2 = 0; u2 < 2; ++u2)
#pragma unroll
      for (int s = 0; s < 4; ++s) mac_pair(u2, s, a, b);
  };
  float a0[2][4], a1[2][4];
  f32x4 b0[2][4], b1[2][4];
  const int step = 256;
  int mb = m_lo;
  if (!PIPE) {
    load_half(mb, 0, a0, b0);
    load_half(mb, 1, a1, b1);
    LOADS_FIRST();
    DW_STAMP(stamps, 1);
    mac_half(a0, b0);
    mac_half(a1, b1);
  } else {                                                    // (several chunks: dw_hot_tile)
    if (mb < m_hi) load_half(mb, 0, a0, b0);
    DW_STAMP(stamps, 1);
    for (; mb < m_hi; mb += step) {
      const int mbn = (mb + step < m_hi) ? mb + step : mb;    // (unconditional loads: dw_hot_tile)
#pragma unroll
      for (int u2 = 0; u2 < 2; ++u2)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          load_pair(mb, 1, u2, s, a1, b1);
          __builtin_amdgcn_sched_barrier(0);
          mac_pair(u2, s, a0, b0);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
      for (int u2 = 0; u2 < 2; ++u2)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          load_pair(mbn, 0, u2, s, a0, b0);
          __builtin_amdgcn_sched_barrier(0);
          mac_pair(u2, s, a1, b1);
          __builtin_amdgcn_sched_barrier(0);
        }
    }
  }
  DW_STAMP(stamps, 2);
  if (ADAM) {
    adam_alphas_late(A, early, aQ, aPi, eo);
    faulted = adam_early_faulted(A, early);
  }
  int orow, c4;
  f32x4 v = reduce_tile(red, acc, wave, q, j, tid, orow, c4);
  float gb = 0.f;
  auto bias_reduce = [&]() {
    __syncthreads();
    *reinterpret_cast<f32x4*>(red + (wave * 4 + q) * 64 + 4 * j) = bsum;
    __syncthreads();
    if (tid < 64) {
#pragma unroll
      for (int r = 0; r < 16; ++r) gb += red[r * 64 + tid];
    }
  };
  const bool split = PIPE && S > 1;
  if (split) {
    if (P.db && by == 0) bias_reduce();
    if (!dw_split_combine(*sp, gt, S, seg, tid, v, gb, red)) return;
    if (ADAM) fetch_params();
  }
  if (own) {
    const float na = (pidx < A.n_Q) ? -aQ : -aPi;
    if (YV) {
      *reinterpret_cast<f32x4*>(dst) = v;
      if (ADAM && !faulted) adam_apply4(A, na, pidx + eo, v, pre);
    } else {
      dst[0] = v[0];                                        // N == 1 (gcol == 0)
      if (ADAM && !faulted) {
        float m = pre.m[0], vv = pre.v[0];
        const float th = adam_elem(A, na, v[0], m, vv, pre.th[0]);
        A.m[pidx + eo] = m; A.v[pidx + eo] = vv; A.theta[pidx + eo] = th;
      }
    }
  }
  if (P.db && by == 0) {
    if (!split) bias_reduce();
    if (own_b) {
      P.db[eg + n0 + tid] = gb;
      if (ADAM && !faulted) {
        const float th = adam_elem(A, (bidx < A.n_Q) ? -aQ : -aPi, gb, bm, bv, bth);
        A.m[bidx + eo] = bm; A.v[bidx + eo] = bv; A.theta[bidx + eo] = th;
      }
    }
  }
}

// losses (ddpg.py:439-441) from the per-row terms, summed in a fixed order
__device__ inline void dw_loss_fin(const LossFin& F, float* red, const int64_t eo, const int64_t eg) {
  const int tid = threadIdx.x;
  {
    float lq = 0.f, lp = 0.f, ll = 0.f;
    for (int m = tid; m < F.Bl; m += 256) {                  // (the first rank's rows; the others: loss_fin_ranks)
      lq += F.rows[eo + m];
      lp += F.rows[eo + F.B + m];
      ll += F.rows[eo + 2 * F.B + m];
    }
    red[tid] = lq; red[256 + tid] = lp; red[512 + tid] = ll;
    __syncthreads();
    for (int h = 128; h >= 1; h >>= 1) {
      if (tid < h) {
        red[tid] += red[tid + h];
        red[256 + tid] += red[256 + tid + h];
        red[512 + tid] += red[512 + tid + h];
      }
      __syncthreads();
    }
    if (tid == 0) {
      const float invB = 1.0f / (float)F.Bl;
      if (F.step_ctr) *ex_i64(F.step_ctr, eo) += 1;
      loss_fin_flag(F, eo, eg);
      F.out[eo + 0] = red[0] * invB;
      F.out[eo + 1] = -red[256] * invB + F.action_l2 * red[512] / (float)(F.Bl * F.U);
    }
    if (F.Bl < F.B) loss_fin_ranks(F, F.rows + eo, F.out + eo);
  }
}

// Every weight/bias gradient of both networks + the loss finalisation in ONE launch: blocks [0, n_hot) run the
// hidden-layer tiles, the rest the small-problem tile list (the two lists are independent, so splitting them over two
// launches only bought a second ~4.5 us dependent stage).
// (batched experts: blockIdx.y = expert, as in dw_adam_her_kernel below)
// XCD-aware block placement (xcd = 1; grid.x = 8 * (r_her + r_hot + r_small)).  Workgroups are dealt round-robin over
// the 8 XCDs in block-id order, so x = blockIdx.x & 7 IS the XCD and r = blockIdx.x >> 3 a row of 8 blocks, one per XCD:
//   r <  r_her                : gather block r * 8 + x
//   r <  r_her + r_hot        : hidden-layer tile; XCD x owns `units`-th part u = x % units of matrix x / units -- its
//                               r_hot = 64 / units tiles are the row strips [u, u + 1) * 16 / units x all 4 column panels,
//                               i.e. ONE XCD's L2 fetches that part of X and the matrix's dY once, instead of every XCD
//                               fetching a quarter panel of every matrix (1.25 MB per XCD at Arm4 -> 0.38 MB)
//   otherwise                 : small problem p = x + 8 * (j / slots), tile j % slots (j = r - r_her - r_hot): the tiles
//                               of one small problem share an XCD; the block behind the last problem finalises the losses
// Speed only -- nothing depends on the placement for correctness.
// The routing scalars travel as the kernel's LEADING arguments: with -mllvm -amdgpu-kernarg-preload-count they arrive in
// scalar registers with the wave (kernarg preloading, gfx940+), so a block knows its role without a single memory access.
// That matters because reads of the kernarg segment are not cached on this machine: EVERY dependent s_load of an argument
// costs a full ~3 k-cycle (1.4 us) round trip (tools/dw_stamps.py: a block that finds it has nothing to do used to need
// 3 k cycles to find out; a hidden tile issued its operand loads 6 k cycles after its start, a small tile 10-12 k).
struct DwMap { int32_t r_her, r_hot, units; };                // units == 0: plain block order
struct DwAllArgs {
  DwHotArgs hot; DwSmallArgs small; int32_t n_hot; unsigned long long* stamps; DwSplit split;
  float* pbuf64;                  // non-NULL: the hidden matrices as 64 x 64 tiles (dw_hot_tile64; tiles_per = 16), their partial tiles here
};
// lab (a build with -DDW_STAMPS, tools/build_variant.py, + option "lab_dw_stamps"): 8 x 64-bit words per block of expert 0 -- [0] entry, [1] operand loads issued /
// gather: tables in, [2] MFMA loop over / gather: rows in LDS, [3] exit, [4] kind (0 gather, 1 hidden tile, 2 small),
// [5] s_memrealtime at entry (100 MHz, device-wide)
__device__ inline unsigned long long* dw_stamp_base(const DwAllArgs& a) {
  return (a.stamps && blockIdx.y == 0) ? a.stamps + (size_t)blockIdx.x * 8 : nullptr;
}
// (split reduction, DwSplit: `units_s` = units | S_hot << 8 | S_small << 16; the S segments of a tile are S consecutive rows of
//  8 blocks / S consecutive block ids -- dispatched together)
struct DwRole { int kind, pi, idx, seg, S; };
__device__ __forceinline__ DwRole dw_role(const int n_hot, const int tiles_per, const int hot_nprob, const int slots,
                                          const int n_her, const int r_her, const int r_hot, const int units_s) {
  DwRole R;
  R.kind = -1; R.pi = 0; R.idx = 0; R.seg = 0; R.S = 1;
  const int units = units_s & 0xff;
  const int S_hot = ((units_s >> 8) & 0xff) ? ((units_s >> 8) & 0xff) : 1, S_small = ((units_s >> 16) & 0xff) ? ((units_s >> 16) & 0xff) : 1;
  const int b = (int)blockIdx.x;
  if (units == 0) {
    int bid = b - n_her;
    if (bid < 0) { R.kind = 0; R.idx = b; return R; }
    if (bid < n_hot * S_hot) {
      R.S = S_hot; R.seg = bid % S_hot; bid /= S_hot;
      R.kind = 1; R.pi = bid / tiles_per; R.idx = bid - R.pi * tiles_per;
    } else {
      bid -= n_hot * S_hot;
      R.S = S_small; R.seg = bid % S_small;
      R.kind = 2; R.idx = bid / S_small;
    }
    return R;
  }
  const int x = b & 7;
  int r = b >> 3;
  if (r_her < 0) {
    // r_her < 0: the -r_her rows of gather blocks come LAST instead of first (batches of several virtual ranks: the
    // hidden tiles, each a long reduction over V x 256 rows, are then dispatched together at the start of the launch
    // and walk the rows in step -- the tiles of an XCD share X strips and dY panels through its L2 only while they do)
    const int rows = (int)gridDim.x >> 3, gr = -r_her;
    if (r >= rows - gr) {
      const int i = (r - (rows - gr)) * 8 + x;
      if (i < n_her) { R.kind = 0; R.idx = i; }
      return R;
    }
    r += gr;                                                 // (the rest of the map as if the gather rows came first)
  }
  const int r_her_ = r_her < 0 ? -r_her : r_her;
  if (r < r_her_) {
    if (r * 8 + x < n_her) { R.kind = 0; R.idx = r * 8 + x; }
    return R;
  }
  r -= r_her_;
  if (r < r_hot * S_hot) {
    R.S = S_hot; R.seg = r % S_hot; r /= S_hot;
    const int pi = x / units, u = x - pi * units;
    if (pi < hot_nprob) { R.kind = 1; R.pi = pi; R.idx = u * r_hot + r; }
  } else {
    int j = r - r_hot * S_hot;
    R.S = S_small; R.seg = j % S_small; j /= S_small;
    if ((units_s >> 24) & 1) {
      // balanced (several virtual ranks): every small problem is two HALF problems, and the halves -- the host has sorted the
      // problems by their number of tiles -- are dealt to the XCDs back and forth (0..7, 7..0, ..).  A small tile reads as
      // many lines per batch row as a hidden tile; with whole problems on XCDs p % 8 three XCDs carried 12-16 of them beside
      // their 32 hidden tiles and three none, and the launch ended with the hidden tiles of the loaded ones (19 ranks: 101 k
      // cycles against 74 k).  R.idx = half problem * hs + slot; dw_tile_role turns it into a tile of the problem
      const int hs = (slots + 1) >> 1;
      const int round = j / hs, t = j - round * hs;
      const int vp = round * 8 + ((round & 1) ? 7 - x : x);
      R.kind = 3; R.idx = vp * hs + t;
      return R;
    }
    const int p = x + 8 * (j / slots);
    R.kind = 2; R.idx = p * slots + j % slots;
  }
  return R;
}
// One batch of argument loads per block: a by-value copy of everything the role needs, taken through an empty asm that
// wants every field in a scalar register AT THIS POINT -- the compiler then issues all the s_loads together and waits
// once, instead of fetching field by field, branch by branch (one uncached round trip each).
// (Input-only operands: the values must be in scalar registers at this point, but they are not redefined -- a pointer that
//  came OUT of an asm would have lost its provenance, and the compiler would address it with FLAT instructions, which cost
//  an s_waitcnt vmcnt(0) lgkmcnt(0) whenever their results are needed and cannot be counted past.)
__device__ __forceinline__ void pin_hot(const GemmHot& P) {
  asm volatile("" :: "s"(P.A), "s"(P.B), "s"(P.C), "s"(P.aux_out), "s"(P.dot_out), "s"(P.lda), "s"(P.ldb), "s"(P.ldc),
               "s"(P.M), "s"(P.N), "s"(P.K));
}
__device__ __forceinline__ void pin_small(const DwSmall& P) {
  asm volatile("" :: "s"(P.x), "s"(P.dY), "s"(P.dW), "s"(P.db), "s"(P.ldx), "s"(P.lddy), "s"(P.w), "s"(P.N), "s"(P.div));
}
__device__ __forceinline__ void pin_adam(const AdamFuse& A) {
  asm volatile("" :: "s"(A.theta), "s"(A.m), "s"(A.v), "s"(A.grad), "s"(A.n_Q), "s"(A.alpha_tab), "s"(A.step_ctr),
               "s"(A.tab_base), "s"(A.tab_len), "s"(A.a_Q), "s"(A.a_pi), "s"(A.b1), "s"(A.omb1), "s"(A.b2), "s"(A.omb2),
               "s"(A.eps), "s"(A.fault));
}

// (the leading scalars: preloaded into SGPRs, see DwMap)
#define DW_ROUTE_PARAMS const int tiles_per, const int hot_nprob, const int slots, const int small_nprob, const int n_her, \
                        const int r_her, const int r_hot, const int units   /* units | S_hot << 8 | S_small << 16: dw_role */
// the tile work of a block whose role is known: ONE batch of argument loads, then the tile
template <bool ADAM, bool PIPE, bool T64 = false>
__device__ __forceinline__ void dw_tile_role(const DwRole& R, const DwAllArgs& args, const AdamFuse& A_, const int slots,
                                             const int small_nprob, float* red, const int64_t eo, int64_t grad_stride,
                                             DwStamp* sp, const AdamEarly* early) {
  AdamFuse A = A_;
  if (R.kind == 1) {
    GemmHot P = args.hot.p[R.pi];
    pin_hot(P);
    if (ADAM) pin_adam(A);
    asm volatile("" :: "s"(grad_stride));
    DW_STAMP(sp, 3);
    if constexpr (T64) {
      {
        dw_hot_tile64<ADAM>(P, A, R.idx, red, eo, (int64_t)blockIdx.y * grad_stride, early, args.pbuf64, args.split.cnt,
                            R.pi * args.hot.tiles_per + R.idx, DwSeg{R.seg, R.S});
        return;
      }
    }
    dw_hot_tile<ADAM, PIPE>(P, A, R.idx, red, eo, (int64_t)blockIdx.y * grad_stride, sp, early, &args.split,
                                R.pi * args.hot.tiles_per + R.idx, DwSeg{R.seg, R.S});
    return;
  }
  int pi = R.idx / slots, t = R.idx - pi * slots, half = -1;
  if (R.kind == 3) {                                          // (dw_role: half problem * hs + slot)
    const int hs = (slots + 1) >> 1, vp = R.idx / hs;
    t = R.idx - vp * hs; pi = vp >> 1; half = vp & 1;
    if (pi == small_nprob && half) return;
  }
  if (pi >= small_nprob) {
    // the block right behind the last problem finalises the losses (fin.rows != NULL); any other id beyond exits
    if (pi == small_nprob && t == 0 && R.seg == 0) dw_loss_fin(args.small.fin, red, eo, (int64_t)blockIdx.y * grad_stride);
    return;
  }
#ifdef DW_NO_SMALL            // (lab: the launch without its small tiles)
  return;
#endif
  DwSmall P = args.small.p[pi];
  int M = args.small.M;
  pin_small(P);
  if (ADAM) pin_adam(A);
  asm volatile("" :: "s"(grad_stride), "s"(M));
  DW_STAMP(sp, 3);
  int gidx = R.idx;
  if (half >= 0) {
    // half 0 / 1 of the problem's tiles in COLUMN-panel-major order: the halves of a layer-0 problem split the panels of dY,
    // those of an output layer the strips of X (neighbouring strips share cache lines) -- neither half reads what the other reads
    const int nby = (P.w + 15) >> 4, nx = (P.N + 63) >> 6, T = nby * nx, hT = (T + 1) >> 1;
    const int tb = half * hT + t;
    if (t >= hT || tb >= T) return;
    const int bx = tb / nby, by = tb - bx * nby;
    t = by * nx + bx;
    gidx = pi * slots + t;
  }
  if (t >= ((P.w + 15) >> 4) * ((P.N + 63) >> 6)) return;
  const int64_t eg = (int64_t)blockIdx.y * grad_stride;
  const int gt = args.n_hot + gidx;
  if ((P.N & 3) != 0) dw_small_tile<ADAM, false, PIPE, false>(P, M, A, t, red, eo, eg, sp, early, &args.split, gt, DwSeg{R.seg, R.S});
  else if (P.div != 1.0f) dw_small_tile<ADAM, true, PIPE, true>(P, M, A, t, red, eo, eg, sp, early, &args.split, gt, DwSeg{R.seg, R.S});
  else dw_small_tile<ADAM, true, PIPE, false>(P, M, A, t, red, eo, eg, sp, early, &args.split, gt, DwSeg{R.seg, R.S});
}

// (the pipelined form with 4 waves per SIMD: 113 registers instead of 113 + 20 accumulation registers, and a fourth workgroup
//  per CU -- the small tiles and the gather blocks of a several-rank launch, whose lives are round trips to memory, wait less
//  for a place: 17.0 -> 16.7 ms per cycle at 19 ranks; the one-chunk kernel keeps the compiler's choice, 3)
#define DW_WAVES_ATTR __attribute__((amdgpu_waves_per_eu((PIPE && !T64) ? 4 : 1, (PIPE && !T64) ? 4 : 8)))
template <bool PIPE, bool T64 = false>
__global__ __launch_bounds__(256) DW_WAVES_ATTR void dw_all_kernel(DW_ROUTE_PARAMS, int64_t ex_stride, DwAllArgs args,
                                                     int64_t grad_stride) {
  __shared__ __attribute__((aligned(16))) float red[T64 ? DW64_LDS : 4 * 16 * 64];
  AdamFuse none;
  none.fault = nullptr;
  const int64_t eo = (int64_t)blockIdx.y * ex_stride;
  const DwRole R = dw_role(hot_nprob * tiles_per, tiles_per, hot_nprob, slots, 0, r_her, r_hot, units);
  if (R.kind > 0) dw_tile_role<false, PIPE, T64>(R, args, none, slots, small_nprob, red, eo, grad_stride, nullptr, nullptr);
}

// The tail of a whole single-rank update in one launch (curious_ddpg_update): every weight/bias gradient with Adam
// applied in the tile epilogue, the loss finalisation, and -- in the first n_her blocks -- the HER gather of the NEXT
// update's batch (it depends on nothing this update computes; it must target a different staging buffer than the one
// the layer-0 gradient tiles of this launch still read).
// Batched experts: blockIdx.y = expert; its slab offset shifts every pointer except the (shared) replay storage, its
// sampler seed is h.rng.seed + expert * seed_stride.
// ADAM = false (curious_ddpg_grads with `next` on batches of >= 1 280 rows: several processes, several virtual ranks each): the
// gradients only, as dw_all_kernel, WITH the gather blocks -- ridden in the row-local launch instead, as smaller batches have
// it, the gather of 4 864 transitions ends that launch 11 us late (19 ranks: 120.7 us against 109.4)
template <bool PIPE, bool T64 = false, bool ADAM = true>
__global__ __launch_bounds__(256) DW_WAVES_ATTR void dw_adam_her_kernel(DW_ROUTE_PARAMS, const int32_t* fault0, const int64_t* ctr0,
                                                          int64_t ex_stride, DwAllArgs args, AdamFuse A, HerArgs h,
                                                          int64_t grad_stride, uint64_t seed_stride) {
  __shared__ __attribute__((aligned(16))) float red[T64 ? DW64_LDS : 4 * 16 * 64];
#ifdef DW_STAMPS
  DwStamp stamp;
  stamp.t[0] = __builtin_readcyclecounter();                 // (before anything of the arguments is read)
  stamp.t[1] = stamp.t[2] = stamp.t[3] = 0;
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
  DwStamp* sp = &stamp;
#else
  DwStamp* sp = nullptr;
#endif
  const int64_t eo = (int64_t)blockIdx.y * ex_stride;
  const DwRole R = dw_role(hot_nprob * tiles_per, tiles_per, hot_nprob, slots, n_her, r_her, r_hot, units);
  if (R.kind == 0) {
#ifndef DW_NO_GATHER        // (lab, tools/build_variant.py: what the gather blocks cost the launch -- wrong batches, right timing)
    her_sample_body(h, R.idx, red, eo, (uint64_t)blockIdx.y * seed_stride, 0, sp);
#endif
  } else if (R.kind > 0) {
    if constexpr (!ADAM) {
      AdamFuse none;
      none.fault = nullptr;
      dw_tile_role<false, PIPE, T64>(R, args, none, slots, small_nprob, red, eo, grad_stride, nullptr, nullptr);
    } else {
    // the optimiser's two scalar inputs (fault word, step counter): their pointers came with the wave, so the loads go
    // out before the first argument is fetched from memory (fault0 / ctr0 == A.fault / A.step_ctr or a valid dummy)
    AdamEarly early;
    {
      early.fw = *reinterpret_cast<const int32_t*>(reinterpret_cast<const float*>(fault0) + eo);
      const int64_t c = *ex_i64(ctr0, eo);
      early.lo = (int32_t)c; early.hi = (int32_t)(c >> 32);
    }
    dw_tile_role<true, PIPE, T64>(R, args, A, slots, small_nprob, red, eo, grad_stride, sp, &early);
    }
  }
#ifdef DW_STAMPS
  unsigned long long* st = dw_stamp_base(args);
  if (st && threadIdx.x == 0) {
    st[0] = stamp.t[0]; st[1] = stamp.t[1]; st[2] = stamp.t[2]; st[3] = __builtin_readcyclecounter();
    st[4] = (unsigned long long)(R.kind + 1); st[5] = rt0;
    // (+ where the block ran: HW_ID (wave / SIMD / CU / SH / SE) in bits 32-47 of word 7, XCC_ID in bits 48-51)
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    st[6] = stamp.t[3];
    st[7] = (unsigned long long)(unsigned)R.idx | ((unsigned long long)(hw & 0xffffu) << 32) | ((unsigned long long)(xcc & 0xfu) << 48);
  }
#endif
}
