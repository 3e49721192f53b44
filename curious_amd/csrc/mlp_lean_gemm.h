// Lean 16x64 split-K tiles for the 256-wide hidden layers: forward / input-gradient kernels with the dot epilogue,
// weight-gradient tiles with Adam in the epilogue, the update tail launch (included by mlp.hip).
#pragma once

// ================================================================== lean kernels for the hot 256-wide layers
// Adam applied where the gradient is produced (curious_ddpg_update, single-rank): the workgroup that finishes a tile
// of dW / db owns the matching elements of theta, m and v, so the optimiser needs no launch of its own.  Arithmetic and
// step-size lookup are those of optim.hip's adam_body (mpi_adam.py:29-35), bit for bit.
struct AdamFuse {
  float* theta; float* m; float* v;
  const float* grad;              // base of the gradient vector: (gradient pointer - grad) = parameter index
  int64_t n_Q;
  const float* alpha_tab; const int64_t* step_ctr; int64_t tab_base; int32_t tab_len;
  float a_Q, a_pi, b1, omb1, b2, omb2, eps;
  const int32_t* fault;           // fault word of the gradient workspace (mlp_rows.h) or NULL: non-zero = skip the optimiser
};
__device__ inline bool adam_faulted(const AdamFuse& A, int64_t eo) {
  return A.fault && *reinterpret_cast<const int32_t*>(reinterpret_cast<const float*>(A.fault) + eo) != 0;
}

// eo: slab offset of the expert this block works for (0 for a single agent); i / pidx / bidx below are indices into
// the parameter vector, the moments and the parameters are addressed at index + eo.  eg: the same for the GRADIENT
// vectors, which the experts keep in a contiguous [N, P] block of their own (grad_stride = P) so that several ranks sum
// all of them in ONE all-reduce (curious_ddpg_grads_experts); every pointer into a gradient vector is shifted by eg.
__device__ inline void adam_alphas(const AdamFuse& A, float& aQ, float& aPi, int64_t eo) {
  aQ = A.a_Q; aPi = A.a_pi;
  if (A.alpha_tab) {
    int64_t idx = ((*ex_i64(A.step_ctr, eo)) - 1 - A.tab_base) % A.tab_len;
    if (idx < 0) idx += A.tab_len;
    aQ = A.alpha_tab[eo + 2 * idx];
    aPi = A.alpha_tab[eo + 2 * idx + 1];
  }
}

// The optimiser's two scalar inputs without their latency: a tile used to begin with  load fault word -> wait -> load
// step counter -> wait -> (64-bit modulo) -> load step sizes -> ... -> first operand load, three dependent round trips
// through cold caches before its own 80 KB were even requested.  adam_early() only ISSUES the two loads (branch-free: a
// NULL pointer reads a valid dummy address and is masked later); the verdict on the fault word is taken in the tile's
// epilogue, and the step sizes are looked up (adam_alphas_late) once the tile's operand loads are in flight -- their
// round trip hides behind the matrix instructions.
// (PIN_V: an empty asm that takes the value through a vector register.  The loaded words are uniform, and hipcc would
//  move them to scalar registers -- v_readfirstlane behind an s_waitcnt -- right where they are loaded; behind the pin
//  they count as per-lane values, so the wait sits where the pin is and the arithmetic that follows stays in the VALU.)
#define PIN_V(x) asm volatile("" : "+v"(x))
#define DW_INLINE __forceinline__
// Operand address = wave-uniform base pointer + 32-bit per-lane byte offset: the form hipcc emits as
//   global_load vdst, voff, s[base:base+1]
// i.e. the row part of every address is SALU work (an s_add / s_addc pair beside the vector pipeline) and the lane part is ONE
// register for the whole tile.  With 64-bit per-lane pointers a tile spent 160 VALU instructions (5 per load, 2 waves per
// SIMD: ~1.5 k cycles) on addresses before its first operand load went out (tools/dw_stamps.py).
__device__ __forceinline__ float ld_su(const float* ubase, uint32_t lane_bytes) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(ubase) + lane_bytes);
}
__device__ __forceinline__ f32x4 ld4_su(const float* ubase, uint32_t lane_bytes) {
  return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(ubase) + lane_bytes);
}
// t / nx, t % nx for a uniform t: a shift when nx is a power of two (the hidden layers: nx = 4) instead of the
// v_rcp-based division sequence
__device__ __forceinline__ void tile_divmod(const int t, const int nx, int& by, int& bx) {
  if ((nx & (nx - 1)) == 0) {
    const int sh = __builtin_ctz(nx);
    by = t >> sh; bx = t & (nx - 1);
  } else {
    by = t / nx; bx = t - by * nx;
  }
}

struct AdamEarly { int32_t fw, lo, hi; };
__device__ inline AdamEarly adam_early(const AdamFuse& A, int64_t eo) {
  AdamEarly e;

  const int32_t* fp = A.fault ? reinterpret_cast<const int32_t*>(reinterpret_cast<const float*>(A.fault) + eo)
                              : reinterpret_cast<const int32_t*>(A.theta);
  const int64_t* cp = A.alpha_tab ? ex_i64(A.step_ctr, eo) : reinterpret_cast<const int64_t*>(A.theta);
  e.fw = *fp;
  const int64_t c = *cp;
  e.lo = (int32_t)c; e.hi = (int32_t)(c >> 32);
  return e;
}
__device__ inline bool adam_early_faulted(const AdamFuse& A, AdamEarly& e) {
  PIN_V(e.fw);
  return A.fault && e.fw != 0;
}
__device__ inline void adam_alphas_late(const AdamFuse& A, AdamEarly& e, float& aQ, float& aPi, int64_t eo) {
  aQ = A.a_Q; aPi = A.a_pi;
  if (A.alpha_tab) {
    PIN_V(e.lo); PIN_V(e.hi);
    const int64_t ctr = (int64_t)(((uint64_t)(uint32_t)e.hi << 32) | (uint32_t)e.lo);
    const int64_t v = ctr - 1 - A.tab_base;
    int64_t idx;
    if ((A.tab_len & (A.tab_len - 1)) == 0) {
      idx = v & (int64_t)(A.tab_len - 1);                   // the ring of ALPHA_TAB = 4096 entries: no 64-bit division
    } else {
      idx = v % A.tab_len;
      if (idx < 0) idx += A.tab_len;
    }
    aQ = A.alpha_tab[eo + 2 * idx];
    aPi = A.alpha_tab[eo + 2 * idx + 1];
  }
}

__device__ inline float adam_elem(const AdamFuse& A, float na, float g, float& m, float& v, float th) {
  m = __fadd_rn(__fmul_rn(A.b1, m), __fmul_rn(A.omb1, g));                       // mpi_adam.py:31
  v = __fadd_rn(__fmul_rn(A.b2, v), __fmul_rn(A.omb2, __fmul_rn(g, g)));         // mpi_adam.py:32
  const float step = fdiv(__fmul_rn(na, m), __fadd_rn(sqrtf(v), A.eps));         // mpi_adam.py:33
  return __fadd_rn(th, step);                                                    // mpi_adam.py:34
}

struct AdamPre4 { f32x4 m, v, th; };
__device__ inline AdamPre4 adam_prefetch4(const AdamFuse& A, int64_t i) {
  AdamPre4 p;
  p.m = ldv(A.m + i); p.v = ldv(A.v + i); p.th = ldv(A.theta + i);
  return p;
}
__device__ inline void adam_apply4(const AdamFuse& A, float na, int64_t i, const f32x4& g, AdamPre4& p) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float m = p.m[e], v = p.v[e];
    p.th[e] = adam_elem(A, na, g[e], m, v, p.th[e]);
    p.m[e] = m; p.v[e] = v;
  }
  *reinterpret_cast<f32x4*>(A.m + i) = p.m;
  *reinterpret_cast<f32x4*>(A.v + i) = p.v;
  *reinterpret_cast<f32x4*>(A.theta + i) = p.th;
}
__device__ inline void adam_apply1(const AdamFuse& A, float na, int64_t i, float g) {
  float m = A.m[i], v = A.v[i];
  const float th = adam_elem(A, na, g, m, v, A.theta[i]);
  A.m[i] = m; A.v[i] = v; A.theta[i] = th;
}

// Same tiling as the generic kernels above, but with 56-byte problem descriptors, no bounds checks and no segment
// machinery: tools/gemm_lab.hip measures 3.7 us per launch inside a hipGraph for this form (2.0 us of which is the
// launch floor of an empty kernel) against 5.5-7 us for the generic form with its 1 KB kernarg.
// Preconditions (checked on the host, else the generic kernel runs): M % 16 == 0, N % 64 == 0, reduction dim % 256
// == 0, all pointers 16-byte aligned, leading dims % 4 == 0.
struct GemmHot {
  const float* A; const float* B; const float* aux; float* C; float* aux_out;
  int32_t lda, ldb, ldc, M, N, K;
  // (weight-gradient tiles with the optimiser epilogue: dot_out = where the TRANSPOSED copy of the updated matrix is
  //  kept, [N][K], or NULL)
  // optional epilogue (DOT kernels): partial products of the output tile with a narrow matrix that the NEXT launch
  // would otherwise have to contract over whole rows (output layers, the critic's action rows):
  //   dot_out[tile][m][d] = sum_{c in this 64-column tile} C[m][c] * w(c, d)
  // dot_mode 1: D = 1, w = dot_w[c];  2: D = 4, w = dot_w[c * 4 + d];  3: D = 4, w = dot_w[d * dot_ld + c]
  const float* dot_w; float* dot_out; int32_t dot_mode, dot_ld;
};
struct HotArgs { GemmHot p[4]; };       // <= 3 problems; p[3] stays zeroed (the XCD-aware grids below index it)

// Block id -> (column tile, row block, problem).  XR == 0: the plain grid (N/64, M/16, nprob).  XR == 4 / 8 (M == 256,
// N == 256 only): XCD-aware placement.  Workgroups are dealt round-robin over the 8 XCDs in linear block-id order, so with
// grid (8, 4 * XR, rounds) blockIdx.x IS the XCD (speed only -- nothing depends on it for correctness).  A "unit" = XR
// consecutive row blocks x all 4 column tiles of one problem lives on one XCD: its activation rows are fetched into
// that XCD's L2 once instead of four times, its weight matrix 16 / XR times instead of twice per column tile.  Units
// beyond the last problem read the zeroed descriptor and exit.
template <int XR>
__device__ __forceinline__ void tile_ids(int& bx, int& by, int& pz) {
  if (XR == 0) { bx = blockIdx.x; by = blockIdx.y; pz = blockIdx.z; return; }
  constexpr int UPC = 16 / (XR ? XR : 1);                  // units per problem at M = 256
  const int unit = blockIdx.x + 8 * blockIdx.z;
  pz = unit / UPC;
  bx = blockIdx.y & 3;
  by = (unit % UPC) * XR + (blockIdx.y >> 2);
}

__device__ inline void hot_store(float* red, const f32x4 acc[4], int wave, int q, int j, int tid, f32x4& v, int& orow,
                                 int& c4) {
  v = reduce_tile(red, acc, wave, q, j, tid, orow, c4);
}

struct DotW { f32x4 w[4]; };
__device__ inline DotW dot_prefetch(const GemmHot& P, int c, int64_t eo) {
  // branch-free (4 unconditional loads at selected addresses): a branch on dot_mode here would make every load that
  // follows in program order wait for the scalar load of dot_mode.  dot_w is a valid address for every problem of a
  // DOT launch (the host points it at the weight matrix when dot_mode == 0).
  DotW d;
  const int m = P.dot_mode;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int64_t off = (m == 2) ? (int64_t)(c + e) * 4 : (m == 3) ? (int64_t)e * P.dot_ld + c : (m == 1) ? c : 0;
    d.w[e] = ldv(P.dot_w + eo + off);
  }
  return d;
}
// v: this thread's 4 consecutive output columns of row `row`; the 16 threads of a row are one DPP row
__device__ inline void dot_epilogue(const GemmHot& P, const DotW& d, const f32x4& v, int row, int tile, int c4,
                                    int64_t eo) {
  if (P.dot_mode == 0) return;
  f32x4 pd = zero4();
  if (P.dot_mode == 1) {
    pd[0] = v[0] * d.w[0][0] + v[1] * d.w[0][1] + v[2] * d.w[0][2] + v[3] * d.w[0][3];
    pd[0] = row16_sum(pd[0]);
    if (c4 == 0) P.dot_out[eo + (int64_t)tile * P.M + row] = pd[0];
    return;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float t = 0.f;
    if (P.dot_mode == 2) t = v[0] * d.w[0][k] + v[1] * d.w[1][k] + v[2] * d.w[2][k] + v[3] * d.w[3][k];
    else t = v[0] * d.w[k][0] + v[1] * d.w[k][1] + v[2] * d.w[k][2] + v[3] * d.w[k][3];
    pd[k] = row16_sum(t);
  }
  if (c4 == 0) *reinterpret_cast<f32x4*>(P.dot_out + eo + ((int64_t)tile * P.M + row) * 4) = pd;
}

// C[M,N] = relu(A[M,K] . B[K,N] + bias)        grid (N/64, M/16, nprob)
template <bool DOT, bool EX, int XR = 0>
__global__ __launch_bounds__(256) void fwd_hot_kernel(HotArgs args, Ex ex) {
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];
  int64_t eo;
  int bx, by, bz;
  tile_ids<XR>(bx, by, bz);
  const GemmHot& P = args.p[ex_decode<EX>(ex, bz, eo)];
  if (XR != 0 && P.A == nullptr) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, q = lane >> 4;
  const int m0 = by * 16, n0 = bx * 64;
  const float* xr = P.A + eo + (int64_t)(m0 + j) * P.lda;
  const float* wc = P.B + eo + n0 + 4 * j;
  const f32x4 bias = ldv(P.aux + eo + n0 + 4 * (tid & 15));      // epilogue operand, issued with the first batch
  DotW dw;
  if (DOT) dw = dot_prefetch(P, n0 + 4 * (tid & 15), eo);
  f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
  for (int kb = 0; kb < P.K; kb += 256) {
    f32x4 a[4], b[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int kq = kb + (wave + 4 * u) * 16 + 4 * q;
      a[u] = ldv(xr + kq);
#pragma unroll
      for (int s = 0; s < 4; ++s) b[u][s] = ldv(wc + (int64_t)(kq + s) * P.ldb);
    }
    LOADS_FIRST();
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = MFMA(a[u][s], b[u][s][e], acc[e]);
  }
  f32x4 v; int orow, c4;
  hot_store(red, acc, wave, q, j, tid, v, orow, c4);
  v += bias;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
  *reinterpret_cast<f32x4*>(P.C + eo + (int64_t)(m0 + orow) * P.ldc + n0 + 4 * c4) = v;
  if (DOT) dot_epilogue(P, dw, v, m0 + orow, bx, c4, eo);
}

// C[M,K'] = (A[M,N] . B[K',N]^T) * relu'(aux[M,K'])    (K' = P.N output columns, reduction over P.K)   grid (K'/64, M/16, nprob)
template <bool DOT, bool EX, int XR = 0>
__global__ __launch_bounds__(256) void dx_hot_kernel(HotArgs args, Ex ex) {
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];
  int64_t eo;
  int bx, by, bz;
  tile_ids<XR>(bx, by, bz);
  const GemmHot& P = args.p[ex_decode<EX>(ex, bz, eo)];
  if (XR != 0 && P.A == nullptr) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, q = lane >> 4;
  const int m0 = by * 16, k0 = bx * 64;
  const float* dyr = P.A + eo + (int64_t)(m0 + j) * P.lda;
  const float* wr = P.B + eo + (int64_t)(k0 + 4 * j) * P.ldb;
  const int64_t o = eo + (int64_t)(m0 + (tid >> 4)) * P.ldc + k0 + 4 * (tid & 15);
  const f32x4 h = ldv(P.aux + o);                           // relu mask source, issued with the first batch
  DotW dw;
  if (DOT) dw = dot_prefetch(P, k0 + 4 * (tid & 15), eo);
  f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
  for (int nb = 0; nb < P.K; nb += 256) {
    f32x4 a[4], b[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int nq = nb + (wave + 4 * u) * 16 + 4 * q;
      a[u] = ldv(dyr + nq);
#pragma unroll
      for (int e = 0; e < 4; ++e) b[u][e] = ldv(wr + (int64_t)e * P.ldb + nq);
    }
    LOADS_FIRST();
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = MFMA(a[u][s], b[u][e][s], acc[e]);
  }
  f32x4 v; int orow, c4;
  hot_store(red, acc, wave, q, j, tid, v, orow, c4);
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = (h[e] > 0.f) ? v[e] : 0.f;
  *reinterpret_cast<f32x4*>(P.C + o) = v;
  if (DOT) dot_epilogue(P, dw, v, m0 + orow, bx, c4, eo);
}

// C[K',N] = A[M,K']^T . B[M,N];  aux_out[N] = colsum(B)    (reduction over P.M)     1-D grid over a tile list
struct DwHotArgs { GemmHot p[4]; int32_t tiles_per; int32_t nprob; };   // every problem has tiles_per tiles
template <bool ADAM>
__device__ DW_INLINE void dw_hot_tile(const GemmHot& P, const AdamFuse& A, const int t, float* red,
                                   const int64_t eo, const int64_t eg, DwStamp* stamps = nullptr,
                                   const AdamEarly* given = nullptr) {
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
    pre = adam_prefetch4(A, pidx + eo);
    if (by == 0 && tid < 64) { bm = A.m[bidx + eo]; bv = A.v[bidx + eo]; bth = A.theta[bidx + eo]; }
  }
  f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
  f32x4 bsum = zero4();
  for (int mb = 0; mb < P.M; mb += 256) {
    float a[4][4];
    f32x4 b[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int mu = mb + (wv + 4 * u) * 16;                  // (+ 4 q: in the lane offsets)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        a[u][s] = ld_su(xu + (int64_t)(mu + s) * P.lda, xo);
        b[u][s] = ld4_su(yu + (int64_t)(mu + s) * P.ldb, yo);
      }
    }
    LOADS_FIRST();
    DW_STAMP(stamps, 1);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bsum += b[u][s];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = MFMA(a[u][s], b[u][s][e], acc[e]);
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
    // column sums of B: 16 partials (4 waves x 4 lane groups) per column through LDS
    __syncthreads();
    *reinterpret_cast<f32x4*>(red + (wave * 4 + q) * 64 + 4 * j) = bsum;
    __syncthreads();
    if (tid < 64) {
      float gb = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) gb += red[r * 64 + tid];
      P.aux_out[eg + n0 + tid] = gb;
      if (ADAM && !faulted) {
        const float th = adam_elem(A, (bidx < A.n_Q) ? -aQ : -aPi, gb, bm, bv, bth);
        A.m[bidx + eo] = bm; A.v[bidx + eo] = bv; A.theta[bidx + eo] = th;
      }
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
template <bool ADAM, bool YV>
__device__ DW_INLINE void dw_small_tile(const DwSmall& P, const int M, const AdamFuse& A, const int t, float* red,
                                     const int64_t eo, const int64_t eg, DwStamp* stamps = nullptr,
                                     const AdamEarly* given = nullptr) {
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
    if (YV) {
      pre = adam_prefetch4(A, (own ? pidx : 0) + eo);
    } else if (own) {                                       // N == 1: one element per owning thread
      pre.m[0] = A.m[pidx + eo]; pre.v[0] = A.v[pidx + eo]; pre.th[0] = A.theta[pidx + eo];
    }
    if (own_b) { bm = A.m[bidx + eo]; bv = A.v[bidx + eo]; bth = A.theta[bidx + eo]; }
  }
  f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
  f32x4 bsum = zero4();
  for (int mb = 0; mb < M; mb += 256) {
    float a[4][4];
    f32x4 b[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int mu = mb + (wv + 4 * u) * 16;                  // (+ 4 q: in the lane offsets)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        a[u][s] = ld_su(xu + (int64_t)(mu + s) * P.ldx, xo);
        if (YV) {
          b[u][s] = ld4_su(yu + (int64_t)(mu + s) * P.lddy, yo);
        } else {
          b[u][s] = zero4();
          b[u][s][0] = ld_su(yu + (int64_t)(mu + s) * P.lddy, yo);
        }
      }
    }
    LOADS_FIRST();
    DW_STAMP(stamps, 1);
    if (P.div != 1.0f) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s) a[u][s] = fdiv(a[u][s], P.div);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float av = k_ok ? a[u][s] : 0.f;
        bsum += b[u][s];
        if (YV) {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = MFMA(av, b[u][s][e], acc[e]);
        } else {
          acc[0] = MFMA(av, b[u][s][0], acc[0]);
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
    __syncthreads();
    *reinterpret_cast<f32x4*>(red + (wave * 4 + q) * 64 + 4 * j) = bsum;
    __syncthreads();
    if (own_b) {
      float gb = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) gb += red[r * 64 + tid];
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
struct DwAllArgs { DwHotArgs hot; DwSmallArgs small; int32_t n_hot; unsigned long long* stamps; };
// lab (a build with -DDW_STAMPS, tools/build_variant.py, + option "lab_dw_stamps"): 8 x 64-bit words per block of expert 0 -- [0] entry, [1] operand loads issued /
// gather: tables in, [2] MFMA loop over / gather: rows in LDS, [3] exit, [4] kind (0 gather, 1 hidden tile, 2 small),
// [5] s_memrealtime at entry (100 MHz, device-wide)
__device__ inline unsigned long long* dw_stamp_base(const DwAllArgs& a) {
  return (a.stamps && blockIdx.y == 0) ? a.stamps + (size_t)blockIdx.x * 8 : nullptr;
}
struct DwRole { int kind, pi, idx; };
__device__ __forceinline__ DwRole dw_role(const int n_hot, const int tiles_per, const int hot_nprob, const int slots,
                                          const int n_her, const int r_her, const int r_hot, const int units) {
  DwRole R;
  R.kind = -1; R.pi = 0; R.idx = 0;
  const int b = (int)blockIdx.x;
  if (units == 0) {
    const int bid = b - n_her;
    if (bid < 0) { R.kind = 0; R.idx = b; }
    else if (bid < n_hot) { R.kind = 1; R.pi = bid / tiles_per; R.idx = bid - R.pi * tiles_per; }
    else { R.kind = 2; R.idx = bid - n_hot; }
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
  if (r < r_hot) {
    const int pi = x / units, u = x - pi * units;
    if (pi < hot_nprob) { R.kind = 1; R.pi = pi; R.idx = u * r_hot + r; }
  } else {
    const int j = r - r_hot;
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
                        const int r_her, const int r_hot, const int units
// the tile work of a block whose role is known: ONE batch of argument loads, then the tile
template <bool ADAM>
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
    dw_hot_tile<ADAM>(P, A, R.idx, red, eo, (int64_t)blockIdx.y * grad_stride, sp, early);
    return;
  }
  const int pi = R.idx / slots, t = R.idx - pi * slots;
  if (pi >= small_nprob) {
    // the block right behind the last problem finalises the losses (fin.rows != NULL); any other id beyond exits
    if (pi == small_nprob && t == 0) dw_loss_fin(args.small.fin, red, eo, (int64_t)blockIdx.y * grad_stride);
    return;
  }
  DwSmall P = args.small.p[pi];
  int M = args.small.M;
  pin_small(P);
  if (ADAM) pin_adam(A);
  asm volatile("" :: "s"(grad_stride), "s"(M));
  DW_STAMP(sp, 3);
  if (t >= ((P.w + 15) >> 4) * ((P.N + 63) >> 6)) return;
  const int64_t eg = (int64_t)blockIdx.y * grad_stride;
  if ((P.N & 3) == 0) dw_small_tile<ADAM, true>(P, M, A, t, red, eo, eg, sp, early);
  else dw_small_tile<ADAM, false>(P, M, A, t, red, eo, eg, sp, early);
}

__global__ __launch_bounds__(256) void dw_all_kernel(DW_ROUTE_PARAMS, int64_t ex_stride, DwAllArgs args,
                                                     int64_t grad_stride) {
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];
  AdamFuse none;
  none.fault = nullptr;
  const int64_t eo = (int64_t)blockIdx.y * ex_stride;
  const DwRole R = dw_role(hot_nprob * tiles_per, tiles_per, hot_nprob, slots, 0, r_her, r_hot, units);
  if (R.kind > 0) dw_tile_role<false>(R, args, none, slots, small_nprob, red, eo, grad_stride, nullptr, nullptr);
}

// The tail of a whole single-rank update in one launch (curious_ddpg_update): every weight/bias gradient with Adam
// applied in the tile epilogue, the loss finalisation, and -- in the first n_her blocks -- the HER gather of the NEXT
// update's batch (it depends on nothing this update computes; it must target a different staging buffer than the one
// the layer-0 gradient tiles of this launch still read).
// Batched experts: blockIdx.y = expert; its slab offset shifts every pointer except the (shared) replay storage, its
// sampler seed is h.rng.seed + expert * seed_stride.
__global__ __launch_bounds__(256) void dw_adam_her_kernel(DW_ROUTE_PARAMS, const int32_t* fault0, const int64_t* ctr0,
                                                          int64_t ex_stride, DwAllArgs args, AdamFuse A, HerArgs h,
                                                          int64_t grad_stride, uint64_t seed_stride) {
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];
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
    her_sample_body(h, R.idx, red, eo, (uint64_t)blockIdx.y * seed_stride, 0, sp);
  } else if (R.kind > 0) {
    // the optimiser's two scalar inputs (fault word, step counter): their pointers came with the wave, so the loads go
    // out before the first argument is fetched from memory (fault0 / ctr0 == A.fault / A.step_ctr or a valid dummy)
    AdamEarly early;
    {
      early.fw = *reinterpret_cast<const int32_t*>(reinterpret_cast<const float*>(fault0) + eo);
      const int64_t c = *ex_i64(ctr0, eo);
      early.lo = (int32_t)c; early.hi = (int32_t)(c >> 32);
    }
    dw_tile_role<true>(R, args, A, slots, small_nprob, red, eo, grad_stride, sp, &early);
  }
#ifdef DW_STAMPS
  unsigned long long* st = dw_stamp_base(args);
  if (st && threadIdx.x == 0) {
    st[0] = stamp.t[0]; st[1] = stamp.t[1]; st[2] = stamp.t[2]; st[3] = __builtin_readcyclecounter();
    st[4] = (unsigned long long)(R.kind + 1); st[5] = rt0;
    st[6] = stamp.t[3]; st[7] = (unsigned long long)R.idx;
  }
#endif
}
