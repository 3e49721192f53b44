// Lean 16x64 split-K tiles for the 256-wide hidden layers: forward / input-gradient kernels with the dot epilogue,
// (the weight-gradient tiles with Adam in the epilogue: mlp_dw.h; included by mlp.hip).
#pragma once

// ================================================================== lean kernels for the hot 256-wide layers
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
