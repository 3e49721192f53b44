// Lean layer-0 forward (segmented input) and the layers-0+1 launch (included by mlp.hip).
#pragma once

// ------------------------------------------------------------------ lean layer-0 forward (total K <= 64)
// Y[M,N] = relu(sum_seg (clip(X_seg) / div) . W_seg + bias): the input is a virtual concatenation of up to 4 column
// segments of row matrices (batch columns [o | td | u], the actor output, g ...), each a multiple of 4 wide.  With
// K <= 64 every wave owns exactly one 16-wide chunk: 1 + 4 loads and 16 MFMAs per wave.
struct SegL { const float* x; const float* W; int32_t ld, w; float div, clip; };
struct L0Prob { SegL seg[MAX_SEG]; const float* bias; float* Y; int32_t nseg, M, N, ldy, relu, ktot; };
struct L0Args { L0Prob p[5]; };

// Branch-free lookup of the segment holding virtual input columns kv .. kv+3 (a 16-byte group never straddles a
// segment: widths % 4 == 0).  The segment table is read unconditionally (unused entries are zero-width); no divergent
// branches, no dependent scalar loads.
struct SegPick { const float* x; const float* W; int ld, kl; float dv, cl; bool ok; };
__device__ __forceinline__ SegPick seg_pick(const L0Prob& P, int kv) {
  // (fields are read straight from the kernarg struct: copying SegL structs around sent them through scratch memory
  //  and turned every dependent load into a flat load)
  const int e0 = P.seg[0].w, e1 = e0 + P.seg[1].w, e2 = e1 + P.seg[2].w, e3 = e2 + P.seg[3].w;   // exclusive ends
  const bool in0 = kv < e0, in1 = kv < e1, in2 = kv < e2, in3 = kv < e3;
  SegPick p;
  p.x = in0 ? P.seg[0].x : in1 ? P.seg[1].x : in2 ? P.seg[2].x : in3 ? P.seg[3].x : P.seg[0].x;
  p.W = in0 ? P.seg[0].W : in1 ? P.seg[1].W : in2 ? P.seg[2].W : in3 ? P.seg[3].W : P.seg[0].W;
  p.ld = in0 ? P.seg[0].ld : in1 ? P.seg[1].ld : in2 ? P.seg[2].ld : in3 ? P.seg[3].ld : P.seg[0].ld;
  p.kl = in3 ? kv - (in0 ? 0 : in1 ? e0 : in2 ? e1 : e2) : 0;                   // column inside the segment
  p.dv = in0 ? P.seg[0].div : in1 ? P.seg[1].div : in2 ? P.seg[2].div : in3 ? P.seg[3].div : 1.0f;
  p.cl = in0 ? P.seg[0].clip : in1 ? P.seg[1].clip : in2 ? P.seg[2].clip : in3 ? P.seg[3].clip : 0.0f;
  p.ok = in3;
  return p;
}
__device__ __forceinline__ bool seg_any_div(const L0Prob& P) {   // uniform: only the critic's action segment divides
  return (P.seg[0].div != 1.0f) || (P.seg[1].w > 0 && P.seg[1].div != 1.0f) ||
         (P.seg[2].w > 0 && P.seg[2].div != 1.0f) || (P.seg[3].w > 0 && P.seg[3].div != 1.0f);
}
__device__ inline f32x4 seg_prep(f32x4 a, float cl, float dv, bool any_div, bool ok) {
  const float c = (cl > 0.0f) ? cl : INFINITY;
#pragma unroll
  for (int e = 0; e < 4; ++e) a[e] = fclip(a[e], -c, c);
  if (any_div) {
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] = fdiv(a[e], dv);
  }
  return sel4(ok, a);
}

// NC = number of 64-wide k chunks (total K <= 64 * NC): wave w owns the 16-wide group w of every chunk
template <int NC>
__device__ inline void fwd_l0_body(const L0Prob& P, float* red, const int64_t eo) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 64;
  const int row = min(m0 + j, P.M - 1);
  f32x4 a[NC], b[NC][4];
  float dv[NC], cl[NC];
  bool ok[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const SegPick sp = seg_pick(P, 64 * c + 16 * wave + 4 * q);
    dv[c] = sp.dv; cl[c] = sp.cl; ok[c] = sp.ok;
    a[c] = ldv(sp.x + eo + sp.kl + (int64_t)row * sp.ld);
    const float* wc = sp.W + eo + (int64_t)sp.kl * P.N + n0 + 4 * j;
#pragma unroll
    for (int s = 0; s < 4; ++s) b[c][s] = ldv(wc + (int64_t)s * P.N);
  }
  const f32x4 bias = ldv(P.bias + eo + n0 + 4 * (tid & 15));
  LOADS_FIRST();
  const bool any_div = seg_any_div(P);
  f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const f32x4 av = seg_prep(a[c], cl[c], dv[c], any_div, ok[c] && (m0 + j < P.M));
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = MFMA(av[s], b[c][s][e], acc[e]);
  }
  int orow, c4;
  f32x4 v = reduce_tile(red, acc, wave, q, j, tid, orow, c4);
  if (m0 + orow >= P.M) return;
  v += bias;
  if (P.relu) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
  }
  *reinterpret_cast<f32x4*>(P.Y + eo + (int64_t)(m0 + orow) * P.ldy + n0 + 4 * c4) = v;
}


template <int NC>
__global__ __launch_bounds__(256) void fwd_l0_kernel(L0Args args) {
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];
  fwd_l0_body<NC>(args.p[blockIdx.z], red, 0);
}

// ------------------------------------------------------------------ layers 0 + 1 in one launch
// C[M,256] = relu(relu(X . W0 + b0) . W1 + b1) for H == 256, total layer-0 K <= 64, M % 16 == 0.  Every workgroup
// first computes the FULL layer-0 tile h0[16 rows][256] of its batch rows (wave w: columns 64w..64w+63, all of K: 4 + 16
// loads and 64 MFMAs), parks it in LDS, and then runs the usual split-K layer-1 tile out of LDS.  The 4 column tiles of
// a row block recompute h0 (0.25 MFLOP each) -- that buys one dependent launch (~4.3 us) per forward pass.  Column
// tile 0 stores h0 when the backward pass needs it.  Problems z >= n01 of the same launch are plain layer-0 problems
// (the action-free pre-activations of fwd_pi_kernel).
struct L01Prob { L0Prob l0; const float* W1; const float* b1; float* C; };
struct L01Args { L01Prob p[3]; L0Prob pre[2]; int32_t n01; };
// LDS row stride of the h0 tile.  A ds_read_b128 is served in 4 groups of 16 lanes (MI355X_MICROARCH.md, LDS), each
// lane covering 4 of the 64 banks; lane (j = row, q) reads dwords j * H0_LD + 4 q + const.  With 264 (= 8 mod 64) the 16
// lanes of every group land on 16 disjoint bank quads and the 8-lane groups of the b128 tile stores stay conflict free
// too; 260 left a 2-way conflict in every read group (SQ_LDS_BANK_CONFLICT = 11 % of this kernel's LDS cycles in
// profiles/r01_pmc_sq_counters.json).
#define H0_LD 264

template <int NC>
__device__ inline void fwd_l01_body(const L01Prob& Q, float* red, float* h0s, const int64_t eo) {
  const L0Prob& P = Q.l0;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 64;
  const int H = 256;
  // ---- all global loads: 4 input fragments + 16 layer-0 weight fragments, 16 layer-1 weight fragments, biases
  constexpr int NG = 4 * NC;                       // 16-wide k groups of layer 0
  f32x4 xa[NG], w0[NG][4], w1[4][4];
  float dv[NG], cl[NG];
  bool okv[NG];
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    const SegPick sp = seg_pick(P, 16 * i + 4 * q);  // this lane's virtual input columns of k-group i
    dv[i] = sp.dv; cl[i] = sp.cl; okv[i] = sp.ok;
    xa[i] = ldv(sp.x + eo + sp.kl + (int64_t)(m0 + j) * sp.ld);
    const float* wc0 = sp.W + eo + (int64_t)sp.kl * H + 64 * wave + 4 * j;
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) w0[i][s2] = ldv(wc0 + (int64_t)s2 * H);
  }
  const f32x4 bias0 = ldv(P.bias + eo + 64 * wave + 4 * j);
  const float* wc1 = Q.W1 + eo + n0 + 4 * j;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int kq = (wave + 4 * u) * 16 + 4 * q;
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) w1[u][s2] = ldv(wc1 + (int64_t)(kq + s2) * H);
  }
  const f32x4 bias1 = ldv(Q.b1 + eo + n0 + 4 * (tid & 15));
  LOADS_FIRST();
  // ---- layer 0: h0[rows 4q..4q+3][cols 64*wave + 4j + e]
  const bool any_div = seg_any_div(P);
  f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    const f32x4 a = seg_prep(xa[i], cl[i], dv[i], any_div, okv[i]);
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = MFMA(a[s2], w0[i][s2][e], acc[e]);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    f32x4 hv = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
    hv += bias0;
#pragma unroll
    for (int e = 0; e < 4; ++e) hv[e] = fmaxf(hv[e], 0.f);
    *reinterpret_cast<f32x4*>(h0s + (4 * q + r) * H0_LD + 64 * wave + 4 * j) = hv;
    if (blockIdx.x == 0 && P.Y)
      *reinterpret_cast<f32x4*>(P.Y + eo + (int64_t)(m0 + 4 * q + r) * H + 64 * wave + 4 * j) = hv;
  }
  __syncthreads();
  // ---- layer 1 out of LDS
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] = zero4();
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int kq = (wave + 4 * u) * 16 + 4 * q;
    const f32x4 a = *reinterpret_cast<const f32x4*>(h0s + j * H0_LD + kq);
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = MFMA(a[s2], w1[u][s2][e], acc[e]);
  }
  int orow, c4;
  f32x4 v = reduce_tile(red, acc, wave, q, j, tid, orow, c4);
  v += bias1;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
  *reinterpret_cast<f32x4*>(Q.C + eo + (int64_t)(m0 + orow) * H + n0 + 4 * c4) = v;
}

template <int NC, bool EX>
__global__ __launch_bounds__(256) void fwd_l01_kernel(L01Args args, Ex ex) {
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];
  __shared__ __attribute__((aligned(16))) float h0s[16 * H0_LD];
  int64_t eo;
  const int pz = ex_decode<EX>(ex, blockIdx.z, eo);
  if (pz < args.n01) fwd_l01_body<NC>(args.p[pz], red, h0s, eo);
  else fwd_l0_body<NC>(args.pre[pz - args.n01], red, eo);
}
