// Generic (any shape) layer kernels: forward, input gradient, grouped weight gradient (included by mlp.hip).
#pragma once

// ------------------------------------------------------------------ forward layer
// Y[M,N] = act(sum_seg X_seg . W_seg + bias).  grid: x = ceil(N/64), y = ceil(M/16), z = problem
__global__ __launch_bounds__(256) void fwd_layer_kernel(FwdArgs args) {
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];
  const FwdProb& P = args.p[blockIdx.z];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int j = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 64;
  if (m0 >= P.M || n0 >= P.N) return;                       // uniform per workgroup
  const int row = m0 + j;
  const bool row_ok = row < P.M;
  const int col = n0 + 4 * j;                               // this lane's 4 output columns: col + e
  f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
  int ci = 0;                                               // running chunk index over the virtual K
  if (P.fast) {
    const int rowc = min(row, P.M - 1), colc = min(col, P.N - 4);
    for (int sidx = 0; sidx < P.nseg; ++sidx) {
      const Seg& S = P.seg[sidx];
      const int nch = (S.w + 15) >> 4;
      const float* xr = S.x + (int64_t)rowc * S.ld;
      for (int c0 = ((wave - ci) & 3); c0 < nch; c0 += 16) {
        f32x4 a[4], b[4][4];
        int kqs[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int kq = (c0 + 4 * u) * 16 + 4 * q;
          ok[u] = row_ok && (kq < S.w);
          kqs[u] = min(kq, S.w - 4);
          a[u] = ldv(xr + kqs[u]);
#pragma unroll
          for (int s = 0; s < 4; ++s) b[u][s] = ldv(S.W + (int64_t)(kqs[u] + s) * P.N + colc);
        }
        LOADS_FIRST();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          a[u] = sel4(ok[u], seg_xform4(S, a[u], kqs[u]));
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = MFMA(a[u][s], b[u][s][e], acc[e]);
        }
      }
      ci += nch;
    }
  } else {
    for (int sidx = 0; sidx < P.nseg; ++sidx) {
      const Seg& S = P.seg[sidx];
      const int nch = (S.w + 15) >> 4;
      // this wave's chunks of the segment: (ci + c) % 4 == wave; up to 4 chunks are loaded before any MFMA
      for (int c0 = ((wave - ci) & 3); c0 < nch; c0 += 16) {
        f32x4 a[4], b[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int kq = (c0 + 4 * u) * 16 + 4 * q;
          const bool cok = (c0 + 4 * u) < nch;
          a[u] = cok ? seg_load4(S, row, kq, row_ok) : zero4();
#pragma unroll
          for (int s = 0; s < 4; ++s)
            b[u][s] = (cok && kq + s < S.w) ? ldg4(S.W + (int64_t)(kq + s) * P.N + col, P.N - col, P.wvec != 0)
                                            : zero4();
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = MFMA(a[u][s], b[u][s][e], acc[e]);
      }
      ci += nch;
    }
  }
  int orow, c4;
  f32x4 v = reduce_tile(red, acc, wave, q, j, tid, orow, c4);
  const int grow = m0 + orow, gcol = n0 + 4 * c4;
  if (grow >= P.M || gcol >= P.N) return;
  f32x4 bias = P.bias ? ldg4(P.bias + gcol, P.N - gcol, P.wvec != 0) : zero4();
  v += bias;
  if (P.act == 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
  }
  float* dst = P.Y + (int64_t)grow * P.ldy + gcol;
  if (gcol + 3 < P.N && (P.ldy & 3) == 0) {
    *reinterpret_cast<f32x4*>(dst) = v;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (gcol + e < P.N) dst[e] = v[e];
  }
}

// ------------------------------------------------------------------ backward: input gradient
// dX[m][k] = sum_n dY[m][n] W[k][n], masked by relu'(H).  grid: x = ceil(K/64), y = ceil(M/16), z = problem
__global__ __launch_bounds__(256) void dx_kernel(DxArgs args) {
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];
  const DxProb& P = args.p[blockIdx.z];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int j = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.y * 16, k0 = blockIdx.x * 64;
  if (m0 >= P.M || k0 >= P.K) return;
  const int row = m0 + j;
  const bool row_ok = row < P.M;
  const int kc = k0 + 4 * j;                                 // output columns kc + e  <->  weight rows kc + e
  f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
  const int nch = (P.N + 15) >> 4;
  if (P.fast) {
    const float* dyr = P.dY + (int64_t)min(row, P.M - 1) * P.lddy;
    const float* wr[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) wr[e] = P.W + (int64_t)min(kc + e, P.K - 1) * P.ldw;
    for (int c0 = wave; c0 < nch; c0 += 16) {
      f32x4 a[4], b[4][4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int nq = (c0 + 4 * u) * 16 + 4 * q;
        ok[u] = row_ok && (nq < P.N);
        const int nqc = min(nq, P.N - 4);
        a[u] = ldv(dyr + nqc);
#pragma unroll
        for (int e = 0; e < 4; ++e) b[u][e] = ldv(wr[e] + nqc);
      }
      LOADS_FIRST();
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = sel4(ok[u], a[u]);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = MFMA(a[u][s], b[u][e][s], acc[e]);
      }
    }
  } else {
    const float* dyr = P.dY + (int64_t)row * P.lddy;
    for (int c0 = wave; c0 < nch; c0 += 16) {
      f32x4 a[4], b[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int nq = (c0 + 4 * u) * 16 + 4 * q;
        const bool cok = (c0 + 4 * u) < nch;
        a[u] = (cok && row_ok) ? ldg4(dyr + nq, P.N - nq, P.vec != 0) : zero4();
#pragma unroll
        for (int e = 0; e < 4; ++e)
          b[u][e] = (cok && kc + e < P.K) ? ldg4(P.W + (int64_t)(kc + e) * P.ldw + nq, P.N - nq, P.vec != 0)
                                          : zero4();
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = MFMA(a[u][s], b[u][e][s], acc[e]);
    }
  }
  int orow, c4;
  f32x4 v = reduce_tile(red, acc, wave, q, j, tid, orow, c4);
  const int grow = m0 + orow, gcol = k0 + 4 * c4;
  if (grow >= P.M || gcol >= P.K) return;
  if (P.H) {
    f32x4 h = ldg4(P.H + (int64_t)grow * P.ldh + gcol, P.K - gcol, (P.ldh & 3) == 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (h[e] > 0.f) ? v[e] : 0.f;
  }
  float* dst = P.dX + (int64_t)grow * P.lddx + gcol;
  if (gcol + 3 < P.K && (P.lddx & 3) == 0) {
    *reinterpret_cast<f32x4*>(dst) = v;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (gcol + e < P.K) dst[e] = v[e];
  }
}

// ------------------------------------------------------------------ backward: weight gradient (grouped)
// dW[k][n] = sum_m X[m][k] dY[m][n]; db[n] = sum_m dY[m][n].  grid: x = ceil(N/64), y = ceil(w/16), z = problem;
// slice z == nprob finalises the losses.
__global__ __launch_bounds__(256) void dw_kernel(DwArgs args) {
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if ((int)blockIdx.z == args.nprob) {
    // losses (ddpg.py:439-441) from the per-row terms, summed in a fixed order by one workgroup
    if (blockIdx.x != 0 || blockIdx.y != 0) return;
    const LossFin& F = args.fin;
    float lq = 0.f, lp = 0.f, ll = 0.f;
    for (int m = tid; m < F.Bl; m += 256) {                  // (the first rank's rows; the others: loss_fin_ranks)
      lq += F.rows[m];
      lp += F.rows[F.B + m];
      ll += F.rows[2 * F.B + m];
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
      if (F.step_ctr) *F.step_ctr += 1;
      loss_fin_flag(F, 0, 0);
      F.out[0] = red[0] * invB;
      F.out[1] = -red[256] * invB + F.action_l2 * red[512] / (float)(F.Bl * F.U);
    }
    if (F.Bl < F.B) loss_fin_ranks(F, F.rows, F.out);
    return;
  }
  const DwProb& P = args.p[blockIdx.z];
  const int j = lane & 15, q = lane >> 4;
  const int k0 = blockIdx.y * 16, n0 = blockIdx.x * 64;
  if (k0 >= P.x.w || n0 >= P.N) return;
  const int krow = k0 + j;                                   // A operand row index = weight row
  const bool k_ok = krow < P.x.w;
  const int col = n0 + 4 * j;
  f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
  f32x4 bsum = zero4();
  const int nch = (P.M + 15) >> 4;
  if (P.fast) {
    const Seg& X = P.x;
    const int krc = min(krow, X.w - 1), colc = min(col, P.N - 4);
    const bool plain = X.clip <= 0.0f && !X.mean && X.div == 1.0f;
    const float mu = X.mean ? X.mean[krc] : 0.f, sd = X.mean ? X.stdv[krc] : 1.f;
    for (int c0 = wave; c0 < nch; c0 += 16) {
      float a[4][4];
      f32x4 b[4][4];
      bool ok[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int mq = (c0 + 4 * u) * 16 + 4 * q;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          ok[u][s] = (mq + s) < P.M;
          const int mc = min(mq + s, P.M - 1);
          a[u][s] = X.x[(int64_t)mc * X.ld + krc];
          b[u][s] = ldv(P.dY + (int64_t)mc * P.lddy + colc);
        }
      }
      LOADS_FIRST();
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          float av = a[u][s];
          if (!plain) {
            if (X.clip > 0.0f) av = fclip(av, -X.clip, X.clip);
            if (X.mean) av = fclip(fdiv(__fsub_rn(av, mu), sd), -X.nclip, X.nclip);
            if (X.div != 1.0f) av = fdiv(av, X.div);
          }
          av = (ok[u][s] && k_ok) ? av : 0.f;
          f32x4 bv = sel4(ok[u][s], b[u][s]);
          bsum += bv;
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = MFMA(av, bv[e], acc[e]);
        }
    }
  } else {
    for (int c0 = wave; c0 < nch; c0 += 16) {
      float a[4][4];
      f32x4 b[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int mq = (c0 + 4 * u) * 16 + 4 * q;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const bool mok = ((c0 + 4 * u) < nch) && (mq + s < P.M);
          a[u][s] = seg_load1(P.x, mq + s, krow, mok && k_ok);
          b[u][s] = mok ? ldg4(P.dY + (int64_t)(mq + s) * P.lddy + col, P.N - col, P.yvec != 0) : zero4();
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          bsum += b[u][s];
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = MFMA(a[u][s], b[u][s][e], acc[e]);
        }
    }
  }
  int orow, c4;
  f32x4 v = reduce_tile(red, acc, wave, q, j, tid, orow, c4);
  const int grow = k0 + orow, gcol = n0 + 4 * c4;
  if (grow < P.x.w && gcol < P.N) {
    float* dst = P.dW + (int64_t)grow * P.N + gcol;
    if (gcol + 3 < P.N && (P.N & 3) == 0 && (((uintptr_t)P.dW) & 15) == 0) {
      *reinterpret_cast<f32x4*>(dst) = v;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (gcol + e < P.N) dst[e] = v[e];
    }
  }
  if (P.db && blockIdx.y == 0) {
    // bias gradient: every lane summed its (q, s, chunk) share of 4 columns; fold q-groups, then the 4 waves
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = bsum[e];
      t += __shfl_xor(t, 16);
      t += __shfl_xor(t, 32);
      bsum[e] = t;
    }
    __syncthreads();                                         // `red` is free again
    if (q == 0) *reinterpret_cast<f32x4*>(red + wave * 64 + 4 * j) = bsum;
    __syncthreads();
    if (tid < 64 && n0 + tid < P.N) P.db[n0 + tid] = red[tid] + red[64 + tid] + red[128 + tid] + red[192 + tid];
  }
}
