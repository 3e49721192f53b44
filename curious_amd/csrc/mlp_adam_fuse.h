// Adam applied where the gradient is produced: the optimiser's operands and arithmetic for the tile epilogues of the
// weight-gradient launch (included by mlp.hip ahead of mlp_lean_gemm.h / mlp_dw.h).
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

