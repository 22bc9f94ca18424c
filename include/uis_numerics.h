/*
 * uis_numerics.h -- the canonical float32 arithmetic of the UIS-RNN decode path.
 *
 * The reference (uisrnn/uisrnn.py:388-453, uisrnn/loss_func.py:19-41) delegates
 * its arithmetic to PyTorch CPU kernels whose summation order and libm are not
 * part of its contract.  This header fixes ONE order of operations that both
 * the gfx950 kernels (uisrnn_amd/csrc) and the CPU restatement (oracle/) follow,
 * built only from IEEE-754 single operations (add, mul, div, fma, rint), so the
 * two agree bit for bit; agreement with the reference itself is then a
 * tolerance statement checked by tests/golden.
 *
 * Compile with -ffp-contract=off on both sides: every fused multiply-add here
 * is an explicit fmaf().
 *
 * Plain C99 / HIP C++.  No dependencies.
 */
#ifndef UIS_NUMERICS_H_
#define UIS_NUMERICS_H_

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define UIS_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#define UIS_HD static inline
#endif

/* Bumped whenever the order of operations below changes; liboracle.so and
   libuisrnn_hip.so both export it and the tests require them to agree, so a
   stale binary on either side is caught instead of showing up as 1-ulp noise. */
#define UIS_NUMERICS_VERSION 3

/* Dense chains pad their contraction length to a multiple of this. */
#define UIS_KBLOCK 16

/* Every dense contraction is cut into this many K segments. */
#define UIS_KSPLIT 8

/*
 * Contraction order of every dense chain (GRU gates, mean head, input
 * projection):  out = (((c_0 + c_1) + c_2) + ... + c_7).
 * The K axis (zero padded to a multiple of 16) is nKb blocks of 16; segment s
 * owns blocks [s*q, min((s+1)*q, nKb)) with q = ceil(nKb / UIS_KSPLIT).  c_0
 * starts at the bias, c_s (s > 0) at +0.0f, an empty segment stays +0.0f and is
 * still added.  Inside a segment the blocks are walked in increasing order and
 * inside a block in the order
 *   0,4,8,12, 1,5,9,13, 2,6,10,14, 3,7,11,15
 * with one fmaf per element.  That is what a v_mfma_f32_16x16x4_f32 chain
 * computes when every lane fetches four consecutive k of its row with one
 * 16-byte load (register r of k-lane q holds k = 4q + r; MFMA number r sums
 * q = 0..3 in order); the eight segments are eight waves of one workgroup whose
 * partial tiles meet in LDS -- eight times the loads in flight and an eighth of
 * the serial stream per wave, which is what bounds a skinny GEMM on MI355X.
 */
UIS_HD int uis_kseg_blocks(int nKb) { return (nKb + UIS_KSPLIT - 1) / UIS_KSPLIT; }

UIS_HD int uis_korder(int i) { /* i in [0,16) -> k offset inside the block */
  return ((i & 3) << 2) | (i >> 2);
}

UIS_HD float uis_fma(float a, float b, float c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_fmaf(a, b, c);
#else
  return fmaf(a, b, c);
#endif
}

UIS_HD float uis_rint(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_rintf(x);
#else
  return rintf(x);
#endif
}

UIS_HD float uis_bits2f(uint32_t u) {
  union { uint32_t u; float f; } c;
  c.u = u;
  return c.f;
}

UIS_HD uint32_t uis_f2bits(float f) {
  union { uint32_t u; float f; } c;
  c.f = f;
  return c.u;
}

/* exp(x), |rel err| ~ 1e-7; clamps to [-87, 88] so the result stays normal. */
UIS_HD float uis_expf(float x) {
  if (x != x) return x;
  if (x > 88.0f) x = 88.0f;
  if (x < -87.0f) x = -87.0f;
  float n = uis_rint(x * 1.44269504088896341f);
  float r = uis_fma(n, -0.693145751953125f, x);          /* ln2 hi (exact in 12 bits) */
  r = uis_fma(n, -1.42860682030941723212e-6f, r);        /* ln2 lo */
  float p = 1.9841270e-4f;                                /* 1/5040 */
  p = uis_fma(p, r, 1.3888889e-3f);                       /* 1/720 */
  p = uis_fma(p, r, 8.3333333e-3f);                       /* 1/120 */
  p = uis_fma(p, r, 4.1666667e-2f);                       /* 1/24 */
  p = uis_fma(p, r, 1.6666667e-1f);                       /* 1/6 */
  p = uis_fma(p, r, 0.5f);
  p = uis_fma(p, r, 1.0f);
  p = uis_fma(p, r, 1.0f);
  int e = (int)n + 127;                                   /* in [2, 254] after the clamp */
  return p * uis_bits2f((uint32_t)e << 23);
}

/* logistic function, as torch.sigmoid in the GRU gates (uisrnn/uisrnn.py:47). */
UIS_HD float uis_sigmoidf(float x) {
  return 1.0f / (1.0f + uis_expf(-x));
}

/* tanh(x): expm1 series on |x| <= 0.5, exp form outside. */
UIS_HD float uis_tanhf(float x) {
  if (x != x) return x;
  float ax = x < 0.0f ? -x : x;
  if (ax <= 0.5f) {
    float y = x + x; /* |y| <= 1 */
    float p = 2.5052108e-8f;              /* 1/11! */
    p = uis_fma(p, y, 2.7557319e-7f);     /* 1/10! */
    p = uis_fma(p, y, 2.7557319e-6f);     /* 1/9!  */
    p = uis_fma(p, y, 2.4801587e-5f);     /* 1/8!  */
    p = uis_fma(p, y, 1.9841270e-4f);     /* 1/7!  */
    p = uis_fma(p, y, 1.3888889e-3f);     /* 1/6!  */
    p = uis_fma(p, y, 8.3333333e-3f);     /* 1/5!  */
    p = uis_fma(p, y, 4.1666667e-2f);     /* 1/4!  */
    p = uis_fma(p, y, 1.6666667e-1f);     /* 1/3!  */
    p = uis_fma(p, y, 0.5f);              /* 1/2!  */
    p = uis_fma(p, y, 1.0f);
    float em = p * y;                     /* expm1(2x) */
    return em / (em + 2.0f);
  }
  float t = uis_expf(-2.0f * ax);
  float v = (1.0f - t) / (1.0f + t);
  return x < 0.0f ? -v : v;
}

/*
 * One GRU unit (PyTorch gate order r|z|n; torch.gru called at
 * uisrnn/uisrnn.py:47).  gi_* = W_i* inp + b_i*, gh_* = W_h* h + b_h*.
 */
UIS_HD float uis_gru_unit(float gi_r, float gi_z, float gi_n,
                          float gh_r, float gh_z, float gh_n, float h) {
  float r = uis_sigmoidf(gi_r + gh_r);
  float z = uis_sigmoidf(gi_z + gh_z);
  float n = uis_tanhf(gi_n + r * gh_n);
  return (h - n) * z + n;
}

/*
 * Weighted squared error of uisrnn/loss_func.py:19-41 for one row:
 * term_d = ((a_d - b_d) * (a_d - b_d)) * w_d, summed by uis_mse_finish over the
 * canonical tree below, then the reference's mean * D * 1 / nnz with
 * nnz = ((a_0 - b_0)^2 != 0) (quirk: exact zero in dim 0 gives inf or nan).
 */
UIS_HD float uis_mse_term(float a, float b, float w) {
  float d = a - b;
  float s = d * d;
  return s * w;
}

UIS_HD float uis_mse_finish(float sum, float first_sq, int dim) {
  float fd = (float)dim;
  float mean = sum / fd;
  float v = (mean * fd) * 1.0f;
  float nnz = (first_sq != 0.0f) ? 1.0f : 0.0f;
  return v / nnz;
}

/*
 * Canonical summation order of the D weighted terms (version 3).  The reference promises none
 * (torch.mean over a CPU tensor, uisrnn/loss_func.py:33-41); this one is chosen so that the
 * producer of a cluster mean -- the linear_mean2 epilogue, which holds one 16-feature tile of a
 * row in 16 adjacent lanes -- can emit the tile's partial sum itself, and a select then adds a
 * handful of partial sums instead of re-reading the mean:
 *   1. tile sums: the terms are cut into tiles of UIS_MSE_TILE = 16 consecutive d (terms past D
 *      are +0.0f); inside a tile every 4 consecutive terms are added left to right,
 *      q_i = ((t[4i] + t[4i+1]) + t[4i+2]) + t[4i+3]  (one 16-byte load of a lane), and the tile is
 *      P = (q0 + q1) + (q2 + q3)  (an xor butterfly over four adjacent lanes);
 *   2. sixteen accumulators: A[p] = P[p] + P[p + 16] + P[p + 32] + ... left to right, +0.0f where
 *      there is no tile p;
 *   3. an xor butterfly over p with offsets 8, 4, 2, 1:
 *        (((A0+A8)+(A4+A12)) + ((A2+A10)+(A6+A14))) + (((A1+A9)+(A5+A13)) + ((A3+A11)+(A7+A15))).
 * (Which of the many legal orders: every reference-recorded fixture of tests/golden keeps its
 * labels under any of them; the WHOLE final beam of d32_lookahead3 -- hypotheses one float32 ulp
 * apart -- is reproduced by this one and not, e.g., by the plain adjacent-pair tree.)
 * Host-side helpers; the kernels implement the same order with DPP / shuffles / registers.
 */
#define UIS_MSE_TILE 16

UIS_HD float uis_mse_quad_sum(float t0, float t1, float t2, float t3) { return ((t0 + t1) + t2) + t3; }

UIS_HD float uis_mse_tile_sum(const float* t) { /* 16 terms */
  float q0 = uis_mse_quad_sum(t[0], t[1], t[2], t[3]), q1 = uis_mse_quad_sum(t[4], t[5], t[6], t[7]);
  float q2 = uis_mse_quad_sum(t[8], t[9], t[10], t[11]), q3 = uis_mse_quad_sum(t[12], t[13], t[14], t[15]);
  return (q0 + q1) + (q2 + q3);
}

UIS_HD float uis_mse_acc_sum(const float* A) { /* 16 accumulators */
  float a0 = A[0] + A[8], a1 = A[1] + A[9], a2 = A[2] + A[10], a3 = A[3] + A[11];
  float a4 = A[4] + A[12], a5 = A[5] + A[13], a6 = A[6] + A[14], a7 = A[7] + A[15];
  float b0 = a0 + a4, b1 = a1 + a5, b2 = a2 + a6, b3 = a3 + a7;
  float c0 = b0 + b2, c1 = b1 + b3;
  return c0 + c1;
}

#if !defined(__HIP_DEVICE_COMPILE__)
static inline float uis_tree_sum(const float* term, int dim) {
  float A[16];
  for (int p = 0; p < 16; ++p) A[p] = 0.0f;
  int ntile = (dim + UIS_MSE_TILE - 1) / UIS_MSE_TILE;
  for (int ft = 0; ft < ntile; ++ft) {
    float t[UIS_MSE_TILE];
    for (int j = 0; j < UIS_MSE_TILE; ++j) {
      int d = ft * UIS_MSE_TILE + j;
      t[j] = d < dim ? term[d] : 0.0f;
    }
    float P = uis_mse_tile_sum(t);
    A[ft & 15] = ft < 16 ? P : A[ft & 15] + P;
  }
  return uis_mse_acc_sum(A);
}
#endif

/*
 * Running "mean" update of uisrnn/uisrnn.py:425-429: n = frames assigned to the
 * cluster BEFORE this one.  Three separately rounded float ops, as torch does.
 */
UIS_HD float uis_mean_update(float old_mean, float m, int n) {
  float a = old_mean * (float)(n - 1);
  float b = a + m;
  return b / (float)n;
}

/*
 * Step loss: float32( float64(mse) - prior ), prior in float64
 * (uisrnn/uisrnn.py:415-420,444-446; numpy in-place subtract on a 0-d f32 array).
 */
UIS_HD float uis_step_loss(float mse, double prior) {
  return (float)((double)mse - prior);
}

/* Total order on scores for the prune: ascending value, then ascending index. */
UIS_HD uint32_t uis_score_key(float s) {
  if (s == 0.0f) s = 0.0f; /* -0 == +0 */
  uint32_t u = uis_f2bits(s);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

UIS_HD int uis_isfinite(float s) {
  return (uis_f2bits(s) & 0x7f800000u) != 0x7f800000u;
}

#endif /* UIS_NUMERICS_H_ */
