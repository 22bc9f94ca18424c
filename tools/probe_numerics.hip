// Probe: does the gfx950 path reproduce include/uis_numerics.h bit for bit?
//  (1) v_mfma_f32_16x16x4_f32 fed by 16-byte row loads == the canonical fmaf chain
//  (2) uis_expf / uis_sigmoidf / uis_tanhf / division on device == host
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I include tools/probe_numerics.hip -o /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
#include "uis_numerics.h"

#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(err_), __LINE__); exit(1);} } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// C[16x16] = bias + A[16xK] * B[16xK]^T, one wave.
__global__ void mfma_chain(const float* A, const float* B, const float* bias, float* C, int K) {
  int lane = threadIdx.x & 63;
  int i = lane & 15, q = lane >> 4;
  f32x4 acc;
  // C/D layout: col = lane&15, row = (lane>>4)*4 + reg
  for (int r = 0; r < 4; ++r) acc[r] = bias[lane & 15];
  for (int kb = 0; kb < K; kb += 16) {
    f32x4 a = *reinterpret_cast<const f32x4*>(A + (size_t)i * K + kb + 4 * q);
    f32x4 b = *reinterpret_cast<const f32x4*>(B + (size_t)i * K + kb + 4 * q);
    for (int r = 0; r < 4; ++r)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], b[r], acc, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) C[(q * 4 + r) * 16 + (lane & 15)] = acc[r];
}

__global__ void math_probe(const float* x, float* e, float* s, float* t, float* dv, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  e[i] = uis_expf(x[i]);
  s[i] = uis_sigmoidf(x[i]);
  t[i] = uis_tanhf(x[i]);
  dv[i] = uis_mean_update(x[i], x[(i + 1) % n], (i % 37) + 1) + uis_mse_finish(x[i] * x[i], x[i], 255);
}

int main() {
  int K = 512;
  std::vector<float> A(16 * K), B(16 * K), bias(16), C(256), Cref(256);
  srand(1);
  auto rnd = []() { return (float)rand() / RAND_MAX * 2.0f - 1.0f; };
  for (auto& v : A) v = rnd();
  for (auto& v : B) v = rnd() * 0.1f;
  for (auto& v : bias) v = rnd();
  float *dA, *dB, *db, *dC;
  CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&db, 64)); CK(hipMalloc(&dC, 1024));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, bias.data(), 64, hipMemcpyHostToDevice));
  mfma_chain<<<1, 64>>>(dA, dB, db, dC, K);
  CK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
  // D[row][col] = sum_k Aop[row][k] * Bop[k][col]; Aop row index = lane&15 of the A operand.
  int bad = 0, bad_seq = 0;
  for (int row = 0; row < 16; ++row)
    for (int col = 0; col < 16; ++col) {
      float acc = bias[col], seq = bias[col];
      for (int kb = 0; kb < K; kb += 16)
        for (int i2 = 0; i2 < 16; ++i2) {
          int k = kb + uis_korder(i2);
          acc = fmaf(A[row * K + k], B[col * K + k], acc);
          seq = fmaf(A[row * K + kb + i2], B[col * K + kb + i2], seq);
        }
      Cref[row * 16 + col] = acc;
      if (memcmp(&acc, &C[row * 16 + col], 4)) ++bad;
      if (memcmp(&seq, &C[row * 16 + col], 4)) ++bad_seq;
    }
  printf("mfma_chain: %d / 256 mismatches vs canonical korder chain (sequential-k chain: %d mismatches)\n", bad, bad_seq);
  if (bad) printf("  e.g. C[0]=%.9g ref=%.9g\n", C[0], Cref[0]);

  int n = 1 << 20;
  std::vector<float> x(n), e(n), s(n), t(n), dv(n);
  for (int i = 0; i < n; ++i) {
    float u = rnd();
    int m = i & 7;
    x[i] = m == 0 ? u * 100.f : m == 1 ? u * 10.f : m == 2 ? u * 1e-3f : m == 3 ? u * 0.5f : u * 4.f;
  }
  x[0] = 0.f; x[1] = -0.f; x[2] = 1e-30f; x[3] = 88.5f; x[4] = -100.f; x[5] = 0.5f; x[6] = -0.5f; x[7] = INFINITY; x[8] = -INFINITY; x[9] = NAN;
  float *dx, *de, *ds, *dt, *dd;
  CK(hipMalloc(&dx, n * 4)); CK(hipMalloc(&de, n * 4)); CK(hipMalloc(&ds, n * 4)); CK(hipMalloc(&dt, n * 4)); CK(hipMalloc(&dd, n * 4));
  CK(hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice));
  math_probe<<<n / 256, 256>>>(dx, de, ds, dt, dd, n);
  CK(hipMemcpy(e.data(), de, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(t.data(), dt, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(dv.data(), dd, n * 4, hipMemcpyDeviceToHost));
  int be = 0, bs = 0, bt = 0, bd = 0;
  double maxe = 0, maxt = 0, maxs = 0;
  for (int i = 0; i < n; ++i) {
    float he = uis_expf(x[i]), hs = uis_sigmoidf(x[i]), ht = uis_tanhf(x[i]);
    float hd = uis_mean_update(x[i], x[(i + 1) % n], (i % 37) + 1) + uis_mse_finish(x[i] * x[i], x[i], 255);
    if (memcmp(&he, &e[i], 4)) ++be;
    if (memcmp(&hs, &s[i], 4)) ++bs;
    if (memcmp(&ht, &t[i], 4)) ++bt;
    if (memcmp(&hd, &dv[i], 4) && !(hd != hd && dv[i] != dv[i])) ++bd;
    if (std::isfinite(x[i]) && fabsf(x[i]) < 80) {
      double re = exp((double)x[i]); maxe = fmax(maxe, fabs(he - re) / re);
      double rt = tanh((double)x[i]); if (rt != 0) maxt = fmax(maxt, fabs(ht - rt) / fabs(rt));
      double rs = 1.0 / (1.0 + exp(-(double)x[i])); maxs = fmax(maxs, fabs(hs - rs) / rs);
    }
  }
  printf("device-vs-host bit mismatches over %d inputs: exp %d sigmoid %d tanh %d div/mean %d\n", n, be, bs, bt, bd);
  printf("accuracy vs double libm (max rel err): exp %.3g sigmoid %.3g tanh %.3g\n", maxe, maxs, maxt);
  return (bad || be || bs || bt || bd) ? 1 : 0;
}
