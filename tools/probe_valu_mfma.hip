// Probe (round 6): can the f32 VALU pipe carry row tiles NEXT TO the f32 MFMA pipe of the same SIMD, bit for bit?
//
// A 512-thread workgroup has two waves per SIMD.  Today both run v_mfma_f32_16x16x4_f32 chains and share the
// SIMD's matrix pipe (a row tile of the GRU stage: 384 MFMAs = 12288 cycles alone, twice that in a pair).  The
// f32 VALU rate equals the f32 MFMA rate (64 FLOP / clk / SIMD) and the two pipes are separate, so a wave that
// computes its row tile with v_fmac_f32 could run beside an MFMA wave at (nearly) full speed -- IF the VALU
// wave gets its operands without an instruction per value.  This probe does it with DPP: the four lanes of a quad
// hold four different k of one row (the MFMA B-operand layout within the quad), and `v_fmac_f32_dpp quad_perm:[q,q,q,q]`
// broadcasts lane q's value to the quad inside the multiply-add: acc[f][row] += W[f][k] * h[row][k] with ONE
// instruction per fma, in the canonical order of include/uis_numerics.h (0,4,8,12, 1,5,9,13, ...).
//
//   (1) bits: a 16-feature x 32-row x K tile by the VALU schedule == by the MFMA chain == by the host fmaf chain
//   (2) time: per-wave cycles of N tile-equivalents for {MFMA on 8 waves, MFMA on 4, VALU on 4, MFMA 4 + VALU 4}
//       with every CU busy (one workgroup per CU, 256 workgroups)
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I include tools/probe_valu_mfma.hip -o /tmp/probe_vm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
#include "uis_numerics.h"

#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(err_), __LINE__); exit(1);} } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// acc += dpp_quad_broadcast<Q>(h) * w   (one VALU instruction)
template <int Q>
__device__ __forceinline__ void fmac_qb(float& acc, float h, float w) {
  if constexpr (Q == 0) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(h), "v"(w));
  if constexpr (Q == 1) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(h), "v"(w));
  if constexpr (Q == 2) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(h), "v"(w));
  if constexpr (Q == 3) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(h), "v"(w));
}

// One 16-k block of the VALU schedule for NR rows of this lane: wreg[q][e] = W[f][16 kb + 4 q + e] (this lane's
// feature), hreg[i][e] = h[row_i][16 kb + 4 j + e] (j = lane & 3).  Canonical order: e outer, q inner.
template <int NR>
__device__ __forceinline__ void valu_block(const f32x4 (&wreg)[4], const f32x4 (&hreg)[NR], float (&acc)[NR]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
#pragma unroll
    for (int i = 0; i < NR; ++i) fmac_qb<0>(acc[i], hreg[i][e], wreg[0][e]);
#pragma unroll
    for (int i = 0; i < NR; ++i) fmac_qb<1>(acc[i], hreg[i][e], wreg[1][e]);
#pragma unroll
    for (int i = 0; i < NR; ++i) fmac_qb<2>(acc[i], hreg[i][e], wreg[2][e]);
#pragma unroll
    for (int i = 0; i < NR; ++i) fmac_qb<3>(acc[i], hreg[i][e], wreg[3][e]);
  }
}

// (1) bits.  W [16 features][K], H [32 rows][K], bias [16] -> out [32 rows][16 features]; one wave.
// lane: j = lane & 3, fq = (lane >> 2) & 3, rg = lane >> 4; feature f = 4 fq + j; rows rg + 4 i, i < 8.
__global__ void valu_tile(const float* W, const float* H, const float* bias, float* out, int K) {
  const int lane = threadIdx.x & 63, j = lane & 3, fq = (lane >> 2) & 3, rg = lane >> 4, f = 4 * fq + j;
  const int nKb = K / 16, per = uis_kseg_blocks(nKb);
  float total[8];
  for (int sgm = 0; sgm < UIS_KSPLIT; ++sgm) {
    float acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = sgm == 0 ? bias[f] : 0.0f;
    for (int kb = sgm * per; kb < (sgm + 1) * per && kb < nKb; ++kb) {
      f32x4 wreg[4], hreg[8];
      for (int q = 0; q < 4; ++q) wreg[q] = *reinterpret_cast<const f32x4*>(W + (size_t)f * K + 16 * kb + 4 * q);
      for (int i = 0; i < 8; ++i) hreg[i] = *reinterpret_cast<const f32x4*>(H + (size_t)(rg + 4 * i) * K + 16 * kb + 4 * j);
      valu_block<8>(wreg, hreg, acc);
    }
    for (int i = 0; i < 8; ++i) total[i] = sgm == 0 ? acc[i] : total[i] + acc[i];
  }
  for (int i = 0; i < 8; ++i) out[(size_t)(rg + 4 * i) * 16 + f] = total[i];
}

// the same tile by MFMA chains (rows in two 16-row halves), the order of the product kernels
__global__ void mfma_tile(const float* W, const float* H, const float* bias, float* out, int K) {
  const int lane = threadIdx.x & 63, m = lane & 15, q = lane >> 4;
  const int nKb = K / 16, per = uis_kseg_blocks(nKb);
  for (int half = 0; half < 2; ++half) {
    f32x4 total;
    for (int sgm = 0; sgm < UIS_KSPLIT; ++sgm) {
      f32x4 acc;
      for (int r = 0; r < 4; ++r) acc[r] = sgm == 0 ? bias[4 * q + r] : 0.0f;
      for (int kb = sgm * per; kb < (sgm + 1) * per && kb < nKb; ++kb) {
        f32x4 a = *reinterpret_cast<const f32x4*>(W + (size_t)m * K + 16 * kb + 4 * q);
        f32x4 b = *reinterpret_cast<const f32x4*>(H + (size_t)(16 * half + m) * K + 16 * kb + 4 * q);
        for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], b[r], acc, 0, 0, 0);
      }
      if (sgm == 0) total = acc;
      else for (int r = 0; r < 4; ++r) total[r] = total[r] + acc[r];
    }
    for (int r = 0; r < 4; ++r) out[(size_t)(16 * half + m) * 16 + 4 * q + r] = total[r];  // D: feature 4q + r, row m
  }
}

// (2) time.  mode bit 0: waves 0-3 run MFMA tiles; bit 1: waves 4-7 run MFMA tiles; bit 2: waves 4-7 run VALU tiles;
// bit 3: waves 0-3 run VALU tiles.  Operands live in registers (the question is the pipes, not the memory system): a "tile" is
// 3 gates x 32 k-blocks: MFMA 384 instructions (16 rows), VALU 3 x 32 x 16 x NR fmacs per lane for 4 NR rows.
// ticks[blockIdx][wave] = s_memtime cycles of `tiles` tile-equivalents (16 rows each).
template <int NR>
__global__ __launch_bounds__(512) void pipes(int mode, int tiles, const float* seed, float* sink, unsigned long long* ticks) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const bool do_mfma = (w < 4 && (mode & 1)) || (w >= 4 && (mode & 2));
  const bool do_valu = (w >= 4 && (mode & 4)) || (w < 4 && (mode & 8));
  float s0 = seed[lane], s1 = seed[64 + lane];
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  float res = 0.0f;
  if (do_mfma) {
    for (int tl = 0; tl < tiles; ++tl) {
      f32x4 acc[3] = {{s0, s0, s0, s0}, {s1, s1, s1, s1}, {s0, s1, s0, s1}};
      f32x4 a[3], b;
      for (int g = 0; g < 3; ++g) a[g] = f32x4{s0 + tl, s1 - g, s0 * 0.5f, s1 + g};
      b = f32x4{s1, s0, s1 * 0.25f, s0 - tl};
      for (int kb = 0; kb < 32; ++kb) {
        asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(b));  // (operands in registers: opaque, no instruction)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g][e], b[e], acc[g], 0, 0, 0);
      }
      for (int g = 0; g < 3; ++g) res += acc[g][0] + acc[g][1] + acc[g][2] + acc[g][3];
    }
  } else if (do_valu) {
    // a VALU "pass" covers 4 NR rows: tiles * 16 rows = tiles * 4 / NR passes (NR = 4: one pass per tile; NR = 8: one per two)
    const int passes = tiles * 4 / NR;
    for (int ps = 0; ps < passes; ++ps) {
      float acc[3][NR];
      for (int g = 0; g < 3; ++g) for (int i = 0; i < NR; ++i) acc[g][i] = s0 + g + i;
      f32x4 hreg[NR], wreg[4];
      for (int i = 0; i < NR; ++i) hreg[i] = f32x4{s1 + i, s0 - ps, s1 * 0.25f, s0 + i};
      for (int q = 0; q < 4; ++q) wreg[q] = f32x4{s0 + q, s1 - ps, s0 * 0.5f, s1 + q};
      for (int kb = 0; kb < 32; ++kb) {
#pragma unroll
        for (int i = 0; i < NR; ++i) asm volatile("" : "+v"(hreg[i]));
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          asm volatile("" : "+v"(wreg[0]), "+v"(wreg[1]), "+v"(wreg[2]), "+v"(wreg[3]));
          valu_block<NR>(wreg, hreg, acc[g]);
        }
      }
      for (int g = 0; g < 3; ++g) for (int i = 0; i < NR; ++i) res += acc[g][i];
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) ticks[blockIdx.x * 8 + w] = t1 - t0;
  if (res == 12345.678f) sink[threadIdx.x] = res;
}

int main() {
  // ---- (1) bits
  const int K = 512;
  std::vector<float> W(16 * K), H(32 * K), bias(16), o_valu(32 * 16), o_mfma(32 * 16), o_ref(32 * 16);
  srand(7);
  auto rnd = []() { return (float)rand() / RAND_MAX * 2.0f - 1.0f; };
  for (auto& v : W) v = rnd() * 0.2f;
  for (auto& v : H) v = rnd();
  for (auto& v : bias) v = rnd();
  H[5] = 1e-30f; W[5] = 1e-12f;  // a product in the denormal range
  float *dW, *dH, *db, *dO;
  CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&dH, H.size() * 4)); CK(hipMalloc(&db, 64)); CK(hipMalloc(&dO, o_valu.size() * 4));
  CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dH, H.data(), H.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, bias.data(), 64, hipMemcpyHostToDevice));
  valu_tile<<<1, 64>>>(dW, dH, db, dO, K);
  CK(hipMemcpy(o_valu.data(), dO, o_valu.size() * 4, hipMemcpyDeviceToHost));
  mfma_tile<<<1, 64>>>(dW, dH, db, dO, K);
  CK(hipMemcpy(o_mfma.data(), dO, o_mfma.size() * 4, hipMemcpyDeviceToHost));
  const int nKb = K / 16, per = uis_kseg_blocks(nKb);
  for (int r = 0; r < 32; ++r)
    for (int f = 0; f < 16; ++f) {
      float total = 0.0f;
      for (int sgm = 0; sgm < UIS_KSPLIT; ++sgm) {
        float acc = sgm == 0 ? bias[f] : 0.0f;
        for (int kb = sgm * per; kb < (sgm + 1) * per && kb < nKb; ++kb)
          for (int i = 0; i < 16; ++i) {
            const int k = 16 * kb + uis_korder(i);
            acc = fmaf(W[(size_t)f * K + k], H[(size_t)r * K + k], acc);
          }
        total = sgm == 0 ? acc : total + acc;
      }
      o_ref[r * 16 + f] = total;
    }
  int bad_vm = 0, bad_vr = 0, bad_mr = 0;
  for (size_t i = 0; i < o_ref.size(); ++i) {
    bad_vm += memcmp(&o_valu[i], &o_mfma[i], 4) != 0;
    bad_vr += memcmp(&o_valu[i], &o_ref[i], 4) != 0;
    bad_mr += memcmp(&o_mfma[i], &o_ref[i], 4) != 0;
  }
  printf("[bits] 32 rows x 16 features x K %d: VALU(dpp) vs MFMA %d differing, VALU vs host fmaf chain %d, MFMA vs host %d  (of %zu)\n",
         K, bad_vm, bad_vr, bad_mr, o_ref.size());

  // ---- (2) time
  float* dseed; float* dsink; unsigned long long* dticks;
  std::vector<float> seed(128);
  for (auto& v : seed) v = rnd();
  CK(hipMalloc(&dseed, 512)); CK(hipMalloc(&dsink, 4096)); CK(hipMalloc(&dticks, 256 * 8 * 8));
  CK(hipMemcpy(dseed, seed.data(), 512, hipMemcpyHostToDevice));
  const int tiles = 64;
  struct { int mode; const char* what; } modes[] = {
      {1, "MFMA on waves 0-3 only"}, {3, "MFMA on all 8 waves (today)"}, {4, "VALU on waves 4-7 only"}, {12, "VALU on all 8 waves"},
      {5, "MFMA on waves 0-3 + VALU on waves 4-7"}, {10, "VALU on waves 0-3 + MFMA on waves 4-7"}};
  for (int nr = 4; nr <= 8; nr += 4)
    for (auto& md : modes) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        if (nr == 4) pipes<4><<<256, 512>>>(md.mode, tiles, dseed, dsink, dticks);
        else pipes<8><<<256, 512>>>(md.mode, tiles, dseed, dsink, dticks);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0.0f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 0) continue;
        std::vector<unsigned long long> tk(256 * 8);
        CK(hipMemcpy(tk.data(), dticks, tk.size() * 8, hipMemcpyDeviceToHost));
        double lo = 0.0, hi = 0.0;
        for (int b = 0; b < 256; ++b) {
          for (int w = 0; w < 4; ++w) lo += (double)tk[b * 8 + w];
          for (int w = 4; w < 8; ++w) hi += (double)tk[b * 8 + w];
        }
        lo /= 1024.0 * tiles; hi /= 1024.0 * tiles;
        const int active = ((md.mode & 1) || (md.mode & 8) ? 4 : 0) + ((md.mode & 2) || (md.mode & 4) ? 4 : 0);
        printf("[time] NR %d  %-42s kernel %.3f ms = %.2f us per 16-row tile and SIMD-slot; cycles per tile: waves 0-3 %.0f, waves 4-7 %.0f; "
               "tiles per us and CU %.3f\n", nr, md.what, ms, 1e3 * ms / tiles, lo, hi, active * tiles / (1e3 * ms));
      }
    }
  return 0;
}
