// uis_kernels.hip -- gfx950 kernels of the UIS-RNN beam-search decode.
//
// Reference behaviour being reproduced (google/uis-rnn):
//   k_dense_* : CoreRNN.forward                uisrnn/uisrnn.py:45-52
//   k_mse0    : weighted_mse_loss vs m0        uisrnn/loss_func.py:19-41, uisrnn.py:440-443
//   k_select  : _calculate_score + prune + the bookkeeping half of _update_beam_state
//                                              uisrnn/uisrnn.py:388-453,455-477,534-559
//   k_backtrace : trace[-N:] of beam_set[0]    uisrnn/uisrnn.py:561
// Arithmetic order: include/uis_numerics.h (bit-identical to oracle/uis_oracle.c).
//
// Written for wave64 / MFMA f32 16x16x4 / 8 XCDs; no other target.
#include "uis_kernels.h"
#include "uis_numerics.h"


// ------------------------------------------------------------------ helpers

// The canonical order of the weighted-MSE sum (include/uis_numerics.h, version 3) on the device.
// dpp_perm<CTRL>: this lane's copy of another lane's value, DPP control CTRL (every lane is a
// valid source for the permutations used here).
template <int CTRL>
__device__ __forceinline__ float dpp_perm(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// the quad sum of one 16-byte load's worth of terms
__device__ __forceinline__ float mse_quad(const f32x4& a, const f32x4& b, const f32x4& w) {
  return uis_mse_quad_sum(uis_mse_term(a[0], b[0], w[0]), uis_mse_term(a[1], b[1], w[1]), uis_mse_term(a[2], b[2], w[2]),
                          uis_mse_term(a[3], b[3], w[3]));
}
// q = this lane's quad sum; a tile's four quads sit in four adjacent lanes: (q0 + q1) + (q2 + q3), in all four
__device__ __forceinline__ float mse_tile_of_lanes4(float q) {
  q = q + dpp_perm<0xB1>(q);  // quad_perm [1,0,3,2]: lane ^ 1
  q = q + dpp_perm<0x4E>(q);  // quad_perm [2,3,0,1]: lane ^ 2
  return q;
}
// Sixteen lanes per row: lane p (= lane & 15) holds the 16-byte chunks d = 256 q + 4 (p + 16 k),
// k = 0..3, of 256-float block q; A[k] is this lane's copy of accumulator (p >> 2) + 4 k.
// Chunks past the padded dimension must arrive zero-filled (their terms are +0).
__device__ __forceinline__ void mse16_block(const f32x4 (&mv)[4], const f32x4 (&xv)[4], const f32x4 (&wv)[4], float (&A)[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) A[k] = A[k] + mse_tile_of_lanes4(mse_quad(mv[k], xv[k], wv[k]));
}
__device__ __forceinline__ float mse16_total(const float (&A)[4]) {  // the butterfly over the accumulators: 8, 4, 2, 1
  float r = (A[0] + A[2]) + (A[1] + A[3]);  // accumulator index ^ 8, then ^ 4: registers
  r = r + dpp_perm<0x128>(r);               // ^ 2: row_ror:8 = lane ^ 8
  r = r + dpp_perm<0x141>(r);               // ^ 1: row_half_mirror = the neighbouring quad (the value is quad-uniform)
  return r;
}

// Weighted MSE of one row against the frame staged in LDS, canonical order.
// All 64 lanes of the calling wave participate; every lane returns the value.
// (lane l holds the chunk d = 256 q + 4 l: quad l & 3 of accumulator l >> 2)
__device__ __forceinline__ float wave_weighted_mse(const float* __restrict__ mean,
                                                   const float* sx, const float* swgt,
                                                   int Dp, int D, int lane) {
  float acc = 0.0f;
  for (int base0 = 0; base0 < Dp; base0 += 256) {
    const int base = base0 + 4 * lane;
    f32x4 m = {0.0f, 0.0f, 0.0f, 0.0f}, x = m, w = m;
    if (base < Dp) {
      m = *reinterpret_cast<const f32x4*>(mean + base);
      x = *reinterpret_cast<const f32x4*>(sx + base);
      w = *reinterpret_cast<const f32x4*>(swgt + base);
    }
    acc = acc + mse_tile_of_lanes4(mse_quad(m, x, w));
  }
#pragma unroll
  for (int off = 32; off >= 4; off >>= 1) acc = acc + __shfl_xor(acc, off, 64);  // accumulator index ^ 8, 4, 2, 1
  float d0 = mean[0] - sx[0];
  return uis_mse_finish(acc, d0 * d0, D);
}

// -------------------------------------------------------------- dense chains
//
// out[row][f] = bias[f] + sum_k W[f][k] * in[row][k] as v_mfma_f32_16x16x4_f32 chains in
// the canonical order of uis_numerics.h (UIS_KSPLIT segment chains combined left to right).
// A operand = weights (16 features x 4 k), B operand = 16 rows.  Lane l holds, for its
// row (l & 15), features 4*(l>>4) .. +3 of the tile.  K is walked in blocks of 16 with one
// 16-byte load per operand per lane; the weights are pre-tiled on the host so that a
// wave's A load is one contiguous KiB.
//
// Two schedules of the same arithmetic:
//   splitk_tile  -- per-step kernels (few hundred rows: a skinny GEMM whose cost is the
//                   serial operand stream of a wave).  One 512-thread workgroup per
//                   16-row x 16-feature tile; wave w owns K segment w, so a tile's operand
//                   stream is cut in 8 and 8x as many loads are in flight; the partial
//                   tiles meet in LDS and thread t < 256 combines element (row t>>4,
//                   feature t&15) in segment order.
//   fullk_tile   -- input projection (tens of thousands of rows, once per decode): one
//                   wave walks all segments of its tile and combines on the fly.

#define UIS_STAGE 4   // most k-blocks fetched per pipeline stage

// NA A-operand streams (feature tiles x gates) against NB B-operand streams (row tiles):
// NA*NB accumulators, NA+NB 16-byte loads per lane per k-block, STG k-blocks per stage.
template <int NA, int NB, int STG>
__device__ __forceinline__ void chain_blocks_stg(const f32x4* const (&wp)[NA], const f32x4* const (&bp)[NB], int kb0,
                                                 int kb1, f32x4 (&acc)[NB][NA]) {
  for (int kb = kb0; kb < kb1; kb += STG) {
    f32x4 a[STG][NA], b[STG][NB];
#pragma unroll
    for (int u = 0; u < STG; ++u) {
      const int kk = kb + u < kb1 ? kb + u : kb1 - 1;
#pragma unroll
      for (int g = 0; g < NA; ++g) a[u][g] = wp[g][(size_t)kk * 64];
#pragma unroll
      for (int r = 0; r < NB; ++r) b[u][r] = bp[r][(size_t)kk * 4];
    }
#pragma unroll
    for (int u = 0; u < STG; ++u) {
      if (kb + u < kb1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int r = 0; r < NB; ++r) {
#pragma unroll
            for (int g = 0; g < NA; ++g)
              acc[r][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][g][e], b[u][r][e], acc[r][g], 0, 0, 0);
          }
        }
      }
    }
  }
}
// stage depth = the segment length when that is short (a deeper stage would only re-request
// the segment's last block): 1, 2 or UIS_STAGE k-blocks
template <int NA, int NB>
__device__ __forceinline__ void chain_blocks(const f32x4* const (&wp)[NA], const f32x4* const (&bp)[NB], int kb0,
                                             int kb1, f32x4 (&acc)[NB][NA]) {
  const int len = kb1 - kb0;
  if (len >= UIS_STAGE) chain_blocks_stg<NA, NB, UIS_STAGE>(wp, bp, kb0, kb1, acc);
  else if (len >= 2) chain_blocks_stg<NA, NB, 2>(wp, bp, kb0, kb1, acc);
  else chain_blocks_stg<NA, NB, 1>(wp, bp, kb0, kb1, acc);
}

// Split-K schedule of a workgroup tile of R row tiles x C feature tiles x NG gates.
// A stream index a = c*NG + g.  bias points at feature tile `tile0`'s 16 biases of gate 0;
// tile c is 16 floats further, gate g gate_stride floats further.  inrow[r] is this lane's
// row of row tile r.  spart: LDS [UIS_KSPLIT][R][C*NG][256].  Ends with the barrier;
// afterwards splitk_combine() returns element (row t>>4, feature t&15) of a sub-tile.
template <int NG, int R, int C>
__device__ __forceinline__ void splitk_tile(const float* __restrict__ Wt, int tiles_per_gate, int tile0, int nKb,
                                            const float* const (&inrow)[R], const float* __restrict__ bias,
                                            int gate_stride, float* spart) {
  constexpr int NA = C * NG;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int q = lane >> 4;
  const int per = uis_kseg_blocks(nKb);
  const int kb0 = w * per;
  const int kb1 = kb0 + per < nKb ? kb0 + per : nKb;
  const f32x4* wp[NA];
  const f32x4* bp[R];
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int g = 0; g < NG; ++g)
      wp[c * NG + g] =
          reinterpret_cast<const f32x4*>(Wt) + ((size_t)(g * tiles_per_gate + tile0 + c) * nKb) * 64 + lane;
#pragma unroll
  for (int r = 0; r < R; ++r) bp[r] = reinterpret_cast<const f32x4*>(inrow[r]) + q;
  f32x4 acc[R][NA];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int g = 0; g < NG; ++g)
        acc[r][c * NG + g] =
            w == 0 ? *reinterpret_cast<const f32x4*>(bias + (size_t)g * gate_stride + c * 16 + 4 * q)
                   : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  if (kb0 < kb1) chain_blocks_stg<NA, R, UIS_STAGE>(wp, bp, kb0, kb1, acc);  // per-step kernels: one code path
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int a = 0; a < NA; ++a)
      *reinterpret_cast<f32x4*>(spart + ((size_t)((w * R + r) * NA + a) * 256) + (lane & 15) * 16 + 4 * q) =
          acc[r][a];
  __syncthreads();
}

// element (row t>>4, feature t&15) of row tile r, A stream a, combined in segment order
template <int R, int NA>
__device__ __forceinline__ float splitk_combine(const float* spart, int r, int a, int t) {
  float v = spart[(size_t)(r * NA + a) * 256 + t];
#pragma unroll
  for (int sgm = 1; sgm < UIS_KSPLIT; ++sgm) v = v + spart[(size_t)((sgm * R + r) * NA + a) * 256 + t];
  return v;
}

// Full-K schedule: one wave, all segments, combined on the fly.
__device__ __forceinline__ f32x4 fullk_tile(const float* __restrict__ Wt, int tile, int nKb,
                                            const float* __restrict__ inrow, const float* __restrict__ bias) {
  const int lane = threadIdx.x & 63;
  const int q = lane >> 4;
  const int per = uis_kseg_blocks(nKb);
  const f32x4* wp[1] = {reinterpret_cast<const f32x4*>(Wt) + ((size_t)tile * nKb) * 64 + lane};
  const f32x4* bp[1] = {reinterpret_cast<const f32x4*>(inrow) + q};
  f32x4 total = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 1
  for (int sgm = 0; sgm < UIS_KSPLIT; ++sgm) {
    const int kb0 = sgm * per;
    const int kb1 = kb0 + per < nKb ? kb0 + per : nKb;
    f32x4 acc[1][1];
    acc[0][0] = sgm == 0 ? *reinterpret_cast<const f32x4*>(bias + 4 * q) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (kb0 < kb1) chain_blocks<1, 1>(wp, bp, kb0, kb1, acc);
    if (sgm == 0) total = acc[0][0];
    else {
#pragma unroll
      for (int i = 0; i < 4; ++i) total[i] = total[i] + acc[0][0][i];
    }
  }
  return total;
}

// Workgroup -> (row tile, feature tile) with XCD affinity: workgroup b runs on XCD b % 8
// (observed dispatch order, used for speed only), so each XCD gets a fixed subset of the
// feature tiles and its L2 streams only that slice of the weights.
__host__ __device__ inline int dense_grid_blocks(int nrt, int nft) {
  if (nft < 8 && 8 % nft == 0) { const int share = 8 / nft; return ((nrt + share - 1) / share) * 8; }
  return nrt * nft;
}
__device__ __forceinline__ void dense_block_map(int b, int nrt, int nft, int& rt, int& ft) {
  if (nft >= 8 && nft % 8 == 0) {
    const int xcd = b & 7, slot = b >> 3, per = nft >> 3;
    ft = xcd + 8 * (slot % per);
    rt = slot / per;
  } else if (nft < 8 && 8 % nft == 0) {
    const int xcd = b & 7, slot = b >> 3, share = 8 / nft;
    ft = xcd % nft;
    rt = (xcd / nft) + share * slot;
  } else {
    ft = b % nft;
    rt = b / nft;
  }
}

// gi0[frame][G] = b_ih0 + W_ih0 x[frame]   for every frame of the packed stream.
__global__ __launch_bounds__(256) void k_dense_input_proj(DevModel m, const float* __restrict__ x,
                                                          float* __restrict__ gi0, long nframes) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ntiles = m.G / 16;
  const int tile = blockIdx.y * 4 + wave;
  if (tile >= ntiles) return;
  const long row0 = (long)blockIdx.x * 16;
  if (row0 >= nframes) return;
  long row = row0 + (lane & 15);
  const bool valid = row < nframes;
  if (!valid) row = nframes - 1;
  const int f = tile * 16 + (lane >> 4) * 4;
  const f32x4 v = fullk_tile(m.wih[0], tile, m.Dp / 16, x + (size_t)row * m.Dp, m.bih[0] + tile * 16);
  if (valid) *reinterpret_cast<f32x4*>(gi0 + (size_t)row * m.G + f) = v;
}

// The same arithmetic for many frames (a whole decode's input): a wave owns 2 row tiles x 4 feature
// tiles, so the rows are fetched once for four feature tiles and the weights once for two row
// tiles -- the one-tile kernel above asks L2 for 32 KB per 16 x 16 outputs and is bound by that.
__global__ __launch_bounds__(256) void k_dense_input_proj_wide(DevModel m, const float* __restrict__ x,
                                                               float* __restrict__ gi0, long nframes) {
  constexpr int NA = 4, NB = 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4;
  const int ntiles = m.G / 16, nKb = m.Dp / 16;
  const int tile0 = (blockIdx.y * 4 + wave) * NA;
  if (tile0 >= ntiles) return;
  const long row0 = (long)blockIdx.x * (16 * NB);
  if (row0 >= nframes) return;
  long rows[NB];
  bool valid[NB];
  const f32x4* bp[NB];
  const f32x4* wp[NA];
  int tiles[NA];
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    rows[r] = row0 + 16 * r + (lane & 15);
    valid[r] = rows[r] < nframes;
    if (!valid[r]) rows[r] = nframes - 1;
    bp[r] = reinterpret_cast<const f32x4*>(x + (size_t)rows[r] * m.Dp) + q;
  }
#pragma unroll
  for (int g = 0; g < NA; ++g) {
    tiles[g] = tile0 + g < ntiles ? tile0 + g : ntiles - 1;  // (a tail group recomputes its last tile, not stored twice)
    wp[g] = reinterpret_cast<const f32x4*>(m.wih[0]) + ((size_t)tiles[g] * nKb) * 64 + lane;
  }
  const int per = uis_kseg_blocks(nKb);
  f32x4 total[NB][NA];
#pragma unroll 1
  for (int sgm = 0; sgm < UIS_KSPLIT; ++sgm) {
    const int kb0 = sgm * per;
    const int kb1 = kb0 + per < nKb ? kb0 + per : nKb;
    f32x4 acc[NB][NA];
#pragma unroll
    for (int r = 0; r < NB; ++r) {
#pragma unroll
      for (int g = 0; g < NA; ++g)
        acc[r][g] = sgm == 0 ? *reinterpret_cast<const f32x4*>(m.bih[0] + tiles[g] * 16 + 4 * q) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    if (kb0 < kb1) chain_blocks<NA, NB>(wp, bp, kb0, kb1, acc);
#pragma unroll
    for (int r = 0; r < NB; ++r) {
#pragma unroll
      for (int g = 0; g < NA; ++g) {
        if (sgm == 0) total[r][g] = acc[r][g];
        else {
#pragma unroll
          for (int i = 0; i < 4; ++i) total[r][g][i] = total[r][g][i] + acc[r][g][i];
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < NB; ++r) {
#pragma unroll
    for (int g = 0; g < NA; ++g)
      if (valid[r] && tile0 + g < ntiles)
        *reinterpret_cast<f32x4*>(gi0 + (size_t)rows[r] * m.G + (tile0 + g) * 16 + q * 4) = total[r][g];
  }
}

// The same tiles with the K walk software-pipelined, for observation_dim 128 * PER (PER k-blocks per
// K segment, PER = 1, 2, 4): the operands of the next stage (two k-blocks; one when PER = 1) are
// requested before the MFMAs of the current one, everything unrolled, two register sets in turn.
// Without it a segment's loads wait for the previous segment's MFMAs and the MFMAs for the loads.
// Same order of operations per accumulator as chain_blocks (k-blocks ascending, e = 0..3).
// (batch_stride: rows between the batches of `nframes` rows that gridDim.z counts -- a time slice [t0, t1) of every
// utterance of a list of equal-length utterances; one batch and 0 otherwise)
// ... or, batch_tab given, batch z = rows [batch_tab[2 z], + batch_tab[2 z + 1]) of x / gi0: a time slice of a list of
// utterances of any lengths (nframes then only sizes the grid)
template <int PER>
__global__ __launch_bounds__(256) void k_dense_input_proj_pipe(DevModel m, const float* __restrict__ x,
                                                               float* __restrict__ gi0, long nframes, long batch_stride,
                                                               const long* __restrict__ batch_tab) {
  constexpr int NA = 4, NB = 2;
  constexpr int STG = PER >= 2 ? 2 : 1, SPS = PER / STG, NS = UIS_KSPLIT * SPS, NKB = UIS_KSPLIT * PER;
  if (batch_tab) {
    x += (size_t)batch_tab[2 * blockIdx.z] * (NKB * 16);
    gi0 += (size_t)batch_tab[2 * blockIdx.z] * m.G;
    nframes = batch_tab[2 * blockIdx.z + 1];
  } else {
    x += (size_t)blockIdx.z * (size_t)batch_stride * (NKB * 16);
    gi0 += (size_t)blockIdx.z * (size_t)batch_stride * m.G;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4;
  const int ntiles = m.G / 16;
  const int tile0 = (blockIdx.y * 4 + wave) * NA;
  if (tile0 >= ntiles) return;
  const long row0 = (long)blockIdx.x * (16 * NB);
  if (row0 >= nframes) return;
  long rows[NB];
  bool valid[NB];
  const f32x4* bp[NB];
  const f32x4* wp[NA];
  int tiles[NA];
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    rows[r] = row0 + 16 * r + (lane & 15);
    valid[r] = rows[r] < nframes;
    if (!valid[r]) rows[r] = nframes - 1;
    bp[r] = reinterpret_cast<const f32x4*>(x + (size_t)rows[r] * (NKB * 16)) + q;
  }
#pragma unroll
  for (int g = 0; g < NA; ++g) {
    tiles[g] = tile0 + g < ntiles ? tile0 + g : ntiles - 1;
    wp[g] = reinterpret_cast<const f32x4*>(m.wih[0]) + ((size_t)tiles[g] * NKB) * 64 + lane;
  }
  f32x4 a[2][STG][NA], b[2][STG][NB], acc[NB][NA], total[NB][NA];
#pragma unroll
  for (int u = 0; u < STG; ++u) {
#pragma unroll
    for (int g = 0; g < NA; ++g) a[0][u][g] = wp[g][(size_t)u * 64];
#pragma unroll
    for (int r = 0; r < NB; ++r) b[0][u][r] = bp[r][(size_t)u * 4];
  }
#pragma unroll
  for (int st = 0; st < NS; ++st) {
    const int cur = st & 1, nxt = cur ^ 1;
    if (st + 1 < NS) {
#pragma unroll
      for (int u = 0; u < STG; ++u) {
        const int kk = (st + 1) * STG + u;
#pragma unroll
        for (int g = 0; g < NA; ++g) a[nxt][u][g] = wp[g][(size_t)kk * 64];
#pragma unroll
        for (int r = 0; r < NB; ++r) b[nxt][u][r] = bp[r][(size_t)kk * 4];
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // the requests go out before this stage's MFMAs, not in the middle of them
    if (st % SPS == 0) {  // a K segment starts: bias in front of the first, zeros otherwise
#pragma unroll
      for (int r = 0; r < NB; ++r) {
#pragma unroll
        for (int g = 0; g < NA; ++g)
          acc[r][g] = st == 0 ? *reinterpret_cast<const f32x4*>(m.bih[0] + tiles[g] * 16 + 4 * q) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      }
    }
#pragma unroll
    for (int u = 0; u < STG; ++u) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int r = 0; r < NB; ++r) {
#pragma unroll
          for (int g = 0; g < NA; ++g)
            acc[r][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][u][g][e], b[cur][u][r][e], acc[r][g], 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // (or the compiler hoists every load to the top: 256 AGPRs, one wave per SIMD)
    if (st % SPS == SPS - 1) {  // the segment is complete: segments are combined left to right
#pragma unroll
      for (int r = 0; r < NB; ++r) {
#pragma unroll
        for (int g = 0; g < NA; ++g) {
          if (st == SPS - 1) total[r][g] = acc[r][g];
          else {
#pragma unroll
            for (int i = 0; i < 4; ++i) total[r][g][i] = total[r][g][i] + acc[r][g][i];
          }
          // the sum is wanted HERE: left alone the compiler sinks all seven additions to the end
          // of the kernel and keeps every segment's accumulators alive until then (396 registers)
          asm volatile("" : "+v"(total[r][g]));
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < NB; ++r) {
#pragma unroll
    for (int g = 0; g < NA; ++g)
      if (valid[r] && tile0 + g < ntiles)
        *reinterpret_cast<f32x4*>(gi0 + (size_t)rows[r] * m.G + (tile0 + g) * 16 + q * 4) = total[r][g];
  }
}

// Common prologue of the per-step kernels.  A workgroup owns RT consecutive row tiles and CT
// consecutive feature tiles; which ones follows from blockIdx alone.  The row count (written
// by this step's select) and the row descriptors are fetched together, so only ONE global
// round trip precedes the operand stream.  Rows at or beyond the count hold descriptors of
// an earlier step (or zeros): they still name valid slots, so their lanes stream valid
// memory and their results are simply not stored.
struct StepTile {
  int rt0, ft0, row0, nrows;
  bool active;
};
template <int RT, int CT>
__device__ __forceinline__ StepTile step_tile(const DecodeState& st, int par, int nft, RnnRow (&lane_row)[RT],
                                              int block = -1) {
  StepTile t;
  const int max_rt = (st.max_rows + 15) >> 4;
  const int n_rg = (max_rt + RT - 1) / RT, n_fg = (nft + CT - 1) / CT;
  int rg, fg;
  dense_block_map(block < 0 ? (int)blockIdx.x : block, n_rg, n_fg, rg, fg);
  t.rt0 = rg * RT; t.ft0 = fg * CT;
  t.row0 = t.rt0 * 16;
  const bool in_grid = rg < n_rg && fg < n_fg;
#pragma unroll
  for (int r = 0; r < RT; ++r) lane_row[r] = st.rows[in_grid ? t.row0 + 16 * r + (threadIdx.x & 15) : 0];
  t.nrows = st.nrows[par];
  t.active = in_grid && t.row0 < t.nrows;
  return t;
}
__host__ __device__ inline int step_grid_blocks(int max_rows, int nft, int RT, int CT) {
  const int max_rt = (max_rows + 15) >> 4;
  return dense_grid_blocks((max_rt + RT - 1) / RT, (nft + CT - 1) / CT);
}

// Tile shapes: a workgroup owns RT row tiles (x CT feature tiles for the head).  RT = CT = 1
// keeps the most workgroups in flight and wins while a launch is latency bound (up to a few
// hundred rnn rows); RT = CT = 2 re-uses operands out of registers and wins once the L2->CU
// stream is the limit (measured crossover ~1000 rows).  The host picks per decode.

__device__ __forceinline__ const float* hid_ptr(const DevModel& m, const DecodeState& st, const RnnRow& r, int slot,
                                                int layer) {
  return st.pool_hid + ((size_t)r.utt * st.S + slot) * m.depth * m.Hp + (size_t)layer * m.Hp;
}

// Input-side gates of GRU layer `layer` >= 1: gi_up[row][G] = b_ih + W_ih h'_{layer-1}
__global__ __launch_bounds__(512) void k_dense_upper_in(DevModel m, DecodeState st, int par, int layer) {
  __shared__ __attribute__((aligned(16))) float spart[UIS_KSPLIT * 256];
  RnnRow rb[1];
  const StepTile tl = step_tile<1, 1>(st, par, m.G / 16, rb);
  if (!tl.active) return;
  const int t = threadIdx.x;
  const float* in[1] = {hid_ptr(m, st, rb[0], rb[0].dst, layer - 1)};
  splitk_tile<1, 1, 1>(m.wih[layer], 0, tl.ft0, m.Hp / 16, in, m.bih[layer] + tl.ft0 * 16, 0, spart);
  if (t >= 256) return;
  const int row = tl.row0 + (t >> 4);
  if (row >= tl.nrows) return;
  st.gi_up[(size_t)row * m.G + tl.ft0 * 16 + (t & 15)] = splitk_combine<1, 1>(spart, 0, 0, t);
}

// GRU layer: gh = b_hh + W_hh h_src (three gate chains per unit), gates, h' -> dst slot.
template <int RT>
__global__ __launch_bounds__(512) void k_dense_gru(DevModel m, DecodeState st, int par, int layer) {
  __shared__ __attribute__((aligned(16))) float spart[UIS_KSPLIT * RT * 3 * 256];
  const int nft = m.Hp / 16;
  RnnRow rb[RT];
  const StepTile tl = step_tile<RT, 1>(st, par, nft, rb);
  if (!tl.active) return;
  const int t = threadIdx.x;
  // epilogue operands: thread t owns element (row t>>4 of row tile (t>>8) + 2k, unit t&15);
  // fetched now, used after the chains
  const int j = tl.ft0 * 16 + (t & 15);
  constexpr int EPT = (RT + 1) / 2;  // elements per thread (512 threads cover 2 row tiles)
  RnnRow re[EPT];
  float gir[EPT], giz[EPT], gin[EPT], hprev[EPT];
  bool ework[EPT];
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int r = (t >> 8) + 2 * k;
    const int erow = tl.row0 + 16 * r + ((t & 255) >> 4);
    ework[k] = r < RT && erow < tl.nrows;
    gir[k] = giz[k] = gin[k] = hprev[k] = 0.0f;
    re[k] = RnnRow{};
    if (ework[k]) {
      re[k] = st.rows[erow];
      const float* gi = layer == 0 ? st.gi0 + (size_t)re[k].frame * m.G : st.gi_up + (size_t)erow * m.G;
      const float* hs = re[k].src >= 0 ? hid_ptr(m, st, re[k], re[k].src, layer) : m.h1 + (size_t)layer * m.Hp;
      gir[k] = gi[j]; giz[k] = gi[m.Hp + j]; gin[k] = gi[2 * m.Hp + j]; hprev[k] = hs[j];
    }
  }
  const float* hsrc[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r)
    hsrc[r] = rb[r].src >= 0 ? hid_ptr(m, st, rb[r], rb[r].src, layer) : m.h1 + (size_t)layer * m.Hp;
  splitk_tile<3, RT, 1>(m.whh[layer], nft, tl.ft0, m.Hp / 16, hsrc, m.bhh[layer] + tl.ft0 * 16, m.Hp, spart);
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    if (!ework[k]) continue;
    const int r = (t >> 8) + 2 * k, e = t & 255;
    const float ghr = splitk_combine<RT, 3>(spart, r, 0, e);
    const float ghz = splitk_combine<RT, 3>(spart, r, 1, e);
    const float ghn = splitk_combine<RT, 3>(spart, r, 2, e);
    const float out = j < m.H ? uis_gru_unit(gir[k], giz[k], gin[k], ghr, ghz, ghn, hprev[k]) : 0.0f;
    const_cast<float*>(hid_ptr(m, st, re[k], re[k].dst, layer))[j] = out;
  }
}

// a1[row] = relu(b1 + W1 h'_top)
template <int RT, int CT>
__global__ __launch_bounds__(512) void k_dense_head1(DevModel m, DecodeState st, int par) {
  __shared__ __attribute__((aligned(16))) float spart[UIS_KSPLIT * RT * CT * 256];
  const int nft = m.Hp / 16;
  RnnRow rb[RT];
  const StepTile tl = step_tile<RT, CT>(st, par, nft, rb);
  if (!tl.active) return;
  const int t = threadIdx.x;
  const float* in[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) in[r] = hid_ptr(m, st, rb[r], rb[r].dst, m.depth - 1);
  // a feature-tile pair may stick out past the last tile: stream tile nft-1 twice, store once
  const int ft0 = tl.ft0 + CT <= nft ? tl.ft0 : nft - CT < 0 ? 0 : nft - CT;
  if (nft < CT) {  // fewer feature tiles than a workgroup covers (tiny models): one at a time
    for (int c = 0; c < nft; ++c) {
      splitk_tile<1, RT, 1>(m.w1, 0, c, m.Hp / 16, in, m.b1 + c * 16, 0, spart);
      for (int e = t; e < RT * 256; e += 512) {
        const int r = e >> 8, row = tl.row0 + 16 * r + ((e & 255) >> 4);
        if (row < tl.nrows) {
          const float v = splitk_combine<RT, 1>(spart, r, 0, e & 255);
          st.a1[(size_t)row * m.Hp + c * 16 + (e & 15)] = v > 0.0f ? v : 0.0f;
        }
      }
      __syncthreads();
    }
    return;
  }
  splitk_tile<1, RT, CT>(m.w1, 0, ft0, m.Hp / 16, in, m.b1 + ft0 * 16, 0, spart);
  for (int e = t; e < RT * CT * 256; e += 512) {
    const int r = e / (CT * 256), c = (e / 256) % CT, el = e & 255;
    const int row = tl.row0 + 16 * r + (el >> 4);
    if (row < tl.nrows) {
      const float v = splitk_combine<RT, CT>(spart, r, c, el);
      st.a1[(size_t)row * m.Hp + (ft0 + c) * 16 + (el & 15)] = v > 0.0f ? v : 0.0f;
    }
  }
}

// m = b2 + W2 a1; running-mean update (uisrnn.py:425-429) -> dst slot
template <int RT>
__global__ __launch_bounds__(512) void k_dense_head2(DevModel m, DecodeState st, int par) {
  __shared__ __attribute__((aligned(16))) float spart[UIS_KSPLIT * RT * 256];
  RnnRow rb[RT];
  const StepTile tl = step_tile<RT, 1>(st, par, m.Dp / 16, rb);
  if (!tl.active) return;
  const int t = threadIdx.x;
  const int f = tl.ft0 * 16 + (t & 15);
  constexpr int EPT = (RT + 1) / 2;
  RnnRow re[EPT];
  float old[EPT];
  bool ework[EPT];
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int r = (t >> 8) + 2 * k;
    const int erow = tl.row0 + 16 * r + ((t & 255) >> 4);
    ework[k] = r < RT && erow < tl.nrows;
    old[k] = 0.0f;
    re[k] = RnnRow{};
    if (ework[k]) {
      re[k] = st.rows[erow];
      if (re[k].src >= 0) old[k] = st.pool_mean[((size_t)re[k].utt * st.S + re[k].src) * m.Dp + f];
    }
  }
  const float* in[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) in[r] = st.a1 + (size_t)(tl.row0 + 16 * r + (t & 15)) * m.Hp;
  splitk_tile<1, RT, 1>(m.w2, 0, tl.ft0, m.Hp / 16, in, m.b2 + tl.ft0 * 16, 0, spart);
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    if (!ework[k]) continue;
    const int r = (t >> 8) + 2 * k;
    float v = splitk_combine<RT, 1>(spart, r, 0, t & 255);
    if (re[k].src >= 0) v = uis_mean_update(old[k], v, re[k].nprev);
    if (f >= m.D) v = 0.0f;
    st.pool_mean[((size_t)re[k].utt * st.S + re[k].dst) * m.Dp + f] = v;
  }
}

// ------------------------------------------------- big-tile dense kernels (thousands of rows)
//
// With thousands of rnn rows per step (wide beams under look_ahead, thousands of utterances) the
// split-K kernels above are bound by the L2 -> CU stream: a workgroup re-reads 3 x 32 KB of weights
// for every 32 rows.  Here a 256-thread workgroup owns FOUR row tiles x FT feature tiles (x 3
// gates); a wave owns one row tile and walks the FULL K of every weight stream (segment chains
// combined on the fly, the order of uis_numerics.h -- bit-identical to the split-K schedule), so a
// weight fragment fetched once serves the four waves out of the CU's L1 and a row fragment serves
// NA MFMAs.  No LDS: after the chains lane l holds row (l & 15), features 4 (l >> 4) .. + 3 of
// every stream and runs the stage's elementwise tail on them as float4s.
template <int NA>
__device__ __forceinline__ void fullk_multi(const f32x4* const (&wp)[NA], const float* const (&bias)[NA], int nKb,
                                            const float* __restrict__ inrow, f32x4 (&total)[NA]) {
  const int lane = threadIdx.x & 63;
  const int q = lane >> 4;
  const int per = uis_kseg_blocks(nKb);
  const f32x4* bp[1] = {reinterpret_cast<const f32x4*>(inrow) + q};
#pragma unroll 1
  for (int sgm = 0; sgm < UIS_KSPLIT; ++sgm) {
    const int kb0 = sgm * per;
    const int kb1 = kb0 + per < nKb ? kb0 + per : nKb;
    f32x4 acc[1][NA];
#pragma unroll
    for (int a = 0; a < NA; ++a)
      acc[0][a] = sgm == 0 ? *reinterpret_cast<const f32x4*>(bias[a] + 4 * q) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (kb0 < kb1) chain_blocks<NA, 1>(wp, bp, kb0, kb1, acc);
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      if (sgm == 0) total[a] = acc[0][a];
      else {
#pragma unroll
        for (int i = 0; i < 4; ++i) total[a][i] = total[a][i] + acc[0][a][i];
      }
    }
  }
}

#define UIS_BIG_RT 4  // row tiles per workgroup = waves per workgroup

// Which (row tiles, feature tiles) a big-tile workgroup owns; this lane's row.
struct BigTile {
  int ft0, row, nrows;
  bool active, valid;
};
template <int FT>
__device__ __forceinline__ BigTile big_tile(const DecodeState& st, int par, int nft, RnnRow& mine) {
  BigTile t;
  const int max_rt = (st.max_rows + 15) >> 4;
  const int n_rg = (max_rt + UIS_BIG_RT - 1) / UIS_BIG_RT, n_fg = nft / FT;
  int rg, fg;
  dense_block_map((int)blockIdx.x, n_rg, n_fg, rg, fg);
  t.ft0 = fg * FT;
  const bool in_grid = rg < n_rg && fg < n_fg;
  t.row = (rg * UIS_BIG_RT + (int)(threadIdx.x >> 6)) * 16 + (int)(threadIdx.x & 15);
  if (!in_grid || t.row >= st.max_rows) t.row = 0;  // (never a valid row: nrows <= max_rows; keeps every address inside its buffer)
  mine = st.rows[t.row];
  t.nrows = st.nrows[par];
  t.active = in_grid && (rg * UIS_BIG_RT + (int)(threadIdx.x >> 6)) * 16 < t.nrows;  // per wave
  t.valid = t.active && t.row < t.nrows;
  return t;
}
__host__ __device__ inline int big_grid_blocks(int max_rows, int nft, int FT) {
  const int max_rt = (max_rows + 15) >> 4;
  return dense_grid_blocks((max_rt + UIS_BIG_RT - 1) / UIS_BIG_RT, nft / FT);
}

// GRU layer, FT feature tiles x 3 gates per workgroup
template <int FT>
__global__ __launch_bounds__(256) void k_big_gru(DevModel m, DecodeState st, int par, int layer) {
  const int nft = m.Hp / 16, nKb = m.Hp / 16;
  RnnRow me;
  const BigTile tl = big_tile<FT>(st, par, nft, me);
  if (!tl.active) return;
  const int lane = threadIdx.x & 63, q = lane >> 4;
  const float* hs = me.src >= 0 ? hid_ptr(m, st, me, me.src, layer) : m.h1 + (size_t)layer * m.Hp;
  const f32x4* wp[3 * FT];
  const float* bias[3 * FT];
#pragma unroll
  for (int c = 0; c < FT; ++c)
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      wp[c * 3 + g] = reinterpret_cast<const f32x4*>(m.whh[layer]) + ((size_t)(g * nft + tl.ft0 + c) * nKb) * 64 + lane;
      bias[c * 3 + g] = m.bhh[layer] + (size_t)g * m.Hp + (tl.ft0 + c) * 16;
    }
  // epilogue operands first: they are used after the chains
  const float* gi = layer == 0 ? st.gi0 + (size_t)me.frame * m.G : st.gi_up + (size_t)tl.row * m.G;
  f32x4 gir[FT], giz[FT], gin[FT], hprev[FT];
#pragma unroll
  for (int c = 0; c < FT; ++c) {
    const int j4 = (tl.ft0 + c) * 16 + 4 * q;
    gir[c] = *reinterpret_cast<const f32x4*>(gi + j4);
    giz[c] = *reinterpret_cast<const f32x4*>(gi + m.Hp + j4);
    gin[c] = *reinterpret_cast<const f32x4*>(gi + 2 * m.Hp + j4);
    hprev[c] = *reinterpret_cast<const f32x4*>(hs + j4);
  }
  f32x4 total[3 * FT];
  fullk_multi<3 * FT>(wp, bias, nKb, hs, total);
  if (!tl.valid) return;
  float* hd = const_cast<float*>(hid_ptr(m, st, me, me.dst, layer));
#pragma unroll
  for (int c = 0; c < FT; ++c) {
    const int j4 = (tl.ft0 + c) * 16 + 4 * q;
    f32x4 out;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      out[i] = j4 + i < m.H ? uis_gru_unit(gir[c][i], giz[c][i], gin[c][i], total[c * 3 + 0][i], total[c * 3 + 1][i],
                                           total[c * 3 + 2][i], hprev[c][i])
                            : 0.0f;
    *reinterpret_cast<f32x4*>(hd + j4) = out;
  }
}

// a1[row] = relu(b1 + W1 h'_top), FT feature tiles per workgroup
template <int FT>
__global__ __launch_bounds__(256) void k_big_head1(DevModel m, DecodeState st, int par) {
  const int nft = m.Hp / 16, nKb = m.Hp / 16;
  RnnRow me;
  const BigTile tl = big_tile<FT>(st, par, nft, me);
  if (!tl.active) return;
  const int lane = threadIdx.x & 63, q = lane >> 4;
  const float* in = hid_ptr(m, st, me, me.dst, m.depth - 1);
  const f32x4* wp[FT];
  const float* bias[FT];
#pragma unroll
  for (int c = 0; c < FT; ++c) {
    wp[c] = reinterpret_cast<const f32x4*>(m.w1) + ((size_t)(tl.ft0 + c) * nKb) * 64 + lane;
    bias[c] = m.b1 + (tl.ft0 + c) * 16;
  }
  f32x4 total[FT];
  fullk_multi<FT>(wp, bias, nKb, in, total);
  if (!tl.valid) return;
#pragma unroll
  for (int c = 0; c < FT; ++c) {
    f32x4 v = total[c];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.0f ? v[i] : 0.0f;
    *reinterpret_cast<f32x4*>(st.a1 + (size_t)tl.row * m.Hp + (tl.ft0 + c) * 16 + 4 * q) = v;
  }
}

// m = b2 + W2 a1; running-mean update -> dst slot, FT feature tiles per workgroup
template <int FT>
__global__ __launch_bounds__(256) void k_big_head2(DevModel m, DecodeState st, int par) {
  const int nft = m.Dp / 16, nKb = m.Hp / 16;
  RnnRow me;
  const BigTile tl = big_tile<FT>(st, par, nft, me);
  if (!tl.active) return;
  const int lane = threadIdx.x & 63, q = lane >> 4;
  const float* in = st.a1 + (size_t)tl.row * m.Hp;  // (rows past nrows: stale but inside the buffer)
  const f32x4* wp[FT];
  const float* bias[FT];
  f32x4 old[FT];
#pragma unroll
  for (int c = 0; c < FT; ++c) {
    wp[c] = reinterpret_cast<const f32x4*>(m.w2) + ((size_t)(tl.ft0 + c) * nKb) * 64 + lane;
    bias[c] = m.b2 + (tl.ft0 + c) * 16;
    old[c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (tl.valid && me.src >= 0)
      old[c] = *reinterpret_cast<const f32x4*>(st.pool_mean + ((size_t)me.utt * st.S + me.src) * m.Dp + (tl.ft0 + c) * 16 + 4 * q);
  }
  f32x4 total[FT];
  fullk_multi<FT>(wp, bias, nKb, in, total);
  if (!tl.valid) return;
#pragma unroll
  for (int c = 0; c < FT; ++c) {
    const int f4 = (tl.ft0 + c) * 16 + 4 * q;
    f32x4 v = total[c];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (me.src >= 0) v[i] = uis_mean_update(old[c][i], v[i], me.nprev);
      if (f4 + i >= m.D) v[i] = 0.0f;
    }
    *reinterpret_cast<f32x4*>(st.pool_mean + ((size_t)me.utt * st.S + me.dst) * m.Dp + f4) = v;
  }
}

// -------------------------------------------------- L2-coherent loads (resident decode)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// 16 bytes with sc1: bypasses this CU's L1, served by the XCD's L2 (where the sibling
// workgroups' plain stores land)
__device__ __forceinline__ f32x4 load_sc1(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 16 /* sc1 */));
}

// mse0[frame] = weighted MSE(m0, x[frame])   (fresh-cluster score term; one wave per frame)
__global__ __launch_bounds__(256) void k_mse0(DevModel m, const float* __restrict__ x,
                                              float* __restrict__ mse0, long nframes, long batch_stride,
                                              const long* __restrict__ batch_tab) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if (batch_tab) {  // (batches along grid.z: see k_dense_input_proj_pipe)
    x += (size_t)batch_tab[2 * blockIdx.z] * m.Dp;
    mse0 += (size_t)batch_tab[2 * blockIdx.z];
    nframes = batch_tab[2 * blockIdx.z + 1];
  } else {
    x += (size_t)blockIdx.z * (size_t)batch_stride * m.Dp;
    mse0 += (size_t)blockIdx.z * (size_t)batch_stride;
  }
  float* swgt = reinterpret_cast<float*>(smem_raw);
  float* sx = swgt + m.Dp;  // 4 waves x Dp
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < m.Dp; i += 256) swgt[i] = m.wgt[i];
  const long frame = (long)blockIdx.x * 4 + wave;
  float* myx = sx + (size_t)wave * m.Dp;
  if (frame < nframes)
    for (int i = lane; i < m.Dp; i += 64) myx[i] = x[(size_t)frame * m.Dp + i];
  __syncthreads();
  if (frame >= nframes) return;
  float v = wave_weighted_mse(m.m0, myx, swgt, m.Dp, m.D, lane);
  if (lane == 0) mse0[frame] = v;
}

// Zero-pad the frame stream to Dp columns when D is not a multiple of 16.
// (round 5) A ragged list's time slice arrives from the host as ONE block (the cast writes it utterance after
// utterance); row i of utterance z's part goes to its place in the utterance-major frame stream.  tab[3 z] = first row
// in the block, [3 z + 1] = first row in the stream, [3 z + 2] = rows.  16 bytes per thread.
__global__ __launch_bounds__(256) void k_scatter_rows(const float* __restrict__ block, float* __restrict__ x,
                                                      const long* __restrict__ tab, int D) {
  const long s0 = tab[3 * blockIdx.y], d0 = tab[3 * blockIdx.y + 1], n = tab[3 * blockIdx.y + 2];
  const long per_row = D / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n * per_row; i += (long)gridDim.x * 256)
    reinterpret_cast<f32x4*>(x + (size_t)d0 * D)[i] = reinterpret_cast<const f32x4*>(block + (size_t)s0 * D)[i];
}

__global__ void k_pad_frames(const float* __restrict__ src, float* __restrict__ dst, long nframes, int D, int Dp) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = nframes * Dp;
  if (i >= total) return;
  long r = i / Dp; int c = (int)(i % Dp);
  dst[i] = c < D ? src[r * D + c] : 0.0f;
}

// Reset the per-decode state: every utterance starts with one empty hypothesis
// (beam_set = [BeamState()], uisrnn.py:528).
__global__ void k_init_state(DecodeState st) {
  int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u == 0) { st.nrows[0] = 0; st.nrows[1] = 0; for (int i = 0; i < 4; ++i) st.counters[i] = 0ull;  // this group's
#if defined(UIS_SELECT_TIMING)
    for (int i = 16; i < 48; ++i) st.counters[i] = 0ull;
#endif
  }
  if (u >= st.U) return;
  st.utt_step[u] = 0;
  st.overflow[u] = 0;
  st.beam_n[u] = 1;            // parity 0
  st.beam_n[st.U + u] = 0;
  size_t e = (size_t)u * st.B;
  st.beam_K[e] = 0; st.beam_last[e] = -1; st.beam_sum[e] = 0; st.beam_score[e] = 0.0f;
}

// ------------------------------------------------------------------ select
//
// One workgroup per utterance, one decode step (look_ahead == 1 window):
//   A  weighted MSE of the frame against every live cluster state (one wave per slot)
//   B  score every (hypothesis, cluster) candidate: float32(mse - prior) accumulated in float32
//   C  keep the min(#finite, B) lowest (ties: lowest (hypothesis, cluster))
//   D  build the next beam tables, allocate destination slots, emit the rnn rows
// Dynamic LDS layout is carved by select_lds_bytes() below.

struct SelectLds {
  int off_wgt, off_slot, off_blk, off_K, off_last, off_sum, off_base, off_score;
  int off_live, off_livelist, off_mse, off_cnt, off_key, off_cscore, off_win, off_src, off_dst, off_lead, off_free, off_misc;
  int total;
};

__host__ __device__ inline SelectLds select_lds_layout(int Dp, int B, int Kmax, int S) {
  SelectLds l;
  int o = 0;
  auto take = [&](int bytes) { int r = o; o += (bytes + 15) & ~15; return r; };
  const int C = B * (Kmax + 1);
  l.off_wgt = take(Dp * 4);
  l.off_slot = take(B * Kmax * 4);
  l.off_blk = take(B * Kmax * 4);
  l.off_K = take(B * 4);
  l.off_last = take(B * 4);
  l.off_sum = take(B * 4);
  l.off_base = take((B + 1) * 4);
  l.off_score = take(B * 4);
  l.off_live = take(S * 4);
  l.off_livelist = take(S * 4);
  l.off_mse = take(S * 4);
  l.off_cnt = take(S * 4);
  l.off_key = take(C * 8);
  l.off_cscore = take(C * 4);
  l.off_win = take(B * 4);
  l.off_src = take(B * 4);
  l.off_dst = take(B * 4);
  l.off_lead = take(B * 4);
  l.off_free = take(B * 4);
  l.off_misc = take(16 * 4);
  l.total = o;
  return l;
}

// Memory round trips on the critical path: (1) step counter, offsets and the whole beam
// tables; (2) the frame, the live cluster states (+ frame counts), the prior-table entries
// and the fresh-cluster MSE -- all issued before the first of them is consumed.
// Diagnostic build (-DUIS_SELECT_TIMING): thread 0 of every workgroup adds the shader-clock
// cycles of each phase to counters[16 + phase]; decode_impl prints the averages.
#if defined(UIS_SELECT_TIMING)
#define TSTAMP(k) do { if (threadIdx.x == 0) { const unsigned long long t_now_ = __builtin_readcyclecounter(); \
    atomicAdd(&st.counters[16 + (k)], t_now_ - t_prev_); t_prev_ = t_now_; } } while (0)
#else
#define TSTAMP(k) do {} while (0)
#endif

__global__ __launch_bounds__(256) void k_select(DevModel m, DecodeState st, int par) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int u = blockIdx.x, tid = threadIdx.x;
#if defined(UIS_SELECT_TIMING)
  unsigned long long t_prev_ = __builtin_readcyclecounter();
#endif
  const int B = st.B, Kmax = st.Kmax, S = st.S, U = st.U;
  // streaming: every utterance has its own step count, so the parity of ITS tables is its own
  // step's; `par` stays the parity of the launch (row counter)
  const int tpar = st.avail ? (st.utt_step[u] & 1) : par;
  const int nxt = tpar ^ 1;

  const SelectLds L = select_lds_layout(m.Dp, B, Kmax, S);
  float* swgt = reinterpret_cast<float*>(smem_raw + L.off_wgt);
  int* sslot = reinterpret_cast<int*>(smem_raw + L.off_slot);
  int* sblk = reinterpret_cast<int*>(smem_raw + L.off_blk);
  int* sK = reinterpret_cast<int*>(smem_raw + L.off_K);
  int* slast = reinterpret_cast<int*>(smem_raw + L.off_last);
  int* ssum = reinterpret_cast<int*>(smem_raw + L.off_sum);
  int* sbase = reinterpret_cast<int*>(smem_raw + L.off_base);
  float* sscore = reinterpret_cast<float*>(smem_raw + L.off_score);
  int* slive = reinterpret_cast<int*>(smem_raw + L.off_live);
  int* slivelist = reinterpret_cast<int*>(smem_raw + L.off_livelist);
  float* smse = reinterpret_cast<float*>(smem_raw + L.off_mse);
  int* scnt = reinterpret_cast<int*>(smem_raw + L.off_cnt);
  unsigned long long* skey = reinterpret_cast<unsigned long long*>(smem_raw + L.off_key);
  float* scscore = reinterpret_cast<float*>(smem_raw + L.off_cscore);
  int* swin = reinterpret_cast<int*>(smem_raw + L.off_win);
  int* ssrc = reinterpret_cast<int*>(smem_raw + L.off_src);
  int* sdst = reinterpret_cast<int*>(smem_raw + L.off_dst);
  int* slead = reinterpret_cast<int*>(smem_raw + L.off_lead);
  int* sfree = reinterpret_cast<int*>(smem_raw + L.off_free);
  int* smisc = reinterpret_cast<int*>(smem_raw + L.off_misc);  // [0] nlive [1] nfinite [2] nlead [3] nfree

  const size_t bcur = ((size_t)tpar * U + u) * B;
  const size_t bnxt = ((size_t)nxt * U + u) * B;

  // ---- round trip 1 (entries of the tables beyond K_b / nb are garbage and never used)
  const int step = st.utt_step[u];
  const long off0 = (long)st.off[u], off1 = (long)st.off[u + 1];
  const int nb = st.beam_n[(size_t)tpar * U + u];
  // next step's row counter: its readers (the previous step's GEMMs) finished a launch ago
  if (u == 0 && tid == 0) st.nrows[par ^ 1] = 0;
  for (int i = tid; i < m.Dp; i += 256) swgt[i] = m.wgt[i];
  for (int e = tid; e < B * Kmax; e += 256) {
    sslot[e] = st.beam_slot[bcur * Kmax + e];
    sblk[e] = st.beam_blk[bcur * Kmax + e];
  }
  for (int b = tid; b < B; b += 256) {
    sK[b] = st.beam_K[bcur + b]; slast[b] = st.beam_last[bcur + b];
    ssum[b] = st.beam_sum[bcur + b]; sscore[b] = st.beam_score[bcur + b];
  }
  for (int sl = tid; sl < S; sl += 256) slive[sl] = 0;
  if (tid < 16) smisc[tid] = 0;
  const long N = off1 - off0;
  const long T = st.avail ? (long)st.avail[u] : (long)st.tau * N;
  if (step >= T) return;  // utterance finished (uniform over the workgroup)
  const long frame = st.foff ? (long)st.foff[u] + step : off0 + (step % N);  // np.tile(seq, (tau, 1)), uisrnn.py:524
  __syncthreads();
  TSTAMP(0);
  for (int e = tid; e < nb * Kmax; e += 256) {
    const int b = e / Kmax, c = e - b * Kmax;
    if (c < sK[b]) slive[sslot[e]] = 1;
  }
  if (tid == 0) {  // candidate offsets: hypothesis b owns candidates sbase[b] .. sbase[b] + K_b
    int acc = 0;
    for (int b = 0; b < nb; ++b) { sbase[b] = acc; acc += sK[b] + 1; }
    sbase[nb] = acc;
  }
  __syncthreads();
  for (int sl = tid; sl < S; sl += 256)
    if (slive[sl]) slivelist[atomicAdd(&smisc[0], 1)] = sl;
  __syncthreads();
  const int nlive = smisc[0];
  const int C = sbase[nb];
  TSTAMP(1);

  // ---- round trip 2, part 1: this thread's candidate (the first 256) -- prior-table entries
  const float mse_new = st.mse0[frame];
  int my_b = 0, my_c = 0;
  double my_lb = 0.0, my_ld = 0.0;
  if (tid < C) {
    while (tid >= sbase[my_b + 1]) ++my_b;
    my_c = tid - sbase[my_b];
    my_ld = st.logden[ssum[my_b]];
    if (my_c < sK[my_b] && my_c != slast[my_b]) my_lb = st.logblk[sblk[my_b * Kmax + my_c]];
  }

  // ---- A: weighted MSE of the frame against every live cluster state.
  // 16 lanes per cluster state (16 states per pass over the workgroup), canonical order
  // (uis_numerics.h): quad sums in registers, tiles and accumulators by DPP inside the 16-lane row.
  const float* pmean = st.pool_mean + (size_t)u * S * m.Dp;
  const float* xrow = st.x + (size_t)frame * m.Dp;
  {
    const int grp = tid >> 4, p = tid & 15;
    for (int i0 = 0; i0 < nlive; i0 += 16) {
      const int i = i0 + grp;
      const bool act = i < nlive;
      const int sl = slivelist[act ? i : 0];
      const float* mean = pmean + (size_t)sl * m.Dp;
      const int cnt = (p == 0) ? st.pool_cnt[(size_t)u * S + sl] : 0;
      float A[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      float first_sq = 0.0f;
      for (int q = 0; q < m.Dp; q += 256) {
        f32x4 mv[4], xv[4], wv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int d = q + 4 * (p + 16 * k);
          const bool in = d < m.Dp;
          mv[k] = in ? *reinterpret_cast<const f32x4*>(mean + d) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
          xv[k] = in ? *reinterpret_cast<const f32x4*>(xrow + d) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
          wv[k] = in ? *reinterpret_cast<const f32x4*>(swgt + d) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        mse16_block(mv, xv, wv, A);
        if (q == 0) { const float d0 = mv[0][0] - xv[0][0]; first_sq = d0 * d0; }
      }
      const float t = mse16_total(A);
      if (p == 0 && act) { smse[sl] = uis_mse_finish(t, first_sq, m.D); scnt[sl] = cnt; }
    }
  }
  __syncthreads();

  TSTAMP(2);
  // ---- B: candidate scores
  for (int i = tid; i < C; i += 256) {
    int b, c;
    double lb, ld;
    if (i == tid) { b = my_b; c = my_c; lb = my_lb; ld = my_ld; }
    else {
      b = 0;
      while (i >= sbase[b + 1]) ++b;
      c = i - sbase[b];
      ld = st.logden[ssum[b]];
      lb = (c < sK[b] && c != slast[b]) ? st.logblk[sblk[b * Kmax + c]] : 0.0;
    }
    float mse; double prior;
    if (c < sK[b]) {  // existing cluster, uisrnn.py:409-420
      mse = smse[sslot[b * Kmax + c]];
      prior = (c == slast[b]) ? m.lp_stay : (m.lp_sw + lb) - ld;
    } else {          // new cluster, uisrnn.py:440-446
      mse = mse_new;
      prior = m.lp_new - ld;
    }
    const float sc = sscore[b] + uis_step_loss(mse, prior);  // float32 accumulate, uisrnn.py:452
    scscore[i] = sc;
    if (st.dbg_scores) st.dbg_scores[(((size_t)step * U + u) * B + b) * (Kmax + 1) + c] = sc;
    const bool fin = uis_isfinite(sc);
    skey[i] = fin ? (((unsigned long long)uis_score_key(sc) << 32) | (unsigned)i) : ~0ull;
    if (fin) atomicAdd(&smisc[1], 1);
  }
  __syncthreads();

  TSTAMP(3);
  // ---- C: rank by counting (keys are unique)
  const int lane = tid & 63, wave = tid >> 6;
  const int nfin = smisc[1];
  const int keep = nfin < B ? nfin : B;  // uisrnn.py:551-552
  for (int i = tid; i < C; i += 256) {
    const unsigned long long k = skey[i];
    if (k == ~0ull) continue;
    int rank = 0;
    for (int j2 = 0; j2 < C; ++j2) rank += skey[j2] < k;
    if (rank < keep) swin[rank] = i;
  }
  __syncthreads();

  TSTAMP(4);
  // ---- D: winners -> (parent, cluster, source slot); dedup rows by source slot
  const bool nodedup = (st.flags & 1u) != 0;
  if (tid < keep) {
    const int i = swin[tid];
    int b = 0;
    while (i >= sbase[b + 1]) ++b;
    const int c = i - sbase[b];
    ssrc[tid] = c < sK[b] ? sslot[b * Kmax + c] : -1;
  }
  __syncthreads();
  if (tid < keep) {
    int lead = tid;
    if (!nodedup)
      for (int r2 = 0; r2 < tid; ++r2) if (ssrc[r2] == ssrc[tid]) { lead = r2; break; }
    slead[tid] = lead;
  }
  __syncthreads();
  if (tid == 0) {  // ordinal of each leader (keep <= B, serial is fine)
    int n = 0;
    for (int r = 0; r < keep; ++r) if (slead[r] == r) sdst[r] = n++; else sdst[r] = -1;
    smisc[2] = n;
  }
  __syncthreads();
  const int nlead = smisc[2];
  // first nlead free slots (not referenced by the current beam), in slot order;
  // smisc[3] = free slots seen so far, smisc[8..11] = per-wave counts
  for (int base = 0; base < S && smisc[3] < nlead; base += 256) {
    const int s = base + tid;
    const bool fr = s < S && !slive[s];
    const unsigned long long mask = __ballot(fr);
    if (lane == 0) smisc[8 + wave] = __popcll(mask);
    __syncthreads();
    int before = smisc[3];
    for (int w = 0; w < wave; ++w) before += smisc[8 + w];
    const int pos = before + __popcll(mask & ((1ull << lane) - 1ull));
    if (fr && pos < nlead) sfree[pos] = s;
    __syncthreads();
    if (tid == 0) smisc[3] += smisc[8] + smisc[9] + smisc[10] + smisc[11];
    __syncthreads();
  }
  // resolve dst: leaders take sfree[ordinal]; followers copy their leader's
  if (tid < keep && slead[tid] == tid) sdst[tid] = sfree[sdst[tid]];
  __syncthreads();
  if (tid < keep && slead[tid] != tid) sdst[tid] = sdst[slead[tid]];
  __syncthreads();

  TSTAMP(5);
  // next beam tables (BeamState copy + the list updates of uisrnn.py:425-433,451)
  for (int e = tid; e < keep * Kmax; e += 256) {
    const int r = e / Kmax, c2 = e - r * Kmax;
    const int i = swin[r];
    int b = 0;
    while (i >= sbase[b + 1]) ++b;
    const int c = i - sbase[b];
    const int Kb = sK[b];
    const bool is_new = c == Kb;
    const int Knew = Kb + (is_new ? 1 : 0);
    if (c2 < Knew && c2 < Kmax) {
      int slot, blk;
      if (c2 == c) { slot = sdst[r]; blk = is_new ? 1 : sblk[b * Kmax + c] + (c != slast[b] ? 1 : 0); }
      else { slot = sslot[b * Kmax + c2]; blk = sblk[b * Kmax + c2]; }
      st.beam_slot[(bnxt + r) * Kmax + c2] = slot;
      st.beam_blk[(bnxt + r) * Kmax + c2] = blk;
    }
  }
  if (tid < keep) {
    const int r = tid;
    const int i = swin[r];
    int b = 0;
    while (i >= sbase[b + 1]) ++b;
    const int c = i - sbase[b];
    const int Kb = sK[b];
    const bool is_new = c == Kb;
    int Knew = Kb + (is_new ? 1 : 0);
    if (Knew > Kmax) { Knew = Kmax; st.overflow[u] = 1; }  // cluster cap hit: utterance flagged
    st.beam_K[bnxt + r] = Knew;
    st.beam_last[bnxt + r] = c;
    st.beam_sum[bnxt + r] = ssum[b] + ((is_new || c != slast[b]) ? 1 : 0);
    st.beam_score[bnxt + r] = scscore[i];
    st.bp[((size_t)st.tau * st.off[u] + step) * B + r] = ((unsigned)b << 16) | (unsigned)c;
    atomicMax(&st.counters[3], (unsigned long long)Knew);
    if (slead[r] == r) {  // emit the rnn row
      const int src = ssrc[r];
      const int nprev = src >= 0 ? scnt[src] : 0;
      st.pool_cnt[(size_t)u * S + sdst[r]] = nprev + 1;
      const int pos = atomicAdd(&st.nrows[par], 1);
      RnnRow rr; rr.utt = u; rr.src = src; rr.dst = sdst[r]; rr.nprev = nprev; rr.frame = frame; rr.pad = 0;
      st.rows[pos] = rr;
    }
  }
  TSTAMP(6);
  if (tid == 0) {
    st.beam_n[(size_t)nxt * U + u] = keep;
    st.utt_step[u] = step + 1;
    atomicAdd(&st.counters[0], (unsigned long long)nlead);
    atomicAdd(&st.counters[1], (unsigned long long)keep);
    atomicAdd(&st.counters[2], (unsigned long long)C);
  }
  TSTAMP(7);
}

// ------------------------------------------------------------- select, fast path
//
// Same result as k_select for the common shape: at most 64 hypotheses, at most 256 candidates
// (thread i owns candidate i), at most 1024 slots.  Six workgroup barriers instead of fifteen:
// prefix sums, leader election, free-slot choice and the table writes are done by wave 0 with
// ballots and shuffles, and the row-counter atomic is issued as soon as the row count is known
// so its round trip overlaps the table writes.

// set bits of a wave mask below this lane (v_mbcnt: two instructions)
__device__ __forceinline__ int wave_below(unsigned long long mask) {
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
// inclusive prefix sum over the wave: row_shr 1, 2, 4, 8 inside each row of 16 lanes (zeros shifted
// in), then the totals of the rows before this one
__device__ __forceinline__ int wave_incl_scan_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
  const int t0 = __builtin_amdgcn_readlane(v, 15), t1 = __builtin_amdgcn_readlane(v, 31), t2 = __builtin_amdgcn_readlane(v, 47);
  const int row = (int)(threadIdx.x & 63) >> 4;
  return v + (row >= 1 ? t0 : 0) + (row >= 2 ? t1 : 0) + (row >= 3 ? t2 : 0);
}

struct FastLds {
  int off_wgt, off_slot, off_blk, off_K, off_last, off_sum, off_score, off_nb, set_stride;
  int off_base, off_live, off_livelist, off_mse, off_cnt, off_key, off_win, off_wscore, off_misc, off_pcnt, off_cand;
  int total;
};
// The beam tables (slot, blk, K, last, sum, score, nb) form a SET; there are two of them so that
// the resident decode can keep the beam in LDS across steps (set = step parity).  The
// launch-per-step kernel uses set 0 only.
__host__ __device__ inline FastLds fast_lds_layout(int Dp, int B, int Kmax, int S) {
  FastLds l;
  int o = 0;
  auto take = [&](int bytes) { int r = o; o += (bytes + 15) & ~15; return r; };
  l.off_wgt = take(Dp * 4);
  const int set0 = o;
  l.off_slot = take(B * Kmax * 4);
  l.off_blk = take(B * Kmax * 4);
  l.off_K = take(B * 4);
  l.off_last = take(B * 4);
  l.off_sum = take(B * 4);
  l.off_score = take(B * 4);
  l.off_nb = take(16);
  l.set_stride = o - set0;
  o += l.set_stride;  // the second set
  l.off_base = take((B + 1) * 4);
  l.off_live = take(S * 4);
  l.off_livelist = take(S * 4);
  l.off_mse = take(S * 4);
  l.off_cnt = take(S * 4);
  l.off_key = take(256 * 8);
  l.off_win = take(64 * 4);
  l.off_wscore = take(64 * 4);
  l.off_misc = take(64 * 4);  // [0] nlive [1] nfinite; [16..31] phase clocks of the timing build
  l.off_pcnt = take(S * 4);   // resident decode: frames per slot (what pool_cnt holds), kept across steps
  l.off_cand = take(256 * 4);  // candidate i -> (hypothesis << 16) | cluster
  l.total = o;
  return l;
}
__host__ __device__ inline bool select_fast_ok(int B, int Kmax, int S) {
  return B <= 64 && B * (Kmax + 1) <= 256 && S <= 1024;
}

// Where a select appends its rnn rows: the launch-per-step path has one list per utterance
// group, the resident decode one per XCD.
struct RowSink {
  RnnRow* rows;
  int32_t* count;
};

// The select of utterance u by an NT-thread workgroup.  RES: called from the resident decode --
// the cluster means were written by other CUs of this XCD inside the same launch, so they are
// read with sc1 loads; everything else a select reads is either immutable or was written by
// this very workgroup.  The caller provides the workgroup barrier that ends the call.
#if defined(UIS_RESIDENT_TIMING)
// resident timing build: thread 0 accumulates wall-clock ticks (10 ns) per select phase in LDS
#undef TSTAMP
#define TSTAMP(k) do { if (threadIdx.x == 0) { const unsigned long long t_now_ = wall_clock64(); \
    reinterpret_cast<unsigned long long*>(smem_raw + fast_lds_layout(m.Dp, st.B, st.Kmax, st.S).off_misc + 64)[k] += t_now_ - t_prev_; \
    t_prev_ = t_now_; } } while (0)
#endif
// KEEP (resident decode, one utterance per workgroup): the beam tables and the per-slot frame
// counts stay in LDS from one step to the next -- a step reads table set `par` and writes set
// `par ^ 1` -- so the only global state a select reads back is the cluster means; the caller
// passes the step number and the utterance's frame range.  In a streaming session (st.avail) the
// step number is the utterance's own (the launch starts at first_step_in, where the tables and
// the per-slot frame counts are fetched from the global copies the previous push left), `par` its
// parity, and the resident decode writes the tables back when the launch ends.
// DPT: the padded observation_dim when the caller knows it at compile time (0 = read m.Dp).
// PARTS (bit mask, KEEP only; 7 = everything): 1 and 4 = the two halves of the PREPARATION of a
// step -- live slots and the candidate table, then the list of live slots: all of it depends on
// this workgroup's LDS tables only, so the resident decode runs the halves while it waits at
// the previous step's barriers; 2 = the rest.
// `published` is called by wave 0 as soon as the step's rnn rows are written (before the table
// updates nobody else reads): the resident decode arrives at its barrier there.
struct SelectNoHook { __device__ __forceinline__ void operator()() const {} };
template <int NT, bool RES, bool KEEP, int DPT = 0, int PARTS = 7, typename Pub = SelectNoHook>
__device__ __forceinline__ void select_fast_body(const DevModel& m, const DecodeState& st, int par, int u,
                                                 unsigned char* smem_raw, RowSink sink, int step_in = 0,
                                                 long off0_in = 0, long off1_in = 0, Pub published = Pub(),
                                                 int first_step_in = 0, unsigned char* keep_sets = nullptr,
                                                 int* keep_pcnt = nullptr) {
  static_assert(PARTS == 7 || KEEP, "the split needs the beam in LDS");
  int tid_ = threadIdx.x;
  // inside the resident decode's step loop: keep the compiler from hoisting every tid-derived
  // address out of the loop (they would have to live -- spilled -- across the dense stages)
  if (RES) asm volatile("" : "+v"(tid_));
  const int tid = tid_, lane = tid & 63, wave = tid >> 6;
#if defined(UIS_SELECT_TIMING)
  unsigned long long t_prev_ = __builtin_readcyclecounter();
#elif defined(UIS_RESIDENT_TIMING)
  unsigned long long t_prev_ = wall_clock64();
#endif
  const int B = st.B, Kmax = st.Kmax, S = st.S, U = st.U;
  // streaming: the parity of an utterance's tables is its own step's (KEEP: the caller passes it)
  const int tpar = (!KEEP && st.avail) ? (st.utt_step[u] & 1) : par;
  const int nxt = tpar ^ 1;
  const FastLds L = fast_lds_layout(m.Dp, B, Kmax, S);
  float* swgt = reinterpret_cast<float*>(smem_raw + L.off_wgt);
  // KEEP with several utterances per workgroup (k_decode_big): the two table sets and the per-slot
  // counts of utterance i live in a block of their own (keep_sets = block - L.off_slot: the set's
  // fields keep their relative offsets), the scratch areas are shared
  unsigned char* const sets0 = (KEEP && keep_sets) ? keep_sets : smem_raw;
  unsigned char* const set_cur = sets0 + (KEEP ? tpar * L.set_stride : 0);
  unsigned char* const set_nxt = sets0 + (KEEP ? nxt * L.set_stride : 0);
  int* sslot = reinterpret_cast<int*>(set_cur + L.off_slot);
  int* sblk = reinterpret_cast<int*>(set_cur + L.off_blk);
  int* sK = reinterpret_cast<int*>(set_cur + L.off_K);
  int* slast = reinterpret_cast<int*>(set_cur + L.off_last);
  int* ssum = reinterpret_cast<int*>(set_cur + L.off_sum);
  float* sscore = reinterpret_cast<float*>(set_cur + L.off_score);
  int* nslot = reinterpret_cast<int*>(set_nxt + L.off_slot);   // KEEP: next step's tables
  int* nblk = reinterpret_cast<int*>(set_nxt + L.off_blk);
  int* nK = reinterpret_cast<int*>(set_nxt + L.off_K);
  int* nlast = reinterpret_cast<int*>(set_nxt + L.off_last);
  int* nsum = reinterpret_cast<int*>(set_nxt + L.off_sum);
  float* nscore = reinterpret_cast<float*>(set_nxt + L.off_score);
  int* spcnt = (KEEP && keep_pcnt) ? keep_pcnt : reinterpret_cast<int*>(smem_raw + L.off_pcnt);
  unsigned* scand = reinterpret_cast<unsigned*>(smem_raw + L.off_cand);
  int* sbase = reinterpret_cast<int*>(smem_raw + L.off_base);
  int* slive = reinterpret_cast<int*>(smem_raw + L.off_live);
  int* slivelist = reinterpret_cast<int*>(smem_raw + L.off_livelist);
  float* smse = reinterpret_cast<float*>(smem_raw + L.off_mse);
  int* scnt = reinterpret_cast<int*>(smem_raw + L.off_cnt);
  unsigned long long* skey = reinterpret_cast<unsigned long long*>(smem_raw + L.off_key);
  int* swin = reinterpret_cast<int*>(smem_raw + L.off_win);
  float* swscore = reinterpret_cast<float*>(smem_raw + L.off_wscore);
  int* smisc = reinterpret_cast<int*>(smem_raw + L.off_misc);  // [0] nlive [1] nfinite

  const size_t bcur = ((size_t)tpar * U + u) * B;
  const size_t bnxt = ((size_t)nxt * U + u) * B;

  // ---- round trip 1: every load is issued before the first result is stored to LDS (written
  // as load-then-store pairs the compiler waits for each load before it issues the next).
  // B * Kmax < 256 <= NT (select_fast_ok): one table entry per thread.
  // KEEP, after the first step: nothing to fetch -- the tables are in LDS set `par`.
  const bool fresh = !KEEP || step_in == first_step_in;
  int step = step_in, nb = 0, myK = 0;
  long off0 = off0_in, off1 = off1_in;
  const bool has_e = tid < B * Kmax, has_b = tid < B;
  if (PARTS & 1) {
    if (fresh) {
      if (!KEEP) { step = st.utt_step[u]; off0 = (long)st.off[u]; off1 = (long)st.off[u + 1]; }
      nb = st.beam_n[(size_t)tpar * U + u];
      const int r_slot = has_e ? st.beam_slot[bcur * Kmax + tid] : 0;
      const int r_blk = has_e ? st.beam_blk[bcur * Kmax + tid] : 0;
      myK = has_b ? st.beam_K[bcur + tid] : 0;
      const int r_last = has_b ? st.beam_last[bcur + tid] : 0;
      const int r_sum = has_b ? st.beam_sum[bcur + tid] : 0;
      const float r_score = has_b ? st.beam_score[bcur + tid] : 0.0f;
      float r_w[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) r_w[k] = tid + k * NT < m.Dp ? m.wgt[tid + k * NT] : 0.0f;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (tid + k * NT < m.Dp) swgt[tid + k * NT] = r_w[k];
      for (int i = tid + 4 * NT; i < m.Dp; i += NT) swgt[i] = m.wgt[i];  // observation_dim > 4 * NT
      if (has_e) { sslot[tid] = r_slot; sblk[tid] = r_blk; }
      if (has_b) { sK[tid] = myK; slast[tid] = r_last; ssum[tid] = r_sum; sscore[tid] = r_score; }
      if (KEEP) {
        // frames per slot: none yet at the start of a decode; a streaming session -- or a decode in several launches
        // past its first (round 5) -- continues
        for (int sl = tid; sl < S; sl += NT) spcnt[sl] = (st.avail || step_in > 0) ? st.pool_cnt[(size_t)u * S + sl] : 0;
        if (tid == 0) *reinterpret_cast<int*>(set_cur + L.off_nb) = nb;
      }
    } else {
      nb = *reinterpret_cast<const int*>(set_cur + L.off_nb);
      myK = has_b ? sK[tid] : 0;
    }
    for (int sl = tid; sl < S; sl += NT) slive[sl] = 0;
    if (tid < 16) smisc[tid] = 0;
  } else {
    nb = *reinterpret_cast<const int*>(set_cur + L.off_nb);
  }
  const long N = off1 - off0;
  const long T = st.avail ? (long)st.avail[u] : (long)st.tau * N;
  if (step >= T) return;
  const long frame = st.foff ? (long)st.foff[u] + step : off0 + (step % N);
  if (PARTS & 1) {
    // candidate offsets: exclusive scan of K_b + 1 over the beam, by wave 0 (B <= 64)
    if (wave == 0) {  // DPP row scans + three row totals (a readlane broadcast per hypothesis cost nb scalar round trips)
      const int v = lane < nb ? myK + 1 : 0;
      const int incl = wave_incl_scan_i32(v);
      if (lane < nb) sbase[lane] = incl - v;
      if (lane == 63) sbase[nb] = incl;
    }
    __syncthreads();  // (1) tables staged
    TSTAMP(0);
    for (int e = tid; e < nb * Kmax; e += NT) {
      const int b = e / Kmax, c = e - b * Kmax;
      if (c < sK[b]) slive[sslot[e]] = 1;
    }
    // candidate i = sbase[b] + c  ->  (b, c): one LDS word per candidate instead of a search
    // over the prefix sums wherever a candidate index has to be decoded
    if (tid < nb * (Kmax + 1)) {  // <= 256 <= NT (select_fast_ok)
      const int b = tid / (Kmax + 1), c = tid - b * (Kmax + 1);
      if (c <= sK[b]) scand[sbase[b] + c] = ((unsigned)b << 16) | (unsigned)c;
    }
    __syncthreads();  // (2) live flags
  }
  if (PARTS & 4) {
    for (int base = 0; base < S; base += NT) {  // compaction: ballot per wave, one LDS atomic per wave
      const int sl = base + tid;
      const bool lv = sl < S && slive[sl] != 0;
      const unsigned long long mask = __ballot(lv);
      int wbase = 0;
      if (lane == 0 && mask) wbase = atomicAdd(&smisc[0], __popcll(mask));
      wbase = __shfl(wbase, 0, 64);
      if (lv) slivelist[wbase + __popcll(mask & ((1ull << lane) - 1ull))] = sl;
    }
    __syncthreads();  // (3) live list
    TSTAMP(1);
  }
  if (!(PARTS & 2)) return;
  const int nlive = smisc[0];
  const int C = sbase[nb];

  // ---- round trip 2: this thread's candidate (prior-table entries), then the cluster states
  // (resident decode: a streaming push's mse0 is written by this workgroup moments earlier, and a
  // plain uniform load would go through the scalar cache, which neighbouring CUs may share and may
  // have filled with this line before the write)
  const float mse_new = RES ? __hip_atomic_load(st.mse0 + frame, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : st.mse0[frame];
  int my_b = 0, my_c = 0;
  double my_lb = 0.0, my_ld = 0.0;
  if (tid < C) {
    const unsigned bc = scand[tid];
    my_b = (int)(bc >> 16);
    my_c = (int)(bc & 0xffffu);
    my_ld = st.logden[ssum[my_b]];
    if (my_c < sK[my_b] && my_c != slast[my_b]) my_lb = st.logblk[sblk[my_b * Kmax + my_c]];
  }
  const int Dp = DPT ? DPT : m.Dp;
  const float* pmean = st.pool_mean + (size_t)u * S * Dp;
  const __amdgpu_buffer_rsrc_t rs_mean =
      __builtin_amdgcn_make_buffer_rsrc((void*)st.pool_mean, (short)0, 0x7fffffff, 0x00020000);
  const float* xrow = st.x + (size_t)frame * Dp;
  {
    const int grp = tid >> 4, p = tid & 15;
    if (Dp <= 256) {  // one 256-float chunk: keep the frame in registers, two slots per lane in flight
      f32x4 xv[4], wv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int d = 4 * (p + 16 * k);
        const bool in = d < Dp;
        xv[k] = in ? *reinterpret_cast<const f32x4*>(xrow + d) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        wv[k] = in ? *reinterpret_cast<const f32x4*>(swgt + d) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      }
      for (int i0 = 0; i0 < nlive; i0 += NT / 8) {
        int sl[2]; bool act[2]; int cnt[2]; f32x4 mv[2][4];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const int i = i0 + (NT / 16) * h2 + grp;
          act[h2] = i < nlive;
          sl[h2] = slivelist[act[h2] ? i : 0];
        }
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const float* mean = pmean + (size_t)sl[h2] * Dp;
          cnt[h2] = (p == 0) ? (KEEP ? spcnt[sl[h2]] : st.pool_cnt[(size_t)u * S + sl[h2]]) : 0;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int d = 4 * (p + 16 * k);
            if (d >= Dp) mv[h2][k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            else if (RES) mv[h2][k] = load_sc1(rs_mean, (uint32_t)((((size_t)u * S + sl[h2]) * Dp + d) * 4));
            else mv[h2][k] = *reinterpret_cast<const f32x4*>(mean + d);
          }
        }
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          float A[4] = {0.0f, 0.0f, 0.0f, 0.0f};
          mse16_block(mv[h2], xv, wv, A);  // (chunks past Dp are zero-filled)
          const float d0 = mv[h2][0][0] - xv[0][0];
          const float t = mse16_total(A);
          if (p == 0 && act[h2]) { smse[sl[h2]] = uis_mse_finish(t, d0 * d0, m.D); scnt[sl[h2]] = cnt[h2]; }
        }
      }
    } else {
      for (int i0 = 0; i0 < nlive; i0 += NT / 16) {
        const int i = i0 + grp;
        const bool act = i < nlive;
        const int sl = slivelist[act ? i : 0];
        const float* mean = pmean + (size_t)sl * Dp;
        const int cnt = (p == 0) ? (KEEP ? spcnt[sl] : st.pool_cnt[(size_t)u * S + sl]) : 0;
        float A[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        float first_sq = 0.0f;
        for (int q = 0; q < Dp; q += 256) {
          f32x4 mv[4], xv[4], wv[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int d = q + 4 * (p + 16 * k);
            const bool in = d < Dp;
            if (!in) mv[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            else if (RES) mv[k] = load_sc1(rs_mean, (uint32_t)((((size_t)u * S + sl) * Dp + d) * 4));
            else mv[k] = *reinterpret_cast<const f32x4*>(mean + d);
            xv[k] = in ? *reinterpret_cast<const f32x4*>(xrow + d) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            wv[k] = in ? *reinterpret_cast<const f32x4*>(swgt + d) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
          }
          mse16_block(mv, xv, wv, A);
          if (q == 0) { const float d0 = mv[0][0] - xv[0][0]; first_sq = d0 * d0; }
        }
        const float t = mse16_total(A);
        if (p == 0 && act) { smse[sl] = uis_mse_finish(t, first_sq, m.D); scnt[sl] = cnt; }
      }
    }
  }
  __syncthreads();  // (4) MSE per live slot
  TSTAMP(2);

  // ---- candidate score (thread i = candidate i), finite count
  float my_sc = 0.0f;
  unsigned long long my_key = ~0ull;
  if (tid < C) {
    float mse; double prior;
    if (my_c < sK[my_b]) {
      mse = smse[sslot[my_b * Kmax + my_c]];
      prior = (my_c == slast[my_b]) ? m.lp_stay : (m.lp_sw + my_lb) - my_ld;
    } else {
      mse = mse_new;
      prior = m.lp_new - my_ld;
    }
    my_sc = sscore[my_b] + uis_step_loss(mse, prior);
    if (uis_isfinite(my_sc)) my_key = ((unsigned long long)uis_score_key(my_sc) << 32) | (unsigned)tid;
    if (st.dbg_scores) st.dbg_scores[(((size_t)step * U + u) * B + my_b) * (Kmax + 1) + my_c] = my_sc;
  }
  int keep;
  if (C <= 64) {
    // every candidate is a lane of wave 0: its keys go through LDS without a workgroup barrier
    if (wave == 0) {
      const unsigned long long fm = __ballot(my_key != ~0ull);
      const int nfin = __popcll(fm);
      keep = nfin < B ? nfin : B;
      skey[lane] = my_key;
      if (my_key != ~0ull) {  // rank by counting; keys are unique; two keys per 16-byte LDS read
        int rank = 0;
        const int C2 = (C + 1) & ~1;
#pragma unroll 8
        for (int j2 = 0; j2 < C2; j2 += 2) {
          const ulonglong2 kk = *reinterpret_cast<const ulonglong2*>(skey + j2);
          rank += (kk.x < my_key) + (kk.y < my_key);
        }
        if (rank < keep) { swin[rank] = tid; swscore[rank] = my_sc; }
      }
      if (lane == 0) smisc[1] = nfin;
    }
    TSTAMP(3);
  } else {
    if (tid < 256) skey[tid] = my_key;  // candidates are threads 0..C-1, C <= 256
    {
      const unsigned long long fm = __ballot(my_key != ~0ull);
      if (lane == 0 && fm) atomicAdd(&smisc[1], __popcll(fm));
    }
    __syncthreads();  // (5) keys
    TSTAMP(3);
    const int nfin = smisc[1];
    keep = nfin < B ? nfin : B;
    if (my_key != ~0ull) {  // rank by counting; keys are unique; two keys per 16-byte LDS read
      int rank = 0;
      const int C2 = (C + 1) & ~1;  // skey[C] (if C is odd) holds ~0ull or a non-candidate's ~0ull
#pragma unroll 8
      for (int j2 = 0; j2 < C2; j2 += 2) {
        const ulonglong2 kk = *reinterpret_cast<const ulonglong2*>(skey + j2);
        rank += (kk.x < my_key) + (kk.y < my_key);
      }
      if (rank < keep) { swin[rank] = tid; swscore[rank] = my_sc; }
    }
  }
  __syncthreads();  // (6) winners
  { const int nfin = smisc[1]; keep = nfin < B ? nfin : B; }
  if (wave != 0) {
    // waves 1-3: copy the UNCHANGED entries of every winner's tables (BeamState(source),
    // uisrnn.py:66-69) while wave 0 works out the changed ones
    for (int e = tid - 64; e < keep * Kmax; e += NT - 64) {
      const int rr = e / Kmax, c2 = e - rr * Kmax;
      const unsigned bc = scand[swin[rr]];
      const int rb = (int)(bc >> 16), rc = (int)(bc & 0xffffu);
      const int Knew = sK[rb] + (rc == sK[rb] ? 1 : 0);
      if (c2 < Knew && c2 != rc) {
        if (KEEP) {
          nslot[rr * Kmax + c2] = sslot[rb * Kmax + c2];
          nblk[rr * Kmax + c2] = sblk[rb * Kmax + c2];
        } else {
          st.beam_slot[(bnxt + rr) * Kmax + c2] = sslot[rb * Kmax + c2];
          st.beam_blk[(bnxt + rr) * Kmax + c2] = sblk[rb * Kmax + c2];
        }
      }
    }
    return;
  }
  TSTAMP(4);

  // ---- wave 0: lane r = winner r
  const bool nodedup = (st.flags & 1u) != 0;
  const int r = lane;
  const bool isw = r < keep;
  int wb = 0, wc = 0, src = -2, Kb = 0;
  if (isw) {
    const unsigned bc = scand[swin[r]];
    wb = (int)(bc >> 16);
    wc = (int)(bc & 0xffffu);
    Kb = sK[wb];
    src = wc < Kb ? sslot[wb * Kmax + wc] : -1;
  }
  int lead = r;
  if (!nodedup) {
    // lowest rank with the same source wins: an LDS minimum per source slot, in the MSE area (dead
    // since the scores) -- every winner resets its cell first; one wave, LDS in program order
    uint32_t* cell = src >= 0 ? reinterpret_cast<uint32_t*>(smse) + src : reinterpret_cast<uint32_t*>(smisc) + 4;
    if (isw) *cell = 0xffffffffu;
    asm volatile("" ::: "memory");
    if (isw) __hip_atomic_fetch_min(cell, (uint32_t)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    asm volatile("" ::: "memory");
    if (isw) lead = (int)*cell;
  }
  const bool is_lead = isw && lead == r;
  const unsigned long long lmask = __ballot(is_lead);
  const int nlead = __popcll(lmask);
  const int ord = wave_below(lmask);
  // reserve the rnn rows now; the returned position is needed only at the very end
  int row_base = 0;
  if (lane == 0) row_base = atomicAdd(sink.count, nlead);
  // the ord-th free slot (not referenced by the current beam), in slot order: every free slot knows
  // its own rank among the free ones (bits below it + the chunks before) and files itself under it
  // (in the key area, dead since the ranking; the other waves still read the winners' list)
  int dst = -1;
  {
    int* sfree = reinterpret_cast<int*>(skey);
    int before = 0;
    for (int base = 0; base < S && before < nlead; base += 64) {
      const int sl = base + lane;
      const bool fr = sl < S && slive[sl] == 0;
      const unsigned long long fm = __ballot(fr);
      const int rk = before + wave_below(fm);
      if (fr && rk < nlead) sfree[rk] = sl;
      before += __popcll(fm);
    }
    asm volatile("" ::: "memory");
    if (is_lead && ord < before) dst = sfree[ord];
  }
  {
    const int dl = __shfl(dst, lead, 64);
    if (isw && !is_lead) dst = dl;
  }
  TSTAMP(5);
  if (KEEP) {
    // resident decode: the rows first -- they are all the other workgroups wait for
    row_base = __shfl(row_base, 0, 64);
    if (is_lead) {
      const int nprev = src >= 0 ? scnt[src] : 0;
      spcnt[dst] = nprev + 1;
      RnnRow rr; rr.utt = u; rr.src = src; rr.dst = dst; rr.nprev = nprev; rr.frame = frame; rr.pad = 0;
      sink.rows[row_base + ord] = rr;
    }
    published();
  }
  // the one changed entry of each winner's tables (the rest is being copied by waves 1-3)
  if (isw && wc < Kmax) {
    const bool is_new = wc == Kb;
    const int blk_new = is_new ? 1 : sblk[wb * Kmax + wc] + (wc != slast[wb] ? 1 : 0);
    if (KEEP) {
      nslot[r * Kmax + wc] = dst;
      nblk[r * Kmax + wc] = blk_new;
    } else {
      st.beam_slot[(bnxt + r) * Kmax + wc] = dst;
      st.beam_blk[(bnxt + r) * Kmax + wc] = blk_new;
    }
  }
  int Kmaxseen = 0;
  if (isw) {
    const bool is_new = wc == Kb;
    int Knew = Kb + (is_new ? 1 : 0);
    if (Knew > Kmax) { Knew = Kmax; st.overflow[u] = 1; }
    Kmaxseen = Knew;
    const int sum_new = ssum[wb] + ((is_new || wc != slast[wb]) ? 1 : 0);
    if (KEEP) {
      nK[r] = Knew; nlast[r] = wc; nsum[r] = sum_new; nscore[r] = swscore[r];
    } else {
      st.beam_K[bnxt + r] = Knew;
      st.beam_last[bnxt + r] = wc;
      st.beam_sum[bnxt + r] = sum_new;
    }
    st.beam_score[bnxt + r] = swscore[r];  // (the final beam's scores are read back by k_backtrace)
    st.bp[((size_t)st.tau * off0 + step) * B + r] = ((unsigned)wb << 16) | (unsigned)wc;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { const int o = __shfl_xor(Kmaxseen, off, 64); Kmaxseen = o > Kmaxseen ? o : Kmaxseen; }
  TSTAMP(6);
  if (!KEEP) {
    row_base = __shfl(row_base, 0, 64);
    if (is_lead) {
      const int nprev = src >= 0 ? scnt[src] : 0;
      st.pool_cnt[(size_t)u * S + dst] = nprev + 1;
      RnnRow rr; rr.utt = u; rr.src = src; rr.dst = dst; rr.nprev = nprev; rr.frame = frame; rr.pad = 0;
      sink.rows[row_base + ord] = rr;
    }
  }
  if (lane == 0) {
    st.beam_n[(size_t)nxt * U + u] = keep;
    if (KEEP) *reinterpret_cast<int*>(set_nxt + L.off_nb) = keep;
    else st.utt_step[u] = step + 1;
    atomicMax(&st.counters[3], (unsigned long long)Kmaxseen);
    atomicAdd(&st.counters[0], (unsigned long long)nlead);
    atomicAdd(&st.counters[1], (unsigned long long)keep);
    atomicAdd(&st.counters[2], (unsigned long long)C);
  }
  TSTAMP(7);
}

__global__ __launch_bounds__(256) void k_select_fast(DevModel m, DecodeState st, int par) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int u = blockIdx.x, tid = threadIdx.x;
  if (u == 0 && tid == 0) st.nrows[par ^ 1] = 0;
  select_fast_body<256, false, false>(m, st, par, u, smem_raw, RowSink{st.rows, st.nrows + par});
}

// ------------------------------------------------------------ resident decode
//
// The whole lock-step decode (look_ahead 1, depth 1, rnn_hidden_size 512) in ONE launch with the
// weights held in registers.  256 workgroups of 512 threads, one per CU; workgroup b belongs to
// cluster b % ncl (ncl = CUs / 32 = the number of XCDs: 8 on a whole MI355X) -- the XCD it is
// observed to run on, checked against HW_REG_XCC_ID -- with rank b / ncl.  A cluster decodes
// utterances c, c+ncl, ... on its own: nothing is exchanged between
// XCDs, so everything the 32 workgroups of a cluster hand each other stays in that XCD's L2
// (plain stores, `s_waitcnt vmcnt(0)`, a cluster barrier, sc1 loads that bypass the reader's
// L1; tools/probe_cluster.hip: 0 stale reads, ~0.9 us per barrier).
//
// Rank r owns feature tile r of the GRU and of linear_mean1 and (with 16 linear_mean2 tiles,
// two ranks each taking every other row tile) tile r/2 of linear_mean2; wave w owns K segment
// w, exactly as in splitk_tile, so a thread keeps 48 + 16 + 16 weight registers for the whole
// decode and a step streams only the hypotheses' rows.  Per step: the ranks run the selects of
// their utterances (select_fast_body), barrier, GRU, barrier, linear_mean1, barrier,
// linear_mean2 + running mean, barrier.  Arithmetic and its order are those of the per-step
// kernels (bit-identical, tested).  A barrier that does not complete sets cl_abort instead of
// hanging; the host then reports an error.

#define UIS_RES_RC 3           // row tiles per pass
#define UIS_RES_HEAD_TILES 24  // row tiles whose descriptors are staged in LDS at a time (even)

// The barrier in two halves, so that work that needs nobody else's data can sit between them.
//   xcd_arrive     every wave drains its stores, thread 0 signals (s_ctl[2] = "already arrived")
//   xcd_wait       thread 0 arrives now if nobody did, then polls; ends with the workgroup barrier
__device__ __forceinline__ void xcd_arrive(const DecodeState& st, int cluster, int* s_ctl) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    (void)__hip_atomic_fetch_add(st.rx_bar + cluster * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    s_ctl[2] = 1;
  }
}
// wave 0 only, from inside a select whose other waves have no global stores in flight
__device__ __forceinline__ void xcd_arrive_wave0(const DecodeState& st, int cluster, int* s_ctl) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x == 0) {
    (void)__hip_atomic_fetch_add(st.rx_bar + cluster * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    s_ctl[2] = 1;
  }
}

__device__ __forceinline__ bool xcd_barrier(const DecodeState& st, int cluster, uint32_t target, int* s_abort) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its stores have reached L2
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t* ctr = st.rx_bar + cluster * 32;
    // every participant sits on this XCD (checked), so the arrival is an L2 atomic (no device-scope
    // write-through); the poll is an sc1 load: it skips this CU's L1 and is served by that L2
    // (the arrival's return value is not used -- no round trip before the first poll, which queues
    // behind it at the same L2 channel)
    if (!s_abort[2]) (void)__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    s_abort[2] = 0;
    unsigned spins = 0;
    int bad = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);  // 32 pollers per counter: a short nap between polls measured +1.5 %
      if (++spins > (1u << 21)) {  // ~1 s: give up instead of hanging the device
        __hip_atomic_store(st.cl_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bad = 1;
        break;
      }
      if ((spins & 255u) == 0 && __hip_atomic_load(st.cl_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        bad = 1;
        break;
      }
    }
    *s_abort = bad;
  }
  __syncthreads();
  return *s_abort != 0;
}

// Hand-offs between dense stages without a cluster-wide barrier.  What a consumer wave reads in
// linear_mean1 / linear_mean2 is ITS K-slice of every row -- 1/8 of the features, produced by four
// of the cluster's 32 workgroups (ranks 4w .. 4w + 3 for wave w, whatever the hidden size: a rank's
// feature tile is rank / SH1 and a wave's slice PER = NFT1 / 8 tiles).  So a producer, once its
// stores have reached L2, publishes a phase word (3 step + 1 behind the GRU stage, + 2 behind
// linear_mean1, + 3 behind linear_mean2) and a consumer wave polls the four words of its producers
// -- one 16-byte load -- instead of everybody waiting for the slowest of 32 and for thread 0 to
// tell the rest.  No atomic, no counter: a word has one writer.
__device__ __forceinline__ void rs_flag_publish(uint32_t* flags, int rank, uint32_t phase, bool agent_scope = false) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its stores have reached L2
  __syncthreads();
  // Scope of the store.  The pollers are other workgroups of the SAME XCD (checked at run time,
  // HW_REG_XCC_ID), reading with sc1 (L1 bypass) from the L2 all 32 share.  An agent-scope store is
  // `global_store_dword sc1`: a scalar fabric write that also DROPS the line from that L2
  // (MI355X_MICROARCH.md, "stores of each flavour") -- every poll of 32 workgroups then goes
  // beyond L2.  A plain store stays in the shared L2, which is the point of coherence that matters
  // here; that is outside what the HIP memory model promises for workgroup scope, hence the placement
  // check, the give-up timer and the fallback path.  The by-the-book variant is a
  // run-time choice in one binary (round 6: UIS_FLAG_AGENT_FLAGS / UIS_AGENT_FLAGS=1; -DUIS_RS_FLAG_AGENT
  // still forces it) and its cost is on record (profiles/r06_agent_flags_ab.txt): a decode step of
  // k_decode_rs 25.7 against 18.7 us, of k_decode_resident 29.5 against 22.1 us -- so it stays opt-in.
#if defined(UIS_RS_FLAG_AGENT)
  agent_scope = true;
#endif
  if (threadIdx.x == 0) {
    if (agent_scope) __hip_atomic_store(flags + rank, phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(flags + rank, phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}
// The first look at the phase words, split in two so that the load can be requested early (from
// inside the work a wave does between publishing and waiting) and examined late.
__device__ __forceinline__ u32x4 rs_flag_peek4(__amdgpu_buffer_rsrc_t rs_flags, uint32_t byte_off) {
  return __builtin_bit_cast(u32x4, load_sc1(rs_flags, byte_off));
}
__device__ __forceinline__ bool rs_flag_ready4(const u32x4& f, uint32_t phase) {
  uint32_t mn = f[0] < f[1] ? f[0] : f[1];
  mn = mn < f[2] ? mn : f[2];
  mn = mn < f[3] ? mn : f[3];
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)mn) >= phase;
}
__device__ __forceinline__ uint32_t rs_flag_peek_all(const uint32_t* flags) {
  const int lane = threadIdx.x & 63;
  return __hip_atomic_load(flags + (lane & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool rs_flag_ready_all(uint32_t f, uint32_t phase) { return __ballot(f < phase) == 0ull; }
// true: gave up (a producer never published, or somebody else gave up)
__device__ __forceinline__ bool rs_flag_wait(const DecodeState& st, __amdgpu_buffer_rsrc_t rs_flags, uint32_t byte_off, uint32_t phase) {
  unsigned spins = 0;
  for (;;) {
    const u32x4 f = __builtin_bit_cast(u32x4, load_sc1(rs_flags, byte_off));
    asm volatile("" ::: "memory");  // (a fresh load every round)
    uint32_t mn = f[0] < f[1] ? f[0] : f[1];
    mn = mn < f[2] ? mn : f[2];
    mn = mn < f[3] ? mn : f[3];
    if ((uint32_t)__builtin_amdgcn_readfirstlane((int)mn) >= phase) return false;
    __builtin_amdgcn_s_sleep(1);
    if (++spins > (1u << 21)) {  // ~1 s: give up instead of hanging the device
      __hip_atomic_store(st.cl_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return true;
    }
    if ((spins & 255u) == 0 && __hip_atomic_load(st.cl_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return true;
  }
}
// ... and the step's last hand-off: the select needs every workgroup's partial sums and early MSEs,
// so a wave waits for all 32 words (lane l < 32 looks at producer l's).  Passing it also means every
// workgroup is through with the step's reads, which is what lets the next step overwrite the row
// tiles and reuse freed slots.
__device__ __forceinline__ bool rs_flag_wait_all(const DecodeState& st, const uint32_t* flags, uint32_t phase) {
  const int lane = threadIdx.x & 63;
  unsigned spins = 0;
  for (;;) {
    const uint32_t f = lane < 32 ? __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : phase;
    if (__ballot(f < phase) == 0ull) return false;
    __builtin_amdgcn_s_sleep(1);
    if (++spins > (1u << 21)) {
      __hip_atomic_store(st.cl_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return true;
    }
    if ((spins & 255u) == 0 && __hip_atomic_load(st.cl_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return true;
  }
}

struct RowHead { int utt, src, dst, nprev; };
__device__ __forceinline__ RowHead load_row_head(__amdgpu_buffer_rsrc_t rs_rows, int row) {
  const u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(rs_rows, (uint32_t)row * 32u, 0, 16);
  return RowHead{(int)d[0], (int)d[1], (int)d[2], (int)d[3]};
}
__device__ __forceinline__ RowHead lds_row_head(const u32x4* s_head, int row) {
  const u32x4 d = s_head[row];
  return RowHead{(int)d[0], (int)d[1], (int)d[2], (int)d[3]};
}
__device__ __forceinline__ long load_row_frame(__amdgpu_buffer_rsrc_t rs_rows, int row) {
  const u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(rs_rows, (uint32_t)row * 32u + 16u, 0, 16);
  return (long)(((unsigned long long)d[1] << 32) | d[0]);
}
__device__ __forceinline__ float load_f32_sc1(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one float through a buffer descriptor with a 32-bit byte offset (no 64-bit address arithmetic per lane)
__device__ __forceinline__ float rs_buf_load_f32_sc1(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, byte_off, 0, 16 /* sc1 */));
}
__device__ __forceinline__ void rs_buf_store_f32(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, byte_off, 0, 0);
}
__device__ __forceinline__ void rs_buf_store_f32x4(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, byte_off, 0, 0);
}

// NV (<= RC) row tiles x NG gates of one feature tile, weights from registers: wave w walks its
// K segment (PER k-blocks), partial tiles to LDS [UIS_KSPLIT][RC][NG][256], ends with the
// barrier.  Every CU of the XCD reads all rows (~100 GB/s per CU out of L2): that stream, not
// the MFMA chain, bounds a stage.  Measured and NOT adopted: pinning the request order with
// scheduling fences (k-block-major or row-tile-major) and making every load unconditional so
// that the vmcnt bookkeeping is exact -- both 3-10 % slower end to end than what the compiler
// schedules from this plain form.
// KBS = bytes between a row's consecutive k-blocks (64: a plain row; 1024: the k-block-major
// staging layout).  `after_issue` runs once the row loads are in flight: loads it issues are
// younger in the in-order vmcnt queue, so the MFMA chain never waits for them.
template <int NG, int PER, int RC, int NV, int KBS, typename After>
__device__ __forceinline__ void resident_tile_nv(const f32x4 (&wr)[NG][PER], const float* __restrict__ bias, int gate_stride,
                                                 __amdgpu_buffer_rsrc_t rsrc, const uint32_t (&boff)[RC], float* spart,
                                                 After after_issue) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, q = lane >> 4;
  f32x4 bv[NG];  // oldest in the vmcnt queue: the chain's first operand
#pragma unroll
  for (int g = 0; g < NG; ++g)
    bv[g] = w == 0 ? *reinterpret_cast<const f32x4*>(bias + (size_t)g * gate_stride + 4 * q) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  f32x4 b[NV][PER];
#pragma unroll
  for (int kb = 0; kb < PER; ++kb)
#pragma unroll
    for (int r = 0; r < NV; ++r) b[r][kb] = load_sc1(rsrc, boff[r] + (uint32_t)((w * PER + kb) * KBS + q * 16));
  after_issue();
  f32x4 acc[NV][NG];
#pragma unroll
  for (int r = 0; r < NV; ++r)
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[r][g] = bv[g];
#pragma unroll
  for (int kb = 0; kb < PER; ++kb)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int r = 0; r < NV; ++r)
#pragma unroll
        for (int g = 0; g < NG; ++g)
          acc[r][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[g][kb][e], b[r][kb][e], acc[r][g], 0, 0, 0);
#if defined(UIS_RES_DUP_MFMA)  // diagnostic: the MFMA chain a second time, result discarded
  {
    f32x4 acc2[NV][NG];
#pragma unroll
    for (int r = 0; r < NV; ++r)
#pragma unroll
      for (int g = 0; g < NG; ++g) acc2[r][g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int kb = 0; kb < PER; ++kb)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int r = 0; r < NV; ++r)
#pragma unroll
          for (int g = 0; g < NG; ++g)
            acc2[r][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[g][kb][e], b[r][kb][e], acc2[r][g], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < NV; ++r)
#pragma unroll
      for (int g = 0; g < NG; ++g) asm volatile("" ::"v"(acc2[r][g]));
  }
#endif
#if defined(UIS_RES_DUP_LOAD)  // diagnostic: the row stream a second time, result discarded
  {
#pragma unroll
    for (int kb = 0; kb < PER; ++kb)
#pragma unroll
      for (int r = 0; r < NV; ++r) {
        const f32x4 d = load_sc1(rsrc, boff[r] + (uint32_t)((w * PER + kb) * KBS + q * 16));
        asm volatile("" ::"v"(d));
      }
  }
#endif
#pragma unroll
  for (int r = 0; r < NV; ++r)
#pragma unroll
    for (int g = 0; g < NG; ++g)
      *reinterpret_cast<f32x4*>(spart + ((size_t)((w * RC + r) * NG + g) * 256) + (lane & 15) * 16 + 4 * q) = acc[r][g];
  __syncthreads();
}
// nvalid (wave-uniform, 1..RC) picks the straight-line variant
template <int NG, int PER, int RC, int KBS, typename After>
__device__ __forceinline__ void resident_tile(const f32x4 (&wr)[NG][PER], const float* __restrict__ bias, int gate_stride,
                                              __amdgpu_buffer_rsrc_t rsrc, const uint32_t (&boff)[RC], int nvalid,
                                              float* spart, After after_issue) {
  static_assert(RC == 3, "dispatch below");
  if (nvalid >= 3) resident_tile_nv<NG, PER, RC, 3, KBS>(wr, bias, gate_stride, rsrc, boff, spart, after_issue);
  else if (nvalid == 2) resident_tile_nv<NG, PER, RC, 2, KBS>(wr, bias, gate_stride, rsrc, boff, spart, after_issue);
  else resident_tile_nv<NG, PER, RC, 1, KBS>(wr, bias, gate_stride, rsrc, boff, spart, after_issue);
}

// LDS of k_decode_resident: select area | split-K partial tiles | control words | the rank's
// linear_mean1 / linear_mean2 weight tiles | this step's row descriptors of the cluster (16-byte
// head + 8-byte frame per row)
__host__ __device__ inline size_t resident_lds_bytes(int Hp, int Dp, int B, int Kmax, int S) {
  return (size_t)((fast_lds_layout(Dp, B, Kmax, S).total + 255) & ~255) + (size_t)UIS_KSPLIT * UIS_RES_RC * 3 * 256 * 4 + 64 +
         (size_t)2 * (Hp / 16) * 64 * 16 + (size_t)UIS_RES_HEAD_TILES * 16 * 24;
}

// Diagnostic build (-DUIS_RESIDENT_TIMING): thread 0 of workgroup 0 (runs a select) and of
// workgroup 248 (rank 31: never runs one at 64 utterances) accumulate wall-clock ticks (10 ns)
// per phase into counters[48 + k] / counters[64 + k].
#if defined(UIS_RESIDENT_TIMING)
#define RSTAMP(k) do { if (t == 0) { const unsigned long long now_ = wall_clock64(); rt_acc[k] += now_ - rt_prev; rt_prev = now_; } } while (0)
#else
#define RSTAMP(k) do {} while (0)
#endif
#if defined(UIS_RESIDENT_TIMING)
#define FSTAMP(k) do { if (t == 0) { const unsigned long long now_ = wall_clock64(); ft_acc[k] += now_ - rt_prev2; rt_prev2 = now_; } } while (0)
#else
#define FSTAMP(k) do {} while (0)
#endif

__device__ __forceinline__ void backtrace_body(const DecodeState& st, int u, int32_t* __restrict__ labels,
                                               float* __restrict__ scores, float* __restrict__ beam_scores,
                                               unsigned char* bt_map);

// Loads from / stores to host-coherent pinned memory (the persistent session's mailbox): system
// scope, no cache on the way.
__device__ __forceinline__ uint32_t sys_load_u32(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ int32_t sys_load_i32(const int32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ long long sys_load_i64(const int64_t* p) {
  return __hip_atomic_load(reinterpret_cast<const long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ f32x4 sys_load_f32x4(const float* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// PERSIST: the launch of a persistent streaming session (pm.ctl): after its first push it
// stays on the device and takes further commands (push / labels / quit) from the session's
// mailbox in host-coherent memory, polled by rank 0 of every cluster and passed on through the
// cluster's line of pm.go; it leaves by itself after pm.idle_ticks without a command.
// Beam tables stay in LDS from push to push; needs at most one utterance per workgroup.
// CB, CK: beam_size and max_clusters as compile-time constants (0: run-time values) -- the
// instantiations for the shapes of BASELINE's configs, dispatched for unpadded models only: every
// table offset, division and loop bound that depends on them folds (round 4: k_decode_rs gained 4 %
// from the same substitution).
template <int HP, int DP, bool PERSIST = false, int CB = 0, int CK = 0>
__global__ __launch_bounds__(512) void k_decode_resident(DevModel m, DecodeState st) {
  m.Hp = HP; m.Dp = DP; m.G = 3 * HP;  // (what the template arguments say)
  if (CB) { st.B = CB; st.Kmax = CK; st.S = CB * CK + CB; m.H = HP; m.D = DP; }
  constexpr int NKB = HP / 16, PER = NKB / UIS_KSPLIT, RC = UIS_RES_RC;
  constexpr int NFT1 = HP / 16, SH1 = 32 / NFT1;  // ranks sharing one GRU / linear_mean1 feature tile
  constexpr int NFT2 = DP / 16, SH2 = 32 / NFT2;  // ranks sharing one linear_mean2 feature tile
  constexpr int EPT = (RC + 1) / 2;
  static_assert(NFT1 * SH1 == 32 && NFT2 * SH2 == 32 && PER * UIS_KSPLIT == NKB,
                "hidden size 256 / 512, observation_dim 128 / 256 / 512 (padded)");
  static_assert(UIS_RES_HEAD_TILES % SH1 == 0 && UIS_RES_HEAD_TILES % SH2 == 0, "chunks start at a rank's first tile");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int ncl = st.ncl;  // clusters of 32 workgroups: workgroup b is observed on XCD b % (number of XCDs)
  const int cluster = blockIdx.x % ncl, rank = blockIdx.x / ncl;
  const int U = st.U, S = st.S;
  const FastLds L = fast_lds_layout(m.Dp, st.B, st.Kmax, S);
  float* spart = reinterpret_cast<float*>(smem_raw + ((L.total + 255) & ~255));
  int* s_ctl = reinterpret_cast<int*>(spart + UIS_KSPLIT * RC * 3 * 256);  // [0] abort  [1] steps
  f32x4* s_w1 = reinterpret_cast<f32x4*>(s_ctl + 16);                         // [NKB][64] this rank's linear_mean1 tile
  f32x4* s_w2 = s_w1 + NKB * 64;                                             // [NKB][64] ... linear_mean2 tile
  u32x4* s_head = reinterpret_cast<u32x4*>(s_w2 + NKB * 64);                  // [head tiles x 16] {utt, src, dst, nprev}
  long* s_frame = reinterpret_cast<long*>(s_head + UIS_RES_HEAD_TILES * 16);  // [head tiles x 16]

  uint32_t xcc = 0;
  if (t == 0) {
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xfu;
    if (rank == 0) __hip_atomic_store(st.cl_xcc + cluster, xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_ctl[0] = 0;
    s_ctl[1] = 0;
    s_ctl[2] = 0;  // "this workgroup has already arrived at the barrier it is about to wait at"
    s_ctl[9] = 0;  // a wave gave up waiting for its producers' phase words
    s_ctl[6] = 0;  // PERSIST, rank 0: the host's sequence number of the last command taken
#if defined(UIS_RESIDENT_TIMING) || defined(UIS_RESIDENT_PROBE)
    for (int k = 0; k < 8; ++k) reinterpret_cast<unsigned long long*>(smem_raw + L.off_misc + 64)[k] = 0;
#endif
  }
  __syncthreads();
  // ---- this thread's share of the weights, for the whole decode
  // (hidden size 512: W_hh = 48 registers per thread, the two mean-head tiles 2 x 32 KB of LDS)
  f32x4 wg[3][PER];
  const int ft1 = rank / SH1, tpar1 = rank % SH1;  // ranks sharing a feature tile take alternate row tiles
  const int ft2 = rank / SH2, tpar2 = rank % SH2;
#pragma unroll
  for (int kb = 0; kb < PER; ++kb) {
#pragma unroll
    for (int g = 0; g < 3; ++g)
      wg[g][kb] = reinterpret_cast<const f32x4*>(m.whh[0])[((size_t)(g * NFT1 + ft1) * NKB + w * PER + kb) * 64 + lane];
    s_w1[(w * PER + kb) * 64 + lane] = reinterpret_cast<const f32x4*>(m.w1)[((size_t)ft1 * NKB + w * PER + kb) * 64 + lane];
    s_w2[(w * PER + kb) * 64 + lane] = reinterpret_cast<const f32x4*>(m.w2)[((size_t)ft2 * NKB + w * PER + kb) * 64 + lane];
  }

  const __amdgpu_buffer_rsrc_t rs_rows =
      __builtin_amdgcn_make_buffer_rsrc((void*)st.rows, (short)0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_hid =
      __builtin_amdgcn_make_buffer_rsrc((void*)st.pool_hid, (short)0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a1 =
      __builtin_amdgcn_make_buffer_rsrc((void*)st.a1, (short)0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_hst =
      __builtin_amdgcn_make_buffer_rsrc((void*)st.gi_up, (short)0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_mean =
      __builtin_amdgcn_make_buffer_rsrc((void*)st.pool_mean, (short)0, 0x7fffffff, 0x00020000);
  // Round 4: GRU -> linear_mean1 -> linear_mean2 hand over through per-producer phase words, as in
  // k_decode_rs (rs_flag_publish / rs_flag_wait): a consumer wave reads ITS K-slice of every row -- what
  // ranks 4 w .. 4 w + 3 produced -- so it waits for those four workgroups, not for the slowest of 32.
  // The barriers behind linear_mean2 and behind the select stay cluster-wide (every workgroup needs
  // every mean / every row), which is also what makes the reuse of the staging buffers safe.
  // UIS_FLAG_CLUSTER_BARRIERS keeps the two barriers (A/B switch, bit-identical).
  uint32_t* const flags_c = st.rx_flags + cluster * 32;
  const __amdgpu_buffer_rsrc_t rs_flags = __builtin_amdgcn_make_buffer_rsrc((void*)flags_c, (short)0, 128, 0x00020000);
  const bool flag_handoff = (st.flags & 0x8000u) == 0u;
  uint32_t fphase = 0;  // hand-offs of this launch so far (every workgroup of the cluster counts the same)
  // hand-off buffers between the stages (h' -> linear_mean1, a1 -> linear_mean2), k-block major:
  // [row tile][k block = the producer's feature tile][16 rows][16] -- a producer tile is one
  // contiguous KiB and so is a consumer wave's 16-byte-per-lane load
  const size_t tile0 = (size_t)(cluster * st.rx_stride) >> 4;  // first row tile of this cluster
  const int rbase = cluster * st.rx_stride;
  const uint32_t h1_off = (uint32_t)((size_t)U * S * HP * 4);  // the extra slot holding h1
  RowSink sink{st.rows + rbase, nullptr};
  uint32_t bar = 0;
  const bool keep_beam = U <= 32 * ncl;
  const bool did_select = cluster + ncl * rank < U;  // keep_beam: this workgroup owns an utterance
  long my_off0 = 0, my_off1 = 0;
  if (keep_beam && did_select) {
    my_off0 = (long)st.off[cluster + ncl * rank];
    my_off1 = (long)st.off[cluster + ncl * rank + 1];
  }
  // streaming: the beam tables (LDS) go back to their global copies when the launch ends
  auto write_back = [&](int final_step) {
    const int u = cluster + ncl * rank;
    const int B = st.B, Kmax = st.Kmax;
    const int cur = final_step & 1;
    const unsigned char* set = smem_raw + cur * L.set_stride;
    const size_t bb = ((size_t)cur * U + u) * B;
    for (int e = t; e < B * Kmax; e += 512) {
      st.beam_slot[bb * Kmax + e] = reinterpret_cast<const int*>(set + L.off_slot)[e];
      st.beam_blk[bb * Kmax + e] = reinterpret_cast<const int*>(set + L.off_blk)[e];
    }
    for (int b2 = t; b2 < B; b2 += 512) {
      st.beam_K[bb + b2] = reinterpret_cast<const int*>(set + L.off_K)[b2];
      st.beam_last[bb + b2] = reinterpret_cast<const int*>(set + L.off_last)[b2];
      st.beam_sum[bb + b2] = reinterpret_cast<const int*>(set + L.off_sum)[b2];
    }
    const int* spc = reinterpret_cast<const int*>(smem_raw + L.off_pcnt);
    for (int sl = t; sl < S; sl += 512) st.pool_cnt[(size_t)u * S + sl] = spc[sl];
    if (t == 0) st.utt_step[u] = final_step;
  };
  // PERSIST: the session's per-push tables are this cluster's device copies, filled by rank 0
  // (read through the pointer where needed -- all of it outside the step loop)
  const PersistArgs& pm = *st.pm;
  // PERSIST: a launch that gives up at an in-launch barrier tells the host so (LEFT word 3): the
  // host's command loop then stops waiting for a completion that will not come
  auto left_aborted = [&]() {
    if (PERSIST && rank == 0 && t == 0)
      __hip_atomic_store(pm.ctl + UIS_PM_LEFT_WORD + 16 * cluster, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  };
  if (PERSIST) {
    st.foff = reinterpret_cast<const int64_t*>(pm.hdr + (size_t)cluster * pm.hdr_stride);
    st.avail = reinterpret_cast<const int32_t*>(pm.hdr + (size_t)cluster * pm.hdr_stride + (size_t)U * 8);
  }
  uint32_t pseq = 0;    // PERSIST: commands taken so far (the cluster's own count, passed on through pm_go)
  uint32_t gstep = 0;   // PERSIST: steps run by earlier pushes (parity of the cluster's row counters)
  int launch_step0 = 0, my_cur = 0;  // PERSIST: the owned utterance's step count at launch / now
  if (PERSIST && did_select) launch_step0 = my_cur = st.utt_step[cluster + ncl * rank];
  uint32_t ctype = UIS_PM_PUSH;
#if defined(UIS_RESIDENT_TIMING)
  unsigned long long rt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rt_prev = 0, ft_acc[4] = {0, 0, 0, 0}, rt_prev2 = 0;
#endif
#if defined(UIS_PM_TIMING)  // diagnostic: phases of a push as workgroup 0 sees them, 10 ns ticks, into the mailbox
  unsigned long long pm_t = 0, pm_acc[6] = {0, 0, 0, 0, 0, 0};
#define PMSTAMP(k) do { if (PERSIST && blockIdx.x == 0 && t == 0) { const unsigned long long n_ = wall_clock64(); pm_acc[k] += n_ - pm_t; pm_t = n_; } } while (0)
#else
#define PMSTAMP(k) do {} while (0)
#endif
  for (;;) {
  int push_F = st.push_F;
  int prow0 = 0, prows = 0;  // PERSIST: this cluster's rows of the push
  if (PERSIST) {
    // ---- the next command.  Rank 0 polls the mailbox (host memory), brings a push's tables and
    // this cluster's new frames onto the device and passes the command on through pm_go (this
    // XCD's L2); everybody drops the CU's L1 -- chunk buffers and step counters were rewritten.
    ++pseq;
    unsigned long long* go = pm.go + cluster * 16;
    if (rank == 0) {
      if (t == 0) {
        const unsigned long long t0 = wall_clock64();
        uint32_t ty = UIS_PM_IDLE, nf = 0u, polls = 0u;
        for (;;) {
          // this cluster's doorbell line in one 16-byte read (the host writes the sequence number last)
          u32x4 bell;
          asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(bell) : "v"(pm.ctl + UIS_PM_BELL_WORD + 16 * cluster) : "memory");
          if (bell[0] != (uint32_t)s_ctl[6] && bell[3] == bell[0]) {  // (word 3 = the number again, written first: no torn line)
            ty = bell[1] & 0xffu; nf = bell[1] >> 8;
            s_ctl[6] = (int)bell[0]; s_ctl[7] = (int)(bell[2] & 0xffffu); s_ctl[8] = (int)(bell[2] >> 16);
#if defined(UIS_PM_TIMING)
            if (blockIdx.x == 0) pm_t = wall_clock64();
#endif
            break;
          }
          if (wall_clock64() - t0 > pm.idle_ticks) break;
          // a failed placement / barrier check anywhere on the device: leave (first poll and every 64th)
          if ((polls++ & 63u) == 0 && __hip_atomic_load(st.cl_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ty = UIS_PM_QUIT; break; }
          __builtin_amdgcn_s_sleep(8);
        }
        s_ctl[4] = (int)ty; s_ctl[5] = (int)nf;
      }
      __syncthreads();
      ctype = (uint32_t)s_ctl[4]; push_F = s_ctl[5];
      if (ctype == UIS_PM_PUSH) {
        // the host packs a push cluster by cluster: this cluster's new frames are rows
        // [prow0, prow0 + prows) of the chunk -- tables and frames cross PCIe in ONE round trip
        prow0 = s_ctl[7]; prows = s_ctl[8];
        const int q4 = m.D >> 2;  // (the host takes this path only when D == Dp)
        float* xdev = const_cast<float*>(st.x);
        int64_t* foff_c = const_cast<int64_t*>(st.foff);
        int32_t* avail_c = const_cast<int32_t*>(st.avail);
        for (int e0 = t; e0 < prows * q4; e0 += 4 * 512) {  // four PCIe reads in flight per thread
          f32x4 v[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int e = e0 + k * 512;
            v[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (e < prows * q4) {  // (no request for what is past the end: a one-frame push is ONE read per thread)
              const float* src = pm.frames + (size_t)(prow0 + e / q4) * m.D + (e % q4) * 4;
              asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "+v"(v[k]) : "v"(src) : "memory");
            }
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int e = e0 + k * 512;
            asm volatile("" : "+v"(v[k]));  // (the values exist only after the wait above)
            if (e < prows * q4) *reinterpret_cast<f32x4*>(xdev + (size_t)(prow0 + e / q4) * m.Dp + (e % q4) * 4) = v[k];
          }
        }
        for (int u = t; u < U; u += 512) { foff_c[u] = sys_load_i64(pm.foff + u); avail_c[u] = sys_load_i32(pm.avail + u); }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      PMSTAMP(0);  // tables and frames fetched from the host
      if (t == 0) {  // ONE 64-bit atomic: sequence number (16 bits) | command (4) | frames (12) | first row (16) | rows (16)
        const unsigned long long word = (unsigned long long)(pseq & 0xffffu) | ((unsigned long long)(ctype & 0xfu) << 16) |
                                        ((unsigned long long)((uint32_t)push_F & 0xfffu) << 20) |
                                        ((unsigned long long)((uint32_t)prow0 & 0xffffu) << 32) | ((unsigned long long)((uint32_t)prows & 0xffffu) << 48);
        __hip_atomic_store(go, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __builtin_amdgcn_s_dcache_inv();  // (uniform addresses are read through the scalar cache: tables, step counters)
      }
      __syncthreads();
    } else {
      if (t == 0) {
        const unsigned long long t0 = wall_clock64();
        unsigned long long v = 0;
        uint32_t ty = UIS_PM_QUIT;
        for (;;) {
          v = __hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((uint32_t)(v & 0xffffu) == (pseq & 0xffffu)) { ty = (uint32_t)(v >> 16) & 0xfu; break; }
          if (wall_clock64() - t0 > 4 * pm.idle_ticks + 200000000ull) {  // rank 0 went missing: give up
            __hip_atomic_store(st.cl_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
          __builtin_amdgcn_s_sleep(8);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __builtin_amdgcn_s_dcache_inv();  // (uniform addresses are read through the scalar cache: tables, step counters)
        s_ctl[4] = (int)ty; s_ctl[5] = (int)((v >> 20) & 0xfffu); s_ctl[7] = (int)((v >> 32) & 0xffffu); s_ctl[8] = (int)(v >> 48);
      }
      __syncthreads();
      ctype = (uint32_t)s_ctl[4]; push_F = s_ctl[5]; prow0 = s_ctl[7]; prows = s_ctl[8];
    }
    if (ctype != UIS_PM_PUSH && ctype != UIS_PM_LABELS) break;
    if (ctype == UIS_PM_LABELS) {
      // the owned utterance's labels / scores / cap flag straight into the mailbox
      if (did_select) {
        const int u = cluster + ncl * rank;
        DecodeState sl = st;
        sl.lab_off = pm.lab_off;
        backtrace_body(sl, u, pm.labels, pm.scores, pm.beam_scores, reinterpret_cast<unsigned char*>(spart));
        if (t == 0) pm.overflow[u] = st.overflow[u];
      }
      if (xcd_barrier(st, cluster, 32u * ++bar, s_ctl)) { left_aborted(); return; }
      if (rank == 0 && t == 0) {
        __threadfence_system();
        __hip_atomic_store(pm.ctl + UIS_PM_DONE_WORD + 16 * cluster, (uint32_t)s_ctl[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      continue;
    }
  }
  PMSTAMP(1);  // command passed on, L1 dropped
  {  // decode steps of this cluster = the longest of its utterances
    if (t == 0) s_ctl[1] = 0;
    __syncthreads();
    int myT = 0;
    for (int i = t; cluster + ncl * i < U; i += 512) {
      const int u = cluster + ncl * i;
      // streaming (uis_stream_push): the steps this utterance can run now = frames received - steps done
      const long T = st.avail ? (long)st.avail[u] - (long)st.utt_step[u]
                              : (long)st.tau * (long)(st.off[u + 1] - st.off[u]);
      myT = T > myT ? (int)T : myT;
    }
    if (myT > 0) atomicMax(&s_ctl[1], myT);
  }
  __syncthreads();
  // (round 5) an ordinary decode in several launches (DecodeState::step0 / step1, with at most one utterance per
  // workgroup): this launch runs steps [step0, step0 + nsteps) of every utterance; like a streaming launch it fetches the
  // beam tables from their global copies at its first step and, if steps are left, writes them back at its end
  const int total_steps = s_ctl[1];
  const int range0 = (!PERSIST && !st.avail) ? st.step0 : 0;
  const int range1 = (!PERSIST && !st.avail && st.step1 > 0 && st.step1 < total_steps) ? st.step1 : total_steps;
  const int nsteps = (!PERSIST && !st.avail) ? (range1 > range0 ? range1 - range0 : 0) : total_steps;
  // a streaming session (st.avail): the utterance continues at its own step count; the beam
  // tables are fetched from their global copies at the launch's first step and written back at its end
  int my_step0 = range0;
  if (keep_beam && did_select && st.avail) my_step0 = PERSIST ? my_cur : st.utt_step[cluster + ncl * rank];
  const int first_step = PERSIST ? launch_step0 : my_step0;
  if (push_F > 0) {
    // ---- streaming push: the chunk's once-per-frame work, by this cluster for its own utterances'
    // new frames -- gi0 = W_ih0 x + b_ih0 (k_dense_input_proj's fullk_tile per 16 rows x 16
    // features, one wave each) and mse0 (k_mse0's wave_weighted_mse, one wave per frame); same
    // arithmetic, same order.  The chunk rows are listed in LDS first (the descriptor area is
    // not in use yet).
    int* s_prow = reinterpret_cast<int*>(s_head);
    const int prow_cap = UIS_RES_HEAD_TILES * 16 * 6;
    if (!PERSIST) {
      // (every thread fetches one utterance's three words; thread 0 then lists from LDS)
      const int ncu = (U - cluster + ncl - 1) / ncl, lcap = UIS_KSPLIT * UIS_RES_RC * 3 * 256 / 4;
      int* l_s0 = reinterpret_cast<int*>(spart);
      int* l_a = l_s0 + lcap;
      int* l_f0 = l_a + lcap;
      const bool staged = ncu <= lcap;
      if (staged)
        for (int i = t; i < ncu; i += 512) {
          const int u = cluster + ncl * i;
          l_s0[i] = st.utt_step[u]; l_a[i] = st.avail[u]; l_f0[i] = (int)st.foff[u];
        }
      __syncthreads();
      if (t == 0) {
        int n = 0;
        for (int i = 0; i < ncu; ++i) {
          const int u = cluster + ncl * i;
          const int s0 = staged ? l_s0[i] : st.utt_step[u], a = staged ? l_a[i] : st.avail[u];
          const long f0 = staged ? (long)l_f0[i] : (long)st.foff[u];
          for (int k = s0; k < a && n < prow_cap; ++k) s_prow[n++] = (int)(f0 + k);
        }
        s_ctl[3] = n;
      }
      __syncthreads();
    }
    // PERSIST: the host packs a push cluster by cluster, the rows are [prow0, prow0 + prows)
    const int R = PERSIST ? prows : s_ctl[3];
    auto chunk_row = [&](int li) -> size_t { return PERSIST ? (size_t)(prow0 + li) : (size_t)s_prow[li]; };
    // With the beam in LDS and workgroups to spare, the projection does not get a barrier of its
    // own: the workgroups that own an utterance compute mse0 of THEIR rows (all their first select
    // needs) and go on to select, the others compute gi0 and meet them at the first step's barrier.
    const int nown = keep_beam ? ((U - cluster + ncl - 1) / ncl < 32 ? (U - cluster + ncl - 1) / ncl : 32) : 32;
    const bool overlap = keep_beam && nown < 32;
    if (R > 0) {
      const int NGT = m.G / 16;
      const int prt = (R + 15) >> 4;
      float* gi0w = const_cast<float*>(st.gi0);
      float* mse0w = const_cast<float*>(st.mse0);
      const int slot0 = overlap ? (rank - nown) * 8 + w : rank * 8 + w, nslots = overlap ? (32 - nown) * 8 : 256;
      if (!overlap || !did_select) {
        for (int task = slot0; task < prt * NGT; task += nslots) {
          const int rt = task / NGT, ft = task - rt * NGT;
          int li = rt * 16 + (lane & 15);
          const bool valid = li < R;
          if (!valid) li = R - 1;
          const size_t row = chunk_row(li);
          const f32x4 v = fullk_tile(m.wih[0], ft, m.Dp / 16, st.x + row * m.Dp, m.bih[0] + ft * 16);
          if (valid) *reinterpret_cast<f32x4*>(gi0w + row * m.G + ft * 16 + (lane >> 4) * 4) = v;
        }
      }
      if (!overlap) {
        for (int i = rank * 8 + w; i < R; i += 256) {
          const size_t row = chunk_row(i);
          const float v = wave_weighted_mse(m.m0, st.x + row * m.Dp, m.wgt, m.Dp, m.D, lane);
          if (lane == 0) mse0w[row] = v;
        }
      } else if (did_select) {
        const int u = cluster + ncl * rank;
        const long f0 = (long)st.foff[u];
        const int a = st.avail[u];
        for (int k = my_step0 + w; k < a; k += 8) {
          const size_t row = (size_t)(f0 + k);
          const float v = wave_weighted_mse(m.m0, st.x + row * m.Dp, m.wgt, m.Dp, m.D, lane);
          if (lane == 0) mse0w[row] = v;
        }
      }
    }
    if (!overlap) {
      if (xcd_barrier(st, cluster, 32u * ++bar, s_ctl)) { left_aborted(); return; }
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the owner reads its mse0 back from L2)
      __syncthreads();
    }
  }
  PMSTAMP(2);  // steps counted, gi0 / mse0 of the chunk, barrier
  if (PERSIST && did_select) {  // the owned utterance's step count after this push
    long done = (long)st.avail[cluster + ncl * rank] - my_step0;
    done = done < 0 ? 0 : (done > nsteps ? nsteps : done);
    my_cur = my_step0 + (int)done;
  }
#if defined(UIS_RESIDENT_TIMING)
  rt_prev = rt_prev2 = wall_clock64();
#endif

  for (int s = 0; s < nsteps; ++s) {
    const int par = PERSIST ? (int)((gstep + (uint32_t)s) & 1u) : ((range0 + s) & 1);
    sink.count = st.rx_nrows + cluster * 32 + par;
#if defined(UIS_RESIDENT_PROBE)  // diagnostic: dependent-load latencies seen by thread 0 at the top of a step
    if (t == 0 && blockIdx.x == 0) {
      unsigned long long* pa = reinterpret_cast<unsigned long long*>(smem_raw + L.off_misc + 64);
      const unsigned long long c0 = __builtin_readcyclecounter();
      const int v0 = __builtin_nontemporal_load(st.utt_step + cluster);          // written by this CU last step
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long c1 = __builtin_readcyclecounter();
      const float v1 = __hip_atomic_load(st.pool_mean + ((size_t)cluster * S) * m.Dp + (s & 63) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long c2 = __builtin_readcyclecounter();
      const float v2 = m.wgt[(s & 7) * 32];                                      // immutable, tiny, hot
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long c3 = __builtin_readcyclecounter();
      const float v3 = m.wgt[(s & 7) * 32 + 1];                                  // same line again: L1 hit
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long c4 = __builtin_readcyclecounter();
      asm volatile("" ::"v"(v0), "v"(v1), "v"(v2), "v"(v3));
      pa[0] += c1 - c0; pa[1] += c2 - c1; pa[2] += c3 - c2; pa[3] += c4 - c3;
    }
#endif
    const int ustep = my_step0 + s, upar = ustep & 1;  // the utterance's own step and table parity
    if (keep_beam) {  // at most one utterance per workgroup: its beam lives in LDS
      if (did_select) {
        if (s == 0) {  // (later steps: prepared while waiting for the previous step's last barrier)
          select_fast_body<512, true, true, DP, 5>(m, st, upar, cluster + ncl * rank, smem_raw, sink, ustep, my_off0, my_off1,
                                                   SelectNoHook(), first_step);
          __syncthreads();
        }
        select_fast_body<512, true, true, DP, 2>(m, st, upar, cluster + ncl * rank, smem_raw, sink, ustep, my_off0, my_off1,
                                                 [&]() { xcd_arrive_wave0(st, cluster, s_ctl); }, first_step);
      }
      __syncthreads();
    } else {
      for (int i = rank; cluster + ncl * i < U; i += 32) {
        select_fast_body<512, true, false, DP>(m, st, par, cluster + ncl * i, smem_raw, sink);
        __syncthreads();
      }
    }
    RSTAMP(0);
    if (xcd_barrier(st, cluster, 32u * ++bar, s_ctl)) { left_aborted(); return; }
    RSTAMP(1);
    // PERSIST: the step count after this push goes to its global word (the next command reads it)
    // once every workgroup has derived nsteps from the old value, i.e. behind this barrier; the
    // next barrier drains the store
    if (PERSIST && s == 0 && did_select && t == 0) st.utt_step[cluster + ncl * rank] = my_cur;
    if (s == 0 && t == 0 && rank == 1 && (st.flags & 0x100u)) xcc ^= 1u;  // UIS_FLAG_TEST_MISPLACED: pretend
    if (s == 0 && t == 0 && __hip_atomic_load(st.cl_xcc + cluster, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != xcc)
      __hip_atomic_store(st.cl_abort, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // not on one XCD
    const int nrows = __hip_atomic_load(st.rx_nrows + cluster * 32 + par, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (rank == 0 && t == 0)
      __hip_atomic_store(st.rx_nrows + cluster * 32 + (par ^ 1), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int nrt = (nrows + 15) >> 4;
    // This step's row descriptors go to LDS, UIS_RES_HEAD_TILES row tiles at a time (rows past
    // nrows: stale but valid).  Up to that many tiles -- the common case -- they are staged once
    // for the three stages; beyond, every stage walks the row tiles in chunks and re-stages.
    auto stage_heads = [&](int c0) {
      const int r0 = 16 * c0, r1 = 16 * (nrt < c0 + UIS_RES_HEAD_TILES ? nrt : c0 + UIS_RES_HEAD_TILES);
      for (int row = r0 + t; row < r1; row += 512) {
        s_head[row - r0] = __builtin_amdgcn_raw_buffer_load_b128(rs_rows, (uint32_t)(rbase + row) * 32u, 0, 16);
        s_frame[row - r0] = load_row_frame(rs_rows, rbase + row);
      }
      __syncthreads();
    };
    const bool single = nrt <= UIS_RES_HEAD_TILES;
    if (single) stage_heads(0);

    // ---- GRU: h' = gru(gi0[frame], W_hh h_src + b_hh) -> dst slot
    for (int c0 = 0; c0 < nrt; c0 += UIS_RES_HEAD_TILES) {
      if (!single) stage_heads(c0);
      const int c1 = nrt < c0 + UIS_RES_HEAD_TILES ? nrt : c0 + UIS_RES_HEAD_TILES;
      // of the chunk this rank takes the row tiles c0 + tpar1, c0 + tpar1 + SH1, ...: index i
      const int my1 = c1 - c0 > tpar1 ? (c1 - c0 - tpar1 + SH1 - 1) / SH1 : 0;
      for (int i0 = 0; i0 < my1; i0 += RC) {
        uint32_t boff[RC];
#pragma unroll
        for (int r = 0; r < RC; ++r) {
          const int tile = c0 + tpar1 + SH1 * (i0 + r < my1 ? i0 + r : i0);
          const RowHead rh = lds_row_head(s_head, 16 * (tile - c0) + (t & 15));
          boff[r] = rh.src >= 0 ? (uint32_t)((((size_t)rh.utt * S + rh.src) * HP) * 4) : h1_off;
        }
        const int j = ft1 * 16 + (t & 15);
        RowHead re[EPT];
        float gir[EPT], giz[EPT], gin[EPT], hprev[EPT];
        bool ework[EPT];
        auto epilogue_operands = [&]() {
#pragma unroll
          for (int k = 0; k < EPT; ++k) {
            const int r = (t >> 8) + 2 * k;
            const int lrow = 16 * (c0 + tpar1 + SH1 * (i0 + r)) + ((t & 255) >> 4);
            ework[k] = r < RC && i0 + r < my1 && lrow < nrows;
            gir[k] = giz[k] = gin[k] = hprev[k] = 0.0f;
            re[k] = RowHead{0, 0, 0, 0};
            if (ework[k]) {  // (the branch-free form that pays in k_decode_rs measured 1-2 % slower here: round 4)
              re[k] = lds_row_head(s_head, lrow - 16 * c0);
              const long frame = s_frame[lrow - 16 * c0];
              const float* gi = st.gi0 + (size_t)frame * (3 * HP);
              gir[k] = gi[j]; giz[k] = gi[HP + j]; gin[k] = gi[2 * HP + j];
              hprev[k] = rs_buf_load_f32_sc1(rs_hid, (uint32_t)(((re[k].src >= 0 ? re[k].utt * S + re[k].src : U * S) * HP + j) * 4));
            }
          }
        };
        FSTAMP(0);
        resident_tile<3, PER, RC, 64>(wg, m.bhh[0] + ft1 * 16, HP, rs_hid, boff, my1 - i0 < RC ? my1 - i0 : RC, spart,
                                      epilogue_operands);
        FSTAMP(1);
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
          if (!ework[k]) continue;
          const int r = (t >> 8) + 2 * k, e = t & 255;
          const float ghr = splitk_combine<RC, 3>(spart, r, 0, e);
          const float ghz = splitk_combine<RC, 3>(spart, r, 1, e);
          const float ghn = splitk_combine<RC, 3>(spart, r, 2, e);
          const float out = j < m.H ? uis_gru_unit(gir[k], giz[k], gin[k], ghr, ghz, ghn, hprev[k]) : 0.0f;
          rs_buf_store_f32(rs_hid, (uint32_t)(((re[k].utt * S + re[k].dst) * HP + j) * 4), out);
          rs_buf_store_f32(rs_hst, (uint32_t)((((int)tile0 + c0 + tpar1 + SH1 * (i0 + r)) * NFT1 + ft1) * 256 + e) * 4u, out);  // the copy linear_mean1 streams
        }
        FSTAMP(2);
        __syncthreads();  // spart (and, chunked, the descriptors) are reused
        FSTAMP(3);
      }
    }
    RSTAMP(2);
    const bool prep_next = keep_beam && did_select && s + 1 < nsteps;
    ++fphase;
    if (flag_handoff) {
      // publish; the first half of the next step's select preparation (this workgroup's own LDS tables:
      // nobody else's data); then every wave waits for the four producers of its K-slice
      rs_flag_publish(flags_c, rank, fphase, (st.flags & 0x20000u) != 0u);
      if (prep_next)
        select_fast_body<512, true, true, DP, 1>(m, st, upar ^ 1, cluster + ncl * rank, smem_raw, sink, ustep + 1, my_off0, my_off1,
                                                 SelectNoHook(), first_step);
      if (nrt > tpar1 && !rs_flag_ready4(rs_flag_peek4(rs_flags, (uint32_t)(16 * w)), fphase) &&
          rs_flag_wait(st, rs_flags, (uint32_t)(16 * w), fphase))
        s_ctl[9] = 1;  // gave up (cl_abort is set): every wave leaves at the next cluster barrier
    } else {
      if (prep_next) {
        // arrive, do the first half of the next step's select preparation, then wait
        xcd_arrive(st, cluster, s_ctl);
        select_fast_body<512, true, true, DP, 1>(m, st, upar ^ 1, cluster + ncl * rank, smem_raw, sink, ustep + 1, my_off0, my_off1,
                                                 SelectNoHook(), first_step);
      }
      if (xcd_barrier(st, cluster, 32u * ++bar, s_ctl)) { left_aborted(); return; }
    }
    RSTAMP(3);

    // ---- linear_mean1 + relu -> a1 (needs no descriptors: row tile in, row tile out)
    const int my1h = nrt > tpar1 ? (nrt - tpar1 + SH1 - 1) / SH1 : 0;  // this rank's row tiles tpar1, tpar1 + SH1, ...
    for (int i0 = 0; i0 < my1h; i0 += RC) {
      uint32_t boff[RC];
#pragma unroll
      for (int r = 0; r < RC; ++r) {
        const int tile = tpar1 + SH1 * (i0 + r < my1h ? i0 + r : i0);
        boff[r] = (uint32_t)((((tile0 + tile) * NFT1) * 256 + (t & 15) * 16) * 4);
      }
      f32x4 w1r[1][PER];
#pragma unroll
      for (int kb = 0; kb < PER; ++kb) w1r[0][kb] = s_w1[(w * PER + kb) * 64 + lane];
      resident_tile<1, PER, RC, 1024>(w1r, m.b1 + ft1 * 16, 0, rs_hst, boff, my1h - i0 < RC ? my1h - i0 : RC, spart, []() {});
      for (int e = t; e < RC * 256; e += 512) {
        const int r = e >> 8, tile = tpar1 + SH1 * (i0 + r), lrow = 16 * tile + ((e & 255) >> 4);
        if (i0 + r < my1h && lrow < nrows) {
          const float v = splitk_combine<RC, 1>(spart, r, 0, e & 255);
          rs_buf_store_f32(rs_a1, (uint32_t)((((int)tile0 + tile) * NFT1 + ft1) * 256 + (e & 255)) * 4u, v > 0.0f ? v : 0.0f);
        }
      }
      __syncthreads();
    }
    RSTAMP(4);
    ++fphase;
    if (flag_handoff) {  // ... and the second half inside the next hand-off
      rs_flag_publish(flags_c, rank, fphase, (st.flags & 0x20000u) != 0u);
      if (prep_next)
        select_fast_body<512, true, true, DP, 4>(m, st, upar ^ 1, cluster + ncl * rank, smem_raw, sink, ustep + 1, my_off0, my_off1,
                                                 SelectNoHook(), first_step);
      if (nrt > tpar2 && !rs_flag_ready4(rs_flag_peek4(rs_flags, (uint32_t)(16 * w)), fphase) &&
          rs_flag_wait(st, rs_flags, (uint32_t)(16 * w), fphase))
        s_ctl[9] = 1;
    } else {
      if (prep_next) {
        xcd_arrive(st, cluster, s_ctl);
        select_fast_body<512, true, true, DP, 4>(m, st, upar ^ 1, cluster + ncl * rank, smem_raw, sink, ustep + 1, my_off0, my_off1,
                                                 SelectNoHook(), first_step);
      }
      if (xcd_barrier(st, cluster, 32u * ++bar, s_ctl)) { left_aborted(); return; }
    }
    RSTAMP(5);

    // ---- linear_mean2 + running mean -> dst slot; of every chunk this rank takes the row tiles
    // c0 + tpar2, c0 + tpar2 + SH2, ... (chunks start at even tiles)
    for (int c0 = 0; c0 < nrt; c0 += UIS_RES_HEAD_TILES) {
      if (!single) stage_heads(c0);
      const int c1 = nrt < c0 + UIS_RES_HEAD_TILES ? nrt : c0 + UIS_RES_HEAD_TILES;
      const int my_tiles = c1 - c0 > tpar2 ? (c1 - c0 - tpar2 + SH2 - 1) / SH2 : 0;
      for (int i0 = 0; i0 < my_tiles; i0 += RC) {
        uint32_t boff[RC];
#pragma unroll
        for (int r = 0; r < RC; ++r) {
          const int tile = c0 + tpar2 + SH2 * (i0 + r < my_tiles ? i0 + r : i0);
          boff[r] = (uint32_t)((((tile0 + tile) * NFT1) * 256 + (t & 15) * 16) * 4);
        }
        const int f = ft2 * 16 + (t & 15);
        RowHead re[EPT];
        float old[EPT];
        bool ework[EPT];
        auto epilogue_operands = [&]() {
#pragma unroll
          for (int k = 0; k < EPT; ++k) {
            const int r = (t >> 8) + 2 * k;
            const int lrow = 16 * (c0 + tpar2 + SH2 * (i0 + r)) + ((t & 255) >> 4);
            ework[k] = r < RC && i0 + r < my_tiles && lrow < nrows;
            old[k] = 0.0f;
            re[k] = RowHead{0, 0, 0, 0};
            if (ework[k]) {
              re[k] = lds_row_head(s_head, lrow - 16 * c0);
              if (re[k].src >= 0) old[k] = rs_buf_load_f32_sc1(rs_mean, (uint32_t)(((re[k].utt * S + re[k].src) * DP + f) * 4));
            }
          }
        };
        f32x4 w2r[1][PER];
#pragma unroll
        for (int kb = 0; kb < PER; ++kb) w2r[0][kb] = s_w2[(w * PER + kb) * 64 + lane];
        resident_tile<1, PER, RC, 1024>(w2r, m.b2 + ft2 * 16, 0, rs_a1, boff, my_tiles - i0 < RC ? my_tiles - i0 : RC, spart,
                                        epilogue_operands);
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
          if (!ework[k]) continue;
          const int r = (t >> 8) + 2 * k;
          float v = splitk_combine<RC, 1>(spart, r, 0, t & 255);
          if (re[k].src >= 0) v = uis_mean_update(old[k], v, re[k].nprev);
          if (f >= m.D) v = 0.0f;
          rs_buf_store_f32(rs_mean, (uint32_t)(((re[k].utt * S + re[k].dst) * DP + f) * 4), v);
        }
        __syncthreads();
      }
    }
    RSTAMP(6);
    if (xcd_barrier(st, cluster, 32u * ++bar, s_ctl) || s_ctl[9]) { left_aborted(); return; }
    RSTAMP(7);
  }
  if (!PERSIST) {
    if (st.avail && keep_beam && did_select) {
      // streaming: the session outlives this launch -- the beam tables (the set the NEXT step reads)
      // and the per-slot frame counts go back to their global copies; beam_n / beam_score / the
      // back-pointers were written step by step
      long done = (long)st.avail[cluster + ncl * rank] - my_step0;
      done = done < 0 ? 0 : (done > nsteps ? nsteps : done);
      if (done > 0) write_back(my_step0 + (int)done);
    } else if (!st.avail && keep_beam && did_select && range1 < total_steps) {
      write_back(range1);  // (a decode in several launches: the next one fetches the tables at its first step)
    }
    break;
  }
  // ---- PERSIST: this push is done.  The owned utterance's step count goes to its global word
  // (the next command's listing reads it), the cluster reports to the host.
  // (the last step's closing barrier has drained everybody's stores, the step count's included)
  gstep += (uint32_t)nsteps;
  if (nsteps == 0 && xcd_barrier(st, cluster, 32u * ++bar, s_ctl)) { left_aborted(); return; }  // (nothing to run: my_cur is the old count)
  PMSTAMP(3);  // the steps
#if defined(UIS_PM_TIMING)
  if (blockIdx.x == 0 && t == 0) {
    pm_acc[5] += 1;
    for (int k = 0; k < 6; ++k) reinterpret_cast<volatile unsigned long long*>(pm.ctl + UIS_PM_TIMING_WORD)[k] = pm_acc[k];
  }
#endif
  // (a push leaves nothing in host memory but this word: no system-scope fence, which would write
  // back the whole L2)
  // (... and nothing at all when a placement / barrier check has failed: the host finds the launch gone)
  if (rank == 0 && t == 0 && !__hip_atomic_load(st.cl_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    __hip_atomic_store(pm.ctl + UIS_PM_DONE_WORD + 16 * cluster, (uint32_t)s_ctl[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }  // command loop
  if (PERSIST) {
    // leaving (told to, or idle): the session goes on with ordinary launches from the global tables
    if (did_select && my_cur > launch_step0) write_back(my_cur);
    if (xcd_barrier(st, cluster, 32u * ++bar, s_ctl)) { left_aborted(); return; }
    if (rank == 0 && t == 0) {
      __threadfence_system();
      __hip_atomic_store(pm.ctl + UIS_PM_LEFT_WORD + 16 * cluster, ctype == UIS_PM_QUIT ? 1u : 2u, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
#if defined(UIS_RESIDENT_PROBE)
  if (t == 0 && blockIdx.x == 0)
    for (int k = 0; k < 8; ++k) st.counters[80 + k] = reinterpret_cast<unsigned long long*>(smem_raw + L.off_misc + 64)[k];
#endif
#if defined(UIS_RESIDENT_TIMING)
  if (t == 0 && (blockIdx.x == 0 || blockIdx.x == 248))
  {
    for (int k = 0; k < 8; ++k) st.counters[(blockIdx.x == 0 ? 48 : 64) + k] = rt_acc[k];
    if (blockIdx.x == 248) for (int k = 0; k < 4; ++k) st.counters[72 + k] = ft_acc[k];
    if (blockIdx.x == 0) for (int k = 0; k < 8; ++k) st.counters[80 + k] = reinterpret_cast<unsigned long long*>(smem_raw + L.off_misc + 64)[k];
  }
#endif
}

// The schedule of k_decode_big's stages with plain loads from this lane's row (launch-per-step
// kernels: the rows were written by earlier launches): NA weight streams in LDS against one row
// tile.  The first operand group arrives preloaded in `bfirst`; the groups alternate between bfirst
// and a second register set, the next one requested while the current one is multiplied, and
// during the LAST group the first group of the wave's NEXT row tile (next_row, if has_next) goes
// into bfirst: a tile's dependent start-up (descriptor -> address -> rows) hides behind its
// predecessor's chain.
template <int GB>
__device__ __forceinline__ void rows_first_group_plain(const float* __restrict__ row, f32x4 (&bfirst)[GB]) {
  const f32x4* bp = reinterpret_cast<const f32x4*>(row) + ((threadIdx.x & 63) >> 4);
#pragma unroll
  for (int k = 0; k < GB; ++k) bfirst[k] = bp[(size_t)k * 4];
}
template <int NA, int NKB, int GS>
__device__ __forceinline__ void fullk_rows_plain(const f32x4* wbase, int wstride, const float* const (&bias)[NA],
                                                 const float* __restrict__ row, f32x4 (&total)[NA],
                                                 f32x4 (&bfirst)[GS * (NKB / UIS_KSPLIT)], const float* __restrict__ next_row,
                                                 bool has_next) {
  constexpr int PER = NKB / UIS_KSPLIT, NGRP = UIS_KSPLIT / GS, GB = GS * PER;
  static_assert(PER * UIS_KSPLIT == NKB && NGRP * GS == UIS_KSPLIT && NGRP % 2 == 0,
                "k-blocks divide into segments, segments into an even number of groups");
  const int lane = threadIdx.x & 63, q = lane >> 4;
  const f32x4* bp = reinterpret_cast<const f32x4*>(row) + q;
  const f32x4* bn = reinterpret_cast<const f32x4*>(next_row) + q;
  f32x4 bsec[GB];
#pragma unroll
  for (int grp = 0; grp < NGRP; ++grp) {
    if (grp + 1 < NGRP) {
      if ((grp + 1) & 1) {
#pragma unroll
        for (int k = 0; k < GB; ++k) bsec[k] = bp[(size_t)((grp + 1) * GB + k) * 4];
      } else {
#pragma unroll
        for (int k = 0; k < GB; ++k) bfirst[k] = bp[(size_t)((grp + 1) * GB + k) * 4];
      }
    }
    const f32x4 (&b)[GB] = (grp & 1) ? bsec : bfirst;
    if (grp + 1 == NGRP && has_next) {  // (the last group reads bsec) bfirst is free: the next tile's first group
#pragma unroll
      for (int k = 0; k < GB; ++k) bfirst[k] = bn[(size_t)k * 4];
    }
#pragma unroll
    for (int sg = 0; sg < GS; ++sg) {
      const int sgm = grp * GS + sg;
      f32x4 acc[NA];
#pragma unroll
      for (int a = 0; a < NA; ++a)
        acc[a] = sgm == 0 ? *reinterpret_cast<const f32x4*>(bias[a] + 4 * q) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int kb = 0; kb < PER; ++kb) {
        f32x4 wa[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) wa[a] = wbase[(size_t)a * wstride + (size_t)(sgm * PER + kb) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int a = 0; a < NA; ++a)
            acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[a][e], b[sg * PER + kb][e], acc[a], 0, 0, 0);
      }
#pragma unroll
      for (int a = 0; a < NA; ++a) {
        if (sgm == 0) total[a] = acc[a];
        else {
#pragma unroll
          for (int i = 0; i < 4; ++i) total[a][i] = total[a][i] + acc[a][i];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// ---------------------------------------------- launch-per-step dense kernels, weights in LDS
//
// Thousands of rnn rows per step (wide beams under look_ahead): the schedule of k_decode_big's
// stages as ordinary kernels.  One 512-thread workgroup per CU-sized share of the work: workgroup
// (feature tile ft, row group g) copies its weight slice into LDS once per launch (96 KB for the
// GRU at hidden size 512) and its eight waves walk the row tiles g, g + NG, ... of the step, a
// whole tile per wave.  The big-tile kernels above stream every weight fragment from L2 for every
// four row tiles and wait ~1 us per four k-blocks for it (0.38 of the MFMA peak); here the A
// operands come from LDS and only the rows are streamed, one operand group ahead.
__host__ __device__ inline int wt_groups(int n_cu, int nft) {  // row groups: about one workgroup per CU, a multiple of 8 (XCDs)
  int ng = n_cu / nft;
  ng = ng < 8 ? 8 : ng - ng % 8;
  return ng;
}
// UP: the input-side gates of GRU layer `layer` >= 1 instead -- gi_up[row][G] = b_ih + W_ih h'_{layer-1} (what
// k_dense_upper_in computes with split-K tiles: 204 us per step at 1024 utterances against 80 for a GRU
// layer of the same size here) -- same streams, no gate arithmetic.
template <int NKB, bool UP = false>
__global__ __launch_bounds__(512) void k_wt_gru(DevModel m, DecodeState st, int par, int layer, int ng) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  f32x4* s_w = reinterpret_cast<f32x4*>(smem_raw);  // [3][NKB][64]
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, q = lane >> 4;
  const int grp = blockIdx.x % ng, ft = blockIdx.x / ng;  // (a row group's workgroups share an XCD's L2: block b -> XCD b % 8)
  const int nft = m.Hp / 16;
  const int nrows = st.nrows[par];
  const int nrt = (nrows + 15) >> 4;
  if (grp >= nrt) return;  // nothing in this group
  for (int e = t; e < NKB * 64; e += 512) {
#pragma unroll
    for (int g = 0; g < 3; ++g) s_w[g * NKB * 64 + e] = reinterpret_cast<const f32x4*>(UP ? m.wih[layer] : m.whh[layer])[(size_t)(g * nft + ft) * NKB * 64 + e];
  }
  __syncthreads();
  const float* bvec = UP ? m.bih[layer] : m.bhh[layer];
  const float* bias[3] = {bvec + ft * 16, bvec + m.Hp + ft * 16, bvec + 2 * m.Hp + ft * 16};
  constexpr int GSG = 2, GBG = GSG * (NKB / UIS_KSPLIT);
  auto fetch = [&](int tl, RnnRow& r_, const float*& hs_) {
    const int row = 16 * tl + (lane & 15);
    r_ = st.rows[row < nrows ? row : 16 * tl];  // (a tile's first row always exists)
    if (UP) hs_ = hid_ptr(m, st, r_, r_.dst, layer - 1);  // (this step's output of the layer below)
    else hs_ = r_.src >= 0 ? hid_ptr(m, st, r_, r_.src, layer) : m.h1 + (size_t)layer * m.Hp;
  };
  int tile = grp + ng * w;
  RnnRow me{};
  const float* hs = m.h1;
  f32x4 bfirst[GBG];
  if (tile < nrt) { fetch(tile, me, hs); rows_first_group_plain<GBG>(hs, bfirst); }
  while (tile < nrt) {
    const int next = tile + ng * 8;
    const bool has_next = next < nrt;
    RnnRow me_n{};
    const float* hs_n = m.h1;
    if (has_next) fetch(next, me_n, hs_n);  // requested now, needed when this tile's chain is almost done
    const int row = 16 * tile + (lane & 15);
    const bool valid = row < nrows;
    const int j4 = ft * 16 + 4 * q;
    f32x4 gir = {0.0f, 0.0f, 0.0f, 0.0f}, giz = gir, gin = gir, hprev = gir;
    if (!UP) {
      const float* gi = layer == 0 ? st.gi0 + (size_t)me.frame * m.G : st.gi_up + (size_t)(valid ? row : 16 * tile) * m.G;
      gir = *reinterpret_cast<const f32x4*>(gi + j4);
      giz = *reinterpret_cast<const f32x4*>(gi + m.Hp + j4);
      gin = *reinterpret_cast<const f32x4*>(gi + 2 * m.Hp + j4);
      hprev = *reinterpret_cast<const f32x4*>(hs + j4);
    }
    f32x4 gh[3];
    fullk_rows_plain<3, NKB, GSG>(s_w, NKB * 64, bias, hs, gh, bfirst, hs_n, has_next);
    if (valid && UP) {
      float* go = st.gi_up + (size_t)row * m.G + j4;
      *reinterpret_cast<f32x4*>(go) = gh[0];
      *reinterpret_cast<f32x4*>(go + m.Hp) = gh[1];
      *reinterpret_cast<f32x4*>(go + 2 * m.Hp) = gh[2];
    } else if (valid) {
      f32x4 out;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        out[i] = j4 + i < m.H ? uis_gru_unit(gir[i], giz[i], gin[i], gh[0][i], gh[1][i], gh[2][i], hprev[i]) : 0.0f;
      *reinterpret_cast<f32x4*>(const_cast<float*>(hid_ptr(m, st, me, me.dst, layer)) + j4) = out;
    }
    tile = next; me = me_n; hs = hs_n;
  }
}
// HEAD 1: a1[row] = relu(b1 + W1 h'_top); HEAD 2: mean = b2 + W2 a1, running-mean update -> dst slot.
// NA feature tiles per workgroup (their weight slices side by side in LDS, 32 KB each at hidden size
// 512): with one, a wave's 16 rows feed 4 MFMAs per k-block and eight waves ask L2 for the CU's full
// 64 bytes per clock (the heads ran at 0.36-0.38 of the MFMA peak next to the three-gate GRU's 0.60);
// every further tile divides that stream.
template <int NKB, int HEAD, int NA>
__global__ __launch_bounds__(512) void k_wt_head(DevModel m, DecodeState st, int par, int ng) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  f32x4* s_w = reinterpret_cast<f32x4*>(smem_raw);  // [NA][NKB][64]
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, q = lane >> 4;
  const int grp = blockIdx.x % ng, ft0 = (blockIdx.x / ng) * NA;
  const int nrows = st.nrows[par];
  const int nrt = (nrows + 15) >> 4;
  if (grp >= nrt) return;
  const f32x4* wg = reinterpret_cast<const f32x4*>(HEAD == 1 ? m.w1 : m.w2) + (size_t)ft0 * NKB * 64;
  for (int e = t; e < NA * NKB * 64; e += 512) s_w[e] = wg[e];  // (consecutive feature tiles are consecutive in memory)
  __syncthreads();
  const float* bias[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) bias[a] = (HEAD == 1 ? m.b1 : m.b2) + (ft0 + a) * 16;
  constexpr int GSH = 2, GBH = GSH * (NKB / UIS_KSPLIT);
  auto fetch = [&](int tl, RnnRow& r_, const float*& in_) {
    const int row = 16 * tl + (lane & 15);
    const int use = row < nrows ? row : 16 * tl;
    r_ = st.rows[use];
    in_ = HEAD == 1 ? hid_ptr(m, st, r_, r_.dst, m.depth - 1) : st.a1 + (size_t)use * m.Hp;
  };
  int tile = grp + ng * w;
  RnnRow me{};
  const float* in = st.a1;
  f32x4 bfirst[GBH];
  if (tile < nrt) { fetch(tile, me, in); rows_first_group_plain<GBH>(in, bfirst); }
  while (tile < nrt) {
    const int next = tile + ng * 8;
    const bool has_next = next < nrt;
    RnnRow me_n{};
    const float* in_n = st.a1;
    if (has_next) fetch(next, me_n, in_n);
    const int row = 16 * tile + (lane & 15);
    const bool valid = row < nrows;
    f32x4 old[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      old[a] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      if (HEAD == 2 && valid && me.src >= 0)
        old[a] = *reinterpret_cast<const f32x4*>(st.pool_mean + ((size_t)me.utt * st.S + me.src) * m.Dp + (ft0 + a) * 16 + 4 * q);
    }
    f32x4 v[NA];
    fullk_rows_plain<NA, NKB, GSH>(s_w, NKB * 64, bias, in, v, bfirst, in_n, has_next);
    if (valid) {
#pragma unroll
      for (int a = 0; a < NA; ++a) {
        const int f4 = (ft0 + a) * 16 + 4 * q;
        if (HEAD == 1) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[a][i] = v[a][i] > 0.0f ? v[a][i] : 0.0f;
          *reinterpret_cast<f32x4*>(st.a1 + (size_t)row * m.Hp + f4) = v[a];
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (me.src >= 0) v[a][i] = uis_mean_update(old[a][i], v[a][i], me.nprev);
            if (f4 + i >= m.D) v[a][i] = 0.0f;
          }
          *reinterpret_cast<f32x4*>(st.pool_mean + ((size_t)me.utt * st.S + me.dst) * m.Dp + f4) = v[a];
        }
      }
    }
    tile = next; me = me_n; in = in_n;
  }
}

// ------------------------------------------------------------ one-launch decode of SMALL models
//
// Round 4.  The one-launch kernels above keep a 32nd of a hidden-size-256 / 512 model per workgroup and
// take depth 1 only; everything else -- every model of the reference's own tests among them: hidden size
// 8, depth 1 and 2 (tests/uisrnn_test.py:26-70, tests/integration_test.py:56-134) -- went through four
// or more launches per decode step (~40 us per step whatever the model's size).  A small model needs no
// co-operation between workgroups at all: ONE workgroup per utterance runs the whole beam search --
// the select with the beam tables in LDS (select_fast_body<KEEP>, as a workgroup of the resident decode
// that owns one utterance), then the step's CoreRNN rows (uisrnn.py:45-52) tile by tile through
// splitk_tile, the very function (and arithmetic) of the launch-per-step kernels: for every GRU layer
// the input-side gates of layers >= 1, the hidden-side gates and the gate arithmetic, then linear_mean1,
// then linear_mean2 and the running mean.  Weights come from L2 / L1 (a few KB); what one stage writes
// the next reads after a workgroup barrier (same workgroup: no cluster protocol).  Any rnn_depth,
// look_ahead 1, models up to small_model_ok's size; a plain launch of one workgroup per utterance (no
// co-residency needed).  Bit-identical to the launch-per-step path
// (tests/test_gpu_parity.py::_compare runs both).
__host__ __device__ inline size_t small_lds_bytes(int Dp, int B, int Kmax, int S) {
  return (size_t)((fast_lds_layout(Dp, B, Kmax, S).total + 255) & ~255) + (size_t)UIS_KSPLIT * 3 * 256 * 4 + 64;
}
// What "small" means: the workgroup walks the step's dense work one (row tile, feature tile) pass at a
// time, ~1.1 us each (measured: hidden 8 / depth 1: 3 passes, 11 us per step against 20-25 us on the
// launch-per-step path; hidden 8 / depth 2: 7 passes, 12.6 against 27-32; hidden 24 / depth 3: 22
// passes, 25 against 35-41; hidden 100 / depth 2: 49 passes, 65 against 32-42 -- too many), so the
// kernel takes a model when its passes per row tile number at most 24.
__host__ __device__ inline bool small_model_ok(int Hp, int Dp, int depth) {
  const int nft = Hp / 16;
  return nft + (depth - 1) * 4 * nft + nft + Dp / 16 <= 24;
}

// ------------------------------------------------------------------ window
//
// look_ahead >= 2 (uisrnn.py:469-477,529-559): a window of Lw <= L frames is scored jointly.
// Sub-step j < Lw-1 EXPANDS every hypothesis of the current level into all its finite
// children (one per cluster assignment, in lexicographic order) and schedules the CoreRNN
// rows that produce the children's cluster states; the last sub-step PRUNES to the beam
// exactly like k_select.  Children of one parent share everything but one cluster state, and
// children of different parents that extend the same cluster state share the new one too
// (one rnn row per distinct source state), so a window costs far fewer CoreRNN evaluations
// than the reference's enumeration.
//
// One workgroup per utterance; work arrays live in a per-utterance global scratch region
// (levels can hold thousands of hypotheses), ordering steps are block scans.

struct HypView {
  int cap;
  int32_t* n; int32_t* K; int32_t* last; int32_t* sum; float* score;
  int32_t* slot; int32_t* blk; int32_t* origin; int16_t* path;
};

__device__ __forceinline__ HypView beam_view(const DecodeState& st, int u, int par) {
  HypView v;
  const size_t e = ((size_t)par * st.U + u) * st.B;
  v.cap = st.B;
  v.n = st.beam_n + (size_t)par * st.U + u;
  v.K = st.beam_K + e; v.last = st.beam_last + e; v.sum = st.beam_sum + e; v.score = st.beam_score + e;
  v.slot = st.beam_slot + e * st.Kmax; v.blk = st.beam_blk + e * st.Kmax;
  v.origin = nullptr; v.path = nullptr;
  return v;
}
__device__ __forceinline__ HypView level_view(const DecodeState& st, int u, int buf) {
  HypView v;
  const size_t e = ((size_t)buf * st.U + u) * st.NC;
  v.cap = st.NC;
  v.n = st.lv_n + (size_t)buf * st.U + u;
  v.K = st.lv_K + e; v.last = st.lv_last + e; v.sum = st.lv_sum + e; v.score = st.lv_score + e;
  v.slot = st.lv_slot + e * st.Kmax; v.blk = st.lv_blk + e * st.Kmax;
  v.origin = st.lv_origin + e; v.path = st.lv_path + e * st.L;
  return v;
}

struct WindowScratch {
  // hot block (LDS when it fits, see window_lds_bytes): flags, per-slot values, candidate keys
  size_t live, livelist, mse, cnt, first, cbase, key, hot_total;
  // cold block (always global scratch)
  size_t cscore, win, src, dst, lead, ord, freelist, total;
};
__host__ __device__ inline WindowScratch window_scratch_layout(int S, int NC, int Kmax, int B) {
  WindowScratch w;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 15) & ~(size_t)15; return r; };
  const size_t in_cap = (size_t)(NC > B ? NC : B);
  const size_t C = in_cap * (size_t)(Kmax + 1);
  const size_t out_cap = in_cap;
  w.live = take((size_t)S * 4);
  w.livelist = take((size_t)S * 4);
  w.mse = take((size_t)S * 4);
  w.cnt = take((size_t)S * 4);
  w.first = take((size_t)(S + 1) * 4);
  w.cbase = take((in_cap + 1) * 4);
  w.key = take((C + 2) * 8);
  w.hot_total = o;
  w.cscore = take(C * 4);
  w.win = take(out_cap * 4);
  w.src = take(out_cap * 4);
  w.dst = take(out_cap * 4);
  w.lead = take(out_cap * 4);
  w.ord = take(out_cap * 4);
  w.freelist = take(out_cap * 4);
  w.total = o;
  return w;
}
// dynamic LDS to launch k_window with: everything if it fits next to the static LDS (round 4: the
// survivors' work arrays too -- every phase that went through them paid a global round trip), else
// the hot block, else 0
__host__ __device__ inline size_t window_lds_bytes(const WindowScratch& w) {
  return w.total <= 150 * 1024 ? w.total : (w.hot_total <= 150 * 1024 ? w.hot_total : 0);
}

// Exclusive prefix sums over i in [0, n) of val(i) by NT threads (contiguous chunk each);
// emit(i, prefix) is called for every i; returns the total.  lds4: NT / 64 ints of LDS.
template <int NT, typename V, typename E>
__device__ __forceinline__ int block_scan(int n, V val, E emit, int* lds4) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int chunk = (n + NT - 1) / NT;
  const int lo = tid * chunk < n ? tid * chunk : n;
  const int hi = lo + chunk < n ? lo + chunk : n;
  int cnt = 0;
  for (int i = lo; i < hi; ++i) cnt += val(i);
  int incl = cnt;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  if (lane == 63) lds4[wave] = incl;
  __syncthreads();
  int base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) { const int c = lds4[w]; if (w < wave) base += c; total += c; }
  int run = base + incl - cnt;
  for (int i = lo; i < hi; ++i) { emit(i, run); run += val(i); }
  __syncthreads();
  return total;
}

#define UIS_WINDOW_LOGTAB 128
// Diagnostic build (-DUIS_SELECT_TIMING): thread 0 of every workgroup adds the wall-clock ticks (10 ns)
// of each phase to counters[16 + phase] (expanding sub-steps) / counters[32 + phase] (the pruning one)
#if defined(UIS_SELECT_TIMING)
#define WSTAMP(k) do { if (threadIdx.x == 0) { const unsigned long long t_now_ = wall_clock64(); \
    atomicAdd(&st.counters[(last ? 32 : 16) + (k)], t_now_ - w_prev_); w_prev_ = t_now_; } } while (0)
#else
#define WSTAMP(k) do {} while (0)
#endif

// The body: one sub-step of utterance u's current window by the NT threads of the calling workgroup
// (every thread calls it; returns are workgroup-uniform; every phase is a scan or a count over the
// level's candidates).  `wlds` = the dynamic LDS the work arrays may use (window_lds_bytes: the
// launch's choice), `sink` = where the sub-step's rnn rows go (the step's list and its counter).
// INL: called from inside a one-launch decode (k_decode_big<WIN>) -- the cluster means were written
// by other workgroups of this XCD inside the same launch, so they are read with sc1 loads (past this
// CU's L1); everything else the body reads is immutable or was written by this very workgroup.
// phase: 0 the whole sub-step; 1 only the part that does not read the cluster means (live list,
// candidate offsets: k_decode_big<WIN> runs it while it waits for the previous sub-step's last
// barrier), 2 the rest (the LDS between the two calls untouched).
template <int NT, bool INL = false>
__device__ __forceinline__ void window_body(const DevModel& m, const DecodeState& st, int u, unsigned char* wlds, RowSink sink,
                                            int phase = 0) {
  constexpr int NW = NT / 64;
  __shared__ int lds4[2 * NW];  // (NW for the scans, 2 NW for the two-word reductions of the prune)
  __shared__ int lds_misc[8];  // [0] nlive [1] nfinite [2..4] the prune's digit search
  __shared__ int radix_hist[256];
  __shared__ double s_logblk[UIS_WINDOW_LOGTAB];  // log(block count) for the small counts (larger ones: the global table)
  const int tid = threadIdx.x;
  const int B = st.B, Kmax = st.Kmax, S = st.S, L = st.L, NC = st.NC;
  const int step = st.utt_step[u];
  const long off0 = (long)st.off[u], off1 = (long)st.off[u + 1];
  const long N = off1 - off0;
  const long T = (long)st.tau * N;
  if (step >= T) return;
  // an intermediate level overflowed earlier: the utterance's result is void (the host reports
  // UIS_ERR_UNSUPPORTED); do not grind through full levels for it
  if (st.overflow[u] & 2) return;
  const long frame = off0 + (step % N);
  const long win = step / L;
  const long t0 = win * L;
  const int Lw = (int)((T - t0) < L ? (T - t0) : L);   // ragged last window, uisrnn.py:532-533
  const int j = (int)(step - t0);
  const bool last = j == Lw - 1;
  const int wpar = (int)(win & 1);
  const HypView in = j == 0 ? beam_view(st, u, wpar) : level_view(st, u, (j - 1) & 1);
  const HypView out = last ? beam_view(st, u, wpar ^ 1) : level_view(st, u, j & 1);
  const int n_in = *in.n;
#if defined(UIS_SELECT_TIMING)
  unsigned long long w_prev_ = wall_clock64();
#endif

  unsigned char* scr = st.scratch + (size_t)u * st.scratch_stride;
  const WindowScratch W = window_scratch_layout(S, NC, Kmax, B);
  unsigned char* hot = window_lds_bytes(W) ? wlds : scr;  // same choice as the host's launch
  unsigned char* warm = window_lds_bytes(W) == W.total ? wlds : scr;
  int* live = reinterpret_cast<int*>(hot + W.live);
  int* livelist = reinterpret_cast<int*>(hot + W.livelist);
  float* mse = reinterpret_cast<float*>(hot + W.mse);
  int* cntv = reinterpret_cast<int*>(hot + W.cnt);
  int* first = reinterpret_cast<int*>(hot + W.first);
  int* cbase = reinterpret_cast<int*>(hot + W.cbase);
  unsigned long long* key = reinterpret_cast<unsigned long long*>(hot + W.key);
  float* cscore = reinterpret_cast<float*>(warm + W.cscore);
  int* winv = reinterpret_cast<int*>(warm + W.win);
  int* srcv = reinterpret_cast<int*>(warm + W.src);
  int* dstv = reinterpret_cast<int*>(warm + W.dst);
  int* leadv = reinterpret_cast<int*>(warm + W.lead);
  int* ordv = reinterpret_cast<int*>(warm + W.ord);
  int* freelist = reinterpret_cast<int*>(warm + W.freelist);

  // ---- live cluster states of the input level, candidate offsets
  int C = 0;
  if (phase != 2) {
    for (int sl = tid; sl <= S; sl += NT) { if (sl < S) live[sl] = 0; first[sl] = 0x7fffffff; }
    if (tid < UIS_WINDOW_LOGTAB) s_logblk[tid] = st.logblk[tid];  // (the host fills at least UIS_RS_LOGTAB entries)
    if (tid < 8) lds_misc[tid] = 0;
    __syncthreads();
    for (int e = tid; e < n_in * Kmax; e += NT) {
      const int i = e / Kmax, c = e - i * Kmax;
      if (c < in.K[i]) live[in.slot[(size_t)i * Kmax + c]] = 1;
    }
    C = block_scan<NT>(n_in, [&](int i) { return in.K[i] + 1; }, [&](int i, int pre) { cbase[i] = pre; }, lds4);
    if (tid == 0) cbase[n_in] = C;
    for (int s0 = 0; s0 < S; s0 += NT) {  // (one reservation per wave, not per live slot: the list's order is free)
      const int sl = s0 + tid;
      const bool on = sl < S && live[sl] != 0;
      const unsigned long long m = __ballot(on);
      int base = 0;
      if ((tid & 63) == 0 && m) base = atomicAdd(&lds_misc[0], __popcll(m));
      base = __shfl(base, 0, 64);
      if (on) livelist[base + wave_below(m)] = sl;
    }
    __syncthreads();
    if (phase == 1) return;
  } else {
    C = cbase[n_in];
  }
  const int nlive = lds_misc[0];
  WSTAMP(0);

  // ---- weighted MSE of the frame against every live cluster state (as k_select, phase A)
  const float* pmean = st.pool_mean + (size_t)u * S * m.Dp;
  const float* xrow = st.x + (size_t)frame * m.Dp;
  const __amdgpu_buffer_rsrc_t rs_pmean = __builtin_amdgcn_make_buffer_rsrc((void*)pmean, (short)0, 0x7fffffff, 0x00020000);
  auto load_mean4 = [&](int slot, int d) -> f32x4 {  // four consecutive floats of the mean in `slot`
    if (INL) return load_sc1(rs_pmean, (uint32_t)(((size_t)slot * m.Dp + d) * 4));
    return *reinterpret_cast<const f32x4*>(pmean + (size_t)slot * m.Dp + d);
  };
  {
    const int grp = tid >> 4, p = tid & 15;
    if (m.Dp <= 256) {
      // one 256-float chunk: the frame and the weights stay in registers and FOUR slots per 16-lane
      // group are in flight (a level of a wide beam has hundreds of live states: the loop was one
      // dependent L2 round trip per 16 slots, the largest phase of the kernel)
      f32x4 xv[4], wv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int d = 4 * (p + 16 * k);
        const bool in_ = d < m.Dp;
        xv[k] = in_ ? *reinterpret_cast<const f32x4*>(xrow + d) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        wv[k] = in_ ? *reinterpret_cast<const f32x4*>(m.wgt + d) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      }
      for (int i0 = 0; i0 < nlive; i0 += 4 * (NT / 16)) {
        int sl[4]; bool act[4]; f32x4 mv[4][4];
#pragma unroll
        for (int h2 = 0; h2 < 4; ++h2) {
          const int i = i0 + (NT / 16) * h2 + grp;
          act[h2] = i < nlive;
          sl[h2] = livelist[act[h2] ? i : 0];
        }
#pragma unroll
        for (int h2 = 0; h2 < 4; ++h2) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int d = 4 * (p + 16 * k);
            // (no branch around the load: sixteen of them are in flight per group; a chunk past Dp reads
            // the slot's first floats instead and is zeroed)
            const f32x4 got = load_mean4(sl[h2], d < m.Dp ? d : 0);
            mv[h2][k] = d < m.Dp ? got : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
          }
        }
#pragma unroll
        for (int h2 = 0; h2 < 4; ++h2) {
          float A[4] = {0.0f, 0.0f, 0.0f, 0.0f};
          mse16_block(mv[h2], xv, wv, A);  // (chunks past Dp are zero-filled)
          const float d0 = mv[h2][0][0] - xv[0][0];
          const float t = mse16_total(A);
          if (p == 0 && act[h2]) { mse[sl[h2]] = uis_mse_finish(t, d0 * d0, m.D); cntv[sl[h2]] = st.pool_cnt[(size_t)u * S + sl[h2]]; }
        }
      }
    } else {
      for (int i0 = 0; i0 < nlive; i0 += NT / 16) {
        const int i = i0 + grp;
        const bool act = i < nlive;
        const int sl = livelist[act ? i : 0];
        float A[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        float first_sq = 0.0f;
        for (int q = 0; q < m.Dp; q += 256) {
          f32x4 mv[4], xv[4], wv[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int d = q + 4 * (p + 16 * k);
            const bool in_ = d < m.Dp;
            const f32x4 got = load_mean4(sl, in_ ? d : 0);
            mv[k] = in_ ? got : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            xv[k] = in_ ? *reinterpret_cast<const f32x4*>(xrow + d) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            wv[k] = in_ ? *reinterpret_cast<const f32x4*>(m.wgt + d) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
          }
          mse16_block(mv, xv, wv, A);
          if (q == 0) { const float d0 = mv[0][0] - xv[0][0]; first_sq = d0 * d0; }
        }
        const float t = mse16_total(A);
        if (p == 0 && act) { mse[sl] = uis_mse_finish(t, first_sq, m.D); cntv[sl] = st.pool_cnt[(size_t)u * S + sl]; }
      }
    }
  }
  __syncthreads();
  WSTAMP(1);

  // ---- candidate scores
  const float mse_new = st.mse0[frame];
  auto node_of = [&](int i) {  // largest node with cbase[node] <= i
    int lo = 0, hi = n_in;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (cbase[mid] <= i) lo = mid; else hi = mid; }
    return lo;
  };
  // a thread per input hypothesis (its table row fetched once), its K_b + 1 candidates in turn --
  // not a thread per candidate, each finding its hypothesis by bisection and re-reading the row
  for (int b = tid; b < n_in; b += NT) {
    const int Kb = in.K[b], lastb = in.last[b], i0 = cbase[b];
    const float base_score = in.score[b];
    const double ld = st.logden[in.sum[b]];
    const int origin = (st.dbg_scores && last) ? (in.origin ? in.origin[b] : b) : 0;
    auto emit = [&](int c, float ms, double prior) {
      const float sc = base_score + uis_step_loss(ms, prior);
      const int i = i0 + c;
      cscore[i] = sc;
      const bool fin = uis_isfinite(sc);
      key[i] = fin ? (((unsigned long long)uis_score_key(sc) << 32) | (unsigned)i) : ~0ull;
      if (st.dbg_scores && last) {
        // UIS_FLAG_DEBUG_SCORES: the window's _calculate_score array (uisrnn.py:455-477) -- the score
        // of the whole assignment tuple (c_1 .. c_Lw) of beam hypothesis `origin`, at
        // [window][utterance][origin][c_1] .. [c_L] (a ragged last window: index 0 in the missing
        // dimensions, as numpy lays a lower-dimensional array into predict_single's score_set)
        size_t idx = ((size_t)win * st.U + u) * B + origin;
        for (int k = 0; k < L; ++k)
          idx = idx * (size_t)(Kmax + 1) + (size_t)(k < j ? (int)in.path[(size_t)b * L + k] : (k == j ? c : 0));
        st.dbg_scores[idx] = sc;
      }
    };
    // the row's clusters four at a time: slots and block counts requested together, then their logs
    for (int c0 = 0; c0 < Kb; c0 += 4) {
      int sl[4], bk[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = c0 + e < Kb ? c0 + e : c0;
        sl[e] = in.slot[(size_t)b * Kmax + c];
        bk[e] = in.blk[(size_t)b * Kmax + c];
      }
      double lb[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) lb[e] = bk[e] < UIS_WINDOW_LOGTAB ? s_logblk[bk[e]] : st.logblk[bk[e]];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = c0 + e;
        if (c < Kb) emit(c, mse[sl[e]], (c == lastb) ? m.lp_stay : (m.lp_sw + lb[e]) - ld);
      }
    }
    emit(Kb, mse_new, m.lp_new - ld);
  }
  __syncthreads();
  WSTAMP(2);

  // ---- which candidates go on: all finite ones in order (expand) or the B best (prune)
  int keep;
  if (!last) {
    const int nfin = block_scan<NT>(C, [&](int i) { return key[i] != ~0ull ? 1 : 0; },
                                [&](int i, int pre) { if (key[i] != ~0ull && pre < NC) winv[pre] = i; }, lds4);
    keep = nfin;
    if (keep > NC) { keep = NC; if (tid == 0) atomicOr(&st.overflow[u], 2); }  // level capacity (not the cluster cap): bit 1
  } else {
    const int nfin = block_scan<NT>(C, [&](int i) { return key[i] != ~0ull ? 1 : 0; }, [&](int, int) {}, lds4);
    keep = nfin < B ? nfin : B;
    if (keep > 0) {
      // The `keep` smallest keys, in order.  Keys are unique (score bits, candidate index), so
      // the keep-th smallest one, K*, is the largest T with #{key < T} <= keep - 1: found bit by
      // bit from the top, one workgroup-wide count per bit (64 counts over C keys instead of
      // the C^2 / 2 comparisons of ranking every candidate against every other).
      // Done in two halves.  The score half (high 32 bits) first, starting below the prefix all
      // finite keys share (scores of one step have the same sign and exponent: ~10 bits decide
      // nothing); if the candidates at the threshold score are exactly as many as are still
      // needed -- the usual case: ties are rare -- the index half needs no search at all.
      // (64 workgroup-wide counts over up to 15 k keys -> about 22.)
      auto count_wg = [&](auto pred) {  // workgroup-wide count of pred(key)
        int c = 0;
        for (int i = tid; i < C; i += NT) c += pred(key[i]) ? 1 : 0;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
        __syncthreads();  // (lds4 of the previous count has been read by everybody)
        if ((tid & 63) == 0) lds4[tid >> 6] = c;
        __syncthreads();
        int tot = 0;
#pragma unroll
        for (int w2 = 0; w2 < NW; ++w2) tot += lds4[w2];
        return tot;
      };
      unsigned hi_and = 0xffffffffu, hi_or = 0u;
      for (int i = tid; i < C; i += NT) {
        const unsigned long long k = key[i];
        if (k != ~0ull) { hi_and &= (unsigned)(k >> 32); hi_or |= (unsigned)(k >> 32); }
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) { hi_and &= __shfl_xor(hi_and, off, 64); hi_or |= __shfl_xor(hi_or, off, 64); }
      __syncthreads();
      if ((tid & 63) == 0) { lds4[tid >> 6] = (int)hi_and; lds4[NW + (tid >> 6)] = (int)hi_or; }
      __syncthreads();
      hi_and = 0xffffffffu; hi_or = 0u;
#pragma unroll
      for (int w2 = 0; w2 < NW; ++w2) { hi_and &= (unsigned)lds4[w2]; hi_or |= (unsigned)lds4[NW + w2]; }
      const unsigned diff = hi_or & ~hi_and;                    // score bits in which the finite keys differ
      const int top = diff ? 31 - __builtin_clz(diff) : -1;      // the highest of them
      unsigned khi = top >= 31 ? 0u : (hi_and & ~((2u << top) - 1u));  // the common prefix above it
      if (top < 0) khi = hi_and;
      // The undecided bits top .. 0 eight at a time (round 4; one workgroup-wide count PER BIT before:
      // ~22 counts of two barriers each were the largest phase of the pruning sub-step): a histogram of
      // the next digit over the keys that still match the prefix, then the digit in which the
      // keep-th smallest key lies.  n_lt = keys below the prefix so far.
      int n_lt = 0, n_eq = nfin;
      for (int bit = top; bit >= 0;) {
        const int nbits = bit + 1 < 8 ? bit + 1 : 8, sh = bit + 1 - nbits, nbin = 1 << nbits;
        for (int i = tid; i < nbin; i += NT) radix_hist[i] = 0;
        __syncthreads();
        for (int i = tid; i < C; i += NT) {
          const unsigned long long k = key[i];
          const unsigned hi = (unsigned)(k >> 32);
          // (bits above `bit` of every finite key that is still in play equal khi's)
          if (k != ~0ull && (bit == 31 || (hi >> (bit + 1)) == (khi >> (bit + 1)))) atomicAdd(&radix_hist[(hi >> sh) & (unsigned)(nbin - 1)], 1);
        }
        __syncthreads();
        if (tid < 64) {  // wave 0: lane l owns bins 4 l .. 4 l + 3
          int c4[4], mine = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e) { c4[e] = 4 * tid + e < nbin ? radix_hist[4 * tid + e] : 0; mine += c4[e]; }
          int incl = mine;
#pragma unroll
          for (int off = 1; off < 64; off <<= 1) {
            const int t2 = __shfl_up(incl, off, 64);
            if (tid >= off) incl += t2;
          }
          const unsigned long long reach = __ballot(n_lt + incl >= keep);   // (the last lane always does: keep <= the keys in play)
          const int l = __ffsll((long long)reach) - 1;
          if (tid == l) {
            int below = incl - mine, d = 0, here = c4[0];
#pragma unroll
            for (int e = 0; e < 3; ++e)
              if (d == e && n_lt + below + c4[e] < keep) { below += c4[e]; d = e + 1; here = c4[e + 1]; }
            lds_misc[2] = 4 * tid + d;   // the digit
            lds_misc[3] = below;         // keys in play with a smaller digit
            lds_misc[4] = here;          // keys in play with this digit
          }
        }
        __syncthreads();
        khi |= (unsigned)lds_misc[2] << sh;
        n_lt += lds_misc[3];
        n_eq = lds_misc[4];
        bit = sh - 1;
      }
      const int need = keep - n_lt;                               // how many of the threshold score go on (>= 1)
      unsigned klo = 0xffffffffu;
      if (n_eq != need) {                                         // a tie at the threshold: lowest candidate indices first
        klo = 0u;
        for (int bit = 31; bit >= 0; --bit) {
          const unsigned t2 = klo | (1u << bit);
          if (count_wg([&](unsigned long long k) { return (unsigned)(k >> 32) == khi && (unsigned)k < t2; }) <= need - 1) klo = t2;
        }
      }
      const unsigned long long kstar = ((unsigned long long)khi << 32) | klo;
      // the winners (key <= K*) in candidate order, then each one's rank among them
      int* wl = ordv;  // (free until the leader bookkeeping below)
      block_scan<NT>(C, [&](int i) { return key[i] <= kstar ? 1 : 0; },
                 [&](int i, int pre) { if (key[i] <= kstar) wl[pre] = i; }, lds4);
      if (keep <= 64) {
        // one wave, a winner per lane, the others' keys by lane broadcast (the loop below it was fifty
        // dependent LDS round trips per winner)
        if (tid < 64) {
          const int mine = tid < keep ? wl[tid] : 0;
          const unsigned long long kl = tid < keep ? key[mine] : ~0ull;
          const unsigned klo32 = (unsigned)kl, khi32 = (unsigned)(kl >> 32);
          int rank = 0;
          for (int b2 = 0; b2 < keep; ++b2) {
            const unsigned long long kb = ((unsigned long long)__builtin_amdgcn_readlane(khi32, b2) << 32) |
                                          __builtin_amdgcn_readlane(klo32, b2);
            rank += kb < kl ? 1 : 0;
          }
          if (tid < keep) winv[rank] = mine;
        }
      } else {
        for (int a = tid; a < keep; a += NT) {
          const unsigned long long ka = key[wl[a]];
          int rank = 0;
          for (int b2 = 0; b2 < keep; ++b2) rank += key[wl[b2]] < ka ? 1 : 0;
          winv[rank] = wl[a];
        }
      }
    }
  }
  __syncthreads();
  WSTAMP(3);

  // ---- source cluster state per survivor; one rnn row per distinct source (index S = fresh cluster)
  const bool nodedup = (st.flags & 1u) != 0;
  // (survivor r = hypothesis b, cluster c of the input level: found once by bisection and kept in the
  // live list's LDS -- free since the MSE pass -- for the table and record loops below)
  int* wbc = livelist;
  for (int r = tid; r < keep; r += NT) {
    const int i = winv[r];
    const int b = node_of(i);
    const int c = i - cbase[b];
    wbc[r] = b | (c << 19);  // (hypothesis of the level: < 2^19 = the largest level capacity; cluster: <= 4096)
    const int src = c < in.K[b] ? in.slot[(size_t)b * Kmax + c] : S;
    srcv[r] = src;
    if (!nodedup) atomicMin(&first[src], r);
  }
  __syncthreads();
  for (int r = tid; r < keep; r += NT) leadv[r] = nodedup ? r : first[srcv[r]];
  __syncthreads();
  const int nlead = block_scan<NT>(keep, [&](int r) { return leadv[r] == r ? 1 : 0; },
                               [&](int r, int pre) { ordv[r] = pre; }, lds4);
  // this utterance's rows of the step's row list: ONE reservation (an atomic per row before; the order
  // of the utterances' blocks in the list is whatever the reservations make it: nothing depends on it)
  if (tid == 0) { lds_misc[5] = nlead > 0 ? atomicAdd(sink.count, nlead) : 0; lds_misc[6] = 0; }
  block_scan<NT>(S, [&](int sl) { return live[sl] ? 0 : 1; },
             [&](int sl, int pre) { if (!live[sl] && pre < nlead) freelist[pre] = sl; }, lds4);
  for (int r = tid; r < keep; r += NT) if (leadv[r] == r) dstv[r] = freelist[ordv[r]];
  __syncthreads();
  for (int r = tid; r < keep; r += NT) if (leadv[r] != r) dstv[r] = dstv[leadv[r]];
  __syncthreads();
  WSTAMP(4);

  // ---- write the next level / the next beam
  for (long e = tid; e < (long)keep * Kmax; e += NT) {
    const int r = (int)(e / Kmax), c2 = (int)(e - (long)r * Kmax);
    const int b = wbc[r] & 0x7ffff, c = (int)((unsigned)wbc[r] >> 19);
    const int Kb = in.K[b];
    const bool is_new = c == Kb;
    const int Knew = Kb + (is_new ? 1 : 0);
    if (c2 < Knew) {
      int slot, blk;
      if (c2 == c) { slot = dstv[r]; blk = is_new ? 1 : in.blk[(size_t)b * Kmax + c] + (c != in.last[b] ? 1 : 0); }
      else { slot = in.slot[(size_t)b * Kmax + c2]; blk = in.blk[(size_t)b * Kmax + c2]; }
      out.slot[(size_t)r * Kmax + c2] = slot;
      out.blk[(size_t)r * Kmax + c2] = blk;
    }
  }
  WSTAMP(5);
  uint16_t* bp = st.bp16 + ((size_t)st.bp_base[u] + (size_t)win * B) * (L + 1);
  const int row_base = lds_misc[5];
  for (int r = tid; r < keep; r += NT) {
    const int i = winv[r];
    const int b = wbc[r] & 0x7ffff, c = (int)((unsigned)wbc[r] >> 19);
    const int Kb = in.K[b];
    const bool is_new = c == Kb;
    int Knew = Kb + (is_new ? 1 : 0);
    if (Knew > Kmax) { Knew = Kmax; atomicOr(&st.overflow[u], 1); }
    out.K[r] = Knew;
    out.last[r] = c;
    out.sum[r] = in.sum[b] + ((is_new || c != in.last[b]) ? 1 : 0);
    out.score[r] = cscore[i];
    const int origin = in.origin ? in.origin[b] : b;
    if (!last) {
      out.origin[r] = origin;
      for (int k = 0; k < j; ++k) out.path[(size_t)r * L + k] = in.path[(size_t)b * L + k];
      out.path[(size_t)r * L + j] = (int16_t)c;
    } else {
      uint16_t* rec = bp + (size_t)r * (L + 1);
      rec[0] = (uint16_t)origin;
      for (int k = 0; k < j; ++k) rec[1 + k] = (uint16_t)in.path[(size_t)b * L + k];
      rec[1 + j] = (uint16_t)c;
      for (int k = j + 1; k < L; ++k) rec[1 + k] = 0xffffu;
      atomicMax(&lds_misc[6], Knew);  // surviving hypotheses only (one global atomic per utterance below)
    }
    if (leadv[r] == r) {
      const int src = srcv[r];
      const int nprev = src < S ? cntv[src] : 0;
      st.pool_cnt[(size_t)u * S + dstv[r]] = nprev + 1;
      const int pos = row_base + ordv[r];
      RnnRow rr; rr.utt = u; rr.src = src < S ? src : -1; rr.dst = dstv[r]; rr.nprev = nprev; rr.frame = frame; rr.pad = 0;
      sink.rows[pos] = rr;
    }
  }
  WSTAMP(6);
  __syncthreads();
  if (tid == 0) {
    if (last && lds_misc[6] > 0) atomicMax(&st.counters[3], (unsigned long long)lds_misc[6]);
#if defined(UIS_SELECT_TIMING)
    atomicAdd(&st.counters[(last ? 32 : 16) + 7], 1ull);                              // launches
    atomicAdd(&st.counters[(last ? 32 : 16) + 8], (unsigned long long)C);             // candidates
    atomicAdd(&st.counters[(last ? 32 : 16) + 9], (unsigned long long)nlive);         // live states
    atomicAdd(&st.counters[(last ? 32 : 16) + 10], (unsigned long long)n_in);         // input hypotheses
    atomicAdd(&st.counters[(last ? 32 : 16) + 11], (unsigned long long)nlead);        // rnn rows
#endif
    *out.n = keep;
    st.utt_step[u] = step + 1;
    atomicAdd(&st.counters[0], (unsigned long long)nlead);
    atomicAdd(&st.counters[1], (unsigned long long)keep);
    atomicAdd(&st.counters[2], (unsigned long long)C);
  }
}

// NT threads per utterance: 256 for narrow beams, more when a level holds hundreds of hypotheses
template <int NT>
__global__ __launch_bounds__(NT) void k_window(DevModel m, DecodeState st, int par) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wlds[];
  if (blockIdx.x == 0 && threadIdx.x == 0) st.nrows[par ^ 1] = 0;
  window_body<NT, false>(m, st, (int)blockIdx.x, wlds, RowSink{st.rows, st.nrows + par});
}

// ---- k_decode_small (described above, next to small_lds_bytes; here because its WIN form runs window_body)
// WIN (round 4): look_ahead >= 2 -- the select is a sub-step of the window kernel (window_body: expand / prune),
// a level holds up to NC hypotheses and emits up to NC rows; its work arrays take the select's place in LDS.
__host__ __device__ inline size_t small_win_lds_bytes(int S, int NC, int Kmax, int B) {
  return (size_t)((window_scratch_layout(S, NC, Kmax, B).total + 255) & ~255) + (size_t)UIS_KSPLIT * 3 * 256 * 4 + 64;
}
template <bool WIN>
__global__ __launch_bounds__(512) void k_decode_small(DevModel m, DecodeState st) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int t = threadIdx.x, u = blockIdx.x;
  const int B = st.B, S = st.S, U = st.U;
  const int rpu = WIN ? st.NC : B;  // rows an utterance can emit per (sub-)step
  const size_t sel_bytes = WIN ? window_scratch_layout(S, st.NC, st.Kmax, B).total : (size_t)fast_lds_layout(m.Dp, B, st.Kmax, S).total;
  float* spart = reinterpret_cast<float*>(smem_raw + ((sel_bytes + 255) & ~(size_t)255));  // [UIS_KSPLIT][3][256]
  const long off0 = (long)st.off[u], off1 = (long)st.off[u + 1];
  const long T = (long)st.tau * (off1 - off0);
  RnnRow* const rows_u = st.rows + (size_t)u * rpu;        // this utterance's rows of a step (at most beam_size / a level)
  float* const gi_up_u = st.gi_up + (size_t)u * rpu * m.G;  // (rnn_depth >= 2)
  float* const a1_u = st.a1 + (size_t)u * rpu * m.Hp;
  int32_t* const cnt = st.utt_nrows + 2 * (size_t)u;        // row counters by step parity
  const int nft = m.Hp / 16, nKb = m.Hp / 16;
  if (t == 0) { cnt[0] = 0; cnt[1] = 0; }
  __syncthreads();
  for (long s = 0; s < T; ++s) {
    const int par = (int)(s & 1);
    if (t == 0) cnt[par ^ 1] = 0;  // (its last reader, the previous step's dense part, is behind a barrier)
    if constexpr (WIN) {
      window_body<512, true>(m, st, u, smem_raw, RowSink{rows_u, cnt + par});
    } else {
      select_fast_body<512, true, true, 0, 7>(m, st, par, u, smem_raw, RowSink{rows_u, cnt + par}, (int)s, off0, off1,
                                              SelectNoHook(), 0);
    }
    __syncthreads();
    const int n = __hip_atomic_load(cnt + par, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int r0 = 0; r0 < n; r0 += 16) {
      // the row whose vectors this lane streams as the B operand (a tile's first row always exists)
      const int lrow = r0 + (t & 15);
      const RnnRow rb = rows_u[lrow < n ? lrow : r0];
      // thread t < 256 owns element (row r0 + (t >> 4), feature (t & 15) of the current feature tile)
      const int erow = r0 + ((t & 255) >> 4);
      const bool ework = t < 256 && erow < n;
      RnnRow re = rb;
      if (ework) re = rows_u[erow];
      for (int l = 0; l < m.depth; ++l) {
        if (l > 0) {  // input-side gates of layer l: gi_up[row] = b_ih + W_ih h'_{l-1}
          const float* in[1] = {hid_ptr(m, st, rb, rb.dst, l - 1)};
          for (int ft = 0; ft < m.G / 16; ++ft) {
            splitk_tile<1, 1, 1>(m.wih[l], 0, ft, nKb, in, m.bih[l] + ft * 16, 0, spart);
            if (ework) gi_up_u[(size_t)erow * m.G + ft * 16 + (t & 15)] = splitk_combine<1, 1>(spart, 0, 0, t);
            __syncthreads();
          }
        }
        const float* hsrc[1] = {rb.src >= 0 ? hid_ptr(m, st, rb, rb.src, l) : m.h1 + (size_t)l * m.Hp};
        for (int ft = 0; ft < nft; ++ft) {
          splitk_tile<3, 1, 1>(m.whh[l], nft, ft, nKb, hsrc, m.bhh[l] + ft * 16, m.Hp, spart);
          if (ework) {
            const int j = ft * 16 + (t & 15);
            const float* gi = l == 0 ? st.gi0 + (size_t)re.frame * m.G : gi_up_u + (size_t)erow * m.G;
            const float* hs = re.src >= 0 ? hid_ptr(m, st, re, re.src, l) : m.h1 + (size_t)l * m.Hp;
            const float ghr = splitk_combine<1, 3>(spart, 0, 0, t);
            const float ghz = splitk_combine<1, 3>(spart, 0, 1, t);
            const float ghn = splitk_combine<1, 3>(spart, 0, 2, t);
            const float out = j < m.H ? uis_gru_unit(gi[j], gi[m.Hp + j], gi[2 * m.Hp + j], ghr, ghz, ghn, hs[j]) : 0.0f;
            const_cast<float*>(hid_ptr(m, st, re, re.dst, l))[j] = out;
          }
          __syncthreads();
        }
      }
      {  // a1[row] = relu(b1 + W1 h'_top)
        const float* in[1] = {hid_ptr(m, st, rb, rb.dst, m.depth - 1)};
        for (int ft = 0; ft < nft; ++ft) {
          splitk_tile<1, 1, 1>(m.w1, 0, ft, nKb, in, m.b1 + ft * 16, 0, spart);
          if (ework) {
            const float v = splitk_combine<1, 1>(spart, 0, 0, t);
            a1_u[(size_t)erow * m.Hp + ft * 16 + (t & 15)] = v > 0.0f ? v : 0.0f;
          }
          __syncthreads();
        }
      }
      {  // mean = b2 + W2 a1; running-mean update (uisrnn.py:425-429) -> dst slot
        const float* in[1] = {a1_u + (size_t)(lrow < n ? lrow : r0) * m.Hp};
        for (int ft = 0; ft < m.Dp / 16; ++ft) {
          splitk_tile<1, 1, 1>(m.w2, 0, ft, nKb, in, m.b2 + ft * 16, 0, spart);
          if (ework) {
            const int f = ft * 16 + (t & 15);
            float v = splitk_combine<1, 1>(spart, 0, 0, t);
            if (re.src >= 0) v = uis_mean_update(st.pool_mean[((size_t)u * S + re.src) * m.Dp + f], v, re.nprev);
            if (f >= m.D) v = 0.0f;
            st.pool_mean[((size_t)u * S + re.dst) * m.Dp + f] = v;
          }
          __syncthreads();
        }
      }
    }
    __syncthreads();
  }
  (void)U;
}


// look_ahead >= 2: trace[-N:] from the per-window back-pointers
__global__ void k_backtrace_window(DecodeState st, int32_t* __restrict__ labels, float* __restrict__ scores,
                                   float* __restrict__ beam_scores) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= st.U) return;
  const int L = st.L;
  const long N = (long)(st.off[u + 1] - st.off[u]);
  const long T = (long)st.tau * N;
  const long n_win = (T + L - 1) / L;
  const int par = (int)(n_win & 1);
  const int nb = N > 0 ? st.beam_n[(size_t)par * st.U + u] : 0;
  const size_t e = ((size_t)par * st.U + u) * st.B;
  if (beam_scores)
    for (int b = 0; b < st.B; ++b) beam_scores[(size_t)u * st.B + b] = b < nb ? st.beam_score[e + b] : INFINITY;
  if (scores) scores[u] = nb > 0 ? st.beam_score[e] : (N > 0 ? INFINITY : 0.0f);
  if (N == 0) return;
  int32_t* out = labels + st.off[u];
  if (nb == 0) { for (long i = 0; i < N; ++i) out[i] = -1; return; }
  int r = 0;
  for (long w = n_win - 1; w >= 0; --w) {
    const long t0 = w * L;
    if (t0 + L <= T - N) break;
    const uint16_t* rec = st.bp16 + ((size_t)st.bp_base[u] + (size_t)w * st.B + r) * (L + 1);
    const int Lw = (int)((T - t0) < L ? (T - t0) : L);
    for (int k = 0; k < Lw; ++k) {
      const long tt = t0 + k;
      if (tt >= T - N) out[tt - (T - N)] = (int32_t)rec[1 + k];
    }
    r = (int)rec[0];
    if (r >= st.B) r = st.B - 1;  // records of windows that never ran (overflowed utterance) are stale
  }
}

// ------------------------------------------------------------ resident decode, many utterances
//
// k_decode_resident's stages walk the row tiles in passes of three, every pass split in K over
// the eight waves and closed by two workgroup barriers (LDS combine, epilogue): right for the
// three row tiles of 64 utterances, wasteful for the dozens of row tiles of a thousand (the passes
// of a stage run at ~55 % of their MFMA time: operand fetch, combine and epilogue are exposed in
// every pass).  k_decode_big is the same launch -- same clusters, same in-launch barriers, same
// selects from the global beam tables -- with the dense stages turned around: the workgroup's
// W_hh slice (3 gates x all k-blocks, 96 KB at hidden size 512) and linear_mean1 slice live in
// LDS, and a WAVE owns a row tile: it walks the full K of its tile (the segment chains of
// uis_numerics.h combined on the fly, as in the big-tile per-step kernels) with A operands from
// LDS and B operands (its 16 rows) streamed from L2 one segment ahead, then runs the epilogue on
// its own accumulators.  No split-K partials, no workgroup barrier inside a stage, eight row
// tiles in flight per CU.  The two mean-head weight slices share one LDS slot, refilled from L2
// at the start of their stage.  Used for
// ordinary decodes with more utterances than workgroups (U > 32 x clusters).

// NA weight streams from `wbase` (LDS or global; stream a at wbase + a * wstride, [k block][lane])
// against one row tile whose row for this lane starts at byte `boff` of `rsrc` (16 bytes per k
// block at + kb * 64 + q * 16): total[a] = this lane's 4 features x its row.
// GS = segments per operand group: the rows of group g + 1 are requested while group g is
// multiplied (two register sets in turn; everything unrolled, scheduling fenced per segment so
// that the weight reads of later segments are not hoisted into spills).
// KBS = bytes between consecutive k-blocks of this lane's row (64: a plain row; 1024: the
// k-block-major staging layout, where a wave's load is one contiguous KiB).
// The first group arrives preloaded in `bfirst` (rows_first_group); while the LAST group is
// multiplied the first group of the wave's NEXT row tile (at next_boff, if has_next) is requested
// into `bfirst` again, so that a tile's dependent start-up (row descriptor -> address -> rows)
// hides behind its predecessor's chain.
template <int GB, int KBS>
__device__ __forceinline__ void rows_first_group(__amdgpu_buffer_rsrc_t rsrc, uint32_t boff, f32x4 (&bfirst)[GB]) {
  const int q = (threadIdx.x & 63) >> 4;
#pragma unroll
  for (int k = 0; k < GB; ++k) bfirst[k] = load_sc1(rsrc, boff + (uint32_t)(k * KBS + q * 16));
}
template <int NA, int NKB, int GS, int KBS>
__device__ __forceinline__ void fullk_rows_sc1(const f32x4* wbase, int wstride, const float* const (&bias)[NA],
                                               __amdgpu_buffer_rsrc_t rsrc, uint32_t boff, f32x4 (&total)[NA],
                                               f32x4 (&bfirst)[GS * (NKB / UIS_KSPLIT)], uint32_t next_boff, bool has_next) {
  constexpr int PER = NKB / UIS_KSPLIT, NGRP = UIS_KSPLIT / GS, GB = GS * PER;
  static_assert(PER * UIS_KSPLIT == NKB && NGRP * GS == UIS_KSPLIT && NGRP % 2 == 0,
                "k-blocks divide into segments, segments into an even number of groups (the last one uses the second register set)");
  const int lane = threadIdx.x & 63, q = lane >> 4;
  f32x4 bsec[GB];  // the second register set; groups alternate bfirst / bsec
#pragma unroll
  for (int grp = 0; grp < NGRP; ++grp) {
    if (grp + 1 < NGRP) {
      if ((grp + 1) & 1) {
#pragma unroll
        for (int k = 0; k < GB; ++k) bsec[k] = load_sc1(rsrc, boff + (uint32_t)(((grp + 1) * GB + k) * KBS + q * 16));
      } else {
#pragma unroll
        for (int k = 0; k < GB; ++k) bfirst[k] = load_sc1(rsrc, boff + (uint32_t)(((grp + 1) * GB + k) * KBS + q * 16));
      }
    }
    const f32x4 (&b)[GB] = (grp & 1) ? bsec : bfirst;  // this group's operands
    if (grp + 1 == NGRP && has_next) {  // (the last group reads bsec) bfirst is free: the next tile's first group
#pragma unroll
      for (int k = 0; k < GB; ++k) bfirst[k] = load_sc1(rsrc, next_boff + (uint32_t)(k * KBS + q * 16));
    }
#pragma unroll
    for (int sg = 0; sg < GS; ++sg) {
      const int sgm = grp * GS + sg;
      f32x4 acc[NA];
#pragma unroll
      for (int a = 0; a < NA; ++a)
        acc[a] = sgm == 0 ? *reinterpret_cast<const f32x4*>(bias[a] + 4 * q) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int kb = 0; kb < PER; ++kb) {
        f32x4 wa[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) wa[a] = wbase[(size_t)a * wstride + (size_t)(sgm * PER + kb) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int a = 0; a < NA; ++a)
            acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[a][e], b[sg * PER + kb][e], acc[a], 0, 0, 0);
      }
#pragma unroll
      for (int a = 0; a < NA; ++a) {
        if (sgm == 0) total[a] = acc[a];
        else {
#pragma unroll
          for (int i = 0; i < 4; ++i) total[a][i] = total[a][i] + acc[a][i];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// LDS of k_decode_big: select scratch | W_hh slice | mean-head slot | control words | beam store
// (table sets + per-slot counts of as many of the rank's utterances as fit in what is left of 160 KB)
__host__ __device__ inline size_t big_lds_fixed(int Hp, int Dp, int B, int Kmax, int S) {
  return (size_t)((fast_lds_layout(Dp, B, Kmax, S).total + 255) & ~255) + (size_t)4 * (Hp / 16) * 64 * 16 + 64;
}
__host__ __device__ inline size_t big_store_stride(int Dp, int B, int Kmax, int S) {
  return (size_t)((2 * fast_lds_layout(Dp, B, Kmax, S).set_stride + S * 4 + 15) & ~15);
}
__host__ __device__ inline int big_store_slots(int Hp, int Dp, int B, int Kmax, int S) {
  const size_t fixed = big_lds_fixed(Hp, Dp, B, Kmax, S);
  return fixed >= 160 * 1024 ? 0 : (int)((160 * 1024 - fixed) / big_store_stride(Dp, B, Kmax, S));
}
__host__ __device__ inline size_t big_lds_bytes(int Hp, int Dp, int B, int Kmax, int S) {
  return big_lds_fixed(Hp, Dp, B, Kmax, S) + (size_t)big_store_slots(Hp, Dp, B, Kmax, S) * big_store_stride(Dp, B, Kmax, S);
}

#include "uis_select_rs.hip"

// WS (wave select): the selects of a rank's utterances run CONCURRENTLY, one wave each, on the
// single-wave select of uis_select_rs.hip (rs_prep / rs_front<FULL> / rs_back; the wave computes
// every live cluster's MSE itself) instead of one after the other on the whole workgroup: at 1024
// utterances a rank owns four, and their 4 x 8 us were a fifth of the step.  LDS of the select part:
// 1 / (2 sigma^2) | log tables | nws persistent blocks | eight per-wave scratches.
__host__ __device__ inline size_t big_ws_select_bytes(int Dp, int B, int Kmax, int S, int nws) {
  const RsLds L = rs_lds_layout(B, Kmax, S);
  return (size_t)Dp * 4 + (size_t)2 * UIS_RS_LOGTAB * 8 + (size_t)nws * L.persist_stride + (size_t)8 * L.scratch_stride;
}
__host__ __device__ inline size_t big_ws_lds_bytes(int Hp, int Dp, int B, int Kmax, int S, int nws) {
  return ((big_ws_select_bytes(Dp, B, Kmax, S, nws) + 255) & ~(size_t)255) + (size_t)4 * (Hp / 16) * 64 * 16 + 64;
}

// WIN (round 4): look_ahead >= 2 in ONE launch.  The select stage is a sub-step of the window kernel
// (window_body: expand / prune, uisrnn.py:469-477,529-559) run by the workgroup that owns the
// utterance (rank i owns utterances cluster + ncl (i + 32 k)), the dense stages are
// the ones below -- a sub-step was four launches before.  The window's work arrays and the weight
// slices take turns in the LDS: the owners refill their W_hh slice behind the window stage (96 KB from
// L2, ~2 us; the mean-head slices are refilled per stage anyway).
__host__ __device__ inline size_t big_win_lds_bytes(int Hp, int S, int NC, int Kmax, int B) {
  const size_t win = (window_lds_bytes(window_scratch_layout(S, NC, Kmax, B)) + 255) & ~(size_t)255;
  const size_t weights = (size_t)4 * (Hp / 16) * 64 * 16;
  return (win > weights ? win : weights) + 64;
}

template <int HP, int DP, bool WS = false, int CB = 0, int CK = 0, bool WIN = false>
__global__ __launch_bounds__(512) void k_decode_big(DevModel m, DecodeState st) {
  static_assert(!(WIN && WS), "one kind of select stage");
  m.Hp = HP; m.Dp = DP; m.G = 3 * HP;  // (what the template arguments say)
  if (CB && !WIN) { st.B = CB; st.Kmax = CK; st.S = CB * CK + CB; m.H = HP; m.D = DP; }  // (see k_decode_resident)
  // WIN with a fixed shape (round 5: BASELINE configs[2], beam 50 / cap 12): look_ahead 2, one intermediate level of
  // beam_size * (max_clusters + 1) hypotheses (below the level capacity: the host checks), its slots behind the beam's
  if (CB && WIN) { st.B = CB; st.Kmax = CK; st.L = 2; st.NC = CB * (CK + 1); st.S = CB * CK + CB + CB * (CK + 1); m.H = HP; m.D = DP; }
  constexpr int NKB = HP / 16;
  constexpr int NFT1 = HP / 16, SH1 = 32 / NFT1;  // ranks sharing one GRU / linear_mean1 feature tile
  constexpr int NFT2 = DP / 16, SH2 = 32 / NFT2;  // ranks sharing one linear_mean2 feature tile
  static_assert(NFT1 * SH1 == 32 && NFT2 * SH2 == 32, "hidden size 256 / 512, observation_dim 128 / 256 / 512 (padded)");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, q = lane >> 4;
  const int ncl = st.ncl;
  const int cluster = blockIdx.x % ncl, rank = blockIdx.x / ncl;
  const int U = st.U, S = st.S;
  const FastLds L = fast_lds_layout(m.Dp, st.B, st.Kmax, S);
  // WS: utterances per rank (wave k owns the rank's k-th utterance)
  const int nws = WS ? (((U + ncl - 1) / ncl) + 31) / 32 : 0;
  const RsLds RL = rs_lds_layout(st.B, st.Kmax, S);
  const size_t select_bytes = WIN ? 0 : WS ? big_ws_select_bytes(DP, st.B, st.Kmax, S, nws) : (size_t)L.total;
  f32x4* s_whh = reinterpret_cast<f32x4*>(smem_raw + ((select_bytes + 255) & ~(size_t)255));  // [3][NKB][64]
  f32x4* s_wm = s_whh + 3 * NKB * 64;                                          // [NKB][64] linear_mean1's slice, then linear_mean2's
  // [0] abort [1] steps [2] arrived (WIN: behind whichever is larger, the window's arrays or the weights)
  int* s_ctl = WIN ? reinterpret_cast<int*>(smem_raw + big_win_lds_bytes(HP, S, st.NC, st.Kmax, st.B) - 64)
                   : reinterpret_cast<int*>(s_wm + NKB * 64);
  // the beams of this rank's first `nstore` utterances stay in LDS from step to step (the rest, if
  // the rank has more, goes through the global tables every step)
  unsigned char* s_store = reinterpret_cast<unsigned char*>(s_ctl + 16);
  const int store_stride = (int)big_store_stride(m.Dp, st.B, st.Kmax, S);
  const int nstore = (WS || WIN) ? 0 : big_store_slots(HP, m.Dp, st.B, st.Kmax, S);
  // WIN: this workgroup's utterance, if it owns one
  // WIN: this workgroup's utterances (cluster + ncl (rank + 32 i)); with exactly one -- up to 32 per XCD --
  // the first part of its next sub-step runs in the shadow of the step's last barrier
  const int u_own = cluster + ncl * rank;
  const bool win_owner = WIN && u_own < U;
  const bool win_single = win_owner && u_own + 32 * ncl >= U;
  // WS: the select part of the LDS
  float* ws_swgt = reinterpret_cast<float*>(smem_raw);
  double* ws_lblk = reinterpret_cast<double*>(smem_raw + (size_t)DP * 4);
  double* ws_lden = ws_lblk + UIS_RS_LOGTAB;
  unsigned char* ws_pers = reinterpret_cast<unsigned char*>(ws_lden + UIS_RS_LOGTAB);
  unsigned char* ws_scr = ws_pers + (size_t)nws * RL.persist_stride;
  const int wu = __builtin_amdgcn_readfirstlane(w);
  const int u_w = cluster + ncl * (rank + 32 * wu);
  const bool has_u = WS && wu < nws && u_w < U;
  unsigned char* const pers_w = ws_pers + (size_t)(has_u ? wu : 0) * RL.persist_stride;
  unsigned char* const scr_w = ws_scr + (size_t)wu * RL.scratch_stride;
  long off0_w = 0, N_w = 0, fpos_w = 0;
  if (has_u) { off0_w = (long)st.off[u_w]; N_w = (long)st.off[u_w + 1] - off0_w; }
  const long T_w = (long)st.tau * N_w;

  uint32_t xcc = 0;
  if (t == 0) {
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xfu;
    if (rank == 0) __hip_atomic_store(st.cl_xcc + cluster, xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_ctl[0] = 0; s_ctl[1] = 0; s_ctl[2] = 0;
  }
  if (WS) {
    for (int i = t; i < DP; i += 512) ws_swgt[i] = m.wgt[i];
    for (int i = t; i < UIS_RS_LOGTAB; i += 512) { ws_lblk[i] = st.logblk[i]; ws_lden[i] = st.logden[i]; }
    // beam_set = [BeamState()] (uisrnn.py:528): one empty hypothesis, nothing live
    if (has_u)
      for (int i = lane; i < RL.persist_stride / 4; i += 64) reinterpret_cast<int*>(pers_w)[i] = 0;
  }
  __syncthreads();
  if (WS && has_u && lane == 0) {
    int* hdr = reinterpret_cast<int*>(pers_w + RL.off_hdr);
    hdr[0] = 1; hdr[1] = 1; hdr[2] = 1 << 20;  // one hypothesis, grid stride 1
    reinterpret_cast<int*>(pers_w + RL.off_hyp)[1] = -1;  // {K 0, last -1, sum 0, score 0}
  }
  {  // decode steps of this cluster = the longest of its utterances
    int myT = 0;
    for (int i = t; cluster + ncl * i < U; i += 512) {
      const int u = cluster + ncl * i;
      const long T = (long)st.tau * (long)(st.off[u + 1] - st.off[u]);
      myT = T > myT ? (int)T : myT;
    }
    if (myT > 0) atomicMax(&s_ctl[1], myT);
  }
  const int ft1 = rank / SH1, tpar1 = rank % SH1;  // ranks sharing a feature tile take alternate row tiles
  const int ft2 = rank / SH2, tpar2 = rank % SH2;
  for (int e = t; e < NKB * 64; e += 512) {
#pragma unroll
    for (int g = 0; g < 3; ++g)
      s_whh[g * NKB * 64 + e] = reinterpret_cast<const f32x4*>(m.whh[0])[(size_t)(g * NFT1 + ft1) * NKB * 64 + e];
  }
  __syncthreads();
  const int nsteps = s_ctl[1];
  // (round 5, WS only) this launch runs steps [step0, s_end) of the decode: a launch that starts late picks up the
  // persistent blocks the previous one left in st.resume (one per utterance, by the wave that owns it)
  const int step0 = WS ? st.step0 : 0;
  const int s_end = (WS && st.step1 > 0 && st.step1 < nsteps) ? st.step1 : nsteps;
  if (WS && step0 > 0 && has_u) {
    const int* src = reinterpret_cast<const int*>(st.resume + (size_t)u_w * RL.persist_stride);
    for (int i = lane; i < RL.persist_stride / 4; i += 64) reinterpret_cast<int*>(pers_w)[i] = src[i];
    fpos_w = N_w > 0 ? (long)step0 % N_w : 0;
  }
  const f32x4* w1g = reinterpret_cast<const f32x4*>(m.w1) + (size_t)ft1 * NKB * 64;
  const f32x4* w2g = reinterpret_cast<const f32x4*>(m.w2) + (size_t)ft2 * NKB * 64;

  const __amdgpu_buffer_rsrc_t rs_rows =
      __builtin_amdgcn_make_buffer_rsrc((void*)st.rows, (short)0, 0x7fffffff, 0x00020000);
  // (round 5: the cluster states, hand-off tiles and means are addressed with UNSIGNED 32-bit byte offsets through
  // descriptors of 4 GB - 1: a look-ahead decode of 1024 utterances at beam 50 -- 2.7 GB of hidden states -- stays in
  // one launch; rounds 2-4 stopped at 2 GB)
  const __amdgpu_buffer_rsrc_t rs_hid =
      __builtin_amdgcn_make_buffer_rsrc((void*)st.pool_hid, (short)0, 0xffffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a1 =
      __builtin_amdgcn_make_buffer_rsrc((void*)st.a1, (short)0, 0xffffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_mean =
      __builtin_amdgcn_make_buffer_rsrc((void*)st.pool_mean, (short)0, 0xffffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_hst =
      __builtin_amdgcn_make_buffer_rsrc((void*)st.gi_up, (short)0, 0xffffffff, 0x00020000);
  // (hand-off buffers h' -> linear_mean1, a1 -> linear_mean2: k-block major: rs_hst, rs_a1)
  const size_t tile0 = (size_t)(cluster * st.rx_stride) >> 4;  // first row tile of this cluster
  const int rbase = cluster * st.rx_stride;   // this cluster's rows of `rows`
  const uint32_t h1_off = (uint32_t)((size_t)U * S * HP * 4);  // the extra slot holding h1
  RowSink sink{st.rows + rbase, nullptr};
  uint32_t bar = 0;
  const float* bias_hh[3] = {m.bhh[0] + ft1 * 16, m.bhh[0] + HP + ft1 * 16, m.bhh[0] + 2 * HP + ft1 * 16};
  const float* bias_1[1] = {m.b1 + ft1 * 16};
  const float* bias_2[1] = {m.b2 + ft2 * 16};

#if defined(UIS_RESIDENT_TIMING)
  unsigned long long rt_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // (WIN: [8 ..) the odd sub-steps)
  unsigned long long wv_acc = 0;
  unsigned long long nrt_hist[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // row tiles mod 8 per step, and their sum (workgroup 0)
  unsigned long long rt_prev = wall_clock64();
#endif
  for (int s = step0; s < s_end; ++s) {
    const int par = s & 1;
    sink.count = st.rx_nrows + cluster * 32 + par;
    if constexpr (WS) {
      // every utterance of this rank at once, one wave each: candidate grid, MSEs, scores, prune,
      // winners; the rows go to the cluster's list (their order is whatever the reservations make
      // it: nothing depends on it); the table update runs inside the barrier
      const bool act_w = has_u && (long)s < T_w;
      const long frame_w = off0_w + fpos_w;
      RsWin win;
      win.keep = 0; win.C = 0; win.nlead = 0; win.a = 0u; win.b = 0u; win.c = 0u; win.score = 0.0f;
      if (act_w) {
        const RsDims dm{st.B, st.Kmax, S, m.D};
        const RsPrep<3> prep = rs_prep<true, 3>(m, st, RL, dm, s, pers_w, scr_w, ws_lblk, ws_lden, []() {});
        win = rs_front<DP, true, 3>(m, st, RL, dm, u_w, s, frame_w, pers_w, scr_w, rs_mean /* unused: FULL */, 0u, prep, nullptr, ws_swgt);
        int row_base = 0;
        if (lane == 0 && win.nlead > 0) row_base = atomicAdd(sink.count, win.nlead);
        row_base = __shfl(row_base, 0, 64);
        if (win.is_lead()) {
          RnnRow rr; rr.utt = u_w; rr.src = win.src(); rr.dst = win.dst(); rr.nprev = win.nprev(); rr.frame = frame_w; rr.pad = 0;
          sink.rows[row_base + win.ord()] = rr;
        }
      }
      RSTAMP(0 + (WIN ? 8 * (s & 1) : 0));
      xcd_arrive(st, cluster, s_ctl);
      if (act_w) {
        rs_back<3>(m, st, RL, RsDims{st.B, st.Kmax, S, m.D}, u_w, s, off0_w, pers_w, true, win, []() {});
        fpos_w = fpos_w + 1 == N_w ? 0 : fpos_w + 1;
      }
      if (rs_xcd_wait(st, cluster, 32u * ++bar, s_ctl)) return;
    } else if constexpr (WIN) {
      // this sub-step of the owned utterance's window: scores, expand / prune, next level or beam, rows
      // (its first part ran while this workgroup waited at the previous sub-step's last barrier)
      // (ONE call site for the whole body: inlined twice it doubled the kernel and spilled into the dense loops)
      if (win_owner) {
        for (int u = u_own; u < U; u += 32 * ncl) {
          window_body<512, true>(m, st, u, smem_raw, sink, win_single && s > 0 ? 2 : 0);
          if (!win_single) __syncthreads();
        }
      }
      RSTAMP(0 + (WIN ? 8 * (s & 1) : 0));
      if (xcd_barrier(st, cluster, 32u * ++bar, s_ctl)) return;
      if (win_owner) {  // the window's arrays sat where the W_hh slice lives
        for (int e = t; e < NKB * 64; e += 512) {
#pragma unroll
          for (int g = 0; g < 3; ++g)
            s_whh[g * NKB * 64 + e] = reinterpret_cast<const f32x4*>(m.whh[0])[(size_t)(g * NFT1 + ft1) * NKB * 64 + e];
        }
      }
      __syncthreads();
    } else {
      for (int i = rank, k = 0; cluster + ncl * i < U; i += 32, ++k) {
        const int u = cluster + ncl * i;
        if (k < nstore) {
          unsigned char* blk = s_store + (size_t)k * store_stride;
          select_fast_body<512, true, true, DP, 7>(m, st, par, u, smem_raw, sink, s, (long)st.off[u], (long)st.off[u + 1], SelectNoHook(), 0,
                                                   blk - L.off_slot, reinterpret_cast<int*>(blk + 2 * L.set_stride));
        } else {
          select_fast_body<512, true, false, DP>(m, st, par, u, smem_raw, sink);
        }
        __syncthreads();
      }
      RSTAMP(0 + (WIN ? 8 * (s & 1) : 0));
      if (xcd_barrier(st, cluster, 32u * ++bar, s_ctl)) return;
    }
    RSTAMP(1 + (WIN ? 8 * (s & 1) : 0));
    if (s == step0 && t == 0 && rank == 1 && (st.flags & 0x100u)) xcc ^= 1u;  // UIS_FLAG_TEST_MISPLACED: pretend
    if (s == step0 && t == 0 && __hip_atomic_load(st.cl_xcc + cluster, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != xcc)
      __hip_atomic_store(st.cl_abort, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // not on one XCD
    const int nrows = __hip_atomic_load(st.rx_nrows + cluster * 32 + par, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (rank == 0 && t == 0)
      __hip_atomic_store(st.rx_nrows + cluster * 32 + (par ^ 1), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int nrt = (nrows + 15) >> 4;
#if defined(UIS_RESIDENT_TIMING)
    const unsigned long long wv_t0 = wall_clock64();
#endif

    // ---- GRU: h' = gru(gi0[frame], W_hh h_src + b_hh) -> dst slot; wave w of this rank takes the
    // row tiles tpar1 + SH1 * (w, w + 8, ...); the next tile's descriptor and first rows are
    // requested while the current tile's chain runs
    {
      constexpr int GSG = 2, GBG = GSG * (NKB / UIS_KSPLIT);
      int tile = tpar1 + SH1 * w;
      RowHead rh{0, 0, 0, 0};
      long frame = 0;
      uint32_t hoff = h1_off;
      f32x4 bfirst[GBG];
      auto fetch_head = [&](int tl, RowHead& h_, long& f_, uint32_t& o_) {
        const int row = 16 * tl + (lane & 15);
        const int use = rbase + (row < nrows ? row : 16 * tl);  // (a tile's first row always exists)
        h_ = load_row_head(rs_rows, use);
        f_ = load_row_frame(rs_rows, use);
        o_ = h_.src >= 0 ? (uint32_t)((((size_t)h_.utt * S + h_.src) * HP) * 4) : h1_off;
      };
      if (tile < nrt) {
        fetch_head(tile, rh, frame, hoff);
        rows_first_group<GBG, 64>(rs_hid, hoff, bfirst);
      }
      while (tile < nrt) {
        const int next = tile + SH1 * 8;
        const bool has_next = next < nrt;
        RowHead rh_n{0, 0, 0, 0};
        long frame_n = 0;
        uint32_t hoff_n = h1_off;
        if (has_next) fetch_head(next, rh_n, frame_n, hoff_n);
        const bool valid = 16 * tile + (lane & 15) < nrows;
        const int j4 = ft1 * 16 + 4 * q;
        const float* gi = st.gi0 + (size_t)frame * (3 * HP);
        const f32x4 gir = *reinterpret_cast<const f32x4*>(gi + j4);
        const f32x4 giz = *reinterpret_cast<const f32x4*>(gi + HP + j4);
        const f32x4 gin = *reinterpret_cast<const f32x4*>(gi + 2 * HP + j4);
        const f32x4 hprev = load_sc1(rs_hid, hoff + (uint32_t)(j4 * 4));
        f32x4 gh[3];
        fullk_rows_sc1<3, NKB, GSG, 64>(s_whh, NKB * 64, bias_hh, rs_hid, hoff, gh, bfirst, hoff_n, has_next);
        if (valid) {
          f32x4 out;
#pragma unroll
          for (int i = 0; i < 4; ++i)
            out[i] = j4 + i < m.H ? uis_gru_unit(gir[i], giz[i], gin[i], gh[0][i], gh[1][i], gh[2][i], hprev[i]) : 0.0f;
          rs_buf_store_f32x4(rs_hid, (uint32_t)((rh.utt * S + rh.dst) * HP + j4) * 4u, out);
          // ... and the copy linear_mean1 streams: [row tile][feature tile][16 rows][16], so that a
          // consumer wave's 16-byte-per-lane load is one contiguous KiB (plain rows cost one 64-byte L2
          // request per row and k-block: the request rate, not the MFMA chain, bounded the heads)
          rs_buf_store_f32x4(rs_hst, (uint32_t)((((int)tile0 + tile) * NFT1 + ft1) * 256 + (lane & 15) * 16 + 4 * q) * 4u, out);
        }
        tile = next; rh = rh_n; frame = frame_n; hoff = hoff_n;
      }
    }
#if defined(UIS_RESIDENT_TIMING)
    if (!WIN || !(s & 1)) wv_acc += wall_clock64() - wv_t0;  // this wave's own GRU time (WIN: even sub-steps)
    if (!WIN && t == 0 && blockIdx.x == 0) { ++nrt_hist[nrt & 7]; nrt_hist[8] += nrt; }
#endif
    RSTAMP(2 + (WIN ? 8 * (s & 1) : 0));
    // ---- linear_mean1 + relu -> a1 (same staging layout); its weight slice takes the LDS slot
    // (32 KB from L2) between this workgroup's arrival at the barrier and the barrier's completion
    xcd_arrive(st, cluster, s_ctl);
    for (int e = t; e < NKB * 64; e += 512) s_wm[e] = w1g[e];
    if (rs_xcd_wait(st, cluster, 32u * ++bar, s_ctl)) return;
    RSTAMP(3 + (WIN ? 8 * (s & 1) : 0));
    constexpr int GBH = 4 * (NKB / UIS_KSPLIT);
    auto stage_off = [&](int tl) { return (uint32_t)((((tile0 + tl) * NFT1) * 256 + (lane & 15) * 16) * 4); };
    for (int tile = tpar1 + SH1 * w; tile < nrt; tile += SH1 * 8) {
      const int row = 16 * tile + (lane & 15);
      const bool valid = row < nrows;
      f32x4 v[1], bfirst1[GBH];
      rows_first_group<GBH, 1024>(rs_hst, stage_off(tile), bfirst1);
      fullk_rows_sc1<1, NKB, 4, 1024>(s_wm, 0, bias_1, rs_hst, stage_off(tile), v, bfirst1, 0u, false);
      if (valid) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[0][i] = v[0][i] > 0.0f ? v[0][i] : 0.0f;
        rs_buf_store_f32x4(rs_a1, (uint32_t)((((int)tile0 + tile) * NFT1 + ft1) * 256 + (lane & 15) * 16 + 4 * q) * 4u, v[0]);
      }
    }
    RSTAMP(4 + (WIN ? 8 * (s & 1) : 0));
    // ---- linear_mean2 + running mean -> dst slot; its slice likewise (every wave of this workgroup
    // is past linear_mean1 once the arrival's workgroup barrier is: the slot is free)
    xcd_arrive(st, cluster, s_ctl);
    for (int e = t; e < NKB * 64; e += 512) s_wm[e] = w2g[e];
    if (rs_xcd_wait(st, cluster, 32u * ++bar, s_ctl)) return;
    RSTAMP(5 + (WIN ? 8 * (s & 1) : 0));
    for (int tile = tpar2 + SH2 * w; tile < nrt; tile += SH2 * 8) {
      const int row = 16 * tile + (lane & 15);
      const bool valid = row < nrows;
      const RowHead rh = load_row_head(rs_rows, rbase + (valid ? row : 16 * tile));
      const int f4 = ft2 * 16 + 4 * q;
      f32x4 old = {0.0f, 0.0f, 0.0f, 0.0f};
      if (valid && rh.src >= 0) old = load_sc1(rs_mean, (uint32_t)((rh.utt * S + rh.src) * DP + f4) * 4u);
      f32x4 v[1], bfirst2[GBH];
      rows_first_group<GBH, 1024>(rs_a1, stage_off(tile), bfirst2);
      fullk_rows_sc1<1, NKB, 4, 1024>(s_wm, 0, bias_2, rs_a1, stage_off(tile), v, bfirst2, 0u, false);
      if (valid) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (rh.src >= 0) v[0][i] = uis_mean_update(old[i], v[0][i], rh.nprev);
          if (f4 + i >= m.D) v[0][i] = 0.0f;
        }
        rs_buf_store_f32x4(rs_mean, (uint32_t)((rh.utt * S + rh.dst) * DP + f4) * 4u, v[0]);
      }
    }
    RSTAMP(6 + (WIN ? 8 * (s & 1) : 0));
    if constexpr (WIN) {
      // the next sub-step's live list and candidate offsets need nothing the other workgroups are still
      // writing: between the arrival and the barrier's completion (this workgroup's own linear_mean2
      // is done: the LDS is free; the W_hh slice is reloaded behind the window stage anyway)
      xcd_arrive(st, cluster, s_ctl);
      if (win_single && s + 1 < nsteps) window_body<512, true>(m, st, u_own, smem_raw, sink, 1);
      if (rs_xcd_wait(st, cluster, 32u * ++bar, s_ctl)) return;
    } else {
      if (xcd_barrier(st, cluster, 32u * ++bar, s_ctl)) return;
    }
    RSTAMP(7 + (WIN ? 8 * (s & 1) : 0));
  }
#if defined(UIS_RESIDENT_TIMING)
  if (t == 0 && (blockIdx.x == 0 || blockIdx.x == 248))
    for (int k = 0; k < (WIN ? 16 : 8); ++k) st.counters[(blockIdx.x == 0 ? 48 : 64) + k] = rt_acc[k];
  if (WIN && t == 0) {  // every workgroup's GRU time and wait behind it (even sub-steps), and the same for the mean heads
    st.counters[96 + blockIdx.x] = rt_acc[2];
    st.counters[96 + 256 + blockIdx.x] = rt_acc[3];
    st.counters[96 + 512 + blockIdx.x] = rt_acc[4];
    st.counters[96 + 768 + blockIdx.x] = rt_acc[6];
  }
  if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 248)) st.counters[(WIN ? 80 : 96) + (blockIdx.x ? 8 : 0) + w] = wv_acc;
  if (!WIN && t == 0 && blockIdx.x == 0)
    for (int k = 0; k < 9; ++k) st.counters[112 + k] = nrt_hist[k];
#endif
  if (WS && s_end < nsteps) {  // more steps to come in another launch
    if (has_u) {
      int* dst = reinterpret_cast<int*>(st.resume + (size_t)u_w * RL.persist_stride);
      for (int i = lane; i < RL.persist_stride / 4; i += 64) dst[i] = reinterpret_cast<const int*>(pers_w)[i];
    }
    return;
  }
  if (WS && has_u && lane == 0) {  // this utterance's statistics
    const unsigned long long* acc = reinterpret_cast<const unsigned long long*>(pers_w + RL.off_stats);
    atomicAdd(&st.counters[0], acc[0]);
    atomicAdd(&st.counters[1], acc[1]);
    atomicAdd(&st.counters[2], acc[2]);
    atomicMax(&st.counters[3], acc[3]);
  }
}

#if defined(UIS_WITH_COHORTS)
#include "uis_decode_coh.hip"
#endif

// rnn_depth >= 2 in ONE launch (round 4): k_decode_big's grid, barriers and wave-per-row-tile stages with one
// more pair of stages per upper layer.  A workgroup's 96 KB weight slot cannot hold W_hh of every layer and
// W_ih of the upper ones, so the slot is REFILLED stage by stage from L2 / the Infinity Cache (the price of
// depth in this design: three refills per step at depth 2, where the copy can it runs between a barrier's
// arrival and its completion).  Per step: selects (each utterance on its owner workgroup, tables in global
// memory) | layer 0: GRU | layers l >= 1: input-side gates gi = b_ih + W_ih h'_{l-1} into gi_up (read back by the
// very wave that wrote them: no barrier, only the slot swap), then the GRU of layer l | linear_mean1 |
// linear_mean2.  A layer's h' goes to its cluster-state slot and, k-block-major, to the hand-off buffer
// hst[l & 1] that the next stage of EVERY rank streams.  Same canonical sums as everywhere (uis_numerics.h).
__host__ __device__ inline size_t deep_lds_bytes(int Hp, int Dp, int B, int Kmax, int S) {
  return (size_t)((fast_lds_layout(Dp, B, Kmax, S).total + 255) & ~255) + (size_t)4 * (Hp / 16) * 64 * 16 + 64;
}
// WIN: look_ahead >= 2 -- the select stage is a sub-step of the window kernel run by the utterance's owner
// workgroup, as in k_decode_big<WIN>; its work arrays lie over the weight slot, which is refilled behind it anyway.
template <int HP, int DP, bool WIN = false>
__global__ __launch_bounds__(512) void k_decode_deep(DevModel m, DecodeState st) {
  // (m.Hp == HP, m.Dp == DP: the host picks the instantiation; m is NOT written here -- its per-layer pointer
  // arrays are indexed by a run-time layer number and stay in the kernel-argument segment only if it is read-only)
  constexpr int NKB = HP / 16;
  constexpr int NFT1 = HP / 16, SH1 = 32 / NFT1;
  constexpr int NFT2 = DP / 16, SH2 = 32 / NFT2;
  static_assert(NFT1 * SH1 == 32 && NFT2 * SH2 == 32, "hidden size 128 / 256 / 512, observation_dim 128 / 256 / 512 (padded)");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, q = lane >> 4;
  const int ncl = st.ncl;
  const int cluster = blockIdx.x % ncl, rank = blockIdx.x / ncl;
  const int U = st.U, S = st.S, depth = m.depth;
  const FastLds L = fast_lds_layout(m.Dp, st.B, st.Kmax, S);
  f32x4* s_slot = reinterpret_cast<f32x4*>(smem_raw + (WIN ? (size_t)0 : (((size_t)L.total + 255) & ~(size_t)255)));  // [3][NKB][64]: W_hh / W_ih of the layer at hand
  f32x4* s_wm = s_slot + 3 * NKB * 64;                                                            // [NKB][64]: linear_mean1's slice, then linear_mean2's
  // [0] abort [1] steps [2] arrived (WIN: behind whichever is larger, the window's arrays or the weights)
  int* s_ctl = WIN ? reinterpret_cast<int*>(smem_raw + big_win_lds_bytes(HP, S, st.NC, st.Kmax, st.B) - 64)
                   : reinterpret_cast<int*>(s_wm + NKB * 64);
  uint32_t xcc = 0;
  if (t == 0) {
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xfu;
    if (rank == 0) __hip_atomic_store(st.cl_xcc + cluster, xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_ctl[0] = 0; s_ctl[1] = 0; s_ctl[2] = 0;
  }
  __syncthreads();
  {
    int myT = 0;
    for (int i = t; cluster + ncl * i < U; i += 512) {
      const int u = cluster + ncl * i;
      const long T = (long)st.tau * (long)(st.off[u + 1] - st.off[u]);
      myT = T > myT ? (int)T : myT;
    }
    if (myT > 0) atomicMax(&s_ctl[1], myT);
  }
  __syncthreads();
  const int nsteps = s_ctl[1];
  const int ft1 = rank / SH1, tpar1 = rank % SH1;
  const int ft2 = rank / SH2, tpar2 = rank % SH2;
  auto fill_slot = [&](const float* wmat) {  // this rank's feature tile of a [3 gates][HP] x HP matrix in tile order
    for (int e = t; e < NKB * 64; e += 512) {
#pragma unroll
      for (int g = 0; g < 3; ++g) s_slot[g * NKB * 64 + e] = reinterpret_cast<const f32x4*>(wmat)[(size_t)(g * NFT1 + ft1) * NKB * 64 + e];
    }
  };
  const f32x4* w1g = reinterpret_cast<const f32x4*>(m.w1) + (size_t)ft1 * NKB * 64;
  const f32x4* w2g = reinterpret_cast<const f32x4*>(m.w2) + (size_t)ft2 * NKB * 64;
  const __amdgpu_buffer_rsrc_t rs_rows = __builtin_amdgcn_make_buffer_rsrc((void*)st.rows, (short)0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_hid = __builtin_amdgcn_make_buffer_rsrc((void*)st.pool_hid, (short)0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a1 = __builtin_amdgcn_make_buffer_rsrc((void*)st.a1, (short)0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_mean = __builtin_amdgcn_make_buffer_rsrc((void*)st.pool_mean, (short)0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_gup = __builtin_amdgcn_make_buffer_rsrc((void*)st.gi_up, (short)0, 0x7fffffff, 0x00020000);
  // (one descriptor per use, built from the layer's parity: an indexed pair of descriptors lives in scratch)
  auto hs_rsrc = [&](int l_) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(st.hst + ((l_ & 1) ? st.hst_elems : (size_t)0)), (short)0, 0x7fffffff, 0x00020000);
  };
  const size_t tile0 = (size_t)(cluster * st.rx_stride) >> 4;
  const int rbase = cluster * st.rx_stride;
  const uint32_t h1_off = (uint32_t)((size_t)U * S * depth * HP * 4);  // the extra slot: h1 of every layer
  RowSink sink{st.rows + rbase, nullptr};
  uint32_t bar = 0;
  const float* bias_1[1] = {m.b1 + ft1 * 16};
  const float* bias_2[1] = {m.b2 + ft2 * 16};
  constexpr int GSG = 2, GBG = GSG * (NKB / UIS_KSPLIT);
  constexpr int GBH = 4 * (NKB / UIS_KSPLIT);
  auto stage_off = [&](int tl) { return (uint32_t)((((tile0 + tl) * NFT1) * 256 + (lane & 15) * 16) * 4); };

  for (int s = 0; s < nsteps; ++s) {
    const int par = s & 1;
    sink.count = st.rx_nrows + cluster * 32 + par;
    for (int i = rank; cluster + ncl * i < U; i += 32) {
      if constexpr (WIN) window_body<512, true>(m, st, cluster + ncl * i, smem_raw, sink, 0);
      else select_fast_body<512, true, false, DP>(m, st, par, cluster + ncl * i, smem_raw, sink);
      __syncthreads();
    }
    // (the slot last held W_hh of the top layer: layer 0's goes in while the barrier completes)
    xcd_arrive(st, cluster, s_ctl);
    fill_slot(m.whh[0]);
    if (rs_xcd_wait(st, cluster, 32u * ++bar, s_ctl)) return;
    if (s == 0 && t == 0 && __hip_atomic_load(st.cl_xcc + cluster, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != xcc)
      __hip_atomic_store(st.cl_abort, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // not on one XCD
    const int nrows = __hip_atomic_load(st.rx_nrows + cluster * 32 + par, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (rank == 0 && t == 0)
      __hip_atomic_store(st.rx_nrows + cluster * 32 + (par ^ 1), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int nrt = (nrows + 15) >> 4;

    for (int l = 0; l < depth; ++l) {
      if (l > 0) {
        // ---- input-side gates of layer l: gi_up[row] = b_ih + W_ih h'_{l-1} (this rank's 16 units of each gate;
        // W_ih went into the slot while the barrier behind layer l - 1 completed)
        const float* bias_ih[3] = {m.bih[l] + ft1 * 16, m.bih[l] + HP + ft1 * 16, m.bih[l] + 2 * HP + ft1 * 16};
        const __amdgpu_buffer_rsrc_t rs_below = hs_rsrc(l - 1);
        for (int tile = tpar1 + SH1 * w; tile < nrt; tile += SH1 * 8) {
          const int row = 16 * tile + (lane & 15);
          f32x4 gi[3], bfirst[GBG];
          rows_first_group<GBG, 1024>(rs_below, stage_off(tile), bfirst);
          fullk_rows_sc1<3, NKB, GSG, 1024>(s_slot, NKB * 64, bias_ih, rs_below, stage_off(tile), gi, bfirst, 0u, false);
          if (row < nrows) {
            const uint32_t go = (uint32_t)(((rbase + row) * 3 * HP + ft1 * 16 + 4 * q) * 4);
            rs_buf_store_f32x4(rs_gup, go, gi[0]);
            rs_buf_store_f32x4(rs_gup, go + (uint32_t)(HP * 4), gi[1]);
            rs_buf_store_f32x4(rs_gup, go + (uint32_t)(2 * HP * 4), gi[2]);
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // every wave is done with W_ih: the slot takes W_hh of this layer
        fill_slot(m.whh[l]);
        __syncthreads();
      }
      // ---- GRU of layer l: h' = gru(gi, W_hh h_src + b_hh) -> dst slot (layer l) and hst[l & 1]
      {
        const float* bias_hh[3] = {m.bhh[l] + ft1 * 16, m.bhh[l] + HP + ft1 * 16, m.bhh[l] + 2 * HP + ft1 * 16};
        const uint32_t lay = (uint32_t)(l * HP * 4);
        const __amdgpu_buffer_rsrc_t rs_out = hs_rsrc(l);
        int tile = tpar1 + SH1 * w;
        RowHead rh{0, 0, 0, 0};
        long frame = 0;
        uint32_t hoff = h1_off + lay;
        f32x4 bfirst[GBG];
        auto fetch_head = [&](int tl, RowHead& h_, long& f_, uint32_t& o_) {
          const int row = 16 * tl + (lane & 15);
          const int use = rbase + (row < nrows ? row : 16 * tl);  // (a tile's first row always exists)
          h_ = load_row_head(rs_rows, use);
          f_ = load_row_frame(rs_rows, use);
          o_ = h_.src >= 0 ? (uint32_t)((((size_t)h_.utt * S + h_.src) * depth * HP) * 4) + lay : h1_off + lay;
        };
        if (tile < nrt) {
          fetch_head(tile, rh, frame, hoff);
          rows_first_group<GBG, 64>(rs_hid, hoff, bfirst);
        }
        while (tile < nrt) {
          const int next = tile + SH1 * 8;
          const bool has_next = next < nrt;
          RowHead rh_n{0, 0, 0, 0};
          long frame_n = 0;
          uint32_t hoff_n = h1_off + lay;
          if (has_next) fetch_head(next, rh_n, frame_n, hoff_n);
          const int row = 16 * tile + (lane & 15);
          const bool valid = row < nrows;
          const int j4 = ft1 * 16 + 4 * q;
          f32x4 gir, giz, gin;
          if (l == 0) {
            const float* gi = st.gi0 + (size_t)frame * (3 * HP);
            gir = *reinterpret_cast<const f32x4*>(gi + j4);
            giz = *reinterpret_cast<const f32x4*>(gi + HP + j4);
            gin = *reinterpret_cast<const f32x4*>(gi + 2 * HP + j4);
          } else {  // (written a moment ago by this very wave, but last step's line may sit in this CU's L1)
            const uint32_t go = (uint32_t)(((rbase + (valid ? row : 16 * tile)) * 3 * HP + j4) * 4);
            gir = load_sc1(rs_gup, go);
            giz = load_sc1(rs_gup, go + (uint32_t)(HP * 4));
            gin = load_sc1(rs_gup, go + (uint32_t)(2 * HP * 4));
          }
          const f32x4 hprev = load_sc1(rs_hid, hoff + (uint32_t)(j4 * 4));
          f32x4 gh[3];
          fullk_rows_sc1<3, NKB, GSG, 64>(s_slot, NKB * 64, bias_hh, rs_hid, hoff, gh, bfirst, hoff_n, has_next);
          if (valid) {
            f32x4 out;
#pragma unroll
            for (int i = 0; i < 4; ++i)
              out[i] = j4 + i < m.H ? uis_gru_unit(gir[i], giz[i], gin[i], gh[0][i], gh[1][i], gh[2][i], hprev[i]) : 0.0f;
            rs_buf_store_f32x4(rs_hid, (uint32_t)((((size_t)rh.utt * S + rh.dst) * depth * HP + j4) * 4) + lay, out);
            rs_buf_store_f32x4(rs_out, (uint32_t)((((int)tile0 + tile) * NFT1 + ft1) * 256 + (lane & 15) * 16 + 4 * q) * 4u, out);
          }
          tile = next; rh = rh_n; frame = frame_n; hoff = hoff_n;
        }
      }
      // every rank's slice of h'_l has to be there before anybody streams it
      xcd_arrive(st, cluster, s_ctl);
      if (l + 1 == depth) {
        for (int e = t; e < NKB * 64; e += 512) s_wm[e] = w1g[e];  // linear_mean1's slice, in the barrier's shadow
      } else {
        fill_slot(m.wih[l + 1]);  // (this workgroup's waves are past the GRU: the slot is free)
      }
      if (rs_xcd_wait(st, cluster, 32u * ++bar, s_ctl)) return;
    }

    // ---- linear_mean1 + relu -> a1 (hand-off layout)
    const __amdgpu_buffer_rsrc_t rs_top = hs_rsrc(depth - 1);
    for (int tile = tpar1 + SH1 * w; tile < nrt; tile += SH1 * 8) {
      const int row = 16 * tile + (lane & 15);
      f32x4 v[1], bfirst1[GBH];
      rows_first_group<GBH, 1024>(rs_top, stage_off(tile), bfirst1);
      fullk_rows_sc1<1, NKB, 4, 1024>(s_wm, 0, bias_1, rs_top, stage_off(tile), v, bfirst1, 0u, false);
      if (row < nrows) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[0][i] = v[0][i] > 0.0f ? v[0][i] : 0.0f;
        rs_buf_store_f32x4(rs_a1, (uint32_t)((((int)tile0 + tile) * NFT1 + ft1) * 256 + (lane & 15) * 16 + 4 * q) * 4u, v[0]);
      }
    }
    xcd_arrive(st, cluster, s_ctl);
    for (int e = t; e < NKB * 64; e += 512) s_wm[e] = w2g[e];
    if (rs_xcd_wait(st, cluster, 32u * ++bar, s_ctl)) return;

    // ---- linear_mean2 + running mean -> dst slot
    for (int tile = tpar2 + SH2 * w; tile < nrt; tile += SH2 * 8) {
      const int row = 16 * tile + (lane & 15);
      const bool valid = row < nrows;
      const RowHead rh = load_row_head(rs_rows, rbase + (valid ? row : 16 * tile));
      const int f4 = ft2 * 16 + 4 * q;
      f32x4 old = {0.0f, 0.0f, 0.0f, 0.0f};
      if (valid && rh.src >= 0) old = load_sc1(rs_mean, (uint32_t)(((rh.utt * S + rh.src) * DP + f4) * 4));
      f32x4 v[1], bfirst2[GBH];
      rows_first_group<GBH, 1024>(rs_a1, stage_off(tile), bfirst2);
      fullk_rows_sc1<1, NKB, 4, 1024>(s_wm, 0, bias_2, rs_a1, stage_off(tile), v, bfirst2, 0u, false);
      if (valid) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (rh.src >= 0) v[0][i] = uis_mean_update(old[i], v[0][i], rh.nprev);
          if (f4 + i >= m.D) v[0][i] = 0.0f;
        }
        rs_buf_store_f32x4(rs_mean, (uint32_t)(((rh.utt * S + rh.dst) * DP + f4) * 4), v[0]);
      }
    }
    if (xcd_barrier(st, cluster, 32u * ++bar, s_ctl)) return;
  }
}

// trace[-N:] of the best hypothesis (uisrnn.py:561) by walking the back-pointers.
// One wave per utterance.  The walk best hypothesis -> parent -> ... is a chain of N dependent
// loads (0.15 us each: 157 us for 1000 steps with one thread per utterance).  Here lane l owns
// the l-th segment of the chain: (1) it follows ALL B possible entry ranks through its segment
// -- B independent chains, so the loads overlap -- and records where each one leaves, (2) the
// wave stitches the 64 maps (entry of segment l = exit of segment l-1), (3) every lane walks its
// segment once more from its true entry rank and writes the labels.
// The body of k_backtrace for utterance u, by wave 0 of the calling workgroup (every thread of
// the workgroup calls it: there is one workgroup barrier inside).  bt_map: 64 * B bytes of LDS.
__device__ __forceinline__ void backtrace_body(const DecodeState& st, int u, int32_t* __restrict__ labels,
                                               float* __restrict__ scores, float* __restrict__ beam_scores,
                                               unsigned char* bt_map) {
  const int lane = threadIdx.x & 63;
  const bool w0 = threadIdx.x < 64;
  // streaming: the frames received so far (test_iteration 1); labels go where the caller packs them
  const long N = st.avail ? (long)st.avail[u] : (long)(st.off[u + 1] - st.off[u]);
  const long T = st.avail ? N : (long)st.tau * N;
  const int par = (int)(T & 1);  // parity holding the final beam
  const int nb = N > 0 ? st.beam_n[(size_t)par * st.U + u] : 0;
  const size_t e = ((size_t)par * st.U + u) * st.B;
  if (w0) {
    if (beam_scores)
      for (int b = lane; b < st.B; b += 64) beam_scores[(size_t)u * st.B + b] = b < nb ? st.beam_score[e + b] : INFINITY;
    if (scores && lane == 0) scores[u] = nb > 0 ? st.beam_score[e] : (N > 0 ? INFINITY : 0.0f);
  }
  int32_t* out = labels + (st.lab_off ? st.lab_off[u] : st.off[u]);
  if (w0 && N > 0 && nb == 0) for (long i = lane; i < N; i += 64) out[i] = -1;
  const bool walk = N > 0 && nb > 0;
  const uint32_t* bp = st.bp + (size_t)st.tau * st.off[u] * st.B;
  const int B = st.B;
  // segment l: steps hi(l) .. lo(l) walked downwards, hi(0) = T - 1, the last lo = T - N
  const long seg = (N + 63) / 64;
  const long hi = T - 1 - (long)lane * seg;
  long lo = hi - seg + 1;
  if (lo < T - N) lo = T - N;
  const bool work = w0 && walk && hi >= T - N;
  unsigned char* mine = bt_map + (size_t)lane * B;
  if (work) {
    for (int r0 = 0; r0 < B; r0 += 8) {  // 8 entry ranks at a time: 8 loads in flight
      int r[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) r[k] = r0 + k < B ? r0 + k : 0;
      for (long s2 = hi; s2 >= lo; --s2) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          // ranks >= that step's beam width were never written (stale words): clamp, so the walk
          // stays inside this utterance's records; such entry ranks are never stitched in
          const int pr = (int)(bp[(size_t)s2 * B + r[k]] >> 16);
          r[k] = pr < B ? pr : B - 1;
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (r0 + k < B) mine[r0 + k] = (unsigned char)r[k];
    }
  }
  __syncthreads();
  // stitch: the entry rank of segment l (ranks < 256: select kernels cap the beam at 256)
  int entry = 0;
  if (w0 && walk) {
    int cur = 0;  // the best hypothesis of the final beam
    for (int l = 0; l < 64; ++l) {
      if (lane == l) entry = cur;
      const long hl = T - 1 - (long)l * seg;
      if (hl < T - N) break;
      cur = bt_map[(size_t)l * B + cur];
    }
  }
  if (work) {
    int r = entry;
    for (long s2 = hi; s2 >= lo; --s2) {
      const uint32_t v = bp[(size_t)s2 * B + r];
      out[s2 - (T - N)] = (int32_t)(v & 0xffffu);
      r = (int)(v >> 16);
    }
  }
}

__global__ __launch_bounds__(64) void k_backtrace(DecodeState st, int32_t* __restrict__ labels, float* __restrict__ scores,
                                                  float* __restrict__ beam_scores) {
  extern __shared__ unsigned char bt_map[];  // [64][B] exit rank of entry rank r through segment l
  if ((int)blockIdx.x >= st.U) return;
  backtrace_body(st, blockIdx.x, labels, scores, beam_scores, bt_map);
}
