// uis_select_rs.hip -- the one-launch decode with the select REPLICATED (k_decode_rs).
// (Included by uis_kernels.hip: uses its dense-stage helpers.)
//
// Reference behaviour: _calculate_score + the prune + the bookkeeping half of
// _update_beam_state, uisrnn/uisrnn.py:388-453,455-477,534-559 -- the same decisions, keys, leader
// and arithmetic rules as select_fast_body; results are bit-identical (tests/test_gpu_parity.py
// runs this path, the owner-select path and the launch-per-step path against the oracle).
//
// Why.  All utterances advance in lock-step and a step needs the step before it: with one
// workgroup per utterance running the select (k_decode_resident) a step crosses the XCD four times
// -- select -> GRU -> linear_mean1 -> linear_mean2 -> select -- and the first crossing alone (row
// descriptors published, barrier, descriptors staged) costs ~2.4 us on top of a 6 us select that
// 24 of the 32 CUs sit out.  Here EVERY workgroup of an XCD keeps the beam tables of all of the
// cluster's (at most 8) utterances in LDS and wave w decides utterance w, so the step's row list
// exists in every workgroup without being exchanged: three hand-offs per step, no row reservation,
// no descriptor staging.  What makes that affordable is a select that fits one wave:
//   * its only heavy input -- the weighted MSE of the frame against every live cluster mean --
//     arrives as one float per cluster for the clusters the previous step did not rewrite (the
//     cluster's workgroups compute them one step ahead, round robin, between two stages: mse_tab,
//     double buffered by step parity) and as sixteen tile sums for the at most beam_size clusters
//     the previous step DID rewrite (emitted by the linear_mean2 epilogue that wrote the new
//     mean: mse_part); everything is requested in one round trip (rs_front_loads);
//   * the prior terms come from LDS (log tables), the candidate grid -- position b * Kcur + c, no
//     prefix sums, no candidate table -- is prepared a step ahead (rs_prep);
//   * the prune keeps what scores at or below the worst "stay" candidate (a wave-wide maximum by
//     DPP), compacts the survivors in grid order and lets every survivor count the ones that beat
//     it against broadcast LDS reads of the list; ties go to the lowest grid position = the
//     lowest (hypothesis, cluster), the order of the 64-bit keys of select_fast_body; row
//     de-duplication is an LDS minimum per source slot -- no cross-lane loops anywhere;
//   * everything nobody waits for -- the next step's tables, live masks, back-pointers (rs_back),
//     the early MSEs, the next candidate grid -- runs between a stage's publishing its output and
//     the wave's first look at its producers' phase words;
//   * the stages hand over through per-producer phase words, not barriers (rs_flag_publish /
//     rs_flag_wait): a consumer wave polls the four workgroups that produce its K-slice, the
//     step's last hand-off all 32.
#pragma once

// 1: k_decode_rs is the default wherever it applies (UIS_FLAG_OWNER_SELECT keeps k_decode_resident);
// 0: it runs only with UIS_FLAG_REPLICATED_SELECT
#ifndef UIS_RS_DEFAULT
#define UIS_RS_DEFAULT 1
#endif
// 1: two utterances per wave (9 .. 16 utterances per XCD) is the default where it applies; 0: only with
// UIS_FLAG_REPLICATED_SELECT (k_decode_resident keeps those batches)
#ifndef UIS_RS_UPW2_DEFAULT
#define UIS_RS_UPW2_DEFAULT 0   // measured (profiles/r04_rs_shape_classes.txt): 1.20 / 1.40 / 1.60 M frames/s at 65 / 96 / 128 utterances against 1.37 / 1.67 / 1.94 M
#endif
// ... and the same switch for the wide class (beam_size 17 .. 32, observation dim 512)
#ifndef UIS_RS_WIDE_DEFAULT
#define UIS_RS_WIDE_DEFAULT 0   // measured: configs[4] 0.80 M against k_decode_resident's 0.84 M
#endif
#define UIS_RS_UTT 8        // waves per workgroup = utterance slots per cluster and UPW (utterances per wave)
#define UIS_RS_MAXS 256     // slots per utterance (four 64-bit masks)
#define UIS_RS_LOGTAB 128   // entries of the LDS copies of the log tables (larger counts: global)
#define UIS_RS_NOKEY 0xffffffffu
// NPOS = candidate-grid positions per lane (template parameter of the select): 3 -> at most 192
// candidates (beam_size * (max_clusters + 1)) and beam_size <= 16; 4 -> 256 and beam_size <= 32
__host__ __device__ constexpr int rs_max_beam(int npos) { return npos <= 3 ? 16 : 32; }
__host__ __device__ constexpr int rs_max_grid(int npos) { return 64 * npos; }

// beam_size, max_clusters, slots per utterance: run-time values, or compile-time constants in the
// instantiations built for one shape (every RsLds offset then folds into the instruction stream)
struct RsDims { int B, Kmax, S, D; };  // D = observation_dim (unpadded)

struct RsLds {
  // per utterance, persistent: two table sets (by step parity) + frames per slot + masks
  int off_hyp;                               // {K, last, sum(block_counts), score bits} int32 x 4 [B]: one 16-byte read per hypothesis
  int off_ent;                               // uint32 [B][Kmax]: slot | block count << 16
  int off_hdr;                               // int32 [4]    {hypotheses, grid stride, its magic, 0}
  int set_stride;
  int off_pcnt;                              // uint16 [S]   frames assigned to the cluster state in slot s
  int off_flag;                              // uint8 [S]    scratch of rs_back: 1 = referenced by the next beam, 3 = and written by this step
  int off_live;                              // u64 [4]      slots referenced by the CURRENT beam
  int off_new;                               // u64 [4]      of those, written by the previous step
  int off_newlist;                           // int32 [1 + B] count, slots written by the previous step
  int off_stats;                             // u64 [4]      decode statistics (the owner flushes them at the end)
  int persist_stride;
  // per utterance slot, scratch (lives in the split-K area: the select's front part and the dense stages
  // never overlap; rs_back, which runs inside the GRU stage, only touches the persistent blocks)
  int sc_mse;                                // float [S]
  int sc_dst;                                // int32 [B]    the ord-th free slot
  int sc_ckey, sc_ce, sc_csc;                // [64] each: the prune's short list (key, grid position, score)
  int sc_list;                               // uint8 [S]: the live slots, compacted (a select that computes every MSE itself)
  int sc_lead;                               // uint32 [S + 1]: lowest winner rank per source slot (+ 1; 0 = fresh cluster); all ones outside rs_front
  int scratch_stride;
};

__host__ __device__ inline RsLds rs_lds_layout(int B, int Kmax, int S) {
  RsLds l;
  int o = 0;
  auto take = [&](int bytes) { int r = o; o += (bytes + 15) & ~15; return r; };
  l.off_hyp = take(B * 16);
  l.off_ent = take(B * Kmax * 4);
  l.off_hdr = take(16);
  l.set_stride = o;
  o += l.set_stride;
  l.off_pcnt = take(S * 2);
  l.off_flag = take(S);
  l.off_live = take(4 * 8);
  l.off_new = take(4 * 8);
  l.off_newlist = take((1 + B) * 4);
  l.off_stats = take(4 * 8);
  l.persist_stride = o;
  o = 0;
  l.sc_mse = take(S * 4);
  l.sc_dst = take(B * 4);
  l.sc_ckey = take(64 * 4);
  l.sc_ce = take(64 * 4);
  l.sc_csc = take(64 * 4);
  l.sc_list = take(S);
  l.sc_lead = take((S + 1) * 4);
  l.scratch_stride = o;
  return l;
}

// the single-wave select applies (NPOS grid positions per lane)
__host__ __device__ inline bool rs_select_ok(int B, int Kmax, int S, long max_steps, int npos = 3) {
  return B <= rs_max_beam(npos) && B * (Kmax + 1) <= rs_max_grid(npos) && S <= UIS_RS_MAXS && max_steps < 65535;
}

// floats per row of mse_part: the 16-feature tiles' partial sums, then the squared first difference
__host__ __device__ constexpr int rs_part_stride(int Dp) { return Dp <= 256 ? 32 : 48; }
__host__ __device__ constexpr int rs_part_first(int Dp) { return Dp <= 256 ? 16 : 32; }

// row-tile descriptors built locally: enough tiles for every utterance slot's beam_size rows
__host__ __device__ inline int rs_head_tiles(int B, int upw) { return (UIS_RS_UTT * upw * B + 15) / 16; }

// bytes of the split-K partial tiles: all three GRU gates of RC row tiles from eight waves at once,
// or -- SPLIT2, the instantiations whose tables need the room -- two gates, then the third
__host__ __device__ inline size_t rs_spart_bytes(bool split2) { return (size_t)UIS_KSPLIT * UIS_RES_RC * (split2 ? 2 : 3) * 256 * 4; }

// LDS of k_decode_rs: 1 / (2 sigma^2) | log tables | the utterance slots' persistent blocks | split-K
// partial tiles (the slots' select scratch lives in the same bytes) | control words | the rank's
// linear_mean1 / linear_mean2 weight tiles | this step's row descriptors (built locally) | frames
__host__ __device__ inline size_t resident_rs_lds_bytes(int Hp, int Dp, int B, int Kmax, int S, int upw = 1, bool split2 = false) {
  const RsLds L = rs_lds_layout(B, Kmax, S);
  const int slots = UIS_RS_UTT * upw;
  const size_t spart = rs_spart_bytes(split2);
  const size_t scratch = (size_t)slots * L.scratch_stride;
  return (size_t)Dp * 4 + (size_t)2 * UIS_RS_LOGTAB * 8 + (size_t)slots * L.persist_stride +
         (spart > scratch ? spart : scratch) + 128 + (size_t)2 * (Hp / 16) * 64 * 16 + (size_t)rs_head_tiles(B, upw) * 16 * 16 +
         (size_t)2 * slots * 8;
}

// ceil(2^20 / d) for 1 <= d < 4096 without the integer-division sequence: the float quotient of
// two exactly representable numbers is off by at most one, which two multiplications settle
__device__ __forceinline__ unsigned rs_magic20(unsigned d) {
  asm volatile("" : "+v"(d));  // (a wave-uniform d sends the float ops to the scalar unit, which gfx950 does not have: "illegal instruction")
  const unsigned n = (1u << 20) + d - 1u;
  unsigned q = (unsigned)((float)n * __builtin_amdgcn_rcpf((float)d));
  if (q * d > n) --q;
  if ((q + 1u) * d <= n) ++q;
  return q;
}

// set bits of a wave mask below this lane (v_mbcnt: two instructions)
__device__ __forceinline__ int rs_below(unsigned long long mask) {
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// minimum of a row of 16 lanes in its lane 15 (DPP row shifts; lanes shifted in from outside the
// row contribute the identity)
__device__ __forceinline__ uint32_t rs_row_min_u32(uint32_t v) {
  uint32_t o;
  o = (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffff, (int)v, 0x111, 0xf, 0xf, false); v = o < v ? o : v;  // row_shr:1
  o = (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffff, (int)v, 0x112, 0xf, 0xf, false); v = o < v ? o : v;  // row_shr:2
  o = (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffff, (int)v, 0x114, 0xf, 0xf, false); v = o < v ? o : v;  // row_shr:4
  o = (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffff, (int)v, 0x118, 0xf, 0xf, false); v = o < v ? o : v;  // row_shr:8
  return v;
}
__device__ __forceinline__ uint32_t rs_wave_min_u32(uint32_t v) {  // wave-uniform result
  v = rs_row_min_u32(v);
  const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 15), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 31);
  const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 47), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
  const uint32_t ab = a < b ? a : b, cd = c < d ? c : d;
  return ab < cd ? ab : cd;
}
__device__ __forceinline__ int rs_wave_max_i32(int v) {  // small non-negative values
  return (int)~rs_wave_min_u32(~(uint32_t)v);
}

// Weighted MSE of the frame (xv: this lane's float4 chunks) against the mean whose chunks are in mv, by
// the 16 lanes of a quarter wave in the canonical tree (include/uis_numerics.h): lane p holds
// d = 256 c + 4 * (p + 16 k), k = 0..3, of 256-float block c (index 4 c + k).  Lane p == 0 returns the
// value.  DP <= 512: one or two blocks.
template <int DP>
__device__ __forceinline__ float rs_mse16_regs(int D, const f32x4 (&mv)[4 * ((DP + 255) / 256)],
                                               const f32x4 (&xv)[4 * ((DP + 255) / 256)], const float* swgt, int p) {
  constexpr int NB = (DP + 255) / 256;
  float A[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    f32x4 wv[4], mc[4], xc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int d = 256 * c + 4 * (p + 16 * k);
      wv[k] = d < DP ? *reinterpret_cast<const f32x4*>(swgt + d) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      mc[k] = mv[4 * c + k];
      xc[k] = xv[4 * c + k];
    }
    mse16_block(mc, xc, wv, A);  // (chunks past DP are zero-filled)
  }
  const float d0 = mv[0][0] - xv[0][0];  // meaningful on lane p == 0 (d = 0): the only lane whose value is stored
  return uis_mse_finish(mse16_total(A), d0 * d0, D);
}
template <int DP>
__device__ __forceinline__ void rs_load_mean16(__amdgpu_buffer_rsrc_t rs_mean, size_t slot_index, int p,
                                               f32x4 (&mv)[4 * ((DP + 255) / 256)]) {
#pragma unroll
  for (int k = 0; k < 4 * ((DP + 255) / 256); ++k) {
    const int d = 256 * (k >> 2) + 4 * (p + 16 * (k & 3));
    mv[k] = d < DP ? load_sc1(rs_mean, (uint32_t)((slot_index * DP + d) * 4)) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  }
}
template <int DP>
__device__ __forceinline__ void rs_load_frame16(const float* xrow, int p, f32x4 (&xv)[4 * ((DP + 255) / 256)]) {
#pragma unroll
  for (int k = 0; k < 4 * ((DP + 255) / 256); ++k) {
    const int d = 256 * (k >> 2) + 4 * (p + 16 * (k & 3));
    xv[k] = d < DP ? *reinterpret_cast<const f32x4*>(xrow + d) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  }
}

// What the front part of a wave's select leaves in registers: lane r = winner r (four packed
// words per lane, so that they can stay live across the GRU stage's MFMA chain).
struct RsWin {
  int keep, C, nlead;  // wave-uniform: winners, candidates, rnn rows of this utterance
  unsigned a;          // wb | wc << 8 | Kb << 20
  unsigned b;          // (src + 1) & 0xffff | (dst & 0xffff) << 16      (src = -1: fresh cluster)
  unsigned c;          // nprev | ord << 16 | is_lead << 24
  float score;
  __device__ __forceinline__ int wb() const { return (int)(a & 0xffu); }
  __device__ __forceinline__ int wc() const { return (int)((a >> 8) & 0xfffu); }
  __device__ __forceinline__ int Kb() const { return (int)(a >> 20); }
  __device__ __forceinline__ int src() const { return (int)(b & 0xffffu) - 1; }
  __device__ __forceinline__ int dst() const { return (int)(b >> 16); }
  __device__ __forceinline__ int nprev() const { return (int)(c & 0xffffu); }
  __device__ __forceinline__ int ord() const { return (int)((c >> 16) & 0xffu); }
  __device__ __forceinline__ bool is_lead() const { return ((c >> 24) & 1u) != 0u; }
};

__device__ __forceinline__ void rs_lds_fence() { asm volatile("" ::: "memory"); }  // lanes of ONE wave talk through LDS: program order is enough

// PREP: everything about a step's candidates that needs nothing but the utterance's tables -- the
// candidate grid with each candidate's slot, prior and hypothesis score, the live / new slot masks,
// the first beam_size free slots.  Runs one step ahead, inside the previous step's last barrier.
// The grid: position e = b * Kcur + c (hypothesis b, cluster c <= K_b), NPOS positions per lane.
template <int NPOS>
struct RsPrep {
  int nb, nch, C, nn;                 // wave-uniform
  int Kcur, kmagic;
  unsigned long long old0, old1, old2, old3;  // live slots the previous step left alone (their MSEs are published); FULL: every live slot
  int cslot[NPOS];                    // >= 0: slot whose MSE the candidate takes; -1: fresh cluster; -2: no candidate
  int stay;                           // bit k: the candidate at position lane + 64 k keeps its hypothesis' last cluster
  double pr[NPOS];
  float bs[NPOS];
};

template <bool FULL = false, int NPOS = 3, typename Mid>
__device__ __forceinline__ RsPrep<NPOS> rs_prep(const DevModel& m, const DecodeState& st, const RsLds& L, const RsDims dm, int step,
                                                const unsigned char* pers, unsigned char* scr, const double* s_lblk,
                                                const double* s_lden, Mid mid) {
  int lane_ = threadIdx.x & 63;
  asm volatile("" : "+v"(lane_));
  const int lane = lane_;
  const int B = dm.B, Kmax = dm.Kmax, S = dm.S;
  const int par = step & 1;
  const unsigned char* const set_cur = pers + par * L.set_stride;
  const u32x4* shyp = reinterpret_cast<const u32x4*>(set_cur + L.off_hyp);
  const uint32_t* sent = reinterpret_cast<const uint32_t*>(set_cur + L.off_ent);
  const int* shdr = reinterpret_cast<const int*>(set_cur + L.off_hdr);
  const unsigned long long* slive = reinterpret_cast<const unsigned long long*>(pers + L.off_live);
  const unsigned long long* snew = reinterpret_cast<const unsigned long long*>(pers + L.off_new);
  const int* snewlist = reinterpret_cast<const int*>(pers + L.off_newlist);
  int* sdst = reinterpret_cast<int*>(scr + L.sc_dst);
  RsPrep<NPOS> P;
  P.nb = shdr[0]; P.Kcur = shdr[1]; P.kmagic = shdr[2];
  P.nn = snewlist[0];
  const int nb = P.nb, Kcur = P.Kcur, kmagic = P.kmagic;
  unsigned long long lv[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) lv[k] = 64 * k < S ? slive[k] : 0ull;
  P.old0 = FULL ? lv[0] : lv[0] & ~snew[0];
  P.old1 = 64 < S ? (FULL ? lv[1] : lv[1] & ~snew[1]) : 0ull;
  P.old2 = 128 < S ? (FULL ? lv[2] : lv[2] & ~snew[2]) : 0ull;
  P.old3 = 192 < S ? (FULL ? lv[3] : lv[3] & ~snew[3]) : 0ull;
  // the first beam_size free slots (not referenced by the current beam), in slot order: slot-lane
  // l + 64 k knows its own rank among the free ones
  {
    int before = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (64 * k < S && before < B) {
        unsigned long long fm = ~lv[k];
        if (S - 64 * k < 64) fm &= (1ull << (S - 64 * k)) - 1ull;
        const int rk = before + rs_below(fm);
        if (((fm >> lane) & 1ull) && rk < B) sdst[rk] = lane + 64 * k;
        before += __popcll(fm);
      }
    }
  }
  {
    uint32_t* slead = reinterpret_cast<uint32_t*>(scr + L.sc_lead);
    for (int i = lane; i <= S; i += 64) slead[i] = 0xffffffffu;
  }
  P.nch = (nb * Kcur + 63) >> 6;
  P.stay = 0;
#pragma unroll
  for (int k = 0; k < NPOS; ++k) { P.cslot[k] = -2; P.pr[k] = 0.0; P.bs[k] = 0.0f; }
  auto prep_at = [&](int e, int k, int& cslot, double& prior, float& base) {
    const int b = (int)(((unsigned)e * (unsigned)kmagic) >> 20), c = e - b * Kcur;
    if (b < nb) {
      const u32x4 h = shyp[b];  // {K, last, sum, score}
      const int Kb = (int)h[0];
      if (c <= Kb) {
        const int sum = (int)h[2];
        double ld = s_lden[sum < UIS_RS_LOGTAB ? sum : 0];  // (no pointer select between LDS and global: two loads)
        if (sum >= UIS_RS_LOGTAB) ld = st.logden[sum];
        base = __builtin_bit_cast(float, (uint32_t)h[3]);
        if (c < Kb) {
          const uint32_t en = sent[b * Kmax + c];
          cslot = (int)(en & 0xffffu);
          if (c == (int)h[1]) { prior = m.lp_stay; P.stay |= 1 << k; }
          else {
            const int blk = (int)(en >> 16);
            double lb = s_lblk[blk < UIS_RS_LOGTAB ? blk : 0];
            if (blk >= UIS_RS_LOGTAB) lb = st.logblk[blk];
            prior = (m.lp_sw + lb) - ld;
          }
        } else {
          cslot = -1;
          prior = m.lp_new - ld;
        }
      }
    }
  };
  mid();  // (the caller's early load)
  P.C = 0;
#pragma unroll
  for (int k = 0; k < NPOS; ++k) {
    // (three positions: chunks past the grid are skipped by a wave-uniform branch -- a beam of 10 rarely has
    // more than 64 candidates; four positions: no branch, so that the chunks' LDS round trips overlap)
    if (NPOS > 3 || k == 0 || P.nch > k) prep_at(lane + 64 * k, k, P.cslot[k], P.pr[k], P.bs[k]);
    P.C += __popcll(__ballot(P.cslot[k] != -2));
  }
  return P;
}

// The weighted MSE of frame row `frame` against the cluster means in the slots of `mask` (utterance u), by one
// wave: the slots compacted into a list, 16 lanes per mean, eight means in flight per pass; one float per slot
// into the wave's scratch (sc_mse).  The FULL select's first part; a caller may run it EARLY for the slots
// the step in flight does not rewrite (their means are final) and leave the rewritten ones to rs_front.
template <int DP>
__device__ __forceinline__ void rs_full_mse(const DecodeState& st, const RsLds& L, const RsDims dm, int u, long frame,
                                            unsigned char* scr, const unsigned long long (&mask)[4], const float* swgt_full) {
  int lane_ = threadIdx.x & 63;
  asm volatile("" : "+v"(lane_));
  const int lane = lane_;
  const int S = dm.S;
  float* smse = reinterpret_cast<float*>(scr + L.sc_mse);
  unsigned char* s_list = scr + L.sc_list;
  int n = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (64 * k < S) {
      const bool on = (mask[k] >> lane) & 1ull;
      if (on) s_list[n + rs_below(mask[k])] = (unsigned char)(lane + 64 * k);
      n += __popcll(mask[k]);
    }
  }
  rs_lds_fence();
  if (n > 0) {
    constexpr int NV = 4 * ((DP + 255) / 256);
    const __amdgpu_buffer_rsrc_t rs_mean =
        __builtin_amdgcn_make_buffer_rsrc((void*)st.pool_mean, (short)0, 0x7fffffff, 0x00020000);
    const int grp = lane >> 4, p = lane & 15;
    f32x4 xv[NV];
    rs_load_frame16<DP>(st.x + (size_t)frame * DP, p, xv);
    for (int i0 = 0; i0 < n; i0 += 8) {
      f32x4 mv[2][NV];
      int sl[2];
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const int i = i0 + 4 * h2 + grp;
        sl[h2] = (int)s_list[i < n ? i : 0];
        rs_load_mean16<DP>(rs_mean, (size_t)u * S + sl[h2], p, mv[h2]);
      }
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const float v = rs_mse16_regs<DP>(dm.D, mv[h2], xv, swgt_full, p);
        if (p == 0 && i0 + 4 * h2 + grp < n) smse[sl[h2]] = v;
      }
    }
  }
  rs_lds_fence();
}

// FRONT: the MSEs, scores, prune, winners, rows -- what the step's dense stages wait for.  One wave
// (all 64 lanes), utterance u, decode step `step` whose frame is row `frame` of the stream.  pers =
// the utterance's persistent block, scr = its scratch (rs_prep left the free slots there).
// rs_part + part_off = where the partial sums of the rows this utterance emitted in the previous step
// start (the i-th slot of its new-slot list was written by its i-th row).
// FULL: the wave computes the MSE of the frame against EVERY live cluster mean itself (no published
// values, no partial sums: k_decode_big, where a wave owns its utterance alone); P.old* then lists all
// live slots and `swgt_full` is 1 / (2 sigma^2) in LDS.
template <int DP, bool FULL = false, int NPOS = 3>
__device__ __forceinline__ RsWin rs_front(const DevModel& m, const DecodeState& st, const RsLds& L, const RsDims dm, int u, int step,
                                          long frame, unsigned char* pers, unsigned char* scr, __amdgpu_buffer_rsrc_t rs_part,
                                          uint32_t part_off, const RsPrep<NPOS>& P, unsigned long long* ph,
                                          const float* swgt_full = nullptr) {
  // (opaque to the optimiser: nothing lane-derived is hoisted out of the kernel's step loop, where
  // it would have to stay live -- spilled -- across the dense stages)
  int lane_ = threadIdx.x & 63;
  asm volatile("" : "+v"(lane_));
  const int lane = lane_;
#if defined(UIS_RESIDENT_TIMING)
  unsigned long long ph_prev = wall_clock64();
#define PSTAMP(k) do { if (ph) { const unsigned long long n_ = wall_clock64(); ph[k] += n_ - ph_prev; ph_prev = n_; } } while (0)
#else
#define PSTAMP(k) do {} while (0)
#endif
  const int B = dm.B, Kmax = dm.Kmax, S = dm.S, U = st.U;
  const int par = step & 1;
  const unsigned char* const set_cur = pers + par * L.set_stride;
  const u32x4* shyp = reinterpret_cast<const u32x4*>(set_cur + L.off_hyp);
  const uint32_t* sent = reinterpret_cast<const uint32_t*>(set_cur + L.off_ent);
  const unsigned short* spcnt = reinterpret_cast<const unsigned short*>(pers + L.off_pcnt);
  const int* snewlist = reinterpret_cast<const int*>(pers + L.off_newlist);
  float* smse = reinterpret_cast<float*>(scr + L.sc_mse);
  const int* sdst = reinterpret_cast<const int*>(scr + L.sc_dst);

  const int Kcur = P.Kcur, kmagic = P.kmagic, nch = P.nch, C = P.C, nn = P.nn;
  const unsigned long long old[4] = {P.old0, P.old1, P.old2, P.old3};
  const float mse_new = st.mse0[frame];
  if (FULL) {
    // ---- every live cluster's MSE from its mean (P.old*: the slots whose MSE is still to be computed -- all live
    // ones, unless the caller ran rs_full_mse on some of them earlier)
    rs_full_mse<DP>(st, L, dm, u, frame, scr, old, swgt_full);
  } else {
  // ---- ONE round trip: the fresh-cluster MSE, the published MSEs of the clusters the previous
  // step left alone, and for the ones it rewrote the tile sums its linear_mean2 epilogue emitted
  // (one float per 16-feature tile + the squared first difference per cluster: lane i takes new cluster i)
  constexpr int PSTR = rs_part_stride(DP), PV = DP <= 256 ? 4 : 8;
  float vold[4];
  {
    const float* tab = st.mse_tab + ((size_t)par * U + u) * S;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      vold[k] = 0.0f;
      if ((old[k] >> lane) & 1ull) vold[k] = load_f32_sc1(tab + lane + 64 * k);
    }
  }
  f32x4 pv[PV];
  float pfirst = 0.0f;
  int nsl = 0;
  if (lane < nn) {
    nsl = snewlist[1 + lane];
#pragma unroll
    for (int k = 0; k < PV; ++k) pv[k] = load_sc1(rs_part, part_off + (uint32_t)(lane * PSTR * 4 + 16 * k));
    pfirst = rs_buf_load_f32_sc1(rs_part, part_off + (uint32_t)((lane * PSTR + rs_part_first(DP)) * 4));
  }
  PSTAMP(0);
  // ---- the MSEs: the published values first (they arrive first), then the rewritten clusters
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if ((old[k] >> lane) & 1ull) smse[lane + 64 * k] = vold[k];
  if (lane < nn) {
    float A[16];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // uis_numerics.h: accumulator p = tile p (+ tile p + 16, left to right)
        A[4 * k + e] = PV == 4 ? pv[k][e] : pv[k][e] + pv[(k + 4) & (PV - 1)][e];
      }
    }
    smse[nsl] = uis_mse_finish(uis_mse_acc_sum(A), pfirst, dm.D);
  }
  }
  rs_lds_fence();
  PSTAMP(1);

  // ---- candidate scores
  uint32_t key[NPOS];
  float sc[NPOS];
#pragma unroll
  for (int k = 0; k < NPOS; ++k) {
    key[k] = UIS_RS_NOKEY;
    sc[k] = 0.0f;
    if ((NPOS > 3 || k == 0 || nch > k) && P.cslot[k] != -2) {
      const float mse = P.cslot[k] >= 0 ? smse[P.cslot[k]] : mse_new;
      sc[k] = P.bs[k] + uis_step_loss(mse, P.pr[k]);
      if (uis_isfinite(sc[k])) key[k] = uis_score_key(sc[k]);
    }
  }
  if (st.dbg_scores) {  // UIS_FLAG_DEBUG_SCORES: the step's _calculate_score arrays
#pragma unroll
    for (int k = 0; k < NPOS; ++k) {
      if ((k == 0 || nch > k) && P.cslot[k] != -2) {
        const int e = lane + 64 * k;
        const int b = (int)(((unsigned)e * (unsigned)kmagic) >> 20), c = e - b * Kcur;
        st.dbg_scores[(((size_t)step * U + u) * B + b) * (Kmax + 1) + c] = sc[k];
      }
    }
  }
  int nfin = 0;
#pragma unroll
  for (int k = 0; k < NPOS; ++k) nfin += __popcll(__ballot(key[k] != UIS_RS_NOKEY));
  const int keep = nfin < B ? nfin : B;
  PSTAMP(2);

  // ---- prune: the `keep` best, ascending, ties to the lowest grid position.
  // Short list first: every hypothesis has one candidate that keeps its last cluster; with a full
  // beam there are beam_size of those, so the beam_size best candidates all score at or below the
  // worst of them -- in practice the short list is those plus a handful of switches.  It is
  // compacted (in grid order) and every entry counts the entries that beat it.
  int win_e = 0;
  float win_sc = 0.0f;
  uint32_t thr = UIS_RS_NOKEY;
  {
    int nstay = 0;
    uint32_t wk = 0u;
#pragma unroll
    for (int k = 0; k < NPOS; ++k) {
      const bool sk = ((P.stay >> k) & 1) != 0;
      nstay += __popcll(__ballot(sk));
      wk = sk && key[k] > wk ? key[k] : wk;
    }
    if (nstay >= B) thr = ~rs_wave_min_u32(~wk);  // the worst of them (a non-finite one: no threshold)
  }
  bool v[NPOS];
  unsigned long long vm[NPOS];
  int nbefore[NPOS + 1];
  nbefore[0] = 0;
#pragma unroll
  for (int k = 0; k < NPOS; ++k) {
    v[k] = key[k] != UIS_RS_NOKEY && key[k] <= thr;
    vm[k] = __ballot(v[k]);
    nbefore[k + 1] = nbefore[k] + __popcll(vm[k]);
  }
  const int nsv = nbefore[NPOS];
#if defined(UIS_RS_COUNT_PATHS)  // diagnostic: how long the short lists are (workgroup 0's copies; uis_decoder.hip prints them)
  if (!FULL && blockIdx.x == 0 && lane == 0) atomicAdd(&st.counters[88 + (nsv <= 16 ? 0 : nsv <= 32 ? 1 : nsv <= 64 ? 2 : 3)], 1ull);
#endif
  if (nsv <= 64) {
    uint32_t* sck = reinterpret_cast<uint32_t*>(scr + L.sc_ckey);
    int* sce = reinterpret_cast<int*>(scr + L.sc_ce);
    float* scs = reinterpret_cast<float*>(scr + L.sc_csc);
#pragma unroll
    for (int k = 0; k < NPOS; ++k)
      if (v[k]) { const int q = nbefore[k] + rs_below(vm[k]); sck[q] = key[k]; sce[q] = lane + 64 * k; scs[q] = sc[k]; }
    if (lane >= nsv) sck[lane] = UIS_RS_NOKEY;  // (beats nobody)
    rs_lds_fence();
    const bool mine = lane < nsv;
    const uint32_t ck = mine ? sck[lane] : UIS_RS_NOKEY;
    const int ce = sce[lane];
    const float cs = scs[lane];
    // every entry counts the entries that beat it: the list comes back sixteen keys at a time (the
    // same address in every lane: a broadcast); compaction kept the grid order, so an EARLIER entry
    // also wins a tie -- k_j <= ck, written k_j < ck + 1 (keys of finite scores are below all ones)
    const uint32_t ck1 = ck + 1u;
    const u32x4* sck4 = reinterpret_cast<const u32x4*>(sck);
    int rank = 0;
    auto count16 = [&](int j0) {
      u32x4 kk[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) kk[i] = sck4[(j0 >> 2) + i];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) rank += kk[i][e] < ((j0 + 4 * i + e) < lane ? ck1 : ck) ? 1 : 0;
    };
    count16(0);
    if (nsv > 16) count16(16);
    if (nsv > 32) { count16(32); if (nsv > 48) count16(48); }
    // entry -> lane `rank` of the winners: a forward permute (the others send to lane 63, which is
    // never a winner: keep <= 32)
    const int to = (mine && rank < keep) ? rank : 63;
    win_e = __builtin_amdgcn_ds_permute(to << 2, ce);
    win_sc = __builtin_bit_cast(float, __builtin_amdgcn_ds_permute(to << 2, __builtin_bit_cast(int, cs)));
  } else {
    // (more than 64 survivors -- a model that switches freely, a beam not yet full: `keep` rounds of
    // the wave-wide minimum; among equal keys the lowest grid position: chunk k before k + 1, lowest lane first)
    for (int r = 0; r < keep; ++r) {
      uint32_t loc = key[0];
#pragma unroll
      for (int k = 1; k < NPOS; ++k) loc = loc < key[k] ? loc : key[k];
      const uint32_t mn = rs_wave_min_u32(loc);
      int e = 0;
      float sv = 0.0f;
      bool found = false;
#pragma unroll
      for (int k = 0; k < NPOS; ++k) {
        const unsigned long long mk = __ballot(key[k] == mn);
        if (!found && mk) {
          const int l = __ffsll((long long)mk) - 1;
          e = l + 64 * k;
          sv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sc[k]), l));
          if (lane == l) key[k] = UIS_RS_NOKEY;
          found = true;
        }
      }
      if (lane == r) { win_e = e; win_sc = sv; }
    }
  }
  PSTAMP(3);

  // ---- winners: lane r = winner r
  RsWin out;
  out.keep = keep;
  out.C = C;
  const bool nodedup = (st.flags & 1u) != 0;
  const int r = lane;
  const bool isw = r < keep;
  int wb = 0, wc = 0, src = -2, Kb = 0;
  if (isw) {
    wb = (int)(((unsigned)win_e * (unsigned)kmagic) >> 20);
    wc = win_e - wb * Kcur;
    Kb = (int)reinterpret_cast<const uint32_t*>(shyp + wb)[0];
    const uint32_t en = sent[wb * Kmax + (wc < Kmax ? wc : Kmax - 1)];  // (both reads in flight together)
    src = wc < Kb ? (int)(en & 0xffffu) : -1;
  }
  int lead = r;
  if (!nodedup) {  // lowest rank with the same source wins: a minimum per source slot in LDS
    uint32_t* slead = reinterpret_cast<uint32_t*>(scr + L.sc_lead);
    if (isw) __hip_atomic_fetch_min(slead + (src + 1), (uint32_t)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    rs_lds_fence();
    if (isw) lead = (int)slead[src + 1];
  }
  const bool is_lead = isw && lead == r;
  const unsigned long long lmask = __ballot(is_lead);
  const int nlead = __popcll(lmask);
  const int ord = rs_below(lmask);
  // (the ord-th free slot: rs_prep listed them)
  int dst = 0xffff, nprev = 0;
  if (is_lead) {
    dst = sdst[ord];
    nprev = src >= 0 ? (int)spcnt[src] : 0;
  }
  {
    const int dl = __shfl(dst, lead, 64);
    if (isw && !is_lead) dst = dl;
  }
  PSTAMP(4);
  out.nlead = nlead;
  out.a = (unsigned)wb | ((unsigned)wc << 8) | ((unsigned)Kb << 20);
  out.b = ((unsigned)(src + 1) & 0xffffu) | (((unsigned)dst & 0xffffu) << 16);
  out.c = (unsigned)nprev | ((unsigned)ord << 16) | (is_lead ? 1u << 24 : 0u);
  out.score = win_sc;
  return out;
}

// BACK: the next step's tables, masks and counts, the back-pointers -- nothing anybody waits for.
// `owner`: this workgroup writes what outlives the step to memory.
template <int NPOS = 3, typename Mid>
__device__ __forceinline__ void rs_back(const DevModel& m, const DecodeState& st, const RsLds& L, const RsDims dm, int u, int step,
                                        long off0, unsigned char* pers, bool owner, const RsWin& w, Mid mid) {
  int lane_ = threadIdx.x & 63;
  asm volatile("" : "+v"(lane_));
  const int lane = lane_;
  const int B = dm.B, Kmax = dm.Kmax, S = dm.S, U = st.U;
  const int par = step & 1, nxt = par ^ 1;
  const unsigned char* const set_cur = pers + par * L.set_stride;
  unsigned char* const set_nxt = pers + nxt * L.set_stride;
  const u32x4* shyp = reinterpret_cast<const u32x4*>(set_cur + L.off_hyp);
  const uint32_t* sent = reinterpret_cast<const uint32_t*>(set_cur + L.off_ent);
  unsigned char* sflag = pers + L.off_flag;
  u32x4* nhyp = reinterpret_cast<u32x4*>(set_nxt + L.off_hyp);
  uint32_t* nent = reinterpret_cast<uint32_t*>(set_nxt + L.off_ent);
  int* nhdr = reinterpret_cast<int*>(set_nxt + L.off_hdr);
  unsigned short* spcnt = reinterpret_cast<unsigned short*>(pers + L.off_pcnt);
  unsigned long long* slive = reinterpret_cast<unsigned long long*>(pers + L.off_live);
  unsigned long long* snew = reinterpret_cast<unsigned long long*>(pers + L.off_new);
  int* snewlist = reinterpret_cast<int*>(pers + L.off_newlist);

  const int r = lane;
  const bool isw = r < w.keep;
  int Knew_w = 0;
  unsigned info_b = 0u;  // per winner, for the table copy below: changed entry's slot | block count << 16
  if (isw) {
    const int wb = w.wb(), wc = w.wc(), Kb = w.Kb();
    const bool is_new = wc == Kb;
    const u32x4 h = shyp[wb];  // {K, last, sum, score}
    const int lastb = (int)h[1];
    Knew_w = Kb + (is_new ? 1 : 0);
    const int blk_new = is_new ? 1 : (int)(sent[wb * Kmax + (is_new ? 0 : wc)] >> 16) + (wc != lastb ? 1 : 0);
    if (Knew_w > Kmax) { Knew_w = Kmax; if (owner) st.overflow[u] = 1; }
    const int sum_new = (int)h[2] + ((is_new || wc != lastb) ? 1 : 0);
    nhyp[r] = u32x4{(uint32_t)Knew_w, (uint32_t)wc, (uint32_t)sum_new, __builtin_bit_cast(uint32_t, w.score)};
    info_b = ((unsigned)w.dst() & 0xffffu) | ((unsigned)blk_new << 16);
    if (owner) {
      st.beam_score[((size_t)nxt * U + u) * B + r] = w.score;  // (the final beam's scores are read back by k_backtrace)
      st.bp[((size_t)st.tau * off0 + step) * B + r] = ((unsigned)wb << 16) | (unsigned)wc;
    }
  }
  // which slots the next beam references: byte flags (plain stores; LDS atomics on four mask words
  // would serialise 64 lanes x 8 waves), read back by the slot's own lane below
  for (int i = lane; 4 * i < S; i += 64) reinterpret_cast<uint32_t*>(sflag)[i] = 0u;
  rs_lds_fence();
  const int Kmaxseen = rs_wave_max_i32(Knew_w);
  {
    // every entry of every winner: LPW lanes per winner (four with beam_size <= 16, two up to 32), lane q
    // of them takes the clusters q, q + LPW, ...; the changed entry from the winner's record, the
    // unchanged ones from the parent's row (BeamState(source), uisrnn.py:66-69)
    constexpr int LPW = 64 / rs_max_beam(NPOS);
    const int rr = lane / LPW, q = lane % LPW;
    const unsigned ia = (unsigned)__shfl((int)w.a, rr, 64), ib = (unsigned)__shfl((int)info_b, rr, 64);
    const int Knew = __shfl(Knew_w, rr, 64);
    const int rb = (int)(ia & 0xffu), rc = (int)((ia >> 8) & 0xfffu);
    for (int c2 = q; c2 < Kmaxseen; c2 += LPW) {
      if (rr < w.keep && c2 < Knew) {
        const uint32_t en = c2 == rc ? ib : sent[rb * Kmax + c2];  // slot | block count << 16
        nent[rr * Kmax + c2] = en;
        sflag[en & 0xffffu] = (unsigned char)1;
      }
    }
  }
  rs_lds_fence();
  if (w.is_lead()) {
    const int dst = w.dst();
    spcnt[dst] = (unsigned short)(w.nprev() + 1);
    if (sflag[dst]) sflag[dst] = (unsigned char)3;  // (not referenced: a cluster beyond the cap, the call reports the overflow)
    snewlist[1 + w.ord()] = dst;
  }
  rs_lds_fence();
  mid();  // (the caller's early load: about a round trip of work is left)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (64 * k < S) {
      const int sl = lane + 64 * k;
      const unsigned f = sl < S ? (unsigned)sflag[sl] : 0u;
      const unsigned long long lm = __ballot((f & 1u) != 0u), nm = __ballot((f & 2u) != 0u);
      if (lane == 0) { slive[k] = lm; snew[k] = nm; }
    }
  }
  if (lane == 0) {
    snewlist[0] = w.nlead;
    const int Kc = Kmaxseen + 1;  // grid stride of the next step: clusters 0 .. K of every hypothesis
    nhdr[0] = w.keep;
    nhdr[1] = Kc;
    nhdr[2] = (int)rs_magic20((unsigned)Kc);  // e / Kc = (e * magic) >> 20, exact for e < 2048
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(pers + L.off_stats);
    acc[0] += (unsigned long long)w.nlead;
    acc[1] += (unsigned long long)w.keep;
    acc[2] += (unsigned long long)w.C;
    if ((unsigned long long)Kmaxseen > acc[3]) acc[3] = (unsigned long long)Kmaxseen;
    if (owner) st.beam_n[(size_t)nxt * U + u] = w.keep;
  }
}

// The early MSEs, one step ahead: for this wave's utterance u, every slot the NEXT beam references
// that this step did not rewrite, against the next step's frame (row `frame_next`); one float per
// slot into mse_tab[(step + 1) parity].  The 32 workgroups of the cluster share the slots of an
// utterance round robin (every workgroup holds the same masks), so each wave computes at most a
// handful, in one round trip, between the arrival at the barrier behind the GRU stage and the wait
// (it needs nobody else's data of this step: those means were final a step ago).
template <int DP, typename Mid>
__device__ __forceinline__ void rs_early_mse(const DevModel& m, const DecodeState& st, const RsLds& L, const RsDims dm, int u, int step,
                                             long frame_next, const unsigned char* pers, const float* swgt,
                                             int rank, int w, Mid mid) {
  int lane_ = threadIdx.x & 63;
  asm volatile("" : "+v"(lane_));
  const int lane = lane_;
  const int S = dm.S, U = st.U;
  const unsigned long long* slive = reinterpret_cast<const unsigned long long*>(pers + L.off_live);
  const unsigned long long* snew = reinterpret_cast<const unsigned long long*>(pers + L.off_new);
  // (the list lives in rs_back's byte-flag area, free between two steps' table updates -- NOT in the
  // slot's scratch: that aliases the split-K tiles, which a faster wave of this workgroup may
  // already be writing for the next stage; slots are < 256)
  unsigned char* s_list = const_cast<unsigned char*>(pers) + L.off_flag;
  int n = 0, before = 0;
  const int mine_mod = (rank + 4 * w) & 31;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (64 * k < S) {
      const unsigned long long mask = slive[k] & ~snew[k];
      const int idx = before + rs_below(mask);
      const bool mine = ((mask >> lane) & 1ull) && (idx & 31) == mine_mod;
      const unsigned long long mm = __ballot(mine);
      if (mine) s_list[n + rs_below(mm)] = (unsigned char)(lane + 64 * k);
      n += __popcll(mm);
      before += __popcll(mask);
    }
  }
  mid();  // (the caller's early load: queued ahead of the means)
  if (n == 0) return;
  rs_lds_fence();
  constexpr int NV = 4 * ((DP + 255) / 256);
  const __amdgpu_buffer_rsrc_t rs_mean =
      __builtin_amdgcn_make_buffer_rsrc((void*)st.pool_mean, (short)0, 0x7fffffff, 0x00020000);
  const int grp = lane >> 4, p = lane & 15;
  f32x4 xv[NV];
  rs_load_frame16<DP>(st.x + (size_t)frame_next * DP, p, xv);
  float* tab = st.mse_tab + ((size_t)((step + 1) & 1) * U + u) * S;
  for (int i0 = 0; i0 < n; i0 += 4) {
    const int i = i0 + grp;
    const int sl = (int)s_list[i < n ? i : 0];
    f32x4 mv[NV];
    rs_load_mean16<DP>(rs_mean, (size_t)u * S + sl, p, mv);
    const float v = rs_mse16_regs<DP>(dm.D, mv, xv, swgt, p);
    if (p == 0 && i < n) tab[sl] = v;
  }
}

// resident_tile_nv without the workgroup barrier that ends the pass: the caller places it (and may
// put work that needs no other wave's partial tiles in front of it).
// SPLIT2 (NG == 3): the partial tiles of gates 0 and 1 first ([wave][RC][2][256]); barrier; `mid`
// -- the caller combines those two; barrier; gate 2 into the same bytes ([wave][RC][1][256]): two
// thirds of the LDS for two more workgroup barriers per pass.
template <int NG, int PER, int RC, int NV, int KBS, bool SPLIT2, typename After, typename Mid>
__device__ __forceinline__ void rs_tile_nv(const f32x4 (&wr)[NG][PER], const float* __restrict__ bias, int gate_stride,
                                           __amdgpu_buffer_rsrc_t rsrc, const uint32_t (&boff)[RC], float* spart,
                                           After after_issue, Mid mid) {
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
  if constexpr (!SPLIT2) {
#pragma unroll
    for (int r = 0; r < NV; ++r)
#pragma unroll
      for (int g = 0; g < NG; ++g)
        *reinterpret_cast<f32x4*>(spart + ((size_t)((w * RC + r) * NG + g) * 256) + (lane & 15) * 16 + 4 * q) = acc[r][g];
  } else {
    static_assert(!SPLIT2 || NG == 3, "the GRU's three gates");
#pragma unroll
    for (int r = 0; r < NV; ++r)
#pragma unroll
      for (int g = 0; g < 2; ++g)
        *reinterpret_cast<f32x4*>(spart + ((size_t)((w * RC + r) * 2 + g) * 256) + (lane & 15) * 16 + 4 * q) = acc[r][g];
    __syncthreads();
    mid();
    __syncthreads();
#pragma unroll
    for (int r = 0; r < NV; ++r)
      *reinterpret_cast<f32x4*>(spart + ((size_t)(w * RC + r) * 256) + (lane & 15) * 16 + 4 * q) = acc[r][NG - 1];
  }
}
template <int NG, int PER, int RC, int KBS, bool SPLIT2, typename After, typename Mid>
__device__ __forceinline__ void rs_tile(const f32x4 (&wr)[NG][PER], const float* __restrict__ bias, int gate_stride,
                                        __amdgpu_buffer_rsrc_t rsrc, const uint32_t (&boff)[RC], int nvalid, float* spart,
                                        After after_issue, Mid mid) {
  static_assert(RC == 3, "dispatch below");
  if (nvalid >= 3) rs_tile_nv<NG, PER, RC, 3, KBS, SPLIT2>(wr, bias, gate_stride, rsrc, boff, spart, after_issue, mid);
  else if (nvalid == 2) rs_tile_nv<NG, PER, RC, 2, KBS, SPLIT2>(wr, bias, gate_stride, rsrc, boff, spart, after_issue, mid);
  else rs_tile_nv<NG, PER, RC, 1, KBS, SPLIT2>(wr, bias, gate_stride, rsrc, boff, spart, after_issue, mid);
}

// The wait half of the in-launch barrier for a workgroup that has ARRIVED already (xcd_arrive: its
// stores were drained there) and did work of its own since: no second drain -- what that work
// stored is nobody's input before the next barrier, whose arrival drains it -- and no workgroup
// barrier in front of the poll.
__device__ __forceinline__ bool rs_xcd_wait(const DecodeState& st, int cluster, uint32_t target, int* s_abort) {
  if (threadIdx.x == 0) {
    uint32_t* ctr = st.rx_bar + cluster * 32;
    s_abort[2] = 0;
    unsigned spins = 0;
    int bad = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
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

// The one-launch decode with the replicated select (see the top of this file).  Same grid, same
// weight residency, same dense stages and arithmetic as k_decode_resident; three in-launch
// hand-offs per step instead of four barriers, no row reservation, no descriptor staging.
// Template parameters beyond the model's padded sizes (the shape classes):
//   NPOS    candidate-grid positions per lane: 3 (beam_size <= 16, <= 192 candidates) or 4 (<= 32, <= 256)
//   UPW     utterances per wave: 1 (at most 8 utterances per XCD) or 2 (16: wave w decides slots w and w + 8,
//           one after the other)
//   CB, CK  beam_size and max_clusters as compile-time constants (0: run-time values) -- the instantiation
//           of a shape whose LDS layout then folds into the instruction stream
//   SPLIT2  the GRU's split-K partial tiles in two rounds (rs_tile_nv): 24 KB of LDS for the larger tables
#if defined(UIS_RS_LONG_SCALARS)  // (A/B: rounds 3-4 kept the per-wave frame numbers in 64 bits)
typedef long rs_idx_t;
#else
typedef int rs_idx_t;
#endif
template <int HP, int DP, int NPOS = 3, int UPW = 1, int CB = 0, int CK = 0, bool SPLIT2 = false>
__global__ __launch_bounds__(512) void k_decode_rs(DevModel m, DecodeState st) {
  constexpr int NKB = HP / 16, PER = NKB / UIS_KSPLIT, RC = UIS_RES_RC;
  constexpr int NFT1 = HP / 16, SH1 = 32 / NFT1;
  constexpr int NFT2 = DP / 16, SH2 = 32 / NFT2;
  constexpr int EPT = (RC + 1) / 2;
  constexpr int SLOTS = UIS_RS_UTT * UPW;
  constexpr int PSTR = rs_part_stride(DP);
  static_assert(NFT1 * SH1 == 32 && NFT2 * SH2 == 32 && PER * UIS_KSPLIT == NKB && DP <= 512, "shapes");
  static_assert(UPW == 1 || UPW == 2, "utterances per wave");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);  // the wave's number, known to be uniform (scalar addresses)
  const int ncl = st.ncl;
  const int cluster = blockIdx.x % ncl, rank = blockIdx.x / ncl;
  const int U = st.U;
  // (the fixed-shape classes are dispatched for unpadded models only: observation_dim = DP, rnn_hidden_size = HP)
  const RsDims dm{CB ? CB : st.B, CB ? CK : st.Kmax, CB ? CB * CK + CB : st.S, CB ? DP : m.D};
  const int Hreal = CB ? HP : m.H;
  const int S = dm.S, B = dm.B;
  const RsLds L = rs_lds_layout(dm.B, dm.Kmax, dm.S);
  float* swgt = reinterpret_cast<float*>(smem_raw);
  double* s_lblk = reinterpret_cast<double*>(smem_raw + (size_t)DP * 4);
  double* s_lden = s_lblk + UIS_RS_LOGTAB;
  unsigned char* s_pers = reinterpret_cast<unsigned char*>(s_lden + UIS_RS_LOGTAB);
  float* spart = reinterpret_cast<float*>(s_pers + (size_t)SLOTS * L.persist_stride);
  unsigned char* s_scr = reinterpret_cast<unsigned char*>(spart);
  const size_t spart_bytes = rs_spart_bytes(SPLIT2) > (size_t)SLOTS * L.scratch_stride ? rs_spart_bytes(SPLIT2)
                                                                                        : (size_t)SLOTS * L.scratch_stride;
  int* s_ctl = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(spart) + spart_bytes);  // [0] abort [1] steps [2] arrived [8 .. 8 + SLOTS) rows per slot
  f32x4* s_w1 = reinterpret_cast<f32x4*>(s_ctl + 32);
  f32x4* s_w2 = s_w1 + NKB * 64;
  const int head_tiles = rs_head_tiles(B, UPW);
  u32x4* s_head = reinterpret_cast<u32x4*>(s_w2 + NKB * 64);
  long* s_wframe = reinterpret_cast<long*>(s_head + head_tiles * 16);  // [SLOTS] this step's frame of every slot's utterance
  long* s_wnext = s_wframe + SLOTS;                                    // [SLOTS] ... and the next step's

  uint32_t xcc = 0;
  if (t == 0) {
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xfu;
    if (rank == 0) __hip_atomic_store(st.cl_xcc + cluster, xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int k = 0; k < 32; ++k) s_ctl[k] = 0;
  }
  for (int i = t; i < DP; i += 512) swgt[i] = m.wgt[i];
  for (int i = t; i < UIS_RS_LOGTAB; i += 512) { s_lblk[i] = st.logblk[i]; s_lden[i] = st.logden[i]; }
  for (int i = t; i < head_tiles * 16; i += 512) s_head[i] = u32x4{0u, 0u, 0u, 0u};
  if (t < SLOTS) { s_wframe[t] = 0; s_wnext[t] = 0; }
  // ---- this wave's utterances: slots w (and w + 8) of the cluster
  int u_w[UPW];
  bool has_u[UPW];
  unsigned char* pers_w[UPW];
  unsigned char* scr_w[UPW];
  // (32-bit: the one-launch path decodes fewer than 2^31 frames -- the host checks -- and fewer than 65535 steps)
  rs_idx_t off0_w[UPW], N_w[UPW], T_w[UPW], fpos_w[UPW];
  int prev_base[UPW];  // first row of the slot's utterance in the previous step's row list
#pragma unroll
  for (int q = 0; q < UPW; ++q) {
    const int slot = w + UIS_RS_UTT * q;
    u_w[q] = cluster + ncl * slot;
    has_u[q] = u_w[q] < U;
    pers_w[q] = s_pers + (size_t)slot * L.persist_stride;
    scr_w[q] = s_scr + (size_t)slot * L.scratch_stride;
    off0_w[q] = 0; N_w[q] = 0; fpos_w[q] = 0; prev_base[q] = 0;
    if (has_u[q]) { off0_w[q] = (rs_idx_t)st.off[u_w[q]]; N_w[q] = (rs_idx_t)st.off[u_w[q] + 1] - off0_w[q]; }
    T_w[q] = st.tau * N_w[q];
    // beam_set = [BeamState()] (uisrnn.py:528): one empty hypothesis, nothing live
    for (int i = lane; i < L.persist_stride / 4; i += 64) reinterpret_cast<int*>(pers_w[q])[i] = 0;
  }
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < UPW; ++q) {
      int* hdr = reinterpret_cast<int*>(pers_w[q] + L.off_hdr);
      hdr[0] = 1; hdr[1] = 1; hdr[2] = 1 << 20;  // one hypothesis, grid stride 1
      reinterpret_cast<int*>(pers_w[q] + L.off_hyp)[1] = -1;  // {K 0, last -1, sum 0, score 0}
    }
  }
  {
    int myT = 0;
#pragma unroll
    for (int q = 0; q < UPW; ++q)
      if (has_u[q] && lane == 0 && (int)T_w[q] > myT) myT = (int)T_w[q];
    if (myT > 0) atomicMax(&s_ctl[1], myT);
  }
  __syncthreads();
  const int nsteps = s_ctl[1];
  // (round 5) this launch runs steps [step0, s_end) of the decode: a launch that starts late picks up what the
  // previous one left in st.resume -- per utterance its persistent block, then the rows its last step emitted
  const int step0 = st.step0;
  const int s_end = (st.step1 > 0 && st.step1 < nsteps) ? st.step1 : nsteps;
  if (step0 > 0) {
    for (int k = 0; k < SLOTS; ++k) {
      const int u = cluster + ncl * k;
      if (u >= U) break;
      const int* src = reinterpret_cast<const int*>(st.resume + (size_t)u * L.persist_stride);
      int* dst = reinterpret_cast<int*>(s_pers + (size_t)k * L.persist_stride);
      for (int i = t; i < L.persist_stride / 4; i += 512) dst[i] = src[i];
      if (t == 0) s_ctl[8 + k] = reinterpret_cast<const int*>(st.resume + (size_t)U * L.persist_stride)[u];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < UPW; ++q) {
      fpos_w[q] = N_w[q] > 0 ? step0 % N_w[q] : 0;
      int base = 0;
      for (int k = 0; k < w + UIS_RS_UTT * q; ++k) base += s_ctl[8 + k];
      prev_base[q] = base;
    }
  }

  f32x4 wg[3][PER];
  const int ft1 = rank / SH1, tpar1 = rank % SH1;
  const int ft2 = rank / SH2, tpar2 = rank % SH2;
#pragma unroll
  for (int kb = 0; kb < PER; ++kb) {
#pragma unroll
    for (int g = 0; g < 3; ++g)
      wg[g][kb] = reinterpret_cast<const f32x4*>(m.whh[0])[((size_t)(g * NFT1 + ft1) * NKB + w * PER + kb) * 64 + lane];
    s_w1[(w * PER + kb) * 64 + lane] = reinterpret_cast<const f32x4*>(m.w1)[((size_t)ft1 * NKB + w * PER + kb) * 64 + lane];
    s_w2[(w * PER + kb) * 64 + lane] = reinterpret_cast<const f32x4*>(m.w2)[((size_t)ft2 * NKB + w * PER + kb) * 64 + lane];
  }
  const __amdgpu_buffer_rsrc_t rs_hid =
      __builtin_amdgcn_make_buffer_rsrc((void*)st.pool_hid, (short)0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a1 =
      __builtin_amdgcn_make_buffer_rsrc((void*)st.a1, (short)0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_hst =
      __builtin_amdgcn_make_buffer_rsrc((void*)st.gi_up, (short)0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_mean =
      __builtin_amdgcn_make_buffer_rsrc((void*)st.pool_mean, (short)0, 0x7fffffff, 0x00020000);
  // per-producer phase words of this cluster (one 128-byte line of the control block)
  uint32_t* const flags_c = st.rx_flags + cluster * 32;
  const __amdgpu_buffer_rsrc_t rs_flags =
      __builtin_amdgcn_make_buffer_rsrc((void*)flags_c, (short)0, 128, 0x00020000);
  // UIS_FLAG_TEST_STALL: one workgroup publishes phases below 8 only (it goes silent after two steps)
  const uint32_t live_mask = ((st.flags & 0x4000u) != 0u && cluster == 0 && rank == 5) ? 7u : 0xffffffffu;
  const int tile0 = (cluster * st.rx_stride) >> 4;
  const uint32_t h1_off = (uint32_t)((size_t)U * S * HP * 4);
  // this cluster's rows of partial sums
  const __amdgpu_buffer_rsrc_t rs_part = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(st.mse_part + (size_t)cluster * st.rx_stride * PSTR), (short)0, 0x7fffffff, 0x00020000);
  __syncthreads();
#if defined(UIS_RESIDENT_TIMING)
  unsigned long long rt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long rt_prev = wall_clock64();
  unsigned long long ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long ft_acc[4] = {0, 0, 0, 0}, rt_prev2 = rt_prev;
#endif

  RsPrep<NPOS> prep[UPW];  // (later steps: prepared inside the previous step's last hand-off)
#pragma unroll
  for (int q = 0; q < UPW; ++q) prep[q] = rs_prep<false, NPOS>(m, st, L, dm, step0, pers_w[q], scr_w[q], s_lblk, s_lden, []() {});
  for (int s = step0; s < s_end; ++s) {
    // ---- select, replicated: wave w decides utterance slot w (then w + 8); every workgroup gets the same rows
    RsWin win[UPW];
    bool act_w[UPW];
#pragma unroll
    for (int q = 0; q < UPW; ++q) {
      win[q].keep = 0; win[q].C = 0; win[q].nlead = 0; win[q].a = 0u; win[q].b = 0u; win[q].c = 0u; win[q].score = 0.0f;
      act_w[q] = has_u[q] && s < T_w[q];
      const long frame_w = (long)(off0_w[q] + fpos_w[q]);
      if (act_w[q]) {
#if defined(UIS_RESIDENT_TIMING)
        win[q] = rs_front<DP, false, NPOS>(m, st, L, dm, u_w[q], s, frame_w, pers_w[q], scr_w[q], rs_part, (uint32_t)(prev_base[q] * PSTR * 4),
                                           prep[q], (blockIdx.x == 0 && w == 0 && q == 0) ? ph_acc : nullptr);
#else
        win[q] = rs_front<DP, false, NPOS>(m, st, L, dm, u_w[q], s, frame_w, pers_w[q], scr_w[q], rs_part, (uint32_t)(prev_base[q] * PSTR * 4),
                                           prep[q], nullptr);
#endif
      }
      if (lane == 0) {
        const int slot = w + UIS_RS_UTT * q;
        s_ctl[8 + slot] = win[q].nlead;
        s_wframe[slot] = frame_w;
        s_wnext[slot] = (long)(off0_w[q] + (fpos_w[q] + 1 == N_w[q] ? 0 : fpos_w[q] + 1));  // (after the last step: some frame of the utterance, unused)
      }
    }
    RSTAMP(0);
    __syncthreads();
    if (s_ctl[0]) return;  // a hand-off of the previous step gave up (cl_abort tells the host): all waves leave here
    int nrows = 0;
    {
      int base[UPW];
#pragma unroll
      for (int q = 0; q < UPW; ++q) base[q] = 0;
#pragma unroll
      for (int k = 0; k < SLOTS; ++k) {
        const int c = s_ctl[8 + k];
#pragma unroll
        for (int q = 0; q < UPW; ++q) if (k < w + UIS_RS_UTT * q) base[q] += c;
        nrows += c;
      }
#pragma unroll
      for (int q = 0; q < UPW; ++q) {
        if (win[q].is_lead()) {
          // (the slot's number rides in the top bits of the frame count: the row's frame is s_wframe[that])
          s_head[base[q] + win[q].ord()] = u32x4{(unsigned)u_w[q], (unsigned)win[q].src(), (unsigned)win[q].dst(),
                                                 (unsigned)win[q].nprev() | ((unsigned)(w + UIS_RS_UTT * q) << 16)};
        }
        prev_base[q] = base[q];
      }
    }
    __syncthreads();
    const int nrt = (nrows + 15) >> 4;
    RSTAMP(1);

    // ---- GRU: h' = gru(gi0[frame], W_hh h_src + b_hh) -> dst slot
    {
      const int my1 = nrt > tpar1 ? (nrt - tpar1 + SH1 - 1) / SH1 : 0;
      for (int i0 = 0; i0 < my1; i0 += RC) {
        uint32_t boff[RC];
#pragma unroll
        for (int r = 0; r < RC; ++r) {
          const int tile = tpar1 + SH1 * (i0 + r < my1 ? i0 + r : i0);
          const RowHead rh = lds_row_head(s_head, 16 * tile + (t & 15));
          boff[r] = rh.src >= 0 ? (uint32_t)((((size_t)rh.utt * S + rh.src) * HP) * 4) : h1_off;
        }
        const int j = ft1 * 16 + (t & 15);
        RowHead re[EPT];
        float gir[EPT], giz[EPT], gin[EPT], hprev[EPT];
        float ghr[EPT], ghz[EPT];
        bool ework[EPT];
        auto epilogue_operands = [&]() {
#pragma unroll
          for (int k = 0; k < EPT; ++k) {
            const int r = (t >> 8) + 2 * k;
            const int lrow = 16 * (tpar1 + SH1 * (i0 + r)) + ((t & 255) >> 4);
            ework[k] = r < RC && i0 + r < my1 && lrow < nrows;
            // (no branch around the loads: behind divergent control flow the compiler waits for
            // EVERYTHING in flight -- these gi0 rows, first touched here, come from HBM -- before
            // the stage's first MFMA; a thread without a row fetches row 0's operands instead)
            re[k] = lds_row_head(s_head, ework[k] ? lrow : 0);
            const long frame = s_wframe[((unsigned)re[k].nprev >> 16) & (unsigned)(SLOTS - 1)];
            re[k].nprev &= 0xffff;
            const float* gi = st.gi0 + (size_t)frame * (3 * HP);  // (m.G)
            gir[k] = gi[j]; giz[k] = gi[HP + j]; gin[k] = gi[2 * HP + j];
            hprev[k] = rs_buf_load_f32_sc1(rs_hid, (uint32_t)(((re[k].src >= 0 ? re[k].utt * S + re[k].src : U * S) * HP + j) * 4));
          }
        };
        auto combine_rz = [&]() {  // SPLIT2: gates r and z while gate n's partial tiles wait in registers
#pragma unroll
          for (int k = 0; k < EPT; ++k) {
            if (!ework[k]) continue;
            const int r = (t >> 8) + 2 * k, e = t & 255;
            ghr[k] = splitk_combine<RC, 2>(spart, r, 0, e);
            ghz[k] = splitk_combine<RC, 2>(spart, r, 1, e);
          }
        };
        FSTAMP(0);
        rs_tile<3, PER, RC, 64, SPLIT2>(wg, m.bhh[0] + ft1 * 16, HP, rs_hid, boff, my1 - i0 < RC ? my1 - i0 : RC, spart,
                                        epilogue_operands, combine_rz);
        __syncthreads();
        FSTAMP(1);
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
          if (!ework[k]) continue;
          const int r = (t >> 8) + 2 * k, e = t & 255;
          float ghn;
          if constexpr (SPLIT2) {
            ghn = splitk_combine<RC, 1>(spart, r, 0, e);
          } else {
            ghr[k] = splitk_combine<RC, 3>(spart, r, 0, e);
            ghz[k] = splitk_combine<RC, 3>(spart, r, 1, e);
            ghn = splitk_combine<RC, 3>(spart, r, 2, e);
          }
          const float out = j < Hreal ? uis_gru_unit(gir[k], giz[k], gin[k], ghr[k], ghz[k], ghn, hprev[k]) : 0.0f;
          rs_buf_store_f32(rs_hid, (uint32_t)(((re[k].utt * S + re[k].dst) * HP + j) * 4), out);
          rs_buf_store_f32(rs_hst, (uint32_t)(((tile0 + tpar1 + SH1 * (i0 + r)) * NFT1 + ft1) * 256 + e) * 4u, out);
        }
        FSTAMP(2);
        __syncthreads();
        FSTAMP(3);
      }
    }
    RSTAMP(2);
    // publish; then the select's back part (the next step's tables and masks: nothing anybody waits
    // for); the first look at the producers' words is requested from inside it, so that its round
    // trip is over when the wave gets there; then wait
    {
      rs_flag_publish(flags_c, rank, (3u * (uint32_t)s + 1u) & live_mask, (st.flags & 0x20000u) != 0u);
      u32x4 pk = u32x4{0u, 0u, 0u, 0u};
      auto peek = [&]() { pk = rs_flag_peek4(rs_flags, (uint32_t)(16 * w)); };
      bool peeked = false;
#pragma unroll
      for (int q = 0; q < UPW; ++q) {
        if (act_w[q]) {
          // (the owner of utterance slot r is rank r: it alone writes that utterance's lasting outputs)
          if (q == UPW - 1 || !act_w[UPW - 1]) {
            rs_back<NPOS>(m, st, L, dm, u_w[q], s, off0_w[q], pers_w[q], rank == w + UIS_RS_UTT * q, win[q], peek);
            peeked = true;
          } else {
            rs_back<NPOS>(m, st, L, dm, u_w[q], s, off0_w[q], pers_w[q], rank == w + UIS_RS_UTT * q, win[q], []() {});
          }
          fpos_w[q] = fpos_w[q] + 1 == N_w[q] ? 0 : fpos_w[q] + 1;
        }
      }
      if (!peeked) peek();
      if (nrt > tpar1 && !rs_flag_ready4(pk, 3u * (uint32_t)s + 1u) &&
          rs_flag_wait(st, rs_flags, (uint32_t)(16 * w), 3u * (uint32_t)s + 1u))
        s_ctl[0] = 1;
    }
    if (s == step0 && t == 0 && rank == 1 && (st.flags & 0x100u)) xcc ^= 1u;  // UIS_FLAG_TEST_MISPLACED: pretend
    if (s == step0 && t == 0 && __hip_atomic_load(st.cl_xcc + cluster, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != xcc)
      __hip_atomic_store(st.cl_abort, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // not on one XCD
    RSTAMP(3);

    // ---- linear_mean1 + relu -> a1
    {
      const int my1h = nrt > tpar1 ? (nrt - tpar1 + SH1 - 1) / SH1 : 0;
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
            rs_buf_store_f32(rs_a1, (uint32_t)(((tile0 + tile) * NFT1 + ft1) * 256 + (e & 255)) * 4u, v > 0.0f ? v : 0.0f);
          }
        }
        __syncthreads();
      }
    }
    RSTAMP(4);
    // publish; then the next step's MSEs of the clusters this step did not rewrite (every workgroup
    // its share of every utterance's; visible to all behind the step's last hand-off); then wait
    {
      rs_flag_publish(flags_c, rank, (3u * (uint32_t)s + 2u) & live_mask, (st.flags & 0x20000u) != 0u);
      u32x4 pk = u32x4{0u, 0u, 0u, 0u};
      auto peek = [&]() { pk = rs_flag_peek4(rs_flags, (uint32_t)(16 * w)); };
      bool peeked = false;
#pragma unroll
      for (int q = 0; q < UPW; ++q) {
        if (has_u[q] && s + 1 < T_w[q]) {
          if (!peeked) {
            rs_early_mse<DP>(m, st, L, dm, u_w[q], s, (long)(off0_w[q] + fpos_w[q]), pers_w[q], swgt, rank, w, peek);
            peeked = true;
          } else {
            rs_early_mse<DP>(m, st, L, dm, u_w[q], s, (long)(off0_w[q] + fpos_w[q]), pers_w[q], swgt, rank, w, []() {});
          }
        }
      }
      if (!peeked) peek();
      if (nrt > tpar2 && !rs_flag_ready4(pk, 3u * (uint32_t)s + 2u) &&
          rs_flag_wait(st, rs_flags, (uint32_t)(16 * w), 3u * (uint32_t)s + 2u))
        s_ctl[0] = 1;
    }
    RSTAMP(5);

    // ---- linear_mean2 + running mean -> dst slot
    {
      const int my_tiles = nrt > tpar2 ? (nrt - tpar2 + SH2 - 1) / SH2 : 0;
      for (int i0 = 0; i0 < my_tiles; i0 += RC) {
        uint32_t boff[RC];
#pragma unroll
        for (int r = 0; r < RC; ++r) {
          const int tile = tpar2 + SH2 * (i0 + r < my_tiles ? i0 + r : i0);
          boff[r] = (uint32_t)((((tile0 + tile) * NFT1) * 256 + (t & 15) * 16) * 4);
        }
        const int f = ft2 * 16 + (t & 15);
        RowHead re[EPT];
        float old[EPT], xn[EPT];
        bool ework[EPT];
        auto epilogue_operands = [&]() {
#pragma unroll
          for (int k = 0; k < EPT; ++k) {
            const int r = (t >> 8) + 2 * k;
            const int lrow = 16 * (tpar2 + SH2 * (i0 + r)) + ((t & 255) >> 4);
            ework[k] = r < RC && i0 + r < my_tiles && lrow < nrows;
            re[k] = lds_row_head(s_head, ework[k] ? lrow : 0);  // (no branch around the load: see the GRU stage)
            xn[k] = st.x[(size_t)s_wnext[((unsigned)re[k].nprev >> 16) & (unsigned)(SLOTS - 1)] * DP + f];  // the NEXT frame of the row's utterance
            re[k].nprev &= 0xffff;
            old[k] = rs_buf_load_f32_sc1(rs_mean, (uint32_t)(((re[k].utt * S + (re[k].src >= 0 ? re[k].src : 0)) * DP + f) * 4));
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
          if (f >= dm.D) v = 0.0f;
          rs_buf_store_f32(rs_mean, (uint32_t)(((re[k].utt * S + re[k].dst) * DP + f) * 4), v);
          // this tile's share of the next step's weighted MSE against the mean just written
          // (uis_numerics.h: the row's 16 features of the tile sit in 16 adjacent lanes -- quad sums
          // left to right, then (q0 + q1) + (q2 + q3)); the select adds the tiles' sums
          const float term = uis_mse_term(v, xn[k], swgt[f]);
          float q = ((dpp_perm<0x00>(term) + dpp_perm<0x55>(term)) + dpp_perm<0xAA>(term)) + dpp_perm<0xFF>(term);
          q = q + dpp_perm<0x141>(q);  // row_half_mirror: the neighbouring quad's sum
          q = q + dpp_perm<0x140>(q);  // row_mirror: the other half's
          const int lrow = 16 * (tpar2 + SH2 * (i0 + r)) + ((t & 255) >> 4);
          if ((t & 15) == 0) rs_buf_store_f32(rs_part, (uint32_t)((lrow * PSTR + ft2) * 4), q);
          if (f == 0) { const float d0 = v - xn[k]; rs_buf_store_f32(rs_part, (uint32_t)((lrow * PSTR + rs_part_first(DP)) * 4), d0 * d0); }
        }
        __syncthreads();
      }
    }
    RSTAMP(6);
    // publish; then the next step's candidate grid (this wave's own tables: nobody else's data); then
    // wait -- every wave, every step, for all 32 producers (rs_flag_wait_all)
    {
      rs_flag_publish(flags_c, rank, (3u * (uint32_t)s + 3u) & live_mask, (st.flags & 0x20000u) != 0u);
      uint32_t pk = 0u;
      auto peek = [&]() { pk = rs_flag_peek_all(flags_c); };
      bool peeked = false;
#pragma unroll
      for (int q = 0; q < UPW; ++q) {
        if (has_u[q] && s + 1 < T_w[q]) {
          if (!peeked) {
            prep[q] = rs_prep<false, NPOS>(m, st, L, dm, s + 1, pers_w[q], scr_w[q], s_lblk, s_lden, peek);
            peeked = true;
          } else {
            prep[q] = rs_prep<false, NPOS>(m, st, L, dm, s + 1, pers_w[q], scr_w[q], s_lblk, s_lden, []() {});
          }
        }
      }
      if (!peeked) peek();
      if (!rs_flag_ready_all(pk, 3u * (uint32_t)s + 3u) && rs_flag_wait_all(st, flags_c, 3u * (uint32_t)s + 3u)) s_ctl[0] = 1;
    }
    RSTAMP(7);
  }
#if defined(UIS_RESIDENT_TIMING)
  if (t == 0 && (blockIdx.x == 0 || blockIdx.x == 31 * ncl))
    for (int k = 0; k < 8; ++k) st.counters[(blockIdx.x == 0 ? 48 : 64) + k] = rt_acc[k];
  if (t == 0 && blockIdx.x == 0) for (int k = 0; k < 8; ++k) st.counters[80 + k] = ph_acc[k];
  if (t == 0 && blockIdx.x == 31 * ncl) for (int k = 0; k < 4; ++k) st.counters[72 + k] = ft_acc[k];
#endif
  if (s_end < nsteps) {  // more steps to come in another launch: rank 0's copy of the cluster's tables goes to st.resume
    __syncthreads();
    if (s_ctl[0]) return;
    if (rank == 0) {
      for (int k = 0; k < SLOTS; ++k) {
        const int u = cluster + ncl * k;
        if (u >= U) break;
        const int* src = reinterpret_cast<const int*>(s_pers + (size_t)k * L.persist_stride);
        int* dst = reinterpret_cast<int*>(st.resume + (size_t)u * L.persist_stride);
        for (int i = t; i < L.persist_stride / 4; i += 512) dst[i] = src[i];
        if (t == 0) reinterpret_cast<int*>(st.resume + (size_t)U * L.persist_stride)[u] = s_ctl[8 + k];
      }
    }
    return;
  }
  if (rank < SLOTS && cluster + ncl * rank < U && t == 0) {  // this utterance's statistics, by its owner rank
    const unsigned long long* acc =
        reinterpret_cast<const unsigned long long*>(s_pers + (size_t)rank * L.persist_stride + L.off_stats);
    atomicAdd(&st.counters[0], acc[0]);
    atomicAdd(&st.counters[1], acc[1]);
    atomicAdd(&st.counters[2], acc[2]);
    atomicMax(&st.counters[3], acc[3]);
  }
}
