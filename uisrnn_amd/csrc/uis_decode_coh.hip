// uis_decode_coh.hip -- k_decode_coh: the one-launch decode for MANY utterances per XCD with TWO
// utterance cohorts in flight (round 5).  (Included by uis_kernels.hip: uses k_decode_big's dense-stage
// helpers and the single-wave select of uis_select_rs.hip.)
//
// Reference behaviour: predict_single for every utterance of the list, uisrnn/uisrnn.py:479-562 -- the
// decisions, arithmetic and its order are those of k_decode_big<WS> (bit-identical: tests/test_gpu_parity.py
// runs this kernel, k_decode_big<WS>, the launch-per-step path and the oracle on the same inputs).  What
// makes two cohorts legal is uisrnn.py:588-589: utterances are independent, only an utterance's own steps
// are ordered.
//
// Why.  k_decode_big<WS> walks select | GRU | linear_mean1 | linear_mean2 with an XCD-wide barrier behind
// each: at the configs[3] share (1024 utterances, 128 per XCD) 24 of the step's 122 us were the select
// (two to four waves of eight busy, the MFMA pipes idle) and the four barrier waits (the slowest of 32
// workgroups, a weight-slot refill inside two of them).  Here an XCD's utterances form cohorts A and B
// with their own row lists, hand-off tiles and completion counters, and a workgroup alternates between
// them, so that what one cohort waits for -- every workgroup's contribution to its previous stage --
// completes while the workgroup runs a stage of the OTHER cohort:
//
//   phase    0 GRU A | 1 GRU B | 2 mean1 A | 3 mean1 B | 4 mean2 A | 5 mean2 B     (step s)
//   rides    select B(s)                                             select A(s + 1)
//
//   * every dense phase's dependency (the same cohort's previous stage, in all 32 workgroups) is two
//     phases back; it is awaited per WAVE (an sc1 poll of the cohort's counter in the XCD's L2, cached in
//     LDS for the other waves), not by a workgroup barrier;
//   * the selects ride on a dense phase of the other cohort: the waves that own a cohort-A utterance
//     run their select at the start of phase 5 while the other waves of the workgroup pull mean2-B
//     tiles, and join them afterwards (B likewise on phase 0);
//   * row tiles are PULLED from an LDS counter per phase (not dealt round robin), so a wave that comes
//     late -- out of a select, or behind a long tile -- takes fewer; a workgroup's LAST wave out of a
//     phase signals the cohort's counter (one L2 atomic per workgroup and phase, as before) and, behind
//     mean1-B / mean2-B, refills the 32 KB mean-head weight slot while the other seven waves go on
//     (the refill behind mean2-B hides behind the two GRU phases; the one behind mean1-B is exposed);
//   * no workgroup barrier inside the step loop at all.
//
// MEASURED (round 5, configs[3] share, one MI355X, profiles/r05_coh_*): bit-identical, and SLOWER than
// k_decode_big<WS>: 3.77 M frames/s against 4.03 M (this form), 3.68 M with the selects' old-cluster MSEs computed a
// phase early and the next phase's counter prefetched, 3.82 M as three merged phases with cohort A's tiles before
// cohort B's (tools/experiments/r05_cohorts_three_phases.patch).  The stage clocks say why: a workgroup runs two
// waves per SIMD (256 registers each), and a wave that runs a select is a wave that takes no row tiles -- its
// partner alone does not keep the SIMD's MFMA pipe busy, the select itself takes 14-22 us next to tile waves
// instead of 6.3 alone, and six phase starts per step (counter look, tile pull, descriptor and row round trips,
// store drain) cost more than the four barriers they replace: fixed cost per step 45 us against 18 (257
// utterances: 71 against 47 us per step).  The hand-off waits do disappear (0.4-1.4 us each); what they were
// hiding was not idle MFMA time that another cohort could use, but the wave slots.  Hence OPT-IN
// (UIS_FLAG_COHORTS), kept as a tested experiment; k_decode_big<WS> stays the default.
#pragma once

// LDS: the select part of k_decode_big<WS> | W_hh slice | mean-head slot | control words
#define UIS_COH_CTL_BYTES 512
__host__ __device__ inline size_t coh_lds_bytes(int Hp, int Dp, int B, int Kmax, int S, int nws) {
  return ((big_ws_select_bytes(Dp, B, Kmax, S, nws) + 255) & ~(size_t)255) + (size_t)4 * (Hp / 16) * 64 * 16 + UIS_COH_CTL_BYTES;
}
// control words (int32 view)
#define COH_NSTEPS 1
#define COH_NUTT 2     // + cohort: utterances of the cohort in this cluster
#define COH_TILE 4     // + phase: next row tile of the phase (reset by the phase's last wave)
#define COH_FIN 10     // + phase: waves of this workgroup through with the phase
#define COH_SLOT 16    // generation of the mean-head slot: 2 s + 1 = linear_mean1's slice for step s, 2 s + 2 = linear_mean2's
#define COH_SEEN 17    // + {dense A, dense B, select A, select B}: the largest value of the cluster's counter seen by a wave
#define COH_FREE 21    // generation the slot is FREE for: every wave is through with the previous content
#define COH_PARTS 22   // waves that have written their eighth of the slot

// true = gave up (time-out, or somebody else did)
__device__ __forceinline__ bool coh_wait(const DecodeState& st, const uint32_t* ctr, uint32_t target, int* s_seen) {
  if (target == 0u) return false;
  if ((uint32_t)__hip_atomic_load(s_seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= target) {
    asm volatile("" ::: "memory");
    return false;
  }
  unsigned spins = 0;
  for (;;) {
    // (sc1: served by the XCD's L2, where the arrivals are executed -- see xcd_barrier)
    const uint32_t v = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    asm volatile("" ::: "memory");
    if (v >= target) {
      if ((threadIdx.x & 63) == 0) __hip_atomic_store(s_seen, (int)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      return false;
    }
    __builtin_amdgcn_s_sleep(1);
    if (++spins > (1u << 21)) {  // ~1 s: give up instead of hanging the device
      __hip_atomic_store(st.cl_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return true;
    }
    if ((spins & 255u) == 0 && __hip_atomic_load(st.cl_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return true;
  }
}
// the mean-head slot holds generation `gen` (written by another wave of this workgroup)
__device__ __forceinline__ bool coh_slot_wait(const DecodeState& st, const int* s_slot, int gen) {
  unsigned spins = 0;
  while (__hip_atomic_load(s_slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < gen) {
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 1023u) == 0 && __hip_atomic_load(st.cl_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return true;
    if (spins > (1u << 22)) {
      __hip_atomic_store(st.cl_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return true;
    }
  }
  asm volatile("" ::: "memory");
  return false;
}
// wave-uniform fetch-and-add on an LDS word (lane 0 asks)
__device__ __forceinline__ int coh_lds_add(int* p, int v) {
  int r = 0;
  if ((threadIdx.x & 63) == 0) r = __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return __builtin_amdgcn_readfirstlane(r);
}

template <int HP, int DP, int CB = 0, int CK = 0>
__global__ __launch_bounds__(512) void k_decode_coh(DevModel m, DecodeState st) {
  m.Hp = HP; m.Dp = DP; m.G = 3 * HP;  // (what the template arguments say)
  if (CB) { st.B = CB; st.Kmax = CK; st.S = CB * CK + CB; m.H = HP; m.D = DP; }  // (see k_decode_resident)
  constexpr int NKB = HP / 16;
  constexpr int NFT1 = HP / 16, SH1 = 32 / NFT1;  // ranks sharing one GRU / linear_mean1 feature tile
  constexpr int NFT2 = DP / 16, SH2 = 32 / NFT2;  // ranks sharing one linear_mean2 feature tile
  static_assert(NFT1 * SH1 == 32 && NFT2 * SH2 == 32, "hidden size 128 / 256 / 512, observation_dim 128 / 256 (padded)");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int t = threadIdx.x, lane = t & 63, q = lane >> 4;
  const int wu = __builtin_amdgcn_readfirstlane(t >> 6);
  const int ncl = st.ncl;
  const int cluster = blockIdx.x % ncl, rank = blockIdx.x / ncl;
  const int U = st.U, S = st.S;
  const int nws = (((U + ncl - 1) / ncl) + 31) / 32;  // utterances per rank (wave k owns the rank's k-th utterance)
  const RsLds RL = rs_lds_layout(st.B, st.Kmax, S);
  const size_t select_bytes = big_ws_select_bytes(DP, st.B, st.Kmax, S, nws);
  f32x4* s_whh = reinterpret_cast<f32x4*>(smem_raw + ((select_bytes + 255) & ~(size_t)255));  // [3][NKB][64]
  f32x4* s_wm = s_whh + 3 * NKB * 64;                                                          // [NKB][64] linear_mean1's slice, then linear_mean2's
  int* s_ctl = reinterpret_cast<int*>(s_wm + NKB * 64);
  float* ws_swgt = reinterpret_cast<float*>(smem_raw);
  double* ws_lblk = reinterpret_cast<double*>(smem_raw + (size_t)DP * 4);
  double* ws_lden = ws_lblk + UIS_RS_LOGTAB;
  unsigned char* ws_pers = reinterpret_cast<unsigned char*>(ws_lden + UIS_RS_LOGTAB);
  unsigned char* ws_scr = ws_pers + (size_t)nws * RL.persist_stride;
  const int u_w = cluster + ncl * (rank + 32 * wu);
  const bool has_u = wu < nws && u_w < U;
  const int cw = (wu + rank) & 1;  // this wave's utterance belongs to cohort cw (the ranks alternate, so that a rank's only utterance does not always land in A)
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
  }
  if (t < UIS_COH_CTL_BYTES / 4) s_ctl[t] = t == COH_SLOT ? 1 : 0;  // (linear_mean1's slice goes into the slot below)
  for (int i = t; i < DP; i += 512) ws_swgt[i] = m.wgt[i];
  for (int i = t; i < UIS_RS_LOGTAB; i += 512) { ws_lblk[i] = st.logblk[i]; ws_lden[i] = st.logden[i]; }
  // beam_set = [BeamState()] (uisrnn.py:528): one empty hypothesis, nothing live
  if (has_u)
    for (int i = lane; i < RL.persist_stride / 4; i += 64) reinterpret_cast<int*>(pers_w)[i] = 0;
  __syncthreads();
  if (has_u && lane == 0) {
    int* hdr = reinterpret_cast<int*>(pers_w + RL.off_hdr);
    hdr[0] = 1; hdr[1] = 1; hdr[2] = 1 << 20;  // one hypothesis, grid stride 1
    reinterpret_cast<int*>(pers_w + RL.off_hyp)[1] = -1;  // {K 0, last -1, sum 0, score 0}
  }
  {  // decode steps of this cluster = the longest of its utterances; utterances per cohort
    int myT = 0, mine[2] = {0, 0};
    for (int i = t; cluster + ncl * i < U; i += 512) {
      const int u = cluster + ncl * i;
      const long T = (long)st.tau * (long)(st.off[u + 1] - st.off[u]);
      myT = T > myT ? (int)T : myT;
      ++mine[((i >> 5) + (i & 31)) & 1];  // (utterance i of the cluster: rank i % 32, wave i / 32)
    }
    if (myT > 0) atomicMax(&s_ctl[COH_NSTEPS], myT);
    if (mine[0]) atomicAdd(&s_ctl[COH_NUTT], mine[0]);
    if (mine[1]) atomicAdd(&s_ctl[COH_NUTT + 1], mine[1]);
  }
  const int ft1 = rank / SH1, tpar1 = rank % SH1;  // ranks sharing a feature tile take alternate row tiles
  const int ft2 = rank / SH2, tpar2 = rank % SH2;
  const f32x4* w1g = reinterpret_cast<const f32x4*>(m.w1) + (size_t)ft1 * NKB * 64;
  const f32x4* w2g = reinterpret_cast<const f32x4*>(m.w2) + (size_t)ft2 * NKB * 64;
  for (int e = t; e < NKB * 64; e += 512) {
#pragma unroll
    for (int g = 0; g < 3; ++g)
      s_whh[g * NKB * 64 + e] = reinterpret_cast<const f32x4*>(m.whh[0])[(size_t)(g * NFT1 + ft1) * NKB * 64 + e];
    s_wm[e] = w1g[e];
  }
  __syncthreads();
  const int nsteps = s_ctl[COH_NSTEPS];
  const int n_utt0 = s_ctl[COH_NUTT], n_utt1 = s_ctl[COH_NUTT + 1];

  const __amdgpu_buffer_rsrc_t rs_rows =
      __builtin_amdgcn_make_buffer_rsrc((void*)st.rows, (short)0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_hid =
      __builtin_amdgcn_make_buffer_rsrc((void*)st.pool_hid, (short)0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a1 =
      __builtin_amdgcn_make_buffer_rsrc((void*)st.a1, (short)0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_mean =
      __builtin_amdgcn_make_buffer_rsrc((void*)st.pool_mean, (short)0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_hst =
      __builtin_amdgcn_make_buffer_rsrc((void*)st.gi_up, (short)0, 0x7fffffff, 0x00020000);
  // a cohort's rows, row counters (by step parity) and hand-off tiles: cohort B's region starts behind the
  // most rows cohort A can emit per step (beam_size per utterance), rounded up to a row tile (the host
  // sized rx_stride for both roundings)
  const int rows_b0 = (n_utt0 * st.B + 15) & ~15;
  const uint32_t h1_off = (uint32_t)((size_t)U * S * HP * 4);  // the extra slot holding h1
  uint32_t* const dbar = st.rx_flags + cluster * 32;           // [cohort] dense phases completed x 32 workgroups
  uint32_t* const selbar = dbar + 2;                            // [cohort] selects completed (one per utterance and step)
  int32_t* const rown = st.rx_nrows + cluster * 32;             // [2 cohort + parity]
  const float* bias_hh[3] = {m.bhh[0] + ft1 * 16, m.bhh[0] + HP + ft1 * 16, m.bhh[0] + 2 * HP + ft1 * 16};
  const float* bias_1[1] = {m.b1 + ft1 * 16};
  const float* bias_2[1] = {m.b2 + ft2 * 16};
  int nrows_c0 = 0, nrows_c1 = 0;

#if defined(UIS_RESIDENT_TIMING)
  // [ph] dependency wait, [6 + ph] tiles, [12] select (with its wait), [13] slot wait, [14] arrival (store drain, refill)
  // (kept in the control block's spare bytes: a register array indexed by the phase number would live in scratch)
  const bool rt_on = blockIdx.x == 0 && lane == 0 && (wu == 0 || wu == 1 || wu == 7);
  unsigned long long* rt_acc = reinterpret_cast<unsigned long long*>(s_ctl + 32) + 16 * (wu == 0 ? 0 : wu == 1 ? 1 : 2);
  unsigned long long rt_prev = wall_clock64();
#define CSTAMP(k) do { if (rt_on) { const unsigned long long now_ = wall_clock64(); rt_acc[k] += now_ - rt_prev; rt_prev = now_; } } while (0)
#else
#define CSTAMP(k) do {} while (0)
#endif

  for (int s = -1; s < nsteps; ++s) {
#pragma unroll 1
    for (int ph = s < 0 ? 5 : 0; ph < 6; ++ph) {
      const int kind = ph >> 1;  // 0 GRU, 1 linear_mean1, 2 linear_mean2
      const int c = ph & 1;      // the cohort whose rows this phase takes
      // ---- the select that rides on this phase: cohort B's for step s on phase 0, cohort A's for step s + 1 on phase 5
      const int s_sel = ph == 0 ? s : s + 1;
      if (has_u && ((ph == 0 && cw == 1) || (ph == 5 && cw == 0)) && s_sel < nsteps) {
        const bool act_w = (long)s_sel < T_w;
        const long frame_w = off0_w + fpos_w;
        const RsDims dm{st.B, st.Kmax, S, m.D};
        RsPrep<3> prep;
        if (act_w) prep = rs_prep<true, 3>(m, st, RL, dm, s_sel, pers_w, scr_w, ws_lblk, ws_lden, []() {});
        // every workgroup is through with this cohort's linear_mean2 of step s_sel - 1 (the means it reads,
        // the slots it reuses, the row list and hand-off tiles it overwrites)
        if (coh_wait(st, dbar + cw, 96u * (uint32_t)s_sel, &s_ctl[COH_SEEN + cw])) return;
        RsWin win;
        win.keep = 0; win.C = 0; win.nlead = 0; win.a = 0u; win.b = 0u; win.c = 0u; win.score = 0.0f;
        if (act_w) {
          win = rs_front<DP, true, 3>(m, st, RL, dm, u_w, s_sel, frame_w, pers_w, scr_w, rs_mean /* unused: FULL */, 0u, prep, nullptr, ws_swgt);
          int row_base = 0;
          if (lane == 0 && win.nlead > 0) row_base = atomicAdd(rown + 2 * cw + (s_sel & 1), win.nlead);
          row_base = __shfl(row_base, 0, 64);
          if (win.is_lead()) {
            RnnRow rr; rr.utt = u_w; rr.src = win.src(); rr.dst = win.dst(); rr.nprev = win.nprev(); rr.frame = frame_w; rr.pad = 0;
            st.rows[cluster * st.rx_stride + (cw ? rows_b0 : 0) + row_base + win.ord()] = rr;
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the rows have reached L2
        if (lane == 0) (void)__hip_atomic_fetch_add(selbar + cw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (act_w) {
          rs_back<3>(m, st, RL, dm, u_w, s_sel, off0_w, pers_w, true, win, []() {});
          fpos_w = fpos_w + 1 == N_w ? 0 : fpos_w + 1;
        }
        CSTAMP(12);
      }
      if (s < 0) continue;

      // ---- this phase's dependency: the cohort's previous stage, in every workgroup of the XCD
      if (kind == 0) {
        if (coh_wait(st, selbar + c, (uint32_t)(c ? n_utt1 : n_utt0) * (uint32_t)(s + 1), &s_ctl[COH_SEEN + 2 + c])) return;
        const int nr_ = __builtin_amdgcn_readfirstlane(__hip_atomic_load(rown + 2 * c + (s & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (c) nrows_c1 = nr_; else nrows_c0 = nr_;
        // (the other parity's counter: its readers -- step s - 1 -- are done, its next writers -- the selects of
        // step s + 1 -- wait for this cohort's linear_mean2 of step s, which this wave takes part in)
        if (rank == 0 && t == 0) __hip_atomic_store(rown + 2 * c + ((s & 1) ^ 1), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (s == 0 && c == 0 && t == 0) {
          if (rank == 1 && (st.flags & 0x100u)) xcc ^= 1u;  // UIS_FLAG_TEST_MISPLACED: pretend
          if (__hip_atomic_load(st.cl_xcc + cluster, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != xcc)
            __hip_atomic_store(st.cl_abort, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // not on one XCD
        }
        CSTAMP(ph);
      } else {
        if (coh_wait(st, dbar + c, 32u * (uint32_t)(3 * s + kind), &s_ctl[COH_SEEN + c])) return;
        CSTAMP(ph);
        if (coh_slot_wait(st, &s_ctl[COH_SLOT], 2 * s + kind)) return;
        CSTAMP(13);
      }
      const int nrows = c ? nrows_c1 : nrows_c0;
      const int nrt = (nrows + 15) >> 4;
      const int rbase = cluster * st.rx_stride + (c ? rows_b0 : 0);  // this cohort's rows of `rows`
      const int tile0 = rbase >> 4;                                  // ... and its first row tile of the hand-off buffers
      int* const s_next = &s_ctl[COH_TILE + ph];

      if (kind == 0) {
        // ---- GRU: h' = gru(gi0[frame], W_hh h_src + b_hh) -> dst slot and, k-block-major, the hand-off tile
        // linear_mean1 streams; the next tile's descriptor and first rows are requested while the current
        // tile's chain runs
        constexpr int GSG = 2, GBG = GSG * (NKB / UIS_KSPLIT);
        const int ntl = nrt > tpar1 ? (nrt - tpar1 + SH1 - 1) / SH1 : 0;
        int k = coh_lds_add(s_next, 1);
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
        if (k < ntl) {
          fetch_head(tpar1 + SH1 * k, rh, frame, hoff);
          rows_first_group<GBG, 64>(rs_hid, hoff, bfirst);
        }
        while (k < ntl) {
          const int tile = tpar1 + SH1 * k;
          const int kn = coh_lds_add(s_next, 1);
          const bool has_next = kn < ntl;
          RowHead rh_n{0, 0, 0, 0};
          long frame_n = 0;
          uint32_t hoff_n = h1_off;
          if (has_next) fetch_head(tpar1 + SH1 * kn, rh_n, frame_n, hoff_n);
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
            rs_buf_store_f32x4(rs_hid, (uint32_t)(((rh.utt * S + rh.dst) * HP + j4) * 4), out);
            rs_buf_store_f32x4(rs_hst, (uint32_t)(((tile0 + tile) * NFT1 + ft1) * 256 + (lane & 15) * 16 + 4 * q) * 4u, out);
          }
          k = kn; rh = rh_n; frame = frame_n; hoff = hoff_n;
        }
      } else if (kind == 1) {
        // ---- linear_mean1 + relu -> a1 (same staging layout)
        constexpr int GBH = 4 * (NKB / UIS_KSPLIT);
        const int ntl = nrt > tpar1 ? (nrt - tpar1 + SH1 - 1) / SH1 : 0;
        for (int k = coh_lds_add(s_next, 1); k < ntl; k = coh_lds_add(s_next, 1)) {
          const int tile = tpar1 + SH1 * k;
          const uint32_t soff = (uint32_t)((((tile0 + tile) * NFT1) * 256 + (lane & 15) * 16) * 4);
          const bool valid = 16 * tile + (lane & 15) < nrows;
          f32x4 v[1], bfirst1[GBH];
          rows_first_group<GBH, 1024>(rs_hst, soff, bfirst1);
          fullk_rows_sc1<1, NKB, 4, 1024>(s_wm, 0, bias_1, rs_hst, soff, v, bfirst1, 0u, false);
          if (valid) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[0][i] = v[0][i] > 0.0f ? v[0][i] : 0.0f;
            rs_buf_store_f32x4(rs_a1, (uint32_t)(((tile0 + tile) * NFT1 + ft1) * 256 + (lane & 15) * 16 + 4 * q) * 4u, v[0]);
          }
        }
      } else {
        // ---- linear_mean2 + running mean -> dst slot
        constexpr int GBH = 4 * (NKB / UIS_KSPLIT);
        const int ntl = nrt > tpar2 ? (nrt - tpar2 + SH2 - 1) / SH2 : 0;
        for (int k = coh_lds_add(s_next, 1); k < ntl; k = coh_lds_add(s_next, 1)) {
          const int tile = tpar2 + SH2 * k;
          const uint32_t soff = (uint32_t)((((tile0 + tile) * NFT1) * 256 + (lane & 15) * 16) * 4);
          const int row = 16 * tile + (lane & 15);
          const bool valid = row < nrows;
          const RowHead rh = load_row_head(rs_rows, rbase + (valid ? row : 16 * tile));
          const int f4 = ft2 * 16 + 4 * q;
          f32x4 old = {0.0f, 0.0f, 0.0f, 0.0f};
          if (valid && rh.src >= 0) old = load_sc1(rs_mean, (uint32_t)(((rh.utt * S + rh.src) * DP + f4) * 4));
          f32x4 v[1], bfirst2[GBH];
          rows_first_group<GBH, 1024>(rs_a1, soff, bfirst2);
          fullk_rows_sc1<1, NKB, 4, 1024>(s_wm, 0, bias_2, rs_a1, soff, v, bfirst2, 0u, false);
          if (valid) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              if (rh.src >= 0) v[0][i] = uis_mean_update(old[i], v[0][i], rh.nprev);
              if (f4 + i >= m.D) v[0][i] = 0.0f;
            }
            rs_buf_store_f32x4(rs_mean, (uint32_t)(((rh.utt * S + rh.dst) * DP + f4) * 4), v[0]);
          }
        }
      }
      CSTAMP(6 + ph);

      // ---- this wave is through with the phase; the workgroup's last wave says so to the XCD (its own and
      // the other waves' stores have reached L2: every wave drains before it counts itself) and, behind the
      // phases that end a use of the mean-head slot, refills it
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (coh_lds_add(&s_ctl[COH_FIN + ph], 1) == 7) {
        if (lane == 0) {
          __hip_atomic_store(&s_ctl[COH_FIN + ph], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_store(s_next, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // (all eight failing pulls are in)
          if (ph == 3) __hip_atomic_store(&s_ctl[COH_FREE], 2 * s + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          (void)__hip_atomic_fetch_add(dbar + c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (ph == 5 && s + 1 < nsteps) {
          // linear_mean1's slice for the next step: nobody needs it before the two GRU phases are over, so this
          // wave alone fetches it (sixteen 16-byte loads per lane in flight) while the other seven go on
          constexpr int CH = NKB < 16 ? NKB : 16;
#pragma unroll 1
          for (int e0 = 0; e0 < NKB * 64; e0 += CH * 64) {
            f32x4 tmp[CH];
#pragma unroll
            for (int i = 0; i < CH; ++i) tmp[i] = w1g[e0 + 64 * i + lane];
#pragma unroll
            for (int i = 0; i < CH; ++i) s_wm[e0 + 64 * i + lane] = tmp[i];
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (lane == 0) __hip_atomic_store(&s_ctl[COH_SLOT], 2 * s + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
      if (ph == 3) {
        // linear_mean2's slice: the next phase needs it at once, so all eight waves fetch an eighth each as soon
        // as the last of them is through with linear_mean1's (one round trip)
        if (coh_slot_wait(st, &s_ctl[COH_FREE], 2 * s + 2)) return;
        constexpr int PART = NKB * 64 / 8;  // 16-byte elements per wave
        f32x4 tmp[PART / 64];
#pragma unroll
        for (int i = 0; i < PART / 64; ++i) tmp[i] = w2g[wu * PART + 64 * i + lane];
#pragma unroll
        for (int i = 0; i < PART / 64; ++i) s_wm[wu * PART + 64 * i + lane] = tmp[i];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (coh_lds_add(&s_ctl[COH_PARTS], 1) == 7 && lane == 0) {
          __hip_atomic_store(&s_ctl[COH_PARTS], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_store(&s_ctl[COH_SLOT], 2 * s + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
      CSTAMP(14);
    }
  }
#undef CSTAMP
#if defined(UIS_RESIDENT_TIMING)
  if (rt_on)
    for (int k = 0; k < 16; ++k) st.counters[(wu == 0 ? 48 : wu == 1 ? 64 : 80) + k] = rt_acc[k];
#endif
  if (has_u && lane == 0) {  // this utterance's statistics
    const unsigned long long* acc = reinterpret_cast<const unsigned long long*>(pers_w + RL.off_stats);
    atomicAdd(&st.counters[0], acc[0]);
    atomicAdd(&st.counters[1], acc[1]);
    atomicAdd(&st.counters[2], acc[2]);
    atomicMax(&st.counters[3], acc[3]);
  }
}
