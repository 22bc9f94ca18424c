// uis_kernels.h -- device data layout and kernel argument blocks of the gfx950
// UIS-RNN decoder.  See DESIGN.md for the layout rationale.
//
// Vocabulary (follows the reference, uisrnn/uisrnn.py):
//   utterance   one test sequence [N, D]
//   hypothesis  one BeamState (uisrnn.py:55-77): K clusters, block counts, last
//               cluster, neg_likelihood
//   cluster state  (mean, hidden, frame count) of one cluster of one hypothesis;
//               lives in a per-utterance SLOT POOL and is shared between
//               hypotheses copy-on-write, like the reference's shallow list copies
//   rnn row     one CoreRNN.forward evaluation (uisrnn.py:45-52) = one row of the
//               batched GRU / mean-head GEMMs
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define UIS_MAX_DEPTH 8
#define UIS_MAX_LOOKAHEAD 1024  // (window records are (look_ahead + 1) uint16 per hypothesis; nothing else is sized by it)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// One CoreRNN evaluation scheduled by the select/expand kernels.
struct RnnRow {
  int32_t utt;     // utterance index
  int32_t src;     // source slot (cluster state before this frame), -1 = fresh cluster (h1)
  int32_t dst;     // destination slot
  int32_t nprev;   // frames already assigned to the cluster (0 for a fresh cluster)
  int64_t frame;   // row of the packed frame stream
  int64_t pad;
};

#define UIS_PM_PUSH 1u
#define UIS_PM_LABELS 2u
#define UIS_PM_QUIT 3u
#define UIS_PM_IDLE 4u          // (kernel-side only: nothing arrived for pm_idle_ticks)
#define UIS_PM_MAX_CLUSTERS 16
// the mailbox's control words (uint32 view of the block's first bytes), one 64-byte line each:
#define UIS_PM_BELL_WORD 0      // + 16 c: cluster c's doorbell {sequence number, command | frames << 8, first row | rows << 16, sequence number again}
#define UIS_PM_DONE_WORD 256    // + 16 c: sequence number of the last command cluster c completed
#define UIS_PM_LEFT_WORD 512    // + 16 c: non-zero once cluster c has left the kernel (1 told to, 2 idle)
#define UIS_PM_TIMING_WORD 768  // 6 x uint64 of the -DUIS_PM_TIMING build
#define UIS_PM_CTL_WORDS 832

struct DevModel {
  int D, H, depth, Dp, Hp, G;  // G = 3*Hp (gates r|z|n, each padded to Hp)
  // weights in MFMA tile order: [feature tile][k block][lane 0..63][4]
  const float* wih[UIS_MAX_DEPTH];
  const float* whh[UIS_MAX_DEPTH];
  const float* bih[UIS_MAX_DEPTH];  // [G]
  const float* bhh[UIS_MAX_DEPTH];  // [G]
  const float* w1;  const float* b1;   // (Hp x Hp) tiled, [Hp]
  const float* w2;  const float* b2;   // (Dp x Hp) tiled, [Dp]
  const float* wgt;    // [Dp]  1 / (2 sigma2), 0 in the padding
  const float* m0;     // [Dp]  mean of a fresh cluster before its first frame
  const float* h1;     // [depth][Hp] hidden of a fresh cluster before its first frame
  double lp_stay, lp_sw, l_alpha;
  double lp_new;  // lp_sw + l_alpha, added once on the host (the same IEEE sum the kernels formed per candidate)
};

// Persistent streaming launch (UIS_FLAG_PERSISTENT sessions): k_decode_resident stays on the device
// between pushes and takes its commands from a block of host-coherent pinned memory.  `ctl` and
// the other pointers below name HOST memory mapped into the device's address space, except go / hdr.
// ctl: the UIS_PM_*_WORD lines above.
struct PersistArgs {
  uint32_t* ctl;
  const int64_t* foff;      // [U]   as DecodeState::foff, of the current push
  const int32_t* avail;     // [U]   as DecodeState::avail
  const int64_t* lab_off;   // [U]   where utterance u's labels go in `labels` (LABELS command)
  const float* frames;      // [push frames][D]
  int32_t* labels;
  float* scores;            // [U]
  float* beam_scores;       // [U][B]
  int32_t* overflow;        // [U]
  unsigned long long* go;   // device, one 128-byte line per cluster, one 64-bit word: sequence number (16 bits) | command (4) | frames (12) | first row (16) | rows (16)
  unsigned char* hdr;       // device, per cluster: [foff U x 8][avail U x 4]; stride hdr_stride
  size_t hdr_stride;
  unsigned long long idle_ticks;  // 10 ns ticks without a command after which the launch ends by itself
};

// Everything the per-step kernels need about the running decode.
struct DecodeState {
  int U, B, Kmax, S, L, tau;
  int max_rows;           // capacity of `rows` = the most rnn rows one step can emit
  uint32_t flags;
  // round 5: a decode in TWO launches (k_decode_rs, k_decode_big<WS>): this launch runs decode steps [step0, step1) of
  // every utterance (step1 = 0: to the end); a launch that stops early leaves the beam state it keeps in LDS in
  // `resume` (resume_stride bytes per cluster) and the next one picks it up -- so that the later frames of every
  // utterance may still be on their way to the device while the first launch decodes the earlier ones
  int step0, step1;
  unsigned char* resume;
  size_t resume_stride;
  int wnd;                // 1: the window machinery decodes (k_window, level buffers, window records): look_ahead >= 2,
                          // and look_ahead 1 with a beam / cluster cap the select kernels do not take (round 5)
  // utterances
  const int64_t* off;     // [U+1] frame offsets
  int32_t* utt_step;      // [U] next decode step of each utterance
  int32_t* overflow;      // [U]
  // frame stream
  const float* x;         // [frames][Dp]
  const float* gi0;       // [frames][G]   W_ih0 x + b_ih0
  const float* mse0;      // [frames]      weighted MSE against m0
  // prior tables (float64)
  const double* logblk;   // [Tmax+2] log(n)
  const double* logden;   // [Tmax+2] log(n + alpha)
  // slot pool
  float* pool_mean;       // [U][S][Dp]
  float* pool_hid;        // [U][S][depth][Hp]
  int32_t* pool_cnt;      // [U][S]
  // beam tables, double buffered on step parity: index ((par*U + u)*B + b)
  int32_t* beam_n;        // [2][U]
  int32_t* beam_K;        // [2][U][B]
  int32_t* beam_last;     // [2][U][B]
  int32_t* beam_sum;      // [2][U][B]   sum(block_counts)
  float*   beam_score;    // [2][U][B]
  int32_t* beam_slot;     // [2][U][B][Kmax]
  int32_t* beam_blk;      // [2][U][B][Kmax]
  // back-pointers: bp[tau*off[u]*B + step*B + r] = (parent << 16) | cluster
  uint32_t* bp;
  // rnn rows of the current step, appended with atomics
  RnnRow* rows;           // [U*B]
  int32_t* nrows;         // [2] row counters, by step parity
  // intermediates, one row per rnn row
  float* gi_up;           // [U*B][G]   input-side gates of GRU layers >= 1
  float* a1;              // [U*B][Hp]  relu(linear_mean1)
  // UIS_FLAG_DEBUG_SCORES: every candidate score of every step, [step][U][B][Kmax + 1] (+inf filled), or null
  float* dbg_scores;
  // counters (device): [0] rnn rows, [1] rnn rows without dedup, [2] candidates, [3] max K
  unsigned long long* counters;
  // in-launch barrier bookkeeping of k_decode_resident: the XCC id each cluster's rank 0 saw
  // [ncl] and a sticky abort word (1 = barrier timed out, 2 = a cluster is not on one XCD)
  uint32_t* cl_xcc;
  uint32_t* cl_abort;
  // k_decode_resident: the device's CUs form `ncl` clusters of 32 workgroups (one per XCD: 8 on
  // a whole MI355X, fewer in partitioned modes).  Cluster c owns utterances c, c+ncl, ... and
  // rows [c*rx_stride, (c+1)*rx_stride) of `rows` / `a1`; its row counters (by step parity) are
  // rx_nrows[c*32 + par], its barrier counter rx_bar[c*32].  pool_hid carries one extra slot
  // [U*S] holding h1 so that every GRU source row lives in one buffer.
  int ncl;
  int rx_stride;
  int32_t* rx_nrows;
  uint32_t* rx_bar;
  int32_t* utt_nrows;     // [U][2] k_decode_small: an utterance's row counters, by step parity
  float* hst;             // k_decode_deep: two hand-off buffers [2][hst_elems] (layer l writes [l & 1]), k-block-major row tiles
  size_t hst_elems;
  uint32_t* rx_flags;     // [ncl][32] per-producer phase words (one 128-byte line per cluster): the hand-offs between the dense stages
  // k_decode_rs (replicated select): mse_tab[((step parity) * U + u) * S + slot] = weighted MSE of
  // that step's frame against the cluster mean in `slot`, published one step ahead by the
  // utterance's owner rank for the clusters the step in between does not rewrite
  float* mse_tab;
  // ... and mse_part[(c * rx_stride + row) * 32 + ft] = tile ft's partial sum (uis_numerics.h) of the
  // NEXT step's weighted MSE against the mean that row `row` of cluster c's current step writes,
  // [.. + 16] = the squared first difference; emitted by the linear_mean2 epilogue
  float* mse_part;
  // streaming (uis_stream_*): utterances are NOT in lock-step.  avail[u] = frames received so far
  // (= decode steps that may run; test_iteration is 1), foff[u] + step = row of step `step`'s
  // frame in the current chunk's x / gi0 / mse0, lab_off[u] = where the utterance's labels go.
  // `off` then holds capacity offsets (u * max_frames): it only addresses the back-pointers.
  // All three are null in an ordinary decode.
  const int32_t* avail;
  const int64_t* foff;
  const int64_t* lab_off;
  // streaming push with the chunk's once-per-frame work fused into k_decode_resident: push_F > 0 =
  // the chunk holds push_F frames whose gi0 / mse0 the kernel computes itself (each cluster for
  // its own utterances' frames) before its first step, instead of two extra launches
  int push_F;
  // persistent streaming launch (UIS_FLAG_PERSISTENT sessions): where the session's mailbox is
  // (device memory copy of a PersistArgs); null in every other launch
  const struct PersistArgs* pm;

  // ---- look_ahead >= 2 only (k_window): intermediate hypothesis levels of the current window.
  // Two level buffers (ping-pong over sub-steps), NC hypotheses each per utterance.
  int NC;                 // level capacity per utterance
  int32_t* lv_n;          // [2][U]
  int32_t* lv_K;          // [2][U][NC]
  int32_t* lv_last;       // [2][U][NC]
  int32_t* lv_sum;        // [2][U][NC]
  float*   lv_score;      // [2][U][NC]
  int32_t* lv_origin;     // [2][U][NC]   beam hypothesis the node descends from
  int16_t* lv_path;       // [2][U][NC][L] clusters chosen so far in this window
  int32_t* lv_slot;       // [2][U][NC][Kmax]
  int32_t* lv_blk;        // [2][U][NC][Kmax]
  unsigned char* scratch; // per-utterance work arrays of k_window
  size_t scratch_stride;  // bytes per utterance
  // back-pointers per window: bp16[(bp_base[u] + w*B + r)*(L+1)] = {parent, c_1 .. c_L}
  uint16_t* bp16;
  const int64_t* bp_base; // [U]
};
