/*
 * uisrnn_hip.h -- C ABI of the MI355X (gfx950) UIS-RNN beam-search decoder.
 *
 * This is the drop-in boundary for ONE path of google/uis-rnn: the inference
 * beam search  UISRNN.predict -> predict_single -> _calculate_score ->
 * _update_beam_state -> CoreRNN / BeamState
 * (reference: uisrnn/uisrnn.py:388-590, uisrnn/loss_func.py:19-41).
 *
 * The reference has no FFI of its own -- its boundary is the Python method set
 * on uisrnn.UISRNN (uisrnn/__init__.py:26-30).  A maintainer binds these entry
 * points with ctypes (cffi is the other option) exactly as uisrnn_amd/_capi.py
 * does; INTEGRATION.md shows the stub.  Plain C types only: no torch, no HIP
 * types in the signatures.  Device pointers are passed as void* / float*.
 *
 * Threading: a handle is bound to one HIP device and is not thread-safe; use
 * one handle per thread / per rank.  All calls are blocking.
 * Errors: every function returns UIS_OK (0) or a negative uis_status;
 * uis_last_error() returns a thread-local message for the last failure; the
 * pointer is good until this thread's next failing call (copy the text if it
 * is to be kept, and call it AFTER the call it explains -- as an argument of
 * the same printf its evaluation order is unspecified in C).
 */
#ifndef UISRNN_HIP_H_
#define UISRNN_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UIS_ABI_VERSION 6

typedef enum uis_status {
  UIS_OK = 0,
  UIS_ERR_INVALID_ARG = -1,   /* NULL pointer, non-positive size, bad option      */
  UIS_ERR_DIM_MISMATCH = -2,  /* maps to the reference's ValueError (uisrnn.py:518-521) */
  UIS_ERR_NO_DEVICE = -3,     /* no usable gfx950 device / HIP runtime failure at create */
  UIS_ERR_HIP = -4,           /* a HIP call failed during decode                   */
  UIS_ERR_OOM = -5,           /* device or host allocation failed                  */
  UIS_ERR_CLUSTER_CAP = -6,   /* a surviving hypothesis needed more than max_clusters
                                 clusters; labels_out of the flagged utterances are
                                 invalid -- retry with a larger cap (the Python host does) */
  UIS_ERR_UNSUPPORTED = -7    /* an option value beyond a field width (beam_size > 32767, look_ahead > 1024,
                                 max_clusters > 4096), a UIS_FLAG_RESIDENT request where no one-launch kernel
                                 applies; also: a look_ahead >= 2 window had more live assignment prefixes than
                                 the level capacity (uis_decode_opts.level_cap: bit 1 of the utterance's flags in
                                 uis_last_decode_info; decode those again with a larger one -- a larger
                                 max_clusters cannot help)                                    */
} uis_status;

/*
 * Model parameters consumed by predict (reference: UISRNN.__init__ / load,
 * uisrnn/uisrnn.py:83-107,149-170; CoreRNN, uisrnn/uisrnn.py:32-52).
 * All arrays are host pointers, float32, row-major, PyTorch layouts; they are
 * copied at uis_create and need not outlive it.
 */
typedef struct uis_model_desc {
  int32_t observation_dim;            /* D  (args.observation_dim)            */
  int32_t rnn_hidden_size;            /* H  (args.rnn_hidden_size)            */
  int32_t rnn_depth;                  /* number of GRU layers                 */
  int32_t reserved0;
  const float* const* gru_weight_ih;  /* [depth] -> (3H, D) layer 0, (3H, H) above; gates r|z|n */
  const float* const* gru_weight_hh;  /* [depth] -> (3H, H)                   */
  const float* const* gru_bias_ih;    /* [depth] -> (3H)                      */
  const float* const* gru_bias_hh;    /* [depth] -> (3H)                      */
  const float* linear_mean1_weight;   /* (H, H)                               */
  const float* linear_mean1_bias;     /* (H)                                  */
  const float* linear_mean2_weight;   /* (D, H)                               */
  const float* linear_mean2_bias;     /* (D)                                  */
  const float* rnn_init_hidden;       /* (depth, H)  (rnn_init_hidden[:,0,:]) */
  const float* sigma2;                /* (D)                                  */
  double transition_bias;             /* p0 in (0,1)                          */
  double crp_alpha;                   /* alpha > 0                            */
} uis_model_desc;

/* Inference options (reference: uisrnn/arguments.py:172-193). */
typedef struct uis_decode_opts {
  int32_t beam_size;       /* args.beam_size       (>= 1; beyond 256 -- and wherever max_clusters outgrows the
                              select kernels' LDS -- the window machinery decodes: a launch per sub-step,
                              candidate lists in HBM; at most 32767)                                */
  int32_t look_ahead;      /* args.look_ahead      (>= 1, at most 1024)       */
  int32_t test_iteration;  /* args.test_iteration  (>= 1)                     */
  int32_t max_clusters;    /* per-hypothesis cluster cap; 0 = default (16)    */
  uint32_t flags;          /* UIS_FLAG_*                                      */
  int32_t n_streams;       /* utterance groups decoded concurrently, each on its
                              own HIP stream; 0 = default (1), at most 8       */
  int32_t level_cap;       /* look_ahead >= 2: hypotheses an intermediate level of a window may hold per
                              utterance; 0 = default (32768), at most 524287.  A window with more live
                              assignment prefixes sets bit 1 of the utterance's flags and the call returns
                              UIS_ERR_UNSUPPORTED: decode those utterances again with a larger level_cap
                              (the Python host does, until the device's memory says UIS_ERR_OOM)      */
  int32_t reserved[1];
} uis_decode_opts;

#define UIS_FLAG_NO_DEDUP   0x1u /* run one RNN row per surviving hypothesis even when
                                    several share the same cluster state (A/B switch;
                                    results are bit-identical either way)            */
#define UIS_FLAG_GRAPH      0x2u /* replay the per-step kernels from a captured hipGraph
                                    (32 steps per graph) instead of launching eagerly  */
#define UIS_FLAG_GENERIC_SELECT 0x8u /* use the general score/prune kernel even where the
                                    wave-synchronous fast path applies (A/B switch;
                                    results are bit-identical either way)            */
#define UIS_FLAG_RESIDENT   0x40u /* REQUIRE the one-launch decode (k_decode_resident: one workgroup
                                    per CU in clusters of 32 -- one cluster per XCD -- W_hh in
                                    registers and the mean-head tiles in LDS for the whole decode,
                                    one XCD per utterance subset, in-launch XCD barriers between
                                    the stages of a step; launched cooperatively so that all
                                    workgroups are co-resident) and fail with UIS_ERR_UNSUPPORTED
                                    where it does not apply.  It is the DEFAULT wherever it
                                    applies: look_ahead 1, rnn_depth 1, rnn_hidden_size 65 .. 256 or
                                    385 .. 512 and observation_dim up to 256 or 385 .. 512 (the model is
                                    padded up to 128 / 256 / 512 x 128 / 256 / 512 at uis_create; rnn_depth >= 2
                                    at those sizes: k_decode_deep, UIS_DK_DEEP),
                                    beam_size * (max_clusters + 1) <= 256, one stream, a device
                                    whose CU count is a multiple of 32; small models (hidden size up
                                    to about 64, any rnn_depth) with one workgroup per utterance;
                                    look_ahead >= 2 at rnn_depth 1 and those hidden sizes with the
                                    window's sub-step as the select stage (UIS_DK_WINDOW)        */
#define UIS_FLAG_STEPWISE   0x80u /* keep the launch-per-step path (four kernels per decode
                                    step) even where the one-launch decode applies (A/B switch;
                                    results are bit-identical either way)                     */
#define UIS_FLAG_SMALL_TILES 0x200u /* keep the split-K dense stages even where the wave-per-row-tile
                                    ones apply: the big-tile kernels of the launch-per-step path
                                    (thousands of rnn rows per step) and k_decode_big of the one-launch
                                    path (more utterances than workgroups); A/B switch, results are
                                    bit-identical either way                                     */
#define UIS_FLAG_PERSISTENT 0x400u /* uis_stream_begin only: the one-launch decode kernel STAYS on the device
                                    between pushes and takes pushes / label requests from a mailbox in
                                    host-coherent pinned memory -- a push costs no launch and no copy
                                    engine, only the steps themselves.  The kernel occupies every compute
                                    unit until uis_stream_end or until it has been idle for
                                    UIS_PERSIST_IDLE_MS (environment, default 50 ms; it then leaves and
                                    the next push starts a new one).  Needs the one-launch shape, at
                                    most one utterance per compute unit, unpadded observation_dim;
                                    pushes of more than 16 frames per utterance go the ordinary way */
#define UIS_FLAG_OWNER_SELECT 0x800u /* one-launch decode: keep the select of an utterance on ONE
                                    workgroup (k_decode_resident) even where the replicated select
                                    applies (k_decode_rs: at most 8 utterances per XCD, every
                                    workgroup of the XCD decides all of them, one wave each -- one
                                    in-launch hand-off less per step and a select short enough for a
                                    single wave); A/B switch, results are bit-identical either way */
#define UIS_FLAG_REPLICATED_SELECT 0x1000u /* one-launch decode: use the replicated select (k_decode_rs) also in
                                    the shape classes where it is NOT the default because the owner-select
                                    kernel measured faster there: beam_size 17 .. 32 or observation_dim 512
                                    (its "wide" class) and 9 .. 16 utterances per XCD (two utterances per
                                    wave); A/B switch, results are bit-identical either way         */
#define UIS_FLAG_CLUSTER_BARRIERS 0x8000u /* one-launch decode with the owner select (k_decode_resident): keep the
                                    cluster-wide barriers between GRU, linear_mean1 and linear_mean2 instead of
                                    the per-producer phase words (a consumer wave waits for the four workgroups
                                    that produce its K-slice); A/B switch, results are bit-identical either way */
#define UIS_FLAG_COHORTS 0x10000u /* one-launch decode with many utterances per XCD (where k_decode_big<WS> applies): run an
                                    XCD's utterances as TWO cohorts whose stages alternate on every workgroup, the selects
                                    riding on the other cohort's dense phases, row tiles pulled from LDS counters, no
                                    workgroup barrier in the step loop (k_decode_coh, UIS_DK_BIG_COH).  Bit-identical;
                                    measured SLOWER than the lock-step batch on MI355X (3.8 against 4.0 M frames/s at
                                    1024 utterances: DESIGN.md / LABNOTES.md): since round 6 the kernel is compiled only
                                    into builds with -DUIS_WITH_COHORTS (uis_build_flags() & UIS_BUILD_COHORTS; the test
                                    variant build/variants/cohorts.so); the product library ignores the flag          */
#define UIS_FLAG_AGENT_FLAGS 0x20000u /* one-launch decode: publish the per-producer phase words of the dense-stage hand-offs
                                    with an AGENT-scope store (`global_store sc1`: the HIP memory model's by-the-book form for
                                    a word other workgroups read) instead of the workgroup-scope store that stays in the
                                    XCD's L2 (uis_kernels.hip: rs_flag_publish).  A run-time choice in one binary since round
                                    6 (the environment's UIS_AGENT_FLAGS=1 does the same); bit-identical; its cost is on
                                    record: profiles/r06_agent_flags_ab.txt -- it is NOT small, hence opt-in          */
#define UIS_FLAG_DEBUG_SCORES 0x2000u /* test hook: keep every candidate score of every window (step) --
                                    the arrays _calculate_score returns (uisrnn/uisrnn.py:455-477) -- for
                                    uis_debug_scores(); costs device memory and one store per candidate */
#define UIS_FLAG_TEST_MISPLACED 0x100u /* test hook: one workgroup of the one-launch decode reports
                                    a wrong XCD, as if the (observed, not promised) workgroup
                                    placement had changed.  The call must then fall back to the
                                    launch-per-step path by itself -- and stay there for this
                                    handle -- or, with UIS_FLAG_RESIDENT, fail with UIS_ERR_HIP */
#define UIS_FLAG_TEST_STALL 0x4000u /* test hook (k_decode_rs): one workgroup stops publishing its phase
                                    word after a few steps, as if it had died.  The waves that wait for
                                    it give up after ~1 s, the launch ends, and the call falls back to
                                    the launch-per-step path (with UIS_FLAG_RESIDENT: UIS_ERR_HIP) */
#define UIS_FLAG_PROFILE    0x4u /* launch every kernel with start/stop HIP events on the
                                    decode stream (hipExtLaunchKernelGGL: the dispatch's
                                    own begin/end timestamps) and fill uis_stats.kernel_* */

#define UIS_N_KERNELS 8
typedef struct uis_stats {
  int32_t n_steps;                    /* lock-step decode steps executed            */
  int32_t max_clusters_seen;          /* max clusters in any surviving hypothesis   */
  int64_t rnn_rows;                   /* RNN rows actually evaluated (after dedup)  */
  int64_t rnn_rows_nodedup;           /* rows without dedup = surviving hypotheses  */
  int64_t candidates;                 /* candidates scored                           */
  double  decode_ms;                  /* device time of the whole decode (events)   */
  double  kernel_ms[UIS_N_KERNELS];   /* UIS_FLAG_PROFILE: summed per kernel class  */
  int64_t kernel_launches[UIS_N_KERNELS];
  int32_t n_overflow;                 /* utterances that hit UIS_ERR_CLUSTER_CAP    */
  int32_t n_streams;                  /* utterance groups used                       */
  int32_t decode_kernel;              /* UIS_DK_*: the kernel family that ran the decode steps */
  int32_t decode_launches;            /* launches of the one-launch decode kernel in this decode: 1, or 2 and more (2 - 4 by
                                         the library's own slice schedule, up to 8 with UIS_SPLIT_FRAMES) when the list came
                                         from host memory and its later frames travelled behind the earlier launches
                                         (uis_decode / uis_decode_f64: k_decode_rs, k_decode_big<WS>, k_decode_resident with
                                         one utterance per workgroup; UIS_NO_SPLIT=1 keeps 1); 0 on the launch-per-step path */
} uis_stats;

/* uis_stats.decode_kernel: which kernels ran the decode steps (the dispatch rule lives in the
 * library, uis_decoder.hip: callers that want to NAME the kernel read it here) */
enum {
  UIS_DK_NONE = 0,
  UIS_DK_STEPWISE = 1,   /* launch per step: k_select* / k_window + the dense kernels below   */
  UIS_DK_RS = 2,         /* one launch, replicated single-wave select (k_decode_rs)           */
  UIS_DK_RESIDENT = 3,   /* one launch, owner select (k_decode_resident)                      */
  UIS_DK_BIG = 4,        /* one launch, a wave per row tile (k_decode_big)                    */
  UIS_DK_BIG_WS = 5,     /* ... with a rank's selects running concurrently (k_decode_big<WS>) */
  UIS_DK_SMALL = 6,      /* one launch, one workgroup per utterance: small models, any rnn_depth, any look_ahead (k_decode_small) */
  UIS_DK_WINDOW = 7,     /* one launch, look_ahead >= 2: a window sub-step as the select stage (k_decode_big<WIN>) */
  UIS_DK_DEEP = 8,       /* one launch, rnn_depth >= 2 at hidden size 128 / 256 / 512: the weight slot refilled per stage (k_decode_deep) */
  UIS_DK_BIG_COH = 9     /* one launch, a wave per row tile, two utterance cohorts in flight per XCD (k_decode_coh; UIS_FLAG_COHORTS) */
};
/* ... in bits 16..23 for UIS_DK_RS its instantiation: 1 base, 2 base with the shape of BASELINE configs[1] as
 * compile-time constants, 3 two utterances per wave (9 .. 16 per XCD), 4 wide (beam_size <= 32 / observation
 * dim 512); and, in bits 8..15 for UIS_DK_STEPWISE, the dense kernels' family */
enum { UIS_DF_DENSE = 1 /* k_dense_* split-K */, UIS_DF_BIG = 2 /* k_big_* */, UIS_DF_WT = 3 /* k_wt_* */ };

/* kernel classes for uis_stats.kernel_ms */
enum {
  UIS_K_INPUT_PROJ = 0, /* W_ih0 x + b over all frames, new-cluster MSE      */
  UIS_K_SELECT = 1,     /* score + prune + commit                           */
  UIS_K_GRU = 2,        /* hidden-side GRU GEMM + gates                     */
  UIS_K_HEAD1 = 3,      /* linear_mean1 + relu                              */
  UIS_K_HEAD2 = 4,      /* linear_mean2 + running-mean update               */
  UIS_K_BACKTRACE = 5,  /* back-pointer walk -> labels                      */
  UIS_K_UPPER_IN = 6,   /* input-side GEMM of GRU layers >= 1               */
  UIS_K_EXPAND = 7      /* look_ahead >= 2: intermediate sub-step expansion */
};

typedef struct uis_handle uis_handle;

int32_t uis_abi_version(void);

/* UIS_NUMERICS_VERSION of include/uis_numerics.h this library was built with. */
int32_t uis_numerics_version(void);

/* What this binary was built with (round 6): a mask of UIS_BUILD_*. */
#define UIS_BUILD_COHORTS 0x1u  /* -DUIS_WITH_COHORTS: k_decode_coh is there and UIS_FLAG_COHORTS selects it */
uint32_t uis_build_flags(void);

/* Number of visible HIP devices (0 if none / runtime unusable). */
int32_t uis_device_count(void);

/*
 * Create a decoder on HIP device `device`: uploads the weights in the padded
 * layouts the kernels use and precomputes the per-model constants
 * (m0, h1) = CoreRNN(0, rnn_init_hidden)  (uisrnn/uisrnn.py:435-439) and
 * w = 1 / (2 sigma2)  (uisrnn/uisrnn.py:414).
 */
int32_t uis_create(const uis_model_desc* desc, int32_t device, uis_handle** out);

void uis_destroy(uis_handle* h);

/*
 * Decode n_utt utterances (reference: UISRNN.predict over a list,
 * uisrnn/uisrnn.py:564-590; each one UISRNN.predict_single, :479-562).
 *   frames   : host, float32, [offsets[n_utt], D] row-major (utterance u owns rows
 *              offsets[u] .. offsets[u+1]); the caller has already cast the
 *              reference's float64 input to float32 (uisrnn.py:525-526)
 *   offsets  : host, int64, [n_utt + 1], offsets[0] == 0, non-decreasing
 *   labels_out : host, int32, [offsets[n_utt]]  -- predicted cluster id per frame
 *              (trace[-N:] of the best hypothesis, uisrnn.py:561)
 *   scores_out : host, float32, [n_utt] or NULL -- neg_likelihood of the best hypothesis
 *   stats    : optional
 * Empty utterances (N == 0) are allowed and produce no labels.
 * The frames travel to the device inside the call; where the one-launch kernels apply and the utterances are
 * equally long, the decode runs as a few launches with the later frames of every utterance copied and
 * projected behind the earlier launches (uis_stats.decode_launches; identical results).
 */
int32_t uis_decode(uis_handle* h, const float* frames, const int64_t* offsets,
                   int32_t n_utt, const uis_decode_opts* opts,
                   int32_t* labels_out, float* scores_out, uis_stats* stats);

/*
 * Same, taking the utterances the way the reference's predict() receives them
 * (uisrnn/uisrnn.py:564-590: a list of [N_u, D] float64 arrays, dtype checked at
 * :511-513) -- no packed float32 copy on the caller's side:
 *   utterances : host, [n_utt] pointers, utterances[u] -> float64 [n_frames[u], D]
 *                row-major, contiguous (may be NULL where n_frames[u] == 0)
 *   n_frames   : host, int64, [n_utt]
 * The cast to float32 (round to nearest even, what torch's .float() does at
 * :524-526) is done by the library: a few host threads fill a pinned staging
 * buffer chunk by chunk, each chunk's copy to the device and input projection
 * running while the next chunk is cast.  labels_out covers sum(n_frames) frames
 * in utterance order.
 */
int32_t uis_decode_f64(uis_handle* h, const double* const* utterances, const int64_t* n_frames,
                       int32_t n_utt, const uis_decode_opts* opts,
                       int32_t* labels_out, float* scores_out, uis_stats* stats);

/*
 * Same, with frames / labels_out / scores_out resident on the handle's device
 * (HBM pointers, e.g. torch tensor data_ptr()); offsets stays on the host.
 * No PCIe transfer of the frame stream happens inside the call.
 */
int32_t uis_decode_device(uis_handle* h, const float* d_frames, const int64_t* offsets,
                          int32_t n_utt, const uis_decode_opts* opts,
                          int32_t* d_labels_out, float* d_scores_out, uis_stats* stats);

/*
 * After a decode: per-utterance overflow flags (1 = hit the cluster cap) and
 * the full final beam scores, for the host-side retry loop and the parity tests.
 *   overflow_out : host int32 [n_utt] or NULL
 *   beam_scores_out : host float32 [n_utt * beam_size] or NULL (+inf padded)
 */
int32_t uis_last_decode_info(uis_handle* h, int32_t* overflow_out, float* beam_scores_out);

/*
 * The sizes uis_last_decode_info copies with: n_utt and beam_size of this handle's last
 * uis_decode* call (0 / 0 when that call was refused before it started, e.g. UIS_ERR_UNSUPPORTED
 * for its options -- a caller must size its buffers from here, not from what it asked for).
 */
int32_t uis_last_decode_shape(uis_handle* h, int32_t* n_utt_out, int32_t* beam_size_out);

/*
 * After a decode with UIS_FLAG_DEBUG_SCORES: the candidate scores of every window (look_ahead 1: of
 * every decode step),
 *   scores_out[(((w * n_utt + u) * beam_size + b) * C + c_1) * C ... + c_L],   C = max_clusters + 1
 * = what _calculate_score (uisrnn/uisrnn.py:455-477) returns for hypothesis b of utterance u in
 * window w (0 .. ceil(test_iteration * longest utterance / look_ahead) - 1) for the assignment
 * tuple (c_1 .. c_L), +inf where the reference's padded score_set (uisrnn.py:534-545) holds +inf:
 * clusters past K_b, hypotheses past the live beam, windows past the utterance's end, tuples
 * through a non-finite prefix.  A ragged last window of Lw < L frames has its scores at index 0 of
 * the missing dimensions.  `capacity` = floats available at scores_out; returns
 * UIS_ERR_INVALID_ARG when the last decode kept none or the buffer is too small.
 */
int32_t uis_debug_scores(uis_handle* h, float* scores_out, int64_t capacity);

/*
 * Read back the per-model constants computed at uis_create with the decode
 * kernels: m0 [D] and h1 [depth * H], (m0, h1) = CoreRNN(0, rnn_init_hidden)
 * (uisrnn/uisrnn.py:435-439).  Host pointers; either may be NULL.
 */
int32_t uis_model_constants(uis_handle* h, float* m0_out, float* h1_out);

/*
 * One CoreRNN.forward for seq_len = 1, batch = 1 (uisrnn/uisrnn.py:45-52) through the
 * decode kernels: x [D], h_in [depth * H] -> mean_out [D], h_out [depth * H] (host
 * pointers).  Unit-level entry point for parity tests; the decode does not call it.
 */
int32_t uis_rnn_step(uis_handle* h, const float* x, const float* h_in, float* mean_out, float* h_out);

/*
 * Online decoding (the caller side of the path: UIS-RNN is an online model; the reference
 * only offers offline predict(), uisrnn/uisrnn.py:479-590, with the test_iteration replay).
 * A session keeps the beam, the cluster states and the back-pointers of n_utt utterances on
 * the device.  Semantics = predict_single with test_iteration 1 and look_ahead 1: whatever the
 * chunking, the labels and scores equal those of one uis_decode over the whole utterances
 * (bit for bit).  One session per handle; uis_decode is refused while it is open.
 *
 *   uis_stream_begin  opts->test_iteration and look_ahead must be 1; max_frames = the most frames
 *                     any utterance will receive in this session (4 * beam_size bytes each).
 *                     opts->flags: UIS_FLAG_PERSISTENT keeps the decode kernel on the device
 *                     between pushes (lowest latency, occupies the whole device; UIS_ERR_UNSUPPORTED
 *                     where the session's shape does not allow it)
 *   uis_stream_push   frames: host float32, the new frames of utterance 0, then 1, ...;
 *                     counts[u] >= 0 = how many of them belong to utterance u (0 is fine)
 *   uis_stream_labels labels_out: host int32, for every utterance all frames received so far
 *                     (packed in utterance order) under the currently best hypothesis -- earlier
 *                     labels may still change with later frames, as in any beam search;
 *                     scores_out [n_utt] / overflow_out [n_utt] or NULL.  Returns
 *                     UIS_ERR_CLUSTER_CAP (labels still written) if a cap was hit.
 *   uis_stream_end    frees the session
 */
int32_t uis_stream_begin(uis_handle* h, int32_t n_utt, const uis_decode_opts* opts, int64_t max_frames);
int32_t uis_stream_push(uis_handle* h, const float* frames, const int32_t* counts);
int32_t uis_stream_labels(uis_handle* h, int32_t* labels_out, float* scores_out, int32_t* overflow_out);
int32_t uis_stream_end(uis_handle* h);

/*
 * Sequence-match accuracy on the device -- the step after predict() in the reference's demo
 * (demo.py:61-66; uisrnn/evals.py:40-73: confusion matrix + scipy linear_sum_assignment).
 * For every utterance u (labels offsets[u] .. offsets[u+1] of both sequences) matched_out[u]
 * = the number of positions that agree under the best one-to-one mapping between the two
 * label sets; accuracy = matched_out[u] / length in float64 (evals.py:72), left to the caller.
 * Labels are int32 in [0, 65536) (map other id types to integers first, as evals.py:58-61
 * does), at most 64 distinct values per sequence -- otherwise UIS_ERR_UNSUPPORTED.  Empty
 * utterances give 0 (the reference raises ValueError there; the Python mirror does too).
 *   uis_eval_accuracy         both sequences on the host
 *   uis_eval_accuracy_device  both in HBM on the handle's device
 *   uis_eval_last_decode      sequence a = the labels of this handle's last successful
 *                             uis_decode, still resident in HBM; truth: host, same packing
 * offsets / matched_out are host pointers.
 */
int32_t uis_eval_accuracy(uis_handle* h, const int32_t* labels_a, const int32_t* labels_b,
                          const int64_t* offsets, int32_t n_utt, int64_t* matched_out);
int32_t uis_eval_accuracy_device(uis_handle* h, const int32_t* d_labels_a, const int32_t* d_labels_b,
                                 const int64_t* offsets, int32_t n_utt, int64_t* matched_out);
int32_t uis_eval_last_decode(uis_handle* h, const int32_t* truth, int32_t n_utt, int64_t* matched_out);

/*
 * Pinned (page-locked) host memory for the frames / labels handed to uis_decode: with it the
 * H2D copy of the frame stream is asynchronous and overlaps the input projection of the chunks
 * already on the device.  Pageable memory works too (staged by the runtime).
 */
int32_t uis_host_alloc(size_t bytes, void** out);
void uis_host_free(void* p);

const char* uis_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* UISRNN_HIP_H_ */
