// uis_decoder.hip -- host side of libuisrnn_hip.so: the C ABI of include/uisrnn_hip.h.
//
// Replaces, for the decode path only, what the reference does in
//   UISRNN.__init__/load  (weights in)            uisrnn/uisrnn.py:83-107,149-170
//   UISRNN.predict / predict_single (beam search) uisrnn/uisrnn.py:479-590
// Utterances advance in lock-step: per step one select launch (one workgroup
// per utterance) and one batched CoreRNN evaluation over the surviving
// hypotheses of ALL utterances (GRU GEMM, mean-head GEMMs).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>
// (host pass only, x86 only: the float64 -> float32 cast below writes around the cache with SSE2 streaming
// stores; any other host takes the scalar loop, same rounding)
#if !defined(__HIP_DEVICE_COMPILE__) && (defined(__SSE2__) || defined(__x86_64__))
#define UIS_HOST_SSE2 1
#include <emmintrin.h>
#endif

#include "uis_kernels.hip"
#include "uis_eval.hip"
#include "uisrnn_hip.h"

#define UIS_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define HIPCHK(expr)                                                                   \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess)                                                              \
      return fail(e_ == hipErrorOutOfMemory ? UIS_ERR_OOM : UIS_ERR_HIP,               \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                  \
  } while (0)

int round_up(int v, int m) { return (v + m - 1) / m * m; }

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  bool borrowed = false;  // a view into the handle's workspace arena, not an allocation of its own
  int ensure(size_t bytes) {
    if (borrowed) { p = nullptr; cap = 0; borrowed = false; }
    if (bytes <= cap) return UIS_OK;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
      p = nullptr;
      return fail(UIS_ERR_OOM, "hipMalloc of " + std::to_string(want) + " bytes failed: " + hipGetErrorString(e));
    }
    cap = want;
    return UIS_OK;
  }
  void release() { if (p && !borrowed) (void)hipFree(p); p = nullptr; cap = 0; borrowed = false; }
  template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

// Device allocations of a one-off call (uis_rnn_step, the bootstrap of uis_create): freed on every
// return path.
struct Scratch {
  std::vector<void*> ptrs;
  ~Scratch() { for (void* p : ptrs) (void)hipFree(p); }
  template <typename T> int get(T** out, size_t count) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T));
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? UIS_ERR_OOM : UIS_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
    ptrs.push_back(p);
    *out = static_cast<T*>(p);
    return UIS_OK;
  }
};

#define UIS_WIDE_TILE_ROWS 2048   // row capacity (about twice the rows actually run, after dedup) above which the 2x2 tiles win
#define UIS_WT_ROWS 1280          // row capacity above which the LDS-weight kernels (k_wt_*) take over at hidden size 256 / 512
#define UIS_MAX_GROUPS 8
#define UIS_MAX_CLUSTERS 16         // clusters of 32 CUs the one-launch decode can address
#define UIS_LEVEL_CAP 32768        // hypotheses per intermediate look-ahead level and utterance: the DEFAULT (uis_decode_opts.level_cap)
#define UIS_LEVEL_CAP_MAX 524287    // ... and the most a caller may ask for (window_body packs a level's hypothesis index into 19 bits)
#define UIS_WINDOW_WIDE_LEVEL 256  // level capacity (hypotheses) from which k_window runs with more threads per utterance
#define UIS_WINDOW_WIDE_NT 512   // (1024 measured the same)
#define UIS_GRAPH_STEPS 32   // decode steps per captured graph (even)
#define UIS_STREAM_RESIDENT_MIN_STEPS 4  // uis_stream_push: steps per push from which the one-launch kernel is used
#ifndef UIS_H2D_CHUNKS
#define UIS_H2D_CHUNKS 2     // uis_decode: host frames are copied in this many pieces, overlapped with the input projection (1 / 2 / 3 / 4 pieces measured: 1.505 / 1.517 / 1.424 / 1.43 M frames/s from pinned buffers at configs[1])
#endif
#ifndef UIS_F64_PIECES
#define UIS_F64_PIECES 4     // uis_decode_f64: copies per projection chunk (the cast runs ahead of them)
#endif
#ifndef UIS_H2D_MIN_FRAMES
#define UIS_H2D_MIN_FRAMES 4096  // ... of at least this many frames each
#endif

struct GraphCache {
  hipGraphExec_t exec = nullptr;
  DecodeState st{};
  size_t lds = 0;
};

struct ProfileEvents {
  std::vector<hipEvent_t> ev;   // pairs
  std::vector<int> cls;
  size_t used = 0;
};

}  // namespace

struct uis_handle {
  int device = 0;
  hipStream_t stream = nullptr, copy_stream = nullptr;
  std::vector<hipEvent_t> h2d_done;
  DevModel m{};
  std::vector<void*> model_allocs;
  double alpha = 1.0;
  int n_cu = 0;  // compute units of the device
  int hid_map_seg = 0, hid_map_seg_p = 0, H_model = 0;  // (round 6) HidMap of this model's hidden axis; rnn_hidden_size as the caller gave it
  // the one-launch decode relies on observed, not promised, placement (workgroup b on XCD b % 8,
  // all 256 workgroups resident); when its own checks fail once, this handle stops using it
  bool inlaunch_failed = false, resident_off = false;
  // Where the one-launch decode's control words (barrier and row counters of the clusters) sit
  // inside their buffer.  The kernel runs ~5 % faster or slower depending on address bit 13 / 20 of
  // that one 4 KB block (L2 channel hashing; which value is the good one depends on the physical
  // pages: tools/experiments/README.md), so the first decodes of a shape try the four placements
  // and the handle keeps the fastest.
  struct CtlTune {
    uint64_t sig = 0;     // shape the measurements belong to
    int phase = 0;        // 0: first (cold) decode of the shape; 1..4: trying placement phase-1; 5: decided
    int best = 0;
    float ms[4] = {0, 0, 0, 0};
  } ctl_tune;
  // streaming session (uis_stream_*): owns its device memory
  struct Stream {
    bool active = false;
    int U = 0, B = 0, Kmax = 0, S = 0;
    int64_t cap = 0;                  // frames per utterance the session can hold
    std::vector<int32_t> have;        // frames received per utterance
    std::vector<void*> allocs;
    DevBuf chunk_x, chunk_pad, chunk_gi0, chunk_mse0, labels, scores;
    // one push = one H2D copy: [foff U x int64][avail U x int32, padded][frames] staged in pinned
    // host memory (h_stage) and mirrored in chunk_x
    void* h_stage = nullptr;
    size_t h_stage_cap = 0;
    DecodeState st{};
    int32_t* d_avail = nullptr;
    int32_t* d_have = nullptr;        // frames received per utterance, refreshed for every back-trace
    int64_t* d_foff = nullptr;
    int64_t* d_lab_off = nullptr;
    float* d_beam_scores = nullptr;
    int64_t steps_run = 0;
    // one-launch steps (k_decode_resident per push) where the shape allows it
    bool resident = false;
    bool coop_checked = false;        // one push of this session already went through the cooperative launch
    uint32_t* d_ctl = nullptr;
    size_t ctl_words = 0;
    // persistent launch (UIS_FLAG_PERSISTENT): k_decode_resident<.., true> stays on the device between
    // pushes; commands, tables, frames and labels travel through pm_block (host-coherent pinned memory)
    bool persist = false;             // the session asked for it and its shape allows it
    bool pm_running = false;          // the launch is on the device
    unsigned char* pm_block = nullptr;
    uint32_t pm_seq = 0;              // commands issued to the running launch
    int64_t pm_cap_frames = 0;        // rows of the mailbox's frame area = ncl * pm_cluster_rows
    int64_t pm_cluster_rows = 0;      // each cluster's FIXED share of the chunk buffers (rows), see uis_stream_begin
    size_t pm_o_foff = 0, pm_o_avail = 0, pm_o_laboff = 0, pm_o_frames = 0, pm_o_labels = 0, pm_o_scores = 0,
           pm_o_bscores = 0, pm_o_overflow = 0;
    unsigned long long* d_go = nullptr;
    unsigned char* d_hdr = nullptr;
    PersistArgs pm_args{};            // host copy of what d_pm_args holds
    PersistArgs* d_pm_args = nullptr;
    size_t hdr_stride = 0;
    int64_t pm_launches = 0, pm_commands = 0;
  } stream_state;
  // workspace (grow only)
  DevBuf off, utt_step, overflow, xpad, gi0, mse0, logblk, logden, pool_mean, pool_hid, pool_cnt;
  DevBuf beam_n, beam_K, beam_last, beam_sum, beam_score, beam_slot, beam_blk, bp, rows, nrows;
  DevBuf gi_up, a1, counters, beam_scores_out, io_frames, io_labels, io_scores, mse_tab, dbg_scores, utt_nrows, hst;
  size_t dbg_floats = 0;  // what the last decode left in dbg_scores (UIS_FLAG_DEBUG_SCORES)
  DevBuf lv_n, lv_K, lv_last, lv_sum, lv_score, lv_origin, lv_path, lv_slot, lv_blk, scratch, bp16, bp_base, cluster_ctl, resume, split_tab, scatter_tab, stage;
  DevBuf arena;  // one allocation behind all of the above: the per-step tables share pages (TLB reach)
  // uis_decode_f64: the caller's float64 utterances (set for the duration of that call) and the
  // pinned float32 staging buffer they are cast into, chunk by chunk, ahead of each H2D copy
  const double* const* src64 = nullptr;
  float* h_cast = nullptr;
  size_t h_cast_cap = 0;
  void* cast_pool = nullptr;  // CastPool: the threads that cast (created by the first uis_decode_f64)
  void* h_out = nullptr;      // pinned landing block of uis_decode_f64's labels and scores
  size_t h_out_cap = 0;
  ProfileEvents prof;
  hipEvent_t ev_begin = nullptr, ev_end = nullptr, ev_pre = nullptr;
  // utterance groups: one stream + one cached step graph each
  std::vector<hipStream_t> gstreams;
  std::vector<hipEvent_t> gdone;
  std::vector<GraphCache> gcache;
  // info of the last decode
  int last_U = 0, last_B = 0;
  std::vector<int64_t> io_offsets;  // offsets of the last uis_decode (its labels are still in io_labels)
  DevBuf ev_a, ev_b, ev_off, ev_out;  // uis_eval_* staging
  std::vector<int32_t> last_overflow;
  std::vector<float> last_beam_scores;
};

namespace {

inline volatile uint32_t* pm_ctl(uis_handle::Stream& ss) { return reinterpret_cast<volatile uint32_t*>(ss.pm_block); }

// every cluster's doorbell: command words first, then the sequence numbers
void pm_ring(uis_handle::Stream& ss, uint32_t seq, uint32_t type, uint32_t frames, const uint32_t* row0 = nullptr,
             const uint32_t* nrow = nullptr) {
  volatile uint32_t* ctl = pm_ctl(ss);
  const int ncl = ss.st.ncl;
  for (int c = 0; c < ncl; ++c) {
    ctl[UIS_PM_BELL_WORD + 16 * c + 1] = type | (frames << 8);
    ctl[UIS_PM_BELL_WORD + 16 * c + 2] = row0 ? (row0[c] | (nrow[c] << 16)) : 0u;
    ctl[UIS_PM_BELL_WORD + 16 * c + 3] = seq;
  }
  std::atomic_thread_fence(std::memory_order_release);
  for (int c = 0; c < ncl; ++c) ctl[UIS_PM_BELL_WORD + 16 * c] = seq;
  std::atomic_thread_fence(std::memory_order_seq_cst);
}

// Where hidden unit j of the model sits in the padded hidden vector (round 6).  Padding a hidden size up to a kernel's
// shape is exact only if every canonical K segment (uis_numerics.h: q = ceil(blocks / 8) k-blocks each) keeps its
// blocks: appending zeros does that where q is the kernel's (q 1 -> 128, 2 -> 256, 4 -> 512), and for q = 3 (hidden sizes
// 257 .. 384) the zeros go INSIDE: segment s's three blocks into the first three of the kernel's four
// (unit j -> (j / 48) * 64 + j % 48), a zero block behind each.  fma(0, 0, acc) = acc: no bit moves, every sum keeps
// its association, and the units in between stay exactly 0 (zero weights and biases: gates 1/2, candidate 0, h' = h / 2 = 0).
struct HidMap {
  int seg = 0, seg_p = 0;  // floats per canonical segment of the model / of the kernel's shape (0: identity)
  int operator()(int j) const { return seg ? (j / seg) * seg_p + j % seg : j; }
};

// Re-pack a (n_out x K) row-major matrix (optionally 3 stacked gates of `rows_per_gate`
// rows each, padded to `rows_per_gate_p`) into MFMA tile order:
//   out[((tile*nKb + kb)*64 + lane)*4 + r] = W[tile*16 + (lane&15)][kb*16 + 4*(lane>>4) + r]
// rmap / kmap: where a row (within its gate) / a column goes in the padded layout (the hidden axis: HidMap).
std::vector<float> tile_weights(const float* W, int gates, int rows_per_gate, int rows_per_gate_p, int K, int Kp,
                                HidMap rmap = HidMap(), HidMap kmap = HidMap()) {
  const int Fp = gates * rows_per_gate_p;
  const int nKb = Kp / 16;
  std::vector<float> out((size_t)Fp * Kp, 0.0f);
  for (int g = 0; g < gates; ++g)
    for (int j = 0; j < rows_per_gate; ++j) {
      const int fp = g * rows_per_gate_p + rmap(j), tile = fp / 16;
      const float* src = W + (size_t)(g * rows_per_gate + j) * K;
      for (int k = 0; k < K; ++k) {
        const int kp = kmap(k), kb = kp / 16, q = (kp % 16) / 4, r = kp % 4;
        out[(((size_t)tile * nKb + kb) * 64 + (q * 16 + fp % 16)) * 4 + r] = src[k];
      }
    }
  return out;
}

std::vector<float> pad_bias(const float* b, int gates, int n, int np, HidMap map = HidMap()) {
  std::vector<float> out((size_t)gates * np, 0.0f);
  for (int g = 0; g < gates; ++g)
    for (int j = 0; j < n; ++j) out[(size_t)g * np + map(j)] = b[(size_t)g * n + j];
  return out;
}

int upload(uis_handle* h, const std::vector<float>& v, const float** dst) {
  void* p = nullptr;
  HIPCHK(hipMalloc(&p, std::max<size_t>(v.size(), 4) * sizeof(float)));
  h->model_allocs.push_back(p);
  HIPCHK(hipMemcpy(p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
  *dst = reinterpret_cast<const float*>(p);
  return UIS_OK;
}

#define UIS_PROJ_WIDE_ROWS 2048  // frames from which the input projection runs its 2 x 4-tile-per-wave kernel
inline dim3 dense_grid(long rows, int tiles) { return dim3((unsigned)((rows + 15) / 16), (unsigned)((tiles + 3) / 4), 1); }
// 1-D, XCD-aware grid of the per-step dense kernels (dense_block_map in uis_kernels.hip)
inline dim3 dense_grid_xcd(long rows, int tiles) {
  return dim3((unsigned)dense_grid_blocks((int)((rows + 15) / 16), tiles), 1, 1);
}

// Launches go through here.  With UIS_FLAG_PROFILE every kernel is launched with
// hipExtLaunchKernelGGL's start/stop events, which carry the dispatch's own begin/end
// timestamps (what rocprofv3 --kernel-trace reports), not host-side bracket times.
struct Launcher {
  uis_handle* h;
  hipStream_t stream;
  bool profile;
  int events(hipEvent_t* a, hipEvent_t* b, int cls) {
    ProfileEvents& p = h->prof;
    if (p.used + 2 > p.ev.size()) {
      for (int i = 0; i < 2; ++i) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        p.ev.push_back(e);
      }
    }
    p.cls.push_back(cls);
    *a = p.ev[p.used];
    *b = p.ev[p.used + 1];
    p.used += 2;
    return UIS_OK;
  }
  template <typename... KArgs, typename... Args>
  int run(int cls, void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shmem, Args... args) {
    if (profile) {
      hipEvent_t a, b;
      int rc = events(&a, &b, cls);
      if (rc) return rc;
      hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, a, b, 0, args...);
    } else {
      hipLaunchKernelGGL(kernel, grid, block, shmem, stream, args...);
    }
    HIPCHK(hipGetLastError());
    return UIS_OK;
  }
  // The one-launch decode spins on in-launch barriers: every workgroup of the grid must be
  // resident at once.  hipLaunchCooperativeKernel guarantees that (or refuses the launch); the
  // occupancy query is checked as well so that the refusal has a readable reason.  With
  // `profile`, events recorded around the launch on the same (otherwise idle) stream.
  // `cooperative` = false: a plain launch of the same grid after the occupancy check (same
  // residency, 15-19 us less host time per launch: MI355X_MICROARCH.md "coop-launch"); used by the
  // streaming pushes after the session's first push went through the cooperative path.
  int run_cooperative(int cls, void (*kernel)(DevModel, DecodeState), int n_cu, dim3 grid, dim3 block, size_t shmem,
                      DevModel m, DecodeState st, bool cooperative = true) {
    int per_cu = 0;
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kernel), (int)block.x, shmem));
    if ((long)per_cu * n_cu < (long)grid.x) {
      h->inlaunch_failed = true;
      return fail(UIS_ERR_HIP, "one-launch decode: " + std::to_string(grid.x) + " workgroups cannot be co-resident (" +
                                   std::to_string(per_cu) + " per CU x " + std::to_string(n_cu) + " CUs)");
    }
    hipEvent_t a = nullptr, b = nullptr;
    if (profile) {
      int rc = events(&a, &b, cls);
      if (rc) return rc;
      HIPCHK(hipEventRecord(a, stream));
    }
    void* argv[2] = {&m, &st};
    hipError_t e;
    if (cooperative) {
      e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(kernel), grid, block, argv, (unsigned)shmem, stream);
    } else {
      hipLaunchKernelGGL(kernel, grid, block, shmem, stream, m, st);
      e = hipGetLastError();
    }
    if (e != hipSuccess) {
      (void)hipGetLastError();
      h->inlaunch_failed = true;  // the caller falls back to the launch-per-step path
      return fail(UIS_ERR_HIP, std::string("hipLaunchCooperativeKernel(k_decode_resident): ") + hipGetErrorString(e));
    }
    if (profile) HIPCHK(hipEventRecord(b, stream));
    return UIS_OK;
  }
};

#define LAUNCH(...)                       \
  do {                                    \
    int rc_ = lch.run(__VA_ARGS__);       \
    if (rc_) return rc_;                  \
  } while (0)

// One batched CoreRNN evaluation over the rows emitted for step parity `par`.
// Which family of dense kernels the launch-per-step path runs for a step of at most `max_rows`
// rnn rows (UIS_DF_*; uis_stats.decode_kernel reports it).
int dense_family(const uis_handle* h, uint32_t flags, long max_rows) {
  const DevModel& m = h->m;
  if (max_rows > UIS_WT_ROWS && (m.Hp == 512 || m.Hp == 256) && m.Dp % 16 == 0 && !(flags & UIS_FLAG_SMALL_TILES) && h->n_cu >= 64)
    return UIS_DF_WT;
  if (max_rows > UIS_WIDE_TILE_ROWS && (m.Hp / 16) % 4 == 0 && (m.Dp / 16) % 4 == 0 && !(flags & UIS_FLAG_SMALL_TILES))
    return UIS_DF_BIG;
  return UIS_DF_DENSE;
}

int launch_rnn(uis_handle* h, Launcher& lch, const DecodeState& st, int par, long max_rows) {
  const DevModel& m = h->m;
  const int mr = (int)max_rows;
  const bool wide = max_rows > UIS_WIDE_TILE_ROWS;  // tile shape, see uis_kernels.hip
  const int family = dense_family(h, st.flags, max_rows);
  // thousands of rows: the big-tile kernels (4 row tiles x several feature tiles per workgroup,
  // full-K chains per wave) where the feature-tile counts divide
  // thousands of rows and hidden size 256 / 512: weights in LDS, a wave per row tile (k_wt_*)
  // (measured crossover against the 1x1 split-K tiles: row capacity 1280 about equal, 1920 +11 %)
  if (family == UIS_DF_WT) {
    const int nft = m.Hp / 16, nft2 = m.Dp / 16;
    const int ng1 = wt_groups(h->n_cu, nft);
    const size_t kb_bytes = (size_t)nft * 1024;  // one weight stream of a feature tile: all k-blocks
    for (int l = 0; l < m.depth; ++l) {
      // (layers >= 1: the input-side gates by the same kernel, W_ih in the LDS slot; UIS_WT_NO_UPPER=1 keeps the split-K tiles)
      static const bool wt_upper = getenv("UIS_WT_NO_UPPER") == nullptr;
      if (l > 0 && !wt_upper) LAUNCH(UIS_K_UPPER_IN, k_dense_upper_in, dim3(step_grid_blocks(mr, m.G / 16, 1, 1)), dim3(512), 0, m, st, par, l);
      if (l > 0 && wt_upper && m.Hp == 512) LAUNCH(UIS_K_UPPER_IN, (k_wt_gru<32, true>), dim3(nft * ng1), dim3(512), 3 * kb_bytes, m, st, par, l, ng1);
      if (l > 0 && wt_upper && m.Hp != 512) LAUNCH(UIS_K_UPPER_IN, (k_wt_gru<16, true>), dim3(nft * ng1), dim3(512), 3 * kb_bytes, m, st, par, l, ng1);
      if (m.Hp == 512) LAUNCH(UIS_K_GRU, k_wt_gru<32>, dim3(nft * ng1), dim3(512), 3 * kb_bytes, m, st, par, l, ng1);
      else LAUNCH(UIS_K_GRU, k_wt_gru<16>, dim3(nft * ng1), dim3(512), 3 * kb_bytes, m, st, par, l, ng1);
    }
    // two feature tiles per workgroup of the heads where the tile count is even (measured on
    // configs[2]: one 66.4 / 36.7 ms per pass for the two heads, two 49.0 / 30.9, four 55.1 / 39.4)
    const int na1 = nft % 2 == 0 ? 2 : 1, na2 = nft2 % 2 == 0 ? 2 : 1;
    const int nh1 = wt_groups(h->n_cu, nft / na1), nh2 = wt_groups(h->n_cu, nft2 / na2);
#define UIS_WT_HEAD(NKB_, HEAD_, NA_, tiles_, ng_) \
  LAUNCH(HEAD_ == 1 ? UIS_K_HEAD1 : UIS_K_HEAD2, (k_wt_head<NKB_, HEAD_, NA_>), dim3((tiles_) / NA_ * (ng_)), dim3(512), NA_ * kb_bytes, m, st, par, ng_)
    if (m.Hp == 512) {
      if (na1 == 2) UIS_WT_HEAD(32, 1, 2, nft, nh1); else UIS_WT_HEAD(32, 1, 1, nft, nh1);
      if (na2 == 2) UIS_WT_HEAD(32, 2, 2, nft2, nh2); else UIS_WT_HEAD(32, 2, 1, nft2, nh2);
    } else {
      if (na1 == 2) UIS_WT_HEAD(16, 1, 2, nft, nh1); else UIS_WT_HEAD(16, 1, 1, nft, nh1);
      if (na2 == 2) UIS_WT_HEAD(16, 2, 2, nft2, nh2); else UIS_WT_HEAD(16, 2, 1, nft2, nh2);
    }
#undef UIS_WT_HEAD
    return UIS_OK;
  }
  if (family == UIS_DF_BIG) {
    for (int l = 0; l < m.depth; ++l) {
      if (l > 0) LAUNCH(UIS_K_UPPER_IN, k_dense_upper_in, dim3(step_grid_blocks(mr, m.G / 16, 1, 1)), dim3(512), 0, m, st, par, l);
      LAUNCH(UIS_K_GRU, k_big_gru<2>, dim3(big_grid_blocks(mr, m.Hp / 16, 2)), dim3(256), 0, m, st, par, l);
    }
    LAUNCH(UIS_K_HEAD1, k_big_head1<4>, dim3(big_grid_blocks(mr, m.Hp / 16, 4)), dim3(256), 0, m, st, par);
    LAUNCH(UIS_K_HEAD2, k_big_head2<4>, dim3(big_grid_blocks(mr, m.Dp / 16, 4)), dim3(256), 0, m, st, par);
    return UIS_OK;
  }
  for (int l = 0; l < m.depth; ++l) {
    if (l > 0) LAUNCH(UIS_K_UPPER_IN, k_dense_upper_in, dim3(step_grid_blocks(mr, m.G / 16, 1, 1)), dim3(512), 0, m, st, par, l);
    if (wide) LAUNCH(UIS_K_GRU, k_dense_gru<2>, dim3(step_grid_blocks(mr, m.Hp / 16, 2, 1)), dim3(512), 0, m, st, par, l);
    else LAUNCH(UIS_K_GRU, k_dense_gru<1>, dim3(step_grid_blocks(mr, m.Hp / 16, 1, 1)), dim3(512), 0, m, st, par, l);
  }
  if (wide) {
    LAUNCH(UIS_K_HEAD1, (k_dense_head1<2, 2>), dim3(step_grid_blocks(mr, m.Hp / 16, 2, 2)), dim3(512), 0, m, st, par);
    LAUNCH(UIS_K_HEAD2, k_dense_head2<2>, dim3(step_grid_blocks(mr, m.Dp / 16, 2, 1)), dim3(512), 0, m, st, par);
  } else {
    LAUNCH(UIS_K_HEAD1, (k_dense_head1<1, 1>), dim3(step_grid_blocks(mr, m.Hp / 16, 1, 1)), dim3(512), 0, m, st, par);
    LAUNCH(UIS_K_HEAD2, k_dense_head2<1>, dim3(step_grid_blocks(mr, m.Dp / 16, 1, 1)), dim3(512), 0, m, st, par);
  }
  return UIS_OK;
}

// One CoreRNN.forward (uisrnn.py:45-52) of a single row with the decode kernels themselves:
// x [Dp] and h_in [depth][Hp] on the device -> mean [Dp], h_out [depth][Hp] on the device.
int rnn_step_once(uis_handle* h, const float* d_x, const float* d_hin, float* d_mean, float* d_hout) {
  DevModel& m = h->m;
  Launcher lch{h, h->stream, false};
  float *d_gi0 = nullptr, *d_gi_up = nullptr, *d_a1 = nullptr;
  RnnRow* d_rows = nullptr;
  int32_t* d_nrows = nullptr;
  Scratch tmp;
  int rc;
  if ((rc = tmp.get(&d_gi0, (size_t)m.G)) || (rc = tmp.get(&d_gi_up, (size_t)64 * m.G)) ||
      (rc = tmp.get(&d_a1, (size_t)64 * m.Hp)) || (rc = tmp.get(&d_rows, 64)) || (rc = tmp.get(&d_nrows, 2)))
    return rc;
  HIPCHK(hipMemsetAsync(d_rows, 0, 64 * sizeof(RnnRow), h->stream));
  HIPCHK(hipMemsetAsync(d_a1, 0, 64 * m.Hp * sizeof(float), h->stream));
  RnnRow rr{};
  rr.utt = 0; rr.src = -1; rr.dst = 0; rr.nprev = 0; rr.frame = 0;
  int32_t nr[2] = {1, 1};
  HIPCHK(hipMemcpyAsync(d_rows, &rr, sizeof(rr), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(d_nrows, nr, sizeof(nr), hipMemcpyHostToDevice, h->stream));
  LAUNCH(UIS_K_INPUT_PROJ, k_dense_input_proj, dense_grid(1, m.G / 16), dim3(256), 0, m, d_x, d_gi0, 1L);
  DecodeState st{};
  st.U = 1; st.B = 1; st.Kmax = 1; st.S = 1; st.L = 1; st.tau = 1; st.max_rows = 1;
  st.gi0 = d_gi0; st.pool_mean = d_mean; st.pool_hid = d_hout; st.rows = d_rows; st.nrows = d_nrows;
  st.gi_up = d_gi_up; st.a1 = d_a1;
  const float* saved_h1 = m.h1;
  m.h1 = d_hin;  // a row with src = -1 reads its hidden state from "h1": point that at h_in
  rc = launch_rnn(h, lch, st, 0, 1);
  m.h1 = saved_h1;
  hipError_t se = hipStreamSynchronize(h->stream);  // before `tmp` frees what the kernels use
  if (rc) return rc;
  HIPCHK(se);
  return UIS_OK;
}

// (m0, h1) = CoreRNN(0, rnn_init_hidden)  (uisrnn.py:435-439).
int bootstrap_constants(uis_handle* h, const float* d_hinit) {
  DevModel& m = h->m;
  float *d_x = nullptr, *d_pm = nullptr, *d_ph = nullptr;
  const size_t hid_elems = (size_t)m.depth * m.Hp;
  Scratch tmp;
  int rc;
  if ((rc = tmp.get(&d_x, (size_t)m.Dp)) || (rc = tmp.get(&d_pm, (size_t)m.Dp)) || (rc = tmp.get(&d_ph, hid_elems))) return rc;
  HIPCHK(hipMemsetAsync(d_x, 0, m.Dp * sizeof(float), h->stream));
  if ((rc = rnn_step_once(h, d_x, d_hinit, d_pm, d_ph))) return rc;
  HIPCHK(hipMemcpy(const_cast<float*>(m.m0), d_pm, m.Dp * sizeof(float), hipMemcpyDeviceToDevice));
  HIPCHK(hipMemcpy(const_cast<float*>(m.h1), d_ph, hid_elems * sizeof(float), hipMemcpyDeviceToDevice));
  return UIS_OK;
}

// Everything one utterance group needs: a view of the shared buffers (pointers offset to
// the group's first utterance), its stream and its captured step graph.
struct GroupPlan {
  int u0 = 0, U = 0;
  int64_t maxT = 0;
  DecodeState st{};
};

// The kernels of `nsteps` consecutive decode steps (starting at an even step) on `stream`.
int enqueue_steps(uis_handle* h, Launcher& lch, const DecodeState& st, size_t select_lds, int nsteps) {
  const DevModel& m = h->m;
  const long max_rows = st.max_rows;
  for (int s = 0; s < nsteps; ++s) {
    const int par = s & 1;
    if (!st.wnd && select_fast_ok(st.B, st.Kmax, st.S) && !(st.flags & UIS_FLAG_GENERIC_SELECT))
      LAUNCH(UIS_K_SELECT, k_select_fast, dim3(st.U), dim3(256), (size_t)fast_lds_layout(m.Dp, st.B, st.Kmax, st.S).total, m, st, par);
    else if (!st.wnd) LAUNCH(UIS_K_SELECT, k_select, dim3(st.U), dim3(256), select_lds, m, st, par);
    else if (st.NC >= UIS_WINDOW_WIDE_LEVEL)  // hundreds of hypotheses per level: more threads per utterance
      LAUNCH(UIS_K_EXPAND, k_window<UIS_WINDOW_WIDE_NT>, dim3(st.U), dim3(UIS_WINDOW_WIDE_NT), window_lds_bytes(window_scratch_layout(st.S, st.NC, st.Kmax, st.B)), m, st, par);
    else LAUNCH(UIS_K_EXPAND, k_window<256>, dim3(st.U), dim3(256), window_lds_bytes(window_scratch_layout(st.S, st.NC, st.Kmax, st.B)), m, st, par);
    int rc = launch_rnn(h, lch, st, par, max_rows);
    if (rc) return rc;
  }
  return UIS_OK;
}

// float64 -> float32 of the packed frame matrix, read from the caller's per-utterance arrays (the
// reference casts once with torch's .float(), round to nearest even, uisrnn/uisrnn.py:524-526; so
// does a C++ double -> float conversion), by a few host threads -- one team for a whole decode:
// the threads are started once and take blocks of rows in order; the caller asks for a prefix of the
// rows (`wait_rows`), helping with blocks while it waits, and hands each finished piece to the copy
// engine while the team is already in the next.
static bool agent_flags_env() {  // UIS_AGENT_FLAGS=1: the conforming phase-word stores (UIS_FLAG_AGENT_FLAGS) for every decode of the process
  static const bool v = getenv("UIS_AGENT_FLAGS") != nullptr && atoi(getenv("UIS_AGENT_FLAGS")) != 0;
  return v;
}
static bool getenv_flag_no_stream() {  // UIS_CAST_PLAIN_STORES=1: the scalar loop with ordinary stores (A/B)
  static const bool v = getenv("UIS_CAST_PLAIN_STORES") != nullptr;
  return v;
}
struct CastTeam {
  static constexpr int64_t kBlockRows = 512;
  const double* const* utt; const int64_t* offsets; int n_utt, D; int64_t F; float* dst;
  int64_t nblocks = 0;
  std::atomic<int64_t> next{0};
  std::vector<std::atomic<unsigned char>> done;
  // (round 5) `order`, if given: the row ranges to cast, in this order, instead of the packed matrix front to back --
  // the first frames of EVERY utterance before the later ones, when a decode starts on a time slice
  std::vector<std::pair<int64_t, int64_t>> order;
  // ... and `dst_rows`, if given: where block b's first row goes in the staging block (a ragged list's slices are laid
  // out slice after slice, so that a slice is ONE copy); without it a row keeps its place in the packed matrix
  std::vector<int64_t> dst_rows;
  CastTeam(const double* const* utt_, const int64_t* offsets_, int n_utt_, int D_, int64_t F_, float* dst_,
           std::vector<std::pair<int64_t, int64_t>> order_ = {}, std::vector<int64_t> dst_rows_ = {})
      : utt(utt_), offsets(offsets_), n_utt(n_utt_), D(D_), F(F_), dst(dst_),
        nblocks(order_.empty() ? (F_ + kBlockRows - 1) / kBlockRows : (int64_t)order_.size()),
        done((size_t)(order_.empty() ? (F_ + kBlockRows - 1) / kBlockRows : (int64_t)order_.size())), order(std::move(order_)),
        dst_rows(std::move(dst_rows_)) {
    for (auto& d : done) d.store(0, std::memory_order_relaxed);
  }
  void wait_blocks(int64_t b1) {  // blocks [0, b1) of `order` are cast when this returns
    for (int64_t b = 0; b < b1; ++b)
      while (!done[(size_t)b].load(std::memory_order_acquire))
        if (!take()) std::this_thread::yield();
  }
  // how many threads are worth waking: >= 1 MB of input each
  unsigned want_threads(unsigned have) const {
    return (unsigned)std::min<int64_t>(have, std::max<int64_t>(1, F * D / (1 << 17)));
  }
  bool take() {  // one block, if there is one left
    const int64_t b = next.fetch_add(1, std::memory_order_relaxed);
    if (b >= nblocks) return false;
    if (order.empty()) cast_block(b * kBlockRows, std::min(F, (b + 1) * kBlockRows), -1);
    else cast_block(order[(size_t)b].first, order[(size_t)b].second, dst_rows.empty() ? -1 : dst_rows[(size_t)b]);
#if defined(UIS_HOST_SSE2)
    _mm_sfence();  // (the streaming stores above are ordered before the flag)
#endif
    done[(size_t)b].store(1, std::memory_order_release);
    return true;
  }
  void cast_block(int64_t r0, int64_t r1, int64_t drow) {  // (drow >= 0: row r0 goes to row drow of the staging block)
    int u = (int)(std::upper_bound(offsets, offsets + n_utt + 1, r0) - offsets) - 1;  // the utterance holding row r0
    for (int64_t r = r0; r < r1;) {
      while (offsets[u + 1] <= r) ++u;  // (empty utterances)
      const int64_t e = std::min(r1, offsets[u + 1]);
      const double* src = utt[u] + (size_t)(r - offsets[u]) * D;
      float* d = dst + (size_t)(drow >= 0 ? drow + (r - r0) : r) * D;
      const int64_t n = (e - r) * D;
      int64_t i = 0;
#if defined(UIS_HOST_SSE2)
      // four values per step, written around the cache (cvtpd2ps rounds like the scalar conversion: MXCSR,
      // to nearest even): the float32 block is read next by the copy engine, not by this core, and an
      // ordinary store would first fetch every destination line -- a third more DRAM traffic on the
      // NUMA node that bounds this loop
      if (!getenv_flag_no_stream()) {
        while (i < n && (reinterpret_cast<uintptr_t>(d + i) & 15u)) { d[i] = (float)src[i]; ++i; }
        for (; i + 4 <= n; i += 4) {
          const __m128 lo = _mm_cvtpd_ps(_mm_loadu_pd(src + i)), hi = _mm_cvtpd_ps(_mm_loadu_pd(src + i + 2));
          _mm_stream_ps(d + i, _mm_movelh_ps(lo, hi));
        }
      }
#endif
      for (; i < n; ++i) d[i] = (float)src[i];
      r = e;
    }
  }
  void wait_rows(int64_t f1) {  // rows [0, f1) are cast when this returns
    const int64_t b1 = (f1 + kBlockRows - 1) / kBlockRows;
    for (int64_t b = 0; b < b1; ++b)
      while (!done[(size_t)b].load(std::memory_order_acquire))
        if (!take()) std::this_thread::yield();
  }
};

// The threads that cast: created once per handle (a decode used to spawn and join up to sixteen
// std::threads of its own -- a third of a millisecond before the first block was cast), asleep on a
// condition variable between decodes.  They follow the affinity mask of the thread that created the
// pool (bench.py pins a rank to its share of the cores first).
struct CastPool {
  std::vector<std::thread> threads;
  std::mutex mu;
  std::condition_variable cv_work, cv_idle;
  CastTeam* job = nullptr;
  uint64_t generation = 0;
  unsigned wanted = 0, busy = 0;
  bool quit = false;
  void ensure_started() {
    if (!threads.empty() || quit) return;
    // (measured on the 256-core GPU box, configs[1], frames/s of the float64 leg: 4 / 8 threads 1.600 M, 16 1.56-1.59 M,
    // 24 1.59 M, 32 1.58 M, 48 1.59 M -- the cast is a few hundred microseconds of memory traffic; more threads only
    // add wake-up jitter: profiles/r04_f64_leg.txt)
    unsigned nt = std::min(std::max(1u, std::thread::hardware_concurrency()), 8u);
    if (const char* e = getenv("UIS_CAST_THREADS")) nt = (unsigned)std::max(1, atoi(e));
    try {
      for (unsigned k = 1; k < nt; ++k) threads.emplace_back([this, k]() { worker(k); });
    } catch (...) {  // no more threads to be had: the caller's thread does what is left (CastTeam::wait_rows)
    }
  }
  void worker(unsigned index) {
    uint64_t seen = 0;
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      cv_work.wait(lk, [&]() { return quit || generation != seen; });
      if (quit) return;
      seen = generation;
      CastTeam* j = job;
      if (!j || index > wanted) continue;
      ++busy;
      lk.unlock();
      while (j->take()) {}
      lk.lock();
      if (--busy == 0) cv_idle.notify_all();
    }
  }
  void post(CastTeam* j) {
    ensure_started();
    std::lock_guard<std::mutex> lk(mu);
    job = j;
    wanted = j->want_threads((unsigned)threads.size() + 1) - 1;  // (the caller's thread is one of the team)
    ++generation;
    cv_work.notify_all();
  }
  void finish() {  // the job is about to go out of scope: nobody may still be inside it
    std::unique_lock<std::mutex> lk(mu);
    job = nullptr;
    cv_idle.wait(lk, [&]() { return busy == 0; });
  }
  ~CastPool() {
    {
      std::lock_guard<std::mutex> lk(mu);
      quit = true;
      job = nullptr;
    }
    cv_work.notify_all();
    for (auto& th : threads) th.join();
  }
};

int decode_once(uis_handle* h, const float* d_frames, const int64_t* offsets, int32_t n_utt,
                const uis_decode_opts* opts, int32_t* d_labels, float* d_scores, uis_stats* stats,
                const float* h_frames = nullptr) {
  if (!h || !offsets || !opts || n_utt < 0) return fail(UIS_ERR_INVALID_ARG, "null handle/offsets/opts or negative n_utt");
  // (whatever refuses this decode below: uis_last_decode_info must not hand out the PREVIOUS decode's arrays)
  h->last_U = 0; h->last_B = 0;
  h->last_overflow.clear(); h->last_beam_scores.clear();
  if (h->stream_state.active) return fail(UIS_ERR_INVALID_ARG, "a streaming session is open on this handle (uis_stream_end first)");
  const DevModel& m = h->m;
  const int B = opts->beam_size, L = opts->look_ahead, tau = opts->test_iteration;
  int Kmax = opts->max_clusters > 0 ? opts->max_clusters : 16;
  // (round 5: no option value the reference takes is refused for its size any more -- a beam beyond the select
  // kernels' 256, a cluster cap beyond their LDS budget and any look_ahead go through the window machinery, a launch
  // per sub-step with the candidate lists in HBM; what is left are the widths of the window records' fields)
  if (B < 1 || B > 32767) return fail(UIS_ERR_UNSUPPORTED, "beam_size must be in [1, 32767]");
  if (L < 1 || tau < 1) return fail(UIS_ERR_INVALID_ARG, "look_ahead and test_iteration must be >= 1");
  if (L > UIS_MAX_LOOKAHEAD) return fail(UIS_ERR_UNSUPPORTED, "look_ahead must be <= 1024");
  if (Kmax > 4096) return fail(UIS_ERR_UNSUPPORTED, "max_clusters must be <= 4096");
  if (opts->level_cap < 0) return fail(UIS_ERR_INVALID_ARG, "level_cap must be >= 0");
  const int64_t level_cap = opts->level_cap > 0 ? std::min<int64_t>(opts->level_cap, UIS_LEVEL_CAP_MAX) : UIS_LEVEL_CAP;
  if (offsets[0] != 0) return fail(UIS_ERR_INVALID_ARG, "offsets[0] must be 0");
  int64_t maxN = 0;
  for (int u = 0; u < n_utt; ++u) {
    const int64_t n = offsets[u + 1] - offsets[u];
    if (n < 0) return fail(UIS_ERR_INVALID_ARG, "offsets must be non-decreasing");
    maxN = std::max(maxN, n);
  }
  const int64_t F = n_utt ? offsets[n_utt] : 0;
  bool ragged_list = false;  // (utterances of different lengths)
  for (int u = 1; u < n_utt; ++u) ragged_list = ragged_list || offsets[u + 1] - offsets[u] != offsets[1] - offsets[0];
  if (stats) memset(stats, 0, sizeof(*stats));
  h->last_U = n_utt; h->last_B = B;
  h->last_overflow.assign(n_utt, 0);
  h->last_beam_scores.assign((size_t)n_utt * B, INFINITY);
  if (n_utt == 0) return UIS_OK;
  if (F > 0 && (!d_frames || !d_labels)) return fail(UIS_ERR_INVALID_ARG, "frames/labels_out is null");
  const int64_t maxT = (int64_t)tau * maxN;
  if (maxT > 0x7fffff00LL) return fail(UIS_ERR_UNSUPPORTED, "test_iteration * N too large");
  HIPCHK(hipSetDevice(h->device));

  const int U = n_utt;
  // look_ahead >= 2: capacity of an intermediate level = every assignment of the window's first
  // j frames, N_j = B * prod_{i=1..j} (Kmax + i), capped; the slot pool holds the beam's states,
  // every level's new ones and the winners'.
  int64_t NC = B, S64 = (int64_t)B * Kmax + B;
  if (L > 1) {
    int64_t nj = B;
    NC = 0;
    for (int j2 = 1; j2 < L; ++j2) {
      nj = std::min<int64_t>(nj * (Kmax + j2), level_cap);
      NC = std::max(NC, nj);
      S64 += nj;
    }
  }
  if (S64 > 0x3fffffff) return fail(UIS_ERR_UNSUPPORTED, "beam_size * max_clusters ^ look_ahead too large");
  const int S = (int)S64;
  const bool profile = (opts->flags & UIS_FLAG_PROFILE) != 0;
  const bool use_graph = !profile && (opts->flags & UIS_FLAG_GRAPH) != 0;
  Launcher lch{h, h->stream, profile};
  h->prof.used = 0; h->prof.cls.clear();

  // wnd: the window machinery decodes -- look_ahead >= 2, and look_ahead 1 where the select kernels do not apply
  // (beam_size > 256, or tables beyond their LDS budget): k_window with a window of ONE frame is the prune
  // sub-step alone, its work arrays in LDS where they fit and in HBM where they do not
  SelectLds lds{};
  bool wnd = L > 1 || B > 256;
  if (!wnd) {
    lds = select_lds_layout(m.Dp, B, Kmax, S);
    if (lds.total > 160 * 1024) wnd = true;
  }
  // (a field-width limit that no smaller list cures is UNSUPPORTED -- the Python host halves a list on UIS_ERR_OOM, which
  // only helps the term that grows with the number of utterances)
  if (NC * (int64_t)(Kmax + 1) > 0x3fffffff)
    return fail(UIS_ERR_UNSUPPORTED, "level capacity * max_clusters beyond the kernels' 32-bit indices");
  if ((int64_t)U * std::max<int64_t>(NC, B) > 0x3fffffff)
    return fail(UIS_ERR_OOM, "utterances * level capacity beyond the kernels' 32-bit indices");
  const WindowScratch wsl = window_scratch_layout(S, (int)NC, Kmax, B);
  {  // refuse configurations whose state would not fit the device instead of failing in hipMalloc
    const double bytes = (double)U * S * (m.Dp + (double)m.depth * m.Hp) * 4.0 +
                         (wnd ? (double)U * (wsl.total + 2.0 * NC * (Kmax * 8.0 + 32.0) + NC * (m.Hp + m.G) * 4.0) : 0.0);
    // (UIS_MAX_STATE_BYTES: a smaller ceiling, for tests of the host layer's answer -- it decodes the list in halves)
    const char* lim = getenv("UIS_MAX_STATE_BYTES");
    if (bytes > (lim ? atof(lim) : 200e9))
      return fail(UIS_ERR_OOM, "decode state would need " + std::to_string((long long)(bytes / 1e9)) + " GB");
  }

  // ---- utterance groups: independent lock-step chains, one stream each.  Measured on
  // MI355X: the device overlaps at most ~2 of these small kernels, so more
  // groups mean more launches, not more throughput -- the default is one group.
  int G = opts->n_streams > 0 ? opts->n_streams : 1;
  if (profile || (opts->flags & UIS_FLAG_DEBUG_SCORES)) G = 1;
  G = std::max(1, std::min(std::min(G, UIS_MAX_GROUPS), U));
  while ((int)h->gstreams.size() < G) {
    hipStream_t sgrp;
    HIPCHK(hipStreamCreateWithFlags(&sgrp, hipStreamNonBlocking));
    h->gstreams.push_back(sgrp);
    hipEvent_t e;
    HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    h->gdone.push_back(e);
    h->gcache.emplace_back();
  }
  std::vector<GroupPlan> plan(G);
  for (int g = 0; g < G; ++g) {
    plan[g].u0 = (int)((int64_t)U * g / G);
    plan[g].U = (int)((int64_t)U * (g + 1) / G) - plan[g].u0;
    for (int u = plan[g].u0; u < plan[g].u0 + plan[g].U; ++u)
      plan[g].maxT = std::max<int64_t>(plan[g].maxT, (int64_t)tau * (offsets[u + 1] - offsets[u]));
  }
  const long max_rows = (long)U * (L == 1 ? B : (long)NC);
  // back-pointer records per utterance (look_ahead >= 2): windows x B
  std::vector<int64_t> bp_base(U + 1, 0);
  if (wnd)
    for (int u = 0; u < U; ++u)
      bp_base[u + 1] = bp_base[u] + (((int64_t)tau * (offsets[u + 1] - offsets[u]) + L - 1) / L) * B;

  // ---- workspace
  int rc;
  // every buffer is a 4 KB-aligned view into ONE allocation (h->arena), sized first
  std::vector<std::pair<DevBuf*, size_t>> want;
#define ENSURE(buf, bytes) want.emplace_back(&h->buf, (size_t)(bytes))
  ENSURE(off, (size_t)(U + 1) * 8);
  ENSURE(utt_step, (size_t)U * 4);
  ENSURE(overflow, (size_t)U * 4);
  if (m.D != m.Dp) ENSURE(xpad, (size_t)std::max<int64_t>(F, 1) * m.Dp * 4);
  ENSURE(gi0, (size_t)std::max<int64_t>(F, 1) * m.G * 4);
  ENSURE(mse0, (size_t)std::max<int64_t>(F, 1) * 4);
  // (k_decode_rs / k_decode_big<WS> copy the first UIS_RS_LOGTAB entries into LDS whatever the decode's length)
  const int64_t n_log = std::max<int64_t>(maxT + 2, UIS_RS_LOGTAB);
  ENSURE(logblk, (size_t)n_log * 8);
  ENSURE(logden, (size_t)n_log * 8);
  ENSURE(pool_mean, (size_t)U * S * m.Dp * 4);
  ENSURE(pool_hid, ((size_t)U * S + 1) * m.depth * m.Hp * 4);  // + the slot k_decode_resident keeps h1 in
  ENSURE(pool_cnt, (size_t)U * S * 4);
  ENSURE(beam_n, (size_t)2 * U * 4);
  ENSURE(beam_K, (size_t)2 * U * B * 4);
  ENSURE(beam_last, (size_t)2 * U * B * 4);
  ENSURE(beam_sum, (size_t)2 * U * B * 4);
  ENSURE(beam_score, (size_t)2 * U * B * 4);
  ENSURE(beam_slot, (size_t)2 * U * B * Kmax * 4);
  ENSURE(beam_blk, (size_t)2 * U * B * Kmax * 4);
  ENSURE(bp, !wnd ? (size_t)std::max<int64_t>(tau * F, 1) * B * 4 : 16);
  // k_decode_resident: the CUs form ncl clusters of 32 (one per XCD: 8 on a whole MI355X, 1 in
  // CPX mode); one row region per cluster, a multiple of 16 rows
  const int ncl = (h->n_cu >= 32 && h->n_cu % 32 == 0 && h->n_cu / 32 <= UIS_MAX_CLUSTERS) ? h->n_cu / 32 : 0;
  const int nclq = std::max(ncl, 1);
  // (rows an utterance can emit per step: beam_size, or a level's capacity inside a look-ahead window)
  // (+ 16 at look_ahead 1: k_decode_coh cuts a cluster's region into two cohorts, each rounded up to a row tile)
  const int rx_stride = (int)(((((long)U + nclq - 1) / nclq) * (L == 1 ? (long)B : (long)NC) + 15) / 16 * 16) + (L == 1 ? 16 : 0);
  const long rows_cap = std::max(max_rows + 48L * G, (long)nclq * rx_stride);  // every group's last row tile may run past its rows
  ENSURE(rows, (size_t)rows_cap * sizeof(RnnRow));
  ENSURE(nrows, (size_t)UIS_MAX_GROUPS * 2 * 4);
  // depth 1: k_decode_resident's h' staging buffer
  ENSURE(gi_up, m.depth > 1 ? (size_t)rows_cap * m.G * 4 : (size_t)rows_cap * m.Hp * 4);
  ENSURE(a1, (size_t)rows_cap * m.Hp * 4);
  // rnn_depth >= 2 in one launch (k_decode_deep): the cluster kernels' shapes, the select's LDS budget; the
  // two hand-off buffers a layer's h' goes through
  const bool deep_shape = m.depth >= 2 && G == 1 &&
                          ((m.Hp == 512 && (m.Dp == 128 || m.Dp == 256 || m.Dp == 512)) ||
                           (m.Hp == 256 && (m.Dp == 128 || m.Dp == 256)) || (m.Hp == 128 && (m.Dp == 128 || m.Dp == 256)));
  if (deep_shape) ENSURE(hst, (size_t)2 * rows_cap * m.Hp * 4);
#if defined(UIS_RESIDENT_TIMING)
  ENSURE(counters, (size_t)UIS_MAX_GROUPS * 4 * 8 + (96 + 1024) * 8);
#else
  ENSURE(counters, (size_t)UIS_MAX_GROUPS * 4 * 8 + 96 * 8);
#endif
  ENSURE(beam_scores_out, (size_t)U * B * 4);
  ENSURE(utt_nrows, (size_t)U * 2 * 4);
  // (a decode in several launches: DecodeState::resume -- only lists given in host memory can split, and never through
  // the window machinery: wide beams, large caps and look-ahead decodes do not pay for it)
  ENSURE(resume, (h_frames && !wnd) ? (size_t)U * (rs_lds_layout(B, Kmax, S).persist_stride + 4) + 16 : (size_t)16);
  ENSURE(split_tab, (size_t)8 * U * 2 * sizeof(long));
  ENSURE(scatter_tab, (size_t)64 * U * 3 * sizeof(long));                                  // (... and of its copy units: the scatter's tables)
  if (h->src64 && h_frames && F > 0 && ragged_list && (double)F * m.D * 4.0 >= (getenv("UIS_SPLIT_MIN_MB") ? 1e6 * atof(getenv("UIS_SPLIT_MIN_MB")) : 64e6))
    ENSURE(stage, (size_t)F * m.D * 4);                   // (... the device's copy of the time-major staging block)
                                     // (... of a ragged list: batch tables of up to 8 slices)
  // the whole decode in one launch with register-resident weights (k_decode_resident)
  const bool resident_ok = L == 1 && m.depth == 1 && (m.Hp == 128 || m.Hp == 256 || m.Hp == 512) &&
                           (m.Dp == 128 || m.Dp == 256 || m.Dp == 512) && G == 1 &&
                           select_fast_ok(B, Kmax, S) && !(opts->flags & UIS_FLAG_GENERIC_SELECT) && ncl >= 1 &&
                           ((double)U * S + 1) * m.Hp * 4.0 < 2.0e9 && (double)rows_cap * m.Hp * 4.0 < 2.0e9 &&
                           (double)U * S * m.Dp * 4.0 < 2.0e9 &&  // (the cluster means too are addressed through a 2 GB buffer descriptor)
                           resident_lds_bytes(m.Hp, m.Dp, B, Kmax, S) <= 160 * 1024;
  // the default wherever it applies; UIS_FLAG_STEPWISE (or any of the per-step experiments) keeps
  // the launch-per-step path, UIS_FLAG_RESIDENT turns "does not apply" into an error
  const bool resident = resident_ok && !use_graph && (!h->resident_off || (opts->flags & UIS_FLAG_RESIDENT)) &&
                        !(opts->flags & UIS_FLAG_STEPWISE);
  // ... and for SMALL models (small_model_ok: hidden size up to about 64, any rnn_depth -- the shapes of
  // the reference's own tests) the whole beam search of an utterance on ONE workgroup, one launch per decode
  // (k_decode_small); the default where the kernels above do not apply
  // (look_ahead >= 2: with a sub-step of the window kernel in the select's place, its work arrays in LDS)
  const bool small_shape = G == 1 && !use_graph && !(opts->flags & (UIS_FLAG_STEPWISE | UIS_FLAG_GENERIC_SELECT)) &&
                           small_model_ok(m.Hp, m.Dp, m.depth) && !getenv("UIS_NO_SMALL_KERNEL");
  // (a model of the cluster kernels' shapes -- hidden size 128 with a small observation dim also counts as
  // "small" -- goes to them: k_decode_big<WIN> below)
  const bool cluster_shape = m.depth == 1 && (m.Hp == 128 || m.Hp == 256 || m.Hp == 512) && (m.Dp == 128 || m.Dp == 256 || m.Dp == 512);
  // rnn_depth >= 2 at the cluster kernels' shapes: k_decode_big's stages with the weight slot refilled per stage
  // (look_ahead >= 2: with the window's sub-step as the select stage, for the shapes instantiated below)
  const bool deep_win_shape = (m.Hp == 512 && m.Dp == 256) || (m.Hp == 256 && (m.Dp == 128 || m.Dp == 256)) || (m.Hp == 128 && m.Dp == 128);
  const bool deep = deep_shape && !small_shape && !use_graph && ncl >= 1 &&
                    (L == 1 ? select_fast_ok(B, Kmax, S)
                            : deep_win_shape && big_win_lds_bytes(m.Hp, S, (int)NC, Kmax, B) <= 157 * 1024 && !getenv("UIS_NO_WINDOW_LAUNCH")) &&
                    !(opts->flags & (UIS_FLAG_STEPWISE | UIS_FLAG_GENERIC_SELECT)) &&
                    (!h->resident_off || (opts->flags & UIS_FLAG_RESIDENT)) &&
                    ((double)U * S + 1) * m.depth * m.Hp * 4.0 < 2.0e9 && (double)rows_cap * m.G * 4.0 < 2.0e9 &&
                    (double)U * S * m.Dp * 4.0 < 2.0e9 && (L > 1 || deep_lds_bytes(m.Hp, m.Dp, B, Kmax, S) <= 157 * 1024) &&
                    !getenv("UIS_NO_DEEP_KERNEL");  // (look_ahead >= 2: the window's layout, checked above, not the fast select's)
  const bool small = !resident_ok && small_shape &&
                     (L == 1 ? select_fast_ok(B, Kmax, S) && small_lds_bytes(m.Dp, B, Kmax, S) <= 160 * 1024
                             : !cluster_shape && wsl.total <= 128 * 1024 && (double)U * NC * std::max(m.G, m.Hp) * 4.0 < 2.0e9);
  // ... and look_ahead >= 2 in one launch (k_decode_big<WIN>: the window kernel's sub-step as the select stage
  // of the wave-per-row-tile decode)
  const bool win = !small && L > 1 && m.depth == 1 && G == 1 && !use_graph && ncl >= 1 &&
                   (U <= 32 * ncl || !getenv("UIS_WINDOW_LAUNCH_ONE_EACH")) &&
                   cluster_shape &&
                   !(opts->flags & UIS_FLAG_STEPWISE) && (!h->resident_off || (opts->flags & UIS_FLAG_RESIDENT)) &&
                   // (k_decode_big addresses its state through 4 GB descriptors with unsigned offsets since round 5; element
                   // counts stay below 2^31 for its int arithmetic)
                   ((double)U * S + 1) * m.Hp * 4.0 < 4.0e9 && (double)rows_cap * m.Hp * 4.0 < 4.0e9 &&
                   ((double)U * S + 1) * m.Hp < 2.0e9 && (double)U * S * m.Dp * 4.0 < 4.0e9 &&
                   big_win_lds_bytes(m.Hp, S, (int)NC, Kmax, B) <= 157 * 1024 &&
                   !getenv("UIS_NO_WINDOW_LAUNCH");
  if ((opts->flags & UIS_FLAG_RESIDENT) && !resident && !small && !win && !deep)
    return fail(UIS_ERR_UNSUPPORTED, "UIS_FLAG_RESIDENT needs (look_ahead 1:) one stream, beam_size * (max_clusters + 1) <= 256, no "
                                     "per-step path flag and either a small model (rnn_hidden_size up to about 64, any rnn_depth) "
                                     "or rnn_depth 1 with rnn_hidden_size 128, 256 or 512 (padded), observation_dim 128, "
                                     "256 or 512 (padded) and a device whose CU count is a multiple of 32");
  // control words: [0, 16) XCC id per cluster, [16] abort, [32, 32 + 32 ncl) row counters,
  // then 32 ncl barrier counters, then 32 ncl phase words (one 128-byte line per cluster each)
  const size_t ctl_words = (size_t)32 + 3 * UIS_MAX_CLUSTERS * 32;
  static const size_t ctl_place[4] = {0, 8192, (size_t)1 << 20, ((size_t)1 << 20) + 8192};
  ENSURE(cluster_ctl, ctl_place[3] + ((ctl_words * 4 + 4095) & ~(size_t)4095));
  // the one-launch decode with the REPLICATED select (k_decode_rs, uis_select_rs.hip): every workgroup of
  // an XCD decides all of the cluster's utterances, one wave each; the default where it applies.
  // Its instantiations (shape classes), in the order tried:
  //   RS_BASE   beam_size <= 16, <= 192 candidates, observation dim <= 256, at most 8 utterances per XCD
  //   RS_C1     ... with beam_size 10 / max_clusters 16 as compile-time constants (BASELINE configs[1])
  //   RS_UPW2   ... 9 .. 16 utterances per XCD: two utterances per wave
  //   RS_WIDE   beam_size <= 32, <= 256 candidates, observation dim 256 or 512 (configs[4]), at most 8 per XCD
  enum { RS_NONE = 0, RS_BASE, RS_C1, RS_UPW2, RS_WIDE, RS_UPW2_C1, RS_WIDE_C4 };
  int rs_kind = RS_NONE;
  if (resident && !(opts->flags & UIS_FLAG_OWNER_SELECT) && (UIS_RS_DEFAULT || (opts->flags & UIS_FLAG_REPLICATED_SELECT))) {
    const int per_xcd = (U + ncl - 1) / ncl;
    const bool base_shape = m.Dp <= 256 && rs_select_ok(B, Kmax, S, (long)maxT, 3) && F < 0x7fffffffLL;  // (k_decode_rs keeps frame numbers in 32 bits)
    if (base_shape && per_xcd <= UIS_RS_UTT && resident_rs_lds_bytes(m.Hp, m.Dp, B, Kmax, S) <= 160 * 1024)
      rs_kind = (m.Hp == 512 && m.Dp == 256 && m.H == 512 && m.D == 256 && B == 10 && Kmax == 16 && !getenv("UIS_RS_NO_C1") && !getenv("UIS_NO_SHAPE_CLASSES")) ? RS_C1 : RS_BASE;
    else if (base_shape && per_xcd <= 2 * UIS_RS_UTT && m.Hp == 512 && m.Dp == 256 &&
             ((UIS_RS_UPW2_DEFAULT && !getenv("UIS_RS_NO_UPW2")) || (opts->flags & UIS_FLAG_REPLICATED_SELECT)) &&
             resident_rs_lds_bytes(m.Hp, m.Dp, B, Kmax, S, 2, true) <= 160 * 1024)
      rs_kind = (m.H == 512 && m.D == 256 && B == 10 && Kmax == 16 && getenv("UIS_RS_UPW2_C1")) ? RS_UPW2_C1 : RS_UPW2;
    else if (per_xcd <= UIS_RS_UTT && m.Hp == 512 && (m.Dp == 256 || m.Dp == 512) &&
             ((UIS_RS_WIDE_DEFAULT && !getenv("UIS_RS_NO_WIDE")) || (opts->flags & UIS_FLAG_REPLICATED_SELECT)) &&
             rs_select_ok(B, Kmax, S, (long)maxT, 4) && F < 0x7fffffffLL && resident_rs_lds_bytes(m.Hp, m.Dp, B, Kmax, S, 1, true) <= 160 * 1024)
      rs_kind = (m.Dp == 512 && m.D == 512 && m.H == 512 && B == 20 && Kmax == 11 && getenv("UIS_RS_WIDE_C4")) ? RS_WIDE_C4 : RS_WIDE;
  }
  const bool rs = rs_kind != RS_NONE;
  const size_t mse_tab_bytes = ((size_t)2 * U * S * 4 + 255) & ~(size_t)255;
  const size_t mse_part_bytes = (size_t)nclq * rx_stride * rs_part_stride(m.Dp) * 4;
  if (rs) ENSURE(mse_tab, mse_tab_bytes + mse_part_bytes);
  const bool dbg = (opts->flags & UIS_FLAG_DEBUG_SCORES) != 0;
  // one array per window: [windows][U][B][Kmax + 1] ^ look_ahead
  double dbg_want = dbg ? (double)((maxT + L - 1) / L) * U * B : 0.0;
  for (int k = 0; k < L; ++k) dbg_want *= (double)(Kmax + 1);
  if (dbg_want > 1e9) return fail(UIS_ERR_UNSUPPORTED, "UIS_FLAG_DEBUG_SCORES: more than 1e9 candidate scores (a test hook for small decodes)");
  const size_t dbg_floats = (size_t)dbg_want;
  if (dbg) ENSURE(dbg_scores, std::max<size_t>(dbg_floats, 1) * 4);
  h->dbg_floats = 0;
  if (wnd) {
    ENSURE(lv_n, (size_t)2 * U * 4);
    ENSURE(lv_K, (size_t)2 * U * NC * 4);
    ENSURE(lv_last, (size_t)2 * U * NC * 4);
    ENSURE(lv_sum, (size_t)2 * U * NC * 4);
    ENSURE(lv_score, (size_t)2 * U * NC * 4);
    ENSURE(lv_origin, (size_t)2 * U * NC * 4);
    ENSURE(lv_path, (size_t)2 * U * NC * L * 2);
    ENSURE(lv_slot, (size_t)2 * U * NC * Kmax * 4);
    ENSURE(lv_blk, (size_t)2 * U * NC * Kmax * 4);
    ENSURE(scratch, (size_t)U * wsl.total);
    ENSURE(bp16, (size_t)std::max<int64_t>(bp_base[U], 1) * (L + 1) * 2);
    ENSURE(bp_base, (size_t)(U + 1) * 8);
  }
#undef ENSURE
  {
    const bool use_arena = getenv("UIS_NO_ARENA") == nullptr;
    size_t total = 0;
    for (auto& w : want) total += (w.second + 4095) & ~(size_t)4095;
    if (use_arena) {
      // (experiment knob, tools/experiments/bimodal.py: the one-launch decode runs in one of two modes
      // 4 % apart depending on where its buffers land)
      const size_t shift = getenv("UIS_ARENA_SHIFT") ? ((size_t)atol(getenv("UIS_ARENA_SHIFT")) & ~(size_t)4095) : 0;
      if ((rc = h->arena.ensure(total + shift))) return rc;
      size_t o = shift;
      for (auto& w : want) {
        if (w.first->p && !w.first->borrowed) (void)hipFree(w.first->p);
        w.first->p = static_cast<char*>(h->arena.p) + o;
        w.first->cap = w.second;
        w.first->borrowed = true;
        o += (w.second + 4095) & ~(size_t)4095;
      }
    } else {
      for (auto& w : want)
        if ((rc = w.first->ensure(w.second))) return rc;
    }
  }

  // ---- placement of the control words for this decode (uis_handle::CtlTune)
  uis_handle::CtlTune& tn = h->ctl_tune;
  int ctl_cand = 0;
  // (not for k_decode_rs: its row descriptors and row counters live in LDS, only the barrier
  // counters are polled in memory, and what is left of the placement effect is 1 % --
  // profiles/r03_bimodal.txt -- against 5 % for the kernels that keep them in global memory)
  if (resident && !rs && !getenv("UIS_NO_CTL_TUNE")) {
    const uint64_t sig = ((uint64_t)U << 44) ^ ((uint64_t)F << 16) ^ ((uint64_t)maxT << 6) ^ ((uint64_t)B << 1) ^ ((uint64_t)Kmax << 54);
    if (tn.sig != sig) { tn = uis_handle::CtlTune{}; tn.sig = sig; }
    ctl_cand = (tn.phase >= 1 && tn.phase <= 4) ? tn.phase - 1 : tn.best;
  }
  size_t ctl_off = ctl_place[ctl_cand];
  if (const char* e = getenv("UIS_CTL_OFFSET")) ctl_off = std::min<size_t>((size_t)atol(e) & ~(size_t)127, ctl_place[3]);  // experiments
  uint32_t* const ctl = reinterpret_cast<uint32_t*>(h->cluster_ctl.as<char>() + ctl_off);

  // ... with the selects of a rank's utterances running concurrently, one wave each (k_decode_big<.., true>),
  // where the single-wave select applies; UIS_FLAG_OWNER_SELECT keeps them one after the other
  const int per_rank = (((U + nclq - 1) / nclq) + 31) / 32;
  const bool ws_shape = !(opts->flags & UIS_FLAG_OWNER_SELECT) && m.Dp <= 256 && per_rank <= 8 &&
                        rs_select_ok(B, Kmax, S, (long)maxT, 3) &&
                        big_ws_lds_bytes(m.Hp, m.Dp, B, Kmax, S, per_rank) <= 160 * 1024;
  // Where k_decode_big takes over from k_decode_resident (round 5, from the sweep profiles/r05_usweep_dispatch.json:
  // 128 utterances 2.03 against 1.89 M frames/s, 160: 2.20 / 2.22, 192: 2.31 / 2.46, 224: 2.42 / 2.52, 256: 2.43 /
  // 2.65): with the concurrent single-wave selects from 21 utterances per XCD on (rounds 2-4 switched at "more
  // utterances than workgroups", 33 per XCD); without them (observation dim 512, wide beams) the sequential
  // selects keep the old switch (profiles/r05_usweep_c4_shape.json: a tie at 128).  UIS_BIG_MIN_U: experiments.
  const int big_from = getenv("UIS_BIG_MIN_U") ? atoi(getenv("UIS_BIG_MIN_U")) : (ws_shape ? 20 : 32) * ncl + 1;
  const bool big = resident && U >= big_from && !(opts->flags & UIS_FLAG_SMALL_TILES) &&
                   big_lds_bytes(m.Hp, m.Dp, B, Kmax, S) <= 160 * 1024;
  const bool big_ws = big && ws_shape;
  // ... or, on request (UIS_FLAG_COHORTS / UIS_COHORTS=1: measured slower, an experiment that stays tested), as two
  // utterance cohorts in flight per XCD (k_decode_coh: a cohort's select and hand-off waits filled with the other
  // cohort's dense stages)
#if defined(UIS_WITH_COHORTS)
  const bool coh = big_ws && ((opts->flags & UIS_FLAG_COHORTS) || getenv("UIS_COHORTS")) &&
                   coh_lds_bytes(m.Hp, m.Dp, B, Kmax, S, per_rank) <= 160 * 1024;
#else
  const bool coh = false;  // (round 6: the cohort kernel lost every measurement; it lives on in the -DUIS_WITH_COHORTS test variant)
#endif
  // ---- (round 5) ingestion overlapped with the decode.  The one-launch kernels own every CU, so nothing can be
  // copied-and-projected "behind" them -- but k_decode_rs / k_decode_big<WS> / k_decode_resident (one utterance per
  // workgroup) can stop after any step and pick up again
  // (DecodeState::step0 / step1 / resume).  For a list of equal-length utterances given in HOST memory the decode is
  // several launches: the first slice of every utterance's frames travels (one strided copy) and is projected, the
  // first launch decodes the steps that need nothing else (a step looks one frame ahead: the early MSEs and the
  // partial sums of the next select), the next slice travels and is projected behind it, and so on.  What is left
  // exposed of the PCIe leg is the first slice.  UIS_NO_SPLIT=1 keeps one launch (A/B switch, bit-identical).
  // (utterances of equal length: a slice is ONE strided copy and the projection's batches are a constant stride apart.
  // A ragged list: only through the float64 entry, whose staging block the library lays out itself -- slice after
  // slice, so that a slice is one copy too, scattered to the utterance-major frame stream on the device; a copy per
  // utterance and slice measured 2.48 against 3.59 M frames/s at a ragged configs[3] share -- and from 64 MB of frames on:
  // a ragged configs[1] (24 MB) loses 3-5 % to the extra launches, 256 ragged utterances (96 MB) gain 2 %)
  const bool uniform = n_utt > 0 && !ragged_list;
  const int64_t uniN = maxN;  // the longest utterance: slice boundaries are frame indices inside an utterance
  // (... and a list too small to spend a launch on keeps one: below 8 MB of frames the whole copy takes less than the
  // ~0.15 ms a further launch costs.  UIS_SPLIT_MIN_MB moves both sizes: tests, experiments)
  const double split_min_bytes = getenv("UIS_SPLIT_MIN_MB") ? 1e6 * atof(getenv("UIS_SPLIT_MIN_MB")) : (uniform ? 8e6 : 64e6);
  const bool split_shape = (uniform || h->src64 != nullptr) && (double)F * m.D * 4.0 >= split_min_bytes;
  std::vector<int64_t> cuts;
  if (uniN >= 128) {
    if (const char* e = getenv("UIS_SPLIT_FRAMES")) {
      for (const char* p2 = e; *p2;) {
        char* end = nullptr;
        const long v = strtol(p2, &end, 10);
        if (end == p2) break;
        const int64_t lo = cuts.empty() ? 32 : cuts.back() + 32;
        if (lo <= uniN - 32 && cuts.size() < 7) cuts.push_back(std::max<int64_t>(lo, std::min<int64_t>(v, uniN - 32)));
        p2 = *end ? end + 1 : end;
      }
    } else {
      // A launch must last as long as the next slice travels, and every further launch costs ~0.15 ms (measured:
      // configs[1] with cuts 32 | 32,128 | 32,96,288: 1.622 / 1.604 / 1.591 M frames/s from pinned float32).  Model:
      // a decode step takes ~(11.7 + 0.108 U) us (profiles/r05_usweep.json), a frame of every utterance U D 4 bytes at
      // ~45 GB/s (a quarter more through the float64 cast); slice k + 1 = what travels during 0.9 of launch k, and a
      // last slice below a quarter of the utterance is not worth a launch of its own.
      const double step_us = 11.7 + 0.108 * U, frame_us = (double)U * m.D * 4.0 / 45e3 * (h->src64 ? 1.25 : 1.0);
      // (round 6) ... and no further cut once everything that is left travels within the launch in front of it plus two
      // relaunches' worth (0.3 ms): configs[1] got the cuts {32, 326} and paid a second relaunch for frames that had
      // arrived five milliseconds earlier -- 1.607 M frames/s through the float64 list against 1.629 M with the one cut
      // at 32 (profiles/r06_f64_leg_knobs.txt); the configs[3] share keeps its slices (a launch there lasts 4 ms, the rest 28)
      int64_t prev = 0, cur = 32;
      while (cur <= uniN - 32 && (int)cuts.size() < 6) {
        if (!cuts.empty() && uniN - cur < uniN / 4) break;
        cuts.push_back(cur);
        if ((double)(uniN - cur) * frame_us <= (double)(cur - prev) * step_us + 300.0) break;
        const int64_t next = cur + std::max<int64_t>(32, (int64_t)(0.9 * (double)(cur - prev) * step_us / frame_us));
        prev = cur; cur = next;
      }
    }
  }
  const int64_t T1 = cuts.empty() ? 0 : cuts[0];
  const bool split = T1 > 0 && split_shape && h_frames && F > 0 && resident && (rs_kind == RS_BASE || rs_kind == RS_C1 || (big_ws && !coh) || (!rs && !big && U <= 32 * nclq)) && !profile &&
                     !dbg && m.D == m.Dp && (m.Dp == 128 || m.Dp == 256 || m.Dp == 512) && !(opts->flags & UIS_FLAG_SMALL_TILES) &&
                     !getenv("UIS_NO_SPLIT");
  std::unique_ptr<CastTeam> team;
  struct TeamGuard {  // nobody may still be inside the team when it goes out of scope
    CastPool* pool = nullptr;
    ~TeamGuard() { if (pool) pool->finish(); }
  } team_guard;
  int64_t cast_blocks_a = 0;
  // (split) what travels as ONE strided copy: a slice, or -- float64 lists, whose cast feeds the copies -- a piece of a
  // slice, so that a piece is on its way while the next one is cast; slice k = units [unit_first[k], unit_first[k + 1])
  struct CopyUnit { int64_t t0, t1; };
  std::vector<CopyUnit> units;
  std::vector<size_t> unit_first;
  std::vector<int64_t> cast_blocks_upto;  // blocks of the cast's order that end unit i
  std::vector<int64_t> unit_row0;         // (ragged) first row of unit i in the staging block (one more: the end)
  std::vector<long> scatter_tab_host;     // (ragged) k_scatter_rows' tables, unit after unit, {block row, stream row, rows} per utterance
  if (split) {
    for (size_t k = 0; k <= cuts.size(); ++k) {
      const int64_t t0 = k ? cuts[k - 1] : 0, t1 = k < cuts.size() ? cuts[k] : uniN;
      const int np = (k > 0 && h->src64) ? (int)std::max<int64_t>(1, std::min<int64_t>(8, (t1 - t0) / 64)) : 1;
      unit_first.push_back(units.size());
      for (int q = 0; q < np; ++q) units.push_back(CopyUnit{t0 + (t1 - t0) * q / np, t0 + (t1 - t0) * (q + 1) / np});
    }
    unit_first.push_back(units.size());
  }
  if (split && h->src64) {
    // the cast starts NOW, in the order the frames are needed (every utterance's first slice, then the next ...), while
    // this thread is still busy with the decode's tables and memsets
    std::vector<std::pair<int64_t, int64_t>> order;
    std::vector<int64_t> dst_rows;
    int64_t cursor = 0;  // (ragged) next free row of the staging block
    for (size_t un = 0; un < units.size(); ++un) {
      unit_row0.push_back(cursor);
      for (int u = 0; u < n_utt; ++u) {
        const int64_t nu = offsets[u + 1] - offsets[u];
        const int64_t r0 = offsets[u] + std::min(units[un].t0, nu), r1 = offsets[u] + std::min(units[un].t1, nu);
        if (!uniform) {  // the scatter's table: {row in the block, row in the stream, rows}
          scatter_tab_host.push_back((long)cursor); scatter_tab_host.push_back((long)r0); scatter_tab_host.push_back((long)(r1 - r0));
        }
        for (int64_t r = r0; r < r1; r += CastTeam::kBlockRows) {
          order.emplace_back(r, std::min(r1, r + CastTeam::kBlockRows));
          if (!uniform) dst_rows.push_back(cursor + (r - r0));
        }
        cursor += r1 - r0;
      }
      cast_blocks_upto.push_back((int64_t)order.size());
    }
    unit_row0.push_back(cursor);
    cast_blocks_a = cast_blocks_upto[0];
    if (!h->cast_pool) h->cast_pool = new CastPool();
    team.reset(new CastTeam(h->src64, offsets, n_utt, m.D, F, h->h_cast, std::move(order), std::move(dst_rows)));
    team_guard.pool = static_cast<CastPool*>(h->cast_pool);
    team_guard.pool->post(team.get());
  }
  // ---- per-decode tables
  std::vector<double> logblk(n_log), logden(n_log);
  for (int64_t n = 0; n < n_log; ++n) {
    logblk[n] = n > 0 ? std::log((double)n) : 0.0;      // np.log(block_counts[cluster]), uisrnn.py:418-419
    logden[n] = std::log((double)n + h->alpha);          // np.log(sum(block_counts) + crp_alpha)
  }
  HIPCHK(hipMemcpyAsync(h->off.p, offsets, (size_t)(U + 1) * 8, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->logblk.p, logblk.data(), logblk.size() * 8, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->logden.p, logden.data(), logden.size() * 8, hipMemcpyHostToDevice, h->stream));
  if (wnd)
    HIPCHK(hipMemcpyAsync(h->bp_base.p, bp_base.data(), (size_t)(U + 1) * 8, hipMemcpyHostToDevice, h->stream));

  HIPCHK(hipEventRecord(h->ev_begin, h->stream));
  // never-written row descriptors must still name valid slots (step_tile in uis_kernels.hip)
  HIPCHK(hipMemsetAsync(h->rows.p, 0, (size_t)rows_cap * sizeof(RnnRow), h->stream));
  HIPCHK(hipMemsetAsync(ctl, 0, ctl_words * 4, h->stream));
  if (dbg && dbg_floats) HIPCHK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(h->dbg_scores.p), 0x7f800000, dbg_floats, h->stream));
  if (rs)  // tiles a model does not have stay +0 in every row's partial sums
    HIPCHK(hipMemsetAsync(h->mse_tab.as<char>() + mse_tab_bytes, 0, mse_part_bytes, h->stream));
  // once per decode: pad (only when D is not a multiple of 16), gi0 = W_ih0 x + b_ih0, mse0.
  // Host frames (uis_decode) arrive in chunks on the copy stream; chunk i's kernels overlap the
  // H2D of chunk i+1 (true overlap needs pinned host memory, uis_host_alloc).
  const float* d_x = (m.D != m.Dp && F > 0) ? h->xpad.as<float>() : d_frames;
  auto pre_chunk = [&](int64_t f0, int64_t f1) -> int {
    const long n = (long)(f1 - f0);
    if (m.D != m.Dp) {
      const long total = n * m.Dp;
      hipLaunchKernelGGL(k_pad_frames, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream,
                         d_frames + (size_t)f0 * m.D, h->xpad.as<float>() + (size_t)f0 * m.Dp, n, m.D, m.Dp);
      HIPCHK(hipGetLastError());
    }
    if (n >= UIS_PROJ_WIDE_ROWS) {  // enough rows to fill the device with 32-row x 16-tile workgroups
      const dim3 wgrid((unsigned)((n + 31) / 32), (unsigned)((m.G / 16 + 15) / 16));
      const float* xin = d_x + (size_t)f0 * m.Dp;
      float* gout = h->gi0.as<float>() + (size_t)f0 * m.G;
      const bool pipe = !(opts->flags & UIS_FLAG_SMALL_TILES);  // (the flag keeps the plain walk for A/B runs)
      if (pipe && m.Dp == 128) LAUNCH(UIS_K_INPUT_PROJ, k_dense_input_proj_pipe<1>, wgrid, dim3(256), 0, m, xin, gout, n, 0L, (const long*)nullptr);
      else if (pipe && m.Dp == 256) LAUNCH(UIS_K_INPUT_PROJ, k_dense_input_proj_pipe<2>, wgrid, dim3(256), 0, m, xin, gout, n, 0L, (const long*)nullptr);
      else if (pipe && m.Dp == 512) LAUNCH(UIS_K_INPUT_PROJ, k_dense_input_proj_pipe<4>, wgrid, dim3(256), 0, m, xin, gout, n, 0L, (const long*)nullptr);
      else LAUNCH(UIS_K_INPUT_PROJ, k_dense_input_proj_wide, wgrid, dim3(256), 0, m, xin, gout, n);
    } else
      LAUNCH(UIS_K_INPUT_PROJ, k_dense_input_proj, dense_grid(n, m.G / 16), dim3(256), 0, m, d_x + (size_t)f0 * m.Dp,
             h->gi0.as<float>() + (size_t)f0 * m.G, n);
    LAUNCH(UIS_K_INPUT_PROJ, k_mse0, dim3((unsigned)((n + 3) / 4)), dim3(256), (size_t)5 * m.Dp * 4, m,
           d_x + (size_t)f0 * m.Dp, h->mse0.as<float>() + f0, n, 0L, (const long*)nullptr);
    return UIS_OK;
  };
  // From here on DMA from the caller's (or the pinned staging) memory may be in flight: whichever way
  // this function is left -- an error return inside the chunk loop included -- both streams are
  // drained first, so the caller never gets its buffers back while the copy engine still reads them.
  std::vector<long> split_tab_host;  // (the source of an asynchronous copy: declared BEFORE the drain, so that it outlives the stream sync on every way out)
  struct Drain {
    uis_handle* h;
    ~Drain() { (void)hipStreamSynchronize(h->copy_stream); (void)hipStreamSynchronize(h->stream); }
  } drain_on_exit{h};
  // a time slice [t0, t1) of every utterance (equal lengths): input projection and fresh-cluster MSE, the
  // utterances as batches along grid.z
  const size_t pitch = (size_t)uniN * m.D * 4;  // (split, equal lengths: bytes between utterances, in the staging block and on the device)
  auto pre_rows = [&](Launcher& lch, size_t k, int64_t t0, int64_t t1) -> int {
    const long n = (long)(t1 - t0);
    const dim3 wgrid((unsigned)((n + 31) / 32), (unsigned)((m.G / 16 + 15) / 16), (unsigned)U);
    // (uniform: batch z starts z * uniN rows behind the first; ragged: the slice's table of {first row, rows} per utterance)
    const long* tab = uniform ? nullptr : h->split_tab.as<long>() + k * (size_t)U * 2;
    const float* xin = uniform ? d_x + (size_t)t0 * m.Dp : d_x;
    float* gout = uniform ? h->gi0.as<float>() + (size_t)t0 * m.G : h->gi0.as<float>();
    float* mout = uniform ? h->mse0.as<float>() + t0 : h->mse0.as<float>();
    if (m.Dp == 128) LAUNCH(UIS_K_INPUT_PROJ, k_dense_input_proj_pipe<1>, wgrid, dim3(256), 0, m, xin, gout, n, (long)uniN, tab);
    else if (m.Dp == 256) LAUNCH(UIS_K_INPUT_PROJ, k_dense_input_proj_pipe<2>, wgrid, dim3(256), 0, m, xin, gout, n, (long)uniN, tab);
    else LAUNCH(UIS_K_INPUT_PROJ, k_dense_input_proj_pipe<4>, wgrid, dim3(256), 0, m, xin, gout, n, (long)uniN, tab);
    LAUNCH(UIS_K_INPUT_PROJ, k_mse0, dim3((unsigned)((n + 3) / 4), 1, (unsigned)U), dim3(256), (size_t)5 * m.Dp * 4, m, xin, mout, n,
           (long)uniN, tab);
    return UIS_OK;
  };
  // rows [t0, t1) of every utterance, host -> device
  auto copy_rows = [&](size_t un) -> int {
    if (uniform) {
      const int64_t t0 = units[un].t0, t1 = units[un].t1;
      HIPCHK(hipMemcpy2DAsync(const_cast<float*>(d_frames) + (size_t)t0 * m.D, pitch, h_frames + (size_t)t0 * m.D, pitch,
                              (size_t)(t1 - t0) * m.D * 4, (size_t)U, hipMemcpyHostToDevice, h->copy_stream));
      return UIS_OK;
    }
    // ragged (float64 lists only): the unit is one block of the staging buffer -> the same rows of the device's block,
    // then every utterance's part to its place in the frame stream (on the copy stream too: ordered behind the copy)
    const int64_t r0 = unit_row0[un], r1 = unit_row0[un + 1];
    if (r1 > r0) {
      HIPCHK(hipMemcpyAsync(h->stage.as<float>() + (size_t)r0 * m.D, h_frames + (size_t)r0 * m.D, (size_t)(r1 - r0) * m.D * 4,
                            hipMemcpyHostToDevice, h->copy_stream));
      const long max_rows = (long)(units[un].t1 - units[un].t0);
      hipLaunchKernelGGL(k_scatter_rows, dim3((unsigned)std::min<long>(64, (max_rows * (m.D / 4) + 255) / 256), (unsigned)U), dim3(256), 0,
                         h->copy_stream, h->stage.as<float>(), const_cast<float*>(d_frames), h->scatter_tab.as<long>() + un * (size_t)U * 3, m.D);
      HIPCHK(hipGetLastError());
    }
    return UIS_OK;
  };
  if (split) {
    while (h->h2d_done.size() < cuts.size() + 1) {
      hipEvent_t e;
      HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      h->h2d_done.push_back(e);
    }
    HIPCHK(hipStreamWaitEvent(h->copy_stream, h->ev_begin, 0));
    if (!uniform) {  // the slices' batch tables: {first row, rows} per utterance and slice
      split_tab_host.assign((cuts.size() + 1) * (size_t)U * 2, 0L);
      std::vector<long>& tab = split_tab_host;
      for (size_t k = 0; k <= cuts.size(); ++k) {
        const int64_t t0 = k ? cuts[k - 1] : 0, t1 = k < cuts.size() ? cuts[k] : uniN;
        for (int u = 0; u < n_utt; ++u) {
          const int64_t nu = offsets[u + 1] - offsets[u], a = std::min(t0, nu);
          tab[(k * U + u) * 2] = (long)(offsets[u] + a);
          tab[(k * U + u) * 2 + 1] = (long)(std::min(t1, nu) - a);
        }
      }
      HIPCHK(hipMemcpyAsync(h->split_tab.p, tab.data(), tab.size() * sizeof(long), hipMemcpyHostToDevice, h->stream));
      // (the scatter runs on the copy stream: its tables go up on that stream, ahead of the first block)
      HIPCHK(hipMemcpyAsync(h->scatter_tab.p, scatter_tab_host.data(), scatter_tab_host.size() * sizeof(long), hipMemcpyHostToDevice,
                            h->copy_stream));
    }
    if (team) team->wait_blocks(cast_blocks_a);
    if ((rc = copy_rows(0))) return rc;
    HIPCHK(hipEventRecord(h->h2d_done[0], h->copy_stream));
    HIPCHK(hipStreamWaitEvent(h->stream, h->h2d_done[0], 0));
    if ((rc = pre_rows(lch, 0, 0, T1))) return rc;
  } else if (F > 0 && h_frames) {
    const int n_chunks = (int)std::max<int64_t>(1, std::min<int64_t>(UIS_H2D_CHUNKS, F / UIS_H2D_MIN_FRAMES));
    while ((int)h->h2d_done.size() < n_chunks) {
      hipEvent_t e;
      HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      h->h2d_done.push_back(e);
    }
    HIPCHK(hipStreamWaitEvent(h->copy_stream, h->ev_begin, 0));
    // float64 utterances: cast into the pinned staging buffer by the handle's pool (and this thread),
    // piece p + 1 while piece p copies / projects.  A projection chunk then travels in several pieces,
    // so that the first copy starts after an eighth of the cast, not half of it.
    if (h->src64) {
      if (!h->cast_pool) h->cast_pool = new CastPool();
      team.reset(new CastTeam(h->src64, offsets, n_utt, m.D, F, h->h_cast));
      team_guard.pool = static_cast<CastPool*>(h->cast_pool);
      team_guard.pool->post(team.get());
    }
    const int pieces = team ? (int)std::max<int64_t>(1, std::min<int64_t>(UIS_F64_PIECES, (F / n_chunks) / 1024)) : 1;
    for (int c = 0; c < n_chunks; ++c) {
      const int64_t f0 = F * c / n_chunks, f1 = F * (c + 1) / n_chunks;
      for (int pc = 0; pc < pieces; ++pc) {
        const int64_t g0 = f0 + (f1 - f0) * pc / pieces, g1 = f0 + (f1 - f0) * (pc + 1) / pieces;
        if (team) team->wait_rows(g1);
        HIPCHK(hipMemcpyAsync(const_cast<float*>(d_frames) + (size_t)g0 * m.D, h_frames + (size_t)g0 * m.D,
                              (size_t)(g1 - g0) * m.D * 4, hipMemcpyHostToDevice, h->copy_stream));
      }
      HIPCHK(hipEventRecord(h->h2d_done[c], h->copy_stream));
      HIPCHK(hipStreamWaitEvent(h->stream, h->h2d_done[c], 0));
      if ((rc = pre_chunk(f0, f1))) return rc;
    }
  } else if (F > 0) {
    if ((rc = pre_chunk(0, F))) return rc;
  }
  HIPCHK(hipEventRecord(h->ev_pre, h->stream));

  // ---- group views of the shared buffers
  const size_t rows_per_utt = (size_t)(L == 1 ? B : NC);
  for (int g = 0; g < G; ++g) {
    GroupPlan& gp = plan[g];
    DecodeState& st = gp.st;
    const size_t u0 = (size_t)gp.u0;
    st.U = gp.U; st.B = B; st.Kmax = Kmax; st.S = S; st.L = L; st.tau = tau; st.flags = opts->flags | (agent_flags_env() ? UIS_FLAG_AGENT_FLAGS : 0u); st.wnd = wnd ? 1 : 0;
    st.max_rows = (int)((size_t)gp.U * rows_per_utt);
    st.off = h->off.as<int64_t>() + u0;
    st.utt_step = h->utt_step.as<int32_t>() + u0;
    st.overflow = h->overflow.as<int32_t>() + u0;
    st.x = d_x; st.gi0 = h->gi0.as<float>(); st.mse0 = h->mse0.as<float>();
    st.logblk = h->logblk.as<double>(); st.logden = h->logden.as<double>();
    st.pool_mean = h->pool_mean.as<float>() + u0 * S * m.Dp;
    st.pool_hid = h->pool_hid.as<float>() + u0 * S * m.depth * m.Hp;
    st.pool_cnt = h->pool_cnt.as<int32_t>() + u0 * S;
    // beam tables: groups back to back, each laid out [2][U_g][...]
    st.beam_n = h->beam_n.as<int32_t>() + 2 * u0;
    st.beam_K = h->beam_K.as<int32_t>() + 2 * u0 * B;
    st.beam_last = h->beam_last.as<int32_t>() + 2 * u0 * B;
    st.beam_sum = h->beam_sum.as<int32_t>() + 2 * u0 * B;
    st.beam_score = h->beam_score.as<float>() + 2 * u0 * B;
    st.beam_slot = h->beam_slot.as<int32_t>() + 2 * u0 * B * Kmax;
    st.beam_blk = h->beam_blk.as<int32_t>() + 2 * u0 * B * Kmax;
    st.bp = h->bp.as<uint32_t>();
    st.rows = h->rows.as<RnnRow>() + u0 * rows_per_utt + 48 * (size_t)g;
    st.nrows = h->nrows.as<int32_t>() + 2 * g;
    st.gi_up = h->gi_up.as<float>() + (m.depth > 1 ? (u0 * rows_per_utt + 48 * (size_t)g) * m.G : 0);
    st.a1 = h->a1.as<float>() + (u0 * rows_per_utt + 48 * (size_t)g) * m.Hp;
    st.counters = h->counters.as<unsigned long long>() + 4 * g;
    st.cl_abort = ctl + 16;
    st.utt_nrows = h->utt_nrows.as<int32_t>() + 2 * u0;
    st.hst = deep_shape ? h->hst.as<float>() : nullptr;
    st.step0 = 0; st.step1 = 0; st.resume = h->resume.as<unsigned char>(); st.resume_stride = 0;
    st.hst_elems = (size_t)rows_cap * m.Hp;
    st.dbg_scores = dbg ? h->dbg_scores.as<float>() + 0 : nullptr;  // (one group: groups would need their own utterance offset)
    if (resident || win || deep) {
      st.ncl = ncl;
      st.cl_xcc = ctl;
      st.rx_stride = rx_stride;
      st.rx_nrows = reinterpret_cast<int32_t*>(ctl) + 32;
      st.rx_bar = ctl + 32 + UIS_MAX_CLUSTERS * 32;
      st.rx_flags = ctl + 32 + 2 * UIS_MAX_CLUSTERS * 32;
      if (rs) {  // (one group: resident_ok)
        st.mse_tab = h->mse_tab.as<float>();
        st.mse_part = reinterpret_cast<float*>(h->mse_tab.as<char>() + mse_tab_bytes);
      }
    }
    if (wnd) {  // level buffers: groups back to back, each [2][U_g][NC]...
      st.NC = (int)NC;
      st.lv_n = h->lv_n.as<int32_t>() + 2 * u0;
      st.lv_K = h->lv_K.as<int32_t>() + 2 * u0 * NC;
      st.lv_last = h->lv_last.as<int32_t>() + 2 * u0 * NC;
      st.lv_sum = h->lv_sum.as<int32_t>() + 2 * u0 * NC;
      st.lv_score = h->lv_score.as<float>() + 2 * u0 * NC;
      st.lv_origin = h->lv_origin.as<int32_t>() + 2 * u0 * NC;
      st.lv_path = h->lv_path.as<int16_t>() + 2 * u0 * NC * L;
      st.lv_slot = h->lv_slot.as<int32_t>() + 2 * u0 * NC * Kmax;
      st.lv_blk = h->lv_blk.as<int32_t>() + 2 * u0 * NC * Kmax;
      st.scratch = h->scratch.as<unsigned char>() + u0 * wsl.total;
      st.scratch_stride = wsl.total;
      st.bp16 = h->bp16.as<uint16_t>();
      st.bp_base = h->bp_base.as<int64_t>() + u0;
    }
  }

  // ---- lock-step decode of every group on its own stream
  int decode_kernel = UIS_DK_STEPWISE | (dense_family(h, opts->flags, (long)plan[0].st.max_rows) << 8);
  for (int g = 0; g < G; ++g) {
    GroupPlan& gp = plan[g];
    hipStream_t sg = h->gstreams[g];
    Launcher gl{h, sg, profile};
    Launcher& lch = gl;  // LAUNCH() below targets this group's stream
    HIPCHK(hipStreamWaitEvent(sg, h->ev_pre, 0));
    LAUNCH(-1, k_init_state, dim3((gp.U + 255) / 256), dim3(256), 0, gp.st);
    if (resident) {
      // h1 into the extra slot, then ONE launch for every step of every utterance
      HIPCHK(hipMemcpyAsync(gp.st.pool_hid + (size_t)U * S * m.Hp, m.h1, (size_t)m.Hp * 4, hipMemcpyDeviceToDevice, sg));
      // more utterances than workgroups: the variant whose dense stages give a wave a whole row tile
      // (k_decode_big: +6 % at 288 utterances, +17 % at 768 / 1024; up to 256 the LDS-resident beam of
      // k_decode_resident wins); UIS_FLAG_SMALL_TILES keeps the split-K passes (A/B switch, bit-identical)
      const bool rs_two = rs_kind == RS_UPW2 || rs_kind == RS_UPW2_C1, rs_wide = rs_kind == RS_WIDE || rs_kind == RS_WIDE_C4;
      const size_t shmem = std::max<size_t>(rs       ? resident_rs_lds_bytes(m.Hp, m.Dp, B, Kmax, S, rs_two ? 2 : 1, rs_two || rs_wide)
#if defined(UIS_WITH_COHORTS)
                                            : coh    ? coh_lds_bytes(m.Hp, m.Dp, B, Kmax, S, per_rank)
#endif
                                            : big_ws ? big_ws_lds_bytes(m.Hp, m.Dp, B, Kmax, S, per_rank)
                                            : big    ? big_lds_bytes(m.Hp, m.Dp, B, Kmax, S)
                                                     : resident_lds_bytes(m.Hp, m.Dp, B, Kmax, S),
                                            96 * 1024);  // one workgroup per CU
      decode_kernel = rs ? (UIS_DK_RS | (rs_kind << 16)) : coh ? UIS_DK_BIG_COH : big_ws ? UIS_DK_BIG_WS : big ? UIS_DK_BIG : UIS_DK_RESIDENT;
      // the shapes of BASELINE's configs as compile-time constants (unpadded models only; UIS_NO_SHAPE_CLASSES=1
      // keeps the run-time instantiations: A/B switch, bit-identical)
      const bool exact = m.D == m.Dp && m.H == m.Hp && !getenv("UIS_NO_SHAPE_CLASSES");
      const bool cls_c1 = exact && m.Hp == 512 && m.Dp == 256 && B == 10 && Kmax == 16;   // configs[1] / [3]: beam 10, cap 16
      const bool cls_c4 = exact && m.Hp == 512 && m.Dp == 512 && B == 20 && Kmax == 11;   // configs[4]: beam 20, cap 11
      // (round 5) the launch as a function of the step range: once for the whole decode, or twice with the rest of the
      // frames arriving behind the first launch (split, above)
      auto launch_resident = [&]() -> int {
#if defined(UIS_WITH_COHORTS)
#define UIS_COH_CASE(HPV, DPV, COND, ...)                                                                             \
  if (m.Hp == HPV && m.Dp == DPV && coh && (COND)) {                                                                 \
    void (*kern)(DevModel, DecodeState) = &k_decode_coh<HPV, DPV, ##__VA_ARGS__>;                                    \
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,     \
                               (int)shmem));                                                                         \
    if ((rc = gl.run_cooperative(UIS_K_GRU, kern, h->n_cu, dim3(32 * ncl), dim3(512), shmem, m, gp.st)))           \
      return rc;                                                                                                     \
  }
      UIS_COH_CASE(512, 256, cls_c1, 10, 16)
      UIS_COH_CASE(512, 256, !cls_c1)
      UIS_COH_CASE(512, 128, true)
      UIS_COH_CASE(256, 256, true)
      UIS_COH_CASE(256, 128, true)
      UIS_COH_CASE(128, 256, true)
      UIS_COH_CASE(128, 128, true)
#undef UIS_COH_CASE
#endif
#define UIS_BIGWS_CASE(HPV, DPV, COND, ...)                                                                           \
  if (m.Hp == HPV && m.Dp == DPV && big_ws && !coh && (COND)) {                                                              \
    void (*kern)(DevModel, DecodeState) = &k_decode_big<HPV, DPV, true, ##__VA_ARGS__>;                              \
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,     \
                               (int)shmem));                                                                         \
    if ((rc = gl.run_cooperative(UIS_K_GRU, kern, h->n_cu, dim3(32 * ncl), dim3(512), shmem, m, gp.st)))           \
      return rc;                                                                                                     \
  }
      UIS_BIGWS_CASE(512, 256, cls_c1, 10, 16)
      UIS_BIGWS_CASE(512, 256, !cls_c1)
      UIS_BIGWS_CASE(512, 128, true)
      UIS_BIGWS_CASE(256, 256, true)
      UIS_BIGWS_CASE(256, 128, true)
      UIS_BIGWS_CASE(128, 256, true)
      UIS_BIGWS_CASE(128, 128, true)
#undef UIS_BIGWS_CASE
#define UIS_RS_CASE(KIND, HPV, DPV, ...)                                                                              \
  if (m.Hp == HPV && m.Dp == DPV && rs_kind == KIND) {                                                               \
    void (*kern)(DevModel, DecodeState) = &k_decode_rs<HPV, DPV, __VA_ARGS__>;                                      \
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,     \
                               (int)shmem));                                                                         \
    if ((rc = gl.run_cooperative(UIS_K_GRU, kern, h->n_cu, dim3(32 * ncl), dim3(512), shmem, m, gp.st)))           \
      return rc;                                                                                                     \
  }
      //          kind      HP   DP   NPOS UPW CB  CK  SPLIT2
      UIS_RS_CASE(RS_BASE, 512, 256, 3, 1, 0, 0, false)
      UIS_RS_CASE(RS_BASE, 512, 128, 3, 1, 0, 0, false)
      UIS_RS_CASE(RS_BASE, 256, 256, 3, 1, 0, 0, false)
      UIS_RS_CASE(RS_BASE, 256, 128, 3, 1, 0, 0, false)
      UIS_RS_CASE(RS_BASE, 128, 256, 3, 1, 0, 0, false)
      UIS_RS_CASE(RS_BASE, 128, 128, 3, 1, 0, 0, false)
      UIS_RS_CASE(RS_C1, 512, 256, 3, 1, 10, 16, false)
      UIS_RS_CASE(RS_UPW2, 512, 256, 3, 2, 0, 0, true)
      UIS_RS_CASE(RS_UPW2_C1, 512, 256, 3, 2, 10, 16, true)
      UIS_RS_CASE(RS_WIDE, 512, 256, 4, 1, 0, 0, true)
      UIS_RS_CASE(RS_WIDE, 512, 512, 4, 1, 0, 0, true)
      UIS_RS_CASE(RS_WIDE_C4, 512, 512, 4, 1, 20, 11, true)
#undef UIS_RS_CASE
#define UIS_RESIDENT_CLASS(HPV, DPV, COND, CBV, CKV)                                                                  \
  if (m.Hp == HPV && m.Dp == DPV && !rs && !big_ws && !big && (COND)) {                                              \
    void (*kern)(DevModel, DecodeState) = &k_decode_resident<HPV, DPV, false, CBV, CKV>;                            \
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,     \
                               (int)shmem));                                                                         \
    if ((rc = gl.run_cooperative(UIS_K_GRU, kern, h->n_cu, dim3(32 * ncl), dim3(512), shmem, m, gp.st)))           \
      return rc;                                                                                                     \
  }
      UIS_RESIDENT_CLASS(512, 256, cls_c1, 10, 16)
      UIS_RESIDENT_CLASS(512, 512, cls_c4, 20, 11)
#undef UIS_RESIDENT_CLASS
      const bool in_class = !big && (cls_c1 || cls_c4);
#define UIS_RESIDENT_CASE(HPV, DPV)                                                                                   \
  if (m.Hp == HPV && m.Dp == DPV && !rs && !big_ws && !in_class) {                                                   \
    void (*kern)(DevModel, DecodeState) = big ? &k_decode_big<HPV, DPV> : &k_decode_resident<HPV, DPV>;             \
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,     \
                               (int)shmem));                                                                         \
    if ((rc = gl.run_cooperative(UIS_K_GRU, kern, h->n_cu, dim3(32 * ncl), dim3(512), shmem, m, gp.st)))           \
      return rc;                                                                                                     \
  }
      UIS_RESIDENT_CASE(512, 256)
      UIS_RESIDENT_CASE(512, 512)
      UIS_RESIDENT_CASE(512, 128)
      UIS_RESIDENT_CASE(256, 256)
      UIS_RESIDENT_CASE(256, 128)
      UIS_RESIDENT_CASE(256, 512)
      UIS_RESIDENT_CASE(128, 256)
      UIS_RESIDENT_CASE(128, 128)
      UIS_RESIDENT_CASE(128, 512)
#undef UIS_RESIDENT_CASE
        return UIS_OK;
      };
      if (!split) {
        if ((rc = launch_resident())) return rc;
      } else {
        for (size_t k = 0; k <= cuts.size(); ++k) {
          if (k > 0) {
            // slice k of every utterance: cast (float64 lists), one strided copy, projection -- behind launch k - 1
            const int64_t t0 = cuts[k - 1], t1 = k < cuts.size() ? cuts[k] : uniN;
            for (size_t un = unit_first[k]; un < unit_first[k + 1]; ++un) {
              if (team) team->wait_blocks(cast_blocks_upto[un]);
              if ((rc = copy_rows(un))) return rc;
            }
            HIPCHK(hipEventRecord(h->h2d_done[k], h->copy_stream));
            HIPCHK(hipStreamWaitEvent(sg, h->h2d_done[k], 0));
            if ((rc = pre_rows(gl, k, t0, t1))) return rc;
            // (k_decode_big<WS> counts its barriers and rows from zero in every launch; the abort word and the XCC ids stay)
            if (!rs) HIPCHK(hipMemsetAsync(ctl + 32, 0, (ctl_words - 32) * 4, sg));
          }
          gp.st.step0 = k ? (int)cuts[k - 1] - 1 : 0;
          gp.st.step1 = k < cuts.size() ? (int)cuts[k] - 1 : 0;
          if ((rc = launch_resident())) return rc;
        }
      }
    } else if (win) {
      // h1 into the extra slot, then ONE launch for every sub-step of every window
      HIPCHK(hipMemcpyAsync(gp.st.pool_hid + (size_t)U * S * m.Hp, m.h1, (size_t)m.Hp * 4, hipMemcpyDeviceToDevice, sg));
      const size_t shmem = big_win_lds_bytes(m.Hp, S, (int)NC, Kmax, B);
      decode_kernel = UIS_DK_WINDOW;
      // (BASELINE configs[2]'s shape as compile-time constants: beam 50, cap 12, look_ahead 2, one level of 650 hypotheses;
      // UIS_NO_SHAPE_CLASSES=1 keeps the run-time instantiation: A/B switch, bit-identical)
      const bool win_c2 = m.D == m.Dp && m.H == m.Hp && m.Hp == 512 && m.Dp == 256 && B == 50 && Kmax == 12 && L == 2 &&
                          NC == (int64_t)B * (Kmax + 1) && S == B * Kmax + B + B * (Kmax + 1) && !getenv("UIS_NO_SHAPE_CLASSES");
      if (win_c2) {
        void (*kern)(DevModel, DecodeState) = &k_decode_big<512, 256, false, 50, 12, true>;
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        if ((rc = gl.run_cooperative(UIS_K_GRU, kern, h->n_cu, dim3(32 * ncl), dim3(512), shmem, m, gp.st))) return rc;
      }
#define UIS_WIN_CASE(HPV, DPV)                                                                                        \
  if (m.Hp == HPV && m.Dp == DPV && !win_c2) {                                                                       \
    void (*kern)(DevModel, DecodeState) = &k_decode_big<HPV, DPV, false, 0, 0, true>;                               \
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,     \
                               (int)shmem));                                                                         \
    if ((rc = gl.run_cooperative(UIS_K_GRU, kern, h->n_cu, dim3(32 * ncl), dim3(512), shmem, m, gp.st)))           \
      return rc;                                                                                                     \
  }
      UIS_WIN_CASE(512, 256)
      UIS_WIN_CASE(512, 128)
      UIS_WIN_CASE(512, 512)
      UIS_WIN_CASE(256, 256)
      UIS_WIN_CASE(256, 128)
      UIS_WIN_CASE(256, 512)
      UIS_WIN_CASE(128, 256)
      UIS_WIN_CASE(128, 128)
      UIS_WIN_CASE(128, 512)
#undef UIS_WIN_CASE
    } else if (deep) {
      // h1 of every layer into the extra slot, then ONE launch for every step of every utterance
      HIPCHK(hipMemcpyAsync(gp.st.pool_hid + (size_t)U * S * m.depth * m.Hp, m.h1, (size_t)m.depth * m.Hp * 4, hipMemcpyDeviceToDevice, sg));
      const size_t shmem = L == 1 ? deep_lds_bytes(m.Hp, m.Dp, B, Kmax, S) : big_win_lds_bytes(m.Hp, S, (int)NC, Kmax, B);
      decode_kernel = UIS_DK_DEEP;
#define UIS_DEEP_CASE(HPV, DPV)                                                                                       \
  if (m.Hp == HPV && m.Dp == DPV && L == 1) {                                                                        \
    void (*kern)(DevModel, DecodeState) = &k_decode_deep<HPV, DPV>;                                                 \
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,     \
                               (int)shmem));                                                                         \
    if ((rc = gl.run_cooperative(UIS_K_GRU, kern, h->n_cu, dim3(32 * ncl), dim3(512), shmem, m, gp.st)))           \
      return rc;                                                                                                     \
  }
      UIS_DEEP_CASE(512, 256)
      UIS_DEEP_CASE(512, 128)
      UIS_DEEP_CASE(512, 512)
      UIS_DEEP_CASE(256, 256)
      UIS_DEEP_CASE(256, 128)
      UIS_DEEP_CASE(128, 128)
      UIS_DEEP_CASE(128, 256)
#undef UIS_DEEP_CASE
#define UIS_DEEPW_CASE(HPV, DPV)                                                                                      \
  if (m.Hp == HPV && m.Dp == DPV && L > 1) {                                                                         \
    void (*kern)(DevModel, DecodeState) = &k_decode_deep<HPV, DPV, true>;                                           \
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,     \
                               (int)shmem));                                                                         \
    if ((rc = gl.run_cooperative(UIS_K_GRU, kern, h->n_cu, dim3(32 * ncl), dim3(512), shmem, m, gp.st)))           \
      return rc;                                                                                                     \
  }
      UIS_DEEPW_CASE(512, 256)
      UIS_DEEPW_CASE(256, 256)
      UIS_DEEPW_CASE(256, 128)
      UIS_DEEPW_CASE(128, 128)
#undef UIS_DEEPW_CASE
    } else if (small) {
      const size_t shmem = L == 1 ? small_lds_bytes(m.Dp, B, Kmax, S) : small_win_lds_bytes(S, (int)NC, Kmax, B);
      decode_kernel = UIS_DK_SMALL;
      if (L == 1) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decode_small<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        LAUNCH(UIS_K_GRU, k_decode_small<false>, dim3(gp.U), dim3(512), shmem, m, gp.st);
      } else {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decode_small<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        LAUNCH(UIS_K_GRU, k_decode_small<true>, dim3(gp.U), dim3(512), shmem, m, gp.st);
      }
    } else if (use_graph && gp.maxT >= UIS_GRAPH_STEPS) {
      GraphCache& gc = h->gcache[g];
      const bool same = gc.exec && gc.lds == (size_t)lds.total && memcmp(&gc.st, &gp.st, sizeof(DecodeState)) == 0;
      if (!same) {
        if (gc.exec) { (void)hipGraphExecDestroy(gc.exec); gc.exec = nullptr; }
        hipGraph_t graph = nullptr;
        HIPCHK(hipStreamBeginCapture(sg, hipStreamCaptureModeThreadLocal));
        rc = enqueue_steps(h, gl, gp.st, lds.total, UIS_GRAPH_STEPS);
        hipError_t ce = hipStreamEndCapture(sg, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (ce != hipSuccess) return fail(UIS_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(ce));
        ce = hipGraphInstantiate(&gc.exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ce != hipSuccess) { gc.exec = nullptr; return fail(UIS_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ce)); }
        gc.st = gp.st; gc.lds = (size_t)lds.total;
      }
      const int64_t nlaunch = (gp.maxT + UIS_GRAPH_STEPS - 1) / UIS_GRAPH_STEPS;  // the tail steps are no-ops
      for (int64_t i = 0; i < nlaunch; ++i) HIPCHK(hipGraphLaunch(gc.exec, sg));
    } else {
      const int64_t nsteps = gp.maxT + (gp.maxT & 1);
      for (int64_t s0 = 0; s0 < nsteps; s0 += 2)
        if ((rc = enqueue_steps(h, gl, gp.st, lds.total, 2))) return rc;
    }
    if (!wnd)
      LAUNCH(UIS_K_BACKTRACE, k_backtrace, dim3(gp.U), dim3(64), (size_t)64 * B, gp.st, d_labels,
             d_scores ? d_scores + gp.u0 : nullptr, h->beam_scores_out.as<float>() + (size_t)gp.u0 * B);
    else
      LAUNCH(UIS_K_BACKTRACE, k_backtrace_window, dim3((gp.U + 63) / 64), dim3(64), 0, gp.st, d_labels,
             d_scores ? d_scores + gp.u0 : nullptr, h->beam_scores_out.as<float>() + (size_t)gp.u0 * B);
    HIPCHK(hipEventRecord(h->gdone[g], sg));
  }
  for (int g = 0; g < G; ++g) HIPCHK(hipStreamWaitEvent(h->stream, h->gdone[g], 0));
  HIPCHK(hipEventRecord(h->ev_end, h->stream));

  std::vector<unsigned long long> counters((size_t)UIS_MAX_GROUPS * 4, 0ull);
  HIPCHK(hipMemcpyAsync(counters.data(), h->counters.p, (size_t)G * 4 * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(h->last_overflow.data(), h->overflow.p, (size_t)U * 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(h->last_beam_scores.data(), h->beam_scores_out.p, (size_t)U * B * 4, hipMemcpyDeviceToHost,
                        h->stream));
  uint32_t abort_word = 0;
  HIPCHK(hipMemcpyAsync(&abort_word, ctl + 16, 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (dbg) h->dbg_floats = dbg_floats;
  if (abort_word) {
    h->inlaunch_failed = true;
    return fail(UIS_ERR_HIP, abort_word == 2 ? "workgroup cluster not placed on one XCD (in-launch barrier path)"
                                             : "in-launch barrier timed out");
  }
#if defined(UIS_SELECT_TIMING)
  {
    unsigned long long tc[48];
    HIPCHK(hipMemcpy(tc, h->counters.as<unsigned long long>(), sizeof(tc), hipMemcpyDeviceToHost));

    const double launches = (double)maxT * U;
    if (L == 1) {
      fprintf(stderr, "[select timing] cycles per workgroup-launch:");
      for (int k = 0; k < 8; ++k) fprintf(stderr, " p%d=%.0f", k, (double)tc[16 + k] / launches);
      fprintf(stderr, "\n");
    } else {
      static const char* names[7] = {"live+offsets", "mse", "scores", "expand/prune", "leaders+slots", "tables", "records+rows"};
      for (int half = 0; half < 2; ++half) {
        const unsigned long long* c = tc + (half ? 32 : 16);
        const double n = (double)std::max<unsigned long long>(c[7], 1);
        fprintf(stderr, "[window timing] %s sub-steps, us per workgroup-launch:", half ? "pruning" : "expanding");
        double sum = 0.0;
        for (int k = 0; k < 7; ++k) { fprintf(stderr, " %s=%.2f", names[k], (double)c[k] * 0.01 / n); sum += (double)c[k] * 0.01 / n; }
        fprintf(stderr, " | total=%.2f; per launch: candidates=%.0f live=%.0f hypotheses in=%.0f rows=%.1f\n", sum, (double)c[8] / n,
                (double)c[9] / n, (double)c[10] / n, (double)c[11] / n);
      }
    }
  }
#endif
#if defined(UIS_RESIDENT_PROBE)
  if (resident) {
    unsigned long long tc[88];
    HIPCHK(hipMemcpy(tc, h->counters.as<unsigned long long>(), sizeof(tc), hipMemcpyDeviceToHost));
    fprintf(stderr, "[resident probe] cycles per dependent load: own-table(nt)=%.0f mean(sc1)=%.0f wgt(plain)=%.0f wgt-again=%.0f\n",
            (double)tc[80] / (double)maxT, (double)tc[81] / (double)maxT, (double)tc[82] / (double)maxT, (double)tc[83] / (double)maxT);
  }
#endif
#if defined(UIS_RS_COUNT_PATHS)
  if (rs) {
    unsigned long long tc[96];
    HIPCHK(hipMemcpy(tc, h->counters.as<unsigned long long>(), sizeof(tc), hipMemcpyDeviceToHost));
    fprintf(stderr, "[rs short lists] selects with <= 16 / <= 32 / <= 64 / more survivors: %llu %llu %llu %llu\n", tc[88], tc[89], tc[90], tc[91]);
  }
#endif
#if defined(UIS_RESIDENT_TIMING)
  if (win) {  // even sub-steps (expanding, at look_ahead 2) and odd ones (pruning) apart
    unsigned long long tc[88];
    HIPCHK(hipMemcpy(tc, h->counters.as<unsigned long long>(), sizeof(tc), hipMemcpyDeviceToHost));
    static const char* names[8] = {"window", "barA", "gru", "barB", "head1", "barC", "head2", "barD"};
    for (int wg = 0; wg < 2; ++wg)
      for (int odd = 0; odd < 2; ++odd) {
        fprintf(stderr, "[window launch timing] workgroup %3d, %s sub-steps, us per sub-step:", wg ? 248 : 0, odd ? "odd" : "even");
        double sum = 0.0;
        for (int k = 0; k < 8; ++k) {
          const double us = (double)tc[(wg ? 64 : 48) + 8 * odd + k] * 0.01 / ((double)maxT * 0.5);
          fprintf(stderr, " %s=%.2f", names[k], us);
          sum += us;
        }
        fprintf(stderr, " | total=%.2f\n", sum);
      }
    std::vector<unsigned long long> per((size_t)96 + 1024);
    HIPCHK(hipMemcpy(per.data(), h->counters.as<unsigned long long>(), per.size() * 8, hipMemcpyDeviceToHost));
    for (int wg = 0; wg < 2; ++wg) {
      fprintf(stderr, "[window launch timing] workgroup %3d, gru of the even sub-steps, us by wave:", wg ? 248 : 0);
      for (int w = 0; w < 8; ++w) fprintf(stderr, " %.1f", (double)per[80 + 8 * wg + w] * 0.01 / ((double)maxT * 0.5));
      fprintf(stderr, "\n");
    }
    static const char* what[4] = {"gru", "wait B", "head1", "head2"};
    for (int k = 0; k < 4; ++k) {
      fprintf(stderr, "[window launch timing] %s, even sub-steps, us by rank (mean over the clusters):", what[k]);
      for (int r = 0; r < 32; ++r) {
        double sum = 0.0;
        for (int c = 0; c < ncl; ++c) sum += (double)per[(size_t)96 + 256 * k + c + ncl * r];
        fprintf(stderr, " %.1f", sum / ncl * 0.01 / ((double)maxT * 0.5));
      }
      fprintf(stderr, "\n");
    }
  }
  if (resident && (decode_kernel & 0xff) == UIS_DK_BIG_COH) {  // k_decode_coh: waves 0 (cohort A), 1 (cohort B) and 7 (no utterance at <= 7 per rank) of workgroup 0
    unsigned long long tc[96];
    HIPCHK(hipMemcpy(tc, h->counters.as<unsigned long long>(), sizeof(tc), hipMemcpyDeviceToHost));
    static const char* names[16] = {"wait select A", "-", "-", "gru", "mean1", "mean2", "select", "early mse", "slot", "leave", "-", "-", "-", "-", "-", "-"};
    for (int k3 = 0; k3 < 3; ++k3) {
      fprintf(stderr, "[cohort timing] workgroup 0 wave %d, us per step:", k3 == 0 ? 0 : k3 == 1 ? 1 : 7);
      double sum = 0.0;
      for (int k = 0; k < 10; ++k) {
        if (k == 1 || k == 2) continue;
        const double us = (double)tc[48 + 16 * k3 + k] * 0.01 / (double)maxT;
        fprintf(stderr, " %s=%.2f", names[k], us);
        sum += us;
      }
      fprintf(stderr, " | total=%.2f\n", sum);
    }
  } else if (resident) {
    unsigned long long tc[88];
    HIPCHK(hipMemcpy(tc, h->counters.as<unsigned long long>(), sizeof(tc), hipMemcpyDeviceToHost));
    static const char* names[8] = {"select", "barA", "gru", "barB", "head1", "barC", "head2", "barD"};
    for (int wg = 0; wg < 2; ++wg) {
      fprintf(stderr, "[resident timing] workgroup %3d, us per step:", wg ? 248 : 0);
      for (int k = 0; k < 8; ++k) fprintf(stderr, " %s=%.2f", names[k], (double)tc[(wg ? 64 : 48) + k] * 0.01 / (double)maxT);
      fprintf(stderr, "\n");
    }
    fprintf(stderr, "[resident timing] select phases (wg 0), us per step:");
    for (int k = 0; k < 8; ++k) fprintf(stderr, " p%d=%.2f", k, (double)tc[80 + k] * 0.01 / (double)maxT);
    fprintf(stderr, "\n");
    fprintf(stderr, "[resident timing] gru fine (wg 248): other=%.2f tile=%.2f combine=%.2f sync=%.2f\n",
            (double)tc[72] * 0.01 / (double)maxT, (double)tc[73] * 0.01 / (double)maxT, (double)tc[74] * 0.01 / (double)maxT,
            (double)tc[75] * 0.01 / (double)maxT);
    if ((decode_kernel & 0xff) == UIS_DK_BIG || (decode_kernel & 0xff) == UIS_DK_BIG_WS) {  // k_decode_big: the GRU stage wave by wave, the row tiles per step
      unsigned long long wv[32];
      HIPCHK(hipMemcpy(wv, h->counters.as<unsigned long long>() + 96, sizeof(wv), hipMemcpyDeviceToHost));
      for (int wg = 0; wg < 2; ++wg) {
        fprintf(stderr, "[resident timing] workgroup %3d, gru us per step by wave:", wg ? 248 : 0);
        for (int w = 0; w < 8; ++w) fprintf(stderr, " %.1f", (double)wv[8 * wg + w] * 0.01 / (double)maxT);
        fprintf(stderr, "\n");
      }
      fprintf(stderr, "[resident timing] cluster 0: row tiles per step mean %.2f; steps by (row tiles mod 8):", (double)wv[24] / (double)maxT);
      for (int k = 0; k < 8; ++k) fprintf(stderr, " %d:%llu", k, wv[16 + k]);
      fprintf(stderr, "\n");
    }
  }
#endif
  if (resident && !rs && tn.sig != 0 && tn.phase <= 4 && !getenv("UIS_NO_CTL_TUNE")) {  // the decode's device time goes to the placement it ran with
    float ms = 0.0f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev_begin, h->ev_end));
    if (tn.phase >= 1) tn.ms[tn.phase - 1] = ms;
    if (++tn.phase == 5) {
      tn.best = 0;
      for (int k = 1; k < 4; ++k)
        if (tn.ms[k] < 0.995f * tn.ms[tn.best]) tn.best = k;  // (another placement has to win by 0.5 %: repeats agree to 0.1 %)
    }
  }
  int n_over = 0, n_level = 0;
  for (int u = 0; u < U; ++u) {
    n_level += (h->last_overflow[u] & 2) != 0;  // look_ahead >= 2: an intermediate level was full
    n_over += h->last_overflow[u] != 0;
  }
  if (stats) {
    float ms = 0.0f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev_begin, h->ev_end));
    stats->n_steps = (int32_t)maxT;
    stats->decode_ms = ms;
    for (int g = 0; g < G; ++g) {
      stats->rnn_rows += (int64_t)counters[4 * g + 0];
      stats->rnn_rows_nodedup += (int64_t)counters[4 * g + 1];
      stats->candidates += (int64_t)counters[4 * g + 2];
      stats->max_clusters_seen = std::max(stats->max_clusters_seen, (int32_t)counters[4 * g + 3]);
    }
    stats->n_overflow = n_over;
    stats->n_streams = G;
    stats->decode_kernel = decode_kernel;
    stats->decode_launches = (resident || win || deep || small) ? (split ? (int)cuts.size() + 1 : 1) : 0;
    if (profile) {
      for (size_t i = 0; i + 1 < h->prof.used; i += 2) {
        float t = 0.0f;
        HIPCHK(hipEventElapsedTime(&t, h->prof.ev[i], h->prof.ev[i + 1]));
        const int c = h->prof.cls[i / 2];
        if (c < 0) continue;
        stats->kernel_ms[c] += t;
        stats->kernel_launches[c] += 1;
      }
    }
  }
  if (n_level)
    return fail(UIS_ERR_UNSUPPORTED,
                std::to_string(n_level) + " utterance(s) had more than " + std::to_string((long long)NC) +
                    " live assignment prefixes inside a look-ahead window (beam_size * clusters ^ (look_ahead - 1)); "
                    "a larger max_clusters cannot help: lower look_ahead or beam_size");
  if (n_over)
    return fail(UIS_ERR_CLUSTER_CAP, std::to_string(n_over) + " utterance(s) needed more than max_clusters=" +
                                         std::to_string(Kmax) + " clusters per hypothesis");
  return UIS_OK;
}

// One decode; if the one-launch path was chosen automatically and its placement / barrier checks
// failed, repeat on the launch-per-step path and stay there for this handle.
int decode_impl(uis_handle* h, const float* d_frames, const int64_t* offsets, int32_t n_utt,
                const uis_decode_opts* opts, int32_t* d_labels, float* d_scores, uis_stats* stats,
                const float* h_frames = nullptr) {
  if (h) h->inlaunch_failed = false;
  int rc = decode_once(h, d_frames, offsets, n_utt, opts, d_labels, d_scores, stats, h_frames);
  if (rc == UIS_ERR_HIP && h && h->inlaunch_failed && opts &&
      !(opts->flags & UIS_FLAG_RESIDENT)) {
    // (the frames, if they came from the host, travel again: a decode in several launches that was refused at its
    // first launch has only the first slice on the device)
    h->resident_off = true;
    rc = decode_once(h, d_frames, offsets, n_utt, opts, d_labels, d_scores, stats, h_frames);
  }
  return rc;
}

}  // namespace

UIS_EXPORT int32_t uis_abi_version(void) { return UIS_ABI_VERSION; }
UIS_EXPORT int32_t uis_numerics_version(void) { return UIS_NUMERICS_VERSION; }
UIS_EXPORT uint32_t uis_build_flags(void) {
  uint32_t f = 0;
#if defined(UIS_WITH_COHORTS)
  f |= UIS_BUILD_COHORTS;
#endif
  return f;
}

UIS_EXPORT int32_t uis_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

UIS_EXPORT const char* uis_last_error(void) { return g_err.c_str(); }

static void stream_free_on_destroy(uis_handle* h) {
  for (void* p : h->stream_state.allocs) (void)hipFree(p);
  h->stream_state.allocs.clear();
  DevBuf* bufs[] = {&h->stream_state.chunk_x, &h->stream_state.chunk_pad, &h->stream_state.chunk_gi0,
                    &h->stream_state.chunk_mse0, &h->stream_state.labels, &h->stream_state.scores};
  for (DevBuf* b : bufs) b->release();
  if (h->stream_state.h_stage) { (void)hipHostFree(h->stream_state.h_stage); h->stream_state.h_stage = nullptr; }
  if (h->stream_state.pm_block) { (void)hipHostFree(h->stream_state.pm_block); h->stream_state.pm_block = nullptr; }
}

UIS_EXPORT void uis_destroy(uis_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream_state.pm_running) {  // tell the resident launch to leave before waiting for the stream
    pm_ring(h->stream_state, h->stream_state.pm_seq + 1, UIS_PM_QUIT, 0);
    h->stream_state.pm_running = false;
  }
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  stream_free_on_destroy(h);
  for (void* p : h->model_allocs) (void)hipFree(p);
  DevBuf* bufs[] = {&h->off, &h->utt_step, &h->overflow, &h->xpad, &h->gi0, &h->mse0, &h->logblk, &h->logden,
                    &h->pool_mean, &h->pool_hid, &h->pool_cnt, &h->beam_n, &h->beam_K, &h->beam_last, &h->beam_sum,
                    &h->beam_score, &h->beam_slot, &h->beam_blk, &h->bp, &h->rows, &h->nrows, &h->gi_up, &h->a1,
                    &h->counters, &h->beam_scores_out, &h->io_frames, &h->io_labels, &h->io_scores, &h->mse_tab, &h->dbg_scores, &h->utt_nrows, &h->hst, &h->resume, &h->split_tab, &h->scatter_tab, &h->stage,
                    &h->lv_n, &h->lv_K, &h->lv_last, &h->lv_sum, &h->lv_score, &h->lv_origin, &h->lv_path, &h->lv_slot,
                    &h->lv_blk, &h->scratch, &h->bp16, &h->bp_base, &h->cluster_ctl, &h->arena,
                    &h->ev_a, &h->ev_b, &h->ev_off, &h->ev_out};
  for (DevBuf* b : bufs) b->release();
  for (hipEvent_t e : h->prof.ev) (void)hipEventDestroy(e);
  if (h->ev_begin) (void)hipEventDestroy(h->ev_begin);
  if (h->ev_end) (void)hipEventDestroy(h->ev_end);
  if (h->ev_pre) (void)hipEventDestroy(h->ev_pre);
  for (GraphCache& gc : h->gcache) if (gc.exec) (void)hipGraphExecDestroy(gc.exec);
  for (hipEvent_t e : h->gdone) (void)hipEventDestroy(e);
  for (hipStream_t sg : h->gstreams) { (void)hipStreamSynchronize(sg); (void)hipStreamDestroy(sg); }
  for (hipEvent_t e : h->h2d_done) (void)hipEventDestroy(e);
  if (h->cast_pool) { delete static_cast<CastPool*>(h->cast_pool); h->cast_pool = nullptr; }
  if (h->h_cast) (void)hipHostFree(h->h_cast);
  if (h->h_out) (void)hipHostFree(h->h_out);
  if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

UIS_EXPORT int32_t uis_create(const uis_model_desc* d, int32_t device, uis_handle** out) {
  if (!d || !out) return fail(UIS_ERR_INVALID_ARG, "null desc/out");
  *out = nullptr;
  const int D = d->observation_dim, H = d->rnn_hidden_size, depth = d->rnn_depth;
  if (D < 1 || H < 1 || depth < 1 || depth > UIS_MAX_DEPTH)
    return fail(UIS_ERR_INVALID_ARG, "observation_dim, rnn_hidden_size >= 1 and 1 <= rnn_depth <= 8 required");
  if (!(d->transition_bias > 0.0 && d->transition_bias < 1.0))
    return fail(UIS_ERR_INVALID_ARG, "transition_bias must be in (0, 1) (the reference takes its log)");
  if (!(d->crp_alpha > 0.0)) return fail(UIS_ERR_INVALID_ARG, "crp_alpha must be > 0");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(UIS_ERR_NO_DEVICE, "no HIP device visible; this decoder has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(UIS_ERR_NO_DEVICE, "device index out of range");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(UIS_ERR_NO_DEVICE, "hipGetDeviceProperties failed");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(UIS_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
  if (hipSetDevice(device) != hipSuccess) return fail(UIS_ERR_NO_DEVICE, "hipSetDevice failed");

  uis_handle* h = new uis_handle();
  h->device = device;
  h->alpha = d->crp_alpha;
  if (hipDeviceGetAttribute(&h->n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) h->n_cu = 0;
  auto bail = [&](int rc) { uis_destroy(h); return rc; };
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking) != hipSuccess)
    return bail(fail(UIS_ERR_HIP, "stream create failed"));
  if (hipEventCreate(&h->ev_begin) != hipSuccess || hipEventCreate(&h->ev_end) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_pre, hipEventDisableTiming) != hipSuccess)
    return bail(fail(UIS_ERR_HIP, "event create failed"));
  DevModel& m = h->m;
  m.D = D; m.H = H; m.depth = depth;
  m.Dp = round_up(D, 16); m.Hp = round_up(H, 16);
  // The one-launch cluster kernels exist for padded hidden sizes 256 / 512 and observation dims 128 / 256 /
  // 512.  Padding further than to 16 is free of numerical consequences exactly where it keeps the canonical
  // K-segment length q = ceil(blocks / 8) (uis_numerics.h: the extra blocks are zeros inside the last
  // segments, empty segments add +0.0f either way; the MSE's sixteen tile accumulators take zero tiles):
  // hidden sizes 65 .. 256 and 385 .. 512, observation dims up to 256 and 385 .. 512 -- so those models
  // (rnn_depth 1) get the kernels' shapes instead of the launch-per-step path.
  HidMap hmap;
  if (!getenv("UIS_PAD_TO_16_ONLY")) {  // (any rnn_depth: the upper layers' K axis is the hidden size too)
    const int qh = (m.Hp / 16 + UIS_KSPLIT - 1) / UIS_KSPLIT, qd = (m.Dp / 16 + UIS_KSPLIT - 1) / UIS_KSPLIT;
    // (round 6) hidden sizes 257 .. 384 have segments of three k-blocks: into the 512 shape with a zero block behind
    // every segment (HidMap) -- they ran a launch per step before (UIS_NO_SEGMENT_PADDING=1: still do)
    const bool seg3 = qh == 3 && !getenv("UIS_NO_SEGMENT_PADDING");
    const int hp = (qh == 1 && H > 64) ? 128 : qh == 2 ? 256 : (qh == 4 || seg3) ? 512 : 0;  // (up to 64: k_decode_small's)
    const int dp = m.Dp <= 128 ? 128 : qd == 2 ? 256 : qd == 4 ? 512 : 0;
    if (hp && dp) {
      m.Hp = hp; m.Dp = dp;
      if (seg3) {
        hmap.seg = 3 * 16; hmap.seg_p = 4 * 16;
        m.H = m.Hp;  // (the kernels' `unit < H` masks: the model's units are spread over the whole padded vector; the rest stay 0 by themselves)
      }
    }
  }
  h->hid_map_seg = hmap.seg; h->hid_map_seg_p = hmap.seg_p; h->H_model = H;
  m.G = 3 * m.Hp;
  m.lp_stay = std::log(1.0 - d->transition_bias);  // np.log(1 - transition_bias), uisrnn.py:416
  m.lp_sw = std::log(d->transition_bias);
  m.l_alpha = std::log(d->crp_alpha);
  m.lp_new = m.lp_sw + m.l_alpha;
  int rc;
  for (int l = 0; l < depth; ++l) {
    const int K = l == 0 ? D : H, Kp = l == 0 ? m.Dp : m.Hp;
    if ((rc = upload(h, tile_weights(d->gru_weight_ih[l], 3, H, m.Hp, K, Kp, hmap, l == 0 ? HidMap() : hmap), &m.wih[l]))) return bail(rc);
    if ((rc = upload(h, tile_weights(d->gru_weight_hh[l], 3, H, m.Hp, H, m.Hp, hmap, hmap), &m.whh[l]))) return bail(rc);
    if ((rc = upload(h, pad_bias(d->gru_bias_ih[l], 3, H, m.Hp, hmap), &m.bih[l]))) return bail(rc);
    if ((rc = upload(h, pad_bias(d->gru_bias_hh[l], 3, H, m.Hp, hmap), &m.bhh[l]))) return bail(rc);
  }
  if ((rc = upload(h, tile_weights(d->linear_mean1_weight, 1, H, m.Hp, H, m.Hp, hmap, hmap), &m.w1))) return bail(rc);
  if ((rc = upload(h, pad_bias(d->linear_mean1_bias, 1, H, m.Hp, hmap), &m.b1))) return bail(rc);
  if ((rc = upload(h, tile_weights(d->linear_mean2_weight, 1, D, m.Dp, H, m.Hp, HidMap(), hmap), &m.w2))) return bail(rc);
  if ((rc = upload(h, pad_bias(d->linear_mean2_bias, 1, D, m.Dp), &m.b2))) return bail(rc);
  std::vector<float> wgt(m.Dp, 0.0f);
  for (int i = 0; i < D; ++i) wgt[i] = 1.0f / (2.0f * d->sigma2[i]);  // 1 / (2 * sigma2), uisrnn.py:414
  if ((rc = upload(h, wgt, &m.wgt))) return bail(rc);
  std::vector<float> hinit((size_t)depth * m.Hp, 0.0f);
  for (int l = 0; l < depth; ++l)
    for (int j = 0; j < H; ++j) hinit[(size_t)l * m.Hp + hmap(j)] = d->rnn_init_hidden[(size_t)l * H + j];
  const float* d_hinit = nullptr;
  if ((rc = upload(h, hinit, &d_hinit))) return bail(rc);
  if ((rc = upload(h, std::vector<float>(m.Dp, 0.0f), &m.m0))) return bail(rc);
  if ((rc = upload(h, std::vector<float>((size_t)depth * m.Hp, 0.0f), &m.h1))) return bail(rc);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_select_fast), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wt_gru<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wt_gru<32, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wt_head<32, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wt_head<32, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_window<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_window<UIS_WINDOW_WIDE_NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_select), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return bail(fail(UIS_ERR_HIP, std::string("hipFuncSetAttribute(k_select): ") + hipGetErrorString(e)));
  if ((rc = bootstrap_constants(h, d_hinit))) return bail(rc);
  *out = h;
  return UIS_OK;
}

UIS_EXPORT int32_t uis_decode_device(uis_handle* h, const float* d_frames, const int64_t* offsets, int32_t n_utt,
                                     const uis_decode_opts* opts, int32_t* d_labels_out, float* d_scores_out,
                                     uis_stats* stats) {
  return decode_impl(h, d_frames, offsets, n_utt, opts, d_labels_out, d_scores_out, stats);
}

UIS_EXPORT int32_t uis_decode(uis_handle* h, const float* frames, const int64_t* offsets, int32_t n_utt,
                              const uis_decode_opts* opts, int32_t* labels_out, float* scores_out, uis_stats* stats) {
  if (!h || !offsets || n_utt < 0) return fail(UIS_ERR_INVALID_ARG, "null handle/offsets or negative n_utt");
  const int64_t F = n_utt ? offsets[n_utt] : 0;
  if (F < 0) return fail(UIS_ERR_INVALID_ARG, "offsets must be non-decreasing");
  if (F > 0 && (!frames || !labels_out)) return fail(UIS_ERR_INVALID_ARG, "frames/labels_out is null");
  HIPCHK(hipSetDevice(h->device));
  int rc;
  if ((rc = h->io_frames.ensure((size_t)std::max<int64_t>(F, 1) * h->m.D * 4))) return rc;
  if ((rc = h->io_labels.ensure((size_t)std::max<int64_t>(F, 1) * 4))) return rc;
  if ((rc = h->io_scores.ensure((size_t)std::max(n_utt, 1) * 4))) return rc;
  h->io_offsets.clear();
  rc = decode_impl(h, h->io_frames.as<float>(), offsets, n_utt, opts, h->io_labels.as<int32_t>(),
                   h->io_scores.as<float>(), stats, frames);
  if (rc != UIS_OK && rc != UIS_ERR_CLUSTER_CAP) return rc;
  if (rc == UIS_OK) h->io_offsets.assign(offsets, offsets + n_utt + 1);
  // uis_decode_f64 (the caller's label array is ordinary pageable memory, a numpy array): the labels
  // come down into a pinned block and are copied out by the CPU -- a device-to-pageable copy is staged
  // by the runtime in small pieces
  int32_t* lab_dst = labels_out;
  float* sc_dst = scores_out;
  if (h->src64) {
    const size_t need = (size_t)std::max<int64_t>(F, 1) * 4 + (size_t)std::max(n_utt, 1) * 4;
    if (need > h->h_out_cap) {
      if (h->h_out) { (void)hipHostFree(h->h_out); h->h_out = nullptr; h->h_out_cap = 0; }
      void* p = nullptr;
      if (hipHostMalloc(&p, need, hipHostMallocDefault) == hipSuccess) { h->h_out = p; h->h_out_cap = need; }
      else (void)hipGetLastError();  // (no pinned memory to be had: straight into the caller's arrays)
    }
    if (h->h_out) {
      lab_dst = static_cast<int32_t*>(h->h_out);
      sc_dst = reinterpret_cast<float*>(static_cast<char*>(h->h_out) + (size_t)std::max<int64_t>(F, 1) * 4);
    }
  }
  if (F > 0) HIPCHK(hipMemcpyAsync(lab_dst, h->io_labels.p, (size_t)F * 4, hipMemcpyDeviceToHost, h->stream));
  if (scores_out && n_utt > 0)
    HIPCHK(hipMemcpyAsync(sc_dst, h->io_scores.p, (size_t)n_utt * 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (lab_dst != labels_out) {
    if (F > 0) memcpy(labels_out, lab_dst, (size_t)F * 4);
    if (scores_out && n_utt > 0) memcpy(scores_out, sc_dst, (size_t)n_utt * 4);
  }
  return rc;
}

UIS_EXPORT int32_t uis_decode_f64(uis_handle* h, const double* const* utterances, const int64_t* n_frames, int32_t n_utt,
                                  const uis_decode_opts* opts, int32_t* labels_out, float* scores_out, uis_stats* stats) {
  if (!h || n_utt < 0 || (n_utt > 0 && (!utterances || !n_frames)))
    return fail(UIS_ERR_INVALID_ARG, "null handle/utterances/n_frames or negative n_utt");
  std::vector<int64_t> offsets((size_t)n_utt + 1, 0);
  for (int u = 0; u < n_utt; ++u) {
    if (n_frames[u] < 0 || (n_frames[u] > 0 && !utterances[u])) return fail(UIS_ERR_INVALID_ARG, "negative n_frames or null utterance");
    offsets[u + 1] = offsets[u] + n_frames[u];
  }
  const int64_t F = offsets[n_utt];
  const size_t need = (size_t)std::max<int64_t>(F, 1) * h->m.D * 4;
  HIPCHK(hipSetDevice(h->device));
  if (need > h->h_cast_cap) {  // grow only, like the device workspace
    if (h->h_cast) { (void)hipHostFree(h->h_cast); h->h_cast = nullptr; h->h_cast_cap = 0; }
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, need, hipHostMallocDefault);
    if (e != hipSuccess) return fail(UIS_ERR_OOM, std::string("hipHostMalloc (float32 staging): ") + hipGetErrorString(e));
    h->h_cast = static_cast<float*>(p);
    h->h_cast_cap = need;
  }
  h->src64 = utterances;
  const int rc = uis_decode(h, h->h_cast, offsets.data(), n_utt, opts, labels_out, scores_out, stats);
  h->src64 = nullptr;
  return rc;
}

UIS_EXPORT int32_t uis_model_constants(uis_handle* h, float* m0_out, float* h1_out) {
  if (!h) return fail(UIS_ERR_INVALID_ARG, "null handle");
  const DevModel& m = h->m;
  HIPCHK(hipSetDevice(h->device));
  if (m0_out) {
    std::vector<float> tmp(m.Dp);
    HIPCHK(hipMemcpy(tmp.data(), m.m0, (size_t)m.Dp * 4, hipMemcpyDeviceToHost));
    memcpy(m0_out, tmp.data(), (size_t)m.D * 4);
  }
  if (h1_out) {
    std::vector<float> tmp((size_t)m.depth * m.Hp);
    HIPCHK(hipMemcpy(tmp.data(), m.h1, tmp.size() * 4, hipMemcpyDeviceToHost));
    const HidMap hmap{h->hid_map_seg, h->hid_map_seg_p};
    for (int l = 0; l < m.depth; ++l)
      for (int j = 0; j < h->H_model; ++j) h1_out[(size_t)l * h->H_model + j] = tmp[(size_t)l * m.Hp + hmap(j)];
  }
  return UIS_OK;
}

UIS_EXPORT int32_t uis_rnn_step(uis_handle* h, const float* x, const float* h_in, float* mean_out, float* h_out) {
  if (!h || !x || !h_in || !mean_out || !h_out) return fail(UIS_ERR_INVALID_ARG, "null argument");
  const DevModel& m = h->m;
  HIPCHK(hipSetDevice(h->device));
  const size_t hid_elems = (size_t)m.depth * m.Hp;
  std::vector<float> xp(m.Dp, 0.0f), hp(hid_elems, 0.0f), mo(m.Dp), ho(hid_elems);
  memcpy(xp.data(), x, (size_t)m.D * 4);
  const HidMap hmap{h->hid_map_seg, h->hid_map_seg_p};
  for (int l = 0; l < m.depth; ++l)
    for (int j = 0; j < h->H_model; ++j) hp[(size_t)l * m.Hp + hmap(j)] = h_in[(size_t)l * h->H_model + j];
  float *d_x = nullptr, *d_h = nullptr, *d_m = nullptr, *d_o = nullptr;
  Scratch tmp;
  int rc;
  if ((rc = tmp.get(&d_x, xp.size())) || (rc = tmp.get(&d_h, hp.size())) || (rc = tmp.get(&d_m, mo.size())) ||
      (rc = tmp.get(&d_o, ho.size())))
    return rc;
  HIPCHK(hipMemcpy(d_x, xp.data(), xp.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_h, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
  if ((rc = rnn_step_once(h, d_x, d_h, d_m, d_o))) return rc;
  HIPCHK(hipMemcpy(mo.data(), d_m, mo.size() * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(ho.data(), d_o, ho.size() * 4, hipMemcpyDeviceToHost));
  memcpy(mean_out, mo.data(), (size_t)m.D * 4);
  for (int l = 0; l < m.depth; ++l)
    for (int j = 0; j < h->H_model; ++j) h_out[(size_t)l * h->H_model + j] = ho[(size_t)l * m.Hp + hmap(j)];
  return UIS_OK;
}

UIS_EXPORT int32_t uis_last_decode_shape(uis_handle* h, int32_t* n_utt_out, int32_t* beam_size_out) {
  if (!h) return fail(UIS_ERR_INVALID_ARG, "null handle");
  if (n_utt_out) *n_utt_out = h->last_U;
  if (beam_size_out) *beam_size_out = h->last_B;
  return UIS_OK;
}

UIS_EXPORT int32_t uis_last_decode_info(uis_handle* h, int32_t* overflow_out, float* beam_scores_out) {
  if (!h) return fail(UIS_ERR_INVALID_ARG, "null handle");
  if (overflow_out && h->last_U) memcpy(overflow_out, h->last_overflow.data(), (size_t)h->last_U * 4);
  if (beam_scores_out && h->last_U) memcpy(beam_scores_out, h->last_beam_scores.data(), (size_t)h->last_U * h->last_B * 4);
  return UIS_OK;
}

UIS_EXPORT int32_t uis_debug_scores(uis_handle* h, float* scores_out, int64_t capacity) {
  if (!h || !scores_out) return fail(UIS_ERR_INVALID_ARG, "null argument");
  if (!h->dbg_floats) return fail(UIS_ERR_INVALID_ARG, "the last decode kept no candidate scores (UIS_FLAG_DEBUG_SCORES)");
  if (capacity < (int64_t)h->dbg_floats)
    return fail(UIS_ERR_INVALID_ARG, "uis_debug_scores: " + std::to_string((long long)h->dbg_floats) + " floats needed");
  if (hipSetDevice(h->device) != hipSuccess) return fail(UIS_ERR_HIP, "hipSetDevice");
  HIPCHK(hipMemcpy(scores_out, h->dbg_scores.p, h->dbg_floats * 4, hipMemcpyDeviceToHost));
  return UIS_OK;
}


// ------------------------------------------------------------------ streaming
//
// Online decoding (the caller side of the path -- UIS-RNN is an online model, the
// reference only offers offline predict()).  A session keeps the beam, the cluster-state pool
// and the back-pointers of n_utt utterances on the device; uis_stream_push() appends frames (any
// number per utterance, also none) and advances every utterance by the frames it received;
// uis_stream_labels() reads the best hypothesis' labels for everything received so far.
// Semantics = predict_single with test_iteration 1 (uisrnn.py:479-562): pushing an utterance
// in any chunking gives bit for bit the labels / scores of one uis_decode over the whole of it
// (tests/test_gpu_stream.py).  look_ahead 1.  A push of four or more steps runs as ONE launch of
// k_decode_resident where that kernel applies, shorter pushes on the launch-per-step kernels.

namespace {

void stream_free(uis_handle* h) {
  uis_handle::Stream& ss = h->stream_state;
  for (void* p : ss.allocs) (void)hipFree(p);
  ss.allocs.clear();
  DevBuf* bufs[] = {&ss.chunk_x, &ss.chunk_pad, &ss.chunk_gi0, &ss.chunk_mse0, &ss.labels, &ss.scores};
  for (DevBuf* b : bufs) b->release();
  if (ss.h_stage) { (void)hipHostFree(ss.h_stage); ss.h_stage = nullptr; ss.h_stage_cap = 0; }
  if (ss.pm_block) { (void)hipHostFree(ss.pm_block); ss.pm_block = nullptr; }
  ss.persist = false; ss.pm_running = false;
  ss.active = false;
  ss.have.clear();
}

template <typename T>
int stream_alloc(uis_handle* h, T** out, size_t count, bool zero = false) {
  void* p = nullptr;
  const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) return fail(UIS_ERR_OOM, "hipMalloc of " + std::to_string(bytes) + " bytes failed: " + hipGetErrorString(e));
  h->stream_state.allocs.push_back(p);
  if (zero) HIPCHK(hipMemsetAsync(p, 0, bytes, h->stream));
  *out = static_cast<T*>(p);
  return UIS_OK;
}

// ---- the persistent launch of a UIS_FLAG_PERSISTENT session
//
// Mailbox protocol (pm_block, host-coherent pinned memory; uint32 view, one 64-byte line per item,
// uis_kernels.h UIS_PM_*_WORD): a doorbell line per cluster {sequence number, command | frames << 8,
// first row | rows << 16} whose sequence number the host writes LAST (release) and rank 0 of the
// cluster polls with one 16-byte read; the sequence number of the last command each cluster
// completed; a word per cluster that turns non-zero when the cluster has left the kernel.
// While the launch is on the device the host makes NO HIP call that could wait for the device:
// everything a command needs was allocated by uis_stream_begin.

double pm_now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

unsigned long long pm_idle_ticks() {
  double ms = 50.0;  // without a command for this long the launch ends by itself (the next push starts a new one)
  if (const char* e = getenv("UIS_PERSIST_IDLE_MS")) ms = atof(e);
  ms = std::min(std::max(ms, 0.05), 2000.0);
  return (unsigned long long)(ms * 1e5);  // s_memrealtime ticks of 10 ns
}

// After the launch has ended (every cluster left, or an in-launch barrier gave up): look at the abort word.
int pm_reap(uis_handle* h) {
  uis_handle::Stream& ss = h->stream_state;
  // (a launch that is really stuck must not take the caller with it: poll with a deadline instead
  // of an unbounded hipStreamSynchronize; the kernel's own barrier time-out is ~1 s)
  {
    const double t0 = pm_now_s();
    hipError_t q;
    while ((q = hipStreamQuery(h->stream)) == hipErrorNotReady) {
      if (pm_now_s() - t0 > 15.0) {
        ss.persist = false;
        h->resident_off = true;
        return fail(UIS_ERR_HIP, "the persistent streaming launch does not leave the device (15 s); the handle's "
                                 "stream is unusable -- destroy the handle");
      }
      __builtin_ia32_pause();
    }
    if (q != hipSuccess) return fail(UIS_ERR_HIP, std::string("hipStreamQuery: ") + hipGetErrorString(q));
  }
  ss.pm_running = false;
  uint32_t abort_word = 0;
  HIPCHK(hipMemcpy(&abort_word, ss.d_ctl + 16, 4, hipMemcpyDeviceToHost));
  if (abort_word) {
    ss.persist = false;
    h->resident_off = true;
    return fail(UIS_ERR_HIP, "in-launch barrier failed inside the persistent streaming launch; close the session "
                             "(uis_stream_end) and reopen it with UIS_FLAG_STEPWISE");
  }
  return UIS_OK;
}

int pm_launch(uis_handle* h) {
  uis_handle::Stream& ss = h->stream_state;
  const DevModel& m = h->m;
  DecodeState st = ss.st;
  // the mailbox as the device sees it (the same address under unified addressing; asked for anyway)
  void* blk_dev = nullptr;
  HIPCHK(hipHostGetDevicePointer(&blk_dev, ss.pm_block, 0));
  unsigned char* blk = static_cast<unsigned char*>(blk_dev);
  st.x = reinterpret_cast<const float*>(ss.chunk_x.as<char>());
  st.gi0 = ss.chunk_gi0.as<float>();
  st.mse0 = ss.chunk_mse0.as<float>();
  st.push_F = 0;
  PersistArgs& pa = ss.pm_args;
  pa.ctl = reinterpret_cast<uint32_t*>(blk);
  pa.foff = reinterpret_cast<const int64_t*>(blk + ss.pm_o_foff);
  pa.avail = reinterpret_cast<const int32_t*>(blk + ss.pm_o_avail);
  pa.lab_off = reinterpret_cast<const int64_t*>(blk + ss.pm_o_laboff);
  pa.frames = reinterpret_cast<const float*>(blk + ss.pm_o_frames);
  pa.labels = reinterpret_cast<int32_t*>(blk + ss.pm_o_labels);
  pa.scores = reinterpret_cast<float*>(blk + ss.pm_o_scores);
  pa.beam_scores = reinterpret_cast<float*>(blk + ss.pm_o_bscores);
  pa.overflow = reinterpret_cast<int32_t*>(blk + ss.pm_o_overflow);
  pa.go = ss.d_go;
  pa.hdr = ss.d_hdr;
  pa.hdr_stride = ss.hdr_stride;
  pa.idle_ticks = pm_idle_ticks();
  HIPCHK(hipMemcpyAsync(ss.d_pm_args, &pa, sizeof(pa), hipMemcpyHostToDevice, h->stream));
  st.pm = ss.d_pm_args;
  // avail / foff only have to be non-null here (the kernel points them at its cluster's copies)
  st.avail = reinterpret_cast<const int32_t*>(ss.d_hdr);
  st.foff = reinterpret_cast<const int64_t*>(ss.d_hdr);
  HIPCHK(hipMemsetAsync(ss.d_ctl, 0, ss.ctl_words * 4, h->stream));
  HIPCHK(hipMemsetAsync(ss.d_go, 0, (size_t)UIS_PM_MAX_CLUSTERS * 128, h->stream));
  HIPCHK(hipMemsetAsync(ss.st.nrows, 0, 8, h->stream));
  Launcher lch{h, h->stream, false};
  const size_t shmem = std::max<size_t>(resident_lds_bytes(m.Hp, m.Dp, ss.B, ss.Kmax, ss.S), 96 * 1024);
  int rc = UIS_ERR_UNSUPPORTED;
  h->inlaunch_failed = false;
#define UIS_PERSIST_CASE(HPV, DPV)                                                                                    \
  if (m.Hp == HPV && m.Dp == DPV) {                                                                                  \
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decode_resident<HPV, DPV, true>),                   \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));                             \
    rc = lch.run_cooperative(UIS_K_GRU, &k_decode_resident<HPV, DPV, true>, h->n_cu, dim3(32 * st.ncl), dim3(512), \
                             shmem, m, st, true);                                                                    \
  }
  UIS_PERSIST_CASE(512, 256)
  UIS_PERSIST_CASE(512, 512)
  UIS_PERSIST_CASE(256, 256)
#undef UIS_PERSIST_CASE
  if (rc) return rc;
  ss.pm_running = true;
  ss.pm_launches += 1;
  return UIS_OK;
}

// Issue one command and wait until every cluster has completed it.  A launch that is not on the
// device (never started, or left because it was idle) is started first when `may_launch`; a
// launch that left while the command was on its way is reaped and the command issued again to a
// new one -- harmless: a cluster that did take a push has nothing left to do for it.
// Returns UIS_OK, an error, or 1 = not running and may_launch was false.
int pm_command(uis_handle* h, uint32_t type, uint32_t frames, bool may_launch, const uint32_t* row0 = nullptr,
               const uint32_t* nrow = nullptr) {
  uis_handle::Stream& ss = h->stream_state;
  volatile uint32_t* ctl = pm_ctl(ss);
  const int ncl = ss.st.ncl;
  for (int attempt = 0; attempt < 3; ++attempt) {
    int rc;
    if (!ss.pm_running) {
      if (!may_launch) return 1;
      for (int i = 0; i < UIS_PM_CTL_WORDS; ++i) ctl[i] = 0;
      ss.pm_seq = 1;
      pm_ring(ss, 1, type, frames, row0, nrow);
      if ((rc = pm_launch(h))) return rc;
    } else {
      ss.pm_seq += 1;
      pm_ring(ss, ss.pm_seq, type, frames, row0, nrow);
    }
    ss.pm_commands += 1;
    const double t0 = pm_now_s();
    bool left = false;
    unsigned spins = 0;
    for (;;) {
      bool all = true;
      for (int c = 0; c < ncl; ++c) all = all && ctl[UIS_PM_DONE_WORD + 16 * c] == ss.pm_seq;
      if (all) return UIS_OK;
      for (int c = 0; c < ncl; ++c) left = left || ctl[UIS_PM_LEFT_WORD + 16 * c] != 0;
      if (left) break;
      if ((++spins & 4095u) == 0) {
        if (pm_now_s() - t0 > 10.0) break;
        // (a launch that ended without saying so -- an in-launch barrier gave up -- is noticed here)
        if ((spins & 0xffffu) == 0 && hipStreamQuery(h->stream) == hipSuccess) { left = true; break; }
      }
      __builtin_ia32_pause();
    }
    // a cluster left before (or instead of) completing the command: tell the others to leave too
    // (they complete this command first if they had not seen it yet), then look at what happened
    ss.pm_seq += 1;
    pm_ring(ss, ss.pm_seq, UIS_PM_QUIT, 0);
    if ((rc = pm_reap(h))) return rc;
    if (!left) return fail(UIS_ERR_HIP, "the persistent streaming launch did not answer within 10 s");
  }
  return fail(UIS_ERR_HIP, "the persistent streaming launch kept leaving before it took the command");
}

int pm_quit(uis_handle* h) {
  uis_handle::Stream& ss = h->stream_state;
  if (!ss.pm_running) return UIS_OK;
  ss.pm_seq += 1;
  pm_ring(ss, ss.pm_seq, UIS_PM_QUIT, 0);
  return pm_reap(h);
}

}  // namespace

UIS_EXPORT int32_t uis_stream_begin(uis_handle* h, int32_t n_utt, const uis_decode_opts* opts, int64_t max_frames) {
  if (!h || !opts || n_utt < 1 || max_frames < 1) return fail(UIS_ERR_INVALID_ARG, "null handle/opts, n_utt < 1 or max_frames < 1");
  uis_handle::Stream& ss = h->stream_state;
  if (ss.active) return fail(UIS_ERR_INVALID_ARG, "a streaming session is already open on this handle");
  const DevModel& m = h->m;
  const int B = opts->beam_size;
  const int Kmax = opts->max_clusters > 0 ? opts->max_clusters : 16;
  if (B < 1 || B > 256) return fail(UIS_ERR_UNSUPPORTED, "beam_size must be in [1, 256]");
  if (opts->look_ahead != 1) return fail(UIS_ERR_UNSUPPORTED, "streaming needs look_ahead 1");
  if (opts->test_iteration != 1) return fail(UIS_ERR_UNSUPPORTED, "streaming is online decoding: test_iteration must be 1");
  if (Kmax > 4096) return fail(UIS_ERR_UNSUPPORTED, "max_clusters must be <= 4096");
  if (max_frames > 0x7fffff00LL) return fail(UIS_ERR_UNSUPPORTED, "max_frames too large");
  const int U = n_utt, S = B * Kmax + B;
  const SelectLds lds = select_lds_layout(m.Dp, B, Kmax, S);
  if (lds.total > 160 * 1024) return fail(UIS_ERR_UNSUPPORTED, "beam_size * max_clusters too large for the select kernel's LDS budget");
  const double bytes = (double)U * S * (m.Dp + (double)m.depth * m.Hp) * 4.0 + (double)U * max_frames * B * 4.0;
  if (bytes > 200e9) return fail(UIS_ERR_OOM, "streaming state would need " + std::to_string((long long)(bytes / 1e9)) + " GB");
  HIPCHK(hipSetDevice(h->device));
  ss = uis_handle::Stream{};
  ss.U = U; ss.B = B; ss.Kmax = Kmax; ss.S = S; ss.cap = max_frames;
  ss.have.assign(U, 0);
  DecodeState& st = ss.st;
  st.U = U; st.B = B; st.Kmax = Kmax; st.S = S; st.L = 1; st.tau = 1; st.flags = opts->flags | (agent_flags_env() ? UIS_FLAG_AGENT_FLAGS : 0u);
  st.max_rows = U * B;
  // a push advances the session with ONE launch of the resident decode kernel where that kernel
  // applies (same conditions as uis_decode); UIS_FLAG_STEPWISE keeps the four kernels per step
  const int ncl = (h->n_cu >= 32 && h->n_cu % 32 == 0 && h->n_cu / 32 <= UIS_MAX_CLUSTERS) ? h->n_cu / 32 : 0;
  const int nclq = std::max(ncl, 1);
  const int rx_stride = (int)(((((long)U + nclq - 1) / nclq) * B + 15) / 16 * 16);
  const long rows_cap = std::max((long)U * B + 48, (long)nclq * rx_stride);
  ss.resident = m.depth == 1 && (m.Hp == 256 || m.Hp == 512) && (m.Dp == 128 || m.Dp == 256 || m.Dp == 512) &&
                select_fast_ok(B, Kmax, S) && ncl >= 1 && !(opts->flags & (UIS_FLAG_STEPWISE | UIS_FLAG_GENERIC_SELECT)) &&
                ((double)U * S + 1) * m.Hp * 4.0 < 2.0e9 && (double)rows_cap * m.Hp * 4.0 < 2.0e9 &&
                (double)U * S * m.Dp * 4.0 < 2.0e9 && resident_lds_bytes(m.Hp, m.Dp, B, Kmax, S) <= 160 * 1024;
  if ((opts->flags & UIS_FLAG_RESIDENT) && !ss.resident)
    { ss = uis_handle::Stream{}; return fail(UIS_ERR_UNSUPPORTED, "UIS_FLAG_RESIDENT: the one-launch decode does not apply to this session's shape"); }
  int rc = UIS_OK;
  int64_t* d_off = nullptr; double *d_logblk = nullptr, *d_logden = nullptr;
#define SALLOC(ptr, count, zero) if ((rc = stream_alloc(h, &(ptr), (size_t)(count), zero))) { stream_free(h); return rc; }
  SALLOC(d_off, U + 1, false);
  SALLOC(st.utt_step, U, false);
  SALLOC(st.overflow, U, false);
  SALLOC(ss.d_avail, U, true);
  SALLOC(ss.d_have, U, true);
  SALLOC(ss.d_foff, U, true);
  SALLOC(ss.d_lab_off, U, true);
  SALLOC(d_logblk, max_frames + 2, false);
  SALLOC(d_logden, max_frames + 2, false);
  SALLOC(st.pool_mean, (size_t)U * S * m.Dp, false);
  SALLOC(st.pool_hid, ((size_t)U * S + 1) * m.depth * m.Hp, false);
  SALLOC(st.pool_cnt, (size_t)U * S, false);
  SALLOC(st.beam_n, 2 * (size_t)U, false);
  SALLOC(st.beam_K, 2 * (size_t)U * B, false);
  SALLOC(st.beam_last, 2 * (size_t)U * B, false);
  SALLOC(st.beam_sum, 2 * (size_t)U * B, false);
  SALLOC(st.beam_score, 2 * (size_t)U * B, false);
  SALLOC(st.beam_slot, 2 * (size_t)U * B * Kmax, false);
  SALLOC(st.beam_blk, 2 * (size_t)U * B * Kmax, false);
  SALLOC(st.bp, (size_t)U * max_frames * B, false);
  SALLOC(st.rows, rows_cap, true);
  SALLOC(st.nrows, 2, true);
  SALLOC(st.gi_up, m.depth > 1 ? (size_t)rows_cap * m.G : (size_t)rows_cap * m.Hp, false);  // depth 1: the resident kernel's h' staging
  SALLOC(st.a1, (size_t)rows_cap * m.Hp, true);
  SALLOC(st.counters, 96, true);
  ss.ctl_words = (size_t)32 + 3 * UIS_MAX_CLUSTERS * 32;
  SALLOC(ss.d_ctl, ss.ctl_words, true);
  st.cl_abort = ss.d_ctl + 16;
  if (ss.resident) {
    st.ncl = ncl;
    st.cl_xcc = ss.d_ctl;
    st.rx_stride = rx_stride;
    st.rx_nrows = reinterpret_cast<int32_t*>(ss.d_ctl) + 32;
    st.rx_bar = ss.d_ctl + 32 + UIS_MAX_CLUSTERS * 32;
    st.rx_flags = ss.d_ctl + 32 + 2 * UIS_MAX_CLUSTERS * 32;
  }
  SALLOC(ss.d_beam_scores, (size_t)U * B, false);
  if (opts->flags & UIS_FLAG_PERSISTENT) {
    // the launch that stays: needs the one-launch shape with the beam in LDS (at most one utterance
    // per workgroup), unpadded frames, and a mailbox that holds every label of the session
    const bool shape = (m.Hp == 512 && (m.Dp == 256 || m.Dp == 512)) || (m.Hp == 256 && m.Dp == 256);
    const double label_bytes = (double)U * (double)max_frames * 4.0;
    if (!(ss.resident && shape && U <= 32 * ncl && m.D == m.Dp && label_bytes <= 256e6)) {
      stream_free(h);
      return fail(UIS_ERR_UNSUPPORTED, "UIS_FLAG_PERSISTENT needs the one-launch shape (rnn_depth 1, rnn_hidden_size 512 with "
                                       "observation_dim 256 / 512 or 256 with 256, unpadded), at most one utterance per compute "
                                       "unit and n_utt * max_frames <= 64 M labels");
    }
    // Every cluster gets a FIXED row range of the chunk buffers (x, gi0, mse0) and of the mailbox's
    // frame area: room for 16 frames of each of its utterances.  Fixed, because the launch never
    // ends between pushes: a row that changed hands from one push to the next would leave a stale
    // dirty line in the previous owner's XCD-private L2, free to be written back over the new
    // owner's data at any time (seen as rare score differences before the ranges were fixed).
    ss.pm_cluster_rows = std::min<int64_t>(round_up(((U + ncl - 1) / ncl) * 16, 32), (int64_t)UIS_RES_HEAD_TILES * 16 * 6);
    ss.pm_cap_frames = ss.pm_cluster_rows * ncl;
    size_t o = (size_t)UIS_PM_CTL_WORDS * 4;
    auto take = [&](size_t bytes) { o = (o + 127) & ~(size_t)127; const size_t r = o; o += bytes; return r; };
    ss.pm_o_foff = take((size_t)U * 8);
    ss.pm_o_avail = take((size_t)U * 4);
    ss.pm_o_laboff = take((size_t)U * 8);
    ss.pm_o_scores = take((size_t)U * 4);
    ss.pm_o_bscores = take((size_t)U * B * 4);
    ss.pm_o_overflow = take((size_t)U * 4);
    ss.pm_o_frames = take((size_t)ss.pm_cap_frames * m.D * 4);
    ss.pm_o_labels = take((size_t)U * (size_t)max_frames * 4);
    void* blk = nullptr;
    hipError_t e = hipHostMalloc(&blk, o, hipHostMallocMapped | hipHostMallocCoherent);
    if (e != hipSuccess) { stream_free(h); return fail(UIS_ERR_OOM, std::string("hipHostMalloc (mailbox): ") + hipGetErrorString(e)); }
    memset(blk, 0, o);
    ss.pm_block = static_cast<unsigned char*>(blk);
    ss.hdr_stride = (((size_t)U * 12) + 127) & ~(size_t)127;
    SALLOC(ss.d_go, (size_t)UIS_PM_MAX_CLUSTERS * 16, true);
    SALLOC(ss.d_hdr, (size_t)ncl * ss.hdr_stride, true);
    SALLOC(ss.d_pm_args, 1, false);
    // everything a push through the mailbox touches, now: no allocation while the launch is resident
    if ((rc = ss.chunk_x.ensure((size_t)U * 16 + (size_t)ss.pm_cap_frames * m.Dp * 4)) ||
        (rc = ss.chunk_gi0.ensure((size_t)ss.pm_cap_frames * m.G * 4)) || (rc = ss.chunk_mse0.ensure((size_t)ss.pm_cap_frames * 4))) {
      stream_free(h);
      return rc;
    }
    ss.persist = true;
  }
#undef SALLOC
  if (ss.resident)  // the extra slot every GRU source row of a fresh cluster reads
    HIPCHK(hipMemcpyAsync(st.pool_hid + (size_t)U * S * m.Hp, m.h1, (size_t)m.Hp * 4, hipMemcpyDeviceToDevice, h->stream));
  std::vector<int64_t> off(U + 1);
  for (int u = 0; u <= U; ++u) off[u] = (int64_t)u * max_frames;  // capacity offsets: they address the back-pointers
  std::vector<double> logblk(max_frames + 2), logden(max_frames + 2);
  for (int64_t n = 0; n < max_frames + 2; ++n) {
    logblk[n] = n > 0 ? std::log((double)n) : 0.0;
    logden[n] = std::log((double)n + h->alpha);
  }
  HIPCHK(hipMemcpyAsync(d_off, off.data(), off.size() * 8, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(d_logblk, logblk.data(), logblk.size() * 8, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(d_logden, logden.data(), logden.size() * 8, hipMemcpyHostToDevice, h->stream));
  st.off = d_off; st.logblk = d_logblk; st.logden = d_logden;
  st.avail = ss.d_avail; st.foff = ss.d_foff; st.lab_off = ss.d_lab_off;
  hipLaunchKernelGGL(k_init_state, dim3((U + 255) / 256), dim3(256), 0, h->stream, st);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));  // the host vectors above go out of scope
  ss.active = true;
  return UIS_OK;
}

UIS_EXPORT int32_t uis_stream_push(uis_handle* h, const float* frames, const int32_t* counts) {
  if (!h || !counts) return fail(UIS_ERR_INVALID_ARG, "null handle/counts");
  uis_handle::Stream& ss = h->stream_state;
  if (!ss.active) return fail(UIS_ERR_INVALID_ARG, "no streaming session (uis_stream_begin first)");
  const DevModel& m = h->m;
  const int U = ss.U;
  int64_t F = 0, max_new = 0;
  for (int u = 0; u < U; ++u) {
    if (counts[u] < 0) return fail(UIS_ERR_INVALID_ARG, "negative frame count");
    if ((int64_t)ss.have[u] + counts[u] > ss.cap) return fail(UIS_ERR_INVALID_ARG, "utterance exceeds the session's max_frames");
    F += counts[u];
    max_new = std::max<int64_t>(max_new, counts[u]);
  }
  if (F == 0) return UIS_OK;
  if (!frames) return fail(UIS_ERR_INVALID_ARG, "frames is null");
  HIPCHK(hipSetDevice(h->device));
  int rc;
#if defined(UIS_PM_TIMING)
  const double t_enter = pm_now_s();
#endif
  bool pm_fits = ss.persist && !h->resident_off;
  if (pm_fits) {  // every cluster's new frames must fit its fixed row range
    const int ncl = ss.st.ncl;
    for (int c = 0; c < ncl && pm_fits; ++c) {
      int64_t rows = 0;
      for (int u = c; u < U; u += ncl) rows += counts[u];
      pm_fits = rows <= ss.pm_cluster_rows;
    }
  }
  if (pm_fits) {
    // ---- the launch that stays on the device: tables and frames into the mailbox, ring, wait
    int64_t* p_foff = reinterpret_cast<int64_t*>(ss.pm_block + ss.pm_o_foff);
    int32_t* p_avail = reinterpret_cast<int32_t*>(ss.pm_block + ss.pm_o_avail);
    // frames cluster by cluster (cluster c owns utterances c, c + ncl, ...): each cluster's rank 0
    // fetches ONE contiguous row range
    const int ncl = ss.st.ncl;
    uint32_t row0[UIS_PM_MAX_CLUSTERS + 1];
    std::vector<int64_t> src(U + 1, 0);  // where utterance u's frames start in the caller's buffer
    for (int u = 0; u < U; ++u) src[u + 1] = src[u] + counts[u];
    uint32_t nrow[UIS_PM_MAX_CLUSTERS];
    float* dst = reinterpret_cast<float*>(ss.pm_block + ss.pm_o_frames);
    for (int c = 0; c < ncl; ++c) {
      int64_t pos = (int64_t)c * ss.pm_cluster_rows;  // the cluster's fixed range
      row0[c] = (uint32_t)pos;
      for (int u = c; u < U; u += ncl) {
        p_foff[u] = pos - ss.have[u];
        p_avail[u] = ss.have[u] + counts[u];
        if (counts[u]) memcpy(dst + (size_t)pos * m.D, frames + (size_t)src[u] * m.D, (size_t)counts[u] * m.D * 4);
        pos += counts[u];
      }
      nrow[c] = (uint32_t)(pos - (int64_t)c * ss.pm_cluster_rows);
    }
    h->inlaunch_failed = false;
#if defined(UIS_PM_TIMING)
    static double fill_s = 0.0, wait_s = 0.0; static long n_push = 0;
    const double t_mid = pm_now_s();
    fill_s += t_mid - t_enter;
#endif
    rc = pm_command(h, UIS_PM_PUSH, (uint32_t)std::min<int64_t>(F, 4095), true, row0, nrow);
#if defined(UIS_PM_TIMING)
    wait_s += pm_now_s() - t_mid;
    if (++n_push % 100 == 0) {
      const volatile unsigned long long* k = reinterpret_cast<const volatile unsigned long long*>(pm_ctl(ss) + UIS_PM_TIMING_WORD);
      const unsigned long long k5 = k[5]; const double n = (double)(k5 ? k5 : 1);
      fprintf(stderr, "[pm timing] host per push: fill %.1f us, ring + wait %.1f us; kernel (workgroup 0) per push: fetch %.1f, pass on %.1f, "
              "count + chunk projection %.1f, steps %.1f us\n", 1e6 * fill_s / n_push, 1e6 * wait_s / n_push, k[0] * 0.01 / n, k[1] * 0.01 / n,
              k[2] * 0.01 / n, k[3] * 0.01 / n);
    }
#endif
    if (rc == UIS_OK) {
      for (int u = 0; u < U; ++u) ss.have[u] += counts[u];
      ss.steps_run += max_new;
      return UIS_OK;
    }
    if (!h->inlaunch_failed) return rc;
    ss.persist = false;  // the cooperative launch was refused: ordinary launches from here on
  }
  if (ss.pm_running && (rc = pm_quit(h))) return rc;
  // ---- one staging block, one H2D: [foff][avail][frames]
  const size_t hdr = (size_t)U * 8 + (((size_t)U * 4 + 15) & ~(size_t)15);
  const size_t need = hdr + (size_t)F * m.D * 4;
  if (need > ss.h_stage_cap) {
    if (ss.h_stage) { (void)hipHostFree(ss.h_stage); ss.h_stage = nullptr; ss.h_stage_cap = 0; }
    const size_t want = need + need / 4 + 4096;
    hipError_t e = hipHostMalloc(&ss.h_stage, want, hipHostMallocDefault);
    if (e != hipSuccess) { ss.h_stage = nullptr; return fail(UIS_ERR_OOM, std::string("hipHostMalloc: ") + hipGetErrorString(e)); }
    ss.h_stage_cap = want;
  }
  if ((rc = ss.chunk_x.ensure(need))) return rc;
  if ((rc = ss.chunk_gi0.ensure((size_t)F * m.G * 4))) return rc;
  if ((rc = ss.chunk_mse0.ensure((size_t)F * 4))) return rc;
  int64_t* h_foff = static_cast<int64_t*>(ss.h_stage);
  int32_t* h_avail = reinterpret_cast<int32_t*>(static_cast<char*>(ss.h_stage) + (size_t)U * 8);
  {
    int64_t pos = 0;
    for (int u = 0; u < U; ++u) {
      h_foff[u] = pos - ss.have[u];  // row of step s's frame in this chunk = foff + s
      pos += counts[u];
      h_avail[u] = ss.have[u] + counts[u];
    }
  }
  memcpy(static_cast<char*>(ss.h_stage) + hdr, frames, (size_t)F * m.D * 4);
  HIPCHK(hipMemcpyAsync(ss.chunk_x.p, ss.h_stage, need, hipMemcpyHostToDevice, h->stream));
  ss.d_foff = ss.chunk_x.as<int64_t>();
  ss.d_avail = reinterpret_cast<int32_t*>(ss.chunk_x.as<char>() + (size_t)U * 8);
  ss.st.foff = ss.d_foff;
  ss.st.avail = ss.d_avail;
  const float* d_x = reinterpret_cast<const float*>(ss.chunk_x.as<char>() + hdr);
  if (m.D != m.Dp) {
    if ((rc = ss.chunk_pad.ensure((size_t)F * m.Dp * 4))) return rc;
    const long total = (long)F * m.Dp;
    hipLaunchKernelGGL(k_pad_frames, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, d_x,
                       ss.chunk_pad.as<float>(), (long)F, m.D, m.Dp);
    HIPCHK(hipGetLastError());
    d_x = ss.chunk_pad.as<float>();
  }
  Launcher lch{h, h->stream, false};
  DecodeState st = ss.st;
  st.x = d_x; st.gi0 = ss.chunk_gi0.as<float>(); st.mse0 = ss.chunk_mse0.as<float>();
  const SelectLds lds = select_lds_layout(m.Dp, ss.B, ss.Kmax, ss.S);
  // One-launch path: the chunk's projection fused into the kernel and plain launches after the
  // session's first push make a push one H2D, one memset and ONE kernel.  Measured
  // (tools/stream_latency.py, profiles/): that kernel re-reads its weights into registers / LDS at
  // every launch (~10 us), so for 1-3 steps per push the four small kernels per step are still
  // quicker (79 vs 96 us for one frame of 64 utterances); from 4 steps on the single launch wins
  // (16 frames: 690 vs 880 us).  UIS_FLAG_RESIDENT forces it, UIS_FLAG_STEPWISE forbids it.
  bool stepwise = !ss.resident || h->resident_off ||
                  (max_new < UIS_STREAM_RESIDENT_MIN_STEPS && !(ss.st.flags & UIS_FLAG_RESIDENT));
  // the chunk's gi0 / mse0: inside the one-launch kernel when the frames need no padding and the
  // chunk's rows fit the kernel's LDS list, else by the two once-per-chunk kernels
  const bool fused = !stepwise && m.D == m.Dp && F <= (int64_t)UIS_RES_HEAD_TILES * 16 * 6;
  if (!fused) {
    LAUNCH(UIS_K_INPUT_PROJ, k_dense_input_proj, dense_grid(F, m.G / 16), dim3(256), 0, m, d_x, ss.chunk_gi0.as<float>(), (long)F);
    LAUNCH(UIS_K_INPUT_PROJ, k_mse0, dim3((unsigned)((F + 3) / 4)), dim3(256), (size_t)5 * m.Dp * 4, m, d_x,
           ss.chunk_mse0.as<float>(), (long)F, 0L, (const long*)nullptr);
  }
  HIPCHK(hipMemsetAsync(ss.st.nrows, 0, 8, h->stream));
  st.push_F = fused ? (int)F : 0;
  bool ran_resident = false;
  if (!stepwise) {
    // every step of this push in ONE launch (the kernel runs max over utterances of
    // avail - utt_step steps; utterances without new frames sit them out)
    HIPCHK(hipMemsetAsync(ss.d_ctl, 0, ss.ctl_words * 4, h->stream));
    const size_t shmem = std::max<size_t>(resident_lds_bytes(m.Hp, m.Dp, ss.B, ss.Kmax, ss.S), 96 * 1024);
    h->inlaunch_failed = false;
    // Every push is a COOPERATIVE launch: the kernel spins on in-launch barriers and needs all its
    // workgroups co-resident, which only that launch path checks against whatever else runs on the
    // device at that moment (another handle's decode, a second session).  A plain launch of the
    // same grid saves 15-19 us of host time per push; it is opt-in (UIS_STREAM_PLAIN_LAUNCH=1) for
    // callers that own the device, and used only after the session's first push went through the
    // cooperative path.
    static const bool plain_ok = getenv("UIS_STREAM_PLAIN_LAUNCH") != nullptr && atoi(getenv("UIS_STREAM_PLAIN_LAUNCH")) != 0;
#define UIS_RESIDENT_CASE(HPV, DPV)                                                                                   \
  if (m.Hp == HPV && m.Dp == DPV) {                                                                                  \
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decode_resident<HPV, DPV>),                         \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));                             \
    rc = lch.run_cooperative(UIS_K_GRU, &k_decode_resident<HPV, DPV>, h->n_cu, dim3(32 * st.ncl), dim3(512), shmem, \
                             m, st, !(ss.coop_checked && plain_ok));                                                 \
  }
    UIS_RESIDENT_CASE(512, 256)
    UIS_RESIDENT_CASE(512, 512)
    UIS_RESIDENT_CASE(512, 128)
    UIS_RESIDENT_CASE(256, 256)
    UIS_RESIDENT_CASE(256, 128)
    UIS_RESIDENT_CASE(256, 512)
#undef UIS_RESIDENT_CASE
    if (rc && h->inlaunch_failed) {  // refused before anything ran: the per-step kernels take over
      h->resident_off = true; stepwise = true;
      if (fused) {  // ... and they need the chunk's gi0 / mse0
        LAUNCH(UIS_K_INPUT_PROJ, k_dense_input_proj, dense_grid(F, m.G / 16), dim3(256), 0, m, d_x, ss.chunk_gi0.as<float>(), (long)F);
        LAUNCH(UIS_K_INPUT_PROJ, k_mse0, dim3((unsigned)((F + 3) / 4)), dim3(256), (size_t)5 * m.Dp * 4, m, d_x,
               ss.chunk_mse0.as<float>(), (long)F, 0L, (const long*)nullptr);
        st.push_F = 0;
      }
    } else if (rc) return rc;
    else { ran_resident = true; ss.coop_checked = true; }
  }
  if (stepwise && (rc = enqueue_steps(h, lch, st, lds.total, (int)max_new))) return rc;
  uint32_t abort_word = 0;
  if (ran_resident) HIPCHK(hipMemcpyAsync(&abort_word, ss.d_ctl + 16, 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));  // the staging block and the caller's frames may be reused
  if (abort_word)  // an in-launch barrier gave up mid-push: the session's state is not trustworthy any more
    return fail(UIS_ERR_HIP, "in-launch barrier failed during uis_stream_push; close the session (uis_stream_end) and reopen it "
                             "with UIS_FLAG_STEPWISE");
  for (int u = 0; u < U; ++u) ss.have[u] += counts[u];
  ss.steps_run += max_new;
  return UIS_OK;
}

UIS_EXPORT int32_t uis_stream_labels(uis_handle* h, int32_t* labels_out, float* scores_out, int32_t* overflow_out) {
  if (!h) return fail(UIS_ERR_INVALID_ARG, "null handle");
  uis_handle::Stream& ss = h->stream_state;
  if (!ss.active) return fail(UIS_ERR_INVALID_ARG, "no streaming session (uis_stream_begin first)");
  const int U = ss.U;
  std::vector<int64_t> lab_off(U);
  int64_t F = 0;
  for (int u = 0; u < U; ++u) { lab_off[u] = F; F += ss.have[u]; }
  if (F > 0 && !labels_out) return fail(UIS_ERR_INVALID_ARG, "labels_out is null");
  HIPCHK(hipSetDevice(h->device));
  int rc;
  if (ss.persist && ss.pm_running) {
    // the resident launch back-traces every utterance and writes into the mailbox
    memcpy(ss.pm_block + ss.pm_o_laboff, lab_off.data(), (size_t)U * 8);
    rc = pm_command(h, UIS_PM_LABELS, 0, false);
    if (rc == UIS_OK) {
      if (F > 0) memcpy(labels_out, ss.pm_block + ss.pm_o_labels, (size_t)F * 4);
      if (scores_out) memcpy(scores_out, ss.pm_block + ss.pm_o_scores, (size_t)U * 4);
      h->last_U = U; h->last_B = ss.B;
      h->last_overflow.assign(reinterpret_cast<const int32_t*>(ss.pm_block + ss.pm_o_overflow),
                              reinterpret_cast<const int32_t*>(ss.pm_block + ss.pm_o_overflow) + U);
      h->last_beam_scores.assign(reinterpret_cast<const float*>(ss.pm_block + ss.pm_o_bscores),
                                 reinterpret_cast<const float*>(ss.pm_block + ss.pm_o_bscores) + (size_t)U * ss.B);
      int n_over = 0;
      for (int u = 0; u < U; ++u) {
        if (overflow_out) overflow_out[u] = h->last_overflow[u];
        n_over += h->last_overflow[u] != 0;
      }
      if (n_over)
        return fail(UIS_ERR_CLUSTER_CAP, std::to_string(n_over) + " utterance(s) needed more than max_clusters=" +
                                             std::to_string(ss.Kmax) + " clusters per hypothesis");
      return UIS_OK;
    }
    if (rc != 1) return rc;  // (1: the launch had left -- its tables are back in global memory)
  }
  if ((rc = ss.labels.ensure((size_t)std::max<int64_t>(F, 1) * 4))) return rc;
  if ((rc = ss.scores.ensure((size_t)U * 4))) return rc;
  HIPCHK(hipMemcpyAsync(ss.d_lab_off, lab_off.data(), (size_t)U * 8, hipMemcpyHostToDevice, h->stream));
  // frames received = steps run, from the host's own count: the `avail` table of the last push may
  // live in a chunk buffer this path did not fill (pushes taken by the persistent launch)
  HIPCHK(hipMemcpyAsync(ss.d_have, ss.have.data(), (size_t)U * 4, hipMemcpyHostToDevice, h->stream));
  DecodeState stl = ss.st;
  stl.avail = ss.d_have;
  hipLaunchKernelGGL(k_backtrace, dim3(U), dim3(64), (size_t)64 * ss.B, h->stream, stl, ss.labels.as<int32_t>(),
                     ss.scores.as<float>(), ss.d_beam_scores);
  HIPCHK(hipGetLastError());
  if (F > 0) HIPCHK(hipMemcpyAsync(labels_out, ss.labels.p, (size_t)F * 4, hipMemcpyDeviceToHost, h->stream));
  if (scores_out) HIPCHK(hipMemcpyAsync(scores_out, ss.scores.p, (size_t)U * 4, hipMemcpyDeviceToHost, h->stream));
  h->last_U = U; h->last_B = ss.B;
  h->last_overflow.assign(U, 0);
  h->last_beam_scores.assign((size_t)U * ss.B, INFINITY);
  HIPCHK(hipMemcpyAsync(h->last_overflow.data(), ss.st.overflow, (size_t)U * 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(h->last_beam_scores.data(), ss.d_beam_scores, (size_t)U * ss.B * 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  int n_over = 0;
  for (int u = 0; u < U; ++u) {
    if (overflow_out) overflow_out[u] = h->last_overflow[u];
    n_over += h->last_overflow[u] != 0;
  }
  if (n_over)
    return fail(UIS_ERR_CLUSTER_CAP, std::to_string(n_over) + " utterance(s) needed more than max_clusters=" +
                                         std::to_string(ss.Kmax) + " clusters per hypothesis");
  return UIS_OK;
}

UIS_EXPORT int32_t uis_stream_end(uis_handle* h) {
  if (!h) return fail(UIS_ERR_INVALID_ARG, "null handle");
  if (!h->stream_state.active) return UIS_OK;
  HIPCHK(hipSetDevice(h->device));
  const int rc_quit = pm_quit(h);
  HIPCHK(hipStreamSynchronize(h->stream));
  stream_free(h);
  return rc_quit;
}


// ------------------------------------------------------------------ evaluation
//
// Sequence-match accuracy on the device (uis_eval.hip): the step after predict() in the
// reference's demo (demo.py:61-66, uisrnn/evals.py:40-73).

namespace {

int eval_run(uis_handle* h, const int32_t* d_a, const int32_t* d_b, const int64_t* offsets, int32_t n_utt,
             int64_t* matched_out) {
  if (n_utt == 0) return UIS_OK;
  if (offsets[0] != 0) return fail(UIS_ERR_INVALID_ARG, "offsets[0] must be 0");
  for (int u = 0; u < n_utt; ++u)
    if (offsets[u + 1] < offsets[u]) return fail(UIS_ERR_INVALID_ARG, "offsets must be non-decreasing");
  int rc;
  if ((rc = h->ev_off.ensure((size_t)(n_utt + 1) * 8))) return rc;
  if ((rc = h->ev_out.ensure((size_t)n_utt * 12))) return rc;
  long long* d_matched = h->ev_out.as<long long>();
  int32_t* d_status = reinterpret_cast<int32_t*>(d_matched + n_utt);
  HIPCHK(hipMemcpyAsync(h->ev_off.p, offsets, (size_t)(n_utt + 1) * 8, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_eval, dim3(n_utt), dim3(256), 0, h->stream, d_a, d_b, h->ev_off.as<int64_t>(), n_utt, d_matched,
                     d_status);
  HIPCHK(hipGetLastError());
  std::vector<int32_t> status(n_utt);
  static_assert(sizeof(long long) == sizeof(int64_t), "matched counts");
  HIPCHK(hipMemcpyAsync(matched_out, d_matched, (size_t)n_utt * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(status.data(), d_status, (size_t)n_utt * 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  for (int u = 0; u < n_utt; ++u) {
    if (status[u] == 1)
      return fail(UIS_ERR_UNSUPPORTED, "utterance " + std::to_string(u) + ": labels must lie in [0, 65536)");
    if (status[u] == 2)
      return fail(UIS_ERR_UNSUPPORTED, "utterance " + std::to_string(u) + ": more than 64 distinct labels in a sequence");
  }
  return UIS_OK;
}

}  // namespace

UIS_EXPORT int32_t uis_eval_accuracy_device(uis_handle* h, const int32_t* d_labels_a, const int32_t* d_labels_b,
                                            const int64_t* offsets, int32_t n_utt, int64_t* matched_out) {
  if (!h || !offsets || n_utt < 0 || (n_utt > 0 && !matched_out)) return fail(UIS_ERR_INVALID_ARG, "null argument or negative n_utt");
  if (n_utt > 0 && offsets[n_utt] > 0 && (!d_labels_a || !d_labels_b)) return fail(UIS_ERR_INVALID_ARG, "label pointer is null");
  HIPCHK(hipSetDevice(h->device));
  return eval_run(h, d_labels_a, d_labels_b, offsets, n_utt, matched_out);
}

UIS_EXPORT int32_t uis_eval_accuracy(uis_handle* h, const int32_t* labels_a, const int32_t* labels_b,
                                     const int64_t* offsets, int32_t n_utt, int64_t* matched_out) {
  if (!h || !offsets || n_utt < 0 || (n_utt > 0 && !matched_out)) return fail(UIS_ERR_INVALID_ARG, "null argument or negative n_utt");
  const int64_t F = n_utt ? offsets[n_utt] : 0;
  if (F > 0 && (!labels_a || !labels_b)) return fail(UIS_ERR_INVALID_ARG, "label pointer is null");
  HIPCHK(hipSetDevice(h->device));
  int rc;
  if ((rc = h->ev_a.ensure((size_t)std::max<int64_t>(F, 1) * 4))) return rc;
  if ((rc = h->ev_b.ensure((size_t)std::max<int64_t>(F, 1) * 4))) return rc;
  if (F > 0) {
    HIPCHK(hipMemcpyAsync(h->ev_a.p, labels_a, (size_t)F * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->ev_b.p, labels_b, (size_t)F * 4, hipMemcpyHostToDevice, h->stream));
  }
  return eval_run(h, h->ev_a.as<int32_t>(), h->ev_b.as<int32_t>(), offsets, n_utt, matched_out);
}

UIS_EXPORT int32_t uis_eval_last_decode(uis_handle* h, const int32_t* truth, int32_t n_utt, int64_t* matched_out) {
  if (!h || n_utt < 0 || (n_utt > 0 && !matched_out)) return fail(UIS_ERR_INVALID_ARG, "null argument or negative n_utt");
  if (h->io_offsets.empty() || (int)h->io_offsets.size() != n_utt + 1)
    return fail(UIS_ERR_INVALID_ARG, "no completed uis_decode with this many utterances on this handle");
  const int64_t F = h->io_offsets[n_utt];
  if (F > 0 && !truth) return fail(UIS_ERR_INVALID_ARG, "truth is null");
  HIPCHK(hipSetDevice(h->device));
  int rc;
  if ((rc = h->ev_b.ensure((size_t)std::max<int64_t>(F, 1) * 4))) return rc;
  if (F > 0) HIPCHK(hipMemcpyAsync(h->ev_b.p, truth, (size_t)F * 4, hipMemcpyHostToDevice, h->stream));
  // the predicted labels of the last uis_decode never left HBM for this
  return eval_run(h, h->io_labels.as<int32_t>(), h->ev_b.as<int32_t>(), h->io_offsets.data(), n_utt, matched_out);
}

// ------------------------------------------------------------------ pinned host memory
UIS_EXPORT int32_t uis_host_alloc(size_t bytes, void** out) {
  if (!out) return fail(UIS_ERR_INVALID_ARG, "null out");
  *out = nullptr;
  hipError_t e = hipHostMalloc(out, std::max<size_t>(bytes, 1), hipHostMallocDefault);
  if (e != hipSuccess) { *out = nullptr; return fail(UIS_ERR_OOM, std::string("hipHostMalloc: ") + hipGetErrorString(e)); }
  return UIS_OK;
}

UIS_EXPORT void uis_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}
