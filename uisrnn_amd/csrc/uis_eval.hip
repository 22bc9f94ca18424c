// uis_eval.hip -- sequence-match accuracy on the device.
//
// The step after predict() in the reference's demo (demo.py:61-66) is
// evals.compute_sequence_match_accuracy (uisrnn/evals.py:40-73): the confusion matrix of two
// label sequences and the best one-to-one matching of their label sets
// (scipy.optimize.linear_sum_assignment on the negated counts, evals.py:70); accuracy =
// matched positions / length.  Here: one workgroup per utterance,
//   1. which labels occur (LDS bitmaps) -> dense indices in sorted order, as evals.py:58-61;
//   2. confusion counts with LDS atomics (evals.py:63-69);
//   3. the assignment, exactly, by the Hungarian algorithm with potentials run by ONE wave:
//      lane j owns column j (its potential, its matched row, its slack), a row's potential
//      lives in LDS, every "minimum over the unvisited columns" is a wave reduction.
// Output: the number of matched positions per utterance (an integer: the optimum VALUE is
// unique even where the matching is not); the caller divides by the length in float64 like
// evals.py:72.  Integer work: bit-exact by construction.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define UIS_EVAL_MAX_LABEL 65536   // label values must lie in [0, 65536)
#define UIS_EVAL_MAX_IDS 64        // distinct labels per sequence (one wave lane per column)

// status per utterance: 0 ok, 1 label out of range, 2 too many distinct labels
__global__ __launch_bounds__(256) void k_eval(const int32_t* __restrict__ seq_a, const int32_t* __restrict__ seq_b,
                                              const int64_t* __restrict__ off, int n_utt,
                                              long long* __restrict__ matched, int32_t* __restrict__ status) {
  __shared__ unsigned long long bits_a[UIS_EVAL_MAX_LABEL / 64], bits_b[UIS_EVAL_MAX_LABEL / 64];
  __shared__ unsigned short pre_a[UIS_EVAL_MAX_LABEL / 64], pre_b[UIS_EVAL_MAX_LABEL / 64];
  __shared__ int cnt[UIS_EVAL_MAX_IDS][UIS_EVAL_MAX_IDS + 1];
  __shared__ long long pot_u[UIS_EVAL_MAX_IDS + 1];
  __shared__ int s_misc[4];  // [0] #ids a  [1] #ids b  [2] error
  const int u = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  if (u >= n_utt) return;
  const long n = (long)(off[u + 1] - off[u]);
  const int32_t* a = seq_a + off[u];
  const int32_t* b = seq_b + off[u];
  for (int i = tid; i < UIS_EVAL_MAX_LABEL / 64; i += 256) { bits_a[i] = 0ull; bits_b[i] = 0ull; }
  for (int i = tid; i < UIS_EVAL_MAX_IDS * (UIS_EVAL_MAX_IDS + 1); i += 256) (&cnt[0][0])[i] = 0;
  if (tid < 4) s_misc[tid] = 0;
  __syncthreads();
  if (n == 0) {  // the reference raises ValueError for empty sequences: the host checks that
    if (tid == 0) { matched[u] = 0; status[u] = 0; }
    return;
  }
  // ---- 1. label sets
  for (long i = tid; i < n; i += 256) {
    const int va = a[i], vb = b[i];
    if ((unsigned)va >= UIS_EVAL_MAX_LABEL || (unsigned)vb >= UIS_EVAL_MAX_LABEL) { s_misc[2] = 1; continue; }
    atomicOr(&bits_a[va >> 6], 1ull << (va & 63));
    atomicOr(&bits_b[vb >> 6], 1ull << (vb & 63));
  }
  __syncthreads();
  if (s_misc[2]) {
    if (tid == 0) { matched[u] = 0; status[u] = 1; }
    return;
  }
  // exclusive prefix of the popcounts (1024 words per bitmap): wave 0 scans a, wave 1 scans b
  if (tid < 128) {
    unsigned long long* bits = tid < 64 ? bits_a : bits_b;
    unsigned short* pre = tid < 64 ? pre_a : pre_b;
    constexpr int PER = UIS_EVAL_MAX_LABEL / 64 / 64;  // words per lane
    int mine = 0;
    for (int k = 0; k < PER; ++k) mine += __popcll(bits[lane * PER + k]);
    int incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    int run = incl - mine;
    for (int k = 0; k < PER; ++k) {
      pre[lane * PER + k] = (unsigned short)(run > 0xffff ? 0xffff : run);
      run += __popcll(bits[lane * PER + k]);
    }
    if (lane == 63) s_misc[tid < 64 ? 0 : 1] = incl;
  }
  __syncthreads();
  const int na = s_misc[0], nb = s_misc[1];
  if (na > UIS_EVAL_MAX_IDS || nb > UIS_EVAL_MAX_IDS) {
    if (tid == 0) { matched[u] = 0; status[u] = 2; }
    return;
  }
  // ---- 2. confusion counts: dense index = number of smaller labels present
  for (long i = tid; i < n; i += 256) {
    const int va = a[i], vb = b[i];
    const int ia = pre_a[va >> 6] + __popcll(bits_a[va >> 6] & ((1ull << (va & 63)) - 1ull));
    const int ib = pre_b[vb >> 6] + __popcll(bits_b[vb >> 6] & ((1ull << (vb & 63)) - 1ull));
    atomicAdd(&cnt[ia][ib], 1);
  }
  __syncthreads();
  if (tid >= 64) return;

  // ---- 3. maximum-weight assignment on the square matrix padded with zeros; one wave.
  // Minimising cost[i][j] = -cnt[i][j].  Rows 1..sz are added one by one; column 0 is the
  // virtual column that holds the row being inserted (kept in wave-uniform scalars).
  const int sz = na > nb ? na : nb;
  const int j = lane;                 // this lane's column is j + 1 in the textbook numbering
  const bool col = j < sz;
  long long v = 0;                    // column potential
  int p = 0;                          // row matched to this column (1-based, 0 = none)
  int way = 0;                        // previous column on the alternating path (0 = virtual)
  const long long INF = 1ll << 60;
  if (lane <= sz) pot_u[lane] = 0;
  for (int i = 1; i <= sz; ++i) {
    int p0 = i;                       // row held by the virtual column
    int j0 = 0;                       // current column (0 = virtual), wave-uniform
    long long minv = INF;
    bool used = false;
    while (true) {
      if (j0 > 0 && j == j0 - 1) used = true;
      const int i0 = j0 == 0 ? p0 : __shfl(p, j0 - 1, 64);
      const long long ui0 = pot_u[i0];
      if (col && !used) {
        const long long cur = -(long long)cnt[i0 - 1][j] - ui0 - v;
        if (cur < minv) { minv = cur; way = j0; }
      }
      // delta = min over unvisited columns of minv, j1 = the lowest such column
      long long best = (col && !used) ? minv : INF;
      int bj = j;
      for (int o = 32; o >= 1; o >>= 1) {
        const long long ob = __shfl_xor(best, o, 64);
        const int oj = __shfl_xor(bj, o, 64);
        if (ob < best || (ob == best && oj < bj)) { best = ob; bj = oj; }
      }
      const long long delta = best;
      // potentials: visited columns (and their rows) move by delta, the others' slack shrinks
      if (col && used) { pot_u[p] += delta; v -= delta; }
      else if (col) minv -= delta;
      if (lane == 0) pot_u[p0] += delta;  // the virtual column is always visited
      j0 = bj + 1;
      const int pj = __shfl(p, bj, 64);
      if (pj == 0) break;             // reached a free column: augment
    }
    // augment along `way` back to the virtual column
    while (j0 != 0) {
      const int j1 = __shfl(way, j0 - 1, 64);
      const int pnew = j1 == 0 ? p0 : __shfl(p, j1 - 1, 64);
      if (j == j0 - 1) p = pnew;
      j0 = j1;
    }
  }
  long long mine = (col && p > 0) ? (long long)cnt[p - 1][j] : 0;
  for (int o = 32; o >= 1; o >>= 1) mine += __shfl_xor(mine, o, 64);
  if (lane == 0) { matched[u] = mine; status[u] = 0; }
}
