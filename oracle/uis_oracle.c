/*
 * uis_oracle.c -- CPU restatement of the UIS-RNN beam-search decode.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker the HIP path is compared
 * against; nothing in the product path (uisrnn_amd/) may import, link or call
 * it.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * use it.
 *
 * It follows the reference line by line in behaviour (not in code):
 *   advance()        -- UISRNN._update_beam_state      uisrnn/uisrnn.py:388-453
 *   enumerate()      -- UISRNN._calculate_score        uisrnn/uisrnn.py:455-477
 *   decode_one()     -- UISRNN.predict_single          uisrnn/uisrnn.py:479-562
 *   rnn_step()       -- CoreRNN.forward                uisrnn/uisrnn.py:45-52
 *                       (torch.nn.GRU equations, gate order r|z|n, + 2-layer head)
 *   weighted MSE     -- loss_func.weighted_mse_loss    uisrnn/loss_func.py:19-41
 * with the float32 order of operations fixed by include/uis_numerics.h.
 *
 * Parity pinning: the reference ships no golden vectors for this path
 * (SURVEY.md 8c).  tests/golden/make_golden.py imports the reference itself
 * (/root/reference) and records its predict() outputs; tests/test_oracle_golden.py
 * checks this restatement against those fixtures.
 *
 * Deliberately simple: dense per-hypothesis bookkeeping, winners are replayed
 * from their parent exactly like uisrnn.py:551-559 does.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "uis_numerics.h"
#include "uisrnn_hip.h"

#define ORACLE_EXPORT __attribute__((visibility("default")))

static int round_up(int v, int m) { return (v + m - 1) / m * m; }

/* ------------------------------------------------------------------ model */

typedef struct {
  int D, H, depth, Dp, Hp;
  /* transposed, zero padded, K walked in canonical order:
     wt[kk * n_out + j] = W[j][korder(kk)] */
  float** wih_t; /* [depth]: (Kp_l x 3H) */
  float** whh_t; /* [depth]: (Hp x 3H)  */
  float** bih;
  float** bhh;
  float* w1_t;  /* (Hp x H) */
  float* b1;
  float* w2_t;  /* (Hp x D) */
  float* b2;
  float* h_init; /* depth*H */
  float* wgt;    /* 1/(2 sigma2) */
  double lp_stay, lp_sw, l_alpha, alpha;
  float* m0;     /* D */
  float* h1;     /* depth*H */
} omodel;

/* kk-th contraction index in canonical order, or -1 when it is padding */
static int canon_k(int kk, int K) {
  int k = (kk & ~15) + uis_korder(kk & 15);
  return k < K ? k : -1;
}

static float* transpose_canon(const float* W, int n_out, int K) {
  int Kp = round_up(K, UIS_KBLOCK);
  float* t = (float*)calloc((size_t)Kp * n_out, sizeof(float));
  for (int kk = 0; kk < Kp; ++kk) {
    int k = canon_k(kk, K);
    if (k < 0) continue;
    for (int j = 0; j < n_out; ++j) t[(size_t)kk * n_out + j] = W[(size_t)j * K + k];
  }
  return t;
}

/* out[j] = bias[j] + sum_k W[j][k] v[k] in the canonical order of uis_numerics.h:
   UIS_KSPLIT segment chains (one fmaf per term, padding terms are fmaf(0, 0, acc)
   like the zero-padded device buffers) combined left to right. */
static void dense_chain(const float* wt, const float* bias, const float* v, int K,
                        int n_out, float* out) {
  int Kp = round_up(K, UIS_KBLOCK);
  int nKb = Kp / UIS_KBLOCK;
  int q = uis_kseg_blocks(nKb);
  float* seg = (float*)malloc((size_t)n_out * sizeof(float));
  for (int s = 0; s < UIS_KSPLIT; ++s) {
    for (int j = 0; j < n_out; ++j) seg[j] = s == 0 ? bias[j] : 0.0f;
    int kb0 = s * q, kb1 = (s + 1) * q < nKb ? (s + 1) * q : nKb;
    for (int kk = kb0 * UIS_KBLOCK; kk < kb1 * UIS_KBLOCK; ++kk) {
      int k = canon_k(kk, K);
      float vk = k < 0 ? 0.0f : v[k];
      const float* row = wt + (size_t)kk * n_out;
      for (int j = 0; j < n_out; ++j) seg[j] = fmaf(row[j], vk, seg[j]);
    }
    if (s == 0) for (int j = 0; j < n_out; ++j) out[j] = seg[j];
    else for (int j = 0; j < n_out; ++j) out[j] = out[j] + seg[j];
  }
  free(seg);
}

/* CoreRNN.forward for seq_len = 1, batch = 1 (uisrnn/uisrnn.py:45-52). */
static void rnn_step(const omodel* m, const float* x, const float* h_in,
                     float* mean_out, float* h_out, float* scratch) {
  int H = m->H, D = m->D;
  float* gi = scratch;          /* 3H */
  float* gh = scratch + 3 * H;  /* 3H */
  float* a1 = scratch + 6 * H;  /* H  */
  const float* inp = x;
  int in_dim = D;
  for (int l = 0; l < m->depth; ++l) {
    const float* h = h_in + (size_t)l * H;
    float* ho = h_out + (size_t)l * H;
    dense_chain(m->wih_t[l], m->bih[l], inp, in_dim, 3 * H, gi);
    dense_chain(m->whh_t[l], m->bhh[l], h, H, 3 * H, gh);
    for (int j = 0; j < H; ++j)
      ho[j] = uis_gru_unit(gi[j], gi[H + j], gi[2 * H + j], gh[j], gh[H + j], gh[2 * H + j], h[j]);
    inp = ho;
    in_dim = H;
  }
  dense_chain(m->w1_t, m->b1, inp, H, H, a1);
  for (int j = 0; j < H; ++j) a1[j] = a1[j] > 0.0f ? a1[j] : 0.0f; /* F.relu */
  dense_chain(m->w2_t, m->b2, a1, H, D, mean_out);
}

static float weighted_mse(const omodel* m, const float* mean, const float* x, float* term) {
  for (int d = 0; d < m->D; ++d) term[d] = uis_mse_term(mean[d], x[d], m->wgt[d]);
  float sum = uis_tree_sum(term, m->D);
  float d0 = mean[0] - x[0];
  return uis_mse_finish(sum, d0 * d0, m->D);
}

static omodel* model_build(const uis_model_desc* d) {
  omodel* m = (omodel*)calloc(1, sizeof(omodel));
  int D = d->observation_dim, H = d->rnn_hidden_size, L = d->rnn_depth;
  m->D = D; m->H = H; m->depth = L;
  m->Dp = round_up(D, UIS_KBLOCK); m->Hp = round_up(H, UIS_KBLOCK);
  m->wih_t = (float**)calloc(L, sizeof(float*));
  m->whh_t = (float**)calloc(L, sizeof(float*));
  m->bih = (float**)calloc(L, sizeof(float*));
  m->bhh = (float**)calloc(L, sizeof(float*));
  for (int l = 0; l < L; ++l) {
    int in_dim = l == 0 ? D : H;
    m->wih_t[l] = transpose_canon(d->gru_weight_ih[l], 3 * H, in_dim);
    m->whh_t[l] = transpose_canon(d->gru_weight_hh[l], 3 * H, H);
    m->bih[l] = (float*)malloc(3 * H * sizeof(float));
    m->bhh[l] = (float*)malloc(3 * H * sizeof(float));
    memcpy(m->bih[l], d->gru_bias_ih[l], 3 * H * sizeof(float));
    memcpy(m->bhh[l], d->gru_bias_hh[l], 3 * H * sizeof(float));
  }
  m->w1_t = transpose_canon(d->linear_mean1_weight, H, H);
  m->w2_t = transpose_canon(d->linear_mean2_weight, D, H);
  m->b1 = (float*)malloc(H * sizeof(float)); memcpy(m->b1, d->linear_mean1_bias, H * sizeof(float));
  m->b2 = (float*)malloc(D * sizeof(float)); memcpy(m->b2, d->linear_mean2_bias, D * sizeof(float));
  m->h_init = (float*)malloc((size_t)L * H * sizeof(float));
  memcpy(m->h_init, d->rnn_init_hidden, (size_t)L * H * sizeof(float));
  m->wgt = (float*)malloc(D * sizeof(float));
  for (int i = 0; i < D; ++i) m->wgt[i] = 1.0f / (2.0f * d->sigma2[i]); /* uisrnn.py:414 */
  m->alpha = d->crp_alpha;
  m->lp_stay = log(1.0 - d->transition_bias);
  m->lp_sw = log(d->transition_bias);
  m->l_alpha = log(d->crp_alpha);
  /* (m0, h1) = CoreRNN(0, rnn_init_hidden): uisrnn.py:435-439 */
  m->m0 = (float*)malloc(D * sizeof(float));
  m->h1 = (float*)malloc((size_t)L * H * sizeof(float));
  float* zero = (float*)calloc(D, sizeof(float));
  float* scratch = (float*)malloc(7 * H * sizeof(float));
  rnn_step(m, zero, m->h_init, m->m0, m->h1, scratch);
  free(zero); free(scratch);
  return m;
}

static void model_free(omodel* m) {
  for (int l = 0; l < m->depth; ++l) { free(m->wih_t[l]); free(m->whh_t[l]); free(m->bih[l]); free(m->bhh[l]); }
  free(m->wih_t); free(m->whh_t); free(m->bih); free(m->bhh);
  free(m->w1_t); free(m->w2_t); free(m->b1); free(m->b2); free(m->h_init); free(m->wgt); free(m->m0); free(m->h1);
  free(m);
}

/* ------------------------------------------------------------- hypotheses */

/* BeamState (uisrnn/uisrnn.py:55-77); cluster states are shared between
   hypotheses like the reference's shallow list copies, via refcounts. */
typedef struct cstate {
  int refs;
  int cnt;      /* frames assigned so far == (np.array(trace) == cluster).sum() */
  float* mean;  /* D */
  float* hid;   /* depth*H */
} cstate;

typedef struct hyp {
  int K, cap;
  cstate** cl;
  int* blk;       /* block_counts */
  int last;       /* trace[-1] */
  long sumblk;    /* sum(block_counts) */
  float score;    /* neg_likelihood (float32 accumulate, uisrnn.py:452) */
} hyp;

static cstate* cs_new(const omodel* m) {
  cstate* c = (cstate*)malloc(sizeof(cstate));
  c->refs = 1; c->cnt = 0;
  c->mean = (float*)malloc(m->D * sizeof(float));
  c->hid = (float*)malloc((size_t)m->depth * m->H * sizeof(float));
  return c;
}
static void cs_unref(cstate* c) {
  if (--c->refs == 0) { free(c->mean); free(c->hid); free(c); }
}
static void hyp_init(hyp* h) { memset(h, 0, sizeof(*h)); h->last = -1; }
static void hyp_free(hyp* h) {
  for (int i = 0; i < h->K; ++i) cs_unref(h->cl[i]);
  free(h->cl); free(h->blk);
  memset(h, 0, sizeof(*h));
}
static void hyp_copy(hyp* dst, const hyp* src) { /* BeamState(source) */
  dst->K = src->K; dst->cap = src->K + 4;
  dst->cl = (cstate**)malloc(dst->cap * sizeof(cstate*));
  dst->blk = (int*)malloc(dst->cap * sizeof(int));
  for (int i = 0; i < src->K; ++i) { dst->cl[i] = src->cl[i]; dst->cl[i]->refs++; dst->blk[i] = src->blk[i]; }
  dst->last = src->last; dst->sumblk = src->sumblk; dst->score = src->score;
}
static void hyp_reserve(hyp* h, int need) {
  if (need <= h->cap) return;
  h->cap = need + 4;
  h->cl = (cstate**)realloc(h->cl, h->cap * sizeof(cstate*));
  h->blk = (int*)realloc(h->blk, h->cap * sizeof(int));
}

typedef struct {
  const omodel* m;
  float* term;     /* D */
  float* scratch;  /* 7H */
  float* mean_tmp; /* D */
  float* hid_tmp;  /* depth*H */
  long rnn_calls;
} octx;

/*
 * One sub-step of _update_beam_state (uisrnn/uisrnn.py:405-452) applied in
 * place to h.  Returns 0 for an invalid assignment (cluster > K).
 * with_state == 0 computes only the score (enough for the last sub-step of a
 * candidate; the winners are replayed with_state == 1).
 */
static int advance(octx* cx, hyp* h, const float* x, int c, int with_state) {
  const omodel* m = cx->m;
  if (c > h->K) { h->score = INFINITY; return 0; }
  float loss;
  if (c < h->K) { /* existing cluster: uisrnn.py:409-433 */
    cstate* cs = h->cl[c];
    float mse = weighted_mse(m, cs->mean, x, cx->term);
    double prior;
    if (c == h->last) prior = m->lp_stay;
    else prior = m->lp_sw + log((double)h->blk[c]) - log((double)h->sumblk + m->alpha);
    loss = uis_step_loss(mse, prior);
    if (with_state) {
      rnn_step(m, x, cs->hid, cx->mean_tmp, cx->hid_tmp, cx->scratch);
      cx->rnn_calls++;
      cstate* ns = cs_new(m);
      int n = cs->cnt;
      for (int d = 0; d < m->D; ++d) ns->mean[d] = uis_mean_update(cs->mean[d], cx->mean_tmp[d], n);
      memcpy(ns->hid, cx->hid_tmp, (size_t)m->depth * m->H * sizeof(float));
      ns->cnt = n + 1;
      cs_unref(cs);
      h->cl[c] = ns;
    }
    if (c != h->last) { h->blk[c] += 1; h->sumblk += 1; }
  } else { /* new cluster: uisrnn.py:434-451 */
    float mse = weighted_mse(m, m->m0, x, cx->term);
    double prior = m->lp_sw + m->l_alpha - log((double)h->sumblk + m->alpha);
    loss = uis_step_loss(mse, prior);
    hyp_reserve(h, h->K + 1);
    cstate* ns = cs_new(m);
    if (with_state) {
      rnn_step(m, x, m->h1, ns->mean, ns->hid, cx->scratch);
      cx->rnn_calls++;
    } else {
      memset(ns->mean, 0, m->D * sizeof(float));
      memset(ns->hid, 0, (size_t)m->depth * m->H * sizeof(float));
    }
    ns->cnt = 1;
    h->cl[h->K] = ns; h->blk[h->K] = 1; h->sumblk += 1; h->K += 1;
  }
  h->score = h->score + loss; /* float32 accumulate */
  h->last = c;
  return 1;
}

#define ORACLE_MAX_L 16 /* look_ahead the checker takes (the device path has no such bound; tests go up to 9) */
typedef struct { float score; int beam; int path[ORACLE_MAX_L]; long order; } cand;

typedef struct { cand* v; long n, cap; } candvec;
static void cv_push(candvec* cv, const cand* c) {
  if (cv->n == cv->cap) { cv->cap = cv->cap ? cv->cap * 2 : 256; cv->v = (cand*)realloc(cv->v, cv->cap * sizeof(cand)); }
  cv->v[cv->n++] = *c;
}

/* _calculate_score (uisrnn.py:455-477): every tuple (c_1..c_Lw), c_j <= K + j - 1,
   row-major; invalid tuples are the +inf entries and are simply not emitted. */
static void enumerate(octx* cx, const hyp* base, const float* xs, int Lw, int sub,
                      int beam, int* path, candvec* out) {
  int D = cx->m->D;
  for (int c = 0; c <= base->K; ++c) {
    hyp h;
    hyp_copy(&h, base);
    int last_sub = (sub == Lw - 1);
    advance(cx, &h, xs + (size_t)sub * D, c, !last_sub);
    path[sub] = c;
    if (last_sub) {
      cand cd; cd.score = h.score; cd.beam = beam; cd.order = out->n;
      for (int i = 0; i < ORACLE_MAX_L; ++i) cd.path[i] = i < Lw ? path[i] : -1;
      cv_push(out, &cd);
    } else {
      enumerate(cx, &h, xs, Lw, sub + 1, beam, path, out);
    }
    hyp_free(&h);
  }
}

static int cand_cmp(const void* a, const void* b) {
  const cand* x = (const cand*)a; const cand* y = (const cand*)b;
  uint32_t kx = uis_score_key(x->score), ky = uis_score_key(y->score);
  if (kx != ky) return kx < ky ? -1 : 1;
  return x->order < y->order ? -1 : (x->order > y->order ? 1 : 0);
}

typedef struct {
  float best_score;
  float min_rel_margin; /* smallest relative gap between adjacent ranked candidates
                           up to and including the prune boundary, over all steps */
  int max_clusters;
  long rnn_calls;
  long candidates;
} oinfo;

/* predict_single (uisrnn/uisrnn.py:479-562) for one utterance. */
/* dbg (may be NULL): every candidate's score, dense --
   dbg[((win * B + beam) * Cmax + c_1) * Cmax + c_2 ...] with Lw factors of Cmax per window
   (stride Cmax^L per hypothesis), the caller pre-fills it with +inf: the arrays _calculate_score
   returns (uisrnn.py:455-477) inside predict_single's padded score_set (uisrnn.py:534-545). */
static void decode_one(const omodel* m, const float* seq, long N, int B, int L, int tau,
                       int32_t* labels, float* beam_scores, oinfo* info, float* dbg, int Cmax) {
  octx cx; cx.m = m; cx.rnn_calls = 0;
  cx.term = (float*)malloc(m->D * sizeof(float));
  cx.scratch = (float*)malloc(7 * m->H * sizeof(float));
  cx.mean_tmp = (float*)malloc(m->D * sizeof(float));
  cx.hid_tmp = (float*)malloc((size_t)m->depth * m->H * sizeof(float));
  info->best_score = 0.0f; info->min_rel_margin = INFINITY; info->max_clusters = 0; info->candidates = 0;
  long T = (long)tau * N;
  long n_win = (T + L - 1) / L;
  hyp* beam = (hyp*)calloc(B, sizeof(hyp));
  hyp* next = (hyp*)calloc(B, sizeof(hyp));
  int nb = 1;
  hyp_init(&beam[0]);
  /* back-pointers instead of per-hypothesis trace copies */
  int16_t* bp_parent = (int16_t*)malloc((size_t)(n_win ? n_win : 1) * B * sizeof(int16_t));
  int32_t* bp_path = (int32_t*)malloc((size_t)(n_win ? n_win : 1) * B * L * sizeof(int32_t));
  float* xs = (float*)malloc((size_t)L * m->D * sizeof(float));
  candvec cv = {0, 0, 0};
  long win = 0;
  for (long t = 0; t < T; t += L, ++win) { /* np.arange(0, tau*N, look_ahead) */
    int Lw = (int)((T - t) < L ? (T - t) : L);
    for (int j = 0; j < Lw; ++j) /* np.tile(seq, (tau, 1)) */
      memcpy(xs + (size_t)j * m->D, seq + (size_t)((t + j) % N) * m->D, m->D * sizeof(float));
    cv.n = 0;
    int path[ORACLE_MAX_L];
    for (int b = 0; b < nb; ++b) enumerate(&cx, &beam[b], xs, Lw, 0, b, path, &cv);
    info->candidates += cv.n;
    if (dbg) {
      long stride = 1;
      for (int j = 0; j < L; ++j) stride *= Cmax;
      for (long i = 0; i < cv.n; ++i) {
        const cand* cd = &cv.v[i];
        long flat = 0;
        int ok = 1;
        for (int j = 0; j < Lw; ++j) { if (cd->path[j] >= Cmax) ok = 0; flat = flat * Cmax + cd->path[j]; }
        for (int j = Lw; j < L; ++j) flat *= Cmax; /* a ragged last window: the trailing indices stay 0 */
        if (ok) dbg[(win * B + cd->beam) * stride + flat] = cd->score;
      }
    }
    qsort(cv.v, cv.n, sizeof(cand), cand_cmp);
    long n_fin = 0;
    while (n_fin < cv.n && uis_isfinite(cv.v[n_fin].score)) ++n_fin; /* non-finite sort last */
    int keep = (int)(n_fin < B ? n_fin : B);
    for (int r = 0; r + 1 <= keep && r + 1 < n_fin; ++r) {
      float a = cv.v[r].score, b2 = cv.v[r + 1].score;
      float den = fabsf(b2) > 1e-30f ? fabsf(b2) : 1e-30f;
      float rel = (b2 - a) / den;
      if (rel < info->min_rel_margin) info->min_rel_margin = rel;
    }
    for (int r = 0; r < keep; ++r) { /* replay the winners: uisrnn.py:551-559 */
      const cand* cd = &cv.v[r];
      hyp_copy(&next[r], &beam[cd->beam]);
      for (int j = 0; j < Lw; ++j) advance(&cx, &next[r], xs + (size_t)j * m->D, cd->path[j], 1);
      bp_parent[win * B + r] = (int16_t)cd->beam;
      for (int j = 0; j < L; ++j) bp_path[(win * B + r) * L + j] = j < Lw ? cd->path[j] : -1;
      if (next[r].K > info->max_clusters) info->max_clusters = next[r].K;
    }
    for (int b = 0; b < nb; ++b) hyp_free(&beam[b]);
    hyp* tmp = beam; beam = next; next = tmp;
    nb = keep;
    if (nb == 0) break; /* every candidate non-finite: the reference would raise IndexError */
  }
  /* trace[-N:] of beam_set[0] (uisrnn.py:561) */
  if (N > 0 && nb > 0) {
    int r = 0;
    long pos = T; /* one past the last written frame */
    for (long w = n_win - 1; w >= 0 && pos > T - N; --w) {
      long t0 = w * L;
      int Lw = (int)((T - t0) < L ? (T - t0) : L);
      for (int j = Lw - 1; j >= 0; --j) {
        long tt = t0 + j;
        if (tt >= T - N) labels[tt - (T - N)] = bp_path[(w * B + r) * L + j];
      }
      pos = t0;
      r = bp_parent[w * B + r];
    }
    info->best_score = beam[0].score;
  } else if (N > 0) {
    for (long i = 0; i < N; ++i) labels[i] = -1;
    info->best_score = INFINITY;
  }
  if (beam_scores)
    for (int b = 0; b < B; ++b) beam_scores[b] = b < nb ? beam[b].score : INFINITY;
  for (int b = 0; b < nb; ++b) hyp_free(&beam[b]);
  info->rnn_calls = cx.rnn_calls;
  free(beam); free(next); free(bp_parent); free(bp_path); free(xs); free(cv.v);
  free(cx.term); free(cx.scratch); free(cx.mean_tmp); free(cx.hid_tmp);
}

/* ------------------------------------------------------------------- API */

typedef struct {
  const omodel* m; const float* frames; const int64_t* offsets; int n_utt;
  int B, L, tau; int32_t* labels; float* scores; float* beam_scores; float* margins;
  int32_t* max_clusters; long rnn_calls, candidates; int next; pthread_mutex_t mu;
} job;

static void* worker(void* arg) {
  job* jb = (job*)arg;
  for (;;) {
    pthread_mutex_lock(&jb->mu);
    int u = jb->next++;
    pthread_mutex_unlock(&jb->mu);
    if (u >= jb->n_utt) break;
    long N = (long)(jb->offsets[u + 1] - jb->offsets[u]);
    oinfo info;
    decode_one(jb->m, jb->frames + (size_t)jb->offsets[u] * jb->m->D, N, jb->B, jb->L, jb->tau,
               jb->labels + jb->offsets[u], jb->beam_scores ? jb->beam_scores + (size_t)u * jb->B : NULL, &info, NULL, 0);
    if (jb->scores) jb->scores[u] = info.best_score;
    if (jb->margins) jb->margins[u] = info.min_rel_margin;
    if (jb->max_clusters) jb->max_clusters[u] = info.max_clusters;
    pthread_mutex_lock(&jb->mu);
    jb->rnn_calls += info.rnn_calls; jb->candidates += info.candidates;
    pthread_mutex_unlock(&jb->mu);
  }
  return NULL;
}

/*
 * Decode n_utt utterances on n_threads host threads (utterances are
 * independent, like uisrnn.parallel_predict, uisrnn/uisrnn.py:593-623).
 * Optional outputs (may be NULL): scores [n_utt], beam_scores [n_utt*B],
 * margins [n_utt], max_clusters [n_utt], counters[2] = {rnn calls, candidates}.
 */
ORACLE_EXPORT int32_t uis_oracle_decode(const uis_model_desc* desc, const float* frames,
                                        const int64_t* offsets, int32_t n_utt,
                                        const uis_decode_opts* opts, int32_t n_threads,
                                        int32_t* labels_out, float* scores_out,
                                        float* beam_scores_out, float* margins_out,
                                        int32_t* max_clusters_out, int64_t* counters_out) {
  if (!desc || !offsets || !opts || n_utt < 0) return UIS_ERR_INVALID_ARG;
  if (opts->beam_size < 1 || opts->look_ahead < 1 || opts->look_ahead > ORACLE_MAX_L || opts->test_iteration < 1)
    return UIS_ERR_INVALID_ARG;
  omodel* m = model_build(desc);
  job jb;
  memset(&jb, 0, sizeof(jb));
  jb.m = m; jb.frames = frames; jb.offsets = offsets; jb.n_utt = n_utt;
  jb.B = opts->beam_size; jb.L = opts->look_ahead; jb.tau = opts->test_iteration;
  jb.labels = labels_out; jb.scores = scores_out; jb.beam_scores = beam_scores_out;
  jb.margins = margins_out; jb.max_clusters = max_clusters_out;
  pthread_mutex_init(&jb.mu, NULL);
  if (n_threads < 1) n_threads = 1;
  if (n_threads > n_utt) n_threads = n_utt > 0 ? n_utt : 1;
  if (n_threads == 1) {
    worker(&jb);
  } else {
    pthread_t* th = (pthread_t*)malloc(n_threads * sizeof(pthread_t));
    for (int i = 0; i < n_threads; ++i) pthread_create(&th[i], NULL, worker, &jb);
    for (int i = 0; i < n_threads; ++i) pthread_join(th[i], NULL);
    free(th);
  }
  if (counters_out) { counters_out[0] = jb.rnn_calls; counters_out[1] = jb.candidates; }
  pthread_mutex_destroy(&jb.mu);
  model_free(m);
  return UIS_OK;
}

ORACLE_EXPORT int32_t uis_oracle_numerics_version(void) { return UIS_NUMERICS_VERSION; }

/*
 * One utterance, with every candidate score kept: scores_out is
 * [windows][beam_size][Cmax]^look_ahead floats (see decode_one), pre-filled with +inf here.
 */
ORACLE_EXPORT int32_t uis_oracle_candidate_scores(const uis_model_desc* desc, const float* frames, int64_t n_frames,
                                                  const uis_decode_opts* opts, int32_t Cmax, float* scores_out,
                                                  int32_t* labels_out) {
  if (!desc || !opts || !scores_out || Cmax < 1) return UIS_ERR_INVALID_ARG;
  omodel* m = model_build(desc);
  const int B = opts->beam_size, L = opts->look_ahead, tau = opts->test_iteration;
  long T = (long)tau * n_frames, n_win = (T + L - 1) / L, stride = 1;
  for (int j = 0; j < L; ++j) stride *= Cmax;
  for (long i = 0; i < n_win * B * stride; ++i) scores_out[i] = INFINITY;
  oinfo info;
  int32_t* lab = labels_out ? labels_out : (int32_t*)malloc((size_t)(n_frames ? n_frames : 1) * sizeof(int32_t));
  decode_one(m, frames, n_frames, B, L, tau, lab, NULL, &info, scores_out, Cmax);
  if (!labels_out) free(lab);
  model_free(m);
  return UIS_OK;
}

/* Unit-level entry points so tests can pin single functions against the reference. */

/* CoreRNN.forward (uisrnn.py:45-52): x [D], h_in [depth*H] -> mean [D], h_out [depth*H] */
ORACLE_EXPORT int32_t uis_oracle_rnn_step(const uis_model_desc* desc, const float* x, const float* h_in,
                                          float* mean_out, float* h_out) {
  omodel* m = model_build(desc);
  float* scratch = (float*)malloc(7 * m->H * sizeof(float));
  rnn_step(m, x, h_in, mean_out, h_out, scratch);
  free(scratch);
  model_free(m);
  return UIS_OK;
}

/* loss_func.weighted_mse_loss for one row with weight 1/(2 sigma2) (uisrnn.py:411-414) */
ORACLE_EXPORT float uis_oracle_weighted_mse(const uis_model_desc* desc, const float* mean, const float* x) {
  omodel* m = model_build(desc);
  float* term = (float*)malloc(m->D * sizeof(float));
  float v = weighted_mse(m, mean, x, term);
  free(term);
  model_free(m);
  return v;
}

/* The per-model constants (m0, h1) = CoreRNN(0, rnn_init_hidden) (uisrnn.py:435-439). */
ORACLE_EXPORT int32_t uis_oracle_constants(const uis_model_desc* desc, float* m0_out, float* h1_out) {
  omodel* m = model_build(desc);
  memcpy(m0_out, m->m0, m->D * sizeof(float));
  memcpy(h1_out, m->h1, (size_t)m->depth * m->H * sizeof(float));
  model_free(m);
  return UIS_OK;
}
