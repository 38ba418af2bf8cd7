/*
 * niqki_oracle.c -- CPU restatement of the NIQKI sketch/query hot path.
 * TEST INFRASTRUCTURE ONLY; see niqki_oracle.h for the rules and the parity
 * status (PINNED against the compiled reference, tests/golden/).
 *
 * Written from the behaviour of /root/reference/src/niqki_index.cpp (line
 * numbers cited per function), not copied from it: the data structures are
 * flat arrays (CSR index, byte tables) instead of the reference's
 * vector<vector>, string and switch statements.
 */
#include "niqki_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- scalar pieces ------------------------------------------------------ */

uint32_t nqo_min_score(double min_fract, uint32_t S) {
  /* src/niqki_index.cpp:21-22: F=1<<lF; min_score=min_fract*F (uint32 member) */
  double f = (double)((uint32_t)1 << S);
  return (uint32_t)(min_fract * f);
}

static inline uint64_t mix64(uint64_t x, uint64_t c) {
  x = ((x >> 32) ^ x) * c;
  x = ((x >> 32) ^ x) * c;
  return (x >> 32) ^ x;
}

uint64_t nqo_rev64(uint64_t x) { /* src/niqki_index.cpp:291-296 */
  return mix64(x, 0xD6E8FEB86659FD93ULL);
}

uint64_t nqo_unrev64(uint64_t x) { /* src/niqki_index.cpp:300-305 */
  return mix64(x, 0xCFEE444D8B59A89BULL);
}

int32_t nqo_fingerprint(uint64_t h, uint32_t W, uint32_t H) {
  /* src/niqki_index.cpp:277-287: low M bits of h, plus a saturating
   * (2^H-1 - leading_zeros) in the H bits above them. */
  uint32_t M = W - H;
  uint32_t mask_m = ((uint32_t)1 << M) - 1u;
  int32_t lz = h ? __builtin_clzll(h) : 64; /* bsr(0) is UB there; observed 0 */
  int32_t rem = (int32_t)(((uint32_t)1 << H) - 1u) - lz;
  if (rem < 0) rem = 0;
  return (int32_t)((uint32_t)(h & mask_m) + ((uint32_t)rem << M));
}

int32_t nqo_fingerprint_stale(uint64_t h, uint32_t W, uint32_t H, uint32_t H0) {
  /* get_fingerprint after select_best_H (src/niqki_index.cpp:126-138) replaced
   * the constructor's H0 by H: only H and M = W-H are updated there, mask_M and
   * maximal_remainder keep their constructor values (:24-25), so the low part
   * still has W-H0 bits and the saturation still counts from 2^H0-1, while the
   * HyperLogLog part is shifted by the NEW M.  The two parts are ADDED (:285)
   * and may overlap or leave [0, 2^W). */
  uint32_t M = W - H;
  uint32_t mask_m = ((uint32_t)1 << (W - H0)) - 1u;
  int32_t lz = h ? __builtin_clzll(h) : 64;
  int32_t rem = (int32_t)(((uint32_t)1 << H0) - 1u) - lz;
  if (rem < 0) rem = 0;
  return (int32_t)((uint32_t)(h & mask_m) + ((uint32_t)rem << M));
}

/* score_H: src/niqki_index.cpp:142-164, double arithmetic as written. */
static double score_H(double x, int try_h, uint32_t W) {
  double epsilon = 0.02;
  double try_m = (double)(uint32_t)(W - (uint32_t)try_h); /* uint32 - int: unsigned, wraps for try_h > W */
  double two_h = pow(2, try_h);
  double ua = (((double)1 - pow(1 - epsilon, 1 / x)) * pow(2, 64));
  double ia = log2(ua) + two_h - 64;
  double ja = ua * pow(2, try_m - 64 - ia + two_h);
  double ka;
  if (ua < pow(2, 64 - two_h + 1)) ka = ua * pow(2, two_h - 64 - try_m - 1);
  else ka = ia * pow(2, try_m) + ja;
  double ub = ((double)1 - pow(epsilon, 1 / x)) * pow(2, 64);
  double ib = log2(ub) + two_h - 64;
  double jb = ub * pow(2, try_m - 64 - ib + two_h);
  double kb;
  if (ub < pow(2, 64 - two_h + 1)) kb = ub * pow(2, two_h - 64 - try_m - 1);
  else kb = ib * pow(2, try_m) + jb;
  return kb - ka;
}

uint32_t nqo_select_best_H(double genome_size, uint32_t S, uint32_t W, uint32_t H) {
  /* src/niqki_index.cpp:126-138: the widest interval over try_h = 2..6 wins,
   * the current H stays when no interval is positive (NaN compares false). */
  double x = genome_size / (double)((uint64_t)1 << S);
  double best = 0;
  for (int try_h = 2; try_h < 7; try_h++) {
    double v = score_H(x, try_h, W);
    if (v > best) { best = v; H = (uint32_t)try_h; }
  }
  return H;
}

uint64_t nqo_hash_family(uint64_t x, uint32_t step) { /* :308-310 */
  return nqo_unrev64(x) + (uint64_t)step * nqo_rev64(x);
}

/* Forward / reverse-complement 2-bit codes of the rolling updates,
 * src/niqki_index.cpp:114-123 and :211-221: upper case only, anything else 0
 * in BOTH tables. */
static inline uint64_t code_fwd(uint8_t c) {
  return c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 0u;
}
static inline uint64_t code_rc(uint8_t c) {
  return c == 'A' ? 3u : c == 'C' ? 2u : c == 'G' ? 1u : 0u;
}

/* First K-1 bases, case-insensitive; any other byte zeroes the whole word:
 * src/niqki_index.cpp:255-273. */
static uint64_t pack_prefix(const uint8_t *s, uint32_t n) {
  uint64_t w = 0;
  for (uint32_t i = 0; i < n; ++i) {
    uint8_t c = s[i];
    uint64_t d;
    if (c == 'A' || c == 'a') d = 0;
    else if (c == 'C' || c == 'c') d = 1;
    else if (c == 'G' || c == 'g') d = 2;
    else if (c == 'T' || c == 't') d = 3;
    else return 0;
    w = (w << 2) | d;
  }
  return w;
}

/* Reverse complement over K digit positions: src/niqki_index.cpp:240-250. */
static uint64_t revcomp_k(uint64_t w, uint32_t K) {
  uint64_t r = 0;
  for (uint32_t i = 0; i < K; ++i) {
    r = (r << 2) | (3u - (w & 3u));
    w >>= 2;
  }
  return r;
}

uint64_t nqo_sketch_accumulate(const nqo_params *p, const uint8_t *seq,
                               uint64_t len, int32_t *sk) {
  const uint32_t K = p->K, S = p->S;
  if (len <= K) return 0;
  /* src/niqki_index.cpp:340-341 */
  uint64_t fw = pack_prefix(seq, K - 1);
  uint64_t rc = revcomp_k(fw, K);
  const uint64_t kmask = (K < 32) ? (((uint64_t)1 << (2 * K)) - 1) : ~(uint64_t)0;
  const uint32_t rc_shift = 2 * K - 2;
  const uint64_t n_kmers = len - K; /* loop bound i+K<len: last k-mer skipped (:342) */
  for (uint64_t i = 0; i < n_kmers; ++i) {
    uint8_t c = seq[i + K - 1];
    fw = ((fw << 2) + code_fwd(c)) & kmask; /* :225-229 */
    rc = (rc >> 2) + (code_rc(c) << rc_shift); /* :233-236 */
    uint64_t canon = fw < rc ? fw : rc;        /* :345 */
    uint64_t slot = nqo_unrev64(canon) >> (64 - S); /* :347 */
    int32_t fp = nqo_fingerprint_stale(nqo_rev64(canon), p->W, p->H, p->H0p1 ? p->H0p1 - 1 : p->H); /* :346,348 */
    int32_t cur = sk[slot];
    if (cur == -1 || cur > fp) sk[slot] = fp; /* :350-355: min over fp */
  }
  return n_kmers;
}

int64_t nqo_densify(const nqo_params *p, int32_t *sk) {
  /* src/niqki_index.cpp:313-331 */
  const uint32_t F = (uint32_t)1 << p->S;
  uint32_t empty = 0;
  for (uint32_t i = 0; i < F; ++i) empty += (sk[i] == -1);
  if (empty == 0) return 0;
  if (empty == F) return -1; /* nothing to copy from: the reference spins */
  uint32_t step = 0;
  uint32_t idle = 0; /* consecutive passes without a fill */
  int64_t passes = 0;
  for (;;) {
    ++passes;
    uint32_t filled = 0;
    for (uint32_t i = 0; i < F; ++i) {
      int32_t v = sk[i];
      if (v == -1) continue;
      uint64_t t = nqo_hash_family((uint64_t)(int64_t)v, step) % F;
      if (sk[t] == -1) {
        sk[t] = v;
        ++filled;
        if (--empty == 0) return passes;
      }
    }
    ++step;
    /* Targets are periodic in step with period dividing F, and a pass
     * without a fill leaves the sketch unchanged: F fruitless passes in a
     * row prove that no later pass can fill anything. */
    idle = filled ? 0 : idle + 1;
    if (idle >= F) return -1;
  }
}

int64_t nqo_compute_sketch(const nqo_params *p, const uint8_t *seq,
                           uint64_t len, int32_t *sk) {
  const uint32_t F = (uint32_t)1 << p->S;
  for (uint32_t i = 0; i < F; ++i) sk[i] = -1;
  nqo_sketch_accumulate(p, seq, len, sk);
  return nqo_densify(p, sk);
}

/* ---- index --------------------------------------------------------------- */

static inline int slot_valid(int32_t fp, uint32_t R) {
  return fp >= 0 && (uint32_t)fp < R; /* src/niqki_index.cpp:364 */
}

/* threads > 1: the slots are cut into ranges, one per thread -- a bucket belongs to one slot, and every thread
 * walks the genomes in ascending order, so the arrays are those of the single-threaded build (threads <= 0: all). */
nqo_index *nqo_index_build_mt(const nqo_params *p, const int32_t *sketches, uint32_t n, int threads) {
  const uint64_t F = (uint64_t)1 << p->S, R = (uint64_t)1 << p->W;
  nqo_index *ix = (nqo_index *)calloc(1, sizeof(*ix));
  if (!ix) return NULL;
  if (threads <= 0) threads = omp_get_max_threads();
  if ((uint64_t)threads > F) threads = (int)F;
  ix->p = *p;
  ix->n_genomes = n;
  ix->n_buckets = F * R;
  ix->offsets = (uint64_t *)calloc(ix->n_buckets + 1, sizeof(uint64_t));
  if (!ix->offsets) { free(ix); return NULL; }
  /* counting sort by bucket, gids ascending inside a bucket = the order
   * single-threaded push_back produces (:362-370).  Every phase runs over the threads' slot ranges (the 8-byte
   * offsets of all 2^(S+W) buckets are 1 GB at S=15 W=12: their page faults, prefix and copy are most of a small build) */
  uint64_t *cursor = (uint64_t *)malloc(ix->n_buckets * sizeof(uint64_t));
  uint64_t *tsum = (uint64_t *)calloc((size_t)threads + 1, sizeof(uint64_t));
  if (!cursor || !tsum) { free(cursor); free(tsum); nqo_index_free(ix); return NULL; }
  int failed = 0;
#pragma omp parallel num_threads(threads)
  {
    const uint64_t t = (uint64_t)omp_get_thread_num(), nt = (uint64_t)omp_get_num_threads();
    const uint64_t s0 = F * t / nt, s1 = F * (t + 1) / nt;
    for (uint32_t g = 0; g < n; ++g) {
      const int32_t *sk = sketches + (uint64_t)g * F;
      for (uint64_t s = s0; s < s1; ++s)
        if (slot_valid(sk[s], (uint32_t)R)) ix->offsets[s * R + (uint32_t)sk[s] + 1]++;
    }
    uint64_t sum = 0;
    for (uint64_t bkt = s0 * R; bkt < s1 * R; ++bkt) sum += ix->offsets[bkt + 1];
    tsum[t + 1] = sum;
#pragma omp barrier
#pragma omp single
    {
      for (uint64_t i = 0; i < nt; ++i) tsum[i + 1] += tsum[i];
      const uint64_t total = tsum[nt];
      ix->gids = (uint32_t *)malloc((total ? total : 1) * sizeof(uint32_t));
      if (!ix->gids) failed = 1;
    }   /* (implicit barrier) */
    if (!failed) {
      uint64_t run = tsum[t];   /* ids in front of this thread's first bucket */
      for (uint64_t bkt = s0 * R; bkt < s1 * R; ++bkt) {
        cursor[bkt] = run;
        run += ix->offsets[bkt + 1];
        ix->offsets[bkt + 1] = run;
      }
      for (uint32_t g = 0; g < n; ++g) {
        const int32_t *sk = sketches + (uint64_t)g * F;
        for (uint64_t s = s0; s < s1; ++s)
          if (slot_valid(sk[s], (uint32_t)R)) ix->gids[cursor[s * R + (uint32_t)sk[s]]++] = g;
      }
    }
  }
  free(tsum);
  if (failed) { free(cursor); nqo_index_free(ix); return NULL; }
  free(cursor);
  return ix;
}

nqo_index *nqo_index_build(const nqo_params *p, const int32_t *sketches,
                           uint32_t n) {
  return nqo_index_build_mt(p, sketches, n, 1);
}

void nqo_index_free(nqo_index *ix) {
  if (!ix) return;
  free(ix->offsets);
  free(ix->gids);
  free(ix);
}

void nqo_query_counts(const nqo_index *ix, const int32_t *sk, uint32_t *counts) {
  /* src/niqki_index.cpp:633-682; the three counter widths there cannot
   * overflow (count <= F), so one uint32 path restates all of them. */
  const uint64_t F = (uint64_t)1 << ix->p.S, R = (uint64_t)1 << ix->p.W;
  memset(counts, 0, (size_t)ix->n_genomes * sizeof(uint32_t));
  for (uint64_t s = 0; s < F; ++s) {
    if (!slot_valid(sk[s], (uint32_t)R)) continue;
    uint64_t b = s * R + (uint32_t)sk[s];
    for (uint64_t j = ix->offsets[b]; j < ix->offsets[b + 1]; ++j) counts[ix->gids[j]]++;
  }
}

uint64_t nqo_query_gathered(const nqo_index *ix, const int32_t *sk) {
  const uint64_t F = (uint64_t)1 << ix->p.S, R = (uint64_t)1 << ix->p.W;
  uint64_t t = 0;
  for (uint64_t s = 0; s < F; ++s) {
    if (!slot_valid(sk[s], (uint32_t)R)) continue;
    uint64_t b = s * R + (uint32_t)sk[s];
    t += ix->offsets[b + 1] - ix->offsets[b];
  }
  return t;
}

typedef struct { uint32_t count, gid; } hit_t;

static int hit_desc(const void *a, const void *b) {
  /* greater<pair<count,gid>>: src/niqki_index.cpp:685 */
  const hit_t *x = (const hit_t *)a, *y = (const hit_t *)b;
  if (x->count != y->count) return x->count > y->count ? -1 : 1;
  if (x->gid != y->gid) return x->gid > y->gid ? -1 : 1;
  return 0;
}

uint32_t nqo_hits_from_counts(const uint32_t *counts, uint32_t n,
                              uint32_t min_score, uint32_t *hit_counts,
                              uint32_t *hit_gids, uint32_t cap) {
  uint32_t nh = 0;
  for (uint32_t g = 0; g < n; ++g) nh += (counts[g] >= min_score);
  hit_t *h = (hit_t *)malloc((nh ? nh : 1) * sizeof(hit_t));
  uint32_t k = 0;
  for (uint32_t g = 0; g < n; ++g)
    if (counts[g] >= min_score) { h[k].count = counts[g]; h[k].gid = g; ++k; }
  qsort(h, nh, sizeof(hit_t), hit_desc);
  for (uint32_t i = 0; i < nh && i < cap; ++i) {
    hit_counts[i] = h[i].count;
    hit_gids[i] = h[i].gid;
  }
  free(h);
  return nh;
}

void nqo_matrix_range(const nqo_index *ix, uint32_t begin, uint32_t end,
                      uint16_t *counts) {
  /* src/niqki_index.cpp:570-597: per bucket, every member x every member that
   * lies in [begin,end) */
  const uint64_t batch = end - begin;
  memset(counts, 0, (size_t)ix->n_genomes * batch * sizeof(uint16_t));
  for (uint64_t b = 0; b < ix->n_buckets; ++b) {
    uint64_t lo = ix->offsets[b], hi = ix->offsets[b + 1];
    for (uint64_t t = lo; t < hi; ++t) {
      uint32_t gt = ix->gids[t];
      if (gt < begin || gt >= end) continue;
      for (uint64_t a = lo; a < hi; ++a)
        counts[(uint64_t)ix->gids[a] * batch + (gt - begin)]++;
    }
  }
}

/* ---- dump / load ---------------------------------------------------------- */

static void put32(uint8_t *buf, uint64_t cap, uint64_t *pos, uint32_t v) {
  if (buf && *pos + 4 <= cap) memcpy(buf + *pos, &v, 4); /* host endianness like the reference */
  *pos += 4;
}

uint64_t nqo_dump_bytes(const nqo_index *ix, uint8_t *buf, uint64_t cap) {
  /* src/niqki_index.cpp:42-55: lF K H W min_score genome_numbers, then
   * per bucket u32 size + gids */
  uint64_t pos = 0;
  put32(buf, cap, &pos, ix->p.S);
  put32(buf, cap, &pos, ix->p.K);
  put32(buf, cap, &pos, ix->p.H);
  put32(buf, cap, &pos, ix->p.W);
  put32(buf, cap, &pos, ix->p.min_score);
  put32(buf, cap, &pos, ix->n_genomes);
  for (uint64_t b = 0; b < ix->n_buckets; ++b) {
    uint64_t lo = ix->offsets[b], hi = ix->offsets[b + 1];
    put32(buf, cap, &pos, (uint32_t)(hi - lo));
    for (uint64_t j = lo; j < hi; ++j) put32(buf, cap, &pos, ix->gids[j]);
  }
  return pos;
}

nqo_index *nqo_load_bytes(const uint8_t *buf, uint64_t len, uint64_t *consumed) {
  /* src/niqki_index.cpp:63-90 */
  if (len < 24) return NULL;
  uint32_t hdr[6];
  memcpy(hdr, buf, 24);
  nqo_index *ix = (nqo_index *)calloc(1, sizeof(*ix));
  ix->p.S = hdr[0]; ix->p.K = hdr[1]; ix->p.H = hdr[2]; ix->p.W = hdr[3];
  ix->p.min_score = hdr[4];
  ix->p.H0p1 = 0;
  ix->n_genomes = hdr[5];
  ix->n_buckets = ((uint64_t)1 << ix->p.S) << ix->p.W;
  ix->offsets = (uint64_t *)calloc(ix->n_buckets + 1, sizeof(uint64_t));
  /* first pass sizes, second pass payload */
  uint64_t pos = 24, total = 0;
  for (uint64_t b = 0; b < ix->n_buckets; ++b) {
    if (pos + 4 > len) { nqo_index_free(ix); return NULL; }
    uint32_t sz; memcpy(&sz, buf + pos, 4);
    pos += 4 + (uint64_t)sz * 4;
    total += sz;
    ix->offsets[b + 1] = total;
  }
  if (pos > len) { nqo_index_free(ix); return NULL; }
  ix->gids = (uint32_t *)malloc((total ? total : 1) * sizeof(uint32_t));
  pos = 24;
  for (uint64_t b = 0; b < ix->n_buckets; ++b) {
    uint64_t sz = ix->offsets[b + 1] - ix->offsets[b];
    pos += 4;
    memcpy(ix->gids + ix->offsets[b], buf + pos, sz * 4);
    pos += sz * 4;
  }
  if (consumed) *consumed = pos;
  return ix;
}

/* ---- batch helpers (CPU baseline) ---------------------------------------- */

int nqo_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void nqo_sketch_batch(const nqo_params *p, const uint8_t *seqs,
                      const uint64_t *rec_off, uint32_t n, int32_t *sketches,
                      int threads) {
  const uint64_t F = (uint64_t)1 << p->S;
#ifdef _OPENMP
  if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
  for (int64_t i = 0; i < (int64_t)n; ++i)
    nqo_compute_sketch(p, seqs + rec_off[i], rec_off[i + 1] - rec_off[i],
                       sketches + (uint64_t)i * F);
  (void)threads;
}

uint64_t nqo_query_batch(const nqo_index *ix, const int32_t *sketches,
                         uint32_t n, uint64_t *hit_off, uint32_t *hit_counts,
                         uint32_t *hit_gids, uint64_t cap_total, int threads) {
  const uint64_t F = (uint64_t)1 << ix->p.S;
  const uint32_t N = ix->n_genomes;
  uint32_t *nh = (uint32_t *)calloc(n ? n : 1, sizeof(uint32_t));
  uint32_t **hc = (uint32_t **)calloc(n ? n : 1, sizeof(uint32_t *));
  uint32_t **hg = (uint32_t **)calloc(n ? n : 1, sizeof(uint32_t *));
#ifdef _OPENMP
  if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel num_threads(threads)
#endif
  {
    uint32_t *counts = (uint32_t *)malloc((N ? N : 1) * sizeof(uint32_t));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (int64_t i = 0; i < (int64_t)n; ++i) {
      nqo_query_counts(ix, sketches + (uint64_t)i * F, counts);
      uint32_t k = 0;
      for (uint32_t g = 0; g < N; ++g) k += (counts[g] >= ix->p.min_score);
      hc[i] = (uint32_t *)malloc((k ? k : 1) * sizeof(uint32_t));
      hg[i] = (uint32_t *)malloc((k ? k : 1) * sizeof(uint32_t));
      nh[i] = nqo_hits_from_counts(counts, N, ix->p.min_score, hc[i], hg[i], k);
    }
    free(counts);
  }
  uint64_t total = 0;
  for (uint32_t i = 0; i < n; ++i) {
    hit_off[i] = total;
    for (uint32_t j = 0; j < nh[i]; ++j) {
      if (total + j < cap_total) {
        hit_counts[total + j] = hc[i][j];
        hit_gids[total + j] = hg[i][j];
      }
    }
    total += nh[i];
    free(hc[i]);
    free(hg[i]);
  }
  hit_off[n] = total;
  free(nh); free(hc); free(hg);
  (void)threads;
  return total;
}

/* CPU baseline only: the CSR arrays again, first touched by `threads` threads in equal static parts, so that on a
 * multi-socket host the pages are spread over the memory of all the threads that will gather from them
 * (the index is built by one thread: everything would otherwise sit on that thread's node). */
void nqo_index_spread(nqo_index *ix, int threads) {
  const uint64_t nb = ix->n_buckets + 1, tot = ix->offsets[ix->n_buckets];
  uint64_t *no = (uint64_t *)malloc(nb * sizeof(uint64_t));
  uint32_t *ng = (uint32_t *)malloc((tot ? tot : 1) * sizeof(uint32_t));
  if (!no || !ng) { free(no); free(ng); return; }
#ifdef _OPENMP
  if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel num_threads(threads)
#endif
  {
#ifdef _OPENMP
#pragma omp for schedule(static) nowait
#endif
    for (int64_t i = 0; i < (int64_t)nb; ++i) no[i] = ix->offsets[i];
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (int64_t i = 0; i < (int64_t)tot; ++i) ng[i] = ix->gids[i];
  }
  free(ix->offsets);
  free(ix->gids);
  ix->offsets = no;
  ix->gids = ng;
  (void)threads;
}

uint64_t nqo_fnv1a64(const void *data, uint64_t len) {
  const uint8_t *p = (const uint8_t *)data;
  uint64_t h = 0xcbf29ce484222325ULL;
  for (uint64_t i = 0; i < len; ++i) { h ^= p[i]; h *= 0x100000001b3ULL; }
  return h;
}
