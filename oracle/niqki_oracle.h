/*
 * niqki_oracle.h -- CPU restatement of the NIQKI sketch/query hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle and the timed CPU
 * baseline ("port").  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product library
 * (niqki_amd/csrc) never links, includes or calls anything in oracle/.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle_*.py)
 * against
 *   - known-answer values of the reference hash pair / fingerprint
 *     (SURVEY.md 8a rows a5,a6),
 *   - golden vectors produced by the reference's own sources compiled into
 *     oracle/_ref (see oracle/Makefile, oracle/make_goldens.py) and committed
 *     under tests/golden/,
 *   - the 9 E. coli sketches / hit counts / README matrix when
 *     /root/reference is present.
 *
 * Every function cites the reference lines (relative to /root/reference)
 * whose behaviour it restates.  Plain C99, no dependencies beyond libc
 * (+ OpenMP for the batch helpers).
 */
#ifndef NIQKI_ORACLE_H
#define NIQKI_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Sketch parameters: src/niqki_index.cpp:13-29 (constructor-derived constants). */
typedef struct nqo_params {
  uint32_t K;         /* k-mer length, 1..31 */
  uint32_t S;         /* lF: log2 of the number of sketch slots F */
  uint32_t W;         /* fingerprint bits */
  uint32_t H;         /* HyperLogLog bits inside the fingerprint (M = W-H) */
  uint32_t min_score; /* (uint32)(min_fract * F), src/niqki_index.cpp:22 */
  uint32_t H0p1;      /* 0, or 1 + the constructor's H when select_best_H (-G) replaced it
                         afterwards: mask_M / maximal_remainder stay at that H's values */
} nqo_params;

/* src/niqki_index.cpp:22  min_score = min_fract*F  (double -> uint32 truncation) */
uint32_t nqo_min_score(double min_fract, uint32_t S);

/* src/niqki_index.cpp:291-296 / :300-305 */
uint64_t nqo_rev64(uint64_t x);
uint64_t nqo_unrev64(uint64_t x);

/* src/niqki_index.cpp:277-287 (+ asm_log2 :199-206).  h==0 -> 0 (bsr UB in the
 * reference, observed result 0). */
int32_t nqo_fingerprint(uint64_t h, uint32_t W, uint32_t H);

/* get_fingerprint with the stale mask_M / maximal_remainder a `-G` run has:
 * src/niqki_index.cpp:126-138 changes H and M only.  H0 = constructor's H. */
int32_t nqo_fingerprint_stale(uint64_t h, uint32_t W, uint32_t H, uint32_t H0);

/* select_best_H + score_H: src/niqki_index.cpp:126-164.  Returns the H the
 * reference ends with (the given H when no candidate scores above 0). */
uint32_t nqo_select_best_H(double genome_size, uint32_t S, uint32_t W, uint32_t H);

/* src/niqki_index.cpp:308-310 */
uint64_t nqo_hash_family(uint64_t x, uint32_t step);

/* Rolling canonical k-mers of one record min-accumulated into sk[F]
 * (-1 = empty), WITHOUT densification: src/niqki_index.cpp:335-356.
 * Returns the number of k-mers processed (len-K, or 0 when len<=K). */
uint64_t nqo_sketch_accumulate(const nqo_params *p, const uint8_t *seq,
                               uint64_t len, int32_t *sk);

/* src/niqki_index.cpp:313-331, serial, order dependent.  Returns the number
 * of full passes started, or -1 when the reference would loop forever
 * (no occupied cell, or a whole period of F passes without a fill); the
 * sketch is then left with its empty cells. */
int64_t nqo_densify(const nqo_params *p, int32_t *sk);

/* compute_sketch on a fresh sketch: fill with -1, accumulate, densify.
 * src/niqki_index.cpp:335-358.  Returns nqo_densify's result. */
int64_t nqo_compute_sketch(const nqo_params *p, const uint8_t *seq,
                           uint64_t len, int32_t *sk);

/* ---- inverted index ---------------------------------------------------- */

/* The reference's vector<gid> Buckets[2^W * F] (src/niqki_index.h:55) kept as
 * CSR: bucket b = fp + slot*2^W holds gids[offsets[b] .. offsets[b+1]) in
 * insertion order (ascending gid when built single threaded). */
typedef struct nqo_index {
  nqo_params p;
  uint32_t n_genomes;
  uint64_t n_buckets; /* F * 2^W */
  uint64_t *offsets;  /* n_buckets + 1 */
  uint32_t *gids;     /* offsets[n_buckets] */
} nqo_index;

/* insert_sketch for gid = 0..n-1 in order: src/niqki_index.cpp:362-370.
 * sketches is n x F int32, row major. */
nqo_index *nqo_index_build(const nqo_params *p, const int32_t *sketches,
                           uint32_t n);
/* the same arrays, built by `threads` threads over slot ranges (<= 0: all) */
nqo_index *nqo_index_build_mt(const nqo_params *p, const int32_t *sketches, uint32_t n, int threads);
void nqo_index_free(nqo_index *ix);

/* query_sketch counting loop: src/niqki_index.cpp:633-682.  counts[n_genomes]. */
void nqo_query_counts(const nqo_index *ix, const int32_t *sk, uint32_t *counts);

/* Threshold + sort: src/niqki_index.cpp:646-650,685.  Writes at most cap
 * (count,gid) pairs ordered by descending (count, gid); returns the number of
 * hits (which may exceed cap). */
uint32_t nqo_hits_from_counts(const uint32_t *counts, uint32_t n,
                              uint32_t min_score, uint32_t *hit_counts,
                              uint32_t *hit_gids, uint32_t cap);

/* Sum over slots of the length of the bucket the query touches (the T of
 * SURVEY.md 8d). */
uint64_t nqo_query_gathered(const nqo_index *ix, const int32_t *sk);

/* query_range counting loop: src/niqki_index.cpp:570-597.
 * counts is n_genomes x (end-begin) uint16, counts[a*(end-begin)+t]. */
void nqo_matrix_range(const nqo_index *ix, uint32_t begin, uint32_t end,
                      uint16_t *counts);

/* dump_index_disk payload before gzip, names excluded:
 * src/niqki_index.cpp:42-55.  Returns bytes written (call with buf==NULL to
 * size). */
uint64_t nqo_dump_bytes(const nqo_index *ix, uint8_t *buf, uint64_t cap);

/* Loading constructor, gunzipped bytes: src/niqki_index.cpp:63-90.  *consumed
 * receives the offset of the first name byte. */
nqo_index *nqo_load_bytes(const uint8_t *buf, uint64_t len, uint64_t *consumed);

/* ---- batch helpers for the timed CPU baseline (OpenMP over records, one
 * record per thread like src/niqki_index.cpp:415,525) ---------------------- */

/* n records stored back to back in seqs; record i = seqs[rec_off[i]..rec_off[i+1]).
 * sketches is n x F.  threads<=0 -> omp default. */
void nqo_sketch_batch(const nqo_params *p, const uint8_t *seqs,
                      const uint64_t *rec_off, uint32_t n, int32_t *sketches,
                      int threads);

/* Sketch + query + threshold + sort per record.  hit_off[n+1] prefix offsets
 * into hit_counts/hit_gids (capacity cap_total; hits beyond it are dropped,
 * the returned total tells).  Returns total hits. */
uint64_t nqo_query_batch(const nqo_index *ix, const int32_t *sketches,
                         uint32_t n, uint64_t *hit_off, uint32_t *hit_counts,
                         uint32_t *hit_gids, uint64_t cap_total, int threads);

int nqo_max_threads(void);

/* CPU baseline: re-places the index' arrays so that `threads` threads first touch equal parts (NUMA). */
void nqo_index_spread(nqo_index *ix, int threads);

/* 64-bit FNV-1a over a byte range (used for golden checksums). */
uint64_t nqo_fnv1a64(const void *data, uint64_t len);

#ifdef __cplusplus
}
#endif
#endif
