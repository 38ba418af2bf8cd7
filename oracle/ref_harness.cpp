// ref_harness.cpp -- thin extern "C" access to the REAL reference implementation.
//
// TEST INFRASTRUCTURE ONLY.  This file is our own code; it contains no
// reference source.  oracle/Makefile compiles it together with the reference's
// own src/niqki_index.cpp and src/genome.cpp, taken where they lie under
// /root/reference, into oracle/_ref/libniqki_ref.so (git-ignored).  It exists
// so that oracle/make_goldens*.py and tests/test_oracle_golden.py can pin the
// C restatement (niqki_oracle.c) against the reference's actual outputs, and so that
// bench.py's cpu_baseline leg can time the reference's own code beside the port.
// Nothing here is reachable from the product library.
//
// Every member of the reference's Index class is public
// (src/niqki_index.h:35-213), so the harness calls the hot-path methods
// directly.
#include "niqki_index.h"

#include <omp.h>

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

extern "C" {

// Index(lF,K,W,H,out,min_fract): src/niqki_index.cpp:13-38.  The constructor
// opens its output file, so callers pass a scratch path.
void *ref_create(uint32_t lF, uint32_t K, uint32_t W, uint32_t H,
                 const char *out_path, double min_fract) {
  return new Index(lF, K, W, H, std::string(out_path), min_fract);
}

void ref_destroy(void *h) { delete static_cast<Index *>(h); }

// select_best_H: src/niqki_index.cpp:126-138 (prints "I chosed H=...")
uint32_t ref_select_best_H(void *h, double genome_size) {
  Index *ix = static_cast<Index *>(h);
  ix->select_best_H(genome_size);
  return ix->H;
}

uint32_t ref_min_score(void *h) { return static_cast<Index *>(h)->min_score; }

uint64_t ref_rev64(void *h, uint64_t x) { return static_cast<Index *>(h)->revhash64(x); }
uint64_t ref_unrev64(void *h, uint64_t x) { return static_cast<Index *>(h)->unrevhash64(x); }
int32_t ref_fingerprint(void *h, uint64_t x) { return static_cast<Index *>(h)->get_fingerprint(x); }
uint64_t ref_hash_family(void *h, uint64_t x, uint32_t step) {
  return static_cast<Index *>(h)->hash_family(x, step);
}

// compute_sketch on a fresh vector: src/niqki_index.cpp:335-358
void ref_compute_sketch(void *h, const char *seq, uint64_t len, int32_t *out) {
  Index *ix = static_cast<Index *>(h);
  std::string s(seq, len);
  std::vector<int32_t> sk;
  ix->compute_sketch(s, sk);
  std::memcpy(out, sk.data(), sk.size() * sizeof(int32_t));
}

// sketch_densification alone: src/niqki_index.cpp:313-331
void ref_densify(void *h, int32_t *sk, uint32_t empty) {
  Index *ix = static_cast<Index *>(h);
  std::vector<int32_t> v(sk, sk + ix->F);
  ix->sketch_densification(v, empty);
  std::memcpy(sk, v.data(), v.size() * sizeof(int32_t));
}

// insert_sketch + the bookkeeping the file drivers do around it
// (src/niqki_index.cpp:362-370, :396-401)
void ref_insert_sketch(void *h, const int32_t *sk, const char *name) {
  Index *ix = static_cast<Index *>(h);
  std::vector<int32_t> v(sk, sk + ix->F);
  uint32_t id = ix->genome_numbers++;
  ix->filenames.push_back(std::string(name));
  ix->insert_sketch(v, id);
}

// query_sketch: src/niqki_index.cpp:633-687.  Returns the number of hits and
// writes up to cap (count,gid) pairs in the reference's output order.
uint32_t ref_query_sketch(void *h, const int32_t *sk, uint32_t *counts,
                          uint32_t *gids, uint32_t cap) {
  Index *ix = static_cast<Index *>(h);
  std::vector<int32_t> v(sk, sk + ix->F);
  query_output r = ix->query_sketch(v);
  for (uint32_t i = 0; i < r.size() && i < cap; ++i) {
    counts[i] = r[i].first;
    gids[i] = r[i].second;
  }
  return (uint32_t)r.size();
}

// ---- the reference's threaded driver loops without their file reading ------------------------------------
// Its drivers call compute_sketch / insert_sketch / query_sketch from every thread of an `omp parallel` region
// on one Index (src/niqki_index.cpp:391-401, :479-490, :525-538).  These three do the same on records that are
// already in memory, so that bench.py can time the REAL reference on the host's threads beside the port
// (cpu_baseline.reference): one record per thread at a time, the reference's own methods, its own locks.

// compute_sketch of n records (record i = seqs[rec_off[i] .. rec_off[i+1])) -> out[n][F]
void ref_sketch_batch(void *h, const char *seqs, const uint64_t *rec_off, uint32_t n, int32_t *out, int threads) {
  Index *ix = static_cast<Index *>(h);
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : omp_get_max_threads())
  for (uint32_t i = 0; i < n; ++i) {
    std::string s(seqs + rec_off[i], rec_off[i + 1] - rec_off[i]);
    std::vector<int32_t> sk;
    ix->compute_sketch(s, sk);
    std::memcpy(out + (size_t)i * ix->F, sk.data(), sk.size() * sizeof(int32_t));
  }
}

// insert_sketch of n sketches: ids in order (the bookkeeping of :396-401 done up front), the inserts themselves
// from all threads under the reference's striped locks (:365-367), like insert_file_of_file_whole (:479-490)
void ref_insert_batch(void *h, const int32_t *sk, uint32_t n, int threads) {
  Index *ix = static_cast<Index *>(h);
  const uint32_t first = ix->genome_numbers;
  for (uint32_t i = 0; i < n; ++i) ix->filenames.push_back("g" + std::to_string(first + i));
  ix->genome_numbers += n;
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads > 0 ? threads : omp_get_max_threads())
  for (uint32_t i = 0; i < n; ++i) {
    std::vector<int32_t> v(sk + (size_t)i * ix->F, sk + (size_t)(i + 1) * ix->F);
    ix->insert_sketch(v, first + i);
  }
}

// query_sketch of n sketches (:525-538 without the output): hit_off[n+1] exact; the first `cap` hits written, in
// query order, each list in the reference's order.  Returns the total.
uint64_t ref_query_batch(void *h, const int32_t *sk, uint32_t n, uint64_t *hit_off, uint32_t *counts, uint32_t *gids,
                         uint64_t cap, int threads) {
  Index *ix = static_cast<Index *>(h);
  std::vector<query_output> res(n);
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : omp_get_max_threads())
  for (uint32_t i = 0; i < n; ++i) {
    std::vector<int32_t> v(sk + (size_t)i * ix->F, sk + (size_t)(i + 1) * ix->F);
    res[i] = ix->query_sketch(v);
  }
  uint64_t at = 0;
  hit_off[0] = 0;
  for (uint32_t i = 0; i < n; ++i) {
    for (size_t j = 0; j < res[i].size(); ++j)
      if (at + j < cap) { counts[at + j] = res[i][j].first; gids[at + j] = res[i][j].second; }
    at += res[i].size();
    hit_off[i + 1] = at;
  }
  return at;
}

uint64_t ref_bucket_size(void *h, uint64_t bucket) {
  return static_cast<Index *>(h)->Buckets[bucket].size();
}

// dump_index_disk: src/niqki_index.cpp:42-59 (gzip file)
void ref_dump(void *h, const char *path) {
  static_cast<Index *>(h)->dump_index_disk(std::string(path));
}

}  // extern "C"
