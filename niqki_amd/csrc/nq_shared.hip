// nq_shared.hip -- the reference's threading contract at the boundary.
//
// The reference's file drivers call compute_sketch / insert_sketch / query_sketch from every thread of an
// `omp parallel` region on ONE Index (src/niqki_index.cpp:391-401, :415-428, :479-490, :525-538; query_* are
// const, insert takes striped locks).  A niqki_index handle is single-caller: its stream, workspaces and staged
// state belong to one call at a time.  The *_shared entry points below restore the reference's contract for a
// maintainer who keeps the reference's per-record loops ("Minimal patch"): any number of host
// threads may call them on one handle at the same time.  Callers are COMBINED: the first thread to arrive becomes
// the batch's leader, threads that arrive while a batch is on the GPU queue up and form the next batch, so with T
// calling threads the GPU sees batches of about T records -- the per-batch calls of the C ABI, fed by the
// reference's own loop structure.  Results are those of the single-caller calls (which these call); genome ids of
// concurrent inserts are handed out in batch order = arrival order, like the reference's `omp critical` id counter
// (:396-401, :486-490).  Host code only: no kernel lives here; the batching itself is nq_combiner.h, which knows
// nothing of HIP (the CPU test suite runs it on a fake engine under ThreadSanitizer / AddressSanitizer).
#include "nq_combiner.h"
#include "nq_handle.h"

#include <mutex>
#include <new>

namespace {

using nqc::Combiner;
using nqc::Request;

std::mutex g_make;   // guards niqki_index::shared_state: made by the handle's first *_shared call, read by the stats call

// the single-caller C ABI as the combiner's engine (host memory)
int eng_sketch(void *ctx, const uint8_t *seqs, const uint64_t *rec_off, uint32_t n, int32_t *sketches) {
  return niqki_sketch((niqki_index *)ctx, seqs, rec_off, n, nullptr, n, sketches, NIQKI_MEM_HOST);
}
int eng_insert(void *ctx, const int32_t *sketches, uint32_t n, uint32_t *first_gid) {
  niqki_index *ix = (niqki_index *)ctx;
  *first_gid = ix->n_genomes;
  return niqki_insert(ix, sketches, n, NIQKI_MEM_HOST);
}
int eng_query(void *ctx, const int32_t *sketches, uint32_t n, uint64_t *hit_off, uint32_t *hc, uint32_t *hg, uint64_t cap) {
  return niqki_query((niqki_index *)ctx, sketches, n, hit_off, hc, hg, cap, NIQKI_MEM_HOST);
}

int submit(niqki_index *ix, Request &r) {
  if (!ix) return NIQKI_E_INVALID;
  Combiner *c;
  {
    std::lock_guard<std::mutex> g(g_make);
    if (!ix->shared_state) ix->shared_state = new (std::nothrow) Combiner();
    c = (Combiner *)ix->shared_state;
  }
  if (!c) return NIQKI_E_NOMEM;
  nqc::Engine e;
  e.ctx = ix;
  e.F = ix->d.F;
  e.sketch = eng_sketch;
  e.insert = eng_insert;
  e.query = eng_query;
  return c->submit(e, r);
}

}  // namespace

namespace nqi {
void shared_free(niqki_index *ix) {
  std::lock_guard<std::mutex> g(g_make);
  delete (Combiner *)ix->shared_state;
  ix->shared_state = nullptr;
}
}  // namespace nqi

extern "C" {

int niqki_sketch_shared(niqki_index *ix, const uint8_t *seq, uint64_t len, int32_t *sketch) {
  if (!ix || (!seq && len) || !sketch) return NIQKI_E_INVALID;
  Request r;
  r.kind = nqc::kSketch; r.seq = seq; r.len = len; r.sketch_out = sketch;
  return submit(ix, r);
}

int niqki_insert_shared(niqki_index *ix, const int32_t *sketch, uint32_t *genome_id) {
  if (!ix || !sketch) return NIQKI_E_INVALID;
  Request r;
  r.kind = nqc::kInsert; r.sketch_in = sketch; r.gid_out = genome_id;
  return submit(ix, r);
}

int niqki_query_shared(niqki_index *ix, const int32_t *sketch, uint64_t *n_hits, uint32_t *hit_counts, uint32_t *hit_gids,
                       uint64_t capacity) {
  if (!ix || !sketch || !n_hits || (capacity && (!hit_counts || !hit_gids))) return NIQKI_E_INVALID;
  Request r;
  r.kind = nqc::kQuery; r.sketch_in = sketch; r.n_hits = n_hits; r.hit_counts = hit_counts; r.hit_gids = hit_gids; r.capacity = capacity;
  return submit(ix, r);
}

int niqki_query_sequence_shared(niqki_index *ix, const uint8_t *seq, uint64_t len, uint64_t *n_hits, uint32_t *hit_counts,
                                uint32_t *hit_gids, uint64_t capacity) {
  if (!ix || (!seq && len) || !n_hits || (capacity && (!hit_counts || !hit_gids))) return NIQKI_E_INVALID;
  Request r;
  r.kind = nqc::kQuerySeq; r.seq = seq; r.len = len; r.n_hits = n_hits; r.hit_counts = hit_counts; r.hit_gids = hit_gids; r.capacity = capacity;
  return submit(ix, r);
}

int niqki_shared_stats(const niqki_index *ix, uint64_t *batches, uint64_t *requests, uint64_t *largest_batch) {
  if (!ix) return NIQKI_E_INVALID;
  Combiner *c;
  {
    std::lock_guard<std::mutex> g(g_make);
    c = (Combiner *)ix->shared_state;
  }
  if (batches) *batches = 0;
  if (requests) *requests = 0;
  if (largest_batch) *largest_batch = 0;
  if (c) c->stats(batches, requests, largest_batch);   // (under the combiner's own mutex: callable beside *_shared calls)
  return NIQKI_OK;
}

}  // extern "C"
