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
// (:396-401, :486-490).  Host code only: no kernel lives here.
#include "nq_handle.h"

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <vector>

namespace {

enum Kind { kSketch = 0, kInsert = 1, kQuery = 2, kQuerySeq = 3 };

struct Request {
  Kind kind;
  // inputs
  const uint8_t *seq = nullptr;
  uint64_t len = 0;
  const int32_t *sketch_in = nullptr;
  // outputs
  int32_t *sketch_out = nullptr;
  uint32_t *gid_out = nullptr;
  uint64_t *n_hits = nullptr;
  uint32_t *hit_counts = nullptr, *hit_gids = nullptr;
  uint64_t capacity = 0;
  // completion
  bool done = false;
  int rc = NIQKI_OK;
};

struct Combiner {
  std::mutex m;
  std::condition_variable cv;
  std::vector<Request *> pending;
  bool leader = false;
  uint64_t batches = 0, requests = 0, largest = 0;
  // leader-only scratch (one leader at a time)
  std::vector<uint8_t> seqs;
  std::vector<uint64_t> rec_off;
  std::vector<int32_t> sk;
  std::vector<uint64_t> off;
  std::vector<uint32_t> hc, hg;
};

constexpr size_t kMaxBatch = 4096;

// one kind's requests of a batch through the single-caller ABI
void run_kind(niqki_index *ix, Combiner &c, std::vector<Request *> &rs) {
  const uint32_t n = (uint32_t)rs.size();
  const size_t F = ix->d.F;
  const Kind kind = rs[0]->kind;
  int rc = NIQKI_OK;
  if (kind == kSketch || kind == kQuerySeq) {
    uint64_t total = 0;
    c.rec_off.assign(n + 1, 0);
    for (uint32_t i = 0; i < n; ++i) { c.rec_off[i] = total; total += rs[i]->len; }
    c.rec_off[n] = total;
    c.seqs.resize(total + NIQKI_SEQ_PAD);
    for (uint32_t i = 0; i < n; ++i)
      if (rs[i]->len) std::memcpy(c.seqs.data() + c.rec_off[i], rs[i]->seq, rs[i]->len);
    c.sk.resize((size_t)n * F);
    rc = niqki_sketch(ix, c.seqs.data(), c.rec_off.data(), n, nullptr, n, c.sk.data(), NIQKI_MEM_HOST);
    if (rc == NIQKI_OK && kind == kSketch)
      for (uint32_t i = 0; i < n; ++i) std::memcpy(rs[i]->sketch_out, c.sk.data() + (size_t)i * F, F * 4);
  } else {
    c.sk.resize((size_t)n * F);
    for (uint32_t i = 0; i < n; ++i) std::memcpy(c.sk.data() + (size_t)i * F, rs[i]->sketch_in, F * 4);
  }
  if (rc == NIQKI_OK && kind == kInsert) {
    const uint32_t first = ix->n_genomes;
    rc = niqki_insert(ix, c.sk.data(), n, NIQKI_MEM_HOST);
    if (rc == NIQKI_OK)
      for (uint32_t i = 0; i < n; ++i)
        if (rs[i]->gid_out) *rs[i]->gid_out = first + i;
  }
  if (rc == NIQKI_OK && (kind == kQuery || kind == kQuerySeq)) {
    c.off.assign(n + 1, 0);
    size_t cap = std::max<size_t>(c.hc.size(), (size_t)n * 64);
    for (int attempt = 0; attempt < 2 && rc == NIQKI_OK; ++attempt) {
      c.hc.resize(cap);
      c.hg.resize(cap);
      rc = niqki_query(ix, c.sk.data(), n, c.off.data(), c.hc.data(), c.hg.data(), cap, NIQKI_MEM_HOST);
      if (rc != NIQKI_OK || c.off[n] <= cap) break;
      cap = (size_t)c.off[n];   // hit_off is exact whatever the capacity: once more with room for all
    }
    if (rc == NIQKI_OK)
      for (uint32_t i = 0; i < n; ++i) {
        const uint64_t lo = c.off[i], k = c.off[i + 1] - lo, w = std::min<uint64_t>(k, rs[i]->capacity);
        *rs[i]->n_hits = k;   // (may exceed the caller's capacity: the first `capacity` hits are written)
        if (w) {
          std::memcpy(rs[i]->hit_counts, c.hc.data() + lo, w * 4);
          std::memcpy(rs[i]->hit_gids, c.hg.data() + lo, w * 4);
        }
      }
  }
  for (Request *r : rs) r->rc = rc;
}

int submit(niqki_index *ix, Request &r) {
  if (!ix) return NIQKI_E_INVALID;
  Combiner *c;
  {
    static std::mutex make;   // the handle's combiner is made by its first *_shared call
    std::lock_guard<std::mutex> g(make);
    if (!ix->shared_state) ix->shared_state = new Combiner();
    c = (Combiner *)ix->shared_state;
  }
  std::unique_lock<std::mutex> lk(c->m);
  c->pending.push_back(&r);
  if (c->leader) {   // a batch is on the GPU: wait for a leader to take this request along
    c->cv.wait(lk, [&] { return r.done || !c->leader; });
    if (r.done) return r.rc;
  }
  // leader: batches of whatever is pending, until nothing is (my own request is in the first of them)
  c->leader = true;
  while (!c->pending.empty()) {
    std::vector<Request *> batch;
    const size_t take = std::min(c->pending.size(), kMaxBatch);
    batch.assign(c->pending.begin(), c->pending.begin() + take);
    c->pending.erase(c->pending.begin(), c->pending.begin() + take);
    lk.unlock();
    // inserts first (arrival order), then sketches, then queries: requests of one thread never overlap, and the
    // order between different threads' requests is as undefined as in the reference's parallel loops
    try {
      for (Kind k : {kInsert, kSketch, kQuery, kQuerySeq}) {
        std::vector<Request *> rs;
        for (Request *q : batch)
          if (q->kind == k) rs.push_back(q);
        if (!rs.empty()) run_kind(ix, *c, rs);
      }
    } catch (...) {   // (an allocation of the leader's scratch failed: the waiting threads must still be released)
      for (Request *q : batch)
        if (q->rc == NIQKI_OK) q->rc = NIQKI_E_NOMEM;
    }
    lk.lock();
    c->batches += 1;
    c->requests += batch.size();
    c->largest = std::max<uint64_t>(c->largest, batch.size());
    for (Request *q : batch) q->done = true;
    c->cv.notify_all();
    if (r.done && !c->pending.empty()) {
      // my own request is answered: hand the leadership to one of the waiting threads
      c->leader = false;
      c->cv.notify_all();
      return r.rc;
    }
  }
  c->leader = false;
  c->cv.notify_all();
  return r.rc;
}

}  // namespace

namespace nqi {
void shared_free(niqki_index *ix) {
  delete (Combiner *)ix->shared_state;
  ix->shared_state = nullptr;
}
}  // namespace nqi

extern "C" {

int niqki_sketch_shared(niqki_index *ix, const uint8_t *seq, uint64_t len, int32_t *sketch) {
  if (!ix || (!seq && len) || !sketch) return NIQKI_E_INVALID;
  Request r;
  r.kind = kSketch; r.seq = seq; r.len = len; r.sketch_out = sketch;
  return submit(ix, r);
}

int niqki_insert_shared(niqki_index *ix, const int32_t *sketch, uint32_t *genome_id) {
  if (!ix || !sketch) return NIQKI_E_INVALID;
  Request r;
  r.kind = kInsert; r.sketch_in = sketch; r.gid_out = genome_id;
  return submit(ix, r);
}

int niqki_query_shared(niqki_index *ix, const int32_t *sketch, uint64_t *n_hits, uint32_t *hit_counts, uint32_t *hit_gids,
                       uint64_t capacity) {
  if (!ix || !sketch || !n_hits || (capacity && (!hit_counts || !hit_gids))) return NIQKI_E_INVALID;
  Request r;
  r.kind = kQuery; r.sketch_in = sketch; r.n_hits = n_hits; r.hit_counts = hit_counts; r.hit_gids = hit_gids; r.capacity = capacity;
  return submit(ix, r);
}

int niqki_query_sequence_shared(niqki_index *ix, const uint8_t *seq, uint64_t len, uint64_t *n_hits, uint32_t *hit_counts,
                                uint32_t *hit_gids, uint64_t capacity) {
  if (!ix || (!seq && len) || !n_hits || (capacity && (!hit_counts || !hit_gids))) return NIQKI_E_INVALID;
  Request r;
  r.kind = kQuerySeq; r.seq = seq; r.len = len; r.n_hits = n_hits; r.hit_counts = hit_counts; r.hit_gids = hit_gids; r.capacity = capacity;
  return submit(ix, r);
}

int niqki_shared_stats(const niqki_index *ix, uint64_t *batches, uint64_t *requests, uint64_t *largest_batch) {
  if (!ix) return NIQKI_E_INVALID;
  const Combiner *c = (const Combiner *)ix->shared_state;
  if (batches) *batches = c ? c->batches : 0;
  if (requests) *requests = c ? c->requests : 0;
  if (largest_batch) *largest_batch = c ? c->largest : 0;
  return NIQKI_OK;
}

}  // extern "C"
