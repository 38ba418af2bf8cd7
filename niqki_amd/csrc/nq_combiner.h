// nq_combiner.h -- the reference's threading contract at the boundary, as host code that knows nothing of HIP.
//
// The reference's file drivers call compute_sketch / insert_sketch / query_sketch from every thread of an
// `omp parallel` region on ONE Index (src/niqki_index.cpp:391-401, :415-428, :479-490, :525-538; query_* are
// const, insert takes striped locks).  A niqki_index handle is single-caller.  The Combiner restores the
// reference's contract on top of a single-caller ENGINE (three batch calls, below): any number of host threads
// submit one request each; the first thread to arrive becomes the batch's leader, threads that arrive while a
// batch is on the engine queue up and form the next batch, so with T calling threads the engine sees batches of
// about T records.  Genome ids of concurrent inserts are handed out in batch order = arrival order, like the
// reference's `omp critical` id counter (:396-401, :486-490).
//
// nq_shared.hip binds it to niqki_sketch / niqki_insert / niqki_query; the CPU test suite binds it to a fake engine
// and runs it under ThreadSanitizer and AddressSanitizer.
#pragma once
#include "../../include/niqki_hip.h"

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <vector>

namespace nqc {

enum Kind { kSketch = 0, kInsert = 1, kQuery = 2, kQuerySeq = 3 };

struct Request {
  Kind kind = kSketch;
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
  bool done = false;   // answered (under the combiner's mutex)
  bool ran = false;    // its engine call has come back: rc is that call's
  int rc = NIQKI_OK;
};

// The single-caller engine behind a combiner: host-memory batch calls with the meaning of niqki_sketch (one
// record per sketch), niqki_insert (*first_gid = id of the batch's first sketch) and niqki_query (hit_off is
// exact also when the call returns NIQKI_E_CAPACITY).  F = cells per sketch.
struct Engine {
  void *ctx = nullptr;
  uint32_t F = 0;
  int (*sketch)(void *ctx, const uint8_t *seqs, const uint64_t *rec_off, uint32_t n, int32_t *sketches) = nullptr;
  int (*insert)(void *ctx, const int32_t *sketches, uint32_t n, uint32_t *first_gid) = nullptr;
  int (*query)(void *ctx, const int32_t *sketches, uint32_t n, uint64_t *hit_off, uint32_t *hit_counts, uint32_t *hit_gids,
               uint64_t capacity) = nullptr;
};

constexpr size_t kMaxBatch = 4096;
constexpr size_t kSeqPad = NIQKI_SEQ_PAD;

class Combiner {
 public:
  // One request of the calling thread; returns when it has been answered.  No exception leaves this call: an
  // allocation failure of the leader's scratch answers the requests it concerns with NIQKI_E_NOMEM, and the
  // waiting threads are always released.
  int submit(const Engine &e, Request &r) noexcept {
    std::unique_lock<std::mutex> lk(m_);
    try {
      pending_.push_back(&r);
    } catch (...) {
      return NIQKI_E_NOMEM;
    }
    if (leader_) {   // a batch is on the engine: wait for a leader to take this request along
      cv_.wait(lk, [&] { return r.done || !leader_; });
      if (r.done) return r.rc;
    }
    // leader: batches of whatever is pending, until nothing is (my own request is in the first of them)
    leader_ = true;
    while (!pending_.empty()) {
      bool listed = false;
      try {
        const size_t take = std::min(pending_.size(), kMaxBatch);
        batch_.assign(pending_.begin(), pending_.begin() + take);
        pending_.erase(pending_.begin(), pending_.begin() + take);
        listed = true;
      } catch (...) {
      }
      if (!listed) {   // not even the list of the batch could be made: everything pending fails, nobody waits on
        for (Request *q : pending_) { q->rc = NIQKI_E_NOMEM; q->done = true; }
        pending_.clear();
        break;
      }
      lk.unlock();
      run_batch(e);
      lk.lock();
      batches_ += 1;
      requests_ += batch_.size();
      largest_ = std::max<uint64_t>(largest_, batch_.size());
      for (Request *q : batch_) q->done = true;
      cv_.notify_all();
      if (r.done && !pending_.empty()) {
        // my own request is answered: hand the leadership to one of the waiting threads
        leader_ = false;
        cv_.notify_all();
        return r.rc;
      }
    }
    leader_ = false;
    cv_.notify_all();
    return r.rc;
  }

  void stats(uint64_t *batches, uint64_t *requests, uint64_t *largest) {
    std::lock_guard<std::mutex> g(m_);
    if (batches) *batches = batches_;
    if (requests) *requests = requests_;
    if (largest) *largest = largest_;
  }

 private:
  // inserts first (arrival order), then sketches, then queries: requests of one thread never overlap, and the
  // order between different threads' requests is as undefined as in the reference's parallel loops.  A kind whose
  // scratch cannot be allocated fails with NIQKI_E_NOMEM -- that kind's requests only: an insert that has gone
  // through keeps its NIQKI_OK and its id (a caller that retried it would insert the genome twice).
  void run_batch(const Engine &e) noexcept {
    for (Kind k : {kInsert, kSketch, kQuery, kQuerySeq}) {
      try {
        rs_.clear();
        for (Request *q : batch_)
          if (q->kind == k) rs_.push_back(q);
        if (!rs_.empty()) run_kind(e, k);
      } catch (...) {
        for (Request *q : batch_)
          if (q->kind == k && !q->ran) q->rc = NIQKI_E_NOMEM;
      }
    }
  }

  // one kind's requests of a batch (rs_) through the single-caller engine
  void run_kind(const Engine &e, Kind kind) {
    const uint32_t n = (uint32_t)rs_.size();
    const size_t F = e.F;
    int rc = NIQKI_OK;
    if (kind == kSketch || kind == kQuerySeq) {
      uint64_t total = 0;
      rec_off_.assign((size_t)n + 1, 0);
      for (uint32_t i = 0; i < n; ++i) { rec_off_[i] = total; total += rs_[i]->len; }
      rec_off_[n] = total;
      seqs_.resize(total + kSeqPad);
      for (uint32_t i = 0; i < n; ++i)
        if (rs_[i]->len) std::memcpy(seqs_.data() + rec_off_[i], rs_[i]->seq, rs_[i]->len);
      sk_.resize((size_t)n * F);
      rc = e.sketch(e.ctx, seqs_.data(), rec_off_.data(), n, sk_.data());
      if (rc == NIQKI_OK && kind == kSketch)
        for (uint32_t i = 0; i < n; ++i) std::memcpy(rs_[i]->sketch_out, sk_.data() + (size_t)i * F, F * 4);
    } else {
      sk_.resize((size_t)n * F);
      for (uint32_t i = 0; i < n; ++i) std::memcpy(sk_.data() + (size_t)i * F, rs_[i]->sketch_in, F * 4);
    }
    if (rc == NIQKI_OK && kind == kInsert) {
      uint32_t first = 0;
      rc = e.insert(e.ctx, sk_.data(), n, &first);
      if (rc == NIQKI_OK)
        for (uint32_t i = 0; i < n; ++i)
          if (rs_[i]->gid_out) *rs_[i]->gid_out = first + i;
    }
    if (rc == NIQKI_OK && (kind == kQuery || kind == kQuerySeq)) {
      off_.assign((size_t)n + 1, 0);
      size_t cap = std::max<size_t>(hc_.size(), (size_t)n * 64);
      for (int attempt = 0; attempt < 2; ++attempt) {
        hc_.resize(cap);
        hg_.resize(cap);
        rc = e.query(e.ctx, sk_.data(), n, off_.data(), hc_.data(), hg_.data(), cap);
        if (rc != NIQKI_E_CAPACITY || attempt) break;
        cap = (size_t)off_[n];   // hit_off is exact whatever the capacity: once more with room for all
      }
      if (rc == NIQKI_OK)
        for (uint32_t i = 0; i < n; ++i) {
          const uint64_t lo = off_[i], k = off_[i + 1] - lo, w = std::min<uint64_t>(k, rs_[i]->capacity);
          *rs_[i]->n_hits = k;   // (may exceed the caller's capacity: the first `capacity` hits are written)
          if (w) {
            std::memcpy(rs_[i]->hit_counts, hc_.data() + lo, w * 4);
            std::memcpy(rs_[i]->hit_gids, hg_.data() + lo, w * 4);
          }
        }
    }
    for (Request *r : rs_) { r->rc = rc; r->ran = true; }
  }

  std::mutex m_;
  std::condition_variable cv_;
  std::vector<Request *> pending_;
  bool leader_ = false;
  uint64_t batches_ = 0, requests_ = 0, largest_ = 0;
  // leader-only scratch (one leader at a time; batch_ is read under the mutex after the batch, by the leader itself)
  std::vector<Request *> batch_, rs_;
  std::vector<uint8_t> seqs_;
  std::vector<uint64_t> rec_off_;
  std::vector<int32_t> sk_;
  std::vector<uint64_t> off_;
  std::vector<uint32_t> hc_, hg_;
};

}  // namespace nqc
