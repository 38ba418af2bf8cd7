// tests/host_san/combiner_stress.cpp -- TEST INFRASTRUCTURE: niqki_amd/csrc/nq_combiner.h (the batching behind
// niqki_sketch_shared / _insert_shared / _query_shared / _query_sequence_shared) on a FAKE single-caller engine, to be
// built with -fsanitize=thread and with -fsanitize=address,undefined (tests/test_host_sanitizers.py).
//
// 64 threads issue random requests with random capacities on one combiner.  The fake engine
//   * asserts that it is never entered by two threads at once (the single-caller contract of a niqki_index),
//   * answers deterministically (every answer is checked by the thread that asked),
//   * fails whole calls now and then (an "early error": every request of that call must see the error, nobody hangs),
//   * reports NIQKI_E_CAPACITY with exact offsets like niqki_query (the combiner must retry with room for all),
// and, in the second half of the run, allocations made INSIDE Combiner::submit fail at random (operator new throws):
// requests may then end with NIQKI_E_NOMEM but every thread must come back, an insert that went through keeps its id,
// and no id is handed out twice.  The reference contract being reproduced: src/niqki_index.cpp:391-401, :479-490.
#include "../../niqki_amd/csrc/nq_combiner.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <random>
#include <thread>

namespace {

thread_local bool t_inject = false;           // this thread is inside Combiner::submit with failure injection on
std::atomic<uint64_t> g_allocs{0};
std::atomic<uint32_t> g_fail_every{0};        // 0 = never

}  // namespace

void *operator new(std::size_t n) {
  if (t_inject && g_fail_every.load(std::memory_order_relaxed)) {
    const uint64_t k = g_allocs.fetch_add(1, std::memory_order_relaxed);
    if (k % g_fail_every.load(std::memory_order_relaxed) == 0) throw std::bad_alloc();
  }
  void *p = std::malloc(n ? n : 1);
  if (!p) throw std::bad_alloc();
  return p;
}
void operator delete(void *p) noexcept { std::free(p); }
void operator delete(void *p, std::size_t) noexcept { std::free(p); }

namespace {

constexpr uint32_t F = 64;

struct Fake {
  std::atomic<int> inside{0};
  std::atomic<uint64_t> calls{0}, overlaps{0}, capacity_replies{0};
  std::vector<std::vector<int32_t>> genomes;   // inserted sketches, id = position
  std::atomic<uint32_t> fail_every{0};         // every k-th engine call fails (0 = never)
  bool enter() {
    if (inside.fetch_add(1) != 0) overlaps++;
    const uint64_t c = calls.fetch_add(1) + 1;
    const uint32_t k = fail_every.load();
    return !(k && c % k == 0);
  }
  void leave() { inside.fetch_sub(1); }
};
// one engine call: like the C ABI it stands for, the engine itself never throws (allocation failures are injected into
// the combiner's own code only)
struct Call {
  Fake *f;
  bool ok, inject;
  explicit Call(Fake *f_) : f(f_), ok(f_->enter()), inject(t_inject) { t_inject = false; }
  ~Call() { t_inject = inject; f->leave(); }
};

void sketch_of(const uint8_t *seq, uint64_t len, int32_t *out) {
  uint64_t h = 1469598103934665603ull;
  for (uint64_t i = 0; i < len; ++i) h = (h ^ seq[i]) * 1099511628211ull;
  for (uint32_t f = 0; f < F; ++f) {
    h = (h ^ (h >> 29)) * 0x9E3779B97F4A7C15ull + f;
    out[f] = (int32_t)(h % 1000);
  }
}

// hits of a sketch against n genomes: ids g with (s[0] + g) % 3 == 0, count 1 + (s[1] + g) % 7, descending (count, gid)
void hits_of(const int32_t *s, uint32_t n, std::vector<uint32_t> &hc, std::vector<uint32_t> &hg) {
  hc.clear();
  hg.clear();
  for (uint32_t c = 7; c >= 1; --c)
    for (uint32_t g = n; g-- > 0;)
      if ((uint32_t)(s[0] + (int32_t)g) % 3 == 0 && 1 + (uint32_t)(s[1] + (int32_t)g) % 7 == c) { hc.push_back(c); hg.push_back(g); }
}

int eng_sketch(void *ctx, const uint8_t *seqs, const uint64_t *rec_off, uint32_t n, int32_t *sk) {
  Call c((Fake *)ctx);
  if (c.ok)
    for (uint32_t i = 0; i < n; ++i) sketch_of(seqs + rec_off[i], rec_off[i + 1] - rec_off[i], sk + (size_t)i * F);
  return c.ok ? NIQKI_OK : NIQKI_E_HIP;
}
int eng_insert(void *ctx, const int32_t *sk, uint32_t n, uint32_t *first) {
  Call c((Fake *)ctx);
  Fake *f = c.f;
  if (c.ok) {
    *first = (uint32_t)f->genomes.size();
    for (uint32_t i = 0; i < n; ++i) f->genomes.emplace_back(sk + (size_t)i * F, sk + (size_t)(i + 1) * F);
  }
  return c.ok ? NIQKI_OK : NIQKI_E_HIP;
}
int eng_query(void *ctx, const int32_t *sk, uint32_t n, uint64_t *off, uint32_t *hc, uint32_t *hg, uint64_t cap) {
  Call call((Fake *)ctx);
  Fake *f = call.f;
  int rc = NIQKI_E_HIP;
  if (call.ok) {
    std::vector<uint32_t> c, g;
    uint64_t at = 0;
    off[0] = 0;
    const uint32_t N = (uint32_t)f->genomes.size();
    for (uint32_t i = 0; i < n; ++i) {
      hits_of(sk + (size_t)i * F, N, c, g);
      for (size_t j = 0; j < c.size(); ++j)
        if (at + j < cap) { hc[at + j] = c[j]; hg[at + j] = g[j]; }
      at += c.size();
      off[i + 1] = at;
    }
    rc = at > cap ? NIQKI_E_CAPACITY : NIQKI_OK;   // offsets exact either way, like niqki_query
    if (rc == NIQKI_E_CAPACITY) f->capacity_replies++;
  }
  return rc;
}

#define CHECK(cond, ...)                                                  \
  do {                                                                    \
    if (!(cond)) { std::fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); std::abort(); } \
  } while (0)

}  // namespace

int main(int argc, char **argv) {
  const int threads = argc > 1 ? std::atoi(argv[1]) : 64;
  const int ops = argc > 2 ? std::atoi(argv[2]) : 200;
  Fake fake;
  nqc::Combiner comb;
  nqc::Engine eng;
  eng.ctx = &fake;
  eng.F = F;
  eng.sketch = eng_sketch;
  eng.insert = eng_insert;
  eng.query = eng_query;
  // 300 genomes in, single caller: the queries below have ~100 hits each, far beyond the combiner's first 64 per query
  for (int g = 0; g < 300; ++g) {
    std::vector<int32_t> s(F, g);
    uint32_t first = 0;
    CHECK(eng_insert(&fake, s.data(), 1, &first) == NIQKI_OK && first == (uint32_t)g, "setup insert");
  }
  std::atomic<uint64_t> n_ok{0}, n_err{0}, n_nomem{0}, n_over_capacity{0};

  auto phase = [&](bool queries, bool inserts, uint32_t engine_fail_every, uint32_t alloc_fail_every) {
    fake.fail_every = engine_fail_every;
    g_fail_every = alloc_fail_every;
    std::vector<std::vector<uint32_t>> got_ids(threads);
    std::vector<std::thread> ts;
    const uint32_t N0 = (uint32_t)fake.genomes.size();
    for (int t = 0; t < threads; ++t)
      ts.emplace_back([&, t] {
        std::mt19937 rnd(1234 + t);
        std::vector<uint8_t> seq;
        std::vector<int32_t> sk(F), exp(F);
        std::vector<uint32_t> hc, hg, ehc, ehg;
        for (int k = 0; k < ops; ++k) {
          const uint32_t len = rnd() % 300;
          seq.resize(len);
          for (auto &b : seq) b = (uint8_t)("ACGT"[rnd() % 4]);
          sketch_of(seq.data(), len, exp.data());
          nqc::Request r;
          const uint32_t what = rnd() % (queries ? 3 : 1) + (inserts && rnd() % 2 ? 10 : 0);
          const uint64_t cap = (uint64_t[]){0, 1, 8, 64, 1000}[rnd() % 5];
          uint64_t n_hits = ~0ull;
          uint32_t gid = ~0u;
          hc.assign(cap ? cap : 1, 0xDEAD);
          hg.assign(cap ? cap : 1, 0xDEAD);
          if (what >= 10) { r.kind = nqc::kInsert; r.sketch_in = exp.data(); r.gid_out = &gid; }
          else if (what == 0) { r.kind = nqc::kSketch; r.seq = seq.data(); r.len = len; r.sketch_out = sk.data(); }
          else if (what == 1) { r.kind = nqc::kQuery; r.sketch_in = exp.data(); r.n_hits = &n_hits; r.hit_counts = hc.data(); r.hit_gids = hg.data(); r.capacity = cap; }
          else { r.kind = nqc::kQuerySeq; r.seq = seq.data(); r.len = len; r.n_hits = &n_hits; r.hit_counts = hc.data(); r.hit_gids = hg.data(); r.capacity = cap; }
          t_inject = true;
          const int rc = comb.submit(eng, r);
          t_inject = false;
          if (rc == NIQKI_E_NOMEM) { CHECK(alloc_fail_every != 0, "NIQKI_E_NOMEM without injected allocation failures"); n_nomem++; continue; }
          if (rc == NIQKI_E_HIP) { CHECK(engine_fail_every != 0, "engine error without injected engine failures"); n_err++; continue; }
          CHECK(rc == NIQKI_OK, "request ended with %d (never NIQKI_E_CAPACITY: that is the combiner's to handle)", rc);
          n_ok++;
          if (r.kind == nqc::kSketch) CHECK(sk == exp, "sketch of thread %d", t);
          else if (r.kind == nqc::kInsert) { CHECK(gid != ~0u && gid >= N0, "insert id"); got_ids[t].push_back(gid); }
          else {
            hits_of(exp.data(), N0, ehc, ehg);      // (no inserts run beside queries in this harness' query phases)
            CHECK(n_hits == ehc.size(), "n_hits %llu != %zu", (unsigned long long)n_hits, ehc.size());
            if (n_hits > cap) n_over_capacity++;
            const size_t w = std::min<size_t>(cap, ehc.size());
            for (size_t j = 0; j < w; ++j) CHECK(hc[j] == ehc[j] && hg[j] == ehg[j], "hit %zu of thread %d", j, t);
            if (w < hc.size() && cap) CHECK(hc[w] == 0xDEAD, "wrote beyond min(n_hits, capacity)");
          }
        }
      });
    for (auto &th : ts) th.join();
    // ids of successful inserts: each exactly once, and each is that sketch
    std::vector<int> seen(fake.genomes.size(), 0);
    for (auto &v : got_ids)
      for (uint32_t g : v) { CHECK(g < fake.genomes.size(), "id beyond the store"); seen[g]++; }
    for (size_t g = N0; g < seen.size(); ++g) CHECK(seen[g] <= 1, "id %zu handed out %d times", g, seen[g]);
  };

  phase(true, false, 0, 0);      // sketches + queries, everything works: every answer checked
  phase(false, true, 0, 0);      // sketches + inserts beside each other
  phase(true, false, 7, 0);      // every 7th engine call fails: the whole call's requests see it, nobody hangs
  phase(true, false, 0, 23);     // allocations inside submit fail at random
  phase(false, true, 5, 17);     // inserts with engine errors and allocation failures: no id twice
  g_fail_every = 0;
  uint64_t b = 0, rq = 0, lg = 0;
  comb.stats(&b, &rq, &lg);
  CHECK(fake.overlaps == 0, "the engine was entered by two threads at once (%llu times)", (unsigned long long)fake.overlaps.load());
  CHECK(fake.capacity_replies > 0 && n_over_capacity > 0, "the capacity retry never ran");
  CHECK(b > 0 && b < rq && lg >= 2, "callers were not combined: %llu batches, %llu requests, largest %llu", (unsigned long long)b,
        (unsigned long long)rq, (unsigned long long)lg);
  std::printf("combiner ok: %d threads x %d ops x 5 phases: %llu ok, %llu engine errors, %llu NOMEM; %llu batches for %llu requests (largest %llu), "
              "%llu capacity retries, %llu answers beyond the caller's capacity\n", threads, ops, (unsigned long long)n_ok.load(),
              (unsigned long long)n_err.load(), (unsigned long long)n_nomem.load(), (unsigned long long)b, (unsigned long long)rq,
              (unsigned long long)lg, (unsigned long long)fake.capacity_replies.load(), (unsigned long long)n_over_capacity.load());
  return 0;
}
