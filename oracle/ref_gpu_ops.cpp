// ref_gpu_ops.cpp -- the reference's OWN program with its three hot-path operators bound to libniqki_hip.so.
//
// TEST INFRASTRUCTURE ONLY, our own code: no line of the reference is copied or modified.  oracle/Makefile
// compiles the reference's main (src/niqki.cpp, where it lies) together with this file and links the result
// against oracle/_ref/libniqki_ref.so -- the reference's Index class with its file drivers, built as
// position-independent code, whose calls to
//
//     void         Index::compute_sketch(const string&, vector<int32_t>&) const     src/niqki_index.cpp:335-358
//     void         Index::insert_sketch(const vector<int32_t>&, uint32_t)           :362-370
//     query_output Index::query_sketch(const vector<int32_t>&) const                :633-687
//
// go through the PLT -- and against libniqki_hip.so.  This file DEFINES those three members; an executable's
// definition takes precedence over a shared library's, so the reference's own `omp parallel` record loops
// (:383-456, :505-540) -- threads, critical sections and all -- call the GPU.  It is INTEGRATION.md's "minimal
// patch" carried out without touching a reference file: oracle/_ref/niqki_ref_gpu is the reference's command
// line, option parser, file readers and writers on top of the C ABI's *_shared entry points.
//
// Also bound, because they walk the reference's own bucket vectors (which stay empty here): the counting loop of
// --matrix, Index::query_range (:570-610; the reference's output_matrix still formats the rows), and the payload of
// --dump, Index::dump_index_disk (:42-59).  --load needs no binding of its own: the reference's constructor fills its
// bucket vectors from the file, and the first bound call rebuilds the GPU index from them (bound_of).
#include "niqki_index.h"          // the reference's header: -I/root/reference/src
#include "../include/niqki_hip.h"
#include "../include/niqki_hip_bench.h"   // (niqki_shared_stats: the report line of NIQKI_REF_GPU_REPORT)

#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <zlib.h>

namespace {

struct Bound {
  niqki_index *h = nullptr;
  std::mutex m;
  std::map<uint32_t, std::vector<int32_t>> pending;   // inserted sketches by the id the reference gave them
  uint32_t flushed = 0;                               // ids [0, flushed) are in the GPU index
};

std::mutex g_m;
std::map<const Index *, Bound *> g_bound;

[[noreturn]] void die(const char *what, niqki_index *h) {
  std::fprintf(stderr, "niqki_ref_gpu: %s: %s\n", what, niqki_last_error(h));
  std::exit(3);
}

// at exit: what went through the library (so that a test can tell the binding really took: if the executable's
// definitions did not take precedence the reference's CPU code would answer, with the same text)
void report() {
  for (auto &kv : g_bound) {
    uint64_t batches = 0, requests = 0, largest = 0;
    niqki_shared_stats(kv.second->h, &batches, &requests, &largest);
    std::fprintf(stderr, "niqki_ref_gpu: %llu calls of the reference's operators answered by libniqki_hip.so in %llu batches (largest %llu), %u genomes indexed on the GPU\n",
                 (unsigned long long)requests, (unsigned long long)batches, (unsigned long long)largest, kv.second->flushed);
  }
}

Bound *bound_of(const Index *ix) {
  std::lock_guard<std::mutex> g(g_m);
  auto it = g_bound.find(ix);
  if (it != g_bound.end()) return it->second;
  if (g_bound.empty() && std::getenv("NIQKI_REF_GPU_REPORT")) std::atexit(report);
  Bound *b = new Bound();
  niqki_params p{};
  p.K = ix->K; p.S = ix->lF; p.W = ix->W; p.H = ix->H; p.min_score = ix->min_score; p.device = -1;
  if (niqki_create(&p, &b->h) != NIQKI_OK) die("niqki_create", nullptr);
  // An Index that came out of the reference's LOADING constructor (--load, :63-102) holds the dump in its bucket
  // vectors (in this binary nothing else ever fills them: insert_sketch is bound).  A densified sketch has a value
  // in every slot, so a look at the buckets of the first slots tells.
  bool loaded = false;
  for (uint64_t bkt = 0, n = std::min<uint64_t>(ix->F, 8) * (uint64_t)ix->fingerprint_range; bkt < n && !loaded; ++bkt)
    loaded = !ix->Buckets[bkt].empty();
  if (loaded) {
    // Bucket fp + slot * 2^W lists the genomes whose sketch has fp in that slot (:362-370): the sketches are read
    // back from the buckets and inserted in id order -- the GPU index is the loaded one.
    const uint64_t F_ = ix->F, R = (uint64_t)ix->fingerprint_range, N = ix->genome_numbers;
    std::vector<int32_t> sk(N * F_, -1);
    for (uint64_t bkt = 0; bkt < F_ * R; ++bkt)
      for (gid g : ix->Buckets[bkt]) sk[(uint64_t)g * F_ + bkt / R] = (int32_t)(bkt % R);
    if (niqki_insert(b->h, sk.data(), (uint32_t)N, NIQKI_MEM_HOST) != NIQKI_OK) die("niqki_insert (loaded index)", b->h);
    b->flushed = (uint32_t)N;
  }
  g_bound[ix] = b;
  return b;
}

// The reference hands out genome ids in its own critical section (:396-401, :486-490) and calls insert_sketch
// outside it, so with several threads the calls may arrive out of id order.  Sketches wait here until every
// smaller id has arrived, then go to the GPU in id order (through the entry point that other threads' sketch
// and query calls may overlap with).  Called with b->m held.
void flush_ready(Bound *b) {
  auto it = b->pending.begin();
  while (it != b->pending.end() && it->first == b->flushed) {
    uint32_t got = 0;
    if (niqki_insert_shared(b->h, it->second.data(), &got) != NIQKI_OK) die("niqki_insert_shared", b->h);
    if (got != b->flushed) die("genome ids out of step with the reference's", b->h);
    ++b->flushed;
    it = b->pending.erase(it);
  }
}

}  // namespace

void Index::compute_sketch(const string &reference, vector<int32_t> &sketch) const {
  Bound *b = bound_of(this);
  sketch.assign(F, -1);
  if (niqki_sketch_shared(b->h, (const uint8_t *)reference.data(), reference.size(), sketch.data()) != NIQKI_OK)
    die("niqki_sketch_shared", b->h);
}

void Index::insert_sketch(const vector<int32_t> &sketch, uint32_t genome_id) {
  Bound *b = bound_of(this);
  std::lock_guard<std::mutex> g(b->m);   // (inserts are rare next to sketching: one lock is enough)
  b->pending[genome_id] = sketch;
  flush_ready(b);
}

query_output Index::query_sketch(const vector<int32_t> &sketch) const {
  Bound *b = bound_of(this);
  {
    std::lock_guard<std::mutex> g(b->m);
    flush_ready(b);
    if (!b->pending.empty()) die("a genome id was never inserted", b->h);
  }
  std::vector<uint32_t> c(genome_numbers ? genome_numbers : 1), gg(genome_numbers ? genome_numbers : 1);
  uint64_t n = 0;
  if (niqki_query_shared(b->h, sketch.data(), &n, c.data(), gg.data(), genome_numbers) != NIQKI_OK) die("niqki_query_shared", b->h);
  query_output r;
  r.reserve(n);
  for (uint64_t i = 0; i < n; ++i) r.push_back({c[i], gg[i]});   // already (count desc, gid desc), :685
  return r;
}

// everything the reference has inserted so far is in the GPU index (single caller: main's thread)
static Bound *settled(const Index *ix) {
  Bound *b = bound_of(ix);
  std::lock_guard<std::mutex> g(b->m);
  flush_ready(b);
  if (!b->pending.empty()) die("a genome id was never inserted", b->h);
  return b;
}

// --matrix: the columns [begin, end) of the all-pairs hit counts, then the reference's own row formatting
void Index::query_range(uint32_t begin, uint32_t end) const {
  Bound *b = settled(this);
  const uint32_t batch = end - begin, N = genome_numbers;
  const uint64_t stride = NIQKI_ROW_STRIDE(N);
  std::vector<uint16_t> counts((size_t)batch * stride);
  if (niqki_matrix_range(b->h, begin, end, counts.data(), stride, NIQKI_MEM_HOST) != NIQKI_OK) die("niqki_matrix_range", b->h);
  query_output row;
  for (uint32_t t = 0; t < batch; ++t) {
    row.clear();
    for (uint32_t a = 0; a < N; ++a) {
      const uint32_t c = counts[(size_t)t * stride + a];
      if (c >= min_score) row.push_back({c, a});
    }
    output_matrix(row, filenames[begin + t]);   // the reference's (through the PLT)
  }
}

// --dump: the index payload from the GPU, the names behind it, gzip (the reference's reader takes any gzip stream)
void Index::dump_index_disk(const string &filestr) const {
  Bound *b = settled(this);
  uint64_t need = 0;
  if (niqki_export_dump(b->h, nullptr, 0, &need) != NIQKI_OK) die("niqki_export_dump (size)", b->h);
  std::vector<uint8_t> buf(need);
  if (niqki_export_dump(b->h, buf.data(), need, &need) != NIQKI_OK) die("niqki_export_dump", b->h);
  gzFile f = gzopen(filestr.c_str(), "wb1");
  if (!f) { std::fprintf(stderr, "niqki_ref_gpu: cannot write %s\n", filestr.c_str()); std::exit(3); }
  for (uint64_t off = 0; off < need; off += (1u << 30))
    gzwrite(f, buf.data() + off, (unsigned)std::min<uint64_t>(need - off, 1u << 30));
  for (const string &name : filenames) {
    gzwrite(f, name.data(), (unsigned)name.size());
    gzwrite(f, "\n", 1);
  }
  gzclose(f);
}
